#!/bin/bash
# instruction-cache counters of the headline kernels (bench.py --no-extras)
mkdir -p gpurun_out/r3/icache
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 -L > gpurun_out/r3/icache/avail.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INSTS_VALU[A-Z_0-9]*" gpurun_out/r3/icache/avail.txt | sort -u | tr '\n' ' '
echo
i=0
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" ; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/r3/icache/p$i -- python3 bench.py --steps 2 --warmup 1 --no-extras > /dev/null 2> gpurun_out/r3/icache/p$i.err || tail -3 gpurun_out/r3/icache/p$i.err
done
python3 tools/pmc_summary.py gpurun_out/r3/icache/p* > gpurun_out/r3/icache/summary.json 2>/dev/null
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3/icache/summary.json'))
for k,v in d.items():
    if any(x in k for x in ('k_finalexp2s','k_millerf2s','k_lines2s','k_prepare<1>')):
        print(k, v)
PY
find gpurun_out/r3/icache -name "*counter_collection.csv" -size +1M -delete
