#!/bin/bash
# PMC passes of the bench command (lane-split kernels): instruction counts, wait shares, busy cycles, memory instructions
set -o pipefail
OUT=gpurun_out/r3/pmc3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python3 bench.py --steps 2 --warmup 1 --no-extras > /dev/null 2> $OUT/p$i.err || { echo "pass $i failed"; tail -5 $OUT/p$i.err; }
  echo "pass $i done"
done
python3 tools/pmc_summary.py $OUT/p* > gpurun_out/r3/pmc_v3.json
find $OUT -name "*.csv" -size +2M -delete
cat gpurun_out/r3/pmc_v3.json | head -80
