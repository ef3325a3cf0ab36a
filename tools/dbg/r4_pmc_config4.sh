#!/bin/bash
# wait / issue shares and HBM-side traffic of config 4's kernels (k_prepare_agg, k_linesp, k_line_quad, k_f12_fold4): four --pmc passes of
# bench.py --config 4 at a quarter of the size (65 536 pairs: one chunk).  Writes gpurun_out/r4b/pmc_config4.json
set -e -o pipefail
OUT=gpurun_out/r4b/pmc4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python3 bench.py --config 4 --size ${SIZE:-65536} --steps 1 --warmup 1 > /dev/null 2> $OUT/p$i.err
  echo "pass $i done"
done
python3 tools/pmc_summary.py $OUT/p* > gpurun_out/r4b/pmc_config4.json
find $OUT -name "*counter_collection.csv" -size +1M -delete
