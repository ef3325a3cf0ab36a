#!/bin/bash
OUT=gpurun_out/r3/pmc5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python3 bench.py --steps 2 --warmup 1 --no-extras > /dev/null 2> $OUT/p$i.err || tail -3 $OUT/p$i.err
done
python3 tools/pmc_summary.py $OUT/p* > gpurun_out/r3/pmc_v5.json
find $OUT -name "*.csv" -size +1M -delete
