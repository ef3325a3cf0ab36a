#!/bin/bash
# where the waits of the headline kernels come from: LDS (conflicts, wait-for-issue), vector memory (levels = in-flight instructions
# integrated over time: level / instructions = average latency), instruction fetch.  Writes gpurun_out/r4b/pmc_waits.json
set -e -o pipefail
OUT=gpurun_out/r4b/pmcw
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -- python3 bench.py --steps 2 --warmup 1 --no-extras > /dev/null 2> $OUT/p$i.err
  echo "pass $i done"
done
python3 tools/pmc_summary.py $OUT/p* > gpurun_out/r4b/pmc_waits.json
find $OUT -name "*counter_collection.csv" -size +1M -delete
