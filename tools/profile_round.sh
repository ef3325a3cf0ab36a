#!/bin/bash
# Round profiles on the GPU box (run through gpurun from the repo root): rocprofv3 kernel statistics of the bench command and
# of all configs, PMC passes (HBM traffic, wait share, VALU count) for the default build and for the 1-wave-per-SIMD build
# of the lane-split kernels (BLS_SPLIT_WAVES=1, agora-blsful_amd/libblsgpu_w1.so: build it first with tools/build_w1.py; skipped
# when the file is absent), and the CPU legs of configs 1/3/4/5 (SKIP_CPU_LEGS=1 skips them).
# Outputs under gpurun_out/prof_r02/; tools/pmc_summary.py turns the PMC directories into JSON.
set -e -o pipefail
OUT=gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
R=$PWD
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- python3 bench.py --steps 5 --warmup 2 --no-extras > $OUT/bench_noextras.json 2> $OUT/bench_noextras.err
echo "stats bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_all -- python3 bench.py --steps 3 --warmup 1 > $OUT/bench_all.json 2> $OUT/bench_all.err
echo "stats all done"
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  d=$OUT/pmc_default_$(echo $c | tr ' ' '_' | cut -c1-20)
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 --no-extras > /dev/null 2> $d.err
  echo "pmc default $c done"
done
if [ -f $R/agora-blsful_amd/libblsgpu_w1.so ]; then
export BLSGPU_LIB=$R/agora-blsful_amd/libblsgpu_w1.so
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  d=$OUT/pmc_w1_$(echo $c | tr ' ' '_' | cut -c1-20)
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 --no-extras > /dev/null 2> $d.err
  echo "pmc w1 $c done"
done
unset BLSGPU_LIB
python3 tools/pmc_summary.py $OUT/pmc_w1_* > $OUT/pmc_w1.json
fi
python3 tools/pmc_summary.py $OUT/pmc_default_* > $OUT/pmc_default.json
if [ -z "$SKIP_CPU_LEGS" ]; then
  python3 tools/bench_configs.py --cpu-seconds 6 > $OUT/configs_cpu_legs.jsonl 2> $OUT/configs_cpu_legs.err
  echo "cpu legs done"
fi
# keep only the summaries (the raw traces are large)
find $OUT -name "*_kernel_trace.csv" -delete
du -sh $OUT
