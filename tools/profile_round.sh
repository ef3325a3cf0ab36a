#!/bin/bash
# Round profiles on the GPU box (run through gpurun from the repo root): rocprofv3 kernel statistics of the bench command and
# of all configs, PMC passes (HBM traffic, wait shares, instruction counts) of the bench command, the dominant kernels' HBM-side
# traffic as bench.py reads it (pmc_traffic.json), and the CPU legs of configs 1/3/4/5 (SKIP_CPU_LEGS=1 skips them).
# Outputs under gpurun_out/prof_$ROUND/ (ROUND defaults to r04); copy the summaries into profiles/ with the round's prefix.
set -e -o pipefail
ROUND=${ROUND:-r04}
OUT=gpurun_out/prof_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- python3 bench.py --steps 5 --warmup 2 --no-extras > $OUT/bench_noextras.json 2> $OUT/bench_noextras.err
echo "stats bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_all -- python3 bench.py --steps 3 --warmup 1 > $OUT/bench_all.json 2> $OUT/bench_all.err
echo "stats all done"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  d=$OUT/pmc_$i
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 --no-extras > /dev/null 2> $d.err
  echo "pmc pass $i ($c) done"
done
python3 tools/pmc_summary.py $OUT/pmc_* > $OUT/pmc_default.json
# the same wait / issue counters for the Bls12381G2Impl batch (k_prepare<2>, two-pass k_lines2s): two passes
if [ -z "$SKIP_G2IMPL" ]; then
  j=0
  for c in "SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    j=$((j+1))
    d=$OUT/pmcg2_$j
    rocprofv3 --pmc $c --output-format csv -d $d -- python3 bench.py --impl g2 --steps 2 --warmup 1 > /dev/null 2> $d.err
    echo "pmc g2impl pass $j ($c) done"
  done
  python3 tools/pmc_summary.py $OUT/pmcg2_* > $OUT/pmc_g2impl.json
fi
# what bench.py reads for roofline.traffic: FETCH_SIZE / WRITE_SIZE (KB) per launch, keyed by the library's profile names
python3 - $OUT $ROUND <<'PY'
import json, sys
out, rnd = sys.argv[1], sys.argv[2]
d = json.load(open(out + '/pmc_default.json'))
names = {'k_finalexp2s': 'k_finalexp2s', 'k_millerf2s': 'k_millerf2s', 'k_lines2s': 'k_lines2s', 'k_prepare<1>': 'k_prepare', 'k_finalexps': 'k_finalexps', 'k_miller2s': 'k_miller2s'}   # rocprof kernel name -> the library's profile id (blsgpu_profile_get)
res = {}
for k, v in d.items():
    if k in names and 'FETCH_SIZE' in v and 'WRITE_SIZE' in v and names[k] not in res:
        res[names[k]] = {'FETCH_SIZE': v['FETCH_SIZE'], 'WRITE_SIZE': v['WRITE_SIZE'], 'kernel': k,
                         'source': 'profiles/%s_pmc_default.json (rocprofv3 --pmc passes of bench.py --steps 2 --no-extras, n=65536; tools/profile_round.sh)' % rnd}
json.dump(res, open(out + '/pmc_traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
if [ -z "$SKIP_CPU_LEGS" ]; then
  python3 tools/bench_configs.py --cpu-seconds 6 > $OUT/configs_cpu_legs.jsonl 2> $OUT/configs_cpu_legs.err
  echo "cpu legs done"
fi
# keep only the summaries (the raw traces are large)
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +1M -delete
du -sh $OUT
