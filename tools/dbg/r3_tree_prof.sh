#!/bin/bash
# per-kernel times of AggregateSignature::verify (config 4) at one size: SIZE=262144 bash tools/dbg/r3_tree_prof.sh
mkdir -p gpurun_out/r3
SIZE=${SIZE:-262144}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/tree_prof_$SIZE -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --size $SIZE --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r3/tree_prof_$SIZE.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r3/tree_prof_$SIZE -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), '%9.3f ms avg' % (float(r['AverageNs'])/1e6), '%9.2f ms total' % (float(r['TotalDurationNs'])/1e6), r['Percentage'])
PY
tail -1 gpurun_out/r3/tree_prof_$SIZE.log | cut -c1-300
