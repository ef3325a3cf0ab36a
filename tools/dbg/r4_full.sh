#!/bin/bash
# the whole GPU suite, then the default bench run (all configs ride along); usage: bash tools/dbg/r4_full.sh TAG
set -o pipefail
TAG=${1:-full}
mkdir -p gpurun_out/r4b
python -m pytest tests -m gpu -x -q > gpurun_out/r4b/gputests_$TAG.log 2>&1
rc=$?
tail -4 gpurun_out/r4b/gputests_$TAG.log
[ $rc -eq 0 ] || exit $rc
python bench.py > gpurun_out/r4b/bench_full_$TAG.json 2> gpurun_out/r4b/bench_full_$TAG.err || { tail -5 gpurun_out/r4b/bench_full_$TAG.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/r4b/bench_full_$TAG.json").read())
print(d["value"], d["ms_per_step"], d["kernel_ms"], d.get("valu_roofline"))
for k, v in d.get("other_configs", {}).items():
    print(" ", k, v.get("value"), v.get("ms_per_step"), v.get("error"))
PY
