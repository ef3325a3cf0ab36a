#!/bin/bash
# quick GPU loop of round 4 (second session): the per-item parity subset, then the headline bench without the ride-along.
# usage (on the GPU box, from the repo root): bash tools/dbg/r4_quick.sh TAG [pytest -k expression]
set -o pipefail
TAG=${1:-q}
K=${2:-}
mkdir -p gpurun_out/r4b
if [ -n "$K" ]; then python -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/r4b/t_$TAG.log 2>&1; else python -m pytest tests/test_gpu_verify.py -m gpu -x -q > gpurun_out/r4b/t_$TAG.log 2>&1; fi
tail -3 gpurun_out/r4b/t_$TAG.log
python bench.py --no-extras > gpurun_out/r4b/bench_$TAG.json 2> gpurun_out/r4b/bench_$TAG.err || { tail -5 gpurun_out/r4b/bench_$TAG.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/r4b/bench_$TAG.json").read())
print(d["value"], d["ms_per_step"], d["kernel_ms"])
PY
