#!/bin/bash
# the whole GPU suite, then the bench
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu > gpurun_out/r3/suite.log 2>&1; rc=$?
tail -5 gpurun_out/r3/suite.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 10 --warmup 3 --no-extras > gpurun_out/r3/bench_now.json 2> gpurun_out/r3/bench_now.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3/bench_now.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('kernel_ms'))
PY
