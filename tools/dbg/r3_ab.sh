#!/bin/bash
# round 3 A/B on the GPU box: lane-split parity tests, then the bench with the two-kernel Miller loop and with the one-kernel loop
set -o pipefail
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_api.py -x -q -k "verify_batch_vs_c_oracle or ragged or lane_split or large_property" > gpurun_out/r3/t_api.log 2>&1 || { tail -30 gpurun_out/r3/t_api.log; exit 1; }
tail -2 gpurun_out/r3/t_api.log
python -m pytest tests/test_gpu_fullsize.py -x -q -k config2 > gpurun_out/r3/t_full.log 2>&1 || { tail -30 gpurun_out/r3/t_full.log; exit 1; }
tail -2 gpurun_out/r3/t_full.log
python bench.py --steps 10 --warmup 3 --no-extras > gpurun_out/r3/bench_v2.json 2> gpurun_out/r3/bench_v2.err || { tail -20 gpurun_out/r3/bench_v2.err; exit 1; }
BLSGPU_FINALEXP_SEG=0 python bench.py --steps 10 --warmup 3 --no-extras > gpurun_out/r3/bench_v1.json 2> gpurun_out/r3/bench_v1.err
python - <<'PY'
import json
for f in ('v2', 'v1'):
    d = json.loads(open('gpurun_out/r3/bench_%s.json' % f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d.get('kernel_ms'))
PY
