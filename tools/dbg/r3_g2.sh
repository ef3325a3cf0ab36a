#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_api.py -x -q -k "verify_batch_vs_c_oracle or ragged or lane_split or pairing2" > gpurun_out/r3/t_api2.log 2>&1 || { tail -30 gpurun_out/r3/t_api2.log; exit 1; }
tail -2 gpurun_out/r3/t_api2.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r3/bench_all.json 2> gpurun_out/r3/bench_all.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3/bench_all.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('kernel_ms'))
for k, v in d['other_configs'].items():
    print(k, v.get('error') or (round(v['value']), round(v['ms_per_step'], 3), v.get('kernel_ms')))
PY
