#!/bin/bash
mkdir -p gpurun_out/r3
for ch in 65536 32768 16384; do
BLSGPU_MILLER_CHUNK=$ch python bench.py --config 4 --steps 3 --warmup 1 > gpurun_out/r3/bench_c4_$ch.json 2> gpurun_out/r3/bench_c4.err
done
BLSGPU_MILLER_V1=1 python bench.py --config 4 --steps 3 --warmup 1 > gpurun_out/r3/bench_c4_v1.json 2> gpurun_out/r3/bench_c4.err
python - <<'PY'
import json
for f in ('65536', '32768', '16384', 'v1'):
    d = json.loads(open('gpurun_out/r3/bench_c4_%s.json' % f).read().strip().splitlines()[-1])
    print(f, round(d['ms_per_step'], 2), d.get('kernel_ms'))
PY
