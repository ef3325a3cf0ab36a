#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_api.py tests/test_gpu_round2.py tests/test_gpu_fullsize.py -x -q -k "aggregate or pairing_product or campaign or neutral or config4 or device_pointer" > gpurun_out/r3/t_agg.log 2>&1 || { tail -30 gpurun_out/r3/t_agg.log; exit 1; }
tail -2 gpurun_out/r3/t_agg.log
python bench.py --config 4 --steps 3 --warmup 1 > gpurun_out/r3/bench_c4.json 2> gpurun_out/r3/bench_c4.err || tail -5 gpurun_out/r3/bench_c4.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3/bench_c4.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('kernel_ms'))
PY
BLSGPU_ROW_PAD=0 python bench.py --config 4 --steps 3 --warmup 1 > gpurun_out/r3/bench_c4_nopad.json 2> gpurun_out/r3/bench_c4.err
python bench.py --steps 5 --warmup 2 --no-extras > gpurun_out/r3/bench_pad.json 2>/dev/null
BLSGPU_ROW_PAD=0 python bench.py --steps 5 --warmup 2 --no-extras > gpurun_out/r3/bench_nopad.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench_c4_nopad', 'bench_pad', 'bench_nopad'):
    d = json.loads(open('gpurun_out/r3/%s.json' % f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d.get('kernel_ms'))
PY
