#!/bin/bash
# config 4 sizes with the four-lane line kernel (default) and without (BLSGPU_LINES4_MAX=0)
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_api.py tests/test_gpu_round2.py tests/test_dist.py -x -q -m gpu -k "aggregate or pairing_product or campaign or neutral or device_pointer or sharded or tree_and" > gpurun_out/r3/t_agg.log 2>&1 || { tail -30 gpurun_out/r3/t_agg.log; exit 1; }
tail -2 gpurun_out/r3/t_agg.log
for mx in 40000 0; do for sz in 32768 31000 28672 24576 20480 16384 12288 8192 4096 600; do BLSGPU_LINES4_MAX=$mx python bench.py --config 4 --size $sz --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($mx, $sz, round(d[\"ms_per_step\"],2), d[\"kernel_ms\"][\"k_lines2\"])"; done; done
