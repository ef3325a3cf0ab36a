#!/bin/bash
# aggregate tests, then config 4 sizes (default build)
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_api.py tests/test_gpu_round2.py tests/test_dist.py tests/test_gpu_fullsize.py -x -q -m gpu -k "aggregate or pairing_product or campaign or neutral or device_pointer or sharded or tree_and or config4 or chunk_bound" > gpurun_out/r3/t_agg.log 2>&1 || { tail -30 gpurun_out/r3/t_agg.log; exit 1; }
tail -2 gpurun_out/r3/t_agg.log
for sz in 262144 131072 65536 32768 16384 8192 4096 600; do python bench.py --config 4 --size $sz --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($sz, round(d[\"ms_per_step\"],2), d[\"kernel_ms\"])"; done
