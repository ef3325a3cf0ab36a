#!/bin/bash
# config 4 sizes with k_prepare_agg<1> on two lanes per item (default) and on one (BLSGPU_AGG_LANES=1)
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_api.py tests/test_gpu_round2.py tests/test_dist.py -x -q -m gpu -k "aggregate or pairing_product or campaign or neutral or device_pointer or sharded" > gpurun_out/r3/t_agg.log 2>&1 || { tail -30 gpurun_out/r3/t_agg.log; exit 1; }
tail -2 gpurun_out/r3/t_agg.log
for lanes in 2 1; do for sz in 262144 65536 32768 4096 600; do BLSGPU_AGG_LANES=$lanes python bench.py --config 4 --size $sz --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($lanes, $sz, round(d[\"ms_per_step\"],2), d[\"kernel_ms\"][\"k_prepare_agg\"])"; done; done
