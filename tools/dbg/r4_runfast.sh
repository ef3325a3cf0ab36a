#!/bin/bash
# round 4, VERDICT r3 next #4 (a): the interpreter's run fast path (-DWIDE_RUN_FASTPATH: the table rows of a repeated operation stay in
# registers) against the default engine on the same box: per-operation step times, single-verify latency, the engine parity tests on the variant
mkdir -p gpurun_out/r4
for lib in default runfast; do
  if [ $lib = default ]; then unset BLSGPU_LIB; else export BLSGPU_LIB=$PWD/agora-blsful_amd/libblsgpu_runfast.so; fi
  echo "== $lib"
  python tools/dbg/engine_ops.py
  python tools/dbg/lat1.py
done > gpurun_out/r4/runfast.txt 2>&1
BLSGPU_LIB=$PWD/agora-blsful_amd/libblsgpu_runfast.so python -m pytest tests/test_gpu_wide.py tests/test_gpu_verify.py -x -q -m gpu >> gpurun_out/r4/runfast.txt 2>&1
tail -30 gpurun_out/r4/runfast.txt
