#!/bin/bash
# the streamed cut (k_pairing_stream) against the two launches it replaces, same box, same call: its parity tests first, then
# config 3, config 5 (G1Impl) and Bls12381G2Impl small batches with BLSGPU_STREAM_LINES = 1 and 0.
# usage (on the GPU box, from the repo root): bash tools/dbg/r4_stream.sh
set -o pipefail
O=gpurun_out/r4b
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "streamed or multi_verify or verify_secure or g2impl or G2Impl or single" > $O/stream_tests.log 2>&1 || { tail -30 $O/stream_tests.log; exit 1; }
tail -2 $O/stream_tests.log
for v in 1 0; do
  echo "== BLSGPU_STREAM_LINES=$v"
  BLSGPU_STREAM_LINES=$v timeout -k 10 300 python bench.py --config 3 --steps 10 --warmup 3 > $O/stream_c3_$v.json 2> $O/stream_c3_$v.err || { tail -5 $O/stream_c3_$v.err; exit 1; }
  BLSGPU_STREAM_LINES=$v timeout -k 10 300 python bench.py --config 5 --variant g1m --steps 10 --warmup 3 > $O/stream_c5_$v.json 2> $O/stream_c5_$v.err || { tail -5 $O/stream_c5_$v.err; exit 1; }
  python - <<PY
import json
for c in ("c3", "c5"):
    d = json.loads(open("$O/stream_%s_$v.json" % c).read())
    print(c, d["ms_per_step"], d["kernel_ms"])
PY
  BLSGPU_STREAM_LINES=$v BLSGPU_WIDE_MAX=512 timeout -k 10 300 python tools/dbg/small.py 2 2>&1 | head -4 > $O/stream_small_$v.txt; cat $O/stream_small_$v.txt
done
