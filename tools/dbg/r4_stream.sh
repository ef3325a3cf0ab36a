#!/bin/bash
# the streamed cut (k_pairing_stream) and the split Miller loop (k_pairing_post2) against the plain forms, same box, same call: their
# parity tests first, then config 3, config 5 (G1Impl) and small batches of both orientations with the knobs at 1 / 3 (default), 1 / 2 and 0 / 0.
# usage (on the GPU box, from the repo root): bash tools/dbg/r4_stream.sh
set -o pipefail
O=gpurun_out/r4b
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "streamed or multi_verify or verify_secure or g2impl or G2Impl or single" > $O/stream_tests.log 2>&1 || { tail -30 $O/stream_tests.log; exit 1; }
tail -2 $O/stream_tests.log
for pair in "1 3" "1 2" "0 0"; do
  set -- $pair; v=$1; w=$2
  echo "== BLSGPU_STREAM_LINES=$v BLSGPU_POST_SPLIT=$w"
  BLSGPU_STREAM_LINES=$v BLSGPU_POST_SPLIT=$w timeout -k 10 300 python bench.py --config 3 --steps 10 --warmup 3 > $O/stream_c3_$v$w.json 2> $O/stream_c3_$v$w.err || { tail -5 $O/stream_c3_$v$w.err; exit 1; }
  BLSGPU_STREAM_LINES=$v BLSGPU_POST_SPLIT=$w timeout -k 10 300 python bench.py --config 5 --variant g1m --steps 10 --warmup 3 > $O/stream_c5_$v$w.json 2> $O/stream_c5_$v$w.err || { tail -5 $O/stream_c5_$v$w.err; exit 1; }
  python - <<PY
import json
for c in ("c3", "c5"):
    d = json.loads(open("$O/stream_%s_$v$w.json" % c).read())
    print(c, d["ms_per_step"], d["kernel_ms"])
PY
  for sg in 2 1; do
    BLSGPU_STREAM_LINES=$v BLSGPU_POST_SPLIT=$w BLSGPU_WIDE_MAX=512 SMALL_SIZES=1,16,64 timeout -k 10 300 python tools/dbg/small.py $sg > $O/stream_small_${sg}_$v$w.txt 2>&1; grep " ms " $O/stream_small_${sg}_$v$w.txt
  done
done
