#!/bin/bash
# single-verdict latency of two builds on one box, alternating: usage bash tools/dbg/r4_ab_single.sh LIB_A LIB_B (paths from the repo root;
# A/B libraries live under tools/ab/, which git ignores).  small.py at 1 and 16 items, both orientations, three rounds each.
set -o pipefail
O=gpurun_out/r4b
mkdir -p $O
for round in 1 2 3; do
  for lib in "$1" "$2"; do
    for sg in 1 2; do
      echo "== $lib sg=$sg round $round"
      BLSGPU_LIB=$PWD/$lib SMALL_SIZES=1,16 timeout -k 10 200 python tools/dbg/small.py $sg 2>&1 | grep " ms "
    done
  done
done | tee $O/ab_single.txt
