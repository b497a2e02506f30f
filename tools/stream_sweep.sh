#!/bin/bash
# Same-box A/B of the stream kernel against the strip kernel (GPU box): the `big` threshold of the planner (C4, 4 ... 31 frame-sets,
# depth only).  Usage: tools/stream_sweep.sh  (prints one block per size)
cd ${GRAFT_REPO_ROOT:-.}
for n in 4 6 8 10 12 16 20 24 28 31; do
  echo "== C4 x $n frame-sets, depth only: strip (stream=1) vs stream (stream=2)"
  AB_SETS=$n AB_ROUNDS=5 python tools/ab.py "0:stream=1" "0:stream=2" 2>/dev/null
done
