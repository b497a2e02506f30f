#!/bin/bash
# A/B of two builds of libslx.so (tmp_ab/libslx_<name>.so) on the tracker bench, alternating A B A B on one box.
P=structured-light-calculation_amd/libslx.so
KEEP=$(mktemp /tmp/libslx_keep.XXXXXX.so)
cp $P $KEEP
# whatever ends this script (a failing arm, a timeout, a signal) the product library comes back
trap 'cp $KEEP $P; rm -f $KEEP' EXIT
for L in $1 $2 $1 $2; do
  cp tmp_ab/libslx_$L.so $P || exit 1
  echo "== $L $(python tools/track_bench.py --frames 400 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.2f us  %.0f GB/s" % (d["us_per_frame"], d["achieved_GBps"]))')"
done
