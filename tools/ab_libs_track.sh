#!/bin/bash
# A/B of several builds of libslx.so (tmp_ab/libslx_<name>.so) on the device-resident tracker bench, three rounds round-robin on one box.
# Usage: tools/ab_libs_track.sh name1 name2 ...
P=structured-light-calculation_amd/libslx.so
KEEP=$(mktemp /tmp/libslx_keep.XXXXXX.so)
cp $P $KEEP
trap 'cp $KEEP $P; rm -f $KEEP' EXIT      # whatever ends this script, the product library comes back
for round in 1 2 3; do
  for L in "$@"; do
    cp tmp_ab/libslx_$L.so $P || exit 1
    echo "== $L $(python tools/track_bench.py --frames 600 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.2f us  %.0f GB/s" % (d["us_per_frame"], d["achieved_GBps"]))')"
  done
done
