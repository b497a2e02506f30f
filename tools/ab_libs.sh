#!/bin/bash
# A/B of two builds of libslx.so on one GPU box: tools/ab_libs.sh "C4 C3" base new   (libs in tmp_ab/libslx_<name>.so)
# Alternates the builds (A B A B) so that box-to-box and drift effects cancel; prints tools/ab.py's medians.
P=structured-light-calculation_amd/libslx.so
KEEP=$(mktemp /tmp/libslx_keep.XXXXXX.so)
cp $P $KEEP
# whatever ends this script (a failing arm, a timeout, a signal) the product library comes back
trap 'cp $KEEP $P; rm -f $KEEP' EXIT
for C in $1; do
  for L in $2 $3 $2 $3; do
    cp tmp_ab/libslx_$L.so $P || exit 1
    echo "== $C $L"
    AB_CONFIG=$C AB_SETS=$([ $C = C5 ] && echo 4 || ([ $C = C3 ] && echo 16 || echo 32)) timeout -k 10 120 python tools/ab.py ${AB_ARMS:-2} 2>&1 | grep median || exit 1
  done
done
