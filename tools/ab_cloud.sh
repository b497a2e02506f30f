#!/bin/bash
# Timing diagnostics of the fused point-cloud kernel (GPU box): the product build, the two-launch path, and the builds of
# tools/build (csrc/slx_cloud.hip -DSLX_CLOUD_EXP=N, tmp_ab/libslx_cloudexp<N>.so: results wrong, timing only), twice round-robin,
# then the kernel traces of both paths.
cd ${GRAFT_REPO_ROOT:-.}
for round in 1 2; do
  python tools/cloud_bench.py --passes 2 2>/dev/null
  python tools/cloud_bench.py --passes 0 2>/dev/null
  for N in 4 1 2; do [ -f tmp_ab/libslx_cloudexp$N.so ] && python tools/cloud_bench.py --passes 0 --lib tmp_ab/libslx_cloudexp$N.so 2>/dev/null; done
done
cd /tmp && export TMPDIR=/tmp
for P in 0 2; do
  rm -rf /tmp/cloudtrace$P
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cloudtrace$P -- python3 $GRAFT_REPO_ROOT/tools/cloud_bench.py --passes $P --reps 200 > /dev/null 2>&1
  f=$(ls /tmp/cloudtrace$P/*/*kernel_stats.csv | head -1); echo "== kernel stats, passes=$P"; head -1 $f; grep "slx_cloud" $f
done
