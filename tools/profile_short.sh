#!/bin/bash
# Runs on the GPU box (via gpurun): the three clocks of the short launches on ONE box (tools/short_kernels.py), then the same loops under
# rocprofv3 --kernel-trace (no in-kernel stamps there).  Output: gpurun_out/short/{clocks.json, trace_kernel_trace.csv, trace_kernel_stats.csv}
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/short
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/short_kernels.py > $OUT/clocks.json 2> $OUT/clocks.err
rm -rf $OUT/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/short_kernels.py --no-stamps > $OUT/clocks_under_profiler.json 2> $OUT/trace.log
f=$(ls $OUT/trace/*/*kernel_trace.csv | head -1); head -1 $f > $OUT/trace_kernel_trace.csv; grep "slx_" $f >> $OUT/trace_kernel_trace.csv
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1); head -1 $f > $OUT/trace_kernel_stats.csv; grep "slx_" $f >> $OUT/trace_kernel_stats.csv
rm -rf $OUT/trace
echo profiled short kernels
