#!/bin/bash
# The round's record in ONE gpurun call on ONE box: the bench line as the driver runs it and with its defaults, the kernel trace and counter
# passes of the headline and of the reference's own case, one kernel trace of the whole bench command (every other_configs kernel), the three
# clocks of the short launches, the text call.  Usage: tools/final_round.sh <tag, e.g. r06>
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out
cd $ROOT
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_n1_driver_args.json 2> $O/${TAG}_bench_n1_driver_args.err
python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
tools/profile_gpu.sh ${TAG}c4 > $O/${TAG}_prof_c4.log 2>&1
tools/profile_gpu.sh ${TAG}ref --config REF --sets-per-gpu 32 > $O/${TAG}_prof_ref.log 2>&1
(cd /tmp && export TMPDIR=/tmp && rm -rf $O/${TAG}_all && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_all -- python3 $ROOT/bench.py --no-cpu-baseline --no-traffic-probe --no-power-probe > $O/${TAG}_bench_all.json 2> $O/${TAG}_bench_all.err; f=$(ls $O/${TAG}_all/*/*kernel_stats.csv | grep -v membench | head -1); for g in $O/${TAG}_all/*/*kernel_stats.csv; do if grep -q slx_ $g; then f=$g; fi; done; head -1 $f > $O/${TAG}_bench_all_kernel_stats.csv; grep slx_ $f >> $O/${TAG}_bench_all_kernel_stats.csv; rm -rf $O/${TAG}_all)
tools/profile_short.sh > $O/${TAG}_short.log 2>&1
timeout -k 10 200 python tools/text_bench.py > $O/${TAG}_text_bench.json 2> $O/${TAG}_text_bench.err
echo final round record done
