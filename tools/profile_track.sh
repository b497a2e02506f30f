#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of the device-resident tracker bench (slx_track_fused_kernel<10>)
# and of the point-cloud bench.  Usage: tools/profile_track.sh <tag>
# Output: gpurun_out/prof_<tag>/{trace_stats.csv, pmc_*.csv} (rows of the slx_ kernels only) + the bench lines
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/track_bench.py --frames 300"
python3 $ROOT/tools/track_bench.py --frames 600 > $OUT/bench.json 2>/dev/null
rm -rf $OUT/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1); head -1 $f > $OUT/trace_stats.csv; grep "slx_" $f >> $OUT/trace_stats.csv
rm -rf $OUT/trace
i=0
for GROUP in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
             "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_WAVE_CYCLES" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $GROUP --output-format csv -d $OUT/pmc$i -- $CMD > $OUT/pmc$i.log 2>&1 || true
  f=$(ls $OUT/pmc$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then head -1 $f > $OUT/pmc$i.csv; grep "slx_track_fused" $f >> $OUT/pmc$i.csv || true; fi
  rm -rf $OUT/pmc$i
done
echo profiled $TAG
