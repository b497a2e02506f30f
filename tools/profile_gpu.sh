#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of the bench workload.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
# Output: gpurun_out/prof_<tag>/{trace,pmc_sq,pmc_fetch,pmc_write}/...csv
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-traffic-probe --no-power-probe $@"
# kernel trace: the bench's own default step counts, so that the average launch duration includes the same
# clock ramp as the un-profiled bench line it is compared with
rm -rf $OUT/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs --no-traffic-probe --no-power-probe $@ > $OUT/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq3 -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_sq3.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_CYCLES --output-format csv -d $OUT/pmc_sq4 -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_sq4.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1
# keep only the rows of our kernel from the (large) per-dispatch counter files
for d in pmc_sq pmc_sq2 pmc_sq3 pmc_sq4 pmc_fetch pmc_write; do
  f=$(ls $OUT/$d/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then head -1 $f > $OUT/$d.csv; grep "slx_" $f >> $OUT/$d.csv || true; fi
done
rm -rf $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_sq3 $OUT/pmc_sq4 $OUT/pmc_fetch $OUT/pmc_write
echo profiled $TAG
