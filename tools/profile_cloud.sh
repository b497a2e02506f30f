#!/bin/bash
# GPU: kernel trace + PMC passes (SQ_*, FETCH_SIZE, WRITE_SIZE) of the point-cloud bench (slx_cloud_fused_kernel) and of the text bench
# (slx_text_len_kernel / slx_text_emit_kernel); means per dispatch -> gpurun_out/prof_<tag>/summary.txt.  Usage: tools/profile_cloud.sh <tag>
TAG=${1:-cloud}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/summary.txt
for WHAT in "cloud_bench.py --reps 100" "text_bench.py --config C4 --reps 6"; do
  CMD="python3 $ROOT/tools/$WHAT"
  echo "== $WHAT" >> $OUT/summary.txt
  rm -rf /tmp/pc_trace
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_trace -- $CMD > /dev/null 2>&1
  f=$(grep -l "slx_" /tmp/pc_trace/*/*kernel_stats.csv | head -1); head -1 $f >> $OUT/summary.txt; grep "slx_cloud\|slx_text" $f >> $OUT/summary.txt
  for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/pc_pmc
    timeout -k 10 300 rocprofv3 --pmc $G --output-format csv -d /tmp/pc_pmc -- $CMD > /dev/null 2>&1
    f=$(ls /tmp/pc_pmc/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" >> $OUT/summary.txt <<'PY'
import collections, csv, re, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"slx_(cloud_\w+|text_\w+)_kernel", r["Kernel_Name"])
    if m:
        acc[(m.group(0), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-28s %-22s mean per dispatch %14.1f  (%d dispatches)" % (k, c, sum(v) / len(v), len(v)))
PY
  done
done
cat $OUT/summary.txt
