#!/bin/bash
# FETCH_SIZE of C3 with and without the XCD remap
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
for d in 0 4; do
  SLX_DBG=$d timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/fx_$d -- python3 $ROOT/bench.py --config C3 --sets-per-gpu 16 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  f=$(ls $ROOT/gpurun_out/fx_$d/*/*counter_collection.csv | head -1)
  python3 - "$f" $d <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "slx_strip" in r["Kernel_Name"]]
v=[float(r["Counter_Value"]) for r in rows]
print("dbg",sys.argv[2],"FETCH_SIZE mean KiB",sum(v)/len(v), "-> read MB (x2):", sum(v)/len(v)*2048/1e6)
PY
  rm -rf $ROOT/gpurun_out/fx_$d
done
