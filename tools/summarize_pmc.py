#!/usr/bin/env python3
"""Means per dispatch of every counter in gpurun_out/prof_<tag>/pmc*.csv + the kernel trace's average -> profiles/<prefix>_pmc_summary.json
Usage: tools/summarize_pmc.py <tag> <prefix> "<workload text>" """
import collections, csv, glob, json, os, sys
tag, prefix, what = sys.argv[1], sys.argv[2], sys.argv[3]
base = "gpurun_out/prof_%s" % tag
acc = collections.defaultdict(list)
kernel = None
for path in sorted(glob.glob(os.path.join(base, "pmc*.csv"))):
    for r in csv.DictReader(open(path)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        kernel = r["Kernel_Name"]
stats = list(csv.DictReader(open(os.path.join(base, "trace_stats.csv"))))
out = {"kernel": kernel, "workload": what, "kernel_stats": [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs")} for r in stats],
       "counters": {k: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for k, v in sorted(acc.items())}}
try:
    out["bench_line"] = json.load(open(os.path.join(base, "bench.json")))
except Exception:
    pass
json.dump(out, open(os.path.join("profiles", prefix + "_pmc_summary.json"), "w"), indent=1)
with open(os.path.join("profiles", prefix + "_kernel_stats.csv"), "w") as g:
    g.write(open(os.path.join(base, "trace_stats.csv")).read())
print(json.dumps(out["counters"], indent=1))
