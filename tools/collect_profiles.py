#!/usr/bin/env python3
"""Copies the judged summaries out of gpurun_out/prof_<tag>/ (scratch) into profiles/ (tracked).
Usage: tools/collect_profiles.py <tag> <config> <sets_per_launch> <out_prefix>"""
import collections, csv, glob, json, os, re, sys
tag, config, sets, prefix = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
out = "profiles"
os.makedirs(out, exist_ok=True)
# the bench runs tools/membench as a child process, which the profiler traces into files of its own: take the decode's
stats = [f for f in sorted(glob.glob("gpurun_out/prof_%s/trace/*/*kernel_stats.csv" % tag), key=os.path.getmtime) if "slx_" in open(f).read()][-1]
rows = list(csv.reader(open(stats)))
csv.writer(open(os.path.join(out, prefix + "_kernel_stats.csv"), "w")).writerows([rows[0]] + [r for r in rows[1:] if "slx_" in r[0]])
trace = stats.replace("kernel_stats", "kernel_trace")
mine = [r for r in csv.DictReader(open(trace)) if "slx_" in r["Kernel_Name"]]
with open(os.path.join(out, prefix + "_kernel_trace_slx.csv"), "w") as g:
    w = csv.DictWriter(g, fieldnames=list(mine[0].keys())); w.writeheader(); w.writerows(mine)
vals = {}
for name in ["pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_sq4", "pmc_fetch", "pmc_write"]:
    acc = collections.defaultdict(list)
    path = "gpurun_out/prof_%s/%s.csv" % (tag, name)
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        vals[k] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
kernel = mine[0]["Kernel_Name"]
json.dump({"kernel": kernel, "workload": "%s x %d frame-sets per launch" % (config, sets),
           "command": "tools/profile_gpu.sh %s: rocprofv3 --kernel-trace --stats, then one --pmc pass per counter group, each around "
                      "`python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline ...`" % tag,
           "kernel_avg_ns_under_kernel_trace": sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mine) / len(mine),
           "counters": vals}, open(os.path.join(out, prefix + "_pmc_summary.json"), "w"), indent=1)
fetch = vals["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2
write = vals["WRITE_SIZE"]["mean_per_dispatch"] * 1024
tp = os.path.join(out, "traffic.json")
t = json.load(open(tp)) if os.path.exists(tp) else {}
t[config] = {"round": (re.match(r"r0*(\d+)", prefix) or [None, "?"])[1], "hbm_bytes_per_launch": fetch + write, "read_bytes": fetch, "write_bytes": write, "sets_per_launch": sets, "kernel": kernel,
             "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (%s_pmc_summary.json); FETCH_SIZE (KiB) x 1024 x 2 "
                       "(gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM), WRITE_SIZE (KiB) x 1024" % prefix}
json.dump(t, open(tp, "w"), indent=1)
print(config, "read %.1f MB write %.1f MB" % (fetch / 1e6, write / 1e6))
