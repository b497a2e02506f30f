#!/usr/bin/env python3
"""gpurun_out/short/ (tools/profile_short.sh: ONE box) -> profiles/<prefix>_short_kernels.json: per short launch, side by side,
  wall_us / events_us        host clock and HIP events over N back-to-back calls (what bench.py's entries are)
  stamps_us                  s_memrealtime in the kernel: first workgroup's start to last workgroup's end of one launch
  kernel_trace               rocprofv3 --kernel-trace of the same loops: average / median interval, and for consecutive dispatches the
                             start-to-start period and the gap between one interval's end and the next one's begin
Usage: tools/collect_short.py <prefix>"""
import collections, csv, json, os, statistics, sys
prefix = sys.argv[1]
base = "gpurun_out/short"
clocks = json.load(open(os.path.join(base, "clocks.json")))
under = json.load(open(os.path.join(base, "clocks_under_profiler.json")))
rows = list(csv.DictReader(open(os.path.join(base, "trace_kernel_trace.csv"))))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
match = {"tracker": "slx_track_fused_kernel", "cloud": "slx_cloud_fused_kernel", "c4x1": "slx_strip_kernel<3, 3, 0, 4, false>"}
out = {"method": __doc__, "n_back_to_back": clocks["n_back_to_back"], "m_stamped_launches": clocks["m_stamped_launches"]}
for key, needle in match.items():
    name = next(k for k in by if needle in k)
    v = sorted(by[name])
    dur = [e - s for s, e in v]
    per = [v[i + 1][0] - v[i][0] for i in range(len(v) - 1)]
    gap = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    b2b = [i for i in range(len(v) - 1) if per[i] < 2.5 * statistics.median(dur)]
    ent = dict(clocks[key])
    ent["kernel_trace"] = {"kernel": name, "dispatches": len(v), "average_us": sum(dur) / len(dur) / 1e3, "median_us": statistics.median(dur) / 1e3,
                           "consecutive_pairs": len(b2b),
                           "start_to_start_median_us": statistics.median([per[i] for i in b2b]) / 1e3 if b2b else None,
                           "gap_end_to_next_begin_median_us": statistics.median([gap[i] for i in b2b]) / 1e3 if b2b else None}
    ent["under_the_profiler"] = {"wall_us": under[key]["wall_us"], "events_us": under[key]["events_us"]}
    ab = ent["algorithmic_bytes"]
    ent["frac_of_8TBps_by_kernel_trace_average"] = ab / ent["kernel_trace"]["average_us"] / 1e6 / 8.0
    out[key] = ent
json.dump(out, open(os.path.join("profiles", prefix + "_short_kernels.json"), "w"), indent=1)
with open(os.path.join("profiles", prefix + "_short_kernel_stats.csv"), "w") as g:
    g.write(open(os.path.join(base, "trace_kernel_stats.csv")).read())
for key in match:
    e = out[key]
    print("%-8s wall %.2f  events %.2f  trace avg %.2f (start-to-start %.2f, gap %.2f)  in-kernel %.2f us" % (
        key, e["wall_us"], e["events_us"], e["kernel_trace"]["average_us"], e["kernel_trace"]["start_to_start_median_us"] or 0,
        e["kernel_trace"]["gap_end_to_next_begin_median_us"] or 0, e["stamps_us"]))
