#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag>/ (written by tools/profile_gpu.sh): kernel duration and PMC means."""
import collections, csv, glob, sys
for tag in sys.argv[1:]:
    print("==", tag)
    for f in glob.glob("gpurun_out/prof_%s/trace/*/*kernel_stats.csv" % tag):
        for r in csv.DictReader(open(f)):
            if "slx_" in r["Name"] and "lut_init" not in r["Name"]:
                print("  %s calls=%s avg_ns=%s" % (r["Name"][:70], r["Calls"], r["AverageNs"]))
    vals = {}
    for name in ["pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"]:
        try:
            rows = list(csv.DictReader(open("gpurun_out/prof_%s/%s.csv" % (tag, name))))
        except Exception as e:
            print("  ", name, "missing")
            continue
        acc = collections.defaultdict(list)
        for r in rows:
            if "lut_init" in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            vals[k] = sum(v) / len(v)
            print("   %-22s %.5g" % (k, vals[k]))
    if "SQ_ACTIVE_INST_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
        # ACTIVE_INST_VALU is in quad-cycles summed over waves; 1024 SIMDs; GRBM_GUI_ACTIVE summed over 8 XCDs
        cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
        print("   VALU busy fraction ~ %.3f (per-SIMD VALU cycles / kernel cycles)" % (vals["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc))
        print("   per-wave: VALU insts %.0f, VALU cycles %.0f, wave lifetime cycles %.0f" % (
            vals["SQ_INSTS_VALU"] / vals["SQ_WAVES"], vals["SQ_ACTIVE_INST_VALU"] * 4 / vals["SQ_WAVES"], vals["SQ_WAVE_CYCLES"] * 4 / vals["SQ_WAVES"]))
    if "FETCH_SIZE" in vals:
        print("   HBM read  = %.1f MB (FETCH_SIZE x 1024 x 2: gfx950 counts 128-B requests as 64 B)" % (vals["FETCH_SIZE"] * 1024 * 2 / 1e6))
    if "WRITE_SIZE" in vals:
        print("   HBM write = %.1f MB (WRITE_SIZE x 1024)" % (vals["WRITE_SIZE"] * 1024 / 1e6))
