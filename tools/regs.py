#!/usr/bin/env python3
"""Register / spill report of every strip-kernel instantiation (CPU only: compiles the device code to assembly)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "structured-light-calculation_amd", "csrc")
out = "/tmp/slx_kernels_regs.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                       "-I" + src, "-S", "--cuda-device-only", os.path.join(src, "slx_kernels.hip"), "-o", out], stderr=subprocess.DEVNULL)
s = open(out).read()
only = sys.argv[1:] 
for m in re.finditer(r'\.name:\s+(\S*slx_strip_kernel\S*)\n(.*?)\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', s, re.S):
    name, blk = m.group(1), m.group(2)
    sg = re.search(r'\.sgpr_count:\s+(\d+)', blk).group(1)
    ss = re.search(r'\.sgpr_spill_count:\s+(\d+)', blk).group(1)
    t = re.search(r'slx_strip_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)', name).groups()
    tag = "mode%s F%s GB%s NS%s AUX%s" % t
    if only and not any(o in tag for o in only):
        continue
    if t[0] == "2" and t[1] != "1":
        continue
    print("%-28s vgpr %3s (spill %s)  sgpr %3s (spill %s)" % (tag, m.group(3), m.group(4), sg, ss))
