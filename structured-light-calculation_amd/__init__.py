"""MI355X-native DynaFrame static depth path (structured-light decode -> depth).

Layout: csrc/ (HIP kernels + C ABI + the C++ mirror of the reference's decoder
classes), api.py (ctypes binding of include/slx.h), synth.py (synthetic fringe
source and the BASELINE.json workload configurations), shard.py (frame-set /
row-tile partition across ranks and the RCCL depth-map gather).
"""
