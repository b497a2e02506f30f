#!/usr/bin/env python3
"""Writes tests/golden/static_parameters.json from the COMPILED reference translation unit R/StaticParameters.cpp
(oracle/_ref/libdynaframe_static.so, built by `make -C oracle ref` from the source where it lies).  Data, not source: the
values of the reference's compiled-in constants.  Run in the container that holds /root/reference."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
import ref_static

assert ref_static.available(), "oracle/_ref/libdynaframe_static.so was not built: is /root/reference present?"
out = {"source": "DynaFrame/DynaFrame/StaticParameters.cpp of the reference, compiled by g++ (oracle/Makefile: ref), constants read with ctypes",
       "constants": ref_static.constants()}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "static_parameters.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out["constants"], sort_keys=True))
