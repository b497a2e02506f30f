#!/bin/bash
# The long differential-fuzz pass of a round (GPU box): every fuzzer and profile with fresh seeds, fixed case counts, one log.
# Usage: tools/fuzz_long.sh <seed base> > gpurun_out/fuzz.log
cd ${GRAFT_REPO_ROOT:-.}
S=${1:-500}
python tools/fuzz_parity.py 0 $((S+1)) --cases 12000 2>&1 | grep -v amdgpu.ids
FUZZ_PROFILE=strip python tools/fuzz_parity.py 0 $((S+2)) --cases 1500 2>&1 | grep -v amdgpu.ids
FUZZ_PROFILE=big python tools/fuzz_parity.py 0 $((S+3)) --cases 5000 2>&1 | grep -v amdgpu.ids
FUZZ_PROFILE=bigstrip python tools/fuzz_parity.py 0 $((S+4)) --cases 6000 2>&1 | grep -v amdgpu.ids
FUZZ_PROFILE=calib python tools/fuzz_parity.py 0 $((S+5)) --cases 15000 2>&1 | grep -v amdgpu.ids
FUZZ_PROFILE=refstream python tools/fuzz_parity.py 0 $((S+8)) --cases 400 2>&1 | grep -v amdgpu.ids
python tools/fuzz_track.py 0 $((S+6)) --cases 5000 2>&1 | grep -v amdgpu.ids
python tools/fuzz_api.py 0 $((S+7)) --cases 8000 2>&1 | grep -v amdgpu.ids
