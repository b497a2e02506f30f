#!/bin/bash
# try two RCCL ranks on the one GPU of the box (expected: RCCL refuses duplicate devices)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np
np.random.default_rng(1).integers(0,256,size=(4,12,37,128),dtype=np.uint8).tofile("/tmp/in.bin")
PY
rm -f /tmp/id.bin
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 5 60 tests/cpp/gather_host_loop 0 2 /tmp/id.bin rows 128 37 4 /tmp/in.bin /tmp/out0.bin > /tmp/r0.log 2>&1 &
P0=$!
timeout -k 5 60 tests/cpp/gather_host_loop 1 2 /tmp/id.bin rows 128 37 4 /tmp/in.bin /tmp/out1.bin > /tmp/r1.log 2>&1 &
P1=$!
wait $P0; echo "rank0 rc=$?"; wait $P1; echo "rank1 rc=$?"
for f in /tmp/r0.log /tmp/r1.log; do echo "== $f"; tail -n 4 $f; done
