#!/bin/bash
# Builds tmp_ab/libslx_exp<N>.so for every experiment mask N given (SLX_EXP in csrc/slx_kernels.hip), for tools/ab.py "2@tmp_ab/libslx_exp<N>.so".
# CPU only (hipcc cross-compiles).  The product library is not touched.
set -e
cd "$(dirname "$0")/../structured-light-calculation_amd/csrc"
make -s
mkdir -p ../../tmp_ab
for N in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-format-security -I../../include -I. \
      -DSLX_EXP=$N -c slx_kernels.hip -o /tmp/slx_kernels_exp$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tmp_ab/libslx_exp$N.so /tmp/slx_kernels_exp$N.o slx_plan.o slx_track.o slx_gather.o slx_cloud.o slx_text.o slx_api.o slx_comm.o dynaframe.o sensor.o -L/opt/rocm/lib -lrccl
  echo built tmp_ab/libslx_exp$N.so
done
