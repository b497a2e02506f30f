// no_gpu_launchers.cpp -- TEST BUILD ONLY (tests/cpp/host_sanitize): the library's own kernel launchers, answered with
// hipErrorNoDevice, so that the host side of libslx (slx_api.cpp, slx_comm.cpp, slx_plan.cpp, sensor.cpp, dynaframe.cpp) links
// without the device code and runs under AddressSanitizer / UndefinedBehaviorSanitizer on a box without a GPU.  Never part of
// libslx.so; nothing here computes anything.
#include <hip/hip_runtime_api.h>

#include "slx_kernels.h"

namespace { constexpr int kNoDevice = (int)hipErrorNoDevice; }

int slx_launch_fused(const SlxKParams &, int, bool, int, int, void *, const SlxTuning *, SlxStreamState *) { return kNoDevice; }
int slx_launch_cloud_count(const SlxKParams &, const double *, unsigned *, unsigned *, void *) { return kNoDevice; }
int slx_launch_cloud_write(const SlxKParams &, const double *, const unsigned *, const unsigned *, double *, unsigned *, unsigned *, void *) { return kNoDevice; }
int slx_launch_strip_regression(const uint8_t *, size_t, int, int, int, float *, float *, void *, const float *, const float *, float *) { return kNoDevice; }
int slx_launch_delta_p(const float *, const float *, const float *, const float *, size_t, float *, void *) { return kNoDevice; }
int slx_launch_track_update(const SlxKParams &, const float *, float *, double *, double *, double *, double *, double *, void *) { return kNoDevice; }
bool slx_track_fusable(int, int, int) { return false; }
int slx_launch_track_fused(const SlxKParams &, const uint8_t *, size_t, float *, float *, const float *, const float *, float *, double *, double *, double *,
                           double *, double *, void *) { return kNoDevice; }
int slx_launch_text(const double *, unsigned long long, unsigned *, unsigned *, unsigned, unsigned char *, unsigned long long *, unsigned long long *, int, unsigned long long *, void *) { return kNoDevice; }
int slx_launch_row_scatter(const SlxScatterSegs &, const double *, double *, void *) { return kNoDevice; }
int slx_launch_cloud_fused(const SlxCloudFused &, void *) { return kNoDevice; }
int slx_launch_text_lengths(const double *, const unsigned *, unsigned long long, unsigned *, unsigned *, unsigned, unsigned, unsigned long long *, unsigned long long *, int, unsigned long long *, void *) { return kNoDevice; }
int slx_launch_text_piece(const double *, unsigned long long, const unsigned *, unsigned char *, unsigned long long *, unsigned, unsigned, int, const unsigned long long *, void *) { return kNoDevice; }
