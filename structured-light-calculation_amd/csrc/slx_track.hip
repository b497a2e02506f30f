// slx_track.hip -- the dynamic-frame tracker of DynaFrame for gfx950.
//
//   CCalculation::StripRegression      R/CCalculation.cpp:789-892  -> slx_strip_regression_kernel
//   CCalculation::FillOtherDeltaProU   R/CCalculation.cpp:595-663  -> slx_delta_p_kernel + slx_track_update_kernel
//   CCalculation::FillCoordinate(fN)   R/CCalculation.cpp:666-785  -> inside slx_track_update_kernel (deltaZ: :772-775)
// cv::blur (OpenCV 2.4.9 boxFilter) is restated: normalised 3x3, BORDER_REFLECT_101, double sums scaled by 1./9.
// All sums here are sums of small integers held in floats/doubles and therefore exact in any order.
// R/ = DynaFrame/DynaFrame/ of the reference repository.
#include <hip/hip_runtime.h>

#include "slx_kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int kTile = 256;            // threads per workgroup = columns per tile, halo included
constexpr int kRowsPerBand = 64;

// One lane per column keeps the 21-row sliding sum of its column while the workgroup walks down a band of rows;
// every row's sums go through LDS so that a lane can scan its 20 horizontal neighbours (win/2 to the left,
// win/2 - 1 to the right, in the reference's order: the centre wins ties, then the leftmost).
__global__ __launch_bounds__(kTile) void slx_strip_regression_kernel(const uint8_t *cam, size_t stride, int W, int H, int win,
                                                                     float *stripW, float *stripB)
{
    __shared__ float row_sum[2][kTile];
    const int hw = win / 2;
    const int out_cols = kTile - 2 * hw;
    const int tx = threadIdx.x;
    // lane tx holds column tile_first + tx; lanes hw .. kTile-hw-1 produce output, i.e. a tile yields the out_cols
    // interior columns from blockIdx.x * out_cols + hw on (halo lanes may be outside the image)
    const int c = blockIdx.x * out_cols + tx;
    const bool col_interior = c >= hw && c < W - hw;               // valSum is 0 elsewhere (R/CCalculation.cpp:799-802)
    const int h0 = hw + blockIdx.y * kRowsPerBand;
    const int h1 = h0 + kRowsPerBand < H - hw ? h0 + kRowsPerBand : H - hw;
    if (h0 >= h1) return;
    float sum = 0.f;
    if (col_interior)
        for (int r = h0 - hw; r <= h0 + hw; r++) sum += (float)cam[(size_t)r * stride + c];
    for (int h = h0; h < h1; h++) {
        float *buf = row_sum[(h - h0) & 1];
        buf[tx] = col_interior ? sum : 0.f;
        __syncthreads();
        if (tx >= hw && tx < kTile - hw && col_interior) {
            float mx = buf[tx], mn = mx, mxi = 0.f, mni = 0.f;
            for (int i = -hw; i < hw; i++) {                       // :838-851
                const float v = buf[tx + i];
                if (v > mx) { mx = v; mxi = (float)i; }
                if (v < mn) { mn = v; mni = (float)i; }
            }
            stripB[(size_t)h * W + c] = mni;
            stripW[(size_t)h * W + c] = mxi;
        }
        if (col_interior && h + 1 < h1)                            // :820-822
            sum = sum - (float)cam[(size_t)(h - hw) * stride + c] + (float)cam[(size_t)(h + hw + 1) * stride + c];
        // the other LDS buffer is written next; this one is read again only two rows later
    }
}

// R/CCalculation.cpp:602-617
__global__ __launch_bounds__(256) void slx_delta_p_kernel(const float *W0, const float *B0, const float *W1, const float *B1, size_t n, float *raw)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float f0W = W0[i], f0B = B0[i], f1W = W1[i], f1B = B1[i];
    const float dB = f0B - f1B, dW = f0W - f1W;
    raw[i] = (__builtin_fabsf(dB) < __builtin_fabsf(dW)) ? dB : dW;
}

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// blur 3x3 (:650), U += deltaP (:656-658), FillCoordinate(fN) (:672-708, :756-771), deltaZ (:772-775)
__global__ __launch_bounds__(256) void slx_track_update_kernel(const float *raw, float *deltaP, double *U, double *z, double *x, double *y,
                                                               double *deltaZ, const SlxKParams p)
{
    const int W = p.width, H = p.height;
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= W || v >= H) return;
    double s = 0.0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) s += (double)raw[(size_t)reflect101(v + dy, H) * W + reflect101(u + dx, W)];
    const float dp = (float)(s * (1. / 9));
    const size_t i = (size_t)v * W + u;
    deltaP[i] = dp;
    const double Uv = U[i] + (double)dp;
    U[i] = Uv;
    const double uc = (double)u - p.cx, vc = (double)(v + p.row_offset) - p.cy;
    const double cC = ((uc * p.fv) * p.P00 + (vc * p.fu) * p.P01) + p.K1;
    const double cD = ((uc * p.fv) * p.P20 + (vc * p.fu) * p.P21) + p.K2;
    double zz = -(p.cA - p.cB * Uv) / (cC - cD * Uv);
    if ((zz < p.fov_min) || (zz > p.fov_max)) zz = 0.0;
    if (Uv == 0.0) zz = 0.0;                                        // the reference leaves z untouched here; defined 0
    deltaZ[i] = zz - z[i];
    z[i] = zz;
    if (x) x[i] = zz * uc / p.fu;
    if (y) y[i] = zz * vc / p.fv;
}

}  // namespace

int slx_launch_strip_regression(const uint8_t *cam, size_t stride, int W, int H, int win, float *stripW, float *stripB, void *stream)
{
    const int hw = win / 2;
    if (win < 3 || 2 * hw >= kTile - 1 || H <= 2 * hw || W <= 2 * hw) return 0;     // no interior: the strips stay 0
    const int out_cols = kTile - 2 * hw;
    const dim3 grid((unsigned)((W - 2 * hw + out_cols - 1) / out_cols), (unsigned)((H - 2 * hw + kRowsPerBand - 1) / kRowsPerBand));
    hipLaunchKernelGGL(slx_strip_regression_kernel, grid, dim3(kTile), 0, (hipStream_t)stream, cam, stride, W, H, win, stripW, stripB);
    return (int)hipGetLastError();
}

int slx_launch_delta_p(const float *W0, const float *B0, const float *W1, const float *B1, size_t n, float *raw, void *stream)
{
    hipLaunchKernelGGL(slx_delta_p_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W0, B0, W1, B1, n, raw);
    return (int)hipGetLastError();
}

int slx_launch_track_update(const SlxKParams &kp, const float *raw, float *deltaP, double *U, double *z, double *x, double *y, double *deltaZ,
                            void *stream)
{
    const dim3 grid((unsigned)((kp.width + 63) / 64), (unsigned)((kp.height + 3) / 4));
    hipLaunchKernelGGL(slx_track_update_kernel, grid, dim3(256), 0, (hipStream_t)stream, raw, deltaP, U, z, x, y, deltaZ, kp);
    return (int)hipGetLastError();
}
