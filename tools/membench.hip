// Micro-benchmark of the memory access patterns the decode kernel could use (GPU box only):
// 12 u8 input planes read once, one f64 plane written once, 32 frame-sets of 1920x1200, no arithmetic.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o /tmp/membench && /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NP = 12;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct Args { const uint8_t *in; double *out; size_t plane, set_stride; size_t hw; int nt; };

// A: dword loads (4 px per lane), two 16-B stores per lane at 32-B lane stride (today's kernel)
template <int NT>
__global__ __launch_bounds__(256) void kA(Args a)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;      // quad index within the set
    const size_t set = blockIdx.y;
    uint32_t acc = 0;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.in + set * a.set_stride + p * a.plane) + q;
        acc += NT ? __builtin_nontemporal_load(src) : *src;
    }
    f64x2 v0 = {(double)acc, (double)(acc >> 1)}, v1 = {(double)(acc >> 2), (double)(acc >> 3)};
    f64x2 *dst = reinterpret_cast<f64x2 *>(a.out + set * a.hw + q * 4);
    if (NT) { __builtin_nontemporal_store(v0, dst); __builtin_nontemporal_store(v1, dst + 1); }
    else { dst[0] = v0; dst[1] = v1; }
}

// B: dword loads, stores lane-contiguous: a wave's 256 px = 2 KB written as two 1-KB instructions
template <int NT>
__global__ __launch_bounds__(256) void kB(Args a)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t set = blockIdx.y;
    uint32_t acc = 0;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.in + set * a.set_stride + p * a.plane) + q;
        acc += NT ? __builtin_nontemporal_load(src) : *src;
    }
    const unsigned lane = threadIdx.x & 63;
    const size_t wave_px = (q - lane) * 4;                         // first pixel of this wave
    f64x2 v0 = {(double)acc, (double)(acc >> 1)}, v1 = {(double)(acc >> 2), (double)(acc >> 3)};
    f64x2 *dst = reinterpret_cast<f64x2 *>(a.out + set * a.hw + wave_px);
    if (NT) { __builtin_nontemporal_store(v0, dst + lane); __builtin_nontemporal_store(v1, dst + 64 + lane); }
    else { dst[lane] = v0; dst[64 + lane] = v1; }
}

// C: 16-byte loads (16 px per lane), eight lane-contiguous 16-B stores (1 KB per instruction)
template <int NT>
__global__ __launch_bounds__(256) void kC(Args a)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;      // 16-px group index
    const size_t set = blockIdx.y;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(a.in + set * a.set_stride + p * a.plane) + q;
        acc += NT ? __builtin_nontemporal_load(src) : *src;
    }
    const unsigned lane = threadIdx.x & 63;
    const size_t wave_px = (q - lane) * 16;
    f64x2 *dst = reinterpret_cast<f64x2 *>(a.out + set * a.hw + wave_px);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        f64x2 v = {(double)(acc[k & 3] >> k), (double)(acc[(k + 1) & 3] >> k)};
        if (NT) __builtin_nontemporal_store(v, dst + k * 64 + lane); else dst[k * 64 + lane] = v;
    }
}

// D: 16-byte loads, row-per-lane stores (each lane writes its own 128 contiguous bytes)
template <int NT>
__global__ __launch_bounds__(256) void kD(Args a)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t set = blockIdx.y;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(a.in + set * a.set_stride + p * a.plane) + q;
        acc += NT ? __builtin_nontemporal_load(src) : *src;
    }
    f64x2 *dst = reinterpret_cast<f64x2 *>(a.out + set * a.hw + q * 16);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        f64x2 v = {(double)(acc[k & 3] >> k), (double)(acc[(k + 1) & 3] >> k)};
        if (NT) __builtin_nontemporal_store(v, dst + k); else dst[k] = v;
    }
}

// E: 8-byte loads (8 px per lane), four lane-contiguous stores
template <int NT>
__global__ __launch_bounds__(256) void kE(Args a)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t set = blockIdx.y;
    u32x2 acc = {0, 0};
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const u32x2 *src = reinterpret_cast<const u32x2 *>(a.in + set * a.set_stride + p * a.plane) + q;
        acc += NT ? __builtin_nontemporal_load(src) : *src;
    }
    const unsigned lane = threadIdx.x & 63;
    const size_t wave_px = (q - lane) * 8;
    f64x2 *dst = reinterpret_cast<f64x2 *>(a.out + set * a.hw + wave_px);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        f64x2 v = {(double)(acc[k & 1] >> k), (double)(acc[(k + 1) & 1] >> k)};
        if (NT) __builtin_nontemporal_store(v, dst + k * 64 + lane); else dst[k * 64 + lane] = v;
    }
}

// F: plain 16-byte copy moving the same number of bytes in and out (reference ceiling)
__global__ __launch_bounds__(256) void kF(const u32x4 *in, u32x4 *out, size_t n_in, size_t n_out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32x4 v = {0, 0, 0, 0};
    if (i < n_in) v = in[i];
    // 12 bytes in per 8 bytes out: every lane reads one vector, two of three write one
    if (i < n_out && (i % 3) != 2) out[i - i / 3] = v;
}

int main()
{
    const int W = 1920, H = 1200, SETS = 32;
    const size_t hw = (size_t)W * H, plane = hw, set_stride = plane * NP;
    uint8_t *in; double *out;
    CHECK(hipMalloc(&in, set_stride * SETS));
    CHECK(hipMalloc(&out, hw * SETS * sizeof(double)));
    CHECK(hipMemset(in, 1, set_stride * SETS));
    Args a{in, out, plane, set_stride, hw, 0};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double bytes = (double)set_stride * SETS + (double)hw * SETS * 8;
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 5; i++) launch();
        hipEventRecord(e0, 0);
        const int reps = 50;
        for (int i = 0; i < reps; i++) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms * 1000 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
    };
    run("A  dword loads, 32B-stride store pairs", [&] { hipLaunchKernelGGL(kA<0>, dim3(hw / 4 / 256, SETS), dim3(256), 0, 0, a); });
    run("A' same, nontemporal", [&] { hipLaunchKernelGGL(kA<1>, dim3(hw / 4 / 256, SETS), dim3(256), 0, 0, a); });
    run("B  dword loads, lane-contiguous stores", [&] { hipLaunchKernelGGL(kB<0>, dim3(hw / 4 / 256, SETS), dim3(256), 0, 0, a); });
    run("B' same, nontemporal", [&] { hipLaunchKernelGGL(kB<1>, dim3(hw / 4 / 256, SETS), dim3(256), 0, 0, a); });
    run("E  8B loads, lane-contiguous stores", [&] { hipLaunchKernelGGL(kE<0>, dim3(hw / 8 / 256, SETS), dim3(256), 0, 0, a); });
    run("E' same, nontemporal", [&] { hipLaunchKernelGGL(kE<1>, dim3(hw / 8 / 256, SETS), dim3(256), 0, 0, a); });
    run("C  16B loads, lane-contiguous stores", [&] { hipLaunchKernelGGL(kC<0>, dim3(hw / 16 / 256, SETS), dim3(256), 0, 0, a); });
    run("C' same, nontemporal", [&] { hipLaunchKernelGGL(kC<1>, dim3(hw / 16 / 256, SETS), dim3(256), 0, 0, a); });
    run("D  16B loads, 128B-per-lane stores", [&] { hipLaunchKernelGGL(kD<0>, dim3(hw / 16 / 256, SETS), dim3(256), 0, 0, a); });
    run("D' same, nontemporal", [&] { hipLaunchKernelGGL(kD<1>, dim3(hw / 16 / 256, SETS), dim3(256), 0, 0, a); });
    const size_t n_in = set_stride * SETS / 16, n_out = hw * SETS * 8 / 16;
    run("F  16B copy, same bytes (reference)", [&] { hipLaunchKernelGGL(kF, dim3((n_in + 255) / 256), dim3(256), 0, 0, (const u32x4 *)in, (u32x4 *)out, n_in, n_out); });
    return 0;
}
