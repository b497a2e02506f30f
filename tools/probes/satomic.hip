// Probe: does gfx950 execute the scalar atomic s_atomic_add (returns the old value to an SGPR, counted by lgkmcnt)?
// Every wave takes 1000 tickets from one counter; the tickets of all waves must be a permutation of 0..N-1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void take(unsigned *ctr, unsigned *out, int per_wave)
{
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int i = 0; i < per_wave; i++) {
        unsigned one = 1u, old;
        asm volatile("s_mov_b32 %0, 1\n\ts_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(old) : "s"(ctr) : "memory");
        (void)one;
        if ((threadIdx.x & 63) == 0) out[wave * per_wave + i] = old;
    }
}
int main()
{
    const int waves = 4096, per = 1000;
    unsigned *ctr, *out;
    hipMalloc(&ctr, 4); hipMemset(ctr, 0, 4);
    hipMalloc(&out, sizeof(unsigned) * waves * per);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(take, dim3(waves / 4), dim3(256), 0, 0, ctr, out, per);
    hipEventRecord(b);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    if (e != hipSuccess) { printf("FAILED: %s\n", hipGetErrorString(e)); return 1; }
    std::vector<unsigned> h(waves * per);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = true;
    for (size_t i = 0; i < h.size(); i++) if (h[i] != i) { ok = false; printf("ticket %zu is %u\n", i, h[i]); break; }
    printf("s_atomic_add: %s, %.1f tickets per us on ONE counter (%d tickets in %.3f ms)\n", ok ? "permutation OK" : "WRONG", waves * per / (ms * 1e3), waves * per, ms);
    return ok ? 0 : 1;
}
