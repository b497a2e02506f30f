// slx_gather.hip -- the root-side row scatter of the STAGED depth-map gather (slx_comm.cpp, SLX_GATHER_STAGED).
//
// In the staged shape every peer sends the row tiles of a whole chunk of frame-sets as ONE contiguous message into a staging
// slot of the root; this kernel then moves every tile to its rows of its frame-set in the full [set][H][W] array.  One launch
// per chunk serves all peers: a segment is one peer's message (n_runs tiles of `run` doubles, src_stride apart in the slot,
// dst_stride = H*W apart in the full array).  Pure HBM copy, 16 bytes per lane where the tile size allows it, nontemporal on
// both sides (every byte is touched once); it runs on the comm's scatter stream while the NEXT chunk's messages are arriving.
#include <hip/hip_runtime.h>

#include "slx_kernels.h"

namespace {

template <typename V>
__global__ __launch_bounds__(256) void slx_row_scatter_kernel(const SlxScatterSegs segs, const double *__restrict__ stage, double *__restrict__ full)
{
    const SlxScatterSeg s = segs.seg[blockIdx.z];
    const unsigned tile = blockIdx.y;
    if (tile >= s.n_runs) return;                                    // the launch's grid is sized for the longest segment
    constexpr unsigned PER = sizeof(V) / sizeof(double);
    const unsigned long long n = s.run / PER;                        // vectors of one tile (host: run % PER == 0 for this instantiation)
    const V *src = reinterpret_cast<const V *>(stage + s.src + (unsigned long long)tile * s.src_stride);
    V *dst = reinterpret_cast<V *>(full + s.dst + (unsigned long long)tile * s.dst_stride);
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

}  // namespace

int slx_launch_row_scatter(const SlxScatterSegs &segs, const double *stage, double *full, void *stream)
{
    if (segs.n < 1 || segs.n > SLX_SCATTER_MAX_SEGS || !stage || !full) return (int)hipErrorInvalidValue;
    typedef double vec2 __attribute__((ext_vector_type(2)));
    unsigned long long longest = 0, tiles = 0;
    bool wide = (reinterpret_cast<uintptr_t>(stage) % 16 == 0) && (reinterpret_cast<uintptr_t>(full) % 16 == 0);
    for (int k = 0; k < segs.n; k++) {
        const SlxScatterSeg &s = segs.seg[k];
        longest = s.run > longest ? s.run : longest;
        tiles = s.n_runs > tiles ? s.n_runs : tiles;
        // 16-byte lanes need every tile of every segment to start and end on a 16-byte boundary
        if ((s.run | s.src | s.dst | s.src_stride | s.dst_stride) & 1ull) wide = false;
    }
    if (longest == 0 || tiles == 0) return 0;
    if (tiles > 65535ull) return (int)hipErrorInvalidValue;
    const unsigned long long vecs = wide ? longest / 2 : longest;
    // 4 vectors per lane in flight: a 150 x 1920 tile (2.3 MB) is 144 000 16-byte vectors = 141 workgroups per tile
    unsigned long long gx = (vecs + 256ull * 4ull - 1ull) / (256ull * 4ull);
    gx = gx < 1 ? 1 : gx > 4096ull ? 4096ull : gx;
    const dim3 grid((unsigned)gx, (unsigned)tiles, (unsigned)segs.n);
    if (wide) hipLaunchKernelGGL(slx_row_scatter_kernel<vec2>, grid, dim3(256), 0, (hipStream_t)stream, segs, stage, full);
    else hipLaunchKernelGGL(slx_row_scatter_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, segs, stage, full);
    return (int)hipGetLastError();
}
