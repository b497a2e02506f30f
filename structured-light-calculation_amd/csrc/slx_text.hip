// slx_text.hip -- the point-cloud TEXT of CCalculation::Result, formatted on the device.
//
// CCalculation::Result (R/CCalculation.cpp:323-357) writes "x y z\n" for every point, each number as `ostream << double` prints it
// (%g with 6 significant digits).  Formatting IS the cost of that function (3.6 s per 1920 x 1200 frame in the reference's loop;
// 8-10 ms with the host-side writer of sensor.cpp on 16 threads, more than the rest of a dynamic frame together).  Here the packed
// (x, y, z) triples the cloud kernels leave in device memory become that text in device memory -- exactly the bytes of the host writer
// (fmt_g6_fast, sensor.cpp: six significant digits = round-half-even of the EXACT binary value scaled by a power of ten, in integer
// arithmetic) -- and only the text crosses PCIe.
//
// Two launches over the points, 1 024 per workgroup (a wave takes 256 consecutive points, a lane every 64th of them):
//   slx_text_len_kernel   the length of every workgroup's text (digits and exponent of each number, no characters)
//   slx_text_emit_kernel  a workgroup sums the lengths of the workgroups before it (a few thousand words, read by all lanes at once),
//                         scans its lanes' lengths, writes the characters into LDS at their place and copies the packed text out as
//                         aligned dwords (the partial dwords at its two ends byte by byte: they are shared with the neighbours)
// A number outside the fast range (|v| < 1e-5 other than zero, |v| >= 1e15, NaN, infinity) raises a flag instead: the caller formats
// that cloud on the host (std::to_chars / the C library's spelling of nan and inf).
#include <hip/hip_runtime.h>

#include "slx_kernels.h"

namespace {

constexpr unsigned kThreads = 256, kPerLane = SLX_TEXT_POINTS_PER_WG / kThreads;
static_assert(kPerLane * kThreads == SLX_TEXT_POINTS_PER_WG && kPerLane == 4, "4 points per lane");

__device__ const unsigned long long kPow10[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull,
                                                  10000000000ull, 100000000000ull, 1000000000000ull, 10000000000000ull, 100000000000000ull,
                                                  1000000000000000ull, 10000000000000000ull, 100000000000000000ull, 1000000000000000000ull,
                                                  10000000000000000000ull};

// One number, digested, in a word: the six significant digits as BCD (digit i, the most significant first, in bits 20 - 4 i ..; all zero
// for a zero), bits 24-28 the decimal exponent X of the first digit + 16, bit 29 the sign.  (The kernels keep ONE copy of the digit
// arithmetic in a loop over a lane's twelve numbers: unrolled it was 24 000 instructions, more than the instruction cache holds.)
constexpr unsigned kG6Neg = 1u << 29;
__device__ __forceinline__ unsigned g6_bcd(unsigned g) { return g & 0xffffffu; }
__device__ __forceinline__ int g6_X(unsigned g) { return (int)((g >> 24) & 31u) - 16; }
__device__ __forceinline__ unsigned g6_nd(unsigned g)               // digits left once %g has dropped the trailing zeros
{
    const unsigned z = (unsigned)__builtin_ctz(g6_bcd(g) | 0x100000u) >> 2;   // trailing zero digits, 5 at most
    return 6u - z;
}

// sensor.cpp fmt_g6_fast: the same six digits.  The integer arithmetic of that function (the EXACT value a * 10^p, rounded half to even) is
// what decides a tie; in front of it stands a shortcut in double arithmetic for everything that is not near one: t = fl(a * 10^p)
// -- one rounding, 10^p is exact; p >= 0, i.e. |v| < 10^6 -- lies within 2^-33 of the exact value e (e < 10^6 < 2^20), so whenever t is farther than
// 2^-30 from every half-integer, rint(e) = rint(t); and whenever t is farther than 10^-6 from 10^5 and 10^6, floor(t) and floor(e) fall on
// the same side of the range test.  Anything nearer (a tie or near-tie of the sixth digit: decimal-looking inputs such as 123.4565) takes
// the integer path.  Measured cloud coordinates never do; the kernels were 87 + 117 us per 2.27 M points without the shortcut.
__device__ __forceinline__ unsigned g6_digits(double v, bool &ok)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const unsigned neg = (bits >> 63) ? kG6Neg : 0u;
    const unsigned long long mag = bits & 0x7fffffffffffffffull;
    if (mag == 0) return neg | (16u << 24);                          // zero: no digits
    const int k = (int)(mag >> 52) - 1023;
    const double a = __builtin_fabs(v);
    if (k < -17 || k > 49 || !(a >= 1e-5 && a < 1e15)) {             // 1e-5 > 2^-17, 1e15 < 2^50
        ok = false;
        return neg | (16u << 24);
    }
    const unsigned long long m = (mag & 0x000fffffffffffffull) | 0x0010000000000000ull;   // a = m * 2^(k - 52), exactly
    const int s = 52 - k;                                            // > 0 in this range: a = m / 2^s
    int X = k >= 0 ? (k * 1233) >> 12 : -(((-k) * 1233 + 4095) >> 12);   // within one of floor(log10 a)
    unsigned long long q = 0;
#pragma unroll 1
    for (int tries = 0; tries < 4; tries++) {                        // at most two corrections of the estimate
        const int p = 5 - X;                                         // digits = a * 10^p, wanted in [10^5, 10^6)
        if (p >= 0 && p <= 10) {                                     // (|v| >= 10^6, a division: rare in a cloud, left to the integers)
            // 10^p from its bits: exact products below 2^53, no table in memory on the way
            const double P = ((p & 1) ? 10.0 : 1.0) * ((p & 2) ? 100.0 : 1.0) * (((p & 4) ? 1e4 : 1.0) * ((p & 8) ? 1e8 : 1.0));
            const double t = a * P;
            const double f = __builtin_floor(t), fr = t - f;
            if (__builtin_fabs(fr - 0.5) > 0x1p-30 && __builtin_fabs(t - 1e5) > 1e-6 && __builtin_fabs(t - 1e6) > 1e-6) {
                if (f < 1e5) { X--; continue; }
                if (f >= 1e6) { X++; continue; }
                unsigned q32 = (unsigned)f + (fr > 0.5 ? 1u : 0u);   // f < 2^20
                if (q32 == 1000000u) { q32 = 100000u; X++; }
                q = q32;
                break;
            }
        }
        bool up;
        if (p >= 0) {
            const unsigned __int128 T = (unsigned __int128)m * kPow10[p];   // p <= 10: T < 2^87
            q = (unsigned long long)(T >> s);
            if (q < 100000ull) { X--; continue; }
            if (q >= 1000000ull) { X++; continue; }
            const unsigned __int128 rem = T & ((((unsigned __int128)1) << s) - 1), half = ((unsigned __int128)1) << (s - 1);
            up = rem > half || (rem == half && (q & 1ull));
        } else {
            const unsigned long long D = kPow10[-p] << s;            // -p <= 9, s <= 33: < 2^63
            q = m / D;
            if (q < 100000ull) { X--; continue; }
            if (q >= 1000000ull) { X++; continue; }
            const unsigned long long rem = m - q * D;
            up = 2 * rem > D || (2 * rem == D && (q & 1ull));
        }
        if (up && ++q == 1000000ull) { q = 100000ull; X++; }
        break;
    }
    unsigned t = (unsigned)q, bcd = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {                                    // least significant digit first
        bcd |= (t % 10u) << (4 * i);
        t /= 10u;
    }
    return neg | ((unsigned)(X + 16) << 24) | bcd;
}

// MSVC: the dialect of the reference as built (MSVC 2013 runtime, text-mode stream; include/slx.h, enum slx_text_dialect): three exponent
// digits ("5e-005") and CR LF -- one more character per number in exponent notation, one more per line
template <bool MSVC>
__device__ __forceinline__ unsigned g6_length(unsigned g)
{
    const unsigned n = (g & kG6Neg) ? 1u : 0u;
    if (g6_bcd(g) == 0u) return n + 1u;
    const int X = g6_X(g);
    const unsigned nd = g6_nd(g);
    if (X < -4 || X >= 6) return n + (nd > 1 ? nd + 5u : 5u) + (MSVC ? 1u : 0u);   // d[.ddd]e+XX, e+0XX
    if (X >= 0) return n + (unsigned)(X + 1) + (nd > (unsigned)(X + 1) ? 1u + nd - (unsigned)(X + 1) : 0u);
    return n + 2u + (unsigned)(-X - 1) + nd;                         // 0.000ddd
}

// the characters, at out[0 ..): returns the end.  d(i) = digit i of q, the most significant first.
template <bool MSVC>
__device__ __forceinline__ unsigned g6_put(unsigned g, unsigned char *out, unsigned o)
{
    if (g & kG6Neg) out[o++] = '-';
    const unsigned bcd = g6_bcd(g), nd = g6_nd(g);
    if (bcd == 0u) {
        out[o++] = '0';
        return o;
    }
    auto d = [&](unsigned i) { return (unsigned char)('0' + ((bcd >> (20u - 4u * i)) & 15u)); };
    const int X = g6_X(g);
    if (X < -4 || X >= 6) {
        out[o++] = d(0);
        if (nd > 1) {
            out[o++] = '.';
            for (unsigned i = 1; i < nd; i++) out[o++] = d(i);
        }
        out[o++] = 'e';
        int e = X;
        if (e < 0) {
            out[o++] = '-';
            e = -e;
        } else {
            out[o++] = '+';
        }
        if (MSVC) out[o++] = '0';
        out[o++] = (unsigned char)('0' + e / 10);
        out[o++] = (unsigned char)('0' + e % 10);
    } else if (X >= 0) {
        for (unsigned i = 0; i <= (unsigned)X; i++) out[o++] = i < nd ? d(i) : (unsigned char)'0';
        if (nd > (unsigned)(X + 1)) {
            out[o++] = '.';
            for (unsigned i = (unsigned)X + 1u; i < nd; i++) out[o++] = d(i);
        }
    } else {
        out[o++] = '0';
        out[o++] = '.';
        for (int i = 0; i < -X - 1; i++) out[o++] = '0';
        for (unsigned i = 0; i < nd; i++) out[o++] = d(i);
    }
    return o;
}

// sum over the workgroup of one value per lane; every lane gets it
__device__ __forceinline__ unsigned long long wg_sum(unsigned long long v, unsigned long long *scratch)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63u) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long t = 0;
#pragma unroll
    for (unsigned k = 0; k < kThreads / 64u; k++) t += scratch[k];
    __syncthreads();
    return t;
}

// flag: a word (pinned host memory) that receives `tag` when a number is outside the formatter's range -- tags differ from call to call,
// so nothing has to be cleared
template <bool MSVC>
__global__ __launch_bounds__(kThreads) void slx_text_len_kernel(const double *__restrict__ xyz, unsigned long long n_points, unsigned *__restrict__ sums,
                                                               unsigned *__restrict__ flag, unsigned tag, const unsigned *__restrict__ n_dev)
{
    // n_dev: the number of points is still on its way when this launch is queued (the cloud kernel before it on the stream leaves it
    // there): the grid then covers every pixel and the workgroups beyond the cloud have nothing to do
    if (n_dev) n_points = *n_dev;
    if ((unsigned long long)blockIdx.x * SLX_TEXT_POINTS_PER_WG >= n_points) return;
    __shared__ unsigned long long scratch[kThreads / 64u];
    // a wave takes 256 consecutive points, a lane every 64th of them: the lanes of a load stand 24 bytes apart
    const unsigned long long base = (unsigned long long)blockIdx.x * SLX_TEXT_POINTS_PER_WG + (threadIdx.x >> 6) * (64u * kPerLane) + (threadIdx.x & 63u);
    unsigned len = 0;
    bool bad = false;
    double v[kPerLane * 3];
#pragma unroll
    for (unsigned i = 0; i < kPerLane; i++)                          // all twelve loads first
#pragma unroll
        for (unsigned c = 0; c < 3; c++) v[3 * i + c] = base + 64u * i < n_points ? xyz[3ull * (base + 64u * i) + c] : 1.0;
    const unsigned mine_n = base >= n_points ? 0u : (unsigned)((n_points - base + 63u) / 64u < kPerLane ? (n_points - base + 63u) / 64u : kPerLane) * 3u;
    bool ok = true;
#pragma unroll 1
    for (unsigned j = 0; j < kPerLane * 3u; j++) {                   // ONE copy of the digit arithmetic
        double x = v[0];
#pragma unroll
        for (unsigned k = 1; k < kPerLane * 3u; k++) x = j == k ? v[k] : x;
        const unsigned g = g6_digits(x, ok);
        len += j < mine_n ? g6_length<MSVC>(g) + 1u + (MSVC && j % 3u == 2u ? 1u : 0u) : 0u;   // + the blank or the newline (CR LF) behind it
    }
    bad = !ok;
    if (bad) *flag = tag;
    const unsigned long long total = wg_sum(len, scratch);
    if (threadIdx.x == 0) sums[blockIdx.x] = (unsigned)total;
}

template <bool MSVC>
__global__ __launch_bounds__(kThreads) void slx_text_emit_kernel(const double *__restrict__ xyz, unsigned long long n_points, const unsigned *__restrict__ sums,
                                                                unsigned char *__restrict__ text, unsigned long long *__restrict__ total_dev,
                                                                unsigned long long *__restrict__ total_host, unsigned wg_base, unsigned wgs_total,
                                                                const unsigned long long *__restrict__ bases)
{
    // wg_base / wgs_total: this launch emits the workgroups [wg_base, wg_base + gridDim.x) of the text's wgs_total -- the text leaves
    // in pieces, each copied to the host while the next one is formatted (slx_get_point_cloud_text)
    const unsigned bid = blockIdx.x + wg_base;
    __shared__ __attribute__((aligned(16))) unsigned char buf[SLX_TEXT_POINTS_PER_WG * (MSVC ? SLX_TEXT_LINE_MAX_MSVC : SLX_TEXT_LINE_MAX) + 128 + 16];
    __shared__ unsigned long long scratch[kThreads / 64u];
    __shared__ unsigned wave_len[kThreads / 64u];
    __shared__ unsigned digested[kPerLane * 3][kThreads];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // the numbers first: their loads are what everything waits for.  A wave takes 256 consecutive points, a lane every 64th of them
    // (the lanes of a load stand 24 bytes apart); part i of a wave is its points 64 i .. 64 i + 63.
    const unsigned long long base = (unsigned long long)bid * SLX_TEXT_POINTS_PER_WG + wave * (64u * kPerLane) + lane;
    double v[kPerLane * 3];
#pragma unroll
    for (unsigned i = 0; i < kPerLane; i++)
#pragma unroll
        for (unsigned c = 0; c < 3; c++) v[3 * i + c] = base + 64u * i < n_points ? xyz[3ull * (base + 64u * i) + c] : 1.0;
    unsigned plen[kPerLane];                                         // length of this lane's line of part i
#pragma unroll
    for (unsigned i = 0; i < kPerLane; i++) plen[i] = 0;
    {
        bool ok = true;
#pragma unroll 1
        for (unsigned j = 0; j < kPerLane * 3u; j++) {               // ONE copy of the digit arithmetic; the digested numbers wait in LDS
            double x = v[0];
#pragma unroll
            for (unsigned k = 1; k < kPerLane * 3u; k++) x = j == k ? v[k] : x;
            const unsigned g = g6_digits(x, ok);
            digested[j][tid] = g;
            const unsigned l = base + 64u * (j / 3u) < n_points ? g6_length<MSVC>(g) + 1u + (MSVC && j % 3u == 2u ? 1u : 0u) : 0u;
#pragma unroll
            for (unsigned i = 0; i < kPerLane; i++) plen[i] += j / 3u == i ? l : 0u;
        }
    }
    // where this workgroup's text starts: the lengths of all workgroups before it -- every workgroup adds them up itself (a few thousand
    // words for a frame's cloud); for clouds of more than SLX_TEXT_BASES_FROM workgroups a small launch in between has left the sum in front
    // of every run of SLX_TEXT_BASE_RUN workgroups in `bases`, and only the rest of the run is added here (the work stays linear in the points)
    unsigned long long before = (bases && tid == 0) ? bases[bid / SLX_TEXT_BASE_RUN] : 0ull;
    {
        constexpr unsigned UNROLL = 8;                               // all of a lane's loads in flight (2 220 workgroups for a 1920 x 1200 cloud: 9 per lane)
        for (unsigned i = (bases ? bid / SLX_TEXT_BASE_RUN * SLX_TEXT_BASE_RUN : 0u) + tid; i < bid; i += kThreads * UNROLL) {
            unsigned t[UNROLL];
#pragma unroll
            for (unsigned k = 0; k < UNROLL; k++) t[k] = sums[i + kThreads * k < bid ? i + kThreads * k : i];
#pragma unroll
            for (unsigned k = 0; k < UNROLL; k++) before += i + kThreads * k < bid ? t[k] : 0u;
        }
    }
    before = wg_sum(before, scratch);
    // where this lane's lines start within the workgroup's text: part after part of the wave, lane after lane within a part
    unsigned pstart[kPerLane], wave_total = 0;
#pragma unroll
    for (unsigned i = 0; i < kPerLane; i++) {
        unsigned incl = plen[i];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(incl, d);
            if (lane >= (unsigned)d) incl += t;
        }
        pstart[i] = wave_total + incl - plen[i];
        wave_total += __shfl(incl, 63);
    }
    if (lane == 0) wave_len[wave] = wave_total;
    __syncthreads();
    unsigned start = 0, mine = 0;
#pragma unroll
    for (unsigned k = 0; k < kThreads / 64u; k++) {
        start += k < wave ? wave_len[k] : 0u;
        mine += wave_len[k];
    }
    // characters into LDS: the buffer starts `mis` bytes in, so that its dwords are the aligned dwords of the text and a wave's 64 dwords
    // of the copy below one aligned 256-byte piece of it (with mis = before & 3 the stores straddled lines: 64.1 MB written for 56.1 MB)
    const unsigned mis = (unsigned)(before & 127ull);
#pragma unroll 1
    for (unsigned j = 0, o = 0; j < kPerLane * 3u; j++) {
        const unsigned i = j / 3u, c = j - 3u * i;
        if (base + 64u * i >= n_points) break;
        if (c == 0) {
            unsigned ps = pstart[0];
#pragma unroll
            for (unsigned k = 1; k < kPerLane; k++) ps = i == k ? pstart[k] : ps;
            o = mis + start + ps;
        }
        o = g6_put<MSVC>(digested[j][tid], buf, o);
        if (MSVC && c == 2) buf[o++] = '\r';
        buf[o++] = c == 2 ? '\n' : ' ';
    }
    __syncthreads();
    // out: whole dwords where all four bytes are this workgroup's, single bytes at the two ends
    unsigned char *dst = text + (before - mis);
    const unsigned end = mis + mine;
    for (unsigned w = tid; 4u * w < end; w += kThreads) {
        const unsigned lo = 4u * w;
        if (lo >= mis && lo + 4u <= end) {
            __builtin_nontemporal_store(*reinterpret_cast<const unsigned *>(buf + lo), reinterpret_cast<unsigned *>(dst + lo));
        } else {
            for (unsigned b = lo; b < lo + 4u; b++)
                if (b >= mis && b < end) dst[b] = buf[b];
        }
    }
    if (bid + 1u == wgs_total && tid == 0) {                  // the last workgroup knows the length of the text
        *total_dev = before + mine;
        if (total_host) *total_host = before + mine;
    }
}

// bases[r] = the text bytes in front of workgroup r * SLX_TEXT_BASE_RUN, for the clouds of very many workgroups (see slx_text_emit_kernel).
// One workgroup walks the runs in order; a run is a block-wide sum.
__global__ __launch_bounds__(1024) void slx_text_bases_kernel(const unsigned *__restrict__ sums, unsigned long long n_points, const unsigned *__restrict__ n_dev,
                                                             unsigned long long *__restrict__ bases)
{
    __shared__ unsigned long long part[16];
    if (n_dev) n_points = *n_dev;
    const unsigned wgs = (unsigned)((n_points + SLX_TEXT_POINTS_PER_WG - 1ull) / SLX_TEXT_POINTS_PER_WG);
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long carry = 0;
    for (unsigned r = 0; r * SLX_TEXT_BASE_RUN < wgs; r++) {
        if (threadIdx.x == 0) bases[r] = carry;
        unsigned long long sum = 0;
        for (unsigned i = r * SLX_TEXT_BASE_RUN + threadIdx.x; i < (r + 1u) * SLX_TEXT_BASE_RUN && i < wgs; i += 1024u) sum += sums[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        if (lane == 0) part[wave] = sum;
        __syncthreads();
        unsigned long long t = 0;
#pragma unroll
        for (unsigned k = 0; k < 16u; k++) t += part[k];
        carry += t;
        __syncthreads();
    }
}

// Where the pieces of the text begin: the workgroups are cut into K runs of ceil(wgs / K) (the host cuts them the same way once it
// knows the number of points) and offsets[k] receives the text bytes in front of run k, offsets[K] the length of the text -- pinned
// host words, read after ONE wait for cloud + lengths + this.  One wave per run.
__global__ __launch_bounds__(64 * SLX_TEXT_MAX_PIECES) void slx_text_bounds_kernel(const unsigned *__restrict__ sums, const unsigned *__restrict__ n_dev, unsigned K,
                                                                                  unsigned long long *__restrict__ offsets_host, unsigned long long *__restrict__ total_dev)
{
    __shared__ unsigned long long part[SLX_TEXT_MAX_PIECES];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const unsigned long long n = *n_dev;
    const unsigned wgs = (unsigned)((n + SLX_TEXT_POINTS_PER_WG - 1ull) / SLX_TEXT_POINTS_PER_WG), per = (wgs + K - 1u) / K;
    unsigned long long sum = 0;
    if (wave < K) {
        const unsigned a = wave * per < wgs ? wave * per : wgs, b = a + per < wgs ? a + per : wgs;
        for (unsigned i = a + lane; i < b; i += 64u) sum += sums[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        if (lane == 0) part[wave] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long acc = 0;
        offsets_host[0] = 0;
        for (unsigned k = 0; k < K; k++) {
            acc += part[k];
            offsets_host[k + 1u] = acc;
        }
        *total_dev = acc;
    }
}

}  // namespace

// The pieces of slx_get_point_cloud_text's pipeline.  (1) lengths + piece offsets of a cloud whose size is still a device word;
// (2) the characters of the workgroups [wg_base, wg_base + n_wgs).
int slx_launch_text_lengths(const double *xyz, const unsigned *n_dev, unsigned long long max_points, unsigned *sums, unsigned *flag, unsigned tag, unsigned pieces,
                            unsigned long long *offsets_host, unsigned long long *total_dev, int msvc, unsigned long long *bases, void *stream)
{
    if (!xyz || !n_dev || max_points == 0 || !sums || !flag || !offsets_host || !total_dev || pieces < 1 || pieces > SLX_TEXT_MAX_PIECES) return (int)hipErrorInvalidValue;
    const unsigned long long wgs = (max_points + SLX_TEXT_POINTS_PER_WG - 1ull) / SLX_TEXT_POINTS_PER_WG;
    if (wgs >= (1ull << 31)) return (int)hipErrorInvalidValue;
    if (msvc) hipLaunchKernelGGL(slx_text_len_kernel<true>, dim3((unsigned)wgs), dim3(kThreads), 0, (hipStream_t)stream, xyz, max_points, sums, flag, tag, n_dev);
    else hipLaunchKernelGGL(slx_text_len_kernel<false>, dim3((unsigned)wgs), dim3(kThreads), 0, (hipStream_t)stream, xyz, max_points, sums, flag, tag, n_dev);
    hipLaunchKernelGGL(slx_text_bounds_kernel, dim3(1), dim3(64u * pieces), 0, (hipStream_t)stream, sums, n_dev, pieces, offsets_host, total_dev);
    if (bases) hipLaunchKernelGGL(slx_text_bases_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sums, max_points, n_dev, bases);
    return (int)hipGetLastError();
}

int slx_launch_text_piece(const double *xyz, unsigned long long n_points, const unsigned *sums, unsigned char *text, unsigned long long *total_dev, unsigned wg_base,
                          unsigned n_wgs, int msvc, const unsigned long long *bases, void *stream)
{
    if (!xyz || n_points == 0 || !sums || !text || !total_dev || n_wgs == 0 || (reinterpret_cast<uintptr_t>(text) & 3u)) return (int)hipErrorInvalidValue;
    const unsigned long long wgs = (n_points + SLX_TEXT_POINTS_PER_WG - 1ull) / SLX_TEXT_POINTS_PER_WG;
    if (wgs >= (1ull << 31) || (unsigned long long)wg_base + n_wgs > wgs) return (int)hipErrorInvalidValue;
    if (msvc) hipLaunchKernelGGL(slx_text_emit_kernel<true>, dim3(n_wgs), dim3(kThreads), 0, (hipStream_t)stream, xyz, n_points, sums, text, total_dev, (unsigned long long *)nullptr, wg_base, (unsigned)wgs, bases);
    else hipLaunchKernelGGL(slx_text_emit_kernel<false>, dim3(n_wgs), dim3(kThreads), 0, (hipStream_t)stream, xyz, n_points, sums, text, total_dev, (unsigned long long *)nullptr, wg_base, (unsigned)wgs, bases);
    return (int)hipGetLastError();
}

int slx_launch_text(const double *xyz, unsigned long long n_points, unsigned *sums, unsigned *flag, unsigned tag, unsigned char *text,
                    unsigned long long *total_dev, unsigned long long *total_host, int msvc, unsigned long long *bases, void *stream)
{
    if (!xyz || n_points == 0 || !sums || !flag || !text || !total_dev || (reinterpret_cast<uintptr_t>(text) & 3u)) return (int)hipErrorInvalidValue;
    const unsigned long long wgs = (n_points + SLX_TEXT_POINTS_PER_WG - 1ull) / SLX_TEXT_POINTS_PER_WG;
    if (wgs >= (1ull << 31)) return (int)hipErrorInvalidValue;
    if (msvc) {
        hipLaunchKernelGGL(slx_text_len_kernel<true>, dim3((unsigned)wgs), dim3(kThreads), 0, (hipStream_t)stream, xyz, n_points, sums, flag, tag, (const unsigned *)nullptr);
        if (bases) hipLaunchKernelGGL(slx_text_bases_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sums, n_points, (const unsigned *)nullptr, bases);
        hipLaunchKernelGGL(slx_text_emit_kernel<true>, dim3((unsigned)wgs), dim3(kThreads), 0, (hipStream_t)stream, xyz, n_points, sums, text, total_dev, total_host, 0u, (unsigned)wgs,
                           (const unsigned long long *)bases);
    } else {
        hipLaunchKernelGGL(slx_text_len_kernel<false>, dim3((unsigned)wgs), dim3(kThreads), 0, (hipStream_t)stream, xyz, n_points, sums, flag, tag, (const unsigned *)nullptr);
        if (bases) hipLaunchKernelGGL(slx_text_bases_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sums, n_points, (const unsigned *)nullptr, bases);
        hipLaunchKernelGGL(slx_text_emit_kernel<false>, dim3((unsigned)wgs), dim3(kThreads), 0, (hipStream_t)stream, xyz, n_points, sums, text, total_dev, total_host, 0u, (unsigned)wgs,
                           (const unsigned long long *)bases);
    }
    return (int)hipGetLastError();
}
