// What a wave instruction costs in ENERGY on gfx950 (GPU box only; driven by tools/energy_probe.py, which reads the socket power
// while this program keeps one instruction class issuing on every SIMD).  The decode kernels of the Gray-free 4-step class hold
// the card at its 1400 W cap (tools/power_probe.py): their launch time is energy / 1400 W, so the price list that matters for
// them is joules per instruction, not cycles (tools/valubench.hip has those).
//   hipcc --offload-arch=gfx950 -O3 tools/energybench.hip -o tools/energybench
//   tools/energybench CLASS SECONDS [WAVES_PER_SIMD]   -> one JSON line: wave-instructions per second, chip-wide
// Each wave runs ITER x 8 independent instances of the instruction; operands are chosen so that values stay finite and differ
// from lane to lane and from instance to instance (consecutive operations through an ALU see unrelated operands; whether that
// toggles as many bits as image data does is not claimed -- the table is a price list to within tens of percent).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 16384;

#define BODY8(INS)                                                                                          \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                    \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(b), "v"(c)                                                                           \
                 : "vcc");

// a[k]: per lane and per instance in [1, 2); b: a contraction factor in (-0.95, -0.75) (fma chains settle on c / (1 - b), no
// overflow); c: per lane in [1, 2).  For the pure multiplies b is replaced by 1 + a few ulps, for the adds c by a small addend.
#define KERNEL(NAME, T_, INS, BEXPR, CEXPR)                                                           \
    __global__ __launch_bounds__(256) void NAME(T_ *out, int iters)                                  \
    {                                                                                                 \
        typedef T_ TYPE;                                                                              \
        const TYPE lane = (TYPE)(threadIdx.x & 63u), wave = (TYPE)(blockIdx.x & 7u);                  \
        TYPE a[8], b = (BEXPR), c = (CEXPR);                                                          \
        for (int k = 0; k < 8; k++) a[k] = (TYPE)1 + (lane * (TYPE)8 + (TYPE)k) * (TYPE)(1.0 / 520.0) + wave * (TYPE)(1.0 / 16384.0); \
        for (int i = 0; i < iters; i += 8) { BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) }   \
        TYPE s = 0;                                                                                   \
        for (int k = 0; k < 8; k++) s += a[k];                                                        \
        if (s == (TYPE)12345) out[0] = s;                                                             \
    }

#define B_CONTRACT ((TYPE) - 0.75 - lane * (TYPE)(0.2 / 64.0))
#define B_NEARONE ((TYPE)1 + lane * (TYPE)(1.0 / 1073741824.0))
#define C_UNIT ((TYPE)1 + lane * (TYPE)(1.0 / 64.0))
#define C_SMALL (((TYPE)1 + lane * (TYPE)(1.0 / 64.0)) * (TYPE)(1.0 / 65536.0))

#define I_FMA32(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_MUL32(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define I_ADD32(k) "v_add_f32 %" #k ", %" #k ", %9\n"
#define I_PKFMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_PKMUL(k) "v_pk_mul_f32 %" #k ", %" #k ", %8\n"
#define I_PKADD(k) "v_pk_add_f32 %" #k ", %" #k ", %9\n"
#define I_FMA64(k) "v_fma_f64 %" #k ", %" #k ", %8, %9\n"
#define I_MUL64(k) "v_mul_f64 %" #k ", %" #k ", %8\n"
#define I_ADD64(k) "v_add_f64 %" #k ", %" #k ", %9\n"
#define I_RCP32(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define I_RCP64(k) "v_rcp_f64 %" #k ", %" #k "\n"
#define I_FLOOR64(k) "v_floor_f64 %" #k ", %" #k "\n"
#define I_CMP64(k) "v_cmp_lt_f64 vcc, %" #k ", %9\n"
#define I_CMP32(k) "v_cmp_lt_f32 vcc, %" #k ", %9\n"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %9, vcc\n"
#define I_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_MIN32(k) "v_min_f32_e64 %" #k ", |%" #k "|, |%9|\n"
#define I_SUBSDWA(k) "v_sub_f32_sdwa %" #k ", %" #k ", %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
#define I_MOV32(k) "v_mov_b32 %" #k ", %9\n"
#define I_XOR(k) "v_xor_b32 %" #k ", %" #k ", %9\n"
#define I_NOP(k) "s_nop 3\n"

// the pk classes run on a 64-bit register pair: TYPE = double only names the register class, the halves are two floats
// (1 + small, built from the double's bit pattern would be meaningless) -- so they get their own kernel with float2 operands
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define KERNEL_PK(NAME, INS, BEXPR, CEXPR)                                                            \
    __global__ __launch_bounds__(256) void NAME(double *out, int iters)                              \
    {                                                                                                 \
        typedef float TYPE;                                                                           \
        const float lane = (float)(threadIdx.x & 63u), wave = (float)(blockIdx.x & 7u);               \
        f32x2 a[8];                                                                                   \
        const float b1 = (BEXPR), c1 = (CEXPR);                                                       \
        f32x2 b = {b1, b1 * 1.03125f}, c = {c1, c1 * 0.96875f};                                       \
        for (int k = 0; k < 8; k++) {                                                                 \
            const float v = 1.f + (lane * 8.f + (float)k) * (1.f / 520.f) + wave * (1.f / 16384.f);   \
            a[k] = f32x2{v, 3.f - v};                                                                 \
        }                                                                                             \
        for (int i = 0; i < iters; i += 8) { BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) }   \
        float s = 0;                                                                                  \
        for (int k = 0; k < 8; k++) s += a[k].x + a[k].y;                                             \
        if (s == 12345.f) out[0] = s;                                                                 \
    }

KERNEL(k_fma32, float, I_FMA32, B_CONTRACT, C_UNIT)
KERNEL(k_mul32, float, I_MUL32, B_NEARONE, C_UNIT)
KERNEL(k_add32, float, I_ADD32, B_CONTRACT, C_SMALL)
// the same multiply with a factor of 1.5: the accumulators overflow to +inf within a hundred iterations and stay there -- what
// tools/valubench's loop did, and why its table showed plain multiplies / adds at half the cycles they take on finite data
KERNEL(k_mul32_inf, float, I_MUL32, ((TYPE)1.5), C_UNIT)
KERNEL(k_add32_inf, float, I_ADD32, B_CONTRACT, ((TYPE)3.0e38f))
KERNEL_PK(k_pkfma, I_PKFMA, B_CONTRACT, C_UNIT)
KERNEL_PK(k_pkmul, I_PKMUL, B_NEARONE, C_UNIT)
KERNEL_PK(k_pkadd, I_PKADD, B_CONTRACT, C_SMALL)
KERNEL(k_fma64, double, I_FMA64, B_CONTRACT, C_UNIT)
KERNEL(k_mul64, double, I_MUL64, B_NEARONE, C_UNIT)
KERNEL(k_add64, double, I_ADD64, B_CONTRACT, C_SMALL)
KERNEL(k_rcp32, float, I_RCP32, B_CONTRACT, C_UNIT)
KERNEL(k_rcp64, double, I_RCP64, B_CONTRACT, C_UNIT)
KERNEL(k_floor64, double, I_FLOOR64, B_CONTRACT, C_UNIT)
KERNEL(k_cmp64, double, I_CMP64, B_CONTRACT, C_UNIT)
KERNEL(k_cmp32, float, I_CMP32, B_CONTRACT, C_UNIT)
KERNEL(k_cndmask, float, I_CNDMASK, B_CONTRACT, C_UNIT)
KERNEL(k_max3, float, I_MAX3, B_CONTRACT, C_UNIT)
KERNEL(k_min32, float, I_MIN32, B_CONTRACT, C_UNIT)
KERNEL(k_subsdwa, float, I_SUBSDWA, B_CONTRACT, C_UNIT)
KERNEL(k_mov32, float, I_MOV32, B_CONTRACT, C_UNIT)
KERNEL(k_xor, float, I_XOR, B_CONTRACT, C_UNIT)
KERNEL(k_nop, float, I_NOP, B_CONTRACT, C_UNIT)

__global__ __launch_bounds__(256) void k_cvt64_32(double *out, int iters)
{
    float a[8];
    double d[8];
    for (int k = 0; k < 8; k++) a[k] = 1.f + (float)((threadIdx.x & 63u) * 8u + k) * (1.f / 520.f);
    for (int i = 0; i < iters; i++) {
        asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\n"
                     "v_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15\n"
                     : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]), "=v"(d[4]), "=v"(d[5]), "=v"(d[6]), "=v"(d[7])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    }
    double s = 0;
    for (int k = 0; k < 8; k++) s += d[k];
    if (s == 12345.0) out[0] = s;
}

// conflict-free ds_read_b32 (what a row step's plane dwords cost on the way from the DMA ring to the registers)
__global__ __launch_bounds__(256) void k_ldsr32(double *out, int iters)
{
    __shared__ uint32_t lds[4096];
    for (unsigned i = threadIdx.x; i < 4096u; i += 256u) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t acc = 0;
    const uint32_t *src = lds + (threadIdx.x & 63u);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((uint32_t)(uintptr_t)src), "n"(k * 1024));
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            acc += v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (acc == 12345u) out[0] = acc;
}

struct Entry { const char *name; void (*fn)(void *, int); };
template <typename T, void (*K)(T *, int)>
static void launcher(void *out, int blocks) { hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, (T *)out, ITER); }
#define ENTRY(NAME, T, K) {NAME, launcher<T, K>}
static const Entry kTable[] = {
    ENTRY("s_nop", float, k_nop),           ENTRY("v_mov_b32", float, k_mov32),     ENTRY("v_xor_b32", float, k_xor),
    ENTRY("v_mul_f32", float, k_mul32),     ENTRY("v_add_f32", float, k_add32),     ENTRY("v_fma_f32", float, k_fma32),
    ENTRY("v_mul_f32_inf", float, k_mul32_inf), ENTRY("v_add_f32_inf", float, k_add32_inf),
    ENTRY("v_pk_mul_f32", double, k_pkmul), ENTRY("v_pk_add_f32", double, k_pkadd), ENTRY("v_pk_fma_f32", double, k_pkfma),
    ENTRY("v_sub_f32_sdwa", float, k_subsdwa), ENTRY("v_max3_f32", float, k_max3),  ENTRY("v_min_f32", float, k_min32),
    ENTRY("v_cmp_lt_f32", float, k_cmp32),  ENTRY("v_cndmask_b32", float, k_cndmask), ENTRY("v_rcp_f32", float, k_rcp32),
    ENTRY("v_mul_f64", double, k_mul64),    ENTRY("v_add_f64", double, k_add64),    ENTRY("v_fma_f64", double, k_fma64),
    ENTRY("v_floor_f64", double, k_floor64), ENTRY("v_cmp_lt_f64", double, k_cmp64), ENTRY("v_rcp_f64", double, k_rcp64),
    ENTRY("v_cvt_f64_f32", double, k_cvt64_32), ENTRY("ds_read_b32", double, k_ldsr32),
};

int main(int argc, char **argv)
{
    if (argc < 3) {
        printf("usage: energybench CLASS SECONDS [WAVES_PER_SIMD]; classes:");
        for (const Entry &e : kTable) printf(" %s", e.name);
        printf("\n");
        return 2;
    }
    const Entry *entry = nullptr;
    for (const Entry &e : kTable)
        if (!strcmp(e.name, argv[1])) entry = &e;
    if (!entry) { printf("{\"error\": \"unknown class %s\"}\n", argv[1]); return 2; }
    const double seconds = atof(argv[2]);
    const int wps = argc > 3 ? atoi(argv[3]) : 4;
    if (!(seconds > 0 && seconds <= 30) || wps < 1 || wps > 8) { printf("{\"error\": \"bad arguments\"}\n"); return 2; }
    void *out;
    CHECK(hipMalloc(&out, 64));
    const int blocks = 256 * wps;                      // 256 CUs x 4 SIMDs: one 256-thread block is one wave per SIMD of a CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // launches back to back until the time is up; the second half is what gets timed (the clock has settled by then)
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    long launches_timed = 0;
    bool timing = false;
    while (elapsed() < seconds) {
        if (!timing && elapsed() >= seconds * 0.5) {
            CHECK(hipEventRecord(e0));
            timing = true;
        }
        for (int i = 0; i < 8; i++) entry->fn(out, blocks);
        if (timing) launches_timed += 8;
        CHECK(hipStreamSynchronize(0));
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    if (timing) CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)launches_timed * blocks * 4.0 * ITER * 8.0;      // wave-instructions, chip-wide
    printf("{\"class\": \"%s\", \"waves_per_simd\": %d, \"seconds\": %.2f, \"timed_ms\": %.2f, \"launches_timed\": %ld, "
           "\"wave_instr_per_s\": %.6g, \"ns_per_wave_instr_per_simd\": %.4f}\n",
           entry->name, wps, elapsed(), ms, launches_timed, ms > 0 ? instr / (ms * 1e-3) : 0.0,
           ms > 0 ? ms * 1e6 / ((double)launches_timed * wps * ITER * 8.0) : 0.0);
    CHECK(hipFree(out));
    return 0;
}
