// Micro-benchmark of VALU issue cost per instruction type on gfx950 (GPU box only).
// Every SIMD gets WAVES waves; each wave runs ITER x 8 independent instances of one instruction.
// Prints the time per wave-instruction per SIMD relative to v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/valubench.hip -o tools/valubench && tools/valubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 32768;

#define BODY8(INS)                                                                                          \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                    \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(b), "v"(c));

#define KERNEL(NAME, TYPE, INS)                                               \
    __global__ __launch_bounds__(256) void NAME(TYPE *out, TYPE seed)        \
    {                                                                         \
        TYPE a[8], b = seed, c = seed;                                        \
        for (int k = 0; k < 8; k++) a[k] = seed + (TYPE)(threadIdx.x + k);    \
        for (int i = 0; i < ITER; i++) { BODY8(INS) }                         \
        TYPE s = 0;                                                           \
        for (int k = 0; k < 8; k++) s += a[k];                                \
        if (s == (TYPE)12345) out[0] = s;                                     \
    }

#define I_FMA32(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_PKFMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_PKMUL(k) "v_pk_mul_f32 %" #k ", %" #k ", %8\n"
#define I_PKADD(k) "v_pk_add_f32 %" #k ", %" #k ", %8\n"
#define I_FMA64(k) "v_fma_f64 %" #k ", %" #k ", %8, %9\n"
#define I_ADD64(k) "v_add_f64 %" #k ", %" #k ", %8\n"
#define I_MUL64(k) "v_mul_f64 %" #k ", %" #k ", %8\n"
#define I_RCP32(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define I_RCP64(k) "v_rcp_f64 %" #k ", %" #k "\n"
#define I_FLOOR64(k) "v_floor_f64 %" #k ", %" #k "\n"
#define I_CVTI(k) "v_cvt_f32_i32 %" #k ", %" #k "\n"
#define I_CVTUB(k) "v_cvt_f32_ubyte1 %" #k ", %" #k "\n"
#define I_SDWA(k) "v_sub_u32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_MAX32(k) "v_max_f32 %" #k ", %" #k ", %8\n"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n"
#define I_MOV64(k) "v_mov_b64 %" #k ", %8\n"
#define I_MAX64(k) "v_max_f64 %" #k ", %" #k ", %8\n"
#define I_CMP64(k) "v_cmp_lt_f64 vcc, %" #k ", %8\n"
#define I_CMP32(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n"

#define I_ADDU(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define I_LSHL(k) "v_lshlrev_b32 %" #k ", 3, %" #k "\n"
#define I_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8\n"
#define I_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 8, 8\n"
#define I_ADD32(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define I_MUL32(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define I_MOV32(k) "v_mov_b32 %" #k ", %8\n"
#define I_CNDS(k) "v_cndmask_b32 %" #k ", %" #k ", %8, s[20:21]\n"
#define I_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_MAXI(k) "v_max_i32 %" #k ", %" #k ", %8\n"
#define I_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define I_MULCLAMP(k) "v_mul_f32 %" #k ", %" #k ", %8 clamp\n"
#define I_CVTU32(k) "v_cvt_u32_f32 %" #k ", %" #k "\n"
#define I_FRACT(k) "v_fract_f32 %" #k ", %" #k "\n"
#define I_SQRT(k) "v_sqrt_f32 %" #k ", %" #k "\n"
#define I_DPP(k) "v_mov_b32_dpp %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADD64S(k) "v_add_f64 %" #k ", %" #k ", s[20:21]\n"
#define I_LSHL64(k) "v_lshlrev_b64 %" #k ", 3, %" #k "\n"
#define I_LDEXP64(k) "v_ldexp_f64 %" #k ", %" #k ", 1\n"
#define I_FRACT64(k) "v_fract_f64 %" #k ", %" #k "\n"
#define I_CNDV3(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, vcc\n"
#define I_CNDC(k) "v_cndmask_b32_e32 %" #k ", 0, %" #k ", vcc\n"
#define I_CNDB(k) "v_cndmask_b32_e32 %" #k ", %8, %9, vcc\n"
#define I_SUBF_SDWA(k) "v_sub_f32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n"
#define I_SUBU(k) "v_sub_u32 %" #k ", %" #k ", %8\n"
#define I_FMAC32(k) "v_fmac_f32 %" #k ", %8, %9\n"
#define I_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_MINE64(k) "v_min_f32_e64 %" #k ", |%" #k "|, |%8|\n"
#define I_MULE64(k) "v_mul_f32_e64 %" #k ", |%" #k "|, %8\n"
#define I_OR(k) "v_or_b32 %" #k ", %" #k ", %8\n"
#define I_BFI(k) "v_bfi_b32 %" #k ", %" #k ", %8, %9\n"
#define I_LSHR(k) "v_lshrrev_b32 %" #k ", 8, %" #k "\n"
#define I_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n"
#define I_PKMULBC(k) "v_pk_mul_f32 %" #k ", %8, %9\n"
#define I_SUBSDWABC(k) "v_sub_f32_sdwa %" #k ", %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n"
#define I_MULBC(k) "v_mul_f32 %" #k ", %8, %9\n"
// b, c taken from the kernel argument bit for bit: denormal or normal operands on demand
#define KERNEL_BC(NAME, TYPE, INS)                                            \
    __global__ __launch_bounds__(256) void NAME(TYPE *out, TYPE bb, TYPE cc)  \
    {                                                                         \
        TYPE a[8], b = bb, c = cc;                                            \
        for (int k = 0; k < 8; k++) a[k] = (TYPE)(threadIdx.x + k);           \
        for (int i = 0; i < ITER; i++) { BODY8(INS) }                         \
        TYPE s = 0;                                                           \
        for (int k = 0; k < 8; k++) s += a[k];                                \
        if (s == (TYPE)12345) out[0] = s;                                     \
    }
KERNEL_BC(k_pkmul_bc, double, I_PKMULBC)
KERNEL_BC(k_subsdwa_bc, float, I_SUBSDWABC)
KERNEL_BC(k_mul_bc, float, I_MULBC)
KERNEL(k_cndv3, float, I_CNDV3)
KERNEL(k_cndc, float, I_CNDC)
KERNEL(k_cndb, float, I_CNDB)
KERNEL(k_subf_sdwa, float, I_SUBF_SDWA)
KERNEL(k_subu, float, I_SUBU)
KERNEL(k_fmac32, float, I_FMAC32)
KERNEL(k_max3, float, I_MAX3)
KERNEL(k_mine64, float, I_MINE64)
KERNEL(k_mule64, float, I_MULE64)
KERNEL(k_or, float, I_OR)
KERNEL(k_bfi, float, I_BFI)
KERNEL(k_lshr, float, I_LSHR)
KERNEL(k_andor, float, I_ANDOR)
KERNEL(k_fma32, float, I_FMA32)
KERNEL(k_addu, float, I_ADDU)
KERNEL(k_lshl, float, I_LSHL)
KERNEL(k_xor, float, I_XOR)
KERNEL(k_perm, float, I_PERM)
KERNEL(k_bfe, float, I_BFE)
KERNEL(k_add32, float, I_ADD32)
KERNEL(k_mul32, float, I_MUL32)
KERNEL(k_mov32, float, I_MOV32)
KERNEL(k_cnds, float, I_CNDS)
KERNEL(k_med3, float, I_MED3)
KERNEL(k_maxi, float, I_MAXI)
KERNEL(k_add3, float, I_ADD3)
KERNEL(k_mullo, float, I_MULLO)
KERNEL(k_mulclamp, float, I_MULCLAMP)
KERNEL(k_cvtu32, float, I_CVTU32)
KERNEL(k_fract, float, I_FRACT)
KERNEL(k_sqrt, float, I_SQRT)
KERNEL(k_dpp, float, I_DPP)
KERNEL(k_add64s, double, I_ADD64S)
KERNEL(k_lshl64, double, I_LSHL64)
KERNEL(k_ldexp64, double, I_LDEXP64)
KERNEL(k_fract64, double, I_FRACT64)
KERNEL(k_max32, float, I_MAX32)
KERNEL(k_and, float, I_AND)
KERNEL(k_rcp32, float, I_RCP32)
KERNEL(k_cvti, float, I_CVTI)
KERNEL(k_cvtub, float, I_CVTUB)
KERNEL(k_sdwa, float, I_SDWA)
KERNEL(k_cndmask, float, I_CNDMASK)
KERNEL(k_cmp32, float, I_CMP32)
KERNEL(k_pkfma, double, I_PKFMA)
KERNEL(k_pkmul, double, I_PKMUL)
KERNEL(k_pkadd, double, I_PKADD)
KERNEL(k_fma64, double, I_FMA64)
KERNEL(k_add64, double, I_ADD64)
KERNEL(k_mul64, double, I_MUL64)
KERNEL(k_rcp64, double, I_RCP64)
KERNEL(k_floor64, double, I_FLOOR64)
KERNEL(k_mov64, double, I_MOV64)
KERNEL(k_max64, double, I_MAX64)
KERNEL(k_cmp64, double, I_CMP64)

// conversions between widths: separate register classes
__global__ __launch_bounds__(256) void k_cvt64_32(double *out, float seed)
{
    float a[8];
    double d[8];
    for (int k = 0; k < 8; k++) a[k] = seed + (float)(threadIdx.x + k);
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\n"
                     "v_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15\n"
                     : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]), "=v"(d[4]), "=v"(d[5]), "=v"(d[6]), "=v"(d[7])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    }
    double s = 0;
    for (int k = 0; k < 8; k++) s += d[k];
    if (s == 12345.0) out[0] = s;
}

__global__ __launch_bounds__(256) void k_cvt32_64(double *out, double seed)
{
    float a[8];
    double d[8];
    for (int k = 0; k < 8; k++) d[k] = seed + (double)(threadIdx.x + k);
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\n"
                     "v_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15\n"
                     : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]), "=v"(a[4]), "=v"(a[5]), "=v"(a[6]), "=v"(a[7])
                     : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]));
    }
    float s = 0;
    for (int k = 0; k < 8; k++) s += a[k];
    if (s == 12345.f) out[0] = s;
}

__global__ __launch_bounds__(256) void k_cvti64(double *out, float seed)
{
    int a[8];
    double d[8];
    for (int k = 0; k < 8; k++) a[k] = (int)seed + (int)(threadIdx.x + k);
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_cvt_f64_i32 %0, %8\nv_cvt_f64_i32 %1, %9\nv_cvt_f64_i32 %2, %10\nv_cvt_f64_i32 %3, %11\n"
                     "v_cvt_f64_i32 %4, %12\nv_cvt_f64_i32 %5, %13\nv_cvt_f64_i32 %6, %14\nv_cvt_f64_i32 %7, %15\n"
                     : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]), "=v"(d[4]), "=v"(d[5]), "=v"(d[6]), "=v"(d[7])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    }
    double s = 0;
    for (int k = 0; k < 8; k++) s += d[k];
    if (s == 12345.0) out[0] = s;
}

// LDS: conflict-free ds_read_b32 / ds_read_b128
__global__ __launch_bounds__(256) void k_ldsr32(double *out, float seed)
{
    __shared__ uint32_t lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t acc = 0;
    const uint32_t *src = lds + threadIdx.x;
    for (int i = 0; i < ITER; i++) {
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

template <typename T, typename S>
static int run(const char *name, void (*fn)(T *, S), double &base, int waves_per_simd)
{
    T *out;
    CHECK(hipMalloc(&out, 64));
    const int blocks = 256 * waves_per_simd;                    // 256 CUs x 4 SIMDs; one 256-thread block = 1 wave per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, (S)1.5);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, (S)1.5);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double ns_per_instr = best * 1e6 / ((double)ITER * 8.0 * waves_per_simd);
    if (base == 0) base = ns_per_instr;
    printf("%-12s waves/SIMD %d  %8.3f ms  %6.3f ns per wave-instruction per SIMD  x%.2f\n", name, waves_per_simd, best, ns_per_instr, ns_per_instr / base);
    CHECK(hipFree(out));
    return 0;
}

template <typename T>
static int run_bc(const char *name, void (*fn)(T *, T, T), T b, T c, double &base, int waves_per_simd)
{
    T *out;
    CHECK(hipMalloc(&out, 64));
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, b, c);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, b, c);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double ns_per_instr = best * 1e6 / ((double)ITER * 8.0 * waves_per_simd);
    printf("%-12s waves/SIMD %d  %8.3f ms  %6.3f ns per wave-instruction per SIMD  x%.2f\n", name, waves_per_simd, best, ns_per_instr, ns_per_instr / base);
    CHECK(hipFree(out));
    return 0;
}

static float f_bits(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
static double d_bits(uint32_t lo, uint32_t hi) { uint64_t u = ((uint64_t)hi << 32) | lo; double d; __builtin_memcpy(&d, &u, 8); return d; }

int main()
{
    for (int w : {4}) {
        double base = 0;
        run("fma_f32", k_fma32, base, w);
        run_bc("pkmul_norm", k_pkmul_bc, d_bits(0x40400000u, 0x40400000u), d_bits(0x7e800000u, 0x3f000000u), base, w);
        run_bc("pkmul_denorm", k_pkmul_bc, d_bits(0x00000037u, 0x80000091u), d_bits(0x7e800000u, 0x7e800000u), base, w);
        run_bc("mul_norm", k_mul_bc, f_bits(0x40400000u), f_bits(0x7e800000u), base, w);
        run_bc("mul_denorm", k_mul_bc, f_bits(0x00000037u), f_bits(0x7e800000u), base, w);
        run_bc("subsdwa_bytes", k_subsdwa_bc, f_bits(0x00003700u), f_bits(0x00009100u), base, w);
        run("max_f32", k_max32, base, w);
        run("cnd_e64_vcc", k_cndv3, base, w);
        run("cnd_e32_0", k_cndc, base, w);
        run("cnd_e32_indep", k_cndb, base, w);
        run("sub_f32_sdwa", k_subf_sdwa, base, w);
        run("sub_u32", k_subu, base, w);
        run("fmac_f32", k_fmac32, base, w);
        run("max3_f32", k_max3, base, w);
        run("min_f32_e64abs", k_mine64, base, w);
        run("mul_f32_e64abs", k_mule64, base, w);
        run("or_b32", k_or, base, w);
        run("bfi_b32", k_bfi, base, w);
        run("lshr_b32", k_lshr, base, w);
        run("and_or_b32", k_andor, base, w);
        run("add_f32", k_add32, base, w);
        run("mul_f32", k_mul32, base, w);
        run("mul_f32clamp", k_mulclamp, base, w);
        run("mov_b32", k_mov32, base, w);
        run("mov_dpp", k_dpp, base, w);
        run("add_u32", k_addu, base, w);
        run("lshl_b32", k_lshl, base, w);
        run("xor_b32", k_xor, base, w);
        run("perm_b32", k_perm, base, w);
        run("bfe_u32", k_bfe, base, w);
        run("cndmask_sgpr", k_cnds, base, w);
        run("med3_f32", k_med3, base, w);
        run("max_i32", k_maxi, base, w);
        run("add3_u32", k_add3, base, w);
        run("mul_lo_u32", k_mullo, base, w);
        run("cvt_u32_f32", k_cvtu32, base, w);
        run("fract_f32", k_fract, base, w);
        run("sqrt_f32", k_sqrt, base, w);
        run("add_f64_sgpr", k_add64s, base, w);
        run("lshl_b64", k_lshl64, base, w);
        run("ldexp_f64", k_ldexp64, base, w);
        run("fract_f64", k_fract64, base, w);
        run("and_b32", k_and, base, w);
        run("cndmask", k_cndmask, base, w);
        run("cmp_f32", k_cmp32, base, w);
        run("cvt_f32_i32", k_cvti, base, w);
        run("cvt_f32_ub", k_cvtub, base, w);
        run("sub_sdwa", k_sdwa, base, w);
        run("rcp_f32", k_rcp32, base, w);
        run("pk_fma_f32", k_pkfma, base, w);
        run("pk_mul_f32", k_pkmul, base, w);
        run("pk_add_f32", k_pkadd, base, w);
        run("fma_f64", k_fma64, base, w);
        run("add_f64", k_add64, base, w);
        run("mul_f64", k_mul64, base, w);
        run("max_f64", k_max64, base, w);
        run("cmp_f64", k_cmp64, base, w);
        run("mov_b64", k_mov64, base, w);
        run("floor_f64", k_floor64, base, w);
        run("rcp_f64", k_rcp64, base, w);
        run("cvt_f64_f32", k_cvt64_32, base, w);
        run("cvt_f32_f64", k_cvt32_64, base, w);
        run("cvt_f64_i32", k_cvti64, base, w);
        run("ds_read_b32", k_ldsr32, base, w);
    }
    return 0;
}
