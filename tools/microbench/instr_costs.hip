// instr_costs.hip - cost table of the wave-instructions the step kernels are made of, on gfx950, at 8 waves/SIMD requested
// (the regime of k_step_implicit_fast).  For each instruction: 8 independent copies per trip, unrolled x4, 2048 trips;
// reported = kernel wall time x measured shader clock / instructions per SIMD, minus nothing (loop overhead is 3 SALU per
// 32 instructions).  The clock is Delta s_memtime / Delta s_memrealtime x 100 MHz.
//
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/instr_costs.hip -o instr_costs && ./instr_costs
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int kIters = 2048;
constexpr int kUnroll = 4;

#define R8 "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
#define F8 "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
// one instruction template applied to 8 registers; %8, %9 are two extra read-only VGPRs
#define REP8(T) T("%0") T("%1") T("%2") T("%3") T("%4") T("%5") T("%6") T("%7")

#define I_ADD(x) "v_add_u32 " x ", " x ", %8\n"
#define I_AND(x) "v_and_b32 " x ", " x ", %8\n"
#define I_LSHR(x) "v_lshrrev_b32 " x ", 3, " x "\n"
#define I_ADD3(x) "v_add3_u32 " x ", " x ", %8, %9\n"
#define I_LSHLADD(x) "v_lshl_add_u32 " x ", " x ", 2, %8\n"
#define I_ANDOR(x) "v_and_or_b32 " x ", " x ", %8, %9\n"
#define I_BFE(x) "v_bfe_u32 " x ", " x ", 3, 8\n"
#define I_XOR3(x) "v_bitop3_b32 " x ", " x ", %8, %9 bitop3:0x96\n"
#define I_CNDMASK(x) "v_cndmask_b32 " x ", " x ", %8, vcc\n"
#define I_CMP(x) "v_cmp_lt_u32 vcc, " x ", %8\n"
#define I_CMP_S(x) "v_cmp_lt_u32 s[20:21], " x ", %8\n"
#define I_CMPX(x) "v_cmp_lt_u32 vcc, " x ", %8\n v_addc_co_u32 " x ", vcc, 0, " x ", vcc\n"
#define I_MULHI(x) "v_mul_hi_u32 " x ", " x ", %8\n"
#define I_MULLO(x) "v_mul_lo_u32 " x ", " x ", %8\n"
#define I_MUL24(x) "v_mul_u32_u24 " x ", " x ", %8\n"
#define I_MAD24(x) "v_mad_u32_u24 " x ", " x ", %8, %9\n"
#define I_MBCNT(x) "v_mbcnt_lo_u32_b32 " x ", s20, " x "\n"
#define I_CVT_F_U(x) "v_cvt_f32_u32 " x ", " x "\n"
#define I_CVT_U_F(x) "v_cvt_u32_f32 " x ", " x "\n"
#define I_FADD(x) "v_add_f32 " x ", " x ", %8\n"
#define I_FMUL(x) "v_mul_f32 " x ", " x ", %8\n"
#define I_FMAC(x) "v_fmac_f32 " x ", %8, %9\n"
#define I_FMA3(x) "v_fma_f32 " x ", " x ", %8, %9\n"
#define I_FMAK(x) "v_fmaak_f32 " x ", " x ", %8, 0x3f000000\n"
#define I_RNDNE(x) "v_rndne_f32 " x ", " x "\n"
#define I_FMIN(x) "v_min_f32 " x ", " x ", %8\n"
#define I_SQRT(x) "v_sqrt_f32 " x ", " x "\n"
#define I_RCP(x) "v_rcp_f32 " x ", " x "\n"
#define I_LOG(x) "v_log_f32 " x ", " x "\n"
#define I_EXP(x) "v_exp_f32 " x ", " x "\n"
#define I_PKFMA(x) "v_pk_fma_f32 " x ", " x ", %8, %9\n"
#define I_PKMUL(x) "v_pk_mul_f32 " x ", " x ", %8\n"
#define I_PKADD(x) "v_pk_add_f32 " x ", " x ", %8\n"
#define I_DFMA(x) "v_fma_f64 " x ", " x ", %8, %9\n"
#define I_DMUL(x) "v_mul_f64 " x ", " x ", %8\n"
#define I_DADD(x) "v_add_f64 " x ", " x ", %8\n"
#define I_LSHL64(x) "v_lshlrev_b64 " x ", 3, " x "\n"
#define I_CND1(x) "v_cndmask_b32 " x ", " x ", %8, vcc\n"
#define I_SUBCO(x) "v_sub_co_u32 " x ", vcc, " x ", %8\n"
#define I_ALIGNBIT(x) "v_alignbit_b32 " x ", " x ", %8, 7\n"
#define I_XOR(x) "v_xor_b32 " x ", " x ", %8\n"
#define I_OR(x) "v_or_b32 " x ", " x ", %8\n"
#define I_LSHL(x) "v_lshlrev_b32 " x ", 3, " x "\n"
#define I_ASHR(x) "v_ashrrev_i32 " x ", 3, " x "\n"
#define I_SUB(x) "v_sub_u32 " x ", " x ", %8\n"
#define I_MAXU(x) "v_max_u32 " x ", " x ", %8\n"
#define I_MINU(x) "v_min_u32 " x ", " x ", %8\n"
#define I_MED3(x) "v_med3_i32 " x ", " x ", %8, %9\n"
#define I_BFI(x) "v_bfi_b32 " x ", %8, " x ", %9\n"
#define I_LSHLOR(x) "v_lshl_or_b32 " x ", " x ", 1, %8\n"
#define I_XAD(x) "v_xad_u32 " x ", " x ", %8, %9\n"
#define I_CVT_F_I(x) "v_cvt_f32_i32 " x ", " x "\n"
#define I_CVT_I_F(x) "v_cvt_i32_f32 " x ", " x "\n"
#define I_FREXPE(x) "v_frexp_exp_i32_f32 " x ", " x "\n"
#define I_FREXPM(x) "v_frexp_mant_f32 " x ", " x "\n"
#define I_LDEXP(x) "v_ldexp_f32 " x ", " x ", %8\n"
#define I_FSUB(x) "v_sub_f32 " x ", " x ", %8\n"
#define I_FMAX(x) "v_max_f32 " x ", " x ", %8\n"
#define I_FMULABS(x) "v_mul_f32 " x ", |" x "|, %8\n"
#define I_MOV(x) "v_mov_b32 " x ", %8\n"
#define I_MADU64(x) "v_mad_u64_u32 " x ", vcc, %8, %9, 0\n"
#define I_DSREAD(x) "ds_read_b32 " x ", " x "\n"
#define I_DSREAD64(x) "ds_read_b64 " x ", %8\n"

enum Kind { K_U32, K_F32, K_U64, K_F64, K_LDS32, K_LDS64 };

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *stamps, uint32_t seed)
{
    __shared__ uint32_t lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = (i * 4 + 64) & 8191 & ~3;      // pointer-chasing table of valid byte offsets
    uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3u, r2 = r0 ^ 0x9E3779B9u, r3 = r1 ^ 0xBB67AE85u, r4 = r0 + 11u, r5 = r1 + 13u,
             r6 = r2 + 17u, r7 = r3 + 19u;
    float f0 = 1.0f + (float)(r0 & 255), f1 = 2.0f, f2 = 3.0f, f3 = 4.0f, f4 = 1.5f, f5 = 2.5f, f6 = 3.5f, f7 = 4.5f;
    unsigned long long q0 = r0, q1 = r1, q2 = r2, q3 = r3, q4 = r4, q5 = r5, q6 = r6, q7 = r7;
    double d0 = f0, d1 = 2.0, d2 = 3.0, d3 = 4.0, d4 = 1.5, d5 = 2.5, d6 = 3.5, d7 = 4.5;
    const uint32_t c = 0xD2511F53u, c2 = 0x00FFFF00u;
    const float fc = 0.999f, fc2 = 1e-3f;
    const double dc = 0.999, dc2 = 1e-3;
    const unsigned long long qc = 0x3f7fbe773f7fbe77ull, qc2 = 0x3a83126f3a83126full;    // packed (0.999f, 0.999f), (1e-3f, 1e-3f)
    if (MODE >= 101) { r0 = (threadIdx.x * 8) & 8184; r1 = r0 ^ 4096; r2 = ((threadIdx.x * 2654435761u) >> 17) & 8184; r3 = ((threadIdx.x * 40503u + 77u) * 8u) & 8184; }
    else if (MODE >= 100) { r0 = (threadIdx.x * 4) & 8188; r1 = r0 ^ 64; r2 = r0 ^ 128; r3 = r0 ^ 256; r4 = r0 ^ 512; r5 = r0 ^ 1024; r6 = r0 ^ 2048; r7 = r0 ^ 4096; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
#define U32CASE(M, T) if (MODE == M) asm volatile(REP8(T) : R8 : "v"(c), "v"(c2) : "vcc", "s20", "s21");
#define F32CASE(M, T) if (MODE == M) asm volatile(REP8(T) : F8 : "v"(fc), "v"(fc2) : "vcc");
            U32CASE(0, I_ADD) U32CASE(1, I_AND) U32CASE(2, I_LSHR) U32CASE(3, I_ADD3) U32CASE(4, I_LSHLADD) U32CASE(5, I_ANDOR)
            U32CASE(6, I_BFE) U32CASE(7, I_XOR3) U32CASE(8, I_CNDMASK) U32CASE(9, I_CMP) U32CASE(10, I_CMP_S) U32CASE(11, I_CMPX)
            U32CASE(12, I_MULHI) U32CASE(13, I_MULLO) U32CASE(14, I_MUL24) U32CASE(15, I_MAD24) U32CASE(16, I_MBCNT)
            F32CASE(20, I_FADD) F32CASE(21, I_FMUL) F32CASE(22, I_FMAC) F32CASE(23, I_FMA3) F32CASE(24, I_FMAK) F32CASE(25, I_RNDNE)
            F32CASE(26, I_FMIN) F32CASE(27, I_SQRT) F32CASE(28, I_RCP) F32CASE(29, I_LOG) F32CASE(30, I_EXP) F32CASE(31, I_CVT_F_U)
            F32CASE(32, I_CVT_U_F)
            if (MODE == 60) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n" REP8(I_CND1) : R8 : "v"(c), "v"(c2) : "vcc");
            U32CASE(61, I_SUBCO) U32CASE(62, I_ALIGNBIT) U32CASE(63, I_XOR) U32CASE(64, I_OR) U32CASE(65, I_LSHL) U32CASE(66, I_ASHR)
            U32CASE(67, I_SUB) U32CASE(68, I_MAXU) U32CASE(69, I_MINU) U32CASE(70, I_MED3) U32CASE(71, I_BFI) U32CASE(72, I_LSHLOR)
            U32CASE(73, I_XAD) U32CASE(74, I_MOV)
            F32CASE(80, I_CVT_F_I) F32CASE(81, I_CVT_I_F) F32CASE(82, I_FREXPE) F32CASE(83, I_FREXPM) F32CASE(84, I_LDEXP) F32CASE(85, I_FSUB)
            F32CASE(86, I_FMAX) F32CASE(87, I_FMULABS)
            if (MODE == 90) asm volatile(REP8(I_MADU64) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(c), "v"(c2) : "vcc");
            if (MODE == 101) asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %9\n ds_read_b64 %2, %8\n ds_read_b64 %3, %9\n ds_read_b64 %4, %8\n ds_read_b64 %5, %9\n ds_read_b64 %6, %8\n ds_read_b64 %7, %9\n s_waitcnt lgkmcnt(0)\n" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(r0), "v"(r1) : "memory");
            if (MODE == 102) asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %9\n ds_read_b64 %2, %8\n ds_read_b64 %3, %9\n ds_read_b64 %4, %8\n ds_read_b64 %5, %9\n ds_read_b64 %6, %8\n ds_read_b64 %7, %9\n s_waitcnt lgkmcnt(0)\n" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(r2), "v"(r3) : "memory");
            if (MODE == 103) asm volatile("ds_write_b64 %8, %0\n ds_write_b64 %8, %1\n ds_write_b64 %8, %2\n ds_write_b64 %8, %3\n ds_write_b64 %8, %4\n ds_write_b64 %8, %5\n ds_write_b64 %8, %6\n ds_write_b64 %8, %7\n s_waitcnt lgkmcnt(0)\n" : : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "v"(r0) : "memory");
            if (MODE == 104) asm volatile("ds_add_u64 %8, %0\n ds_add_u64 %8, %1\n ds_add_u64 %8, %2\n ds_add_u64 %8, %3\n ds_add_u64 %8, %4\n ds_add_u64 %8, %5\n ds_add_u64 %8, %6\n ds_add_u64 %8, %7\n s_waitcnt lgkmcnt(0)\n" : : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "v"(r2) : "memory");
            if (MODE == 105) asm volatile("ds_add_u32 %8, %0\n ds_add_u32 %8, %1\n ds_add_u32 %8, %2\n ds_add_u32 %8, %3\n ds_add_u32 %8, %4\n ds_add_u32 %8, %5\n ds_add_u32 %8, %6\n ds_add_u32 %8, %7\n s_waitcnt lgkmcnt(0)\n" : : "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(r2) : "memory");
            if (MODE == 40) asm volatile(REP8(I_PKFMA) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(qc), "v"(qc2));
            if (MODE == 41) asm volatile(REP8(I_PKMUL) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(qc), "v"(qc2));
            if (MODE == 42) asm volatile(REP8(I_PKADD) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(qc), "v"(qc2));
            if (MODE == 43) asm volatile(REP8(I_LSHL64) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(qc), "v"(qc2));
            if (MODE == 50) asm volatile(REP8(I_DFMA) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc), "v"(dc2));
            if (MODE == 51) asm volatile(REP8(I_DMUL) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc), "v"(dc2));
            if (MODE == 52) asm volatile(REP8(I_DADD) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc), "v"(dc2));
            if (MODE == 100) asm volatile(REP8(I_DSREAD) "s_waitcnt lgkmcnt(0)\n" : R8 : "v"(c), "v"(c2) : "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ lds[threadIdx.x];
    acc ^= __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    acc ^= (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ (uint32_t)((q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) >> 32);
    acc ^= (uint32_t)__double2ll_rn(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = w1 - w0;
    }
}

static double g_base = 0.0;

template <int MODE>
void run(const char *name, int instr_per_template = 1, int blocks_per_cu = 8)
{
    const int blocks = 256 * blocks_per_cu;
    uint32_t *d;
    unsigned long long *st;
    (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&st, (size_t)blocks * 4 * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, st, 1u + r);
    (void)hipDeviceSynchronize();
    const int reps = 10;
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, st, 7u + r);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<unsigned long long> h((size_t)blocks * 4 * 2);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (size_t i = 0; i < h.size() / 2; ++i)
        if (h[2 * i + 1]) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
    std::sort(clk.begin(), clk.end());
    const double med_clk = clk.empty() ? 0.0 : clk[clk.size() / 2];
    const double templates_per_simd = (double)kIters * kUnroll * 8 * blocks_per_cu;     // blocks_per_cu waves land on each SIMD
    const double cyc = ms * 1e-3 * med_clk * 1e6 / templates_per_simd;
    if (MODE == 0) g_base = cyc;
    printf("%-34s %7.3f ms  clock %4.0f MHz  %5.2f cycles per wave-instruction per SIMD", name, ms, med_clk, cyc / instr_per_template);
    if (instr_per_template > 1) printf("  (%d instructions per template: %.2f for the pair)", instr_per_template, cyc);
    printf("  = %.2f x v_add_u32\n", cyc / instr_per_template / g_base);
    (void)hipFree(d); (void)hipFree(st);
}

int main()
{
    run<0>("v_add_u32 (VOP2)");
    run<0>("v_add_u32 (VOP2)");
    run<1>("v_and_b32 (VOP2)");
    run<2>("v_lshrrev_b32 (VOP2, inline const)");
    run<3>("v_add3_u32 (VOP3, 3 VGPR)");
    run<4>("v_lshl_add_u32 (VOP3, 2 VGPR)");
    run<5>("v_and_or_b32 (VOP3, 3 VGPR)");
    run<6>("v_bfe_u32 (VOP3, 1 VGPR)");
    run<7>("v_bitop3_b32 (VOP3, 3 VGPR)");
    run<8>("v_cndmask_b32 (VOP2, vcc)");
    run<9>("v_cmp_lt_u32 -> vcc");
    run<10>("v_cmp_lt_u32 -> sgpr pair (VOP3)");
    run<11>("v_cmp + v_addc_co_u32", 2);
    run<12>("v_mul_hi_u32");
    run<13>("v_mul_lo_u32");
    run<14>("v_mul_u32_u24");
    run<15>("v_mad_u32_u24 (VOP3, 3 VGPR)");
    run<16>("v_mbcnt_lo_u32_b32");
    run<20>("v_add_f32 (VOP2)");
    run<21>("v_mul_f32 (VOP2)");
    run<22>("v_fmac_f32 (VOP2)");
    run<23>("v_fma_f32 (VOP3, 3 VGPR)");
    run<24>("v_fmaak_f32 (VOP2 + literal)");
    run<25>("v_rndne_f32");
    run<26>("v_min_f32");
    run<27>("v_sqrt_f32");
    run<28>("v_rcp_f32");
    run<29>("v_log_f32");
    run<30>("v_exp_f32");
    run<31>("v_cvt_f32_u32");
    run<32>("v_cvt_u32_f32");
    run<40>("v_pk_fma_f32 (2 fma per lane)");
    run<41>("v_pk_mul_f32");
    run<42>("v_pk_add_f32");
    run<43>("v_lshlrev_b64");
    run<50>("v_fma_f64");
    run<51>("v_mul_f64");
    run<52>("v_add_f64");
    run<60>("v_cndmask_b32 (1 v_cmp + 8 cndmask)");
    run<61>("v_sub_co_u32");
    run<62>("v_alignbit_b32");
    run<63>("v_xor_b32");
    run<64>("v_or_b32");
    run<65>("v_lshlrev_b32");
    run<66>("v_ashrrev_i32");
    run<67>("v_sub_u32");
    run<68>("v_max_u32");
    run<69>("v_min_u32");
    run<70>("v_med3_i32");
    run<71>("v_bfi_b32");
    run<72>("v_lshl_or_b32");
    run<73>("v_xad_u32");
    run<74>("v_mov_b32");
    run<80>("v_cvt_f32_i32");
    run<81>("v_cvt_i32_f32");
    run<82>("v_frexp_exp_i32_f32");
    run<83>("v_frexp_mant_f32");
    run<84>("v_ldexp_f32");
    run<85>("v_sub_f32");
    run<86>("v_max_f32");
    run<87>("v_mul_f32 with |abs| modifier (VOP3)");
    run<90>("v_mad_u64_u32");
    run<100>("ds_read_b32 (8 dependent, then wait)");
    run<101>("ds_read_b64 conflict-free x8");
    run<102>("ds_read_b64 scattered x8");
    run<103>("ds_write_b64 conflict-free x8");
    run<104>("ds_add_u64 scattered x8");
    run<105>("ds_add_u32 scattered x8");
    return 0;
}
