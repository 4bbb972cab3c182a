// cndmask.hip - what v_cndmask_b32 costs on gfx950 depending on where its lane mask comes from.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/cndmask.hip -o cndmask && ./cndmask
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int kIters = 2048, kUnroll = 4;
#define R8 "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *stamps, uint32_t seed)
{
    uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3u, r2 = r0 ^ 0x9E3779B9u, r3 = r1 ^ 0xBB67AE85u, r4 = r0 + 11u, r5 = r1 + 13u, r6 = r2 + 17u, r7 = r3 + 19u;
    const uint32_t c = 0xD2511F53u;
    unsigned long long m = 0x5555AAAA3333CCCCull ^ seed;
    asm volatile("s_mov_b64 vcc, %0\n s_mov_b64 s[20:21], %0" : : "s"(m) : "vcc", "s20", "s21");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (MODE == 0)      // vcc set once by SALU before the loop
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n" : R8 : "v"(c) : "vcc");
            if (MODE == 1)      // mask in an SGPR pair (VOP3 encoding)
                asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n"
                             "v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n" : R8 : "v"(c) : "s20", "s21");
            if (MODE == 2)      // cmp -> vcc, cndmask on a DIFFERENT register (independent), pairs
                asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n" : R8 : "v"(c) : "vcc");
            if (MODE == 3)      // 4 cmps into 4 SGPR pairs, then 4 cndmasks reading them
                asm volatile("v_cmp_lt_u32 s[20:21], %0, %8\n v_cmp_lt_u32 s[22:23], %2, %8\n v_cmp_lt_u32 s[24:25], %4, %8\n v_cmp_lt_u32 s[26:27], %6, %8\n"
                             "v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[22:23]\n v_cndmask_b32 %5, %5, %8, s[24:25]\n v_cndmask_b32 %7, %7, %8, s[26:27]\n"
                             : R8 : "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            if (MODE == 4)      // v_addc_co_u32 with carry-in from an SGPR pair set outside the loop
                asm volatile("v_addc_co_u32 %0, s[22:23], 0, %0, s[20:21]\n v_addc_co_u32 %1, s[22:23], 0, %1, s[20:21]\n v_addc_co_u32 %2, s[22:23], 0, %2, s[20:21]\n"
                             "v_addc_co_u32 %3, s[22:23], 0, %3, s[20:21]\n v_addc_co_u32 %4, s[22:23], 0, %4, s[20:21]\n v_addc_co_u32 %5, s[22:23], 0, %5, s[20:21]\n"
                             "v_addc_co_u32 %6, s[22:23], 0, %6, s[20:21]\n v_addc_co_u32 %7, s[22:23], 0, %7, s[20:21]\n" : R8 : "v"(c) : "s20", "s21", "s22", "s23");
            if (MODE == 5)      // exec-masked add: s_mov exec (SALU) + v_add under the mask + restore: the "branch-free if" alternative
                asm volatile("s_mov_b64 s[22:23], exec\n s_and_b64 exec, exec, s[20:21]\n v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n s_mov_b64 exec, s[22:23]\n"
                             "s_and_b64 exec, exec, s[20:21]\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n s_mov_b64 exec, s[22:23]\n"
                             "s_and_b64 exec, exec, s[20:21]\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n s_mov_b64 exec, s[22:23]\n"
                             "s_and_b64 exec, exec, s[20:21]\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n s_mov_b64 exec, s[22:23]\n" : R8 : "v"(c) : "s22", "s23", "scc");
            if (MODE == 6)      // v_cmp writing an SGPR pair, nothing reads it (4) + 4 adds
                asm volatile("v_cmp_lt_u32 s[20:21], %0, %8\n v_add_u32 %1, %1, %8\n v_cmp_lt_u32 s[22:23], %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_cmp_lt_u32 s[24:25], %4, %8\n v_add_u32 %5, %5, %8\n v_cmp_lt_u32 s[26:27], %6, %8\n v_add_u32 %7, %7, %8\n"
                             : R8 : "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            if (MODE == 8)      // vcc set before the loop; cndmask (e32, vcc) alternating with v_add
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_add_u32 %1, %1, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_u32 %3, %3, %8\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_add_u32 %5, %5, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_add_u32 %7, %7, %8\n" : R8 : "v"(c) : "vcc");
            if (MODE == 9)      // one cmp -> vcc, then 3 cndmasks on it, then 4 adds
                asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n" : R8 : "v"(c) : "vcc");
            if (MODE == 10)     // one cmp -> vcc, then 7 cndmasks on it
                asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n" : R8 : "v"(c) : "vcc");
            if (MODE == 11)     // 8 cndmasks (e32, vcc), each with src0 != dst (no same-register chain): dst r_i = vcc ? c : r_{i+1}
                asm volatile("v_cndmask_b32 %0, %1, %8, vcc\n v_cndmask_b32 %1, %2, %8, vcc\n v_cndmask_b32 %2, %3, %8, vcc\n v_cndmask_b32 %3, %4, %8, vcc\n"
                             "v_cndmask_b32 %4, %5, %8, vcc\n v_cndmask_b32 %5, %6, %8, vcc\n v_cndmask_b32 %6, %7, %8, vcc\n v_cndmask_b32 %7, %0, %8, vcc\n" : R8 : "v"(c) : "vcc");
            if (MODE == 12)     // 8 v_addc_co_u32 e32 (vcc in, vcc out)
                asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_addc_co_u32 %3, vcc, 0, %3, vcc\n"
                             "v_addc_co_u32 %4, vcc, 0, %4, vcc\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n v_addc_co_u32 %6, vcc, 0, %6, vcc\n v_addc_co_u32 %7, vcc, 0, %7, vcc\n" : R8 : "v"(c) : "vcc");
            if (MODE == 7)      // v_cmp + s_bcnt1 of its result + s_add (ballot/popcount accumulate on the scalar unit)
                asm volatile("v_cmp_lt_u32 s[20:21], %0, %8\n v_add_u32 %1, %1, %8\n s_bcnt1_i32_b64 s24, s[20:21]\n s_add_u32 s25, s25, s24\n"
                             "v_cmp_lt_u32 s[22:23], %2, %8\n v_add_u32 %3, %3, %8\n s_bcnt1_i32_b64 s24, s[22:23]\n s_add_u32 s25, s25, s24\n"
                             "v_cmp_lt_u32 s[20:21], %4, %8\n v_add_u32 %5, %5, %8\n s_bcnt1_i32_b64 s24, s[20:21]\n s_add_u32 s25, s25, s24\n"
                             "v_cmp_lt_u32 s[22:23], %6, %8\n v_add_u32 %7, %7, %8\n s_bcnt1_i32_b64 s24, s[22:23]\n s_add_u32 s25, s25, s24\n"
                             : R8 : "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25", "scc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    if ((threadIdx.x & 63) == 0) { const size_t w = (size_t)blockIdx.x * 4 + threadIdx.x / 64; stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = w1 - w0; }
}
template <int MODE> void run(const char *name, int valu_per_block)
{
    const int blocks = 256 * 8;
    uint32_t *d; unsigned long long *st;
    (void)hipMalloc(&d, (size_t)blocks * 256 * 4); (void)hipMalloc(&st, (size_t)blocks * 4 * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, st, 1u + r);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, st, 7u + r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    std::vector<unsigned long long> h((size_t)blocks * 8);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (size_t i = 0; i < h.size() / 2; ++i) if (h[2 * i + 1]) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
    std::sort(clk.begin(), clk.end());
    const double mc = clk[clk.size() / 2];
    const double blocks_per_simd = (double)kIters * kUnroll * 8;
    const double cyc = ms * 1e-3 * mc * 1e6 / blocks_per_simd;
    printf("%-70s %7.3f ms  clock %4.0f MHz  %6.2f cycles per asm block per SIMD (%d VALU instructions: %.2f each)\n", name, ms, mc, cyc, valu_per_block, cyc / valu_per_block);
    fflush(stdout);
    (void)hipFree(d); (void)hipFree(st);
}
int main()
{
    run<0>("8 v_cndmask, vcc set by SALU before the loop", 8);
    run<1>("8 v_cndmask (VOP3), SGPR-pair mask set before the loop", 8);
    run<2>("4 x (v_cmp -> vcc, v_cndmask vcc)", 8);
    run<3>("4 v_cmp -> 4 SGPR pairs, then 4 v_cndmask reading them", 8);
    run<4>("8 v_addc_co_u32 with SGPR carry-in set before the loop", 8);
    run<5>("4 x (s_and exec, 2 v_add, restore exec)", 8);
    run<6>("4 v_cmp -> SGPR pairs (unread) + 4 v_add", 8);
    run<7>("4 x (v_cmp -> SGPR, v_add, s_bcnt1, s_add)", 8);
    run<8>("4 x (v_cndmask e32 vcc, v_add), vcc set before the loop", 8);
    run<9>("v_cmp -> vcc, 3 v_cndmask vcc, 4 v_add", 8);
    run<10>("v_cmp -> vcc, 7 v_cndmask vcc", 8);
    run<11>("8 v_cndmask e32 vcc, dst != src0", 8);
    run<12>("8 v_addc_co_u32 e32 (vcc -> vcc)", 8);
    return 0;
}
