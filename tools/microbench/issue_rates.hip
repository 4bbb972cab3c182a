// issue_rates.hip - what one VALU wave-instruction really costs on gfx950, in SHADER cycles (s_memtime) and in wall time,
// and what clock the chip holds while doing it (s_memtime ticks per s_memrealtime tick x 100 MHz).
//
// Reconciles profiles/r01_microbench_mul_rates.txt ("2.99 cycles per plain VALU wave-instruction", computed from wall
// time at a NOMINAL 2.4 GHz) with the guide's 2 cycles (v_fma_f32, wave64 on a SIMD-32).
//
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/issue_rates.hip -o issue_rates && ./issue_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int kIters = 2048;        // loop trips; each trip = kUnroll x 8 instructions
constexpr int kUnroll = 4;

// MODE 0: 8 independent v_add_u32 chains        MODE 1: ONE dependent v_add_u32 chain (8 in a row on one register)
// MODE 2: 8 independent v_fma_f32               MODE 3: 8 independent v_mad_u64_u32
// MODE 4: 8 independent v_mul_hi_u32            MODE 5: 8 independent v_mul_lo_u32
// MODE 6: 8 independent v_bitop3_b32 (xor3)     MODE 7: 4 v_add_u32 + 4 s_nop 0 interleaved
// MODE 8: Philox-like round mix: 2 v_mad_u64_u32 + 2 bitop3 + 4 v_add_u32 (independent)
// MODE 9: 8 v_add_u32 + 4 independent s_add_u32 (SALU beside VALU)
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *stamps, uint32_t seed)
{
    uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3u, r2 = r0 ^ 0x9E3779B9u, r3 = r1 ^ 0xBB67AE85u, r4 = r0 + 11u, r5 = r1 + 13u,
             r6 = r2 + 17u, r7 = r3 + 19u;
    float f0 = (float)r0, f1 = (float)r1, f2 = (float)r2, f3 = (float)r3, f4 = 1.5f, f5 = 2.5f, f6 = 3.5f, f7 = 4.5f;
    unsigned long long q0 = r0, q1 = r1, q2 = r2, q3 = r3, q4 = r4, q5 = r5, q6 = r6, q7 = r7;
    const uint32_t c = 0xD2511F53u;
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (MODE == 0) {
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
            } else if (MODE == 1) {
                asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                             "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                             : "+v"(r0) : "v"(c));
            } else if (MODE == 2) {
                asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                             "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(0.999f));
            } else if (MODE == 3) {
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, 0\n v_mad_u64_u32 %1, vcc, %8, %10, 0\n v_mad_u64_u32 %2, vcc, %8, %11, 0\n"
                             "v_mad_u64_u32 %3, vcc, %8, %12, 0\n v_mad_u64_u32 %4, vcc, %8, %9, 0\n v_mad_u64_u32 %5, vcc, %8, %10, 0\n"
                             "v_mad_u64_u32 %6, vcc, %8, %11, 0\n v_mad_u64_u32 %7, vcc, %8, %12, 0\n"
                             : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7)
                             : "v"(c), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "vcc");
            } else if (MODE == 4) {
                asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                             "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
            } else if (MODE == 5) {
                asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                             "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
            } else if (MODE == 6) {
                asm volatile("v_bitop3_b32 %0, %0, %8, %1 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %2 bitop3:0x96\n"
                             "v_bitop3_b32 %2, %2, %8, %3 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %4 bitop3:0x96\n"
                             "v_bitop3_b32 %4, %4, %8, %5 bitop3:0x96\n v_bitop3_b32 %5, %5, %8, %6 bitop3:0x96\n"
                             "v_bitop3_b32 %6, %6, %8, %7 bitop3:0x96\n v_bitop3_b32 %7, %7, %8, %0 bitop3:0x96\n"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
            } else if (MODE == 7) {
                asm volatile("v_add_u32 %0, %0, %4\n s_nop 0\n v_add_u32 %1, %1, %4\n s_nop 0\n v_add_u32 %2, %2, %4\n s_nop 0\n"
                             "v_add_u32 %3, %3, %4\n s_nop 0\n"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(c));
            } else if (MODE == 8) {
                asm volatile("v_mad_u64_u32 %0, vcc, %6, %2, 0\n v_mad_u64_u32 %1, vcc, %6, %3, 0\n"
                             "v_bitop3_b32 %2, %2, %6, %4 bitop3:0x96\n v_bitop3_b32 %3, %3, %6, %5 bitop3:0x96\n"
                             "v_add_u32 %4, %4, %6\n v_add_u32 %5, %5, %6\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %6\n"
                             : "+v"(q0), "+v"(q1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5) : "v"(c) : "vcc");
            } else if (MODE == 9) {
                asm volatile("v_add_u32 %0, %0, %12\n s_add_u32 %8, %8, 3\n v_add_u32 %1, %1, %12\n v_add_u32 %2, %2, %12\n s_add_u32 %9, %9, 5\n"
                             "v_add_u32 %3, %3, %12\n v_add_u32 %4, %4, %12\n s_add_u32 %10, %10, 7\n v_add_u32 %5, %5, %12\n"
                             "v_add_u32 %6, %6, %12\n s_add_u32 %11, %11, 9\n v_add_u32 %7, %7, %12\n"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                             : "v"(c) : "scc");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ s0 ^ s1 ^ s2 ^ s3;
    acc ^= __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    acc ^= (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ (uint32_t)((q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {       // stamps go to a buffer of their own
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = w1 - w0;
    }
}

static const char *kNames[] = {"v_add_u32 x8 independent", "v_add_u32 x8 ONE dependent chain", "v_fma_f32 x8 independent", "v_mad_u64_u32 x8 independent",
                               "v_mul_hi_u32 x8 independent", "v_mul_lo_u32 x8 independent", "v_bitop3_b32 x8", "4 v_add_u32 + 4 s_nop 0",
                               "philox-like: 2 mad_u64 + 2 bitop3 + 4 add", "8 v_add_u32 + 4 s_add_u32"};
static const int kValuPerTrip[] = {8, 8, 8, 8, 8, 8, 8, 4, 8, 8};

template <int MODE>
void run(int waves_per_simd, int seconds_of_warm)
{
    // waves_per_simd w: one CU hosts 4*w waves = w blocks of 256 threads
    const int blocks = 256 * waves_per_simd;
    uint32_t *d;
    unsigned long long *st;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipMalloc(&st, (size_t)blocks * 4 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3 + seconds_of_warm * 200; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, st, 1u + r);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, st, 7u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<unsigned long long> h((size_t)blocks * 4 * 2);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (size_t i = 0; i < h.size() / 2; ++i) {
        cyc.push_back((double)h[2 * i]);
        if (h[2 * i + 1]) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);     // MHz (s_memrealtime ticks at 100 MHz)
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const double med_cyc = cyc[cyc.size() / 2], med_clk = clk.empty() ? 0.0 : clk[clk.size() / 2];
    const double instr_per_wave = (double)kIters * kUnroll * kValuPerTrip[MODE];
    // in-kernel: shader cycles per wave-instruction PER SIMD = wave's cycles / its instructions / waves sharing the SIMD
    const double cyc_per_instr = med_cyc / instr_per_wave / waves_per_simd;
    // wall: the same from the kernel's duration at the measured clock
    const double wall_cyc_per_instr = ms * 1e-3 * med_clk * 1e6 / instr_per_wave / waves_per_simd;
    printf("%-44s w/SIMD %d  %.3f ms  clock %.0f MHz  %.2f cyc/wave-instr/SIMD (s_memtime)  %.2f (wall x clock)  %.2f at nominal 2400\n",
           kNames[MODE], waves_per_simd, ms, med_clk, cyc_per_instr, wall_cyc_per_instr, ms * 1e-3 * 2.4e9 / instr_per_wave / waves_per_simd);
    hipFree(d); hipFree(st);
}

int main()
{
    // warm the chip so DVFS has settled (the guide: stamp after >= 2 s of back-to-back launches)
    run<0>(8, 2);
    for (int w : {1, 2, 4, 8}) run<0>(w, 0);
    for (int w : {1, 2, 4, 8}) run<1>(w, 0);
    for (int w : {1, 2, 8}) run<2>(w, 0);
    for (int w : {1, 2, 8}) run<3>(w, 0);
    for (int w : {2, 8}) run<4>(w, 0);
    for (int w : {2, 8}) run<5>(w, 0);
    for (int w : {2, 8}) run<6>(w, 0);
    for (int w : {1, 8}) run<7>(w, 0);
    for (int w : {2, 8}) run<8>(w, 0);
    for (int w : {2, 8}) run<9>(w, 0);
    return 0;
}
