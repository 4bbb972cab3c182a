// Throughput of the integer multiply forms Philox can use on gfx950, with 8 waves per SIMD resident
// (256 CUs x 8 blocks x 256 threads), each lane running independent chains.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mul_rates.hip -o mul_rates && ./mul_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int kIters = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a = seed + threadIdx.x, b = seed * 3u + blockIdx.x, c = a ^ 0x9E3779B9u, d = b ^ 0xBB67AE85u;
#pragma unroll 1
    for (int i = 0; i < kIters; ++i) {
        if (MODE == 0) {          // v_mad_u64_u32 (both halves), 4 independent chains
            uint64_t p0 = (uint64_t)0xD2511F53u * a, p1 = (uint64_t)0xCD9E8D57u * b, p2 = (uint64_t)0xD2511F53u * c, p3 = (uint64_t)0xCD9E8D57u * d;
            a = (uint32_t)(p0 >> 32) ^ (uint32_t)p1; b = (uint32_t)(p1 >> 32) ^ (uint32_t)p2;
            c = (uint32_t)(p2 >> 32) ^ (uint32_t)p3; d = (uint32_t)(p3 >> 32) ^ (uint32_t)p0;
        } else if (MODE == 1) {   // v_mul_hi_u32 only
            a = __umulhi(a, 0xD2511F53u) ^ b; b = __umulhi(b, 0xCD9E8D57u) ^ c; c = __umulhi(c, 0xD2511F53u) ^ d; d = __umulhi(d, 0xCD9E8D57u) ^ a;
        } else if (MODE == 2) {   // v_mul_lo_u32 only
            a = a * 0xD2511F53u ^ b; b = b * 0xCD9E8D57u ^ c; c = c * 0xD2511F53u ^ d; d = d * 0xCD9E8D57u ^ a;
        } else if (MODE == 3) {   // plain VALU reference: xor/add
            a = (a + 0xD2511F53u) ^ b; b = (b + 0xCD9E8D57u) ^ c; c = (c + 0xD2511F53u) ^ d; d = (d + 0xCD9E8D57u) ^ a;
        } else if (MODE == 4) {   // v_mul_u32_u24 (24-bit, full rate?)
            a = __umul24(a, 0x511F53u) ^ b; b = __umul24(b, 0x9E8D57u) ^ c; c = __umul24(c, 0x511F53u) ^ d; d = __umul24(d, 0x9E8D57u) ^ a;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int MODE>
double run(const char *name, int mul_per_iter, int other_per_iter)
{
    const int blocks = 256 * 8;
    uint32_t *d;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 7u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double waves_per_simd = blocks * 4.0 / (256 * 4);
    const double clk = 2.4e9;      // nominal; the ratio between modes is what matters
    const double cycles = ms * 1e-3 * clk;
    const double per_iter = cycles / kIters / waves_per_simd;      // SIMD cycles per wave-iteration
    printf("%-28s %.3f ms  %.1f SIMD-cycles per wave-iteration (%d mul + %d other wave-instr)\n", name, ms, per_iter, mul_per_iter, other_per_iter);
    hipFree(d);
    return per_iter;
}

int main()
{
    const double base = run<3>("add+xor (8 plain VALU)", 0, 8);
    const double mad = run<0>("v_mad_u64_u32 x4 + 4 xor", 4, 4);
    const double hi = run<1>("v_mul_hi_u32 x4 + 4 xor", 4, 4);
    const double lo = run<2>("v_mul_lo_u32 x4 + 4 xor", 4, 4);
    const double m24 = run<4>("v_mul_u32_u24 x4 + 4 xor", 4, 4);
    const double plain = base / 8.0;
    printf("plain VALU ~ %.2f cycles; v_mad_u64_u32 ~ %.1f; v_mul_hi_u32 ~ %.1f; v_mul_lo_u32 ~ %.1f; v_mul_u32_u24 ~ %.1f cycles per wave-instruction\n",
           plain, (mad - 4 * plain) / 4, (hi - 4 * plain) / 4, (lo - 4 * plain) / 4, (m24 - 4 * plain) / 4);
    return 0;
}
