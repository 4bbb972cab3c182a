// Which HIP streams share a hardware queue?  Creates n streams (optionally touching the null stream after the k-th), then for every
// pair launches a ~300 us spin kernel on both and times the pair: streams on one queue serialise (2x), on two they overlap (1x).
// Build: hipcc --offload-arch=gfx950 -O2 tools/microbench/stream_queues.hip -o tools/microbench/stream_queues
// Run (GPU box): tools/microbench/stream_queues <n_streams> <null_after|-1> <priority_mask_hex> [use_first 0/1]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_spin(long long ticks, int *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (sink && ticks < 0) *sink = 1;
}
__global__ void k_touch(int *p) { if (p) *p = 0; }

int main(int argc, char **argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 8;
    const int null_after = argc > 2 ? std::atoi(argv[2]) : -1;
    const unsigned prio_mask = argc > 3 ? (unsigned)std::strtoul(argv[3], nullptr, 16) : 0u;
    const int touch_each = argc > 4 ? std::atoi(argv[4]) : 1;      // use every stream right after creating it
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    int *d = nullptr;
    hipMalloc((void **)&d, 4);
    std::vector<hipStream_t> st(n);
    for (int i = 0; i < n; ++i) {
        if ((prio_mask >> i) & 1u) hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, greatest);
        else hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        if (touch_each) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(1), 0, st[i], d); hipStreamSynchronize(st[i]); }
        if (i == null_after) { int h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); }      // the null stream comes into being
    }
    if (!touch_each) for (int i = n - 1; i >= 0; --i) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(1), 0, st[i], d); hipStreamSynchronize(st[i]); }   // first use in reverse order
    const long long ticks = 300 * 100;          // wall_clock64: 100 MHz
    auto pair_us = [&](int a, int b) {
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[a], ticks, d);
        if (b != a) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[b], ticks, d);
        hipStreamSynchronize(st[a]);
        hipStreamSynchronize(st[b]);
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    pair_us(0, 0);
    std::printf("n=%d null_after=%d prio_mask=%x touch_each=%d  (S = the pair serialised)\n", n, null_after, prio_mask, touch_each);
    for (int a = 0; a < n; ++a) {
        std::printf("%2d%s ", a, ((prio_mask >> a) & 1u) ? "h" : " ");
        for (int b = 0; b < n; ++b) {
            if (b <= a) { std::printf("  ."); continue; }
            const double us = pair_us(a, b);
            std::printf("  %c", us > 480.0 ? 'S' : '-');
        }
        std::printf("\n");
    }
    // four at once: streams 0..3, then the last four
    auto quad_us = [&](int first) {
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[first + i], ticks, d);
        for (int i = 0; i < 4; ++i) hipStreamSynchronize(st[first + i]);
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    if (n >= 4) std::printf("four at once: streams 0-3 %.0f us, streams %d-%d %.0f us\n", quad_us(0), n - 4, n - 1, quad_us(n - 4));
    return 0;
}
