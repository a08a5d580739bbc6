// Measured HBM read / copy bandwidth on the box (reference point for roofline fractions; not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ src, size_t n16, unsigned* sink) {
    // each wave reads UNROLL consecutive 1 KiB pieces per iteration (like one attention K/V tile when UNROLL = 8)
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    unsigned acc = 0;
    for (size_t base = wave * UNROLL * 64; base + UNROLL * 64 <= n16; base += nwaves * UNROLL * 64) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) v[i] = NT ? __builtin_nontemporal_load(src + base + i * 64 + lane) : src[base + i * 64 + lane];
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < reps; ++r) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const size_t bytes = (size_t)4 << 30, n16 = bytes / 16;        // 4 GiB >> 256 MiB Infinity Cache
    u32x4 *src, *dst; unsigned* sink;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&sink, 4);
    hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
    for (int grid : {1024, 2048, 4096, 8192}) {
        float t8 = timeit([&] { hipLaunchKernelGGL((read_kernel<8, false>), dim3(grid), dim3(256), 0, 0, src, n16, sink); }, 5);
        float t8n = timeit([&] { hipLaunchKernelGGL((read_kernel<8, true>), dim3(grid), dim3(256), 0, 0, src, n16, sink); }, 5);
        float t16 = timeit([&] { hipLaunchKernelGGL((read_kernel<16, true>), dim3(grid), dim3(256), 0, 0, src, n16, sink); }, 5);
        float tc = timeit([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, src, dst, n16); }, 5);
        printf("grid %5d: read 8x1KiB/wave %.2f TB/s | nt %.2f TB/s | nt 16x1KiB %.2f TB/s | copy (r+w) %.2f TB/s\n", grid,
               bytes / t8 / 1e9, bytes / t8n / 1e9, bytes / t16 / 1e9, 2.0 * bytes / tc / 1e9);
    }
    // a 150 MB read, the size of one attention launch at ctx 570 (includes launch + tail)
    const size_t small = (size_t)150 << 20;
    float ts = timeit([&] { hipLaunchKernelGGL((read_kernel<16, true>), dim3(1024), dim3(256), 0, 0, src, small / 16, sink); }, 20);
    printf("150 MB read as one launch (1024 WGs): %.1f us = %.2f TB/s\n", ts * 1e3, small / ts / 1e9);
    return 0;
}
