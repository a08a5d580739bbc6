// Probe: do kernels of two HIP streams overlap on this part?  Stream A: HBM-bound "attention-like" launches (S), stream B: a chain of
// short latency-bound "GEMM-like" launches (G).  Reports S alone, G alone, both serial on one stream, both on two streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int WPS>
__global__ __launch_bounds__(256, WPS) void stream_kernel(const u4* src, size_t per_wg_vec, unsigned* sink) {
    const u4* p = src + (size_t)blockIdx.x * per_wg_vec;
    u4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < per_wg_vec; i += 256 * 4) {
        u4 a = __builtin_nontemporal_load(p + i), b = i + 256 < per_wg_vec ? __builtin_nontemporal_load(p + i + 256) : acc;
        u4 c = i + 512 < per_wg_vec ? __builtin_nontemporal_load(p + i + 512) : acc, d = i + 768 < per_wg_vec ? __builtin_nontemporal_load(p + i + 768) : acc;
        acc ^= a ^ b ^ c ^ d;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}
__global__ __launch_bounds__(256) void small_kernel(const u4* src, size_t per_wg_vec, unsigned* sink, unsigned short* out) {
    const u4* p = src + (size_t)blockIdx.x * per_wg_vec;
    u4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < per_wg_vec; i += 256) acc ^= __builtin_nontemporal_load(p + i);
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned short)(acc.x ^ acc.y ^ acc.z ^ acc.w);
    if ((acc.x ^ acc.y) == 0x12345u) *sink = 1;
}
int main(int argc, char** argv) {
    const int s_wgs = argc > 1 ? atoi(argv[1]) : 1024;          // WGs of the streaming kernel
    const size_t s_mb = argc > 2 ? atoi(argv[2]) : 146;         // MB per streaming launch
    const int wps = argc > 3 ? atoi(argv[3]) : 4;               // launch-bounds waves per SIMD of the streaming kernel (4 = fills the VGPR file at 128 regs)
    const int NL = 30, G_PER = 4;
    const size_t g_kb = 32;                                     // KB per small-kernel WG (256 WGs -> 8 MB per launch)
    u4 *sbuf, *gbuf; unsigned* sink; unsigned short* out;
    const size_t s_bytes = s_mb << 20, g_bytes = (size_t)256 * g_kb * 1024;
    CK(hipMalloc((void**)&sbuf, s_bytes * 4)); CK(hipMalloc((void**)&gbuf, g_bytes * 8)); CK(hipMalloc((void**)&sink, 4)); CK(hipMalloc((void**)&out, 256 * 256 * 2));
    CK(hipMemset(sbuf, 1, s_bytes * 4)); CK(hipMemset(gbuf, 2, g_bytes * 8));
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t e0, e1, eb; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&eb));
    auto S = [&](hipStream_t st, int l) {
        const u4* src = sbuf + (size_t)(l & 3) * (s_bytes / 16);
        if (wps == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(s_wgs), dim3(256), 0, st, src, s_bytes / 16 / s_wgs, sink);
        else hipLaunchKernelGGL(stream_kernel<2>, dim3(s_wgs), dim3(256), 0, st, src, s_bytes / 16 / s_wgs, sink);
    };
    auto G = [&](hipStream_t st, int l) { hipLaunchKernelGGL(small_kernel, dim3(256), dim3(256), 0, st, gbuf + (size_t)(l & 7) * (g_bytes / 16), g_kb * 1024 / 16, sink, out); };
    auto timeit = [&](const char* name, auto fn) {
        for (int w = 0; w < 2; ++w) fn();
        hipDeviceSynchronize();
        hipEventRecord(e0, a);
        for (int r = 0; r < 5; ++r) fn();
        hipEventRecord(eb, b); hipStreamWaitEvent(a, eb, 0);
        hipEventRecord(e1, a); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-46s %8.1f us per layer-equivalent\n", name, ms * 1e3 / 5 / NL);
    };
    printf("streaming kernel: %d WGs, %zu MB per launch, launch bounds %d waves/SIMD; small kernel: 256 WGs x %zu KB, %d per layer\n", s_wgs, s_mb, wps, g_kb, G_PER);
    timeit("S alone (stream A)", [&] { for (int l = 0; l < NL; ++l) S(a, l); });
    timeit("G chain alone (stream A)", [&] { for (int l = 0; l < NL; ++l) for (int g = 0; g < G_PER; ++g) G(a, l * G_PER + g); });
    timeit("S then G chain, one stream (serial)", [&] { for (int l = 0; l < NL; ++l) { S(a, l); for (int g = 0; g < G_PER; ++g) G(a, l * G_PER + g); } });
    timeit("S on stream A || G chain on stream B", [&] { hipEventRecord(eb, a); hipStreamWaitEvent(b, eb, 0); for (int l = 0; l < NL; ++l) { S(a, l); for (int g = 0; g < G_PER; ++g) G(b, l * G_PER + g); } });
    return 0;
}
