// Diagnostic: which XCD does block b of a launch land on?  Records HW_REG_XCC_ID per workgroup for a sequence of launches of different
// grid sizes on one stream (eager and as one hipGraph), prints the XCD of block 0 and whether block b sits on (xcd0 + b) % 8.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/xcd_probe.hip -o tools/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out, int spin) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) out[lin] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID, 4 bits
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    const int grids[][2] = {{256, 1}, {256, 1}, {1024, 1}, {256, 1}, {192, 1}, {16, 64}, {64, 4}, {128, 2}, {64, 4}, {129, 1}, {256, 1}, {32, 1}, {256, 1}, {16, 4}, {256, 1}, {3, 1}, {256, 1}, {256, 1}};
    const int n = sizeof(grids) / sizeof(grids[0]);
    int* d; CK(hipMalloc((void**)&d, n * 4096 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipMemset(d, 0xff, n * 4096 * 4));
        hipGraph_t g; hipGraphExec_t ge;
        if (mode == 1) CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(probe, dim3(grids[i][0], grids[i][1]), dim3(256), 0, s, d + i * 4096, 20);
        if (mode == 1) { CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); CK(hipGraphLaunch(ge, s)); CK(hipGraphLaunch(ge, s)); }
        CK(hipStreamSynchronize(s));
        std::vector<int> h(n * 4096); CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
        printf("%s\n", mode ? "one hipGraph (second replay)" : "eager launches");
        for (int i = 0; i < n; ++i) {
            const int nb = grids[i][0] * grids[i][1], x0 = h[i * 4096];
            int ok = 0; for (int b = 0; b < nb; ++b) ok += h[i * 4096 + b] == (x0 + b) % 8;
            printf("  launch %2d grid (%4d,%3d): block 0 on XCD %d, blocks 1..7 on", i, grids[i][0], grids[i][1], x0);
            for (int b = 1; b < 8 && b < nb; ++b) printf(" %d", h[i * 4096 + b]);
            printf("; %d of %d blocks on (xcd0 + b) %% 8\n", ok, nb);
        }
    }
    return 0;
}
