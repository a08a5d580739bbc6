// Diagnostic: in-kernel phase timeline of the decode GEMMs in their real launch order (qkv -> o -> gate/up -> down over 30
// layers of distinct weights, back to back on one stream).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// -fhip-fp32-correctly-rounded-divide-sqrt -std=c++17 -DT3_GEMM_CLK tools/gemm_clk.hip -o tools/gemm_clk
// Reports the stamps of gemm2_kernel (the schedule the engine runs at <= 64-80 rows): entry, A staged, first weights, last MFMA, barrier, store.
#include "../chatterbox-vllm2_amd/csrc/t3_gemm.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace t3;
// Experiment: a helper kernel on a SECOND stream pulls the next GEMM's weights into the XCD-local L2 slices while the current GEMM
// runs (unit u = what consumer workgroup x = u streams, consumed on XCD u % 8).  argv[4] = 1 enables it.
__global__ __launch_bounds__(256) void l2_warm_kernel(const uint4* base, int units, int unit_kib, uint32_t* sink) {
    const int p = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    const int wp = (gridDim.x >> 3) * 4, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lines = (units >> 3) * unit_kib;
    uint32_t x = 0;
    for (int l = (blockIdx.x >> 3) * 4 + wave; l < lines; l += wp) {
        const int unit = (l / unit_kib) * 8 + p, off = l % unit_kib;
        const uint4 v = base[((size_t)unit * unit_kib + off) * 64 + lane];
        x ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (x == 0x9e3779b9u && lane == 77) *sink = x;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64, NL = 30;
    const int qkv_mt_arg = argc > 2 ? atoi(argv[2]) : 0, gu_mt_arg = argc > 3 ? atoi(argv[3]) : 0;     // 0 = the engine's choice
    std::vector<uint16_t> rnd(1 << 20);
    uint32_t st = 12345;
    for (auto& v : rnd) { st = st * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((st >> 9) & 0x3ff)) ^ ((st >> 3) & 0x8000); }   // small-magnitude bf16
    auto dev_fill = [&](uint16_t** p, size_t n) { if (hipMalloc((void**)p, n * 2) != hipSuccess) return false; for (size_t o = 0; o < n; o += rnd.size()) (void)hipMemcpy(*p + o, rnd.data(), std::min(rnd.size(), n - o) * 2, hipMemcpyHostToDevice); return true; };
    std::vector<uint16_t*> wq(NL), wo(NL), wg(NL), wd(NL);
    for (int l = 0; l < NL; ++l) if (!dev_fill(&wq[l], (size_t)QKV * D) || !dev_fill(&wo[l], (size_t)D * D) || !dev_fill(&wg[l], (size_t)2 * F * D) || !dev_fill(&wd[l], (size_t)D * F)) return 1;
    uint16_t *h, *qkv, *att, *act, *ln;
    if (!dev_fill(&h, (size_t)M * D) || !dev_fill(&qkv, (size_t)M * QKV) || !dev_fill(&att, (size_t)M * D) || !dev_fill(&act, (size_t)M * F) || !dev_fill(&ln, D)) return 1;
    hipStream_t s; CK(hipStreamCreate(&s));
    const int warm = argc > 4 ? atoi(argv[4]) : 0;
    const int mask = argc > 6 ? atoi(argv[6]) : 15;            // which of qkv (1) / o (2) / gate-up (4) / down (8) the chain launches
    const int nlw = argc > 5 ? atoi(argv[5]) : NL;            // distinct weight sets the chain cycles through (1: 33.5 MB, resident in the Infinity Cache)
    hipStream_t s2; CK(hipStreamCreate(&s2));
    hipEvent_t ev[8]; for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    uint32_t* sink; CK(hipMalloc((void**)&sink, 4));
    int evi = 0;
    auto warm_next = [&](const uint16_t* w, int units, int unit_kib) {      // runs beside the GEMM launched right after this call
        if (!warm) return;
        hipEvent_t e = ev[evi++ & 7];
        (void)hipEventRecord(e, s); (void)hipStreamWaitEvent(s2, e, 0);
        hipLaunchKernelGGL(l2_warm_kernel, dim3(warm), dim3(256), 0, s2, (const uint4*)w, units, unit_kib, sink);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto chain = [&]() {
        for (int li = 0; li < NL; ++li) {
            const int l = li % nlw;
            warm_next(wo[l], 64, 32);                          // beside qkv: o_proj weights (64 units of 32 KiB)
            if (mask & 1) { GemmArgs a{h, (const uint4*)wq[l], M, D, QKV, qkv, QKV, 4, 1, nullptr}; if (launch_gemm(a, EPI_BF16, qkv_mt_arg ? qkv_mt_arg : choose_mt(M, QKV / 16, 4, true), s) != hipSuccess) return false; }
            warm_next(wg[l], 128, 128);                        // beside o_proj: gate/up weights (128 units of 4 packed tiles)
            if (mask & 2) { GemmArgs a{att, (const uint4*)wo[l], M, D, D, h, D, 16, 0, nullptr}; if (launch_gemm(a, EPI_RESID, choose_mt(M, D / 16, 16, false), s) != hipSuccess) return false; }
            warm_next(wd[l], 64, 128);                         // beside gate/up: down_proj weights (64 units of 128 KiB)
            if (mask & 4) { GemmArgs a{h, (const uint4*)wg[l], M, D, F, act, F, 4, 1, nullptr}; if (launch_gemm(a, EPI_SILU, gu_mt_arg ? gu_mt_arg : choose_mt(M, F / 16, 4, true), s) != hipSuccess) return false; }
            warm_next(wq[(l + 1) % NL], 48, 128);              // beside down_proj: the next layer's qkv weights (48 units of 4 tiles)
            if (mask & 8) { GemmArgs a{act, (const uint4*)wd[l], M, F, D, h, D, 16, 0, nullptr}; if (launch_gemm(a, EPI_RESID, choose_mt(M, D / 16, 16, false), s) != hipSuccess) return false; }
        }
        return true;
    };
    for (int w = 0; w < 3; ++w) if (!chain()) return 1;
    CK(hipStreamSynchronize(s));
    { static unsigned long long z[8][2048][5]; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_clk), z, sizeof(z))); }
    { static unsigned long long z2[4][2048][6]; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm2_clk), z2, sizeof(z2))); }
    CK(hipEventRecord(e0, s));
    for (int w = 0; w < 5; ++w) if (!chain()) return 1;
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("M=%d, %d distinct layers of weights: %d launches, %.2f us per launch, %.1f us per layer (GEMM mask %d, eager launches, stamped build)\n", M, nlw, 5 * NL * __builtin_popcount(mask), ms * 1e3 / (5 * NL * __builtin_popcount(mask)), ms * 1e3 / (5 * NL), mask);
    // gemm2_kernel (the current decode schedule): the LAST launch of each class left its stamps
    static unsigned long long clk[4][2048][6];
    CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_gemm2_clk), sizeof(clk)));
    const char* names[4] = {"qkv (NORM, 4 waves)", "gate/up (NORM, SiLU, 4 waves)", "o (16 waves, +resid)", "down (16 waves, +resid)"};
    for (int k = 0; k < 4; ++k) {
        int n = 0; while (n < 2048 && clk[k][n][0]) ++n;       // workgroups that left stamps (grid of the last launch of this class)
        if (!n) { printf("%s: no gemm2_kernel stamps (another schedule ran at this row count)\n", names[k]); continue; }
        unsigned long long t0min = ~0ull, t5max = 0;
        for (int i = 0; i < n; ++i) { t0min = std::min(t0min, clk[k][i][0]); t5max = std::max(t5max, clk[k][i][5]); }
        auto stat = [&](auto f, const char* what) {
            std::vector<double> v; for (int i = 0; i < n; ++i) v.push_back(f(clk[k][i]) / 100.0);
            std::sort(v.begin(), v.end());
            printf("    %-44s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us\n", what, v[0], v[n / 2], v[n * 9 / 10], v[n - 1]);
        };
        printf("%s: %d workgroups, first entry -> last exit %.2f us\n", names[k], n, (t5max - t0min) / 100.0);
        stat([&](unsigned long long* c) { return (double)(c[0] - t0min); }, "entry after first workgroup");
        stat([&](unsigned long long* c) { return (double)(c[1] - c[0]); }, "entry -> A rows landed + staged in LDS");
        stat([&](unsigned long long* c) { return (double)(c[2] - c[1]); }, "-> first weight k-block landed");
        stat([&](unsigned long long* c) { return (double)(c[3] - c[2]); }, "-> last MFMA issued (rest of the K slice)");
        stat([&](unsigned long long* c) { return (double)(c[4] - c[3]); }, "-> partials exchanged (LDS write + barrier)");
        stat([&](unsigned long long* c) { return (double)(c[5] - c[4]); }, "-> fold + epilogue + store issued");
        stat([&](unsigned long long* c) { return (double)(c[5] - c[0]); }, "workgroup lifetime");
        stat([&](unsigned long long* c) { return (double)(c[5] - t0min); }, "exit time since the kernel's first entry");
    }
    return 0;
}
