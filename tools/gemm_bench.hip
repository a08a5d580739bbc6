// Diagnostic: each of a layer's four GEMM launches alone, as the engine dispatches it at a given row count, 30 layers of distinct weights
// per graph replay (the harness of tools/chain_proto.hip without the persistent chain).  Switches come from the environment
// (T3_GEMM_PIPE, T3_GEMM_PIPE_QKV_MIN_ROWS, T3_GEMM_SPLIT_EPI, T3_GEMM_LOOP*_*, T3_PGEMM_MIN_ROWS ...).
//   make -C tools gemm_bench && tools/gemm_bench <rows> [repeats]
#include "../chatterbox-vllm2_amd/csrc/t3_gemm.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace t3;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64, REP = argc > 2 ? atoi(argv[2]) : 20, NL = 30;
    std::vector<uint16_t> rnd(1 << 20);
    uint32_t st = 12345;
    for (auto& v : rnd) { st = st * 1664525u + 1013904223u; v = (uint16_t)((0x3c00 + ((st >> 9) & 0x3ff)) ^ ((st >> 3) & 0x8000)); }
    auto dev_fill = [&](uint16_t** p, size_t n, size_t shift, int exp_shift) {
        if (hipMalloc((void**)p, n * 2) != hipSuccess) return false;
        std::vector<uint16_t> t(rnd.begin() + shift, rnd.end());
        for (auto& v : t) v = (uint16_t)(v - (exp_shift << 7));
        for (size_t o = 0; o < n; o += t.size()) (void)hipMemcpy(*p + o, t.data(), std::min(t.size(), n - o) * 2, hipMemcpyHostToDevice);
        return true;
    };
    std::vector<uint16_t*> wq(NL), wo(NL), wg(NL), wd(NL);
    for (int l = 0; l < NL; ++l)
        if (!dev_fill(&wq[l], (size_t)QKV * D, 7 * l, 6) || !dev_fill(&wo[l], (size_t)D * D, 11 * l + 1, 6) || !dev_fill(&wg[l], (size_t)2 * F * D, 13 * l + 2, 6) || !dev_fill(&wd[l], (size_t)D * F, 17 * l + 3, 6)) return 1;
    uint16_t *att, *h, *act, *qkv;
    if (!dev_fill(&att, (size_t)M * D, 99, 2) || !dev_fill(&h, (size_t)M * D, 31, 0) || !dev_fill(&act, (size_t)M * F, 37, 0) || !dev_fill(&qkv, (size_t)M * QKV, 41, 0)) return 1;
    float* rstd; CK(hipMalloc((void**)&rstd, (size_t)M * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    gemm_refresh_switches();
    CK(prepare_gemm2());
    auto one = [&](int l, int which) {
        if (which == 0) { GemmArgs a{att, (const uint4*)wo[l], M, D, D, h, D, 16, 0, nullptr}; return launch_gemm(a, EPI_RESID, choose_mt(M, D / 16, 16, false), s) == hipSuccess; }
        if (which == 1) { GemmArgs a{h, (const uint4*)wg[l], M, D, F, act, F, 4, 1, nullptr, 0, rstd}; return launch_gemm(a, EPI_SILU, choose_mt(M, F / 16, 4, true), s) == hipSuccess; }
        if (which == 2) { GemmArgs a{act, (const uint4*)wd[l], M, F, D, h, D, 16, 0, nullptr}; return launch_gemm(a, EPI_RESID, choose_mt(M, D / 16, 16, false), s) == hipSuccess; }
        GemmArgs a{h, (const uint4*)wq[l], M, D, QKV, qkv, QKV, 4, 1, nullptr, 0, rstd}; return launch_gemm(a, EPI_BF16, choose_mt(M, QKV / 16, 4, true), s) == hipSuccess;
    };
    const char* names[5] = {"o", "gate/up", "down", "qkv", "all four"};
    for (int which = 0; which < 5; ++which) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        bool ok = true;
        for (int l = 0; l < NL; ++l) { if (which < 4) ok = ok && one(l, which); else for (int w : {3, 0, 1, 2}) ok = ok && one(l, w); }
        CK(hipStreamEndCapture(s, &g));
        if (!ok) { printf("capture failed\n"); return 1; }
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int w = 0; w < REP; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rows %4d  %-9s %7.2f us per %s\n", M, names[which], ms * 1e3 / (REP * NL), which < 4 ? "launch" : "layer");
    }
    return 0;
}
