// Diagnostic: the persistent layer-chain kernel (csrc/chain_kernel.hip) against the four-launch chain it replaces, on 30 layers of
// distinct random weights: (1) bit-exact comparison of h / act / qkv after every layer, (2) time per layer of both forms, each
// replayed from a hipGraph.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/chain_proto.hip -o tools/chain_proto
// args: rows (default 64), repeats (default 20)
#include "../chatterbox-vllm2_amd/csrc/t3_gemm.hip"
#include "chain_kernel.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace t3;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64, REP = argc > 2 ? atoi(argv[2]) : 20, NL = 30;
    std::vector<uint16_t> rnd(1 << 20);
    uint32_t st = 12345;
    for (auto& v : rnd) { st = st * 1664525u + 1013904223u; v = (uint16_t)((0x3c00 + ((st >> 9) & 0x3ff)) ^ ((st >> 3) & 0x8000)); }   // +-[1, 2) mantissas, sign random
    auto dev_fill = [&](uint16_t** p, size_t n, size_t shift, int exp_shift) {
        if (hipMalloc((void**)p, n * 2) != hipSuccess) return false;
        std::vector<uint16_t> t(rnd.begin() + shift, rnd.end());
        for (auto& v : t) v = (uint16_t)(v - (exp_shift << 7));           // scale by 2^-exp_shift
        for (size_t o = 0; o < n; o += t.size()) (void)hipMemcpy(*p + o, t.data(), std::min(t.size(), n - o) * 2, hipMemcpyHostToDevice);
        return true;
    };
    std::vector<uint16_t*> wq(NL + 1), wo(NL), wg(NL), wd(NL);
    for (int l = 0; l < NL; ++l) if (!dev_fill(&wq[l], (size_t)QKV * D, 7 * l, 6) || !dev_fill(&wo[l], (size_t)D * D, 11 * l + 1, 6) || !dev_fill(&wg[l], (size_t)2 * F * D, 13 * l + 2, 6) || !dev_fill(&wd[l], (size_t)D * F, 17 * l + 3, 7)) return 1;
    wq[NL] = wq[0];
    uint16_t* att;
    if (!dev_fill(&att, (size_t)M * D, 99, 2)) return 1;
    struct Bufs { uint16_t *h, *act, *qkv; } A, B;
    for (Bufs* b : {&A, &B}) if (!dev_fill(&b->h, (size_t)M * D, 31, 0) || !dev_fill(&b->act, (size_t)M * F, 37, 0) || !dev_fill(&b->qkv, (size_t)M * QKV, 41, 0)) return 1;
    float* rstd; CK(hipMalloc((void**)&rstd, (size_t)M * 4));
    unsigned *flags, *err; CK(hipMalloc((void**)&flags, CHAIN_WGS * 4 + 64)); CK(hipMemset(flags, 0, CHAIN_WGS * 4 + 64)); err = flags + CHAIN_WGS;
    hipStream_t s; CK(hipStreamCreate(&s));
    CK(prepare_kernels());
    auto one_launch = [&](int l, Bufs& b, int which) {
        if (which == 0) { GemmArgs a{att, (const uint4*)wo[l], M, D, D, b.h, D, 16, 0, nullptr}; return launch_gemm(a, EPI_RESID, choose_mt(M, D / 16, 16, false), s) == hipSuccess; }
        if (which == 1) { GemmArgs a{b.h, (const uint4*)wg[l], M, D, F, b.act, F, 4, 1, nullptr, 0, rstd}; return launch_gemm(a, EPI_SILU, choose_mt(M, F / 16, 4, true), s) == hipSuccess; }
        if (which == 2) { GemmArgs a{b.act, (const uint4*)wd[l], M, F, D, b.h, D, 16, 0, nullptr}; return launch_gemm(a, EPI_RESID, choose_mt(M, D / 16, 16, false), s) == hipSuccess; }
        GemmArgs a{b.h, (const uint4*)wq[l + 1], M, D, QKV, b.qkv, QKV, 4, 1, nullptr, 0, rstd}; return launch_gemm(a, EPI_BF16, choose_mt(M, QKV / 16, 4, true), s) == hipSuccess;
    };
    auto layer_launches = [&](int l, Bufs& b) {
        { GemmArgs a{att, (const uint4*)wo[l], M, D, D, b.h, D, 16, 0, nullptr}; if (launch_gemm(a, EPI_RESID, choose_mt(M, D / 16, 16, false), s) != hipSuccess) return false; }
        { GemmArgs a{b.h, (const uint4*)wg[l], M, D, F, b.act, F, 4, 1, nullptr, 0, rstd}; if (launch_gemm(a, EPI_SILU, choose_mt(M, F / 16, 4, true), s) != hipSuccess) return false; }
        { GemmArgs a{b.act, (const uint4*)wd[l], M, F, D, b.h, D, 16, 0, nullptr}; if (launch_gemm(a, EPI_RESID, choose_mt(M, D / 16, 16, false), s) != hipSuccess) return false; }
        { GemmArgs a{b.h, (const uint4*)wq[l + 1], M, D, QKV, b.qkv, QKV, 4, 1, nullptr, 0, rstd}; if (launch_gemm(a, EPI_BF16, choose_mt(M, QKV / 16, 4, true), s) != hipSuccess) return false; }
        return true;
    };
    auto layer_chain = [&](int l, Bufs& b, int phases) {
        ChainArgs c{(const uint4*)wo[l], (const uint4*)wg[l], (const uint4*)wd[l], (const uint4*)wq[l + 1], att, b.h, b.act, b.qkv, M, phases, flags, err};
        return launch_chain(c, s) == hipSuccess;
    };
    // ---- (1) parity, layer by layer (h keeps evolving: every layer starts from the previous layer's output in both forms)
    std::vector<uint16_t> ha((size_t)M * D), hb(ha.size()), qa((size_t)M * QKV), qb(qa.size()), aa((size_t)M * F), ab(aa.size());
    size_t bad = 0;
    for (int l = 0; l < NL; ++l) {
        if (!layer_launches(l, A) || !(M <= 64 ? layer_chain(l, B, 15) : layer_launches(l, B))) { printf("launch failed\n"); return 1; }
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(ha.data(), A.h, ha.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), B.h, hb.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(qa.data(), A.qkv, qa.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(qb.data(), B.qkv, qb.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(aa.data(), A.act, aa.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ab.data(), B.act, ab.size() * 2, hipMemcpyDeviceToHost));
        size_t dh = 0, dq = 0, da = 0;
        for (size_t i = 0; i < ha.size(); ++i) dh += ha[i] != hb[i];
        for (size_t i = 0; i < qa.size(); ++i) dq += qa[i] != qb[i];
        for (size_t i = 0; i < aa.size(); ++i) da += aa[i] != ab[i];
        if (dh | dq | da) { printf("layer %d: %zu h, %zu act, %zu qkv elements differ (first h diff at %zu)\n", l, dh, da, dq, (size_t)(std::mismatch(ha.begin(), ha.end(), hb.begin()).first - ha.begin())); bad += dh + dq + da; }
        if (l == 0 || l == NL - 1) printf("layer %d: h[0..3] = %04x %04x %04x %04x  qkv[0..1] = %04x %04x\n", l, ha[0], ha[1], ha[2], ha[3], qa[0], qa[1]);
    }
    unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("parity over %d layers at M=%d: %s (barrier timeout flag %u)\n", NL, M, bad ? "MISMATCH" : "bit-exact", herr);
    // ---- (2) timing: 30 layers per graph
    auto time_graph = [&](auto&& body, const char* what) -> int {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        bool ok = true; for (int l = 0; l < NL; ++l) ok = ok && body(l);
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
        printf("%-52s %7.2f us per layer\n", what, ms * 1e3 / (REP * NL));
        return 0;
    };
    if (time_graph([&](int l) { return layer_launches(l, A); }, "four launches per layer (graph replay)")) return 1;
    if (time_graph([&](int l) { return one_launch(l, A, 0); }, "  launch: o only")) return 1;
    if (time_graph([&](int l) { return one_launch(l, A, 1); }, "  launch: gate/up only")) return 1;
    if (time_graph([&](int l) { return one_launch(l, A, 2); }, "  launch: down only")) return 1;
    if (time_graph([&](int l) { return one_launch(l, A, 3); }, "  launch: qkv only")) return 1;
    if (getenv("CHAIN_SKIP") || M > 64) return 0;
    if (time_graph([&](int l) { return layer_chain(l, B, 15); }, "persistent chain, one launch per layer")) return 1;
    if (time_graph([&](int l) { return layer_chain(l, B, 1); }, "  chain: o only")) return 1;
    if (time_graph([&](int l) { return layer_chain(l, B, 2); }, "  chain: gate/up only")) return 1;
    if (time_graph([&](int l) { return layer_chain(l, B, 4); }, "  chain: down only")) return 1;
    if (time_graph([&](int l) { return layer_chain(l, B, 8); }, "  chain: qkv only")) return 1;
    if (time_graph([&](int l) { return layer_chain(l, B, 3); }, "  chain: o + gate/up (one barrier)")) return 1;
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("barrier timeout flag after timing: %u\n", herr);
    return 0;
}
