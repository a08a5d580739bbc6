// Embedding, sampler and hand-off kernels of the T3 decode engine for gfx950 (MI355X, CDNA4); GEMMs: t3_gemm.hip, attention: t3_attention.hip.  wave = 64 lanes.
//
// Every floating-point rounding point and summation order in this file is part of the numerics
// contract written down in DESIGN.md ("Numerics contract"); the contract is what makes the emitted
// token ids bit-identical to the CPU restatement of the reference semantics.  Compile with
// -ffp-contract=off: every fused multiply-add below is an explicit __builtin_fmaf.
//
// Reference semantics being implemented (paths relative to the reference repo):
//   embed ............ src/chatterbox_vllm/models/t3/t3.py:440-486, 542-561
//   CFG logits ....... t3.py:650-673
//   sampler .......... vllm SamplingParams as configured at src/chatterbox_vllm/tts.py:455-464
#include "t3_kernels.h"
#include "t3_device.h"

#include <math.h>
#include <string.h>

#include <utility>

namespace t3 {
void arm_launch_events(hipEvent_t start, hipEvent_t stop) { g_arm_start = start; g_arm_stop = stop; }
bool launch_events_armed() { return g_arm_start != nullptr; }

// ------------------------------------------------------------------------------------------------
// Embedding rows (t3.py:440-486 decode, 542-561 prefill): one wave per row.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 add_bf8(const uint4& a, const uint4& b) {
    float fa[8], fb[8]; unpack8(a, fa); unpack8(b, fb);
    uint4 o;
    o.x = pack2(fa[0] + fb[0], fa[1] + fb[1]); o.y = pack2(fa[2] + fb[2], fa[3] + fb[3]);
    o.z = pack2(fa[4] + fb[4], fa[5] + fb[5]); o.w = pack2(fa[6] + fb[6], fa[7] + fb[7]);
    return o;
}
__global__ __launch_bounds__(256) void embed_kernel(EmbedArgs a) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.meta_vec; i += gridDim.x * 256) a.dev_meta[i] = a.host_meta[i];
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int* rec = (a.host_rowrec ? a.host_rowrec : a.rowrec) + (size_t)row * a.row_stride;
    int4 d = make_int4(rec[2], rec[3], rec[4], 0);
    if (d.x == EMB_SPEECH_PREV) { d.x = EMB_SPEECH; d.y = a.prev_tok[d.y]; }   // token sampled by the step still in flight when this one was scheduled
    uint4* out = reinterpret_cast<uint4*>(a.h + (size_t)row * D);
    if (d.x == EMB_ZERO) {
        out[lane] = make_uint4(0, 0, 0, 0); out[64 + lane] = make_uint4(0, 0, 0, 0);
    } else if (d.x == EMB_COND) {
        const float4* src = reinterpret_cast<const float4*>(a.cond + ((size_t)d.y * T3_COND_ROWS + d.z) * D);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const float4 p = src[(half * 64 + lane) * 2], q = src[(half * 64 + lane) * 2 + 1];
            uint4 o; o.x = pack2(p.x, p.y); o.y = pack2(p.z, p.w); o.z = pack2(q.x, q.y); o.w = pack2(q.z, q.w);
            out[half * 64 + lane] = o;
        }
    } else {
        const uint16_t* e = (d.x == EMB_TEXT) ? a.text_emb : a.speech_emb;
        const uint16_t* p = (d.x == EMB_TEXT) ? a.text_pos : a.speech_pos;
        const uint4* er = reinterpret_cast<const uint4*>(e + (size_t)d.y * D);
        const uint4* pr = reinterpret_cast<const uint4*>(p + (size_t)d.z * D);
        out[lane] = add_bf8(er[lane], pr[lane]);
        out[64 + lane] = add_bf8(er[64 + lane], pr[64 + lane]);
    }
}
hipError_t launch_embed(const EmbedArgs& a, hipStream_t s) {
    if (a.rows <= 0) return hipSuccess;
    launch_k(embed_kernel, dim3((a.rows + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}
// ------------------------------------------------------------------------------------------------
// CFG (t3.py:662, bf16 tensor arithmetic) + sampler.  One workgroup per sampled utterance.
// Probability mass lives on integer weights w = floor(exp(l - max) * 2^32), so every sum is exact and
// order-free; thresholds come from binary searches on the weight value (DESIGN.md "Sampler").
// ------------------------------------------------------------------------------------------------
// Waves per utterance.  With 4 waves (round 1-2) every SIMD of the CU ran ONE wave through ~3 000 dependent vector instructions (40 logits
// of CFG / penalties / correctly rounded division per thread, then 33 exps): 10.8 + 4.2 us of a 29 us launch (tools/sampler_clk.py).
// With 16 waves a thread owns 9 entries and four waves share a SIMD's issue slots.
#ifndef T3_SAMPLER_WAVES
#define T3_SAMPLER_WAVES 16
#endif
constexpr int SWV = T3_SAMPLER_WAVES, STH = SWV * 64;
constexpr int SPT = (8194 + STH - 1) / STH;     // elements per thread in contiguous ownership: STH * SPT >= 8194 (33 at 4 waves, 9 at 16)
constexpr int SLOTS = STH * SPT;
constexpr int SSCR = 32 + 5 * SWV + 4;          // reduction words behind the logits: [32] the two-barrier helpers' | five single-use arrays of [SWV] | [4] a radix round's result

// Wave-level reductions and scans on the DPP cross-lane path (a ds_bpermute butterfly costs ~0.2 us per level here;
// the sampler is a chain of such reductions).  All operands are integers (or a float max), so order is immaterial.
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {     // bound_ctrl: lanes without a source read 0
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, true);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_SHR = 0x110;
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
    v += dpp_u64<DPP_XOR1>(v); v += dpp_u64<DPP_XOR2>(v); v += dpp_u64<DPP_HALF_MIRROR>(v); v += dpp_u64<DPP_MIRROR>(v);   // row of 16
    return readlane_u64(v, 0) + readlane_u64(v, 16) + readlane_u64(v, 32) + readlane_u64(v, 48);
}
__device__ __forceinline__ unsigned long long max_u64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    v = max_u64(v, dpp_u64<DPP_XOR1>(v)); v = max_u64(v, dpp_u64<DPP_XOR2>(v));
    v = max_u64(v, dpp_u64<DPP_HALF_MIRROR>(v)); v = max_u64(v, dpp_u64<DPP_MIRROR>(v));
    return max_u64(max_u64(readlane_u64(v, 0), readlane_u64(v, 16)), max_u64(readlane_u64(v, 32), readlane_u64(v, 48)));
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ unsigned long long wave_scan_u64(unsigned long long v, int lane) {
    v += dpp_u64<DPP_SHR + 1>(v); v += dpp_u64<DPP_SHR + 2>(v); v += dpp_u64<DPP_SHR + 4>(v); v += dpp_u64<DPP_SHR + 8>(v);  // within rows of 16
    const unsigned long long t0 = readlane_u64(v, 15), t1 = readlane_u64(v, 31), t2 = readlane_u64(v, 47);
    return v + (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
}
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long* scr) {
    v = wave_sum_u64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < SWV; ++w) t += scr[w];
    return t;
}
__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long v, unsigned long long* scr) {
    v = wave_max_u64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < SWV; ++w) t = max_u64(t, scr[w]);
    return t;
}
// max over finite-or-minus-infinity floats through the order-preserving integer key
__device__ __forceinline__ float block_max_f32(float v, unsigned long long* scr) {
    uint32_t b = __float_as_uint(v + 0.0f);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    b = (uint32_t)block_max_u64(b, scr);
    b = (b & 0x80000000u) ? (b & 0x7fffffffu) : ~b;
    return __uint_as_float(b);
}
// inclusive prefix sum over the threads of the workgroup; *total = sum over all
__device__ __forceinline__ unsigned long long block_scan_u64(unsigned long long v, unsigned long long* scr, unsigned long long* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_scan_u64(v, lane);
    __syncthreads();
    if (lane == 63) scr[wave] = v;
    __syncthreads();
    unsigned long long tot = 0, before = 0;
#pragma unroll
    for (int w = 0; w < SWV; ++w) { const unsigned long long sw_ = scr[w]; before += w < wave ? sw_ : 0; tot += sw_; }
    *total = tot;
    return v + before;
}

// Combining the SWV (<= 16) per-wave words of a reduction: lane w reads wave w's word and the row of 16 lanes folds them on the DPP path
// (~20 instructions; a loop over the words was ~6 per word, and the sampler's sixteen waves are bound by instruction issue -- four waves
// per SIMD, every instruction of "all waves" costs 16 cycles).  Call with all lanes active.
__device__ __forceinline__ unsigned long long waves_sum_u64(const unsigned long long* arr, int lane) {
    unsigned long long v = lane < SWV ? arr[lane] : 0;
    v += dpp_u64<DPP_XOR1>(v); v += dpp_u64<DPP_XOR2>(v); v += dpp_u64<DPP_HALF_MIRROR>(v); v += dpp_u64<DPP_MIRROR>(v);
    return readlane_u64(v, 0);
}
__device__ __forceinline__ unsigned long long waves_max_u64(const unsigned long long* arr, int lane) {
    unsigned long long v = lane < SWV ? arr[lane] : 0;
    v = max_u64(v, dpp_u64<DPP_XOR1>(v)); v = max_u64(v, dpp_u64<DPP_XOR2>(v));
    v = max_u64(v, dpp_u64<DPP_HALF_MIRROR>(v)); v = max_u64(v, dpp_u64<DPP_MIRROR>(v));
    return readlane_u64(v, 0);
}
// arr[w] = inclusive total of wave w's own scan: returns the sum over the waves before `wave`; *total = the sum over all
__device__ __forceinline__ unsigned long long waves_before_u64(const unsigned long long* arr, int lane, int wave, unsigned long long* total) {
    const unsigned long long own = lane < SWV ? arr[lane] : 0;
    unsigned long long v = own;
    v += dpp_u64<DPP_SHR + 1>(v); v += dpp_u64<DPP_SHR + 2>(v); v += dpp_u64<DPP_SHR + 4>(v); v += dpp_u64<DPP_SHR + 8>(v);      // inclusive within the row of 16
    *total = readlane_u64(v, 15);
    return readlane_u64(v - own, wave);
}

__device__ __forceinline__ uint32_t t3_cvt_u32_trunc(float f) {      // v_cvt_u32_f32: truncation toward zero, NaN -> 0 (callers pass 0 <= f < 2^32 or NaN)
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}
__device__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#ifdef T3_SAMPLER_CLK      // diagnostic build only: phase timestamps (100 MHz ticks) overwrite dbg[0..] at exit
#define T3_CLK(i) clk[i] = wall_clock64()
#else
#define T3_CLK(i)
#endif
__global__ __launch_bounds__(STH) void sampler_kernel(SampleArgs a) {
#ifdef T3_SAMPLER_CLK
    unsigned long long clk[12] = {0};
#endif
    T3_CLK(0);
    // LDS: [SLOTS] x (fp32) | [SSCR] reduction words | 3 x [2][HC][64] histogram copies (mass + count, smallest, largest).  Every reduction of the sampled path has words of its
    // own, so it is write -> ONE barrier -> read (round 3 shared one scratch array: two barriers per reduction, 35 barriers per launch
    // for sixteen waves; tools/sampler_clk.py, profiles/NOTES.md).
    extern __shared__ __attribute__((aligned(16))) unsigned long long sw[];
    float* xs = reinterpret_cast<float*>(sw);
    unsigned long long* scr = sw + SLOTS / 2;              // legacy two-barrier helpers (greedy, min-p / top-k, degenerate top-p): [SWV] + [24..25]
    unsigned long long* s_mx = scr + 32;                   // [SWV] max of the scaled logits
    unsigned long long* s_wm = s_mx + SWV;                 // [SWV] largest weight
    unsigned long long* s_ws = s_wm + SWV;                 // [SWV] sum of the weights
    unsigned long long* s_ti = s_ws + SWV;                 // [SWV] ties scan
    unsigned long long* s_dr = s_ti + SWV;                 // [SWV] draw scan
    unsigned long long* psum = scr + SSCR;
    const int tid = threadIdx.x, u = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = a.sel[u].x;
    const uint32_t step = (uint32_t)a.sel[u].y;
    const T3Sampling sp = a.sp[slot];
    const uint16_t* lc = a.logits + (size_t)(2 * u) * a.ldl;
    const uint16_t* lu = lc + a.ldl;
    uint16_t* counts = a.counts + (size_t)slot * VPAD;

    // the drawn id leaves the kernel through whichever thread knows it
    auto finish = [&](int token) {
        token = token < 0 ? 0 : (token >= V ? V - 1 : token);       // counts[] / speech_emb[] are indexed with it
        a.out_tok[u] = token;
        if (a.out_tok_host) a.out_tok_host[u] = token;
        if (a.hist && (int)step < a.hist_cap) a.hist[(size_t)slot * a.hist_cap + step] = token;      // the utterance's ids stay on the device
        const uint16_t cnt = counts[token];
        if (cnt < 65535) counts[token] = cnt + 1;
#ifdef T3_SAMPLER_CLK
        T3_CLK(10);
        if (a.dbg) for (int i = 0; i < 11; ++i) a.dbg[(size_t)slot * V + i] = (float)(clk[i] - clk[0]);
#endif
    };

    // ---- phase A: CFG, penalties.  16-byte loads of 8 logits / counts, all issued before the first use
    // (a scalar loop here is a chain of dependent memory latencies, not bandwidth).
    const bool greedy = sp.temperature < 1e-5f;
    float mx = -INFINITY;
    unsigned long long best = 0;
    // 8 logits per 16-byte load: KF full rounds of STH vectors; the LV vectors left over (8 208 = 1 024 x 8 + 16: two of them hold the ids
    // 8 192 and 8 193) go element by element to the lanes of the last wave -- as a second, nearly empty round of the unrolled vector
    // loop they doubled wave 0's phase A (3 us) while fifteen waves waited at the barrier.
    constexpr int NVEC = VPAD / 8, KF = NVEC / STH, LV = NVEC - KF * STH, TAIL0 = KF * STH * 8;
    static_assert(VPAD % 8 == 0 && SLOTS >= VPAD && SLOTS % 2 == 0 && LV * 8 <= 64, "sampler vector layout");
    uint4 c4[KF], u4[KF], n4[KF];
#pragma unroll
    for (int k = 0; k < KF; ++k) {
        const int vi = tid + STH * k;
        c4[k] = reinterpret_cast<const uint4*>(lc)[vi];
        u4[k] = reinterpret_cast<const uint4*>(lu)[vi];
        n4[k] = reinterpret_cast<const uint4*>(counts)[vi];
    }
    const bool tail = LV > 0 && wave == SWV - 1 && lane < LV * 8;
    uint16_t tc = 0, tu = 0, tn = 0;
    if (tail) { tc = lc[TAIL0 + lane]; tu = lu[TAIL0 + lane]; tn = counts[TAIL0 + lane]; }
    for (int v = VPAD + tid; v < SLOTS; v += STH) xs[v] = -INFINITY;
    constexpr int HC = 4, HSZ = HC * 64;                 // histogram copies (see the radix rounds), words per histogram
    unsigned long long* hmin = psum + 2 * HSZ;           // [2][HC][64] smallest / largest entry of a bin (rounds behind the octave round)
    unsigned long long* hmax = psum + 4 * HSZ;
    for (int j = tid; j < 2 * HSZ; j += STH) { psum[j] = 0; hmin[j] = ~0ull; hmax[j] = 0; }
    T3_CLK(1);
    auto element = [&](uint16_t cb, uint16_t ub, uint32_t cnt, int v) -> float {
        float x = -INFINITY;
        if (v < V) {
            const float c = bf2f(cb), un = bf2f(ub);
            const float d = rbf(c - un);
            const float e = rbf(a.cfg * d);
            x = rbf(c + e);
            if (a.dbg) a.dbg[(size_t)slot * V + v] = x;
            if (cnt > 0) {
                if (sp.repetition_penalty != 1.0f) x = (x > 0.0f) ? x / sp.repetition_penalty : x * sp.repetition_penalty;
                x = x - sp.frequency_penalty * (float)cnt;
                x = x - sp.presence_penalty;
            }
            if (greedy) {
                const float xz = x + 0.0f;                       // -0 -> +0 so that the key order equals '>' on floats
                uint32_t b = __float_as_uint(xz);
                b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
                const unsigned long long key = ((unsigned long long)b << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)v);
                best = key > best ? key : best;
            } else {
                x = x / sp.temperature;
                mx = fmaxf(mx, x);
            }
        }
        return x;
    };
#pragma unroll
    for (int k = 0; k < KF; ++k) {
        const int vi = tid + STH * k;
        const uint32_t cw[4] = {c4[k].x, c4[k].y, c4[k].z, c4[k].w}, uw[4] = {u4[k].x, u4[k].y, u4[k].z, u4[k].w},
                       nw[4] = {n4[k].x, n4[k].y, n4[k].z, n4[k].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int sh = (j & 1) * 16;
            xs[vi * 8 + j] = element((uint16_t)(cw[j >> 1] >> sh), (uint16_t)(uw[j >> 1] >> sh), (nw[j >> 1] >> sh) & 0xFFFFu, vi * 8 + j);
        }
    }
    if (tail) xs[TAIL0 + lane] = element(tc, tu, tn, TAIL0 + lane);
    T3_CLK(2);
    if (greedy) {
        best = block_max_u64(best, scr);
        if (tid == 0) {
            const int token = (int)(0xFFFFFFFFu - (uint32_t)best);
            if (a.dbg_keep) a.dbg_keep[(size_t)slot * V + (token < 0 ? 0 : (token >= V ? V - 1 : token))] = 1;      // (zeroed by the caller)
            finish(token);
        }
        return;
    }
    // ---- max of the scaled logits: finite-or-minus-infinity floats through the order-preserving integer key
    {
        uint32_t b = __float_as_uint(mx + 0.0f);
        b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
        const unsigned long long wm_ = wave_max_u64(b);
        if (lane == 0) s_mx[wave] = wm_;
        __syncthreads();                                   // also: every x of phase A is in LDS
        b = (uint32_t)waves_max_u64(s_mx, lane);
        b = (b & 0x80000000u) ? (b & 0x7fffffffu) : ~b;
        mx = __uint_as_float(b);
    }
    // ---- weights.  From here on a thread owns SPT consecutive vocabulary entries and keeps them in registers (it reads their x from LDS:
    // the weights themselves never go through LDS).  floor(e x 2^32): e < 1 for all but the largest logits, and then the product is below
    // 2^32 and one v_cvt_u32_f32 (truncation) is the floor; the general 64-bit conversion only where e >= 1.
    const int v0 = tid * SPT;
    unsigned long long wr[SPT];
    unsigned long long wmax = 0, W = 0;
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
        const float e = t3_expf(xs[v0 + i] - mx);
        unsigned long long w = 0;
        if (v0 + i < V) {
            if (e >= 1.0f) w = (unsigned long long)(e * 4294967296.0f);
            else w = (unsigned long long)t3_cvt_u32_trunc(e * 4294967296.0f);
        }
        wr[i] = w;
        wmax = max_u64(wmax, w);
        W += w;
    }
    T3_CLK(3);
    {
        const unsigned long long m_ = wave_max_u64(wmax), s_ = wave_sum_u64(W);
        if (lane == 0) { s_wm[wave] = m_; s_ws[wave] = s_; }
        __syncthreads();
        wmax = waves_max_u64(s_wm, lane); W = waves_sum_u64(s_ws, lane);
    }
    bool resum = false;
    if (sp.min_p > 0.0f) {
        const double thr = (double)sp.min_p * (double)wmax;
#pragma unroll
        for (int i = 0; i < SPT; ++i) if ((double)wr[i] < thr) wr[i] = 0;
        resum = true;
    }
    if (sp.top_k > 0 && sp.top_k < V) {
        unsigned long long lo = 0, hi = wmax;
        while (lo < hi) {
            const unsigned long long mid = lo + (hi - lo + 1) / 2;
            unsigned long long cnt = 0;
#pragma unroll
            for (int i = 0; i < SPT; ++i) cnt += (wr[i] >= mid) ? 1 : 0;
            cnt = block_sum_u64(cnt, scr);
            if (cnt >= (unsigned long long)sp.top_k) lo = mid; else hi = mid - 1;
        }
#pragma unroll
        for (int i = 0; i < SPT; ++i) if (wr[i] < lo) wr[i] = 0;
        resum = true;
    }
    T3_CLK(4);
    if (sp.top_p < 1.0f) {
        if (resum) {
            W = 0;
#pragma unroll
            for (int i = 0; i < SPT; ++i) W += wr[i];
            W = block_sum_u64(W, scr);
        }
        const unsigned long long Tm = (unsigned long long)((1.0 - (double)sp.top_p) * (double)W);
        // lo = max{mid : sum of weights < mid is <= Tm} is the weight value at which the ascending cumulative
        // mass first exceeds Tm.  Radix descent on the value: one round over the octaves, then 6 bits per
        // round; per round an LDS histogram of masses (exact integer sums, so order-free) and a 64-bin scan.
        T3_CLK(5);
        // The histogram is kept in HC copies, one per lane group (lane & (HC - 1)): the masses of a round fall into a handful of the 64
        // bins (a softmax's weights share few octaves), and 64-bit LDS atomics of one wave instruction onto the same bin serialise.  The sums
        // are exact integers, so the copies add up to the same histogram in any order.
        // A bin word carries the mass in its low 48 bits (all of them together stay below 8 194 x 2^32 < 2^46) and the NUMBER of entries above:
        // when the descent ends on a single value, the chosen bin's count is the number of ties and the mass below it is `base`.
        // The sixteen waves are bound by instruction issue, so a round spends as few instructions as it can: a thread keeps a bit mask of its
        // entries that still match the prefix (a wave without any skips the round's loop), and ONE wave scans the 64 bins and leaves the
        // chosen bin, the mass below it and its count in LDS for the others (wave 1 clears the other histogram meanwhile): two barriers
        // per round.  A round is a chain of latencies (LDS atomics, barrier, four-copy reads, 64-bit scan, barrier: ~0.85 us,
        // tools/sampler_clk.py), and a 33-bit weight is seven rounds deep -- but the entries of the chosen bin are usually ONE value long
        // before that (a handful of entries per bin after two rounds; and logits that went through bf16 tie in droves): the rounds behind
        // the octave round also keep every bin's smallest and largest entry, and the descent ends as soon as the chosen bin's two agree.
        constexpr unsigned long long M48 = (1ull << 48) - 1, ONE = 1ull << 48;
        unsigned long long* s_res = s_dr + SWV;          // [4]: chosen bin + 1 (0: none), mass below it inside the round, its word
        unsigned long long lo = wmax, base = 0, prefix = 0, below = 0, ties = 0;
        bool counted = false;                            // below / ties known from the descent
        int nb = -1;                                     // bits of the value still undetermined; -1: octave round
        uint32_t cand = 0;
#pragma unroll
        for (int i = 0; i < SPT; ++i) cand |= wr[i] ? (1u << i) : 0u;
        for (int rd = 0;; ++rd) {
            unsigned long long* hist = psum + (rd & 1) * HSZ;
            unsigned long long* myhist = hist + (lane & (HC - 1)) * 64;
            unsigned long long* mymin = hmin + (rd & 1) * HSZ + (lane & (HC - 1)) * 64;
            unsigned long long* mymax = hmax + (rd & 1) * HSZ + (lane & (HC - 1)) * 64;
            if (cand) {
                if (nb < 0) {
#pragma unroll
                    for (int i = 0; i < SPT; ++i) if (wr[i]) atomicAdd(&myhist[64 - __clzll((long long)wr[i])], wr[i] + ONE);
                } else {
                    const int shift = nb > 6 ? nb - 6 : 0;
                    const unsigned long long msk = (1ull << (nb - shift)) - 1;
#pragma unroll
                    for (int i = 0; i < SPT; ++i) {
                        if (!((cand >> i) & 1u)) continue;
                        if ((wr[i] >> nb) == prefix) {
                            const int bin = (int)((wr[i] >> shift) & msk);
                            atomicAdd(&myhist[bin], wr[i] + ONE);
                            atomicMin(&mymin[bin], wr[i]); atomicMax(&mymax[bin], wr[i]);
                        } else cand &= ~(1u << i);
                    }
                }
            }
            __syncthreads();
            if (wave == 0) {
                const unsigned long long own = (hist[lane] + hist[64 + lane]) + (hist[128 + lane] + hist[192 + lane]);
                const unsigned long long c = wave_scan_u64(own, lane);      // inclusive scan over the 64 bins (masses and counts side by side)
                const unsigned long long over = __ballot(base + (c & M48) > Tm);
                const int b = over ? __ffsll((long long)over) - 1 : 0;
                const unsigned long long add = readlane_u64(c - own, b) & M48, word = readlane_u64(own, b);
                const unsigned long long* mn = hmin + (rd & 1) * HSZ; const unsigned long long* mxh = hmax + (rd & 1) * HSZ;
                unsigned long long lowest = mn[lane], highest = mxh[lane];
#pragma unroll
                for (int k = 1; k < HC; ++k) { const unsigned long long a_ = mn[64 * k + lane], b_ = mxh[64 * k + lane]; lowest = a_ < lowest ? a_ : lowest; highest = max_u64(highest, b_); }
                const unsigned long long one = (lowest == highest) ? lowest : 0ull;      // octave round: no entries here (~0 against 0)
                const unsigned long long same = readlane_u64(one, b);
                if (lane == 0) { s_res[0] = over ? (unsigned long long)(b + 1) : 0ull; s_res[1] = add; s_res[2] = word; s_res[3] = same; }
            } else if (wave == 1 % SWV) {
                for (int j = lane; j < HSZ; j += 64) {       // the other set: last read in round rd - 1 (before that round's second barrier)
                    const int o = ((rd + 1) & 1) * HSZ + j;
                    psum[o] = 0; hmin[o] = ~0ull; hmax[o] = 0;
                }
            }
            __syncthreads();
            const int b1 = (int)s_res[0];
            if (!b1) break;                              // only in the octave round, when Tm == W: lo = wmax
            const int b = b1 - 1;
            base += s_res[1];
            if (s_res[3]) { lo = s_res[3]; below = base; ties = s_res[2] >> 48; counted = true; break; }      // the chosen bin holds one value
            if (nb < 0) { prefix = 1; nb = b - 1; }
            else { const int shift = nb > 6 ? nb - 6 : 0; prefix = (prefix << (nb - shift)) | (unsigned long long)b; nb = shift; }
            if (nb == 0) { lo = prefix; below = base; ties = s_res[2] >> 48; counted = true; break; }
        }
        T3_CLK(6);
        if (lo > 0) {
            if (!counted) {                              // degenerate threshold (Tm == W): lo = wmax, counted the long way
                unsigned long long my_ties = 0;
#pragma unroll
                for (int i = 0; i < SPT; ++i) { below += (wr[i] < lo) ? wr[i] : 0; my_ties += (wr[i] == lo) ? 1 : 0; }
                below = block_sum_u64(below, scr);
                ties = block_sum_u64(my_ties, scr);
            }
            unsigned long long r = (Tm - below) / lo;
            if (lo == wmax && r > ties - 1) r = ties - 1;
            if (r > ties) r = ties;
            // the r ties of highest index are dropped.  r == 0 / r == ties need no ranks.
            unsigned long long above = 0;                // ties owned by higher threads
            if (r > 0 && r < ties) {
                unsigned long long my_ties = 0;
#pragma unroll
                for (int i = 0; i < SPT; ++i) my_ties += (wr[i] == lo) ? 1 : 0;
                const unsigned long long incl_w = wave_scan_u64(my_ties, lane);
                if (lane == 63) s_ti[wave] = incl_w;
                __syncthreads();
                unsigned long long tot;
                above = ties - (incl_w + waves_before_u64(s_ti, lane, wave, &tot));
            } else if (r == 0) above = ties;             // nobody's rank from the top is below 0
#pragma unroll
            for (int i = SPT - 1; i >= 0; --i) {
                if (wr[i] < lo) wr[i] = 0;
                else if (wr[i] == lo) { if (above < r) wr[i] = 0; ++above; }
            }
        }
    }
    T3_CLK(7);
    if (a.dbg_keep) {                                   // parity hook: the support of the draw (what the masks left)
#pragma unroll
        for (int i = 0; i < SPT; ++i) if (v0 + i < V) a.dbg_keep[(size_t)slot * V + v0 + i] = wr[i] != 0;
    }
    // ---- draw
    unsigned long long mine = 0;
#pragma unroll
    for (int i = 0; i < SPT; ++i) mine += wr[i];
    unsigned long long Wk = 0, excl;
    {
        const unsigned long long incl_w = wave_scan_u64(mine, lane);
        if (lane == 63) s_dr[wave] = incl_w;
        __syncthreads();
        excl = incl_w + waves_before_u64(s_dr, lane, wave, &Wk) - mine;
    }
    T3_CLK(8);
    uint32_t rnd[4];
    philox4x32_10(step, (uint32_t)sp.uid, (uint32_t)(sp.uid >> 32), 0u, (uint32_t)sp.seed, (uint32_t)(sp.seed >> 32), rnd);
    const unsigned long long uu = ((unsigned long long)rnd[1] << 32) | rnd[0];
    const unsigned long long target = __umul64hi(uu, Wk);
    T3_CLK(9);
    // Exactly one thread owns the target (the threads' [excl, excl + mine) tile [0, Wk)) and writes the id out itself.  No thread owns it
    // when there is no mass at all (NaN logits: a checkpoint or conditioning with NaN / Inf makes every weight 0): the draw then falls back
    // to the stop id.
    if (Wk == 0) {
        if (tid == 0) finish((sp.stop_token >= 0 && sp.stop_token < V) ? sp.stop_token : 0);
    } else if (mine > 0 && target >= excl && target < excl + mine) {
        unsigned long long cum = excl; int found = -1;      // first entry whose running mass passes the target
#pragma unroll
        for (int i = 0; i < SPT; ++i) { cum += wr[i]; if (found < 0 && cum > target) found = v0 + i; }
        finish(found);
    }
}
hipError_t prepare_kernels() {
    gemm_refresh_switches();
    static bool done[MAX_DEVICES] = {};
    if (done[cur_device()]) return hipSuccess;
    const size_t lds = (size_t)(SLOTS / 2 + SSCR + 6 * 256) * sizeof(unsigned long long);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sampler_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = prepare_gemm2();
    if (e == hipSuccess) done[cur_device()] = true;
    return e;
}
hipError_t launch_sampler(const SampleArgs& a, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    const size_t lds = (size_t)(SLOTS / 2 + SSCR + 6 * 256) * sizeof(unsigned long long);
    hipError_t e = prepare_kernels();
    if (e != hipSuccess) return e;
    launch_k(sampler_kernel, dim3(a.n), dim3(STH), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// f4 hand-off: post-filter + range filter + padding of one utterance per workgroup (see t3_kernels.h).  The analyzer's rules in
// closed form: it forces EOS at the first index i >= 2 with ids[i] == ids[i-1] == ids[i-2], or at index completed_at + 9 where
// completed_at = max(1, 2 (text_token_count - 3)) is the first frame whose estimated text position (frame / 2, capped at
// text_token_count - 1) reaches text_token_count - 3; the kept tokens are the prefix before that index (t3_clean_tokens).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void handoff_kernel(const HandoffItem* items, int flags, int* out, int ld, int* lens) {
    __shared__ int s_min[4], s_cnt[4], s_base;
    const HandoffItem it = items[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int first_rep = 0x7fffffff;
    for (int i = 2 + tid; i < it.n; i += 256)
        if (it.src[i] == it.src[i - 1] && it.src[i] == it.src[i - 2]) { first_rep = i; break; }       // ascending per thread: its first hit is its minimum
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) first_rep = min(first_rep, __shfl_xor(first_rep, off));
    if (lane == 0) s_min[wave] = first_rep;
    if (tid == 0) s_base = 0;
    __syncthreads();
    first_rep = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
    const int completed_at = max(1, 2 * (it.text_token_count - 3));
    const int cut = min(min(first_rep, completed_at + 9), it.n);
    int* row = out + (size_t)blockIdx.x * ld;
    for (int base = 0; base < cut; base += 256) {
        const int i = base + tid;
        const int v = i < cut ? it.src[i] : -1;
        const bool keep = i < cut && (!(flags & 1) || (v >= 0 && v < 6561));
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_cnt[wave] = __popcll(m);
        __syncthreads();
        int pre = s_base;
        for (int w = 0; w < wave; ++w) pre += s_cnt[w];
        if (keep && pre + before < ld) row[pre + before] = v;
        __syncthreads();
        if (tid == 0) s_base += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
    }
    const int kept = min(s_base, ld);
    for (int i = kept + tid; i < ld; i += 256) row[i] = 0;
    if (tid == 0) lens[blockIdx.x] = kept;
}
hipError_t launch_handoff(const HandoffItem* items, int n_utt, int flags, int* out, int ld, int* lens, hipStream_t s) {
    if (n_utt <= 0) return hipSuccess;
    hipLaunchKernelGGL(handoff_kernel, dim3(n_utt), dim3(256), 0, s, items, flags, out, ld, lens);
    return hipGetLastError();
}

__global__ void expf_kernel(const float* x, float* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = t3_expf(x[i]);
}
hipError_t launch_expf(const float* x, float* y, int n, hipStream_t s) {
    hipLaunchKernelGGL(expf_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n);
    return hipGetLastError();
}

// llama3-scaled RoPE table (t3-model/config.json:21-28): inv_freq fp32-valued, angle and cos/sin in
// double, rounded to fp32 then bf16 (the cache is held in the model dtype, as vLLM and HF do).
void rope_tables(int max_pos, float* cos_t, float* sin_t) {
    double inv[32];
    const double theta = 500000.0, factor = 8.0, lo = 1.0, hi = 4.0, old = 8192.0, pi = 3.14159265358979323846;
    for (int i = 0; i < 32; ++i) {
        double f = pow(theta, -(2.0 * i) / 64.0);
        const double wl = 2.0 * pi / f;
        if (wl > old / lo) f = f / factor;
        else if (wl >= old / hi) { const double sm = (old / wl - lo) / (hi - lo); f = (1.0 - sm) * f / factor + sm * f; }
        inv[i] = (double)(float)f;
    }
    auto rb = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; float r; memcpy(&r, &u, 4); return r; };
    for (int p = 0; p < max_pos; ++p)
        for (int i = 0; i < 32; ++i) {
            const double ang = (double)p * inv[i];
            cos_t[(size_t)p * 32 + i] = rb((float)cos(ang));
            sin_t[(size_t)p * 32 + i] = rb((float)sin(ang));
        }
}

}  // namespace t3
