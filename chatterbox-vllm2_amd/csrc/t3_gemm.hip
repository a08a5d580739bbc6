// GEMM kernels of the T3 decode engine for gfx950 (MI355X, CDNA4): the decode schedules (gemm2_kernel, gemm2_loop_kernel), the
// prefill schedule (pgemm_kernel) and their launchers.  wave = 64 lanes.
//
// Every floating-point rounding point and summation order in this file is part of the numerics contract written down in
// DESIGN.md ("Numerics contract").  Compile with -ffp-contract=off: every fused multiply-add is an explicit __builtin_fmaf.
// Reference semantics: the Llama block of src/chatterbox_vllm/models/t3/t3.py:696-713 -> vllm LlamaModel
// (hyper-parameters t3-model/config.json:1-33), logits t3.py:650-673.
#include "t3_kernels.h"
#include "t3_device.h"

#include <math.h>
#include <string.h>

#include <utility>

namespace t3 {
// ------------------------------------------------------------------------------------------------
// Skinny GEMM  y[m][n] = sum_k x[m][k] * W[n][k]   (x, W bf16, fp32 accumulation on the matrix cores)
//
// Contract order (DESIGN.md "GEMM"): K is cut into NW contiguous segments (one per wave of the workgroup,
// NW = 4 or 16); a wave folds its segment with a chain of v_mfma_f32_16x16x32_bf16 in ascending k (each
// instruction folds 4 blocks of 8 consecutive k into the fp32 accumulator; its exact arithmetic was
// identified on-device and is restated in the checker); four consecutive segments give a group sum
// G = ((s0 + s1) + s2) + s3 in fp32; with 16 segments (o_proj, down_proj) the result is ((G0 + G1) + G2) + G3.
//
// NORM form (qkv, gate/up, speech head; K = 1024, NW = 4): the RMSNorm that precedes the projection is folded in -- its
// weight into the packed matrix at load time (W' = bf16(W * w_ln), fold_norm_weight()), its row statistic sum(h^2) onto the
// matrix cores (a wave multiplies its A fragments with themselves and reads the diagonal: the same MFMA chain per segment as
// the GEMM itself), and rstd = 1/sqrt(ss/1024 + eps) into the epilogue.  The activations reach the MFMAs untouched.
// Statistic order: per wave (segment) one chain from +0 in ascending k; the four segment sums fold ((S0 + S1) + S2) + S3.
//
// One workgroup = NW waves = NT n-tiles of 16 columns x MT m-tiles of 16 rows.  Weights are packed so that a
// wave's weight load is one contiguous 1 KiB (pack_weight).  (The round-1 schedule -- both operands through a symmetric
// register ring, fragment-shaped activation loads -- is kept for the record in tools/legacy/gemm_kernel_r1.hip.)
// ------------------------------------------------------------------------------------------------
#ifdef T3_GEMM_CLK      // diagnostic build only (tools/gemm_clk.hip): per-workgroup phase stamps, 100 MHz ticks
// gemm2_kernel (the decode schedule): class 0 qkv / head, 1 gate/up, 2 o, 3 down; stamps: 0 entry, 1 A rows landed and staged
// in LDS, 2 first weight k-block landed, 3 last MFMA issued, 4 partials exchanged (barrier passed), 5 outputs stored
__device__ unsigned long long g_gemm2_clk[4][2048][6];
extern "C" int t3_debug_gemm2_clk(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm2_clk), sizeof(g_gemm2_clk)); }
#define T3_G2STAMP(i) do { if (threadIdx.x == 0) g_gemm2_clk[NORM ? (EPI == EPI_SILU ? 1 : 0) : (KBS == 2 ? 2 : 3)][(blockIdx.y * gridDim.x + blockIdx.x) & 2047][i] = wall_clock64(); } while (0)
#else
#define T3_G2STAMP(i)
#endif
// ------------------------------------------------------------------------------------------------
// gemm2_kernel: the decode schedule of the NORM forms (NW = 4, K = 1024, MT <= 2) and of the 16-segment forms at one
// m-tile per workgroup (o_proj / down_proj).  What matters in how the bytes move:
//   * A operand: a wave reads its K slice of its 16 rows as FULL row segments (KBS * 64 contiguous bytes per row, whole
//     128-byte lines) into a wave-private, XOR-swizzled LDS image and takes its MFMA fragments from there with ds_read_b128.
//     Fragment-shaped global loads (16 rows x 64 B per instruction, half lines: the round-1 kernel) cost the texture path twice
//     the cycles per byte: at 64 rows the activations were 3.8 us of a 29 us layer (tools/chain_proto.hip, -DT3_GEMM_XDUMMY).
//   * every weight tile of the wave's K slice is requested up front (KBS * NT KiB in flight per wave, no refill logic), behind
//     the A loads, so the A image is in LDS while the weights are still in flight.
// The partial sums of the cross-wave fold reuse the wave's own A image (dead after the K loop), so LDS = the A images only.
// NORM: rstd from the MFMA diagonal (see the header of this section); the weights carry the norm weight already.
// ------------------------------------------------------------------------------------------------
// Loads whose ISSUE ORDER matters (gemm2_kernel): inline asm, so hipcc neither reorders nor counts them.  Every wait below is
// hand-counted (vmcnt retires in issue order), and names the registers it releases as read-write operands, so no consumer can be
// scheduled above it (cdna_hip_programming.md 5.7, form (ii)).
typedef unsigned int uint4_v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gload16(uint4_v& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void gload16_nt(uint4_v& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(d) : "v"(p) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <typename Fn, int... Is>
__device__ __forceinline__ void static_for(Fn&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
__device__ __forceinline__ void landed(uint4_v& d) { asm volatile("" : "+v"(d)); }          // d is defined from here on
__device__ __forceinline__ bf16x8 as_frag4(const uint4_v& v) { union { uint4_v u; bf16x8 f; } c; c.u = v; return c.f; }
template <int KBS>
__device__ __forceinline__ unsigned a_img_off(int row, int ch) {     // byte offset of 16-byte chunk ch of row `row` in a wave's A image
    if constexpr (KBS >= 4) return (unsigned)(row * (KBS * 64) + ((ch ^ (row & 15)) << 4));        // >= 256-byte rows: one row per bank row
    else return (unsigned)(row * (KBS * 64) + ((ch ^ ((row >> 1) & (KBS * 4 - 1))) << 4));          // 128-byte rows: two rows per bank row
}

// PrefetchArgs (t3_kernels.h): this workgroup's share of a later launch's weight lines.  cls = this workgroup's XCD (its linear id % 8),
// idx / n = its index among / the number of (workgroup, wave) slots of that XCD that take part; dump = LDS byte address (wave-uniform) of
// 256 bytes nobody reads.  The loads count in vmcnt like any other and retire in issue order: issue them BEHIND every load the wave
// still waits for; nothing waits for them (s_endpgm does).
__device__ __forceinline__ void prefetch_next_weights(const PrefetchArgs& pf, int cls, int idx, int n, int lane, unsigned dump) {
    if (!pf.base) return;
    const int groups = pf.n_tiles / pf.group, groups_cls = groups >> 3;      // callers: groups is a multiple of 8
    const int glines = pf.max_lines > 0 && pf.max_lines < pf.group * pf.tile_lines ? pf.max_lines : pf.group * pf.tile_lines;
    const int total = groups_cls * glines, cnt = (total + n - 1) / n;
    for (int j = 0; j * 64 < cnt && j < 32; ++j) {
        const int k = lane + 64 * j, l = idx * cnt + k;
        if (k < cnt && l < total) {
            const int g = cls + 8 * (l / glines), rem = l % glines;
            const unsigned char* p = pf.base + ((size_t)g * pf.group * pf.tile_lines + rem) * 128;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(p), "s"(dump) : "memory");
        }
    }
}

// EW: extra waves that sleep at the barrier and then share the epilogue.  The fold + epilogue of the 4-wave forms is a dependent chain of
// ~300 vector instructions per thread (eight LDS reads, rstd, four SiLU-mul outputs with correctly rounded divisions) issued by ONE wave per
// SIMD: 1.32 us of gate/up's 6.1 us, 0.52 of qkv's 3.6 (stamps, profiles/r03_gemm_clk_m64.txt).  With four more waves every thread
// finishes two columns instead of four and two waves share each SIMD's issue slots.
#ifndef T3_GEMM2_EW
#define T3_GEMM2_EW 4
#endif
template <int NW> constexpr int gemm2_ew() { return NW == 4 ? T3_GEMM2_EW : 0; }
// AV: how many of the KBS A-row instructions per m-tile a wave issues.  An instruction covers RPI consecutive rows, and the texture
// path charges a padded row like a real one: a decode step of 1-4 utterances has 2-8 rows in its 16-row tile, and at 2 rows the
// padding was as many bytes through the CU's load path as the workgroup's weights.  The launcher picks the smallest AV whose rows
// cover M (one m-tile, one m-group); the image rows beyond are zero (row m of the accumulator depends on image row m alone, and
// rows >= M are never stored).  A template parameter, not a branch: conditional asm loads make hipcc build the register tuples by copies.
// EWV: the epilogue waves of this instantiation (default: four for the 4-wave forms).  EWV = 0 halves the workgroup to 4 waves, so that TWO
// workgroups fit a CU at up to 256 registers: the speech head at 64 rows is 129 tile groups x 2 row groups = 258 workgroups, and with one
// workgroup per CU the last two ran a second, almost empty round (12.3 us for 16.8 MB against gate/up's 7.7).
// k-blocks of A fragments in flight from LDS ahead of the MFMAs (gemm2_kernel's K loop): every one of them where the registers are there
// (one m-tile: 4 registers per k-block), two k-blocks ahead for the two-m-tile forms (up to 228 registers already)
template <int MT, int NT, int KBS> constexpr int gemm2_afd() { return MT == 1 ? KBS : (MT * NT >= 8 ? 2 : 4) < KBS ? (MT * NT >= 8 ? 2 : 4) : KBS; }
template <int MT, int NT, int EPI, int NW, int KBS, bool NORM, int AV = KBS, int EWV = gemm2_ew<NW>()>
__global__ __launch_bounds__((NW + EWV) * 64, (NW == 4 && EWV == 0 && MT * NT >= 8) ? 2 : 1) void gemm2_kernel(GemmArgs a) {
    T3_G2STAMP(0);
    if (a.gx_real && (int)blockIdx.x >= a.gx_real) return;       // padding of the grid (launch_gemm2_av)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];      // [NW waves][MT][16 rows][KBS * 64 B] | NORM: float [NW][MT*16]
    static_assert(NT <= KBS && (KBS == 8 || KBS == 2) && (NW == 4 || NW == 16) && AV >= 1 && AV <= KBS && (AV == KBS || MT == 1), "gemm2 shapes");
    constexpr int LPR = KBS * 4, RPI = 64 / LPR;                // lanes (= 16-byte chunks) per row slice, rows per wave instruction
    constexpr int ABYTES = MT * KBS * 1024, TILES = MT * NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int KB = a.K >> 5, kb0 = wave * KBS;
    unsigned char* aimg = lds2 + (size_t)wave * ABYTES;
    float* rowsum = reinterpret_cast<float*>(lds2 + (size_t)NW * ABYTES);

    // EPI_RESID (16-wave form, one output per thread): the residual operand is requested first (a compiler-counted load: it must be older than the asm loads)
    uint16_t hres = 0;
    if constexpr (EPI == EPI_RESID) {
        const int r = (tid >> 6) & 3, l2 = tid & 63;
        const int m = blockIdx.y * 16 + 4 * (l2 >> 4) + r, n = blockIdx.x * 16 + (l2 & 15);
        if (tid < 256 && m < a.M && n < a.N) hres = reinterpret_cast<const uint16_t*>(a.out)[(size_t)m * a.ldo + n];
    }
    constexpr int EW = EWV;
    if (EW == 0 || wave < NW) {          // the compute waves; the EW epilogue waves go straight to the barrier
    // ---- A: full row segments of this wave's K slice
    uint4_v ar[MT][KBS];
    const int rsub = lane / LPR, ch = lane % LPR;
    int mrow[MT][KBS];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < KBS; ++t) {
            const int m = (blockIdx.y * MT + i) * 16 + t * RPI + rsub;
            mrow[i][t] = m < a.M ? m : a.M - 1;        // padded rows re-read the last row; their outputs are dropped
        }
    if (a.row_index) {                                 // ONE branch around all the gather loads (a select per element would serialise them)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int t = 0; t < KBS; ++t) mrow[i][t] = a.row_index[mrow[i][t]];
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < KBS; ++t) {
            if constexpr (AV < KBS) { if (t >= AV) { ar[i][t] = (uint4_v){0u, 0u, 0u, 0u}; continue; } }
            gload16(ar[i][t], a.X + (size_t)mrow[i][t] * a.K + kb0 * 32 + ch * 8);
        }
    // ---- W: every tile of this wave's K slice, behind the A loads and before the first wait (left to itself, hipcc sinks these
    // loads below the staging block to save registers, i.e. behind a full L2 round trip)
    uint4_v wr[KBS][NT];
#pragma unroll
    for (int kb = 0; kb < KBS; ++kb)
#pragma unroll
        for (int t = 0; t < NT; ++t) gload16_nt(wr[kb][t], a.Wp + ((size_t)(blockIdx.x * NT + t) * KB + kb0 + kb) * 64 + lane);
    // ---- stage A (wave-private: a wave's DS operations execute in order, no barrier).  The KBS * NT weight loads are younger.
    wait_vmcnt<KBS * NT>();
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < AV; ++t) landed(ar[i][t]);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < KBS; ++t)
            *reinterpret_cast<uint4_v*>(aimg + i * (KBS * 1024) + a_img_off<KBS>(t * RPI + rsub, ch)) = ar[i][t];
    asm volatile("" ::: "memory");
    T3_G2STAMP(1);
    f32x4 acc[MT][NT], ss[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        ss[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // The A fragments of the next AFD k-blocks are requested from LDS ahead of the MFMAs that use them.  Read per k-block right before its
    // MFMAs (rounds 2-3), a k-block cost an LDS round trip (~130-200 cycles) that only its own few MFMAs could cover: the 16-wave forms (ONE
    // MFMA per k-block) and qkv spent 0.5-0.8 us between "rows staged" and "last MFMA" (profiles/r03_gemm_clk_m64.txt) for 8-32 MFMAs.
    constexpr int AFD = gemm2_afd<MT, NT, KBS>();
    uint4 afr[AFD][MT];
    auto read_af = [&](int slot, int kb) {
#pragma unroll
        for (int i = 0; i < MT; ++i) afr[slot][i] = *reinterpret_cast<const uint4*>(aimg + i * (KBS * 1024) + a_img_off<KBS>(c, 4 * kb + q));
    };
    static_for([&](auto dc) { read_af(decltype(dc)::value, decltype(dc)::value); }, std::make_integer_sequence<int, AFD>{});
    static_for([&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        wait_vmcnt<(KBS - 1 - kb) * NT>();            // this k-block's NT weight tiles have landed ((KBS - 1 - kb) * NT younger loads may still fly)
#pragma unroll
        for (int t = 0; t < NT; ++t) landed(wr[kb][t]);
        if constexpr (kb == 0) T3_G2STAMP(2);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const uint4 af = afr[kb % AFD][i];
            if constexpr (NORM) ss[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af), as_frag(af), ss[i], 0, 0, 0);   // diagonal = sum of squares
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af), as_frag4(wr[kb][t]), acc[i][t], 0, 0, 0);
        }
        if constexpr (kb + AFD < KBS) read_af(kb % AFD, kb + AFD);
    }, std::make_integer_sequence<int, KBS>{});
    T3_G2STAMP(3);
    // ---- partials over the wave's own (now dead) A image: [tile][r][lane]
    asm volatile("" ::: "memory");
    float* redw = reinterpret_cast<float*>(aimg);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) redw[((i * NT + t) * 4 + r) * 64 + lane] = acc[i][t][r];
    if constexpr (NORM) {
        // D[row = 4 q + r][col = c]: the diagonal element of row c sits in lane group q = c / 4, register c % 4
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int r = c & 3;
            const float d = r == 0 ? ss[i][0] : r == 1 ? ss[i][1] : r == 2 ? ss[i][2] : ss[i][3];
            if ((c >> 2) == q) rowsum[wave * (MT * 16) + i * 16 + c] = d;
        }
    }
    }                                     // compute waves
    __syncthreads();
    T3_G2STAMP(4);
    if constexpr (EW > 0) {
        // PrefetchArgs: every operand of the workgroup has landed and the memory system idles until the stores: the epilogue waves ask for
        // their share of the next launch's weights before they start folding (dump corner: the 256 bytes behind everything else in LDS)
        if (wave >= NW && ((gridDim.x * gridDim.y) & 7) == 0) {
            const int lin = blockIdx.y * gridDim.x + blockIdx.x;
            const unsigned dump = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(lds2 + (size_t)NW * ABYTES + (NORM ? NW * MT * 16 * sizeof(float) : 0));
            prefetch_next_weights(a.pf, lin & 7, (lin >> 3) * EW + (wave - NW), ((gridDim.x * gridDim.y) >> 3) * EW, lane, dump);
        }
    }
    auto part = [&](int w) { return reinterpret_cast<const float*>(lds2 + (size_t)w * ABYTES); };

    if constexpr (NW == 4) {
        // CW outputs (one row, CW columns) per thread and step: four (16-byte LDS reads, 8-byte stores) with the compute waves alone,
        // two with the epilogue waves
        constexpr int CW = EW ? 2 : 4, PPT = 16 / CW;                       // pieces per 16-column tile row
        constexpr int NTO = (EPI == EPI_SILU) ? NT / 2 : NT, PIECES = MT * NTO * 16 * PPT, TH = (NW + EW) * 64, PIT = (PIECES + TH - 1) / TH;
#pragma unroll
        for (int k = 0; k < PIT; ++k) {
            const int p = tid + k * TH;
            if (p >= PIECES) continue;
            const int ito = p / (16 * PPT), r16 = (p / PPT) & 15, qq = p % PPT;
            const int i = ito / NTO, to = ito % NTO;
            const int m = (blockIdx.y * MT + i) * 16 + r16;
            if (m >= a.M) continue;
            const int g = r16 >> 2, r = r16 & 3;
            float v[EPI == EPI_SILU ? 2 : 1][CW];
#pragma unroll
            for (int u = 0; u < (EPI == EPI_SILU ? 2 : 1); ++u) {
                const int it = i * NT + (EPI == EPI_SILU ? 2 * to + u : to);
                const int o = (it * 4 + r) * 64 + 16 * g + CW * qq;
                if constexpr (CW == 4) {
                    const float4 s0 = *reinterpret_cast<const float4*>(part(0) + o), s1 = *reinterpret_cast<const float4*>(part(1) + o),
                                 s2 = *reinterpret_cast<const float4*>(part(2) + o), s3 = *reinterpret_cast<const float4*>(part(3) + o);
                    v[u][0] = ((s0.x + s1.x) + s2.x) + s3.x; v[u][1] = ((s0.y + s1.y) + s2.y) + s3.y;
                    v[u][2] = ((s0.z + s1.z) + s2.z) + s3.z; v[u][3] = ((s0.w + s1.w) + s2.w) + s3.w;
                } else {
                    const float2 s0 = *reinterpret_cast<const float2*>(part(0) + o), s1 = *reinterpret_cast<const float2*>(part(1) + o),
                                 s2 = *reinterpret_cast<const float2*>(part(2) + o), s3 = *reinterpret_cast<const float2*>(part(3) + o);
                    v[u][0] = ((s0.x + s1.x) + s2.x) + s3.x; v[u][1] = ((s0.y + s1.y) + s2.y) + s3.y;
                }
            }
            if constexpr (NORM) {
                const int rl = i * 16 + r16;
                const float ssum = ((rowsum[rl] + rowsum[MT * 16 + rl]) + rowsum[2 * MT * 16 + rl]) + rowsum[3 * MT * 16 + rl];
                const float rstd = 1.0f / sqrtf(ssum * (1.0f / 1024.0f) + 1e-5f);
#pragma unroll
                for (int u = 0; u < (EPI == EPI_SILU ? 2 : 1); ++u)
#pragma unroll
                    for (int e = 0; e < CW; ++e) v[u][e] = v[u][e] * rstd;
            }
            const int n = (blockIdx.x * NTO + to) * 16 + CW * qq;
            if (n >= a.N) continue;
            if constexpr (EPI == EPI_F32) {
                float* op = reinterpret_cast<float*>(a.out) + (size_t)m * a.ldo + n;
#pragma unroll
                for (int e = 0; e < CW; ++e) if (n + e < a.N) op[e] = v[0][e];
            } else {
                uint32_t ob[CW];
#pragma unroll
                for (int e = 0; e < CW; ++e) ob[e] = (EPI == EPI_SILU) ? silu_mul_bf(f2bf(v[0][e]), f2bf(v[EPI == EPI_SILU ? 1 : 0][e])) : f2bf(v[0][e]);
                uint16_t* op = reinterpret_cast<uint16_t*>(a.out) + (size_t)m * a.ldo + n;
                if (n + CW - 1 < a.N || a.ldo >= ((a.N + CW - 1) & ~(CW - 1))) {
                    if constexpr (CW == 4) *reinterpret_cast<uint2*>(op) = make_uint2(ob[0] | (ob[1] << 16), ob[2] | (ob[3] << 16));
                    else *reinterpret_cast<uint32_t*>(op) = ob[0] | (ob[1] << 16);
                } else {
#pragma unroll
                    for (int e = 0; e < CW; ++e) if (n + e < a.N) op[e] = (uint16_t)ob[e];
                }
            }
        }
    } else {
        // 16 segments: one output per thread (256 of the 1024 threads), ((G0 + G1) + G2) + G3 with G = ((s0 + s1) + s2) + s3
        if (tid < 256) {
            const int r = (tid >> 6) & 3, l2 = tid & 63;
            const int m = blockIdx.y * 16 + 4 * (l2 >> 4) + r, n = blockIdx.x * 16 + (l2 & 15);
            if (m < a.M && n < a.N) {
                const int o = r * 64 + l2;
                float tot = 0.0f;
#pragma unroll
                for (int gsum = 0; gsum < 4; ++gsum) {
                    float s4 = part(4 * gsum)[o];
                    s4 = s4 + part(4 * gsum + 1)[o]; s4 = s4 + part(4 * gsum + 2)[o]; s4 = s4 + part(4 * gsum + 3)[o];
                    tot = gsum == 0 ? s4 : tot + s4;
                }
                if constexpr (EPI == EPI_F32) reinterpret_cast<float*>(a.out)[(size_t)m * a.ldo + n] = tot;
                else if constexpr (EPI == EPI_BF16) reinterpret_cast<uint16_t*>(a.out)[(size_t)m * a.ldo + n] = (uint16_t)f2bf(tot);
                else reinterpret_cast<uint16_t*>(a.out)[(size_t)m * a.ldo + n] = (uint16_t)f2bf(bf2f(hres) + rbf(tot));     // EPI_RESID: h = bf16(h + bf16(y))
            }
        }
    }
    T3_G2STAMP(5);
}

// ------------------------------------------------------------------------------------------------
// Fold + epilogue shared by the pipelined forms below (and by tools/diag/gemm2_split_kernel.inc, the measured-and-rejected variant with
// the first (gate, up) pair's epilogue under the second pair's weight stream: 0.6 % slower at C3, profiles/NOTES.md).
// ------------------------------------------------------------------------------------------------
// fold + epilogue of NTL output tiles (each from a (gate, up) pair of packed tiles) whose per-wave partials [packed tile 2 to + u][r][lane]
// start at p0 (wave w: + w * pstride floats); the workgroup's output tiles are bx * NTO_WG + to_off + to, its m-tiles mt0 .. mt0 + MT - 1.
// Thread t of nth takes pieces of CW columns.  NORM SiLU form only.
template <int MT, int NTL, int NTO_WG, int CW, int EPI = EPI_SILU>
__device__ __forceinline__ void gemm2_fold_silu(const GemmArgs& a, const float* p0, int pstride, const float* rowsum, int to_off, int t, int nth, int mt0) {
    static_assert(EPI == EPI_SILU || EPI == EPI_BF16, "gemm2_fold_silu epilogues");
    constexpr int NU = EPI == EPI_SILU ? 2 : 1;               // packed tiles per output tile: a (gate, up) pair, or the tile itself (EPI_BF16)
    constexpr int PPT = 16 / CW, PIECES = MT * NTL * 16 * PPT, NTP = NU * NTL;
    for (int p = t; p < PIECES; p += nth) {
        const int ito = p / (16 * PPT), r16 = (p / PPT) & 15, qq = p % PPT;
        const int i = ito / NTL, to = ito % NTL;
        const int m = (mt0 + i) * 16 + r16;
        if (m >= a.M) continue;
        const int g = r16 >> 2, r = r16 & 3;
        float v[2][CW];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int o = ((i * NTP + NU * to + u) * 4 + r) * 64 + 16 * g + CW * qq;
            if constexpr (CW == 2) {
                const float2 s0 = *reinterpret_cast<const float2*>(p0 + o), s1 = *reinterpret_cast<const float2*>(p0 + pstride + o),
                             s2 = *reinterpret_cast<const float2*>(p0 + 2 * pstride + o), s3 = *reinterpret_cast<const float2*>(p0 + 3 * pstride + o);
                v[u][0] = ((s0.x + s1.x) + s2.x) + s3.x; v[u][1] = ((s0.y + s1.y) + s2.y) + s3.y;
            } else {
                static_assert(CW == 1 || CW == 2, "gemm2_fold_silu piece width");
                v[u][0] = ((p0[o] + p0[pstride + o]) + p0[2 * pstride + o]) + p0[3 * pstride + o];
            }
        }
        const int rl = i * 16 + r16;
        const float ssum = ((rowsum[rl] + rowsum[MT * 16 + rl]) + rowsum[2 * MT * 16 + rl]) + rowsum[3 * MT * 16 + rl];
        const float rstd = 1.0f / sqrtf(ssum * (1.0f / 1024.0f) + 1e-5f);
        const int n = (blockIdx.x * NTO_WG + to_off + to) * 16 + CW * qq;
        if (n >= a.N) continue;
        uint32_t ob[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) ob[e] = EPI == EPI_SILU ? silu_mul_bf(f2bf(v[0][e] * rstd), f2bf(v[NU - 1][e] * rstd)) : f2bf(v[0][e] * rstd);
        uint16_t* op = reinterpret_cast<uint16_t*>(a.out) + (size_t)m * a.ldo + n;
        if constexpr (CW == 2) {
            if (n + 1 < a.N || a.ldo >= ((a.N + 1) & ~1)) *reinterpret_cast<uint32_t*>(op) = ob[0] | (ob[1] << 16);
            else op[0] = (uint16_t)ob[0];
        } else op[0] = (uint16_t)ob[0];
    }
}
// ------------------------------------------------------------------------------------------------
// gemm2_pipe_kernel: gate/up from 81 rows on (decode steps of 41+ utterances; C4 runs at 256 rows).  Weights stationary as in
// gemm2_loop_kernel -- a workgroup owns two (gate, up) pairs = 4 packed n-tiles, 128 registers per compute wave, and a share of the
// 16-row groups -- but the groups are PIPELINED through two wave sets instead of walked serially: the four compute waves run group
// j + 1 on the matrix cores while the four epilogue waves fold, normalise, SiLU-multiply and store group j.  The looped form spent ~5 of
// its 16.6 us at 256 rows in four serial epilogues (~300 dependent vector instructions per thread, nothing else running on the CU) and
// had no registers left for epilogue waves (4 stationary tiles + 64 registers of rows in flight = 256 + 44 spill-over AGPRs).  Here
//   * the activation rows come by LDS-DMA straight into a wave-private, double-buffered, XOR-swizzled A image (the swizzle is applied to
//     the SOURCE address: a DMA piece writes LDS linearly), two groups ahead, so no row ever sits in a register: ~170 registers, 8 waves;
//   * partials and row statistics go to a double-buffered exchange area; ONE workgroup barrier per group hands group j to the epilogue
//     waves and, because they reach it only after finishing group j - 1, also frees buffer (j + 1) & 1 for the compute waves.
// Same numbers as gemm2_kernel: per (row, column) the same MFMA chain per segment, ((s0 + s1) + s2) + s3, rstd, SiLU.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_byte_addr);
// NT packed n-tiles per workgroup: 4 = two (gate, up) pairs (EPI_SILU), or 3 / 4 plain tiles (EPI_BF16: qkv)
template <int NT, int EPI>
__global__ __launch_bounds__(512) void gemm2_pipe_kernel(GemmArgs a) {
    constexpr int NW = 4, EW = 4, KBS = 8, AIMG = 16 * KBS * 64;                  // a wave's A image of one group: 16 rows x 512 B = 8 KiB
    constexpr int NTO = EPI == EPI_SILU ? NT / 2 : NT;                              // output tiles per workgroup
    constexpr int PFL = NT * 256;                                                   // a wave's partials of one group: [4 packed tiles][4][64] floats
    // LDS: [2 buffers][4 waves] A image (64 KiB) | [2][4 waves] partials (32 KiB) | [2][4][16] row statistic
    extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* pbase = reinterpret_cast<float*>(lds2 + 2 * NW * AIMG);
    float* rbase = pbase + 2 * NW * PFL;
    const int mgroups = (a.M + 15) / 16;
    const int g0 = blockIdx.y, gs = gridDim.y;                  // this workgroup's groups: g0, g0 + gs, ...
    const int ng = g0 < mgroups ? (mgroups - g0 + gs - 1) / gs : 0;
    if (ng == 0) return;
    if (wave >= NW) {
        for (int j = 0; j < ng; ++j) {
            __builtin_amdgcn_s_barrier();                        // group j's partials are complete (and the compute waves may take buffer (j + 1) & 1)
            gemm2_fold_silu<1, NTO, NTO, 2, EPI>(a, pbase + (size_t)(j & 1) * NW * PFL, PFL, rbase + (j & 1) * NW * 16, 0, tid - NW * 64, EW * 64, g0 + j * gs);
        }
        return;
    }
    const int c = lane & 15, q = lane >> 4;
    const int KB = a.K >> 5, kb0 = wave * KBS;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds2;
    // DMA piece t of a group covers image rows 2 t, 2 t + 1: lane l fills position p = l % 32 of row 2 t + l / 32, which holds the row's
    // 16-byte chunk p ^ (row & 15) (a_img_off<8>): the permutation stays inside the row's 512 bytes, whole 128-byte lines are fetched
    const int drow = lane >> 5, dpos = lane & 31;
    auto issue_a = [&](int g, int buf) {
        const unsigned img = lds0 + (unsigned)((buf * NW + wave) * AIMG);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = 2 * t + drow;
            int m = g * 16 + row; m = m < a.M ? m : a.M - 1;     // padded rows re-read the last row; their outputs are dropped
            glds16(a.X + (size_t)m * a.K + kb0 * 32 + ((dpos ^ (row & 15)) << 3), img + t * 1024);
        }
    };
    uint4_v wr[KBS][NT];
    issue_a(g0, 0);
#pragma unroll
    for (int kb = 0; kb < KBS; ++kb)
#pragma unroll
        for (int t = 0; t < NT; ++t) gload16_nt(wr[kb][t], a.Wp + ((size_t)(blockIdx.x * NT + t) * KB + kb0 + kb) * 64 + lane);
    if (ng > 1) issue_a(g0 + gs, 1);
    for (int j = 0; j < ng; ++j) {
        const unsigned char* aimg = lds2 + (size_t)((j & 1) * NW + wave) * AIMG;
        const bool next_in_flight = j + 1 < ng;                  // the next group's 8 pieces are younger than this group's
        // this group's rows have landed (j = 0: the 32 weight tiles are younger too, and are waited for k-block by k-block below)
        if (j == 0) { if (next_in_flight) wait_vmcnt<KBS * NT + 8>(); else wait_vmcnt<KBS * NT>(); }
        else { if (next_in_flight) wait_vmcnt<8>(); else wait_vmcnt<0>(); }
        f32x4 acc[NT], ss = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        uint4 afr[KBS];                                          // every A fragment of the group requested ahead of the MFMAs (see gemm2_kernel)
#pragma unroll
        for (int kb = 0; kb < KBS; ++kb) afr[kb] = *reinterpret_cast<const uint4*>(aimg + a_img_off<KBS>(c, 4 * kb + q));
        static_for([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            const uint4 af = afr[kb];
            if (j == 0) {
                if (next_in_flight) wait_vmcnt<(KBS - 1 - kb) * NT + 8>(); else wait_vmcnt<(KBS - 1 - kb) * NT>();
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) landed(wr[kb][t]);
            ss = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af), as_frag(af), ss, 0, 0, 0);       // diagonal = sum of squares
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af), as_frag4(wr[kb][t]), acc[t], 0, 0, 0);
        }, std::make_integer_sequence<int, KBS>{});
        float* myp = pbase + (size_t)((j & 1) * NW + wave) * PFL;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) myp[(t * 4 + r) * 64 + lane] = acc[t][r];
        {
            const int r = c & 3;
            const float d = r == 0 ? ss[0] : r == 1 ? ss[1] : r == 2 ? ss[2] : ss[3];
            if ((c >> 2) == q) rbase[((j & 1) * NW + wave) * 16 + c] = d;
        }
        // raw barrier: the next group's DMA pieces stay in flight across it; this wave's LDS writes must be out and its reads of the A
        // image are (the MFMAs consumed them), so the image may be refilled right behind the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (j + 2 < ng) issue_a(g0 + (j + 2) * gs, j & 1);
    }
}

// ------------------------------------------------------------------------------------------------
// gemm2_pipe16_kernel: gate/up from 49 rows on (four 16-row groups = two per workgroup: C3's 64-row steps included, round 4) at 16 waves per workgroup.  In gemm2_pipe_kernel<4, EPI_SILU> the four epilogue waves set
// the pace: a group's 512 outputs at two per thread are ~215 dependent vector instructions per wave, ~1.0 us, against ~0.4 us of matrix
// work (tools/gemm_bench: +4.2 us per 128 rows).  Epilogue waves are only as many as the registers allow, and a compute wave that holds
// two pairs' weight tiles needs 197.  Here the two (gate, up) pairs go to two SETS of four compute waves (64 registers of weights each,
// the two sets share the A images: the set of pair 0 requests them, its statistic is the row statistic), which leaves room for EIGHT
// epilogue waves at <= 128 registers: one output per thread, four waves per SIMD to fill the issue slots.  One barrier per group as before;
// a wave of set 0 makes sure the next group's rows have landed before it enters the barrier behind which set 1 reads them.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gemm2_pipe16_kernel(GemmArgs a) {
    constexpr int NS = 4, KBS = 8, AIMG = 16 * KBS * 64;                          // K segments; a segment's A image of one group: 16 rows x 512 B
    constexpr int NA = 3;                                                           // A buffers: a group's rows are asked for TWO groups ahead (an L2 round trip under load is about one group)
    constexpr int PFL = 2 * 256;                                                    // a compute wave's partials of one group: [gate | up][4][64] floats
    // LDS: [3 buffers][4 segments] A image (96 KiB) | [2][2 pairs][4 segments] partials (32 KiB) | [2][4][16] row statistic | 256 B prefetch dump
    extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* pbase = reinterpret_cast<float*>(lds2 + NA * NS * AIMG);
    float* rbase = pbase + 2 * 2 * NS * PFL;
    const int mgroups = (a.M + 15) / 16;
    const int g0 = blockIdx.y, gs = gridDim.y;
    const int ng = g0 < mgroups ? (mgroups - g0 + gs - 1) / gs : 0;
    if (ng == 0) return;
    if (wave >= 8) {
        const int te = tid - 512, pair = te >> 8;
        __builtin_amdgcn_s_barrier();                            // the first group's rows are in (compute waves only)
        for (int j = 0; j < ng; ++j) {
            __builtin_amdgcn_s_barrier();                        // group j's partials are complete
            // PrefetchArgs (64-row decode steps: down_proj's weights into the L2 of the XCD that will read them): behind the FIRST group's
            // barrier every weight tile of this workgroup has landed and only rows (L2) move until the launch ends
            if (j == 0 && ((gridDim.x * gridDim.y) & 7) == 0) {
                const int lin = blockIdx.y * gridDim.x + blockIdx.x;
                const unsigned dump = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(rbase + 2 * NS * 16);
                prefetch_next_weights(a.pf, lin & 7, (lin >> 3) * 8 + (wave - 8), ((gridDim.x * gridDim.y) >> 3) * 8, lane, dump);
            }
            gemm2_fold_silu<1, 1, 2, 1>(a, pbase + (size_t)((j & 1) * 2 + pair) * NS * PFL, PFL, rbase + (j & 1) * NS * 16, pair, te & 255, 256, g0 + j * gs);
        }
        return;
    }
    const int seg = wave & 3, pair = wave >> 2;
    const int c = lane & 15, q = lane >> 4;
    const int KB = a.K >> 5, kb0 = seg * KBS;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds2;
    const int drow = lane >> 5, dpos = lane & 31;
    auto issue_a = [&](int g, int buf) {                        // set 0 only: see gemm2_pipe_kernel
        const unsigned img = lds0 + (unsigned)((buf * NS + seg) * AIMG);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = 2 * t + drow;
            int m = g * 16 + row; m = m < a.M ? m : a.M - 1;
            glds16(a.X + (size_t)m * a.K + kb0 * 32 + ((dpos ^ (row & 15)) << 3), img + t * 1024);
        }
    };
    // younger = 8 * ahead + BASE: waits of set 0 while `ahead` (0, 1 or 2) later groups' pieces are in flight behind what is waited for
    auto wait_ahead = [&](int ahead, auto base_c) {
        constexpr int BASE = decltype(base_c)::value;
        if (ahead == 2) wait_vmcnt<BASE + 16>(); else if (ahead == 1) wait_vmcnt<BASE + 8>(); else wait_vmcnt<BASE>();
    };
    uint4_v wr[KBS][2];
    if (pair == 0) issue_a(g0, 0);
#pragma unroll
    for (int kb = 0; kb < KBS; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t) gload16_nt(wr[kb][t], a.Wp + ((size_t)(blockIdx.x * 4 + 2 * pair + t) * KB + kb0 + kb) * 64 + lane);
    const int ahead0 = pair == 0 ? (ng > 2 ? 2 : ng - 1) : 0;    // groups asked for behind the weights in the prologue
    if (pair == 0) {
        if (ng > 1) issue_a(g0 + gs, 1);
        if (ng > 2) issue_a(g0 + 2 * gs, 2);
        wait_ahead(ahead0, std::integral_constant<int, 2 * KBS>{});      // the first group's rows have landed
    }
    __builtin_amdgcn_s_barrier();
    int buf = 0;
    for (int j = 0; j < ng; ++j) {
        const unsigned char* aimg = lds2 + (size_t)(buf * NS + seg) * AIMG;
        f32x4 acc[2], ss = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[0] = ss; acc[1] = ss;
        constexpr int AFD = 3;                                   // A fragments this many k-blocks ahead of their MFMAs (see gemm2_kernel; 128 registers here)
        uint4 afr[AFD];
#pragma unroll
        for (int d = 0; d < AFD; ++d) afr[d] = *reinterpret_cast<const uint4*>(aimg + a_img_off<KBS>(c, 4 * d + q));
        static_for([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            const uint4 af = afr[kb % AFD];
            if constexpr (kb + AFD < KBS) afr[kb % AFD] = *reinterpret_cast<const uint4*>(aimg + a_img_off<KBS>(c, 4 * (kb + AFD) + q));
            if (j == 0) wait_ahead(ahead0, std::integral_constant<int, (KBS - 1 - kb) * 2>{});
            landed(wr[kb][0]); landed(wr[kb][1]);
            if (pair == 0) ss = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af), as_frag(af), ss, 0, 0, 0);       // diagonal = sum of squares
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af), as_frag4(wr[kb][0]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af), as_frag4(wr[kb][1]), acc[1], 0, 0, 0);
        }, std::make_integer_sequence<int, KBS>{});
        float* myp = pbase + (size_t)(((j & 1) * 2 + pair) * NS + seg) * PFL;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) myp[(t * 4 + r) * 64 + lane] = acc[t][r];
        if (pair == 0) {
            const int r = c & 3;
            const float d = r == 0 ? ss[0] : r == 1 ? ss[1] : r == 2 ? ss[2] : ss[3];
            if ((c >> 2) == q) rbase[((j & 1) * NS + seg) * 16 + c] = d;
            // group j + 1's rows (asked for two groups ago) are in LDS before set 1 may read them; group j + 2's may still fly
            if (j + 2 < ng) wait_vmcnt<8>(); else wait_vmcnt<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (pair == 0 && j + 3 < ng) issue_a(g0 + (j + 3) * gs, buf);      // into the image every wave has just finished reading
        buf = buf == NA - 1 ? 0 : buf + 1;
    }
}

// ------------------------------------------------------------------------------------------------
// gemm2_loop_kernel: gate/up (4 waves, two gate/up pairs per workgroup), o and down (16 waves) from 65-81 rows on (decode steps of
// 41+ utterances, C4; thresholds and measurements in launch_gemm).  gemm2_kernel launches one
// workgroup per (n-group, m-group): at 256 rows that is 1024 single-occupancy workgroups in four rounds, each of which streams its
// weight tiles again (268 MB of L2 -> CU traffic per layer) and pays a cold start.  Here a workgroup OWNS an n-group: its weight
// tiles are loaded once and stay in registers, and it walks the m-groups, the next group's activation rows in flight (asm loads)
// while the current group runs on the matrix cores.  Same numbers: every (row, column) is computed exactly as in gemm2_kernel.
// ------------------------------------------------------------------------------------------------
template <int MT, int NT, int EPI, int NW, int KBS, bool NORM>
__global__ __launch_bounds__(NW * 64) void gemm2_loop_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];      // [NW waves][MT][16 rows][KBS * 64 B] | NORM: float [NW][MT*16]
    static_assert(NT <= KBS && (KBS == 8 || KBS == 2) && (NW == 4 || NW == 16) && (NW == 4 || (MT == 1 && NT <= 2)), "gemm2_loop shapes");
    constexpr int LPR = KBS * 4, RPI = 64 / LPR, ABYTES = MT * KBS * 1024, TILES = MT * NT;
    constexpr int NTO = (EPI == EPI_SILU) ? NT / 2 : NT, PIECES = MT * NTO * 64, PIT = (PIECES + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int KB = a.K >> 5, kb0 = wave * KBS;
    unsigned char* aimg = lds2 + (size_t)wave * ABYTES;
    float* rowsum = reinterpret_cast<float*>(lds2 + (size_t)NW * ABYTES);
    const int rsub = lane / LPR, ch = lane % LPR;
    const int mgroups = ((a.M + 15) / 16 + MT - 1) / MT;
    uint4_v ar[MT][KBS], wr[KBS][NT];
    auto issue_a = [&](int g) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int t = 0; t < KBS; ++t) {
                int m = (g * MT + i) * 16 + t * RPI + rsub;
                m = m < a.M ? m : a.M - 1;
                gload16(ar[i][t], a.X + (size_t)m * a.K + kb0 * 32 + ch * 8);
            }
    };
    auto stage_a = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int t = 0; t < KBS; ++t) landed(ar[i][t]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int t = 0; t < KBS; ++t)
                *reinterpret_cast<uint4_v*>(aimg + i * (KBS * 1024) + a_img_off<KBS>(t * RPI + rsub, ch)) = ar[i][t];
        asm volatile("" ::: "memory");
    };
    auto group = [&](int g, auto first_c, uint16_t hres) {
        constexpr bool FIRST = decltype(first_c)::value;
        f32x4 acc[MT][NT], ss[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            ss[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // A fragments AFD k-blocks ahead of their MFMAs (see gemm2_kernel); the two-m-tile forms are at the register file's limit: one ahead
        constexpr int AFD = MT == 1 ? KBS : 1;
        uint4 afr[AFD][MT];
        auto read_af = [&](int slot, int kb) {
#pragma unroll
            for (int i = 0; i < MT; ++i) afr[slot][i] = *reinterpret_cast<const uint4*>(aimg + i * (KBS * 1024) + a_img_off<KBS>(c, 4 * kb + q));
        };
        static_for([&](auto dc) { read_af(decltype(dc)::value, decltype(dc)::value); }, std::make_integer_sequence<int, AFD>{});
        static_for([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            if constexpr (FIRST) {     // the weights land during the first group: younger = the later weight tiles + the next group's rows
                wait_vmcnt<(KBS - 1 - kb) * NT + MT * KBS>();
#pragma unroll
                for (int t = 0; t < NT; ++t) landed(wr[kb][t]);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint4 af = afr[kb % AFD][i];
                if constexpr (NORM) ss[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af), as_frag(af), ss[i], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af), as_frag4(wr[kb][t]), acc[i][t], 0, 0, 0);
            }
            if constexpr (kb + AFD < KBS) read_af(kb % AFD, kb + AFD);
        }, std::make_integer_sequence<int, KBS>{});
        asm volatile("" ::: "memory");
        float* redw = reinterpret_cast<float*>(aimg);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) redw[((i * NT + t) * 4 + r) * 64 + lane] = acc[i][t][r];
        if constexpr (NORM) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int r = c & 3;
                const float d = r == 0 ? ss[i][0] : r == 1 ? ss[i][1] : r == 2 ? ss[i][2] : ss[i][3];
                if ((c >> 2) == q) rowsum[wave * (MT * 16) + i * 16 + c] = d;
            }
        }
        __syncthreads();
        auto part = [&](int w) { return reinterpret_cast<const float*>(lds2 + (size_t)w * ABYTES); };
        if constexpr (NW == 16) {
            // 16 segments: one output per thread (256 NT of the 1024 threads), ((G0 + G1) + G2) + G3 with G = ((s0 + s1) + s2) + s3
            if (tid < 256 * NT) {
                const int tl = tid >> 8, r = (tid >> 6) & 3, l2 = tid & 63;
                const int m = g * 16 + 4 * (l2 >> 4) + r, n = (blockIdx.x * NT + tl) * 16 + (l2 & 15);
                if (m < a.M && n < a.N) {
                    const int o = (tl * 4 + r) * 64 + l2;
                    float tot = 0.0f;
#pragma unroll
                    for (int gsum = 0; gsum < 4; ++gsum) {
                        float s4 = part(4 * gsum)[o];
                        s4 = s4 + part(4 * gsum + 1)[o]; s4 = s4 + part(4 * gsum + 2)[o]; s4 = s4 + part(4 * gsum + 3)[o];
                        tot = gsum == 0 ? s4 : tot + s4;
                    }
                    if constexpr (EPI == EPI_F32) reinterpret_cast<float*>(a.out)[(size_t)m * a.ldo + n] = tot;
                    else reinterpret_cast<uint16_t*>(a.out)[(size_t)m * a.ldo + n] = (uint16_t)f2bf(bf2f(hres) + rbf(tot));     // EPI_RESID
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < PIT; ++k) {
            const int p = tid + k * 256;
            if (p >= PIECES) continue;
            const int ito = p >> 6, r16 = (p >> 2) & 15, qq = p & 3;
            const int i = ito / NTO, to = ito % NTO;
            const int m = (g * MT + i) * 16 + r16;
            if (m >= a.M) continue;
            const int gq = r16 >> 2, r = r16 & 3;
            float v[EPI == EPI_SILU ? 2 : 1][4];
#pragma unroll
            for (int u = 0; u < (EPI == EPI_SILU ? 2 : 1); ++u) {
                const int it = i * NT + (EPI == EPI_SILU ? 2 * to + u : to);
                const int o = (it * 4 + r) * 64 + 16 * gq + 4 * qq;
                const float4 s0 = *reinterpret_cast<const float4*>(part(0) + o), s1 = *reinterpret_cast<const float4*>(part(1) + o),
                             s2 = *reinterpret_cast<const float4*>(part(2) + o), s3 = *reinterpret_cast<const float4*>(part(3) + o);
                v[u][0] = ((s0.x + s1.x) + s2.x) + s3.x; v[u][1] = ((s0.y + s1.y) + s2.y) + s3.y;
                v[u][2] = ((s0.z + s1.z) + s2.z) + s3.z; v[u][3] = ((s0.w + s1.w) + s2.w) + s3.w;
            }
            if constexpr (NORM) {
                const int rl = i * 16 + r16;
                const float ssum = ((rowsum[rl] + rowsum[MT * 16 + rl]) + rowsum[2 * MT * 16 + rl]) + rowsum[3 * MT * 16 + rl];
                const float rstd = 1.0f / sqrtf(ssum * (1.0f / 1024.0f) + 1e-5f);
#pragma unroll
                for (int u = 0; u < (EPI == EPI_SILU ? 2 : 1); ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[u][e] = v[u][e] * rstd;
            }
            const int n = (blockIdx.x * NTO + to) * 16 + 4 * qq;
            if (n >= a.N) continue;
            uint32_t ob[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) ob[e] = (EPI == EPI_SILU) ? silu_mul_bf(f2bf(v[0][e]), f2bf(v[EPI == EPI_SILU ? 1 : 0][e])) : f2bf(v[0][e]);
            uint16_t* op = reinterpret_cast<uint16_t*>(a.out) + (size_t)m * a.ldo + n;
            if (n + 3 < a.N || a.ldo >= ((a.N + 3) & ~3)) *reinterpret_cast<uint2*>(op) = make_uint2(ob[0] | (ob[1] << 16), ob[2] | (ob[3] << 16));
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < a.N) op[e] = (uint16_t)ob[e];
            }
        }
    };

    // the m-groups are dealt round-robin over gridDim.y workgroups per n-group (two workgroups per CU overlap one's epilogue with
    // the other's matrix work); the launcher guarantees every workgroup at least two groups
    const int g0 = blockIdx.y, gs = gridDim.y;
    // EPI_RESID: the residual operand of a group is a compiler-counted load: requested before the asm loads of the first group (it
    // must be the oldest there: the hand-counted waits assume only asm loads behind them), at the start of every later one
    auto load_res = [&](int g) -> uint16_t {
        if constexpr (EPI == EPI_RESID) {
            const int tl = tid >> 8, r = (tid >> 6) & 3, l2 = tid & 63;
            const int m = g * 16 + 4 * (l2 >> 4) + r, n = (blockIdx.x * NT + tl) * 16 + (l2 & 15);
            if (tid < 256 * NT && m < a.M && n < a.N) return reinterpret_cast<const uint16_t*>(a.out)[(size_t)m * a.ldo + n];
        }
        return 0;
    };
    uint16_t hres = load_res(g0);
    issue_a(g0);
#pragma unroll
    for (int kb = 0; kb < KBS; ++kb)
#pragma unroll
        for (int t = 0; t < NT; ++t) gload16_nt(wr[kb][t], a.Wp + ((size_t)(blockIdx.x * NT + t) * KB + kb0 + kb) * 64 + lane);
    wait_vmcnt<KBS * NT>();                        // the first group's rows are in (the weight tiles are younger)
    stage_a();
    issue_a(g0 + gs);
    group(g0, std::true_type{}, hres);
    for (int g = g0 + gs; g < mgroups; g += gs) {
        __syncthreads();                           // every wave is done with the previous group's partials: the A images may be overwritten
        wait_vmcnt<0>();                           // this group's rows (nothing younger is in flight)
        stage_a();
        hres = load_res(g);
        if (g + gs < mgroups) issue_a(g + gs);
        group(g, std::false_type{}, hres);
    }
}

// ------------------------------------------------------------------------------------------------
// down_proj from ~113 rows on (decode steps of 57+ utterances, C4): TWO n-tiles per workgroup.  With one n-tile per workgroup
// (gemm2_loop_kernel<1, 1, EPI_RESID, 16, 8>) the launch is what a CU can take in from L2 (~70 GB/s): at 256 rows 128 KiB of weights +
// 4 x 128 KiB of rows per workgroup = 164 MB through L2 per launch, 11.8 us.  Two tiles per workgroup (32 x 8 workgroups at 256 rows:
// 256 KiB of weights + 2 x 128 KiB of rows each) is the balanced cut of the same work, 20 % less per CU -- but two tiles of a 256-deep K
// slice are 64 registers, and with the rows staged through registers as well (32) the sixteen-wave workgroup's 128 are gone
// (profiles/NOTES.md).  Here the rows come by LDS-DMA straight into the wave's own (swizzled) image: a wave reads a group's eight A
// fragments into registers, asks for the NEXT group's rows into the same image, and runs its sixteen MFMAs from registers while they
// fly; the partials have LDS of their own (128 KiB of images + 32 KiB of partials = all of a CU's LDS).
// Same numbers: per output the sixteen segment chains and the fold of gemm2_kernel's 16-wave form.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gemm2_down2_kernel(GemmArgs a) {
    constexpr int KBS = 8, NT = 2, NW = 16, AIMG = 16 * KBS * 64, PFL = NT * 256;       // a wave's A image: 16 rows x 512 B; its partials: [2 tiles][4][64] floats
    extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];                  // [16] A images | [16] partials
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int KB = a.K >> 5, kb0 = wave * KBS;
    const unsigned char* aimg = lds2 + (size_t)wave * AIMG;
    const unsigned img = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds2 + (unsigned)(wave * AIMG);
    float* pbase = reinterpret_cast<float*>(lds2 + (size_t)NW * AIMG);
    float* myp = pbase + (size_t)wave * PFL;
    const int mgroups = (a.M + 15) / 16;
    const int g0 = blockIdx.y, gs = gridDim.y;
    const int ng = g0 < mgroups ? (mgroups - g0 + gs - 1) / gs : 0;
    if (ng == 0) return;
    const int drow = lane >> 5, dpos = lane & 31;
    auto issue_a = [&](int g) {                                  // 8 pieces of 1 KiB: rows 2 t, 2 t + 1 of the group, this wave's K slice
        int dr_ = drow, dp_ = dpos;
        asm volatile("" : "+v"(dr_), "+v"(dp_));                 // (opaque per call: hipcc would keep the eight pieces' offsets in registers across the group loop, and this kernel has none to spare)
        const int drow = dr_, dpos = dp_;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = 2 * t + drow;
            int m = g * 16 + row; m = m < a.M ? m : a.M - 1;
            glds16(a.X + (size_t)m * a.K + kb0 * 32 + ((dpos ^ (row & 15)) << 3), img + t * 1024);
        }
    };
    // the residual operand of this thread's output (threads 0..511: tile tid >> 8): a compiler-counted load
    const int tl = tid >> 8, fr = (tid >> 6) & 3, fl = tid & 63;
    const int n_out = (blockIdx.x * NT + tl) * 16 + (fl & 15);
    auto load_res = [&](int g) -> uint16_t {
        const int m = g * 16 + 4 * (fl >> 4) + fr;
        if (tid < 256 * NT && m < a.M && n_out < a.N) return reinterpret_cast<const uint16_t*>(a.out)[(size_t)m * a.ldo + n_out];
        return 0;
    };
    auto fold = [&](int g, uint16_t hres) {
        if (tid < 256 * NT) {
            const int m = g * 16 + 4 * (fl >> 4) + fr;
            if (m < a.M && n_out < a.N) {
                const int o = (tl * 4 + fr) * 64 + fl;
                float tot = 0.0f;
#pragma unroll
                for (int gsum = 0; gsum < 4; ++gsum) {
                    float s4 = pbase[(size_t)(4 * gsum) * PFL + o];
                    s4 = s4 + pbase[(size_t)(4 * gsum + 1) * PFL + o]; s4 = s4 + pbase[(size_t)(4 * gsum + 2) * PFL + o]; s4 = s4 + pbase[(size_t)(4 * gsum + 3) * PFL + o];
                    tot = gsum == 0 ? s4 : tot + s4;
                }
                reinterpret_cast<uint16_t*>(a.out)[(size_t)m * a.ldo + n_out] = (uint16_t)f2bf(bf2f(hres) + rbf(tot));     // h = bf16(h + bf16(y))
            }
        }
    };
    uint16_t hres = load_res(g0);                                // the oldest load: the hand-counted waits below only count what is younger than their target
    issue_a(g0);
    uint4_v wr[KBS][NT];
#pragma unroll
    for (int kb = 0; kb < KBS; ++kb)
#pragma unroll
        for (int t = 0; t < NT; ++t) gload16_nt(wr[kb][t], a.Wp + ((size_t)(blockIdx.x * NT + t) * KB + kb0 + kb) * 64 + lane);
    wait_vmcnt<KBS * NT>();                                      // the first group's rows are in the image (the sixteen weight tiles are younger)
    for (int j = 0; j < ng; ++j) {
        const int g = g0 + j * gs;
        const bool next = j + 1 < ng;
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        constexpr int AFD = 3;                                   // A fragments this many k-blocks ahead of their MFMAs
        uint4 afr[AFD];
#pragma unroll
        for (int d = 0; d < AFD; ++d) afr[d] = *reinterpret_cast<const uint4*>(aimg + a_img_off<KBS>(c, 4 * d + q));
        static_for([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            const uint4 af = afr[kb % AFD];
            if constexpr (kb + AFD < KBS) afr[kb % AFD] = *reinterpret_cast<const uint4*>(aimg + a_img_off<KBS>(c, 4 * (kb + AFD) + q));
            if (j == 0) wait_vmcnt<(KBS - 1 - kb) * NT>();       // first group: this k-block's two weight tiles have landed (the later tiles are younger)
#pragma unroll
            for (int t = 0; t < NT; ++t) landed(wr[kb][t]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af), as_frag4(wr[kb][t]), acc[t], 0, 0, 0);
        }, std::make_integer_sequence<int, KBS>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every fragment of the image is in registers: it may be overwritten
        if (next) issue_a(g + gs);                               // flies under the partial writes, the fold and its two barriers
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) myp[(t * 4 + r) * 64 + lane] = acc[t][r];
        if (j > 0) hres = load_res(g);                           // (behind the next group's pieces: the compiler's own wait for it covers them too)
        __syncthreads();
        fold(g, hres);
        if (next) {
            __syncthreads();                                     // every thread is done with this group's partials
            wait_vmcnt<0>();                                     // the next group's rows are in the image
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Prefill-sized GEMM (M >= 256 rows): the same numbers as gemm2_kernel, another schedule.  A workgroup of four waves owns a
// 128-row x 64-column tile (4 packed n-tiles); every K step of 32 is staged once through LDS (activations 128 x 64 B row
// pieces; weights: 4 packed 1 KiB fragments, the NORM forms' carrying the norm weight) and feeds 32 MFMAs, so a weight byte is
// re-read once per 128 rows instead of once per 32 and an activation byte once per 64 columns instead of once per workgroup.
// Contract order per output: one MFMA chain per K segment FROM ZERO in ascending k; segments folded left to right in groups
// of four (G = ((s0 + s1) + s2) + s3), groups folded left to right -- hence three accumulator sets (segment, group, total).
// The row statistic of the NORM forms comes from row_rstd_kernel (the same MFMA chains as gemm2_kernel's own).
// ------------------------------------------------------------------------------------------------
// XCD-aware tile order of the prefill schedules (speed only: any order computes the same tiles).  Workgroup L = by * gx + bx of a launch
// runs on XCD L % 8 (tools/xcd_probe.hip), each XCD has its own 4 MiB L2, and in plain row-major order the workgroups that share an
// operand tile sit on DIFFERENT XCDs: at 8 178 rows down_proj's eight column tiles of a row tile went to the eight XCDs, every L2 streamed
// all of A, and the launch pulled 562 MB over the fabric for 75 MB of operands (gate/up: 596 MB for 33.5; FETCH_SIZE,
// profiles/r04_p_pmc_fetch_size_by_kernel.json).  The map gives an XCD
//   * a block of gx / 8 column tiles and every row tile (wide outputs: gate/up, qkv) -- its weight columns stay resident in its L2, a row
//     tile is fetched once per XCD and shared by the column workgroups that run side by side; or
//   * whole row tiles, eight at a time dealt over the XCDs (narrow outputs: o, down) -- a row tile's column workgroups share its rows in
//     one L2, the (small) weight matrix streams through every L2 once.
__device__ __forceinline__ void pgemm_tile_map(int gx, int gy, int& tx, int& ty) {
    const int L = blockIdx.y * gx + blockIdx.x, xcd = L & 7, slot = L >> 3;
    tx = blockIdx.x; ty = blockIdx.y;
    if ((gx & 7) == 0 && gx >= 24) {                         // column blocks (gx * gy is a multiple of 8)
        const int cpx = gx >> 3;
        tx = xcd * cpx + slot % cpx; ty = slot / cpx;
    } else if (gx <= 16) {                                   // row tiles, in bands of eight; the rows beyond the last full band keep the plain order
        const int band_rows = gy & ~7;
        if (L < gx * band_rows) { tx = slot % gx; ty = (slot / gx) * 8 + xcd; }
    }
}

// Epilogue of the prefill schedules.  Their MFMAs take the WEIGHT fragment as the row operand and the activation fragment as the column
// operand (D = W X^T: per output the same products and the same sums -- the matrix instruction is symmetric in its two operands, and the
// parity tests hold it to the oracle), so a lane holds FOUR CONSECUTIVE COLUMNS n = 4 (lane >> 4) + r of ONE row m = lane & 15 of every
// 16 x 16 tile: one 8-byte (bf16) or 16-byte (fp32) store where the row-major orientation needed four 2-byte ones, and one rstd per tile row.
// v = the tile's accumulators (EPI_SILU: vu = the up tile's); nt = packed n-tile (EPI_SILU: the gate tile of the pair); rs = rstd or 1.
template <int EPI>
__device__ __forceinline__ void pgemm_store4(const GemmArgs& a, int m, int nt, int q, f32x4 v, f32x4 vu, float rs) {
    if (m >= a.M) return;
    const int n0 = (EPI == EPI_SILU ? (nt >> 1) : nt) * 16 + 4 * q;
    if (n0 >= a.N) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = v[r] * rs; if (EPI == EPI_SILU) vu[r] = vu[r] * rs; }      // (NORM forms; rs = 1 multiplies exactly)
    const bool wide = n0 + 3 < a.N && (a.ldo & 3) == 0;
    if constexpr (EPI == EPI_F32) {
        float* op = reinterpret_cast<float*>(a.out) + (size_t)m * a.ldo + n0;
        if (wide) *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        else { for (int r = 0; r < 4; ++r) if (n0 + r < a.N) op[r] = v[r]; }
    } else {
        uint16_t* op = reinterpret_cast<uint16_t*>(a.out) + (size_t)m * a.ldo + n0;
        uint32_t ob[4];
        if constexpr (EPI == EPI_SILU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = silu_mul_bf(f2bf(v[r]), f2bf(vu[r]));
        } else if constexpr (EPI == EPI_BF16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = f2bf(v[r]);
        } else {                                                 // EPI_RESID: h = bf16(h + bf16(y))
            uint32_t hw[4];
            if (wide) { const uint2 h2 = *reinterpret_cast<const uint2*>(op); hw[0] = h2.x & 0xffffu; hw[1] = h2.x >> 16; hw[2] = h2.y & 0xffffu; hw[3] = h2.y >> 16; }
            else { for (int r = 0; r < 4; ++r) hw[r] = n0 + r < a.N ? op[r] : 0u; }
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = f2bf(bf2f((uint16_t)hw[r]) + rbf(v[r]));
        }
        if (wide) *reinterpret_cast<uint2*>(op) = make_uint2(ob[0] | (ob[1] << 16), ob[2] | (ob[3] << 16));
        else { for (int r = 0; r < 4; ++r) if (n0 + r < a.N) op[r] = (uint16_t)ob[r]; }
    }
}

__global__ __launch_bounds__(256) void row_rstd_kernel(const uint16_t* h, float* rstd, int rows) {
    // one wave per 16 rows: per segment of 256 k the wave multiplies its A fragments with themselves (one MFMA chain from +0,
    // ascending k) and keeps the diagonal; the four segment sums fold ((S0 + S1) + S2) + S3 -- gemm2_kernel's own statistic
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int mt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (mt * 16 >= rows) return;
    int m = mt * 16 + c; m = m < rows ? m : rows - 1;
    const uint4* xp = reinterpret_cast<const uint4*>(h + (size_t)m * D + q * 8);
    float tot = 0.0f;
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
        f32x4 ss = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const uint4 af = xp[(sg * 8 + kb) * 4];
            ss = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af), as_frag(af), ss, 0, 0, 0);
        }
        const int r = c & 3;
        const float d = r == 0 ? ss[0] : r == 1 ? ss[1] : r == 2 ? ss[2] : ss[3];
        tot = sg == 0 ? d : tot + d;
    }
    if ((c >> 2) == q && mt * 16 + c < rows) rstd[mt * 16 + c] = 1.0f / sqrtf(tot * (1.0f / 1024.0f) + 1e-5f);
}

__device__ __forceinline__ int pgemm_a_pos(int row, int q) { return (row << 2) + (((row >> 2) & 3) ^ ((4 - q) & 3)); }     // uint4 index in a stage's A image
// one 1 KiB LDS-DMA piece: every lane's 16 bytes at gsrc land at lds_byte_addr (wave-uniform) + 16 * lane.  M0 carries the LDS
// address and is written in the statement that uses it (cdna_hip_programming.md 5.7); hipcc does not count this load: every wait
// for it below is hand-counted.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_byte_addr) {      // (declared ahead of gemm2_pipe_kernel)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// WC = packed n-tiles per wave: 2 (workgroup tile 128 x 64) or 4 (128 x 128: a fragment read from LDS feeds twice the MFMAs and a
// weight / activation byte leaves L2 1.5 times less often; two accumulator sets of 64 registers, two workgroups per CU)
template <int EPI, int NSEG, bool NORM, int WC>
__global__ __launch_bounds__(256, WC == 4 ? 2 : (NSEG == 4 ? 4 : 3)) void pgemm_kernel(GemmArgs a, const float* rstd) {
    // Operand ring in LDS, filled by LDS-DMA three K steps ahead of the MFMAs (no staging registers: the global-load latency of
    // a step is covered by three steps of arithmetic instead of one).  Per stage: A image 128 rows x 64 B (swizzled, below) | 4
    // weight fragments x 1 KiB.  The NORM forms' activations go in untouched (the norm weight lives in the packed matrix).
#ifndef T3_PGEMM_NS
#define T3_PGEMM_NS 3      // measured at 8192 rows: 3 stages (36 KiB, 4 workgroups per CU) 214 / 92 / 91 us (gate-up / o+down / qkv), 4 stages 235 / 91 / 99, 6 stages 296 / 96 / 121
#endif
#ifndef T3_PGEMM_NS_WIDE
#define T3_PGEMM_NS_WIDE 3
#endif
    constexpr int NS = WC == 4 ? T3_PGEMM_NS_WIDE : T3_PGEMM_NS, AHEAD = NS - 1, NTW = 2 * WC, STAGE = 512 + NTW * 64;     // stages in the ring; n-tiles per workgroup; uint4 per stage
    constexpr int PP = 2 + NTW / 4;                                              // DMA pieces per wave and stage
    __shared__ __attribute__((aligned(16))) uint4 ring[NS * STAGE];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;                                   // wave tile: rows 64 wr .., packed n-tiles 2 wc, 2 wc + 1
    const int KB = a.K >> 5, kbs = KB / NSEG;
    int tile_x, tile_y;
    pgemm_tile_map((int)gridDim.x, (int)gridDim.y, tile_x, tile_y);
    const int m0 = tile_y * 128, nt0 = tile_x * NTW;
    // A image: 64-byte rows, so four rows share a 256-byte bank row and the 16 rows of a fragment read would hit 4 bank slots.
    // Chunk q of row r sits at position ((r >> 2) & 3) ^ T[q], T = {0, 3, 2, 1} (an involution): the 16 lanes of each hardware
    // lane group of ds_read_b128 then land on 16 different slots.  A DMA piece writes LDS linearly, so the permutation is applied
    // to the SOURCE: the lane that fills position P = 4 row + p fetches chunk q = T[p ^ ((row >> 2) & 3)] of that row.
    const uint16_t* xsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int P = t + 256 * j, row = P >> 2, q = (4 - ((P & 3) ^ ((row >> 2) & 3))) & 3;
        int m = m0 + row; m = m < a.M ? m : a.M - 1;
        xsrc[j] = a.X + (size_t)m * a.K + q * 8;
    }
    const uint4* wsrc = a.Wp + ((size_t)(nt0 + wave) * KB) * 64 + lane;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
    auto issue = [&](int kb) {                                                 // PP 1 KiB pieces per wave and stage
        const unsigned base = lds0 + (unsigned)(((kb % NS) * STAGE + wave * 64) * 16);
        glds16(xsrc[0] + kb * 32, base);
        glds16(xsrc[1] + kb * 32, base + 256 * 16);
#pragma unroll
        for (int j = 0; j < NTW / 4; ++j) glds16(wsrc + ((size_t)(4 * j) * KB + kb) * 64, base + (512 + 256 * j) * 16);
    };
    f32x4 sg[4][WC], gr[4][WC], tot[4][NSEG > 4 ? WC : 1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int u = 0; u < WC; ++u) { sg[i][u] = (f32x4){0.f, 0.f, 0.f, 0.f}; gr[i][u] = sg[i][u]; if (NSEG > 4) tot[i][u] = sg[i][u]; }

#pragma unroll
    for (int k0 = 0; k0 < AHEAD; ++k0) if (k0 < KB) issue(k0);
    int kin = 0, seg = 0;
    for (int kb = 0; kb < KB; ++kb) {
        // stage kb has landed once every wave has seen its own three pieces of it: vmcnt retires in issue order, the pieces of the
        // (up to two) younger stages may still fly.  lgkmcnt(0): this wave's fragment reads of the previous step are back, so the
        // buffer that is refilled below is free.  A raw barrier: __syncthreads() would drain the DMA queue.
        const int younger = KB - 1 - kb < AHEAD - 1 ? KB - 1 - kb : AHEAD - 1;      // stages behind this one that may still be in flight
        switch (younger) {
            case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PP) : "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PP) : "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * PP) : "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * PP) : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(5 * PP) : "memory"); break;
        }
        __builtin_amdgcn_s_barrier();
        const uint4* As = ring + (kb % NS) * STAGE;
        const uint4* Bs = As + 512;
        uint4 af[4], bf[WC];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = As[pgemm_a_pos(wr * 64 + i * 16 + (lane & 15), lane >> 4)];
#pragma unroll
        for (int u = 0; u < WC; ++u) bf[u] = Bs[(wc * WC + u) * 64 + lane];
#ifndef T3_PGEMM_NODMA       // diagnostic builds only: which resource bounds the schedule
        if (kb + AHEAD < KB) issue(kb + AHEAD);          // into the buffer of step kb - 1: every wave is past its reads (barrier above)
#endif
#ifdef T3_PGEMM_NOMFMA
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < WC; ++u) { sg[i][u][0] += __uint_as_float(af[i].x ^ bf[u].y); sg[i][u][1] += __uint_as_float(af[i].z ^ bf[u].w); }
#else
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < WC; ++u)
                sg[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(bf[u]), as_frag(af[i]), sg[i][u], 0, 0, 0);      // D = W X^T: see pgemm_store4
#endif
        if (++kin == kbs) {                      // segment complete: fold it
            const bool first_in_group = (seg & 3) == 0, last_in_group = (seg & 3) == 3;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int u = 0; u < WC; ++u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gr[i][u][r] = first_in_group ? sg[i][u][r] : gr[i][u][r] + sg[i][u][r];
                        if constexpr (NSEG > 4) { if (last_in_group) tot[i][u][r] = seg == 3 ? gr[i][u][r] : tot[i][u][r] + gr[i][u][r]; }
                        sg[i][u][r] = 0.0f;
                    }
                }
            kin = 0; ++seg;
        }
    }
    // epilogue: D[n = 4 (lane >> 4) + r][m = lane & 15] of every 16 x 16 tile (pgemm_store4)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 64 + i * 16 + (lane & 15);
        float rs = 1.0f;
        if constexpr (NORM) rs = rstd[m < a.M ? m : a.M - 1];
#pragma unroll
        for (int u = 0; u < WC; u += (EPI == EPI_SILU ? 2 : 1)) {
            f32x4 v, vu;
            if constexpr (NSEG > 4) { v = tot[i][u]; vu = tot[i][EPI == EPI_SILU ? u + 1 : u]; } else { v = gr[i][u]; vu = gr[i][EPI == EPI_SILU ? u + 1 : u]; }
            pgemm_store4<EPI>(a, m, nt0 + wc * WC + u, lane >> 4, v, vu, rs);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pgemm2_kernel: the prefill schedule at 256 x 128 workgroup tiles (8 waves as 4 x 2, a wave tile of 64 rows x 64 columns = 4 x 4 MFMA
// tiles), K staged 64 deep.  What bounds pgemm_kernel is what a CU can take in (L2 -> LDS, ~70 GB/s per CU, MI355X_MICROARCH.md "Indexed
// rows"): a workgroup tile of BM x BN stages (BM + BN) x 64 B per 32-deep K step for BM x BN / 256 MFMAs, and at 128 x 64 (four workgroups
// per CU, nothing shared between them) that is 384 B per MFMA against the ~116 B per MFMA a CU can take in at the full matrix rate:
// a ceiling of ~30 % (measured: 26 % MFMA busy on gate/up at 8 178 rows).  256 x 128 stages 192 B per MFMA (ceiling ~60 %), reads an
// LDS fragment for two MFMAs instead of 1.3, and meets half as many barriers per K.  MEASURED (round 4): the same time per launch as
// pgemm_kernel within +-5 % except down_proj (-20 % at 8 178 rows): a stage iteration takes ~1.8 us for 0.2 us of MFMA time per wave --
// one workgroup per CU with every wave both issuing DMA pieces (~150 cycles each) and computing, re-synchronised by a barrier per stage,
// leaves the matrix pipe idle three quarters of the time whatever the tile.  What is missing is a loader / consumer split (the guide's
// ring-gemm), not a bigger tile; launch_pgemm takes this kernel only where it measured faster.  Same numbers: per output one MFMA chain per segment
// from +0 in ascending k, segments folded left to right in groups of four, groups left to right (two / three accumulator sets).
// Stage = A image 256 rows x 128 B (XOR-swizzled through the DMA source address: 16-byte chunk ch of row r sits at position
// ch ^ ((r >> 1) & 7), which makes the 16 lanes of every ds_read_b128 lane group hit 16 different bank slots) | 8 packed n-tiles x 2
// k-blocks x 1 KiB.  Ring of 3 stages (144 KiB), filled by LDS-DMA two stages ahead, 6 pieces per wave and stage, counted vmcnt, raw barriers.
// ------------------------------------------------------------------------------------------------
template <int EPI, int NSEG, bool NORM>
__global__ __launch_bounds__(NSEG == 4 ? 768 : 512) void pgemm2_kernel(GemmArgs a, const float* rstd) {
    // LOADERS (the 4-segment forms: gate/up, qkv; 176 registers, so twelve waves fit): four more waves do nothing but feed the ring -- twelve
    // DMA pieces per stage each, then the wait for the next stage's pieces and the stage's barrier -- and the eight MFMA waves never issue a
    // load.  With every wave doing both (the 16-segment forms below still do: 240 registers) a stage was issue 0.24-0.40 us + MFMAs
    // 0.40-0.55 + barrier skew 0.25 one after the other, 1.8 us for 48 KiB = 2.7 x what the CU's load path needs (profiles/NOTES.md).
    constexpr bool LOADERS = NSEG == 4;
    constexpr int NS = 3, AHEAD = NS - 1, PP = 6;                 // ring stages; stages in flight; DMA pieces per wave and stage
    constexpr int STAGE = 2048 + 1024;                            // uint4 per stage: A image 32 KiB | B 16 KiB
    extern __shared__ __attribute__((aligned(16))) uint4 ring2[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;                     // wave tile: rows 64 wm .., packed n-tiles 4 wn .. 4 wn + 3
    const int KB = a.K >> 5, NST = KB >> 1;                      // 32-deep k-blocks; 64-deep stages
    const int spseg = (KB / NSEG) >> 1;                          // stages per segment (segments are 64 or 256 deep: 1 or 4)
    int tile_x, tile_y;
    pgemm_tile_map((int)gridDim.x, (int)gridDim.y, tile_x, tile_y);
    const int m0 = tile_y * 256, nt0 = tile_x * 8;
    // DMA sources.  A: piece j (of 32) = image rows 8 j .. 8 j + 7; lane l fills position p = l & 7 of row 8 j + (l >> 3), i.e. fetches
    // the row's chunk p ^ ((row >> 1) & 7): the 8 lanes of a row fetch one whole 128-byte line.  A wave takes pieces 4 wave .. 4 wave + 3.
    // B: piece (tile u, k-block kk) = 1 KiB of the packed matrix as it lies; wave w takes tile w, both k-blocks.
    const uint16_t* xsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (4 * wave + j) + (lane >> 3);
        int m = m0 + row; m = m < a.M ? m : a.M - 1;
        xsrc[j] = a.X + (size_t)m * a.K + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring2;
    if constexpr (LOADERS) {
        if (wave >= 8) {
            // loader wave lw: A pieces 8 lw .. 8 lw + 7 (image rows 64 lw .. 64 lw + 63), B tiles 2 lw, 2 lw + 1 (both k-blocks)
            const int lw = wave - 8;
            const uint16_t* xs_[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = 8 * (8 * lw + j) + (lane >> 3);
                int m = m0 + row; m = m < a.M ? m : a.M - 1;
                xs_[j] = a.X + (size_t)m * a.K + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
            }
            const uint4* ws_ = a.Wp + ((size_t)(nt0 + 2 * lw) * KB) * 64 + lane;
            auto feed = [&](int st) {
                const unsigned base = lds0 + (unsigned)((st % NS) * STAGE * 16);
#pragma unroll
                for (int j = 0; j < 8; ++j) glds16(xs_[j] + st * 64, base + (unsigned)((8 * lw + j) * 1024));
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
                        glds16(ws_ + ((size_t)u * KB + 2 * st + kk) * 64, base + (unsigned)(2048 * 16 + ((2 * lw + u) * 2 + kk) * 1024));
            };
#pragma unroll
            for (int k0 = 0; k0 < AHEAD; ++k0) if (k0 < NST) feed(k0);
            for (int st = 0; st < NST; ++st) {
                // stage st has landed (its twelve pieces of this wave; the next stage's twelve may still fly) before the barrier lets the MFMA
                // waves onto it; behind the barrier every MFMA wave is past its reads of stage st - 1, whose buffer stage st + 2 refills
                if (st + 1 < NST) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#ifndef T3_PG2_NOFEED
                if (st + AHEAD < NST) feed(st + AHEAD);
#endif
            }
            return;
        }
    }
    const uint4* wsrc = a.Wp + ((size_t)(nt0 + wave) * KB) * 64 + lane;
    auto issue = [&](int st) {
        const unsigned base = lds0 + (unsigned)((st % NS) * STAGE * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(xsrc[j] + st * 64, base + (unsigned)((4 * wave + j) * 1024));
        glds16(wsrc + (size_t)(2 * st) * 64, base + (unsigned)(2048 * 16 + (wave * 2) * 1024));
        glds16(wsrc + (size_t)(2 * st + 1) * 64, base + (unsigned)(2048 * 16 + (wave * 2 + 1) * 1024));
    };
    f32x4 sg[4][4], gr[4][4], tot[4][NSEG > 4 ? 4 : 1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int u = 0; u < 4; ++u) { sg[i][u] = (f32x4){0.f, 0.f, 0.f, 0.f}; gr[i][u] = sg[i][u]; if (NSEG > 4) tot[i][u] = sg[i][u]; }
    if constexpr (!LOADERS) {
#pragma unroll
        for (int k0 = 0; k0 < AHEAD; ++k0) if (k0 < NST) issue(k0);
    }
    const int c = lane & 15, q = lane >> 4;
    int kin = 0, seg = 0;
    for (int st = 0; st < NST; ++st) {
        // stage st has landed once every wave has seen its own six pieces of it (vmcnt retires in issue order; the pieces of the next
        // stage may still fly).  lgkmcnt(0): this wave's fragment reads of the previous stage are back, so the buffer refilled below is
        // free once every wave is past the barrier.  A raw barrier: __syncthreads() would drain the DMA queue.
        if constexpr (LOADERS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else { if (st + 1 < NST) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PP) : "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        const uint4* As = ring2 + (st % NS) * STAGE;
        const uint4* Bs = As + 2048;
        // Issuing a stage's six DMA pieces costs a wave 0.24-0.40 us (the CU's load path takes 64 B per clock: 48 KiB per stage) and its 32 MFMAs
        // 0.4-0.5 us (stamps, profiles/r04_r_pgemm2_stamps.txt); with every wave doing one after the other in the same order, the load path
        // idles while the matrix pipes run and the other way round.  Waves w and w + 4 share a SIMD: the first four issue, then compute;
        // the other four compute, then issue (into the buffer of stage st - 1: every wave is past its reads of it, barrier above).
        const bool issue_first = wave < 4;
        if (!LOADERS && issue_first && st + AHEAD < NST) issue(st + AHEAD);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // The weight fragments of the k-block and ONE activation fragment ahead of the MFMAs that use it: with all eight fragments asked for
            // at once the 4-segment forms (168 registers: twelve waves) had none left, and hipcc read the second k-block's activation fragments
            // one by one, each behind a full lgkmcnt(0) in front of its four MFMAs (six exposed LDS round trips per stage).
            uint4 bf[4], acur, anxt;
            auto read_a = [&](int i) { const int row = wm * 64 + i * 16 + c; return As[row * 8 + ((4 * kk + q) ^ ((row >> 1) & 7))]; };
#ifdef T3_PG2_NOREAD      // ablation (tools/): no fragment reads
#pragma unroll
            for (int u = 0; u < 4; ++u) bf[u] = make_uint4(u, lane, st, kk);
            acur = make_uint4(lane, kk, 0, st);
#else
#pragma unroll
            for (int u = 0; u < 4; ++u) bf[u] = Bs[((wn * 4 + u) * 2 + kk) * 64 + lane];
            acur = read_a(0);
#endif
            const bool fresh = kk == 0 && kin == 0;   // a segment's chain starts from +0: the accumulator operand is a zero tuple, nothing is cleared
            const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#ifndef T3_PG2_NOREAD
                if (i < 3) anxt = read_a(i + 1);
#endif
#ifdef T3_PG2_NOMFMA      // ablation (tools/): the fragments are read and dropped
                asm volatile("" :: "v"(acur.x), "v"(acur.w), "v"(bf[i].x), "v"(bf[i].w));
#else
                if (fresh) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) sg[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(bf[u]), as_frag(acur), zero, 0, 0, 0);      // D = W X^T: see pgemm_store4
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) sg[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(bf[u]), as_frag(acur), sg[i][u], 0, 0, 0);
                }
#endif
                acur = anxt;
            }
        }
        if (!LOADERS && !issue_first && st + AHEAD < NST) issue(st + AHEAD);
        if (++kin == spseg) {                    // segment complete: fold it
            const bool first_in_group = (seg & 3) == 0, last_in_group = (seg & 3) == 3;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gr[i][u][r] = first_in_group ? sg[i][u][r] : gr[i][u][r] + sg[i][u][r];
                        if constexpr (NSEG > 4) { if (last_in_group) tot[i][u][r] = seg == 3 ? gr[i][u][r] : tot[i][u][r] + gr[i][u][r]; }
                    }
                }
            kin = 0; ++seg;
        }
    }
    // epilogue: D[n = 4 (lane >> 4) + r][m = lane & 15] of every 16 x 16 tile (pgemm_store4)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + c;
        float rs = 1.0f;
        if constexpr (NORM) rs = rstd[m < a.M ? m : a.M - 1];
#pragma unroll
        for (int u = 0; u < 4; u += (EPI == EPI_SILU ? 2 : 1)) {
            f32x4 v, vu;
            if constexpr (NSEG > 4) { v = tot[i][u]; vu = tot[i][EPI == EPI_SILU ? u + 1 : u]; } else { v = gr[i][u]; vu = gr[i][EPI == EPI_SILU ? u + 1 : u]; }
            pgemm_store4<EPI>(a, m, nt0 + wn * 4 + u, q, v, vu, rs);
        }
    }
}
template <int EPI, int NSEG, bool NORM>
static hipError_t launch_pgemm2_t(const GemmArgs* a, const float* rs, dim3 grid, hipStream_t s) {
    constexpr size_t lds = (size_t)3 * 3072 * 16;
    auto kern = pgemm2_kernel<EPI, NSEG, NORM>;
    static bool raised[MAX_DEVICES] = {};
    if (!raised[cur_device()]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[cur_device()] = true;
    }
    if (!a) return hipSuccess;
    hipLaunchKernelGGL(kern, grid, dim3(NSEG == 4 ? 768 : 512), lds, s, *a, rs);
    return hipGetLastError();
}

static int g_pgemm_min_rows = -1, g_pgemm_wide_rows = -1;
static int g_pgemm2_min_wgs = 192, g_pgemm2_all = 0;
static int g_gemm_pipe_head = 1;         // T3_GEMM_PIPE_HEAD=0: the speech head keeps the one-shot form at every row count
static int g_gemm_down2_min = 113;      // T3_GEMM_DOWN2_MIN_ROWS
static int g_gemm_pad_gx = 1;      // T3_GEMM_PAD_GX=0: no padding of a decode GEMM's grid to whole XCD rounds (launch_gemm2_av)
static int g_gemm_pipe = 1, g_gemm_head_2percu = 1, g_gemm_pipe_qkv_min = 129, g_gemm_pipe_min = 49;      // T3_GEMM_PIPE / T3_GEMM_HEAD_2PERCU / T3_GEMM_PIPE_QKV_MIN_ROWS (measurement switches, re-read with the next one)
static int g_gemm_small_m = 1;          // T3_GEMM_SMALL_M=0: the one-tile GEMMs issue every activation-row load (read again by every prepare_kernels call, i.e. per engine)
void set_pgemm_min_rows(int rows) { g_pgemm_min_rows = rows; }
void set_pgemm_wide_rows(int rows) { g_pgemm_wide_rows = rows; }

// Large-M path of launch_gemm: returns hipErrorNotSupported when the shape is not one of the layer forms.
static hipError_t launch_pgemm(const GemmArgs& a, int epi, hipStream_t s) {
    const bool norm = a.norm != 0;
    const int nseg = a.nw == 16 ? 16 : 4;
    const int ntiles = (a.N + 15) / 16 * (epi == EPI_SILU ? 2 : 1);
    if (a.row_index || ntiles % 4 || a.K % (32 * nseg) || (norm && (!a.rstd_scratch || a.K != D))) return hipErrorNotSupported;
    // (round 2: 128 x 128 tiles measured slower than 128 x 64 in plain row-major tile order, 18.76 against 18.35 ms per 8 178-row step)
    // 256 x 128 tiles (pgemm2_kernel) where they still cover the chip (>= g_pgemm2_min_wgs workgroups; T3_PGEMM2_MIN_WGS, 0 = never) AND
    // where they measured faster (tools/gemm_bench, us per launch at 1 024 | 2 048 | 8 178 rows, 128 x 64 -> 256 x 128, profiles/
    // r04_l_prefill_pgemm2_vs_pgemm.txt): down 34.1 -> 34.1 | 54.6 -> 53.0 | 125.3 -> 100.5, gate/up 42.8 -> 38.7 | 70.8 -> 67.8 | 224.1 -> 236.4,
    // qkv 23.4 | 34.5 -> 37.6 | 93.1 -> 97.4, o 11.5 | 24.8 | 56.7 -> 58.3.  With the XCD-aware tile order, the stage's DMA issue and MFMAs in
    // antiphase between the wave halves and no accumulator clearing (profiles/r04_t_pgemm_xcd_map.txt, 8 178 rows): down 84.5 (128 x 64: 119.0),
    // o 49.5 (54.5), gate/up 230 (201), qkv 93.3 (91.6): the 16-segment forms always, gate/up below 4 096 rows (T3_PGEMM2_ALL=1: every form)
    // With four loader waves feeding the ring of the 4-segment forms (8 178 rows, profiles/r04_as_pgemm2_loaders.txt): gate/up 236 -> 201 (128 x 128
    // tiles: 196), qkv 97.4 -> 72.2 (77.9); at 2 048 rows qkv 28.4 (31.1), at 4 096 46.9 (46.1): qkv joins the 256 x 128 forms.
    // One activation fragment ahead of its MFMAs instead of eight fragments at once (r04_au_*): gate/up 196.8 at 8 178 rows (128 x 128: 196.6),
    // 104.6 (108.3) at 4 096: gate/up at every row count too.
    const bool pg2_form = g_pgemm2_all || nseg == 16 || epi == EPI_SILU || (epi == EPI_BF16 && norm);
    if (pg2_form && g_pgemm2_min_wgs > 0 && ntiles % 8 == 0 && (a.K & 63) == 0 && (long)(ntiles / 8) * ((a.M + 255) / 256) >= g_pgemm2_min_wgs) {
        const dim3 grid2(ntiles / 8, (a.M + 255) / 256);
        if (norm) {
            hipLaunchKernelGGL(row_rstd_kernel, dim3((a.M + 63) / 64), dim3(256), 0, s, a.X, a.rstd_scratch, a.M);
            if (epi == EPI_BF16) return launch_pgemm2_t<EPI_BF16, 4, true>(&a, a.rstd_scratch, grid2, s);
            if (epi == EPI_F32) return launch_pgemm2_t<EPI_F32, 4, true>(&a, a.rstd_scratch, grid2, s);
            if (epi == EPI_SILU) return launch_pgemm2_t<EPI_SILU, 4, true>(&a, a.rstd_scratch, grid2, s);
            return hipErrorNotSupported;
        }
        if (epi == EPI_RESID && nseg == 16) return launch_pgemm2_t<EPI_RESID, 16, false>(&a, nullptr, grid2, s);
        if (epi == EPI_F32 && nseg == 16) return launch_pgemm2_t<EPI_F32, 16, false>(&a, nullptr, grid2, s);
        if (epi == EPI_F32) return launch_pgemm2_t<EPI_F32, 4, false>(&a, nullptr, grid2, s);
        return hipErrorNotSupported;
    }
    // 128 x 128 tiles for the 4-segment forms from 4 096 rows on (round 4, with the XCD-aware tile order: gate/up 204 -> 198 us, qkv 84.7 ->
    // 79.8 at 8 178 rows, nothing at 2 048: profiles/r04_v_pgemm_128x128.txt); T3_PGEMM_WC=2: never, =4: from 2 048 rows
    static int wc_env = -1;
    if (wc_env < 0) { const char* e = getenv("T3_PGEMM_WC"); wc_env = e ? atoi(e) : 0; }
    const int wide_rows = g_pgemm_wide_rows >= 0 ? g_pgemm_wide_rows : (wc_env == 4 ? 2048 : 4096);          // 0 = never
    const bool wide = wc_env != 2 && nseg == 4 && ntiles % 8 == 0 && wide_rows > 0 && a.M >= wide_rows;
    const dim3 grid(ntiles / (wide ? 8 : 4), (a.M + 127) / 128);
#define T3_PG(E, SEG, NRM, RS) do { if (wide) hipLaunchKernelGGL((pgemm_kernel<E, SEG, NRM, 4>), grid, dim3(256), 0, s, a, (const float*)(RS)); \
                                    else hipLaunchKernelGGL((pgemm_kernel<E, SEG, NRM, 2>), grid, dim3(256), 0, s, a, (const float*)(RS)); } while (0)
    if (norm) {
        hipLaunchKernelGGL(row_rstd_kernel, dim3((a.M + 63) / 64), dim3(256), 0, s, a.X, a.rstd_scratch, a.M);
        if (epi == EPI_BF16) T3_PG(EPI_BF16, 4, true, a.rstd_scratch);
        else if (epi == EPI_F32) T3_PG(EPI_F32, 4, true, a.rstd_scratch);
        else if (epi == EPI_SILU) T3_PG(EPI_SILU, 4, true, a.rstd_scratch);
        else return hipErrorNotSupported;
    } else {
        if (epi == EPI_RESID && nseg == 16) hipLaunchKernelGGL((pgemm_kernel<EPI_RESID, 16, false, 2>), grid, dim3(256), 0, s, a, (const float*)nullptr);
        else if (epi == EPI_F32 && nseg == 16) hipLaunchKernelGGL((pgemm_kernel<EPI_F32, 16, false, 2>), grid, dim3(256), 0, s, a, (const float*)nullptr);
        else if (epi == EPI_F32) T3_PG(EPI_F32, 4, false, nullptr);
        else return hipErrorNotSupported;
    }
#undef T3_PG
    return hipGetLastError();
}

int choose_mt(int M, int ntiles_x, int nw, bool norm) {
    const int mtiles = (M + 15) / 16;
    if (const char* e = getenv("T3_GEMM_MT")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) return v; }
    // Largest row tile that (a) fits the register file / LDS (NORM form: 2 m-tiles; 16-wave form: 4) and (b) still launches
    // enough workgroups: >= 512 for the 4-wave forms (measured: qkv is fastest at 768 workgroups, gate/up at 512),
    // >= 256 for the 16-wave form.  Workgroups with the same blockIdx.x differ by a multiple of gridDim.x in linear id,
    // and gridDim.x is a multiple of 8 for the layer GEMMs, so they land on the same XCD and share the weight tile in L2.
    int cap = norm ? 2 : (nw == 16 ? 4 : 8);
    if (norm) { if (const char* e = getenv("T3_GEMM_MT_NORM")) cap = atoi(e) >= 2 ? 2 : 1; }
    const long want = nw == 16 ? 256 : 512;
    int best = 1;
    for (int mt = 1; mt <= cap; mt <<= 1) {
        if (mt > 1 && mt / 2 >= mtiles) break;
        const long wgs = (long)ntiles_x * ((mtiles + mt - 1) / mt);
        if (mt == 1 || wgs >= want) best = mt;
    }
    return best;
}

// gemm2_kernel launcher; a == nullptr: only raise the kernel's dynamic-LDS limit (prepare_kernels, before any stream capture)
template <int MT, int NT, int EPI, int NW, int KBS, bool NORM, int AV, int EWV = gemm2_ew<NW>()>
static hipError_t launch_gemm2_av(const GemmArgs* a, hipStream_t s) {
    constexpr size_t lds = (size_t)NW * MT * KBS * 1024 + (NORM ? (size_t)NW * MT * 16 * sizeof(float) : 0) + (EWV ? 256 : 0);    // + the prefetch dump corner
    auto kern = gemm2_kernel<MT, NT, EPI, NW, KBS, NORM, AV, EWV>;
    static bool raised[MAX_DEVICES] = {};
    if (lds > 64 * 1024 && !raised[cur_device()]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[cur_device()] = true;
    }
    if (!a) return hipSuccess;
    const int ntiles = a->packed_tiles > 0 ? a->packed_tiles / (EPI == EPI_SILU ? 2 : 1) : (a->N + 15) / 16;
    const int gx = (EPI == EPI_SILU) ? (ntiles + NT / 2 - 1) / (NT / 2) : (ntiles + NT - 1) / NT;
    const int gy = ((a->M + 15) / 16 + MT - 1) / MT;
    // Workgroup L = y * gx + x runs on XCD L % 8, and the gy workgroups of one tile group x read the SAME weight tiles: with gx a multiple of
    // 8 they share one XCD's L2 (gate/up at 64 rows: 128 x 2), with any other gx every one of them fetches the tiles into another L2 -- the
    // speech head (129 tile groups) pulled its 16.8 MB twice at 64 rows and eight times at 256 (11.6 us against gate/up's 7.1 for the same
    // bytes).  The grid is padded to the next multiple of 8; the extra workgroups leave at once.
    GemmArgs b = *a;
    int gxl = gx;
    if (g_gemm_pad_gx && gy > 1 && (gx & 7) && gx > 8) { gxl = (gx + 7) & ~7; b.gx_real = gx; }
    launch_k(kern, dim3(gxl, gy), dim3((NW + EWV) * 64), lds, s, b);
    return hipGetLastError();
}
// picks AV (see gemm2_kernel): the fewest A-row instructions that cover the rows of a one-tile call
template <int MT, int NT, int EPI, int NW, int KBS, bool NORM>
static hipError_t launch_gemm2_t(const GemmArgs* a, hipStream_t s) {
    if constexpr (MT == 1) {
        constexpr int RPI = 64 / (KBS * 4);            // rows per A instruction: 2 (KBS 8) or 8 (KBS 2)
        const int small = g_gemm_small_m;
        if (!a) {
            hipError_t e;
            if ((e = launch_gemm2_av<MT, NT, EPI, NW, KBS, NORM, 1>(a, s)) != hipSuccess) return e;
            if constexpr (KBS == 8) {
                if ((e = launch_gemm2_av<MT, NT, EPI, NW, KBS, NORM, 2>(a, s)) != hipSuccess) return e;
                if ((e = launch_gemm2_av<MT, NT, EPI, NW, KBS, NORM, 4>(a, s)) != hipSuccess) return e;
            }
        } else if (small && a->M <= 16) {
            if (a->M <= RPI) return launch_gemm2_av<MT, NT, EPI, NW, KBS, NORM, 1>(a, s);
            if constexpr (KBS == 8) {
                if (a->M <= 2 * RPI) return launch_gemm2_av<MT, NT, EPI, NW, KBS, NORM, 2>(a, s);
                if (a->M <= 4 * RPI) return launch_gemm2_av<MT, NT, EPI, NW, KBS, NORM, 4>(a, s);
            }
        }
    }
    return launch_gemm2_av<MT, NT, EPI, NW, KBS, NORM, KBS>(a, s);
}
static hipError_t launch_gemm2_pipe16(const GemmArgs* a, hipStream_t s) {
    constexpr size_t lds = (size_t)3 * 4 * 8192 + (size_t)2 * 2 * 4 * 2048 + (size_t)2 * 4 * 16 * sizeof(float) + 256;
    static bool raised[MAX_DEVICES] = {};
    if (!raised[cur_device()]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm2_pipe16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[cur_device()] = true;
    }
    if (!a) return hipSuccess;
    static int split_env = -1;
    if (split_env < 0) { const char* e = getenv("T3_GEMM_PIPE_SPLIT"); split_env = e ? atoi(e) : 0; }
    const int mgroups = (a->M + 15) / 16, gx = a->N / 32;
    int gy = split_env > 0 ? split_env : std::max(1, 256 / gx);
    while (gy > 1 && mgroups / gy < 2) --gy;
    launch_k(gemm2_pipe16_kernel, dim3(gx, gy), dim3(1024), lds, s, *a);
    return hipGetLastError();
}
template <int NT, int EPI>
static hipError_t launch_gemm2_pipe(const GemmArgs* a, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * 4 * 8192 + (size_t)2 * 4 * NT * 1024 + (size_t)2 * 4 * 16 * sizeof(float);
    auto kern = gemm2_pipe_kernel<NT, EPI>;
    static bool raised[MAX_DEVICES] = {};
    if (!raised[cur_device()]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[cur_device()] = true;
    }
    if (!a) return hipSuccess;
    static int split_env = -1;
    if (split_env < 0) { const char* e = getenv("T3_GEMM_PIPE_SPLIT"); split_env = e ? atoi(e) : 0; }
    const int mgroups = (a->M + 15) / 16;
    const int gx = (a->packed_tiles > 0 ? a->packed_tiles : a->N / 16 * (EPI == EPI_SILU ? 2 : 1)) / NT;      // (the speech head: 516 packed tiles, the last ones beyond N)
    int gy = split_env > 0 ? split_env : std::max(1, 256 / gx);    // workgroups per n-group: one workgroup per CU (gate/up 128 x 2, qkv 64 x 4, speech head 172 x 1)
    while (gy > 1 && mgroups / gy < 2) --gy;
    launch_k(kern, dim3(gx, gy), dim3(512), lds, s, *a);
    return hipGetLastError();
}
// looped NORM form (>= 4 m-groups of 32 rows, no row gather): one workgroup per n-group, weights stationary in registers
template <int MT, int NT, int EPI, int NW, int KBS, bool NORM>
static hipError_t launch_gemm2_loop_t(const GemmArgs* a, hipStream_t s) {
    constexpr size_t lds = (size_t)NW * MT * KBS * 1024 + (NORM ? (size_t)NW * MT * 16 * sizeof(float) : 0);
    auto kern = gemm2_loop_kernel<MT, NT, EPI, NW, KBS, NORM>;
    static bool raised[MAX_DEVICES] = {};
    if (lds > 64 * 1024 && !raised[cur_device()]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[cur_device()] = true;
    }
    if (!a) return hipSuccess;
    const int ntiles = (a->N + 15) / 16;
    const int gx = (EPI == EPI_SILU) ? (ntiles + NT / 2 - 1) / (NT / 2) : (ntiles + NT - 1) / NT;
    const int mgroups = ((a->M + 15) / 16 + MT - 1) / MT;
    static int split_env = -1;
    if (split_env < 0) { const char* e = getenv(NW == 16 ? "T3_GEMM_LOOP16_SPLIT" : "T3_GEMM_LOOP_SPLIT"); split_env = e ? atoi(e) : 0; }
    int gy = split_env > 0 ? split_env : (NW == 16 ? (NT == 2 ? 8 : 4) : 2);
    while (gy > 1 && mgroups / gy < (NW == 16 ? 1 : 2)) --gy;      // 4 waves: every workgroup walks at least two groups (16 waves: one is enough to win, measured)
    launch_k(kern, dim3(gx, gy), dim3(NW * 64), lds, s, *a);
    return hipGetLastError();
}
static hipError_t launch_gemm2_down2(const GemmArgs* a, hipStream_t s) {
    constexpr size_t lds = (size_t)16 * 8192 + (size_t)16 * 2 * 1024;       // 160 KiB: all of a CU's LDS
    static bool raised[MAX_DEVICES] = {};
    if (!raised[cur_device()]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm2_down2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        raised[cur_device()] = true;
    }
    if (!a) return hipSuccess;
    static int split_env = -1;
    if (split_env < 0) { const char* e = getenv("T3_GEMM_DOWN2_SPLIT"); split_env = e ? atoi(e) : 0; }
    const int mgroups = (a->M + 15) / 16, gx = a->N / 32;
    int gy = split_env > 0 ? split_env : std::max(1, 256 / gx);
    while (gy > 1 && mgroups / gy < 1) --gy;
    launch_k(gemm2_down2_kernel, dim3(gx, gy), dim3(1024), lds, s, *a);
    return hipGetLastError();
}
// NORM forms (4 waves, K = 1024): MT in {1, 2}, NT in {1, 2, 3, 4}
static hipError_t launch_gemm2_norm(const GemmArgs* a, int epi, int mt, int nt, hipStream_t s) {
#define T3_G2(E, MTV, NTV) return launch_gemm2_t<MTV, NTV, E, 4, 8, true>(a, s)
#define T3_G2_NT(E, MTV) switch (nt) { case 1: T3_G2(E, MTV, 1); case 2: T3_G2(E, MTV, 2); case 3: T3_G2(E, MTV, 3); default: T3_G2(E, MTV, 4); }
    if (epi == EPI_F32) { if (mt >= 2) T3_G2(EPI_F32, 2, 1); else T3_G2(EPI_F32, 1, 1); }
    if (epi == EPI_BF16) { if (mt >= 2) { T3_G2_NT(EPI_BF16, 2) } else { T3_G2_NT(EPI_BF16, 1) } }
    if (epi == EPI_SILU) {
        if (mt >= 2) { if (nt == 4) T3_G2(EPI_SILU, 2, 4); else T3_G2(EPI_SILU, 2, 2); }
        else { if (nt == 4) T3_G2(EPI_SILU, 1, 4); else T3_G2(EPI_SILU, 1, 2); }
    }
#undef T3_G2_NT
#undef T3_G2
    return hipErrorInvalidValue;
}
// 16-segment forms at one m-tile per workgroup: K = 1024 (o_proj, 64-wide segments) or K = 4096 (down_proj, 256-wide segments)
static hipError_t launch_gemm2_16(const GemmArgs* a, int epi, int kbs, hipStream_t s) {
    if (epi == EPI_RESID) return kbs == 2 ? launch_gemm2_t<1, 1, EPI_RESID, 16, 2, false>(a, s) : launch_gemm2_t<1, 1, EPI_RESID, 16, 8, false>(a, s);
    if (epi == EPI_F32) return kbs == 2 ? launch_gemm2_t<1, 1, EPI_F32, 16, 2, false>(a, s) : launch_gemm2_t<1, 1, EPI_F32, 16, 8, false>(a, s);
    return hipErrorInvalidValue;
}
void gemm_refresh_switches() {
    auto rd = [](const char* name, int dflt) { const char* ev = getenv(name); return ev ? atoi(ev) : dflt; };
    g_gemm_small_m = rd("T3_GEMM_SMALL_M", 1); g_gemm_pipe = rd("T3_GEMM_PIPE", 1);
    g_pgemm2_min_wgs = rd("T3_PGEMM2_MIN_WGS", 192); g_pgemm2_all = rd("T3_PGEMM2_ALL", 0);
    g_gemm_pad_gx = rd("T3_GEMM_PAD_GX", 1); g_gemm_down2_min = rd("T3_GEMM_DOWN2_MIN_ROWS", 113); g_gemm_pipe_head = rd("T3_GEMM_PIPE_HEAD", 1);
    g_gemm_head_2percu = rd("T3_GEMM_HEAD_2PERCU", 1); g_gemm_pipe_qkv_min = rd("T3_GEMM_PIPE_QKV_MIN_ROWS", 129); g_gemm_pipe_min = rd("T3_GEMM_PIPE_MIN_ROWS", 49);
}
hipError_t prepare_gemm2() {
    hipError_t e;
    for (int epi : {EPI_F32, EPI_BF16, EPI_SILU})
        for (int mt = 1; mt <= 2; ++mt)
            for (int nt = 1; nt <= 4; ++nt)
                if ((e = launch_gemm2_norm(nullptr, epi, mt, nt, nullptr)) != hipSuccess) return e;
    for (int epi : {EPI_F32, EPI_RESID})
        for (int kbs : {2, 8})
            if ((e = launch_gemm2_16(nullptr, epi, kbs, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_av<2, 4, EPI_BF16, 4, 8, true, 8, 0>(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_t<1, 1, EPI_F32, 4, 8, false>(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_t<2, 1, EPI_F32, 4, 8, false>(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_loop_t<2, 1, EPI_BF16, 4, 8, true>(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_loop_t<2, 2, EPI_SILU, 4, 8, true>(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_loop_t<2, 4, EPI_SILU, 4, 8, true>(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_pipe<4, EPI_SILU>(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_pipe16(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_pgemm2_t<EPI_BF16, 4, true>(nullptr, nullptr, dim3(), nullptr)) != hipSuccess) return e;
    if ((e = launch_pgemm2_t<EPI_F32, 4, true>(nullptr, nullptr, dim3(), nullptr)) != hipSuccess) return e;
    if ((e = launch_pgemm2_t<EPI_SILU, 4, true>(nullptr, nullptr, dim3(), nullptr)) != hipSuccess) return e;
    if ((e = launch_pgemm2_t<EPI_RESID, 16, false>(nullptr, nullptr, dim3(), nullptr)) != hipSuccess) return e;
    if ((e = launch_pgemm2_t<EPI_F32, 16, false>(nullptr, nullptr, dim3(), nullptr)) != hipSuccess) return e;
    if ((e = launch_pgemm2_t<EPI_F32, 4, false>(nullptr, nullptr, dim3(), nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_pipe<3, EPI_BF16>(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_loop_t<1, 1, EPI_RESID, 16, 8, false>(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_down2(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_loop_t<1, 2, EPI_RESID, 16, 2, false>(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_loop_t<1, 1, EPI_RESID, 16, 2, false>(nullptr, nullptr)) != hipSuccess) return e;
    return hipSuccess;
}

// epi: GemmEpi; nw: 4 (qkv / gate-up / head form) or 16 (o_proj / down_proj form); a.norm: RMSNorm folded (needs K = 1024, nw = 4)
hipError_t launch_gemm(const GemmArgs& a, int epi, int mt, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    const int nw = a.nw == 16 ? 16 : 4;
    const bool norm = a.norm != 0;
    if (a.K % (32 * nw) != 0 || (norm && (a.K != D || nw != 4))) return hipErrorInvalidValue;
    {
        // rows from which the LDS-tiled schedule takes over (0 = never).  1024: a 256-row call is a DECODE step of 128 utterances,
        // where 128 x 64 tiles leave 32-128 workgroups (measured on the continuous-batching run of tools/bench_serving.py:
        // 16.8 k tok/s with the switch at 256 rows, 25.0 k at 1024 or 2048).
        // Per form since the looped schedules exist (tools/chain_proto at 320-1023 rows, us per launch looped | LDS-tiled: gate/up
        // 24.9 | 25.8 at 384 rows, 30.6 | 28.6 at 512; qkv 16.8 | 22.0 at 512, 25.4 | 23.6 at 768; o 17.0 | 20.4 and down 34.1 | 46.5
        // at 1023: the 16-segment fold of a 128 x 64 tile is a fixed ~18 / ~40 us): -2 = these per-form switches.
        if (g_pgemm_min_rows == -1) { const char* e = getenv("T3_PGEMM_MIN_ROWS"); g_pgemm_min_rows = e ? atoi(e) : -2; }
        // Round 4 (tools/gemm_bench sweeps, profiles/r04_x_schedule_thresholds_*.txt): with gate/up and qkv pipelined through compute and
        // epilogue waves the weights-stationary forms hold their per-row cost far beyond the old switches -- gate/up 21.9 | 27.9 | 41.0 us at
        // 545 | 800 | 1 280 rows against 32.8 | 34.6 | 58.8 on the LDS-tiled schedule (which also pays a row-statistic launch), level at 2 048
        // (61.8 both); qkv level at ~1 100 (20.4 | 20.9 at 1 024, 24.3 | 23.5 at 1 280); o looped 13.2 | 19.9 at 1 280, level at 2 048; down
        // level from 1 280 to 1 600 -- and with two n-tiles per workgroup (gemm2_down2_kernel) 31.4 | 47.6 at 1 600 rows, 37.0 | 49.2 at 2 048,
        // 51.4 | 52.6 at 3 072, 58.7 | 56.2 at 3 584 (profiles/r04_aq_down2_vs_tiled.txt): down switches at 3 072.
        // The mixed steps of continuous batching (256 decode rows + a prompt: 300-1 100 rows) now stay on them.
        const int pg_min = g_pgemm_min_rows != -2 ? g_pgemm_min_rows
                         : epi == EPI_SILU ? 2048 : (nw == 4 ? (a.row_index ? 1024 : 1152) : (a.K == D ? 2048 : 3072));
        if (pg_min > 0 && a.M >= pg_min) {
            const hipError_t pe = launch_pgemm(a, epi, s);
            if (pe != hipErrorNotSupported) return pe;
        }
    }
    if (norm) {
        // n-tiles per workgroup.  Every workgroup re-reads its rows of the activation operand, so more n-tiles per workgroup divide
        // that traffic, as long as the grid still covers the chip: the largest tile group that leaves >= 256 workgroups (one per
        // CU: qkv at 64 rows takes groups of 3 = 256 workgroups rather than groups of 4 = 192), else the largest that leaves >= 192
        // (256 for gate/up).  A weight whose last tile is partial (the speech head: 513 tiles) takes part when its packed buffer
        // was padded to a multiple of the tile group (GemmArgs::packed_tiles).
        if (mt > 2) mt = 2;
        {
            // gate/up from ~130 rows on (decode steps of 65+ utterances, C4): a workgroup per (gate/up pair, half of the row groups)
            // walks its 32-row groups with the pair's weight tiles stationary in registers.  Measured at 256 rows: 25.8 -> 18.9 us
            // (two workgroups per n-group; one: 24.3, four: 21.0); at 128 rows 13.9 -> 13.3.  qkv loses with it (8.3 -> 12.7 us at 256
            // rows: 32 KiB of weights per workgroup do not pay for the walk) and keeps the one-workgroup-per-tile schedule.
            static int loop_min = -1;
            if (loop_min < 0) { const char* e = getenv("T3_GEMM_LOOP_MIN_ROWS"); loop_min = e ? atoi(e) : 81; }
            // two gate/up pairs per workgroup (the weights of 4 packed tiles = 128 registers stationary, 256 in all, no spill): the
            // rows pass through LDS once per 32 output columns instead of 16.  13.3 -> 10.5 us at 128 rows, 18.9 -> 16.6 at 256
            // (T3_GEMM_LOOP_NT=2: one pair)
            static int loop_nt = -1;
            if (loop_nt < 0) { const char* e = getenv("T3_GEMM_LOOP_NT"); loop_nt = e ? atoi(e) : 4; }
            // pipelined through compute and epilogue waves (gemm2_pipe_kernel; T3_GEMM_PIPE=0: the serial walk of gemm2_loop_kernel)
            const int pipe = g_gemm_pipe;
            if (pipe == 2 && loop_min > 0 && a.M >= loop_min && !a.row_index && a.N % 32 == 0 && epi == EPI_SILU) return launch_gemm2_pipe<4, EPI_SILU>(&a, s);     // 4 + 4 waves
            if (pipe && g_gemm_pipe_min > 0 && a.M >= g_gemm_pipe_min && !a.row_index && a.N % 32 == 0 && epi == EPI_SILU) return launch_gemm2_pipe16(&a, s);
            // qkv the same way from T3_GEMM_PIPE_QKV_MIN_ROWS rows on (3 n-tiles per workgroup: 64 x 4 workgroups)
            // ... and the speech head of a decode-only step of 65+ utterances (no row gather; 516 packed tiles = 172 groups of 3: the one-shot form
            // streamed the 16.8 MB of weights once per 32 rows, 28 us at 256 rows)
            if (pipe && g_gemm_pipe_qkv_min > 0 && a.M >= g_gemm_pipe_qkv_min && !a.row_index && epi == EPI_BF16 &&
                (a.packed_tiles == 0 ? a.N % 48 == 0 : (g_gemm_pipe_head && a.packed_tiles % 3 == 0))) return launch_gemm2_pipe<3, EPI_BF16>(&a, s);
            if (loop_min > 0 && a.M >= loop_min && !a.row_index && a.N % 32 == 0 && epi == EPI_SILU && loop_nt == 4) return launch_gemm2_loop_t<2, 4, EPI_SILU, 4, 8, true>(&a, s);
            if (loop_min > 0 && a.M >= loop_min && !a.row_index && a.N % 16 == 0 && epi == EPI_SILU) return launch_gemm2_loop_t<2, 2, EPI_SILU, 4, 8, true>(&a, s);
            if (loop_min > 0 && a.M >= loop_min && !a.row_index && a.N % 16 == 0 && epi == EPI_BF16 && getenv("T3_GEMM_LOOP_QKV")) return launch_gemm2_loop_t<2, 1, EPI_BF16, 4, 8, true>(&a, s);
        }
        int nt = epi == EPI_SILU ? 2 : 1;
        if (epi != EPI_F32 && (!a.row_index || a.packed_tiles > 0)) {
            static int force = -1;
            if (force < 0) { const char* e = getenv("T3_GEMM_NT"); force = e ? atoi(e) : 0; }
            const int ntiles = a.packed_tiles > 0 ? a.packed_tiles : (a.N + 15) / 16 * (epi == EPI_SILU ? 2 : 1);      // packed weight tiles
            const int groups = ((a.M + 15) / 16 + mt - 1) / mt;
            const int want = epi == EPI_SILU ? 256 : 192;
            int pick = 0;
            // a 2 x 4 workgroup holds 32 weight tiles + 16 activation pieces in registers: one workgroup per CU.  A grid a little over
            // 256 of those (the speech head at 64 rows: 129 x 2 = 258) would run a second, almost empty round: take the next group size
            auto partial_round = [&](int c) { const long w = (long)(ntiles / c) * groups; return mt * c >= 8 && w > 256 && w < 512; };
            for (int c = 4; c > nt && !pick; --c)
                if ((epi != EPI_SILU || c % 2 == 0) && ntiles % c == 0 && (long)(ntiles / c) * groups >= 256 && !partial_round(c)) pick = c;
            for (int c = 4; c > nt && !pick; --c)
                if ((epi != EPI_SILU || c % 2 == 0) && ntiles % c == 0 && (long)(ntiles / c) * groups >= want) pick = c;
            if (pick) nt = pick;
            if (force >= 1 && force <= 4 && (epi != EPI_SILU || force % 2 == 0) && ntiles % force == 0) nt = force;
            // a grid a little over one round of 2 x 4 workgroups (the speech head at 64 rows: 129 x 2 = 258): the 4-wave variant of that
            // form, two workgroups per CU, all of them resident at once (T3_GEMM_HEAD_2PERCU=0: the 2 x 3 form, 344 workgroups in 1.3 rounds)
            const int two_per_cu = g_gemm_head_2percu;
            if (two_per_cu && force == 0 && epi == EPI_BF16 && a.packed_tiles > 0 && mt == 2 && ntiles % 4 == 0 && partial_round(4))
                return launch_gemm2_av<2, 4, EPI_BF16, 4, 8, true, 8, 0>(&a, s);
        }
        return launch_gemm2_norm(&a, epi, mt, nt, s);
    }
    if (nw == 16 && (a.K == D || a.K == F) && a.N % 16 == 0 && !a.row_index && epi == EPI_RESID) {
        // o / down from 81 rows on (6+ m-tiles): a workgroup per (n-tile, quarter of the m-tiles) walks its m-tiles with the tile's
        // weights stationary in registers (T3_GEMM_LOOP16_MIN_ROWS; 0 = off).  Measured, us per launch old -> looped: down 10.9 -> 8.3
        // at 96 rows, 11.9 -> 8.5 at 128, 21.9 -> 10.2 at 192, 19.4 -> 11.9 at 256; o 7.5 -> 6.0 at 256; at 64 rows the old form wins
        static int loop16_min = -1;
        if (loop16_min < 0) { const char* e = getenv("T3_GEMM_LOOP16_MIN_ROWS"); loop16_min = e ? atoi(e) : 81; }
        {
            static int nt16 = -1;
            if (nt16 < 0) { const char* e = getenv("T3_GEMM_LOOP16_NT"); nt16 = e ? atoi(e) : 2; }
            // two n-tiles per workgroup and 8 workgroups per n-tile pair: o only (6.0 -> 4.7 us at 256 rows, 4.3 -> 3.7 at 128).  The down form (8 k-blocks per wave) would need 170 registers at 16 waves per
            // workgroup (128 available): hipcc spills, and a spilled destination of an in-flight asm load is a corrupted register
            // later (tests/test_build.py keeps every asm-load kernel at zero spills)
            static int loop16_min_o = -1;
            if (loop16_min_o < 0) { const char* e = getenv("T3_GEMM_LOOP16_MIN_ROWS_O"); loop16_min_o = e ? atoi(e) : (loop16_min > 0 ? 65 : 0); }     // o: 4.03 -> 3.74 us at 80 rows; at 64 rows the one-shot form wins (3.53 against 3.74)
            if (loop16_min_o > 0 && a.M >= loop16_min_o && nt16 == 2 && a.N % 32 == 0 && a.K == D)
                return launch_gemm2_loop_t<1, 2, EPI_RESID, 16, 2, false>(&a, s);
        }
        // down_proj with two n-tiles per workgroup and the rows by LDS-DMA (gemm2_down2_kernel) from T3_GEMM_DOWN2_MIN_ROWS rows on (0 = never)
        if (g_gemm_down2_min > 0 && a.M >= g_gemm_down2_min && a.K == F && a.N % 32 == 0) return launch_gemm2_down2(&a, s);
        if (loop16_min > 0 && a.M >= loop16_min)
            return a.K == D ? launch_gemm2_loop_t<1, 1, EPI_RESID, 16, 2, false>(&a, s) : launch_gemm2_loop_t<1, 1, EPI_RESID, 16, 8, false>(&a, s);
    }
    // one m-tile per workgroup, any row count (grid.y = m-tiles): o / down below the looped forms' thresholds, and the parity hooks
    if (nw == 16 && (a.K == D || a.K == F) && a.N % 16 == 0 && !a.row_index && (epi == EPI_F32 || epi == EPI_RESID))
        return launch_gemm2_16(&a, epi, a.K / 512, s);
    // 4 segments without the folded norm: no engine path, the parity hook t3k_gemm only (K = 1024, fp32 out)
    if (nw == 4 && a.K == D && !a.row_index && epi == EPI_F32)
        return mt >= 2 ? launch_gemm2_t<2, 1, EPI_F32, 4, 8, false>(&a, s) : launch_gemm2_t<1, 1, EPI_F32, 4, 8, false>(&a, s);
    return hipErrorInvalidValue;      // not a shape of this engine (K = 1024 | 4096; the generic round-1 kernel lives in tools/legacy/)
}

// W'[n][k] = bf16(W[n][k] * ln[k]): the load-time fold of an RMSNorm weight into the projection that consumes its output
// (contract: DESIGN.md "RMSNorm"; the checker's fold_ln is the same arithmetic).
void fold_norm_weight(const uint16_t* W, int N, int K, const uint16_t* ln, uint16_t* out) {
    auto b2f = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
    auto f2b = [](float f) { uint32_t u; memcpy(&u, &f, 4); if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); };
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) out[(size_t)n * K + k] = f2b(b2f(W[(size_t)n * K + k]) * b2f(ln[k]));
}

void pack_weight(const uint16_t* W, int N, int K, int Npad, uint16_t* out) {
    const int KB = K / 32;
    for (int nt = 0; nt < Npad / 16; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = nt * 16 + (lane & 15), k0 = kb * 32 + 8 * (lane >> 4);
                uint16_t* o = out + (((size_t)nt * KB + kb) * 64 + lane) * 8;
                if (n < N) memcpy(o, W + (size_t)n * K + k0, 16); else memset(o, 0, 16);
            }
}

void pack_gate_up(const uint16_t* Wg, const uint16_t* Wu, int Fdim, int K, uint16_t* out) {
    const int KB = K / 32; const size_t tile = (size_t)KB * 64 * 8;
    for (int t = 0; t < Fdim / 16; ++t) {
        pack_weight(Wg + (size_t)t * 16 * K, 16, K, 16, out + (size_t)(2 * t) * tile);
        pack_weight(Wu + (size_t)t * 16 * K, 16, K, 16, out + (size_t)(2 * t + 1) * tile);
    }
}

}  // namespace t3
