// HIP kernels of the T3 decode engine for gfx950 (MI355X, CDNA4).  wave = 64 lanes.
//
// Every floating-point rounding point and summation order in this file is part of the numerics
// contract written down in DESIGN.md ("Numerics contract"); the contract is what makes the emitted
// token ids bit-identical to the CPU restatement of the reference semantics.  Compile with
// -ffp-contract=off: every fused multiply-add below is an explicit __builtin_fmaf.
//
// Reference semantics being implemented (paths relative to the reference repo):
//   embed ............ src/chatterbox_vllm/models/t3/t3.py:440-486, 542-561
//   Llama block ...... t3.py:696-713 -> vllm LlamaModel, hyper-parameters t3-model/config.json:1-33
//   CFG logits ....... t3.py:650-673
//   sampler .......... vllm SamplingParams as configured at src/chatterbox_vllm/tts.py:455-464
#include "t3_kernels.h"
#include "t3_device.h"

#include <math.h>
#include <string.h>

#include <utility>

namespace t3 {
// Profile mode (engine.cpp: Prof): the NEXT single-kernel launch of this thread carries these two events as its start / stop events
// (hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps, what rocprofv3 reports), instead of being bracketed by two
// hipEventRecord barrier packets, which add ~2-3 us of command-processor time to a 5-30 us kernel.
static thread_local hipEvent_t g_arm_start = nullptr, g_arm_stop = nullptr;
void arm_launch_events(hipEvent_t start, hipEvent_t stop) { g_arm_start = start; g_arm_stop = stop; }
bool launch_events_armed() { return g_arm_start != nullptr; }
template <typename F, typename... Args>
static inline void launch_k(F kernel, const dim3& grid, const dim3& block, size_t lds, hipStream_t s, Args... args) {
    if (g_arm_start) {
        hipEvent_t a = g_arm_start, b = g_arm_stop;
        g_arm_start = nullptr; g_arm_stop = nullptr;
        hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, s, a, b, 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
    }
}

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
// wave's weight load is one contiguous 1 KiB (pack_weight); activations are read row-major, 16 rows x 64 B
// per wave instruction (L2-resident, tiny next to the weights).  Both operand streams run through a PD-deep
// register ring: vmcnt retires in issue order, so the activation loads must be issued as far ahead as the
// weight loads or they would drain the weight prefetch every step.
// ------------------------------------------------------------------------------------------------
#ifdef T3_GEMM_CLK      // diagnostic build only (tools/gemm_clk.hip): per-workgroup phase stamps, 100 MHz ticks
__device__ unsigned long long g_gemm_clk[8][2048][5];
#define T3_GSTAMP(i) do { if (threadIdx.x == 0) g_gemm_clk[(EPI * 2 + (NW == 16)) & 7][(blockIdx.y * gridDim.x + blockIdx.x) & 2047][i] = wall_clock64(); } while (0)
// gemm2_kernel (the current decode schedule): class 0 qkv / head, 1 gate/up, 2 o, 3 down; stamps: 0 entry, 1 A rows landed and staged
// in LDS, 2 first weight k-block landed, 3 last MFMA issued, 4 partials exchanged (barrier passed), 5 outputs stored
__device__ unsigned long long g_gemm2_clk[4][2048][6];
extern "C" int t3_debug_gemm2_clk(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm2_clk), sizeof(g_gemm2_clk)); }
#define T3_G2STAMP(i) do { if (threadIdx.x == 0) g_gemm2_clk[NORM ? (EPI == EPI_SILU ? 1 : 0) : (KBS == 2 ? 2 : 3)][(blockIdx.y * gridDim.x + blockIdx.x) & 2047][i] = wall_clock64(); } while (0)
#else
#define T3_GSTAMP(i)
#define T3_G2STAMP(i)
#endif
template <int MT, int NT, int EPI, int PD, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_kernel(GemmArgs a) {
    T3_GSTAMP(0);
    extern __shared__ __attribute__((aligned(16))) float red[];   // [NW waves][MT*NT][4 regs][64 lanes]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int KB = a.K >> 5, kbs = KB / NW, kb0 = wave * kbs;

    const uint4* wp[NT];
    const uint4* xp[MT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = a.Wp + ((size_t)(blockIdx.x * NT + t) * KB + kb0) * 64 + lane;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = (blockIdx.y * MT + i) * 16 + c;
        m = m < a.M ? m : a.M - 1;                 // padded rows re-read the last row; their outputs are dropped
        if (a.row_index) m = a.row_index[m];
        xp[i] = reinterpret_cast<const uint4*>(a.X + (size_t)m * a.K + kb0 * 32 + q * 8);
#ifdef T3_GEMM_XDUMMY      // timing diagnostic only (wrong results): every wave reads the same 1 KiB of activations
        xp[i] = reinterpret_cast<const uint4*>(a.X + q * 8 + c * 32);
#endif
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // every thread finishes MT*NT*256 / (64*NW) outputs; D[row = 4*(lane>>4) + reg][col = lane&15]
    constexpr int TOTAL = MT * NT * 256, STEP = NW * 64, ITER = (TOTAL + STEP - 1) / STEP;
    // EPI_RESID: the residual operand of this thread's outputs is requested now, so that its HBM round trip overlaps
    // the weight stream instead of sitting between the reduction and the store
    float hres[ITER];
    if constexpr (EPI == EPI_RESID) {
#pragma unroll
        for (int k = 0; k < ITER; ++k) {
            const int idx = threadIdx.x + k * STEP;
            const int it = idx >> 8, r = (idx >> 6) & 3, l2 = idx & 63;
            const int m = (blockIdx.y * MT + it / NT) * 16 + 4 * (l2 >> 4) + r, n = (blockIdx.x * NT + it % NT) * 16 + (l2 & 15);
            hres[k] = (idx < TOTAL && m < a.M && n < a.N) ? bf2f(reinterpret_cast<const uint16_t*>(a.out)[(size_t)m * a.ldo + n]) : 0.0f;
        }
    }

    uint4 wr[PD][NT], xr[PD][MT];
#pragma unroll
    for (int j = 0; j < PD; ++j)
        if (j < kbs) {
#pragma unroll
            for (int t = 0; t < NT; ++t) wr[j][t] = ld_nt(wp[t] + j * 64);
#pragma unroll
            for (int i = 0; i < MT; ++i) xr[j][i] = xp[i][j * 4];
        }
    for (int kbase = 0; kbase < kbs; kbase += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const int kb = kbase + j;
            if (kb < kbs) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xr[j][i]), as_frag(wr[j][t]), acc[i][t], 0, 0, 0);
                if (kb == 0) T3_GSTAMP(1);
                if (kb + PD < kbs) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) wr[j][t] = ld_nt(wp[t] + (kb + PD) * 64);
#pragma unroll
                    for (int i = 0; i < MT; ++i) xr[j][i] = xp[i][(kb + PD) * 4];
                }
            }
        }
    }

    T3_GSTAMP(2);
    // cross-wave (= cross-segment) reduction in segment order
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave * (MT * NT) + i * NT + t) * 4 + r) * 64 + lane] = acc[i][t][r];
    __syncthreads();
    T3_GSTAMP(3);

#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const int idx = threadIdx.x + k * STEP;
        if (idx >= TOTAL) continue;
        const int it = idx >> 8, r = (idx >> 6) & 3, l2 = idx & 63;
        const int i = it / NT, t = it % NT;
        const int m = (blockIdx.y * MT + i) * 16 + 4 * (l2 >> 4) + r;
        if (m >= a.M) continue;
        if (EPI == EPI_SILU && (t & 1)) continue;                 // packed tiles come in (gate, up) pairs
        float v[EPI == EPI_SILU ? 2 : 1];
#pragma unroll
        for (int u = 0; u < (EPI == EPI_SILU ? 2 : 1); ++u) {
            const int itu = it + u;
            float tot = 0.0f;
#pragma unroll
            for (int gsum = 0; gsum < NW / 4; ++gsum) {
                float s4 = red[(((4 * gsum + 0) * (MT * NT) + itu) * 4 + r) * 64 + l2];
                s4 = s4 + red[(((4 * gsum + 1) * (MT * NT) + itu) * 4 + r) * 64 + l2];
                s4 = s4 + red[(((4 * gsum + 2) * (MT * NT) + itu) * 4 + r) * 64 + l2];
                s4 = s4 + red[(((4 * gsum + 3) * (MT * NT) + itu) * 4 + r) * 64 + l2];
                tot = gsum == 0 ? s4 : tot + s4;
            }
            v[u] = tot;
        }
        if constexpr (EPI == EPI_SILU) {
            const int n = (blockIdx.x * (NT / 2) + (t >> 1)) * 16 + (l2 & 15);      // tile pair index == output tile index
            if (n < a.N)
                reinterpret_cast<uint16_t*>(a.out)[(size_t)m * a.ldo + n] = (uint16_t)silu_mul_bf(f2bf(v[0]), f2bf(v[1]));
        } else {
            const int n = (blockIdx.x * NT + t) * 16 + (l2 & 15);
            if (n >= a.N) continue;
            if constexpr (EPI == EPI_F32) {
                reinterpret_cast<float*>(a.out)[(size_t)m * a.ldo + n] = v[0];
            } else if constexpr (EPI == EPI_BF16) {
                reinterpret_cast<uint16_t*>(a.out)[(size_t)m * a.ldo + n] = (uint16_t)f2bf(v[0]);
            } else {   // EPI_RESID: h = bf16(h + bf16(y))
                reinterpret_cast<uint16_t*>(a.out)[(size_t)m * a.ldo + n] = (uint16_t)f2bf(hres[k] + rbf(v[0]));
            }
        }
    }
    T3_GSTAMP(4);
}

// ------------------------------------------------------------------------------------------------
// gemm2_kernel: the decode schedule of the NORM forms (NW = 4, K = 1024, MT <= 2) and of the 16-segment forms at one
// m-tile per workgroup (o_proj / down_proj).  Same numbers as gemm_kernel, two differences in how the bytes move:
//   * A operand: a wave reads its K slice of its 16 rows as FULL row segments (KBS * 64 contiguous bytes per row, whole
//     128-byte lines) into a wave-private, XOR-swizzled LDS image and takes its MFMA fragments from there with ds_read_b128.
//     The fragment-shaped global loads of gemm_kernel (16 rows x 64 B per instruction, half lines) cost the texture path twice
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
template <int MT, int NT, int EPI, int NW, int KBS, bool NORM, int AV = KBS>
__global__ __launch_bounds__((NW + gemm2_ew<NW>()) * 64) void gemm2_kernel(GemmArgs a) {
    T3_G2STAMP(0);
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
    constexpr int EW = gemm2_ew<NW>();
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
    static_for([&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        uint4 af[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const uint4*>(aimg + i * (KBS * 1024) + a_img_off<KBS>(c, 4 * kb + q));
        wait_vmcnt<(KBS - 1 - kb) * NT>();            // this k-block's NT weight tiles have landed ((KBS - 1 - kb) * NT younger loads may still fly)
#pragma unroll
        for (int t = 0; t < NT; ++t) landed(wr[kb][t]);
        if constexpr (kb == 0) T3_G2STAMP(2);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if constexpr (NORM) ss[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af[i]), as_frag(af[i]), ss[i], 0, 0, 0);   // diagonal = sum of squares
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af[i]), as_frag4(wr[kb][t]), acc[i][t], 0, 0, 0);
        }
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
        static_for([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            uint4 af[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const uint4*>(aimg + i * (KBS * 1024) + a_img_off<KBS>(c, 4 * kb + q));
            if constexpr (FIRST) {     // the weights land during the first group: younger = the later weight tiles + the next group's rows
                wait_vmcnt<(KBS - 1 - kb) * NT + MT * KBS>();
#pragma unroll
                for (int t = 0; t < NT; ++t) landed(wr[kb][t]);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (NORM) ss[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af[i]), as_frag(af[i]), ss[i], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af[i]), as_frag4(wr[kb][t]), acc[i][t], 0, 0, 0);
            }
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
// Prefill-sized GEMM (M >= 256 rows): the same numbers as gemm_kernel, another schedule.  A workgroup of four waves owns a
// 128-row x 64-column tile (4 packed n-tiles); every K step of 32 is staged once through LDS (activations 128 x 64 B row
// pieces; weights: 4 packed 1 KiB fragments, the NORM forms' carrying the norm weight) and feeds 32 MFMAs, so a weight byte is
// re-read once per 128 rows instead of once per 32 and an activation byte once per 64 columns instead of once per workgroup.
// Contract order per output: one MFMA chain per K segment FROM ZERO in ascending k; segments folded left to right in groups
// of four (G = ((s0 + s1) + s2) + s3), groups folded left to right -- hence three accumulator sets (segment, group, total).
// The row statistic of the NORM forms comes from row_rstd_kernel (the same MFMA chains as gemm2_kernel's own).
// ------------------------------------------------------------------------------------------------
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
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_byte_addr) {
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
    const int m0 = blockIdx.y * 128, nt0 = blockIdx.x * NTW;
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
                sg[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af[i]), as_frag(bf[u]), sg[i][u], 0, 0, 0);
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
    // epilogue: D[row = 4 (lane >> 4) + r][col = lane & 15] of every 16 x 16 tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wr * 64 + i * 16 + 4 * (lane >> 4) + r;
            if (m >= a.M) continue;
            float v[WC];
#pragma unroll
            for (int u = 0; u < WC; ++u) { if constexpr (NSEG > 4) v[u] = tot[i][u][r]; else v[u] = gr[i][u][r]; }
            if constexpr (NORM) {
                const float rs = rstd[m];
#pragma unroll
                for (int u = 0; u < WC; ++u) v[u] = v[u] * rs;
            }
            if constexpr (EPI == EPI_SILU) {
#pragma unroll
                for (int u = 0; u < WC; u += 2) {                             // packed pair (gate, up) -> one output tile
                    const int n = ((nt0 + wc * WC + u) >> 1) * 16 + (lane & 15);
                    if (n < a.N) reinterpret_cast<uint16_t*>(a.out)[(size_t)m * a.ldo + n] = (uint16_t)silu_mul_bf(f2bf(v[u]), f2bf(v[u + 1]));
                }
            } else {
#pragma unroll
                for (int u = 0; u < WC; ++u) {
                    const int n = (nt0 + wc * WC + u) * 16 + (lane & 15);
                    if (n >= a.N) continue;
                    if constexpr (EPI == EPI_F32) {
                        reinterpret_cast<float*>(a.out)[(size_t)m * a.ldo + n] = v[u];
                    } else {
                        uint16_t* op = reinterpret_cast<uint16_t*>(a.out) + (size_t)m * a.ldo + n;
                        if constexpr (EPI == EPI_BF16) *op = (uint16_t)f2bf(v[u]);
                        else *op = (uint16_t)f2bf(bf2f(*op) + rbf(v[u]));     // EPI_RESID: h = bf16(h + bf16(y))
                    }
                }
            }
        }
    }
}

static int g_pgemm_min_rows = -1, g_pgemm_wide_rows = -1;
static int g_gemm_small_m = 1;          // T3_GEMM_SMALL_M=0: the one-tile GEMMs issue every activation-row load (read again by every prepare_kernels call, i.e. per engine)
void set_pgemm_min_rows(int rows) { g_pgemm_min_rows = rows; }
void set_pgemm_wide_rows(int rows) { g_pgemm_wide_rows = rows; }

// Large-M path of launch_gemm: returns hipErrorNotSupported when the shape is not one of the layer forms.
static hipError_t launch_pgemm(const GemmArgs& a, int epi, hipStream_t s) {
    const bool norm = a.norm != 0;
    const int nseg = a.nw == 16 ? 16 : 4;
    const int ntiles = (a.N + 15) / 16 * (epi == EPI_SILU ? 2 : 1);
    if (a.row_index || ntiles % 4 || a.K % (32 * nseg) || (norm && (!a.rstd_scratch || a.K != D))) return hipErrorNotSupported;
    // 128 x 128 tiles for the 4-segment forms from 2048 rows on only with T3_PGEMM_WC=4 (or the parity tests' hook): measured at
    // 8 178 rows x 30 layers, a prefill step takes 18.35 ms with 128 x 64 tiles, 18.76 with 128 x 128 and a 3-stage ring, 19.07
    // with a 4-stage ring -- the two workgroups per CU that fit hide less latency than the four of the narrow form
    static int wc_env = -1;
    if (wc_env < 0) { const char* e = getenv("T3_PGEMM_WC"); wc_env = e ? atoi(e) : 2; }
    const int wide_rows = g_pgemm_wide_rows >= 0 ? g_pgemm_wide_rows : 2048;          // 0 = never
    const bool wide = (wc_env == 4 || g_pgemm_wide_rows > 0) && nseg == 4 && ntiles % 8 == 0 && wide_rows > 0 && a.M >= wide_rows;
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

template <int MT, int NT, int EPI, int NW>
static hipError_t launch_gemm_t(const GemmArgs& a, hipStream_t s) {
    // ring depth, bounded by the register file: 4-wave workgroups may use ~200 VGPRs, 16-wave ones 128
    constexpr int PD = NW == 16 ? (MT <= 2 ? 4 : 2) : ((MT + NT) <= 6 ? 8 : 4);
    const int ntiles = (a.N + 15) / 16;           // EPI_SILU: N = F -> one workgroup per output tile (2 packed tiles)
    const int gx = (EPI == EPI_SILU) ? (ntiles + NT / 2 - 1) / (NT / 2) : (ntiles + NT - 1) / NT;
    const int gy = ((a.M + 15) / 16 + MT - 1) / MT;
    const size_t lds = (size_t)NW * MT * NT * 256 * sizeof(float);        // <= 64 KiB for every instantiation below
    hipLaunchKernelGGL((gemm_kernel<MT, NT, EPI, PD, NW>), dim3(gx, gy), dim3(NW * 64), lds, s, a);
    return hipGetLastError();
}

// hipFuncSetAttribute applies to the CURRENT device: every "already raised" flag below is kept per device, so a process that
// drives engines on several GPUs (LLM(device_id=...)) raises the limits on each of them
constexpr int MAX_DEVICES = 64;
static inline int cur_device() { int d = 0; (void)hipGetDevice(&d); return d >= 0 && d < MAX_DEVICES ? d : 0; }

// gemm2_kernel launcher; a == nullptr: only raise the kernel's dynamic-LDS limit (prepare_kernels, before any stream capture)
template <int MT, int NT, int EPI, int NW, int KBS, bool NORM, int AV>
static hipError_t launch_gemm2_av(const GemmArgs* a, hipStream_t s) {
    constexpr size_t lds = (size_t)NW * MT * KBS * 1024 + (NORM ? (size_t)NW * MT * 16 * sizeof(float) : 0) + (gemm2_ew<NW>() ? 256 : 0);    // + the prefetch dump corner
    auto kern = gemm2_kernel<MT, NT, EPI, NW, KBS, NORM, AV>;
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
    launch_k(kern, dim3(gx, gy), dim3((NW + gemm2_ew<NW>()) * 64), lds, s, *a);
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
hipError_t prepare_gemm2() {
    hipError_t e;
    for (int epi : {EPI_F32, EPI_BF16, EPI_SILU})
        for (int mt = 1; mt <= 2; ++mt)
            for (int nt = 1; nt <= 4; ++nt)
                if ((e = launch_gemm2_norm(nullptr, epi, mt, nt, nullptr)) != hipSuccess) return e;
    for (int epi : {EPI_F32, EPI_RESID})
        for (int kbs : {2, 8})
            if ((e = launch_gemm2_16(nullptr, epi, kbs, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_loop_t<2, 1, EPI_BF16, 4, 8, true>(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_loop_t<2, 2, EPI_SILU, 4, 8, true>(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_loop_t<2, 4, EPI_SILU, 4, 8, true>(nullptr, nullptr)) != hipSuccess) return e;
    if ((e = launch_gemm2_loop_t<1, 1, EPI_RESID, 16, 8, false>(nullptr, nullptr)) != hipSuccess) return e;
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
    if (nw == 16 && mt > 4) mt = 4;               // LDS: 16 waves x MT x 1 KiB x 4
    {
        // rows from which the LDS-tiled schedule takes over (0 = never).  1024: a 256-row call is a DECODE step of 128 utterances,
        // where 128 x 64 tiles leave 32-128 workgroups (measured on the continuous-batching run of tools/bench_serving.py:
        // 16.8 k tok/s with the switch at 256 rows, 25.0 k at 1024 or 2048).
        // Per form since the looped schedules exist (tools/chain_proto at 320-1023 rows, us per launch looped | LDS-tiled: gate/up
        // 24.9 | 25.8 at 384 rows, 30.6 | 28.6 at 512; qkv 16.8 | 22.0 at 512, 25.4 | 23.6 at 768; o 17.0 | 20.4 and down 34.1 | 46.5
        // at 1023: the 16-segment fold of a 128 x 64 tile is a fixed ~18 / ~40 us): -2 = these per-form switches.
        if (g_pgemm_min_rows == -1) { const char* e = getenv("T3_PGEMM_MIN_ROWS"); g_pgemm_min_rows = e ? atoi(e) : -2; }
        const int pg_min = g_pgemm_min_rows != -2 ? g_pgemm_min_rows
                         : epi == EPI_SILU ? 448 : (nw == 4 ? (a.row_index ? 1024 : 704) : (a.K == D ? 1280 : 1600));
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
        if (loop16_min > 0 && a.M >= loop16_min)
            return a.K == D ? launch_gemm2_loop_t<1, 1, EPI_RESID, 16, 2, false>(&a, s) : launch_gemm2_loop_t<1, 1, EPI_RESID, 16, 8, false>(&a, s);
    }
    if (nw == 16 && mt == 1 && (a.K == D || a.K == F) && a.N % 16 == 0 && !a.row_index && (epi == EPI_F32 || epi == EPI_RESID))
        return launch_gemm2_16(&a, epi, a.K / 512, s);
#define T3_MT(E, NT, NWV)                                                    \
    switch (mt) {                                                            \
        case 1: return launch_gemm_t<1, NT, E, NWV>(a, s);                   \
        case 2: return launch_gemm_t<2, NT, E, NWV>(a, s);                   \
        case 4: return launch_gemm_t<4, NT, E, NWV>(a, s);                   \
        default: return launch_gemm_t<(NWV == 16 ? 4 : 8 / NT), NT, E, NWV>(a, s); \
    }
    if (nw == 16) {
        switch (epi) {
            case EPI_F32: T3_MT(EPI_F32, 1, 16)
            case EPI_RESID: T3_MT(EPI_RESID, 1, 16)
            default: return hipErrorInvalidValue;
        }
    }
    switch (epi) {
        case EPI_F32: T3_MT(EPI_F32, 1, 4)
        case EPI_BF16: T3_MT(EPI_BF16, 1, 4)
        case EPI_RESID: T3_MT(EPI_RESID, 1, 4)
        case EPI_SILU: T3_MT(EPI_SILU, 2, 4)
    }
#undef T3_MT
    return hipErrorInvalidValue;
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
    // the first kernel of a step also zeroes the step's hand-off words (tickets and flags of the qkv-in-attention launches)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n_zero; i += gridDim.x * 256) a.zero_words[i] = 0u;
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
// RoPE (rotate-half, llama3-scaled table) + paged KV write: one wave per row.
// lane = 4*head + part; part covers pairs i in [8*part, 8*part+8):  o1 = x1*c - x2*s, o2 = x2*c + x1*s
// (cos/sin are bf16-valued so both products are exact; one fp32 rounding, then bf16).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rope8(const uint4& x1, const uint4& x2, const float* c, const float* s, uint4& o1, uint4& o2) {
    float a[8], b[8]; unpack8(x1, a); unpack8(x2, b);
    float r1[8], r2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { r1[e] = a[e] * c[e] - b[e] * s[e]; r2[e] = b[e] * c[e] + a[e] * s[e]; }
    o1.x = pack2(r1[0], r1[1]); o1.y = pack2(r1[2], r1[3]); o1.z = pack2(r1[4], r1[5]); o1.w = pack2(r1[6], r1[7]);
    o2.x = pack2(r2[0], r2[1]); o2.y = pack2(r2[2], r2[3]); o2.z = pack2(r2[4], r2[5]); o2.w = pack2(r2[6], r2[7]);
}
// Paged KV layout of one (block, head): [chunk-in-block (KV_BLOCK/64)][8 fragments][64 lanes][8 bf16] for K and for V.
//   K fragment (tt, ds), lane t + 16 kg, element j  =  K[token 16 tt + t of the chunk][dim 32 ds + 8 kg + j]    (MFMA A operand: rows = tokens);
//     in MEMORY the lane's 16-byte piece sits at piece index 4 t + kg of the fragment (token-major: k_piece() below)
//   V fragment (dt, ts), lane d + 16 kg, element j  =  V[token 32 ts + 8 kg + j of the chunk][dim 16 dt + d]    (MFMA A operand: rows = dims)
// so the attention kernel feeds v_mfma_f32_16x16x32_bf16 straight from fully coalesced 1 KiB wave loads.
__device__ __forceinline__ size_t kv_head_base(int blk, int kv, int h) {
    return (size_t)blk * KV_BLOCK_ELEMS + (size_t)(kv * H + h) * KV_HEAD_ELEMS;
}
// Where the 16-byte piece of lane (token t, dim slice kg) sits inside a K fragment's 1 KiB: token-major (4 t + kg), so that a token's four
// slices are 64 contiguous bytes and the newest token's K write touches 2 lines per (row, head).  In lane order (t + 16 kg: the MFMA A
// operand's own order, rounds 1-3, -DT3_K_TOKEN_MAJOR=0) they are 256 bytes apart, 8 lines per (row, head), and the write's cost follows the
// lines touched: C3 21.12 -> 21.27 k tok/s (profiles/r03_k_token_major_*.json).  A wave still loads the same 1 KiB per fragment, each lane
// from its permuted place.  Every K reader and writer goes through k_piece() / k_lane_piece().
#ifndef T3_K_TOKEN_MAJOR
#define T3_K_TOKEN_MAJOR 1
#endif
__device__ __forceinline__ int k_piece(int t, int kg) { return T3_K_TOKEN_MAJOR ? 4 * t + kg : t + 16 * kg; }
__device__ __forceinline__ int k_lane_piece(int lane) { return T3_K_TOKEN_MAJOR ? 4 * (lane & 15) + (lane >> 4) : lane; }      // the piece lane (t = lane % 16, kg = lane / 16) loads
__device__ __forceinline__ size_t k_slot(int tok_in_block, int ds, int kg) {      // start of the 8-element (16 B) piece
    const int ci = tok_in_block / CHUNK, tc = tok_in_block % CHUNK;
    return (size_t)ci * (CHUNK * HD) + (size_t)((tc >> 4) * 2 + ds) * 512 + (size_t)k_piece(tc & 15, kg) * 8;
}
__device__ __forceinline__ size_t v_elem(int tok_in_block, int dim) {              // one bf16
    const int ci = tok_in_block / CHUNK, tc = tok_in_block % CHUNK;
    return (size_t)ci * (CHUNK * HD) + (size_t)((dim >> 4) * 2 + (tc >> 5)) * 512 + (size_t)((dim & 15) + 16 * ((tc & 31) >> 3)) * 8 + (tc & 7);
}

__global__ __launch_bounds__(256) void rope_kv_kernel(RopeArgs a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int* rec = a.rowrec + (size_t)row * a.row_stride;
    const int pos = rec[1];
    const int blk = rec[ROW_HDR + pos / KV_BLOCK], tok = pos % KV_BLOCK;
    const int h = lane >> 2, part = lane & 3, i0 = part * 8;
    float c[8], s[8];
    {
        const float4* cp = reinterpret_cast<const float4*>(a.cos_t + (size_t)pos * 32 + i0);
        const float4* sp = reinterpret_cast<const float4*>(a.sin_t + (size_t)pos * 32 + i0);
        const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
        c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
        s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w; s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w;
    }
    const uint16_t* qr = a.qkv + (size_t)row * QKV;
    uint4 o1, o2;
    rope8(*reinterpret_cast<const uint4*>(qr + h * 64 + i0), *reinterpret_cast<const uint4*>(qr + h * 64 + 32 + i0), c, s, o1, o2);
    uint16_t* qo = a.q_out + (size_t)row * D + h * 64;
    *reinterpret_cast<uint4*>(qo + i0) = o1; *reinterpret_cast<uint4*>(qo + 32 + i0) = o2;
    rope8(*reinterpret_cast<const uint4*>(qr + D + h * 64 + i0), *reinterpret_cast<const uint4*>(qr + D + h * 64 + 32 + i0), c, s, o1, o2);
    uint16_t* kb = a.kv_layer + kv_head_base(blk, 0, h);
    *reinterpret_cast<uint4*>(kb + k_slot(tok, 0, part)) = o1;       // dims 8*part..   -> ds 0, kg = part
    *reinterpret_cast<uint4*>(kb + k_slot(tok, 1, part)) = o2;       // dims 32+8*part.. -> ds 1, kg = part
    // V is stored token-minor (8 consecutive tokens of one dim = 16 bytes, the MFMA A operand of P.V).  Prefill rows come as runs of
    // consecutive positions: where the launch holds all 8 rows of an aligned token group, the wave of the group's first row
    // transposes the 8 x 1024 block in registers and writes whole 16-byte pieces (128 contiguous bytes per lane); the other seven
    // waves skip V.  Anything else (decode rows, ragged ends of a run) keeps the element-wise writes.
    const int p8 = pos & 7, jl = row - p8;
    bool full = false;
    if (jl >= 0 && jl + 7 < a.rows) {
        int ok = 1;
        if (lane < 8) { const int* r2 = a.rowrec + (size_t)(jl + lane) * a.row_stride; ok = (r2[0] == rec[0]) && (r2[1] == pos - p8 + lane); }
        full = __all(ok);
    }
    if (full) {
        if (p8 != 0) return;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pc = lane + 64 * u, hh = pc >> 3, oct = pc & 7;          // (head, 8 dims 8 oct .. 8 oct + 7)
            uint4 x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = *reinterpret_cast<const uint4*>(a.qkv + (size_t)(row + i) * QKV + 2 * D + hh * 64 + oct * 8);
            uint4* dst = reinterpret_cast<uint4*>(a.kv_layer + kv_head_base(blk, 1, hh) + v_elem(tok, oct * 8));
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const uint32_t sel = (d & 1) ? 0x07060302u : 0x05040100u;     // high or low halves of (second, first) operand
                auto w = [&](const uint4& v) { return (d >> 1) == 0 ? v.x : (d >> 1) == 1 ? v.y : (d >> 1) == 2 ? v.z : v.w; };
                uint4 y;
                y.x = __builtin_amdgcn_perm(w(x[1]), w(x[0]), sel); y.y = __builtin_amdgcn_perm(w(x[3]), w(x[2]), sel);
                y.z = __builtin_amdgcn_perm(w(x[5]), w(x[4]), sel); y.w = __builtin_amdgcn_perm(w(x[7]), w(x[6]), sel);
                dst[d] = y;
            }
        }
        return;
    }
    uint16_t* vb = a.kv_layer + kv_head_base(blk, 1, h);
    const uint16_t* vs = qr + 2 * D + h * 64 + part * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) vb[v_elem(tok, part * 16 + e)] = vs[e];
}
hipError_t launch_rope_kv(const RopeArgs& a, hipStream_t s) {
    if (a.rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(rope_kv_kernel, dim3((a.rows + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}
// parity hook (t3k_decode_attention): the K / V of a row's (stream, position) as the pool holds them, one wave per row, lane = (head, part)
__global__ __launch_bounds__(256) void kv_gather_kernel(const uint16_t* kv_layer, const int* rowrec, int row_stride, int rows, uint16_t* out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int* rec = rowrec + (size_t)row * row_stride;
    const int pos = rec[1], blk = rec[ROW_HDR + pos / KV_BLOCK], tok = pos % KV_BLOCK;
    const int h = lane >> 2, part = lane & 3;
    const uint16_t* kb = kv_layer + kv_head_base(blk, 0, h);
    const uint16_t* vb = kv_layer + kv_head_base(blk, 1, h);
    uint16_t* ko = out + (size_t)row * 2 * D + h * HD;
    uint16_t* vo = ko + D;
    *reinterpret_cast<uint4*>(ko + part * 8) = *reinterpret_cast<const uint4*>(kb + k_slot(tok, 0, part));
    *reinterpret_cast<uint4*>(ko + 32 + part * 8) = *reinterpret_cast<const uint4*>(kb + k_slot(tok, 1, part));
#pragma unroll
    for (int e = 0; e < 16; ++e) vo[part * 16 + e] = vb[v_elem(tok, part * 16 + e)];
}
hipError_t launch_kv_gather(const uint16_t* kv_layer, const int* rowrec, int row_stride, int rows, uint16_t* out, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(kv_gather_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, kv_layer, rowrec, row_stride, rows, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Paged attention for one (row, head): context = positions 0..row_pos of the row's stream.
// One workgroup of NW waves; wave w takes chunks c = w, w+NW, ... (chunk = 64 tokens = 8 KiB K + 8 KiB V of this head,
// contiguous, read with fully coalesced 1 KiB wave loads straight into MFMA operand registers).
// QK^T and P.V run on the matrix cores (v_mfma_f32_16x16x32_bf16; q and the bf16 probabilities are replicated over
// the 16 B-operand columns, so every column of D carries the same numbers); the softmax needs ONE exp per lane
// (lane = token).  Per-chunk (m_c, l_c, o_c[64]) go to LDS; wave 0 folds them in ascending chunk order.
// All orders are the contract's (DESIGN.md "Attention").
// ------------------------------------------------------------------------------------------------
// v[lane ^ off] for off = 32, 16, 8, 4, 2, 1 without the LDS crossbar (ds_bpermute costs a dependent ~100-cycle round trip per
// level): gfx950's half / row swaps for 32 and 16, DPP row rotate / shifts / quad permutes below that.  Same pairing as __shfl_xor,
// so the butterfly sums keep the contract's order.
template <int OFF>
__device__ __forceinline__ float lane_xor(float v, int lane) {
    const int x = __float_as_int(v);
    if constexpr (OFF == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);      // r[0] = {lo, lo}, r[1] = {hi, hi}
        return __int_as_float((int)((lane & 32) ? r[0] : r[1]));
    } else if constexpr (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);      // r[0] = even rows twice, r[1] = odd rows twice
        return __int_as_float((int)((lane & 16) ? r[0] : r[1]));
    } else if constexpr (OFF == 8) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, true));              // row_ror:8
    } else if constexpr (OFF == 4) {
        const int up = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0xf, true), dn = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shl:4 (from lane + 4), row_shr:4 (from lane - 4)
        return __int_as_float((lane & 4) ? dn : up);
    } else if constexpr (OFF == 2) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true));               // quad_perm [2,3,0,1]
    } else {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true));               // quad_perm [1,0,3,2]
    }
}
__device__ __forceinline__ float wave_max_f32(float m, int lane) {
    m = fmaxf(m, lane_xor<32>(m, lane)); m = fmaxf(m, lane_xor<16>(m, lane)); m = fmaxf(m, lane_xor<8>(m, lane));
    m = fmaxf(m, lane_xor<4>(m, lane)); m = fmaxf(m, lane_xor<2>(m, lane)); m = fmaxf(m, lane_xor<1>(m, lane));
    return m;
}
__device__ __forceinline__ float wave_bfly_add_f32(float v, int lane) {      // contract order: xor 32, 16, 8, 4, 2, 1
    v = v + lane_xor<32>(v, lane); v = v + lane_xor<16>(v, lane); v = v + lane_xor<8>(v, lane);
    v = v + lane_xor<4>(v, lane); v = v + lane_xor<2>(v, lane); v = v + lane_xor<1>(v, lane);
    return v;
}

__device__ __forceinline__ void patch16(uint4& v, int j, uint32_t val) {      // replace bf16 element j (0..7) of v
    const uint32_t sh = (j & 1) * 16, keep = ~(0xffffu << sh), ins = val << sh;
    const int w = j >> 1;
    v.x = w == 0 ? ((v.x & keep) | ins) : v.x; v.y = w == 1 ? ((v.y & keep) | ins) : v.y;
    v.z = w == 2 ? ((v.z & keep) | ins) : v.z; v.w = w == 3 ? ((v.w & keep) | ins) : v.w;
}

#ifdef T3_ATTN_CLK      // diagnostic build only (tools/attn_clk.py): per-workgroup phase stamps of the LAST launch, 100 MHz ticks
__device__ unsigned long long g_attn_clk[4096][6];
extern "C" int t3_debug_attn_clk(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_clk), sizeof(g_attn_clk)); }
#define T3_ASTAMP(i) do { if (threadIdx.x == 0) g_attn_clk[(blockIdx.y * gridDim.x + blockIdx.x) & 4095][i] = wall_clock64(); } while (0)
#else
#define T3_ASTAMP(i)
#endif
template <int NW, bool NT, bool FUSE>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 4 : 2) void attention_kernel(AttnArgs a) {
    T3_ASTAMP(0);
    extern __shared__ __attribute__((aligned(16))) float part[];   // [max_chunks] m | [max_chunks] l | [max_chunks][64] o | per wave: 64 scores, 64 bf16 p | FUSE, per wave: [12][64] newest k / v
    float* pm = part; float* pl = part + a.max_chunks; float* po = part + 2 * a.max_chunks;
    // EARLY (the 8-wave form = 1-4 utterances): the launch is a chain of dependent loads (kernel arguments -> row record -> block id -> tile
    // -> arithmetic, 3.6 of its 5.6 us at B = 1), so (a) the wave index is made uniform for the compiler: block ids come by SCALAR loads, (b) the
    // block of the wave's first chunk is asked for together with the context length, (c) the pre-RoPE q / k pieces, which need nothing from the
    // record, are asked for before it has arrived, and the RoPE table rows before the tile.  B = 1: 1 318 -> 1 341 tok/s on one box.  At 64
    // rows (4-wave form) the same order is 0.6 % SLOWER (the launch streams at the memory system's pace from its first microsecond; 21.18 ->
    // 21.06 k tok/s, three alternating runs): the 4-wave form keeps the order of round 2.
    constexpr bool EARLY = FUSE && NW == 8;
    const int lane = threadIdx.x & 63, wave = EARLY ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x >> 6);
    float* sbuf = part + 66 * a.max_chunks + wave * 96;            // 64 floats of scores, then 64 bf16 (32 floats) of probabilities
    uint32_t* stash = reinterpret_cast<uint32_t*>(part + 66 * a.max_chunks + NW * 96) + wave * (12 * 64) + (threadIdx.x & 63);   // FUSE: the newest key / value park here
    uint16_t* pbuf = reinterpret_cast<uint16_t*>(sbuf + 64);
    // -DT3_ATTN_DYNAMIC_CHUNKS (diagnostic build): chunks beyond a wave's first are handed out by a counter in LDS to whichever wave is
    // free (the partials are indexed by chunk and folded in chunk order at the end, so who computes a chunk changes no number).
    // Bit-exact, and measured 0.8 % SLOWER at C3 on one box (20.14 against 20.30 k tok/s, attention 30.3 against 29.9 us per evented
    // launch): the static round-robin stays.
#ifdef T3_ATTN_DYNAMIC_CHUNKS
    constexpr bool DYN = true;
#else
    constexpr bool DYN = false;
#endif
    int* next_chunk = reinterpret_cast<int*>(part + 66 * a.max_chunks + NW * 96 + (FUSE ? NW * 12 * 64 : 0));
    if constexpr (DYN) {
        if (threadIdx.x == 0) *next_chunk = NW;
        __syncthreads();                               // at entry: the waves of a workgroup start together, nothing is in flight yet
    }
    const int h = blockIdx.x, row = blockIdx.y;
    const int col = lane & 15, kg = lane >> 4;
    const uint16_t* src = FUSE ? a.qkv + (size_t)row * QKV + h * HD : nullptr;
    uint4 qraw[2], kraw[2];
    if constexpr (EARLY) {
        qraw[0] = *reinterpret_cast<const uint4*>(src + kg * 8); qraw[1] = *reinterpret_cast<const uint4*>(src + 32 + kg * 8);
        kraw[0] = *reinterpret_cast<const uint4*>(src + D + kg * 8); kraw[1] = *reinterpret_cast<const uint4*>(src + D + 32 + kg * 8);
    }
    const int* rec = a.rowrec + (size_t)row * a.row_stride;
    const int L = rec[1] + 1;
    const int nc = (L + CHUNK - 1) / CHUNK;
    const int* bt = rec + ROW_HDR;                 // the row's KV block ids travel with the row record
    constexpr int CPB = KV_BLOCK / CHUNK;          // chunks per physical block
    // EARLY: before it is known whether the wave has a chunk at all (the word is inside the row record either way)
    int blk_first = 0;
    if constexpr (EARLY) {
        blk_first = __builtin_amdgcn_readfirstlane(bt[wave / CPB]);      // uniform already; says so to every build (the stamped one kept it in a VGPR)
        asm volatile("" : "+s"(blk_first));         // hipcc would sink the load into the `wave < nc` branch, i.e. behind the wait for the context length
    }

    uint4 kf[8], vf[8];                            // K fragments (tt, ds) at 2 tt + ds; V fragments (dt, ts) at 2 dt + ts
    // A chunk's tile is 8 K fragments (16 tokens x 32 dims each) + 8 V fragments (32 tokens x 16 dims each).  Of the context's LAST
    // chunk only the fragments that hold tokens of the pool are requested (the fused form's newest token comes from registers): on
    // average a third of that tile, ~4 % of a launch's bytes at C3.  Fragments left out are zeroed: their scores are masked anyway, but
    // a V fragment meets p = 0 in the MFMA and 0 x (a stale NaN pattern) would not be 0.
#ifdef T3_ATTN_FULL_TILES
    constexpr bool PARTIAL = false;
#else
    constexpr bool PARTIAL = true;
#endif
    auto load_tiles = [&](int c) {
        const int blk = (EARLY && c == wave) ? blk_first : bt[c / CPB], ci = c % CPB;
        const uint4* Kp = reinterpret_cast<const uint4*>(a.kv_layer + kv_head_base(blk, 0, h) + (size_t)ci * (CHUNK * HD)) + k_lane_piece(lane);
        const uint4* Vp = reinterpret_cast<const uint4*>(a.kv_layer + kv_head_base(blk, 1, h) + (size_t)ci * (CHUNK * HD)) + lane;
        const int npool = L - (FUSE ? 1 : 0) - c * CHUNK;       // tokens of this chunk that live in the pool (wave-uniform; >= 64 except in the last chunk)
        if (!PARTIAL || npool >= CHUNK) {
#pragma unroll
            for (int f = 0; f < 8; ++f) kf[f] = NT ? ld_nt(Kp + f * 64) : Kp[f * 64];
#pragma unroll
            for (int f = 0; f < 8; ++f) vf[f] = NT ? ld_nt(Vp + f * 64) : Vp[f * 64];
        } else {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                if (16 * tt < npool) { kf[2 * tt] = NT ? ld_nt(Kp + (2 * tt) * 64) : Kp[(2 * tt) * 64]; kf[2 * tt + 1] = NT ? ld_nt(Kp + (2 * tt + 1) * 64) : Kp[(2 * tt + 1) * 64]; }
                else { kf[2 * tt] = make_uint4(0, 0, 0, 0); kf[2 * tt + 1] = make_uint4(0, 0, 0, 0); }
            }
#pragma unroll
            for (int ts = 0; ts < 2; ++ts)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    if (32 * ts < npool) vf[2 * dt + ts] = NT ? ld_nt(Vp + (2 * dt + ts) * 64) : Vp[(2 * dt + ts) * 64];
                    else vf[2 * dt + ts] = make_uint4(0, 0, 0, 0);
                }
        }
    };

    // the wave's first K/V tile is requested before anything else so that the q / RoPE prologue overlaps its flight (EARLY: right behind the
    // RoPE table rows, which are small and which the prologue needs first: loads retire in issue order)
    if (!EARLY && wave < nc) load_tiles(wave);
    uint4 qfrag[2];                                 // B operand: q[32 ds + 8 kg .. +7], the same in all 16 columns
    uint4 knf[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};      // FUSE: the newest key in A-fragment form
    uint32_t vnew[4] = {0, 0, 0, 0};                // FUSE: the newest value, dims 16 dt + col
    if constexpr (FUSE) {
        // RoPE of this head's q and k exactly as rope_kv_kernel does it (same products, same roundings): a lane holds
        // both halves of its rotation pairs (dims 8 kg + j and 32 + 8 kg + j), i.e. exactly its two operand fragments.
        const int pos = L - 1;
        float c[8], s[8];
        {
            const float4* cp = reinterpret_cast<const float4*>(a.cos_t + (size_t)pos * 32 + kg * 8);
            const float4* sp = reinterpret_cast<const float4*>(a.sin_t + (size_t)pos * 32 + kg * 8);
            const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
            if (EARLY && wave < nc) load_tiles(wave);
            c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
            s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w; s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w;
        }
        if constexpr (EARLY) {
            rope8(qraw[0], qraw[1], c, s, qfrag[0], qfrag[1]);
            rope8(kraw[0], kraw[1], c, s, knf[0], knf[1]);
        } else {
            rope8(*reinterpret_cast<const uint4*>(src + kg * 8), *reinterpret_cast<const uint4*>(src + 32 + kg * 8), c, s, qfrag[0], qfrag[1]);
            rope8(*reinterpret_cast<const uint4*>(src + D + kg * 8), *reinterpret_cast<const uint4*>(src + D + 32 + kg * 8), c, s, knf[0], knf[1]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vnew[dt] = src[2 * D + 16 * dt + col];
#if defined(T3_ATTN_NOKVWRITE) || defined(T3_ATTN_K_TILE_WRITE) || defined(T3_ATTN_NOKWRITE)
        if (false) {                                // NOKVWRITE: timing diagnostic only (the following steps read stale K / V); K_TILE_WRITE: K goes back from the chunk loop
#else
        if (wave == 0) {                            // paged write of the newest K (8 pieces of 16 bytes, early: their latency hides under the tile stream)
#endif
            const int blk = bt[pos / KV_BLOCK], tok = pos % KV_BLOCK;
            uint16_t* kb = a.kv_layer_w + kv_head_base(blk, 0, h);
            uint16_t* vb = a.kv_layer_w + kv_head_base(blk, 1, h);
            if (col == 0) {
                *reinterpret_cast<uint4*>(kb + k_slot(tok, 0, kg)) = knf[0];
                *reinterpret_cast<uint4*>(kb + k_slot(tok, 1, kg)) = knf[1];
            }
#ifdef T3_ATTN_V_ELEMENT_WRITES      // the round-2 form: 64 two-byte stores per (row, head)
            if (kg == 0) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) vb[v_elem(tok, 16 * dt + col)] = (uint16_t)vnew[dt];
            }
#else
            (void)vb;                              // V goes back as whole 16-byte pieces from the wave that holds the last tile (chunk loop)
#endif
        }
    } else {
        const uint16_t* qsrc = a.q + (size_t)row * D + h * HD + kg * 8;
        qfrag[0] = *reinterpret_cast<const uint4*>(qsrc); qfrag[1] = *reinterpret_cast<const uint4*>(qsrc + 32);
    }

    if constexpr (FUSE) {
        // the newest key / value wait in the wave's LDS corner until its last chunk: 12 registers less across the chunk loop,
        // which sits at the 128-VGPR budget of four waves per SIMD (one more live value and hipcc spills a K/V tile register in
        // the middle of the tile request, behind a full vmcnt(0))
        stash[0 * 64] = knf[0].x; stash[1 * 64] = knf[0].y; stash[2 * 64] = knf[0].z; stash[3 * 64] = knf[0].w;
        stash[4 * 64] = knf[1].x; stash[5 * 64] = knf[1].y; stash[6 * 64] = knf[1].z; stash[7 * 64] = knf[1].w;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) stash[(8 + dt) * 64] = vnew[dt];
        asm volatile("" ::: "memory");
    }
    T3_ASTAMP(1);                                   // prologue (q / RoPE / newest KV write) done
    auto next_of = [&](int c) -> int {
        if constexpr (!DYN) return c + NW;
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(next_chunk, 1);
        return __builtin_amdgcn_readfirstlane(nxt);
    };
    for (int c = wave; c < nc; c = next_of(c)) {
        if (c != wave) load_tiles(c);
        if (FUSE && c == nc - 1) {                  // the newest token is patched into the last tile
            knf[0] = make_uint4(stash[0 * 64], stash[1 * 64], stash[2 * 64], stash[3 * 64]);
            knf[1] = make_uint4(stash[4 * 64], stash[5 * 64], stash[6 * 64], stash[7 * 64]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) vnew[dt] = stash[(8 + dt) * 64];
            const int tc = L - 1 - c * CHUNK;
            const int tts = tc >> 4, ts = tc & 15, tss = tc >> 5, kgs = (tc & 31) >> 3, js = tc & 7;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const bool hit = (tt == tts) && (col == ts);
                kf[2 * tt].x = hit ? knf[0].x : kf[2 * tt].x; kf[2 * tt].y = hit ? knf[0].y : kf[2 * tt].y;
                kf[2 * tt].z = hit ? knf[0].z : kf[2 * tt].z; kf[2 * tt].w = hit ? knf[0].w : kf[2 * tt].w;
                kf[2 * tt + 1].x = hit ? knf[1].x : kf[2 * tt + 1].x; kf[2 * tt + 1].y = hit ? knf[1].y : kf[2 * tt + 1].y;
                kf[2 * tt + 1].z = hit ? knf[1].z : kf[2 * tt + 1].z; kf[2 * tt + 1].w = hit ? knf[1].w : kf[2 * tt + 1].w;
            }
#if defined(T3_ATTN_K_TILE_WRITE) && !defined(T3_ATTN_NOKVWRITE)
            static_assert(!T3_K_TOKEN_MAJOR, "the K tile write-back variant was written for lane-order K fragments (-DT3_K_TOKEN_MAJOR=0)");
            // Diagnostic variant: the newest K written back from the patched tile as full lines (the lanes of the token's aligned 8-token group,
            // per (dim half, dim octet) 128 contiguous bytes) instead of 8 pieces of 16 bytes from the prologue.  Bit-exact and 0.3 % SLOWER at C3
            // (20.48 against 20.54 k tok/s): these stores come late in the workgroup's life and their latency is no longer hidden.
            if ((col >> 3) == (ts >> 3)) {
                const int blkk = bt[c / CPB], cik = c % CPB;
                uint4* Kw = reinterpret_cast<uint4*>(a.kv_layer_w + kv_head_base(blkk, 0, h) + (size_t)cik * (CHUNK * HD)) + lane;
                if (tts == 0) { Kw[0 * 64] = kf[0]; Kw[1 * 64] = kf[1]; }
                else if (tts == 1) { Kw[2 * 64] = kf[2]; Kw[3 * 64] = kf[3]; }
                else if (tts == 2) { Kw[4 * 64] = kf[4]; Kw[5 * 64] = kf[5]; }
                else { Kw[6 * 64] = kf[6]; Kw[7 * 64] = kf[7]; }
            }
#endif
            if (kg == kgs) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    if (tss == 0) patch16(vf[2 * dt], js, vnew[dt]); else patch16(vf[2 * dt + 1], js, vnew[dt]);
                }
#if !defined(T3_ATTN_V_ELEMENT_WRITES) && !defined(T3_ATTN_NOKVWRITE) && !defined(T3_ATTN_NOVWRITE)
                // Paged write of the newest V: V is stored token-minor (a lane's 16 bytes = 8 consecutive tokens of one dim), so one token is 64
                // two-byte elements 16 bytes apart.  The patched pieces of this tile ARE the pool's content with the new token merged in: the 16
                // lanes of the token's group write theirs back whole -- per dim tile 256 contiguous bytes (two full lines) instead of 16 partial
                // writes.  Measured at C3 on one box: no K / V write at all 20.88 k tok/s (a bound, not a kernel), element writes (round 2) 20.34 k,
                // this form 20.54 k.
                const int blk = bt[c / CPB], ci = c % CPB;
                uint4* Vw = reinterpret_cast<uint4*>(a.kv_layer_w + kv_head_base(blk, 1, h) + (size_t)ci * (CHUNK * HD)) + lane;
                if (tss == 0) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) Vw[(2 * dt) * 64] = vf[2 * dt];
                } else {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) Vw[(2 * dt + 1) * 64] = vf[2 * dt + 1];
                }
#endif
            }
        }
#ifdef T3_ATTN_DRY
        {   // diagnostic build only: same loads, no arithmetic (measures the memory structure of this kernel)
            uint32_t x = 0;
#pragma unroll
            for (int f = 0; f < 8; ++f) x ^= kf[f].x ^ kf[f].y ^ kf[f].z ^ kf[f].w ^ vf[f].x ^ vf[f].y ^ vf[f].z ^ vf[f].w;
            if (lane == 0) { pm[c] = 0.0f; pl[c] = 1.0f; }
            po[c * 64 + lane] = __uint_as_float(x & 0x3fffffffu);
            continue;
        }
#endif
        // ---- scores on the matrix cores: D[token][col] = K[token][:] . q
        f32x4 sacc[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            sacc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            sacc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(kf[2 * tt]), as_frag(qfrag[0]), sacc[tt], 0, 0, 0);
            sacc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(kf[2 * tt + 1]), as_frag(qfrag[1]), sacc[tt], 0, 0, 0);
        }
        if (col == 0) {                             // lanes 0,16,32,48 hold every score once: rows 4 kg + r of each token tile
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                *reinterpret_cast<float4*>(sbuf + 16 * tt + 4 * kg) = make_float4(sacc[tt][0], sacc[tt][1], sacc[tt][2], sacc[tt][3]);
        }
        asm volatile("" ::: "memory");              // wave-private LDS exchange: keep the reads below the writes (the hardware keeps a wave's DS ops in order)
        // ---- softmax statistics, lane = token
        const bool live = (c * CHUNK + lane) < L;
        const float sc = live ? sbuf[lane] * 0.125f : -INFINITY;
#ifdef T3_ATTN_SHFL
        float m = sc;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        const float p = live ? t3_expf(sc - m) : 0.0f;
        float lsum = p;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) lsum = lsum + __shfl_xor(lsum, off);
#else
        const float m = wave_max_f32(sc, lane);
        const float p = live ? t3_expf(sc - m) : 0.0f;
        const float lsum = wave_bfly_add_f32(p, lane);
#endif
        pbuf[lane] = (p < 0x1p-100f) ? (uint16_t)0 : (uint16_t)f2bf(p);
        asm volatile("" ::: "memory");
        uint4 pfrag[2];                             // B operand: p[32 ts + 8 kg .. +7] as bf16, the same in all 16 columns
        pfrag[0] = *reinterpret_cast<const uint4*>(pbuf + 8 * kg);
        pfrag[1] = *reinterpret_cast<const uint4*>(pbuf + 32 + 8 * kg);
        // ---- P.V on the matrix cores: D[dim][col] = sum_token V[token][dim] * p[token]
        f32x4 oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(vf[2 * dt]), as_frag(pfrag[0]), oacc[dt], 0, 0, 0);
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(vf[2 * dt + 1]), as_frag(pfrag[1]), oacc[dt], 0, 0, 0);
        }
        if (col == 0) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<float4*>(po + c * 64 + 16 * dt + 4 * kg) = make_float4(oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]);
        }
        if (lane == 0) { pm[c] = m; pl[c] = lsum; }
        asm volatile("" ::: "memory");              // the next chunk reuses sbuf / pbuf
        if (c == wave) T3_ASTAMP(2);                // wave 0: first chunk done
    }
    T3_ASTAMP(3);                                   // wave 0: all its chunks done
    __syncthreads();
    T3_ASTAMP(4);
    if (wave == 0) {
        // fold in ascending chunk order (contract).  M and the weights w_c = exp(m_c - M) do not depend on the order: lanes compute
        // them side by side (into the m slots); the sequential part is two fmas per chunk on LDS operands.
        float M = -INFINITY;
        for (int c0 = 0; c0 < nc; c0 += 64) M = fmaxf(M, (c0 + lane < nc) ? pm[c0 + lane] : -INFINITY);
        M = wave_max_f32(M, lane);
        for (int c0 = 0; c0 < nc; c0 += 64) if (c0 + lane < nc) pm[c0 + lane] = t3_expf(pm[c0 + lane] - M);
        asm volatile("" ::: "memory");
        float l = 0.0f, o = 0.0f;
        int c = 0;
        for (; c + 4 <= nc; c += 4) {
            float wc[4], lc[4], oc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { wc[u] = pm[c + u]; lc[u] = pl[c + u]; oc[u] = po[(c + u) * 64 + lane]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                l = __builtin_fmaf(wc[u], lc[u], l);
                o = __builtin_fmaf(wc[u], oc[u], o);
            }
        }
        for (; c < nc; ++c) {
            const float w = pm[c];
            l = __builtin_fmaf(w, pl[c], l);
            o = __builtin_fmaf(w, po[c * 64 + lane], o);
        }
        a.out[(size_t)row * D + h * HD + lane] = (uint16_t)f2bf(o / l);
    }
    T3_ASTAMP(5);
}
// ------------------------------------------------------------------------------------------------
// The qkv projection INSIDE the fused decode attention launch (any decode row count).  The attention of (row, head) needs only head h's
// 192 qkv columns of row's m-tile: that seam is 3 producers -> 1 consumer, not all-to-all, and an attention launch spends its first
// ~10 us waiting for its first K/V tiles anyway.  So the launch is the attention grid (16 heads x rows workgroups), and the projection is
// cut into units (m-tile of 16 rows, head, q | k | v) = 4 n-tiles x K 1024, the shape of gemm2_kernel<1, 4, BF16, 4 waves>; the first
// workgroups (4-wave groups) to ARRIVE each take one unit by a ticket (a workgroup that holds a ticket is running and needs nothing from
// anyone, so a unit cannot wait for a workgroup that is not resident), compute it with the MFMA chains / fold order / rstd epilogue of
// gemm2_kernel, publish the 16 x 64 bf16 outputs write-through (sc1), drain, and raise the unit's flag; every workgroup then polls the
// three flags of its (m-tile, head), reads its q / k / v pieces with sc1 loads (its L1 may hold the previous layer's lines of that
// buffer) and runs attention_kernel's fused body unchanged.  Hand-off recipe: cdna_hip_programming.md Guideline 16 (R1); every polled
// word is zeroed before every launch (embed_kernel does it in the engine); the poll is bounded and leaves a code in sync[1].
// ------------------------------------------------------------------------------------------------
struct QkvInAttnArgs { const uint16_t* x; const uint4* wqkv; uint16_t* qkv; unsigned* sync; AttnArgs a; };
constexpr int QIA_FLAGS = 16;                               // sync words: [0] ticket counter, [1] give-up code, [QIA_FLAGS + u] flag of unit u
constexpr int QIA_GROUP_FLOATS = 4 * 4 * 4 * 64 + 64;       // a producing group's LDS: partials [tile 4][segment 4][reg 4][lane 64] + row statistic [4][16]
inline int qia_units(int rows) { return (rows + 15) / 16 * 48; }
inline size_t qia_lds_floats(int nw, int max_chunks) { return (size_t)max_chunks * 66 + (size_t)nw * (96 + 12 * 64) + (size_t)(nw / 4) * QIA_GROUP_FLOATS + 4; }
typedef __attribute__((address_space(1))) unsigned gu32_t;
typedef __attribute__((address_space(1))) unsigned long long gu64_t;
typedef __attribute__((address_space(1))) unsigned short gu16_t;
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 4 : 2) void qkv_in_attention_kernel(QkvInAttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) float part[];
    const AttnArgs& a = p.a;
    float* pm = part; float* pl = part + a.max_chunks; float* po = part + 2 * a.max_chunks;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* sbuf = part + 66 * a.max_chunks + wave * 96;
    uint32_t* stash = reinterpret_cast<uint32_t*>(part + 66 * a.max_chunks + NW * 96) + wave * (12 * 64) + lane;
    uint16_t* pbuf = reinterpret_cast<uint16_t*>(sbuf + 64);
    const int grp = wave >> 2, sseg = wave & 3;             // 4-wave group (one per workgroup at NW = 4), K segment of the wave in a unit
    float* gpart = part + 66 * a.max_chunks + NW * (96 + 12 * 64) + grp * QIA_GROUP_FLOATS;
    float* growsum = gpart + 4 * 4 * 4 * 64;
    int* tk = reinterpret_cast<int*>(part + 66 * a.max_chunks + NW * (96 + 12 * 64) + (NW / 4) * QIA_GROUP_FLOATS);
    const int h = blockIdx.x, row = blockIdx.y;
    const int col = lane & 15, kg = lane >> 4;
    // ---- ticket: which unit, if any, this 4-wave group computes
    if (sseg == 0 && lane == 0) tk[grp] = (int)atomicAdd(p.sync, 1u);
    const int* rec = a.rowrec + (size_t)row * a.row_stride;
    const int L = rec[1] + 1;
    const int nc = (L + CHUNK - 1) / CHUNK;
    const int* bt = rec + ROW_HDR;
    constexpr int CPB = KV_BLOCK / CHUNK;
    __syncthreads();
    const int unit = __builtin_amdgcn_readfirstlane(tk[grp]);
    const bool producer = unit < ((a.rows + 15) >> 4) * 48;

    uint4 kf[8], vf[8];
    auto load_tiles = [&](int c) {
        const int blk = bt[c / CPB], ci = c % CPB;
        const uint4* Kp = reinterpret_cast<const uint4*>(a.kv_layer + kv_head_base(blk, 0, h) + (size_t)ci * (CHUNK * HD)) + k_lane_piece(lane);
        const uint4* Vp = reinterpret_cast<const uint4*>(a.kv_layer + kv_head_base(blk, 1, h) + (size_t)ci * (CHUNK * HD)) + lane;
        const int npool = L - 1 - c * CHUNK;
        if (npool >= CHUNK) {
#pragma unroll
            for (int f = 0; f < 8; ++f) kf[f] = ld_nt(Kp + f * 64);
#pragma unroll
            for (int f = 0; f < 8; ++f) vf[f] = ld_nt(Vp + f * 64);
        } else {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                if (16 * tt < npool) { kf[2 * tt] = ld_nt(Kp + (2 * tt) * 64); kf[2 * tt + 1] = ld_nt(Kp + (2 * tt + 1) * 64); }
                else { kf[2 * tt] = make_uint4(0, 0, 0, 0); kf[2 * tt + 1] = make_uint4(0, 0, 0, 0); }
            }
#pragma unroll
            for (int ts = 0; ts < 2; ++ts)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    if (32 * ts < npool) vf[2 * dt + ts] = ld_nt(Vp + (2 * dt + ts) * 64);
                    else vf[2 * dt + ts] = make_uint4(0, 0, 0, 0);
                }
        }
    };
    // ---- producers, phase 1: the unit's 16 rows x 64 columns, one MFMA chain per (tile, K segment)
    const int u_mt = unit / 48, u_h = (unit % 48) / 3, u_part = unit % 3;
    if (producer) {
        int m = u_mt * 16 + col; m = m < a.rows ? m : a.rows - 1;     // padded rows re-read the last row; their outputs are dropped
        const uint4* xp = reinterpret_cast<const uint4*>(p.x + (size_t)m * D + sseg * 256 + kg * 8);
        uint4 af[8];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) af[kb] = xp[kb * 4];
        f32x4 ss = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nt = u_part * 64 + u_h * 4 + j;                // packed n-tile of the [3072][1024] matrix
            const uint4* wp = p.wqkv + ((size_t)nt * 32 + sseg * 8) * 64 + lane;
            uint4 w[8];
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) w[kb] = ld_nt(wp + kb * 64);
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af[kb]), as_frag(w[kb]), acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) gpart[((j * 4 + sseg) * 4 + r) * 64 + lane] = acc[r];
        }
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) ss = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(af[kb]), as_frag(af[kb]), ss, 0, 0, 0);   // diagonal = sum of squares of the segment
        const int r = col & 3;
        const float d = r == 0 ? ss[0] : r == 1 ? ss[1] : r == 2 ? ss[2] : ss[3];
        if ((col >> 2) == kg) growsum[sseg * 16 + col] = d;
    }
    __syncthreads();
    // ---- producers, phase 2: fold ((s0 + s1) + s2) + s3, rstd, bf16; published write-through, drained
    if (producer) {
        const int t = tid & 255, r16 = t >> 4, cq = t & 15, j = cq >> 2, c0 = 4 * (cq & 3);
        const int m = u_mt * 16 + r16;
        if (m < a.rows) {
            const int o = (r16 & 3) * 64 + 16 * (r16 >> 2) + c0;
            const float* pj = gpart + (size_t)j * 4 * 4 * 64;
            const float4 s0 = *reinterpret_cast<const float4*>(pj + o), s1 = *reinterpret_cast<const float4*>(pj + 256 + o),
                         s2 = *reinterpret_cast<const float4*>(pj + 512 + o), s3 = *reinterpret_cast<const float4*>(pj + 768 + o);
            const float ssum = ((growsum[r16] + growsum[16 + r16]) + growsum[32 + r16]) + growsum[48 + r16];
            const float rstd = 1.0f / sqrtf(ssum * (1.0f / 1024.0f) + 1e-5f);
            const uint32_t b0 = f2bf((((s0.x + s1.x) + s2.x) + s3.x) * rstd), b1 = f2bf((((s0.y + s1.y) + s2.y) + s3.y) * rstd),
                           b2 = f2bf((((s0.z + s1.z) + s2.z) + s3.z) * rstd), b3 = f2bf((((s0.w + s1.w) + s2.w) + s3.w) * rstd);
            gu64_t* dst = (gu64_t*)(p.qkv + (size_t)m * QKV + u_part * D + u_h * HD + j * 16 + c0);
            __hip_atomic_store(dst, (unsigned long long)(b0 | (b1 << 16)) | ((unsigned long long)(b2 | (b3 << 16)) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains before the flag goes up
    }
    __syncthreads();
    if (producer && sseg == 0 && lane == 0) __hip_atomic_store((gu32_t*)(p.sync + QIA_FLAGS + unit), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the wave's first K/V tile: requested behind the unit so that its 64 registers never coexist with the unit's operands (a workgroup
    // without a unit gets here two empty barriers after its ticket)
    if (wave < nc) load_tiles(wave);
    // ---- every workgroup: wait for the q | k | v units of its (m-tile, head)
    if (tid == 0) {
        gu32_t* fl = (gu32_t*)(p.sync + QIA_FLAGS + (row >> 4) * 48 + h * 3);
        unsigned spins = 0;
        for (;;) {
            const unsigned f0 = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), f1 = __hip_atomic_load(fl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                           f2 = __hip_atomic_load(fl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((f0 & f1 & f2) == 1u) break;
            if (++spins > (1u << 22)) { __hip_atomic_store((gu32_t*)(p.sync + 1), 0xdead0000u | (unsigned)(row & 0xffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // give up: wrong ids, not a hang
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    // ---- attention_kernel's fused body; q / k / v pieces by sc1 loads
    uint4 qfrag[2], knf[2];
    uint32_t vnew[4];
    {
        const int pos = L - 1;
        const uint16_t* src = p.qkv + (size_t)row * QKV + h * HD;
        auto ld16 = [&](const uint16_t* q_) {
            const unsigned long long lo = __hip_atomic_load((gu64_t*)q_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi = __hip_atomic_load((gu64_t*)(q_ + 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
        };
        const uint4 q1 = ld16(src + kg * 8), q2 = ld16(src + 32 + kg * 8), k1 = ld16(src + D + kg * 8), k2 = ld16(src + D + 32 + kg * 8);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vnew[dt] = __hip_atomic_load((gu16_t*)(src + 2 * D + 16 * dt + col), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float c[8], s[8];
        {
            const float4* cp = reinterpret_cast<const float4*>(a.cos_t + (size_t)pos * 32 + kg * 8);
            const float4* sp = reinterpret_cast<const float4*>(a.sin_t + (size_t)pos * 32 + kg * 8);
            const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
            c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
            s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w; s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w;
        }
        rope8(q1, q2, c, s, qfrag[0], qfrag[1]);
        rope8(k1, k2, c, s, knf[0], knf[1]);
        if (wave == 0) {                            // paged write of the newest K (8 pieces of 16 bytes); V goes back from the last tile
            const int blk = bt[pos / KV_BLOCK], tok = pos % KV_BLOCK;
            uint16_t* kb = a.kv_layer_w + kv_head_base(blk, 0, h);
            if (col == 0) {
                *reinterpret_cast<uint4*>(kb + k_slot(tok, 0, kg)) = knf[0];
                *reinterpret_cast<uint4*>(kb + k_slot(tok, 1, kg)) = knf[1];
            }
        }
        stash[0 * 64] = knf[0].x; stash[1 * 64] = knf[0].y; stash[2 * 64] = knf[0].z; stash[3 * 64] = knf[0].w;
        stash[4 * 64] = knf[1].x; stash[5 * 64] = knf[1].y; stash[6 * 64] = knf[1].z; stash[7 * 64] = knf[1].w;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) stash[(8 + dt) * 64] = vnew[dt];
        asm volatile("" ::: "memory");
    }
    for (int c = wave; c < nc; c += NW) {
        if (c != wave) load_tiles(c);
        if (c == nc - 1) {                          // the newest token is patched into the last tile
            knf[0] = make_uint4(stash[0 * 64], stash[1 * 64], stash[2 * 64], stash[3 * 64]);
            knf[1] = make_uint4(stash[4 * 64], stash[5 * 64], stash[6 * 64], stash[7 * 64]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) vnew[dt] = stash[(8 + dt) * 64];
            const int tc = L - 1 - c * CHUNK;
            const int tts = tc >> 4, ts = tc & 15, tss = tc >> 5, kgs = (tc & 31) >> 3, js = tc & 7;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const bool hit = (tt == tts) && (col == ts);
                kf[2 * tt].x = hit ? knf[0].x : kf[2 * tt].x; kf[2 * tt].y = hit ? knf[0].y : kf[2 * tt].y;
                kf[2 * tt].z = hit ? knf[0].z : kf[2 * tt].z; kf[2 * tt].w = hit ? knf[0].w : kf[2 * tt].w;
                kf[2 * tt + 1].x = hit ? knf[1].x : kf[2 * tt + 1].x; kf[2 * tt + 1].y = hit ? knf[1].y : kf[2 * tt + 1].y;
                kf[2 * tt + 1].z = hit ? knf[1].z : kf[2 * tt + 1].z; kf[2 * tt + 1].w = hit ? knf[1].w : kf[2 * tt + 1].w;
            }
            if (kg == kgs) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    if (tss == 0) patch16(vf[2 * dt], js, vnew[dt]); else patch16(vf[2 * dt + 1], js, vnew[dt]);
                }
                const int blk = bt[c / CPB], ci = c % CPB;
                uint4* Vw = reinterpret_cast<uint4*>(a.kv_layer_w + kv_head_base(blk, 1, h) + (size_t)ci * (CHUNK * HD)) + lane;
                if (tss == 0) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) Vw[(2 * dt) * 64] = vf[2 * dt];
                } else {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) Vw[(2 * dt + 1) * 64] = vf[2 * dt + 1];
                }
            }
        }
        f32x4 sacc[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            sacc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            sacc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(kf[2 * tt]), as_frag(qfrag[0]), sacc[tt], 0, 0, 0);
            sacc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(kf[2 * tt + 1]), as_frag(qfrag[1]), sacc[tt], 0, 0, 0);
        }
        if (col == 0) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                *reinterpret_cast<float4*>(sbuf + 16 * tt + 4 * kg) = make_float4(sacc[tt][0], sacc[tt][1], sacc[tt][2], sacc[tt][3]);
        }
        asm volatile("" ::: "memory");
        const bool live = (c * CHUNK + lane) < L;
        const float sc = live ? sbuf[lane] * 0.125f : -INFINITY;
        const float m = wave_max_f32(sc, lane);
        const float pr = live ? t3_expf(sc - m) : 0.0f;
        const float lsum = wave_bfly_add_f32(pr, lane);
        pbuf[lane] = (pr < 0x1p-100f) ? (uint16_t)0 : (uint16_t)f2bf(pr);
        asm volatile("" ::: "memory");
        uint4 pfrag[2];
        pfrag[0] = *reinterpret_cast<const uint4*>(pbuf + 8 * kg);
        pfrag[1] = *reinterpret_cast<const uint4*>(pbuf + 32 + 8 * kg);
        f32x4 oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(vf[2 * dt]), as_frag(pfrag[0]), oacc[dt], 0, 0, 0);
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(vf[2 * dt + 1]), as_frag(pfrag[1]), oacc[dt], 0, 0, 0);
        }
        if (col == 0) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<float4*>(po + c * 64 + 16 * dt + 4 * kg) = make_float4(oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]);
        }
        if (lane == 0) { pm[c] = m; pl[c] = lsum; }
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (wave == 0) {                                // fold in ascending chunk order (contract), as attention_kernel does
        float M = -INFINITY;
        for (int c0 = 0; c0 < nc; c0 += 64) M = fmaxf(M, (c0 + lane < nc) ? pm[c0 + lane] : -INFINITY);
        M = wave_max_f32(M, lane);
        for (int c0 = 0; c0 < nc; c0 += 64) if (c0 + lane < nc) pm[c0 + lane] = t3_expf(pm[c0 + lane] - M);
        asm volatile("" ::: "memory");
        float l = 0.0f, o = 0.0f;
        for (int c = 0; c < nc; ++c) {
            const float wgt = pm[c];
            l = __builtin_fmaf(wgt, pl[c], l);
            o = __builtin_fmaf(wgt, po[c * 64 + lane], o);
        }
        a.out[(size_t)row * D + h * HD + lane] = (uint16_t)f2bf(o / l);
    }
}
int qkv_in_attention_sync_words(int rows) { return QIA_FLAGS + qia_units(rows); }
bool qkv_in_attention_fits(int rows, int max_chunks) {
    const int nw = rows <= 8 ? 8 : 4;
    // 4-wave form: four workgroups per CU must still fit (the attention's occupancy); 8-wave form: two
    return rows >= 2 && qia_lds_floats(nw, max_chunks) * sizeof(float) <= (nw == 4 ? 40 * 1024 : 80 * 1024);
}
hipError_t launch_qkv_in_attention(const uint16_t* x, const uint4* wqkv, uint16_t* qkv, unsigned* sync, const AttnArgs& a, hipStream_t s) {
    if (!qkv_in_attention_fits(a.rows, a.max_chunks)) return hipErrorInvalidValue;
    const int nw = a.rows <= 8 ? 8 : 4;
    const size_t lds = qia_lds_floats(nw, a.max_chunks) * sizeof(float);
    QkvInAttnArgs p{x, wqkv, qkv, sync, a};
    static size_t raised[MAX_DEVICES] = {};
    size_t& have = raised[cur_device()];
    if (nw == 8 && lds > 64 * 1024 && lds > have) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qkv_in_attention_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        have = lds;
    }
    if (nw == 8) hipLaunchKernelGGL(qkv_in_attention_kernel<8>, dim3(H, a.rows), dim3(512), lds, s, p);
    else hipLaunchKernelGGL(qkv_in_attention_kernel<4>, dim3(H, a.rows), dim3(256), lds, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Prefill form of the same attention: one workgroup per (head, 16 consecutive rows of the launch).  The decode kernel replicates
// one q over the 16 B-operand columns of the MFMA; here the 16 columns are 16 different rows (positions) of one stream, so a K/V
// tile is read once per 16 rows instead of once per row.  An MFMA output column depends on its own B column only, so every
// row's numbers are those of the per-row kernel: same score and P.V chains, the butterfly sum of the 64 probabilities rebuilt
// level by level on the (token = 16 tt + 4 kg + r) register layout, per-chunk partials folded in ascending chunk order.
// Rows of different streams in one tile (prompt boundaries, decode rows) are served segment by segment.
// ------------------------------------------------------------------------------------------------
constexpr int TILE_OS = 68;             // floats per (chunk, row) line of partial outputs (64 + padding against LDS bank conflicts)
constexpr int TILE_PS = 72;             // bf16 per row of a wave's probability image (144 B: 16-byte aligned, conflict-free enough)
__global__ __launch_bounds__(256, 2) void attention_tile_kernel(AttnArgs a, int row_base, int chunks_cap) {
    extern __shared__ __attribute__((aligned(16))) float part[];   // [cap][16][TILE_OS] o | [cap][16] m | [cap][16] l | per wave: [16][TILE_PS] bf16 p
    float* po = part; float* pm = part + (size_t)chunks_cap * 16 * TILE_OS; float* pl = pm + chunks_cap * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint16_t* pimg = reinterpret_cast<uint16_t*>(pl + chunks_cap * 16) + wave * (16 * TILE_PS);
    const int col = lane & 15, kg = lane >> 4;
    const int h = blockIdx.x, r0 = row_base + blockIdx.y * 16;
    const int nrows = min(16, a.rows - r0);
    const int myrow = r0 + min(col, nrows - 1);
    const int* myrec = a.rowrec + (size_t)myrow * a.row_stride;
    const int my_stream = myrec[0], my_L = myrec[1] + 1;
    uint4 qfrag[2];                                 // B operand: column col = row r0 + col
    {
        const uint16_t* qsrc = a.q + (size_t)myrow * D + h * HD + kg * 8;
        qfrag[0] = *reinterpret_cast<const uint4*>(qsrc); qfrag[1] = *reinterpret_cast<const uint4*>(qsrc + 32);
    }
    constexpr int CPB = KV_BLOCK / CHUNK;
    const unsigned valid = nrows >= 16 ? 0xffffu : ((1u << nrows) - 1u);
    unsigned done = 0;
    while ((done & valid) != valid) {               // one pass per stream present in the tile (wave-uniform control flow)
        const int lead = __builtin_ctz(~done & valid);
        const int lead_stream = __builtin_amdgcn_readlane(my_stream, lead);
        const bool in_seg = col < nrows && my_stream == lead_stream;
        const unsigned seg = (unsigned)(__ballot(in_seg) & 0xffffull);
        int Lmax = in_seg ? my_L : 0;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) Lmax = max(Lmax, __shfl_xor(Lmax, off));
        Lmax = __builtin_amdgcn_readfirstlane(Lmax);
        const int nc = min((Lmax + CHUNK - 1) / CHUNK, chunks_cap);
        const int* bt = a.rowrec + (size_t)(r0 + lead) * a.row_stride + ROW_HDR;
        for (int c = wave; c < nc; c += 4) {
            uint4 kf[8], vf[8];
            {
                const int blk = bt[c / CPB], ci = c % CPB;
                const uint4* Kp = reinterpret_cast<const uint4*>(a.kv_layer + kv_head_base(blk, 0, h) + (size_t)ci * (CHUNK * HD)) + k_lane_piece(lane);
                const uint4* Vp = reinterpret_cast<const uint4*>(a.kv_layer + kv_head_base(blk, 1, h) + (size_t)ci * (CHUNK * HD)) + lane;
#pragma unroll
                for (int f = 0; f < 8; ++f) kf[f] = Kp[f * 64];
#pragma unroll
                for (int f = 0; f < 8; ++f) vf[f] = Vp[f * 64];
            }
            f32x4 sacc[4];                          // sacc[tt][r] = score of token 16 tt + 4 kg + r of the chunk for row col
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                sacc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                sacc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(kf[2 * tt]), as_frag(qfrag[0]), sacc[tt], 0, 0, 0);
                sacc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(kf[2 * tt + 1]), as_frag(qfrag[1]), sacc[tt], 0, 0, 0);
            }
            const int nlive = my_L - c * CHUNK;     // tokens of this chunk the row sees (causal), may be <= 0 or >= 64
            float sc[4][4], m = -INFINITY;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sc[tt][r] = (16 * tt + 4 * kg + r) < nlive ? sacc[tt][r] * 0.125f : -INFINITY;
                    m = fmaxf(m, sc[tt][r]);
                }
            m = fmaxf(m, lane_xor<16>(m, lane)); m = fmaxf(m, lane_xor<32>(m, lane));
            float pr[4][4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[tt][r] = (16 * tt + 4 * kg + r) < nlive ? t3_expf(sc[tt][r] - m) : 0.0f;
            // the contract's butterfly sum over the 64 tokens (partners t ^ 32, 16, 8, 4, 2, 1): bits 5, 4 of the token are tt,
            // bits 3, 2 are kg (lane bits 5, 4), bits 1, 0 are r
            float b4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[r] = (pr[0][r] + pr[2][r]) + (pr[1][r] + pr[3][r]);
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[r] = b4[r] + lane_xor<32>(b4[r], lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[r] = b4[r] + lane_xor<16>(b4[r], lane);
            const float lsum = (b4[0] + b4[2]) + (b4[1] + b4[3]);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                uint32_t pk[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (pr[tt][r] < 0x1p-100f) ? 0u : (uint32_t)f2bf(pr[tt][r]);
                *reinterpret_cast<uint2*>(pimg + col * TILE_PS + 16 * tt + 4 * kg) = make_uint2(pk[0] | (pk[1] << 16), pk[2] | (pk[3] << 16));
            }
            asm volatile("" ::: "memory");          // wave-private LDS exchange (a wave's DS operations execute in order)
            uint4 pfrag[2];
            pfrag[0] = *reinterpret_cast<const uint4*>(pimg + col * TILE_PS + 8 * kg);
            pfrag[1] = *reinterpret_cast<const uint4*>(pimg + col * TILE_PS + 32 + 8 * kg);
            asm volatile("" ::: "memory");
            if (kg == 0) { pm[c * 16 + col] = m; pl[c * 16 + col] = lsum; }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 oacc = (f32x4){0.f, 0.f, 0.f, 0.f};
                oacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(vf[2 * dt]), as_frag(pfrag[0]), oacc, 0, 0, 0);
                oacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(vf[2 * dt + 1]), as_frag(pfrag[1]), oacc, 0, 0, 0);
                *reinterpret_cast<float4*>(po + (size_t)(c * 16 + col) * TILE_OS + 16 * dt + 4 * kg) = make_float4(oacc[0], oacc[1], oacc[2], oacc[3]);
            }
        }
        __syncthreads();
        {   // fold: thread = (row, 4 dims), ascending chunk order over the row's own chunks
            const int row = tid >> 4, d4 = (tid & 15) * 4;
            if ((seg >> row) & 1u) {
                const int L = a.rowrec[(size_t)(r0 + row) * a.row_stride + 1] + 1;
                const int ncr = min((L + CHUNK - 1) / CHUNK, chunks_cap);
                float M = -INFINITY;
                for (int c = 0; c < ncr; ++c) M = fmaxf(M, pm[c * 16 + row]);
                float l = 0.0f, o[4] = {0.f, 0.f, 0.f, 0.f};
                for (int c = 0; c < ncr; ++c) {
                    const float w = t3_expf(pm[c * 16 + row] - M);
                    const float4 oc = *reinterpret_cast<const float4*>(po + (size_t)(c * 16 + row) * TILE_OS + d4);
                    l = __builtin_fmaf(w, pl[c * 16 + row], l);
                    o[0] = __builtin_fmaf(w, oc.x, o[0]); o[1] = __builtin_fmaf(w, oc.y, o[1]);
                    o[2] = __builtin_fmaf(w, oc.z, o[2]); o[3] = __builtin_fmaf(w, oc.w, o[3]);
                }
                const uint32_t lo = (uint32_t)f2bf(o[0] / l) | ((uint32_t)f2bf(o[1] / l) << 16), hi = (uint32_t)f2bf(o[2] / l) | ((uint32_t)f2bf(o[3] / l) << 16);
                *reinterpret_cast<uint2*>(a.out + (size_t)(r0 + row) * D + h * HD + d4) = make_uint2(lo, hi);
            }
        }
        done |= seg;
        if ((done & valid) != valid) __syncthreads();      // the next segment reuses the partial slots
    }
}
hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
    if (a.rows <= 0) return hipSuccess;
    static int nw_env = -1, nt = 0, tile_on = 1;
    if (nw_env < 0) { const char* e = getenv("T3_ATTN_WAVES"); nw_env = e ? atoi(e) : 0; const char* t = getenv("T3_ATTN_NT"); nt = t ? atoi(t) : 1; const char* p = getenv("T3_ATTN_TILE"); tile_on = p ? atoi(p) : 1; }
    // unfused form: rows [tile_from, rows) are prefill rows (runs of consecutive positions of a stream): 16 rows per workgroup.
    // The tile kernel keeps every chunk's partials of its 16 rows in LDS (4.5 KiB per chunk): beyond 35 chunks (a prefill context
    // over 2 240 tokens) that no longer fits the 160 KiB of a gfx950 CU, and such rows take the per-row kernel below (34 KiB at
    // max_model_len 8192), which computes the same numbers.
    const int cap = a.tile_chunks > 0 ? min(a.tile_chunks, a.max_chunks) : a.max_chunks;
    const size_t lds_t = ((size_t)cap * 16 * (TILE_OS + 2)) * sizeof(float) + (size_t)4 * 16 * TILE_PS * 2;
    constexpr size_t LDS_PER_CU = 160 * 1024;
    if (!a.qkv && tile_on && a.tile_from >= 0 && a.tile_from < a.rows && lds_t <= LDS_PER_CU) {
        static size_t raised[MAX_DEVICES] = {};
        size_t& have = raised[cur_device()];
        if (have < 64 * 1024) have = 64 * 1024;
        if (lds_t > have) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t);
            if (e != hipSuccess) return e;
            have = lds_t;
        }
        hipLaunchKernelGGL(attention_tile_kernel, dim3(H, (a.rows - a.tile_from + 15) / 16), dim3(256), lds_t, s, a, a.tile_from, cap);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess || a.tile_from == 0) return e;
        AttnArgs head = a; head.rows = a.tile_from; head.tile_from = -1;
        return launch_attention(head, s);
    }
    // 4 waves per (row, head) fill the chip from 16 rows on (16 heads x 16 rows x 4 waves = 4 waves per CU); below that the
    // launch is latency-bound and 8 waves halve the number of sequential 64-token chunks per wave (B = 1: 2 rows -> 32 workgroups)
    const int nw = a.force_waves == 4 || a.force_waves == 8 ? a.force_waves : nw_env == 4 || nw_env == 8 ? nw_env : (a.rows <= 8 ? 8 : 4);
    const dim3 grid(H, a.rows);
    const bool fuse = a.qkv != nullptr;
    const size_t lds = ((size_t)a.max_chunks * 66 + (size_t)(nw == 8 ? 8 : 4) * (96 + (fuse ? 12 * 64 : 0)) + 4) * sizeof(float);     // + the chunk counter
#define T3_ATTN(NW, NTF, FU) launch_k((attention_kernel<NW, NTF, FU>), grid, dim3(NW * 64), lds, s, a)
    if (nw == 8) { if (fuse) { if (nt) T3_ATTN(8, true, true); else T3_ATTN(8, false, true); } else { if (nt) T3_ATTN(8, true, false); else T3_ATTN(8, false, false); } }
    else { if (fuse) { if (nt) T3_ATTN(4, true, true); else T3_ATTN(4, false, true); } else { if (nt) T3_ATTN(4, true, false); else T3_ATTN(4, false, false); } }
#undef T3_ATTN
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
constexpr int SSCR = 32;                        // scratch words behind the weights: [SWV] wave partials | [24] the drawn token

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
    extern __shared__ __attribute__((aligned(16))) unsigned long long sw[];   // [SLOTS] weights | [SSCR] scratch | [256] histogram copies
    unsigned long long* scr = sw + SLOTS;
    unsigned long long* psum = scr + SSCR;
    const int tid = threadIdx.x, u = blockIdx.x;
    const int slot = a.sel[u].x;
    const uint32_t step = (uint32_t)a.sel[u].y;
    const T3Sampling sp = a.sp[slot];
    const uint16_t* lc = a.logits + (size_t)(2 * u) * a.ldl;
    const uint16_t* lu = lc + a.ldl;
    uint16_t* counts = a.counts + (size_t)slot * VPAD;
    float* xs = reinterpret_cast<float*>(sw);     // phase A: xs[v] (fp32) lives in the low half of slot v

    // ---- phase A: CFG, penalties.  16-byte loads of 8 logits / counts, all issued before the first use
    // (a scalar loop here is a chain of dependent memory latencies, not bandwidth).
    const bool greedy = sp.temperature < 1e-5f;
    float mx = -INFINITY;
    unsigned long long best = 0;
    constexpr int NVEC = VPAD / 8, VIT = (NVEC + STH - 1) / STH;
    static_assert(VPAD % 8 == 0 && SLOTS >= VPAD, "sampler vector layout");
    uint4 c4[VIT], u4[VIT], n4[VIT];
#pragma unroll
    for (int k = 0; k < VIT; ++k) {
        const int vi = tid + STH * k;
        if (vi < NVEC) {
            c4[k] = reinterpret_cast<const uint4*>(lc)[vi];
            u4[k] = reinterpret_cast<const uint4*>(lu)[vi];
            n4[k] = reinterpret_cast<const uint4*>(counts)[vi];
        }
    }
    for (int v = VPAD + tid; v < SLOTS; v += STH) xs[2 * v] = -INFINITY;
    T3_CLK(1);
#pragma unroll
    for (int k = 0; k < VIT; ++k) {
        const int vi = tid + STH * k;
        if (vi >= NVEC) continue;
        const uint32_t cw[4] = {c4[k].x, c4[k].y, c4[k].z, c4[k].w}, uw[4] = {u4[k].x, u4[k].y, u4[k].z, u4[k].w},
                       nw[4] = {n4[k].x, n4[k].y, n4[k].z, n4[k].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int v = vi * 8 + j;
            const int sh = (j & 1) * 16;
            float x = -INFINITY;
            if (v < V) {
                const float c = bf2f((uint16_t)(cw[j >> 1] >> sh)), un = bf2f((uint16_t)(uw[j >> 1] >> sh));
                const float d = rbf(c - un);
                const float e = rbf(a.cfg * d);
                x = rbf(c + e);
                if (a.dbg) a.dbg[(size_t)slot * V + v] = x;
                const uint32_t cnt = (nw[j >> 1] >> sh) & 0xFFFFu;
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
            xs[2 * v] = x;
        }
    }
    T3_CLK(2);
    int token;
    if (greedy) {
        best = block_max_u64(best, scr);
        token = (int)(0xFFFFFFFFu - (uint32_t)best);
    } else {
        mx = block_max_f32(mx, scr);
        // ---- weights (block_max above is the barrier between writing xs and overwriting it slot by slot)
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            const int v = tid + STH * k;
            const float x = xs[2 * v];
            unsigned long long w = 0;
            if (v < V) w = (unsigned long long)(t3_expf(x - mx) * 4294967296.0f);
            sw[v] = w;
        }
        __syncthreads();
        T3_CLK(3);
        // From here on a thread owns SPT consecutive vocabulary entries and keeps them in registers.
        const int v0 = tid * SPT;
        unsigned long long wr[SPT];
#pragma unroll
        for (int i = 0; i < SPT; ++i) wr[i] = sw[v0 + i];
        unsigned long long wmax = 0;
#pragma unroll
        for (int i = 0; i < SPT; ++i) wmax = max_u64(wmax, wr[i]);
        wmax = block_max_u64(wmax, scr);
        if (sp.min_p > 0.0f) {
            const double thr = (double)sp.min_p * (double)wmax;
#pragma unroll
            for (int i = 0; i < SPT; ++i) if ((double)wr[i] < thr) wr[i] = 0;
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
        }
        T3_CLK(4);
        if (sp.top_p < 1.0f) {
            unsigned long long W = 0;
#pragma unroll
            for (int i = 0; i < SPT; ++i) W += wr[i];
            W = block_sum_u64(W, scr);
            const unsigned long long Tm = (unsigned long long)((1.0 - (double)sp.top_p) * (double)W);
            // lo = max{mid : sum of weights < mid is <= Tm} is the weight value at which the ascending cumulative
            // mass first exceeds Tm.  Radix descent on the value: one round over the octaves, then 6 bits per
            // round; per round an LDS histogram of masses (exact integer sums, so order-free) and a 64-bin
            // wave scan that every wave repeats for itself.
            T3_CLK(5);
            // The histogram is kept in HC copies, one per lane group (lane & (HC - 1)): the masses of a round fall into a handful of the 64
            // bins (a softmax's weights share few octaves), and 64-bit LDS atomics of one wave instruction onto the same bin serialise.  The sums
            // are exact integers, so the copies add up to the same histogram in any order.
            constexpr int HC = 4;
            unsigned long long* hist = psum;                 // [HC][64]
            const int lane = tid & 63;
            unsigned long long* myhist = hist + (lane & (HC - 1)) * 64;
            unsigned long long lo = wmax, base = 0, prefix = 0;
            int nb = -1;                                     // bits of the value still undetermined; -1: octave round
            for (;;) {
                __syncthreads();
                if (tid < HC * 64) hist[tid] = 0;
                __syncthreads();
                if (nb < 0) {
#pragma unroll
                    for (int i = 0; i < SPT; ++i) if (wr[i]) atomicAdd(&myhist[64 - __clzll((long long)wr[i])], wr[i]);
                } else {
                    const int shift = nb > 6 ? nb - 6 : 0;
                    const unsigned long long msk = (1ull << (nb - shift)) - 1;
#pragma unroll
                    for (int i = 0; i < SPT; ++i) if ((wr[i] >> nb) == prefix) atomicAdd(&myhist[(wr[i] >> shift) & msk], wr[i]);
                }
                __syncthreads();
                const unsigned long long own = (hist[lane] + hist[64 + lane]) + (hist[128 + lane] + hist[192 + lane]);
                const unsigned long long c = wave_scan_u64(own, lane);      // inclusive scan over the 64 bins
                const unsigned long long over = __ballot(base + c > Tm);
                if (!over) break;                            // only in the octave round, when Tm == W: lo = wmax
                const int b = __ffsll((long long)over) - 1;
                base += readlane_u64(c - own, b);
                if (nb < 0) { prefix = 1; nb = b - 1; }
                else { const int shift = nb > 6 ? nb - 6 : 0; prefix = (prefix << (nb - shift)) | (unsigned long long)b; nb = shift; }
                if (nb == 0) { lo = prefix; break; }
            }
            T3_CLK(6);
            if (lo > 0) {
                unsigned long long below = 0, my_ties = 0;
#pragma unroll
                for (int i = 0; i < SPT; ++i) { below += (wr[i] < lo) ? wr[i] : 0; my_ties += (wr[i] == lo) ? 1 : 0; }
                below = block_sum_u64(below, scr);
                unsigned long long ties;
                const unsigned long long incl = block_scan_u64(my_ties, scr, &ties);
                unsigned long long r = (Tm - below) / lo;
                if (lo == wmax && r > ties - 1) r = ties - 1;
                if (r > ties) r = ties;
                // ties are dropped highest index first
                unsigned long long above = ties - incl;          // ties owned by higher threads
#pragma unroll
                for (int i = SPT - 1; i >= 0; --i) {
                    if (wr[i] < lo) wr[i] = 0;
                    else if (wr[i] == lo) { if (above < r) wr[i] = 0; ++above; }
                }
            }
        }
        T3_CLK(7);
        // ---- draw
        unsigned long long mine = 0;
#pragma unroll
        for (int i = 0; i < SPT; ++i) mine += wr[i];
        unsigned long long Wk;
        const unsigned long long excl = block_scan_u64(mine, scr, &Wk) - mine;
        T3_CLK(8);
        uint32_t rnd[4];
        philox4x32_10(step, (uint32_t)sp.uid, (uint32_t)(sp.uid >> 32), 0u, (uint32_t)sp.seed, (uint32_t)(sp.seed >> 32), rnd);
        const unsigned long long uu = ((unsigned long long)rnd[1] << 32) | rnd[0];
        const unsigned long long target = __umul64hi(uu, Wk);
        int* tokp = reinterpret_cast<int*>(scr + 24);
        // No thread owns the target when there is no mass at all (NaN logits: a checkpoint or conditioning with NaN / Inf makes
        // every weight 0): the draw then falls back to the stop id instead of whatever the LDS word held.  The write is ordered
        // before the owner's by the barriers inside block_scan_u64 above (scr + 24 is not one of its SWV words).
        if (Wk == 0 && tid == 0) *tokp = (sp.stop_token >= 0 && sp.stop_token < V) ? sp.stop_token : 0;
        if (mine > 0 && target >= excl && target < excl + mine) {
            unsigned long long cum = excl; int found = -1;      // first entry whose running mass passes the target
#pragma unroll
            for (int i = 0; i < SPT; ++i) { cum += wr[i]; if (found < 0 && cum > target) found = v0 + i; }
            *tokp = found;
        }
        __syncthreads();
        token = *tokp;
        T3_CLK(9);
    }
    if (tid == 0) {
        token = token < 0 ? 0 : (token >= V ? V - 1 : token);       // counts[] / speech_emb[] are indexed with it
        a.out_tok[u] = token;
        if (a.out_tok_host) a.out_tok_host[u] = token;
        if (a.hist && (int)step < a.hist_cap) a.hist[(size_t)slot * a.hist_cap + step] = token;      // the utterance's ids stay on the device
        const uint16_t cnt = counts[token];
        if (cnt < 65535) counts[token] = cnt + 1;
#ifdef T3_SAMPLER_CLK
        if (a.dbg) for (int i = 0; i < 10; ++i) a.dbg[(size_t)slot * V + i] = (float)(clk[i] - clk[0]);
#endif
    }
}
hipError_t prepare_kernels() {
    { const char* ev = getenv("T3_GEMM_SMALL_M"); g_gemm_small_m = ev ? atoi(ev) : 1; }
    static bool done[MAX_DEVICES] = {};
    if (done[cur_device()]) return hipSuccess;
    const size_t lds = (size_t)(SLOTS + SSCR + 256) * sizeof(unsigned long long);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sampler_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = prepare_gemm2();
    if (e == hipSuccess) done[cur_device()] = true;
    return e;
}
hipError_t launch_sampler(const SampleArgs& a, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    const size_t lds = (size_t)(SLOTS + SSCR + 256) * sizeof(unsigned long long);
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
