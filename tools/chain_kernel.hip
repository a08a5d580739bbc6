// DIAGNOSTIC (built by tools/chain_proto.hip only; not part of libt3engine.so -- measured slower than the launches it would replace,
// DESIGN.md section 8).  Persistent layer-chain kernel of the T3 decode engine for gfx950 (MI355X): the four dependent projections between two
// attention calls -- o_proj(+residual) -> gate/up(RMSNorm folded, SiLU*mul) -> down_proj(+residual) -> the NEXT layer's qkv
// (RMSNorm folded) -- in ONE launch of 256 workgroups (one per CU) instead of four launches.
//
// Why: at decode sizes (2..64 rows against 2..17 MB of weights per projection) every launch of the chain sits at its latency
// floor (grid ramp, kernel-argument fetch, one cold HBM round trip for the first weight tile, drain, ~1.5 us boundary).  Inside
// one launch the three all-to-all seams become grid barriers, and -- the point of the exercise -- every workgroup requests the
// weight tiles of its NEXT phase before it waits at the barrier, so the cold HBM fetch hides behind the synchronisation.
//
// Numerics: exactly the contract of gemm_kernel (DESIGN.md "GEMM" / "RMSNorm"): per output, one MFMA chain per K segment from
// +0 in ascending k, segments folded ((s0+s1)+s2)+s3 per group of four, groups folded left to right.  Wave w of a workgroup owns
// the K range [w K/4, (w+1) K/4): for the 4-segment forms that is segment w; for the 16-segment forms (o: 64-wide, down:
// 256-wide segments) it is group w, folded in registers.  The cross-wave fold goes through LDS in wave order.
//
// Inter-workgroup protocol (cdna_hip_programming.md Guideline 16, form R1 with the `sc1` loads of its "Valid forms" table):
//   * every byte that crosses workgroups inside the launch (h, act) is stored with 16-byte `sc1` (write-through) stores and
//     loaded with 16-byte `sc1` buffer loads, nothing else touches those buffers in this kernel;
//   * arrive: every storing wave `s_waitcnt vmcnt(0)` -> workgroup barrier -> ONE lane of the workgroup's sync wave stores the
//     workgroup's flag (relaxed agent-scope store = `sc1`);
//   * wait: the sync wave polls the 256 flags (one 1 KiB `sc1` load per poll, 4 flags per lane) until every flag has reached
//     the barrier's epoch, then joins the workgroup barrier the compute waves wait at;
//   * epochs count within the flags themselves: a workgroup reads its OWN flag at entry (only it writes it) and its k-th
//     barrier of the launch publishes entry value + k.  All workgroups run the same number of barriers per launch, so the flags
//     stay equal across launches, nothing is reset, and a captured graph replays correctly.  Comparison is wrap-safe.
//   * every spin is bounded: on a timeout the kernel sets *err and runs to its end (results are then garbage and the engine
//     reports the step as failed); it never hangs the GPU.
// Placement-independent: nothing depends on which XCD a workgroup lands on; `blockIdx % 8` only groups the workgroups that
// share a weight tile so that they share an L2 (speed).
#include "../chatterbox-vllm2_amd/csrc/t3_device.h"
#include "../chatterbox-vllm2_amd/csrc/t3_kernels.h"

namespace t3 {

constexpr int CHAIN_WGS = 256;                    // one workgroup per CU
struct ChainArgs {
    const uint4 *Wo, *Wgu, *Wd, *Wqkv;            // packed weights of this layer (gate/up and qkv with their norm weights folded in); Wqkv = the NEXT layer's
    const uint16_t* att;                          // [M][1024] attention output (previous launch)
    uint16_t* h;                                  // [M][1024] residual stream, updated in place
    uint16_t* act;                                // [M][4096] scratch
    uint16_t* qkv;                                // [M][3072] out
    int M;                                        // rows, <= 64
    int phases;                                   // bit 0 o, bit 1 gate/up, bit 2 down, bit 3 qkv (executed in this order)
    unsigned* flags;                              // [CHAIN_WGS] barrier epochs, zeroed once at allocation, never reset
    unsigned* err;                                // set to 1 when a barrier wait gave up
};

typedef unsigned int cu32x4 __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;                      // cache-policy bits of the raw buffer builtins on gfx94x/gfx950: sc0 = 1, nt = 2, sc1 = 16

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint4 ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const cu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX_SC1);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, const uint4& v) {
    __builtin_amdgcn_raw_buffer_store_b128((cu32x4){v.x, v.y, v.z, v.w}, r, byte_off, 0, AUX_SC1);
}

// One (m-tile group, n-tile group) item of one projection, computed by the four compute waves of a workgroup.
//   MT m-tiles x NT packed n-tiles; KBS = k-blocks (of 32) per wave; SEGKB = k-blocks per contract segment (KBS for the
//   4-segment forms, KBS/4 for the 16-segment forms); PDK = k-blocks in the operand ring.
template <int MT, int NT, int KBS, int SEGKB, int PDK, int EPI, bool NORM>
struct ChainItem {
    static constexpr int TILES = MT * NT;
    static constexpr int NTO = (EPI == EPI_SILU) ? NT / 2 : NT;         // output tiles per m-tile
    static constexpr int PIECES = MT * NTO * 32;                        // 16-byte output pieces (one row x 8 columns)
    static constexpr int PITER = (PIECES + 255) / 256;
    static constexpr size_t RED_FLOATS = (size_t)4 * TILES * 256 + (NORM ? 4 * MT * 16 : 0);
    static_assert(KBS % PDK == 0 && KBS % SEGKB == 0 && (PDK % SEGKB == 0 || SEGKB == KBS), "ring / segment shapes");

    // weight tiles of the first PDK k-blocks of this wave's K range: issued BEFORE the grid barrier that precedes the phase
    __device__ static __forceinline__ void prefetch(uint4 (&wr)[16], const uint4* wbase, int KB) {
#pragma unroll
        for (int j = 0; j < PDK; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) wr[j * NT + t] = ld_nt(wbase + (size_t)t * KB * 64 + j * 64);
    }

    // wbase = packed weights of the item's first n-tile at this wave's first k-block, + lane.  X: A operand [rows][32 KB] bf16.
    // out: [rows][ldo] bf16; ot0 = index of the item's first output tile (column ot0 * 16).  red: LDS scratch (RED_FLOATS).
    __device__ static __forceinline__ void run(uint4 (&wr)[16], const uint4* wbase, int KB, __amdgpu_buffer_rsrc_t xr_, int M, int mt0,
                                               __amdgpu_buffer_rsrc_t or_, int ldo, int ot0, float* red) {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int c = lane & 15, q = lane >> 4;
        const int K = KB * 32, k0 = wave * KBS * 32;
        unsigned xoff[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int m = (mt0 + i) * 16 + c;
            m = m < M ? m : M - 1;                  // padded rows re-read the last row; their outputs are dropped
            xoff[i] = (unsigned)(((size_t)m * K + k0 + q * 8) * 2);
        }
        uint4 xr[PDK][MT];
#pragma unroll
        for (int j = 0; j < PDK; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i) xr[j][i] = ld_sc1(xr_, xoff[i] + j * 64);
        }
        // residual operand of this thread's output pieces: requested now, consumed in the epilogue
        uint4 hres[PITER];
        if constexpr (EPI == EPI_RESID) {
#pragma unroll
            for (int k = 0; k < PITER; ++k) {
                const int p = tid + k * 256;
                const int it = p >> 5, r16 = (p >> 1) & 15, half = p & 1;
                const int m = (mt0 + it / NTO) * 16 + r16, n = (ot0 + it % NTO) * 16 + 8 * half;
                hres[k] = (p < PIECES && m < M) ? ld_sc1(or_, (unsigned)(((size_t)m * ldo + n) * 2)) : make_uint4(0, 0, 0, 0);
            }
        }
        f32x4 acc[MT][NT], P[MT][NT], ssq[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            ssq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f}; P[i][t] = acc[i][t]; }
        }
        for (int kbase = 0; kbase < KBS; kbase += PDK) {
#pragma unroll
            for (int j = 0; j < PDK; ++j) {
                const int kb = kbase + j;
                if constexpr (NORM) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) ssq[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xr[j][i]), as_frag(xr[j][i]), ssq[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xr[j][i]), as_frag(wr[j * NT + t]), acc[i][t], 0, 0, 0);
                if (SEGKB < KBS && (j + 1) % SEGKB == 0) {           // a contract segment ends inside this wave's K range (kbase % SEGKB == 0): fold it
                    const bool first = (kb + 1) == SEGKB;
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) { P[i][t][r] = first ? acc[i][t][r] : P[i][t][r] + acc[i][t][r]; acc[i][t][r] = 0.0f; }
                }
                if (kb + PDK < KBS) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) wr[j * NT + t] = ld_nt(wbase + (size_t)t * KB * 64 + (kb + PDK) * 64);
#pragma unroll
                    for (int i = 0; i < MT; ++i) xr[j][i] = ld_sc1(xr_, xoff[i] + (kb + PDK) * 64);
                }
            }
        }
        // cross-wave fold through LDS, in wave order
        float* rowsum = red + (size_t)4 * TILES * 256;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((wave * TILES + i * NT + t) * 4 + r) * 64 + lane] = (SEGKB < KBS) ? P[i][t][r] : acc[i][t][r];
        if constexpr (NORM) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int r = c & 3;
                const float d = r == 0 ? ssq[i][0] : r == 1 ? ssq[i][1] : r == 2 ? ssq[i][2] : ssq[i][3];
                if ((c >> 2) == q) rowsum[wave * (MT * 16) + i * 16 + c] = d;      // diagonal of the A fragment times itself
            }
        }
        __syncthreads();                              // B1 (the sync wave joins it)
#pragma unroll
        for (int k = 0; k < PITER; ++k) {
            const int p = tid + k * 256;
            if (p >= PIECES) continue;
            const int ito = p >> 5, r16 = (p >> 1) & 15, half = p & 1;
            const int i = ito / NTO, to = ito % NTO;
            const int m = (mt0 + i) * 16 + r16;
            if (m >= M) continue;
            const int g = r16 >> 2, r = r16 & 3;
            float v[EPI == EPI_SILU ? 2 : 1][8];
#pragma unroll
            for (int u = 0; u < (EPI == EPI_SILU ? 2 : 1); ++u) {
                const int it = i * NT + (EPI == EPI_SILU ? 2 * to + u : to);
                float4 a0[4], a1[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float* src = red + ((w * TILES + it) * 4 + r) * 64 + 16 * g + 8 * half;
                    a0[w] = *reinterpret_cast<const float4*>(src); a1[w] = *reinterpret_cast<const float4*>(src + 4);
                }
                const float s0[8] = {a0[0].x, a0[0].y, a0[0].z, a0[0].w, a1[0].x, a1[0].y, a1[0].z, a1[0].w};
                const float s1[8] = {a0[1].x, a0[1].y, a0[1].z, a0[1].w, a1[1].x, a1[1].y, a1[1].z, a1[1].w};
                const float s2[8] = {a0[2].x, a0[2].y, a0[2].z, a0[2].w, a1[2].x, a1[2].y, a1[2].z, a1[2].w};
                const float s3[8] = {a0[3].x, a0[3].y, a0[3].z, a0[3].w, a1[3].x, a1[3].y, a1[3].z, a1[3].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][e] = ((s0[e] + s1[e]) + s2[e]) + s3[e];
            }
            if constexpr (NORM) {
                const int rl = i * 16 + r16;
                const float ss = ((rowsum[rl] + rowsum[MT * 16 + rl]) + rowsum[2 * MT * 16 + rl]) + rowsum[3 * MT * 16 + rl];
                const float rstd = 1.0f / sqrtf(ss * (1.0f / 1024.0f) + 1e-5f);
#pragma unroll
                for (int u = 0; u < (EPI == EPI_SILU ? 2 : 1); ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[u][e] = v[u][e] * rstd;
            }
            uint32_t ob[8];
            if constexpr (EPI == EPI_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ob[e] = silu_mul_bf(f2bf(v[0][e]), f2bf(v[1][e]));
            } else if constexpr (EPI == EPI_RESID) {
                float hf[8]; unpack8(hres[k], hf);
#pragma unroll
                for (int e = 0; e < 8; ++e) ob[e] = f2bf(hf[e] + rbf(v[0][e]));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) ob[e] = f2bf(v[0][e]);
            }
            const uint4 o = make_uint4(ob[0] | (ob[1] << 16), ob[2] | (ob[3] << 16), ob[4] | (ob[5] << 16), ob[6] | (ob[7] << 16));
            const int n = (ot0 + to) * 16 + 8 * half;
            st_sc1(or_, (unsigned)(((size_t)m * ldo + n) * 2), o);
        }
    }
};

// grid barrier, executed by the sync wave only.  flags: CHAIN_WGS words; target = the epoch this barrier publishes.
__device__ __forceinline__ void chain_grid_barrier(unsigned* flags, unsigned target, unsigned* err) {
    const int lane = threadIdx.x & 63;
    if (lane == 0) __hip_atomic_store(flags + blockIdx.x, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const __amdgpu_buffer_rsrc_t fr = make_rsrc(flags, CHAIN_WGS * 4);
    for (unsigned spins = 0;; ++spins) {
        const cu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(fr, lane * 16, 0, AUX_SC1);
        const bool ok = (int)(v.x - target) >= 0 && (int)(v.y - target) >= 0 && (int)(v.z - target) >= 0 && (int)(v.w - target) >= 0;
        if (__all(ok)) break;
        // Bounded: ~0.3 s without every workgroup arriving (one of them not resident?) -> give up, loudly, and let every later
        // wait of this and the following launches fall through at once (the engine reports the step as failed and stops).
        if ((spins & 255u) == 255u) {
            const unsigned e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (e != 0u) break;
            if (spins >= (1u << 18)) { if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// Work map (grid = 256 workgroups, b = blockIdx.x, x = b % 8 labels the workgroups that share an L2, y = b / 8):
//   o / down : item (n-tile, m-tile): n-tile = x + 8 (y / MTT), m-tile = y % MTT             (64 n-tiles, active while y < 8 MTT)
//   gate/up  : MTT > 2: item (2 pairs, 2 m-tiles): pair group = x + 8 (y / 2), m-group = y % 2; else item (1 pair, MTT m-tiles): pair = b
//   qkv      : MTT > 2: item (3 n-tiles, 1 m-tile): group = x + 8 (y / MTT), m-tile = y % MTT (active while y < 8 MTT);
//              MTT = 2: item (2 n-tiles, 1 m-tile): group = x + 8 (y / 2) < 96, m-tile = y % 2;  MTT = 1: item (1 n-tile): b < 192
template <int MTT>
__global__ __launch_bounds__(320) void chain_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, x = b & 7, y = b >> 3;
    constexpr int KB1 = D / 32, KB4 = F / 32;

    const bool do_o = a.phases & 1, do_gu = a.phases & 2, do_dn = a.phases & 4, do_q = a.phases & 8;
    // items
    const bool it_od = y < 8 * MTT;
    const int od_nt = x + 8 * (y / MTT), od_mt = y % MTT;
    constexpr int GU_MT = MTT > 2 ? 2 : MTT, GU_PAIRS = MTT > 2 ? 2 : 1;
    const int gu_pg = MTT > 2 ? x + 8 * (y >> 1) : b, gu_mt0 = MTT > 2 ? 2 * (y & 1) : 0;
    constexpr int Q_NT = MTT > 2 ? 3 : MTT;
    const bool it_q = MTT > 2 ? (y < 8 * MTT) : (MTT == 2 ? (x + 8 * (y >> 1)) < 96 : b < 192);
    const int q_g = MTT > 2 ? x + 8 * (y / MTT) : (MTT == 2 ? x + 8 * (y >> 1) : b), q_mt = MTT > 2 ? y % MTT : (MTT == 2 ? (y & 1) : 0);

    typedef ChainItem<1, 1, 8, 2, 8, EPI_RESID, false> ItemO;
    typedef ChainItem<GU_MT, 2 * GU_PAIRS, 8, 8, 4, EPI_SILU, true> ItemGU;
    typedef ChainItem<1, 1, 32, 8, 8, EPI_RESID, false> ItemD;
    typedef ChainItem<1, Q_NT, 8, 8, 4, EPI_BF16, true> ItemQ;

    if (wave == 4) {
        // ---- sync wave: mirrors the compute waves' workgroup barriers and runs the grid barriers
        unsigned e0 = 0;
        if (lane == 0) e0 = __hip_atomic_load(a.flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        e0 = __builtin_amdgcn_readfirstlane(e0);
        unsigned k = 0;
        const bool ph[4] = {do_o, do_gu, do_dn, do_q};
        const bool has[4] = {it_od, true, it_od, it_q};
        int last = -1;
#pragma unroll
        for (int p = 0; p < 4; ++p) if (ph[p]) last = p;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!ph[p]) continue;
            if (has[p]) __syncthreads();              // B1 of the item
            __syncthreads();                          // B2: the compute waves' stores have drained
            if (p != last) chain_grid_barrier(a.flags, e0 + (++k), a.err);
            __syncthreads();                          // B3: inputs of the next phase are visible
        }
        return;
    }

    // ---- compute waves
    const __amdgpu_buffer_rsrc_t r_att = make_rsrc(a.att, (unsigned)((size_t)a.M * D * 2));
    const __amdgpu_buffer_rsrc_t r_h = make_rsrc(a.h, (unsigned)((size_t)a.M * D * 2));
    const __amdgpu_buffer_rsrc_t r_act = make_rsrc(a.act, (unsigned)((size_t)a.M * F * 2));
    const __amdgpu_buffer_rsrc_t r_qkv = make_rsrc(a.qkv, (unsigned)((size_t)a.M * QKV * 2));
    const uint4* w_o = a.Wo + ((size_t)od_nt * KB1 + wave * 8) * 64 + lane;
    const uint4* w_gu = a.Wgu + ((size_t)(gu_pg * 2 * GU_PAIRS) * KB1 + wave * 8) * 64 + lane;
    const uint4* w_d = a.Wd + ((size_t)od_nt * KB4 + wave * 32) * 64 + lane;
    const uint4* w_q = a.Wqkv + ((size_t)(q_g * Q_NT) * KB1 + wave * 8) * 64 + lane;
    uint4 wr[16];
    auto drain_and_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its write-through stores have left
        __syncthreads();                                          // B2
    };
    // first phase: its inputs come from the previous launch, nothing to wait for
    if (do_o) { if (it_od) ItemO::prefetch(wr, w_o, KB1); }
    else if (do_gu) ItemGU::prefetch(wr, w_gu, KB1);
    else if (do_dn) { if (it_od) ItemD::prefetch(wr, w_d, KB4); }
    else if (do_q) { if (it_q) ItemQ::prefetch(wr, w_q, KB1); }

    if (do_o) {
        if (it_od) ItemO::run(wr, w_o, KB1, r_att, a.M, od_mt, r_h, D, od_nt, red);
        drain_and_sync();
        if (do_gu) ItemGU::prefetch(wr, w_gu, KB1);
        else if (do_dn) { if (it_od) ItemD::prefetch(wr, w_d, KB4); }
        else if (do_q) { if (it_q) ItemQ::prefetch(wr, w_q, KB1); }
        __syncthreads();                                          // B3
    }
    if (do_gu) {
        ItemGU::run(wr, w_gu, KB1, r_h, a.M, gu_mt0, r_act, F, gu_pg * GU_PAIRS, red);
        drain_and_sync();
        if (do_dn) { if (it_od) ItemD::prefetch(wr, w_d, KB4); }
        else if (do_q) { if (it_q) ItemQ::prefetch(wr, w_q, KB1); }
        __syncthreads();
    }
    if (do_dn) {
        if (it_od) ItemD::run(wr, w_d, KB4, r_act, a.M, od_mt, r_h, D, od_nt, red);
        drain_and_sync();
        if (do_q) { if (it_q) ItemQ::prefetch(wr, w_q, KB1); }
        __syncthreads();
    }
    if (do_q) {
        if (it_q) ItemQ::run(wr, w_q, KB1, r_h, a.M, q_mt, r_qkv, QKV, q_g * Q_NT, red);
        drain_and_sync();
        __syncthreads();
    }
}

size_t chain_lds_bytes(int mtt) {
    // the largest item: gate/up (GU_MT x 2 GU_PAIRS tiles, NORM)
    const int gu_mt = mtt > 2 ? 2 : mtt, gu_nt = mtt > 2 ? 4 : 2;
    const size_t gu = ((size_t)4 * gu_mt * gu_nt * 256 + 4 * gu_mt * 16) * sizeof(float);
    const size_t q = ((size_t)4 * 3 * 256 + 4 * 16) * sizeof(float);
    return gu > q ? gu : q;
}

bool chain_supported() {
    // the grid barrier needs all 256 workgroups resident at once: one per CU
    static int ok = -1;
    if (ok < 0) {
        int dev = 0; hipDeviceProp_t p;
        ok = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount >= CHAIN_WGS) ? 1 : 0;
    }
    return ok == 1;
}

hipError_t launch_chain(const ChainArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.M > 64 || !a.phases) return hipErrorInvalidValue;
    if (!chain_supported()) return hipErrorNotSupported;
    const int mtt = (a.M + 15) / 16;
    const size_t lds = chain_lds_bytes(mtt);
    switch (mtt) {
        case 1: hipLaunchKernelGGL(chain_kernel<1>, dim3(CHAIN_WGS), dim3(320), lds, s, a); break;
        case 2: hipLaunchKernelGGL(chain_kernel<2>, dim3(CHAIN_WGS), dim3(320), lds, s, a); break;
        case 3: hipLaunchKernelGGL(chain_kernel<3>, dim3(CHAIN_WGS), dim3(320), lds, s, a); break;
        default: hipLaunchKernelGGL(chain_kernel<4>, dim3(CHAIN_WGS), dim3(320), lds, s, a); break;
    }
    return hipGetLastError();
}

}  // namespace t3
