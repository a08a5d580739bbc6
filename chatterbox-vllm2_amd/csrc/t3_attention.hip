// RoPE + paged KV write and the attention kernels of the T3 decode engine for gfx950 (MI355X, CDNA4).  wave = 64 lanes.
//
// Every floating-point rounding point and summation order in this file is part of the numerics contract written down in
// DESIGN.md ("Attention", "RoPE").  Compile with -ffp-contract=off.
// Reference semantics: the attention of the Llama block, src/chatterbox_vllm/models/t3/t3.py:696-713 -> vllm LlamaModel
// (paged attention of the vLLM engine, hyper-parameters t3-model/config.json:1-33).
#include "t3_kernels.h"
#include "t3_device.h"

#include <math.h>
#include <string.h>

namespace t3 {
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
// operand's own order, rounds 1-3; tools/diag/t3_attention_diag.hip -DT3_K_TOKEN_MAJOR=0) they are 256 bytes apart, 8 lines per (row, head), and the write's cost follows the
// lines touched: C3 21.12 -> 21.27 k tok/s (profiles/r03_k_token_major_*.json).  A wave still loads the same 1 KiB per fragment, each lane
// from its permuted place.  Every K reader and writer goes through k_piece() / k_lane_piece().
__device__ __forceinline__ int k_piece(int t, int kg) { return 4 * t + kg; }
__device__ __forceinline__ int k_lane_piece(int lane) { return 4 * (lane & 15) + (lane >> 4); }      // the piece lane (t = lane % 16, kg = lane / 16) loads
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

// (Diagnostic variants of this kernel -- phase stamps, loads-only, no / other K-V write-back forms, dynamic chunk hand-out, full last
// tiles, ds_bpermute reductions -- live in tools/diag/t3_attention_diag.hip, built only by tools/Makefile; what they measured: profiles/NOTES.md.)
template <int NW, bool NT, bool FUSE>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 4 : 2) void attention_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float part[];   // [max_chunks] m | [max_chunks] l | [max_chunks][64] o | per wave: 64 scores, 64 bf16 p | FUSE, per wave: [12][64] newest k / v
    float* pm = part; float* pl = part + a.max_chunks; float* po = part + 2 * a.max_chunks;
    // EARLY (the 8-wave form = 1-4 utterances): the launch is a chain of dependent loads (kernel arguments -> row record -> block id -> tile
    // -> arithmetic, 3.6 of its 5.6 us at B = 1), so (a) the wave index is made uniform for the compiler: block ids come by SCALAR loads, (b) the
    // block of the wave's first chunk is asked for together with the context length, (c) the pre-RoPE q / k pieces, which need nothing from the
    // record, are asked for before it has arrived, and the RoPE table rows before the tile.  B = 1: 1 318 -> 1 341 tok/s on one box.  At 64
    // rows (4-wave form) the same order is 0.6 % SLOWER (the launch streams at the memory system's pace from its first microsecond; 21.18 ->
    // 21.06 k tok/s, three alternating runs): the 4-wave form keeps the order of round 2.
    constexpr bool EARLY = FUSE && NW == 8;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sbuf = part + 66 * a.max_chunks + wave * 96;            // 64 floats of scores, then 64 bf16 (32 floats) of probabilities
    uint32_t* stash = reinterpret_cast<uint32_t*>(part + 66 * a.max_chunks + NW * 96) + wave * (12 * 64) + (threadIdx.x & 63);   // FUSE: the newest key / value park here
    uint16_t* pbuf = reinterpret_cast<uint16_t*>(sbuf + 64);
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
    auto load_tiles = [&](int c) {
        const int blk = (EARLY && c == wave) ? blk_first : bt[c / CPB], ci = c % CPB;
        const uint4* Kp = reinterpret_cast<const uint4*>(a.kv_layer + kv_head_base(blk, 0, h) + (size_t)ci * (CHUNK * HD)) + k_lane_piece(lane);
        const uint4* Vp = reinterpret_cast<const uint4*>(a.kv_layer + kv_head_base(blk, 1, h) + (size_t)ci * (CHUNK * HD)) + lane;
        const int npool = L - (FUSE ? 1 : 0) - c * CHUNK;       // tokens of this chunk that live in the pool (wave-uniform; >= 64 except in the last chunk)
        if (npool >= CHUNK) {
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
        if (wave == 0) {                            // paged write of the newest K (8 pieces of 16 bytes, early: their latency hides under the tile stream)
            const int blk = bt[pos / KV_BLOCK], tok = pos % KV_BLOCK;
            uint16_t* kb = a.kv_layer_w + kv_head_base(blk, 0, h);
            if (col == 0) {
                *reinterpret_cast<uint4*>(kb + k_slot(tok, 0, kg)) = knf[0];
                *reinterpret_cast<uint4*>(kb + k_slot(tok, 1, kg)) = knf[1];
            }                                      // V goes back as whole 16-byte pieces from the wave that holds the last tile (chunk loop)
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
    for (int c = wave; c < nc; c += NW) {
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
            if (kg == kgs) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    if (tss == 0) patch16(vf[2 * dt], js, vnew[dt]); else patch16(vf[2 * dt + 1], js, vnew[dt]);
                }
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
            }
        }
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
        const float m = wave_max_f32(sc, lane);
        const float p = live ? t3_expf(sc - m) : 0.0f;
        const float lsum = wave_bfly_add_f32(p, lane);
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
    }
    __syncthreads();
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

}  // namespace t3
