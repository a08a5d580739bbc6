// Kernel-level entry points of include/t3_engine.h (t3k_*): host buffers in, host buffers out.
// They run exactly the kernels the engine runs, so the parity tests can check each kernel against the
// oracle's function of the same name through the C ABI.
#include <cstring>
#include <string>
#include <vector>

#include "t3_kernels.h"

using namespace t3;

namespace {
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes, bool zero = false) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e == hipSuccess && zero) e = hipMemset(p, 0, bytes ? bytes : 16);
        return e;
    }
    hipError_t from(const void* h, size_t bytes) {
        hipError_t e = alloc(bytes);
        if (e == hipSuccess && bytes) e = hipMemcpy(p, h, bytes, hipMemcpyHostToDevice);
        return e;
    }
    template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};
bool have_device() { int n = 0; return hipGetDeviceCount(&n) == hipSuccess && n > 0; }
}  // namespace

#define K_TRY(expr) do { if ((expr) != hipSuccess) return T3_E_DEVICE; } while (0)

extern "C" int t3k_set_prefill_rows(int32_t rows) { set_pgemm_min_rows(rows); return T3_OK; }
extern "C" int t3k_set_prefill_wide_rows(int32_t rows) { set_pgemm_wide_rows(rows); return T3_OK; }

extern "C" int t3k_gemm(const void* x, const void* w, int32_t M, int32_t K, int32_t N, float* out, int32_t mt, int32_t nw) {
    if (!x || !w || !out || M <= 0 || N <= 0 || (nw != 4 && nw != 16) || !(nw == 4 ? K == D : ((K == D || K == F) && N % 16 == 0))) return T3_E_INVALID;
    if (!have_device()) return T3_E_DEVICE;
    gemm_refresh_switches();
    const int Npad = (N + 15) / 16 * 16;
    std::vector<uint16_t> packed((size_t)Npad * K);
    pack_weight((const uint16_t*)w, N, K, Npad, packed.data());
    DevBuf dx, dw, dout;
    K_TRY(dx.from(x, (size_t)M * K * 2)); K_TRY(dw.from(packed.data(), packed.size() * 2)); K_TRY(dout.alloc((size_t)M * N * 4, true));
    GemmArgs a{dx.as<uint16_t>(), dw.as<uint4>(), M, K, N, dout.p, N, nw, 0, nullptr};
    K_TRY(launch_gemm(a, EPI_F32, mt > 0 ? mt : choose_mt(M, Npad / 16, nw, false), nullptr));
    K_TRY(hipDeviceSynchronize());
    K_TRY(hipMemcpy(out, dout.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    return T3_OK;
}

/* RMSNorm folded into the GEMM: out[r][n] = rstd * GEMM(bf16(h[row_index[r]] * ln_w), W); h [Mh][1024] */
extern "C" int t3k_norm_gemm(const void* h, const void* ln_w, const void* w, int32_t M, int32_t N, float* out,
                             const int32_t* row_index, int32_t Mh) {
    if (!h || !ln_w || !w || !out || M <= 0 || N <= 0 || Mh <= 0) return T3_E_INVALID;
    if (!have_device()) return T3_E_DEVICE;
    gemm_refresh_switches();
    const int Npad = (N + 15) / 16 * 16;
    std::vector<uint16_t> folded((size_t)N * D), packed((size_t)Npad * D);
    fold_norm_weight((const uint16_t*)w, N, D, (const uint16_t*)ln_w, folded.data());      // what the engine does once at load time
    pack_weight(folded.data(), N, D, Npad, packed.data());
    DevBuf dh, dw, dout, dri;
    K_TRY(dh.from(h, (size_t)Mh * D * 2)); K_TRY(dw.from(packed.data(), packed.size() * 2));
    K_TRY(dout.alloc((size_t)M * N * 4, true));
    if (row_index) K_TRY(dri.from(row_index, (size_t)M * 4));
    DevBuf drs; K_TRY(drs.alloc((size_t)M * 4));
    GemmArgs a{dh.as<uint16_t>(), dw.as<uint4>(), M, D, N, dout.p, N, 4, 1, row_index ? dri.as<int>() : nullptr, 0, drs.as<float>()};
    K_TRY(launch_gemm(a, EPI_F32, choose_mt(M, Npad / 16, 4, true), nullptr));
    K_TRY(hipDeviceSynchronize());
    K_TRY(hipMemcpy(out, dout.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    return T3_OK;
}

/* the qkv projection exactly as a step launches it: input RMSNorm folded, bf16 out [M][3072] */
extern "C" int t3k_qkv_gemm(const void* h, const void* ln_w, const void* w, int32_t M, void* out_bf16) {
    if (!h || !ln_w || !w || !out_bf16 || M <= 0) return T3_E_INVALID;
    if (!have_device()) return T3_E_DEVICE;
    gemm_refresh_switches();
    std::vector<uint16_t> folded((size_t)QKV * D), packed((size_t)QKV * D);
    fold_norm_weight((const uint16_t*)w, QKV, D, (const uint16_t*)ln_w, folded.data());
    pack_weight(folded.data(), QKV, D, QKV, packed.data());
    DevBuf dh, dw, dout, drs;
    K_TRY(dh.from(h, (size_t)M * D * 2)); K_TRY(dw.from(packed.data(), packed.size() * 2)); K_TRY(dout.alloc((size_t)M * QKV * 2, true)); K_TRY(drs.alloc((size_t)M * 4));
    GemmArgs a{dh.as<uint16_t>(), dw.as<uint4>(), M, D, QKV, dout.p, QKV, 4, 1, nullptr, 0, drs.as<float>()};
    K_TRY(launch_gemm(a, EPI_BF16, choose_mt(M, QKV / 16, 4, true), nullptr));
    K_TRY(hipDeviceSynchronize());
    K_TRY(hipMemcpy(out_bf16, dout.p, (size_t)M * QKV * 2, hipMemcpyDeviceToHost));
    return T3_OK;
}

/* the speech head exactly as a step launches it: final RMSNorm folded, the sampled rows gathered through row_index, bf16 logits with
 * leading dimension 8208, the packed matrix padded to 516 n-tiles (tile groups may overhang the 8 194 columns) */
extern "C" int t3k_head_gemm(const void* h, const void* ln_w, const void* w, int32_t M, const int32_t* row_index, int32_t Mh, void* out_bf16) {
    if (!h || !ln_w || !w || !out_bf16 || !row_index || M <= 0 || Mh <= 0) return T3_E_INVALID;
    if (!have_device()) return T3_E_DEVICE;
    gemm_refresh_switches();
    std::vector<uint16_t> folded((size_t)V * D), packed((size_t)HEAD_TILES * 16 * D);
    fold_norm_weight((const uint16_t*)w, V, D, (const uint16_t*)ln_w, folded.data());
    pack_weight(folded.data(), V, D, HEAD_TILES * 16, packed.data());
    DevBuf dh, dw, dout, dri, drs;
    K_TRY(dh.from(h, (size_t)Mh * D * 2)); K_TRY(dw.from(packed.data(), packed.size() * 2)); K_TRY(dout.alloc((size_t)M * VPAD * 2, true));
    K_TRY(dri.from(row_index, (size_t)M * 4)); K_TRY(drs.alloc((size_t)M * 4));
    // as the engine does: a decode-only step samples every row in row order, the gather is the identity and the head reads its rows directly
    bool identity = M == Mh;
    for (int i = 0; identity && i < M; ++i) identity = row_index[i] == i;
    GemmArgs a{dh.as<uint16_t>(), dw.as<uint4>(), M, D, V, dout.p, VPAD, 4, 1, identity ? nullptr : dri.as<int>(), HEAD_TILES, drs.as<float>()};
    K_TRY(launch_gemm(a, EPI_BF16, choose_mt(M, VPAD / 16, 4, true), nullptr));
    K_TRY(hipDeviceSynchronize());
    K_TRY(hipMemcpy(out_bf16, dout.p, (size_t)M * VPAD * 2, hipMemcpyDeviceToHost));
    return T3_OK;
}

/* residual epilogue of the o_proj / down_proj form: h [M][N] bf16 updated in place: h = bf16(h + bf16(x W^T)) */
extern "C" int t3k_gemm_resid(const void* x, const void* w, int32_t M, int32_t K, int32_t N, void* h_bf16) {
    if (!x || !w || !h_bf16 || M <= 0 || N <= 0 || N % 16 || K % 512) return T3_E_INVALID;
    if (!have_device()) return T3_E_DEVICE;
    gemm_refresh_switches();
    std::vector<uint16_t> packed((size_t)N * K);
    pack_weight((const uint16_t*)w, N, K, N, packed.data());
    DevBuf dx, dw, dh;
    K_TRY(dx.from(x, (size_t)M * K * 2)); K_TRY(dw.from(packed.data(), packed.size() * 2)); K_TRY(dh.from(h_bf16, (size_t)M * N * 2));
    GemmArgs a{dx.as<uint16_t>(), dw.as<uint4>(), M, K, N, dh.p, N, 16, 0, nullptr};
    K_TRY(launch_gemm(a, EPI_RESID, choose_mt(M, N / 16, 16, false), nullptr));
    K_TRY(hipDeviceSynchronize());
    K_TRY(hipMemcpy(h_bf16, dh.p, (size_t)M * N * 2, hipMemcpyDeviceToHost));
    return T3_OK;
}

/* gate/up form: out = bf16( bf16(silu(g)) * u ), g/u = bf16(rstd * GEMM(bf16(h*ln_w), Wg/Wu)) */
extern "C" int t3k_silu_mul_gemm(const void* h, const void* ln_w, const void* wg, const void* wu, int32_t M, int32_t Fd, void* out) {
    if (!h || !ln_w || !wg || !wu || !out || M <= 0 || Fd <= 0 || Fd % 16) return T3_E_INVALID;
    if (!have_device()) return T3_E_DEVICE;
    gemm_refresh_switches();
    std::vector<uint16_t> fg((size_t)Fd * D), fu((size_t)Fd * D), packed((size_t)2 * Fd * D);
    fold_norm_weight((const uint16_t*)wg, Fd, D, (const uint16_t*)ln_w, fg.data());
    fold_norm_weight((const uint16_t*)wu, Fd, D, (const uint16_t*)ln_w, fu.data());
    pack_gate_up(fg.data(), fu.data(), Fd, D, packed.data());
    DevBuf dx, dw, dout;
    K_TRY(dx.from(h, (size_t)M * D * 2)); K_TRY(dw.from(packed.data(), packed.size() * 2)); K_TRY(dout.alloc((size_t)M * Fd * 2, true));
    DevBuf drs; K_TRY(drs.alloc((size_t)M * 4));
    GemmArgs a{dx.as<uint16_t>(), dw.as<uint4>(), M, D, Fd, dout.p, Fd, 4, 1, nullptr, 0, drs.as<float>()};
    K_TRY(launch_gemm(a, EPI_SILU, choose_mt(M, Fd / 16, 4, true), nullptr));
    K_TRY(hipDeviceSynchronize());
    K_TRY(hipMemcpy(out, dout.p, (size_t)M * Fd * 2, hipMemcpyDeviceToHost));
    return T3_OK;
}

extern "C" int t3k_rope_attention(const void* qkv, const int32_t* row_stream, const int32_t* row_pos, int32_t rows,
                                  int32_t n_streams, int32_t max_pos, void* out) {
    if (!qkv || !row_stream || !row_pos || !out || rows <= 0 || n_streams <= 0 || max_pos <= 0) return T3_E_INVALID;
    if (!have_device()) return T3_E_DEVICE;
    for (int r = 0; r < rows; ++r)
        if (row_stream[r] < 0 || row_stream[r] >= n_streams || row_pos[r] < 0 || row_pos[r] >= max_pos) return T3_E_INVALID;
    const int max_blocks = (max_pos + KV_BLOCK - 1) / KV_BLOCK;
    // a deliberately scrambled block table, so that the test exercises the paging
    std::vector<int> table((size_t)n_streams * max_blocks);
    const int nb = n_streams * max_blocks;
    for (int i = 0; i < nb; ++i) table[i] = (int)(((long)i * 7919 + 13) % nb);
    {   // make it a permutation: 7919 is coprime with nb unless nb is a multiple of 7919
        std::vector<char> seen(nb, 0); bool ok = true;
        for (int i = 0; i < nb; ++i) { if (seen[table[i]]) ok = false; seen[table[i]] = 1; }
        if (!ok) for (int i = 0; i < nb; ++i) table[i] = nb - 1 - i;
    }
    std::vector<float> c((size_t)max_pos * 32), s((size_t)max_pos * 32);
    rope_tables(max_pos, c.data(), s.data());
    const int stride = row_stride_words(max_blocks);
    std::vector<int> recs((size_t)rows * stride, 0);
    for (int r = 0; r < rows; ++r) {
        int* rec = recs.data() + (size_t)r * stride;
        rec[0] = row_stream[r]; rec[1] = row_pos[r];
        for (int b = 0; b < max_blocks; ++b) rec[ROW_HDR + b] = table[(size_t)row_stream[r] * max_blocks + b];
    }
    DevBuf dqkv, drec, dc, ds, dq, dkv, dout;
    K_TRY(dqkv.from(qkv, (size_t)rows * QKV * 2)); K_TRY(drec.from(recs.data(), recs.size() * 4));
    K_TRY(dc.from(c.data(), c.size() * 4)); K_TRY(ds.from(s.data(), s.size() * 4));
    K_TRY(dq.alloc((size_t)rows * D * 2)); K_TRY(dkv.alloc((size_t)nb * KV_BLOCK_ELEMS * 2, true)); K_TRY(dout.alloc((size_t)rows * D * 2, true));
    RopeArgs ra{dqkv.as<uint16_t>(), dq.as<uint16_t>(), dkv.as<uint16_t>(), drec.as<int>(), stride, dc.as<float>(), ds.as<float>(), rows};
    K_TRY(launch_rope_kv(ra, nullptr));
    AttnArgs aa{dq.as<uint16_t>(), dkv.as<uint16_t>(), drec.as<int>(), stride, dout.as<uint16_t>(), rows, (max_pos + CHUNK - 1) / CHUNK, nullptr, nullptr, nullptr, nullptr, 0, 0};     // every row through the 16-row tile schedule
    K_TRY(launch_attention(aa, nullptr));
    K_TRY(hipDeviceSynchronize());
    K_TRY(hipMemcpy(out, dout.p, (size_t)rows * D * 2, hipMemcpyDeviceToHost));
    return T3_OK;
}

/* The FUSED decode attention exactly as a decode step launches it (attention_kernel<waves, nt, FUSE = true>: RoPE of q / k, paged
 * write of the newest K / V, attention over the paged context), `steps` consecutive launches of `rows` rows:
 *   - stream r (= row r) holds ctx[r] - 1 context tokens whose pre-RoPE qkv rows are ctx_qkv[content r % n_content][0 .. ctx[r] - 2]
 *     (written into a scrambled paged pool by rope_kv_kernel, the prefill path);
 *   - launch s processes, for every row r, the pre-RoPE row new_qkv[s][r] at position ctx[r] - 1 + s: launch s >= 1 therefore
 *     reads the K / V that launch s - 1 wrote through the fused path.
 * out [steps][rows][1024]; kv_new (nullable) [steps][rows][2][1024]: K (rotated) and V of the positions the launches wrote, read
 * back from the pool. */
extern "C" int t3k_decode_attention(const void* ctx_qkv, int32_t n_content, int32_t content_rows, const void* new_qkv, const int32_t* ctx,
                                    int32_t rows, int32_t steps, int32_t max_pos, int32_t waves, void* out, void* kv_new) {
    if (!ctx_qkv || !new_qkv || !ctx || !out || rows <= 0 || steps <= 0 || n_content <= 0 || content_rows <= 0 || max_pos <= 0 ||
        (waves != 0 && waves != 4 && waves != 8)) return T3_E_INVALID;
    for (int r = 0; r < rows; ++r)
        if (ctx[r] < 1 || ctx[r] - 1 > content_rows || ctx[r] - 1 + steps > max_pos) return T3_E_INVALID;
    if (!have_device()) return T3_E_DEVICE;
    const int max_blocks = (max_pos + KV_BLOCK - 1) / KV_BLOCK, nb = rows * max_blocks;
    std::vector<int> table(nb);
    for (int i = 0; i < nb; ++i) table[i] = (int)(((long)i * 7919 + 13) % nb);
    {
        std::vector<char> seen(nb, 0); bool ok = true;
        for (int i = 0; i < nb; ++i) { if (seen[table[i]]) ok = false; seen[table[i]] = 1; }
        if (!ok) for (int i = 0; i < nb; ++i) table[i] = nb - 1 - i;
    }
    std::vector<float> c((size_t)max_pos * 32), s((size_t)max_pos * 32);
    rope_tables(max_pos, c.data(), s.data());
    const int stride = row_stride_words(max_blocks);
    DevBuf dctx, dnew, dc, ds, dq, dkv, dout, dkvn, drec_fill, drec;
    K_TRY(dctx.from(ctx_qkv, (size_t)n_content * content_rows * QKV * 2)); K_TRY(dnew.from(new_qkv, (size_t)steps * rows * QKV * 2));
    K_TRY(dc.from(c.data(), c.size() * 4)); K_TRY(ds.from(s.data(), s.size() * 4));
    K_TRY(dq.alloc((size_t)content_rows * D * 2));                     // rotated q of the fill rows: not used
    K_TRY(dkv.alloc((size_t)nb * KV_BLOCK_ELEMS * 2, true)); K_TRY(dout.alloc((size_t)steps * rows * D * 2, true));
    K_TRY(dkvn.alloc((size_t)steps * rows * 2 * D * 2, true));
    // ---- context fill: one rope_kv launch per stream over its content's rows (positions 0 .. ctx[r] - 2)
    std::vector<int> recs((size_t)content_rows * stride);
    K_TRY(drec_fill.alloc(recs.size() * 4));
    for (int r = 0; r < rows; ++r) {
        const int n = ctx[r] - 1;
        if (n <= 0) continue;
        for (int p = 0; p < n; ++p) {
            int* rec = recs.data() + (size_t)p * stride;
            memset(rec, 0, (size_t)stride * 4);
            rec[0] = r; rec[1] = p;
            for (int b = 0; b < max_blocks; ++b) rec[ROW_HDR + b] = table[(size_t)r * max_blocks + b];
        }
        K_TRY(hipMemcpy(drec_fill.p, recs.data(), (size_t)n * stride * 4, hipMemcpyHostToDevice));
        RopeArgs ra{dctx.as<uint16_t>() + (size_t)(r % n_content) * content_rows * QKV, dq.as<uint16_t>(), dkv.as<uint16_t>(), drec_fill.as<int>(), stride,
                    dc.as<float>(), ds.as<float>(), n};
        K_TRY(launch_rope_kv(ra, nullptr));
        K_TRY(hipDeviceSynchronize());                                 // the record buffer is reused by the next stream
    }
    // ---- the decode launches
    std::vector<int> drecs((size_t)rows * stride, 0);
    K_TRY(drec.alloc(drecs.size() * 4));
    for (int st = 0; st < steps; ++st) {
        for (int r = 0; r < rows; ++r) {
            int* rec = drecs.data() + (size_t)r * stride;
            rec[0] = r; rec[1] = ctx[r] - 1 + st;
            for (int b = 0; b < max_blocks; ++b) rec[ROW_HDR + b] = table[(size_t)r * max_blocks + b];
        }
        K_TRY(hipMemcpy(drec.p, drecs.data(), drecs.size() * 4, hipMemcpyHostToDevice));
        AttnArgs aa{nullptr, dkv.as<uint16_t>(), drec.as<int>(), stride, dout.as<uint16_t>() + (size_t)st * rows * D, rows, (max_pos + CHUNK - 1) / CHUNK,
                    dnew.as<uint16_t>() + (size_t)st * rows * QKV, dkv.as<uint16_t>(), dc.as<float>(), ds.as<float>()};
        aa.force_waves = waves;
        K_TRY(launch_attention(aa, nullptr));
        K_TRY(launch_kv_gather(dkv.as<uint16_t>(), drec.as<int>(), stride, rows, dkvn.as<uint16_t>() + (size_t)st * rows * 2 * D, nullptr));
        K_TRY(hipDeviceSynchronize());
    }
    K_TRY(hipMemcpy(out, dout.p, (size_t)steps * rows * D * 2, hipMemcpyDeviceToHost));
    if (kv_new) K_TRY(hipMemcpy(kv_new, dkvn.p, (size_t)steps * rows * 2 * D * 2, hipMemcpyDeviceToHost));
    return T3_OK;
}

static int sample_impl(const void* logits2, int32_t ldl, uint16_t* counts, const T3Sampling* sp, float cfg, uint32_t step,
                       int32_t* token_out, float* logits_out, uint8_t* keep_out);
extern "C" int t3k_sample(const void* logits2, int32_t ldl, uint16_t* counts, const T3Sampling* sp, float cfg, uint32_t step,
                          int32_t* token_out, float* logits_out) { return sample_impl(logits2, ldl, counts, sp, cfg, step, token_out, logits_out, nullptr); }
/* t3k_sample + the SUPPORT of the draw: keep_out [8194] = 1 where the masks (penalties -> /T -> min-p -> top-k -> top-p) leave the id drawable */
extern "C" int t3k_sample_support(const void* logits2, int32_t ldl, uint16_t* counts, const T3Sampling* sp, float cfg, uint32_t step,
                                  int32_t* token_out, uint8_t* keep_out) {
    if (!keep_out) return T3_E_INVALID;
    return sample_impl(logits2, ldl, counts, sp, cfg, step, token_out, nullptr, keep_out);
}
static int sample_impl(const void* logits2, int32_t ldl, uint16_t* counts, const T3Sampling* sp, float cfg, uint32_t step,
                       int32_t* token_out, float* logits_out, uint8_t* keep_out) {
    if (!logits2 || !counts || !sp || !token_out || ldl < V) return T3_E_INVALID;
    if (!have_device()) return T3_E_DEVICE;
    std::vector<uint16_t> cpad(VPAD, 0);
    memcpy(cpad.data(), counts, V * 2);
    const int4 sel = make_int4(0, (int)step, 0, 0);
    DevBuf dl, dc, dsp, dsel, dtok, ddbg;
    K_TRY(dl.from(logits2, (size_t)2 * ldl * 2)); K_TRY(dc.from(cpad.data(), VPAD * 2)); K_TRY(dsp.from(sp, sizeof(T3Sampling)));
    K_TRY(dsel.from(&sel, sizeof(sel))); K_TRY(dtok.alloc(4, true)); K_TRY(ddbg.alloc((size_t)V * 4, true));
    SampleArgs sa{dl.as<uint16_t>(), ldl, dsel.as<int4>(), dc.as<uint16_t>(), dsp.as<T3Sampling>(), cfg, dtok.as<int>(), ddbg.as<float>(), 1};
    DevBuf dkeep;
    if (keep_out) { K_TRY(dkeep.alloc(V, true)); sa.dbg_keep = dkeep.as<unsigned char>(); }
    K_TRY(launch_sampler(sa, nullptr));
    K_TRY(hipDeviceSynchronize());
    if (keep_out) K_TRY(hipMemcpy(keep_out, dkeep.p, V, hipMemcpyDeviceToHost));
    K_TRY(hipMemcpy(token_out, dtok.p, 4, hipMemcpyDeviceToHost));
    K_TRY(hipMemcpy(cpad.data(), dc.p, VPAD * 2, hipMemcpyDeviceToHost));
    memcpy(counts, cpad.data(), V * 2);
    if (logits_out) K_TRY(hipMemcpy(logits_out, ddbg.p, (size_t)V * 4, hipMemcpyDeviceToHost));
    return T3_OK;
}

/* the hand-off kernel on host buffers: ids [n] speech-space -> out [ld] (padded with 0), *len */
extern "C" int t3k_handoff(const int32_t* ids, int32_t n, int32_t text_token_count, int32_t flags, int32_t* out, int32_t ld, int32_t* len) {
    if ((!ids && n > 0) || !out || !len || n < 0 || ld <= 0) return T3_E_INVALID;
    if (!have_device()) return T3_E_DEVICE;
    DevBuf dids, ditem, dout, dlen;
    K_TRY(dids.from(ids, (size_t)n * 4)); K_TRY(dout.alloc((size_t)ld * 4, true)); K_TRY(dlen.alloc(4, true));
    const HandoffItem item{dids.as<int>(), n, text_token_count, 0};
    K_TRY(ditem.from(&item, sizeof(item)));
    K_TRY(launch_handoff(ditem.as<HandoffItem>(), 1, flags, dout.as<int>(), ld, dlen.as<int>(), nullptr));
    K_TRY(hipDeviceSynchronize());
    K_TRY(hipMemcpy(out, dout.p, (size_t)ld * 4, hipMemcpyDeviceToHost));
    K_TRY(hipMemcpy(len, dlen.p, 4, hipMemcpyDeviceToHost));
    return T3_OK;
}

extern "C" int t3k_expf(const float* x, float* y, int32_t n) {
    if (!x || !y || n <= 0) return T3_E_INVALID;
    if (!have_device()) return T3_E_DEVICE;
    DevBuf dx, dy;
    K_TRY(dx.from(x, (size_t)n * 4)); K_TRY(dy.alloc((size_t)n * 4));
    K_TRY(launch_expf(dx.as<float>(), dy.as<float>(), n, nullptr));
    K_TRY(hipDeviceSynchronize());
    K_TRY(hipMemcpy(y, dy.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return T3_OK;
}
