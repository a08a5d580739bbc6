/*
 * t3_oracle.c -- CPU ORACLE for the T3 speech-token decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (chatterbox-vllm2_amd/, include/)
 * may include, link, import or execute this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * What it restates (reference = groxaxo/chatterbox-vllm2, paths relative to /root/reference):
 *   - prompt/embedding layout ............ src/chatterbox_vllm/models/t3/t3.py:189-221, 542-561
 *   - decode embedding ................... t3.py:440-486 (speech position index: SURVEY.md section 9 Q1)
 *   - dual-stream CFG forward ............ t3.py:696-713
 *   - CFG logits ......................... t3.py:650-673 (bf16 tensor arithmetic, t3.py:662)
 *   - Llama hyper-parameters ............. t3-model/config.json:1-33
 *   - constants .......................... models/t3/modules/t3_config.py:1-38, t3.py:38-49
 *   - sampling parameters ................ tts.py:455-464
 * The Llama block arithmetic, paged attention and the sampler live in vllm==0.10.0
 * (pyproject.toml:29), which is NOT in /root/reference and not installable here, and the
 * reference has no tests or golden vectors for this path:  PARITY UNPINNED by the reference
 * itself.  The oracle is pinned instead against (a) transformers.LlamaModel on the same
 * weights (tests/hf_gate.py: the tolerances; tests/test_oracle.py: the live and the committed-vector gates), (b) outputs of the
 * reference's own importable code (tests/golden/make_golden.py), (c) committed golden token streams, (d) for the sampler's masks
 * an independent restatement of vLLM's documented order over random cases (tests/test_oracle.py::test_sampler_support_*).
 *
 * NUMERICS CONTRACT (DESIGN.md "Numerics contract"): every rounding point and every
 * floating-point summation ORDER below is part of the specification.  The HIP kernels
 * implement the same orders, so logits and token ids are bit-identical, not just close.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define T3_D 1024
#define T3_H 16
#define T3_HD 64
#define T3_F 4096
#define T3_V 8194          /* speech vocab, t3_config.py:10 */
#define T3_COND 34         /* t3.py:42 */
#define T3_BOS 6561        /* start_speech_token, t3_config.py:8 */
#define T3_EOS 6562        /* stop_speech_token,  t3_config.py:9 */
#define T3_TEXT_POS 2050   /* t3.py:280 */
#define T3_SPEECH_POS 4100 /* t3.py:283 */
#define T3_EPS 1e-5f       /* config.json:20 */
#define T3_CHUNK 64        /* attention chunk in tokens (numerics contract; the product's physical KV block is 256 tokens = 4 chunks) */

/* ------------------------------------------------------------------ scalar helpers */
static inline float bf2f(uint16_t b) {
    uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f;
}
/* fp32 -> bf16, round to nearest even (NaN kept quiet) */
static inline uint16_t f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float rbf(float f) { return bf2f(f2bf(f)); }

/* Contract exp: Cody-Waite reduction + degree-6 Horner, only fma/mul/rint, so that the
 * GPU evaluates the identical sequence.  |rel err| ~ 1e-7.  Domain clamp [-87, 88]. */
float orc_expf(float x) {
    if (!(x >= -87.0f)) return 0.0f;      /* also -inf and NaN -> 0 */
    if (x > 88.0f) x = 88.0f;
    float n = rintf(x * 1.44269502162933349609375f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.428606765330187045e-06f, r);
    float p = 1.388888922519981861e-03f;            /* 1/720 */
    p = fmaf(p, r, 8.333333767950534821e-03f);      /* 1/120 */
    p = fmaf(p, r, 4.166666790843009949e-02f);      /* 1/24  */
    p = fmaf(p, r, 1.666666716337203979e-01f);      /* 1/6   */
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int32_t ni = (int32_t)n;
    uint32_t sb = (uint32_t)(ni + 127) << 23; float s; memcpy(&s, &sb, 4);
    return p * s;
}

/* ------------------------------------------------------------------ bf16 MFMA block model
 * Bit-exact model of how v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 (gfx950) fold ONE block of 8
 * consecutive k into the fp32 accumulator; fitted on-device with tools/mfma_probe*.{hip,py} (DESIGN.md
 * "bf16 MFMA numerics") -- 100 % of 170k random and crafted samples:
 *   products p_i = a_i*b_i are exact; e_ref = max_i (Ea_i + Eb_i) over non-zero products (operand exponents,
 *   not the product's); L = e_ref - 24; every product is truncated TOWARD ZERO to a multiple of 2^L, the
 *   incoming accumulator is FLOORED to a multiple of 2^L; the integer sum is exact; result = RNE_fp32(sum * 2^L).
 * An MFMA instruction applies this to its K/8 blocks in ascending k, chaining through the fp32 accumulator. */
static inline float mfma_finish(int32_t S32, int eref, float acc) {
    /* S32: exact sum of the block's truncated products on the grid 2^L, L = eref - 24 */
    int L = eref - 24;
    int64_t S = S32;
    uint32_t cu; memcpy(&cu, &acc, 4);
    const uint32_t ce = (cu >> 23) & 0xff;
    const int64_t cm = (int64_t)((cu & 0x7fffff) | (ce ? 0x800000u : 0u));
    if (cm) {
        const int sh = (int)(ce ? ce : 1) - 127 - 23 - L;
        if (sh > 38) return acc;                      /* the accumulator dwarfs the block: it comes back unchanged */
        int64_t t;
        if (sh >= 0) t = cm << sh;
        else if (sh > -63) t = (cu >> 31) ? ((cm + (((int64_t)1) << -sh) - 1) >> -sh) : (cm >> -sh);   /* |floor(acc / 2^L)| */
        else t = (cu >> 31) ? 1 : 0;
        S += (cu >> 31) ? -t : t;
    }
    const uint64_t mag = S < 0 ? (uint64_t)(-S) : (uint64_t)S;
    const int nb = mag ? 64 - __builtin_clzll(mag) : 0;
    if (nb > 32) { const int d = nb - 32; S >>= d; L += d; }     /* the adder keeps 32 significant bits (floor) */
    return scalbnf((float)S, L);                      /* int64 -> fp32 is RNE; the scaling is exact */
}
static inline float mfma_block8(float acc, const uint16_t* a, const uint16_t* b) {
    int32_t P[8], E[8]; int eref = -100000;
    for (int i = 0; i < 8; ++i) {
        const uint32_t ea = (a[i] >> 7) & 0xff, eb = (b[i] >> 7) & 0xff;
        const uint32_t ma = (a[i] & 0x7f) | (ea ? 0x80u : 0u), mb = (b[i] & 0x7f) | (eb ? 0x80u : 0u);
        int32_t pr = (int32_t)(ma * mb);
        if ((a[i] ^ b[i]) & 0x8000) pr = -pr;
        P[i] = pr; E[i] = (int)(ea ? ea : 1) + (int)(eb ? eb : 1) - 254;
        if (pr != 0 && E[i] > eref) eref = E[i];
    }
    if (eref == -100000) return acc;
    int32_t S = 0;
    for (int i = 0; i < 8; ++i) {
        if (!P[i]) continue;
        const int r = eref - E[i];                    /* value = P * 2^(E-14); on the grid 2^(eref-24): (|P| << 10) >> r */
        const int32_t mag = P[i] < 0 ? -P[i] : P[i];
        const int32_t t = r < 31 ? ((mag << 10) >> r) : 0;
        S += P[i] < 0 ? -t : t;
    }
    return mfma_finish(S, eref, acc);
}
float orc_mfma_bf16_dot(const uint16_t* a, const uint16_t* b, int K, float c) {
    for (int k = 0; k < K; k += 8) c = mfma_block8(c, a + k, b + k);
    return c;
}
/* probe replay: A [T][M][K], B [T][K][N] bf16 bits, C/D [T][M][N] */
void orc_mfma_batch(const uint16_t* A, const uint16_t* B, const float* C, float* Dm, int T, int M, int N, int K) {
#pragma omp parallel for
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < N; ++j) {
                uint16_t bc[64];
                for (int k = 0; k < K; ++k) bc[k] = B[((size_t)t * K + k) * N + j];
                Dm[((size_t)t * M + i) * N + j] = orc_mfma_bf16_dot(A + ((size_t)t * M + i) * K, bc, K, C[((size_t)t * M + i) * N + j]);
            }
}

/* ------------------------------------------------------------------ GEMM
 * y[m][n] = sum_k x[m][k] * W[n][k], x and W bf16, accumulated the way the gfx950 bf16 MFMA does:
 * K is cut into contiguous segments of seg_len (one per wave of a GPU workgroup); inside a segment the
 * blocks of 8 consecutive k are folded in ascending order with mfma_block8() starting from +0 (this is
 * what a chain of v_mfma_f32_16x16x32_bf16 over the segment computes).  Four consecutive segments form a
 * group G = ((s0 + s1) + s2) + s3 (one workgroup); with 16 segments the result is ((G0 + G1) + G2) + G3
 * (four workgroups, folded by the consumer kernel), with 4 segments it is G0.  All adds in fp32.
 * seg_len per op: qkv 256, o_proj 64, gate/up 256, down 256, speech head 256.
 * Weights are held pre-decoded and transposed ([K][N]): exponent (unbiased, -20000 for zero) and signed
 * 8-bit significand, so that the inner loops over n vectorise.                                     */
typedef struct { int16_t* e; int16_t* m; int K, N; } OrcW;

static OrcW orcw_make(const uint16_t* W /* [N][K] natural */, int N, int K) {
    OrcW w; w.K = K; w.N = N;
    w.e = (int16_t*)malloc((size_t)K * N * 2); w.m = (int16_t*)malloc((size_t)K * N * 2);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const uint32_t b = W[(size_t)n * K + k], eb = (b >> 7) & 0xff;
            const int32_t mb = (int32_t)((b & 0x7f) | (eb ? 0x80u : 0u));
            w.e[(size_t)k * N + n] = mb ? (int16_t)((int32_t)(eb ? eb : 1) - 127) : (int16_t)-20000;
            w.m[(size_t)k * N + n] = (int16_t)((b & 0x8000) ? -mb : mb);
        }
    return w;
}
static void orcw_free(OrcW* w) { free(w->e); free(w->m); w->e = w->m = NULL; }

#define GEMM_NB 256
void orc_gemm_w(const uint16_t* x, const OrcW* W, int M, float* out, int seg_len) {
    const int K = W->K, N = W->N, nseg = K / seg_len;   /* 4 or 16 */
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int nb = 0; nb < N; nb += GEMM_NB)
        for (int m = 0; m < M; ++m) {
            const int nw = (N - nb < GEMM_NB) ? (N - nb) : GEMM_NB;
            float acc[GEMM_NB], res[GEMM_NB], grp[GEMM_NB];
            int32_t eref[GEMM_NB], S[GEMM_NB];
            const uint16_t* xr = x + (size_t)m * K;
            for (int seg = 0; seg < nseg; ++seg) {
                for (int n = 0; n < nw; ++n) acc[n] = 0.0f;
                for (int k0 = seg * seg_len; k0 < (seg + 1) * seg_len; k0 += 8) {
                    int32_t Ea[8], Ma[8]; int live = 0;
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t a = xr[k0 + i], ea = (a >> 7) & 0xff;
                        const int32_t ma = (int32_t)((a & 0x7f) | (ea ? 0x80u : 0u));
                        Ma[i] = (a & 0x8000) ? -ma : ma;
                        Ea[i] = ma ? (int32_t)(ea ? ea : 1) - 127 : -20000;
                        live |= ma;
                    }
                    if (!live) continue;                       /* an all-zero x block leaves every accumulator unchanged */
                    for (int n = 0; n < nw; ++n) { eref[n] = -30000; S[n] = 0; }
                    for (int i = 0; i < 8; ++i) {
                        if (!Ma[i]) continue;
                        const int16_t* we = W->e + (size_t)(k0 + i) * N + nb;
                        const int32_t ea = Ea[i];
#pragma omp simd
                        for (int n = 0; n < nw; ++n) { const int32_t e = (int32_t)we[n] + ea; eref[n] = e > eref[n] ? e : eref[n]; }
                    }
                    for (int i = 0; i < 8; ++i) {
                        if (!Ma[i]) continue;
                        const int16_t* we = W->e + (size_t)(k0 + i) * N + nb;
                        const int16_t* wm = W->m + (size_t)(k0 + i) * N + nb;
                        const int32_t ea = Ea[i], ma = Ma[i];
#pragma omp simd
                        for (int n = 0; n < nw; ++n) {
                            const int32_t pr = ma * (int32_t)wm[n];
                            int32_t r = eref[n] - ((int32_t)we[n] + ea); r = r > 31 ? 31 : r;
                            const int32_t sg = pr >> 31, mag = (pr ^ sg) - sg;
                            const int32_t t = (mag << 10) >> r;
                            S[n] += (t ^ sg) - sg;
                        }
                    }
                    for (int n = 0; n < nw; ++n)
                        if (eref[n] > -10000) acc[n] = mfma_finish(S[n], eref[n], acc[n]);
                }
                for (int n = 0; n < nw; ++n) grp[n] = (seg % 4 == 0) ? acc[n] : (grp[n] + acc[n]);
                if (seg % 4 == 3)
                    for (int n = 0; n < nw; ++n) res[n] = (seg == 3) ? grp[n] : (res[n] + grp[n]);
            }
            memcpy(out + (size_t)m * N + nb, res, sizeof(float) * nw);
        }
}

/* RMSNorm folded into the consuming GEMM (DESIGN.md "RMSNorm"):  y = rstd[m] * GEMM(h, W'),  W'[n][k] = bf16(W[n][k] * w_ln[k])
 * (the norm weight is folded into the projection matrix ONCE, when the weights are loaded; the activations go to the matrix
 * cores untouched).  Row statistic, on the matrix cores as well: segment s = k / 256 (one wave of the GPU workgroup) folds
 * S_s = sum_k h_k * h_k over its 256 k with the SAME MFMA chain as a GEMM segment (blocks of 8 k ascending from +0: the wave
 * multiplies its A fragment with itself and reads the diagonal);  ss = ((S_0 + S_1) + S_2) + S_3;  rstd = 1 / sqrt(ss / 1024 + eps). */
void orc_row_rstd(const uint16_t* h, int rows, float* rstd) {
    for (int r = 0; r < rows; ++r) {
        const uint16_t* x = h + (size_t)r * T3_D;
        float S[4];
        for (int s = 0; s < 4; ++s) S[s] = orc_mfma_bf16_dot(x + 256 * s, x + 256 * s, 256, 0.0f);
        const float ss = ((S[0] + S[1]) + S[2]) + S[3];
        rstd[r] = 1.0f / sqrtf(ss * (1.0f / 1024.0f) + T3_EPS);
    }
}
/* W' = bf16(W (.) w) row by row: the load-time fold of an RMSNorm weight into the projection that consumes it */
static uint16_t* fold_ln(const uint16_t* W /* [N][1024] */, int N, const uint16_t* w) {
    uint16_t* out = (uint16_t*)malloc((size_t)N * T3_D * 2);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < T3_D; ++k) out[(size_t)n * T3_D + k] = f2bf(bf2f(W[(size_t)n * T3_D + k]) * bf2f(w[k]));
    return out;
}
/* out[m][n] = rstd[m] * GEMM(h, W')   (K = 1024, segments of 256; W' already carries the norm weight) */
void orc_norm_gemm_w(const uint16_t* h, const OrcW* Wf, int M, float* out) {
    float* rstd = (float*)malloc(sizeof(float) * M);
    orc_row_rstd(h, M, rstd);
    orc_gemm_w(h, Wf, M, out, 256);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < Wf->N; ++n) out[(size_t)m * Wf->N + n] = out[(size_t)m * Wf->N + n] * rstd[m];
    free(rstd);
}
void orc_norm_gemm_nk(const uint16_t* h, const uint16_t* w, const uint16_t* Wn, int M, int N, float* out) {
    uint16_t* f = fold_ln(Wn, N, w);
    OrcW W = orcw_make(f, N, T3_D);
    orc_norm_gemm_w(h, &W, M, out);
    orcw_free(&W); free(f);
}

/* Helper for tests: W given in its natural [N][K] layout. */
void orc_gemm_nk(const uint16_t* x, const uint16_t* W, int M, int K, int N, float* out, int seg_len) {
    OrcW w = orcw_make(W, N, K);
    orc_gemm_w(x, &w, M, out, seg_len);
    orcw_free(&w);
}

/* ------------------------------------------------------------------ 64-lane butterfly */
static void bfly_add(float v[64], const int* offs, int n) {
    float t[64];
    for (int s = 0; s < n; ++s) {
        for (int l = 0; l < 64; ++l) t[l] = v[l] + v[l ^ offs[s]];
        memcpy(v, t, sizeof(t));
    }
}

/* ------------------------------------------------------------------ RoPE table (llama3 scaling)
 * config.json:21-28.  inv_freq in double, angle = pos * inv_freq in double, cos/sin in
 * double -> fp32 -> bf16 (the cache is held in the model dtype as vLLM/HF do).
 * The product's host code computes the same table the same way.                         */
void orc_rope_table(int max_pos, float* cos_t, float* sin_t /* [max_pos][32] */) {
    double inv[32];
    const double theta = 500000.0, factor = 8.0, lo = 1.0, hi = 4.0, old = 8192.0;
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < 32; ++i) {
        double f = pow(theta, -(2.0 * i) / 64.0);
        double wl = 2.0 * pi / f;
        if (wl > old / lo) f = f / factor;
        else if (wl >= old / hi) {
            double s = (old / wl - lo) / (hi - lo);
            f = (1.0 - s) * f / factor + s * f;
        }
        inv[i] = (double)(float)f;       /* inv_freq is an fp32 buffer in HF */
    }
    for (int p = 0; p < max_pos; ++p)
        for (int i = 0; i < 32; ++i) {
            double a = (double)p * inv[i];
            cos_t[p * 32 + i] = rbf((float)cos(a));
            sin_t[p * 32 + i] = rbf((float)sin(a));
        }
}

/* rotate one 64-wide head in place (bf16), rotate-half pairing (i, i+32) */
static void rope_head(uint16_t* v, const float* c, const float* s) {
    for (int i = 0; i < 32; ++i) {
        float x1 = bf2f(v[i]), x2 = bf2f(v[i + 32]);
        float o1 = x1 * c[i] - x2 * s[i];     /* both products exact (bf16 x bf16) */
        float o2 = x2 * c[i] + x1 * s[i];
        v[i] = f2bf(o1); v[i + 32] = f2bf(o2);
    }
}
void orc_rope(uint16_t* qk /* [rows][1024] */, const int* pos, int rows, const float* cos_t, const float* sin_t) {
    for (int r = 0; r < rows; ++r)
        for (int h = 0; h < T3_H; ++h)
            rope_head(qk + (size_t)r * T3_D + h * T3_HD, cos_t + pos[r] * 32, sin_t + pos[r] * 32);
}

/* ------------------------------------------------------------------ attention (one query row, one head)
 * K,V: [L][stride] bf16 with this head's 64 values at kv[t*stride .. +64).
 * Context is cut into position-aligned chunks of 64 tokens; inside chunk c "lane" t is token 64c + t:
 *   score_t = MFMA chain over the 64 dims (8 blocks of 8 dims, ascending, from +0) of k_t . q;  s_t = score_t * 0.125,
 *             masked (position >= L) -> -inf
 *   m_c = max_t s_t (exact);  p_t = exp(s_t - m_c), masked -> 0
 *   l_c = 64-lane butterfly add of p (xor 32,16,8,4,2,1), fp32, UNROUNDED p
 *   pb_t = bf16(p_t), flushed to 0 below 2^-100 (the probabilities enter P.V in bf16, as in the bf16 HF / vLLM pipeline)
 *   o_c[d] = MFMA chain over the 64 tokens (8 blocks of 8 tokens, ascending, from +0) of v_t[d] . pb_t
 * Across chunks: M = max m_c; ascending c: w = exp(m_c - M); l = fma(w, l_c, l);
 *   o[d] = fma(w, o_c[d], o[d]);  out[d] = bf16(o[d] / l).                                   */
void orc_attn_row(const uint16_t* q, const uint16_t* K, const uint16_t* Vv, int L, int stride, uint16_t* out) {
    static const int offs[6] = {32, 16, 8, 4, 2, 1};
    const int nc = (L + T3_CHUNK - 1) / T3_CHUNK;
    float* mc = (float*)malloc(sizeof(float) * nc);
    float* lc = (float*)malloc(sizeof(float) * nc);
    float* oc = (float*)malloc(sizeof(float) * nc * 64);
    for (int c = 0; c < nc; ++c) {
        float s[64], p[64];
        uint16_t pb[64], vcol[64];
        float m = -INFINITY;
        for (int t = 0; t < 64; ++t) {
            const int pos = 64 * c + t;
            s[t] = (pos < L) ? orc_mfma_bf16_dot(K + (size_t)pos * stride, q, 64, 0.0f) * 0.125f : -INFINITY;
            m = fmaxf(m, s[t]);
        }
        for (int t = 0; t < 64; ++t) {
            p[t] = (s[t] == -INFINITY) ? 0.0f : orc_expf(s[t] - m);
            pb[t] = (p[t] < 0x1p-100f) ? 0 : f2bf(p[t]);
        }
        bfly_add(p, offs, 6);
        for (int d = 0; d < 64; ++d) {
            for (int t = 0; t < 64; ++t) { const int pos = 64 * c + t; vcol[t] = (pos < L) ? Vv[(size_t)pos * stride + d] : 0; }
            oc[c * 64 + d] = orc_mfma_bf16_dot(vcol, pb, 64, 0.0f);
        }
        mc[c] = m; lc[c] = p[0];
    }
    float M = -INFINITY;
    for (int c = 0; c < nc; ++c) M = fmaxf(M, mc[c]);
    float l = 0.0f, o[64];
    for (int d = 0; d < 64; ++d) o[d] = 0.0f;
    for (int c = 0; c < nc; ++c) {
        const float w = orc_expf(mc[c] - M);
        l = fmaf(w, lc[c], l);
        for (int d = 0; d < 64; ++d) o[d] = fmaf(w, oc[c * 64 + d], o[d]);
    }
    for (int d = 0; d < 64; ++d) out[d] = f2bf(o[d] / l);
    free(mc); free(lc); free(oc);
}

/* SwiGLU: act = bf16( bf16(silu(g)) * u ),  silu(g) = g / (1 + exp(-g)) */
static inline uint16_t silu_mul(uint16_t g, uint16_t u) {
    const float gf = bf2f(g);
    const float sg = rbf(gf / (1.0f + orc_expf(-gf)));
    return f2bf(sg * bf2f(u));
}
void orc_silu_mul(const uint16_t* g, const uint16_t* u, uint16_t* out, int n) {
    for (int i = 0; i < n; ++i) out[i] = silu_mul(g[i], u[i]);
}

/* ------------------------------------------------------------------ model */
typedef struct {
    OrcW wqkv;          /* N = 3072: q(0..1023) k v */
    OrcW wo;            /* N = 1024 */
    OrcW wgu;           /* N = 8192: gate(0..4095) up(4096..8191) */
    OrcW wd;            /* K = 4096, N = 1024 */
    uint16_t *sq, *sk, *sv, *sg, *su;   /* staging of the natural-layout parts until all have arrived (and the norm weight that is folded in) */
    uint16_t *ln1, *ln2;/* [1024] */
} OrcLayer;

typedef struct {
    int n_layers, text_vocab, max_pos;
    OrcLayer* layers;
    uint16_t *norm, *text_emb, *speech_emb, *text_pos, *speech_pos; OrcW head /* N = 8194, final norm weight folded in */;
    uint16_t* head_raw;
    float *cos_t, *sin_t;
    /* KV cache: [stream][layer][pos][2][1024] bf16 */
    int n_streams; uint16_t* kv;
} OrcModel;

OrcModel* orc_create(int n_layers, int text_vocab, int max_pos, int n_streams) {
    OrcModel* m = (OrcModel*)calloc(1, sizeof(OrcModel));
    m->n_layers = n_layers; m->text_vocab = text_vocab; m->max_pos = max_pos; m->n_streams = n_streams;
    m->layers = (OrcLayer*)calloc(n_layers, sizeof(OrcLayer));
    m->cos_t = (float*)malloc(sizeof(float) * max_pos * 32);
    m->sin_t = (float*)malloc(sizeof(float) * max_pos * 32);
    orc_rope_table(max_pos, m->cos_t, m->sin_t);
    m->kv = (uint16_t*)calloc((size_t)n_streams * n_layers * max_pos * 2 * T3_D, 2);
    return m;
}

static uint16_t* dup(const uint16_t* W, size_t n) { uint16_t* t = (uint16_t*)malloc(n * 2); memcpy(t, W, n * 2); return t; }

/* concatenate natural-layout [rows_i][1024] parts along N, fold the norm weight in, decode */
static OrcW orcw_concat_folded(const uint16_t* const* parts, const int* rows, int nparts, const uint16_t* ln) {
    int N = 0; for (int i = 0; i < nparts; ++i) N += rows[i];
    uint16_t* all = (uint16_t*)malloc((size_t)N * T3_D * 2); size_t off = 0;
    for (int i = 0; i < nparts; ++i) { memcpy(all + off, parts[i], (size_t)rows[i] * T3_D * 2); off += (size_t)rows[i] * T3_D; }
    uint16_t* f = fold_ln(all, N, ln);
    OrcW w = orcw_make(f, N, T3_D); free(all); free(f); return w;
}
/* build the folded matrices of a layer / the head as soon as their parts and their norm weight are all present */
static void orc_try_fold(OrcModel* m, int L) {
    if (L >= 0) {
        OrcLayer* y = &m->layers[L];
        if (y->sq && y->sk && y->sv && y->ln1 && !y->wqkv.e) {
            const uint16_t* p[3] = {y->sq, y->sk, y->sv}; const int r[3] = {T3_D, T3_D, T3_D};
            y->wqkv = orcw_concat_folded(p, r, 3, y->ln1);
            free(y->sq); free(y->sk); free(y->sv); y->sq = y->sk = y->sv = NULL;
        }
        if (y->sg && y->su && y->ln2 && !y->wgu.e) {
            const uint16_t* p[2] = {y->sg, y->su}; const int r[2] = {T3_F, T3_F};
            y->wgu = orcw_concat_folded(p, r, 2, y->ln2);
            free(y->sg); free(y->su); y->sg = y->su = NULL;
        }
    } else if (m->head_raw && m->norm && !m->head.e) {
        const uint16_t* p[1] = {m->head_raw}; const int r[1] = {T3_V};
        m->head = orcw_concat_folded(p, r, 1, m->norm);
        free(m->head_raw); m->head_raw = NULL;
    }
}

/* Tensor names follow the checkpoint (t3.py:300-332, tts.py:112-137): "tfmr.layers.N.self_attn.q_proj.weight" ...
 * All tensors bf16, natural [out][in] layout.  Returns 0 ok, -1 unknown name (ignored, as t3.py:316-319). */
int orc_set_tensor(OrcModel* m, const char* name, const uint16_t* data, int rows, int cols) {
    int L; char rest[128];
    if (sscanf(name, "tfmr.layers.%d.%127s", &L, rest) == 2) {
        if (L < 0 || L >= m->n_layers) return -1;
        OrcLayer* y = &m->layers[L];
        uint16_t** slot = NULL;
        if (!strcmp(rest, "self_attn.q_proj.weight")) slot = &y->sq;
        else if (!strcmp(rest, "self_attn.k_proj.weight")) slot = &y->sk;
        else if (!strcmp(rest, "self_attn.v_proj.weight")) slot = &y->sv;
        else if (!strcmp(rest, "mlp.gate_proj.weight")) slot = &y->sg;
        else if (!strcmp(rest, "mlp.up_proj.weight")) slot = &y->su;
        if (slot) { free(*slot); *slot = dup(data, (size_t)rows * cols); orc_try_fold(m, L); return 0; }
        if (!strcmp(rest, "self_attn.o_proj.weight")) { y->wo = orcw_make(data, rows, cols); return 0; }
        if (!strcmp(rest, "mlp.down_proj.weight")) { y->wd = orcw_make(data, rows, cols); return 0; }
        if (!strcmp(rest, "input_layernorm.weight")) { y->ln1 = dup(data, T3_D); orc_try_fold(m, L); return 0; }
        if (!strcmp(rest, "post_attention_layernorm.weight")) { y->ln2 = dup(data, T3_D); orc_try_fold(m, L); return 0; }
        return -1;
    }
    if (!strcmp(name, "tfmr.norm.weight")) { m->norm = dup(data, T3_D); orc_try_fold(m, -1); return 0; }
    if (!strcmp(name, "text_emb.weight")) { m->text_emb = dup(data, (size_t)rows * cols); return 0; }
    if (!strcmp(name, "speech_emb.weight")) { m->speech_emb = dup(data, (size_t)rows * cols); return 0; }
    if (!strcmp(name, "text_pos_emb.emb.weight")) { m->text_pos = dup(data, (size_t)rows * cols); return 0; }
    if (!strcmp(name, "speech_pos_emb.emb.weight")) { m->speech_pos = dup(data, (size_t)rows * cols); return 0; }
    if (!strcmp(name, "speech_head.weight")) { m->head_raw = dup(data, (size_t)rows * cols); orc_try_fold(m, -1); return 0; }
    return -1;
}

void orc_destroy(OrcModel* m) {
    for (int i = 0; i < m->n_layers; ++i) {
        OrcLayer* y = &m->layers[i];
        orcw_free(&y->wqkv); orcw_free(&y->wo); orcw_free(&y->wgu); orcw_free(&y->wd);
        free(y->sq); free(y->sk); free(y->sv); free(y->sg); free(y->su); free(y->ln1); free(y->ln2);
    }
    orcw_free(&m->head); free(m->head_raw);
    free(m->layers); free(m->norm); free(m->text_emb); free(m->speech_emb); free(m->text_pos);
    free(m->speech_pos); free(m->cos_t); free(m->sin_t); free(m->kv); free(m);
}

static inline uint16_t* kv_at(OrcModel* m, int stream, int layer, int pos) {
    return m->kv + ((((size_t)stream * m->n_layers + layer) * m->max_pos + pos) * 2) * T3_D;
}

/* Unified step over a set of rows (prefill rows and decode rows are the same thing):
 * h [rows][1024] bf16 residual stream in/out; row r belongs to stream row_stream[r] at position
 * row_pos[r]; all KV for positions < row_pos[r] of that stream must be in the cache or in this call.
 * If tap_layer >= 0, the residual stream after that many layers is copied to tap.                */
void orc_forward_rows(OrcModel* m, uint16_t* h, const int* row_stream, const int* row_pos, int rows,
                      int tap_layer, uint16_t* tap) {
    float* f = (float*)malloc((size_t)rows * 8192 * sizeof(float));
    uint16_t* qkv = (uint16_t*)malloc((size_t)rows * 3072 * 2);
    uint16_t* att = (uint16_t*)malloc((size_t)rows * T3_D * 2);
    uint16_t* act = (uint16_t*)malloc((size_t)rows * T3_F * 2);
    for (int L = 0; L < m->n_layers; ++L) {
        if (tap_layer == L && tap) memcpy(tap, h, (size_t)rows * T3_D * 2);
        OrcLayer* y = &m->layers[L];
        orc_norm_gemm_w(h, &y->wqkv, rows, f);
        for (size_t i = 0; i < (size_t)rows * 3072; ++i) qkv[i] = f2bf(f[i]);
        for (int r = 0; r < rows; ++r) {
            uint16_t* q = qkv + (size_t)r * 3072;
            const float* c = m->cos_t + row_pos[r] * 32; const float* s = m->sin_t + row_pos[r] * 32;
            for (int hh = 0; hh < T3_H; ++hh) { rope_head(q + hh * 64, c, s); rope_head(q + 1024 + hh * 64, c, s); }
            uint16_t* dst = kv_at(m, row_stream[r], L, row_pos[r]);
            memcpy(dst, q + 1024, T3_D * 2); memcpy(dst + T3_D, q + 2048, T3_D * 2);
        }
#pragma omp parallel for schedule(dynamic, 4) collapse(2)
        for (int r = 0; r < rows; ++r)
            for (int hh = 0; hh < T3_H; ++hh) {
                const uint16_t* base = kv_at(m, row_stream[r], L, 0);
                orc_attn_row(qkv + (size_t)r * 3072 + hh * 64, base + hh * 64, base + T3_D + hh * 64,
                             row_pos[r] + 1, 2 * T3_D, att + (size_t)r * T3_D + hh * 64);
            }
        orc_gemm_w(att, &y->wo, rows, f, 64);
        for (size_t i = 0; i < (size_t)rows * T3_D; ++i) h[i] = f2bf(bf2f(h[i]) + rbf(f[i]));
        orc_norm_gemm_w(h, &y->wgu, rows, f);
        for (int r = 0; r < rows; ++r)
            for (int i = 0; i < T3_F; ++i)
                act[(size_t)r * T3_F + i] = silu_mul(f2bf(f[(size_t)r * 8192 + i]), f2bf(f[(size_t)r * 8192 + 4096 + i]));
        orc_gemm_w(act, &y->wd, rows, f, 256);
        for (size_t i = 0; i < (size_t)rows * T3_D; ++i) h[i] = f2bf(bf2f(h[i]) + rbf(f[i]));
    }
    if (tap_layer == m->n_layers && tap) memcpy(tap, h, (size_t)rows * T3_D * 2);
    free(f); free(qkv); free(att); free(act);
}

/* final norm + speech head + CFG for one utterance.  hc/hu: residual rows of the cond / uncond
 * stream (bf16 [1024]).  out: 8194 fp32 logits (bf16-valued).  t3.py:650-662 in bf16 arithmetic:
 * d = bf16(lc - lu); e = bf16(cfg * d); l = bf16(lc + e).                                        */
void orc_cfg_logits(OrcModel* m, const uint16_t* hc, const uint16_t* hu, float cfg, float* out,
                    float* out_cond, float* out_uncond) {
    uint16_t x[2 * T3_D];
    memcpy(x, hc, T3_D * 2); memcpy(x + T3_D, hu, T3_D * 2);
    float* f = (float*)malloc(sizeof(float) * 2 * T3_V);
    orc_norm_gemm_w(x, &m->head, 2, f);
    for (int v = 0; v < T3_V; ++v) {
        const float lc = rbf(f[v]), lu = rbf(f[T3_V + v]);
        const float d = rbf(lc - lu);
        const float e = rbf(cfg * d);
        out[v] = rbf(lc + e);
        if (out_cond) out_cond[v] = lc;
        if (out_uncond) out_uncond[v] = lu;
    }
    free(f);
}

/* ------------------------------------------------------------------ Philox4x32-10 */
void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ------------------------------------------------------------------ sampler
 * Mirrors the order of the vLLM sampler configured at tts.py:455-464: penalties -> greedy
 * short-cut -> temperature -> min_p -> top_k -> top_p -> draw.  All probability mass
 * arithmetic is done on integer weights w = floor(exp(l - max) * 2^32) so that sums are
 * exact and order-free (DESIGN.md "Sampler").                                           */
typedef struct {
    float temperature, top_p, min_p, repetition_penalty, presence_penalty, frequency_penalty;
    int32_t top_k;           /* <= 0: disabled */
    int32_t max_tokens;
    int32_t ignore_eos;
    int32_t stop_token;      /* speech-space id, -1 none (reference: 6562 = 9062-2500, tts.py:458) */
    uint64_t seed;
    uint64_t uid;            /* utterance id: RNG stream key, independent of batch / sharding */
    int32_t pos_policy;      /* 0 exact speech position, 1 literal index 0 */
    int32_t _pad;
} OrcSampling;

/* keep_out (nullable): [8194] the SUPPORT of the draw -- 1 where the token can be returned (greedy: the one id).  The support is
 * what vLLM's masks decide; tests/test_oracle.py compares it with an independent restatement of those masks over random cases. */
static int orc_sample_impl(const float* logits_in, const uint16_t* counts, const OrcSampling* sp, uint32_t step, uint8_t* keep_out) {
    static float l[T3_V]; static uint64_t w[T3_V]; static uint8_t keep[T3_V];
    for (int v = 0; v < T3_V; ++v) {
        float x = logits_in[v];
        if (counts[v] > 0) {
            if (sp->repetition_penalty != 1.0f) x = (x > 0.0f) ? x / sp->repetition_penalty : x * sp->repetition_penalty;
            x = x - sp->frequency_penalty * (float)counts[v];
            x = x - sp->presence_penalty;
        }
        l[v] = x;
    }
    if (sp->temperature < 1e-5f) {           /* greedy: first maximum */
        int best = 0; for (int v = 1; v < T3_V; ++v) if (l[v] > l[best]) best = v;
        if (keep_out) { memset(keep_out, 0, T3_V); keep_out[best] = 1; }
        return best;
    }
    float mx = -INFINITY;
    for (int v = 0; v < T3_V; ++v) { l[v] = l[v] / sp->temperature; mx = fmaxf(mx, l[v]); }
    uint64_t wmax = 0;
    for (int v = 0; v < T3_V; ++v) {
        const float e = orc_expf(l[v] - mx);
        w[v] = (uint64_t)(e * 4294967296.0f);
        keep[v] = w[v] > 0; if (w[v] > wmax) wmax = w[v];
    }
    if (sp->min_p > 0.0f) {
        const double thr = (double)sp->min_p * (double)wmax;
        for (int v = 0; v < T3_V; ++v) if ((double)w[v] < thr) keep[v] = 0;
    }
    if (sp->top_k > 0 && sp->top_k < T3_V) {
        /* k-th largest kept weight; all ties at that value stay (vLLM masks strictly-below) */
        uint64_t lo = 0, hi = wmax;          /* largest t with count(w >= t) >= k */
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo + 1) / 2; int c = 0;
            for (int v = 0; v < T3_V; ++v) c += keep[v] && w[v] >= mid;
            if (c >= sp->top_k) lo = mid; else hi = mid - 1;
        }
        for (int v = 0; v < T3_V; ++v) if (w[v] < lo) keep[v] = 0;
    }
    if (sp->top_p < 1.0f) {
        uint64_t W = 0; for (int v = 0; v < T3_V; ++v) if (keep[v]) W += w[v];
        const uint64_t Tm = (uint64_t)((1.0 - (double)sp->top_p) * (double)W);
        /* ascending order (w asc, idx desc): drop the longest prefix whose cumulative mass <= Tm,
         * never the last element.  t* = largest t in [0, wmax] with sum(w < t) <= Tm.             */
        uint64_t lo = 0, hi = wmax;
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo + 1) / 2; uint64_t s = 0;
            for (int v = 0; v < T3_V; ++v) if (keep[v] && w[v] < mid) s += w[v];
            if (s <= Tm) lo = mid; else hi = mid - 1;
        }
        uint64_t below = 0; int ties = 0;
        for (int v = 0; v < T3_V; ++v) if (keep[v]) { if (w[v] < lo) below += w[v]; else if (w[v] == lo) ++ties; }
        uint64_t r = (lo > 0) ? (Tm - below) / lo : (uint64_t)ties;
        if (lo == wmax && r > (uint64_t)(ties - 1)) r = (uint64_t)(ties - 1);
        if (r > (uint64_t)ties) r = (uint64_t)ties;
        for (int v = T3_V - 1; v >= 0; --v) {
            if (!keep[v]) continue;
            if (w[v] < lo) keep[v] = 0;
            else if (w[v] == lo && r > 0) { keep[v] = 0; --r; }
        }
    }
    uint64_t Wk = 0; for (int v = 0; v < T3_V; ++v) if (keep[v]) Wk += w[v];
    if (keep_out) memcpy(keep_out, keep, T3_V);
    if (Wk == 0) return (sp->stop_token >= 0 && sp->stop_token < T3_V) ? sp->stop_token : 0;   /* no mass at all (NaN logits): end the utterance */
    uint32_t rnd[4];
    orc_philox(step, (uint32_t)sp->uid, (uint32_t)(sp->uid >> 32), 0, (uint32_t)sp->seed, (uint32_t)(sp->seed >> 32), rnd);
    const uint64_t u = ((uint64_t)rnd[1] << 32) | rnd[0];
    const uint64_t target = (uint64_t)(((unsigned __int128)u * Wk) >> 64);   /* in [0, Wk) */
    uint64_t cum = 0; int last = 0;
    for (int v = 0; v < T3_V; ++v) if (keep[v]) { cum += w[v]; last = v; if (cum > target) return v; }
    return last;
}
int orc_sample(const float* logits_in, const uint16_t* counts /* [8194] generated-token counts */,
               const OrcSampling* sp, uint32_t step) { return orc_sample_impl(logits_in, counts, sp, step, NULL); }
int orc_sample_support(const float* logits_in, const uint16_t* counts, const OrcSampling* sp, uint32_t step, uint8_t* keep_out /* [8194] */) {
    return orc_sample_impl(logits_in, counts, sp, step, keep_out);
}

/* ------------------------------------------------------------------ prompt embeddings  (t3.py:542-561)
 * prompt_ids [T] = [695, x*32, 696, text ids..., 697]; cond_emb fp32 [34][1024].
 * emb_c / emb_u: [T][1024] bf16.                                                            */
int orc_prompt_embeds(OrcModel* m, const int32_t* prompt_ids, int T, const float* cond_emb,
                      uint16_t* emb_c, uint16_t* emb_u) {
    if (T < T3_COND + 1) return -1;
    for (int i = 0; i < T3_COND; ++i)
        for (int d = 0; d < T3_D; ++d) emb_c[i * T3_D + d] = emb_u[i * T3_D + d] = f2bf(cond_emb[i * T3_D + d]);
    for (int i = T3_COND; i < T - 1; ++i) {
        const int id = prompt_ids[i], j = i - T3_COND;
        if (id < 0 || id >= m->text_vocab || j >= T3_TEXT_POS) return -2;
        for (int d = 0; d < T3_D; ++d) {
            emb_c[i * T3_D + d] = f2bf(bf2f(m->text_emb[(size_t)id * T3_D + d]) + bf2f(m->text_pos[(size_t)j * T3_D + d]));
            emb_u[i * T3_D + d] = 0;                 /* t3.py:556 zeros_like(text_emb) */
        }
    }
    for (int d = 0; d < T3_D; ++d)
        emb_c[(T - 1) * T3_D + d] = emb_u[(T - 1) * T3_D + d] =
            f2bf(bf2f(m->speech_emb[(size_t)T3_BOS * T3_D + d]) + bf2f(m->speech_pos[d]));
    return 0;
}

/* ------------------------------------------------------------------ decode embedding  (t3.py:440-486)
 * The row fed back for a generated speech token: speech_emb[tok] + speech_pos_emb[k], the same for the conditional and the
 * unconditional half (t3.py:480).  k = number of tokens generated so far (pos_policy 0, SURVEY.md 9 Q1 "exact") or 0 (pos_policy 1:
 * what row [0, 0, :] of the reference's decode branch holds for a single token; tests/golden/prompt_embeds.npz records both). */
int orc_decode_embed(OrcModel* m, int tok, int k, uint16_t* out /* [1024] */) {
    if (tok < 0 || tok >= T3_V || k < 0 || k >= T3_SPEECH_POS) return -1;
    for (int d = 0; d < T3_D; ++d)
        out[d] = f2bf(bf2f(m->speech_emb[(size_t)tok * T3_D + d]) + bf2f(m->speech_pos[(size_t)k * T3_D + d]));
    return 0;
}

/* ------------------------------------------------------------------ end-to-end generate (one utterance)
 * Uses streams 2*slot (cond) and 2*slot+1 (uncond) of the oracle's KV cache.
 * out_ids: speech-space ids (NOT offset by 2500).  logits_out (optional): [n][8194] post-CFG logits
 * of every step.  Returns the number of generated tokens, <0 on error.                     */
int orc_generate(OrcModel* m, int slot, const int32_t* prompt_ids, int T, const float* cond_emb,
                 const OrcSampling* sp, float cfg, int max_model_len, int32_t* out_ids, float* logits_out) {
    uint16_t* e = (uint16_t*)malloc((size_t)2 * T * T3_D * 2);
    int rc = orc_prompt_embeds(m, prompt_ids, T, cond_emb, e, e + (size_t)T * T3_D);
    if (rc) { free(e); return rc; }
    int* rs = (int*)malloc(sizeof(int) * 2 * T); int* rp = (int*)malloc(sizeof(int) * 2 * T);
    for (int i = 0; i < T; ++i) { rs[i] = 2 * slot; rp[i] = i; rs[T + i] = 2 * slot + 1; rp[T + i] = i; }
    orc_forward_rows(m, e, rs, rp, 2 * T, -1, NULL);
    uint16_t hc[T3_D], hu[T3_D], h2[2 * T3_D];
    memcpy(hc, e + (size_t)(T - 1) * T3_D, T3_D * 2);
    memcpy(hu, e + (size_t)(2 * T - 1) * T3_D, T3_D * 2);
    free(e); free(rs); free(rp);
    uint16_t* counts = (uint16_t*)calloc(T3_V, 2);
    float* lg = (float*)malloc(sizeof(float) * T3_V);
    int n = 0, limit = sp->max_tokens;
    if (limit > max_model_len - T) limit = max_model_len - T;
    while (n < limit) {
        orc_cfg_logits(m, hc, hu, cfg, lg, NULL, NULL);
        if (logits_out) memcpy(logits_out + (size_t)n * T3_V, lg, sizeof(float) * T3_V);
        const int tok = orc_sample(lg, counts, sp, (uint32_t)n);
        out_ids[n++] = tok;
        if (counts[tok] < 65535) counts[tok]++;
        if (!sp->ignore_eos && tok == sp->stop_token) break;
        if (n >= limit) break;
        /* decode embedding: speech_emb[tok] + speech_pos[k], k = n (exact) or 0 (literal) -- t3.py:440-480 */
        const int k = (sp->pos_policy == 0) ? (n % T3_SPEECH_POS) : 0;
        orc_decode_embed(m, tok, k, h2);
        memcpy(h2 + T3_D, h2, T3_D * 2);
        int s2[2] = {2 * slot, 2 * slot + 1}, p2[2] = {T - 1 + n, T - 1 + n};
        orc_forward_rows(m, h2, s2, p2, 2, -1, NULL);
        memcpy(hc, h2, T3_D * 2); memcpy(hu, h2 + T3_D, T3_D * 2);
    }
    free(counts); free(lg);
    return n;
}

/* Batched decode timing helper for bench.py's cpu_baseline leg: runs `steps` decode steps for
 * `B` utterances that have already been prefetched to context length ctx (KV content = whatever is
 * in the cache; timing only).  Returns nothing; caller times it.                            */
void orc_decode_steps_timing(OrcModel* m, int B, int ctx, int steps) {
    const int rows = 2 * B;
    uint16_t* h = (uint16_t*)calloc((size_t)rows * T3_D, 2);
    int* rs = (int*)malloc(sizeof(int) * rows); int* rp = (int*)malloc(sizeof(int) * rows);
    float* lg = (float*)malloc(sizeof(float) * T3_V);
    for (int s = 0; s < steps; ++s) {
        for (int r = 0; r < rows; ++r) {
            rs[r] = r; rp[r] = ctx + s;
            for (int d = 0; d < T3_D; ++d) h[(size_t)r * T3_D + d] = m->speech_emb[(size_t)((r * 131 + s) % T3_V) * T3_D + d];
        }
        orc_forward_rows(m, h, rs, rp, rows, -1, NULL);
        for (int b = 0; b < B; ++b) orc_cfg_logits(m, h + (size_t)(2 * b) * T3_D, h + (size_t)(2 * b + 1) * T3_D, 0.5f, lg, NULL, NULL);
    }
    free(h); free(rs); free(rp); free(lg);
}

/* ====================================================================================================
 * Conditioning encoder: reference cond_enc.py:57-123 (T3CondEnc.forward) and
 * perceiver.py:118-215 (AttentionBlock2, Perceiver), fp32 as the reference runs it (tts.py:277-284).
 *   row 0      = spkr_enc(speaker_emb)                                  cond_enc.py:86-87
 *   rows 1..32 = Perceiver(cond_prompt_speech_emb): attn(query, h) then attn(pre, pre), ONE AttentionBlock2
 *                (shared weights): LayerNorm both inputs, q/k/v Linear, 4 heads x 256 softmax(q k^T / 16) v,
 *                proj_out, + x1                                          perceiver.py:150-167, 203-212
 *   row 33     = emotion_adv_fc(emotion_adv) (Linear 1 -> 1024, no bias) cond_enc.py:103-106
 * Pinned against outputs of the reference's own T3CondEnc (tests/golden/cond_enc.npz, make_golden.py g1) within
 * an fp32 tolerance; the ORDER of every sum below is the contract the HIP kernels repeat bit for bit:
 *   dot      : acc = 0; k ascending: acc = fma(x[k], w[k], acc); linear = (acc + bias) [+ residual, added last]
 *   LayerNorm: lane l of 64 adds elements l, l+64, ... (16, ascending), butterfly xor 32..1; mean = sum/1024;
 *              second pass the same over fma(d, d, acc), d = x - mean; y = fma((x - mean) * rstd, w, b),
 *              rstd = 1/sqrt(var + 1e-5)
 *   attention: one (head, query): s_j = dot_256(q, k_j) * 0.0625; m = max; p_j = orc_expf(s_j - m); lane l adds
 *              p_l, p_{l+64}, p_{l+128} (ascending), butterfly; out_d = (j ascending: acc = fma(p_j, v_jd, acc)) / l
 * ==================================================================================================== */
#define CE_HEADS 4
#define CE_HD 256
#define CE_Q 32
#define CE_MAXK 192

void orc_ce_layernorm(const float* x, const float* w, const float* b, float* y, int rows) {
    static const int offs[6] = {32, 16, 8, 4, 2, 1};
    for (int r = 0; r < rows; ++r) {
        const float* xr = x + (size_t)r * T3_D;
        float v[64];
        for (int l = 0; l < 64; ++l) { float a = 0.0f; for (int i = 0; i < 16; ++i) a = a + xr[l + 64 * i]; v[l] = a; }
        bfly_add(v, offs, 6);
        const float mean = v[0] * (1.0f / 1024.0f);
        for (int l = 0; l < 64; ++l) { float a = 0.0f; for (int i = 0; i < 16; ++i) { const float d = xr[l + 64 * i] - mean; a = fmaf(d, d, a); } v[l] = a; }
        bfly_add(v, offs, 6);
        const float rstd = 1.0f / sqrtf(v[0] * (1.0f / 1024.0f) + 1e-5f);
        for (int i = 0; i < T3_D; ++i) y[(size_t)r * T3_D + i] = fmaf((xr[i] - mean) * rstd, w[i], b[i]);
    }
}

void orc_ce_linear(const float* x, const float* W, const float* bias, const float* resid, float* out, int M, int K, int N) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (int m = 0; m < M; ++m) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc = fmaf(x[(size_t)m * K + k], W[(size_t)n * K + k], acc);
            if (bias) acc = acc + bias[n];
            if (resid) acc = resid[(size_t)m * N + n] + acc;
            out[(size_t)m * N + n] = acc;
        }
}

int orc_ce_attention(const float* q, const float* k, const float* v, float* out, int nq, int nk) {
    static const int offs[6] = {32, 16, 8, 4, 2, 1};
    if (nk <= 0 || nk > CE_MAXK) return -1;
    for (int h = 0; h < CE_HEADS; ++h)
        for (int i = 0; i < nq; ++i) {
            float s[CE_MAXK], p[CE_MAXK], lane[64];
            for (int j = 0; j < nk; ++j) {
                float acc = 0.0f;
                for (int d = 0; d < CE_HD; ++d) acc = fmaf(q[(size_t)i * T3_D + h * CE_HD + d], k[(size_t)j * T3_D + h * CE_HD + d], acc);
                s[j] = acc * 0.0625f;
            }
            float m = -INFINITY;
            for (int j = 0; j < nk; ++j) m = fmaxf(m, s[j]);
            for (int j = 0; j < nk; ++j) p[j] = orc_expf(s[j] - m);
            for (int l = 0; l < 64; ++l) { float a = 0.0f; for (int j = l; j < nk; j += 64) a = a + p[j]; lane[l] = a; }
            bfly_add(lane, offs, 6);
            const float lsum = lane[0];
            for (int d = 0; d < CE_HD; ++d) {
                float acc = 0.0f;
                for (int j = 0; j < nk; ++j) acc = fmaf(p[j], v[(size_t)j * T3_D + h * CE_HD + d], acc);
                out[(size_t)i * T3_D + h * CE_HD + d] = acc / lsum;
            }
        }
    return 0;
}

/* params, in this order: spkr_w[1024][256], spkr_b[1024], emotion_w[1024], query[32][1024], ln_w[1024], ln_b[1024],
 * wq, bq, wk, bk, wv, bv, wo, bo  (matrices [1024][1024], biases [1024]).  prompt_emb [n][1024]; out [34][1024]. */
static void ce_block(const float* const* P, const float* x1, int n1, const float* x2, int n2, float* out) {
    float* a1 = (float*)malloc((size_t)n1 * T3_D * 4), *a2 = (float*)malloc((size_t)n2 * T3_D * 4);
    float* q = (float*)malloc((size_t)n1 * T3_D * 4), *k = (float*)malloc((size_t)n2 * T3_D * 4), *v = (float*)malloc((size_t)n2 * T3_D * 4);
    float* at = (float*)malloc((size_t)n1 * T3_D * 4);
    orc_ce_layernorm(x1, P[4], P[5], a1, n1);
    orc_ce_layernorm(x2, P[4], P[5], a2, n2);
    orc_ce_linear(a1, P[6], P[7], NULL, q, n1, T3_D, T3_D);
    orc_ce_linear(a2, P[8], P[9], NULL, k, n2, T3_D, T3_D);
    orc_ce_linear(a2, P[10], P[11], NULL, v, n2, T3_D, T3_D);
    orc_ce_attention(q, k, v, at, n1, n2);
    orc_ce_linear(at, P[12], P[13], x1, out, n1, T3_D, T3_D);
    free(a1); free(a2); free(q); free(k); free(v); free(at);
}

int orc_cond_enc(const float* const* P, const float* spk, const float* prompt_emb, int n, float emotion, float* out) {
    if (n <= 0 || n > CE_MAXK) return -1;
    orc_ce_linear(spk, P[0], P[1], NULL, out, 1, 256, T3_D);
    float* pre = (float*)malloc((size_t)CE_Q * T3_D * 4);
    ce_block(P, P[3], CE_Q, prompt_emb, n, pre);
    ce_block(P, pre, CE_Q, pre, CE_Q, out + T3_D);
    free(pre);
    for (int i = 0; i < T3_D; ++i) out[(size_t)33 * T3_D + i] = P[2][i] * emotion;
    return 0;
}
