/*
 * t3_engine.h -- C ABI of the MI355X-native T3 speech-token decode engine (libt3engine.so).
 *
 * This is the drop-in boundary for the ONE hot path of groxaxo/chatterbox-vllm2: everything that
 * happens between `LLM.generate(prompts, sampling_params)` and `output.token_ids`
 * (reference src/chatterbox_vllm/tts.py:445-465, 485) -- i.e. what the reference delegates to
 * vllm==0.10.0 plus its model plugin src/chatterbox_vllm/models/t3/t3.py.  The reference is pure
 * Python, so the "FFI" a maintainer binds is ctypes (see INTEGRATION.md for the stub).
 *
 * Conventions: every function returns 0 on success or a negative T3_E_* code; the message is
 * available from t3_last_error().  No exceptions cross the boundary.  The caller owns all host
 * buffers; the engine owns all device memory.  One handle per GPU; a handle is NOT thread-safe
 * (one driver thread per handle -- the reference's caller is a single blocking FastAPI handler,
 * api_server.py:265-271).  Token ids at this boundary are in the reference's OFFSET space
 * (speech id + 2500, t3.py:49,669-672) unless a function says "speech-space".
 */
#ifndef T3_ENGINE_H
#define T3_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T3_OK 0
#define T3_E_INVALID (-1)   /* bad argument (maps to ValueError -> HTTP 400, api_server.py:323-326) */
#define T3_E_DEVICE (-2)    /* HIP error */
#define T3_E_NOMEM (-3)     /* KV pool / device memory exhausted */
#define T3_E_STATE (-4)     /* call not valid in this state (e.g. step before finalize_weights) */
#define T3_E_NOTFOUND (-5)  /* unknown request id / tensor name */

#define T3_SPEECH_TOKEN_OFFSET 2500 /* t3.py:49 */
#define T3_COND_ROWS 34             /* t3.py:42 */
#define T3_HIDDEN 1024
#define T3_SPEECH_VOCAB 8194        /* t3_config.py:10 */

typedef struct T3Engine* T3Handle;

/* Replaces the kwargs of `LLM(model=..., gpu_memory_utilization=, enforce_eager=, max_model_len=, **kw)`
 * (tts.py:150-171) and the env var CHATTERBOX_CFG_SCALE (t3.py:296). */
typedef struct {
    int32_t device_id;
    int32_t n_layers;          /* 30 for the real model (config.json:17); small values for tests */
    int32_t text_vocab;        /* 704 English / 2454 multilingual (t3.py:270) */
    int32_t max_model_len;     /* tts.py:164 */
    int32_t max_seqs;          /* utterance slots (vLLM max_num_seqs); each slot = 2 CFG streams */
    int32_t max_batched_rows;  /* rows (stream-tokens) per step budget, >= 2*max_seqs; 0 = default */
    int64_t kv_bytes;          /* KV pool size; 0 = derive from gpu_memory_utilization */
    float gpu_memory_utilization; /* fraction of total HBM the engine may use (tts.py:162) */
    float cfg_scale;           /* CHATTERBOX_CFG_SCALE, default 0.5 (t3.py:296) */
    int32_t enforce_eager;     /* 1: never use hipGraph replay (tts.py:163) */
    int32_t debug_logits;      /* 1: keep post-CFG logits of each sampled step for t3_debug_logits */
    int32_t n_groups;          /* utterance groups run concurrently on separate HIP streams; 0 = auto */
    int32_t _pad;
} T3EngineConfig;

/* Replaces vllm.SamplingParams as configured at tts.py:455-464 (+ the kwargs it forwards). */
typedef struct {
    float temperature;         /* < 1e-5 => greedy */
    float top_p;               /* 1.0 = off */
    float min_p;               /* 0.0 = off (reference accepts min_p but never forwards it, tts.py:415) */
    float repetition_penalty;  /* 1.0 = off; reference default 2.0 (tts.py:416) */
    float presence_penalty;
    float frequency_penalty;
    int32_t top_k;             /* <= 0 = off */
    int32_t max_tokens;        /* tts.py:459 */
    int32_t ignore_eos;        /* fixed-length generation for synthetic benchmarks */
    int32_t stop_token;        /* SPEECH-SPACE stop id, 6562 (= 9062 - 2500, tts.py:458); -1 none */
    uint64_t seed;
    uint64_t uid;              /* utterance id keying the RNG stream: results do not depend on batch
                                  composition or on data-parallel sharding */
    int32_t pos_policy;        /* 0: k-th generated token uses speech_pos_emb[k] (SURVEY.md 9 Q1 "exact");
                                  1: index 0 for every decode token (the literal fallback of t3.py:464) */
    int32_t _pad;
} T3Sampling;

typedef struct {
    int32_t n_rows;            /* rows processed this step */
    int32_t n_prefill_rows;
    int32_t n_sampled;         /* utterances that produced a token this step */
    int32_t n_finished;        /* finished this step */
    int32_t n_running;         /* after the step */
    int32_t n_waiting;
    int64_t finished_ids[64];  /* first min(n_finished, 64); every finished id, in order, is also queued for t3_pop_finished */
} T3StepResult;

typedef struct {
    int64_t steps;
    int64_t decode_steps;          /* steps with no prefill rows */
    int64_t tokens_generated;
    int64_t prefill_rows;
    int64_t decode_rows;
    double gpu_ms_total;           /* HIP-event time of all steps */
    double gpu_ms_decode;          /* ... of decode-only steps */
    double algo_bytes_decode;      /* SURVEY.md 8(d) formula summed over decode-only steps */
    double sum_ctx_decode;         /* sum over decode rows of context length */
    int64_t kv_blocks_total, kv_blocks_free;
    int64_t weight_bytes;
    int64_t finished_dropped;      /* ids t3_pop_finished will never return: the queue of finished ids holds 4 * max_seqs + 4096 entries and drops the oldest beyond that (a caller that never pops must not grow it) */
} T3Stats;

/* ---- lifecycle ------------------------------------------------------------------------- */
int t3_create(const T3EngineConfig* cfg, T3Handle* out);
int t3_destroy(T3Handle h);                      /* `del self.t3`, tts.py:527-529 */
const char* t3_last_error(T3Handle h);           /* h may be NULL for create() failures */

/* ---- weights: T3VllmModel.load_weights (t3.py:300-332) ---------------------------------
 * name: checkpoint tensor name ("tfmr.layers.3.self_attn.q_proj.weight", "speech_emb.weight",
 * "text_pos_emb.emb.weight", "speech_head.weight", ...).  Unknown prefixes (cond_enc.*, text_head.*)
 * return T3_E_NOTFOUND and may be ignored by the caller, as t3.py:316-319 does.
 * data: host or device pointer to bf16 ([rows][cols], row-major).                          */
int t3_load_tensor(T3Handle h, const char* name, const void* data_bf16, int32_t rows, int32_t cols);
int t3_finalize_weights(T3Handle h);             /* packs weights for the GEMM kernels, builds RoPE tables, allocates the KV pool */

/* ---- requests: LLM.generate (tts.py:445-465) -------------------------------------------
 * prompt_ids: the FINAL prompt [695, x*32, 696, text ids..., 697] of length T (t3.py:189-200);
 * cond_emb: fp32 [34][1024] host buffer (tts.py:286 hands the conditionals over on CPU).   */
int t3_add_request(T3Handle h, int64_t req_id, const int32_t* prompt_ids, int32_t T,
                   const float* cond_emb, const T3Sampling* sampling);
int t3_step(T3Handle h, T3StepResult* res);      /* admit -> prefill/decode rows -> sample, once */
int t3_run_until_done(T3Handle h);               /* C++ loop, no Python per step */
/* Runs at most n steps (stops early when nothing is left); *done = steps actually run. */
int t3_run_steps(T3Handle h, int32_t n, int32_t* done);
int t3_num_unfinished(T3Handle h);
/* Ids of the requests that finished since the previous call (any of t3_step / t3_run_steps / t3_run_until_done), oldest first:
 * up to `cap` of them are written to ids and leave the queue.  Returns how many were written (a step of 128 utterances can retire
 * more than the 64 that fit T3StepResult), or a negative T3_E_* code.  The queue keeps the most recent 4 * max_seqs + 4096 ids: a
 * serving loop pops after every step; a caller that reads outputs by request id need not pop at all.  vLLM hands finished RequestOutputs back from every
 * engine step (LLM.generate collects them, tts.py:445-465); this is that stream of ids. */
int t3_pop_finished(T3Handle h, int64_t* ids, int32_t cap);
/* ids: offset-space token ids (>= 2500), the stop id included when hit.
 * On entry *n = capacity; on exit *n = number of tokens.  finish_reason: 0 running, 1 stop, 2 length. */
int t3_get_output(T3Handle h, int64_t req_id, int32_t* ids, int32_t* n, int32_t* finish_reason);
/* Host wall-clock marks of a request, seconds since t3_create: out4 = {added (t3_add_request), admitted (slot + KV blocks
 * granted), first token read back, finished}; 0 where not reached yet.  The reference's clock is the wall time of its
 * `self.t3.generate` call (tts.py:444,466-467); per request, RTF = (finished - admitted) / (n_tokens / 25). */
int t3_get_timing(T3Handle h, int64_t req_id, double* out4);
int t3_release_request(T3Handle h, int64_t req_id);   /* forget a finished request */
/* Drop a request in any state -- vLLM's abort_request: a WAITING one leaves the queue, a running one gives its slot and KV
 * blocks back -- and forget it.  The host mirror uses it to roll back a generate() call whose later prompt was rejected
 * (api_server.py:323-326 turns that ValueError into HTTP 400 and the engine must be left empty).  Not valid while a step is
 * in flight (the engine is single-threaded: call it between t3_step / t3_run_* calls). */
int t3_abort_request(T3Handle h, int64_t req_id);

/* ---- token post-filter: ChatterboxTTS.analyze_and_clean_tokens (tts.py:300-365) driving the fork's
 * token-heuristic AlignmentStreamAnalyzer.step (models/t3/inference/alignment_stream_analyzer.py:111-201), plus the
 * range filter of tts.py:514.  Pure integer logic on the host (the reference does it with one CUDA tensor + .item()
 * sync per token).  speech_ids: speech-space ids (already un-offset, tts.py:492); text_token_count as tts.py:496.
 * The output is the prefix before the first token at which the analyzer forces EOS -- three identical tokens in a
 * row, or >= 10 frames after the estimated end of the text (frame/2 >= text_token_count - 3).
 * flags bit 0: also drop ids outside [0, 6561) (tts.py:514).  reason (nullable): 0 none, 1 repetition, 2 long tail.
 * Returns the number of tokens written to out (capacity n), or a negative T3_E_* code.                        */
int t3_clean_tokens(const int32_t* speech_ids, int32_t n, int32_t text_token_count, int32_t flags,
                    int32_t* out, int32_t* reason);

/* ---- T3 -> S3Gen hand-off --------------------------------------------------------------------------
 * Replaces the per-utterance loop of ChatterboxTTS.generate_with_conds after the engine call (tts.py:483-514): `token - 2500`,
 * analyze_and_clean_tokens (one CUDA tensor + .item() sync per token, tts.py:335-344), torch.tensor(ids, device="cuda")
 * (tts.py:365) and the [0, 6561) mask (tts.py:514), once per utterance.  Here the ids never leave the device: the sampler keeps
 * every utterance's ids in HBM, and this call filters, compacts and pads a whole batch in one launch, straight into the caller's
 * DEVICE buffers in the layout a batched vocoder front-end takes (speech_tokens [n][ld] int32, speech_token_lens [n]; the
 * reference's flow.inference asserts batch 1, flow.py:256, so today's caller slices row i to lens[i]).
 * req_ids: n FINISHED, not yet released requests; text_token_counts as tts.py:496; flags bit 0: range filter of tts.py:514.
 * dev_tokens: device int32 [n][ld], rows padded with 0 beyond lens; dev_lens: device int32 [n].  Returns when the buffers are
 * ready (the call synchronises its own stream).                                                                        */
int t3_handoff_tokens(T3Handle h, const int64_t* req_ids, int32_t n, const int32_t* text_token_counts, int32_t flags,
                      int32_t* dev_tokens, int32_t ld, int32_t* dev_lens);
/* Switches the retention of finished utterances' ids in device memory on (n_requests > 0) or off (0; the default: a caller
 * that reads ids with t3_get_output pays nothing for the hand-off), and makes sure buffers for n_requests finished-and-unreleased
 * utterances exist BEFORE the step loop runs (max_model_len int32 each).  If more than that finish unreleased, the pool grows inside
 * the loop and an allocation failure there is reported by the step call (T3_E_NOMEM), not by a later t3_handoff_tokens. */
int t3_reserve_handoff(T3Handle h, int32_t n_requests);

/* ---- parity / measurement hooks --------------------------------------------------------- */
/* post-CFG logits [8194] (speech-space, before the 2500-wide -inf pad of t3.py:669-672) of the
 * most recent sampled step of req_id; needs cfg.debug_logits = 1. */
int t3_debug_logits(T3Handle h, int64_t req_id, float* out_8194);
/* The embedded input rows of the most recent step -- what get_input_embeddings (t3.py:424-647) hands to the Llama blocks: cond /
 * text + position / zero / speech + position rows, bf16 [rows][1024] -- with every row's stream (2 * slot for the conditional
 * half, 2 * slot + 1 for the unconditional half of the reference's [N, 2048] layout) and position; needs cfg.debug_logits = 1.
 * On entry *n = capacity in rows (out_bf16 / row_stream / row_pos may be NULL); on exit *n = rows of the step. */
int t3_debug_embeddings(T3Handle h, void* out_bf16, int32_t* row_stream, int32_t* row_pos, int32_t* n);
/* Durations (ms) of the most recent steps since t3_reset_stats, oldest first (at most cap, at most the last 16 384): the time each step
 * had the GPU to itself, i.e. from the completion of the step before it (or its own enqueue, if later) to its own completion as the host
 * sees it (the same clock gpu_ms_total sums).  rows (nullable): the step's row count, negative when it carried prefill rows.
 * Returns the number written.  For the p50 / p90 / p99 step latency SURVEY.md 8(d) asks for. */
int t3_step_times(T3Handle h, float* ms, int32_t* rows, int32_t cap);
int t3_stats(T3Handle h, T3Stats* out);
int t3_reset_stats(T3Handle h);
/* Average duration (ms) per launch of each kernel class over decode-only steps since the last reset,
 * measured with HIP events on the engine's stream when profiling is on (t3_set_profile).
 * names: "gemm_qkv","gemm_o","gemm_gateup","gemm_down","gemm_head","attention","rope_kv","embed","sampler".
 * Single-kernel launches (the decode GEMM forms, the fused / per-row attention, embed, sampler) are timed by the dispatch's own start / stop
 * events (what rocprofv3 reports); launchers that issue several kernels (prefill-sized GEMMs with their row-statistic pass, the tile attention
 * with its per-row head, rope_kv) are bracketed by two event records, ~2-3 us more per launch.  Only decode-only steps are accumulated, where every
 * class is of the first kind. */
int t3_set_profile(T3Handle h, int32_t on);
/* Restrict the events to ONE kernel class (name as above; NULL or "" = every class): the other launches of the step then run
 * undisturbed, so the class's average duration is not inflated by its neighbours' event packets. */
int t3_set_profile_kernel(T3Handle h, const char* name);
int t3_kernel_ms(T3Handle h, const char* name, double* avg_ms, int64_t* launches);

/* ---- conditioning encoder --------------------------------------------------
 * Replaces the reference's torch module `T3CondEnc` (models/t3/modules/cond_enc.py:57-123 with modules/perceiver.py:118-215),
 * as ChatterboxTTS uses it: `self.t3_cond_enc(T3Cond(speaker_emb, cond_prompt_speech_tokens, cond_prompt_speech_emb,
 * emotion_adv))` (tts.py:279-284) and `self.t3_cond_enc.emotion_adv_fc(exaggeration)` (tts.py:287-298).  fp32, on the device;
 * a handle of its own because the reference keeps this module outside the vLLM engine.  No CPU fallback.            */
typedef struct T3CondEncoder* T3CondHandle;
int t3_cond_create(int32_t device_id, T3CondHandle* out);
int t3_cond_destroy(T3CondHandle c);
const char* t3_cond_last_error(T3CondHandle c);          /* c may be NULL after a failed create */
/* fp32 parameters under their checkpoint names, with or without the "cond_enc." prefix (state_dict of T3CondEnc):
 * spkr_enc.{weight [1024][256], bias}, emotion_adv_fc.weight [1024][1], perceiver.pre_attention_query [1][32][1024],
 * perceiver.attn.norm.{weight,bias}, perceiver.attn.{to_q,to_k,to_v,proj_out}.{weight [1024][1024], bias}.
 * Other names return T3_E_NOTFOUND. */
int t3_cond_load_tensor(T3CondHandle c, const char* name, const float* data, int64_t numel);
/* T3CondEnc.forward: speaker_emb [256], cond_prompt_speech_emb [n][1024] (= speech_emb(tokens) + speech_pos_emb(tokens),
 * tts.py:277; 1 <= n <= 192, the reference uses 150), emotion_adv scalar -> out [34][1024] = [speaker; 32 perceiver rows; emotion]. */
int t3_cond_encode(T3CondHandle c, const float* speaker_emb, const float* prompt_emb, int32_t n, float emotion_adv, float* out);
/* emotion_adv_fc(exaggeration) -> out [1024] (the row ChatterboxTTS.update_exaggeration writes into cond_emb[-1]) */
int t3_cond_emotion_row(T3CondHandle c, float exaggeration, float* out);
/* kernel-level entry points of the encoder (host buffers), for the parity tests */
int t3k_ce_layernorm(const float* x /*[rows][1024]*/, const float* w, const float* b, float* y, int32_t rows);
int t3k_ce_linear(const float* x /*[M][K]*/, const float* W /*[N][K]*/, const float* bias /*nullable*/, const float* resid /*nullable [M][N]*/,
                  float* out /*[M][N]*/, int32_t M, int32_t K /*multiple of 32*/, int32_t N);
int t3k_ce_attention(const float* q /*[nq][1024]*/, const float* k /*[nk][1024]*/, const float* v, float* out /*[nq][1024]*/, int32_t nq, int32_t nk);

/* ---- kernel-level entry points (host buffers in, host buffers out; used by the parity tests) ----
 * Each runs exactly the kernel the engine uses, on the current device, and waits for it.
 * t3k_gemm: the engine's K only -- 1024 with 4 segments; 1024 or 4096 with 16 (N a multiple of 16 there); anything else is T3_E_INVALID. */
int t3k_gemm(const void* x_bf16 /*[M][K]*/, const void* w_bf16 /*[N][K]*/, int32_t M, int32_t K, int32_t N,
             float* out_f32 /*[M][N]*/, int32_t mt /*0 = auto*/, int32_t nw /*K segments: 4, or 16 = o_proj/down_proj form*/);
/* RMSNorm folded into the projection (qkv / gate-up / speech-head form): out[r] = rstd * GEMM(bf16(h[row_index[r]] * ln_w), W);
 * h [Mh][1024] bf16, row_index nullable (then M == Mh). */
int t3k_norm_gemm(const void* h_bf16, const void* ln_w_bf16, const void* w_bf16 /*[N][1024]*/, int32_t M, int32_t N,
                  float* out_f32, const int32_t* row_index, int32_t Mh);
/* the qkv projection as a step launches it (input RMSNorm folded): w [3072][1024] (q, k, v rows), out bf16 [M][3072] */
int t3k_qkv_gemm(const void* h_bf16, const void* ln_w_bf16, const void* w_bf16, int32_t M, void* out_bf16);
/* the speech head as a decode step launches it (final RMSNorm folded, rows gathered through row_index [M] into h [Mh][1024], t3.py:650-673
 * before the CFG): out bf16 [M][8208], columns >= 8194 unspecified */
int t3k_head_gemm(const void* h_bf16, const void* ln_w_bf16, const void* w_bf16 /*[8194][1024]*/, int32_t M, const int32_t* row_index, int32_t Mh, void* out_bf16);
/* residual epilogue (o_proj / down_proj form, 16 K-segments): h [M][N] bf16 updated in place, h = bf16(h + bf16(x W^T)) */
int t3k_gemm_resid(const void* x_bf16, const void* w_bf16, int32_t M, int32_t K, int32_t N, void* h_bf16);
/* gate/up form with folded RMSNorm and SiLU*mul epilogue */
int t3k_silu_mul_gemm(const void* h_bf16 /*[M][1024]*/, const void* ln_w_bf16, const void* wg_bf16 /*[F][1024]*/, const void* wu_bf16,
                      int32_t M, int32_t F, void* out_bf16 /*[M][F]*/);
/* RoPE + paged-KV write + paged attention for `rows` rows of ONE layer over a scratch pool:
 * qkv [rows][3072] bf16 (pre-RoPE), row_stream/row_pos int32 [rows]; rows are processed in the given
 * order in ONE launch sequence (all KV writes, then all attention), so rows of the same stream must
 * cover a prefix of positions.  out [rows][1024] bf16.                                        */
int t3k_rope_attention(const void* qkv_bf16, const int32_t* row_stream, const int32_t* row_pos, int32_t rows,
                       int32_t n_streams, int32_t max_pos, void* out_bf16);
/* The FUSED decode attention as a decode step launches it (RoPE of q / k + paged write of the newest K / V + attention over the
 * paged context in ONE kernel; what vLLM's rotary_embedding + reshape_and_cache + paged attention do per layer behind
 * t3.py:703-708), `steps` consecutive launches of `rows` rows over a scrambled scratch pool:
 *   stream r (= row r) holds ctx[r] - 1 context tokens, pre-RoPE qkv rows ctx_qkv[r % n_content][0 .. ctx[r] - 2] ([n_content][content_rows][3072]
 *   bf16), written through the prefill path; launch s takes new_qkv[s][r] ([steps][rows][3072]) at position ctx[r] - 1 + s, so launch
 *   s >= 1 reads what launch s - 1 wrote.  waves: 0 = as the engine picks (8 up to 8 rows, else 4), or 4 / 8.
 * out bf16 [steps][rows][1024]; kv_new (nullable) bf16 [steps][rows][2][1024] = K (rotated) and V of the written positions, read back. */
int t3k_decode_attention(const void* ctx_qkv_bf16, int32_t n_content, int32_t content_rows, const void* new_qkv_bf16, const int32_t* ctx,
                         int32_t rows, int32_t steps, int32_t max_pos, int32_t waves, void* out_bf16, void* kv_new_bf16);
/* CFG + sampler: logits bf16 [2][ldl] (cond row, uncond row), counts uint16 [8194] (updated). */
int t3k_sample(const void* logits2_bf16, int32_t ldl, uint16_t* counts, const T3Sampling* sp, float cfg,
               uint32_t step, int32_t* token_out, float* logits_out_8194 /*nullable*/);
/* t3k_sample plus the SUPPORT of its draw: keep_out [8194] = 1 where penalties -> /T -> min-p -> top-k -> top-p leave the id drawable
 * (greedy: the one id).  Parity hook of the randomized mask test (tests/vllm_masks.py). */
int t3k_sample_support(const void* logits2_bf16, int32_t ldl, uint16_t* counts, const T3Sampling* sp, float cfg,
                       uint32_t step, int32_t* token_out, uint8_t* keep_out_8194);
int t3k_expf(const float* x, float* y, int32_t n);
/* the hand-off kernel on host buffers (one utterance): out [ld] padded with 0, *len = kept tokens */
int t3k_handoff(const int32_t* speech_ids, int32_t n, int32_t text_token_count, int32_t flags, int32_t* out, int32_t ld, int32_t* len);
/* Row count from which the GEMM launcher switches to its prefill schedule (same numbers, LDS-tiled); process-wide.
 * < 0 restores the default (per form, 448-1600 rows).  For the parity tests, which check the prefill schedule at small row counts. */
int t3k_set_prefill_rows(int32_t rows);
/* Row count from which the 4-segment forms of that schedule take 128 x 128 workgroup tiles instead of 128 x 64 (0 = never,
 * < 0 = default 2048).  Same purpose. */
int t3k_set_prefill_wide_rows(int32_t rows);

#ifdef __cplusplus
}
#endif
#endif /* T3_ENGINE_H */
