// T3 decode engine runtime: weight store, paged-KV block allocator, continuous-batching scheduler,
// the per-step launch sequence, and the C ABI of include/t3_engine.h.
//
// What it replaces in the reference (paths relative to the reference repo): the vLLM engine
// constructed at src/chatterbox_vllm/tts.py:150-171 and driven at tts.py:445-465, plus the model
// plugin methods it calls (src/chatterbox_vllm/models/t3/t3.py:300-332 load_weights, :424-647
// get_input_embeddings, :676-713 forward, :650-673 compute_logits).  The reference's per-step
// split_prefill_decode (t3.py:340-421) has no equivalent here: the scheduler knows which rows are
// prefill and which are decode.
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <unordered_map>
#include <tuple>
#include <vector>

#include "t3_kernels.h"

using namespace t3;

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) return e->fail(T3_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

static thread_local std::string g_create_error;

namespace {

struct LayerW {
    uint16_t *qkv = nullptr, *o = nullptr, *gu = nullptr, *down = nullptr;  // device, packed (qkv / gu carry their RMSNorm weight, fold_norm_weight)
    std::vector<uint16_t> h_q, h_k, h_v, h_g, h_u, h_ln1, h_ln2;   // host staging until finalize
    bool have_o = false, have_d = false;
};

enum ReqState { WAITING = 0, PREFILL = 1, DECODE = 2, FINISHED = 3 };

struct Request {
    int64_t id;
    std::vector<int32_t> prompt;
    std::vector<float> cond;
    T3Sampling sp;
    int state = WAITING, slot = -1, n_prefilled = 0, finish_reason = 0, limit = 0;
    std::vector<int32_t> out;         // speech-space ids
    std::vector<int> blocks[2];
    // run-ahead scheduling: tokens whose sampling has been enqueued (>= out.size(); the difference is in flight),
    // index of the newest one in its group's sampler output array, and whether a row of an already finished
    // request is still in flight (its slot is released when that step completes)
    int n_sched = 0, last_idx = -1;
    int n_seen = 0;                   // sampled results read back (kept or, once finished, dropped): == n_sched when nothing of it is in flight
    bool zombie = false;
    int32_t* d_out = nullptr;         // finished: the utterance's speech-space ids on the device (f4 hand-off), from the engine's buffer pool
    double t_add = 0, t_admit = 0, t_first = 0, t_finish = 0;     // seconds since the engine was created (t3_get_timing)
};

enum KClass { K_QKV, K_O, K_GU, K_DOWN, K_HEAD, K_ATTN, K_ROPE, K_EMBED, K_SAMPLE, K_COUNT };
static const char* kclass_names[K_COUNT] = {"gemm_qkv", "gemm_o", "gemm_gateup", "gemm_down", "gemm_head",
                                            "attention", "rope_kv", "embed", "sampler"};

}  // namespace

struct T3Engine {
    T3EngineConfig cfg{};
    std::string err;
    hipStream_t stream = nullptr;     // admission copies (cond, sampling params, block table); groups wait on ev_admit
    hipEvent_t ev_admit = nullptr;
    uint64_t admit_seq = 0;           // admissions recorded on ev_admit so far
    bool finalized = false;
    int max_blocks = 0;        // per stream
    int64_t n_blocks = 0;      // pool
    int rmax = 0;              // row budget per step (all groups)
    int n_groups = 1;
    int row_stride = 0;        // int32 words per row record
    bool fuse_rope = true;
    int prefetch_down_lines = 768;    // T3_PREFETCH_DOWN_LINES: 128-byte lines of every down_proj tile (1024) the gate/up launch fetches
    bool zero_copy = true;            // T3_ZERO_COPY=0: the step's metadata / sampled ids travel by hipMemcpyAsync (two copy kernels per step) instead of being read / written in pinned host memory by the step's own kernels
    int prefetch = 1;                 // T3_PREFETCH=0: gate/up's epilogue waves do not fetch down_proj's weights into L2

    // weights (device)
    std::vector<LayerW> layers;
    std::vector<uint16_t> h_norm, h_head;   // host staging until finalize (the final norm weight is folded into the speech head)
    uint16_t *text_emb = nullptr, *speech_emb = nullptr, *text_pos = nullptr, *speech_pos = nullptr, *head = nullptr;
    bool have[6] = {false, false, false, false, false, false};
    float *cos_t = nullptr, *sin_t = nullptr;
    int64_t weight_bytes = 0;

    // KV pool + tables
    uint16_t* kv = nullptr;
    std::vector<int> free_blocks;
    int* d_block_table = nullptr;
    std::vector<int> h_block_table;

    // utterance groups: each group owns a stream, activation buffers, step metadata and captured graphs (measured: kernels of two
    // streams do not overlap usefully on this part, profiles/NOTES.md; one group is the default)
    struct Meta { int* sel_rows; int4* sel; int* rows; int* out_tok; };   // sel arrays first, then the row records (one contiguous upload)
    // Decode steps per graph replay ("burst").  A replay's hand-over costs the GPU ~13 us of idle time against ~1.5 us between two kernels
    // of one graph (profiles/r03_e_step_timeline.json), so while nothing can change the row set -- no request waits for admission, none is in
    // prefill, none reaches its length limit -- BURST_MAX consecutive decode steps are captured and replayed as ONE graph: step j + 1 reads
    // the token step j drew straight from the sampler's device array (EMB_SPEECH_PREV), its metadata from its own pinned buffer of the ring.
    static constexpr int BURST_MAX = 4, NBUF = 2 * BURST_MAX;
    struct Group {
        hipStream_t stream = nullptr;
        uint16_t *h = nullptr, *qkv = nullptr, *qrot = nullptr, *att = nullptr, *act = nullptr, *logits = nullptr;
        float* rstd = nullptr;     // row statistic of the prefill-sized NORM GEMMs
        char *h_meta[NBUF] = {}, *d_meta = nullptr;     // host staging ring: the steps of the burst that runs and of the one built ahead of it
        size_t meta_bytes = 0, meta_rows_off = 0;
        Meta hm[NBUF]{}, dm{};
        int* h_out_tok[NBUF] = {};
        uint64_t waited_admit_seq = 0;     // the admission this group's stream has been ordered behind
        hipEvent_t ev_done[NBUF] = {};
        int rcap = 0;              // row budget per step
        std::map<std::tuple<int, int, int, int>, hipGraphExec_t> graphs;   // (M, n_sel, first staging buffer, steps) -> captured decode step(s)
    };
    // One enqueued step: what the scheduler put on each group's stream, kept until its tokens are back.
    struct StepRec {
        int M = 0, n_sel = 0, n_prefill_rows = 0, decode_rows = 0;
        int max_prefill_ctx = 0;       // longest context among the step's prefill rows (sizes the tile attention's LDS)
        double sum_ctx = 0;
        std::vector<Request*> sampled;
    };
    struct Step {
        std::vector<StepRec> g;
        int buf = 0, M_all = 0, n_prefill_rows = 0, decode_rows = 0, n_sampled = 0;
        double sum_ctx = 0;
        std::chrono::steady_clock::time_point t_begin;
    };
    std::vector<Group> groups;
    unsigned step_seq = 0;
    bool run_ahead = true;     // t3_run_steps / t3_run_until_done enqueue step N+1 before reading step N's tokens
    std::vector<Request*> dec_order;      // scratch of build_step
    int burst = BURST_MAX;     // T3_STEPS_PER_GRAPH: decode steps per graph replay in the run loops (1 = one step per replay)
    bool longest_first = true; // T3_LONGEST_FIRST=0: decode rows in admission order instead of longest context first
    int64_t graph_captures = 0; double graph_capture_ms = 0;      // T3_GRAPH_STATS=1 prints them at destroy
    std::chrono::steady_clock::time_point t_last_complete{};
    std::vector<float> step_ms_ring = std::vector<float>(16384, 0.0f);   // t3_step_times: duration of the most recent steps (as accounted in gpu_ms_total)
    std::vector<int32_t> step_rows_ring = std::vector<int32_t>(16384, 0);
    uint64_t steps_recorded = 0;
    float* d_cond = nullptr;
    uint16_t* d_counts = nullptr;
    T3Sampling* d_sp = nullptr;
    float* d_dbg = nullptr;
    int32_t* d_hist = nullptr;         // [max_seqs][max_model_len]: the sampler appends every token it draws (slot-indexed)
    // f4 hand-off: finished utterances keep their ids in a device buffer of their own (the slot's history is reused by the next
    // occupant) ONLY while the caller has asked for it (t3_reserve_handoff); the pool is grown there, outside the step loop
    bool keep_device_ids = false;
    std::vector<int32_t*> out_pool;    // free per-utterance device id buffers (max_model_len ints each)
    std::vector<void*> out_slabs;      // the allocations the pool was carved from
    HandoffItem* d_handoff_items = nullptr; int handoff_items_cap = 0;      // persistent argument buffer of t3_handoff_tokens
    hipEvent_t ev_handoff = nullptr;
    std::deque<int64_t> finished_q;    // ids finished since the last t3_pop_finished (T3StepResult carries only the first 64 of a step)
    uint16_t* d_dbg_emb = nullptr;     // debug_logits: the embedded input rows of the most recent step (t3_debug_embeddings)
    std::vector<int> dbg_emb_rec;      // their (stream, position) pairs

    // scheduler
    std::unordered_map<int64_t, Request> reqs;
    std::deque<int64_t> waiting;
    std::vector<int64_t> slot_req;     // slot -> req id or -1
    std::vector<int64_t> running;      // admission order

    // stats / profiling
    T3Stats st{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool profile = false;
    int profile_only = -1;     // >= 0: events around this kernel class only
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pev[K_COUNT];
    size_t pev_used[K_COUNT] = {0};
    double k_ms[K_COUNT] = {0};
    int64_t k_n[K_COUNT] = {0};

    std::chrono::steady_clock::time_point t_created = std::chrono::steady_clock::now();
    double now_s() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_created).count(); }
    int fail(int code, const std::string& m) { err = m; return code; }
    // algorithmic weight bytes streamed once per step (SURVEY.md 8(d) "W"): bf16 backbone incl. norms + speech head
    double weight_bytes_for_step() const {
        const double per_layer = (4.0 * D * D + 3.0 * F * D + 2.0 * D) * 2.0;
        return cfg.n_layers * per_layer + D * 2.0 + (double)V * D * 2.0;
    }
};

static int fail_create(int code, const std::string& m) { g_create_error = m; return code; }

// ------------------------------------------------------------------------------------------------
extern "C" const char* t3_last_error(T3Handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" int t3_create(const T3EngineConfig* cfg, T3Handle* out) {
    if (!cfg || !out) return fail_create(T3_E_INVALID, "null argument");
    if (cfg->n_layers <= 0 || cfg->n_layers > 256) return fail_create(T3_E_INVALID, "n_layers out of range");
    if (cfg->max_model_len < T3_COND_ROWS + 2 || cfg->max_model_len > 8192) return fail_create(T3_E_INVALID, "max_model_len out of range [36, 8192]");
    if (cfg->max_seqs <= 0 || cfg->max_seqs > 4096) return fail_create(T3_E_INVALID, "max_seqs out of range");
    if (cfg->text_vocab <= 697) return fail_create(T3_E_INVALID, "text_vocab must cover the placeholder ids 695..697");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail_create(T3_E_DEVICE, "no HIP device: this engine has no CPU fallback");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail_create(T3_E_INVALID, "device_id out of range");
    if (hipSetDevice(cfg->device_id) != hipSuccess) return fail_create(T3_E_DEVICE, "hipSetDevice failed");
    T3Engine* e = new T3Engine();
    e->cfg = *cfg;
    if (e->cfg.cfg_scale != e->cfg.cfg_scale) e->cfg.cfg_scale = 0.5f;
    e->layers.resize(cfg->n_layers);
    e->max_blocks = (cfg->max_model_len + KV_BLOCK - 1) / KV_BLOCK;
    e->row_stride = row_stride_words(e->max_blocks);
    e->rmax = cfg->max_batched_rows > 0 ? cfg->max_batched_rows : std::max(2048, 2 * cfg->max_seqs);
    e->rmax = std::max(e->rmax, 2 * cfg->max_seqs);
    e->slot_req.assign(cfg->max_seqs, -1);
    {
        int g = cfg->n_groups > 0 ? cfg->n_groups : 1;   // measured on MI355X/ROCm 7.2: kernels of different streams do not overlap usefully (2 groups +0 %, 4 groups -60 %)
        if (const char* ev = getenv("T3_GROUPS")) g = atoi(ev);
        e->n_groups = std::max(1, std::min(g, std::min(8, cfg->max_seqs)));
        e->groups.resize(e->n_groups);
        if (const char* ev = getenv("T3_FUSE_ROPE")) e->fuse_rope = atoi(ev) != 0;
        if (const char* ev = getenv("T3_PREFETCH")) e->prefetch = atoi(ev);
        if (const char* ev = getenv("T3_ZERO_COPY")) e->zero_copy = atoi(ev) != 0;
        if (const char* ev = getenv("T3_PREFETCH_DOWN_LINES")) e->prefetch_down_lines = atoi(ev);
        if (const char* ev = getenv("T3_RUN_AHEAD")) e->run_ahead = atoi(ev) != 0;
        if (const char* ev = getenv("T3_STEPS_PER_GRAPH")) e->burst = std::max(1, std::min(atoi(ev), (int)T3Engine::BURST_MAX));
        if (const char* ev = getenv("T3_LONGEST_FIRST")) e->longest_first = atoi(ev) != 0;
    }
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { delete e; return fail_create(T3_E_DEVICE, "hipStreamCreate failed"); }
    hipEventCreate(&e->ev0); hipEventCreate(&e->ev1); hipEventCreateWithFlags(&e->ev_admit, hipEventDisableTiming);
    for (auto& g : e->groups)
        if (hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking) != hipSuccess) { delete e; return fail_create(T3_E_DEVICE, "hipStreamCreate failed"); }
    *out = e;
    return T3_OK;
}


static void free_dev(void* p) { if (p) (void)hipFree(p); }

extern "C" int t3_destroy(T3Handle e) {
    if (!e) return T3_E_INVALID;
    (void)hipSetDevice(e->cfg.device_id);
    (void)hipStreamSynchronize(e->stream);
    if (getenv("T3_GRAPH_STATS")) fprintf(stderr, "[t3] decode-step graphs captured: %lld, %.1f ms in capture + instantiate\n", (long long)e->graph_captures, e->graph_capture_ms);
    for (auto& L : e->layers) { free_dev(L.qkv); free_dev(L.o); free_dev(L.gu); free_dev(L.down); }
    free_dev(e->text_emb); free_dev(e->speech_emb); free_dev(e->text_pos); free_dev(e->speech_pos); free_dev(e->head);
    free_dev(e->cos_t); free_dev(e->sin_t); free_dev(e->kv); free_dev(e->d_block_table);
    for (auto& g : e->groups) {
        if (g.stream) (void)hipStreamSynchronize(g.stream);
        for (auto& kv : g.graphs) (void)hipGraphExecDestroy(kv.second);
        free_dev(g.h); free_dev(g.qkv); free_dev(g.qrot); free_dev(g.att); free_dev(g.act); free_dev(g.logits); free_dev(g.rstd);
        free_dev(g.d_meta); free_dev(g.dm.out_tok);
        for (int b = 0; b < T3Engine::NBUF; ++b) {
            if (g.h_meta[b]) (void)hipHostFree(g.h_meta[b]);
            if (g.h_out_tok[b]) (void)hipHostFree(g.h_out_tok[b]);
            if (g.ev_done[b]) (void)hipEventDestroy(g.ev_done[b]);
        }
        if (g.stream) (void)hipStreamDestroy(g.stream);
    }
    free_dev(e->d_cond); free_dev(e->d_counts); free_dev(e->d_sp); free_dev(e->d_dbg); free_dev(e->d_hist);
    for (void* b : e->out_slabs) free_dev(b);      // every hand-off buffer (pooled or held by a request) lives in one of these
    free_dev(e->d_handoff_items); free_dev(e->d_dbg_emb);
    if (e->ev_handoff) (void)hipEventDestroy(e->ev_handoff);
    if (e->ev_admit) (void)hipEventDestroy(e->ev_admit);
    for (int k = 0; k < K_COUNT; ++k) for (auto& p : e->pev[k]) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return T3_OK;
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
static int to_host(T3Engine* e, const void* src, size_t bytes, std::vector<uint16_t>& dst) {
    dst.resize(bytes / 2);
    if (hipMemcpy(dst.data(), src, bytes, hipMemcpyDefault) != hipSuccess) return e->fail(T3_E_DEVICE, "hipMemcpy (weight) failed");
    return T3_OK;
}
static int upload(T3Engine* e, const uint16_t* host, size_t elems, uint16_t** dev) {
    free_dev(*dev); *dev = nullptr;
    if (hipMalloc((void**)dev, elems * 2) != hipSuccess) return e->fail(T3_E_NOMEM, "hipMalloc (weight) failed");
    if (hipMemcpy(*dev, host, elems * 2, hipMemcpyHostToDevice) != hipSuccess) return e->fail(T3_E_DEVICE, "hipMemcpy H2D (weight) failed");
    e->weight_bytes += (int64_t)elems * 2;
    return T3_OK;
}
static int upload_packed(T3Engine* e, const uint16_t* W, int N, int K, int Npad, uint16_t** dev) {
    std::vector<uint16_t> p((size_t)Npad * K);
    pack_weight(W, N, K, Npad, p.data());
    return upload(e, p.data(), p.size(), dev);
}

extern "C" int t3_load_tensor(T3Handle e, const char* name, const void* data, int32_t rows, int32_t cols) {
    if (!e || !name || !data) return T3_E_INVALID;
    if (e->finalized) return e->fail(T3_E_STATE, "weights already finalized");
    (void)hipSetDevice(e->cfg.device_id);
    const size_t bytes = (size_t)rows * cols * 2;
    std::vector<uint16_t> host;
    auto need = [&](int r, int c) -> bool { return rows == r && cols == c; };
    int L; char rest[128];
    if (sscanf(name, "tfmr.layers.%d.%127s", &L, rest) == 2) {
        if (L < 0 || L >= e->cfg.n_layers) return e->fail(T3_E_NOTFOUND, std::string("layer index beyond n_layers: ") + name);
        LayerW& y = e->layers[L];
        std::string r(rest);
        int rc;
        if (r == "self_attn.q_proj.weight" || r == "self_attn.k_proj.weight" || r == "self_attn.v_proj.weight") {
            if (!need(D, D)) return e->fail(T3_E_INVALID, std::string("bad shape for ") + name);
            auto& dst = r[10] == 'q' ? y.h_q : r[10] == 'k' ? y.h_k : y.h_v;
            return to_host(e, data, bytes, dst);
        }
        if (r == "mlp.gate_proj.weight" || r == "mlp.up_proj.weight") {
            if (!need(F, D)) return e->fail(T3_E_INVALID, std::string("bad shape for ") + name);
            return to_host(e, data, bytes, r[4] == 'g' ? y.h_g : y.h_u);
        }
        if (r == "self_attn.o_proj.weight") {
            if (!need(D, D)) return e->fail(T3_E_INVALID, std::string("bad shape for ") + name);
            if ((rc = to_host(e, data, bytes, host))) return rc;
            y.have_o = true; return upload_packed(e, host.data(), D, D, D, &y.o);
        }
        if (r == "mlp.down_proj.weight") {
            if (!need(D, F)) return e->fail(T3_E_INVALID, std::string("bad shape for ") + name);
            if ((rc = to_host(e, data, bytes, host))) return rc;
            y.have_d = true; return upload_packed(e, host.data(), D, F, D, &y.down);
        }
        if (r == "input_layernorm.weight" || r == "post_attention_layernorm.weight") {
            if ((size_t)rows * cols != (size_t)D) return e->fail(T3_E_INVALID, std::string("bad shape for ") + name);
            return to_host(e, data, bytes, r[0] == 'i' ? y.h_ln1 : y.h_ln2);
        }
        return e->fail(T3_E_NOTFOUND, std::string("unknown tensor ") + name);
    }
    std::string n(name);
    int rc;
    if (n == "tfmr.norm.weight") {
        if ((size_t)rows * cols != (size_t)D) return e->fail(T3_E_INVALID, "bad shape for tfmr.norm.weight");
        e->have[0] = true; return to_host(e, data, bytes, e->h_norm);
    }
    struct { const char* nm; int r, c; uint16_t** dst; int idx; } tabs[] = {
        {"text_emb.weight", e->cfg.text_vocab, D, &e->text_emb, 1},
        {"speech_emb.weight", V, D, &e->speech_emb, 2},
        {"text_pos_emb.emb.weight", 2050, D, &e->text_pos, 3},
        {"speech_pos_emb.emb.weight", 4100, D, &e->speech_pos, 4},
    };
    for (auto& t : tabs)
        if (n == t.nm) {
            if (!need(t.r, t.c)) return e->fail(T3_E_INVALID, std::string("bad shape for ") + name);
            if ((rc = to_host(e, data, bytes, host))) return rc;
            e->have[t.idx] = true; return upload(e, host.data(), host.size(), t.dst);
        }
    if (n == "speech_head.weight") {
        if (!need(V, D)) return e->fail(T3_E_INVALID, "bad shape for speech_head.weight");
        e->have[5] = true; return to_host(e, data, bytes, e->h_head);
    }
    return e->fail(T3_E_NOTFOUND, std::string("unknown tensor ") + name);   // cond_enc.*, text_head.*, tfmr.embed_tokens.* ...
}

template <typename T>
static int dalloc(T3Engine* e, T** p, size_t n, bool zero = false) {
    if (hipMalloc((void**)p, n * sizeof(T)) != hipSuccess) return e->fail(T3_E_NOMEM, "hipMalloc failed (" + std::to_string(n * sizeof(T)) + " bytes)");
    if (zero && hipMemset(*p, 0, n * sizeof(T)) != hipSuccess) return e->fail(T3_E_DEVICE, "hipMemset failed");
    return T3_OK;
}

extern "C" int t3_finalize_weights(T3Handle e) {
    if (!e) return T3_E_INVALID;
    if (e->finalized) return T3_OK;
    (void)hipSetDevice(e->cfg.device_id);
    int rc;
    for (int i = 0; i < 6; ++i) if (!e->have[i]) return e->fail(T3_E_STATE, "missing non-layer tensor #" + std::to_string(i));
    for (int L = 0; L < e->cfg.n_layers; ++L) {
        LayerW& y = e->layers[L];
        if (y.h_q.empty() || y.h_k.empty() || y.h_v.empty() || y.h_g.empty() || y.h_u.empty() || !y.have_o || !y.have_d || y.h_ln1.empty() || y.h_ln2.empty())
            return e->fail(T3_E_STATE, "missing tensors for layer " + std::to_string(L));
        // the RMSNorm weights are folded into the projections that consume the normalised rows (contract: DESIGN.md "RMSNorm")
        std::vector<uint16_t> w((size_t)QKV * D), p((size_t)QKV * D);
        memcpy(w.data(), y.h_q.data(), (size_t)D * D * 2);
        memcpy(w.data() + (size_t)D * D, y.h_k.data(), (size_t)D * D * 2);
        memcpy(w.data() + (size_t)2 * D * D, y.h_v.data(), (size_t)D * D * 2);
        fold_norm_weight(w.data(), QKV, D, y.h_ln1.data(), w.data());
        pack_weight(w.data(), QKV, D, QKV, p.data());
        if ((rc = upload(e, p.data(), p.size(), &y.qkv))) return rc;
        fold_norm_weight(y.h_g.data(), F, D, y.h_ln2.data(), y.h_g.data());
        fold_norm_weight(y.h_u.data(), F, D, y.h_ln2.data(), y.h_u.data());
        std::vector<uint16_t> g((size_t)2 * F * D);
        pack_gate_up(y.h_g.data(), y.h_u.data(), F, D, g.data());
        if ((rc = upload(e, g.data(), g.size(), &y.gu))) return rc;
        for (auto* v : {&y.h_q, &y.h_k, &y.h_v, &y.h_g, &y.h_u, &y.h_ln1, &y.h_ln2}) { v->clear(); v->shrink_to_fit(); }
    }
    {
        fold_norm_weight(e->h_head.data(), V, D, e->h_norm.data(), e->h_head.data());
        if ((rc = upload_packed(e, e->h_head.data(), V, D, HEAD_TILES * 16, &e->head))) return rc;
        e->h_head.clear(); e->h_head.shrink_to_fit(); e->h_norm.clear(); e->h_norm.shrink_to_fit();
    }
    // RoPE tables
    {
        const int mp = e->cfg.max_model_len;
        std::vector<float> c((size_t)mp * 32), s((size_t)mp * 32);
        rope_tables(mp, c.data(), s.data());
        if ((rc = dalloc(e, &e->cos_t, c.size()))) return rc;
        if ((rc = dalloc(e, &e->sin_t, s.size()))) return rc;
        HIP_TRY(hipMemcpy(e->cos_t, c.data(), c.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->sin_t, s.data(), s.size() * 4, hipMemcpyHostToDevice));
    }
    // per-group activations + metadata
    const size_t S = (size_t)e->cfg.max_seqs;
    const size_t Sg = (S + e->n_groups - 1) / e->n_groups;          // utterance slots per group (slot % n_groups)
    for (auto& g : e->groups) {
        g.rcap = std::max((int)(2 * Sg), e->rmax / e->n_groups);
        const size_t R = (size_t)g.rcap;
        if ((rc = dalloc(e, &g.h, R * D, true))) return rc;
        if ((rc = dalloc(e, &g.qkv, R * QKV, true))) return rc;
        if ((rc = dalloc(e, &g.qrot, R * D, true))) return rc;
        if ((rc = dalloc(e, &g.att, R * D, true))) return rc;
        if ((rc = dalloc(e, &g.act, R * F, true))) return rc;
        if ((rc = dalloc(e, &g.logits, 2 * Sg * VPAD, true))) return rc;
        if ((rc = dalloc(e, &g.rstd, R, true))) return rc;
        size_t off = 0;
        auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        const size_t o_selr = carve(2 * Sg * 4), o_sel = carve(Sg * 16), o_rows = carve(R * (size_t)e->row_stride * 4);
        g.meta_rows_off = o_rows;
        g.meta_bytes = off;
        HIP_TRY(hipMalloc((void**)&g.d_meta, g.meta_bytes));
        auto fill = [&](T3Engine::Meta& m, char* base) {
            m.sel_rows = (int*)(base + o_selr); m.sel = (int4*)(base + o_sel); m.rows = (int*)(base + o_rows); m.out_tok = nullptr;
        };
        for (int b = 0; b < T3Engine::NBUF; ++b) {
            HIP_TRY(hipHostMalloc((void**)&g.h_meta[b], g.meta_bytes, hipHostMallocDefault));
            memset(g.h_meta[b], 0, g.meta_bytes);
            fill(g.hm[b], g.h_meta[b]);
            HIP_TRY(hipHostMalloc((void**)&g.h_out_tok[b], Sg * 4, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&g.ev_done[b], hipEventDisableTiming));
        }
        fill(g.dm, g.d_meta);
        int* dtok = nullptr;
        if ((rc = dalloc(e, &dtok, Sg, true))) return rc;
        g.dm.out_tok = dtok;
    }
    if ((rc = dalloc(e, &e->d_cond, S * T3_COND_ROWS * D, true))) return rc;
    if ((rc = dalloc(e, &e->d_counts, S * VPAD, true))) return rc;
    if ((rc = dalloc(e, &e->d_sp, S, true))) return rc;
    if (e->cfg.debug_logits && (rc = dalloc(e, &e->d_dbg, S * V, true))) return rc;
    if (e->cfg.debug_logits && e->n_groups == 1 && (rc = dalloc(e, &e->d_dbg_emb, (size_t)e->groups[0].rcap * D, true))) return rc;
    HIP_TRY(hipEventCreateWithFlags(&e->ev_handoff, hipEventDisableTiming));
    if ((rc = dalloc(e, &e->d_hist, S * (size_t)e->cfg.max_model_len, true))) return rc;
    e->h_block_table.assign(2 * S * e->max_blocks, 0);
    HIP_TRY(t3::prepare_kernels());
    // KV pool
    {
        const size_t per_block = (size_t)e->cfg.n_layers * KV_BLOCK_ELEMS * 2;   // bytes
        int64_t bytes = e->cfg.kv_bytes;
        const int64_t want_all = (int64_t)2 * S * e->max_blocks * per_block;       // every slot at full length
        if (bytes <= 0) {
            size_t fr = 0, tot = 0;
            HIP_TRY(hipMemGetInfo(&fr, &tot));
            const float util = (e->cfg.gpu_memory_utilization > 0 && e->cfg.gpu_memory_utilization <= 1) ? e->cfg.gpu_memory_utilization : 0.9f;
            int64_t budget = (int64_t)((double)tot * util) - (int64_t)(tot - fr);
            budget = std::min<int64_t>(budget, (int64_t)fr - ((int64_t)512 << 20));
            bytes = std::min<int64_t>(budget, want_all);
        }
        e->n_blocks = bytes / (int64_t)per_block;
        if (e->n_blocks < 2 * e->max_blocks) return e->fail(T3_E_NOMEM, "KV pool too small for even one utterance at max_model_len");
        // layout [layer][block][kv][head][tok][64]
        if ((rc = dalloc(e, &e->kv, (size_t)e->n_blocks * e->cfg.n_layers * KV_BLOCK_ELEMS, true))) return rc;
        e->free_blocks.resize(e->n_blocks);
        for (int64_t i = 0; i < e->n_blocks; ++i) e->free_blocks[i] = (int)(e->n_blocks - 1 - i);
    }
    HIP_TRY(hipDeviceSynchronize());
    e->st.kv_blocks_total = e->n_blocks; e->st.kv_blocks_free = e->n_blocks; e->st.weight_bytes = e->weight_bytes;
    e->finalized = true;
    return T3_OK;
}

// ------------------------------------------------------------------------------------------------
// requests
// ------------------------------------------------------------------------------------------------
extern "C" int t3_add_request(T3Handle e, int64_t req_id, const int32_t* ids, int32_t T, const float* cond, const T3Sampling* sp) {
    if (!e || !ids || !cond || !sp) return T3_E_INVALID;
    if (!e->finalized) return e->fail(T3_E_STATE, "finalize_weights first");
    if (e->reqs.count(req_id)) return e->fail(T3_E_INVALID, "duplicate request id");
    if (T < T3_COND_ROWS + 1) return e->fail(T3_E_INVALID, "prompt shorter than the 34 conditioning rows + BOS");
    if (T >= e->cfg.max_model_len) return e->fail(T3_E_INVALID, "prompt (" + std::to_string(T) + " tokens) does not fit max_model_len " + std::to_string(e->cfg.max_model_len));
    if (ids[0] != 695 || ids[T3_COND_ROWS - 1] != 696 || ids[T - 1] != 697) return e->fail(T3_E_INVALID, "prompt is not in the [695, x*32, 696, text..., 697] layout (t3.py:189-200)");
    for (int i = T3_COND_ROWS; i < T - 1; ++i)
        if (ids[i] < 0 || ids[i] >= e->cfg.text_vocab) return e->fail(T3_E_INVALID, "text token id out of range");
    if (T - 1 - T3_COND_ROWS > 2050) return e->fail(T3_E_INVALID, "more text tokens than learned text positions (2050)");
    if (sp->max_tokens <= 0) return e->fail(T3_E_INVALID, "max_tokens must be positive");
    if (!(sp->temperature >= 0.0f) || !(sp->top_p > 0.0f && sp->top_p <= 1.0f) || !(sp->min_p >= 0.0f && sp->min_p <= 1.0f) || !(sp->repetition_penalty > 0.0f))
        return e->fail(T3_E_INVALID, "sampling parameter out of range");
    // a NaN / Inf conditioning row would turn every logit into NaN (the sampler then has no mass to draw from)
    for (size_t i = 0; i < (size_t)T3_COND_ROWS * D; ++i) {
        uint32_t u; memcpy(&u, cond + i, 4);
        if ((u & 0x7f800000u) == 0x7f800000u) return e->fail(T3_E_INVALID, "conditioning embedding contains a non-finite value");
    }
    Request r;
    r.id = req_id; r.prompt.assign(ids, ids + T); r.cond.assign(cond, cond + (size_t)T3_COND_ROWS * D); r.sp = *sp;
    r.limit = std::min(sp->max_tokens, e->cfg.max_model_len - T);
    r.t_add = e->now_s();
    e->reqs.emplace(req_id, std::move(r));
    e->waiting.push_back(req_id);
    return T3_OK;
}

extern "C" int t3_num_unfinished(T3Handle e) { return e ? (int)(e->waiting.size() + e->running.size()) : 0; }

static void release_slot(T3Engine* e, Request& r) {
    for (int s = 0; s < 2; ++s) { for (int b : r.blocks[s]) e->free_blocks.push_back(b); r.blocks[s].clear(); }
    if (r.slot >= 0) e->slot_req[r.slot] = -1;
    r.slot = -1;
    e->st.kv_blocks_free = (int64_t)e->free_blocks.size();
}

static int admit(T3Engine* e) {
    bool table_dirty = false;
    while (!e->waiting.empty()) {
        Request& r = e->reqs[e->waiting.front()];
        const int T = (int)r.prompt.size();
        const int need = (T + r.limit - 1 + KV_BLOCK - 1) / KV_BLOCK;        // positions 0 .. T+limit-2
        int slot = -1;
        for (int s = 0; s < e->cfg.max_seqs; ++s) if (e->slot_req[s] < 0) { slot = s; break; }
        if (slot < 0 || (int64_t)e->free_blocks.size() < 2 * (int64_t)need) break;
        e->waiting.pop_front();
        r.slot = slot; r.state = PREFILL; e->slot_req[slot] = r.id; r.t_admit = e->now_s();
        for (int s = 0; s < 2; ++s)
            for (int b = 0; b < need; ++b) {
                const int blk = e->free_blocks.back(); e->free_blocks.pop_back();
                r.blocks[s].push_back(blk);
                e->h_block_table[(size_t)(2 * slot + s) * e->max_blocks + b] = blk;
            }
        table_dirty = true;
        HIP_TRY(hipMemcpyAsync(e->d_cond + (size_t)slot * T3_COND_ROWS * D, r.cond.data(), (size_t)T3_COND_ROWS * D * 4, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemsetAsync(e->d_counts + (size_t)slot * VPAD, 0, VPAD * 2, e->stream));
        HIP_TRY(hipMemcpyAsync(e->d_sp + slot, &r.sp, sizeof(T3Sampling), hipMemcpyHostToDevice, e->stream));
        // r.cond / r.sp are kept alive in the request map until the copy is consumed (stream-ordered, pageable -> staged synchronously)
        e->running.push_back(r.id);
    }
    // block ids reach the device inside the per-step row records; the admission copies are ordered in front of the next step by an
    // event, recorded (and waited for, enqueue_step) only when something was admitted: a cross-queue wait in front of EVERY step cost
    // ~18 us of idle GPU between two steps (profiles/r03_c_step_timeline.json against r03_d)
    if (table_dirty) { HIP_TRY(hipEventRecord(e->ev_admit, e->stream)); ++e->admit_seq; }
    e->st.kv_blocks_free = (int64_t)e->free_blocks.size();
    return T3_OK;
}

struct Prof {
    T3Engine* e; int k; bool on; hipEvent_t a, b;
    hipStream_t st;
    Prof(T3Engine* e_, int k_, hipStream_t st_) : e(e_), k(k_), on(e_->profile && (e_->profile_only < 0 || e_->profile_only == k_)), st(st_) {
        if (!on) return;
        auto& v = e->pev[k]; size_t& u = e->pev_used[k];
        if (u == v.size()) { hipEvent_t x, y; hipEventCreate(&x); hipEventCreate(&y); v.emplace_back(x, y); }
        a = v[u].first; b = v[u].second; ++u;
        // the launch inside this scope takes the pair as its own start / stop events (the dispatch's begin / end, as rocprofv3 sees it);
        // a launcher that does not (several kernels, the prefill schedules) leaves them armed and gets the bracketing records instead
        hipEventRecord(a, st);
        arm_launch_events(a, b);
    }
    ~Prof() { if (on && launch_events_armed()) { arm_launch_events(nullptr, nullptr); hipEventRecord(b, st); } }
};

// One group's kernel sequence for one step, in phases (eager, or recorded into a hipGraph by the caller): embed | per layer: qkv,
// attention (RoPE / KV write fused for decode rows), o + gate/up + down | head + sampler.
static int launch_embed_phase(T3Engine* e, T3Engine::Group& g, const T3Engine::StepRec& sr, int buf, hipStream_t s) {
    Prof p(e, K_EMBED, s);
    EmbedArgs ea{g.dm.rows, e->row_stride, e->d_cond, e->text_emb, e->text_pos, e->speech_emb, e->speech_pos, g.h, sr.M, g.dm.out_tok};
    if (e->zero_copy) {       // the step's metadata comes straight out of the pinned buffer the scheduler filled (enqueue_step)
        ea.host_meta = reinterpret_cast<const int4*>(g.h_meta[buf]); ea.dev_meta = reinterpret_cast<int4*>(g.d_meta);
        ea.meta_vec = (int)((g.meta_rows_off + (size_t)sr.M * e->row_stride * 4) / 16);
        ea.host_rowrec = g.hm[buf].rows;
    }
    HIP_TRY(launch_embed(ea, s));
    if (e->d_dbg_emb) HIP_TRY(hipMemcpyAsync(e->d_dbg_emb, g.h, (size_t)sr.M * D * 2, hipMemcpyDeviceToDevice, s));
    return T3_OK;
}
static int launch_qkv_phase(T3Engine* e, T3Engine::Group& g, const T3Engine::StepRec& sr, int L, hipStream_t s) {
    LayerW& y = e->layers[L];
    const int M = sr.M;
    // 5 launches per layer: RMSNorm is folded into the qkv / gate-up GEMMs, the residual add into the o / down GEMMs
    Prof p(e, K_QKV, s); GemmArgs a{g.h, (const uint4*)y.qkv, M, D, QKV, g.qkv, QKV, 4, 1, nullptr, 0, g.rstd};
    HIP_TRY(launch_gemm(a, EPI_BF16, choose_mt(M, QKV / 16, 4, true), s));
    return T3_OK;
}
static int launch_attention_phase(T3Engine* e, T3Engine::Group& g, const T3Engine::StepRec& sr, int L, hipStream_t s) {
    const int M = sr.M;
    const size_t layer_elems = (size_t)e->n_blocks * KV_BLOCK_ELEMS;
    const int max_chunks = (e->cfg.max_model_len + CHUNK - 1) / CHUNK;
    uint16_t* kvL = e->kv + (size_t)L * layer_elems;
    const int dec = sr.n_prefill_rows > 0 ? sr.decode_rows : M;      // decode rows come first in the step's row list
    if (e->fuse_rope) {
        // decode rows: every one is the newest position of its stream -> RoPE + KV write inside the attention kernel
        if (dec > 0) {
            Prof p(e, K_ATTN, s);
            AttnArgs aa{nullptr, kvL, g.dm.rows, e->row_stride, g.att, dec, max_chunks, g.qkv, kvL, e->cos_t, e->sin_t};
            HIP_TRY(launch_attention(aa, s));
        }
        // prefill rows (runs of consecutive positions): RoPE + paged KV write, then the 16-rows-per-workgroup attention
        if (M > dec) {
            const int* rows_p = g.dm.rows + (size_t)dec * e->row_stride;
            { Prof p(e, K_ROPE, s); RopeArgs ra{g.qkv + (size_t)dec * QKV, g.qrot + (size_t)dec * D, kvL, rows_p, e->row_stride, e->cos_t, e->sin_t, M - dec}; HIP_TRY(launch_rope_kv(ra, s)); }
            { Prof p(e, K_ATTN, s); AttnArgs aa{g.qrot + (size_t)dec * D, kvL, rows_p, e->row_stride, g.att + (size_t)dec * D, M - dec, max_chunks, nullptr, nullptr, nullptr, nullptr, 0, (sr.max_prefill_ctx + CHUNK - 1) / CHUNK}; HIP_TRY(launch_attention(aa, s)); }
        }
    } else {
        { Prof p(e, K_ROPE, s); RopeArgs ra{g.qkv, g.qrot, kvL, g.dm.rows, e->row_stride, e->cos_t, e->sin_t, M}; HIP_TRY(launch_rope_kv(ra, s)); }
        { Prof p(e, K_ATTN, s); AttnArgs aa{g.qrot, kvL, g.dm.rows, e->row_stride, g.att, M, max_chunks, nullptr, nullptr, nullptr, nullptr, sr.n_prefill_rows > 0 ? sr.decode_rows : -1, (sr.max_prefill_ctx + CHUNK - 1) / CHUNK}; HIP_TRY(launch_attention(aa, s)); }
    }
    return T3_OK;
}
static int launch_mlp_phase(T3Engine* e, T3Engine::Group& g, const T3Engine::StepRec& sr, int L, hipStream_t s) {
    LayerW& y = e->layers[L];
    const int M = sr.M;
    { Prof p(e, K_O, s); GemmArgs a{g.att, (const uint4*)y.o, M, D, D, g.h, D, 16, 0, nullptr}; HIP_TRY(launch_gemm(a, EPI_RESID, choose_mt(M, D / 16, 16, false), s)); }
    {
        Prof p(e, K_GU, s); GemmArgs a{g.h, (const uint4*)y.gu, M, D, F, g.act, F, 4, 1, nullptr, 0, g.rstd};
        // down_proj at <= 80 rows is one workgroup per n-tile (64 tiles of 128 KiB, XCD = tile mod 8): gate/up's epilogue waves fetch the head
        // of every tile into L2 while they fold (nothing else is in flight then).  C3: 20.73 -> 20.91 / 20.98 / 20.96 k tok/s at 512 / 768 / 1024
        // of a tile's 1024 lines (profiles/r03_prefetch_*.json); B = 1: no difference
        if (e->prefetch && M <= 80) a.pf = PrefetchArgs{reinterpret_cast<const unsigned char*>(y.down), D / 16, (F / 32) * 8, 1, e->prefetch_down_lines};
        HIP_TRY(launch_gemm(a, EPI_SILU, choose_mt(M, F / 16, 4, true), s));
    }
    { Prof p(e, K_DOWN, s); GemmArgs a{g.act, (const uint4*)y.down, M, F, D, g.h, D, 16, 0, nullptr}; HIP_TRY(launch_gemm(a, EPI_RESID, choose_mt(M, D / 16, 16, false), s)); }
    return T3_OK;
}
static int launch_sample_phase(T3Engine* e, T3Engine::Group& g, const T3Engine::StepRec& sr, int buf, hipStream_t s) {
    const int n_sel = sr.n_sel;
    if (n_sel <= 0) return T3_OK;
    // final RMSNorm folded into the speech-head GEMM, which gathers the sampled rows itself
    // decode-only steps sample every row, in row order: the gather is the identity and the head reads its rows directly (no dependent index
    // loads in front of the activation loads)
    const int* sel_rows = (sr.n_prefill_rows == 0 && sr.M == 2 * n_sel) ? nullptr : g.dm.sel_rows;
    { Prof p(e, K_HEAD, s); GemmArgs a{g.h, (const uint4*)e->head, 2 * n_sel, D, V, g.logits, VPAD, 4, 1, sel_rows, HEAD_TILES}; HIP_TRY(launch_gemm(a, EPI_BF16, choose_mt(2 * n_sel, VPAD / 16, 4, true), s)); }
    { Prof p(e, K_SAMPLE, s); SampleArgs sa{g.logits, VPAD, g.dm.sel, e->d_counts, e->d_sp, e->cfg.cfg_scale, g.dm.out_tok, e->d_dbg, n_sel, e->d_hist, e->cfg.max_model_len, e->zero_copy ? g.h_out_tok[buf] : nullptr}; HIP_TRY(launch_sampler(sa, s)); }
    return T3_OK;
}
static int launch_step(T3Engine* e, T3Engine::Group& g, const T3Engine::StepRec& sr, int buf, hipStream_t s) {
    int rc;
    if ((rc = launch_embed_phase(e, g, sr, buf, s))) return rc;
    for (int L = 0; L < e->cfg.n_layers; ++L) {
        if ((rc = launch_qkv_phase(e, g, sr, L, s))) return rc;
        if ((rc = launch_attention_phase(e, g, sr, L, s))) return rc;
        if ((rc = launch_mlp_phase(e, g, sr, L, s))) return rc;
    }
    return launch_sample_phase(e, g, sr, buf, s);
}
// Schedule one step into its staging buffer (nothing is launched).  Does not wait for anything: a decode row whose input token is
// still being sampled by the previous step refers to it by its index in the sampler's output array (EMB_SPEECH_PREV).
// may_admit = false: a later step of a burst -- the row set must stay what the burst's first step found.
static int build_step(T3Engine* e, T3Engine::Step& st, bool may_admit) {
    int rc;
    if (may_admit && (rc = admit(e))) return rc;
    st = T3Engine::Step{};
    st.g.resize(e->n_groups);
    st.buf = (int)(e->step_seq++ % T3Engine::NBUF);
    const int buf = st.buf;
    auto add_row = [&](int gi, int stream, int pos, int kind, int a, int b) {
        T3Engine::StepRec& sr = st.g[gi];
        int* rec = e->groups[gi].hm[buf].rows + (size_t)sr.M * e->row_stride;
        rec[0] = stream; rec[1] = pos; rec[2] = kind; rec[3] = a; rec[4] = b;
        memcpy(rec + ROW_HDR, &e->h_block_table[(size_t)stream * e->max_blocks], (size_t)e->max_blocks * 4);
        ++sr.M;
    };
    // decode rows of every running utterance, then prefill rows within the group's budget.  Longest context first: the fused attention's
    // workgroups (one per row and head, dispatched in row order) take a time proportional to the row's context, and a long row handed out
    // last is a tail the rest of the chip waits for.  (Every context grows by one per step: the order only changes when the row set does.)
    std::vector<Request*>& dec = e->dec_order;
    dec.clear();
    for (int64_t id : e->running) {
        Request& r = e->reqs[id];
        if (r.state != DECODE || r.n_sched >= r.limit) continue;       // limit reached: its last token is in flight
        dec.push_back(&r);
    }
    if (e->longest_first)
        std::stable_sort(dec.begin(), dec.end(), [](const Request* x, const Request* y) {
            return (int)x->prompt.size() + x->n_sched > (int)y->prompt.size() + y->n_sched; });
    for (Request* rp : dec) {
        Request& r = *rp;
        const int gi = r.slot % e->n_groups;
        T3Engine::StepRec& sr = st.g[gi];
        auto& hm = e->groups[gi].hm[buf];
        const int T = (int)r.prompt.size(), n = r.n_sched;
        const int pos = T - 1 + n, spos = r.sp.pos_policy == 0 ? (n % 4100) : 0;
        const bool known = (int)r.out.size() == n;
        const int kind = known ? EMB_SPEECH : EMB_SPEECH_PREV, a = known ? r.out.back() : r.last_idx;
        hm.sel_rows[2 * sr.n_sel] = sr.M;     add_row(gi, 2 * r.slot, pos, kind, a, spos);
        hm.sel_rows[2 * sr.n_sel + 1] = sr.M; add_row(gi, 2 * r.slot + 1, pos, kind, a, spos);
        hm.sel[sr.n_sel] = make_int4(r.slot, n, 0, 0);
        sr.sum_ctx += 2.0 * (pos + 1);
        r.last_idx = sr.n_sel; r.n_sched = n + 1;
        sr.sampled.push_back(&r); ++sr.n_sel;
    }
    for (auto& sr : st.g) sr.decode_rows = sr.M;
    for (int64_t id : e->running) {
        Request& r = e->reqs[id];
        if (r.state != PREFILL) continue;
        const int gi = r.slot % e->n_groups;
        T3Engine::StepRec& sr = st.g[gi];
        auto& hm = e->groups[gi].hm[buf];
        const int T = (int)r.prompt.size();
        const int chunk = std::min(T - r.n_prefilled, (e->groups[gi].rcap - sr.M) / 2);
        if (chunk <= 0) continue;
        const int p0 = r.n_prefilled, p1 = p0 + chunk;
        for (int sI = 0; sI < 2; ++sI)
            for (int p = p0; p < p1; ++p) {
                if (p < T3_COND_ROWS) add_row(gi, 2 * r.slot + sI, p, EMB_COND, r.slot, p);
                else if (p < T - 1) { if (sI == 0) add_row(gi, 2 * r.slot, p, EMB_TEXT, r.prompt[p], p - T3_COND_ROWS); else add_row(gi, 2 * r.slot + 1, p, EMB_ZERO, 0, 0); }
                else add_row(gi, 2 * r.slot + sI, p, EMB_SPEECH, 6561, 0);      // BOS: speech_emb[start] + speech_pos[0], t3.py:550-551
                if (p == T - 1) hm.sel_rows[2 * sr.n_sel + sI] = sr.M - 1;
            }
        r.n_prefilled = p1; sr.n_prefill_rows += 2 * chunk; sr.max_prefill_ctx = std::max(sr.max_prefill_ctx, p1);
        if (p1 == T) {
            hm.sel[sr.n_sel] = make_int4(r.slot, 0, 0, 0);
            r.state = DECODE; r.n_sched = 1; r.last_idx = sr.n_sel;
            sr.sampled.push_back(&r); ++sr.n_sel;
        }
    }
    for (auto& sr : st.g) { st.M_all += sr.M; st.n_prefill_rows += sr.n_prefill_rows; st.n_sampled += sr.n_sel; st.decode_rows += sr.decode_rows; st.sum_ctx += sr.sum_ctx; }
    st.t_begin = std::chrono::steady_clock::now();
    return T3_OK;
}

// Can the step after `st` (a decode-only step) be built now and run with exactly st's rows?  Nothing waits in prefill, every decoding
// request has a step left, and nothing changes the row set in between (an admission needs a finish, and a finish needs a token the host
// has not seen: a stop id inside a burst is met when the burst completes -- the utterance's later rows of the burst are dropped, as the
// one row of the run-ahead step always was).
static bool burst_can_continue(T3Engine* e, const T3Engine::Step& st) {
    if (st.n_prefill_rows != 0 || st.M_all == 0) return false;
    int decoding = 0;
    for (int64_t id : e->running) {
        const Request& r = e->reqs[id];
        if (r.state == PREFILL) return false;
        if (r.state != DECODE) continue;
        if (r.n_sched >= r.limit) return false;      // its last token is in flight: the next step has fewer rows
        ++decoding;
    }
    return 2 * decoding == st.M_all;
}

// Put `n` built steps (n > 1: a burst of decode-only steps with one row set) on the streams: eager, one graph replay per step, or ONE
// graph replay for the burst.  The completion event is recorded behind the last step only (its ev_done slot).
static int launch_steps(T3Engine* e, T3Engine::Step* steps, int n) {
    T3Engine::Step& st = steps[0];
    if (st.M_all == 0) return T3_OK;
    const int buf = st.buf, buf_last = steps[n - 1].buf;
    const bool graphs_ok = !e->cfg.enforce_eager && !e->profile && st.n_prefill_rows == 0;
    for (int gi = 0; gi < e->n_groups; ++gi) {
        T3Engine::Group& g = e->groups[gi];
        const T3Engine::StepRec& sr = st.g[gi];
        if (sr.M == 0) continue;
        hipStream_t s = g.stream;
        if (g.waited_admit_seq != e->admit_seq) { HIP_TRY(hipStreamWaitEvent(s, e->ev_admit, 0)); g.waited_admit_seq = e->admit_seq; }
        if (!e->zero_copy) HIP_TRY(hipMemcpyAsync(g.d_meta, g.h_meta[buf], g.meta_rows_off + (size_t)sr.M * e->row_stride * 4, hipMemcpyHostToDevice, s));   // sel arrays + the used row records
        if (e->d_dbg_emb) {
            e->dbg_emb_rec.resize((size_t)2 * sr.M);
            for (int r = 0; r < sr.M; ++r) { const int* rec = g.hm[buf].rows + (size_t)r * e->row_stride; e->dbg_emb_rec[2 * r] = rec[0]; e->dbg_emb_rec[2 * r + 1] = rec[1]; }
        }
        if (graphs_ok) {
            const auto key = std::make_tuple(sr.M, sr.n_sel, e->zero_copy ? buf : 0, n);      // zero-copy: the pinned buffers of `buf` ... are kernel arguments
            auto it = g.graphs.find(key);
            if (it == g.graphs.end()) {
                hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
                const auto tc0 = std::chrono::steady_clock::now();
                HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                int lrc = T3_OK;
                for (int j = 0; j < n && !lrc; ++j) lrc = launch_step(e, g, steps[j].g[gi], steps[j].buf, s);
                const hipError_t ce = hipStreamEndCapture(s, &graph);
                if (lrc) return lrc;
                HIP_TRY(ce);
                HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                (void)hipGraphDestroy(graph);
                if (g.graphs.size() > 192) {         // a replay of one of them may still be running (run-ahead): drain first
                    HIP_TRY(hipStreamSynchronize(s));
                    for (auto& kv : g.graphs) (void)hipGraphExecDestroy(kv.second);
                    g.graphs.clear();
                }
                it = g.graphs.emplace(key, exec).first;
                ++e->graph_captures; e->graph_capture_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count();
            }
            HIP_TRY(hipGraphLaunch(it->second, s));
        } else {
            for (int j = 0; j < n; ++j) { const int lrc = launch_step(e, g, steps[j].g[gi], steps[j].buf, s); if (lrc) return lrc; }
        }
        if (sr.n_sel > 0 && !e->zero_copy) HIP_TRY(hipMemcpyAsync(g.h_out_tok[buf], g.dm.out_tok, (size_t)sr.n_sel * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(g.ev_done[buf_last], s));
    }
    return T3_OK;
}
static bool bursts_enabled(const T3Engine* e) { return e->burst > 1 && e->zero_copy && !e->cfg.enforce_eager && !e->profile && !e->cfg.debug_logits; }
// Every item (a step or a burst) starts on a window boundary of the staging ring: its graph is keyed by its first buffer, so windows keep
// the number of captured variants per shape at NBUF / burst instead of NBUF, and two items in flight never share a buffer.
static void begin_item(T3Engine* e) {
    const unsigned B = bursts_enabled(e) ? (unsigned)e->burst : 1u;
    if (e->step_seq % B) e->step_seq += B - e->step_seq % B;
}
// one step: schedule + launch (t3_step)
static int enqueue_step(T3Engine* e, T3Engine::Step& st) {
    int rc;
    begin_item(e);
    if ((rc = build_step(e, st, true))) return rc;
    return launch_steps(e, &st, 1);
}

// n more per-utterance device id buffers for the hand-off pool, carved from one allocation
static int grow_out_pool(T3Engine* e, int n) {
    const size_t per = (size_t)e->cfg.max_model_len;
    int32_t* slab = nullptr;
    const hipError_t me = hipMalloc((void**)&slab, (size_t)n * per * 4);
    if (me != hipSuccess) return e->fail(T3_E_NOMEM, "hipMalloc of " + std::to_string((size_t)n * per * 4) + " bytes for " + std::to_string(n) +
                                                     " hand-off id buffers failed: " + hipGetErrorString(me));
    e->out_slabs.push_back(slab);
    for (int i = 0; i < n; ++i) e->out_pool.push_back(slab + (size_t)i * per);
    return T3_OK;
}

// Wait for an enqueued step, account for it and hand its tokens to the requests.
// burst_ms < 0: wait for the step's own completion event and clock it.  burst_ms >= 0: the step belongs to a burst whose event (recorded
// behind its last step) the caller has waited for; burst_ms = the burst's time / its steps (the steps of one replay cannot be clocked apart).
static int complete_step(T3Engine* e, T3Engine::Step& st, T3StepResult* res, double burst_ms = -1.0) {
    res->n_rows = st.M_all; res->n_prefill_rows = st.n_prefill_rows; res->n_sampled = st.n_sampled;
    if (st.M_all == 0) { res->n_running = (int)e->running.size(); res->n_waiting = (int)e->waiting.size(); return T3_OK; }
    double ms = burst_ms;
    if (burst_ms < 0) {
        for (int gi = 0; gi < e->n_groups; ++gi) if (st.g[gi].M) HIP_TRY(hipEventSynchronize(e->groups[gi].ev_done[st.buf]));
        const auto now = std::chrono::steady_clock::now();
        // with a step running ahead, this step had the GPU to itself only since the previous one completed
        const auto t0 = std::max(st.t_begin, e->t_last_complete);
        e->t_last_complete = now;
        ms = std::chrono::duration<double, std::milli>(now - t0).count();
    }
    const bool decode_only = (st.n_prefill_rows == 0);
    e->st.steps++; e->st.gpu_ms_total += ms; e->st.prefill_rows += st.n_prefill_rows; e->st.decode_rows += st.decode_rows;
    e->step_ms_ring[e->steps_recorded % e->step_ms_ring.size()] = (float)ms; e->step_rows_ring[e->steps_recorded % e->step_ms_ring.size()] = st.n_prefill_rows > 0 ? -st.M_all : st.M_all; ++e->steps_recorded;
    if (decode_only) {
        e->st.decode_steps++; e->st.gpu_ms_decode += ms; e->st.sum_ctx_decode += st.sum_ctx;
        // W + KV read + KV write + embedding rows, scaled to n_layers
        const double per_tok_stream = 2.0 * e->cfg.n_layers * H * HD * 2;
        e->st.algo_bytes_decode += (double)e->weight_bytes_for_step() + per_tok_stream * st.sum_ctx + per_tok_stream * st.decode_rows + (st.decode_rows / 2) * 4096.0;
    }
    if (e->profile) {
        for (int k = 0; k < K_COUNT; ++k) {
            if (decode_only)
                for (size_t i = 0; i < e->pev_used[k]; ++i) { float t = 0; (void)hipEventElapsedTime(&t, e->pev[k][i].first, e->pev[k][i].second); e->k_ms[k] += t; e->k_n[k]++; }
            e->pev_used[k] = 0;
        }
    }
    // ---- host bookkeeping (a device error met here is reported after the bookkeeping is complete: the scheduler state stays consistent)
    int dev_rc = T3_OK;
    for (int gi = 0; gi < e->n_groups; ++gi) {
        const T3Engine::StepRec& sr = st.g[gi];
        const int* toks = e->groups[gi].h_out_tok[st.buf];
        for (int i = 0; i < sr.n_sel; ++i) {
            Request& r = *sr.sampled[i];
            ++r.n_seen;
            if (r.state == FINISHED) {       // row scheduled before its stop token was seen: drop the result; the slot is free once its last row in flight is back
                if (r.zombie && r.n_seen >= r.n_sched) { r.zombie = false; release_slot(e, r); }
                continue;
            }
            const int tok = toks[i];
            if (r.out.empty()) r.t_first = e->now_s();
            r.out.push_back(tok); e->st.tokens_generated++;
            int fin = 0;
            if (!r.sp.ignore_eos && tok == r.sp.stop_token) fin = 1;
            else if ((int)r.out.size() >= r.limit) fin = 2;
            if (fin) {
                r.state = FINISHED; r.finish_reason = fin; r.t_finish = e->now_s();
                if (e->keep_device_ids) {
                    // the ids stay on the device for the hand-off: out of the slot's history (the slot will be reused) into a buffer of the request's own.
                    // Stream order does the rest: the copy runs behind the step that drew the last token and ahead of the slot's next occupant.
                    if (e->out_pool.empty()) {       // more finished-and-unreleased requests than t3_reserve_handoff was told about
                        const int grc = grow_out_pool(e, 16);
                        if (grc && !dev_rc) { dev_rc = grc; }
                    }
                    if (!e->out_pool.empty()) {
                        r.d_out = e->out_pool.back(); e->out_pool.pop_back();
                        const hipError_t ce = hipMemcpyAsync(r.d_out, e->d_hist + (size_t)r.slot * e->cfg.max_model_len, r.out.size() * 4, hipMemcpyDeviceToDevice, e->groups[gi].stream);
                        if (ce != hipSuccess && !dev_rc) dev_rc = e->fail(T3_E_DEVICE, std::string("hand-off copy of a finished utterance's ids failed: ") + hipGetErrorString(ce));
                    }
                }
                if (res->n_finished < 64) res->finished_ids[res->n_finished] = r.id;
                res->n_finished++;
                e->finished_q.push_back(r.id);
                if (e->finished_q.size() > (size_t)(4 * e->cfg.max_seqs + 4096)) { e->finished_q.pop_front(); e->st.finished_dropped++; }     // a caller that never pops (LLM.generate reads outputs by id) must not grow it
                if (r.n_sched > (int)r.out.size()) r.zombie = true;     // the step running ahead still uses its slot and KV blocks
                else release_slot(e, r);
            }
        }
    }
    if (res->n_finished) e->running.erase(std::remove_if(e->running.begin(), e->running.end(), [&](int64_t id) { return e->reqs[id].state == FINISHED; }), e->running.end());
    res->n_running = (int)e->running.size(); res->n_waiting = (int)e->waiting.size();
    return dev_rc;
}

static int check_ready(T3Engine* e) {
    if (!e) return T3_E_INVALID;
    if (!e->finalized) return e->fail(T3_E_STATE, "finalize_weights first");
    (void)hipSetDevice(e->cfg.device_id);
    return T3_OK;
}

extern "C" int t3_step(T3Handle e, T3StepResult* res) {
    int rc;
    if ((rc = check_ready(e))) return rc;
    T3StepResult local{}; if (!res) res = &local;
    memset(res, 0, sizeof(*res));
    T3Engine::Step st;
    if ((rc = enqueue_step(e, st))) return rc;
    return complete_step(e, st, res);
}

// The step loop of t3_run_steps / t3_run_until_done.  With run-ahead, the next item (a step, or a burst of decode steps replayed as one
// graph) is scheduled and enqueued before the host waits for the current one, so the device never idles across the token read-back
// and the scheduler.  An utterance whose stop token turns up in step N has wasted row pairs in the steps already enqueued behind N (one
// with single steps, up to 2 * burst - 1 with bursts); its token stream is unaffected.
namespace {
struct Item { T3Engine::Step steps[T3Engine::BURST_MAX]; int n = 0; };
}
// schedule + launch the next item: up to `room` steps (>= 1); a burst only where the row set provably stays the same
static int enqueue_item(T3Engine* e, Item& it, int64_t room) {
    int rc;
    it.n = 0;
    const bool burst_ok = bursts_enabled(e);
    begin_item(e);
    if ((rc = build_step(e, it.steps[0], true))) return rc;
    it.n = 1;
    if (it.steps[0].M_all == 0) return T3_OK;
    while (burst_ok && it.n < e->burst && it.n < room && burst_can_continue(e, it.steps[it.n - 1])) {
        if ((rc = build_step(e, it.steps[it.n], false))) return rc;
        ++it.n;
    }
    return launch_steps(e, it.steps, it.n);
}
static int complete_item(T3Engine* e, Item& it, T3StepResult* r) {
    if (it.n == 1) { memset(r, 0, sizeof(*r)); return complete_step(e, it.steps[0], r); }
    T3Engine::Step& last = it.steps[it.n - 1];
    for (int gi = 0; gi < e->n_groups; ++gi) if (last.g[gi].M) HIP_TRY(hipEventSynchronize(e->groups[gi].ev_done[last.buf]));
    const auto now = std::chrono::steady_clock::now();
    const auto t0 = std::max(it.steps[0].t_begin, e->t_last_complete);
    e->t_last_complete = now;
    const double ms = std::chrono::duration<double, std::milli>(now - t0).count() / it.n;
    int rc = T3_OK;
    for (int j = 0; j < it.n; ++j) { memset(r, 0, sizeof(*r)); const int rc2 = complete_step(e, it.steps[j], r, ms); if (!rc) rc = rc2; }
    return rc;
}
static int run_loop(T3Engine* e, int64_t n, int32_t* done, bool stall_is_error) {
    int rc;
    if ((rc = check_ready(e))) return rc;
    const bool ahead_ok = e->run_ahead && !e->profile && !e->cfg.debug_logits;
    static thread_local Item items[2];
    Item *cur = &items[0], *nxt = &items[1];
    T3StepResult r;
    bool have = false;
    int64_t k = 0;            // steps completed
    rc = T3_OK;
    while (k < n && (have || t3_num_unfinished(e) > 0)) {
        if (!have) {
            if ((rc = enqueue_item(e, *cur, n - k))) break;
            if (cur->steps[0].M_all == 0) { if (stall_is_error) rc = e->fail(T3_E_NOMEM, "scheduler stalled: waiting requests cannot be admitted"); break; }
            have = true;
        }
        bool ahead = false;
        if (ahead_ok && k + cur->n < n) {
            if ((rc = enqueue_item(e, *nxt, n - k - cur->n))) { (void)complete_item(e, *cur, &r); k += cur->n; have = false; break; }
            ahead = nxt->steps[0].M_all > 0;
        }
        rc = complete_item(e, *cur, &r);
        k += cur->n; have = false;
        if (ahead) { std::swap(cur, nxt); have = true; }
        if (rc) break;
    }
    if (have) { const int rc2 = complete_item(e, *cur, &r); k += cur->n; if (!rc) rc = rc2; }
    if (done) *done = (int32_t)k;
    return rc;
}

extern "C" int t3_run_until_done(T3Handle e) { return run_loop(e, INT64_MAX, nullptr, true); }

extern "C" int t3_run_steps(T3Handle e, int32_t n, int32_t* done) { return run_loop(e, n, done, false); }

extern "C" int t3_get_output(T3Handle e, int64_t req_id, int32_t* ids, int32_t* n, int32_t* finish_reason) {
    if (!e || !n) return T3_E_INVALID;
    auto it = e->reqs.find(req_id);
    if (it == e->reqs.end()) return e->fail(T3_E_NOTFOUND, "unknown request id");
    const Request& r = it->second;
    const int have = (int)r.out.size();
    if (ids) { const int m = std::min(have, *n); for (int i = 0; i < m; ++i) ids[i] = r.out[i] + T3_SPEECH_TOKEN_OFFSET; }
    *n = have;
    if (finish_reason) *finish_reason = r.finish_reason;
    return T3_OK;
}

extern "C" int t3_get_timing(T3Handle e, int64_t req_id, double* out4) {
    if (!e || !out4) return T3_E_INVALID;
    auto it = e->reqs.find(req_id);
    if (it == e->reqs.end()) return e->fail(T3_E_NOTFOUND, "unknown request id");
    const Request& r = it->second;
    out4[0] = r.t_add; out4[1] = r.t_admit; out4[2] = r.t_first; out4[3] = r.t_finish;
    return T3_OK;
}

extern "C" int t3_release_request(T3Handle e, int64_t req_id) {
    if (!e) return T3_E_INVALID;
    auto it = e->reqs.find(req_id);
    if (it == e->reqs.end()) return e->fail(T3_E_NOTFOUND, "unknown request id");
    if (it->second.state != FINISHED) return e->fail(T3_E_STATE, "request still running");
    if (it->second.d_out) e->out_pool.push_back(it->second.d_out);
    e->reqs.erase(it);
    return T3_OK;
}

// f4: batched hand-off of finished utterances to the vocoder (replaces the per-utterance loop of tts.py:483-514)
extern "C" int t3_handoff_tokens(T3Handle e, const int64_t* req_ids, int32_t n, const int32_t* text_token_counts, int32_t flags,
                                 int32_t* dev_tokens, int32_t ld, int32_t* dev_lens) {
    if (!e || !req_ids || !text_token_counts || !dev_tokens || !dev_lens || n <= 0 || ld <= 0) return T3_E_INVALID;
    (void)hipSetDevice(e->cfg.device_id);
    std::vector<HandoffItem> items(n);
    for (int i = 0; i < n; ++i) {
        auto it = e->reqs.find(req_ids[i]);
        if (it == e->reqs.end()) return e->fail(T3_E_NOTFOUND, "unknown request id");
        const Request& r = it->second;
        if (r.state != FINISHED) return e->fail(T3_E_STATE, "request " + std::to_string(req_ids[i]) + " has not finished");
        if (!r.d_out && !r.out.empty()) return e->fail(T3_E_STATE, "the device ids of request " + std::to_string(req_ids[i]) + " were not kept: call t3_reserve_handoff before the run");
        items[i] = HandoffItem{r.d_out, (int)r.out.size(), text_token_counts[i], 0};
    }
    // the per-request copies were queued on the group streams: this launch waits for them on the device, not on the host
    for (auto& g : e->groups) { HIP_TRY(hipEventRecord(e->ev_handoff, g.stream)); HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_handoff, 0)); }
    if (n > e->handoff_items_cap) {
        HIP_TRY(hipStreamSynchronize(e->stream));          // a previous call's launch may still read the old buffer
        free_dev(e->d_handoff_items); e->d_handoff_items = nullptr; e->handoff_items_cap = 0;
        const int cap = std::max(n, 2 * e->cfg.max_seqs);
        if (hipMalloc((void**)&e->d_handoff_items, (size_t)cap * sizeof(HandoffItem)) != hipSuccess) return e->fail(T3_E_NOMEM, "hipMalloc (hand-off argument buffer) failed");
        e->handoff_items_cap = cap;
    }
    HIP_TRY(hipMemcpyAsync(e->d_handoff_items, items.data(), items.size() * sizeof(HandoffItem), hipMemcpyHostToDevice, e->stream));   // pageable source: staged before the call returns
    HIP_TRY(launch_handoff(e->d_handoff_items, n, flags, dev_tokens, ld, dev_lens, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return T3_OK;
}

extern "C" int t3_reserve_handoff(T3Handle e, int32_t n_requests) {
    if (!e || n_requests < 0) return T3_E_INVALID;
    if (!e->finalized) return e->fail(T3_E_STATE, "finalize_weights first");
    (void)hipSetDevice(e->cfg.device_id);
    e->keep_device_ids = n_requests > 0;
    const int missing = n_requests - (int)e->out_pool.size();
    return missing > 0 ? grow_out_pool(e, missing) : T3_OK;
}

extern "C" int t3_pop_finished(T3Handle e, int64_t* ids, int32_t cap) {
    if (!e || (!ids && cap > 0) || cap < 0) return T3_E_INVALID;
    int n = 0;
    while (n < cap && !e->finished_q.empty()) { ids[n++] = e->finished_q.front(); e->finished_q.pop_front(); }
    return n;
}

extern "C" int t3_debug_embeddings(T3Handle e, void* out_bf16, int32_t* row_stream, int32_t* row_pos, int32_t* n) {
    if (!e || !n) return T3_E_INVALID;
    if (!e->d_dbg_emb) return e->fail(T3_E_STATE, "engine was created without debug_logits (or with several utterance groups)");
    const int rows = (int)(e->dbg_emb_rec.size() / 2), m = std::min(rows, *n);
    (void)hipSetDevice(e->cfg.device_id);
    if (out_bf16 && m > 0) HIP_TRY(hipMemcpy(out_bf16, e->d_dbg_emb, (size_t)m * D * 2, hipMemcpyDeviceToHost));
    for (int r = 0; r < m; ++r) { if (row_stream) row_stream[r] = e->dbg_emb_rec[2 * r]; if (row_pos) row_pos[r] = e->dbg_emb_rec[2 * r + 1]; }
    *n = rows;
    return T3_OK;
}

extern "C" int t3_abort_request(T3Handle e, int64_t req_id) {
    if (!e) return T3_E_INVALID;
    auto it = e->reqs.find(req_id);
    if (it == e->reqs.end()) return e->fail(T3_E_NOTFOUND, "unknown request id");
    Request& r = it->second;
    // a row of a finished request can only be on a stream inside a run loop, and those never return with work in flight
    if (r.zombie) return e->fail(T3_E_STATE, "request has a step in flight");
    if (r.state == WAITING) {
        e->waiting.erase(std::remove(e->waiting.begin(), e->waiting.end(), req_id), e->waiting.end());
    } else if (r.state != FINISHED) {
        (void)hipSetDevice(e->cfg.device_id);
        for (auto& g : e->groups) if (g.stream) (void)hipStreamSynchronize(g.stream);      // nothing of it may still be running
        e->running.erase(std::remove(e->running.begin(), e->running.end(), req_id), e->running.end());
        release_slot(e, r);
    }
    if (r.d_out) e->out_pool.push_back(r.d_out);
    e->reqs.erase(it);
    return T3_OK;
}

extern "C" int t3_debug_logits(T3Handle e, int64_t req_id, float* out) {
    if (!e || !out) return T3_E_INVALID;
    if (!e->d_dbg) return e->fail(T3_E_STATE, "engine was created without debug_logits");
    auto it = e->reqs.find(req_id);
    if (it == e->reqs.end()) return e->fail(T3_E_NOTFOUND, "unknown request id");
    int slot = it->second.slot;
    if (slot < 0) return e->fail(T3_E_STATE, "request holds no slot (finished): read logits before the finishing step or use max_tokens+1");
    HIP_TRY(hipMemcpy(out, e->d_dbg + (size_t)slot * V, (size_t)V * 4, hipMemcpyDeviceToHost));
    return T3_OK;
}

extern "C" int t3_step_times(T3Handle e, float* ms, int32_t* rows, int32_t cap) {
    if (!e || cap < 0 || (cap > 0 && !ms)) return T3_E_INVALID;
    const uint64_t have = std::min<uint64_t>(e->steps_recorded, e->step_ms_ring.size());
    const int n = (int)std::min<uint64_t>(have, (uint64_t)cap);
    for (int i = 0; i < n; ++i) {
        const uint64_t k = (e->steps_recorded - n + i) % e->step_ms_ring.size();
        ms[i] = e->step_ms_ring[k]; if (rows) rows[i] = e->step_rows_ring[k];
    }
    return n;
}
extern "C" int t3_stats(T3Handle e, T3Stats* out) { if (!e || !out) return T3_E_INVALID; *out = e->st; return T3_OK; }
extern "C" int t3_reset_stats(T3Handle e) {
    if (!e) return T3_E_INVALID;
    const int64_t bt = e->st.kv_blocks_total, bf = e->st.kv_blocks_free, wb = e->st.weight_bytes, fd = e->st.finished_dropped;
    e->steps_recorded = 0;
    e->st = T3Stats{}; e->st.kv_blocks_total = bt; e->st.kv_blocks_free = bf; e->st.weight_bytes = wb; e->st.finished_dropped = fd;
    for (int k = 0; k < K_COUNT; ++k) { e->k_ms[k] = 0; e->k_n[k] = 0; }
    return T3_OK;
}
extern "C" int t3_set_profile(T3Handle e, int32_t on) { if (!e) return T3_E_INVALID; e->profile = on != 0; return T3_OK; }
extern "C" int t3_set_profile_kernel(T3Handle e, const char* name) {
    if (!e) return T3_E_INVALID;
    if (!name || !*name) { e->profile_only = -1; return T3_OK; }
    for (int k = 0; k < K_COUNT; ++k) if (!strcmp(name, kclass_names[k])) { e->profile_only = k; return T3_OK; }
    return e->fail(T3_E_NOTFOUND, "unknown kernel class");
}
extern "C" int t3_kernel_ms(T3Handle e, const char* name, double* avg_ms, int64_t* launches) {
    if (!e || !name) return T3_E_INVALID;
    for (int k = 0; k < K_COUNT; ++k)
        if (!strcmp(name, kclass_names[k])) {
            if (avg_ms) *avg_ms = e->k_n[k] ? e->k_ms[k] / (double)e->k_n[k] : 0.0;
            if (launches) *launches = e->k_n[k];
            return T3_OK;
        }
    return e->fail(T3_E_NOTFOUND, "unknown kernel class");
}

// ------------------------------------------------------------------------------------------------
// Token post-filter: restatement of tts.py:300-365 + alignment_stream_analyzer.py:111-201 (token-heuristic version).
// ------------------------------------------------------------------------------------------------
extern "C" int t3_clean_tokens(const int32_t* ids, int32_t n, int32_t text_token_count, int32_t flags, int32_t* out, int32_t* reason) {
    if ((!ids && n > 0) || !out || n < 0) return T3_E_INVALID;
    int kept = 0, why = 0;
    bool complete = false;
    int completed_at = 0;
    for (int i = 0; i < n; ++i) {
        const int frame = i + 1;                                             // curr_frame_pos after the increment (:140)
        int est = frame / 2; if (est > text_token_count - 1) est = text_token_count - 1;      // :144
        if (!complete && est >= text_token_count - 3) { complete = true; completed_at = frame; }   // :148-151
        const bool repetition = i >= 2 && ids[i] == ids[i - 1] && ids[i] == ids[i - 2];     // :203-213 (window of 8 >= 3)
        const bool long_tail = complete && (frame - completed_at) >= 10;                    // :157-160
        if (long_tail || repetition) { why = repetition ? 1 : 2; break; }                   // forced EOS logit 2^15 > 2^14 (tts.py:341-345)
        out[kept++] = ids[i];
    }
    if (flags & 1) {                                                          // tts.py:514
        int w = 0;
        for (int i = 0; i < kept; ++i) if (out[i] >= 0 && out[i] < 6561) out[w++] = out[i];
        kept = w;
    }
    if (reason) *reason = why;
    return kept;
}
