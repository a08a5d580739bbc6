// Internal launcher interface between engine.cpp / kernel_abi.cpp and the kernel translation units t3_gemm.hip, t3_attention.hip,
// t3_kernels.hip (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/t3_engine.h"

namespace t3 {

constexpr int D = 1024, H = 16, HD = 64, F = 4096, V = 8194, VPAD = 8208, QKV = 3072;
constexpr int HEAD_TILES = 516;                   // speech head packed to a multiple of 4 n-tiles (513 hold rows)
constexpr int CHUNK = 64;                          // attention chunk in tokens (numerics contract)
#ifndef T3_KV_BLOCK
#define T3_KV_BLOCK 256
#endif
constexpr int KV_BLOCK = T3_KV_BLOCK;              // tokens per physical KV block (a multiple of CHUNK): per head 32 KiB K + 32 KiB V contiguous
#ifndef T3_KV_PAD
#define T3_KV_PAD 512
#endif
constexpr int KV_HEAD_PAD = T3_KV_PAD;               // elements of padding after every (kv, head) region: breaks the power-of-two
                                                   // strides (32 KiB per head, 512 KiB K->V) that make concurrent workgroups hit the same HBM channels
constexpr int KV_HEAD_ELEMS = KV_BLOCK * HD + KV_HEAD_PAD;
constexpr int KV_BLOCK_ELEMS = 2 * H * KV_HEAD_ELEMS;  // per layer per block: [kv][head][chunk][8 fragments][64 lanes][8]

enum GemmEpi { EPI_F32 = 0, EPI_BF16 = 1, EPI_RESID = 2, EPI_SILU = 3 };

// Weights of a LATER launch on the stream, pulled into the L2 of the XCD that will read them while this launch leaves the memory system
// idle: the fold + epilogue of a 4-wave GEMM form (0.4-1.3 us with nothing in flight), or behind the first group's barrier of the pipelined
// gate/up form (gemm2_pipe16_kernel: every weight tile has landed, only rows move).  The consumer is a gemm2_kernel whose workgroup bx
// reads the `group` packed n-tiles bx * group ... (tile_lines 128-byte lines each, contiguous); block b of EVERY launch lands on XCD
// (x0 + b) % 8 with the same x0 (tools/xcd_probe.hip: eager and graph launches, 1-D and 2-D grids of any size), so the lines of tile group g
// are requested by workgroups of this launch whose linear id is g mod 8, one line per lane and instruction, by LDS-DMA into a 256-byte
// corner of LDS nobody reads (a load into a register would be written whenever it lands, long after the compiler has given the register
// to something else: that build faulted).  Speed only: nothing depends on where the lines end up.
struct PrefetchArgs {
    const unsigned char* base = nullptr;   // null: nothing to prefetch
    int n_tiles = 0, tile_lines = 0, group = 1;
    int max_lines = 0;                     // > 0: only the first max_lines lines of every tile group
};

struct GemmArgs {
    const uint16_t* X;   // [M][K] bf16 row-major (NORM form: the un-normalised residual stream)
    const uint4* Wp;     // packed weight, see pack_weight() (NORM form: with the norm weight folded in, fold_norm_weight())
    int M, K, N;         // N = number of valid output columns (EPI_SILU: F)
    void* out;           // EPI_F32: float [M][ldo]; else bf16 [M][ldo]; EPI_RESID: residual stream, updated in place
    int ldo;
    int nw;              // waves = K segments per workgroup: 4 (qkv, gate/up, head) or 16 (o_proj, down_proj)
    int norm;                // 1: NORM form -- scale every row by rstd = 1/sqrt(mean(x^2) + eps) in the epilogue (K = 1024, nw = 4)
    const int* row_index;    // optional gather: source row of X per GEMM row (speech head over the sampled rows)
    int packed_tiles = 0;    // > 0: the packed weight holds this many n-tiles (zero rows beyond N), so tile groups may overhang N
    float* rstd_scratch = nullptr;   // [M] floats: lets the NORM forms take the prefill-sized schedule (row statistic in its own pass)
    PrefetchArgs pf;                 // gemm2_kernel, 4-wave forms: requested by the epilogue waves once the workgroup's own operands have landed
    int gx_real = 0;                 // set by the launcher: > 0 = the grid's x extent was padded (to a multiple of 8), workgroups with blockIdx.x >= gx_real leave at once
};

// Host-side packing of a [N][K] bf16 matrix into MFMA-B-operand order:
// out[((nt*KB + kb)*64 + lane)*8 + j] = W[nt*16 + (lane&15)][kb*32 + 8*(lane>>4) + j], rows >= N are zero.
void pack_weight(const uint16_t* W, int N, int K, int Npad, uint16_t* out);
// gate/up interleave: packed tile 2t = gate tile t, 2t+1 = up tile t.
void pack_gate_up(const uint16_t* Wg, const uint16_t* Wu, int Fdim, int K, uint16_t* out);
// W'[n][k] = bf16(W[n][k] * ln[k]) (host): an RMSNorm weight folded into the projection behind it, before packing
void fold_norm_weight(const uint16_t* W, int N, int K, const uint16_t* ln, uint16_t* out);

int choose_mt(int M, int ntiles_x, int nw = 4, bool norm = false);
hipError_t launch_gemm(const GemmArgs& a, int epi, int mt, hipStream_t s);
void set_pgemm_min_rows(int rows);   // row count from which launch_gemm takes the prefill schedule (process-wide; < 0 = default)
void set_pgemm_wide_rows(int rows);  // row count from which its 4-segment forms take 128 x 128 tiles (0 = never; < 0 = default 2048)

// Per-row record of a step (int32 words), built by the scheduler and uploaded once per step:
//   [0] stream  [1] position  [2] embed kind  [3] embed a  [4] embed b  [5..7] unused  [8 ..] KV block ids of the row's stream
// One lookup gives a kernel everything it needs to address the row's paged KV (no stream -> block-table hop).
constexpr int ROW_HDR = 8;
inline int row_stride_words(int max_blocks) { return ROW_HDR + ((max_blocks + 3) & ~3); }

struct EmbedArgs {
    const int* rowrec; int row_stride;   // row records
    const float* cond;         // [max_seqs][34][1024] fp32
    const uint16_t *text_emb, *text_pos, *speech_emb, *speech_pos;
    uint16_t* h;               // [rows][1024]
    int rows;
    const int* prev_tok;       // the sampler's output array of the previous step (EMB_SPEECH_PREV)
    // The step's metadata (selection arrays + row records) as the host wrote it, in pinned host memory: this launch reads its own row
    // records from there and leaves the device copy every later launch of the step reads (no copy kernel in front of the step).
    const int4* host_meta = nullptr; int4* dev_meta = nullptr; int meta_vec = 0;      // meta_vec 16-byte pieces
    const int* host_rowrec = nullptr;                                               // rowrec's twin inside host_meta
};
enum EmbedKind { EMB_COND = 0 /*a=slot,b=idx*/, EMB_TEXT = 1 /*a=id,b=pos*/, EMB_ZERO = 2, EMB_SPEECH = 3 /*a=id,b=pos*/,
                 EMB_SPEECH_PREV = 4 /*a=index into prev_tok,b=pos*/ };
hipError_t launch_embed(const EmbedArgs& a, hipStream_t s);

struct RopeArgs {
    const uint16_t* qkv;       // [rows][3072]
    uint16_t* q_out;           // [rows][1024]
    uint16_t* kv_layer;        // pool + layer*n_blocks*KV_BLOCK_ELEMS
    const int* rowrec; int row_stride;   // row records
    const float *cos_t, *sin_t;  // [max_pos][32]
    int rows;
};
hipError_t launch_rope_kv(const RopeArgs& a, hipStream_t s);

struct AttnArgs {
    const uint16_t* q;         // [rows][1024] rotated (unfused form)
    const uint16_t* kv_layer;
    const int* rowrec; int row_stride;   // row records
    uint16_t* out;             // [rows][1024]
    int rows;
    int max_chunks;            // LDS sizing: ceil(max_model_len/64)
    // fused decode form (every row is the newest position of its stream): RoPE of q,k + KV write happen here
    const uint16_t* qkv;       // [rows][3072] pre-RoPE, or null for the unfused form
    uint16_t* kv_layer_w;      // writable alias of kv_layer
    const float *cos_t, *sin_t;
    // unfused form: rows [tile_from, rows) are prefill rows and take the 16-rows-per-workgroup schedule (-1: none); tile_chunks =
    // ceil(longest context among them / 64) sizes its LDS (0: max_chunks)
    int tile_from = -1, tile_chunks = 0;
    int force_waves = 0;       // 4 / 8: waves per (row, head) workgroup of the per-row kernel (0: by row count; parity tests check both)
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);
// parity hook: K (as stored, i.e. rotated) and V of every row's (stream, position) read back from the paged pool -> out [rows][2][1024]
hipError_t launch_kv_gather(const uint16_t* kv_layer, const int* rowrec, int row_stride, int rows, uint16_t* out, hipStream_t s);

struct SampleArgs {
    const uint16_t* logits;    // [2*n][ldl] bf16: row 2i cond, 2i+1 uncond
    int ldl;
    const int4* sel;           // per sampled utterance {slot, step, 0, 0}
    uint16_t* counts;          // [max_seqs][VPAD]
    const T3Sampling* sp;      // [max_seqs], indexed by slot
    float cfg;
    int* out_tok;              // [n]
    float* dbg;                // nullable: [max_seqs][V]
    int n;
    int* hist = nullptr;       // nullable: [max_seqs][hist_cap] speech-space ids of every utterance, kept on the device (f4 hand-off)
    int hist_cap = 0;
    int* out_tok_host = nullptr;   // nullable: [n] in pinned host memory, written beside out_tok (no copy kernel behind the step)
    unsigned char* dbg_keep = nullptr;   // nullable (parity hook t3k_sample_support): [max_seqs][V] 1 where the draw can return the id
};
hipError_t launch_sampler(const SampleArgs& a, hipStream_t s);
// Profile mode: the next single-kernel launch of this thread (decode GEMM forms, fused / per-row attention, embed, sampler) takes these as
// its start / stop events (hipExtLaunchKernelGGL); launch_events_armed() afterwards tells whether a launcher consumed them.
void arm_launch_events(hipEvent_t start, hipEvent_t stop);
bool launch_events_armed();
hipError_t prepare_kernels();   // one-time function attributes (must run before any stream capture)
hipError_t prepare_gemm2();     // its GEMM part (t3_gemm.hip)
void gemm_refresh_switches();   // re-reads the measurement switches of the GEMM launchers (per engine)
hipError_t launch_expf(const float* x, float* y, int n, hipStream_t s);

// f4: batched token hand-off to the vocoder.  One workgroup per utterance applies the reference's post-filter (tts.py:300-365 +
// alignment_stream_analyzer.py:111-201 + the range filter of tts.py:514 -- the same integer rules as t3_clean_tokens) to the
// utterance's device-resident ids and writes one padded row + length.
struct HandoffItem { const int* src; int n; int text_token_count; int _pad; };
hipError_t launch_handoff(const HandoffItem* items /*device [n_utt]*/, int n_utt, int flags, int* out /*device [n_utt][ld]*/, int ld, int* lens /*device [n_utt]*/, hipStream_t s);

void rope_tables(int max_pos, float* cos_t, float* sin_t);   // host, llama3 scaling, bf16-valued

}  // namespace t3
