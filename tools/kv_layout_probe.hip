// Probe: how much of the attention kernel's time is its ADDRESS PATTERN?  Loads-only replicas of the fused decode attention's K/V traffic
// (64 rows x 16 heads = 1024 workgroups of 4 waves, chunks of 64 tokens = 8 KB K + 8 KB V dealt round-robin over the waves, 16 x 1 KB
// wave loads per chunk, a full wait, next chunk) over different pool layouts, against a plain contiguous stream of the same bytes.
//   layout 0: the engine's  [block 256 tok][k|v][head][chunk][8 KB] with the 1 KB pad per (k|v, head) region   (K and V of a chunk 528 KB apart)
//   layout 1: [block][head][chunk][k 8 KB | v 8 KB]            (a chunk's 16 KB contiguous, a head's block 64 KB contiguous)
//   layout 2: per (stream, head) fully contiguous: [stream][head][chunk][k|v]   (what a non-paged cache would look like)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kv_layout_probe.hip -o tools/kv_layout_probe ; run: tools/kv_layout_probe [ctx] [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr int H = 16, CHUNK_B = 8192, BLOCK_TOK = 256, CPB = 4;
struct P { const char* base; const int* table; int max_blocks; int nc; int layout; size_t head_stride, kv_stride, block_stride, stream_stride; unsigned* sink; };
__global__ __launch_bounds__(256, 4) void attn_like(P p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = blockIdx.x, row = blockIdx.y;
    u4 acc = {0, 0, 0, 0};
    for (int c = wave; c < p.nc; c += 4) {
        const char *kp, *vp;
        if (p.layout == 2) { kp = p.base + (size_t)row * p.stream_stride + (size_t)h * p.head_stride + (size_t)c * 2 * CHUNK_B; vp = kp + CHUNK_B; }
        else {
            const int blk = p.table[row * p.max_blocks + c / CPB], ci = c % CPB;
            if (p.layout == 0) { kp = p.base + (size_t)blk * p.block_stride + (size_t)h * p.head_stride + (size_t)ci * CHUNK_B; vp = kp + p.kv_stride; }
            else { kp = p.base + (size_t)blk * p.block_stride + (size_t)h * p.head_stride + (size_t)ci * 2 * CHUNK_B; vp = kp + CHUNK_B; }
        }
        u4 k[8], v[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) k[f] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(kp) + f * 64 + lane);
#pragma unroll
        for (int f = 0; f < 8; ++f) v[f] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(vp) + f * 64 + lane);
#pragma unroll
        for (int f = 0; f < 8; ++f) acc ^= k[f] ^ v[f];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *p.sink = 1;
}
__global__ __launch_bounds__(256, 4) void plain_stream(const u4* src, size_t per_wg_vec, unsigned* sink) {
    const u4* q = src + (size_t)blockIdx.x * per_wg_vec;
    u4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < per_wg_vec; i += 1024) {
        u4 a = __builtin_nontemporal_load(q + i), b = i + 256 < per_wg_vec ? __builtin_nontemporal_load(q + i + 256) : acc;
        u4 c = i + 512 < per_wg_vec ? __builtin_nontemporal_load(q + i + 512) : acc, d = i + 768 < per_wg_vec ? __builtin_nontemporal_load(q + i + 768) : acc;
        acc ^= a ^ b ^ c ^ d;
    }
    if ((acc.x ^ acc.y) == 0x12345u) *sink = 1;
}
int main(int argc, char** argv) {
    const int ctx = argc > 1 ? atoi(argv[1]) : 559, rows = argc > 2 ? atoi(argv[2]) : 64, NL = 8;
    const int nc = (ctx + 63) / 64, max_blocks = 4, nblocks = rows * max_blocks;
    const size_t pad = 1024;
    // layout 0 strides (bytes): per (k|v, head) region 4 chunks x 8 KB + pad
    const size_t h0 = CPB * CHUNK_B + pad, kv0 = H * h0, b0 = 2 * kv0;
    const size_t h1 = CPB * 2 * CHUNK_B + pad, b1 = H * h1;
    const size_t h2 = (size_t)max_blocks * CPB * 2 * CHUNK_B + pad, s2 = H * h2;
    const size_t layer_bytes = std::max((size_t)nblocks * b0, std::max((size_t)nblocks * b1, (size_t)rows * s2));
    char* pool; unsigned* sink; int* dtab;
    CK(hipMalloc((void**)&pool, layer_bytes * NL)); CK(hipMemset(pool, 1, layer_bytes * NL)); CK(hipMalloc((void**)&sink, 4));
    std::vector<int> tab(nblocks);
    for (int i = 0; i < nblocks; ++i) tab[i] = (int)(((long)i * 7919 + 13) % nblocks);
    CK(hipMalloc((void**)&dtab, nblocks * 4)); CK(hipMemcpy(dtab, tab.data(), nblocks * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double mb = (double)rows * H * nc * 2 * CHUNK_B / 1e6;
    auto run = [&](const char* name, auto launch) {
        for (int w = 0; w < 16; ++w) launch(w % NL);
        hipStreamSynchronize(st); hipEventRecord(e0, st);
        const int reps = 160;
        for (int r = 0; r < reps; ++r) launch(r % NL);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %7.2f us per launch  %6.2f TB/s\n", name, ms * 1e3 / reps, mb / (ms * 1e3 / reps) / 1e6 * 1e6 / 1e6);
    };
    printf("ctx %d (%d chunks), %d rows x 16 heads, %.1f MB per launch (back-to-back launches over %d layer regions)\n", ctx, nc, rows, mb, NL);
    for (int layout = 0; layout < 3; ++layout) {
        P p{pool, dtab, max_blocks, nc, layout, layout == 0 ? h0 : layout == 1 ? h1 : h2, kv0, layout == 0 ? b0 : b1, s2, sink};
        const char* names[3] = {"attention pattern, engine layout [blk][k|v][head][chunk]", "attention pattern, layout [blk][head][chunk][k|v]", "attention pattern, contiguous per (stream, head)"};
        run(names[layout], [&](int l) { P q = p; q.base = pool + (size_t)l * layer_bytes; hipLaunchKernelGGL(attn_like, dim3(H, rows), dim3(256), 0, st, q); });
    }
    const size_t per_wg = (size_t)nc * 2 * CHUNK_B / 16;
    run("plain contiguous stream, same bytes, 1024 workgroups", [&](int l) { hipLaunchKernelGGL(plain_stream, dim3(rows * H), dim3(256), 0, st, (const u4*)(pool + (size_t)l * layer_bytes), per_wg, sink); });
    return 0;
}
