// Numerics probe (not part of the product): dumps inputs/outputs of single MFMA instructions on random
// data so that their internal summation order / rounding can be identified offline
// (tools/analyze_mfma_probe.py).  Usage: mfma_probe <out_dir>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <random>
#include <string>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// one wave per tile.  f32 16x16x4: A[16][4], B[4][16], C[16][16]
__global__ void k_f32_16x16x4(const float* A, const float* B, const float* C, float* Dm) {
    const int t = blockIdx.x, l = threadIdx.x;
    const float a = A[t * 64 + (l & 15) * 4 + (l >> 4)];
    const float b = B[t * 64 + (l >> 4) * 16 + (l & 15)];
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[t * 256 + (4 * (l >> 4) + r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) Dm[t * 256 + (4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
// bf16 16x16x32: A[16][32], B[32][16] (bf16 bits), C[16][16]
__global__ void k_bf16_16x16x32(const uint16_t* A, const uint16_t* B, const float* C, float* Dm) {
    const int t = blockIdx.x, l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[t * 512 + (l & 15) * 32 + 8 * (l >> 4) + j];
        b[j] = (short)B[t * 512 + (8 * (l >> 4) + j) * 16 + (l & 15)];
    }
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[t * 256 + (4 * (l >> 4) + r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) Dm[t * 256 + (4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
// bf16 32x32x16: A[32][16], B[16][32], C[32][32]
__global__ void k_bf16_32x32x16(const uint16_t* A, const uint16_t* B, const float* C, float* Dm) {
    const int t = blockIdx.x, l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[t * 512 + (l & 31) * 16 + 8 * (l >> 5) + j];
        b[j] = (short)B[t * 512 + (8 * (l >> 5) + j) * 32 + (l & 31)];
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[t * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) Dm[t * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

static void dump(const std::string& p, const void* d, size_t n) { FILE* f = fopen(p.c_str(), "wb"); fwrite(d, 1, n, f); fclose(f); }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    const int T = 512;
    std::mt19937 rng(12345);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_int_distribution<int> ex(-6, 6);
    auto rnd = [&]() { return ldexpf(nd(rng), ex(rng)); };
    auto run = [&](const char* name, auto kernel, size_t na, size_t nb, size_t nc, bool bf) {
        std::vector<float> A(na * T), B(nb * T), C(nc * T), Dm(nc * T);
        std::vector<uint16_t> Ab(na * T), Bb(nb * T);
        for (auto& v : A) v = rnd();
        for (auto& v : B) v = rnd();
        for (size_t i = 0; i < C.size(); ++i) C[i] = (i / nc) % 2 ? rnd() * 8.f : 0.f;   // even tiles: C = 0
        if (bf) for (size_t i = 0; i < A.size(); ++i) { Ab[i] = f2bf(A[i]); }
        if (bf) for (size_t i = 0; i < B.size(); ++i) { Bb[i] = f2bf(B[i]); }
        void *dA, *dB, *dC, *dD;
        const size_t ea = bf ? 2 : 4;
        hipMalloc(&dA, A.size() * ea); hipMalloc(&dB, B.size() * ea); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, C.size() * 4);
        hipMemcpy(dA, bf ? (void*)Ab.data() : (void*)A.data(), A.size() * ea, hipMemcpyHostToDevice);
        hipMemcpy(dB, bf ? (void*)Bb.data() : (void*)B.data(), B.size() * ea, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        kernel(dA, dB, dC, dD);
        hipDeviceSynchronize();
        hipMemcpy(Dm.data(), dD, C.size() * 4, hipMemcpyDeviceToHost);
        const std::string base = dir + "/" + name;
        dump(base + "_A.bin", bf ? (void*)Ab.data() : (void*)A.data(), A.size() * ea);
        dump(base + "_B.bin", bf ? (void*)Bb.data() : (void*)B.data(), B.size() * ea);
        dump(base + "_C.bin", C.data(), C.size() * 4); dump(base + "_D.bin", Dm.data(), C.size() * 4);
        printf("%s: %d tiles dumped, D[0]=%g\n", name, T, Dm[0]);
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    };
    run("f32_16x16x4", [&](void* a, void* b, void* c, void* d) { hipLaunchKernelGGL(k_f32_16x16x4, dim3(T), dim3(64), 0, 0, (float*)a, (float*)b, (float*)c, (float*)d); }, 64, 64, 256, false);
    run("bf16_16x16x32", [&](void* a, void* b, void* c, void* d) { hipLaunchKernelGGL(k_bf16_16x16x32, dim3(T), dim3(64), 0, 0, (uint16_t*)a, (uint16_t*)b, (float*)c, (float*)d); }, 512, 512, 256, true);
    run("bf16_32x32x16", [&](void* a, void* b, void* c, void* d) { hipLaunchKernelGGL(k_bf16_32x32x16, dim3(T), dim3(64), 0, 0, (uint16_t*)a, (uint16_t*)b, (float*)c, (float*)d); }, 512, 512, 1024, true);
    return 0;
}
