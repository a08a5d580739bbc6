// File-driven MFMA probe (numerics research, not part of the product).
//   mfma_probe2 <kind> <A.bin> <B.bin> <C.bin> <D.bin>     kind: bf16_16x16x32 | bf16_32x32x16 | f32_16x16x4
// A: [T][M][K], B: [T][K][N] (bf16 bits as uint16, or fp32), C/D: [T][M][N] fp32.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k_f32_16x16x4(const float* A, const float* B, const float* C, float* Dm) {
    const int t = blockIdx.x, l = threadIdx.x;
    const float a = A[t * 64 + (l & 15) * 4 + (l >> 4)];
    const float b = B[t * 64 + (l >> 4) * 16 + (l & 15)];
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[t * 256 + (4 * (l >> 4) + r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) Dm[t * 256 + (4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
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

static std::vector<char> slurp(const char* p) {
    FILE* f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> v(n); if (fread(v.data(), 1, n, f) != (size_t)n) exit(2); fclose(f); return v;
}

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage\n"); return 2; }
    const std::string kind = argv[1];
    auto A = slurp(argv[2]), B = slurp(argv[3]), C = slurp(argv[4]);
    const size_t csz = kind == "bf16_32x32x16" ? 1024 : 256;
    const int T = (int)(C.size() / 4 / csz);
    void *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, C.size()); hipMalloc(&dD, C.size());
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size(), hipMemcpyHostToDevice);
    if (kind == "bf16_16x16x32") hipLaunchKernelGGL(k_bf16_16x16x32, dim3(T), dim3(64), 0, 0, (uint16_t*)dA, (uint16_t*)dB, (float*)dC, (float*)dD);
    else if (kind == "bf16_32x32x16") hipLaunchKernelGGL(k_bf16_32x32x16, dim3(T), dim3(64), 0, 0, (uint16_t*)dA, (uint16_t*)dB, (float*)dC, (float*)dD);
    else hipLaunchKernelGGL(k_f32_16x16x4, dim3(T), dim3(64), 0, 0, (float*)dA, (float*)dB, (float*)dC, (float*)dD);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    std::vector<char> D(C.size());
    hipMemcpy(D.data(), dD, C.size(), hipMemcpyDeviceToHost);
    FILE* f = fopen(argv[5], "wb"); fwrite(D.data(), 1, D.size(), f); fclose(f);
    printf("%s: %d tiles\n", kind.c_str(), T);
    return 0;
}
