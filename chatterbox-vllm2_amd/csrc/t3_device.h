// Device-side helpers shared by the kernel translation units (t3_gemm.hip, t3_attention.hip, t3_kernels.hip, tools/chain_kernel.hip): bf16 conversions at the
// contract's rounding points, the contract exp, SwiGLU, MFMA fragment types and streamed loads.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace t3 {

// Profile mode (engine.cpp: Prof): the NEXT single-kernel launch of this thread carries these two events as its start / stop events
// (hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps, what rocprofv3 reports), instead of being bracketed by two
// hipEventRecord barrier packets, which add ~2-3 us of command-processor time to a 5-30 us kernel.
inline thread_local hipEvent_t g_arm_start = nullptr, g_arm_stop = nullptr;
template <typename F, typename... Args>
static inline void launch_k(F kernel, const dim3& grid, const dim3& block, size_t lds, hipStream_t s, Args... args) {
    if (g_arm_start) {
        hipEvent_t a = g_arm_start, b = g_arm_stop;
        g_arm_start = nullptr; g_arm_stop = nullptr;
        hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, s, a, b, 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
    }
}
// hipFuncSetAttribute applies to the CURRENT device: every "already raised" flag of a launcher is kept per device, so a process that
// drives engines on several GPUs (LLM(device_id=...)) raises the limits on each of them
constexpr int MAX_DEVICES = 64;
static inline int cur_device() { int d = 0; (void)hipGetDevice(&d); return d >= 0 && d < MAX_DEVICES ? d : 0; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
// fp32 -> bf16 round-to-nearest-even, NaN stays quiet NaN
__device__ __forceinline__ uint32_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float rbf(float f) { return __uint_as_float(f2bf(f) << 16); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { return f2bf(lo) | (f2bf(hi) << 16); }

// element j (0..7) of a 16-byte vector of 8 bf16, as fp32
template <int J>
__device__ __forceinline__ float elem(const uint4& v) {
    const uint32_t w = (J >> 1) == 0 ? v.x : (J >> 1) == 1 ? v.y : (J >> 1) == 2 ? v.z : v.w;
    return (J & 1) ? bf_hi(w) : bf_lo(w);
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}

// Contract exp: Cody-Waite + degree-6 Horner; only fma / mul / round-to-nearest-even.
__device__ __forceinline__ float t3_expf(float x) {
    if (!(x >= -87.0f)) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    const float n = __builtin_rintf(x * 1.44269502162933349609375f);
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.428606765330187045e-06f, r);
    float p = 1.388888922519981861e-03f;
    p = __builtin_fmaf(p, r, 8.333333767950534821e-03f);
    p = __builtin_fmaf(p, r, 4.166666790843009949e-02f);
    p = __builtin_fmaf(p, r, 1.666666716337203979e-01f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    const int ni = (int)n;
    const float s = __uint_as_float((uint32_t)(ni + 127) << 23);
    return p * s;
}

__device__ __forceinline__ uint32_t silu_mul_bf(uint32_t g, uint32_t u) {   // bf16 bits in, bf16 bits out
    const float gf = __uint_as_float(g << 16);
    const float sg = rbf(gf / (1.0f + t3_expf(-gf)));
    return f2bf(sg * __uint_as_float(u << 16));
}

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_nt(const uint4* p) {      // streamed-once data: non-temporal load
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ bf16x8 as_frag(const uint4& v) {
    union { uint4 u; bf16x8 f; } c; c.u = v; return c.f;
}
// two fp32 -> packed bf16 pair, round-to-nearest-even (v_cvt_pk_bf16_f32; equals f2bf() for every non-NaN input)
__device__ __forceinline__ uint32_t cvt_pk(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    union { bf2 v; uint32_t u; } c;
    c.v = (bf2){(__bf16)lo, (__bf16)hi};
    return c.u;
}


}  // namespace t3
