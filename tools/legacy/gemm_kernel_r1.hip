// RECORD, not shipped: the round-1 skinny GEMM (symmetric register ring for weights and fragment-shaped activation loads, any K that is a
// multiple of 32 x waves).  No engine path selected it after round 2 (gemm2_kernel / gemm2_loop_kernel / pgemm_kernel cover every
// row count); it left the shipped translation unit in round 4.  Compiles when included behind csrc/t3_gemm.hip (it uses GemmArgs, f32x4,
// ld_nt, as_frag, silu_mul_bf ...): `make -C tools legacy-check`.
#define T3_GSTAMP(i)
namespace t3 {
template <int MT, int NT, int EPI, int PD, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_kernel(GemmArgs a) {
    T3_GSTAMP(0);
    extern __shared__ __attribute__((aligned(16))) float red[];   // [NW waves][MT*NT][4 regs][64 lanes]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int KB = a.K >> 5, kbs = KB / NW, kb0 = wave * kbs;

    const uint4* wp[NT];
    const uint4* xp[MT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = a.Wp + ((size_t)(blockIdx.x * NT + t) * KB + kb0) * 64 + lane;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = (blockIdx.y * MT + i) * 16 + c;
        m = m < a.M ? m : a.M - 1;                 // padded rows re-read the last row; their outputs are dropped
        if (a.row_index) m = a.row_index[m];
        xp[i] = reinterpret_cast<const uint4*>(a.X + (size_t)m * a.K + kb0 * 32 + q * 8);
#ifdef T3_GEMM_XDUMMY      // timing diagnostic only (wrong results): every wave reads the same 1 KiB of activations
        xp[i] = reinterpret_cast<const uint4*>(a.X + q * 8 + c * 32);
#endif
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // every thread finishes MT*NT*256 / (64*NW) outputs; D[row = 4*(lane>>4) + reg][col = lane&15]
    constexpr int TOTAL = MT * NT * 256, STEP = NW * 64, ITER = (TOTAL + STEP - 1) / STEP;
    // EPI_RESID: the residual operand of this thread's outputs is requested now, so that its HBM round trip overlaps
    // the weight stream instead of sitting between the reduction and the store
    float hres[ITER];
    if constexpr (EPI == EPI_RESID) {
#pragma unroll
        for (int k = 0; k < ITER; ++k) {
            const int idx = threadIdx.x + k * STEP;
            const int it = idx >> 8, r = (idx >> 6) & 3, l2 = idx & 63;
            const int m = (blockIdx.y * MT + it / NT) * 16 + 4 * (l2 >> 4) + r, n = (blockIdx.x * NT + it % NT) * 16 + (l2 & 15);
            hres[k] = (idx < TOTAL && m < a.M && n < a.N) ? bf2f(reinterpret_cast<const uint16_t*>(a.out)[(size_t)m * a.ldo + n]) : 0.0f;
        }
    }

    uint4 wr[PD][NT], xr[PD][MT];
#pragma unroll
    for (int j = 0; j < PD; ++j)
        if (j < kbs) {
#pragma unroll
            for (int t = 0; t < NT; ++t) wr[j][t] = ld_nt(wp[t] + j * 64);
#pragma unroll
            for (int i = 0; i < MT; ++i) xr[j][i] = xp[i][j * 4];
        }
    for (int kbase = 0; kbase < kbs; kbase += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const int kb = kbase + j;
            if (kb < kbs) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(xr[j][i]), as_frag(wr[j][t]), acc[i][t], 0, 0, 0);
                if (kb == 0) T3_GSTAMP(1);
                if (kb + PD < kbs) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) wr[j][t] = ld_nt(wp[t] + (kb + PD) * 64);
#pragma unroll
                    for (int i = 0; i < MT; ++i) xr[j][i] = xp[i][(kb + PD) * 4];
                }
            }
        }
    }

    T3_GSTAMP(2);
    // cross-wave (= cross-segment) reduction in segment order
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave * (MT * NT) + i * NT + t) * 4 + r) * 64 + lane] = acc[i][t][r];
    __syncthreads();
    T3_GSTAMP(3);

#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const int idx = threadIdx.x + k * STEP;
        if (idx >= TOTAL) continue;
        const int it = idx >> 8, r = (idx >> 6) & 3, l2 = idx & 63;
        const int i = it / NT, t = it % NT;
        const int m = (blockIdx.y * MT + i) * 16 + 4 * (l2 >> 4) + r;
        if (m >= a.M) continue;
        if (EPI == EPI_SILU && (t & 1)) continue;                 // packed tiles come in (gate, up) pairs
        float v[EPI == EPI_SILU ? 2 : 1];
#pragma unroll
        for (int u = 0; u < (EPI == EPI_SILU ? 2 : 1); ++u) {
            const int itu = it + u;
            float tot = 0.0f;
#pragma unroll
            for (int gsum = 0; gsum < NW / 4; ++gsum) {
                float s4 = red[(((4 * gsum + 0) * (MT * NT) + itu) * 4 + r) * 64 + l2];
                s4 = s4 + red[(((4 * gsum + 1) * (MT * NT) + itu) * 4 + r) * 64 + l2];
                s4 = s4 + red[(((4 * gsum + 2) * (MT * NT) + itu) * 4 + r) * 64 + l2];
                s4 = s4 + red[(((4 * gsum + 3) * (MT * NT) + itu) * 4 + r) * 64 + l2];
                tot = gsum == 0 ? s4 : tot + s4;
            }
            v[u] = tot;
        }
        if constexpr (EPI == EPI_SILU) {
            const int n = (blockIdx.x * (NT / 2) + (t >> 1)) * 16 + (l2 & 15);      // tile pair index == output tile index
            if (n < a.N)
                reinterpret_cast<uint16_t*>(a.out)[(size_t)m * a.ldo + n] = (uint16_t)silu_mul_bf(f2bf(v[0]), f2bf(v[1]));
        } else {
            const int n = (blockIdx.x * NT + t) * 16 + (l2 & 15);
            if (n >= a.N) continue;
            if constexpr (EPI == EPI_F32) {
                reinterpret_cast<float*>(a.out)[(size_t)m * a.ldo + n] = v[0];
            } else if constexpr (EPI == EPI_BF16) {
                reinterpret_cast<uint16_t*>(a.out)[(size_t)m * a.ldo + n] = (uint16_t)f2bf(v[0]);
            } else {   // EPI_RESID: h = bf16(h + bf16(y))
                reinterpret_cast<uint16_t*>(a.out)[(size_t)m * a.ldo + n] = (uint16_t)f2bf(hres[k] + rbf(v[0]));
            }
        }
    }
    T3_GSTAMP(4);
}
template <int MT, int NT, int EPI, int NW>
static hipError_t launch_gemm_t(const GemmArgs& a, hipStream_t s) {
    // ring depth, bounded by the register file: 4-wave workgroups may use ~200 VGPRs, 16-wave ones 128
    constexpr int PD = NW == 16 ? (MT <= 2 ? 4 : 2) : ((MT + NT) <= 6 ? 8 : 4);
    const int ntiles = (a.N + 15) / 16;           // EPI_SILU: N = F -> one workgroup per output tile (2 packed tiles)
    const int gx = (EPI == EPI_SILU) ? (ntiles + NT / 2 - 1) / (NT / 2) : (ntiles + NT - 1) / NT;
    const int gy = ((a.M + 15) / 16 + MT - 1) / MT;
    const size_t lds = (size_t)NW * MT * NT * 256 * sizeof(float);        // <= 64 KiB for every instantiation below
    hipLaunchKernelGGL((gemm_kernel<MT, NT, EPI, PD, NW>), dim3(gx, gy), dim3(NW * 64), lds, s, a);
    return hipGetLastError();
}
}  // namespace t3
