// lm_gemm_norm.h - the RMSNorm prologue shared by k_gemm_norm (lm_kernels.hip) and k_gemm_norm_q (lm_qgemm.hip): every block of a
// consumer GEMM rebuilds its X operand from the residual stream h [Mpad][d] (bf16 rows) in registers,
//     x = T(w . T(h . rsqrt(mean h^2 + eps)))                              (LlamaTTS.swift:306, the glue kernel's rounding points)
// Wave `wave` of a KSB-wave block owns the k-tiles kt0 .. kt0 + XT - 1 of all 16 MT rows: its 16-byte loads of h are the MFMA B
// fragments (lane (j, q) of fragment (kt, mt) holds row 16 mt + j, columns 32 kt + 8 q ..+8); the row sums of squares go lane ->
// the four lanes of a row (shuffles) -> the KSB waves (rsum in LDS, summed in wave order: deterministic).  Contains one __syncthreads.
// The arithmetic is ~10 VALU operations per element and every block repeats it for the whole operand, so the block is WIDE: 8 or 16
// waves (XT = 1..4 k-tiles each) - with 4 waves x 8 k-tiles the prologue alone took ~3 us at hidden 1024 (profiles/r03/q3_fused_*.csv).
#pragma once
#include "common.h"

template <int MT, int XT, int KSB>
__device__ __forceinline__ void gemm_norm_prologue(const bf16_t* __restrict__ h, const bf16_t* __restrict__ wnorm, int KT, float eps, int kt0,
                                                   int lane, int wave, float (*rsum)[MT * 16], bf16x8_t (&xr)[XT][MT]) {
    const int d = KT * 32;
    const int j = lane & 15, q = lane >> 4;
    uint4 hq[XT][MT], wq[XT];
#pragma unroll
    for (int u = 0; u < XT; ++u) {
        int kk = kt0 + u;
        kk = kk < KT ? kk : KT - 1;
        const int k0 = kk * 32 + q * 8;
        wq[u] = *reinterpret_cast<const uint4*>(wnorm + k0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hq[u][mt] = *reinterpret_cast<const uint4*>(h + (size_t)(mt * 16 + j) * d + k0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float v = 0.0f;
#pragma unroll
        for (int u = 0; u < XT; ++u) {
            const uint32_t hw[4] = {hq[u][mt].x, hq[u][mt].y, hq[u][mt].z, hq[u][mt].w};
            float t = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h0 = bf16_to_f32((bf16_t)(hw[e] & 0xffffu)), h1 = bf16_to_f32((bf16_t)(hw[e] >> 16));
                t += h0 * h0 + h1 * h1;
            }
            v += (kt0 + u < KT) ? t : 0.0f;
        }
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) rsum[wave][mt * 16 + j] = v;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float tot = rsum[0][mt * 16 + j];
#pragma unroll
        for (int w = 1; w < KSB; ++w) tot += rsum[w][mt * 16 + j];
        const float inv = 1.0f / sqrtf(tot / (float)d + eps);
#pragma unroll
        for (int u = 0; u < XT; ++u) {
            const uint32_t ww[4] = {wq[u].x, wq[u].y, wq[u].z, wq[u].w};
            const uint32_t hw[4] = {hq[u][mt].x, hq[u][mt].y, hq[u][mt].z, hq[u][mt].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float w0 = bf16_to_f32((bf16_t)(ww[e] & 0xffffu)), w1 = bf16_to_f32((bf16_t)(ww[e] >> 16));
                const float n0 = bf16_to_f32((bf16_t)(hw[e] & 0xffffu)), n1 = bf16_to_f32((bf16_t)(hw[e] >> 16));
                xr[u][mt][2 * e] = (short)f32_to_bf16(w0 * bf16_round_f32(n0 * inv));
                xr[u][mt][2 * e + 1] = (short)f32_to_bf16(w1 * bf16_round_f32(n1 * inv));
            }
        }
    }
}
