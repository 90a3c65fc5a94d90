// whisper_kernels.hip - Whisper encoder kernels (compute-bound side of the STT path) for gfx950.
//
// Reference being replaced: WhisperEncoder / WhisperEncoderLayer / WhisperAttention
// (Sources/MLXAudioSTT/Models/Whisper/WhisperLayers.swift:11-156) whose arithmetic is MLX (Conv1d, Linear,
// LayerNorm, gelu, MLXFast.scaledDotProductAttention).  The decoder's per-token path reuses the
// weight-streaming kernels of lm_kernels.hip.  bf16 storage, f32 accumulation, rounding at every MLX
// primitive boundary (oracle/whisper.py).
//
//   k_gemm_big      C[M][N] = X[M][K] W[N][K]^T on v_mfma_f32_16x16x32_bf16, 128x128x32 LDS tiles, fused
//                   bias / exact-erf GELU / residual / positional-embedding epilogues (M = batch*1500 rows)
//   k_layernorm     row LayerNorm (f32 statistics)
//   k_im2col3       k=3 convolution patches (stride 1 / 2) so both stem convs run on k_gemm_big
//   k_scatter_kv    [M][*] K and V columns -> the tiled MFMA-fragment cache layouts of lm_kernels.hip
//   k_attn_prefill  non-causal flash attention over the tiled K/V, 16 query rows per wave
#include "common.h"
#include "whisper_kernels.h"
#include <type_traits>

// GELU (erf form, WhisperLayers.swift:142-156 via MLXNN.GELU) for the encoder's big-GEMM epilogues.  erff() from the device library is
// ~60 instructions per element once both of its range branches run in a wave - 15 us per 256 x 256 output tile, fully exposed with one
// block per CU (k_gemm_big3<GELU> 237.7 us against 187.6 us for the same main loop without epilogue, profiles/r04/c6_whisper_kernel_stats.csv).
// Here: erfc(|z|) = poly5(t) exp(-z^2), t = 1 / (1 + 0.3275911 |z|) (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7), branch-free, and
// 1 + erf(z) taken as erfc(-z) for z < 0 - no cancellation on the negative side.  Against the float64 value, after the bf16 rounding
// that follows: max |diff| 1.0e-6 of full scale, fewer mismatching roundings than the float32 0.5 x (1 + erf) form itself has.
__device__ __forceinline__ float gelu_erf_w(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __frcp_rn(1.0f + 0.3275911f * z);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float ec = p * t * __expf(-(z * z));            // erfc(|x| / sqrt 2)
    return 0.5f * x * (x >= 0.0f ? 2.0f - ec : ec);
}

// ============================================================================ big GEMM
#define BG_BM 128
#define BG_BN 128
#define BG_BK 32
#define BG_LD 40          // padded LDS row (bf16 elements): 80 B rows keep 16-B fragments aligned

template <int EPI>
__global__ void __launch_bounds__(256) k_gemm_big(BigGemmParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t Ws[BG_BN][BG_LD];
    __shared__ __attribute__((aligned(16))) bf16_t Xs[BG_BM][BG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * BG_BN, m0 = blockIdx.y * BG_BM;
    const int wn = wave >> 1, wm = wave & 1;            // wave tile: 64 n x 64 m
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // staging: 128 rows x 32 k = 512 chunks of 16 B per operand, two per thread
    const int r0 = tid >> 2, c0 = (tid & 3) * 8;
    uint4 wreg[2], xreg[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int row = r0 + 64 * h;
            int n = n0 + row, m = m0 + row;
            wreg[h] = (n < p.N) ? *reinterpret_cast<const uint4*>(p.W + (size_t)n * p.K + k0 + c0) : make_uint4(0, 0, 0, 0);
            xreg[h] = (m < p.M) ? *reinterpret_cast<const uint4*>(p.X + (size_t)m * p.ldx + k0 + c0) : make_uint4(0, 0, 0, 0);
        }
    };
    load(0);
    for (int k0 = 0; k0 < p.K; k0 += BG_BK) {
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<uint4*>(&Ws[r0 + 64 * h][c0]) = wreg[h];
            *reinterpret_cast<uint4*>(&Xs[r0 + 64 * h][c0]) = xreg[h];
        }
        __syncthreads();
        if (k0 + BG_BK < p.K) load(k0 + BG_BK);
        bf16x8_t a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[i] = *reinterpret_cast<const bf16x8_t*>(&Ws[wn * 64 + i * 16 + (lane & 15)][(lane >> 4) * 8]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            b[j] = *reinterpret_cast<const bf16x8_t*>(&Xs[wm * 64 + j * 16 + (lane & 15)][(lane >> 4) * 8]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    // epilogue: C/D lane l, reg r: n = (l>>4)*4 + r (A rows), m = l&15 (B cols) -> 4 consecutive n per lane
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
        if (n >= p.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = (n + e < p.N) ? bf16_to_f32(p.bias[n + e]) : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = m0 + wm * 64 + j * 16 + (lane & 15);
            if (m >= p.M) continue;
            uint16_t res[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = bf16_round_f32(acc[i][j][e] + bv[e]);                       // T(xW^T + b)
                if (EPI == BG_GELU || EPI == BG_GELU_POS) v = bf16_round_f32(gelu_erf_w(v));
                if (EPI == BG_RESID) v = bf16_round_f32(v + bf16_to_f32(p.R[(size_t)m * p.N + n + e]));
                if (EPI == BG_GELU_POS) v = bf16_round_f32(v + bf16_to_f32(p.R[(size_t)(m % p.pos_rows) * p.N + n + e]));
                res[e] = f32_to_bf16(v);
            }
            bf16_t* o = p.C + (size_t)m * p.N + n;
            if (n + 3 < p.N) {
                uint2 v;
                v.x = (uint32_t)res[0] | ((uint32_t)res[1] << 16);
                v.y = (uint32_t)res[2] | ((uint32_t)res[3] << 16);
                *reinterpret_cast<uint2*>(o) = v;
            } else {
                for (int e = 0; e < 4 && n + e < p.N; ++e) o[e] = res[e];
            }
        }
    }
}

// ---- the same contraction with direct global -> LDS staging (global_load_lds_dwordx4), BK = 64, two LDS buffers, ONE barrier per
// k-step: tile kt+1 is in flight while the 32 MFMAs per wave of tile kt run.  A wave's LDS-DMA writes 64 lanes x 16 B = 1 KiB
// contiguously (lane-linear destination), i.e. 8 rows x 128 B of the [128][64] bf16 tile; bank conflicts of the fragment reads are
// avoided on the SOURCE side: LDS position (row r, 16-B chunk c) holds global chunk c ^ ((r >> 1) & 7), and a fragment read of
// chunk kc of row r goes to position kc ^ ((r >> 1) & 7).  With ds_read_b128's 16-lane service groups ({0-3,12-15,20-27}, ...: every
// group touches the 16 rows of a fragment once, at two neighbouring kc) the 16 lanes of a group land on 16 distinct 16-B slots of
// the 256-B bank row.  Rows past M / N are loaded clamped and masked in the epilogue.
#define BG2_BK 64
template <int EPI>
__global__ void __launch_bounds__(256, 2) k_gemm_big2(BigGemmParams p) {
    __shared__ __attribute__((aligned(1024))) bf16_t Ws[2][BG_BN * BG2_BK];
    __shared__ __attribute__((aligned(1024))) bf16_t Xs[2][BG_BM * BG2_BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * BG_BN, m0 = blockIdx.y * BG_BM;
    const int wn = wave >> 1, wm = wave & 1;            // wave tile: 64 n x 64 m
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // staging: instruction q (0..3) of wave w moves rows (w*4 + q)*8 .. +8 of each operand tile; lane l -> row + (l >> 3), position l & 7
    const bf16_t* wsrc[4];
    const bf16_t* xsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (wave * 4 + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        const int n = min(n0 + r, p.N - 1), m = min(m0 + r, p.M - 1);
        wsrc[q] = p.W + (size_t)n * p.K + chunk * 8;
        xsrc[q] = p.X + (size_t)m * p.ldx + chunk * 8;
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row0 = (wave * 4 + q) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[q] + k0),
                                             (__attribute__((address_space(3))) void*)&Ws[buf][row0 * BG2_BK], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[q] + k0),
                                             (__attribute__((address_space(3))) void*)&Xs[buf][row0 * BG2_BK], 16, 0, 0);
        }
    };
    const int sw = (lane >> 1) & 7;                      // ((row >> 1) & 7) of this lane's fragment rows (row = 16*t + (lane & 15))
    const int KT = p.K / BG2_BK;
    stage(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        __syncthreads();                                 // waits for this wave's LDS-DMA (vmcnt(0)), then the block: tile kt is complete
        const bf16_t* wt = Ws[kt & 1];
        const bf16_t* xt = Xs[kt & 1];
        // all fragment reads of the tile FIRST, then the next tile's DMA: hipcc puts a vmcnt(0) in front of any ds_read that follows an
        // LDS-DMA in program order (it cannot prove the read does not alias the DMA's destination), which would serialise the
        // prefetch with the reads; issued after them, the DMA runs under the 32 MFMAs of this tile
        bf16x8_t a[2][4], b[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pos = ((ks * 4 + (lane >> 4)) ^ sw) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[ks][i] = *reinterpret_cast<const bf16x8_t*>(wt + (wn * 64 + i * 16 + (lane & 15)) * BG2_BK + pos);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[ks][j] = *reinterpret_cast<const bf16x8_t*>(xt + (wm * 64 + j * 16 + (lane & 15)) * BG2_BK + pos);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < KT) stage((kt + 1) & 1, (kt + 1) * BG2_BK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
    }
    // epilogue: as k_gemm_big
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
        if (n >= p.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = (n + e < p.N) ? bf16_to_f32(p.bias[n + e]) : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = m0 + wm * 64 + j * 16 + (lane & 15);
            if (m >= p.M) continue;
            uint16_t res[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = bf16_round_f32(acc[i][j][e] + bv[e]);                       // T(xW^T + b)
                if (EPI == BG_GELU || EPI == BG_GELU_POS) v = bf16_round_f32(gelu_erf_w(v));
                if (EPI == BG_RESID) v = bf16_round_f32(v + bf16_to_f32(p.R[(size_t)m * p.N + n + e]));
                if (EPI == BG_GELU_POS) v = bf16_round_f32(v + bf16_to_f32(p.R[(size_t)(m % p.pos_rows) * p.N + n + e]));
                res[e] = f32_to_bf16(v);
            }
            bf16_t* o = p.C + (size_t)m * p.N + n;
            if (n + 3 < p.N) {
                uint2 v;
                v.x = (uint32_t)res[0] | ((uint32_t)res[1] << 16);
                v.y = (uint32_t)res[2] | ((uint32_t)res[3] << 16);
                *reinterpret_cast<uint2*>(o) = v;
            } else {
                for (int e = 0; e < 4 && n + e < p.N; ++e) o[e] = res[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------- 256 x 256 tiles, eight waves, counted waits
// Round 4 (tools/gemm_lab/big_lab, profiles/r04/big_lab.jsonl): at the encoder's shapes (M = 8 x 1500 rows, N, K in 1280 .. 5120) a
// 256 x 256 x 64 tile on eight waves - one block per CU, two 64 KB LDS buffers - runs the four GEMMs of a layer in 135.7 / 44.6 / 187.6 /
// 158.4 us against 151.2 / 52.0 / 203.9 / 181.5 for k_gemm_big2 (35-40 % of the dense bf16 MFMA peak instead of 31-35 %): each wave's
// fragment reads feed twice the MFMAs (44 FLOP per LDS byte against 32).  The deeper rings of the same laboratory (three buffers at
// 256 x 128 / 128 x 256) were slower.  What makes the two-buffer form work is WHEN the waves wait: the LDS-DMA of tile kt + 1 is issued
// behind the fragment reads of tile kt and waited for with an explicit vmcnt(0) right in front of the next raw s_barrier - with compiler
// loads hipcc puts a vmcnt(0) in front of every ds_read that follows an LDS-DMA in program order, hence asm reads (ds_read_b128, one
// lgkmcnt(0) naming every fragment in front of the MFMAs).  Same k order and accumulators as k_gemm_big2: bit-identical C.
__device__ __forceinline__ uint32_t bg3_lds_offset(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ bf16x8_t bg3_lds_read16(uint32_t addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
#define BG3_BM 256
#define BG3_BN 256
template <int EPI, int WM, int WN>
__global__ void __launch_bounds__(512, 1) k_gemm_big3(BigGemmParams p) {
    constexpr int NW = WM * WN;                                  // 8 waves
    constexpr int TM = BG3_BM / WM / 16, TN = BG3_BN / WN / 16;  // 16 x 16 MFMA tiles per wave (m, n)
    constexpr int QW = BG3_BN / 8 / NW, QX = BG3_BM / 8 / NW;    // LDS-DMA instructions per wave and tile (8 rows of 128 B each)
    static_assert(NW == 8, "eight waves");
    extern __shared__ __attribute__((aligned(1024))) bf16_t bg3_lds[];
    bf16_t* Ws = bg3_lds;                                        // [2][BN * 64]
    bf16_t* Xs = bg3_lds + (size_t)2 * BG3_BN * BG2_BK;          // [2][BM * 64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * BG3_BN, m0 = blockIdx.y * BG3_BM;
    const int wn = wave / WM, wm = wave % WM;
    f32x4_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16_t* wsrc[QW];
    const bf16_t* xsrc[QX];
#pragma unroll
    for (int q = 0; q < QW; ++q) {
        const int r = (wave * QW + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        wsrc[q] = p.W + (size_t)min(n0 + r, p.N - 1) * p.K + chunk * 8;
    }
#pragma unroll
    for (int q = 0; q < QX; ++q) {
        const int r = (wave * QX + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        xsrc[q] = p.X + (size_t)min(m0 + r, p.M - 1) * p.ldx + chunk * 8;
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int q = 0; q < QW; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[q] + k0),
                                             (__attribute__((address_space(3))) void*)&Ws[((size_t)buf * BG3_BN + (wave * QW + q) * 8) * BG2_BK], 16, 0, 0);
#pragma unroll
        for (int q = 0; q < QX; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[q] + k0),
                                             (__attribute__((address_space(3))) void*)&Xs[((size_t)buf * BG3_BM + (wave * QX + q) * 8) * BG2_BK], 16, 0, 0);
    };
    const int KT = p.K / BG2_BK;
    const int sw = (lane >> 1) & 7;
    const uint32_t ws_base = bg3_lds_offset(Ws) + (uint32_t)((wn * (BG3_BN / WN) + (lane & 15)) * BG2_BK * 2);
    const uint32_t xs_base = bg3_lds_offset(Xs) + (uint32_t)((wm * (BG3_BM / WM) + (lane & 15)) * BG2_BK * 2);
    stage(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's part of tile kt has landed
        __builtin_amdgcn_s_barrier();                            // every wave's part is in LDS; the other buffer is free
        asm volatile("" ::: "memory");
        const int buf = kt & 1;
        bf16x8_t a[2][TN], b[2][TM];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint32_t pos = (uint32_t)(((ks * 4 + (lane >> 4)) ^ sw) * 16);
#pragma unroll
            for (int i = 0; i < TN; ++i) a[ks][i] = bg3_lds_read16(ws_base + (uint32_t)((buf * BG3_BN + i * 16) * BG2_BK * 2) + pos);
#pragma unroll
            for (int j = 0; j < TM; ++j) b[ks][j] = bg3_lds_read16(xs_base + (uint32_t)((buf * BG3_BM + j * 16) * BG2_BK * 2) + pos);
        }
        // the next DMA goes into the buffer read one k-step ago; issued behind this tile's reads, it runs under its MFMAs
        if (kt + 1 < KT) stage((kt + 1) & 1, (kt + 1) * BG2_BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TN; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[ks][i]));      // one wait, naming every fragment
#pragma unroll
            for (int j = 0; j < TM; ++j) asm volatile("" : "+v"(b[ks][j]));
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
        }
    }
    // epilogue: as k_gemm_big2 (C/D lane = column m (l & 15), rows n = 4 (l >> 4) + e)
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int n = n0 + wn * (BG3_BN / WN) + i * 16 + (lane >> 4) * 4;
        if (n >= p.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = (n + e < p.N) ? bf16_to_f32(p.bias[n + e]) : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = m0 + wm * (BG3_BM / WM) + j * 16 + (lane & 15);
            if (m >= p.M) continue;
            uint16_t res[4];
            float rv[4] = {0.f, 0.f, 0.f, 0.f};                                     // residual / positional row: ONE 8-byte load (N % 4 == 0)
            if (EPI == BG_RESID || EPI == BG_GELU_POS) {
                const size_t rrow = EPI == BG_RESID ? (size_t)m : (size_t)(m % p.pos_rows);
                const uint2 rq = *reinterpret_cast<const uint2*>(p.R + rrow * p.N + n);
                rv[0] = __uint_as_float(rq.x << 16); rv[1] = __uint_as_float(rq.x & 0xffff0000u);
                rv[2] = __uint_as_float(rq.y << 16); rv[3] = __uint_as_float(rq.y & 0xffff0000u);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = bf16_round_f32(acc[i][j][e] + bv[e]);                       // T(xW^T + b)
                if (EPI == BG_GELU || EPI == BG_GELU_POS) v = bf16_round_f32(gelu_erf_w(v));
                if (EPI == BG_RESID || EPI == BG_GELU_POS) v = bf16_round_f32(v + rv[e]);
                res[e] = f32_to_bf16(v);
            }
            bf16_t* o = p.C + (size_t)m * p.N + n;
            uint2 v;
            v.x = (uint32_t)res[0] | ((uint32_t)res[1] << 16);
            v.y = (uint32_t)res[2] | ((uint32_t)res[3] << 16);
            *reinterpret_cast<uint2*>(o) = v;                                       // (n < N and N % 4 == 0: the four columns exist)
        }
    }
}
template <int EPI, int WM, int WN>
static bool launch_big3_one(const BigGemmParams& p, hipStream_t s) {
    constexpr int lds_bytes = 2 * (BG3_BM + BG3_BN) * BG2_BK * 2;            // 128 KB
    static const bool ok = hipFuncSetAttribute((const void*)k_gemm_big3<EPI, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); return false; }
    hipLaunchKernelGGL((k_gemm_big3<EPI, WM, WN>), dim3(cdiv(p.N, BG3_BN), cdiv(p.M, BG3_BM)), dim3(512), lds_bytes, s, p);
    return true;
}
template <int EPI>
static bool launch_big3(const BigGemmParams& p, hipStream_t s) {
    // wave grid as measured: 2 (m) x 4 (n) for the wide / deep MLP shapes, 4 x 2 otherwise
    return (p.N >= 4096 || p.K >= 4096) ? launch_big3_one<EPI, 2, 4>(p, s) : launch_big3_one<EPI, 4, 2>(p, s);
}

void launch_gemm_big(int epi, const BigGemmParams& p, hipStream_t s) {
    MIS_REQUIRE(p.K % BG_BK == 0 && p.ldx % 8 == 0 && p.N % 4 == 0, MIS_ERR_INVALID_INPUT, "big GEMM needs K %% 32 == 0");
    dim3 grid(cdiv(p.N, BG_BN), cdiv(p.M, BG_BM)), block(256);
    {   // the 256 x 256 tile where its grid still fills most of the chip (out_proj of 8 windows: 235 blocks, 52.0 -> 44.6 us); below that the
        // 128 x 128 kernel has four times the blocks.  MIS_GEMM_BIG3=0: never (A/B); =2: whenever the shape allows (parity tests)
        const char* e3 = getenv("MIS_GEMM_BIG3");
        const int mode = e3 ? atoi(e3) : 1;
        const long blocks3 = (long)cdiv(p.N, BG3_BN) * cdiv(p.M, BG3_BM);
        if (mode != 0 && p.K % BG2_BK == 0 && p.K >= 2 * BG2_BK && (mode == 2 || blocks3 >= 200)) {
            bool done = false;
            switch (epi) {
                case BG_NONE: done = launch_big3<BG_NONE>(p, s); break;
                case BG_GELU: done = launch_big3<BG_GELU>(p, s); break;
                case BG_RESID: done = launch_big3<BG_RESID>(p, s); break;
                case BG_GELU_POS: done = launch_big3<BG_GELU_POS>(p, s); break;
                default: throw MisError(MIS_ERR_GENERATION_FAILED, "unknown big GEMM epilogue");
            }
            if (done) return;
        }
    }
    if (p.K % BG2_BK == 0 && p.K >= 2 * BG2_BK) {
        switch (epi) {
            case BG_NONE: hipLaunchKernelGGL((k_gemm_big2<BG_NONE>), grid, block, 0, s, p); return;
            case BG_GELU: hipLaunchKernelGGL((k_gemm_big2<BG_GELU>), grid, block, 0, s, p); return;
            case BG_RESID: hipLaunchKernelGGL((k_gemm_big2<BG_RESID>), grid, block, 0, s, p); return;
            case BG_GELU_POS: hipLaunchKernelGGL((k_gemm_big2<BG_GELU_POS>), grid, block, 0, s, p); return;
            default: throw MisError(MIS_ERR_GENERATION_FAILED, "unknown big GEMM epilogue");
        }
    }
    switch (epi) {
        case BG_NONE: hipLaunchKernelGGL((k_gemm_big<BG_NONE>), grid, block, 0, s, p); break;
        case BG_GELU: hipLaunchKernelGGL((k_gemm_big<BG_GELU>), grid, block, 0, s, p); break;
        case BG_RESID: hipLaunchKernelGGL((k_gemm_big<BG_RESID>), grid, block, 0, s, p); break;
        case BG_GELU_POS: hipLaunchKernelGGL((k_gemm_big<BG_GELU_POS>), grid, block, 0, s, p); break;
        default: throw MisError(MIS_ERR_GENERATION_FAILED, "unknown big GEMM epilogue");
    }
}

// ============================================================================ LayerNorm over rows
__global__ void __launch_bounds__(256) k_layernorm(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                   const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias, int d,
                                                   float eps) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const bf16_t* xr = x + row * d;
    float s = 0.0f;
    for (int i = threadIdx.x; i < d; i += 256) s += bf16_to_f32(xr[i]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)d;
    __syncthreads();
    float q = 0.0f;
    for (int i = threadIdx.x; i < d; i += 256) { float t = bf16_to_f32(xr[i]) - mean; q += t * t; }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)d + eps);
    for (int i = threadIdx.x; i < d; i += 256)
        y[row * d + i] = f32_to_bf16((bf16_to_f32(xr[i]) - mean) * rstd * bf16_to_f32(w[i]) + bf16_to_f32(bias[i]));
}
// d % 8 == 0, d <= 2048: the row lives in registers (one 16-byte piece per thread, read once), two block sums, 16-byte stores - the
// element-wise kernel above reads the row three times with 2-byte loads (32.6 us per 12 000 x 1280 call, 1.9 TB/s)
__global__ void __launch_bounds__(256) k_layernorm8(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, const bf16_t* __restrict__ w,
                                                    const bf16_t* __restrict__ bias, int d, float eps) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const int tid = threadIdx.x, nch = d >> 3;
    const bool live = tid < nch;
    const int ch = live ? tid : 0;
    const uint4 xq = reinterpret_cast<const uint4*>(x + row * d)[ch];
    const uint4 wq = reinterpret_cast<const uint4*>(w)[ch];
    const uint4 bq = reinterpret_cast<const uint4*>(bias)[ch];
    const uint32_t xw[4] = {xq.x, xq.y, xq.z, xq.w}, ww[4] = {wq.x, wq.y, wq.z, wq.w}, bw[4] = {bq.x, bq.y, bq.z, bq.w};
    float v[8], s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = bf16_to_f32((bf16_t)(xw[j] & 0xffffu)); v[2 * j + 1] = bf16_to_f32((bf16_t)(xw[j] >> 16));
        s += live ? v[2 * j] + v[2 * j + 1] : 0.0f;
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)d;
    __syncthreads();
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float t = v[e] - mean; q += live ? t * t : 0.0f; }
    q = wave_sum(q);
    if ((tid & 63) == 0) red[tid >> 6] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)d + eps);
    if (live) {
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float w0 = bf16_to_f32((bf16_t)(ww[j] & 0xffffu)), w1 = bf16_to_f32((bf16_t)(ww[j] >> 16));
            const float b0 = bf16_to_f32((bf16_t)(bw[j] & 0xffffu)), b1 = bf16_to_f32((bf16_t)(bw[j] >> 16));
            o[j] = (uint32_t)f32_to_bf16((v[2 * j] - mean) * rstd * w0 + b0) | ((uint32_t)f32_to_bf16((v[2 * j + 1] - mean) * rstd * w1 + b1) << 16);
        }
        reinterpret_cast<uint4*>(y + row * d)[ch] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
void launch_layernorm(const bf16_t* x, bf16_t* y, const bf16_t* w, const bf16_t* b, int rows, int d, float eps, hipStream_t s) {
    if (rows <= 0) return;
    const bool a16 = ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)w) | ((uintptr_t)b)) & 15) == 0;
    if (d % 8 == 0 && d <= 2048 && a16) hipLaunchKernelGGL(k_layernorm8, dim3(rows), dim3(256), 0, s, x, y, w, b, d, eps);
    else hipLaunchKernelGGL(k_layernorm, dim3(rows), dim3(256), 0, s, x, y, w, b, d, eps);
}

// ============================================================================ conv stem patches
// out[(b*Tout + t)][k*C + c] = in[b][t*stride + k - 1][c]  (zero outside), k = 0..2   (Conv1d k3 p1, NLC)
template <typename TIN>
__global__ void k_im2col3(const TIN* __restrict__ in, bf16_t* __restrict__ out, int B, int Tin, int C, int Tout, int stride) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)B * Tout * 3 * C;
    if (i >= total) return;
    int c = (int)(i % C);
    size_t r = i / C;
    int k = (int)(r % 3);
    r /= 3;
    int t = (int)(r % Tout);
    int b = (int)(r / Tout);
    int ti = t * stride + k - 1;
    float v = 0.0f;
    if (ti >= 0 && ti < Tin) {
        TIN raw = in[((size_t)b * Tin + ti) * C + c];
        if constexpr (sizeof(TIN) == 4) v = (float)raw; else v = bf16_to_f32((bf16_t)raw);
    }
    out[i] = f32_to_bf16(v);
}
void launch_im2col3_f32(const float* in, bf16_t* out, int B, int Tin, int C, int Tout, int stride, hipStream_t s) {
    size_t total = (size_t)B * Tout * 3 * C;
    hipLaunchKernelGGL((k_im2col3<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, B, Tin, C, Tout, stride);
}
void launch_im2col3_bf16(const bf16_t* in, bf16_t* out, int B, int Tin, int C, int Tout, int stride, hipStream_t s) {
    size_t total = (size_t)B * Tout * 3 * C;
    hipLaunchKernelGGL((k_im2col3<bf16_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, B, Tin, C, Tout, stride);
}

// ============================================================================ K / V -> tiled fragment caches
// src rows [B*T][ld]; K columns at kcol0 + h*D + d, V columns at vcol0 + h*D + d.
// kcache [B][H][Spad/32][2][D/32][64][8],  vcache [B][H][Spad/32][D/16][64][8]   (layouts of lm_kernels.hip)
// One block per (32-key tile, head, batch row): the tile's K and V rows (32 x D, read as 16-byte pieces of the source rows) go
// through LDS and leave as the tile's two contiguous 32*D*2-byte images, written with 16-byte stores - the element-wise version
// (2-byte stores, the V image strided by 16 B between neighbouring threads) moved 122 MB per large-v3 layer at 1.5 TB/s.
template <int D>
__global__ void __launch_bounds__(256) k_scatter_kv(const bf16_t* __restrict__ src, int ld, int kcol0, int vcol0, bf16_t* __restrict__ kc,
                                                    bf16_t* __restrict__ vc, int T, int H, int Spad) {
    __shared__ __attribute__((aligned(16))) bf16_t ks[32][D + 8];          // +8: rows 16 B apart in bank space
    __shared__ __attribute__((aligned(16))) bf16_t vs[32][D + 8];
    const int tile = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    constexpr int CH = D / 8;                                                // 16-byte pieces per row
    for (int i = tid; i < 32 * CH; i += 256) {
        const int r = i / CH, c = i - r * CH;
        const int t = tile * 32 + r;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = kv;
        if (t < T) {
            const bf16_t* row = src + ((size_t)b * T + t) * ld;
            kv = *reinterpret_cast<const uint4*>(row + kcol0 + h * D + c * 8);
            vv = *reinterpret_cast<const uint4*>(row + vcol0 + h * D + c * 8);
        }
        *reinterpret_cast<uint4*>(&ks[r][c * 8]) = kv;
        *reinterpret_cast<uint4*>(&vs[r][c * 8]) = vv;
    }
    __syncthreads();
    const size_t base = ((size_t)b * H + h) * (size_t)Spad * D + (size_t)tile * 32 * D;
    // K image [2][D/32][64][8]: lane = q*16 + prow holds key pr (prow = ((pr>>3)<<2)|(pr&3), half = (pr>>2)&1), dims dc*32 + q*8 .. +8
    for (int i = tid; i < 2 * (D / 32) * 64; i += 256) {
        const int lane = i & 63, dc = (i >> 6) % (D / 32), half = i / (64 * (D / 32));
        const int prow = lane & 15, q = lane >> 4;
        const int pr = ((prow >> 2) << 3) | (half << 2) | (prow & 3);
        *reinterpret_cast<uint4*>(kc + base + (size_t)i * 8) = *reinterpret_cast<const uint4*>(&ks[pr][dc * 32 + q * 8]);
    }
    // V image [D/16][64][8]: lane = (pr>>3)*16 + (d&15) holds keys (pr>>3)*8 .. +8 of dim dt*16 + (d&15)
    for (int i = tid; i < (D / 16) * 64; i += 256) {
        const int lane = i & 63, dt = i >> 6;
        const int d = dt * 16 + (lane & 15), p0 = (lane >> 4) * 8;
        bf16_t v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v8[e] = vs[p0 + e][d];
        *reinterpret_cast<uint4*>(vc + base + (size_t)i * 8) = *reinterpret_cast<const uint4*>(v8);
    }
}
void launch_scatter_kv(const bf16_t* src, int ld, int kcol0, int vcol0, bf16_t* kc, bf16_t* vc, int B, int T, int H, int D,
                       int Spad, hipStream_t s) {
    MIS_REQUIRE((D == 64 || D == 128) && ld % 8 == 0 && kcol0 % 8 == 0 && vcol0 % 8 == 0 && Spad % 32 == 0, MIS_ERR_INVALID_INPUT,
                "K/V scatter: unsupported layout");
    dim3 grid(cdiv(T, 32), H, B);
    if (D == 64) hipLaunchKernelGGL((k_scatter_kv<64>), grid, dim3(256), 0, s, src, ld, kcol0, vcol0, kc, vc, T, H, Spad);
    else hipLaunchKernelGGL((k_scatter_kv<128>), grid, dim3(256), 0, s, src, ld, kcol0, vcol0, kc, vc, T, H, Spad);
}

// ============================================================================ encoder self attention (non causal)
// grid (ceil(T/64), H, B), 4 waves x 16 query rows.  S^T = K Q^T (A = K fragment, B = Q^T): lane (q row j, group g4)
// holds the scores of keys base + g4*8 + e; online softmax per query row; O += P V with P split into bf16 hi + lo.
template <int D>
__global__ void __launch_bounds__(256) k_attn_prefill(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kc,
                                                      const bf16_t* __restrict__ vc, bf16_t* __restrict__ out, int ldo,
                                                      int T, int H, int Spad, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, h = blockIdx.y, t0 = blockIdx.x * 64 + wave * 16;
    if (t0 >= T) return;
    const int j = lane & 15, g4 = lane >> 4;
    const int tq = t0 + j;
    bf16x8_t qf[D / 32];
#pragma unroll
    for (int c = 0; c < D / 32; ++c) {
        if (tq < T) qf[c] = *reinterpret_cast<const bf16x8_t*>(q + ((size_t)b * T + tq) * ldq + h * D + c * 32 + g4 * 8);
        else qf[c] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
    }
    const size_t base = ((size_t)b * H + h) * (size_t)Spad * D;
    const bf16x8_t* kbase = reinterpret_cast<const bf16x8_t*>(kc + base) + lane;
    const bf16x8_t* vbase = reinterpret_cast<const bf16x8_t*>(vc + base) + lane;
    f32x4_t O[D / 16];
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) O[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.0f;
    const int n_tiles = (T + 31) >> 5;
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int kb = tile * 32;
        f32x4_t S0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, S1 = S0;
#pragma unroll
        for (int c = 0; c < D / 32; ++c) {
            bf16x8_t a0 = kbase[((size_t)tile * 2) * (D / 32) * 64 + c * 64];
            bf16x8_t a1 = kbase[((size_t)tile * 2 + 1) * (D / 32) * 64 + c * 64];
            S0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[c], S0, 0, 0, 0);
            S1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[c], S1, 0, 0, 0);
        }
        float sc[8], mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = (e < 4 ? S0[e] : S1[e - 4]) * scale;
            v = (kb + g4 * 8 + e < T) ? v : -INFINITY;
            sc[e] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float m_new = fmaxf(m_run, mx);
        float alpha = __expf(m_run - m_new);
        float psum = 0.0f;
        bf16x8_t ph, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float pe = __expf(sc[e] - m_new);
            psum += pe;
            bf16_t hi = f32_to_bf16(pe);
            bf16_t lo = f32_to_bf16(pe - bf16_to_f32(hi));
            ph[e] = (short)hi;
            pl[e] = (short)lo;
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        float ar[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, g4 * 4 + r, 64);
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) {
            bf16x8_t vb = vbase[((size_t)tile * (D / 16) + dt) * 64];
            f32x4_t o = O[dt];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] *= ar[r];
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vb, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, vb, o, 0, 0, 0);
            O[dt] = o;
        }
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    // O[dt][r]: row = query (g4*4 + r), col = d (dt*16 + j); the row's normaliser lives in lane (g4*4 + r)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float lr = __shfl(l_run, g4 * 4 + r, 64);
        int tr = t0 + g4 * 4 + r;
        if (tr < T) {
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt)
                out[((size_t)b * T + tr) * ldo + h * D + dt * 16 + j] = f32_to_bf16(O[dt][r] / lr);
        }
    }
}
// ---- the same attention, tiled for the matrix core's appetite (round 6; VERDICT r03-r05: k_attn_prefill<64> at MfmaUtil 0.21 was the worst
// MFMA kernel of the tree).  What the kernel above costs per 32-key tile and wave: 16 fragment loads straight from global memory, waited
// for where they are used (no tile in flight while another is multiplied), for 12 MFMAs on 16 query rows - and every one of the 94
// waves of a head streams the head's whole 393 KB of K/V for itself.  Here a block is 256 query rows (8 waves x 2 groups of 16): the
// tile's K and V images (8 KB, already in fragment order - lm_kernels.hip's cache tiling) are staged ONCE per block through LDS, one
// 16-byte piece per thread, tile t + 1 requested before tile t is multiplied (two LDS buffers, one barrier per tile); a wave reads each
// fragment once (lane-linear ds_read_b128, conflict-free) and uses it for both of its row groups: 24 MFMAs per 8 LDS reads.  The key
// mask is applied on the last tile only, and the running output is rescaled only when some row's maximum moved (alpha = 1 otherwise:
// skipping the multiply is exact).  Same operations per query row in the same order as k_attn_prefill: results are bit-identical
// (tests/test_gpu_whisper.py).  Bound after this: the softmax's VALU work (exp, the bf16 hi/lo split of P), not the matrix core.
template <int D>
__global__ void __launch_bounds__(512, 1) k_attn_prefill2(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kc,
                                                          const bf16_t* __restrict__ vc, bf16_t* __restrict__ out, int ldo,
                                                          int T, int H, int Spad, float scale) {
    static_assert(D == 64, "one 16-byte piece of the tile's 8 KB per thread");
    constexpr int NW = 8, QR = 32 * NW, IMG = 32 * D;                // bf16 elements of one K (or V) tile image
    __shared__ __attribute__((aligned(16))) bf16_t kv[2][2 * IMG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QR + wave * 32;
    const int j = lane & 15, g4 = lane >> 4;
    bf16x8_t qf[2][D / 32];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int tq = q0 + 16 * g + j;
#pragma unroll
        for (int c = 0; c < D / 32; ++c) {
            if (tq < T) qf[g][c] = *reinterpret_cast<const bf16x8_t*>(q + ((size_t)b * T + tq) * ldq + h * D + c * 32 + g4 * 8);
            else qf[g][c] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    const size_t base = ((size_t)b * H + h) * (size_t)Spad * D;
    // staging: threads 0 .. 255 move the K image, 256 .. 511 the V image (IMG / 8 = 256 pieces each)
    const uint4* src = reinterpret_cast<const uint4*>((tid < 256 ? kc : vc) + base) + (tid & 255);
    const int dst_piece = tid;                                            // K image first, V image behind it
    const int n_tiles = (T + 31) >> 5;
    uint4 st = src[0];
    *(reinterpret_cast<uint4*>(kv[0]) + dst_piece) = st;
    __syncthreads();
    f32x4_t O[2][D / 16];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) O[g][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.0f, 0.0f};
    auto tile_math = [&](int tile, const bf16_t* img, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int kb = tile * 32;
        const bf16x8_t* kf = reinterpret_cast<const bf16x8_t*>(img) + lane;
        const bf16x8_t* vf = reinterpret_cast<const bf16x8_t*>(img + IMG) + lane;
        bf16x8_t a0[D / 32], a1[D / 32], vb[D / 16];
#pragma unroll
        for (int c = 0; c < D / 32; ++c) { a0[c] = kf[c * 64]; a1[c] = kf[((D / 32) + c) * 64]; }
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) vb[dt] = vf[dt * 64];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4_t S0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, S1 = S0;
#pragma unroll
            for (int c = 0; c < D / 32; ++c) {
                S0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[c], qf[g][c], S0, 0, 0, 0);
                S1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[c], qf[g][c], S1, 0, 0, 0);
            }
            float sc[8], mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = (e < 4 ? S0[e] : S1[e - 4]) * scale;
                if constexpr (MASKED) v = (kb + g4 * 8 + e < T) ? v : -INFINITY;
                sc[e] = v;
                mx = fmaxf(mx, v);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[g], mx);
            const float alpha = __expf(m_run[g] - m_new);
            float psum = 0.0f;
            bf16x8_t ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pe = __expf(sc[e] - m_new);
                psum += pe;
                const bf16_t hi = f32_to_bf16(pe);
                const bf16_t lo = f32_to_bf16(pe - bf16_to_f32(hi));
                ph[e] = (short)hi;
                pl[e] = (short)lo;
            }
            l_run[g] = l_run[g] * alpha + psum;
            const bool moved = m_new != m_run[g];
            m_run[g] = m_new;
            if (__any(moved)) {                              // (wave-uniform; alpha = exp(0) = 1 for every row otherwise)
                float ar[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, g4 * 4 + r, 64);
#pragma unroll
                for (int dt = 0; dt < D / 16; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) O[g][dt][r] *= ar[r];
            }
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                f32x4_t o = O[g][dt];
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vb[dt], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, vb[dt], o, 0, 0, 0);
                O[g][dt] = o;
            }
        }
    };
    for (int tile = 0; tile < n_tiles; ++tile) {
        const bool more = tile + 1 < n_tiles;
        if (more) st = src[(size_t)(tile + 1) * (IMG / 8)];
        if (more || (T & 31) == 0) tile_math(tile, kv[tile & 1], std::false_type{});
        else tile_math(tile, kv[tile & 1], std::true_type{});
        if (more) *(reinterpret_cast<uint4*>(kv[(tile + 1) & 1]) + dst_piece) = st;
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float lr_all = l_run[g];
        lr_all += __shfl_xor(lr_all, 16, 64);
        lr_all += __shfl_xor(lr_all, 32, 64);
        // O[g][dt][r]: row = query (g4*4 + r), col = d (dt*16 + j); the row's normaliser lives in lane (g4*4 + r)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lr = __shfl(lr_all, g4 * 4 + r, 64);
            const int tr = q0 + 16 * g + g4 * 4 + r;
            if (tr < T) {
#pragma unroll
                for (int dt = 0; dt < D / 16; ++dt)
                    out[((size_t)b * T + tr) * ldo + h * D + dt * 16 + j] = f32_to_bf16(O[g][dt][r] / lr);
            }
        }
    }
}
void launch_attn_prefill(const bf16_t* q, int ldq, const bf16_t* kc, const bf16_t* vc, bf16_t* out, int ldo, int B, int T,
                         int H, int D, int Spad, hipStream_t s) {
    float scale = 1.0f / sqrtf((float)D);
    const char* ve = getenv("MIS_ATTN_PREFILL_V1");                      // tests: the 16-rows-per-wave kernel (read per launch)
    const bool v1 = ve && atoi(ve) != 0;
    if (D == 64 && !v1 && ((uintptr_t)kc & 15) == 0 && ((uintptr_t)vc & 15) == 0) {
        hipLaunchKernelGGL((k_attn_prefill2<64>), dim3(cdiv(T, 256), H, B), dim3(512), 0, s, q, ldq, kc, vc, out, ldo, T, H, Spad, scale);
        return;
    }
    dim3 grid(cdiv(T, 64), H, B), block(256);
    if (D == 64) hipLaunchKernelGGL((k_attn_prefill<64>), grid, block, 0, s, q, ldq, kc, vc, out, ldo, T, H, Spad, scale);
    else if (D == 128) hipLaunchKernelGGL((k_attn_prefill<128>), grid, block, 0, s, q, ldq, kc, vc, out, ldo, T, H, Spad, scale);
    else throw MisError(MIS_ERR_INVALID_INPUT, "head_dim must be 64 or 128");
}

// ============================================================================ decoder step helpers
// h = T(E[tok] + P[pos]) ; x = LayerNorm(h) packed ; advances positions like k_embed_rmsnorm
__global__ void __launch_bounds__(256) k_whisper_embed_ln(const bf16_t* __restrict__ emb, const bf16_t* __restrict__ pos_emb,
                                                          const int32_t* __restrict__ ids, const uint8_t* __restrict__ active,
                                                          int* __restrict__ pos_cur, int* __restrict__ pos_next,
                                                          const bf16_t* __restrict__ lw, const bf16_t* __restrict__ lb,
                                                          bf16_t* __restrict__ h, bf16_t* __restrict__ x, int d, int vocab,
                                                          int max_pos, int batch) {
    __shared__ float red[4];
    const int m = blockIdx.x, MT = gridDim.x >> 4;
    int id = (m < batch) ? ids[m] : 0;
    if (id < 0 || id >= vocab) id = 0;
    int p = (m < batch) ? pos_next[m] : 0;
    if (threadIdx.x == 0 && m < batch) {
        pos_cur[m] = p;
        if (active[m]) pos_next[m] = p + 1;
    }
    if (p >= max_pos) p = max_pos - 1;
    float s = 0.0f;
    for (int i = threadIdx.x; i < d; i += 256) {
        float v = bf16_round_f32(bf16_to_f32(emb[(size_t)id * d + i]) + bf16_to_f32(pos_emb[(size_t)p * d + i]));
        h[(size_t)m * d + i] = f32_to_bf16(v);
        s += v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)d;
    __syncthreads();
    float q = 0.0f;
    for (int i = threadIdx.x; i < d; i += 256) { float t = bf16_to_f32(h[(size_t)m * d + i]) - mean; q += t * t; }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)d + 1e-5f);
    for (int i = threadIdx.x; i < d; i += 256) {
        float f = bf16_to_f32(h[(size_t)m * d + i]);
        size_t off = ((((size_t)(i >> 5) * MT + (m >> 4)) * 64) + (((i & 31) >> 3) << 4) + (m & 15)) * 8 + (i & 7);
        x[off] = f32_to_bf16((f - mean) * rstd * bf16_to_f32(lw[i]) + bf16_to_f32(lb[i]));
    }
}
void launch_whisper_embed_ln(const bf16_t* emb, const bf16_t* pos_emb, const int32_t* ids, const uint8_t* active, int* pos_cur,
                             int* pos_next, const bf16_t* lw, const bf16_t* lb, bf16_t* h, bf16_t* x, int d, int vocab,
                             int max_pos, int batch, int Mpad, hipStream_t s) {
    hipLaunchKernelGGL(k_whisper_embed_ln, dim3(Mpad), dim3(256), 0, s, emb, pos_emb, ids, active, pos_cur, pos_next, lw, lb, h,
                       x, d, vocab, max_pos, batch);
}

// suppress masks of the greedy loop (WhisperModel.swift:228-236,293-309): logits += -1e9 on the listed ids
// (begin list only while the row has generated nothing); ids >= timestamp_begin are excluded via the sampler range.
__global__ void k_whisper_suppress(bf16_t* __restrict__ logits, int Vpad, int vocab, const int32_t* __restrict__ sup, int n_sup,
                                   const int32_t* __restrict__ bsup, int n_bsup, const int32_t* __restrict__ n_gen,
                                   const uint8_t* __restrict__ active) {
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    bf16_t* l = logits + (size_t)b * Vpad;
    for (int i = threadIdx.x; i < n_sup; i += blockDim.x) {
        int id = sup[i];
        if (id >= 0 && id < vocab) l[id] = f32_to_bf16(bf16_to_f32(l[id]) + -1e9f);
    }
    if (n_gen[b] == 0)
        for (int i = threadIdx.x; i < n_bsup; i += blockDim.x) {
            int id = bsup[i];
            bool dup = false;
            for (int k = 0; k < n_sup; ++k) dup = dup || (sup[k] == id);
            if (id >= 0 && id < vocab && !dup) l[id] = f32_to_bf16(bf16_to_f32(l[id]) + -1e9f);
        }
}
void launch_whisper_suppress(bf16_t* logits, int Vpad, int vocab, const int32_t* sup, int n_sup, const int32_t* bsup, int n_bsup,
                             const int32_t* n_gen, const uint8_t* active, int batch, hipStream_t s) {
    if (n_sup + n_bsup == 0) return;
    hipLaunchKernelGGL(k_whisper_suppress, dim3(batch), dim3(128), 0, s, logits, Vpad, vocab, sup, n_sup, bsup, n_bsup, n_gen, active);
}
