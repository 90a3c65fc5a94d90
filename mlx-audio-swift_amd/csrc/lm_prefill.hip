// lm_prefill.hip - the prompt as ONE [positions x rows] pass: compute-bound MFMA GEMMs over the packed LM weights.
//
// Reference being replaced: the prefill call of the generate loop, `model(inputIds [1, L], cache:)` (LlamaTTS.swift:711), which runs
// the whole prompt through every layer at once.  The decode engine used to feed the prompt one position per step (L weight-streaming
// passes: 56 ms of the 1.6 s bench job for 32-token prompts, and linear in L - voice-cloning prompts carry hundreds of audio tokens).
// Here all L x B prompt tokens form the M dimension of four GEMMs per layer (M = 1024 at the bench shape), each weight byte is read
// once per CHUNK of positions instead of once per position, and only the attention walks the positions one by one (the decode
// attention kernel itself: it appends the position's key / value and attends causally, so caches, RoPE and q/k-norm are exactly the
// sequential path's).  Rows are laid out position-major: row r = t * Mpad + b.
//
//   k_pf_embed_rmsnorm  h[r] = E[token(b, t)], x[r] = RMSNorm(h[r]) (row-major), active / position tables of every (t, b)
//   k_gemm_pf<EPI>      C[M][N] = X[M][K] . W^T with W in the decode path's packed MFMA-A tile layout [N/16][K/32][64][8]:
//                       a weight fragment is one 1 KiB global_load_lds piece and its LDS image IS the A fragment (lane-linear:
//                       ds_read_b128 at lane*16, conflict free); X tiles as in k_gemm_big2 (whisper_kernels.hip: source-side
//                       XOR swizzle).  128 x 128 x 64 tiles, two LDS buffers, one barrier per k-step.  Epilogues:
//                         PF_F32    float32 [M][N]                       (q|k|v: the attention prologue reads it as a 1-slab split-K sum)
//                         PF_RESID  h = T(h + T(acc)) in place, bf16      (o_proj, down_proj: LlamaTTS.swift:306-309)
//                         PF_SILU   act = T(T(g T(sigmoid g)) u), gate / up tiles interleaved as packed (:283)
//   k_pf_rmsnorm        x[r] = RMSNorm(h[r]) row-major
//   k_pf_pack_rows      rows of the last position -> the packed x operand of the decode step (lm_head of the first new token)
// Rounding points are those of the decode step (bf16 at every MLX primitive boundary); only float32 summation orders differ.
#include "common.h"
#include "lm_kernels.h"

#define PF_BM 128
#define PF_BN 128
#define PF_BK 64

__device__ __forceinline__ float pf_block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    const float t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
}

// one 256-thread block per row r = t*Mpad + b of the chunk [t0, t0 + Tc)
__global__ void __launch_bounds__(256) k_pf_embed_rmsnorm(const bf16_t* __restrict__ emb, const int32_t* __restrict__ prompt /*[batch][Lmax] left padded*/,
                                                          const int32_t* __restrict__ lens, int Lmax, int t0, int batch, int Mpad, int vocab,
                                                          const bf16_t* __restrict__ wnorm, bf16_t* __restrict__ h, bf16_t* __restrict__ x,
                                                          int32_t* __restrict__ pos_tab, uint8_t* __restrict__ act_tab, int d, float eps) {
    __shared__ float red[4];
    const int r = blockIdx.x, tl = r / Mpad, b = r - tl * Mpad, t = t0 + tl;
    const bool row = b < batch;
    const int len = row ? lens[b] : 0;
    const bool on = row && t >= Lmax - len;
    if (threadIdx.x == 0) { pos_tab[r] = on ? t - (Lmax - len) : 0; act_tab[r] = on ? 1 : 0; }
    int id = on ? prompt[(size_t)b * Lmax + t] : -1;
    if (id >= vocab) id = -1;
    const bf16_t* e = emb + (size_t)(id < 0 ? 0 : id) * d;
    float ss = 0.0f;
    for (int i = threadIdx.x; i < d; i += 256) {
        const bf16_t v = id < 0 ? (bf16_t)0 : e[i];
        h[(size_t)r * d + i] = v;
        const float f = bf16_to_f32(v);
        ss += f * f;
    }
    const float inv = 1.0f / sqrtf(pf_block_sum_256(ss, red) / (float)d + eps);
    for (int i = threadIdx.x; i < d; i += 256) {
        const float f = id < 0 ? 0.0f : bf16_to_f32(e[i]);
        x[(size_t)r * d + i] = f32_to_bf16(bf16_to_f32(wnorm[i]) * bf16_round_f32(f * inv));       // T(w * T(x * rsqrt(mean + eps)))
    }
}
// the same for engines that feed input embeddings instead of token ids (Qwen3-TTS talker): rows [Lmax][src_rows][d] computed by the
// caller; `Mpad` here and in k_pf_embed_rmsnorm is the chunk's rows per position (the engine's Mpad, or the batch itself)
__global__ void __launch_bounds__(256) k_pf_rows_rmsnorm(const bf16_t* __restrict__ rows, int src_rows, const int32_t* __restrict__ lens, int Lmax, int t0, int batch,
                                                         int Mpad, const bf16_t* __restrict__ wnorm, bf16_t* __restrict__ h, bf16_t* __restrict__ x,
                                                         int32_t* __restrict__ pos_tab, uint8_t* __restrict__ act_tab, int d, float eps) {
    __shared__ float red[4];
    const int r = blockIdx.x, tl = r / Mpad, b = r - tl * Mpad, t = t0 + tl;
    const bool row = b < batch;
    const int len = row ? lens[b] : 0;
    const bool on = row && t >= Lmax - len;
    if (threadIdx.x == 0) { pos_tab[r] = on ? t - (Lmax - len) : 0; act_tab[r] = on ? 1 : 0; }
    const bf16_t* e = rows + ((size_t)t * src_rows + b) * d;                   // the caller's matrix keeps its own rows per position
    float ss = 0.0f;
    for (int i = threadIdx.x; i < d; i += 256) {
        const bf16_t v = on ? e[i] : (bf16_t)0;
        h[(size_t)r * d + i] = v;
        const float f = bf16_to_f32(v);
        ss += f * f;
    }
    const float inv = 1.0f / sqrtf(pf_block_sum_256(ss, red) / (float)d + eps);
    for (int i = threadIdx.x; i < d; i += 256) {
        const float f = on ? bf16_to_f32(e[i]) : 0.0f;
        x[(size_t)r * d + i] = f32_to_bf16(bf16_to_f32(wnorm[i]) * bf16_round_f32(f * inv));
    }
}
__global__ void __launch_bounds__(256) k_pf_rmsnorm(const bf16_t* __restrict__ h, const bf16_t* __restrict__ wnorm, bf16_t* __restrict__ x,
                                                    int d, float eps) {
    __shared__ float red[4];
    const size_t r = blockIdx.x;
    float ss = 0.0f;
    for (int i = threadIdx.x; i < d; i += 256) { const float f = bf16_to_f32(h[r * d + i]); ss += f * f; }
    const float inv = 1.0f / sqrtf(pf_block_sum_256(ss, red) / (float)d + eps);
    for (int i = threadIdx.x; i < d; i += 256)
        x[r * d + i] = f32_to_bf16(bf16_to_f32(wnorm[i]) * bf16_round_f32(bf16_to_f32(h[r * d + i]) * inv));
}
// rows [Mpad][d] row-major -> packed MFMA-B fragments of the decode step's x operand
__global__ void k_pf_pack_rows(const bf16_t* __restrict__ rows, bf16_t* __restrict__ xpk, int d, int MT) {
    const int m = blockIdx.x;
    for (int k = threadIdx.x; k < d; k += blockDim.x) xpk[xpk_index(m, k, MT)] = rows[(size_t)m * d + k];
}

struct PfGemmParams {
    const bf16_t* X;      // [M][K] row-major
    const bf16_t* Wp;     // packed [NT][KT][64][8]
    void* C;              // PF_F32: float [M][N]; PF_RESID: bf16 h [M][N] (read-modify-write); PF_SILU: bf16 act [M][N/2]
    int M, N, K;          // N = rows of W (PF_SILU: 2 * ff, gate / up tiles interleaved)
};

template <int EPI>
__global__ void __launch_bounds__(256, 2) k_gemm_pf(PfGemmParams p) {
    __shared__ __attribute__((aligned(1024))) bf16_t Ws[2][PF_BN * PF_BK];
    __shared__ __attribute__((aligned(1024))) bf16_t Xs[2][PF_BM * PF_BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * PF_BN, m0 = blockIdx.y * PF_BM;
    const int wn = wave >> 1, wm = wave & 1;            // wave tile: 64 n x 64 m
    const int NT = p.N >> 4, KT = p.K >> 5;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // staging.  W: fragment f = wave*4 + q of the 8 n-tiles x 2 k-tiles of a k-step, 1 KiB each, lane-linear both in memory and in LDS.
    // X: piece q of wave w = rows (w*4 + q)*8 .. +8, lane l -> row + (l >> 3), 16-byte position l & 7 holding chunk (l & 7) ^ ((row >> 1) & 7)
    const bf16_t* wsrc[4];
    const bf16_t* xsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f = wave * 4 + q, ntl = f >> 1, ktl = f & 1;
        const int nt = min((n0 >> 4) + ntl, NT - 1);
        wsrc[q] = p.Wp + ((size_t)nt * KT + ktl) * 512 + lane * 8;
        const int r = (wave * 4 + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        xsrc[q] = p.X + (size_t)min(m0 + r, p.M - 1) * p.K + chunk * 8;
    }
    auto stage = [&](int buf, int ks) {                  // k-step ks covers k-tiles 2 ks, 2 ks + 1 = columns 64 ks .. +64
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = wave * 4 + q;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[q] + (size_t)ks * 1024),
                                             (__attribute__((address_space(3))) void*)&Ws[buf][f * 512], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[q] + ks * PF_BK),
                                             (__attribute__((address_space(3))) void*)&Xs[buf][(wave * 4 + q) * 8 * PF_BK], 16, 0, 0);
        }
    };
    const int sw = (lane >> 1) & 7;
    const int KS = p.K / PF_BK;
    stage(0, 0);
    for (int ks = 0; ks < KS; ++ks) {
        __syncthreads();                                 // this wave's LDS-DMA has landed (vmcnt(0)), then everybody's: k-step ks complete
        const bf16_t* wt = Ws[ks & 1];
        const bf16_t* xt = Xs[ks & 1];
        bf16x8_t a[2][4], b[2][4];                       // all fragment reads first, then the next DMA (see k_gemm_big2)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int pos = ((kk * 4 + (lane >> 4)) ^ sw) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) a[kk][i] = *reinterpret_cast<const bf16x8_t*>(wt + ((wn * 4 + i) * 2 + kk) * 512 + lane * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[kk][j] = *reinterpret_cast<const bf16x8_t*>(xt + (wm * 64 + j * 16 + (lane & 15)) * PF_BK + pos);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < KS) stage((ks + 1) & 1, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
    }
    // epilogue.  C/D lane l, reg e: W row (n) = 16*tile + (l >> 4)*4 + e, X row (m) = l & 15
    const int nl = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + (lane & 15);
        if (m >= p.M) continue;
        if (EPI == PF_SILU) {
            bf16_t* act = reinterpret_cast<bf16_t*>(p.C);
            const int ff = p.N >> 1;
#pragma unroll
            for (int i = 0; i < 4; i += 2) {             // tile pair (gate, up) -> feature tile
                const int tile = (n0 >> 4) + wn * 4 + i;
                if (tile + 1 >= NT) continue;             // (tile, tile + 1) = (gate, up) of feature tile `tile >> 1`
                uint16_t res[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = bf16_round_f32(acc[i][j][e]);
                    const float u = bf16_round_f32(acc[i + 1][j][e]);
                    const float sg = bf16_round_f32(1.0f / (1.0f + __expf(-g)));
                    const float av = bf16_round_f32(g * sg);
                    res[e] = f32_to_bf16(av * u);
                }
                uint2 v;
                v.x = (uint32_t)res[0] | ((uint32_t)res[1] << 16);
                v.y = (uint32_t)res[2] | ((uint32_t)res[3] << 16);
                *reinterpret_cast<uint2*>(act + (size_t)m * ff + (tile >> 1) * 16 + nl) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tile = (n0 >> 4) + wn * 4 + i;
                if (tile >= NT) continue;
                const int n = tile * 16 + nl;
                if (EPI == PF_F32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.N + n) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                } else {
                    bf16_t* h = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.N + n;
                    const uint2 old = *reinterpret_cast<const uint2*>(h);
                    const uint32_t ow[2] = {old.x, old.y};
                    uint16_t res[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float hv = bf16_to_f32((bf16_t)((ow[e >> 1] >> (16 * (e & 1))) & 0xffffu));
                        res[e] = f32_to_bf16(hv + bf16_round_f32(acc[i][j][e]));                  // h = T(h + T(x W^T))
                    }
                    uint2 v;
                    v.x = (uint32_t)res[0] | ((uint32_t)res[1] << 16);
                    v.y = (uint32_t)res[2] | ((uint32_t)res[3] << 16);
                    *reinterpret_cast<uint2*>(h) = v;
                }
            }
        }
    }
}

void launch_gemm_pf(int epi, const bf16_t* X, const bf16_t* Wp, void* C, int M, int N, int K, hipStream_t s) {
    MIS_REQUIRE(K % PF_BK == 0 && K >= 2 * PF_BK && N % 16 == 0 && M >= 1, MIS_ERR_GENERATION_FAILED, "prefill GEMM: K must be a multiple of 64");
    MIS_REQUIRE(epi != PF_SILU || N % 32 == 0, MIS_ERR_GENERATION_FAILED, "prefill GEMM: gate / up tiles come in pairs");
    PfGemmParams p{X, Wp, C, M, N, K};
    dim3 grid(cdiv(N, PF_BN), cdiv(M, PF_BM)), block(256);
    if (epi == PF_F32) hipLaunchKernelGGL((k_gemm_pf<PF_F32>), grid, block, 0, s, p);
    else if (epi == PF_RESID) hipLaunchKernelGGL((k_gemm_pf<PF_RESID>), grid, block, 0, s, p);
    else if (epi == PF_SILU) hipLaunchKernelGGL((k_gemm_pf<PF_SILU>), grid, block, 0, s, p);
    else throw MisError(MIS_ERR_GENERATION_FAILED, "unknown prefill GEMM epilogue");
}
void launch_pf_embed_rmsnorm(const bf16_t* emb, const int32_t* prompt, const int32_t* lens, int Lmax, int t0, int Tc, int batch, int Mpad, int vocab,
                             const bf16_t* wnorm, bf16_t* h, bf16_t* x, int32_t* pos_tab, uint8_t* act_tab, int d, float eps, hipStream_t s) {
    hipLaunchKernelGGL(k_pf_embed_rmsnorm, dim3(Tc * Mpad), dim3(256), 0, s, emb, prompt, lens, Lmax, t0, batch, Mpad, vocab, wnorm, h, x, pos_tab,
                       act_tab, d, eps);
}
void launch_pf_rows_rmsnorm(const bf16_t* rows, int src_rows, const int32_t* lens, int Lmax, int t0, int Tc, int batch, int Mpad, const bf16_t* wnorm,
                            bf16_t* h, bf16_t* x, int32_t* pos_tab, uint8_t* act_tab, int d, float eps, hipStream_t s) {
    hipLaunchKernelGGL(k_pf_rows_rmsnorm, dim3(Tc * Mpad), dim3(256), 0, s, rows, src_rows, lens, Lmax, t0, batch, Mpad, wnorm, h, x, pos_tab, act_tab, d, eps);
}
void launch_pf_rmsnorm(const bf16_t* h, const bf16_t* wnorm, bf16_t* x, int rows, int d, float eps, hipStream_t s) {
    hipLaunchKernelGGL(k_pf_rmsnorm, dim3(rows), dim3(256), 0, s, h, wnorm, x, d, eps);
}
// h = T(h + T(part)): the residual add behind a GEMM that wrote float32 sums (quantised roles run the decode step's code-streaming
// kernel on 64-row chunks; same rounding points as the PF_RESID epilogue and the decode step's glue)
__global__ void k_pf_add_resid(bf16_t* __restrict__ h, const float* __restrict__ part, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) h[i] = f32_to_bf16(bf16_to_f32(h[i]) + bf16_round_f32(part[i]));
}
void launch_pf_add_resid(bf16_t* h, const float* part, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_pf_add_resid, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h, part, n);
}
void launch_pf_pack_rows(const bf16_t* rows, bf16_t* xpk, int Mpad, int d, hipStream_t s) {
    hipLaunchKernelGGL(k_pf_pack_rows, dim3(Mpad), dim3(256), 0, s, rows, xpk, d, Mpad / 16);
}
