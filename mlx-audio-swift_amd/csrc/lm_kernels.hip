// lm_kernels.hip - Orpheus / Llama-3 decode-step kernels for gfx950.
//
// Reference being replaced: LlamaTTSModelInner / LlamaTTSAttention / LlamaTTSMLP
// (Sources/MLXAudioTTS/Models/Llama/LlamaTTS.swift:206-346,557-567) whose arithmetic is MLX
// (Linear, MLXFast.rmsNorm / RoPE / scaledDotProductAttention, KVCacheSimple).  Numerics follow
// the rounding points of MLX's bf16 graph (oracle/llama.py): every primitive's output is bf16,
// accumulation is f32.
//
// Data layout (all in HBM, resident for the whole generate call):
//   weights   bf16, pre-packed at load into MFMA-A tiles  [N/16][K/32][64 lanes][8]  so that one wave
//             instruction streams one contiguous 1 KiB tile (lane l = q*16+i holds W[16*nt+i][32*kt+8*q..+8])
//   x / attn_out / act   bf16, PACKED as MFMA-B fragments [K/32][MT][64 lanes][8]: lane l = q*16+j of tile
//             (kt, mt) holds X[m = 16*mt + j][k = 32*kt + 8*q .. +8]  (MT = Mpad/16, Mpad = batch rounded up to 16);
//             one fragment load = one contiguous 1 KiB read (row-major X costs 16 x 64 B segments per load:
//             measured -25 % on the big GEMMs).  The residual stream h stays row-major [Mpad][d].
//   K cache   bf16 tiled as QK^T A-fragments  [B][Hkv][Smax/32][2][D/32][64][8]
//   V cache   bf16 tiled as P.V  B-fragments  [B][Hkv][Smax/32][D/16][64][8]   (8 consecutive keys per lane)
//   split-K partial slabs f32 [S][Mpad][N]; reduced in the consumer's prologue (deterministic order)
#include <hip/hip_fp16.h>
#include "common.h"
#include <algorithm>
#include <type_traits>
#include <string>
#include "lm_kernels.h"


// ============================================================================ weight staging

__global__ void k_convert_to_bf16(const void* __restrict__ src, int dtype, bf16_t* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (dtype == MIS_F32) dst[i] = f32_to_bf16(((const float*)src)[i]);
    else if (dtype == MIS_F16) dst[i] = f32_to_bf16((float)((const _Float16*)src)[i]);
    else dst[i] = ((const bf16_t*)src)[i];
}

// row-major [N][K] bf16 -> packed tiles; destination tile index = nt * tile_stride + tile_offset
// (gate/up interleave: stride 2, offset 0/1).  Rows >= N are zero.  One thread per 16-byte unit.
__global__ void k_pack_weight(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N, int K, int NT,
                              int tile_stride, int tile_offset) {
    int KT = K / 32;
    size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)NT * KT * 64;
    if (u >= total) return;
    int lane = (int)(u & 63);
    size_t tk = u >> 6;
    int kt = (int)(tk % KT);
    int nt = (int)(tk / KT);
    int i = lane & 15, q = lane >> 4;
    int row = nt * 16 + i, col = kt * 32 + q * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < N) v = *reinterpret_cast<const uint4*>(src + (size_t)row * K + col);
    size_t dt = (size_t)nt * tile_stride + tile_offset;
    *reinterpret_cast<uint4*>(dst + ((dt * KT + kt) * 64 + lane) * 8) = v;
}

// mis-synth-v1 fill (oracle/synth.py).  plus_one: value = bf16(1 + bf16(x))  (norm weights)
__global__ void k_synth_fill_bf16(bf16_t* __restrict__ dst, size_t n, uint64_t key, float amp, int plus_one) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = mis_synth_value(key, i, amp);
    if (plus_one != 2) v = bf16_round_f32(v);        // plus_one: 1 = bf16(1 + bf16(x)), 2 = bf16(1 + x)
    if (plus_one) v = 1.0f + v;
    dst[i] = f32_to_bf16(v);
}

// MLX affine quantisation (mlx quantize / dequantize [3P]): element i of row o is (wq[o][i / epw] >> (bits * (i % epw))) & mask,
// epw = 32 / bits; w = scale[o][i / group] * q + bias[o][i / group].  Output bf16 row-major [N][K].
__global__ void k_dequant_affine(const uint32_t* __restrict__ wq, const void* __restrict__ scales, const void* __restrict__ biases,
                                 int sb_dtype, bf16_t* __restrict__ dst, int N, int K, int group, int bits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * K) return;
    const int o = (int)(i / K), k = (int)(i - (size_t)o * K);
    const int epw = 32 / bits;
    const uint32_t word = wq[(size_t)o * (K / epw) + k / epw];
    const float q = (float)((word >> (bits * (k % epw))) & ((1u << bits) - 1u));
    const size_t g = (size_t)o * (K / group) + k / group;
    float sc, bi;
    if (sb_dtype == 0) { sc = ((const float*)scales)[g]; bi = ((const float*)biases)[g]; }
    else if (sb_dtype == 2) { sc = bf16_to_f32(((const bf16_t*)scales)[g]); bi = bf16_to_f32(((const bf16_t*)biases)[g]); }
    else { sc = __half2float(((const __half*)scales)[g]); bi = __half2float(((const __half*)biases)[g]); }
    dst[i] = f32_to_bf16(sc * q + bi);
}
void launch_dequant_affine(const uint32_t* wq, const void* scales, const void* biases, int sb_dtype, bf16_t* dst, int N, int K,
                           int group, int bits, hipStream_t s) {
    size_t n = (size_t)N * K;
    hipLaunchKernelGGL(k_dequant_affine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, wq, scales, biases, sb_dtype, dst, N, K, group, bits);
}

void launch_convert_to_bf16(const void* src, int dtype, bf16_t* dst, size_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_convert_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dtype, dst, n);
}
void launch_pack_weight(const bf16_t* src, bf16_t* dst, int N, int K, int NT, int tile_stride, int tile_offset,
                        hipStream_t s) {
    size_t total = (size_t)NT * (K / 32) * 64;
    hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, N, K, NT,
                       tile_stride, tile_offset);
}
void launch_synth_fill_bf16(bf16_t* dst, size_t n, uint64_t key, float amp, int plus_one, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_synth_fill_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, n, key, amp, plus_one);
}

// (Round 3, measured and removed again: reading the NEXT launches' weights into the 256 MB Infinity Cache ahead of time.  As a forked
// branch of the step graph every fork cost ~16 us on ROCm 7.2 (step 2.13 -> 2.95..4.1 ms); as extra blocks of the chain's own launches
// - glue, split-K GEMMs, the attention waves that run out of key tiles - every schedule was slower too (2.144 -> 2.17..2.35 ms): the
// carriers lose more than the consumers gain, the step's HBM time is conserved.  profiles/r03/ab1_*.json, ab2_*.json; DESIGN.md.)

// (Round 3, measured and removed again: a 5-launch layer for hidden sizes <= 1024 - RMSNorm rebuilt per block in the consumer GEMM's
// prologue, residual added in place by the producer GEMM's epilogue.  Parity-green; slower on Qwen3-TTS-0.6B (4.35-4.41 vs 3.92
// ms/frame): the prologue costs 2.5-3 us per consumer launch (VALU work plus a block-wide reduction in front of the first MFMA) and
// a producer without inter-block split-K keeps only 64 of 256 CUs streaming (7.2 vs 4.8 us).  profiles/r03/q3/; DESIGN.md.)

// ============================================================================ step bookkeeping

// prefill feeder: step j of a left-padded prompt matrix [B][Lmax]; row b starts at j0 = Lmax - len[b]
// (the pad tokens of prepareInputIds, LlamaTTS.swift:497-506, are never fed: SURVEY App. D.1).
__global__ void k_prefill_feed(const int32_t* __restrict__ prompt, const int32_t* __restrict__ lens, int Lmax,
                               int* __restrict__ step_counter, int32_t* __restrict__ ids, uint8_t* __restrict__ active,
                               int batch) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    int j = *step_counter;
    if (b < batch) {
        int j0 = Lmax - lens[b];
        if (j >= j0 && j < Lmax) { ids[b] = prompt[(size_t)b * Lmax + j]; active[b] = 1; }
        else { ids[b] = 0; active[b] = 0; }
    }
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        // every block has read *step_counter before the LAST block's thread can run this only if a
        // single block is used; the launcher guarantees gridDim.x == 1.
        *step_counter = j + 1;
    }
}
void launch_prefill_feed(const int32_t* prompt, const int32_t* lens, int Lmax, int* step_counter, int32_t* ids,
                         uint8_t* active, int batch, hipStream_t s) {
    hipLaunchKernelGGL(k_prefill_feed, dim3(1), dim3(64 * ((batch + 63) / 64)), 0, s, prompt, lens, Lmax, step_counter,
                       ids, active, batch);
}

// ============================================================================ embed + RMSNorm

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
}

// One 256-thread block per row.  h = E[id] ; x = RMSNorm(h) * w   (LlamaTTS.swift:336, :306)
// Also advances the per-row position: pos_cur = pos_next ; pos_next += active.
__global__ void __launch_bounds__(256) k_embed_rmsnorm(const bf16_t* __restrict__ emb, const int32_t* __restrict__ ids,
                                                       const uint8_t* __restrict__ active, int* __restrict__ pos_cur,
                                                       int* __restrict__ pos_next, const bf16_t* __restrict__ wnorm,
                                                       bf16_t* __restrict__ h, bf16_t* __restrict__ x, int d, int vocab,
                                                       float eps, int batch) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    // two memory round trips: (ids, positions, norm weights) then the embedding row.  8 columns (16 B) per thread and slot,
    // EMB_SLOTS slots of 256 threads; loads unconditional on clamped addresses, everything kept in registers (no second pass).
    constexpr int EMB_SLOTS = 6;                                        // d <= 6 * 256 * 8 = 12288
    const int nch = d >> 3;                                             // 16-byte chunks per row (d % 8 == 0, checked at launch)
    int id = ids[m < batch ? m : 0];
    uint4 wq[EMB_SLOTS];
#pragma unroll
    for (int k = 0; k < EMB_SLOTS; ++k) {
        const int ch = threadIdx.x + 256 * k;
        wq[k] = reinterpret_cast<const uint4*>(wnorm)[ch < nch ? ch : 0];
    }
    if (m >= batch || id < 0 || id >= vocab) id = 0;
    if (threadIdx.x == 0 && m < batch) {
        int p = pos_next[m];
        pos_cur[m] = p;
        if (active[m]) pos_next[m] = p + 1;
    }
    const uint4* e = reinterpret_cast<const uint4*>(emb + (size_t)id * d);
    uint4 eq[EMB_SLOTS];
#pragma unroll
    for (int k = 0; k < EMB_SLOTS; ++k) {
        const int ch = threadIdx.x + 256 * k;
        eq[k] = e[ch < nch ? ch : 0];
    }
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < EMB_SLOTS; ++k) {
        const int ch = threadIdx.x + 256 * k;
        if (ch < nch) {
            reinterpret_cast<uint4*>(h + (size_t)m * d)[ch] = eq[k];
            const uint32_t q[4] = {eq[k].x, eq[k].y, eq[k].z, eq[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float f0 = bf16_to_f32((bf16_t)(q[j] & 0xffffu)), f1 = bf16_to_f32((bf16_t)(q[j] >> 16));
                ss += f0 * f0; ss += f1 * f1;
            }
        }
    }
    float tot = block_sum_256(ss, red);
    float inv = 1.0f / sqrtf(tot / (float)d + eps);
    const int MT = gridDim.x >> 4;
#pragma unroll
    for (int k = 0; k < EMB_SLOTS; ++k) {
        const int ch = threadIdx.x + 256 * k;
        if (ch < nch) {
            const uint32_t q[4] = {eq[k].x, eq[k].y, eq[k].z, eq[k].w};
            const uint32_t w[4] = {wq[k].x, wq[k].y, wq[k].z, wq[k].w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float f0 = bf16_to_f32((bf16_t)(q[j] & 0xffffu)), f1 = bf16_to_f32((bf16_t)(q[j] >> 16));
                const float w0 = bf16_to_f32((bf16_t)(w[j] & 0xffffu)), w1 = bf16_to_f32((bf16_t)(w[j] >> 16));
                const bf16_t r0 = f32_to_bf16(w0 * bf16_round_f32(f0 * inv));      // T(w * T(x * rsqrt(mean + eps)))
                const bf16_t r1 = f32_to_bf16(w1 * bf16_round_f32(f1 * inv));
                o[j] = (uint32_t)r0 | ((uint32_t)r1 << 16);
            }
            // 8 consecutive columns of one row are one 16-byte element of the packed operand layout
            *reinterpret_cast<uint4*>(x + xpk_index(m, ch * 8, MT)) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}
void launch_embed_rmsnorm(const bf16_t* emb, const int32_t* ids, const uint8_t* active, int* pos_cur, int* pos_next,
                          const bf16_t* wnorm, bf16_t* h, bf16_t* x, int d, int vocab, float eps, int batch, int Mpad,
                          hipStream_t s) {
    MIS_REQUIRE(d % 8 == 0 && d <= 6 * 256 * 8, MIS_ERR_INVALID_INPUT, "hidden size must be a multiple of 8 and <= 12288");
    hipLaunchKernelGGL(k_embed_rmsnorm, dim3(Mpad), dim3(256), 0, s, emb, ids, active, pos_cur, pos_next, wnorm, h, x, d,
                       vocab, eps, batch);
}

// One 1024-thread block per row: o = T(sum_s slab[s]) ; h = T(h + o) ; x = RMSNorm(h) * w  (LlamaTTS.swift:306-309)
// Each thread owns <= RR_MAXC columns; all S x RR_MAXC slab loads of a thread are independent, so one L2
// round trip covers them (the v0 kernel chained 120 dependent loads: 34 us per call, 31 % of a step).
#define RR_THREADS 1024
#define RR_MAXC 3
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    v = wave_sum(v);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < RR_THREADS / 64; ++i) t += red[i];
    __syncthreads();
    return t;
}
// SG = slabs fetched per round trip (2, 4 or 8 >= S where possible): the loads are unconditional on clamped slab indices, so a
// fixed 8 cost the S = 2 launches (o_proj) 18 redundant load instructions per thread
template <int SG>
__global__ void __launch_bounds__(RR_THREADS) k_reduce_residual_rmsnorm(const float* __restrict__ slabs, int S,
                                                                        int Mpad, int N, bf16_t* __restrict__ h,
                                                                        const bf16_t* __restrict__ wnorm,
                                                                        bf16_t* __restrict__ x, float eps,
                                                                        const bf16_t* __restrict__ ln_bias) {
    // ln_bias == nullptr: RMSNorm (Llama).  Otherwise LayerNorm with weight wnorm and bias ln_bias (Whisper,
    // WhisperLayers.swift:90-107): f32 statistics, one rounding at the output.
    __shared__ float red[RR_THREADS / 64];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int MT = gridDim.x >> 4;
    float ss = 0.0f;
    float keep[RR_MAXC];                 // h_new of this thread's columns (valid when N <= RR_THREADS*RR_MAXC)
    float wv[RR_MAXC];
    for (int c0 = 0; c0 < N; c0 += RR_THREADS * RR_MAXC) {
        float acc[RR_MAXC], hv[RR_MAXC];
#pragma unroll
        // every load below is UNCONDITIONAL (out-of-range columns / slabs read a clamped address and are masked afterwards):
        // a guarded load compiles to a branch with its own s_waitcnt vmcnt(0), i.e. one dependent memory round trip per
        // guard - four in a row here before this was straight-line code, against the single one the data flow needs.
        for (int k = 0; k < RR_MAXC; ++k) {
            int i = c0 + tid + k * RR_THREADS;
            const int ic = i < N ? i : N - 1;
            acc[k] = 0.0f;
            hv[k] = bf16_to_f32(h[(size_t)m * N + ic]);
            wv[k] = bf16_to_f32(wnorm[ic]);
        }
        for (int s0 = 0; s0 < S; s0 += SG) {    // SG slabs x RR_MAXC columns: one round trip (slabs come from HBM/MALL:
            float v[SG][RR_MAXC];               // the producer's L2 lines were written back at the kernel boundary)
#pragma unroll
            for (int j = 0; j < SG; ++j)
#pragma unroll
                for (int k = 0; k < RR_MAXC; ++k) {
                    int i = c0 + tid + k * RR_THREADS;
                    const int ic = i < N ? i : N - 1;
                    const int sj = s0 + j < S ? s0 + j : S - 1;
                    v[j][k] = slabs[((size_t)sj * Mpad + m) * N + ic];
                }
#pragma unroll
            for (int j = 0; j < SG; ++j)       // slab order s = 0,1,2,... (fixed => deterministic)
#pragma unroll
                for (int k = 0; k < RR_MAXC; ++k) acc[k] += (s0 + j < S) ? v[j][k] : 0.0f;
        }
        // pin the norm-weight loads here (they were issued ahead of the slab loads, so they have landed): left alone the compiler
        // sinks them to their use after the block reduction - one more dependent round trip at the tail of the kernel
#pragma unroll
        for (int k = 0; k < RR_MAXC; ++k) asm volatile("" : "+v"(wv[k]));
#pragma unroll
        for (int k = 0; k < RR_MAXC; ++k) {
            int i = c0 + tid + k * RR_THREADS;
            keep[k] = 0.0f;
            if (i < N) {
                float o = bf16_round_f32(acc[k]);
                float hn = bf16_round_f32(hv[k] + o);
                h[(size_t)m * N + i] = f32_to_bf16(hn);
                keep[k] = hn;
                ss += hn * hn;
            }
        }
    }
    if (ln_bias) {
        float sum = 0.0f;
        if (N <= RR_THREADS * RR_MAXC) {
#pragma unroll
            for (int k = 0; k < RR_MAXC; ++k) sum += keep[k];
        } else {
            for (int i = tid; i < N; i += RR_THREADS) sum += bf16_to_f32(h[(size_t)m * N + i]);
        }
        const float mean = block_sum_1024(sum, red) / (float)N;
        float sq = 0.0f;
        if (N <= RR_THREADS * RR_MAXC) {
#pragma unroll
            for (int k = 0; k < RR_MAXC; ++k) {
                int i = tid + k * RR_THREADS;
                if (i < N) { float dlt = keep[k] - mean; sq += dlt * dlt; }
            }
        } else {
            for (int i = tid; i < N; i += RR_THREADS) { float dlt = bf16_to_f32(h[(size_t)m * N + i]) - mean; sq += dlt * dlt; }
        }
        const float rstd = 1.0f / sqrtf(block_sum_1024(sq, red) / (float)N + eps);
        for (int i = tid; i < N; i += RR_THREADS) {
            float f = bf16_to_f32(h[(size_t)m * N + i]);
            x[xpk_index(m, i, MT)] = f32_to_bf16((f - mean) * rstd * bf16_to_f32(wnorm[i]) + bf16_to_f32(ln_bias[i]));
        }
        return;
    }
    float tot = block_sum_1024(ss, red);
    float inv = 1.0f / sqrtf(tot / (float)N + eps);
    if (N <= RR_THREADS * RR_MAXC) {          // single pass: everything still in registers
#pragma unroll
        for (int k = 0; k < RR_MAXC; ++k) {
            int i = tid + k * RR_THREADS;
            if (i < N) x[xpk_index(m, i, MT)] = f32_to_bf16(wv[k] * bf16_round_f32(keep[k] * inv));
        }
    } else {
        for (int i = tid; i < N; i += RR_THREADS) {
            float f = bf16_to_f32(h[(size_t)m * N + i]);          // written by this same thread above
            x[xpk_index(m, i, MT)] = f32_to_bf16(bf16_to_f32(wnorm[i]) * bf16_round_f32(f * inv));
        }
    }
}
// The RMSNorm case again with FOUR consecutive columns per thread (N / 4 <= 1024 threads, single pass): one float4 per slab, one
// 8-byte load each for h and the norm weight, 8-byte stores (four consecutive columns are half of a 16-byte unit of the packed
// operand layout) - S + 2 load instructions per thread instead of 3 S + 6, and 12 waves per row instead of 16 at d = 3072.
// LN = true: LayerNorm with weight and bias instead (Whisper, WhisperLayers.swift:90-107): float32 mean and variance over the row (two
// block sums), one rounding at the output - the arithmetic of k_reduce_residual_rmsnorm's ln_bias branch.
template <int SG, bool LN>
__global__ void __launch_bounds__(1024) k_glue4(const float* __restrict__ slabs, int S, int Mpad, int N, bf16_t* __restrict__ h,
                                                const bf16_t* __restrict__ wnorm, bf16_t* __restrict__ x, float eps,
                                                const bf16_t* __restrict__ ln_bias) {
    __shared__ float red[16];
    const int m = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const int MT = Mpad >> 4;
    const int c4 = tid < (N >> 2) ? tid : (N >> 2) - 1;          // clamped: threads past the row load valid addresses, results masked
    const bool live = tid < (N >> 2);
    // The loads are written as asm statements: hipcc splits, sinks and re-orders plain loads of `const __restrict__` data around
    // any pin (a slab load ended up behind the wait for the others: two dependent round trips).  Volatile asm statements keep their
    // order: S + 2 loads in flight, ONE wait naming every destination, then the math.  (hipcc does not count asm loads in its own
    // s_waitcnt bookkeeping; the kernel has no other vector loads.)
    unsigned long long hq, wq, bq = 0;
    f32x4_t v[SG];
    {
        const bf16_t* hp = h + (size_t)m * N + 4 * c4;
        const bf16_t* wp = wnorm + 4 * c4;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(hq) : "v"(hp));
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(wq) : "v"(wp));
        if constexpr (LN) {
            const bf16_t* bp = ln_bias + 4 * c4;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(bq) : "v"(bp));
        }
#pragma unroll
        for (int j = 0; j < SG; ++j) {
            const int sj = j < S ? j : S - 1;
            const float* sp = slabs + ((size_t)sj * Mpad + m) * N + 4 * c4;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(sp));
        }
        if constexpr (SG == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hq), "+v"(wq), "+v"(v[0]), "+v"(v[1]));
        else if constexpr (SG == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hq), "+v"(wq), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(hq), "+v"(wq), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]),
                          "+v"(v[6]), "+v"(v[7]));
        if constexpr (LN) asm volatile("" : "+v"(bq));          // covered by the vmcnt(0) above (issued before it)
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < SG; ++j) {                              // slab order 0, 1, 2, ... (fixed => deterministic)
        const bool on = j < S;
        acc[0] += on ? v[j][0] : 0.0f; acc[1] += on ? v[j][1] : 0.0f; acc[2] += on ? v[j][2] : 0.0f; acc[3] += on ? v[j][3] : 0.0f;
    }
    const uint32_t hq0 = (uint32_t)hq, hq1 = (uint32_t)(hq >> 32);
    const float hv[4] = {bf16_to_f32((bf16_t)(hq0 & 0xffffu)), bf16_to_f32((bf16_t)(hq0 >> 16)), bf16_to_f32((bf16_t)(hq1 & 0xffffu)),
                         bf16_to_f32((bf16_t)(hq1 >> 16))};
    const uint32_t wq0 = (uint32_t)wq, wq1 = (uint32_t)(wq >> 32);
    const float wv[4] = {bf16_to_f32((bf16_t)(wq0 & 0xffffu)), bf16_to_f32((bf16_t)(wq0 >> 16)), bf16_to_f32((bf16_t)(wq1 & 0xffffu)),
                         bf16_to_f32((bf16_t)(wq1 >> 16))};
    float hn[4], ss = 0.0f;
    bf16_t hb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hn[e] = bf16_round_f32(hv[e] + bf16_round_f32(acc[e]));   // o = T(sum slabs); h = T(h + o)
        hb[e] = f32_to_bf16(hn[e]);
        ss += live ? hn[e] * hn[e] : 0.0f;
    }
    if (live) *reinterpret_cast<uint2*>(h + (size_t)m * N + 4 * c4) =
        make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
    if constexpr (LN) {
        float sm = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) sm += live ? hn[e] : 0.0f;
        sm = wave_sum_dpp(sm);
        if ((tid & 63) == 0) red[tid >> 6] = sm;
        __syncthreads();
        float tot = 0.0f;
        const int nw = nth >> 6;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += i < nw ? red[i] : 0.0f;
        const float mean = tot / (float)N;
        float sq = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float dl = hn[e] - mean; sq += live ? dl * dl : 0.0f; }
        sq = wave_sum_dpp(sq);
        __syncthreads();                                         // everyone has read the first sums
        if ((tid & 63) == 0) red[tid >> 6] = sq;
        __syncthreads();
        float tq = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) tq += i < nw ? red[i] : 0.0f;
        const float rstd = 1.0f / sqrtf(tq / (float)N + eps);
        if (live) {
            const uint32_t bq0 = (uint32_t)bq, bq1 = (uint32_t)(bq >> 32);
            const float bv[4] = {bf16_to_f32((bf16_t)(bq0 & 0xffffu)), bf16_to_f32((bf16_t)(bq0 >> 16)), bf16_to_f32((bf16_t)(bq1 & 0xffffu)),
                                 bf16_to_f32((bf16_t)(bq1 >> 16))};
            bf16_t xb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) xb[e] = f32_to_bf16((hn[e] - mean) * rstd * wv[e] + bv[e]);
            *reinterpret_cast<uint2*>(x + xpk_index(m, 4 * c4, MT)) =
                make_uint2((uint32_t)xb[0] | ((uint32_t)xb[1] << 16), (uint32_t)xb[2] | ((uint32_t)xb[3] << 16));
        }
        return;
    }
    ss = wave_sum_dpp(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.0f;
    const int nw = nth >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += i < nw ? red[i] : 0.0f;
    const float inv = 1.0f / sqrtf(tot / (float)N + eps);
    if (live) {
        bf16_t xb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) xb[e] = f32_to_bf16(wv[e] * bf16_round_f32(hn[e] * inv));
        *reinterpret_cast<uint2*>(x + xpk_index(m, 4 * c4, MT)) =
            make_uint2((uint32_t)xb[0] | ((uint32_t)xb[1] << 16), (uint32_t)xb[2] | ((uint32_t)xb[3] << 16));
    }
}

// The glue with SEVERAL column groups per thread (round 4): CPT groups of four columns per thread, N / (4 CPT) threads per row - 256 at
// d = 3072 (CPT = 3: four waves instead of twelve; the only dispatch - see the launcher for the single-wave forms that lost).  The launch is latency-bound (one dependent round trip, a row reduction, 32 blocks), so what a block costs before
// its first load is issued and at its barriers counts: fewer waves to place and to meet.  Same arithmetic and the same summation order per
// column as k_glue4 (slab order 0, 1, ...); the row statistics are reduced in another order (CPT groups per lane, DPP, then the waves) -
// float32 sums of N terms either way.  LN = true: LayerNorm with weight and bias (Whisper, WhisperLayers.swift:90-107): float32 mean and
// variance over the row, one rounding at the output.  MIS_GLUE_CPT=1 keeps one group per thread (A/B).
template <int SG, int CPT, bool LN>
__global__ void __launch_bounds__(256) k_glue_cpt(const float* __restrict__ slabs, int S, int Mpad, int N, bf16_t* __restrict__ h,
                                                  const bf16_t* __restrict__ wnorm, bf16_t* __restrict__ x, float eps, const bf16_t* __restrict__ ln_bias) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;          // N == 4 * CPT * nth (checked by the launcher)
    const int MT = Mpad >> 4;
    const int nw = nth >> 6;
    unsigned long long hq[CPT], wq[CPT], bq[CPT];
    f32x4_t v[CPT][SG];
#pragma unroll
    for (int g = 0; g < CPT; ++g) {
        const int c4 = tid + g * nth;
        const bf16_t* hp = h + (size_t)m * N + 4 * c4;
        const bf16_t* wp = wnorm + 4 * c4;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(hq[g]) : "v"(hp));
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(wq[g]) : "v"(wp));
        if constexpr (LN) {
            const bf16_t* bp = ln_bias + 4 * c4;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(bq[g]) : "v"(bp));
        } else bq[g] = 0;
    }
#pragma unroll
    for (int j = 0; j < SG; ++j) {
        const int sj = j < S ? j : S - 1;
#pragma unroll
        for (int g = 0; g < CPT; ++g) {
            const float* sp = slabs + ((size_t)sj * Mpad + m) * N + 4 * (tid + g * nth);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[g][j]) : "v"(sp));
        }
    }
    // ONE wait, then every destination named (volatile asm statements keep their order; hipcc does not count asm loads itself)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = 0; g < CPT; ++g) {
        asm volatile("" : "+v"(hq[g]), "+v"(wq[g]));
        if constexpr (LN) asm volatile("" : "+v"(bq[g]));
#pragma unroll
        for (int j = 0; j < SG; ++j) asm volatile("" : "+v"(v[g][j]));
    }
    auto row_sum = [&](float part, bool again) {                // the same total in every thread of the block
        part = wave_sum_dpp(part);
        if (nw == 1) return part;
        if (again) __syncthreads();                             // (a second use of `red`: everyone has read the first totals)
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        float tot = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) tot += i < nw ? red[i] : 0.0f;
        return tot;
    };
    float hn[CPT][4], wv[CPT][4], ss = 0.0f, sm = 0.0f;
#pragma unroll
    for (int g = 0; g < CPT; ++g) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < SG; ++j) {                           // slab order 0, 1, 2, ... (fixed => deterministic)
            const bool on = j < S;
            acc[0] += on ? v[g][j][0] : 0.0f; acc[1] += on ? v[g][j][1] : 0.0f; acc[2] += on ? v[g][j][2] : 0.0f; acc[3] += on ? v[g][j][3] : 0.0f;
        }
        const uint32_t h0 = (uint32_t)hq[g], h1 = (uint32_t)(hq[g] >> 32), w0 = (uint32_t)wq[g], w1 = (uint32_t)(wq[g] >> 32);
        const float hv[4] = {bf16_to_f32((bf16_t)(h0 & 0xffffu)), bf16_to_f32((bf16_t)(h0 >> 16)), bf16_to_f32((bf16_t)(h1 & 0xffffu)), bf16_to_f32((bf16_t)(h1 >> 16))};
        wv[g][0] = bf16_to_f32((bf16_t)(w0 & 0xffffu)); wv[g][1] = bf16_to_f32((bf16_t)(w0 >> 16));
        wv[g][2] = bf16_to_f32((bf16_t)(w1 & 0xffffu)); wv[g][3] = bf16_to_f32((bf16_t)(w1 >> 16));
        bf16_t hb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hn[g][e] = bf16_round_f32(hv[e] + bf16_round_f32(acc[e]));   // o = T(sum slabs); h = T(h + o)
            hb[e] = f32_to_bf16(hn[g][e]);
            ss += hn[g][e] * hn[g][e];
            sm += hn[g][e];
        }
        *reinterpret_cast<uint2*>(h + (size_t)m * N + 4 * (tid + g * nth)) =
            make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
    }
    if constexpr (LN) {
        const float mean = row_sum(sm, false) / (float)N;
        float sq = 0.0f;
#pragma unroll
        for (int g = 0; g < CPT; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dl = hn[g][e] - mean; sq += dl * dl; }
        const float rstd = 1.0f / sqrtf(row_sum(sq, true) / (float)N + eps);
#pragma unroll
        for (int g = 0; g < CPT; ++g) {
            const uint32_t b0 = (uint32_t)bq[g], b1 = (uint32_t)(bq[g] >> 32);
            const float bv[4] = {bf16_to_f32((bf16_t)(b0 & 0xffffu)), bf16_to_f32((bf16_t)(b0 >> 16)), bf16_to_f32((bf16_t)(b1 & 0xffffu)), bf16_to_f32((bf16_t)(b1 >> 16))};
            bf16_t xb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) xb[e] = f32_to_bf16((hn[g][e] - mean) * rstd * wv[g][e] + bv[e]);
            *reinterpret_cast<uint2*>(x + xpk_index(m, 4 * (tid + g * nth), MT)) =
                make_uint2((uint32_t)xb[0] | ((uint32_t)xb[1] << 16), (uint32_t)xb[2] | ((uint32_t)xb[3] << 16));
        }
    } else {
        const float inv = 1.0f / sqrtf(row_sum(ss, false) / (float)N + eps);
#pragma unroll
        for (int g = 0; g < CPT; ++g) {
            bf16_t xb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) xb[e] = f32_to_bf16(wv[g][e] * bf16_round_f32(hn[g][e] * inv));
            *reinterpret_cast<uint2*>(x + xpk_index(m, 4 * (tid + g * nth), MT)) =
                make_uint2((uint32_t)xb[0] | ((uint32_t)xb[1] << 16), (uint32_t)xb[2] | ((uint32_t)xb[3] << 16));
        }
    }
}
template <int CPT, bool LN>
static void launch_glue_cpt(const float* slabs, int S, int Mpad, int N, bf16_t* h, const bf16_t* wnorm, bf16_t* x, float eps, const bf16_t* ln_bias, hipStream_t s) {
    const int nth = N / (4 * CPT);
    if (S <= 2) hipLaunchKernelGGL((k_glue_cpt<2, CPT, LN>), dim3(Mpad), dim3(nth), 0, s, slabs, S, Mpad, N, h, wnorm, x, eps, ln_bias);
    else if (S <= 4) hipLaunchKernelGGL((k_glue_cpt<4, CPT, LN>), dim3(Mpad), dim3(nth), 0, s, slabs, S, Mpad, N, h, wnorm, x, eps, ln_bias);
    else hipLaunchKernelGGL((k_glue_cpt<8, CPT, LN>), dim3(Mpad), dim3(nth), 0, s, slabs, S, Mpad, N, h, wnorm, x, eps, ln_bias);
}

void launch_reduce_residual_rmsnorm(const float* slabs, int S, int Mpad, int N, bf16_t* h, const bf16_t* wnorm,
                                    bf16_t* x, float eps, hipStream_t s, const bf16_t* ln_bias) {
    // Dispatch by shape only (the A/B switches of rounds 3-4 are gone; their records: profiles/r03/ab1_*.json, profiles/r04/c13_ab.json):
    // d = 3072 -> three column groups per thread (256 threads instead of 768; step 2.0867 against 2.0934 ms).  The same kernel as ONE wave
    // per row for the small models (d = 1024 / 1280) measured SLOWER - Qwen3-TTS 3.86 -> 3.91 ms per frame, Whisper 243 -> 246 ms per
    // 8 x 30 s (profiles/r04/c14_secondary_ab.txt) - and six groups per thread at d = 3072 as well (2.121 against 2.087 ms per step,
    // c15_ab.json); every other width up to 4096 -> one group of four columns per thread (k_glue4); beyond that, or more than 8 slabs, the
    // general kernel.
    if (!ln_bias && N == 3072 && S <= 8) { launch_glue_cpt<3, false>(slabs, S, Mpad, N, h, wnorm, x, eps, ln_bias, s); return; }
    if (N % 4 == 0 && N / 4 <= 1024 && S <= 8) {
        const int nth = ((N / 4 + 63) / 64) * 64;
        if (ln_bias) {
            if (S <= 2) hipLaunchKernelGGL((k_glue4<2, true>), dim3(Mpad), dim3(nth), 0, s, slabs, S, Mpad, N, h, wnorm, x, eps, ln_bias);
            else if (S <= 4) hipLaunchKernelGGL((k_glue4<4, true>), dim3(Mpad), dim3(nth), 0, s, slabs, S, Mpad, N, h, wnorm, x, eps, ln_bias);
            else hipLaunchKernelGGL((k_glue4<8, true>), dim3(Mpad), dim3(nth), 0, s, slabs, S, Mpad, N, h, wnorm, x, eps, ln_bias);
        } else {
            if (S <= 2) hipLaunchKernelGGL((k_glue4<2, false>), dim3(Mpad), dim3(nth), 0, s, slabs, S, Mpad, N, h, wnorm, x, eps, ln_bias);
            else if (S <= 4) hipLaunchKernelGGL((k_glue4<4, false>), dim3(Mpad), dim3(nth), 0, s, slabs, S, Mpad, N, h, wnorm, x, eps, ln_bias);
            else hipLaunchKernelGGL((k_glue4<8, false>), dim3(Mpad), dim3(nth), 0, s, slabs, S, Mpad, N, h, wnorm, x, eps, ln_bias);
        }
        return;
    }
    if (S <= 2) hipLaunchKernelGGL((k_reduce_residual_rmsnorm<2>), dim3(Mpad), dim3(RR_THREADS), 0, s, slabs, S, Mpad, N, h, wnorm, x, eps, ln_bias);
    else if (S <= 4) hipLaunchKernelGGL((k_reduce_residual_rmsnorm<4>), dim3(Mpad), dim3(RR_THREADS), 0, s, slabs, S, Mpad, N, h, wnorm, x, eps, ln_bias);
    else hipLaunchKernelGGL((k_reduce_residual_rmsnorm<8>), dim3(Mpad), dim3(RR_THREADS), 0, s, slabs, S, Mpad, N, h, wnorm, x, eps, ln_bias);
}

// ============================================================================ weight-streaming skinny GEMM
//
// Y[m][n] = sum_k X[m][k] * W[n][k],  M = 16*MT <= 64 rows, bf16 in / f32 accumulate on
// v_mfma_f32_16x16x32_bf16 with A = W tile (16 n x 32 k), B = X^T (32 k x 16 m):
//   A lane l: W[n = l&15][k = (l>>4)*8 + e]      B lane l: X[m = l&15][k = (l>>4)*8 + e]
//   C/D lane l, reg r: n = (l>>4)*4 + r, m = l&15
// Work item = R consecutive n-tiles x one K slice (S slices across blocks).  KSB = 1: every wave is an
// independent item (no LDS, no barrier).  KSB = 4: the 4 waves of a block split their item's K range and
// combine through LDS in a fixed order (deterministic) - 4x the waves in flight without 4x the partial
// slabs.  Every weight load is one contiguous 1 KiB tile read exactly once from HBM (non-temporal); X
// comes from L2.  The main loop is software pipelined over two static register buffers of GEMM_U k-tiles,
// so 2*GEMM_U*(R+MT) KiB per wave are in flight while the MFMAs of the previous group run.
// HBM-bound: algorithmic bytes = N*K*2 per launch.

#ifndef GEMM_U
#define GEMM_U 4
#endif

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <int MT, int R, int EPI>
__device__ __forceinline__ void gemm_epilogue(const f32x4_t (&acc)[R][MT], void* __restrict__ out, int ntg, int ks,
                                              int NT, int N_out, int Mpad, int lane, int mt_only,
                                              const bf16_t* __restrict__ bias) {
    const int nl = (lane >> 4) * 4, ml = lane & 15;
    if (EPI == EPI_PARTIAL) {
        float* o = reinterpret_cast<float*>(out);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int tile = ntg * R + r;
            if (tile >= NT) continue;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias && ks == 0) {                 // Linear bias rides on slab 0 (addMM: T(acc + b) after the slab sum)
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = bf16_to_f32(bias[tile * 16 + nl + e]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt_only >= 0 && mt != mt_only) continue;
                size_t off = ((size_t)ks * Mpad + mt * 16 + ml) * N_out + tile * 16 + nl;
                *reinterpret_cast<float4*>(o + off) =
                    make_float4(acc[r][mt][0] + bv[0], acc[r][mt][1] + bv[1], acc[r][mt][2] + bv[2], acc[r][mt][3] + bv[3]);
            }
        }
    } else if (EPI == EPI_BF16 || EPI == EPI_GELU_PACKED || EPI == EPI_SILU_PACKED) {
        bf16_t* o = reinterpret_cast<bf16_t*>(out);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int tile = ntg * R + r;
            if (tile >= NT) continue;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = bf16_to_f32(bias[tile * 16 + nl + e]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt_only >= 0 && mt != mt_only) continue;
                uint16_t res[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[r][mt][e] + bv[e];
                    if (EPI == EPI_GELU_PACKED) v = gelu_erf(bf16_round_f32(v));       // T(gelu(T(xW + b)))
                    if (EPI == EPI_SILU_PACKED) {                                        // h = T(xW + b); T(h * T(sigmoid(h)))
                        float hh = bf16_round_f32(v);
                        v = hh * bf16_round_f32(1.0f / (1.0f + __expf(-hh)));
                    }
                    res[e] = f32_to_bf16(v);
                }
                size_t off = (EPI == EPI_GELU_PACKED || EPI == EPI_SILU_PACKED) ? xpk_index(mt * 16 + ml, tile * 16 + nl, MT)
                                                      : ((size_t)mt * 16 + ml) * N_out + tile * 16 + nl;
                uint2 v;
                v.x = (uint32_t)res[0] | ((uint32_t)res[1] << 16);
                v.y = (uint32_t)res[2] | ((uint32_t)res[3] << 16);
                *reinterpret_cast<uint2*>(o + off) = v;
            }
        }
    } else {   // EPI_SILU_MUL: tile 2t = gate rows, 2t+1 = up rows  (LlamaTTS.swift:283)
        bf16_t* o = reinterpret_cast<bf16_t*>(out);
#pragma unroll
        for (int p = 0; p < R / 2; ++p) {                                  // tile pair p of the item: feature tile ntg * R / 2 + p
            if (R > 2 && (ntg * R + 2 * p + 1) >= NT) continue;            // (two tiles per item: NT is even, the item exists)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt_only >= 0 && mt != mt_only) continue;
                uint16_t res[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float g = bf16_round_f32(acc[2 * p][mt][e]);
                    float u = bf16_round_f32(acc[2 * p + 1][mt][e]);
                    float sg = bf16_round_f32(1.0f / (1.0f + __expf(-g)));     // T(sigmoid(g))
                    float a = bf16_round_f32(g * sg);                          // T(g * sigmoid)
                    res[e] = f32_to_bf16(a * u);                               // T(silu * up)
                }
                // act is the down-projection's X operand: packed fragment layout (k = feature index)
                size_t off = xpk_index(mt * 16 + ml, (ntg * (R / 2) + p) * 16 + nl, MT);
                uint2 v;
                v.x = (uint32_t)res[0] | ((uint32_t)res[1] << 16);
                v.y = (uint32_t)res[2] | ((uint32_t)res[3] << 16);
                *reinterpret_cast<uint2*>(o + off) = v;
            }
        }
    }
}

// Two register buffers of GEMM_U k-tiles per wave (one group in flight while one is consumed).  A ring of THREE (two groups in
// flight, 238 VGPRs, still two blocks per CU) was measured and is slower: gate+up 19.0 -> 19.7 us, step 165.0 -> 161.3 audio-s/s
// (profiles/r02_gemm_nbuf_ab.json) - and the same kernel streams weights that already sit in the Infinity Cache only 3.5 % faster
// than from HBM (profiles/r02_mall_probe.txt), so neither the memory nor the bytes in flight bound this launch; what is left is
// its ramp-up and tail (the lm_head instance, 8x longer, reaches 5.8 TB/s with the same loop).
// Round 4 (profiles/r04/gemm_lab_rows32.jsonl): the arrangement is a template parameter set - R n-tiles per wave, KSB waves per item
// (1: four independent items per 256-thread block; 2 / 4: the block's waves split the item's K range), U k-tiles per register buffer.
// Four n-tiles per wave halve the x fragments a wave re-reads out of L2 (as many bytes as the weights at two): gate+up 19.4 -> 17.6,
// qkv 8.5 -> 7.8, the output projection 173.5 -> 150.7 us.
template <int MT, int R, int EPI, int KSB, int U>
__global__ void __launch_bounds__((KSB == 1 ? 4 : KSB) * 64, 2) k_gemm_skinny(const bf16_t* __restrict__ Wp, const bf16_t* __restrict__ X,
                                                     void* __restrict__ out, int NT, int KT, int S, int n_items,
                                                     int N_out, int Mpad, const bf16_t* __restrict__ bias) {
    static_assert(EPI != EPI_SILU_MUL || (R & 1) == 0, "silu-mul epilogue pairs a gate tile with an up tile");
    // k-tiles per register buffer: 4 at R <= 2; 3 or 2 at four n-tiles per wave (4 would spill: 2 x 4 x (4 + 2) fragments + 32 accumulators)
    constexpr int GU = U;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = (KSB == 1) ? blockIdx.x * 4 + wave : blockIdx.x;
    if (item >= n_items) return;                      // (KSB == 1: a wave of the last block may have no item)
    const int ntg = item / S, ks = item - ntg * S;
    int kt0 = (int)(((long long)KT * ks) / S), kt1 = (int)(((long long)KT * (ks + 1)) / S);
    if (KSB > 1) {                                    // this wave's quarter of the item's K range
        int len = kt1 - kt0;
        int a = kt0 + (int)(((long long)len * wave) / KSB), b = kt0 + (int)(((long long)len * (wave + 1)) / KSB);
        kt0 = a; kt1 = b;
    }
    const bf16x8_t* wp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int tile = ntg * R + r;
        if (tile >= NT) tile = NT - 1;                     // clamp (store is skipped in the epilogue)
        wp[r] = reinterpret_cast<const bf16x8_t*>(Wp) + (size_t)tile * KT * 64 + lane;
    }
    const bf16x8_t* xp[MT];                                // packed fragments: tile (kt, mt) at (kt*MT + mt)*64 + lane
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xp[mt] = reinterpret_cast<const bf16x8_t*>(X) + mt * 64 + lane;

    f32x4_t acc[R][MT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[r][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    bf16x8_t wA[GU][R], xA[GU][MT], wB[GU][R], xB[GU][MT];
    const int klast = kt1 - 1;
#define GEMM_LOAD_W(WBUF, KBASE)                                                                  \
    _Pragma("unroll") for (int u = 0; u < GU; ++u) {                                           \
        int kk = (KBASE) + u;                                                                      \
        kk = kk > klast ? klast : kk;               /* tail: redundant reload, MFMA is skipped */  \
        _Pragma("unroll") for (int r = 0; r < R; ++r)                                              \
            WBUF[u][r] = __builtin_nontemporal_load(wp[r] + (size_t)kk * 64);                      \
    }
#define GEMM_LOAD_X(XBUF, KBASE)                                                                  \
    _Pragma("unroll") for (int u = 0; u < GU; ++u) {                                           \
        int kk = (KBASE) + u;                                                                      \
        kk = kk > klast ? klast : kk;                                                              \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                          \
            XBUF[u][mt] = xp[mt][(size_t)kk * (MT * 64)];                                          \
    }
#define GEMM_LOAD(WBUF, XBUF, KBASE) GEMM_LOAD_W(WBUF, KBASE) GEMM_LOAD_X(XBUF, KBASE)
/* FULL: every k-tile of the group is inside the wave's range (no guard).  TAIL: the last group(s) of the range, guarded.       */
/* The main loop below uses FULL and unconditional loads only: with a guard inside the loop the waitcnt pass has to assume the   */
/* no-load path at the join and drains vmcnt far enough to stall the group just requested - one group in flight per wave.        */
#define GEMM_MATH_FULL(WBUF, XBUF)                                                                \
    _Pragma("unroll") for (int u = 0; u < GU; ++u) {                                           \
        _Pragma("unroll") for (int r = 0; r < R; ++r)                                              \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                      \
                acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WBUF[u][r], XBUF[u][mt],      \
                                                                     acc[r][mt], 0, 0, 0);         \
    }
#define GEMM_MATH_TAIL(WBUF, XBUF, KBASE)                                                         \
    _Pragma("unroll") for (int u = 0; u < GU; ++u) {                                           \
        if ((KBASE) + u < kt1) {                                                                   \
            _Pragma("unroll") for (int r = 0; r < R; ++r)                                          \
                _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                  \
                    acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WBUF[u][r], XBUF[u][mt],  \
                                                                         acc[r][mt], 0, 0, 0);     \
        }                                                                                          \
    }
    const bool has_k = kt0 < kt1;
    if (has_k) {
        int kt = kt0;
        GEMM_LOAD(wA, xA, kt)
        // steady state: groups kt and kt + U are full and both following loads exist
        // (sched barriers: the next group is REQUESTED before the current one is waited for - left alone the scheduler sinks
        // the loads in between the MFMAs, i.e. behind the wait for the current group)
        while (kt + 3 * GU <= kt1) {
            GEMM_LOAD(wB, xB, kt + GU)
            __builtin_amdgcn_sched_barrier(0);
            GEMM_MATH_FULL(wA, xA)
            __builtin_amdgcn_sched_barrier(0);
            GEMM_LOAD(wA, xA, kt + 2 * GU)
            __builtin_amdgcn_sched_barrier(0);
            GEMM_MATH_FULL(wB, xB)
            __builtin_amdgcn_sched_barrier(0);
            kt += 2 * GU;
        }
        // tail: one to three (partial) groups left, the first of them already in buffer A
        if (kt + GU < kt1) {
            GEMM_LOAD(wB, xB, kt + GU)
            __builtin_amdgcn_sched_barrier(0);
            GEMM_MATH_TAIL(wA, xA, kt)
            if (kt + 2 * GU < kt1) {
                GEMM_LOAD(wA, xA, kt + 2 * GU)
                __builtin_amdgcn_sched_barrier(0);
                GEMM_MATH_TAIL(wB, xB, kt + GU)
                GEMM_MATH_TAIL(wA, xA, kt + 2 * GU)
            } else {
                GEMM_MATH_TAIL(wB, xB, kt + GU)
            }
        } else {
            GEMM_MATH_TAIL(wA, xA, kt)
        }
    }
#undef GEMM_LOAD
#undef GEMM_LOAD_W
#undef GEMM_LOAD_X
#undef GEMM_MATH_FULL
#undef GEMM_MATH_TAIL

    if (KSB == 1) {
        gemm_epilogue<MT, R, EPI>(acc, out, ntg, ks, NT, N_out, Mpad, lane, -1, bias);
    } else {
        __shared__ float4 red[KSB][R * MT][64];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                red[wave][r * MT + mt][lane] = make_float4(acc[r][mt][0], acc[r][mt][1], acc[r][mt][2], acc[r][mt][3]);
        __syncthreads();
        // wave w finishes the m-tiles mt = w, w+KSB, ...: sum the KSB partials in fixed order, then epilogue
        for (int mt = wave; mt < MT; mt += KSB) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float4 s0 = red[0][r * MT + mt][lane];
#pragma unroll
                for (int w = 1; w < KSB; ++w) {
                    float4 t = red[w][r * MT + mt][lane];
                    s0.x += t.x; s0.y += t.y; s0.z += t.z; s0.w += t.w;
                }
#pragma unroll
                for (int m2 = 0; m2 < MT; ++m2)
                    if (m2 == mt) acc[r][m2] = (f32x4_t){s0.x, s0.y, s0.z, s0.w};
            }
            gemm_epilogue<MT, R, EPI>(acc, out, ntg, ks, NT, N_out, Mpad, lane, mt, bias);
        }
    }
}

template <int MT>
static void launch_gemm_mt(int epi, int R, int ksb, int U, const bf16_t* Wp, const bf16_t* X, void* out, int NT, int KT, int S,
                           int N_out, int Mpad, const bf16_t* bias, hipStream_t s) {
    int n_items = ((NT + R - 1) / R) * S;
    dim3 grid(ksb == 1 ? (n_items + 3) / 4 : n_items), block((ksb == 1 ? 4 : ksb) * 64);
#define GEMM_CASE(E, RR, KS, UU)                                                                              \
    if (epi == E && R == RR && ksb == KS && U == UU) {                                                        \
        hipLaunchKernelGGL((k_gemm_skinny<MT, RR, E, KS, UU>), grid, block, 0, s, Wp, X, out, NT, KT, S, n_items, \
                           N_out, Mpad, bias);                                                                    \
        return;                                                                                               \
    }
    GEMM_CASE(EPI_PARTIAL, 1, 1, 4)
    GEMM_CASE(EPI_PARTIAL, 1, 4, 4)
    GEMM_CASE(EPI_PARTIAL, 2, 1, 4)
    GEMM_CASE(EPI_PARTIAL, 2, 2, 4)
    GEMM_CASE(EPI_PARTIAL, 2, 4, 4)
    GEMM_CASE(EPI_BF16, 2, 1, 4)
    GEMM_CASE(EPI_BF16, 2, 4, 4)
    GEMM_CASE(EPI_SILU_MUL, 2, 1, 4)
    GEMM_CASE(EPI_SILU_MUL, 2, 4, 4)
    if constexpr (MT <= 2) {                       // four n-tiles per wave: 8 accumulator tiles, U <= 3 (no scratch up to 32 rows)
        GEMM_CASE(EPI_SILU_MUL, 4, 4, 3)
        GEMM_CASE(EPI_PARTIAL, 4, 4, 2)
        GEMM_CASE(EPI_PARTIAL, 4, 4, 3)
        GEMM_CASE(EPI_PARTIAL, 4, 2, 2)
        GEMM_CASE(EPI_PARTIAL, 4, 2, 3)
        GEMM_CASE(EPI_BF16, 4, 4, 3)
        GEMM_CASE(EPI_BF16, 4, 4, 2)
        GEMM_CASE(EPI_BF16, 4, 2, 3)
    }
    GEMM_CASE(EPI_GELU_PACKED, 2, 4, 4)
    GEMM_CASE(EPI_GELU_PACKED, 1, 4, 4)
    GEMM_CASE(EPI_BF16, 1, 4, 4)
    GEMM_CASE(EPI_SILU_PACKED, 2, 4, 4)
#undef GEMM_CASE
    throw MisError(MIS_ERR_GENERATION_FAILED, "unsupported GEMM variant (epilogue " + std::to_string(epi) + ", R " + std::to_string(R) + ", KSB " +
                                                  std::to_string(ksb) + ", U " + std::to_string(U) + ", " + std::to_string(Mpad) + " rows)");
}

void launch_gemm_skinny(int epi, int R, int ksb, const bf16_t* Wp, const bf16_t* X, void* out, int NT, int KT, int S,
                        int N_out, int Mpad, hipStream_t s, const bf16_t* bias, int U) {
    MIS_REQUIRE(epi == EPI_PARTIAL || S == 1, MIS_ERR_GENERATION_FAILED, "split-K needs the partial epilogue");
    if (U <= 0) U = R >= 4 ? 3 : GEMM_U;
    switch (Mpad / 16) {
        case 1: launch_gemm_mt<1>(epi, R, ksb, U, Wp, X, out, NT, KT, S, N_out, Mpad, bias, s); break;
        case 2: launch_gemm_mt<2>(epi, R, ksb, U, Wp, X, out, NT, KT, S, N_out, Mpad, bias, s); break;
        case 3: launch_gemm_mt<3>(epi, R, ksb, U, Wp, X, out, NT, KT, S, N_out, Mpad, bias, s); break;
        case 4: launch_gemm_mt<4>(epi, R, ksb, U, Wp, X, out, NT, KT, S, N_out, Mpad, bias, s); break;
        default: throw MisError(MIS_ERR_INVALID_INPUT, "batch per GPU must be <= 64");
    }
}

// ============================================================================ skinny GEMM with the glue in its prologue (<= 16 rows)
//
// VERDICT r05 item 2 (Whisper decoder at 8 rows: three LayerNorm glue launches of 4.8 us per layer doing nothing but a norm on 8 rows).
// At <= 16 rows the glue's whole input - the S_in split-K slabs of the producing GEMM and the residual stream - is 41 KB per slab and
// 20 KB at d = 1280: small enough for EVERY block of the consuming GEMM to rebuild it (out of L2 / the Infinity Cache) instead of a
// launch of its own doing it once.  Per block: the 8 waves first request their weight tiles (HBM, the long trip), then wave w rebuilds
// row w (and w + 8): h_new = T(h + T(sum slabs)) over the WHOLE row (the statistics need it; wave-local DPP sums, no block barrier),
// LayerNorm / RMSNorm with the glue's rounding points, and writes the columns of the block's own K range as MFMA-B fragments into LDS;
// one barrier; MFMAs with x from LDS; the waves' partials meet in LDS like k_gemm_skinny's.  The blocks of n-tile group 0 also write
// h_new (their K ranges cover the row) into the OTHER residual buffer - the old one is still being read by every other block.
// Same arithmetic per element as k_glue4 (slab order 0, 1, ..; float32 statistics - summed in another order, like k_glue_cpt).
// (At 32 rows x d = 1024 the same fold lost in round 3 - profiles/r03/q3: four times the bytes per block and a block-wide reduction in
// front of the first MFMA; here the rows are few enough for one wave per row and the weight trip hides the prologue.)
// Measured at Whisper-large-v3's decoder (profiles/r06/c2, c3): fc1 7.0 + glue 4.6 -> 9.6 us under the profiler, transcribe -1 %: kept for
// fc1; the q|k|v GEMM behind 8 slabs (5.1 + 4.7 -> 10.0 us) and the cross query (superseded by the fold into the attention kernel) are not
// instantiated.
template <int R, int EPI, bool LN, int SG, int KW, int G>
__global__ void __launch_bounds__(512, 1) k_gemm_skinny_norm(const bf16_t* __restrict__ Wp, const float* __restrict__ slabs, int S_in,
                                                             const bf16_t* __restrict__ h_in, bf16_t* __restrict__ h_out,
                                                             const bf16_t* __restrict__ wnorm, const bf16_t* __restrict__ ln_bias, float eps,
                                                             int nrows, void* __restrict__ out, int NT, int KT, int S, int n_items, int N_out,
                                                             const bf16_t* __restrict__ bias) {
    constexpr int NW = 8, Mpad = 16;
    const int N = KT * 32;                                // the norm's width = this GEMM's K
    extern __shared__ __attribute__((aligned(16))) unsigned char gn_lds[];
    bf16x8_t* xs = reinterpret_cast<bf16x8_t*>(gn_lds);                     // [k-tiles of the block][64] B fragments (swizzled rows)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x;
    const int ntg = item / S, ks = item - ntg * S;
    const int kb0 = (int)(((long long)KT * ks) / S), kb1 = (int)(((long long)KT * (ks + 1)) / S);
    const int len = kb1 - kb0;
    const int kt0 = kb0 + (len * wave) / NW, kt1 = kb0 + (len * (wave + 1)) / NW;
    // ---- this wave's weight tiles, requested before anything else (clamped: a dead tile is multiplied by a zero fragment)
    bf16x8_t wt[KW][R];
#pragma unroll
    for (int u = 0; u < KW; ++u) {
        int kk = kt0 + u;
        kk = kk < kb1 ? kk : kb1 - 1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int tile = ntg * R + r;
            tile = tile < NT ? tile : NT - 1;
            wt[u][r] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(Wp) + ((size_t)tile * KT + kk) * 64 + lane);
        }
    }
    // ---- the glue: wave w -> rows w, w + 8; lane -> column groups 4 (lane + 64 g) .. + 3
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int m = wave + NW * rr;
        if (m >= nrows) {                                 // padding rows: zero fragments
            for (int g = 0; g < G; ++g) {
                const int col = 4 * (lane + 64 * g), kt = col >> 5;
                if (col < N && kt >= kb0 && kt < kb1) {
                    const int k8 = (col & 31) >> 3, swz = (kt * 4 + k8) & 15;
                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(xs) + ((size_t)((kt - kb0) * 64 + k8 * 16 + (m ^ swz)) * 8 + (col & 7))) = make_uint2(0u, 0u);
                }
            }
            continue;
        }
        uint2 hq[G], wq[G], bq[G];
        f32x4_t v[SG][G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            int col = 4 * (lane + 64 * g);
            col = col < N ? col : N - 4;
            hq[g] = *reinterpret_cast<const uint2*>(h_in + (size_t)m * N + col);
            wq[g] = *reinterpret_cast<const uint2*>(wnorm + col);
            if constexpr (LN) bq[g] = *reinterpret_cast<const uint2*>(ln_bias + col);
#pragma unroll
            for (int j = 0; j < SG; ++j) {
                const int sj = j < S_in ? j : S_in - 1;
                v[j][g] = *reinterpret_cast<const f32x4_t*>(slabs + ((size_t)sj * Mpad + m) * N + col);
            }
        }
        float hn[G][4];
        float sum = 0.f, ss = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const bool live = 4 * (lane + 64 * g) < N;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < SG; ++j) {                // slab order 0, 1, 2, ... (fixed => deterministic)
                const bool on = j < S_in;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += on ? v[j][g][e] : 0.0f;
            }
            const float hv[4] = {bf16_to_f32((bf16_t)(hq[g].x & 0xffffu)), bf16_to_f32((bf16_t)(hq[g].x >> 16)), bf16_to_f32((bf16_t)(hq[g].y & 0xffffu)),
                                 bf16_to_f32((bf16_t)(hq[g].y >> 16))};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hn[g][e] = bf16_round_f32(hv[e] + bf16_round_f32(acc[e]));          // o = T(sum slabs); h = T(h + o)
                sum += live ? hn[g][e] : 0.0f;
                ss += live ? hn[g][e] * hn[g][e] : 0.0f;
            }
        }
        float mean = 0.f, scale;
        if constexpr (LN) {
            mean = wave_sum_dpp(sum) / (float)N;
            float sq = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const bool live = 4 * (lane + 64 * g) < N;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float dl = hn[g][e] - mean; sq += live ? dl * dl : 0.0f; }
            }
            scale = 1.0f / sqrtf(wave_sum_dpp(sq) / (float)N + eps);
        } else {
            scale = 1.0f / sqrtf(wave_sum_dpp(ss) / (float)N + eps);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int col = 4 * (lane + 64 * g), kt = col >> 5;
            if (col < N && kt >= kb0 && kt < kb1) {
                const float wv[4] = {bf16_to_f32((bf16_t)(wq[g].x & 0xffffu)), bf16_to_f32((bf16_t)(wq[g].x >> 16)), bf16_to_f32((bf16_t)(wq[g].y & 0xffffu)),
                                     bf16_to_f32((bf16_t)(wq[g].y >> 16))};
                bf16_t xb[4], hb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (LN) {
                        const float bv = bf16_to_f32((bf16_t)(e < 2 ? (bq[g].x >> (16 * e)) & 0xffffu : (bq[g].y >> (16 * (e - 2))) & 0xffffu));
                        xb[e] = f32_to_bf16((hn[g][e] - mean) * scale * wv[e] + bv);
                    } else {
                        xb[e] = f32_to_bf16(wv[e] * bf16_round_f32(hn[g][e] * scale));
                    }
                    hb[e] = f32_to_bf16(hn[g][e]);
                }
                const int k8 = (col & 31) >> 3, swz = (kt * 4 + k8) & 15;
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(xs) + ((size_t)((kt - kb0) * 64 + k8 * 16 + (m ^ swz)) * 8 + (col & 7))) =
                    make_uint2((uint32_t)xb[0] | ((uint32_t)xb[1] << 16), (uint32_t)xb[2] | ((uint32_t)xb[3] << 16));
                if (ntg == 0)                              // the n-tile group 0 blocks (one per K slice) keep the residual stream
                    *reinterpret_cast<uint2*>(h_out + (size_t)m * N + col) =
                        make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
            }
        }
    }
    __syncthreads();
    // ---- MFMAs: x fragments from LDS (row m of lane l sits at l ^ swizzle inside its 16-lane group)
    f32x4_t acc[R][1];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KW; ++u) {
        const int kt = kt0 + u;
        const int ktc = kt < kb1 ? kt : kb1 - 1;
        const int k8 = lane >> 4, swz = (ktc * 4 + k8) & 15;
        bf16x8_t xf = xs[(size_t)(ktc - kb0) * 64 + k8 * 16 + ((lane & 15) ^ swz)];
        if (kt >= kt1) xf = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[u][r], xf, acc[r][0], 0, 0, 0);
    }
    float4* red = reinterpret_cast<float4*>(gn_lds + (size_t)len * 64 * 16);      // [NW][R][64], behind the fragments
#pragma unroll
    for (int r = 0; r < R; ++r) red[(wave * R + r) * 64 + lane] = make_float4(acc[r][0][0], acc[r][0][1], acc[r][0][2], acc[r][0][3]);
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float4 s0 = red[r * 64 + lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) {                // fixed order (deterministic)
                const float4 t = red[(w * R + r) * 64 + lane];
                s0.x += t.x; s0.y += t.y; s0.z += t.z; s0.w += t.w;
            }
            acc[r][0] = (f32x4_t){s0.x, s0.y, s0.z, s0.w};
        }
        gemm_epilogue<1, R, EPI>(acc, out, ntg, ks, NT, N_out, Mpad, lane, -1, bias);
    }
}

bool gemm_skinny_norm_ok(int epi, int R, int KT, int S, int S_in, int Mpad, int nrows) {
    const int N = KT * 32;
    return Mpad == 16 && nrows >= 1 && nrows <= 16 && R == 2 && (epi == EPI_PARTIAL || epi == EPI_GELU_PACKED) && N % 4 == 0 && N <= 1280 && S_in >= 1 &&
           S_in <= 8 && S >= 1 && S <= KT && ((KT + S - 1) / S + 7) / 8 <= 5;
}
void launch_gemm_skinny_norm(int epi, int R, const bf16_t* Wp, const float* slabs, int S_in, const bf16_t* h_in, bf16_t* h_out, const bf16_t* wnorm,
                             const bf16_t* ln_bias, float eps, int nrows, void* out, int NT, int KT, int S, int N_out, int Mpad, hipStream_t s,
                             const bf16_t* bias) {
    MIS_REQUIRE(gemm_skinny_norm_ok(epi, R, KT, S, S_in, Mpad, nrows), MIS_ERR_GENERATION_FAILED, "unsupported norm-in-prologue GEMM shape");
    MIS_REQUIRE(epi == EPI_PARTIAL || S == 1, MIS_ERR_GENERATION_FAILED, "split-K needs the partial epilogue");
    const int n_items = ((NT + R - 1) / R) * S;
    const int len_max = (KT + S - 1) / S;                 // k-tiles of the longest block range
    const int kw = (len_max + 7) / 8;                     // k-tiles of the longest wave range
    const size_t lds = (size_t)len_max * 64 * 16 + (size_t)8 * R * 64 * 16;
    const bool ln = ln_bias != nullptr;
#define GN_CASE(E, LNB, SGv, KWv)                                                                                                         \
    if (epi == E && ln == LNB && S_in <= SGv && kw <= KWv) {                                                                              \
        hipLaunchKernelGGL((k_gemm_skinny_norm<2, E, LNB, SGv, KWv, 5>), dim3(n_items), dim3(512), lds, s, Wp, slabs, S_in, h_in, h_out, wnorm,  \
                           ln_bias, eps, nrows, out, NT, KT, S, n_items, N_out, bias);                                                    \
        return;                                                                                                                           \
    }
    GN_CASE(EPI_GELU_PACKED, true, 4, 5)                  // (the instantiations the product launches: Whisper's fc1 behind LayerNorm 3)
    GN_CASE(EPI_GELU_PACKED, true, 8, 5)
#undef GN_CASE
    throw MisError(MIS_ERR_GENERATION_FAILED, "unsupported norm-in-prologue GEMM variant");
}

// Inter-block split-K factor of a weight-streaming GEMM with `items` n-tile groups and KT k-tiles.  All blocks of a launch are
// co-resident and share their CU's load bandwidth, so the launch takes as long as the most loaded CU needs:
//   time ~ ceil(blocks / CUs) / blocks            (the share of the weight bytes behind the busiest CU)
//        x loaded / needed k-tiles                (waves load whole groups of GEMM_U = 4 k-tiles; a ragged range re-reads its tail)
//        x (1 + 1 % per slab)                     (partial slabs written here and summed by the consumer).
// This reproduces the measured optimum of every shape in profiles/r01_v1..v5 gemm sweeps (Orpheus-3B, batch 32: qkv S = 3,
// o_proj S = 2, down S = 8); the former "about 800 blocks" rule picked S = 5 for qkv (4.8 k-tiles per wave: 67 % extra loads).
int gemm_choose_split(int items, int KT, int ksb, int s_max) {
    static const int n_cu = [] {
        hipDeviceProp_t prop{};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) return 256;
        return prop.multiProcessorCount;
    }();
    const int U = 4;                                  // GEMM_U (lm_kernels.hip)
    int best = 1;
    double best_cost = 1e30;
    const int limit = std::max(1, std::min(std::min(s_max, 16), KT / (2 * ksb)));
    for (int S = 1; S <= limit; ++S) {
        const long blocks = ksb == 1 ? ((long)items * S + 3) / 4 : (long)items * S;
        const int waves_per_item = ksb == 1 ? 1 : ksb;
        const int kw = (KT + S * waves_per_item - 1) / (S * waves_per_item);      // k-tiles of the longest wave range
        const int loaded = (kw + U - 1) / U * U;
        const double cost = (double)((blocks + n_cu - 1) / n_cu) / (double)blocks * ((double)loaded * S * waves_per_item / KT) *
                            (1.0 + 0.01 * S);
        if (cost < best_cost * (1.0 - 1e-9)) { best_cost = cost; best = S; }
    }
    return best;
}

// ============================================================================ decode attention
//
// One 512-thread block (8 waves) per (kv head, row).  Prologue: reduce the QKV split-K slabs, round to
// bf16, RoPE q and k at the row's position (tables), append k / v to the caches.  Main loop: each wave
// walks 32-key tiles (w, w+8, ...):  S^T = K . Q^T on MFMA (A = K tile with a key permutation that makes
// the score registers line up with the P.V A-operand, B = Q^T), online softmax in f32, O += P . V with
// P split into bf16 hi + lo parts (f32-accurate probabilities) and V^T tiles as B operand.  Epilogue:
// cross-wave log-sum-exp combine through LDS.   (LlamaTTS.swift:235-266; SDPA semantics: oracle/llama.py)
// Measured alternative (round 2, removed again): waves 1..7 request their first tile at once while WAVE 0 ALONE runs the whole
// prologue barrier-free, the second tile of a pair requested after the single barrier (counted wait in front of the first tile's
// math).  Parity-green (103 GPU tests) and slower: 14.24 -> 14.58 us at context 368, bench 173.9 -> 171.1 audio-s/s
// (profiles/r02_attn_w0_ab.json): the 512-thread prologue is NOT what the K/V stream waits for.

#define ATT_WAVES 8

// phase timestamps of block (0, 0) for tools/attn_phases.py (compiled in with -DMIS_ATTN_TIMING only)
#ifdef MIS_ATTN_TIMING
#define ATT_STAMP(i) do { if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.dbg[i] = __builtin_readcyclecounter(); } while (0)
#else
#define ATT_STAMP(i) do { } while (0)
#endif

// XS ("cross stream", round 4): the schedule for cross-attention over a LONG static cache (Whisper: 1500 keys = 47 tiles, five or six per
// wave).  The loop below requests a pair of tiles, waits for it and multiplies - per wave two or three dependent round trips, of which
// only the first hides behind the prologue.  With XS a wave holds TWO pairs in registers (D = 64: 4 x 32 registers): while one group is
// multiplied the next pair streams, and the pair behind that is requested before the wait for it.  An odd tile count is taken out up
// front (the first group is ONE tile, every later group a full pair); the code is straight-line per tile count (see the main loop for
// why not a loop).  Tiles are multiplied in the same order as without XS - results are bit-identical.  The K/V loads are
// non-temporal: a decode step reads each of them once (61 MB per layer at eight windows, 2 GB per step - nothing a cache can keep).
// QP ("query projection", round 6; Whisper cross-attention at <= 16 rows): the residual add + LayerNorm glue in front of the
// cross-attention AND its query projection run inside this kernel's prologue instead of as two launches of their own (k_glue4 4.6 us +
// k_gemm_skinny 5.1 us per layer under the profiler).  A block is one (row, head): it needs ONE row of the glue - S_in slabs x 5 KB at
// d = 1280, rebuilt by each of its eight waves with wave-local DPP sums (no barrier) - and 64 rows of W_q (164 KB out of L2: the eight
// rows' blocks of a head share them).  Wave w owns k-tiles [KT w / 8, KT (w + 1) / 8): it rebuilds THAT slice of the row (<= 160 columns, one
// group of four per lane), the row statistics meet in LDS (two raw barriers: mean, then variance - the glue's two-pass form), and it
// multiplies its slice with the head's four n-tiles on the matrix core (B = the row as column 0); the eight partial vectors meet in LDS
// behind the barrier the prologue has anyway.  The W_q tiles, the slice and the wave's first K/V tile are requested together, the second
// tile behind the projection.  The head-0 blocks write h_new to the OTHER residual buffer.  Contrast with k_gemm_skinny_norm (same fold
// on the GEMM side, measured neutral: every one of its 160-480 blocks re-reads ALL rows' slabs): a row-parallel consumer re-reads its row.
// (First form, profiles/r06/c4: every wave rebuilt the WHOLE row and the K/V request waited behind the projection: 13.5 -> 17.8 us.)
#define ATT_QP_KW 5
#define ATT_QP_LDS ((size_t)ATT_WAVES * ATT_QP_KW * 32 * 2 + (size_t)ATT_WAVES * 64 * 4 + 2 * ATT_WAVES * 4)
#define ATT_XS_MIN_J 5
#define ATT_XS_MAX_J 6
template <int D, int NIT, bool XS = false, int QP = 0>          // QP = slabs fetched per trip by the query-projection prologue (0: off)
__global__ void __launch_bounds__(512) k_attn_decode(AttnParams p) {
    static_assert(QP == 0 || (XS && D == 64), "QP: the cross-attention schedule at head_dim 64");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int kvh = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = p.H / p.Hkv;
    ATT_STAMP(0);

    // LDS carve-up
    float* sraw = reinterpret_cast<float*>(smem);                      // [(G+2)][D] f32, padded to NIT*512 (see the slab sum)
    bf16_t* qs = reinterpret_cast<bf16_t*>(sraw + NIT * 512);          // [16][D] bf16
    bf16_t* ksh = qs + 16 * D;                                         // [D] bf16: the new key after RoPE (patched into its tile)
    float* sm = reinterpret_cast<float*>(ksh + D);                     // [W][16]
    float* sl = sm + ATT_WAVES * 16;                                   // [W][16]
    float* sO = sl + ATT_WAVES * 16;                                   // [W][G][D]

    // ---- scalar round trip: the row's state.  Both addresses are wave-uniform, so these are s_load (scalar cache, lgkmcnt) - they
    // neither queue behind nor hold up the vector loads below.  `active` is read as the aligned 32-bit word holding byte b (there is
    // no sub-dword scalar load on gfx950).  With pos known up front, the RoPE table rows join round trip 1 and nothing after it
    // depends on memory any more - which is what lets the K/V stream overlap the whole prologue (see below).
    const uint32_t active_word = reinterpret_cast<const uint32_t*>(p.active)[b >> 2];
    const unsigned char row_active = (unsigned char)((active_word >> (8 * (b & 3))) & 0xffu);
    const int pos = p.cross ? 0 : p.pos[b];
    if (!row_active) return;
    const int kv_len = p.cross ? p.cross_len : pos + 1;

    // ---- memory round trip 1 (vector): the split-K slabs of this (row, kv head), the q/k-norm weights and the RoPE table rows of
    // this position.  All loads are unconditional (clamped addresses, masked at use): a guarded load compiles to a branch with its
    // own s_waitcnt vmcnt(0), i.e. one more dependent round trip per guard.
    const int n_pro = p.cross ? G : G + 2;          // cross attention: queries only (K/V cached once per utterance)
    const int n_el = n_pro * D;
    float pv[NIT][8];                                // NIT * 512 >= (G + 2) * D
    float* qp_part = sO + (size_t)ATT_WAVES * G * D + (size_t)ATT_WAVES * ATT_QP_KW * 16;      // [W][64] behind the waves' x strips (QP only)
    float* qp_red = qp_part + ATT_WAVES * 64;                                                  // [2][W] row statistics
    // QP, round trip 1: this wave's W_q tiles (k-tiles [k0, k1) of the head's four n-tiles) and its slice of the row
    [[maybe_unused]] bf16x8_t qp_wt[QP > 0 ? ATT_QP_KW : 1][4];
    [[maybe_unused]] uint2 qp_hq, qp_wq, qp_bq;
    [[maybe_unused]] f32x4_t qp_v[QP > 0 ? QP : 1];
    [[maybe_unused]] int qp_k0 = 0, qp_k1 = 0, qp_col = 0;
    if constexpr (QP > 0) {
        const int wq_ = __builtin_amdgcn_readfirstlane(wave);
        const int KT = p.qp_KT, N = KT * 32;
        qp_k0 = (KT * wq_) / ATT_WAVES; qp_k1 = (KT * (wq_ + 1)) / ATT_WAVES;
#pragma unroll
        for (int u = 0; u < ATT_QP_KW; ++u) {
            int kk = qp_k0 + u;
            kk = kk < KT ? kk : KT - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r)      // (cached loads: the other rows' blocks of this head read the same tiles)
                qp_wt[u][r] = *(reinterpret_cast<const bf16x8_t*>(p.qp_w) + ((size_t)(kvh * 4 + r) * KT + kk) * 64 + lane);
        }
        qp_col = 32 * qp_k0 + 4 * lane;                                  // one group of four columns per lane (<= 160 per wave)
        const int colc = qp_col < 32 * qp_k1 ? qp_col : N - 4;
        qp_hq = *reinterpret_cast<const uint2*>(p.qp_h_in + (size_t)b * N + colc);
        qp_wq = *reinterpret_cast<const uint2*>(p.qp_lnw + colc);
        qp_bq = *reinterpret_cast<const uint2*>(p.qp_lnb + colc);
#pragma unroll
        for (int j = 0; j < QP; ++j) {
            const int sj = j < p.qp_S ? j : p.qp_S - 1;
            qp_v[j] = *reinterpret_cast<const f32x4_t*>(p.qp_slabs + ((size_t)sj * p.Mpad + b) * N + colc);
        }
    }
#pragma unroll
    for (int it = 0; it < (QP > 0 ? 0 : NIT); ++it) {
        const int idx = tid + it * 512;
        int hh = idx / D, d = idx - hh * D;
        if (idx >= n_el) { hh = 0; d = 0; }          // clamped: loaded, never used
        int col;
        if (hh < G) col = (kvh * G + hh) * D + d;
        else if (hh == G) col = p.H * D + kvh * D + d;
        else col = p.H * D + p.Hkv * D + kvh * D + d;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            pv[it][j] = p.qkv_part[((size_t)(j < p.S ? j : p.S - 1) * p.Mpad + b) * p.Nqkv + col];
    }
    float qw[D / 64], kw[D / 64];
    {
        const bf16_t* qp = p.qnorm_w ? p.qnorm_w : reinterpret_cast<const bf16_t*>(p.qkv_part);
        const bf16_t* kp = p.qnorm_w ? p.knorm_w : reinterpret_cast<const bf16_t*>(p.qkv_part);
#pragma unroll
        for (int j = 0; j < D / 64; ++j) { qw[j] = bf16_to_f32(qp[p.qnorm_w ? lane + 64 * j : 0]); kw[j] = bf16_to_f32(kp[p.qnorm_w ? lane + 64 * j : 0]); }
    }
    const int n_rot_el = (p.cross ? G : G + 1) * (D / 2);
    float rc[NIT], rs[NIT];
    {
        const float* ct = p.rope_cos ? p.rope_cos + (size_t)pos * (D / 2) : p.qkv_part;
        const float* st = p.rope_cos ? p.rope_sin + (size_t)pos * (D / 2) : p.qkv_part;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 512;
            const int i = idx % (D / 2);
            rc[it] = ct[p.rope_cos ? i : 0];          // unconditional; entries past n_rot_el are never used
            rs[it] = st[p.rope_cos ? i : 0];
        }
    }
    // one wait covers the whole round trip: everything is pinned here, BEFORE the K/V stream is requested, so that no later use of
    // these values has to wait behind 32 KiB of K/V per wave (vmcnt retires in issue order, and the wave-uniform guards around the
    // tile loads make the compiler's waitcnt at the join conservative - before this ordering the RoPE step waited for the wave's
    // whole first tile pair, i.e. at contexts <= 512 for the entire KV stream of the launch)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if constexpr (QP == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(pv[it][j]));
        }
        asm volatile("" : "+v"(rc[it]));
        asm volatile("" : "+v"(rs[it]));
    }
#pragma unroll
    for (int j = 0; j < D / 64; ++j) { asm volatile("" : "+v"(qw[j])); asm volatile("" : "+v"(kw[j])); }
    ATT_STAMP(1);

    // ---- the K/V stream: the wave's first pair of tiles (32 KiB per wave) is requested now and lands while the prologue below
    // (slab reduce, q/k norm, RoPE, cache append, LDS staging, two barriers) runs - INCLUDING the tile the new key belongs to: its
    // slot is patched in registers from LDS below, so no load ever waits for this step's own cache append.
    const int cb = p.cache_rows ? b % p.cache_rows : b;            // batched prefill: rows are (position, sequence) pairs
    bf16_t* kc = p.kcache + ((size_t)(cb * p.Hkv + kvh) * p.Smax) * D;
    bf16_t* vt = p.vtcache + ((size_t)(cb * p.Hkv + kvh) * D) * p.Smax;
    const bf16x8_t* kbase = reinterpret_cast<const bf16x8_t*>(kc) + lane;
    const bf16x8_t* vbase = reinterpret_cast<const bf16x8_t*>(vt) + lane;
    const int n_tiles = p.append_only ? 0 : (kv_len + 31) >> 5;    // append-only launches request no tiles
    const int new_tile = p.cross ? -1 : (pos >> 5);
    bf16x8_t kA[2][D / 32], kB[2][D / 32], vA[D / 16], vB[D / 16];
    if constexpr (XS && QP == 0) {                  // (an odd first group leaves B unrequested; the binding statement below names it)
        const bf16x8_t z = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < D / 32; ++c) { kB[0][c] = z; kB[1][c] = z; }
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) vB[dt] = z;
    }
    auto load_tile = [&](int tile, bf16x8_t (&ka)[2][D / 32], bf16x8_t (&vb)[D / 16]) {
        if constexpr (XS) {
#pragma unroll
            for (int c = 0; c < D / 32; ++c) {
                ka[0][c] = __builtin_nontemporal_load(kbase + ((size_t)tile * 2) * (D / 32) * 64 + c * 64);
                ka[1][c] = __builtin_nontemporal_load(kbase + ((size_t)tile * 2 + 1) * (D / 32) * 64 + c * 64);
            }
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) vb[dt] = __builtin_nontemporal_load(vbase + ((size_t)tile * (D / 16) + dt) * 64);
        } else {
#pragma unroll
            for (int c = 0; c < D / 32; ++c) {
                ka[0][c] = kbase[((size_t)tile * 2) * (D / 32) * 64 + c * 64];
                ka[1][c] = kbase[((size_t)tile * 2 + 1) * (D / 32) * 64 + c * 64];
            }
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) vb[dt] = vbase[((size_t)tile * (D / 16) + dt) * 64];
        }
    };
    // XS: tiles of this wave (wave, wave + 8, ...); an odd count is taken out up front so that every later group is a full pair
    // (the wave index as an SGPR: with `tid >> 6` the compiler treats every branch on it as divergent, lowers the loop below through
    // exec masks and re-joins its arms - with a vmcnt(0) at the join)
    const int wu = XS ? __builtin_amdgcn_readfirstlane(wave) : wave;
    const int xs_nj = XS && wu < n_tiles ? (n_tiles - wu + ATT_WAVES - 1) / ATT_WAVES : 0;
    const bool xs_odd = (xs_nj & 1) != 0;
    __builtin_amdgcn_sched_barrier(0);
    // wave-uniform guards: a wave without a tile requests nothing (at short contexts seven of eight waves would otherwise each
    // pull a redundant 32 KB pair through the CU's 64 B/clk return path - 2-3 us per launch on the small models)
    if constexpr (QP > 0) {
        // the first tile UNCONDITIONALLY (the launcher admits >= ATT_XS_MIN_J tiles per wave): no join between the requests above and the
        // projection below, so its waits count loads instead of draining them; the second tile of an even count follows the projection
        load_tile(wu, kA, vA);
        __builtin_amdgcn_sched_barrier(0);
        const int N = p.qp_KT * 32;
        const bool live = qp_col < 32 * qp_k1;
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < QP; ++j) {                    // slab order 0, 1, 2, ... (fixed => deterministic)
            const bool on = j < p.qp_S;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc4[e] += on ? qp_v[j][e] : 0.0f;
        }
        const float hv[4] = {bf16_to_f32((bf16_t)(qp_hq.x & 0xffffu)), bf16_to_f32((bf16_t)(qp_hq.x >> 16)), bf16_to_f32((bf16_t)(qp_hq.y & 0xffffu)),
                             bf16_to_f32((bf16_t)(qp_hq.y >> 16))};
        float hn[4], sum = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hn[e] = bf16_round_f32(hv[e] + bf16_round_f32(acc4[e]));                // o = T(sum slabs); h = T(h + o)
            sum += live ? hn[e] : 0.0f;
        }
        sum = wave_sum_dpp(sum);
        if (lane == 0) qp_red[wu] = sum;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // (raw: __syncthreads() would wait for the K/V tile as well)
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) tot += qp_red[w];
        const float mean = tot / (float)N;
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float dl = hn[e] - mean; sq += live ? dl * dl : 0.0f; }
        sq = wave_sum_dpp(sq);
        if (lane == 0) qp_red[ATT_WAVES + wu] = sq;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float tq = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) tq += qp_red[ATT_WAVES + w];
        const float rstd = 1.0f / sqrtf(tq / (float)N + p.qp_eps);
        bf16_t* xrow = reinterpret_cast<bf16_t*>(sO + (size_t)ATT_WAVES * G * D) + (size_t)wu * ATT_QP_KW * 32;
        if (live) {
            const float wv[4] = {bf16_to_f32((bf16_t)(qp_wq.x & 0xffffu)), bf16_to_f32((bf16_t)(qp_wq.x >> 16)), bf16_to_f32((bf16_t)(qp_wq.y & 0xffffu)),
                                 bf16_to_f32((bf16_t)(qp_wq.y >> 16))};
            const float bv[4] = {bf16_to_f32((bf16_t)(qp_bq.x & 0xffffu)), bf16_to_f32((bf16_t)(qp_bq.x >> 16)), bf16_to_f32((bf16_t)(qp_bq.y & 0xffffu)),
                                 bf16_to_f32((bf16_t)(qp_bq.y >> 16))};
            bf16_t xb[4], hb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xb[e] = f32_to_bf16((hn[e] - mean) * rstd * wv[e] + bv[e]);
                hb[e] = f32_to_bf16(hn[e]);
            }
            *reinterpret_cast<uint2*>(xrow + 4 * lane) = make_uint2((uint32_t)xb[0] | ((uint32_t)xb[1] << 16), (uint32_t)xb[2] | ((uint32_t)xb[3] << 16));
            if (kvh == 0)                                 // the head-0 block of the row keeps the residual stream (its waves cover the row)
                *reinterpret_cast<uint2*>(p.qp_h_out + (size_t)b * N + qp_col) =
                    make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the strip is this wave's own: LDS operations of one wave execute in order)
        f32x4_t qacc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) qacc[r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const bf16x8_t z = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < ATT_QP_KW; ++u) {
            bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(xrow + u * 32 + (lane >> 4) * 8);
            if ((lane & 15) != 0 || qp_k0 + u >= qp_k1) xf = z;         // B = the row as column m = 0; dead k-tiles multiply zero
#pragma unroll
            for (int r = 0; r < 4; ++r) qacc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qp_wt[u][r], xf, qacc[r], 0, 0, 0);
        }
        if ((lane & 15) == 0) {                               // C/D lane l, reg e: n = (l >> 4) * 4 + e, m = l & 15
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<float4*>(qp_part + wu * 64 + r * 16 + (lane >> 4) * 4) = make_float4(qacc[r][0], qacc[r][1], qacc[r][2], qacc[r][3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        {                                               // (B's registers are free of the W_q tiles only now)
            const bf16x8_t zb = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < D / 32; ++c) { kB[0][c] = zb; kB[1][c] = zb; }
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) vB[dt] = zb;
        }
        if (xs_nj >= 2 && !xs_odd) load_tile(wu + ATT_WAVES, kB, vB);
    } else if constexpr (XS) {
        if (xs_nj >= 1) load_tile(wu, kA, vA);
        if (xs_nj >= 2 && !xs_odd) load_tile(wu + ATT_WAVES, kB, vB);
    } else {
        if (wave < n_tiles) load_tile(wave, kA, vA);
        if (wave + ATT_WAVES < n_tiles) load_tile(wave + ATT_WAVES, kB, vB);
    }
    __builtin_amdgcn_sched_barrier(0);
    ATT_STAMP(2);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 512;
        // unconditional (sraw is padded to NIT*512 entries; entries >= n_el hold clamped-address garbage nobody reads)
        float acc = 0.0f;
        if constexpr (QP == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (j < p.S) ? pv[it][j] : 0.0f;   // slab order 0..7 (S <= 8, checked by the launcher)
        }
        sraw[idx] = bf16_round_f32(acc);
    }
    for (int idx = tid; idx < 16 * D; idx += 512) qs[idx] = 0;
    __syncthreads();
    ATT_STAMP(3);
    if (p.qnorm_w) {
        // Qwen3-style per-head RMSNorm of q and k BEFORE RoPE (Soprano.swift:75-76): n = T(x rsqrt(mean x^2 + eps)), T(w n)
        const int n_rows = p.cross ? G : G + 1;
        for (int row = wave; row < n_rows; row += ATT_WAVES) {
            float ss = 0.0f;
            for (int d = lane; d < D; d += 64) { float v = sraw[row * D + d]; ss += v * v; }
            ss = wave_sum(ss);
            const float inv = 1.0f / sqrtf(ss / (float)D + p.qk_eps);
#pragma unroll
            for (int j = 0; j < D / 64; ++j) {
                const int d = lane + 64 * j;
                sraw[row * D + d] = bf16_round_f32((row < G ? qw[j] : kw[j]) * bf16_round_f32(sraw[row * D + d] * inv));
            }
        }
        __syncthreads();
    }
    // ---- RoPE (rotate-half, pair (i, i + D/2), angle pos / freqs[i]; LlamaTTS.swift:192-200) + cache append
    // tiled cache addressing of the new key `pos` (see header): 32-key tile, A-fragment row i, half hf
    const int ptile = pos >> 5, pr = pos & 31;
    const int prow = ((pr >> 3) << 2) | (pr & 3), phalf = (pr >> 2) & 1;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 512;
        if (idx >= n_rot_el) continue;
        int hh = idx / (D / 2), i = idx - hh * (D / 2);
        float x1 = sraw[hh * D + i], x2 = sraw[hh * D + i + D / 2];
        if constexpr (QP > 0) {                      // (cross-attention, G = 1: hh = 0) q = T(W_q x + b): the waves' partials in wave order
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w = 0; w < ATT_WAVES; ++w) { a1 += qp_part[w * 64 + i]; a2 += qp_part[w * 64 + i + D / 2]; }
            x1 = bf16_round_f32(a1 + (p.qp_bias ? bf16_to_f32(p.qp_bias[kvh * D + i]) : 0.0f));
            x2 = bf16_round_f32(a2 + (p.qp_bias ? bf16_to_f32(p.qp_bias[kvh * D + i + D / 2]) : 0.0f));
        }
        bf16_t r1, r2;
        if (p.rope_cos) {
            float c = rc[it], s = rs[it];
            if (p.rope_in_dtype) {
                c = bf16_round_f32(c); s = bf16_round_f32(s);
                r1 = f32_to_bf16(bf16_round_f32(x1 * c) + bf16_round_f32(-x2 * s));
                r2 = f32_to_bf16(bf16_round_f32(x2 * c) + bf16_round_f32(x1 * s));
            } else { r1 = f32_to_bf16(x1 * c - x2 * s); r2 = f32_to_bf16(x1 * s + x2 * c); }
        } else { r1 = f32_to_bf16(x1); r2 = f32_to_bf16(x2); }      // Whisper: learned positions, no rotary
        if (hh < G) { qs[hh * D + i] = r1; qs[hh * D + i + D / 2] = r2; }
        else {
            int d1 = i, d2 = i + D / 2;
            ksh[d1] = r1; ksh[d2] = r2;        // for the register patch of the new key's tile; the cache append is fire-and-forget
            kc[((((size_t)ptile * 2 + phalf) * (D / 32) + (d1 >> 5)) * 64 + (((d1 & 31) >> 3) << 4) + prow) * 8 + (d1 & 7)] = r1;
            kc[((((size_t)ptile * 2 + phalf) * (D / 32) + (d2 >> 5)) * 64 + (((d2 & 31) >> 3) << 4) + prow) * 8 + (d2 & 7)] = r2;
        }
    }
    if (!p.cross)
        for (int d = tid; d < D; d += 512)
            vt[(((size_t)ptile * (D / 16) + (d >> 4)) * 64 + ((pr >> 3) << 4) + (d & 15)) * 8 + (pr & 7)] =
                f32_to_bf16(sraw[(G + 1) * D + d]);
    __syncthreads();       // LDS q / new key visible (nobody reads this step's K/V append back from memory)
    if (p.append_only) return;
    ATT_STAMP(4);
    // the new key (position pos) inside its tile's fragments: K row `prow` of half `phalf`, V^T column pr (see the cache tiling)
    auto patch_new_key = [&](bf16x8_t (&ka)[2][D / 32], bf16x8_t (&vb)[D / 16]) {
        if ((lane & 15) == prow) {
#pragma unroll
            for (int c = 0; c < D / 32; ++c) {
                const bf16x8_t kn = *reinterpret_cast<const bf16x8_t*>(ksh + c * 32 + (lane >> 4) * 8);
                if (phalf) ka[1][c] = kn; else ka[0][c] = kn;
            }
        }
        if ((lane >> 4) == (pr >> 3)) {
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) {
                const short vn = (short)f32_to_bf16(sraw[(G + 1) * D + dt * 16 + (lane & 15)]);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (e == (pr & 7)) vb[dt][e] = vn;
            }
        }
    };

    // ---- main loop
    const int h = lane & 15, g4 = lane >> 4;
    bf16x8_t qf[D / 32];
#pragma unroll
    for (int c = 0; c < D / 32; ++c) qf[c] = *reinterpret_cast<const bf16x8_t*>(qs + h * D + c * 32 + g4 * 8);
    f32x4_t O[D / 16];
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) O[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.0f;
    // MFMA row i <-> key base + (i>>2)*8 + (i&3) (+4 for the second half): baked into the cache tiling.
    // Each wave owns tiles (wave + 8j); they are processed in PAIRS with all K and V fragments of both tiles
    // (32 KiB per wave) requested before the first MFMA, so a typical context (<= 512 keys) costs one memory
    // round trip per wave instead of four dependent ones.
    auto process = [&](int tile, const bf16x8_t (&ka)[2][D / 32], const bf16x8_t (&vb)[D / 16]) {
        const int base = tile * 32;
        f32x4_t S0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, S1 = S0;
#pragma unroll
        for (int c = 0; c < D / 32; ++c) {
            S0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[0][c], qf[c], S0, 0, 0, 0);
            S1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[1][c], qf[c], S1, 0, 0, 0);
        }
        // lane (head h, group g4) now holds scores of keys base + g4*8 + e, e = 0..7
        float sc[8];
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = (e < 4 ? S0[e] : S1[e - 4]) * p.scale;
            v = (base + g4 * 8 + e < kv_len) ? v : -INFINITY;
            sc[e] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float m_new = fmaxf(m_run, mx);
        float alpha = __expf(m_run - m_new);            // m_run = -inf -> 0
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
        for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, g4 * 4 + r, 64);     // alpha of head (g4*4 + r)
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) {
            f32x4_t o = O[dt];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] *= ar[r];
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vb[dt], o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, vb[dt], o, 0, 0, 0);
            O[dt] = o;
        }
    };
    if constexpr (XS) {
        // (cross-attention: no new key to patch.)  Straight-line code per tile count (ATT_XS_MIN_J .. ATT_XS_MAX_J; the launcher checks
        // the cache length): group 0 sits in A (/ B), the pair behind it goes to C/D, the one behind that to A/B again.  A loop over
        // pairs was tried first: across its back-edge hipcc rotates the buffers with v_mov copies of registers whose loads are
        // still in flight (each behind its own wait) and waits for vmcnt(0) at the header - no two pairs in flight together.
        // A pair is REQUESTED first and the group before it multiplied behind a binding statement that names its registers: the
        // compiler's wait for them lands there, behind the requests (vmcnt(16): "everything but the youngest pair").
        static_assert(!XS || D == 64, "XS: two pairs of tiles in registers (D = 64)");
#define XS_BIND(KA, VA, KB, VB)                                                                                                   \
        asm volatile("" : "+v"(KA[0][0]), "+v"(KA[0][1]), "+v"(KA[1][0]), "+v"(KA[1][1]), "+v"(VA[0]), "+v"(VA[1]), "+v"(VA[2]), "+v"(VA[3]), \
                          "+v"(KB[0][0]), "+v"(KB[0][1]), "+v"(KB[1][0]), "+v"(KB[1][1]), "+v"(VB[0]), "+v"(VB[1]), "+v"(VB[2]), "+v"(VB[3]) :: "memory")
        auto xs_run = [&](auto tag) {
            constexpr int NJ = decltype(tag)::value, W = ATT_WAVES;
            constexpr int G0 = (NJ & 1) ? 1 : 2, P = (NJ - G0) / 2;     // tiles in group 0, full pairs behind it (0 .. 2)
            bf16x8_t kC[2][D / 32], kD[2][D / 32], vC[D / 16], vD[D / 16];
            const int t1 = wu + G0 * W, t2 = t1 + 2 * W;
            if constexpr (P >= 1) {
                load_tile(t1, kC, vC);
                load_tile(t1 + W, kD, vD);
                XS_BIND(kA, vA, kB, vB);
            }
            process(wu, kA, vA);
            if constexpr (G0 == 2) process(wu + W, kB, vB);
            if constexpr (P >= 1) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (P >= 2) {
                    load_tile(t2, kA, vA);
                    load_tile(t2 + W, kB, vB);
                }
                XS_BIND(kC, vC, kD, vD);
                process(t1, kC, vC);
                process(t1 + W, kD, vD);
            }
            if constexpr (P >= 2) {
                __builtin_amdgcn_sched_barrier(0);
                XS_BIND(kA, vA, kB, vB);
                process(t2, kA, vA);
                process(t2 + W, kB, vB);
            }
        };
        // (the launcher admits 40 .. 48 tiles: every wave has five or six - the only counts compiled, both run by every Whisper test)
        if (xs_nj == 6) xs_run(std::integral_constant<int, 6>{});
        else xs_run(std::integral_constant<int, 5>{});
#undef XS_BIND
    } else {
        for (int tile = wave; tile < n_tiles; tile += 2 * ATT_WAVES) {
            const int tile2 = tile + ATT_WAVES;
            const bool has2 = tile2 < n_tiles;
            if (tile != wave) {                                             // later pairs (the first one was requested up front)
                load_tile(tile, kA, vA);
                if (has2) load_tile(tile2, kB, vB);
            }
            if (tile == new_tile) patch_new_key(kA, vA);
            process(tile, kA, vA);
            if (has2) {
                if (tile2 == new_tile) patch_new_key(kB, vB);
                process(tile2, kB, vB);
            }
        }
    }
    ATT_STAMP(5);
    // ---- per-wave partials -> LDS
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (g4 == 0) { sm[wave * 16 + h] = m_run; sl[wave * 16 + h] = l_run; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int head = g4 * 4 + r;
        if (head < G) {
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) sO[((size_t)wave * G + head) * D + dt * 16 + h] = O[dt][r];
        }
    }
    __syncthreads();
    ATT_STAMP(6);
    // ---- combine waves
    for (int idx = tid; idx < G * D; idx += 512) {
        int head = idx / D, d = idx - head * D;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) M = fmaxf(M, sm[w * 16 + head]);
        float num = 0.0f, den = 0.0f;
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) {
            float f = __expf(sm[w * 16 + head] - M);
            num += f * sO[((size_t)w * G + head) * D + d];
            den += f * sl[w * 16 + head];
        }
        const bf16_t ov = f32_to_bf16(num / den);
        if (p.out_ld) p.out[(size_t)b * p.out_ld + (kvh * G + head) * D + d] = ov;            // row-major (batched prefill)
        else p.out[xpk_index(b, (kvh * G + head) * D + d, p.Mpad >> 4)] = ov;                 // packed o_proj operand
    }
    ATT_STAMP(7);
}

// ---------------------------------------------------------------------------- decode attention, second schedule
//
// Same arithmetic and the same cache layout as k_attn_decode; what changes is WHEN things happen in a wave.  The phase stamps of the
// kernel above (profiles/r01_v7_attn_phases.txt, context 431) show the stream and the math in series: a wave needs ~5 us to get its
// 34 tile loads ACCEPTED (VMEM issue blocks while the CU's queue is full, i.e. for most of the 48 MB stream of the launch), only then
// joins the two prologue barriers (~6 us: slab sum, RoPE, staging through LDS), and only then multiplies - ~7 us of tile math,
// partials and combine with HBM idle.  Here
//   * the prologue is WAVE-LOCAL: every wave sums the q slabs itself (7.5 KB from L2), redistributes them to the MFMA B layout through
//     a private 2.5 KB LDS strip (no block barrier: LDS operations of one wave execute in order), and applies RoPE in registers - the
//     rotation partner d +- 64 of a B-fragment element lives in the same lane.  The wave that owns the new key's tile does the same
//     for k and v and patches them into its fragment registers; the cache append is issued last;
//   * a wave requests ONE tile (16 loads), does its prologue while that tile is in flight, requests the next tile, and multiplies the
//     first while the second streams: issue never blocks for long, and tile math overlaps the stream instead of following it;
//   * the loads are asm statements with COUNTED waits (s_waitcnt vmcnt(16) = "everything but the youngest tile"): with compiler loads
//     the wave-uniform guards around a tile request make every wait at the join conservative (vmcnt(0)), which serialises exactly
//     the overlap wanted here.  Between the first asm load and the last counted wait the kernel issues no other vector memory
//     operation (stores count in vmcnt on gfx9): prologue loads are asm too, the cache append and the output come after the loop.
// Restrictions (launcher falls back to k_attn_decode otherwise): head_dim 128, self-attention with RoPE tables in float32 ops, no
// q/k-norm, at most 4 qkv slabs, at most 32 key tiles (context <= 1024), GQA group <= 4.
#define ATT2_MAX_J 4
__device__ __forceinline__ void att2_issue_tile(const bf16_t* kt, const bf16_t* vt, unsigned voff, bf16x8_t (&ka)[2][4], bf16x8_t (&vb)[8]) {
    // kt / vt: the tile's first K / V fragment (wave-uniform: SGPR base), voff = lane * 16; fragments are 1 KiB apart and the
    // immediate offset reaches 3 KiB, hence a second base 4 KiB on.  (s_nop 4: an SGPR written by a VALU instruction - readfirstlane -
    // needs five wait states before a VMEM instruction reads it, and hipcc pads nothing inside an asm statement.)
    const bf16_t* kt1 = kt + 4 * 512;
    const bf16_t* vt1 = vt + 4 * 512;
    // non-temporal (round 4): a decode step reads every K/V tile exactly once - with the default policy the 48 MB stream of a launch
    // displaces the x fragments and slabs the neighbouring launches share through L2.  Step 2.0809 -> 2.0580 ms over three alternating
    // runs each, 13.0 -> 12.5 us in isolation (profiles/r04/c23_ab.json).
#define ATT2_NT " nt"
#define ATT2_LD(DST, BASE, OFF) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF ATT2_NT : "=v"(DST) : "v"(voff), "s"(BASE))
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" ATT2_NT : "=v"(ka[0][0]) : "v"(voff), "s"(kt));
    ATT2_LD(ka[0][1], kt, 1024); ATT2_LD(ka[0][2], kt, 2048); ATT2_LD(ka[0][3], kt, 3072);
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" ATT2_NT : "=v"(ka[1][0]) : "v"(voff), "s"(kt1));
    ATT2_LD(ka[1][1], kt1, 1024); ATT2_LD(ka[1][2], kt1, 2048); ATT2_LD(ka[1][3], kt1, 3072);
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" ATT2_NT : "=v"(vb[0]) : "v"(voff), "s"(vt));
    ATT2_LD(vb[1], vt, 1024); ATT2_LD(vb[2], vt, 2048); ATT2_LD(vb[3], vt, 3072);
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" ATT2_NT : "=v"(vb[4]) : "v"(voff), "s"(vt1));
    ATT2_LD(vb[5], vt1, 1024); ATT2_LD(vb[6], vt1, 2048); ATT2_LD(vb[7], vt1, 3072);
#undef ATT2_LD
}
// A counted wait is two statements: the s_waitcnt itself (register-free, so it can sit in either arm of a wave-uniform branch
// without giving the register allocator 64 phi values to reconcile) and ONE binding statement behind the join that names every
// register of the tile as read-write - no consumer is scheduled above it, and volatile asm statements keep their order.
#define ATT2_VMCNT(CNT) asm volatile("s_waitcnt vmcnt(" #CNT ")" ::: "memory")
#define ATT2_BIND_TILE(KA, VB)                                                                                               \
    asm volatile("" : "+v"(KA[0][0]), "+v"(KA[0][1]), "+v"(KA[0][2]), "+v"(KA[0][3]), "+v"(KA[1][0]), "+v"(KA[1][1]), "+v"(KA[1][2]),  \
                      "+v"(KA[1][3]), "+v"(VB[0]), "+v"(VB[1]), "+v"(VB[2]), "+v"(VB[3]), "+v"(VB[4]), "+v"(VB[5]), "+v"(VB[6]), "+v"(VB[7]))

// (Measured and removed in round 5: a wave with two or more tiles requesting its first TWO tiles before the prologue - 13.04-13.06 against
// 13.05-13.10 us in isolation, step 2.0982 against 2.0975 ms over alternating runs, profiles/r04/c9_ab.json, c12_ab.json: nothing.)
template <int NS>
__global__ void __launch_bounds__(512) k_attn_decode2(AttnParams p) {
    constexpr int D = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int kvh = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = p.H / p.Hkv;
    // LDS: per-wave strip of 7 x D floats (rows 0 .. G+1 <= 5: raw bf16-rounded q | k | v, row 6: the roped new key as bf16), then the
    // combine buffers of k_attn_decode
    float* strip = reinterpret_cast<float*>(smem) + (size_t)wave * 7 * D;
    float* sm = reinterpret_cast<float*>(smem) + (size_t)ATT_WAVES * 7 * D;        // [W][16]
    float* sl = sm + ATT_WAVES * 16;                                               // [W][16]
    float* sO = sl + ATT_WAVES * 16;                                               // [W][G][D]

    const uint32_t active_word = reinterpret_cast<const uint32_t*>(p.active)[b >> 2];
    const unsigned char row_active = (unsigned char)((active_word >> (8 * (b & 3))) & 0xffu);
    const int pos = p.pos[b];
    if (!row_active) return;
    const int kv_len = pos + 1;
    const int n_tiles = (kv_len + 31) >> 5;
    const int new_tile = pos >> 5;                                                  // = n_tiles - 1: the LAST tile of the wave that owns it
    const bool owner = (new_tile & (ATT_WAVES - 1)) == wave;                        // wave-uniform
    const int nj = wave < n_tiles ? (n_tiles - wave + ATT_WAVES - 1) / ATT_WAVES : 0;   // tiles wave, wave + 8, ... (<= ATT2_MAX_J)

    bf16_t* kc = p.kcache + ((size_t)(b * p.Hkv + kvh) * p.Smax) * D;
    bf16_t* vt = p.vtcache + ((size_t)(b * p.Hkv + kvh) * D) * p.Smax;
    const unsigned voff = (unsigned)lane * 16u;
    auto ktile = [&](int tile) { return kc + (size_t)tile * 32 * D; };              // 32 keys x D bf16 = 8 KiB per tile, K and V alike
    auto vtile = [&](int tile) { return vt + (size_t)tile * 32 * D; };
    const int h = lane & 15, g4 = lane >> 4;
    const int pr = pos & 31;
    const int prow = ((pr >> 3) << 2) | (pr & 3), phalf = (pr >> 2) & 1;            // the new key's place in its tile (cache tiling, header)
    bf16_t* kro = reinterpret_cast<bf16_t*>(strip + 6 * D);

    f32x4_t O[D / 16];
    float m_run, l_run;

    // Everything from the first tile request to the last tile's wait is ONE straight-line region per tile count (switch below): a
    // register an asm load is still writing must not cross a control-flow join - at a join hipcc is free to move it with v_mov, which
    // copies the stale value while the data lands in the old register (seen in the first version of this kernel; the audit is
    // tools/isa_audit_attn2.py).  Hence no `if` in here: lane predicates are selects, LDS writes are unconditional (in-bounds by
    // construction, redundant lanes write identical values), and the new key is patched at the wave's last tile, when nothing is in
    // flight any more.
    auto run = [&](auto tag) {
        constexpr int NJ = decltype(tag)::value;
        bf16x8_t kA[2][4], vA[8], kB[2][4], vB[8];
        // ---- prologue requests (asm, in this order): the slabs of q | k | v in linear order, three 16-byte units per lane and slab
        // (unit u = lane + 64 i holds elements 4u .. 4u+3 of the (G + 2) x D strip; units past it re-read unit 0), then the RoPE table
        // entries this lane's B-fragment elements need: i = c 32 + (lane >> 4) 8 + e, c = 0, 1
        const int n_unit = (G + 2) * D / 4;
        f32x4_t sv[NS][3];
    #pragma unroll
        for (int i = 0; i < 3; ++i) {
            int u = lane + 64 * i;
            u = u < n_unit ? u : 0;
            const int e0 = u * 4, hh = e0 / D, d = e0 - hh * D;
            const int col = hh < G ? (kvh * G + hh) * D + d : (hh == G ? p.H * D + kvh * D + d : p.H * D + p.Hkv * D + kvh * D + d);
    #pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float* sp = p.qkv_part + ((size_t)(s < p.S ? s : p.S - 1) * p.Mpad + b) * p.Nqkv + col;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sv[s][i]) : "v"(sp));
            }
        }
        f32x4_t rc[2][2], rs[2][2];
        {
            const float* ct = p.rope_cos + (size_t)pos * (D / 2) + g4 * 8;
            const float* st = p.rope_sin + (size_t)pos * (D / 2) + g4 * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rc[0][0]) : "v"(ct));
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(rc[0][1]) : "v"(ct));
            asm volatile("global_load_dwordx4 %0, %1, off offset:128" : "=v"(rc[1][0]) : "v"(ct));
            asm volatile("global_load_dwordx4 %0, %1, off offset:144" : "=v"(rc[1][1]) : "v"(ct));
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rs[0][0]) : "v"(st));
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(rs[0][1]) : "v"(st));
            asm volatile("global_load_dwordx4 %0, %1, off offset:128" : "=v"(rs[1][0]) : "v"(st));
            asm volatile("global_load_dwordx4 %0, %1, off offset:144" : "=v"(rs[1][1]) : "v"(st));
        }

        if constexpr (NJ >= 1) att2_issue_tile(ktile(wave), vtile(wave), voff, kA, vA);
        if constexpr (NJ >= 1) ATT2_VMCNT(16); else ATT2_VMCNT(0);
        // (one binding statement; every register exactly once - a variable named twice is copied ahead of the statement, i.e. ahead
        // of the wait)
        asm volatile("" : "+v"(rc[0][0]), "+v"(rc[0][1]), "+v"(rc[1][0]), "+v"(rc[1][1]), "+v"(rs[0][0]), "+v"(rs[0][1]), "+v"(rs[1][0]), "+v"(rs[1][1]),
                          "+v"(sv[0][0]), "+v"(sv[0][1]), "+v"(sv[0][2]));
        if constexpr (NS > 1) asm volatile("" : "+v"(sv[NS > 1 ? 1 : 0][0]), "+v"(sv[NS > 1 ? 1 : 0][1]), "+v"(sv[NS > 1 ? 1 : 0][2]));
        if constexpr (NS > 2) asm volatile("" : "+v"(sv[NS > 2 ? 2 : 0][0]), "+v"(sv[NS > 2 ? 2 : 0][1]), "+v"(sv[NS > 2 ? 2 : 0][2]));
        if constexpr (NS > 3) asm volatile("" : "+v"(sv[NS > 3 ? 3 : 0][0]), "+v"(sv[NS > 3 ? 3 : 0][1]), "+v"(sv[NS > 3 ? 3 : 0][2]));
        // slab sum (slab order 0, 1, ...: deterministic), T(), into the wave's strip in linear order (all 192 units are stored: rows
        // past G + 1 hold don't-care copies of unit 0 - a lane predicate here would be a branch)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int u = lane + 64 * i;
            f32x4_t a = sv[0][i];
#pragma unroll
            for (int s = 1; s < NS; ++s) {
                const f32x4_t t = a + sv[s][i];
                a = s < p.S ? t : a;
            }
            *reinterpret_cast<f32x4_t*>(strip + 4 * u) = (f32x4_t){bf16_round_f32(a[0]), bf16_round_f32(a[1]), bf16_round_f32(a[2]), bf16_round_f32(a[3])};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // B-fragment view + RoPE in registers: lane (h, g4) holds row h, elements c 32 + g4 8 + e; pair (c, c + 2) is (i, i + D/2)
        auto rope_row = [&](int row, bf16x8_t (&out)[4]) {
            f32x4_t x[4][2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                x[c][0] = *reinterpret_cast<const f32x4_t*>(strip + row * D + c * 32 + g4 * 8);
                x[c][1] = *reinterpret_cast<const f32x4_t*>(strip + row * D + c * 32 + g4 * 8 + 4);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x1 = x[c][e >> 2][e & 3], x2 = x[c + 2][e >> 2][e & 3];
                    const float cs = rc[c][e >> 2][e & 3], sn = rs[c][e >> 2][e & 3];
                    out[c][e] = (short)f32_to_bf16(x1 * cs - x2 * sn);
                    out[c + 2][e] = (short)f32_to_bf16(x1 * sn + x2 * cs);
                }
        };
        bf16x8_t qf[4];
        rope_row(h < G ? h : G - 1, qf);
        {   // (opaque pins: otherwise hipcc sinks the whole RoPE computation into an `if (h < G)` branch - a join in this region)
            asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));
            const bf16x8_t zero = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 4; ++c) qf[c] = h < G ? qf[c] : zero;
        }
        {   // the roped new key, parked as bf16 in row 6 (every wave does it: a branch here would be a join; 16 lanes write each value)
            bf16x8_t kn[4];
            rope_row(G, kn);
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<bf16x8_t*>(kro + c * 32 + g4 * 8) = kn[c];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // (accumulators start here, not ahead of the prologue: 32 registers less while the slabs, the tables and tile A are in flight)
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) O[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        m_run = -INFINITY; l_run = 0.0f;
        auto process = [&](int tile, const bf16x8_t (&ka)[2][4], const bf16x8_t (&vb)[8]) {      // identical math to k_attn_decode
            const int base = tile * 32;
            f32x4_t S0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, S1 = S0;
#pragma unroll
            for (int c = 0; c < D / 32; ++c) {
                S0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[0][c], qf[c], S0, 0, 0, 0);
                S1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[1][c], qf[c], S1, 0, 0, 0);
            }
            float sc[8];
            float mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = (e < 4 ? S0[e] : S1[e - 4]) * p.scale;
                v = (base + g4 * 8 + e < kv_len) ? v : -INFINITY;
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
                f32x4_t o = O[dt];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] *= ar[r];
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vb[dt], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, vb[dt], o, 0, 0, 0);
                O[dt] = o;
            }
        };
        // the wave's LAST tile: nothing is in flight behind it.  The owner of the new key patches it into the fragments (K row `prow`
        // of half `phalf`, V^T column pr) and appends it to the cache (fire and forget: nothing reads it back in this launch)
        auto last = [&](int tile, bf16x8_t (&ka)[2][4], bf16x8_t (&vb)[8]) {
            if (owner) {
                if ((lane & 15) == prow) {
                    bf16x8_t* dst = reinterpret_cast<bf16x8_t*>(kc) + ((size_t)new_tile * 2 + phalf) * (D / 32) * 64 + lane;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bf16x8_t kn = *reinterpret_cast<const bf16x8_t*>(kro + c * 32 + g4 * 8);
                        if (phalf) ka[1][c] = kn; else ka[0][c] = kn;
                        dst[c * 64] = kn;
                    }
                }
                if ((lane >> 4) == (pr >> 3)) {
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt) {
                        const bf16_t vn = f32_to_bf16(strip[(G + 1) * D + dt * 16 + (lane & 15)]);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (e == (pr & 7)) vb[dt][e] = (short)vn;
                        vt[(((size_t)new_tile * (D / 16) + dt) * 64 + lane) * 8 + (pr & 7)] = vn;
                    }
                }
            }
            process(tile, ka, vb);
        };
        // buffers A, B, A, B: while tile j is multiplied tile j + 1 is in flight, tile j + 2 is requested once its buffer is free.
        // (sched barriers: left alone the scheduler starts requesting tile j + 2 while the P.V MFMAs of tile j still read the buffer -
        // the asm outputs then need 64 NEW registers next to the old buffer and the tile in flight, and the kernel spills.)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NJ >= 2) att2_issue_tile(ktile(wave + ATT_WAVES), vtile(wave + ATT_WAVES), voff, kB, vB);
        if constexpr (NJ >= 1) {
            if constexpr (NJ >= 2) ATT2_VMCNT(16); else ATT2_VMCNT(0);
            ATT2_BIND_TILE(kA, vA);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NJ == 1) last(wave, kA, vA); else process(wave, kA, vA);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NJ >= 2) {
            if constexpr (NJ >= 3) att2_issue_tile(ktile(wave + 2 * ATT_WAVES), vtile(wave + 2 * ATT_WAVES), voff, kA, vA);
            if constexpr (NJ >= 3) ATT2_VMCNT(16); else ATT2_VMCNT(0);
            ATT2_BIND_TILE(kB, vB);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NJ == 2) last(wave + ATT_WAVES, kB, vB); else process(wave + ATT_WAVES, kB, vB);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NJ >= 3) {
            if constexpr (NJ >= 4) att2_issue_tile(ktile(wave + 3 * ATT_WAVES), vtile(wave + 3 * ATT_WAVES), voff, kB, vB);
            if constexpr (NJ >= 4) ATT2_VMCNT(16); else ATT2_VMCNT(0);
            ATT2_BIND_TILE(kA, vA);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NJ == 3) last(wave + 2 * ATT_WAVES, kA, vA); else process(wave + 2 * ATT_WAVES, kA, vA);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NJ >= 4) {
            ATT2_VMCNT(0);
            ATT2_BIND_TILE(kB, vB);
            __builtin_amdgcn_sched_barrier(0);
            last(wave + 3 * ATT_WAVES, kB, vB);
        }
    };
    switch (nj) {
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        case 4: run(std::integral_constant<int, 4>{}); break;
        default: run(std::integral_constant<int, 0>{}); break;      // a wave without a tile (context < 256): waits for its prologue loads
    }
    // ---- per-wave partials -> LDS, combine (k_attn_decode)
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (g4 == 0) { sm[wave * 16 + h] = m_run; sl[wave * 16 + h] = l_run; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int head = g4 * 4 + r;
        if (head < G) {
#pragma unroll
            for (int dt = 0; dt < D / 16; ++dt) sO[((size_t)wave * G + head) * D + dt * 16 + h] = O[dt][r];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < G * D; idx += 512) {
        int head = idx / D, d = idx - head * D;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) M = fmaxf(M, sm[w * 16 + head]);
        float num = 0.0f, den = 0.0f;
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) {
            float f = __expf(sm[w * 16 + head] - M);
            num += f * sO[((size_t)w * G + head) * D + d];
            den += f * sl[w * 16 + head];
        }
        const bf16_t ov = f32_to_bf16(num / den);
        if (p.out_ld) p.out[(size_t)b * p.out_ld + (kvh * G + head) * D + d] = ov;
        else p.out[xpk_index(b, (kvh * G + head) * D + d, p.Mpad >> 4)] = ov;
    }
}
static size_t attn2_smem_bytes(int G) { return ((size_t)ATT_WAVES * 7 * 128 + 2 * ATT_WAVES * 16 + (size_t)ATT_WAVES * G * 128) * 4; }

#ifdef MIS_ATTN_TIMING
#include <string.h>
static unsigned long long* g_attn_dbg = nullptr;
static int g_attn_slot = 0;
extern "C" int mis_debug_attn_timing_init() {          // outside any stream capture
    if (!g_attn_dbg) {
        if (hipHostMalloc((void**)&g_attn_dbg, 4096 * 16 * sizeof(unsigned long long), 0) != hipSuccess) return 1;
        memset(g_attn_dbg, 0, 4096 * 16 * 8);
    }
    g_attn_slot = 0;
    return 0;
}
extern "C" int mis_debug_attn_timing(unsigned long long* out, int max_slots) {      // copies [slots][16] stamps, returns slots used
    int n = g_attn_slot < max_slots ? g_attn_slot : max_slots;
    if (g_attn_dbg) memcpy(out, g_attn_dbg, (size_t)n * 16 * 8);
    return n;
}
#endif
static int attn_nit(int G, int D) {                  // prologue elements per thread: NIT * 512 >= (G + 2) * D
    const int n_el = (G + 2) * D;
    return n_el <= 1024 ? 2 : (D == 128 ? 5 : 3);
}
size_t attn_smem_bytes(int G, int D) {
    return (size_t)attn_nit(G, D) * 512 * 4 + 16 * D * 2 + D * 2 + 2 * ATT_WAVES * 16 * 4 + (size_t)ATT_WAVES * G * D * 4;
}

// the cross-attention launches whose prologue can take the LayerNorm glue and the query projection (k_attn_decode<64, 2, true, QP>)
bool attn_qp_ok(const AttnParams& p) {
    const char* xe = getenv("MIS_ATTN_XS");
    return !(xe && atoi(xe) == 0) && p.cross && p.D == 64 && p.H == p.Hkv && !p.append_only && !p.cache_rows && !p.rope_cos && !p.qnorm_w &&
           (p.cross_len + 31) / 32 >= ATT_WAVES * ATT_XS_MIN_J && (p.cross_len + 31) / 32 <= ATT_WAVES * ATT_XS_MAX_J && p.qp_KT >= 1 &&
           (p.qp_KT + ATT_WAVES - 1) / ATT_WAVES <= ATT_QP_KW && p.qp_KT * 32 == p.H * p.D && p.qp_S >= 1 && p.qp_S <= 8;
}
void launch_attn_decode(const AttnParams& p, int batch, hipStream_t s) {
    int G = p.H / p.Hkv;
    MIS_REQUIRE(G >= 1 && G <= 16 && p.H % p.Hkv == 0, MIS_ERR_INVALID_INPUT, "GQA group size must be 1..16");
    size_t smem = attn_smem_bytes(G, p.D);
    MIS_REQUIRE(smem <= 64 * 1024, MIS_ERR_INVALID_INPUT, "attention LDS footprint too large");
    MIS_REQUIRE(p.qp_w || (p.S >= 1 && p.S <= 8), MIS_ERR_GENERATION_FAILED, "attention prologue reduces at most 8 split-K slabs (got %d)", p.S);
    MIS_REQUIRE(((uintptr_t)p.active & 3) == 0, MIS_ERR_GENERATION_FAILED, "attention: the active-flag array must be 4-byte aligned");
    dim3 grid(p.Hkv, batch), block(512);
    const int n_el = (G + 2) * p.D;
    // second schedule (k_attn_decode2) where it applies: head_dim 128, RoPE tables, no q/k norm, <= 4 slabs, a decode step (the batched
    // prefill keeps the first schedule: its two arrangements must give the same bits whatever the batch size)
    if (!p.first_schedule && !p.cache_rows && !p.append_only && p.D == 128 && !p.cross && p.rope_cos && !p.rope_in_dtype && !p.qnorm_w && p.S <= 4 && G <= 4 && p.Smax <= 32 * ATT_WAVES * ATT2_MAX_J &&
        p.Smax % 32 == 0 && ((uintptr_t)p.qkv_part & 15) == 0 && p.Nqkv % 4 == 0) {
        const size_t sm2 = attn2_smem_bytes(G);
        // k_attn_decode2 is compiled for 0 .. ATT2_MAX_J key tiles per wave (its switch has no case beyond): the cache must not hold more
        MIS_REQUIRE((p.Smax / 32 + ATT_WAVES - 1) / ATT_WAVES <= ATT2_MAX_J, MIS_ERR_GENERATION_FAILED,
                    "attention: %d cache positions need more than %d key tiles per wave", p.Smax, ATT2_MAX_J);
        switch (p.S) {
            case 1: hipLaunchKernelGGL((k_attn_decode2<1>), grid, block, sm2, s, p); break;
            case 2: hipLaunchKernelGGL((k_attn_decode2<2>), grid, block, sm2, s, p); break;
            case 3: hipLaunchKernelGGL((k_attn_decode2<3>), grid, block, sm2, s, p); break;
            default: hipLaunchKernelGGL((k_attn_decode2<4>), grid, block, sm2, s, p); break;
        }
        return;
    }
#ifdef MIS_ATTN_TIMING
    // one 16-stamp slot per enqueued launch (graph replays rewrite their slot); dumped by mis_debug_attn_timing()
    AttnParams pt = p;
    pt.dbg = g_attn_dbg ? g_attn_dbg + (size_t)(g_attn_slot++ % 4096) * 16 : nullptr;   // mis_debug_attn_timing_init() first
    const AttnParams& p2 = pt;
#else
    const AttnParams& p2 = p;
#endif
    const char* xe = getenv("MIS_ATTN_XS");                                                // 0: the pair-at-a-time loop for cross-attention too (A/B,
    const bool xs_on = !(xe && atoi(xe) == 0);                                             // parity tests: read per launch)
    if (p.D == 128 && n_el <= 1024) hipLaunchKernelGGL((k_attn_decode<128, 2>), grid, block, smem, s, p2);
    else if (p.D == 128) hipLaunchKernelGGL((k_attn_decode<128, 5>), grid, block, smem, s, p2);
    else if (p.D == 64 && n_el <= 1024 && p.cross && !p.append_only && !p.cache_rows && xs_on && (p.cross_len + 31) / 32 >= ATT_WAVES * ATT_XS_MIN_J &&
             (p.cross_len + 31) / 32 <= ATT_WAVES * ATT_XS_MAX_J)
    {
        if (p.qp_w) {                                                                       // LayerNorm + query projection in the prologue
            MIS_REQUIRE(attn_qp_ok(p), MIS_ERR_GENERATION_FAILED, "attention: the query-projection prologue does not apply to this launch");
            AttnParams pq = p2;                            // (the kernel's unconditional dummy loads - norm weights, RoPE rows - read qkv_part[0])
            if (!pq.qkv_part) pq.qkv_part = pq.qp_slabs;
            const AttnParams& p2 = pq;
            const size_t smq = smem + ATT_QP_LDS;
            if (p.qp_S <= 4) hipLaunchKernelGGL((k_attn_decode<64, 2, true, 4>), grid, block, smq, s, p2);
            else hipLaunchKernelGGL((k_attn_decode<64, 2, true, 8>), grid, block, smq, s, p2);
        } else
        hipLaunchKernelGGL((k_attn_decode<64, 2, true>), grid, block, smem, s, p2);       // cross-attention: two pairs of tiles in flight
    }
    else if (p.D == 64 && n_el <= 1024) hipLaunchKernelGGL((k_attn_decode<64, 2>), grid, block, smem, s, p2);
    else if (p.D == 64) hipLaunchKernelGGL((k_attn_decode<64, 3>), grid, block, smem, s, p2);
    else throw MisError(MIS_ERR_INVALID_INPUT, "head_dim must be 64 or 128");
}
