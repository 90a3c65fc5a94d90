// lm_qgemm.hip - weight-streaming skinny GEMM on MLX affine-quantised weights (8 / 4 bit, group size 64), dequantised in registers.
//
// Reference being replaced: QuantizedLinear -> quantizedMatmul(x, w, scales, biases, transpose: true, groupSize, bits)
// (created for every path with `.scales`, Sources/MLXAudioTTS/Models/Llama/LlamaTTS.swift:958-968,
// Sources/MLXAudioTTS/Models/Qwen3TTS/Qwen3TTS.swift:1157-1170; arithmetic in mlx [3P]: per group of `group_size` inputs
// y += scale * sum_k(x_k * q_k) + bias * sum_k(x_k), float32 accumulation, output in x's dtype).
//
// Here: the integer codes q in [0, 2^bits) are exact in bf16, so sum_k x_k q_k runs on v_mfma_f32_16x16x32_bf16 with exact
// products and float32 accumulation, one accumulator per 64-wide scale group (= 2 k-tiles); sum_k x_k comes from one more MFMA
// against an all-ones A tile; the group's scale and bias (bf16 in the checkpoint, exact in float32) are applied to those two
// float32 sums.  Nothing is rounded that the reference does not round - the dequantise-at-load path rounds s*q+b to bf16 per
// weight.  HBM bytes per launch: N*K*bits/8 codes + N*(K/64)*4 scale/bias bytes (8 bit: 0.53x, 4 bit: 0.28x of bf16).
//
// Layouts (written by k_pack_qweight at load):
//   codes   [NT][KT][64 lanes][8 codes]: lane l = q*16 + i holds W[16nt + i][32kt + 8q .. +8) - the bf16 pack of lm_kernels.hip with
//           codes instead of values: 8 bytes (uint2) per lane at 8 bit, 4 bytes at 4 bit (code e in bits [bits*e, bits*(e+1)))
//   scales  [NT][G][2][16] bf16: scale then bias of rows 16nt .. 16nt+15 for scale group g (a lane reads its 4 C/D rows as 8 bytes)
// Work decomposition, epilogues and the in-block split-K combine are those of k_gemm_skinny; K ranges are cut at scale groups.
#include "common.h"
#include "lm_kernels.h"


typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <int BITS> struct QTile;
template <> struct QTile<8> { typedef u32x2_t type; };
template <> struct QTile<4> { typedef unsigned int type; };

// 8 codes -> 8 bf16 values (exact).  (float)(byte) is v_cvt_f32_ubyteN; the pair conversion is v_cvt_pk_bf16_f32.
// (Measured alternative: v_perm_b32 placing byte k under the exponent of 2^23, one v_pk_add_f32 per pair, then the same pack - 16
// instead of 12 instructions per fragment and no faster anywhere: profiles/r03/qgemm_loads_vs_math.jsonl.)
__device__ __forceinline__ bf16x8_t dq_codes(u32x2_t w) {
    bf16x8_t r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = (short)f32_to_bf16((float)((w.x >> (8 * e)) & 0xffu));
        r[4 + e] = (short)f32_to_bf16((float)((w.y >> (8 * e)) & 0xffu));
    }
    return r;
}
__device__ __forceinline__ bf16x8_t dq_codes(unsigned int w) {
    bf16x8_t r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (short)f32_to_bf16((float)((w >> (4 * e)) & 0xfu));
    return r;
}
__device__ __forceinline__ void unpack_bf16x4(uint2 v, float (&o)[4]) {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}

// epilogues: identical arithmetic to gemm_epilogue of lm_kernels.hip (kept in step with it)
template <int MT, int R, int EPI>
__device__ __forceinline__ void qgemm_epilogue(const f32x4_t (&acc)[R][MT], void* __restrict__ out, int ntg, int ks, int NT, int N_out,
                                               int Mpad, int lane, int mt_only, const bf16_t* __restrict__ bias) {
    const int nl = (lane >> 4) * 4, ml = lane & 15;
    if (EPI == EPI_PARTIAL) {
        float* o = reinterpret_cast<float*>(out);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int tile = ntg * R + r;
            if (tile >= NT) continue;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias && ks == 0) {
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
    } else if (EPI == EPI_BF16) {
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
                for (int e = 0; e < 4; ++e) res[e] = f32_to_bf16(acc[r][mt][e] + bv[e]);
                size_t off = ((size_t)mt * 16 + ml) * N_out + tile * 16 + nl;
                uint2 v;
                v.x = (uint32_t)res[0] | ((uint32_t)res[1] << 16);
                v.y = (uint32_t)res[2] | ((uint32_t)res[3] << 16);
                *reinterpret_cast<uint2*>(o + off) = v;
            }
        }
    } else {   // EPI_SILU_MUL: tile 2t = gate rows, 2t+1 = up rows  (LlamaTTS.swift:283); R / 2 feature tiles per item
        bf16_t* o = reinterpret_cast<bf16_t*>(out);
#pragma unroll
        for (int pr = 0; pr < R / 2; ++pr) {
            if (R > 2 && (ntg * R + 2 * pr + 1) >= NT) continue;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt_only >= 0 && mt != mt_only) continue;
                uint16_t res[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float g = bf16_round_f32(acc[2 * pr][mt][e]);
                    float u = bf16_round_f32(acc[2 * pr + 1][mt][e]);
                    float sg = bf16_round_f32(1.0f / (1.0f + __expf(-g)));
                    float a = bf16_round_f32(g * sg);
                    res[e] = f32_to_bf16(a * u);
                }
                size_t off = xpk_index(mt * 16 + ml, (ntg * (R / 2) + pr) * 16 + nl, MT);
                uint2 v;
                v.x = (uint32_t)res[0] | ((uint32_t)res[1] << 16);
                v.y = (uint32_t)res[2] | ((uint32_t)res[3] << 16);
                *reinterpret_cast<uint2*>(o + off) = v;
            }
        }
    }
}

// QGEMM_U = scale groups per register buffer: 2 (= 4 k-tiles, 196 VGPRs at MT = 2, two blocks per CU) or 1 (four blocks per CU; the
// only one that fits without spills at MT >= 3)
// Round 4: R = 4 n-tiles per wave with KSB = 4 or 8 waves per item (8: 512-thread blocks) for the wide roles.  What the round-3
// diagnostics (loads-only build: 17.7 of 19.0 us) did not say is WHICH loads: the bf16 laboratory of round 4 shows launch time following
// x-fragment bytes + weight bytes alike (every wave re-reads its x fragments out of L2; R = 2 -> 4 took the bf16 gate+up from 19.4 to
// 17.6 us) - and at 8 bit the x fragments are TWICE the code bytes at R = 2.  Four tiles per wave halve them; one scale group per
// buffer keeps the wave under 256 registers.
template <int MT, int R, int EPI, int KSB, int BITS, int QGEMM_U>
__global__ void __launch_bounds__(KSB == 8 ? 512 : 256, 2) k_gemm_skinny_q(const void* __restrict__ Qp, const bf16_t* __restrict__ SB, const bf16_t* __restrict__ X,
                                                       void* __restrict__ out, int NT, int G, int S, int n_items, int N_out, int Mpad,
                                                       const bf16_t* __restrict__ bias) {
    static_assert(EPI != EPI_SILU_MUL || (R & 1) == 0, "silu-mul epilogue pairs a gate tile with an up tile");
    typedef typename QTile<BITS>::type WT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = (KSB == 1) ? blockIdx.x * 4 + wave : blockIdx.x;
    if (item >= n_items) return;
    const int ntg = item / S, ks = item - ntg * S;
    const int KT = 2 * G;
    int g0 = (int)(((long long)G * ks) / S), g1 = (int)(((long long)G * (ks + 1)) / S);          // this item's scale groups
    if (KSB > 1) {
        int len = g1 - g0;
        int a = g0 + (int)(((long long)len * wave) / KSB), b = g0 + (int)(((long long)len * (wave + 1)) / KSB);
        g0 = a; g1 = b;
    }
    const WT* wp[R];
    const uint2* sp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int tile = ntg * R + r;
        if (tile >= NT) tile = NT - 1;                     // clamp (store is skipped in the epilogue)
        wp[r] = reinterpret_cast<const WT*>(Qp) + (size_t)tile * KT * 64 + lane;
        sp[r] = reinterpret_cast<const uint2*>(SB) + (size_t)tile * G * 8 + (lane >> 4);       // group g: + g*8 (scale), + g*8 + 4 (bias)
    }
    const bf16x8_t* xp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xp[mt] = reinterpret_cast<const bf16x8_t*>(X) + mt * 64 + lane;

    f32x4_t acc[R][MT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[r][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;               // bf16 1.0

#if defined(MIS_QGEMM_ABLATE) && MIS_QGEMM_ABLATE == 2
    WT wA[QGEMM_U][2][R] = {}, wB[QGEMM_U][2][R] = {};
    bf16x8_t xA[QGEMM_U][2][MT] = {}, xB[QGEMM_U][2][MT] = {};
    uint2 sA[QGEMM_U][R][2] = {}, sB[QGEMM_U][R][2] = {};
#else
    WT wA[QGEMM_U][2][R], wB[QGEMM_U][2][R];
    bf16x8_t xA[QGEMM_U][2][MT], xB[QGEMM_U][2][MT];
    uint2 sA[QGEMM_U][R][2], sB[QGEMM_U][R][2];
#endif
    const int glast = g1 - 1;
#if defined(MIS_QGEMM_ABLATE) && MIS_QGEMM_ABLATE == 2   /* diagnostics build (make ablate): no loads in the K loop, registers made opaque */
#define QG_LOAD(WBUF, XBUF, SBUF, GBASE)                                                            \
    _Pragma("unroll") for (int u = 0; u < QGEMM_U; ++u) {                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                             \
            _Pragma("unroll") for (int r = 0; r < R; ++r) asm volatile("" : "+v"(WBUF[u][j][r]));   \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(XBUF[u][j][mt])); \
        }                                                                                           \
        _Pragma("unroll") for (int r = 0; r < R; ++r) { asm volatile("" : "+v"(SBUF[u][r][0])); asm volatile("" : "+v"(SBUF[u][r][1])); } \
    }
#else
#define QG_LOAD(WBUF, XBUF, SBUF, GBASE)                                                            \
    _Pragma("unroll") for (int u = 0; u < QGEMM_U; ++u) {                                           \
        int gg = (GBASE) + u;                                                                       \
        gg = gg > glast ? glast : gg;               /* tail: redundant reload, math is skipped */   \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                             \
            _Pragma("unroll") for (int r = 0; r < R; ++r)                                           \
                WBUF[u][j][r] = __builtin_nontemporal_load(wp[r] + (size_t)(2 * gg + j) * 64);      \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                       \
                XBUF[u][j][mt] = xp[mt][(size_t)(2 * gg + j) * (MT * 64)];                          \
        }                                                                                           \
        _Pragma("unroll") for (int r = 0; r < R; ++r) {                                             \
            SBUF[u][r][0] = sp[r][(size_t)gg * 8];                                                  \
            SBUF[u][r][1] = sp[r][(size_t)gg * 8 + 4];                                              \
        }                                                                                           \
    }
#endif
#if defined(MIS_QGEMM_ABLATE) && MIS_QGEMM_ABLATE == 1   /* diagnostics build: loads only - every loaded register is consumed by one xor */
#define QG_GROUP(WBUF, XBUF, SBUF, U)                                                               \
    {                                                                                               \
        uint32_t h = 0;                                                                             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                             \
            _Pragma("unroll") for (int r = 0; r < R; ++r) {                                         \
                const WT t = WBUF[U][j][r];                                                         \
                const uint32_t* tp = reinterpret_cast<const uint32_t*>(&t);                         \
                _Pragma("unroll") for (int e = 0; e < (int)(sizeof(WT) / 4); ++e) h ^= tp[e];       \
            }                                                                                       \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                     \
                const bf16x8_t t = XBUF[U][j][mt];                                                  \
                const uint32_t* tp = reinterpret_cast<const uint32_t*>(&t);                         \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) h ^= tp[e];                           \
            }                                                                                       \
        }                                                                                           \
        _Pragma("unroll") for (int r = 0; r < R; ++r) h ^= SBUF[U][r][0].x ^ SBUF[U][r][0].y ^ SBUF[U][r][1].x ^ SBUF[U][r][1].y; \
        acc[0][0][0] += __uint_as_float(h & 0x007fffffu);                                           \
    }
#else
#define QG_GROUP(WBUF, XBUF, SBUF, U)                                                               \
    {                                                                                               \
        f32x4_t ag[R][MT], sx[MT];                                                                  \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                         \
            sx[mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};                                                 \
            _Pragma("unroll") for (int r = 0; r < R; ++r) ag[r][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; \
        }                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                             \
            bf16x8_t fr[R];                                                                         \
            _Pragma("unroll") for (int r = 0; r < R; ++r) fr[r] = dq_codes(WBUF[U][j][r]);          \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                     \
                sx[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, XBUF[U][j][mt], sx[mt], 0, 0, 0); \
                _Pragma("unroll") for (int r = 0; r < R; ++r)                                       \
                    ag[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[r], XBUF[U][j][mt], ag[r][mt], 0, 0, 0); \
            }                                                                                       \
        }                                                                                           \
        _Pragma("unroll") for (int r = 0; r < R; ++r) {                                             \
            float sc[4], bi[4];                                                                     \
            unpack_bf16x4(SBUF[U][r][0], sc);                                                       \
            unpack_bf16x4(SBUF[U][r][1], bi);                                                       \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                       \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                       \
                    acc[r][mt][e] += sc[e] * ag[r][mt][e] + bi[e] * sx[mt][e];                      \
        }                                                                                           \
    }
#endif
#define QG_MATH_FULL(WBUF, XBUF, SBUF)                                                              \
    _Pragma("unroll") for (int u = 0; u < QGEMM_U; ++u) QG_GROUP(WBUF, XBUF, SBUF, u)
#define QG_MATH_TAIL(WBUF, XBUF, SBUF, GBASE)                                                       \
    _Pragma("unroll") for (int u = 0; u < QGEMM_U; ++u) {                                           \
        if ((GBASE) + u < g1) QG_GROUP(WBUF, XBUF, SBUF, u)                                         \
    }
    if (g0 < g1) {
        int g = g0;
        QG_LOAD(wA, xA, sA, g)
        while (g + 3 * QGEMM_U <= g1) {                    // steady state: unguarded loads and math (see k_gemm_skinny)
            QG_LOAD(wB, xB, sB, g + QGEMM_U)
            __builtin_amdgcn_sched_barrier(0);
            QG_MATH_FULL(wA, xA, sA)
            __builtin_amdgcn_sched_barrier(0);
            QG_LOAD(wA, xA, sA, g + 2 * QGEMM_U)
            __builtin_amdgcn_sched_barrier(0);
            QG_MATH_FULL(wB, xB, sB)
            __builtin_amdgcn_sched_barrier(0);
            g += 2 * QGEMM_U;
        }
        if (g + QGEMM_U < g1) {
            QG_LOAD(wB, xB, sB, g + QGEMM_U)
            __builtin_amdgcn_sched_barrier(0);
            QG_MATH_TAIL(wA, xA, sA, g)
            if (g + 2 * QGEMM_U < g1) {
                QG_LOAD(wA, xA, sA, g + 2 * QGEMM_U)
                __builtin_amdgcn_sched_barrier(0);
                QG_MATH_TAIL(wB, xB, sB, g + QGEMM_U)
                QG_MATH_TAIL(wA, xA, sA, g + 2 * QGEMM_U)
            } else {
                QG_MATH_TAIL(wB, xB, sB, g + QGEMM_U)
            }
        } else {
            QG_MATH_TAIL(wA, xA, sA, g)
        }
    }
#undef QG_LOAD
#undef QG_GROUP
#undef QG_MATH_FULL
#undef QG_MATH_TAIL

    if (KSB == 1) {
        qgemm_epilogue<MT, R, EPI>(acc, out, ntg, ks, NT, N_out, Mpad, lane, -1, bias);
    } else {
        __shared__ float4 red[KSB][R * MT][64];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                red[wave][r * MT + mt][lane] = make_float4(acc[r][mt][0], acc[r][mt][1], acc[r][mt][2], acc[r][mt][3]);
        __syncthreads();
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
            qgemm_epilogue<MT, R, EPI>(acc, out, ntg, ks, NT, N_out, Mpad, lane, mt, bias);
        }
    }
}

// ---- one-shot arrangement (MT <= 2; MIS_QGEMM_V2=0 turns it off): for launches where a wave's share of K is at most U scale groups - the split-K
// roles of the decode step and every Qwen3-TTS-sized matrix - all loads of the wave are issued up front and the kernel is ONE memory
// round trip instead of one per register buffer.  What pays for the registers: the MFMA operands are swapped (x fragment as the A
// operand, codes as B: D[m][n], lane l holds n = l & 15, m = 4 (l >> 4) + e), so a lane's C/D column is one output row n and the scale
// and bias of a group are one bf16 each per lane (2-byte loads from the same [NT][G][2][16] table) instead of four each; the wave
// index is made scalar, so tile and K-range arithmetic and the base addresses live in SGPRs.  Groups past the wave's range are loaded
// from its last group (clamped) and enter with scale = bias = 0: straight-line code, no load can sink behind a branch.  The
// accumulators pass through LDS once at the end (a 16 x 16 float tile per (r, mt), written [m][n], read back as the streaming
// kernel's lanes hold them), so the epilogues and the in-block split-K combine are shared; eight waves per item are possible here.
// Per-element arithmetic is that of k_gemm_skinny_q; the in-block combine adds KSB partials in wave order.
template <int MT, int R, int EPI, int KSB, int BITS, int U>
__global__ void __launch_bounds__(256, 2) k_gemm_skinny_q1(const void* Qp, const bf16_t* SB, const bf16_t* X,     // not __restrict__: see the
                                                                            void* __restrict__ out, int NT, int G, int S, int n_items, int N_out, int Mpad,   // fence below
                                                                            const bf16_t* __restrict__ bias) {
    static_assert(EPI != EPI_SILU_MUL || R == 2, "silu-mul epilogue pairs a gate tile with an up tile");
    typedef typename QTile<BITS>::type WT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    static_assert(KSB == 4, "one block of four waves per item, its waves split the item's K range (eight waves measured worse and were removed)");
    const int item = blockIdx.x;
    if (item >= n_items) return;
    const int ntg = item / S, ks = item - ntg * S;
    const int KT = 2 * G;
    int g0 = (int)(((long long)G * ks) / S), g1 = (int)(((long long)G * (ks + 1)) / S);
    {
        int len = g1 - g0;
        int a = g0 + (int)(((long long)len * wave) / KSB), b = g0 + (int)(((long long)len * (wave + 1)) / KSB);
        g0 = a; g1 = b;
    }
    f32x4_t acc[R][MT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[r][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    if (g0 < g1) {                                         // the launcher guarantees g1 - g0 <= U
        const WT* wp[R];
        const uint16_t* sp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int tile = ntg * R + r;
            if (tile >= NT) tile = NT - 1;                 // clamp (store is skipped in the epilogue)
            wp[r] = reinterpret_cast<const WT*>(Qp) + (size_t)tile * KT * 64 + lane;
            sp[r] = reinterpret_cast<const uint16_t*>(SB) + (size_t)tile * G * 32 + (lane & 15);   // group g: + 32 g (scale), + 32 g + 16 (bias)
        }
        const bf16x8_t* xp[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xp[mt] = reinterpret_cast<const bf16x8_t*>(X) + mt * 64 + lane;
        bf16x8_t ones;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;           // bf16 1.0

        WT w[U][2][R];
        bf16x8_t x[U][2][MT];
        uint32_t sb[U][R][2];
        const int glast = g1 - 1;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int gg = g0 + u;
            gg = gg > glast ? glast : gg;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < R; ++r) w[u][j][r] = __builtin_nontemporal_load(wp[r] + (size_t)(2 * gg + j) * 64);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) x[u][j][mt] = xp[mt][(size_t)(2 * gg + j) * (MT * 64)];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                sb[u][r][0] = sp[r][(size_t)gg * 32];
                sb[u][r][1] = sp[r][(size_t)gg * 32 + 16];
            }
            __builtin_amdgcn_sched_barrier(0);             // issue order = group order (the math below waits group by group)
        }
        // every load is issued before the first use.  sched_barrier alone does not hold loads back (instruction selection does not
        // order them against it, and loads through noalias read-only pointers may be moved across anything - the SLP vectoriser did):
        // the operands are plain pointers and this fence may write memory as far as the compiler knows.
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = g0 + u < g1;
            // nothing that consumes a loaded value may rise above the fence (conversions and the x-only MFMA are pure and would,
            // taking their vmcnt wait with them): every buffer register passes through an empty volatile asm here, which is also
            // where the wait for this group's loads belongs
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < R; ++r) asm volatile("" : "+v"(w[u][j][r]));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(x[u][j][mt]));
            }
#pragma unroll
            for (int r = 0; r < R; ++r) { asm volatile("" : "+v"(sb[u][r][0])); asm volatile("" : "+v"(sb[u][r][1])); }
            f32x4_t ag[R][MT], sx[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                sx[mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < R; ++r) ag[r][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf16x8_t fr[R];
#pragma unroll
                for (int r = 0; r < R; ++r) fr[r] = dq_codes(w[u][j][r]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    sx[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[u][j][mt], ones, sx[mt], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < R; ++r) ag[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[u][j][mt], fr[r], ag[r][mt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float sc = live ? __uint_as_float(sb[u][r][0] << 16) : 0.0f;
                const float bi = live ? __uint_as_float(sb[u][r][1] << 16) : 0.0f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[r][mt][e] += sc * ag[r][mt][e] + bi * sx[mt][e];
            }
            __builtin_amdgcn_sched_barrier(0);             // groups are consumed in the order their loads were issued (vmcnt is in order)
        }
    }

    // hand-over to the streaming kernel's lane layout: tile [m][n] floats in LDS; lane L then owns m = L & 15, n = 4 (L >> 4) .. + 3
    __shared__ float4 red[KSB][R * MT][64];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float* t = reinterpret_cast<float*>(red[wave][r * MT + mt]);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[(4 * (lane >> 4) + e) * 16 + (lane & 15)] = acc[r][mt][e];
        }
    const int rd = (lane & 15) * 4 + (lane >> 4);
    __syncthreads();
    for (int mt = wave; mt < MT; mt += KSB) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float4 s0 = red[0][r * MT + mt][rd];
#pragma unroll
            for (int wv = 1; wv < KSB; ++wv) {
                float4 t = red[wv][r * MT + mt][rd];
                s0.x += t.x; s0.y += t.y; s0.z += t.z; s0.w += t.w;
            }
#pragma unroll
            for (int m2 = 0; m2 < MT; ++m2)
                if (m2 == mt) acc[r][m2] = (f32x4_t){s0.x, s0.y, s0.z, s0.w};
        }
        qgemm_epilogue<MT, R, EPI>(acc, out, ntg, ks, NT, N_out, Mpad, lane, mt, bias);
    }
}

template <int MT, int BITS, int U>
static void launch_qgemm1_mt(int epi, int R, int ksb, const void* Qp, const bf16_t* SB, const bf16_t* X, void* out, int NT, int G, int S,
                             int N_out, int Mpad, const bf16_t* bias, hipStream_t s) {
    int n_items = ((NT + R - 1) / R) * S;
    dim3 grid(n_items), block(256);
#define QGEMM1_CASE(E, RR, KS)                                                                                      \
    if (epi == E && R == RR && ksb == KS) {                                                                         \
        hipLaunchKernelGGL((k_gemm_skinny_q1<MT, RR, E, KS, BITS, U>), grid, block, 0, s, Qp, SB, X, out, NT, G, S, n_items, \
                           N_out, Mpad, bias);                                                                      \
        return;                                                                                                     \
    }
    QGEMM1_CASE(EPI_PARTIAL, 1, 4)
    QGEMM1_CASE(EPI_PARTIAL, 2, 4)
    QGEMM1_CASE(EPI_BF16, 2, 4)
    QGEMM1_CASE(EPI_SILU_MUL, 2, 4)
#undef QGEMM1_CASE
    throw MisError(MIS_ERR_GENERATION_FAILED, "unsupported quantised GEMM variant");
}

template <int MT, int BITS, int U>
static void launch_qgemm_mt(int epi, int R, int ksb, const void* Qp, const bf16_t* SB, const bf16_t* X, void* out, int NT, int G, int S,
                            int N_out, int Mpad, const bf16_t* bias, hipStream_t s) {
    int n_items = ((NT + R - 1) / R) * S;
    dim3 grid(ksb == 1 ? (n_items + 3) / 4 : n_items), block(ksb == 8 ? 512 : 256);
#define QGEMM_CASE(E, RR, KS)                                                                                       \
    if (epi == E && R == RR && ksb == KS) {                                                                         \
        hipLaunchKernelGGL((k_gemm_skinny_q<MT, RR, E, KS, BITS, U>), grid, block, 0, s, Qp, SB, X, out, NT, G, S, n_items, \
                           N_out, Mpad, bias);                                                                      \
        return;                                                                                                     \
    }
    QGEMM_CASE(EPI_PARTIAL, 1, 4)
    QGEMM_CASE(EPI_PARTIAL, 2, 1)
    QGEMM_CASE(EPI_PARTIAL, 2, 4)
    QGEMM_CASE(EPI_BF16, 2, 1)
    QGEMM_CASE(EPI_BF16, 2, 4)
    QGEMM_CASE(EPI_SILU_MUL, 2, 1)
    QGEMM_CASE(EPI_SILU_MUL, 2, 4)
    if constexpr (MT <= 2 && U == 1) {             // four n-tiles per wave: one scale group per buffer, <= 32 rows
        QGEMM_CASE(EPI_SILU_MUL, 4, 4)
        QGEMM_CASE(EPI_SILU_MUL, 4, 8)
        QGEMM_CASE(EPI_BF16, 4, 4)
        QGEMM_CASE(EPI_BF16, 4, 8)
        QGEMM_CASE(EPI_PARTIAL, 4, 4)
        QGEMM_CASE(EPI_PARTIAL, 4, 8)
    }
    if constexpr (MT <= 2 && U == 2) {
        QGEMM_CASE(EPI_SILU_MUL, 2, 8)
        QGEMM_CASE(EPI_BF16, 2, 8)
    }
#undef QGEMM_CASE
    throw MisError(MIS_ERR_GENERATION_FAILED, "unsupported quantised GEMM variant");
}

// Qp / SB: packed codes and scale/bias pairs (see the file header); G = K / 64 scale groups; otherwise as launch_gemm_skinny
void launch_gemm_skinny_q(int bits, int epi, int R, int ksb, const void* Qp, const bf16_t* SB, const bf16_t* X, void* out, int NT, int G,
                          int S, int N_out, int Mpad, hipStream_t s, const bf16_t* bias) {
    MIS_REQUIRE(epi == EPI_PARTIAL || S == 1, MIS_ERR_GENERATION_FAILED, "split-K needs the partial epilogue");
    MIS_REQUIRE(bits == 8 || bits == 4, MIS_ERR_GENERATION_FAILED, "quantised GEMM: 8 or 4 bits");
    MIS_REQUIRE(S >= 1 && S <= G, MIS_ERR_GENERATION_FAILED, "quantised GEMM: %d K slices for %d scale groups", S, G);   // a wave may get none
    // one-shot arrangement (k_gemm_skinny_q1) wherever a wave's K share fits its buffer.  MIS_QGEMM_V2=0 (read once per process) sends
    // those launches through the streaming kernel instead: that kernel is the product path of the wide roles at Orpheus width, and the
    // switch is how tests/test_gpu_loader.py holds it to the oracle at small widths too.  (Eight waves per item and capped buffer sizes
    // were measured and removed: profiles/r03/qgemm_one_shot.jsonl.)
    static const int v2 = getenv("MIS_QGEMM_V2") ? atoi(getenv("MIS_QGEMM_V2")) : 1;
    if (R == 4 || ksb == 8) {                      // the wide roles' arrangement (streaming kernel only): R = 4 -> one group per buffer
        MIS_REQUIRE(Mpad / 16 <= 2, MIS_ERR_GENERATION_FAILED, "quantised GEMM: four n-tiles per wave / eight waves per item are built for <= 32 rows");
#define QGEMM_R4(M, UU)                                                                                              \
        { if (bits == 8) launch_qgemm_mt<M, 8, UU>(epi, R, ksb, Qp, SB, X, out, NT, G, S, N_out, Mpad, bias, s);         \
          else launch_qgemm_mt<M, 4, UU>(epi, R, ksb, Qp, SB, X, out, NT, G, S, N_out, Mpad, bias, s);                   \
          return; }
        if (R == 4) { if (Mpad / 16 == 1) QGEMM_R4(1, 1) else QGEMM_R4(2, 1) }
        else { if (Mpad / 16 == 1) QGEMM_R4(1, 2) else QGEMM_R4(2, 2) }
#undef QGEMM_R4
    }
    if (v2 && Mpad / 16 <= 2) {
        // where a wave's K share fits a buffer of 2, 4 or 6 scale groups (the smallest that holds it: groups past the share cost loads
        // and MFMAs); everything else streams through k_gemm_skinny_q.
        const int per_item = (G + S - 1) / S;
        const int k2 = ksb, n = (per_item + ksb - 1) / ksb;
        const int u2 = ksb == 1 ? 0 : n <= 2 ? 2 : n <= 4 ? 4 : n <= 6 ? 6 : 0;       // one wave per item: the streaming kernel (long K shares)
#define QGEMM1_GO(M, UU)                                                                                             \
        { if (bits == 8) launch_qgemm1_mt<M, 8, UU>(epi, R, k2, Qp, SB, X, out, NT, G, S, N_out, Mpad, bias, s);         \
          else launch_qgemm1_mt<M, 4, UU>(epi, R, k2, Qp, SB, X, out, NT, G, S, N_out, Mpad, bias, s);                   \
          return; }
#define QGEMM1_MT(M)                                                                                                 \
        if (u2 == 2) QGEMM1_GO(M, 2)                                                                                 \
        if (u2 == 4) QGEMM1_GO(M, 4)                                                                                 \
        if (u2 == 6) QGEMM1_GO(M, 6)
        if (u2) { if (Mpad / 16 == 1) { QGEMM1_MT(1) } else { QGEMM1_MT(2) } }
#undef QGEMM1_MT
#undef QGEMM1_GO
    }
#define QGEMM_MT(M, UU)                                                                                              \
    if (bits == 8) launch_qgemm_mt<M, 8, UU>(epi, R, ksb, Qp, SB, X, out, NT, G, S, N_out, Mpad, bias, s);           \
    else launch_qgemm_mt<M, 4, UU>(epi, R, ksb, Qp, SB, X, out, NT, G, S, N_out, Mpad, bias, s);
    switch (Mpad / 16) {
        case 1: { QGEMM_MT(1, 2) } break;
        case 2: { QGEMM_MT(2, 2) } break;
        case 3: { QGEMM_MT(3, 1) } break;
        case 4: { QGEMM_MT(4, 1) } break;
        default: throw MisError(MIS_ERR_INVALID_INPUT, "batch per GPU must be <= 64");
    }
#undef QGEMM_MT
}

// ---- load-time packing: MLX layout (wq uint32 [N][K*bits/32], scales / biases bf16 [N][K/64]) -> the layouts above.
// Tile placement as launch_pack_weight: source n-tile t lands at tile t*tile_stride + tile_offset of a matrix with KT = K/32.
template <int BITS>
__global__ void k_pack_qweight(const uint32_t* __restrict__ wq, const bf16_t* __restrict__ scales, const bf16_t* __restrict__ biases,
                               void* __restrict__ qdst, bf16_t* __restrict__ sbdst, int N, int K, int tile_stride, int tile_offset) {
    const int KT = K / 32, G = K / 64;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (n, kt, q)
    const size_t total = (size_t)N * KT * 4;
    if (idx < total) {
        const int q = (int)(idx & 3);
        const int kt = (int)((idx >> 2) % KT);
        const int n = (int)(idx / ((size_t)4 * KT));
        const int tile = (n >> 4) * tile_stride + tile_offset, lane = q * 16 + (n & 15);
        const int k0 = kt * 32 + q * 8;
        if (BITS == 8) {
            const uint32_t* src = wq + (size_t)n * (K / 4) + k0 / 4;
            reinterpret_cast<uint2*>(qdst)[((size_t)tile * KT + kt) * 64 + lane] = make_uint2(src[0], src[1]);
        } else {
            reinterpret_cast<uint32_t*>(qdst)[((size_t)tile * KT + kt) * 64 + lane] = wq[(size_t)n * (K / 8) + k0 / 8];
        }
    }
    if (idx < (size_t)N * G) {                                                  // one thread per (n, g): scale and bias
        const int g = (int)(idx % G), n = (int)(idx / G);
        const int tile = (n >> 4) * tile_stride + tile_offset;
        sbdst[(((size_t)tile * G + g) * 2 + 0) * 16 + (n & 15)] = scales[(size_t)n * G + g];
        sbdst[(((size_t)tile * G + g) * 2 + 1) * 16 + (n & 15)] = biases[(size_t)n * G + g];
    }
}
void launch_pack_qweight(int bits, const uint32_t* wq, const bf16_t* scales, const bf16_t* biases, void* qdst, bf16_t* sbdst, int N, int K,
                         int tile_stride, int tile_offset, hipStream_t s) {
    MIS_REQUIRE((bits == 8 || bits == 4) && N >= 1 && K % 64 == 0, MIS_ERR_INVALID_INPUT, "quantised pack: bad shape");
    const size_t total = (size_t)N * (K / 32) * 4;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (bits == 8) hipLaunchKernelGGL((k_pack_qweight<8>), grid, block, 0, s, wq, scales, biases, qdst, sbdst, N, K, tile_stride, tile_offset);
    else hipLaunchKernelGGL((k_pack_qweight<4>), grid, block, 0, s, wq, scales, biases, qdst, sbdst, N, K, tile_stride, tile_offset);
}
