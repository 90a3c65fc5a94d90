// codec_bf3.hip - split-bf16 dense contractions for the codec decoders (SNAC / DAC / Qwen3-TTS speech tokenizer / Vocos / EnCodec).
//
// The codec weights and activations are f32 (SNAC ships f32 checkpoints; the waveform gate is 1e-4 RMS), and gfx950 has no TF32-like
// MFMA: exact-f32 MFMA runs at 1/16 of the bf16 rate.  Here every f32 operand is split into two bf16 halves, x = xh + xl with
// xh = bf16(x), xl = bf16(x - xh) (relative error of the pair <= 2^-18), and a product becomes three bf16 MFMAs accumulated in f32:
// xh.wh + xh.wl + xl.wh (the lo.lo term, <= 2^-16 of the product, is dropped).  Measured on the CPU emulation of exactly this
// arithmetic inside the SNAC oracle (tools/bf16x3_emulation.py, profiles/r02_bf16x3_emulation.json): 1.06e-5 RMS on the 24 kHz decode
// against the 1e-4 gate (plain bf16: 5.7e-3).  The exact-f32 kernels of snac.hip stay: encoders (codebook decisions), small or
// memory-bound shapes, and MIS_CODEC_EXACT_F32=1.
//
// Three kernels:
//   k_bf3_pack_w   A^T f32 [taps][Cin][M] -> MFMA-A fragments of v_mfma_f32_32x32x16_bf16, hi and lo, [tap][M/32][Cp/16][hl][64][8]
//                  (once per weight matrix, cached per model in CodecPack)
//   k_bf3_split    activation pre-pass: (Snake) -> split -> transpose to time-major planes xh, xl [B][Tp][Cp] with zero pads, so that a
//                  B fragment (8 consecutive channels of one time column) is one aligned 16-byte piece; Snake runs ONCE per element
//                  (the staged f32 kernels re-run it per 64-row block and per transposed-conv phase)
//   k_bf3_gemm     128 x 128 output tile per block: 4 MFMA waves (64 x 64 each) + 1 loader wave.  The loader streams the activation
//                  tile (with its tap halo) by LDS-DMA into an NBUF-deep ring, chunk = 32 channels, running NBUF-1 chunks ahead on its
//                  OWN vmcnt counter; the MFMA waves take the packed weight fragments straight from L2 into registers (prefetched two
//                  steps ahead) and the activation fragments from LDS (XOR-swizzled at the DMA source, conflict-free ds_read_b128).
//                  One s_barrier per chunk.  Taps (dense k-tap convs), transposed-conv phases and 1x1 convs share the tile code:
//                  tap j reads the staged tile at a column offset.
#include "common.h"
#include "codec_kernels.h"

#define B3_BM 128
#define B3_BN 128
#define B3_KC 32
#define B3_THREADS 320

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct Bf3Params {
    GemmParams g;
    const uint16_t* wp;
    const uint16_t* xh;
    const uint16_t* xl;
    int mode, ntaps, MT32, KS, Cp, Tp, t_org;
    int ksplit;           // > 1: blockIdx.z = batch row x slice; the block contracts chunks [nchunks slice / ksplit, nchunks (slice + 1) / ksplit)
    float* part;          //      and writes its raw float32 sums to part[slice][b][M][ldy] (k_bf3_splitk_epilogue finishes)
    int batch;
};

__device__ __forceinline__ float bf3_snake(float x, float a, float ra) { return fmaf(ra, mis_sin_sq(a * x), x); }

// ---- weights: A^T [ntaps][Cin][M] f32 -> fragments --------------------------------------------------------------------------------
__global__ void k_bf3_pack_w(const float* __restrict__ AT, uint16_t* __restrict__ wp, int ntaps, int Cin, int M, int MT32, int KS) {
    const size_t total = (size_t)ntaps * MT32 * KS * 512;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
        size_t r = e >> 9;
        const int ks = (int)(r % KS); r /= KS;
        const int mt = (int)(r % MT32);
        const int tap = (int)(r / MT32);
        const int m = mt * 32 + (lane & 31), c = ks * 16 + (lane >> 5) * 8 + j;
        float v = (m < M && c < Cin) ? AT[((size_t)tap * Cin + c) * M + m] : 0.0f;
        const uint16_t h = f32_to_bf16(v);
        const uint16_t l = f32_to_bf16(v - bf16_to_f32(h));
        const size_t o = ((((size_t)tap * MT32 + mt) * KS + ks) * 2) * 512 + (size_t)lane * 8 + j;
        wp[o] = h;
        wp[o + 512] = l;
    }
}

// ---- activations: x f32 [B][C][ldx] -> planes [B][Tp][Cp] (column t' holds input column t' + t_org; zero outside [x_lo, Tin)) ---------
// Block: 64 columns x SP_CB channels.  Loads are coalesced along time (a wave reads 64 consecutive columns of one channel), the split
// halves go through an LDS transpose, stores are coalesced along channels (16-byte pieces, a full SP_CB-channel row per 16 threads)
#define SP_CB 128
#define SP_LD (SP_CB + 8)             // LDS row stride in bf16 (272 B: 16-byte aligned rows; the 2-byte transposing writes of a wave hit
                                      // rows 68 dwords apart: 4 | 68 only by 4, so the 64 lanes spread over 16 banks x 2 ways at worst)
__global__ void __launch_bounds__(256) k_bf3_split(const float* __restrict__ X, uint16_t* __restrict__ xh, uint16_t* __restrict__ xl,
                                                   const float* __restrict__ alpha, const float* __restrict__ ralpha,
                                                   int C, int ldx, int x_lo, int Tin, int Cp, int Tp, int t_org) {
    __shared__ __attribute__((aligned(16))) uint16_t sh[64 * SP_LD], sl[64 * SP_LD];
    const int tl = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int tp0 = blockIdx.x * 64, c0 = blockIdx.y * SP_CB, b = blockIdx.z;
    const int cw = min(SP_CB, Cp - c0);                         // channels of this block (multiple of 32)
    const int t = tp0 + tl + t_org;
    const bool tin = tp0 + tl < Tp && t >= x_lo && t < Tin;
    for (int cc = cg; cc < cw; cc += 4) {
        const int c = c0 + cc;
        float x = 0.0f;
        if (tin && c < C) {
            x = X[((int64_t)b * C + c) * ldx + t];
            if (alpha) x = bf3_snake(x, alpha[c], ralpha[c]);
        }
        const uint16_t h = f32_to_bf16(x);
        sh[tl * SP_LD + cc] = h;
        sl[tl * SP_LD + cc] = f32_to_bf16(x - bf16_to_f32(h));
    }
    __syncthreads();
    const int pieces = cw / 8;                                  // 16-byte pieces per row
    for (int i = threadIdx.x; i < 64 * pieces; i += 256) {
        const int r = i / pieces, pc = i - r * pieces;
        if (tp0 + r >= Tp) continue;
        const size_t o = ((size_t)b * Tp + tp0 + r) * Cp + c0 + pc * 8;
        *reinterpret_cast<uint4*>(xh + o) = *reinterpret_cast<const uint4*>(sh + r * SP_LD + pc * 8);
        *reinterpret_cast<uint4*>(xl + o) = *reinterpret_cast<const uint4*>(sl + r * SP_LD + pc * 8);
    }
}

// ---- the contraction -----------------------------------------------------------------------------------------------------------------
struct Bf3A { bf16x8_t v[2][2]; };     // [m tile][hi, lo]
struct Bf3B { bf16x8_t v[2][2]; };     // [n tile][hi, lo]

// sched_group_barrier masks: 0x008 MFMA, 0x002 VALU, 0x020 VMEM read, 0x100 DS read
#define B3_IL_V __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#define B3_IL_D __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#define B3_IL_M __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
#define B3_INTERLEAVE B3_IL_V B3_IL_V B3_IL_V B3_IL_V B3_IL_D B3_IL_D B3_IL_D B3_IL_D B3_IL_M B3_IL_M B3_IL_M B3_IL_M

// The body takes the arrays the epilogue touches as restrict-qualified PARAMETERS: GemmParams is a struct, clang ignores restrict on
// struct members, and without it every bias / residual load of the epilogue waits behind the previous row's store (the same effect
// cost the fused unit kernel of snac.hip 25 %).
template <int NQ, int NBUF, int NTAPS>
__device__ __forceinline__ void bf3_body(const Bf3Params& P, const float* __restrict__ pX, const float* __restrict__ pR, float* __restrict__ pY,
                                         const float* __restrict__ pbias, const float* __restrict__ pscale, const float* __restrict__ pnoise) {
    constexpr int PLANE = NQ * 512;                       // bf16 per plane: NQ DMA instructions of 1 KiB (16 columns x 32 channels)
    constexpr int BUF = 2 * PLANE;
    constexpr int NST = 2 * NTAPS;                        // MFMA steps (16 channels of one tap) per chunk
    __shared__ __attribute__((aligned(1024))) uint16_t lds[NBUF * BUF];
    const GemmParams& p = P.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.z, phase = 0, sh0 = 0, dsh = 0, tapbase = 0, slice = 0;
    if (P.mode == GEMM_CONVT) { b = blockIdx.z / p.s; phase = blockIdx.z - b * p.s; sh0 = (phase + p.pad) / p.s; dsh = -1; tapbase = phase * NTAPS; }
    else if (P.mode == GEMM_TAPS) { sh0 = -p.pad; dsh = p.dil; }
    else if (P.ksplit > 1) { b = blockIdx.z / P.ksplit; slice = blockIdx.z - b * P.ksplit; }
    const int shmin = min(sh0, sh0 + (NTAPS - 1) * dsh);
    const int nchunks_all = P.Cp / B3_KC;
    const int c_lo = P.ksplit > 1 ? nchunks_all * slice / P.ksplit : 0;                        // this block's chunks (split-K: a slice of them)
    const int nchunks = (P.ksplit > 1 ? nchunks_all * (slice + 1) / P.ksplit : nchunks_all) - c_lo;
    const int n0 = blockIdx.x * B3_BN;

    if (wave == 4) {
        // ---- loader: tile column i = input column n0 + shmin + i.  DMA instruction q moves columns 16q .. 16q+15, four lanes per column;
        // lane slot gs holds channel group gs ^ ((i >> 2) & 3) (the swizzle is on the SOURCE address, the LDS side of a DMA is lane-linear)
        const int il = lane >> 2, grp = (lane & 3) ^ ((il >> 2) & 3);
        const size_t col0 = ((size_t)b * P.Tp + (size_t)(n0 + shmin - P.t_org + il)) * P.Cp + grp * 8;
        const uint16_t* sh = P.xh + col0 + (size_t)c_lo * B3_KC;
        const uint16_t* sl = P.xl + col0 + (size_t)c_lo * B3_KC;
        const size_t qstride = (size_t)16 * P.Cp;
        auto issue = [&](int cc) {
            const uint16_t* h = sh + cc * B3_KC;
            const uint16_t* l = sl + cc * B3_KC;
            const int buf = cc % NBUF;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                __builtin_amdgcn_global_load_lds((gptr_t)(h + q * qstride), (lptr_t)&lds[buf * BUF + q * 512], 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(l + q * qstride), (lptr_t)&lds[buf * BUF + PLANE + q * 512], 16, 0, 0);
            }
        };
        for (int c = 0; c < NBUF - 1 && c < nchunks; ++c) issue(c);
        for (int cc = 0; cc < nchunks; ++cc) {
            // chunk cc must have landed; up to NBUF-2 younger chunks stay in flight
            if (nchunks - 1 - cc >= NBUF - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * 2 * NQ) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (cc + NBUF - 1 < nchunks) issue(cc + NBUF - 1);
        }
        return;
    }

    // ---- MFMA waves: 2 x 2, wave tile 64 rows x 64 columns = 2 x 2 tiles of 32 x 32
    const int wm = wave >> 1, wn = wave & 1;
    const int mt0 = blockIdx.y * 4 + wm * 2;
    if (mt0 >= P.MT32) {                                   // no rows for this wave (M <= 64 in this block row): keep the barrier count
        for (int cc = 0; cc < nchunks; ++cc) __builtin_amdgcn_s_barrier();
        return;
    }
    f32x16_t acc[2][2];

    // weight stream: step (chunk, tap, ks) -> fragments of both row tiles, hi and lo.  It runs two steps ahead of the MFMAs, across
    // chunk boundaries (it does not depend on the staged tile).  A second row tile past M re-reads the first (its rows are
    // never stored).  The body below is three chunks (3 * NST steps) of straight-line code: the weight registers rotate with period 3,
    // the tile-fragment registers with period 2, every index is a compile-time constant and there is no branch between the MFMAs,
    // so the waitcnt pass keeps exactly the two younger weight loads in flight (with branches in the body it fell back to vmcnt(0)).
    const size_t tap_stride = (size_t)P.MT32 * P.KS * 1024, mt_stride = (size_t)P.KS * 1024;
    const uint16_t* wbase = P.wp + (size_t)lane * 8 + (size_t)tapbase * tap_stride + (size_t)mt0 * mt_stride + (size_t)c_lo * 2 * 1024;
    const size_t mt1 = (mt0 + 1 < P.MT32) ? mt_stride : 0;
    const int ncol = wn * 64 + (lane & 31), kgrp = lane >> 5;
    Bf3A aq[3];
    Bf3B bq[2];
    auto loadA = [&](Bf3A& f, int chunk, int tap, int ks) {
        chunk = min(chunk, nchunks - 1);                     // the last steps prefetch past the end: re-read the last chunk
        const uint16_t* q = wbase + (size_t)tap * tap_stride + (size_t)(chunk * 2 + ks) * 1024;
        f.v[0][0] = *reinterpret_cast<const bf16x8_t*>(q);
        f.v[0][1] = *reinterpret_cast<const bf16x8_t*>(q + 512);
        f.v[1][0] = *reinterpret_cast<const bf16x8_t*>(q + mt1);
        f.v[1][1] = *reinterpret_cast<const bf16x8_t*>(q + mt1 + 512);
    };
    auto readB = [&](Bf3B& f, const uint16_t* tile, int tap, int ks) {
        const int toff = sh0 + tap * dsh - shmin;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int i = ncol + ni * 32 + toff;
            const uint16_t* q = tile + i * 32 + (((ks * 2 + kgrp) ^ ((i >> 2) & 3)) * 8);
            f.v[ni][0] = *reinterpret_cast<const bf16x8_t*>(q);
            f.v[ni][1] = *reinterpret_cast<const bf16x8_t*>(q + PLANE);
        }
    };
    // three products per tile pair: hi.hi, hi.lo, lo.hi; four independent accumulators between dependent MFMAs
    auto mfma12 = [&](const Bf3A& A, const Bf3B& B) {
#pragma unroll
        for (int term = 0; term < 3; ++term) {
            const int ta = term == 2 ? 1 : 0, tb = term == 1 ? 1 : 0;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v[mi][ta], B.v[ni][tb], acc[mi][ni], 0, 0, 0);
        }
    };
    loadA(aq[0], 0, 0, 0);
    loadA(aq[1], 0, 0, 1);
    const uint16_t* tile = lds;
    int gbuf = 0;                                           // ring slot of the next chunk
    const float* Xb = pX + (size_t)b * p.Cin * p.ldx;
    const bool convt = P.mode == GEMM_CONVT, gelu = P.mode == GEMM_GELU, noise = P.mode == GEMM_NOISE;
    const float* Rr = (convt || gelu || noise) ? nullptr : pR;

    {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        int cc = 0;
        for (; cc + 3 <= nchunks; cc += 3) {
#pragma unroll
            for (int I = 0; I < 3 * NST; ++I) {
                const int t = I % NST;
                if (t == 0) {
                    __builtin_amdgcn_s_barrier();
                    tile = lds + gbuf * BUF;
                    gbuf = gbuf + 1 == NBUF ? 0 : gbuf + 1;
                    readB(bq[I & 1], tile, 0, 0);
                }
                // one scheduling region per step: the prefetches of later steps cannot sink out of it (left alone the scheduler moves
                // them next to their use), and inside it they are spread over the shadows of the 12 MFMAs instead of running ahead
                // of them with the matrix pipe idle: MFMA, then a weight load or a tile read with its address arithmetic
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < NST) readB(bq[(I + 1) & 1], tile, (t + 1) >> 1, (t + 1) & 1);
                const int J = I + 2;
                loadA(aq[J % 3], cc + J / NST, (J % NST) >> 1, J & 1);
                mfma12(aq[I % 3], bq[I & 1]);
                B3_INTERLEAVE
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // tail: one or two chunks, same schedule with a runtime bound (rotation phase 0 here: 3 * NST steps per group)
        if (cc < nchunks) {
            const int rem = (nchunks - cc) * NST;
#pragma unroll
            for (int I = 0; I < 2 * NST; ++I) {
                if (I < rem) {
                    const int t = I % NST;
                    if (t == 0) {
                        __builtin_amdgcn_s_barrier();
                        tile = lds + gbuf * BUF;
                        gbuf = gbuf + 1 == NBUF ? 0 : gbuf + 1;
                        readB(bq[I & 1], tile, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 1 < NST) readB(bq[(I + 1) & 1], tile, (t + 1) >> 1, (t + 1) & 1);
                    const int J = I + 2;
                    loadA(aq[J % 3], cc + J / NST, (J % NST) >> 1, J & 1);
                    mfma12(aq[I % 3], bq[I & 1]);
                    B3_INTERLEAVE
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        // ---- epilogue (the modes of k_snac_gemm / k_conv_taps).  C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        // opaque: left visible, the 32 rows' 64-bit offsets are computed ahead of the main loop and spilled (860 B of scratch per
        // lane in the multi-tile experiment: 1.9x slower)
        int lane_hi = lane >> 5;
        asm volatile("" : "+v"(lane_hi));
        if (P.ksplit > 1) {                                  // split-K: the raw sums of this slice; k_bf3_splitk_epilogue adds the slices up
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int n = n0 + wn * 64 + ni * 32 + (lane & 31);
                if (n >= p.N) continue;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = (mt0 + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lane_hi;
                        if (m < p.M) P.part[(((size_t)slice * P.batch + b) * p.M + m) * p.ldy + n] = acc[mi][ni][r];
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + wn * 64 + ni * 32 + (lane & 31);
            if (n >= p.N) continue;
            float nz = 0.0f;
            if (noise) {
                if (pnoise) nz = pnoise[(size_t)b * p.N + n];
                else if (p.noise_rng) {
                    uint64_t row = (uint64_t)(p.row_offset + (p.row_ids ? p.row_ids[b] : b));
                    uint64_t u = mis_splitmix64((p.noise_key ^ (row * 0xD1B54A32D192ED03ull)) + (uint64_t)n);
                    float u1 = ((float)(uint32_t)(u >> 40) + 0.5f) * 5.9604644775390625e-08f;
                    float u2 = ((float)(uint32_t)((u >> 16) & 0xFFFFFF) + 0.5f) * 5.9604644775390625e-08f;
                    nz = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
                }
            }
            const int o = convt ? p.s * n + phase : n;
            if (o >= p.Tout) continue;
            const bool dupb = convt && n == 0 && p.dup_bias_n0;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (mt0 + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lane_hi;
                    if (m >= p.M) continue;
                    float v = acc[mi][ni][r];
                    const float bm = pbias ? pbias[m] : 0.0f;
                    v += bm;
                    if (dupb) v += bm;
                    const size_t rowo = ((size_t)b * p.M + m) * p.ldy;
                    if (gelu) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                    if (Rr) { if (pscale) v *= pscale[m]; v += Rr[rowo + n]; }
                    if (noise) v = Xb[(size_t)m * p.ldx + n] + nz * v;
                    pY[rowo + o] = v;
                }
            }
        }
    }
}

template <int NQ, int NBUF, int NTAPS, int MINW>
__global__ void __launch_bounds__(B3_THREADS, MINW) k_bf3_gemm(Bf3Params P) {
    bf3_body<NQ, NBUF, NTAPS>(P, P.g.X, P.g.R, P.g.Y, P.g.bias, P.g.scale, P.g.noise);
}

// split-K: Y = epilogue(sum over slices in slice order + bias) - the 1x1 modes' epilogues of bf3_body (plain / GELU / residual with scale)
__global__ void k_bf3_splitk_epilogue(const float* __restrict__ part, int S, int batch, int mode, const float* __restrict__ bias, const float* __restrict__ scale,
                                      const float* __restrict__ R, float* __restrict__ Y, int M, int N, int ldy) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y, b = blockIdx.z;
    if (n >= N) return;
    const size_t rowo = ((size_t)b * M + m) * ldy;
    float v = 0.0f;
    for (int s = 0; s < S; ++s) v += part[(((size_t)s * batch + b) * M + m) * ldy + n];
    v += bias ? bias[m] : 0.0f;
    if (mode == GEMM_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    if (mode == GEMM_RESID && R) { if (scale) v *= scale[m]; v += R[rowo + n]; }
    Y[rowo + n] = v;
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------
CodecPack::~CodecPack() {
    for (auto& e : entries) if (e.wp) (void)hipFree(e.wp);
}

const CodecPack::Entry& CodecPack::get(const float* at, int ntaps, int Cin, int M, hipStream_t s) {
    for (auto& e : entries)
        if (e.at == at && e.ntaps == ntaps && e.Cin == Cin && e.M == M) return e;
    Entry e{};
    e.at = at; e.ntaps = ntaps; e.Cin = Cin; e.M = M;
    e.Cp = (Cin + B3_KC - 1) / B3_KC * B3_KC;
    e.KS = e.Cp / 16;
    e.MT32 = (M + 31) / 32;
    const size_t n = (size_t)ntaps * e.MT32 * e.KS * 1024;
    HIP_CHECK(hipMalloc((void**)&e.wp, n * sizeof(uint16_t)));
    hipLaunchKernelGGL(k_bf3_pack_w, dim3((unsigned)std::min<size_t>((n / 2 + 255) / 256, 8192)), dim3(256), 0, s, at, e.wp, ntaps, Cin, M, e.MT32, e.KS);
    entries.push_back(e);
    return entries.back();
}

static thread_local CodecPack* tl_pack = nullptr;
CodecPackScope::CodecPackScope(CodecPack* p) : prev(tl_pack) { tl_pack = p; }
CodecPackScope::~CodecPackScope() { tl_pack = prev; }

static int bf3_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

bool launch_gemm_bf3(int mode, bool snake, const GemmParams& p_in, int batch, hipStream_t s) {
    GemmParams p = p_in;
    if (!p.pack) p.pack = tl_pack;
    if (!p.pack || bf3_env("MIS_CODEC_EXACT_F32", 0)) return false;
    const bool one = mode == GEMM_PLAIN || mode == GEMM_GELU || mode == GEMM_RESID || mode == GEMM_NOISE;
    int ntaps = 1, Cin = p.K, span = 0, t_org = 0, wtaps = 1;
    if (mode == GEMM_TAPS) { ntaps = p.taps; Cin = p.Cin; span = (p.taps - 1) * p.dil; t_org = -p.pad; wtaps = ntaps; }
    else if (mode == GEMM_CONVT) {
        if (p.Cin <= 0 || p.K % p.Cin || p.K / p.Cin > 2) return false;
        ntaps = p.K / p.Cin;                                         // 2: kernel 2s (overlap-add); 1: kernel = stride (A^T holds one tap)
        Cin = p.Cin; span = ntaps - 1; wtaps = ntaps * p.s;
        t_org = p.pad / p.s - (ntaps - 1);                           // phase 0 has the smallest q = (phase + pad) / s
        if ((p.s - 1 + p.pad) / p.s - p.pad / p.s > 1) return false;
    } else if (!one) return false;
    // what pays: MFMA-bound shapes.  A 1x1 conv over few channels is HBM-bound and the pre-pass would only add traffic
    // (measured per dispatch, profiles/r02_codec/: 1x1 over <= 512 channels and the last SNAC transposed conv, M*K = 16 K,
    // are faster on the exact-f32 kernels; every 7-tap conv and the transposed convs from M*K = 36 K up gain 2.3 - 4x)
    const int min_k1 = bf3_env("MIS_BF3_MIN_K1", 768), min_k = bf3_env("MIS_BF3_MIN_K", 32), min_mk = bf3_env("MIS_BF3_MIN_MK_CONVT", 32768);
    if (one ? Cin < min_k1 : Cin < min_k) return false;
    if (mode == GEMM_CONVT && (int64_t)p.M * p.K < min_mk) return false;
    if (p.M < 32 || span > 64 || (ntaps != 1 && ntaps != 2 && ntaps != 7)) return false;
    const CodecPack::Entry& e = p.pack->get(p.AT, wtaps, Cin, p.M, s);

    const int nbx = (p.N + B3_BN - 1) / B3_BN;
    const int nq = span <= 16 ? 9 : 12;
    // the transposed conv's phases share one pre-pass: columns from the smallest shift of any phase to the largest
    const int extra = mode == GEMM_CONVT ? ((p.s - 1 + p.pad) / p.s - p.pad / p.s) : 0;
    const int Tp = (nbx - 1) * B3_BN + nq * 16 + extra;
    p.pack->xh.alloc((size_t)batch * Tp * e.Cp);
    p.pack->xl.alloc((size_t)batch * Tp * e.Cp);
    const bool sn = snake && p.alpha;
    hipLaunchKernelGGL(k_bf3_split, dim3((Tp + 63) / 64, (e.Cp + SP_CB - 1) / SP_CB, batch), dim3(256), 0, s, p.X, p.pack->xh.p, p.pack->xl.p,
                       sn ? p.alpha : nullptr, sn ? p.ralpha : nullptr, Cin, p.ldx, p.x_lo, p.Tin, e.Cp, Tp, t_org);
    Bf3Params P{};
    P.g = p; P.g.Cin = Cin;
    P.wp = e.wp; P.xh = p.pack->xh.p; P.xl = p.pack->xl.p;
    P.mode = mode; P.ntaps = ntaps; P.MT32 = e.MT32; P.KS = e.KS; P.Cp = e.Cp; P.Tp = Tp; P.t_org = t_org;
    const int phases = mode == GEMM_CONVT ? p.s : 1;
    // launch-shaped 1x1 contractions (Soprano's ConvNeXt GEMMs at 257 frames: 18 or 54 blocks per row of a 256-CU part, 60 us per launch):
    // the K range split over blocks.  The factor is a function of the SHAPE of one batch row only (a row's bits do not depend on the batch
    // it shares); float32 slabs, summed in slice order by k_bf3_splitk_epilogue
    const int nchunks = e.Cp / B3_KC, per_row = nbx * ((p.M + B3_BM - 1) / B3_BM);
    int ksplit = 1;
    if (p.split_k_ok && ntaps == 1 && (mode == GEMM_PLAIN || mode == GEMM_GELU || mode == GEMM_RESID) && !bf3_env("MIS_BF3_NO_SPLITK", 0)) {
        if (per_row <= 32 && nchunks >= 32) ksplit = 4;
        else if (per_row <= 64 && nchunks >= 16) ksplit = 2;
    }
    if (ksplit > 1) {
        p.pack->part.alloc((size_t)ksplit * batch * p.M * p.ldy);
        P.ksplit = ksplit; P.part = p.pack->part.p; P.batch = batch;
    }
    dim3 grid(nbx, (p.M + B3_BM - 1) / B3_BM, batch * phases * ksplit), block(B3_THREADS);
    // MINW 3 = two blocks (ten waves) per CU: 168 registers; the 7-tap body then spills a few address temporaries (A/B by MIS_BF3_MINW)
    // MINW 3 = two blocks (ten waves) per CU, 168 registers: fits 1 and 2 taps; the 7-tap body would spill (measured 2.3x slower), so it
    // runs one block per CU (a persistent loop over column tiles was measured too: no gain, profiles/r02_codec/ab_record.json)
    if (ntaps == 1) {
        hipLaunchKernelGGL((k_bf3_gemm<9, 4, 1, 3>), grid, block, 0, s, P);
        if (ksplit > 1)
            hipLaunchKernelGGL(k_bf3_splitk_epilogue, dim3((p.N + 255) / 256, p.M, batch), dim3(256), 0, s, P.part, ksplit, batch, mode, p.bias, p.scale, p.R, p.Y, p.M,
                               p.N, p.ldy);
    }
    else if (ntaps == 2) hipLaunchKernelGGL((k_bf3_gemm<9, 4, 2, 3>), grid, block, 0, s, P);
    else if (nq == 9) hipLaunchKernelGGL((k_bf3_gemm<9, 4, 7, 2>), grid, block, 0, s, P);
    else hipLaunchKernelGGL((k_bf3_gemm<12, 4, 7, 2>), grid, block, 0, s, P);
    return true;
}
