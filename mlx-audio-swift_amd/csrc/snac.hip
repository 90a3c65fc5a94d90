// snac.hip - SNAC codec decode for gfx950 (fp32, exact-f32 MFMA for the dense contractions).
//
// Reference being replaced (Sources/MLXAudioCodecs/SNAC/): SNAC.decode SNACDecoder.swift:127-131 =
// ResidualVectorQuantize.fromCodes (VQ.swift:165-191) -> Decoder (Layers.swift:364-421).
// The reference recomputes weight-norm every call, transposes NCT<->NTC around every conv and forces
// ~18 evals per decode (Layers.swift:28,101-116); here weight-norm, the codebook x out_proj product and
// the transposed-conv phase split are folded ONCE at finalize and activations stay NCT in HBM.
//
// Kernels (activations f32 [B][C][T], T contiguous):
//   k_snac_embed   z_q = sum_i table_i[codes_i[b, t/stride_i]]       (table_i = codebook_i x out_proj_i + b_i)
//   k_snac_dw      depthwise k=7 dilated conv with optional Snake before/after, LDS halo tile
//   k_snac_gemm    Y = A x X on v_mfma_f32_32x32x2_f32 with fused Snake prologue and
//                  bias / residual / NoiseBlock / transposed-conv-phase epilogues
//   k_snac_final   Snake -> conv k7 (C -> 1) -> tanh
#include "common.h"
#include "kernels.h"
#include "codec_kernels.h"

#include <math.h>
#include <string.h>
#include <algorithm>

// ============================================================================ device kernels

__device__ __forceinline__ float snake_f(float x, float a, float ra) {
    return fmaf(ra, mis_sin_sq(a * x), x);      // Layers.swift:44-50: x + 1/(alpha+1e-9) * sin(alpha x)^2 (mis_sin_sq: codec_kernels.h)
}

// ---- fromCodes ---------------------------------------------------------------------------------
struct EmbedParams {
    const float* tables;        // [n_q][codebook_size][C]
    const int32_t* codes[8];    // [B][T0/stride_i]
    int strides[8];
    int n_q, codebook_size, C, T0;
};

// block: 32 channels x 32 frames, 256 threads; LDS transpose so that table reads are coalesced over
// c and z_q writes are coalesced over t.
__global__ void __launch_bounds__(256) k_snac_embed(EmbedParams p, float* __restrict__ zq) {
    __shared__ float tile[32][33];
    int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    int tid = threadIdx.x;
    int cc = tid & 31;
    for (int tt = tid >> 5; tt < 32; tt += 8) {
        int t = t0 + tt, c = c0 + cc;
        float acc = 0.0f;
        if (t < p.T0 && c < p.C) {
            for (int i = 0; i < p.n_q; ++i) {          // (0 + z0) + z1 + z2 : VQ.swift:186 order
                int Ti = p.T0 / p.strides[i];
                int code = p.codes[i][(size_t)b * Ti + t / p.strides[i]];
                code = code < 0 ? 0 : (code >= p.codebook_size ? p.codebook_size - 1 : code);
                acc += p.tables[((size_t)i * p.codebook_size + code) * p.C + c];
            }
        }
        tile[cc][tt] = acc;
    }
    __syncthreads();
    int tt = tid & 31;
    for (int c2 = tid >> 5; c2 < 32; c2 += 8) {
        int t = t0 + tt, c = c0 + c2;
        if (t < p.T0 && c < p.C) zq[((size_t)b * p.C + c) * p.T0 + t] = tile[c2][tt];
    }
}

// ---- depthwise conv k=7 ---------------------------------------------------------------------------
#define DW_TILE 1024
#define DW_HALO 27      // 3 * max dilation (9)

template <bool SNAKE_IN, bool SNAKE_OUT>
__global__ void __launch_bounds__(256) k_snac_dw(const float* __restrict__ X, float* __restrict__ Y,
                                                 const float* __restrict__ w /*[C][7]*/,
                                                 const float* __restrict__ bias,
                                                 const float* __restrict__ a_in, const float* __restrict__ ra_in,
                                                 const float* __restrict__ a_out, const float* __restrict__ ra_out,
                                                 int C, int T, int dil) {
    __shared__ float s[DW_TILE + 2 * DW_HALO];
    int b = blockIdx.z, c = blockIdx.y, t0 = blockIdx.x * DW_TILE;
    const float* x = X + ((size_t)b * C + c) * T;
    float* y = Y + ((size_t)b * C + c) * T;
    int halo = 3 * dil;
    float ai = 0.f, rai = 0.f;
    if (SNAKE_IN) { ai = a_in[c]; rai = ra_in[c]; }
    for (int i = threadIdx.x; i < DW_TILE + 2 * halo; i += 256) {
        int t = t0 - halo + i;
        float v = (t >= 0 && t < T) ? x[t] : 0.0f;     // zero padding AFTER snake == snake(0) = 0
        if (SNAKE_IN) v = snake_f(v, ai, rai);
        s[i] = v;
    }
    __syncthreads();
    float wk[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) wk[k] = w[c * 7 + k];
    float bv = bias[c];
    float ao = 0.f, rao = 0.f;
    if (SNAKE_OUT) { ao = a_out[c]; rao = ra_out[c]; }
    for (int i = threadIdx.x; i < DW_TILE; i += 256) {
        int t = t0 + i;
        if (t >= T) break;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 7; ++k) acc += wk[k] * s[i + k * dil];
        acc += bv;
        if (SNAKE_OUT) acc = snake_f(acc, ao, rao);
        y[t] = acc;
    }
}

// ---- LocalMHA (Attention.swift:14-94): the 32 / 44 kHz models ---------------------------------------------------------
// LayerNorm over the channels of every column (NCT data: a thread owns one column, a wave reads 64 consecutive columns of a channel)
__global__ void __launch_bounds__(128) k_snac_ln_ct(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ w,
                                                    const float* __restrict__ bsh, int C, int T, float eps) {
    const int t = blockIdx.x * 128 + threadIdx.x, b = blockIdx.y;
    if (t >= T) return;
    const float* xb = x + (size_t)b * C * T + t;
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += xb[(size_t)c * T];
    const float mu = s / (float)C;
    float v = 0.0f;
    for (int c = 0; c < C; ++c) { const float d = xb[(size_t)c * T] - mu; v += d * d; }
    const float inv = rsqrtf(v / (float)C + eps);
    float* yb = y + (size_t)b * C * T + t;
    for (int c = 0; c < C; ++c) yb[(size_t)c * T] = (xb[(size_t)c * T] - mu) * inv * w[c] + bsh[c];
}
// attention inside one window of WIN frames for one head (dim_head 64): qkv [B][3C][T] (rows: q | k | v, head h at rows h*64 ..),
// rotary embedding of q and k over the in-window position (freqs = [n f, n f], rotate_half = [-x2, x1]; xpos off: scale 1),
// softmax(q k^T / 8) v  ->  out [B][C][T].  Block = (window, head, batch), 256 threads.
#define MHA_D 64
#define MHA_MAXW 64
__global__ void __launch_bounds__(256) k_snac_local_attn(const float* __restrict__ qkv, float* __restrict__ out, const float* __restrict__ inv_freq,
                                                         int C, int T, int win) {
    __shared__ float q[MHA_D][MHA_MAXW + 1], k[MHA_D][MHA_MAXW + 1], v[MHA_D][MHA_MAXW + 1], p[MHA_MAXW][MHA_MAXW + 1];
    const int w0 = blockIdx.x * win, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const float* base = qkv + (size_t)b * 3 * C * T;
    for (int i = tid; i < MHA_D * win; i += 256) {
        const int d = i / win, n = i - d * win;
        const int dp = d < 32 ? d + 32 : d - 32;                      // rotate_half partner
        const float sg = d < 32 ? -1.0f : 1.0f;
        const float f = (float)n * inv_freq[d & 31];
        const float cs = cosf(f), sn = sinf(f);
        const size_t col = (size_t)w0 + n;
        const float qd = base[((size_t)(h * MHA_D + d)) * T + col], qp = base[((size_t)(h * MHA_D + dp)) * T + col];
        const float kd = base[((size_t)(C + h * MHA_D + d)) * T + col], kp = base[((size_t)(C + h * MHA_D + dp)) * T + col];
        q[d][n] = qd * cs + sg * qp * sn;
        k[d][n] = kd * cs + sg * kp * sn;
        v[d][n] = base[((size_t)(2 * C + h * MHA_D + d)) * T + col];
    }
    __syncthreads();
    for (int i = tid; i < win * win; i += 256) {
        const int a = i / win, c = i - a * win;                       // query a, key c
        float acc = 0.0f;
#pragma unroll 8
        for (int d = 0; d < MHA_D; ++d) acc += q[d][a] * k[d][c];
        p[a][c] = acc * 0.125f;
    }
    __syncthreads();
    if (tid < win) {                                                  // softmax over the keys of query tid
        float m = -INFINITY;
        for (int c = 0; c < win; ++c) m = fmaxf(m, p[tid][c]);
        float sum = 0.0f;
        for (int c = 0; c < win; ++c) { const float e = expf(p[tid][c] - m); p[tid][c] = e; sum += e; }
        const float r = 1.0f / sum;
        for (int c = 0; c < win; ++c) p[tid][c] *= r;
    }
    __syncthreads();
    float* ob = out + (size_t)b * C * T;
    for (int i = tid; i < MHA_D * win; i += 256) {
        const int d = i / win, a = i - d * win;
        float acc = 0.0f;
        for (int c = 0; c < win; ++c) acc += p[a][c] * v[d][c];
        ob[((size_t)(h * MHA_D + d)) * T + w0 + a] = acc;
    }
}

// ---- fused narrow ResidualUnit tails -------------------------------------------------------------------------------------------------
// DW = true  (SNAC, Layers.swift:202-231):  y = x + W2 . snake2(dw7_dil(snake1(x)) + b1) + b2  for C = 64 / 128 channels in ONE pass over
//            the tensor: as two kernels the unit reads x, writes t, reads t, reads x again and writes y (five passes at 3.0-3.5 TB/s;
//            the two late decoder blocks are HBM-bound, not MFMA-bound).
// DW = false (Qwen3-TTS / DAC decoder units): y = r + W2 . snake(t) + b2, the 1x1 conv with Snake prologue and residual epilogue over
//            C <= 192 channels with the WHOLE channel range in one block: the generic GEMM runs ceil(C / 64) row blocks that each
//            re-stage (and re-Snake) the same input tile, and 96 rows fill only 1.5 of them.
// Per 128-column tile and 32-channel chunk: the input (DW: with its dilation halo) is staged once, Snake applied on the way in; DW:
// the depthwise taps read it from LDS; the chunk of t goes straight into the B-operand tile of the exact-f32 MFMA contraction with
// the chunk's rows of W2^T.  The residual is read at the end (DW: L2-hot).  Same arithmetic per element as the unfused kernels.
#define RU_NT 128
#define RU_KC 32
#define RU_XS (RU_NT + 2 * DW_HALO + 2)
struct PwFusedParams {
    const float* X; const float* R; float* Y;
    const float *w7, *b1, *a1, *ra1;       // DW: depthwise weights [C][7], bias, Snake before the taps
    const float *a2, *ra2;                 // Snake in front of the 1x1 conv
    const float *AT, *b2;                  // W2^T [C][C] ([k][m]), bias
    int T, dil, ldx, ldr, ldy;
};
// The body takes restrict-qualified PARAMETERS (clang ignores restrict on struct members and on locals; after inlining the parameter
// attributes survive as scoped-noalias metadata): without them every residual / bias load of the epilogue has to wait behind the
// previous row's store (measured: the SNAC units 2.25 -> 2.83 ms when the pointers were read from the parameter struct directly).
struct PwFusedScalars { int T, dil, ldx, ldr, ldy; };
template <int C, bool DW>
__device__ __forceinline__ void pw_fused_body(const float* __restrict__ pX, const float* __restrict__ pR, float* __restrict__ pY,
                                              const float* __restrict__ pw7, const float* __restrict__ pb1, const float* __restrict__ pa1,
                                              const float* __restrict__ pra1, const float* __restrict__ pa2, const float* __restrict__ pra2,
                                              const float* __restrict__ pAT, const float* __restrict__ pb2, const PwFusedScalars sc) {
    struct { const float *X, *R; float* Y; const float *w7, *b1, *a1, *ra1, *a2, *ra2, *AT, *b2; int T, dil, ldx, ldr, ldy; } p =
        {pX, pR, pY, pw7, pb1, pa1, pra1, pa2, pra2, pAT, pb2, sc.T, sc.dil, sc.ldx, sc.ldr, sc.ldy};
    __shared__ float xs[DW ? RU_KC : 1][DW ? RU_XS : 1];
    __shared__ float ts[RU_KC][RU_NT];
    __shared__ float As[RU_KC][C];
    constexpr bool SQ = C == 128;                         // 2 x 2 waves of 64 x 64; else every wave: all C rows x 32 columns
    constexpr int MTW = SQ ? 2 : C / 32, NTW = SQ ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * RU_NT, b = blockIdx.y;
    const int wr = SQ ? (wave >> 1) * 64 : 0;             // first row / first column of the wave's tile
    const int wc = SQ ? (wave & 1) * 64 : wave * 32;
    const float* xb = p.X + (size_t)b * C * p.ldx;
    const int T = p.T, halo = 3 * p.dil, ncols = RU_NT + 2 * halo;
    f32x16_t acc[MTW][NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    for (int c0 = 0; c0 < C; c0 += RU_KC) {
        __syncthreads();                                   // the previous chunk's MFMAs are done with ts / As
        if (DW) {
            // x chunk with halo, Snake on the way in (zero padding AFTER Snake: snake(0) = 0).  All 24 loads of a thread are issued
            // before the first use: lane -> column within a 64-column strip (3 strips cover 128 + 54), wave -> rows wave, wave + 4, ...
            // (fetching the next chunk under this chunk's MFMAs was measured too: no gain at 128 channels, 8 % slower at 64)
            float v[8][3];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int c = c0 + wave + 4 * r;
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    const int j = sx * 64 + lane, t = n0 - halo + j;
                    const bool ok = j < ncols && t >= 0 && t < T;
                    v[r][sx] = xb[(size_t)c * p.ldx + (ok ? t : 0)];
                    v[r][sx] = ok ? v[r][sx] : 0.0f;
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int cc = wave + 4 * r, c = c0 + cc;
                const float al = p.a1[c], ral = p.ra1[c];
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    const int j = sx * 64 + lane;
                    if (j < RU_XS) xs[cc][j] = snake_f(v[r][sx], al, ral);
                }
            }
        } else {
            // the chunk of the 1x1 conv's input, Snake on the way in: 16 loads per thread up front (lane -> column, wave -> rows)
            float v[8][2];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int c = c0 + wave + 4 * r;
#pragma unroll
                for (int hx = 0; hx < 2; ++hx) {
                    const int t = n0 + hx * 64 + lane;
                    v[r][hx] = xb[(size_t)c * p.ldx + (t < T ? t : 0)];
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int cc = wave + 4 * r, c = c0 + cc;
                const float al = p.a2[c], ral = p.ra2[c];
#pragma unroll
                for (int hx = 0; hx < 2; ++hx) ts[cc][hx * 64 + lane] = snake_f(v[r][hx], al, ral);
            }
        }
        for (int i = tid; i < RU_KC * (C / 4); i += 256) { // rows c0 .. c0+31 of W2^T
            const int r = i / (C / 4), m4 = (i - r * (C / 4)) * 4;
            *reinterpret_cast<float4*>(&As[r][m4]) = *reinterpret_cast<const float4*>(p.AT + (size_t)(c0 + r) * C + m4);
        }
        if (DW) {
            __syncthreads();
            // depthwise taps + bias + Snake -> the chunk of t.  lane -> column (consecutive lanes read consecutive LDS words: no bank
            // conflicts), wave -> channels wave, wave + 4, ...
#pragma unroll 2
            for (int r = 0; r < 8; ++r) {
                const int cc = wave + 4 * r, c = c0 + cc;
                float wk[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) wk[k] = p.w7[c * 7 + k];
                const float bv = p.b1[c], ao = p.a2[c], rao = p.ra2[c];
#pragma unroll
                for (int hx = 0; hx < 2; ++hx) {
                    const int n = hx * 64 + lane;
                    float a = 0.0f;
#pragma unroll
                    for (int k = 0; k < 7; ++k) a += wk[k] * xs[cc][n + k * p.dil];
                    a += bv;
                    ts[cc][n] = snake_f(a, ao, rao);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < RU_KC; kk += 2) {
            float av[MTW], bv2[NTW];
#pragma unroll
            for (int mi = 0; mi < MTW; ++mi) av[mi] = As[kk + (lane >> 5)][wr + mi * 32 + (lane & 31)];
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni) bv2[ni] = ts[kk + (lane >> 5)][wc + ni * 32 + (lane & 31)];
#pragma unroll
            for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
                for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv2[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    // epilogue: + bias + residual (C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
    const float* rb = p.R + (size_t)b * C * p.ldr;
    float* yb = p.Y + (size_t)b * C * p.ldy;
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) {
        const int n = n0 + wc + ni * 32 + (lane & 31);
        if (n >= T) continue;
#pragma unroll
        for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wr + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                yb[(size_t)m * p.ldy + n] = rb[(size_t)m * p.ldr + n] + (acc[mi][ni][r] + p.b2[m]);
            }
    }
}
template <int C, bool DW>
__global__ void __launch_bounds__(256, 2) k_pw_fused(PwFusedParams p) {
    pw_fused_body<C, DW>(p.X, p.R, p.Y, p.w7, p.b1, p.a1, p.ra1, p.a2, p.ra2, p.AT, p.b2, PwFusedScalars{p.T, p.dil, p.ldx, p.ldr, p.ldy});
}

// ---- dense contraction on f32 MFMA ----------------------------------------------------------------
#define G_BM 64
#define G_BN 128
#define G_BK 16                       // k rows per staged chunk.  32 (twice the MFMA steps per barrier pair, twice the staging registers) was
                                      // measured and is SLOWER: SNAC decode of the bench 48.6 -> 53.8 ms (profiles/r02_codec_bk_ab.json)
#define G_AH (G_BK / 16)              // staged A float4 per thread: rows ar + 16 h
#define G_XH (G_BK / 8)               // staged X float4 per thread: rows xr + 8 h

// (body behind restrict-qualified parameters for the arrays the epilogue touches: see k_pw_fused)
template <int MODE, bool SNAKE>
__device__ __forceinline__ void snac_gemm_body(const GemmParams& p, const float* __restrict__ pX, const float* __restrict__ pR,
                                               float* __restrict__ pY, const float* __restrict__ pbias, const float* __restrict__ pscale) {
    __shared__ float As[G_BK][G_BM];
    __shared__ float Xs[G_BK][G_BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * G_BN, m0 = blockIdx.y * G_BM;
    int b, phase = 0;
    if (MODE == GEMM_CONVT) { b = blockIdx.z / p.s; phase = blockIdx.z % p.s; }
    else b = blockIdx.z;
    const int Kx = (MODE == GEMM_CONVT || MODE == GEMM_TAPS) ? p.Cin : p.K;
    const float* AT = p.AT + (size_t)phase * p.K * p.M;
    const float* Xb = pX + (size_t)b * Kx * p.ldx;
    // transposed conv phase: out o = s*n + phase takes taps j=0,1 from x[:, n + q - j]
    const int q = (MODE == GEMM_CONVT) ? (phase + p.pad) / p.s : 0;

    f32x16_t acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // staging registers: A: G_AH float4 / thread, X: G_XH float4 / thread
    const int ar = tid >> 4, ac = (tid & 15) * 4;          // A tile rows ar + 16 h (k), col (m)
    const int xr = tid >> 5, xc = (tid & 31) * 4;          // X tile rows xr + 8 h, col (n)
    float ra[G_AH][4], rx[G_XH][4];

    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int h = 0; h < G_AH; ++h) {   // A^T tile
            int k = k0 + ar + 16 * h, m = m0 + ac;
            if (k < p.K && (p.M & 3) == 0 && m + 3 < p.M) {
                float4 v = *reinterpret_cast<const float4*>(AT + (size_t)k * p.M + m);
                ra[h][0] = v.x; ra[h][1] = v.y; ra[h][2] = v.z; ra[h][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[h][e] = (k < p.K && m + e < p.M) ? AT[(size_t)k * p.M + m + e] : 0.0f;
            }
        }
#pragma unroll
        for (int h = 0; h < G_XH; ++h) {   // X tile
            int k = k0 + xr + 8 * h;
            int row = k, shift = 0;
            if (MODE == GEMM_CONVT) { int j = k / p.Cin; row = k - j * p.Cin; shift = q - j; }
            if (MODE == GEMM_TAPS) { int j = k / p.Cin; row = k - j * p.Cin; shift = (j - (p.taps - 1)) * p.dil; }
            int n = n0 + xc + shift;
            bool kin = k < p.K;
            const float* src = Xb + (size_t)row * p.ldx;
            if (MODE != GEMM_CONVT && MODE != GEMM_TAPS && kin && (p.ldx & 3) == 0 && n + 3 < p.Tin) {
                float4 v = *reinterpret_cast<const float4*>(src + n);
                rx[h][0] = v.x; rx[h][1] = v.y; rx[h][2] = v.z; rx[h][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int nn = n + e;
                    rx[h][e] = (kin && nn >= p.x_lo && nn < p.Tin) ? src[nn] : 0.0f;
                }
            }
            if (SNAKE && kin) {
                float a = p.alpha[row], r = p.ralpha[row];
#pragma unroll
                for (int e = 0; e < 4; ++e) rx[h][e] = snake_f(rx[h][e], a, r);   // snake(0)=0 keeps the zero pad
            }
        }
    };

    const int nchunks = (p.K + G_BK - 1) / G_BK;
    load_chunk(0);
    for (int kc = 0; kc < nchunks; ++kc) {
        __syncthreads();
#pragma unroll
        for (int h = 0; h < G_AH; ++h)
            *reinterpret_cast<float4*>(&As[ar + 16 * h][ac]) = make_float4(ra[h][0], ra[h][1], ra[h][2], ra[h][3]);
#pragma unroll
        for (int h = 0; h < G_XH; ++h)
            *reinterpret_cast<float4*>(&Xs[xr + 8 * h][xc]) = make_float4(rx[h][0], rx[h][1], rx[h][2], rx[h][3]);
        __syncthreads();
        if (kc + 1 < nchunks) load_chunk((kc + 1) * G_BK);
#pragma unroll
        for (int kk = 0; kk < G_BK; kk += 2) {
            // v_mfma_f32_32x32x2_f32: A lane l = A[i=l&31][k=l>>5], B lane l = B[k=l>>5][j=l&31]
            float a = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float bv = Xs[kk + (lane >> 5)][wn * 64 + t * 32 + (lane & 31)];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
            }
        }
    }

    // NoiseBlock noise for this lane's two columns (MLXRandom.normal([B,1,T]), Layers.swift:274):
    // explicit tensor, or Box-Muller on mis-synth-v1 uniforms keyed by (seed, block, GLOBAL row, t)
    float nzv[2] = {0.0f, 0.0f};
    if (MODE == GEMM_NOISE) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int n = n0 + wn * 64 + t * 32 + (lane & 31);
            if (n >= p.N) continue;
            if (p.noise) nzv[t] = p.noise[(size_t)b * p.N + n];
            else if (p.noise_rng) {
                uint64_t row = (uint64_t)(p.row_offset + (p.row_ids ? p.row_ids[b] : b));
                uint64_t u = mis_splitmix64((p.noise_key ^ (row * 0xD1B54A32D192ED03ull)) + (uint64_t)n);
                float u1 = ((float)(uint32_t)(u >> 40) + 0.5f) * 5.9604644775390625e-08f;
                float u2 = ((float)(uint32_t)((u >> 16) & 0xFFFFFF) + 0.5f) * 5.9604644775390625e-08f;
                nzv[t] = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
            }
        }
    }
    // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int n = n0 + wn * 64 + t * 32 + (lane & 31);
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= p.M) continue;
            float v = acc[t][r];
            if (pbias) v += pbias[m];
            if (MODE == GEMM_PLAIN || MODE == GEMM_TAPS) {
                pY[((size_t)b * p.M + m) * p.ldy + n] = v;
            } else if (MODE == GEMM_GELU) {                            // exact-erf GELU (VocosBackbone.swift:89)
                pY[((size_t)b * p.M + m) * p.ldy + n] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
            } else if (MODE == GEMM_RESID) {
                size_t o = ((size_t)b * p.M + m) * p.ldy + n;
                if (pscale) v *= pscale[m];                          // ConvNeXt layer scale gamma (VocosBackbone.swift:92-95)
                pY[o] = pR[o] + v;                                   // Layers.swift:230
            } else if (MODE == GEMM_NOISE) {
                size_t o = ((size_t)b * p.M + m) * p.ldy + n;
                pY[o] = Xb[(size_t)m * p.ldx + n] + nzv[t] * v;       // Layers.swift:276-277
            } else {
                int o = p.s * n + phase;
                if (n == 0 && p.dup_bias_n0 && pbias) v += pbias[m];   // streaming overlap-add counts the bias twice (GemmParams)
                if (o < p.Tout) pY[((size_t)b * p.M + m) * p.ldy + o] = v;
            }
        }
    }
}
template <int MODE, bool SNAKE>
__global__ void __launch_bounds__(256) k_snac_gemm(GemmParams p) { snac_gemm_body<MODE, SNAKE>(p, p.X, p.R, p.Y, p.bias, p.scale); }

// ---- encoder kernels (Layers.swift:319-360, VQ.swift:47-120) ---------------------------------------------------------
// first layer: conv k7 "same", 1 -> C
__global__ void k_enc_first(const float* __restrict__ audio, float* __restrict__ y, const float* __restrict__ w /*[C][7]*/,
                            const float* __restrict__ bias, int C, int T) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const float* a = audio + (size_t)b * T;
    float acc = bias[c];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        int ts = t + j - 3;
        if (ts >= 0 && ts < T) acc += w[c * 7 + j] * a[ts];
    }
    y[((size_t)b * C + c) * T + t] = acc;
}
// Snake, then phase split for a stride-s conv: y[(c*s + r)][m] = snake(x[c][s*m + r])   (T = s*Tm)
__global__ void k_enc_phase_split(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ a, const float* __restrict__ ra,
                                  int C, int T, int s) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const int Tm = T / s, m = t / s, r = t - m * s;
    float v = snake_f(x[((size_t)b * C + c) * T + t], a[c], ra[c]);
    y[((size_t)b * C * s + (size_t)c * s + r) * Tm + m] = v;
}
// average pool over `stride` consecutive samples (VQ.swift:47-57)
__global__ void k_vq_pool(const float* __restrict__ x, float* __restrict__ y, int C, int T, int stride) {
    int m = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    const int Tm = T / stride;
    if (m >= Tm) return;
    const float* xr = x + ((size_t)b * C + c) * T + (size_t)m * stride;
    float acc = 0.0f;
    for (int j = 0; j < stride; ++j) acc += xr[j];
    y[((size_t)b * C + c) * Tm + m] = acc / (float)stride;
}
// nearest codebook entry of the L2-normalised latent (VQ.swift:96-120): dist = |e|^2 - 2 e.c + |c|^2 on normalised vectors,
// argmax(-dist) with the first index on ties.  One block per (b, t); ze [B][CD][Tm]; codes [B][Tm].
__global__ void __launch_bounds__(256) k_vq_nearest(const float* __restrict__ ze, const float* __restrict__ cn, const float* __restrict__ cn2,
                                                    int32_t* __restrict__ codes, int CD, int CB, int Tm) {
    __shared__ float e[64];
    __shared__ float redv[4];
    __shared__ int redi[4];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (tid < CD) e[tid] = ze[((size_t)b * CD + tid) * Tm + t];
    __syncthreads();
    float n2 = 0.0f;
    for (int d = 0; d < CD; ++d) n2 += e[d] * e[d];
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
    float en2 = 0.0f;
    for (int d = 0; d < CD; ++d) { float v = e[d] * inv; en2 += v * v; }
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int k = tid; k < CB; k += 256) {
        float dot = 0.0f;
        for (int d = 0; d < CD; ++d) dot += (e[d] * inv) * cn[(size_t)k * CD + d];
        float dist = en2 - 2.0f * dot + cn2[k];
        if (dist < best) { best = dist; bi = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if ((tid & 63) == 0) { redv[tid >> 6] = best; redi[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) if (redv[w] < best || (redv[w] == best && redi[w] < bi)) { best = redv[w]; bi = redi[w]; }
        codes[(size_t)b * Tm + t] = bi;
    }
}
// residual -= repeat_interleave(out_proj(codebook[code]), stride)   (VQ.swift:60-80,151-158); table [CB][C]
__global__ void k_vq_residual(float* __restrict__ r, const int32_t* __restrict__ codes, const float* __restrict__ table, int C, int T, int stride) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    int code = codes[(size_t)b * (T / stride) + t / stride];
    r[((size_t)b * C + c) * T + t] -= table[(size_t)code * C + c];
}

// ---- dense k-tap conv on f32 MFMA (Qwen3-TTS / DAC-style decoders: dilated k7 convs with Cin = Cout up to 1536) ---------
// y[m][n] = bias[m] + sum_{j < taps} sum_c W[m][j][c] * act(x[c][n - pad_left + j*dil]),  zero outside [0, Tin).
// Loop order (channel chunk outer, tap inner): the activation tile WITH its halo is staged once per 16 channels (aligned
// float4 loads, Snake applied on the way in) and every tap reads it from LDS at a column offset, so x is fetched once instead
// of `taps` times and never with scalar loads; the `taps` weight tiles of the chunk are staged together (2 barriers per chunk
// for taps*16 MFMA steps).  A^T row (j*Cin + c), as for the other modes.
#define CT_XS 224                     // staged columns per row: G_BN + halo (<= 92) + alignment slack (<= 3)
#define CT_MAXT 7
#define CT_BK 16                      // input channels per staged chunk (7 tap tiles of 16 x 64 + the halo tile: 43 KB of LDS)
template <bool RESID>
__device__ __forceinline__ void conv_taps_body(const GemmParams& p, const float* __restrict__ pX, const float* __restrict__ pR,
                                               float* __restrict__ pY, const float* __restrict__ pbias, const float* __restrict__ pscale) {
    __shared__ float As[CT_MAXT][CT_BK][G_BM];
    __shared__ float Xs[CT_BK][CT_XS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * G_BN, m0 = blockIdx.y * G_BM, b = blockIdx.z;
    const float* Xb = pX + (size_t)b * p.Cin * p.ldx;
    const int halo = (p.taps - 1) * p.dil;
    const int c_lo = n0 - p.pad;                       // first needed input column (may be negative)
    const int a0 = (c_lo >= 0 ? c_lo : c_lo - 3) / 4 * 4;     // aligned-down staging origin
    const int off = c_lo - a0;                         // 0..3
    const int ncols = off + G_BN + halo;               // staged columns
    const bool vec = (p.ldx & 3) == 0;
    f32x16_t acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    // staging registers: A: taps float4 per thread (row tid>>4, cols (tid&15)*4); X: 4 float4 per thread
    const int ar = tid >> 4, ac = (tid & 15) * 4;
    float4 ra[CT_MAXT], rx[4];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int j = 0; j < CT_MAXT; ++j) {
            if (j >= p.taps) break;
            const int k = j * p.Cin + c0 + ar, m = m0 + ac;
            if (c0 + ar < p.Cin && (p.M & 3) == 0 && m + 3 < p.M) ra[j] = *reinterpret_cast<const float4*>(p.AT + (size_t)k * p.M + m);
            else {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (c0 + ar < p.Cin && m + e < p.M) ? p.AT[(size_t)k * p.M + m + e] : 0.0f;
                ra[j] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                    // 16 rows x 56 float4
            const int idx = tid + i * 256;
            const int row = idx / (CT_XS / 4), c4 = (idx - row * (CT_XS / 4)) * 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (row < CT_BK && c0 + row < p.Cin && c4 < ncols) {
                const float* src = Xb + (size_t)(c0 + row) * p.ldx;
                const int g = a0 + c4;
                if (vec && g >= p.x_lo && g + 3 < p.Tin) { float4 t = *reinterpret_cast<const float4*>(src + g); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (g + e >= p.x_lo && g + e < p.Tin) ? src[g + e] : 0.0f;
                }
                if (p.alpha) {
                    const float al = p.alpha[c0 + row], ra_ = p.ralpha[c0 + row];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = snake_f(v[e], al, ra_);      // snake(0) = 0 keeps the zero padding
                }
            }
            rx[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    const int nchunks = (p.Cin + CT_BK - 1) / CT_BK;
    load_chunk(0);
    for (int cc = 0; cc < nchunks; ++cc) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CT_MAXT; ++j) {
            if (j >= p.taps) break;
            *reinterpret_cast<float4*>(&As[j][ar][ac]) = ra[j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / (CT_XS / 4), c4 = (idx - row * (CT_XS / 4)) * 4;
            if (row < CT_BK) *reinterpret_cast<float4*>(&Xs[row][c4]) = rx[i];
        }
        __syncthreads();
        if (cc + 1 < nchunks) load_chunk((cc + 1) * CT_BK);
        for (int j = 0; j < p.taps; ++j) {
            const int xo = off + j * p.dil + wn * 64 + (lane & 31);
#pragma unroll
            for (int kk = 0; kk < CT_BK; kk += 2) {
                float a = As[j][kk + (lane >> 5)][wm * 32 + (lane & 31)];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float bv = Xs[kk + (lane >> 5)][xo + t * 32];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int n = n0 + wn * 64 + t * 32 + (lane & 31);
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= p.M) continue;
            float v = acc[t][r];
            if (pbias) v += pbias[m];
            size_t o = ((size_t)b * p.M + m) * p.ldy + n;
            if (RESID) { if (pscale) v *= pscale[m]; v += pR[o]; }
            pY[o] = v;
        }
    }
}
template <bool RESID>
__global__ void __launch_bounds__(256) k_conv_taps(GemmParams p) { conv_taps_body<RESID>(p, p.X, p.R, p.Y, p.bias, p.scale); }

// ---- final Snake -> conv k7 (C -> 1) -> tanh ----------------------------------------------------------
#define FIN_TILE 256
#define FIN_CH 16
__global__ void __launch_bounds__(256) k_snac_final(const float* __restrict__ X, float* __restrict__ out,
                                                    int64_t out_stride, const float* __restrict__ w /*[C][7]*/,
                                                    float bias, const float* __restrict__ alpha,
                                                    const float* __restrict__ ralpha, int C, int T) {
    __shared__ float s[FIN_CH][FIN_TILE + 8];
    int b = blockIdx.y, t0 = blockIdx.x * FIN_TILE, tid = threadIdx.x;
    float acc = 0.0f;
    for (int c0 = 0; c0 < C; c0 += FIN_CH) {
        __syncthreads();
        for (int i = tid; i < FIN_CH * (FIN_TILE + 6); i += 256) {
            int cc = i / (FIN_TILE + 6), j = i % (FIN_TILE + 6);
            int c = c0 + cc, t = t0 - 3 + j;
            float v = 0.0f;
            if (c < C && t >= 0 && t < T) v = snake_f(X[((size_t)b * C + c) * T + t], alpha[c], ralpha[c]);
            s[cc][j] = v;
        }
        __syncthreads();
        int cmax = min(FIN_CH, C - c0);
        for (int cc = 0; cc < cmax; ++cc) {
            const float* wc = w + (size_t)(c0 + cc) * 7;
#pragma unroll
            for (int k = 0; k < 7; ++k) acc += wc[k] * s[cc][tid + k];
        }
    }
    int t = t0 + tid;
    if (t < T) out[(size_t)b * out_stride + t] = tanhf(acc + bias);
}

// ============================================================================ host side

struct HostTensor {
    std::vector<float> v;
    std::vector<int64_t> shape;
};

struct ConvW { size_t w = 0, b = 0; bool has_bias = false; };        // offsets into the weight arena (floats)
struct SnakeW { size_t a = 0, ra = 0; };

struct mis_snac {
    int device = 0;
    hipStream_t stream = nullptr;
    mis_snac_config cfg{};
    std::map<std::string, HostTensor> raw;
    bool finalized = false;

    DevBuf<float> arena;      // folded weights
    size_t tables_off = 0;
    ConvW stem_dw, stem_pw, fin;
    float fin_bias = 0.f;
    SnakeW fin_snake;
    struct Block {
        int cin, cout, stride, pad;
        SnakeW snake0;
        ConvW convT, noise;
        struct RU { SnakeW s1, s2; ConvW dw, pw; } ru[3];
    };
    std::vector<Block> blocks;
    // encoder (built when "encoder.*" tensors were provided): Layers.swift:236-259,319-360
    struct EncBlock {
        int cin, cout, stride;
        struct RU { SnakeW s1, s2; ConvW dw, pw; } ru[3];
        SnakeW snake;
        ConvW down;                       // strided conv as a 3-tap conv over the phase-split tensor: A^T [3*s*cin][cout]
    };
    // LocalMHA of the 32 / 44 kHz models (Attention.swift): decoder layer 2, encoder layer n + 1
    struct MhaW { bool on = false; int dim = 0; size_t ln_w = 0, ln_b = 0, qkv = 0, out = 0, inv_freq = 0; };
    MhaW dec_attn, enc_attn;
    bool has_encoder = false;
    int enc_dim = 0;
    ConvW enc_first, enc_last;            // first: [d][7] (Cin = 1); last: depthwise [C][7]
    std::vector<EncBlock> enc_blocks;
    struct VqEnc { ConvW in_proj; size_t cn = 0, cn2 = 0; };    // normalised codebook [CB][CD] and its squared norms
    std::vector<VqEnc> vq_enc;
    DevBuf<float> enc_buf[3], vq_pool, vq_ze;
    DevBuf<int32_t> enc_codes;

    // workspaces
    DevBuf<float> buf[3];
    CodecPack pack;                      // split-bf16 weight fragments + activation scratch (codec_bf3.hip)
    DevBuf<int32_t> codes_ws;
    DevBuf<float> noise_ws;
    size_t act_capacity = 0;
    // noise == NULL policy: 0 = draw N(0,1) internally (the reference's behaviour), 1 = add nothing
    int null_noise_zero = 0;
    uint64_t noise_seed = 0;
    // last-call memo for debug taps
    int last_batch = 0, last_t = 0;
    bool last_noise = false;
    std::vector<const int32_t*> last_codes;
    std::vector<const float*> last_noise_ptrs;
};

hipStream_t snac_stream(mis_snac* c) { return c->stream; }
int snac_device(const mis_snac* c) { return c->device; }
const mis_snac_config* snac_config(const mis_snac* c) { return &c->cfg; }

static const HostTensor& need(mis_snac* c, const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = c->raw.find(name);
    MIS_REQUIRE(it != c->raw.end(), MIS_ERR_NOT_INITIALIZED, "SNAC weight missing: %s", name.c_str());
    std::vector<int64_t> want(shape);
    if (it->second.shape != want) {
        std::string got, exp;
        for (auto d : it->second.shape) got += std::to_string(d) + ",";
        for (auto d : want) exp += std::to_string(d) + ",";
        throw MisError(MIS_ERR_INVALID_INPUT, "SNAC weight " + name + " has shape [" + got + "] expected [" + exp + "]");
    }
    return it->second;
}

// w = g * v / (||v||_(1,2) + eps)   (Layers.swift:35-42,102-103; eps = 0 for the transposed conv :166)
static std::vector<float> fold_weight_norm(const HostTensor& g, const HostTensor& v, float eps) {
    int64_t d0 = v.shape[0], inner = v.shape[1] * v.shape[2];
    std::vector<float> out(v.v.size());
    for (int64_t o = 0; o < d0; ++o) {
        float ss = 0.0f;
        for (int64_t i = 0; i < inner; ++i) { float x = v.v[o * inner + i]; ss += x * x; }
        float nrm = sqrtf(ss) + eps;
        float gg = g.v[o];
        for (int64_t i = 0; i < inner; ++i) out[o * inner + i] = gg * v.v[o * inner + i] / nrm;
    }
    return out;
}

static int64_t convt_out_len(int64_t T, int s) {
    int pad = (s + 1) / 2;
    return (T - 1) * s - 2 * pad + 2 * s;     // (T-1)s - 2pad + (K-1) + 1, K = 2s
}

extern "C" mis_status mis_snac_create(const mis_snac_config* cfg, int device, mis_snac** out) {
    MIS_API_BEGIN
    MIS_REQUIRE(cfg && out, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(cfg->depthwise == 1, MIS_ERR_INVALID_INPUT, "only depthwise SNAC decoders are supported");
    MIS_REQUIRE(cfg->attn_window_size >= 0 && cfg->attn_window_size <= MHA_MAXW, MIS_ERR_INVALID_INPUT, "attn_window_size must be 0 .. %d", MHA_MAXW);
    MIS_REQUIRE(cfg->n_decoder_rates >= 1 && cfg->n_decoder_rates <= 8 && cfg->n_codebooks >= 1 && cfg->n_codebooks <= 8,
                MIS_ERR_INVALID_INPUT, "bad decoder_rates / vq_strides count");
    for (int i = 0; i < cfg->n_codebooks; ++i)
        MIS_REQUIRE(cfg->vq_strides[i] >= 1 && cfg->vq_strides[0] % cfg->vq_strides[i] == 0, MIS_ERR_INVALID_INPUT,
                    "vq_strides must divide vq_strides[0]");
    for (int i = 0; i < cfg->n_decoder_rates; ++i)
        MIS_REQUIRE(cfg->decoder_rates[i] >= 1 && cfg->decoder_rates[i] <= 64, MIS_ERR_INVALID_INPUT, "bad decoder rate");
    MIS_REQUIRE(cfg->latent_dim > 0 && cfg->decoder_dim > 0 && cfg->codebook_size > 0 && cfg->codebook_dim > 0,
                MIS_ERR_INVALID_INPUT, "bad dimensions");
    int n = 0;
    HIP_CHECK(hipGetDeviceCount(&n));
    MIS_REQUIRE(device >= 0 && device < n, MIS_ERR_DEVICE, "device %d not available (%d GPUs visible)", device, n);
    HIP_CHECK(hipSetDevice(device));
    mis_snac* c = new mis_snac();
    c->device = device;
    c->cfg = *cfg;
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    *out = c;
    MIS_API_END
}

static int64_t mis_snac_hop_for_encoder(const mis_snac* c);
static void snac_local_mha(const mis_snac::MhaW& m, const float* W, int win, const float* x, float* y, float* t1, float* t2, int batch, int64_t T,
                           hipStream_t s);

extern "C" void mis_snac_destroy(mis_snac* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    delete c;
}

extern "C" mis_status mis_snac_set_tensor(mis_snac* c, const char* name, const void* data, mis_dtype dtype,
                                          const int64_t* shape, int ndim) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && name && data && shape && ndim >= 1 && ndim <= 4, MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(!c->finalized, MIS_ERR_INVALID_INPUT, "set_tensor after finalize");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { MIS_REQUIRE(shape[i] > 0, MIS_ERR_INVALID_INPUT, "bad shape"); n *= (size_t)shape[i]; t.shape.push_back(shape[i]); }
    t.v.resize(n);
    if (dtype == MIS_F32) memcpy(t.v.data(), data, n * 4);
    else if (dtype == MIS_BF16) { const uint16_t* s = (const uint16_t*)data; for (size_t i = 0; i < n; ++i) t.v[i] = bf16_to_f32(s[i]); }
    else if (dtype == MIS_F16) { const uint16_t* s = (const uint16_t*)data; for (size_t i = 0; i < n; ++i) t.v[i] = f16_to_f32_host(s[i]); }
    else throw MisError(MIS_ERR_INVALID_INPUT, "unsupported dtype for SNAC tensor");
    c->raw[name] = std::move(t);
    MIS_API_END
}

extern "C" mis_status mis_snac_finalize(mis_snac* c) {
    MIS_API_BEGIN
    MIS_REQUIRE(c, MIS_ERR_INVALID_INPUT, "null handle");
    MIS_REQUIRE(!c->finalized, MIS_ERR_INVALID_INPUT, "already finalized");
    HIP_CHECK(hipSetDevice(c->device));
    const mis_snac_config& cf = c->cfg;
    const int64_t D = cf.latent_dim, CB = cf.codebook_size, CD = cf.codebook_dim;
    std::vector<float> arena;
    auto push = [&](const std::vector<float>& v) { size_t o = arena.size(); arena.insert(arena.end(), v.begin(), v.end()); while (arena.size() & 3) arena.push_back(0.f); return o; };
    auto push_snake = [&](const std::string& name, int64_t C) {
        const HostTensor& a = need(c, name, {1, C, 1});
        std::vector<float> ra(C);
        for (int64_t i = 0; i < C; ++i) ra[i] = 1.0f / (a.v[i] + 1e-9f);
        SnakeW s;
        s.a = push(a.v);
        s.ra = push(ra);
        return s;
    };
    // 1x1 conv -> A^T [K=cin][M=cout]
    auto push_pw = [&](const std::string& p, int64_t cout, int64_t cin, bool bias) {
        std::vector<float> w = fold_weight_norm(need(c, p + ".weight_g", {cout, 1, 1}), need(c, p + ".weight_v", {cout, 1, cin}), 1e-12f);
        std::vector<float> at((size_t)cin * cout);
        for (int64_t o = 0; o < cout; ++o) for (int64_t i = 0; i < cin; ++i) at[i * cout + o] = w[o * cin + i];
        ConvW cw;
        cw.w = push(at);
        cw.has_bias = bias;
        if (bias) cw.b = push(need(c, p + ".bias", {cout}).v);
        return cw;
    };
    auto push_dw = [&](const std::string& p, int64_t C) {
        std::vector<float> w = fold_weight_norm(need(c, p + ".weight_g", {C, 1, 1}), need(c, p + ".weight_v", {C, 7, 1}), 1e-12f);
        ConvW cw;
        cw.w = push(w);          // [C][7]
        cw.has_bias = true;
        cw.b = push(need(c, p + ".bias", {C}).v);
        return cw;
    };

    // quantizer tables: table_i[code][c] = sum_d W_i[c][d] * codebook_i[code][d] + b_i[c]   (VQ.swift:88-94,170-172)
    {
        std::vector<float> tables((size_t)cf.n_codebooks * CB * D);
        for (int i = 0; i < cf.n_codebooks; ++i) {
            std::string p = "quantizer.quantizers." + std::to_string(i);
            const HostTensor& cb = need(c, p + ".codebook.weight", {CB, CD});
            std::vector<float> w = fold_weight_norm(need(c, p + ".out_proj.weight_g", {D, 1, 1}), need(c, p + ".out_proj.weight_v", {D, 1, CD}), 1e-12f);
            const HostTensor& bias = need(c, p + ".out_proj.bias", {D});
            for (int64_t code = 0; code < CB; ++code)
                for (int64_t ch = 0; ch < D; ++ch) {
                    float acc = 0.0f;
                    for (int64_t d = 0; d < CD; ++d) acc += w[ch * CD + d] * cb.v[code * CD + d];
                    tables[((size_t)i * CB + code) * D + ch] = acc + bias.v[ch];
                }
        }
        c->tables_off = push(tables);
    }
    // LocalMHA parameters: norm.{weight,bias} [dim], to_qkv.weight [3 dim][dim], to_out.weight [dim][dim] (Linear, no bias), rel_pos.inv_freq [32]
    auto push_mha = [&](const std::string& p, int64_t dim) {
        mis_snac::MhaW m;
        MIS_REQUIRE(dim % MHA_D == 0, MIS_ERR_INVALID_INPUT, "LocalMHA width %lld is not a multiple of the head size 64", (long long)dim);
        m.on = true; m.dim = (int)dim;
        m.ln_w = push(need(c, p + ".norm.weight", {dim}).v);
        m.ln_b = push(need(c, p + ".norm.bias", {dim}).v);
        auto lin_t = [&](const HostTensor& w, int64_t out_f, int64_t in_f) {            // Linear weight [out][in] -> A^T [in][out]
            std::vector<float> at((size_t)in_f * out_f);
            for (int64_t o = 0; o < out_f; ++o) for (int64_t i = 0; i < in_f; ++i) at[i * out_f + o] = w.v[o * in_f + i];
            return at;
        };
        m.qkv = push(lin_t(need(c, p + ".to_qkv.weight", {3 * dim, dim}), 3 * dim, dim));
        m.out = push(lin_t(need(c, p + ".to_out.weight", {dim, dim}), dim, dim));
        auto fq = c->raw.find(p + ".rel_pos.inv_freq");
        std::vector<float> inv(32);
        if (fq != c->raw.end() && fq->second.v.size() == 32) inv = fq->second.v;
        else for (int j = 0; j < 32; ++j) inv[j] = 1.0f / powf(10000.0f, (float)(2 * j) / 64.0f);      // SinusoidalEmbeddings.init (:105-107)
        m.inv_freq = push(inv);
        return m;
    };
    const std::string L = "decoder.model.layers";
    c->stem_dw = push_dw(L + ".0", D);
    c->stem_pw = push_pw(L + ".1", cf.decoder_dim, D, true);
    const int first_block = cf.attn_window_size > 0 ? 3 : 2;                                            // Layers.swift:395-397
    c->dec_attn = mis_snac::MhaW{};
    if (cf.attn_window_size > 0) c->dec_attn = push_mha(L + ".2", cf.decoder_dim);
    c->blocks.clear();
    for (int bi = 0; bi < cf.n_decoder_rates; ++bi) {
        mis_snac::Block blk;
        blk.cin = cf.decoder_dim >> bi;
        blk.cout = cf.decoder_dim >> (bi + 1);
        MIS_REQUIRE(blk.cout >= 1, MIS_ERR_INVALID_INPUT, "decoder_dim too small for %d blocks", cf.n_decoder_rates);
        blk.stride = cf.decoder_rates[bi];
        blk.pad = (blk.stride + 1) / 2;           // ceil(stride/2), Layers.swift:294
        std::string b = L + "." + std::to_string(first_block + bi) + ".block.layers";
        blk.snake0 = push_snake(b + ".0.alpha", blk.cin);
        {   // transposed conv, weight_v [in, 2s, out]; per phase p: AT[p][j*Cin+ci][co] = w[ci][((p+pad)%s) + j*s][co]
            int s = blk.stride, K = 2 * s;
            std::vector<float> w = fold_weight_norm(need(c, b + ".1.weight_g", {blk.cin, 1, 1}), need(c, b + ".1.weight_v", {blk.cin, K, blk.cout}), 0.0f);
            std::vector<float> at((size_t)s * 2 * blk.cin * blk.cout);
            for (int p = 0; p < s; ++p)
                for (int j = 0; j < 2; ++j) {
                    int k = ((p + blk.pad) % s) + j * s;
                    for (int ci = 0; ci < blk.cin; ++ci)
                        for (int co = 0; co < blk.cout; ++co)
                            at[(((size_t)p * 2 + j) * blk.cin + ci) * blk.cout + co] = w[((size_t)ci * K + k) * blk.cout + co];
                }
            blk.convT.w = push(at);
            blk.convT.has_bias = true;
            blk.convT.b = push(need(c, b + ".1.bias", {blk.cout}).v);
        }
        int idx = 2;
        if (cf.noise) { blk.noise = push_pw(b + ".2.linear", blk.cout, blk.cout, false); idx = 3; }
        for (int j = 0; j < 3; ++j) {
            std::string r = b + "." + std::to_string(idx + j) + ".block.layers";
            blk.ru[j].s1 = push_snake(r + ".0.alpha", blk.cout);
            blk.ru[j].dw = push_dw(r + ".1", blk.cout);
            blk.ru[j].s2 = push_snake(r + ".2.alpha", blk.cout);
            blk.ru[j].pw = push_pw(r + ".3", blk.cout, blk.cout, true);
        }
        c->blocks.push_back(blk);
    }
    {
        int n = first_block + cf.n_decoder_rates;
        int64_t cl = cf.decoder_dim >> cf.n_decoder_rates;
        c->fin_snake = push_snake(L + "." + std::to_string(n) + ".alpha", cl);
        std::string p = L + "." + std::to_string(n + 1);
        std::vector<float> w = fold_weight_norm(need(c, p + ".weight_g", {1, 1, 1}), need(c, p + ".weight_v", {1, 7, cl}), 1e-12f);
        std::vector<float> wt((size_t)cl * 7);       // [C][7]
        for (int64_t k = 0; k < 7; ++k) for (int64_t ci = 0; ci < cl; ++ci) wt[ci * 7 + k] = w[k * cl + ci];
        c->fin.w = push(wt);
        c->fin_bias = need(c, p + ".bias", {1}).v[0];
    }
    // ---- encoder (optional): dimensions are read off the tensors (encoder_dim / encoder_rates are not part of mis_snac_config)
    c->has_encoder = false;
    const std::string E = "encoder.block.layers";
    auto e0 = c->raw.find(E + ".0.weight_v");
    if (e0 != c->raw.end()) {
        MIS_REQUIRE(cf.depthwise, MIS_ERR_INVALID_INPUT, "SNAC encoder: only depthwise residual units are built");
        MIS_REQUIRE(e0->second.shape.size() == 3 && e0->second.shape[1] == 7 && e0->second.shape[2] == 1, MIS_ERR_INVALID_INPUT, "bad encoder stem");
        int64_t ch = e0->second.shape[0];
        c->enc_dim = (int)ch;
        {
            std::vector<float> w = fold_weight_norm(need(c, E + ".0.weight_g", {ch, 1, 1}), need(c, E + ".0.weight_v", {ch, 7, 1}), 1e-12f);
            c->enc_first.w = push(w); c->enc_first.has_bias = true; c->enc_first.b = push(need(c, E + ".0.bias", {ch}).v);
        }
        c->enc_blocks.clear();
        int li = 1;
        for (;; ++li) {
            const std::string b = E + "." + std::to_string(li) + ".block.layers";
            auto dn = c->raw.find(b + ".4.weight_v");
            if (dn == c->raw.end()) break;
            mis_snac::EncBlock eb;
            eb.cin = (int)ch;
            MIS_REQUIRE(dn->second.shape.size() == 3 && dn->second.shape[2] == ch && dn->second.shape[1] % 2 == 0, MIS_ERR_INVALID_INPUT, "bad encoder block %d", li);
            eb.cout = (int)dn->second.shape[0];
            eb.stride = (int)dn->second.shape[1] / 2;
            for (int j = 0; j < 3; ++j) {
                const std::string r = b + "." + std::to_string(j) + ".block.layers";
                eb.ru[j].s1 = push_snake(r + ".0.alpha", ch);
                eb.ru[j].dw = push_dw(r + ".1", ch);
                eb.ru[j].s2 = push_snake(r + ".2.alpha", ch);
                eb.ru[j].pw = push_pw(r + ".3", ch, ch, true);
            }
            eb.snake = push_snake(b + ".3.alpha", ch);
            {   // WNConv1d(k = 2s, stride s, pad ceil(s/2)) (Layers.swift:248-254) over the phase-split input [ch*s][T/s]:
                // y[n] = sum_{q in -1..1} sum_{c,r} W[co][s*q + r + pad][c] * xph[c*s + r][n + q]
                const int sdn = eb.stride, K = 2 * sdn, pad = (sdn + 1) / 2;
                std::vector<float> w = fold_weight_norm(need(c, b + ".4.weight_g", {eb.cout, 1, 1}), need(c, b + ".4.weight_v", {eb.cout, K, ch}), 1e-12f);
                std::vector<float> at((size_t)3 * sdn * ch * eb.cout, 0.0f);
                for (int qi = 0; qi < 3; ++qi)
                    for (int64_t ci = 0; ci < ch; ++ci)
                        for (int r = 0; r < sdn; ++r) {
                            const int j = sdn * (qi - 1) + r + pad;
                            if (j < 0 || j >= K) continue;
                            for (int co = 0; co < eb.cout; ++co)
                                at[(((size_t)qi * ch * sdn) + (size_t)ci * sdn + r) * eb.cout + co] = w[((size_t)co * K + j) * ch + ci];
                        }
                eb.down.w = push(at); eb.down.has_bias = true; eb.down.b = push(need(c, b + ".4.bias", {eb.cout}).v);
            }
            ch = eb.cout;
            c->enc_blocks.push_back(eb);
        }
        MIS_REQUIRE(!c->enc_blocks.empty() && ch == D, MIS_ERR_INVALID_INPUT, "SNAC encoder output width %lld != latent_dim %lld", (long long)ch, (long long)D);
        c->enc_attn = mis_snac::MhaW{};
        if (cf.attn_window_size > 0) { c->enc_attn = push_mha(E + "." + std::to_string(li), D); ++li; }      // Layers.swift:339-341
        c->enc_last = push_dw(E + "." + std::to_string(li), D);
        c->vq_enc.clear();
        for (int i = 0; i < cf.n_codebooks; ++i) {
            const std::string p = "quantizer.quantizers." + std::to_string(i);
            mis_snac::VqEnc v;
            v.in_proj = push_pw(p + ".in_proj", CD, D, true);
            const HostTensor& cb = need(c, p + ".codebook.weight", {CB, CD});
            std::vector<float> cn((size_t)CB * CD), cn2(CB);
            for (int64_t k = 0; k < CB; ++k) {
                float n2 = 0.0f;
                for (int64_t d = 0; d < CD; ++d) n2 += cb.v[k * CD + d] * cb.v[k * CD + d];
                const float inv = 1.0f / std::max(sqrtf(n2), 1e-12f);
                float s2 = 0.0f;
                for (int64_t d = 0; d < CD; ++d) { float x = cb.v[k * CD + d] * inv; cn[k * CD + d] = x; s2 += x * x; }
                cn2[k] = s2;
            }
            v.cn = push(cn); v.cn2 = push(cn2);
            c->vq_enc.push_back(v);
        }
        c->has_encoder = true;
    }
    c->arena.alloc(arena.size());
    HIP_CHECK(hipMemcpy(c->arena.p, arena.data(), arena.size() * sizeof(float), hipMemcpyHostToDevice));
    c->raw.clear();
    c->finalized = true;
    MIS_API_END
}

// ---------------------------------------------------------------------------- encode (SNACDecoder.swift:86-125)
extern "C" int64_t mis_snac_padded_length(const mis_snac* c, int64_t n_samples) {
    if (!c || n_samples < 0) return 0;
    int64_t l = 1;
    auto lcm = [](int64_t a, int64_t b) { int64_t x = a, y = b; while (y) { int64_t t = x % y; x = y; y = t; } return a / x * b; };
    for (int i = 0; i < c->cfg.n_codebooks; ++i) l = lcm(l, c->cfg.vq_strides[i]);
    if (c->cfg.attn_window_size > 0) l = lcm(l, c->cfg.attn_window_size);
    int64_t hop = 1;
    for (int i = 0; i < c->cfg.n_decoder_rates; ++i) hop *= c->cfg.decoder_rates[i];      // == prod(encoder_rates) for every published config
    if (c->has_encoder) { hop = 1; for (auto& b : c->enc_blocks) hop *= b.stride; }
    const int64_t pad_to = hop * l;
    return (n_samples + pad_to - 1) / pad_to * pad_to;
}

// audio f32 [batch, n_samples] (host or device) -> codes_out[i] int32 [batch, T_i], T_i = padded / hop / vq_strides[i];
// z_out (optional) f32 [batch, latent, padded / hop] = encoder output before quantisation (parity taps)
extern "C" mis_status mis_snac_encode(mis_snac* c, const float* audio, int batch, int64_t n_samples, int32_t* const* codes_out, float* z_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && audio && codes_out && batch >= 1 && n_samples >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "SNAC model not finalized");
    MIS_REQUIRE(c->has_encoder, MIS_ERR_AUDIO_ENCODE, "this SNAC handle was loaded without encoder weights");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const mis_snac_config& cf = c->cfg;
    const float* W = c->arena.p;
    const int64_t Tp = mis_snac_padded_length(c, n_samples);
    const int D = cf.latent_dim;
    size_t cap = (size_t)c->enc_dim * Tp;
    {
        int64_t T = Tp;
        for (auto& b : c->enc_blocks) { cap = std::max(cap, (size_t)b.cin * T); T /= b.stride; cap = std::max(cap, (size_t)b.cout * T); }
    }
    if (c->enc_attn.on) cap = std::max(cap, (size_t)3 * D * (Tp / std::max<int64_t>(1, mis_snac_hop_for_encoder(c))));
    cap *= (size_t)batch;
    for (int i = 0; i < 3; ++i) c->enc_buf[i].alloc(cap);
    DevBuf<float> ain;
    ain.alloc((size_t)batch * Tp);
    HIP_CHECK(hipMemsetAsync(ain.p, 0, (size_t)batch * Tp * 4, s));
    HIP_CHECK(hipMemcpy2DAsync(ain.p, (size_t)Tp * 4, audio, (size_t)n_samples * 4, (size_t)n_samples * 4, batch, hipMemcpyDefault, s));
    float *x = c->enc_buf[0].p, *f1 = c->enc_buf[1].p, *f2 = c->enc_buf[2].p;
    int64_t T = Tp;
    hipLaunchKernelGGL(k_enc_first, dim3(cdiv(T, 256), c->enc_dim, batch), dim3(256), 0, s, ain.p, x, W + c->enc_first.w, W + c->enc_first.b, c->enc_dim, (int)T);
    const int dils[3] = {1, 3, 9};
    DevBuf<float> zeros;
    for (auto& eb : c->enc_blocks) {
        for (int j = 0; j < 3; ++j) {
            const auto& ru = eb.ru[j];
            hipLaunchKernelGGL((k_snac_dw<true, true>), dim3(cdiv(T, DW_TILE), eb.cin, batch), dim3(256), 0, s, x, f1, W + ru.dw.w, W + ru.dw.b,
                               W + ru.s1.a, W + ru.s1.ra, W + ru.s2.a, W + ru.s2.ra, eb.cin, (int)T, dils[j]);
            GemmParams g{};
            g.AT = W + ru.pw.w; g.bias = W + ru.pw.b; g.X = f1; g.Y = f2; g.R = x;
            g.M = eb.cin; g.K = eb.cin; g.N = (int)T; g.Tin = (int)T; g.Tout = (int)T;
            launch_gemm(GEMM_RESID, false, g, batch, s);
            std::swap(x, f2);
        }
        MIS_REQUIRE(T % eb.stride == 0, MIS_ERR_AUDIO_ENCODE, "internal: length not divisible by the stride");
        hipLaunchKernelGGL(k_enc_phase_split, dim3(cdiv(T, 256), eb.cin, batch), dim3(256), 0, s, x, f1, W + eb.snake.a, W + eb.snake.ra, eb.cin, (int)T, eb.stride);
        T /= eb.stride;
        GemmParams g{};
        g.AT = W + eb.down.w; g.bias = W + eb.down.b; g.X = f1; g.Y = f2;
        g.M = eb.cout; g.K = 3 * eb.stride * eb.cin; g.N = (int)T; g.Tin = (int)T; g.Tout = (int)T;
        g.Cin = eb.stride * eb.cin; g.taps = 3; g.dil = 1; g.pad = 1;
        launch_gemm(GEMM_TAPS, false, g, batch, s);
        std::swap(x, f2);
    }
    if (c->enc_attn.on) {                                                 // Layers.swift:339-341
        snac_local_mha(c->enc_attn, W, cf.attn_window_size, x, /*y*/ f2, /*t1*/ f1, /*t2*/ f2, batch, T, s);
        std::swap(x, f2);
    }
    hipLaunchKernelGGL((k_snac_dw<false, false>), dim3(cdiv(T, DW_TILE), D, batch), dim3(256), 0, s, x, f1, W + c->enc_last.w, W + c->enc_last.b,
                       nullptr, nullptr, nullptr, nullptr, D, (int)T, 1);
    float* resid = f1;                                                    // z [B][D][T]
    if (z_out) HIP_CHECK(hipMemcpyAsync(z_out, resid, (size_t)batch * D * T * 4, hipMemcpyDefault, s));
    // ---- residual VQ (VQ.swift:141-161)
    c->vq_pool.alloc((size_t)batch * D * T);
    c->vq_ze.alloc((size_t)batch * cf.codebook_dim * T);
    c->enc_codes.alloc((size_t)batch * T);
    for (int i = 0; i < cf.n_codebooks; ++i) {
        const int st = cf.vq_strides[i];
        MIS_REQUIRE(T % st == 0, MIS_ERR_AUDIO_ENCODE, "internal: latent length not divisible by vq stride");
        const int Tm = (int)(T / st);
        const float* pooled = resid;
        if (st > 1) {
            hipLaunchKernelGGL(k_vq_pool, dim3(cdiv(Tm, 128), D, batch), dim3(128), 0, s, resid, c->vq_pool.p, D, (int)T, st);
            pooled = c->vq_pool.p;
        }
        GemmParams g{};
        g.AT = W + c->vq_enc[i].in_proj.w; g.bias = W + c->vq_enc[i].in_proj.b; g.X = pooled; g.Y = c->vq_ze.p;
        g.M = cf.codebook_dim; g.K = D; g.N = Tm; g.Tin = Tm; g.Tout = Tm;
        launch_gemm(GEMM_PLAIN, false, g, batch, s);
        MIS_REQUIRE(cf.codebook_dim <= 64, MIS_ERR_INVALID_INPUT, "codebook_dim > 64 unsupported by the nearest-code kernel");
        hipLaunchKernelGGL(k_vq_nearest, dim3(Tm, batch), dim3(256), 0, s, c->vq_ze.p, W + c->vq_enc[i].cn, W + c->vq_enc[i].cn2, c->enc_codes.p,
                           cf.codebook_dim, cf.codebook_size, Tm);
        HIP_CHECK(hipMemcpyAsync(codes_out[i], c->enc_codes.p, (size_t)batch * Tm * 4, hipMemcpyDefault, s));
        if (i + 1 < cf.n_codebooks)
            hipLaunchKernelGGL(k_vq_residual, dim3(cdiv(T, 256), D, batch), dim3(256), 0, s, resid, c->enc_codes.p,
                               W + c->tables_off + (size_t)i * cf.codebook_size * D, D, (int)T, st);
        HIP_CHECK(hipStreamSynchronize(s));                                // enc_codes is reused by the next level
    }
    HIP_CHECK(hipGetLastError());
    MIS_API_END
}

extern "C" int64_t mis_snac_num_samples(const mis_snac* c, int t_coarse) {
    if (!c || t_coarse <= 0) return 0;
    int64_t T = (int64_t)t_coarse * c->cfg.vq_strides[0];
    for (int i = 0; i < c->cfg.n_decoder_rates; ++i) T = convt_out_len(T, c->cfg.decoder_rates[i]);
    return T;
}
extern "C" int64_t mis_snac_noise_len(const mis_snac* c, int block, int t_coarse) {
    if (!c || t_coarse <= 0 || block < 0 || block >= c->cfg.n_decoder_rates) return 0;
    int64_t T = (int64_t)t_coarse * c->cfg.vq_strides[0];
    for (int i = 0; i <= block; ++i) T = convt_out_len(T, c->cfg.decoder_rates[i]);
    return T;
}

// ---------------------------------------------------------------------------- pipeline
void launch_dw7(const float* X, float* Y, const float* w7, const float* bias, int batch, int C, int T, int dil, hipStream_t s) {
    hipLaunchKernelGGL((k_snac_dw<false, false>), dim3(cdiv(T, DW_TILE), C, batch), dim3(256), 0, s, X, Y, w7, bias, nullptr,
                       nullptr, nullptr, nullptr, C, T, dil);
}

static bool ru_fused_enabled() {                       // MIS_CODEC_FUSED_UNITS=0: the unfused kernels (A/B)
    const char* v = getenv("MIS_CODEC_FUSED_UNITS");
    return !v || atoi(v) != 0;
}

// encoder pieces shared with the Descript DAC encoder (dac.hip)
void launch_enc_first(const float* audio, float* y, const float* w, const float* bias, int batch, int C, int T, hipStream_t s) {
    hipLaunchKernelGGL(k_enc_first, dim3(cdiv(T, 256), C, batch), dim3(256), 0, s, audio, y, w, bias, C, T);
}
void launch_enc_phase_split(const float* x, float* y, const float* a, const float* ra, int batch, int C, int T, int stride, hipStream_t s) {
    hipLaunchKernelGGL(k_enc_phase_split, dim3(cdiv(T, 256), C, batch), dim3(256), 0, s, x, y, a, ra, C, T, stride);
}
void launch_vq_nearest(const float* ze, const float* cn, const float* cn2, int32_t* codes, int batch, int CD, int CB, int Tm, hipStream_t s) {
    MIS_REQUIRE(CD <= 64, MIS_ERR_INVALID_INPUT, "codebook_dim > 64 unsupported by the nearest-code kernel");
    hipLaunchKernelGGL(k_vq_nearest, dim3(Tm, batch), dim3(256), 0, s, ze, cn, cn2, codes, CD, CB, Tm);
}
void launch_vq_residual(float* r, const int32_t* codes, const float* table, int batch, int C, int T, int stride, hipStream_t s) {
    hipLaunchKernelGGL(k_vq_residual, dim3(cdiv(T, 256), C, batch), dim3(256), 0, s, r, codes, table, C, T, stride);
}

void launch_gemm(int mode, bool snake, const GemmParams& p_in, int batch, hipStream_t s) {
    GemmParams p = p_in;
    if (!p.ldx) p.ldx = p.Tin;                                          // dense [B][C][T] tensors unless the caller says otherwise
    if (!p.ldy) p.ldy = p.Tout;
    MIS_REQUIRE(p.x_lo <= 0 && p.ldx >= p.Tin, MIS_ERR_GENERATION_FAILED, "bad codec GEMM strides");
    if (mode == GEMM_RESID && snake && p.alpha && p.R && !p.scale && p.bias && p.M == p.K && ru_fused_enabled() &&
        (p.M == 64 || p.M == 96 || p.M == 128 || p.M == 192)) {           // narrow unit tails: whole channel range per block (k_pw_fused)
        PwFusedParams fp{};
        fp.X = p.X; fp.R = p.R; fp.Y = p.Y; fp.a2 = p.alpha; fp.ra2 = p.ralpha; fp.AT = p.AT; fp.b2 = p.bias;
        fp.T = p.N; fp.dil = 0; fp.ldx = p.ldx; fp.ldr = p.ldy; fp.ldy = p.ldy;
        dim3 fg(cdiv(p.N, RU_NT), batch);
        if (p.M == 64) hipLaunchKernelGGL((k_pw_fused<64, false>), fg, dim3(256), 0, s, fp);
        else if (p.M == 96) hipLaunchKernelGGL((k_pw_fused<96, false>), fg, dim3(256), 0, s, fp);
        else if (p.M == 128) hipLaunchKernelGGL((k_pw_fused<128, false>), fg, dim3(256), 0, s, fp);
        else hipLaunchKernelGGL((k_pw_fused<192, false>), fg, dim3(256), 0, s, fp);
        return;
    }
    if (launch_gemm_bf3(mode, snake, p, batch, s)) return;
    int phases = (mode == GEMM_CONVT) ? p.s : 1;
    dim3 grid(cdiv(p.N, G_BN), cdiv(p.M, G_BM), batch * phases), block(256);
    if (mode == GEMM_PLAIN) hipLaunchKernelGGL((k_snac_gemm<GEMM_PLAIN, false>), grid, block, 0, s, p);
    else if (mode == GEMM_GELU) hipLaunchKernelGGL((k_snac_gemm<GEMM_GELU, false>), grid, block, 0, s, p);
    else if (mode == GEMM_RESID && snake) hipLaunchKernelGGL((k_snac_gemm<GEMM_RESID, true>), grid, block, 0, s, p);
    else if (mode == GEMM_RESID) hipLaunchKernelGGL((k_snac_gemm<GEMM_RESID, false>), grid, block, 0, s, p);
    else if (mode == GEMM_TAPS) {
        MIS_REQUIRE(p.taps >= 1 && p.taps <= CT_MAXT && (p.taps - 1) * p.dil + 3 + G_BN <= CT_XS, MIS_ERR_INVALID_INPUT,
                    "dense conv: %d taps with dilation %d exceed the staged halo", p.taps, p.dil);
        GemmParams q = p;
        if (!snake) { q.alpha = nullptr; q.ralpha = nullptr; }
        if (q.R) hipLaunchKernelGGL((k_conv_taps<true>), grid, block, 0, s, q);
        else hipLaunchKernelGGL((k_conv_taps<false>), grid, block, 0, s, q);
    }
    else if (mode == GEMM_NOISE) hipLaunchKernelGGL((k_snac_gemm<GEMM_NOISE, false>), grid, block, 0, s, p);
    else { MIS_REQUIRE(snake, MIS_ERR_GENERATION_FAILED, "convT without snake"); hipLaunchKernelGGL((k_snac_gemm<GEMM_CONVT, true>), grid, block, 0, s, p); }
}

static int64_t mis_snac_hop_for_encoder(const mis_snac* c) {
    int64_t h = 1;
    for (auto& b : c->enc_blocks) h *= b.stride;
    return h;
}

// LocalMHA on x [B][C][T] (Attention.swift:31-63): LayerNorm -> to_qkv -> windowed attention -> to_out + residual.
// t1 (>= C*T per row) and t2 (>= 3*C*T per row) are work buffers; the result lands in y.
static void snac_local_mha(const mis_snac::MhaW& m, const float* W, int win, const float* x, float* y, float* t1, float* t2, int batch, int64_t T,
                           hipStream_t s) {
    MIS_REQUIRE(T % win == 0, MIS_ERR_INVALID_INPUT, "LocalMHA needs a whole number of %d-frame windows (got %lld frames)", win, (long long)T);
    const int C = m.dim;
    hipLaunchKernelGGL(k_snac_ln_ct, dim3(cdiv(T, 128), batch), dim3(128), 0, s, x, t1, W + m.ln_w, W + m.ln_b, C, (int)T, 1e-5f);
    GemmParams g{};
    g.AT = W + m.qkv; g.X = t1; g.Y = t2; g.M = 3 * C; g.K = C; g.N = (int)T; g.Tin = (int)T; g.Tout = (int)T;
    launch_gemm(GEMM_PLAIN, false, g, batch, s);
    hipLaunchKernelGGL(k_snac_local_attn, dim3((unsigned)(T / win), C / MHA_D, batch), dim3(256), 0, s, t2, t1, W + m.inv_freq, C, (int)T, win);
    GemmParams o{};
    o.AT = W + m.out; o.X = t1; o.Y = y; o.R = x; o.M = C; o.K = C; o.N = (int)T; o.Tin = (int)T; o.Tout = (int)T;
    launch_gemm(GEMM_RESID, false, o, batch, s);
}

// Runs the decode on device pointers.  stop_after: -1 = full; 0 zq, 1 stem_dw, 2 stem_pw, 3+i block i.
// Returns the buffer holding the stage output (for taps) and its [C, T].
struct NoiseRng { int enabled = 0; uint64_t seed = 0; const int32_t* row_ids = nullptr; int64_t row_offset = 0; };

static const float* snac_run(mis_snac* c, const int32_t* const* codes, int batch, int t_coarse,
                             const float* const* noise, const NoiseRng& rng, float* pcm, int64_t pcm_stride,
                             int stop_after, int* outC, int64_t* outT, hipStream_t s) {
    CodecPackScope pack_scope(&c->pack);
    const mis_snac_config& cf = c->cfg;
    const float* W = c->arena.p;
    int64_t T0 = (int64_t)t_coarse * cf.vq_strides[0];
    int64_t Tfinal = mis_snac_num_samples(c, t_coarse);
    // capacity: max over stages of C*T
    size_t cap = (size_t)std::max<int64_t>(cf.latent_dim, cf.decoder_dim) * T0;
    {
        int64_t T = T0;
        for (int i = 0; i < cf.n_decoder_rates; ++i) {
            T = convt_out_len(T, cf.decoder_rates[i]);
            cap = std::max(cap, (size_t)(cf.decoder_dim >> (i + 1)) * (size_t)T);
        }
    }
    if (c->dec_attn.on) cap = std::max(cap, (size_t)3 * cf.decoder_dim * T0);          // q | k | v
    cap *= (size_t)batch;
    for (int i = 0; i < 3; ++i) c->buf[i].alloc(cap);
    float *A = c->buf[0].p, *B = c->buf[1].p, *Cc = c->buf[2].p;

    // fromCodes
    EmbedParams ep{};
    ep.tables = W + c->tables_off;
    for (int i = 0; i < cf.n_codebooks; ++i) { ep.codes[i] = codes[i]; ep.strides[i] = cf.vq_strides[i]; }
    ep.n_q = cf.n_codebooks; ep.codebook_size = cf.codebook_size; ep.C = cf.latent_dim; ep.T0 = (int)T0;
    hipLaunchKernelGGL(k_snac_embed, dim3(cdiv(T0, 32), cdiv(cf.latent_dim, 32), batch), dim3(256), 0, s, ep, A);
    if (stop_after == 0) { *outC = cf.latent_dim; *outT = T0; return A; }
    // stem: depthwise k7 then 1x1 (Layers.swift:378-388)
    hipLaunchKernelGGL((k_snac_dw<false, false>), dim3(cdiv(T0, DW_TILE), cf.latent_dim, batch), dim3(256), 0, s,
                       A, B, W + c->stem_dw.w, W + c->stem_dw.b, nullptr, nullptr, nullptr, nullptr, cf.latent_dim, (int)T0, 1);
    if (stop_after == 1) { *outC = cf.latent_dim; *outT = T0; return B; }
    {
        GemmParams g{};
        g.AT = W + c->stem_pw.w; g.bias = W + c->stem_pw.b; g.X = B; g.Y = A;
        g.M = cf.decoder_dim; g.K = cf.latent_dim; g.N = (int)T0; g.Tin = (int)T0; g.Tout = (int)T0;
        launch_gemm(GEMM_PLAIN, false, g, batch, s);
    }
    if (stop_after == 2) { *outC = cf.decoder_dim; *outT = T0; return A; }
    float* x = A;        // current activation
    float* f1 = B;
    float* f2 = Cc;
    if (c->dec_attn.on) {                                                              // Layers.swift:395-397
        CodecPackScope exact(nullptr);                // exact-f32 contractions here: the softmax amplifies operand noise
        snac_local_mha(c->dec_attn, W, cf.attn_window_size, A, /*y*/ Cc, /*t1*/ B, /*t2*/ Cc, batch, T0, s);
        x = Cc; f1 = A; f2 = B;
    }
    int64_t T = T0;
    for (size_t bi = 0; bi < c->blocks.size(); ++bi) {
        const mis_snac::Block& blk = c->blocks[bi];
        int64_t To = convt_out_len(T, blk.stride);
        {   // Snake -> transposed conv (Layers.swift:288-296)
            GemmParams g{};
            g.AT = W + blk.convT.w; g.bias = W + blk.convT.b; g.X = x; g.Y = f1;
            g.alpha = W + blk.snake0.a; g.ralpha = W + blk.snake0.ra;
            g.M = blk.cout; g.K = 2 * blk.cin; g.Tin = (int)T; g.Tout = (int)To;
            g.s = blk.stride; g.pad = blk.pad; g.Cin = blk.cin;
            g.N = (int)((To + blk.stride - 1) / blk.stride);     // columns per phase; stores are bounds-checked
            launch_gemm(GEMM_CONVT, true, g, batch, s);
        }
        std::swap(x, f1);
        T = To;
        if (cf.noise) {   // NoiseBlock (Layers.swift:270-278)
            GemmParams g{};
            g.AT = W + blk.noise.w; g.bias = nullptr; g.X = x; g.Y = f1;
            g.noise = noise ? noise[bi] : nullptr;
            g.noise_rng = (!noise && rng.enabled) ? 1 : 0;
            g.noise_key = mis_splitmix64(rng.seed + 0x51AC0000ull + (uint64_t)bi);
            g.row_ids = rng.row_ids; g.row_offset = rng.row_offset;
            g.M = blk.cout; g.K = blk.cout; g.N = (int)T; g.Tin = (int)T; g.Tout = (int)T;
            launch_gemm(GEMM_NOISE, false, g, batch, s);
            std::swap(x, f1);
        }
        const int dils[3] = {1, 3, 9};
        for (int j = 0; j < 3; ++j) {   // ResidualUnit (Layers.swift:202-231)
            const auto& ru = blk.ru[j];
            if ((blk.cout == 64 || blk.cout == 128) && ru_fused_enabled()) {           // HBM-bound late blocks: one pass instead of five
                PwFusedParams fp{};
                fp.X = x; fp.R = x; fp.Y = f2; fp.w7 = W + ru.dw.w; fp.b1 = W + ru.dw.b; fp.a1 = W + ru.s1.a; fp.ra1 = W + ru.s1.ra;
                fp.a2 = W + ru.s2.a; fp.ra2 = W + ru.s2.ra; fp.AT = W + ru.pw.w; fp.b2 = W + ru.pw.b;
                fp.T = (int)T; fp.dil = dils[j]; fp.ldx = fp.ldr = fp.ldy = (int)T;
                dim3 fg(cdiv(T, RU_NT), batch);
                if (blk.cout == 128) hipLaunchKernelGGL((k_pw_fused<128, true>), fg, dim3(256), 0, s, fp);
                else hipLaunchKernelGGL((k_pw_fused<64, true>), fg, dim3(256), 0, s, fp);
                std::swap(x, f2);
                continue;
            }
            hipLaunchKernelGGL((k_snac_dw<true, true>), dim3(cdiv(T, DW_TILE), blk.cout, batch), dim3(256), 0, s,
                               x, f1, W + ru.dw.w, W + ru.dw.b, W + ru.s1.a, W + ru.s1.ra, W + ru.s2.a, W + ru.s2.ra,
                               blk.cout, (int)T, dils[j]);
            GemmParams g{};
            g.AT = W + ru.pw.w; g.bias = W + ru.pw.b; g.X = f1; g.Y = f2; g.R = x;
            g.M = blk.cout; g.K = blk.cout; g.N = (int)T; g.Tin = (int)T; g.Tout = (int)T;
            launch_gemm(GEMM_RESID, false, g, batch, s);
            std::swap(x, f2);
        }
        if (stop_after == 3 + (int)bi) { *outC = blk.cout; *outT = T; return x; }
    }
    MIS_REQUIRE(T == Tfinal, MIS_ERR_AUDIO_DECODE, "internal length mismatch");
    int cl = cf.decoder_dim >> cf.n_decoder_rates;
    hipLaunchKernelGGL(k_snac_final, dim3(cdiv(T, FIN_TILE), batch), dim3(256), 0, s, x, pcm, pcm_stride,
                       W + c->fin.w, c->fin_bias, W + c->fin_snake.a, W + c->fin_snake.ra, cl, (int)T);
    *outC = 1; *outT = T;
    return pcm;
}

void snac_decode_device(mis_snac* c, const int32_t* const* codes_dev, int batch, int t_coarse,
                        const float* const* noise_dev, int rng_enabled, uint64_t rng_seed, const int32_t* row_ids,
                        int64_t row_offset, float* pcm_dev, int64_t pcm_stride, hipStream_t s) {
    MIS_REQUIRE(c && c->finalized, MIS_ERR_NOT_INITIALIZED, "SNAC codec not finalized");
    int C; int64_t T;
    NoiseRng rng;
    rng.enabled = rng_enabled; rng.seed = rng_seed; rng.row_ids = row_ids; rng.row_offset = row_offset;
    snac_run(c, codes_dev, batch, t_coarse, noise_dev, rng, pcm_dev, pcm_stride, -1, &C, &T, s);
    HIP_CHECK(hipGetLastError());
}

static void snac_stage_inputs(mis_snac* c, const int32_t* const* codes, int batch, int t_coarse,
                              const float* const* noise, std::vector<const int32_t*>& dcodes,
                              std::vector<const float*>& dnoise) {
    const mis_snac_config& cf = c->cfg;
    size_t total = 0;
    std::vector<size_t> offs(cf.n_codebooks);
    for (int i = 0; i < cf.n_codebooks; ++i) {
        offs[i] = total;
        total += (size_t)batch * t_coarse * (cf.vq_strides[0] / cf.vq_strides[i]);
    }
    c->codes_ws.alloc(total);
    dcodes.resize(cf.n_codebooks);
    for (int i = 0; i < cf.n_codebooks; ++i) {
        size_t n = (size_t)batch * t_coarse * (cf.vq_strides[0] / cf.vq_strides[i]);
        MIS_REQUIRE(codes[i], MIS_ERR_INVALID_INPUT, "codes[%d] is null", i);
        HIP_CHECK(hipMemcpyAsync(c->codes_ws.p + offs[i], codes[i], n * sizeof(int32_t), hipMemcpyDefault, c->stream));
        dcodes[i] = c->codes_ws.p + offs[i];
    }
    dnoise.clear();
    if (noise && cf.noise) {
        size_t tot = 0;
        std::vector<size_t> no(cf.n_decoder_rates);
        for (int i = 0; i < cf.n_decoder_rates; ++i) { no[i] = tot; tot += (size_t)batch * mis_snac_noise_len(c, i, t_coarse); }
        c->noise_ws.alloc(tot);
        dnoise.resize(cf.n_decoder_rates);
        for (int i = 0; i < cf.n_decoder_rates; ++i) {
            MIS_REQUIRE(noise[i], MIS_ERR_INVALID_INPUT, "noise[%d] is null", i);
            size_t n = (size_t)batch * mis_snac_noise_len(c, i, t_coarse);
            HIP_CHECK(hipMemcpyAsync(c->noise_ws.p + no[i], noise[i], n * sizeof(float), hipMemcpyDefault, c->stream));
            dnoise[i] = c->noise_ws.p + no[i];
        }
    }
}

extern "C" mis_status mis_snac_decode(mis_snac* c, const int32_t* const* codes, int batch, int t_coarse,
                                      const float* const* noise, float* pcm_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c, MIS_ERR_INVALID_INPUT, "null handle");
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "SNAC codec not finalized");
    MIS_REQUIRE(batch >= 0 && t_coarse >= 0, MIS_ERR_INVALID_INPUT, "negative batch / length");
    if (batch == 0 || t_coarse == 0) return MIS_OK;       // empty input -> empty waveform
    MIS_REQUIRE(codes && pcm_out, MIS_ERR_INVALID_INPUT, "null pointer");
    HIP_CHECK(hipSetDevice(c->device));
    std::vector<const int32_t*> dcodes;
    std::vector<const float*> dnoise;
    snac_stage_inputs(c, codes, batch, t_coarse, noise, dcodes, dnoise);
    int64_t N = mis_snac_num_samples(c, t_coarse);
    DevBuf<float> pcm;
    pcm.alloc((size_t)batch * N);
    int C; int64_t T;
    NoiseRng rng;
    rng.enabled = c->null_noise_zero ? 0 : 1; rng.seed = c->noise_seed;
    snac_run(c, dcodes.data(), batch, t_coarse, dnoise.empty() ? nullptr : dnoise.data(), rng, pcm.p, N, -1, &C, &T, c->stream);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(pcm_out, pcm.p, (size_t)batch * N * sizeof(float), hipMemcpyDefault, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->last_batch = batch; c->last_t = t_coarse;
    c->last_codes = dcodes; c->last_noise_ptrs = dnoise; c->last_noise = !dnoise.empty();
    MIS_API_END
}

extern "C" mis_status mis_snac_set_noise(mis_snac* c, int null_noise_is_zero, uint64_t seed) {
    MIS_API_BEGIN
    MIS_REQUIRE(c, MIS_ERR_INVALID_INPUT, "null handle");
    c->null_noise_zero = null_noise_is_zero ? 1 : 0;
    c->noise_seed = seed;
    MIS_API_END
}

extern "C" mis_status mis_snac_debug_tap(mis_snac* c, const char* name, float* out, int64_t capacity,
                                         int32_t* channels, int64_t* length) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && name && out && channels && length, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(c->finalized && c->last_batch > 0, MIS_ERR_NOT_INITIALIZED, "no previous decode to tap");
    HIP_CHECK(hipSetDevice(c->device));
    std::string n = name;
    int stop = -2;
    if (n == "zq") stop = 0;
    else if (n == "stem_dw") stop = 1;
    else if (n == "stem_pw") stop = 2;
    else if (n.rfind("block", 0) == 0) stop = 3 + atoi(n.c_str() + 5);
    MIS_REQUIRE(stop >= 0 && stop < 3 + c->cfg.n_decoder_rates, MIS_ERR_INVALID_INPUT, "unknown tap %s", name);
    int C; int64_t T;
    NoiseRng rng;
    rng.enabled = c->null_noise_zero ? 0 : 1; rng.seed = c->noise_seed;
    const float* src = snac_run(c, c->last_codes.data(), c->last_batch, c->last_t,
                                c->last_noise ? c->last_noise_ptrs.data() : nullptr, rng, nullptr, 0, stop, &C, &T, c->stream);
    HIP_CHECK(hipGetLastError());
    size_t n_el = (size_t)c->last_batch * C * T;
    MIS_REQUIRE((int64_t)n_el <= capacity, MIS_ERR_INVALID_INPUT, "tap buffer too small (%zu needed)", n_el);
    HIP_CHECK(hipMemcpyAsync(out, src, n_el * sizeof(float), hipMemcpyDefault, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    *channels = C; *length = T;
    MIS_API_END
}

// SNAC.fromModelDirectory (SNACDecoder.swift:156-189)
extern "C" mis_status mis_snac_load(const char* model_dir, int device, mis_snac** out) {
    mis_snac* c = nullptr;
    try {
        MIS_REQUIRE(model_dir && out, MIS_ERR_INVALID_INPUT, "null argument");
        std::string dir = model_dir;
        JsonValue j = json_parse(read_text_file(dir + "/config.json"));
        mis_snac_config cf{};
        cf.sampling_rate = (int)j.number_or("sampling_rate", 24000);
        const JsonValue* er = j.get("encoder_rates");
        int enc_dim = (int)j.number_or("encoder_dim", 64);
        const JsonValue* ld = j.get("latent_dim");
        MIS_REQUIRE(er && er->type == JsonValue::ARR, MIS_ERR_INVALID_INPUT, "config.json: encoder_rates missing");
        cf.latent_dim = (ld && ld->type == JsonValue::NUM) ? (int)ld->num : enc_dim << er->arr.size();
        cf.decoder_dim = (int)j.number_or("decoder_dim", 1536);
        const JsonValue* dr = j.get("decoder_rates");
        MIS_REQUIRE(dr && dr->type == JsonValue::ARR && dr->arr.size() <= 8, MIS_ERR_INVALID_INPUT, "config.json: decoder_rates");
        cf.n_decoder_rates = (int)dr->arr.size();
        for (size_t i = 0; i < dr->arr.size(); ++i) cf.decoder_rates[i] = (int)dr->arr[i].num;
        cf.codebook_size = (int)j.number_or("codebook_size", 4096);
        cf.codebook_dim = (int)j.number_or("codebook_dim", 8);
        const JsonValue* vs = j.get("vq_strides");
        MIS_REQUIRE(vs && vs->type == JsonValue::ARR && vs->arr.size() <= 8, MIS_ERR_INVALID_INPUT, "config.json: vq_strides");
        cf.n_codebooks = (int)vs->arr.size();
        for (size_t i = 0; i < vs->arr.size(); ++i) cf.vq_strides[i] = (int)vs->arr[i].num;
        cf.noise = j.bool_or("noise", true) ? 1 : 0;
        cf.depthwise = j.bool_or("depthwise", true) ? 1 : 0;
        const JsonValue* aw = j.get("attn_window_size");
        cf.attn_window_size = (aw && aw->type == JsonValue::NUM) ? (int)aw->num : 0;
        mis_status st = mis_snac_create(&cf, device, &c);
        if (st != MIS_OK) return st;
        SafeTensorFile f;
        f.open(dir + "/model.safetensors");
        for (auto& e : f.entries) {
            if (e.name.find(".in_proj.") != std::string::npos) continue;          // encode path
            st = mis_snac_set_tensor(c, e.name.c_str(), e.data, dtype_from_safetensors(e.dtype), e.shape.data(), (int)e.shape.size());
            if (st != MIS_OK) { mis_snac_destroy(c); return st; }
        }
        st = mis_snac_finalize(c);
        if (st != MIS_OK) { mis_snac_destroy(c); return st; }
        *out = c;
        return MIS_OK;
    } catch (const MisError& e) {
        if (c) mis_snac_destroy(c);
        mis_set_error("%s", e.what());
        return e.code;
    } catch (const std::exception& e) {
        if (c) mis_snac_destroy(c);
        mis_set_error("%s", e.what());
        return MIS_ERR_GENERATION_FAILED;
    }
}
