// codec_kernels.h - f32 kernels shared by the codec back ends (SNAC decoder, Soprano / Vocos decoder); defined in snac.hip
#pragma once
#include "common.h"
#include <vector>

enum { GEMM_PLAIN = 0, GEMM_RESID = 1, GEMM_NOISE = 2, GEMM_CONVT = 3, GEMM_GELU = 4, GEMM_TAPS = 5 };

// Y[b][m][n] = sum_k A[m][k] X[b][k][n] on v_mfma_f32_32x32x2_f32 (activations NCT, time contiguous)
// sin^2(y) for the Snake activations of the streaming-rate kernels: y is reduced by multiples of pi (two-term Cody-Waite, exact for
// |y| < 1e5) to |r| <= pi/2, sin r by the degree-11 odd polynomial (truncation 5.7e-8 at the interval end), and the sign drops out
// in the square.  12 VALU instructions instead of the ~30 of the full-range sinf: the split pre-pass and the final conv were bound by
// that (33 instructions per element measured from their element rate); absolute error <= 3.3e-7 for |y| <= 274 (CPU sweep).
__device__ __forceinline__ float mis_sin_sq(float y) {
    const float n = rintf(y * 0.318309886183790672f);
    float r = fmaf(n, -3.140625f, y);
    r = fmaf(n, -9.67653589793e-4f, r);
    const float s = r * r;
    float p = fmaf(s, -2.50521084e-8f, 2.75573192e-6f);
    p = fmaf(s, p, -1.98412698e-4f);
    p = fmaf(s, p, 8.33333333e-3f);
    p = fmaf(s, p, -1.66666667e-1f);
    p = fmaf(r * s, p, r);
    return p * p;
}

// Per-model state of the split-bf16 path (codec_bf3.hip): packed hi/lo weight fragments keyed by the A^T pointer (packed on first use,
// on the calling stream) and the split-activation scratch planes.  One in-flight call per model handle, as for the rest of the engine.
struct CodecPack {
    struct Entry { const float* at; int ntaps, Cin, M, Cp, KS, MT32; uint16_t* wp; };
    std::vector<Entry> entries;
    DevBuf<uint16_t> xh, xl;
    DevBuf<float> part;   // split-K slabs of launch-shaped contractions (GemmParams.split_k_ok)
    CodecPack() {}
    CodecPack(const CodecPack&) = delete;
    CodecPack& operator=(const CodecPack&) = delete;
    ~CodecPack();
    const Entry& get(const float* at, int ntaps, int Cin, int M, hipStream_t s);
};

// Decoders open a scope around their launches: launch_gemm picks the scope's CodecPack up when GemmParams.pack is null.  Encoders
// (codebook decisions) do not, and stay on the exact-f32 kernels.
struct CodecPackScope {
    CodecPack* prev;
    explicit CodecPackScope(CodecPack* p);
    ~CodecPackScope();
};

struct GemmParams {
    CodecPack* pack;      // non-null: MFMA-bound shapes run as split-bf16 contractions (codec_bf3.hip); null: exact-f32 kernels only
    const float* AT;      // [K][M]   (CONVT: [s][K][M], K = 2*Cin)
    const float* bias;    // [M] or null
    const float* X;       // [B][Kx][Tin]  (Kx = K; CONVT: Cin)
    float* Y;             // [B][M][Tout]
    const float* R;       // RESID: [B][M][N]
    const float* scale;   // RESID: optional per-row scale of (acc + bias)  (ConvNeXt gamma)
    const float* noise;   // NOISE: explicit [B][N], or null
    int noise_rng;        // NOISE with noise == null: 1 = draw N(0,1) from the documented generator, 0 = zeros
    uint64_t noise_key;   // (seed, block) key of the generator
    const int32_t* row_ids;   // optional global row id per batch row (rng keyed by GLOBAL row)
    int64_t row_offset;
    const float* alpha;   // Snake prologue on X rows (null = none)
    const float* ralpha;
    int M, K, N;          // N = output columns per phase
    int Tin, Tout;
    int s, pad, Cin;      // CONVT only (Cin also TAPS)
    int ldx, ldy;         // row strides of X and of Y / R in floats; 0 = dense (Tin, Tout)
    int x_lo;             // lowest valid X column (<= 0): columns [x_lo, 0) hold carried history (streaming decode), zero below
    int dup_bias_n0;      // CONVT: add the bias a second time to output frame n = 0 - the reference's streaming overlap-add sums
                          // two biased outputs there (DecoderBlockUpsample.step, Qwen3TTSSpeechTokenizer.swift:553-576)
    int split_k_ok;       // 1x1 modes on the split-bf16 path: the launch may split its K range over blocks when one batch row gives too few
                          // blocks to fill the chip (the factor depends on the SHAPE only, never on the batch: a row's result is the same
                          // whatever batch it shares); float32 slabs summed in slice order by a second launch that applies the epilogue
    int taps, dil;        // TAPS: dense conv, K = taps*Cin, tap j reads x[:, n - pad + j*dil] (zero outside); A^T row j*Cin + c;
                          // pad = (taps-1)*dil: causal, (taps-1)*dil/2: "same"; optional R (+scale) residual epilogue
};
// snake: Snake prologue on the X rows (CONVT always runs it: pass alpha = ralpha = zeros for identity)
void launch_gemm(int mode, bool snake, const GemmParams& p, int batch, hipStream_t s);
// split-bf16 path; false = shape not eligible (or MIS_CODEC_EXACT_F32=1), nothing launched.  p must have ldx / ldy resolved
bool launch_gemm_bf3(int mode, bool snake, const GemmParams& p, int batch, hipStream_t s);
// depthwise 7-tap conv (zero padded, dilation dil): shorter odd kernels ride in centred 7-tap weights
void launch_dw7(const float* X, float* Y, const float* w7 /*[C][7]*/, const float* bias, int batch, int C, int T, int dil, hipStream_t s);
// encoder pieces (SNAC / DAC): first conv k7 "same" 1 -> C (w [C][7]); Snake + phase split for a stride-s conv
// (y[(c*s + r)][m] = snake(x[c][s*m + r])); nearest code of the L2-normalised latent (first index on ties; cn = normalised codebook
// [CB][CD], cn2 = its squared norms); residual -= table[code] (table [CB][C], code index t / stride)
void launch_enc_first(const float* audio, float* y, const float* w, const float* bias, int batch, int C, int T, hipStream_t s);
void launch_enc_phase_split(const float* x, float* y, const float* a, const float* ra, int batch, int C, int T, int stride, hipStream_t s);
void launch_vq_nearest(const float* ze, const float* cn, const float* cn2, int32_t* codes, int batch, int CD, int CB, int Tm, hipStream_t s);
void launch_vq_residual(float* r, const int32_t* codes, const float* table, int batch, int C, int T, int stride, hipStream_t s);
