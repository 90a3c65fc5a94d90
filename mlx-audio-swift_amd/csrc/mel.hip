// mel.hip - log-mel / STFT front end for gfx950.
//
// Replaces WhisperAudio.logMelSpectrogram / encoderFeatures (Sources/MLXAudioSTT/Models/Whisper/
// WhisperAudio.swift:38-87) and the generic computeMelSpectrogram (Sources/MLXAudioCore/DSP.swift:230-273),
// whose arithmetic is MLX: reflect pad -> asStrided frames -> window -> MLXFFT.rfft -> |.|^2 -> mel
// filterbank matmul -> log10 -> clamp to (max - 8) -> (x + 4) / 4.
//
// One fused kernel per 32-frame tile: the windowed frames are built once in LDS (reflect padding resolved
// while loading), the length-n_fft real DFT is evaluated as an exact-f32 MFMA contraction against
// host-built cos/sin tables (n_fft = 400 is not a power of two; the table is 640 KiB and L2 resident,
// and the whole front end is ~0.2 % of a Whisper encoder pass, so a split-radix FFT buys nothing),
// |X|^2 stays in registers, goes once through LDS and is contracted with the mel filterbank on MFMA again.
// Per 30 s utterance: 1.9 MB in, 1.5 MB out (HBM bound by the algorithm, MFMA bound in this form).
// A second tiny kernel applies the per-utterance dynamic-range clamp (needs the global max).
#include "common.h"

#include <math.h>
#include <mutex>
#include <string.h>
#include <algorithm>

struct MelPlan {
    mis_mel_config cfg{};
    int device = 0;
    int n_freqs = 0, nfp = 0, nmp = 0;
    DevBuf<float> window, cosT, sinT, filt;
};

// ---------------------------------------------------------------------------- kernels
#define MEL_FRAMES 32

__device__ __forceinline__ float padded_sample(const float* __restrict__ a, long long n, long long p, int pad) {
    // WhisperAudio.reflectPad (:89-112): [zeros][reversed a[1..lc]][a][reversed a[n-1-lc..n-2]][zeros], lc = min(pad, n-1)
    long long q = p - pad;
    if (q >= 0 && q < n) return a[q];
    if (n <= 1) return 0.0f;
    if (q < 0) {
        long long j = -q;
        return (j <= n - 1) ? a[j] : 0.0f;
    }
    long long j = q - n + 1;
    return (j <= n - 1) ? a[n - 1 - j] : 0.0f;
}

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
    if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

struct MelParams {
    const float* pcm;        // [B][n_samples]
    float* out;              // [B][frames_out][n_mels]   (log10 values; normalised by k_mel_normalize)
    float* row_max;          // [B]
    const float* window;     // [n_fft]
    const float* cosT;       // [n_fft][nfp]
    const float* sinT;
    const float* filt;       // [nfp][nmp]
    long long n_samples;
    int n_fft, hop, n_mels, nfp, nmp, frames_total, frames_out;
    int raw;                 // 1: frames taken from the signal as given (no centre padding): streaming overlap-save
};

__global__ void __launch_bounds__(256) k_mel_tile(MelParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [32][n_fft + 1], reused as P[32][nfp + 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, f0 = blockIdx.x * MEL_FRAMES;
    const int FS = p.n_fft + 1;
    const float* a = p.pcm + (size_t)b * p.n_samples;
    const int pad = p.n_fft / 2;
    // ---- A: windowed frames
    for (int idx = tid; idx < MEL_FRAMES * p.n_fft; idx += 256) {
        int f = idx / p.n_fft, k = idx - f * p.n_fft;
        int fr = f0 + f;
        float v = 0.0f;
        if (fr < p.frames_total) {
            const long long q = (long long)fr * p.hop + k;
            v = (p.raw ? (q < p.n_samples ? a[q] : 0.0f) : padded_sample(a, p.n_samples, q, pad)) * p.window[k];
        }
        lds[f * FS + k] = v;
    }
    __syncthreads();
    // ---- B: real DFT on v_mfma_f32_32x32x2_f32.  Wave w owns bin tiles w and w+4 (re and im of each).
    const int n_btiles = p.nfp / 32;
    f32x16_t re[2], im[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { re[t][r] = 0.0f; im[t][r] = 0.0f; }
    const int bt0 = wave, bt1 = wave + 4;
    const bool has0 = bt0 < n_btiles, has1 = bt1 < n_btiles;
    const int kh = lane >> 5, col = lane & 31;
    if (has0) {
        const float* c0 = p.cosT + bt0 * 32 + col;
        const float* s0 = p.sinT + bt0 * 32 + col;
        const float* c1 = p.cosT + (has1 ? bt1 : bt0) * 32 + col;
        const float* s1 = p.sinT + (has1 ? bt1 : bt0) * 32 + col;
#pragma unroll 4
        for (int k = 0; k < p.n_fft; k += 2) {
            float av = lds[col * FS + k + kh];                       // A[i = lane&31][k + (lane>>5)]
            size_t row = (size_t)(k + kh) * p.nfp;
            re[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, c0[row], re[0], 0, 0, 0);
            im[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s0[row], im[0], 0, 0, 0);
            if (has1) {
                re[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, c1[row], re[1], 0, 0, 0);
                im[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s1[row], im[1], 0, 0, 0);
            }
        }
    }
    __syncthreads();                       // every wave is done reading the frames
    // ---- C: power spectrum -> LDS P[frame][bin]   (C/D: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    const int PS = p.nfp + 1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int bt = t == 0 ? bt0 : bt1;
        if (bt >= n_btiles) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int frow = (r & 3) + 8 * (r >> 2) + 4 * kh;
            lds[frow * PS + bt * 32 + col] = re[t][r] * re[t][r] + im[t][r] * im[t][r];
        }
    }
    __syncthreads();
    // ---- D: mel filterbank contraction, wave w owns mel tile w (and w+4)
    const int n_mtiles = p.nmp / 32;
    float wmax = -INFINITY;
    for (int mt = wave; mt < n_mtiles; mt += 4) {
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float* fp = p.filt + mt * 32 + col;
#pragma unroll 4
        for (int k = 0; k < p.nfp; k += 2) {
            float av = lds[col * PS + k + kh];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, fp[(size_t)(k + kh) * p.nmp], acc, 0, 0, 0);
        }
        // ---- E: log10, store, running max
        int mel = mt * 32 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int fr = f0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (fr < p.frames_out && mel < p.n_mels) {
                float v = log10f(fmaxf(acc[r], 1e-10f));
                p.out[((size_t)b * p.frames_out + fr) * p.n_mels + mel] = v;
                wmax = fmaxf(wmax, v);
            }
        }
    }
    wmax = wave_max(wmax);
    if (lane == 0 && wmax > -INFINITY) atomic_max_float(p.row_max + b, wmax);
}

// n_fft > 512 (computeMelSpectrogram(nFft: 1024) of the Qwen3-TTS speaker encoder, Qwen3TTS.swift:839-880): the frame tile no longer
// fits LDS whole, so the DFT walks the frame in MELB_KC-sample chunks (restaged per group of 8 bin tiles: the samples are L2 hits)
// and the filterbank contraction consumes the power spectrum 256 bins at a time; accumulators stay in registers across groups.
#define MELB_KC 128
__global__ void __launch_bounds__(256) k_mel_tile_big(MelParams p) {
    __shared__ float As[MEL_FRAMES][MELB_KC + 1];
    __shared__ float Ps[MEL_FRAMES][256 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, f0 = blockIdx.x * MEL_FRAMES;
    const float* a = p.pcm + (size_t)b * p.n_samples;
    const int pad = p.n_fft / 2;
    const int n_btiles = p.nfp / 32, n_mtiles = p.nmp / 32;
    const int kh = lane >> 5, col = lane & 31;
    f32x16_t macc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) macc[t][r] = 0.0f;
    for (int g = 0; g < n_btiles; g += 8) {
        f32x16_t re[2], im[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { re[t][r] = 0.0f; im[t][r] = 0.0f; }
        const int bt0 = g + wave, bt1 = g + wave + 4;
        const bool has0 = bt0 < n_btiles, has1 = bt1 < n_btiles;
        const float* c0 = p.cosT + (has0 ? bt0 : 0) * 32 + col;
        const float* s0 = p.sinT + (has0 ? bt0 : 0) * 32 + col;
        const float* c1 = p.cosT + (has1 ? bt1 : 0) * 32 + col;
        const float* s1 = p.sinT + (has1 ? bt1 : 0) * 32 + col;
        for (int k0 = 0; k0 < p.n_fft; k0 += MELB_KC) {
            __syncthreads();
            for (int idx = tid; idx < MEL_FRAMES * MELB_KC; idx += 256) {
                const int f = idx / MELB_KC, kk = idx - f * MELB_KC, k = k0 + kk, fr = f0 + f;
                float v = 0.0f;
                if (fr < p.frames_total && k < p.n_fft) {
                    const long long q = (long long)fr * p.hop + k;
                    v = (p.raw ? (q < p.n_samples ? a[q] : 0.0f) : padded_sample(a, p.n_samples, q, pad)) * p.window[k];
                }
                As[f][kk] = v;
            }
            __syncthreads();
            const int kn = min(MELB_KC, p.n_fft - k0);
            if (has0) {
#pragma unroll 4
                for (int kk = 0; kk < kn; kk += 2) {
                    const float av = As[col][kk + kh];
                    const size_t row = (size_t)(k0 + kk + kh) * p.nfp;
                    re[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, c0[row], re[0], 0, 0, 0);
                    im[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s0[row], im[0], 0, 0, 0);
                    if (has1) {
                        re[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, c1[row], re[1], 0, 0, 0);
                        im[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s1[row], im[1], 0, 0, 0);
                    }
                }
            }
        }
        // power spectrum of this group -> Ps (the previous group's filterbank reads finished before the barriers above)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int lt = wave + 4 * t;                              // local bin tile 0..7
            const bool has = (g + lt) < n_btiles;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int frow = (r & 3) + 8 * (r >> 2) + 4 * kh;
                Ps[frow][lt * 32 + col] = has ? re[t][r] * re[t][r] + im[t][r] * im[t][r] : 0.0f;
            }
        }
        __syncthreads();
        const int bins = min(256, p.nfp - g * 32);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int mt = wave + 4 * t;
            if (mt >= n_mtiles) continue;
            const float* fp = p.filt + (size_t)g * 32 * p.nmp + mt * 32 + col;
#pragma unroll 4
            for (int k = 0; k < bins; k += 2) {
                const float av = Ps[col][k + kh];
                macc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, fp[(size_t)(k + kh) * p.nmp], macc[t], 0, 0, 0);
            }
        }
    }
    float wmax = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int mt = wave + 4 * t;
        if (mt >= n_mtiles) continue;
        const int mel = mt * 32 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int fr = f0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (fr < p.frames_out && mel < p.n_mels) {
                const float v = log10f(fmaxf(macc[t][r], 1e-10f));
                p.out[((size_t)b * p.frames_out + fr) * p.n_mels + mel] = v;
                wmax = fmaxf(wmax, v);
            }
        }
    }
    wmax = wave_max(wmax);
    if (lane == 0 && wmax > -INFINITY) atomic_max_float(p.row_max + b, wmax);
}

__global__ void k_mel_normalize(float* __restrict__ out, const float* __restrict__ row_max, size_t per_row, int batch) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int b = blockIdx.y;
    if (i >= per_row || b >= batch) return;
    float floor_v = row_max[b] - 8.0f;                          // WhisperAudio.swift:75-76
    float v = fmaxf(out[(size_t)b * per_row + i], floor_v);
    out[(size_t)b * per_row + i] = (v + 4.0f) * 0.25f;          // :77  (x + 4) / 4
}

__global__ void k_fill_f32(float* p, float v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------- host
static std::mutex g_plan_mu;
static std::vector<MelPlan*> g_plans;

// melFilters, DSP.swift:76-168, float32 scalar arithmetic (same operation order as the reference)
static std::vector<float> build_mel_filters(const mis_mel_config& c, int n_freqs, int nfp, int nmp) {
    const float fMin = 0.0f, fMax = (float)c.sample_rate / 2.0f;
    std::vector<float> all(n_freqs);
    for (int i = 0; i < n_freqs; ++i) all[i] = (float)i * (float)c.sample_rate / (float)c.n_fft;
    const float fSp = 200.0f / 3.0f, minLogHz = 1000.0f, minLogMel = (minLogHz - fMin) / fSp, logStep = logf(6.4f) / 27.0f;
    auto hz2mel = [&](float f) -> float {
        if (c.mel_scale == 0) return 2595.0f * log10f(1.0f + f / 700.0f);
        return f < minLogHz ? (f - fMin) / fSp : minLogMel + logf(f / minLogHz) / logStep;
    };
    auto mel2hz = [&](float m) -> float {
        if (c.mel_scale == 0) return 700.0f * (powf(10.0f, m / 2595.0f) - 1.0f);
        return m < minLogMel ? fMin + fSp * m : minLogHz * expf(logStep * (m - minLogMel));
    };
    float mMin = hz2mel(fMin), mMax = hz2mel(fMax);
    std::vector<float> fpts(c.n_mels + 2);
    for (int i = 0; i < c.n_mels + 2; ++i) fpts[i] = mel2hz(mMin + (float)i * (mMax - mMin) / (float)(c.n_mels + 1));
    std::vector<float> fb((size_t)nfp * nmp, 0.0f);
    for (int j = 0; j < c.n_mels; ++j) {
        float low = fpts[j], center = fpts[j + 1], high = fpts[j + 2];
        float enorm = 2.0f / (high - low);
        for (int i = 0; i < n_freqs; ++i) {
            float f = all[i], v = 0.0f;
            if (f >= low && f < center) v = (f - low) / (center - low);
            else if (f >= center && f <= high) v = (high - f) / (high - center);
            if (c.slaney_norm) v *= enorm;
            fb[(size_t)i * nmp + j] = v;
        }
    }
    return fb;
}

static MelPlan* get_plan(int device, const mis_mel_config& c) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    for (MelPlan* p : g_plans)
        if (p->device == device && memcmp(&p->cfg, &c, sizeof(c)) == 0) return p;
    MIS_REQUIRE(c.n_fft >= 16 && c.n_fft <= 2048 && (c.n_fft % 2) == 0, MIS_ERR_INVALID_INPUT, "n_fft must be even, 16..2048");
    MIS_REQUIRE(c.hop_length >= 1 && c.n_mels >= 1 && c.n_mels <= 256 && c.sample_rate > 0, MIS_ERR_INVALID_INPUT, "bad mel config");
    MelPlan* p = new MelPlan();
    p->cfg = c; p->device = device;
    p->n_freqs = c.n_fft / 2 + 1;
    p->nfp = (int)round_up(p->n_freqs, 32);
    p->nmp = (int)round_up(c.n_mels, 32);
    std::vector<float> win(c.n_fft), ct((size_t)c.n_fft * p->nfp, 0.0f), st((size_t)c.n_fft * p->nfp, 0.0f);
    for (int n = 0; n < c.n_fft; ++n) {
        // periodic Hann (WhisperAudio.swift:42-43) or symmetric Hann (DSP.hanningWindow :15-22), float32 like the reference
        float denom = c.window == 0 ? (float)c.n_fft : (float)(c.n_fft - 1);
        win[n] = 0.5f * (1.0f - cosf(2.0f * (float)M_PI * (float)n / denom));
        for (int k = 0; k < p->n_freqs; ++k) {
            // exact angle reduction: (n*k) mod N keeps the argument in [0, 2pi)
            double ang = 2.0 * M_PI * (double)((long long)n * k % c.n_fft) / (double)c.n_fft;
            ct[(size_t)n * p->nfp + k] = (float)cos(ang);
            st[(size_t)n * p->nfp + k] = (float)(-sin(ang));
        }
    }
    std::vector<float> fb = build_mel_filters(c, p->n_freqs, p->nfp, p->nmp);
    p->window.alloc(win.size()); p->cosT.alloc(ct.size()); p->sinT.alloc(st.size()); p->filt.alloc(fb.size());
    HIP_CHECK(hipMemcpy(p->window.p, win.data(), win.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(p->cosT.p, ct.data(), ct.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(p->sinT.p, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(p->filt.p, fb.data(), fb.size() * 4, hipMemcpyHostToDevice));
    g_plans.push_back(p);
    return p;
}

extern "C" int64_t mis_mel_num_frames(const mis_mel_config* c, int64_t n_samples) {
    if (!c || n_samples < 0 || c->hop_length < 1) return 0;
    int64_t padded = n_samples + 2 * (c->n_fft / 2);
    int64_t total = padded >= c->n_fft ? 1 + (padded - c->n_fft) / c->hop_length : 0;
    if (c->drop_last_frame && total > 0) total -= 1;
    return total;
}

static void run_mel(int device, const mis_mel_config& c, const float* pcm_dev, int batch, int64_t n_samples,
                    float* out_dev, hipStream_t s) {
    MelPlan* pl = get_plan(device, c);
    int64_t padded = n_samples + 2 * (c.n_fft / 2);
    int frames_total = padded >= c.n_fft ? (int)(1 + (padded - c.n_fft) / c.hop_length) : 0;
    int frames_out = (int)mis_mel_num_frames(&c, n_samples);
    if (frames_out <= 0 || batch <= 0) return;
    DevBuf<float> rmax;
    rmax.alloc(batch);
    hipLaunchKernelGGL(k_fill_f32, dim3(cdiv(batch, 64)), dim3(64), 0, s, rmax.p, -INFINITY, batch);
    MelParams mp{};
    mp.pcm = pcm_dev; mp.out = out_dev; mp.row_max = rmax.p;
    mp.window = pl->window.p; mp.cosT = pl->cosT.p; mp.sinT = pl->sinT.p; mp.filt = pl->filt.p;
    mp.n_samples = n_samples; mp.n_fft = c.n_fft; mp.hop = c.hop_length; mp.n_mels = c.n_mels;
    mp.nfp = pl->nfp; mp.nmp = pl->nmp; mp.frames_total = frames_total; mp.frames_out = frames_out;
    size_t smem = (size_t)MEL_FRAMES * (std::max(c.n_fft, pl->nfp) + 1) * sizeof(float);
    if (c.n_fft <= 512) {
        MIS_REQUIRE(smem <= 64 * 1024, MIS_ERR_INVALID_INPUT, "mel tile does not fit LDS");
        hipLaunchKernelGGL(k_mel_tile, dim3(cdiv(frames_out, MEL_FRAMES), batch), dim3(256), smem, s, mp);
    } else
        hipLaunchKernelGGL(k_mel_tile_big, dim3(cdiv(frames_out, MEL_FRAMES), batch), dim3(256), 0, s, mp);
    size_t per_row = (size_t)frames_out * c.n_mels;
    hipLaunchKernelGGL(k_mel_normalize, dim3((unsigned)((per_row + 255) / 256), batch), dim3(256), 0, s, out_dev, rmax.p,
                       per_row, batch);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(s));        // rmax is freed on return
}

// device-pointer entry for other engines (the Qwen3-TTS speaker encoder): pcm_dev [batch][n_samples] -> out_dev [batch][frames][n_mels]
void mel_spectrogram_device(int device, const mis_mel_config& c, const float* pcm_dev, int batch, int64_t n_samples, float* out_dev, hipStream_t s) {
    run_mel(device, c, pcm_dev, batch, n_samples, out_dev, s);
}

// device-pointer entry used by the Whisper engine: pcm_dev [batch][480000] (already padded) -> out_dev [batch][3000][n_mels]
void whisper_features_device(int device, const float* pcm_dev, int batch, int n_mels, float* out_dev, hipStream_t s) {
    mis_mel_config c{};
    c.sample_rate = 16000; c.n_fft = 400; c.hop_length = 160; c.n_mels = n_mels;
    c.window = 0; c.mel_scale = 1; c.slaney_norm = 1; c.drop_last_frame = 1;
    run_mel(device, c, pcm_dev, batch, 480000, out_dev, s);
}

extern "C" mis_status mis_mel_spectrogram(int device, const mis_mel_config* cfg, const float* pcm, int batch,
                                          int64_t n_samples, float* out, int64_t* n_frames_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(cfg && n_frames_out, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(batch >= 0 && n_samples >= 0, MIS_ERR_INVALID_INPUT, "negative size");
    int64_t frames = mis_mel_num_frames(cfg, n_samples);
    *n_frames_out = frames;
    if (batch == 0 || frames == 0) return MIS_OK;
    MIS_REQUIRE(pcm && out, MIS_ERR_INVALID_INPUT, "null pointer");
    HIP_CHECK(hipSetDevice(device));
    DevBuf<float> din, dout;
    din.alloc((size_t)batch * n_samples);
    dout.alloc((size_t)batch * frames * cfg->n_mels);
    HIP_CHECK(hipMemcpy(din.p, pcm, (size_t)batch * n_samples * 4, hipMemcpyDefault));
    run_mel(device, *cfg, din.p, batch, n_samples, dout.p, 0);
    HIP_CHECK(hipMemcpy(out, dout.p, (size_t)batch * frames * cfg->n_mels * 4, hipMemcpyDefault));
    MIS_API_END
}

// WhisperAudio.encoderFeatures (:83-87): pad/trim every row to 30 s, log-mel, [batch, 3000, n_mels]
extern "C" mis_status mis_whisper_encoder_features(int device, const float* pcm, const int64_t* lens, int batch,
                                                   int64_t stride, int n_mels, float* out) {
    MIS_API_BEGIN
    MIS_REQUIRE(batch >= 0 && stride >= 0, MIS_ERR_INVALID_INPUT, "negative size");
    MIS_REQUIRE(n_mels == 80 || n_mels == 128, MIS_ERR_INVALID_INPUT, "Whisper uses 80 or 128 mel bins");
    if (batch == 0) return MIS_OK;
    MIS_REQUIRE(out && (stride == 0 || pcm), MIS_ERR_INVALID_INPUT, "null pointer");
    HIP_CHECK(hipSetDevice(device));
    const int64_t W = 480000;                                   // WhisperConfig.swift:188-193
    mis_mel_config c{};
    c.sample_rate = 16000; c.n_fft = 400; c.hop_length = 160; c.n_mels = n_mels;
    c.window = 0; c.mel_scale = 1; c.slaney_norm = 1; c.drop_last_frame = 1;
    std::vector<int64_t> hl(batch, stride);
    if (lens) HIP_CHECK(hipMemcpy(hl.data(), lens, batch * sizeof(int64_t), hipMemcpyDefault));
    DevBuf<float> din, dout;
    din.alloc((size_t)batch * W);
    dout.alloc((size_t)batch * 3000 * n_mels);
    HIP_CHECK(hipMemset(din.p, 0, (size_t)batch * W * 4));      // padOrTrimToWindow (:7-13): zero pad
    for (int b = 0; b < batch; ++b) {
        int64_t n = std::min<int64_t>(std::min<int64_t>(hl[b], stride), W);
        MIS_REQUIRE(hl[b] >= 0, MIS_ERR_INVALID_INPUT, "negative length");
        if (n > 0) HIP_CHECK(hipMemcpy(din.p + (size_t)b * W, pcm + (size_t)b * stride, (size_t)n * 4, hipMemcpyDefault));
    }
    run_mel(device, c, din.p, batch, W, dout.p, 0);
    HIP_CHECK(hipMemcpy(out, dout.p, (size_t)batch * 3000 * n_mels * 4, hipMemcpyDefault));
    MIS_API_END
}

// ---------------------------------------------------------------------------- streaming front end
// IncrementalMelSpectrogram (Sources/MLXAudioSTT/Streaming/IncrementalMelSpectrogram.swift:17-215): overlap-save framing across
// chunk boundaries (n_fft - hop samples carried), reflected prefix on the first chunk, log10 clamp against the RUNNING maximum of
// the session.  The chunk bookkeeping is host state exactly as in the reference; framing, DFT, filterbank, log and normalisation
// run in k_mel_tile / k_mel_normalize (raw mode, row_max seeded with the running maximum).
struct mis_mel_stream {
    int device = 0;
    mis_mel_config cfg{};
    std::vector<float> overlap;
    bool first = true;
    float running_max = -INFINITY;
    int64_t total_frames = 0;
    hipStream_t stream = nullptr;
    DevBuf<float> sig, out, rmax;
};

extern "C" mis_status mis_mel_stream_create(int device, int sample_rate, int n_fft, int hop_length, int n_mels, mis_mel_stream** out) {
    MIS_API_BEGIN
    MIS_REQUIRE(out && n_fft >= 2 && hop_length >= 1 && hop_length <= n_fft && n_mels >= 1 && sample_rate >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    int n = 0;
    HIP_CHECK(hipGetDeviceCount(&n));
    MIS_REQUIRE(device >= 0 && device < n, MIS_ERR_DEVICE, "device %d not available (%d GPUs visible)", device, n);
    HIP_CHECK(hipSetDevice(device));
    mis_mel_stream* h = new mis_mel_stream();
    h->device = device;
    h->cfg.sample_rate = sample_rate; h->cfg.n_fft = n_fft; h->cfg.hop_length = hop_length; h->cfg.n_mels = n_mels;
    h->cfg.window = 1; h->cfg.mel_scale = 0; h->cfg.slaney_norm = 1; h->cfg.drop_last_frame = 0;      // hanningWindow + melFilters(norm: "slaney") (:54-60)
    HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->rmax.alloc(1);
    *out = h;
    MIS_API_END
}
extern "C" void mis_mel_stream_destroy(mis_mel_stream* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    delete h;
}
extern "C" mis_status mis_mel_stream_reset(mis_mel_stream* h) {
    MIS_API_BEGIN
    MIS_REQUIRE(h, MIS_ERR_INVALID_INPUT, "null handle");
    h->overlap.clear(); h->first = true; h->running_max = -INFINITY; h->total_frames = 0;
    MIS_API_END
}
extern "C" int64_t mis_mel_stream_total_frames(const mis_mel_stream* h) { return h ? h->total_frames : 0; }

static int64_t mel_stream_frames(mis_mel_stream* h, const std::vector<float>& signal, int64_t n_frames, float* out_host) {
    HIP_CHECK(hipSetDevice(h->device));
    const mis_mel_config& c = h->cfg;
    MelPlan* pl = get_plan(h->device, c);
    hipStream_t s = h->stream;
    h->sig.alloc(signal.size());
    h->out.alloc((size_t)n_frames * c.n_mels);
    HIP_CHECK(hipMemcpyAsync(h->sig.p, signal.data(), signal.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(h->rmax.p, &h->running_max, 4, hipMemcpyHostToDevice, s));
    MelParams mp{};
    mp.pcm = h->sig.p; mp.out = h->out.p; mp.row_max = h->rmax.p;
    mp.window = pl->window.p; mp.cosT = pl->cosT.p; mp.sinT = pl->sinT.p; mp.filt = pl->filt.p;
    mp.n_samples = (long long)signal.size(); mp.n_fft = c.n_fft; mp.hop = c.hop_length; mp.n_mels = c.n_mels;
    mp.nfp = pl->nfp; mp.nmp = pl->nmp; mp.frames_total = (int)n_frames; mp.frames_out = (int)n_frames; mp.raw = 1;
    size_t smem = (size_t)MEL_FRAMES * (std::max(c.n_fft, pl->nfp) + 1) * sizeof(float);
    MIS_REQUIRE(smem <= 64 * 1024, MIS_ERR_INVALID_INPUT, "mel tile does not fit LDS");
    hipLaunchKernelGGL(k_mel_tile, dim3(cdiv(n_frames, MEL_FRAMES), 1), dim3(256), smem, s, mp);
    size_t per_row = (size_t)n_frames * c.n_mels;
    hipLaunchKernelGGL(k_mel_normalize, dim3((unsigned)((per_row + 255) / 256), 1), dim3(256), 0, s, h->out.p, h->rmax.p, per_row, 1);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(out_host, h->out.p, per_row * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(&h->running_max, h->rmax.p, 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    h->total_frames += n_frames;
    return n_frames;
}

// process(samples:) (:68-147): out f32 [capacity_frames, n_mels]; *n_frames = 0 when the chunk does not complete a frame
extern "C" mis_status mis_mel_stream_process(mis_mel_stream* h, const float* samples, int64_t n, float* out, int64_t capacity_frames,
                                             int64_t* n_frames) {
    MIS_API_BEGIN
    MIS_REQUIRE(h && n_frames && n >= 0, MIS_ERR_INVALID_INPUT, "bad argument");
    *n_frames = 0;
    if (n == 0) return MIS_OK;
    MIS_REQUIRE(samples, MIS_ERR_INVALID_INPUT, "null samples");
    const int nfft = h->cfg.n_fft, hop = h->cfg.hop_length, ov = nfft - hop;
    std::vector<float> sm((size_t)n);
    HIP_CHECK(hipMemcpy(sm.data(), samples, (size_t)n * 4, hipMemcpyDefault));
    std::vector<float> signal;
    if (h->first) {
        const int pad = nfft / 2;
        std::vector<float> prefix;
        if (n > 1) {
            const int64_t rl = std::min<int64_t>(pad, n - 1);
            for (int64_t i = rl; i >= 1; --i) prefix.push_back(sm[(size_t)i]);
        }
        if (prefix.empty()) prefix.assign(pad, sm[0]);
        else while ((int)prefix.size() < pad) {
            size_t need = (size_t)pad - prefix.size(), have = prefix.size();
            for (size_t i = 0; i < std::min(need, have); ++i) prefix.push_back(prefix[i]);
        }
        signal = prefix;
        signal.insert(signal.end(), sm.begin(), sm.end());
        h->first = false;
    } else {
        signal = h->overlap;
        signal.insert(signal.end(), sm.begin(), sm.end());
    }
    const int64_t nf = std::max<int64_t>(0, ((int64_t)signal.size() - nfft) / hop + 1);
    if ((int64_t)signal.size() < nfft || nf <= 0) { h->overlap = signal; return MIS_OK; }
    MIS_REQUIRE(out && nf <= capacity_frames, MIS_ERR_INVALID_INPUT, "output buffer holds %lld frames, chunk produces %lld", (long long)capacity_frames, (long long)nf);
    const int64_t consumed = (nf - 1) * hop + nfft;
    if (consumed < (int64_t)signal.size()) h->overlap.assign(signal.begin() + (consumed - ov), signal.end());
    else h->overlap.assign(signal.end() - ov, signal.end());
    *n_frames = mel_stream_frames(h, signal, nf, out);
    MIS_API_END
}
// flush() (:150-200)
extern "C" mis_status mis_mel_stream_flush(mis_mel_stream* h, float* out, int64_t capacity_frames, int64_t* n_frames) {
    MIS_API_BEGIN
    MIS_REQUIRE(h && n_frames, MIS_ERR_INVALID_INPUT, "bad argument");
    *n_frames = 0;
    if (h->overlap.empty()) return MIS_OK;
    const int nfft = h->cfg.n_fft, hop = h->cfg.hop_length;
    std::vector<float> signal = h->overlap;
    if ((int)signal.size() < nfft) signal.resize(nfft, 0.0f);
    const int64_t len = (int64_t)signal.size(), rl = std::min<int64_t>(nfft / 2, len - 1);
    for (int64_t i = len - 2; i >= len - 1 - rl; --i) signal.push_back(signal[(size_t)i]);
    h->overlap.clear();
    const int64_t nf = std::max<int64_t>(0, ((int64_t)signal.size() - nfft) / hop + 1);
    if (nf <= 0) return MIS_OK;
    MIS_REQUIRE(out && nf <= capacity_frames, MIS_ERR_INVALID_INPUT, "output buffer too small");
    *n_frames = mel_stream_frames(h, signal, nf, out);
    MIS_API_END
}
