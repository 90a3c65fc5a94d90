// soprano.hip - Soprano TTS: Qwen3-style token LM (lm_engine.hip, q/k-norm variant) whose per-token hidden states
// are decoded by a Vocos / ConvNeXt backbone + ISTFT head, all in float32 on the GPU.
//
// Reference being replaced: SopranoModel / SopranoDecoder / ISTFTHead / interpolate1d
// (Sources/MLXAudioTTS/Models/Soprano/Soprano.swift:201-690,801-901, SopranoDecoder.swift:22-284) and
// VocosBackbone / ConvNeXtBlock (Sources/MLXAudioCodecs/Vocos/VocosBackbone.swift:18-204).  The reference runs
// the ISTFT overlap-add as a Swift host loop over asArray copies (SopranoDecoder.swift:161-186); here the irfft is
// an exact-f32 MFMA contraction against a host-built inverse-DFT table and the overlap-add is one gather kernel.
// Activations are NCT ([B][C][T], time contiguous) so the f32 codec kernels of snac.hip are reused unchanged.
#include "common.h"
#include "kernels.h"
#include "codec_kernels.h"

#include <math.h>
#include <string.h>
#include <algorithm>
#include <chrono>

struct mis_soprano {
    int device = 0;
    mis_soprano_config cfg{};
    mis_tts* lm = nullptr;
    std::map<std::string, std::vector<float>> raw;
    std::map<std::string, std::vector<int64_t>> raw_shape;
    bool finalized = false;
    DevBuf<float> arena;
    size_t embed_w = 0, embed_b = 0, norm_w = 0, norm_b = 0, fin_w = 0, fin_b = 0, head_w = 0, head_b = 0, idft = 0, window = 0;
    struct Blk { size_t dw, dwb, lnw, lnb, p1, b1, p2, b2, gamma; };
    std::vector<Blk> blocks;
    DevBuf<float> buf[4];
    CodecPack pack;                      // split-bf16 weight fragments + activation scratch (codec_bf3.hip)
    DevBuf<float> hidden;
    TokenEngineScratch* te_scratch = nullptr;   // batch-1 token engine: buffers kept between requests (token_engine.hip)
    int group_batch = 0;                 // inside a group call: the request's batch over ALL shards (0: not in a group) - the LM program is chosen on it
    int lm_path = 0;                     // last generate: 0 launch chain (by rule), 1 token engine, 2 launch chain after an engine time-out
};

// ---------------------------------------------------------------------------- kernels
// hidden [B][L][C] -> x [B][C][T], T = up*(L-1)+1, linear interpolation with align_corners (SopranoDecoder.swift:22-80)
__global__ void k_sop_interp(const float* __restrict__ hid, int64_t hid_row_stride, float* __restrict__ x, int L, int C, int T) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    float v;
    const float* hb = hid + (size_t)b * hid_row_stride * C;
    if (L == 1 || T == L) v = hb[(size_t)(L == 1 ? 0 : t) * C + c];
    else {
        float pos = (float)t * ((float)(L - 1) / (float)(T - 1));
        int lo = (int)floorf(pos);
        int hi = min(lo + 1, L - 1);
        float fr = pos - (float)lo;
        v = hb[(size_t)lo * C + c] * (1.0f - fr) + hb[(size_t)hi * C + c] * fr;
    }
    x[((size_t)b * C + c) * T + t] = v;
}

// [B][C][T] -> [B][k*C][T]: row kk*C + c holds x[c][t + kk - k/2] (zero outside)   (Conv1d padding k/2)
__global__ void k_sop_im2col(const float* __restrict__ x, float* __restrict__ y, int C, int T, int k) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    int kk = r / C, c = r - kk * C;
    int ts = t + kk - k / 2;
    y[((size_t)b * k * C + r) * T + t] = (ts >= 0 && ts < T) ? x[((size_t)b * C + c) * T + ts] : 0.0f;
}

// LayerNorm over the CHANNEL axis of NCT data (eps 1e-6).  Block = 32 time steps x 8 channel groups: a thread walks every eighth channel
// of its column (loads coalesced over t), the eight partial sums of a column meet in LDS in a fixed order; mean first, then the centred
// sum of squares (two passes, as the reference's LayerNorm).  (Round 5: the first version - one thread per time step looping over all
// 768 channels three times - took 357 us per call at one row of 257 frames, ten calls per decode = 3.6 of the 24 ms of a batch-1
// generate, profiles/r05_final_soprano_engine_kernel_stats.csv: two blocks on a 256-CU part.)
#define SOP_LN_TX 32
#define SOP_LN_CG 8
__global__ void __launch_bounds__(SOP_LN_TX * SOP_LN_CG) k_sop_ln_ct(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ w,
                                                                      const float* __restrict__ bias, int C, int T, float eps) {
    __shared__ float red[SOP_LN_CG][SOP_LN_TX];
    const int tx = threadIdx.x & (SOP_LN_TX - 1), cg = threadIdx.x / SOP_LN_TX;
    const int t = blockIdx.x * SOP_LN_TX + tx, b = blockIdx.y;
    const bool live = t < T;
    const float* xb = x + (size_t)b * C * T + (live ? t : 0);
    float s = 0.0f;
    for (int c = cg; c < C; c += SOP_LN_CG) s += xb[(size_t)c * T];
    red[cg][tx] = s;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int g = 0; g < SOP_LN_CG; ++g) tot += red[g][tx];
    const float mean = tot / (float)C;
    __syncthreads();
    float q = 0.0f;
    for (int c = cg; c < C; c += SOP_LN_CG) { const float d = xb[(size_t)c * T] - mean; q += d * d; }
    red[cg][tx] = q;
    __syncthreads();
    float qt = 0.0f;
#pragma unroll
    for (int g = 0; g < SOP_LN_CG; ++g) qt += red[g][tx];
    const float rstd = 1.0f / sqrtf(qt / (float)C + eps);
    if (!live) return;
    float* yb = y + (size_t)b * C * T + t;
    for (int c = cg; c < C; c += SOP_LN_CG) yb[(size_t)c * T] = (xb[(size_t)c * T] - mean) * rstd * w[c] + bias[c];
}

// The same LayerNorm with a column's channels spread over 64 threads (16 time steps x 64 channel groups per block, <= 12 channels per thread
// held in registers: ONE pass over memory instead of three, 17 blocks instead of 9 at 257 frames).  Round 6: the kernel above still took
// 51 us per call at batch 1 - 96 strided loads per thread, three times - ten calls per decode = 0.5 ms of a 21 ms generate
// (profiles/r05_final_soprano_engine_kernel_stats.csv).  Partial sums meet in LDS in a fixed order; mean first, then the centred squares.
#define SOP_LN2_TX 16
#define SOP_LN2_CG 64
#define SOP_LN2_PER 12
__global__ void __launch_bounds__(SOP_LN2_TX * SOP_LN2_CG) k_sop_ln_ct2(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ w,
                                                                        const float* __restrict__ bias, int C, int T, float eps) {
    __shared__ float red[SOP_LN2_CG][SOP_LN2_TX];
    const int tx = threadIdx.x & (SOP_LN2_TX - 1), cg = threadIdx.x / SOP_LN2_TX;
    const int t = blockIdx.x * SOP_LN2_TX + tx, b = blockIdx.y;
    const bool live = t < T;
    const float* xb = x + (size_t)b * C * T + (live ? t : 0);
    float v[SOP_LN2_PER], wv[SOP_LN2_PER], bv[SOP_LN2_PER];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < SOP_LN2_PER; ++k) {
        const int c = cg + k * SOP_LN2_CG;
        const int cc = c < C ? c : C - 1;                 // (clamped: loaded, masked below)
        v[k] = xb[(size_t)cc * T]; wv[k] = w[cc]; bv[k] = bias[cc];
    }
#pragma unroll
    for (int k = 0; k < SOP_LN2_PER; ++k) s += (cg + k * SOP_LN2_CG < C) ? v[k] : 0.0f;
    red[cg][tx] = s;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll 8
    for (int g = 0; g < SOP_LN2_CG; ++g) tot += red[g][tx];
    const float mean = tot / (float)C;
    __syncthreads();
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < SOP_LN2_PER; ++k) { const float d = v[k] - mean; q += (cg + k * SOP_LN2_CG < C) ? d * d : 0.0f; }
    red[cg][tx] = q;
    __syncthreads();
    float qt = 0.0f;
#pragma unroll 8
    for (int g = 0; g < SOP_LN2_CG; ++g) qt += red[g][tx];
    const float rstd = 1.0f / sqrtf(qt / (float)C + eps);
    if (!live) return;
    float* yb = y + (size_t)b * C * T + t;
#pragma unroll
    for (int k = 0; k < SOP_LN2_PER; ++k) {
        const int c = cg + k * SOP_LN2_CG;
        if (c < C) yb[(size_t)c * T] = (v[k] - mean) * rstd * wv[k] + bv[k];
    }
}
static void launch_sop_ln_ct(const float* x, float* y, const float* w, const float* bias, int C, int T, int batch, hipStream_t s) {
    if (C <= SOP_LN2_CG * SOP_LN2_PER)
        hipLaunchKernelGGL(k_sop_ln_ct2, dim3(cdiv(T, SOP_LN2_TX), batch), dim3(SOP_LN2_TX * SOP_LN2_CG), 0, s, x, y, w, bias, C, T, 1e-6f);
    else
        hipLaunchKernelGGL(k_sop_ln_ct, dim3(cdiv(T, SOP_LN_TX), batch), dim3(SOP_LN_TX * SOP_LN_CG), 0, s, x, y, w, bias, C, T, 1e-6f);
}

// head output hh [B][n_fft+2][T] -> spec [B][2*bins][T]: rows 0..bins-1 = mag*cos(phase), bins.. = mag*sin(phase),
// mag = min(exp(.), 100)   (SopranoDecoder.swift:109-122)
__global__ void k_sop_spec(const float* __restrict__ hh, float* __restrict__ spec, int bins, int T) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const float* hb = hh + (size_t)b * 2 * bins * T;
    float mag = fminf(expf(hb[(size_t)k * T + t]), 100.0f);
    float ph = hb[(size_t)(bins + k) * T + t];
    float* sb = spec + (size_t)b * 2 * bins * T;
    sb[(size_t)k * T + t] = mag * cosf(ph);
    sb[(size_t)(bins + k) * T + t] = mag * sinf(ph);
}

// frames [B][n_fft][T] -> audio [B][out_stride]: windowed overlap-add normalised by the window SUM, trimmed by n_fft/2
// at both ends (SopranoDecoder.swift:155-195)
__global__ void k_sop_ola(const float* __restrict__ frames, const float* __restrict__ window, float* __restrict__ audio,
                          int64_t out_stride, int n_fft, int hop, int T) {
    int s = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    int n_out = (T - 1) * hop;
    if (s >= n_out) return;
    int sp = s + n_fft / 2;
    int i1 = min(sp / hop, T - 1);
    int i0 = max(0, (sp - n_fft + hop) / hop);
    const float* fb = frames + (size_t)b * n_fft * T;
    float acc = 0.0f, ws = 0.0f;
    for (int i = i0; i <= i1; ++i) {
        int j = sp - i * hop;
        if (j < 0 || j >= n_fft) continue;
        float w = window[j];
        acc += fb[(size_t)j * T + i] * w;
        ws += w;
    }
    audio[(size_t)b * out_stride + s] = (ws != 0.0f) ? acc / ws : acc;
}

// ---------------------------------------------------------------------------- host
static std::string sop_sanitize(const std::string& key) {          // SopranoModel.sanitize, Soprano.swift:314-361
    std::string k = key;
    if (k.rfind("model.", 0) == 0) k = k.substr(6);
    if (k.rfind("decoder.", 0) == 0) return k;
    if (k.rfind("language_model.lm_head", 0) == 0) return k.substr(strlen("language_model."));
    if (k.rfind("language_model.", 0) == 0) return "model." + k.substr(strlen("language_model."));
    if (k.rfind("lm_head", 0) == 0) return k;
    return "model." + k;
}

extern "C" mis_status mis_soprano_create(const mis_soprano_config* cfg, int device, mis_soprano** out) {
    MIS_API_BEGIN
    MIS_REQUIRE(cfg && out, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(cfg->decoder_num_layers >= 0 && cfg->decoder_dim > 0 && cfg->decoder_intermediate_dim > 0, MIS_ERR_INVALID_INPUT, "bad decoder dims");
    MIS_REQUIRE(cfg->n_fft >= 8 && cfg->n_fft % 2 == 0 && cfg->hop_length >= 1 && cfg->upscale >= 1, MIS_ERR_INVALID_INPUT, "bad ISTFT config");
    MIS_REQUIRE(cfg->input_kernel >= 1 && cfg->input_kernel % 2 == 1 && cfg->dw_kernel >= 1 && cfg->dw_kernel <= 7 && cfg->dw_kernel % 2 == 1,
                MIS_ERR_INVALID_INPUT, "input_kernel must be odd, dw_kernel odd and <= 7");
    mis_lm_config lmc = cfg->lm;
    lmc.qk_norm = 1;                                   // SopranoAttention: q_norm / k_norm, plain RoPE (Soprano.swift:38-61)
    lmc.rope_plain = 1;
    mis_tts* lm = nullptr;
    mis_status st = mis_tts_create(&lmc, nullptr, device, &lm);
    if (st != MIS_OK) return st;
    mis_soprano* c = new mis_soprano();
    c->device = device; c->cfg = *cfg; c->cfg.lm = lmc; c->lm = lm;
    *out = c;
    MIS_API_END
}
extern "C" void mis_soprano_destroy(mis_soprano* c) {
    if (!c) return;
    if (c->lm) mis_tts_destroy(c->lm);
    (void)hipSetDevice(c->device);
    if (c->te_scratch) token_engine_scratch_destroy(c->te_scratch);
    delete c;
}
extern "C" mis_tts* mis_soprano_lm(mis_soprano* c) { return c ? c->lm : nullptr; }

extern "C" mis_status mis_soprano_set_tensor(mis_soprano* c, const char* name_, const void* data, mis_dtype dtype,
                                             const int64_t* shape, int ndim) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && name_ && data && shape && ndim >= 1 && ndim <= 3, MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(!c->finalized, MIS_ERR_INVALID_INPUT, "set_tensor after finalize");
    std::string name = sop_sanitize(name_);
    if (name.rfind("decoder.", 0) != 0) return mis_tts_set_tensor(c->lm, name.c_str(), data, dtype, shape, ndim);
    size_t n = 1;
    std::vector<int64_t> sh;
    for (int i = 0; i < ndim; ++i) { MIS_REQUIRE(shape[i] > 0, MIS_ERR_INVALID_INPUT, "bad shape"); n *= (size_t)shape[i]; sh.push_back(shape[i]); }
    HIP_CHECK(hipSetDevice(c->device));
    size_t esz = dtype == MIS_F32 ? 4 : 2;
    std::vector<uint8_t> host(n * esz);
    HIP_CHECK(hipMemcpy(host.data(), data, n * esz, hipMemcpyDefault));
    std::vector<float> v(n);                                         // decoder weights are float32 (Soprano.swift:332-339)
    if (dtype == MIS_F32) memcpy(v.data(), host.data(), n * 4);
    else if (dtype == MIS_BF16) for (size_t i = 0; i < n; ++i) v[i] = bf16_to_f32(((uint16_t*)host.data())[i]);
    else if (dtype == MIS_F16) for (size_t i = 0; i < n; ++i) v[i] = f16_to_f32_host(((uint16_t*)host.data())[i]);
    else throw MisError(MIS_ERR_INVALID_INPUT, "unsupported dtype");
    c->raw[name] = std::move(v);
    c->raw_shape[name] = sh;
    MIS_API_END
}

static const std::vector<float>& sneed(mis_soprano* c, const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = c->raw.find(name);
    MIS_REQUIRE(it != c->raw.end(), MIS_ERR_NOT_INITIALIZED, "Soprano weight missing: %s", name.c_str());
    MIS_REQUIRE(c->raw_shape[name] == std::vector<int64_t>(shape), MIS_ERR_INVALID_INPUT, "Soprano weight %s has the wrong shape", name.c_str());
    return it->second;
}

extern "C" mis_status mis_soprano_finalize(mis_soprano* c) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && !c->finalized, MIS_ERR_INVALID_INPUT, "bad handle");
    mis_status st = mis_tts_finalize(c->lm);
    if (st != MIS_OK) return st;
    HIP_CHECK(hipSetDevice(c->device));
    const mis_soprano_config& cf = c->cfg;
    const int64_t C = cf.lm.hidden_size, d = cf.decoder_dim, inter = cf.decoder_intermediate_dim, nf = cf.n_fft, bins = nf / 2 + 1;
    const int64_t ik = cf.input_kernel, dk = cf.dw_kernel;
    std::vector<float> arena;
    auto push = [&](const std::vector<float>& v) { size_t o = arena.size(); arena.insert(arena.end(), v.begin(), v.end()); while (arena.size() & 3) arena.push_back(0.f); return o; };
    auto transpose = [&](const std::vector<float>& w, int64_t out_f, int64_t in_f) {       // [out][in] -> A^T [in][out]
        std::vector<float> at((size_t)in_f * out_f);
        for (int64_t o = 0; o < out_f; ++o) for (int64_t i = 0; i < in_f; ++i) at[i * out_f + o] = w[o * in_f + i];
        return at;
    };
    const std::string P = "decoder.decoder";
    {   // embed conv: weight [d][ik][C] (MLX Conv1d) == [d][ik*C] row-major with column kk*C + c
        const auto& w = sneed(c, P + ".embed.weight", {d, ik, C});
        c->embed_w = push(transpose(w, d, ik * C));
        c->embed_b = push(sneed(c, P + ".embed.bias", {d}));
    }
    c->norm_w = push(sneed(c, P + ".norm.weight", {d})); c->norm_b = push(sneed(c, P + ".norm.bias", {d}));
    c->blocks.clear();
    for (int i = 0; i < cf.decoder_num_layers; ++i) {
        std::string q = P + ".convnext." + std::to_string(i);
        mis_soprano::Blk b{};
        const auto& dw = sneed(c, q + ".dwconv.weight", {d, dk, 1});
        std::vector<float> w7((size_t)d * 7, 0.0f);                    // centre the dk taps in a 7-tap kernel
        for (int64_t ch = 0; ch < d; ++ch) for (int64_t j = 0; j < dk; ++j) w7[ch * 7 + (3 - dk / 2) + j] = dw[ch * dk + j];
        b.dw = push(w7); b.dwb = push(sneed(c, q + ".dwconv.bias", {d}));
        b.lnw = push(sneed(c, q + ".norm.weight", {d})); b.lnb = push(sneed(c, q + ".norm.bias", {d}));
        b.p1 = push(transpose(sneed(c, q + ".pwconv1.weight", {inter, d}), inter, d)); b.b1 = push(sneed(c, q + ".pwconv1.bias", {inter}));
        b.p2 = push(transpose(sneed(c, q + ".pwconv2.weight", {d, inter}), d, inter)); b.b2 = push(sneed(c, q + ".pwconv2.bias", {d}));
        b.gamma = push(sneed(c, q + ".gamma", {d}));
        c->blocks.push_back(b);
    }
    c->fin_w = push(sneed(c, P + ".final_layer_norm.weight", {d})); c->fin_b = push(sneed(c, P + ".final_layer_norm.bias", {d}));
    c->head_w = push(transpose(sneed(c, "decoder.head.out.weight", {nf + 2, d}), nf + 2, d));
    c->head_b = push(sneed(c, "decoder.head.out.bias", {nf + 2}));
    {   // irfft as a contraction: frames[n] = (1/N) sum_k c_k (Re_k cos(2 pi k n / N) - Im_k sin(2 pi k n / N)),
        // c_0 = c_{N/2} = 1, else 2; imaginary parts of DC / Nyquist are ignored (MLXFFT.irfft semantics)
        std::vector<float> at((size_t)2 * bins * nf);
        for (int64_t k = 0; k < bins; ++k) {
            double ck = (k == 0 || k == nf / 2) ? 1.0 : 2.0;
            for (int64_t n = 0; n < nf; ++n) {
                double ang = 2.0 * M_PI * (double)((k * n) % nf) / (double)nf;
                at[(size_t)k * nf + n] = (float)(ck * cos(ang) / (double)nf);
                at[(size_t)(bins + k) * nf + n] = (k == 0 || k == nf / 2) ? 0.0f : (float)(-ck * sin(ang) / (double)nf);
            }
        }
        c->idft = push(at);
        std::vector<float> win(nf);
        float factor = (float)M_PI / (float)(nf - 1);                   // hanningWindow, SopranoDecoder.swift:209-217
        for (int64_t n = 0; n < nf; ++n) win[n] = 0.5f - 0.5f * cosf(2.0f * factor * (float)n);
        if (nf == 1) win[0] = 1.0f;
        c->window = push(win);
    }
    c->arena.alloc(arena.size());
    HIP_CHECK(hipMemcpy(c->arena.p, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    c->raw.clear(); c->raw_shape.clear();
    c->finalized = true;
    MIS_API_END
}

extern "C" int64_t mis_soprano_num_samples(const mis_soprano* c, int n_hidden) {
    if (!c || n_hidden < 1) return 0;
    int64_t T = (int64_t)c->cfg.upscale * (n_hidden - 1) + 1;
    return (T - 1) * c->cfg.hop_length;
}

// hidden_dev [B][row_stride][C] (first L rows used) -> audio_dev [B][out_stride]
static void soprano_decode_device(mis_soprano* c, const float* hidden_dev, int64_t row_stride, int batch, int L, float* audio_dev,
                                  int64_t out_stride, hipStream_t s) {
    CodecPackScope pack_scope(&c->pack);
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "Soprano model not finalized");
    const mis_soprano_config& cf = c->cfg;
    const int C = cf.lm.hidden_size, d = cf.decoder_dim, inter = cf.decoder_intermediate_dim, nf = cf.n_fft, bins = nf / 2 + 1;
    const int T = cf.upscale * (L - 1) + 1;
    if (T < 2) return;                                                 // (T-1)*hop == 0 samples
    const float* W = c->arena.p;
    size_t rows = (size_t)std::max({(int)(cf.input_kernel * C), d, inter, nf + 2, 2 * bins, nf});
    for (int i = 0; i < 4; ++i) c->buf[i].alloc((size_t)batch * rows * T);
    float *a = c->buf[0].p, *b = c->buf[1].p, *t1 = c->buf[2].p, *t2 = c->buf[3].p;
    dim3 tb(128), tg(cdiv(T, 128));
    hipLaunchKernelGGL(k_sop_interp, dim3(tg.x, C, batch), tb, 0, s, hidden_dev, row_stride, a, L, C, T);
    const float* xin = a;
    int Kin = C;
    if (cf.input_kernel > 1) {
        hipLaunchKernelGGL(k_sop_im2col, dim3(tg.x, cf.input_kernel * C, batch), tb, 0, s, a, t1, C, T, cf.input_kernel);
        xin = t1; Kin = cf.input_kernel * C;
    }
    GemmParams g{};
    g.AT = W + c->embed_w; g.bias = W + c->embed_b; g.X = xin; g.Y = b; g.M = d; g.K = Kin; g.N = T; g.Tin = T; g.Tout = T;
    launch_gemm(GEMM_PLAIN, false, g, batch, s);
    launch_sop_ln_ct(b, a, W + c->norm_w, W + c->norm_b, d, T, batch, s);
    float* h = a;       // residual stream
    float* o = b;
    for (const auto& blk : c->blocks) {                                // ConvNeXtBlock, VocosBackbone.swift:64-99
        launch_dw7(h, t1, W + blk.dw, W + blk.dwb, batch, d, T, 1, s);
        launch_sop_ln_ct(t1, t2, W + blk.lnw, W + blk.lnb, d, T, batch, s);
        g = GemmParams{};
        g.AT = W + blk.p1; g.bias = W + blk.b1; g.X = t2; g.Y = t1; g.M = inter; g.K = d; g.N = T; g.Tin = T; g.Tout = T;
        g.split_k_ok = 1;                                                // (a sentence is a few hundred frames: 18-54 blocks per row without it)
        launch_gemm(GEMM_GELU, false, g, batch, s);
        g = GemmParams{};
        g.AT = W + blk.p2; g.bias = W + blk.b2; g.X = t1; g.Y = o; g.R = h; g.scale = W + blk.gamma;
        g.M = d; g.K = inter; g.N = T; g.Tin = T; g.Tout = T;
        g.split_k_ok = 1;
        launch_gemm(GEMM_RESID, false, g, batch, s);
        std::swap(h, o);
    }
    launch_sop_ln_ct(h, o, W + c->fin_w, W + c->fin_b, d, T, batch, s);
    g = GemmParams{};
    g.AT = W + c->head_w; g.bias = W + c->head_b; g.X = o; g.Y = t1; g.M = nf + 2; g.K = d; g.N = T; g.Tin = T; g.Tout = T;
    launch_gemm(GEMM_PLAIN, false, g, batch, s);
    hipLaunchKernelGGL(k_sop_spec, dim3(tg.x, bins, batch), tb, 0, s, t1, t2, bins, T);
    g = GemmParams{};
    g.AT = W + c->idft; g.X = t2; g.Y = t1; g.M = nf; g.K = 2 * bins; g.N = T; g.Tin = T; g.Tout = T;
    launch_gemm(GEMM_PLAIN, false, g, batch, s);
    int n_out = (T - 1) * cf.hop_length;
    hipLaunchKernelGGL(k_sop_ola, dim3(cdiv(n_out, 256), batch), dim3(256), 0, s, t1, W + c->window, audio_dev, out_stride, nf,
                       cf.hop_length, T);
    HIP_CHECK(hipGetLastError());
}

extern "C" mis_status mis_soprano_decode(mis_soprano* c, const float* hidden, int batch, int L, float* audio_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c, MIS_ERR_INVALID_INPUT, "null handle");
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "Soprano model not finalized");
    MIS_REQUIRE(batch >= 0 && L >= 0, MIS_ERR_INVALID_INPUT, "negative size");
    int64_t n = mis_soprano_num_samples(c, L);
    if (batch == 0 || n == 0) return MIS_OK;
    MIS_REQUIRE(hidden && audio_out, MIS_ERR_INVALID_INPUT, "null pointer");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = tts_stream(c->lm);
    const int C = c->cfg.lm.hidden_size;
    DevBuf<float> hd, ad;
    hd.alloc((size_t)batch * L * C); ad.alloc((size_t)batch * n);
    HIP_CHECK(hipMemcpyAsync(hd.p, hidden, (size_t)batch * L * C * 4, hipMemcpyDefault, s));
    soprano_decode_device(c, hd.p, L, batch, L, ad.p, n, s);
    HIP_CHECK(hipMemcpyAsync(audio_out, ad.p, (size_t)batch * n * 4, hipMemcpyDefault, s));
    HIP_CHECK(hipStreamSynchronize(s));
    MIS_API_END
}

// SopranoModel.generate / generateStream for already-tokenised sentences (Soprano.swift:577-690 / :693-800 per prompt chunk): LM
// loop with the Soprano sampler flavour collecting hidden states until the stop token, then SopranoDecoder on each row.
// on_event (stream form): .token per sampled id while the loop runs (the stop token is not announced, :855-857), then per row
// .info and ONE .audio (:771-787).
static void soprano_generate_impl(mis_soprano* c, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                  const mis_gen_params* params, float** pcm_out, int64_t* pcm_stride, int64_t* pcm_lens,
                                  int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens, mis_event_cb on_event, void* user,
                                  const volatile int* cancel_flag) {
    MIS_REQUIRE(c && prompt_ids && prompt_lens && params, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "Soprano model not finalized");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = tts_stream(c->lm);
    mis_gen_params gp = *params;
    gp.sampler_flavor = 1;
    gp.frame_constrained = 0;
    if (gp.max_tokens <= 0) gp.max_tokens = 512;                       // parameters.maxTokens ?? 512 (:635)
    std::vector<int32_t> n_hidden, ntok, toks;
    int64_t tstride = 0;
    // One row: the whole LM loop as ONE persistent launch on the compute units of four XCDs (csrc/token_engine.hip: 0.26 ms per position
    // against the launch chain's 0.60 at Soprano-80M's widths) - same sampler arithmetic, same hidden-state rows, and the same program for
    // generate and generateStream (the ids reach host-visible memory while the launch runs; .token events are fired from there,
    // Soprano.swift:877 - the reference's generate is itself built on streamGenerate, :801-885).
    // WHICH program runs the LM loop is a function of the request and the handle only: the engine iff the request is ONE
    // row (of a group call: one row over all shards), the checkpoint has the widths the engine is compiled for (TeShape: Soprano-80M,
    // bf16 - other widths and MLX-quantised checkpoints, SopranoConfig.swift:167-190, take the launch chain at ~0.57 ms per token), the
    // device is 8 XCDs x 32 compute units not shared with another replica, and the request fits the engine's context.
    // MIS_TOKEN_ENGINE = 0 keeps the launch chain, 1 / 2 / 4 / 8 picks the number of XCDs.  The one run-time exception is reported, never
    // silent: if the engine's workers cannot be co-resident (another stream holds compute units: its bounded polls run out before the
    // first id) the request runs on the launch chain, mis_soprano_lm_path() returns 2 and a line goes to stderr - the launch chain's
    // logits differ from the engine's in float32 summation order, so near-tie ids can differ between the two.
    bool by_engine = false;
    c->lm_path = 0;
    {
        const char* e = getenv("MIS_TOKEN_ENGINE");
        const int xcds = e ? atoi(e) : 4;
        int32_t len0 = 0;
        if (batch == 1) HIP_CHECK(hipMemcpy(&len0, prompt_lens, 4, hipMemcpyDefault));
        const bool one_row_request = batch == 1 && (c->group_batch == 0 || c->group_batch == 1);
        if (one_row_request && !tts_internal_shared_device(c->lm) && (xcds == 1 || xcds == 2 || xcds == 4 || xcds == 8) &&
            token_engine_supports(c->lm) && len0 >= 1 &&
            len0 + gp.max_tokens <= 1024 && gp.repetition_context <= 64 && gp.temperature >= 0.0f) {
            TokenEngineRequest rq;
            rq.prompt = prompt_ids; rq.n_prompt = len0; rq.max_new = gp.max_tokens; rq.xcds = xcds; rq.generate = true;
            rq.sample = gp.temperature > 0.0f; rq.temperature = gp.temperature; rq.penalty = gp.repetition_penalty;
            rq.win_cap = std::max(gp.repetition_context, 0); rq.seed = gp.seed; rq.row = gp.row_offset; rq.stop_id = c->cfg.stop_token_id;
            c->hidden.alloc((size_t)(gp.max_tokens + 1) * c->cfg.lm.hidden_size);
            rq.hidden_dev = c->hidden.p; rq.want_hidden = true;
            rq.prefill_by_chain = true;                                   // the prompt (but its last position) in one batched pass of the launch chain
            if (!c->te_scratch) c->te_scratch = token_engine_scratch_create();
            rq.scratch = c->te_scratch;
            rq.cancel = cancel_flag;
            const int stop_id = c->cfg.stop_token_id;
            if (on_event)
                rq.on_token = [&](int, int32_t id) {
                    if (id == stop_id) return;                            // [STOP] ends the row unannounced (Soprano.swift:855-857)
                    int32_t t = id;
                    on_event(user, 0, MIS_EVENT_TOKEN, &t, 1);
                };
            TokenEngineResult r;
            const auto t0 = std::chrono::steady_clock::now();
            try {
                token_engine_run(c->lm, rq, r);
                by_engine = true;
            } catch (const MisError& err) {
                // a time-out after ids were announced cannot be taken back: the request fails (never seen: workers that were co-resident
                // for the first edge stay resident - the launch is persistent)
                if (err.code != MIS_ERR_GENERATION_FAILED || r.n_announced > 0) throw;
                fprintf(stderr, "mi_speech: Soprano token engine timed out (compute units held by another stream); this request runs on the launch chain\n");
                c->lm_path = 2;
            }
            if (by_engine) {
                c->lm_path = 1;
                tstride = gp.max_tokens;
                toks.assign((size_t)gp.max_tokens, 0);
                for (int k = 0; k < r.n_sampled; ++k) toks[k] = r.next_tokens[len0 - 1 + k];
                ntok.assign(1, r.n_sampled);
                n_hidden.assign(1, r.n_positions - (len0 - 1));
                tts_internal_set_decode_ms(c->lm, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
        }
    }
    if (!by_engine)
        tts_generate_hidden(c->lm, prompt_ids, prompt_lens, batch, &gp, c->cfg.stop_token_id, c->hidden, n_hidden, ntok, toks, tstride,
                            on_event, user, cancel_flag);
    const int C = c->cfg.lm.hidden_size;
    const int64_t hid_rows = gp.max_tokens + 1;
    int64_t longest = 0;
    std::vector<int64_t> plens(batch);
    for (int b = 0; b < batch; ++b) { plens[b] = mis_soprano_num_samples(c, n_hidden[b]); longest = std::max(longest, plens[b]); }
    MIS_REQUIRE(longest > 0, MIS_ERR_GENERATION_FAILED, "No audio generated");             // Soprano.swift:684-686
    DevBuf<float> audio;
    audio.alloc((size_t)batch * longest);
    HIP_CHECK(hipMemsetAsync(audio.p, 0, (size_t)batch * longest * 4, s));
    bool same = true;
    for (int b = 1; b < batch; ++b) same = same && n_hidden[b] == n_hidden[0];
    if (same && plens[0] > 0) {                                         // all rows ended at the same step: one batched decode
        soprano_decode_device(c, c->hidden.p, hid_rows, batch, n_hidden[0], audio.p, longest, s);
    } else {
        for (int b = 0; b < batch; ++b)                                 // ragged rows: decode one by one
            if (plens[b] > 0)
                soprano_decode_device(c, c->hidden.p + (size_t)b * hid_rows * C, hid_rows, 1, n_hidden[b], audio.p + (size_t)b * longest, longest, s);
    }
    PinnedBuf<float> host_pin((size_t)batch * longest);
    float* host = host_pin.p;
    HIP_CHECK(hipMemcpyAsync(host, audio.p, (size_t)batch * longest * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (int b = 0; b < batch; ++b) {                                   // audio[0, (-audioLength)...], Soprano.swift:666-671
        int64_t want = (int64_t)(n_hidden[b] - 1) * c->cfg.token_size;
        if (want > 0 && want < plens[b]) {
            memmove(host + (size_t)b * longest, host + (size_t)b * longest + (plens[b] - want), (size_t)want * 4);
            plens[b] = want;
        }
    }
    std::vector<int32_t> ngen(batch);
    for (int b = 0; b < batch; ++b) {
        int n = ntok[b];                                                // the stop token is not a generated token (:855-857)
        if (n > 0 && toks[(size_t)b * tstride + n - 1] == c->cfg.stop_token_id) n -= 1;
        ngen[b] = n;
    }
    if (on_event) {
        const double secs = tts_last_decode_ms(c->lm) * 1e-3;
        for (int b = 0; b < batch; ++b) {
            mis_gen_info info{};                                        // SopranoGenerationInfo (:771-779): prompt count / prefill time are 0
            info.generation_token_count = n_hidden[b];                  // totalTokens += tokenCount (hidden states, :752)
            info.generate_time = secs;
            info.tokens_per_second = secs > 0 ? n_hidden[b] / secs : 0;
            size_t free_b = 0, total_b = 0;
            (void)hipMemGetInfo(&free_b, &total_b);
            info.peak_memory_gb = (double)(total_b - free_b) / 1e9;
            on_event(user, b, MIS_EVENT_INFO, &info, 1);
            on_event(user, b, MIS_EVENT_AUDIO, host + (size_t)b * longest, plens[b]);
        }
    }
    if (tokens_out) {
        PinnedBuf<int32_t> th(toks.size() + 1);
        memcpy(th.p, toks.data(), toks.size() * 4);
        *tokens_out = th.release();
        if (tokens_stride) *tokens_stride = tstride;
    }
    if (pcm_out) { *pcm_out = host_pin.release(); *pcm_stride = longest; }
    if (pcm_lens) for (int b = 0; b < batch; ++b) pcm_lens[b] = plens[b];
    if (n_tokens) for (int b = 0; b < batch; ++b) n_tokens[b] = ngen[b];
}

mis_tts* soprano_internal_lm(mis_soprano* c) { return c ? c->lm : nullptr; }
void soprano_internal_set_group_batch(mis_soprano* c, int batch) { if (c) c->group_batch = batch; }
extern "C" int32_t mis_soprano_lm_path(const mis_soprano* c) { return c ? c->lm_path : 0; }
int soprano_internal_device(const mis_soprano* c) { return c ? c->device : -1; }

extern "C" mis_status mis_soprano_generate(mis_soprano* c, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                           const mis_gen_params* params, float** pcm_out, int64_t* pcm_stride, int64_t* pcm_lens,
                                           int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens) {
    MIS_API_BEGIN
    MIS_REQUIRE(pcm_out && pcm_stride && pcm_lens, MIS_ERR_INVALID_INPUT, "null argument");
    soprano_generate_impl(c, prompt_ids, prompt_lens, batch, params, pcm_out, pcm_stride, pcm_lens, tokens_out, tokens_stride, n_tokens,
                          nullptr, nullptr, nullptr);
    MIS_API_END
}

extern "C" mis_status mis_soprano_generate_stream(mis_soprano* c, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                                  const mis_gen_params* params, mis_event_cb on_event, void* user,
                                                  const volatile int* cancel_flag) {
    MIS_API_BEGIN
    MIS_REQUIRE(on_event, MIS_ERR_INVALID_INPUT, "null callback");
    soprano_generate_impl(c, prompt_ids, prompt_lens, batch, params, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, on_event, user,
                          cancel_flag);
    MIS_API_END
}
