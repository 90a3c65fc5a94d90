// q3_reference.hip - the reference-audio front end of Qwen3-TTS in-context voice cloning, float32:
//   * speaker encoder: log-mel (mel.hip, nFft 1024 / hop 256 / 128 mels) -> ECAPA-TDNN x-vector;
//   * speech-tokenizer encoder: waveform -> 12.5 Hz codec codes (Mimi: SEANet encoder, causal transformer, edge-padded downsampling
//     conv, split residual VQ encode).
//
// Reference being replaced: extractSpeakerEmbedding (Sources/MLXAudioTTS/Models/Qwen3TTS/Qwen3TTS.swift:839-881) ->
// Qwen3TTSSpeakerEncoder (Qwen3TTSSpeakerEncoder.swift: reflectPad1D :6-16, TimeDelayNetBlock :20-42, Res2NetBlock :46-96,
// SqueezeExcitationBlock :100-129, SqueezeExcitationRes2NetBlock :133-184, AttentiveStatisticsPooling :188-233, encoder :237-307);
// Qwen3TTSSpeechTokenizerEncoder.encode (Qwen3TTSSpeechTokenizer.swift:792-881) = the Mimi encoder of Sources/MLXAudioCodecs/Mimi:
// StreamableConv1d (Conv.swift:171-227), SeanetEncoder (Seanet.swift:92-258), ProjectedTransformer (Transformer.swift:107-369),
// ConvDownsample1d (Conv.swift:346-359), SplitResidualVectorQuantizer.encode (Quantization.swift:6-211).
//
// Both run once per reference recording (the model caches the result, Qwen3TTS.swift:268-300), so the design goal is exactness and
// few moving parts, not the last microsecond: activations are [C][T] (time contiguous), every contraction is an exact-f32 MFMA GEMM of
// snac.hip (k_snac_gemm / k_conv_taps: the codebook decisions of the VQ do not survive reduced precision), strided convs become
// two-tap convs over a phase-split copy of their input, the transformer reuses the decoder's attention and normalisation kernels
// (q3_codec.hip; interleaved RoPE = rotate-half RoPE after permuting the q / k rows of every head to evens-then-odds, which leaves
// q.k unchanged), and the residual VQ of a frame runs in one block that keeps the residual in LDS across all quantizer layers.
#include "common.h"
#include "kernels.h"
#include "codec_kernels.h"
#include "q3_kernels.h"

#include <math.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>

struct mis_q3ref {
    int device = 0;
    hipStream_t s = nullptr;
    mis_qwen3tts_reference_config cfg{};
    std::map<std::string, std::vector<float>> raw;
    std::map<std::string, std::vector<int64_t>> raw_shape;
    bool finalized = false, has_spk = false, has_enc = false;
    DevBuf<float> arena;
    struct Lin { size_t w = 0, b = (size_t)-1; int M = 0, K = 0, taps = 1, dil = 1, cin = 0; };
    // speaker encoder
    struct SeBlock { Lin tdnn1, tdnn2, se1, se2; std::vector<Lin> res; };
    Lin spk_first, spk_mfa, asp_tdnn, asp_conv, spk_fc;
    std::vector<SeBlock> spk_blocks;
    // speech-tokenizer encoder
    struct Res { Lin c1, c2, sc; bool has_sc = false; };
    struct EncLayer { std::vector<Res> res; Lin down; int stride = 1, cin = 0; };
    Lin enc_init, enc_final, enc_down;
    std::vector<EncLayer> enc_layers;
    struct TL { size_t n1w, n1b, n2w, n2b, ls1, ls2; Lin qkv, o, f1, f2; };
    std::vector<TL> tlayers;
    struct VqGroup { Lin in_proj; size_t embT = 0, emb = 0, e2h = 0; int nq = 0; } vq[2];
    int ds_stride = 1;
};

// ---------------------------------------------------------------------------- kernels
// y[c][t] = act(x[c][t]): 1 ELU (Seanet.swift:120-121), 2 ReLU, 3 tanh(ReLU) (AttentiveStatisticsPooling :221)
__global__ void k_ref_act(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int T, int act) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (t >= T) return;
    float v = x[(size_t)c * ldx + t];
    if (act == 1) v = v > 0.0f ? v : (expf(v) - 1.0f);
    else if (act == 2) v = fmaxf(v, 0.0f);
    else if (act == 3) v = tanhf(fmaxf(v, 0.0f));
    y[(size_t)c * ldy + t] = v;
}
// reflectPad1D (:6-16) of x (+ add): y[c][i] = x[c][r(i - pad)] + add[c][r(i - pad)], i in [0, T + 2 pad), pad < T
__global__ void k_ref_reflect(const float* __restrict__ x, int ldx, const float* __restrict__ add, int lda, float* __restrict__ y, int ldy,
                              int T, int pad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (i >= T + 2 * pad) return;
    int t = i - pad;
    if (t < 0) t = -t;
    else if (t >= T) t = 2 * (T - 1) - t;
    float v = x[(size_t)c * ldx + t];
    if (add) v += add[(size_t)c * lda + t];
    y[(size_t)c * ldy + i] = v;
}
// strided conv (k = 2 s, stride s, Conv.swift:206-226) as a two-tap conv: y[(c s + r)][m] = act(xp[c][s m + r]) over the padded signal
// xp = [s left pads | x | right pads up to s (M + 1)]; pads are zeros (ELU(0) = 0) or edge copies (ConvDownsample1d)
__global__ void k_ref_phase_split(const float* __restrict__ x, float* __restrict__ y, int T, int s, int M1, int edge, int elu) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (i >= M1 * s) return;
    const int m = i / s, r = i - m * s;
    int t = i - s;
    float v = 0.0f;
    if (t >= 0 && t < T) v = x[(size_t)c * T + t];
    else if (edge) v = x[(size_t)c * T + (t < 0 ? 0 : T - 1)];
    if (elu) v = v > 0.0f ? v : (expf(v) - 1.0f);
    y[((size_t)c * s + r) * M1 + m] = v;
}
__global__ void k_ref_transpose(const float* __restrict__ x /*[T][C]*/, float* __restrict__ y /*[C][T]*/, int C, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (t < T) y[(size_t)c * T + t] = x[(size_t)t * C + c];
}
// fixed-order block sum (256 threads)
__device__ __forceinline__ float ref_block_sum(float v, float* red) {
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    return red[0];
}
__device__ __forceinline__ float ref_block_max(float v, float* red) {
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
        __syncthreads();
    }
    return red[0];
}
// per-channel mean over time, and (sd != null) sqrt(mean((x - mean)^2) + eps)   (SE :121, ASP :210-214)
__global__ void __launch_bounds__(256) k_ref_stats(const float* __restrict__ x, int ld, int T, float* __restrict__ mean, float* __restrict__ sd,
                                                   float eps) {
    __shared__ float red[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    const float* xr = x + (size_t)c * ld;
    float a = 0.0f;
    for (int t = tid; t < T; t += 256) a += xr[t];
    const float mu = ref_block_sum(a, red) / (float)T;
    if (tid == 0) mean[c] = mu;
    if (sd) {
        float q = 0.0f;
        for (int t = tid; t < T; t += 256) { const float d = xr[t] - mu; q += d * d; }
        const float var = ref_block_sum(q, red) / (float)T;
        if (tid == 0) sd[c] = sqrtf(var + eps);
    }
}
// y[m] = act(b[m] + sum_k W[m][k] x[k]); one wave per row; act 0 none, 2 ReLU, 4 sigmoid
__global__ void __launch_bounds__(256) k_ref_matvec(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ x,
                                                    float* __restrict__ y, int M, int K, int act) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    float a = 0.0f;
    for (int k = lane; k < K; k += 64) a += W[(size_t)m * K + k] * x[k];
    a = wave_sum(a);
    if (lane == 0) {
        a += b ? b[m] : 0.0f;
        if (act == 2) a = fmaxf(a, 0.0f);
        else if (act == 4) a = 1.0f / (1.0f + expf(-a));
        y[m] = a;
    }
}
// SqueezeExcitationRes2NetBlock tail (:181-183): out = y * gate + residual
__global__ void k_ref_se_apply(const float* __restrict__ y, const float* __restrict__ gate, const float* __restrict__ res, int ldr,
                               float* __restrict__ out, int ldo, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (t < T) out[(size_t)c * ldo + t] = y[(size_t)c * T + t] * gate[c] + res[(size_t)c * ldr + t];
}
// rows [C, 2C) <- mean, rows [2C, 3C) <- sd, broadcast over time (:215-217)
__global__ void k_ref_asp_fill(float* __restrict__ a, const float* __restrict__ mean, const float* __restrict__ sd, int C, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (t >= T) return;
    a[((size_t)C + c) * T + t] = mean[c];
    a[((size_t)2 * C + c) * T + t] = sd[c];
}
// softmax over time of the attention logits, then the attention-weighted mean and std of x (:223-232): out[c], out[C + c]
__global__ void __launch_bounds__(256) k_ref_asp_pool(const float* __restrict__ logit, const float* __restrict__ x, int T, int C,
                                                      float* __restrict__ out, float eps) {
    __shared__ float red[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    const float* lr = logit + (size_t)c * T;
    const float* xr = x + (size_t)c * T;
    float mx = -INFINITY;
    for (int t = tid; t < T; t += 256) mx = fmaxf(mx, lr[t]);
    mx = ref_block_max(mx, red);
    float z = 0.0f;
    for (int t = tid; t < T; t += 256) z += expf(lr[t] - mx);
    z = ref_block_sum(z, red);
    float m = 0.0f;
    for (int t = tid; t < T; t += 256) m += (expf(lr[t] - mx) / z) * xr[t];
    m = ref_block_sum(m, red);
    float v = 0.0f;
    for (int t = tid; t < T; t += 256) { const float d = xr[t] - m; v += (expf(lr[t] - mx) / z) * d * d; }
    v = ref_block_sum(v, red);
    if (tid == 0) { out[c] = m; out[C + c] = sqrtf(fmaxf(v, eps)); }
}

// residual VQ encode of one frame per block (Quantization.swift:121-199): r <- z[:, t]; per layer: argmin_c (|e_c|^2 / 2 - r . e_c),
// first index on ties; r -= e_idx.  embT [nq][cd][bins] (coalesced over codes), emb [nq][bins][cd], e2h [nq][bins]
#define REF_VQ_MAXD 1024
__global__ void __launch_bounds__(256) k_ref_rvq(const float* __restrict__ z, int T, const float* __restrict__ embT, const float* __restrict__ emb,
                                                 const float* __restrict__ e2h, int nq, int cd, int bins, int32_t* __restrict__ codes) {
    __shared__ float r[REF_VQ_MAXD];
    __shared__ float bd[256];
    __shared__ int bi[256];
    const int t = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < cd; k += 256) r[k] = z[(size_t)k * T + t];
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
        const float* eT = embT + (size_t)q * cd * bins;
        float best = INFINITY;
        int bidx = 0x7fffffff;
        for (int c = tid; c < bins; c += 256) {
            float dot = 0.0f;
            for (int k = 0; k < cd; ++k) dot += r[k] * eT[(size_t)k * bins + c];
            const float dist = e2h[(size_t)q * bins + c] - dot;
            if (dist < best) { best = dist; bidx = c; }
        }
        bd[tid] = best; bi[tid] = bidx;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                const float d2 = bd[tid + o]; const int i2 = bi[tid + o];
                if (d2 < bd[tid] || (d2 == bd[tid] && i2 < bi[tid])) { bd[tid] = d2; bi[tid] = i2; }
            }
            __syncthreads();
        }
        const int idx = min(bi[0], bins - 1);
        if (tid == 0) codes[(size_t)q * T + t] = idx;
        __syncthreads();
        for (int k = tid; k < cd; k += 256) r[k] -= emb[((size_t)q * bins + idx) * cd + k];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------- handle
mis_q3ref* q3ref_create(const mis_qwen3tts_reference_config* cfg, int device, hipStream_t s) {
    MIS_REQUIRE(cfg, MIS_ERR_INVALID_INPUT, "null reference config");
    mis_q3ref* r = new mis_q3ref();
    r->device = device; r->s = s; r->cfg = *cfg;
    r->has_spk = cfg->spk_n_blocks > 0;
    r->has_enc = cfg->enc_num_filters > 0;
    try {
        if (r->has_spk) {
            MIS_REQUIRE(cfg->spk_n_blocks >= 2 && cfg->spk_n_blocks <= 8, MIS_ERR_INVALID_INPUT, "speaker encoder: 2..8 enc_channels entries");
            MIS_REQUIRE(cfg->spk_mel_dim >= 1 && cfg->spk_mel_dim <= 256 && cfg->spk_enc_dim >= 1 && cfg->spk_res2net_scale >= 2 &&
                            cfg->spk_se_channels >= 1 && cfg->spk_attention_channels >= 1,
                        MIS_ERR_INVALID_INPUT, "bad speaker encoder dims");
            for (int i = 1; i + 1 < cfg->spk_n_blocks; ++i)
                MIS_REQUIRE(cfg->spk_channels[i] == cfg->spk_channels[i - 1] && cfg->spk_channels[i] % cfg->spk_res2net_scale == 0,
                            MIS_ERR_INVALID_INPUT, "speaker encoder: SE-Res2Net blocks need equal channel counts divisible by the scale");
        }
        if (r->has_enc) {
            const int hd = cfg->enc_num_heads > 0 ? cfg->enc_hidden_size / cfg->enc_num_heads : 0;
            MIS_REQUIRE(cfg->enc_n_ratios >= 1 && cfg->enc_n_ratios <= 8 && cfg->enc_audio_channels == 1, MIS_ERR_INVALID_INPUT, "bad SEANet configuration");
            MIS_REQUIRE(hd == 16 || hd == 32 || hd == 64, MIS_ERR_INVALID_INPUT, "tokenizer encoder head_dim must be 16, 32 or 64");
            MIS_REQUIRE(cfg->enc_codebook_dim <= REF_VQ_MAXD && cfg->enc_num_quantizers >= 1 && cfg->enc_valid_num_quantizers >= 1,
                        MIS_ERR_INVALID_INPUT, "bad tokenizer encoder quantizer configuration");
            MIS_REQUIRE(cfg->enc_kernel_size <= 7 && cfg->enc_last_kernel_size <= 7 && cfg->enc_residual_kernel_size <= 7,
                        MIS_ERR_INVALID_INPUT, "SEANet kernel sizes above 7 are unsupported");
            int64_t prod = 1;
            for (int i = 0; i < cfg->enc_n_ratios; ++i) prod *= cfg->enc_upsampling_ratios[i];
            const double enc_rate = (double)cfg->enc_sampling_rate / (double)prod;                 // Qwen3TTSSpeechTokenizer.swift:805-808
            r->ds_stride = std::max(1, (int)(enc_rate / (double)cfg->enc_frame_rate));
        }
    } catch (...) { delete r; throw; }
    return r;
}
void q3ref_destroy(mis_q3ref* r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    delete r;
}
int q3ref_speaker_dim(const mis_q3ref* r) { return r && r->has_spk ? r->cfg.spk_enc_dim : 0; }
bool q3ref_owns(const char* name) { return !strncmp(name, "speaker_encoder.", 16) || !strncmp(name, "encoder_model.", 14); }

void q3ref_set_tensor(mis_q3ref* r, const char* name, const void* data, mis_dtype dtype, const int64_t* shape, int ndim) {
    MIS_REQUIRE(!r->finalized && ndim >= 1 && ndim <= 3, MIS_ERR_INVALID_INPUT, "bad tensor %s", name);
    size_t n = 1;
    std::vector<int64_t> sh;
    for (int i = 0; i < ndim; ++i) { MIS_REQUIRE(shape[i] > 0, MIS_ERR_INVALID_INPUT, "bad shape"); n *= (size_t)shape[i]; sh.push_back(shape[i]); }
    HIP_CHECK(hipSetDevice(r->device));
    const size_t esz = dtype == MIS_F32 ? 4 : 2;
    std::vector<uint8_t> host(n * esz);
    HIP_CHECK(hipMemcpy(host.data(), data, n * esz, hipMemcpyDefault));
    std::vector<float> v(n);
    if (dtype == MIS_F32) memcpy(v.data(), host.data(), n * 4);
    else if (dtype == MIS_BF16) for (size_t i = 0; i < n; ++i) v[i] = bf16_to_f32(((bf16_t*)host.data())[i]);
    else if (dtype == MIS_F16) for (size_t i = 0; i < n; ++i) v[i] = f16_to_f32_host(((uint16_t*)host.data())[i]);
    else throw MisError(MIS_ERR_INVALID_INPUT, "unsupported dtype");
    r->raw[name] = std::move(v);
    r->raw_shape[name] = sh;
}

static const std::vector<float>& rneed(mis_q3ref* r, const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = r->raw.find(name);
    MIS_REQUIRE(it != r->raw.end(), MIS_ERR_NOT_INITIALIZED, "reference front-end weight missing: %s", name.c_str());
    MIS_REQUIRE(r->raw_shape[name] == std::vector<int64_t>(shape), MIS_ERR_INVALID_INPUT, "reference front-end weight %s has the wrong shape", name.c_str());
    return it->second;
}

void q3ref_finalize(mis_q3ref* r) {
    MIS_REQUIRE(!r->finalized, MIS_ERR_INVALID_INPUT, "already finalized");
    HIP_CHECK(hipSetDevice(r->device));
    const mis_qwen3tts_reference_config& cf = r->cfg;
    std::vector<float> arena;
    auto push = [&](const std::vector<float>& v) { size_t o = arena.size(); arena.insert(arena.end(), v.begin(), v.end()); while (arena.size() & 3) arena.push_back(0.f); return o; };
    // conv weight [co][k][ci] (MLX layout) -> A^T [(j ci + c)][co]
    auto conv = [&](const std::string& p, int64_t co, int64_t k, int64_t ci, int dil, bool bias) {
        const auto& w = rneed(r, p + ".weight", {co, k, ci});
        std::vector<float> at((size_t)k * ci * co);
        for (int64_t o = 0; o < co; ++o) for (int64_t j = 0; j < k; ++j) for (int64_t c = 0; c < ci; ++c) at[(j * ci + c) * co + o] = w[(o * k + j) * ci + c];
        mis_q3ref::Lin L; L.M = (int)co; L.K = (int)(k * ci); L.taps = (int)k; L.dil = dil; L.cin = (int)ci;
        L.w = push(at);
        if (bias) L.b = push(rneed(r, p + ".bias", {co}));
        return L;
    };
    // strided conv k = 2 s: [co][2s][ci] -> two taps over the phase-split input: A^T [(j (ci s) + c s + ph)][co] = w[co][j s + ph][c]
    auto conv_strided = [&](const std::string& p, int64_t co, int64_t s, int64_t ci, bool bias) {
        const auto& w = rneed(r, p + ".weight", {co, 2 * s, ci});
        std::vector<float> at((size_t)2 * s * ci * co);
        for (int64_t o = 0; o < co; ++o) for (int64_t j = 0; j < 2; ++j) for (int64_t ph = 0; ph < s; ++ph) for (int64_t c = 0; c < ci; ++c)
            at[((j * ci * s) + c * s + ph) * co + o] = w[(o * 2 * s + j * s + ph) * ci + c];
        mis_q3ref::Lin L; L.M = (int)co; L.K = (int)(2 * s * ci); L.taps = 2; L.dil = 1; L.cin = (int)(ci * s);
        L.w = push(at);
        if (bias) L.b = push(rneed(r, p + ".bias", {co}));
        return L;
    };
    auto matvec = [&](const std::string& p, int64_t co, int64_t ci) {      // 1x1 conv kept row-major [co][ci] for k_ref_matvec
        mis_q3ref::Lin L; L.M = (int)co; L.K = (int)ci;
        L.w = push(rneed(r, p + ".weight", {co, 1, ci}));
        L.b = push(rneed(r, p + ".bias", {co}));
        return L;
    };
    auto linear = [&](const std::vector<float>& w, int64_t out_f, int64_t in_f) {           // [out][in] -> A^T [in][out]
        std::vector<float> at((size_t)in_f * out_f);
        for (int64_t o = 0; o < out_f; ++o) for (int64_t i = 0; i < in_f; ++i) at[i * out_f + o] = w[o * in_f + i];
        mis_q3ref::Lin L; L.M = (int)out_f; L.K = (int)in_f; L.cin = (int)in_f;
        L.w = push(at);
        return L;
    };
    if (r->has_spk) {
        const std::string P = "speaker_encoder.";
        const int nb = cf.spk_n_blocks;
        const int* ch = cf.spk_channels; const int* ks = cf.spk_kernel_sizes; const int* dl = cf.spk_dilations;
        for (int i = 0; i < nb; ++i)
            MIS_REQUIRE(ks[i] >= 1 && ks[i] <= 7 && (ks[i] & 1) && dl[i] >= 1 && (ks[i] - 1) * dl[i] <= 88, MIS_ERR_INVALID_INPUT,
                        "speaker encoder: kernel sizes must be odd and <= 7");
        r->spk_first = conv(P + "blocks.0.conv", ch[0], ks[0], cf.spk_mel_dim, dl[0], true);
        r->spk_blocks.clear();
        int cat = 0;
        for (int i = 1; i + 1 < nb; ++i) {
            const std::string p = P + "blocks." + std::to_string(i);
            mis_q3ref::SeBlock B;
            const int w = ch[i] / cf.spk_res2net_scale;
            B.tdnn1 = conv(p + ".tdnn1.conv", ch[i], 1, ch[i - 1], 1, true);
            for (int j = 0; j + 1 < cf.spk_res2net_scale; ++j)
                B.res.push_back(conv(p + ".res2net_block.blocks." + std::to_string(j) + ".conv", w, ks[i], w, dl[i], true));
            B.tdnn2 = conv(p + ".tdnn2.conv", ch[i], 1, ch[i], 1, true);
            B.se1 = matvec(p + ".se_block.conv1", cf.spk_se_channels, ch[i]);
            B.se2 = matvec(p + ".se_block.conv2", ch[i], cf.spk_se_channels);
            r->spk_blocks.push_back(B);
            cat += ch[i];
        }
        if (cat == 0) cat = ch[0];
        MIS_REQUIRE(cat == ch[nb - 1], MIS_ERR_INVALID_INPUT, "speaker encoder: mfa input (%d) != concatenated block channels (%d)", ch[nb - 1], cat);
        r->spk_mfa = conv(P + "mfa.conv", ch[nb - 1], ks[nb - 1], ch[nb - 1], dl[nb - 1], true);
        r->asp_tdnn = conv(P + "asp.tdnn.conv", cf.spk_attention_channels, 1, 3 * ch[nb - 1], 1, true);
        r->asp_conv = conv(P + "asp.conv", ch[nb - 1], 1, cf.spk_attention_channels, 1, true);
        r->spk_fc = matvec(P + "fc", cf.spk_enc_dim, 2 * ch[nb - 1]);
    }
    if (r->has_enc) {
        const std::string P = "encoder_model.";
        const int nf = cf.enc_num_filters;
        r->enc_init = conv(P + "encoder.init_conv1d.conv.conv", nf, cf.enc_kernel_size, cf.enc_audio_channels, 1, true);
        r->enc_layers.clear();
        int mult = 1;
        for (int li = 0; li < cf.enc_n_ratios; ++li) {
            const int ratio = cf.enc_upsampling_ratios[cf.enc_n_ratios - 1 - li];                      // reversed (Seanet.swift:222)
            const int dim = mult * nf;
            const std::string p = P + "encoder.layers." + std::to_string(li);
            mis_q3ref::EncLayer L;
            L.stride = ratio; L.cin = dim;
            int dil = 1;
            for (int ri = 0; ri < cf.enc_num_residual_layers; ++ri) {
                const std::string q = p + ".residuals." + std::to_string(ri);
                mis_q3ref::Res R;
                MIS_REQUIRE((cf.enc_residual_kernel_size - 1) * dil <= 88, MIS_ERR_INVALID_INPUT, "SEANet residual dilation too large");
                R.c1 = conv(q + ".block.0.conv.conv", dim / cf.enc_compress, cf.enc_residual_kernel_size, dim, dil, true);
                R.c2 = conv(q + ".block.1.conv.conv", dim, 1, dim / cf.enc_compress, 1, true);
                R.has_sc = cf.enc_use_conv_shortcut != 0;
                if (R.has_sc) R.sc = conv(q + ".shortcut.conv.conv", dim, 1, dim, 1, true);
                L.res.push_back(R);
                dil *= cf.enc_dilation_growth_rate;
            }
            L.down = conv_strided(p + ".downsample.conv.conv", 2 * dim, ratio, dim, true);
            r->enc_layers.push_back(L);
            mult *= 2;
        }
        const int D = cf.enc_hidden_size, H = cf.enc_num_heads, hd = D / H, I = cf.enc_intermediate_size;
        r->enc_final = conv(P + "encoder.final_conv1d.conv.conv", D, cf.enc_last_kernel_size, mult * nf, 1, true);
        r->tlayers.clear();
        for (int li = 0; li < cf.enc_num_layers; ++li) {
            const std::string p = P + "encoder_transformer.transformer.layers." + std::to_string(li);
            mis_q3ref::TL L{};
            L.n1w = push(rneed(r, p + ".norm1.weight", {D})); L.n1b = push(rneed(r, p + ".norm1.bias", {D}));
            L.n2w = push(rneed(r, p + ".norm2.weight", {D})); L.n2b = push(rneed(r, p + ".norm2.bias", {D}));
            L.ls1 = push(rneed(r, p + ".layer_scale_1.scale", {D})); L.ls2 = push(rneed(r, p + ".layer_scale_2.scale", {D}));
            {   // q / k rows of every head reordered evens-then-odds: interleaved RoPE pairs (2i, 2i+1) become rotate-half pairs (i, i + hd/2)
                const auto& w = rneed(r, p + ".self_attn.in_proj.weight", {3 * D, D});
                std::vector<float> pw(w.size());
                for (int part = 0; part < 3; ++part)
                    for (int h = 0; h < H; ++h)
                        for (int i = 0; i < hd; ++i) {
                            const int src = part < 2 ? (i < hd / 2 ? 2 * i : 2 * (i - hd / 2) + 1) : i;
                            memcpy(&pw[((size_t)part * D + (size_t)h * hd + i) * D], &w[((size_t)part * D + (size_t)h * hd + src) * D], (size_t)D * 4);
                        }
                L.qkv = linear(pw, 3 * D, D);
            }
            L.o = linear(rneed(r, p + ".self_attn.out_proj.weight", {D, D}), D, D);
            L.f1 = linear(rneed(r, p + ".gating.linear1.weight", {I, D}), I, D);
            L.f2 = linear(rneed(r, p + ".gating.linear2.weight", {D, I}), D, I);
            r->tlayers.push_back(L);
        }
        r->enc_down = conv_strided(P + "downsample.conv.conv.conv", D, r->ds_stride, D, false);
        const int cd = cf.enc_codebook_dim, bins = cf.enc_codebook_size;
        const int keep = std::min(cf.enc_valid_num_quantizers, cf.enc_num_quantizers);
        for (int g = 0; g < 2; ++g) {
            const std::string p = P + "quantizer." + (g == 0 ? "rvq_first" : "rvq_rest");
            mis_q3ref::VqGroup& V = r->vq[g];
            V.nq = g == 0 ? 1 : std::max(0, keep - 1);          // layers beyond the kept ones never influence the kept codes
            if (V.nq == 0) continue;
            {
                const auto& w = rneed(r, p + ".input_proj.weight", {cd, 1, D});
                V.in_proj = linear(w, cd, D);
            }
            std::vector<float> eT((size_t)V.nq * cd * bins), em((size_t)V.nq * bins * cd), e2((size_t)V.nq * bins);
            for (int q = 0; q < V.nq; ++q) {
                const std::string cbk = p + ".vq.layers." + std::to_string(q) + ".codebook";
                const auto& es = rneed(r, cbk + ".embedding_sum", {bins, cd});
                const auto& cu = rneed(r, cbk + ".cluster_usage", {bins});
                for (int v = 0; v < bins; ++v) {
                    const float den = std::max(cu[v], 1e-5f);                                    // Quantization.swift:24-27
                    float n2 = 0.0f;
                    for (int k = 0; k < cd; ++k) {
                        const float e = es[(size_t)v * cd + k] / den;
                        em[((size_t)q * bins + v) * cd + k] = e;
                        eT[((size_t)q * cd + k) * bins + v] = e;
                        n2 += e * e;
                    }
                    e2[(size_t)q * bins + v] = n2 / 2.0f;
                }
            }
            V.embT = push(eT); V.emb = push(em); V.e2h = push(e2);
        }
    }
    r->arena.alloc(std::max<size_t>(arena.size(), 4));
    HIP_CHECK(hipMemcpy(r->arena.p, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    r->raw.clear(); r->raw_shape.clear();
    r->finalized = true;
}

// ---------------------------------------------------------------------------- launch helpers
static void ref_conv(const mis_q3ref* r, const mis_q3ref::Lin& L, int mode, const float* X, int ldx, int Tin, int pad, float* Y, int ldy, int N,
                     const float* R = nullptr, const float* scale = nullptr) {
    const float* W = r->arena.p;
    GemmParams g{};
    g.AT = W + L.w; g.bias = L.b == (size_t)-1 ? nullptr : W + L.b; g.X = X; g.Y = Y; g.R = R; g.scale = scale;
    g.M = L.M; g.K = L.K; g.N = N; g.Tin = Tin; g.Tout = N; g.ldx = ldx; g.ldy = ldy;
    g.Cin = L.cin; g.taps = L.taps; g.dil = L.dil; g.pad = pad;
    launch_gemm(mode, false, g, 1, r->s);
}
static void ref_act(const mis_q3ref* r, const float* x, int ldx, float* y, int ldy, int C, int T, int act) {
    hipLaunchKernelGGL(k_ref_act, dim3(cdiv(T, 256), C), dim3(256), 0, r->s, x, ldx, y, ldy, T, act);
}
static void ref_tap_out(const mis_q3ref* r, const float* src, int C, int64_t T, float* out, int64_t capacity, int* oC, int64_t* oT) {
    if (oC) *oC = C;
    if (oT) *oT = T;
    if (!out) return;
    MIS_REQUIRE((int64_t)C * T <= capacity, MIS_ERR_INVALID_INPUT, "tap buffer too small (%lld floats needed)", (long long)((int64_t)C * T));
    HIP_CHECK(hipMemcpyAsync(out, src, (size_t)C * T * 4, hipMemcpyDefault, r->s));
    HIP_CHECK(hipStreamSynchronize(r->s));
}

// ---------------------------------------------------------------------------- speaker encoder
// TimeDelayNetBlock (:20-42): reflect pad (k-1) d / 2, conv, ReLU.  x rows [cin][T] (+ add), y rows [M][T]; pad_buf >= cin (T + 2 pad)
static void ref_tdnn(const mis_q3ref* r, const mis_q3ref::Lin& L, const float* x, int ldx, const float* add, int lda, float* y, int ldy, int T,
                     float* pad_buf) {
    const int pad = (L.taps - 1) * L.dil / 2;
    if (L.taps > 1 || add) {
        const int Tp = T + 2 * pad;
        hipLaunchKernelGGL(k_ref_reflect, dim3(cdiv(Tp, 256), L.cin), dim3(256), 0, r->s, x, ldx, add, lda, pad_buf, Tp, T, pad);
        ref_conv(r, L, GEMM_TAPS, pad_buf, Tp, Tp, 0, y, ldy, T);
    } else
        ref_conv(r, L, GEMM_PLAIN, x, ldx, T, 0, y, ldy, T);
    ref_act(r, y, ldy, y, ldy, L.M, T, 2);
}

void q3ref_speaker(mis_q3ref* r, const float* audio, int64_t n, int stage, float* out, int64_t capacity, int* oC, int64_t* oT) {
    MIS_REQUIRE(r && r->finalized && r->has_spk, MIS_ERR_NOT_INITIALIZED, "this model has no speaker encoder (tts_model_type != base)");
    MIS_REQUIRE(audio && n >= 1, MIS_ERR_INVALID_INPUT, "empty reference audio");
    HIP_CHECK(hipSetDevice(r->device));
    const mis_qwen3tts_reference_config& cf = r->cfg;
    hipStream_t s = r->s;
    mis_mel_config mc{};
    mc.sample_rate = cf.spk_sample_rate; mc.n_fft = 1024; mc.hop_length = 256; mc.n_mels = cf.spk_mel_dim;
    mc.window = 1; mc.mel_scale = 0; mc.slaney_norm = 1; mc.drop_last_frame = 0;           // computeMelSpectrogram defaults (DSP.swift:230-273)
    const int T = (int)mis_mel_num_frames(&mc, n);
    int max_pad = 0;                                                            // reflectPad1D needs pad < frames (:6-16 clamps; here: refuse)
    for (int i = 0; i < cf.spk_n_blocks; ++i) max_pad = std::max(max_pad, (cf.spk_kernel_sizes[i] - 1) * cf.spk_dilations[i] / 2);
    MIS_REQUIRE(T >= 16 && T > max_pad, MIS_ERR_INVALID_INPUT, "reference audio too short for the speaker encoder (%d mel frames)", T);
    const int nb = cf.spk_n_blocks, Cl = cf.spk_channels[nb - 1];
    int Cmax = std::max(std::max(cf.spk_mel_dim, 3 * Cl), std::max(cf.spk_se_channels, cf.spk_attention_channels));
    for (int i = 0; i < nb; ++i) Cmax = std::max(Cmax, cf.spk_channels[i]);
    DevBuf<float> ain, melTC, x0, cat, y1, y2, y3, padb, a3, vec;
    ain.alloc(n); melTC.alloc((size_t)T * cf.spk_mel_dim); x0.alloc((size_t)Cmax * T); cat.alloc((size_t)Cl * T);
    y1.alloc((size_t)Cmax * T); y2.alloc((size_t)Cmax * T); y3.alloc((size_t)Cmax * T); padb.alloc((size_t)Cmax * (T + 96)); a3.alloc((size_t)3 * Cl * T);
    vec.alloc((size_t)8 * Cmax + cf.spk_enc_dim);
    HIP_CHECK(hipMemcpyAsync(ain.p, audio, (size_t)n * 4, hipMemcpyDefault, s));
    mel_spectrogram_device(r->device, mc, ain.p, 1, n, melTC.p, s);
    float* mel = y1.p;                                                                          // [mel_dim][T]
    hipLaunchKernelGGL(k_ref_transpose, dim3(cdiv(T, 256), cf.spk_mel_dim), dim3(256), 0, s, melTC.p, mel, cf.spk_mel_dim, T);
    ref_tdnn(r, r->spk_first, mel, T, nullptr, 0, x0.p, T, T, padb.p);
    if (stage == 0) { ref_tap_out(r, x0.p, cf.spk_channels[0], T, out, capacity, oC, oT); return; }
    const float* xin = x0.p;
    int cat_off = 0;
    float* mean = vec.p; float* sd = vec.p + Cmax; float* se_h = vec.p + 2 * Cmax; float* gate = vec.p + 3 * Cmax;
    float* pooled = vec.p + 4 * Cmax; float* xvec = vec.p + 8 * Cmax;
    for (size_t bi = 0; bi < r->spk_blocks.size(); ++bi) {
        const mis_q3ref::SeBlock& B = r->spk_blocks[bi];
        const int C = B.tdnn1.M, w = C / cf.spk_res2net_scale;
        ref_tdnn(r, B.tdnn1, xin, T, nullptr, 0, y1.p, T, T, padb.p);
        // Res2NetBlock (:73-95): chunk 0 passes through; chunk i >= 1 = tdnn(chunk_i (+ previous output for i >= 2))
        HIP_CHECK(hipMemcpyAsync(y2.p, y1.p, (size_t)w * T * 4, hipMemcpyDeviceToDevice, s));
        for (int i = 1; i < cf.spk_res2net_scale; ++i)
            ref_tdnn(r, B.res[i - 1], y1.p + (size_t)i * w * T, T, i >= 2 ? y2.p + (size_t)(i - 1) * w * T : nullptr, T, y2.p + (size_t)i * w * T, T, T, padb.p);
        ref_tdnn(r, B.tdnn2, y2.p, T, nullptr, 0, y3.p, T, T, padb.p);
        // SqueezeExcitationBlock (:121-128)
        hipLaunchKernelGGL(k_ref_stats, dim3(C), dim3(256), 0, s, y3.p, T, T, mean, (float*)nullptr, 0.0f);
        const float* W = r->arena.p;
        hipLaunchKernelGGL(k_ref_matvec, dim3(cdiv(B.se1.M, 4)), dim3(256), 0, s, W + B.se1.w, W + B.se1.b, mean, se_h, B.se1.M, B.se1.K, 2);
        hipLaunchKernelGGL(k_ref_matvec, dim3(cdiv(B.se2.M, 4)), dim3(256), 0, s, W + B.se2.w, W + B.se2.b, se_h, gate, B.se2.M, B.se2.K, 4);
        float* dst = cat.p + (size_t)cat_off * T;
        hipLaunchKernelGGL(k_ref_se_apply, dim3(cdiv(T, 256), C), dim3(256), 0, s, y3.p, gate, xin, T, dst, T, T);
        if (stage == (int)bi + 1) { ref_tap_out(r, dst, C, T, out, capacity, oC, oT); return; }
        xin = dst; cat_off += C;
    }
    const float* mfa_in = r->spk_blocks.empty() ? x0.p : cat.p;
    ref_tdnn(r, r->spk_mfa, mfa_in, T, nullptr, 0, a3.p, T, T, padb.p);                         // rows [0, Cl) of the ASP input
    if (stage == nb - 1) { ref_tap_out(r, a3.p, Cl, T, out, capacity, oC, oT); return; }
    // AttentiveStatisticsPooling (:209-232)
    hipLaunchKernelGGL(k_ref_stats, dim3(Cl), dim3(256), 0, s, a3.p, T, T, mean, sd, 1e-12f);
    hipLaunchKernelGGL(k_ref_asp_fill, dim3(cdiv(T, 256), Cl), dim3(256), 0, s, a3.p, mean, sd, Cl, T);
    ref_conv(r, r->asp_tdnn, GEMM_PLAIN, a3.p, T, T, 0, y1.p, T, T);
    ref_act(r, y1.p, T, y1.p, T, r->asp_tdnn.M, T, 3);                                         // ReLU (TDNN) then tanh
    ref_conv(r, r->asp_conv, GEMM_PLAIN, y1.p, T, T, 0, y2.p, T, T);
    hipLaunchKernelGGL(k_ref_asp_pool, dim3(Cl), dim3(256), 0, s, y2.p, a3.p, T, Cl, pooled, 1e-12f);
    if (stage == nb) { ref_tap_out(r, pooled, 2 * Cl, 1, out, capacity, oC, oT); return; }
    {
        const float* W = r->arena.p;
        hipLaunchKernelGGL(k_ref_matvec, dim3(cdiv(r->spk_fc.M, 4)), dim3(256), 0, s, W + r->spk_fc.w, W + r->spk_fc.b, pooled, xvec, r->spk_fc.M,
                           r->spk_fc.K, 0);
    }
    HIP_CHECK(hipGetLastError());
    MIS_REQUIRE(stage < 0, MIS_ERR_INVALID_INPUT, "unknown speaker-encoder tap %d", stage);
    ref_tap_out(r, xvec, cf.spk_enc_dim, 1, out, capacity, oC, oT);
    HIP_CHECK(hipStreamSynchronize(s));
}

// ---------------------------------------------------------------------------- speech-tokenizer encoder
void q3ref_encode(mis_q3ref* r, const float* audio, int64_t n, int stage, float* out, int64_t capacity, int* oC, int64_t* oT,
                  std::vector<int32_t>* codes, int* n_q) {
    MIS_REQUIRE(r && r->finalized && r->has_enc, MIS_ERR_NOT_INITIALIZED, "this speech tokenizer has no encoder (encoder_config missing)");
    MIS_REQUIRE(audio && n >= 1 && n <= ((int64_t)1 << 25), MIS_ERR_INVALID_INPUT, "bad reference audio length (1 .. 2^25 samples)");
    HIP_CHECK(hipSetDevice(r->device));
    const mis_qwen3tts_reference_config& cf = r->cfg;
    hipStream_t s = r->s;
    const float* W = r->arena.p;
    const int D = cf.enc_hidden_size, H = cf.enc_num_heads, hd = D / H, I = cf.enc_intermediate_size;
    // buffer capacity: the largest [C][T] of any stage (phase-split copies carry one extra column per row)
    size_t cap = (size_t)cf.enc_num_filters * n;
    {
        int64_t T = n; int C = cf.enc_num_filters;
        for (auto& L : r->enc_layers) {
            const int64_t M = (T + L.stride - 1) / L.stride;
            cap = std::max(cap, (size_t)C * L.stride * (size_t)(M + 1));
            C *= 2; T = M;
            cap = std::max(cap, (size_t)C * (size_t)T);
        }
        cap = std::max(cap, (size_t)std::max(3 * D, I) * (size_t)(T + 2));
    }
    DevBuf<float> ain, b0, b1, b2;
    ain.alloc(n); b0.alloc(cap); b1.alloc(cap); b2.alloc(cap);
    HIP_CHECK(hipMemcpyAsync(ain.p, audio, (size_t)n * 4, hipMemcpyDefault, s));
    float *x = b0.p, *f1 = b1.p, *f2 = b2.p;
    const bool causal = cf.enc_use_causal_conv != 0;
    auto left_pad = [&](const mis_q3ref::Lin& L) { const int pt = (L.taps - 1) * L.dil; return causal ? pt : pt - pt / 2; };   // Conv.swift:212-221
    int T = (int)n, C = cf.enc_num_filters;
    // SeanetEncoder (Seanet.swift:203-258)
    ref_conv(r, r->enc_init, GEMM_TAPS, ain.p, T, T, left_pad(r->enc_init), x, T, T);
    for (auto& L : r->enc_layers) {
        for (auto& R : L.res) {
            ref_act(r, x, T, f1, T, C, T, 1);
            ref_conv(r, R.c1, GEMM_TAPS, f1, T, T, left_pad(R.c1), f2, T, T);
            ref_act(r, f2, T, f2, T, R.c1.M, T, 1);
            const float* skip = x;
            if (R.has_sc) { ref_conv(r, R.sc, GEMM_PLAIN, x, T, T, 0, f1, T, T); skip = f1; }
            // the output may not alias the residual operand of another column tile: write to the third buffer
            float* dst = (skip == f1) ? x : f1;
            ref_conv(r, R.c2, GEMM_RESID, f2, T, T, 0, dst, T, T, skip);
            if (dst != x) std::swap(x, f1);
        }
        const int M = (T + L.stride - 1) / L.stride;
        hipLaunchKernelGGL(k_ref_phase_split, dim3(cdiv((int64_t)(M + 1) * L.stride, 256), C), dim3(256), 0, s, x, f1, T, L.stride, M + 1, 0, 1);
        ref_conv(r, L.down, GEMM_TAPS, f1, M + 1, M + 1, 0, f2, M, M);
        std::swap(x, f2);
        T = M; C *= 2;
    }
    ref_act(r, x, T, f1, T, C, T, 1);
    ref_conv(r, r->enc_final, GEMM_TAPS, f1, T, T, left_pad(r->enc_final), f2, T, T);
    std::swap(x, f2);
    if (stage == 0) { ref_tap_out(r, x, D, T, out, capacity, oC, oT); return; }
    // ProjectedTransformer (Transformer.swift:121-369): pre-norm, causal attention with RoPE, GELU MLP, per-channel layer scale
    for (auto& L : r->tlayers) {
        launch_q3_norm_ct(x, f1, W + L.n1w, W + L.n1b, 1, D, T, T, cf.enc_norm_eps, 0, s);
        ref_conv(r, L.qkv, GEMM_PLAIN, f1, T, T, 0, f2, T, T);
        Q3AttnArgs aa{};
        aa.q = f2; aa.q_bs = (int64_t)3 * D * T; aa.q_ld = T;
        aa.k = f2 + (size_t)D * T; aa.v = f2 + (size_t)2 * D * T; aa.kv_bs = aa.q_bs; aa.kv_ld = T;
        aa.out = f1; aa.o_bs = (int64_t)D * T; aa.o_ld = T;
        aa.H = H; aa.Hkv = H; aa.Tq = T; aa.pos0 = 0; aa.theta = cf.enc_rope_theta; aa.scale = 1.0f / sqrtf((float)hd);
        launch_q3_attn(aa, hd, 1, s);
        ref_conv(r, L.o, GEMM_RESID, f1, T, T, 0, f2, T, T, x, W + L.ls1);
        std::swap(x, f2);
        launch_q3_norm_ct(x, f1, W + L.n2w, W + L.n2b, 1, D, T, T, cf.enc_norm_eps, 0, s);
        ref_conv(r, L.f1, GEMM_GELU, f1, T, T, 0, f2, T, T);
        ref_conv(r, L.f2, GEMM_RESID, f2, T, T, 0, f1, T, T, x, W + L.ls2);
        std::swap(x, f1);
    }
    if (stage == 1) { ref_tap_out(r, x, D, T, out, capacity, oC, oT); return; }
    // ConvDownsample1d (Conv.swift:346-359): k = 2 stride, edge padding, no bias
    {
        const int st = r->ds_stride, M = (T + st - 1) / st;
        hipLaunchKernelGGL(k_ref_phase_split, dim3(cdiv((int64_t)(M + 1) * st, 256), D), dim3(256), 0, s, x, f1, T, st, M + 1, 1, 0);
        ref_conv(r, r->enc_down, GEMM_TAPS, f1, M + 1, M + 1, 0, f2, M, M);
        std::swap(x, f2);
        T = M;
    }
    if (stage == 2) { ref_tap_out(r, x, D, T, out, capacity, oC, oT); return; }
    MIS_REQUIRE(stage < 0, MIS_ERR_INVALID_INPUT, "unknown tokenizer-encoder tap %d", stage);
    // SplitResidualVectorQuantizer.encode (Quantization.swift:121-199)
    const int cd = cf.enc_codebook_dim, bins = cf.enc_codebook_size;
    const int keep = r->vq[0].nq + r->vq[1].nq;
    DevBuf<int32_t> cdev;
    cdev.alloc((size_t)keep * T);
    int row = 0;
    for (int g = 0; g < 2; ++g) {
        const mis_q3ref::VqGroup& V = r->vq[g];
        if (V.nq == 0) continue;
        ref_conv(r, V.in_proj, GEMM_PLAIN, x, T, T, 0, f1, T, T);
        hipLaunchKernelGGL(k_ref_rvq, dim3(T), dim3(256), 0, s, f1, T, W + V.embT, W + V.emb, W + V.e2h, V.nq, cd, bins, cdev.p + (size_t)row * T);
        row += V.nq;
    }
    HIP_CHECK(hipGetLastError());
    if (codes) {
        codes->resize((size_t)keep * T);
        HIP_CHECK(hipMemcpyAsync(codes->data(), cdev.p, (size_t)keep * T * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));
    if (n_q) *n_q = keep;
    if (oC) *oC = keep;
    if (oT) *oT = T;
}
