// whisper_engine.hip - host side of the Whisper STT engine: weights, encoder pass, cached cross K/V, the
// per-token decoder chain (weight-streaming kernels of lm_kernels.hip) and the greedy generate loop.
//
// Reference being replaced: WhisperModel (Sources/MLXAudioSTT/Models/Whisper/WhisperModel.swift:36-309),
// WhisperEncoder / WhisperDecoder (WhisperLayers.swift:110-328).  Tokenisation, prompt construction
// (buildPromptTokens) and text decoding stay on the host side, as in the reference (WhisperTokenizer.swift).
#include "common.h"
#include "kernels.h"
#include "lm_kernels.h"
#include "whisper_kernels.h"

#include <math.h>
#include <string.h>
#include <algorithm>
#include <memory>

struct WTensor {
    DevBuf<bf16_t> buf;
    std::vector<int64_t> shape;
};

struct EncLayer { bf16_t *wqkv, *bqkv, *wo, *bo, *fc1, *b1, *fc2, *b2, *ln1w, *ln1b, *ln2w, *ln2b; };
struct DecLayer {
    bf16_t *sqkv, *sbqkv, *so, *sbo, *cq, *cbq, *ckv, *cbkv, *co, *cbo, *fc1, *b1, *fc2, *b2;
    bf16_t *ln1w, *ln1b, *ln2w, *ln2b, *ln3w, *ln3b;
};

struct mis_whisper {
    int device = 0;
    hipStream_t stream = nullptr;
    mis_whisper_config cfg{};
    int d = 0, He = 0, Hd = 0, D = 0, V = 0, Vpad = 0, nmel = 0, K1 = 0;
    std::map<std::string, std::unique_ptr<WTensor>> raw;
    bool finalized = false;
    DevBuf<bf16_t> arena;                    // all assembled weights
    bf16_t *conv1w = nullptr, *conv1b = nullptr, *conv2w = nullptr, *conv2b = nullptr, *enc_pos = nullptr, *enc_lnw = nullptr,
           *enc_lnb = nullptr, *emb = nullptr, *emb_packed = nullptr, *dec_pos = nullptr, *dec_lnw = nullptr, *dec_lnb = nullptr;
    std::vector<EncLayer> enc;
    std::vector<DecLayer> dec;
    // state
    int batch = 0, Mpad = 0, Smax = 0, Spad = 1536;
    int S_qkv = 1, S_o = 1, S_cq = 1, S_fc2 = 1;
    DevBuf<bf16_t> cross_k, cross_v, self_k, self_v, enc_out;
    DevBuf<int32_t> ids, pos_cur, pos_next, n_gen, tokens_out, next_ids, done_count, sup, bsup;
    DevBuf<uint8_t> active;
    DevBuf<bf16_t> h, h2, x, attn_out, act, logits;      // h2: the other residual buffer of the folded glue (enqueue_decoder_step)
    DevBuf<float> qkv_part, part, e_buf, logits_f32;
    DevBuf<SamplerScratch> scratch;
    bool shared_device = false;      // another replica's streams run on this device (group.hip): never a kernel that waits for co-resident blocks
};

static const float LN_EPS = 1e-5f;

extern "C" mis_status mis_whisper_create(const mis_whisper_config* cfg, int device, mis_whisper** out) {
    MIS_API_BEGIN
    MIS_REQUIRE(cfg && out, MIS_ERR_INVALID_INPUT, "null argument");
    int n = 0;
    HIP_CHECK(hipGetDeviceCount(&n));
    MIS_REQUIRE(device >= 0 && device < n, MIS_ERR_DEVICE, "device %d not available (%d GPUs visible)", device, n);
    const int d = cfg->d_model;
    MIS_REQUIRE(d > 0 && cfg->encoder_layers > 0 && cfg->decoder_layers > 0 && cfg->vocab_size > 0, MIS_ERR_INVALID_INPUT, "bad dims");
    MIS_REQUIRE(cfg->encoder_attention_heads > 0 && d % cfg->encoder_attention_heads == 0 &&
                    cfg->decoder_attention_heads == cfg->encoder_attention_heads, MIS_ERR_INVALID_INPUT, "bad head counts");
    const int D = d / cfg->encoder_attention_heads;
    MIS_REQUIRE(D == 64 || D == 128, MIS_ERR_INVALID_INPUT, "head_dim %d unsupported (64 or 128)", D);
    MIS_REQUIRE(d % 32 == 0 && cfg->encoder_ffn_dim % 32 == 0 && cfg->decoder_ffn_dim % 32 == 0, MIS_ERR_INVALID_INPUT,
                "d_model / ffn dims must be multiples of 32");
    MIS_REQUIRE(cfg->num_mel_bins == 80 || cfg->num_mel_bins == 128, MIS_ERR_INVALID_INPUT, "num_mel_bins must be 80 or 128");
    MIS_REQUIRE(cfg->max_source_positions == 1500, MIS_ERR_INVALID_INPUT, "max_source_positions must be 1500");
    HIP_CHECK(hipSetDevice(device));
    mis_whisper* c = new mis_whisper();
    c->device = device; c->cfg = *cfg;
    c->d = d; c->He = cfg->encoder_attention_heads; c->Hd = cfg->decoder_attention_heads; c->D = D;
    c->V = cfg->vocab_size; c->Vpad = (int)round_up(c->V, 16); c->nmel = cfg->num_mel_bins;
    c->K1 = (int)round_up(3 * c->nmel, 32);
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    *out = c;
    MIS_API_END
}

extern "C" void mis_whisper_destroy(mis_whisper* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    delete c;
}

// mlx-whisper checkpoint keys ("encoder.blocks.N.attn.query.weight", ...) -> the HF names the engine indexes by
// (WhisperModel.remapMlxWhisperKey / remapBlockSuffix / remapAttnSuffix, WhisperModel.swift:393-478).  false = not an mlx key.
static bool whisper_remap_mlx_key(const std::string& raw, std::string* out) {
    auto starts = [&](const std::string& s, const char* p) { return s.rfind(p, 0) == 0; };
    if (raw == "encoder.positional_embedding") { *out = "model.encoder.embed_positions.weight"; return true; }
    if (raw == "decoder.positional_embedding") { *out = "model.decoder.embed_positions.weight"; return true; }
    if (starts(raw, "decoder.token_embedding.")) { *out = "model.decoder.embed_tokens." + raw.substr(strlen("decoder.token_embedding.")); return true; }
    if (starts(raw, "encoder.ln_post.")) { *out = "model.encoder.layer_norm." + raw.substr(strlen("encoder.ln_post.")); return true; }
    if (starts(raw, "decoder.ln.")) { *out = "model.decoder.layer_norm." + raw.substr(strlen("decoder.ln.")); return true; }
    for (const char* stem : {"encoder", "decoder"}) {
        const std::string pre = std::string(stem) + ".blocks.";
        if (!starts(raw, pre.c_str())) continue;
        const std::string rest = raw.substr(pre.size());
        const size_t dot = rest.find('.');
        if (dot == std::string::npos) return false;
        const std::string idx = rest.substr(0, dot), suf = rest.substr(dot + 1);
        const bool dec = std::string(stem) == "decoder";
        auto attn = [&](const std::string& s2, const char* container, std::string* m) {
            const size_t d2 = s2.find('.');
            if (d2 == std::string::npos) return false;
            const std::string proj = s2.substr(0, d2), tail = s2.substr(d2 + 1);
            const char* mp = proj == "query" ? "q_proj" : proj == "key" ? "k_proj" : proj == "value" ? "v_proj" : proj == "out" ? "out_proj" : nullptr;
            if (!mp) return false;
            *m = std::string(container) + "." + mp + "." + tail;
            return true;
        };
        std::string m;
        if (starts(suf, "attn_ln.")) m = "self_attn_layer_norm." + suf.substr(8);
        else if (dec && starts(suf, "cross_attn_ln.")) m = "encoder_attn_layer_norm." + suf.substr(14);
        else if (starts(suf, "mlp_ln.")) m = "final_layer_norm." + suf.substr(7);
        else if (starts(suf, "mlp1.")) m = "fc1." + suf.substr(5);
        else if (starts(suf, "mlp2.")) m = "fc2." + suf.substr(5);
        else if (starts(suf, "attn.")) { if (!attn(suf.substr(5), "self_attn", &m)) return false; }
        else if (dec && starts(suf, "cross_attn.")) { if (!attn(suf.substr(11), "encoder_attn", &m)) return false; }
        else return false;
        *out = std::string("model.") + stem + ".layers." + idx + "." + m;
        return true;
    }
    return false;
}

extern "C" mis_status mis_whisper_set_tensor(mis_whisper* c, const char* name_, const void* data, mis_dtype dtype,
                                             const int64_t* shape, int ndim) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && name_ && data && shape && ndim >= 1 && ndim <= 3, MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(!c->finalized, MIS_ERR_INVALID_INPUT, "set_tensor after finalize");
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_F16 || dtype == MIS_BF16, MIS_ERR_INVALID_INPUT, "unsupported dtype");
    std::string name = name_;
    if (name == "proj_out.weight" || name == "model.proj_out.weight") return MIS_OK;          // tied, WhisperModel.swift:343-346
    if (name == "alignment_heads") return MIS_OK;                                              // mlx-whisper extra (:371)
    {   // mlx-whisper key layout -> HF names (WhisperModel.remapMlxWhisperKey / remapBlockSuffix, :393-478)
        std::string mapped;
        if (whisper_remap_mlx_key(name, &mapped)) name = mapped;
    }
    if (name.rfind("model.", 0) != 0 && (name.rfind("encoder.", 0) == 0 || name.rfind("decoder.", 0) == 0)) name = "model." + name;
    HIP_CHECK(hipSetDevice(c->device));
    size_t n = 1;
    auto t = std::make_unique<WTensor>();
    for (int i = 0; i < ndim; ++i) { MIS_REQUIRE(shape[i] > 0, MIS_ERR_INVALID_INPUT, "bad shape"); n *= (size_t)shape[i]; t->shape.push_back(shape[i]); }
    size_t esz = dtype == MIS_F32 ? 4 : 2;
    DevBuf<uint8_t> rawb;
    rawb.alloc(n * esz);
    t->buf.alloc(n);
    std::vector<uint8_t> permuted;
    if ((name == "model.encoder.conv1.weight" || name == "model.encoder.conv2.weight") && ndim == 3 && shape[1] == 3 && shape[2] != 3) {
        // MLX Conv1d layout [out, k, in] (mlx-whisper checkpoints; the reference transposes HF's [out, in, k] INTO this, :354-358):
        // the engine keeps HF's order, so bring it back
        std::vector<uint8_t> host(n * esz);
        HIP_CHECK(hipMemcpy(host.data(), data, n * esz, hipMemcpyDefault));
        permuted.resize(n * esz);
        const int64_t O = shape[0], K = shape[1], I = shape[2];
        for (int64_t o = 0; o < O; ++o) for (int64_t k = 0; k < K; ++k) for (int64_t i = 0; i < I; ++i)
            memcpy(&permuted[((o * I + i) * K + k) * esz], &host[((o * K + k) * I + i) * esz], esz);
        data = permuted.data();
        t->shape = {O, I, K};
    }
    HIP_CHECK(hipMemcpyAsync(rawb.p, data, n * esz, hipMemcpyDefault, c->stream));
    launch_convert_to_bf16(rawb.p, dtype, t->buf.p, n, c->stream);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->raw[name] = std::move(t);
    MIS_API_END
}

// conv weight [out][in][3] (HF / torch) -> [out][ldk] with column k*in + c (MLX [out, k, in] flattened), zero padded
__global__ void k_conv_w_reorder(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int out_c, int in_c, int ldk) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)out_c * ldk) return;
    int col = (int)(i % ldk), o = (int)(i / ldk);
    bf16_t v = 0;
    if (col < 3 * in_c) { int k = col / in_c, ci = col - k * in_c; v = src[((size_t)o * in_c + ci) * 3 + k]; }
    dst[i] = v;
}

static WTensor* wneed(mis_whisper* c, const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = c->raw.find(name);
    MIS_REQUIRE(it != c->raw.end(), MIS_ERR_NOT_INITIALIZED, "Whisper weight missing: %s", name.c_str());
    MIS_REQUIRE(it->second->shape == std::vector<int64_t>(shape), MIS_ERR_INVALID_INPUT, "Whisper weight %s has the wrong shape", name.c_str());
    return it->second.get();
}

extern "C" mis_status mis_whisper_finalize(mis_whisper* c) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && !c->finalized, MIS_ERR_INVALID_INPUT, "bad handle");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int64_t d = c->d, fe = c->cfg.encoder_ffn_dim, fd = c->cfg.decoder_ffn_dim, V = c->V, nm = c->nmel;
    const int Le = c->cfg.encoder_layers, Ld = c->cfg.decoder_layers;
    // arena size
    size_t total = (size_t)d * c->K1 + d + (size_t)d * 3 * d + d + (size_t)1500 * d + 2 * d;
    total += (size_t)Le * ((size_t)3 * d * d + 3 * d + (size_t)d * d + d + (size_t)fe * d + fe + (size_t)d * fe + d + 4 * d);
    total += (size_t)V * d + (size_t)c->Vpad * d + (size_t)c->cfg.max_target_positions * d + 2 * d;
    total += (size_t)Ld * ((size_t)3 * d * d + 3 * d + (size_t)d * d + d + (size_t)d * d + d + (size_t)2 * d * d + 2 * d +
                           (size_t)d * d + d + (size_t)fd * d + fd + (size_t)d * fd + d + 6 * d);
    total += 64 * (size_t)(Le * 12 + Ld * 20 + 16);
    c->arena.alloc(total);
    HIP_CHECK(hipMemsetAsync(c->arena.p, 0, total * 2, s));
    size_t off = 0;
    auto take = [&](size_t n) { bf16_t* p = c->arena.p + off; off += round_up(n, 64); MIS_REQUIRE(off <= total, MIS_ERR_GENERATION_FAILED, "arena overflow"); return p; };
    auto copy = [&](bf16_t* dst, WTensor* t) { HIP_CHECK(hipMemcpyAsync(dst, t->buf.p, t->buf.n * 2, hipMemcpyDeviceToDevice, s)); };
    auto vec = [&](const std::string& name, int64_t n) { bf16_t* p = take(n); copy(p, wneed(c, name, {n})); return p; };
    auto pack = [&](const std::string& name, int64_t N, int64_t K, bf16_t* dst, int NT_total_offset_tiles) {
        launch_pack_weight(wneed(c, name, {N, K})->buf.p, dst, (int)N, (int)K, (int)(N / 16), 1, NT_total_offset_tiles, s);
    };
    const std::string E = "model.encoder", Dd = "model.decoder";
    // ---- encoder
    c->conv1w = take((size_t)d * c->K1);
    hipLaunchKernelGGL(k_conv_w_reorder, dim3((unsigned)((d * c->K1 + 255) / 256)), dim3(256), 0, s,
                       wneed(c, E + ".conv1.weight", {d, nm, 3})->buf.p, c->conv1w, (int)d, (int)nm, c->K1);
    c->conv1b = vec(E + ".conv1.bias", d);
    c->conv2w = take((size_t)d * 3 * d);
    hipLaunchKernelGGL(k_conv_w_reorder, dim3((unsigned)((d * 3 * d + 255) / 256)), dim3(256), 0, s,
                       wneed(c, E + ".conv2.weight", {d, d, 3})->buf.p, c->conv2w, (int)d, (int)d, (int)(3 * d));
    c->conv2b = vec(E + ".conv2.bias", d);
    c->enc_pos = take((size_t)1500 * d);
    if (!c->raw.count(E + ".embed_positions.weight")) {
        // mlx-whisper checkpoints omit the fixed sinusoid (WhisperModel.swift:376-392 synthesises it): [sin | cos] halves
        const int64_t half = d / 2;
        const double inc = log(10000.0) / (double)std::max<int64_t>(half - 1, 1);
        std::vector<bf16_t> pe((size_t)1500 * d);
        for (int64_t pos = 0; pos < 1500; ++pos)
            for (int64_t i = 0; i < half; ++i) {
                const double st = (double)pos * exp(-inc * (double)i);
                pe[pos * d + i] = f32_to_bf16((float)sin(st));
                pe[pos * d + half + i] = f32_to_bf16((float)cos(st));
            }
        HIP_CHECK(hipMemcpyAsync(c->enc_pos, pe.data(), pe.size() * 2, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
    } else
        copy(c->enc_pos, wneed(c, E + ".embed_positions.weight", {1500, d}));
    c->enc.resize(Le);
    for (int li = 0; li < Le; ++li) {
        std::string q = E + ".layers." + std::to_string(li);
        EncLayer& L = c->enc[li];
        L.wqkv = take((size_t)3 * d * d);
        copy(L.wqkv, wneed(c, q + ".self_attn.q_proj.weight", {d, d}));
        copy(L.wqkv + (size_t)d * d, wneed(c, q + ".self_attn.k_proj.weight", {d, d}));
        copy(L.wqkv + (size_t)2 * d * d, wneed(c, q + ".self_attn.v_proj.weight", {d, d}));
        L.bqkv = take(3 * d);                                                 // k_proj has no bias (WhisperLayers.swift:29)
        copy(L.bqkv, wneed(c, q + ".self_attn.q_proj.bias", {d}));
        copy(L.bqkv + 2 * d, wneed(c, q + ".self_attn.v_proj.bias", {d}));
        L.wo = take((size_t)d * d); copy(L.wo, wneed(c, q + ".self_attn.out_proj.weight", {d, d}));
        L.bo = vec(q + ".self_attn.out_proj.bias", d);
        L.fc1 = take((size_t)fe * d); copy(L.fc1, wneed(c, q + ".fc1.weight", {fe, d}));
        L.b1 = vec(q + ".fc1.bias", fe);
        L.fc2 = take((size_t)d * fe); copy(L.fc2, wneed(c, q + ".fc2.weight", {d, fe}));
        L.b2 = vec(q + ".fc2.bias", d);
        L.ln1w = vec(q + ".self_attn_layer_norm.weight", d); L.ln1b = vec(q + ".self_attn_layer_norm.bias", d);
        L.ln2w = vec(q + ".final_layer_norm.weight", d); L.ln2b = vec(q + ".final_layer_norm.bias", d);
    }
    c->enc_lnw = vec(E + ".layer_norm.weight", d); c->enc_lnb = vec(E + ".layer_norm.bias", d);
    // ---- decoder
    c->emb = take((size_t)V * d);
    copy(c->emb, wneed(c, Dd + ".embed_tokens.weight", {V, d}));
    c->emb_packed = take((size_t)c->Vpad * d);
    launch_pack_weight(c->emb, c->emb_packed, (int)V, (int)d, c->Vpad / 16, 1, 0, s);        // projectToVocab: tied (:325-327)
    c->dec_pos = take((size_t)c->cfg.max_target_positions * d);
    copy(c->dec_pos, wneed(c, Dd + ".embed_positions.weight", {(int64_t)c->cfg.max_target_positions, d}));
    c->dec.resize(Ld);
    for (int li = 0; li < Ld; ++li) {
        std::string q = Dd + ".layers." + std::to_string(li);
        DecLayer& L = c->dec[li];
        L.sqkv = take((size_t)3 * d * d);
        pack(q + ".self_attn.q_proj.weight", d, d, L.sqkv, 0);
        pack(q + ".self_attn.k_proj.weight", d, d, L.sqkv, (int)(d / 16));
        pack(q + ".self_attn.v_proj.weight", d, d, L.sqkv, (int)(2 * d / 16));
        L.sbqkv = take(3 * d);
        copy(L.sbqkv, wneed(c, q + ".self_attn.q_proj.bias", {d}));
        copy(L.sbqkv + 2 * d, wneed(c, q + ".self_attn.v_proj.bias", {d}));
        L.so = take((size_t)d * d); pack(q + ".self_attn.out_proj.weight", d, d, L.so, 0);
        L.sbo = vec(q + ".self_attn.out_proj.bias", d);
        L.cq = take((size_t)d * d); pack(q + ".encoder_attn.q_proj.weight", d, d, L.cq, 0);
        L.cbq = vec(q + ".encoder_attn.q_proj.bias", d);
        L.ckv = take((size_t)2 * d * d);
        copy(L.ckv, wneed(c, q + ".encoder_attn.k_proj.weight", {d, d}));
        copy(L.ckv + (size_t)d * d, wneed(c, q + ".encoder_attn.v_proj.weight", {d, d}));
        L.cbkv = take(2 * d);
        copy(L.cbkv + d, wneed(c, q + ".encoder_attn.v_proj.bias", {d}));
        L.co = take((size_t)d * d); pack(q + ".encoder_attn.out_proj.weight", d, d, L.co, 0);
        L.cbo = vec(q + ".encoder_attn.out_proj.bias", d);
        L.fc1 = take((size_t)fd * d); pack(q + ".fc1.weight", fd, d, L.fc1, 0);
        L.b1 = vec(q + ".fc1.bias", fd);
        L.fc2 = take((size_t)d * fd); pack(q + ".fc2.weight", d, fd, L.fc2, 0);
        L.b2 = vec(q + ".fc2.bias", d);
        L.ln1w = vec(q + ".self_attn_layer_norm.weight", d); L.ln1b = vec(q + ".self_attn_layer_norm.bias", d);
        L.ln2w = vec(q + ".encoder_attn_layer_norm.weight", d); L.ln2b = vec(q + ".encoder_attn_layer_norm.bias", d);
        L.ln3w = vec(q + ".final_layer_norm.weight", d); L.ln3b = vec(q + ".final_layer_norm.bias", d);
    }
    c->dec_lnw = vec(Dd + ".layer_norm.weight", d); c->dec_lnb = vec(Dd + ".layer_norm.bias", d);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(s));
    c->raw.clear();
    c->finalized = true;
    MIS_API_END
}

// mis-synth-v1 weights, same key order / amplitudes as oracle/whisper.py make_synthetic_weights
extern "C" mis_status mis_whisper_init_synthetic(mis_whisper* c, uint64_t seed) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && !c->finalized, MIS_ERR_INVALID_INPUT, "bad handle");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    uint64_t key = seed * 100000ull;
    const int64_t d = c->d, nm = c->nmel;
    auto put = [&](const std::string& name, std::vector<int64_t> shape, double amp, int plus_one) {
        auto t = std::make_unique<WTensor>();
        size_t n = 1;
        for (auto v : shape) n *= (size_t)v;
        t->shape = shape;
        t->buf.alloc(n);
        launch_synth_fill_bf16(t->buf.p, n, ++key, (float)amp, plus_one, s);
        c->raw[name] = std::move(t);
    };
    auto lin = [&](const std::string& p, int64_t out_f, int64_t in_f, bool bias, double gain) {
        put(p + ".weight", {out_f, in_f}, gain * sqrt(3.0 / (double)in_f), 0);
        if (bias) put(p + ".bias", {out_f}, 0.05, 0);
    };
    auto lnp = [&](const std::string& p) { put(p + ".weight", {d}, 0.1, 2); put(p + ".bias", {d}, 0.05, 0); };
    auto attn = [&](const std::string& p) {
        lin(p + ".q_proj", d, d, true, 1.0); lin(p + ".k_proj", d, d, false, 1.0); lin(p + ".v_proj", d, d, true, 1.0);
        lin(p + ".out_proj", d, d, true, 0.5);
    };
    const std::string E = "model.encoder", Dd = "model.decoder";
    put(E + ".conv1.weight", {d, nm, 3}, sqrt(3.0 / (3.0 * nm)), 0); put(E + ".conv1.bias", {d}, 0.05, 0);
    put(E + ".conv2.weight", {d, d, 3}, sqrt(3.0 / (3.0 * d)), 0); put(E + ".conv2.bias", {d}, 0.05, 0);
    put(E + ".embed_positions.weight", {1500, d}, 0.3, 0);
    for (int li = 0; li < c->cfg.encoder_layers; ++li) {
        std::string q = E + ".layers." + std::to_string(li);
        attn(q + ".self_attn"); lnp(q + ".self_attn_layer_norm");
        lin(q + ".fc1", c->cfg.encoder_ffn_dim, d, true, 1.0); lin(q + ".fc2", d, c->cfg.encoder_ffn_dim, true, 0.5);
        lnp(q + ".final_layer_norm");
    }
    lnp(E + ".layer_norm");
    put(Dd + ".embed_tokens.weight", {(int64_t)c->V, d}, 0.5, 0);
    put(Dd + ".embed_positions.weight", {(int64_t)c->cfg.max_target_positions, d}, 0.3, 0);
    for (int li = 0; li < c->cfg.decoder_layers; ++li) {
        std::string q = Dd + ".layers." + std::to_string(li);
        attn(q + ".self_attn"); lnp(q + ".self_attn_layer_norm");
        attn(q + ".encoder_attn"); lnp(q + ".encoder_attn_layer_norm");
        lin(q + ".fc1", c->cfg.decoder_ffn_dim, d, true, 1.0); lin(q + ".fc2", d, c->cfg.decoder_ffn_dim, true, 0.5);
        lnp(q + ".final_layer_norm");
    }
    lnp(Dd + ".layer_norm");
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(s));
    MIS_API_END
}

// ---------------------------------------------------------------------------- encoder
static int split_for(int items, int KT, int s_max = 16) { return gemm_choose_split(items, KT, 4, s_max); }   // R = 2, KSB = 4 launches below

static void whisper_alloc_state(mis_whisper* c, int batch) {
    const int d = c->d, Ld = c->cfg.decoder_layers;
    int Mpad = (int)round_up(batch, 16);
    int Smax = (int)round_up(c->cfg.max_target_positions, 64);
    c->batch = batch; c->Mpad = Mpad; c->Smax = Smax;
    c->S_qkv = split_for(3 * d / 16 / 2, d / 32, 8);      // attention prologue: <= 8 slabs
    c->S_o = split_for(d / 16 / 2, d / 32);
    c->S_cq = c->S_o;
    c->S_fc2 = split_for(d / 16 / 2, c->cfg.decoder_ffn_dim / 32);
    size_t ck = (size_t)Ld * batch * c->Hd * c->Spad * c->D;
    c->cross_k.alloc(ck); c->cross_v.alloc(ck);
    size_t sk = (size_t)Ld * batch * c->Hd * Smax * c->D;
    c->self_k.alloc(sk); c->self_v.alloc(sk);
    c->enc_out.alloc((size_t)batch * 1500 * d);
    c->ids.alloc(Mpad); c->pos_cur.alloc(Mpad); c->pos_next.alloc(Mpad); c->active.alloc(Mpad); c->n_gen.alloc(Mpad);
    c->next_ids.alloc(Mpad); c->done_count.alloc(1);
    c->h.alloc((size_t)Mpad * d); c->h2.alloc((size_t)Mpad * d); c->x.alloc((size_t)Mpad * std::max(d, c->cfg.decoder_ffn_dim));
    c->attn_out.alloc((size_t)Mpad * d); c->act.alloc((size_t)Mpad * c->cfg.decoder_ffn_dim);
    c->logits.alloc((size_t)Mpad * c->Vpad); c->e_buf.alloc((size_t)Mpad * c->Vpad);
    c->qkv_part.alloc((size_t)c->S_qkv * Mpad * 3 * d);
    c->part.alloc((size_t)std::max(std::max(c->S_o, c->S_cq), c->S_fc2) * Mpad * d);
    c->scratch.alloc(batch);
    sampler_scratch_init(c->scratch.p, batch, c->stream);
}

static void whisper_decoder_reset(mis_whisper* c) {
    hipStream_t s = c->stream;
    HIP_CHECK(hipMemsetAsync(c->self_k.p, 0, c->self_k.bytes(), s));
    HIP_CHECK(hipMemsetAsync(c->self_v.p, 0, c->self_v.bytes(), s));
    c->ids.zero(s); c->pos_cur.zero(s); c->pos_next.zero(s); c->active.zero(s); c->n_gen.zero(s); c->next_ids.zero(s);
    c->done_count.zero(s);
    c->h.zero(s); c->h2.zero(s); c->x.zero(s); c->attn_out.zero(s); c->act.zero(s); c->logits.zero(s);
    HIP_CHECK(hipStreamSynchronize(s));
}

// features_dev f32 [B][3000][nmel] -> enc_out bf16 [B*1500][d]; cross K/V of every decoder layer
static void whisper_encode_device(mis_whisper* c, const float* features_dev, int batch) {
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "model not finalized");
    MIS_REQUIRE(batch >= 1 && batch <= 64, MIS_ERR_INVALID_INPUT, "batch per GPU must be 1..64");
    hipStream_t s = c->stream;
    const int d = c->d, fe = c->cfg.encoder_ffn_dim, H = c->He, D = c->D;
    if (batch != c->batch) whisper_alloc_state(c, batch);
    HIP_CHECK(hipMemsetAsync(c->cross_k.p, 0, c->cross_k.bytes(), s));
    HIP_CHECK(hipMemsetAsync(c->cross_v.p, 0, c->cross_v.bytes(), s));
    // the encoder runs in sub-batches of <= 8 utterances to bound the activation buffers
    const int SB = std::min(batch, 8);
    const size_t M1 = (size_t)SB * 3000, M = (size_t)SB * 1500;
    DevBuf<bf16_t> col1, h1, col2, h, x, qkv, att, ff, kc, vc, ckv;
    col1.alloc(M1 * c->K1); h1.alloc(M1 * d); col2.alloc(M * 3 * d); h.alloc(M * d); x.alloc(M * d); qkv.alloc(M * 3 * d);
    att.alloc(M * d); ff.alloc(M * fe); ckv.alloc(M * 2 * d);
    size_t kvn = (size_t)SB * H * c->Spad * D;
    kc.alloc(kvn); vc.alloc(kvn);
    HIP_CHECK(hipMemsetAsync(col1.p, 0, col1.bytes(), s));
    HIP_CHECK(hipMemsetAsync(kc.p, 0, kvn * 2, s));
    HIP_CHECK(hipMemsetAsync(vc.p, 0, kvn * 2, s));
    for (int b0 = 0; b0 < batch; b0 += SB) {
        const int nb = std::min(SB, batch - b0);
        const int m1 = nb * 3000, m = nb * 1500;
        BigGemmParams g{};
        // gelu(conv1), gelu(conv2) + positions     (WhisperLayers.swift:147-151)
        {   // conv1 patches with row stride K1 (zero padded columns)
            DevBuf<bf16_t> tight;
            tight.alloc((size_t)m1 * 3 * c->nmel);
            launch_im2col3_f32(features_dev + (size_t)b0 * 3000 * c->nmel, tight.p, nb, 3000, c->nmel, 3000, 1, s);
            HIP_CHECK(hipMemcpy2DAsync(col1.p, (size_t)c->K1 * 2, tight.p, (size_t)3 * c->nmel * 2, (size_t)3 * c->nmel * 2, m1,
                                       hipMemcpyDeviceToDevice, s));
            HIP_CHECK(hipStreamSynchronize(s));
        }
        g = BigGemmParams{col1.p, c->conv1w, c->conv1b, nullptr, h1.p, m1, d, c->K1, c->K1, 0};
        launch_gemm_big(BG_GELU, g, s);
        launch_im2col3_bf16(h1.p, col2.p, nb, 3000, d, 1500, 2, s);
        g = BigGemmParams{col2.p, c->conv2w, c->conv2b, c->enc_pos, h.p, m, d, 3 * d, 3 * d, 1500};
        launch_gemm_big(BG_GELU_POS, g, s);
        for (size_t li = 0; li < c->enc.size(); ++li) {
            const EncLayer& L = c->enc[li];
            launch_layernorm(h.p, x.p, L.ln1w, L.ln1b, m, d, LN_EPS, s);
            g = BigGemmParams{x.p, L.wqkv, L.bqkv, nullptr, qkv.p, m, 3 * d, d, d, 0};
            launch_gemm_big(BG_NONE, g, s);
            launch_scatter_kv(qkv.p, 3 * d, d, 2 * d, kc.p, vc.p, nb, 1500, H, D, c->Spad, s);
            launch_attn_prefill(qkv.p, 3 * d, kc.p, vc.p, att.p, d, nb, 1500, H, D, c->Spad, s);
            g = BigGemmParams{att.p, L.wo, L.bo, h.p, h.p, m, d, d, d, 0};
            launch_gemm_big(BG_RESID, g, s);                                   // h = h + out_proj(attn)
            launch_layernorm(h.p, x.p, L.ln2w, L.ln2b, m, d, LN_EPS, s);
            g = BigGemmParams{x.p, L.fc1, L.b1, nullptr, ff.p, m, fe, d, d, 0};
            launch_gemm_big(BG_GELU, g, s);
            g = BigGemmParams{ff.p, L.fc2, L.b2, h.p, h.p, m, d, fe, fe, 0};
            launch_gemm_big(BG_RESID, g, s);
        }
        bf16_t* eo = c->enc_out.p + (size_t)b0 * 1500 * d;
        launch_layernorm(h.p, eo, c->enc_lnw, c->enc_lnb, m, d, LN_EPS, s);
        // cross-attention K/V of every decoder layer, computed once per utterance (WhisperLayers.swift:216-243)
        for (size_t li = 0; li < c->dec.size(); ++li) {
            const DecLayer& L = c->dec[li];
            g = BigGemmParams{eo, L.ckv, L.cbkv, nullptr, ckv.p, m, 2 * d, d, d, 0};
            launch_gemm_big(BG_NONE, g, s);
            size_t lstride = (size_t)c->batch * c->Hd * c->Spad * D;
            size_t boff = (size_t)b0 * c->Hd * c->Spad * D;
            launch_scatter_kv(ckv.p, 2 * d, 0, d, c->cross_k.p + li * lstride + boff, c->cross_v.p + li * lstride + boff, nb, 1500,
                              c->Hd, D, c->Spad, s);
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(s));
    }
    whisper_decoder_reset(c);
}

__global__ void k_bf16_to_f32_flat(const bf16_t* __restrict__ src, float* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = bf16_to_f32(src[i]);
}

extern "C" mis_status mis_whisper_encode(mis_whisper* c, const float* features, int batch, float* enc_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && features, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(batch >= 1 && batch <= 64, MIS_ERR_INVALID_INPUT, "batch per GPU must be 1..64");
    HIP_CHECK(hipSetDevice(c->device));
    DevBuf<float> f;
    size_t n = (size_t)batch * 3000 * c->nmel;
    f.alloc(n);
    HIP_CHECK(hipMemcpy(f.p, features, n * 4, hipMemcpyDefault));
    whisper_encode_device(c, f.p, batch);
    if (enc_out) {
        size_t ne = (size_t)batch * 1500 * c->d;
        DevBuf<float> o;
        o.alloc(ne);
        hipLaunchKernelGGL(k_bf16_to_f32_flat, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, c->stream, c->enc_out.p, o.p, ne);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        HIP_CHECK(hipMemcpy(enc_out, o.p, ne * 4, hipMemcpyDefault));
    }
    MIS_API_END
}

// ---------------------------------------------------------------------------- decoder step
// Up to 16 rows (the BASELINE share: 8 windows per GPU) two of the three residual + LayerNorm glue launches of a layer - and one GEMM launch -
// run inside the prologue of their consumer: 8 launches per layer instead of 11 (transcribe 228 -> 214 ms per 8 x 30 s, profiles/r06/c4, c5).
//   bit 4 (16): h += self-attention output, LayerNorm 2 AND the cross-attention's query projection inside the cross-attention kernel
//               (k_attn_decode<64, 2, true, QP>, lm_kernels.hip: a block is one (row, head) and rebuilds only its row);
//   bit 2 (4):  h += cross-attention output and LayerNorm 3 inside fc1's prologue (k_gemm_skinny_norm: every block rebuilds all rows);
//   bit 3 (8):  fail where a requested fold does not apply (tests).  MIS_WHISPER_FOLD=0 keeps the separate launches.
// The residual stream alternates between two buffers: every block of a consumer still reads the old one while one of them writes the new.
// Measured and not kept (profiles/r06/c3, c5): the same GEMM-side fold for q|k|v and the cross query (every one of 120-480 blocks re-reads all
// rows' slabs: neutral to +1 %), LayerNorm 1 + q|k|v inside the SELF-attention kernel (480 KB of weights per block in three trips: the
// kernel 5.0 -> 15.6 us against 9.6 saved; the variant is on file as c5_self_attention_qkv_fold_measured_variant.patch), and the ONE-SLAB
// arrangement (profiles/r06/c9: every producer GEMM as eight-wave items over the whole K range writing one float32 slab, so that LayerNorm 1
// could move into q|k|v's prologue as well - 7 launches per layer, parity green, transcribe 214 -> 224 ms: fc2 at 80 blocks 5.2 -> 8.5 us).
#define WHISPER_FOLD_DEFAULT 20
static void enqueue_decoder_step(mis_whisper* c) {
    hipStream_t s = c->stream;
    const int d = c->d, fd = c->cfg.decoder_ffn_dim, Mpad = c->Mpad, H = c->Hd, D = c->D;
    const DecLayer& L0 = c->dec[0];
    const char* fe = getenv("MIS_WHISPER_FOLD");
    const int want = fe ? atoi(fe) : WHISPER_FOLD_DEFAULT;
    AttnParams probe{};
    probe.cross = 1; probe.cross_len = 1500; probe.D = D; probe.H = H; probe.Hkv = H; probe.qp_KT = d / 32; probe.qp_S = c->S_o;
    const bool f_cqa = (want & 16) && Mpad == 16 && attn_qp_ok(probe);
    const bool f_fc1 = (want & 4) && gemm_skinny_norm_ok(EPI_GELU_PACKED, 2, d / 32, 1, c->S_o, Mpad, c->batch);
    MIS_REQUIRE(!(want & 8) || (f_cqa == !!(want & 16) && f_fc1 == !!(want & 4)), MIS_ERR_GENERATION_FAILED,
                "MIS_WHISPER_FOLD: the folded decoder step does not apply to this shape");
    bf16_t* hcur = c->h.p;
    bf16_t* hoth = c->h2.p;
    launch_whisper_embed_ln(c->emb, c->dec_pos, c->ids.p, c->active.p, c->pos_cur.p, c->pos_next.p, L0.ln1w, L0.ln1b, hcur,
                            c->x.p, d, c->V, c->cfg.max_target_positions, c->batch, Mpad, s);
    for (size_t li = 0; li < c->dec.size(); ++li) {
        const DecLayer& L = c->dec[li];
        // self attention (WhisperLayers.swift:202-214)
        launch_gemm_skinny(EPI_PARTIAL, 2, 4, L.sqkv, c->x.p, c->qkv_part.p, 3 * d / 16, d / 32, c->S_qkv, 3 * d, Mpad, s, L.sbqkv);
        AttnParams ap{};
        ap.qkv_part = c->qkv_part.p; ap.S = c->S_qkv; ap.Mpad = Mpad; ap.Nqkv = 3 * d;
        size_t ls = (size_t)c->batch * H * c->Smax * D;
        ap.kcache = c->self_k.p + li * ls; ap.vtcache = c->self_v.p + li * ls;
        ap.pos = c->pos_cur.p; ap.active = c->active.p; ap.rope_cos = nullptr; ap.rope_sin = nullptr;
        ap.out = c->attn_out.p; ap.H = H; ap.Hkv = H; ap.D = D; ap.Smax = c->Smax; ap.scale = 1.0f / sqrtf((float)D);
        launch_attn_decode(ap, c->batch, s);
        launch_gemm_skinny(EPI_PARTIAL, 2, 4, L.so, c->attn_out.p, c->part.p, d / 16, d / 32, c->S_o, d, Mpad, s, L.sbo);
        // cross attention over the cached encoder K/V (:216-243)
        AttnParams cp{};
        cp.Mpad = Mpad;
        if (f_cqa) {                                       // the launch below does h += self-attention output, LayerNorm 2 and q = W_q x + b itself
            cp.qp_w = L.cq; cp.qp_bias = L.cbq; cp.qp_slabs = c->part.p; cp.qp_S = c->S_o; cp.qp_KT = d / 32; cp.qp_h_in = hcur; cp.qp_h_out = hoth;
            cp.qp_lnw = L.ln2w; cp.qp_lnb = L.ln2b; cp.qp_eps = LN_EPS;
            std::swap(hcur, hoth);
        } else {
            launch_reduce_residual_rmsnorm(c->part.p, c->S_o, Mpad, d, hcur, L.ln2w, c->x.p, LN_EPS, s, L.ln2b);
            launch_gemm_skinny(EPI_PARTIAL, 2, 4, L.cq, c->x.p, c->qkv_part.p, d / 16, d / 32, c->S_cq, d, Mpad, s, L.cbq);
            cp.qkv_part = c->qkv_part.p; cp.S = c->S_cq; cp.Nqkv = d;
        }
        size_t cs = (size_t)c->batch * H * c->Spad * D;
        cp.kcache = c->cross_k.p + li * cs; cp.vtcache = c->cross_v.p + li * cs;
        cp.pos = c->pos_cur.p; cp.active = c->active.p; cp.rope_cos = nullptr; cp.rope_sin = nullptr;
        cp.out = c->attn_out.p; cp.H = H; cp.Hkv = H; cp.D = D; cp.Smax = c->Spad; cp.scale = 1.0f / sqrtf((float)D);
        cp.cross = 1; cp.cross_len = 1500;
        launch_attn_decode(cp, c->batch, s);
        launch_gemm_skinny(EPI_PARTIAL, 2, 4, L.co, c->attn_out.p, c->part.p, d / 16, d / 32, c->S_o, d, Mpad, s, L.cbo);
        // MLP (:245-249)
        if (f_fc1) {
            launch_gemm_skinny_norm(EPI_GELU_PACKED, 2, L.fc1, c->part.p, c->S_o, hcur, hoth, L.ln3w, L.ln3b, LN_EPS, c->batch, c->act.p, fd / 16, d / 32, 1,
                                    fd, Mpad, s, L.b1);
            std::swap(hcur, hoth);
        } else {
            launch_reduce_residual_rmsnorm(c->part.p, c->S_o, Mpad, d, hcur, L.ln3w, c->x.p, LN_EPS, s, L.ln3b);
            launch_gemm_skinny(EPI_GELU_PACKED, 2, 4, L.fc1, c->x.p, c->act.p, fd / 16, d / 32, 1, fd, Mpad, s, L.b1);
        }
        launch_gemm_skinny(EPI_PARTIAL, 2, 4, L.fc2, c->act.p, c->part.p, d / 16, fd / 32, c->S_fc2, d, Mpad, s, L.b2);
        const bf16_t* nw = (li + 1 < c->dec.size()) ? c->dec[li + 1].ln1w : c->dec_lnw;
        const bf16_t* nb = (li + 1 < c->dec.size()) ? c->dec[li + 1].ln1b : c->dec_lnb;
        launch_reduce_residual_rmsnorm(c->part.p, c->S_fc2, Mpad, d, hcur, nw, c->x.p, LN_EPS, s, nb);
    }
}
static void enqueue_vocab(mis_whisper* c) {
    launch_gemm_skinny(EPI_BF16, 2, 1, c->emb_packed, c->x.p, c->logits.p, c->Vpad / 16, c->d / 32, 1, c->Vpad, c->Mpad, c->stream);
}

__global__ void k_bf16_rows_to_f32_w(const bf16_t* __restrict__ src, int src_stride, float* __restrict__ dst, int cols, int rows) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * cols) return;
    int r = (int)(i / cols), cidx = (int)(i - (size_t)r * cols);
    dst[i] = bf16_to_f32(src[(size_t)r * src_stride + cidx]);
}

extern "C" mis_status mis_whisper_decoder_reset(mis_whisper* c) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && c->batch > 0, MIS_ERR_NOT_INITIALIZED, "encode first");
    HIP_CHECK(hipSetDevice(c->device));
    whisper_decoder_reset(c);
    MIS_API_END
}

extern "C" mis_status mis_whisper_decoder_forward(mis_whisper* c, const int32_t* tokens, const uint8_t* active, float* logits_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && tokens, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(c->batch > 0, MIS_ERR_NOT_INITIALIZED, "encode first");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    std::vector<uint8_t> act(c->batch, 1);
    if (active) HIP_CHECK(hipMemcpy(act.data(), active, c->batch, hipMemcpyDefault));
    std::vector<int32_t> pn(c->batch);
    HIP_CHECK(hipMemcpy(pn.data(), c->pos_next.p, c->batch * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < c->batch; ++b)
        MIS_REQUIRE(!act[b] || pn[b] < c->cfg.max_target_positions, MIS_ERR_INVALID_INPUT, "row %d exceeds max_target_positions", b);
    HIP_CHECK(hipMemcpyAsync(c->ids.p, tokens, c->batch * 4, hipMemcpyDefault, s));
    HIP_CHECK(hipMemcpyAsync(c->active.p, act.data(), c->batch, hipMemcpyHostToDevice, s));
    enqueue_decoder_step(c);
    if (logits_out) {
        enqueue_vocab(c);
        size_t n = (size_t)c->batch * c->V;
        c->logits_f32.alloc(n);
        hipLaunchKernelGGL(k_bf16_rows_to_f32_w, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->logits.p, c->Vpad,
                           c->logits_f32.p, c->V, c->batch);
        HIP_CHECK(hipMemcpyAsync(logits_out, c->logits_f32.p, n * 4, hipMemcpyDefault, s));
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(s));
    MIS_API_END
}

// transcribeChunk for a batch of <= 30 s windows (WhisperModel.swift:186-282): mel -> encoder -> prompt prefill ->
// greedy / temperature loop with the suppress masks, until EOT or max_tokens.  Token ids only; text stays host side.
static void whisper_generate_impl(mis_whisper* c, const float* pcm, const int64_t* lens, int batch, int64_t stride,
                                  const int32_t* prompt_ids, int n_prompt, const mis_stt_params* sp,
                                  int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens, mis_event_cb on_event, void* user,
                                  const volatile int* cancel_flag) {
    MIS_REQUIRE(c && prompt_ids && sp, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "model not finalized");
    MIS_REQUIRE(batch >= 1 && batch <= 64 && n_prompt >= 1 && stride >= 0, MIS_ERR_INVALID_INPUT, "bad sizes");
    MIS_REQUIRE(stride == 0 || pcm, MIS_ERR_INVALID_INPUT, "null audio");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // ---- features (WhisperAudio.encoderFeatures) on the device, then the encoder
    const int64_t W = 480000;
    std::vector<int64_t> hl(batch, stride);
    if (lens) HIP_CHECK(hipMemcpy(hl.data(), lens, batch * sizeof(int64_t), hipMemcpyDefault));
    DevBuf<float> padded, feats;
    padded.alloc((size_t)batch * W);
    feats.alloc((size_t)batch * 3000 * c->nmel);
    HIP_CHECK(hipMemset(padded.p, 0, (size_t)batch * W * 4));
    for (int b = 0; b < batch; ++b) {
        int64_t n = std::min<int64_t>(std::min<int64_t>(hl[b], stride), W);
        if (n > 0) HIP_CHECK(hipMemcpy(padded.p + (size_t)b * W, pcm + (size_t)b * stride, (size_t)n * 4, hipMemcpyDefault));
    }
    whisper_features_device(c->device, padded.p, batch, c->nmel, feats.p, s);
    whisper_encode_device(c, feats.p, batch);
    // ---- prompt prefill, one token per step (same prompt for every row)
    std::vector<int32_t> prompt(n_prompt);
    HIP_CHECK(hipMemcpy(prompt.data(), prompt_ids, n_prompt * 4, hipMemcpyDefault));
    // maxTokens = max(1, min(maxTokens, maxTargetPositions - prompt - 1))   (:222-226)
    int max_tokens = std::max(1, std::min(sp->max_tokens > 0 ? sp->max_tokens : c->cfg.max_target_positions,
                                          c->cfg.max_target_positions - n_prompt - 1));
    std::vector<uint8_t> ones(batch, 1);
    for (int j = 0; j < n_prompt; ++j) MIS_REQUIRE(prompt[j] >= 0 && prompt[j] < c->V, MIS_ERR_INVALID_INPUT, "prompt token outside the vocabulary");
    std::vector<int32_t> emitted(batch, 0);              // ids already announced to the callback (kept across a second attempt)
    // One attempt of prompt prefill + decode loop on the encoder output already in place.  Returns false when a row barrier of the
    // one-launch sampler timed out (its blocks were not co-resident: another stream holds compute units) - seen at the poll, before any
    // id of that poll interval is announced; the caller resets the decoder state and runs the attempt again on the multi-launch
    // sampler (deterministic: same ids, the callback continues behind `emitted`).
    auto attempt = [&](bool multi_launch_only) -> bool {
        for (int j = 0; j < n_prompt; ++j) {
            std::vector<int32_t> row(batch, prompt[j]);
            HIP_CHECK(hipMemcpyAsync(c->ids.p, row.data(), batch * 4, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(c->active.p, ones.data(), batch, hipMemcpyHostToDevice, s));
            enqueue_decoder_step(c);
            HIP_CHECK(hipStreamSynchronize(s));
        }
        // ---- decode loop
        c->tokens_out.alloc((size_t)batch * max_tokens);
        c->tokens_out.zero(s);
        c->sup.alloc(std::max(sp->n_suppress, 1)); c->bsup.alloc(std::max(sp->n_begin_suppress, 1));
        if (sp->n_suppress > 0) HIP_CHECK(hipMemcpyAsync(c->sup.p, sp->suppress, sp->n_suppress * 4, hipMemcpyDefault, s));
        if (sp->n_begin_suppress > 0) HIP_CHECK(hipMemcpyAsync(c->bsup.p, sp->begin_suppress, sp->n_begin_suppress * 4, hipMemcpyDefault, s));
        SamplerParams q{};
        q.scratch = c->scratch.p;
        sampler_plan(c->V, &q.n_chunks, &q.chunk_w);
        q.logits = c->logits.p; q.e_buf = c->e_buf.p; q.Vpad = c->Vpad; q.vocab = c->V;
        q.active_in = c->active.p; q.n_gen = c->n_gen.p; q.tokens_out = c->tokens_out.p; q.tokens_stride = max_tokens;
        q.next_ids = c->ids.p; q.active = c->active.p; q.done_count = c->done_count.p;
        q.temperature = sp->temperature > 0 ? sp->temperature : 0.0f; q.top_p = 1.0f; q.penalty = 0.0f; q.seed = sp->seed; q.row_offset = sp->row_offset;
        q.lo = 0; q.hi = (sp->timestamp_begin > 0 && sp->timestamp_begin < c->V) ? sp->timestamp_begin : c->V;   // suppressFromIndex
        q.eos_id = sp->eot_id; q.max_tokens = max_tokens;
        sampler_resolve(q, multi_launch_only);               // (environment switches read once per attempt)
        PinnedBuf<int32_t> done_pin(1);
        PinnedBuf<unsigned> fail_pin(batch);
        bool sampler_failed = false;
        int32_t* done_host = done_pin.p;
        *done_host = 0;
        // one decode step = logits GEMM -> suppress masks -> argmax / sampler -> next token through the decoder: every argument
        // is a device pointer that stays put, so the step is captured once and replayed (≈430 kernel nodes per step)
        auto step_body = [&]() {
            enqueue_vocab(c);
            launch_whisper_suppress(c->logits.p, c->Vpad, c->V, c->sup.p, sp->n_suppress, c->bsup.p, sp->n_begin_suppress, c->n_gen.p,
                                    c->active.p, batch, s);
            launch_sampler(q, batch, s);
            enqueue_decoder_step(c);
    };
    hipGraphExec_t gexec = nullptr;
    const bool use_graph = getenv("MIS_NO_GRAPH") == nullptr;
    try {
        if (use_graph) {
            hipGraph_t g = nullptr;
            HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            try { step_body(); } catch (...) { hipGraph_t dead = nullptr; (void)hipStreamEndCapture(s, &dead); if (dead) (void)hipGraphDestroy(dead); throw; }
            HIP_CHECK(hipStreamEndCapture(s, &g));
            HIP_CHECK(hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0));
            HIP_CHECK(hipGraphDestroy(g));
        }
        // stream form: the ids sampled since the last poll are announced in step order per row (the reference decodes them to a
        // text delta per step, WhisperModel.swift:242-254; detokenisation stays with the host); EOT is never announced (:238)
        std::vector<int32_t> h_ng(batch), h_tok;
        const int poll = on_event ? 3 : 7;
        for (int step = 0; step < max_tokens; ++step) {
            if (use_graph) HIP_CHECK(hipGraphLaunch(gexec, s)); else step_body();
            if ((step & poll) == poll || step + 1 == max_tokens) {
                HIP_CHECK(hipMemcpyAsync(done_host, c->done_count.p, 4, hipMemcpyDeviceToHost, s));
                sampler_fail_flags_async(c->scratch.p, batch, fail_pin.p, s);
                if (on_event) {
                    h_tok.resize((size_t)batch * max_tokens);
                    HIP_CHECK(hipMemcpyAsync(h_ng.data(), c->n_gen.p, batch * 4, hipMemcpyDeviceToHost, s));
                    HIP_CHECK(hipMemcpyAsync(h_tok.data(), c->tokens_out.p, h_tok.size() * 4, hipMemcpyDeviceToHost, s));
                }
                HIP_CHECK(hipStreamSynchronize(s));
                if (sampler_fail_flags_any(fail_pin.p, batch)) { sampler_failed = true; break; }
                if (on_event)
                    for (int b = 0; b < batch; ++b)
                        for (; emitted[b] < h_ng[b]; ++emitted[b]) {
                            int32_t t = h_tok[(size_t)b * max_tokens + emitted[b]];
                            if (t != sp->eot_id) on_event(user, b, MIS_EVENT_TOKEN, &t, 1);
                        }
                if (cancel_flag && *cancel_flag) throw MisError(MIS_ERR_CANCELLED, "transcription cancelled");
                if (*done_host >= batch) break;
            }
        }
    } catch (...) {
        if (gexec) (void)hipGraphExecDestroy(gexec);
        throw;
    }
    if (gexec) (void)hipGraphExecDestroy(gexec);
    HIP_CHECK(hipGetLastError());
    if (sampler_failed) sampler_note_failure(c->scratch.p, batch, s);
    return !sampler_failed;
    };
    if (!attempt(c->shared_device)) {
        MIS_REQUIRE(!c->shared_device, MIS_ERR_GENERATION_FAILED, "sampler: time-out flag raised on the multi-launch path");
        whisper_decoder_reset(c);
        MIS_REQUIRE(attempt(true), MIS_ERR_GENERATION_FAILED, "sampler: time-out flag raised on the multi-launch path");
    }
    std::vector<int32_t> ng(batch), toks((size_t)batch * max_tokens);
    HIP_CHECK(hipMemcpy(ng.data(), c->n_gen.p, batch * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(toks.data(), c->tokens_out.p, toks.size() * 4, hipMemcpyDeviceToHost));
    if (tokens_out) {
        PinnedBuf<int32_t> th(toks.size() + 1);
        memcpy(th.p, toks.data(), toks.size() * 4);
        *tokens_out = th.release();
        if (tokens_stride) *tokens_stride = max_tokens;
    }
    for (int b = 0; b < batch; ++b) {
        // the EOT token ends the row and is not part of `generated` (:238-240)
        int n = ng[b];
        if (n > 0 && toks[(size_t)b * max_tokens + n - 1] == sp->eot_id) n -= 1;
        if (n_tokens) n_tokens[b] = n;
        if (on_event) {                                   // per-row counts of the final .result (STTOutput promptTokens / generationTokens)
            mis_gen_info info{};
            info.prompt_token_count = n_prompt; info.generation_token_count = n;
            on_event(user, b, MIS_EVENT_INFO, &info, 1);
        }
    }
}

int whisper_internal_device(const mis_whisper* c) { return c ? c->device : -1; }
void whisper_internal_set_shared_device(mis_whisper* c, bool shared) { if (c) c->shared_device = shared; }

extern "C" mis_status mis_stt_whisper_generate(mis_whisper* c, const float* pcm, const int64_t* lens, int batch, int64_t stride,
                                               const int32_t* prompt_ids, int n_prompt, const mis_stt_params* sp,
                                               int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens) {
    MIS_API_BEGIN
    MIS_REQUIRE(tokens_out && tokens_stride && n_tokens, MIS_ERR_INVALID_INPUT, "null argument");
    whisper_generate_impl(c, pcm, lens, batch, stride, prompt_ids, n_prompt, sp, tokens_out, tokens_stride, n_tokens, nullptr, nullptr, nullptr);
    MIS_API_END
}

// generateStream (WhisperModel.swift:92-160) for a batch of windows: MIS_EVENT_TOKEN (row, id) while the greedy loop runs - the
// host turns ids into text deltas (onTokenDelta :186-250) - then one MIS_EVENT_INFO per row; cancel_flag is polled every few steps.
// tokens_out / tokens_stride / n_tokens may be NULL.
extern "C" mis_status mis_stt_whisper_generate_stream(mis_whisper* c, const float* pcm, const int64_t* lens, int batch, int64_t stride,
                                                      const int32_t* prompt_ids, int n_prompt, const mis_stt_params* sp,
                                                      mis_event_cb on_event, void* user, const volatile int* cancel_flag,
                                                      int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens) {
    MIS_API_BEGIN
    MIS_REQUIRE(on_event, MIS_ERR_INVALID_INPUT, "null callback");
    whisper_generate_impl(c, pcm, lens, batch, stride, prompt_ids, n_prompt, sp, tokens_out, tokens_stride, n_tokens, on_event, user, cancel_flag);
    MIS_API_END
}
