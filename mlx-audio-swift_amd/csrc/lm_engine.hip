// lm_engine.hip - host side of the Orpheus TTS engine: weight residency, the per-step kernel chain
// (hipGraph-captured), the batched generate loop and the hand-off to the SNAC codec.
//
// Reference being replaced: LlamaTTSModel (LlamaTTS.swift:354-977): fromModelDirectory :942-977,
// callAsFunction :557-567, generate :658-765, generateStream :777-913.  The reference loop costs one
// host<->device sync per token (.item() :723, eval :728); here a decode step is ONE hipGraph replay
// (lm_head -> sampler -> embed -> 28 x [qkv, attention, o, norm, gate/up, down, norm]) and the host only
// looks at a done-counter every few steps.
#include <deque>
#include <map>
#include "common.h"
#include "kernels.h"
#include "lm_kernels.h"

#include <math.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <functional>
#include <set>

#define ORPHEUS_END_OF_SPEECH 128258

struct mis_tts {
    int device = 0;
    hipStream_t stream = nullptr;
    mis_lm_config cfg{};
    mis_snac* codec = nullptr;
    int d = 0, L = 0, ff = 0, H = 0, Hkv = 0, D = 0, V = 0, Vpad = 0, Nqkv = 0;
    bool finalized = false, have_lm_head = false;
    std::set<std::string> loaded;

    // weights (HBM resident)
    DevBuf<bf16_t> emb;        // [V][d] row-major (gather)
    DevBuf<bf16_t> lm_head;    // packed [Vpad/16][d/32][64][8]
    DevBuf<bf16_t> wqkv;       // [L] packed [Nqkv/16][d/32]...
    DevBuf<bf16_t> wo;         // [L] packed [d/16][H*D/32]
    DevBuf<bf16_t> wgu;        // [L] packed [2*ff/16][d/32]  (gate/up tiles interleaved)
    DevBuf<bf16_t> wdown;      // [L] packed [d/16][ff/32]
    DevBuf<bf16_t> norms;      // [L][2][d] + [d]
    DevBuf<bf16_t> qknorm;     // [L][2][D] (qk_norm models)
    DevBuf<bf16_t> staging;    // row-major bf16 staging for one tensor
    DevBuf<uint8_t> raw_staging;
    // MLX affine-quantised checkpoints (8 / 4 bit, group 64, bf16 scales): a role whose every matrix arrived quantised the same way
    // is streamed in that form by k_gemm_skinny_q (lm_qgemm.hip) and its bf16 copy is dropped at finalize
    struct QRole {
        DevBuf<uint8_t> q;     // [L] packed codes, layer stride q_layer bytes
        DevBuf<bf16_t> sb;     // [L] packed scale / bias pairs, layer stride sb_layer elements
        size_t q_layer = 0, sb_layer = 0;
        int bits = 0;
        std::set<std::string> names;   // matrices that hold codes (distinct names: a duplicate set must not stand in for a missing one)
        bool bad = false, on = false;
    } q_qkv, q_o, q_gu, q_down, q_head;

    // per-batch state
    int batch = 0, Mpad = 0, Smax = 0;
    int S_qkv = 1, S_o = 1, S_down = 1;
    int r_part = 1;                                 // n-tiles per work item of the split-K GEMMs
    int r_gu = 2;                                   // n-tiles per wave of gate+up (4 where that still fills the chip, see lm_reset)
    int ksb_part = 4, ksb_gu = 4, ksb_head = 1;     // waves per work item (in-block split-K), see k_gemm_skinny
    GemmArr a_qkv, a_o, a_down, a_head;             // dense bf16 roles: the arrangement picked in lm_reset (gate+up: r_gu / ksb_gu)
    GemmArr qa_gu, qa_head;                         // code-streamed wide roles: (R, KSB) of k_gemm_skinny_q (MIS_QARR_GU / _HEAD = "R,KSB")
    DevBuf<bf16_t> kcache, vtcache;
    DevBuf<float> rope_cos, rope_sin;
    DevBuf<int32_t> ids, pos_cur, pos_next;
    DevBuf<uint8_t> active;
    DevBuf<bf16_t> h, x, attn_out, act, logits;
    DevBuf<float> qkv_part, part, e_buf, logits_f32, samp_l32;
    // generation state
    DevBuf<int32_t> prompt_mat, prompt_lens, step_counter, window, window_len, n_gen, tokens_out, all_ids, all_len,
        done_count, codes, n_codes, l0, l1, l2, row_map;
    DevBuf<float> pcm_tmp;
    // batched prefill (lm_prefill.hip): one chunk of positions x rows
    DevBuf<bf16_t> pf_h, pf_x, pf_attn, pf_act, pf_xpk, pf_actpk;
    DevBuf<float> pf_qkv, pf_tmp;
    DevBuf<int32_t> pf_pos;
    DevBuf<uint8_t> pf_on;
    DevBuf<SamplerScratch> samp_scratch;
    hipGraphExec_t g_prefill = nullptr, g_decode = nullptr;
    bool shared_device = false;      // another replica's streams run on this device (group.hip): never a kernel that waits for co-resident blocks
    uint64_t graph_key = 0;
    bool use_graph = true;
    bool borrowed_stream = false;
    int profiling = 0;
    mis_tts_timing timing{};
    SamplerParams sp{};
};

static size_t layer_qkv_elems(const mis_tts* c) { return (size_t)c->Nqkv * c->d; }
static size_t layer_o_elems(const mis_tts* c) { return (size_t)c->d * c->H * c->D; }
static size_t layer_gu_elems(const mis_tts* c) { return (size_t)2 * c->ff * c->d; }
static size_t layer_down_elems(const mis_tts* c) { return (size_t)c->d * c->ff; }

extern "C" mis_status mis_tts_create(const mis_lm_config* cfg, mis_snac* codec, int device, mis_tts** out) {
    MIS_API_BEGIN
    MIS_REQUIRE(cfg && out, MIS_ERR_INVALID_INPUT, "null argument");
    int n = 0;
    HIP_CHECK(hipGetDeviceCount(&n));
    MIS_REQUIRE(device >= 0 && device < n, MIS_ERR_DEVICE, "device %d not available (%d GPUs visible)", device, n);
    MIS_REQUIRE(!codec || snac_device(codec) == device, MIS_ERR_INVALID_INPUT, "codec lives on another device");
    const int d = cfg->hidden_size, H = cfg->num_attention_heads, Hkv = cfg->num_key_value_heads;
    const int D = cfg->head_dim > 0 ? cfg->head_dim : (H > 0 ? d / H : 0);
    MIS_REQUIRE(d > 0 && cfg->num_hidden_layers > 0 && cfg->intermediate_size > 0 && H > 0 && Hkv > 0 && cfg->vocab_size > 0,
                MIS_ERR_INVALID_INPUT, "bad model dimensions");
    MIS_REQUIRE(D == 64 || D == 128, MIS_ERR_INVALID_INPUT, "head_dim %d unsupported (64 or 128)", D);
    MIS_REQUIRE(H % Hkv == 0 && H / Hkv <= 16, MIS_ERR_INVALID_INPUT, "unsupported GQA grouping %d/%d", H, Hkv);
    MIS_REQUIRE(d % 32 == 0 && cfg->intermediate_size % 32 == 0 && (H * D) % 32 == 0, MIS_ERR_INVALID_INPUT,
                "hidden/intermediate sizes must be multiples of 32");
    HIP_CHECK(hipSetDevice(device));
    mis_tts* c = new mis_tts();
    c->device = device;
    c->cfg = *cfg;
    c->cfg.head_dim = D;
    if (c->cfg.rope_factor <= 0) c->cfg.rope_factor = 32.0f;            // LlamaTTS.swift:181-184 defaults
    if (c->cfg.rope_low_freq_factor <= 0) c->cfg.rope_low_freq_factor = 1.0f;
    if (c->cfg.rope_high_freq_factor <= 0) c->cfg.rope_high_freq_factor = 4.0f;
    if (c->cfg.rope_original_max_pos <= 0) c->cfg.rope_original_max_pos = 8192.0f;
    if (c->cfg.rope_theta <= 0) c->cfg.rope_theta = 10000.0f;
    if (c->cfg.sample_rate <= 0) c->cfg.sample_rate = 24000;
    c->codec = codec;
    c->d = d; c->L = cfg->num_hidden_layers; c->ff = cfg->intermediate_size; c->H = H; c->Hkv = Hkv; c->D = D;
    c->V = cfg->vocab_size; c->Vpad = (int)round_up(c->V, 16); c->Nqkv = (H + 2 * Hkv) * D;
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->emb.alloc((size_t)c->V * d);
    c->lm_head.alloc((size_t)c->Vpad * d);
    c->wqkv.alloc(layer_qkv_elems(c) * c->L);
    c->wo.alloc(layer_o_elems(c) * c->L);
    c->wgu.alloc(layer_gu_elems(c) * c->L);
    c->wdown.alloc(layer_down_elems(c) * c->L);
    c->norms.alloc((size_t)(2 * c->L + 1) * d);
    if (c->cfg.qk_norm) c->qknorm.alloc((size_t)2 * c->L * D);
    c->use_graph = getenv("MIS_NO_GRAPH") == nullptr;
    *out = c;
    MIS_API_END
}

extern "C" void mis_tts_destroy(mis_tts* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->g_prefill) (void)hipGraphExecDestroy(c->g_prefill);
    if (c->g_decode) (void)hipGraphExecDestroy(c->g_decode);
    if (c->stream && !c->borrowed_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// staging (row-major bf16 [N][K]) -> destination by tensor role
static void place_matrix(mis_tts* c, const std::string& name, int64_t N, int64_t K) {
    hipStream_t s = c->stream;
    auto starts = [&](const char* p) { return name.rfind(p, 0) == 0; };
    if (name == "model.embed_tokens.weight") {
        MIS_REQUIRE(N == c->V && K == c->d, MIS_ERR_INVALID_INPUT, "embed_tokens shape mismatch");
        HIP_CHECK(hipMemcpyAsync(c->emb.p, c->staging.p, (size_t)N * K * 2, hipMemcpyDeviceToDevice, s));
        return;
    }
    if (name == "lm_head.weight") {
        MIS_REQUIRE(N == c->V && K == c->d, MIS_ERR_INVALID_INPUT, "lm_head shape mismatch");
        launch_pack_weight(c->staging.p, c->lm_head.p, c->V, c->d, c->Vpad / 16, 1, 0, s);
        c->have_lm_head = true;
        return;
    }
    MIS_REQUIRE(starts("model.layers."), MIS_ERR_INVALID_INPUT, "unexpected tensor %s", name.c_str());
    size_t p1 = strlen("model.layers.");
    size_t p2 = name.find('.', p1);
    MIS_REQUIRE(p2 != std::string::npos, MIS_ERR_INVALID_INPUT, "bad tensor name %s", name.c_str());
    int li = atoi(name.substr(p1, p2 - p1).c_str());
    MIS_REQUIRE(li >= 0 && li < c->L, MIS_ERR_INVALID_INPUT, "layer index out of range in %s", name.c_str());
    std::string rest = name.substr(p2 + 1);
    const int HD = c->H * c->D, KD = c->Hkv * c->D;
    if (rest == "self_attn.q_proj.weight") {
        MIS_REQUIRE(N == HD && K == c->d, MIS_ERR_INVALID_INPUT, "%s shape mismatch", name.c_str());
        bf16_t* dst = c->wqkv.p + layer_qkv_elems(c) * li;
        launch_pack_weight(c->staging.p, dst, HD, c->d, HD / 16, 1, 0, s);
    } else if (rest == "self_attn.k_proj.weight") {
        MIS_REQUIRE(N == KD && K == c->d, MIS_ERR_INVALID_INPUT, "%s shape mismatch", name.c_str());
        bf16_t* dst = c->wqkv.p + layer_qkv_elems(c) * li;
        launch_pack_weight(c->staging.p, dst, KD, c->d, KD / 16, 1, HD / 16, s);
    } else if (rest == "self_attn.v_proj.weight") {
        MIS_REQUIRE(N == KD && K == c->d, MIS_ERR_INVALID_INPUT, "%s shape mismatch", name.c_str());
        bf16_t* dst = c->wqkv.p + layer_qkv_elems(c) * li;
        launch_pack_weight(c->staging.p, dst, KD, c->d, KD / 16, 1, (HD + KD) / 16, s);
    } else if (rest == "self_attn.o_proj.weight") {
        MIS_REQUIRE(N == c->d && K == HD, MIS_ERR_INVALID_INPUT, "%s shape mismatch", name.c_str());
        launch_pack_weight(c->staging.p, c->wo.p + layer_o_elems(c) * li, c->d, HD, c->d / 16, 1, 0, s);
    } else if (rest == "mlp.gate_proj.weight" || rest == "mlp.up_proj.weight") {
        MIS_REQUIRE(N == c->ff && K == c->d, MIS_ERR_INVALID_INPUT, "%s shape mismatch", name.c_str());
        launch_pack_weight(c->staging.p, c->wgu.p + layer_gu_elems(c) * li, c->ff, c->d, c->ff / 16, 2,
                           rest == "mlp.gate_proj.weight" ? 0 : 1, s);
    } else if (rest == "mlp.down_proj.weight") {
        MIS_REQUIRE(N == c->d && K == c->ff, MIS_ERR_INVALID_INPUT, "%s shape mismatch", name.c_str());
        launch_pack_weight(c->staging.p, c->wdown.p + layer_down_elems(c) * li, c->d, c->ff, c->d / 16, 1, 0, s);
    } else {
        throw MisError(MIS_ERR_INVALID_INPUT, "unexpected tensor " + name);
    }
}

// the same matrix in its quantised form (device pointers: MLX layout) -> the role's packed code / scale buffers
static void place_qmatrix(mis_tts* c, const std::string& name, int64_t N, int64_t K, int bits, const uint32_t* wq, const bf16_t* sc,
                          const bf16_t* bi) {
    hipStream_t s = c->stream;
    const int HD = c->H * c->D, KD = c->Hkv * c->D;
    auto put = [&](mis_tts::QRole& R, int li, int n_layers, int64_t Ntot, int tile_stride, int tile_offset) {
        if (R.bits && R.bits != bits) { R.bad = true; return; }
        const size_t NTtot = round_up(Ntot, 16) / 16, KT = K / 32, G = K / 64;
        const size_t q_layer = NTtot * KT * 64 * bits, sb_layer = NTtot * G * 32;
        if (!R.bits) {
            R.bits = bits; R.q_layer = q_layer; R.sb_layer = sb_layer;
            R.q.alloc(q_layer * n_layers); R.sb.alloc(sb_layer * n_layers);
            HIP_CHECK(hipMemsetAsync(R.q.p, 0, q_layer * n_layers, s));          // rows past N (vocabulary padding) decode to 0
            HIP_CHECK(hipMemsetAsync(R.sb.p, 0, sb_layer * n_layers * 2, s));
        }
        if (R.q_layer != q_layer) { R.bad = true; return; }
        launch_pack_qweight(bits, wq, sc, bi, R.q.p + q_layer * li, R.sb.p + sb_layer * li, (int)N, (int)K, tile_stride, tile_offset, s);
        R.names.insert(name);
    };
    if (name == "lm_head.weight") { put(c->q_head, 0, 1, c->V, 1, 0); return; }
    if (name.rfind("model.layers.", 0) != 0) return;
    size_t p1 = strlen("model.layers."), p2 = name.find('.', p1);
    if (p2 == std::string::npos) return;
    const int li = atoi(name.substr(p1, p2 - p1).c_str());
    if (li < 0 || li >= c->L) return;
    const std::string rest = name.substr(p2 + 1);
    if (rest == "self_attn.q_proj.weight") put(c->q_qkv, li, c->L, c->Nqkv, 1, 0);
    else if (rest == "self_attn.k_proj.weight") put(c->q_qkv, li, c->L, c->Nqkv, 1, HD / 16);
    else if (rest == "self_attn.v_proj.weight") put(c->q_qkv, li, c->L, c->Nqkv, 1, (HD + KD) / 16);
    else if (rest == "self_attn.o_proj.weight") put(c->q_o, li, c->L, c->d, 1, 0);
    else if (rest == "mlp.gate_proj.weight") put(c->q_gu, li, c->L, 2 * c->ff, 2, 0);
    else if (rest == "mlp.up_proj.weight") put(c->q_gu, li, c->L, 2 * c->ff, 2, 1);
    else if (rest == "mlp.down_proj.weight") put(c->q_down, li, c->L, c->d, 1, 0);
}

// a matrix that arrives in a form the code-streaming kernels cannot take (dense set_tensor, 2 bit, other group sizes / scale dtypes)
// AFTER its role already holds codes for it: only the bf16 copy was updated, so the role's codes are stale - stream the bf16 copy
static void mark_dense_override(mis_tts* c, const std::string& name) {
    for (mis_tts::QRole* R : {&c->q_qkv, &c->q_o, &c->q_gu, &c->q_down, &c->q_head})
        if (R->names.count(name)) R->bad = true;
}

static bf16_t* norm_slot(mis_tts* c, const std::string& name) {
    if (name == "model.norm.weight") return c->norms.p + (size_t)2 * c->L * c->d;
    size_t p1 = strlen("model.layers.");
    if (name.rfind("model.layers.", 0) != 0) return nullptr;
    size_t p2 = name.find('.', p1);
    if (p2 == std::string::npos) return nullptr;
    int li = atoi(name.substr(p1, p2 - p1).c_str());
    if (li < 0 || li >= c->L) return nullptr;
    std::string rest = name.substr(p2 + 1);
    if (c->cfg.qk_norm && rest == "self_attn.q_norm.weight") return c->qknorm.p + (size_t)(2 * li) * c->D;
    if (c->cfg.qk_norm && rest == "self_attn.k_norm.weight") return c->qknorm.p + (size_t)(2 * li + 1) * c->D;
    if (rest == "input_layernorm.weight") return c->norms.p + (size_t)(2 * li) * c->d;
    if (rest == "post_attention_layernorm.weight") return c->norms.p + (size_t)(2 * li + 1) * c->d;
    return nullptr;
}

extern "C" mis_status mis_tts_set_tensor(mis_tts* c, const char* name_, const void* data, mis_dtype dtype,
                                         const int64_t* shape, int ndim) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && name_ && data && shape, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(!c->finalized, MIS_ERR_INVALID_INPUT, "set_tensor after finalize");
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_F16 || dtype == MIS_BF16, MIS_ERR_INVALID_INPUT, "unsupported dtype");
    std::string name = name_;
    if (name.find("rotary_emb.inv_freq") != std::string::npos) return MIS_OK;      // sanitize, LlamaTTS.swift:584-586
    if (name == "lm_head.weight" && c->cfg.tie_word_embeddings) return MIS_OK;     // :588-590
    HIP_CHECK(hipSetDevice(c->device));
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { MIS_REQUIRE(shape[i] > 0, MIS_ERR_INVALID_INPUT, "bad shape"); n *= (size_t)shape[i]; }
    size_t esz = dtype == MIS_F32 ? 4 : 2;
    c->raw_staging.alloc(n * esz);
    c->staging.alloc(n);
    HIP_CHECK(hipMemcpyAsync(c->raw_staging.p, data, n * esz, hipMemcpyDefault, c->stream));
    if (ndim == 1) {
        bf16_t* slot = norm_slot(c, name);
        bool is_qk = name.find("_norm.weight") != std::string::npos && name.find("self_attn.") != std::string::npos;
        MIS_REQUIRE(slot && (int64_t)n == (is_qk ? c->D : c->d), MIS_ERR_INVALID_INPUT, "unexpected 1-D tensor %s", name.c_str());
        launch_convert_to_bf16(c->raw_staging.p, dtype, slot, n, c->stream);
    } else {
        MIS_REQUIRE(ndim == 2, MIS_ERR_INVALID_INPUT, "tensor %s must be 1-D or 2-D", name.c_str());
        launch_convert_to_bf16(c->raw_staging.p, dtype, c->staging.p, n, c->stream);
        place_matrix(c, name, shape[0], shape[1]);
        mark_dense_override(c, name);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(c->stream));      // staging buffers are reused by the next call
    c->loaded.insert(name);
    MIS_API_END
}

// A Linear / Embedding stored in MLX's affine-quantised form (`weight` uint32 [N, K*bits/32], `scales` / `biases` [N, K/group],
// written by mlx quantize; the reference re-creates QuantizedLinear modules for every path that has `.scales`,
// LlamaTTS.swift:958-968).  The matrix is dequantised once at load into the engine's bf16 layouts.
static void set_quantized_any(mis_tts* c, const std::string& name, const uint32_t* wq, const void* scales, const void* biases,
                              mis_dtype sb_dtype, int64_t N, int64_t K, int group_size, int bits) {
    MIS_REQUIRE(bits == 2 || bits == 4 || bits == 8, MIS_ERR_INVALID_INPUT, "unsupported quantisation width %d (2, 4 or 8 bits)", bits);
    MIS_REQUIRE(group_size >= 1 && N >= 1 && K >= 1 && K % group_size == 0 && K % (32 / bits) == 0, MIS_ERR_INVALID_INPUT,
                "bad quantised shape for %s", name.c_str());
    MIS_REQUIRE(sb_dtype == MIS_F32 || sb_dtype == MIS_F16 || sb_dtype == MIS_BF16, MIS_ERR_INVALID_INPUT, "unsupported scale dtype");
    if (name == "lm_head.weight" && c->cfg.tie_word_embeddings) return;
    HIP_CHECK(hipSetDevice(c->device));
    const size_t words = (size_t)N * K * bits / 32, ng = (size_t)N * (K / group_size), esz = sb_dtype == MIS_F32 ? 4 : 2;
    const size_t wb = round_up(words * 4, 16), sb = round_up(ng * esz, 16);
    c->raw_staging.alloc(wb + 2 * sb);
    uint8_t* p = c->raw_staging.p;
    uint8_t* ps = p + wb;
    uint8_t* pb = ps + sb;
    HIP_CHECK(hipMemcpyAsync(p, wq, words * 4, hipMemcpyDefault, c->stream));
    HIP_CHECK(hipMemcpyAsync(ps, scales, ng * esz, hipMemcpyDefault, c->stream));
    HIP_CHECK(hipMemcpyAsync(pb, biases, ng * esz, hipMemcpyDefault, c->stream));
    c->staging.alloc((size_t)N * K);
    launch_dequant_affine((const uint32_t*)p, ps, pb, (int)sb_dtype, c->staging.p, (int)N, (int)K, group_size, bits, c->stream);
    place_matrix(c, name, N, K);
    // native form (see QRole): bf16 scales are exact in the kernel's float32 arithmetic; anything else keeps the bf16 copy only
    static const bool native = !(getenv("MIS_QUANT_NATIVE") && atoi(getenv("MIS_QUANT_NATIVE")) == 0);
    if (native && (bits == 8 || bits == 4) && group_size == 64 && sb_dtype == MIS_BF16 && K % 64 == 0 && (N % 16 == 0 || name == "lm_head.weight"))
        place_qmatrix(c, name, N, K, bits, (const uint32_t*)p, (const bf16_t*)ps, (const bf16_t*)pb);
    else mark_dense_override(c, name);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(c->stream));
    c->loaded.insert(name);
}

extern "C" mis_status mis_tts_set_tensor_quantized(mis_tts* c, const char* name_, const uint32_t* wq, const void* scales,
                                                   const void* biases, mis_dtype sb_dtype, int64_t N, int64_t K, int group_size,
                                                   int bits) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && name_ && wq && scales && biases, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(!c->finalized, MIS_ERR_INVALID_INPUT, "set_tensor after finalize");
    set_quantized_any(c, name_, wq, scales, biases, sb_dtype, N, K, group_size, bits);
    MIS_API_END
}

// bits of the quantised form a role is streamed in after finalize (0 = dense bf16): role 0 qkv, 1 o_proj, 2 gate/up, 3 down, 4 lm_head
extern "C" int mis_tts_native_quant_bits(const mis_tts* c, int role) {
    if (!c) return 0;
    const mis_tts::QRole* R[5] = {&c->q_qkv, &c->q_o, &c->q_gu, &c->q_down, &c->q_head};
    return (role >= 0 && role < 5 && R[role]->on) ? R[role]->bits : 0;
}

// benches: every Linear as a synthetic MLX-quantised matrix (random codes, group scale ~ 2 amp / (2^bits - 1), bias ~ -amp), group 64,
// bf16 scales - there are no checkpoints offline.  The embedding stays dense.
__global__ void k_synth_quant(uint32_t* __restrict__ wq, bf16_t* __restrict__ sc, bf16_t* __restrict__ bi, size_t words, size_t groups,
                              uint64_t key, float amp, int bits) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) wq[i] = (uint32_t)(mis_splitmix64(key * 0x9E3779B97F4A7C15ull + i) >> 16);
    if (i < groups) {
        const float jitter = 1.0f + 0.25f * mis_synth_value(key ^ 0x5bd1e995ull, i, 1.0f);
        sc[i] = f32_to_bf16(2.0f * amp * jitter / (float)((1 << bits) - 1));
        bi[i] = f32_to_bf16(-amp * jitter);
    }
}
extern "C" mis_status mis_tts_init_synthetic_quantized(mis_tts* c, uint64_t seed, int bits) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && !c->finalized && (bits == 4 || bits == 8), MIS_ERR_INVALID_INPUT, "bad argument");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int d = c->d, ff = c->ff, HD = c->H * c->D, KD = c->Hkv * c->D;
    const uint64_t base = seed * 100000ull;
    DevBuf<uint32_t> wq; DevBuf<bf16_t> sc, bi;
    auto mat = [&](const std::string& name, uint64_t key, int64_t N, int64_t K, double amp) {
        const size_t words = (size_t)N * K * bits / 32, groups = (size_t)N * (K / 64);
        wq.alloc(words); sc.alloc(groups); bi.alloc(groups);
        hipLaunchKernelGGL(k_synth_quant, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, wq.p, sc.p, bi.p, words, groups, base + key, (float)amp, bits);
        HIP_CHECK(hipStreamSynchronize(s));
        set_quantized_any(c, name, wq.p, sc.p, bi.p, MIS_BF16, N, K, 64, bits);
    };
    auto vec = [&](const std::string& name, uint64_t key) {
        bool is_qk = name.find("self_attn.") != std::string::npos;
        launch_synth_fill_bf16(norm_slot(c, name), (size_t)(is_qk ? c->D : d), base + key, 0.1f, 1, s);
        c->loaded.insert(name);
    };
    c->staging.alloc((size_t)c->V * d);
    launch_synth_fill_bf16(c->staging.p, (size_t)c->V * d, base + 1, (float)(0.5 * sqrt(3.0)), 0, s);
    place_matrix(c, "model.embed_tokens.weight", c->V, d);
    c->loaded.insert("model.embed_tokens.weight");
    HIP_CHECK(hipStreamSynchronize(s));
    vec("model.norm.weight", 2);
    if (!c->cfg.tie_word_embeddings) mat("lm_head.weight", 3, c->V, d, sqrt(3.0 / d) * 2.0);
    for (int li = 0; li < c->L; ++li) {
        std::string p = "model.layers." + std::to_string(li);
        uint64_t k = 100 + (uint64_t)li * 16;
        vec(p + ".input_layernorm.weight", k + 0);
        vec(p + ".post_attention_layernorm.weight", k + 1);
        mat(p + ".self_attn.q_proj.weight", k + 2, HD, d, sqrt(3.0 / d) * 1.5);
        mat(p + ".self_attn.k_proj.weight", k + 3, KD, d, sqrt(3.0 / d) * 1.5);
        mat(p + ".self_attn.v_proj.weight", k + 4, KD, d, sqrt(3.0 / d));
        mat(p + ".self_attn.o_proj.weight", k + 5, d, HD, sqrt(3.0 / HD) * 0.5);
        mat(p + ".mlp.gate_proj.weight", k + 6, ff, d, sqrt(3.0 / d));
        mat(p + ".mlp.up_proj.weight", k + 7, ff, d, sqrt(3.0 / d));
        mat(p + ".mlp.down_proj.weight", k + 8, d, ff, sqrt(3.0 / ff) * 0.5);
        if (c->cfg.qk_norm) { vec(p + ".self_attn.q_norm.weight", k + 9); vec(p + ".self_attn.k_norm.weight", k + 10); }
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(s));
    MIS_API_END
}

extern "C" mis_status mis_tts_init_synthetic(mis_tts* c, uint64_t seed) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && !c->finalized, MIS_ERR_INVALID_INPUT, "bad handle");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int d = c->d, ff = c->ff, HD = c->H * c->D, KD = c->Hkv * c->D;
    const uint64_t base = seed * 100000ull;
    size_t big = std::max((size_t)c->V * d, (size_t)ff * d);
    c->staging.alloc(big);
    // keys / amplitudes: oracle/llama.py make_synthetic_weights
    auto mat = [&](const std::string& name, uint64_t key, int64_t N, int64_t K, double amp) {
        launch_synth_fill_bf16(c->staging.p, (size_t)N * K, base + key, (float)amp, 0, s);
        place_matrix(c, name, N, K);
        c->loaded.insert(name);
    };
    auto vec = [&](const std::string& name, uint64_t key) {
        bool is_qk = name.find("self_attn.") != std::string::npos;
        launch_synth_fill_bf16(norm_slot(c, name), (size_t)(is_qk ? c->D : d), base + key, 0.1f, 1, s);
        c->loaded.insert(name);
    };
    mat("model.embed_tokens.weight", 1, c->V, d, 0.5 * sqrt(3.0));
    vec("model.norm.weight", 2);
    if (!c->cfg.tie_word_embeddings) mat("lm_head.weight", 3, c->V, d, sqrt(3.0 / d) * 2.0);
    for (int li = 0; li < c->L; ++li) {
        std::string p = "model.layers." + std::to_string(li);
        uint64_t k = 100 + (uint64_t)li * 16;
        vec(p + ".input_layernorm.weight", k + 0);
        vec(p + ".post_attention_layernorm.weight", k + 1);
        mat(p + ".self_attn.q_proj.weight", k + 2, HD, d, sqrt(3.0 / d) * 1.5);
        mat(p + ".self_attn.k_proj.weight", k + 3, KD, d, sqrt(3.0 / d) * 1.5);
        mat(p + ".self_attn.v_proj.weight", k + 4, KD, d, sqrt(3.0 / d));
        mat(p + ".self_attn.o_proj.weight", k + 5, d, HD, sqrt(3.0 / HD) * 0.5);
        mat(p + ".mlp.gate_proj.weight", k + 6, ff, d, sqrt(3.0 / d));
        mat(p + ".mlp.up_proj.weight", k + 7, ff, d, sqrt(3.0 / d));
        mat(p + ".mlp.down_proj.weight", k + 8, d, ff, sqrt(3.0 / ff) * 0.5);
        if (c->cfg.qk_norm) { vec(p + ".self_attn.q_norm.weight", k + 9); vec(p + ".self_attn.k_norm.weight", k + 10); }
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(s));
    MIS_API_END
}

extern "C" mis_status mis_tts_finalize(mis_tts* c) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && !c->finalized, MIS_ERR_INVALID_INPUT, "bad handle");
    HIP_CHECK(hipSetDevice(c->device));
    // update(parameters:verify:.all), LlamaTTS.swift:971: every parameter must be present
    std::vector<std::string> want = {"model.embed_tokens.weight", "model.norm.weight"};
    if (!c->cfg.tie_word_embeddings) want.push_back("lm_head.weight");
    const char* per_layer[] = {"input_layernorm.weight", "post_attention_layernorm.weight", "self_attn.q_proj.weight",
                               "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight",
                               "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight"};
    for (int li = 0; li < c->L; ++li) {
        for (const char* r : per_layer) want.push_back("model.layers." + std::to_string(li) + "." + r);
        if (c->cfg.qk_norm) {
            want.push_back("model.layers." + std::to_string(li) + ".self_attn.q_norm.weight");
            want.push_back("model.layers." + std::to_string(li) + ".self_attn.k_norm.weight");
        }
    }
    for (auto& w : want)
        MIS_REQUIRE(c->loaded.count(w), MIS_ERR_NOT_INITIALIZED, "LM weight missing: %s", w.c_str());
    if (c->cfg.tie_word_embeddings)        // embedTokens.asLinear, LlamaTTS.swift:563
        launch_pack_weight(c->emb.p, c->lm_head.p, c->V, c->d, c->Vpad / 16, 1, 0, c->stream);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(c->stream));
    {   // quantised roles: complete and uniform -> stream the codes, drop the bf16 copy
        auto decide = [&](mis_tts::QRole& R, int expected, DevBuf<bf16_t>& dense) {
            R.on = R.bits != 0 && !R.bad && (int)R.names.size() == expected;
            if (R.on) dense.release();
            else { R.q.release(); R.sb.release(); R.bits = 0; }
        };
        decide(c->q_qkv, 3 * c->L, c->wqkv);
        decide(c->q_o, c->L, c->wo);
        decide(c->q_gu, 2 * c->L, c->wgu);
        decide(c->q_down, c->L, c->wdown);
        if (c->cfg.tie_word_embeddings) { c->q_head.on = false; c->q_head.q.release(); c->q_head.sb.release(); }
        else decide(c->q_head, 1, c->lm_head);
    }
    c->staging.release();
    c->raw_staging.release();
    c->finalized = true;
    MIS_API_END
}

// ---------------------------------------------------------------------------- per-batch state
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
// inter-block split-K factor: aim at ~800 blocks (3 resident blocks per CU, one wave of blocks; measured
// optimum of the R/KSB/S sweep, profiles/r01_v2_gemm_sweep.json), keep >= 2 k-tiles per wave
static int choose_split(int items, int KT, int ksb, const char* env, int s_max = 16) {
    return std::max(1, std::min(env_int(env, gemm_choose_split(items, KT, ksb, s_max)), 16));
}

static void destroy_graphs(mis_tts* c) {
    if (c->g_prefill) { (void)hipGraphExecDestroy(c->g_prefill); c->g_prefill = nullptr; }
    if (c->g_decode) { (void)hipGraphExecDestroy(c->g_decode); c->g_decode = nullptr; }
}

static void build_rope_tables(mis_tts* c) {
    // Llama3ScaledRoPE freqs, LlamaTTS.swift:126-156, float32 arithmetic; angle = pos / freqs[i]
    const int D = c->D, half = D / 2;
    std::vector<float> inv(half);
    const float base = c->cfg.rope_theta, factor = c->cfg.rope_factor, low = c->cfg.rope_low_freq_factor,
                high = c->cfg.rope_high_freq_factor, old = c->cfg.rope_original_max_pos;
    for (int i = 0; i < half; ++i) {
        float expo = (float)(2 * i) / (float)D;
        float f = powf(base, expo);
        float wl = (float)(2.0f * (float)M_PI) * f;
        float low_wl = old / low, high_wl = old / high;
        float fs = (wl > low_wl) ? f * factor : f;
        bool med = (wl > high_wl) && (wl < low_wl);
        float smooth = (old / wl - low) / (high - low);
        float denom = (1.0f - smooth) / factor + smooth;
        float ff = med ? fs / denom : fs;
        if (c->cfg.rope_plain) ff = f;                 // MLXNN.RoPE(dimensions, base): no rescale (Soprano.swift:56-61)
        inv[i] = 1.0f / ff;
    }
    std::vector<float> cs((size_t)c->Smax * half), sn((size_t)c->Smax * half);
    for (int p = 0; p < c->Smax; ++p)
        for (int i = 0; i < half; ++i) {
            float ang = (float)p * inv[i];
            cs[(size_t)p * half + i] = (float)cos((double)ang);
            sn[(size_t)p * half + i] = (float)sin((double)ang);
        }
    c->rope_cos.alloc(cs.size());
    c->rope_sin.alloc(sn.size());
    HIP_CHECK(hipMemcpyAsync(c->rope_cos.p, cs.data(), cs.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(c->rope_sin.p, sn.data(), sn.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}

static void lm_reset(mis_tts* c, int batch, int max_context) {
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "model not finalized");
    MIS_REQUIRE(batch >= 1 && batch <= 64, MIS_ERR_INVALID_INPUT, "batch per GPU must be 1..64 (got %d)", batch);
    MIS_REQUIRE(max_context >= 1, MIS_ERR_INVALID_INPUT, "max_context must be positive");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    int Mpad = (int)round_up(batch, 16);
    int Smax = (int)round_up(max_context, 64);
    bool regraph = (batch != c->batch || Mpad != c->Mpad || Smax != c->Smax);
    if (regraph) destroy_graphs(c);
    bool new_tables = Smax != c->Smax;
    c->batch = batch; c->Mpad = Mpad; c->Smax = Smax;
    const int d = c->d, HD = c->H * c->D;
    c->ksb_part = 4;                                     // split-K roles and gate+up: the four waves of a block share an item's K range
    c->ksb_gu = 4;
    // output projection: a big vocabulary fills the chip with one wave per tile pair (Orpheus: 4904 pairs); a codec-sized one (Qwen3-TTS
    // 2048 / 3072 ids, Soprano) has 64-96 pairs = 16-24 blocks, each wave walking the whole K range alone: there the four waves of a
    // block split K (10.1 -> see profiles/r03/q3_small_kernels.json).
    c->ksb_head = c->Vpad / 32 < 1024 ? 4 : 1;
    c->r_part = 2;
    // gate+up: four n-tiles per wave halve the x fragments every wave re-reads out of L2 (as many bytes as the weights at two) - 20.3 ->
    // 17.6 us at Orpheus-3B width (profiles/r03/gemm_lab.jsonl) - where a launch still has one wave per SIMD: tile quads x 4 waves >= 4 per CU
    {
        const int quads = 2 * c->ff / 16 / 4;
        c->r_gu = (Mpad / 16 <= 2 && c->ksb_gu == 4 && (2 * c->ff / 16) % 4 == 0 && quads >= 256) ? 4 : 2;
    }
    c->S_qkv = std::min(8, choose_split(c->Nqkv / 16 / c->r_part, d / 32, c->ksb_part, "MIS_S_QKV", 8));   // attention prologue: <= 8 slabs
    c->S_o = choose_split(d / 16 / c->r_part, HD / 32, c->ksb_part, "MIS_S_O");
    c->S_down = choose_split(d / 16 / c->r_part, c->ff / 32, c->ksb_part, "MIS_S_DOWN");
    // Dense roles: (R, KSB, U, S) per role.  Defaults = what tools/gemm_lab measured at Orpheus-3B widths and 32 rows (profiles/r04/);
    // four n-tiles per wave only up to 32 rows (8 accumulator tiles) and where the launch keeps every CU busy.  MIS_ARR_QKV / _O /
    // _DOWN / _HEAD = "R,KSB,U[,S]" overrides one role (A/B).
    {
        const int mt = Mpad / 16;
        auto pick = [&](const char* env, GemmArr dflt) {
            const char* e = getenv(env);
            GemmArr a = dflt;
            if (e && *e) {
                int r = 0, k = 0, u = 0, sp = 0;
                const int n = sscanf(e, "%d,%d,%d,%d", &r, &k, &u, &sp);
                if (n >= 3) { a.R = r; a.ksb = k; a.U = u; if (n >= 4 && sp >= 1) a.S = sp; }
            }
            if (a.R == 4 && mt > 2) a = dflt.R == 4 ? GemmArr{2, dflt.ksb == 2 ? 4 : dflt.ksb, 4, dflt.S} : dflt;
            return a;
        };
        GemmArr q{c->r_part, c->ksb_part, 4, c->S_qkv}, o{c->r_part, c->ksb_part, 4, c->S_o}, dn{c->r_part, c->ksb_part, 4, c->S_down},
            hd{2, c->ksb_head, 4, 1};
        if (mt <= 2) {
            // qkv: four n-tiles per wave (the x fragments re-read half as often), two waves per item: 8.56 -> 7.71 us in the lab, step
            // 2.125 -> 2.110 (R4 KSB4) -> 2.102 ms (R4 KSB2) on one box (profiles/r04/c4_ab.json)
            if ((c->Nqkv / 16) % 4 == 0 && (c->Nqkv / 16 / 4) * q.S >= 192 && c->ksb_part == 4) { q.R = 4; q.ksb = 2; q.U = 2; }
            // (attention output projection: R4 KSB2 U3 with twice the split was 6.59 -> 5.90 us in the lab and +7 us per STEP in the
            // chain - its four slabs cost the glue launch behind it what the GEMM gained; it stays R2 KSB4 at the cost model's split)
            // down projection: two waves per item (128-thread blocks) 11.8 -> 10.9 us in the lab, step 2.105 -> 2.082 ms (c2_ab.json)
            if (c->ksb_part == 4 && c->ff / 32 >= 64) { dn.ksb = 2; }
            // output projection of a big vocabulary: 173.5 -> 150.7 us (the four waves of a block split K)
            if (c->Vpad / 64 >= 1024) { hd.R = 4; hd.ksb = 4; hd.U = 3; }
        }
        c->a_qkv = pick("MIS_ARR_QKV", q); c->a_o = pick("MIS_ARR_O", o); c->a_down = pick("MIS_ARR_DOWN", dn); c->a_head = pick("MIS_ARR_HEAD", hd);
        {   // quantised checkpoints, the two wide roles (gate+up, output projection)
            auto qpick = [&](const char* env, GemmArr dflt) {
                const char* e = getenv(env);
                int r = 0, k = 0;
                if (e && sscanf(e, "%d,%d", &r, &k) == 2 && (r == 2 || r == 4) && (k == 1 || k == 4 || k == 8)) { dflt.R = r; dflt.ksb = k; }
                if ((dflt.R == 4 || dflt.ksb == 8) && mt > 2) { dflt.R = 2; dflt.ksb = dflt.ksb == 8 ? 4 : dflt.ksb; }
                return dflt;
            };
            // measured at Orpheus-3B widths, 8 bit, 32 rows (profiles/r04/c6_qgemm_arr.txt): gate+up 17.80 us as R2 KSB4, 17.37 as R4 KSB4,
            // 16.63 as R4 KSB8 (19.32 as R2 KSB8); output projection 136.2 us as R2 KSB1, 127.5 as R4 KSB4 (148.1 as R4 KSB8)
            GemmArr gu{2, c->ksb_gu, 0, 1}, hd{2, c->ksb_head, 0, 1};
            if (mt <= 2) {
                if (c->ksb_gu == 4 && (2 * c->ff / 16) % 4 == 0 && 2 * c->ff / 16 / 4 >= 256) { gu.R = 4; gu.ksb = 8; }
                if (c->Vpad / 64 >= 1024) { hd.R = 4; hd.ksb = 4; }
            }
            c->qa_gu = qpick("MIS_QARR_GU", gu);
            c->qa_head = qpick("MIS_QARR_HEAD", hd);
        }
        c->S_qkv = std::min(8, c->a_qkv.S); c->a_qkv.S = c->S_qkv;
        c->S_o = std::min(8, c->a_o.S); c->a_o.S = c->S_o;
        c->S_down = std::min(8, c->a_down.S); c->a_down.S = c->S_down;
    }
    size_t kv = (size_t)c->L * batch * c->Hkv * Smax * c->D;
    c->kcache.alloc(kv);
    c->vtcache.alloc(kv);
    HIP_CHECK(hipMemsetAsync(c->kcache.p, 0, kv * 2, s));
    HIP_CHECK(hipMemsetAsync(c->vtcache.p, 0, kv * 2, s));
    if (new_tables || !c->rope_cos.p) build_rope_tables(c);
    c->ids.alloc(Mpad); c->pos_cur.alloc(Mpad); c->pos_next.alloc(Mpad); c->active.alloc(Mpad);
    c->ids.zero(s); c->pos_cur.zero(s); c->pos_next.zero(s); c->active.zero(s);
    c->h.alloc((size_t)Mpad * d); c->x.alloc((size_t)Mpad * d);
    c->attn_out.alloc((size_t)Mpad * HD); c->act.alloc((size_t)Mpad * c->ff);
    c->logits.alloc((size_t)Mpad * c->Vpad);
    c->e_buf.alloc((size_t)Mpad * c->Vpad);
    c->qkv_part.alloc((size_t)c->S_qkv * Mpad * c->Nqkv);
    c->part.alloc((size_t)std::max(c->S_o, c->S_down) * Mpad * d);
    c->h.zero(s); c->x.zero(s); c->attn_out.zero(s); c->act.zero(s); c->logits.zero(s);
    HIP_CHECK(hipStreamSynchronize(s));
}

// the step chain's five GEMMs: dense bf16 tiles or, for a quantised role, codes + scales (lm_qgemm.hip)
static void gemm_qkv(mis_tts* c, size_t li, hipStream_t s) {
    if (c->q_qkv.on) launch_gemm_skinny_q(c->q_qkv.bits, EPI_PARTIAL, c->r_part, c->ksb_part, c->q_qkv.q.p + c->q_qkv.q_layer * li, c->q_qkv.sb.p + c->q_qkv.sb_layer * li,
                                          c->x.p, c->qkv_part.p, c->Nqkv / 16, c->d / 64, c->S_qkv, c->Nqkv, c->Mpad, s);
    else launch_gemm_skinny(EPI_PARTIAL, c->a_qkv.R, c->a_qkv.ksb, c->wqkv.p + layer_qkv_elems(c) * li, c->x.p, c->qkv_part.p, c->Nqkv / 16, c->d / 32, c->S_qkv,
                            c->Nqkv, c->Mpad, s, nullptr, c->a_qkv.U);
}
static void gemm_o(mis_tts* c, size_t li, hipStream_t s) {
    const int HD = c->H * c->D;
    if (c->q_o.on) launch_gemm_skinny_q(c->q_o.bits, EPI_PARTIAL, c->r_part, c->ksb_part, c->q_o.q.p + c->q_o.q_layer * li, c->q_o.sb.p + c->q_o.sb_layer * li,
                                        c->attn_out.p, c->part.p, c->d / 16, HD / 64, c->S_o, c->d, c->Mpad, s);
    else launch_gemm_skinny(EPI_PARTIAL, c->a_o.R, c->a_o.ksb, c->wo.p + layer_o_elems(c) * li, c->attn_out.p, c->part.p, c->d / 16, HD / 32, c->S_o, c->d,
                            c->Mpad, s, nullptr, c->a_o.U);
}
static void gemm_gate_up(mis_tts* c, size_t li, hipStream_t s) {
    if (c->q_gu.on) launch_gemm_skinny_q(c->q_gu.bits, EPI_SILU_MUL, c->qa_gu.R, c->qa_gu.ksb, c->q_gu.q.p + c->q_gu.q_layer * li, c->q_gu.sb.p + c->q_gu.sb_layer * li, c->x.p,
                                         c->act.p, 2 * c->ff / 16, c->d / 64, 1, c->ff, c->Mpad, s);
    else launch_gemm_skinny(EPI_SILU_MUL, c->r_gu, c->ksb_gu, c->wgu.p + layer_gu_elems(c) * li, c->x.p, c->act.p, 2 * c->ff / 16, c->d / 32, 1, c->ff, c->Mpad, s);
}
static void gemm_down(mis_tts* c, size_t li, hipStream_t s) {
    if (c->q_down.on) launch_gemm_skinny_q(c->q_down.bits, EPI_PARTIAL, c->r_part, c->ksb_part, c->q_down.q.p + c->q_down.q_layer * li,
                                           c->q_down.sb.p + c->q_down.sb_layer * li, c->act.p, c->part.p, c->d / 16, c->ff / 64, c->S_down, c->d, c->Mpad, s);
    else launch_gemm_skinny(EPI_PARTIAL, c->a_down.R, c->a_down.ksb, c->wdown.p + layer_down_elems(c) * li, c->act.p, c->part.p, c->d / 16, c->ff / 32, c->S_down,
                            c->d, c->Mpad, s, nullptr, c->a_down.U);
}

// embed -> L x block.  Leaves x = final-norm(h) ready for lm_head.   (LlamaTTS.swift:335-345,303-310)
// table/rows/ids: embedding source override (composite engines feed input embeddings as a [rows][d] table)
static void enqueue_layers(mis_tts* c, const bf16_t* table = nullptr, int table_rows = 0, const int32_t* ids = nullptr) {
    hipStream_t s = c->stream;
    const int d = c->d, Mpad = c->Mpad;
    const float eps = c->cfg.rms_norm_eps;
    launch_embed_rmsnorm(table ? table : c->emb.p, ids ? ids : c->ids.p, c->active.p, c->pos_cur.p, c->pos_next.p, c->norms.p,
                         c->h.p, c->x.p, d, table ? table_rows : c->V, eps, c->batch, Mpad, s);
    for (int li = 0; li < c->L; ++li) {
        const size_t lkv = (size_t)c->batch * c->Hkv * c->Smax * c->D;
        gemm_qkv(c, li, s);
        AttnParams ap{};
        ap.qkv_part = c->qkv_part.p; ap.S = c->S_qkv; ap.Mpad = Mpad; ap.Nqkv = c->Nqkv;
        ap.kcache = c->kcache.p + lkv * li; ap.vtcache = c->vtcache.p + lkv * li;
        ap.pos = c->pos_cur.p; ap.active = c->active.p;
        ap.rope_cos = c->rope_cos.p; ap.rope_sin = c->rope_sin.p;
        ap.out = c->attn_out.p; ap.H = c->H; ap.Hkv = c->Hkv; ap.D = c->D; ap.Smax = c->Smax;
        ap.scale = 1.0f / sqrtf((float)c->D);
        if (c->cfg.qk_norm) {
            ap.qnorm_w = c->qknorm.p + (size_t)(2 * li) * c->D;
            ap.knorm_w = c->qknorm.p + (size_t)(2 * li + 1) * c->D;
            ap.qk_eps = c->cfg.rms_norm_eps;
        }
        ap.rope_in_dtype = c->cfg.rope_ops_in_dtype;
        launch_attn_decode(ap, c->batch, s);
        gemm_o(c, li, s);
        launch_reduce_residual_rmsnorm(c->part.p, c->S_o, Mpad, d, c->h.p, c->norms.p + (size_t)(2 * li + 1) * d, c->x.p, eps, s);
        gemm_gate_up(c, li, s);
        gemm_down(c, li, s);
        const bf16_t* next_norm = c->norms.p + (size_t)(li + 1 < c->L ? 2 * (li + 1) : 2 * c->L) * d;
        launch_reduce_residual_rmsnorm(c->part.p, c->S_down, Mpad, d, c->h.p, next_norm, c->x.p, eps, s);
    }
}
static void enqueue_lm_head(mis_tts* c, const bf16_t* head = nullptr) {
    if (!head && c->q_head.on) {
        launch_gemm_skinny_q(c->q_head.bits, EPI_BF16, c->qa_head.R, c->qa_head.ksb, c->q_head.q.p, c->q_head.sb.p, c->x.p, c->logits.p, c->Vpad / 16, c->d / 64, 1, c->Vpad,
                             c->Mpad, c->stream);
        return;
    }
    launch_gemm_skinny(EPI_BF16, c->a_head.R, c->a_head.ksb, head ? head : c->lm_head.p, c->x.p, c->logits.p, c->Vpad / 16, c->d / 32, 1,
                       c->Vpad, c->Mpad, c->stream, nullptr, c->a_head.U);
}

// ---- the prompt in one pass per chunk of positions (LlamaTTS.swift:711; kernels and layout: lm_prefill.hip).  On return the K/V
// caches hold every prompt position, c->x is the packed final-norm hidden state of every row's LAST prompt token (the operand of the
// first lm_head) and the position counters stand behind the prompts - the state the position-by-position prefill leaves.
static bool prefill_batched_ok(const mis_tts* c, int Lmax) {
    const bool off = getenv("MIS_PREFILL_SEQ") && atoi(getenv("MIS_PREFILL_SEQ")) != 0;             // A/B and parity tests (read per call)
    const int HD = c->H * c->D;
    // dense weights: k_gemm_pf; all four roles streamed as codes: the decode step's k_gemm_skinny_q on 64-row chunks; a mix: position by position
    const bool dense = !c->q_qkv.on && !c->q_o.on && !c->q_gu.on && !c->q_down.on;
    const bool coded = c->q_qkv.on && c->q_o.on && c->q_gu.on && c->q_down.on;
    return !off && Lmax >= 2 && (dense || coded) && c->d % 64 == 0 && HD % 64 == 0 && c->ff % 64 == 0 && c->d >= 128 && HD >= 128 && c->ff >= 128;
}
// ---- short prompts (positions x rows <= 64, dense weights): the DECODE STEP's kernels with (position, sequence) pairs as their rows.
// k_gemm_pf works on 128 x 128 tiles: at 23 positions of one row its o_proj / down_proj grids are 4 blocks (21 us per launch on 256 CUs,
// profiles/r05_final_soprano_engine_kernel_stats.csv: 1.4 ms to prefill 23 positions of Soprano-80M - 7 % of a batch-1 generate).  The
// weight-streaming GEMMs of the decode step take up to 64 rows, their glue does residual + RMSNorm and writes the packed operand, and the
// attention kernel already has the pair arrangement (append-only launch, then every pair as a row; first schedule, like the chunked pass).
// Same rounding points as both other prefill forms; float32 summation orders are the decode step's.  MIS_PREFILL_SMALL=0 keeps k_gemm_pf.
__global__ void k_pf_repack_rows(const bf16_t* __restrict__ src, int MT_src, int row0, bf16_t* __restrict__ dst, int MT_dst, int d) {
    const int m = blockIdx.x;
    for (int k = threadIdx.x; k < d; k += blockDim.x) dst[xpk_index(m, k, MT_dst)] = src[xpk_index(row0 + m, k, MT_src)];
}
// WHO takes it: only the batch-1 token engine's prompt (tts_internal_prefill_kv).  The generic prefill keeps ONE arithmetic whatever the
// batch - a row computed in a 32-row batch equals the same row computed alone or in a shard (tests/test_gpu_generate.py, the group
// tests) - and this form sums its split-K slabs in another order than k_gemm_pf.  MIS_PREFILL_SMALL=1 forces it for every short prompt
// (parity tests), =0 turns it off.
static bool prefill_small_ok(const mis_tts* c, int Lmax, bool engine_prompt) {
    const char* e = getenv("MIS_PREFILL_SMALL");
    const bool dense = !c->q_qkv.on && !c->q_o.on && !c->q_gu.on && !c->q_down.on;
    const bool want = e ? atoi(e) != 0 : engine_prompt;
    return want && dense && Lmax >= 2 && Lmax * c->batch <= 64;
}
static void prefill_small(mis_tts* c, const int32_t* prompt_mat_dev, const int32_t* lens_dev, const std::vector<int32_t>& lens, int Lmax,
                          const bf16_t* embed_rows) {
    hipStream_t s = c->stream;
    const int d = c->d, HD = c->H * c->D, Mpad = c->Mpad, batch = c->batch;
    const float eps = c->cfg.rms_norm_eps;
    const int Mr = Lmax * batch, Mp = (int)round_up((size_t)Mr, 16);
    const size_t Mc = (size_t)Mp + Mpad;                                      // (the last position is repacked Mpad rows wide)
    const int S_qkv = std::min(8, gemm_choose_split(c->Nqkv / 16 / 2, d / 32, 4, 8)), S_o = gemm_choose_split(d / 16 / 2, HD / 32, 4, 8),
              S_down = gemm_choose_split(d / 16 / 2, c->ff / 32, 4, 8);
    c->pf_h.alloc(Mc * d); c->pf_x.alloc(Mc * d); c->pf_xpk.alloc(Mc * d); c->pf_attn.alloc(Mc * HD); c->pf_actpk.alloc(Mc * c->ff);
    c->pf_qkv.alloc((size_t)S_qkv * Mp * c->Nqkv); c->pf_tmp.alloc((size_t)std::max(S_o, S_down) * Mp * d);
    c->pf_pos.alloc(Mc); c->pf_on.alloc(Mc);
    HIP_CHECK(hipMemsetAsync(c->pf_attn.p, 0, Mc * HD * 2, s));              // rows of padded positions are never written by the attention
    HIP_CHECK(hipMemsetAsync(c->pf_x.p, 0, Mc * d * 2, s));
    HIP_CHECK(hipMemsetAsync(c->pf_h.p, 0, Mc * d * 2, s));
    HIP_CHECK(hipMemsetAsync(c->pf_xpk.p, 0, Mc * d * 2, s));
    HIP_CHECK(hipMemsetAsync(c->pf_pos.p, 0, Mc * 4, s));
    HIP_CHECK(hipMemsetAsync(c->pf_on.p, 0, Mc, s));
    if (embed_rows) launch_pf_rows_rmsnorm(embed_rows, Mpad, lens_dev, Lmax, 0, Lmax, batch, batch, c->norms.p, c->pf_h.p, c->pf_x.p, c->pf_pos.p, c->pf_on.p, d, eps, s);
    else launch_pf_embed_rmsnorm(c->emb.p, prompt_mat_dev, lens_dev, Lmax, 0, Lmax, batch, batch, c->V, c->norms.p, c->pf_h.p, c->pf_x.p, c->pf_pos.p,
                                 c->pf_on.p, d, eps, s);
    launch_pf_pack_rows(c->pf_x.p, c->pf_xpk.p, Mp, d, s);
    const size_t lkv = (size_t)batch * c->Hkv * c->Smax * c->D;
    for (int li = 0; li < c->L; ++li) {
        launch_gemm_skinny(EPI_PARTIAL, 2, 4, c->wqkv.p + layer_qkv_elems(c) * li, c->pf_xpk.p, c->pf_qkv.p, c->Nqkv / 16, d / 32, S_qkv, c->Nqkv, Mp, s);
        AttnParams ap{};
        ap.qkv_part = c->pf_qkv.p; ap.S = S_qkv; ap.Mpad = Mp; ap.Nqkv = c->Nqkv;
        ap.kcache = c->kcache.p + lkv * li; ap.vtcache = c->vtcache.p + lkv * li;
        ap.pos = c->pf_pos.p; ap.active = c->pf_on.p;
        ap.rope_cos = c->rope_cos.p; ap.rope_sin = c->rope_sin.p;
        ap.out = c->pf_attn.p; ap.out_ld = 0;                               // packed: the o_proj launch below is the decode step's
        ap.H = c->H; ap.Hkv = c->Hkv; ap.D = c->D; ap.Smax = c->Smax; ap.scale = 1.0f / sqrtf((float)c->D);
        if (c->cfg.qk_norm) {
            ap.qnorm_w = c->qknorm.p + (size_t)(2 * li) * c->D;
            ap.knorm_w = c->qknorm.p + (size_t)(2 * li + 1) * c->D;
            ap.qk_eps = c->cfg.rms_norm_eps;
        }
        ap.rope_in_dtype = c->cfg.rope_ops_in_dtype;
        ap.first_schedule = 1;
        ap.cache_rows = batch;
        ap.append_only = 1;
        launch_attn_decode(ap, Mr, s);
        ap.append_only = 0;
        launch_attn_decode(ap, Mr, s);
        launch_gemm_skinny(EPI_PARTIAL, 2, 4, c->wo.p + layer_o_elems(c) * li, c->pf_attn.p, c->pf_tmp.p, d / 16, HD / 32, S_o, d, Mp, s);
        launch_reduce_residual_rmsnorm(c->pf_tmp.p, S_o, Mp, d, c->pf_h.p, c->norms.p + (size_t)(2 * li + 1) * d, c->pf_xpk.p, eps, s);
        launch_gemm_skinny(EPI_SILU_MUL, 2, 4, c->wgu.p + layer_gu_elems(c) * li, c->pf_xpk.p, c->pf_actpk.p, 2 * c->ff / 16, d / 32, 1, c->ff, Mp, s);
        launch_gemm_skinny(EPI_PARTIAL, 2, 4, c->wdown.p + layer_down_elems(c) * li, c->pf_actpk.p, c->pf_tmp.p, d / 16, c->ff / 32, S_down, d, Mp, s);
        const bf16_t* next_norm = c->norms.p + (size_t)(li + 1 < c->L ? 2 * (li + 1) : 2 * c->L) * d;
        launch_reduce_residual_rmsnorm(c->pf_tmp.p, S_down, Mp, d, c->pf_h.p, next_norm, c->pf_xpk.p, eps, s);
    }
    // left-padded prompts: every row's last token is position Lmax - 1; rows batch .. Mpad - 1 of the packed x belong to no sequence (zeros)
    c->x.zero(s);
    hipLaunchKernelGGL(k_pf_repack_rows, dim3(batch), dim3(256), 0, s, c->pf_xpk.p, Mp / 16, (Lmax - 1) * batch, c->x.p, Mpad / 16, d);
    std::vector<int32_t> pn(Mpad, 0), pc(Mpad, 0);
    for (int b = 0; b < batch; ++b) { pn[b] = lens[b]; pc[b] = lens[b] - 1; }
    HIP_CHECK(hipMemcpyAsync(c->pos_next.p, pn.data(), (size_t)Mpad * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->pos_cur.p, pc.data(), (size_t)Mpad * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(s));                                      // pn / pc are host stack memory
}

// embed_rows != null: the [Lmax][Mpad][d] input embeddings of a composite engine replace the gather from the model's own table
static void prefill_batched(mis_tts* c, const int32_t* prompt_mat_dev, const int32_t* lens_dev, const std::vector<int32_t>& lens, int Lmax,
                            const bf16_t* embed_rows = nullptr, bool engine_prompt = false) {
    if (prefill_small_ok(c, Lmax, engine_prompt)) { prefill_small(c, prompt_mat_dev, lens_dev, lens, Lmax, embed_rows); return; }
    hipStream_t s = c->stream;
    const int d = c->d, HD = c->H * c->D, Mpad = c->Mpad, batch = c->batch;
    const float eps = c->cfg.rms_norm_eps;
    // Causal attention of a chunk, two arrangements: one launch per position (rows = sequences) - fills the chip on its own once
    // batch x kv heads reaches about half the CUs - or every (position, sequence) pair as a row of ONE launch: an append-only launch
    // puts the roped keys and the values of all the chunk's positions into the caches, then the attention launch reads them like any
    // earlier key (its own key is patched in registers as in the decode step).  Measured: batch 32 x 8 heads, 32 positions: 14.3 ms
    // per position against 15.8 ms as pairs; batch 1, 350 positions: 90 -> 32 ms the other way.  MIS_PF_ATTN_LOOP=1 / 0 forces one.
    const char* loop_env = getenv("MIS_PF_ATTN_LOOP");
    const bool attn_loop = loop_env ? atoi(loop_env) != 0 : batch * c->Hkv >= 128;
    // rows of the chunk: position-major, `rs` rows per position.  The pair arrangement needs no padding between positions (a batch of 1
    // would otherwise carry 15 padding rows per position through every GEMM); the per-position one keeps Mpad (its flag words are aligned)
    const int rs = attn_loop ? Mpad : batch;
    const int Tc = std::min(Lmax, std::max(1, 4096 / rs));                   // positions per chunk: <= 4096 rows of activations
    const size_t Mc = round_up((size_t)Tc * rs, 16) + Mpad;                  // GEMM rows round up to 16; the last position is read Mpad rows wide
    c->pf_h.alloc(Mc * d); c->pf_x.alloc(Mc * d); c->pf_attn.alloc(Mc * HD);
    c->pf_qkv.alloc(Mc * c->Nqkv); c->pf_pos.alloc(Mc); c->pf_on.alloc(Mc);
    HIP_CHECK(hipMemsetAsync(c->pf_attn.p, 0, Mc * HD * 2, s));             // rows of padded positions are never written by the attention
    HIP_CHECK(hipMemsetAsync(c->pf_x.p, 0, Mc * d * 2, s));                 // rows past the chunk (rounding, slack) hold zeros, not stale bits
    HIP_CHECK(hipMemsetAsync(c->pf_h.p, 0, Mc * d * 2, s));
    // quantised checkpoints (every role streamed as codes): each GEMM of the chunk runs the decode step's k_gemm_skinny_q on 64-row
    // slices - rows packed into MFMA fragments first, float32 sums out, the residual add as its own small kernel; gate|up writes the
    // packed activation its down projection reads.  The weights are streamed once per 64 rows instead of once per position.
    const bool coded = c->q_qkv.on;
    if (coded) { c->pf_xpk.alloc((size_t)64 * std::max(d, HD)); c->pf_actpk.alloc((size_t)64 * c->ff); c->pf_tmp.alloc((size_t)64 * d); }
    else c->pf_act.alloc(Mc * c->ff);
    auto q_rows = [&](int M, const std::function<void(int r0, int mp)>& body) {
        for (int r0 = 0; r0 < M; r0 += 64) body(r0, std::min(64, M - r0));
    };
    auto q_partial = [&](const mis_tts::QRole& q, int li, const bf16_t* xpk, float* out, int N, int K, int mp) {
        launch_gemm_skinny_q(q.bits, EPI_PARTIAL, c->r_part, c->ksb_part, q.q.p + q.q_layer * li, q.sb.p + q.sb_layer * li, xpk, out, N / 16, K / 64, 1, N, mp, s);
    };
    const size_t lkv = (size_t)batch * c->Hkv * c->Smax * c->D;
    int t0 = 0;
    for (; t0 < Lmax; t0 += Tc) {
        const int tc = std::min(Tc, Lmax - t0);
        const int Mr = tc * rs;                                              // rows that exist
        const int M = (int)round_up((size_t)Mr, 16);                         // rows the GEMMs run over
        if (t0 > 0 && (size_t)Mr < Mc) {                                     // a shorter last chunk: the rows behind it held the previous chunk's
            HIP_CHECK(hipMemsetAsync(c->pf_x.p + (size_t)Mr * d, 0, (Mc - Mr) * d * 2, s));       // values - zeros again, as the comments below say
            HIP_CHECK(hipMemsetAsync(c->pf_h.p + (size_t)Mr * d, 0, (Mc - Mr) * d * 2, s));
            HIP_CHECK(hipMemsetAsync(c->pf_attn.p + (size_t)Mr * HD, 0, (Mc - Mr) * HD * 2, s));
        }
        if (embed_rows) launch_pf_rows_rmsnorm(embed_rows, Mpad, lens_dev, Lmax, t0, tc, batch, rs, c->norms.p, c->pf_h.p, c->pf_x.p, c->pf_pos.p, c->pf_on.p, d, eps, s);
        else launch_pf_embed_rmsnorm(c->emb.p, prompt_mat_dev, lens_dev, Lmax, t0, tc, batch, rs, c->V, c->norms.p, c->pf_h.p, c->pf_x.p,
                                     c->pf_pos.p, c->pf_on.p, d, eps, s);
        for (int li = 0; li < c->L; ++li) {
            if (coded) q_rows(M, [&](int r0, int mp) {
                launch_pf_pack_rows(c->pf_x.p + (size_t)r0 * d, c->pf_xpk.p, mp, d, s);
                q_partial(c->q_qkv, li, c->pf_xpk.p, c->pf_qkv.p + (size_t)r0 * c->Nqkv, c->Nqkv, d, mp);
            });
            else launch_gemm_pf(PF_F32, c->pf_x.p, c->wqkv.p + layer_qkv_elems(c) * li, c->pf_qkv.p, M, c->Nqkv, d, s);
            auto attn_params = [&](int tl0) {
                AttnParams ap{};
                ap.qkv_part = c->pf_qkv.p + (size_t)tl0 * rs * c->Nqkv; ap.S = 1; ap.Mpad = Mpad; ap.Nqkv = c->Nqkv;
                ap.kcache = c->kcache.p + lkv * li; ap.vtcache = c->vtcache.p + lkv * li;
                ap.pos = c->pf_pos.p + (size_t)tl0 * rs; ap.active = c->pf_on.p + (size_t)tl0 * rs;
                ap.rope_cos = c->rope_cos.p; ap.rope_sin = c->rope_sin.p;
                ap.out = c->pf_attn.p + (size_t)tl0 * rs * HD; ap.out_ld = HD;
                ap.H = c->H; ap.Hkv = c->Hkv; ap.D = c->D; ap.Smax = c->Smax; ap.scale = 1.0f / sqrtf((float)c->D);
                if (c->cfg.qk_norm) {
                    ap.qnorm_w = c->qknorm.p + (size_t)(2 * li) * c->D;
                    ap.knorm_w = c->qknorm.p + (size_t)(2 * li + 1) * c->D;
                    ap.qk_eps = c->cfg.rms_norm_eps;
                }
                ap.rope_in_dtype = c->cfg.rope_ops_in_dtype;
                ap.first_schedule = 1;          // one kernel for both arrangements: a row's logits do not depend on how many rows share the call
                return ap;
            };
            if (attn_loop) {
                for (int tl = 0; tl < tc; ++tl) launch_attn_decode(attn_params(tl), batch, s);
            } else {
                AttnParams ap = attn_params(0);
                ap.cache_rows = rs;
                ap.append_only = 1;
                launch_attn_decode(ap, Mr, s);
                ap.append_only = 0;
                launch_attn_decode(ap, Mr, s);
            }
            if (coded) q_rows(M, [&](int r0, int mp) {
                launch_pf_pack_rows(c->pf_attn.p + (size_t)r0 * HD, c->pf_xpk.p, mp, HD, s);
                q_partial(c->q_o, li, c->pf_xpk.p, c->pf_tmp.p, d, HD, mp);
                launch_pf_add_resid(c->pf_h.p + (size_t)r0 * d, c->pf_tmp.p, (size_t)mp * d, s);
            });
            else launch_gemm_pf(PF_RESID, c->pf_attn.p, c->wo.p + layer_o_elems(c) * li, c->pf_h.p, M, d, HD, s);
            launch_pf_rmsnorm(c->pf_h.p, c->norms.p + (size_t)(2 * li + 1) * d, c->pf_x.p, M, d, eps, s);
            if (coded) q_rows(M, [&](int r0, int mp) {
                launch_pf_pack_rows(c->pf_x.p + (size_t)r0 * d, c->pf_xpk.p, mp, d, s);
                launch_gemm_skinny_q(c->q_gu.bits, EPI_SILU_MUL, 2, c->ksb_gu, c->q_gu.q.p + c->q_gu.q_layer * li, c->q_gu.sb.p + c->q_gu.sb_layer * li,
                                     c->pf_xpk.p, c->pf_actpk.p, 2 * c->ff / 16, d / 64, 1, c->ff, mp, s);
                q_partial(c->q_down, li, c->pf_actpk.p, c->pf_tmp.p, d, c->ff, mp);
                launch_pf_add_resid(c->pf_h.p + (size_t)r0 * d, c->pf_tmp.p, (size_t)mp * d, s);
            });
            else {
                launch_gemm_pf(PF_SILU, c->pf_x.p, c->wgu.p + layer_gu_elems(c) * li, c->pf_act.p, M, 2 * c->ff, d, s);
                launch_gemm_pf(PF_RESID, c->pf_act.p, c->wdown.p + layer_down_elems(c) * li, c->pf_h.p, M, d, c->ff, s);
            }
            const bf16_t* next_norm = c->norms.p + (size_t)(li + 1 < c->L ? 2 * (li + 1) : 2 * c->L) * d;
            launch_pf_rmsnorm(c->pf_h.p, next_norm, c->pf_x.p, M, d, eps, s);
        }
        if (t0 + tc >= Lmax) {                                               // left-padded prompts: every row's last token is position Lmax - 1
            // rows batch .. Mpad-1 of the packed x belong to no sequence: what follows the last position in pf_x (zeros, or the rounding rows)
            launch_pf_pack_rows(c->pf_x.p + (size_t)(tc - 1) * rs * d, c->x.p, Mpad, d, s);
        }
    }
    std::vector<int32_t> pn(Mpad, 0), pc(Mpad, 0);
    for (int b = 0; b < batch; ++b) { pn[b] = lens[b]; pc[b] = lens[b] - 1; }
    HIP_CHECK(hipMemcpyAsync(c->pos_next.p, pn.data(), (size_t)Mpad * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->pos_cur.p, pc.data(), (size_t)Mpad * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(s));                                      // pn / pc are host stack memory
}

extern "C" mis_status mis_lm_reset(mis_tts* c, int batch, int max_context) {
    MIS_API_BEGIN
    MIS_REQUIRE(c, MIS_ERR_INVALID_INPUT, "null handle");
    lm_reset(c, batch, max_context);
    MIS_API_END
}

__global__ void k_bf16_rows_to_f32(const bf16_t* __restrict__ src, int src_stride, float* __restrict__ dst, int cols,
                                   int rows) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * cols) return;
    int r = (int)(i / cols), cidx = (int)(i - (size_t)r * cols);
    dst[i] = bf16_to_f32(src[(size_t)r * src_stride + cidx]);
}
__global__ void k_f32_rows_to_bf16(const float* __restrict__ src, int cols, bf16_t* __restrict__ dst, int dst_stride,
                                   int rows) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * cols) return;
    int r = (int)(i / cols), cidx = (int)(i - (size_t)r * cols);
    dst[(size_t)r * dst_stride + cidx] = f32_to_bf16(src[i]);
}

__global__ void k_unpack_x_f32(const bf16_t* __restrict__ x, float* __restrict__ out, int d, int batch, int MT) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)batch * d) return;
    int m = (int)(i / d), k = (int)(i - (size_t)m * d);
    size_t off = ((((size_t)(k >> 5) * MT + (m >> 4)) * 64) + (((k & 31) >> 3) << 4) + (m & 15)) * 8 + (k & 7);
    out[i] = bf16_to_f32(x[off]);
}

extern "C" mis_status mis_lm_forward_hidden(mis_tts* c, const int32_t* ids, const uint8_t* active, float* logits_out,
                                            float* hidden_out);
extern "C" mis_status mis_lm_forward(mis_tts* c, const int32_t* ids, const uint8_t* active, float* logits_out) {
    return mis_lm_forward_hidden(c, ids, active, logits_out, nullptr);
}

extern "C" mis_status mis_lm_forward_hidden(mis_tts* c, const int32_t* ids, const uint8_t* active, float* logits_out,
                                            float* hidden_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && ids, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(c->batch > 0, MIS_ERR_NOT_INITIALIZED, "call mis_lm_reset first");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    std::vector<uint8_t> act(c->batch, 1);
    if (active) {
        std::vector<uint8_t> tmp(c->batch);
        HIP_CHECK(hipMemcpy(tmp.data(), active, c->batch, hipMemcpyDefault));
        act = tmp;
    }
    {   // context overflow check on the host mirror of pos_next
        std::vector<int32_t> pn(c->batch);
        HIP_CHECK(hipMemcpy(pn.data(), c->pos_next.p, c->batch * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < c->batch; ++b)
            MIS_REQUIRE(!act[b] || pn[b] < c->Smax, MIS_ERR_INVALID_INPUT, "row %d exceeds max_context %d", b, c->Smax);
    }
    HIP_CHECK(hipMemcpyAsync(c->ids.p, ids, c->batch * 4, hipMemcpyDefault, s));
    HIP_CHECK(hipMemcpyAsync(c->active.p, act.data(), c->batch, hipMemcpyHostToDevice, s));
    enqueue_layers(c);
    if (hidden_out) {
        size_t n = (size_t)c->batch * c->d;
        c->logits_f32.alloc(std::max(n, c->logits_f32.n));
        hipLaunchKernelGGL(k_unpack_x_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->x.p, c->logits_f32.p, c->d,
                           c->batch, c->Mpad / 16);
        HIP_CHECK(hipMemcpyAsync(hidden_out, c->logits_f32.p, n * 4, hipMemcpyDefault, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    if (logits_out) {
        enqueue_lm_head(c);
        c->logits_f32.alloc((size_t)c->batch * c->V);
        size_t n = (size_t)c->batch * c->V;
        hipLaunchKernelGGL(k_bf16_rows_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->logits.p, c->Vpad,
                           c->logits_f32.p, c->V, c->batch);
        HIP_CHECK(hipMemcpyAsync(logits_out, c->logits_f32.p, n * 4, hipMemcpyDefault, s));
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(s));
    MIS_API_END
}

// The prefill call of the reference loop as an entry point (`model(inputIds, cache:)`, LlamaTTS.swift:711): prompts (ragged, flat
// + lens) through the model with empty caches; logits_out f32 [batch, vocab] (nullable) = logits of the NEXT token of every row.
// The caches then hold the prompts and mis_lm_forward continues behind them.  Runs the batched path when the model allows it
// (dense weights, dimensions multiples of 64), else position by position; MIS_PREFILL_SEQ=1 forces the latter.
extern "C" mis_status mis_lm_prefill(mis_tts* c, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch, int max_context,
                                     float* logits_out, float* hidden_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && prompt_ids && prompt_lens && batch >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "model not finalized");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    std::vector<int32_t> lens(batch);
    HIP_CHECK(hipMemcpy(lens.data(), prompt_lens, batch * 4, hipMemcpyDefault));
    int Lmax = 0;
    size_t total = 0;
    for (int b = 0; b < batch; ++b) {
        MIS_REQUIRE(lens[b] >= 1, MIS_ERR_INVALID_INPUT, "empty prompt in row %d", b);
        Lmax = std::max(Lmax, lens[b]); total += lens[b];
    }
    std::vector<int32_t> flat(total);
    HIP_CHECK(hipMemcpy(flat.data(), prompt_ids, total * 4, hipMemcpyDefault));
    for (auto t : flat) MIS_REQUIRE(t >= 0 && t < c->V, MIS_ERR_INVALID_INPUT, "prompt token %d outside the vocabulary", t);
    lm_reset(c, batch, std::max(max_context, Lmax + 1));
    std::vector<int32_t> pm((size_t)batch * Lmax, 0);
    {
        size_t off = 0;
        for (int b = 0; b < batch; ++b) {
            for (int j = 0; j < lens[b]; ++j) pm[(size_t)b * Lmax + (Lmax - lens[b]) + j] = flat[off + j];
            off += lens[b];
        }
    }
    c->prompt_mat.alloc(pm.size()); c->prompt_lens.alloc(batch); c->step_counter.alloc(1);
    HIP_CHECK(hipMemcpyAsync(c->prompt_mat.p, pm.data(), pm.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->prompt_lens.p, lens.data(), batch * 4, hipMemcpyHostToDevice, s));
    c->step_counter.zero(s);
    HIP_CHECK(hipStreamSynchronize(s));
    if (prefill_batched_ok(c, Lmax)) prefill_batched(c, c->prompt_mat.p, c->prompt_lens.p, lens, Lmax);
    else
        for (int j = 0; j < Lmax; ++j) {
            launch_prefill_feed(c->prompt_mat.p, c->prompt_lens.p, Lmax, c->step_counter.p, c->ids.p, c->active.p, batch, s);
            enqueue_layers(c);
        }
    {
        std::vector<uint8_t> ones(batch, 1);
        HIP_CHECK(hipMemcpyAsync(c->active.p, ones.data(), batch, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    if (hidden_out) {                                                        // model.norm(h) of every row's last prompt token
        const size_t n = (size_t)batch * c->d;
        c->logits_f32.alloc(std::max(n, c->logits_f32.n));
        hipLaunchKernelGGL(k_unpack_x_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->x.p, c->logits_f32.p, c->d, batch, c->Mpad / 16);
        HIP_CHECK(hipMemcpyAsync(hidden_out, c->logits_f32.p, n * 4, hipMemcpyDefault, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    if (logits_out) {
        enqueue_lm_head(c);
        const size_t n = (size_t)batch * c->V;
        c->logits_f32.alloc(n);
        hipLaunchKernelGGL(k_bf16_rows_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->logits.p, c->Vpad, c->logits_f32.p, c->V, batch);
        HIP_CHECK(hipMemcpyAsync(logits_out, c->logits_f32.p, n * 4, hipMemcpyDefault, s));
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(s));
    MIS_API_END
}

extern "C" mis_status mis_sample_logits(int device, const float* logits, int batch, int vocab, const int32_t* window,
                                        const int32_t* window_len, int ctx, const mis_gen_params* params, int step,
                                        int lo, int hi, int32_t* tokens_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(logits && params && tokens_out && batch >= 1 && vocab >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(ctx == 0 || (window && window_len), MIS_ERR_INVALID_INPUT, "window pointers required when ctx > 0");
    HIP_CHECK(hipSetDevice(device));
    int Vpad = (int)round_up(vocab, 16);
    DevBuf<float> lf, eb;
    DevBuf<bf16_t> lb;
    DevBuf<int32_t> win, wl, steps, toks;
    lf.alloc((size_t)batch * vocab); eb.alloc((size_t)batch * Vpad); lb.alloc((size_t)batch * Vpad);
    win.alloc((size_t)batch * std::max(ctx, 1)); wl.alloc(batch); steps.alloc(batch); toks.alloc(batch);
    HIP_CHECK(hipMemcpy(lf.p, logits, (size_t)batch * vocab * 4, hipMemcpyDefault));
    HIP_CHECK(hipMemset(lb.p, 0, (size_t)batch * Vpad * 2));
    size_t n = (size_t)batch * vocab;
    hipLaunchKernelGGL(k_f32_rows_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, lf.p, vocab, lb.p, Vpad, batch);
    if (ctx > 0) {
        HIP_CHECK(hipMemcpy(win.p, window, (size_t)batch * ctx * 4, hipMemcpyDefault));
        HIP_CHECK(hipMemcpy(wl.p, window_len, batch * 4, hipMemcpyDefault));
    }
    std::vector<int32_t> st(batch, step);
    HIP_CHECK(hipMemcpy(steps.p, st.data(), batch * 4, hipMemcpyHostToDevice));
    DevBuf<SamplerScratch> scratch;
    scratch.alloc(batch);
    sampler_scratch_init(scratch.p, batch, 0);
    SamplerParams sp{};
    sp.scratch = scratch.p;
    sampler_plan(vocab, &sp.n_chunks, &sp.chunk_w);
    sp.logits = lb.p; sp.e_buf = eb.p; sp.Vpad = Vpad; sp.vocab = vocab;
    sp.window = ctx > 0 ? win.p : nullptr; sp.window_len = ctx > 0 ? wl.p : nullptr; sp.ctx = ctx;
    sp.tokens_out = toks.p; sp.tokens_stride = 0; sp.next_ids = toks.p; sp.step_override = steps.p;
    sp.temperature = params->temperature; sp.top_p = params->top_p; sp.penalty = params->repetition_penalty;
    sp.seed = params->seed; sp.row_offset = params->row_offset; sp.frame_constrained = params->frame_constrained;
    sp.lo = lo; sp.hi = hi; sp.eos_id = -1; sp.max_tokens = 1 << 30;
    sp.tokens_out = nullptr;
    DevBuf<float> l32;
    if (params->sampler_flavor == 1) {                  // Soprano: f32 per-occurrence penalty, no nucleus cut (see header)
        l32.alloc((size_t)batch * Vpad);
        sp.penalty_flavor = 1; sp.top_p = 1.0f; sp.logits32 = l32.p;
    }
    DevBuf<unsigned long long> dbg;
    const bool want_dbg = getenv("MIS_SAMP_DBG") != nullptr;        // diagnostics: phase stamps of block (0, 0) on stderr (tools/samp_phases.py)
    if (want_dbg) { dbg.alloc(48); dbg.zero(0); sp.dbg = dbg.p; }
    launch_sampler(sp, batch, 0);
    HIP_CHECK(hipGetLastError());
    if (want_dbg) {
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) launch_sampler(sp, batch, 0);         // (stamps of the last launch stay; timing of 20 back to back)
        HIP_CHECK(hipEventRecord(e1, 0));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[48];
        HIP_CHECK(hipMemcpy(st, dbg.p, sizeof(st), hipMemcpyDeviceToHost));
        fprintf(stderr, "SAMP_DBG us_per_launch %.2f stamps", ms * 1e3 / 20);
        for (int i = 0; i < 14; ++i) fprintf(stderr, " %llu", st[i] ? st[i] - st[0] : 0ull);         // 12, 13: inside barriers 1 / 2, after the store drain
        fprintf(stderr, " | row-0 blocks, 10 ns ticks since the earliest entry (entry, logits loaded, at barrier 1, past 1):");
        unsigned long long t0w = ~0ull;
        for (int cb = 0; cb < 8; ++cb) if (st[16 + 4 * cb] && st[16 + 4 * cb] < t0w) t0w = st[16 + 4 * cb];
        for (int cb = 0; cb < 8; ++cb) fprintf(stderr, " [%llu %llu %llu %llu]", st[16 + 4 * cb] - t0w, st[17 + 4 * cb] - t0w, st[18 + 4 * cb] - t0w,
                                               st[19 + 4 * cb] ? st[19 + 4 * cb] - t0w : 0ull);
        fprintf(stderr, "\n");
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    if (sampler_check_failed(scratch.p, batch, 0)) {
        // a row barrier of the one-launch sampler timed out (its 8 x batch blocks were not co-resident: another stream holds CUs).  This
        // entry point owns its inputs, so it falls back instead of failing: fresh logits (the failed launch may have applied penalties
        // in place), fresh scratch (sampler_check_failed re-initialised it), the multi-launch path
        HIP_CHECK(hipMemset(lb.p, 0, (size_t)batch * Vpad * 2));
        hipLaunchKernelGGL(k_f32_rows_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, lf.p, vocab, lb.p, Vpad, batch);
        if (ctx > 0) {                                      // (the failed launch's bookkeeping block slid the windows)
            HIP_CHECK(hipMemcpy(win.p, window, (size_t)batch * ctx * 4, hipMemcpyDefault));
            HIP_CHECK(hipMemcpy(wl.p, window_len, batch * 4, hipMemcpyDefault));
        }
        launch_sampler(sp, batch, 0, true);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipMemcpy(tokens_out, toks.p, batch * 4, hipMemcpyDefault));
    MIS_API_END
}

// ---------------------------------------------------------------------------- generate
struct GenOutputs {
    float* pcm_dev = nullptr;          // caller device buffer (or internal)
    int64_t pcm_stride = 0;
    std::vector<int64_t> pcm_lens;
    std::vector<int32_t> n_tokens;
    std::vector<int32_t> tokens;       // [batch][tokens_stride] host copy (if requested)
    int64_t tokens_stride = 0;
};

static double ms_between(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

// one block per row: hidden[b][slot][:] = unpack(x[b]); prefill: slot 0 (every step overwrites, the last one stays);
// decode: rows still active after the sampler append at hid_count[b]++
__global__ void k_collect_hidden(const bf16_t* __restrict__ x, const uint8_t* __restrict__ active, int32_t* __restrict__ hid_count,
                                 float* __restrict__ out, int64_t stride_rows, int d, int MT, int prefill) {
    const int b = blockIdx.x;
    int slot = 0;
    if (!prefill) {
        if (!active[b]) return;
        slot = hid_count[b];
        if (slot >= stride_rows) return;
    }
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        size_t off = ((((size_t)(k >> 5) * MT + (b >> 4)) * 64) + (((k & 31) >> 3) << 4) + (b & 15)) * 8 + (k & 7);
        out[((size_t)b * stride_rows + slot) * d + k] = bf16_to_f32(x[off]);
    }
    __syncthreads();
    if (!prefill && threadIdx.x == 0) hid_count[b] = slot + 1;
}

struct HiddenMode {            // Soprano: collect model.norm(h) per step instead of decoding SNAC frames
    bool on = false;
    int stop_id = -1;
    DevBuf<float>* hidden = nullptr;      // [batch][max_tokens + 1][d]
    std::vector<int32_t> n_hidden;
};

// A row barrier of the one-launch sampler timed out during the decode loop (its 8 x batch blocks were not co-resident: another stream
// holds compute units).  Thrown at the poll that sees the flag, before any token of that poll interval reaches the callback.
struct SamplerTimeout {};

// One attempt of generate.  `multi_launch_only`: the sampler's multi-launch kernels (no kernel of the step waits for another block).
// `emitted[b]`: tokens of row b already announced to the callback - by this attempt or by the one before it (the request is
// deterministic: prompts, seeds and the RNG counter are the same, so a second attempt reproduces them and continues behind them).
static void run_generate_attempt(mis_tts* c, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                 const mis_gen_params* gp, const float* const* snac_noise, float* pcm_dev, int64_t pcm_stride,
                                 bool want_tokens, mis_event_cb cb, void* user, const volatile int* cancel, GenOutputs& out,
                                 HiddenMode* hm, bool multi_launch_only, std::vector<int32_t>& emitted) {
    MIS_REQUIRE(c && c->finalized, MIS_ERR_NOT_INITIALIZED, "model not initialized");
    const bool hidden_mode = hm && hm->on;
    MIS_REQUIRE(hidden_mode || c->codec, MIS_ERR_NOT_INITIALIZED, "SNAC model not loaded");            // LlamaTTS.swift:672-674
    MIS_REQUIRE(prompt_ids && prompt_lens && gp, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(batch >= 1 && batch <= 64, MIS_ERR_INVALID_INPUT, "batch per GPU must be 1..64");
    const mis_snac_config* sc = hidden_mode ? nullptr : snac_config(c->codec);
    MIS_REQUIRE(hidden_mode || (sc->n_codebooks == 3 && sc->vq_strides[0] == 4 && sc->vq_strides[1] == 2 &&
                                sc->vq_strides[2] == 1 && sc->codebook_size == 4096),
                MIS_ERR_INVALID_INPUT, "Orpheus framing needs a 3-level 4/2/1 SNAC codec with 4096-entry codebooks");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int max_tokens = gp->max_tokens > 0 ? gp->max_tokens : 1200;
    std::vector<int32_t> lens(batch);
    HIP_CHECK(hipMemcpy(lens.data(), prompt_lens, batch * 4, hipMemcpyDefault));
    int Lmax = 0;
    size_t total = 0;
    for (int b = 0; b < batch; ++b) {
        MIS_REQUIRE(lens[b] >= 1, MIS_ERR_INVALID_INPUT, "empty prompt in row %d", b);
        Lmax = std::max(Lmax, lens[b]);
        total += lens[b];
    }
    std::vector<int32_t> flat(total);
    HIP_CHECK(hipMemcpy(flat.data(), prompt_ids, total * 4, hipMemcpyDefault));
    for (auto t : flat) MIS_REQUIRE(t >= 0 && t < c->V, MIS_ERR_INVALID_INPUT, "prompt token %d outside the vocabulary", t);
    const int ctx = std::max(gp->repetition_context, 0);
    const int all_stride = Lmax + max_tokens;
    lm_reset(c, batch, Lmax + max_tokens + 1);

    // host-built state: left-padded prompt matrix, all_ids (prompt, left-aligned), repetition windows
    std::vector<int32_t> pm((size_t)batch * Lmax, 0), all((size_t)batch * all_stride, 0), win((size_t)batch * std::max(ctx, 1), 0),
        wl(batch, 0), alen(batch);
    {
        size_t off = 0;
        for (int b = 0; b < batch; ++b) {
            for (int j = 0; j < lens[b]; ++j) {
                pm[(size_t)b * Lmax + (Lmax - lens[b]) + j] = flat[off + j];
                all[(size_t)b * all_stride + j] = flat[off + j];
            }
            int w = std::min(ctx, lens[b]);                 // processor.prompt(promptTokens), LlamaTTS.swift:695-696
            if (gp->sampler_flavor == 1) w = 0;             // Soprano penalises generated tokens only (Soprano.swift:833-848)
            for (int j = 0; j < w; ++j) win[(size_t)b * ctx + (ctx - w) + j] = flat[off + lens[b] - w + j];
            wl[b] = w;
            alen[b] = lens[b];
            off += lens[b];
        }
    }
    c->prompt_mat.alloc(pm.size()); c->prompt_lens.alloc(batch); c->step_counter.alloc(1);
    c->window.alloc(win.size()); c->window_len.alloc(batch); c->n_gen.alloc(batch);
    c->tokens_out.alloc((size_t)batch * max_tokens); c->all_ids.alloc(all.size()); c->all_len.alloc(batch);
    c->done_count.alloc(1);
    HIP_CHECK(hipMemcpyAsync(c->prompt_mat.p, pm.data(), pm.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->prompt_lens.p, lens.data(), batch * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->window.p, win.data(), win.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->window_len.p, wl.data(), batch * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->all_ids.p, all.data(), all.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->all_len.p, alen.data(), batch * 4, hipMemcpyHostToDevice, s));
    c->step_counter.zero(s); c->n_gen.zero(s); c->done_count.zero(s); c->tokens_out.zero(s);
    HIP_CHECK(hipStreamSynchronize(s));

    c->samp_scratch.alloc(batch);
    sampler_scratch_init(c->samp_scratch.p, batch, s);
    SamplerParams sp{};
    sp.scratch = c->samp_scratch.p;
    sampler_plan(c->V, &sp.n_chunks, &sp.chunk_w);
    sp.logits = c->logits.p; sp.e_buf = c->e_buf.p; sp.Vpad = c->Vpad; sp.vocab = c->V;
    sp.active_in = c->active.p; sp.window = ctx > 0 ? c->window.p : nullptr; sp.window_len = c->window_len.p; sp.ctx = ctx;
    sp.n_gen = c->n_gen.p; sp.tokens_out = c->tokens_out.p; sp.tokens_stride = max_tokens;
    sp.all_ids = c->all_ids.p; sp.all_len = c->all_len.p; sp.all_stride = all_stride;
    sp.next_ids = c->ids.p; sp.active = c->active.p; sp.done_count = c->done_count.p;
    sp.temperature = gp->temperature; sp.top_p = gp->top_p; sp.penalty = gp->repetition_penalty;
    sp.seed = gp->seed; sp.row_offset = gp->row_offset; sp.frame_constrained = gp->frame_constrained;
    sp.lo = 0; sp.hi = 0; sp.max_tokens = max_tokens;
    sp.eos_id = hidden_mode ? hm->stop_id : (c->cfg.end_of_speech_id > 0 ? c->cfg.end_of_speech_id : ORPHEUS_END_OF_SPEECH);
    sp.audio_offset = c->cfg.audio_token_offset;
    if (gp->sampler_flavor == 1) {
        c->samp_l32.alloc((size_t)c->Mpad * c->Vpad);
        sp.penalty_flavor = 1; sp.top_p = 1.0f; sp.logits32 = c->samp_l32.p;
    }
    DevBuf<int32_t> hid_count;
    int64_t hid_rows = max_tokens + 1;
    if (hidden_mode) {
        hm->hidden->alloc((size_t)batch * hid_rows * c->d);
        hid_count.alloc(batch);
        std::vector<int32_t> ones_i(batch, 1);           // slot 0 = last prompt token (Soprano.swift:824-825)
        HIP_CHECK(hipMemcpyAsync(hid_count.p, ones_i.data(), batch * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    sampler_resolve(sp, multi_launch_only);              // (environment switches read once per call; part of the graph key through sp)
    c->sp = sp;
    {   // the captured graphs bake in every pointer and scalar below: re-capture when any of them changes
        uint64_t key = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t n) { const unsigned char* q = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { key ^= q[i]; key *= 1099511628211ull; } };
        mix(&sp, sizeof(sp));
        const void* ptrs[] = {c->prompt_mat.p, c->prompt_lens.p, c->step_counter.p, c->ids.p, c->active.p, c->h.p, c->x.p,
                              c->attn_out.p, c->act.p, c->logits.p, c->qkv_part.p, c->part.p, c->kcache.p, c->vtcache.p,
                              c->rope_cos.p, c->rope_sin.p, c->pos_cur.p, c->pos_next.p};
        mix(ptrs, sizeof(ptrs));
        int ints[] = {batch, Lmax, max_tokens, c->Mpad, c->Smax, c->S_qkv, c->S_o, c->S_down, hidden_mode ? 1 : 0,
                      c->a_qkv.R, c->a_qkv.ksb, c->a_qkv.U, c->a_o.R, c->a_o.ksb, c->a_o.U, c->a_down.R, c->a_down.ksb, c->a_down.U,
                      c->a_head.R, c->a_head.ksb, c->a_head.U, c->r_gu, c->ksb_gu, c->qa_gu.R, c->qa_gu.ksb, c->qa_head.R, c->qa_head.ksb};
        mix(ints, sizeof(ints));
        const void* hp[] = {hidden_mode ? (const void*)hm->hidden->p : nullptr, hidden_mode ? (const void*)hid_count.p : nullptr};
        mix(hp, sizeof(hp));
        if (key != c->graph_key) destroy_graphs(c);
        c->graph_key = key;
    }

    auto collect = [&](int prefill) {
        if (hidden_mode)
            hipLaunchKernelGGL(k_collect_hidden, dim3(batch), dim3(256), 0, s, c->x.p, c->active.p, hid_count.p, hm->hidden->p,
                               hid_rows, c->d, c->Mpad / 16, prefill);
    };
    auto prefill_body = [&]() {
        launch_prefill_feed(c->prompt_mat.p, c->prompt_lens.p, Lmax, c->step_counter.p, c->ids.p, c->active.p, batch, s);
        enqueue_layers(c);
        collect(1);
    };
    auto decode_body = [&]() {
        enqueue_lm_head(c);
        launch_sampler(c->sp, batch, s);
        enqueue_layers(c);
        collect(0);
    };
    auto capture = [&](hipGraphExec_t* exec, auto&& body) {
        hipGraph_t g = nullptr;
        HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        try { body(); } catch (...) { hipGraph_t dead = nullptr; (void)hipStreamEndCapture(s, &dead); if (dead) (void)hipGraphDestroy(dead); throw; }
        HIP_CHECK(hipStreamEndCapture(s, &g));
        HIP_CHECK(hipGraphInstantiate(exec, g, nullptr, nullptr, 0));
        HIP_CHECK(hipGraphDestroy(g));
    };

    hipEvent_t ev[4];
    for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
    auto t_host0 = std::chrono::steady_clock::now();
    HIP_CHECK(hipEventRecord(ev[0], s));
    // ---- prefill: Lmax steps of the ragged, left-padded batch (prefill :711)
    if (prefill_batched_ok(c, Lmax)) {
        prefill_batched(c, c->prompt_mat.p, c->prompt_lens.p, lens, Lmax);
        collect(1);
    } else {
        if (c->use_graph && !c->g_prefill) capture(&c->g_prefill, prefill_body);
        for (int j = 0; j < Lmax; ++j) {
            if (c->use_graph) HIP_CHECK(hipGraphLaunch(c->g_prefill, s)); else prefill_body();
        }
    }
    HIP_CHECK(hipEventRecord(ev[1], s));
    // after the last prompt token every row is active
    {
        std::vector<uint8_t> ones(batch, 1);
        HIP_CHECK(hipMemcpyAsync(c->active.p, ones.data(), batch, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    // ---- decode loop (:714-744)
    if (c->use_graph && !c->g_decode) capture(&c->g_decode, decode_body);
    // (Several decode steps per graph launch were measured and removed: between two hipGraphLaunch calls of the one-step graph the
    // device idles for 8.2 us - profiles/r04/c1_gaps.json - and yet eight steps per graph are SLOWER, 2.1047 against 2.0953 ms per step
    // over five A/B runs, profiles/r04/c2_ab.json: a 1376-node graph costs more per node than its seven removed seams save.)
    int steps = 0;
    const int poll = cb ? 8 : 32;
    std::vector<int32_t> host_ngen(batch, 0), host_tok;
    bool cancelled = false;
    PinnedBuf<int32_t> done_pin(1);
    PinnedBuf<unsigned> fail_pin(batch);                 // the one-launch sampler's per-row time-out flags, read back with every poll
    int32_t* done_host = done_pin.p;
    *done_host = 0;
    while (steps < max_tokens) {
        int chunk = std::min(poll, max_tokens - steps);
        for (int i = 0; i < chunk; ++i) {
            if (c->use_graph) HIP_CHECK(hipGraphLaunch(c->g_decode, s)); else decode_body();
        }
        steps += chunk;
        HIP_CHECK(hipMemcpyAsync(done_host, c->done_count.p, 4, hipMemcpyDeviceToHost, s));
        sampler_fail_flags_async(c->samp_scratch.p, batch, fail_pin.p, s);
        if (cb) {
            host_tok.resize((size_t)batch * max_tokens);
            HIP_CHECK(hipMemcpyAsync(host_ngen.data(), c->n_gen.p, batch * 4, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipMemcpyAsync(host_tok.data(), c->tokens_out.p, host_tok.size() * 4, hipMemcpyDeviceToHost, s));
        }
        HIP_CHECK(hipStreamSynchronize(s));
        if (sampler_fail_flags_any(fail_pin.p, batch)) {
            // nothing of this poll interval has been announced; the steps queued behind the failed one ended at once (sticky flag,
            // k_samp_cluster).  The reference's loop cannot fail for lack of free compute units (LlamaTTS.swift:714-744): the caller
            // runs the request again on kernels that do not wait for each other.
            sampler_note_failure(c->samp_scratch.p, batch, s);
            for (auto& e : ev) (void)hipEventDestroy(e);
            throw SamplerTimeout{};
        }
        if (cb) {
            for (int b = 0; b < batch; ++b)
                for (; emitted[b] < host_ngen[b]; ++emitted[b]) {
                    int32_t t = host_tok[(size_t)b * max_tokens + emitted[b]];
                    if (hidden_mode && t == hm->stop_id) continue;             // Soprano: [STOP] ends the row unannounced (Soprano.swift:855-857)
                    cb(user, b, MIS_EVENT_TOKEN, &t, 1);                       // .token(id), LlamaTTS.swift:862
                }
        }
        if (cancel && *cancel) { cancelled = true; break; }                    // Task.checkCancellation :715
        if (*done_host >= batch) break;
    }
    HIP_CHECK(hipEventRecord(ev[2], s));
    if (cancelled) {
        for (auto& e : ev) (void)hipEventDestroy(e);
        throw MisError(MIS_ERR_CANCELLED, "generation cancelled");
    }

    if (hidden_mode) {
        hm->n_hidden.resize(batch);
        out.n_tokens.resize(batch);
        HIP_CHECK(hipMemcpyAsync(hm->n_hidden.data(), hid_count.p, batch * 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(out.n_tokens.data(), c->n_gen.p, batch * 4, hipMemcpyDeviceToHost, s));
        if (want_tokens) {
            out.tokens.resize((size_t)batch * max_tokens);
            out.tokens_stride = max_tokens;
            HIP_CHECK(hipMemcpyAsync(out.tokens.data(), c->tokens_out.p, out.tokens.size() * 4, hipMemcpyDeviceToHost, s));
        }
        HIP_CHECK(hipStreamSynchronize(s));
        // the graphs reference hid_count / hidden: never reuse them after this call
        destroy_graphs(c);
        c->graph_key = 0;
        c->timing.prefill_ms = ms_between(ev[0], ev[1]);
        c->timing.decode_ms = ms_between(ev[1], ev[2]);
        c->timing.codec_ms = 0;
        c->timing.steps = steps;
        c->timing.step_ms_avg = steps ? c->timing.decode_ms / steps : 0;
        for (auto& e2 : ev) (void)hipEventDestroy(e2);
        return;
    }
    // ---- parseOutput (:749-752) + de-interleave (:41-64) + SNAC decode (:759)
    c->codes.alloc((size_t)batch * all_stride); c->n_codes.alloc(batch);
    {
        SpeechTokenIds tk{c->cfg.start_of_speech_id, c->cfg.end_of_speech_id, c->cfg.audio_token_offset, c->cfg.start_of_ai_id > 0 ? c->cfg.start_of_ai_id : -1};
        launch_orpheus_parse_output(c->all_ids.p, c->all_len.p, batch, all_stride, c->codes.p, c->n_codes.p, s,
                                    c->cfg.start_of_speech_id > 0 ? &tk : nullptr);
    }
    std::vector<int32_t> ncodes(batch);
    out.n_tokens.resize(batch);
    HIP_CHECK(hipMemcpyAsync(ncodes.data(), c->n_codes.p, batch * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(out.n_tokens.data(), c->n_gen.p, batch * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    int gmax = 0, gany = 0;
    for (int b = 0; b < batch; ++b) { gmax = std::max(gmax, ncodes[b] / 7); gany += ncodes[b] > 0; }
    MIS_REQUIRE(gany > 0, MIS_ERR_GENERATION_FAILED, "No audio codes generated");          // :754-756
    const int64_t hop = mis_snac_num_samples(c->codec, 1);
    out.pcm_lens.assign(batch, 0);
    for (int b = 0; b < batch; ++b) out.pcm_lens[b] = (int64_t)(ncodes[b] / 7) * hop;
    int64_t need = (int64_t)gmax * hop;
    DevBuf<float> own;   // (only used when the caller gave no device buffer)
    (void)own;
    MIS_REQUIRE(pcm_dev && pcm_stride >= need, MIS_ERR_INVALID_INPUT, "pcm buffer too small (%lld needed per row)", (long long)need);
    out.pcm_dev = pcm_dev; out.pcm_stride = pcm_stride;
    // rows with equal group counts decode together; padding a short row would change its tail samples
    std::vector<int> order(batch);
    for (int b = 0; b < batch; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return ncodes[a] < ncodes[b2]; });
    bool all_equal = ncodes[order.front()] == ncodes[order.back()];
    MIS_REQUIRE(!snac_noise || all_equal, MIS_ERR_INVALID_INPUT,
                "explicit SNAC noise requires all rows to produce the same number of frames");
    c->l0.alloc((size_t)batch * gmax); c->l1.alloc((size_t)batch * gmax * 2); c->l2.alloc((size_t)batch * gmax * 4);
    std::vector<const float*> dnoise;
    DevBuf<float> noise_dev[8];
    if (snac_noise) {
        for (int i = 0; i < sc->n_decoder_rates; ++i) {
            size_t n = (size_t)batch * mis_snac_noise_len(c->codec, i, gmax);
            noise_dev[i].alloc(n);
            HIP_CHECK(hipMemcpyAsync(noise_dev[i].p, snac_noise[i], n * 4, hipMemcpyDefault, s));
            dnoise.push_back(noise_dev[i].p);
        }
    }
    const int chunk = c->cfg.codec_chunk_groups;
    if (chunk > 0 && gmax > chunk) {
        // VyvoTTS decodeAudioFromCodes (Qwen3.swift:47-83): rows longer than `chunk` groups are decoded as INDEPENDENT chunks of
        // `chunk` groups (the last one shorter) whose samples are concatenated; shorter rows are a single chunk.  Per chunk index
        // the rows are grouped by their chunk length, gathered, decoded together and scattered to sample offset ci*chunk*hop.
        // Explicit noise: the caller's per-row tensors (sized for the whole utterance) are sliced at the same offsets; internal
        // noise: the chunk index is folded into the seed (the reference draws fresh noise per chunk).
        DevBuf<float> noise_stage[8];
        const int n_ch = (gmax + chunk - 1) / chunk;
        for (int ci = 0; ci < n_ch; ++ci) {
            std::vector<std::pair<int, int>> part;                      // (chunk length in groups, row)
            for (int b = 0; b < batch; ++b) {
                int g = ncodes[b] / 7;
                if (g > ci * chunk) part.push_back({std::min(chunk, g - ci * chunk), b});
            }
            std::stable_sort(part.begin(), part.end());
            size_t j0 = 0;
            while (j0 < part.size()) {
                size_t j1 = j0;
                while (j1 < part.size() && part[j1].first == part[j0].first) ++j1;
                const int gc = part[j0].first, nsub = (int)(j1 - j0);
                c->pcm_tmp.alloc((size_t)nsub * gc * hop);
                std::vector<int32_t> rows(nsub);
                for (int k = 0; k < nsub; ++k) {
                    const int b = part[j0 + k].second;
                    rows[k] = b;
                    launch_orpheus_deinterleave(c->codes.p + (size_t)b * all_stride + (size_t)ci * chunk * 7, 7 * gc, 1, gc,
                                                c->l0.p + (size_t)k * gc, c->l1.p + (size_t)k * gc * 2, c->l2.p + (size_t)k * gc * 4, gc, s);
                }
                const int32_t* cp[3] = {c->l0.p, c->l1.p, c->l2.p};
                std::vector<const float*> nz;
                if (snac_noise) {                                       // all rows have gmax groups here (checked above)
                    for (int i = 0; i < sc->n_decoder_rates; ++i) {
                        const size_t full = mis_snac_noise_len(c->codec, i, gmax), per_group = full / gmax, len = per_group * gc;
                        noise_stage[i].alloc((size_t)nsub * len);
                        for (int k = 0; k < nsub; ++k)
                            HIP_CHECK(hipMemcpyAsync(noise_stage[i].p + (size_t)k * len,
                                                     noise_dev[i].p + (size_t)rows[k] * full + (size_t)ci * chunk * per_group, len * 4,
                                                     hipMemcpyDeviceToDevice, s));
                        nz.push_back(noise_stage[i].p);
                    }
                }
                c->row_map.alloc(batch);
                HIP_CHECK(hipMemcpyAsync(c->row_map.p, rows.data(), nsub * sizeof(int32_t), hipMemcpyHostToDevice, s));
                snac_decode_device(c->codec, cp, nsub, gc, snac_noise ? nz.data() : nullptr, 1,
                                   gp->seed + 0x9E3779B97F4A7C15ull * (uint64_t)ci, c->row_map.p, gp->row_offset, c->pcm_tmp.p,
                                   (int64_t)gc * hop, s);
                for (int k = 0; k < nsub; ++k)
                    HIP_CHECK(hipMemcpyAsync(pcm_dev + (size_t)rows[k] * pcm_stride + (size_t)ci * chunk * hop,
                                             c->pcm_tmp.p + (size_t)k * gc * hop, (size_t)gc * hop * 4, hipMemcpyDeviceToDevice, s));
                HIP_CHECK(hipStreamSynchronize(s));                     // `rows` / staging are reused by the next sub-batch
                j0 = j1;
            }
        }
        order.clear();                                                  // the whole-utterance path below has nothing left to do
    }
    size_t i0 = 0;
    while (i0 < order.size()) {
        size_t i1 = i0;
        while (i1 < order.size() && ncodes[order[i1]] == ncodes[order[i0]]) ++i1;
        int g = ncodes[order[i0]] / 7, nsub = (int)(i1 - i0);
        if (g > 0) {
            if (all_equal) {
                launch_orpheus_deinterleave_ragged(c->codes.p, all_stride, c->n_codes.p, batch, c->l0.p, c->l1.p, c->l2.p, g, s);
                const int32_t* cp[3] = {c->l0.p, c->l1.p, c->l2.p};
                snac_decode_device(c->codec, cp, batch, g, snac_noise ? dnoise.data() : nullptr, 1, gp->seed, nullptr,
                                   gp->row_offset, pcm_dev, pcm_stride, s);
            } else {
                // gather the subset row by row (rare path: ragged EOS); decode the subset; scatter PCM rows
                c->pcm_tmp.alloc((size_t)nsub * g * hop);
                for (int k = 0; k < nsub; ++k) {
                    int b = order[i0 + k];
                    launch_orpheus_deinterleave_ragged(c->codes.p + (size_t)b * all_stride, all_stride, c->n_codes.p + b, 1,
                                                       c->l0.p + (size_t)k * g, c->l1.p + (size_t)k * g * 2,
                                                       c->l2.p + (size_t)k * g * 4, g, s);
                }
                const int32_t* cp[3] = {c->l0.p, c->l1.p, c->l2.p};
                c->row_map.alloc(batch);
                HIP_CHECK(hipMemcpyAsync(c->row_map.p, order.data() + i0, nsub * sizeof(int32_t), hipMemcpyHostToDevice, s));
                snac_decode_device(c->codec, cp, nsub, g, nullptr, 1, gp->seed, c->row_map.p, gp->row_offset, c->pcm_tmp.p,
                                   (int64_t)g * hop, s);
                for (int k = 0; k < nsub; ++k)
                    HIP_CHECK(hipMemcpyAsync(pcm_dev + (size_t)order[i0 + k] * pcm_stride, c->pcm_tmp.p + (size_t)k * g * hop,
                                             (size_t)g * hop * 4, hipMemcpyDeviceToDevice, s));
            }
        }
        i0 = i1;
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(ev[3], s));
    if (want_tokens) {
        out.tokens.resize((size_t)batch * max_tokens);
        out.tokens_stride = max_tokens;
        HIP_CHECK(hipMemcpyAsync(out.tokens.data(), c->tokens_out.p, out.tokens.size() * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));
    auto t_host1 = std::chrono::steady_clock::now();
    (void)t_host0; (void)t_host1;
    c->timing.prefill_ms = ms_between(ev[0], ev[1]);
    c->timing.decode_ms = ms_between(ev[1], ev[2]);
    c->timing.codec_ms = ms_between(ev[2], ev[3]);
    c->timing.steps = steps;
    c->timing.step_ms_avg = steps ? c->timing.decode_ms / steps : 0;
    {
        // bytes per weight element as streamed: 2 (bf16) or bits / 8 + 4 / 64 (codes + a bf16 scale and bias per 64 codes)
        auto bpe = [](const mis_tts::QRole& q) { return q.on ? q.bits / 8.0 + 4.0 / 64.0 : 2.0; };
        double w = (double)c->L * (bpe(c->q_qkv) * c->Nqkv * c->d + bpe(c->q_o) * c->d * c->H * c->D + bpe(c->q_gu) * 2.0 * c->ff * c->d +
                                   bpe(c->q_down) * c->ff * c->d) + bpe(c->q_head) * c->V * c->d;
        double mean_ctx = Lmax + steps / 2.0;
        double kvb = (double)batch * mean_ctx * c->L * 2.0 * c->Hkv * c->D * 2.0;
        c->timing.hbm_bytes_per_step = w + kvb;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
}

// generate = one attempt on the fastest kernels; if the one-launch sampler could not get its blocks co-resident (the only kernel of
// the step that needs that), the whole request once more on the multi-launch sampler - the request is deterministic, the second
// attempt's tokens are the ones the first would have produced, and the callback continues behind what it has already been told.
// A handle that shares its device with another replica's streams never takes the one-launch path in the first place.
static void run_generate(mis_tts* c, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                         const mis_gen_params* gp, const float* const* snac_noise, float* pcm_dev, int64_t pcm_stride,
                         bool want_tokens, mis_event_cb cb, void* user, const volatile int* cancel, GenOutputs& out,
                         HiddenMode* hm = nullptr) {
    std::vector<int32_t> emitted(std::max(batch, 1), 0);
    const bool multi_first = c && c->shared_device;
    try {
        run_generate_attempt(c, prompt_ids, prompt_lens, batch, gp, snac_noise, pcm_dev, pcm_stride, want_tokens, cb, user, cancel, out, hm,
                             multi_first, emitted);
    } catch (const SamplerTimeout&) {
        MIS_REQUIRE(!multi_first, MIS_ERR_GENERATION_FAILED, "sampler: time-out flag raised on the multi-launch path");   // (cannot happen: nothing spins there)
        out = GenOutputs{};
        run_generate_attempt(c, prompt_ids, prompt_lens, batch, gp, snac_noise, pcm_dev, pcm_stride, want_tokens, cb, user, cancel, out, hm,
                             true, emitted);
    }
}
void tts_internal_set_shared_device(mis_tts* c, bool shared) { if (c) c->shared_device = shared; }
bool tts_internal_shared_device(const mis_tts* c) { return c && c->shared_device; }

static void default_params_check(const mis_gen_params* p) {
    MIS_REQUIRE(p, MIS_ERR_INVALID_INPUT, "null generation parameters");
    MIS_REQUIRE(p->temperature >= 0.0f, MIS_ERR_INVALID_INPUT, "temperature must be >= 0");
}

extern "C" mis_status mis_tts_generate_device(mis_tts* c, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                              const mis_gen_params* params, const float* const* snac_noise, float* pcm_dev,
                                              int64_t pcm_stride, int64_t* pcm_lens, int32_t* n_tokens) {
    MIS_API_BEGIN
    MIS_REQUIRE(c, MIS_ERR_INVALID_INPUT, "null handle");
    default_params_check(params);
    GenOutputs out;
    run_generate(c, prompt_ids, prompt_lens, batch, params, snac_noise, pcm_dev, pcm_stride, false, nullptr, nullptr, nullptr, out);
    if (pcm_lens) for (int b = 0; b < batch; ++b) pcm_lens[b] = out.pcm_lens[b];
    if (n_tokens) for (int b = 0; b < batch; ++b) n_tokens[b] = out.n_tokens[b];
    MIS_API_END
}

// Upper bound of the audio one row can decode to.  parseOutput (LlamaTTS.swift:749-752) keeps everything after the LAST
// start-of-speech marker - or the whole sequence, prompt included, when the prompt carries none - so the prompt length counts.
static int64_t max_pcm_per_row(mis_tts* c, const mis_gen_params* p, const int32_t* prompt_lens, int batch) {
    MIS_REQUIRE(prompt_lens && batch >= 1, MIS_ERR_INVALID_INPUT, "null argument");
    std::vector<int32_t> lens(batch);
    HIP_CHECK(hipMemcpy(lens.data(), prompt_lens, (size_t)batch * 4, hipMemcpyDefault));
    int lmax = 0;
    for (int b = 0; b < batch; ++b) lmax = std::max(lmax, lens[b]);
    int mt = p->max_tokens > 0 ? p->max_tokens : 1200;
    return mis_snac_num_samples(c->codec, std::max(1, (mt + lmax) / 7 + 1));
}

extern "C" mis_status mis_tts_generate(mis_tts* c, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                       const mis_gen_params* params, const float* const* snac_noise, float** pcm_out,
                                       int64_t* pcm_stride, int64_t* pcm_lens, int32_t** tokens_out,
                                       int64_t* tokens_stride, int32_t* n_tokens) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && pcm_out && pcm_stride && pcm_lens, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(c->codec, MIS_ERR_NOT_INITIALIZED, "SNAC model not loaded");
    default_params_check(params);
    HIP_CHECK(hipSetDevice(c->device));
    int64_t cap = max_pcm_per_row(c, params, prompt_lens, batch);
    DevBuf<float> pcm;
    pcm.alloc((size_t)batch * cap);
    HIP_CHECK(hipMemsetAsync(pcm.p, 0, (size_t)batch * cap * 4, c->stream));
    GenOutputs out;
    run_generate(c, prompt_ids, prompt_lens, batch, params, snac_noise, pcm.p, cap, tokens_out != nullptr, nullptr, nullptr, nullptr, out);
    int64_t longest = 0;
    for (int b = 0; b < batch; ++b) longest = std::max(longest, out.pcm_lens[b]);
    PinnedBuf<float> host((size_t)batch * std::max<int64_t>(longest, 1));
    HIP_CHECK(hipMemcpy2D(host.p, (size_t)longest * 4, pcm.p, (size_t)cap * 4, (size_t)longest * 4, batch, hipMemcpyDeviceToHost));
    if (tokens_out) {
        PinnedBuf<int32_t> th(out.tokens.size() + 1);
        memcpy(th.p, out.tokens.data(), out.tokens.size() * 4);
        *tokens_out = th.release();
        if (tokens_stride) *tokens_stride = out.tokens_stride;
    }
    *pcm_out = host.release(); *pcm_stride = longest;
    for (int b = 0; b < batch; ++b) pcm_lens[b] = out.pcm_lens[b];
    if (n_tokens) for (int b = 0; b < batch; ++b) n_tokens[b] = out.n_tokens[b];
    MIS_API_END
}

extern "C" mis_status mis_tts_generate_stream(mis_tts* c, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                              const mis_gen_params* params, const float* const* snac_noise,
                                              mis_event_cb on_event, void* user, const volatile int* cancel_flag) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && on_event, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(c->codec, MIS_ERR_NOT_INITIALIZED, "SNAC model not loaded");
    default_params_check(params);
    HIP_CHECK(hipSetDevice(c->device));
    int64_t cap = max_pcm_per_row(c, params, prompt_lens, batch);
    DevBuf<float> pcm;
    pcm.alloc((size_t)batch * cap);
    GenOutputs out;
    run_generate(c, prompt_ids, prompt_lens, batch, params, snac_noise, pcm.p, cap, false, on_event, user, cancel_flag, out);
    std::vector<int32_t> lens(batch);
    HIP_CHECK(hipMemcpy(lens.data(), prompt_lens, batch * 4, hipMemcpyDefault));
    std::vector<float> host;
    for (int b = 0; b < batch; ++b) {
        mis_gen_info info{};
        info.prompt_token_count = lens[b];
        info.generation_token_count = out.n_tokens[b];
        info.prefill_time = c->timing.prefill_ms * 1e-3;
        info.generate_time = c->timing.decode_ms * 1e-3;
        info.tokens_per_second = info.generate_time > 0 ? out.n_tokens[b] / info.generate_time : 0;
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        info.peak_memory_gb = (double)(total_b - free_b) / 1e9;
        on_event(user, b, MIS_EVENT_INFO, &info, 1);                         // .info, LlamaTTS.swift:893-901
        host.resize((size_t)std::max<int64_t>(out.pcm_lens[b], 1));
        HIP_CHECK(hipMemcpy(host.data(), pcm.p + (size_t)b * cap, (size_t)out.pcm_lens[b] * 4, hipMemcpyDeviceToHost));
        on_event(user, b, MIS_EVENT_AUDIO, host.data(), out.pcm_lens[b]);   // ONE final .audio, :904
    }
    MIS_API_END
}

extern "C" mis_status mis_tts_set_profiling(mis_tts* c, int enabled) {
    MIS_API_BEGIN
    MIS_REQUIRE(c, MIS_ERR_INVALID_INPUT, "null handle");
    c->profiling = enabled;
    MIS_API_END
}
extern "C" mis_status mis_tts_last_timing(mis_tts* c, mis_tts_timing* out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && out, MIS_ERR_INVALID_INPUT, "null argument");
    *out = c->timing;
    MIS_API_END
}

extern "C" mis_status mis_tts_time_gemm(mis_tts* c, int which, int batch, int iters, double* avg_ms, double* bytes) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && avg_ms && bytes && iters >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "model not finalized");
    static const int attn_ctx = std::max(1, env_int("MIS_TIME_ATTN_CTX", 368));      // mean context of the C3 workload
    if (c->batch != batch || (which == 5 && c->Smax < attn_ctx + 1)) lm_reset(c, batch, which == 5 ? attn_ctx + 1 : 64);
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int d = c->d, HD = c->H * c->D, Mpad = c->Mpad;
    if (which == 5) {           // decode attention of every row at context attn_ctx (positions / active flags set once)
        std::vector<int32_t> pos(Mpad, attn_ctx);
        std::vector<uint8_t> act(Mpad, 0);
        for (int b = 0; b < batch; ++b) act[b] = 1;
        HIP_CHECK(hipMemcpyAsync(c->pos_cur.p, pos.data(), Mpad * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(c->active.p, act.data(), Mpad, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
    // rotate over the layers so consecutive launches stream DIFFERENT weights (the 256 MB Infinity Cache
    // must not serve them); lm_head (0.96 GB) exceeds the cache by itself.
    auto run = [&](int it) {
        static const int fixed = env_int("MIS_TIME_GEMM_FIXED_LAYER", -1);      // experiment: weights stay in the Infinity Cache
        size_t li = fixed >= 0 ? (size_t)fixed : (size_t)(it % c->L);
        switch (which) {
            case 0: gemm_qkv(c, li, s); break;
            case 1: gemm_o(c, li, s); break;
            case 2: gemm_gate_up(c, li, s); break;
            case 3: gemm_down(c, li, s); break;
            case 4: enqueue_lm_head(c); break;
            case 5: {
                AttnParams ap{};
                ap.qkv_part = c->qkv_part.p; ap.S = c->S_qkv; ap.Mpad = Mpad; ap.Nqkv = c->Nqkv;
                size_t lkv = (size_t)c->batch * c->Hkv * c->Smax * c->D;
                ap.kcache = c->kcache.p + lkv * li; ap.vtcache = c->vtcache.p + lkv * li;
                ap.pos = c->pos_cur.p; ap.active = c->active.p; ap.rope_cos = c->rope_cos.p; ap.rope_sin = c->rope_sin.p;
                ap.out = c->attn_out.p; ap.H = c->H; ap.Hkv = c->Hkv; ap.D = c->D; ap.Smax = c->Smax; ap.scale = 1.0f / sqrtf((float)c->D);
                if (c->cfg.qk_norm) { ap.qnorm_w = c->qknorm.p + (size_t)(2 * li) * c->D; ap.knorm_w = c->qknorm.p + (size_t)(2 * li + 1) * c->D; ap.qk_eps = c->cfg.rms_norm_eps; }
                ap.rope_in_dtype = c->cfg.rope_ops_in_dtype;
                launch_attn_decode(ap, c->batch, s);
                break;
            }
            case 6: launch_reduce_residual_rmsnorm(c->part.p, c->S_down, Mpad, d, c->h.p, c->norms.p + (size_t)(2 * li) * d, c->x.p, c->cfg.rms_norm_eps, s); break;
            default: throw MisError(MIS_ERR_INVALID_INPUT, "unknown kernel id");
        }
    };
    double b = 0;
    // algorithmic bytes of a weight matrix: dense bf16, or codes + one bf16 scale/bias pair per 64 inputs
    auto wbytes = [&](const mis_tts::QRole& R, double N, double K) { return R.on ? N * K * R.bits / 8.0 + N * (K / 64.0) * 4.0 : 2.0 * N * K; };
    switch (which) {
        case 0: b = wbytes(c->q_qkv, c->Nqkv, d); break;
        case 1: b = wbytes(c->q_o, d, HD); break;
        case 2: b = wbytes(c->q_gu, 2.0 * c->ff, d); break;
        case 3: b = wbytes(c->q_down, d, c->ff); break;
        case 5: b = (double)batch * (attn_ctx + 1) * 2.0 * c->Hkv * c->D * 2.0; break;      // K and V rows of every cached key, bf16
        case 6: b = (double)c->S_down * Mpad * d * 4.0 + 2.0 * (double)Mpad * d * 2.0; break;  // slabs + residual stream (cache resident)
        default: b = wbytes(c->q_head, c->V, d); break;
    }
    run(0);   // warm
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) run(i + 1);
    HIP_CHECK(hipEventRecord(e1, s));
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipGetLastError());
    *avg_ms = ms_between(e0, e1) / iters;
    *bytes = b;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    MIS_API_END
}

// LlamaTTSModel.fromModelDirectory, LlamaTTS.swift:942-977 (bf16 / f16 / f32 and MLX affine-quantised checkpoints)
extern "C" mis_status mis_tts_load(const char* model_dir, mis_snac* codec, int device, mis_tts** out) {
    mis_tts* c = nullptr;
    try {
        MIS_REQUIRE(model_dir && out, MIS_ERR_INVALID_INPUT, "null argument");
        std::string dir = model_dir;
        JsonValue j = json_parse(read_text_file(dir + "/config.json"));
        mis_lm_config cf{};
        cf.hidden_size = (int)j.number_or("hidden_size", 0);
        cf.num_hidden_layers = (int)j.number_or("num_hidden_layers", 0);
        cf.intermediate_size = (int)j.number_or("intermediate_size", 0);
        cf.num_attention_heads = (int)j.number_or("num_attention_heads", 0);
        cf.num_key_value_heads = (int)j.number_or("num_key_value_heads", cf.num_attention_heads);
        cf.head_dim = (int)j.number_or("head_dim", 0);
        cf.vocab_size = (int)j.number_or("vocab_size", 0);
        cf.rms_norm_eps = (float)j.number_or("rms_norm_eps", 1e-5);
        cf.rope_theta = (float)j.number_or("rope_theta", 10000.0);
        if (const JsonValue* rs = j.get("rope_scaling")) {
            if (rs->type == JsonValue::OBJ) {
                MIS_REQUIRE(rs->get("factor"), MIS_ERR_INVALID_INPUT, "rope_scaling must contain 'factor'");   // LlamaTTSConfig.swift:140-145
                cf.rope_factor = (float)rs->number_or("factor", 32.0);
                cf.rope_low_freq_factor = (float)rs->number_or("low_freq_factor", 1.0);
                cf.rope_high_freq_factor = (float)rs->number_or("high_freq_factor", 4.0);
                cf.rope_original_max_pos = (float)rs->number_or("original_max_position_embeddings", 8192.0);
            }
        }
        cf.tie_word_embeddings = j.bool_or("tie_word_embeddings", true) ? 1 : 0;                     // default true :28
        cf.sample_rate = (int)j.number_or("sample_rate", 24000);
        {   // Qwen3-style checkpoints (VyvoTTS): model_type "qwen3" => q/k norm + plain rope
            const JsonValue* mt = j.get("model_type");
            if (mt && mt->type == JsonValue::STR && mt->str.find("qwen3") != std::string::npos) {
                cf.qk_norm = 1; cf.rope_plain = 1;
                cf.start_of_speech_id = 151670; cf.end_of_speech_id = 151671; cf.audio_token_offset = 151679; cf.start_of_ai_id = 151674;   // Qwen3.swift:19-29
                cf.codec_chunk_groups = 50;                                                                                              // Qwen3.swift:47
            }
        }
        mis_status st = mis_tts_create(&cf, codec, device, &c);
        if (st != MIS_OK) return st;
        // MLX affine-quantised checkpoints: "quantization": {"group_size": g, "bits": b, "<module path>": {...} | false, ...}
        // (BaseConfiguration.perLayerQuantization; a module is quantised iff "<path>.scales" exists, LlamaTTS.swift:958-968)
        const JsonValue* qz = j.get("quantization");
        if (!qz) qz = j.get("quantization_config");
        int q_group = qz ? (int)qz->number_or("group_size", 64) : 0, q_bits = qz ? (int)qz->number_or("bits", 4) : 0;
        std::deque<SafeTensorFile> files;                       // deque: no relocation (the files own their mappings)
        for (auto& path : list_safetensors(dir)) { files.emplace_back(); files.back().open(path); }
        std::map<std::string, const SafeTensorEntry*> index;
        for (auto& f : files) for (auto& e : f.entries) index[e.name] = &e;
        for (auto& kv : index) {
            const SafeTensorEntry& e = *kv.second;
            const std::string& nm = e.name;
            auto ends = [&](const char* suf) { size_t l = strlen(suf); return nm.size() > l && nm.compare(nm.size() - l, l, suf) == 0; };
            if (ends(".scales") || ends(".biases")) {
                MIS_REQUIRE(qz, MIS_ERR_INVALID_INPUT, "%s present but config.json has no quantization entry", nm.c_str());
                continue;
            }
            const std::string base = ends(".weight") ? nm.substr(0, nm.size() - 7) : std::string();
            auto sc = base.empty() ? index.end() : index.find(base + ".scales");
            if (sc != index.end()) {
                auto bi = index.find(base + ".biases");
                MIS_REQUIRE(bi != index.end() && e.dtype == "U32" && e.shape.size() == 2 && sc->second->shape.size() == 2, MIS_ERR_INVALID_INPUT,
                            "malformed quantised tensor %s", nm.c_str());
                int g = q_group, b = q_bits;
                if (const JsonValue* ov = qz ? qz->get(base) : nullptr)
                    if (ov->type == JsonValue::OBJ) { g = (int)ov->number_or("group_size", g); b = (int)ov->number_or("bits", b); }
                const int64_t N = e.shape[0], K = sc->second->shape[1] * g;
                MIS_REQUIRE(b > 0 && e.shape[1] * 32 / b == K, MIS_ERR_INVALID_INPUT, "quantised tensor %s: packed width does not match bits/group_size", nm.c_str());
                st = mis_tts_set_tensor_quantized(c, nm.c_str(), (const uint32_t*)e.data, sc->second->data, bi->second->data,
                                                  dtype_from_safetensors(sc->second->dtype), N, K, g, b);
            } else {
                st = mis_tts_set_tensor(c, nm.c_str(), e.data, dtype_from_safetensors(e.dtype), e.shape.data(), (int)e.shape.size());
            }
            if (st != MIS_OK) { mis_tts_destroy(c); return st; }
        }
        st = mis_tts_finalize(c);
        if (st != MIS_OK) { mis_tts_destroy(c); return st; }
        *out = c;
        return MIS_OK;
    } catch (const MisError& e) {
        if (c) mis_tts_destroy(c);
        mis_set_error("%s", e.what());
        return e.code;
    } catch (const std::exception& e) {
        if (c) mis_tts_destroy(c);
        mis_set_error("%s", e.what());
        return MIS_ERR_GENERATION_FAILED;
    }
}

// ---------------------------------------------------------------------------- diagnostics
// Cost of a dependent kernel boundary on this machine: a hipGraph of n trivial kernels (mode 0: one
// 64-thread block doing one load + one store; mode 1: 32 blocks x 1024 threads, each thread one load +
// one store; mode 2: 1024 blocks x 256 threads) replayed `reps` times.  Returns microseconds per kernel.
__global__ void k_floor_probe(const float* __restrict__ a, float* __restrict__ b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i] + 1.0f;
}
extern "C" int32_t mis_debug_choose_split(int32_t items, int32_t k_tiles, int32_t waves_per_item, int32_t s_max) {
    if (items < 1 || k_tiles < 1 || (waves_per_item != 1 && waves_per_item != 4) || s_max < 1) return 0;
    return gemm_choose_split(items, k_tiles, waves_per_item, s_max);
}

extern "C" mis_status mis_debug_launch_floor(int device, int n_kernels, int mode, int reps, double* us_per_kernel) {
    MIS_API_BEGIN
    MIS_REQUIRE(us_per_kernel && n_kernels >= 1 && reps >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    HIP_CHECK(hipSetDevice(device));
    hipStream_t s;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    DevBuf<float> a, b;
    const int n = 1 << 18;
    a.alloc(n); b.alloc(n);
    HIP_CHECK(hipMemset(a.p, 0, n * 4));
    dim3 grid(mode == 0 ? 1 : (mode == 1 ? 32 : 1024)), block(mode == 0 ? 64 : (mode == 1 ? 1024 : 256));
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n_kernels; ++i) {
        const float* src = (i & 1) ? b.p : a.p;
        float* dst = (i & 1) ? a.p : b.p;
        hipLaunchKernelGGL(k_floor_probe, grid, block, 0, s, src, dst, n);
    }
    HIP_CHECK(hipStreamEndCapture(s, &g));
    HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HIP_CHECK(hipGraphLaunch(ge, s));
    HIP_CHECK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) HIP_CHECK(hipGraphLaunch(ge, s));
    HIP_CHECK(hipEventRecord(e1, s));
    HIP_CHECK(hipStreamSynchronize(s));
    *us_per_kernel = ms_between(e0, e1) * 1e3 / ((double)reps * n_kernels);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    (void)hipStreamDestroy(s);
    MIS_API_END
}

// ---------------------------------------------------------------------------- Soprano hooks (soprano.hip)
hipStream_t tts_stream(mis_tts* c) { return c->stream; }
int tts_hidden_size(const mis_tts* c) { return c->d; }
double tts_last_decode_ms(const mis_tts* c) { return c->timing.prefill_ms + c->timing.decode_ms; }
void tts_internal_set_decode_ms(mis_tts* c, double ms) { c->timing.prefill_ms = 0; c->timing.decode_ms = ms; }
int tts_device(const mis_tts* c) { return c->device; }
void tts_generate_hidden(mis_tts* c, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch, const mis_gen_params* gp,
                         int stop_id, DevBuf<float>& hidden, std::vector<int32_t>& n_hidden, std::vector<int32_t>& n_tokens,
                         std::vector<int32_t>& tokens, int64_t& tokens_stride, mis_event_cb cb, void* user, const volatile int* cancel) {
    HiddenMode hm;
    hm.on = true; hm.stop_id = stop_id; hm.hidden = &hidden;
    GenOutputs out;
    run_generate(c, prompt_ids, prompt_lens, batch, gp, nullptr, nullptr, 0, true, cb, user, cancel, out, &hm);
    n_hidden = hm.n_hidden; n_tokens = out.n_tokens; tokens = out.tokens; tokens_stride = out.tokens_stride;
}

// ---------------------------------------------------------------------------- hooks for composite engines (qwen3tts.hip)
void tts_internal_reset(mis_tts* c, int batch, int max_context) { lm_reset(c, batch, max_context); }
// batched prefill of right-aligned prompts given as input embeddings rows [Lmax][Mpad][d] (device; rows of positions a prompt does not
// have are ignored): the caches then hold the prompts, the packed x of the view holds the final norm of every row's last position
bool tts_internal_prefill_rows_ok(const mis_tts* c, int Lmax) { return prefill_batched_ok(c, Lmax); }
void tts_internal_prefill_rows(mis_tts* c, const bf16_t* rows, const int32_t* lens_host, int Lmax) {
    std::vector<int32_t> lens(lens_host, lens_host + c->batch);
    c->prompt_lens.alloc(c->Mpad);
    HIP_CHECK(hipMemcpyAsync(c->prompt_lens.p, lens.data(), (size_t)c->batch * 4, hipMemcpyHostToDevice, c->stream));
    prefill_batched(c, nullptr, c->prompt_lens.p, lens, Lmax, rows);
}
void tts_internal_use_stream(mis_tts* c, hipStream_t s) {
    if (c->stream && !c->borrowed_stream && c->stream != s) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    c->stream = s; c->borrowed_stream = true;
}
void tts_internal_enqueue_layers(mis_tts* c, const bf16_t* table, int table_rows, const int32_t* ids) { enqueue_layers(c, table, table_rows, ids); }
void tts_internal_enqueue_head(mis_tts* c, const bf16_t* head_packed) { enqueue_lm_head(c, head_packed); }
TtsWeightsView tts_internal_weights(mis_tts* c) {
    TtsWeightsView v{};
    v.emb = c->emb.p; v.wqkv = c->wqkv.p; v.wo = c->wo.p; v.wgu = c->wgu.p; v.wdown = c->wdown.p;
    v.head = c->lm_head.p; v.norms = c->norms.p; v.qknorm = c->qknorm.p;
    v.d = c->d; v.L = c->L; v.ff = c->ff; v.H = c->H; v.Hkv = c->Hkv; v.D = c->D; v.V = c->V; v.Vpad = c->Vpad; v.Nqkv = c->Nqkv;
    v.device = c->device; v.finalized = c->finalized ? 1 : 0; v.qk_norm = c->cfg.qk_norm ? 1 : 0; v.rope_plain = c->cfg.rope_plain ? 1 : 0;
    v.quantised = (c->q_qkv.on || c->q_o.on || c->q_gu.on || c->q_down.on || c->q_head.on) ? 1 : 0;
    v.eps = c->cfg.rms_norm_eps; v.stream = c->stream;
    return v;
}
void tts_internal_rope_tables(mis_tts* c, int max_context, const float** cos_out, const float** sin_out) {
    lm_reset(c, 1, max_context);
    *cos_out = c->rope_cos.p; *sin_out = c->rope_sin.p;
}
// The launch chain's prefill for ONE row (batched [positions x 1] pass where it applies), leaving the row's K/V in the engine's tiled caches:
// the batch-1 token engine imports them instead of walking the prompt position by position (token_engine.hip)
TtsKvView tts_internal_prefill_kv(mis_tts* c, const int32_t* prompt_host, int n, int max_context) {
    MIS_REQUIRE(c && c->finalized && prompt_host && n >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    hipStream_t s = c->stream;
    lm_reset(c, 1, std::max(max_context, n + 1));
    std::vector<int32_t> lens(1, n);
    c->prompt_mat.alloc(n); c->prompt_lens.alloc(1); c->step_counter.alloc(1);
    HIP_CHECK(hipMemcpyAsync(c->prompt_mat.p, prompt_host, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->prompt_lens.p, lens.data(), 4, hipMemcpyHostToDevice, s));
    c->step_counter.zero(s);
    HIP_CHECK(hipStreamSynchronize(s));
    if (prefill_batched_ok(c, n)) prefill_batched(c, c->prompt_mat.p, c->prompt_lens.p, lens, n, nullptr, true);       // (short prompts: prefill_small)
    else
        for (int j = 0; j < n; ++j) {
            launch_prefill_feed(c->prompt_mat.p, c->prompt_lens.p, n, c->step_counter.p, c->ids.p, c->active.p, 1, s);
            enqueue_layers(c);
        }
    HIP_CHECK(hipGetLastError());
    TtsKvView v{};
    v.kcache = c->kcache.p; v.vtcache = c->vtcache.p; v.Smax = c->Smax; v.Hkv = c->Hkv; v.D = c->D;
    v.layer_stride = (size_t)1 * c->Hkv * c->Smax * c->D;
    v.rope_cos = c->rope_cos.p; v.rope_sin = c->rope_sin.p;
    return v;
}
TtsView tts_internal_view(mis_tts* c) {
    TtsView v{};
    v.x = c->x.p; v.h = c->h.p; v.logits = c->logits.p; v.emb = c->emb.p; v.ids = c->ids.p; v.pos_next = c->pos_next.p;
    v.active = c->active.p; v.d = c->d; v.Mpad = c->Mpad; v.V = c->V; v.Vpad = c->Vpad; v.batch = c->batch; v.stream = c->stream;
    v.finalized = c->finalized ? 1 : 0; v.L = c->L;
    return v;
}
