// qwen3tts.hip - Qwen3-TTS generate: talker LM (input-embedding driven) + code predictor (15 sequential small-LM steps per
// 12.5 Hz frame) + on-device sampleToken, one hipGraph replay per frame; speech-tokenizer decode in q3_codec.hip.
//
// Reference being replaced: Qwen3TTSModel.generateVoiceDesign (Sources/MLXAudioTTS/Models/Qwen3TTS/Qwen3TTS.swift:306-569),
// prepareGenerationInputs (:883-1000), sampleToken (:1003-1118), Qwen3TTSTalkerForConditionalGeneration
// (Qwen3TTSTalker.swift:274-366), Qwen3TTSCodePredictor (Qwen3TTSCodePredictor.swift:195-243).  The reference syncs with the
// host once per frame (`eval(inputEmbeds, isEOS)` + `.item`, :481-485) and runs batch 1; here a frame of a whole batch is
// ~17 LM step chains + 16 sampler blocks inside one graph, and the host polls a done counter every few frames.
// Both LMs run on the weight-streaming step chain of lm_engine.hip (q/k-norm variant, RoPE as bf16 array ops).
#include "common.h"
#include "kernels.h"
#include "lm_kernels.h"
#include "q3_kernels.h"

#include <math.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <deque>
#include <functional>
#include <memory>
#include <set>

struct mis_qwen3tts {
    int device = 0;
    mis_qwen3tts_config cfg{};
    mis_tts* talker = nullptr;
    mis_tts* pred = nullptr;
    mis_q3dec* dec = nullptr;
    hipStream_t s = nullptr;
    hipStream_t s_dec = nullptr;            // speech-tokenizer decoder stream while the frame loop keeps running on s (generateStream)
    bool proj = false, finalized = false, stream_exact = false;
    int G = 0, d = 0, dp = 0, th = 0, Vt = 0, Vc = 0, Vp = 0, VpPad = 0;
    std::set<std::string> loaded;
    DevBuf<uint8_t> raw;
    DevBuf<bf16_t> stage;
    DevBuf<bf16_t> text_emb, fc1, fc2, fc1_b, fc2_b, proj_w, proj_b, codec_emb, codec_emb_proj;
    DevBuf<bf16_t> pred_emb[32], pred_emb_proj[32], pred_head[32];      // num_code_groups - 1 <= 31 tables / heads
    // per-call state
    DevBuf<bf16_t> tproj, in_emb, hid_rows, hid_proj, xpk, act, pf_rows;
    DevBuf<int32_t> iota, tidx, cidx, plen, trail_idx, trail_len, cur_codes, codes, n_frames, frame, step_counter, done, row_max,
        ids_tmp;
    DevBuf<uint8_t> seen;
    // in-context voice cloning: the reference-audio front end and the reference contexts of the handle (ReferenceAudioContext,
    // Qwen3TTS.swift:268-300).  Prompt rows [n_ref_rows][d] follow the codec vocabulary in the prefill feed.
    mis_q3ref* ref = nullptr;
    struct RefCtx { int speaker_row = -1, frame_row0 = 0, T = 0; std::vector<int32_t> codes; /* [G][T] */ };
    std::vector<RefCtx> refs;
    std::vector<int> ref_row_ctx;           // row -> context index
    DevBuf<bf16_t> ref_rows;
    int n_ref_rows = 0;
};

// ---------------------------------------------------------------------------- kernels
// rows of `table` ([*][K] bf16) selected by ids (or base + m) -> packed MFMA-B fragments of a [Mpad][K] activation
__global__ void k_q3_gather_pack(const bf16_t* __restrict__ table, int K, const int32_t* __restrict__ ids, int base, int n_valid,
                                 int table_rows, bf16_t* __restrict__ xpk, int MT) {
    const int m = blockIdx.x;
    int id = (m < n_valid) ? (ids ? ids[m] : base + m) : -1;
    if (id >= table_rows) id = -1;
    for (int k = threadIdx.x; k < K; k += blockDim.x)
        xpk[xpk_index(m, k, MT)] = id >= 0 ? table[(size_t)id * K + k] : (bf16_t)0;
}

__global__ void k_q3_iota(int32_t* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

// packed x (final-norm output of the step chain) -> row-major bf16 [Mpad][d]
__global__ void k_q3_unpack_x(const bf16_t* __restrict__ xpk, bf16_t* __restrict__ rows, int d, int MT) {
    const int m = blockIdx.x;
    for (int k = threadIdx.x; k < d; k += blockDim.x) rows[(size_t)m * d + k] = xpk[xpk_index(m, k, MT)];
}

// prefill position j of the right-aligned prompt matrix: row b is fed its position j - (Lmax - len[b]) (inactive before its
// first position).  Every position is text_proj(text_emb[t]) and/or codec_emb[c] (prepareGenerationInputs :883-1000).
__global__ void k_q3_prefill_feed(const int32_t* __restrict__ tidx, const int32_t* __restrict__ cidx, const int32_t* __restrict__ plen,
                                  int P, int Lmax, const int* __restrict__ step_counter, const bf16_t* __restrict__ tproj,
                                  const bf16_t* __restrict__ codec_emb, int Vc, const bf16_t* __restrict__ ref_rows, int n_ref_rows,
                                  bf16_t* __restrict__ in_emb, uint8_t* __restrict__ active, int d, int batch) {
    const int b = blockIdx.x;
    const int j = *step_counter;
    int idx = -1;
    if (b < batch) idx = j - (Lmax - plen[b]);
    const bool on = idx >= 0 && b < batch;
    if (threadIdx.x == 0) active[b] = on ? 1 : 0;
    const int t = on ? tidx[(size_t)b * P + idx] : -1;
    int c = on ? cidx[(size_t)b * P + idx] : -1;
    // ids past the codec vocabulary address the handle's reference rows (speaker vector / codecEmbedIcl frames, Qwen3TTS.swift:249-266)
    const bf16_t* crow = nullptr;
    if (c >= Vc) { if (c - Vc < n_ref_rows) crow = ref_rows + (size_t)(c - Vc) * d; }
    else if (c >= 0) crow = codec_emb + (size_t)c * d;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float v = 0.0f;
        if (t >= 0 && crow) v = bf16_round_f32(bf16_to_f32(tproj[(size_t)t * d + k]) + bf16_to_f32(crow[k]));
        else if (t >= 0) v = bf16_to_f32(tproj[(size_t)t * d + k]);
        else if (crow) v = bf16_to_f32(crow[k]);
        in_emb[(size_t)b * d + k] = f32_to_bf16(v);
    }
}                                                                // the counter is bumped by a separate launch (k_q3_bump)

// the whole right-aligned prompt matrix at once for the batched prefill: rows[(j * Mpad + b)] = what k_q3_prefill_feed feeds at position j
__global__ void k_q3_prefill_rows(const int32_t* __restrict__ tidx, const int32_t* __restrict__ cidx, const int32_t* __restrict__ plen, int P,
                                  int Lmax, const bf16_t* __restrict__ tproj, const bf16_t* __restrict__ codec_emb, int Vc,
                                  const bf16_t* __restrict__ ref_rows, int n_ref_rows, bf16_t* __restrict__ rows, int d, int batch, int Mpad) {
    const int b = blockIdx.x, j = blockIdx.y;
    int idx = -1;
    if (b < batch) idx = j - (Lmax - plen[b]);
    const bool on = idx >= 0 && b < batch;
    const int t = on ? tidx[(size_t)b * P + idx] : -1;
    int c = on ? cidx[(size_t)b * P + idx] : -1;
    const bf16_t* crow = nullptr;
    if (c >= Vc) { if (c - Vc < n_ref_rows) crow = ref_rows + (size_t)(c - Vc) * d; }
    else if (c >= 0) crow = codec_emb + (size_t)c * d;
    bf16_t* out = rows + ((size_t)j * Mpad + b) * d;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float v = 0.0f;
        if (t >= 0 && crow) v = bf16_round_f32(bf16_to_f32(tproj[(size_t)t * d + k]) + bf16_to_f32(crow[k]));
        else if (t >= 0) v = bf16_to_f32(tproj[(size_t)t * d + k]);
        else if (crow) v = bf16_to_f32(crow[k]);
        out[k] = f32_to_bf16(v);
    }
}

// codecEmbedIcl rows (Qwen3TTS.swift:249-262): rows[t] = codec_emb[code 0] + sum_i pred_emb[i][code i+1], one bf16 rounding per add
__global__ void k_q3_ref_rows(const int32_t* __restrict__ codes /*[nq][T]*/, int nq, int T, const bf16_t* __restrict__ codec_emb, int Vc,
                              const bf16_t* const* __restrict__ pred_emb, int Vp, int d, bf16_t* __restrict__ rows) {
    const int t = blockIdx.x;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float e = bf16_to_f32(codec_emb[(size_t)min(max(codes[t], 0), Vc - 1) * d + k]);
        for (int i = 0; i + 1 < nq; ++i) {
            const int ci = min(max(codes[(size_t)(i + 1) * T + t], 0), Vp - 1);
            e = bf16_round_f32(e + bf16_to_f32(pred_emb[i][(size_t)ci * d + k]));
        }
        rows[(size_t)t * d + k] = f32_to_bf16(e);
    }
}

__global__ void k_q3_bump(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = *p + 1; }

struct Q3NextArgs {
    const int32_t* cur_codes;     // [G][Mpad]
    int Mpad, G, d, Vc, Vp;
    const bf16_t* codec_emb;      // [Vc][d]   talker codec_embedding
    const bf16_t* const* pred_emb;   // device array of G-1 table pointers [Vp][d]
    const bf16_t* tproj;          // projected text rows
    const int32_t* trail_idx;     // [B][Tt] rows of tproj
    const int32_t* trail_len;     // [B]
    int Tt, pad_row;
    bf16_t* in_emb;               // [Mpad][d] next talker input
    int32_t* codes;               // [B][max_frames][G]
    int32_t* n_frames;            // [B]
    const int32_t* row_max;       // [B]
    int max_frames;
    uint8_t* active_a; uint8_t* active_b;
    int32_t* done_count;
};
// end of a frame: store the 16 codes, next input = text + codec_emb(code0) + sum_i pred_emb[i](code_{i+1}) with one bf16
// rounding per add in the reference's order (Qwen3TTS.swift:464-480)
__global__ void k_q3_next_input(Q3NextArgs a) {
    __shared__ int cs[32];
    const int b = blockIdx.x;
    if (!a.active_a[b]) return;
    const int f = a.n_frames[b];
    const int t = (f < a.trail_len[b]) ? a.trail_idx[(size_t)b * a.Tt + f] : a.pad_row;
    // the frame's codes once into LDS (clamped like the table lookups): the sums below then issue their G table reads back to back
    // instead of re-reading the code before every add (26 -> see profiles/r03/q3_small_kernels.json)
    if (threadIdx.x < a.G) {
        const int raw = a.cur_codes[(size_t)threadIdx.x * a.Mpad + b];
        cs[threadIdx.x] = min(raw, (threadIdx.x == 0 ? a.Vc : a.Vp) - 1);
        a.codes[((size_t)b * a.max_frames + f) * a.G + threadIdx.x] = raw;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < a.d; k += blockDim.x) {
        float e = bf16_to_f32(a.codec_emb[(size_t)cs[0] * a.d + k]);
        for (int i = 0; i + 1 < a.G; ++i) e = bf16_round_f32(e + bf16_to_f32(a.pred_emb[i][(size_t)cs[i + 1] * a.d + k]));
        a.in_emb[(size_t)b * a.d + k] = f32_to_bf16(bf16_to_f32(a.tproj[(size_t)t * a.d + k]) + e);
    }
    if (threadIdx.x == 0) {
        a.n_frames[b] = f + 1;
        if (f + 1 >= a.row_max[b]) {                               // `for step in 0 ..< effectiveMaxTokens` (:412)
            a.active_a[b] = 0; a.active_b[b] = 0;
            atomicAdd(a.done_count, 1);
        }
    }
}

// ---------------------------------------------------------------------------- model handle
static void upload_bf16(mis_qwen3tts* c, const void* data, mis_dtype dtype, size_t n, bf16_t* dst) {
    size_t esz = dtype == MIS_F32 ? 4 : 2;
    c->raw.alloc(n * esz);
    HIP_CHECK(hipMemcpyAsync(c->raw.p, data, n * esz, hipMemcpyDefault, c->s));
    launch_convert_to_bf16(c->raw.p, dtype, dst, n, c->s);
    HIP_CHECK(hipStreamSynchronize(c->s));
}

extern "C" mis_status mis_qwen3tts_create(const mis_qwen3tts_config* cfg, int device, mis_qwen3tts** out) {
    MIS_API_BEGIN
    MIS_REQUIRE(cfg && out, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(cfg->num_code_groups >= 2 && cfg->num_code_groups <= 32, MIS_ERR_INVALID_INPUT, "num_code_groups must be 2..32");
    MIS_REQUIRE(cfg->text_hidden_size > 0 && cfg->text_hidden_size % 32 == 0 && cfg->text_vocab_size > 0, MIS_ERR_INVALID_INPUT, "bad text dims");
    MIS_REQUIRE(cfg->talker.vocab_size > 1024 && cfg->talker.vocab_size <= 4096 && cfg->predictor.vocab_size <= 4096, MIS_ERR_INVALID_INPUT,
                "codec vocabularies must fit the in-register sampler (<= 4096)");
    mis_qwen3tts* c = new mis_qwen3tts();
    c->device = device; c->cfg = *cfg;
    mis_lm_config tc = cfg->talker, pc = cfg->predictor;
    tc.qk_norm = pc.qk_norm = 1; tc.rope_plain = pc.rope_plain = 1; tc.rope_ops_in_dtype = pc.rope_ops_in_dtype = 1;
    tc.tie_word_embeddings = pc.tie_word_embeddings = 0;
    mis_status st = mis_tts_create(&tc, nullptr, device, &c->talker);
    if (st == MIS_OK) st = mis_tts_create(&pc, nullptr, device, &c->pred);
    if (st == MIS_OK) st = mis_q3dec_create(cfg, device, &c->dec);
    if (st != MIS_OK) { mis_qwen3tts_destroy(c); return st; }
    c->cfg.talker = tc; c->cfg.predictor = pc;
    c->s = tts_stream(c->talker);
    tts_internal_use_stream(c->pred, c->s);
    HIP_CHECK(hipSetDevice(device));
    {   // the decoder stream yields to the frame loop: its kernels fill the gaps of the latency-bound LM chain instead of delaying it
        int lo = 0, hi = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));           // lo = least priority (numerically largest)
        HIP_CHECK(hipStreamCreateWithPriority(&c->s_dec, hipStreamNonBlocking, lo));
    }
    c->G = cfg->num_code_groups; c->d = tc.hidden_size; c->dp = pc.hidden_size; c->th = cfg->text_hidden_size;
    c->Vt = cfg->text_vocab_size; c->Vc = tc.vocab_size; c->Vp = pc.vocab_size; c->VpPad = (int)round_up(c->Vp, 16);
    c->proj = c->d != c->dp;
    HIP_CHECK(hipSetDevice(device));
    c->text_emb.alloc((size_t)c->Vt * c->th);
    c->fc1.alloc((size_t)c->th * c->th); c->fc1_b.alloc(c->th);
    c->fc2.alloc((size_t)c->d * c->th); c->fc2_b.alloc(c->d);
    c->codec_emb.alloc((size_t)c->Vc * c->d);
    for (int i = 0; i + 1 < c->G; ++i) { c->pred_emb[i].alloc((size_t)c->Vp * c->d); c->pred_head[i].alloc((size_t)c->VpPad * c->dp); }
    if (c->proj) { c->proj_w.alloc((size_t)c->dp * c->d); c->proj_b.alloc(c->dp); }
    {   // the predictor's own embedding / lm_head slots are unused (per-step tables and heads live here): satisfy its loader
        std::vector<bf16_t> z((size_t)c->Vp * c->dp, (bf16_t)0);
        int64_t sh[2] = {c->Vp, c->dp};
        st = mis_tts_set_tensor(c->pred, "model.embed_tokens.weight", z.data(), MIS_BF16, sh, 2);
        if (st == MIS_OK) st = mis_tts_set_tensor(c->pred, "lm_head.weight", z.data(), MIS_BF16, sh, 2);
        if (st != MIS_OK) { mis_qwen3tts_destroy(c); return st; }
    }
    *out = c;
    MIS_API_END
}

extern "C" void mis_qwen3tts_destroy(mis_qwen3tts* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->s) (void)hipStreamSynchronize(c->s);
    if (c->s_dec) { (void)hipStreamSynchronize(c->s_dec); (void)hipStreamDestroy(c->s_dec); }
    if (c->ref) q3ref_destroy(c->ref);
    if (c->dec) mis_q3dec_destroy(c->dec);
    if (c->pred) mis_tts_destroy(c->pred);          // borrows the talker's stream (or still owns its own if create failed early)
    if (c->talker) mis_tts_destroy(c->talker);
    delete c;
}
extern "C" mis_tts* mis_qwen3tts_talker(mis_qwen3tts* c) { return c ? c->talker : nullptr; }

static bool parse_indexed(const std::string& name, const char* prefix, const char* suffix, int* idx) {
    size_t pl = strlen(prefix), sl = strlen(suffix);
    if (name.size() <= pl + sl || name.compare(0, pl, prefix) != 0 || name.compare(name.size() - sl, sl, suffix) != 0) return false;
    std::string mid = name.substr(pl, name.size() - pl - sl);
    if (mid.empty() || mid.find_first_not_of("0123456789") != std::string::npos) return false;
    *idx = atoi(mid.c_str());
    return true;
}

extern "C" mis_status mis_qwen3tts_set_tensor(mis_qwen3tts* c, const char* name_, const void* data, mis_dtype dtype,
                                              const int64_t* shape, int ndim) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && name_ && data && shape && ndim >= 1, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(!c->finalized, MIS_ERR_INVALID_INPUT, "set_tensor after finalize");
    std::string name = name_;
    if (name.rfind("talker.", 0) == 0) name = name.substr(7);        // Qwen3TTSTalkerForConditionalGeneration.sanitize (:352-365)
    if (name.rfind("decoder.", 0) == 0) return mis_q3dec_set_tensor(c->dec, name.c_str(), data, dtype, shape, ndim);
    if (q3ref_owns(name.c_str())) {
        MIS_REQUIRE(c->ref, MIS_ERR_INVALID_INPUT, "tensor %s needs mis_qwen3tts_enable_reference first", name.c_str());
        q3ref_set_tensor(c->ref, name.c_str(), data, dtype, shape, ndim);
        return MIS_OK;
    }
    HIP_CHECK(hipSetDevice(c->device));
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { MIS_REQUIRE(shape[i] > 0, MIS_ERR_INVALID_INPUT, "bad shape"); n *= (size_t)shape[i]; }
    auto want2 = [&](int64_t a, int64_t b) {
        MIS_REQUIRE(ndim == 2 && shape[0] == a && shape[1] == b, MIS_ERR_INVALID_INPUT, "%s has the wrong shape", name.c_str());
    };
    auto want1 = [&](int64_t a) { MIS_REQUIRE(ndim == 1 && shape[0] == a, MIS_ERR_INVALID_INPUT, "%s has the wrong shape", name.c_str()); };
    auto packed = [&](bf16_t* dst, int N, int K) {
        c->stage.alloc((size_t)N * K);
        upload_bf16(c, data, dtype, (size_t)N * K, c->stage.p);
        launch_pack_weight(c->stage.p, dst, N, K, (int)round_up(N, 16) / 16, 1, 0, c->s);
        HIP_CHECK(hipStreamSynchronize(c->s));
    };
    int idx = -1;
    if (name == "model.codec_embedding.weight") {
        want2(c->Vc, c->d);
        upload_bf16(c, data, dtype, n, c->codec_emb.p);
        mis_status st = mis_tts_set_tensor(c->talker, "model.embed_tokens.weight", data, dtype, shape, ndim);
        if (st != MIS_OK) return st;
    } else if (name == "codec_head.weight") {
        mis_status st = mis_tts_set_tensor(c->talker, "lm_head.weight", data, dtype, shape, ndim);
        if (st != MIS_OK) return st;
    } else if (name == "model.text_embedding.weight") {
        want2(c->Vt, c->th); upload_bf16(c, data, dtype, n, c->text_emb.p);
    } else if (name == "text_projection.linear_fc1.weight") { want2(c->th, c->th); packed(c->fc1.p, c->th, c->th);
    } else if (name == "text_projection.linear_fc1.bias") { want1(c->th); upload_bf16(c, data, dtype, n, c->fc1_b.p);
    } else if (name == "text_projection.linear_fc2.weight") { want2(c->d, c->th); packed(c->fc2.p, c->d, c->th);
    } else if (name == "text_projection.linear_fc2.bias") { want1(c->d); upload_bf16(c, data, dtype, n, c->fc2_b.p);
    } else if (name == "code_predictor.small_to_mtp_projection.weight") {
        MIS_REQUIRE(c->proj, MIS_ERR_INVALID_INPUT, "unexpected tensor %s (talker and predictor widths are equal)", name.c_str());
        want2(c->dp, c->d); packed(c->proj_w.p, c->dp, c->d);
    } else if (name == "code_predictor.small_to_mtp_projection.bias") {
        MIS_REQUIRE(c->proj, MIS_ERR_INVALID_INPUT, "unexpected tensor %s", name.c_str());
        want1(c->dp); upload_bf16(c, data, dtype, n, c->proj_b.p);
    } else if (parse_indexed(name, "code_predictor.model.codec_embedding.", ".weight", &idx)) {
        MIS_REQUIRE(idx >= 0 && idx + 1 < c->G, MIS_ERR_INVALID_INPUT, "index out of range in %s", name.c_str());
        want2(c->Vp, c->d); upload_bf16(c, data, dtype, n, c->pred_emb[idx].p);
    } else if (parse_indexed(name, "code_predictor.lm_head.", ".weight", &idx)) {
        MIS_REQUIRE(idx >= 0 && idx + 1 < c->G, MIS_ERR_INVALID_INPUT, "index out of range in %s", name.c_str());
        want2(c->Vp, c->dp);
        HIP_CHECK(hipMemsetAsync(c->pred_head[idx].p, 0, (size_t)c->VpPad * c->dp * 2, c->s));
        packed(c->pred_head[idx].p, c->Vp, c->dp);
    } else if (name.rfind("code_predictor.model.", 0) == 0) {
        mis_status st = mis_tts_set_tensor(c->pred, name.substr(strlen("code_predictor.")).c_str(), data, dtype, shape, ndim);
        if (st != MIS_OK) return st;
    } else if (name.rfind("model.", 0) == 0) {
        mis_status st = mis_tts_set_tensor(c->talker, name.c_str(), data, dtype, shape, ndim);
        if (st != MIS_OK) return st;
    } else {
        throw MisError(MIS_ERR_INVALID_INPUT, "unexpected tensor " + name);
    }
    c->loaded.insert(name);
    MIS_API_END
}

// A tensor of a quantised checkpoint (mlx quantize: uint32 words + scales + biases, Qwen3TTS.swift:1157-1170).  The Linear layers of
// the two LMs (q/k/v/o/gate/up/down, codec_head) keep their quantised form and are streamed as codes (mis_tts_set_tensor_quantized);
// everything that is gathered or folded at load (embeddings, text projection, predictor tables and heads, mtp projection) is
// dequantised here to bf16 rows - for the embeddings that IS the reference's arithmetic (QuantizedEmbedding returns model-dtype rows).
extern "C" mis_status mis_qwen3tts_set_tensor_quantized(mis_qwen3tts* c, const char* name_, const uint32_t* wq, const void* scales,
                                                        const void* biases, mis_dtype sb_dtype, int64_t N, int64_t K, int group_size,
                                                        int bits) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && name_ && wq && scales && biases, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(!c->finalized, MIS_ERR_INVALID_INPUT, "set_tensor after finalize");
    std::string name = name_;
    if (name.rfind("talker.", 0) == 0) name = name.substr(7);
    const bool pred_lm = name.rfind("code_predictor.model.layers.", 0) == 0;
    const bool talker_lm = name.rfind("model.layers.", 0) == 0 || name == "codec_head.weight";
    if (pred_lm || talker_lm) {
        const std::string inner = pred_lm ? name.substr(strlen("code_predictor.")) : (name == "codec_head.weight" ? "lm_head.weight" : name);
        mis_status st = mis_tts_set_tensor_quantized(pred_lm ? c->pred : c->talker, inner.c_str(), wq, scales, biases, sb_dtype, N, K, group_size, bits);
        if (st != MIS_OK) return st;
        c->loaded.insert(name);
        return MIS_OK;
    }
    MIS_REQUIRE(bits == 2 || bits == 4 || bits == 8, MIS_ERR_INVALID_INPUT, "unsupported quantisation width %d", bits);
    MIS_REQUIRE(group_size >= 1 && N >= 1 && K >= 1 && K % group_size == 0 && K % (32 / bits) == 0, MIS_ERR_INVALID_INPUT, "bad quantised shape for %s", name_);
    HIP_CHECK(hipSetDevice(c->device));
    const size_t words = (size_t)N * K * bits / 32, ng = (size_t)N * (K / group_size), esz = sb_dtype == MIS_F32 ? 4 : 2;
    DevBuf<uint8_t> raw;
    DevBuf<bf16_t> rows;
    const size_t wb = round_up(words * 4, 16), sb = round_up(ng * esz, 16);
    raw.alloc(wb + 2 * sb); rows.alloc((size_t)N * K);
    HIP_CHECK(hipMemcpyAsync(raw.p, wq, words * 4, hipMemcpyDefault, c->s));
    HIP_CHECK(hipMemcpyAsync(raw.p + wb, scales, ng * esz, hipMemcpyDefault, c->s));
    HIP_CHECK(hipMemcpyAsync(raw.p + wb + sb, biases, ng * esz, hipMemcpyDefault, c->s));
    launch_dequant_affine((const uint32_t*)raw.p, raw.p + wb, raw.p + wb + sb, (int)sb_dtype, rows.p, (int)N, (int)K, group_size, bits, c->s);
    HIP_CHECK(hipStreamSynchronize(c->s));
    const int64_t shape[2] = {N, K};
    return mis_qwen3tts_set_tensor(c, name_, rows.p, MIS_BF16, shape, 2);
    MIS_API_END
}

// out[rows][dp] = T(table[rows][d] . Wp^T + b): the mtp projection applied once to every embedding table (identical per row to
// projecting after the gather, Qwen3TTSCodePredictor.swift:231-234)
static void project_table(mis_qwen3tts* c, const bf16_t* table, int rows, DevBuf<bf16_t>& out) {
    out.alloc((size_t)round_up(rows, 64) * c->dp);
    c->xpk.alloc((size_t)64 * std::max(c->d, c->th));
    for (int r0 = 0; r0 < rows; r0 += 64) {
        int nv = std::min(64, rows - r0);
        hipLaunchKernelGGL(k_q3_gather_pack, dim3(64), dim3(256), 0, c->s, table, c->d, (const int32_t*)nullptr, r0, nv, rows, c->xpk.p, 4);
        launch_gemm_skinny(EPI_BF16, 2, 4, c->proj_w.p, c->xpk.p, out.p + (size_t)r0 * c->dp, c->dp / 16, c->d / 32, 1, c->dp, 64, c->s, c->proj_b.p);
    }
    HIP_CHECK(hipGetLastError());
}

extern "C" mis_status mis_qwen3tts_finalize(mis_qwen3tts* c) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && !c->finalized, MIS_ERR_INVALID_INPUT, "bad handle");
    HIP_CHECK(hipSetDevice(c->device));
    std::vector<std::string> want = {"model.codec_embedding.weight", "codec_head.weight", "model.text_embedding.weight",
                                     "text_projection.linear_fc1.weight", "text_projection.linear_fc1.bias",
                                     "text_projection.linear_fc2.weight", "text_projection.linear_fc2.bias"};
    for (int i = 0; i + 1 < c->G; ++i) {
        want.push_back("code_predictor.model.codec_embedding." + std::to_string(i) + ".weight");
        want.push_back("code_predictor.lm_head." + std::to_string(i) + ".weight");
    }
    if (c->proj) { want.push_back("code_predictor.small_to_mtp_projection.weight"); want.push_back("code_predictor.small_to_mtp_projection.bias"); }
    for (auto& w : want) MIS_REQUIRE(c->loaded.count(w), MIS_ERR_NOT_INITIALIZED, "Qwen3-TTS weight missing: %s", w.c_str());
    mis_status st = mis_tts_finalize(c->talker);
    if (st == MIS_OK) st = mis_tts_finalize(c->pred);
    if (st == MIS_OK) st = mis_q3dec_finalize(c->dec);
    if (st != MIS_OK) return st;
    if (c->ref) q3ref_finalize(c->ref);
    if (c->proj) {
        project_table(c, c->codec_emb.p, c->Vc, c->codec_emb_proj);
        for (int i = 0; i + 1 < c->G; ++i) project_table(c, c->pred_emb[i].p, c->Vp, c->pred_emb_proj[i]);
    }
    HIP_CHECK(hipStreamSynchronize(c->s));
    c->raw.release(); c->stage.release();
    c->finalized = true;
    MIS_API_END
}

// ---------------------------------------------------------------------------- generate (codes)
// generateStream: the frame loop tells the caller whenever another `chunk_frames` frames of every row are final on the loop's
// stream (Qwen3TTS.swift:492-505), and lets it look for finished work after every host synchronisation.
struct Q3StreamHook {
    int chunk_frames = 0;
    std::function<void(int f0, int fn)> on_boundary;    // frames [f0, f0 + fn) were just enqueued on c->s
    std::function<void(int f)> pre_sync;                // f frames enqueued; the loop is about to synchronise with the host
    std::function<void(int f)> poll;                    // ... and has
    std::function<void()> loop_done;                    // every row has ended (before the frames after the last full chunk go out)
};

static void q3_text_projection(mis_qwen3tts* c, const std::vector<int32_t>& text_ids) {
    // textProjection(textEmbedding(ids)) for every text id of the call (ResizeMLP, Qwen3TTSTalker.swift:212-225), 64 rows a time
    const int n = (int)text_ids.size();
    c->tproj.alloc((size_t)round_up(n, 64) * c->d);
    c->ids_tmp.alloc(round_up(n, 64));
    HIP_CHECK(hipMemcpyAsync(c->ids_tmp.p, text_ids.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->s));
    c->xpk.alloc((size_t)64 * std::max(c->d, c->th));
    c->act.alloc((size_t)64 * c->th);
    for (int r0 = 0; r0 < n; r0 += 64) {
        int nv = std::min(64, n - r0);
        hipLaunchKernelGGL(k_q3_gather_pack, dim3(64), dim3(256), 0, c->s, c->text_emb.p, c->th, c->ids_tmp.p + r0, 0, nv, c->Vt, c->xpk.p, 4);
        launch_gemm_skinny(EPI_SILU_PACKED, 2, 4, c->fc1.p, c->xpk.p, c->act.p, c->th / 16, c->th / 32, 1, c->th, 64, c->s, c->fc1_b.p);
        launch_gemm_skinny(EPI_BF16, 2, 4, c->fc2.p, c->act.p, c->tproj.p + (size_t)r0 * c->d, c->d / 16, c->th / 32, 1, c->d, 64, c->s, c->fc2_b.p);
    }
    HIP_CHECK(hipGetLastError());
}

void q3_generate_codes(mis_qwen3tts* c, const int32_t* text_ids, const int32_t* codec_ids, const int32_t* prefill_lens, int P,
                       const int32_t* trailing_ids, const int32_t* trailing_lens, int Tt, int batch, const mis_qwen3tts_params* gp,
                       const int32_t* row_max_frames, std::vector<int32_t>& codes_host, std::vector<int32_t>& n_frames_host,
                       int* stride_out, const volatile int* cancel, Q3StreamHook* hook = nullptr) {
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "Qwen3-TTS model not finalized");
    MIS_REQUIRE(batch >= 1 && batch <= 64 && P >= 1 && Tt >= 0, MIS_ERR_INVALID_INPUT, "bad batch / prompt sizes");
    MIS_REQUIRE(gp->max_frames >= 1, MIS_ERR_INVALID_INPUT, "max_frames must be positive");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->s;
    const int G = c->G, d = c->d, max_frames = gp->max_frames;
    // ---- host prep: every distinct text position becomes a row of tproj
    std::vector<int32_t> tlist, tidx((size_t)batch * P, -1), cidx((size_t)batch * P, -1), tr((size_t)batch * std::max(Tt, 1), 0);
    int Lmax = 0;
    for (int b = 0; b < batch; ++b) {
        MIS_REQUIRE(prefill_lens[b] >= 1 && prefill_lens[b] <= P, MIS_ERR_INVALID_INPUT, "row %d: bad prefill length", b);
        MIS_REQUIRE(trailing_lens[b] >= 0 && trailing_lens[b] <= Tt, MIS_ERR_INVALID_INPUT, "row %d: bad trailing length", b);
        Lmax = std::max(Lmax, prefill_lens[b]);
        for (int p = 0; p < prefill_lens[b]; ++p) {
            int t = text_ids[(size_t)b * P + p], cc = codec_ids[(size_t)b * P + p];
            MIS_REQUIRE(t < c->Vt && cc < c->Vc + c->n_ref_rows && (t >= 0 || cc >= 0), MIS_ERR_INVALID_INPUT, "row %d position %d: bad ids", b, p);
            if (t >= 0) { tidx[(size_t)b * P + p] = (int)tlist.size(); tlist.push_back(t); }
            cidx[(size_t)b * P + p] = cc;
        }
        for (int j = 0; j < trailing_lens[b]; ++j) {
            int t = trailing_ids[(size_t)b * Tt + j];
            MIS_REQUIRE(t >= 0 && t < c->Vt, MIS_ERR_INVALID_INPUT, "row %d: bad trailing text id", b);
            tr[(size_t)b * std::max(Tt, 1) + j] = (int)tlist.size(); tlist.push_back(t);
        }
    }
    const int pad_row = (int)tlist.size();
    tlist.push_back(c->cfg.tts_pad_token_id);
    q3_text_projection(c, tlist);

    tts_internal_reset(c->talker, batch, Lmax + max_frames + 1);
    tts_internal_reset(c->pred, batch, 64);
    TtsView tv = tts_internal_view(c->talker), pv = tts_internal_view(c->pred);
    const int Mpad = tv.Mpad;
    c->in_emb.alloc((size_t)Mpad * d); c->hid_rows.alloc((size_t)Mpad * d); if (c->proj) c->hid_proj.alloc((size_t)Mpad * c->dp);
    c->iota.alloc(Mpad); c->tidx.alloc(tidx.size()); c->cidx.alloc(cidx.size()); c->plen.alloc(batch);
    c->trail_idx.alloc(tr.size()); c->trail_len.alloc(batch); c->cur_codes.alloc((size_t)G * Mpad);
    c->codes.alloc((size_t)batch * max_frames * G); c->n_frames.alloc(Mpad); c->frame.alloc(1); c->step_counter.alloc(1);
    c->done.alloc(1); c->row_max.alloc(batch); c->seen.alloc((size_t)Mpad * tv.Vpad);
    c->in_emb.zero(s); c->cur_codes.zero(s); c->codes.zero(s); c->n_frames.zero(s); c->frame.zero(s); c->step_counter.zero(s);
    c->done.zero(s); c->seen.zero(s);
    hipLaunchKernelGGL(k_q3_iota, dim3(1), dim3(64), 0, s, c->iota.p, Mpad);
    std::vector<int32_t> rmax(batch);
    for (int b = 0; b < batch; ++b) rmax[b] = row_max_frames ? std::max(1, std::min(row_max_frames[b], max_frames)) : max_frames;
    HIP_CHECK(hipMemcpyAsync(c->tidx.p, tidx.data(), tidx.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->cidx.p, cidx.data(), cidx.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->plen.p, prefill_lens, batch * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->trail_idx.p, tr.data(), tr.size() * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->trail_len.p, trailing_lens, batch * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(c->row_max.p, rmax.data(), batch * 4, hipMemcpyHostToDevice, s));
    // device array of the predictor embedding tables (unprojected: summed into the talker's next input)
    std::vector<const bf16_t*> ptabs(G - 1);
    for (int i = 0; i + 1 < G; ++i) ptabs[i] = c->pred_emb[i].p;
    DevBuf<const bf16_t*> ptabs_dev;
    ptabs_dev.alloc(G - 1);
    HIP_CHECK(hipMemcpyAsync(ptabs_dev.p, ptabs.data(), (G - 1) * sizeof(void*), hipMemcpyHostToDevice, s));

    auto capture = [&](hipGraphExec_t* exec, auto&& body) {
        hipGraph_t g = nullptr;
        HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        try { body(); } catch (...) { hipGraph_t dead = nullptr; (void)hipStreamEndCapture(s, &dead); if (dead) (void)hipGraphDestroy(dead); throw; }
        HIP_CHECK(hipStreamEndCapture(s, &g));
        HIP_CHECK(hipGraphInstantiate(exec, g, nullptr, nullptr, 0));
        HIP_CHECK(hipGraphDestroy(g));
    };
    const bool use_graph = getenv("MIS_NO_GRAPH") == nullptr;

    // ---- prefill: one talker position per replay over the right-aligned prompt matrix
    auto prefill_body = [&]() {
        hipLaunchKernelGGL(k_q3_prefill_feed, dim3(Mpad), dim3(256), 0, s, c->tidx.p, c->cidx.p, c->plen.p, P, Lmax, c->step_counter.p,
                           c->tproj.p, c->codec_emb.p, c->Vc, c->ref_rows.p, c->n_ref_rows, c->in_emb.p, tv.active, d, batch);
        hipLaunchKernelGGL(k_q3_bump, dim3(1), dim3(64), 0, s, c->step_counter.p);
        tts_internal_enqueue_layers(c->talker, c->in_emb.p, Mpad, c->iota.p);
    };
    hipGraphExec_t g_prefill = nullptr, g_frame = nullptr;
    try {
        if (tts_internal_prefill_rows_ok(c->talker, Lmax)) {
            // all prompt positions through the talker at once (lm_prefill.hip: [positions x rows] GEMMs on the packed weights); the
            // position-by-position graph below remains for quantised roles, odd widths and MIS_PREFILL_SEQ=1
            c->pf_rows.alloc((size_t)Lmax * Mpad * d);
            hipLaunchKernelGGL(k_q3_prefill_rows, dim3(Mpad, Lmax), dim3(256), 0, s, c->tidx.p, c->cidx.p, c->plen.p, P, Lmax, c->tproj.p,
                               c->codec_emb.p, c->Vc, c->ref_rows.p, c->n_ref_rows, c->pf_rows.p, d, batch, Mpad);
            tts_internal_prefill_rows(c->talker, c->pf_rows.p, prefill_lens, Lmax);
        } else {
            if (use_graph) capture(&g_prefill, prefill_body);
            for (int j = 0; j < Lmax; ++j) { if (use_graph) HIP_CHECK(hipGraphLaunch(g_prefill, s)); else prefill_body(); }
        }
        {
            std::vector<uint8_t> ones(Mpad, 0);
            for (int b = 0; b < batch; ++b) ones[b] = 1;
            HIP_CHECK(hipMemcpyAsync(tv.active, ones.data(), Mpad, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipMemcpyAsync(pv.active, ones.data(), Mpad, hipMemcpyHostToDevice, s));
        }
        // ---- one frame
        Q3SampleArgs sa{};
        sa.temperature = gp->temperature; sa.top_p = gp->top_p; sa.min_p = gp->min_p; sa.top_k = gp->top_k;
        sa.log_min_p = gp->min_p > 0.0f ? bf16_round_f32((float)log((double)gp->min_p)) : 0.0f;
        sa.seed = gp->seed; sa.row_offset = gp->row_offset; sa.frame = c->frame.p; sa.G = G; sa.cur_codes = c->cur_codes.p; sa.Mpad = Mpad;
        sa.active_a = tv.active; sa.active_b = pv.active; sa.done_count = c->done.p;
        auto frame_body = [&]() {
            // talker logits of the current position (the previous frame's / the prefill's final norm is in the packed x)
            tts_internal_enqueue_head(c->talker, nullptr);
            Q3SampleArgs t = sa;
            t.logits = tv.logits; t.Vpad = tv.Vpad; t.V = c->Vc; t.penalty = gp->repetition_penalty; t.seen = c->seen.p;
            t.sup_lo = c->Vc - 1024; t.sup_hi = c->Vc; t.eos = c->cfg.codec_eos_token_id; t.slot = 0;     // :402-404
            launch_q3_sample(t, batch, s);
            // code predictor: fresh cache, positions [talker hidden, codec_emb(code0)], then one embedding per step (:431-461)
            hipLaunchKernelGGL(k_q3_unpack_x, dim3(Mpad), dim3(256), 0, s, tv.x, c->hid_rows.p, d, Mpad / 16);
            const bf16_t* hid = c->hid_rows.p;
            if (c->proj) {
                launch_gemm_skinny(EPI_BF16, 2, 4, c->proj_w.p, tv.x, c->hid_proj.p, c->dp / 16, d / 32, 1, c->dp, Mpad, s, c->proj_b.p);
                hid = c->hid_proj.p;
            }
            HIP_CHECK(hipMemsetAsync(pv.pos_next, 0, (size_t)Mpad * 4, s));
            tts_internal_enqueue_layers(c->pred, hid, Mpad, c->iota.p);
            for (int i = 0; i + 1 < G; ++i) {
                const bf16_t* table = (i == 0) ? (c->proj ? c->codec_emb_proj.p : c->codec_emb.p)
                                               : (c->proj ? c->pred_emb_proj[i - 1].p : c->pred_emb[i - 1].p);
                tts_internal_enqueue_layers(c->pred, table, i == 0 ? c->Vc : c->Vp, c->cur_codes.p + (size_t)i * Mpad);
                tts_internal_enqueue_head(c->pred, c->pred_head[i].p);
                Q3SampleArgs p = sa;
                p.logits = pv.logits; p.Vpad = pv.Vpad; p.V = c->Vp; p.penalty = 1.0f; p.seen = nullptr;
                p.sup_lo = p.sup_hi = 0; p.eos = -1; p.slot = i + 1; p.done_count = nullptr;
                launch_q3_sample(p, batch, s);
            }
            Q3NextArgs na{};
            na.cur_codes = c->cur_codes.p; na.Mpad = Mpad; na.G = G; na.d = d; na.Vc = c->Vc; na.Vp = c->Vp; na.codec_emb = c->codec_emb.p;
            na.pred_emb = ptabs_dev.p; na.tproj = c->tproj.p; na.trail_idx = c->trail_idx.p; na.trail_len = c->trail_len.p;
            na.Tt = std::max(Tt, 1); na.pad_row = pad_row; na.in_emb = c->in_emb.p; na.codes = c->codes.p; na.n_frames = c->n_frames.p;
            na.row_max = c->row_max.p; na.max_frames = max_frames; na.active_a = tv.active; na.active_b = pv.active; na.done_count = c->done.p;
            hipLaunchKernelGGL(k_q3_next_input, dim3(batch), dim3(256), 0, s, na);
            hipLaunchKernelGGL(k_q3_bump, dim3(1), dim3(64), 0, s, c->frame.p);
            // the next frame's talker position
            tts_internal_enqueue_layers(c->talker, c->in_emb.p, Mpad, c->iota.p);
        };
        if (use_graph) capture(&g_frame, frame_body);
        PinnedBuf<int32_t> done_pin(1);
        int32_t* done_host = done_pin.p;
        *done_host = 0;
        int f = 0, last_boundary = 0;
        const int poll = 8;
        while (f < max_frames) {
            int chunk = std::min(poll, max_frames - f);
            if (hook) chunk = std::min(chunk, last_boundary + hook->chunk_frames - f);
            for (int i = 0; i < chunk; ++i) { if (use_graph) HIP_CHECK(hipGraphLaunch(g_frame, s)); else frame_body(); }
            f += chunk;
            if (hook && f - last_boundary == hook->chunk_frames) { hook->on_boundary(last_boundary, f - last_boundary); last_boundary = f; }
            HIP_CHECK(hipMemcpyAsync(done_host, c->done.p, 4, hipMemcpyDeviceToHost, s));
            if (hook) hook->pre_sync(f);
            HIP_CHECK(hipStreamSynchronize(s));
            if (hook) hook->poll(f);
            if (*done_host >= batch) break;
            if (cancel && *cancel) throw MisError(MIS_ERR_CANCELLED, "generation cancelled");
        }
        HIP_CHECK(hipGetLastError());
        if (hook) {   // the frames after the last full chunk (:537-546): rows still running there have <= f - last_boundary of them
            n_frames_host.resize(batch);
            HIP_CHECK(hipMemcpyAsync(n_frames_host.data(), c->n_frames.p, batch * 4, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            int longest = 0;
            for (int b = 0; b < batch; ++b) longest = std::max(longest, n_frames_host[b]);
            hook->loop_done();
            if (longest > last_boundary) hook->on_boundary(last_boundary, longest - last_boundary);
        }
    } catch (...) {
        if (g_prefill) (void)hipGraphExecDestroy(g_prefill);
        if (g_frame) (void)hipGraphExecDestroy(g_frame);
        throw;
    }
    if (g_prefill) (void)hipGraphExecDestroy(g_prefill);
    if (g_frame) (void)hipGraphExecDestroy(g_frame);
    codes_host.resize((size_t)batch * max_frames * G);
    n_frames_host.resize(batch);
    HIP_CHECK(hipMemcpyAsync(codes_host.data(), c->codes.p, codes_host.size() * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(n_frames_host.data(), c->n_frames.p, batch * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    *stride_out = max_frames;
}

extern "C" mis_status mis_qwen3tts_generate_codes(mis_qwen3tts* c, const int32_t* text_ids, const int32_t* codec_ids,
                                                  const int32_t* prefill_lens, int P, const int32_t* trailing_ids,
                                                  const int32_t* trailing_lens, int Tt, int batch, const mis_qwen3tts_params* params,
                                                  const int32_t* row_max_frames, int32_t** codes_out, int64_t* codes_stride,
                                                  int32_t* n_frames) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && text_ids && codec_ids && prefill_lens && trailing_lens && params && codes_out && codes_stride && n_frames,
                MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(Tt == 0 || trailing_ids, MIS_ERR_INVALID_INPUT, "null trailing ids");
    std::vector<int32_t> codes, nf;
    int stride = 0;
    q3_generate_codes(c, text_ids, codec_ids, prefill_lens, P, trailing_ids, trailing_lens, Tt, batch, params, row_max_frames, codes, nf,
                      &stride, nullptr);
    PinnedBuf<int32_t> host(codes.size() + 1);
    memcpy(host.p, codes.data(), codes.size() * 4);
    *codes_out = host.release(); *codes_stride = stride;
    for (int b = 0; b < batch; ++b) n_frames[b] = nf[b];
    MIS_API_END
}

// stand-alone sampleToken for parity tests: logits f32 [batch, vocab] (bf16-rounded on upload), seen u8 [batch, vocab] or NULL
extern "C" mis_status mis_qwen3tts_sample_logits(int device, const float* logits, int batch, int vocab, const uint8_t* seen,
                                                 const mis_qwen3tts_params* gp, int suppress_lo, int suppress_hi, int eos_id, int step,
                                                 int32_t* tokens_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(logits && gp && tokens_out && batch >= 1 && vocab >= 1 && vocab <= 4096, MIS_ERR_INVALID_INPUT, "bad argument");
    HIP_CHECK(hipSetDevice(device));
    const int Vpad = (int)round_up(vocab, 16), Mpad = (int)round_up(batch, 16);
    std::vector<bf16_t> lb((size_t)Mpad * Vpad, (bf16_t)0);
    for (int b = 0; b < batch; ++b) for (int i = 0; i < vocab; ++i) lb[(size_t)b * Vpad + i] = f32_to_bf16(logits[(size_t)b * vocab + i]);
    DevBuf<bf16_t> dl; DevBuf<uint8_t> ds, act; DevBuf<int32_t> cur, frame, tok;
    dl.alloc(lb.size()); ds.alloc((size_t)Mpad * Vpad); act.alloc(Mpad); cur.alloc(Mpad); frame.alloc(1); tok.alloc(Mpad);
    HIP_CHECK(hipMemcpy(dl.p, lb.data(), lb.size() * 2, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(ds.p, 0, (size_t)Mpad * Vpad));
    if (seen) for (int b = 0; b < batch; ++b) HIP_CHECK(hipMemcpy(ds.p + (size_t)b * Vpad, seen + (size_t)b * vocab, vocab, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(act.p, 1, Mpad));
    HIP_CHECK(hipMemcpy(frame.p, &step, 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(tok.p, 0, Mpad * 4));
    Q3SampleArgs a{};
    a.logits = dl.p; a.Vpad = Vpad; a.V = vocab; a.temperature = gp->temperature; a.top_p = gp->top_p; a.min_p = gp->min_p;
    a.penalty = gp->repetition_penalty; a.top_k = gp->top_k;
    a.log_min_p = gp->min_p > 0.0f ? bf16_round_f32((float)log((double)gp->min_p)) : 0.0f;
    a.sup_lo = suppress_lo; a.sup_hi = suppress_hi; a.eos = eos_id; a.seen = seen ? ds.p : nullptr; a.seed = gp->seed;
    a.row_offset = gp->row_offset; a.frame = frame.p; a.slot = 0; a.G = 1; a.cur_codes = cur.p; a.Mpad = Mpad; a.active_a = act.p;
    a.tokens_dbg = tok.p;
    launch_q3_sample(a, batch, 0);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpy(tokens_out, tok.p, batch * 4, hipMemcpyDeviceToHost));
    MIS_API_END
}

// ---------------------------------------------------------------------------- decode / generate
extern "C" int mis_qwen3tts_samples_per_frame(const mis_qwen3tts* c) { return c ? q3dec_total_upsample(c->dec) : 0; }
extern "C" int mis_qwen3tts_num_code_groups(const mis_qwen3tts* c) { return c ? c->G : 0; }

// ---------------------------------------------------------------------------- in-context voice cloning: front end + reference contexts
extern "C" mis_status mis_qwen3tts_enable_reference(mis_qwen3tts* c, const mis_qwen3tts_reference_config* cfg) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && cfg, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(!c->finalized && !c->ref, MIS_ERR_INVALID_INPUT, "enable_reference comes once, before finalize");
    c->ref = q3ref_create(cfg, c->device, c->s);
    MIS_API_END
}
// extractSpeakerEmbedding (Qwen3TTS.swift:839-881)
extern "C" mis_status mis_qwen3tts_speaker_embedding(mis_qwen3tts* c, const float* audio, int64_t n_samples, float* out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && audio && out, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(c->finalized && c->ref, MIS_ERR_NOT_INITIALIZED, "no reference front end on this handle (mis_qwen3tts_enable_reference)");
    q3ref_speaker(c->ref, audio, n_samples, -1, out, q3ref_speaker_dim(c->ref), nullptr, nullptr);
    MIS_API_END
}
// speechTokenizer.encode (Qwen3TTSSpeechTokenizer.swift:1052-1058 -> :868-881)
extern "C" mis_status mis_qwen3tts_encode_audio(mis_qwen3tts* c, const float* audio, int64_t n_samples, int32_t** codes_out, int32_t* n_q,
                                                int32_t* n_frames) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && audio && codes_out && n_q && n_frames, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(c->finalized && c->ref, MIS_ERR_NOT_INITIALIZED, "no reference front end on this handle (mis_qwen3tts_enable_reference)");
    std::vector<int32_t> codes;
    int nq = 0; int64_t T = 0;
    q3ref_encode(c->ref, audio, n_samples, -1, nullptr, 0, nullptr, &T, &codes, &nq);
    PinnedBuf<int32_t> host(codes.size() + 1);
    memcpy(host.p, codes.data(), codes.size() * 4);
    *codes_out = host.release(); *n_q = nq; *n_frames = (int32_t)T;
    MIS_API_END
}
extern "C" mis_status mis_qwen3tts_reference_tap(mis_qwen3tts* c, int kind, const float* audio, int64_t n_samples, int stage, float* out,
                                                 int64_t capacity, int32_t* channels, int64_t* length) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && audio && out && channels && length && stage >= 0 && (kind == 0 || kind == 1), MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(c->finalized && c->ref, MIS_ERR_NOT_INITIALIZED, "no reference front end on this handle (mis_qwen3tts_enable_reference)");
    int C = 0; int64_t T = 0;
    if (kind == 0) q3ref_speaker(c->ref, audio, n_samples, stage, out, capacity, &C, &T);
    else q3ref_encode(c->ref, audio, n_samples, stage, out, capacity, &C, &T, nullptr, nullptr);
    *channels = C; *length = T;
    MIS_API_END
}
extern "C" mis_status mis_qwen3tts_add_reference(mis_qwen3tts* c, const int32_t* codes, int n_q, int T, const float* speaker_embedding,
                                                 int speaker_dim, int32_t* speaker_row, int32_t* first_frame_row) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && codes && speaker_row && first_frame_row, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "Qwen3-TTS model not finalized");
    MIS_REQUIRE(n_q >= 1 && n_q <= c->G && T >= 1 && T <= 8192, MIS_ERR_INVALID_INPUT, "reference codes must be [1..%d, 1..8192]", c->G);
    MIS_REQUIRE(!speaker_embedding || speaker_dim == c->d, MIS_ERR_INVALID_INPUT, "speaker vector has %d entries, the talker is %d wide", speaker_dim, c->d);
    MIS_REQUIRE(c->refs.size() < 64, MIS_ERR_INVALID_INPUT, "too many reference contexts (mis_qwen3tts_clear_references)");
    HIP_CHECK(hipSetDevice(c->device));
    for (int i = 0; i < n_q * T; ++i)
        MIS_REQUIRE(codes[i] >= 0 && codes[i] < (i < T ? c->Vc : c->Vp), MIS_ERR_INVALID_INPUT, "reference code %d out of range", codes[i]);
    const int d = c->d, n_new = T + (speaker_embedding ? 1 : 0), n_old = c->n_ref_rows;
    {   // grow the row table, keeping what is there
        DevBuf<bf16_t> grown;
        grown.alloc((size_t)(n_old + n_new) * d);
        if (n_old) HIP_CHECK(hipMemcpyAsync(grown.p, c->ref_rows.p, (size_t)n_old * d * 2, hipMemcpyDeviceToDevice, c->s));
        HIP_CHECK(hipStreamSynchronize(c->s));
        std::swap(grown.p, c->ref_rows.p); std::swap(grown.n, c->ref_rows.n);
    }
    mis_qwen3tts::RefCtx ctx;
    int row = n_old;
    if (speaker_embedding) {
        std::vector<bf16_t> sv(d);
        for (int k = 0; k < d; ++k) sv[k] = f32_to_bf16(speaker_embedding[k]);
        HIP_CHECK(hipMemcpyAsync(c->ref_rows.p + (size_t)row * d, sv.data(), (size_t)d * 2, hipMemcpyHostToDevice, c->s));
        HIP_CHECK(hipStreamSynchronize(c->s));
        ctx.speaker_row = row++;
    }
    ctx.frame_row0 = row; ctx.T = T;
    ctx.codes.assign((size_t)c->G * T, 0);                           // missing groups decode as code 0
    memcpy(ctx.codes.data(), codes, (size_t)n_q * T * 4);
    {
        DevBuf<int32_t> cd; DevBuf<const bf16_t*> ptabs_dev;
        cd.alloc((size_t)n_q * T); ptabs_dev.alloc(c->G - 1);
        std::vector<const bf16_t*> ptabs(c->G - 1);
        for (int i = 0; i + 1 < c->G; ++i) ptabs[i] = c->pred_emb[i].p;
        HIP_CHECK(hipMemcpyAsync(cd.p, codes, (size_t)n_q * T * 4, hipMemcpyHostToDevice, c->s));
        HIP_CHECK(hipMemcpyAsync(ptabs_dev.p, ptabs.data(), (c->G - 1) * sizeof(void*), hipMemcpyHostToDevice, c->s));
        hipLaunchKernelGGL(k_q3_ref_rows, dim3(T), dim3(256), 0, c->s, cd.p, n_q, T, c->codec_emb.p, c->Vc, ptabs_dev.p, c->Vp, d,
                           c->ref_rows.p + (size_t)row * d);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(c->s));
    }
    c->n_ref_rows = n_old + n_new;
    c->ref_row_ctx.resize(c->n_ref_rows, (int)c->refs.size());
    *speaker_row = ctx.speaker_row; *first_frame_row = ctx.frame_row0;
    c->refs.push_back(std::move(ctx));
    MIS_API_END
}
extern "C" mis_status mis_qwen3tts_clear_references(mis_qwen3tts* c) {
    MIS_API_BEGIN
    MIS_REQUIRE(c, MIS_ERR_INVALID_INPUT, "null handle");
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipStreamSynchronize(c->s));
    c->refs.clear(); c->ref_row_ctx.clear(); c->n_ref_rows = 0;
    c->ref_rows.release();
    MIS_API_END
}

// Qwen3TTSSpeechTokenizerDecoder.callAsFunction / streamingStep over the whole sequence: codes int32 [batch, num_quantizers, T]
// (host or device) -> wav f32 [batch, T * samples_per_frame]
extern "C" mis_status mis_qwen3tts_decode(mis_qwen3tts* c, const int32_t* codes, int batch, int T, float* wav_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && codes && wav_out && batch >= 1 && T >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "Qwen3-TTS model not finalized");
    q3dec_decode_host(c->dec, codes, batch, T, wav_out, 0, nullptr, nullptr, c->s);
    MIS_API_END
}
// debug tap for parity tests: stage 1 quantizer, 2 transformer, 3 upsample, 4+i decoder block i; out f32 [batch, C, T'] (capacity floats)
extern "C" mis_status mis_qwen3tts_decoder_tap(mis_qwen3tts* c, const int32_t* codes, int batch, int T, int stage, float* out,
                                               int64_t capacity, int32_t* channels, int64_t* length) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && codes && out && channels && length && stage >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    int C = 0; int64_t Tt = 0;
    q3dec_decode_host(c->dec, codes, batch, T, nullptr, stage, &C, &Tt, c->s);      // sizes first
    MIS_REQUIRE((int64_t)batch * C * Tt <= capacity, MIS_ERR_INVALID_INPUT, "tap buffer too small");
    q3dec_decode_host(c->dec, codes, batch, T, out, stage, &C, &Tt, c->s);
    *channels = C; *length = Tt;
    MIS_API_END
}

// ---- streamingStep / resetStreamingState as a session on the handle (Qwen3TTSSpeechTokenizer.swift:948-1006)
// exact = 1: chunked decode bitwise equal to the whole-sequence decode; 0 (default): the reference's streaming arithmetic, whose
// overlap-add counts the transposed-conv bias twice on the first `stride` samples after every chunk boundary (:553-576).
extern "C" mis_status mis_qwen3tts_set_stream_exact(mis_qwen3tts* c, int exact) {
    MIS_API_BEGIN
    MIS_REQUIRE(c, MIS_ERR_INVALID_INPUT, "null handle");
    c->stream_exact = exact != 0;
    MIS_API_END
}
extern "C" mis_status mis_qwen3tts_decode_stream_begin(mis_qwen3tts* c, int batch, int max_frames, int max_chunk_frames) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && batch >= 1 && max_frames >= 1 && max_chunk_frames >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "Qwen3-TTS model not finalized");
    // (begin on an open session = resetStreamingState: the host may restart its own session at any time; what must not happen is
    // generate / generateStream taking the session over - checked there)
    q3dec_stream_begin(c->dec, batch, max_frames, max_chunk_frames, !c->stream_exact, c->s);
    HIP_CHECK(hipStreamSynchronize(c->s));
    MIS_API_END
}
// codes int32 [batch, num_quantizers, n_frames] (host or device): the NEXT n_frames frames of every row of the session ->
// wav_out f32 [batch, n_frames * samples_per_frame] (host or device)
extern "C" mis_status mis_qwen3tts_decode_stream_step(mis_qwen3tts* c, const int32_t* codes, int n_frames, float* wav_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && codes && wav_out && n_frames >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    q3dec_stream_step_host(c->dec, codes, n_frames, wav_out, c->s);
    MIS_API_END
}
extern "C" mis_status mis_qwen3tts_decode_stream_end(mis_qwen3tts* c) {
    MIS_API_BEGIN
    MIS_REQUIRE(c, MIS_ERR_INVALID_INPUT, "null handle");
    q3dec_stream_end(c->dec);
    MIS_API_END
}

// generateVoiceDesign for a batch of prepared prompts (Qwen3TTS.swift:306-569).
// on_event == NULL or chunk_frames <= 0: codes first, then ONE whole-sequence decode of the batch (rows right-padded to the
// longest: every layer is causal, so a row's samples do not depend on what follows them).
// on_event != NULL and chunk_frames > 0 (generateStream, streamingInterval * 12.5 frames, :394-395): whenever another chunk_frames
// frames exist, a streamingStep of the whole batch runs on a second stream WHILE the frame loop continues, and each row's new samples
// are delivered as MIS_EVENT_AUDIO as soon as they are on the host (:492-505); the frames after the last full chunk follow when the
// loop ends (:537-546).  A row's pcm is the concatenation of its chunks.
extern "C" mis_status mis_qwen3tts_generate(mis_qwen3tts* c, const int32_t* text_ids, const int32_t* codec_ids, const int32_t* prefill_lens,
                                            int P, const int32_t* trailing_ids, const int32_t* trailing_lens, int Tt, int batch,
                                            const mis_qwen3tts_params* params, const int32_t* row_max_frames, float** pcm_out,
                                            int64_t* pcm_stride, int64_t* pcm_lens, int32_t** codes_out, int64_t* codes_stride,
                                            int32_t* n_frames, int chunk_frames, mis_event_cb on_event, void* user,
                                            const volatile int* cancel_flag) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && text_ids && codec_ids && prefill_lens && trailing_lens && params && pcm_out && pcm_stride && pcm_lens,
                MIS_ERR_INVALID_INPUT, "null argument");
    std::vector<int32_t> codes, nf;
    int stride = 0;
    const int G = c->G, up = q3dec_total_upsample(c->dec);
    const bool streaming = on_event && chunk_frames > 0;
    MIS_REQUIRE(G == c->cfg.dec_num_quantizers, MIS_ERR_INVALID_INPUT, "num_code_groups (%d) != decoder quantizers (%d)", G, c->cfg.dec_num_quantizers);

    struct Chunk { int f0, fn; PinnedBuf<float> wav; PinnedBuf<int32_t> nf; hipEvent_t done = nullptr; bool emitted = false; };
    std::deque<std::unique_ptr<Chunk>> chunks;
    DevBuf<float> wav_dev;
    hipEvent_t ev_lm = nullptr;
    bool own_session = false;                       // the decode-stream session of the handle was opened by THIS call
    auto cleanup = [&]() {
        if (ev_lm) { (void)hipEventDestroy(ev_lm); ev_lm = nullptr; }
        for (auto& ch : chunks) if (ch->done) { (void)hipEventDestroy(ch->done); ch->done = nullptr; }
        if (own_session) { q3dec_stream_end(c->dec); own_session = false; }      // never a session the host opened itself
    };
    auto emit_ready = [&](bool wait) {
        for (auto& ch : chunks) {
            if (ch->emitted) continue;
            if (wait) HIP_CHECK(hipEventSynchronize(ch->done));
            else if (hipEventQuery(ch->done) != hipSuccess) { (void)hipGetLastError(); break; }     // chunks finish in order
            for (int b = 0; b < batch; ++b) {
                const int valid = std::min(std::max(ch->nf.p[b] - ch->f0, 0), ch->fn);
                if (valid > 0) on_event(user, b, MIS_EVENT_AUDIO, ch->wav.p + (size_t)b * ch->fn * up, (int64_t)valid * up);
            }
            ch->emitted = true;
        }
    };
    Q3StreamHook hook;
    // MIS_EVENT_TOKEN: code 0 of every frame as the host learns of it (onToken, :484), the EOS id included when a row ends on it
    PinnedBuf<int32_t> tok_codes, tok_nf;
    std::vector<char> eos_sent(batch, 0);
    int tok_f = 0, tok_pending_f = 0;
    const auto t_start = std::chrono::steady_clock::now();
    try {
        if (streaming) {
            HIP_CHECK(hipSetDevice(c->device));
            const int cap = params->max_frames;
            MIS_REQUIRE(cap >= 1, MIS_ERR_INVALID_INPUT, "max_frames must be positive");
            MIS_REQUIRE(q3dec_stream_pos(c->dec) < 0, MIS_ERR_INVALID_INPUT,
                        "generateStream needs the handle's decode-stream session, but the host has one open (mis_qwen3tts_decode_stream_end first)");
            q3dec_stream_begin(c->dec, batch, cap, std::min(chunk_frames, cap), !c->stream_exact, c->s_dec);
            own_session = true;
            wav_dev.alloc((size_t)batch * std::min(chunk_frames, cap) * up);
            HIP_CHECK(hipEventCreateWithFlags(&ev_lm, hipEventDisableTiming));
            hook.chunk_frames = chunk_frames;
            hook.on_boundary = [&](int f0, int fn) {
                auto ch = std::make_unique<Chunk>();
                ch->f0 = f0; ch->fn = fn;
                ch->wav.alloc((size_t)batch * fn * up);
                ch->nf.alloc(batch);
                HIP_CHECK(hipEventCreateWithFlags(&ch->done, hipEventDisableTiming));
                HIP_CHECK(hipEventRecord(ev_lm, c->s));                  // frames < f0 + fn are final once this fires
                HIP_CHECK(hipStreamWaitEvent(c->s_dec, ev_lm, 0));
                // the loop's code store is [B][max_frames][G]: no transposition, the quantizer kernel takes the strides
                q3dec_stream_step(c->dec, c->codes.p + (size_t)f0 * G, (int64_t)params->max_frames * G, 1, G, fn, wav_dev.p, (int64_t)fn * up,
                                  c->s_dec);
                HIP_CHECK(hipMemcpyAsync(ch->wav.p, wav_dev.p, (size_t)batch * fn * up * 4, hipMemcpyDeviceToHost, c->s_dec));
                // rows that ended inside / before this chunk keep their count; running rows have >= f0 + fn (clipped at emission)
                HIP_CHECK(hipMemcpyAsync(ch->nf.p, c->n_frames.p, (size_t)batch * 4, hipMemcpyDeviceToHost, c->s_dec));
                HIP_CHECK(hipEventRecord(ch->done, c->s_dec));
                chunks.push_back(std::move(ch));
            };
            tok_codes.alloc((size_t)batch * 8 * G);
            tok_nf.alloc(batch);
            hook.pre_sync = [&](int f) {
                MIS_REQUIRE(f - tok_f <= 8, MIS_ERR_GENERATION_FAILED, "token window");
                if (f > tok_f)
                    HIP_CHECK(hipMemcpy2DAsync(tok_codes.p, (size_t)8 * G * 4, c->codes.p + (size_t)tok_f * G, (size_t)params->max_frames * G * 4,
                                               (size_t)(f - tok_f) * G * 4, batch, hipMemcpyDeviceToHost, c->s));
                HIP_CHECK(hipMemcpyAsync(tok_nf.p, c->n_frames.p, (size_t)batch * 4, hipMemcpyDeviceToHost, c->s));
                tok_pending_f = f;
            };
            hook.poll = [&](int f) {
                for (int b = 0; b < batch; ++b) {
                    const int cap = row_max_frames ? std::max(1, std::min(row_max_frames[b], params->max_frames)) : params->max_frames;
                    for (int i = tok_f; i < std::min(tok_nf.p[b], tok_pending_f); ++i)
                        on_event(user, b, MIS_EVENT_TOKEN, &tok_codes.p[((size_t)b * 8 + (i - tok_f)) * G], 1);
                    if (!eos_sent[b] && tok_nf.p[b] < std::min(tok_pending_f, cap)) {
                        const int32_t eos = c->cfg.codec_eos_token_id;
                        on_event(user, b, MIS_EVENT_TOKEN, &eos, 1);
                        eos_sent[b] = 1;
                    }
                }
                tok_f = tok_pending_f;
                (void)f;
                emit_ready(false);
            };
            hook.loop_done = [&]() {                                   // AudioGenerationInfo (:524-534): one per row
                const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
                size_t free_b = 0, total_b = 0;
                (void)hipMemGetInfo(&free_b, &total_b);
                for (int b = 0; b < batch; ++b) {
                    mis_gen_info info{};
                    info.generation_token_count = tok_nf.p[b];
                    info.generate_time = secs;
                    info.tokens_per_second = secs > 0 ? tok_nf.p[b] / secs : 0;
                    info.peak_memory_gb = (double)(total_b - free_b) / 1e9;
                    on_event(user, b, MIS_EVENT_INFO, &info, 1);
                }
            };
        }
        q3_generate_codes(c, text_ids, codec_ids, prefill_lens, P, trailing_ids, trailing_lens, Tt, batch, params, row_max_frames, codes, nf,
                          &stride, cancel_flag, streaming ? &hook : nullptr);
        if (streaming) emit_ready(true);
    } catch (...) {
        (void)hipStreamSynchronize(c->s_dec);
        cleanup();
        throw;
    }
    // in-context voice cloning, non-streaming tail (Qwen3TTS.swift:547-563): a row whose prompt holds reference frame rows decodes
    // [reference codes | generated codes] and loses the proportional head of the waveform; the streaming path decodes only what it generated
    std::vector<int> row_ref(batch, -1), ntot(nf.begin(), nf.end());
    std::vector<int64_t> cut(batch, 0);
    bool any_ref = false;
    if (!streaming && c->n_ref_rows > 0)
        for (int b = 0; b < batch; ++b)
            for (int p = 0; p < prefill_lens[b]; ++p) {
                const int cc = codec_ids[(size_t)b * P + p];
                if (cc < c->Vc) continue;
                const int ci = c->ref_row_ctx[cc - c->Vc];
                if (cc - c->Vc < c->refs[ci].frame_row0) continue;          // the speaker row alone does not make a row in-context
                MIS_REQUIRE(row_ref[b] < 0 || row_ref[b] == ci, MIS_ERR_INVALID_INPUT, "row %d: prompt mixes two reference contexts", b);
                row_ref[b] = ci; any_ref = true;
            }
    std::vector<int32_t> dcodes_host;
    int dstride = stride;
    if (any_ref) {
        dstride = 1;
        for (int b = 0; b < batch; ++b) {
            if (row_ref[b] >= 0 && nf[b] > 0) ntot[b] = c->refs[row_ref[b]].T + nf[b];
            dstride = std::max(dstride, ntot[b]);
        }
        dcodes_host.assign((size_t)batch * dstride * G, 0);
        for (int b = 0; b < batch; ++b) {
            int32_t* dst = dcodes_host.data() + (size_t)b * dstride * G;
            int R = 0;
            if (row_ref[b] >= 0 && nf[b] > 0) {
                const auto& ctx = c->refs[row_ref[b]];
                R = ctx.T;
                for (int t = 0; t < R; ++t) for (int g = 0; g < G; ++g) dst[(size_t)t * G + g] = ctx.codes[(size_t)g * R + t];
            }
            memcpy(dst + (size_t)R * G, codes.data() + (size_t)b * stride * G, (size_t)nf[b] * G * 4);
        }
    }
    const int32_t* dec_src_host = any_ref ? dcodes_host.data() : codes.data();
    int64_t longest = 0;
    for (int b = 0; b < batch; ++b) { pcm_lens[b] = (int64_t)nf[b] * up; longest = std::max(longest, pcm_lens[b]); }
    if (!streaming) {        // decodeChunk's validLen (:223-228): frames whose first code is > 0, times the upsample rate; a shorter
        longest = 0;         // non-zero count trims the tail (code 0 is read as padding there) - mirrored
        for (int b = 0; b < batch; ++b) {
            int64_t valid = 0, len = (int64_t)ntot[b] * up;
            for (int f = 0; f < ntot[b]; ++f) valid += dec_src_host[((size_t)b * dstride + f) * G] > 0;
            valid *= up;
            if (valid > 0 && valid < len) len = valid;
            if (row_ref[b] >= 0 && ntot[b] > 0) {
                const int64_t k = (int64_t)((double)c->refs[row_ref[b]].T / (double)std::max(ntot[b], 1) * (double)len);
                if (k > 0 && k < len) cut[b] = k;
            }
            pcm_lens[b] = len - cut[b];
            longest = std::max(longest, pcm_lens[b]);
        }
    }
    PinnedBuf<float> host_pin((size_t)std::max<int64_t>(longest, 1) * batch);
    float* host = host_pin.p;
    memset(host, 0, (size_t)std::max<int64_t>(longest, 1) * batch * 4);
    if (streaming) {
        for (auto& ch : chunks)
            for (int b = 0; b < batch; ++b) {
                const int valid = std::min(std::max(nf[b] - ch->f0, 0), ch->fn);
                if (valid > 0) memcpy(host + (size_t)b * longest + (size_t)ch->f0 * up, ch->wav.p + (size_t)b * ch->fn * up, (size_t)valid * up * 4);
            }
        cleanup();
        chunks.clear();
    } else {
        DevBuf<float> wav;
        DevBuf<int32_t> dcodes_dev;
        const int32_t* dec_src = c->codes.p;
        if (any_ref) {
            dcodes_dev.alloc(dcodes_host.size());
            HIP_CHECK(hipMemcpyAsync(dcodes_dev.p, dcodes_host.data(), dcodes_host.size() * 4, hipMemcpyHostToDevice, c->s));
            dec_src = dcodes_dev.p;
        }
        // consecutive rows decode together, right-padded to the slice's longest row (bounded by ~16 GB of activations)
        int b0 = 0;
        while (b0 < batch) {
            int n = ntot[b0], b1 = b0 + 1;
            auto fits = [&](int rows, int frames) { return (size_t)4 * 4 * 96 * (size_t)frames * up * rows <= ((size_t)16 << 30); };
            while (b1 < batch && b1 - b0 < 64 && fits(b1 - b0 + 1, std::max(n, ntot[b1]))) { n = std::max(n, ntot[b1]); ++b1; }
            const int gb = b1 - b0;
            if (n > 0) {                                                // generatedCodes.isEmpty -> zeros([1]) (:520-522): length 0 here
                wav.alloc((size_t)gb * n * up);
                // decodeChunk (Qwen3TTS.swift:214-231) = streamingDecode(chunkTokens: 300) (Qwen3TTSSpeechTokenizer.swift:1070-1091):
                // carried-state steps of 300 frames from a fresh state.  One step IS the whole-sequence decode; beyond 300 frames the
                // reference's overlap-add counts the block biases twice right behind every 300-frame boundary (dup_bias, see
                // q3_codec.hip) - mirrored unless mis_qwen3tts_set_stream_exact(1).  Rows of a slice share the boundaries (every row
                // starts at frame 0) and are right-padded: causal layers, so a row's samples do not depend on the padding.
                constexpr int kDecodeChunk = 300;
                const int32_t* src = dec_src + (size_t)b0 * dstride * G;
                if (n <= kDecodeChunk) {
                    q3dec_decode_strided(c->dec, src, (int64_t)dstride * G, 1, G, gb, n, wav.p, (int64_t)n * up, c->s);
                } else {
                    MIS_REQUIRE(q3dec_stream_pos(c->dec) < 0, MIS_ERR_INVALID_INPUT,
                                "decoding more than 300 frames uses the handle's decode-stream session, but the host has one open");
                    q3dec_stream_begin(c->dec, gb, n, kDecodeChunk, !c->stream_exact, c->s);
                    try {
                        for (int f0 = 0; f0 < n; f0 += kDecodeChunk) {
                            const int fn = std::min(kDecodeChunk, n - f0);
                            q3dec_stream_step(c->dec, src + (size_t)f0 * G, (int64_t)dstride * G, 1, G, fn, wav.p + (size_t)f0 * up, (int64_t)n * up, c->s);
                        }
                    } catch (...) { q3dec_stream_end(c->dec); throw; }
                    q3dec_stream_end(c->dec);
                }
                for (int r = 0; r < gb; ++r)
                    if (ntot[b0 + r] > 0 && pcm_lens[b0 + r] > 0)
                        HIP_CHECK(hipMemcpyAsync(host + (size_t)(b0 + r) * longest, wav.p + (size_t)r * n * up + cut[b0 + r],
                                                 (size_t)pcm_lens[b0 + r] * 4, hipMemcpyDeviceToHost, c->s));
                HIP_CHECK(hipStreamSynchronize(c->s));
            }
            if (cancel_flag && *cancel_flag) throw MisError(MIS_ERR_CANCELLED, "generation cancelled");
            b0 = b1;
        }
        if (on_event)                                                   // no chunking requested: one AUDIO event per row
            for (int b = 0; b < batch; ++b)
                if (nf[b] > 0) on_event(user, b, MIS_EVENT_AUDIO, host + (size_t)b * longest, pcm_lens[b]);
    }
    if (codes_out) {
        PinnedBuf<int32_t> ch(codes.size() + 1);
        memcpy(ch.p, codes.data(), codes.size() * 4);
        *codes_out = ch.release();
        if (codes_stride) *codes_stride = stride;
    }
    *pcm_out = host_pin.release(); *pcm_stride = longest;
    if (n_frames) for (int b = 0; b < batch; ++b) n_frames[b] = nf[b];
    MIS_API_END
}
