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
#include <set>

struct mis_qwen3tts {
    int device = 0;
    mis_qwen3tts_config cfg{};
    mis_tts* talker = nullptr;
    mis_tts* pred = nullptr;
    mis_q3dec* dec = nullptr;
    hipStream_t s = nullptr;
    bool proj = false, finalized = false;
    int G = 0, d = 0, dp = 0, th = 0, Vt = 0, Vc = 0, Vp = 0, VpPad = 0;
    std::set<std::string> loaded;
    DevBuf<uint8_t> raw;
    DevBuf<bf16_t> stage;
    DevBuf<bf16_t> text_emb, fc1, fc2, fc1_b, fc2_b, proj_w, proj_b, codec_emb, codec_emb_proj;
    DevBuf<bf16_t> pred_emb[32], pred_emb_proj[32], pred_head[32];      // num_code_groups - 1 <= 31 tables / heads
    // per-call state
    DevBuf<bf16_t> tproj, in_emb, hid_rows, hid_proj, xpk, act;
    DevBuf<int32_t> iota, tidx, cidx, plen, trail_idx, trail_len, cur_codes, codes, n_frames, frame, step_counter, done, row_max,
        ids_tmp;
    DevBuf<uint8_t> seen;
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
                                  const bf16_t* __restrict__ codec_emb, int Vc, bf16_t* __restrict__ in_emb,
                                  uint8_t* __restrict__ active, int d, int batch) {
    const int b = blockIdx.x;
    const int j = *step_counter;
    int idx = -1;
    if (b < batch) idx = j - (Lmax - plen[b]);
    const bool on = idx >= 0 && b < batch;
    if (threadIdx.x == 0) active[b] = on ? 1 : 0;
    const int t = on ? tidx[(size_t)b * P + idx] : -1;
    int c = on ? cidx[(size_t)b * P + idx] : -1;
    if (c >= Vc) c = -1;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float v = 0.0f;
        if (t >= 0 && c >= 0) v = bf16_round_f32(bf16_to_f32(tproj[(size_t)t * d + k]) + bf16_to_f32(codec_emb[(size_t)c * d + k]));
        else if (t >= 0) v = bf16_to_f32(tproj[(size_t)t * d + k]);
        else if (c >= 0) v = bf16_to_f32(codec_emb[(size_t)c * d + k]);
        in_emb[(size_t)b * d + k] = f32_to_bf16(v);
    }
}                                                                // the counter is bumped by a separate launch (k_q3_bump)

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
    const int b = blockIdx.x;
    if (!a.active_a[b]) return;
    const int f = a.n_frames[b];
    const int t = (f < a.trail_len[b]) ? a.trail_idx[(size_t)b * a.Tt + f] : a.pad_row;
    for (int k = threadIdx.x; k < a.d; k += blockDim.x) {
        int c0 = a.cur_codes[b];
        float e = bf16_to_f32(a.codec_emb[(size_t)min(c0, a.Vc - 1) * a.d + k]);
        for (int i = 0; i + 1 < a.G; ++i) {
            int ci = min(a.cur_codes[(size_t)(i + 1) * a.Mpad + b], a.Vp - 1);
            e = bf16_round_f32(e + bf16_to_f32(a.pred_emb[i][(size_t)ci * a.d + k]));
        }
        a.in_emb[(size_t)b * a.d + k] = f32_to_bf16(bf16_to_f32(a.tproj[(size_t)t * a.d + k]) + e);
    }
    if (threadIdx.x < a.G) a.codes[((size_t)b * a.max_frames + f) * a.G + threadIdx.x] = a.cur_codes[(size_t)threadIdx.x * a.Mpad + b];
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
struct Q3Run { int batch, P, Tt, Lmax, max_frames; };

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
                       int* stride_out, const volatile int* cancel) {
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
            MIS_REQUIRE(t < c->Vt && cc < c->Vc && (t >= 0 || cc >= 0), MIS_ERR_INVALID_INPUT, "row %d position %d: bad ids", b, p);
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
                           c->tproj.p, c->codec_emb.p, c->Vc, c->in_emb.p, tv.active, d, batch);
        hipLaunchKernelGGL(k_q3_bump, dim3(1), dim3(64), 0, s, c->step_counter.p);
        tts_internal_enqueue_layers(c->talker, c->in_emb.p, Mpad, c->iota.p);
    };
    hipGraphExec_t g_prefill = nullptr, g_frame = nullptr;
    try {
        if (use_graph) capture(&g_prefill, prefill_body);
        for (int j = 0; j < Lmax; ++j) { if (use_graph) HIP_CHECK(hipGraphLaunch(g_prefill, s)); else prefill_body(); }
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
        int f = 0;
        const int poll = 8;
        while (f < max_frames) {
            int chunk = std::min(poll, max_frames - f);
            for (int i = 0; i < chunk; ++i) { if (use_graph) HIP_CHECK(hipGraphLaunch(g_frame, s)); else frame_body(); }
            f += chunk;
            HIP_CHECK(hipMemcpyAsync(done_host, c->done.p, 4, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            if (*done_host >= batch) break;
            if (cancel && *cancel) throw MisError(MIS_ERR_CANCELLED, "generation cancelled");
        }
        HIP_CHECK(hipGetLastError());
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

// generateVoiceDesign for a batch of prepared prompts (Qwen3TTS.swift:306-569): codes, then the speech tokenizer per row.
// on_event (nullable): MIS_EVENT_AUDIO chunks of `chunk_frames` frames per row while decoding (streamingInterval * 12.5,
// :394-395,492-505), emitted after generation in row order; the concatenation of a row's chunks equals its pcm row.
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
    q3_generate_codes(c, text_ids, codec_ids, prefill_lens, P, trailing_ids, trailing_lens, Tt, batch, params, row_max_frames, codes, nf,
                      &stride, cancel_flag);
    const int G = c->G, up = q3dec_total_upsample(c->dec);
    int64_t longest = 0;
    for (int b = 0; b < batch; ++b) { pcm_lens[b] = (int64_t)nf[b] * up; longest = std::max(longest, pcm_lens[b]); }
    PinnedBuf<float> host_pin((size_t)std::max<int64_t>(longest, 1) * batch);
    float* host = host_pin.p;
    memset(host, 0, (size_t)std::max<int64_t>(longest, 1) * batch * 4);
    {
        DevBuf<float> wav;
        DevBuf<int32_t> cd;
        // rows with the same frame count decode together (bounded by ~16 GB of activations); ragged rows one by one
        std::vector<char> done_row(batch, 0);
        for (int b0 = 0; b0 < batch; ++b0) {
            const int n = nf[b0];
            if (done_row[b0] || n == 0) continue;                       // generatedCodes.isEmpty -> zeros([1]) (:520-522): length 0 here
            const size_t per_row = (size_t)4 * 4 * 96 * (size_t)n * up;            // 4 buffers x f32 x widest stage (approx.)
            const int cap = (int)std::max<size_t>(1, std::min<size_t>(64, ((size_t)16 << 30) / std::max<size_t>(per_row, 1)));
            std::vector<int> grp;
            for (int b = b0; b < batch && (int)grp.size() < cap; ++b) if (!done_row[b] && nf[b] == n) { grp.push_back(b); done_row[b] = 1; }
            const int gb = (int)grp.size();
            std::vector<int32_t> rows((size_t)gb * G * n);              // [frames][G] -> [G][frames] per row
            for (int r = 0; r < gb; ++r)
                for (int f = 0; f < n; ++f) for (int g = 0; g < G; ++g)
                    rows[((size_t)r * G + g) * n + f] = codes[((size_t)grp[r] * stride + f) * G + g];
            cd.alloc(rows.size()); wav.alloc((size_t)gb * n * up);
            HIP_CHECK(hipMemcpyAsync(cd.p, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, c->s));
            q3dec_decode_device(c->dec, cd.p, gb, n, wav.p, (int64_t)n * up, c->s);
            for (int r = 0; r < gb; ++r)
                HIP_CHECK(hipMemcpyAsync(host + (size_t)grp[r] * longest, wav.p + (size_t)r * n * up, (size_t)n * up * 4, hipMemcpyDeviceToHost, c->s));
            HIP_CHECK(hipStreamSynchronize(c->s));
            if (cancel_flag && *cancel_flag) throw MisError(MIS_ERR_CANCELLED, "generation cancelled");
        }
        if (on_event)
            for (int b = 0; b < batch; ++b) {
                const int n = nf[b], step = chunk_frames > 0 ? chunk_frames : std::max(n, 1);
                for (int f0 = 0; f0 < n; f0 += step) {
                    const int fn = std::min(step, n - f0);
                    on_event(user, b, MIS_EVENT_AUDIO, host + (size_t)b * longest + (size_t)f0 * up, (int64_t)fn * up);
                }
            }
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
