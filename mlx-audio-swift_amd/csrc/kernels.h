// kernels.h - launchers shared between translation units of libmi_speech.
#pragma once
#include "common.h"
#include <functional>

// orpheus_codes.hip
void launch_orpheus_deinterleave(const int32_t* codes7, int in_stride, int batch, int groups, int32_t* l0,
                                 int32_t* l1, int32_t* l2, int out_groups, hipStream_t s);
void launch_orpheus_deinterleave_ragged(const int32_t* codes7, int in_stride, const int32_t* n_codes, int batch,
                                        int32_t* l0, int32_t* l1, int32_t* l2, int out_groups, hipStream_t s);
struct SpeechTokenIds { int start_of_speech, end_of_speech, audio_offset, start_of_ai; };   // nullptr = Orpheus (LlamaTTS.swift:20-30)
void launch_orpheus_parse_output(const int32_t* ids, const int32_t* lens, int batch, int stride, int32_t* codes_out,
                                 int32_t* n_codes_out, hipStream_t s, const SpeechTokenIds* tk = nullptr);

// snac.hip : device-pointer decode used by the TTS engine (codes already on device, ragged rows padded)
struct mis_snac;
// noise_dev: explicit per-block noise, or null -> rng_enabled ? internal N(0,1) keyed by (rng_seed, block,
// row_offset + row_ids[b] (or b), t) : nothing added.
void snac_decode_device(mis_snac* c, const int32_t* const* codes_dev, int batch, int t_coarse,
                        const float* const* noise_dev, int rng_enabled, uint64_t rng_seed, const int32_t* row_ids,
                        int64_t row_offset, float* pcm_dev, int64_t pcm_stride, hipStream_t s);
hipStream_t snac_stream(mis_snac* c);
int snac_device(const mis_snac* c);
const mis_snac_config* snac_config(const mis_snac* c);

// mel.hip
void whisper_features_device(int device, const float* pcm_dev, int batch, int n_mels, float* out_dev, hipStream_t s);
void mel_spectrogram_device(int device, const mis_mel_config& c, const float* pcm_dev, int batch, int64_t n_samples, float* out_dev, hipStream_t s);

// lm_engine.hip hooks used by soprano.hip
struct mis_tts;
hipStream_t tts_stream(mis_tts* c);
int tts_hidden_size(const mis_tts* c);
int tts_device(const mis_tts* c);
// generate collecting model.norm(h) per step (row b: n_hidden[b] states, first = last prompt token), stop token ends a row
void tts_generate_hidden(mis_tts* c, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch, const mis_gen_params* gp,
                         int stop_id, DevBuf<float>& hidden, std::vector<int32_t>& n_hidden, std::vector<int32_t>& n_tokens,
                         std::vector<int32_t>& tokens, int64_t& tokens_stride, mis_event_cb cb = nullptr, void* user = nullptr,
                         const volatile int* cancel = nullptr);
double tts_last_decode_ms(const mis_tts* c);

// hooks for composite engines built on the LM step chain (qwen3tts.hip); defined in lm_engine.hip
struct TtsView { bf16_t *x, *h, *logits, *emb; int32_t *ids, *pos_next; uint8_t* active; int d, Mpad, V, Vpad, batch, finalized, L; hipStream_t stream; };
void tts_internal_reset(mis_tts* c, int batch, int max_context);
void tts_internal_use_stream(mis_tts* c, hipStream_t s);
bool tts_internal_prefill_rows_ok(const mis_tts* c, int Lmax);
void tts_internal_prefill_rows(mis_tts* c, const bf16_t* rows /*[Lmax][Mpad][d] device*/, const int32_t* lens_host, int Lmax);
// one token position per active row: embedding rows gathered from `table` ([table_rows][d] bf16) by `ids` (device), through all
// layers and the final norm (result: packed x of the view).  table == nullptr: the model's own embedding / id buffer.
void tts_internal_enqueue_layers(mis_tts* c, const bf16_t* table, int table_rows, const int32_t* ids);
void tts_internal_enqueue_head(mis_tts* c, const bf16_t* head_packed);      // logits of the view; nullptr = own lm_head
TtsView tts_internal_view(mis_tts* c);
// the packed weights of a finalized handle, for engines that stream them in their own kernels (token_engine.hip)
struct TtsWeightsView {
    const bf16_t *emb, *wqkv, *wo, *wgu, *wdown, *head, *norms, *qknorm;
    int d, L, ff, H, Hkv, D, V, Vpad, Nqkv, device, finalized, qk_norm, rope_plain, quantised;
    float eps;
    hipStream_t stream;
};
TtsWeightsView tts_internal_weights(mis_tts* c);
// RoPE cos / sin tables [positions][D/2] for at least `max_context` positions (re-initialises the handle's per-batch state for one row)
void tts_internal_rope_tables(mis_tts* c, int max_context, const float** cos_out, const float** sin_out);
// K/V of one row after the launch chain's prefill, in its tiled cache layouts (lm_kernels.hip, k_attn_decode): element (pos, d) of
//   K: ((((pos >> 5) * 2 + ((pos & 31) >> 2 & 1)) * (D / 32) + (d >> 5)) * 64 + (((d & 31) >> 3) << 4) + ((((pos & 31) >> 3) << 2) | (pos & 3))) * 8 + (d & 7)
//   V: ((((pos >> 5) * (D / 16) + (d >> 4)) * 64 + (((pos & 31) >> 3) << 4) + (d & 15)) * 8 + (pos & 7)
// per (layer, kv head); layer li starts at li * layer_stride
struct TtsKvView { const bf16_t *kcache, *vtcache; const float *rope_cos, *rope_sin; int Smax, Hkv, D; size_t layer_stride; };
TtsKvView tts_internal_prefill_kv(mis_tts* c, const int32_t* prompt_host, int n, int max_context);
// batch-1 decode engine: one persistent launch per request on the compute units of `xcds` XCDs (token_engine.hip)
struct TokenEngineScratch;             // per-handle buffers of the engine (K/V copy, exchange buffers, host-visible token row), reused by every request
TokenEngineScratch* token_engine_scratch_create();
void token_engine_scratch_destroy(TokenEngineScratch*);
struct TokenEngineRequest {
    const int32_t* prompt = nullptr;   // host or device
    int n_prompt = 0, max_new = 0, xcds = 2;
    bool generate = false;             // false: laboratory form (arg-max after every position); true: generate semantics (see token_engine_run)
    bool sample = false;               // generate only: mis-sampler-v1 behind the Soprano flavour's penalty; false = arg-max
    float temperature = 0.f, penalty = 0.f;
    int win_cap = 0;
    uint64_t seed = 0;
    int64_t row = 0;
    int stop_id = -1;
    bool want_logits = false, want_hidden = false;
    bool prefill_by_chain = false;     // generate only: the prompt but its last position through the launch chain's batched prefill, K/V imported
    float* hidden_dev = nullptr;       // device rows [positions from the last prompt token on][hidden] written in place (else returned in `hidden`)
    TokenEngineScratch* scratch = nullptr;   // null: buffers of this call only
    // stream form (generate only): called on the CALLING thread, in order, for the k-th chosen id WHILE the launch runs (the ids land in
    // host-visible memory one system-scope store each); cancel: polled by the host beside the ids, forwarded to the launch, which ends at
    // its next position - token_engine_run then throws MIS_ERR_CANCELLED
    std::function<void(int k, int32_t id)> on_token;
    const volatile int* cancel = nullptr;
    int spin = 0;                      // polls per granule before an edge counts as timed out (0: the default, ~1 s; MIS_TE_SPIN overrides - tests force a time-out with 0)
};
struct TokenEngineResult {
    std::vector<int32_t> next_tokens;  // [n_prompt + max_new]: the id chosen after position t (positions without an output projection: 0)
    std::vector<float> logits, hidden;
    int n_positions = 0, n_sampled = 0, head_from = 0;
    int n_announced = 0;               // ids handed to on_token (also on the failure paths: what the caller has already seen)
    double ms = 0;
};
// shape AND device: the widths the engine is compiled for, on a device of 8 XCDs x 32 compute units (block b -> XCD b mod 8)
bool token_engine_supports(mis_tts* lm);
// throws MIS_ERR_GENERATION_FAILED when an edge timed out (workers not co-resident): nothing of the request is valid, the caller runs it
// on the launch chain (if result.n_announced == 0) - and MIS_ERR_CANCELLED when rq.cancel was raised
void token_engine_run(mis_tts* lm, const TokenEngineRequest& rq, TokenEngineResult& out);
// a handle whose device also runs ANOTHER replica's streams (logical shards of a group on one GPU) must not launch kernels whose blocks
// wait for each other to be co-resident (the one-launch sampler): set by the group entry points in group.hip
void tts_internal_set_shared_device(mis_tts* c, bool shared);
bool tts_internal_shared_device(const mis_tts* c);
mis_tts* soprano_internal_lm(mis_soprano* c);
void soprano_internal_set_group_batch(mis_soprano* c, int batch);   // group entry points: the request's batch over all shards (0 = leave)
void tts_internal_set_decode_ms(mis_tts* c, double ms);
int soprano_internal_device(const mis_soprano* c);
int whisper_internal_device(const mis_whisper* c);
void whisper_internal_set_shared_device(mis_whisper* c, bool shared);

// Qwen3-TTS speech-tokenizer decoder (q3_codec.hip)
struct mis_q3dec;
mis_status mis_q3dec_create(const mis_qwen3tts_config* cfg, int device, mis_q3dec** out);
mis_status mis_q3dec_set_tensor(mis_q3dec*, const char* name, const void* data, mis_dtype dtype, const int64_t* shape, int ndim);
mis_status mis_q3dec_finalize(mis_q3dec*);
void mis_q3dec_destroy(mis_q3dec*);
int q3dec_total_upsample(const mis_q3dec*);
void q3dec_decode_device(mis_q3dec*, const int32_t* codes_dev /*[B][nq][T]*/, int batch, int T, float* wav_dev, int64_t wav_stride, hipStream_t s);
void q3dec_decode_host(mis_q3dec*, const int32_t* codes, int batch, int T, float* out, int stop_after, int* outC, int64_t* outT, hipStream_t s);
void q3dec_decode_strided(mis_q3dec*, const int32_t* codes_dev, int64_t cs_b, int64_t cs_q, int64_t cs_t, int batch, int T, float* wav_dev,
                          int64_t wav_stride, hipStream_t s);
// streaming session (resetStreamingState / streamingStep): code (b, q, t) of a step at codes_dev[b*cs_b + q*cs_q + t*cs_t]
void q3dec_stream_begin(mis_q3dec*, int batch, int cap_frames, int chunk_cap, bool dup_bias, hipStream_t s);
void q3dec_stream_step(mis_q3dec*, const int32_t* codes_dev, int64_t cs_b, int64_t cs_q, int64_t cs_t, int Tn, float* wav_dev,
                       int64_t wav_stride, hipStream_t s);
void q3dec_stream_step_host(mis_q3dec*, const int32_t* codes /*[B][nq][Tn]*/, int Tn, float* out, hipStream_t s);
void q3dec_stream_end(mis_q3dec*);
int q3dec_stream_pos(const mis_q3dec*);
