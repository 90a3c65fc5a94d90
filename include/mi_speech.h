/*
 * mi_speech.h - C ABI of libmi_speech.so: MI355X-native (gfx950) kernels for the
 * TTS generate()/generateStream() + neural-codec hot path of Blaizzy/mlx-audio-swift.
 *
 * The reference has NO FFI: its boundary is Swift protocol conformance
 *   SpeechGenerationModel   Sources/MLXAudioTTS/Generation.swift:8-39
 *   AudioCodecModel         Sources/MLXAudioCodecs/AudioCodecModel.swift:4-27
 * implemented per model by classes whose arithmetic is MLX (un-vendored mlx-swift 0.31.4).
 * This header is what a thin Swift class conforming to those protocols binds instead of MLX
 * (see INTEGRATION.md for the Swift shim); every entry point cites the reference symbol it
 * replaces.  Plain C types only: pointers + sizes, no torch / MLX types.
 *
 * Pointer convention: every data pointer may be HOST or DEVICE memory of the handle's GPU
 * (copies use hipMemcpyDefault; unified addressing tells them apart).  Inputs are borrowed for
 * the duration of the call.  Library-allocated outputs are pinned host memory released with
 * mis_free().  No entry point aborts: failures return a status and set mis_last_error().
 * Threading: one in-flight call per handle; distinct handles are independent.
 */
#ifndef MI_SPEECH_H
#define MI_SPEECH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIS_ABI_VERSION 1

/* <-> AudioGenerationError, Sources/MLXAudioCore/Generation/GenerationTypes.swift:66-87 */
typedef enum {
    MIS_OK = 0,
    MIS_ERR_NOT_INITIALIZED = 1,   /* .modelNotInitialized */
    MIS_ERR_GENERATION_FAILED = 2, /* .generationFailed    */
    MIS_ERR_INVALID_INPUT = 3,     /* .invalidInput        */
    MIS_ERR_AUDIO_DECODE = 4,      /* .audioDecodingFailed */
    MIS_ERR_AUDIO_ENCODE = 5,      /* .audioEncodingFailed */
    MIS_ERR_CANCELLED = 6,         /* Task cancellation, LlamaTTS.swift:715,911 */
    MIS_ERR_DEVICE = 7             /* HIP runtime failure / no GPU */
} mis_status;

typedef enum { MIS_F32 = 0, MIS_F16 = 1, MIS_BF16 = 2, MIS_I32 = 3 } mis_dtype;

const char* mis_last_error(void);          /* thread-local message of the last failing call */
int         mis_abi_version(void);
void        mis_free(void* p);             /* releases library-allocated (pinned host) outputs */
int         mis_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Orpheus <-> SNAC token framing (integer-exact).
 * ---------------------------------------------------------------------------------------- */

/* llamaDecodeAudioFromCodes, LlamaTTS.swift:41-64: split 7-token frames into the three SNAC
 * levels.  codes7: [batch, 7*groups] values in [0, 7*4096); l0 [batch,groups], l1 [batch,2*groups],
 * l2 [batch,4*groups], values in [0,4096).  Runs on the GPU `device`. */
mis_status mis_orpheus_deinterleave(int device, const int32_t* codes7, int batch, int groups,
                                    int32_t* l0, int32_t* l1, int32_t* l2);

/* LlamaTTSModel.parseOutput, LlamaTTS.swift:383-434, per row (App. D.2 of SURVEY.md): crop after
 * the last start-of-speech (128257), drop end-of-speech (128258), trim to a multiple of 7, subtract
 * 128266.  ids: [batch, stride] with lens[batch] valid entries per row; codes_out: [batch, stride];
 * n_codes_out: [batch]. */
mis_status mis_speech_parse_output(int device, const int32_t* ids, const int32_t* lens, int batch, int stride, int32_t* codes_out,
                                   int32_t* n_codes_out, int start_of_speech, int end_of_speech, int audio_token_offset,
                                   int start_of_ai);     /* explicit ids (VyvoTTS); start_of_ai < 0: no fallback */
mis_status mis_orpheus_parse_output(int device, const int32_t* ids, const int32_t* lens, int batch,
                                    int stride, int32_t* codes_out, int32_t* n_codes_out);

/* ------------------------------------------------------------------------------------------
 * SNAC codec (decode path).   Replaces class SNAC, Sources/MLXAudioCodecs/SNAC/SNACDecoder.swift.
 * ---------------------------------------------------------------------------------------- */
typedef struct mis_snac mis_snac;

/* SNACConfig, SNAC/Config.swift:10-37 (decode-relevant fields) */
typedef struct {
    int32_t sampling_rate;
    int32_t latent_dim;        /* = encoder_dim * 2^len(encoder_rates), SNACDecoder.swift:50 */
    int32_t decoder_dim;
    int32_t n_decoder_rates;   /* <= 8 */
    int32_t decoder_rates[8];
    int32_t codebook_size;
    int32_t codebook_dim;
    int32_t n_codebooks;       /* <= 8 */
    int32_t vq_strides[8];
    int32_t noise;             /* bool */
    int32_t depthwise;         /* bool; only 1 is supported (24 kHz model) */
    int32_t attn_window_size;  /* 0 = none (24 kHz model); 32: LocalMHA of the 32 / 44 kHz models (SNAC/Attention.swift; <= 64) */
} mis_snac_config;

/* SNAC.fromModelDirectory, SNACDecoder.swift:156-189: config.json + model.safetensors */
mis_status mis_snac_load(const char* model_dir, int device, mis_snac** out);
/* programmatic construction (tests / hosts that already hold the arrays) */
mis_status mis_snac_create(const mis_snac_config* cfg, int device, mis_snac** out);
/* one tensor in the reference's key layout, e.g.
 * "decoder.model.layers.1.weight_v"; dtype F32/F16/BF16; shape as stored. */
mis_status mis_snac_set_tensor(mis_snac*, const char* name, const void* data, mis_dtype dtype,
                               const int64_t* shape, int ndim);
/* update(parameters:verify:.all), SNACDecoder.swift:185: every decoder/quantizer key must be
 * present; folds weight-norm (Layers.swift:35-42,102-103,166) and builds device tables. */
mis_status mis_snac_finalize(mis_snac*);
void       mis_snac_destroy(mis_snac*);
/* number of samples decode() produces for t_coarse entries of codes[0] */
int64_t    mis_snac_num_samples(const mis_snac*, int t_coarse);
/* length of NoiseBlock i's input (i < n_decoder_rates) for t_coarse */
int64_t    mis_snac_noise_len(const mis_snac*, int block, int t_coarse);

/* NoiseBlock policy when a decode is given noise == NULL: null_noise_is_zero = 0 (default, the
 * reference's behaviour: x + N(0,1)[B,1,T] * conv(x), Layers.swift:270-278) draws the noise on the
 * device from the documented counter generator keyed by (seed, block, global row, t);
 * null_noise_is_zero = 1 adds nothing (deterministic decode). */
mis_status mis_snac_set_noise(mis_snac*, int null_noise_is_zero, uint64_t seed);

/* SNAC.decode / decodeAudio, SNACDecoder.swift:127-131,201-203 (fromCodes VQ.swift:165-191 +
 * Decoder Layers.swift:364-421).  codes[i]: int32 [batch, t_coarse * vq_strides[0]/vq_strides[i]];
 * noise: NULL (see mis_snac_set_noise) or n_decoder_rates pointers to f32 [batch, noise_len(i)]
 * standing in for MLXRandom.normal (Layers.swift:274); pcm_out: f32 [batch, num_samples]. */
mis_status mis_snac_decode(mis_snac*, const int32_t* const* codes, int batch, int t_coarse,
                           const float* const* noise, float* pcm_out);
/* debug/parity taps: copy an intermediate ("zq","stem_dw","stem_pw","block0".."blockN") of the
 * LAST decode into out (f32 [batch, C, T]); returns its C and T. */
/* encode path (SNAC.encode, SNACDecoder.swift:86-125: preprocess right-pad -> Encoder (Layers.swift:319-360) -> residual VQ with
 * nearest normalised code, VQ.swift:47-163).  Available when the checkpoint's "encoder.*" tensors were loaded (depthwise
 * configs without LocalMHA, i.e. snac_24khz).  padded_length = n rounded up to hop * lcm(vq_strides). */
int64_t    mis_snac_padded_length(const mis_snac*, int64_t n_samples);
/* audio f32 [batch, n_samples]; codes_out[i] int32 [batch, padded / hop / vq_strides[i]]; z_out (nullable) f32
 * [batch, latent_dim, padded / hop] = encoder output before quantisation */
mis_status mis_snac_encode(mis_snac*, const float* audio, int batch, int64_t n_samples, int32_t* const* codes_out, float* z_out);
mis_status mis_snac_debug_tap(mis_snac*, const char* name, float* out, int64_t capacity,
                              int32_t* channels, int64_t* length);

/* ------------------------------------------------------------------------------------------
 * Orpheus / Llama token LM + TTS generate.  Replaces LlamaTTSModel, LlamaTTS.swift:354-977.
 * ---------------------------------------------------------------------------------------- */
typedef struct mis_tts mis_tts;

/* LlamaTTSConfiguration, LlamaTTSConfig.swift:15-61 */
typedef struct {
    int32_t hidden_size, num_hidden_layers, intermediate_size;
    int32_t num_attention_heads, num_key_value_heads, head_dim;
    int32_t vocab_size;
    float   rms_norm_eps;
    float   rope_theta;
    /* llama3 rope scaling, LlamaTTS.swift:111-186 (defaults 32 / 1 / 4 / 8192) */
    float   rope_factor, rope_low_freq_factor, rope_high_freq_factor, rope_original_max_pos;
    int32_t tie_word_embeddings;
    int32_t sample_rate;
    /* Qwen3-style variants (Soprano Soprano.swift:24-97, VyvoTTS Qwen3.swift:204-205): per-head RMSNorm of q and k
     * before RoPE (weights model.layers.N.self_attn.{q,k}_norm.weight [head_dim]) and plain RoPE(base) without the
     * llama3 rescale.  Both 0 for Orpheus (LlamaTTS.swift always builds Llama3ScaledRoPE, :161-186). */
    int32_t qk_norm;
    int32_t rope_plain;
    /* Qwen3-TTS talker / code predictor write the rotation as array ops in the model dtype (cos/sin cast to bf16,
     * T(T(x*cos) + T(rotate_half(x)*sin)), Qwen3TTSTalker.swift:15-24,92-95) instead of MLXFast.RoPE */
    int32_t rope_ops_in_dtype;
    /* speech token ids of the generate loop (0 = Orpheus: 128257 / 128258 / 128266, LlamaTTS.swift:20-30).  VyvoTTS
     * (Qwen3.swift:19-29): start_of_speech 151670, end_of_speech 151671, audio_token_offset 151679, start_of_ai 151674
     * (parse fallback, :332-358; 0 = none) */
    int32_t start_of_speech_id, end_of_speech_id, audio_token_offset, start_of_ai_id;
    /* SNAC decode granularity of generate(): 0 = the whole utterance in one decode (LlamaTTS.swift:759).  VyvoTTS decodes
     * INDEPENDENT chunks of 50 code groups and concatenates the samples (decodeAudioFromCodes, Qwen3.swift:47-83; nothing is
     * carried across a chunk boundary, so the waveform differs from a single decode near every boundary) */
    int32_t codec_chunk_groups;
} mis_lm_config;

/* GenerateParameters as used by LlamaTTS.swift:573-581,691-696 (mlx-swift-lm) */
typedef struct {
    int32_t  max_tokens;           /* default 1200 */
    float    temperature;          /* 0 => greedy argmax */
    float    top_p;                /* outside (0,1) => no nucleus cut */
    float    repetition_penalty;   /* <= 0 or 1 => off */
    int32_t  repetition_context;   /* default 20 */
    uint64_t seed;                 /* mis-sampler-v1 RNG key (see csrc/lm_sampler.hip) */
    int32_t  frame_constrained;    /* 0 normal. 1 (synthetic-weight benches): step i may only emit
                                      128266 + (i%7)*4096 + [0,4096); EOS can then never be sampled; the
                                      sampler only visits that range (k_samp_narrow).  2: the same
                                      constraint through the FULL-vocabulary sampler (every id visited,
                                      masked ones get zero mass) - the code path of an unconstrained
                                      checkpoint, with tokens that still form valid frames */
    int64_t  row_offset;           /* global index of row 0 (RNG keyed by global row => sharding-invariant) */
    int32_t  sampler_flavor;       /* 0 mlx-lm processor/sampler (Orpheus).  1 Soprano streamGenerate (Soprano.swift:801-901):
                                      penalty window = generated tokens only, applied per occurrence in float32; the reference's
                                      TopPSampler thresholds UNNORMALISED exp(logit) sums (:1003-1036) and so only ever drops
                                      tokens of vanishing mass - this flavour keeps them (top_p is ignored) */
    int32_t  reserved;
} mis_gen_params;

/* AudioGeneration events, GenerationTypes.swift:50-61 */
typedef enum { MIS_EVENT_TOKEN = 0, MIS_EVENT_INFO = 1, MIS_EVENT_AUDIO = 2 } mis_event_kind;
/* AudioGenerationInfo, GenerationTypes.swift:14-45 */
typedef struct {
    int32_t prompt_token_count, generation_token_count;
    double  prefill_time, generate_time, tokens_per_second, peak_memory_gb;
} mis_gen_info;
/* payload: TOKEN -> const int32_t* (n=1); INFO -> const mis_gen_info* (n=1); AUDIO -> const float* (n samples) */
typedef void (*mis_event_cb)(void* user, int row, mis_event_kind kind, const void* payload, int64_t n);

/* LlamaTTSModel.fromModelDirectory, LlamaTTS.swift:942-977: config.json + *.safetensors of the LM;
 * codec is borrowed (post_load_hook loads SNAC separately, :595-602). */
mis_status mis_tts_load(const char* model_dir, mis_snac* codec, int device, mis_tts** out);
mis_status mis_tts_create(const mis_lm_config* cfg, mis_snac* codec /* may be NULL */, int device, mis_tts** out);
/* HF/MLX key layout: model.embed_tokens.weight, model.layers.N.*, model.norm.weight, [lm_head.weight] */
mis_status mis_tts_set_tensor(mis_tts*, const char* name, const void* data, mis_dtype dtype,
                              const int64_t* shape, int ndim);
/* fills every weight on the device with the documented generator "mis-synth-v1"
 * (oracle/synth.py has the same formula); benches only - there are no checkpoints offline. */
/* Linear / Embedding in MLX's affine-quantised form (mlx quantize [3P]): wq uint32 [N, K*bits/32] (element i of a row =
 * (word[i / (32/bits)] >> (bits * (i % (32/bits)))) & mask), scales / biases [N, K/group_size] of dtype sb_dtype;
 * w = scale * q + bias.  The reference keeps QuantizedLinear (LlamaTTS.swift:958-968, Qwen3TTS.swift:1157-1170) and never
 * materialises the weight: quantizedMatmul applies scale and bias to float32 group sums.  So does the engine for a role
 * (q|k|v, o_proj, gate|up, down_proj, lm_head) whose matrices ALL arrive with 8 or 4 bits, group size 64 and bf16 scales: the
 * codes are streamed as stored (0.53x / 0.28x of the bf16 bytes) and dequantised in registers (csrc/lm_qgemm.hip).  Any other
 * combination (2 bit, other groups, f16 / f32 scales, per-layer overrides inside a role, MIS_QUANT_NATIVE=0) is dequantised once at
 * load into the bf16 layout, which rounds s*q+b to bf16 per weight - a stated deviation for those cases only.  The embedding
 * (QuantizedEmbedding: a gather of dequantised rows in the model dtype) is always dequantised at load. */
mis_status mis_tts_set_tensor_quantized(mis_tts*, const char* name, const uint32_t* wq, const void* scales, const void* biases,
                                        mis_dtype sb_dtype, int64_t N, int64_t K, int group_size, int bits);
/* after finalize: bits of the quantised form a role is streamed in (0 = dense bf16); role 0 q|k|v, 1 o_proj, 2 gate|up, 3 down, 4 lm_head */
int        mis_tts_native_quant_bits(const mis_tts*, int role);
/* benches: every Linear as a synthetic MLX-quantised matrix (bits 8 or 4, group 64, bf16 scales) - no checkpoints offline */
mis_status mis_tts_init_synthetic_quantized(mis_tts*, uint64_t seed, int bits);
mis_status mis_tts_init_synthetic(mis_tts*, uint64_t seed);
mis_status mis_tts_finalize(mis_tts*);     /* verify all keys present; pack weights for MFMA streaming */
void       mis_tts_destroy(mis_tts*);

/* LM-only taps (parity tests, teacher forcing).  reset: new batch, empty KV caches (makeCache,
 * LlamaTTS.swift:604-608).  forward: ONE token per active row through the model
 * (LlamaTTSModel.callAsFunction :557-567 with L=1 and cache), logits_out f32 [batch, vocab]
 * (bf16 values, widened) or NULL. */
mis_status mis_lm_reset(mis_tts*, int batch, int max_context);
/* the prefill call of the reference loop, `model(inputIds, cache:)` (LlamaTTS.swift:711): ragged prompts (flat ids + lens) through
 * the model with empty caches as ONE [positions x rows] pass of MFMA GEMMs (csrc/lm_prefill.hip) - position by position only for
 * natively quantised roles or MIS_PREFILL_SEQ=1.  logits_out f32 [batch, vocab] (nullable): the next-token logits of every row;
 * hidden_out f32 [batch, hidden] (nullable): model.norm(h) of every row's last prompt token (Soprano's first decoder input,
 * Soprano.swift:824-825).  The caches then hold the prompts; mis_lm_forward continues behind them.  mis_tts_generate* prefill the
 * same way. */
mis_status mis_lm_prefill(mis_tts*, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch, int max_context,
                          float* logits_out, float* hidden_out);
mis_status mis_lm_forward(mis_tts*, const int32_t* ids, const uint8_t* active, float* logits_out);
/* same, also returning model.norm(h) of the fed token: hidden_out f32 [batch, hidden_size] (Soprano decodes these,
 * Soprano.swift:254-275); either output may be NULL */
mis_status mis_lm_forward_hidden(mis_tts*, const int32_t* ids, const uint8_t* active, float* logits_out, float* hidden_out);
/* processor + sampler of the generate loop (LlamaTTS.swift:717-721) on caller-provided logits:
 * logits f32 [batch, vocab]; window [batch, ctx] (ids, right-aligned valid part = window_len[b]);
 * lo/hi: optional allowed id range (hi<=0 => vocab); tokens_out [batch].  mis-sampler-v1.
 * Wide ranges run in ONE launch whose 8 blocks per row wait for each other (bounded spins); if a row's blocks were not all resident
 * (another stream holding CUs) the time-out is detected after the launch and the call falls back to the multi-launch path on fresh
 * inputs - the tokens are the same.  Inside mis_*_generate* (a captured graph) the same condition is reported as
 * MIS_ERR_GENERATION_FAILED after the decode loop; MIS_SAMPLER_WIDE=1 selects the multi-launch path for such deployments. */
mis_status mis_sample_logits(int device, const float* logits, int batch, int vocab,
                             const int32_t* window, const int32_t* window_len, int ctx,
                             const mis_gen_params* params, int step, int lo, int hi,
                             int32_t* tokens_out);

/* generate(text:voice:...) LlamaTTS.swift:658-765 for a BATCH of already-tokenised prompts
 * (tokenisation stays on the host side: prepareInputIds :446-553 / swift-transformers).
 * prompt_ids: concatenated rows; prompt_lens[batch].  snac_noise: NULL (N(0,1) drawn on the device,
 * keyed by params->seed and the global row) or explicit noise as in mis_snac_decode (then every row
 * must produce the same number of frames).  Outputs (library-allocated, mis_free):
 * *pcm_out f32 [batch, *pcm_stride] with pcm_lens[batch] valid samples per row;
 * *tokens_out (optional, may be NULL) int32 [batch, *tokens_stride] generated ids (EOS included
 * when sampled, as in the .token stream :862-866), n_tokens[batch]. */
mis_status mis_tts_generate(mis_tts*, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                            const mis_gen_params* params, const float* const* snac_noise,
                            float** pcm_out, int64_t* pcm_stride, int64_t* pcm_lens,
                            int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens);
/* Same, leaving PCM resident in HBM: pcm_dev f32 [batch, pcm_stride] caller-allocated DEVICE memory
 * (pcm_stride >= num_samples for max_tokens/7 groups).  What bench.py times. */
mis_status mis_tts_generate_device(mis_tts*, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                   const mis_gen_params* params, const float* const* snac_noise,
                                   float* pcm_dev, int64_t pcm_stride, int64_t* pcm_lens, int32_t* n_tokens);
/* generateStream(...) LlamaTTS.swift:777-913: .token per step and row, then per row .info and ONE
 * final .audio (Orpheus does not stream audio chunks, :893-904).  cancel_flag polled per step. */
mis_status mis_tts_generate_stream(mis_tts*, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                   const mis_gen_params* params, const float* const* snac_noise,
                                   mis_event_cb on_event, void* user, const volatile int* cancel_flag);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU: utterance-batch data parallelism behind the boundary (the reference is single-device and has
 * no counterpart).  Rows are independent units; every GPU holds a full replica of the weights (the host loads one mis_tts per
 * device, exactly as it loads one); a batch is cut into contiguous row blocks (mis_shard_rows); the sampler / codec-noise RNG is
 * keyed by the GLOBAL row index (row_offset), so every row's tokens and samples are independent of the number of shards; the only
 * exchange is ONE all-gather of the decoded PCM (+ lengths) at the end.
 * ---------------------------------------------------------------------------------------- */
/* contiguous block [*lo, *hi) of `rank` among `world` shards; block sizes differ by at most one row */
void mis_shard_rows(int n_rows, int rank, int world, int* lo, int* hi);

/* (1) SINGLE PROCESS, N devices (a Swift host): a group of replicas - one finalized mis_tts (with its own codec) per device;
 * devices may repeat (logical shards of one GPU).  The group borrows the handles. */
typedef struct mis_group mis_group;
typedef struct { int32_t n_shards; double generate_ms, slowest_shard_ms, gather_ms; } mis_group_timing;
mis_status mis_tts_group_create(mis_tts* const* replicas, int n, mis_group** out);
void       mis_tts_group_destroy(mis_group*);
int        mis_tts_group_size(const mis_group*);
/* mis_tts_generate over the group: one worker thread + stream per replica, each generating its block with row_offset advanced by
 * the block start; outputs as mis_tts_generate (rows in batch order, one pinned host buffer) - identical, bit for bit, to the
 * single-handle call. */
mis_status mis_tts_group_generate(mis_group*, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                  const mis_gen_params* params, float** pcm_out, int64_t* pcm_stride, int64_t* pcm_lens,
                                  int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens);
/* PCM left in HBM and all-gathered: pcm_dev[i] = replica i's DEVICE buffer [batch, pcm_stride] on its own GPU; on return every
 * buffer holds every row (each replica writes its block into every peer's buffer over xGMI: direct peer copies). */
mis_status mis_tts_group_generate_device(mis_group*, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                         const mis_gen_params* params, float* const* pcm_dev, int64_t pcm_stride,
                                         int64_t* pcm_lens, int32_t* n_tokens);
mis_status mis_tts_group_last_timing(mis_group*, mis_group_timing* out);

/* (2) ONE PROCESS PER GPU (any launcher; what bench.py --gpus N runs): an RCCL communicator behind the ABI.  Rank 0 obtains a
 * 128-byte id (ncclUniqueId), the host distributes it to the other ranks by whatever means it has, every rank creates its
 * communicator, generates its block with mis_tts_generate_device(row_offset = block start) and calls the all-gather:
 * pcm_local_dev [rows_local, stride] -> pcm_all_dev [world * rows_local, stride] (device), lens (host) likewise; RCCL
 * ncclAllGather over xGMI on the communicator's stream; *gather_ms (nullable) = device time of the exchange. */
typedef struct mis_comm mis_comm;
#define MIS_COMM_ID_BYTES 128
mis_status mis_comm_unique_id(void* id_out /* MIS_COMM_ID_BYTES */);
mis_status mis_comm_create(int device, int rank, int world, const void* unique_id, mis_comm** out);
void       mis_comm_destroy(mis_comm*);
mis_status mis_comm_all_gather_pcm(mis_comm*, const float* pcm_local_dev, const int64_t* lens_local, int rows_local,
                                   int64_t stride, float* pcm_all_dev, int64_t* lens_all, double* gather_ms);

/* per-phase device timings of the last generate (ms): [0]=prefill [1]=decode loop [2]=parse+codec,
 * plus average duration (ms) and launch count of the dominant GEMM kernel measured with HIP events
 * on the library's own stream when profiling is enabled with mis_tts_set_profiling(ctx, 1). */
typedef struct {
    double prefill_ms, decode_ms, codec_ms;
    double step_ms_avg;            /* decode_ms / steps */
    int32_t steps;
    double gemm_probe_ms;          /* avg duration of ONE lm_head GEMM launch (HIP events), 0 if off */
    double gemm_probe_bytes;       /* algorithmic bytes of that launch */
    double hbm_bytes_per_step;     /* algorithmic bytes per decode step: weights + KV read */
} mis_tts_timing;
mis_status mis_tts_set_profiling(mis_tts*, int enabled);
mis_status mis_tts_last_timing(mis_tts*, mis_tts_timing* out);
/* time `iters` launches of one kernel of the decode step in isolation (HIP events on the library stream), rotating over the
 * layers so the Infinity Cache cannot serve the operands: which = 0 qkv, 1 o_proj, 2 gate_up, 3 down, 4 lm_head (weight-streaming
 * GEMMs), 5 decode attention at context 368 (the C3 mean; bytes = K/V rows read), 6 the slab-reduce + residual + RMSNorm kernel.
 * avg_ms and the algorithmic bytes of one launch are returned.  Used by bench.py for the roofline object. */
mis_status mis_tts_time_gemm(mis_tts*, int which, int batch, int iters, double* avg_ms, double* bytes);

/* ------------------------------------------------------------------------------------------
 * Soprano TTS.  Replaces SopranoModel / SopranoDecoder (Sources/MLXAudioTTS/Models/Soprano/Soprano.swift:201-690,
 * SopranoDecoder.swift:225-284, Vocos backbone Sources/MLXAudioCodecs/Vocos/VocosBackbone.swift:109-204).
 * Text splitting / cleaning / tokenisation stay on the host (TextUtils.swift).
 * ---------------------------------------------------------------------------------------- */
typedef struct mis_soprano mis_soprano;
/* SopranoConfiguration, SopranoConfig.swift:103-167 */
typedef struct {
    mis_lm_config lm;              /* qk_norm / rope_plain are forced on (SopranoAttention) */
    int32_t decoder_num_layers, decoder_dim, decoder_intermediate_dim;
    int32_t hop_length, n_fft, upscale, input_kernel, dw_kernel, token_size;
    int32_t stop_token_id;         /* "[STOP]" (Soprano.swift:855) */
} mis_soprano_config;
mis_status mis_soprano_create(const mis_soprano_config*, int device, mis_soprano** out);
/* checkpoint keys as stored; SopranoModel.sanitize (Soprano.swift:314-361) is applied: "decoder.*" -> float32 decoder
 * weights, everything else -> the token LM (model.* / lm_head.weight) */
mis_status mis_soprano_set_tensor(mis_soprano*, const char* name, const void* data, mis_dtype dtype,
                                  const int64_t* shape, int ndim);
mis_status mis_soprano_finalize(mis_soprano*);
void       mis_soprano_destroy(mis_soprano*);
mis_tts*   mis_soprano_lm(mis_soprano*);                 /* borrowed handle of the token LM (taps, synthetic init) */
int64_t    mis_soprano_num_samples(const mis_soprano*, int n_hidden);   /* (upscale*(n-1)) * hop_length */
/* SopranoDecoder.callAsFunction: hidden f32 [batch, L, hidden_size] -> audio f32 [batch, num_samples(L)] */
mis_status mis_soprano_decode(mis_soprano*, const float* hidden, int batch, int L, float* audio_out);
/* generate for a batch of tokenised sentences: outputs as mis_tts_generate (pcm rows padded to the longest) */
mis_status mis_soprano_generate(mis_soprano*, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                const mis_gen_params* params, float** pcm_out, int64_t* pcm_stride, int64_t* pcm_lens,
                                int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens);
/* Which program ran the LM loop of the handle's LAST generate / generateStream call: 0 = the launch chain, chosen by rule; 1 = the batch-1
 * token engine (one persistent launch; chosen by rule: a ONE-row request - over all shards of a group call - on a bf16 checkpoint of
 * Soprano-80M's widths, device of 8 XCDs x 32 compute units not shared with another replica, prompt + max_tokens <= 1024,
 * repetition context <= 64; MIS_TOKEN_ENGINE=0 disables it); 2 = the launch chain AFTER the engine's workers could not be made
 * co-resident (another stream held compute units; also written to stderr).  The two programs round at the same points but sum in other
 * orders: ids can differ on near-ties, so the choice never depends on stream vs non-stream, only on the request (and, reported, on 2). */
int32_t    mis_soprano_lm_path(const mis_soprano*);
/* generateStream (Soprano.swift:693-800) for a batch of tokenised sentences: MIS_EVENT_TOKEN per sampled id while the loop runs
 * (the [STOP] token is not announced, :855-857), then per row MIS_EVENT_INFO (SopranoGenerationInfo :771-779: prompt count and
 * prefill time 0, generation count = hidden states decoded) and ONE MIS_EVENT_AUDIO (:781).  cancel_flag polled every 8 steps
 * (token engine: every ~20 us, honoured at the launch's next position) (continuation.onTermination -> task.cancel(), :798)
 * -> MIS_ERR_CANCELLED.  At one row the events are fired while ONE persistent launch runs (csrc/token_engine.hip): the ids arrive in
 * host-visible memory, the calling thread polls them. */
mis_status mis_soprano_generate_stream(mis_soprano*, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                       const mis_gen_params* params, mis_event_cb on_event, void* user,
                                       const volatile int* cancel_flag);

/* ------------------------------------------------------------------------------------------
 * Qwen3-TTS.  Replaces Qwen3TTSModel.generate / generateStream -> generateVoiceDesign
 * (Sources/MLXAudioTTS/Models/Qwen3TTS/Qwen3TTS.swift:60-133,306-569), the talker and code predictor LMs
 * (Qwen3TTSTalker.swift, Qwen3TTSCodePredictor.swift), sampleToken (:1003-1118) and the speech-tokenizer decoder
 * (Qwen3TTSSpeechTokenizer.swift:888-1006).  Tokenisation and the ChatML prompt text stay on the host.
 * ---------------------------------------------------------------------------------------- */
typedef struct mis_qwen3tts mis_qwen3tts;
typedef struct {
    mis_lm_config talker;          /* Qwen3TTSTalkerConfig (:200-305): vocab_size = codec vocabulary (3072) */
    mis_lm_config predictor;       /* Qwen3TTSTalkerCodePredictorConfig (:6-67): vocab_size 2048 */
    int32_t num_code_groups;       /* 16 */
    int32_t text_hidden_size, text_vocab_size;
    int32_t codec_eos_token_id;    /* 2150 */
    int32_t tts_pad_token_id;      /* 151671 */
    /* Qwen3TTSTokenizerDecoderConfig (:307-385) */
    int32_t dec_latent_dim, dec_codebook_dim, dec_codebook_size, dec_decoder_dim, dec_hidden_size, dec_intermediate_size;
    int32_t dec_head_dim, dec_num_heads, dec_num_layers, dec_num_kv_heads, dec_num_quantizers, dec_num_semantic_quantizers;
    float   dec_rms_norm_eps, dec_rope_theta;
    int32_t n_upsample_rates;    int32_t upsample_rates[8];      /* [8,5,4,3] */
    int32_t n_upsampling_ratios; int32_t upsampling_ratios[8];   /* [2,2] */
    int32_t sample_rate;           /* 24000 */
} mis_qwen3tts_config;
/* sampleToken parameters (VoiceDesignGenerationSettings, Qwen3TTS.swift:651-664; defaults 0.9 / 1.0 / - / 1.05 / 0) */
typedef struct {
    int32_t  max_frames;           /* maxTokens (4096); per-row caps via row_max_frames = min(maxTokens, max(75, 6 * text tokens)) (:383) */
    float    temperature, top_p;
    int32_t  top_k;                /* 0 = off */
    float    repetition_penalty, min_p;
    uint64_t seed;
    int64_t  row_offset;
} mis_qwen3tts_params;
mis_status mis_qwen3tts_create(const mis_qwen3tts_config*, int device, mis_qwen3tts** out);
/* keys after the reference's sanitize steps: talker tensors with or without the "talker." prefix
 * (model.*, codec_head.weight, text_projection.*, code_predictor.*), speech-tokenizer decoder tensors as "decoder.*"
 * with the module-tree names Qwen3TTSSpeechTokenizer.sanitize produces */
mis_status mis_qwen3tts_set_tensor(mis_qwen3tts*, const char* name, const void* data, mis_dtype dtype,
                                   const int64_t* shape, int ndim);
/* the same keys in MLX's affine-quantised form (the published Qwen3-TTS checkpoints are 8 bit, Qwen3TTS.swift:1157-1170): the
 * Linear layers of the talker and the code predictor are streamed as codes (see mis_tts_set_tensor_quantized); embeddings,
 * text projection, predictor tables / heads are dequantised at load (gathered or folded tensors) */
mis_status mis_qwen3tts_set_tensor_quantized(mis_qwen3tts*, const char* name, const uint32_t* wq, const void* scales,
                                             const void* biases, mis_dtype sb_dtype, int64_t N, int64_t K, int group_size, int bits);
mis_status mis_qwen3tts_finalize(mis_qwen3tts*);
void       mis_qwen3tts_destroy(mis_qwen3tts*);
mis_tts*   mis_qwen3tts_talker(mis_qwen3tts*);            /* borrowed handle (parity taps) */
int        mis_qwen3tts_samples_per_frame(const mis_qwen3tts*);   /* 1920 */
int        mis_qwen3tts_num_code_groups(const mis_qwen3tts*);     /* 16: ints per frame of codes_out */
/* Prompts as prepareGenerationInputs builds them (:883-1000): prefill position p of row b is
 * text_projection(text_embedding[text_ids[b,p]]) (text id >= 0) plus codec_embedding[codec_ids[b,p]] (codec id >= 0);
 * trailing_ids = the text ids added to the generated frames' embeddings (then tts_pad).  int32 [batch, P] / [batch, Tt].
 * *codes_out (mis_free) int32 [batch, *codes_stride, num_code_groups]; n_frames[batch]. */
mis_status mis_qwen3tts_generate_codes(mis_qwen3tts*, const int32_t* text_ids, const int32_t* codec_ids,
                                       const int32_t* prefill_lens, int P, const int32_t* trailing_ids,
                                       const int32_t* trailing_lens, int Tt, int batch, const mis_qwen3tts_params* params,
                                       const int32_t* row_max_frames, int32_t** codes_out, int64_t* codes_stride,
                                       int32_t* n_frames);
mis_status mis_qwen3tts_decode(mis_qwen3tts*, const int32_t* codes, int batch, int T, float* wav_out);
mis_status mis_qwen3tts_decoder_tap(mis_qwen3tts*, const int32_t* codes, int batch, int T, int stage, float* out,
                                    int64_t capacity, int32_t* channels, int64_t* length);
/* streamingStep / resetStreamingState (Qwen3TTSSpeechTokenizer.swift:948-1006) as a session on the handle: carried conv
 * inputs (CausalConv1d.step :199-227, k7 conv steps :655-667,:710-722), transposed-conv overlap (:553-576) and the decoder
 * transformer's K/V cache live on the device between steps; a step computes only the new frames.
 * begin: batch rows, at most max_frames frames in the session, at most max_chunk_frames per step.
 * step:  codes int32 [batch, num_quantizers, n_frames] (host or device) = the NEXT n_frames frames of every row ->
 *        wav_out f32 [batch, n_frames * samples_per_frame] (host or device).
 * set_stream_exact(1): chunked decode bitwise equal to mis_qwen3tts_decode of the whole sequence.  Default 0 = the
 * reference's arithmetic: its overlap-add sums two biased transposed-conv outputs, so the first `stride` samples of each
 * decoder block after a chunk boundary carry that block's bias twice. */
mis_status mis_qwen3tts_set_stream_exact(mis_qwen3tts*, int exact);
mis_status mis_qwen3tts_decode_stream_begin(mis_qwen3tts*, int batch, int max_frames, int max_chunk_frames);
mis_status mis_qwen3tts_decode_stream_step(mis_qwen3tts*, const int32_t* codes, int n_frames, float* wav_out);
mis_status mis_qwen3tts_decode_stream_end(mis_qwen3tts*);
/* generateVoiceDesign for a batch of prepared prompts (Qwen3TTS.swift:306-569).  *pcm_out (mis_free) f32 [batch, *pcm_stride].
 * on_event == NULL or chunk_frames <= 0: all frames, then one whole-sequence decode (one MIS_EVENT_AUDIO per row if on_event).
 * on_event != NULL and chunk_frames > 0 = generateStream (chunk_frames = streamingInterval * 12.5, :394-395): every
 * chunk_frames frames a streaming step of the whole batch runs on a second stream WHILE the frame loop continues; each
 * row's new samples arrive as MIS_EVENT_AUDIO as soon as they are on the host (:492-505), the frames after the last full
 * chunk when the loop ends (:537-546).  A row's pcm is the concatenation of its chunks (see set_stream_exact). */
mis_status mis_qwen3tts_generate(mis_qwen3tts*, const int32_t* text_ids, const int32_t* codec_ids, const int32_t* prefill_lens,
                                 int P, const int32_t* trailing_ids, const int32_t* trailing_lens, int Tt, int batch,
                                 const mis_qwen3tts_params* params, const int32_t* row_max_frames, float** pcm_out,
                                 int64_t* pcm_stride, int64_t* pcm_lens, int32_t** codes_out, int64_t* codes_stride,
                                 int32_t* n_frames, int chunk_frames, mis_event_cb on_event, void* user,
                                 const volatile int* cancel_flag);
/* In-context voice cloning.  Replaces the reference-audio front end - extractSpeakerEmbedding (Qwen3TTS.swift:839-881:
 * computeMelSpectrogram(24 kHz, nFft 1024, hop 256, 128 mels) -> Qwen3TTSSpeakerEncoder, Qwen3TTSSpeakerEncoder.swift:20-307) and
 * Qwen3TTSSpeechTokenizerEncoder.encode (Qwen3TTSSpeechTokenizer.swift:792-881: Mimi SEANet encoder, causal transformer, edge-padded
 * downsampling conv, split residual VQ) - and the ReferenceAudioContext the model keeps (:268-300): codecEmbedIcl (:249-266) rows
 * and the speaker vector become extra prompt rows of the handle that prefill positions address by codec id. */
typedef struct {
    /* Qwen3TTSSpeakerEncoderConfig (Qwen3TTSConfig.swift:69-117); spk_n_blocks = entries of enc_channels, 0 = no speaker encoder */
    int32_t spk_mel_dim, spk_enc_dim, spk_n_blocks;
    int32_t spk_channels[8], spk_kernel_sizes[8], spk_dilations[8];
    int32_t spk_attention_channels, spk_res2net_scale, spk_se_channels, spk_sample_rate;
    /* Qwen3TTSTokenizerEncoderConfig (Qwen3TTSConfig.swift:388-497); enc_num_filters == 0 = the tokenizer has no encoder */
    int32_t enc_audio_channels, enc_num_filters, enc_kernel_size, enc_last_kernel_size, enc_residual_kernel_size;
    int32_t enc_num_residual_layers, enc_dilation_growth_rate, enc_compress;
    int32_t enc_n_ratios, enc_upsampling_ratios[8];      /* as configured ([8,6,5,4]); the encoder applies them reversed */
    int32_t enc_use_causal_conv, enc_use_conv_shortcut;
    int32_t enc_hidden_size, enc_num_layers, enc_num_heads, enc_intermediate_size;
    int32_t enc_codebook_dim, enc_codebook_size, enc_num_quantizers, enc_valid_num_quantizers, enc_sampling_rate;
    float   enc_rope_theta, enc_frame_rate, enc_norm_eps;
} mis_qwen3tts_reference_config;
/* before the first set_tensor of these keys: "speaker_encoder.*" (names as Qwen3TTSSpeakerEncoder.sanitize leaves them behind the
 * prefix, conv weights [out, k, in]) and "encoder_model.*" (Qwen3TTSSpeechTokenizer.sanitize :1093-1440) are then routed to the
 * front end by mis_qwen3tts_set_tensor and checked by mis_qwen3tts_finalize */
mis_status mis_qwen3tts_enable_reference(mis_qwen3tts*, const mis_qwen3tts_reference_config*);
/* audio f32 [n_samples] mono at spk_sample_rate (host or device) -> out f32 [spk_enc_dim] */
mis_status mis_qwen3tts_speaker_embedding(mis_qwen3tts*, const float* audio, int64_t n_samples, float* out);
/* audio f32 [n_samples] mono at enc_sampling_rate -> *codes_out (mis_free) int32 [*n_q][*n_frames], n_q = min(valid, num_quantizers) */
mis_status mis_qwen3tts_encode_audio(mis_qwen3tts*, const float* audio, int64_t n_samples, int32_t** codes_out, int32_t* n_q,
                                     int32_t* n_frames);
/* parity taps, out f32 [channels, length] (capacity floats).  kind 0 = speaker encoder: stage i < spk_n_blocks - 1 = output of
 * blocks[i], then mfa, then asp ([2C, 1]); kind 1 = tokenizer encoder: 0 SEANet, 1 transformer, 2 downsampled latent */
mis_status mis_qwen3tts_reference_tap(mis_qwen3tts*, int kind, const float* audio, int64_t n_samples, int stage, float* out,
                                      int64_t capacity, int32_t* channels, int64_t* length);
/* A reference context on the handle: codes int32 [n_q, T] (n_q <= num_code_groups) and optionally the speaker vector
 * (f32 [hidden_size of the talker], rounded to the talker's bf16 on entry).  Its prompt rows follow the codec vocabulary: a prefill
 * position whose codec id is vocab_size + *speaker_row reads the speaker vector, vocab_size + *first_frame_row + t reads
 * codec_embedding[code 0 of frame t] + sum_i code_predictor.codec_embedding[i][code i+1 of frame t] (one bf16 rounding per add,
 * codecEmbedIcl :249-266).  A non-streaming generate prepends the reference codes of a row whose prompt uses such frame rows to the
 * generated ones before decoding and cuts the proportional head (ref frames / total frames) off the waveform (:550-563).
 * *speaker_row = -1 without a speaker vector.  Contexts live until clear_references / destroy; at most 64. */
mis_status mis_qwen3tts_add_reference(mis_qwen3tts*, const int32_t* codes, int n_q, int T, const float* speaker_embedding,
                                      int speaker_dim, int32_t* speaker_row, int32_t* first_frame_row);
mis_status mis_qwen3tts_clear_references(mis_qwen3tts*);
/* stand-alone sampleToken (parity tests): logits f32 [batch, vocab], seen u8 [batch, vocab] or NULL */
mis_status mis_qwen3tts_sample_logits(int device, const float* logits, int batch, int vocab, const uint8_t* seen,
                                      const mis_qwen3tts_params* params, int suppress_lo, int suppress_hi, int eos_id, int step,
                                      int32_t* tokens_out);

/* ------------------------------------------------------------------------------------------
 * Descript DAC decoder.  Replaces DescriptDAC.decodeFromCodes / decode (Sources/MLXAudioCodecs/Descript/DescriptDAC.swift:
 * 103-160,235-242), DescriptResidualVectorQuantize.fromCodes (DescriptQuantization.swift:150-163) and, when the checkpoint
 * carries the encoder tensors ("encoder.*", "*.in_proj.*"), DescriptDAC.encode / encodeAudio (:40-95,216-233,340-345).
 * ---------------------------------------------------------------------------------------- */
typedef struct mis_dac mis_dac;
typedef struct {                   /* DescriptDACConfig.swift:3-34; latent_dim resolved (encoder_dim * 2^len(encoder_rates)) */
    int32_t latent_dim, decoder_dim;
    int32_t n_decoder_rates; int32_t decoder_rates[8];
    int32_t n_codebooks, codebook_size, codebook_dim;
    int32_t sample_rate;
} mis_dac_config;
mis_status mis_dac_create(const mis_dac_config*, int device, mis_dac** out);
/* checkpoint keys as stored; DescriptDAC.sanitize (:274-286) is applied */
mis_status mis_dac_set_tensor(mis_dac*, const char* name, const void* data, mis_dtype dtype, const int64_t* shape, int ndim);
mis_status mis_dac_finalize(mis_dac*);
void       mis_dac_destroy(mis_dac*);
int64_t    mis_dac_num_samples(const mis_dac*, int n_frames);    /* 250 frames @ [8,5,4,2] -> 80 043 (output_padding 1 per block) */
mis_status mis_dac_decode_codes(mis_dac*, const int32_t* codes, int batch, int T, float* wav_out);
/* encode / encodeAudio (DescriptDAC.swift:216-233,340-345; needs the checkpoint's encoder.* and in_proj tensors, else
 * MIS_ERR_AUDIO_ENCODE): audio f32 [batch, n_samples] is right-padded to the hop length (prod of the encoder strides) ->
 * codes int32 [batch, nq, padded / hop] with nq = n_quantizers (0 = all; the reference's nQuantizers); z_out (nullable) f32 [batch, latent, T] = the encoder output before the RVQ.
 * Codebook lookup = nearest L2-normalised code, first index on ties (DescriptQuantization.swift:76-94). */
int64_t    mis_dac_padded_length(const mis_dac*, int64_t n_samples);
mis_status mis_dac_encode(mis_dac*, const float* audio, int batch, int64_t n_samples, int n_quantizers, int32_t* codes_out, float* z_out);
mis_status mis_dac_debug_tap(mis_dac*, const int32_t* codes, int batch, int T, int block, float* out, int64_t capacity,
                             int32_t* channels, int64_t* length);

/* ------------------------------------------------------------------------------------------
 * EnCodec decoder.  Replaces Encodec.decodeFrame / EncodecDecoder (Sources/MLXAudioCodecs/Encodec/Encodec.swift:94-170,295-302)
 * incl. the 2-layer LSTM (EncodecLayers.swift:15-80) and EncodecResidualVectorQuantizer.decode (EncodecQuantization.swift:117-133).
 * norm_type "weight_norm" models (no GroupNorm), mono, causal.  Chunked decode = decode_frame per chunk + the host
 * linearOverlapAdd of the reference (Encodec.swift:304-355; mirrored in the Python host layer).
 * ---------------------------------------------------------------------------------------- */
typedef struct mis_encodec mis_encodec;
typedef struct {                   /* EncodecConfig.swift:64-89 */
    int32_t audio_channels, num_filters, kernel_size, num_residual_layers, dilation_growth_rate;
    int32_t codebook_size, codebook_dim, hidden_size, num_lstm_layers, residual_kernel_size;
    int32_t use_causal_conv, pad_reflect, last_kernel_size, compress, use_conv_shortcut;
    float   trim_right_ratio;
    int32_t n_upsampling_ratios; int32_t upsampling_ratios[8];
    int32_t n_quantizers;          /* EncodecQuantization.swift:60-64 */
    int32_t sampling_rate;
    int32_t group_norm;            /* 1: norm_type "time_group_norm" (the 48 kHz model): GroupNorm(1 group) after every conv, in the
                                      transposed-conv layer before the trim (EncodecLayers.swift:128-131,203-212,244-262); 0: plain convs */
} mis_encodec_config;
mis_status mis_encodec_create(const mis_encodec_config*, int device, mis_encodec** out);
/* module-tree keys: quantizer.layers.N.codebook.embed, decoder.layers.N.conv.{weight [out,k,in],bias},
 * decoder.layers.1.lstm.N.{Wx,Wh,bias}, decoder.layers.N.{block.1,block.3,shortcut}.conv.*, with group_norm also
 * <conv prefix>.norm.{weight,bias} [out]; "encoder.*" ignored */
mis_status mis_encodec_set_tensor(mis_encodec*, const char* name, const void* data, mis_dtype dtype, const int64_t* shape, int ndim);
mis_status mis_encodec_finalize(mis_encodec*);
void       mis_encodec_destroy(mis_encodec*);
int        mis_encodec_hop_length(const mis_encodec*);
/* wav_out f32 [batch, audio_channels, T * hop] (audio_channels 1 or 2) */
mis_status mis_encodec_decode_frame(mis_encodec*, const int32_t* codes, int batch, int n_q, int T, const float* scales, float* wav_out);
/* stage 1 conv0, 2 LSTM block, 3 + i upsampling block i: out f32 [batch, C, T'] */
mis_status mis_encodec_debug_tap(mis_encodec*, const int32_t* codes, int batch, int n_q, int T, int stage, float* out, int64_t capacity,
                                 int32_t* channels, int64_t* length);

/* ------------------------------------------------------------------------------------------
 * Log-mel / STFT front end.  Replaces WhisperAudio.logMelSpectrogram / encoderFeatures
 * (Sources/MLXAudioSTT/Models/Whisper/WhisperAudio.swift:38-87) and computeMelSpectrogram
 * (Sources/MLXAudioCore/DSP.swift:230-273): reflect pad, window, rfft, |.|^2, mel filterbank
 * (melFilters DSP.swift:76-168), log10, clamp to (max - 8), (x + 4) / 4.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t sample_rate, n_fft /* even, <= 2048 */, hop_length, n_mels /* <= 256 */;
    int32_t window;           /* 0 periodic Hann (WhisperAudio.swift:42-43), 1 symmetric Hann (DSP.swift:15-22) */
    int32_t mel_scale;        /* 0 HTK (DSP default), 1 Slaney (Whisper) */
    int32_t slaney_norm;      /* 1 = area normalisation (DSP.swift:155-162) */
    int32_t drop_last_frame;  /* 1 = Whisper (WhisperAudio.swift:65-67) */
} mis_mel_config;
int64_t    mis_mel_num_frames(const mis_mel_config*, int64_t n_samples);
/* pcm f32 [batch, n_samples] -> out f32 [batch, n_frames, n_mels]; every row is normalised by its own max */
mis_status mis_mel_spectrogram(int device, const mis_mel_config*, const float* pcm, int batch, int64_t n_samples,
                               float* out, int64_t* n_frames_out);
/* WhisperAudio.encoderFeatures: rows of `stride` samples with lens[b] valid (NULL = stride) are zero-padded /
 * trimmed to 480000 samples; out f32 [batch, 3000, n_mels], n_mels in {80, 128}. */
mis_status mis_whisper_encoder_features(int device, const float* pcm, const int64_t* lens, int batch, int64_t stride,
                                        int n_mels, float* out);

/* Streaming front end: IncrementalMelSpectrogram (Sources/MLXAudioSTT/Streaming/IncrementalMelSpectrogram.swift:17-215).
 * One handle per audio session; overlap-save framing and the running log-maximum are carried between calls.
 * out f32 [capacity_frames, n_mels]; *n_frames = frames produced by this call (0 while a frame is incomplete). */
typedef struct mis_mel_stream mis_mel_stream;
mis_status mis_mel_stream_create(int device, int sample_rate, int n_fft, int hop_length, int n_mels, mis_mel_stream** out);
mis_status mis_mel_stream_process(mis_mel_stream*, const float* samples, int64_t n, float* out, int64_t capacity_frames,
                                  int64_t* n_frames);
mis_status mis_mel_stream_flush(mis_mel_stream*, float* out, int64_t capacity_frames, int64_t* n_frames);
mis_status mis_mel_stream_reset(mis_mel_stream*);
int64_t    mis_mel_stream_total_frames(const mis_mel_stream*);
void       mis_mel_stream_destroy(mis_mel_stream*);

/* ------------------------------------------------------------------------------------------
 * Whisper STT.  Replaces WhisperModel / WhisperEncoder / WhisperDecoder
 * (Sources/MLXAudioSTT/Models/Whisper/WhisperModel.swift:36-309, WhisperLayers.swift:110-328).
 * bf16 compute (fp16 checkpoints are converted at load).  Tokenisation / prompt construction /
 * text decoding stay on the host (WhisperTokenizer.swift), as in the reference.
 * ---------------------------------------------------------------------------------------- */
typedef struct mis_whisper mis_whisper;
/* WhisperConfig, WhisperConfig.swift:3-22 */
typedef struct {
    int32_t vocab_size, num_mel_bins, d_model;
    int32_t encoder_layers, encoder_attention_heads, encoder_ffn_dim, max_source_positions;
    int32_t decoder_layers, decoder_attention_heads, decoder_ffn_dim, max_target_positions;
} mis_whisper_config;
/* generation knobs of transcribeChunk (WhisperModel.swift:213-236) + WhisperGenerationConfig suppress lists */
typedef struct {
    int32_t max_tokens;            /* STTGenerateParameters.maxTokens */
    float   temperature;           /* <= 0: greedy argmax (:284-287) */
    uint64_t seed;
    int32_t eot_id;                /* tokenizer.endOfTextId: ends a row, not emitted */
    int32_t timestamp_begin;       /* suppressFromIndex(:300-309): ids >= this are never sampled (0 = off) */
    const int32_t* suppress;       /* suppress_tokens, every step */
    int32_t n_suppress;
    const int32_t* begin_suppress; /* begin_suppress_tokens, first step only (default [eot]) */
    int32_t n_begin_suppress;
    int64_t row_offset;            /* global index of row 0 (sharding): the sampler's RNG is keyed by the global row */
} mis_stt_params;
mis_status mis_whisper_create(const mis_whisper_config*, int device, mis_whisper** out);
/* HF transformers key layout (model.encoder.* / model.decoder.*; conv weights [out, in, k]; proj_out ignored: tied) or the
 * mlx-whisper layout (encoder.blocks.N.attn.query.*, conv weights [out, k, in], encoder positional embedding synthesised):
 * WhisperModel.sanitize, WhisperModel.swift:321-478 */
mis_status mis_whisper_set_tensor(mis_whisper*, const char* name, const void* data, mis_dtype dtype,
                                  const int64_t* shape, int ndim);
mis_status mis_whisper_init_synthetic(mis_whisper*, uint64_t seed);
mis_status mis_whisper_finalize(mis_whisper*);
void       mis_whisper_destroy(mis_whisper*);
/* WhisperEncoder (+ the cross-attention K/V of every decoder layer): features f32 [batch, 3000, n_mels];
 * enc_out f32 [batch, 1500, d_model] or NULL.  Resets the decoder state. */
mis_status mis_whisper_encode(mis_whisper*, const float* features, int batch, float* enc_out);
mis_status mis_whisper_decoder_reset(mis_whisper*);
/* one decoder token per active row through the KV caches (WhisperDecoder.callAsFunction with Tnew = 1);
 * logits_out f32 [batch, vocab] (tied projection) or NULL */
mis_status mis_whisper_decoder_forward(mis_whisper*, const int32_t* tokens, const uint8_t* active, float* logits_out);
/* transcribeChunk for a batch of <= 30 s windows: pcm f32 [batch, stride] (lens[b] valid samples, NULL = stride),
 * prompt_ids = buildPromptTokens(...) shared by all rows.  *tokens_out (mis_free) int32 [batch, *tokens_stride],
 * n_tokens[batch] generated ids per row (EOT excluded). */
mis_status mis_stt_whisper_generate(mis_whisper*, const float* pcm, const int64_t* lens, int batch, int64_t stride,
                                    const int32_t* prompt_ids, int n_prompt, const mis_stt_params*,
                                    int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens);
/* generateStream (WhisperModel.swift:92-160): same work, announcing every sampled id as MIS_EVENT_TOKEN (row, id) while the loop
 * runs (every 4 steps; EOT is never announced) - the host turns ids into the reference's text deltas (decode-and-diff,
 * onTokenDelta :186-250) - then one MIS_EVENT_INFO per row (prompt_token_count, generation_token_count of the final .result).
 * cancel_flag polled at the same cadence -> MIS_ERR_CANCELLED.  tokens_out / tokens_stride / n_tokens may be NULL. */
mis_status mis_stt_whisper_generate_stream(mis_whisper*, const float* pcm, const int64_t* lens, int batch, int64_t stride,
                                           const int32_t* prompt_ids, int n_prompt, const mis_stt_params*,
                                           mis_event_cb on_event, void* user, const volatile int* cancel_flag,
                                           int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens);

/* ------------------------------------------------------------------------------------------
 * Device groups for the other families ("Whisper (30 s chunks), Soprano (sentence prompts) and Qwen3-TTS rows shard the
 * same way").  replicas[n]: one finalized handle per GPU holding the same weights.  The rows of the call are split into n contiguous
 * blocks (mis_shard_rows), every replica runs its block on its own worker thread through the single-device entry point - with
 * row_offset advanced by the block start, so a row's tokens / samples do not depend on n - and the results are gathered on the host
 * in global row order.  No collective is on the critical path; outputs and ownership are those of the single-device calls.
 * Callbacks are serialised (one at a time, any worker thread) and carry GLOBAL row indices.  Errors: the first failing shard's
 * status, message prefixed "shard r:". */
mis_status mis_whisper_group_generate(mis_whisper* const* replicas, int n, const float* pcm, const int64_t* lens, int batch, int64_t stride,
                                      const int32_t* prompt_ids, int n_prompt, const mis_stt_params*,
                                      int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens);
mis_status mis_soprano_group_generate(mis_soprano* const* replicas, int n, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                      const mis_gen_params* params, float** pcm_out, int64_t* pcm_stride, int64_t* pcm_lens,
                                      int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens);
/* arguments as mis_qwen3tts_generate; with on_event and chunk_frames > 0 every replica streams its rows' chunks as they are decoded
 * (generateStream): per-rank chunk emission, nothing is exchanged between GPUs.  In-context prompts address reference rows by codec id:
 * register the same reference contexts, in the same order, on every replica (mis_qwen3tts_add_reference). */
mis_status mis_qwen3tts_group_generate(mis_qwen3tts* const* replicas, int n, const int32_t* text_ids, const int32_t* codec_ids,
                                       const int32_t* prefill_lens, int P, const int32_t* trailing_ids, const int32_t* trailing_lens, int Tt,
                                       int batch, const mis_qwen3tts_params* params, const int32_t* row_max_frames, float** pcm_out,
                                       int64_t* pcm_stride, int64_t* pcm_lens, int32_t** codes_out, int64_t* codes_stride,
                                       int32_t* n_frames, int chunk_frames, mis_event_cb on_event, void* user,
                                       const volatile int* cancel_flag);

/* Diagnostics and test scaffolding (launch floor, split-factor model, CU spinner, failure counters) are NOT part of the product
 * surface: include/mi_speech_debug.h. */

#ifdef __cplusplus
}
#endif
#endif /* MI_SPEECH_H */
