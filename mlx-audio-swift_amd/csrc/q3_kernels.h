// q3_kernels.h - Qwen3-TTS kernels shared between q3_sampler.hip, qwen3tts.hip and q3_codec.hip
#pragma once
#include "common.h"
#include <vector>

struct Q3SampleArgs {
    const bf16_t* logits;      // [Mpad][Vpad]
    int Vpad, V;
    float temperature, top_p, min_p, penalty, log_min_p;   // log_min_p = T(float(log(double(min_p)))) from the host
    int top_k;
    int sup_lo, sup_hi, eos;   // suppressed id range (eos exempt); eos < 0: none
    uint8_t* seen;             // [B][Vpad] generated-id bitmap (repetition penalty over unique ids) or null
    uint64_t seed;
    int64_t row_offset;
    const int* frame;          // device frame counter; RNG step = frame * G + slot
    int slot, G;
    int32_t* cur_codes;        // [G][Mpad]: codes of the frame being built
    int Mpad;
    uint8_t* active_a;         // rows to sample; cleared on EOS
    uint8_t* active_b;         // second flag array cleared on EOS (code predictor) or null
    int32_t* done_count;
    int32_t* tokens_dbg;       // [B] sampled id incl. EOS (stand-alone entry point) or null
};
void launch_q3_sample(const Q3SampleArgs& a, int batch, hipStream_t s);

// q3_codec.hip kernels shared with the reference-audio front end (q3_reference.hip)
struct Q3AttnArgs {
    const float* q; int64_t q_bs; int q_ld;
    const float* k; const float* v; int64_t kv_bs; int kv_ld;
    float* out; int64_t o_bs; int o_ld;
    int H, Hkv, Tq, pos0;
    float theta, scale;
};
void launch_q3_attn(const Q3AttnArgs& a, int head_dim /*16, 32, 64*/, int batch, hipStream_t s);
// normalisation over the channel axis of [B][C][T] data (row stride ld); rms = 1: RMSNorm (bias unused), 0: LayerNorm
void launch_q3_norm_ct(const float* x, float* y, const float* w, const float* bias, int batch, int C, int Tn, int ld, float eps, int rms,
                       hipStream_t s);

// q3_reference.hip: speaker encoder (ECAPA-TDNN) + speech-tokenizer encoder (Mimi) of the in-context voice-cloning path
struct mis_q3ref;
mis_q3ref* q3ref_create(const mis_qwen3tts_reference_config* cfg, int device, hipStream_t s);
void q3ref_destroy(mis_q3ref*);
bool q3ref_owns(const char* name);                                   // "speaker_encoder.*" / "encoder_model.*"
void q3ref_set_tensor(mis_q3ref*, const char* name, const void* data, mis_dtype dtype, const int64_t* shape, int ndim);
void q3ref_finalize(mis_q3ref*);
// audio: host or device pointer; stage < 0: the final result (out = x-vector [enc_dim] / nothing, codes filled)
void q3ref_speaker(mis_q3ref*, const float* audio, int64_t n, int stage, float* out, int64_t capacity, int* C, int64_t* T);
void q3ref_encode(mis_q3ref*, const float* audio, int64_t n, int stage, float* out, int64_t capacity, int* C, int64_t* T,
                  std::vector<int32_t>* codes, int* n_q);
int q3ref_speaker_dim(const mis_q3ref*);
