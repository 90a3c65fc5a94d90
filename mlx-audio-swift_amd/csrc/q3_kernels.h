// q3_kernels.h - Qwen3-TTS kernels shared between q3_sampler.hip, qwen3tts.hip and q3_codec.hip
#pragma once
#include "common.h"

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
