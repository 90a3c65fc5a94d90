// whisper_kernels.h - launchers of the Whisper encoder kernels (whisper_kernels.hip)
#pragma once
#include "common.h"

enum { BG_NONE = 0, BG_GELU = 1, BG_RESID = 2, BG_GELU_POS = 3 };
struct BigGemmParams {
    const bf16_t* X;     // [M][ldx] row-major, first K columns used
    const bf16_t* W;     // [N][K] row-major (Linear.weight layout)
    const bf16_t* bias;  // [N] or null
    const bf16_t* R;     // BG_RESID: [M][N]; BG_GELU_POS: positional table [pos_rows][N] indexed by m % pos_rows
    bf16_t* C;           // [M][N]
    int M, N, K, ldx, pos_rows;
};
void launch_gemm_big(int epi, const BigGemmParams& p, hipStream_t s);
void launch_layernorm(const bf16_t* x, bf16_t* y, const bf16_t* w, const bf16_t* b, int rows, int d, float eps, hipStream_t s);
void launch_im2col3_f32(const float* in, bf16_t* out, int B, int Tin, int C, int Tout, int stride, hipStream_t s);
void launch_im2col3_bf16(const bf16_t* in, bf16_t* out, int B, int Tin, int C, int Tout, int stride, hipStream_t s);
void launch_scatter_kv(const bf16_t* src, int ld, int kcol0, int vcol0, bf16_t* kc, bf16_t* vc, int B, int T, int H, int D,
                       int Spad, hipStream_t s);
void launch_attn_prefill(const bf16_t* q, int ldq, const bf16_t* kc, const bf16_t* vc, bf16_t* out, int ldo, int B, int T,
                         int H, int D, int Spad, hipStream_t s);
void launch_whisper_embed_ln(const bf16_t* emb, const bf16_t* pos_emb, const int32_t* ids, const uint8_t* active, int* pos_cur,
                             int* pos_next, const bf16_t* lw, const bf16_t* lb, bf16_t* h, bf16_t* x, int d, int vocab,
                             int max_pos, int batch, int Mpad, hipStream_t s);
void launch_whisper_suppress(bf16_t* logits, int Vpad, int vocab, const int32_t* sup, int n_sup, const int32_t* bsup, int n_bsup,
                             const int32_t* n_gen, const uint8_t* active, int batch, hipStream_t s);
