// lm_kernels.h - launchers of the Llama decode-step kernels (lm_kernels.hip, lm_sampler.hip)
#pragma once
#include "common.h"

// element (m, k) of a packed activation [K/32][MT][64][8] (MFMA-B fragments: a wave's operand is one contiguous 1 KiB load)
__host__ __device__ __forceinline__ size_t xpk_index(int m, int k, int MT) {
    return ((((size_t)(k >> 5) * MT + (m >> 4)) * 64) + (((k & 31) >> 3) << 4) + (m & 15)) * 8 + (k & 7);
}

enum { EPI_PARTIAL = 0, EPI_BF16 = 1, EPI_SILU_MUL = 2, EPI_GELU_PACKED = 3, EPI_SILU_PACKED = 4 };

void launch_convert_to_bf16(const void* src, int dtype, bf16_t* dst, size_t n, hipStream_t s);
// MLX affine-quantised matrix (uint32 words, per-group scales / biases of dtype sb_dtype = mis_dtype) -> bf16 [N][K]
void launch_dequant_affine(const uint32_t* wq, const void* scales, const void* biases, int sb_dtype, bf16_t* dst, int N, int K,
                           int group, int bits, hipStream_t s);
void launch_pack_weight(const bf16_t* src, bf16_t* dst, int N, int K, int NT, int tile_stride, int tile_offset,
                        hipStream_t s);
void launch_synth_fill_bf16(bf16_t* dst, size_t n, uint64_t key, float amp, int plus_one, hipStream_t s);

void launch_prefill_feed(const int32_t* prompt, const int32_t* lens, int Lmax, int* step_counter, int32_t* ids,
                         uint8_t* active, int batch, hipStream_t s);
void launch_embed_rmsnorm(const bf16_t* emb, const int32_t* ids, const uint8_t* active, int* pos_cur, int* pos_next,
                          const bf16_t* wnorm, bf16_t* h, bf16_t* x, int d, int vocab, float eps, int batch, int Mpad,
                          hipStream_t s);
// ln_bias == nullptr: RMSNorm; otherwise LayerNorm(weight wnorm, bias ln_bias)
void launch_reduce_residual_rmsnorm(const float* slabs, int S, int Mpad, int N, bf16_t* h, const bf16_t* wnorm,
                                    bf16_t* x, float eps, hipStream_t s, const bf16_t* ln_bias = nullptr);
// Wp packed [NT][KT][64][8]; X bf16 [Mpad][KT*32]; out: f32 slabs [S][Mpad][N_out] (EPI_PARTIAL) or bf16 [Mpad][N_out]
// ksb = 1: each wave an independent item; ksb = 4: the block's 4 waves split the item's K range (LDS combine)
// bias (bf16 [N], optional): added once (slab 0 / final epilogue).  EPI_GELU_PACKED: T(gelu(T(xW+b))) written
// in the packed fragment layout (it is the next GEMM's X operand).  EPI_SILU_PACKED: h = T(xW+b), T(h * T(sigmoid(h))) packed.
// ksb = 2: 128-thread blocks, two waves per item.  U: k-tiles per register buffer (0: 4, or 3 at R = 4); R = 4 is built for <= 32 rows.
void launch_gemm_skinny(int epi, int R, int ksb, const bf16_t* Wp, const bf16_t* X, void* out, int NT, int KT, int S,
                        int N_out, int Mpad, hipStream_t s, const bf16_t* bias = nullptr, int U = 0);
// The same GEMM with the residual add + LayerNorm / RMSNorm glue in its prologue (<= 16 rows, K <= 1280; see k_gemm_skinny_norm): X is rebuilt per
// block from the producer's S_in split-K slabs [S_in][16][K] and the residual stream h_in [16][K]; h_new is written to h_out (!= h_in).
// ln_bias == nullptr: RMSNorm.  R = 2, eight waves per item.
bool gemm_skinny_norm_ok(int epi, int R, int KT, int S, int S_in, int Mpad, int nrows);
void launch_gemm_skinny_norm(int epi, int R, const bf16_t* Wp, const float* slabs, int S_in, const bf16_t* h_in, bf16_t* h_out, const bf16_t* wnorm,
                             const bf16_t* ln_bias, float eps, int nrows, void* out, int NT, int KT, int S, int N_out, int Mpad, hipStream_t s,
                             const bf16_t* bias = nullptr);
// one role's arrangement of the weight-streaming GEMM: n-tiles per wave, waves per item, k-tiles per register buffer, inter-block split
struct GemmArr { int R = 2, ksb = 4, U = 4, S = 1; };

// the same on MLX affine-quantised weights (lm_qgemm.hip): Qp packed codes, SB packed bf16 scale/bias pairs, G = K/64 scale groups
void launch_gemm_skinny_q(int bits, int epi, int R, int ksb, const void* Qp, const bf16_t* SB, const bf16_t* X, void* out, int NT, int G,
                          int S, int N_out, int Mpad, hipStream_t s, const bf16_t* bias = nullptr);
void launch_pack_qweight(int bits, const uint32_t* wq, const bf16_t* scales, const bf16_t* biases, void* qdst, bf16_t* sbdst, int N, int K,
                         int tile_stride, int tile_offset, hipStream_t s);

// batched prefill (lm_prefill.hip): M = positions x rows
enum { PF_F32 = 0, PF_RESID = 1, PF_SILU = 2 };
void launch_gemm_pf(int epi, const bf16_t* X, const bf16_t* Wp, void* C, int M, int N, int K, hipStream_t s);
void launch_pf_embed_rmsnorm(const bf16_t* emb, const int32_t* prompt, const int32_t* lens, int Lmax, int t0, int Tc, int batch, int Mpad, int vocab,
                             const bf16_t* wnorm, bf16_t* h, bf16_t* x, int32_t* pos_tab, uint8_t* act_tab, int d, float eps, hipStream_t s);
void launch_pf_add_resid(bf16_t* h, const float* part, size_t n, hipStream_t s);
void launch_pf_rows_rmsnorm(const bf16_t* rows, int src_rows, const int32_t* lens, int Lmax, int t0, int Tc, int batch, int Mpad, const bf16_t* wnorm,
                            bf16_t* h, bf16_t* x, int32_t* pos_tab, uint8_t* act_tab, int d, float eps, hipStream_t s);
void launch_pf_rmsnorm(const bf16_t* h, const bf16_t* wnorm, bf16_t* x, int rows, int d, float eps, hipStream_t s);
void launch_pf_pack_rows(const bf16_t* rows, bf16_t* xpk, int Mpad, int d, hipStream_t s);

// split-K factor of a weight-streaming GEMM (items = n-tile groups, KT = k-tiles, ksb = waves per item), see the definition
int gemm_choose_split(int items, int KT, int ksb, int s_max);

struct AttnParams;
bool attn_qp_ok(const AttnParams& p);
struct AttnParams {
    const float* qkv_part;   // [S][Mpad][Nqkv]
    int S, Mpad, Nqkv;
    bf16_t* kcache;          // layer slice [B][Hkv][Smax][D]
    bf16_t* vtcache;         // layer slice [B][Hkv][D][Smax]
    const int* pos;          // [Mpad] position of the token being processed
    const uint8_t* active;   // [Mpad]
    const float* rope_cos;   // [Smax][D/2]; null = no rotary (Whisper)
    const float* rope_sin;
    const bf16_t* qnorm_w;   // [D] per-head q RMSNorm weight (Qwen3-style), null = none
    const bf16_t* knorm_w;   // [D]
    float qk_eps;
    int rope_in_dtype;       // 1: T(T(x cos) + T(rot(x) sin)) with cos/sin rounded to bf16 (Qwen3-TTS)
    int cross;               // 1: cross attention - queries only, no append, keys 0..cross_len-1 of the given caches
    int cross_len;
    bf16_t* out;             // [Mpad][H*D] as packed MFMA-B fragments (out_ld == 0) or row-major with row stride out_ld
    int out_ld;
    int H, Hkv, D, Smax;
    float scale;
    unsigned long long* dbg; // phase timestamps (MIS_ATTN_TIMING builds only), else null
    // cross-attention with the glue and the query projection in its prologue (qp_w != null; see attn_qp_row_gemv): q = T(W_q LN(h_new) + b),
    // h_new = T(h_in + T(sum of the qp_S slabs [qp_S][Mpad][H D])) written to qp_h_out (!= qp_h_in) by the head-0 blocks; qkv_part / S unused
    const bf16_t* qp_w;      // W_q packed [H D / 16][qp_KT][64][8]
    const bf16_t* qp_bias;   // [H D] or null
    const float* qp_slabs;
    const bf16_t* qp_h_in;
    bf16_t* qp_h_out;
    const bf16_t *qp_lnw, *qp_lnb;
    float qp_eps;
    int qp_S, qp_KT;
    // batched prefill: the rows of a launch are (position, sequence) pairs - row r uses the caches of sequence r % cache_rows (0 = r).
    // append_only = 1: RoPE / norm the new key and value, write them to the caches, stop (all positions of a chunk first: the second
    // launch then finds every earlier key of its own chunk in memory)
    int cache_rows, append_only;
    int first_schedule;      // 1: never k_attn_decode2 - the prefill's two arrangements must give the same bits whatever the batch size
};
void launch_attn_decode(const AttnParams& p, int batch, hipStream_t s);

#define SAMP_MAX_CHUNKS 64
#define SAMP_CLUSTER_NB 8
struct SamplerScratch {             // per row, device memory
    unsigned long long hist1[256], hist2[256], cmass[SAMP_MAX_CHUNKS];
    unsigned long long below1, Z, thr;
    float pmax[SAMP_MAX_CHUNKS];
    int pidx[SAMP_MAX_CHUNKS];
    unsigned bin1, kstar;
    // k_samp_cluster (one launch, 8 blocks per row): the row's exchange area - one slot per block, overwritten by its owner before the
    // barrier that releases it to the readers (nothing to zero between launches); `c_sync` only ever grows; c_fail: a row barrier timed out
    alignas(128) unsigned long long x_hist1[SAMP_CLUSTER_NB][256];
    alignas(128) unsigned long long x_hist2[SAMP_CLUSTER_NB][256];
    alignas(128) unsigned long long x_max[SAMP_CLUSTER_NB];
    alignas(128) unsigned long long x_above[SAMP_CLUSTER_NB];
    alignas(128) unsigned int c_sync;
    alignas(128) unsigned int c_fail;
};
void sampler_scratch_init(SamplerScratch* scratch, int batch, hipStream_t s);
// true when any row of the scratch reported a timed-out row barrier of the one-launch sampler (synchronises `s`); the caller raises
// MIS_ERR_GENERATION_FAILED and re-initialises the scratch
bool sampler_check_failed(SamplerScratch* scratch, int batch, hipStream_t s);
void sampler_plan(int vocab, int* n_chunks, int* chunk_w);

struct SamplerParams {
    SamplerScratch* scratch; // [batch]
    int n_chunks, chunk_w;   // vocabulary split (sampler_plan)
    bf16_t* logits;          // [Mpad][Vpad]  (penalty is applied in place)
    float* e_buf;            // [Mpad][Vpad] scratch
    float* logits32;         // [Mpad][Vpad] float32 processed logits (penalty_flavor 1), else null
    int Vpad, vocab;
    const uint8_t* active_in;   // rows to sample (null = all)
    // generation state (all device memory)
    int32_t* window;         // [B][ctx] ring, right-aligned valid part
    int32_t* window_len;     // [B]
    int ctx;
    int32_t* n_gen;          // [B] tokens generated so far (RNG step / frame slot)
    int32_t* tokens_out;     // [B][tokens_stride] every sampled id (EOS included) or null
    int tokens_stride;
    int32_t* all_ids;        // [B][all_stride] prompt + generated (EOS excluded) or null
    int32_t* all_len;        // [B]
    int all_stride;
    int32_t* next_ids;       // [B] next input token or null
    uint8_t* active;         // [B] cleared on EOS (may alias active_in) or null
    int32_t* done_count;     // incremented on EOS or null
    int32_t* step_override;  // null, or [B] explicit RNG step (stand-alone sampling)
    // parameters
    float temperature, top_p, penalty;
    int penalty_flavor;      // 0 mlx-lm RepetitionContext (unique ids, bf16); 1 Soprano (per occurrence, f32, sign rule > 0)
    uint64_t seed;
    int64_t row_offset;
    int frame_constrained;
    int audio_offset;        // first audio token id (frame-constrained range): 0 = Orpheus 128266
    int lo, hi;              // allowed id range when not frame constrained (hi <= 0 -> vocab)
    int eos_id;
    int max_tokens;
    unsigned long long* dbg; // diagnostics (null in the product): [16] s_memtime stamps of block (0, 0) of the one-launch sampler
    // host side only (sampler_resolve): which kernels the launcher may pick - fixed per generate call and hashed into the graph key
    int path_resolved;       // 0: launch_sampler reads MIS_SAMPLER_WIDE / MIS_SAMPLER_SPIN itself, per launch (stand-alone entry point)
    int force_multi;         // 1: the multi-launch kernels whatever the range (shared device, recovery after a time-out, MIS_SAMPLER_WIDE)
    int spin;                // polls per row barrier of the one-launch sampler before it gives up
};
// resolves the launcher's choices once (environment + the caller's multi_launch_only) into p
void sampler_resolve(SamplerParams& p, bool multi_launch_only);
// failure flags of the one-launch sampler (one per row): queued copy into pinned host memory, to be read after the caller's next
// stream synchronisation; sampler_note_failure counts the event and re-initialises the scratch
void sampler_fail_flags_async(SamplerScratch* scratch, int batch, unsigned* host_flags, hipStream_t s);
bool sampler_fail_flags_any(const unsigned* host_flags, int batch);
void sampler_note_failure(SamplerScratch* scratch, int batch, hipStream_t s);
// multi_launch_only: the six-kernel path whatever the range (the fall-back after a failed one-launch attempt, A/B)
void launch_sampler(const SamplerParams& p, int batch, hipStream_t s, bool multi_launch_only = false);
