// token_engine.hip - a batch-1 decode engine for small LMs: ONE persistent launch runs every token of a request on the compute units
// of ONE XCD.
//
// Why (BASELINE configs[1], Soprano-80M at batch 1): the launch chain spends ~125 graph nodes x 4.7 us on a
// token whose weights (160 MB) stream in 25 us - 0.03 of HBM.  At one row every op's output is a vector of 1-5 KB, so the per-op cost
// is the hand-off between compute units, and that hand-off is cheap INSIDE an XCD and expensive across the chip.  Measured before this
// was written (tools/token_engine_lab/probe.hip, profiles/r05/c2_xcd_stream_exchange_probe.json): an all-to-all edge (publish a slice,
// arrive on a counter, poll, gather the vector) among the 32 CUs of one XCD costs 1.65 us at 1 KB / 2.29 us at 4.6 KB, among 64 / 128 /
// 256 CUs 2.4 / 3.2 / 6.2 us; one XCD streams 1.28 TB/s out of the Infinity Cache (2.6 / 4.9 / 7.1 TB/s for 2 / 4 / 8).  Per token
// (69 edges, 160 MB): ~131 us of stream + ~126 us of edges on ONE XCD against 24 + 441 on eight - the chip-wide form is the launch
// chain's price again, the single-XCD form is 2-4x under it.
//
// Structure.  Grid = one 512-thread block per CU; the blocks whose index mod 8 is below `xcds` are the W = 32 x xcds workers (observed
// placement: block b runs on XCD b mod 8 - used for speed only; every hand-off below is placement-independent), the others exit.
// Every worker holds the whole residual stream in LDS and owns a slice of the OUTPUT rows of every matrix:
//   per layer   RMSNorm (local) -> q|k|v slice                       -> edge 1 (Nqkv values)
//               q/k-norm, RoPE, attention over its PRIVATE K/V copy - computed redundantly by every worker (at one row it is ~50 KB
//               of cache reads; a private copy needs no coherence protocol and saves an edge) -> o_proj slice, residual -> edge 2 (d)
//               RMSNorm (local) -> gate|up pairs, SwiGLU             -> edge 3 (ff)
//               down_proj slice, residual                            -> edge 4 (d)
//   per token   final norm -> output-projection slice -> local arg-max -> edge 5 (one candidate per worker) -> next token's embedding
// A GEMV slice runs on v_mfma_f32_16x16x32_bf16 with the engine's pre-packed weight tiles [N/16][K/32][64][8] as the A operand and the
// input vector as row 0 of the B operand (15/16 of the matrix core idles - at one row the tile stream is the cost, not the math); the
// eight waves of a worker split a tile row's K range and combine through LDS in a fixed order.
// An edge = 8-byte agent-scope stores of the worker's values (bf16 x 4 per granule) into a double-buffered vector, one arrival on a
// monotonic counter, a bounded poll, 8-byte agent-scope loads of the whole vector (MI355X_MICROARCH.md: "8-B agent atomics both sides").
// Rounding points are the oracle's (oracle/llama.py = MLX's bf16 graph): every primitive output rounded to bf16, float32 accumulation.
//
// State of this file: a measured laboratory behind include/mi_speech_debug.h (greedy decoding, Soprano-80M's widths compiled in); it
// reads the product handle's weights, so its logits are compared with the product's and the oracle's (tests/test_gpu_token_engine.py).
#include "common.h"
#include "kernels.h"

namespace {
typedef unsigned long long u64;

// the one shape compiled in: Soprano-80M's LM (SopranoConfig.swift:103-167; layer count and vocabulary stay run-time)
struct TeShape {
    static constexpr int d = 512, ff = 2304, H = 4, Hkv = 1, D = 128, Nqkv = (H + 2 * Hkv) * D, HD = H * D;
};
constexpr int TE_NT = 512, TE_NW = 8;          // threads / waves per worker
constexpr int TE_CTX = 512;                    // positions per request (scores in LDS)
constexpr int TE_XG = 1024;                    // granules per exchange buffer (4096 bf16 values)

struct TeParams {
    const bf16_t *emb, *wqkv, *wo, *wgu, *wdown, *head, *norms, *qknorm;
    const float *rope_cos, *rope_sin;
    int L, V, Vpad;
    float eps;
    const int32_t* prompt;
    int n_prompt, n_total;
    int32_t* next_tokens;       // [n_total]: arg-max after position t
    float* logits_out;          // [n_total][V] or null
    float* hidden_out;          // [n_total][d] or null (final-norm output: what Soprano's decoder consumes)
    bf16_t* kv;                 // private K/V copies [W][L][2][TE_CTX][Hkv*D]
    u64* xbuf;                  // [2][TE_XG]
    unsigned* counter;          // monotonic arrivals
    unsigned* fail;             // set when a poll ran out (workers not co-resident)
    int xcds, spin;
};

__device__ __forceinline__ unsigned te_key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

// ---- one edge: the caller has issued its granule stores into `buf` (this edge's half of xbuf); arrive, wait for all W workers
__device__ __forceinline__ bool te_meet(const TeParams& p, unsigned& edge, int W, int* s_ok) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(p.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)W * (edge + 1u);
        int ok = 0;
        for (int it = 0; it < p.spin; ++it) {
            if ((int)(__hip_atomic_load(p.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        *s_ok = ok;
    }
    __syncthreads();
    edge += 1u;
    return *s_ok != 0;
}
__device__ __forceinline__ u64 te_pack4(float a, float b, float c, float d) {
    return (u64)f32_to_bf16(a) | ((u64)f32_to_bf16(b) << 16) | ((u64)f32_to_bf16(c) << 32) | ((u64)f32_to_bf16(d) << 48);
}

// ---- a worker's slice of y = W x: RMAX tile rows (ids nt[r], -1 = none), every wave takes its share of the KT k-tiles; partial sums of
// the row-0 column land in red[wave][r][16].  All loads are issued before the first MFMA (unconditional, clamped addresses).
template <int RMAX, int KPW>
__device__ __forceinline__ void te_gemv(const bf16_t* __restrict__ Wp, const int KT, const int (&nt)[RMAX], const bf16_t* xb, float* red,
                                        const int wave, const int lane) {
    const int kt0 = wave * KT / TE_NW, kt1 = (wave + 1) * KT / TE_NW;
    const bf16x8_t* wp = reinterpret_cast<const bf16x8_t*>(Wp);
    bf16x8_t a[RMAX][KPW], xf[KPW];
    const int klast = kt1 > kt0 ? kt1 - 1 : kt0;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int tile = nt[r] < 0 ? 0 : nt[r];
#pragma unroll
        for (int u = 0; u < KPW; ++u) {
            int kk = kt0 + u;
            kk = kk > klast ? klast : kk;
            kk = kk >= KT ? KT - 1 : kk;
            a[r][u] = __builtin_nontemporal_load(wp + ((size_t)tile * KT + kk) * 64 + lane);
        }
    }
    const bool row0 = (lane & 15) == 0;
    const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < KPW; ++u) {
        int kk = kt0 + u;
        const bool live = kk < kt1;
        kk = kk > klast ? klast : kk;
        kk = kk >= KT ? KT - 1 : kk;
        const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(xb + 32 * kk + 8 * (lane >> 4));
        xf[u] = (row0 && live) ? v : zero;
    }
    f32x4_t acc[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) acc[r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KPW; ++u)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][u], xf[u], acc[r], 0, 0, 0);
    if (row0) {
        const int g = lane >> 4;
#pragma unroll
        for (int r = 0; r < RMAX; ++r)
            *reinterpret_cast<f32x4_t*>(red + ((size_t)(wave * RMAX + r) * 16 + 4 * g)) = acc[r];
    }
}
// sum of the eight waves' partials for element (r, i), fixed order
template <int RMAX>
__device__ __forceinline__ float te_combine(const float* red, int r, int i) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < TE_NW; ++w) s += red[(size_t)(w * RMAX + r) * 16 + i];
    return s;
}

// block-wide sum (512 threads), result to every thread; s_red: 8 floats
__device__ __forceinline__ float te_block_sum(float v, float* s_red, int wave, int lane) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < TE_NW; ++w) t += s_red[w];
    return t;
}

template <int XCDS>
__global__ void __launch_bounds__(TE_NT) k_token_engine(TeParams p) {
    using S = TeShape;
    constexpr int W = 32 * XCDS;
    constexpr int R_QKV = (S::Nqkv / 16 + W - 1) / W, R_O = (S::d / 16 + W - 1) / W, P_GU = (S::ff / 16 + W - 1) / W, R_GU = 2 * P_GU;
    constexpr int R_HEAD = 8;                                        // tile rows of the output projection per pass
    constexpr int KPW_D = (S::d / 32 + TE_NW - 1) / TE_NW, KPW_HD = (S::HD / 32 + TE_NW - 1) / TE_NW, KPW_FF = (S::ff / 32 + TE_NW - 1) / TE_NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char te_lds_pad[];      // (requested size keeps the launch at one block per CU)
    __shared__ __attribute__((aligned(16))) float hf[S::d];                         // residual stream (bf16 values)
    __shared__ __attribute__((aligned(16))) bf16_t xb[S::ff > S::d ? S::ff : S::d]; // the GEMV input vector
    __shared__ __attribute__((aligned(16))) float qkvf[S::Nqkv];
    __shared__ __attribute__((aligned(16))) float qh[S::H][S::D];
    __shared__ __attribute__((aligned(16))) float knew[S::Hkv * S::D], vnew[S::Hkv * S::D];
    __shared__ __attribute__((aligned(16))) float sc[S::H][TE_CTX];
    __shared__ __attribute__((aligned(16))) float red[TE_NW * (R_GU > R_HEAD ? R_GU : R_HEAD) * 16];
    __shared__ float s_red[TE_NW];
    __shared__ u64 s_cand[TE_NW];
    __shared__ int s_ok;
    __shared__ int s_tok;
    if (p.n_total < 0) te_lds_pad[threadIdx.x] = 0;
    const int b = blockIdx.x;
    if ((b & 7) >= XCDS) return;
    const int w = (b >> 3) * XCDS + (b & 7);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned edge = 0;
    bf16_t* kv_mine = p.kv + (size_t)w * p.L * 2 * TE_CTX * (S::Hkv * S::D);
    const float scale = rsqrtf((float)S::D);

    for (int t = 0; t < p.n_total; ++t) {
        // ---- token id and embedding row (every worker reads its own copy: 1 KB)
        if (tid == 0) s_tok = t < p.n_prompt ? p.prompt[t] : s_tok;
        __syncthreads();
        const int tok = s_tok;
        for (int i = tid; i < S::d; i += TE_NT) hf[i] = bf16_to_f32(p.emb[(size_t)tok * S::d + i]);
        __syncthreads();
        for (int li = 0; li < p.L; ++li) {
            // ================= RMSNorm -> q|k|v slice -> edge 1
            {
                const bf16_t* wn = p.norms + (size_t)(2 * li) * S::d;
                float v = tid < S::d ? hf[tid] : 0.f;
                const float ss = te_block_sum(v * v, s_red, wave, lane);
                const float inv = rsqrtf(ss / (float)S::d + p.eps);
                if (tid < S::d) xb[tid] = f32_to_bf16(bf16_to_f32(wn[tid]) * bf16_round_f32(v * inv));
                __syncthreads();
                int nt[R_QKV];
#pragma unroll
                for (int r = 0; r < R_QKV; ++r) nt[r] = (w + r * W) < S::Nqkv / 16 ? w + r * W : -1;
                te_gemv<R_QKV, KPW_D>(p.wqkv + (size_t)li * S::Nqkv * S::d, S::d / 32, nt, xb, red, wave, lane);
                __syncthreads();
                u64* buf = p.xbuf + (size_t)(edge & 1u) * TE_XG;
                if (tid < R_QKV * 4) {
                    const int r = tid >> 2, g = tid & 3;
                    if (nt[r] >= 0)
                        __hip_atomic_store(buf + nt[r] * 4 + g, te_pack4(te_combine<R_QKV>(red, r, 4 * g), te_combine<R_QKV>(red, r, 4 * g + 1),
                                                                          te_combine<R_QKV>(red, r, 4 * g + 2), te_combine<R_QKV>(red, r, 4 * g + 3)),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!te_meet(p, edge, W, &s_ok)) { if (tid == 0) *p.fail = 1u; return; }
                if (tid < S::Nqkv / 4) {
                    const u64 gq = __hip_atomic_load(buf + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int e = 0; e < 4; ++e) qkvf[4 * tid + e] = bf16_to_f32((bf16_t)(gq >> (16 * e)));
                }
                __syncthreads();
            }
            // ================= q/k-norm, RoPE, attention (redundant on every worker, private K/V copy), o_proj slice -> edge 2
            {
                bf16_t* kc = kv_mine + ((size_t)li * 2 + 0) * TE_CTX * (S::Hkv * S::D);
                bf16_t* vc = kv_mine + ((size_t)li * 2 + 1) * TE_CTX * (S::Hkv * S::D);
                const int pos = t, ctx = t + 1;
                if (wave < S::H + S::Hkv) {                                     // one wave per q head / k head: lane holds elements lane, lane + 64
                    const bool is_k = wave >= S::H;
                    const float* src = qkvf + (is_k ? S::HD + (wave - S::H) * S::D : wave * S::D);
                    const bf16_t* nw = p.qknorm + (size_t)(2 * li + (is_k ? 1 : 0)) * S::D;
                    const float x1 = src[lane], x2 = src[lane + 64];
                    const float ss = wave_sum(x1 * x1 + x2 * x2);
                    const float inv = rsqrtf(ss / (float)S::D + p.eps);
                    const float y1 = bf16_round_f32(bf16_to_f32(nw[lane]) * bf16_round_f32(x1 * inv));
                    const float y2 = bf16_round_f32(bf16_to_f32(nw[lane + 64]) * bf16_round_f32(x2 * inv));
                    const float c = p.rope_cos[(size_t)pos * (S::D / 2) + lane], sn = p.rope_sin[(size_t)pos * (S::D / 2) + lane];
                    const float o1 = bf16_round_f32(y1 * c - y2 * sn), o2 = bf16_round_f32(y1 * sn + y2 * c);
                    if (is_k) {
                        const int kh = wave - S::H;
                        knew[kh * S::D + lane] = o1; knew[kh * S::D + lane + 64] = o2;
                        kc[(size_t)pos * (S::Hkv * S::D) + kh * S::D + lane] = f32_to_bf16(o1);
                        kc[(size_t)pos * (S::Hkv * S::D) + kh * S::D + lane + 64] = f32_to_bf16(o2);
                    } else {
                        qh[wave][lane] = o1; qh[wave][lane + 64] = o2;
                    }
                } else if (wave == S::H + S::Hkv) {                              // values: appended as they are
                    for (int i = lane; i < S::Hkv * S::D; i += 64) {
                        const float v = qkvf[S::HD + S::Hkv * S::D + i];
                        vnew[i] = v;
                        vc[(size_t)pos * (S::Hkv * S::D) + i] = f32_to_bf16(v);
                    }
                }
                __syncthreads();
                // scores: 16 lanes per key (8 elements each), 32 keys per pass; the heads of a kv group share the key chunk
                {
                    const int l16 = tid & 15, slot = tid >> 4;
                    constexpr int G = S::H / S::Hkv;
                    for (int kh = 0; kh < S::Hkv; ++kh) {
                        float qr[G][8];
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int e = 0; e < 8; ++e) qr[g][e] = qh[kh * G + g][8 * l16 + e];
                        for (int j0 = 0; j0 < ctx; j0 += TE_NT / 16) {
                            const int j = j0 + slot;
                            const int jc = j < ctx ? j : ctx - 1;
                            float kf[8];
                            if (jc == pos) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) kf[e] = knew[kh * S::D + 8 * l16 + e];
                            } else {
                                const bf16x8_t kk = *reinterpret_cast<const bf16x8_t*>(kc + (size_t)jc * (S::Hkv * S::D) + kh * S::D + 8 * l16);
#pragma unroll
                                for (int e = 0; e < 8; ++e) kf[e] = bf16_to_f32((bf16_t)kk[e]);
                            }
#pragma unroll
                            for (int g = 0; g < G; ++g) {
                                float dsum = 0.f;
#pragma unroll
                                for (int e = 0; e < 8; ++e) dsum += qr[g][e] * kf[e];
                                dsum += __shfl_xor(dsum, 1, 64); dsum += __shfl_xor(dsum, 2, 64);
                                dsum += __shfl_xor(dsum, 4, 64); dsum += __shfl_xor(dsum, 8, 64);
                                if (l16 == 0 && j < ctx) sc[kh * G + g][j] = dsum * scale;
                            }
                        }
                    }
                }
                __syncthreads();
                if (wave < S::H) {                                               // softmax of one head per wave
                    float m = -3.0e38f;
                    for (int j = lane; j < ctx; j += 64) m = fmaxf(m, sc[wave][j]);
                    m = wave_max(m);
                    float sum = 0.f;
                    for (int j = lane; j < ctx; j += 64) { const float e = expf(sc[wave][j] - m); sc[wave][j] = e; sum += e; }
                    sum = wave_sum(sum);
                    const float rinv = 1.0f / sum;
                    for (int j = lane; j < ctx; j += 64) sc[wave][j] *= rinv;
                }
                __syncthreads();
                {   // P.V: thread (head, d)
                    const int hh = tid / S::D, dd = tid % S::D, kh = hh / (S::H / S::Hkv);
                    float acc = 0.f;
                    const bf16_t* vcol = vc + kh * S::D + dd;
                    int j = 0;
                    for (; j + 4 <= pos; j += 4) {
                        const float v0 = bf16_to_f32(vcol[(size_t)(j + 0) * (S::Hkv * S::D)]), v1 = bf16_to_f32(vcol[(size_t)(j + 1) * (S::Hkv * S::D)]);
                        const float v2 = bf16_to_f32(vcol[(size_t)(j + 2) * (S::Hkv * S::D)]), v3 = bf16_to_f32(vcol[(size_t)(j + 3) * (S::Hkv * S::D)]);
                        acc += sc[hh][j] * v0; acc += sc[hh][j + 1] * v1; acc += sc[hh][j + 2] * v2; acc += sc[hh][j + 3] * v3;
                    }
                    for (; j < pos; ++j) acc += sc[hh][j] * bf16_to_f32(vcol[(size_t)j * (S::Hkv * S::D)]);
                    acc += sc[hh][pos] * vnew[kh * S::D + dd];
                    xb[hh * S::D + dd] = f32_to_bf16(acc);
                }
                __syncthreads();
                int nt[R_O];
#pragma unroll
                for (int r = 0; r < R_O; ++r) nt[r] = (w + r * W) < S::d / 16 ? w + r * W : -1;
                te_gemv<R_O, KPW_HD>(p.wo + (size_t)li * S::d * S::HD, S::HD / 32, nt, xb, red, wave, lane);
                __syncthreads();
                u64* buf = p.xbuf + (size_t)(edge & 1u) * TE_XG;
                if (tid < R_O * 4) {
                    const int r = tid >> 2, g = tid & 3;
                    if (nt[r] >= 0) {
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = hf[nt[r] * 16 + 4 * g + e] + bf16_round_f32(te_combine<R_O>(red, r, 4 * g + e));
                        __hip_atomic_store(buf + nt[r] * 4 + g, te_pack4(o[0], o[1], o[2], o[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (!te_meet(p, edge, W, &s_ok)) { if (tid == 0) *p.fail = 1u; return; }
                if (tid < S::d / 4) {
                    const u64 gq = __hip_atomic_load(buf + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int e = 0; e < 4; ++e) hf[4 * tid + e] = bf16_to_f32((bf16_t)(gq >> (16 * e)));
                }
                __syncthreads();
            }
            // ================= RMSNorm -> gate|up pairs -> SwiGLU -> edge 3
            {
                const bf16_t* wn = p.norms + (size_t)(2 * li + 1) * S::d;
                float v = tid < S::d ? hf[tid] : 0.f;
                const float ss = te_block_sum(v * v, s_red, wave, lane);
                const float inv = rsqrtf(ss / (float)S::d + p.eps);
                if (tid < S::d) xb[tid] = f32_to_bf16(bf16_to_f32(wn[tid]) * bf16_round_f32(v * inv));
                __syncthreads();
                int nt[R_GU];
#pragma unroll
                for (int r = 0; r < P_GU; ++r) {
                    const int pr = w + r * W;
                    nt[2 * r] = pr < S::ff / 16 ? 2 * pr : -1;
                    nt[2 * r + 1] = pr < S::ff / 16 ? 2 * pr + 1 : -1;
                }
                te_gemv<R_GU, KPW_D>(p.wgu + (size_t)li * 2 * S::ff * S::d, S::d / 32, nt, xb, red, wave, lane);
                __syncthreads();
                u64* buf = p.xbuf + (size_t)(edge & 1u) * TE_XG;
                if (tid < P_GU * 4) {
                    const int r = tid >> 2, g = tid & 3;
                    if (nt[2 * r] >= 0) {
                        float a[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float gt = bf16_round_f32(te_combine<R_GU>(red, 2 * r, 4 * g + e)), up = bf16_round_f32(te_combine<R_GU>(red, 2 * r + 1, 4 * g + e));
                            const float sg = bf16_round_f32(1.0f / (1.0f + expf(-gt)));
                            a[e] = bf16_round_f32(gt * sg) * up;
                        }
                        __hip_atomic_store(buf + (nt[2 * r] >> 1) * 4 + g, te_pack4(a[0], a[1], a[2], a[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (!te_meet(p, edge, W, &s_ok)) { if (tid == 0) *p.fail = 1u; return; }
                for (int gi = tid; gi < S::ff / 4; gi += TE_NT)
                    *reinterpret_cast<u64*>(xb + 4 * gi) = __hip_atomic_load(buf + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
            }
            // ================= down_proj slice, residual -> edge 4
            {
                int nt[R_O];
#pragma unroll
                for (int r = 0; r < R_O; ++r) nt[r] = (w + r * W) < S::d / 16 ? w + r * W : -1;
                te_gemv<R_O, KPW_FF>(p.wdown + (size_t)li * S::d * S::ff, S::ff / 32, nt, xb, red, wave, lane);
                __syncthreads();
                u64* buf = p.xbuf + (size_t)(edge & 1u) * TE_XG;
                if (tid < R_O * 4) {
                    const int r = tid >> 2, g = tid & 3;
                    if (nt[r] >= 0) {
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = hf[nt[r] * 16 + 4 * g + e] + bf16_round_f32(te_combine<R_O>(red, r, 4 * g + e));
                        __hip_atomic_store(buf + nt[r] * 4 + g, te_pack4(o[0], o[1], o[2], o[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (!te_meet(p, edge, W, &s_ok)) { if (tid == 0) *p.fail = 1u; return; }
                if (tid < S::d / 4) {
                    const u64 gq = __hip_atomic_load(buf + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int e = 0; e < 4; ++e) hf[4 * tid + e] = bf16_to_f32((bf16_t)(gq >> (16 * e)));
                }
                __syncthreads();
            }
        }
        // ================= final norm -> output projection slice -> arg-max -> edge 5
        {
            const bf16_t* wn = p.norms + (size_t)(2 * p.L) * S::d;
            float v = tid < S::d ? hf[tid] : 0.f;
            const float ss = te_block_sum(v * v, s_red, wave, lane);
            const float inv = rsqrtf(ss / (float)S::d + p.eps);
            if (tid < S::d) {
                const bf16_t xo = f32_to_bf16(bf16_to_f32(wn[tid]) * bf16_round_f32(v * inv));
                xb[tid] = xo;
                if (p.hidden_out && w == 0) p.hidden_out[(size_t)t * S::d + tid] = bf16_to_f32(xo);
            }
            __syncthreads();
            const int NTV = p.Vpad / 16;
            u64 cand = 0;
            // (the worker's tile rows in passes of R_HEAD: 16 rows at once need 128 registers of tiles in flight and spill)
            for (int pass = 0; pass * R_HEAD * W < NTV; ++pass) {
                int nt[R_HEAD];
#pragma unroll
                for (int r = 0; r < R_HEAD; ++r) nt[r] = (w + (pass * R_HEAD + r) * W) < NTV ? w + (pass * R_HEAD + r) * W : -1;
                te_gemv<R_HEAD, KPW_D>(p.head, S::d / 32, nt, xb, red, wave, lane);
                __syncthreads();
                if (tid < R_HEAD * 16) {
                    const int r = tid >> 4, i = tid & 15;
                    if (nt[r] >= 0) {
                        const int n = nt[r] * 16 + i;
                        const float lg = bf16_round_f32(te_combine<R_HEAD>(red, r, i));
                        if (n < p.V) {
                            if (p.logits_out) p.logits_out[(size_t)t * p.V + n] = lg;
                            const u64 c1 = ((u64)te_key(lg) << 32) | (u64)(0xffffffffu - (unsigned)n);      // highest logit, lowest id on ties
                            cand = c1 > cand ? c1 : cand;
                        }
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const u64 other = __shfl_xor(cand, o, 64); cand = other > cand ? other : cand; }
            if (lane == 0) s_cand[wave] = cand;
            __syncthreads();
            u64* buf = p.xbuf + (size_t)(edge & 1u) * TE_XG;
            if (tid == 0) {
                u64 best = 0;
#pragma unroll
                for (int q = 0; q < TE_NW; ++q) best = s_cand[q] > best ? s_cand[q] : best;
                __hip_atomic_store(buf + w, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!te_meet(p, edge, W, &s_ok)) { if (tid == 0) *p.fail = 1u; return; }
            u64 c2 = tid < W ? __hip_atomic_load(buf + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const u64 other = __shfl_xor(c2, o, 64); c2 = other > c2 ? other : c2; }
            if (lane == 0) s_cand[wave] = c2;
            __syncthreads();
            if (tid == 0) {
                u64 best = 0;
#pragma unroll
                for (int q = 0; q < TE_NW; ++q) best = s_cand[q] > best ? s_cand[q] : best;
                const int next = (int)(0xffffffffu - (unsigned)(best & 0xffffffffull));
                s_tok = next;
                if (w == 0) p.next_tokens[t] = next;
            }
            __syncthreads();
        }
    }
}
}   // namespace

// ---------------------------------------------------------------------------- host side (include/mi_speech_debug.h)
extern "C" mis_status mis_debug_token_engine(mis_tts* lm, const int32_t* prompt, int n_prompt, int n_new, int xcds, int32_t* next_tokens,
                                             float* logits_out, float* hidden_out, double* ms_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(lm && prompt && next_tokens && n_prompt >= 1 && n_new >= 0, MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(xcds == 1 || xcds == 2, MIS_ERR_INVALID_INPUT, "the engine is compiled for 1 or 2 XCDs");
    const TtsWeightsView v = tts_internal_weights(lm);
    using S = TeShape;
    MIS_REQUIRE(v.finalized, MIS_ERR_NOT_INITIALIZED, "model not finalized");
    MIS_REQUIRE(v.d == S::d && v.ff == S::ff && v.H == S::H && v.Hkv == S::Hkv && v.D == S::D && v.qk_norm && v.rope_plain && !v.quantised,
                MIS_ERR_INVALID_INPUT, "the token engine is compiled for Soprano-80M's widths (d 512, ffn 2304, 4 / 1 heads x 128, q/k norm, plain RoPE, bf16)");
    const int n_total = n_prompt + n_new;
    MIS_REQUIRE(n_total <= TE_CTX, MIS_ERR_INVALID_INPUT, "at most %d positions", TE_CTX);
    MIS_REQUIRE(S::ff / 4 <= TE_XG, MIS_ERR_INVALID_INPUT, "exchange buffer too small");
    HIP_CHECK(hipSetDevice(v.device));
    hipDeviceProp_t prop{};
    HIP_CHECK(hipGetDeviceProperties(&prop, v.device));
    const int grid = prop.multiProcessorCount / 8 * 8;
    MIS_REQUIRE(grid / 8 == 32, MIS_ERR_DEVICE, "the engine expects 32 compute units per XCD (found %d CUs)", prop.multiProcessorCount);
    const float* rc = nullptr; const float* rs = nullptr;
    tts_internal_rope_tables(lm, n_total, &rc, &rs);                     // (builds the tables for this context length)
    hipStream_t s = v.stream;
    const int W = 32 * xcds;
    std::vector<int32_t> hp(n_prompt);
    HIP_CHECK(hipMemcpy(hp.data(), prompt, (size_t)n_prompt * 4, hipMemcpyDefault));
    for (int t : hp) MIS_REQUIRE(t >= 0 && t < v.V, MIS_ERR_INVALID_INPUT, "prompt token %d outside the vocabulary", t);
    DevBuf<int32_t> d_prompt, d_next;
    DevBuf<float> d_logits, d_hidden;
    DevBuf<bf16_t> d_kv;
    DevBuf<u64> d_x;
    DevBuf<unsigned> d_sync;
    d_prompt.alloc(n_prompt); d_next.alloc(n_total);
    d_kv.alloc((size_t)W * v.L * 2 * TE_CTX * (S::Hkv * S::D));
    d_x.alloc(2 * TE_XG); d_sync.alloc(64);
    if (logits_out) d_logits.alloc((size_t)n_total * v.V);
    if (hidden_out) d_hidden.alloc((size_t)n_total * S::d);
    HIP_CHECK(hipMemcpyAsync(d_prompt.p, hp.data(), (size_t)n_prompt * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemsetAsync(d_sync.p, 0, 64 * sizeof(unsigned), s));
    HIP_CHECK(hipMemsetAsync(d_x.p, 0, 2 * TE_XG * sizeof(u64), s));
    HIP_CHECK(hipMemsetAsync(d_next.p, 0, (size_t)n_total * 4, s));
    TeParams p{};
    p.emb = v.emb; p.wqkv = v.wqkv; p.wo = v.wo; p.wgu = v.wgu; p.wdown = v.wdown; p.head = v.head; p.norms = v.norms; p.qknorm = v.qknorm;
    p.rope_cos = rc; p.rope_sin = rs; p.L = v.L; p.V = v.V; p.Vpad = v.Vpad; p.eps = v.eps;
    p.prompt = d_prompt.p; p.n_prompt = n_prompt; p.n_total = n_total; p.next_tokens = d_next.p;
    p.logits_out = logits_out ? d_logits.p : nullptr; p.hidden_out = hidden_out ? d_hidden.p : nullptr;
    p.kv = d_kv.p; p.xbuf = d_x.p; p.counter = d_sync.p; p.fail = d_sync.p + 32; p.xcds = xcds; p.spin = 1 << 20;
    const size_t pad = 64 * 1024;                                        // with the static arrays: more than half a CU's LDS -> one block per CU
    static bool attr_done[3] = {false, false, false};
    if (!attr_done[xcds]) {
        if (xcds == 1) HIP_CHECK(hipFuncSetAttribute((const void*)k_token_engine<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
        else HIP_CHECK(hipFuncSetAttribute((const void*)k_token_engine<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
        attr_done[xcds] = true;
    }
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, s));
    if (xcds == 1) hipLaunchKernelGGL(k_token_engine<1>, dim3(grid), dim3(TE_NT), pad, s, p);
    else hipLaunchKernelGGL(k_token_engine<2>, dim3(grid), dim3(TE_NT), pad, s, p);
    HIP_CHECK(hipEventRecord(e1, s));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    unsigned failed = 0;
    HIP_CHECK(hipMemcpy(&failed, d_sync.p + 32, 4, hipMemcpyDeviceToHost));
    MIS_REQUIRE(!failed, MIS_ERR_GENERATION_FAILED, "token engine: an edge timed out (its workers were not co-resident)");
    HIP_CHECK(hipMemcpy(next_tokens, d_next.p, (size_t)n_total * 4, hipMemcpyDeviceToHost));
    if (logits_out) HIP_CHECK(hipMemcpy(logits_out, d_logits.p, (size_t)n_total * v.V * 4, hipMemcpyDeviceToHost));
    if (hidden_out) HIP_CHECK(hipMemcpy(hidden_out, d_hidden.p, (size_t)n_total * S::d * 4, hipMemcpyDeviceToHost));
    if (ms_out) *ms_out = ms;
    MIS_API_END
}
