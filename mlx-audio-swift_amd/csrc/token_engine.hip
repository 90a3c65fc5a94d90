// token_engine.hip - a batch-1 decode engine for small LMs: ONE persistent launch runs every position of a request (the generate loop
// with its sampler included) on the compute units of a few XCDs.
//
// Why (BASELINE configs[1], Soprano-80M at batch 1): the launch chain spends ~125 graph nodes x 4.7 us on a
// token whose weights (160 MB) stream in 25 us - 0.03 of HBM.  At one row every op's output is a vector of 1-5 KB, so the per-op cost
// is the hand-off between compute units - a kernel boundary in the chain.  Measured before this was written
// (tools/token_engine_lab/probe.hip, profiles/r05/c2_xcd_stream_exchange_probe.json): an all-to-all edge (publish a slice, arrive on a
// counter, poll, gather the vector) among the 32 CUs of one XCD costs 1.65 us at 1 KB / 2.29 us at 4.6 KB, among 64 / 128 / 256 CUs
// 2.4 / 3.2 / 6.2 us; one XCD streams 1.28 TB/s out of the Infinity Cache (2.6 / 4.9 / 7.1 TB/s for 2 / 4 / 8).  Per token (69 edges,
// 160 MB): ~131 us of stream + ~126 us of edges on ONE XCD against 24 + 441 on eight - the chip-wide form is the launch chain's price
// again, few XCDs are 2-4x under it.  With the edge form this file ended up with (tagged granules, below) four XCDs are the optimum:
// 0.26 ms per position against the chain's 0.60 ("Batch-1 token engine": every version with its measurement).
//
// Structure.  Grid = one 512-thread block per CU; the blocks whose index mod 8 is below `xcds` are the W = 32 x xcds workers (observed
// placement: block b runs on XCD b mod 8 - used for speed only; every hand-off below is placement-independent), the others exit.
// Every worker holds the whole residual stream in LDS and owns a slice of the OUTPUT rows of every matrix:
//   per layer   RMSNorm (local) -> q|k|v slice                       -> edge 1 (Nqkv values)
//               q/k-norm, RoPE, attention - computed REDUNDANTLY by every worker over ONE shared K/V copy that all of them write with the
//               same bytes (no coherence protocol, no edge; ~50 KB of cache reads per layer at one row) -> o_proj slice, residual -> edge 2 (d)
//               RMSNorm (local) -> gate|up pairs, SwiGLU             -> edge 3 (ff)
//               down_proj slice, residual                            -> edge 4 (d)
//   per token   final norm (hidden tap) -> output-projection slice -> the token: arg-max (edge 5), arg-max behind the repetition penalty
//               (edges 5-6) or mis-sampler-v1 (edges 5-7: maximum, tile masses, token) -> next position's embedding
// Two programs share a block (te_matrix_role, te_vector_role): waves 4-7 hold the weight tiles - a GEMV slice runs on
// v_mfma_f32_16x16x32_bf16 with the engine's pre-packed tiles [N/16][K/32][64][8] as the A operand and the input vector as row 0 of B
// (15/16 of the matrix core idles: at one row the tile stream is the cost, not the math), each wave a quarter of the K range, tiles
// requested a phase ahead in pieces; waves 0-3 do the gathers, norms, attention (QK^T and P.V on the matrix core too), epilogues and
// edges.  An edge = self-validating 8-byte granules {two bf16 values, the edge's tag}: one agent-scope store by the producer, polled by
// the consumers (MI355X_MICROARCH.md, hand-off form R2).
// Rounding points are the oracle's (oracle/llama.py = MLX's bf16 graph): every primitive output rounded to bf16, float32 accumulation.
//
// In the product: mis_soprano_generate at batch 1 (csrc/soprano.hip; prompt through the launch chain's batched prefill, its K/V
// imported; falls back to the launch chain when the workers cannot be co-resident).  Compiled for Soprano-80M's widths (TeShape).
// Tests: tests/test_gpu_token_engine.py (oracle and launch-chain logits, sampler bit-exact on its own logits, hidden rows, stop id,
// long contexts, 1 / 2 / 4 / 8 XCDs), tests/test_gpu_soprano.py, tests/test_isa_cpu.py (register budget).
#include "common.h"
#include "kernels.h"
#include "sampler_math.h"
#include <type_traits>
#include <string.h>
#include <chrono>
#include <mutex>
#include <thread>

namespace {
typedef unsigned long long u64;

// the one shape compiled in: Soprano-80M's LM (SopranoConfig.swift:103-167; layer count and vocabulary stay run-time)
struct TeShape {
    static constexpr int d = 512, ff = 2304, H = 4, Hkv = 1, D = 128, Nqkv = (H + 2 * Hkv) * D, HD = H * D;
};
constexpr int TE_NT = 512;                     // threads per worker: TE_VW vector waves + TE_MW matrix waves
constexpr int TE_VW = 4, TE_MW = 4;
constexpr int TE_CTX = 1024;                   // positions per request (scores and probabilities in LDS); Soprano's default budget is 512 ids
constexpr int TE_HP = 2;                       // output-projection passes at most (vocabulary <= TE_HP x 8 x 16 x workers ids)
constexpr int TE_KPRE = 2, TE_VPRE = 4;        // key tiles per wave / 32-key value steps requested before the layer's first poll (128 positions)
constexpr int TE_XG = 2048;                    // granules per exchange buffer (two bf16 values + the edge's tag each)

struct TeParams {
    const bf16_t *emb, *wqkv, *wo, *wgu, *wdown, *head, *norms, *qknorm;
    const float *rope_cos, *rope_sin;
    int L, V, Vpad;
    float eps;
    const int32_t* prompt;
    int n_prompt, n_total;
    int t_start;                // first position the engine walks (the K/V of the positions before it were imported from the launch chain's prefill)
    int32_t* tok_dev;           // [n_total] device memory, -1 = not chosen yet: the id chosen after position t (one agent-scope store by worker 0)
    int32_t* next_tokens;       // [n_total] HOST-VISIBLE (pinned, coherent) copy, filled WHILE the launch runs by the relay block (te_relay): the
                                // host reads the ids from here for generateStream's .token events (Soprano.swift:877)
    const int* cancel;          // host-visible word or null, read by the relay block only
    int* cancel_dev;            // device word the relay forwards it to: worker 0 reads it once per position, the value travels with edge 5
                                // (every worker sees the SAME value at the same position) and a non-zero value ends the request like the stop id
    unsigned* relay_done;       // device word: worker 0 has left (everything it chose is in tok_dev)
    float* logits_out;          // [n_total][V] or null
    float* hidden_out;          // [n_total][d] or null (final-norm output: what Soprano's decoder consumes)
    bf16_t* kv;                 // the K/V copy [L][2][TE_CTX][Hkv*D] (every worker writes the same bytes, see te_vector_role)
    u64* xbuf;                  // [2][TE_XG]
    unsigned* fail;             // set when a poll ran out (workers not co-resident)
    int xcds, spin;
    // which positions get an output projection and a token: [head_from, head_until) - the laboratory asks for all of them, generate for
    // the last prompt position and every generated one but the last (whose hidden state is still wanted, not its successor)
    int head_from, head_until;
    // token choice: 0 = arg-max of the logits (laboratory form); 2 = arg-max behind the repetition penalty (generate at temperature 0);
    // 1 = "mis-sampler-v1" (oracle/sampler.py) behind the Soprano flavour's repetition penalty (float32, once
    // per occurrence among the last win_cap GENERATED ids, Soprano.swift:833-901), bit for bit what lm_sampler.hip computes
    int sample, win_cap;
    float temperature, penalty;
    u64 seed;
    long long row;              // global row index of the request (RNG key)
    int stop_id;                // a sampled id that ends the request (-1: none)
    int32_t* n_done;            // [2] out: positions processed, ids sampled
    u64* dbg;                   // diagnostics (MIS_TE_STAMPS=<position>): cycle stamps of worker 0 at the phase boundaries of layer 1 of that position
    int dbg_token;
};

__device__ __forceinline__ unsigned te_key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
// sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), result in every lane of the row: two quad permutes, half-row mirror, row
// mirror - four VALU-rate instructions.  (__shfl_xor compiles to ds_bpermute_b32, an LDS round trip of ~64 cycles each: the score loop's
// 128 of them per thread were 5-7 us per layer.)
__device__ __forceinline__ float te_row16_sum(float x) {
    int xi = __builtin_bit_cast(int, x);
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    xi = __builtin_bit_cast(int, x);
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    xi = __builtin_bit_cast(int, x);
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x141, 0xF, 0xF, true));     // row_half_mirror
    xi = __builtin_bit_cast(int, x);
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x140, 0xF, 0xF, true));     // row_mirror
    return x;
}
__device__ __forceinline__ float te_wave_max_dpp(float x) {
    auto step = [](float v, int tag) {
        const int vi = __builtin_bit_cast(int, v);
        int yi;
        switch (tag) {
            case 0: yi = __builtin_amdgcn_update_dpp(vi, vi, 0xB1, 0xF, 0xF, false); break;
            case 1: yi = __builtin_amdgcn_update_dpp(vi, vi, 0x4E, 0xF, 0xF, false); break;
            case 2: yi = __builtin_amdgcn_update_dpp(vi, vi, 0x141, 0xF, 0xF, false); break;
            default: yi = __builtin_amdgcn_update_dpp(vi, vi, 0x140, 0xF, 0xF, false); break;
        }
        return fmaxf(v, __builtin_bit_cast(float, yi));
    };
    x = step(x, 0); x = step(x, 1); x = step(x, 2); x = step(x, 3);
    const int vi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16)),
                r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// Block barrier that waits for LDS traffic only.  __syncthreads() carries a workgroup-scope fence, i.e. s_waitcnt vmcnt(0): it would make
// the matrix waves wait for every weight tile they have just requested for the NEXT phase.  Global-memory ordering is handled where it is
// needed (the publishers' own s_waitcnt vmcnt(0) before an edge).
__device__ __forceinline__ void te_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- matrix waves.  A worker's slice of y = W x is R tile rows (ids nt[r], -1 = none); each of the TE_MW matrix waves takes a quarter of
// the KT k-tiles of every row and holds them in registers: requested one phase AHEAD (right after the previous phase's MFMAs), so that the
// stream runs under the vector waves' epilogue, the edge and the gather.  Loads are unconditional on clamped addresses.
template <int R, int KPW>
struct TeTiles { bf16x8_t a[R][KPW]; bf16x8_t wn[KPW]; };
// Tiles FROM .. TO - 1 of the flattened list f = u R + r (all of them by default): a phase's tiles are requested in PIECES, one piece behind
// each barrier the matrix waves pass on their way to the phase (te_matrix_role) - a wave that requests 40 tiles at once stays in the issue
// loop for as long as the CU's memory queue is full (~4 us at one XCD's 40 GB/s per CU), and the vector waves wait for it at the next barrier.
template <int R, int KPW, bool NORM, int FROM = 0, int TO = R * KPW>
__device__ __forceinline__ void te_load(TeTiles<R, KPW>& T, const bf16_t* Wp, const int KT, const int (&nt)[R], const bf16_t* wnorm, const int mw,
                                        const int lane) {
    const int kt0 = mw * KT / TE_MW, kt1 = (mw + 1) * KT / TE_MW;
    const int klast = kt1 > kt0 ? kt1 - 1 : kt0;
    const unsigned voff = (unsigned)lane * 16u;                      // the only per-lane part of a tile address (scalar base + 32-bit offset)
#pragma unroll
    for (int u = 0; u < KPW; ++u) {
        int kk = kt0 + u;
        kk = kk > klast ? klast : kk;
        kk = kk >= KT ? KT - 1 : kk;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (u * R + r < FROM || u * R + r >= TO) continue;
            const int tile = nt[r] < 0 ? 0 : nt[r];
            const char* base = reinterpret_cast<const char*>(Wp) + ((size_t)tile * KT + kk) * 1024;        // wave-uniform
            T.a[r][u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(base + voff));
        }
        if (NORM && u * R >= FROM && u * R < TO) {                                                         // the norm weights of this wave's k range
            const char* nb = reinterpret_cast<const char*>(wnorm) + (size_t)kk * 64;
            T.wn[u] = *reinterpret_cast<const bf16x8_t*>(nb + (unsigned)(lane >> 4) * 16u);
        }
    }
}
// B fragments (row 0 of the 16-row operand): lanes with (lane & 15) == 0 hold x[32 kk + 8 (lane >> 4) ..+8], everything else is zero
template <int KPW>
__device__ __forceinline__ void te_xfrag_bf16(bf16x8_t (&xf)[KPW], const bf16_t* xb, const int KT, const int mw, const int lane) {
    const int kt0 = mw * KT / TE_MW, kt1 = (mw + 1) * KT / TE_MW;
    const int klast = kt1 > kt0 ? kt1 - 1 : kt0;
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
}
// the same from the float32 residual stream through RMSNorm: T(w * T(h * inv))
template <int KPW>
__device__ __forceinline__ void te_xfrag_norm(bf16x8_t (&xf)[KPW], const float* hf, const bf16x8_t (&wn)[KPW], const float inv, const int KT,
                                              const int mw, const int lane) {
    const int kt0 = mw * KT / TE_MW, kt1 = (mw + 1) * KT / TE_MW;
    const int klast = kt1 > kt0 ? kt1 - 1 : kt0;
    const bool row0 = (lane & 15) == 0;
#pragma unroll
    for (int u = 0; u < KPW; ++u) {
        int kk = kt0 + u;
        const bool live = kk < kt1;
        kk = kk > klast ? klast : kk;
        kk = kk >= KT ? KT - 1 : kk;
        const float* hp = hf + 32 * kk + 8 * (lane >> 4);
        const f32x4_t h0 = *reinterpret_cast<const f32x4_t*>(hp), h1 = *reinterpret_cast<const f32x4_t*>(hp + 4);
        bf16x8_t v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float hv = e < 4 ? h0[e] : h1[e - 4];
            v[e] = (short)f32_to_bf16(bf16_to_f32((bf16_t)wn[u][e]) * bf16_round_f32(hv * inv));
        }
        const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
        xf[u] = (row0 && live) ? v : zero;
    }
}
template <int R, int KPW>
__device__ __forceinline__ void te_mma(const TeTiles<R, KPW>& T, const bf16x8_t (&xf)[KPW], float* red, const int mw, const int lane) {
    constexpr int RB = R > 5 ? 5 : R;                                // tile rows per block of accumulators (gate|up: 10 rows = 2 blocks - with all
    const int g = lane >> 4;                                         //  ten live next to 176 registers of tiles the kernel spills)
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += RB) {
        f32x4_t acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KPW; ++u)
#pragma unroll
            for (int r = 0; r < RB; ++r)
                if (r0 + r < R) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T.a[r0 + r][u], xf[u], acc[r], 0, 0, 0);
        if ((lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < RB; ++r)
                if (r0 + r < R) *reinterpret_cast<f32x4_t*>(red + ((size_t)(mw * R + r0 + r) * 16 + 4 * g)) = acc[r];
        }
    }
    // the NEXT phase's tile requests follow in program order: left to the scheduler they are hoisted above these MFMAs, and two phases'
    // tiles (gate|up: 176 registers, down: 72) are live at once - spills into scratch, i.e. more traffic in the same vmcnt queue
    __builtin_amdgcn_sched_barrier(0);
}
// sum of the matrix waves' partials for element (r, i), fixed order
template <int R>
__device__ __forceinline__ float te_combine(const float* red, int r, int i) {
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < TE_MW; ++m) s += red[(size_t)(m * R + r) * 16 + i];
    return s;
}

// LDS of one worker
struct TeLds {
    float* hf;          // [d] residual stream (bf16 values)
    bf16_t* xb;         // [max(ff, H D)] attention output / activation: the plain GEMV inputs
    float* qkvf;        // [Nqkv]
    float* qh;          // [H][D]
    float *knew, *vnew; // [D]
    float* sc;          // [H][TE_CTX]
    bf16_t *ph, *pl;    // [H][TE_CTX] probabilities of the keys before this position, bf16 hi and lo
    bf16_t* stage;      // [R x 16] one epilogue value per thread before four of them are packed into a granule
    float* red;         // [TE_MW][R][16] partial sums of the matrix waves
    float* s_ss;        // [TE_VW] sum-of-squares partials of the residual stream
    u64* s_cand;        // [2][TE_VW] the waves' best candidates: own slice, then all workers'
    int *s_ok, *s_tok, *s_done, *s_cancel;
    u64* earr;          // [TE_HP][R_HEAD x 16] fixed-point masses of this worker's ids
    uint32_t* tsum32;   // [2 x tiles] the mass of every 16-id tile of the vocabulary, lo / hi words
    int* win;           // [64] + length: the repetition window (generated ids)
    u64* s_wtot;        // [TE_VW + 2] wave totals of the tile scan; the chosen tile and the draw's remainder inside it
};
template <int XCDS>
struct TeDims {
    using S = TeShape;
    static constexpr int W = 32 * XCDS;
    static constexpr int R_QKV = (S::Nqkv / 16 + W - 1) / W, R_O = (S::d / 16 + W - 1) / W, P_GU = (S::ff / 16 + W - 1) / W, R_GU = 2 * P_GU;
    static constexpr int R_HEAD = 8;                                 // tile rows of the output projection per pass
    static constexpr int KPW_D = (S::d / 32 + TE_MW - 1) / TE_MW, KPW_HD = (S::HD / 32 + TE_MW - 1) / TE_MW, KPW_FF = (S::ff / 32 + TE_MW - 1) / TE_MW;
    static constexpr int R_RED = R_GU > R_HEAD ? R_GU : R_HEAD;
};
// ---------------------------------------------------------------------------- the matrix waves' program.  Every te_sync() here has its
// partner at the same place of te_vector_role (the two programs are listed phase by phase in the same order); between barriers these
// waves only touch their weight tiles, the GEMV inputs in LDS and `red`.  Where the tiles of a phase are requested: as late as the
// registers demand (gate|up's 176 registers can only be filled once q|k|v's and before down's), and in PIECES sized to the vector waves'
// work behind each barrier - a wave stays in the issue loop while the CU's memory queue is full (one XCD streams 40 GB/s per CU: 24 KB
// per piece = 0.6 us), and everybody waits for it at the next barrier.
template <int XCDS>
__device__ __forceinline__ void te_matrix_role(const TeParams& p, const TeLds& L, const int w_in, const int mw_in, const int lane) {
    using S = TeShape;
    using Dm = TeDims<XCDS>;
    constexpr int W = Dm::W, R_QKV = Dm::R_QKV, R_O = Dm::R_O, P_GU = Dm::P_GU, R_GU = Dm::R_GU, R_HEAD = Dm::R_HEAD;
    constexpr int KPW_D = Dm::KPW_D, KPW_HD = Dm::KPW_HD, KPW_FF = Dm::KPW_FF;
    const int NTV = p.Vpad / 16;
    // This worker's tile rows, recomputed from an OPAQUE copy of the (scalar) worker and wave index wherever tiles are requested: every
    // tile address is loop-invariant, and hoisted out of the layer loop the ~100 of them (64-bit each) exhaust the scalar registers,
    // after which the compiler builds them in vector registers and the tile buffers spill.  A few SALU instructions per request instead.
#define TE_OPAQUE_IDS() int wq = w_in, mw = mw_in; asm volatile("" : "+s"(wq), "+s"(mw))
#define TE_ROWS_QKV(NT) int NT[R_QKV]; _Pragma("unroll") for (int r = 0; r < R_QKV; ++r) NT[r] = (wq + r * W) < S::Nqkv / 16 ? wq + r * W : -1
#define TE_ROWS_O(NT) int NT[R_O]; _Pragma("unroll") for (int r = 0; r < R_O; ++r) NT[r] = (wq + r * W) < S::d / 16 ? wq + r * W : -1
#define TE_ROWS_GU(NT) int NT[R_GU]; _Pragma("unroll") for (int r = 0; r < P_GU; ++r) { const int pr = wq + r * W;                       \
        NT[2 * r] = pr < S::ff / 16 ? 2 * pr : -1; NT[2 * r + 1] = pr < S::ff / 16 ? 2 * pr + 1 : -1; }
#define TE_ROWS_HEAD(NT, PASS) int NT[R_HEAD]; _Pragma("unroll") for (int r = 0; r < R_HEAD; ++r)                                      \
        NT[r] = (wq + ((PASS) * R_HEAD + r) * W) < NTV ? wq + ((PASS) * R_HEAD + r) * W : -1
    // tiles [N a / 40, N b / 40) of a phase's N
#define TE_CUT(N, a, b) ((N) * (a) / 40), ((N) * (b) / 40)
    constexpr int N_G = R_GU * KPW_D;
    TeTiles<R_QKV, KPW_D> tq;
    {
        TE_OPAQUE_IDS();
        TE_ROWS_QKV(nt);
        te_load<R_QKV, KPW_D, true>(tq, p.wqkv, S::d / 32, nt, p.norms, mw, lane);
    }
    for (int t = p.t_start; t < p.n_total; ++t) {
        te_sync();                                                   // token id
        if (*L.s_done) return;
        te_sync();                                                   // embedding row + sum of squares
        const bool do_head = t >= p.head_from && t < p.head_until;
        // (the layer loop is rotated by the last barrier of edge 4: the output projection's first tiles are requested AFTER the loop and
        // still ahead of the last layer's edge 4 - loaded inside the loop's last iteration they would be live across the whole loop, 144
        // registers that the gate|up tiles need)
        for (int li = 0; li < p.L; ++li) {
            TeTiles<R_O, KPW_HD> to;
            TeTiles<R_GU, KPW_D> tg;
            TeTiles<R_O, KPW_FF> td;
            const bf16_t* wg_l = p.wgu + (size_t)li * 2 * S::ff * S::d;
            const bf16_t* n2_l = p.norms + (size_t)(2 * li + 1) * S::d;
            if (li > 0) {
                te_sync();                                           // (previous layer) edge 4: residual stream gathered
                if (!*L.s_ok) return;
            }
            {   // q|k|v; then o_proj's tiles and the first quarter of gate|up's (the vector waves publish, wait for the slowest worker and gather)
                TE_OPAQUE_IDS();
                const float inv = rsqrtf(((L.s_ss[0] + L.s_ss[1]) + (L.s_ss[2] + L.s_ss[3])) / (float)S::d + p.eps);
                bf16x8_t xf[KPW_D];
                te_xfrag_norm<KPW_D>(xf, L.hf, tq.wn, inv, S::d / 32, mw, lane);
                te_mma<R_QKV, KPW_D>(tq, xf, L.red, mw, lane);
                te_sync();                                           // red ready
                TE_ROWS_O(nt);
                te_load<R_O, KPW_HD, false>(to, p.wo + (size_t)li * S::d * S::HD, S::HD / 32, nt, nullptr, mw, lane);
                TE_ROWS_GU(ng);
                te_load<R_GU, KPW_D, true, TE_CUT(N_G, 0, 10)>(tg, wg_l, S::d / 32, ng, n2_l, mw, lane);
                te_sync();                                           // edge 1: q|k|v gathered
                if (!*L.s_ok) return;
                te_load<R_GU, KPW_D, true, TE_CUT(N_G, 10, 16)>(tg, wg_l, S::d / 32, ng, n2_l, mw, lane);
                te_sync();                                           // q/k-norm + RoPE
                te_load<R_GU, KPW_D, true, TE_CUT(N_G, 16, 28)>(tg, wg_l, S::d / 32, ng, n2_l, mw, lane);
                te_sync();                                           // scores
                te_load<R_GU, KPW_D, true, TE_CUT(N_G, 28, 34)>(tg, wg_l, S::d / 32, ng, n2_l, mw, lane);
                te_sync();                                           // softmax
                te_load<R_GU, KPW_D, true, TE_CUT(N_G, 34, 40)>(tg, wg_l, S::d / 32, ng, n2_l, mw, lane);
                te_sync();                                           // attention output ready
            }
            {   // o_proj (gate|up's 176 registers of tiles are live: nothing more can be requested before its MFMAs)
                TE_OPAQUE_IDS();
                bf16x8_t xf[KPW_HD];
                te_xfrag_bf16<KPW_HD>(xf, L.xb, S::HD / 32, mw, lane);
                te_mma<R_O, KPW_HD>(to, xf, L.red, mw, lane);
                te_sync();                                           // red ready
            }
            te_sync();                                               // edge 2: residual stream gathered
            if (!*L.s_ok) return;
            {   // gate|up; then down_proj's tiles
                TE_OPAQUE_IDS();
                const float inv = rsqrtf(((L.s_ss[0] + L.s_ss[1]) + (L.s_ss[2] + L.s_ss[3])) / (float)S::d + p.eps);
                bf16x8_t xf[KPW_D];
                te_xfrag_norm<KPW_D>(xf, L.hf, tg.wn, inv, S::d / 32, mw, lane);
                te_mma<R_GU, KPW_D>(tg, xf, L.red, mw, lane);
                te_sync();                                           // red ready
                TE_ROWS_O(nt);
                te_load<R_O, KPW_FF, false>(td, p.wdown + (size_t)li * S::d * S::ff, S::ff / 32, nt, nullptr, mw, lane);
                te_sync();                                           // edge 3: activation gathered
                if (!*L.s_ok) return;
            }
            {   // down; then the next layer's q|k|v tiles
                TE_OPAQUE_IDS();
                bf16x8_t xf[KPW_FF];
                te_xfrag_bf16<KPW_FF>(xf, L.xb, S::ff / 32, mw, lane);
                te_mma<R_O, KPW_FF>(td, xf, L.red, mw, lane);
                te_sync();                                           // red ready
                if (li + 1 < p.L) {
                    TE_ROWS_QKV(nt);
                    te_load<R_QKV, KPW_D, true>(tq, p.wqkv + (size_t)(li + 1) * S::Nqkv * S::d, S::d / 32, nt, p.norms + (size_t)(2 * li + 2) * S::d, mw, lane);
                }
            }
        }
        if (!do_head) {   // a prompt position: straight to the next position's first tiles
            TE_OPAQUE_IDS();
            TE_ROWS_QKV(nt);
            te_load<R_QKV, KPW_D, true>(tq, p.wqkv, S::d / 32, nt, p.norms, mw, lane);
            te_sync();                                               // (last layer) edge 4: residual stream gathered
            if (!*L.s_ok) return;
        } else {   // output projection, passes of R_HEAD tile rows; the first pass's tiles behind the last layer's down_proj
            TeTiles<R_HEAD, KPW_D> th;
            int mw_h;
            {
                TE_OPAQUE_IDS();
                mw_h = mw;
                TE_ROWS_HEAD(nth, 0);
                te_load<R_HEAD, KPW_D, true>(th, p.head, S::d / 32, nth, p.norms + (size_t)(2 * p.L) * S::d, mw, lane);
                te_sync();                                           // (last layer) edge 4: residual stream gathered
                if (!*L.s_ok) return;
            }
            const float inv = rsqrtf(((L.s_ss[0] + L.s_ss[1]) + (L.s_ss[2] + L.s_ss[3])) / (float)S::d + p.eps);
            bf16x8_t xf[KPW_D];
            te_xfrag_norm<KPW_D>(xf, L.hf, th.wn, inv, S::d / 32, mw_h, lane);
            for (int pass = 0; pass * R_HEAD * W < NTV; ++pass) {
                TE_OPAQUE_IDS();
                te_mma<R_HEAD, KPW_D>(th, xf, L.red, mw, lane);
                te_sync();                                           // red ready
                if ((pass + 1) * R_HEAD * W < NTV) {
                    TE_ROWS_HEAD(nt2, pass + 1);
                    te_load<R_HEAD, KPW_D, false>(th, p.head, S::d / 32, nt2, nullptr, mw, lane);
                } else {
                    TE_ROWS_QKV(nt);
                    te_load<R_QKV, KPW_D, true>(tq, p.wqkv, S::d / 32, nt, p.norms, mw, lane);              // layer 0 of the next position
                }
                te_sync();                                           // red consumed
            }
            te_sync();                                               // candidates of the vector waves
            te_sync();                                               // edge 5: candidates (arg-max) / maxima (sampling) gathered
            if (!*L.s_ok) return;
            if (p.sample == 1) {
                te_sync();                                           // edge 6: tile masses gathered
                if (!*L.s_ok) return;
                te_sync(); te_sync();                                // scan of the tile masses: wave totals, the chosen tile
                te_sync();                                           // edge 7: the token
                if (!*L.s_ok) return;
            } else if (p.sample == 2) {
                te_sync();                                           // the waves' id candidates
                te_sync();                                           // edge 6: the id
                if (!*L.s_ok) return;
            }
        }
    }
}

// ---------------------------------------------------------------------------- the vector waves' program (256 threads)
// An EDGE (all-to-all hand-off of one op's output vector): every value travels in a self-validating 8-byte granule {two bf16 values, the
// edge's 32-bit tag}, written with ONE agent-scope store by its producer and polled with agent-scope loads by every consumer thread
// that needs it (MI355X_MICROARCH.md, hand-off form R2: "granule = one naturally aligned 8-byte {data, tag}") - no counter, no drain of
// the producer's stores, no barrier between publishing and gathering: the consumer's load that finds the tag IS the gather.  Tags count
// edges from 1 (buffers start zeroed); two buffers alternate - a worker can only publish edge e + 2 after it has gathered all of edge
// e + 1, which every worker publishes only after it has gathered edge e.  Polls are bounded: a time-out clears s_ok, and both programs
// leave at the next barrier.  (Round 5's first form - values, store drain, barrier, arrival counter, poll, barrier, gather - cost
// 3.7 + 1.9 us per edge inside the engine, three barriers of it shared with the matrix waves.)
template <int XCDS>
__device__ __forceinline__ void te_vector_role(const TeParams& p, const TeLds& L, const int w, const int tid) {
    using S = TeShape;
    using Dm = TeDims<XCDS>;
    constexpr int W = Dm::W, R_QKV = Dm::R_QKV, R_O = Dm::R_O, P_GU = Dm::P_GU, R_GU = Dm::R_GU, R_HEAD = Dm::R_HEAD;
    constexpr int VT = TE_VW * 64;
    const int wave = tid >> 6, lane = tid & 63;
    const int NTV = p.Vpad / 16;
    unsigned edge = 0;                                               // edges passed so far; the current edge's tag is edge + 1
    // ONE K/V copy for all workers, written by every one of them: the new row is computed redundantly from the same gathered q|k|v by
    // the same instructions, so all writers store identical bytes, and a worker only consumes positions it has itself written at an
    // earlier step.  (Private copies - 32 x 17 x 2 x ctx x 256 B - do not fit the XCD's 4 MB L2: every row came from the Infinity
    // Cache, ~2 us per round trip under the weight stream.)
    bf16_t* kv_mine = p.kv;
    const float scale = rsqrtf((float)S::D);
    float(*qh)[S::D] = reinterpret_cast<float(*)[S::D]>(L.qh);
    float(*sc)[TE_CTX] = reinterpret_cast<float(*)[TE_CTX]>(L.sc);
    auto granule = [&](const u64* buf, int gi, unsigned tag) -> uint32_t {         // poll granule gi until it carries `tag`; its two values
        u64 g = __hip_atomic_load(buf + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int it = 0;
        while ((unsigned)(g >> 32) != tag) {
            if (++it > p.spin) { *L.s_ok = 0; break; }
            __builtin_amdgcn_s_sleep(1);
            g = __hip_atomic_load(buf + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return (uint32_t)g;
    };
    // the same for the NG granules tid, tid + VT, ... of a vector of n granules: all first polls in flight together (a thread that polls
    // its granules one after the other pays a memory round trip for each: 5 for the activation vector)
    auto granules = [&](const u64* buf, int n, unsigned tag, auto&& NGc, auto&& sink) {
        constexpr int NG = std::remove_reference_t<decltype(NGc)>::value;
        u64 g[NG];
#pragma unroll
        for (int k = 0; k < NG; ++k) { const int gi = tid + k * VT; g[k] = __hip_atomic_load(buf + (gi < n ? gi : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int gi = tid + k * VT;
            if (gi < n) {
                uint32_t v = (uint32_t)g[k];
                if ((unsigned)(g[k] >> 32) != tag) v = granule(buf, gi, tag);
                sink(gi, v);
            }
        }
    };
    // One epilogue value per thread (tid < n_vals, value index tid = 16 r + i); the even thread of each pair sends both (through LDS: the
    // pair sits in one wave, whose own s_waitcnt orders the write before the read).  first_pair = granule index of value 16 r, or -1.
    auto publish2 = [&](u64* buf, int n_vals, float value, int first_pair, unsigned tag) {
        if (tid < n_vals) L.stage[tid] = f32_to_bf16(value);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (tid < n_vals && (tid & 1) == 0 && first_pair >= 0)
            __hip_atomic_store(buf + first_pair + ((tid & 15) >> 1), (u64)*reinterpret_cast<const uint32_t*>(L.stage + tid) | ((u64)tag << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // gather of the residual stream (d values = d / 2 granules, one per thread) + its sum of squares per wave
    auto gather_h = [&](const u64* buf, unsigned tag) {
        float ss = 0.f;
        if (tid < S::d / 2) {
            const uint32_t g2 = granule(buf, tid, tag);
            const float v0 = bf16_to_f32((bf16_t)(g2 & 0xffffu)), v1 = bf16_to_f32((bf16_t)(g2 >> 16));
            L.hf[2 * tid] = v0; L.hf[2 * tid + 1] = v1;
            ss = v0 * v0 + v1 * v1;
        }
        ss = wave_sum_dpp(ss);
        if (lane == 0) L.s_ss[wave] = ss;
    };
    // residual epilogue of o_proj / down_proj: this worker's R_O x 16 outputs, T(h + T(acc))
    auto publish_resid = [&](u64* buf, unsigned tag) {
        float v = 0.f;
        int gr = -1;
        if (tid < R_O * 16) {
            const int r = tid >> 4, i = tid & 15, nt = w + r * W;
            if (nt < S::d / 16) { v = L.hf[nt * 16 + i] + bf16_round_f32(te_combine<R_O>(L.red, r, i)); gr = nt * 8; }
        }
        publish2(buf, R_O * 16, v, gr, tag);
    };
#define TE_STAMP(i) do { if (p.dbg && w == 0 && tid == 0 && t == p.dbg_token && li == 1) p.dbg[i] = __builtin_readcyclecounter(); } while (0)
#define TE_EDGE_BUF() (p.xbuf + (size_t)(edge & 1u) * TE_XG)
    int t_last = p.t_start - 1, n_sampled = 0;
    for (int t = p.t_start; t < p.n_total; ++t) {
        if (tid == 0 && t < p.n_prompt) *L.s_tok = p.prompt[t];
        // (requested here, consumed at edge 5; the relay block keeps the device word equal to the host's)
        uint32_t cancel_word = 0u;
        if (p.cancel && w == 0 && tid == 0) cancel_word = (uint32_t)__hip_atomic_load(p.cancel_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        te_sync();                                                   // token id
        if (*L.s_done) break;
        t_last = t;
        const bool do_head = t >= p.head_from && t < p.head_until;
        {
            const int tok = *L.s_tok;
            float ss = 0.f;
            if (tid < S::d / 2) {
                const uint32_t g2 = *reinterpret_cast<const uint32_t*>(p.emb + (size_t)tok * S::d + 2 * tid);
                const float v0 = bf16_to_f32((bf16_t)(g2 & 0xffffu)), v1 = bf16_to_f32((bf16_t)(g2 >> 16));
                L.hf[2 * tid] = v0; L.hf[2 * tid + 1] = v1;
                ss = v0 * v0 + v1 * v1;
            }
            ss = wave_sum_dpp(ss);
            if (lane == 0) L.s_ss[wave] = ss;
        }
        te_sync();                                                   // embedding row + sum of squares
        const float rope_c = p.rope_cos[(size_t)t * (S::D / 2) + lane], rope_s = p.rope_sin[(size_t)t * (S::D / 2) + lane];     // this position's row
        for (int li = 0; li < p.L; ++li) {
            bf16_t nw_q[2], nw_k[2];
            bf16x8_t kpre[TE_KPRE][S::D / 32], vpre[TE_VPRE][2];
            // ================= q|k|v slice -> edge 1
            TE_STAMP(0);
            te_sync();                                               // red ready
            TE_STAMP(1);
            {
                u64* buf = TE_EDGE_BUF();
                const unsigned tag = ++edge;
                float v = 0.f;
                int gr = -1;
                if (tid < R_QKV * 16) {
                    const int r = tid >> 4, i = tid & 15, nt = w + r * W;
                    if (nt < S::Nqkv / 16) { v = te_combine<R_QKV>(L.red, r, i); gr = nt * 8; }
                }
                publish2(buf, R_QKV * 16, v, gr, tag);
                TE_STAMP(2);
                // everything the attention phase reads from memory that does not depend on this layer's q|k|v, requested BEFORE the poll:
                // the q/k-norm weights, this wave's first TE_KPRE key tiles, the first TE_VPRE 32-key steps of its two value tiles
                nw_q[0] = p.qknorm[(size_t)(2 * li) * S::D + lane]; nw_q[1] = p.qknorm[(size_t)(2 * li) * S::D + lane + 64];
                nw_k[0] = p.qknorm[(size_t)(2 * li + 1) * S::D + lane]; nw_k[1] = p.qknorm[(size_t)(2 * li + 1) * S::D + lane + 64];
                {
                    const int i16 = lane & 15, q4 = lane >> 4, pos = t;
                    const bf16_t* kcl = kv_mine + ((size_t)li * 2 + 0) * TE_CTX * S::D;
                    const bf16_t* vcl = kv_mine + ((size_t)li * 2 + 1) * TE_CTX * S::D;
#pragma unroll
                    for (int k = 0; k < TE_KPRE; ++k) {
                        int row = 16 * (wave + TE_VW * k) + i16;
                        row = row < pos ? row : (pos > 0 ? pos - 1 : 0);
#pragma unroll
                        for (int ds = 0; ds < S::D / 32; ++ds)
                            kpre[k][ds] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(kcl + (size_t)row * S::D + 8 * q4 + 32 * ds));
                    }
#pragma unroll
                    for (int k = 0; k < TE_VPRE; ++k)
#pragma unroll
                        for (int n = 0; n < 2; ++n)   // (past L1: a value line holds 64 positions, the newest written by this CU a step ago)
                            vpre[k][n] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(vcl + (size_t)(16 * (2 * wave + n) + i16) * TE_CTX + 32 * k + 8 * q4));
                }
                granules(buf, S::Nqkv / 2, tag, std::integral_constant<int, (S::Nqkv / 2 + VT - 1) / VT>{}, [&](int gi, uint32_t g2) {
                    L.qkvf[2 * gi] = bf16_to_f32((bf16_t)(g2 & 0xffffu)); L.qkvf[2 * gi + 1] = bf16_to_f32((bf16_t)(g2 >> 16));
                });
                TE_STAMP(3);
            }
            te_sync();                                               // edge 1: q|k|v gathered
            if (!*L.s_ok) { if (tid == 0) __hip_atomic_store(p.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
            TE_STAMP(4);
            // ================= q/k-norm, RoPE, attention (redundant on every worker)
            bf16_t* kc = kv_mine + ((size_t)li * 2 + 0) * TE_CTX * S::D;
            bf16_t* vc = kv_mine + ((size_t)li * 2 + 1) * TE_CTX * S::D;
            const int pos = t, ctx = t + 1;
            {   // wave v: q head v; wave 0 also the key, wave 1 the value.  Lane holds elements lane, lane + 64 (RoPE partners)
                const float c = rope_c, sn = rope_s;
                for (int which = 0; which < 2; ++which) {
                    if (which == 1 && wave != 0) break;
                    const float* src = L.qkvf + (which ? S::HD : wave * S::D);
                    const float x1 = src[lane], x2 = src[lane + 64];
                    const float ss = wave_sum_dpp(x1 * x1 + x2 * x2);
                    const float inv = rsqrtf(ss / (float)S::D + p.eps);
                    const float y1 = bf16_round_f32(bf16_to_f32(which ? nw_k[0] : nw_q[0]) * bf16_round_f32(x1 * inv));
                    const float y2 = bf16_round_f32(bf16_to_f32(which ? nw_k[1] : nw_q[1]) * bf16_round_f32(x2 * inv));
                    const float o1 = bf16_round_f32(y1 * c - y2 * sn), o2 = bf16_round_f32(y1 * sn + y2 * c);
                    if (which) {
                        L.knew[lane] = o1; L.knew[lane + 64] = o2;
                        kc[(size_t)pos * S::D + lane] = f32_to_bf16(o1);
                        kc[(size_t)pos * S::D + lane + 64] = f32_to_bf16(o2);
                    } else {
                        qh[wave][lane] = o1; qh[wave][lane + 64] = o2;
                    }
                }
                if (wave == 1) {                                                 // values: kept TRANSPOSED [d][position] (the P.V MFMA's A operand)
                    const float v1 = L.qkvf[S::HD + S::D + lane], v2 = L.qkvf[S::HD + S::D + lane + 64];
                    L.vnew[lane] = v1; L.vnew[lane + 64] = v2;
                    vc[(size_t)lane * TE_CTX + pos] = f32_to_bf16(v1);
                    vc[(size_t)(lane + 64) * TE_CTX + pos] = f32_to_bf16(v2);
                }
            }
            te_sync();
            TE_STAMP(5);
            {   // scores on the matrix core: D[key][head] = sum_d K[key][d] q[head][d] - A = 16 keys x 32 d straight from the row-major key
                // cache (16 B per lane), B = q^T with the four heads in columns 0..3.  Wave v takes the key tiles v, v + 4, ...; only keys
                // BEFORE this position come from memory (the new key is in LDS: its row is still on its way to the cache).  (The VALU form -
                // 16 lanes per key, DPP reductions - was 4.5 us per layer at 81 keys.)
                const int i16 = lane & 15, q4 = lane >> 4;
                bf16x8_t qf[S::D / 32];
#pragma unroll
                for (int ds = 0; ds < S::D / 32; ++ds) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) qf[ds][e] = i16 < S::H ? (short)f32_to_bf16(qh[i16 < S::H ? i16 : 0][32 * ds + 8 * q4 + e]) : (short)0;
                }
                const int n_kt = (pos + 15) >> 4;
                auto score_tile = [&](int kt, const bf16x8_t (&ka)[S::D / 32]) {
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ds = 0; ds < S::D / 32; ++ds) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[ds], qf[ds], acc, 0, 0, 0);
                    if (i16 < S::H) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int j = 16 * kt + 4 * q4 + r;
                            if (j < pos) sc[i16][j] = acc[r] * scale;
                        }
                    }
                };
#pragma unroll
                for (int k = 0; k < TE_KPRE; ++k)
                    if (wave + TE_VW * k < n_kt) score_tile(wave + TE_VW * k, kpre[k]);              // requested ahead of edge 1's poll
                for (int kt = wave + TE_VW * TE_KPRE; kt < n_kt; kt += TE_VW) {                        // longer contexts: the tiles behind them
                    int row = 16 * kt + i16;
                    row = row < pos ? row : pos - 1;
                    const bf16_t* kr = kc + (size_t)row * S::D + 8 * q4;
                    bf16x8_t ka[S::D / 32];
#pragma unroll
                    for (int ds = 0; ds < S::D / 32; ++ds) ka[ds] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(kr + 32 * ds));
                    score_tile(kt, ka);
                }
                {   // the new key: head `wave`
                    const float dsum = wave_sum_dpp(qh[wave][lane] * L.knew[lane] + qh[wave][lane + 64] * L.knew[lane + 64]);
                    if (lane == 0) sc[wave][pos] = dsum * scale;
                }
            }
            te_sync();
            TE_STAMP(6);
            {   // softmax of head `wave`; the probabilities of the keys before this position as bf16 hi + lo (the pair keeps float32 accuracy
                // through the bf16 MFMA), zero up to the next multiple of 32 keys; the new key's probability stays float32 (sc[head][pos])
                float m = -3.0e38f;
                for (int j = lane; j < ctx; j += 64) m = fmaxf(m, sc[wave][j]);
                m = te_wave_max_dpp(m);
                float sum = 0.f;
                for (int j = lane; j < ctx; j += 64) { const float e = expf(sc[wave][j] - m); sc[wave][j] = e; sum += e; }
                sum = wave_sum_dpp(sum);
                const float rinv = 1.0f / sum;
                const int pend = (pos + 31) & ~31;
                for (int j = lane; j < pend || j < ctx; j += 64) {
                    const float pj = j < ctx ? sc[wave][j] * rinv : 0.f;
                    if (j < ctx) sc[wave][j] = pj;
                    if (j < pend) {
                        const float pm = j < pos ? pj : 0.f;
                        const bf16_t hi = f32_to_bf16(pm);
                        L.ph[wave * TE_CTX + j] = hi;
                        L.pl[wave * TE_CTX + j] = f32_to_bf16(pm - bf16_to_f32(hi));
                    }
                }
            }
            te_sync();
            {   // P.V on the matrix core: D[d][head] = sum_key Vt[d][key] P[head][key] - A = 16 d x 32 keys from the transposed value cache,
                // B = P^T (hi, then lo).  Wave v owns the d tiles 2 v, 2 v + 1 over ALL keys (no cross-wave sum); the new key joins in the
                // epilogue from LDS.
                const int i16 = lane & 15, q4 = lane >> 4;
                const int n_k32 = (pos + 31) >> 5;
                f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                auto pv_step = [&](int kt, const bf16x8_t (&va)[2]) {
                    const bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
                    const bf16x8_t bh = i16 < S::H ? *reinterpret_cast<const bf16x8_t*>(L.ph + (i16 < S::H ? i16 : 0) * TE_CTX + 32 * kt + 8 * q4) : z;
                    const bf16x8_t bl = i16 < S::H ? *reinterpret_cast<const bf16x8_t*>(L.pl + (i16 < S::H ? i16 : 0) * TE_CTX + 32 * kt + 8 * q4) : z;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[n], bh, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[n], bl, acc[n], 0, 0, 0);
                    }
                };
#pragma unroll
                for (int k = 0; k < TE_VPRE; ++k)
                    if (k < n_k32) pv_step(k, vpre[k]);                                              // requested ahead of edge 1's poll
                for (int kt = TE_VPRE; kt < n_k32; ++kt) {
                    bf16x8_t va[2];
#pragma unroll
                    for (int n = 0; n < 2; ++n)       // (past L1: a value line holds 64 positions, the newest written by this CU a step ago)
                        va[n] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(vc + (size_t)(16 * (2 * wave + n) + i16) * TE_CTX + 32 * kt + 8 * q4));
                    pv_step(kt, va);
                }
                if (i16 < S::H) {
                    const float pn = sc[i16][pos];
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int d0 = 16 * (2 * wave + n) + 4 * q4;
                        bf16_t o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = f32_to_bf16(acc[n][r] + pn * L.vnew[d0 + r]);
                        *reinterpret_cast<u64*>(L.xb + i16 * S::D + d0) = (u64)o[0] | ((u64)o[1] << 16) | ((u64)o[2] << 32) | ((u64)o[3] << 48);
                    }
                }
            }
            te_sync();                                               // attention output ready
            TE_STAMP(7);
            // ================= o_proj slice, residual -> edge 2
            te_sync();                                               // red ready
            TE_STAMP(8);
            {
                u64* buf = TE_EDGE_BUF();
                const unsigned tag = ++edge;
                publish_resid(buf, tag);
                TE_STAMP(9);
                gather_h(buf, tag);
                TE_STAMP(10);
            }
            te_sync();                                               // edge 2: residual stream gathered
            if (!*L.s_ok) { if (tid == 0) __hip_atomic_store(p.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
            TE_STAMP(11);
            // ================= gate|up pairs -> SwiGLU -> edge 3
            te_sync();                                               // red ready
            TE_STAMP(12);
            {
                u64* buf = TE_EDGE_BUF();
                const unsigned tag = ++edge;
                float v = 0.f;
                int gr = -1;
                if (tid < P_GU * 16) {
                    const int r = tid >> 4, i = tid & 15, pr = w + r * W;
                    if (pr < S::ff / 16) {
                        const float gt = bf16_round_f32(te_combine<R_GU>(L.red, 2 * r, i)), up = bf16_round_f32(te_combine<R_GU>(L.red, 2 * r + 1, i));
                        const float sg = bf16_round_f32(1.0f / (1.0f + expf(-gt)));
                        v = bf16_round_f32(gt * sg) * up;
                        gr = pr * 8;
                    }
                }
                publish2(buf, P_GU * 16, v, gr, tag);
                TE_STAMP(13);
                granules(buf, S::ff / 2, tag, std::integral_constant<int, (S::ff / 2 + VT - 1) / VT>{},
                         [&](int gi, uint32_t g2) { *reinterpret_cast<uint32_t*>(L.xb + 2 * gi) = g2; });
                TE_STAMP(14);
            }
            te_sync();                                               // edge 3: activation gathered
            if (!*L.s_ok) { if (tid == 0) __hip_atomic_store(p.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
            TE_STAMP(15);
            // ================= down_proj slice, residual -> edge 4
            te_sync();                                               // red ready
            TE_STAMP(16);
            {
                u64* buf = TE_EDGE_BUF();
                const unsigned tag = ++edge;
                publish_resid(buf, tag);
                TE_STAMP(17);
                gather_h(buf, tag);
                TE_STAMP(18);
            }
            te_sync();                                               // edge 4: residual stream gathered
            if (!*L.s_ok) { if (tid == 0) __hip_atomic_store(p.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
            TE_STAMP(19);
        }
        // ================= final norm (hidden tap) -> output projection slice -> token -> edges 5 (.. 7)
        if (p.hidden_out && w == 0 && t >= p.head_from && tid < S::d / 4) {
            const float inv = rsqrtf(((L.s_ss[0] + L.s_ss[1]) + (L.s_ss[2] + L.s_ss[3])) / (float)S::d + p.eps);
            const bf16_t* wn = p.norms + (size_t)(2 * p.L) * S::d;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                p.hidden_out[(size_t)(t - p.head_from) * S::d + 4 * tid + e] = bf16_round_f32(bf16_to_f32(wn[4 * tid + e]) * bf16_round_f32(L.hf[4 * tid + e] * inv));
        }
        if (do_head) {
            // arg-max candidate = (16-bit order-preserving key of the bf16 logit) << 16 | (0xffff - id): highest logit, lowest id on ties;
            // sampling candidate = the 32-bit key of the penalised float32 logit (only the maximum travels)
            uint32_t cand = 0;
            float lpen[TE_HP];
            int own_n[TE_HP];
#pragma unroll
            for (int k = 0; k < TE_HP; ++k) { lpen[k] = 0.f; own_n[k] = -1; }
#pragma unroll
            for (int pass = 0; pass < TE_HP; ++pass) {
                if (pass * R_HEAD * W >= NTV) break;
                te_sync();                                           // red ready
                if (tid < R_HEAD * 16) {
                    const int r = tid >> 4, i = tid & 15, nt = w + (pass * R_HEAD + r) * W;
                    if (nt < NTV) {
                        const int n = nt * 16 + i;
                        const float lg = bf16_round_f32(te_combine<R_HEAD>(L.red, r, i));
                        if (n < p.V) {
                            if (p.logits_out) p.logits_out[(size_t)(t - p.head_from) * p.V + n] = lg;
                            if (p.sample) {
                                // Soprano applyRepetitionPenalty (Soprano.swift:888-901): float32, once PER OCCURRENCE in the window
                                float v = lg;
                                if (p.penalty > 0.0f && p.penalty != 1.0f) {
                                    const int wl = L.win[64];
                                    int mult = 0;
                                    for (int j = 0; j < wl; ++j) mult += (L.win[j] == n);
                                    for (int k = 0; k < mult; ++k) v = (v > 0.0f) ? __fdiv_rn(v, p.penalty) : v * p.penalty;
                                }
                                lpen[pass] = v; own_n[pass] = n;
                                const uint32_t c1 = te_key(v);
                                cand = c1 > cand ? c1 : cand;
                            } else {
                                const uint32_t c1 = (te_key(lg) & 0xffff0000u) | (0xffffu - (unsigned)n);
                                cand = c1 > cand ? c1 : cand;
                            }
                        }
                    }
                }
                te_sync();                                           // red consumed
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const uint32_t other = __shfl_xor(cand, o, 64); cand = other > cand ? other : cand; }
            if (lane == 0) L.s_cand[wave] = cand;
            te_sync();                                               // candidates of the vector waves
            uint32_t best_all = 0;
            {   // edge 5: every worker's candidate to every worker
                u64* buf = TE_EDGE_BUF();
                const unsigned tag = ++edge;
                if (tid == 0) {
                    uint32_t best = 0;
#pragma unroll
                    for (int q = 0; q < TE_VW; ++q) best = (uint32_t)L.s_cand[q] > best ? (uint32_t)L.s_cand[q] : best;
                    __hip_atomic_store(buf + w, (u64)best | ((u64)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (p.cancel && w == 0)                          // granule W of this edge: the cancel word as worker 0 read it
                        __hip_atomic_store(buf + W, (u64)cancel_word | ((u64)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                uint32_t c2 = tid < W ? granule(buf, tid, tag) : 0u;
                if (p.cancel) {                                      // polled by an idle thread where there is one (W < 256), else by thread 0 behind its own
                    constexpr int CT = W < VT ? W : 0;
                    if (tid == CT && granule(buf, W, tag) != 0u) *L.s_cancel = 1;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const uint32_t other = __shfl_xor(c2, o, 64); c2 = other > c2 ? other : c2; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) L.s_cand[TE_VW + wave] = c2;
            }
            te_sync();                                               // edge 5 gathered
            if (!*L.s_ok) { if (tid == 0) __hip_atomic_store(p.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
#pragma unroll
            for (int q = 0; q < TE_VW; ++q) best_all = (uint32_t)L.s_cand[TE_VW + q] > best_all ? (uint32_t)L.s_cand[TE_VW + q] : best_all;
            int next;
            if (!p.sample) {
                next = (int)(0xffffu - (best_all & 0xffffu));
            } else if (p.sample == 2) {
                // arg-max of the PENALISED float32 logits (temperature 0 in the generate form): the maximum's 32-bit key is known to
                // everybody; the lowest id that holds it travels in a second edge
                uint32_t c3 = 0;
#pragma unroll
                for (int pass = 0; pass < TE_HP; ++pass)
                    if (own_n[pass] >= 0 && te_key(lpen[pass]) == best_all) { const uint32_t c1 = 0x10000u | (0xffffu - (unsigned)own_n[pass]); c3 = c1 > c3 ? c1 : c3; }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const uint32_t other = __shfl_xor(c3, o, 64); c3 = other > c3 ? other : c3; }
                if (lane == 0) L.s_cand[wave] = c3;
                te_sync();
                u64* buf = TE_EDGE_BUF();
                const unsigned tag = ++edge;
                if (tid == 0) {
                    uint32_t best = 0;
#pragma unroll
                    for (int q = 0; q < TE_VW; ++q) best = (uint32_t)L.s_cand[q] > best ? (uint32_t)L.s_cand[q] : best;
                    __hip_atomic_store(buf + w, (u64)best | ((u64)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                uint32_t c2 = tid < W ? granule(buf, tid, tag) : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const uint32_t other = __shfl_xor(c2, o, 64); c2 = other > c2 ? other : c2; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) L.s_cand[TE_VW + wave] = c2;
                te_sync();                                           // edge 6: the id
                if (!*L.s_ok) { if (tid == 0) __hip_atomic_store(p.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                uint32_t b2 = 0;
#pragma unroll
                for (int q = 0; q < TE_VW; ++q) b2 = (uint32_t)L.s_cand[TE_VW + q] > b2 ? (uint32_t)L.s_cand[TE_VW + q] : b2;
                next = (int)(0xffffu - (b2 & 0xffffu));
            } else {
                // ---- mis-sampler-v1 over the whole vocabulary: x = fdiv(l, T), e = det_exp(min(x - max x, 0)), E = trunc(e 2^40); the draw
                // r = mulhi64(rand64(seed, row, step), sum E) picks the first id, in id order, whose running sum of E exceeds r.  Ids are dealt
                // to the workers in tiles of 16, so the running sum is taken over TILES first (edge 6: every tile's mass to everybody; one block
                // scan), then inside the chosen tile by its owner (edge 7: the token to everybody).
                const float xmax = __fdiv_rn(__uint_as_float((best_all & 0x80000000u) ? (best_all & 0x7fffffffu) : ~best_all), p.temperature);
                u64* buf = TE_EDGE_BUF();
                const unsigned tag = ++edge;
#pragma unroll
                for (int pass = 0; pass < TE_HP; ++pass) {
                    if (pass * R_HEAD * W >= NTV) break;
                    u64 E = 0;
                    if (tid < R_HEAD * 16 && own_n[pass] >= 0) {
                        const float x = __fdiv_rn(lpen[pass], p.temperature);
                        E = (u64)(det_exp_dev(fminf(x - xmax, 0.0f)) * E_SCALE);
                    }
                    if (tid < R_HEAD * 16) L.earr[pass * (R_HEAD * 16) + tid] = E;
                    // tile mass = sum over the 16 lanes of a row: E < 2^41 split into two 21-bit halves, each summed in 32 bits on the DPP network
                    uint32_t a = (uint32_t)(E & 0x1fffffu), bb = (uint32_t)(E >> 21);
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); bb += __shfl_xor(bb, o, 64); }
                    const u64 tile_mass = (u64)a + ((u64)bb << 21);
                    if (tid < R_HEAD * 16 && (tid & 15) == 0) {
                        const int nt = w + (pass * R_HEAD + (tid >> 4)) * W;
                        if (nt < NTV) {
                            __hip_atomic_store(buf + 2 * nt, (u64)(uint32_t)tile_mass | ((u64)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(buf + 2 * nt + 1, (u64)(uint32_t)(tile_mass >> 32) | ((u64)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
                granules(buf, 2 * NTV, tag, std::integral_constant<int, TE_XG / VT>{}, [&](int gi, uint32_t g2) { L.tsum32[gi] = g2; });
                te_sync();                                           // edge 6: tile masses gathered
                if (!*L.s_ok) { if (tid == 0) __hip_atomic_store(p.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                // scan over the tiles: thread -> tiles 2 tid, 2 tid + 1
                const int t0i = 2 * tid, t1i = 2 * tid + 1;
                const u64 m0 = t0i < NTV ? ((u64)L.tsum32[2 * t0i] | ((u64)L.tsum32[2 * t0i + 1] << 32)) : 0;
                const u64 m1 = t1i < NTV ? ((u64)L.tsum32[2 * t1i] | ((u64)L.tsum32[2 * t1i + 1] << 32)) : 0;
                u64 incl = m0 + m1;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t ulo = __shfl_up((uint32_t)incl, o, 64), uhi = __shfl_up((uint32_t)(incl >> 32), o, 64);
                    if (lane >= o) incl += (u64)ulo | ((u64)uhi << 32);
                }
                if (lane == 63) L.s_wtot[wave] = incl;
                te_sync();
                u64 base = 0, Z = 0;
#pragma unroll
                for (int q = 0; q < TE_VW; ++q) { if (q < wave) base += L.s_wtot[q]; Z += L.s_wtot[q]; }
                const int step = t - (p.n_prompt - 1);
                const u64 rnd = mis_splitmix64(mis_splitmix64(p.seed ^ (0xD1B54A32D192ED03ull * (u64)(p.row + 1))) + (u64)step);
                const u64 r = __umul64hi(rnd, Z);
                const u64 excl = base + incl - (m0 + m1);
                if (r >= excl && r < excl + m0) { L.s_wtot[TE_VW] = (u64)t0i; L.s_wtot[TE_VW + 1] = r - excl; }
                else if (r >= excl + m0 && r < excl + m0 + m1) { L.s_wtot[TE_VW] = (u64)t1i; L.s_wtot[TE_VW + 1] = r - excl - m0; }
                te_sync();
                const int tile = (int)L.s_wtot[TE_VW];
                const u64 rin = L.s_wtot[TE_VW + 1];
                u64* bufc = TE_EDGE_BUF();
                const unsigned tagc = ++edge;
                if (tile % W == w && tid == 0) {                     // the owner: inside the tile in id order
                    const int slot = (tile - w) / W, pass = slot / R_HEAD, rr = slot % R_HEAD;
                    u64 run = 0;
                    int tok = 16 * tile + 15;
                    for (int i = 0; i < 16; ++i) {
                        run += L.earr[pass * (R_HEAD * 16) + rr * 16 + i];
                        if (run > rin) { tok = 16 * tile + i; break; }
                    }
                    __hip_atomic_store(bufc, (u64)(uint32_t)tok | ((u64)tagc << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (tid == 0) L.s_cand[0] = granule(bufc, 0, tagc);
                te_sync();                                           // edge 7: the token
                if (!*L.s_ok) { if (tid == 0) __hip_atomic_store(p.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                next = (int)(uint32_t)L.s_cand[0];
            }
            n_sampled += 1;
            if (tid == 0) {
                if (t + 1 >= p.n_prompt) *L.s_tok = next;
                if (w == 0) __hip_atomic_store(p.tok_dev + t, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the relay block takes it to the host)
                if (p.sample && p.win_cap > 0) {                     // the window slides over the generated ids (both sampling forms)
                    int wl = L.win[64];
                    if (wl < p.win_cap) { L.win[wl] = next; L.win[64] = wl + 1; }
                    else { for (int j = 0; j + 1 < wl; ++j) L.win[j] = L.win[j + 1]; L.win[wl - 1] = next; }
                }
                if (next == p.stop_id || *L.s_cancel) *L.s_done = 1;
            }
        }
    }
    if (w == 0 && tid == 0) {
        if (p.n_done) { p.n_done[0] = t_last + 1; p.n_done[1] = n_sampled; }
        __threadfence();                                             // every id above is visible before the relay is told that there are no more
        __hip_atomic_store(p.relay_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The relay: ONE wave on a compute unit no worker uses (block 256 + xcds, i.e. on XCD `xcds`: the blocks with index mod 8 >= xcds have left).  It polls the
// ids worker 0 publishes in device memory and copies them into the host-visible row, and polls the host's cancel word and forwards it
// into device memory - so that no worker ever issues an access that crosses PCIe.  (Round 6, first forms: the id stored to host memory
// by the thread that chose it - a system-scope store retires after a PCIe round trip and vmcnt retires in order, so that wave's next
// loads, and with them every worker's first edge of the position, waited for it; then by a matrix wave behind its tile requests - 512
// extra system-scope stores per position: 17.17 -> 17.56 -> 17.88 ms per request, profiles/r06/c7, c8.)  With 8 XCDs every compute unit
// holds a worker and the relay only runs once they have left: the ids then arrive together at the end - still correct, not streamed.
__device__ __forceinline__ void te_relay(const TeParams& p) {
    if (threadIdx.x != 0) return;
    int k = p.head_from;
    for (long it = 0; it < (1L << 26); ++it) {                       // (bounded: ~1 us per round)
        if (p.cancel && __hip_atomic_load(p.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)
            __hip_atomic_store(p.cancel_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (read BEFORE the ids are drained: what worker 0 chose before it raised the flag is then certainly seen below)
        const unsigned fin = __hip_atomic_load(p.relay_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) |
                             __hip_atomic_load(p.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (k < p.head_until) {
            const int v = __hip_atomic_load(p.tok_dev + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v < 0) break;
            __hip_atomic_store(p.next_tokens + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            ++k;
        }
        if (fin) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

template <int XCDS>
__global__ void __launch_bounds__(TE_NT) k_token_engine(TeParams p) {
    using S = TeShape;
    using Dm = TeDims<XCDS>;
    static_assert(S::H == TE_VW && S::Hkv == 1 && S::D == 128 && S::d / 4 <= 128 && S::Nqkv / 4 <= TE_VW * 64, "vector-wave mapping of the engine");
    extern __shared__ __attribute__((aligned(16))) unsigned char te_lds_pad[];      // (requested size keeps the launch at one block per CU)
    __shared__ __attribute__((aligned(16))) float hf[S::d];
    __shared__ __attribute__((aligned(16))) bf16_t xb[S::ff > S::HD ? S::ff : S::HD];
    __shared__ __attribute__((aligned(16))) float qkvf[S::Nqkv];
    __shared__ __attribute__((aligned(16))) float qh[S::H * S::D];
    __shared__ __attribute__((aligned(16))) float knew[S::D], vnew[S::D];
    __shared__ __attribute__((aligned(16))) float sc[S::H * TE_CTX];
    __shared__ __attribute__((aligned(16))) bf16_t ph[S::H * TE_CTX], pl[S::H * TE_CTX];
    __shared__ __attribute__((aligned(16))) bf16_t stage[Dm::R_RED * 16];
    __shared__ __attribute__((aligned(16))) float red[TE_MW * Dm::R_RED * 16];
    __shared__ float s_ss[TE_VW];
    __shared__ u64 s_cand[2 * TE_VW];
    __shared__ int s_ok;
    __shared__ int s_tok;
    __shared__ int s_done;
    __shared__ int s_cancel;
    __shared__ u64 earr[TE_HP * Dm::R_HEAD * 16];
    __shared__ uint32_t tsum32[TE_XG];
    __shared__ int win[65];
    __shared__ u64 s_wtot[TE_VW + 2];
    if (p.n_total < 0) te_lds_pad[threadIdx.x] = 0;
    const int b = blockIdx.x;
    if (b >= 256) { if (b == 256 + XCDS) te_relay(p); return; }     // (block 256 + xcds lands on XCD `xcds`, whose compute units the workers do not use)
    if ((b & 7) >= XCDS) return;
    const int w = (b >> 3) * XCDS + (b & 7);
    const int tid = threadIdx.x, wave = tid >> 6;
    if (tid < TE_VW) s_ss[tid] = 0.f;
    if (tid == 0) { s_ok = 1; s_tok = 0; s_done = 0; s_cancel = 0; win[64] = 0; }
    const TeLds L{hf, xb, qkvf, qh, knew, vnew, sc, ph, pl, stage, red, s_ss, s_cand, &s_ok, &s_tok, &s_done, &s_cancel, earr, tsum32, win, s_wtot};
    te_sync();
    // (the wave index as a SCALAR: every tile address is then scalar base + one shared lane offset.  With a vector wave index the
    // compiler keeps a 64-bit address pair per tile live across the layer loop - ~200 registers of addresses, everything spills)
    if (wave >= TE_VW) te_matrix_role<XCDS>(p, L, w, __builtin_amdgcn_readfirstlane(wave - TE_VW), tid & 63);     // waves 4..7: weight tiles and MFMAs
    else te_vector_role<XCDS>(p, L, w, tid);                                         // waves 0..3: everything else
}
// K/V of the positions the launch chain's prefill has processed, from its tiled caches (kernels.h, TtsKvView) into the engine's copy:
// keys row-major [position][D], values transposed [D][position]
__global__ void k_te_import_kv(const bf16_t* __restrict__ kc_all, const bf16_t* __restrict__ vt_all, size_t layer_stride, bf16_t* __restrict__ kv,
                               int n_pos) {
    constexpr int D = TeShape::D;
    const int li = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pos * D) return;
    const int pos = idx / D, d = idx % D;
    const bf16_t* kc = kc_all + (size_t)li * layer_stride;
    const bf16_t* vt = vt_all + (size_t)li * layer_stride;
    const int ptile = pos >> 5, pr = pos & 31, prow = ((pr >> 3) << 2) | (pr & 3), phalf = (pr >> 2) & 1;
    const bf16_t kval = kc[((((size_t)ptile * 2 + phalf) * (D / 32) + (d >> 5)) * 64 + (((d & 31) >> 3) << 4) + prow) * 8 + (d & 7)];
    const bf16_t vval = vt[(((size_t)ptile * (D / 16) + (d >> 4)) * 64 + ((pr >> 3) << 4) + (d & 15)) * 8 + (pr & 7)];
    kv[((size_t)li * 2 + 0) * TE_CTX * D + (size_t)pos * D + d] = kval;
    kv[((size_t)li * 2 + 1) * TE_CTX * D + (size_t)d * TE_CTX + pos] = vval;
}
}   // namespace

// ---------------------------------------------------------------------------- host side
// Buffers a handle keeps between requests (every request allocated and cleared a ~9 MB K/V copy).  The token row and the
// cancel word are pinned, coherent host memory: the launch writes / reads them with system-scope accesses while the host polls.
struct TokenEngineScratch {
    DevBuf<int32_t> prompt, done, tokd;
    DevBuf<bf16_t> kv;
    DevBuf<u64> x;
    DevBuf<unsigned> sync;
    int32_t* tokens_host = nullptr;    // [TE_CTX + 1]: ids, then the cancel word
    int kv_layers = 0;
    ~TokenEngineScratch() { if (tokens_host) (void)hipHostFree(tokens_host); }
};
TokenEngineScratch* token_engine_scratch_create() { return new TokenEngineScratch(); }
void token_engine_scratch_destroy(TokenEngineScratch* s) { delete s; }

namespace {
struct TeEvents {                      // (destroyed on every path)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    TeEvents() { HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); }
    ~TeEvents() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};
// 8 XCDs x 32 compute units (SPX mode of an MI355X): the worker set is "blocks with index mod 8 below xcds" of a 256-block grid, one block
// per compute unit.  Other partition modes / parts keep the launch chain.
bool te_device_ok(int device) {
    static std::mutex mu;
    static std::map<int, bool> ok;
    std::lock_guard<std::mutex> lk(mu);
    auto it = ok.find(device);
    if (it != ok.end()) return it->second;
    hipDeviceProp_t prop{};
    const bool good = hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount == 256;
    ok[device] = good;
    return good;
}
}   // namespace

bool token_engine_supports(mis_tts* lm) {
    if (!lm) return false;
    const TtsWeightsView v = tts_internal_weights(lm);
    using S = TeShape;
    return v.finalized && v.d == S::d && v.ff == S::ff && v.H == S::H && v.Hkv == S::Hkv && v.D == S::D && v.qk_norm && v.rope_plain && !v.quantised &&
           v.V <= 65536 && v.Vpad / 16 <= 2 * TE_VW * 64 && te_device_ok(v.device);
}

// One request.  generate == false (laboratory): arg-max after EVERY position, no stop; logits / hidden rows are indexed by position.
// generate == true: the product's semantics (tts_generate_hidden): a token after the last prompt position and after every generated one
// until `stop_id` or max_new ids; hidden row k = final-norm output of position n_prompt - 1 + k (row 0 = the last prompt token,
// Soprano.swift:824-825); logits row k = the logits the k-th token was drawn from.
void token_engine_run(mis_tts* lm, const TokenEngineRequest& rq, TokenEngineResult& out) {
    MIS_REQUIRE(lm && rq.prompt && rq.n_prompt >= 1 && rq.max_new >= 0, MIS_ERR_INVALID_INPUT, "bad argument");
    const int xcds = rq.xcds;
    MIS_REQUIRE(xcds == 1 || xcds == 2 || xcds == 4 || xcds == 8, MIS_ERR_INVALID_INPUT, "the engine is compiled for 1, 2, 4 or 8 XCDs");
    MIS_REQUIRE(token_engine_supports(lm), MIS_ERR_INVALID_INPUT,
                "the token engine is compiled for Soprano-80M's widths (d 512, ffn 2304, 4 / 1 heads x 128, q/k norm, plain RoPE, bf16, vocabulary <= 8192) "
                "on a device of 8 XCDs x 32 compute units");
    const TtsWeightsView v = tts_internal_weights(lm);
    using S = TeShape;
    const int n_total = rq.n_prompt + rq.max_new;
    MIS_REQUIRE(n_total <= TE_CTX, MIS_ERR_INVALID_INPUT, "at most %d positions", TE_CTX);
    MIS_REQUIRE(!rq.sample || (rq.temperature > 0.0f && rq.win_cap >= 0 && rq.win_cap <= 64), MIS_ERR_INVALID_INPUT, "sampling needs temperature > 0 and a window of at most 64 ids");
    MIS_REQUIRE(v.Vpad / 16 <= TE_HP * 8 * 32 * xcds, MIS_ERR_INVALID_INPUT, "vocabulary too large for the engine's output-projection passes");
    MIS_REQUIRE(rq.generate || (!rq.on_token && !rq.cancel), MIS_ERR_INVALID_INPUT, "token callback / cancel flag: generate form only");
    HIP_CHECK(hipSetDevice(v.device));
    const int grid = 256 + xcds + 1;                                     // 256 worker slots + the relay block (te_relay) as the last one: blocks go to
                                                                         // XCD (index mod 8), and XCD `xcds` is the first the workers leave free
    const float* rc = nullptr; const float* rs = nullptr;
    hipStream_t s = v.stream;
    out.n_announced = 0;
    std::vector<int32_t> hp(rq.n_prompt);
    HIP_CHECK(hipMemcpy(hp.data(), rq.prompt, (size_t)rq.n_prompt * 4, hipMemcpyDefault));
    for (int t : hp) MIS_REQUIRE(t >= 0 && t < v.V, MIS_ERR_INVALID_INPUT, "prompt token %d outside the vocabulary", t);
    // The prompt, all but its last position, through the launch chain's batched prefill (one [positions x 1] pass: ~1 ms where the engine
    // walks 0.24 ms per position); its K/V are imported below and the engine starts at the last prompt position.
    const int t_start = (rq.generate && rq.prefill_by_chain && rq.n_prompt >= 2) ? rq.n_prompt - 1 : 0;
    TtsKvView kvv{};
    if (t_start > 0) {
        kvv = tts_internal_prefill_kv(lm, hp.data(), t_start, n_total);
        MIS_REQUIRE(kvv.D == S::D && kvv.Hkv == S::Hkv, MIS_ERR_GENERATION_FAILED, "token engine: unexpected cache geometry");
        rc = kvv.rope_cos; rs = kvv.rope_sin;
    } else {
        tts_internal_rope_tables(lm, n_total, &rc, &rs);                 // (builds the tables for this context length)
    }
    const int head_from = rq.generate ? rq.n_prompt - 1 : 0;
    const int head_until = rq.generate ? n_total - 1 : n_total;
    const int n_rows = n_total - head_from;                              // hidden rows at most; logits rows: head_until - head_from
    TokenEngineScratch local;
    TokenEngineScratch& sc = rq.scratch ? *rq.scratch : local;
    DevBuf<float> d_logits, d_hidden;
    sc.prompt.alloc(TE_CTX); sc.done.alloc(2); sc.tokd.alloc(TE_CTX); sc.x.alloc(2 * TE_XG); sc.sync.alloc(64);
    if (!sc.tokens_host)
        HIP_CHECK(hipHostMalloc((void**)&sc.tokens_host, (TE_CTX + 1) * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
    const size_t kv_pos = (size_t)S::Hkv * S::D;
    // The transposed value rows are read in 32-key steps: positions this request has not written yet (x probability 0) must be finite.
    // Cleared when allocated; what earlier requests of the handle left there are finite K/V values, and 0 x finite = 0 exactly.
    if (sc.kv_layers != v.L) {
        sc.kv.alloc((size_t)v.L * 2 * TE_CTX * kv_pos); sc.kv_layers = v.L;
        HIP_CHECK(hipMemsetAsync(sc.kv.p, 0, sc.kv.bytes(), s));
    }
    if (rq.want_logits) d_logits.alloc((size_t)std::max(head_until - head_from, 1) * v.V);
    float* hidden_dev = rq.hidden_dev;
    if (rq.want_hidden && !hidden_dev) { d_hidden.alloc((size_t)n_rows * S::d); hidden_dev = d_hidden.p; }
    volatile int32_t* tok_host = sc.tokens_host;
    volatile int32_t* cancel_host = sc.tokens_host + TE_CTX;
    for (int t = 0; t < n_total; ++t) tok_host[t] = -1;                   // (an id is >= 0: -1 = not chosen yet)
    *cancel_host = 0;
    HIP_CHECK(hipMemcpyAsync(sc.prompt.p, hp.data(), (size_t)rq.n_prompt * 4, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemsetAsync(sc.sync.p, 0, 64 * sizeof(unsigned), s));
    HIP_CHECK(hipMemsetAsync(sc.x.p, 0, 2 * TE_XG * sizeof(u64), s));
    HIP_CHECK(hipMemsetAsync(sc.done.p, 0, 8, s));
    HIP_CHECK(hipMemsetAsync(sc.tokd.p, 0xFF, (size_t)TE_CTX * 4, s));                 // -1: not chosen yet
    if (t_start > 0)
        hipLaunchKernelGGL(k_te_import_kv, dim3((unsigned)((t_start * S::D + 255) / 256), (unsigned)v.L), dim3(256), 0, s, kvv.kcache, kvv.vtcache,
                           kvv.layer_stride, sc.kv.p, t_start);
    TeParams p{};
    p.emb = v.emb; p.wqkv = v.wqkv; p.wo = v.wo; p.wgu = v.wgu; p.wdown = v.wdown; p.head = v.head; p.norms = v.norms; p.qknorm = v.qknorm;
    p.rope_cos = rc; p.rope_sin = rs; p.L = v.L; p.V = v.V; p.Vpad = v.Vpad; p.eps = v.eps;
    p.prompt = sc.prompt.p; p.n_prompt = rq.n_prompt; p.n_total = n_total; p.t_start = t_start; p.next_tokens = sc.tokens_host;
    p.tok_dev = sc.tokd.p; p.cancel_dev = reinterpret_cast<int*>(sc.sync.p + 40); p.relay_done = sc.sync.p + 41;
    p.cancel = rq.cancel ? sc.tokens_host + TE_CTX : nullptr;
    p.logits_out = rq.want_logits ? d_logits.p : nullptr; p.hidden_out = hidden_dev;
    p.kv = sc.kv.p; p.xbuf = sc.x.p; p.fail = sc.sync.p + 32; p.xcds = xcds;
    p.spin = rq.spin > 0 ? rq.spin : 1 << 20;
    if (const char* e = getenv("MIS_TE_SPIN")) p.spin = std::max(atoi(e), 0);      // (tests: 0 = the first poll that misses times out)
    p.head_from = head_from; p.head_until = head_until;
    p.sample = rq.sample ? 1 : (rq.generate ? 2 : 0); p.win_cap = rq.win_cap; p.temperature = rq.temperature; p.penalty = rq.penalty; p.seed = rq.seed; p.row = rq.row;
    p.stop_id = rq.generate ? rq.stop_id : -1; p.n_done = sc.done.p;
    DevBuf<u64> d_dbg;
    const char* stamp_env = getenv("MIS_TE_STAMPS");
    if (stamp_env) {
        d_dbg.alloc(32);
        HIP_CHECK(hipMemsetAsync(d_dbg.p, 0, 32 * 8, s));
        p.dbg = d_dbg.p; p.dbg_token = atoi(stamp_env);
    }
    const size_t pad = 48 * 1024;                                        // with the static arrays: more than half a CU's LDS -> one block per CU
    // (the attribute is per device and per function: set on every request - a table keyed by XCD count alone left a second device without it)
    if (xcds == 1) HIP_CHECK(hipFuncSetAttribute((const void*)k_token_engine<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
    else if (xcds == 2) HIP_CHECK(hipFuncSetAttribute((const void*)k_token_engine<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
    else if (xcds == 4) HIP_CHECK(hipFuncSetAttribute((const void*)k_token_engine<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
    else HIP_CHECK(hipFuncSetAttribute((const void*)k_token_engine<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
    TeEvents ev;
    HIP_CHECK(hipEventRecord(ev.e0, s));
    if (xcds == 1) hipLaunchKernelGGL(k_token_engine<1>, dim3(grid), dim3(TE_NT), pad, s, p);
    else if (xcds == 2) hipLaunchKernelGGL(k_token_engine<2>, dim3(grid), dim3(TE_NT), pad, s, p);
    else if (xcds == 4) hipLaunchKernelGGL(k_token_engine<4>, dim3(grid), dim3(TE_NT), pad, s, p);
    else hipLaunchKernelGGL(k_token_engine<8>, dim3(grid), dim3(TE_NT), pad, s, p);
    HIP_CHECK(hipEventRecord(ev.e1, s));
    HIP_CHECK(hipGetLastError());
    // ---- while the launch runs: the ids it has chosen so far, in order, to the caller; the caller's cancel flag to the launch
    bool cancelled = false;
    int seen = 0;                                                        // chosen ids read so far (index head_from + seen)
    auto drain = [&]() {
        while (head_from + seen < head_until) {
            const int32_t id = tok_host[head_from + seen];
            if (id < 0) break;
            if (rq.on_token) { rq.on_token(seen, id); out.n_announced = seen + 1; }
            seen += 1;
            if (id == p.stop_id) break;
        }
    };
    if (rq.on_token || rq.cancel) {
        for (;;) {
            const hipError_t q = hipEventQuery(ev.e1);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) HIP_CHECK(q);
            drain();
            if (rq.cancel && *rq.cancel && !cancelled) { cancelled = true; *cancel_host = 1; }
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    unsigned failed = 0;
    HIP_CHECK(hipMemcpy(&failed, sc.sync.p + 32, 4, hipMemcpyDeviceToHost));
    MIS_REQUIRE(!failed, MIS_ERR_GENERATION_FAILED, "token engine: an edge timed out (its workers were not co-resident)");
    if (cancelled || (rq.cancel && *rq.cancel)) throw MisError(MIS_ERR_CANCELLED, "generation cancelled");
    drain();
    int32_t done[2] = {0, 0};
    HIP_CHECK(hipMemcpy(done, sc.done.p, 8, hipMemcpyDeviceToHost));
    out.n_positions = done[0]; out.n_sampled = done[1]; out.ms = ms; out.head_from = head_from;
    out.next_tokens.assign(n_total, 0);
    HIP_CHECK(hipMemcpy(out.next_tokens.data(), sc.tokd.p, (size_t)n_total * 4, hipMemcpyDeviceToHost));
    for (int t = 0; t < n_total; ++t) {
        MIS_REQUIRE(out.next_tokens[t] == tok_host[t], MIS_ERR_GENERATION_FAILED, "token engine: the host-visible id row differs from the device's at position %d", t);
        if (out.next_tokens[t] < 0) out.next_tokens[t] = 0;
    }
    if (rq.want_logits) {
        out.logits.assign((size_t)std::max(head_until - head_from, 1) * v.V, 0.f);
        HIP_CHECK(hipMemcpy(out.logits.data(), d_logits.p, out.logits.size() * 4, hipMemcpyDeviceToHost));
    }
    if (rq.want_hidden && !rq.hidden_dev) {
        out.hidden.assign((size_t)n_rows * S::d, 0.f);
        HIP_CHECK(hipMemcpy(out.hidden.data(), d_hidden.p, out.hidden.size() * 4, hipMemcpyDeviceToHost));
    }
    if (stamp_env) {
        u64 st[32];
        HIP_CHECK(hipMemcpy(st, d_dbg.p, sizeof(st), hipMemcpyDeviceToHost));
        static const char* names[20] = {"layer start", "qkv: red ready", "published", "q|k|v polled in", "barrier", "q/k-norm + RoPE", "scores",
                                        "softmax + P.V", "o: red ready", "published", "h polled in", "barrier", "gate|up: red ready", "published",
                                        "act polled in", "barrier", "down: red ready", "published", "h polled in", "barrier"};
        fprintf(stderr, "token engine, %d XCD(s), position %d, layer 1, worker 0 (shader cycles since layer start, delta):\n", xcds, p.dbg_token);
        for (int i = 0; i < 20; ++i)
            fprintf(stderr, "  %2d %-20s %8llu %6lld\n", i, names[i], (unsigned long long)(st[i] - st[0]), i ? (long long)(st[i] - st[i - 1]) : 0ll);
    }
}

// include/mi_speech_debug.h
extern "C" mis_status mis_debug_token_engine(mis_tts* lm, const int32_t* prompt, int n_prompt, int n_new, int xcds, const mis_gen_params* sampling,
                                             int stop_id, int32_t* next_tokens, float* logits_out, float* hidden_out, int32_t* counts,
                                             double* ms_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(next_tokens, MIS_ERR_INVALID_INPUT, "null argument");
    TokenEngineRequest rq{};
    rq.prompt = prompt; rq.n_prompt = n_prompt; rq.max_new = n_new; rq.xcds = xcds;
    rq.generate = sampling != nullptr;
    rq.prefill_by_chain = rq.generate;                                   // the generate form as the product runs it (mis_soprano_generate)
    if (sampling) {
        rq.sample = sampling->temperature > 0.0f;
        rq.temperature = sampling->temperature; rq.penalty = sampling->repetition_penalty; rq.win_cap = std::max(sampling->repetition_context, 0);
        rq.seed = sampling->seed; rq.row = sampling->row_offset; rq.stop_id = stop_id;
    }
    rq.want_logits = logits_out != nullptr; rq.want_hidden = hidden_out != nullptr;
    TokenEngineResult r;
    token_engine_run(lm, rq, r);
    const int n_total = n_prompt + n_new;
    memcpy(next_tokens, r.next_tokens.data(), (size_t)n_total * 4);
    if (logits_out) memcpy(logits_out, r.logits.data(), r.logits.size() * 4);
    if (hidden_out) memcpy(hidden_out, r.hidden.data(), r.hidden.size() * 4);
    if (counts) { counts[0] = r.n_positions; counts[1] = r.n_sampled; }
    if (ms_out) *ms_out = r.ms;
    MIS_API_END
}
