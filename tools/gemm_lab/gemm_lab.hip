// gemm_lab.hip - bench-side laboratory for the weight-streaming bf16 GEMM of the decode step (csrc/lm_kernels.hip, k_gemm_skinny).
// NOT part of the product: a stand-alone executable that times candidate arrangements next to the library's own launcher on the same
// buffers and checks them against it, so that a variant is measured before anything in the (hashed) step-chain sources changes.
//
//   make -C tools/gemm_lab            (links against mlx-audio-swift_amd/libmi_speech.so)
//   tools/gemm_lab/gemm_lab [rows=32] [iters=64] [shape]     -> one JSON line per (shape, variant) on stdout
//
// Why these two candidates:
//   * k_lab_stream<MT, R, KSB, U>: the product's loop with R n-tiles per wave and KSB waves per item as parameters.  Every wave re-reads
//     its x fragments out of L2, so at R = 2 a launch moves as many x bytes through the CUs' vector-memory path as weight bytes; the
//     product kernel streams weights that sit in the Infinity Cache only 3.5 % faster than from HBM and a third register buffer made it
//     slower - what is left is the CU's own queue.  R = 4 halves the x bytes and doubles the weight bytes a wave has in flight.
//   * k_lab_oneshot<MT, R, KSB, UK>: for wave shares of at most UK k-tiles - every load issued before the first MFMA, wave index scalar,
//     straight-line code (the arrangement that made the quantised kernel 8-10 % faster at equal round trips, k_gemm_skinny_q1).
// Output of every variant: float32 slabs [S][Mpad][N] (the product's EPI_PARTIAL), compared with launch_gemm_skinny(EPI_PARTIAL, 2, 4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <algorithm>

#include "common.h"
#include "lm_kernels.h"

#define LAB_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

// ---------------------------------------------------------------------------- shared pieces
// slab store of one (r, mt) accumulator tile in the product's C/D layout: lane = column m (l & 15), rows n = 4 (l >> 4) + e
template <int MT, int R>
__device__ __forceinline__ void lab_store(const f32x4_t (&acc)[R][MT], float* __restrict__ out, int ntg, int ks, int NT, int N_out, int Mpad,
                                          int lane, int mt_only) {
    const int nl = (lane >> 4) * 4, ml = lane & 15;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int tile = ntg * R + r;
        if (tile >= NT) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt_only >= 0 && mt != mt_only) continue;
            const size_t off = ((size_t)ks * Mpad + mt * 16 + ml) * N_out + tile * 16 + nl;
            *reinterpret_cast<float4*>(out + off) = make_float4(acc[r][mt][0], acc[r][mt][1], acc[r][mt][2], acc[r][mt][3]);
        }
    }
}
// in-block split-K combine (KSB partials in wave order) followed by the store
template <int MT, int R, int KSB>
__device__ __forceinline__ void lab_combine_store(f32x4_t (&acc)[R][MT], float4 (*red)[R * MT][64], float* __restrict__ out, int ntg, int ks,
                                                  int NT, int N_out, int Mpad, int lane, int wave) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) red[wave][r * MT + mt][lane] = make_float4(acc[r][mt][0], acc[r][mt][1], acc[r][mt][2], acc[r][mt][3]);
    __syncthreads();
    for (int mt = wave; mt < MT; mt += KSB) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float4 s0 = red[0][r * MT + mt][lane];
#pragma unroll
            for (int w = 1; w < KSB; ++w) {
                const float4 t = red[w][r * MT + mt][lane];
                s0.x += t.x; s0.y += t.y; s0.z += t.z; s0.w += t.w;
            }
#pragma unroll
            for (int m2 = 0; m2 < MT; ++m2)
                if (m2 == mt) acc[r][m2] = (f32x4_t){s0.x, s0.y, s0.z, s0.w};
        }
        lab_store<MT, R>(acc, out, ntg, ks, NT, N_out, Mpad, lane, mt);
    }
}

// ---------------------------------------------------------------------------- candidate 1: the streaming loop with R and KSB free
// KSB = 1: four independent items per 256-thread block (the product's output-projection arrangement), no combine
template <int MT, int R, int KSB, int U>
__global__ void __launch_bounds__((KSB == 1 ? 4 : KSB) * 64, 2) k_lab_stream(const bf16_t* __restrict__ Wp, const bf16_t* __restrict__ X,
                                                                           float* __restrict__ out, int NT, int KT, int S, int n_items, int N_out,
                                                                           int Mpad) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = KSB == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    if (item >= n_items) return;
    const int ntg = item / S, ks = item - ntg * S;
    int kt0 = (int)(((long long)KT * ks) / S), kt1 = (int)(((long long)KT * (ks + 1)) / S);
    if (KSB > 1) {
        const int len = kt1 - kt0;
        const int a = kt0 + (int)(((long long)len * wave) / KSB), b = kt0 + (int)(((long long)len * (wave + 1)) / KSB);
        kt0 = a; kt1 = b;
    }
    const bf16x8_t* wp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int tile = ntg * R + r;
        if (tile >= NT) tile = NT - 1;
        wp[r] = reinterpret_cast<const bf16x8_t*>(Wp) + (size_t)tile * KT * 64 + lane;
    }
    const bf16x8_t* xp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xp[mt] = reinterpret_cast<const bf16x8_t*>(X) + mt * 64 + lane;
    f32x4_t acc[R][MT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[r][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    bf16x8_t wA[U][R], xA[U][MT], wB[U][R], xB[U][MT];
    const int klast = kt1 - 1;
#define LAB_LOAD(WBUF, XBUF, KBASE)                                                                \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                \
        int kk = (KBASE) + u;                                                                      \
        kk = kk > klast ? klast : kk;                                                              \
        _Pragma("unroll") for (int r = 0; r < R; ++r) WBUF[u][r] = __builtin_nontemporal_load(wp[r] + (size_t)kk * 64); \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) XBUF[u][mt] = xp[mt][(size_t)kk * (MT * 64)]; \
    }
#define LAB_MATH_FULL(WBUF, XBUF)                                                                  \
    _Pragma("unroll") for (int u = 0; u < U; ++u)                                                  \
        _Pragma("unroll") for (int r = 0; r < R; ++r)                                              \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                      \
                acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WBUF[u][r], XBUF[u][mt], acc[r][mt], 0, 0, 0);
#define LAB_MATH_TAIL(WBUF, XBUF, KBASE)                                                           \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                \
        if ((KBASE) + u < kt1) {                                                                   \
            _Pragma("unroll") for (int r = 0; r < R; ++r)                                          \
                _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                  \
                    acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WBUF[u][r], XBUF[u][mt], acc[r][mt], 0, 0, 0); \
        }                                                                                          \
    }
    if (kt0 < kt1) {
        int kt = kt0;
        LAB_LOAD(wA, xA, kt)
        while (kt + 3 * U <= kt1) {
            LAB_LOAD(wB, xB, kt + U)
            __builtin_amdgcn_sched_barrier(0);
            LAB_MATH_FULL(wA, xA)
            __builtin_amdgcn_sched_barrier(0);
            LAB_LOAD(wA, xA, kt + 2 * U)
            __builtin_amdgcn_sched_barrier(0);
            LAB_MATH_FULL(wB, xB)
            __builtin_amdgcn_sched_barrier(0);
            kt += 2 * U;
        }
        if (kt + U < kt1) {
            LAB_LOAD(wB, xB, kt + U)
            __builtin_amdgcn_sched_barrier(0);
            LAB_MATH_TAIL(wA, xA, kt)
            if (kt + 2 * U < kt1) {
                LAB_LOAD(wA, xA, kt + 2 * U)
                __builtin_amdgcn_sched_barrier(0);
                LAB_MATH_TAIL(wB, xB, kt + U)
                LAB_MATH_TAIL(wA, xA, kt + 2 * U)
            } else {
                LAB_MATH_TAIL(wB, xB, kt + U)
            }
        } else {
            LAB_MATH_TAIL(wA, xA, kt)
        }
    }
#undef LAB_LOAD
#undef LAB_MATH_FULL
#undef LAB_MATH_TAIL
    if (KSB == 1) {
        lab_store<MT, R>(acc, out, ntg, ks, NT, N_out, Mpad, lane, -1);
    } else {
        __shared__ float4 red[KSB][R * MT][64];
        lab_combine_store<MT, R, KSB>(acc, red, out, ntg, ks, NT, N_out, Mpad, lane, wave);
    }
}

// ---------------------------------------------------------------------------- candidate 2: one shot (wave share <= UK k-tiles)
// Operands are not __restrict__ and a memory fence follows the loads: loads through noalias read-only pointers may be moved across
// anything (csrc/lm_qgemm.hip, k_gemm_skinny_q1, has the story); every buffer register passes an empty volatile asm at its first use
// so that nothing that consumes a load rises above the fence.  K-tiles past the wave's share read the zero fragment Z instead of x.
template <int MT, int R, int KSB, int UK>
__global__ void __launch_bounds__(KSB * 64, 2) k_lab_oneshot(const bf16_t* Wp, const bf16_t* X, const bf16_t* Z, float* __restrict__ out, int NT,
                                                           int KT, int S, int n_items, int N_out, int Mpad) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x;
    if (item >= n_items) return;
    const int ntg = item / S, ks = item - ntg * S;
    int kt0 = (int)(((long long)KT * ks) / S), kt1 = (int)(((long long)KT * (ks + 1)) / S);
    {
        const int len = kt1 - kt0;
        const int a = kt0 + (int)(((long long)len * wave) / KSB), b = kt0 + (int)(((long long)len * (wave + 1)) / KSB);
        kt0 = a; kt1 = b;
    }
    f32x4_t acc[R][MT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[r][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (kt0 < kt1) {
        const bf16x8_t* wp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int tile = ntg * R + r;
            if (tile >= NT) tile = NT - 1;
            wp[r] = reinterpret_cast<const bf16x8_t*>(Wp) + (size_t)tile * KT * 64 + lane;
        }
        const bf16x8_t* zp = reinterpret_cast<const bf16x8_t*>(Z) + lane;
        bf16x8_t w[UK][R], x[UK][MT];
        const int klast = kt1 - 1;
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            const bool live = kt0 + u < kt1;
            const int kk = live ? kt0 + u : klast;
#pragma unroll
            for (int r = 0; r < R; ++r) w[u][r] = __builtin_nontemporal_load(wp[r] + (size_t)kk * 64);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bf16x8_t* xs = live ? reinterpret_cast<const bf16x8_t*>(X) + ((size_t)kk * MT + mt) * 64 + lane : zp;
                x[u][mt] = *xs;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UK; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(w[u][r]));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(x[u][mt]));
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][r], x[u][mt], acc[r][mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __shared__ float4 red[KSB][R * MT][64];
    lab_combine_store<MT, R, KSB>(acc, red, out, ntg, ks, NT, N_out, Mpad, lane, wave);
}

// ---------------------------------------------------------------------------- harness
struct Shape { const char* name; int N, K, S, ksb_ref; };      // S, ksb_ref: what the engine launches today (gemm_choose_split)

struct Ctx {
    Shape sh;
    int rows, Mpad, iters, L, NT, KT;
    size_t wn, on;
    bf16_t *W, *X, *zero;
    float *O0, *O1;
    std::vector<float> ref, got, ref_sum;
    int Smax = 8;
    bool sweep_S = false;
    hipStream_t s;
};

template <typename F>
static double time_launches(F&& launch, int iters, hipStream_t s) {
    hipEvent_t a, b;
    LAB_CHECK(hipEventCreate(&a)); LAB_CHECK(hipEventCreate(&b));
    for (int i = 0; i < 8; ++i) launch(i);
    LAB_CHECK(hipStreamSynchronize(s));
    LAB_CHECK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) launch(i);
    LAB_CHECK(hipEventRecord(b, s));
    LAB_CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    LAB_CHECK(hipEventElapsedTime(&ms, a, b));
    LAB_CHECK(hipEventDestroy(a)); LAB_CHECK(hipEventDestroy(b));
    return (double)ms * 1e3 / iters;                  // us per launch
}

static double max_rel_diff(const std::vector<float>& a, const std::vector<float>& b) {
    double worst = 0.0, scale = 0.0;
    for (size_t i = 0; i < a.size(); ++i) scale = fmax(scale, fabs((double)b[i]));
    for (size_t i = 0; i < a.size(); ++i) worst = fmax(worst, fabs((double)a[i] - (double)b[i]));
    return scale > 0 ? worst / scale : worst;
}

static void report(const Ctx& c, const char* variant, int S, double us, double err) {
    const double mb = (double)c.wn * 2 / 1e6;
    printf("{\"shape\": \"%s\", \"N\": %d, \"K\": %d, \"S\": %d, \"rows\": %d, \"variant\": \"%s\", \"us\": %.2f, \"MB\": %.2f, \"GBps\": %.1f, "
           "\"max_rel_vs_product\": %.3g}\n", c.sh.name, c.sh.N, c.sh.K, S, c.rows, variant, us, mb, mb / us * 1e3, err);
    fflush(stdout);
}

// sum of the S float32 slabs [S][Mpad][N] in slab order (what the consumer's prologue computes)
static void slab_sum(const std::vector<float>& slabs, int S, size_t mn, std::vector<float>& out) {
    out.assign(mn, 0.0f);
    for (int s = 0; s < S; ++s)
        for (size_t i = 0; i < mn; ++i) out[i] += slabs[(size_t)s * mn + i];
}
// one launch on weight copy `layer`, checked once against the product's output (sum over the split-K slabs: variants may use another
// S than the product), then timed
template <typename L>
static void check_and_time(Ctx& c, const char* label, int S, L&& launch) {
    const size_t mn = (size_t)c.Mpad * c.sh.N;
    LAB_CHECK(hipMemsetAsync(c.O1, 0, mn * S * 4, c.s));
    launch(0);
    LAB_CHECK(hipGetLastError());
    c.got.resize(mn * S);
    LAB_CHECK(hipMemcpyAsync(c.got.data(), c.O1, mn * S * 4, hipMemcpyDeviceToHost, c.s));
    LAB_CHECK(hipStreamSynchronize(c.s));
    std::vector<float> gs;
    slab_sum(c.got, S, mn, gs);
    const double err = max_rel_diff(gs, c.ref_sum);
    report(c, label, S, time_launches(launch, c.iters, c.s), err);
}

template <int MT, int R, int KSB, int U>
static void run_stream(Ctx& c, int S = 0) {
    if (S <= 0) S = c.sh.S;
    if (S > c.Smax || c.KT / (S * KSB) < 1) return;
    const int n_items = ((c.NT + R - 1) / R) * S;
    const dim3 grid(KSB == 1 ? (n_items + 3) / 4 : n_items), block((KSB == 1 ? 4 : KSB) * 64);
    char label[96];
    snprintf(label, sizeof label, "stream R%d KSB%d U%d S%d", R, KSB, U, S);
    check_and_time(c, label, S, [&](int i) {
        hipLaunchKernelGGL((k_lab_stream<MT, R, KSB, U>), grid, block, 0, c.s, c.W + (size_t)(i % c.L) * c.wn, c.X, c.O1, c.NT, c.KT, S, n_items,
                           c.sh.N, c.Mpad);
    });
}

template <int MT, int R, int KSB, int UK>
static void run_oneshot(Ctx& c, int S = 0) {
    if (S <= 0) S = c.sh.S;
    if (S > c.Smax) return;
    const int n_items = ((c.NT + R - 1) / R) * S;
    const int share = ((c.KT + S - 1) / S + KSB - 1) / KSB;         // k-tiles of the longest wave share
    if (share > UK || share < 1) return;
    char label[96];
    snprintf(label, sizeof label, "one-shot R%d KSB%d UK%d S%d", R, KSB, UK, S);
    check_and_time(c, label, S, [&](int i) {
        hipLaunchKernelGGL((k_lab_oneshot<MT, R, KSB, UK>), dim3(n_items), dim3(KSB * 64), 0, c.s, c.W + (size_t)(i % c.L) * c.wn, c.X, c.zero, c.O1,
                           c.NT, c.KT, S, n_items, c.sh.N, c.Mpad);
    });
}

template <int MT>
static void run_shape(Ctx& c) {
    const Shape& sh = c.sh;
    // the product launcher on weight copy 0 is the reference of every variant
    launch_gemm_skinny(EPI_PARTIAL, 2, sh.ksb_ref, c.W, c.X, c.O0, c.NT, c.KT, sh.S, sh.N, c.Mpad, c.s);
    LAB_CHECK(hipMemcpyAsync(c.ref.data(), c.O0, c.on * 4, hipMemcpyDeviceToHost, c.s));
    LAB_CHECK(hipStreamSynchronize(c.s));
    char label[96];
    snprintf(label, sizeof label, "product k_gemm_skinny R2 KSB%d", sh.ksb_ref);
    slab_sum(c.ref, sh.S, (size_t)c.Mpad * sh.N, c.ref_sum);
    report(c, label, sh.S, time_launches([&](int i) {
               launch_gemm_skinny(EPI_PARTIAL, 2, sh.ksb_ref, c.W + (size_t)(i % c.L) * c.wn, c.X, c.O0, c.NT, c.KT, sh.S, sh.N, c.Mpad, c.s); },
               c.iters, c.s), 0.0);
    if (sh.ksb_ref == 1) {                            // output projection: one wave per item, long K shares
        run_stream<MT, 2, 1, 4>(c);
        run_stream<MT, 4, 1, 3>(c);
        run_stream<MT, 4, 1, 2>(c);
        run_stream<MT, 8, 1, 1>(c);
        run_stream<MT, 4, 2, 3>(c);
        run_stream<MT, 4, 4, 3>(c);
        run_stream<MT, 4, 4, 2>(c);
        run_stream<MT, 4, 8, 3>(c);
        run_stream<MT, 2, 4, 4>(c);
        run_stream<MT, 2, 2, 4>(c);
        return;
    }
    if (c.sweep_S) {                                  // round 4: the split factor together with the arrangement (profiles/r04/)
        for (int S = 1; S <= 8; ++S) {
            if (S == 5 || S == 7) continue;
            run_stream<MT, 2, 4, 4>(c, S);
            run_stream<MT, 2, 2, 4>(c, S);
            run_stream<MT, 4, 4, 3>(c, S);
            run_stream<MT, 4, 4, 2>(c, S);
            run_stream<MT, 4, 2, 3>(c, S);
            run_stream<MT, 4, 2, 2>(c, S);
            run_oneshot<MT, 2, 4, 8>(c, S);
            run_oneshot<MT, 2, 2, 8>(c, S);
            run_oneshot<MT, 4, 4, 6>(c, S);
            run_oneshot<MT, 4, 2, 6>(c, S);
        }
        return;
    }
    run_stream<MT, 2, 4, 4>(c);                       // the product's shape (scalar wave index)
    run_stream<MT, 2, 2, 4>(c);
    run_stream<MT, 2, 8, 4>(c);
    run_stream<MT, 4, 4, 3>(c);                       // U = 4 at R = 4 spills (68 bytes of scratch)
    run_stream<MT, 4, 4, 2>(c);
    run_stream<MT, 4, 2, 3>(c);
    run_stream<MT, 4, 8, 3>(c);
    run_stream<MT, 4, 8, 2>(c);
    run_stream<MT, 8, 4, 1>(c);
    run_stream<MT, 8, 2, 1>(c);
    run_stream<MT, 8, 8, 1>(c);
    run_oneshot<MT, 2, 4, 8>(c);
    run_oneshot<MT, 2, 4, 12>(c);
    run_oneshot<MT, 2, 8, 8>(c);
    run_oneshot<MT, 2, 8, 12>(c);
    run_oneshot<MT, 4, 4, 6>(c);
    run_oneshot<MT, 4, 8, 6>(c);
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 32, iters = argc > 2 ? atoi(argv[2]) : 64;
    const char* only = argc > 3 ? argv[3] : nullptr;                        // optional: one shape name
    const int Mpad = (rows + 15) / 16 * 16, MT = Mpad / 16, L = 8;           // L weight copies: rotation defeats the Infinity Cache
    if (MT < 1 || MT > 2) { fprintf(stderr, "the lab instantiates 1..32 rows (MT = 1, 2)\n"); return 1; }
    hipStream_t s;
    LAB_CHECK(hipStreamCreate(&s));
    // Orpheus-3B (hidden 3072, ffn 8192, vocabulary 156 940 padded to 156 960) and Qwen3-TTS-0.6B talker (1024 / 3072, 16 x 128 heads) shapes
    const Shape shapes[] = {{"qkv", 5120, 3072, 3, 4}, {"o_proj", 3072, 3072, 2, 4}, {"gate_up", 16384, 3072, 1, 4}, {"down", 3072, 8192, 8, 4},
                            {"lm_head", 156960, 3072, 1, 1},
                            {"q3_qkv", 4096, 1024, 2, 4}, {"q3_o", 1024, 2048, 4, 4}, {"q3_gate_up", 6144, 1024, 1, 4}, {"q3_down", 1024, 3072, 6, 4},
                            {"q3_head", 3072, 1024, 1, 4}};
    bf16_t* zero = nullptr;
    LAB_CHECK(hipMalloc(&zero, 64 * 16));
    LAB_CHECK(hipMemset(zero, 0, 64 * 16));
    for (const Shape& sh : shapes) {
        if (only && strcmp(only, sh.name)) continue;
        Ctx c;
        c.sh = sh; c.rows = rows; c.Mpad = Mpad; c.iters = iters; c.L = L; c.NT = sh.N / 16; c.KT = sh.K / 32; c.s = s; c.zero = zero;
        c.sweep_S = getenv("LAB_SWEEP_S") != nullptr && sh.ksb_ref != 1 && strncmp(sh.name, "q3", 2) != 0;
        c.wn = (size_t)sh.N * sh.K; c.on = (size_t)std::max(sh.S, c.Smax) * Mpad * sh.N;
        const size_t xn = (size_t)Mpad * sh.K;
        LAB_CHECK(hipMalloc(&c.W, c.wn * 2 * L)); LAB_CHECK(hipMalloc(&c.X, xn * 2));
        LAB_CHECK(hipMalloc(&c.O0, c.on * 4)); LAB_CHECK(hipMalloc(&c.O1, c.on * 4));
        launch_synth_fill_bf16(c.W, c.wn * L, 0x1234u + sh.N, 0.02f, 0, s);
        launch_synth_fill_bf16(c.X, xn, 0x9876u + sh.K, 1.0f, 0, s);
        LAB_CHECK(hipStreamSynchronize(s));
        c.ref.resize(c.on); c.got.resize(c.on);
        if (MT == 1) run_shape<1>(c); else run_shape<2>(c);
        LAB_CHECK(hipFree(c.W)); LAB_CHECK(hipFree(c.X)); LAB_CHECK(hipFree(c.O0)); LAB_CHECK(hipFree(c.O1));
    }
    LAB_CHECK(hipFree(zero));
    return 0;
}
