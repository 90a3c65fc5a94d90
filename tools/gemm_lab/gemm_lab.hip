// gemm_lab.hip - bench-side laboratory for the weight-streaming bf16 GEMM of the decode step (csrc/lm_kernels.hip, k_gemm_skinny).
// NOT part of the product: a stand-alone executable that times candidate arrangements next to the library's own launcher on the same
// buffers and checks them against it, so that a variant is measured before anything in the (hashed) step-chain sources changes.
//
//   make -C tools/gemm_lab            (links against mlx-audio-swift_amd/libmi_speech.so)
//   tools/gemm_lab/gemm_lab [rows=32] [iters=64]     -> one JSON line per (shape, variant) on stdout
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
#include <vector>

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
template <int MT, int R, int KSB, int U>
__global__ void __launch_bounds__(KSB * 64, 2) k_lab_stream(const bf16_t* __restrict__ Wp, const bf16_t* __restrict__ X, float* __restrict__ out,
                                                          int NT, int KT, int S, int n_items, int N_out, int Mpad) {
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
    __shared__ float4 red[KSB][R * MT][64];
    lab_combine_store<MT, R, KSB>(acc, red, out, ntg, ks, NT, N_out, Mpad, lane, wave);
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
struct Shape { const char* name; int N, K, S; };

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

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 32, iters = argc > 2 ? atoi(argv[2]) : 64;
    const int Mpad = (rows + 15) / 16 * 16, MT = Mpad / 16, L = 8;           // L weight copies: rotation defeats the Infinity Cache
    if (MT != 2) { fprintf(stderr, "the lab instantiates 17..32 rows (MT = 2) only\n"); return 1; }
    hipStream_t s;
    LAB_CHECK(hipStreamCreate(&s));
    const Shape shapes[] = {{"qkv", 5120, 3072, 3}, {"o_proj", 3072, 3072, 2}, {"gate_up", 16384, 3072, 1}, {"down", 3072, 8192, 8}};
    bf16_t* zero = nullptr;
    LAB_CHECK(hipMalloc(&zero, 64 * 16));
    LAB_CHECK(hipMemset(zero, 0, 64 * 16));
    for (const Shape& sh : shapes) {
        const int NT = sh.N / 16, KT = sh.K / 32, S = sh.S;
        const size_t wn = (size_t)sh.N * sh.K, xn = (size_t)Mpad * sh.K, on = (size_t)S * Mpad * sh.N;
        bf16_t *W = nullptr, *X = nullptr;
        float *O0 = nullptr, *O1 = nullptr;
        LAB_CHECK(hipMalloc(&W, wn * 2 * L)); LAB_CHECK(hipMalloc(&X, xn * 2));
        LAB_CHECK(hipMalloc(&O0, on * 4)); LAB_CHECK(hipMalloc(&O1, on * 4));
        launch_synth_fill_bf16(W, wn * L, 0x1234u + sh.N, 0.02f, 0, s);
        launch_synth_fill_bf16(X, xn, 0x9876u + sh.K, 1.0f, 0, s);
        LAB_CHECK(hipStreamSynchronize(s));
        const double mb = (double)wn * 2 / 1e6;
        std::vector<float> ref(on), got(on);
        // the product launcher on layer 0 is the reference of every variant
        launch_gemm_skinny(EPI_PARTIAL, 2, 4, W, X, O0, NT, KT, S, sh.N, Mpad, s);
        LAB_CHECK(hipMemcpyAsync(ref.data(), O0, on * 4, hipMemcpyDeviceToHost, s));
        LAB_CHECK(hipStreamSynchronize(s));
        auto report = [&](const char* variant, double us, double err) {
            printf("{\"shape\": \"%s\", \"N\": %d, \"K\": %d, \"S\": %d, \"rows\": %d, \"variant\": \"%s\", \"us\": %.2f, \"MB\": %.2f, \"GBps\": %.1f, "
                   "\"max_rel_vs_product\": %.3g}\n", sh.name, sh.N, sh.K, S, rows, variant, us, mb, mb / us * 1e3, err);
            fflush(stdout);
        };
        report("product k_gemm_skinny R2 KSB4", time_launches([&](int i) {
                   launch_gemm_skinny(EPI_PARTIAL, 2, 4, W + (size_t)(i % L) * wn, X, O0, NT, KT, S, sh.N, Mpad, s); }, iters, s), 0.0);
#define LAB_RUN(LABEL, KERNEL, RR, KSBV, ...)                                                                            \
        {                                                                                                                \
            const int n_items = ((NT + RR - 1) / RR) * S;                                                                \
            LAB_CHECK(hipMemsetAsync(O1, 0, on * 4, s));                                                                 \
            hipLaunchKernelGGL(KERNEL, dim3(n_items), dim3(KSBV * 64), 0, s, W, X, __VA_ARGS__ O1, NT, KT, S, n_items, sh.N, Mpad); \
            LAB_CHECK(hipGetLastError());                                                                                \
            LAB_CHECK(hipMemcpyAsync(got.data(), O1, on * 4, hipMemcpyDeviceToHost, s));                                 \
            LAB_CHECK(hipStreamSynchronize(s));                                                                          \
            const double err = max_rel_diff(got, ref);                                                                   \
            report(LABEL, time_launches([&](int i) {                                                                     \
                       hipLaunchKernelGGL(KERNEL, dim3(n_items), dim3(KSBV * 64), 0, s, W + (size_t)(i % L) * wn, X, __VA_ARGS__ O1, NT, KT, S, \
                                          n_items, sh.N, Mpad); }, iters, s), err);                                       \
        }
        LAB_RUN("stream R2 KSB4 U4 (the product's shape, scalar wave index)", (k_lab_stream<2, 2, 4, 4>), 2, 4, )
        LAB_RUN("stream R4 KSB4 U3", (k_lab_stream<2, 4, 4, 3>), 4, 4, )          /* U = 4 at R = 4 spills (68 bytes of scratch) */
        LAB_RUN("stream R4 KSB8 U3", (k_lab_stream<2, 4, 8, 3>), 4, 8, )
        LAB_RUN("stream R4 KSB8 U2", (k_lab_stream<2, 4, 8, 2>), 4, 8, )
        LAB_RUN("stream R2 KSB8 U4", (k_lab_stream<2, 2, 8, 4>), 2, 8, )
        {
            const int share4 = ((KT + S - 1) / S + 3) / 4, share8 = ((KT + S - 1) / S + 7) / 8;     // k-tiles of the longest wave share
            if (share4 <= 8) LAB_RUN("one-shot R2 KSB4 UK8", (k_lab_oneshot<2, 2, 4, 8>), 2, 4, zero,)
            if (share4 <= 12 && share4 > 8) LAB_RUN("one-shot R2 KSB4 UK12", (k_lab_oneshot<2, 2, 4, 12>), 2, 4, zero,)
            if (share8 <= 8) LAB_RUN("one-shot R2 KSB8 UK8", (k_lab_oneshot<2, 2, 8, 8>), 2, 8, zero,)
            if (share8 <= 12 && share8 > 8) LAB_RUN("one-shot R2 KSB8 UK12", (k_lab_oneshot<2, 2, 8, 12>), 2, 8, zero,)
            if (share8 <= 6) LAB_RUN("one-shot R4 KSB8 UK6", (k_lab_oneshot<2, 4, 8, 6>), 4, 8, zero,)
        }
#undef LAB_RUN
        LAB_CHECK(hipFree(W)); LAB_CHECK(hipFree(X)); LAB_CHECK(hipFree(O0)); LAB_CHECK(hipFree(O1));
    }
    LAB_CHECK(hipFree(zero));
    return 0;
}
