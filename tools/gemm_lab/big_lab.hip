// big_lab.hip - bench-side laboratory for the MFMA-bound GEMM of the Whisper encoder (csrc/whisper_kernels.hip, k_gemm_big2).
// NOT part of the product: candidate tile shapes and LDS ring depths are timed next to the library's launcher on the same buffers and
// compared with its output (bf16 C, BG_NONE epilogue, no bias), before the product kernel changes.
//
//   make -C tools/gemm_lab big_lab            (links against mlx-audio-swift_amd/libmi_speech.so)
//   tools/gemm_lab/big_lab [rows=12000] [iters=20]      -> one JSON line per (shape, variant)
//
// What is being tested.  k_gemm_big2 is 128 x 128 x 64 tiles on four waves with two LDS buffers: the
// LDS-DMA of the next tile is issued ONE k-step ahead (32 MFMAs per wave = 512 clocks; with two blocks per CU about 0.43 us) - less
// than an L2 / Infinity-Cache round trip - and a wave's fragment reads cost as many LDS clocks as its MFMAs cost MFMA clocks.
// k_big_ring<BM, BN, WM, WN, NBUF>: BM x BN x 64 tiles, WM x WN waves, an NBUF-deep LDS ring with the DMA NBUF - 1 tiles ahead.  That
// needs counted waits: the compiler puts vmcnt(0) in front of every LDS read that follows an LDS-DMA in program order, and
// __syncthreads() waits for all of them too - so the fragment reads are asm (ds_read_b128, one lgkmcnt(0) in front of the MFMAs that
// names every fragment), the barrier is the raw s_barrier, and the wait for tile kt is an explicit vmcnt(tiles still allowed in flight).
// Operand layout in LDS as in the product: [rows][64] bf16, 16-byte chunk c of row r stored at position c ^ ((r >> 1) & 7).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#include "common.h"
#include "lm_kernels.h"
#include "whisper_kernels.h"

#define LAB_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

#define BL_BK 64

__device__ __forceinline__ uint32_t lds_offset(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ bf16x8_t lds_read16(uint32_t addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

template <int BM, int BN, int WM, int WN, int NBUF>
__global__ void __launch_bounds__(WM * WN * 64, 1) k_big_ring(const bf16_t* X, const bf16_t* W, bf16_t* __restrict__ C, int M, int N, int K, int ldx) {
    constexpr int NW = WM * WN;                          // waves
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;  // 16 x 16 MFMA tiles per wave (m, n)
    constexpr int QW = BN / 8 / NW, QX = BM / 8 / NW;    // LDS-DMA instructions per wave and tile (8 rows of 128 B each)
    static_assert(BN % (8 * NW) == 0 && BM % (8 * NW) == 0, "tile rows must divide over the waves");
    extern __shared__ __attribute__((aligned(1024))) bf16_t lds[];
    bf16_t* Ws = lds;                                    // [NBUF][BN * 64]
    bf16_t* Xs = lds + (size_t)NBUF * BN * BL_BK;        // [NBUF][BM * 64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int wn = wave / WM, wm = wave % WM;

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const bf16_t* wsrc[QW];
    const bf16_t* xsrc[QX];
#pragma unroll
    for (int q = 0; q < QW; ++q) {
        const int r = (wave * QW + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        wsrc[q] = W + (size_t)min(n0 + r, N - 1) * K + chunk * 8;
    }
#pragma unroll
    for (int q = 0; q < QX; ++q) {
        const int r = (wave * QX + q) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        xsrc[q] = X + (size_t)min(m0 + r, M - 1) * ldx + chunk * 8;
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int q = 0; q < QW; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[q] + k0),
                                             (__attribute__((address_space(3))) void*)&Ws[((size_t)buf * BN + (wave * QW + q) * 8) * BL_BK], 16, 0, 0);
#pragma unroll
        for (int q = 0; q < QX; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[q] + k0),
                                             (__attribute__((address_space(3))) void*)&Xs[((size_t)buf * BM + (wave * QX + q) * 8) * BL_BK], 16, 0, 0);
    };
    const int KT = K / BL_BK;
    const int sw = (lane >> 1) & 7;
    const uint32_t ws_base = lds_offset(Ws) + (uint32_t)((wn * (BN / WN) + (lane & 15)) * BL_BK * 2);
    const uint32_t xs_base = lds_offset(Xs) + (uint32_t)((wm * (BM / WM) + (lane & 15)) * BL_BK * 2);

#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t)
        if (t < KT) stage(t, t * BL_BK);
    for (int kt = 0; kt < KT; ++kt) {
        // tile kt has landed when at most the DMAs of the NBUF - 2 younger tiles are outstanding (in-order vmcnt); near the end fewer
        // tiles were issued, there the wait is for everything
        if (kt + NBUF - 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NBUF - 2) * (QW + QX)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // every wave's part of tile kt is in LDS; buffer (kt - 1) % NBUF is free
        asm volatile("" ::: "memory");
        const int buf = kt % NBUF;
        bf16x8_t a[2][TN], b[2][TM];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint32_t pos = (uint32_t)(((ks * 4 + (lane >> 4)) ^ sw) * 16);
#pragma unroll
            for (int i = 0; i < TN; ++i) a[ks][i] = lds_read16(ws_base + (uint32_t)((buf * BN + i * 16) * BL_BK * 2) + pos);
#pragma unroll
            for (int j = 0; j < TM; ++j) b[ks][j] = lds_read16(xs_base + (uint32_t)((buf * BM + j * 16) * BL_BK * 2) + pos);
        }
        // the next DMA goes into the buffer read one k-step ago; it is issued behind this tile's reads and runs under its MFMAs
        if (kt + NBUF - 1 < KT) stage((kt + NBUF - 1) % NBUF, (kt + NBUF - 1) * BL_BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // one wait for the LDS reads, naming every fragment so that no MFMA is scheduled above it
#pragma unroll
            for (int i = 0; i < TN; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[ks][i]));
#pragma unroll
            for (int j = 0; j < TM; ++j) asm volatile("" : "+v"(b[ks][j]));
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
        }
    }
    // epilogue of k_gemm_big2 (BG_NONE, no bias): C/D lane = column m (l & 15), rows n = 4 (l >> 4) + e
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int n = n0 + wn * (BN / WN) + i * 16 + (lane >> 4) * 4;
        if (n >= N) continue;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = m0 + wm * (BM / WM) + j * 16 + (lane & 15);
            if (m >= M) continue;
            uint16_t res[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) res[e] = f32_to_bf16(acc[i][j][e]);
            bf16_t* o = C + (size_t)m * N + n;
            if (n + 3 < N) {
                uint2 v;
                v.x = (uint32_t)res[0] | ((uint32_t)res[1] << 16);
                v.y = (uint32_t)res[2] | ((uint32_t)res[3] << 16);
                *reinterpret_cast<uint2*>(o) = v;
            } else {
                for (int e = 0; e < 4 && n + e < N; ++e) o[e] = res[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------- harness
struct Shape { const char* name; int N, K; };

struct Ctx {
    Shape sh;
    int M, iters;
    bf16_t *X, *W, *C0, *C1;
    std::vector<uint16_t> ref, got;
    hipStream_t s;
};

template <typename F>
static double time_launches(F&& launch, int iters, hipStream_t s) {
    hipEvent_t a, b;
    LAB_CHECK(hipEventCreate(&a)); LAB_CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    LAB_CHECK(hipStreamSynchronize(s));
    LAB_CHECK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) launch();
    LAB_CHECK(hipEventRecord(b, s));
    LAB_CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    LAB_CHECK(hipEventElapsedTime(&ms, a, b));
    LAB_CHECK(hipEventDestroy(a)); LAB_CHECK(hipEventDestroy(b));
    return (double)ms * 1e3 / iters;
}

static float bf16_bits_to_float(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

static void report(const Ctx& c, const char* variant, double us, double err) {
    const double tflop = 2.0 * c.M * (double)c.sh.N * c.sh.K / 1e12;
    printf("{\"shape\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"variant\": \"%s\", \"us\": %.1f, \"TFLOPs\": %.1f, \"frac_of_2500\": %.3f, "
           "\"max_rel_vs_product\": %.3g}\n", c.sh.name, c.M, c.sh.N, c.sh.K, variant, us, tflop / (us * 1e-6), tflop / (us * 1e-6) / 2500.0, err);
    fflush(stdout);
}

template <int BM, int BN, int WM, int WN, int NBUF>
static void run_ring(Ctx& c) {
    const size_t lds_bytes = (size_t)NBUF * (BM + BN) * BL_BK * 2;
    auto kern = k_big_ring<BM, BN, WM, WN, NBUF>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        fprintf(stderr, "k_big_ring<%d,%d,%d,%d,%d>: %zu bytes of LDS refused\n", BM, BN, WM, WN, NBUF, lds_bytes);
        return;
    }
    const dim3 grid((c.sh.N + BN - 1) / BN, (c.M + BM - 1) / BM), block(WM * WN * 64);
    const size_t cn = (size_t)c.M * c.sh.N;
    auto launch = [&]() { hipLaunchKernelGGL(kern, grid, block, lds_bytes, c.s, c.X, c.W, c.C1, c.M, c.sh.N, c.sh.K, c.sh.K); };
    LAB_CHECK(hipMemsetAsync(c.C1, 0, cn * 2, c.s));
    launch();
    LAB_CHECK(hipGetLastError());
    LAB_CHECK(hipMemcpyAsync(c.got.data(), c.C1, cn * 2, hipMemcpyDeviceToHost, c.s));
    LAB_CHECK(hipStreamSynchronize(c.s));
    double worst = 0.0, scale = 0.0;
    for (size_t i = 0; i < cn; ++i) scale = fmax(scale, fabs((double)bf16_bits_to_float(c.ref[i])));
    for (size_t i = 0; i < cn; ++i) worst = fmax(worst, fabs((double)bf16_bits_to_float(c.got[i]) - (double)bf16_bits_to_float(c.ref[i])));
    char label[96];
    snprintf(label, sizeof label, "ring %dx%d tile, %dx%d waves, %d buffers (%zu KB LDS)", BM, BN, WM, WN, NBUF, lds_bytes / 1024);
    report(c, label, time_launches(launch, c.iters, c.s), scale > 0 ? worst / scale : worst);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 12000, iters = argc > 2 ? atoi(argv[2]) : 20;      // 8 x 1500 encoder positions
    hipStream_t s;
    LAB_CHECK(hipStreamCreate(&s));
    const Shape shapes[] = {{"qkv", 3840, 1280}, {"out_proj", 1280, 1280}, {"fc1", 5120, 1280}, {"fc2", 1280, 5120}};   // Whisper-large-v3 encoder
    for (const Shape& sh : shapes) {
        Ctx c;
        c.sh = sh; c.M = M; c.iters = iters; c.s = s;
        const size_t xn = (size_t)M * sh.K, wn = (size_t)sh.N * sh.K, cn = (size_t)M * sh.N;
        LAB_CHECK(hipMalloc(&c.X, xn * 2)); LAB_CHECK(hipMalloc(&c.W, wn * 2));
        LAB_CHECK(hipMalloc(&c.C0, cn * 2)); LAB_CHECK(hipMalloc(&c.C1, cn * 2));
        launch_synth_fill_bf16(c.X, xn, 0x5151u + sh.K, 1.0f, 0, s);
        launch_synth_fill_bf16(c.W, wn, 0x7171u + sh.N, 0.03f, 0, s);
        c.ref.resize(cn); c.got.resize(cn);
        BigGemmParams p{};
        p.X = c.X; p.W = c.W; p.bias = nullptr; p.R = nullptr; p.C = c.C0; p.M = M; p.N = sh.N; p.K = sh.K; p.ldx = sh.K; p.pos_rows = 1;
        launch_gemm_big(BG_NONE, p, s);
        LAB_CHECK(hipMemcpyAsync(c.ref.data(), c.C0, cn * 2, hipMemcpyDeviceToHost, s));
        LAB_CHECK(hipStreamSynchronize(s));
        report(c, "product k_gemm_big2 (128x128, 4 waves, 2 buffers, 2 blocks per CU)", time_launches([&]() { launch_gemm_big(BG_NONE, p, s); }, iters, s), 0.0);
        run_ring<128, 128, 2, 2, 2>(c);                   // the product's shape in this harness (one block per CU asked for: 1 wave per SIMD)
        run_ring<128, 128, 2, 2, 3>(c);
        run_ring<128, 128, 2, 2, 4>(c);
        run_ring<256, 128, 4, 2, 2>(c);
        run_ring<256, 128, 4, 2, 3>(c);
        run_ring<128, 256, 2, 4, 3>(c);
        run_ring<256, 256, 4, 2, 2>(c);
        run_ring<256, 256, 2, 4, 2>(c);
        LAB_CHECK(hipFree(c.X)); LAB_CHECK(hipFree(c.W)); LAB_CHECK(hipFree(c.C0)); LAB_CHECK(hipFree(c.C1));
    }
    return 0;
}
