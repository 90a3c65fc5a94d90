// Laboratory probe for a batch-1 persistent decode kernel (round 5): what ONE XCD - or 2 / 4 / 8 of them - can do for a
// single row, before any engine is built.  Two numbers per worker-set size W = 32 X (X = 1, 2, 4, 8 XCDs):
//   stream   GB/s of W workgroups (one per CU) reading a 160 MB model-sized buffer with non-temporal 16-byte loads, pass after pass
//            (the buffer fits the 256 MB Infinity Cache: what a small LM's weights do token after token), and a 1.28 GB rotation (HBM)
//   exchange microseconds per all-to-all edge inside ONE launch: every worker publishes its slice of an n-value vector (8-byte
//            agent-scope stores), arrives on a monotonic counter, polls it, gathers the whole vector (8-byte agent-scope loads); the
//            gathered values feed the next publication (a dependent chain, like the phases of a decode layer); contents checked.
// Workers are the blocks whose index mod 8 is below X (observed placement: block b runs on XCD b mod 8 - used for speed only; the
// protocol is placement-independent).  The XCC id each worker really ran on is reported.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/token_engine_lab/probe.hip -o tools/token_engine_lab/probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int worker_of(int X, int* W) {          // -1: this block is not a worker
    const int b = blockIdx.x, xcc = b & 7;
    *W = (gridDim.x >> 3) * X;
    return xcc < X ? (b >> 3) * X + xcc : -1;
}

// ---- stream: worker w reads chunks w, w + W, ... of `n16` 16-byte pieces, 8 loads in flight per lane
__global__ void __launch_bounds__(256) k_stream(const u32x4* __restrict__ buf, size_t n16, int X, unsigned* sink) {
    extern __shared__ unsigned char lds_pad[];
    if (n16 == ~(size_t)0) lds_pad[threadIdx.x] = 0;
    int W;
    const int w = worker_of(X, &W);
    if (w < 0) return;
    u32x4 acc = {0, 0, 0, 0};
    const size_t per = 256 * 8;                                     // pieces per block iteration
    for (size_t base = (size_t)w * per; base + per <= n16; base += (size_t)W * per) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(buf + base + j * 256 + threadIdx.x);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

// ---- exchange
struct XArgs { u64* vec; unsigned* counter; int n8; int iters; int X; unsigned* xcc_out; u64* bad; int spin; };
__global__ void __launch_bounds__(256) k_exchange(XArgs a) {
    extern __shared__ unsigned char lds_pad[];
    __shared__ u64 s_sum;
    __shared__ int s_ok;
    if (a.n8 < 0) lds_pad[threadIdx.x] = 0;
    int W;
    const int w = worker_of(a.X, &W);
    if (w < 0) return;
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        a.xcc_out[w] = id & 0xf;
    }
    const int n8 = a.n8;                                            // 8-byte granules in the vector
    const int per = (n8 + W - 1) / W, g0 = w * per, g1 = min(n8, g0 + per);
    u64 seed = 1;                                                   // what the previous gather produced (same on every worker)
    for (int it = 0; it < a.iters; ++it) {
        u64* vec = a.vec + (size_t)(it & 1) * n8;
        for (int g = g0 + tid; g < g1; g += 256)
            __hip_atomic_store(vec + g, seed * 0x9E3779B97F4A7C15ull + (u64)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)W * (unsigned)(it + 1);
            int ok = 0;
            for (int p = 0; p < a.spin; ++p) {
                if ((int)(__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            s_ok = ok;
            s_sum = 0;
        }
        __syncthreads();
        if (!s_ok) { if (tid == 0) atomicAdd(a.bad, 1ull << 32); return; }            // timed out: give up (bounded spin)
        u64 part = 0;
        for (int g = tid; g < n8; g += 256) part += __hip_atomic_load(vec + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int o = 32; o; o >>= 1) part += __shfl_xor(part, o);
        if ((tid & 63) == 0) atomicAdd(&s_sum, part);
        __syncthreads();
        // expected: sum over g of (seed * C + g)
        const u64 expect = seed * 0x9E3779B97F4A7C15ull * (u64)n8 + (u64)n8 * (u64)(n8 - 1) / 2;
        if (tid == 0 && s_sum != expect) atomicAdd(a.bad, 1ull);
        seed = s_sum | 1;
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    hipDeviceProp_t prop{};
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int grid = cus / 8 * 8;
    const size_t lds = 96 * 1024;                                   // one block per CU
    CK(hipFuncSetAttribute((const void*)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)k_exchange, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned* sink; CK(hipMalloc(&sink, 64)); CK(hipMemset(sink, 0, 64));
    printf("{\"device\": \"%s\", \"cus\": %d", prop.gcnArchName, cus);
    // ---- stream
    const size_t model = (size_t)160 << 20, big = (size_t)1280 << 20;
    u32x4* buf; CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 1, big));
    printf(", \"stream\": [");
    bool first = true;
    for (int X : {1, 2, 4, 8}) {
        for (int rot = 0; rot < 2; ++rot) {
            const int reps = 20;
            auto pass = [&](int r) {
                const u32x4* p = rot ? buf + (size_t)(r % 8) * (model / 16) : buf;
                hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), lds, s, p, model / 16, X, sink);
            };
            for (int r = 0; r < 3; ++r) pass(r);
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) pass(r);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s{\"xcds\": %d, \"workers\": %d, \"source\": \"%s\", \"us_per_160MB\": %.1f, \"GBps\": %.0f}", first ? "" : ", ", X, grid / 8 * X,
                   rot ? "hbm (8 x 160 MB rotation)" : "infinity cache (same 160 MB)", ms * 1e3 / reps, model / (ms * 1e-3 / reps) / 1e9);
            first = false;
        }
    }
    printf("]");
    // ---- exchange
    u64* vec; CK(hipMalloc(&vec, 2 * 65536 * 8));
    unsigned* counter; CK(hipMalloc(&counter, 256));
    unsigned* xcc; CK(hipMalloc(&xcc, 256 * 4));
    u64* bad; CK(hipMalloc(&bad, 8));
    printf(", \"exchange\": [");
    first = true;
    for (int X : {1, 2, 4, 8}) {
        for (int n8 : {128, 576, 8192}) {                           // 1 KB (512 bf16), 4.6 KB (2304 bf16), 64 KB (32 x 512 f32 partials)
            const int iters = 400;
            XArgs a{vec, counter, n8, iters, X, xcc, bad, 1 << 16};
            float best = 1e9f;
            u64 hbad = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(counter, 0, 256, s)); CK(hipMemsetAsync(bad, 0, 8, s)); CK(hipMemsetAsync(xcc, 0xff, 256 * 4, s));
                CK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(k_exchange, dim3(grid), dim3(256), lds, s, a);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
                u64 hb; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
                hbad |= hb;
            }
            std::vector<unsigned> hx(256);
            CK(hipMemcpy(hx.data(), xcc, 256 * 4, hipMemcpyDeviceToHost));
            int seen[16] = {0};
            const int Wn = grid / 8 * X;
            for (int i = 0; i < Wn; ++i) if (hx[i] < 16) seen[hx[i]]++;
            int distinct = 0; for (int i = 0; i < 16; ++i) distinct += seen[i] > 0;
            printf("%s{\"xcds\": %d, \"workers\": %d, \"vector_bytes\": %d, \"us_per_edge\": %.3f, \"wrong_sums\": %llu, \"timeouts\": %llu, \"xcc_ids_seen\": %d}",
                   first ? "" : ", ", X, Wn, n8 * 8, best * 1e3 / iters, hbad & 0xffffffffull, hbad >> 32, distinct);
            first = false;
        }
    }
    printf("]}\n");
    return 0;
}
