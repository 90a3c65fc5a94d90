// q3_sampler.hip - Qwen3-TTS `sampleToken` (Qwen3TTS.swift:1003-1118) on the device, one 256-thread block per row.
//
// The codec vocabularies are small (talker 3072, code predictor 2048), so a row's logits live in registers (16 per thread)
// and every selection step is a 256-bin radix pass over the 16-bit order-preserving key of the bf16 logit - exact, because a
// bf16 value IS its 16-bit key.  Specification: oracle/qwen3tts.py::sample_token (deterministic realisation of the
// reference's set semantics; the categorical draw is the inverse CDF on fixed-point masses of "mis-sampler-v1").
//   suppress -> repetition penalty (unique generated ids, bf16 arithmetic) -> [greedy] -> top-k (ties at the k-th value
//   kept) -> top-p on softmax(filtered) at temperature 1 -> min-p -> EOS logit restored -> categorical(T(l / T(temp)))
#include "common.h"
#include "q3_kernels.h"

typedef unsigned long long u64;
#define Q3S_NT 256                 // four waves: a block barrier among 4 waves costs a fraction of one among 16 (the kernel is a chain of ~25 of them)
#define Q3S_PER 16
#define E_SCALE 1099511627776.0f

__device__ __forceinline__ float q3_det_exp(float y) {
#pragma clang fp contract(off)
    const float LOG2E = 1.4426950408889634f;
    float t = y * LOG2E;
    float n = floorf(t);
    float f = t - n;
    float p = 0.00015403530393381608f;
    p = p * f; p = p + 0.0013333558146428443f;
    p = p * f; p = p + 0.009618129107628477f;
    p = p * f; p = p + 0.05550410866482158f;
    p = p * f; p = p + 0.2402265069591007f;
    p = p * f; p = p + 0.6931471805599453f;
    p = p * f; p = p + 1.0f;
    int ni = (int)fmaxf(n, -64.0f);
    float r = p * ldexpf(1.0f, ni);
    return (n < -60.0f) ? 0.0f : r;
}

__device__ __forceinline__ unsigned q3_key(float v) {          // order-preserving 16-bit key of a bf16-valued float
    unsigned u = __float_as_uint(v) >> 16;
    return (u & 0x8000u) ? (~u & 0xFFFFu) : (u | 0x8000u);
}

// inclusive scan inside a wave (integer sums: exact whatever the order).  Every lane of the wave must call it.
__device__ __forceinline__ u64 q3_wave_scan(u64 v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u64 t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
// inclusive scan over the first `n_waves` waves of the block (n_waves * 64 values, one per thread; other threads pass 0); result left in
// sh[0 .. 64 n_waves).  Wave-level shuffles + one exchange of the wave totals: two barriers instead of two per doubling step (the
// 256-bin and 1024-entry Hillis-Steele scans were ~60 of the kernel's ~80 block barriers, 14.9 us per launch, 16 launches per frame)
__device__ __forceinline__ void q3_scan_waves(u64 v, u64* sh, u64* part, int n_waves) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 s = q3_wave_scan(wave < n_waves ? v : 0);
    if (wave < n_waves && lane == 63) part[wave] = s;
    __syncthreads();
    if (wave < n_waves) {
        u64 off = 0;
        for (int w = 0; w < wave; ++w) off += part[w];
        sh[tid] = s + off;
    }
    __syncthreads();
}
__device__ __forceinline__ void q3_scan256(u64 v, u64* sh, u64* part) { q3_scan_waves(v, sh, part, 4); }

// smallest bin index with sh[bin] > thr (sh ascending inclusive prefix); 256 if none.  All threads get the result.
__device__ __forceinline__ int q3_first_above(const u64* sh, u64 thr, int* slot) {
    const int tid = threadIdx.x;
    if (tid == 0) *slot = 256;
    __syncthreads();
    if (tid < 256 && sh[tid] > thr && (tid == 0 || sh[tid - 1] <= thr)) *slot = tid;
    __syncthreads();
    int r = *slot;
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(Q3S_NT) k_q3_sample(Q3SampleArgs a) {
    __shared__ u64 sh[Q3S_NT];
    __shared__ u64 hist[256];
    __shared__ u64 part[Q3S_NT / 64];
    __shared__ float redf[Q3S_NT / 64];
    __shared__ int redi[Q3S_NT / 64];
    __shared__ int slot;
    __shared__ float s_eos;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!a.active_a[b]) return;
    const bf16_t* lg = a.logits + (size_t)b * a.Vpad;
    uint8_t* seen = a.seen ? a.seen + (size_t)b * a.Vpad : nullptr;
    const float pen = bf16_round_f32(a.penalty);
    float l[Q3S_PER];
#pragma unroll
    for (int e = 0; e < Q3S_PER; ++e) {
        int i = tid * Q3S_PER + e;
        float v = -INFINITY;
        if (i < a.V) {
            v = bf16_to_f32(lg[i]);
            if (i >= a.sup_lo && i < a.sup_hi && i != a.eos) v = -INFINITY;
            else if (seen && a.penalty != 1.0f && seen[i]) v = bf16_round_f32(v < 0.0f ? v * pen : __fdiv_rn(v, pen));
        }
        l[e] = v;
    }
    auto block_max = [&](float v, int idx, int& out_idx) {            // max with the smallest index on ties
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(v, o, 64);
            int oi = __shfl_xor(idx, o, 64);
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
        if (lane == 0) { redf[wave] = v; redi[wave] = idx; }
        __syncthreads();
        float bv = redf[0]; int bi = redi[0];
        for (int w = 1; w < Q3S_NT / 64; ++w)
            if (redf[w] > bv || (redf[w] == bv && redi[w] < bi)) { bv = redf[w]; bi = redi[w]; }
        __syncthreads();
        out_idx = bi;
        return bv;
    };
    auto thread_max = [&](const float* v, int& idx) {
        float m = -INFINITY; idx = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < Q3S_PER; ++e) {
            int i = tid * Q3S_PER + e;
            if (i < a.V && (v[e] > m || idx == 0x7fffffff)) { if (v[e] > m || idx == 0x7fffffff) { m = v[e]; idx = i; } }
        }
        return m;
    };
    int token = -1;
    if (a.temperature <= 0.0f) {                                       // argMax(logitsSlice) (:1036-1038)
        int ti, bi;
        float tm = thread_max(l, ti);
        block_max(tm, ti, bi);
        token = bi;
    } else {
        if (a.eos >= 0 && a.eos < a.V && tid == a.eos / Q3S_PER) s_eos = l[a.eos % Q3S_PER];
        // ---- top-k: k-th largest key via two 256-bin count passes from the top
        if (a.top_k > 0 && a.top_k < a.V) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < Q3S_PER; ++e)
                if (tid * Q3S_PER + e < a.V) atomicAdd(&hist[255 - (q3_key(l[e]) >> 8)], 1ull);
            __syncthreads();
            q3_scan256(tid < 256 ? hist[tid] : 0, sh, part);
            int r1 = q3_first_above(sh, (u64)a.top_k - 1, &slot);      // first reversed bin whose cumulative count >= k
            u64 above = (r1 > 0 && r1 < 256) ? sh[r1 - 1] : 0;
            __syncthreads();
            const unsigned B1 = 255u - (unsigned)r1;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < Q3S_PER; ++e) {
                unsigned k = q3_key(l[e]);
                if (tid * Q3S_PER + e < a.V && (k >> 8) == B1) atomicAdd(&hist[255 - (k & 255)], 1ull);
            }
            __syncthreads();
            q3_scan256(tid < 256 ? hist[tid] : 0, sh, part);
            int r2 = q3_first_above(sh, (u64)a.top_k - 1 - above, &slot);
            const unsigned kth = (B1 << 8) | (255u - (unsigned)r2);
#pragma unroll
            for (int e = 0; e < Q3S_PER; ++e)
                if (q3_key(l[e]) < kth) l[e] = -INFINITY;
        }
        // ---- top-p on softmax(l) at temperature 1: drop the value groups whose ascending cumulative mass stays <= (1-p) Z
        if (a.top_p > 0.0f && a.top_p < 1.0f) {
            int ti, bi;
            float m1 = block_max(thread_max(l, ti), ti, bi);
            u64 E[Q3S_PER], loc = 0;
#pragma unroll
            for (int e = 0; e < Q3S_PER; ++e) {
                float ee = (l[e] == -INFINITY) ? 0.0f : q3_det_exp(fminf(fmaxf(l[e] - m1, -100.0f), 0.0f));
                E[e] = (tid * Q3S_PER + e < a.V) ? (u64)(ee * E_SCALE) : 0;
                loc += E[e];
            }
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < Q3S_PER; ++e)
                if (E[e]) atomicAdd(&hist[q3_key(l[e]) >> 8], E[e]);
            __syncthreads();
            q3_scan256(tid < 256 ? hist[tid] : 0, sh, part);
            const u64 Z = sh[255];
            const u64 thr = (u64)((double)(1.0f - a.top_p) * (double)Z);
            int b1 = q3_first_above(sh, thr, &slot);
            u64 below = (b1 > 0 && b1 < 256) ? sh[b1 - 1] : 0;
            __syncthreads();
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < Q3S_PER; ++e) {
                unsigned k = q3_key(l[e]);
                if (E[e] && (int)(k >> 8) == b1) atomicAdd(&hist[k & 255], E[e]);
            }
            __syncthreads();
            q3_scan256(tid < 256 ? hist[tid] : 0, sh, part);
            int b2 = q3_first_above(sh, thr - below, &slot);
            const unsigned kstar = ((unsigned)b1 << 8) | (unsigned)(b2 & 255);
#pragma unroll
            for (int e = 0; e < Q3S_PER; ++e)
                if (q3_key(l[e]) < kstar) l[e] = -INFINITY;
            (void)loc;
        }
        // ---- min-p: remove l < T(max + T(log(min_p)))
        if (a.min_p > 0.0f) {
            int ti, bi;
            float m2 = block_max(thread_max(l, ti), ti, bi);
            const float lim = bf16_round_f32(m2 + a.log_min_p);
#pragma unroll
            for (int e = 0; e < Q3S_PER; ++e)
                if (l[e] < lim) l[e] = -INFINITY;
        }
        __syncthreads();
        if (a.eos >= 0 && a.eos < a.V && tid == a.eos / Q3S_PER) l[a.eos % Q3S_PER] = s_eos;   // EOS stays sample-able (:1041-1046,1107-1110)
        // ---- categorical(T(l / T(temp))): inverse CDF in index order on fixed-point masses
        const float tb = bf16_round_f32(a.temperature);
        float x[Q3S_PER];
#pragma unroll
        for (int e = 0; e < Q3S_PER; ++e) x[e] = (l[e] == -INFINITY) ? -INFINITY : bf16_round_f32(__fdiv_rn(l[e], tb));
        int ti, bi;
        float m3 = block_max(thread_max(x, ti), ti, bi);
        u64 E[Q3S_PER], loc = 0;
#pragma unroll
        for (int e = 0; e < Q3S_PER; ++e) {
            float ee = (x[e] == -INFINITY) ? 0.0f : q3_det_exp(fminf(fmaxf(x[e] - m3, -100.0f), 0.0f));
            E[e] = (tid * Q3S_PER + e < a.V) ? (u64)(ee * E_SCALE) : 0;
            loc += E[e];
        }
        q3_scan_waves(loc, sh, part, Q3S_NT / 64);
        const u64 Z = sh[Q3S_NT - 1];
        const u64 row = (u64)(a.row_offset + b);
        const u64 step = (u64)(*a.frame) * (u64)a.G + (u64)a.slot;
        u64 sa = a.seed ^ (0xD1B54A32D192ED03ull * (row + 1));
        const u64 rnd = mis_splitmix64(mis_splitmix64(sa) + step);
        const u64 r = __umul64hi(rnd, Z);
        const u64 incl = sh[tid], excl = incl - loc;
        if (tid == 0) slot = -1;
        __syncthreads();
        if (r >= excl && r < incl) {
            u64 acc = excl;
            int pick = -1;
#pragma unroll
            for (int e = 0; e < Q3S_PER; ++e) {
                acc += E[e];
                if (pick < 0 && r < acc) pick = tid * Q3S_PER + e;
            }
            slot = pick;
        }
        __syncthreads();
        token = slot;
    }
    if (tid == 0) {
        if (token < 0) token = 0;
        if (a.eos >= 0 && token == a.eos) {                           // isEOS -> break before the frame is stored (:483-485)
            a.active_a[b] = 0;
            if (a.active_b) a.active_b[b] = 0;
            if (a.done_count) atomicAdd(a.done_count, 1);
        } else {
            a.cur_codes[(size_t)a.slot * a.Mpad + b] = token;
            if (seen) seen[token] = 1;
        }
        if (a.tokens_dbg) a.tokens_dbg[b] = token;
    }
}

void launch_q3_sample(const Q3SampleArgs& a, int batch, hipStream_t s) {
    MIS_REQUIRE(a.V <= Q3S_NT * Q3S_PER, MIS_ERR_INVALID_INPUT, "codec vocabulary %d exceeds the in-register sampler (%d)", a.V,
                Q3S_NT * Q3S_PER);
    hipLaunchKernelGGL(k_q3_sample, dim3(batch), dim3(Q3S_NT), 0, s, a);
}
