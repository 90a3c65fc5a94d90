// lm_sampler.hip - on-device logit processing + sampling ("mis-sampler-v1", spec in oracle/sampler.py).
//
// Replaces the per-token host round trip of the reference loop (LlamaTTS.swift:717-723:
// processor.process -> sampler.sample -> .item()): repetition penalty, temperature, nucleus cut and the
// categorical draw run in ONE kernel per step, one 512-thread block per utterance, and the sampled id
// never leaves the GPU (it is written straight into next_ids for the following forward pass).
//
// Every quantity after e_i is an exact integer, so the result is independent of reduction order and
// bit-identical to the numpy oracle:
//   x_i = fdiv(l_i, T); y_i = x_i - max x; e_i = det_exp(y_i) (0 if masked); E_i = trunc(e_i * 2^40)
//   key_i = bits(e_i) >> 16 ;  Z = sum E ; thr = u64(double(1 - topP) * double(Z))
//   k* = min{k : sum_{key_j <= k} E_j > thr} ;  K = {i : key_i >= k*, E_i > 0}
//   r = mulhi64(rand64(seed,row,step), Z_K) ; token = first i in K, in LANE-MAJOR order
//   (i mod 512, i div 512), whose running sum exceeds r.
#include "common.h"
#include "lm_kernels.h"

#define SAMP_NT 512
#define ORPHEUS_AUDIO_OFFSET 128266

typedef unsigned long long u64;

__device__ __forceinline__ float det_exp_dev(float y) {
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

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 block_sum_u64(u64 v, u64* red) {      // red[SAMP_NT/64]
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += shfl_xor_u64(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    u64 t = 0;
#pragma unroll
    for (int w = 0; w < SAMP_NT / 64; ++w) t += red[w];
    __syncthreads();
    return t;
}
__device__ __forceinline__ float block_max_f32(float v, float* red) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int w = 1; w < SAMP_NT / 64; ++w) t = fmaxf(t, red[w]);
    __syncthreads();
    return t;
}

__global__ void __launch_bounds__(SAMP_NT) k_sampler(SamplerParams p) {
    __shared__ u64 bins[8][SAMP_NT];       // 32 KiB: per-thread private mass bins / scan scratch
    __shared__ u64 red64[SAMP_NT / 64];
    __shared__ float redf[SAMP_NT / 64];
    __shared__ int redi[SAMP_NT / 64];
    __shared__ u64 s_cum;
    __shared__ unsigned s_prefix;
    __shared__ int s_token;

    const int b = blockIdx.x, tid = threadIdx.x;
    if (p.active_in && !p.active_in[b]) return;
    bf16_t* logits = p.logits + (size_t)b * p.Vpad;
    float* ebuf = p.e_buf + (size_t)b * p.Vpad;
    const int V = p.vocab;
    const int step = p.step_override ? p.step_override[b] : p.n_gen[b];
    if (!p.step_override && step >= p.max_tokens) return;

    int lo = p.lo, hi = (p.hi <= 0 || p.hi > V) ? V : p.hi;
    if (p.frame_constrained) {
        lo = ORPHEUS_AUDIO_OFFSET + (step % 7) * 4096;
        hi = lo + 4096;
        if (hi > V) hi = V;
        if (lo > hi) lo = hi;
    }

    // ---- repetition penalty, once per unique id of the window (RepetitionContext.process)
    if (p.penalty > 0.0f && p.penalty != 1.0f && p.window) {
        int wl = p.window_len[b];
        const int32_t* win = p.window + (size_t)b * p.ctx + (p.ctx - wl);
        if (tid < wl) {
            int id = win[tid];
            bool first = (id >= 0 && id < V);
            for (int j = 0; j < tid; ++j) first = first && (win[j] != id);
            if (first) {
                float pen = bf16_round_f32(p.penalty);            // scalar weakly typed to bf16
                float l = bf16_to_f32(logits[id]);
                float v = (l < 0.0f) ? l * pen : __fdiv_rn(l, pen);
                logits[id] = f32_to_bf16(v);
            }
        }
        __syncthreads();
    }

    int token;
    if (p.temperature == 0.0f) {
        // ---- greedy: argmax, first index on ties
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < V; i += SAMP_NT) {
            if (i < lo || i >= hi) continue;
            float l = bf16_to_f32(logits[i]);
            if (l > best || (l == best && i < bi)) { best = l; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o, 64);
            int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if ((tid & 63) == 0) { redf[tid >> 6] = best; redi[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            float bb = redf[0];
            int ii = redi[0];
            for (int w = 1; w < SAMP_NT / 64; ++w)
                if (redf[w] > bb || (redf[w] == bb && redi[w] < ii)) { bb = redf[w]; ii = redi[w]; }
            s_token = (ii == 0x7fffffff) ? lo : ii;
        }
        __syncthreads();
        token = s_token;
    } else {
        // ---- pass A: max logit over the allowed range (fdiv is monotone: max x = fdiv(max l, T))
        float lmax = -INFINITY;
        for (int i = tid; i < V; i += SAMP_NT)
            if (i >= lo && i < hi) lmax = fmaxf(lmax, bf16_to_f32(logits[i]));
        lmax = block_max_f32(lmax, redf);
        const float xmax = __fdiv_rn(lmax, p.temperature);
        // ---- pass B: e_i, Z  (every vocabulary entry is visited; masked ones get e = 0)
        u64 zloc = 0;
        for (int i = tid; i < V; i += SAMP_NT) {
            float e = 0.0f;
            float x = __fdiv_rn(bf16_to_f32(logits[i]), p.temperature);
            float y = fminf(x - xmax, 0.0f);
            float ee = det_exp_dev(y);
            if (i >= lo && i < hi) e = ee;
            ebuf[i] = e;
            zloc += (u64)(e * 1099511627776.0f);
        }
        const u64 Z = block_sum_u64(zloc, red64);
        // ---- nucleus threshold: 5 radix passes (3 bits each) over key = bits(e) >> 16
        unsigned kstar = 0;
        if (p.top_p > 0.0f && p.top_p < 1.0f) {
            const u64 thr = (u64)((double)(1.0f - p.top_p) * (double)Z);
            if (tid == 0) { s_cum = 0; s_prefix = 0; }
            __syncthreads();
            for (int pass = 0; pass < 5; ++pass) {
                const int shift = 12 - 3 * pass;
                const unsigned prefix = s_prefix;
#pragma unroll
                for (int q = 0; q < 8; ++q) bins[q][tid] = 0;
                for (int i = tid; i < V; i += SAMP_NT) {
                    float e = ebuf[i];
                    unsigned key = __float_as_uint(e) >> 16;
                    if ((key >> (shift + 3)) == prefix) {
                        u64 E = (u64)(e * 1099511627776.0f);
                        bins[(key >> shift) & 7][tid] += E;
                    }
                }
                __syncthreads();
                for (int st = SAMP_NT / 2; st > 0; st >>= 1) {
                    if (tid < st) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) bins[q][tid] += bins[q][tid + st];
                    }
                    __syncthreads();
                }
                if (tid == 0) {
                    u64 cum = s_cum;
                    int chosen = 7;
                    for (int q = 0; q < 8; ++q) {
                        if (cum + bins[q][0] > thr) { chosen = q; break; }
                        cum += bins[q][0];
                    }
                    s_cum = cum;
                    s_prefix = (prefix << 3) | (unsigned)chosen;
                }
                __syncthreads();
            }
            kstar = s_prefix;
        }
        // ---- kept mass per lane (lane-major order), exclusive scan over lanes, inverse CDF
        u64 mine = 0;
        for (int i = tid; i < V; i += SAMP_NT) {
            float e = ebuf[i];
            if ((__float_as_uint(e) >> 16) >= kstar) mine += (u64)(e * 1099511627776.0f);
        }
        u64* scan = &bins[0][0];
        scan[tid] = mine;
        __syncthreads();
        for (int o = 1; o < SAMP_NT; o <<= 1) {
            u64 t = (tid >= o) ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += t;
            __syncthreads();
        }
        const u64 Zk = scan[SAMP_NT - 1];
        const u64 incl = scan[tid], excl = incl - mine;
        const u64 row = (u64)(p.row_offset + b);
        u64 a = p.seed ^ (0xD1B54A32D192ED03ull * (row + 1));
        u64 rnd = mis_splitmix64(mis_splitmix64(a) + (u64)step);
        const u64 r = __umul64hi(rnd, Zk);
        if (tid == 0) s_token = lo;
        __syncthreads();
        if (mine > 0 && r >= excl && r < incl) {
            u64 run = excl;
            for (int i = tid; i < V; i += SAMP_NT) {
                float e = ebuf[i];
                if ((__float_as_uint(e) >> 16) >= kstar) {
                    run += (u64)(e * 1099511627776.0f);
                    if (run > r) { s_token = i; break; }
                }
            }
        }
        __syncthreads();
        token = s_token;
    }

    // ---- bookkeeping (generate loop, LlamaTTS.swift:721-738)
    if (tid == 0) {
        if (p.tokens_out && step < p.tokens_stride) p.tokens_out[(size_t)b * p.tokens_stride + step] = token;
        if (p.n_gen && !p.step_override) p.n_gen[b] = step + 1;
        if (p.window && p.ctx > 0) {       // didSample: slide the ring (kept right-aligned)
            int32_t* win = p.window + (size_t)b * p.ctx;
            int wl = p.window_len[b];
            if (wl < p.ctx) { wl++; p.window_len[b] = wl; }
            for (int j = p.ctx - wl; j < p.ctx - 1; ++j) win[j] = win[j + 1];
            win[p.ctx - 1] = token;
        }
        if (p.next_ids) p.next_ids[b] = token;
        if (token == p.eos_id) {
            if (p.active) p.active[b] = 0;
            if (p.done_count) atomicAdd(p.done_count, 1);
        } else {
            if (p.all_ids) {
                int n = p.all_len[b];
                if (n < p.all_stride) { p.all_ids[(size_t)b * p.all_stride + n] = token; p.all_len[b] = n + 1; }
            }
            if (p.n_gen && !p.step_override && step + 1 >= p.max_tokens) {
                if (p.active) p.active[b] = 0;            // budget exhausted (LlamaTTS.swift:714)
                if (p.done_count) atomicAdd(p.done_count, 1);
            }
        }
    }
}

void launch_sampler(const SamplerParams& p, int batch, hipStream_t s) {
    hipLaunchKernelGGL(k_sampler, dim3(batch), dim3(SAMP_NT), 0, s, p);
}
