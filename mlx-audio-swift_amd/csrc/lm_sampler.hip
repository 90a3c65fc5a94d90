// lm_sampler.hip - on-device logit processing + sampling ("mis-sampler-v1", spec in oracle/sampler.py).
//
// Replaces the per-token host round trip of the reference loop (LlamaTTS.swift:717-723:
// processor.process -> sampler.sample -> .item()): repetition penalty, temperature, nucleus cut and the
// categorical draw run on the device and the sampled id never leaves the GPU (it is written straight
// into next_ids for the following forward pass).
//
// v0 used one block per row and was latency bound (518 us per step).  Now every pass over the vocabulary
// is spread over (chunks x rows) blocks, six small kernels inside the step's hipGraph:
//   k_samp_prepare   repetition penalty in place (<= ctx ids per row) + scratch reset
//   k_samp_max       per-chunk max over the allowed ids                      (greedy: max + first index)
//   k_samp_exp       e_i = det_exp(fdiv(l_i,T) - max), E_i = trunc(e_i 2^40); 256-bin mass histogram of
//                    key_i >> 8 (LDS u64 atomics, flushed with global u64 atomics: exact integers)
//   k_samp_hist2     scan level-1 bins -> bin holding the nucleus boundary; 256-bin histogram of key & 255
//   k_samp_mass      scan level-2 bins -> k*; kept mass per chunk
//   k_samp_pick      r = mulhi64(rand64, Z_K); chunk, then token by inverse CDF in INDEX order; bookkeeping
// Every quantity after e_i is an exact integer, so the result is independent of reduction order and
// bit-identical to the numpy oracle:
//   key_i = bits(e_i) >> 16 ;  Z = sum E ; thr = u64(double(1 - topP) * double(Z))
//   k* = min{k : sum_{key_j <= k} E_j > thr} ;  K = {i : key_i >= k*, E_i > 0}
//   token = first i in K (index order) with prefix_K(E)_i > r
#include "common.h"
#include "lm_kernels.h"
#include "sampler_math.h"
#include <vector>
#include <atomic>
#include <mutex>

#define SAMP_NT 256
#define ORPHEUS_AUDIO_OFFSET 128266

typedef unsigned long long u64;
static_assert(SAMP_MAX_CHUNKS <= 64, "per-chunk quantities are reduced one per lane");


__device__ __forceinline__ void allowed_range(const SamplerParams& p, int step, int& lo, int& hi) {
    const int V = p.vocab;
    lo = p.lo;
    hi = (p.hi <= 0 || p.hi > V) ? V : p.hi;
    if (p.frame_constrained) {
        lo = (p.audio_offset > 0 ? p.audio_offset : ORPHEUS_AUDIO_OFFSET) + (step % 7) * 4096;
        hi = lo + 4096;
        if (hi > V) hi = V;
        if (lo > hi) lo = hi;
    }
    if (lo < 0) lo = 0;
}
__device__ __forceinline__ int row_step(const SamplerParams& p, int b) {
    return p.step_override ? p.step_override[b] : p.n_gen[b];
}
__device__ __forceinline__ bool row_skipped(const SamplerParams& p, int b) {
    if (p.active_in && !p.active_in[b]) return true;
    return !p.step_override && p.n_gen[b] >= p.max_tokens;
}
__device__ __forceinline__ void chunk_range(const SamplerParams& p, int c, int& i0, int& i1) {
    i0 = c * p.chunk_w;
    i1 = i0 + p.chunk_w;
    if (i1 > p.vocab) i1 = p.vocab;
    if (i0 > i1) i0 = i1;
}

__device__ __forceinline__ u64 shfl_up_u64(u64 v, int o) {
    unsigned lo = __shfl_up((unsigned)v, o, 64), hi = __shfl_up((unsigned)(v >> 32), o, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
    unsigned lo = __shfl((unsigned)v, src, 64), hi = __shfl((unsigned)(v >> 32), src, 64);
    return ((u64)hi << 32) | lo;
}
// 64-bit values on the DPP network: both halves moved with the same control (bound_ctrl: lanes without a source read 0)
template <int CTRL>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, true);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 readlane_u64(u64 v, int lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}
// inclusive scan over the 64 lanes of a wave (integers: the result does not depend on the association order).  Row shifts by 1, 2, 4, 8
// scan each row of 16 lanes on the DPP network, the three row totals are read as scalars - about 30 VALU instructions instead of the six
// dependent ds_bpermute round trips per half of the shuffle form (measured on the one-launch sampler: profiles/r04/c3_samp_phases.txt).
__device__ __forceinline__ u64 wave_scan_incl(u64 v) {
    v += dpp_u64<0x111>(v);       // row_shr:1
    v += dpp_u64<0x112>(v);       // row_shr:2
    v += dpp_u64<0x114>(v);       // row_shr:4
    v += dpp_u64<0x118>(v);       // row_shr:8
    const u64 t0 = readlane_u64(v, 15), t1 = readlane_u64(v, 31), t2 = readlane_u64(v, 47);
    const int row = (threadIdx.x & 63) >> 4;
    return v + (row >= 1 ? t0 : 0) + (row >= 2 ? t1 : 0) + (row >= 3 ? t2 : 0);
}
// sum over the 64 lanes of a wave, the same in every lane
__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
    v += dpp_u64<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_u64<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_u64<0x141>(v);       // row_half_mirror
    v += dpp_u64<0x140>(v);       // row_mirror
    return (readlane_u64(v, 0) + readlane_u64(v, 16)) + (readlane_u64(v, 32) + readlane_u64(v, 48));
}
// maximum of an unsigned 32-bit key over the wave, the same in every lane
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    auto step = [](unsigned x, unsigned y) { return x > y ? x : y; };
    v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
    v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
    v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));
    v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16),
                   c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return step(step(a, b), step(c, d));
}
// inclusive scan of 256 u64 values held one per thread: wave scans in registers + the 4 wave totals through LDS
// (two barriers instead of the seventeen of a Hillis-Steele scan through LDS)
__device__ __forceinline__ u64 block_scan_incl_256(u64 v, u64* sh) {
    const int tid = threadIdx.x, w = tid >> 6;
    u64 incl = wave_scan_incl(v);
    if ((tid & 63) == 63) sh[w] = incl;
    __syncthreads();
    u64 base = 0;
#pragma unroll
    for (int i = 0; i < SAMP_NT / 64; ++i) base += (i < w) ? sh[i] : 0;
    __syncthreads();
    return incl + base;
}

// ---- 0: repetition penalty (RepetitionContext.process), once per unique id; scratch reset
__global__ void __launch_bounds__(64) k_samp_prepare(SamplerParams p) {
    const int b = blockIdx.x, tid = threadIdx.x;
    SamplerScratch* sc = p.scratch + b;
    for (int i = tid; i < 256; i += 64) { sc->hist1[i] = 0; sc->hist2[i] = 0; }
    if (row_skipped(p, b)) return;
    if (p.logits32) {                                   // Soprano flavour: the processed logits are float32
        const bf16_t* src = p.logits + (size_t)b * p.Vpad;
        float* dst = p.logits32 + (size_t)b * p.Vpad;
        for (int i = tid; i < p.vocab; i += 64) dst[i] = bf16_to_f32(src[i]);
        __syncthreads();
    }
    if (p.penalty > 0.0f && p.penalty != 1.0f && p.window) {
        bf16_t* logits = p.logits + (size_t)b * p.Vpad;
        int wl = p.window_len[b];
        const int32_t* win = p.window + (size_t)b * p.ctx + (p.ctx - wl);
        for (int t = tid; t < wl; t += 64) {
            int id = win[t];
            bool first = (id >= 0 && id < p.vocab);
            for (int j = 0; j < t; ++j) first = first && (win[j] != id);
            if (first) {
                float l = bf16_to_f32(logits[id]);
                float v;
                if (p.penalty_flavor == 0) {
                    float pen = bf16_round_f32(p.penalty);        // scalar weakly typed to bf16
                    v = (l < 0.0f) ? l * pen : __fdiv_rn(l, pen);
                } else {
                    // Soprano applyRepetitionPenalty (Soprano.swift:888-901): float32, once PER OCCURRENCE in the window
                    int mult = 0;
                    for (int j = 0; j < wl; ++j) mult += (win[j] == id);
                    v = l;
                    for (int k = 0; k < mult; ++k) v = (v > 0.0f) ? __fdiv_rn(v, p.penalty) : v * p.penalty;
                }
                if (p.logits32) p.logits32[(size_t)b * p.Vpad + id] = v;
                else logits[id] = f32_to_bf16(v);
            }
        }
    }
}

// ---- 1: per-chunk max (and first index of the max, for greedy)
__global__ void __launch_bounds__(SAMP_NT) k_samp_max(SamplerParams p) {
    __shared__ float redf[SAMP_NT / 64];
    __shared__ int redi[SAMP_NT / 64];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (row_skipped(p, b)) return;
    int lo, hi, i0, i1;
    allowed_range(p, row_step(p, b), lo, hi);
    chunk_range(p, c, i0, i1);
    if (i0 < lo) i0 = lo;
    if (i1 > hi) i1 = hi;
    const bf16_t* logits = p.logits + (size_t)b * p.Vpad;
    const float* l32 = p.logits32 ? p.logits32 + (size_t)b * p.Vpad : nullptr;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = i0 + tid; i < i1; i += SAMP_NT) {
        float l = l32 ? l32[i] : bf16_to_f32(logits[i]);
        if (l > best) { best = l; bi = i; }               // ascending i per thread: first index kept on ties
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
        for (int w = 1; w < SAMP_NT / 64; ++w)
            if (redf[w] > best || (redf[w] == best && redi[w] < bi)) { best = redf[w]; bi = redi[w]; }
        p.scratch[b].pmax[c] = best;
        p.scratch[b].pidx[c] = bi;
    }
}

// max over the per-chunk maxima: one load per lane + a wave reduction (n_chunks <= SAMP_MAX_CHUNKS = 64), every wave for itself
__device__ __forceinline__ float row_max(const SamplerParams& p, int b) {
    const int lane = threadIdx.x & 63;
    float m = p.scratch[b].pmax[lane < p.n_chunks ? lane : 0];
    m = lane < p.n_chunks ? m : -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    return m;
}

// ---- 2: e, E, level-1 histogram.  Every vocabulary entry is visited; masked ones get e = 0.
__global__ void __launch_bounds__(SAMP_NT) k_samp_exp(SamplerParams p) {
    __shared__ u64 hist[256];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (row_skipped(p, b)) return;
    int lo, hi, i0, i1;
    allowed_range(p, row_step(p, b), lo, hi);
    chunk_range(p, c, i0, i1);
    hist[tid] = 0;
    __syncthreads();
    const bf16_t* logits = p.logits + (size_t)b * p.Vpad;
    const float* l32 = p.logits32 ? p.logits32 + (size_t)b * p.Vpad : nullptr;
    float* ebuf = p.e_buf + (size_t)b * p.Vpad;
    const float xmax = __fdiv_rn(row_max(p, b), p.temperature);      // fdiv is monotone: max x = fdiv(max l, T)
    for (int i = i0 + tid; i < i1; i += SAMP_NT) {
        float x = __fdiv_rn(l32 ? l32[i] : bf16_to_f32(logits[i]), p.temperature);
        float y = fminf(x - xmax, 0.0f);
        float e = det_exp_dev(y);
        if (i < lo || i >= hi) e = 0.0f;
        ebuf[i] = e;
        u64 E = (u64)(e * E_SCALE);
        if (E) atomicAdd(&hist[__float_as_uint(e) >> 24], E);
    }
    __syncthreads();
    if (hist[tid]) atomicAdd(&p.scratch[b].hist1[tid], hist[tid]);
}

// ---- 3: locate the level-1 bin of the nucleus boundary; level-2 histogram inside it
__global__ void __launch_bounds__(SAMP_NT) k_samp_hist2(SamplerParams p) {
    __shared__ u64 sh[SAMP_NT];
    __shared__ u64 hist[256];
    __shared__ unsigned s_bin;
    __shared__ u64 s_below;
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (row_skipped(p, b)) return;
    SamplerScratch* sc = p.scratch + b;
    u64 mine = sc->hist1[tid];
    u64 incl = block_scan_incl_256(mine, sh);
    sh[tid] = incl;
    __syncthreads();
    const u64 Z = sh[SAMP_NT - 1];
    const u64 thr = (u64)((double)(1.0f - p.top_p) * (double)Z);
    if (tid == 0) { s_bin = 255; s_below = 0; }
    __syncthreads();
    if (incl > thr && incl - mine <= thr) { s_bin = (unsigned)tid; s_below = incl - mine; }   // unique crossing
    hist[tid] = 0;
    __syncthreads();
    const unsigned bin1 = s_bin;
    if (c == 0 && tid == 0) { sc->bin1 = bin1; sc->below1 = s_below; sc->Z = Z; sc->thr = thr; }
    int i0, i1;
    chunk_range(p, c, i0, i1);
    const float* ebuf = p.e_buf + (size_t)b * p.Vpad;
    for (int i = i0 + tid; i < i1; i += SAMP_NT) {
        float e = ebuf[i];
        unsigned key = __float_as_uint(e) >> 16;
        if ((key >> 8) == bin1) {
            u64 E = (u64)(e * E_SCALE);
            if (E) atomicAdd(&hist[key & 255], E);
        }
    }
    __syncthreads();
    if (hist[tid]) atomicAdd(&sc->hist2[tid], hist[tid]);
}

// ---- 4: k* from the level-2 bins; kept mass of every chunk
__global__ void __launch_bounds__(SAMP_NT) k_samp_mass(SamplerParams p, int nucleus) {
    __shared__ u64 sh[SAMP_NT];
    __shared__ u64 red[SAMP_NT / 64];
    __shared__ unsigned s_bin;
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (row_skipped(p, b)) return;
    SamplerScratch* sc = p.scratch + b;
    unsigned kstar = 0;
    if (nucleus) {
        u64 mine = sc->hist2[tid];
        u64 incl = block_scan_incl_256(mine, sh) + sc->below1;
        if (tid == 0) s_bin = 255;
        __syncthreads();
        if (incl > sc->thr && incl - mine <= sc->thr) s_bin = (unsigned)tid;
        __syncthreads();
        kstar = (sc->bin1 << 8) | s_bin;
        if (c == 0 && tid == 0) sc->kstar = kstar;
    } else if (c == 0 && tid == 0) sc->kstar = 0;
    int i0, i1;
    chunk_range(p, c, i0, i1);
    const float* ebuf = p.e_buf + (size_t)b * p.Vpad;
    u64 m = 0;
    for (int i = i0 + tid; i < i1; i += SAMP_NT) {
        float e = ebuf[i];
        if ((__float_as_uint(e) >> 16) >= kstar) m += (u64)(e * E_SCALE);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned lo32 = __shfl_xor((unsigned)m, o, 64), hi32 = __shfl_xor((unsigned)(m >> 32), o, 64);
        m += ((u64)hi32 << 32) | lo32;
    }
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        u64 t = 0;
        for (int w = 0; w < SAMP_NT / 64; ++w) t += red[w];
        sc->cmass[c] = t;
    }
}

// ---- 5: draw + bookkeeping (generate loop, LlamaTTS.swift:721-738).  One block per row.
__global__ void __launch_bounds__(SAMP_NT) k_samp_pick(SamplerParams p, int greedy) {
    __shared__ u64 sh[SAMP_NT];
    __shared__ int s_token;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (row_skipped(p, b)) return;
    SamplerScratch* sc = p.scratch + b;
    const int step = row_step(p, b);
    int lo, hi;
    allowed_range(p, step, lo, hi);
    if (tid == 0) s_token = lo < p.vocab ? lo : 0;
    __syncthreads();
    if (greedy) {
        if (tid == 0) {
            float best = -INFINITY;
            int bi = 0x7fffffff;
            for (int c = 0; c < p.n_chunks; ++c)              // chunks ascend in index: first max wins ties
                if (sc->pmax[c] > best) { best = sc->pmax[c]; bi = sc->pidx[c]; }
            if (bi != 0x7fffffff) s_token = bi;
        }
    } else {
        // chunk selection (serial over <= 64 chunks), then in-chunk inverse CDF in index order
        // chunk selection: lane c holds the kept mass of chunk c; inclusive wave scan; first chunk whose prefix exceeds r
        const int lane = tid & 63;
        u64 cm = sc->cmass[lane < p.n_chunks ? lane : 0];
        cm = lane < p.n_chunks ? cm : 0;
        const u64 cincl = wave_scan_incl(cm);
        const u64 Zk = shfl_u64(cincl, 63);
        const u64 row = (u64)(p.row_offset + b);
        u64 a = p.seed ^ (0xD1B54A32D192ED03ull * (row + 1));
        const u64 rnd = mis_splitmix64(mis_splitmix64(a) + (u64)step);
        const u64 r = __umul64hi(rnd, Zk);
        const unsigned long long hit = __ballot(lane < p.n_chunks && r < cincl);
        const int cs = hit ? __ffsll((long long)hit) - 1 : -1;
        const u64 base = cs >= 0 ? shfl_u64(cincl - cm, cs) : 0;
        if (cs >= 0) {
            const unsigned kstar = sc->kstar;
            int i0, i1;
            chunk_range(p, cs, i0, i1);
            const int per = (i1 - i0 + SAMP_NT - 1) / SAMP_NT;      // contiguous sub-range per thread
            const int j0 = i0 + tid * per, j1 = min(i1, j0 + per);
            const float* ebuf = p.e_buf + (size_t)b * p.Vpad;
            u64 mine = 0;
            for (int i = j0; i < j1; ++i) {
                float e = ebuf[i];
                if ((__float_as_uint(e) >> 16) >= kstar) mine += (u64)(e * E_SCALE);
            }
            u64 incl = block_scan_incl_256(mine, sh) + base;
            u64 excl = incl - mine;
            if (mine > 0 && r >= excl && r < incl) {
                u64 run = excl;
                for (int i = j0; i < j1; ++i) {
                    float e = ebuf[i];
                    if ((__float_as_uint(e) >> 16) >= kstar) {
                        run += (u64)(e * E_SCALE);
                        if (run > r) { s_token = i; break; }
                    }
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int token = s_token;
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
                // budget exhausted (LlamaTTS.swift:714): the row stays active for the forward pass of this last
                // token (Soprano needs its hidden state, Soprano.swift:871-876); later sampler calls skip the row
                if (p.done_count) atomicAdd(p.done_count, 1);
            }
        }
    }
}

// ---- the whole sampler in ONE launch when the allowed id range holds <= 4096 entries (frame-constrained Orpheus / VyvoTTS steps,
// any [lo, hi) that narrow): one 1024-thread block per row, four CONSECUTIVE ids per thread, every quantity of the spec above in
// registers / LDS.  Same integers as the six-kernel path (E, Z, thr, k*, Z_K, r are exact), so the token is bit-identical; what
// goes away is five dependent kernel boundaries and four passes over a [rows][Vpad] float scratch per step.
#define SN_NT 1024
#define SN_W 4096
__device__ __forceinline__ u64 block_scan_incl_1024(u64 v, u64* sh /*[16]*/, u64* total) {
    const int tid = threadIdx.x, w = tid >> 6;
    u64 incl = wave_scan_incl(v);
    if ((tid & 63) == 63) sh[w] = incl;
    __syncthreads();
    u64 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SN_NT / 64; ++i) { const u64 t = sh[i]; base += (i < w) ? t : 0; tot += t; }
    __syncthreads();
    if (total) *total = tot;
    return incl + base;
}
__global__ void __launch_bounds__(SN_NT) k_samp_narrow(SamplerParams p) {
    __shared__ float sl[SN_W];
    __shared__ u64 hist[256];
    __shared__ u64 sh[SN_NT / 64];
    __shared__ float redf[SN_NT / 64];
    __shared__ int redi[SN_NT / 64];
    __shared__ int swin[SN_NT];
    __shared__ unsigned s_bin, s_bin2;
    __shared__ u64 s_below;
    __shared__ int s_token;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (row_skipped(p, b)) return;
    const int step = row_step(p, b);
    int lo, hi;
    allowed_range(p, step, lo, hi);
    bf16_t* logits = p.logits + (size_t)b * p.Vpad;
    // ---- the allowed window of the row -> LDS (ids >= hi: -inf)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lo + tid + j * SN_NT;
        sl[tid + j * SN_NT] = (i < hi) ? bf16_to_f32(logits[i]) : -INFINITY;
    }
    if (tid < 256) hist[tid] = 0;
    if (tid == 0) { s_token = lo < p.vocab ? lo : 0; s_bin = 255; s_bin2 = 255; s_below = 0; }
    __syncthreads();
    // ---- repetition penalty (RepetitionContext.process): once per unique id of the window, bf16 arithmetic; in place in
    // memory like the multi-kernel path, and in the staged window for the ids this step can sample
    if (p.penalty > 0.0f && p.penalty != 1.0f && p.window) {
        const int wl = p.window_len[b];
        const int32_t* win = p.window + (size_t)b * p.ctx + (p.ctx - wl);
        const float pen = bf16_round_f32(p.penalty);
        // the window goes through LDS first: the first-occurrence test below is O(wl) reads per id, and as dependent GLOBAL loads it
        // was half of this kernel's time (17 us -> see profiles/r02_final_bench_kernel_stats.csv)
        const bool staged = wl <= SN_NT;
        if (staged && tid < wl) swin[tid] = win[tid];
        __syncthreads();
        for (int t = tid; t < wl; t += SN_NT) {
            const int id = staged ? swin[t] : win[t];
            bool first = (id >= 0 && id < p.vocab);
            if (staged) { for (int j = 0; j < t; ++j) first = first && (swin[j] != id); }
            else { for (int j = 0; j < t; ++j) first = first && (win[j] != id); }
            if (first) {
                const float l = bf16_to_f32(logits[id]);
                const bf16_t v = f32_to_bf16((l < 0.0f) ? l * pen : __fdiv_rn(l, pen));
                logits[id] = v;
                if (id >= lo && id < hi) sl[id - lo] = bf16_to_f32(v);
            }
        }
        __syncthreads();
    }
    // ---- this thread's four consecutive ids
    float l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) l[j] = sl[4 * tid + j];
    const int i_base = lo + 4 * tid;
    // ---- max (first index on ties)
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (i_base + j < hi && l[j] > best) { best = l[j]; bi = i_base + j; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if ((tid & 63) == 0) { redf[tid >> 6] = best; redi[tid >> 6] = bi; }
    __syncthreads();
    best = redf[0]; bi = redi[0];
#pragma unroll
    for (int w = 1; w < SN_NT / 64; ++w)
        if (redf[w] > best || (redf[w] == best && redi[w] < bi)) { best = redf[w]; bi = redi[w]; }
    if (p.temperature == 0.0f) {
        if (tid == 0 && bi != 0x7fffffff) s_token = bi;
    } else {
        // ---- e, E, keys
        const float xmax = __fdiv_rn(best, p.temperature);
        float e[4];
        u64 E[4];
        unsigned key[4];
        u64 zsum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = __fdiv_rn(l[j], p.temperature);
            const float y = fminf(x - xmax, 0.0f);
            e[j] = (i_base + j < hi) ? det_exp_dev(y) : 0.0f;
            E[j] = (u64)(e[j] * E_SCALE);
            key[j] = __float_as_uint(e[j]) >> 16;
            zsum += E[j];
        }
        unsigned kstar = 0;
        if (p.top_p > 0.0f && p.top_p < 1.0f) {
            // level 1: mass per key >> 8
#pragma unroll
            for (int j = 0; j < 4; ++j) if (E[j]) atomicAdd(&hist[key[j] >> 8], E[j]);
            __syncthreads();
            u64 Z = 0;
            u64 mine = tid < 256 ? hist[tid] : 0;
            u64 incl = block_scan_incl_1024(mine, sh, &Z);
            const u64 thr = (u64)((double)(1.0f - p.top_p) * (double)Z);
            if (tid < 256 && incl > thr && incl - mine <= thr) { s_bin = (unsigned)tid; s_below = incl - mine; }   // unique crossing
            __syncthreads();
            const unsigned bin1 = s_bin;
            const u64 below1 = s_below;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            // level 2: mass per key & 255 inside bin1
#pragma unroll
            for (int j = 0; j < 4; ++j) if (E[j] && (key[j] >> 8) == bin1) atomicAdd(&hist[key[j] & 255], E[j]);
            __syncthreads();
            mine = tid < 256 ? hist[tid] : 0;
            incl = block_scan_incl_1024(mine, sh, nullptr) + below1;
            if (tid < 256 && incl > thr && incl - mine <= thr) s_bin2 = (unsigned)tid;
            __syncthreads();
            kstar = (bin1 << 8) | s_bin2;
        }
        // ---- kept mass in index order, draw, inverse CDF
        u64 Ek[4];
        u64 mine = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { Ek[j] = (key[j] >= kstar) ? E[j] : 0; mine += Ek[j]; }
        u64 Zk = 0;
        const u64 incl = block_scan_incl_1024(mine, sh, &Zk);
        const u64 excl = incl - mine;
        const u64 row = (u64)(p.row_offset + b);
        const u64 a = p.seed ^ (0xD1B54A32D192ED03ull * (row + 1));
        const u64 rnd = mis_splitmix64(mis_splitmix64(a) + (u64)step);
        const u64 r = __umul64hi(rnd, Zk);
        if (mine > 0 && r >= excl && r < incl) {
            u64 run = excl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                run += Ek[j];
                if (Ek[j] && run > r) { s_token = i_base + j; break; }
            }
        }
        (void)zsum;
    }
    __syncthreads();
    if (tid == 0) {      // bookkeeping of the generate loop (LlamaTTS.swift:721-738), as k_samp_pick
        const int token = s_token;
        if (p.tokens_out && step < p.tokens_stride) p.tokens_out[(size_t)b * p.tokens_stride + step] = token;
        if (p.n_gen && !p.step_override) p.n_gen[b] = step + 1;
        if (p.window && p.ctx > 0) {
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
                if (p.done_count) atomicAdd(p.done_count, 1);
            }
        }
    }
}

// ---- the whole sampler in ONE launch for the full vocabulary: SC_NB = 8 blocks of 1024 threads per row, SC_PER = 20 CONSECUTIVE ids
// per thread (8 x 1024 x 20 = 163 840 >= 156 940), every e_i in registers - no [rows][Vpad] float scratch, no kernel boundaries.
// The 8 blocks of a row meet at THREE row-local barriers:
//   1  the block maxima                          -> row maximum
//   2  the blocks' level-1 mass histograms       -> Z, threshold, the level-1 bin of the nucleus boundary
//   3  the blocks' level-2 histograms inside that bin + each block's mass above the bin
//                                                -> k*, and from the SAME slots the kept mass of every block (no fourth exchange)
// Every block publishes into its OWN slot of the row's exchange area (256 eight-byte agent-scope stores per histogram - one coalesced
// 2 KiB write-through instead of 256 read-modify-write atomics on a shared histogram) and nothing has to be zeroed afterwards: a slot
// is overwritten by its owner before the barrier that lets anyone read it.  Everything exchanged is an 8-byte agent-scope atomic
// store / load on both sides (the per-XCD L2s are not coherent with each other; that pairing is one of the valid hand-off forms of
// MI355X_MICROARCH.md), so no fences are needed: a block drains its stores (s_waitcnt vmcnt(0)), syncs, and one thread arrives on
// the row's counter.  The counter only grows (3 x 8 arrivals per launch); a block derives the launch's base from the value it reads
// at entry.  Spins are bounded: a row whose partner blocks never arrive reports through c_fail instead of hanging - the host reads
// the flags back after the decode loop (sampler_check_failed) and the launcher only picks this kernel when 8 x batch blocks fit the
// device at once (sampler_cluster_fits).  Same integers as the six-kernel path (E, Z, thr, k*, Z_K, r are exact), so the token is
// bit-identical to oracle/sampler.py.
// (Round 3's version accumulated into shared histograms with agent atomics and met four times: 42.8 us per launch in the step chain.)
#define SC_NB 8
#define SC_NT 1024
#define SC_PER 20
#define SC_ARRIVALS 3
#define SC_STAMP(i) do { if (p.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.dbg[i] = __builtin_readcyclecounter(); } while (0)
// the same for every block of row 0 on the constant 100 MHz counter (comparable across XCDs): dbg[16 + 4 c + k], k = entry / logits in registers /
// at barrier 1 / past barrier 1 - who waits for whom
#define SC_WALL(k) do { if (p.dbg && blockIdx.y == 0 && threadIdx.x == 0) p.dbg[16 + 4 * blockIdx.x + (k)] = wall_clock64(); } while (0)
static_assert(SC_NB == SAMP_CLUSTER_NB, "exchange slots per row (lm_kernels.h)");
__device__ __forceinline__ u64 sc_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sc_store(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned sc_key32(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float sc_unkey32(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
// row barrier: all of this block's exchange stores are performed, then one arrival; returns false on timeout.
// spin_limit: polls before giving up (a poll is ~0.1 us; tests force a tiny limit to exercise the failure path)
__device__ __forceinline__ bool sc_row_barrier(unsigned int* sync, unsigned target, int spin_limit, unsigned long long* dbg = nullptr) {
    __shared__ int ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *dbg = __builtin_readcyclecounter();      // (diagnostics: stores drained, block synced)
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 0;
        for (int it = 0; it < spin_limit; ++it) {
            if ((int)(__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { good = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}
__global__ void __launch_bounds__(SC_NT) k_samp_cluster(SamplerParams p, int spin_limit) {
    __shared__ u64 hist[256];
    __shared__ u64 sh[SC_NT / 64];
    __shared__ u64 redk[SC_NT / 64];
    __shared__ u64 kept[SC_NB];
    __shared__ __attribute__((aligned(16))) uint32_t stage[SC_NT * SC_PER / 2];     // the block's 20 480 logits (40 KB), then the [8][256] bin table
    __shared__ int pen_id[64];
    __shared__ float pen_val[64];
    __shared__ int swin[64];
    __shared__ int n_pen;
    __shared__ unsigned s_bin;
    __shared__ u64 s_below;
    __shared__ int s_token;
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    SC_STAMP(0);
    SC_WALL(0);
    if (row_skipped(p, b)) return;                           // (row-uniform: all 8 blocks of the row take the same exit)
    SamplerScratch* sc = p.scratch + b;
    const int step = row_step(p, b);
    int lo, hi;
    allowed_range(p, step, lo, hi);
    bf16_t* logits = p.logits + (size_t)b * p.Vpad;
    const int my0 = (c * SC_NT + tid) * SC_PER;              // this thread's ids my0 .. my0 + SC_PER - 1
    // ---- every load of the prologue is requested before anything waits, the small ones FIRST (loads retire in order: queued behind
    // the three 16-byte chunk loads, the window entries held the first barrier up by 2.5 us - profiles/r04/c5_samp_phases.txt): the
    // repetition window in full (its valid part is right-aligned; how long that is comes back with it), the row's barrier counter,
    // then this block's 20 480 logits as coalesced 16-byte loads (three per thread at most; Vpad is a multiple of 16, so the row and
    // the block's range start 32-byte aligned; chunks past the row re-read its last one - those ids are >= hi and masked below)
    const bool penalise = p.penalty > 0.0f && p.penalty != 1.0f && p.window;
    // (unconditional loads on clamped addresses: a guarded load is a branch with its own vmcnt(0) at the join - one dependent round trip
    // per guard; the counter is read by every lane of every wave - one address, one transaction per wave)
    const int32_t* wl_src = penalise ? p.window_len + b : reinterpret_cast<const int32_t*>(&sc->c_sync);
    const int32_t* win_src = penalise ? p.window + (size_t)b * p.ctx + (tid < p.ctx ? tid : 0) : reinterpret_cast<const int32_t*>(&sc->c_sync);
    const int wl_raw = *wl_src;
    const int win_raw = *win_src;
    const unsigned sync_now = __hip_atomic_load(&sc->c_sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // a row that timed out in an EARLIER launch (the flag is sticky until the host has read it) does not spin again: its counter is
    // short of arrivals, every barrier of this launch would run its whole poll budget - the replays queued behind a failed step end
    // in microseconds instead, and the host restarts the request on the multi-launch path at its next poll (run_generate)
    if (__hip_atomic_load(&sc->c_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) spin_limit = 0;
    const int wlen_all = penalise ? wl_raw : 0;
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));       // (a native vector: HIP's uint4 struct kept this array in scratch)
    u32x4_t q[3];
    {
        const u32x4_t* lp = reinterpret_cast<const u32x4_t*>(logits);
        const int n16 = p.Vpad >> 3;                                     // 16-byte chunks in the row
        const int c0 = c * (SC_NT * SC_PER / 8);                         // first chunk of this block
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ch = c0 + tid + j * SC_NT;
            q[j] = lp[ch < n16 ? ch : n16 - 1];
        }
    }
    const int wl = min(wlen_all, 64);
    if (penalise && tid < p.ctx && tid >= p.ctx - wlen_all && tid - (p.ctx - wlen_all) < 64) swin[tid - (p.ctx - wlen_all)] = win_raw;
    unsigned base = 0;
    if (tid == 0) {
        const unsigned v = sync_now;
        base = v - (v % (unsigned)(SC_ARRIVALS * SC_NB));    // < 8 arrivals of this launch can have happened before this block's first
        redk[0] = base;                                      // barrier, so rounding down to a multiple of 24 gives the launch's base
        n_pen = 0;
        s_token = lo < p.vocab ? lo : 0;
        s_bin = 255; s_below = 0;
    }
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    base = (unsigned)redk[0];
    // ---- repetition penalty (RepetitionContext.process): once per unique id of the window, bf16 arithmetic; the block that holds the
    // id writes it back (in place, like the other sampler paths) and lists it so that the owning thread patches its register copy
    if (penalise) {
        const float pen = bf16_round_f32(p.penalty);
        if (tid < wl) {
            const int id = swin[tid];
            bool first = id >= 0 && id < p.vocab && id / (SC_NT * SC_PER) == c;
            for (int j = 0; j < tid; ++j) first = first && (swin[j] != id);
            if (first) {
                const float l = bf16_to_f32(logits[id]);
                const bf16_t v = f32_to_bf16((l < 0.0f) ? l * pen : __fdiv_rn(l, pen));
                logits[id] = v;
                const int slot = atomicAdd(&n_pen, 1);
                pen_id[slot] = id; pen_val[slot] = bf16_to_f32(v);
            }
        }
    }
    // ---- the logits through LDS: 40 consecutive bytes per thread.  (As 4-byte loads at a 40-byte lane stride: 4.3 us for this phase.)
    float l[SC_PER];
    {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (tid + j * SC_NT < SC_NT * SC_PER / 8) reinterpret_cast<u32x4_t*>(stage)[tid + j * SC_NT] = q[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SC_PER / 4; ++j) {
            const uint2 w = *reinterpret_cast<const uint2*>(stage + tid * (SC_PER / 2) + 2 * j);
            l[4 * j] = bf16_to_f32((bf16_t)(w.x & 0xffffu)); l[4 * j + 1] = bf16_to_f32((bf16_t)(w.x >> 16));
            l[4 * j + 2] = bf16_to_f32((bf16_t)(w.y & 0xffffu)); l[4 * j + 3] = bf16_to_f32((bf16_t)(w.y >> 16));
        }
    }
    __syncthreads();
    {
        const int np = n_pen;
        for (int k = 0; k < np; ++k) {
            const int d = pen_id[k] - my0;
            const float v = pen_val[k];
#pragma unroll
            for (int j = 0; j < SC_PER; ++j) l[j] = (d == j) ? v : l[j];
        }
    }
    SC_STAMP(1);                                             // logits loaded, penalties patched
    SC_WALL(1);
    // ---- max (first index on ties) as one comparable 64-bit key: (order-preserving float key, ~index)
    u64 bestk = 0;
#pragma unroll
    for (int j = 0; j < SC_PER; ++j) {
        const int i = my0 + j;
        const u64 k = ((u64)sc_key32(l[j]) << 32) | (unsigned)(~(unsigned)i);
        bestk = (i >= lo && i < hi && k > bestk) ? k : bestk;
    }
    if (p.temperature == 0.0f) {                              // greedy needs the first index of the maximum: the whole 64-bit key
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo32 = __shfl_xor((unsigned)bestk, o, 64), hi32 = __shfl_xor((unsigned)(bestk >> 32), o, 64);
            const u64 ok = ((u64)hi32 << 32) | lo32;
            bestk = ok > bestk ? ok : bestk;
        }
    } else {                                                  // sampling needs its value only: 32 bits on the DPP network
        bestk = (u64)wave_max_u32((unsigned)(bestk >> 32)) << 32;
    }
    if ((tid & 63) == 0) redk[tid >> 6] = bestk;
    __syncthreads();
    if (tid == 0) {
        u64 m = 0;
#pragma unroll
        for (int w = 0; w < SC_NT / 64; ++w) m = redk[w] > m ? redk[w] : m;
        sc_store(&sc->x_max[c], m);
    }
    SC_STAMP(2);                                             // block maximum published
    SC_WALL(2);
    bool alive = sc_row_barrier(&sc->c_sync, base + 1u * SC_NB, spin_limit, p.dbg ? p.dbg + 12 : nullptr);
    SC_STAMP(3);                                             // barrier 1 passed
    SC_WALL(3);
    u64 rowk = 0;
#pragma unroll
    for (int k = 0; k < SC_NB; ++k) { const u64 v = sc_load(&sc->x_max[k]); rowk = v > rowk ? v : rowk; }
    const float best = rowk ? sc_unkey32((unsigned)(rowk >> 32)) : -INFINITY;
    const int best_i = rowk ? (int)~(unsigned)rowk : 0x7fffffff;
    u64 Zk = 0;
    bool mine_blk = false;
    if (p.temperature == 0.0f || !alive) {
        // greedy (or a failed first barrier): the launch's remaining arrivals without waiting, block 0 does the bookkeeping
        if (tid == 0) __hip_atomic_fetch_add(&sc->c_sync, (unsigned)(SC_ARRIVALS - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0 && alive && best_i != 0x7fffffff) s_token = best_i;
        mine_blk = c == 0;
        Zk = 1;
    } else {
        // ---- e, E, keys
        const float xmax = __fdiv_rn(best, p.temperature);
        float e[SC_PER];
#pragma unroll
        for (int j = 0; j < SC_PER; ++j) {
            const int i = my0 + j;
            const float x = __fdiv_rn(l[j], p.temperature);
            const float y = fminf(x - xmax, 0.0f);
            e[j] = (i >= lo && i < hi) ? det_exp_dev(y) : 0.0f;
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // (twenty interleaved polynomial chains spill at 128 registers)
        }
        SC_STAMP(4);                                         // e computed
        // level 1: this block's mass per key >> 8 -> its slot
#pragma unroll
        for (int j = 0; j < SC_PER; ++j) {
            const u64 E = (u64)(e[j] * E_SCALE);
            if (E) atomicAdd(&hist[__float_as_uint(e[j]) >> 24], E);
        }
        __syncthreads();
        SC_STAMP(5);                                         // level-1 histogram in LDS
        if (tid < 256) sc_store(&sc->x_hist1[c][tid], hist[tid]);
        alive = sc_row_barrier(&sc->c_sync, base + 2u * SC_NB, spin_limit, p.dbg ? p.dbg + 13 : nullptr) && alive;
        SC_STAMP(6);                                         // barrier 2 passed
        u64 mine = 0;
#pragma unroll
        for (int k = 0; k < SC_NB; ++k) mine += tid < 256 ? sc_load(&sc->x_hist1[k][tid]) : 0;
        unsigned kstar = 0;
        const bool nucleus = p.top_p > 0.0f && p.top_p < 1.0f;
        if (nucleus) {
            u64 Z = 0;
            u64 incl = block_scan_incl_1024(mine, sh, &Z);
            const u64 thr = (u64)((double)(1.0f - p.top_p) * (double)Z);
            if (tid < 256 && incl > thr && incl - mine <= thr) { s_bin = (unsigned)tid; s_below = incl - mine; }      // unique crossing
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned bin1 = s_bin;
            const u64 below1 = s_below;
            SC_STAMP(7);                                     // slots read, level-1 scan done
            // level 2: mass per key & 255 inside bin1, and everything above the bin (kept whatever k* turns out to be)
            u64 above = 0;
#pragma unroll
            for (int j = 0; j < SC_PER; ++j) {
                const unsigned key = __float_as_uint(e[j]) >> 16;
                const u64 E = (u64)(e[j] * E_SCALE);
                if (E && (key >> 8) == bin1) atomicAdd(&hist[key & 255], E);
                above += (key >> 8) > bin1 ? E : 0;
            }
            above = wave_sum_u64(above);
            if ((tid & 63) == 0) redk[tid >> 6] = above;
            __syncthreads();
            if (tid < 256) sc_store(&sc->x_hist2[c][tid], hist[tid]);
            if (tid == 0) {
                u64 t = 0;
                for (int w = 0; w < SC_NT / 64; ++w) t += redk[w];
                sc_store(&sc->x_above[c], t);
                s_bin = 255;
            }
            SC_STAMP(8);                                     // level-2 histogram published
            alive = sc_row_barrier(&sc->c_sync, base + 3u * SC_NB, spin_limit) && alive;
            SC_STAMP(9);                                     // barrier 3 passed
            mine = 0;
#pragma unroll
            for (int k = 0; k < SC_NB; ++k) mine += tid < 256 ? sc_load(&sc->x_hist2[k][tid]) : 0;
            incl = block_scan_incl_1024(mine, sh, nullptr) + below1;
            if (tid < 256 && incl > thr && incl - mine <= thr) s_bin = (unsigned)tid;
            __syncthreads();
            const unsigned bin2 = s_bin;
            kstar = (bin1 << 8) | bin2;
            // kept mass of every block: its mass above bin1 + its level-2 bins from k* up.  Threads 0..255 hold one bin of each block:
            // the masked bins go through LDS as an [8][256] table, wave w then sums block w's row (one DPP reduction per wave instead
            // of eight shuffle reductions in each of four waves: 5.5 -> see profiles/r04/ for this phase)
            u64* table = reinterpret_cast<u64*>(stage);
            if (tid < 256) {
#pragma unroll
                for (int k = 0; k < SC_NB; ++k) {
                    const u64 hv = sc_load(&sc->x_hist2[k][tid]);            // (read again: 16 registers less across the scan)
                    table[k * 256 + tid] = (unsigned)tid >= bin2 ? hv : 0;
                }
            }
            __syncthreads();
            if (tid < SC_NB * 64) {
                const int w = tid >> 6, ln = tid & 63;
                const u64 v = wave_sum_u64((table[w * 256 + ln] + table[w * 256 + 64 + ln]) + (table[w * 256 + 128 + ln] + table[w * 256 + 192 + ln]));
                if (ln == 0) kept[w] = v + sc_load(&sc->x_above[w]);
            }
        } else {
            if (tid == 0) __hip_atomic_fetch_add(&sc->c_sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (three arrivals per block and launch)
            u64* table = reinterpret_cast<u64*>(stage);          // no nucleus cut: a block keeps all of its mass = the sum of its level-1 bins
            if (tid < 256) {
#pragma unroll
                for (int k = 0; k < SC_NB; ++k) table[k * 256 + tid] = sc_load(&sc->x_hist1[k][tid]);
            }
            __syncthreads();
            if (tid < SC_NB * 64) {
                const int w = tid >> 6, ln = tid & 63;
                const u64 v = wave_sum_u64((table[w * 256 + ln] + table[w * 256 + 64 + ln]) + (table[w * 256 + 128 + ln] + table[w * 256 + 192 + ln]));
                if (ln == 0) kept[w] = v;
            }
        }
        __syncthreads();
        SC_STAMP(10);                                        // k* and every block's kept mass known
        u64 before = 0;
#pragma unroll
        for (int k = 0; k < SC_NB; ++k) { const u64 v = kept[k]; Zk += v; before += (k < c) ? v : 0; }
        // ---- kept mass in index order (consecutive ids per thread, consecutive threads, consecutive blocks), draw, inverse CDF
        mine = 0;
#pragma unroll
        for (int j = 0; j < SC_PER; ++j) mine += ((__float_as_uint(e[j]) >> 16) >= kstar) ? (u64)(e[j] * E_SCALE) : 0;
        u64 Zblk = 0;
        const u64 incl = block_scan_incl_1024(mine, sh, &Zblk);
        const u64 row = (u64)(p.row_offset + b);
        const u64 a = p.seed ^ (0xD1B54A32D192ED03ull * (row + 1));
        const u64 rnd = mis_splitmix64(mis_splitmix64(a) + (u64)step);
        const u64 r = __umul64hi(rnd, Zk);
        const u64 excl = before + incl - mine;
        bool hit = false;
        if (mine > 0 && r >= excl && r < excl + mine) {
            u64 run = excl;
#pragma unroll
            for (int j = 0; j < SC_PER; ++j) {
                const u64 Ek = ((__float_as_uint(e[j]) >> 16) >= kstar) ? (u64)(e[j] * E_SCALE) : 0;
                run += Ek;
                if (!hit && Ek && run > r) { s_token = my0 + j; hit = true; }
            }
        }
        // exactly one block of the row holds r (Z_K > 0 whenever the allowed range is not empty: the maximum has e = 1); that block
        // does the bookkeeping below.  (Zblk == kept[c]: the same integers summed two ways.)
        mine_blk = r >= before && r < before + Zblk;
        SC_STAMP(11);                                        // draw located
    }
    if (!alive) {                                             // a partner block never arrived: reported, block 0 emits the fallback token
        if (tid == 0) __hip_atomic_store(&sc->c_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c != 0) return;
    } else if (Zk ? !mine_blk : c != 0) return;               // (empty range: block 0 reports the fallback token)
    __syncthreads();
    if (tid == 0) {      // bookkeeping of the generate loop (LlamaTTS.swift:721-738), as k_samp_pick
        const int token = s_token;
        if (p.tokens_out && step < p.tokens_stride) p.tokens_out[(size_t)b * p.tokens_stride + step] = token;
        if (p.n_gen && !p.step_override) p.n_gen[b] = step + 1;
        if (p.window && p.ctx > 0) {
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
                if (p.done_count) atomicAdd(p.done_count, 1);
            }
        }
    }
}
void sampler_scratch_init(SamplerScratch* scratch, int batch, hipStream_t s) {
    if (scratch && batch > 0) HIP_CHECK(hipMemsetAsync(scratch, 0, (size_t)batch * sizeof(SamplerScratch), s));
}
// how often a launch of the one-launch sampler reported a timed-out row barrier in this process (diagnostics, tests)
static std::atomic<int> g_sampler_failures{0};
extern "C" int32_t mis_debug_sampler_failures(void) { return g_sampler_failures.load(); }
void sampler_fail_flags_async(SamplerScratch* scratch, int batch, unsigned* host_flags, hipStream_t s) {
    HIP_CHECK(hipMemcpy2DAsync(host_flags, sizeof(unsigned), &scratch[0].c_fail, sizeof(SamplerScratch), sizeof(unsigned), (size_t)batch,
                               hipMemcpyDeviceToHost, s));
}
bool sampler_fail_flags_any(const unsigned* host_flags, int batch) {
    bool any = false;
    for (int b = 0; b < batch; ++b) any = any || host_flags[b] != 0;
    return any;
}
void sampler_note_failure(SamplerScratch* scratch, int batch, hipStream_t s) {
    g_sampler_failures.fetch_add(1);
    sampler_scratch_init(scratch, batch, s);           // counters and flags back to the state a fresh scratch has
    HIP_CHECK(hipStreamSynchronize(s));
}
bool sampler_check_failed(SamplerScratch* scratch, int batch, hipStream_t s) {
    if (!scratch || batch <= 0) return false;
    std::vector<unsigned> f(batch, 0u);
    sampler_fail_flags_async(scratch, batch, f.data(), s);
    HIP_CHECK(hipStreamSynchronize(s));
    const bool any = sampler_fail_flags_any(f.data(), batch);
    if (any) sampler_note_failure(scratch, batch, s);
    return any;
}
// the one-launch sampler's 8 x batch blocks spin on each other: all of them have to be resident at once.  The capacity (occupancy query
// x compute units) is cached per DEVICE: the replicas of a group may sit on devices of different sizes or partition modes.
static bool sampler_cluster_fits(int batch) {
    static std::mutex mu;
    static std::map<int, int> capacity;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(mu);
    auto it = capacity.find(dev);
    if (it == capacity.end()) {
        int per_cu = 0, cap = 0;
        hipDeviceProp_t prop{};
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_samp_cluster, SC_NT, 0) == hipSuccess)
            cap = std::min(per_cu, 2) * prop.multiProcessorCount;      // (never count on more than two 1024-thread blocks per CU: the
                                                                       //  third would depend on the register count of a compiler release)
        it = capacity.emplace(dev, cap).first;
    }
    return SC_NB * batch <= it->second;
}

void sampler_resolve(SamplerParams& p, bool multi_launch_only) {
    const char* ew = getenv("MIS_SAMPLER_WIDE");                     // non-zero: never the one-launch kernels (A/B, parity tests, shared devices)
    const char* es = getenv("MIS_SAMPLER_SPIN");                     // polls per row barrier before a block gives up (tests: the failure path)
    p.path_resolved = 1;
    p.force_multi = (multi_launch_only || (ew && atoi(ew) != 0)) ? 1 : 0;
    p.spin = es ? (atoi(es) > 0 ? atoi(es) : 0) : (1 << 22);         // 0: every barrier times out at once
}

void sampler_plan(int vocab, int* n_chunks, int* chunk_w) {
    int nc = (vocab + 4095) / 4096;
    if (nc > SAMP_MAX_CHUNKS) nc = SAMP_MAX_CHUNKS;
    if (nc < 1) nc = 1;
    int cw = (vocab + nc - 1) / nc;
    cw = (cw + 7) / 8 * 8;
    *n_chunks = (vocab + cw - 1) / cw;
    *chunk_w = cw;
}

void launch_sampler(const SamplerParams& p_in, int batch, hipStream_t s, bool multi_launch_only) {
    MIS_REQUIRE(p_in.scratch && p_in.n_chunks >= 1 && p_in.n_chunks <= SAMP_MAX_CHUNKS && p_in.chunk_w > 0, MIS_ERR_GENERATION_FAILED,
                "sampler scratch not configured");
    // which kernels: resolved ONCE per generate call (sampler_resolve; the result is part of the captured graph's key) - or here, per
    // launch, for the stand-alone entry point whose callers switch the environment between calls (parity tests)
    SamplerParams p = p_in;
    if (!p.path_resolved) sampler_resolve(p, multi_launch_only);
    const bool multi = multi_launch_only || p.force_multi != 0;
    {   // narrow allowed range (every frame-constrained step; any static [lo, hi) of <= 4096 ids): the single-launch sampler
        const int hi = (p.hi <= 0 || p.hi > p.vocab) ? p.vocab : p.hi, lo = p.lo < 0 ? 0 : p.lo;
        // frame_constrained 2: the frame range, but through the full-vocabulary kernels (every id visited, the masked ones get e = 0) -
        // what bench.py times as the honest stand-in for a real checkpoint's unconstrained decode
        const bool narrow = p.frame_constrained == 1 || (!p.frame_constrained && hi - lo <= SN_W);
        if (narrow && !multi && !p.logits32 && p.penalty_flavor == 0) {
            hipLaunchKernelGGL(k_samp_narrow, dim3(batch), dim3(SN_NT), 0, s, p);
            return;
        }
    }
    // full vocabulary in one launch (k_samp_cluster): 16-byte row chunks (Vpad % 16 == 0 keeps every row and block range aligned), every
    // block of the launch resident at once
    if (!multi && !p.logits32 && p.penalty_flavor == 0 && p.vocab <= SC_NB * SC_NT * SC_PER && p.Vpad % 16 == 0 && p.ctx <= 64 &&
        sampler_cluster_fits(batch)) {
        hipLaunchKernelGGL(k_samp_cluster, dim3(SC_NB, batch), dim3(SC_NT), 0, s, p, p.spin);
        return;
    }
    dim3 g2(p.n_chunks, batch);
    hipLaunchKernelGGL(k_samp_prepare, dim3(batch), dim3(64), 0, s, p);
    hipLaunchKernelGGL(k_samp_max, g2, dim3(SAMP_NT), 0, s, p);
    if (p.temperature == 0.0f) {
        hipLaunchKernelGGL(k_samp_pick, dim3(batch), dim3(SAMP_NT), 0, s, p, 1);
        return;
    }
    hipLaunchKernelGGL(k_samp_exp, g2, dim3(SAMP_NT), 0, s, p);
    int nucleus = (p.top_p > 0.0f && p.top_p < 1.0f) ? 1 : 0;
    if (nucleus) hipLaunchKernelGGL(k_samp_hist2, g2, dim3(SAMP_NT), 0, s, p);
    hipLaunchKernelGGL(k_samp_mass, g2, dim3(SAMP_NT), 0, s, p, nucleus);
    hipLaunchKernelGGL(k_samp_pick, dim3(batch), dim3(SAMP_NT), 0, s, p, 0);
}
