// q3_codec.hip - Qwen3-TTS speech-tokenizer decoder (12.5 Hz codes -> 24 kHz waveform), float32, causal end to end.
//
// Reference being replaced: Qwen3TTSSpeechTokenizerDecoder (Sources/MLXAudioTTS/Models/Qwen3TTS/Qwen3TTSSpeechTokenizer.swift:
// 888-1006) = SplitResidualVectorQuantizer.decode (:91-118, EuclideanCodebook Sources/MLXAudioCodecs/Mimi/Quantization.swift:7-60)
// -> CausalConv1d k3 (:132-196) -> DecoderTransformer (:449-503: RMSNorm, RoPE, causal SDPA, SwiGLU, LayerScale) -> 2 x
// [CausalTransposeConv1d (:732-749) + ConvNeXtBlock (:257-299)] -> conv k7 -> 4 x DecoderBlock (:583-637: SnakeBeta,
// transposed conv k=2s with right trim, 3 residual units with dilations 1/3/9) -> SnakeBeta -> conv k7 -> clip.
// Two ways through the same kernels:
//  * whole sequences (callAsFunction :926-946): dense [B][C][T] buffers;
//  * `streamingStep` (:971-1006): only the NEW frames of a chunk are computed.  Carried state per session: the last
//    (k-1)*dilation input columns of every causal conv (CausalConv1d.step :199-227 and the k7 convs :655-667,:710-722), the last
//    input column of every 2s-tap transposed conv (the reference carries the equivalent `overflow` of the OUTPUT, :553-576) and
//    the transformer's K/V cache.  Work buffers then have HP columns of head room in front of every row ([B][C][HP + T']): a
//    tiny kernel drops the carried columns there before the conv runs, so the conv kernels need no second code path - only a
//    row stride and a lowest valid column (GemmParams.ldx / x_lo).  Every output column is computed by the same instruction
//    sequence in both modes (same K order, same 64-key attention tiles counted from key 0), so chunked decode is BITWISE equal
//    to the whole-sequence decode - except where the reference itself differs: its overlap-add sums two biased transposed-conv
//    outputs, so the first `stride` samples after every chunk boundary carry the bias twice (`dup_bias`, on by default).
// Activations are NCT ([B][C][T], time contiguous); dense convs are exact-f32 MFMA contractions over (tap, channel)
// (k_snac_gemm modes TAPS / CONVT of snac.hip) with the SnakeBeta activation fused into the operand load.
#include "common.h"
#include "kernels.h"
#include "codec_kernels.h"
#include "q3_kernels.h"

#include <math.h>
#include <string.h>
#include <algorithm>
#include <map>

struct mis_q3dec {
    int device = 0;
    mis_qwen3tts_config cfg{};
    std::map<std::string, std::vector<float>> raw;
    std::map<std::string, std::vector<int64_t>> raw_shape;
    bool finalized = false;
    DevBuf<float> arena;
    size_t rvq_tables = 0, zeros = 0;
    struct Lin { size_t w = 0, b = 0; int M = 0, K = 0; };
    struct Layer { size_t ln1, ln2, ls1, ls2; Lin qkv, o, gu, down; };
    Lin pre_conv, in_proj, out_proj, dec0;
    size_t tnorm = 0;
    std::vector<Layer> layers;
    struct Up { Lin ct; size_t dw, dwb, lnw, lnb, gamma; Lin p1, p2; int f; };
    std::vector<Up> ups;
    struct RU { size_t a1, ra1, a2, ra2; Lin c1, c2; int dil; };
    struct Blk { size_t a, ra; Lin ct; int s, cin, cout; RU ru[3]; };
    std::vector<Blk> blocks;
    size_t fin_a = 0, fin_ra = 0, fin_w = 0;
    float fin_b = 0.0f;
    int fin_c = 0;
    DevBuf<float> buf[4];
    CodecPack pack;                      // split-bf16 weight fragments + activation scratch (codec_bf3.hip)
    DevBuf<int32_t> codes_dev;
    // streaming session (resetStreamingState / streamingStep)
    struct Stream {
        bool open = false, dup_bias = true;
        int batch = 0, cap_frames = 0, pos = 0, steps = 0;
        size_t hist_n = 0;
        DevBuf<float> hist;               // carried conv inputs of every layer, in call order: [layer][B][C][H]
        DevBuf<float> kv;                 // [layers][B][2 Hkv D][cap_frames]: K rows (unrotated), then V rows
    } st;
};
#define Q3_HP 64                           // head-room columns in front of every work-buffer row (>= 6*9 = longest history)

// ---------------------------------------------------------------------------- kernels
// codes [B][nq][T] -> h [B][C][T]: sum over quantizers of the folded tables [nq][bins][C]
// code (b, q, t) at codes[b*cs_b + q*cs_q + t*cs_t]: [B][nq][T] arrays and the generate loop's [B][frames][nq] store alike
__global__ void k_q3_rvq(const int32_t* __restrict__ codes, int64_t cs_b, int64_t cs_q, int64_t cs_t, const float* __restrict__ tables,
                         float* __restrict__ h, int nq, int bins, int C, int ld) {
    const int t = blockIdx.x, b = blockIdx.y;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.0f;
        for (int q = 0; q < nq; ++q) {
            int code = codes[(size_t)b * cs_b + (size_t)q * cs_q + (size_t)t * cs_t];
            code = min(max(code, 0), bins - 1);
            acc += tables[((size_t)q * bins + code) * C + c];
        }
        h[((size_t)b * C + c) * ld + t] = acc;
    }
}

// streaming: history columns in and out.  st [B][C][H] holds the last H columns this layer's input had before this chunk; x points
// at column 0 of the chunk's [B][C][ld] input (Tn new columns).  Columns [-H, 0) of x <- st, then st <- the last H columns of
// [st | new]  (CausalConv1d.step :201-210).  One block per row, H <= 64.
__global__ void __launch_bounds__(64) k_q3_hist(float* __restrict__ st, float* __restrict__ x, int C, int ld, int H, int Tn) {
    const int c = blockIdx.x, b = blockIdx.y, i = threadIdx.x;
    float* sr = st + ((size_t)b * C + c) * H;
    float* xr = x + ((size_t)b * C + c) * ld;
    float old = 0.0f, nw = 0.0f;
    if (i < H) {
        old = sr[i];
        const int src = Tn - H + i;
        nw = src >= 0 ? xr[src] : sr[i + Tn];
    }
    __syncthreads();
    if (i < H) { xr[i - H] = old; sr[i] = nw; }
}
// streaming: K and V rows of the fused q|k|v output (columns [0, Tn)) -> cache columns [pos0, pos0 + Tn)
__global__ void k_q3_kv_append(const float* __restrict__ qkv, int64_t q_bs, int q_ld, int row0, float* __restrict__ kv, int64_t kv_bs,
                               int kv_ld, int rows, int pos0, int Tn) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
    if (t >= Tn) return;
    kv[(size_t)b * kv_bs + (size_t)r * kv_ld + pos0 + t] = qkv[(size_t)b * q_bs + (size_t)(row0 + r) * q_ld + t];
}

// normalisation over the CHANNEL axis of NCT data; rms = 1: w * x * rsqrt(mean(x^2) + eps), else LayerNorm(w, bias).
// Block = 32 columns x 8 channel groups (channel c belongs to group c % 8): every group accumulates its channels in order, the eight
// partial sums of a column are added in group order - a fixed reduction tree per column, independent of where the column lies, which
// is what the bitwise chunk-invariance of the streaming decode needs.  (The first version ran one thread per column over all C
// channels: 32 blocks for a 100-frame batch, 186 us per call at 2 % CU occupancy - 4.7 ms of a 101 ms decode.)
#define Q3N_COLS 32
__global__ void __launch_bounds__(256) k_q3_norm_ct(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ w, const float* __restrict__ bias,
                                                    int C, int Tn, int T /*row stride*/, float eps, int rms) {
    __shared__ float red[8][Q3N_COLS + 1];
    const int tl = threadIdx.x & (Q3N_COLS - 1), cg = threadIdx.x >> 5;
    const int t = blockIdx.x * Q3N_COLS + tl, b = blockIdx.y;
    const bool ok = t < Tn;
    const float* xb = x + (size_t)b * C * T + (ok ? t : 0);
    float* yb = y + (size_t)b * C * T + (ok ? t : 0);
    auto column_sum = [&](float v) {
        red[cg][tl] = v;
        __syncthreads();
        float s = 0.0f;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += red[g][tl];
        __syncthreads();
        return s;
    };
    if (rms) {
        float q = 0.0f;
        for (int c = cg; c < C; c += 8) { const float v = xb[(size_t)c * T]; q += v * v; }
        const float r = rsqrtf(column_sum(q) / (float)C + eps);
        if (ok)
            for (int c = cg; c < C; c += 8) yb[(size_t)c * T] = w[c] * (xb[(size_t)c * T] * r);
    } else {
        float sm = 0.0f;
        for (int c = cg; c < C; c += 8) sm += xb[(size_t)c * T];
        const float mean = column_sum(sm) / (float)C;
        float q = 0.0f;
        for (int c = cg; c < C; c += 8) { const float d = xb[(size_t)c * T] - mean; q += d * d; }
        const float r = 1.0f / sqrtf(column_sum(q) / (float)C + eps);
        if (ok)
            for (int c = cg; c < C; c += 8) yb[(size_t)c * T] = (xb[(size_t)c * T] - mean) * r * w[c] + bias[c];
    }
}

// causal depthwise conv, k taps (CausalConv1d with groups = C, :132-196): y[c][t] = b[c] + sum_j w[c][j] x[c][t - (k-1-j)]
__global__ void k_q3_dw_causal(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ w, const float* __restrict__ bias,
                               int C, int T, int ld, int x_lo, int k) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const float* xr = x + ((size_t)b * C + c) * ld;
    float acc = bias[c];
    for (int j = 0; j < k; ++j) {
        int ts = t - (k - 1 - j);
        if (ts >= x_lo) acc += w[c * k + j] * xr[ts];
    }
    y[((size_t)b * C + c) * ld + t] = acc;
}

// gu [B][2I][T] (gate rows then up rows) -> act [B][I][T] = silu(gate) * up
__global__ void k_q3_swiglu(const float* __restrict__ gu, float* __restrict__ act, int I, int Tn, int T /*row stride*/) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (t >= Tn) return;
    float g = gu[((size_t)b * 2 * I + i) * T + t], u = gu[((size_t)b * 2 * I + I + i) * T + t];
    act[((size_t)b * I + i) * T + t] = (g / (1.0f + expf(-g))) * u;
}

// causal attention with rotate-half RoPE, f32.  Queries: Tq columns of q (row stride q_ld, batch stride q_bs) at absolute
// positions pos0 + t; keys / values: columns [0, pos0 + Tq) of k / v (row stride kv_ld, batch stride kv_bs) - the same q|k|v
// tensor for whole sequences (pos0 = 0), the session's cache when streaming.  One thread per query, 64 queries per block; K/V
// tiles of 64 keys counted from key 0 are staged (K rotated) in LDS and read by all threads at the same address (broadcast), so
// a query sees the same operation order whichever chunk it arrives in.
template <int D>
__global__ void __launch_bounds__(64) k_q3_attn(Q3AttnArgs a) {
    __shared__ float Ks[D][64];
    __shared__ float Vs[D][64];
    const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int tq = blockIdx.x * 64 + tid;
    const int t = a.pos0 + tq;                                        // absolute position of this thread's query
    const int kvh = h / (a.H / a.Hkv);
    const int Tk = a.pos0 + a.Tq;
    const float* qb = a.q + (size_t)b * a.q_bs + (size_t)h * D * a.q_ld;
    const float* kb = a.k + (size_t)b * a.kv_bs + (size_t)kvh * D * a.kv_ld;
    const float* vb = a.v + (size_t)b * a.kv_bs + (size_t)kvh * D * a.kv_ld;
    float q[D], o[D];
    const bool valid = tq < a.Tq;
#pragma unroll
    for (int i = 0; i < D / 2; ++i) {
        float inv = 1.0f / powf(a.theta, (float)(2 * i) / (float)D);
        float ang = (float)t * inv, c = cosf(ang), s = sinf(ang);
        float x1 = valid ? qb[(size_t)i * a.q_ld + tq] : 0.0f, x2 = valid ? qb[(size_t)(i + D / 2) * a.q_ld + tq] : 0.0f;
        q[i] = (x1 * c - x2 * s) * a.scale;
        q[i + D / 2] = (x2 * c + x1 * s) * a.scale;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.0f;
    float m = -INFINITY, l = 0.0f;
    const int t_hi = a.pos0 + min(blockIdx.x * 64 + 63, a.Tq - 1);
    for (int j0 = 0; j0 <= t_hi; j0 += 64) {
        __syncthreads();
        {   // thread tid stages key j0 + tid
            const int j = j0 + tid;
            const bool kin = j < Tk;
#pragma unroll
            for (int i = 0; i < D / 2; ++i) {
                float inv = 1.0f / powf(a.theta, (float)(2 * i) / (float)D);
                float ang = (float)j * inv, c = cosf(ang), s = sinf(ang);
                float x1 = kin ? kb[(size_t)i * a.kv_ld + j] : 0.0f, x2 = kin ? kb[(size_t)(i + D / 2) * a.kv_ld + j] : 0.0f;
                Ks[i][tid] = x1 * c - x2 * s;
                Ks[i + D / 2][tid] = x2 * c + x1 * s;
            }
#pragma unroll
            for (int d = 0; d < D; ++d) Vs[d][tid] = kin ? vb[(size_t)d * a.kv_ld + j] : 0.0f;
        }
        __syncthreads();
        const int jn = min(64, t_hi - j0 + 1);
        for (int jj = 0; jj < jn; ++jj) {
            if (j0 + jj > t) break;                                   // causal
            float sc = 0.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) sc += q[d] * Ks[d][jj];
            float mn = fmaxf(m, sc);
            float corr = expf(m - mn), p = expf(sc - mn);
            l = l * corr + p;
#pragma unroll
            for (int d = 0; d < D; ++d) o[d] = o[d] * corr + p * Vs[d][jj];
            m = mn;
        }
    }
    if (valid) {
        float* ob = a.out + (size_t)b * a.o_bs + (size_t)h * D * a.o_ld;
        float rl = 1.0f / l;
#pragma unroll
        for (int d = 0; d < D; ++d) ob[(size_t)d * a.o_ld + tq] = o[d] * rl;
    }
}

// SnakeBeta -> causal conv k (C -> 1) -> clip(-1, 1)   (DecoderOutputSnake / DecoderOutputConv :676-730, clip :945)
// 256 output columns per block; 16 channels at a time are staged through LDS with the activation applied ONCE per element (the
// per-thread version evaluated sin() k times per element and ran at 0.6 TB/s); accumulation order as before: channel-major, tap-minor
#define Q3F_TILE 256
#define Q3F_CH 16
__global__ void __launch_bounds__(256) k_q3_final(const float* __restrict__ x, float* __restrict__ out, int64_t out_stride,
                                                  const float* __restrict__ w /*[k][C]*/, float bias, const float* __restrict__ a,
                                                  const float* __restrict__ ra, int C, int T, int ld, int x_lo, int k) {
    __shared__ float sx[Q3F_CH][Q3F_TILE + 8];
    const int b = blockIdx.y, t0 = blockIdx.x * Q3F_TILE, tid = threadIdx.x;
    const int halo = k - 1;                                              // k <= 8
    float acc = bias;
    for (int c0 = 0; c0 < C; c0 += Q3F_CH) {
        __syncthreads();
        for (int i = tid; i < Q3F_CH * (Q3F_TILE + halo); i += 256) {
            const int cc = i / (Q3F_TILE + halo), j = i - cc * (Q3F_TILE + halo);
            const int c = c0 + cc, t = t0 - halo + j;
            float v = 0.0f;
            if (c < C && t >= x_lo && t < T) {
                v = x[((int64_t)b * C + c) * ld + t];
                v = fmaf(ra[c], mis_sin_sq(a[c] * v), v);
            }
            sx[cc][j] = v;
        }
        __syncthreads();
        const int cmax = min(Q3F_CH, C - c0);
        for (int cc = 0; cc < cmax; ++cc)
            for (int j = 0; j < k; ++j) acc += w[j * C + c0 + cc] * sx[cc][tid + j];
    }
    const int t = t0 + tid;
    if (t < T) out[(size_t)b * out_stride + t] = fminf(fmaxf(acc, -1.0f), 1.0f);
}

void launch_q3_attn(const Q3AttnArgs& a, int D, int batch, hipStream_t s) {
    dim3 ag(cdiv(a.Tq, 64), a.H, batch);
    if (D == 64) hipLaunchKernelGGL((k_q3_attn<64>), ag, dim3(64), 0, s, a);
    else if (D == 32) hipLaunchKernelGGL((k_q3_attn<32>), ag, dim3(64), 0, s, a);
    else if (D == 16) hipLaunchKernelGGL((k_q3_attn<16>), ag, dim3(64), 0, s, a);
    else throw MisError(MIS_ERR_INVALID_INPUT, "attention head_dim must be 16, 32 or 64");
}
void launch_q3_norm_ct(const float* x, float* y, const float* w, const float* bias, int batch, int C, int Tn, int ld, float eps, int rms,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_q3_norm_ct, dim3(cdiv(Tn, Q3N_COLS), batch), dim3(256), 0, s, x, y, w, bias, C, Tn, ld, eps, rms);
}

// ---------------------------------------------------------------------------- host: weights
mis_status mis_q3dec_create(const mis_qwen3tts_config* cfg, int device, mis_q3dec** out) {
    MIS_API_BEGIN
    MIS_REQUIRE(cfg->dec_head_dim == 64 || cfg->dec_head_dim == 32 || cfg->dec_head_dim == 16, MIS_ERR_INVALID_INPUT, "decoder head_dim must be 16, 32 or 64");
    MIS_REQUIRE(cfg->n_upsample_rates >= 1 && cfg->n_upsample_rates <= 8 && cfg->n_upsampling_ratios >= 0 && cfg->n_upsampling_ratios <= 8,
                MIS_ERR_INVALID_INPUT, "bad upsample configuration");
    MIS_REQUIRE(cfg->dec_num_heads % cfg->dec_num_kv_heads == 0 && cfg->dec_codebook_dim % 2 == 0, MIS_ERR_INVALID_INPUT, "bad decoder dims");
    mis_q3dec* d = new mis_q3dec();
    d->device = device; d->cfg = *cfg;
    *out = d;
    MIS_API_END
}
void mis_q3dec_destroy(mis_q3dec* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    delete d;
}
int q3dec_total_upsample(const mis_q3dec* d) {
    int u = 1;
    for (int i = 0; i < d->cfg.n_upsample_rates; ++i) u *= d->cfg.upsample_rates[i];
    for (int i = 0; i < d->cfg.n_upsampling_ratios; ++i) u *= d->cfg.upsampling_ratios[i];
    return u;
}

mis_status mis_q3dec_set_tensor(mis_q3dec* d, const char* name, const void* data, mis_dtype dtype, const int64_t* shape, int ndim) {
    MIS_API_BEGIN
    MIS_REQUIRE(!d->finalized && ndim >= 1 && ndim <= 3, MIS_ERR_INVALID_INPUT, "bad tensor %s", name);
    size_t n = 1;
    std::vector<int64_t> sh;
    for (int i = 0; i < ndim; ++i) { MIS_REQUIRE(shape[i] > 0, MIS_ERR_INVALID_INPUT, "bad shape"); n *= (size_t)shape[i]; sh.push_back(shape[i]); }
    HIP_CHECK(hipSetDevice(d->device));
    size_t esz = dtype == MIS_F32 ? 4 : 2;
    std::vector<uint8_t> host(n * esz);
    HIP_CHECK(hipMemcpy(host.data(), data, n * esz, hipMemcpyDefault));
    std::vector<float> v(n);
    if (dtype == MIS_F32) memcpy(v.data(), host.data(), n * 4);
    else if (dtype == MIS_BF16) for (size_t i = 0; i < n; ++i) v[i] = bf16_to_f32(((bf16_t*)host.data())[i]);
    else if (dtype == MIS_F16) for (size_t i = 0; i < n; ++i) v[i] = f16_to_f32_host(((uint16_t*)host.data())[i]);
    else throw MisError(MIS_ERR_INVALID_INPUT, "unsupported dtype");
    d->raw[name] = std::move(v);
    d->raw_shape[name] = sh;
    MIS_API_END
}

static const std::vector<float>& need(mis_q3dec* d, const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = d->raw.find(name);
    MIS_REQUIRE(it != d->raw.end(), MIS_ERR_NOT_INITIALIZED, "speech tokenizer weight missing: %s", name.c_str());
    MIS_REQUIRE(d->raw_shape[name] == std::vector<int64_t>(shape), MIS_ERR_INVALID_INPUT, "speech tokenizer weight %s has the wrong shape", name.c_str());
    return it->second;
}

mis_status mis_q3dec_finalize(mis_q3dec* d) {
    MIS_API_BEGIN
    MIS_REQUIRE(!d->finalized, MIS_ERR_INVALID_INPUT, "already finalized");
    HIP_CHECK(hipSetDevice(d->device));
    const mis_qwen3tts_config& cf = d->cfg;
    const int64_t Cd = cf.dec_codebook_dim, half = Cd / 2, bins = cf.dec_codebook_size, nq = cf.dec_num_quantizers,
                  nsem = cf.dec_num_semantic_quantizers, ld = cf.dec_latent_dim, hs = cf.dec_hidden_size, I = cf.dec_intermediate_size,
                  H = cf.dec_num_heads, Hkv = cf.dec_num_kv_heads, D = cf.dec_head_dim, dd = cf.dec_decoder_dim;
    std::vector<float> arena;
    auto push = [&](const std::vector<float>& v) { size_t o = arena.size(); arena.insert(arena.end(), v.begin(), v.end()); while (arena.size() & 3) arena.push_back(0.f); return o; };
    auto lin_t = [&](const std::vector<float>& w, int64_t out_f, int64_t in_f) {                  // [out][in] -> A^T [in][out]
        std::vector<float> at((size_t)in_f * out_f);
        for (int64_t o = 0; o < out_f; ++o) for (int64_t i = 0; i < in_f; ++i) at[i * out_f + o] = w[o * in_f + i];
        return at;
    };
    auto conv_t = [&](const std::vector<float>& w, int64_t co, int64_t k, int64_t ci) {           // [co][k][ci] -> A^T [(j*ci + c)][co]
        std::vector<float> at((size_t)k * ci * co);
        for (int64_t o = 0; o < co; ++o) for (int64_t j = 0; j < k; ++j) for (int64_t c = 0; c < ci; ++c) at[(j * ci + c) * co + o] = w[(o * k + j) * ci + c];
        return at;
    };
    auto convT_t = [&](const std::vector<float>& w, int64_t co, int64_t k, int64_t ci, int64_t s) { // -> [s][(j*ci + c)][co], tap p + s*j
        const int64_t nt = k / s;
        std::vector<float> at((size_t)s * nt * ci * co);
        for (int64_t p = 0; p < s; ++p) for (int64_t j = 0; j < nt; ++j) for (int64_t c = 0; c < ci; ++c) for (int64_t o = 0; o < co; ++o)
            at[((p * nt + j) * ci + c) * co + o] = w[(o * k + (p + s * j)) * ci + c];
        return at;
    };
    auto snake_pair = [&](const std::string& pa, const std::string& pb, int64_t C, size_t& a, size_t& ra) {
        const auto& al = need(d, pa, {C}); const auto& be = need(d, pb, {C});
        std::vector<float> av(C), rv(C);
        for (int64_t i = 0; i < C; ++i) { av[i] = expf(al[i]); rv[i] = 1.0f / (expf(be[i]) + 1e-9f); }   // SnakeBeta :236-254
        a = push(av); ra = push(rv);
    };
    d->zeros = push(std::vector<float>((size_t)std::max<int64_t>({ld, dd, Cd, 4}), 0.0f));
    {   // folded RVQ tables [nq][bins][Cd] = output_proj . (embedding_sum / max(cluster_usage, 1e-5))
        std::vector<float> tables((size_t)nq * bins * Cd);
        for (int64_t q = 0; q < nq; ++q) {
            const bool first = q < nsem;
            const std::string grp = first ? "rvq_first" : "rvq_rest";
            const std::string p = "decoder.quantizer." + grp + ".vq.layers." + std::to_string(first ? q : q - nsem) + ".codebook";
            const auto& es = need(d, p + ".embedding_sum", {bins, half});
            const auto& cu = need(d, p + ".cluster_usage", {bins});
            const auto& pw = need(d, "decoder.quantizer." + grp + ".output_proj.weight", {Cd, 1, half});
            for (int64_t v = 0; v < bins; ++v) {
                const float inv = 1.0f / std::max(cu[v], 1e-5f);
                for (int64_t c = 0; c < Cd; ++c) {
                    float acc = 0.0f;
                    for (int64_t k = 0; k < half; ++k) acc += pw[c * half + k] * (es[v * half + k] * inv);
                    tables[((size_t)q * bins + v) * Cd + c] = acc;
                }
            }
        }
        d->rvq_tables = push(tables);
    }
    auto lin = [&](const std::string& p, int64_t out_f, int64_t in_f, bool bias) {
        mis_q3dec::Lin L; L.M = (int)out_f; L.K = (int)in_f;
        L.w = push(lin_t(need(d, p + ".weight", {out_f, in_f}), out_f, in_f));
        L.b = bias ? push(need(d, p + ".bias", {out_f})) : (size_t)-1;
        return L;
    };
    auto conv = [&](const std::string& p, int64_t co, int64_t k, int64_t ci) {
        mis_q3dec::Lin L; L.M = (int)co; L.K = (int)(k * ci);
        L.w = push(conv_t(need(d, p + ".weight", {co, k, ci}), co, k, ci));
        L.b = push(need(d, p + ".bias", {co}));
        return L;
    };
    d->pre_conv = conv("decoder.pre_conv.conv", ld, 3, Cd);
    const std::string P = "decoder.pre_transformer";
    d->in_proj = lin(P + ".input_proj", hs, ld, true);
    d->out_proj = lin(P + ".output_proj", ld, hs, true);
    d->tnorm = push(need(d, P + ".norm.weight", {hs}));
    d->layers.clear();
    for (int li = 0; li < cf.dec_num_layers; ++li) {
        const std::string p = P + ".layers." + std::to_string(li);
        mis_q3dec::Layer L{};
        L.ln1 = push(need(d, p + ".input_layernorm.weight", {hs}));
        L.ln2 = push(need(d, p + ".post_attention_layernorm.weight", {hs}));
        L.ls1 = push(need(d, p + ".self_attn_layer_scale.scale", {hs}));
        L.ls2 = push(need(d, p + ".mlp_layer_scale.scale", {hs}));
        {   // q, k, v rows concatenated: one GEMM
            std::vector<float> w;
            for (const char* nm : {"q_proj", "k_proj", "v_proj"}) {
                int64_t rows = (std::string(nm) == "q_proj" ? H : Hkv) * D;
                const auto& t = need(d, p + ".self_attn." + nm + ".weight", {rows, hs});
                w.insert(w.end(), t.begin(), t.end());
            }
            L.qkv.M = (int)((H + 2 * Hkv) * D); L.qkv.K = (int)hs; L.qkv.w = push(lin_t(w, L.qkv.M, hs)); L.qkv.b = (size_t)-1;
        }
        L.o = lin(p + ".self_attn.o_proj", hs, H * D, false);
        {
            std::vector<float> w = need(d, p + ".mlp.gate_proj.weight", {I, hs});
            const auto& u = need(d, p + ".mlp.up_proj.weight", {I, hs});
            w.insert(w.end(), u.begin(), u.end());
            L.gu.M = (int)(2 * I); L.gu.K = (int)hs; L.gu.w = push(lin_t(w, 2 * I, hs)); L.gu.b = (size_t)-1;
        }
        L.down = lin(p + ".mlp.down_proj", hs, I, false);
        d->layers.push_back(L);
    }
    d->ups.clear();
    for (int i = 0; i < cf.n_upsampling_ratios; ++i) {
        const int64_t f = cf.upsampling_ratios[i];
        const std::string p = "decoder.upsample." + std::to_string(i) + ".layers";
        mis_q3dec::Up U{};
        U.f = (int)f;
        U.ct.M = (int)ld; U.ct.K = (int)ld;
        U.ct.w = push(convT_t(need(d, p + ".0.conv.weight", {ld, f, ld}), ld, f, ld, f));
        U.ct.b = push(need(d, p + ".0.conv.bias", {ld}));
        U.dw = push(need(d, p + ".1.dwconv.conv.weight", {ld, 7, 1})); U.dwb = push(need(d, p + ".1.dwconv.conv.bias", {ld}));
        U.lnw = push(need(d, p + ".1.norm.weight", {ld})); U.lnb = push(need(d, p + ".1.norm.bias", {ld}));
        U.p1 = lin(p + ".1.pwconv1", 4 * ld, ld, true);
        U.p2 = lin(p + ".1.pwconv2", ld, 4 * ld, true);
        U.gamma = push(need(d, p + ".1.gamma", {ld}));
        d->ups.push_back(U);
    }
    d->dec0 = conv("decoder.decoder.0.conv", dd, 7, ld);
    d->blocks.clear();
    for (int bi = 0; bi < cf.n_upsample_rates; ++bi) {
        const int64_t cin = dd >> bi, cout = dd >> (bi + 1), s = cf.upsample_rates[bi];
        const std::string p = "decoder.decoder." + std::to_string(bi + 1) + ".block";
        mis_q3dec::Blk B{};
        B.s = (int)s; B.cin = (int)cin; B.cout = (int)cout;
        snake_pair(p + ".0.alpha", p + ".0.beta", cin, B.a, B.ra);
        B.ct.M = (int)cout; B.ct.K = (int)(2 * cin);
        B.ct.w = push(convT_t(need(d, p + ".1.conv.weight", {cout, 2 * s, cin}), cout, 2 * s, cin, s));
        B.ct.b = push(need(d, p + ".1.conv.bias", {cout}));
        const int dils[3] = {1, 3, 9};
        for (int ri = 0; ri < 3; ++ri) {
            const std::string q = p + "." + std::to_string(ri + 2);
            snake_pair(q + ".act1.alpha", q + ".act1.beta", cout, B.ru[ri].a1, B.ru[ri].ra1);
            B.ru[ri].c1 = conv(q + ".conv1.conv", cout, 7, cout);
            snake_pair(q + ".act2.alpha", q + ".act2.beta", cout, B.ru[ri].a2, B.ru[ri].ra2);
            B.ru[ri].c2 = conv(q + ".conv2.conv", cout, 1, cout);
            B.ru[ri].dil = dils[ri];
        }
        d->blocks.push_back(B);
    }
    {
        const int n = cf.n_upsample_rates;
        const int64_t cl = dd >> n;
        d->fin_c = (int)cl;
        snake_pair("decoder.decoder." + std::to_string(n + 1) + ".alpha", "decoder.decoder." + std::to_string(n + 1) + ".beta", cl, d->fin_a, d->fin_ra);
        const auto& w = need(d, "decoder.decoder." + std::to_string(n + 2) + ".conv.weight", {1, 7, cl});     // [1][k][C] == [k][C]
        d->fin_w = push(w);
        d->fin_b = need(d, "decoder.decoder." + std::to_string(n + 2) + ".conv.bias", {1})[0];
    }
    d->arena.alloc(arena.size());
    HIP_CHECK(hipMemcpy(d->arena.p, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    d->raw.clear(); d->raw_shape.clear();
    d->finalized = true;
    MIS_API_END
}

// ---------------------------------------------------------------------------- decode
// history floats one batch row carries through a streaming session (layers in call order of q3dec_run)
static size_t q3dec_hist_floats(const mis_q3dec* d) {
    const mis_qwen3tts_config& cf = d->cfg;
    size_t n = (size_t)cf.dec_codebook_dim * 2;                                 // pre_conv k3
    n += (size_t)d->ups.size() * cf.dec_latent_dim * 6;                         // ConvNeXt depthwise k7
    n += (size_t)cf.dec_latent_dim * 6;                                         // decoder.0 k7
    for (auto& B : d->blocks) n += (size_t)B.cin + (size_t)B.cout * (6 + 18 + 54);
    n += (size_t)d->fin_c * 6;                                                  // output conv k7
    return n;
}

// codes: element (b, q, t) at codes_dev[b*cs_b + q*cs_q + t*cs_t].  st == nullptr: whole sequence of T frames.  st != nullptr: the
// next T frames of an open streaming session (positions st->pos ..).
// stop_after (debug taps, whole-sequence mode only): 0 full; 1 quantizer; 2 transformer; 3 upsample; 4.. block (stop_after - 4).
static const float* q3dec_run(mis_q3dec* d, const int32_t* codes_dev, int64_t cs_b, int64_t cs_q, int64_t cs_t, int batch, int T,
                              float* wav_dev, int64_t wav_stride, int stop_after, int* outC, int64_t* outT, hipStream_t s,
                              mis_q3dec::Stream* st) {
    CodecPackScope pack_scope(&d->pack);
    MIS_REQUIRE(d->finalized, MIS_ERR_NOT_INITIALIZED, "speech tokenizer not finalized");
    MIS_REQUIRE(!st || stop_after == 0, MIS_ERR_INVALID_INPUT, "decoder taps are a whole-sequence facility");
    const mis_qwen3tts_config& cf = d->cfg;
    const float* W = d->arena.p;
    const int Cd = cf.dec_codebook_dim, ld = cf.dec_latent_dim, hs = cf.dec_hidden_size, I = cf.dec_intermediate_size, H = cf.dec_num_heads,
              Hkv = cf.dec_num_kv_heads, D = cf.dec_head_dim;
    const int up = q3dec_total_upsample(d);
    const int HP = st ? Q3_HP : 0;
    auto LD = [&](int64_t Tc) { return (int)(HP ? HP + round_up(Tc, 4) : Tc); };          // row stride of a work buffer
    // buffer sizing: the largest [C][T'] along the pipeline
    size_t need_elems = (size_t)std::max({Cd, ld, (H + 2 * Hkv) * D, 2 * I, hs}) * LD(T);
    {
        int64_t Tc = T;
        for (auto& U : d->ups) { Tc *= U.f; need_elems = std::max(need_elems, (size_t)4 * ld * LD(Tc)); }
        need_elems = std::max(need_elems, (size_t)cf.dec_decoder_dim * LD(Tc));
        for (auto& B : d->blocks) { need_elems = std::max(need_elems, (size_t)B.cin * LD(Tc)); Tc *= B.s; need_elems = std::max(need_elems, (size_t)B.cout * LD(Tc)); }
    }
    for (int i = 0; i < 4; ++i) d->buf[i].alloc((size_t)batch * need_elems + 2 * Q3_HP);
    float *a = d->buf[0].p + HP, *b = d->buf[1].p + HP, *t1 = d->buf[2].p + HP, *t2 = d->buf[3].p + HP;     // column 0 of row 0
    auto gemm = [&](const mis_q3dec::Lin& L, const float* X, float* Y, int N, int Tin, int Tout, const float* R = nullptr,
                    const float* scale = nullptr, const float* al = nullptr, const float* ral = nullptr) {
        GemmParams g{};
        g.AT = W + L.w; g.bias = L.b == (size_t)-1 ? nullptr : W + L.b; g.X = X; g.Y = Y; g.R = R; g.scale = scale; g.alpha = al; g.ralpha = ral;
        g.M = L.M; g.K = L.K; g.N = N; g.Tin = Tin; g.Tout = Tout;
        g.ldx = LD(Tin); g.ldy = LD(Tout); g.x_lo = -HP;
        return g;
    };
    // streaming: carried input columns of the conv about to run on x (C channels, H = (k-1)*dilation columns, Tn new columns)
    size_t hist_cur = 0;
    auto hist = [&](float* x, int C, int Tn, int Hc) {
        if (!st) return;
        MIS_REQUIRE(Hc <= Q3_HP && hist_cur + (size_t)batch * C * Hc <= st->hist_n, MIS_ERR_GENERATION_FAILED, "streaming history overflow");
        hipLaunchKernelGGL(k_q3_hist, dim3(C, batch), dim3(64), 0, s, st->hist.p + hist_cur, x, C, LD(Tn), Hc, Tn);
        hist_cur += (size_t)batch * C * Hc;
    };
    const float* Z = W + d->zeros;
    dim3 tb(128);
    int Tc = T;
    const int pos0 = st ? st->pos : 0;
    hipLaunchKernelGGL(k_q3_rvq, dim3(T, batch), dim3(256), 0, s, codes_dev, cs_b, cs_q, cs_t, W + d->rvq_tables, a, cf.dec_num_quantizers,
                       cf.dec_codebook_size, Cd, LD(T));
    if (stop_after == 1) { *outC = Cd; *outT = T; return a; }
    {   // pre_conv: causal k3
        hist(a, Cd, T, 2);
        GemmParams g = gemm(d->pre_conv, a, b, T, T, T, nullptr, nullptr, Z, Z);
        g.Cin = Cd; g.taps = 3; g.dil = 1; g.pad = 2;
        launch_gemm(GEMM_TAPS, true, g, batch, s);
    }
    // ---- transformer (NCT: channels x time), x in `a`
    launch_gemm(GEMM_PLAIN, false, gemm(d->in_proj, b, a, T, T, T), batch, s);
    float* x = a; float* y = b;
    const int ldT = LD(T);
    int li = 0;
    for (auto& L : d->layers) {
        hipLaunchKernelGGL(k_q3_norm_ct, dim3(cdiv(T, Q3N_COLS), batch), dim3(256), 0, s, x, t1, W + L.ln1, nullptr, hs, T, ldT, cf.dec_rms_norm_eps, 1);
        launch_gemm(GEMM_PLAIN, false, gemm(L.qkv, t1, t2, T, T, T), batch, s);
        Q3AttnArgs aa{};
        aa.q = t2; aa.q_bs = (int64_t)(H + 2 * Hkv) * D * ldT; aa.q_ld = ldT;
        aa.out = t1; aa.o_bs = (int64_t)H * D * ldT; aa.o_ld = ldT;
        aa.H = H; aa.Hkv = Hkv; aa.Tq = T; aa.pos0 = pos0; aa.theta = cf.dec_rope_theta; aa.scale = 1.0f / sqrtf((float)D);
        if (st) {
            const int64_t kv_bs = (int64_t)2 * Hkv * D * st->cap_frames;
            float* cache = st->kv.p + (size_t)li * batch * kv_bs;
            hipLaunchKernelGGL(k_q3_kv_append, dim3(cdiv(T, 64), 2 * Hkv * D, batch), dim3(64), 0, s, t2, aa.q_bs, ldT, H * D, cache, kv_bs,
                               st->cap_frames, 2 * Hkv * D, pos0, T);
            aa.k = cache; aa.v = cache + (size_t)Hkv * D * st->cap_frames; aa.kv_bs = kv_bs; aa.kv_ld = st->cap_frames;
        } else {
            aa.k = t2 + (size_t)H * D * ldT; aa.v = t2 + (size_t)(H + Hkv) * D * ldT; aa.kv_bs = aa.q_bs; aa.kv_ld = ldT;
        }
        dim3 ag(cdiv(T, 64), H, batch);
        if (D == 64) hipLaunchKernelGGL((k_q3_attn<64>), ag, dim3(64), 0, s, aa);
        else if (D == 32) hipLaunchKernelGGL((k_q3_attn<32>), ag, dim3(64), 0, s, aa);
        else hipLaunchKernelGGL((k_q3_attn<16>), ag, dim3(64), 0, s, aa);
        launch_gemm(GEMM_RESID, false, gemm(L.o, t1, y, T, T, T, x, W + L.ls1), batch, s);
        std::swap(x, y);
        hipLaunchKernelGGL(k_q3_norm_ct, dim3(cdiv(T, Q3N_COLS), batch), dim3(256), 0, s, x, t1, W + L.ln2, nullptr, hs, T, ldT, cf.dec_rms_norm_eps, 1);
        launch_gemm(GEMM_PLAIN, false, gemm(L.gu, t1, t2, T, T, T), batch, s);
        hipLaunchKernelGGL(k_q3_swiglu, dim3(cdiv(T, 128), I, batch), tb, 0, s, t2, t1, I, T, ldT);
        launch_gemm(GEMM_RESID, false, gemm(L.down, t1, y, T, T, T, x, W + L.ls2), batch, s);
        std::swap(x, y);
        ++li;
    }
    hipLaunchKernelGGL(k_q3_norm_ct, dim3(cdiv(T, Q3N_COLS), batch), dim3(256), 0, s, x, t1, W + d->tnorm, nullptr, hs, T, ldT, cf.dec_rms_norm_eps, 1);
    launch_gemm(GEMM_PLAIN, false, gemm(d->out_proj, t1, y, T, T, T), batch, s);
    std::swap(x, y);                                                   // x: [B][ld][T]
    if (stop_after == 2) { *outC = ld; *outT = T; return x; }
    // ---- upsample layers: transposed conv (k = stride = f: no overlap, nothing carried) + ConvNeXt
    for (auto& U : d->ups) {
        GemmParams g = gemm(U.ct, x, y, Tc, Tc, Tc * U.f, nullptr, nullptr, Z, Z);
        g.s = U.f; g.pad = 0; g.Cin = ld;
        launch_gemm(GEMM_CONVT, true, g, batch, s);
        Tc *= U.f;
        std::swap(x, y);
        const int ldc = LD(Tc);
        hist(x, ld, Tc, 6);
        hipLaunchKernelGGL(k_q3_dw_causal, dim3(cdiv(Tc, 128), ld, batch), tb, 0, s, x, t1, W + U.dw, W + U.dwb, ld, Tc, ldc, -HP, 7);
        hipLaunchKernelGGL(k_q3_norm_ct, dim3(cdiv(Tc, Q3N_COLS), batch), dim3(256), 0, s, t1, t2, W + U.lnw, W + U.lnb, ld, Tc, ldc, 1e-6f, 0);
        launch_gemm(GEMM_GELU, false, gemm(U.p1, t2, t1, Tc, Tc, Tc), batch, s);
        launch_gemm(GEMM_RESID, false, gemm(U.p2, t1, y, Tc, Tc, Tc, x, W + U.gamma), batch, s);
        std::swap(x, y);
    }
    if (stop_after == 3) { *outC = ld; *outT = Tc; return x; }
    {   // decoder.0: causal k7
        hist(x, ld, Tc, 6);
        GemmParams g = gemm(d->dec0, x, y, Tc, Tc, Tc, nullptr, nullptr, Z, Z);
        g.Cin = ld; g.taps = 7; g.dil = 1; g.pad = 6;
        launch_gemm(GEMM_TAPS, true, g, batch, s);
        std::swap(x, y);
    }
    int bi = 0;
    for (auto& B : d->blocks) {
        hist(x, B.cin, Tc, 1);                                          // tap p + s of output frame n reads input column n - 1
        GemmParams g = gemm(B.ct, x, y, Tc, Tc, Tc * B.s, nullptr, nullptr, W + B.a, W + B.ra);
        g.s = B.s; g.pad = 0; g.Cin = B.cin;
        g.dup_bias_n0 = (st && st->dup_bias && st->steps > 0) ? 1 : 0;
        launch_gemm(GEMM_CONVT, true, g, batch, s);
        Tc *= B.s;
        std::swap(x, y);
        for (int ri = 0; ri < 3; ++ri) {
            const auto& R = B.ru[ri];
            hist(x, B.cout, Tc, 6 * R.dil);
            GemmParams g1 = gemm(R.c1, x, t1, Tc, Tc, Tc, nullptr, nullptr, W + R.a1, W + R.ra1);
            g1.Cin = B.cout; g1.taps = 7; g1.dil = R.dil; g1.pad = 6 * R.dil;
            launch_gemm(GEMM_TAPS, true, g1, batch, s);
            launch_gemm(GEMM_RESID, true, gemm(R.c2, t1, y, Tc, Tc, Tc, x, nullptr, W + R.a2, W + R.ra2), batch, s);
            std::swap(x, y);
        }
        if (stop_after == 4 + bi) { *outC = B.cout; *outT = Tc; return x; }
        ++bi;
    }
    MIS_REQUIRE(Tc == (int64_t)T * up, MIS_ERR_GENERATION_FAILED, "internal length mismatch");
    hist(x, d->fin_c, Tc, 6);
    hipLaunchKernelGGL(k_q3_final, dim3(cdiv(Tc, Q3F_TILE), batch), dim3(256), 0, s, x, wav_dev, wav_stride, W + d->fin_w, d->fin_b, W + d->fin_a,
                       W + d->fin_ra, d->fin_c, Tc, LD(Tc), -HP, 7);
    HIP_CHECK(hipGetLastError());
    if (st) {
        MIS_REQUIRE(hist_cur == st->hist_n, MIS_ERR_GENERATION_FAILED, "streaming history bookkeeping mismatch");
        st->pos += T; st->steps += 1;
    }
    *outC = 1; *outT = Tc;
    return wav_dev;
}

void q3dec_decode_device(mis_q3dec* d, const int32_t* codes_dev, int batch, int T, float* wav_dev, int64_t wav_stride, hipStream_t s) {
    int C; int64_t Tt;
    q3dec_run(d, codes_dev, (int64_t)d->cfg.dec_num_quantizers * T, T, 1, batch, T, wav_dev, wav_stride, 0, &C, &Tt, s, nullptr);
}

// whole sequences straight from a strided code store (element (b, q, t) at codes_dev[b*cs_b + q*cs_q + t*cs_t])
void q3dec_decode_strided(mis_q3dec* d, const int32_t* codes_dev, int64_t cs_b, int64_t cs_q, int64_t cs_t, int batch, int T, float* wav_dev,
                          int64_t wav_stride, hipStream_t s) {
    int C; int64_t Tt;
    q3dec_run(d, codes_dev, cs_b, cs_q, cs_t, batch, T, wav_dev, wav_stride, 0, &C, &Tt, s, nullptr);
}

// ---- streaming session: resetStreamingState (:948-968) + streamingStep (:971-1006)
// cap_frames bounds the session length (K/V cache columns); chunk_cap the frames of one step (work buffers are sized up front so
// no allocation happens between steps).  dup_bias: reproduce the reference's double bias at chunk boundaries (see header).
void q3dec_stream_begin(mis_q3dec* d, int batch, int cap_frames, int chunk_cap, bool dup_bias, hipStream_t s) {
    MIS_REQUIRE(d->finalized, MIS_ERR_NOT_INITIALIZED, "speech tokenizer not finalized");
    MIS_REQUIRE(batch >= 1 && cap_frames >= 1 && chunk_cap >= 1, MIS_ERR_INVALID_INPUT, "bad streaming session sizes");
    HIP_CHECK(hipSetDevice(d->device));
    const mis_qwen3tts_config& cf = d->cfg;
    auto& st = d->st;
    st.open = true; st.dup_bias = dup_bias; st.batch = batch; st.cap_frames = cap_frames; st.pos = 0; st.steps = 0;
    st.hist_n = (size_t)batch * q3dec_hist_floats(d);
    st.hist.alloc(st.hist_n);
    st.kv.alloc((size_t)cf.dec_num_layers * batch * 2 * cf.dec_num_kv_heads * cf.dec_head_dim * cap_frames);
    HIP_CHECK(hipMemsetAsync(st.hist.p, 0, st.hist_n * 4, s));             // no history = the causal zero padding (:206)
    {   // size the work buffers for the largest chunk now
        const int up = q3dec_total_upsample(d);
        const int Cd = cf.dec_codebook_dim, ld = cf.dec_latent_dim, hs = cf.dec_hidden_size, I = cf.dec_intermediate_size, H = cf.dec_num_heads,
                  Hkv = cf.dec_num_kv_heads, D = cf.dec_head_dim;
        auto LD = [&](int64_t Tc) { return (size_t)(Q3_HP + round_up(Tc, 4)); };
        size_t need = (size_t)std::max({Cd, ld, (H + 2 * Hkv) * D, 2 * I, hs}) * LD(chunk_cap);
        int64_t Tc = chunk_cap;
        for (auto& U : d->ups) { Tc *= U.f; need = std::max(need, (size_t)4 * ld * LD(Tc)); }
        need = std::max(need, (size_t)cf.dec_decoder_dim * LD(Tc));
        for (auto& B : d->blocks) { need = std::max(need, (size_t)B.cin * LD(Tc)); Tc *= B.s; need = std::max(need, (size_t)B.cout * LD(Tc)); }
        (void)up;
        for (int i = 0; i < 4; ++i) d->buf[i].alloc((size_t)batch * need + 2 * Q3_HP);
    }
}
// the next Tn frames of every row: wav_dev [batch][wav_stride], Tn * samples_per_frame new samples per row
void q3dec_stream_step(mis_q3dec* d, const int32_t* codes_dev, int64_t cs_b, int64_t cs_q, int64_t cs_t, int Tn, float* wav_dev,
                       int64_t wav_stride, hipStream_t s) {
    auto& st = d->st;
    MIS_REQUIRE(st.open, MIS_ERR_NOT_INITIALIZED, "no streaming session is open");
    MIS_REQUIRE(Tn >= 1 && st.pos + Tn <= st.cap_frames, MIS_ERR_INVALID_INPUT, "streaming step of %d frames at position %d exceeds the session capacity %d",
                Tn, st.pos, st.cap_frames);
    int C; int64_t Tt;
    q3dec_run(d, codes_dev, cs_b, cs_q, cs_t, st.batch, Tn, wav_dev, wav_stride, 0, &C, &Tt, s, &st);
}
void q3dec_stream_end(mis_q3dec* d) { d->st.open = false; }
int q3dec_stream_pos(const mis_q3dec* d) { return d->st.open ? d->st.pos : -1; }

// host entry used by mis_qwen3tts_decode / debug taps: codes host or device [B][nq][T]
void q3dec_decode_host(mis_q3dec* d, const int32_t* codes, int batch, int T, float* out, int stop_after, int* outC, int64_t* outT, hipStream_t s) {
    HIP_CHECK(hipSetDevice(d->device));
    const int nq = d->cfg.dec_num_quantizers;
    d->codes_dev.alloc((size_t)batch * nq * T);
    HIP_CHECK(hipMemcpyAsync(d->codes_dev.p, codes, (size_t)batch * nq * T * 4, hipMemcpyDefault, s));
    const int64_t n = (int64_t)T * q3dec_total_upsample(d);
    DevBuf<float> wav;
    wav.alloc((size_t)batch * n);
    int C = 0; int64_t Tt = 0;
    const float* res = q3dec_run(d, d->codes_dev.p, (int64_t)nq * T, T, 1, batch, T, wav.p, n, stop_after, &C, &Tt, s, nullptr);
    if (out) HIP_CHECK(hipMemcpyAsync(out, res, (size_t)batch * C * Tt * 4, hipMemcpyDefault, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (outC) *outC = C;
    if (outT) *outT = Tt;
}
// streamingStep from host or device codes [B][nq][Tn] -> out (host or device) [B][Tn * samples_per_frame]
void q3dec_stream_step_host(mis_q3dec* d, const int32_t* codes, int Tn, float* out, hipStream_t s) {
    HIP_CHECK(hipSetDevice(d->device));
    const int nq = d->cfg.dec_num_quantizers, batch = d->st.batch;
    MIS_REQUIRE(d->st.open, MIS_ERR_NOT_INITIALIZED, "no streaming session is open");
    d->codes_dev.alloc((size_t)batch * nq * Tn);
    HIP_CHECK(hipMemcpyAsync(d->codes_dev.p, codes, (size_t)batch * nq * Tn * 4, hipMemcpyDefault, s));
    const int64_t n = (int64_t)Tn * q3dec_total_upsample(d);
    DevBuf<float> wav;
    wav.alloc((size_t)batch * n);
    q3dec_stream_step(d, d->codes_dev.p, (int64_t)nq * Tn, Tn, 1, Tn, wav.p, n, s);
    HIP_CHECK(hipMemcpyAsync(out, wav.p, (size_t)batch * n * 4, hipMemcpyDefault, s));
    HIP_CHECK(hipStreamSynchronize(s));
}
