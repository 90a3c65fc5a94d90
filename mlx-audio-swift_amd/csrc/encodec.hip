// encodec.hip - EnCodec (SEANet) decoder: residual-VQ codes -> waveform, float32.  Both model families: 24 kHz (mono, causal padding,
// plain convs) and 48 kHz (stereo, non-causal padding, GroupNorm(1 group) after every conv - in the transposed-conv layer before the trim).
//
// Reference being replaced: Encodec.decodeFrame / EncodecDecoder (Sources/MLXAudioCodecs/Encodec/Encodec.swift:94-170,295-302),
// EncodecLSTM / EncodecLSTMBlock, EncodecConv1d (causal + reflect padding), EncodecConvTranspose1dLayer, EncodecResnetBlock, ELU
// (EncodecLayers.swift:15-450), EncodecResidualVectorQuantizer.decode (EncodecQuantization.swift:117-133).  The reference runs its
// transposed conv as scalar host loops over asArray copies (EncodecLayers.swift:395-420) and the LSTM as a per-step host loop.
// Here: convs are exact-f32 MFMA contractions (k_conv_taps on a reflect-padded, ELU-activated copy; transposed convs as per-phase
// GEMMs), and the LSTM recurrence is ONE persistent launch: every block keeps its slice of W_h resident in LDS for all T steps,
// the hidden state is exchanged through global memory, one monotonic counter barrier per step (release fence -> atomic arrive;
// agent-scope poll -> acquire fence), bounded spins so a lost block cannot hang the GPU.
#include "common.h"
#include "kernels.h"
#include "codec_kernels.h"

#include <math.h>
#include <string.h>
#include <algorithm>
#include <map>

struct mis_encodec {
    int device = 0;
    mis_encodec_config cfg{};
    hipStream_t stream = nullptr;
    std::map<std::string, std::vector<float>> raw;
    std::map<std::string, std::vector<int64_t>> raw_shape;
    bool finalized = false;
    int n_q = 0, dim0 = 0;
    DevBuf<float> arena;
    struct Lin { size_t w = 0, b = 0, nw = 0, nb = 0; int M = 0, K = 0; };      // nw / nb: GroupNorm weight / bias (group_norm models)
    size_t tables = 0, zeros = 0;
    Lin conv0, last;
    struct Lstm { Lin xproj; size_t wh = 0; };
    std::vector<Lstm> lstm;
    struct Res { Lin c1, c2, sc; int dil; bool has_sc; };
    struct Up { Lin ct; int s, cin, cout; std::vector<Res> res; };
    std::vector<Up> ups;
    DevBuf<float> buf[4], hstate;
    CodecPack pack;                      // split-bf16 weight fragments + activation scratch (codec_bf3.hip)
    DevBuf<int32_t> codes_dev, sync;
    DevBuf<double> gn_part;              // GroupNorm partial sums [batch][GN_BLOCKS][2]
    DevBuf<float> gn_stat;               // [batch][2]: mean, 1 / sqrt(var + eps)
};

// ---------------------------------------------------------------------------- kernels
__global__ void k_encodec_embed(const int32_t* __restrict__ codes, const float* __restrict__ tables, float* __restrict__ z, int nq, int bins,
                                int C, int T) {
    const int t = blockIdx.x, b = blockIdx.y;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.0f;
        for (int q = 0; q < nq; ++q) {
            int code = min(max(codes[((size_t)b * nq + q) * T + t], 0), bins - 1);
            acc += tables[((size_t)q * bins + code) * C + c];
        }
        z[((size_t)b * C + c) * T + t] = acc;
    }
}

// y[c][i], i in [0, left + T + right): EncodecConv1d.pad1d (EncodecLayers.swift:130-171) then optional ELU (:340-350).
// reflect: left sample i <- x[min(left - i, T-1)], right sample i <- x[max(T-2-i, 0)]; zero mode: zeros.
__global__ void k_encodec_pad_act(const float* __restrict__ x, float* __restrict__ y, int C, int T, int left, int right, int reflect, int elu) {
    const int Tp = left + T + right;
    int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (i >= Tp) return;
    const float* xr = x + ((size_t)b * C + c) * T;
    float v;
    if (i < left) v = reflect ? xr[min(left - i, T - 1)] : 0.0f;
    else if (i < left + T) v = xr[i - left];
    else v = reflect ? xr[max(T - 2 - (i - left - T), 0)] : 0.0f;
    if (elu) v = v > 0.0f ? v : (expf(v) - 1.0f);
    y[((size_t)b * C + c) * Tp + i] = v;
}

__global__ void k_encodec_scale(float* __restrict__ x, const float* __restrict__ scale, int64_t n_per_row) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int b = blockIdx.y;
    if (i < n_per_row) x[(size_t)b * n_per_row + i] *= scale[b];
}


// ---- GroupNorm(1 group, pytorchCompatible) over the (C, T) block of one sample (EncodecLayers.swift:128-131): two deterministic
// reduction stages (fixed partition, fixed summation order; sums in double), then the affine apply - which also crops a column window
// out of a wider buffer (the transposed-conv layer normalises BEFORE it trims, :244-262) and adds the residual of a resnet block.
#define GN_BLOCKS 64
__global__ void __launch_bounds__(256) k_gn_partial(const float* __restrict__ x, double* __restrict__ part, int64_t n) {
    __shared__ double s1[256], s2[256];
    const int blk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int64_t len = (n + GN_BLOCKS - 1) / GN_BLOCKS, lo = (int64_t)blk * len, hi = lo + len < n ? lo + len : n;
    const float* xb = x + (size_t)b * n;
    double a = 0.0, q = 0.0;
    for (int64_t i = lo + tid; i < hi; i += 256) { const double v = (double)xb[i]; a += v; q += v * v; }
    s1[tid] = a; s2[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { s1[tid] += s1[tid + o]; s2[tid] += s2[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { part[((size_t)b * GN_BLOCKS + blk) * 2] = s1[0]; part[((size_t)b * GN_BLOCKS + blk) * 2 + 1] = s2[0]; }
}
__global__ void k_gn_final(const double* __restrict__ part, float* __restrict__ stat, double n, float eps) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    double a = 0.0, q = 0.0;
    for (int i = 0; i < GN_BLOCKS; ++i) { a += part[((size_t)b * GN_BLOCKS + i) * 2]; q += part[((size_t)b * GN_BLOCKS + i) * 2 + 1]; }
    const double mu = a / n;
    double var = q / n - mu * mu;
    if (var < 0.0) var = 0.0;
    stat[b * 2] = (float)mu;
    stat[b * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}
// y[b][c][t] = norm(xf[b][c][off + t]) * w[c] + bsh[c] (+ R[b][c][t]); stat null: plain crop.  xf rows are ldf wide, y / R rows T.
__global__ void k_gn_apply(const float* xf, int64_t ldf, int off, float* y, const float* __restrict__ stat, const float* __restrict__ w,
                           const float* __restrict__ bsh, const float* R, int C, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    float v = xf[((size_t)b * C + c) * ldf + off + t];
    if (stat) v = (v - stat[b * 2]) * stat[b * 2 + 1] * w[c] + bsh[c];
    const size_t o = ((size_t)b * C + c) * T + t;
    if (R) v += R[o];
    y[o] = v;
}

// LSTM recurrence (EncodecLSTM :33-61), persistent: block blk owns hidden units [blk*U, blk*U + U) = 4U gate rows of W_h
// (rows g*H + j, g = i,f,g,o) held in LDS.  xp [B][4H][T] = x W_x^T + b (computed by a GEMM), out [B][H][T], hbuf [2][B][H].
struct LstmArgs {
    const float* wh;     // [4H][H]
    const float* xp;     // [B][4H][T]
    const float* resid;  // optional [B][H][T] added to the output (EncodecLSTMBlock: h + hiddenStates), null for inner layers
    float* out;          // [B][H][T]
    float* hbuf;         // [2][B][H]
    int* counter;        // monotonic arrivals
    int* error;          // set to 1 on a barrier timeout
    int H, T, B, U, NB;
};
__global__ void __launch_bounds__(256) k_encodec_lstm(LstmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    const int H = a.H, U = a.U, R = 4 * U;
    float* wsl = lds;                 // [R][H]
    float* hs = lds + (size_t)R * H;  // [H]
    float* dots = hs + H;             // [R]
    float* cst = dots + R;            // [B][U] cell state
    for (int idx = tid; idx < R * H; idx += 256) {
        int r = idx / H, k = idx - r * H;
        int g = r / U, u = r - g * U;
        wsl[idx] = a.wh[((size_t)g * H + blk * U + u) * H + k];
    }
    for (int idx = tid; idx < a.B * U; idx += 256) cst[idx] = 0.0f;
    __syncthreads();
    // thread groups: G threads per gate row
    int G = 256 / R;
    if (G < 1) G = 1;
    if (G > 64) G = 64;
    while (G & (G - 1)) G &= G - 1;   // power of two
    const int rows_per_pass = 256 / G;
    for (int t = 0; t < a.T; ++t) {
        if (t > 0) {                  // wait for every block's step t-1
            if (tid == 0) {
                const int target = a.NB * t;
                int spins = 0;
                while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1 << 24)) { *a.error = 1; break; }
                }
            }
            __syncthreads();
            __threadfence();          // acquire: hbuf written by the other blocks is visible below
            if (*a.error) return;
        }
        for (int b = 0; b < a.B; ++b) {
            if (t > 0) {
                const float* hp = a.hbuf + ((size_t)((t - 1) & 1) * a.B + b) * H;
                for (int k = tid; k < H; k += 256) hs[k] = __hip_atomic_load(hp + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                for (int r0 = 0; r0 < R; r0 += rows_per_pass) {
                    const int r = r0 + tid / G, gl = tid % G;
                    float acc = 0.0f;
                    if (r < R)
                        for (int k = gl; k < H; k += G) acc += wsl[(size_t)r * H + k] * hs[k];
                    for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
                    if (r < R && gl == 0) dots[r] = acc;
                }
                __syncthreads();
            }
            if (tid < U) {
                const int j = blk * U + tid;
                const float* xb = a.xp + (size_t)b * 4 * H * a.T + t;
                float gi = xb[(size_t)(0 * H + j) * a.T], gf = xb[(size_t)(1 * H + j) * a.T], gg = xb[(size_t)(2 * H + j) * a.T],
                      go = xb[(size_t)(3 * H + j) * a.T];
                if (t > 0) { gi += dots[0 * U + tid]; gf += dots[1 * U + tid]; gg += dots[2 * U + tid]; go += dots[3 * U + tid]; }
                const float i_ = 1.0f / (1.0f + expf(-gi)), f_ = 1.0f / (1.0f + expf(-gf)), g_ = tanhf(gg), o_ = 1.0f / (1.0f + expf(-go));
                const float c = f_ * cst[b * U + tid] + i_ * g_;
                cst[b * U + tid] = c;
                const float h = o_ * tanhf(c);
                __hip_atomic_store(a.hbuf + ((size_t)(t & 1) * a.B + b) * H + j, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const size_t oi = ((size_t)b * H + j) * a.T + t;
                a.out[oi] = a.resid ? h + a.resid[oi] : h;
            }
            __syncthreads();
        }
        __threadfence();              // release this block's hbuf stores
        if (tid == 0) __hip_atomic_fetch_add(a.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------- host
extern "C" mis_status mis_encodec_create(const mis_encodec_config* cfg, int device, mis_encodec** out) {
    MIS_API_BEGIN
    MIS_REQUIRE(cfg && out, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(cfg->n_upsampling_ratios >= 1 && cfg->n_upsampling_ratios <= 8 && cfg->num_filters >= 1 && cfg->hidden_size >= 1 &&
                    cfg->codebook_size >= 1 && cfg->n_quantizers >= 1 && (cfg->audio_channels == 1 || cfg->audio_channels == 2) && cfg->compress >= 1,
                MIS_ERR_INVALID_INPUT, "bad Encodec config (one or two audio channels)");
    MIS_REQUIRE(cfg->kernel_size <= 7 && cfg->last_kernel_size <= 7 && cfg->residual_kernel_size <= 7, MIS_ERR_INVALID_INPUT, "kernel sizes above 7 are not built");
    int n = 0;
    HIP_CHECK(hipGetDeviceCount(&n));
    MIS_REQUIRE(device >= 0 && device < n, MIS_ERR_DEVICE, "device %d not available (%d GPUs visible)", device, n);
    HIP_CHECK(hipSetDevice(device));
    mis_encodec* c = new mis_encodec();
    c->device = device; c->cfg = *cfg; c->n_q = cfg->n_quantizers;
    c->dim0 = cfg->num_filters << cfg->n_upsampling_ratios;
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    *out = c;
    MIS_API_END
}
extern "C" void mis_encodec_destroy(mis_encodec* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    delete c;
}
extern "C" int mis_encodec_hop_length(const mis_encodec* c) {
    if (!c) return 0;
    int h = 1;
    for (int i = 0; i < c->cfg.n_upsampling_ratios; ++i) h *= c->cfg.upsampling_ratios[i];
    return h;
}

extern "C" mis_status mis_encodec_set_tensor(mis_encodec* c, const char* name_, const void* data, mis_dtype dtype, const int64_t* shape, int ndim) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && name_ && data && shape && ndim >= 1 && ndim <= 3, MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(!c->finalized, MIS_ERR_INVALID_INPUT, "set_tensor after finalize");
    std::string name = name_;
    if (name.rfind("encoder.", 0) == 0) return MIS_OK;                    // encode path: not built
    size_t n = 1;
    std::vector<int64_t> sh;
    for (int i = 0; i < ndim; ++i) { MIS_REQUIRE(shape[i] > 0, MIS_ERR_INVALID_INPUT, "bad shape"); n *= (size_t)shape[i]; sh.push_back(shape[i]); }
    HIP_CHECK(hipSetDevice(c->device));
    size_t esz = dtype == MIS_F32 ? 4 : 2;
    std::vector<uint8_t> host(n * esz);
    HIP_CHECK(hipMemcpy(host.data(), data, n * esz, hipMemcpyDefault));
    std::vector<float> v(n);
    if (dtype == MIS_F32) memcpy(v.data(), host.data(), n * 4);
    else if (dtype == MIS_BF16) for (size_t i = 0; i < n; ++i) v[i] = bf16_to_f32(((bf16_t*)host.data())[i]);
    else if (dtype == MIS_F16) for (size_t i = 0; i < n; ++i) v[i] = f16_to_f32_host(((uint16_t*)host.data())[i]);
    else throw MisError(MIS_ERR_INVALID_INPUT, "unsupported dtype");
    c->raw[name] = std::move(v);
    c->raw_shape[name] = sh;
    MIS_API_END
}

static const std::vector<float>& eneed(mis_encodec* c, const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = c->raw.find(name);
    MIS_REQUIRE(it != c->raw.end(), MIS_ERR_NOT_INITIALIZED, "Encodec weight missing: %s", name.c_str());
    MIS_REQUIRE(c->raw_shape[name] == std::vector<int64_t>(shape), MIS_ERR_INVALID_INPUT, "Encodec weight %s has the wrong shape", name.c_str());
    return it->second;
}

extern "C" mis_status mis_encodec_finalize(mis_encodec* c) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && !c->finalized, MIS_ERR_INVALID_INPUT, "bad handle");
    HIP_CHECK(hipSetDevice(c->device));
    const mis_encodec_config& cf = c->cfg;
    std::vector<float> arena;
    auto push = [&](const std::vector<float>& v) { size_t o = arena.size(); arena.insert(arena.end(), v.begin(), v.end()); while (arena.size() & 3) arena.push_back(0.f); return o; };
    auto conv = [&](const std::string& p, int64_t co, int64_t k, int64_t ci) {              // [co][k][ci] -> A^T [(j*ci + c)][co]
        const auto& w = eneed(c, p + ".conv.weight", {co, k, ci});
        std::vector<float> at((size_t)k * ci * co);
        for (int64_t o = 0; o < co; ++o) for (int64_t j = 0; j < k; ++j) for (int64_t i = 0; i < ci; ++i) at[(j * ci + i) * co + o] = w[(o * k + j) * ci + i];
        mis_encodec::Lin L; L.M = (int)co; L.K = (int)(k * ci); L.w = push(at); L.b = push(eneed(c, p + ".conv.bias", {co}));
        if (cf.group_norm) { L.nw = push(eneed(c, p + ".norm.weight", {co})); L.nb = push(eneed(c, p + ".norm.bias", {co})); }
        return L;
    };
    {
        const int64_t D = cf.codebook_dim;
        std::vector<float> tables((size_t)c->n_q * cf.codebook_size * D);
        for (int q = 0; q < c->n_q; ++q) {
            const auto& e = eneed(c, "quantizer.layers." + std::to_string(q) + ".codebook.embed", {cf.codebook_size, D});
            memcpy(tables.data() + (size_t)q * cf.codebook_size * D, e.data(), e.size() * 4);
        }
        c->tables = push(tables);
    }
    int64_t dim = c->dim0;
    c->zeros = push(std::vector<float>((size_t)dim, 0.0f));
    c->conv0 = conv("decoder.layers.0", dim, cf.kernel_size, cf.hidden_size);
    c->lstm.clear();
    for (int j = 0; j < cf.num_lstm_layers; ++j) {
        const std::string p = "decoder.layers.1.lstm." + std::to_string(j);
        mis_encodec::Lstm L;
        const auto& wx = eneed(c, p + ".Wx", {4 * dim, dim});
        std::vector<float> at((size_t)dim * 4 * dim);
        for (int64_t o = 0; o < 4 * dim; ++o) for (int64_t i = 0; i < dim; ++i) at[i * 4 * dim + o] = wx[o * dim + i];
        L.xproj.M = (int)(4 * dim); L.xproj.K = (int)dim; L.xproj.w = push(at); L.xproj.b = push(eneed(c, p + ".bias", {4 * dim}));
        L.wh = push(eneed(c, p + ".Wh", {4 * dim, dim}));
        c->lstm.push_back(L);
    }
    c->ups.clear();
    int li = 2;
    for (int bi = 0; bi < cf.n_upsampling_ratios; ++bi) {
        const int64_t s = cf.upsampling_ratios[bi], k = 2 * s, cin = dim, cout = dim / 2;
        mis_encodec::Up U;
        U.s = (int)s; U.cin = (int)cin; U.cout = (int)cout;
        {   // causal transposed conv: out[s*n + ph] = sum_j sum_c W[co][ph + s*j][c] x[c][n - j]  (full conv, right trim k - s)
            const std::string p = "decoder.layers." + std::to_string(li + 1);
            const auto& w = eneed(c, p + ".conv.weight", {cout, k, cin});
            std::vector<float> at((size_t)s * 2 * cin * cout);
            for (int64_t ph = 0; ph < s; ++ph) for (int64_t j = 0; j < 2; ++j) for (int64_t i = 0; i < cin; ++i) for (int64_t o = 0; o < cout; ++o)
                at[((ph * 2 + j) * cin + i) * cout + o] = w[(o * k + (ph + s * j)) * cin + i];
            U.ct.M = (int)cout; U.ct.K = (int)(2 * cin); U.ct.w = push(at); U.ct.b = push(eneed(c, p + ".conv.bias", {cout}));
            if (cf.group_norm) { U.ct.nw = push(eneed(c, p + ".norm.weight", {cout})); U.ct.nb = push(eneed(c, p + ".norm.bias", {cout})); }
        }
        li += 2;
        dim = cout;
        for (int j = 0; j < cf.num_residual_layers; ++j) {
            const std::string p = "decoder.layers." + std::to_string(li);
            mis_encodec::Res R;
            R.dil = 1;
            for (int e = 0; e < j; ++e) R.dil *= cf.dilation_growth_rate;
            const int64_t hid = dim / cf.compress;
            R.c1 = conv(p + ".block.1", hid, cf.residual_kernel_size, dim);
            R.c2 = conv(p + ".block.3", dim, 1, hid);
            R.has_sc = cf.use_conv_shortcut != 0;
            if (R.has_sc) R.sc = conv(p + ".shortcut", dim, 1, dim);
            U.res.push_back(R);
            ++li;
        }
        c->ups.push_back(U);
    }
    c->last = conv("decoder.layers." + std::to_string(li + 1), cf.audio_channels, cf.last_kernel_size, dim);
    c->arena.alloc(arena.size());
    HIP_CHECK(hipMemcpy(c->arena.p, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    c->raw.clear(); c->raw_shape.clear();
    c->finalized = true;
    MIS_API_END
}

// stage: 0 waveform; 1 conv0; 2 lstm block; 3 + i upsampling block i
static const float* encodec_run(mis_encodec* c, const int32_t* codes_dev, int nq, int batch, int T, float* wav_dev, int stage, int* outC, int64_t* outT) {
    CodecPackScope pack_scope(&c->pack);
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "Encodec model not finalized");
    const mis_encodec_config& cf = c->cfg;
    hipStream_t s = c->stream;
    const float* W = c->arena.p;
    const int hop = mis_encodec_hop_length(c), H = c->dim0;
    const bool causal = cf.use_causal_conv != 0;
    const int reflect = cf.pad_reflect;
    size_t need_elems = (size_t)std::max(4 * H, cf.hidden_size) * (T + 16);
    {
        int64_t Tc = T; int dim = H;
        for (auto& U : c->ups) { Tc *= U.s; dim = U.cout; need_elems = std::max(need_elems, (size_t)(2 * U.cin) * (Tc + 16)); (void)dim; }
    }
    for (int i = 0; i < 4; ++i) c->buf[i].alloc((size_t)batch * need_elems);
    float *x = c->buf[0].p, *y = c->buf[1].p, *t1 = c->buf[2].p, *t2 = c->buf[3].p;
    const bool gn = cf.group_norm != 0;
    if (gn) { c->gn_part.alloc((size_t)batch * GN_BLOCKS * 2); c->gn_stat.alloc((size_t)batch * 2); }
    // GroupNorm over the [C][ldf] block of each sample in xf, applied to the column window [off, off + T) -> y [C][T] (+ R)
    auto group_norm = [&](const float* xf, int64_t ldf, int off, float* Y, const mis_encodec::Lin& L, const float* R, int Tn) {
        const int64_t n = (int64_t)L.M * ldf;
        hipLaunchKernelGGL(k_gn_partial, dim3(GN_BLOCKS, batch), dim3(256), 0, s, xf, c->gn_part.p, n);
        hipLaunchKernelGGL(k_gn_final, dim3(batch), dim3(64), 0, s, c->gn_part.p, c->gn_stat.p, (double)n, 1e-5f);
        hipLaunchKernelGGL(k_gn_apply, dim3(cdiv(Tn, 256), L.M, batch), dim3(256), 0, s, xf, ldf, off, Y, c->gn_stat.p, W + L.nw, W + L.nb, R, L.M, Tn);
    };
    // EncodecConv1d (:84-214), stride 1: pad (left = k - 1 causal / split otherwise; dilation not included: as the reference), conv
    auto conv1d = [&](const mis_encodec::Lin& L, int Cin, int k, int dil, const float* X, float* Y, int Tin, int elu, const float* R) {
        const int keff = (k - 1) * dil + 1, ptotal = k - 1;
        int left = ptotal, right = 0;
        if (!causal) { right = ptotal / 2; left = ptotal - right; }
        const int Tp = left + Tin + right, Tout = Tp - keff + 1;
        MIS_REQUIRE(Tout >= 1, MIS_ERR_INVALID_INPUT, "input too short for the dilated conv");
        const float* src = X;
        if (left + right > 0 || elu) {
            hipLaunchKernelGGL(k_encodec_pad_act, dim3(cdiv(Tp, 256), Cin, batch), dim3(256), 0, s, X, t2, Cin, Tin, left, right, reflect, elu);
            src = t2;
        }
        GemmParams g{};
        g.AT = W + L.w; g.bias = W + L.b; g.X = src; g.Y = Y; g.R = gn ? nullptr : R; g.M = L.M; g.K = L.K; g.N = Tout; g.Tin = Tp; g.Tout = Tout;
        g.Cin = Cin; g.taps = k; g.dil = dil; g.pad = 0;
        launch_gemm(GEMM_TAPS, false, g, batch, s);
        if (gn) group_norm(Y, Tout, 0, Y, L, R, Tout);          // norm(conv(x)) (+ the residual, which the plain path adds in the GEMM epilogue)
        return Tout;
    };
    hipLaunchKernelGGL(k_encodec_embed, dim3(T, batch), dim3(256), 0, s, codes_dev, W + c->tables, x, nq, cf.codebook_size, cf.codebook_dim, T);
    int Tc = conv1d(c->conv0, cf.hidden_size, cf.kernel_size, 1, x, y, T, 0, nullptr);
    std::swap(x, y);
    if (stage == 1) { *outC = H; *outT = Tc; return x; }
    if (!c->lstm.empty()) {   // EncodecLSTMBlock (:66-80): h = lstm_n(...lstm_1(x)) + x
        int U = H, NB = 1;
        while ((size_t)4 * U * H * 4 > 48 * 1024 && U > 1) { U = (U + 1) / 2; }
        while (H % U) --U;
        NB = H / U;
        const size_t smem = ((size_t)4 * U * H + H + 4 * U + (size_t)batch * U) * sizeof(float);
        MIS_REQUIRE(smem <= 64 * 1024 && NB <= 256, MIS_ERR_INVALID_INPUT, "LSTM width %d does not fit the persistent kernel", H);
        c->hstate.alloc((size_t)2 * batch * H);
        c->sync.alloc(2);
        const float* in = x;
        float* cur = y;
        for (size_t j = 0; j < c->lstm.size(); ++j) {
            GemmParams g{};
            g.AT = W + c->lstm[j].xproj.w; g.bias = W + c->lstm[j].xproj.b; g.X = in; g.Y = t1; g.M = 4 * H; g.K = H; g.N = Tc; g.Tin = Tc; g.Tout = Tc;
            launch_gemm(GEMM_PLAIN, false, g, batch, s);
            HIP_CHECK(hipMemsetAsync(c->sync.p, 0, 8, s));
            LstmArgs la{};
            la.wh = W + c->lstm[j].wh; la.xp = t1; la.resid = (j + 1 == c->lstm.size()) ? x : nullptr; la.out = cur; la.hbuf = c->hstate.p;
            la.counter = c->sync.p; la.error = c->sync.p + 1; la.H = H; la.T = Tc; la.B = batch; la.U = U; la.NB = NB;
            hipLaunchKernelGGL(k_encodec_lstm, dim3(NB), dim3(256), smem, s, la);
            in = cur;
            cur = (cur == y) ? t2 : y;
        }
        int32_t flags[2] = {0, 0};
        HIP_CHECK(hipMemcpyAsync(flags, c->sync.p, 8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        MIS_REQUIRE(flags[1] == 0, MIS_ERR_AUDIO_DECODE, "LSTM step barrier timed out");
        if (in != x) { HIP_CHECK(hipMemcpyAsync(x, in, (size_t)batch * H * Tc * 4, hipMemcpyDeviceToDevice, s)); }
    }
    if (stage == 2) { *outC = H; *outT = Tc; return x; }
    int bi = 0;
    for (auto& U : c->ups) {
        // ELU -> transposed conv (EncodecConvTranspose1dLayer :218-262), kernel 2s: the full output has s * T + s samples, of which
        // padding_total = s are trimmed - all on the right when causal (trim_right_ratio 1), s / 2 on the right otherwise
        hipLaunchKernelGGL(k_encodec_pad_act, dim3(cdiv(Tc, 256), U.cin, batch), dim3(256), 0, s, x, t1, U.cin, Tc, 0, 0, 0, 1);
        GemmParams g{};
        g.AT = W + U.ct.w; g.bias = W + U.ct.b; g.X = t1; g.Y = y; g.alpha = W + c->zeros; g.ralpha = W + c->zeros;
        g.M = U.cout; g.K = 2 * U.cin; g.N = Tc; g.Tin = Tc; g.Tout = Tc * U.s; g.s = U.s; g.pad = 0; g.Cin = U.cin;
        if (causal && cf.trim_right_ratio == 1.0f && !gn) {
            launch_gemm(GEMM_CONVT, true, g, batch, s);            // the trimmed range is exactly the first s * T samples
            Tc *= U.s;
            std::swap(x, y);
        } else {
            MIS_REQUIRE(!causal || cf.trim_right_ratio == 1.0f, MIS_ERR_INVALID_INPUT, "causal transposed convs need trim_right_ratio 1");
            const int right = causal ? U.s : U.s / 2, left = U.s - right;
            const int Tfull = Tc * U.s + U.s;
            g.N = Tc + 1; g.Tout = Tfull;                          // frame n = T holds the tail of the last input column
            launch_gemm(GEMM_CONVT, true, g, batch, s);
            Tc *= U.s;
            // GroupNorm over the UNTRIMMED output (:244-262), then the trim; without a norm the same kernel only crops
            if (gn) group_norm(y, Tfull, left, t1, U.ct, nullptr, Tc);
            else hipLaunchKernelGGL(k_gn_apply, dim3(cdiv(Tc, 256), U.cout, batch), dim3(256), 0, s, y, (int64_t)Tfull, left, t1, nullptr, nullptr, nullptr, nullptr, U.cout, Tc);
            std::swap(x, t1);
        }
        for (auto& R : U.res) {   // EncodecResnetBlock (:266-325): shortcut(x) + conv1(ELU(conv_k(ELU(x))))
            const int hid = R.c1.M;
            int T1 = conv1d(R.c1, U.cout, cf.residual_kernel_size, R.dil, x, t1, Tc, 1, nullptr);
            MIS_REQUIRE(T1 == Tc, MIS_ERR_INVALID_INPUT, "residual conv changes the length (dilation > 1 is not built)");
            const float* res = x;
            if (R.has_sc) { conv1d(R.sc, U.cout, 1, 1, x, y, Tc, 0, nullptr); res = y; }
            // second conv (k = 1) on ELU(t1) with the residual epilogue; the ELU copy lands in t2 inside conv1d
            float* outb = (res == y) ? t1 : y;              // t1 is consumed through t2 before outb is written
            conv1d(R.c2, hid, 1, 1, t1, outb, Tc, 1, res);
            if (outb == t1) { std::swap(x, t1); } else { std::swap(x, y); }
        }
        if (stage == 3 + bi) { *outC = U.cout; *outT = Tc; return x; }
        ++bi;
    }
    MIS_REQUIRE(Tc == T * hop, MIS_ERR_AUDIO_DECODE, "internal length mismatch");
    conv1d(c->last, c->ups.back().cout, cf.last_kernel_size, 1, x, wav_dev, Tc, 1, nullptr);
    HIP_CHECK(hipGetLastError());
    *outC = cf.audio_channels; *outT = Tc;
    return wav_dev;
}

// decodeFrame (Encodec.swift:295-302): codes int32 [batch, n_q, T] (host or device), scales f32 [batch] or NULL -> wav [batch, channels, T*hop]
extern "C" mis_status mis_encodec_decode_frame(mis_encodec* c, const int32_t* codes, int batch, int n_q, int T, const float* scales, float* wav_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && codes && wav_out && batch >= 1 && T >= 1 && n_q >= 1 && n_q <= c->n_q, MIS_ERR_INVALID_INPUT, "bad argument");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int64_t n = (int64_t)T * mis_encodec_hop_length(c) * c->cfg.audio_channels;
    c->codes_dev.alloc((size_t)batch * n_q * T);
    HIP_CHECK(hipMemcpyAsync(c->codes_dev.p, codes, (size_t)batch * n_q * T * 4, hipMemcpyDefault, s));
    DevBuf<float> wav, sc;
    wav.alloc((size_t)batch * n);
    int C; int64_t Tt;
    encodec_run(c, c->codes_dev.p, n_q, batch, T, wav.p, 0, &C, &Tt);
    if (scales) {
        sc.alloc(batch);
        HIP_CHECK(hipMemcpyAsync(sc.p, scales, batch * 4, hipMemcpyDefault, s));
        hipLaunchKernelGGL(k_encodec_scale, dim3(cdiv(n, 256), batch), dim3(256), 0, s, wav.p, sc.p, n);
    }
    HIP_CHECK(hipMemcpyAsync(wav_out, wav.p, (size_t)batch * n * 4, hipMemcpyDefault, s));
    HIP_CHECK(hipStreamSynchronize(s));
    MIS_API_END
}
extern "C" mis_status mis_encodec_debug_tap(mis_encodec* c, const int32_t* codes, int batch, int n_q, int T, int stage, float* out, int64_t capacity,
                                            int32_t* channels, int64_t* length) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && codes && out && channels && length && stage >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    c->codes_dev.alloc((size_t)batch * n_q * T);
    HIP_CHECK(hipMemcpyAsync(c->codes_dev.p, codes, (size_t)batch * n_q * T * 4, hipMemcpyDefault, s));
    DevBuf<float> wav;
    wav.alloc((size_t)batch * T * mis_encodec_hop_length(c) * c->cfg.audio_channels);
    int C = 0; int64_t Tt = 0;
    const float* res = encodec_run(c, c->codes_dev.p, n_q, batch, T, wav.p, stage, &C, &Tt);
    MIS_REQUIRE((int64_t)batch * C * Tt <= capacity, MIS_ERR_INVALID_INPUT, "tap buffer too small");
    HIP_CHECK(hipMemcpyAsync(out, res, (size_t)batch * C * Tt * 4, hipMemcpyDefault, s));
    HIP_CHECK(hipStreamSynchronize(s));
    *channels = C; *length = Tt;
    MIS_API_END
}
