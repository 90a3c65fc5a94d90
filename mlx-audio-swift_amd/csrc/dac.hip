// dac.hip - Descript DAC codec: decoder (residual-VQ codes -> waveform) and encoder (waveform -> codes), float32.
//
// Reference being replaced: DescriptDAC.decodeFromCodes / decode (Sources/MLXAudioCodecs/Descript/DescriptDAC.swift:235-242),
// DescriptDecoder / DescriptDecoderBlock / DescriptResidualUnit (:7-32,103-160), DescriptResidualVectorQuantize.fromCodes
// (DescriptQuantization.swift:150-163), weight-normalised convs (BigVGAN/BigVGANLayers.swift:113-225).  The reference keeps NLC
// activations and recomputes the weight norm on every call; here weights are folded once (A^T per tap / per transposed-conv
// phase, codebook x out_proj tables) and activations stay NCT.  Dense k7 dilated convs run on k_conv_taps, 1x1 convs and the
// transposed convs on k_snac_gemm (exact-f32 MFMA), Snake fused into the operand loads.
// Encode (DescriptEncoder / DescriptEncoderBlock :40-95, preprocess + encode :216-233, DescriptVectorQuantize / residual loop
// DescriptQuantization.swift:54-147): first conv k7, per block 3 residual units (the decoder's: dense dilated k7 on k_conv_taps + 1x1
// with residual epilogue) then Snake + the stride-s conv (k = 2s, pad ceil(s/2)) as a 3-tap conv over the phase-split tensor, last
// conv k3; residual VQ: in_proj GEMM, nearest code of the L2-normalised latent (first index on ties), residual -= the decode table row.
#include "common.h"
#include "kernels.h"
#include "codec_kernels.h"

#include <math.h>
#include <string.h>
#include <algorithm>
#include <map>

struct mis_dac {
    int device = 0;
    mis_dac_config cfg{};
    int latent = 0;
    hipStream_t stream = nullptr;
    std::map<std::string, std::vector<float>> raw;
    std::map<std::string, std::vector<int64_t>> raw_shape;
    bool finalized = false;
    DevBuf<float> arena;
    struct Lin { size_t w = 0, b = 0; int M = 0, K = 0; };
    struct RU { size_t a1, ra1, a2, ra2; Lin c1, c2; int dil; };
    struct Blk { size_t a, ra; Lin ct; int s, pad, cin, cout; RU ru[3]; };
    size_t tables = 0;
    Lin first;
    std::vector<Blk> blocks;
    size_t fin_a = 0, fin_ra = 0, fin_w = 0;
    float fin_b = 0.0f;
    int fin_c = 0;
    DevBuf<float> buf[3];
    CodecPack pack;                      // split-bf16 weight fragments + activation scratch (codec_bf3.hip)
    DevBuf<int32_t> codes_dev;
    // encoder (optional: built when the checkpoint carries encoder.* / in_proj tensors)
    bool has_encoder = false;
    int enc_dim = 0, hop = 1;
    size_t enc_first_w = 0, enc_first_b = 0;
    struct EncBlk { RU ru[3]; size_t a, ra; Lin down; int cin, cout, stride; };
    std::vector<EncBlk> enc_blocks;
    size_t enc_fin_a = 0, enc_fin_ra = 0;
    Lin enc_last;
    struct VqEnc { Lin in_proj; size_t cn, cn2; };
    std::vector<VqEnc> vq_enc;
    DevBuf<float> ebuf[3], vq_ze;
    DevBuf<int32_t> enc_codes;
};

__global__ void k_dac_embed(const int32_t* __restrict__ codes, const float* __restrict__ tables, float* __restrict__ z, int ncb, int bins,
                            int C, int T) {
    const int t = blockIdx.x, b = blockIdx.y;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.0f;
        for (int q = 0; q < ncb; ++q) {
            int code = min(max(codes[((size_t)b * ncb + q) * T + t], 0), bins - 1);
            acc += tables[((size_t)q * bins + code) * C + c];
        }
        z[((size_t)b * C + c) * T + t] = acc;
    }
}

// Snake -> conv k7 "same" (C -> 1) -> tanh   (DescriptDAC.swift:146-148)
__global__ void k_dac_final(const float* __restrict__ x, float* __restrict__ out, int64_t out_stride, const float* __restrict__ w /*[7][C]*/,
                            float bias, const float* __restrict__ a, const float* __restrict__ ra, int C, int T) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (t >= T) return;
    float acc = bias;
    for (int c = 0; c < C; ++c) {
        const float* xr = x + ((size_t)b * C + c) * T;
        const float ac = a[c], rc = ra[c];
        for (int j = 0; j < 7; ++j) {
            int ts = t + j - 3;
            if (ts < 0 || ts >= T) continue;
            const float v = xr[ts];
            acc += w[j * C + c] * fmaf(rc, mis_sin_sq(ac * v), v);
        }
    }
    out[(size_t)b * out_stride + t] = tanhf(acc);
}

extern "C" mis_status mis_dac_create(const mis_dac_config* cfg, int device, mis_dac** out) {
    MIS_API_BEGIN
    MIS_REQUIRE(cfg && out, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(cfg->n_decoder_rates >= 1 && cfg->n_decoder_rates <= 8 && cfg->decoder_dim >> cfg->n_decoder_rates >= 1 && cfg->n_codebooks >= 1 &&
                    cfg->codebook_size >= 1 && cfg->codebook_dim >= 1 && cfg->latent_dim >= 1,
                MIS_ERR_INVALID_INPUT, "bad DAC config");
    int n = 0;
    HIP_CHECK(hipGetDeviceCount(&n));
    MIS_REQUIRE(device >= 0 && device < n, MIS_ERR_DEVICE, "device %d not available (%d GPUs visible)", device, n);
    HIP_CHECK(hipSetDevice(device));
    mis_dac* c = new mis_dac();
    c->device = device; c->cfg = *cfg; c->latent = cfg->latent_dim;
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    *out = c;
    MIS_API_END
}
extern "C" void mis_dac_destroy(mis_dac* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    delete c;
}
// (T-1)*s - 2*ceil(s/2) + (2s-1) + 1 + output_padding(1) per block (DescriptDAC.swift:110-117)
extern "C" int64_t mis_dac_num_samples(const mis_dac* c, int n_frames) {
    if (!c || n_frames < 1) return 0;
    int64_t T = n_frames;
    for (int i = 0; i < c->cfg.n_decoder_rates; ++i) { int s = c->cfg.decoder_rates[i]; T = (T - 1) * s - 2 * ((s + 1) / 2) + 2 * s + 1; }
    return T;
}

extern "C" mis_status mis_dac_set_tensor(mis_dac* c, const char* name_, const void* data, mis_dtype dtype, const int64_t* shape, int ndim) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && name_ && data && shape && ndim >= 1 && ndim <= 3, MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(!c->finalized, MIS_ERR_INVALID_INPUT, "set_tensor after finalize");
    std::string name = name_;                                           // DescriptDAC.sanitize (:274-286)
    for (auto rep : {std::pair<const char*, const char*>{".layers.", "."}, {".in_proj.", ".inProj."}, {".out_proj.", ".outProj."}}) {
        size_t pos;
        while ((pos = name.find(rep.first)) != std::string::npos) name.replace(pos, strlen(rep.first), rep.second);
    }
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

static const std::vector<float>& dneed(mis_dac* c, const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = c->raw.find(name);
    MIS_REQUIRE(it != c->raw.end(), MIS_ERR_NOT_INITIALIZED, "DAC weight missing: %s", name.c_str());
    MIS_REQUIRE(c->raw_shape[name] == std::vector<int64_t>(shape), MIS_ERR_INVALID_INPUT, "DAC weight %s has the wrong shape", name.c_str());
    return it->second;
}

extern "C" mis_status mis_dac_finalize(mis_dac* c) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && !c->finalized, MIS_ERR_INVALID_INPUT, "bad handle");
    HIP_CHECK(hipSetDevice(c->device));
    const mis_dac_config& cf = c->cfg;
    const int64_t D = c->latent, cd = cf.codebook_dim, bins = cf.codebook_size, dd = cf.decoder_dim;
    std::vector<float> arena;
    auto push = [&](const std::vector<float>& v) { size_t o = arena.size(); arena.insert(arena.end(), v.begin(), v.end()); while (arena.size() & 3) arena.push_back(0.f); return o; };
    // w = g * v / (||v|| + 1e-12), norm over all axes except `keep` (0: per output channel; 2: per input channel)
    auto wn = [&](const std::string& p, int64_t co, int64_t k, int64_t ci, int keep) {
        const auto& v = dneed(c, p + ".weight_v", {co, k, ci});
        const auto& g = keep == 0 ? dneed(c, p + ".weight_g", {co, 1, 1}) : dneed(c, p + ".weight_g", {1, 1, ci});
        std::vector<float> w(v.size());
        const int64_t groups = keep == 0 ? co : ci;
        std::vector<double> nrm(groups, 0.0);
        for (int64_t o = 0; o < co; ++o) for (int64_t j = 0; j < k; ++j) for (int64_t i = 0; i < ci; ++i) {
            double x = v[(o * k + j) * ci + i];
            nrm[keep == 0 ? o : i] += x * x;
        }
        for (int64_t o = 0; o < co; ++o) for (int64_t j = 0; j < k; ++j) for (int64_t i = 0; i < ci; ++i) {
            const int64_t gi = keep == 0 ? o : i;
            w[(o * k + j) * ci + i] = g[gi] * v[(o * k + j) * ci + i] / ((float)sqrt(nrm[gi]) + 1e-12f);
        }
        return w;
    };
    auto conv = [&](const std::string& p, int64_t co, int64_t k, int64_t ci) {              // -> A^T [(j*ci + c)][co]
        std::vector<float> w = wn(p, co, k, ci, 0), at((size_t)k * ci * co);
        for (int64_t o = 0; o < co; ++o) for (int64_t j = 0; j < k; ++j) for (int64_t i = 0; i < ci; ++i) at[(j * ci + i) * co + o] = w[(o * k + j) * ci + i];
        mis_dac::Lin L; L.M = (int)co; L.K = (int)(k * ci); L.w = push(at); L.b = push(dneed(c, p + ".bias", {co}));
        return L;
    };
    auto snake = [&](const std::string& p, int64_t C, size_t& a, size_t& ra) {
        const auto& al = dneed(c, p, {1, 1, C});
        std::vector<float> rv(C);
        for (int64_t i = 0; i < C; ++i) rv[i] = 1.0f / (al[i] + 1e-9f);
        a = push(al); ra = push(rv);
    };
    {   // fromCodes: sum_i outProj_i(codebook_i[code]) -> tables [n_cb][bins][D] (bias included once per codebook)
        std::vector<float> tables((size_t)cf.n_codebooks * bins * D);
        for (int q = 0; q < cf.n_codebooks; ++q) {
            const std::string p = "quantizer.quantizers." + std::to_string(q);
            const auto& cb = dneed(c, p + ".codebook.weight", {bins, cd});
            std::vector<float> w = wn(p + ".outProj", D, 1, cd, 0);
            const auto& b = dneed(c, p + ".outProj.bias", {D});
            for (int64_t v = 0; v < bins; ++v) for (int64_t o = 0; o < D; ++o) {
                float acc = 0.0f;
                for (int64_t k = 0; k < cd; ++k) acc += w[o * cd + k] * cb[v * cd + k];
                tables[((size_t)q * bins + v) * D + o] = acc + b[o];
            }
        }
        c->tables = push(tables);
    }
    c->first = conv("decoder.model.0", dd, 7, D);
    c->blocks.clear();
    for (int bi = 0; bi < cf.n_decoder_rates; ++bi) {
        const int64_t cin = dd >> bi, cout = dd >> (bi + 1), s = cf.decoder_rates[bi], k = 2 * s, pad = (s + 1) / 2;
        const std::string p = "decoder.model." + std::to_string(bi + 1) + ".block";
        mis_dac::Blk B{};
        B.s = (int)s; B.pad = (int)pad; B.cin = (int)cin; B.cout = (int)cout;
        snake(p + ".0.alpha", cin, B.a, B.ra);
        {   // per output phase ph: o = s*n + ph takes taps kk = (ph + pad) % s + s*j from x[n + (ph + pad)/s - j]
            std::vector<float> w = wn(p + ".1", cout, k, cin, 2), at((size_t)s * 2 * cin * cout);
            for (int64_t ph = 0; ph < s; ++ph) for (int64_t j = 0; j < 2; ++j) for (int64_t i = 0; i < cin; ++i) for (int64_t o = 0; o < cout; ++o)
                at[((ph * 2 + j) * cin + i) * cout + o] = w[(o * k + ((ph + pad) % s + s * j)) * cin + i];
            B.ct.M = (int)cout; B.ct.K = (int)(2 * cin); B.ct.w = push(at); B.ct.b = push(dneed(c, p + ".1.bias", {cout}));
        }
        const int dils[3] = {1, 3, 9};
        for (int ri = 0; ri < 3; ++ri) {
            const std::string q = p + "." + std::to_string(ri + 2) + ".block";
            snake(q + ".0.alpha", cout, B.ru[ri].a1, B.ru[ri].ra1);
            B.ru[ri].c1 = conv(q + ".1", cout, 7, cout);
            snake(q + ".2.alpha", cout, B.ru[ri].a2, B.ru[ri].ra2);
            B.ru[ri].c2 = conv(q + ".3", cout, 1, cout);
            B.ru[ri].dil = dils[ri];
        }
        c->blocks.push_back(B);
    }
    {
        const int n = cf.n_decoder_rates;
        const int64_t cl = dd >> n;
        c->fin_c = (int)cl;
        snake("decoder.model." + std::to_string(n + 1) + ".alpha", cl, c->fin_a, c->fin_ra);
        std::vector<float> w = wn("decoder.model." + std::to_string(n + 2), 1, 7, cl, 0);
        c->fin_w = push(w);                                              // [1][7][C] == [7][C]
        c->fin_b = dneed(c, "decoder.model." + std::to_string(n + 2) + ".bias", {1})[0];
    }
    // ---- encoder (optional): dimensions are read off the tensors
    c->has_encoder = false;
    {
        auto e0 = c->raw_shape.find("encoder.block.0.weight_v");
        if (e0 != c->raw_shape.end()) {
            MIS_REQUIRE(e0->second.size() == 3 && e0->second[1] == 7 && e0->second[2] == 1, MIS_ERR_INVALID_INPUT, "bad DAC encoder stem");
            int64_t ch = e0->second[0];
            c->enc_dim = (int)ch;
            {
                std::vector<float> w = wn("encoder.block.0", ch, 7, 1, 0);          // [C][7][1] == [C][7]
                c->enc_first_w = push(w); c->enc_first_b = push(dneed(c, "encoder.block.0.bias", {ch}));
            }
            c->enc_blocks.clear();
            c->hop = 1;
            int li = 1;
            const int dils[3] = {1, 3, 9};
            for (;; ++li) {
                const std::string p = "encoder.block." + std::to_string(li) + ".block";
                auto dn = c->raw_shape.find(p + ".4.weight_v");
                if (dn == c->raw_shape.end()) break;
                MIS_REQUIRE(dn->second.size() == 3 && dn->second[2] == ch && dn->second[1] % 2 == 0, MIS_ERR_INVALID_INPUT, "bad DAC encoder block %d", li);
                mis_dac::EncBlk E{};
                E.cin = (int)ch; E.cout = (int)dn->second[0]; E.stride = (int)dn->second[1] / 2;
                for (int ri = 0; ri < 3; ++ri) {
                    const std::string q = p + "." + std::to_string(ri) + ".block";
                    snake(q + ".0.alpha", ch, E.ru[ri].a1, E.ru[ri].ra1);
                    E.ru[ri].c1 = conv(q + ".1", ch, 7, ch);
                    snake(q + ".2.alpha", ch, E.ru[ri].a2, E.ru[ri].ra2);
                    E.ru[ri].c2 = conv(q + ".3", ch, 1, ch);
                    E.ru[ri].dil = dils[ri];
                }
                snake(p + ".3.alpha", ch, E.a, E.ra);
                {   // WNConv1d(k = 2s, stride s, pad ceil(s/2)) over the phase-split input [ch*s][T/s]:
                    // y[n] = sum_{q in -1..1} sum_{c,r} W[co][s*q + r + pad][c] * xph[c*s + r][n + q]
                    const int sdn = E.stride, K = 2 * sdn, pad = (sdn + 1) / 2;
                    std::vector<float> w = wn(p + ".4", E.cout, K, ch, 0);
                    std::vector<float> at((size_t)3 * sdn * ch * E.cout, 0.0f);
                    for (int qi = 0; qi < 3; ++qi)
                        for (int64_t ci = 0; ci < ch; ++ci)
                            for (int r = 0; r < sdn; ++r) {
                                const int j = sdn * (qi - 1) + r + pad;
                                if (j < 0 || j >= K) continue;
                                for (int co = 0; co < E.cout; ++co)
                                    at[(((size_t)qi * ch * sdn) + (size_t)ci * sdn + r) * E.cout + co] = w[((size_t)co * K + j) * ch + ci];
                            }
                    E.down.M = E.cout; E.down.K = 3 * sdn * (int)ch; E.down.w = push(at); E.down.b = push(dneed(c, p + ".4.bias", {E.cout}));
                }
                c->hop *= E.stride;
                ch = E.cout;
                c->enc_blocks.push_back(E);
            }
            MIS_REQUIRE(!c->enc_blocks.empty(), MIS_ERR_INVALID_INPUT, "DAC encoder has no blocks");
            snake("encoder.block." + std::to_string(li) + ".alpha", ch, c->enc_fin_a, c->enc_fin_ra);
            c->enc_last = conv("encoder.block." + std::to_string(li + 1), D, 3, ch);
            c->vq_enc.clear();
            for (int q = 0; q < cf.n_codebooks; ++q) {
                const std::string p = "quantizer.quantizers." + std::to_string(q);
                mis_dac::VqEnc v{};
                v.in_proj = conv(p + ".inProj", cd, 1, D);
                const auto& cb = dneed(c, p + ".codebook.weight", {bins, cd});
                std::vector<float> cn((size_t)bins * cd), cn2(bins);
                for (int64_t k = 0; k < bins; ++k) {                         // descriptNormalize (DescriptQuantization.swift:8-11)
                    float n2 = 0.0f;
                    for (int64_t d2 = 0; d2 < cd; ++d2) n2 += cb[k * cd + d2] * cb[k * cd + d2];
                    const float inv = 1.0f / std::max(sqrtf(n2), 1e-12f);
                    float s2 = 0.0f;
                    for (int64_t d2 = 0; d2 < cd; ++d2) { const float x = cb[k * cd + d2] * inv; cn[k * cd + d2] = x; s2 += x * x; }
                    cn2[k] = s2;
                }
                v.cn = push(cn); v.cn2 = push(cn2);
                c->vq_enc.push_back(v);
            }
            c->has_encoder = true;
        }
    }
    c->arena.alloc(arena.size());
    HIP_CHECK(hipMemcpy(c->arena.p, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    c->raw.clear(); c->raw_shape.clear();
    c->finalized = true;
    MIS_API_END
}

// ---- encode: preprocess (right-pad to the hop length, :216-228) + encode (:230-233)
extern "C" int64_t mis_dac_padded_length(const mis_dac* c, int64_t n_samples) {
    if (!c || !c->has_encoder || n_samples < 0) return 0;
    return (n_samples + c->hop - 1) / c->hop * c->hop;
}
// audio f32 [batch, n_samples] (host or device) -> codes int32 [batch, nq, T], T = padded_length / hop (host or device), nq = n_quantizers
// (0 = all codebooks; the reference's nQuantizers, DescriptQuantization.swift:120-147);
// z_out (nullable) f32 [batch, latent, T]: the encoder output before quantisation
extern "C" mis_status mis_dac_encode(mis_dac* c, const float* audio, int batch, int64_t n_samples, int n_quantizers, int32_t* codes_out, float* z_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && audio && codes_out && batch >= 1 && n_samples >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "DAC model not finalized");
    MIS_REQUIRE(c->has_encoder, MIS_ERR_AUDIO_ENCODE, "this DAC handle was loaded without encoder weights");
    MIS_REQUIRE(n_quantizers >= 0 && n_quantizers <= c->cfg.n_codebooks, MIS_ERR_INVALID_INPUT, "n_quantizers out of range");
    const int nq = n_quantizers ? n_quantizers : c->cfg.n_codebooks;
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const mis_dac_config& cf = c->cfg;
    const float* W = c->arena.p;
    const int64_t Tp = mis_dac_padded_length(c, n_samples);
    const int D = c->latent;
    size_t cap = (size_t)c->enc_dim * Tp;
    {
        int64_t T = Tp;
        for (auto& b : c->enc_blocks) { cap = std::max(cap, (size_t)b.cin * T); T /= b.stride; cap = std::max(cap, (size_t)b.cout * T); }
        cap = std::max(cap, (size_t)D * T);
    }
    cap *= (size_t)batch;
    for (int i = 0; i < 3; ++i) c->ebuf[i].alloc(cap);
    DevBuf<float> ain;
    ain.alloc((size_t)batch * Tp);
    HIP_CHECK(hipMemsetAsync(ain.p, 0, (size_t)batch * Tp * 4, s));
    HIP_CHECK(hipMemcpy2DAsync(ain.p, (size_t)Tp * 4, audio, (size_t)n_samples * 4, (size_t)n_samples * 4, batch, hipMemcpyDefault, s));
    float *x = c->ebuf[0].p, *f1 = c->ebuf[1].p, *f2 = c->ebuf[2].p;
    int64_t T = Tp;
    launch_enc_first(ain.p, x, W + c->enc_first_w, W + c->enc_first_b, batch, c->enc_dim, (int)T, s);
    auto gp = [&](const mis_dac::Lin& L, const float* X, float* Y, int N) {
        GemmParams g{};
        g.AT = W + L.w; g.bias = W + L.b; g.X = X; g.Y = Y; g.M = L.M; g.K = L.K; g.N = N; g.Tin = N; g.Tout = N;
        return g;
    };
    for (auto& E : c->enc_blocks) {
        for (int ri = 0; ri < 3; ++ri) {
            const auto& R = E.ru[ri];
            GemmParams g1 = gp(R.c1, x, f1, (int)T);
            g1.Cin = E.cin; g1.taps = 7; g1.dil = R.dil; g1.pad = 3 * R.dil; g1.alpha = W + R.a1; g1.ralpha = W + R.ra1;
            launch_gemm(GEMM_TAPS, true, g1, batch, s);
            GemmParams g2 = gp(R.c2, f1, f2, (int)T);
            g2.R = x; g2.alpha = W + R.a2; g2.ralpha = W + R.ra2;
            launch_gemm(GEMM_RESID, true, g2, batch, s);
            std::swap(x, f2);
        }
        MIS_REQUIRE(T % E.stride == 0, MIS_ERR_AUDIO_ENCODE, "internal: length not divisible by the stride");
        launch_enc_phase_split(x, f1, W + E.a, W + E.ra, batch, E.cin, (int)T, E.stride, s);
        T /= E.stride;
        GemmParams g = gp(E.down, f1, f2, (int)T);
        g.Cin = E.stride * E.cin; g.taps = 3; g.dil = 1; g.pad = 1;
        launch_gemm(GEMM_TAPS, false, g, batch, s);
        std::swap(x, f2);
    }
    {
        GemmParams g = gp(c->enc_last, x, f1, (int)T);
        g.Cin = c->enc_blocks.back().cout; g.taps = 3; g.dil = 1; g.pad = 1; g.alpha = W + c->enc_fin_a; g.ralpha = W + c->enc_fin_ra;
        launch_gemm(GEMM_TAPS, true, g, batch, s);
    }
    float* resid = f1;                                                    // z [B][D][T]
    if (z_out) HIP_CHECK(hipMemcpyAsync(z_out, resid, (size_t)batch * D * T * 4, hipMemcpyDefault, s));
    c->vq_ze.alloc((size_t)batch * cf.codebook_dim * T);
    c->enc_codes.alloc((size_t)batch * T);
    for (int q = 0; q < nq; ++q) {
        launch_gemm(GEMM_PLAIN, false, gp(c->vq_enc[q].in_proj, resid, c->vq_ze.p, (int)T), batch, s);
        launch_vq_nearest(c->vq_ze.p, W + c->vq_enc[q].cn, W + c->vq_enc[q].cn2, c->enc_codes.p, batch, cf.codebook_dim, cf.codebook_size, (int)T, s);
        HIP_CHECK(hipMemcpy2DAsync(codes_out + (size_t)q * T, (size_t)nq * T * 4, c->enc_codes.p, (size_t)T * 4, (size_t)T * 4, batch,
                                   hipMemcpyDefault, s));
        if (q + 1 < nq)                                                    // residual -= out_proj(codebook[code]) (+ bias): the decode table row
            launch_vq_residual(resid, c->enc_codes.p, W + c->tables + (size_t)q * cf.codebook_size * D, batch, D, (int)T, 1, s);
        HIP_CHECK(hipStreamSynchronize(s));                                // enc_codes is reused by the next quantizer
    }
    HIP_CHECK(hipGetLastError());
    MIS_API_END
}

// stage 0: waveform; 1 + i: output of decoder block i (tap)
static const float* dac_run(mis_dac* c, const int32_t* codes_dev, int batch, int T, float* wav_dev, int64_t wav_stride, int stage, int* outC,
                            int64_t* outT) {
    CodecPackScope pack_scope(&c->pack);
    MIS_REQUIRE(c->finalized, MIS_ERR_NOT_INITIALIZED, "DAC model not finalized");
    const mis_dac_config& cf = c->cfg;
    hipStream_t s = c->stream;
    const float* W = c->arena.p;
    size_t need_elems = (size_t)std::max(c->latent, cf.decoder_dim) * T;
    {
        int64_t Tc = T;
        for (auto& B : c->blocks) { need_elems = std::max(need_elems, (size_t)B.cin * Tc); Tc = (Tc - 1) * B.s - 2 * B.pad + 2 * B.s + 1; need_elems = std::max(need_elems, (size_t)B.cout * Tc); }
    }
    for (int i = 0; i < 3; ++i) c->buf[i].alloc((size_t)batch * need_elems);
    float *x = c->buf[0].p, *y = c->buf[1].p, *t1 = c->buf[2].p;
    auto gp = [&](const mis_dac::Lin& L, const float* X, float* Y, int N, int Tin, int Tout) {
        GemmParams g{};
        g.AT = W + L.w; g.bias = W + L.b; g.X = X; g.Y = Y; g.M = L.M; g.K = L.K; g.N = N; g.Tin = Tin; g.Tout = Tout;
        return g;
    };
    hipLaunchKernelGGL(k_dac_embed, dim3(T, batch), dim3(256), 0, s, codes_dev, W + c->tables, x, cf.n_codebooks, cf.codebook_size, c->latent, T);
    {
        GemmParams g = gp(c->first, x, y, T, T, T);
        g.Cin = c->latent; g.taps = 7; g.dil = 1; g.pad = 3;
        launch_gemm(GEMM_TAPS, false, g, batch, s);
        std::swap(x, y);
    }
    int64_t Tc = T;
    int bi = 0;
    for (auto& B : c->blocks) {
        const int64_t To = (Tc - 1) * B.s - 2 * B.pad + 2 * B.s + 1;
        GemmParams g = gp(B.ct, x, y, (int)((To + B.s - 1) / B.s), (int)Tc, (int)To);
        g.s = B.s; g.pad = B.pad; g.Cin = B.cin; g.alpha = W + B.a; g.ralpha = W + B.ra;
        launch_gemm(GEMM_CONVT, true, g, batch, s);
        Tc = To;
        std::swap(x, y);
        for (int ri = 0; ri < 3; ++ri) {
            const auto& R = B.ru[ri];
            GemmParams g1 = gp(R.c1, x, t1, (int)Tc, (int)Tc, (int)Tc);
            g1.Cin = B.cout; g1.taps = 7; g1.dil = R.dil; g1.pad = 3 * R.dil; g1.alpha = W + R.a1; g1.ralpha = W + R.ra1;
            launch_gemm(GEMM_TAPS, true, g1, batch, s);
            GemmParams g2 = gp(R.c2, t1, y, (int)Tc, (int)Tc, (int)Tc);
            g2.R = x; g2.alpha = W + R.a2; g2.ralpha = W + R.ra2;
            launch_gemm(GEMM_RESID, true, g2, batch, s);
            std::swap(x, y);
        }
        if (stage == 1 + bi) { *outC = B.cout; *outT = Tc; return x; }
        ++bi;
    }
    hipLaunchKernelGGL(k_dac_final, dim3(cdiv(Tc, 128), batch), dim3(128), 0, s, x, wav_dev, wav_stride, W + c->fin_w, c->fin_b, W + c->fin_a,
                       W + c->fin_ra, c->fin_c, (int)Tc);
    HIP_CHECK(hipGetLastError());
    *outC = 1; *outT = Tc;
    return wav_dev;
}

// DescriptDAC.decodeFromCodes (:239-242): codes int32 [batch, n_codebooks, T] (host or device) -> wav f32 [batch, num_samples(T)]
extern "C" mis_status mis_dac_decode_codes(mis_dac* c, const int32_t* codes, int batch, int T, float* wav_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && codes && wav_out && batch >= 1 && T >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    HIP_CHECK(hipSetDevice(c->device));
    const int64_t n = mis_dac_num_samples(c, T);
    c->codes_dev.alloc((size_t)batch * c->cfg.n_codebooks * T);
    HIP_CHECK(hipMemcpyAsync(c->codes_dev.p, codes, (size_t)batch * c->cfg.n_codebooks * T * 4, hipMemcpyDefault, c->stream));
    DevBuf<float> wav;
    wav.alloc((size_t)batch * n);
    int C; int64_t Tt;
    dac_run(c, c->codes_dev.p, batch, T, wav.p, n, 0, &C, &Tt);
    HIP_CHECK(hipMemcpyAsync(wav_out, wav.p, (size_t)batch * n * 4, hipMemcpyDefault, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    MIS_API_END
}
extern "C" mis_status mis_dac_debug_tap(mis_dac* c, const int32_t* codes, int batch, int T, int block, float* out, int64_t capacity,
                                        int32_t* channels, int64_t* length) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && codes && out && channels && length && block >= 0 && block < c->cfg.n_decoder_rates, MIS_ERR_INVALID_INPUT, "bad argument");
    HIP_CHECK(hipSetDevice(c->device));
    c->codes_dev.alloc((size_t)batch * c->cfg.n_codebooks * T);
    HIP_CHECK(hipMemcpyAsync(c->codes_dev.p, codes, (size_t)batch * c->cfg.n_codebooks * T * 4, hipMemcpyDefault, c->stream));
    DevBuf<float> wav;
    wav.alloc((size_t)batch * mis_dac_num_samples(c, T));
    int C = 0; int64_t Tt = 0;
    const float* res = dac_run(c, c->codes_dev.p, batch, T, wav.p, mis_dac_num_samples(c, T), 1 + block, &C, &Tt);
    MIS_REQUIRE((int64_t)batch * C * Tt <= capacity, MIS_ERR_INVALID_INPUT, "tap buffer too small");
    HIP_CHECK(hipMemcpyAsync(out, res, (size_t)batch * C * Tt * 4, hipMemcpyDefault, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    *channels = C; *length = Tt;
    MIS_API_END
}
