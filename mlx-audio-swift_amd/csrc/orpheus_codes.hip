// orpheus_codes.hip - Orpheus <-> SNAC token framing on the GPU (integer-exact).
// Replaces llamaDecodeAudioFromCodes' host loop (LlamaTTS.swift:41-64) and parseOutput's
// per-element .item() walk (LlamaTTS.swift:383-434).
#include "common.h"
#include "kernels.h"

#define ORPHEUS_START_OF_SPEECH 128257
#define ORPHEUS_END_OF_SPEECH 128258
#define ORPHEUS_AUDIO_OFFSET 128266

// codes7 [batch, in_stride] -> l0 [batch, groups], l1 [batch, 2*groups], l2 [batch, 4*groups]
// (output row strides are given so the caller can pad rows to a common length).
// Slots per frame: L0, L1, L2, L2, L1, L2, L2 with offsets k*4096  (LlamaTTS.swift:51-57).
__global__ void k_orpheus_deinterleave(const int32_t* __restrict__ codes7, int in_stride, int groups,
                                       int32_t* __restrict__ l0, int32_t* __restrict__ l1,
                                       int32_t* __restrict__ l2, int out_groups) {
    int b = blockIdx.y;
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= out_groups) return;
    int32_t* o0 = l0 + (size_t)b * out_groups;
    int32_t* o1 = l1 + (size_t)b * out_groups * 2;
    int32_t* o2 = l2 + (size_t)b * out_groups * 4;
    if (g >= groups) {   // padding rows of a ragged batch decode code 0 (results are discarded)
        o0[g] = 0; o1[2 * g] = 0; o1[2 * g + 1] = 0;
        o2[4 * g] = 0; o2[4 * g + 1] = 0; o2[4 * g + 2] = 0; o2[4 * g + 3] = 0;
        return;
    }
    const int32_t* c = codes7 + (size_t)b * in_stride + 7 * g;
    o0[g] = c[0];
    o1[2 * g] = c[1] - 4096;
    o2[4 * g] = c[2] - 2 * 4096;
    o2[4 * g + 1] = c[3] - 3 * 4096;
    o1[2 * g + 1] = c[4] - 4 * 4096;
    o2[4 * g + 2] = c[5] - 5 * 4096;
    o2[4 * g + 3] = c[6] - 6 * 4096;
}

// Ragged variant used by generate(): per-row group counts n_codes[b]/7, rows padded to out_groups.
__global__ void k_orpheus_deinterleave_ragged(const int32_t* __restrict__ codes7, int in_stride,
                                              const int32_t* __restrict__ n_codes,
                                              int32_t* __restrict__ l0, int32_t* __restrict__ l1,
                                              int32_t* __restrict__ l2, int out_groups) {
    int b = blockIdx.y;
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= out_groups) return;
    int groups = n_codes[b] / 7;
    int32_t* o0 = l0 + (size_t)b * out_groups;
    int32_t* o1 = l1 + (size_t)b * out_groups * 2;
    int32_t* o2 = l2 + (size_t)b * out_groups * 4;
    if (g >= groups) {
        o0[g] = 0; o1[2 * g] = 0; o1[2 * g + 1] = 0;
        o2[4 * g] = 0; o2[4 * g + 1] = 0; o2[4 * g + 2] = 0; o2[4 * g + 3] = 0;
        return;
    }
    const int32_t* c = codes7 + (size_t)b * in_stride + 7 * g;
    // clamp into the codebook so that a free-running LM emitting an out-of-slot token cannot index
    // outside the tables (the reference would trap in MLX's gather); valid streams are unaffected.
    auto cl = [](int v) { return v < 0 ? 0 : (v > 4095 ? 4095 : v); };
    o0[g] = cl(c[0]);
    o1[2 * g] = cl(c[1] - 4096);
    o2[4 * g] = cl(c[2] - 2 * 4096);
    o2[4 * g + 1] = cl(c[3] - 3 * 4096);
    o1[2 * g + 1] = cl(c[4] - 4 * 4096);
    o2[4 * g + 2] = cl(c[5] - 5 * 4096);
    o2[4 * g + 3] = cl(c[6] - 6 * 4096);
}

// One 256-thread block per row.  parseOutput (LlamaTTS.swift:383-434) per row:
//   last = last index of 128257 (or -1) ; keep ids[last+1:] != 128258 ; trim to 7k ; -128266.
// Token ids are parameters: Orpheus (128257 / 128258 / 128266, LlamaTTS.swift:20-30) or VyvoTTS (151670 / 151671 / 151679 and the
// START_OF_AI fallback of Qwen3.swift:332-358: without a start-of-speech token, cropping starts at the first audio token after
// the last START_OF_AI).
__global__ void __launch_bounds__(256) k_orpheus_parse_output(const int32_t* __restrict__ ids,
                                                              const int32_t* __restrict__ lens, int stride,
                                                              int32_t* __restrict__ codes_out,
                                                              int32_t* __restrict__ n_codes_out, int sos_id, int eos_id,
                                                              int audio_offset, int soa_id) {
    __shared__ int s_last;
    __shared__ int s_soa, s_first;
    __shared__ int s_scan[256];
    __shared__ int s_base;
    int b = blockIdx.x, tid = threadIdx.x;
    const int32_t* row = ids + (size_t)b * stride;
    int32_t* out = codes_out + (size_t)b * stride;
    int len = lens[b];
    if (len > stride) len = stride;
    if (tid == 0) { s_last = -1; s_base = 0; s_soa = -1; s_first = 0x7fffffff; }
    __syncthreads();
    int my_last = -1, my_soa = -1;
    for (int j = tid; j < len; j += 256) {
        if (row[j] == sos_id) my_last = j;
        if (soa_id >= 0 && row[j] == soa_id) my_soa = j;
    }
    if (my_last >= 0) atomicMax(&s_last, my_last);
    if (my_soa >= 0) atomicMax(&s_soa, my_soa);
    __syncthreads();
    if (s_last < 0 && s_soa >= 0) {                      // fallback: first audio token after the last START_OF_AI
        int my_first = 0x7fffffff;
        for (int j = s_soa + 1 + tid; j < len; j += 256)
            if (row[j] >= audio_offset) { my_first = j; break; }
        if (my_first != 0x7fffffff) atomicMin(&s_first, my_first);
        __syncthreads();
        if (tid == 0 && s_first != 0x7fffffff) s_last = s_first - 1;
        __syncthreads();
    }
    int start = s_last + 1;
    // ordered stream compaction, 256 elements per round
    for (int base = start; base < len; base += 256) {
        int j = base + tid;
        int v = (j < len) ? row[j] : eos_id;
        int keep = (j < len && v != eos_id) ? 1 : 0;
        s_scan[tid] = keep;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {   // Hillis-Steele inclusive scan
            int t = (tid >= o) ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += t;
            __syncthreads();
        }
        int pos = s_base + s_scan[tid] - keep;
        if (keep) out[pos] = v - audio_offset;
        __syncthreads();
        if (tid == 255) s_base += s_scan[255];
        __syncthreads();
    }
    if (tid == 0) n_codes_out[b] = (s_base / 7) * 7;
}

void launch_orpheus_deinterleave(const int32_t* codes7, int in_stride, int batch, int groups, int32_t* l0,
                                 int32_t* l1, int32_t* l2, int out_groups, hipStream_t s) {
    if (out_groups <= 0 || batch <= 0) return;
    dim3 grid(cdiv(out_groups, 128), batch);
    hipLaunchKernelGGL(k_orpheus_deinterleave, grid, dim3(128), 0, s, codes7, in_stride, groups, l0, l1, l2, out_groups);
}
void launch_orpheus_deinterleave_ragged(const int32_t* codes7, int in_stride, const int32_t* n_codes, int batch,
                                        int32_t* l0, int32_t* l1, int32_t* l2, int out_groups, hipStream_t s) {
    if (out_groups <= 0 || batch <= 0) return;
    dim3 grid(cdiv(out_groups, 128), batch);
    hipLaunchKernelGGL(k_orpheus_deinterleave_ragged, grid, dim3(128), 0, s, codes7, in_stride, n_codes, l0, l1, l2,
                       out_groups);
}
void launch_orpheus_parse_output(const int32_t* ids, const int32_t* lens, int batch, int stride, int32_t* codes_out,
                                 int32_t* n_codes_out, hipStream_t s, const SpeechTokenIds* tk) {
    if (batch <= 0) return;
    SpeechTokenIds t = tk ? *tk : SpeechTokenIds{ORPHEUS_START_OF_SPEECH, ORPHEUS_END_OF_SPEECH, ORPHEUS_AUDIO_OFFSET, -1};
    hipLaunchKernelGGL(k_orpheus_parse_output, dim3(batch), dim3(256), 0, s, ids, lens, stride, codes_out, n_codes_out, t.start_of_speech,
                       t.end_of_speech, t.audio_offset, t.start_of_ai);
}

// ---------------------------------------------------------------------------- C ABI
extern "C" mis_status mis_orpheus_deinterleave(int device, const int32_t* codes7, int batch, int groups,
                                               int32_t* l0, int32_t* l1, int32_t* l2) {
    MIS_API_BEGIN
    MIS_REQUIRE(batch >= 0 && groups >= 0, MIS_ERR_INVALID_INPUT, "negative batch/groups");
    MIS_REQUIRE(batch == 0 || groups == 0 || (codes7 && l0 && l1 && l2), MIS_ERR_INVALID_INPUT, "null pointer");
    if (batch == 0 || groups == 0) return MIS_OK;
    HIP_CHECK(hipSetDevice(device));
    size_t n = (size_t)batch * groups;
    DevBuf<int32_t> din, d0, d1, d2;
    din.alloc(n * 7); d0.alloc(n); d1.alloc(n * 2); d2.alloc(n * 4);
    HIP_CHECK(hipMemcpy(din.p, codes7, n * 7 * sizeof(int32_t), hipMemcpyDefault));
    launch_orpheus_deinterleave(din.p, 7 * groups, batch, groups, d0.p, d1.p, d2.p, groups, 0);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpy(l0, d0.p, n * sizeof(int32_t), hipMemcpyDefault));
    HIP_CHECK(hipMemcpy(l1, d1.p, n * 2 * sizeof(int32_t), hipMemcpyDefault));
    HIP_CHECK(hipMemcpy(l2, d2.p, n * 4 * sizeof(int32_t), hipMemcpyDefault));
    MIS_API_END
}

static mis_status parse_output_host(int device, const int32_t* ids, const int32_t* lens, int batch, int stride, int32_t* codes_out,
                                    int32_t* n_codes_out, const SpeechTokenIds* tk);
extern "C" mis_status mis_orpheus_parse_output(int device, const int32_t* ids, const int32_t* lens, int batch,
                                               int stride, int32_t* codes_out, int32_t* n_codes_out) {
    return parse_output_host(device, ids, lens, batch, stride, codes_out, n_codes_out, nullptr);
}
// same with explicit token ids (VyvoTTS: Qwen3.swift:19-29,332-358); start_of_ai < 0 disables the fallback
extern "C" mis_status mis_speech_parse_output(int device, const int32_t* ids, const int32_t* lens, int batch, int stride,
                                              int32_t* codes_out, int32_t* n_codes_out, int start_of_speech, int end_of_speech,
                                              int audio_token_offset, int start_of_ai) {
    SpeechTokenIds t{start_of_speech, end_of_speech, audio_token_offset, start_of_ai};
    return parse_output_host(device, ids, lens, batch, stride, codes_out, n_codes_out, &t);
}
static mis_status parse_output_host(int device, const int32_t* ids, const int32_t* lens, int batch, int stride, int32_t* codes_out,
                                    int32_t* n_codes_out, const SpeechTokenIds* tk) {
    MIS_API_BEGIN
    MIS_REQUIRE(batch >= 0 && stride >= 0, MIS_ERR_INVALID_INPUT, "negative batch/stride");
    if (batch == 0) return MIS_OK;
    MIS_REQUIRE(lens && n_codes_out && (stride == 0 || (ids && codes_out)), MIS_ERR_INVALID_INPUT, "null pointer");
    HIP_CHECK(hipSetDevice(device));
    size_t n = (size_t)batch * (stride ? stride : 1);
    DevBuf<int32_t> din, dl, dout, dn;
    din.alloc(n); dl.alloc(batch); dout.alloc(n); dn.alloc(batch);
    if (stride) HIP_CHECK(hipMemcpy(din.p, ids, (size_t)batch * stride * sizeof(int32_t), hipMemcpyDefault));
    HIP_CHECK(hipMemcpy(dl.p, lens, batch * sizeof(int32_t), hipMemcpyDefault));
    HIP_CHECK(hipMemset(dout.p, 0, n * sizeof(int32_t)));
    launch_orpheus_parse_output(din.p, dl.p, batch, stride, dout.p, dn.p, 0, tk);
    HIP_CHECK(hipGetLastError());
    if (stride) HIP_CHECK(hipMemcpy(codes_out, dout.p, (size_t)batch * stride * sizeof(int32_t), hipMemcpyDefault));
    HIP_CHECK(hipMemcpy(n_codes_out, dn.p, batch * sizeof(int32_t), hipMemcpyDefault));
    MIS_API_END
}
