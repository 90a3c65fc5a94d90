// common.h - shared helpers for libmi_speech (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string>
#include <vector>
#include <map>
#include <stdexcept>

#include "../../include/mi_speech.h"
#include "../../include/mi_speech_debug.h"

// ---------------------------------------------------------------------------- errors
struct MisError : std::runtime_error {
    mis_status code;
    MisError(mis_status c, const std::string& m) : std::runtime_error(m), code(c) {}
};
void mis_set_error(const char* fmt, ...);
mis_status mis_fail(mis_status code, const char* fmt, ...);

#define HIP_CHECK(expr)                                                                      \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            char _b[512];                                                                    \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                     __FILE__, __LINE__);                                                    \
            throw MisError(MIS_ERR_DEVICE, _b);                                              \
        }                                                                                    \
    } while (0)

#define MIS_REQUIRE(cond, code, ...)                                                         \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            char _b[512];                                                                    \
            snprintf(_b, sizeof(_b), __VA_ARGS__);                                           \
            throw MisError(code, _b);                                                        \
        }                                                                                    \
    } while (0)

// Wrap an extern "C" body: exceptions -> status + mis_last_error().
#define MIS_API_BEGIN try {
#define MIS_API_END                                                                          \
    }                                                                                        \
    catch (const MisError& e) { mis_set_error("%s", e.what()); return e.code; }              \
    catch (const std::exception& e) { mis_set_error("%s", e.what()); return MIS_ERR_GENERATION_FAILED; } \
    return MIS_OK;

// ---------------------------------------------------------------------------- device memory
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        if (count <= n && p) return;
        release();
        HIP_CHECK(hipMalloc((void**)&p, (count ? count : 1) * sizeof(T)));
        n = count;
    }
    void zero(hipStream_t s) { if (p && n) HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
    size_t bytes() const { return n * sizeof(T); }
};

// pinned host memory with scope lifetime; release() hands the pointer to the caller (freed later with mis_free)
template <typename T>
struct PinnedBuf {
    T* p = nullptr;
    PinnedBuf() = default;
    explicit PinnedBuf(size_t n) { alloc(n); }
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    void alloc(size_t n) {
        if (p) { (void)hipHostFree(p); p = nullptr; }
        HIP_CHECK(hipHostMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T), 0));
    }
    T* release() { T* q = p; p = nullptr; return q; }
};

static inline size_t round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------- bf16 helpers
typedef uint16_t bf16_t;   // raw bits

__host__ __device__ static inline float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}
__host__ __device__ static inline bf16_t f32_to_bf16(float f) {   // round-to-nearest-even
#if defined(__HIP_DEVICE_COMPILE__)
    // gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, RNE, quiet NaN): one instruction instead of the six of the
    // integer formulation below - the step chain's kernels round at every MLX primitive boundary, so this is a large share
    // of their VALU instruction count
    return __builtin_bit_cast(unsigned short, (__bf16)f);
#endif
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__host__ __device__ static inline float bf16_round_f32(float f) { return bf16_to_f32(f32_to_bf16(f)); }

// host-side fp16 -> f32
static inline float f16_to_f32_host(uint16_t h) {
    uint32_t sign = (h >> 15) & 1, exp = (h >> 10) & 0x1f, man = h & 0x3ff;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) u = sign << 31;
        else {
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400));
            man &= 0x3ff;
            u = (sign << 31) | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) u = (sign << 31) | 0x7f800000u | (man << 13);
    else u = (sign << 31) | ((exp - 15 + 127) << 23) | (man << 13);
    union { uint32_t u; float f; } c;
    c.u = u;
    return c.f;
}

// mis-synth-v1 (oracle/synth.py): uniform in (0,1) from (key, index)
__host__ __device__ static inline uint64_t mis_splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ static inline float mis_synth_value(uint64_t key, uint64_t idx, float amp) {
    uint64_t u = mis_splitmix64(key * 0x9E3779B97F4A7C15ull + idx);
    float x = ((float)(uint32_t)(u >> 40) + 0.5f) * 5.9604644775390625e-08f;   // 2^-24
    return (2.0f * x - 1.0f) * amp;
}

// ---------------------------------------------------------------------------- wave helpers (wave64)
__device__ static inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// the same sum on the DPP network instead of six dependent ds_bpermute round trips (about 60 cycles each): two quad
// permutes, half-row and row mirror give every lane its 16-lane row total, the four row totals are read as scalars.
// (Other summation order than wave_sum: use one or the other consistently where bit-reproducibility across code paths matters.)
__device__ static inline float wave_sum_dpp(float v) {
    auto dpp_add = [](float x, int ctrl_tag) {
        int xi = __builtin_bit_cast(int, x), yi;
        switch (ctrl_tag) {
            case 0: yi = __builtin_amdgcn_update_dpp(0, xi, 0xB1, 0xF, 0xF, true); break;    // quad_perm [1,0,3,2]
            case 1: yi = __builtin_amdgcn_update_dpp(0, xi, 0x4E, 0xF, 0xF, true); break;    // quad_perm [2,3,0,1]
            case 2: yi = __builtin_amdgcn_update_dpp(0, xi, 0x141, 0xF, 0xF, true); break;   // row_half_mirror
            default: yi = __builtin_amdgcn_update_dpp(0, xi, 0x140, 0xF, 0xF, true); break;  // row_mirror
        }
        return x + __builtin_bit_cast(float, yi);
    };
    v = dpp_add(v, 0); v = dpp_add(v, 1); v = dpp_add(v, 2); v = dpp_add(v, 3);
    const int vi = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16)),
                r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ static inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------- tiny JSON + safetensors (host)
struct JsonValue {
    enum Type { NUL, BOOL, NUM, STR, ARR, OBJ } type = NUL;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj;
    const JsonValue* get(const std::string& k) const {
        for (auto& kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    double number_or(const std::string& k, double d) const {
        const JsonValue* v = get(k);
        return (v && v->type == NUM) ? v->num : d;
    }
    bool bool_or(const std::string& k, bool d) const {
        const JsonValue* v = get(k);
        return (v && v->type == BOOL) ? v->b : d;
    }
};
JsonValue json_parse(const std::string& text);
std::string read_text_file(const std::string& path);

struct SafeTensorEntry {
    std::string name, dtype;
    std::vector<int64_t> shape;
    const uint8_t* data;
    size_t nbytes;
};
struct SafeTensorFile {
    void* map = nullptr;
    size_t map_len = 0;
    std::vector<SafeTensorEntry> entries;
    ~SafeTensorFile();
    void open(const std::string& path);
};
std::vector<std::string> list_safetensors(const std::string& dir);
mis_dtype dtype_from_safetensors(const std::string& s);
