// group.hip - utterance data parallelism BEHIND the C ABI: the rows of a batch are independent units, every
// GPU holds a full weight replica, rows are sharded in contiguous blocks, the RNG is keyed by the GLOBAL row (mis_gen_params.
// row_offset) so results do not depend on the shard count, and the only exchange is ONE all-gather of the decoded PCM at the end.
// The reference has no counterpart (single device); a Swift host cannot call torch.distributed, so both forms live here:
//   mis_group  single process, N replicas (one per device): one worker thread + stream per GPU inside mis_tts_group_generate*,
//              all-gather by direct peer writes over xGMI (hipMemcpyPeerAsync: every rank writes its block into every peer's
//              buffer - xGMI is point-to-point, 7 links per GPU, so the direct form uses all links at once where a ring all-gather is
//              bound by one);
//   mis_comm   one process per GPU (the bench contract: torchrun / any launcher): RCCL communicator created from a broadcast
//              ncclUniqueId, ncclAllGather of the PCM blocks and lengths on the library's stream.  librccl is dlopen'ed so that the
//              library still loads on hosts without RCCL.
#include "common.h"
#include "kernels.h"

#include <dlfcn.h>
#include <string.h>
#include <chrono>
#include <mutex>
#include <thread>

// ---------------------------------------------------------------------------- RCCL (dlopen)
namespace {
typedef struct { char internal[128]; } rccl_unique_id;            // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm_t;
enum { RCCL_INT64 = 4, RCCL_FLOAT32 = 7 };                        // ncclDataType_t values used here (rccl.h)
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(rccl_unique_id*) = nullptr;
    int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi load_rccl() {
    RcclApi api;
    std::vector<std::string> names;
    if (const char* p = getenv("MIS_RCCL_PATH")) names.push_back(p);
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    void* h = nullptr;
    for (auto& n : names) if ((h = dlopen(n.c_str(), RTLD_NOW | RTLD_NOLOAD))) break;      // a copy the host already loaded (torch bundles one)
    if (!h) for (auto& n : names) if ((h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return api;
    api.GetUniqueId = (int (*)(rccl_unique_id*))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(rccl_comm_t*, int, rccl_unique_id, int))dlsym(h, "ncclCommInitRank");
    api.AllGather = (int (*)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t))dlsym(h, "ncclAllGather");
    api.CommDestroy = (int (*)(rccl_comm_t))dlsym(h, "ncclCommDestroy");
    api.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (api.GetUniqueId && api.CommInitRank && api.AllGather && api.CommDestroy) api.handle = h;
    return api;
}
RcclApi* rccl() {
    static RcclApi api = load_rccl();       // function-local static: initialised exactly once, concurrent callers wait for it
    return api.handle ? &api : nullptr;
}
#define RCCL_CHECK(expr)                                                                                     \
    do {                                                                                                     \
        int _r = (expr);                                                                                     \
        if (_r != 0) {                                                                                       \
            char _b[256];                                                                                    \
            snprintf(_b, sizeof(_b), "%s failed: %s", #expr, rccl()->GetErrorString ? rccl()->GetErrorString(_r) : "rccl error"); \
            throw MisError(MIS_ERR_DEVICE, _b);                                                              \
        }                                                                                                    \
    } while (0)
}   // namespace

struct mis_comm {
    int device = 0, rank = 0, world = 1;
    rccl_comm_t comm = nullptr;
    hipStream_t stream = nullptr;
    DevBuf<int64_t> lens_local, lens_all;
    double last_gather_ms = 0;
};

extern "C" mis_status mis_comm_unique_id(void* id_out) {
    MIS_API_BEGIN
    MIS_REQUIRE(id_out, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(rccl(), MIS_ERR_DEVICE, "librccl could not be loaded (set MIS_RCCL_PATH)");
    rccl_unique_id id;
    RCCL_CHECK(rccl()->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    MIS_API_END
}

extern "C" mis_status mis_comm_create(int device, int rank, int world, const void* unique_id, mis_comm** out) {
    MIS_API_BEGIN
    MIS_REQUIRE(out && unique_id && world >= 1 && rank >= 0 && rank < world, MIS_ERR_INVALID_INPUT, "bad communicator arguments");
    MIS_REQUIRE(rccl(), MIS_ERR_DEVICE, "librccl could not be loaded (set MIS_RCCL_PATH)");
    int n = 0;
    HIP_CHECK(hipGetDeviceCount(&n));
    MIS_REQUIRE(device >= 0 && device < n, MIS_ERR_DEVICE, "device %d not available (%d GPUs visible)", device, n);
    HIP_CHECK(hipSetDevice(device));
    mis_comm* c = new mis_comm();
    c->device = device; c->rank = rank; c->world = world;
    try {
        rccl_unique_id id;
        memcpy(&id, unique_id, sizeof(id));
        RCCL_CHECK(rccl()->CommInitRank(&c->comm, world, id, rank));
        HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    } catch (...) { delete c; throw; }
    *out = c;
    MIS_API_END
}

extern "C" void mis_comm_destroy(mis_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
    delete c;
}

// The exchange step of SURVEY 8(e): every rank contributes rows_local rows of `stride` samples (device memory, the output of
// mis_tts_generate_device) and their lengths; every rank receives all world * rows_local rows in rank order.  Fixed-size blocks
// (weak scaling: the same number of rows per GPU).  Runs on the communicator's stream and returns when the data has landed.
extern "C" mis_status mis_comm_all_gather_pcm(mis_comm* c, const float* pcm_local_dev, const int64_t* lens_local, int rows_local,
                                              int64_t stride, float* pcm_all_dev, int64_t* lens_all, double* gather_ms) {
    MIS_API_BEGIN
    MIS_REQUIRE(c && pcm_local_dev && lens_local && pcm_all_dev && lens_all && rows_local >= 1 && stride >= 1, MIS_ERR_INVALID_INPUT, "bad argument");
    HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    c->lens_local.alloc(rows_local); c->lens_all.alloc((size_t)rows_local * c->world);
    HIP_CHECK(hipMemcpyAsync(c->lens_local.p, lens_local, (size_t)rows_local * 8, hipMemcpyDefault, s));
    struct EventPair {                             // released on every exit path (RCCL_CHECK / HIP_CHECK throw)
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    HIP_CHECK(hipEventCreate(&ev.a)); HIP_CHECK(hipEventCreate(&ev.b));
    hipEvent_t e0 = ev.a, e1 = ev.b;
    HIP_CHECK(hipEventRecord(e0, s));
    RCCL_CHECK(rccl()->AllGather(pcm_local_dev, pcm_all_dev, (size_t)rows_local * stride, RCCL_FLOAT32, c->comm, s));
    RCCL_CHECK(rccl()->AllGather(c->lens_local.p, c->lens_all.p, (size_t)rows_local, RCCL_INT64, c->comm, s));
    HIP_CHECK(hipEventRecord(e1, s));
    HIP_CHECK(hipMemcpyAsync(lens_all, c->lens_all.p, (size_t)rows_local * c->world * 8, hipMemcpyDefault, s));
    HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    c->last_gather_ms = ms;
    if (gather_ms) *gather_ms = ms;
    MIS_API_END
}

// ---------------------------------------------------------------------------- single-process device group
struct mis_group {
    std::vector<mis_tts*> reps;
    std::vector<mis_tts*> marked_shared;     // replicas this group flagged as sharing their device (cleared again on destroy)
    mis_group_timing timing{};
};

extern "C" mis_status mis_tts_group_create(mis_tts* const* replicas, int n, mis_group** out) {
    MIS_API_BEGIN
    MIS_REQUIRE(replicas && out && n >= 1 && n <= 64, MIS_ERR_INVALID_INPUT, "a group needs 1..64 replicas");
    mis_group* g = new mis_group();
    for (int i = 0; i < n; ++i) {
        if (!replicas[i]) { delete g; throw MisError(MIS_ERR_INVALID_INPUT, "null replica handle"); }
        for (int j = 0; j < i; ++j)
            if (replicas[j] == replicas[i]) { delete g; throw MisError(MIS_ERR_INVALID_INPUT, "a handle may appear only once in a group (one in-flight call per handle)"); }
        g->reps.push_back(replicas[i]);
    }
    // direct peer access between every pair of the group's devices: without it hipMemcpyPeerAsync stages the all-gather of
    // mis_tts_group_generate_device through host memory instead of writing over xGMI.  (Already-enabled is not an error; a pair the
    // platform cannot map keeps the staged path.)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const int di = tts_device(g->reps[i]), dj = tts_device(g->reps[j]);
            if (di == dj) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, di, dj) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
            if (hipSetDevice(di) != hipSuccess) { (void)hipGetLastError(); continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(dj, 0);
            if (e != hipSuccess) (void)hipGetLastError();                   // hipErrorPeerAccessAlreadyEnabled included
        }
    // replicas that share a device (logical shards on one GPU) run their streams side by side: none of them may launch a kernel whose
    // blocks wait for each other to be co-resident (the one-launch sampler would starve the other replica's rows and time out)
    for (int i = 0; i < n; ++i) {
        bool shared = false;
        for (int j = 0; j < n; ++j) shared = shared || (j != i && tts_device(g->reps[j]) == tts_device(g->reps[i]));
        if (shared) { tts_internal_set_shared_device(g->reps[i], true); g->marked_shared.push_back(g->reps[i]); }
    }
    *out = g;
    MIS_API_END
}
extern "C" void mis_tts_group_destroy(mis_group* g) {                 // the replicas stay with their owner
    if (g) for (mis_tts* r : g->marked_shared) tts_internal_set_shared_device(r, false);
    delete g;
}
extern "C" int mis_tts_group_size(const mis_group* g) { return g ? (int)g->reps.size() : 0; }

static void shard_block(int n_rows, int r, int world, int* lo, int* hi) {
    const int base = n_rows / world, rem = n_rows % world;
    *lo = r * base + std::min(r, rem);
    *hi = *lo + base + (r < rem ? 1 : 0);
}
extern "C" void mis_shard_rows(int n_rows, int rank, int world, int* lo, int* hi) {
    int a = 0, b = 0;
    if (world >= 1 && rank >= 0 && rank < world && n_rows >= 0) shard_block(n_rows, rank, world, &a, &b);
    if (lo) *lo = a;
    if (hi) *hi = b;
}

namespace {
struct ShardResult { mis_status st = MIS_OK; std::string err; double ms = 0; };
struct HostPrompts { std::vector<int32_t> lens, flat; std::vector<size_t> off; };
HostPrompts fetch_prompts(const int32_t* prompt_ids, const int32_t* prompt_lens, int batch) {
    HostPrompts h;
    h.lens.resize(batch);
    HIP_CHECK(hipMemcpy(h.lens.data(), prompt_lens, (size_t)batch * 4, hipMemcpyDefault));
    h.off.assign(batch + 1, 0);
    for (int b = 0; b < batch; ++b) { MIS_REQUIRE(h.lens[b] >= 1, MIS_ERR_INVALID_INPUT, "empty prompt in row %d", b); h.off[b + 1] = h.off[b] + h.lens[b]; }
    h.flat.resize(h.off[batch]);
    HIP_CHECK(hipMemcpy(h.flat.data(), prompt_ids, h.off[batch] * 4, hipMemcpyDefault));
    return h;
}
}   // namespace

// generate for a batch sharded over the group's replicas, PCM left in HBM and ALL-GATHERED: pcm_dev[i] is replica i's buffer
// [batch, pcm_stride] on its own device; on return every buffer holds every row.  One worker thread per replica runs
// mis_tts_generate_device on that replica's rows with row_offset advanced by the block start (RNG keyed by the global row), then
// writes its block into every peer's buffer (peer copies).  snac noise: drawn on the device (explicit noise is a single-device
// test facility).
extern "C" mis_status mis_tts_group_generate_device(mis_group* g, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                                    const mis_gen_params* params, float* const* pcm_dev, int64_t pcm_stride,
                                                    int64_t* pcm_lens, int32_t* n_tokens) {
    MIS_API_BEGIN
    MIS_REQUIRE(g && prompt_ids && prompt_lens && params && pcm_dev && pcm_lens, MIS_ERR_INVALID_INPUT, "null argument");
    const int W = (int)g->reps.size();
    MIS_REQUIRE(batch >= 1, MIS_ERR_INVALID_INPUT, "empty batch");
    const int We = std::min(W, batch);          // replicas that get rows (a batch smaller than the group runs on its first `batch` replicas)
    for (int r = 0; r < W; ++r) MIS_REQUIRE(pcm_dev[r], MIS_ERR_INVALID_INPUT, "null PCM buffer for replica %d", r);
    HostPrompts hp = fetch_prompts(prompt_ids, prompt_lens, batch);
    std::vector<ShardResult> res(W);
    std::vector<int32_t> ntok(batch, 0);
    std::vector<int64_t> plens(batch, 0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    th.reserve(We);
    struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{th};
    for (int r = 0; r < We; ++r)
        th.emplace_back([&, r]() {
            int lo, hi;
            shard_block(batch, r, We, &lo, &hi);
            mis_gen_params p = *params;
            p.row_offset += lo;
            auto a = std::chrono::steady_clock::now();
            res[r].st = mis_tts_generate_device(g->reps[r], hp.flat.data() + hp.off[lo], hp.lens.data() + lo, hi - lo, &p, nullptr,
                                                pcm_dev[r] + (size_t)lo * pcm_stride, pcm_stride, plens.data() + lo, ntok.data() + lo);
            if (res[r].st != MIS_OK) res[r].err = mis_last_error();      // thread-local message: capture it on this thread
            res[r].ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
        });
    for (auto& t : th) t.join();
    auto t1 = std::chrono::steady_clock::now();
    for (int r = 0; r < We; ++r)
        if (res[r].st != MIS_OK) throw MisError(res[r].st, "shard " + std::to_string(r) + ": " + res[r].err);
    // ---- all-gather by direct peer writes: rank r's block -> every other replica's buffer, issued on r's own stream
    for (int r = 0; r < We; ++r) {
        int lo, hi;
        shard_block(batch, r, We, &lo, &hi);
        const int dr = tts_device(g->reps[r]);
        HIP_CHECK(hipSetDevice(dr));
        hipStream_t s = tts_stream(g->reps[r]);
        const size_t bytes = (size_t)(hi - lo) * pcm_stride * 4;
        for (int q = 0; q < W; ++q) {
            if (q == r || pcm_dev[q] == pcm_dev[r]) continue;
            const int dq = tts_device(g->reps[q]);
            if (dq == dr) HIP_CHECK(hipMemcpyAsync(pcm_dev[q] + (size_t)lo * pcm_stride, pcm_dev[r] + (size_t)lo * pcm_stride, bytes, hipMemcpyDeviceToDevice, s));
            else HIP_CHECK(hipMemcpyPeerAsync(pcm_dev[q] + (size_t)lo * pcm_stride, dq, pcm_dev[r] + (size_t)lo * pcm_stride, dr, bytes, s));
        }
    }
    for (int r = 0; r < W; ++r) { HIP_CHECK(hipSetDevice(tts_device(g->reps[r]))); HIP_CHECK(hipStreamSynchronize(tts_stream(g->reps[r]))); }
    auto t2 = std::chrono::steady_clock::now();
    g->timing.n_shards = We;
    g->timing.generate_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    g->timing.gather_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    g->timing.slowest_shard_ms = 0;
    for (int r = 0; r < We; ++r) g->timing.slowest_shard_ms = std::max(g->timing.slowest_shard_ms, res[r].ms);
    for (int b = 0; b < batch; ++b) pcm_lens[b] = plens[b];
    if (n_tokens) for (int b = 0; b < batch; ++b) n_tokens[b] = ntok[b];
    MIS_API_END
}

// Same, gathering to HOST memory (what a Swift host hands back as MLXArrays): every shard returns its rows in its own pinned host
// buffer (mis_tts_generate), and the rows are then copied host -> host into their place of one [batch, *pcm_stride] buffer - one
// extra host memcpy per row (3 MB per shard at the bench shape), no peer traffic.
extern "C" mis_status mis_tts_group_generate(mis_group* g, const int32_t* prompt_ids, const int32_t* prompt_lens, int batch,
                                             const mis_gen_params* params, float** pcm_out, int64_t* pcm_stride, int64_t* pcm_lens,
                                             int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens) {
    MIS_API_BEGIN
    MIS_REQUIRE(g && prompt_ids && prompt_lens && params && pcm_out && pcm_stride && pcm_lens, MIS_ERR_INVALID_INPUT, "null argument");
    MIS_REQUIRE(batch >= 1, MIS_ERR_INVALID_INPUT, "empty batch");
    const int W = std::min((int)g->reps.size(), batch);      // a batch smaller than the group runs on its first `batch` replicas
    HostPrompts hp = fetch_prompts(prompt_ids, prompt_lens, batch);
    struct Part { float* pcm = nullptr; int64_t stride = 0; int32_t* tok = nullptr; int64_t tstride = 0; };
    std::vector<Part> part(W);
    std::vector<ShardResult> res(W);
    std::vector<int32_t> ntok(batch, 0);
    std::vector<int64_t> plens(batch, 0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    th.reserve(W);
    struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{th};
    for (int r = 0; r < W; ++r)
        th.emplace_back([&, r]() {
            int lo, hi;
            shard_block(batch, r, W, &lo, &hi);
            mis_gen_params p = *params;
            p.row_offset += lo;
            auto a = std::chrono::steady_clock::now();
            res[r].st = mis_tts_generate(g->reps[r], hp.flat.data() + hp.off[lo], hp.lens.data() + lo, hi - lo, &p, nullptr, &part[r].pcm,
                                         &part[r].stride, plens.data() + lo, tokens_out ? &part[r].tok : nullptr, &part[r].tstride,
                                         ntok.data() + lo);
            if (res[r].st != MIS_OK) res[r].err = mis_last_error();
            res[r].ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
        });
    for (auto& t : th) t.join();
    auto t1 = std::chrono::steady_clock::now();
    auto free_parts = [&]() { for (auto& p : part) { if (p.pcm) mis_free(p.pcm); if (p.tok) mis_free(p.tok); p.pcm = nullptr; p.tok = nullptr; } };
    for (int r = 0; r < W; ++r)
        if (res[r].st != MIS_OK) { free_parts(); throw MisError(res[r].st, "shard " + std::to_string(r) + ": " + res[r].err); }
    int64_t longest = 1, tlongest = 1;
    for (int r = 0; r < W; ++r) { longest = std::max(longest, part[r].stride); tlongest = std::max(tlongest, part[r].tstride); }
    try {
        PinnedBuf<float> host((size_t)batch * longest);
        memset(host.p, 0, (size_t)batch * longest * 4);
        PinnedBuf<int32_t> thost;
        if (tokens_out) { thost.alloc((size_t)batch * tlongest); memset(thost.p, 0, (size_t)batch * tlongest * 4); }
        for (int r = 0; r < W; ++r) {
            int lo, hi;
            shard_block(batch, r, W, &lo, &hi);
            for (int b = lo; b < hi; ++b) {
                memcpy(host.p + (size_t)b * longest, part[r].pcm + (size_t)(b - lo) * part[r].stride, (size_t)plens[b] * 4);
                if (tokens_out) memcpy(thost.p + (size_t)b * tlongest, part[r].tok + (size_t)(b - lo) * part[r].tstride, (size_t)part[r].tstride * 4);
            }
        }
        *pcm_out = host.release(); *pcm_stride = longest;
        if (tokens_out) { *tokens_out = thost.release(); if (tokens_stride) *tokens_stride = tlongest; }
    } catch (...) { free_parts(); throw; }
    free_parts();
    auto t2 = std::chrono::steady_clock::now();
    g->timing.n_shards = W;
    g->timing.generate_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    g->timing.gather_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    g->timing.slowest_shard_ms = 0;
    for (int r = 0; r < W; ++r) g->timing.slowest_shard_ms = std::max(g->timing.slowest_shard_ms, res[r].ms);
    for (int b = 0; b < batch; ++b) pcm_lens[b] = plens[b];
    if (n_tokens) for (int b = 0; b < batch; ++b) n_tokens[b] = ntok[b];
    MIS_API_END
}

extern "C" mis_status mis_tts_group_last_timing(mis_group* g, mis_group_timing* out) {
    MIS_API_BEGIN
    MIS_REQUIRE(g && out, MIS_ERR_INVALID_INPUT, "null argument");
    *out = g->timing;
    MIS_API_END
}


// ---------------------------------------------------------------------------- groups of the other families (stateless: replicas[n])
namespace {
struct EventRelay { mis_event_cb cb; void* user; int lo; std::mutex* mu; };
void relay_event(void* u, int row, mis_event_kind kind, const void* payload, int64_t n) {
    EventRelay* r = static_cast<EventRelay*>(u);
    std::lock_guard<std::mutex> g(*r->mu);
    r->cb(r->user, row + r->lo, kind, payload, n);
}
template <typename H>
int check_replicas(H* const* replicas, int n, int batch) {
    MIS_REQUIRE(replicas && n >= 1 && n <= 64, MIS_ERR_INVALID_INPUT, "a group needs 1..64 replicas");
    MIS_REQUIRE(batch >= 1, MIS_ERR_INVALID_INPUT, "empty batch");
    for (int i = 0; i < n; ++i) {
        MIS_REQUIRE(replicas[i], MIS_ERR_INVALID_INPUT, "null replica handle");
        for (int j = 0; j < i; ++j) MIS_REQUIRE(replicas[j] != replicas[i], MIS_ERR_INVALID_INPUT, "a handle may appear only once in a group");
    }
    return std::min(n, batch);          // a batch smaller than the group (the tail slice of a long request) runs on its first `batch` replicas
}
// run fn(r, lo, hi) on one thread per shard; rethrow the first failure
template <typename F>
void run_shards(int n, int batch, F&& fn) {
    std::vector<ShardResult> res(n);
    std::vector<std::thread> th;
    th.reserve(n);
    struct Joiner {                       // a thread constructor that throws (resource exhaustion) must not leave joinable threads behind:
        std::vector<std::thread>& t;      // their destructors would call std::terminate
        ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); }
    } joiner{th};
    for (int r = 0; r < n; ++r)
        th.emplace_back([&, r]() {
            int lo, hi;
            shard_block(batch, r, n, &lo, &hi);
            res[r].st = fn(r, lo, hi);
            if (res[r].st != MIS_OK) res[r].err = mis_last_error();
        });
    for (auto& t : th) t.join();
    for (int r = 0; r < n; ++r)
        if (res[r].st != MIS_OK) throw MisError(res[r].st, "shard " + std::to_string(r) + ": " + res[r].err);
}
// rows of per-shard [rows_r, stride_r] buffers -> one pinned [batch, longest] buffer (zero padded)
template <typename T>
T* merge_rows(const std::vector<T*>& part, const std::vector<int64_t>& stride, int n, int batch, int64_t* longest_out) {
    int64_t longest = 1;
    for (int r = 0; r < n; ++r) longest = std::max(longest, stride[r]);
    PinnedBuf<T> host((size_t)batch * longest);
    memset(host.p, 0, (size_t)batch * longest * sizeof(T));
    for (int r = 0; r < n; ++r) {
        int lo, hi;
        shard_block(batch, r, n, &lo, &hi);
        if (!part[r]) continue;
        for (int b = lo; b < hi; ++b) memcpy(host.p + (size_t)b * longest, part[r] + (size_t)(b - lo) * stride[r], (size_t)stride[r] * sizeof(T));
    }
    *longest_out = longest;
    return host.release();
}
template <typename T>
struct PartGuard {                                    // per-shard library buffers, released on every path
    std::vector<T*> p;
    explicit PartGuard(int n) : p(n, nullptr) {}
    ~PartGuard() { for (auto q : p) if (q) mis_free(q); }
};
}   // namespace

extern "C" mis_status mis_whisper_group_generate(mis_whisper* const* replicas, int n, const float* pcm, const int64_t* lens, int batch,
                                                 int64_t stride, const int32_t* prompt_ids, int n_prompt, const mis_stt_params* sp,
                                                 int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens) {
    MIS_API_BEGIN
    MIS_REQUIRE(pcm && lens && prompt_ids && sp && tokens_out && tokens_stride && n_tokens, MIS_ERR_INVALID_INPUT, "null argument");
    n = check_replicas(replicas, n, batch);
    PartGuard<int32_t> tok(n);
    std::vector<int64_t> ts(n, 0);
    // replicas that share a device keep to kernels that do not wait for co-resident blocks (see mis_tts_group_create)
    struct SharedMarks {
        std::vector<mis_whisper*> m;
        ~SharedMarks() { for (auto* w : m) whisper_internal_set_shared_device(w, false); }
    } marks;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (j != i && whisper_internal_device(replicas[j]) == whisper_internal_device(replicas[i])) {
                whisper_internal_set_shared_device(replicas[i], true);
                marks.m.push_back(replicas[i]);
                break;
            }
    run_shards(n, batch, [&](int r, int lo, int hi) {
        mis_stt_params p = *sp;
        p.row_offset += lo;
        return mis_stt_whisper_generate(replicas[r], pcm + (size_t)lo * stride, lens + lo, hi - lo, stride, prompt_ids, n_prompt, &p, &tok.p[r], &ts[r],
                                        n_tokens + lo);
    });
    *tokens_out = merge_rows(tok.p, ts, n, batch, tokens_stride);
    MIS_API_END
}

extern "C" mis_status mis_soprano_group_generate(mis_soprano* const* replicas, int n, const int32_t* prompt_ids, const int32_t* prompt_lens,
                                                 int batch, const mis_gen_params* params, float** pcm_out, int64_t* pcm_stride, int64_t* pcm_lens,
                                                 int32_t** tokens_out, int64_t* tokens_stride, int32_t* n_tokens) {
    MIS_API_BEGIN
    MIS_REQUIRE(prompt_ids && prompt_lens && params && pcm_out && pcm_stride && pcm_lens, MIS_ERR_INVALID_INPUT, "null argument");
    n = check_replicas(replicas, n, batch);
    HostPrompts hp = fetch_prompts(prompt_ids, prompt_lens, batch);
    PartGuard<float> pcm(n);
    PartGuard<int32_t> tok(n);
    std::vector<int64_t> ps(n, 0), ts(n, 0);
    std::vector<int32_t> ntok(batch, 0);
    // replicas that share a device keep to kernels that do not wait for co-resident blocks (the batch-1 token engine; see mis_tts_group_create)
    struct SharedMarks {
        std::vector<mis_tts*> m;
        ~SharedMarks() { for (auto* t : m) tts_internal_set_shared_device(t, false); }
    } marks;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (j != i && soprano_internal_device(replicas[j]) == soprano_internal_device(replicas[i])) {
                tts_internal_set_shared_device(soprano_internal_lm(replicas[i]), true);
                marks.m.push_back(soprano_internal_lm(replicas[i]));
                break;
            }
    // the LM program of a shard is chosen on the REQUEST's batch, not on the shard's (soprano.hip): a one-row shard of a larger batch runs
    // the launch chain like every other row of it, so the sharded result equals the single-handle result of the same batch
    struct GroupBatch {
        mis_soprano* const* r; int n;
        ~GroupBatch() { for (int i = 0; i < n; ++i) soprano_internal_set_group_batch(r[i], 0); }
    } group_batch{replicas, n};
    for (int i = 0; i < n; ++i) soprano_internal_set_group_batch(replicas[i], batch);
    run_shards(n, batch, [&](int r, int lo, int hi) {
        mis_gen_params p = *params;
        p.row_offset += lo;
        return mis_soprano_generate(replicas[r], hp.flat.data() + hp.off[lo], hp.lens.data() + lo, hi - lo, &p, &pcm.p[r], &ps[r], pcm_lens + lo,
                                    tokens_out ? &tok.p[r] : nullptr, tokens_out ? &ts[r] : nullptr, ntok.data() + lo);
    });
    *pcm_out = merge_rows(pcm.p, ps, n, batch, pcm_stride);
    if (tokens_out) { int64_t tl = 1; *tokens_out = merge_rows(tok.p, ts, n, batch, &tl); if (tokens_stride) *tokens_stride = tl; }
    if (n_tokens) for (int b = 0; b < batch; ++b) n_tokens[b] = ntok[b];
    MIS_API_END
}

extern "C" mis_status mis_qwen3tts_group_generate(mis_qwen3tts* const* replicas, int n, const int32_t* text_ids, const int32_t* codec_ids,
                                                  const int32_t* prefill_lens, int P, const int32_t* trailing_ids, const int32_t* trailing_lens,
                                                  int Tt, int batch, const mis_qwen3tts_params* params, const int32_t* row_max_frames,
                                                  float** pcm_out, int64_t* pcm_stride, int64_t* pcm_lens, int32_t** codes_out,
                                                  int64_t* codes_stride, int32_t* n_frames, int chunk_frames, mis_event_cb on_event, void* user,
                                                  const volatile int* cancel_flag) {
    MIS_API_BEGIN
    MIS_REQUIRE(text_ids && codec_ids && prefill_lens && trailing_lens && params && pcm_out && pcm_stride && pcm_lens, MIS_ERR_INVALID_INPUT,
                "null argument");
    MIS_REQUIRE(P >= 1 && Tt >= 0, MIS_ERR_INVALID_INPUT, "bad prompt sizes");
    n = check_replicas(replicas, n, batch);
    PartGuard<float> pcm(n);
    PartGuard<int32_t> cod(n);
    std::vector<int64_t> ps(n, 0), cs(n, 0);
    std::vector<int32_t> nf(batch, 0);
    std::mutex mu;
    std::vector<EventRelay> relay(n);
    const int Tt1 = std::max(Tt, 1);
    run_shards(n, batch, [&](int r, int lo, int hi) {
        mis_qwen3tts_params p = *params;
        p.row_offset += lo;
        relay[r] = EventRelay{on_event, user, lo, &mu};
        return mis_qwen3tts_generate(replicas[r], text_ids + (size_t)lo * P, codec_ids + (size_t)lo * P, prefill_lens + lo, P,
                                     trailing_ids ? trailing_ids + (size_t)lo * Tt1 : nullptr, trailing_lens + lo, Tt, hi - lo, &p,
                                     row_max_frames ? row_max_frames + lo : nullptr, &pcm.p[r], &ps[r], pcm_lens + lo, codes_out ? &cod.p[r] : nullptr,
                                     codes_out ? &cs[r] : nullptr, nf.data() + lo, chunk_frames, on_event ? relay_event : nullptr, &relay[r],
                                     cancel_flag);
    });
    *pcm_out = merge_rows(pcm.p, ps, n, batch, pcm_stride);
    if (codes_out) {
        // codes rows are [frames][G]: strides are in frames, G ints per frame -> merge in units of G ints
        int64_t longest = 1;
        for (int r = 0; r < n; ++r) longest = std::max(longest, cs[r]);
        const int groups = mis_qwen3tts_num_code_groups(replicas[0]);
        PinnedBuf<int32_t> host((size_t)batch * longest * groups);
        memset(host.p, 0, (size_t)batch * longest * groups * 4);
        for (int r = 0; r < n; ++r) {
            int lo, hi;
            shard_block(batch, r, n, &lo, &hi);
            if (!cod.p[r]) continue;
            for (int b = lo; b < hi; ++b)
                memcpy(host.p + (size_t)b * longest * groups, cod.p[r] + (size_t)(b - lo) * cs[r] * groups, (size_t)cs[r] * groups * 4);
        }
        *codes_out = host.release();
        if (codes_stride) *codes_stride = longest;
    }
    if (n_frames) for (int b = 0; b < batch; ++b) n_frames[b] = nf[b];
    MIS_API_END
}

// ---------------------------------------------------------------------------- test scaffolding (include/mi_speech_debug.h): hold compute
// units from another stream.  The one-launch sampler's 8 x batch blocks wait for each other and need every one resident; its time-out and
// the engines' recovery are tested by taking the CUs away.  Every spinner block announces itself in a host-visible slot, and the host
// returns only once all of them are resident - without that handshake the sampler launched behind the spinner can simply win the race
// for the CUs (first launch of this kernel on a fresh box; round 4's red suite).
__global__ void k_debug_spin(unsigned long long ticks_100mhz, volatile unsigned* resident_slots, const unsigned* release) {
    extern __shared__ unsigned char spin_lds[];          // (128 KB of the CU's 160 KB requested at launch: ONE spinner per CU, and no block that needs
                                                         //  more than 32 KB of LDS - the one-launch sampler takes ~50 KB - fits beside it)
    if (ticks_100mhz == ~0ull) spin_lds[threadIdx.x] = 0;
    if (threadIdx.x == 0) __hip_atomic_store((unsigned*)&resident_slots[blockIdx.x], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks_100mhz) {
        if (__hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;       // (host memory: every wave sees the same word)
        __builtin_amdgcn_s_sleep(64);
    }
}
static hipStream_t g_occupy_stream = nullptr;
static unsigned* g_occupy_host = nullptr;            // pinned, coherent: [0] = release flag, [1 ..] = one residency slot per spinner block
static constexpr int OCCUPY_MAX_BLOCKS = 4096;
static void occupy_release_and_wait() {
    if (!g_occupy_stream) return;
    if (g_occupy_host) __atomic_store_n(&g_occupy_host[0], 1u, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(g_occupy_stream);
    (void)hipStreamDestroy(g_occupy_stream);
    g_occupy_stream = nullptr;
}
extern "C" mis_status mis_debug_occupy_cus(int device, int blocks, int threads, double seconds) {
    MIS_API_BEGIN
    MIS_REQUIRE(blocks >= 1 && blocks <= OCCUPY_MAX_BLOCKS && threads >= 64 && threads <= 1024 && threads % 64 == 0 && seconds > 0.0 &&
                seconds <= 10.0, MIS_ERR_INVALID_INPUT, "bad argument");
    HIP_CHECK(hipSetDevice(device));
    occupy_release_and_wait();                                        // (a spinner of an earlier call)
    if (!g_occupy_host) HIP_CHECK(hipHostMalloc((void**)&g_occupy_host, (size_t)(OCCUPY_MAX_BLOCKS + 1) * sizeof(unsigned), hipHostMallocCoherent));
    memset(g_occupy_host, 0, (size_t)(OCCUPY_MAX_BLOCKS + 1) * sizeof(unsigned));
    HIP_CHECK(hipStreamCreateWithFlags(&g_occupy_stream, hipStreamNonBlocking));
    static const bool lds_ok = hipFuncSetAttribute((const void*)k_debug_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
    MIS_REQUIRE(lds_ok, MIS_ERR_DEVICE, "cannot reserve 128 KB of LDS for the spinner");
    hipLaunchKernelGGL(k_debug_spin, dim3(blocks), dim3(threads), 128 * 1024, g_occupy_stream, (unsigned long long)(seconds * 1e8),
                       (volatile unsigned*)(g_occupy_host + 1), (const unsigned*)g_occupy_host);
    HIP_CHECK(hipGetLastError());
    // handshake: all `blocks` spinners resident (each wrote its slot) before anything else is launched by the caller
    const auto t0 = std::chrono::steady_clock::now();
    int resident = 0;
    for (;;) {
        resident = 0;
        for (int b = 0; b < blocks; ++b) resident += __atomic_load_n(&g_occupy_host[1 + b], __ATOMIC_ACQUIRE) != 0;
        if (resident == blocks) break;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    if (resident != blocks) {
        occupy_release_and_wait();
        char msg[128];
        snprintf(msg, sizeof(msg), "spinner: only %d of %d blocks became resident within 2 s", resident, blocks);
        throw MisError(MIS_ERR_DEVICE, msg);
    }
    MIS_API_END
}
extern "C" mis_status mis_debug_occupy_wait(void) {
    MIS_API_BEGIN
    occupy_release_and_wait();
    MIS_API_END
}
extern "C" int32_t mis_debug_device_cus(int device) {
    hipDeviceProp_t prop{};
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
    return prop.multiProcessorCount;
}
