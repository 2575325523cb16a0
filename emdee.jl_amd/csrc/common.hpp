// common.hpp -- context, error handling and device buffers shared by every translation unit
// of libemdee_hip.so.  gfx950 only: no CUDA/HIP dual paths, no portability layer.
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/emdee_hip.h"

namespace emdee {

// ---- error plumbing: no exception crosses the C ABI -------------------------------------------
void set_error(const char *fmt, ...);
const char *get_error();

struct Failure {
    int32_t code;
};

#define EMDEE_HIP_CHECK(expr)                                                                    \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            ::emdee::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            (void)hipGetLastError(); /* reported here: do not leave it for an unrelated later call */ \
            throw ::emdee::Failure{EMDEE_ERR_HIP};                                               \
        }                                                                                        \
    } while (0)

#define EMDEE_REQUIRE(cond, code, ...)                                                           \
    do {                                                                                         \
        if (!(cond)) {                                                                           \
            ::emdee::set_error(__VA_ARGS__);                                                     \
            throw ::emdee::Failure{code};                                                        \
        }                                                                                        \
    } while (0)

// ---- bounds-checked device build (make BOUNDS=1 -> libemdee_hip_bounds.so) ------------------------------
// Every device-side write (and list read) whose index comes from DATA -- counts, prefix sums, slots -- into a buffer sized
// by a CAPACITY goes through EMDEE_BOUND(site, index, capacity): in the product build that is `true` and costs nothing; in
// the bounds build an index outside [0, capacity) is NOT used, the first such event is recorded in a sticky device word of
// the translation unit, and every C-ABI call from then on returns EMDEE_ERR_OVERFLOW with the site, index and capacity in
// emdee_last_error() -- an error message where the product build would fault (round 3: SIGABRT of the host process).
enum BoundSite {
    BS_NONE = 0, BS_BUILD_TILE = 1, BS_BUILD_OWN = 2, BS_BUILD_ROWBUF = 3, BS_BUILD_ROW = 4, BS_FORCE_TILE = 5, BS_FORCE_OWN = 6,
    BS_TYPED_TILE = 7, BS_TYPED_OWN = 8, BS_TYPED_ROWBUF = 9, BS_TYPED_ROW = 10, BS_DD_MIG_PACK = 11, BS_DD_ASSEMBLE = 12,
    BS_DD_GHOST_PACK = 13, BS_DD_GHOST_UNPACK = 14, BS_DD_STEP_PACK = 15, BS_DD_STEP_UNPACK = 16, BS_CELL_SCATTER = 17,
    BS_TBUILD_CAND = 18, BS_TBUILD_ROWBUF = 19, BS_PART_SCATTER = 20, BS_COUNT = 21
};
static inline const char *bound_site_name(int s) {
    static const char *names[BS_COUNT] = {"none", "build: tile slot", "build: own-atom table", "build: row buffer", "build: row of the list",
        "force: tile slot", "force: own-atom table", "typed build: tile slot", "typed build: own-atom table", "typed build: row buffer",
        "typed build: row of the list", "dd: migrant message", "dd: assembled owned arrays", "dd: ghost send list", "dd: ghost rows",
        "dd: step message", "dd: step unpack", "cells: scatter", "transposed build: candidate table", "transposed build: row buffer",
        "partition: scatter"};
    return (s > 0 && s < BS_COUNT) ? names[s] : "unknown site";
}
#ifdef EMDEE_BOUNDS
static __device__ int g_bounds_word[4];                // {site, index (clamped to int), capacity, 0}: first event wins, never cleared
__device__ static inline bool bound_ok(int site, long long idx, long long cap) {
    if (idx >= 0 && idx < cap) return true;
    if (atomicCAS(&g_bounds_word[0], 0, site) == 0) {
        g_bounds_word[1] = (int)(idx > 2147483647LL ? 2147483647LL : (idx < -2147483647LL ? -2147483647LL : idx));
        g_bounds_word[2] = (int)(cap > 2147483647LL ? 2147483647LL : cap);
    }
    return false;
}
#define EMDEE_BOUND(site, idx, cap) (::emdee::bound_ok((site), (long long)(idx), (long long)(cap)))
// the sticky word of THIS translation unit (each .hip file defines one poll function with it; capi.hip asks all of them)
static inline void bounds_poll_here(int out[3]) {
    int w[4] = {0, 0, 0, 0};
    if (hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(w, HIP_SYMBOL(g_bounds_word), sizeof(w)) == hipSuccess) {
        out[0] = w[0]; out[1] = w[1]; out[2] = w[2];
    } else {
        (void)hipGetLastError();
        out[0] = out[1] = out[2] = 0;
    }
}
void bounds_poll_f32(int out[3]);
void bounds_poll_f64(int out[3]);
void bounds_poll_capi(int out[3]);
static inline void bounds_raise_if_set() {
    int w[3];
    void (*polls[3])(int *) = {bounds_poll_f64, bounds_poll_f32, bounds_poll_capi};
    for (auto poll : polls) {
        poll(w);
        if (w[0] != 0) {
            set_error("device bounds check (%s): index %d outside capacity %d -- the access was skipped; results from here on are not to be used",
                      bound_site_name(w[0]), w[1], w[2]);
            throw Failure{EMDEE_ERR_OVERFLOW};
        }
    }
}
#else
#define EMDEE_BOUND(site, idx, cap) (true)
#endif

// Wrap a C-ABI entry point body: translates Failure / std::exception into a status code.
template <class F>
static inline int32_t guarded(F &&body) {
    try {
        body();
#ifdef EMDEE_BOUNDS
        bounds_raise_if_set();
#endif
        return EMDEE_OK;
    } catch (const Failure &f) {
        return f.code;
    } catch (const std::exception &e) {
        set_error("unexpected exception: %s", e.what());
        return EMDEE_ERR_INVALID;
    } catch (...) {
        set_error("unknown exception");
        return EMDEE_ERR_INVALID;
    }
}

}  // namespace emdee

// The opaque context of the C ABI: one device, one stream.
struct emdee_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int cu_count = 0;
    size_t hbm_bytes = 0;
    char arch[64] = {0};
    int32_t *host_flags = nullptr;   // pinned, device-visible: [0, 16) small blocking read-backs (copy + synchronize),
                                     // [POST_DATA, POST_DATA + POST_MAX) + stamp at POST_STAMP: read-backs posted by a kernel
    int32_t *post_dev = nullptr;     // the same memory as the device sees it
    uint32_t post_seq = 0;           // (EMDEE_READBACK=copy: every read-back as copy + synchronize, A/B)
    double readback_ms = 0.0;        // host wall-clock spent in blocking read-backs, and how many (emdee_dd_phase_times)
    int64_t readbacks = 0;
};

namespace emdee {

static inline void use_device(const emdee_ctx *ctx) { EMDEE_HIP_CHECK(hipSetDevice(ctx->device)); }

// Grow-only device buffer.
template <typename T>
struct DevBuf {
    T *ptr = nullptr;
    size_t cap = 0;   // elements
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
    // returns true if it (re)allocated; contents are NOT preserved
    bool ensure(size_t n) {
        if (n <= cap) return false;
        release();
        size_t want = n + n / 16 + 64;
        hipError_t e = hipMalloc((void **)&ptr, want * sizeof(T));
        if (e != hipSuccess) {
            ptr = nullptr;
            set_error("hipMalloc(%zu bytes) failed: %s", want * sizeof(T), hipGetErrorString(e));
            throw Failure{EMDEE_ERR_ALLOC};
        }
        cap = want;
        return true;
    }
    void swap(DevBuf &o) {
        std::swap(ptr, o.ptr);
        std::swap(cap, o.cap);
    }
};

static inline unsigned blocks_for(size_t n, unsigned threads) { return (unsigned)((n + threads - 1) / threads); }

// An engine that runs on a stream of the library's own (the domains of an in-process decomposition) fills arrays the CALLER
// handed in -- emdee_dd_get_state, emdee_md_get_state / emdee_md_nbr_list on a domain's engine.  The caller's context
// stream is where the caller orders its own work on those arrays (an allocator may have recycled the block from work that
// is still queued there), so the engine's stream first waits for what that stream holds now, and that stream then waits for
// what the engine writes: two event hops, no device-wide synchronisation, and the results are ordered on the caller's
// stream exactly as the results of every other call are.  (Round 4 fenced in the Python binding with a device
// synchronisation; a C or Julia caller of the same entry points had the same race.)  No-op when the two are one stream.
struct FenceOut {
    hipStream_t caller, engine;
    bool live;
    FenceOut(const emdee_ctx *caller_ctx, hipStream_t engine_stream)
        : caller(caller_ctx ? caller_ctx->stream : nullptr), engine(engine_stream), live(caller_ctx != nullptr && caller_ctx->stream != engine_stream) {
#ifdef EMDEE_NO_FENCE                                     // (A/B build only, profiles/build_variant.sh: shows that the test of the fence sees the race)
        live = false;
#endif
        if (live) hop(caller, engine);
    }
    ~FenceOut() {
        if (live) {
            try { hop(engine, caller); } catch (...) {}
        }
    }
    static void hop(hipStream_t from, hipStream_t to) {
        hipEvent_t ev;
        EMDEE_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, from);
        if (e == hipSuccess) e = hipStreamWaitEvent(to, ev, 0);
        (void)hipEventDestroy(ev);                   // (released once the record has completed)
        EMDEE_HIP_CHECK(e);
    }
};

// ------------------------------------------------------------------------------------ small blocking read-backs
// A rebuild decision, the build's overflow words, the counts of a migration: a few words the host must see before it can
// queue the next kernel.  hipMemcpyAsync + hipStreamSynchronize leaves 16 us of empty queue behind the producing kernel
// on this machine; a one-workgroup kernel that writes the words and then a sequence stamp straight into pinned host
// memory (system-scope release), with the host spinning on the stamp, leaves 6 (profiles/tools/readback_latency.hip).
constexpr int HOST_WORDS = 256, POST_DATA = 64, POST_MAX = 128, POST_STAMP = 200;

static inline void host_words_alloc(emdee_ctx *ctx) {
    EMDEE_HIP_CHECK(hipHostMalloc((void **)&ctx->host_flags, HOST_WORDS * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
    memset(ctx->host_flags, 0, HOST_WORDS * sizeof(int32_t));
    EMDEE_HIP_CHECK(hipHostGetDevicePointer((void **)&ctx->post_dev, ctx->host_flags, 0));
    ctx->post_seq = 0;
}

static __global__ void k_post_words(const int *__restrict__ src, int n, volatile int *dst, volatile int *stamp, int seq,
                                    const int *__restrict__ src2 = nullptr, int n2 = 0) {
    const int t = threadIdx.x;
    if (t < n) dst[t] = src[t];
    else if (t < n + n2) dst[t] = src2[t - n];
    __threadfence_system();
    __syncthreads();
    if (t == 0) *stamp = seq;
}

// n <= POST_MAX words of device memory, as they are when the work queued on s so far has run -> out (host); blocking
// (dev2 / n2 / out2, optional: a second range that rides along -- the words of a decomposed rebuild with the build's)
static inline void read_back_words(emdee_ctx *ctx, hipStream_t s, const int *dev, int n, int32_t *out, const int *dev2 = nullptr,
                                   int n2 = 0, int32_t *out2 = nullptr) {
    EMDEE_REQUIRE(n >= 0 && n2 >= 0 && n + n2 <= POST_MAX, EMDEE_ERR_INVALID, "read_back_words: %d words", n + n2);
    if (n + n2 == 0) return;
    const auto t_begin = std::chrono::steady_clock::now();
    struct Clock {
        emdee_ctx *c;
        std::chrono::steady_clock::time_point t0;
        ~Clock() { c->readback_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); c->readbacks++; }
    } clock{ctx, t_begin};
    int32_t *data = ctx->host_flags + POST_DATA;
    // (EMDEE_READBACK is looked at per call, not per context: a process may hold one context for its whole life)
    const char *form = std::getenv("EMDEE_READBACK");
    if ((form != nullptr && form[0] == 'c') || ctx->post_dev == nullptr) {
        if (n > 0) EMDEE_HIP_CHECK(hipMemcpyAsync(data, dev, n * sizeof(int), hipMemcpyDeviceToHost, s));
        if (n2 > 0) EMDEE_HIP_CHECK(hipMemcpyAsync(data + n, dev2, n2 * sizeof(int), hipMemcpyDeviceToHost, s));
        EMDEE_HIP_CHECK(hipStreamSynchronize(s));
    } else {
        if (++ctx->post_seq == 0) ++ctx->post_seq;                         // never 0 (the buffer's initial contents)
        const int seq = (int)ctx->post_seq;
        hipLaunchKernelGGL(k_post_words, dim3(1), dim3(POST_MAX), 0, s, dev, n, (volatile int *)(ctx->post_dev + POST_DATA),
                           (volatile int *)(ctx->post_dev + POST_STAMP), seq, dev2, n2);
        volatile int32_t *stamp = ctx->host_flags + POST_STAMP;
        for (unsigned spins = 1; *stamp != seq; spins++) {
            __builtin_ia32_pause();                                        // (the sibling hyper-thread may be another rank's host thread)
            if ((spins & 0x3fffu) == 0) {                                  // a stream that failed would never stamp
                // A stream that is blocked, not failed (a peer's message that has not arrived yet), can keep us here for
                // long: after ~2 ms of spinning the core is given up and the stream is waited for the ordinary way; the
                // 6 us fast path of a short queue is untouched.
                if (std::chrono::steady_clock::now() - t_begin > std::chrono::milliseconds(2)) {
                    EMDEE_HIP_CHECK(hipStreamSynchronize(s));
                    EMDEE_REQUIRE(*stamp == seq, EMDEE_ERR_HIP, "read-back: the stream drained without posting its words");
                    break;
                }
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess) {
                    EMDEE_REQUIRE(*stamp == seq, EMDEE_ERR_HIP, "read-back: the stream drained without posting its words");
                } else if (q != hipErrorNotReady) {
                    EMDEE_HIP_CHECK(q);
                }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    for (int k = 0; k < n; k++) out[k] = ((volatile int32_t *)data)[k];
    for (int k = 0; k < n2; k++) out2[k] = ((volatile int32_t *)data)[n + k];
}

// HIP-event pair pool for per-kernel device timing on the context's stream (bench.py's
// roofline numbers come from here).
struct KernelTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;
    size_t used = 0;
    double total_ms = 0.0;
    int64_t launches = 0;
    int64_t dropped = 0;   // recorded launches that turned out to be no-ops (run-ahead steps past a rebuild request)
    ~KernelTimer() {
        for (auto &p : pairs) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    }
    size_t begin(hipStream_t s) {
        if (used == pairs.size()) {
            hipEvent_t a, b;
            EMDEE_HIP_CHECK(hipEventCreate(&a));
            EMDEE_HIP_CHECK(hipEventCreate(&b));
            pairs.emplace_back(a, b);
        }
        EMDEE_HIP_CHECK(hipEventRecord(pairs[used].first, s));
        return used++;
    }
    void end(size_t k, hipStream_t s) { EMDEE_HIP_CHECK(hipEventRecord(pairs[k].second, s)); }
    // blocking: folds all recorded pairs into total_ms / launches
    void collect() {
        for (size_t k = 0; k < used; k++) {
            EMDEE_HIP_CHECK(hipEventSynchronize(pairs[k].second));
            float ms = 0.f;
            EMDEE_HIP_CHECK(hipEventElapsedTime(&ms, pairs[k].first, pairs[k].second));
            total_ms += ms;
            launches++;
        }
        launches -= dropped;
        dropped = 0;
        used = 0;
    }
    void reset() {
        used = 0;
        total_ms = 0.0;
        launches = 0;
        dropped = 0;
    }
};

}  // namespace emdee
