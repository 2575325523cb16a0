// common.hpp -- context, error handling and device buffers shared by every translation unit
// of libemdee_hip.so.  gfx950 only: no CUDA/HIP dual paths, no portability layer.
#pragma once

#include <hip/hip_runtime.h>

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

// Wrap a C-ABI entry point body: translates Failure / std::exception into a status code.
template <class F>
static inline int32_t guarded(F &&body) {
    try {
        body();
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

static __global__ void k_post_words(const int *__restrict__ src, int n, volatile int *dst, volatile int *stamp, int seq) {
    const int t = threadIdx.x;
    if (t < n) dst[t] = src[t];
    __threadfence_system();
    __syncthreads();
    if (t == 0) *stamp = seq;
}

// n <= POST_MAX words of device memory, as they are when the work queued on s so far has run -> out (host); blocking
static inline void read_back_words(emdee_ctx *ctx, hipStream_t s, const int *dev, int n, int32_t *out) {
    EMDEE_REQUIRE(n >= 0 && n <= POST_MAX, EMDEE_ERR_INVALID, "read_back_words: %d words", n);
    if (n == 0) return;
    int32_t *data = ctx->host_flags + POST_DATA;
    // (EMDEE_READBACK is looked at per call, not per context: a process may hold one context for its whole life)
    const char *form = std::getenv("EMDEE_READBACK");
    if ((form != nullptr && form[0] == 'c') || ctx->post_dev == nullptr) {
        EMDEE_HIP_CHECK(hipMemcpyAsync(data, dev, n * sizeof(int), hipMemcpyDeviceToHost, s));
        EMDEE_HIP_CHECK(hipStreamSynchronize(s));
    } else {
        if (++ctx->post_seq == 0) ++ctx->post_seq;                         // never 0 (the buffer's initial contents)
        const int seq = (int)ctx->post_seq;
        hipLaunchKernelGGL(k_post_words, dim3(1), dim3(POST_MAX), 0, s, dev, n, (volatile int *)(ctx->post_dev + POST_DATA),
                           (volatile int *)(ctx->post_dev + POST_STAMP), seq);
        volatile int32_t *stamp = ctx->host_flags + POST_STAMP;
        for (unsigned spins = 1; *stamp != seq; spins++) {
            if ((spins & 0x3fffu) == 0) {                                  // a stream that failed would never stamp
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
