// common.hpp -- context, error handling and device buffers shared by every translation unit
// of libemdee_hip.so.  gfx950 only: no CUDA/HIP dual paths, no portability layer.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
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
    int32_t *host_flags = nullptr;   // pinned, 16 ints, for small blocking read-backs
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
