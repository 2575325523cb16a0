// readback_latency.hip -- what a host round trip between two kernels costs on this box, by method.
//   hipcc -O2 --offload-arch=gfx950 readback_latency.hip -o readback_latency && ./readback_latency
// Kernel A (about 100 us) ends by stamping the device wall clock; the host learns a word A produced, then launches
// kernel B, which stamps its start.  gap = B.start - A.end = empty queue the round trip leaves behind.
//   copy+sync : hipMemcpyAsync(D2H into pinned memory) + hipStreamSynchronize   (what the library did up to round 3)
//   event spin: hipMemcpyAsync + hipEventRecord + hipEventQuery loop
//   mapped    : A writes the word and a sequence stamp straight into pinned host memory (system-scope release),
//               the host spins on the stamp; no copy command, no synchronize
//   none      : B launched behind A without waiting (the floor: back-to-back launches)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_a(int iters, float *sink, int *word, long long *stamp, volatile int *host_word, volatile int *host_seq, int seq) {
    float a = threadIdx.x;
    for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) sink[0] = a;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        word[0] = seq * 3;
        stamp[0] = wall_clock64();
        if (host_word) {
            host_word[0] = seq * 3;
            __threadfence_system();
            host_seq[0] = seq;
        }
    }
}
__global__ void k_b(long long *stamp) {
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *sink; int *word; long long *stamp;
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&word, 4)); CK(hipMalloc(&stamp, 16));
    int *pinned, *mapped;
    CK(hipHostMalloc((void **)&pinned, 64, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&mapped, 64, hipHostMallocMapped | hipHostMallocCoherent));
    int *mapped_dev;
    CK(hipHostGetDevicePointer((void **)&mapped_dev, mapped, 0));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const int iters = 20000, reps = 200;
    const char *names[] = {"none", "copy+sync", "event spin", "mapped"};
    for (int method = 0; method < 4; method++) {
        std::vector<double> gaps, walls;
        for (int r = 0; r < reps + 20; r++) {
            const int seq = method * 1000 + r + 1;
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_a, dim3(256), dim3(256), 0, s, iters, sink, word, stamp, method == 3 ? mapped_dev : nullptr,
                               method == 3 ? mapped_dev + 8 : nullptr, seq);
            int got = seq * 3;
            if (method == 1) {
                CK(hipMemcpyAsync(pinned, word, 4, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
                got = pinned[0];
            } else if (method == 2) {
                CK(hipMemcpyAsync(pinned, word, 4, hipMemcpyDeviceToHost, s));
                CK(hipEventRecord(ev, s));
                while (hipEventQuery(ev) == hipErrorNotReady) {}
                got = pinned[0];
            } else if (method == 3) {
                volatile int *q = mapped + 8;
                while (*q != seq) {}
                got = ((volatile int *)mapped)[0];
            }
            if (got != seq * 3) { printf("method %d: wrong word %d != %d\n", method, got, seq * 3); return 1; }
            hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s, stamp);
            long long h[2];
            CK(hipMemcpyAsync(h, stamp, 16, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            auto t1 = std::chrono::steady_clock::now();
            if (r >= 20) {
                gaps.push_back((double)(h[1] - h[0]) / rate_khz * 1e3);
                walls.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
            }
        }
        std::sort(gaps.begin(), gaps.end());
        std::sort(walls.begin(), walls.end());
        printf("%-10s gap between kernels: median %6.1f us  p10 %6.1f  p90 %6.1f   (host wall per iteration, median %7.1f us)\n",
               names[method], gaps[reps / 2], gaps[reps / 10], gaps[reps * 9 / 10], walls[reps / 2]);
    }
    return 0;
}
