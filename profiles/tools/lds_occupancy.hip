// How many 512-thread workgroups does a CU hold as a function of the dynamic LDS a kernel asks for?
// (hipOccupancyMaxActiveBlocksPerMultiprocessor on a trivial kernel, so LDS is the only limiter.)
//   hipcc --offload-arch=gfx950 -o profiles/tools/lds_occupancy profiles/tools/lds_occupancy.hip && profiles/tools/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned char s_dyn[];
__global__ __launch_bounds__(512) void k(int *out) { s_dyn[threadIdx.x] = 1; __syncthreads(); if (threadIdx.x == 0) out[blockIdx.x] = s_dyn[1]; }
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    std::printf("%s: sharedMemPerBlock %zu, sharedMemPerMultiprocessor %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.gcnArchName,
                p.sharedMemPerBlock, (size_t)p.sharedMemPerMultiprocessor, (size_t)p.maxSharedMemoryPerMultiProcessor);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int prev = -1;
    for (size_t b = 30 * 1024; b <= 160 * 1024; b += 256) {
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 512, b);
        if (nb != prev) std::printf("dynamic LDS %7zu B (%.2f KB): %d workgroups of 512 per CU\n", b, b / 1024.0, nb);
        prev = nb;
    }
    return 0;
}
