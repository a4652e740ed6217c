// Diagnostic (never shipped, built on the GPU box by scripts/hog_probe.sh): a kernel that holds `blocks` CUs for `cycles` shader cycles - the stand-in for an RCCL
// all-reduce kernel on another stream.  1024 threads and 64 KiB of LDS per block: no block of the MFMA kernels (127-157 KiB of LDS, the whole register file) fits beside it.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(1024) void cu_hog_kernel(long long cycles, unsigned* sink) {
    __shared__ unsigned pad[16384];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = (long long)__builtin_readcyclecounter();
    unsigned v = pad[(threadIdx.x * 7) & 16383];
    while ((long long)__builtin_readcyclecounter() - t0 < cycles) {          // every wave reaches the bound: the grid always drains
        v = v * 1664525u + 1013904223u;
        __builtin_amdgcn_s_sleep(8);
    }
    if (v == 0x12345u) sink[0] = v;
}

extern "C" int cu_hog(int blocks, long long cycles, unsigned* sink, void* stream) {
    hipLaunchKernelGGL(cu_hog_kernel, dim3(blocks), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), cycles, sink);
    return (int)hipGetLastError();
}
