#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void ks(unsigned* c, unsigned* out, unsigned long long* lat) {
    unsigned r;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(c), "0"(1u) : "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = r; lat[blockIdx.x] = t1 - t0; }
}
__global__ void kv(unsigned* c, unsigned* out, unsigned long long* lat) {
    unsigned r = 1u, z = 0u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        asm volatile("global_atomic_add %0, %1, %0, %2 sc0\n\ts_waitcnt vmcnt(0)" : "+v"(r) : "v"(z), "s"(c) : "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = r; lat[blockIdx.x] = t1 - t0; }
}
int main() {
    const int NB = 4096;
    unsigned *c, *out; unsigned long long* lat;
    hipMalloc(&c, 4096); hipMalloc(&out, NB * 4); hipMalloc(&lat, NB * 8);
    for (int which = 0; which < 2; ++which) {
        hipMemset(c, 0, 4096);
        if (which == 0) hipLaunchKernelGGL(ks, dim3(NB), dim3(64), 0, 0, c, out, lat);
        else hipLaunchKernelGGL(kv, dim3(NB), dim3(64), 0, 0, c, out, lat);
        hipError_t e = hipDeviceSynchronize();
        std::vector<unsigned> h(NB); std::vector<unsigned long long> l(NB); unsigned cv;
        hipMemcpy(h.data(), out, NB * 4, hipMemcpyDeviceToHost); hipMemcpy(l.data(), lat, NB * 8, hipMemcpyDeviceToHost); hipMemcpy(&cv, c, 4, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        bool uniq = true; for (int i = 0; i < NB; ++i) if (h[i] != (unsigned)i) uniq = false;
        std::sort(l.begin(), l.end());
        printf("%s: err %d counter %u tickets unique 0..N-1: %d  latency (100 MHz ticks) median %llu p90 %llu max %llu\n", which ? "vector" : "scalar", (int)e, cv, (int)uniq, l[NB/2], l[NB*9/10], l[NB-1]);
    }
    return 0;
}
