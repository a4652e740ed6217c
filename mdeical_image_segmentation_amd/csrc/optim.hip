// Global-norm gradient clipping + AdamW on flat fp32 buffers (one launch per parameter region), gfx950.
// Reference behaviour: HF Trainer inner step = torch.nn.utils.clip_grad_norm_(params, 1.0) then torch.optim.AdamW
// (decoupled weight decay, bias correction), driven through trainer/MYtrainer.py:6-11.
#include <math.h>

#include "common.hpp"

constexpr int SS_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const long long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = g4[i];
        s = fmaf(v.x, v.x, s);
        s = fmaf(v.y, v.y, s);
        s = fmaf(v.z, v.z, s);
        s = fmaf(v.w, v.w, s);
    }
    if (blockIdx.x == 0) {
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s = fmaf(g[i], g[i], s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

extern "C" int mis_sumsq_npartials(long long n) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    (void)n;
    return SS_BLOCKS;
}
extern "C" size_t mis_adamw_workspace_bytes(long long n) {
    (void)n;
    return SS_BLOCKS * sizeof(float);
}
extern "C" int mis_sumsq(const float* g, long long n, float* workspace, void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    MIS_REQUIRE(g && workspace && n > 0, MIS_EINVAL, "sumsq: bad argument");
    MIS_REQUIRE((reinterpret_cast<uintptr_t>(g) & 15) == 0, MIS_EINVAL, "sumsq: g must be 16-byte aligned");
    hipLaunchKernelGGL(sumsq_kernel, dim3(SS_BLOCKS), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, n, workspace);
    MIS_LAUNCH_CHECK("sumsq");
    return MIS_OK;
}

// Several sumsq partial arrays may be concatenated (one per parameter region): npartials is the total count.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    long long n, const float* __restrict__ partials, int npartials, float max_norm, float lr,
                                                    float beta1, float beta2, float eps, float decay_mul, float step_size, float bc2_sqrt,
                                                    float* __restrict__ gradnorm_out, const float* __restrict__ hyper) {
    if (hyper != nullptr) {      // step-dependent scalars from device memory (hipGraph replays: the host values would be frozen into the graph)
        decay_mul = hyper[0];
        step_size = hyper[1];
        bc2_sqrt = hyper[2];
    }
    __shared__ double red[4];
    __shared__ float coef_s;
    float coef = 1.f;
    if (partials != nullptr) {
        double s = 0.0;
        for (int i = threadIdx.x; i < npartials; i += 256) s += (double)partials[i];
        s = wave_sum_d(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float total = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
            float c = 1.f;
            if (max_norm > 0.f) {
                c = max_norm / (total + 1e-6f);
                if (c > 1.f) c = 1.f;
            }
            coef_s = c;
            if (blockIdx.x == 0 && gradnorm_out != nullptr) *gradnorm_out = total;
        }
        __syncthreads();
        coef = coef_s;
    }
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * coef;
        float pi = p[i] * decay_mul;
        float mi = m[i];
        mi = mi + omb1 * (gi - mi);
        float vi = v[i] * beta2 + omb2 * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

extern "C" int mis_adamw_step(float* p, const float* g, float* m, float* v, long long n, const float* sumsq_partials, int npartials,
                              float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float* gradnorm_out,
                              void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    MIS_REQUIRE(p && g && m && v && n > 0 && step >= 1, MIS_EINVAL, "adamw: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float decay_mul = 1.f - lr * weight_decay;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, n, sumsq_partials,
                       npartials, max_norm, lr, beta1, beta2, eps, decay_mul, step_size, bc2_sqrt, gradnorm_out, (const float*)nullptr);
    MIS_LAUNCH_CHECK("adamw");
    return MIS_OK;
}

// Device-resident optimizer state for captured (hipGraph) train steps: *step is advanced on the device, lr is read from device memory.
__global__ void adamw_hyper_kernel(int* __restrict__ step, int advance, const float* __restrict__ lr, float beta1, float beta2, float weight_decay,
                                   float* __restrict__ hyper) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int s = *step;
    if (advance) {
        s += 1;
        *step = s;
    }
    const double l = (double)lr[0];
    const double bc1 = 1.0 - pow((double)beta1, (double)s), bc2 = 1.0 - pow((double)beta2, (double)s);
    hyper[0] = 1.f - lr[0] * weight_decay;
    hyper[1] = (float)(l / bc1);
    hyper[2] = (float)sqrt(bc2);
}

extern "C" int mis_adamw_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* sumsq_partials, int npartials, float max_norm,
                                  const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, int* step_dev, int advance,
                                  float* hyper_ws, float* gradnorm_out, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(p && g && m && v && n > 0 && lr_dev && step_dev && hyper_ws, MIS_EINVAL, "adamw_step_dev: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adamw_hyper_kernel, dim3(1), dim3(64), 0, st, step_dev, advance, lr_dev, beta1, beta2, weight_decay, hyper_ws);
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, n, sumsq_partials, npartials, max_norm, 0.f, beta1, beta2, eps,
                       0.f, 0.f, 1.f, gradnorm_out, (const float*)hyper_ws);
    MIS_LAUNCH_CHECK("adamw_step_dev");
    return MIS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Fragment probes (tests/test_gpu_fragments.py): pin the MFMA operand / result lane maps and the LDS transpose read
// that conv_igemm.hip and wgrad.hip rely on, each on its own, with asymmetric data.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

__global__ void probe_kernel(int which, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c) {
    __shared__ __attribute__((aligned(16))) char lds[32 * 160];
    const int lane = threadIdx.x;
    const int li = lane & 15, lg = lane >> 4;
    if (which == 0) {   // bf16 16x16x32: A [16][32], B [32][16] (row-major floats) -> C [16][16]
        float fa[8], fb[8];
        for (int j = 0; j < 8; ++j) {
            fa[j] = a[li * 32 + 8 * lg + j];
            fb[j] = b[(8 * lg + j) * 16 + li];
        }
        u32x4 A = pack_chunk<__bf16>(fa), B = pack_chunk<__bf16>(fb);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mma_b128<__bf16>(acc, A, B);
        for (int r = 0; r < 4; ++r) c[(lg * 4 + r) * 16 + li] = acc[r];
    } else if (which == 1) {   // f32 16x16x4 x4 (the b128 step): A [16][16], B [16][16] -> C [16][16]
        float fa[4], fb[4];
        for (int t = 0; t < 4; ++t) {
            fa[t] = a[li * 16 + 4 * lg + t];
            fb[t] = b[(4 * lg + t) * 16 + li];
        }
        u32x4 A = pack_chunk<float>(fa), B = pack_chunk<float>(fb);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mma_b128<float>(acc, A, B);
        for (int r = 0; r < 4; ++r) c[(lg * 4 + r) * 16 + li] = acc[r];
    } else {   // LDS transpose read: image [32 pixels][16 channels] bf16, pixel stride 160 B; c[lane][8]
        for (int i = lane; i < 32 * 16; i += 64) {
            const int px = i >> 4, ch = i & 15;
            reinterpret_cast<__bf16*>(lds + px * 160)[ch] = (__bf16)a[i];
        }
        __syncthreads();
        const int q = li >> 2, pp = li & 3;
        for (int s = 0; s < 2; ++s) {
            const int px = 8 * lg + 4 * s + q;
            const uint32_t addr = (uint32_t)(uintptr_t)(lds) + (uint32_t)(px * 160 + (pp >> 1) * 16 + (pp & 1) * 8);
            s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(addr));
            for (int e = 0; e < 4; ++e) c[lane * 8 + 4 * s + e] = bf16_bits_to_f32((uint32_t)(uint16_t)r[e]);
        }
    }
}

extern "C" int mis_probe_mfma(int which, const float* a, const float* b, float* c, void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    MIS_REQUIRE(which >= 0 && which <= 2 && a && c, MIS_EINVAL, "probe: bad argument");
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), which, a, b, c);
    MIS_LAUNCH_CHECK("probe");
    return MIS_OK;
}
