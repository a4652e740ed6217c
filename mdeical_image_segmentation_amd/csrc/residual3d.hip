// Small HBM-bound pieces of the residual 3-D U-Net (reference model/unet3d/buildingblocks.py:255-325 `ResNetBlock`, model.py:197-232):
//   mis_add_act      y = [relu](a + b)                       residual join `out += residual; non_linearity(out)` and the decoder's sum-joining
//   mis_expand1_fwd  y[v][c] = w[c] * x[v] + b[c]            the 1x1x1 conv of the first block (1 input channel -> C) on the raw fp32 volume
//   mis_expand1_bwd  dw[c] = sum_v x[v] * dy[v][c], db[c] = sum_v dy[v][c]   (two-stage, fixed order)
#include "common.hpp"

template <typename T>
__global__ __launch_bounds__(256) void add_act_kernel(const T* __restrict__ a, int a_ld, const T* __restrict__ b, int b_ld, T* __restrict__ y, int y_ld,
                                                      long long npix, int C, int relu) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC;
    const long long total = npix * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long long p = i / nch;
        float fa[EPC], fb[EPC];
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(a + p * a_ld + (size_t)ch * EPC), fa);
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(b + p * b_ld + (size_t)ch * EPC), fb);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float v = fa[e] + fb[e];
            fa[e] = relu ? fmaxf(v, 0.f) : v;
        }
        *reinterpret_cast<u32x4*>(y + p * y_ld + (size_t)ch * EPC) = pack_chunk<T>(fa);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void expand1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, T* __restrict__ y,
                                                          int y_ld, long long nvox, int C) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC;
    const long long total = nvox * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long long v = i / nch;
        const float xv = x[v];
        float o[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = fmaf(w[ch * EPC + e], xv, b[ch * EPC + e]);
        *reinterpret_cast<u32x4*>(y + v * y_ld + (size_t)ch * EPC) = pack_chunk<T>(o);
    }
}

constexpr int EX_BLOCKS = 512;
// part[block][2][C]: per-block partial sums of x*dy and dy.  256 threads = (256 / nch) voxel rows x nch channel chunks
template <typename T>
__global__ __launch_bounds__(256) void expand1_bwd_kernel(const float* __restrict__ x, const T* __restrict__ dy, int dy_ld, long long nvox, int C,
                                                          float* __restrict__ part) {
    constexpr int EPC = Tr<T>::EPC;
    __shared__ float red[2][256 * 8];
    const int nch = C / EPC;                     // <= 64
    const int rows = 256 / nch;
    const int tid = threadIdx.x, ch = tid % nch, r = tid / nch;
    float s1[EPC], s0[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s0[e] = 0.f;
    if (r < rows)
        for (long long v = (long long)blockIdx.x * rows + r; v < nvox; v += (long long)gridDim.x * rows) {
            float g[EPC];
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(dy + v * dy_ld + (size_t)ch * EPC), g);
            const float xv = x[v];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                s1[e] = fmaf(xv, g[e], s1[e]);
                s0[e] += g[e];
            }
        }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        red[0][tid * EPC + e] = (r < rows) ? s1[e] : 0.f;
        red[1][tid * EPC + e] = (r < rows) ? s0[e] : 0.f;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float a1 = 0.f, a0 = 0.f;
        const int cch = c / EPC, ce = c % EPC;
        for (int rr = 0; rr < rows; ++rr) {
            a1 += red[0][(rr * nch + cch) * EPC + ce];
            a0 += red[1][(rr * nch + cch) * EPC + ce];
        }
        part[((size_t)blockIdx.x * 2 + 0) * C + c] = a1;
        part[((size_t)blockIdx.x * 2 + 1) * C + c] = a0;
    }
}

__global__ void expand1_bwd_reduce_kernel(const float* __restrict__ part, int nblocks, int C, float* __restrict__ dw, float* __restrict__ db) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a1 = 0.0, a0 = 0.0;
    for (int b = 0; b < nblocks; ++b) {
        a1 += (double)part[((size_t)b * 2 + 0) * C + c];
        a0 += (double)part[((size_t)b * 2 + 1) * C + c];
    }
    dw[c] = (float)a1;
    db[c] = (float)a0;
}

static unsigned r3_grid(long long total) {
    long long b = (total + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (unsigned)b;
}

extern "C" int mis_add_act(int dtype, const void* a, int a_ld, const void* b, int b_ld, void* y, int y_ld, long long npix, int C, int relu, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "add_act: bad dtype %d", dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(a && b && y && npix > 0 && C > 0 && C % EPC == 0 && a_ld % EPC == 0 && b_ld % EPC == 0 && y_ld % EPC == 0, MIS_EINVAL, "add_act: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = r3_grid(npix * (C / EPC));
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(add_act_kernel<__bf16>, dim3(g), dim3(256), 0, s, (const __bf16*)a, a_ld, (const __bf16*)b, b_ld, (__bf16*)y, y_ld, npix, C, relu);
    else
        hipLaunchKernelGGL(add_act_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)a, a_ld, (const float*)b, b_ld, (float*)y, y_ld, npix, C, relu);
    MIS_LAUNCH_CHECK("add_act");
    return MIS_OK;
}

extern "C" int mis_expand1_fwd(int dtype, const float* x, const float* w, const float* b, void* y, int y_ld, long long nvox, int C, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "expand1_fwd: bad dtype %d", dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(x && w && b && y && nvox > 0 && C > 0 && C % EPC == 0 && y_ld % EPC == 0 && y_ld >= C, MIS_EINVAL, "expand1_fwd: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = r3_grid(nvox * (C / EPC));
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(expand1_fwd_kernel<__bf16>, dim3(g), dim3(256), 0, s, x, w, b, (__bf16*)y, y_ld, nvox, C);
    else
        hipLaunchKernelGGL(expand1_fwd_kernel<float>, dim3(g), dim3(256), 0, s, x, w, b, (float*)y, y_ld, nvox, C);
    MIS_LAUNCH_CHECK("expand1_fwd");
    return MIS_OK;
}

extern "C" size_t mis_expand1_bwd_workspace_bytes(int C) { return (size_t)EX_BLOCKS * 2 * C * sizeof(float); }

extern "C" int mis_expand1_bwd(int dtype, const float* x, const void* dy, int dy_ld, long long nvox, int C, float* workspace, float* dw, float* db,
                               void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "expand1_bwd: bad dtype %d", dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(x && dy && workspace && dw && db && nvox > 0, MIS_EINVAL, "expand1_bwd: bad argument");
    MIS_REQUIRE(C > 0 && C % EPC == 0 && C / EPC <= 64 && 256 % (C / EPC) == 0 && dy_ld % EPC == 0, MIS_EUNSUPPORTED, "expand1_bwd: C %d", C);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rows = 256 / (C / EPC);
    long long blocks = (nvox + rows - 1) / rows;
    if (blocks > EX_BLOCKS) blocks = EX_BLOCKS;
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(expand1_bwd_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, x, (const __bf16*)dy, dy_ld, nvox, C, workspace);
    else
        hipLaunchKernelGGL(expand1_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, x, (const float*)dy, dy_ld, nvox, C, workspace);
    hipLaunchKernelGGL(expand1_bwd_reduce_kernel, dim3((C + 63) / 64), dim3(64), 0, s, (const float*)workspace, (int)blocks, C, dw, db);
    MIS_LAUNCH_CHECK("expand1_bwd");
    return MIS_OK;
}
