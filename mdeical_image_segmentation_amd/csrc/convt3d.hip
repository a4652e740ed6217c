// ConvTranspose3d(k3, s2, p1, no bias) + nearest resize to the encoder grid: the reference's optional 'deconv' upsampling
// (model/unet3d/buildingblocks.py:676-728: TransposeConvUpsampling.Upsample = conv_transposed, then F.interpolate(size)) for gfx950.
//
// The contraction runs on the MFMA GEMM kernels as a 1x1x1 "convolution" with 27*C output columns per input voxel
//   cols[i][k*C + c] = sum_ci x[i][ci] * W[ci][c][k]                      (mis_conv_igemm, ksize 1)
// and this file holds the two index-gather passes around it (HBM-bound, one 16-byte channel chunk per thread):
//   forward  mis_convt3_col2im : u[o][c] = ct[src(o)][c],  ct[p] = sum_{(i,k): 2i-1+k = p} cols[i][k]   (<= 8 terms, fixed order)
//            with src(o) = (o == 0 ? 0 : o - 1): nearest resize of the 2n-1 transposed-conv grid to the 2n encoder grid
//   backward mis_convt3_im2col : gcols[i][k][c] = gct[2i-1+k] (0 outside), gct[p] = sum_{o: src(o) = p} gu[o]   (<= 8 terms)
// after which dX = gcols x W^T (mis_conv_igemm, ksize 1) and dW = x^T x gcols (mis_wgrad, ksize 1).
#include "common.hpp"

// per axis: the (input index, tap) pairs that reach transposed-conv output p:  p = 2i - 1 + k
__device__ __forceinline__ int ct_pairs(int p, int* ii, int* kk) {
    if ((p & 1) == 0) {
        ii[0] = p >> 1;
        kk[0] = 1;
        return 1;
    }
    ii[0] = (p + 1) >> 1;
    kk[0] = 0;
    ii[1] = (p - 1) >> 1;
    kk[1] = 2;
    return 2;
}

template <typename T>
__global__ __launch_bounds__(256) void convt3_col2im_kernel(const T* __restrict__ cols, T* __restrict__ u, int u_ld, int N, int d, int h, int w, int C) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC;
    const int OD = 2 * d, OH = 2 * h, OW = 2 * w;
    const long long total = (long long)N * OD * OH * OW * nch;
    const size_t cld = (size_t)27 * C;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int ch = (int)(t % nch);
        long long r = t / nch;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        r /= OH;
        const int oz = (int)(r % OD);
        const int n = (int)(r / OD);
        int iz[2], kz[2], iy[2], ky[2], ix[2], kx[2];
        const int nz = ct_pairs(oz == 0 ? 0 : oz - 1, iz, kz);
        const int ny = ct_pairs(oy == 0 ? 0 : oy - 1, iy, ky);
        const int nx = ct_pairs(ox == 0 ? 0 : ox - 1, ix, kx);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        for (int a = 0; a < nz; ++a)
            for (int b = 0; b < ny; ++b)
                for (int c = 0; c < nx; ++c) {
                    const size_t vox = (((size_t)n * d + iz[a]) * h + iy[b]) * w + ix[c];
                    const int tap = (kz[a] * 3 + ky[b]) * 3 + kx[c];
                    float f[EPC];
                    unpack_chunk<T>(*reinterpret_cast<const u32x4*>(cols + vox * cld + (size_t)tap * C + (size_t)ch * EPC), f);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) acc[e] += f[e];
                }
        const size_t opix = (((size_t)n * OD + oz) * OH + oy) * OW + ox;
        *reinterpret_cast<u32x4*>(u + opix * u_ld + (size_t)ch * EPC) = pack_chunk<T>(acc);
    }
}

// per axis: the resized-grid positions o that read transposed-conv position p (src(o) = p)
__device__ __forceinline__ int rs_sources(int p, int* oo) {
    if (p == 0) {
        oo[0] = 0;
        oo[1] = 1;
        return 2;
    }
    oo[0] = p + 1;
    return 1;
}

template <typename T>
__global__ __launch_bounds__(256) void convt3_im2col_kernel(const T* __restrict__ gu, int gu_ld, T* __restrict__ gcols, int N, int d, int h, int w, int C) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC;
    const int OD = 2 * d, OH = 2 * h, OW = 2 * w;
    const long long total = (long long)N * d * h * w * 27 * nch;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int ch = (int)(t % nch);
        long long r = t / nch;
        const int tap = (int)(r % 27);
        r /= 27;
        const int x = (int)(r % w);
        r /= w;
        const int y = (int)(r % h);
        r /= h;
        const int z = (int)(r % d);
        const int n = (int)(r / d);
        const int pz = 2 * z - 1 + tap / 9, py = 2 * y - 1 + (tap / 3) % 3, px = 2 * x - 1 + tap % 3;
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        if (pz >= 0 && pz <= OD - 2 && py >= 0 && py <= OH - 2 && px >= 0 && px <= OW - 2) {
            int oz[2], oy[2], ox[2];
            const int nz = rs_sources(pz, oz), ny = rs_sources(py, oy), nx = rs_sources(px, ox);
            for (int a = 0; a < nz; ++a)
                for (int b = 0; b < ny; ++b)
                    for (int c = 0; c < nx; ++c) {
                        const size_t opix = (((size_t)n * OD + oz[a]) * OH + oy[b]) * OW + ox[c];
                        float f[EPC];
                        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(gu + opix * gu_ld + (size_t)ch * EPC), f);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) acc[e] += f[e];
                    }
        }
        const size_t vox = (((size_t)n * d + z) * h + y) * w + x;
        *reinterpret_cast<u32x4*>(gcols + vox * (size_t)27 * C + (size_t)tap * C + (size_t)ch * EPC) = pack_chunk<T>(acc);
    }
}

static int convt3_check(const char* what, int dtype, const void* a, const void* b, int ld, int N, int d, int h, int w, int C) {
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "%s: bad dtype %d", what, dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(a && b && a != b, MIS_EINVAL, "%s: null / aliased pointer", what);
    MIS_REQUIRE(N > 0 && d > 0 && h > 0 && w > 0 && C > 0 && C % EPC == 0 && ld >= C && ld % EPC == 0, MIS_EINVAL, "%s: sizes / alignment", what);
    return MIS_OK;
}

extern "C" int mis_convt3_col2im(int dtype, const void* cols, void* u, int u_ld, int N, int d, int h, int w, int C, void* stream) {
    (void)hipGetLastError();
    if (int rc = convt3_check("convt3_col2im", dtype, cols, u, u_ld, N, d, h, w, C)) return rc;
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    long long blocks = ((long long)N * d * h * w * 8 * (C / EPC) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(convt3_col2im_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)cols, (__bf16*)u, u_ld, N, d, h, w, C);
    else
        hipLaunchKernelGGL(convt3_col2im_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)cols, (float*)u, u_ld, N, d, h, w, C);
    MIS_LAUNCH_CHECK("convt3_col2im");
    return MIS_OK;
}

extern "C" int mis_convt3_im2col(int dtype, const void* gu, int gu_ld, void* gcols, int N, int d, int h, int w, int C, void* stream) {
    (void)hipGetLastError();
    if (int rc = convt3_check("convt3_im2col", dtype, gu, gcols, gu_ld, N, d, h, w, C)) return rc;
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    long long blocks = ((long long)N * d * h * w * 27 * (C / EPC) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(convt3_im2col_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)gu, gu_ld, (__bf16*)gcols, N, d, h, w, C);
    else
        hipLaunchKernelGGL(convt3_im2col_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)gu, gu_ld, (float*)gcols, N, d, h, w, C);
    MIS_LAUNCH_CHECK("convt3_im2col");
    return MIS_OK;
}
