// On-device 3-D augmentation (gfx950): the geometric transforms of the reference's numpy/scipy pipeline as exact index
// gathers, and the intensity transforms as one elementwise pass.  HBM-bound, 4/8-byte elements, one thread per voxel.
//
// Reference (augment/unet3d_augment/transforms.py): RandomFlip :25-50 (np.flip per axis), RandomRotate90 :53-80
// (np.rot90(m, k, axes=(1,2))), RandomRotate :83-112 (scipy.ndimage.rotate(reshape=False, order=0, mode='reflect'):
// per-plane affine_transform; input coordinate = ((0 + i*m[d][0]) + j*m[d][1]) + offset[d] in double without fused
// multiply-add, reflect about the half-sample edges, round half up, reflect the index), RandomContrast :115-133,
// AdditiveGaussianNoise :608-619, Standardize :495-523.  The random PARAMETERS are drawn on the host from the same numpy
// RandomState streams as the reference; only the Gaussian noise FIELD comes from an on-device counter-based generator.
#include <math.h>

#include "common.hpp"

// numpy / scipy evaluate these expressions without fused multiply-add: keep hipcc from contracting a*b+c in this file
#pragma clang fp contract(off)

template <typename E>
__global__ __launch_bounds__(256) void aug_flip_rot90_kernel(const E* __restrict__ src, E* __restrict__ dst, long long nvol, int D, int H, int W,
                                                             int flipmask, int k) {
    // src (nvol, D, H, W) -> flip along the axes in flipmask (bit0 = D, bit1 = H, bit2 = W) -> rot90 k times in the (H, W) plane
    const int OH = (k & 1) ? W : H, OW = (k & 1) ? H : W;
    const long long per = (long long)D * OH * OW, total = nvol * per;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / per;
        long long r = i - v * per;
        const int x = (int)(r % OW);
        r /= OW;
        const int y = (int)(r % OH);
        int z = (int)(r / OH);
        // inverse rot90: out = np.rot90(in, k, axes=(1,2))
        int iy, ix;
        switch (k & 3) {
            case 0: iy = y; ix = x; break;
            case 1: iy = x; ix = W - 1 - y; break;            // out[y][x] = in[x][W-1-y]
            case 2: iy = H - 1 - y; ix = W - 1 - x; break;
            default: iy = H - 1 - x; ix = y; break;           // k = 3: out[y][x] = in[H-1-x][y]
        }
        if (flipmask & 1) z = D - 1 - z;
        if (flipmask & 2) iy = H - 1 - iy;
        if (flipmask & 4) ix = W - 1 - ix;
        dst[i] = src[((v * D + z) * H + iy) * W + ix];
    }
}

__device__ __forceinline__ double refl_coord(double x, int n) {
    if (x < 0) {
        const int sz2 = 2 * n;
        if (x < -sz2) x = sz2 * (double)(long long)(-x / sz2) + x;
        x = x < -n ? x + sz2 : ((x > -1e-15 ? 1e-15 : -x) - 1.0);
    } else if (x > n - 1) {
        const int sz2 = 2 * n;
        x -= sz2 * (double)(long long)(x / sz2);
        if (x >= n) x = sz2 - x - 1;
    }
    return x;
}
__device__ __forceinline__ int refl_idx(long long k, int n) {
    // half-sample symmetric ('reflect') extension d c b a | a b c d | d c b a, periodic with 2n: valid for ANY k (the Gaussian of
    // ElasticDeformation reaches hundreds of samples beyond a 10..128-voxel line)
    const long long sz2 = 2LL * n;
    long long m = k % sz2;
    if (m < 0) m += sz2;
    return (int)(m < n ? m : sz2 - 1 - m);
}

struct RotArgs {
    double m00, m01, m10, m11, off0, off1;
};

template <typename E>
__global__ __launch_bounds__(256) void aug_rotate0_kernel(const E* __restrict__ src, E* __restrict__ dst, long long nvol, int D, int H, int W, int a0,
                                                          int a1, RotArgs ra) {
    const long long per = (long long)D * H * W, total = nvol * per;
    const int dims[3] = {D, H, W};
    const int n0 = dims[a0], n1 = dims[a1];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / per;
        long long r = i - v * per;
        int c[3];
        c[2] = (int)(r % W);
        r /= W;
        c[1] = (int)(r % H);
        c[0] = (int)(r / H);
        const double o0 = (double)c[a0], o1 = (double)c[a1];
        // same operation order as scipy's C loop, no FMA contraction
        double x0 = __dadd_rn(__dadd_rn(__dadd_rn(0.0, __dmul_rn(o0, ra.m00)), __dmul_rn(o1, ra.m01)), ra.off0);
        double x1 = __dadd_rn(__dadd_rn(__dadd_rn(0.0, __dmul_rn(o0, ra.m10)), __dmul_rn(o1, ra.m11)), ra.off1);
        x0 = refl_coord(x0, n0);
        x1 = refl_coord(x1, n1);
        c[a0] = refl_idx((long long)floor(x0 + 0.5), n0);
        c[a1] = refl_idx((long long)floor(x1 + 0.5), n1);
        dst[i] = src[((v * D + c[0]) * H + c[1]) * W + c[2]];
    }
}

// ---- scipy.ndimage's other boundary modes for order 0 (RandomRotate(mode=...), reference augment/unet3d_augment/transforms.py:83-112; round 4) ----
// mode: 0 'reflect' (= 'grid-mirror'), 1 'constant', 2 'nearest', 3 'mirror', 4 'wrap', 5 'grid-wrap', 6 'grid-constant'.  The double coordinate goes through scipy's
// map_coordinate (ni_interpolation.c), is rounded (floor(x + 0.5)), and a rounded index that still falls outside the line is mapped with the same extension
// ('constant' / 'grid-constant': the output is cval); oracle/augment_oracle.py::rotate0_modes restates it in numpy, bit-identical to scipy for every mode.
__device__ __forceinline__ double map_coord_mode(double x, int n, int mode) {
    if (mode == 6) return x;
    if (x < 0) {
        switch (mode) {
            case 3:
                if (n <= 1) return 0.0;
                {
                    const int sz2 = 2 * n - 2;
                    x = sz2 * (double)(long long)(-x / sz2) + x;
                    return x <= 1 - n ? x + sz2 : -x;
                }
            case 0:
                if (n <= 1) return 0.0;
                {
                    const int sz2 = 2 * n;
                    if (x < -sz2) x = sz2 * (double)(long long)(-x / sz2) + x;
                    return x < -n ? x + sz2 : ((x > -1e-15 ? 1e-15 : -x) - 1.0);
                }
            case 4:
                if (n <= 1) return 0.0;
                return x + (double)(n - 1) * ((double)(long long)(-x / (n - 1)) + 1.0);
            case 5:
                if (n <= 1) return 0.0;
                return x + (double)n * ((double)(long long)((-1.0 - x) / n) + 1.0);
            case 2: return 0.0;
            default: return -1.0;          // constant
        }
    }
    if (x > n - 1) {
        switch (mode) {
            case 3:
                if (n <= 1) return 0.0;
                {
                    const int sz2 = 2 * n - 2;
                    x -= sz2 * (double)(long long)(x / sz2);
                    return x >= n ? sz2 - x : x;
                }
            case 0:
                if (n <= 1) return 0.0;
                {
                    const int sz2 = 2 * n;
                    x -= sz2 * (double)(long long)(x / sz2);
                    return x >= n ? sz2 - x - 1 : x;
                }
            case 4:
                if (n <= 1) return 0.0;
                return x - (double)(n - 1) * (double)(long long)(x / (n - 1));
            case 5:
                if (n <= 1) return 0.0;
                return x - (double)n * (double)(long long)(x / n);
            case 2: return (double)(n - 1);
            default: return -1.0;
        }
    }
    return x;
}
// rounded index -> a sample of the line, or `outside` (constant modes)
__device__ __forceinline__ int idx_mode(long long k, int n, int mode, bool& outside) {
    if (k >= 0 && k < n) return (int)k;
    switch (mode) {
        case 0: return refl_idx(k, n);
        case 3: {
            if (n == 1) return 0;
            const long long p = 2LL * n - 2;
            long long m = k % p;
            if (m < 0) m += p;
            return (int)(m < n ? m : p - m);
        }
        case 4:
        case 5: {
            long long m = k % n;
            if (m < 0) m += n;
            return (int)m;
        }
        case 2: return k < 0 ? 0 : n - 1;
        default: outside = true; return 0;
    }
}

template <typename E>
__global__ __launch_bounds__(256) void aug_rotate0_mode_kernel(const E* __restrict__ src, E* __restrict__ dst, long long nvol, int D, int H, int W, int a0,
                                                               int a1, RotArgs ra, int mode, E cval) {
    const long long per = (long long)D * H * W, total = nvol * per;
    const int dims[3] = {D, H, W};
    const int n0 = dims[a0], n1 = dims[a1];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / per;
        long long r = i - v * per;
        int c[3];
        c[2] = (int)(r % W);
        r /= W;
        c[1] = (int)(r % H);
        c[0] = (int)(r / H);
        const double o0 = (double)c[a0], o1 = (double)c[a1];
        double x0 = __dadd_rn(__dadd_rn(__dadd_rn(0.0, __dmul_rn(o0, ra.m00)), __dmul_rn(o1, ra.m01)), ra.off0);
        double x1 = __dadd_rn(__dadd_rn(__dadd_rn(0.0, __dmul_rn(o0, ra.m10)), __dmul_rn(o1, ra.m11)), ra.off1);
        x0 = map_coord_mode(x0, n0, mode);
        x1 = map_coord_mode(x1, n1, mode);
        bool outside = mode == 1 && (x0 <= -1.0 || x1 <= -1.0);
        c[a0] = idx_mode((long long)floor(x0 + 0.5), n0, mode, outside);
        c[a1] = idx_mode((long long)floor(x1 + 0.5), n1, mode, outside);
        dst[i] = outside ? cval : src[((v * D + c[0]) * H + c[1]) * W + c[2]];
    }
}

static unsigned aug_grid(long long total) {
    long long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

extern "C" int mis_aug_flip_rot90(const void* src, void* dst, long long nvol, int D, int H, int W, int flipmask, int k, int elem_size,
                                  void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(src && dst && src != dst && nvol > 0 && D > 0 && H > 0 && W > 0, MIS_EINVAL, "aug_flip_rot90: bad argument");
    MIS_REQUIRE(elem_size == 4 || elem_size == 8, MIS_EUNSUPPORTED, "aug_flip_rot90: element size %d", elem_size);
    MIS_REQUIRE(k >= 0 && k <= 3 && flipmask >= 0 && flipmask <= 7, MIS_EINVAL, "aug_flip_rot90: k / flipmask");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = aug_grid(nvol * D * H * W);
    if (elem_size == 4)
        hipLaunchKernelGGL(aug_flip_rot90_kernel<uint32_t>, dim3(g), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, nvol, D, H, W, flipmask, k);
    else
        hipLaunchKernelGGL(aug_flip_rot90_kernel<uint64_t>, dim3(g), dim3(256), 0, s, (const uint64_t*)src, (uint64_t*)dst, nvol, D, H, W, flipmask, k);
    MIS_LAUNCH_CHECK("aug_flip_rot90");
    return MIS_OK;
}

// CropToFixed (transforms.py:194-247): out[v][z][i][j] = src[v][z][r(y0 + i)][r(x0 + j)], r = numpy 'reflect' (edge not repeated, periodic):
// a plain window when the crop is smaller than the plane, the whole plane mirror-padded to the crop size otherwise
template <typename E>
__global__ __launch_bounds__(256) void aug_crop_reflect_kernel(const E* __restrict__ src, E* __restrict__ dst, long long nslices, int H, int W, int y0, int x0,
                                                               int CH, int CW) {
    const long long total = nslices * CH * CW;
    const int ph = H > 1 ? 2 * H - 2 : 1, pw = W > 1 ? 2 * W - 2 : 1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int j = (int)(i % CW);
        long long r = i / CW;
        const int ii = (int)(r % CH);
        const long long sl = r / CH;
        int y = (y0 + ii) % ph, x = (x0 + j) % pw;
        if (y < 0) y += ph;
        if (x < 0) x += pw;
        if (y >= H) y = ph - y;
        if (x >= W) x = pw - x;
        dst[i] = src[(sl * H + y) * W + x];
    }
}

extern "C" int mis_aug_crop_reflect(const void* src, void* dst, long long nslices, int H, int W, int y0, int x0, int CH, int CW, int elem_size,
                                    void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(src && dst && src != dst && nslices > 0 && H > 0 && W > 0 && CH > 0 && CW > 0, MIS_EINVAL, "aug_crop_reflect: bad argument");
    MIS_REQUIRE(elem_size == 4 || elem_size == 8, MIS_EUNSUPPORTED, "aug_crop_reflect: element size %d", elem_size);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = aug_grid(nslices * CH * CW);
    if (elem_size == 4)
        hipLaunchKernelGGL(aug_crop_reflect_kernel<uint32_t>, dim3(g), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, nslices, H, W, y0, x0, CH, CW);
    else
        hipLaunchKernelGGL(aug_crop_reflect_kernel<uint64_t>, dim3(g), dim3(256), 0, s, (const uint64_t*)src, (uint64_t*)dst, nslices, H, W, y0, x0, CH, CW);
    MIS_LAUNCH_CHECK("aug_crop_reflect");
    return MIS_OK;
}

extern "C" int mis_aug_rotate0(const void* src, void* dst, long long nvol, int D, int H, int W, int a0, int a1, const double* m4,
                               const double* off2, int elem_size, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(src && dst && src != dst && m4 && off2 && nvol > 0 && D > 0 && H > 0 && W > 0, MIS_EINVAL, "aug_rotate0: bad argument");
    MIS_REQUIRE(a0 >= 0 && a0 < a1 && a1 <= 2, MIS_EINVAL, "aug_rotate0: axes must be sorted and distinct");
    MIS_REQUIRE(elem_size == 4 || elem_size == 8, MIS_EUNSUPPORTED, "aug_rotate0: element size %d", elem_size);
    RotArgs ra{m4[0], m4[1], m4[2], m4[3], off2[0], off2[1]};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = aug_grid(nvol * D * H * W);
    if (elem_size == 4)
        hipLaunchKernelGGL(aug_rotate0_kernel<uint32_t>, dim3(g), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, nvol, D, H, W, a0, a1, ra);
    else
        hipLaunchKernelGGL(aug_rotate0_kernel<uint64_t>, dim3(g), dim3(256), 0, s, (const uint64_t*)src, (uint64_t*)dst, nvol, D, H, W, a0, a1, ra);
    MIS_LAUNCH_CHECK("aug_rotate0");
    return MIS_OK;
}

extern "C" int mis_aug_rotate0_mode(const void* src, void* dst, long long nvol, int D, int H, int W, int a0, int a1, const double* m4, const double* off2,
                                    int elem_size, int mode, unsigned long long cval_bits, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(src && dst && src != dst && m4 && off2 && nvol > 0 && D > 0 && H > 0 && W > 0, MIS_EINVAL, "aug_rotate0_mode: bad argument");
    MIS_REQUIRE(a0 >= 0 && a0 < a1 && a1 <= 2, MIS_EINVAL, "aug_rotate0_mode: axes must be sorted and distinct");
    MIS_REQUIRE(elem_size == 4 || elem_size == 8, MIS_EUNSUPPORTED, "aug_rotate0_mode: element size %d", elem_size);
    MIS_REQUIRE(mode >= 0 && mode <= 6, MIS_EINVAL, "aug_rotate0_mode: mode %d (0 reflect, 1 constant, 2 nearest, 3 mirror, 4 wrap, 5 grid-wrap, 6 grid-constant)", mode);
    RotArgs ra{m4[0], m4[1], m4[2], m4[3], off2[0], off2[1]};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = aug_grid(nvol * D * H * W);
    if (elem_size == 4)
        hipLaunchKernelGGL(aug_rotate0_mode_kernel<uint32_t>, dim3(g), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, nvol, D, H, W, a0, a1, ra, mode,
                           (uint32_t)cval_bits);
    else
        hipLaunchKernelGGL(aug_rotate0_mode_kernel<uint64_t>, dim3(g), dim3(256), 0, s, (const uint64_t*)src, (uint64_t*)dst, nvol, D, H, W, a0, a1, ra, mode,
                           (uint64_t)cval_bits);
    MIS_LAUNCH_CHECK("aug_rotate0_mode");
    return MIS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// RandomRotate with order=3 (the reference's raw-volume setting in its 3-D configs; class default is 0):
// scipy.ndimage.rotate(reshape=False, order=3, mode='reflect') = per rotation plane
//   1. spline_filter to float64 coefficients: along plane axis 0, then plane axis 1 (scipy ni_splines.c, cubic pole sqrt(3)-2,
//      gain (1-z)(1-1/z), exact 'reflect' initialisation of the causal recursion, anti-causal c[n-1] *= z/(z-1))
//   2. at each output voxel the rotated coordinate (same unfused arithmetic as order 0), reflected, start = floor(x)-1, the
//      4x4 cubic B-spline weights, support indices reflected, t += (coef * w_row) * w_col in row-major tap order, cast to float.
// One thread per line for the recursive filter (fp64, sequential along the line by nature), one thread per voxel for step 2.
// ---------------------------------------------------------------------------------------------------------
template <typename Src>
__global__ __launch_bounds__(256) void aug_spline3_filter_kernel(const Src* __restrict__ src, double* __restrict__ coef, long long nvol, int D, int H,
                                                                 int W, int ax, double z, double z_n) {
    const int dims[3] = {D, H, W};
    const long long strides[3] = {(long long)H * W, (long long)W, 1};
    const int n = dims[ax];
    const long long st = strides[ax];
    const int o1 = ax == 0 ? 1 : 0, o2 = ax == 2 ? 1 : 2;       // the two other axes
    const long long nlines = nvol * dims[o1] * dims[o2];
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    for (long long l = (long long)blockIdx.x * 256 + threadIdx.x; l < nlines; l += (long long)gridDim.x * 256) {
        const int i2 = (int)(l % dims[o2]);
        const long long t = l / dims[o2];
        const int i1 = (int)(t % dims[o1]);
        const long long v = t / dims[o1];
        const long long base = v * D * H * W + i1 * strides[o1] + i2 * strides[o2];
        const Src* s = src + base;
        double* c = coef + base;
        if (n == 1) {                                  // a line of one sample is its own coefficient (scipy returns it unchanged)
            c[0] = (double)s[0];
            continue;
        }
        // causal initialisation over the half-sample-symmetric extension
        const double c0 = (double)s[0] * gain;
        double acc = c0 + z_n * ((double)s[(n - 1) * st] * gain);
        double z_i = z;
        for (int i = 1; i < n; ++i) {
            acc = acc + z_i * ((double)s[i * st] * gain + z_n * ((double)s[(n - 1 - i) * st] * gain));
            z_i *= z;
        }
        acc = acc * (z / (1.0 - z_n * z_n));
        double prev = acc + c0;
        c[0] = prev;
        for (int i = 1; i < n; ++i) {
            prev = (double)s[i * st] * gain + z * prev;
            c[i * st] = prev;
        }
        prev = prev * (z / (z - 1.0));
        c[(n - 1) * st] = prev;
        for (int i = n - 2; i >= 0; --i) {
            prev = z * (prev - c[i * st]);
            c[i * st] = prev;
        }
    }
}

__device__ __forceinline__ void spline3_weights(double x, long long& start, double* w) {
    const double f = floor(x);
    const double y = x - f, zc = 1.0 - y;
    w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
    w[2] = (zc * zc * (zc - 2.0) * 3.0 + 4.0) / 6.0;
    w[0] = zc * zc * zc / 6.0;
    w[3] = 1.0 - w[0] - w[1] - w[2];
    start = (long long)f - 1;
}

__global__ __launch_bounds__(256) void aug_rotate3_kernel(const double* __restrict__ coef, float* __restrict__ dst, long long nvol, int D, int H, int W,
                                                          int a0, int a1, RotArgs ra) {
    const long long per = (long long)D * H * W, total = nvol * per;
    const int dims[3] = {D, H, W};
    const long long strides[3] = {(long long)H * W, (long long)W, 1};
    const int n0 = dims[a0], n1 = dims[a1];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / per;
        long long r = i - v * per;
        int c[3];
        c[2] = (int)(r % W);
        r /= W;
        c[1] = (int)(r % H);
        c[0] = (int)(r / H);
        const double o0 = (double)c[a0], o1 = (double)c[a1];
        double x0 = __dadd_rn(__dadd_rn(__dadd_rn(0.0, __dmul_rn(o0, ra.m00)), __dmul_rn(o1, ra.m01)), ra.off0);
        double x1 = __dadd_rn(__dadd_rn(__dadd_rn(0.0, __dmul_rn(o0, ra.m10)), __dmul_rn(o1, ra.m11)), ra.off1);
        x0 = refl_coord(x0, n0);
        x1 = refl_coord(x1, n1);
        long long s0, s1;
        double w0[4], w1[4];
        spline3_weights(x0, s0, w0);
        spline3_weights(x1, s1, w1);
        long long k1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) k1[b] = (long long)refl_idx(s1 + b, n1) * strides[a1];
        const long long rest = v * per + (long long)c[0] * strides[0] + (long long)c[1] * strides[1] + (long long)c[2] * strides[2] -
                               (long long)c[a0] * strides[a0] - (long long)c[a1] * strides[a1];
        double t = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const long long rowoff = rest + (long long)refl_idx(s0 + a, n0) * strides[a0];
#pragma unroll
            for (int b = 0; b < 4; ++b) t += (coef[rowoff + k1[b]] * w0[a]) * w1[b];
        }
        dst[i] = (float)t;
    }
}

// ---- spline orders 1, 2, 4, 5 (scipy.ndimage.rotate(order=...), mode='reflect'; order 3 keeps its own kernels above) --------------------------------------------
// Prefilter: scipy's spline_filter1d applies the total gain prod (1 - z)(1 - 1/z) once, then for every pole the causal / anticausal recursion with the 'reflect'
// (half-sample symmetric) initialisations used above; poles as in ni_splines.c get_filter_poles.  Weights: the B-spline of the order at the offset from the middle knot
// (even orders: x - floor(x + 0.5), odd: x - floor(x)), the last weight = 1 - sum of the others.  A numpy prototype of exactly this arithmetic agrees with
// scipy.ndimage.map_coordinates to 4e-14 in float64 for all four orders.
struct SplinePoles {
    int n;
    double z[2];
};

template <typename Src>
__global__ __launch_bounds__(256) void aug_spline_filter_kernel(const Src* __restrict__ src, double* __restrict__ coef, long long nvol, int D, int H, int W, int ax,
                                                                SplinePoles sp) {
    const int dims[3] = {D, H, W};
    const long long strides[3] = {(long long)H * W, (long long)W, 1};
    const int n = dims[ax];
    const long long st = strides[ax];
    const int o1 = ax == 0 ? 1 : 0, o2 = ax == 2 ? 1 : 2;
    const long long nlines = nvol * dims[o1] * dims[o2];
    double gain = 1.0;
    for (int p = 0; p < sp.n; ++p) gain *= (1.0 - sp.z[p]) * (1.0 - 1.0 / sp.z[p]);
    for (long long l = (long long)blockIdx.x * 256 + threadIdx.x; l < nlines; l += (long long)gridDim.x * 256) {
        const int i2 = (int)(l % dims[o2]);
        const long long t = l / dims[o2];
        const int i1 = (int)(t % dims[o1]);
        const long long v = t / dims[o1];
        const long long base = v * D * H * W + i1 * strides[o1] + i2 * strides[o2];
        const Src* s = src + base;
        double* c = coef + base;
        if (n == 1) {                                  // a line of one sample is its own coefficient
            c[0] = (double)s[0];
            continue;
        }
        for (int i = 0; i < n; ++i) c[i * st] = (double)s[i * st] * gain;          // (in place when src == coef: read before written, element by element)
        for (int p = 0; p < sp.n; ++p) {
            const double z = sp.z[p], z_n = pow(z, (double)n);
            const double c0 = c[0];
            double acc = c0 + z_n * c[(n - 1) * st];
            double z_i = z;
            for (int i = 1; i < n; ++i) {
                acc = acc + z_i * (c[i * st] + z_n * c[(n - 1 - i) * st]);
                z_i *= z;
            }
            acc = acc * (z / (1.0 - z_n * z_n));
            double prev = acc + c0;
            c[0] = prev;
            for (int i = 1; i < n; ++i) {
                prev = c[i * st] + z * prev;
                c[i * st] = prev;
            }
            prev = prev * (z / (z - 1.0));
            c[(n - 1) * st] = prev;
            for (int i = n - 2; i >= 0; --i) {
                prev = z * (prev - c[i * st]);
                c[i * st] = prev;
            }
        }
    }
}

template <int ORDER> __device__ __forceinline__ void spline_weights(double x, long long& start, double* w) {
    if constexpr (ORDER == 1) {
        const double f = floor(x), t = x - f;
        w[0] = 1.0 - t;
        w[1] = t;
        start = (long long)f;
    } else if constexpr (ORDER == 2) {
        const double f = floor(x + 0.5), t = x - f;
        w[0] = 0.5 * (0.5 - t) * (0.5 - t);
        w[1] = 0.75 - t * t;
        w[2] = 1.0 - w[0] - w[1];
        start = (long long)f - 1;
    } else if constexpr (ORDER == 4) {
        const double f = floor(x + 0.5), t = x - f, t2 = t * t;
        const double h = 0.5 - t;
        w[0] = h * h * h * h / 24.0;
        w[1] = (19.0 - 44.0 * t + 24.0 * t2 + 16.0 * t * t2 - 16.0 * t2 * t2) / 96.0;
        w[2] = (115.0 - 120.0 * t2 + 48.0 * t2 * t2) / 192.0;
        w[3] = (19.0 + 44.0 * t + 24.0 * t2 - 16.0 * t * t2 - 16.0 * t2 * t2) / 96.0;
        w[4] = 1.0 - w[0] - w[1] - w[2] - w[3];
        start = (long long)f - 2;
    } else {
        static_assert(ORDER == 5, "");
        const double f = floor(x), t = x - f, t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
        const double u = 1.0 - t;
        w[0] = u * u * u * u * u / 120.0;
        w[1] = (26.0 - 50.0 * t + 20.0 * t2 + 20.0 * t3 - 20.0 * t4 + 5.0 * t5) / 120.0;
        w[2] = (66.0 - 60.0 * t2 + 30.0 * t4 - 10.0 * t5) / 120.0;
        w[3] = (26.0 + 50.0 * t + 20.0 * t2 - 20.0 * t3 - 20.0 * t4 + 10.0 * t5) / 120.0;
        w[4] = (1.0 + 5.0 * t + 10.0 * t2 + 10.0 * t3 + 5.0 * t4 - 5.0 * t5) / 120.0;
        w[5] = 1.0 - w[0] - w[1] - w[2] - w[3] - w[4];
        start = (long long)f - 2;
    }
}

// coef: float64 spline coefficients (ORDER >= 2) or, for ORDER 1, the fp32 volume itself (CoefT = float)
template <int ORDER, typename CoefT>
__global__ __launch_bounds__(256) void aug_rotate_spline_kernel(const CoefT* __restrict__ coef, float* __restrict__ dst, long long nvol, int D, int H, int W,
                                                                int a0, int a1, RotArgs ra) {
    constexpr int NT = ORDER + 1;
    const long long per = (long long)D * H * W, total = nvol * per;
    const int dims[3] = {D, H, W};
    const long long strides[3] = {(long long)H * W, (long long)W, 1};
    const int n0 = dims[a0], n1 = dims[a1];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / per;
        long long r = i - v * per;
        int c[3];
        c[2] = (int)(r % W);
        r /= W;
        c[1] = (int)(r % H);
        c[0] = (int)(r / H);
        const double o0 = (double)c[a0], o1 = (double)c[a1];
        double x0 = __dadd_rn(__dadd_rn(__dadd_rn(0.0, __dmul_rn(o0, ra.m00)), __dmul_rn(o1, ra.m01)), ra.off0);
        double x1 = __dadd_rn(__dadd_rn(__dadd_rn(0.0, __dmul_rn(o0, ra.m10)), __dmul_rn(o1, ra.m11)), ra.off1);
        x0 = refl_coord(x0, n0);
        x1 = refl_coord(x1, n1);
        long long s0, s1;
        double w0[NT], w1[NT];
        spline_weights<ORDER>(x0, s0, w0);
        spline_weights<ORDER>(x1, s1, w1);
        long long k1[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) k1[b] = (long long)refl_idx(s1 + b, n1) * strides[a1];
        const long long rest = v * per + (long long)c[0] * strides[0] + (long long)c[1] * strides[1] + (long long)c[2] * strides[2] -
                               (long long)c[a0] * strides[a0] - (long long)c[a1] * strides[a1];
        double t = 0.0;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const long long rowoff = rest + (long long)refl_idx(s0 + a, n0) * strides[a0];
#pragma unroll
            for (int b = 0; b < NT; ++b) t += ((double)coef[rowoff + k1[b]] * w0[a]) * w1[b];
        }
        dst[i] = (float)t;
    }
}

extern "C" int mis_aug_rotate3(const float* src, float* dst, double* workspace, long long nvol, int D, int H, int W, int a0, int a1, const double* m4,
                               const double* off2, void* stream);
// scipy.ndimage.rotate(reshape=False, order = 1 .. 5, mode='reflect') of fp32 volumes in the plane of axes (a0, a1); workspace as for order 3 (unused by order 1)
extern "C" int mis_aug_rotate_spline(const float* src, float* dst, double* workspace, long long nvol, int D, int H, int W, int a0, int a1, const double* m4,
                                     const double* off2, int order, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(order >= 1 && order <= 5, MIS_EUNSUPPORTED, "aug_rotate_spline: order %d (1..5; order 0 is mis_aug_rotate0)", order);
    if (order == 3) return mis_aug_rotate3(src, dst, workspace, nvol, D, H, W, a0, a1, m4, off2, stream);
    MIS_REQUIRE(src && dst && src != dst && m4 && off2 && nvol > 0 && D > 0 && H > 0 && W > 0, MIS_EINVAL, "aug_rotate_spline: bad argument");
    MIS_REQUIRE(order == 1 || workspace != nullptr, MIS_EINVAL, "aug_rotate_spline: workspace");
    MIS_REQUIRE(a0 >= 0 && a0 < a1 && a1 <= 2, MIS_EINVAL, "aug_rotate_spline: axes must be sorted and distinct");
    RotArgs ra{m4[0], m4[1], m4[2], m4[3], off2[0], off2[1]};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int dims[3] = {D, H, W};
    const long long vox = nvol * D * H * W;
    if (order == 1) {
        hipLaunchKernelGGL((aug_rotate_spline_kernel<1, float>), dim3(aug_grid(vox)), dim3(256), 0, s, src, dst, nvol, D, H, W, a0, a1, ra);
        MIS_LAUNCH_CHECK("aug_rotate_spline");
        return MIS_OK;
    }
    SplinePoles sp;
    if (order == 2) {
        sp = SplinePoles{1, {sqrt(8.0) - 3.0, 0.0}};
    } else if (order == 4) {
        sp = SplinePoles{2, {sqrt(664.0 - sqrt(438976.0)) + sqrt(304.0) - 19.0, sqrt(664.0 + sqrt(438976.0)) - sqrt(304.0) - 19.0}};
    } else {
        sp = SplinePoles{2, {sqrt(67.5 - sqrt(4436.25)) + sqrt(26.25) - 6.5, sqrt(67.5 + sqrt(4436.25)) - sqrt(26.25) - 6.5}};
    }
    hipLaunchKernelGGL(aug_spline_filter_kernel<float>, dim3(aug_grid(vox / dims[a0])), dim3(256), 0, s, src, workspace, nvol, D, H, W, a0, sp);
    hipLaunchKernelGGL(aug_spline_filter_kernel<double>, dim3(aug_grid(vox / dims[a1])), dim3(256), 0, s, (const double*)workspace, workspace, nvol, D, H, W, a1, sp);
    if (order == 2)
        hipLaunchKernelGGL((aug_rotate_spline_kernel<2, double>), dim3(aug_grid(vox)), dim3(256), 0, s, (const double*)workspace, dst, nvol, D, H, W, a0, a1, ra);
    else if (order == 4)
        hipLaunchKernelGGL((aug_rotate_spline_kernel<4, double>), dim3(aug_grid(vox)), dim3(256), 0, s, (const double*)workspace, dst, nvol, D, H, W, a0, a1, ra);
    else
        hipLaunchKernelGGL((aug_rotate_spline_kernel<5, double>), dim3(aug_grid(vox)), dim3(256), 0, s, (const double*)workspace, dst, nvol, D, H, W, a0, a1, ra);
    MIS_LAUNCH_CHECK("aug_rotate_spline");
    return MIS_OK;
}

extern "C" size_t mis_aug_rotate3_workspace_bytes(long long nvol, int D, int H, int W) {
    return (size_t)nvol * D * H * W * sizeof(double);
}

extern "C" int mis_aug_rotate3(const float* src, float* dst, double* workspace, long long nvol, int D, int H, int W, int a0, int a1,
                               const double* m4, const double* off2, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(src && dst && workspace && src != dst && m4 && off2 && nvol > 0 && D > 0 && H > 0 && W > 0, MIS_EINVAL, "aug_rotate3: bad argument");
    MIS_REQUIRE(a0 >= 0 && a0 < a1 && a1 <= 2, MIS_EINVAL, "aug_rotate3: axes must be sorted and distinct");
    RotArgs ra{m4[0], m4[1], m4[2], m4[3], off2[0], off2[1]};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int dims[3] = {D, H, W};
    const double z = sqrt(3.0) - 2.0;
    const long long vox = nvol * D * H * W;
    hipLaunchKernelGGL(aug_spline3_filter_kernel<float>, dim3(aug_grid(vox / dims[a0])), dim3(256), 0, s, src, workspace, nvol, D, H, W, a0, z,
                       pow(z, (double)dims[a0]));
    hipLaunchKernelGGL(aug_spline3_filter_kernel<double>, dim3(aug_grid(vox / dims[a1])), dim3(256), 0, s, (const double*)workspace, workspace, nvol, D, H,
                       W, a1, z, pow(z, (double)dims[a1]));
    hipLaunchKernelGGL(aug_rotate3_kernel, dim3(aug_grid(vox)), dim3(256), 0, s, (const double*)workspace, dst, nvol, D, H, W, a0, a1, ra);
    MIS_LAUNCH_CHECK("aug_rotate3");
    return MIS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// ElasticDeformation (reference transforms.py:138-191): three smooth random displacement fields
//   d = scipy.ndimage.gaussian_filter(randn(volume), sigma, mode='reflect') * alpha      (float64; fields drawn on the host from the reference's stream)
// and scipy.ndimage.map_coordinates(m, (z + dz, y + dy, x + dx), order, mode='reflect') with order 0 (labels) or 3 (raw).
//   aug_gauss1d_kernel : one axis of the separable Gaussian as scipy's correlate1d evaluates a symmetric kernel,
//                        centre tap first, then (x[l-k] + x[l+k]) * w[k] from the far end inwards, 'reflect' extension (any radius)
//   aug_mapcoord*_kernel: coordinate reflect, order 0 = round-half-up gather, order 3 = 4x4x4 cubic B-spline taps on float64 coefficients
//                        (the prefilter is the per-axis aug_spline3_filter_kernel above, run over all three axes)
// ---------------------------------------------------------------------------------------------------------
// E = double: ElasticDeformation's fields.  E = float: GaussianBlur3D (skimage.filters.gaussian = gaussian_filter on the fp32 volume): scipy runs every axis on a
// double line buffer and rounds the result to the array's type after each axis, which is what the float instantiation does.  NEAREST: mode='nearest' extension.
template <typename E, bool NEAREST>
__global__ __launch_bounds__(256) void aug_gauss1d_kernel(const E* __restrict__ src, E* __restrict__ dst, long long nvol, int D, int H, int W, int ax,
                                                          const double* __restrict__ wgt /*[2r+1], already reversed (symmetric)*/, int radius) {
    const long long per = (long long)D * H * W, total = nvol * per;
    const int dims[3] = {D, H, W};
    const long long strides[3] = {(long long)H * W, (long long)W, 1};
    const int n = dims[ax];
    const long long st = strides[ax];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i % per;
        int c[3];
        c[2] = (int)(r % W);
        c[1] = (int)((r / W) % H);
        c[0] = (int)(r / ((long long)H * W));
        const int l = c[ax];
        const E* line = src + (i - (long long)l * st);
        auto at = [&](int j) -> double {
            const int jj = NEAREST ? (j < 0 ? 0 : (j >= n ? n - 1 : j)) : refl_idx(j, n);
            return (double)line[(long long)jj * st];
        };
        double t = (double)line[(long long)l * st] * wgt[radius];
        for (int k = -radius; k < 0; ++k) t += (at(l + k) + at(l - k)) * wgt[k + radius];
        dst[i] = (E)t;
    }
}

template <typename E>
__global__ __launch_bounds__(256) void aug_mapcoord0_kernel(const E* __restrict__ src, E* __restrict__ dst, long long nvol, int D, int H, int W,
                                                            const double* __restrict__ fz, const double* __restrict__ fy, const double* __restrict__ fx,
                                                            double alpha) {
    const long long per = (long long)D * H * W, total = nvol * per;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / per, r = i - v * per;
        const int x = (int)(r % W), y = (int)((r / W) % H), z = (int)(r / ((long long)H * W));
        const double cz = refl_coord(fz != nullptr ? (double)z + fz[r] * alpha : (double)z, D);
        const double cy = refl_coord((double)y + fy[r] * alpha, H);
        const double cx = refl_coord((double)x + fx[r] * alpha, W);
        const int iz = refl_idx((long long)floor(cz + 0.5), D), iy = refl_idx((long long)floor(cy + 0.5), H), ix = refl_idx((long long)floor(cx + 0.5), W);
        dst[i] = src[((v * D + iz) * H + iy) * W + ix];
    }
}

__global__ __launch_bounds__(256) void aug_mapcoord3_kernel(const double* __restrict__ coef, float* __restrict__ dst, long long nvol, int D, int H, int W,
                                                            const double* __restrict__ fz, const double* __restrict__ fy, const double* __restrict__ fx,
                                                            double alpha) {
    const long long per = (long long)D * H * W, total = nvol * per;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / per, r = i - v * per;
        const int x = (int)(r % W), y = (int)((r / W) % H), z = (int)(r / ((long long)H * W));
        const double cz = refl_coord(fz != nullptr ? (double)z + fz[r] * alpha : (double)z, D);
        const double cy = refl_coord((double)y + fy[r] * alpha, H);
        const double cx = refl_coord((double)x + fx[r] * alpha, W);
        long long sz, sy, sx;
        double wz[4], wy[4], wx[4];
        spline3_weights(cz, sz, wz);
        spline3_weights(cy, sy, wy);
        spline3_weights(cx, sx, wx);
        long long ox[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) ox[c] = refl_idx(sx + c, W);
        const double* cv = coef + v * per;
        double t = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const long long oz = (long long)refl_idx(sz + a, D) * H;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const long long row = (oz + refl_idx(sy + b, H)) * W;
#pragma unroll
                for (int c = 0; c < 4; ++c) t += ((cv[row + ox[c]] * wz[a]) * wy[b]) * wx[c];
            }
        }
        dst[i] = (float)t;
    }
}

extern "C" int mis_aug_gauss1d(const double* src, double* dst, long long nvol, int D, int H, int W, int axis, const double* weights /*device, 2r+1*/,
                               int radius, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(src && dst && src != dst && weights && nvol > 0 && D > 0 && H > 0 && W > 0 && axis >= 0 && axis <= 2 && radius >= 0, MIS_EINVAL,
                "aug_gauss1d: bad argument");
    hipLaunchKernelGGL((aug_gauss1d_kernel<double, false>), dim3(aug_grid(nvol * D * H * W)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, nvol, D,
                       H, W, axis, weights, radius);
    MIS_LAUNCH_CHECK("aug_gauss1d");
    return MIS_OK;
}

// fp32 volumes, mode 0 = 'reflect', 1 = 'nearest' (GaussianBlur3D: skimage.filters.gaussian's default)
extern "C" int mis_aug_gauss1d_f32(const float* src, float* dst, long long nvol, int D, int H, int W, int axis, const double* weights /*device, 2r+1*/, int radius,
                                   int mode, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(src && dst && src != dst && weights && nvol > 0 && D > 0 && H > 0 && W > 0 && axis >= 0 && axis <= 2 && radius >= 0 && (mode == 0 || mode == 1),
                MIS_EINVAL, "aug_gauss1d_f32: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (mode == 1)
        hipLaunchKernelGGL((aug_gauss1d_kernel<float, true>), dim3(aug_grid(nvol * D * H * W)), dim3(256), 0, s, src, dst, nvol, D, H, W, axis, weights, radius);
    else
        hipLaunchKernelGGL((aug_gauss1d_kernel<float, false>), dim3(aug_grid(nvol * D * H * W)), dim3(256), 0, s, src, dst, nvol, D, H, W, axis, weights, radius);
    MIS_LAUNCH_CHECK("aug_gauss1d_f32");
    return MIS_OK;
}

// fields: float64 (D, H, W) displacement fields BEFORE the multiplication by alpha; fz may be NULL (apply_3d=False: no displacement along z).
// order 0: src / dst of 4- or 8-byte elements;  order 3: fp32, workspace = nvol*D*H*W doubles (spline coefficients).
extern "C" int mis_aug_map_coordinates(const void* src, void* dst, double* workspace, long long nvol, int D, int H, int W, const double* fz, const double* fy,
                                       const double* fx, double alpha, int order, int elem_size, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(src && dst && src != dst && fy && fx && nvol > 0 && D > 0 && H > 0 && W > 0, MIS_EINVAL, "aug_map_coordinates: bad argument");
    MIS_REQUIRE(order == 0 || order == 3, MIS_EUNSUPPORTED, "aug_map_coordinates: spline order %d (0 and 3 are built)", order);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long vox = nvol * D * H * W;
    if (order == 0) {
        MIS_REQUIRE(elem_size == 4 || elem_size == 8, MIS_EUNSUPPORTED, "aug_map_coordinates: element size %d", elem_size);
        if (elem_size == 4)
            hipLaunchKernelGGL(aug_mapcoord0_kernel<uint32_t>, dim3(aug_grid(vox)), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, nvol, D, H, W, fz, fy, fx, alpha);
        else
            hipLaunchKernelGGL(aug_mapcoord0_kernel<uint64_t>, dim3(aug_grid(vox)), dim3(256), 0, s, (const uint64_t*)src, (uint64_t*)dst, nvol, D, H, W, fz, fy, fx, alpha);
        MIS_LAUNCH_CHECK("aug_map_coordinates(order 0)");
        return MIS_OK;
    }
    MIS_REQUIRE(workspace != nullptr && elem_size == 4, MIS_EINVAL, "aug_map_coordinates: order 3 needs fp32 data and a workspace");
    const int dims[3] = {D, H, W};
    const double z = sqrt(3.0) - 2.0;
    hipLaunchKernelGGL(aug_spline3_filter_kernel<float>, dim3(aug_grid(vox / dims[0])), dim3(256), 0, s, (const float*)src, workspace, nvol, D, H, W, 0, z,
                       pow(z, (double)dims[0]));
    for (int ax = 1; ax < 3; ++ax)
        hipLaunchKernelGGL(aug_spline3_filter_kernel<double>, dim3(aug_grid(vox / dims[ax])), dim3(256), 0, s, (const double*)workspace, workspace, nvol, D, H, W,
                           ax, z, pow(z, (double)dims[ax]));
    hipLaunchKernelGGL(aug_mapcoord3_kernel, dim3(aug_grid(vox)), dim3(256), 0, s, (const double*)workspace, (float*)dst, nvol, D, H, W, fz, fy, fx, alpha);
    MIS_LAUNCH_CHECK("aug_map_coordinates(order 3)");
    return MIS_OK;
}

// y = a*x + b ; optional clip ; optional additive N(0, noise_std) from a counter-based generator (splitmix64 + Box-Muller)
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void aug_pointwise_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n, float a, float b,
                                                            int do_clip, float lo, float hi, float noise_std, uint64_t seed) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = a * src[i] + b;           // numpy evaluates mean + alpha*(m - mean) unfused; callers fold only exact cases
        if (do_clip) v = fminf(fmaxf(v, lo), hi);
        if (noise_std > 0.f) {
            const uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)i));
            const float u1 = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);     // (0, 1)
            const float u2 = ((float)(uint32_t)((h >> 8) & 0xFFFFFF) + 0.5f) * (1.0f / 16777216.0f);
            v += noise_std * sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
        }
        dst[i] = v;
    }
}

// contrast exactly as numpy evaluates it in float32: clip(mean + alpha * (m - mean), -1, 1)
__global__ __launch_bounds__(256) void aug_contrast_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n, float mean, float alpha) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = __fsub_rn(src[i], mean);
        const float r = __fadd_rn(mean, __fmul_rn(alpha, d));
        dst[i] = fminf(fmaxf(r, -1.f), 1.f);
    }
}

extern "C" int mis_aug_pointwise(const float* src, float* dst, long long n, float a, float b, int do_clip, float lo, float hi, float noise_std,
                                 unsigned long long seed, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(src && dst && n > 0, MIS_EINVAL, "aug_pointwise: bad argument");
    hipLaunchKernelGGL(aug_pointwise_kernel, dim3(aug_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, n, a, b, do_clip, lo, hi,
                       noise_std, (uint64_t)seed);
    MIS_LAUNCH_CHECK("aug_pointwise");
    return MIS_OK;
}

// min / max of a volume (Normalize with data-derived bounds): per-block partials, then one block
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ x, long long n, float* __restrict__ part, int final_stage) {
    __shared__ float smin[256], smax[256];
    float lo = INFINITY, hi = -INFINITY;
    if (!final_stage) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            const float v = x[i];
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    } else {        // x = the partials [nblocks][2]
        for (long long i = threadIdx.x; i < n; i += 256) {
            lo = fminf(lo, x[2 * i]);
            hi = fmaxf(hi, x[2 * i + 1]);
        }
    }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + o]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = smin[0];
        part[2 * blockIdx.x + 1] = smax[0];
    }
}

extern "C" int mis_minmax(const float* x, long long n, float* workspace, float* out, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(x && workspace && out && n > 0, MIS_EINVAL, "minmax: bad argument");
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, workspace, 0);
    MIS_LAUNCH_CHECK("minmax");
    hipLaunchKernelGGL(minmax_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, blocks, out, 1);
    MIS_LAUNCH_CHECK("minmax(final)");
    return MIS_OK;
}

extern "C" int mis_aug_contrast(const float* src, float* dst, long long n, float mean, float alpha, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(src && dst && n > 0, MIS_EINVAL, "aug_contrast: bad argument");
    hipLaunchKernelGGL(aug_contrast_kernel, dim3(aug_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, n, mean, alpha);
    MIS_LAUNCH_CHECK("aug_contrast");
    return MIS_OK;
}
