// conv3x3(pad 1)( bilinear_upsample_s(x) ) without the upsampled tensor - the decoder-to-decoder branches of UNet 3+
// (reference model/unet2d/unet.py:190-192, 229-236, 273-285, 322-339: nn.Upsample(scale_factor=s, mode='bilinear') -> Conv2d(C, 64, 3, padding=1)
// with s = 2..16 and C = 320 / 1024, i.e. up to a 1024-channel full-resolution intermediate) for gfx950.
//
// The channel contraction commutes with the spatial interpolation, so it runs on the LOW-resolution grid:
//   Z[j][tap*C + co] = sum_ci x[j][ci] * w[co][ci][tap]            (mis_conv_igemm, ksize 1, 9*C output columns: s^2 x fewer FLOPs)
//   y[o][co] = bias[co] + sum_{tap : o + tap - 1 inside the s*h x s*w image} bilinear(Z[.][tap*C + co])(o + tap - 1)     mis_upconv_gather_fwd
// (taps that fall into the zero padding of the 3x3 convolution contribute nothing, exactly as in the reference), and
//   dZ[j][tap*C + co] = sum_{o} coef(o + tap - 1 -> j) * dY[o][co]                                                         mis_upconv_gather_bwd
// followed by dX = dZ x W^T (mis_conv_igemm, ksize 1) and dW = x^T x dZ (mis_wgrad, ksize 1) on the low-resolution grid.
// Bilinear rule (align_corners=False): src = max(0, (dst + 0.5)/s - 0.5), i0 = floor(src), i1 = min(i0 + 1, n - 1), lambda = src - i0.
// Both gathers are index work bound by cache/HBM bandwidth: one 16-byte channel chunk per thread, fp32 accumulation in a fixed order.
#include "common.hpp"

__device__ __forceinline__ void upc_src(int dst, float inv_scale, int n, int& i0, int& i1, float& lam) {
    float src = ((float)dst + 0.5f) * inv_scale - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > n - 1) i0 = n - 1;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    lam = src - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void upconv_gather_fwd_kernel(const T* __restrict__ Z, T* __restrict__ y, int y_ld, const float* __restrict__ bias, int N,
                                                                int h, int w, int s, int C) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC, H = h * s, W = w * s;
    const float inv = 1.0f / (float)s;
    const size_t zld = (size_t)9 * C;
    const long long total = (long long)N * H * W * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long r = i / nch;
        const int ox = (int)(r % W);
        r /= W;
        const int oy = (int)(r % H);
        const int n = (int)(r / H);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = bias ? bias[ch * EPC + e] : 0.f;
        const T* img = Z + (size_t)n * h * w * zld + (size_t)ch * EPC;
        for (int ky = 0; ky < 3; ++ky) {
            const int py = oy + ky - 1;
            if (py < 0 || py >= H) continue;
            int y0, y1;
            float ly;
            upc_src(py, inv, h, y0, y1, ly);
            const float hy = 1.f - ly;
            for (int kx = 0; kx < 3; ++kx) {
                const int px = ox + kx - 1;
                if (px < 0 || px >= W) continue;
                int x0, x1;
                float lx;
                upc_src(px, inv, w, x0, x1, lx);
                const float hx = 1.f - lx;
                const T* base = img + (size_t)(ky * 3 + kx) * C;
                float a[EPC], b[EPC], c[EPC], d[EPC];
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + ((size_t)y0 * w + x0) * zld), a);
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + ((size_t)y0 * w + x1) * zld), b);
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + ((size_t)y1 * w + x0) * zld), c);
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + ((size_t)y1 * w + x1) * zld), d);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] += hy * (hx * a[e] + lx * b[e]) + ly * (hx * c[e] + lx * d[e]);
            }
        }
        *reinterpret_cast<u32x4*>(y + (((size_t)n * H + oy) * W + ox) * y_ld + (size_t)ch * EPC) = pack_chunk<T>(acc);
    }
}

// one thread per (low-resolution pixel j, tap, 16-byte channel chunk): walks the <= (2s+1)^2 upsampled positions that reference j
template <typename T>
__global__ __launch_bounds__(256) void upconv_gather_bwd_kernel(const T* __restrict__ dY, int dy_ld, T* __restrict__ dZ, int N, int h, int w, int s, int C) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC, H = h * s, W = w * s;
    const float inv = 1.0f / (float)s;
    const long long total = (long long)N * h * w * 9 * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long r = i / nch;
        const int tap = (int)(r % 9);
        r /= 9;
        const int jx = (int)(r % w);
        r /= w;
        const int jy = (int)(r % h);
        const int n = (int)(r / h);
        const int ky = tap / 3, kx = tap % 3;
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        int qlo = s * (jy - 1) - 1, qhi = s * (jy + 2);
        if (qlo < 0) qlo = 0;
        if (qhi > H - 1) qhi = H - 1;
        int plo = s * (jx - 1) - 1, phi = s * (jx + 2);
        if (plo < 0) plo = 0;
        if (phi > W - 1) phi = W - 1;
        const T* img = dY + (size_t)n * H * W * dy_ld + (size_t)ch * EPC;
        for (int q = qlo; q <= qhi; ++q) {
            const int oy = q - ky + 1;
            if (oy < 0 || oy >= H) continue;
            int i0, i1;
            float lam;
            upc_src(q, inv, h, i0, i1, lam);
            float wy = 0.f;
            if (i0 == jy) wy += 1.f - lam;
            if (i1 == jy) wy += lam;
            if (wy == 0.f) continue;
            for (int p = plo; p <= phi; ++p) {
                const int ox = p - kx + 1;
                if (ox < 0 || ox >= W) continue;
                upc_src(p, inv, w, i0, i1, lam);
                float wx = 0.f;
                if (i0 == jx) wx += 1.f - lam;
                if (i1 == jx) wx += lam;
                if (wx == 0.f) continue;
                float g[EPC];
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(img + ((size_t)oy * W + ox) * dy_ld), g);
                const float cw = wy * wx;
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] = fmaf(cw, g[e], acc[e]);
            }
        }
        *reinterpret_cast<u32x4*>(dZ + ((((size_t)n * h + jy) * w + jx) * 9 + tap) * C + (size_t)ch * EPC) = pack_chunk<T>(acc);
    }
}

static int upc_check(const char* what, int dtype, const void* a, const void* b, int ld, int N, int h, int w, int s, int C) {
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "%s: bad dtype %d", what, dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(a && b && a != b, MIS_EINVAL, "%s: null / aliased pointer", what);
    MIS_REQUIRE(N > 0 && h > 0 && w > 0 && C > 0 && C % EPC == 0 && ld >= C && ld % EPC == 0, MIS_EINVAL, "%s: sizes / alignment", what);
    MIS_REQUIRE(s >= 1 && s <= 32, MIS_EUNSUPPORTED, "%s: scale %d", what, s);
    MIS_REQUIRE((long long)h * s * w * s < (1ll << 31), MIS_EUNSUPPORTED, "%s: image too large", what);
    return MIS_OK;
}

static unsigned upc_grid(long long total) {
    long long b = (total + 255) / 256;
    if (b > 65536) b = 65536;
    if (b < 1) b = 1;
    return (unsigned)b;
}

extern "C" int mis_upconv_gather_fwd(int dtype, const void* z, void* y, int y_ld, const float* bias, int N, int h, int w, int scale, int C, void* stream) {
    (void)hipGetLastError();
    if (int rc = upc_check("upconv_gather_fwd", dtype, z, y, y_ld, N, h, w, scale, C)) return rc;
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    const unsigned g = upc_grid((long long)N * h * scale * w * scale * (C / EPC));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(upconv_gather_fwd_kernel<__bf16>, dim3(g), dim3(256), 0, st, (const __bf16*)z, (__bf16*)y, y_ld, bias, N, h, w, scale, C);
    else
        hipLaunchKernelGGL(upconv_gather_fwd_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)z, (float*)y, y_ld, bias, N, h, w, scale, C);
    MIS_LAUNCH_CHECK("upconv_gather_fwd");
    return MIS_OK;
}

extern "C" int mis_upconv_gather_bwd(int dtype, const void* dy, int dy_ld, void* dz, int N, int h, int w, int scale, int C, void* stream) {
    (void)hipGetLastError();
    if (int rc = upc_check("upconv_gather_bwd", dtype, dy, dz, dy_ld, N, h, w, scale, C)) return rc;
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    const unsigned g = upc_grid((long long)N * h * w * 9 * (C / EPC));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(upconv_gather_bwd_kernel<__bf16>, dim3(g), dim3(256), 0, st, (const __bf16*)dy, dy_ld, (__bf16*)dz, N, h, w, scale, C);
    else
        hipLaunchKernelGGL(upconv_gather_bwd_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)dy, dy_ld, (float*)dz, N, h, w, scale, C);
    MIS_LAUNCH_CHECK("upconv_gather_bwd");
    return MIS_OK;
}
