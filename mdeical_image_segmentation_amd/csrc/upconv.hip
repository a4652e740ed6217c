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
#include "dispatch_cfg.hpp"

__device__ __forceinline__ void upc_src(int dst, float inv_scale, int n, int& i0, int& i1, float& lam) {
    float src = ((float)dst + 0.5f) * inv_scale - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > n - 1) i0 = n - 1;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    lam = src - (float)i0;
}

// per axis and tap offset k in {0,1,2}: the two source indices and their weights for upsampled position o + k - 1 (weights 0 when that position is
// in the zero padding of the 3x3 convolution) - branch-free, so that all 36 chunk loads of an output element are independent and issued together
struct UpcAxis {
    int i0[3], i1[3];
    float w0[3], w1[3];
};
__device__ __forceinline__ void upc_axis(int o, float inv, int n, int n_up, UpcAxis& a) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = o + k - 1;
        const bool ok = p >= 0 && p < n_up;
        float lam;
        upc_src(ok ? p : 0, inv, n, a.i0[k], a.i1[k], lam);
        a.w0[k] = ok ? 1.f - lam : 0.f;
        a.w1[k] = ok ? lam : 0.f;
    }
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
        const long long r = i / nch;
        const int ox = (int)(r % W);
        const int rr = (int)(r / W);
        const int oy = rr % H;
        const int n = rr / H;
        UpcAxis ay, ax;
        upc_axis(oy, inv, h, H, ay);
        upc_axis(ox, inv, w, W, ax);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = bias ? bias[ch * EPC + e] : 0.f;
        const T* img = Z + (size_t)n * h * w * zld + (size_t)ch * EPC;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const T* r0 = img + (size_t)ay.i0[ky] * w * zld + (size_t)(ky * 3) * C;
            const T* r1 = img + (size_t)ay.i1[ky] * w * zld + (size_t)(ky * 3) * C;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float a[EPC], b[EPC], c[EPC], d[EPC];
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(r0 + (size_t)ax.i0[kx] * zld + kx * C), a);
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(r0 + (size_t)ax.i1[kx] * zld + kx * C), b);
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(r1 + (size_t)ax.i0[kx] * zld + kx * C), c);
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(r1 + (size_t)ax.i1[kx] * zld + kx * C), d);
                const float hx = ax.w0[kx], lx = ax.w1[kx], hy = ay.w0[ky], ly = ay.w1[ky];
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] += hy * (hx * a[e] + lx * b[e]) + ly * (hx * c[e] + lx * d[e]);
            }
        }
        *reinterpret_cast<u32x4*>(y + (((size_t)n * H + oy) * W + ox) * y_ld + (size_t)ch * EPC) = pack_chunk<T>(acc);
    }
}

// Two-pass (separable) form of the forward gather: the x-direction first, on the low-resolution ROWS only,
//   V[n][jy][ox][ky][c] = sum_kx [ox+kx-1 inside] bilinear_x(Z[n][jy][.][(ky,kx)*C + c])(ox + kx - 1)          (6 chunk loads per element, 3/s of the output size)
//   y[n][oy][ox][c]     = bias + sum_ky [oy+ky-1 inside] bilinear_y(V[n][.][ox][ky][c])(oy + ky - 1)             (6 chunk loads per element)
// i.e. 6 + 18/s loads per output element instead of 36; V (3/s x the output) goes through HBM once.  Same arithmetic up to fp32 summation order.
template <typename T>
__global__ __launch_bounds__(256) void upconv_xpass_kernel(const T* __restrict__ Z, T* __restrict__ V, int N, int h, int w, int s, int C) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC, W = w * s;
    const float inv = 1.0f / (float)s;
    const size_t zld = (size_t)9 * C;
    const long long total = (long long)N * h * W * 3 * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long r = i / nch;
        const int ky = (int)(r % 3);
        r /= 3;
        const int ox = (int)(r % W);
        const long long row = r / W;                  // n * h + jy
        UpcAxis ax;
        upc_axis(ox, inv, w, W, ax);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        const T* base = Z + (size_t)row * w * zld + (size_t)(ky * 3) * C + (size_t)ch * EPC;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float a[EPC], b[EPC];
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + (size_t)ax.i0[kx] * zld + kx * C), a);
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + (size_t)ax.i1[kx] * zld + kx * C), b);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += ax.w0[kx] * a[e] + ax.w1[kx] * b[e];
        }
        *reinterpret_cast<u32x4*>(V + (size_t)i * EPC) = pack_chunk<T>(acc);      // V is dense in exactly this index order
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upconv_ypass_kernel(const T* __restrict__ V, T* __restrict__ y, int y_ld, const float* __restrict__ bias, int N, int h,
                                                           int w, int s, int C) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC, H = h * s, W = w * s;
    const float inv = 1.0f / (float)s;
    const size_t vrow = (size_t)W * 3 * C;            // one low-resolution row of V
    const long long total = (long long)N * H * W * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long long r = i / nch;
        const int ox = (int)(r % W);
        const int rr = (int)(r / W);
        const int oy = rr % H;
        const int n = rr / H;
        UpcAxis ay;
        upc_axis(oy, inv, h, H, ay);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = bias ? bias[ch * EPC + e] : 0.f;
        const T* base = V + (size_t)n * h * vrow + (size_t)ox * 3 * C + (size_t)ch * EPC;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float a[EPC], b[EPC];
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + (size_t)ay.i0[ky] * vrow + ky * C), a);
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + (size_t)ay.i1[ky] * vrow + ky * C), b);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += ay.w0[ky] * a[e] + ay.w1[ky] * b[e];
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

// weight with which upsampled position q references low-resolution index j (0 when q lies in the zero padding)
__device__ __forceinline__ float upc_coef(int q, int j, float inv, int n, int n_up) {
    const bool ok = q >= 0 && q < n_up;
    int i0, i1;
    float lam;
    upc_src(ok ? q : 0, inv, n, i0, i1, lam);
    const float wgt = (i0 == j ? 1.f - lam : 0.f) + (i1 == j ? lam : 0.f);
    return ok ? wgt : 0.f;
}

// C == 64: one thread per (low-resolution pixel, 16-byte chunk, row part) accumulates ALL nine taps from one pass over the (2s+3)^2 window of dY
// (each dY chunk is loaded once instead of once per tap); for large scales the window rows are dealt round-robin to R threads whose partial sums
// are added in a fixed order through LDS.
template <typename T, int R>
__global__ __launch_bounds__(256) void upconv_gather_bwd64_kernel(const T* __restrict__ dY, int dy_ld, T* __restrict__ dZ, int N, int h, int w, int s) {
    constexpr int EPC = Tr<T>::EPC, NCH = 64 / EPC, JPB = 256 / (NCH * R);
    __shared__ float red[R > 1 ? 256 * 3 * EPC : 1];
    const int H = h * s, W = w * s;
    const float inv = 1.0f / (float)s;
    const int ch = threadIdx.x % NCH;
    const int part = (threadIdx.x / NCH) % R;
    const int jl = threadIdx.x / (NCH * R);
    const long long jtot = (long long)N * h * w;
    for (long long jb = (long long)blockIdx.x * JPB; jb < jtot; jb += (long long)gridDim.x * JPB) {
        const long long j = jb + jl;
        const bool live = j < jtot;
        float acc[9][EPC];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[t][e] = 0.f;
        if (live) {
            const int jx = (int)(j % w);
            const int rr = (int)(j / w);
            const int jy = rr % h;
            const int n = rr / h;
            int oylo = s * (jy - 1) - 2, oyhi = s * (jy + 2) + 1, oxlo = s * (jx - 1) - 2, oxhi = s * (jx + 2) + 1;
            if (oylo < 0) oylo = 0;
            if (oyhi > H - 1) oyhi = H - 1;
            if (oxlo < 0) oxlo = 0;
            if (oxhi > W - 1) oxhi = W - 1;
            const T* img = dY + (size_t)n * H * W * dy_ld + (size_t)ch * EPC;
            for (int oy = oylo + part; oy <= oyhi; oy += R) {
                float wy[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) wy[k] = upc_coef(oy + k - 1, jy, inv, h, H);
                if (wy[0] == 0.f && wy[1] == 0.f && wy[2] == 0.f) continue;
                const T* row = img + (size_t)oy * W * dy_ld;
#pragma unroll 2
                for (int ox = oxlo; ox <= oxhi; ++ox) {
                    float g[EPC], wx[3];
                    unpack_chunk<T>(*reinterpret_cast<const u32x4*>(row + (size_t)ox * dy_ld), g);
#pragma unroll
                    for (int k = 0; k < 3; ++k) wx[k] = upc_coef(ox + k - 1, jx, inv, w, W);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float cw = wy[ky] * wx[kx];
#pragma unroll
                            for (int e = 0; e < EPC; ++e) acc[ky * 3 + kx][e] = fmaf(cw, g[e], acc[ky * 3 + kx][e]);
                        }
                }
            }
        }
        if (R > 1) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) red[(threadIdx.x * 3 + kx) * EPC + e] = acc[ky * 3 + kx][e];
                __syncthreads();
                if (part == 0)
                    for (int pp = 1; pp < R; ++pp)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int e = 0; e < EPC; ++e) acc[ky * 3 + kx][e] += red[((threadIdx.x + pp * NCH) * 3 + kx) * EPC + e];
                __syncthreads();
            }
        }
        if (live && part == 0) {
            T* dst = dZ + (size_t)j * 9 * 64 + (size_t)ch * EPC;
#pragma unroll
            for (int t = 0; t < 9; ++t) *reinterpret_cast<u32x4*>(dst + t * 64) = pack_chunk<T>(acc[t]);
        }
    }
}

template <typename T, int R>
static void upc_launch_bwd64(const void* dy, int dy_ld, void* dz, int N, int h, int w, int s, hipStream_t st) {
    constexpr int JPB = 256 / ((64 / Tr<T>::EPC) * R);
    long long blocks = ((long long)N * h * w + JPB - 1) / JPB;
    if (blocks > 262144) blocks = 262144;
    hipLaunchKernelGGL((upconv_gather_bwd64_kernel<T, R>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)dy, dy_ld, (T*)dz, N, h, w, s);
}

template <typename T>
static void upc_dispatch_bwd64(const void* dy, int dy_ld, void* dz, int N, int h, int w, int s, hipStream_t st) {
    if (s <= 4)
        upc_launch_bwd64<T, 1>(dy, dy_ld, dz, N, h, w, s, st);
    else if (s <= 8)
        upc_launch_bwd64<T, 4>(dy, dy_ld, dz, N, h, w, s, st);
    else
        upc_launch_bwd64<T, 8>(dy, dy_ld, dz, N, h, w, s, st);
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

extern "C" size_t mis_upconv_gather_fwd_workspace_bytes(int dtype, int N, int h, int w, int scale, int C) {
    return (size_t)N * h * w * scale * 3 * C * (dtype == MIS_BF16 ? 2 : 4);
}

/* workspace == NULL: single-pass gather (36 chunk loads per output element); otherwise the two-pass separable form through V = workspace */
extern "C" int mis_upconv_gather_fwd(int dtype, const void* z, void* y, int y_ld, const float* bias, int N, int h, int w, int scale, int C, void* workspace,
                                     void* stream) {
    (void)hipGetLastError();
    if (int rc = upc_check("upconv_gather_fwd", dtype, z, y, y_ld, N, h, w, scale, C)) return rc;
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (workspace != nullptr) {
        const unsigned g1 = upc_grid((long long)N * h * w * scale * 3 * (C / EPC));
        const unsigned g2 = upc_grid((long long)N * h * scale * w * scale * (C / EPC));
        if (dtype == MIS_BF16) {
            hipLaunchKernelGGL(upconv_xpass_kernel<__bf16>, dim3(g1), dim3(256), 0, st, (const __bf16*)z, (__bf16*)workspace, N, h, w, scale, C);
            hipLaunchKernelGGL(upconv_ypass_kernel<__bf16>, dim3(g2), dim3(256), 0, st, (const __bf16*)workspace, (__bf16*)y, y_ld, bias, N, h, w, scale, C);
        } else {
            hipLaunchKernelGGL(upconv_xpass_kernel<float>, dim3(g1), dim3(256), 0, st, (const float*)z, (float*)workspace, N, h, w, scale, C);
            hipLaunchKernelGGL(upconv_ypass_kernel<float>, dim3(g2), dim3(256), 0, st, (const float*)workspace, (float*)y, y_ld, bias, N, h, w, scale, C);
        }
        MIS_LAUNCH_CHECK("upconv_gather_fwd (two-pass)");
        return MIS_OK;
    }
    const unsigned g = upc_grid((long long)N * h * scale * w * scale * (C / EPC));
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
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (C == 64 && !mis_sw(SW_UPCONV_BWD_GENERIC)) {
        if (dtype == MIS_BF16)
            upc_dispatch_bwd64<__bf16>(dy, dy_ld, dz, N, h, w, scale, st);
        else
            upc_dispatch_bwd64<float>(dy, dy_ld, dz, N, h, w, scale, st);
        MIS_LAUNCH_CHECK("upconv_gather_bwd64");
        return MIS_OK;
    }
    const unsigned g = upc_grid((long long)N * h * w * 9 * (C / EPC));
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(upconv_gather_bwd_kernel<__bf16>, dim3(g), dim3(256), 0, st, (const __bf16*)dy, dy_ld, (__bf16*)dz, N, h, w, scale, C);
    else
        hipLaunchKernelGGL(upconv_gather_bwd_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)dy, dy_ld, (float*)dz, N, h, w, scale, C);
    MIS_LAUNCH_CHECK("upconv_gather_bwd");
    return MIS_OK;
}
