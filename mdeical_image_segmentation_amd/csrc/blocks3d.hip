// Kernels of the STAND-ALONE / general-shape 3-D building blocks (reference model/unet3d/buildingblocks.py:14-159 create_conv /
// SingleConv with an arbitrary order string, :365-439 Encoder pooling, :553-673 nearest interpolation to the encoder's size) for gfx950.
// The fused engines (engine3d*.py) cover the benchmarked configurations; these kernels cover what they refuse: channel counts that are
// not multiples of 64, MaxPool3d / AvgPool3d with any window, F.interpolate(size=...) between arbitrary grids, GroupNorm AFTER the
// convolution and the LeakyReLU / ELU non-linearities ('cge', 'cl', 'crg', ...).  All of them are HBM-bound streaming passes over
// channels-last (N, D, H, W, Cld) tensors whose channel padding [C, Cld) is kept exactly zero.
//
//   mis_norm_act_fwd / _bwd   y = act(scale[n,c] * x + shift[n,c])           ;  dz = dy * act'(scale * x + shift)
//   mis_gn_fwd_finalize_ld    per-(sample, group) mean / rstd -> per-(sample, channel) scale / shift, arrays with row stride ld
//   mis_gn_bwd_finalize_ld    p, q, r of dx = p*dy + q*x + r (see groupnorm.hip) + dgamma / dbeta, arrays with row stride ld
//   mis_pool3d_fwd / _bwd     MaxPool3d / AvgPool3d(kernel = stride = (kd, kh, kw)), floor mode
//   mis_gather3d_fwd / _bwd   y[n, d, h, w, c] = x[n, mD[d], mH[h], mW[w], c] over a channel slice (nearest resize, concat copy)
#include "common.hpp"

namespace {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_ELU = 3 };

__device__ __forceinline__ float act_f(float v, int act, float slope) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_LEAKY: return v > 0.f ? v : slope * v;
        case ACT_ELU: return v > 0.f ? v : slope * expm1f(v);
        default: return v;
    }
}
__device__ __forceinline__ float dact_f(float v, int act, float slope) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? 1.f : 0.f;
        case ACT_LEAKY: return v > 0.f ? 1.f : slope;
        case ACT_ELU: return v > 0.f ? 1.f : slope * expf(v);
        default: return 1.f;
    }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void norm_act_kernel(const T* __restrict__ dy, int dy_ld, const T* __restrict__ x, int x_ld, T* __restrict__ out,
                                                       int out_ld, int N, long long npix, int C, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int act, float slope) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC;
    const long long total = (long long)N * npix * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long long pp = i / nch;
        const int n = (int)(pp / npix);
        float f[EPC], g[EPC];
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(x + (size_t)pp * x_ld + (size_t)ch * EPC), f);
        if constexpr (BWD) unpack_chunk<T>(*reinterpret_cast<const u32x4*>(dy + (size_t)pp * dy_ld + (size_t)ch * EPC), g);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float v = f[e];
            if (scale != nullptr) v = fmaf(v, scale[(size_t)n * C + ch * EPC + e], shift[(size_t)n * C + ch * EPC + e]);
            f[e] = BWD ? g[e] * dact_f(v, act, slope) : act_f(v, act, slope);
        }
        *reinterpret_cast<u32x4*>(out + (size_t)pp * out_ld + (size_t)ch * EPC) = pack_chunk<T>(f);
    }
}

__global__ void gn_fwd_finalize_ld_kernel(const float* __restrict__ sum, const float* __restrict__ sq, int N, int C, int ld, int G, double count,
                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ scale,
                                          float* __restrict__ shift, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * G) return;
    const int n = idx / G, g = idx - n * G;
    const int cpg = C / G;
    double s = 0.0, q = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        s += (double)sum[(size_t)n * ld + c];
        q += (double)sq[(size_t)n * ld + c];
    }
    const double m = count * cpg;
    const double mean = s / m;
    double var = q / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    mean_out[idx] = (float)mean;
    rstd_out[idx] = (float)rstd;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        const double a = rstd * (double)gamma[c];
        scale[(size_t)n * ld + c] = (float)a;
        shift[(size_t)n * ld + c] = (float)((double)beta[c] - mean * a);
    }
    if (g == G - 1)
        for (int c = C; c < ld; ++c) {
            scale[(size_t)n * ld + c] = 0.f;
            shift[(size_t)n * ld + c] = 0.f;
        }
}

__global__ void gn_bwd_finalize_ld_kernel(const float* __restrict__ S1, const float* __restrict__ S2, const float* __restrict__ mean,
                                          const float* __restrict__ rstd, const float* __restrict__ gamma, int N, int C, int ld, int G, double count,
                                          float* __restrict__ p, float* __restrict__ q, float* __restrict__ r, float* __restrict__ dgamma,
                                          float* __restrict__ dbeta) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int cpg = C / G;
    if (idx < N * G) {
        const int n = idx / G, g = idx - n * G;
        const double mu = mean[idx], rs = rstd[idx];
        double A = 0.0, B = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            const double s1 = S1[(size_t)n * ld + c], s2 = S2[(size_t)n * ld + c];
            A += (double)gamma[c] * s1;
            B += (double)gamma[c] * rs * (s2 - mu * s1);
        }
        const double m = count * cpg;
        const double qq = -rs * rs * B / m;
        const double rr = -qq * mu - rs * A / m;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            p[(size_t)n * ld + c] = (float)(rs * (double)gamma[c]);
            q[(size_t)n * ld + c] = (float)qq;
            r[(size_t)n * ld + c] = (float)rr;
        }
        if (g == G - 1)
            for (int c = C; c < ld; ++c) p[(size_t)n * ld + c] = q[(size_t)n * ld + c] = r[(size_t)n * ld + c] = 0.f;
    }
    if (idx < C) {
        const int c = idx, g = c / cpg;
        double dg = 0.0, db = 0.0;
        for (int n = 0; n < N; ++n) {
            const double mu = mean[n * G + g], rs = rstd[n * G + g];
            const double s1 = S1[(size_t)n * ld + c], s2 = S2[(size_t)n * ld + c];
            dg += rs * (s2 - mu * s1);
            db += s1;
        }
        dgamma[c] = (float)dg;
        dbeta[c] = (float)db;
    }
}

// one thread = one 16-byte channel chunk of one OUTPUT voxel
template <typename T>
__global__ __launch_bounds__(256) void pool3d_fwd_kernel(int avg, int kd, int kh, int kw, const T* __restrict__ x, int x_ld, T* __restrict__ y, int y_ld,
                                                         int N, int D, int H, int W, int C) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC;
    const int oD = D / kd, oH = H / kh, oW = W / kw;
    const long long total = (long long)N * oD * oH * oW * nch;
    const float inv = 1.f / (float)(kd * kh * kw);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long t = i / nch;
        const int ow = (int)(t % oW);
        t /= oW;
        const int oh = (int)(t % oH);
        t /= oH;
        const int od = (int)(t % oD);
        const int n = (int)(t / oD);
        float a[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) a[e] = avg ? 0.f : -INFINITY;
        for (int dz = 0; dz < kd; ++dz)
            for (int dy = 0; dy < kh; ++dy)
                for (int dx = 0; dx < kw; ++dx) {
                    const size_t pix = (((size_t)n * D + od * kd + dz) * H + oh * kh + dy) * W + ow * kw + dx;
                    float f[EPC];
                    unpack_chunk<T>(*reinterpret_cast<const u32x4*>(x + pix * x_ld + (size_t)ch * EPC), f);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) a[e] = avg ? a[e] + f[e] : (f[e] > a[e] ? f[e] : a[e]);
                }
        if (avg)
#pragma unroll
            for (int e = 0; e < EPC; ++e) a[e] *= inv;
        const size_t opix = (((size_t)n * oD + od) * oH + oh) * oW + ow;
        *reinterpret_cast<u32x4*>(y + opix * y_ld + (size_t)ch * EPC) = pack_chunk<T>(a);
    }
}

// one thread = one chunk of one INPUT voxel: the gradient goes to the FIRST maximum of the window in (d, h, w) scan order (torch's
// max_pool3d rule: a later element replaces the maximum only when strictly greater); voxels beyond the floor-mode windows get 0
template <typename T>
__global__ __launch_bounds__(256) void pool3d_bwd_kernel(int avg, int kd, int kh, int kw, const T* __restrict__ x, int x_ld, const T* __restrict__ dy,
                                                         int dy_ld, T* __restrict__ dx, int dx_ld, int N, int D, int H, int W, int C) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC;
    const int oD = D / kd, oH = H / kh, oW = W / kw;
    const long long total = (long long)N * D * H * W * nch;
    const float inv = 1.f / (float)(kd * kh * kw);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long t = i / nch;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        t /= H;
        const int d = (int)(t % D);
        const int n = (int)(t / D);
        const int od = d / kd, oh = h / kh, ow = w / kw;
        float o[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = 0.f;
        if (od < oD && oh < oH && ow < oW) {
            float g[EPC];
            const size_t opix = (((size_t)n * oD + od) * oH + oh) * oW + ow;
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(dy + opix * dy_ld + (size_t)ch * EPC), g);
            if (avg) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) o[e] = g[e] * inv;
            } else {
                const int me = ((d - od * kd) * kh + (h - oh * kh)) * kw + (w - ow * kw);
                float best[EPC];
                int arg[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) best[e] = -INFINITY, arg[e] = 0;
                int k = 0;
                for (int dz = 0; dz < kd; ++dz)
                    for (int yy = 0; yy < kh; ++yy)
                        for (int xx = 0; xx < kw; ++xx, ++k) {
                            const size_t pix = (((size_t)n * D + od * kd + dz) * H + oh * kh + yy) * W + ow * kw + xx;
                            float f[EPC];
                            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(x + pix * x_ld + (size_t)ch * EPC), f);
#pragma unroll
                            for (int e = 0; e < EPC; ++e)
                                if (f[e] > best[e] || k == 0) best[e] = f[e], arg[e] = k;
                        }
#pragma unroll
                for (int e = 0; e < EPC; ++e) o[e] = arg[e] == me ? g[e] : 0.f;
            }
        }
        const size_t pix = (((size_t)n * D + d) * H + h) * W + w;
        *reinterpret_cast<u32x4*>(dx + pix * dx_ld + (size_t)ch * EPC) = pack_chunk<T>(o);
    }
}

// scalar channels: the slices start at arbitrary channel offsets (concat of C0 + C1 channels with C0 % 8 != 0)
template <typename T>
__global__ __launch_bounds__(256) void gather3d_fwd_kernel(const T* __restrict__ x, int x_ld, T* __restrict__ y, int y_ld, int N, int sD, int sH, int sW,
                                                           int dD, int dH, int dW, int C, const int* __restrict__ mD, const int* __restrict__ mH,
                                                           const int* __restrict__ mW) {
    const long long total = (long long)N * dD * dH * dW * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int w = (int)(t % dW);
        t /= dW;
        const int h = (int)(t % dH);
        t /= dH;
        const int d = (int)(t % dD);
        const int n = (int)(t / dD);
        const size_t sp = (((size_t)n * sD + mD[d]) * sH + mH[h]) * sW + mW[w];
        const size_t dp = (((size_t)n * dD + d) * dH + h) * dW + w;
        y[dp * y_ld + c] = x[sp * x_ld + c];
    }
}

// dx[source voxel] = sum of dy over the destination voxels that read it: [lo, hi) per axis (the maps are monotone), fixed order
template <typename T>
__global__ __launch_bounds__(256) void gather3d_bwd_kernel(const T* __restrict__ dy, int dy_ld, T* __restrict__ dx, int dx_ld, int N, int sD, int sH,
                                                           int sW, int dD, int dH, int dW, int C, const int* __restrict__ rD,
                                                           const int* __restrict__ rH, const int* __restrict__ rW) {
    const long long total = (long long)N * sD * sH * sW * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int w = (int)(t % sW);
        t /= sW;
        const int h = (int)(t % sH);
        t /= sH;
        const int d = (int)(t % sD);
        const int n = (int)(t / sD);
        float acc = 0.f;
        for (int a = rD[2 * d]; a < rD[2 * d + 1]; ++a)
            for (int b = rH[2 * h]; b < rH[2 * h + 1]; ++b)
                for (int e = rW[2 * w]; e < rW[2 * w + 1]; ++e) acc += ld_elem<T>(dy + ((((size_t)n * dD + a) * dH + b) * dW + e) * dy_ld + c);
        st_elem<T>(dx + ((((size_t)n * sD + d) * sH + h) * sW + w) * dx_ld + c, acc);
    }
}

inline unsigned grid_for(long long items, long long cap = 8192) {
    long long b = (items + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}   // namespace

static int norm_act_common(const char* what, bool bwd, int dtype, const void* dy, int dy_ld, const void* x, int x_ld, void* out, int out_ld, int N,
                           long long npix, int C, const float* scale, const float* shift, int act, float slope, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "%s: bad dtype %d", what, dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(x && out && (!bwd || dy), MIS_EINVAL, "%s: null pointer", what);
    MIS_REQUIRE((scale == nullptr) == (shift == nullptr), MIS_EINVAL, "%s: scale / shift come together", what);
    MIS_REQUIRE(act >= ACT_NONE && act <= ACT_ELU, MIS_EINVAL, "%s: activation code %d", what, act);
    MIS_REQUIRE(N > 0 && npix > 0 && C > 0 && C % EPC == 0 && x_ld % EPC == 0 && out_ld % EPC == 0 && (!bwd || dy_ld % EPC == 0) && x_ld >= C &&
                    out_ld >= C,
                MIS_EINVAL, "%s: sizes / alignment", what);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = grid_for((long long)N * npix * (C / EPC));
    if (dtype == MIS_BF16) {
        if (bwd)
            hipLaunchKernelGGL((norm_act_kernel<__bf16, true>), dim3(g), dim3(256), 0, s, (const __bf16*)dy, dy_ld, (const __bf16*)x, x_ld, (__bf16*)out,
                               out_ld, N, npix, C, scale, shift, act, slope);
        else
            hipLaunchKernelGGL((norm_act_kernel<__bf16, false>), dim3(g), dim3(256), 0, s, (const __bf16*)nullptr, 0, (const __bf16*)x, x_ld,
                               (__bf16*)out, out_ld, N, npix, C, scale, shift, act, slope);
    } else {
        if (bwd)
            hipLaunchKernelGGL((norm_act_kernel<float, true>), dim3(g), dim3(256), 0, s, (const float*)dy, dy_ld, (const float*)x, x_ld, (float*)out,
                               out_ld, N, npix, C, scale, shift, act, slope);
        else
            hipLaunchKernelGGL((norm_act_kernel<float, false>), dim3(g), dim3(256), 0, s, (const float*)nullptr, 0, (const float*)x, x_ld, (float*)out,
                               out_ld, N, npix, C, scale, shift, act, slope);
    }
    MIS_LAUNCH_CHECK(what);
    return MIS_OK;
}

extern "C" int mis_norm_act_fwd(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, long long npix, int C, const float* scale,
                                const float* shift, int act, float slope, void* stream) {
    return norm_act_common("norm_act_fwd", false, dtype, nullptr, 0, x, x_ld, y, y_ld, N, npix, C, scale, shift, act, slope, stream);
}

extern "C" int mis_norm_act_bwd(int dtype, const void* dy, int dy_ld, const void* x, int x_ld, void* dz, int dz_ld, int N, long long npix, int C,
                                const float* scale, const float* shift, int act, float slope, void* stream) {
    return norm_act_common("norm_act_bwd", true, dtype, dy, dy_ld, x, x_ld, dz, dz_ld, N, npix, C, scale, shift, act, slope, stream);
}

// y = alpha * x * (mask > 0): training-mode nn.Dropout of a create_conv order ('d', reference model/unet3d/buildingblocks.py:105-106) - the Bernoulli mask is drawn by
// the caller; the same kernel carries the gradient back (dx = alpha * dy * (mask > 0))
template <typename T>
__global__ __launch_bounds__(256) void mask_scale_kernel(const T* __restrict__ x, int x_ld, const T* __restrict__ m, int m_ld, T* __restrict__ y, int y_ld,
                                                         long long npix, int C, float alpha) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC;
    const long long total = npix * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long long pix = i / nch;
        float f[EPC], g[EPC];
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(x + (size_t)pix * x_ld + (size_t)ch * EPC), f);
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(m + (size_t)pix * m_ld + (size_t)ch * EPC), g);
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[e] = g[e] > 0.f ? alpha * f[e] : 0.f;
        *reinterpret_cast<u32x4*>(y + (size_t)pix * y_ld + (size_t)ch * EPC) = pack_chunk<T>(f);
    }
}

extern "C" int mis_mask_scale(int dtype, const void* x, int x_ld, const void* mask, int mask_ld, void* y, int y_ld, long long npix, int C, float alpha, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "mask_scale: bad dtype %d", dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(x && mask && y && npix > 0 && C > 0 && C % EPC == 0 && x_ld % EPC == 0 && mask_ld % EPC == 0 && y_ld % EPC == 0, MIS_EINVAL, "mask_scale: sizes / alignment");
    long long blocks = (npix * (C / EPC) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(mask_scale_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)x, x_ld, (const __bf16*)mask, mask_ld, (__bf16*)y, y_ld, npix, C, alpha);
    else
        hipLaunchKernelGGL(mask_scale_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, x_ld, (const float*)mask, mask_ld, (float*)y, y_ld, npix, C, alpha);
    MIS_LAUNCH_CHECK("mask_scale");
    return MIS_OK;
}

extern "C" int mis_gn_fwd_finalize_ld(const float* sum, const float* sq, int N, int C, int ld, int G, double count, const float* gamma,
                                      const float* beta, float eps, float* scale, float* shift, float* mean, float* rstd, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(sum && sq && gamma && beta && scale && shift && mean && rstd, MIS_EINVAL, "gn_fwd_finalize_ld: null pointer");
    MIS_REQUIRE(N > 0 && G > 0 && C > 0 && C % G == 0 && ld >= C && count > 0.0, MIS_EINVAL, "gn_fwd_finalize_ld: bad sizes");
    hipLaunchKernelGGL(gn_fwd_finalize_ld_kernel, dim3((N * G + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), sum, sq, N, C, ld, G,
                       count, gamma, beta, eps, scale, shift, mean, rstd);
    MIS_LAUNCH_CHECK("gn_fwd_finalize_ld");
    return MIS_OK;
}

extern "C" int mis_gn_bwd_finalize_ld(const float* S1, const float* S2, const float* mean, const float* rstd, const float* gamma, int N, int C,
                                      int ld, int G, double count, float* p, float* q, float* r, float* dgamma, float* dbeta, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(S1 && S2 && mean && rstd && gamma && p && q && r && dgamma && dbeta, MIS_EINVAL, "gn_bwd_finalize_ld: null pointer");
    MIS_REQUIRE(N > 0 && C > 0 && G > 0 && C % G == 0 && ld >= C && count > 0.0, MIS_EINVAL, "gn_bwd_finalize_ld: sizes");
    const int nthreads = (N * G > C) ? N * G : C;
    hipLaunchKernelGGL(gn_bwd_finalize_ld_kernel, dim3((nthreads + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), S1, S2, mean, rstd,
                       gamma, N, C, ld, G, count, p, q, r, dgamma, dbeta);
    MIS_LAUNCH_CHECK("gn_bwd_finalize_ld");
    return MIS_OK;
}

static int pool_check(const char* what, int dtype, int mode, int kd, int kh, int kw, int N, int D, int H, int W, int C, int ld_a, int ld_b) {
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "%s: bad dtype %d", what, dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(mode == 0 || mode == 1, MIS_EINVAL, "%s: mode %d (0 = max, 1 = avg)", what, mode);
    MIS_REQUIRE(kd > 0 && kh > 0 && kw > 0 && kd <= D && kh <= H && kw <= W, MIS_EINVAL, "%s: window %dx%dx%d on a %dx%dx%d grid", what, kd, kh, kw, D,
                H, W);
    MIS_REQUIRE(N > 0 && C > 0 && C % EPC == 0 && ld_a % EPC == 0 && ld_b % EPC == 0 && ld_a >= C && ld_b >= C, MIS_EINVAL, "%s: sizes / alignment",
                what);
    return MIS_OK;
}

extern "C" int mis_pool3d_fwd(int dtype, int mode, int kd, int kh, int kw, const void* x, int x_ld, void* y, int y_ld, int N, int D, int H, int W,
                              int C, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(x && y, MIS_EINVAL, "pool3d_fwd: null pointer");
    if (int rc = pool_check("pool3d_fwd", dtype, mode, kd, kh, kw, N, D, H, W, C, x_ld, y_ld)) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    const unsigned g = grid_for((long long)N * (D / kd) * (H / kh) * (W / kw) * (C / EPC));
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(pool3d_fwd_kernel<__bf16>, dim3(g), dim3(256), 0, s, mode, kd, kh, kw, (const __bf16*)x, x_ld, (__bf16*)y, y_ld, N, D, H, W, C);
    else
        hipLaunchKernelGGL(pool3d_fwd_kernel<float>, dim3(g), dim3(256), 0, s, mode, kd, kh, kw, (const float*)x, x_ld, (float*)y, y_ld, N, D, H, W, C);
    MIS_LAUNCH_CHECK("pool3d_fwd");
    return MIS_OK;
}

extern "C" int mis_pool3d_bwd(int dtype, int mode, int kd, int kh, int kw, const void* x, int x_ld, const void* dy, int dy_ld, void* dx, int dx_ld,
                              int N, int D, int H, int W, int C, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(x && dy && dx, MIS_EINVAL, "pool3d_bwd: null pointer");
    if (int rc = pool_check("pool3d_bwd", dtype, mode, kd, kh, kw, N, D, H, W, C, x_ld, dx_ld)) return rc;
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(dy_ld % EPC == 0 && dy_ld >= C, MIS_EINVAL, "pool3d_bwd: dy_ld");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = grid_for((long long)N * D * H * W * (C / EPC));
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(pool3d_bwd_kernel<__bf16>, dim3(g), dim3(256), 0, s, mode, kd, kh, kw, (const __bf16*)x, x_ld, (const __bf16*)dy, dy_ld,
                           (__bf16*)dx, dx_ld, N, D, H, W, C);
    else
        hipLaunchKernelGGL(pool3d_bwd_kernel<float>, dim3(g), dim3(256), 0, s, mode, kd, kh, kw, (const float*)x, x_ld, (const float*)dy, dy_ld,
                           (float*)dx, dx_ld, N, D, H, W, C);
    MIS_LAUNCH_CHECK("pool3d_bwd");
    return MIS_OK;
}

// mD / mH / mW: DEVICE int arrays of dD / dH / dW source indices (the caller validated them against sD / sH / sW)
extern "C" int mis_gather3d_fwd(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, int sD, int sH, int sW, int dD, int dH, int dW, int C,
                                const int* mD, const int* mH, const int* mW, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "gather3d_fwd: bad dtype %d", dtype);
    MIS_REQUIRE(x && y && mD && mH && mW, MIS_EINVAL, "gather3d_fwd: null pointer");
    MIS_REQUIRE(N > 0 && sD > 0 && sH > 0 && sW > 0 && dD > 0 && dH > 0 && dW > 0 && C > 0 && x_ld >= C && y_ld >= C, MIS_EINVAL, "gather3d_fwd: sizes");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = grid_for((long long)N * dD * dH * dW * C, 16384);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(gather3d_fwd_kernel<__bf16>, dim3(g), dim3(256), 0, s, (const __bf16*)x, x_ld, (__bf16*)y, y_ld, N, sD, sH, sW, dD, dH, dW, C, mD,
                           mH, mW);
    else
        hipLaunchKernelGGL(gather3d_fwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, x_ld, (float*)y, y_ld, N, sD, sH, sW, dD, dH, dW, C, mD, mH,
                           mW);
    MIS_LAUNCH_CHECK("gather3d_fwd");
    return MIS_OK;
}

// rD / rH / rW: DEVICE int arrays of sD / sH / sW [lo, hi) pairs of destination indices
extern "C" int mis_gather3d_bwd(int dtype, const void* dy, int dy_ld, void* dx, int dx_ld, int N, int sD, int sH, int sW, int dD, int dH, int dW, int C,
                                const int* rD, const int* rH, const int* rW, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "gather3d_bwd: bad dtype %d", dtype);
    MIS_REQUIRE(dy && dx && rD && rH && rW, MIS_EINVAL, "gather3d_bwd: null pointer");
    MIS_REQUIRE(N > 0 && sD > 0 && sH > 0 && sW > 0 && dD > 0 && dH > 0 && dW > 0 && C > 0 && dy_ld >= C && dx_ld >= C, MIS_EINVAL, "gather3d_bwd: sizes");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = grid_for((long long)N * sD * sH * sW * C, 16384);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(gather3d_bwd_kernel<__bf16>, dim3(g), dim3(256), 0, s, (const __bf16*)dy, dy_ld, (__bf16*)dx, dx_ld, N, sD, sH, sW, dD, dH, dW, C,
                           rD, rH, rW);
    else
        hipLaunchKernelGGL(gather3d_bwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)dy, dy_ld, (float*)dx, dx_ld, N, sD, sH, sW, dD, dH, dW, C, rD,
                           rH, rW);
    MIS_LAUNCH_CHECK("gather3d_bwd");
    return MIS_OK;
}
