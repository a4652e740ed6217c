// BatchNorm2d of the reference's `unetConv2` block (model/unet2d/layers.py:17-25: Conv2d(bias) -> BatchNorm2d -> ReLU,
// PyTorch defaults eps 1e-5, momentum 0.1, affine, track_running_stats) for gfx950.
//
// The convolution itself is mis_conv_igemm (bias, no ReLU).  The statistics come from mis_chanstats (per (n,c) sum and
// sum of squares, fixed order); this file holds
//   mis_bn_fwd_finalize : batch mean / biased var -> per-channel scale/shift (replicated per sample so that the (n,c) affine
//                         consumers can be shared with GroupNorm), running-stat update with the UNBIASED variance
//   mis_affine_act      : y = [relu](x * scale[n,c] + shift[n,c])
//   mis_bn_bwd_finalize : with g = dy*(y>0), S1 = sum g, S2 = sum g*z (mis_gn_bwd_stats):
//                         dbeta = sum_n S1, dgamma = rstd * sum_n (S2 - mean*S1),
//                         dz = p*g + q*z + r,  p = gamma*rstd, q = -gamma*rstd^2*dgamma/M, r = -q*mean - gamma*rstd*dbeta/M
//                         (applied by mis_gn_bwd_apply); in eval mode dz = gamma*rstd*g.
#include "common.hpp"

__global__ void bn_fwd_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ sq, int N, int C, double count_total,
                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                       float* running_mean, float* running_var, int training, float* __restrict__ scale,
                                       float* __restrict__ shift, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double mean, var;
    if (training) {
        double s = 0.0, q = 0.0;
        for (int n = 0; n < N; ++n) {
            s += (double)sum[n * C + c];
            q += (double)sq[n * C + c];
        }
        mean = s / count_total;
        var = q / count_total - mean * mean;
        if (var < 0.0) var = 0.0;
        if (running_mean != nullptr) {
            const double unbiased = count_total > 1.0 ? var * count_total / (count_total - 1.0) : var;
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
        }
    } else {
        mean = running_mean[c];
        var = running_var[c];
    }
    const double rstd = 1.0 / sqrt(var + (double)eps);
    mean_out[c] = (float)mean;
    rstd_out[c] = (float)rstd;
    const double a = rstd * (double)gamma[c];
    const float sc = (float)a, sh = (float)((double)beta[c] - mean * a);
    for (int n = 0; n < N; ++n) {
        scale[n * C + c] = sc;
        shift[n * C + c] = sh;
    }
}

extern "C" int mis_bn_fwd_finalize(const float* sum, const float* sumsq, int N, int C, double count_total, const float* gamma, const float* beta,
                                   float eps, float momentum, float* running_mean, float* running_var, int training, float* scale, float* shift,
                                   float* mean, float* rstd, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(gamma && beta && scale && shift && mean && rstd, MIS_EINVAL, "bn_fwd_finalize: null pointer");
    MIS_REQUIRE(N > 0 && C > 0 && count_total > 0.0, MIS_EINVAL, "bn_fwd_finalize: sizes");
    MIS_REQUIRE(training ? (sum && sumsq) : (running_mean && running_var), MIS_EINVAL,
                "bn_fwd_finalize: training needs batch sums, eval needs running statistics");
    MIS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), MIS_EINVAL, "bn_fwd_finalize: running_mean/running_var");
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), sum, sumsq, N, C, count_total,
                       gamma, beta, eps, momentum, running_mean, running_var, training, scale, shift, mean, rstd);
    MIS_LAUNCH_CHECK("bn_fwd_finalize");
    return MIS_OK;
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ S1, const float* __restrict__ S2, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, const float* __restrict__ gamma, int N, int C, double count_total, int training,
                                       float* __restrict__ p, float* __restrict__ q, float* __restrict__ r, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < N; ++n) {
        s1 += (double)S1[n * C + c];
        s2 += (double)S2[n * C + c];
    }
    const double mu = mean[c], rs = rstd[c], ga = gamma[c];
    const double dg = rs * (s2 - mu * s1), db = s1;
    dgamma[c] = (float)dg;
    dbeta[c] = (float)db;
    const double pp = ga * rs;
    double qq = 0.0, rr = 0.0;
    if (training) {
        qq = -ga * rs * rs * dg / count_total;
        rr = -qq * mu - ga * rs * db / count_total;
    }
    for (int n = 0; n < N; ++n) {
        p[n * C + c] = (float)pp;
        q[n * C + c] = (float)qq;
        r[n * C + c] = (float)rr;
    }
}

extern "C" int mis_bn_bwd_finalize(const float* S1, const float* S2, const float* mean, const float* rstd, const float* gamma, int N, int C,
                                   double count_total, int training, float* p, float* q, float* r, float* dgamma, float* dbeta, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(S1 && S2 && mean && rstd && gamma && p && q && r && dgamma && dbeta, MIS_EINVAL, "bn_bwd_finalize: null pointer");
    MIS_REQUIRE(N > 0 && C > 0 && count_total > 0.0, MIS_EINVAL, "bn_bwd_finalize: sizes");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), S1, S2, mean, rstd, gamma, N, C,
                       count_total, training, p, q, r, dgamma, dbeta);
    MIS_LAUNCH_CHECK("bn_bwd_finalize");
    return MIS_OK;
}

// y[n][pix][c] = [relu](x * scale[n][c] + shift[n][c])
template <typename T>
__global__ __launch_bounds__(256) void affine_act_kernel(const T* __restrict__ x, int x_ld, T* __restrict__ y, int y_ld, int N, long long npix, int C,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, int relu) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC;
    const long long total = (long long)N * npix * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long long pp = i / nch;
        const int n = (int)(pp / npix);
        float f[EPC];
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(x + (size_t)pp * x_ld + (size_t)ch * EPC), f);
        const float* sc = scale + (size_t)n * C + ch * EPC;
        const float* sh = shift + (size_t)n * C + ch * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float v = fmaf(f[e], sc[e], sh[e]);
            f[e] = relu ? fmaxf(v, 0.f) : v;
        }
        *reinterpret_cast<u32x4*>(y + (size_t)pp * y_ld + (size_t)ch * EPC) = pack_chunk<T>(f);
    }
}

extern "C" int mis_affine_act(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, long long npix, int C, const float* scale,
                              const float* shift, int relu, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "affine_act: bad dtype %d", dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(x && y && scale && shift, MIS_EINVAL, "affine_act: null pointer");
    MIS_REQUIRE(N > 0 && npix > 0 && C > 0 && C % EPC == 0 && x_ld % EPC == 0 && y_ld % EPC == 0, MIS_EINVAL, "affine_act: sizes / alignment");
    long long blocks = ((long long)N * npix * (C / EPC) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(affine_act_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)x, x_ld, (__bf16*)y, y_ld, N, npix, C, scale,
                           shift, relu);
    else
        hipLaunchKernelGGL(affine_act_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, x_ld, (float*)y, y_ld, N, npix, C, scale,
                           shift, relu);
    MIS_LAUNCH_CHECK("affine_act");
    return MIS_OK;
}
