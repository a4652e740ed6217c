// Stand-alone BCE + Dice loss on logits (reference model/unet3d/losses.py: `BCEDiceLoss` :167-178, `DiceLoss` :119-129,
// `_AbstractDiceLoss` :83-116, `compute_per_channel_dice` :7-33, `flatten` :258-270), forward and backward, for gfx950.
//   loss = alpha * mean(BCEWithLogits(x, t)) + beta * (1 - mean_c dice_c),  dice_c = 2 * sum(s*t) / clamp(sum(s^2) + sum(t^2), 1e-6),
//   s = sigmoid(x) (normalize = 1) or x itself (normalize = 0: DiceLoss(normalization='none'), compute_per_channel_dice on probabilities),
//   sums over every sample and voxel of channel c.  x, t: fp32 (N, C, S) contiguous.
// (The fused train step computes the same loss inside the 1x1-head kernel, head_loss.hip; this entry serves the nn.Module surface:
//  an external criterion applied to logits, and the HF wrapper's double-sigmoid quirk.)
// Two HBM passes forward (per-block partial sums in a fixed order -> one finalize block), one pass backward.
#include "common.hpp"

constexpr int BD_BLOCKS = 256;    // partial-sum blocks per channel

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// grid (BD_BLOCKS, C): part[c][b][4] = {sum bce, sum s*t, sum s*s, sum t*t} over this block's share of channel c
__global__ __launch_bounds__(256) void bcedice_partial_kernel(const float* __restrict__ x, const float* __restrict__ t, int N, int C, long long S,
                                                              int normalize, double* __restrict__ part) {
    __shared__ double red[4][4];
    const int c = blockIdx.y;
    double b = 0.0, i1 = 0.0, a2 = 0.0, t2 = 0.0;
    const long long per = (long long)N * S;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < per; e += (long long)gridDim.x * 256) {
        const long long n = e / S, v = e - n * S;
        const size_t idx = ((size_t)n * C + c) * S + v;
        const float xv = x[idx], tv = t[idx];
        const float s = normalize ? sigmoidf_(xv) : xv;
        if (normalize) b += (double)(fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv))));
        i1 += (double)(s * tv);
        a2 += (double)(s * s);
        t2 += (double)(tv * tv);
    }
    b = wave_sum_d(b); i1 = wave_sum_d(i1); a2 = wave_sum_d(a2); t2 = wave_sum_d(t2);
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        red[0][w] = b; red[1][w] = i1; red[2][w] = a2; red[3][w] = t2;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        part[((size_t)c * gridDim.x + blockIdx.x) * 4 + k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
    }
}

// out: [0] loss, [1] bce mean, [2 + 4c .. ] = {I_c, A2_c, T2_c, dice_c}
__global__ void bcedice_finalize_kernel(const double* __restrict__ part, int nb, int N, int C, long long S, float alpha, float beta, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double bce = 0.0, dmean = 0.0;
    for (int c = 0; c < C; ++c) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int b = 0; b < nb; ++b)
            for (int k = 0; k < 4; ++k) s[k] += part[((size_t)c * nb + b) * 4 + k];
        bce += s[0];
        const float I = (float)s[1], den = fmaxf((float)s[2] + (float)s[3], 1e-6f);
        const float dice = 2.0f * (I / den);
        out[2 + 4 * c + 0] = I;
        out[2 + 4 * c + 1] = (float)s[2];
        out[2 + 4 * c + 2] = (float)s[3];
        out[2 + 4 * c + 3] = dice;
        dmean += (double)dice;
    }
    const float bm = (float)(bce / ((double)N * C * (double)S));
    out[1] = bm;
    out[0] = alpha * bm + beta * (1.0f - (float)(dmean / C));
}

__global__ __launch_bounds__(256) void bcedice_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, int N, int C, long long S, float alpha,
                                                          float beta, int normalize, const float* __restrict__ sums, const float* __restrict__ gout, float* __restrict__ dx) {
    const long long total = (long long)N * C * S;
    const float g = gout[0];
    const float kb = alpha / (float)((double)N * C * (double)S), kd = beta / (float)C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)((e / S) % C);
        const float xv = x[e], tv = t[e];
        const float s = normalize ? sigmoidf_(xv) : xv, ds = normalize ? s * (1.f - s) : 1.f;
        const float I = sums[2 + 4 * c], D = sums[2 + 4 * c + 1] + sums[2 + 4 * c + 2];
        float ddice;                                              // d dice_c / dx
        if (D > 1e-6f) ddice = 2.f * (tv * D - 2.f * I * s) * ds / (D * D);
        else ddice = 2.f * tv * ds / 1e-6f;                       // clamp active: the denominator is the constant epsilon
        dx[e] = g * ((normalize ? kb * (s - tv) : 0.f) - kd * ddice);
    }
}

extern "C" size_t mis_bcedice_workspace_bytes(int C) { return (size_t)C * BD_BLOCKS * 4 * sizeof(double); }

extern "C" int mis_bcedice_fwd(const float* x, const float* t, int N, int C, long long S, float alpha, float beta, int normalize, void* workspace,
                               float* out, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(x && t && workspace && out, MIS_EINVAL, "bcedice_fwd: null pointer");
    MIS_REQUIRE(N > 0 && C > 0 && C <= 64 && S > 0, MIS_EINVAL, "bcedice_fwd: sizes (N %d, C %d)", N, C);
    MIS_REQUIRE(normalize || alpha == 0.f, MIS_EINVAL, "bcedice_fwd: the BCE term is defined on logits (normalize = 1)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long per = (long long)N * S;
    int nb = (int)((per + 256 * 8 - 1) / (256 * 8));
    if (nb < 1) nb = 1;
    if (nb > BD_BLOCKS) nb = BD_BLOCKS;
    hipLaunchKernelGGL(bcedice_partial_kernel, dim3(nb, C), dim3(256), 0, s, x, t, N, C, S, normalize, reinterpret_cast<double*>(workspace));
    hipLaunchKernelGGL(bcedice_finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, nb, N, C, S, alpha, beta, out);
    MIS_LAUNCH_CHECK("bcedice_fwd");
    return MIS_OK;
}

extern "C" int mis_bcedice_bwd(const float* x, const float* t, int N, int C, long long S, float alpha, float beta, int normalize, const float* sums,
                               const float* grad_out, float* dx, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(x && t && sums && grad_out && dx, MIS_EINVAL, "bcedice_bwd: null pointer");
    MIS_REQUIRE(N > 0 && C > 0 && C <= 64 && S > 0, MIS_EINVAL, "bcedice_bwd: sizes");
    const long long total = (long long)N * C * S;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bcedice_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, t, N, C, S, alpha, beta, normalize,
                       sums, grad_out, dx);
    MIS_LAUNCH_CHECK("bcedice_bwd");
    return MIS_OK;
}

// ---- the other criteria of the reference's loss factory that apply to this path (model/unet3d/losses.py:309-346 `_create_loss`) -----------------
// CrossEntropyLoss (nn.CrossEntropyLoss(weight=None, ignore_index)): logits fp32 (N, C, S), labels int64 (N, S); mean over the non-ignored voxels.
// MSELoss / L1Loss / SmoothL1Loss (beta 1): elementwise, mean.  Forward: per-block double partials, summed in a fixed order (bitwise
// reproducible); backward: one elementwise pass scaled by the incoming scalar gradient.
constexpr int LS_BLOCKS = 1024;
constexpr int LS_MAXC = 32;

__global__ __launch_bounds__(256) void ce3d_partial_kernel(const float* __restrict__ x, const long long* __restrict__ lab, int N, int C, long long S,
                                                           long long ignore_index, double* __restrict__ part) {
    double ls = 0.0, cnt = 0.0;
    const long long total = (long long)N * S;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long l = lab[i];
        if (l == ignore_index || l < 0 || l >= C) continue;
        const long long n = i / S, s = i - n * S;
        const float* p = x + (size_t)n * C * S + s;
        float m = p[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, p[(size_t)c * S]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(p[(size_t)c * S] - m);
        ls += (double)(logf(se) + m - p[(size_t)l * S]);
        cnt += 1.0;
    }
    __shared__ double r0[256], r1[256];
    r0[threadIdx.x] = ls;
    r1[threadIdx.x] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 256; ++k) {
            ls += r0[k];
            cnt += r1[k];
        }
        part[2 * blockIdx.x] = ls;
        part[2 * blockIdx.x + 1] = cnt;
    }
}

__global__ void loss_finalize_kernel(const double* __restrict__ part, int nb, int stride, double fixed_count, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0, c = 0.0;
    for (int k = 0; k < nb; ++k) {
        s += part[(size_t)stride * k];
        if (stride > 1) c += part[(size_t)stride * k + 1];
    }
    if (stride == 1) c = fixed_count;
    out[0] = (float)(s / c);          // all targets ignored: 0/0 = nan, like torch
    out[1] = (float)c;
}

__global__ __launch_bounds__(256) void ce3d_bwd_kernel(const float* __restrict__ x, const long long* __restrict__ lab, int N, int C, long long S,
                                                       long long ignore_index, const float* __restrict__ fwd_out, const float* __restrict__ grad_out,
                                                       float* __restrict__ dx) {
    const float scale = grad_out[0] / fwd_out[1];
    const long long total = (long long)N * S;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long l = lab[i];
        const long long n = i / S, s = i - n * S;
        const float* p = x + (size_t)n * C * S + s;
        float* q = dx + (size_t)n * C * S + s;
        if (l == ignore_index || l < 0 || l >= C) {
            for (int c = 0; c < C; ++c) q[(size_t)c * S] = 0.f;
            continue;
        }
        float m = p[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, p[(size_t)c * S]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(p[(size_t)c * S] - m);
        const float inv = 1.f / se;
        for (int c = 0; c < C; ++c) q[(size_t)c * S] = scale * (expf(p[(size_t)c * S] - m) * inv - (c == l ? 1.f : 0.f));
    }
}

__device__ __forceinline__ float pl_value(int kind, float d) {
    const float a = fabsf(d);
    return kind == 0 ? d * d : kind == 1 ? a : (a < 1.f ? 0.5f * d * d : a - 0.5f);
}
__device__ __forceinline__ float pl_grad(int kind, float d) {
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    return kind == 0 ? 2.f * d : kind == 1 ? sg : (fabsf(d) < 1.f ? d : sg);
}

__global__ __launch_bounds__(256) void pointloss_partial_kernel(int kind, const float* __restrict__ x, const float* __restrict__ t, long long n,
                                                                double* __restrict__ part) {
    double ls = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) ls += (double)pl_value(kind, x[i] - t[i]);
    __shared__ double r0[256];
    r0[threadIdx.x] = ls;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 256; ++k) ls += r0[k];
        part[blockIdx.x] = ls;
    }
}

__global__ __launch_bounds__(256) void pointloss_bwd_kernel(int kind, const float* __restrict__ x, const float* __restrict__ t, long long n,
                                                            const float* __restrict__ grad_out, float* __restrict__ dx) {
    const float scale = grad_out[0] / (float)n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dx[i] = scale * pl_grad(kind, x[i] - t[i]);
}

static unsigned ls_grid(long long total) {
    long long b = (total + 255) / 256;
    if (b > LS_BLOCKS) b = LS_BLOCKS;
    return (unsigned)(b < 1 ? 1 : b);
}

extern "C" size_t mis_loss_workspace_bytes(void) { return (size_t)2 * LS_BLOCKS * sizeof(double); }

extern "C" int mis_ce3d_fwd(const float* logits, const long long* labels, int N, int C, long long S, long long ignore_index, void* workspace, float* out,
                            void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(logits && labels && workspace && out && N > 0 && S > 0 && C >= 2 && C <= LS_MAXC, MIS_EINVAL, "ce3d_fwd: arguments (2 <= C <= %d)", LS_MAXC);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = ls_grid((long long)N * S);
    hipLaunchKernelGGL(ce3d_partial_kernel, dim3(g), dim3(256), 0, st, logits, labels, N, C, S, ignore_index, (double*)workspace);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, (int)g, 2, 0.0, out);
    MIS_LAUNCH_CHECK("ce3d_fwd");
    return MIS_OK;
}

extern "C" int mis_ce3d_bwd(const float* logits, const long long* labels, int N, int C, long long S, long long ignore_index, const float* fwd_out,
                            const float* grad_out, float* dx, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(logits && labels && fwd_out && grad_out && dx && N > 0 && S > 0 && C >= 2 && C <= LS_MAXC, MIS_EINVAL, "ce3d_bwd: arguments");
    hipLaunchKernelGGL(ce3d_bwd_kernel, dim3(ls_grid((long long)N * S) * 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits, labels, N, C, S,
                       ignore_index, fwd_out, grad_out, dx);
    MIS_LAUNCH_CHECK("ce3d_bwd");
    return MIS_OK;
}

/* kind: 0 MSELoss, 1 L1Loss, 2 SmoothL1Loss (beta 1); mean over n elements; out[2] = {loss, n} */
extern "C" int mis_pointloss_fwd(int kind, const float* x, const float* t, long long n, void* workspace, float* out, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(x && t && workspace && out && n > 0 && kind >= 0 && kind <= 2, MIS_EINVAL, "pointloss_fwd: arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = ls_grid(n);
    hipLaunchKernelGGL(pointloss_partial_kernel, dim3(g), dim3(256), 0, st, kind, x, t, n, (double*)workspace);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, (int)g, 1, (double)n, out);
    MIS_LAUNCH_CHECK("pointloss_fwd");
    return MIS_OK;
}

extern "C" int mis_pointloss_bwd(int kind, const float* x, const float* t, long long n, const float* grad_out, float* dx, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(x && t && grad_out && dx && n > 0 && kind >= 0 && kind <= 2, MIS_EINVAL, "pointloss_bwd: arguments");
    hipLaunchKernelGGL(pointloss_bwd_kernel, dim3(ls_grid(n) * 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), kind, x, t, n, grad_out, dx);
    MIS_LAUNCH_CHECK("pointloss_bwd");
    return MIS_OK;
}
