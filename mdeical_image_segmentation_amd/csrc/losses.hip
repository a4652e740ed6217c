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
