// SegmentationLoss of the UNet 3+ path (reference model/unet2d/loss.py:21-70): F1Loss + MSSSIMLoss + IoULoss on (logits, targets),
// single channel, forward and backward, for gfx950.
//   p = sigmoid(logits);  F1 (:45-56) and IoU (:32-42) are functions of three global sums (sum p*t, sum p, sum t), epsilon 1e-7;
//   MS-SSIM (:21-29) = pytorch_msssim 1.0.0 `MS_SSIM(data_range=1.0, size_average=True, channel=1)` (third-party; algorithm restated from the
//   package: 11-tap Gaussian (sigma 1.5) "valid" filtering of x, y, x^2, y^2, xy; cs and ssim maps averaged per image on 5 scales linked by
//   2x2 average pooling (padding = size % 2, pad counted); relu; product of powers [0.0448, 0.2856, 0.3001, 0.2363, 0.1333]; batch mean).
// Backward: per scale the three adjoint maps a1 = dL/dG(xy), a2 = dL/dG(x^2), m1 = dL/dG(x) are formed on the filtered grid and gathered
// back through the transposed filter, dX = y*G'(a1) + 2x*G'(a2) + G'(m1); then through the pooling pyramid and the sigmoid.
// Direct 121-tap loops: the single-channel maps are small next to the network (N x 512 x 512 floats) and L1/L2 resident.
#include <math.h>

#include "common.hpp"

constexpr int SL_WIN = 11, SL_LEVELS = 5, SL_BLOCKS = 256;
constexpr float SL_C1 = 1e-4f, SL_C2 = 9e-4f;

struct SlWin {
    float w[SL_WIN];
};
struct SlDims {
    int H[SL_LEVELS], W[SL_LEVELS];
    long long off[SL_LEVELS];      // float offset of level l inside a pyramid buffer (per whole batch)
    long long total;               // floats per pyramid
};

static SlWin sl_window() {
    SlWin g;
    float s = 0.f;
    for (int i = 0; i < SL_WIN; ++i) {
        const float c = (float)(i - SL_WIN / 2);
        g.w[i] = expf(-(c * c) / (2.f * 1.5f * 1.5f));
        s += g.w[i];
    }
    for (int i = 0; i < SL_WIN; ++i) g.w[i] /= s;
    return g;
}

static SlDims sl_dims(int N, int H, int W) {
    SlDims d;
    long long off = 0;
    for (int l = 0; l < SL_LEVELS; ++l) {
        d.H[l] = H;
        d.W[l] = W;
        d.off[l] = off;
        off += (long long)N * H * W;
        H = (H + 2 * (H % 2) - 2) / 2 + 1;
        W = (W + 2 * (W % 2) - 2) / 2 + 1;
    }
    d.total = off;
    return d;
}

// X0 = sigmoid(logits), Y0 = target; per-block partial sums of p*t, p, t
__global__ __launch_bounds__(256) void sl_sigmoid_kernel(const float* __restrict__ logits, const float* __restrict__ target, long long n, float* __restrict__ X,
                                                         float* __restrict__ Y, double* __restrict__ part) {
    __shared__ double red[3][4];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float p = 1.0f / (1.0f + expf(-logits[i])), t = target[i];
        X[i] = p;
        Y[i] = t;
        s0 += (double)(p * t);
        s1 += (double)p;
        s2 += (double)t;
    }
    s0 = wave_sum_d(s0); s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; red[2][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 3) part[blockIdx.x * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// 2x2 average pooling, stride 2, padding (H%2, W%2), divisor always 4
__global__ __launch_bounds__(256) void sl_avgpool_kernel(const float* __restrict__ in, int N, int H, int W, float* __restrict__ out, int OH, int OW) {
    const int ph = H % 2, pw = W % 2;
    const long long total = (long long)N * OH * OW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % OW);
        const long long r = i / OW;
        const int oy = (int)(r % OH), n = (int)(r / OH);
        float s = 0.f;
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const int y = 2 * oy - ph + dy, x = 2 * ox - pw + dx;
                if (y >= 0 && y < H && x >= 0 && x < W) s += in[((size_t)n * H + y) * W + x];
            }
        out[i] = s * 0.25f;
    }
}

struct SlStat {
    float mu1, mu2, s11, s22, s12;
};
__device__ __forceinline__ SlStat sl_filter(const float* __restrict__ X, const float* __restrict__ Y, int W, int y0, int x0, const SlWin& g) {
    // "valid" separable Gaussian evaluated as one 11x11 window with top-left corner (y0, x0); rows first, then columns, like the package
    SlStat o{0.f, 0.f, 0.f, 0.f, 0.f};
    for (int dy = 0; dy < SL_WIN; ++dy) {
        float r1 = 0.f, r2 = 0.f, r11 = 0.f, r22 = 0.f, r12 = 0.f;
        const float* xr = X + (size_t)(y0 + dy) * W + x0;
        const float* yr = Y + (size_t)(y0 + dy) * W + x0;
#pragma unroll
        for (int dx = 0; dx < SL_WIN; ++dx) {
            const float a = xr[dx], b = yr[dx], w = g.w[dx];
            r1 = fmaf(w, a, r1);
            r2 = fmaf(w, b, r2);
            r11 = fmaf(w, a * a, r11);
            r22 = fmaf(w, b * b, r22);
            r12 = fmaf(w, a * b, r12);
        }
        const float wy = g.w[dy];
        o.mu1 = fmaf(wy, r1, o.mu1);
        o.mu2 = fmaf(wy, r2, o.mu2);
        o.s11 = fmaf(wy, r11, o.s11);
        o.s22 = fmaf(wy, r22, o.s22);
        o.s12 = fmaf(wy, r12, o.s12);
    }
    return o;
}

// per image: sums of the cs map and of the ssim map over the (H-10) x (W-10) filtered grid.  grid (blocks, N); part[n][b][2]
__global__ __launch_bounds__(256) void sl_level_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Y, int H, int W, SlWin g,
                                                           double* __restrict__ part) {
    __shared__ double red[2][4];
    const int n = blockIdx.y, MH = H - SL_WIN + 1, MW = W - SL_WIN + 1;
    const float* Xn = X + (size_t)n * H * W;
    const float* Yn = Y + (size_t)n * H * W;
    double scs = 0.0, sss = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < MH * MW; i += gridDim.x * 256) {
        const int my = i / MW, mx = i - my * MW;
        const SlStat f = sl_filter(Xn, Yn, W, my, mx, g);
        const float mu11 = f.mu1 * f.mu1, mu22 = f.mu2 * f.mu2, mu12 = f.mu1 * f.mu2;
        const float cs = (2.f * (f.s12 - mu12) + SL_C2) / ((f.s11 - mu11) + (f.s22 - mu22) + SL_C2);
        const float ss = ((2.f * mu12 + SL_C1) / (mu11 + mu22 + SL_C1)) * cs;
        scs += (double)cs;
        sss += (double)ss;
    }
    scs = wave_sum_d(scs);
    sss = wave_sum_d(sss);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = scs;
        red[1][threadIdx.x >> 6] = sss;
    }
    __syncthreads();
    if (threadIdx.x < 2) part[((size_t)n * gridDim.x + blockIdx.x) * 2 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// scalars: out[0] loss, [1] f1 loss, [2] ms-ssim loss, [3] iou loss, [4] c_pt, [5] c_p (dLoss/d(sum p*t), dLoss/d(sum p) of F1 + IoU),
// state[n][l]: {cs mean, ssim mean, dLoss/d(that mean)} for the backward
__global__ void sl_finalize_kernel(const double* __restrict__ part_sig, int nb_sig, const double* __restrict__ part_lvl, int nb_lvl, int N, SlDims d,
                                   float w_f1, float w_ms, float w_iou, float* __restrict__ out, float* __restrict__ state, int with_ms) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double eps = 1e-7;
    double spt = 0.0, sp = 0.0, st = 0.0;
    for (int b = 0; b < nb_sig; ++b) {
        spt += part_sig[b * 3 + 0];
        sp += part_sig[b * 3 + 1];
        st += part_sig[b * 3 + 2];
    }
    // the reference evaluates these in float32
    const float TP = (float)spt, SP = (float)sp, ST = (float)st, e = 1e-7f;
    const float pr = TP / (SP + e), rc = TP / (ST + e);
    const float f1 = 2.f * (pr * rc) / (pr + rc + e);
    const float uni = SP + ST - TP;
    const float iou = (TP + e) / (uni + e);
    {   // derivatives in double
        const double p = spt / (sp + eps), r = spt / (st + eps), den = p + r + eps;
        const double fp = 2.0 * r * (r + eps) / (den * den), fr = 2.0 * p * (p + eps) / (den * den);
        const double df_dpt = fp / (sp + eps) + fr / (st + eps), df_dp = -fp * spt / ((sp + eps) * (sp + eps));
        const double u = sp + st - spt;
        const double di_dpt = ((u + eps) + (spt + eps)) / ((u + eps) * (u + eps)), di_dp = -(spt + eps) / ((u + eps) * (u + eps));
        out[4] = (float)(-((double)w_f1 * df_dpt + (double)w_iou * di_dpt));
        out[5] = (float)(-((double)w_f1 * df_dp + (double)w_iou * di_dp));
    }
    const float wts[SL_LEVELS] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};
    float msum = with_ms ? 0.f : (float)N;          // (without the MS-SSIM term: lm = 0; the pyramid was not built, F1 / IoU alone take any image size)
    for (int n = 0; n < (with_ms ? N : 0); ++n) {
        float val = 1.f, term[SL_LEVELS], base[SL_LEVELS];
        for (int l = 0; l < SL_LEVELS; ++l) {
            const long long npx = (long long)(d.H[l] - SL_WIN + 1) * (d.W[l] - SL_WIN + 1);
            double a = 0.0, c = 0.0;
            const double* pl = part_lvl + ((size_t)l * N + n) * nb_lvl * 2;
            for (int b = 0; b < nb_lvl; ++b) {
                a += pl[b * 2 + 0];
                c += pl[b * 2 + 1];
            }
            const float csm = (float)(a / (double)npx), ssm = (float)(c / (double)npx);
            state[(n * SL_LEVELS + l) * 3 + 0] = csm;
            state[(n * SL_LEVELS + l) * 3 + 1] = ssm;
            base[l] = fmaxf(l < SL_LEVELS - 1 ? csm : ssm, 0.f);
            term[l] = powf(base[l], wts[l]);
            val *= term[l];
        }
        msum += val;
        for (int l = 0; l < SL_LEVELS; ++l)     // d(1 - mean_n val)/d(base_l) = -(1/N) * val * w_l / base_l   (0 where relu cut it)
            state[(n * SL_LEVELS + l) * 3 + 2] = base[l] > 0.f ? -(w_ms / (float)N) * val * wts[l] / base[l] : 0.f;
    }
    const float lm = 1.f - msum / (float)N;
    out[1] = 1.f - f1;
    out[2] = lm;
    out[3] = 1.f - iou;
    out[0] = w_f1 * (1.f - f1) + w_ms * lm + w_iou * (1.f - iou);
}

// adjoint maps of one level on its filtered grid: a1 = dL/dG(xy), a2 = dL/dG(x^2), m1 = dL/dG(x).   grid (blocks, N)
__global__ __launch_bounds__(256) void sl_level_bwd_maps_kernel(const float* __restrict__ X, const float* __restrict__ Y, int H, int W, SlWin g, int level,
                                                                const float* __restrict__ state, float* __restrict__ A1, float* __restrict__ A2,
                                                                float* __restrict__ M1) {
    const int n = blockIdx.y, MH = H - SL_WIN + 1, MW = W - SL_WIN + 1;
    const float* Xn = X + (size_t)n * H * W;
    const float* Yn = Y + (size_t)n * H * W;
    const float gmean = state[(n * SL_LEVELS + level) * 3 + 2] / (float)((long long)MH * MW);      // dL/d(map value)
    const bool last = level == SL_LEVELS - 1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < MH * MW; i += gridDim.x * 256) {
        const int my = i / MW, mx = i - my * MW;
        const SlStat f = sl_filter(Xn, Yn, W, my, mx, g);
        const float mu11 = f.mu1 * f.mu1, mu22 = f.mu2 * f.mu2, mu12 = f.mu1 * f.mu2;
        const float A = 2.f * (f.s12 - mu12) + SL_C2, B = (f.s11 - mu11) + (f.s22 - mu22) + SL_C2;
        const float cs = A / B;
        float gcs = gmean, gmu1 = 0.f;
        if (last) {                                   // ssim = lum * cs
            const float Dn = mu11 + mu22 + SL_C1, lum = (2.f * mu12 + SL_C1) / Dn;
            gmu1 = gmean * cs * (2.f * f.mu2 * Dn - (2.f * mu12 + SL_C1) * 2.f * f.mu1) / (Dn * Dn);
            gcs = gmean * lum;
        }
        const float a1 = gcs * 2.f / B;               // via sigma12 = G(xy) - mu1*mu2
        const float a2 = -gcs * A / (B * B);          // via sigma1^2 = G(x^2) - mu1^2
        const size_t o = (size_t)n * MH * MW + i;
        A1[o] = a1;
        A2[o] = a2;
        M1[o] = gmu1 - a1 * f.mu2 - a2 * 2.f * f.mu1;
    }
}

// dX[p] (+)= y[p]*G'(a1)[p] + 2x[p]*G'(a2)[p] + G'(m1)[p] + pooled-gradient from the coarser level
__global__ __launch_bounds__(256) void sl_level_bwd_gather_kernel(const float* __restrict__ X, const float* __restrict__ Y, int N, int H, int W, SlWin g,
                                                                  const float* __restrict__ A1, const float* __restrict__ A2, const float* __restrict__ M1,
                                                                  const float* __restrict__ dXc /*coarser level gradient or null*/, int CH, int CW,
                                                                  float* __restrict__ dX) {
    const int MH = H - SL_WIN + 1, MW = W - SL_WIN + 1;
    const long long total = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W);
        const long long r = i / W;
        const int y = (int)(r % H), n = (int)(r / H);
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int dy = 0; dy < SL_WIN; ++dy) {
            const int my = y - dy;
            if (my < 0 || my >= MH) continue;
            const float wy = g.w[dy];
            for (int dx = 0; dx < SL_WIN; ++dx) {
                const int mx = x - dx;
                if (mx < 0 || mx >= MW) continue;
                const float w = wy * g.w[dx];
                const size_t o = ((size_t)n * MH + my) * MW + mx;
                s1 = fmaf(w, A1[o], s1);
                s2 = fmaf(w, A2[o], s2);
                s3 = fmaf(w, M1[o], s3);
            }
        }
        float v = Y[i] * s1 + 2.f * X[i] * s2 + s3;
        if (dXc != nullptr) {                          // adjoint of the 2x2 average pooling (padding H%2, W%2)
            const int oy = (y + H % 2) >> 1, ox = (x + W % 2) >> 1;
            v += 0.25f * dXc[((size_t)n * CH + oy) * CW + ox];
        }
        dX[i] = v;
    }
}

__global__ __launch_bounds__(256) void sl_sigmoid_bwd_kernel(const float* __restrict__ X0, const float* __restrict__ T, const float* __restrict__ dX0,
                                                             const float* __restrict__ out, const float* __restrict__ gout, long long n,
                                                             float* __restrict__ dlogits) {
    const float g = gout[0], cpt = out[4], cp = out[5];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float p = X0[i];
        dlogits[i] = g * ((dX0 != nullptr ? dX0[i] : 0.f) + cpt * T[i] + cp) * p * (1.f - p);
    }
}

static unsigned sl_grid(long long total) {
    long long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// workspace (floats): X pyramid | Y pyramid | dX pyramid | A1 | A2 | M1 (level-0 filtered grid) | state [N][5][3] | out-of-band doubles
static size_t sl_ws_floats(int N, int H, int W, const SlDims& d) {
    return (size_t)3 * d.total + (size_t)3 * N * H * W + (size_t)N * SL_LEVELS * 3 + 16;
}
extern "C" size_t mis_segloss_workspace_bytes(int N, int H, int W) {
    const SlDims d = sl_dims(N, H, W);
    return sl_ws_floats(N, H, W, d) * sizeof(float) + ((size_t)SL_BLOCKS * 3 + (size_t)SL_LEVELS * N * SL_BLOCKS * 2) * sizeof(double) + 64;
}

static int sl_check(const char* what, int N, int H, int W, bool with_ms) {
    MIS_REQUIRE(N > 0 && N <= 4096 && H > 0 && W > 0, MIS_EINVAL, "%s: batch %d, image %d x %d", what, N, H, W);
    if (!with_ms) return MIS_OK;          // F1Loss / IoULoss alone (reference model/unet2d/loss.py:32-56) are global sums: any image size
    const int smaller = H < W ? H : W;
    MIS_REQUIRE(smaller > (SL_WIN - 1) * 16, MIS_EUNSUPPORTED, "%s: image side %d must be larger than %d (4 down-samplings of MS-SSIM)", what, smaller,
                (SL_WIN - 1) * 16);
    return MIS_OK;
}

extern "C" int mis_segloss_fwd(const float* logits, const float* target, int N, int H, int W, float w_f1, float w_msssim, float w_iou, void* workspace,
                               float* out /*[8]*/, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(logits && target && workspace && out, MIS_EINVAL, "segloss_fwd: null pointer");
    const bool ms = w_msssim != 0.f;
    if (int rc = sl_check("segloss_fwd", N, H, W, ms)) return rc;
    const SlDims d = sl_dims(N, H, W);
    const SlWin g = sl_window();
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(workspace);
    float* X = ws;
    float* Y = ws + d.total;
    float* state = ws + 3 * d.total + (size_t)3 * N * H * W;
    double* dpart = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + ((sl_ws_floats(N, H, W, d) * sizeof(float) + 63) / 64) * 64);
    double* part_sig = dpart;
    double* part_lvl = dpart + SL_BLOCKS * 3;
    const long long n0 = (long long)N * H * W;
    int nb_sig = (int)sl_grid(n0);
    if (nb_sig > SL_BLOCKS) nb_sig = SL_BLOCKS;
    hipLaunchKernelGGL(sl_sigmoid_kernel, dim3(nb_sig), dim3(256), 0, s, logits, target, n0, X, Y, part_sig);
    for (int l = 1; ms && l < SL_LEVELS; ++l) {
        hipLaunchKernelGGL(sl_avgpool_kernel, dim3(sl_grid((long long)N * d.H[l] * d.W[l])), dim3(256), 0, s, (const float*)(X + d.off[l - 1]), N, d.H[l - 1],
                           d.W[l - 1], X + d.off[l], d.H[l], d.W[l]);
        hipLaunchKernelGGL(sl_avgpool_kernel, dim3(sl_grid((long long)N * d.H[l] * d.W[l])), dim3(256), 0, s, (const float*)(Y + d.off[l - 1]), N, d.H[l - 1],
                           d.W[l - 1], Y + d.off[l], d.H[l], d.W[l]);
    }
    const int nb_lvl = 64;
    for (int l = 0; ms && l < SL_LEVELS; ++l)
        hipLaunchKernelGGL(sl_level_fwd_kernel, dim3(nb_lvl, N), dim3(256), 0, s, (const float*)(X + d.off[l]), (const float*)(Y + d.off[l]), d.H[l], d.W[l], g,
                           part_lvl + (size_t)l * N * nb_lvl * 2);
    hipLaunchKernelGGL(sl_finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)part_sig, nb_sig, (const double*)part_lvl, nb_lvl, N, d, w_f1, w_msssim,
                       w_iou, out, state, ms ? 1 : 0);
    MIS_LAUNCH_CHECK("segloss_fwd");
    return MIS_OK;
}

// after mis_segloss_fwd on the same workspace / out; with_msssim = (the forward's w_msssim != 0)
extern "C" int mis_segloss_bwd(const float* target, int N, int H, int W, void* workspace, const float* out, const float* grad_out, float* dlogits,
                               int with_msssim, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(target && workspace && out && grad_out && dlogits, MIS_EINVAL, "segloss_bwd: null pointer");
    const bool ms = with_msssim != 0;
    if (int rc = sl_check("segloss_bwd", N, H, W, ms)) return rc;
    const SlDims d = sl_dims(N, H, W);
    const SlWin g = sl_window();
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(workspace);
    float* X = ws;
    float* Y = ws + d.total;
    float* dX = ws + 2 * d.total;
    float* A1 = ws + 3 * d.total;
    float* A2 = A1 + (size_t)N * H * W;
    float* M1 = A2 + (size_t)N * H * W;
    const float* state = ws + 3 * d.total + (size_t)3 * N * H * W;
    for (int l = SL_LEVELS - 1; ms && l >= 0; --l) {
        hipLaunchKernelGGL(sl_level_bwd_maps_kernel, dim3(64, N), dim3(256), 0, s, (const float*)(X + d.off[l]), (const float*)(Y + d.off[l]), d.H[l], d.W[l], g,
                           l, state, A1, A2, M1);
        const bool coarse = l < SL_LEVELS - 1;
        hipLaunchKernelGGL(sl_level_bwd_gather_kernel, dim3(sl_grid((long long)N * d.H[l] * d.W[l])), dim3(256), 0, s, (const float*)(X + d.off[l]),
                           (const float*)(Y + d.off[l]), N, d.H[l], d.W[l], g, (const float*)A1, (const float*)A2, (const float*)M1,
                           coarse ? (const float*)(dX + d.off[l + 1]) : (const float*)nullptr, coarse ? d.H[l + 1] : 0, coarse ? d.W[l + 1] : 0,
                           dX + d.off[l]);
    }
    const long long n0 = (long long)N * H * W;
    hipLaunchKernelGGL(sl_sigmoid_bwd_kernel, dim3(sl_grid(n0)), dim3(256), 0, s, (const float*)X, target, ms ? (const float*)dX : (const float*)nullptr, out, grad_out, n0,
                       dlogits);
    MIS_LAUNCH_CHECK("segloss_bwd");
    return MIS_OK;
}
