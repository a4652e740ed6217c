// Fused 1x1 segmentation head + loss + arg-max (+ backward in the same pass) for gfx950.
//
// Reads the 64-channel feature map ONCE (16 B per lane, 8 or 16 lanes per pixel), and emits: fp32 logits in the
// reference's NCHW layout, the arg-max mask, loss partial sums, and - for training - dL/dfeatures (already masked by
// the feature's ReLU), dL/dW and dL/db partials.  Cross entropy / BCE need one pass; BCE+Dice needs the global
// per-class sums first, so it runs a statistics pass and a gradient pass.
//
// Reference: final_conv model/unet2d/unet.py:89,127; criterion unet.py:1184-1188 (nn.CrossEntropyLoss / nn.BCEWithLogitsLoss,
// reduction 'mean'); BCEDiceLoss model/unet3d/losses.py:167-178 with compute_per_channel_dice losses.py:7-33
// (sigmoid-normalised, 2*sum(p*t)/clamp(sum(p^2)+sum(t^2), 1e-6), 1 - mean over channels); arg-max sites
// model/unet3d/predictor.py:167, metrics.py:97 (lowest index wins ties).
#include "common.hpp"

constexpr int HEAD_BLOCKS = 4096;    // (1024 until round 3: 250 dependent load -> shuffle -> store iterations per thread at 160^3 = 1.16 ms for a 1.25 GB pass)
constexpr int HEAD_PSTRIDE = 288;   // floats per block partial: [C*64 dW][C db][1 loss][3*C dice sums], padded

struct HeadArgs {
    long long S;      // pixels per image
    int N;
    const void* y;
    int y_ld;
    const float* w;
    const float* b;
    const void* labels;
    float* logits;
    uint8_t* argmax;
    float* partial;
    void* dy;
    int dy_ld;
    float grad_scale, alpha, beta;
    const float* sums;   // pass 2 of BCE+Dice: [1 + 3*C] global sums (bce, I_c, P_c, T_c)
};

// Hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each; round 4): the pass was bound by the ~50 instructions per class of the libm expansions, executed
// by all 8 / 16 lanes of a pixel - 0.97 ms for 1.25 GB at 2 x 160^3.  Absolute error per loss term < 1e-6 (mean loss bars are 1e-4), gradients relative 1e-6.
// (round 5 measured libm's expf / logf / log1pf and IEEE division for the fp32 parity mode: final_conv.weight against the fp64 oracle did not move in any digit - 2.7e-6 of
// max |g| either way, the error is the forward's, not the transcendentals' - while the pass went from 0.52 + 0.37 to 0.84 + 0.43 ms at 2 x 128^3; ACC is therefore false in
// both precisions and stays as the switch the measurement was made with)
template <bool ACC> __device__ __forceinline__ float h_exp(float x) { return ACC ? expf(x) : __expf(x); }
template <bool ACC> __device__ __forceinline__ float h_log(float x) { return ACC ? logf(x) : __logf(x); }
template <bool ACC> __device__ __forceinline__ float h_log1p_exp_neg_abs(float x) { return ACC ? log1pf(expf(-fabsf(x))) : __logf(1.f + __expf(-fabsf(x))); }
template <bool ACC> __device__ __forceinline__ float h_rcp(float x) { return ACC ? 1.f / x : __frcp_rn(x); }
template <bool ACC> __device__ __forceinline__ float sigmoidf_(float x) { return ACC ? 1.f / (1.f + expf(-x)) : __frcp_rn(1.f + __expf(-x)); }

// LOSS: 0 CE, 1 BCE, 2 BCE+Dice, -1 none.   PASS: 0 = forward outputs (+ backward for LOSS 0/1 when dy != null), 1 = BCE+Dice gradient pass
template <typename T, int C, int LOSS, int PASS>
__global__ __launch_bounds__(256) void head_kernel(const HeadArgs a) {
    constexpr int EPC = Tr<T>::EPC;
    constexpr bool ACC = false;                   // (see h_exp: libm forms measured, no accuracy gain)
    constexpr int LPP = 64 / EPC;     // lanes per pixel (8 bf16 / 16 f32)
    constexpr int PPB = 256 / LPP;    // pixels per block iteration
    __shared__ float red[4][HEAD_PSTRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = tid % LPP;
    const bool bwd = (PASS == 1) || (LOSS == 3) || (LOSS >= 0 && LOSS <= 1 && a.dy != nullptr);

    float wr[C][EPC];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int e = 0; e < EPC; ++e) wr[c][e] = a.w[c * 64 + sub * EPC + e];
    float bb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) bb[c] = a.b[c];

    float dwa[C][EPC];
    float dba[C];
    float lsum = 0.f;
    float dI[C], dP[C], dT[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        dba[c] = 0.f;
        dI[c] = dP[c] = dT[c] = 0.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) dwa[c][e] = 0.f;
    }
    const long long total = (long long)a.N * a.S;
    const float inv_total = 1.f / (float)total;

    // BCE+Dice gradient coefficients (pass 1)
    float kI[C], kD[C];
    if constexpr (PASS == 1) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float I = a.sums[1 + c], den = a.sums[1 + C + c] + a.sums[1 + 2 * C + c];
            if (den > 1e-6f) {
                kI[c] = 1.f / den;
                kD[c] = 2.f * I / (den * den);
            } else {
                kI[c] = 1.f / 1e-6f;   // clamp(min=eps): the denominator is a constant
                kD[c] = 0.f;
            }
        }
    }

    // the pass over one pixel group; `raw` = this lane's 16 bytes of the pixel's features.  The caller loads TWO pixel groups ahead of the arithmetic (round 4: with one
    // dependent load -> shuffle -> store chain per iteration the pass was latency-bound: 0.97 ms for 1.25 GB at 2 x 160^3)
    auto body = [&](const long long p, const u32x4& raw) {
        const long long n = p / a.S, sp = p - n * a.S;
        float f[EPC];
        unpack_chunk<T>(raw, f);
        float lg[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < EPC; ++e) d = fmaf(f[e], wr[c][e], d);
#pragma unroll
            for (int o = 1; o < LPP; o <<= 1) d += __shfl_xor(d, o, 64);
            lg[c] = d + bb[c];
        }
        float dl[C];
#pragma unroll
        for (int c = 0; c < C; ++c) dl[c] = 0.f;

        if constexpr (PASS == 0) {
            if (a.logits != nullptr) {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (sub == c) a.logits[((size_t)n * C + c) * a.S + sp] = lg[c];
            }
            if (a.argmax != nullptr && sub == 0) {
                int am = 0;
                if constexpr (C == 1) {
                    am = lg[0] > 0.f ? 1 : 0;
                } else {
                    float m = lg[0];
#pragma unroll
                    for (int c = 1; c < C; ++c)
                        if (lg[c] > m) {
                            m = lg[c];
                            am = c;
                        }
                }
                a.argmax[p] = (uint8_t)am;
            }
        }
        if constexpr (LOSS == 0) {
            const long long lab = reinterpret_cast<const long long*>(a.labels)[p];
            float m = lg[0];
#pragma unroll
            for (int c = 1; c < C; ++c) m = fmaxf(m, lg[c]);
            float se = 0.f, ex[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                ex[c] = h_exp<ACC>(lg[c] - m);
                se += ex[c];
            }
            const float lse = m + h_log<ACC>(se);
            float xl = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c)
                if (lab == c) xl = lg[c];
            if (sub == 0) lsum += lse - xl;
            const float inv = h_rcp<ACC>(se);
#pragma unroll
            for (int c = 0; c < C; ++c) dl[c] = a.grad_scale * inv_total * (ex[c] * inv - (lab == c ? 1.f : 0.f));
        } else if constexpr (LOSS == 3) {
            const float* ext = reinterpret_cast<const float*>(a.labels);
#pragma unroll
            for (int c = 0; c < C; ++c) dl[c] = a.grad_scale * ext[((size_t)n * C + c) * a.S + sp];
        } else if constexpr (LOSS == 1 || LOSS == 2) {
            const float* tg = reinterpret_cast<const float*>(a.labels);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float t = tg[((size_t)n * C + c) * a.S + sp];
                const float x = lg[c];
                const float sg = sigmoidf_<ACC>(x);
                if constexpr (PASS == 0) {
                    if (sub == 0) {
                        lsum += fmaxf(x, 0.f) - x * t + h_log1p_exp_neg_abs<ACC>(x);
                        if constexpr (LOSS == 2) {
                            dI[c] += sg * t;
                            dP[c] += sg * sg;
                            dT[c] += t * t;
                        }
                    }
                    if constexpr (LOSS == 1) dl[c] = a.grad_scale * inv_total * (1.f / C) * (sg - t);
                } else {
                    const float sp1 = sg * (1.f - sg);
                    const float dbce = a.alpha * inv_total * (1.f / C) * (sg - t);
                    const float ddice = -a.beta * (2.f / C) * sp1 * (t * kI[c] - kD[c] * sg);
                    dl[c] = a.grad_scale * (dbce + ddice);
                }
            }
        }
        if (bwd) {
            float o[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float g = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    g = fmaf(dl[c], wr[c][e], g);
                    dwa[c][e] = fmaf(dl[c], f[e], dwa[c][e]);
                }
                o[e] = f[e] > 0.f ? g : 0.f;
            }
            if (sub == 0) {
#pragma unroll
                for (int c = 0; c < C; ++c) dba[c] += dl[c];
            }
            *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(a.dy) + p * a.dy_ld + sub * EPC) = pack_chunk<T>(o);
        }
    };
    {
        const T* const yb = reinterpret_cast<const T*>(a.y) + sub * EPC;
        const long long stride = (long long)gridDim.x * PPB;
        long long p = (long long)blockIdx.x * PPB + tid / LPP;
        for (; p + stride < total; p += 2 * stride) {
            const u32x4 r0 = *reinterpret_cast<const u32x4*>(yb + p * a.y_ld);
            const u32x4 r1 = *reinterpret_cast<const u32x4*>(yb + (p + stride) * a.y_ld);
            body(p, r0);
            body(p + stride, r1);
        }
        if (p < total) body(p, *reinterpret_cast<const u32x4*>(yb + p * a.y_ld));
    }

    // ---- block reduction: lanes with the same `sub` hold the same channel slice ----
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float v = dwa[c][e];
#pragma unroll
            for (int o = LPP; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            dwa[c][e] = v;
        }
        dba[c] = wave_sum(dba[c]);
        dI[c] = wave_sum(dI[c]);
        dP[c] = wave_sum(dP[c]);
        dT[c] = wave_sum(dT[c]);
    }
    lsum = wave_sum(lsum);
    if (lane < LPP) {
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int e = 0; e < EPC; ++e) red[wave][c * 64 + lane * EPC + e] = dwa[c][e];
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            red[wave][C * 64 + c] = dba[c];
            red[wave][C * 64 + C + 1 + c] = dI[c];
            red[wave][C * 64 + C + 1 + C + c] = dP[c];
            red[wave][C * 64 + C + 1 + 2 * C + c] = dT[c];
        }
        red[wave][C * 64 + C] = lsum;
    }
    __syncthreads();
    constexpr int NP = C * 64 + C + 1 + 3 * C;
    float* out = a.partial + (size_t)blockIdx.x * HEAD_PSTRIDE;
    for (int i = tid; i < NP; i += 256) out[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

// Final reduction over block partials (fixed order), two tiny kernels: (1) one wave per partial column -> double sums
// in the workspace tail; (2) finalisation: grads, loss (mode 0: PASS-0 results, mode 1: BCE+Dice gradient pass).
__global__ __launch_bounds__(256) void head_colreduce_kernel(const float* __restrict__ partial, int nblocks, int NP, double* __restrict__ sums) {
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (col >= NP) return;
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += (double)partial[(size_t)b * HEAD_PSTRIDE + col];
    s = wave_sum_d(s);
    if (lane == 0) sums[col] = s;
}

__global__ void head_finalize_kernel(const double* __restrict__ acc, int C, int loss, int mode, long long total, float alpha, float beta,
                                     float* __restrict__ dw, float* __restrict__ db, float* __restrict__ loss_out) {
    const bool write_grads = (mode == 1) || (loss == 0 || loss == 1 || loss == 3);
    if (write_grads && dw != nullptr) {
        for (int i = threadIdx.x; i < C * 64; i += blockDim.x) dw[i] = (float)acc[i];
        for (int i = threadIdx.x; i < C; i += blockDim.x) db[i] = (float)acc[C * 64 + i];
    }
    if (mode == 0 && threadIdx.x == 0 && loss_out != nullptr && loss >= 0 && loss != 3) {
        const double ls = acc[C * 64 + C];
        if (loss == 0) {
            loss_out[0] = (float)(ls / (double)total);
        } else if (loss == 1) {
            loss_out[0] = (float)(ls / ((double)total * C));
        } else {
            const double bce = ls / ((double)total * C);
            double dice = 0.0;
            for (int c = 0; c < C; ++c) {
                const double I = acc[C * 64 + C + 1 + c];
                double den = acc[C * 64 + C + 1 + C + c] + acc[C * 64 + C + 1 + 2 * C + c];
                if (den < 1e-6) den = 1e-6;
                dice += 2.0 * I / den;
            }
            dice /= C;
            loss_out[0] = (float)(alpha * bce + beta * (1.0 - dice));
            loss_out[1] = (float)bce;
            for (int c = 0; c < 3 * C; ++c) loss_out[2 + c] = (float)acc[C * 64 + C + 1 + c];
        }
    }
}

static int head_reduce(const MisHeadDesc* d, int blocks, int C, int mode, long long total, hipStream_t s) {
    const int NP = C * 64 + C + 1 + 3 * C;
    double* sums = reinterpret_cast<double*>(d->workspace + (size_t)HEAD_BLOCKS * HEAD_PSTRIDE);
    hipLaunchKernelGGL(head_colreduce_kernel, dim3((NP + 3) / 4), dim3(256), 0, s, (const float*)d->workspace, blocks, NP, sums);
    MIS_LAUNCH_CHECK("head_colreduce");
    hipLaunchKernelGGL(head_finalize_kernel, dim3(1), dim3(256), 0, s, (const double*)sums, C, d->loss, mode, total, d->alpha, d->beta, d->dw,
                       d->db, d->loss_out);
    MIS_LAUNCH_CHECK("head_finalize");
    return MIS_OK;
}

// conv_ppd_head.hip (the head fused into the last convolution's epilogue) writes block partials in head_kernel's row layout and finishes through the same two kernels
int head_reduce_partials(const MisHeadDesc* d, int blocks, hipStream_t s) {
    return head_reduce(d, blocks, d->C, 0, (long long)d->N * d->npix_per_image, s);
}

extern "C" size_t mis_head_workspace_bytes(const MisHeadDesc* d) {
    (void)d;
    return (size_t)HEAD_BLOCKS * HEAD_PSTRIDE * sizeof(float) + HEAD_PSTRIDE * sizeof(double);
}

template <typename T, int C, int LOSS, int PASS> static void head_launch(const HeadArgs& a, unsigned blocks, hipStream_t s) {
    hipLaunchKernelGGL((head_kernel<T, C, LOSS, PASS>), dim3(blocks), dim3(256), 0, s, a);
}

template <typename T, int C> static int head_dispatch(const MisHeadDesc* d, HeadArgs& a, unsigned blocks, hipStream_t s) {
    const long long total = (long long)d->N * d->npix_per_image;
    const bool train = d->dy != nullptr;
    if (d->phase == 2) {          // gradient part of BCE + Dice only: the sums in loss_out are the caller's (possibly all-reduced over ranks)
        a.sums = d->loss_out + 1;
        head_launch<T, C, 2, 1>(a, blocks, s);
        MIS_LAUNCH_CHECK("head_pass1");
        return head_reduce(d, (int)blocks, C, 1, total, s);
    }
    switch (d->loss) {
        case -1: head_launch<T, C, -1, 0>(a, blocks, s); break;
        case 0:
            if constexpr (C >= 2) head_launch<T, C, 0, 0>(a, blocks, s);
            else MIS_REQUIRE(false, MIS_EUNSUPPORTED, "head: cross entropy needs C >= 2");
            break;
        case 1: head_launch<T, C, 1, 0>(a, blocks, s); break;
        case 2: head_launch<T, C, 2, 0>(a, blocks, s); break;
        case 3: head_launch<T, C, 3, 0>(a, blocks, s); break;
        default: MIS_REQUIRE(false, MIS_EINVAL, "head: loss %d", d->loss);
    }
    MIS_LAUNCH_CHECK("head");
    {
        const int rc = head_reduce(d, (int)blocks, C, 0, total, s);
        if (rc != MIS_OK) return rc;
    }
    if (d->loss == 2 && train && d->phase != 1) {
        a.sums = d->loss_out + 1;   // [bce, I_c.., P_c.., T_c..]
        head_launch<T, C, 2, 1>(a, blocks, s);
        MIS_LAUNCH_CHECK("head_pass1");
        const int rc = head_reduce(d, (int)blocks, C, 1, total, s);
        if (rc != MIS_OK) return rc;
    }
    return MIS_OK;
}

template <typename T> static int head_by_c(const MisHeadDesc* d, HeadArgs& a, unsigned blocks, hipStream_t s) {
    switch (d->C) {
        case 1: return head_dispatch<T, 1>(d, a, blocks, s);
        case 2: return head_dispatch<T, 2>(d, a, blocks, s);
        case 3: return head_dispatch<T, 3>(d, a, blocks, s);
        case 4: return head_dispatch<T, 4>(d, a, blocks, s);
    }
    MIS_REQUIRE(false, MIS_EUNSUPPORTED, "head: C must be 1..4 (got %d)", d->C);
}

extern "C" int mis_head_loss(const MisHeadDesc* d, void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    MIS_REQUIRE(d != nullptr, MIS_EINVAL, "head: null descriptor");
    MIS_REQUIRE(d->dtype == MIS_F32 || d->dtype == MIS_BF16, MIS_EINVAL, "head: dtype");
    MIS_REQUIRE(d->Cfeat == 64, MIS_EUNSUPPORTED, "head: Cfeat must be 64 (got %d)", d->Cfeat);
    MIS_REQUIRE(d->N > 0 && d->npix_per_image > 0, MIS_EINVAL, "head: empty input");
    MIS_REQUIRE(d->y && d->w && d->b && d->workspace, MIS_EINVAL, "head: null pointer");
    MIS_REQUIRE(d->loss < 0 || (d->labels != nullptr && (d->loss_out != nullptr || d->loss == 3)), MIS_EINVAL, "head: labels / loss_out missing");
    MIS_REQUIRE(d->loss != 3 || d->dy != nullptr, MIS_EINVAL, "head: loss 3 (external dL/dlogits) is a backward-only mode");
    MIS_REQUIRE(d->dy == nullptr || (d->dw != nullptr && d->db != nullptr && d->loss >= 0), MIS_EINVAL, "head: backward needs dw, db and a loss");
    MIS_REQUIRE(d->workspace_bytes >= mis_head_workspace_bytes(d), MIS_EINVAL, "head: workspace too small");
    MIS_REQUIRE(d->phase == 0 || ((d->phase == 1 || d->phase == 2) && d->loss == 2 && d->dy != nullptr), MIS_EINVAL, "head: phase %d needs loss 2 with backward", d->phase);
    const int EPC = d->dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(d->y_ld % EPC == 0 && (d->dy == nullptr || d->dy_ld % EPC == 0), MIS_EINVAL, "head: ld alignment");
    HeadArgs a;
    a.S = d->npix_per_image; a.N = d->N; a.y = d->y; a.y_ld = d->y_ld; a.w = d->w; a.b = d->b; a.labels = d->labels;
    a.logits = d->logits; a.argmax = d->argmax; a.partial = d->workspace; a.dy = d->dy; a.dy_ld = d->dy_ld;
    a.grad_scale = d->grad_scale; a.alpha = d->alpha; a.beta = d->beta; a.sums = nullptr;
    const long long total = (long long)d->N * d->npix_per_image;
    const int ppb = 256 / (64 / EPC);
    long long blocks = (total + ppb - 1) / ppb;
    if (blocks > HEAD_BLOCKS) blocks = HEAD_BLOCKS;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == MIS_BF16) return head_by_c<__bf16>(d, a, (unsigned)blocks, s);
    return head_by_c<float>(d, a, (unsigned)blocks, s);
}
