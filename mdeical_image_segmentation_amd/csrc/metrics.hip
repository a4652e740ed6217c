// Evaluation metrics of the 2-D trainer on the device (reference trainer/metrcis.py:61-109,153-168, `compute_metrics`):
//   p = 1 / (1 + exp(-logit) + 1e-6);  threshold = mean(p) over the whole evaluation set;
//   per sample: pred = p > threshold, lab = label > threshold;
//   IoU  = |pred & lab| / max(|pred| + |lab| - |pred & lab|, 1e-6);   Dice = (2|pred & lab| + 1e-6) / (|pred| + 1e-6 + |lab| + 1e-6)
//   result = mean over samples.  HBM-bound: two passes over the logits (the threshold is a global statistic), fixed-order
//   reductions (bitwise reproducible), counts are exact integers.
#include "common.hpp"

constexpr int MT_BLOCKS = 512;

__device__ __forceinline__ float metric_prob(float x) { return 1.0f / (1.0f + expf(-x) + 1e-6f); }

__global__ __launch_bounds__(256) void metrics_sum_kernel(const float* __restrict__ logits, long long total, double* __restrict__ part) {
    __shared__ double red[4];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) s += (double)metric_prob(logits[i]);
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void metrics_threshold_kernel(const double* __restrict__ part, int nparts, long long total, float* __restrict__ out, int auto_thr, float thr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nparts; ++i) s += part[i];
        out[2] = auto_thr ? (float)(s / (double)total) : thr;
    }
}

// grid (blocks_per_sample, N): counts[n][b][3] = {intersection, pred, label}
__global__ __launch_bounds__(256) void metrics_count_kernel(const float* __restrict__ logits, const float* __restrict__ labels, long long npix,
                                                            const float* __restrict__ out, unsigned int* __restrict__ counts, int is_logits) {
    __shared__ unsigned int red[3][4];
    const float thr = out[2];
    const int n = blockIdx.y;
    const float* lg = logits + (size_t)n * npix;
    const float* lb = labels + (size_t)n * npix;
    unsigned int ci = 0, cp = 0, cl = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const bool p = (is_logits ? metric_prob(lg[i]) : lg[i]) > thr, l = lb[i] > thr;
        ci += (p && l) ? 1u : 0u;
        cp += p ? 1u : 0u;
        cl += l ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ci += __shfl_xor(ci, o, 64);
        cp += __shfl_xor(cp, o, 64);
        cl += __shfl_xor(cl, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = ci;
        red[1][threadIdx.x >> 6] = cp;
        red[2][threadIdx.x >> 6] = cl;
    }
    __syncthreads();
    if (threadIdx.x < 3) counts[((size_t)n * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

__global__ void metrics_finalize_kernel(const unsigned int* __restrict__ counts, int N, int bps, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // numpy evaluates these in float32 (the arrays are float32 masks); the counts are exact in float32 up to 2^24 pixels per sample
    float iou_sum = 0.f, dice_sum = 0.f;
    for (int n = 0; n < N; ++n) {
        unsigned long long ci = 0, cp = 0, cl = 0;
        for (int b = 0; b < bps; ++b) {
            ci += counts[((size_t)n * bps + b) * 3 + 0];
            cp += counts[((size_t)n * bps + b) * 3 + 1];
            cl += counts[((size_t)n * bps + b) * 3 + 2];
        }
        const float inter = (float)ci, sp = (float)cp, sl = (float)cl;
        const float uni = fmaxf(sp + sl - inter, 1e-6f);
        iou_sum += inter / uni;
        dice_sum += (2.0f * inter + 1e-6f) / ((sp + 1e-6f) + (sl + 1e-6f));
    }
    out[0] = iou_sum / (float)N;
    out[1] = dice_sum / (float)N;
}

static int metrics_bps(long long npix) {
    long long b = (npix + 256 * 16 - 1) / (256 * 16);
    if (b < 1) b = 1;
    if (b > 256) b = 256;
    return (int)b;
}

extern "C" size_t mis_seg_metrics_workspace_bytes(int N, long long npix) {
    return (size_t)MT_BLOCKS * sizeof(double) + (size_t)N * metrics_bps(npix) * 3 * sizeof(unsigned int);
}

// values, labels: fp32 (N, npix) contiguous (the trainer's (N,1,H,W) arrays); out[3] = {iou, dice, threshold used}.
// values_are_logits: apply the reference's sigmoid first (compute_metrics) or take the values as they are (compute_iou / compute_dice);
// auto_threshold: threshold = mean of the (sigmoid) values over everything, else `threshold`.
extern "C" int mis_seg_metrics(const float* logits, const float* labels, int N, long long npix, int values_are_logits, int auto_threshold,
                               float threshold, void* workspace, float* out, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(logits && labels && workspace && out, MIS_EINVAL, "seg_metrics: null pointer");
    MIS_REQUIRE(N > 0 && npix > 0 && npix < (1ll << 24), MIS_EINVAL, "seg_metrics: N %d, pixels per sample %lld (must be < 2^24)", N, npix);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    double* part = reinterpret_cast<double*>(workspace);
    unsigned int* counts = reinterpret_cast<unsigned int*>(part + MT_BLOCKS);
    const long long total = (long long)N * npix;
    int nb = (int)((total + 255) / 256 < MT_BLOCKS ? (total + 255) / 256 : MT_BLOCKS);
    MIS_REQUIRE(values_are_logits || !auto_threshold, MIS_EUNSUPPORTED, "seg_metrics: the automatic threshold is defined on logits (compute_metrics)");
    if (auto_threshold) hipLaunchKernelGGL(metrics_sum_kernel, dim3(nb), dim3(256), 0, s, logits, total, part);
    hipLaunchKernelGGL(metrics_threshold_kernel, dim3(1), dim3(64), 0, s, (const double*)part, nb, total, out, auto_threshold, threshold);
    const int bps = metrics_bps(npix);
    hipLaunchKernelGGL(metrics_count_kernel, dim3(bps, N), dim3(256), 0, s, logits, labels, npix, (const float*)out, counts, values_are_logits);
    hipLaunchKernelGGL(metrics_finalize_kernel, dim3(1), dim3(64), 0, s, (const unsigned int*)counts, N, bps, out);
    MIS_LAUNCH_CHECK("seg_metrics");
    return MIS_OK;
}

// ---- MeanIoU of the 3-D validation loop (model/unet3d/metrics.py:33-103; expand_as_one_hot model/unet3d/utils.py:222-254) -------------------------
// Per sample n and channel c:  I = sum(pred & tgt), U = sum(pred | tgt) over the S voxels, with
//   pred = one-hot of the FIRST maximum over channels (C == 1: probability > 0.5)                     [_binarize_predictions :87-98]
//   tgt  = the one-hot float target cast to uint8, or (label == c) for an integer label volume         [expand_as_one_hot]
//   ignore_index: voxels whose target equals it are zeroed in pred and tgt (a label volume carries it into every channel) [:63-68]
// Integer counts accumulated with 64-bit atomic adds (order-independent, exact); the ratio / means are the caller's few scalars.
constexpr int IOU_MAXC = 16;

template <bool LABELS>
__global__ __launch_bounds__(256) void iou3d_counts_kernel(const float* __restrict__ probs, const void* __restrict__ target, int N, int C, long long S,
                                                           int has_ignore, long long ignore_index, unsigned long long* __restrict__ counts) {
    const int n = blockIdx.y;
    unsigned long long inter[IOU_MAXC], uni[IOU_MAXC];
#pragma unroll
    for (int c = 0; c < IOU_MAXC; ++c) inter[c] = uni[c] = 0ull;
    const float* pb = probs + (size_t)n * C * S;
    for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < S; s += (long long)gridDim.x * 256) {
        int best = 0;
        float bv = pb[s];
        for (int c = 1; c < C; ++c) {
            const float v = pb[(size_t)c * S + s];
            if (v > bv) {
                bv = v;
                best = c;
            }
        }
        long long lbl = 0;
        bool ign_all = false;
        if (LABELS) {
            lbl = reinterpret_cast<const long long*>(target)[(size_t)n * S + s];
            ign_all = has_ignore && lbl == ignore_index;
        }
#pragma unroll
        for (int c = 0; c < IOU_MAXC; ++c) {
            if (c >= C) continue;
            unsigned p = (C == 1) ? (bv > 0.5f ? 1u : 0u) : (c == best ? 1u : 0u);
            unsigned t;
            if (LABELS) {
                t = (!ign_all && lbl == c) ? 1u : 0u;
                if (ign_all) p = 0u;
            } else {
                const float tv = reinterpret_cast<const float*>(target)[((size_t)n * C + c) * S + s];
                if (has_ignore && tv == (float)ignore_index) {
                    p = 0u;
                    t = 0u;
                } else {
                    t = (unsigned)(unsigned char)(long long)tv;      // .byte() of a float tensor
                }
            }
            inter[c] += p & t;
            uni[c] += p | t;
        }
    }
    __shared__ unsigned long long red[2 * IOU_MAXC];
    if (threadIdx.x < 2 * IOU_MAXC) red[threadIdx.x] = 0ull;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < IOU_MAXC; ++c) {
        if (c >= C) continue;             // uniform across the block
        unsigned long long a = inter[c], b = uni[c];
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off, 64);
            b += __shfl_down(b, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&red[2 * c], a);
            atomicAdd(&red[2 * c + 1], b);
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * C) atomicAdd(&counts[(size_t)n * 2 * C + threadIdx.x], red[threadIdx.x]);
}

extern "C" int mis_iou3d_counts(const float* probs, const void* target, int target_is_labels, int N, int C, long long S, int has_ignore,
                                long long ignore_index, unsigned long long* counts, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(probs && target && counts, MIS_EINVAL, "iou3d_counts: null pointer");
    MIS_REQUIRE(N > 0 && S > 0 && C >= 1 && C <= IOU_MAXC, MIS_EUNSUPPORTED, "iou3d_counts: N %d, C %d (1..%d), S %lld", N, C, IOU_MAXC, S);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(counts, 0, (size_t)N * C * 2 * sizeof(unsigned long long), st) != hipSuccess) return MIS_EHIP;
    long long bx = (S + 256 * 8 - 1) / (256 * 8);
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    if (target_is_labels)
        hipLaunchKernelGGL(iou3d_counts_kernel<true>, dim3((unsigned)bx, N), dim3(256), 0, st, probs, target, N, C, S, has_ignore, ignore_index, counts);
    else
        hipLaunchKernelGGL(iou3d_counts_kernel<false>, dim3((unsigned)bx, N), dim3(256), 0, st, probs, target, N, C, S, has_ignore, ignore_index, counts);
    MIS_LAUNCH_CHECK("iou3d_counts");
    return MIS_OK;
}
