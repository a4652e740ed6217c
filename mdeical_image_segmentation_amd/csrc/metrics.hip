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
