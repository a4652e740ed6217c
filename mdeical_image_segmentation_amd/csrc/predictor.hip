// Patch-tiled volume prediction on the device (reference model/unet3d/predictor.py:85-168 `StandardPredictor`, with
// dataset/unet3d_dataset/utils.py:85-125 `SliceBuilder`, :314-342 `mirror_pad`, :345-361 `remove_padding`):
//   the raw volume stays in HBM; every patch is gathered with its halo straight from it (np.pad(mode='reflect') as index arithmetic,
//   the padded copy is never built), the network's output has its halo cut off and is accumulated into the prediction map while a
//   visit counter is incremented, and the map is finally divided by the counter (+ optional channel arg-max -> uint16).
// All three kernels are HBM-bound index passes, one thread per voxel.
#include "common.hpp"

__device__ __forceinline__ int reflect101(int i, int n) {   // numpy 'reflect': ... 2 1 | 0 1 2 ... n-1 | n-2 n-3 ...
    if (n == 1) return 0;
    const int period = 2 * n - 2;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ vol, int C, int D, int H, int W, const int* __restrict__ origins,
                                                           int NP, int PD, int PH, int PW, int hd, int hh, int hw, float* __restrict__ out) {
    const long long per = (long long)C * PD * PH * PW, total = per * NP;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int p = (int)(i / per);
        long long r = i - (long long)p * per;
        const int x = (int)(r % PW);
        r /= PW;
        const int y = (int)(r % PH);
        r /= PH;
        const int z = (int)(r % PD);
        const int c = (int)(r / PD);
        const int sz = reflect101(origins[3 * p + 0] - hd + z, D);
        const int sy = reflect101(origins[3 * p + 1] - hh + y, H);
        const int sx = reflect101(origins[3 * p + 2] - hw + x, W);
        out[i] = vol[(((size_t)c * D + sz) * H + sy) * W + sx];
    }
}

// pred: (C, PD, PH, PW) network output of ONE patch (halo included).  activation 0: none (the reference's model returns logits in
// eval mode too, model/unet3d/model.py:145-149), 1: sigmoid, 2: softmax over channels (upstream pytorch-3dunet behaviour).
__global__ __launch_bounds__(256) void patch_accumulate_kernel(const float* __restrict__ pred, int C, int PD, int PH, int PW, int hd, int hh, int hw,
                                                               int activation, int channel, int oz, int oy, int ox, float* __restrict__ map,
                                                               unsigned char* __restrict__ norm, int D, int H, int W) {
    const int id = PD - 2 * hd, ih = PH - 2 * hh, iw = PW - 2 * hw;     // interior
    const long long nin = (long long)id * ih * iw;
    const long long pstride = (long long)PD * PH * PW, vstride = (long long)D * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nin; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % iw);
        const long long t = i / iw;
        const int y = (int)(t % ih), z = (int)(t / ih);
        const long long src = ((long long)(z + hd) * PH + (y + hh)) * PW + (x + hw);
        const long long dst = ((long long)(oz + z) * H + (oy + y)) * W + (ox + x);
        float mx = -INFINITY, den = 0.f;
        if (activation == 2) {
            for (int c = 0; c < C; ++c) mx = fmaxf(mx, pred[c * pstride + src]);
            for (int c = 0; c < C; ++c) den += expf(pred[c * pstride + src] - mx);
        }
        const int c0 = channel < 0 ? 0 : channel, c1 = channel < 0 ? C : channel + 1;
        for (int c = c0; c < c1; ++c) {
            float v = pred[c * pstride + src];
            if (activation == 1) v = 1.0f / (1.0f + expf(-v));
            if (activation == 2) v = expf(v - mx) / den;
            map[(channel < 0 ? c : 0) * vstride + dst] += v;
        }
        norm[dst] = (unsigned char)(norm[dst] + 1);
    }
}

__global__ __launch_bounds__(256) void pred_finalize_kernel(const float* __restrict__ map, const unsigned char* __restrict__ norm, int C, long long nvox,
                                                            float* __restrict__ prob, unsigned short* __restrict__ seg) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvox; i += (long long)gridDim.x * 256) {
        const float cnt = (float)norm[i];
        float best = 0.f;
        int arg = 0;
        for (int c = 0; c < C; ++c) {
            const float v = map[(size_t)c * nvox + i] / cnt;       // numpy: float32 / uint8 -> float32 (0/0 = nan for unvisited voxels)
            if (prob != nullptr) prob[(size_t)c * nvox + i] = v;
            if (c == 0 || v > best) {                              // np.argmax: first maximum wins
                best = v;
                arg = c;
            }
        }
        if (seg != nullptr) seg[i] = (unsigned short)arg;
    }
}

static unsigned pr_grid(long long total) {
    long long b = (total + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (unsigned)b;
}

extern "C" int mis_patch_gather_reflect(const float* vol, int C, int D, int H, int W, const int* origins, int NP, int PD, int PH, int PW, int hd,
                                        int hh, int hw, float* patches, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(vol && origins && patches, MIS_EINVAL, "patch_gather_reflect: null pointer");
    MIS_REQUIRE(C > 0 && D > 0 && H > 0 && W > 0 && NP > 0 && PD > 0 && PH > 0 && PW > 0 && hd >= 0 && hh >= 0 && hw >= 0, MIS_EINVAL,
                "patch_gather_reflect: sizes");
    MIS_REQUIRE(PD > 2 * hd && PH > 2 * hh && PW > 2 * hw, MIS_EINVAL, "patch_gather_reflect: the halo swallows the patch");
    MIS_REQUIRE(hd < D && hh < H && hw < W, MIS_EINVAL, "patch_gather_reflect: numpy 'reflect' padding needs halo < volume size");
    hipLaunchKernelGGL(patch_gather_kernel, dim3(pr_grid((long long)NP * C * PD * PH * PW)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), vol, C, D,
                       H, W, origins, NP, PD, PH, PW, hd, hh, hw, patches);
    MIS_LAUNCH_CHECK("patch_gather_reflect");
    return MIS_OK;
}

extern "C" int mis_patch_accumulate(const float* pred, int C, int PD, int PH, int PW, int hd, int hh, int hw, int activation, int channel, int oz,
                                    int oy, int ox, float* map, unsigned char* norm, int D, int H, int W, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(pred && map && norm, MIS_EINVAL, "patch_accumulate: null pointer");
    MIS_REQUIRE(C > 0 && PD > 2 * hd && PH > 2 * hh && PW > 2 * hw && hd >= 0 && hh >= 0 && hw >= 0, MIS_EINVAL, "patch_accumulate: sizes");
    MIS_REQUIRE(activation >= 0 && activation <= 2 && channel >= -1 && channel < C, MIS_EINVAL, "patch_accumulate: activation / channel");
    MIS_REQUIRE(oz >= 0 && oy >= 0 && ox >= 0 && oz + PD - 2 * hd <= D && oy + PH - 2 * hh <= H && ox + PW - 2 * hw <= W, MIS_EINVAL,
                "patch_accumulate: the patch interior [%d,%d,%d]+ leaves the volume", oz, oy, ox);
    hipLaunchKernelGGL(patch_accumulate_kernel, dim3(pr_grid((long long)(PD - 2 * hd) * (PH - 2 * hh) * (PW - 2 * hw))), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), pred, C, PD, PH, PW, hd, hh, hw, activation, channel, oz, oy, ox, map, norm, D, H, W);
    MIS_LAUNCH_CHECK("patch_accumulate");
    return MIS_OK;
}

extern "C" int mis_pred_finalize(const float* map, const unsigned char* norm, int C, long long nvox, float* prob, unsigned short* seg, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(map && norm && (prob || seg) && C > 0 && nvox > 0, MIS_EINVAL, "pred_finalize: bad argument");
    hipLaunchKernelGGL(pred_finalize_kernel, dim3(pr_grid(nvox)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), map, norm, C, nvox, prob, seg);
    MIS_LAUNCH_CHECK("pred_finalize");
    return MIS_OK;
}
