// The reference's 2-D sample pipeline on the device (dataset/unet2d_dataset/MYDataset.py:127-157): albumentations 1.4.10
//   Resize(OH, OW, INTER_NEAREST) -> HorizontalFlip -> VerticalFlip -> RandomRotate90 (np.rot90, k quarter turns) -> Transpose ->
//   RandomBrightnessContrast (uint8 look-up table, image only) -> ToTensorV2 (HWC -> CHW) -> float / 255
// for one decoded uint8 sample resident in HBM.  Pure index arithmetic + a 256-entry table: one output element per thread, evaluated backwards
// from the output pixel (inverse of the geometric chain), so the sample is read once and nothing intermediate is written.
// The third-party library is absent here ("parity unpinned"): its published algorithm is restated - cv2 nearest resize sx = min(floor(x * W/OW), W-1),
// rot90 = counter-clockwise quarter turns, table = clip(float32(v) * alpha + beta * 255, 0, 255) truncated to uint8.
#include "common.hpp"

struct Aug2dArgs {
    int H, W, C, OH, OW;        // source size, channels of the image (mask has one), size after the resize
    int hflip, vflip, rot_k, transpose, use_bc;
    float alpha, beta255;
    int FH, FW;                 // final size
};

__device__ __forceinline__ void aug2d_source(const Aug2dArgs& a, int y, int x, int& sy, int& sx) {
    if (a.transpose) {          // A4[y][x] = A3[x][y]
        const int t = y;
        y = x;
        x = t;
    }
    // A3 = rot90(A2, k) with A2 of size (OH, OW)
    int y2, x2;
    switch (a.rot_k & 3) {
        case 1: y2 = x; x2 = a.OW - 1 - y; break;
        case 2: y2 = a.OH - 1 - y; x2 = a.OW - 1 - x; break;
        case 3: y2 = a.OH - 1 - x; x2 = y; break;
        default: y2 = y; x2 = x; break;
    }
    if (a.vflip) y2 = a.OH - 1 - y2;
    if (a.hflip) x2 = a.OW - 1 - x2;
    // cv2.resize(..., INTER_NEAREST): source index = min(floor(dst * (1 / (dsize / ssize))), ssize - 1), in double precision
    const double ify = 1.0 / ((double)a.OH / (double)a.H), ifx = 1.0 / ((double)a.OW / (double)a.W);
    sy = (int)floor((double)y2 * ify);
    sx = (int)floor((double)x2 * ifx);
    if (sy > a.H - 1) sy = a.H - 1;
    if (sx > a.W - 1) sx = a.W - 1;
}

__global__ __launch_bounds__(256) void aug2d_u8_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask, Aug2dArgs a,
                                                       float* __restrict__ out_img, float* __restrict__ out_mask) {
    const long long npix = (long long)a.FH * a.FW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / a.FW), x = (int)(i - (long long)y * a.FW);
        int sy, sx;
        aug2d_source(a, y, x, sy, sx);
        const size_t sp = (size_t)sy * a.W + sx;
        for (int c = 0; c < a.C; ++c) {
            unsigned v = img[sp * a.C + c];
            if (a.use_bc) {
                float l = __fadd_rn(__fmul_rn((float)v, a.alpha), a.beta255);       // numpy: lut *= alpha; lut += beta * 255 (float32, no fma)
                l = fminf(fmaxf(l, 0.f), 255.f);
                v = (unsigned)l;                                                     // astype(uint8): truncation
            }
            out_img[(size_t)c * npix + i] = (float)v / 255.0f;
        }
        if (mask != nullptr) out_mask[i] = (float)mask[sp] / 255.0f;
    }
}

extern "C" int mis_aug2d_u8(const unsigned char* img, const unsigned char* mask, int H, int W, int C, int OH, int OW, int hflip, int vflip, int rot_k,
                            int transpose, int use_bc, float alpha, float beta, float* out_img, float* out_mask, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(img && out_img && (mask == nullptr) == (out_mask == nullptr), MIS_EINVAL, "aug2d_u8: pointers");
    MIS_REQUIRE(H > 0 && W > 0 && C >= 1 && C <= 4 && OH > 0 && OW > 0 && rot_k >= 0 && rot_k <= 3, MIS_EINVAL, "aug2d_u8: sizes");
    Aug2dArgs a;
    a.H = H; a.W = W; a.C = C; a.OH = OH; a.OW = OW;
    a.hflip = hflip != 0; a.vflip = vflip != 0; a.rot_k = rot_k; a.transpose = transpose != 0; a.use_bc = use_bc != 0;
    a.alpha = alpha;
    a.beta255 = (float)((double)beta * 255.0);
    const bool swap = ((rot_k & 1) != 0) != (transpose != 0);
    a.FH = swap ? OW : OH;
    a.FW = swap ? OH : OW;
    long long blocks = ((long long)a.FH * a.FW + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(aug2d_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), img, mask, a, out_img, out_mask);
    MIS_LAUNCH_CHECK("aug2d_u8");
    return MIS_OK;
}
