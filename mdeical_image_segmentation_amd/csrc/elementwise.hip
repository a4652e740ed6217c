// HBM-bound kernels of the U-Net hot path (gfx950): first-layer direct conv (fwd + wgrad), per-channel sums
// (bias gradients, GroupNorm statistics), 2x max-pool fwd / fused bwd, weight repack, layout converts.
// All global accesses are 16 bytes per lane along the channel axis (NHWC), grids are capped and grid-strided.
#include <stdlib.h>

#include "common.hpp"
#include "dispatch_cfg.hpp"
#include "relu_bits.hpp"

// =========================================================================================================
// First layer: x fp32 NCHW (Cin <= 4) -> y NHWC [pixel][64], 3x3 pad 1, bias, ReLU.   reference: layers.py:122-123
// 8 lanes per pixel, 8 output channels per lane; weights as [tap*Cin+ci][64] floats in LDS.
// =========================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void first_conv_fwd_kernel(const float* __restrict__ x, int N, int Cin, int H, int W,
                                                             const float* __restrict__ w, const float* __restrict__ bias, T* y,
                                                             int y_ld) {
    __shared__ float wl[36 * 64];
    __shared__ float bl[64];
    const int tid = threadIdx.x;
    for (int i = tid; i < 9 * Cin * 64; i += 256) {
        const int co = i & 63, r = i >> 6;   // r = tap*Cin + ci
        const int tap = r / Cin, ci = r - tap * Cin;
        wl[i] = w[(co * Cin + ci) * 9 + tap];
    }
    if (tid < 64) bl[tid] = bias ? bias[tid] : 0.f;
    __syncthreads();
    const int cg = tid & 7;
    const long long npix = (long long)N * H * W;
    const long long HW = (long long)H * W;
    for (long long p = (long long)blockIdx.x * 32 + (tid >> 3); p < npix; p += (long long)gridDim.x * 32) {
        const int n = (int)(p / HW);
        const int rem = (int)(p - (long long)n * HW);
        const int yy = rem / W, xx = rem - yy * W;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bl[cg * 8 + j];
        for (int ci = 0; ci < Cin; ++ci) {
            const float* xp = x + ((long long)n * Cin + ci) * HW;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
                float xv = 0.f;
                if (sy >= 0 && sy < H && sx >= 0 && sx < W) xv = xp[(long long)sy * W + sx];
                const float* wr = &wl[(tap * Cin + ci) * 64 + cg * 8];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv, wr[j], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
        T* dst = y + p * y_ld + cg * 8;
        if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<u32x4*>(dst) = pack_chunk<__bf16>(acc);
        } else {
            *reinterpret_cast<u32x4*>(dst) = pack_chunk<float>(acc);
            *reinterpret_cast<u32x4*>(dst + 4) = pack_chunk<float>(acc + 4);
        }
    }
}

// Tiled variant: a persistent block walks 2 x 64 pixel tiles whose input halo (Cin x 4 x 66 floats, zero outside the image) is staged in
// LDS; a thread owns 4 consecutive pixels x 8 channels, so one halo row of 6 values feeds 3 taps x 4 pixels and every weight vector read
// from LDS is used 4 times.  No per-tap global loads or bounds checks; the kernel is bound by its 128-byte-per-pixel output stream.
template <typename T>
__global__ __launch_bounds__(256) void first_conv_fwd_tiled_kernel(const float* __restrict__ x, int N, int Cin, int H, int W,
                                                                   const float* __restrict__ w, const float* __restrict__ bias, T* y, int y_ld,
                                                                   unsigned char* __restrict__ rbits) {
    constexpr int TH = 2, TW = 64, HH = TH + 2, HW = TW + 2;
    __shared__ __attribute__((aligned(16))) float wl[36 * 64];
    __shared__ __attribute__((aligned(16))) float bl[64];
    __shared__ __attribute__((aligned(16))) float xs[4][HH][HW + 2];
    const int tid = threadIdx.x;
    for (int i = tid; i < 9 * Cin * 64; i += 256) {
        const int co = i & 63, r = i >> 6;   // r = tap*Cin + ci
        const int tap = r / Cin, ci = r - tap * Cin;
        wl[i] = w[(co * Cin + ci) * 9 + tap];
    }
    if (tid < 64) bl[tid] = bias ? bias[tid] : 0.f;
    const int cg = tid & 7, gq = tid >> 3;
    const int r = gq >> 4, wq = (gq & 15) * 4;
    const int tilesW = (W + TW - 1) / TW, tilesH = (H + TH - 1) / TH;
    const long long ntiles = (long long)N * tilesH * tilesW;
    const long long HWp = (long long)H * W;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        long long b = tile;
        const int tw = (int)(b % tilesW);
        b /= tilesW;
        const int th = (int)(b % tilesH);
        const int n = (int)(b / tilesH);
        const int h0 = th * TH, w0 = tw * TW;
        __syncthreads();                      // previous tile's readers are done (first time: weights are visible after the next barrier)
        for (int i = tid; i < Cin * HH * HW; i += 256) {
            const int px = i % HW, t2 = i / HW;
            const int py = t2 % HH, ci = t2 / HH;
            const int sy = h0 + py - 1, sx = w0 + px - 1;
            float v = 0.f;
            if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = x[((long long)n * Cin + ci) * HWp + (long long)sy * W + sx];
            xs[ci][py][px] = v;
        }
        __syncthreads();
        float acc[4][8];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[v][j] = bl[cg * 8 + j];
        for (int ci = 0; ci < Cin; ++ci) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                float xr[6];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(&xs[ci][r + kh][wq]);
                xr[0] = lo[0]; xr[1] = lo[1]; xr[2] = lo[2]; xr[3] = lo[3];
                xr[4] = xs[ci][r + kh][wq + 4];
                xr[5] = xs[ci][r + kh][wq + 5];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float* wr = &wl[((kh * 3 + kw) * Cin + ci) * 64 + cg * 8];
                    const f32x4 w0v = *reinterpret_cast<const f32x4*>(wr), w1v = *reinterpret_cast<const f32x4*>(wr + 4);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float xv = xr[v + kw];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[v][j] = fmaf(xv, w0v[j], acc[v][j]);
                            acc[v][4 + j] = fmaf(xv, w1v[j], acc[v][4 + j]);
                        }
                    }
                }
            }
        }
        const int yy = h0 + r;
        if (yy < H) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int xx = w0 + wq + v;
                if (xx >= W) break;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[v][j] = fmaxf(acc[v][j], 0.f);
                T* dst = y + (((size_t)n * H + yy) * W + xx) * y_ld + cg * 8;
                if constexpr (sizeof(T) == 2) {
                    const u32x4 pk = pack_chunk<__bf16>(acc[v]);
                    *reinterpret_cast<u32x4*>(dst) = pk;
                    if (rbits != nullptr) rbits[rb_byte_offset((H + 7) >> 3, W, 1, n, yy, xx, cg)] = rb_byte_of(pk);      // ReLU bits of the STORED values (relu_bits.hpp)
                } else {
                    *reinterpret_cast<u32x4*>(dst) = pack_chunk<float>(acc[v]);
                    *reinterpret_cast<u32x4*>(dst + 4) = pack_chunk<float>(acc[v] + 4);
                }
            }
        }
    }
}

// ReLU bits of a bf16 NHWC tensor (relu_bits.hpp): the stand-alone producer - what mis_conv_igemm runs behind a kernel that cannot write them from its epilogue,
// and the reference the tests compare the fused producers with.  One thread per (pixel, 8 channels).
__global__ __launch_bounds__(256) void relu_bits_kernel(const __bf16* __restrict__ y, int y_ld, int N, int H, int W, int C, unsigned char* __restrict__ bits) {
    const int c8n = C >> 3;
    const long long total = (long long)N * H * W * c8n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c8 = (int)(i % c8n);
        const long long pix = i / c8n;
        const int x = (int)(pix % W);
        const long long r = pix / W;
        const int yy = (int)(r % H), n = (int)(r / H);
        const u32x4 v = *reinterpret_cast<const u32x4*>(y + pix * y_ld + c8 * 8);
        bits[rb_byte_offset((H + 7) >> 3, W, C >> 6, n, yy, x, c8)] = rb_byte_of(v);
    }
}

extern "C" size_t mis_relu_bits_bytes(int N, int H, int W, int C) { return rb_bytes(N, H, W, C); }

extern "C" int mis_relu_bits(const void* y, int y_ld, int N, int H, int W, int C, void* bits, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(y && bits && N > 0 && H > 0 && W > 0, MIS_EINVAL, "relu_bits: bad argument");
    MIS_REQUIRE(C > 0 && C % 64 == 0 && y_ld % 8 == 0, MIS_EUNSUPPORTED, "relu_bits: C must be a multiple of 64 and y_ld of 8 (got %d, %d)", C, y_ld);
    long long blocks = ((long long)N * H * W * (C >> 3) + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(relu_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const __bf16*)y, y_ld, N, H, W, C,
                       (unsigned char*)bits);
    MIS_LAUNCH_CHECK("relu_bits");
    return MIS_OK;
}

static int first_fwd_impl(int dtype, const float* x, int N, int Cin, int H, int W, const float* w, const float* bias, void* y, int y_ld, int Cout, void* relu_bits,
                          void* stream);
extern "C" int mis_conv3x3_first_fwd(int dtype, const float* x, int N, int Cin, int H, int W, const float* w, const float* bias, void* y,
                                     int y_ld, int Cout, void* stream) {
    return first_fwd_impl(dtype, x, N, Cin, H, W, w, bias, y, y_ld, Cout, nullptr, stream);
}
// ... and the same with the ReLU bits of the output (relu_bits.hpp; bf16 only) written from the epilogue
extern "C" int mis_conv3x3_first_fwd_rb(int dtype, const float* x, int N, int Cin, int H, int W, const float* w, const float* bias, void* y,
                                        int y_ld, int Cout, void* relu_bits, void* stream) {
    MIS_REQUIRE(relu_bits == nullptr || dtype == MIS_BF16, MIS_EUNSUPPORTED, "first_fwd: ReLU bits are a bf16 feature");
    return first_fwd_impl(dtype, x, N, Cin, H, W, w, bias, y, y_ld, Cout, relu_bits, stream);
}
static int first_fwd_impl(int dtype, const float* x, int N, int Cin, int H, int W, const float* w, const float* bias, void* y, int y_ld, int Cout, void* relu_bits,
                          void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    MIS_REQUIRE(Cout == 64, MIS_EUNSUPPORTED, "first_fwd: Cout must be 64 (got %d)", Cout);
    MIS_REQUIRE(Cin >= 1 && Cin <= 4, MIS_EUNSUPPORTED, "first_fwd: Cin must be 1..4 (got %d)", Cin);
    MIS_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0, MIS_EINVAL, "first_fwd: bad argument");
    MIS_REQUIRE(y_ld % 8 == 0, MIS_EINVAL, "first_fwd: y_ld alignment");
    const long long npix = (long long)N * H * W;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int tiled = !mis_sw(SW_FIRST2D_UNTILED);
    if (tiled) {
        long long tiles = (long long)N * ((H + 1) / 2) * ((W + 63) / 64);
        if (tiles > 4096) tiles = 4096;
        if (dtype == MIS_BF16)
            hipLaunchKernelGGL(first_conv_fwd_tiled_kernel<__bf16>, dim3((unsigned)tiles), dim3(256), 0, s, x, N, Cin, H, W, w, bias, (__bf16*)y, y_ld,
                               (unsigned char*)relu_bits);
        else
            hipLaunchKernelGGL(first_conv_fwd_tiled_kernel<float>, dim3((unsigned)tiles), dim3(256), 0, s, x, N, Cin, H, W, w, bias, (float*)y, y_ld, nullptr);
        MIS_LAUNCH_CHECK("first_conv_fwd(tiled)");
        return MIS_OK;
    }
    long long blocks = (npix + 31) / 32;
    if (blocks > 16384) blocks = 16384;
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(first_conv_fwd_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, x, N, Cin, H, W, w, bias, (__bf16*)y, y_ld);
    else
        hipLaunchKernelGGL(first_conv_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, x, N, Cin, H, W, w, bias, (float*)y, y_ld);
    MIS_LAUNCH_CHECK("first_conv_fwd");
    if (relu_bits != nullptr) return mis_relu_bits(y, y_ld, N, H, W, 64, relu_bits, stream);      // the untiled (A/B) kernel has no fused producer
    return MIS_OK;
}

// dW[co][ci][tap] = sum_p x[p+tap][ci] * dy[p][co];  db[co] = sum_p dy[p][co].  blockIdx.y = ci.
constexpr int FW_BLOCKS = 1024;
template <typename T>
__global__ __launch_bounds__(256) void first_conv_wgrad_kernel(const float* __restrict__ x, int N, int Cin, int H, int W, const T* dy,
                                                               int dy_ld, float* __restrict__ partial /*[blocks][Cin][10][64]*/) {
    __shared__ float red[4][10 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = tid & 7;
    const int ci = blockIdx.y;
    const long long npix = (long long)N * H * W;
    const long long HW = (long long)H * W;
    float acc[10][8];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    for (long long p = (long long)blockIdx.x * 32 + (tid >> 3); p < npix; p += (long long)gridDim.x * 32) {
        const int n = (int)(p / HW);
        const int rem = (int)(p - (long long)n * HW);
        const int yy = rem / W, xx = rem - yy * W;
        float g[8];
        const T* src = dy + p * dy_ld + cg * 8;
        if constexpr (sizeof(T) == 2) {
            unpack_chunk<__bf16>(*reinterpret_cast<const u32x4*>(src), g);
        } else {
            unpack_chunk<float>(*reinterpret_cast<const u32x4*>(src), g);
            unpack_chunk<float>(*reinterpret_cast<const u32x4*>(src + 4), g + 4);
        }
        const float* xp = x + ((long long)n * Cin + ci) * HW;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
            float xv = 0.f;
            if (sy >= 0 && sy < H && sx >= 0 && sx < W) xv = xp[(long long)sy * W + sx];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[tap][j] = fmaf(xv, g[j], acc[tap][j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[9][j] += g[j];
    }
    // lanes with equal (lane & 7) hold the same channels: reduce over lane bits 3..5
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[t][j];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[t][j] = v;
        }
    if (lane < 8) {
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wave][t * 64 + lane * 8 + j] = acc[t][j];
    }
    __syncthreads();
    float* out = partial + ((size_t)blockIdx.x * Cin + ci) * 640;
    for (int i = tid; i < 640; i += 256) out[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

// Tiled variant: 8 x 64 pixel tiles (round 3; 2 x 64 before: two block barriers and one exposed round of loads per 128 pixels made the kernel latency-bound at half of the
// HBM rate), halo of input channel blockIdx.y in LDS, per thread 4 pixels x 8 channels of dy for each of the tile's four row pairs - the row pairs are unrolled, so the
// next pair's loads are in flight under the current pair's 288 FMAs.
template <typename T>
__global__ __launch_bounds__(256) void first_conv_wgrad_tiled_kernel(const float* __restrict__ x, int N, int Cin, int H, int W, const T* dy, int dy_ld,
                                                                     float* __restrict__ partial /*[blocks][Cin][10][64]*/) {
    constexpr int TH = 8, TW = 64, HH = TH + 2, HW = TW + 2, RP = TH / 2;
    __shared__ __attribute__((aligned(16))) float xs[HH][HW + 2];
    __shared__ float red[4][10 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = tid & 7, gq = tid >> 3;
    const int r = gq >> 4, wq = (gq & 15) * 4;
    const int ci = blockIdx.y;
    const int tilesW = (W + TW - 1) / TW, tilesH = (H + TH - 1) / TH;
    const long long ntiles = (long long)N * tilesH * tilesW;
    const long long HWp = (long long)H * W;
    float acc[10][8];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        long long b = tile;
        const int tw = (int)(b % tilesW);
        b /= tilesW;
        const int th = (int)(b % tilesH);
        const int n = (int)(b / tilesH);
        const int h0 = th * TH, w0 = tw * TW;
        __syncthreads();
        for (int i = tid; i < HH * HW; i += 256) {
            const int px = i % HW, py = i / HW;
            const int sy = h0 + py - 1, sx = w0 + px - 1;
            float v = 0.f;
            if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = x[((long long)n * Cin + ci) * HWp + (long long)sy * W + sx];
            xs[py][px] = v;
        }
        u32x4 raw[RP][4][sizeof(T) == 2 ? 1 : 2];
#pragma unroll
        for (int rp = 0; rp < RP; ++rp) {
            const int yy = h0 + 2 * rp + r;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int xx = w0 + wq + v;
#pragma unroll
                for (int k = 0; k < (sizeof(T) == 2 ? 1 : 2); ++k) raw[rp][v][k] = u32x4{0u, 0u, 0u, 0u};
                if (yy < H && xx < W) {
                    const T* src = dy + (((size_t)n * H + yy) * W + xx) * dy_ld + cg * 8;
                    raw[rp][v][0] = *reinterpret_cast<const u32x4*>(src);
                    if constexpr (sizeof(T) == 4) raw[rp][v][1] = *reinterpret_cast<const u32x4*>(src + 4);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int rp = 0; rp < RP; ++rp) {
            float g[4][8];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if constexpr (sizeof(T) == 2) {
                    unpack_chunk<__bf16>(raw[rp][v][0], g[v]);
                } else {
                    unpack_chunk<float>(raw[rp][v][0], g[v]);
                    unpack_chunk<float>(raw[rp][v][1], g[v] + 4);
                }
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                float xr[6];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(&xs[2 * rp + r + kh][wq]);
                xr[0] = lo[0]; xr[1] = lo[1]; xr[2] = lo[2]; xr[3] = lo[3];
                xr[4] = xs[2 * rp + r + kh][wq + 4];
                xr[5] = xs[2 * rp + r + kh][wq + 5];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[kh * 3 + kw][j] = fmaf(xr[v + kw], g[v][j], acc[kh * 3 + kw][j]);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[9][j] += g[v][j];
        }
    }
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[t][j];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[t][j] = v;
        }
    __syncthreads();
    if (lane < 8) {
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wave][t * 64 + lane * 8 + j] = acc[t][j];
    }
    __syncthreads();
    float* out = partial + ((size_t)blockIdx.x * Cin + ci) * 640;
    for (int i = tid; i < 640; i += 256) out[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

__global__ void first_conv_wgrad_reduce_kernel(const float* __restrict__ partial, int nblocks, int Cin, float* __restrict__ dw,
                                               float* __restrict__ db) {
    // 16 lanes per (ci, t, co): each sums every 16th block slab in a fixed order, then four shuffles (4 lanes per output left each lane 256 dependent loads: 92 us)
    const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
    const int idx = gidx >> 4, part = gidx & 15;
    const bool live = idx < Cin * 640;
    const int ci = live ? idx / 640 : 0, r = live ? idx - ci * 640 : 0;
    const int t = r >> 6, co = r & 63;
    float s = 0.f;
    if (live)
        for (int b = part; b < nblocks; b += 16) s += partial[((size_t)b * Cin + ci) * 640 + r];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    if (!live || part != 0) return;
    if (t < 9) dw[(co * Cin + ci) * 9 + t] = s;
    else if (ci == 0 && db != nullptr) db[co] = s;
}

extern "C" size_t mis_conv3x3_first_wgrad_workspace_bytes(int N, int Cin, int H, int W, int Cout) {
    (void)N; (void)H; (void)W; (void)Cout;
    return (size_t)FW_BLOCKS * Cin * 640 * sizeof(float);
}

extern "C" int mis_conv3x3_first_wgrad(int dtype, const float* x, int N, int Cin, int H, int W, const void* dy, int dy_ld, int Cout,
                                       float* workspace, float* dw, float* db, void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    MIS_REQUIRE(Cout == 64, MIS_EUNSUPPORTED, "first_wgrad: Cout must be 64");
    MIS_REQUIRE(Cin >= 1 && Cin <= 4, MIS_EUNSUPPORTED, "first_wgrad: Cin must be 1..4");
    MIS_REQUIRE(x && dy && workspace && dw, MIS_EINVAL, "first_wgrad: null pointer");
    MIS_REQUIRE(dy_ld % 8 == 0, MIS_EINVAL, "first_wgrad: dy_ld alignment");
    const long long npix = (long long)N * H * W;
    long long blocks = (npix + 31) / 32;
    if (blocks > FW_BLOCKS) blocks = FW_BLOCKS;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int tiled = !mis_sw(SW_FIRST2D_UNTILED);
    if (tiled) {
        const long long tiles = (long long)N * ((H + 7) / 8) * ((W + 63) / 64);
        blocks = tiles < FW_BLOCKS ? tiles : FW_BLOCKS;
        if (dtype == MIS_BF16)
            hipLaunchKernelGGL(first_conv_wgrad_tiled_kernel<__bf16>, dim3((unsigned)blocks, Cin), dim3(256), 0, s, x, N, Cin, H, W, (const __bf16*)dy,
                               dy_ld, workspace);
        else
            hipLaunchKernelGGL(first_conv_wgrad_tiled_kernel<float>, dim3((unsigned)blocks, Cin), dim3(256), 0, s, x, N, Cin, H, W, (const float*)dy,
                               dy_ld, workspace);
    } else if (dtype == MIS_BF16)
        hipLaunchKernelGGL(first_conv_wgrad_kernel<__bf16>, dim3((unsigned)blocks, Cin), dim3(256), 0, s, x, N, Cin, H, W, (const __bf16*)dy,
                           dy_ld, workspace);
    else
        hipLaunchKernelGGL(first_conv_wgrad_kernel<float>, dim3((unsigned)blocks, Cin), dim3(256), 0, s, x, N, Cin, H, W, (const float*)dy,
                           dy_ld, workspace);
    MIS_LAUNCH_CHECK("first_conv_wgrad");
    hipLaunchKernelGGL(first_conv_wgrad_reduce_kernel, dim3((Cin * 640 * 16 + 255) / 256), dim3(256), 0, s, (const float*)workspace, (int)blocks,
                       Cin, dw, db);
    MIS_LAUNCH_CHECK("first_conv_wgrad_reduce");
    return MIS_OK;
}

// =========================================================================================================
// Per-channel sums over pixels (and sums of squares), per sample:  sum[n][c], sumsq[n][c].
// Stage 1: blockIdx = (pixel slab, chunk group, n); stage 2 reduces the slabs in a fixed order.
// =========================================================================================================
constexpr int CS_SLABS = 1024;     // (256 until round 4: with ONE 16-byte load in flight per thread the pass ran at ~1 TB/s: 1.27 ms of a cfg5-shaped step for 4 GB)
template <typename T, bool SQ>
__global__ __launch_bounds__(256) void chansum_kernel(const T* __restrict__ x, int ld, long long npix, int C, float* __restrict__ part_sum,
                                                      float* __restrict__ part_sq) {
    constexpr int EPC = Tr<T>::EPC;
    __shared__ float red[256 * 8];
    __shared__ float red2[SQ ? 256 * 8 : 1];
    const int nchunks = C / EPC;
    const int chb = nchunks < 256 ? nchunks : 256;     // chunks handled per block (power of two for our shapes, not required)
    const int rows = 256 / chb;                        // pixel rows per block iteration
    const int tid = threadIdx.x;
    const int cl = tid % chb, r = tid / chb;
    const int chunk = blockIdx.y * chb + cl;
    const int n = blockIdx.z;
    const int nslabs = gridDim.x;
    float s[EPC], q[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = q[e] = 0.f;
    if (chunk < nchunks && r < rows) {
        const T* base = x + (size_t)n * npix * ld + (size_t)chunk * EPC;
        const long long stride = (long long)nslabs * rows;
        long long p = (long long)blockIdx.x * rows + r;
        // four independent 16-byte loads in flight per thread (the sums still take the pixels in index order)
        for (; p + 3 * stride < npix; p += 4 * stride) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4*>(base + (p + u * stride) * ld);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[EPC];
                unpack_chunk<T>(v[u], f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    s[e] += f[e];
                    if (SQ) q[e] = fmaf(f[e], f[e], q[e]);
                }
            }
        }
        for (; p < npix; p += stride) {
            float f[EPC];
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + p * ld), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                s[e] += f[e];
                if (SQ) q[e] = fmaf(f[e], f[e], q[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        red[tid * EPC + e] = s[e];
        if (SQ) red2[tid * EPC + e] = q[e];
    }
    __syncthreads();
    if (r == 0 && chunk < nchunks) {
        for (int k = 1; k < rows; ++k)
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                s[e] += red[(k * chb + cl) * EPC + e];
                if (SQ) q[e] += red2[(k * chb + cl) * EPC + e];
            }
        float* o = part_sum + ((size_t)n * nslabs + blockIdx.x) * C + (size_t)chunk * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = s[e];
        if (SQ) {
            float* o2 = part_sq + ((size_t)n * nslabs + blockIdx.x) * C + (size_t)chunk * EPC;
#pragma unroll
            for (int e = 0; e < EPC; ++e) o2[e] = q[e];
        }
    }
}

// out[n][c % Cq] = alpha * sum over slabs and folds.  16 outputs per block, 16 slab lanes per output (fixed order: lane partials, then lanes 0..15)
__global__ __launch_bounds__(256) void chansum_reduce_kernel(const float* __restrict__ part, int nslabs, int N, int C, int fold, float alpha,
                                                             float* __restrict__ out) {
    __shared__ double red[256];
    const int Cq = C / fold;
    const int o = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + o;
    double s = 0.0;
    if (idx < N * Cq) {
        const int n = idx / Cq, c = idx - n * Cq;
        for (int f = 0; f < fold; ++f) {
            const float* src = part + (size_t)n * nslabs * C + f * Cq + c;
            int k = lane;
            for (; k + 7 * 16 < nslabs; k += 8 * 16) {          // eight independent loads in flight (up to 1024 slabs since round 4), added in slab order
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u * 16) * C];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += (double)v[u];
            }
            for (; k < nslabs; k += 16) s += (double)src[(size_t)k * C];
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (lane == 0 && idx < N * Cq) {
        for (int l = 1; l < 16; ++l) s += red[l * 16 + o];
        out[idx] = (float)(alpha * s);
    }
}

static int chansum_slabs(long long npix, int C, int EPC) {
    const int nchunks = C / EPC;
    const int chb = nchunks < 256 ? nchunks : 256;
    const int rows = 256 / chb;
    long long want = (npix + rows * 8 - 1) / (rows * 8);
    if (want < 1) want = 1;
    if (want > CS_SLABS) want = CS_SLABS;
    return (int)want;
}

template <typename T, bool SQ>
static int chansum_launch(const void* x, int ld, int N, long long npix, int C, float* ws, float* sum, float* sumsq, int fold, float alpha,
                          hipStream_t s) {
    constexpr int EPC = Tr<T>::EPC;
    const int nchunks = C / EPC;
    const int chb = nchunks < 256 ? nchunks : 256;
    const int ngroups = (nchunks + chb - 1) / chb;
    const int nslabs = chansum_slabs(npix, C, EPC);
    float* ps = ws;
    float* pq = ws + (size_t)N * CS_SLABS * C;
    hipLaunchKernelGGL((chansum_kernel<T, SQ>), dim3(nslabs, ngroups, N), dim3(256), 0, s, (const T*)x, ld, npix, C, ps, pq);
    MIS_LAUNCH_CHECK("chansum");
    const int Cq = C / fold;
    hipLaunchKernelGGL(chansum_reduce_kernel, dim3((N * Cq + 15) / 16), dim3(256), 0, s, (const float*)ps, nslabs, N, C, fold, alpha, sum);
    MIS_LAUNCH_CHECK("chansum_reduce");
    if (SQ) {
        hipLaunchKernelGGL(chansum_reduce_kernel, dim3((N * Cq + 15) / 16), dim3(256), 0, s, (const float*)pq, nslabs, N, C, fold, alpha, sumsq);
        MIS_LAUNCH_CHECK("chansum_reduce_sq");
    }
    return MIS_OK;
}

// BatchNorm2d + ReLU backward without a materialised g = dy * (y > 0): the mask is recomputed from z as fma(z, scale, shift) > 0 (the expression
// mis_affine_act evaluated in the forward), so the two passes read dy and z only.
//   stats: S1[n][c] = sum_pix m * dy,  S2[n][c] = sum_pix m * dy * z           apply: dz = p * (m * dy) + q * z + r
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const T* __restrict__ dy, int dy_ld, const T* __restrict__ z, int z_ld, long long npix, int C,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ part1, float* __restrict__ part2) {
    constexpr int EPC = Tr<T>::EPC;
    __shared__ float red[256 * 8];
    __shared__ float red2[256 * 8];
    const int nchunks = C / EPC;
    const int chb = nchunks < 256 ? nchunks : 256;
    const int rows = 256 / chb;
    const int tid = threadIdx.x;
    const int cl = tid % chb, r = tid / chb;
    const int chunk = blockIdx.y * chb + cl;
    const int n = blockIdx.z;
    const int nslabs = gridDim.x;
    float s[EPC], q[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = q[e] = 0.f;
    if (chunk < nchunks && r < rows) {
        float sc[EPC], sh[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            sc[e] = scale[(size_t)n * C + chunk * EPC + e];
            sh[e] = shift[(size_t)n * C + chunk * EPC + e];
        }
        const T* gb = dy + (size_t)n * npix * dy_ld + (size_t)chunk * EPC;
        const T* zb = z + (size_t)n * npix * z_ld + (size_t)chunk * EPC;
        for (long long p = (long long)blockIdx.x * rows + r; p < npix; p += (long long)nslabs * rows) {
            float g[EPC], f[EPC];
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(gb + p * dy_ld), g);
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(zb + p * z_ld), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float m = fmaf(f[e], sc[e], sh[e]) > 0.f ? g[e] : 0.f;
                s[e] += m;
                q[e] = fmaf(m, f[e], q[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        red[tid * EPC + e] = s[e];
        red2[tid * EPC + e] = q[e];
    }
    __syncthreads();
    if (r == 0 && chunk < nchunks) {
        for (int k = 1; k < rows; ++k)
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                s[e] += red[(k * chb + cl) * EPC + e];
                q[e] += red2[(k * chb + cl) * EPC + e];
            }
        float* o = part1 + ((size_t)n * nslabs + blockIdx.x) * C + (size_t)chunk * EPC;
        float* o2 = part2 + ((size_t)n * nslabs + blockIdx.x) * C + (size_t)chunk * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            o[e] = s[e];
            o2[e] = q[e];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, int dy_ld, const T* __restrict__ z, int z_ld, int N, long long npix, int C,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ p,
                                                           const float* __restrict__ q, const float* __restrict__ r, T* __restrict__ dz, int dz_ld) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC;
    const long long total = (long long)N * npix * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long long pp = i / nch;
        const size_t nc = (size_t)(pp / npix) * C + (size_t)ch * EPC;
        float g[EPC], f[EPC], o[EPC];
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(dy + (size_t)pp * dy_ld + (size_t)ch * EPC), g);
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(z + (size_t)pp * z_ld + (size_t)ch * EPC), f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float m = fmaf(f[e], scale[nc + e], shift[nc + e]) > 0.f ? g[e] : 0.f;
            o[e] = fmaf(p[nc + e], m, fmaf(q[nc + e], f[e], r[nc + e]));
        }
        *reinterpret_cast<u32x4*>(dz + (size_t)pp * dz_ld + (size_t)ch * EPC) = pack_chunk<T>(o);
    }
}

extern "C" size_t mis_bn_bwd_stats_workspace_bytes(int N, int C) { return (size_t)2 * N * CS_SLABS * C * sizeof(float); }

extern "C" int mis_bn_bwd_stats(int dtype, const void* dy, int dy_ld, const void* z, int z_ld, int N, long long npix, int C, const float* scale,
                                const float* shift, float* workspace, float* S1, float* S2, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "bn_bwd_stats: bad dtype %d", dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(dy && z && scale && shift && workspace && S1 && S2 && npix > 0 && N > 0, MIS_EINVAL, "bn_bwd_stats: bad argument");
    MIS_REQUIRE(C > 0 && C % EPC == 0 && dy_ld % EPC == 0 && z_ld % EPC == 0 && dy_ld >= C && z_ld >= C, MIS_EINVAL, "bn_bwd_stats: C / ld alignment");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nchunks = C / EPC;
    const int chb = nchunks < 256 ? nchunks : 256;
    const int ngroups = (nchunks + chb - 1) / chb;
    const int nslabs = chansum_slabs(npix, C, EPC);
    float* p1 = workspace;
    float* p2 = workspace + (size_t)N * CS_SLABS * C;
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(bn_bwd_stats_kernel<__bf16>, dim3(nslabs, ngroups, N), dim3(256), 0, s, (const __bf16*)dy, dy_ld, (const __bf16*)z, z_ld, npix, C,
                           scale, shift, p1, p2);
    else
        hipLaunchKernelGGL(bn_bwd_stats_kernel<float>, dim3(nslabs, ngroups, N), dim3(256), 0, s, (const float*)dy, dy_ld, (const float*)z, z_ld, npix, C, scale,
                           shift, p1, p2);
    MIS_LAUNCH_CHECK("bn_bwd_stats");
    hipLaunchKernelGGL(chansum_reduce_kernel, dim3((N * C + 15) / 16), dim3(256), 0, s, (const float*)p1, nslabs, N, C, 1, 1.f, S1);
    hipLaunchKernelGGL(chansum_reduce_kernel, dim3((N * C + 15) / 16), dim3(256), 0, s, (const float*)p2, nslabs, N, C, 1, 1.f, S2);
    MIS_LAUNCH_CHECK("bn_bwd_stats_reduce");
    return MIS_OK;
}

extern "C" int mis_bn_bwd_apply(int dtype, const void* dy, int dy_ld, const void* z, int z_ld, int N, long long npix, int C, const float* scale,
                                const float* shift, const float* p, const float* q, const float* r, void* dz, int dz_ld, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "bn_bwd_apply: bad dtype %d", dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(dy && z && scale && shift && p && q && r && dz && npix > 0 && N > 0, MIS_EINVAL, "bn_bwd_apply: bad argument");
    MIS_REQUIRE(C > 0 && C % EPC == 0 && dy_ld % EPC == 0 && z_ld % EPC == 0 && dz_ld % EPC == 0 && dy_ld >= C && z_ld >= C && dz_ld >= C, MIS_EINVAL,
                "bn_bwd_apply: C / ld alignment");
    long long blocks = ((long long)N * npix * (C / EPC) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)dy, dy_ld, (const __bf16*)z, z_ld, N, npix, C,
                           scale, shift, p, q, r, (__bf16*)dz, dz_ld);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)dy, dy_ld, (const float*)z, z_ld, N, npix, C, scale,
                           shift, p, q, r, (float*)dz, dz_ld);
    MIS_LAUNCH_CHECK("bn_bwd_apply");
    return MIS_OK;
}

extern "C" size_t mis_colsum_workspace_bytes(long long npix, int C) {
    (void)npix;
    return (size_t)CS_SLABS * C * sizeof(float);
}
extern "C" int mis_colsum(int dtype, const void* x, int ld, long long npix, int C, int fold, float alpha, float* workspace, float* out,
                          void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(x && workspace && out && npix > 0, MIS_EINVAL, "colsum: bad argument");
    MIS_REQUIRE(C % EPC == 0 && ld % EPC == 0, MIS_EINVAL, "colsum: C / ld alignment");
    MIS_REQUIRE(fold >= 1 && C % fold == 0, MIS_EINVAL, "colsum: fold");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16) return chansum_launch<__bf16, false>(x, ld, 1, npix, C, workspace, out, nullptr, fold, alpha, s);
    return chansum_launch<float, false>(x, ld, 1, npix, C, workspace, out, nullptr, fold, alpha, s);
}
extern "C" size_t mis_chanstats_workspace_bytes(int N, long long npix, int C) {
    (void)npix;
    return (size_t)2 * N * CS_SLABS * C * sizeof(float);
}
extern "C" int mis_chanstats(int dtype, const void* x, int ld, int N, long long npix, int C, float* workspace, float* sum, float* sumsq,
                             void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(x && workspace && sum && sumsq && npix > 0 && N > 0, MIS_EINVAL, "chanstats: bad argument");
    MIS_REQUIRE(C % EPC == 0 && ld % EPC == 0, MIS_EINVAL, "chanstats: C / ld alignment");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16) return chansum_launch<__bf16, true>(x, ld, N, npix, C, workspace, sum, sumsq, 1, 1.f, s);
    return chansum_launch<float, true>(x, ld, N, npix, C, workspace, sum, sumsq, 1, 1.f, s);
}

// =========================================================================================================
// MaxPool 2 (2-D: 2x2, 3-D: 2x2x2), stride 2, floor.  One thread per (output pixel, 16-byte channel chunk).
// bwd recomputes the arg-max in PyTorch's scan order (kd, kh, kw; first maximum wins) instead of storing indices.
// =========================================================================================================
template <typename T, bool IS3D>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, int x_ld, T* __restrict__ y, int y_ld, int N, int D, int H,
                                                          int W, int C) {
    constexpr int EPC = Tr<T>::EPC;
    const int OD = IS3D ? D / 2 : 1, OH = H / 2, OW = W / 2;
    const int nch = C / EPC;
    const long long total = (long long)N * OD * OH * OW * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long op = i / nch;
        const int ox = (int)(op % OW); op /= OW;
        const int oy = (int)(op % OH); op /= OH;
        const int oz = (int)(op % OD);
        const int n = (int)(op / OD);
        float m[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int kd = 0; kd < (IS3D ? 2 : 1); ++kd)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int kw = 0; kw < 2; ++kw) {
                    const int z = IS3D ? 2 * oz + kd : 0;
                    const size_t pix = (((size_t)n * D + z) * H + 2 * oy + kh) * W + 2 * ox + kw;
                    float f[EPC];
                    unpack_chunk<T>(*reinterpret_cast<const u32x4*>(x + pix * x_ld + (size_t)ch * EPC), f);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) m[e] = (f[e] > m[e]) ? f[e] : m[e];
                }
        const size_t opix = (((size_t)n * OD + oz) * OH + oy) * OW + ox;
        *reinterpret_cast<u32x4*>(y + opix * y_ld + (size_t)ch * EPC) = pack_chunk<T>(m);
    }
}

template <typename T, bool IS3D>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, int x_ld, const T* __restrict__ dy, int dy_ld, const T* add,
                                                          int add_ld, T* dx, int dx_ld, int N, int D, int H, int W, int C, int relu_mask) {
    constexpr int EPC = Tr<T>::EPC;
    constexpr int KD = IS3D ? 2 : 1;
    const int OD = IS3D ? D / 2 : 1, OH = H / 2, OW = W / 2;
    // windows cover ceil(size/2) so that trailing odd rows/cols (no pooling window) still receive mask(add)
    const int WD = IS3D ? (D + 1) / 2 : 1, WH = (H + 1) / 2, WW = (W + 1) / 2;
    const int nch = C / EPC;
    const long long total = (long long)N * WD * WH * WW * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long op = i / nch;
        const int ox = (int)(op % WW); op /= WW;
        const int oy = (int)(op % WH); op /= WH;
        const int oz = (int)(op % WD);
        const int n = (int)(op / WD);
        const bool pooled = (oz < OD) && (oy < OH) && (ox < OW);
        float g[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) g[e] = 0.f;
        if (pooled) {
            const size_t opix = (((size_t)n * OD + oz) * OH + oy) * OW + ox;
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(dy + opix * dy_ld + (size_t)ch * EPC), g);
        }
        float xv[KD * 4][EPC];
        bool inb[KD * 4];
        size_t pixs[KD * 4];
#pragma unroll
        for (int k = 0; k < KD * 4; ++k) {
            const int kd = k >> 2, kh = (k >> 1) & 1, kw = k & 1;
            const int z = IS3D ? 2 * oz + kd : 0, yy = 2 * oy + kh, xx = 2 * ox + kw;
            inb[k] = (z < D) && (yy < H) && (xx < W);
            pixs[k] = (((size_t)n * D + z) * H + yy) * W + xx;
#pragma unroll
            for (int e = 0; e < EPC; ++e) xv[k][e] = -INFINITY;
            if (inb[k]) unpack_chunk<T>(*reinterpret_cast<const u32x4*>(x + pixs[k] * x_ld + (size_t)ch * EPC), xv[k]);
        }
        int amax[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float m = xv[0][e];
            int a = 0;
#pragma unroll
            for (int k = 1; k < KD * 4; ++k)
                if (xv[k][e] > m) {
                    m = xv[k][e];
                    a = k;
                }
            amax[e] = a;
        }
#pragma unroll
        for (int k = 0; k < KD * 4; ++k) {
            if (!inb[k]) continue;
            float o[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) o[e] = 0.f;
            if (add != nullptr) unpack_chunk<T>(*reinterpret_cast<const u32x4*>(add + pixs[k] * add_ld + (size_t)ch * EPC), o);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                if (pooled && amax[e] == k) o[e] += g[e];
                if (relu_mask && !(xv[k][e] > 0.f)) o[e] = 0.f;
            }
            *reinterpret_cast<u32x4*>(dx + pixs[k] * dx_ld + (size_t)ch * EPC) = pack_chunk<T>(o);
        }
    }
}

// ---- 2-D max-pool with "pool bits" (round 3): the forward pass also writes, per POOLED element, one byte = [bit k: window position k = kh*2 + kw is the arg-max
// (first maximum in scan order)] | [bit 4 + k: the input at position k is > 0]; the backward pass then needs neither the input tensor (a third of its traffic in the
// 2-D net: dx = relu_mask(x) * (add + scatter(dy))) nor the comparison.  pbits: (N, H/2, W/2, C) bytes; even H and W.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_pb_kernel(const T* __restrict__ x, int x_ld, T* __restrict__ y, int y_ld, int N, int H, int W, int C,
                                                             unsigned char* __restrict__ pbits) {
    constexpr int EPC = Tr<T>::EPC;
    const int OH = H / 2, OW = W / 2;
    const int nch = C / EPC;
    const long long total = (long long)N * OH * OW * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long long opix = i / nch;
        const int ox = (int)(opix % OW);
        const long long r = opix / OW;
        const int oy = (int)(r % OH), n = (int)(r / OH);
        float m[EPC];
        unsigned bits[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            m[e] = -INFINITY;
            bits[e] = 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t pix = ((size_t)n * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1);
            float f[EPC];
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(x + pix * x_ld + (size_t)ch * EPC), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                if (f[e] > m[e]) {                    // strict: the first maximum wins, as in maxpool_bwd_kernel / PyTorch
                    m[e] = f[e];
                    bits[e] = (bits[e] & 0xf0u) | (1u << k);
                }
                if (f[e] > 0.f) bits[e] |= 0x10u << k;
            }
        }
        *reinterpret_cast<u32x4*>(y + (size_t)opix * y_ld + (size_t)ch * EPC) = pack_chunk<T>(m);
        unsigned char* bp = pbits + (size_t)opix * C + (size_t)ch * EPC;
        if constexpr (EPC == 8) {
            *reinterpret_cast<u32x2*>(bp) = u32x2{bits[0] | (bits[1] << 8) | (bits[2] << 16) | (bits[3] << 24), bits[4] | (bits[5] << 8) | (bits[6] << 16) | (bits[7] << 24)};
        } else {
            *reinterpret_cast<uint32_t*>(bp) = bits[0] | (bits[1] << 8) | (bits[2] << 16) | (bits[3] << 24);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_pb_kernel(const unsigned char* __restrict__ pbits, const T* __restrict__ dy, int dy_ld, const T* add, int add_ld,
                                                             T* dx, int dx_ld, int N, int H, int W, int C) {
    constexpr int EPC = Tr<T>::EPC;
    const int OH = H / 2, OW = W / 2;
    const int nch = C / EPC;
    const long long total = (long long)N * OH * OW * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long long opix = i / nch;
        const int ox = (int)(opix % OW);
        const long long r = opix / OW;
        const int oy = (int)(r % OH), n = (int)(r / OH);
        float g[EPC];
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(dy + (size_t)opix * dy_ld + (size_t)ch * EPC), g);
        unsigned bits[EPC];
        const unsigned char* bp = pbits + (size_t)opix * C + (size_t)ch * EPC;
        if constexpr (EPC == 8) {
            const u32x2 b2 = *reinterpret_cast<const u32x2*>(bp);
#pragma unroll
            for (int e = 0; e < 8; ++e) bits[e] = (b2[e >> 2] >> ((e & 3) * 8)) & 0xffu;
        } else {
            const uint32_t b1 = *reinterpret_cast<const uint32_t*>(bp);
#pragma unroll
            for (int e = 0; e < 4; ++e) bits[e] = (b1 >> (e * 8)) & 0xffu;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t pix = ((size_t)n * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1);
            float o[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) o[e] = 0.f;
            if (add != nullptr) unpack_chunk<T>(*reinterpret_cast<const u32x4*>(add + pix * add_ld + (size_t)ch * EPC), o);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                if ((bits[e] >> k) & 1u) o[e] += g[e];
                if (!((bits[e] >> (4 + k)) & 1u)) o[e] = 0.f;
            }
            *reinterpret_cast<u32x4*>(dx + pix * dx_ld + (size_t)ch * EPC) = pack_chunk<T>(o);
        }
    }
}

static unsigned capped_grid(long long total, int per_block) {
    long long b = (total + per_block - 1) / per_block;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

extern "C" int mis_maxpool2_fwd(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, int D, int H, int W, int C, void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(x && y && N > 0 && D > 0 && H > 1 && W > 1, MIS_EINVAL, "maxpool_fwd: bad argument");
    MIS_REQUIRE(C % EPC == 0 && x_ld % EPC == 0 && y_ld % EPC == 0, MIS_EINVAL, "maxpool_fwd: alignment");
    const bool is3d = D > 1;
    const long long total = (long long)N * (is3d ? D / 2 : 1) * (H / 2) * (W / 2) * (C / EPC);
    const unsigned g = capped_grid(total, 256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define MP_FWD(T, B3) hipLaunchKernelGGL((maxpool_fwd_kernel<T, B3>), dim3(g), dim3(256), 0, s, (const T*)x, x_ld, (T*)y, y_ld, N, D, H, W, C)
    if (dtype == MIS_BF16) { if (is3d) MP_FWD(__bf16, true); else MP_FWD(__bf16, false); }
    else { if (is3d) MP_FWD(float, true); else MP_FWD(float, false); }
#undef MP_FWD
    MIS_LAUNCH_CHECK("maxpool_fwd");
    return MIS_OK;
}

extern "C" int mis_maxpool2_fwd_pb(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, int H, int W, int C, void* pbits, void* stream) {
    (void)hipGetLastError();
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(x && y && pbits && N > 0 && H > 1 && W > 1, MIS_EINVAL, "maxpool_fwd_pb: bad argument");
    MIS_REQUIRE(H % 2 == 0 && W % 2 == 0, MIS_EUNSUPPORTED, "maxpool_fwd_pb: even H and W (got %d x %d)", H, W);
    MIS_REQUIRE(C % EPC == 0 && x_ld % EPC == 0 && y_ld % EPC == 0, MIS_EINVAL, "maxpool_fwd_pb: alignment");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / EPC);
    const unsigned g = capped_grid(total, 256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(maxpool_fwd_pb_kernel<__bf16>, dim3(g), dim3(256), 0, s, (const __bf16*)x, x_ld, (__bf16*)y, y_ld, N, H, W, C, (unsigned char*)pbits);
    else
        hipLaunchKernelGGL(maxpool_fwd_pb_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, x_ld, (float*)y, y_ld, N, H, W, C, (unsigned char*)pbits);
    MIS_LAUNCH_CHECK("maxpool_fwd_pb");
    return MIS_OK;
}

extern "C" int mis_maxpool2_bwd_pb(int dtype, const void* pbits, const void* dy, int dy_ld, const void* add, int add_ld, void* dx, int dx_ld, int N, int H, int W,
                                   int C, void* stream) {
    (void)hipGetLastError();
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(pbits && dy && dx && N > 0 && H > 1 && W > 1, MIS_EINVAL, "maxpool_bwd_pb: bad argument");
    MIS_REQUIRE(H % 2 == 0 && W % 2 == 0, MIS_EUNSUPPORTED, "maxpool_bwd_pb: even H and W (got %d x %d)", H, W);
    MIS_REQUIRE(C % EPC == 0 && dy_ld % EPC == 0 && dx_ld % EPC == 0 && (add == nullptr || add_ld % EPC == 0), MIS_EINVAL, "maxpool_bwd_pb: alignment");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / EPC);
    const unsigned g = capped_grid(total, 256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(maxpool_bwd_pb_kernel<__bf16>, dim3(g), dim3(256), 0, s, (const unsigned char*)pbits, (const __bf16*)dy, dy_ld, (const __bf16*)add, add_ld,
                           (__bf16*)dx, dx_ld, N, H, W, C);
    else
        hipLaunchKernelGGL(maxpool_bwd_pb_kernel<float>, dim3(g), dim3(256), 0, s, (const unsigned char*)pbits, (const float*)dy, dy_ld, (const float*)add, add_ld,
                           (float*)dx, dx_ld, N, H, W, C);
    MIS_LAUNCH_CHECK("maxpool_bwd_pb");
    return MIS_OK;
}

extern "C" int mis_maxpool2_bwd(int dtype, const void* x, int x_ld, const void* dy, int dy_ld, const void* add, int add_ld, void* dx,
                                int dx_ld, int N, int D, int H, int W, int C, int relu_mask, void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(x && dy && dx && N > 0 && D > 0 && H > 1 && W > 1, MIS_EINVAL, "maxpool_bwd: bad argument");
    MIS_REQUIRE(C % EPC == 0 && x_ld % EPC == 0 && dy_ld % EPC == 0 && dx_ld % EPC == 0, MIS_EINVAL, "maxpool_bwd: alignment");
    MIS_REQUIRE(add == nullptr || add_ld % EPC == 0, MIS_EINVAL, "maxpool_bwd: add_ld alignment");
    const bool is3d = D > 1;
    const long long total = (long long)N * (is3d ? (D + 1) / 2 : 1) * ((H + 1) / 2) * ((W + 1) / 2) * (C / EPC);
    const unsigned g = capped_grid(total, 256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define MP_BWD(T, B3)                                                                                                             \
    hipLaunchKernelGGL((maxpool_bwd_kernel<T, B3>), dim3(g), dim3(256), 0, s, (const T*)x, x_ld, (const T*)dy, dy_ld, (const T*)add, \
                       add_ld, (T*)dx, dx_ld, N, D, H, W, C, relu_mask)
    if (dtype == MIS_BF16) { if (is3d) MP_BWD(__bf16, true); else MP_BWD(__bf16, false); }
    else { if (is3d) MP_BWD(float, true); else MP_BWD(float, false); }
#undef MP_BWD
    MIS_LAUNCH_CHECK("maxpool_bwd");
    return MIS_OK;
}

// =========================================================================================================
// Weight repack (fp32 master, reference layout) -> MFMA operand layouts.  32x32 (co x ci) tiles through LDS so
// both outputs are written with the contiguous index on consecutive lanes.
// =========================================================================================================
// one 32 x 32 (co x ci) tile of one layer; `tile` = 9 x 32 x 33 floats of LDS
template <typename T>
__device__ __forceinline__ void pack_conv_tile(float (*tile)[32][33], const float* __restrict__ w, int Cout, int Cin, int taps, T* __restrict__ wf,
                                               T* __restrict__ wd, int co0, int ci0) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int t0 = 0; t0 < taps; t0 += 9) {
        const int nt = (taps - t0) < 9 ? (taps - t0) : 9;
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int co = co0 + r, ci = ci0 + tx;
            if (co < Cout && ci < Cin) {
                const float* src = w + ((size_t)co * Cin + ci) * taps + t0;
                for (int t = 0; t < nt; ++t) {
                    const float v = src[t];
                    tile[t][r][tx] = v;
                    st_elem<T>(wf + ((size_t)(t0 + t) * Cout + co) * Cin + ci, v);
                }
            }
        }
        __syncthreads();
        if (wd != nullptr) {
            for (int r = ty; r < 32; r += 8) {
                const int ci = ci0 + r, co = co0 + tx;
                if (co < Cout && ci < Cin)
                    for (int t = 0; t < nt; ++t) st_elem<T>(wd + ((size_t)(taps - 1 - (t0 + t)) * Cin + ci) * Cout + co, tile[t][tx][r]);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, T* __restrict__ wf,
                                                        T* __restrict__ wd) {
    __shared__ float tile[9][32][33];
    pack_conv_tile<T>(tile, w, Cout, Cin, taps, wf, wd, blockIdx.y * 32, blockIdx.x * 32);
}

extern "C" int mis_pack_conv_weight(int dtype, const float* w, int Cout, int Cin, int taps, void* w_fwd, void* w_dgrad, void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    MIS_REQUIRE(w && w_fwd && Cout > 0 && Cin > 0 && taps > 0, MIS_EINVAL, "pack_conv: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 g((Cin + 31) / 32, (Cout + 31) / 32);
    if (dtype == MIS_BF16) hipLaunchKernelGGL(pack_conv_kernel<__bf16>, g, dim3(256), 0, s, w, Cout, Cin, taps, (__bf16*)w_fwd, (__bf16*)w_dgrad);
    else hipLaunchKernelGGL(pack_conv_kernel<float>, g, dim3(256), 0, s, w, Cout, Cin, taps, (float*)w_fwd, (float*)w_dgrad);
    MIS_LAUNCH_CHECK("pack_conv");
    return MIS_OK;
}

// convT k2s2: w [Cin][Cq][4] -> fwd [ab*Cq + c][ci], dgrad [ci][ab*Cq + c]
template <typename T>
__device__ __forceinline__ void pack_convt_tile(float (*tile)[32][33], const float* __restrict__ w, int Cin, int Cq, T* __restrict__ wf, T* __restrict__ wd,
                                                int ci0, int c0) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, c = c0 + tx;
        if (ci < Cin && c < Cq) {
            const float* src = w + ((size_t)ci * Cq + c) * 4;
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                const float v = src[ab];
                tile[ab][r][tx] = v;
                st_elem<T>(wd + (size_t)ci * (4 * Cq) + ab * Cq + c, v);
            }
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, ci = ci0 + tx;
        if (ci < Cin && c < Cq) {
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) st_elem<T>(wf + ((size_t)ab * Cq + c) * Cin + ci, tile[ab][tx][r]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_convt_kernel(const float* __restrict__ w, int Cin, int Cq, T* __restrict__ wf, T* __restrict__ wd) {
    __shared__ float tile[4][32][33];
    pack_convt_tile<T>(tile, w, Cin, Cq, wf, wd, blockIdx.y * 32, blockIdx.x * 32);
}

// every layer of a network in ONE launch: blockIdx.z = entry of a device-resident table (the per-layer launches cost ~12 us each, 0.26 ms per 2-D train step)
template <typename T>
__global__ __launch_bounds__(256) void pack_batch_kernel(const MisPackItem* __restrict__ items) {
    __shared__ float tile[9][32][33];
    const MisPackItem it = items[blockIdx.z];
    if (it.kind == 0) {
        const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
        if (co0 >= it.rows || ci0 >= it.cols) return;             // block-uniform
        pack_conv_tile<T>(tile, it.w, it.rows, it.cols, it.taps, (T*)it.w_fwd, (T*)it.w_dgrad, co0, ci0);
    } else {
        const int ci0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
        if (ci0 >= it.rows || c0 >= it.cols) return;
        pack_convt_tile<T>(tile, it.w, it.rows, it.cols, (T*)it.w_fwd, (T*)it.w_dgrad, ci0, c0);
    }
}

// ... the same with a COMPACT grid (round 4): an entry carries the index of its first block (`blk0`, ascending) and its tile columns, so the launch has exactly the blocks
// that do work - the (max_cols / 32, max_rows / 32, n) grid above scheduled 21.5 k blocks for the 2-D net, 17 k of which exit at once (pack: 132 us for 250 MB)
template <typename T>
__global__ __launch_bounds__(256) void pack_batch2_kernel(const MisPackItem2* __restrict__ items, int n) {
    __shared__ float tile[9][32][33];
    const int bid = blockIdx.x;
    int lo = 0, hi = n - 1;                                   // last entry with blk0 <= bid (block-uniform binary search)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].blk0 <= bid) lo = mid;
        else hi = mid - 1;
    }
    const MisPackItem2 it = items[lo];
    const int local = bid - it.blk0;
    const int by = local / it.nbx, bx = local - by * it.nbx;
    if (it.kind == 0) pack_conv_tile<T>(tile, it.w, it.rows, it.cols, it.taps, (T*)it.w_fwd, (T*)it.w_dgrad, by * 32, bx * 32);
    else pack_convt_tile<T>(tile, it.w, it.rows, it.cols, (T*)it.w_fwd, (T*)it.w_dgrad, by * 32, bx * 32);
}

extern "C" int mis_pack_batch2(int dtype, const MisPackItem2* items_dev, int n, int total_blocks, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "pack_batch2: bad dtype %d", dtype);
    MIS_REQUIRE(items_dev && n > 0 && n <= 65535 && total_blocks > 0, MIS_EINVAL, "pack_batch2: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16) hipLaunchKernelGGL(pack_batch2_kernel<__bf16>, dim3((unsigned)total_blocks), dim3(256), 0, s, items_dev, n);
    else hipLaunchKernelGGL(pack_batch2_kernel<float>, dim3((unsigned)total_blocks), dim3(256), 0, s, items_dev, n);
    MIS_LAUNCH_CHECK("pack_batch2");
    return MIS_OK;
}

extern "C" int mis_pack_batch(int dtype, const MisPackItem* items_dev, int n, int max_rows, int max_cols, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "pack_batch: bad dtype %d", dtype);
    MIS_REQUIRE(items_dev && n > 0 && n <= 65535 && max_rows > 0 && max_cols > 0, MIS_EINVAL, "pack_batch: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 g((max_cols + 31) / 32, (max_rows + 31) / 32, n);
    MIS_REQUIRE(g.y <= 65535, MIS_EUNSUPPORTED, "pack_batch: a layer with %d rows", max_rows);
    if (dtype == MIS_BF16) hipLaunchKernelGGL(pack_batch_kernel<__bf16>, g, dim3(256), 0, s, items_dev);
    else hipLaunchKernelGGL(pack_batch_kernel<float>, g, dim3(256), 0, s, items_dev);
    MIS_LAUNCH_CHECK("pack_batch");
    return MIS_OK;
}

extern "C" int mis_pack_convt_weight(int dtype, const float* w, int Cin, int Cq, void* w_fwd, void* w_dgrad, void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    MIS_REQUIRE(w && w_fwd && w_dgrad && Cin > 0 && Cq > 0, MIS_EINVAL, "pack_convt: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 g((Cq + 31) / 32, (Cin + 31) / 32);
    if (dtype == MIS_BF16) hipLaunchKernelGGL(pack_convt_kernel<__bf16>, g, dim3(256), 0, s, w, Cin, Cq, (__bf16*)w_fwd, (__bf16*)w_dgrad);
    else hipLaunchKernelGGL(pack_convt_kernel<float>, g, dim3(256), 0, s, w, Cin, Cq, (float*)w_fwd, (float*)w_dgrad);
    MIS_LAUNCH_CHECK("pack_convt");
    return MIS_OK;
}

// =========================================================================================================
// Layout converts (module boundaries / tests; not on the fused hot path)
// =========================================================================================================
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int y_ld, int N, int C, long long S) {
    const long long total = (long long)N * C * S;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long ps = i / C;            // n*S + s
        const long long n = ps / S, sp = ps - n * S;
        st_elem<T>(y + ps * y_ld + c, x[(n * C + c) * S + sp]);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, int x_ld, float* __restrict__ y, int N, int C, long long S) {
    const long long total = (long long)N * C * S;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long sp = i % S;
        const long long nc = i / S;
        const long long n = nc / C;
        const int c = (int)(nc - n * C);
        y[i] = ld_elem<T>(x + (n * S + sp) * x_ld + c);
    }
}
extern "C" int mis_nchw_to_nhwc(int dtype_out, const float* x, void* y, int y_ld, int N, int C, long long spatial, void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    MIS_REQUIRE(x && y && N > 0 && C > 0 && spatial > 0, MIS_EINVAL, "nchw_to_nhwc: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = capped_grid((long long)N * C * spatial, 256);
    if (dtype_out == MIS_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, dim3(g), dim3(256), 0, s, x, (__bf16*)y, y_ld, N, C, spatial);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(g), dim3(256), 0, s, x, (float*)y, y_ld, N, C, spatial);
    MIS_LAUNCH_CHECK("nchw_to_nhwc");
    return MIS_OK;
}
extern "C" int mis_nhwc_to_nchw(int dtype_in, const void* x, int x_ld, float* y, int N, int C, long long spatial, void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    MIS_REQUIRE(x && y && N > 0 && C > 0 && spatial > 0, MIS_EINVAL, "nhwc_to_nchw: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = capped_grid((long long)N * C * spatial, 256);
    if (dtype_in == MIS_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<__bf16>, dim3(g), dim3(256), 0, s, (const __bf16*)x, x_ld, y, N, C, spatial);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, x_ld, y, N, C, spatial);
    MIS_LAUNCH_CHECK("nhwc_to_nchw");
    return MIS_OK;
}
