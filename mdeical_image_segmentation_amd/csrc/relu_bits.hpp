// "ReLU bits": one bit per element of a bf16 NHWC activation tensor, bit = (value > 0) - what the ReLU backward needs of it (reference: the `result > 0` mask inside
// threshold_backward for nn.ReLU, model/unet2d/layers.py:20-25).  A masked dgrad (mis_conv_igemm with `mask`) reads 2 bytes per output element only for its sign; with
// `mask_bits` it reads 1 bit: the full-resolution 64-channel layers of the 2-D net are HBM-bound and their mask is a third of their traffic, and in the column-segment kernels
// a lane's whole tile of mask rows becomes ONE 16-byte load that is issued a K chunk ahead instead of sixteen exposed ones.
//
// Layout (C % 64 == 0): records of 64 bytes, one per (sample n, block of 8 rows yb = y >> 3, column x, block of 64 channels cb = c >> 6):
//     record index = ((n * H8 + yb) * W + x) * (C / 64) + cb,        H8 = ceil(H / 8)
//     byte inside the record = ((g & 3) * 2 + (g >> 2)) * 8 + (y & 7),   g = (c >> 3) & 7 the group of 8 channels,     bit = c & 7
// i.e. a record is [4][2][8 rows] bytes.  The order is that of the column-segment kernels' lanes (conv_pp_common.hpp, pp_epilogue_plain): lane group lg holds channel
// groups g = lg and g = lg + 4 of its wave's 64 channels for 8 consecutive rows - exactly the 16 contiguous bytes [lg][0..1][0..7].  Any other kernel addresses single
// bytes (8 channels of one pixel) with rb_byte_offset.  Rows past H inside the last block are never read.
#pragma once
#include <stddef.h>

__host__ __device__ inline size_t rb_bytes(long long N, int H, int W, int C) { return (size_t)N * ((H + 7) / 8) * W * (size_t)C; }
// byte that holds channels [c8 * 8, c8 * 8 + 8) of pixel (n, y, x); H8 = ceil(H / 8), C64 = C / 64
__host__ __device__ inline size_t rb_byte_offset(int H8, int W, int C64, int n, int y, int x, int c8) {
    const int g = c8 & 7;
    return ((((size_t)n * H8 + (y >> 3)) * W + x) * C64 + (c8 >> 3)) * 64 + (size_t)(((g & 3) * 2 + (g >> 2)) * 8 + (y & 7));
}

#ifdef __HIPCC__
#include "common.hpp"
// the byte of 8 consecutive channels from their packed bf16 values (4 dwords, channel 2k in the low half of dword k): bit = value > 0 as a signed 16-bit compare
__device__ __forceinline__ unsigned char rb_byte_of(const u32x4& v) {
    unsigned b = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned w = v[k];
        b |= ((short)(w & 0xffffu) > 0 ? 1u : 0u) << (2 * k);
        b |= ((short)(w >> 16) > 0 ? 1u : 0u) << (2 * k + 1);
    }
    return (unsigned char)b;
}
#endif
