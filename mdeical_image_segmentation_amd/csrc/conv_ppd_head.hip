// The last 3x3 convolution of the 2-D U-Net with the segmentation head, the loss and their backward in its EPILOGUE (gfx950, bf16) - VERDICT r3 item 1(c).
// Reference arithmetic: `up_conv.3.second` = Conv2d(64, 64, 3, p1) + ReLU (model/unet2d/layers.py:122-126), `final_conv` = Conv2d(64, C, 1) (unet.py:89,127), criterion
// nn.CrossEntropyLoss (C >= 2, int64 labels) / nn.BCEWithLogitsLoss (C == 1, float targets), reduction 'mean' (unet.py:1184-1188, :1208), arg-max of the logits.
//
// Unfused, the step writes the 64-channel feature map u2 (1.07 GB at 32 x 512^2), and mis_head_loss reads it back to emit logits / arg-max / loss and dL/dfeatures
// (another 1.07 GB written): 0.58-0.60 ms per step for a kernel whose only heavy traffic is a tensor the backward pass never needs - the head's weight gradient
// dW = sum dlogit x u2 and the ReLU mask of u2 are both available right where u2 is produced.  Here the convolution's epilogue holds a pixel's 64 features in the
// registers of ONE wave, so per pixel row of a wave tile:
//   * features -> bias (already in the accumulators: C operand of the first MFMAs), ReLU, round to bf16 (the value the unfused path stores);
//   * logits = Wh x features ON THE MATRIX PIPE: the packed bf16 pairs of a lane - channels lg*8 .. lg*8+7 of pixel li, per 32-channel piece - ARE the B fragment of
//     v_mfma_f32_16x16x32_bf16 with K = channel, because the filter rows were permuted for contiguous stores in exactly that order; the A fragment is Wh with its C rows
//     repeated down the 16 rows (row r = class r & 3), so every lane receives the logits of its pixel in its four result registers - no cross-lane traffic at all.
//     Wh is split into bf16 hi + lo parts (two MFMAs per piece): 16 mantissa bits, products exact, fp32 accumulation;
//   * softmax / sigmoid, the loss term, dlogit = grad_scale / (N*H*W[*C]) * (p - onehot), computed redundantly by the four lanes of a pixel;
//   * logits (fp32 NCHW) and arg-max stores from the lanes lg = class / lg = 0;
//   * dL/dfeatures = (feature > 0) * sum_c dlogit_c * Wh[c][k] for the lane's 16 channels, rounded to bf16 and stored WHERE u2 WOULD HAVE GONE;
//   * dW[c][k] += dlogit_c * feature_k, db[c] += dlogit_c, loss += term in per-lane fp32 accumulators that live across the block's tiles; one partial row per block
//     at the end, reduced by mis_head_loss's own fixed-order kernels (head_reduce_partials).
// The convolution itself is conv_ppd_kernel (deep-prefetch column-segment kernel, conv_ppd.hip) with the wave tile turned by 90 degrees: a wave owns 4 pixel rows x 16
// columns x all 64 channels (NF = 4, PF = 4: 48 MFMAs per segment as before, 18 instead of 16 fragment reads) instead of 8 rows x 32 channels - that is what puts a pixel's
// channels into one wave.  Labels of the wave's 64 pixels arrive by ONE 4-byte LDS-DMA per tile (low word of the int64 label / the float target), issued a tile ahead.
#include <stdlib.h>

#include <type_traits>

#include "conv_pp_common.hpp"
#include "dispatch_cfg.hpp"

int head_reduce_partials(const MisHeadDesc* d, int blocks, hipStream_t s);      // head_loss.hip
constexpr int HEAD_PSTRIDE_F = 288;                                             // = HEAD_PSTRIDE of head_loss.hip (floats per block partial)

namespace {
constexpr int PH_PF = 4, PH_NF = 4, PH_BN = 64, PH_TH = 32, PH_TW = 16, PH_HW = 18;
constexpr int PH_HITEMS = (PH_TH + 2) * PH_HW * 4;
constexpr int PH_HINSTR = (PH_HITEMS + 63) / 64;            // 39
constexpr int PH_HBUF = PH_HINSTR * 1024;
constexpr int PH_TAPB = PH_BN * 64;
constexpr int PH_WTILE = 3 * PH_TAPB;
constexpr int PH_HJ = (PH_HINSTR + 3) / 4;
// 3 halo + 3 weight buffers + 2 x bias + 8 x 256 B labels + Wh (fp32 [4][64]) + the final block reduction [8][288] floats (reuses the halo ring)
constexpr int PH_LDS = 3 * PH_HBUF + 3 * PH_WTILE + 2 * PH_BN * 4 + 8 * 256 + 4 * 64 * 4;

struct HeadFusedArgs {
    const float* wh;        // [C][64]
    const float* bh;        // [C]
    const void* labels;     // int64 [N][H][W] (C = 2) / float [N][1][H][W] (C = 1)
    float* logits;          // [N][C][H][W] or nullptr
    unsigned char* argmax;  // [N][H][W] or nullptr
    float* partial;         // [gridDim.x][HEAD_PSTRIDE_F]
    float dscale;           // grad_scale / (N*H*W*(C == 1 ? 1 : 1))   (mean over pixels; BCE: over pixels x channels = pixels for C = 1)
};
}   // namespace

// C = 2 .. 4: cross entropy; C = 1: BCE with logits
template <int C>
__global__ __launch_bounds__(512, 2) void conv_ppd_head_kernel(const ConvArgs a, const HeadFusedArgs hd) {
    using T = __bf16;
    constexpr int PF = PH_PF, NF = PH_NF, BN = PH_BN, HW = PH_HW, HINSTR = PH_HINSTR, HBUF = PH_HBUF, ROWB = HW * 64;
    constexpr int TAPB = PH_TAPB, WTILE = PH_WTILE, HJ = PH_HJ;
    constexpr int HJ0 = 4, HJ1 = 3, HJ2 = HJ - HJ0 - HJ1;
    static_assert(PH_HINSTR == 39 && HJ == 10, "");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hbase = smem;
    char* const wbase = smem + 3 * HBUF;
    char* const bbase = wbase + 3 * WTILE;
    char* const lbase = bbase + 2 * BN * 4;                 // 8 x 256 B: the waves' labels of the current tile
    char* const whl = lbase + 8 * 256;                      // Wh fp32 [4][64] (rows >= C zero)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int li = lane & 15, lg = lane >> 4;

    const int total_tiles = a.nSp;
    const int tstride = (int)gridDim.x;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= total_tiles) return;                        // block-uniform (the launcher sizes the grid <= tiles; partial rows of absent blocks are never read)
    const int nchunks = a.Cin >> 5;
    const int tpi = a.tilesH * a.tilesW;
    auto decode = [&](int t, int& tn, int& th0, int& tw0) {
        tn = t / tpi;
        const int r = t - tn * tpi;
        const int th = r / a.tilesW;
        th0 = th * PH_TH;
        tw0 = (r - th * a.tilesW) * PH_TW;
    };

    const int a_off0 = li * 64 + ((lg ^ ((li >> 1) & 3)) << 4);
    int b_off0[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int px = li + kw;
        b_off0[kw] = (wave * PF * HW + px) * 64 + ((lg ^ ((px >> 1) & 3)) << 4);
    }
    int w_goff0;
    {
        const int slot = (wave & 3) * 64 + lane;
        const int lrow = slot >> 2, pos = slot & 3;
        const int dc16 = pos ^ ((lrow >> 1) & 3);
        const int j = lrow;                                  // one 64-row wave slice: channel order of pp_epilogue_plain<4, *> (piece i = j >> 5)
        const int drow = ((j >> 5) * 32) + ((j & 15) >> 2) * 8 + ((j >> 4) & 1) * 4 + (j & 3);
        w_goff0 = (drow * a.Cin + dc16 * 8) * 2;
    }
    const unsigned img_x = (unsigned)(((long long)a.H * a.W - 1) * a.x0.ld + a.Cin) * 2u;
    const char* const xb = reinterpret_cast<const char*>(a.x0.p);
    const __amdgpu_buffer_rsrc_t rw = pp_make_rsrc(a.w, (unsigned)((long long)9 * a.Cout * a.Cin * 2));

    // per-lane halo offsets precomputed where the registers allow it (C <= 3; with four classes the head's 64 gradient accumulators take their place and the offsets
    // are re-derived at every issue, as conv_ppc_kernel<8, 4> does)
    constexpr bool HPRE = C <= 3;
    int hrel[HPRE ? HJ : 1], hpx[HPRE ? HJ : 1];
    if constexpr (HPRE) {
#pragma unroll
        for (int j = 0; j < HJ; ++j) {
            const int item = (j * 4 + (wave & 3)) * 64 + lane;
            const int p = item >> 2, pos = item & 3;
            const int py = p / HW, px = p - py * HW;
            hrel[j] = ((py * a.W + px) * a.x0.ld + ((pos ^ ((px >> 1) & 3)) << 3)) * 2;
            hpx[j] = item < PH_HITEMS ? px : 0x40000000;
        }
    }
    auto issue_halo = [&](auto jc, int n, int h0, int w0, int c0, char* dst) {
        constexpr int j = decltype(jc)::value;
        const int id = j * 4 + (wave & 3);
        if (id >= HINSTR) return;
        const __amdgpu_buffer_rsrc_t rx = pp_make_rsrc(xb + (size_t)n * a.H * a.W * a.x0.ld * 2, img_x);
        unsigned toff = (unsigned)((((h0 - 1) * a.W + (w0 - 1)) * a.x0.ld + c0) * 2);
        asm volatile("" : "+s"(toff));
        if constexpr (HPRE) {
            const bool ok = (unsigned)(w0 - 1 + hpx[j]) < (unsigned)a.W;
            pp_dma16(rx, ok ? (int)(toff + (unsigned)hrel[j]) : PP_OOB, dst + id * 1024);
        } else {
            int l_;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
            const int item = id * 64 + l_;
            const int p = item >> 2, pos = item & 3;
            const int py = p / HW, px = p - py * HW;
            const unsigned rel = (unsigned)(((py * a.W + px) * a.x0.ld + ((pos ^ ((px >> 1) & 3)) << 3)) * 2);
            const bool ok = item < PH_HITEMS && (unsigned)(w0 - 1 + px) < (unsigned)a.W;
            pp_dma16(rx, ok ? (int)(toff + rel) : PP_OOB, dst + id * 1024);
        }
    };
    const __amdgpu_buffer_rsrc_t rb = pp_make_rsrc(a.bias != nullptr ? (const void*)a.bias : a.w, a.bias != nullptr ? (unsigned)a.Cout * 4u : 0u);
    auto issue_bias = [&](char* dst) {
        if (wave == 0) {
            int l;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (pp_lds_void_t*)dst, 4, l * 4, 0, 0, 0);
        }
    };
    auto issue_weights = [&](int kw, int c0, char* dst) {
        int soff = (int)((((long long)kw * a.Cout) * a.Cin + c0) * 2);
        asm volatile("" : "+s"(soff));
        const int tapstride = 3 * a.Cout * a.Cin * 2;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) pp_dma16(rw, soff + kh * tapstride + w_goff0, dst + kh * TAPB + wave * 1024);
    };
    // every wave: the labels of its 64 pixels (row wave*4 + (lane >> 4), column lane & 15) of tile (n, h0, w0): 4 bytes per lane -> lbase + wave*256
    const unsigned lab_stride = C == 1 ? 4u : 8u;
    const __amdgpu_buffer_rsrc_t rl = pp_make_rsrc(hd.labels, (unsigned)((long long)a.N * a.H * a.W * lab_stride));
    auto issue_labels = [&](int n, int h0, int w0) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const int y = h0 + wave * PF + (l >> 4), x = w0 + (l & 15);
        const int vo = (y < a.H && x < a.W) ? (int)((unsigned)((n * a.H + y) * a.W + x) * lab_stride) : PP_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (pp_lds_void_t*)(lbase + wave * 256), 4, vo, 0, 0, 0);
    };

    // Wh -> LDS (fp32 [C][64], rows >= C zero), read back per tile in the two operand forms
    if (tid < 256) reinterpret_cast<float*>(whl)[tid] = (tid >> 6) < C ? hd.wh[tid] : 0.f;
    float bh[C];
#pragma unroll
    for (int c = 0; c < C; ++c) bh[c] = hd.bh[c];

    int n, h0, w0;
    decode(tile, n, h0, w0);
    f32x4 acc[NF][PF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) acc[f][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
    // per-lane accumulators of the head's parameter gradients and of the loss, over all pixels this lane sees
    pp_f32x2 dwa2[C][8];          // (pairs: the per-row update is one v_pk_fma_f32 per pair and class)
    float dba[C], lsum = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        dba[c] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) dwa2[c][k] = pp_f32x2{0.f, 0.f};
    }
    issue_bias(bbase);
    if (grp == 0) {
        issue_weights(0, 0, wbase);
        issue_weights(1, 0, wbase + WTILE);
    } else {
        pp_static_for<HJ>([&](auto jc) { issue_halo(jc, n, h0, w0, 0, hbase); });
        pp_static_for<HJ>([&](auto jc) { issue_halo(jc, n, h0, w0, 32, hbase + HBUF); });
    }
    issue_labels(n, h0, w0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int wsel = 0, hsel = 0;
    if (grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (; tile < total_tiles; tile += tstride) {
        const bool has_next = tile + tstride < total_tiles;
        int nn = n, nh0 = h0, nw0 = w0;
        if (has_next) decode(tile + tstride, nn, nh0, nw0);
        auto run_chunk = [&](auto firstc, const int chunk) __attribute__((always_inline)) {
            constexpr bool first = decltype(firstc)::value;
            const int c0 = chunk << 5;
            const bool last_chunk = chunk + 1 == nchunks;
            const bool h_same = chunk + 2 < nchunks;
            const bool hnext = h_same || has_next;
            const int hn = h_same ? n : nn, hh0 = h_same ? h0 : nh0, hw0 = h_same ? w0 : nw0, hc0 = h_same ? c0 + 64 : ((chunk + 2 - nchunks) << 5);
            const int hs2 = hsel >= 1 ? hsel - 1 : hsel + 2;
            const uint32_t hb = (uint32_t)(uintptr_t)hbase + hsel * HBUF;
            char* hbn = hbase + hs2 * HBUF;
            pp_static_for<3>([&](auto kc) {
                constexpr int kw = decltype(kc)::value;
                const uint32_t wb = (uint32_t)(uintptr_t)wbase + wsel * WTILE;
                const int ws2 = wsel >= 1 ? wsel - 1 : wsel + 2;
                const bool w2next = kw == 0 || !last_chunk || has_next;
                const int w2kw = (kw + 2) % 3;
                const int w2c0 = kw == 0 ? c0 : (last_chunk ? 0 : c0 + 32);
                // ================= R segment =================
                if (grp == 0) {
                    if (w2next) issue_weights(w2kw, w2c0, wbase + ws2 * WTILE);
                } else if (hnext) {
                    constexpr int J0 = kw == 0 ? 0 : (kw == 1 ? HJ0 : HJ0 + HJ1), NJ = kw == 0 ? HJ0 : (kw == 1 ? HJ1 : HJ2);
                    pp_static_for<NJ>([&](auto jc) { issue_halo(std::integral_constant<int, J0 + decltype(jc)::value>{}, hn, hh0, hw0, hc0, hbn); });
                }
                u32x4 A[3][NF], Brow[PF + 2];
                f32x4 bq[NF];
                if constexpr (first && kw == 0) {
                    int l_;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
                    const uint32_t ba = (uint32_t)(uintptr_t)bbase + ((l_ >> 4) * 8) * 4;
                    pp_static_for<NF>([&](auto fc) {
                        constexpr int f = decltype(fc)::value;
                        const u32x4 r = pp_lds_read128<(f >> 1) * 128 + (f & 1) * 16>(ba);
                        bq[f] = __builtin_bit_cast(f32x4, r);
                    });
                }
                pp_static_for<3>([&](auto hc) {
                    constexpr int kh = decltype(hc)::value;
                    pp_static_for<NF>([&](auto fc) {
                        constexpr int f = decltype(fc)::value;
                        A[kh][f] = pp_lds_read128<kh * TAPB + f * 1024>(wb + a_off0);
                    });
                });
                pp_static_for<PF + 2>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    Brow[r] = pp_lds_read128<r * ROWB>(hb + b_off0[kw]);
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (kw == 2) {
                    if (grp == 1) {
                        if (!hnext) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        else if (wave == 7) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HJ - 1) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HJ) : "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ================= M segment: 3 taps x NF x PF MFMAs =================
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int pf = 0; pf < PF; ++pf) {
                            if constexpr (first && kw == 0) {
                                if (kh == 0) acc[f][pf] = bq[f];
                            }
                            mma_b128<T>(acc[f][pf], A[kh][f], Brow[pf + kh]);
                        }
                __builtin_amdgcn_s_setprio(0);
                if (grp == 0) {
                    if (last_chunk && kw == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (!(first && kw == 0)) {
                        if (w2next) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                wsel = wsel == 2 ? 0 : wsel + 1;
            });
            hsel = hsel == 2 ? 0 : hsel + 1;
        };
        run_chunk(std::true_type{}, 0);
#pragma unroll 1
        for (int chunk = 1; chunk < nchunks; ++chunk) run_chunk(std::false_type{}, chunk);

        // ================= epilogue: features -> head -> loss -> gradients =================
        {
            int lane_;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
            const int eli = lane_ & 15, elg = lane_ >> 4;
            // Wh of this lane's 16 channels (fp32: dL/dfeatures) and the MFMA operand (bf16 hi / lo of Wh[row & 3][piece*32 + elg*8 .. +7], rows >= C zero)
            const uint32_t wl = (uint32_t)(uintptr_t)whl;
            constexpr bool WKR = C <= 3;          // Wh of the lane's channels held in registers for the tile; C = 4: re-read per pixel row (the 64 registers would spill)
            pp_f32x2 wk2[C][8];
            auto load_wk = [&]() {
                u32x4 raw[C][4];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    raw[c][0] = pp_lds_read128<0>(wl + (c * 64 + elg * 8) * 4);
                    raw[c][1] = pp_lds_read128<16>(wl + (c * 64 + elg * 8) * 4);
                    raw[c][2] = pp_lds_read128<128>(wl + (c * 64 + elg * 8) * 4);
                    raw[c][3] = pp_lds_read128<144>(wl + (c * 64 + elg * 8) * 4);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int k = 0; k < 8; ++k) wk2[c][k] = pp_f32x2{__uint_as_float(raw[c][k >> 1][(k & 1) * 2]), __uint_as_float(raw[c][k >> 1][(k & 1) * 2 + 1])};
            };
            if constexpr (WKR) load_wk();
            u32x4 ahi[2], alo[2];
            {
                u32x4 araw[2][2];
                const uint32_t arow = wl + ((eli & 3) * 64 + elg * 8) * 4;          // row r of the operand = class r & 3 (rows >= C are zero in LDS)
                araw[0][0] = pp_lds_read128<0>(arow);
                araw[0][1] = pp_lds_read128<16>(arow);
                araw[1][0] = pp_lds_read128<128>(arow);
                araw[1][1] = pp_lds_read128<144>(arow);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                const bool live = (eli & 3) < C;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq) {
                        const float w0f = live ? __uint_as_float(araw[i][jq >> 1][(jq & 1) * 2]) : 0.f, w1f = live ? __uint_as_float(araw[i][jq >> 1][(jq & 1) * 2 + 1]) : 0.f;
                        const pp_f32x2 w2 = pp_f32x2{w0f, w1f};
                        const uint32_t hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(w2, pp_bf16x2));
                        const pp_f32x2 h2 = pp_f32x2{__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
                        const uint32_t lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(w2 - h2, pp_bf16x2));
                        ahi[i][jq] = hi;
                        alo[i][jq] = lo;
                    }
            }
            // this wave's labels: row pf, column eli
            uint32_t labw[PF];
            {
                const uint32_t la = (uint32_t)(uintptr_t)lbase + wave * 256 + eli * 4;
                asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:64\n\tds_read_b32 %2, %4 offset:128\n\tds_read_b32 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(labw[0]), "=&v"(labw[1]), "=&v"(labw[2]), "=&v"(labw[3])
                             : "v"(la)
                             : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            const size_t img = (size_t)a.H * a.W;
            const __amdgpu_buffer_rsrc_t ry = pp_make_rsrc(reinterpret_cast<const char*>(a.y0) + (size_t)n * img * a.y0_ld * 2, (unsigned)(((img - 1) * a.y0_ld + 64) * 2));
            const int x = w0 + eli;
            const int yv = x < a.W ? (x * a.y0_ld + elg * 8) * 2 : PP_OOB;
            const unsigned yrow = (unsigned)a.W * a.y0_ld * 2;
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) {
                const int y = h0 + wave * PF + pf;                  // wave-uniform
                // features: ReLU, rounded to bf16 - d[i][j] = channels i*32 + elg*8 + 2j, 2j+1 of pixel (y, x)
                u32x4 d[2];
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const pp_f32x2 s2 = pp_f32x2{acc[f][pf][2 * h], acc[f][pf][2 * h + 1]};
                        uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(s2, pp_bf16x2));
                        asm("v_pk_max_i16 %0, %1, 0" : "=v"(pk) : "v"(pk));
                        d[f / 2][(f & 1) * 2 + h] = pk;
                    }
                // logits of pixel (y, x) in lg[0 .. C-1] of EVERY lane of the pixel
                f32x4 lgt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    mma_b128<T>(lgt, ahi[i], d[i]);
                    mma_b128<T>(lgt, alo[i], d[i]);
                }
                float lg_[C], dl[C];
#pragma unroll
                for (int c = 0; c < C; ++c) lg_[c] = lgt[c] + bh[c];
                const bool valid = y < a.H && x < a.W;
                float term;
                int am;
                if constexpr (C == 1) {
                    const float t = __uint_as_float(labw[pf]), xv = lg_[0];
                    const float sg = __frcp_rn(1.f + __expf(-xv));
                    term = fmaxf(xv, 0.f) - xv * t + __logf(1.f + __expf(-fabsf(xv)));
                    dl[0] = hd.dscale * (sg - t);
                    am = xv > 0.f ? 1 : 0;
                } else {
                    const int lab = (int)labw[pf];
                    float m = lg_[0];
                    am = 0;
#pragma unroll
                    for (int c = 1; c < C; ++c)
                        if (lg_[c] > m) {
                            m = lg_[c];
                            am = c;
                        }
                    float se = 0.f, ex[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        ex[c] = __expf(lg_[c] - m);
                        se += ex[c];
                    }
                    float xl = 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        if (lab == c) xl = lg_[c];
                    term = m + __logf(se) - xl;
                    const float inv = __frcp_rn(se);
#pragma unroll
                    for (int c = 0; c < C; ++c) dl[c] = hd.dscale * (ex[c] * inv - (lab == c ? 1.f : 0.f));
                }
                if (!valid) {
                    term = 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c) dl[c] = 0.f;
                }
                if (elg == 0) {
                    lsum += term;
#pragma unroll
                    for (int c = 0; c < C; ++c) dba[c] += dl[c];
                }
                if (valid) {
                    const size_t pix = ((size_t)n * a.H + y) * a.W + x;
                    if (hd.logits != nullptr) {
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            if (elg == c) hd.logits[((size_t)n * C + c) * img + (size_t)y * a.W + x] = lg_[c];
                    }
                    if (hd.argmax != nullptr && elg == 3) hd.argmax[pix] = (unsigned char)am;
                }
                // dW += dl x feature; g = (feature > 0) * sum_c dl_c * Wh[c][k]
                if constexpr (!WKR) load_wk();
                u32x4 go[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq) {
                        const uint32_t dv = d[i][jq];
                        const pp_f32x2 f2 = pp_f32x2{__uint_as_float(dv << 16), __uint_as_float(dv & 0xffff0000u)};
                        const int kp = i * 4 + jq;          // pair index: channels 2 kp, 2 kp + 1 of this lane's 16
                        pp_f32x2 g2 = pp_f32x2{0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            const pp_f32x2 dl2 = pp_f32x2{dl[c], dl[c]};
                            dwa2[c][kp] = __builtin_elementwise_fma(dl2, f2, dwa2[c][kp]);
                            g2 = __builtin_elementwise_fma(dl2, wk2[c][kp], g2);
                        }
                        uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(g2, pp_bf16x2));
                        uint32_t t = dv;          // per 16-bit half: 1 where the feature is positive (it is >= 0), else 0
                        asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_mul_lo_u16 %0, %0, %2" : "=&v"(t) : "v"(t), "v"(pk));
                        go[i][jq] = t;
                    }
                if (y < a.H) {
                    const int srow = __builtin_amdgcn_readfirstlane((int)((unsigned)y * yrow));
                    __builtin_amdgcn_raw_buffer_store_b128(go[0], ry, yv, srow, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(go[1], ry, yv + 64, srow, 0);
                }
            }
        }
        n = nn; h0 = nh0; w0 = nw0;
        if (has_next) issue_labels(n, h0, w0);      // behind this tile's label reads (same wave), ahead of everything the next tile issues (the bias slice - one 64-column
                                                    // tile - stays in LDS for the whole kernel)
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();            // pairs with group 1's last barrier

    // ---- block partial: [C*64 dW][C db][1 loss] (+ zeros for the Dice slots), the row layout of head_kernel's partials ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* const red = reinterpret_cast<float*>(hbase);    // [8][288]: per wave [C*64 dW][C db][loss]
    {
        // sum over the 16 pixel columns (lanes of equal elg) by xor-shuffles within the 16-lane row, then lane eli == 0 of every elg holds the wave's sums
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float v = dwa2[c][k][e];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
                    dwa2[c][k][e] = v;
                }
            float v = dba[c];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
            dba[c] = v;
        }
        float v = lsum;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
        lsum = v;
        if (li == 0) {
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int k = 0; k < 16; ++k) red[wave * 288 + c * 64 + (k >> 3) * 32 + lg * 8 + (k & 7)] = dwa2[c][k >> 1][k & 1];
            if (lg == 0) {
#pragma unroll
                for (int c = 0; c < C; ++c) red[wave * 288 + 256 + c] = dba[c];
                red[wave * 288 + 256 + 4] = lsum;
            }
        }
    }
    __syncthreads();
    float* const out = hd.partial + (size_t)blockIdx.x * HEAD_PSTRIDE_F;
    constexpr int NP = C * 64 + C + 1 + 3 * C;
    for (int i = tid; i < NP; i += 512) {
        float s = 0.f;
        if (i < C * 64) {
#pragma unroll
            for (int w = 0; w < 8; ++w) s += red[w * 288 + i];
        } else if (i < C * 64 + C) {
#pragma unroll
            for (int w = 0; w < 8; ++w) s += red[w * 288 + 256 + (i - C * 64)];
        } else if (i == C * 64 + C) {
#pragma unroll
            for (int w = 0; w < 8; ++w) s += red[w * 288 + 256 + 4];
        }
        out[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
static bool head_fused_ok(const MisConvDesc* d, const MisHeadDesc* h) {
    if (d == nullptr || h == nullptr) return false;
    if (!conv_pp_eligible(d) || d->Cin % 64 != 0 || d->Cout != 64 || d->Cout0 != 64 || d->y0_mode != MIS_OUT_PLAIN || d->y1 != nullptr) return false;
    if (d->mask != nullptr || d->mask_bits != nullptr || d->relu_bits != nullptr || !d->relu || d->bias == nullptr) return false;
    if (((d->H + 31) / 32) * 32 * 100 > d->H * 115) return false;                 // the 32-row tiles must fit (as conv_ppc64_auto)
    const long long img = (long long)d->H * d->W;
    if (((img - 1) * d->y0_ld + 64) * 2 >= (1ll << 32) - 65536) return false;
    if (h->dtype != MIS_BF16 || h->Cfeat != 64 || h->N != d->N || h->npix_per_image != img) return false;
    if (!((h->C >= 2 && h->C <= 4 && h->loss == 0) || (h->C == 1 && h->loss == 1))) return false;
    if (h->dy == nullptr || h->dw == nullptr || h->db == nullptr || h->labels == nullptr || h->loss_out == nullptr || h->workspace == nullptr || h->w == nullptr || h->b == nullptr) return false;
    if (h->dy != d->y0 || h->dy_ld != d->y0_ld) return false;                       // dL/dfeatures goes where the features would have gone
    if (h->phase != 0) return false;
    if ((long long)d->N * img * 8 >= (1ll << 31)) return false;                      // labels within one 32-bit-offset resource
    if (mis_sw(SW_CONV_NOPPD) || mis_sw(SW_CONV_NOPPC) || mis_sw(SW_HEAD_UNFUSED)) return false;
    return true;
}

extern "C" int mis_conv3x3_head_fused_eligible(const MisConvDesc* d, const MisHeadDesc* h) { return head_fused_ok(d, h) ? 1 : 0; }

extern "C" int mis_conv3x3_head_fused(const MisConvDesc* d, const MisHeadDesc* h, void* stream_) {
    (void)hipGetLastError();
    MIS_REQUIRE(head_fused_ok(d, h), MIS_EUNSUPPORTED, "conv3x3_head_fused: not eligible (bf16 2-D 3x3 Cin %% 64 -> 64 with bias + ReLU, C = 2 CE / C = 1 BCE with backward, dy == y0)");
    MIS_REQUIRE(h->workspace_bytes >= mis_head_workspace_bytes(h), MIS_EINVAL, "conv3x3_head_fused: head workspace too small");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ConvArgs a;
    a.tq = nullptr;
    a.N = d->N; a.D = 1; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.Cin0 = d->Cin; a.Cout0 = d->Cout0;
    a.x0 = SrcView{d->x0, d->x0_ld, 1, d->x0_H, d->x0_W};
    a.x1 = SrcView{nullptr, 0, 0, 0, 0};
    a.in_scale = nullptr; a.in_shift = nullptr;
    a.w = d->w; a.bias = d->bias; a.relu = 1; a.mask = nullptr; a.mask_ld = 0; a.mask_bits = nullptr; a.relu_bits = nullptr;
    a.y0 = d->y0; a.y0_ld = d->y0_ld; a.y0_mode = MIS_OUT_PLAIN;
    a.y1 = nullptr; a.y1_ld = 0; a.y1_mode = 0;
    a.tilesH = (d->H + PH_TH - 1) / PH_TH;
    a.tilesW = (d->W + PH_TW - 1) / PH_TW;
    const long long nsp = (long long)d->N * a.tilesH * a.tilesW;
    MIS_REQUIRE(nsp < (1ll << 31), MIS_EUNSUPPORTED, "conv3x3_head_fused: grid too large");
    a.nSp = (int)nsp; a.nCt = 1; a.tilesD = 2; a.order = 0; a.zg = 0;
    a.gn_p = a.gn_q = a.gn_r = nullptr; a.gn_ld = 0; a.gn_relu = 0;
    HeadFusedArgs hd;
    hd.wh = h->w; hd.bh = h->b; hd.labels = h->labels; hd.logits = h->logits; hd.argmax = h->argmax; hd.partial = h->workspace;
    const double total = (double)d->N * (double)d->H * (double)d->W;
    hd.dscale = (float)((double)h->grad_scale / (total * (h->C == 1 ? 1.0 : 1.0)));
    const unsigned grid = (unsigned)(nsp > mis_persist_cus() ? mis_persist_cus() : nsp);
    if (h->C == 2) {
        static std::atomic<unsigned long long> attr_done{0};
        if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&conv_ppd_head_kernel<2>), (size_t)PH_LDS, "conv3x3_head_fused")) return rc;
        hipLaunchKernelGGL((conv_ppd_head_kernel<2>), dim3(grid), dim3(512), (size_t)PH_LDS, stream, a, hd);
    } else if (h->C == 3) {
        static std::atomic<unsigned long long> attr_done{0};
        if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&conv_ppd_head_kernel<3>), (size_t)PH_LDS, "conv3x3_head_fused")) return rc;
        hipLaunchKernelGGL((conv_ppd_head_kernel<3>), dim3(grid), dim3(512), (size_t)PH_LDS, stream, a, hd);
    } else if (h->C == 4) {
        static std::atomic<unsigned long long> attr_done{0};
        if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&conv_ppd_head_kernel<4>), (size_t)PH_LDS, "conv3x3_head_fused")) return rc;
        hipLaunchKernelGGL((conv_ppd_head_kernel<4>), dim3(grid), dim3(512), (size_t)PH_LDS, stream, a, hd);
    } else {
        static std::atomic<unsigned long long> attr_done{0};
        if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&conv_ppd_head_kernel<1>), (size_t)PH_LDS, "conv3x3_head_fused")) return rc;
        hipLaunchKernelGGL((conv_ppd_head_kernel<1>), dim3(grid), dim3(512), (size_t)PH_LDS, stream, a, hd);
    }
    MIS_LAUNCH_CHECK("conv3x3_head_fused");
    return head_reduce_partials(h, (int)grid, stream);
}
