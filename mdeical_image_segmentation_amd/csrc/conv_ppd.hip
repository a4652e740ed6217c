// Deep-prefetch column-segment kernel for the bf16 2-D 3x3 layers with 64-column blocks (gfx950) - the layers with the lowest FLOP per byte of the net
// (64->64 and 128->64 at full resolution, reference model/unet2d/layers.py:122-126: down_conv.0 / up_conv.3 and their dgrads).  Same tile, fragment mapping, LDS images,
// segment structure (one filter COLUMN of one 32-channel chunk = 48 MFMAs between two barriers, two wave groups staggered by one barrier) and epilogue as
// conv_ppc_kernel<8, 2> (conv_pp.hip).  What changes is how the operands get to LDS.
//
// Why: ablation builds of conv_ppc_kernel<8, 2> on 64->64 at 512^2 (scripts/ppt_ablate.sh) - 0.744 ms complete, 0.581 without MFMAs, 0.539 without DMAs, 0.575 without
// the epilogue: the three parts ADD instead of overlapping, and the memory path alone moves 3.7 TB/s where a copy moves 5.4.  The cause is the in-order vmcnt counter of a
// wave: there a wave interleaves halo DMAs (HBM, wanted one chunk later), weight DMAs (L2, wanted one segment later) and the tile's output stores, so every wait for a
// weight tile also waits for everything older - the halo prefetch can never be more than one segment deep (<= 39 KB in flight per CU), and the first weight wait of a
// tile drains the previous tile's stores.  The 1x1 kernel (gemm1_pp.hip), whose only stream is two stages deep (80 KB per CU), runs at the copy rate.
//
// Here each wave group owns ONE stream, so each wave's vmcnt FIFO is homogeneous:
//   * group 1 (waves 4-7) issues every halo DMA, TWO chunks ahead, into a ring of three halo buffers, and waits once per chunk (vmcnt = the instructions it issued during
//     that chunk): 39-78 KB of input in flight per CU at all times, and the previous tile's stores have a whole chunk to drain before anything waits behind them;
//   * group 0 (waves 0-3) issues every weight DMA, two SEGMENTS ahead, into a ring of three 3-tap weight buffers; the last segment of a tile waits for everything (the
//     first two segments of the next tile are then in LDS before the epilogue's stores are issued), so the first wait behind the stores comes two segments later.
// The ReLU bits of a masked dgrad are staged through LDS (two 4-byte DMAs per lane at the start of the tile): a register load would sit in the compiler's vmcnt model.
// LDS: 3 x 39 KiB halo + 3 x 12 KiB weights + bias + bits = 157.5 KiB.
#include <stdlib.h>

#include <type_traits>

#include "conv_pp_common.hpp"
#include "dispatch_cfg.hpp"

namespace {
constexpr int PD_PF = 8, PD_NF = 2, PD_BN = 64, PD_TH = 32, PD_TW = 16, PD_HW = 18;
constexpr int PD_HITEMS = (PD_TH + 2) * PD_HW * 4;          // 16-byte items of one chunk's halo image (64-byte pixels)
constexpr int PD_HINSTR = (PD_HITEMS + 63) / 64;            // 39
constexpr int PD_HBUF = PD_HINSTR * 1024;
constexpr int PD_TAPB = PD_BN * 64;                         // one tap's weight tile: 4 DMA instructions (waves 0-3)
constexpr int PD_WTILE = 3 * PD_TAPB;
constexpr int PD_HJ = (PD_HINSTR + 3) / 4;                  // halo instructions per group-1 wave and chunk: 10 (the last one only for ids < 39)
constexpr int PD_LDS = 3 * PD_HBUF + 3 * PD_WTILE + 2 * PD_BN * 4 + 8 * 512;
}   // namespace

template <int EM>
__global__ __launch_bounds__(512, 2) void conv_ppd_kernel(const ConvArgs a) {
    using T = __bf16;
    constexpr int PF = PD_PF, NF = PD_NF, WAVE_N = NF * 16, BN = PD_BN, HW = PD_HW, HINSTR = PD_HINSTR, HBUF = PD_HBUF, ROWB = HW * 64;
    constexpr int TAPB = PD_TAPB, WTILE = PD_WTILE, HJ = PD_HJ;
    constexpr int HJ0 = 4, HJ1 = 3, HJ2 = HJ - HJ0 - HJ1;   // halo instructions of a chunk by the segment that issues them
    static_assert(PD_HINSTR == 39 && HJ == 10, "waves 4-6 issue HJ instructions per chunk, wave 7 HJ - 1 (its id 39 is past the image)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hbase = smem;                               // 3 x HBUF
    char* const wbase = smem + 3 * HBUF;                    // 3 x WTILE
    char* const bbase = wbase + 3 * WTILE;                  // 2 x BN floats
    char* const mlds = bbase + 2 * BN * 4;                  // 8 x 512 B: the waves' ReLU bits of the current tile (EM == PP_EM_BITS)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    const int li = lane & 15, lg = lane >> 4;

    const int total_tiles = a.nSp * a.nCt;
    const int tstride = (int)gridDim.x;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= total_tiles) return;                        // block-uniform
    // dynamic tile queue (conv_pp_common.hpp TileQ; as in conv_ppc_kernel): the first tile is static, the others are drawn by wave 0's lane 0 one tile ahead.  Wave 0 is a
    // group-0 wave: its vmcnt queue holds weight DMAs only, every wait from the second segment on is vmcnt(3) or less, so a draw issued at the top of a tile has returned
    // when chunk 0 is through.
    const TileQ tq = tq_init(a.tq, total_tiles, tstride, (int)blockIdx.x);
    const bool dyn = tq.ctr != nullptr;
    const bool drawer = dyn && tid == 0;
    unsigned tk = 1u;
    if (drawer) tq_draw(tk, tq.ctr);
    const int nchunks = a.Cin >> 5;                         // >= 2 (the launcher checks Cin % 64 == 0)
    const int tpi = a.tilesH * a.tilesW;
    auto decode = [&](int t, int& tn, int& th0, int& tw0, int& tcol) {      // tile order: conv_ppc_kernel
        int ct, sp;
        if (a.tilesD == 2) {
            sp = t / a.nCt;
            ct = t - sp * a.nCt;
        } else {
            ct = t / a.nSp;
            sp = t - ct * a.nSp;
        }
        tn = sp / tpi;
        const int r = sp - tn * tpi;
        const int th = r / a.tilesW;
        th0 = th * PD_TH;
        tw0 = (r - th * a.tilesW) * PD_TW;
        tcol = ct * BN;
    };

    const int a_off0 = (wn * WAVE_N + li) * 64 + ((lg ^ ((li >> 1) & 3)) << 4);
    int b_off0[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int px = li + kw;
        b_off0[kw] = (wm * PF * HW + px) * 64 + ((lg ^ ((px >> 1) & 3)) << 4);
    }
    int w_goff0;       // group 0: this wave's instruction of a tap tile = LDS rows 16*wave .. +15 (channel order of pp_epilogue_plain, see conv_ppc_kernel)
    {
        const int slot = (wave & 3) * 64 + lane;
        const int lrow = slot >> 2, pos = slot & 3;
        const int dc16 = pos ^ ((lrow >> 1) & 3);
        const int dwv = lrow / WAVE_N, j = lrow % WAVE_N;
        const int drow = dwv * WAVE_N + ((j >> 5) * 32) + ((j & 15) >> 2) * 8 + ((j >> 4) & 1) * 4 + (j & 3);
        w_goff0 = (drow * a.Cin + dc16 * 8) * 2;
    }
    const unsigned img_x = (unsigned)(((long long)a.H * a.W - 1) * a.x0.ld + a.Cin) * 2u;
    const char* const xb = reinterpret_cast<const char*>(a.x0.p);
    const __amdgpu_buffer_rsrc_t rw = pp_make_rsrc(a.w, (unsigned)((long long)9 * a.Cout * a.Cin * 2));

    // halo DMA j of a group-1 wave (instruction id = j*4 + wave - 4).  Unlike conv_ppc_kernel<8, 4> (256 VGPRs: held offsets spill) this kernel has registers to spare, and the
    // issue sits in group 1's R segments, which bound every other barrier interval: the per-lane part is precomputed - hrel = byte offset of the lane's item from the halo
    // origin, hpx = its halo column (a huge value for items past the image: never in range) - so that one issue is 4 vector instructions.  Rows above / below the image need
    // no test: the halo origin of a top tile is a negative offset, i.e. past num_records as unsigned, and so is everything below the last row (per-image resource).
    int hrel[HJ], hpx[HJ];
#pragma unroll
    for (int j = 0; j < HJ; ++j) {
        const int item = (j * 4 + (wave & 3)) * 64 + lane;
        const int p = item >> 2, pos = item & 3;
        const int py = p / HW, px = p - py * HW;
        hrel[j] = ((py * a.W + px) * a.x0.ld + ((pos ^ ((px >> 1) & 3)) << 3)) * 2;
        hpx[j] = item < PD_HITEMS ? px : 0x40000000;
    }
    auto issue_halo = [&](auto jc, int n, int h0, int w0, int c0, char* dst) {
        constexpr int j = decltype(jc)::value;
        const int id = j * 4 + (wave & 3);
        if (id >= HINSTR) return;                           // wave-uniform (j = 9, wave 7)
        const __amdgpu_buffer_rsrc_t rx = pp_make_rsrc(xb + (size_t)n * a.H * a.W * a.x0.ld * 2, img_x);
        unsigned toff = (unsigned)((((h0 - 1) * a.W + (w0 - 1)) * a.x0.ld + c0) * 2);
        asm volatile("" : "+s"(toff));
        const bool ok = (unsigned)(w0 - 1 + hpx[j]) < (unsigned)a.W;
        pp_dma16(rx, ok ? (int)(toff + (unsigned)hrel[j]) : PP_OOB, dst + id * 1024);
    };
    const __amdgpu_buffer_rsrc_t rb = pp_make_rsrc(a.bias != nullptr ? (const void*)a.bias : a.w, a.bias != nullptr ? (unsigned)a.Cout * 4u : 0u);
    auto issue_bias = [&](int col, char* dst) {            // wave 0: 64 floats
        if (wave == 0) {
            int l;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (pp_lds_void_t*)dst, 4, (col + l) * 4, 0, 0, 0);
        }
    };
    // group 0: the three tap tiles (kh = 0..2) of filter column kw, column tile col, channels c0..c0+31 - one instruction per tap per wave
    auto issue_weights = [&](int kw, int col, int c0, char* dst) {
        int soff = (int)((((long long)kw * a.Cout + col) * a.Cin + c0) * 2);
        asm volatile("" : "+s"(soff));
        const int tapstride = 3 * a.Cout * a.Cin * 2;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) pp_dma16(rw, soff + kh * tapstride + w_goff0, dst + kh * TAPB + wave * 1024);
    };
    // every wave: its 8 bytes per lane of ReLU bits of tile (n, h0, w0, col) (pp_bits_voff) as two 4-byte DMAs -> mlds + wave*512 (+256)
    auto issue_bits = [&](int n, int h0, int w0, int col) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const int bo = pp_bits_voff<NF, PF>(a, n, h0, w0, col, wm, wn, l & 15, l >> 4);
        const __amdgpu_buffer_rsrc_t rbm = pp_make_rsrc(a.mask_bits, (unsigned)rb_bytes(a.N, a.H, a.W, a.Cout));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rbm, (pp_lds_void_t*)(mlds + wave * 512), 4, bo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rbm, (pp_lds_void_t*)(mlds + wave * 512 + 256), 4, bo + 4, 0, 0, 0);      // (an instruction offset would move the LDS address too)
    };

    int n, h0, w0, ncol0;
    decode(tile, n, h0, w0, ncol0);
    f32x4 acc[NF][PF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) acc[f][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue_bias(ncol0, bbase);
    if (grp == 0) {
        issue_weights(0, ncol0, 0, wbase);                                   // segments 0 and 1
        issue_weights(1, ncol0, 0, wbase + WTILE);
    } else {
        pp_static_for<HJ>([&](auto jc) { issue_halo(jc, n, h0, w0, 0, hbase); });            // chunks 0 and 1 of the first tile
        pp_static_for<HJ>([&](auto jc) { issue_halo(jc, n, h0, w0, 32, hbase + HBUF); });
    }
    if constexpr (EM == PP_EM_BITS) issue_bits(n, h0, w0, ncol0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t mailbox = (uint32_t)(uintptr_t)(smem + PD_LDS);
    if (drawer) tq_post(tq, tk, mailbox);
    __syncthreads();
    int nxt = tile + tstride;
    if (dyn) nxt = tq_tile(tq, tq_take(mailbox), tstride);

    int wsel = 0, hsel = 0, bsel = 0;                      // ring positions of the segment / chunk being computed
    if (grp == 1) __builtin_amdgcn_s_barrier();            // the stagger
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (;;) {
        const bool has_next = (unsigned)nxt < (unsigned)total_tiles;
        int nn = n, nh0 = h0, nw0 = w0, ncolN = ncol0;
        if (has_next) decode(nxt, nn, nh0, nw0, ncolN);
        tk = 1u;
        if (drawer && has_next) tq_draw(tk, tq.ctr);          // the tile after `nxt`: posted after chunk 0, taken at the end of the tile (nchunks >= 2: barriers in between)
        auto run_chunk = [&](auto firstc, const int chunk) __attribute__((always_inline)) {
            constexpr bool first = decltype(firstc)::value;
            const int c0 = chunk << 5;
            const bool last_chunk = chunk + 1 == nchunks;
            // group 1: the halo of chunk + 2 (this tile's, or chunk + 2 - nchunks of the next tile) -> ring slot hsel + 2
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
                // group 0: the weights of the segment after the next one (kw + 2: this chunk's last column for kw = 0, else column kw - 1 of the next chunk / tile)
                const bool w2next = kw == 0 || !last_chunk || has_next;
                const int w2kw = (kw + 2) % 3;
                const int w2col = (kw == 0 || !last_chunk) ? ncol0 : ncolN;
                const int w2c0 = kw == 0 ? c0 : (last_chunk ? 0 : c0 + 32);
                // ================= R segment =================
#ifndef PPT_NO_DMA          // (PPT_NO_*: timing ablations of a diagnostic build, scripts/ppt_ablate.sh - results are garbage, never shipped)
                if (grp == 0) {
                    if (w2next) issue_weights(w2kw, w2col, w2c0, wbase + ws2 * WTILE);
                } else if (hnext) {
                    constexpr int J0 = kw == 0 ? 0 : (kw == 1 ? HJ0 : HJ0 + HJ1), NJ = kw == 0 ? HJ0 : (kw == 1 ? HJ1 : HJ2);
                    pp_static_for<NJ>([&](auto jc) { issue_halo(std::integral_constant<int, J0 + decltype(jc)::value>{}, hn, hh0, hw0, hc0, hbn); });
                }
#endif
                u32x4 A[3][NF], Brow[PF + 2];
                f32x4 bq[NF];          // first && kw == 0: the tile's bias, the C operand of the first filter row's MFMAs (conv_ppc_kernel)
                if constexpr (first && kw == 0) {
                    int l_;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
                    const uint32_t ba = (uint32_t)(uintptr_t)bbase + bsel * (BN * 4) + (wn * WAVE_N + (l_ >> 4) * 8) * 4;
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
                    // group 1, once per chunk: the NEXT chunk's halo (issued during the previous chunk) has landed when at most this chunk's instructions are in flight
                    // (wave 7 issues HJ - 1: its last id is past the image)
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
#ifndef PPT_NO_MFMA
                            mma_b128<T>(acc[f][pf], A[kh][f], Brow[pf + kh]);
#endif
                        }
                __builtin_amdgcn_s_setprio(0);
                if (grp == 0) {
                    // group 0: the NEXT segment's weights (issued one segment ago) have landed when only this segment's three instructions are in flight.  The last
                    // segment of a tile waits for all of them, the first one for none: whatever it needs was in LDS before the epilogue's stores were issued
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
        if (drawer && has_next) tq_post(tq, tk, mailbox);
        if (has_next) issue_bias(ncolN, bbase + (bsel ^ 1) * (BN * 4));          // (last read in the previous tile's chunk 0; lands under the rest of this tile's K loop)
#pragma unroll 1
        for (int chunk = 1; chunk < nchunks; ++chunk) run_chunk(std::false_type{}, chunk);
        u32x4 mbits = u32x4{0u, 0u, 0u, 0u};
        if constexpr (EM == PP_EM_BITS) {
            int l_;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
            const uint32_t ma = (uint32_t)(uintptr_t)mlds + wave * 512 + l_ * 4;
            uint32_t m0, m1;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:256\n\ts_waitcnt lgkmcnt(0)" : "=&v"(m0), "=&v"(m1) : "v"(ma) : "memory");
            mbits[0] = m0;
            mbits[1] = m1;
            __builtin_amdgcn_sched_barrier(0);
        }
#ifndef PPT_NO_EPI
        pp_epilogue_plain<NF, PF, EM, true, true>(a, acc, 0u, n, h0, w0, ncol0, wm, wn, mbits);
#endif
        n = nn; h0 = nh0; w0 = nw0; ncol0 = ncolN;
        bsel ^= 1;
        // the next tile's bits: behind this tile's reads of mlds (same wave), ahead of everything that tile issues
        if constexpr (EM == PP_EM_BITS) {
            if (has_next) issue_bits(n, h0, w0, ncol0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!has_next) break;
        tile = nxt;
        nxt = dyn ? tq_tile(tq, tq_take(mailbox), tstride) : tile + tstride;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();            // pairs with group 1's last barrier
#ifdef PPT_NO_EPI
    {
        f32x4 s4 = acc[0][0];
        pp_static_for<NF>([&](auto fc) { pp_static_for<PF>([&](auto pc) { s4 += acc[decltype(fc)::value][decltype(pc)::value]; }); });
        if (s4[0] == 123.456f) *reinterpret_cast<f32x4*>(a.y0) = s4;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// conv_ppc_choice(d) == 2 descriptors (64-column blocks of the column-segment kernel, conv_pp.hip) run here unless MIS_CONV_NOPPD is set
template <int EM> static int ppd_launch(const MisConvDesc* d, hipStream_t stream) {
    constexpr int BN = PD_BN;
    ConvArgs a;
    a.tq = nullptr;
    a.N = d->N; a.D = 1; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.Cin0 = d->Cin; a.Cout0 = d->Cout0;
    a.x0 = SrcView{d->x0, d->x0_ld, 1, d->x0_H, d->x0_W};
    a.x1 = SrcView{nullptr, 0, 0, 0, 0};
    a.in_scale = nullptr; a.in_shift = nullptr;
    a.w = d->w; a.bias = d->bias; a.relu = d->relu; a.mask = d->mask; a.mask_ld = d->mask_ld; a.mask_bits = reinterpret_cast<const unsigned char*>(d->mask_bits);
    a.y0 = d->y0; a.y0_ld = d->y0_ld; a.y0_mode = d->y0_mode;
    a.y1 = d->y1; a.y1_ld = d->y1_ld; a.y1_mode = d->y1_mode;
    a.relu_bits = reinterpret_cast<unsigned char*>(d->relu_bits);      // written from the epilogue (pp_epilogue_plain)
    g_conv_bits_fused = d->relu_bits != nullptr;
    a.tilesH = (d->H + PD_TH - 1) / PD_TH;
    a.tilesW = (d->W + PD_TW - 1) / PD_TW;
    const long long nsp = (long long)d->N * a.tilesH * a.tilesW;
    a.nCt = d->Cout / BN;
    a.tilesD = (256 % a.nCt == 0 && !mis_sw(SW_CONV_PPC_COLMAJOR)) ? 2 : 1;
    MIS_REQUIRE(d->Cin % 64 == 0, MIS_EUNSUPPORTED, "conv_igemm(ppd): Cin % 64");
    MIS_REQUIRE(nsp * a.nCt < (1ll << 31), MIS_EUNSUPPORTED, "conv_igemm(ppd): grid too large");
    a.nSp = (int)nsp;
    a.order = 0; a.zg = 0;
    static std::atomic<unsigned long long> attr_done{0};
    if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&conv_ppd_kernel<EM>), (size_t)PD_LDS + 16, "conv_igemm(ppd)")) return rc;
    const long long total = nsp * a.nCt;
    a.tq = mis_tile_queue(stream);
    hipLaunchKernelGGL((conv_ppd_kernel<EM>), dim3((unsigned)(total > mis_persist_cus() ? mis_persist_cus() : total)), dim3(512), (size_t)PD_LDS + 16, stream, a);
    MIS_LAUNCH_CHECK("conv_igemm(ppd)");
    return MIS_OK;
}

int launch_conv_ppd(const MisConvDesc* d, hipStream_t stream, const char** tag) {
    if (d->mask_bits != nullptr) {
        *tag = "k3.2d.ppd8.bits";
        return ppd_launch<PP_EM_BITS>(d, stream);
    }
    if (d->mask != nullptr) {
        *tag = "k3.2d.ppd8.mask";
        return ppd_launch<PP_EM_MASK>(d, stream);
    }
    *tag = "k3.2d.ppd8";
    return ppd_launch<PP_EM_NONE>(d, stream);
}
