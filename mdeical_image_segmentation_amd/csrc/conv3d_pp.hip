// Column-segment ping-pong implicit GEMM for the bf16 3x3x3 layers (gfx950): forward and dgrad of nn.Conv3d(k3, p1, bias=False) of the 'gcr' SingleConv
// (reference model/unet3d/buildingblocks.py:64-66,87-92), on the structure of conv_ppc_kernel (conv_pp.hip) - round 2 left these layers on the lock-step
// conv_igemm_kernel (36-49 % MFMA pipe busy), because the GroupNorm affine was applied while the operand was staged through registers and an LDS-DMA has no ALU.
// Here the operand is the NORMALISED tensor, written once per SingleConv by mis_gn_apply (groupnorm.hip; the same tensor feeds the weight gradient), so the
// kernel sees a plain single-source bf16 convolution and nothing is staged through registers.
//
// A 3x3x3 filter = three 3x3 filters on the depth planes z-1, z, z+1: the output tile is a 2-D tile (4*PF rows x 16 columns) of ONE depth plane (n, z) x BN
// channels, and the K loop walks (dz, 32-channel chunk) pairs - per pair one halo image of plane z + dz - 1 (out-of-range planes: every lane's DMA offset is
// past the buffer, which reads as zero) and three column segments of three taps (tap index dz*9 + kh*3 + kw of the packed [27][Cout][Cin] weights).
// NDHWC makes the planes of a batch one contiguous sequence of H x W images, so the epilogue of the 2-D kernel serves unchanged with the plane index as its image.
//
// Tiles: PF = 5 (20 rows) divides the grids of the 160^3 configuration at every level (160 / 80 / 40 / 20), PF = 8 (32 rows) those of 128^3; the launcher
// takes whichever wastes fewer rows.  NF = 4: 128-column blocks (Cout % 128 == 0), NF = 6: 192-column blocks (wave tile 80 px x 96 ch; the 64 -> 192 dgrad),
// NF = 2: 64-column blocks (wave tile 16*PF px x 32 ch; PF = 10 = 40-row tiles as well: these layers carry 58 % of the net's FLOPs and are bound by their
// fragment reads - 0.30 ds_read_b128 per MFMA at PF 10 against 0.33 at PF 8 and 0.43 at PF 5).
// Plane walk: the tiles of ZG consecutive depth planes are interleaved (plane fastest), so the ~32 tiles an XCD runs at a time cover ZG + 2 input planes for ZG
// output planes out of its L2 instead of 3 for 1.
#include <stdlib.h>

#include "conv_pp_common.hpp"
#include "dispatch_cfg.hpp"

template <int PF, int NF, int EM>
__global__ __launch_bounds__(512, 2) void conv3d_ppc_kernel(const ConvArgs a) {
    using T = __bf16;
    constexpr int WAVE_N = NF * 16, BN = 2 * WAVE_N, NV = 4 * NF;
    constexpr int WINSTR = BN / 16;                                      // weight DMA instructions per tap (16 rows of 64 B each): 4 / 8 / 12 for 64 / 128 / 192 columns
    constexpr int WPW = (WINSTR + 7) / 8;                                // ... per wave (the second round, NF = 6, only reaches waves 0-3)
    constexpr int TH = 4 * PF, TW = 16, HH = TH + 2, HW = 18, HP = HH * HW;
    constexpr int HITEMS = HP * 4, HINSTR = (HITEMS + 63) / 64, HBUF = HINSTR * 1024, ROWB = HW * 64;
    constexpr int HJ = (HINSTR + 7) / 8;             // halo instructions per wave per chunk
    constexpr int HJ0 = HJ < 3 ? HJ : 3, HJ1 = HJ - HJ0;
    static_assert(HJ1 <= 3, "halo instructions of a chunk must fit the R segments of columns 0 and 1");
    constexpr int TAPB = BN * 64;                    // one tap's weight tile
    constexpr int WTILE = 3 * TAPB;                  // one column's weight tiles

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hbase = smem;                        // 2 x HBUF
    char* const wbase = smem + 2 * HBUF;             // 2 x WTILE
    char* const bbase = wbase + 2 * WTILE;           // 2 x BN floats
    char* const pbase = bbase + 2 * BN * 4;          // PP_EM_GN: 2 x [3][BN] floats - p, q, r of the current / the next tile's (sample, column tile)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    const int li = lane & 15, lg = lane >> 4;

    const int total_tiles = a.nSp * a.nCt;
    const int tstride = (int)gridDim.x;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= total_tiles) return;                 // block-uniform
    // dynamic tile queue (conv_pp_common.hpp TileQ; as in conv_ppc_kernel): first tile static, the others drawn by wave 0's lane 0 one tile ahead; wave 0 (group 0) meets a
    // vmcnt(0) at the end of the kw = 2 segment of every (dz, chunk) pair, so a draw issued at the top of a tile has returned when pair 0 is through
    // (not in the 256-VGPR instantiations, PF 8 x 128 columns: one more live register there and hipcc spills - into scratch, whose reloads wait on vmcnt)
    constexpr bool TQ = NF * PF < 32;
    constexpr bool STREAM = NF * PF >= 40;          // PF 10 x 128 columns: pixel-row fragments streamed through the M segment (below)
    const TileQ tq = tq_init(TQ ? a.tq : nullptr, total_tiles, tstride, (int)blockIdx.x);
    const bool dyn = TQ && tq.ctr != nullptr;
    const bool drawer = dyn && tid == 0;
    unsigned tk = 1u;
    if (drawer) tq_draw(tk, tq.ctr);
    const int nchunks = a.Cin >> 5;
    const int nvc = 3 * nchunks;                     // (dz, chunk) pairs per tile
    const int tpi = a.tilesH * a.tilesW;
    // tile -> (plane = n*D + z, z, h0, w0, first column).  Within a volume the planes are walked in groups of a.zg: (group, tile position, plane in group).
    auto decode = [&](int t, int& tpl, int& tz, int& th0, int& tw0, int& tcol) {
        int ct, sp;
        if (a.order == 2) {
            sp = t / a.nCt;
            ct = t - sp * a.nCt;
        } else {
            ct = t / a.nSp;
            sp = t - ct * a.nSp;
        }
        const int per_vol = a.D * tpi;
        const int n = sp / per_vol;
        const int r = sp - n * per_vol;
        const int zgrp = r / (a.zg * tpi);
        const int r2 = r - zgrp * (a.zg * tpi);
        int gsz = a.D - zgrp * a.zg;
        if (gsz > a.zg) gsz = a.zg;
        const int pos = r2 / gsz;
        tz = zgrp * a.zg + (r2 - pos * gsz);
        tpl = n * a.D + tz;
        const int th = pos / a.tilesW;
        th0 = th * TH;
        tw0 = (pos - th * a.tilesW) * TW;
        tcol = ct * BN;
    };

    // LDS swizzle for 64-byte rows (conv_pp.hip): chunk position ^= (column >> 1) & 3 (pixels) / (row >> 1) & 3 (weights) - conflict-free for ds_read_b128
    const int a_off0 = (wn * WAVE_N + li) * 64 + ((lg ^ ((li >> 1) & 3)) << 4);
    int b_off0[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int px = li + kw;
        b_off0[kw] = (wm * PF * HW + px) * 64 + ((lg ^ ((px >> 1) & 3)) << 4);
    }
    int w_goff0[WPW];  // this wave's instruction(s) of a tap tile: LDS rows 16*(wave + 8r) .. +15
#pragma unroll
    for (int r = 0; r < WPW; ++r) {
        const int slot = (wave + 8 * r) * 64 + lane;
        const int lrow = slot >> 2, pos = slot & 3;
        const int dc16 = pos ^ ((lrow >> 1) & 3);
        const int dwv = lrow / WAVE_N, j = lrow % WAVE_N;
        const int drow = dwv * WAVE_N + ((j >> 5) * 32) + ((j & 15) >> 2) * 8 + ((j >> 4) & 1) * 4 + (j & 3);       // channel order of pp_epilogue_plain (see conv_ppc_kernel)
        w_goff0[r] = (drow * a.Cin + dc16 * 8) * 2;
    }
    const unsigned img_x = (unsigned)(((long long)a.H * a.W - 1) * a.x0.ld + a.Cin) * 2u;      // bytes of ONE depth plane of the input view
    const size_t plane_b = (size_t)a.H * a.W * a.x0.ld * 2;
    const char* const xb = reinterpret_cast<const char*>(a.x0.p);
    const __amdgpu_buffer_rsrc_t rw = pp_make_rsrc(a.w, (unsigned)((long long)27 * a.Cout * a.Cin * 2));

    // Every instruction of an R segment costs the wave issue slots it shares with its partner's MFMA cluster: where registers allow (the instantiations below 200
    // VGPRs) the per-lane source offset and halo coordinates of the wave's halo instructions are computed ONCE (conv_pp_kernel's HPRE), ~25 VALU fewer per issue.
    constexpr bool HPRE = (NF * PF <= 20);
    unsigned h_rel[HPRE ? HJ : 1];
    int h_coord[HPRE ? HJ : 1];                   // halo row | halo column << 8; -1: no item
    if constexpr (HPRE) {
#pragma unroll
        for (int j = 0; j < HJ; ++j) {
            const int id = j * 8 + wave;
            const int item = id * 64 + lane;
            const int p = item >> 2, pos = item & 3;
            const int py = p / HW, px = p - py * HW;
            h_rel[j] = (unsigned)(((py * a.W + px) * a.x0.ld + ((pos ^ ((px >> 1) & 3)) << 3)) * 2);
            h_coord[j] = (id < HINSTR && item < HITEMS) ? (py | (px << 8)) : -1;
        }
    }
    // halo DMA j of this wave (instruction id = j*8 + wave) from plane `spl` (valid = the plane exists; otherwise the image is zero-filled)
    auto issue_halo = [&](auto jc, int spl, bool valid, int h0, int w0, int c0, char* dst) {
        constexpr int j = decltype(jc)::value;
        const int id = j * 8 + wave;
        if (id >= HINSTR) return;                     // wave-uniform
        const __amdgpu_buffer_rsrc_t rx = pp_make_rsrc(xb + (size_t)(valid ? spl : 0) * plane_b, img_x);
        unsigned toff = (unsigned)((((h0 - 1) * a.W + (w0 - 1)) * a.x0.ld + c0) * 2);
        if constexpr (HPRE) {
            const bool interior = h0 >= 1 && h0 + TH + 1 <= a.H && w0 >= 1 && w0 + TW + 1 <= a.W;        // block-uniform
            bool ok = valid && h_coord[j] >= 0;
            if (!interior) {
                const int py = h_coord[j] & 0xff, px = h_coord[j] >> 8;
                ok = ok && (unsigned)(h0 - 1 + py) < (unsigned)a.H && (unsigned)(w0 - 1 + px) < (unsigned)a.W;
            }
            pp_dma16(rx, ok ? (int)(toff + h_rel[j]) : PP_OOB, dst + id * 1024);
            return;
        }
        asm volatile("" : "+s"(toff));
        int item;
        if constexpr (STREAM) {          // (256 VGPRs: id * 64 + lane per instruction, kept across the tile loop, was SPILLED - the lane index is re-derived instead, as in conv_ppc_kernel)
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(item));
            item += id * 64;
        } else {
            item = id * 64 + lane;
            asm volatile("" : "+v"(item));
        }
        const int p = item >> 2, pos = item & 3;
        const int py = p / HW, px = p - py * HW;
        const unsigned rel = (unsigned)(((py * a.W + px) * a.x0.ld + ((pos ^ ((px >> 1) & 3)) << 3)) * 2);
        const bool ok = valid && item < HITEMS && (unsigned)(h0 - 1 + py) < (unsigned)a.H && (unsigned)(w0 - 1 + px) < (unsigned)a.W;
        pp_dma16(rx, ok ? (int)(toff + rel) : PP_OOB, dst + id * 1024);
    };
    const __amdgpu_buffer_rsrc_t rb = pp_make_rsrc(a.bias != nullptr ? (const void*)a.bias : a.w, a.bias != nullptr ? (unsigned)a.Cout * 4u : 0u);
    auto issue_bias = [&](int col, char* dst) {
        if (wave < BN / 64) {                         // (BN is a multiple of 64 in every instantiation)
            int l;                                    // lane index re-derived rather than held (or spilled) across the tile loop, as in conv_ppc_kernel
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (pp_lds_void_t*)(dst + wave * 256), 4, (col + wave * 64 + l) * 4, 0, 0, 0);
        }
    };
    // PP_EM_GN: the GroupNorm-backward coefficients of sample n, columns col .. col + BN - 1 (MisConvDesc.gn_p / gn_q / gn_r, [N][gn_ld] floats) -> dst [3][BN]
    auto issue_pqr = [&](int n, int col, char* dst) {
        if constexpr (EM == PP_EM_GN) {
            if (wave < BN / 64) {
                int l;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
                const unsigned bytes = (unsigned)a.N * (unsigned)a.gn_ld * 4u;
                const int c = col + wave * 64 + l;
                const int voff = c < a.gn_ld ? (n * a.gn_ld + c) * 4 : PP_OOB;          // padding columns read as zero (their outputs are dropped anyway)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(pp_make_rsrc(a.gn_p, bytes), (pp_lds_void_t*)(dst + wave * 256), 4, voff, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(pp_make_rsrc(a.gn_q, bytes), (pp_lds_void_t*)(dst + BN * 4 + wave * 256), 4, voff, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(pp_make_rsrc(a.gn_r, bytes), (pp_lds_void_t*)(dst + 2 * BN * 4 + wave * 256), 4, voff, 0, 0, 0);
            }
        }
    };
    // the three tap tiles (kh = 0..2) of depth slice dz, filter column kw, column tile col, channels c0..c0+31: one instruction per tap per wave
    auto issue_weights = [&](int dz, int kw, int col, int c0, char* dst) {
        int soff = (int)((((long long)(dz * 9 + kw) * a.Cout + col) * a.Cin + c0) * 2);
        asm volatile("" : "+s"(soff));
        const int tapstride = 3 * a.Cout * a.Cin * 2;          // tap index = dz*9 + kh*3 + kw
#pragma unroll
        for (int r = 0; r < WPW; ++r) {
            if (wave + 8 * r >= WINSTR) return;        // wave-uniform (NF = 2: the four waves of group 0 carry the whole tile; NF = 6: waves 0-3 carry a second round)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) pp_dma16(rw, soff + kh * tapstride + w_goff0[r], dst + kh * TAPB + (wave + 8 * r) * 1024);
        }
    };

    int pl, z, h0, w0, ncol0;
    decode(tile, pl, z, h0, w0, ncol0);
    f32x4 acc[NF][PF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) acc[f][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue_bias(ncol0, bbase);
    issue_pqr(pl / a.D, ncol0, pbase);
    issue_weights(0, 0, ncol0, 0, wbase);
    if (grp == 1) issue_weights(0, 1, ncol0, 0, wbase + WTILE);       // segment 1 (in the loop group 1 issues two segments ahead)
    pp_static_for<HJ>([&](auto jc) { issue_halo(jc, pl - 1, z >= 1, h0, w0, 0, hbase); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t mailbox = (uint32_t)(uintptr_t)(pbase + (EM == PP_EM_GN ? 2 * 3 * BN * 4 : 0));
    if (drawer) tq_post(tq, tk, mailbox);
    __syncthreads();
    int nxt = tile + tstride;
    if (dyn) nxt = tq_tile(tq, tq_take(mailbox), tstride);

    int wsel = 0, hsel = 0, bsel = 0;
    if (grp == 1) __builtin_amdgcn_s_barrier();       // the stagger: group 1 runs one slot behind group 0
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (;;) {
        const bool has_next = (unsigned)nxt < (unsigned)total_tiles;
        int npl = pl, nz = z, nh0 = h0, nw0 = w0, ncolN = ncol0;
        if (has_next) decode(nxt, npl, nz, nh0, nw0, ncolN);
        tk = 1u;
        if (drawer && has_next) tq_draw(tk, tq.ctr);          // the tile after `nxt`: posted after pair 0, taken at the end of the tile (nvc >= 3 pairs: barriers in between)
        int dz = 0, c0 = 0;
        // The K loop: pair 0 runs a copy of the segment code whose first filter row's MFMAs take the BIAS as their C operand (`first`) - no accumulator zeroing and no
        // bias add by vector instructions anywhere (pp_epilogue_plain<..., BINIT>; the same peeling as conv_ppc_kernel, conv_pp.hip)
        auto run_pair = [&](auto firstc, const int vc) __attribute__((always_inline)) {
            constexpr bool first = decltype(firstc)::value;
            // STREAM (256 VGPRs): hipcc kept `grp == 0` as a VECTOR boolean across the loops and spilled it to scratch - reloaded, with an s_waitcnt vmcnt(0), at the head of
            // every pair.  Laundered through an asm, the comparison is redone from the scalar register here.
            int grp_s = grp;
            if constexpr (STREAM) asm volatile("" : "+s"(grp_s));
#define GRP_ (STREAM ? grp_s : grp)
            // the (dz, chunk) pair after this one within the tile
            int ndz = dz, nc0 = c0 + 32;
            if (nc0 == a.Cin) {
                nc0 = 0;
                ndz = dz + 1;
            }
            const bool last_chunk = vc + 1 == nvc;
            const bool hnext = !last_chunk || has_next;
            // what the halo prefetch of this pair's segments fetches: the next pair of this tile, else pair (dz 0, chunk 0) of the next tile
            const int hdz = last_chunk ? 0 : ndz, hc0 = last_chunk ? 0 : nc0;
            const int hpl = last_chunk ? npl - 1 : pl + ndz - 1;
            const bool hval = last_chunk ? (nz >= 1) : ((unsigned)(z + ndz - 1) < (unsigned)a.D);
            const int hh0 = last_chunk ? nh0 : h0, hw0 = last_chunk ? nw0 : w0;
            const uint32_t hb = (uint32_t)(uintptr_t)hbase + hsel * HBUF;
            char* hbn = hbase + (hsel ^ 1) * HBUF;
            pp_static_for<3>([&](auto kc) {
                constexpr int kw = decltype(kc)::value;
                const uint32_t wb = (uint32_t)(uintptr_t)wbase + wsel * WTILE;
                char* wbn = wbase + (wsel ^ 1) * WTILE;
                char* wb_self = wbase + wsel * WTILE;
                // the segment after this one / the one after that
                const bool wnext = (kw < 2) || hnext;
                const int wkw = (kw < 2) ? kw + 1 : 0;
                const int wcol = (kw < 2 || !last_chunk) ? ncol0 : ncolN;
                const int wc0 = (kw < 2) ? c0 : hc0;
                const int wdz = (kw < 2) ? dz : hdz;
                const bool w2next = (kw < 1) || hnext;
                const int w2kw = (kw + 2) % 3;
                const int w2col = (kw < 1 || !last_chunk) ? ncol0 : ncolN;
                const int w2c0 = (kw < 1) ? c0 : hc0;
                const int w2dz = (kw < 1) ? dz : hdz;
                // ================= R segment =================
                if (GRP_ == 0 && wnext) issue_weights(wdz, wkw, wcol, wc0, wbn);
                constexpr int NH = kw == 0 ? HJ0 : (kw == 1 ? HJ1 : 0);          // halo instructions issued in this segment (per wave; the last may be past the image)
                if constexpr (NH > 0) {
                    if (hnext)
                        pp_static_for<NH>([&](auto jc) { issue_halo(std::integral_constant<int, (kw == 0 ? 0 : HJ0) + decltype(jc)::value>{}, hpl, hval, hh0, hw0, hc0, hbn); });
                }
                u32x4 A[3][NF], Brow[STREAM ? 3 : PF + 2];
                f32x4 bq[NF];          // first && kw == 0: the tile's bias, 4 values per fragment (the C operand of the first filter row's MFMAs)
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
                uint32_t ba_rows = hb + b_off0[kw];
                if constexpr (NF * PF >= 32) {          // 256-VGPR instantiations: the offset is re-derived (volatile lane index) instead of held across the loops - held, it spilled
                    int l_;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
                    const int px = (l_ & 15) + kw;
                    ba_rows = hb + (wm * PF * HW + px) * 64 + (((l_ >> 4) ^ ((px >> 1) & 3)) << 4);
                }
                // STREAM (PF 10 x 128 columns: 160 accumulator + 48 weight-fragment registers leave no room for twelve pixel rows): R reads rows 0 and 1 only, the M segment
                // streams the others through a three-slot rotation, each row read two rows ahead of its use
                pp_static_for<(STREAM ? 2 : PF + 2)>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    Brow[r] = pp_lds_read128<r * ROWB>(ba_rows);
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // how many of this wave's youngest DMAs may stay in flight: the halo instructions of THIS segment
                constexpr int KEEP = NH;
                if (GRP_ == 1) {                                  // group 1: its weight DMAs for the next segment (issued one slot pair ago, in its M) must have landed
                    if (hnext && KEEP > 0) {
                        const int last_id = ((kw == 0 ? 0 : HJ0) + KEEP - 1) * 8 + wave;      // wave-uniform: did this wave really issue KEEP halo instructions?
                        if (last_id < HINSTR) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP > 0 ? KEEP - 1 : 0) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ================= M segment: 3 taps x NF x PF MFMAs =================
                __builtin_amdgcn_s_setprio(1);
                if constexpr (STREAM) {
                    // by INPUT row r = 0 .. PF + 1: row r feeds tap kh of output row r - kh.  Every accumulator still receives its three taps in the order kh = 0, 1, 2 (rows
                    // pf, pf + 1, pf + 2), i.e. the sums are those of the row-major loop below, bit for bit.  Row r + 2 is read IN PLACE into the slot row r - 1 has just left
                    // ("+v": the tie is the write-after-read dependency and keeps hipcc from giving the late rows fresh registers); before row r is used, everything but the
                    // one read issued after it must have returned (LDS returns in order).
                    pp_static_for<PF + 2>([&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        if constexpr (r >= 2) {
                            __builtin_amdgcn_sched_barrier(0);
                            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(Brow[r % 3]) : "n"(r + 1 <= PF + 1 ? 1 : 0) : "memory");
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (r + 2 <= PF + 1) {          // row r + 2 -> the slot row r - 1 left in the previous iteration (row 2: the third, so far unused slot)
                            __builtin_amdgcn_sched_barrier(0);
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(Brow[(r + 2) % 3]) : "v"(ba_rows), "n"((r + 2) * ROWB));
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        pp_static_for<3>([&](auto hc) {
                            constexpr int kh = decltype(hc)::value, pf = r - kh;
                            if constexpr (pf >= 0 && pf < PF) {
#pragma unroll
                                for (int f = 0; f < NF; ++f) mma_b128<T>(acc[f][pf], A[kh][f], Brow[r % 3]);
                            }
                        });
                        if constexpr (r == 0) {
                            if (GRP_ == 1 && w2next) issue_weights(w2dz, w2kw, w2col, w2c0, wb_self);
                        }
                    });
                } else {
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
#pragma unroll
                            for (int pf = 0; pf < PF; ++pf) {
                                if constexpr (first && kw == 0) {
                                    if (kh == 0) acc[f][pf] = bq[f];          // (becomes the MFMA's C operand)
                                }
                                mma_b128<T>(acc[f][pf], A[kh][f], Brow[pf + kh]);
                            }
                            if (kh == 0 && f == 0) {
                                if (GRP_ == 1 && w2next) issue_weights(w2dz, w2kw, w2col, w2c0, wb_self);
                            }
                        }
                }
                __builtin_amdgcn_s_setprio(0);
                if (GRP_ == 0) {                                  // group 0: the weight DMAs it issued in this segment's R
                    if (hnext && KEEP > 0) {
                        const int last_id = ((kw == 0 ? 0 : HJ0) + KEEP - 1) * 8 + wave;
                        if (last_id < HINSTR) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP > 0 ? KEEP - 1 : 0) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                wsel ^= 1;
            });
            hsel ^= 1;
            dz = ndz;
            c0 = nc0;
        };
#undef GRP_
        // (the 256-VGPR instantiations, PF 8 x 128 columns, keep the vector-instruction bias add: with the bias quads live through the peeled pair they spill)
        constexpr bool BINIT = NF * PF < 32;
        if constexpr (BINIT) {
            run_pair(std::true_type{}, 0);
            if (drawer && has_next) tq_post(tq, tk, mailbox);
            // the next tile's bias slice -> the other half of the bias region (last read in the previous tile's first pair); it lands under the rest of this tile's K
            // loop (nvc >= 3) and is read in the next tile's first segment
            if (has_next) issue_bias(ncolN, bbase + (bsel ^ 1) * (BN * 4));
            if (has_next) issue_pqr(npl / a.D, ncolN, pbase + (bsel ^ 1) * (3 * BN * 4));      // (read by the NEXT tile's epilogue; this tile's sits in the other half)
#pragma unroll 1
            for (int vc = 1; vc < nvc; ++vc) run_pair(std::false_type{}, vc);
        } else {
#pragma unroll 1
            for (int vc = 0; vc < nvc; ++vc) run_pair(std::false_type{}, vc);
        }
        pp_epilogue_plain<NF, PF, EM, false, BINIT>(a, acc, (uint32_t)(uintptr_t)bbase + bsel * (BN * 4), pl, h0, w0, ncol0, wm, wn, u32x4{0u, 0u, 0u, 0u},
                                                    (uint32_t)(uintptr_t)pbase + bsel * (3 * BN * 4), BN * 4);
        pl = npl; z = nz; h0 = nh0; w0 = nw0; ncol0 = ncolN;
        bsel ^= 1;
        if constexpr (!BINIT) {
            if (has_next) issue_bias(ncol0, bbase + bsel * (BN * 4));
            if (has_next) issue_pqr(pl / a.D, ncol0, pbase + bsel * (3 * BN * 4));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!has_next) break;
        tile = nxt;
        nxt = dyn ? tq_tile(tq, tq_take(mailbox), tstride) : tile + tstride;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();       // pairs with group 1's last barrier
}

// ---------------------------------------------------------------------------------------------------------
bool conv3d_pp_eligible(const MisConvDesc* d) {
    if (d->dtype != MIS_BF16 || !d->is3d || d->ksize != 3) return false;
    if (d->x1 != nullptr || d->in_scale != nullptr) return false;
    if (d->x0_D != d->D || d->x0_H != d->H || d->x0_W != d->W) return false;
    const bool gn = d->gn_p != nullptr;
    if (d->Cin % 32 != 0 || d->Cout % 64 != 0) return false;
    // (GroupNorm-backward epilogue: columns past Cout0 may be padding that is dropped - then Cout0 only has to be a multiple of a 64-column block's wave slice)
    if (gn ? (d->Cout0 % 32 != 0 || (d->Cout0 % 64 != 0 && (d->y1 != nullptr || d->Cout != 64))) : d->Cout0 % 64 != 0) return false;
    if (gn && (d->gn_q == nullptr || d->gn_r == nullptr || d->mask == nullptr || d->gn_ld <= 0 || d->bias != nullptr || d->relu || (long long)d->N * d->gn_ld * 4 >= (1ll << 31))) return false;
    if (d->y0_mode != MIS_OUT_PLAIN || (d->y1 != nullptr && d->y1_mode != MIS_OUT_PLAIN)) return false;
    // 32-bit buffer offsets, computed in (signed) int: ONE depth plane of the input view and the packed weights must each span less than 2 GiB
    if ((((long long)d->H * d->W - 1) * d->x0_ld + d->Cin) * 2 >= (1ll << 31) - 65536) return false;
    if ((long long)27 * d->Cout * d->Cin * 2 >= (1ll << 31) - 65536) return false;
    // pp_epilogue_plain: one depth plane of each destination view / of the mask within 32-bit buffer offsets
    const long long img = (long long)d->H * d->W, lim = (1ll << 32) - 65536;
    if (((img - 1) * d->y0_ld + d->Cout0) * 2 >= lim) return false;
    if (d->Cout0 < d->Cout && !(gn && d->y1 == nullptr) && (d->y1 == nullptr || ((img - 1) * d->y1_ld + (d->Cout - d->Cout0)) * 2 >= lim)) return false;
    if (d->mask != nullptr && ((img - 1) * d->mask_ld + d->Cout) * 2 >= lim) return false;
    return true;
}

template <int PF, int NF, int EM> static int pp3_launch_em(const MisConvDesc* d, hipStream_t stream);
template <int PF, int NF> static int pp3_launch(const MisConvDesc* d, hipStream_t stream) {      // one instantiation per epilogue mask path (3-D: none, the bf16 mask, GroupNorm backward)
    if constexpr (PF == 10 && NF == 4) {      // the streamed 40-row tile: plain epilogue only (with a mask operand one more register is live and hipcc spills the weight DMA offset)
        MIS_REQUIRE(d->gn_p == nullptr && d->mask == nullptr, MIS_EUNSUPPORTED, "conv_igemm(3d pp): the 40-row 128-column tile has no masked epilogue");
        return pp3_launch_em<PF, NF, PP_EM_NONE>(d, stream);
    } else {
        if (d->gn_p != nullptr) {
            if constexpr (NF == 6) {
                MIS_REQUIRE(false, MIS_EUNSUPPORTED, "conv_igemm(3d pp): the GroupNorm-backward epilogue is not built for 192-column blocks");
            } else {
                return pp3_launch_em<PF, NF, PP_EM_GN>(d, stream);
            }
        }
        if (d->mask != nullptr) return pp3_launch_em<PF, NF, PP_EM_MASK>(d, stream);
        return pp3_launch_em<PF, NF, PP_EM_NONE>(d, stream);
    }
}
template <int PF, int NF, int EM> static int pp3_launch_em(const MisConvDesc* d, hipStream_t stream) {
    constexpr int BN = 2 * NF * 16;
    constexpr int TH = 4 * PF, HINSTR = ((TH + 2) * 18 * 4 + 63) / 64;
    ConvArgs a;
    a.tq = nullptr;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.Cin0 = d->Cin; a.Cout0 = d->Cout0;
    a.x0 = SrcView{d->x0, d->x0_ld, d->x0_D, d->x0_H, d->x0_W};
    a.x1 = SrcView{nullptr, 0, 0, 0, 0};
    a.in_scale = nullptr; a.in_shift = nullptr;
    a.w = d->w; a.bias = d->bias; a.relu = d->relu; a.mask = d->mask; a.mask_ld = d->mask_ld; a.relu_bits = nullptr; a.mask_bits = reinterpret_cast<const unsigned char*>(d->mask_bits);
    a.y0 = d->y0; a.y0_ld = d->y0_ld; a.y0_mode = d->y0_mode;
    a.y1 = d->y1; a.y1_ld = d->y1_ld; a.y1_mode = d->y1_mode;
    a.gn_p = d->gn_p; a.gn_q = d->gn_q; a.gn_r = d->gn_r; a.gn_ld = d->gn_ld; a.gn_relu = d->gn_relu;
    a.tilesD = d->D;
    a.tilesH = (d->H + TH - 1) / TH;
    a.tilesW = (d->W + 15) / 16;
    const long long nsp = (long long)d->N * d->D * a.tilesH * a.tilesW;
    a.nCt = d->Cout / BN;
    // spatial-major tile order when the persistent stride keeps a block on its column tile (256 % nCt == 0)
    a.order = (256 % a.nCt == 0 && !mis_sw(SW_CONV3D_COLMAJOR)) ? 2 : 1;
    int zg = mis_sw(SW_CONV3D_ZG);
    if (zg < 1) zg = 1;
    if (zg > d->D) zg = d->D;
    a.zg = zg;
    MIS_REQUIRE(nsp * a.nCt < (1ll << 31), MIS_EUNSUPPORTED, "conv_igemm(3d pp): grid too large");
    a.nSp = (int)nsp;
    const size_t lds = 2 * (size_t)HINSTR * 1024 + 2 * (size_t)3 * BN * 64 + 2 * (size_t)BN * 4 + (EM == PP_EM_GN ? 2 * (size_t)3 * BN * 4 : 0) + 16;      // (+ the tile queue's mailbox)
    a.tq = mis_tile_queue(stream);
    static std::atomic<unsigned long long> attr_done{0};
    if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&conv3d_ppc_kernel<PF, NF, EM>), lds, "conv_igemm(3d pp)")) return rc;
    const long long total = nsp * a.nCt;
    hipLaunchKernelGGL((conv3d_ppc_kernel<PF, NF, EM>), dim3((unsigned)(total > mis_persist_cus() ? mis_persist_cus() : total)), dim3(512), lds, stream, a);
    MIS_LAUNCH_CHECK("conv_igemm(3d pp)");
    return MIS_OK;
}

int launch_conv3d_pp(const MisConvDesc* d, hipStream_t stream, const char** tag) {
    // column blocks: 128 (Cout % 128 == 0), 192 (Cout % 192 == 0: the 64 -> 192 dgrad of decoders.2 in ONE column tile instead of three 64-column ones), else 64.
    // rows per tile: 20 (PF 5), 32 (PF 8) or - 64-column blocks only, the others have no registers for it - 40 (PF 10): whichever pads the H axis least, ties to the
    // taller tile (fewer fragment reads and barriers per MFMA); MIS_CONV3D_PF forces one.
    const bool gn = d->gn_p != nullptr;
    const int nf = d->Cout % 128 == 0 ? 4 : ((d->Cout % 192 == 0 && d->Cout0 % 192 == 0 && !gn) ? 6 : 2);
    int pf = mis_sw(SW_CONV3D_PF);
    if (!(pf == 5 || (pf == 8 && nf != 6) || (pf == 10 && (nf == 2 || (nf == 4 && !gn && d->mask == nullptr))))) {
        const int cand[3] = {5, 8, 10};
        int best = 1 << 30;
        pf = 5;
        for (int i = 0; i < 3; ++i) {
            if ((cand[i] == 10 && nf == 6) || (cand[i] == 8 && nf == 6)) continue;
            if (cand[i] == 10 && nf == 4 && (gn || d->mask != nullptr || mis_sw(SW_CONV3D_NOPF10N4))) continue;      // the streamed 40-row 128-column tile of round 5: plain epilogue only; MIS_CONV3D_NOPF10N4=1: A/B switch
            const int th = 4 * cand[i], pad = (d->H + th - 1) / th * th;
            if (pad <= best) {
                best = pad;
                pf = cand[i];
            }
        }
    }
    if (nf == 4) {
        if (pf == 5) {
            *tag = gn ? "k3.3d.ppc5.gn" : (d->mask != nullptr ? "k3.3d.ppc5.mask" : "k3.3d.ppc5");
            return pp3_launch<5, 4>(d, stream);
        }
        if (pf == 10) {
            *tag = gn ? "k3.3d.ppc10.gn" : (d->mask != nullptr ? "k3.3d.ppc10.mask" : "k3.3d.ppc10");
            return pp3_launch<10, 4>(d, stream);
        }
        *tag = gn ? "k3.3d.ppc8.gn" : (d->mask != nullptr ? "k3.3d.ppc8.mask" : "k3.3d.ppc8");
        return pp3_launch<8, 4>(d, stream);
    }
    if (nf == 6) {
        *tag = d->mask != nullptr ? "k3.3d.ppc5n6.mask" : "k3.3d.ppc5n6";
        return pp3_launch<5, 6>(d, stream);
    }
    if (pf == 5) {
        *tag = gn ? "k3.3d.ppc5n2.gn" : (d->mask != nullptr ? "k3.3d.ppc5n2.mask" : "k3.3d.ppc5n2");
        return pp3_launch<5, 2>(d, stream);
    }
    if (pf == 8) {
        *tag = gn ? "k3.3d.ppc8n2.gn" : (d->mask != nullptr ? "k3.3d.ppc8n2.mask" : "k3.3d.ppc8n2");
        return pp3_launch<8, 2>(d, stream);
    }
    *tag = gn ? "k3.3d.ppc10n2.gn" : (d->mask != nullptr ? "k3.3d.ppc10n2.mask" : "k3.3d.ppc10n2");
    return pp3_launch<10, 2>(d, stream);
}
