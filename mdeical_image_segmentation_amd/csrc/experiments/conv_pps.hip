// bf16 2-D 3x3 convolution, Cout % 128 == 0: ONE wave per SIMD with both fragment sets in registers (round 5 experiment, MIS_CONV_PPS=1).
//
// conv_ppc_kernel (conv_pp.hip) runs two waves per SIMD in anti-phase: one reads fragments / issues DMAs (R) while the other issues 96 MFMAs (M).  Its stamp table
// (profiles/r05_ppc_stamps.txt) says where that structure stops: a wave spends 43 % of its time in M segments, but an M segment takes ~750 instead of 512 cycles per 32
// MFMAs because the partner's R instructions (22 fragment reads, 4-6 DMA issues whose per-lane offsets are re-derived each time: no register is left at 256) share the
// SIMD's issue port with MFMAs that each hold it for 8 of their 16 cycles: pipe busy 73.5 %.
// The fp32 kernels of this round (conv3d_f32.hip) reach 95-100 % with the opposite structure - every wave prefetches the NEXT step's fragments into a second register
// set under its own MFMAs, one barrier per step placed inside the cluster - which does not fit here at 256 registers (128 accumulators + 2 x 88 fragment registers).  It
// does at 512: four waves per block, one block per CU.  What that buys: nothing but MFMAs, 22 ds_read_b128 and ~8 DMA issues per 96-MFMA segment on a SIMD (the MFMAs
// leave 768 issue cycles per segment), every per-lane DMA offset precomputed once per kernel (registers to spare), no second barrier.  What it costs: half the pixels per
// block (256 px x 128 ch: twice the weight-tile traffic per MFMA, all L2-resident), nobody to cover the tile epilogue.
// Same LDS images, swizzles, weight row permutation and epilogue (pp_epilogue_plain) as conv_ppc_kernel<8, 4>; tile = 16 rows x 16 columns, wave (wm, wn) owns rows
// 8 wm .. 8 wm + 7 x channels 64 wn .. 64 wn + 63.  Segment = one filter column of one 32-channel chunk = 12 groups of 8 MFMAs:
//   group 0 | s_waitcnt vmcnt; s_barrier | groups 1-6: one weight DMA (segment after next) + two fragment reads (next segment) each | groups 7-11: halo pieces of the next
//   chunk (first segment of a chunk only) + two reads each | s_waitcnt lgkmcnt(0).
#include "common.hpp"
#include "conv_args.hpp"
#include "conv_pp_common.hpp"
#include "dispatch_cfg.hpp"

namespace {
constexpr int PS_TH = 16, PS_TW = 16, PS_HW = 18, PS_HP = 18 * 18;
constexpr int PS_HITEMS = PS_HP * 4;             // 1296 16-byte items
constexpr int PS_HINSTR = 24;                    // 20.25 instructions of data; 24 issued (six per wave, the tail out of range) so that every wave counts the same
constexpr int PS_HBUF = PS_HINSTR * 1024, PS_ROWB = PS_HW * 64;
constexpr int PS_TAPB = 128 * 64, PS_WTILE = 3 * PS_TAPB;
constexpr int PS_LDS = 2 * PS_HBUF + 2 * PS_WTILE + 2 * 128 * 4;      // 49,152 + 49,152 + 1,024
// (a __device__ function: a kernel TEMPLATE that names the builtin in its own body may lose its host stub - see conv3d_f32.hip)
__device__ __forceinline__ void ps_dma16s(__amdgpu_buffer_rsrc_t r, int voff, int soff, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (pp_lds_void_t*)lds_dst_wave_uniform, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void ps_dma4(__amdgpu_buffer_rsrc_t r, int voff, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (pp_lds_void_t*)lds_dst_wave_uniform, 4, voff, 0, 0, 0);
}
}   // namespace

template <int EM>
__global__ __launch_bounds__(256, 1) void conv_pps_kernel(const ConvArgs a) {
    using T = __bf16;
    constexpr int NF = 4, PF = 8, WAVE_N = 64, BN = 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hbase = smem;
    char* const wbase = smem + 2 * PS_HBUF;
    char* const bbase = wbase + 2 * PS_WTILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;

    const int total_tiles = a.nSp * a.nCt;
    const int tstride = (int)gridDim.x;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= total_tiles) return;
    const int nchunks = a.Cin >> 5;              // even (Cin % 64 == 0)
    const int tpi = a.tilesH * a.tilesW;
    auto decode = [&](int t, int& tn, int& th0, int& tw0, int& tcol) {      // spatial major: the column tiles of a spatial tile run side by side
        const int sp = t / a.nCt, ct = t - sp * a.nCt;
        tn = sp / tpi;
        const int r = sp - tn * tpi;
        const int th = r / a.tilesW;
        th0 = th * PS_TH;
        tw0 = (r - th * a.tilesW) * PS_TW;
        tcol = ct * BN;
    };

    // ---- fragment addresses (conv_ppc_kernel's) ----
    const uint32_t a_off0 = (uint32_t)((wn * WAVE_N + li) * 64 + ((lg ^ ((li >> 1) & 3)) << 4));
    uint32_t b_off0[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int px = li + kw;
        b_off0[kw] = (uint32_t)((wm * PF * PS_HW + px) * 64 + ((lg ^ ((px >> 1) & 3)) << 4));
    }
    // ---- DMA lane parts, once per kernel ----
    // weights: instruction t = wave * 6 + k of a segment's 24 (tap kh = t >> 3, rows 16 (t & 7) .. + 15 of the tap tile)
    int woff[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int t = wave * 6 + k;
        const int slot = (t & 7) * 64 + lane;
        const int lrow = slot >> 2, pos = slot & 3;
        const int dc16 = pos ^ ((lrow >> 1) & 3);
        const int dwv = lrow / WAVE_N, j = lrow % WAVE_N;
        const int drow = dwv * WAVE_N + ((j >> 5) * 32) + ((j & 15) >> 2) * 8 + ((j >> 4) & 1) * 4 + (j & 3);
        woff[k] = (drow * a.Cin + dc16 * 8) * 2;
    }
    // halo: instruction id = j * 4 + wave; item -> halo pixel (py, px), chunk position
    int hrel[6], hyx[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int item = (j * 4 + wave) * 64 + lane;
        const int p = item >> 2, pos = item & 3;
        const int py = p / PS_HW, px = p - py * PS_HW;
        hrel[j] = item < PS_HITEMS ? ((py * a.W + px) * a.x0.ld + ((pos ^ ((px >> 1) & 3)) << 3)) * 2 : PP_OOB;
        hyx[j] = item < PS_HITEMS ? (py << 8) | px : 0x7fff7fff;
    }
    const unsigned img_x = (unsigned)(((long long)a.H * a.W - 1) * a.x0.ld + a.Cin) * 2u;
    const char* const xb = reinterpret_cast<const char*>(a.x0.p);
    const __amdgpu_buffer_rsrc_t rw = pp_make_rsrc(a.w, (unsigned)((long long)9 * a.Cout * a.Cin * 2));
    const __amdgpu_buffer_rsrc_t rb = pp_make_rsrc(a.bias != nullptr ? (const void*)a.bias : a.w, a.bias != nullptr ? (unsigned)a.Cout * 4u : 0u);
    const int tapstride = 3 * a.Cout * a.Cin * 2;

    // ---- DMA issue, a few instructions each: every per-lane part is a register, every scalar part is prepared once per tile / segment ----
    // weights: voffset = woff[k], soffset = the segment's base + this instruction's tap; halo: voffset = hrel[j], soffset = the tile / chunk origin for INTERIOR tiles (the
    // range check of a raw buffer does not see soffset: invalid lanes carry PP_OOB in voffset); tiles on the image border add the origin on the VALU and test every lane
    // (plain scalars, every one passed through readfirstlane where it meets a DMA: with the tile in a struct selected by reference hipcc kept it in SCRATCH and wrapped
    //  the halo DMAs in waterfall loops - 6 scratch loads and 18 readfirstlane loops per segment, 950 instead of 1240 TFLOP/s)
    auto tile_scalars = [&](int t, int& tn, int& th0, int& tw0, int& tcol, unsigned& blo, unsigned& bhi, int& toff, int& inter) {
        decode(t, tn, th0, tw0, tcol);
        const unsigned long long b = (unsigned long long)(uintptr_t)xb + (unsigned long long)tn * a.H * a.W * a.x0.ld * 2ull;
        blo = (unsigned)b;
        bhi = (unsigned)(b >> 32);
        toff = (((th0 - 1) * a.W + (tw0 - 1)) * a.x0.ld) * 2;
        inter = (th0 >= 1 && th0 + PS_TH + 1 <= a.H && tw0 >= 1 && tw0 + PS_TW + 1 <= a.W) ? 1 : 0;
    };
    auto issue_halo = [&](auto jc, unsigned blo, unsigned bhi, int org, int inter, int th0, int tw0, char* dst) {
        constexpr int j = decltype(jc)::value;
        const unsigned long long b = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)bhi) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)blo);
        const __amdgpu_buffer_rsrc_t rx = pp_make_rsrc(reinterpret_cast<const void*>((uintptr_t)b), img_x);
        const int so = __builtin_amdgcn_readfirstlane(org);
        if (__builtin_amdgcn_readfirstlane(inter)) {
            ps_dma16s(rx, hrel[j], so, dst + (j * 4 + wave) * 1024);
        } else {
            const int py = hyx[j] >> 8, px = hyx[j] & 0xff;
            const bool ok = (unsigned)(th0 - 1 + py) < (unsigned)a.H && (unsigned)(tw0 - 1 + px) < (unsigned)a.W;
            pp_dma16(rx, ok ? (int)((unsigned)so + (unsigned)hrel[j]) : PP_OOB, dst + (j * 4 + wave) * 1024);
        }
    };
    int wtap[6];                        // tap part of a weight instruction's scalar offset
#pragma unroll
    for (int k = 0; k < 6; ++k) wtap[k] = __builtin_amdgcn_readfirstlane(((wave * 6 + k) >> 3) * tapstride);
    auto issue_w = [&](auto kc, int segbase, char* dst) {
        constexpr int k = decltype(kc)::value;
        ps_dma16s(rw, woff[k], __builtin_amdgcn_readfirstlane(segbase + wtap[k]), dst + (wave * 6 + k) * 1024);
    };
    auto seg_base = [&](int kw, int col, int c0) { return (int)((((long long)kw * a.Cout + col) * a.Cin + c0) * 2); };
    auto issue_bias = [&](int col, char* dst) {
        if (wave < 2) ps_dma4(rb, (col + wave * 64 + lane) * 4, dst + wave * 256);
    };

    int c_n, c_h0, c_w0, c_col, c_toff, c_int, n_n, n_h0, n_w0, n_col, n_toff, n_int;
    unsigned c_blo, c_bhi, n_blo, n_bhi;
    tile_scalars(tile, c_n, c_h0, c_w0, c_col, c_blo, c_bhi, c_toff, c_int);
    n_n = c_n; n_h0 = c_h0; n_w0 = c_w0; n_col = c_col; n_toff = c_toff; n_int = c_int; n_blo = c_blo; n_bhi = c_bhi;
    f32x4 acc[NF][PF];

    // ---- prologue: bias, halo of chunk 0, weight tiles of segments 0 and 1 ----
    issue_bias(c_col, bbase);
    pp_static_for<6>([&](auto jc) { issue_halo(jc, c_blo, c_bhi, c_toff, c_int, c_h0, c_w0, hbase); });
    pp_static_for<6>([&](auto kc) { issue_w(kc, seg_base(0, c_col, 0), wbase); });
    pp_static_for<6>([&](auto kc) { issue_w(kc, seg_base(1, c_col, 0), wbase + PS_WTILE); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_sleep(2);

    u32x4 A0[3][NF], B0[PF + 2], A1[3][NF], B1[PF + 2];
    auto read_A = [&](u32x4(&A)[3][NF], uint32_t wba, auto idxc) {         // fragment idx = kh * 4 + f; wba = weight tile base + a_off0
        constexpr int idx = decltype(idxc)::value, kh = idx / 4, f = idx % 4;
        A[kh][f] = pp_lds_read128<kh * PS_TAPB + f * 1024>(wba);
    };
    auto read_B = [&](u32x4(&B)[PF + 2], uint32_t hbk, auto rc) {          // hbk = halo base + b_off0[kw]
        constexpr int r = decltype(rc)::value;
        B[r] = pp_lds_read128<r * PS_ROWB>(hbk);
    };
    {
        const uint32_t wba = (uint32_t)(uintptr_t)wbase + a_off0, hbk = (uint32_t)(uintptr_t)hbase + b_off0[0];
        pp_static_for<12>([&](auto ic) { read_A(A0, wba, ic); });
        pp_static_for<PF + 2>([&](auto rc) { read_B(B0, hbk, rc); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    int wsel = 0, hsel = 0, bsel = 0;

#pragma unroll 1
    for (; tile < total_tiles; tile += tstride) {
        const bool has_next = tile + tstride < total_tiles;
        if (has_next) tile_scalars(tile + tstride, n_n, n_h0, n_w0, n_col, n_blo, n_bhi, n_toff, n_int);
        u32x4 mbits = u32x4{0u, 0u, 0u, 0u};
        if constexpr (EM == PP_EM_BITS) mbits = pp_mask_bits_load<NF, PF>(a, c_n, c_h0, c_w0, c_col, wm, wn);
        f32x4 bq[NF];
        {
            const uint32_t ba = (uint32_t)(uintptr_t)bbase + bsel * (BN * 4) + (wn * WAVE_N + lg * 8) * 4;
            pp_static_for<NF>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                const u32x4 r = pp_lds_read128<(f >> 1) * 128 + (f & 1) * 16>(ba);
                bq[f] = __builtin_bit_cast(f32x4, r);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }

        // one segment: `cur` = this segment's fragments (in registers), `nxt` = the set the next segment's are read into.  first = first segment of the tile (bias as C).
        // The 96 MFMAs go out one by one; behind every second one from the ninth on ONE small action follows (a weight DMA, a halo DMA or a fragment read: 1-3
        // instructions), so that the wave never holds the issue port long enough to let the pipe run dry
        auto segment = [&](auto kwc, auto firstc, u32x4(&Ac)[3][NF], u32x4(&Bc)[PF + 2], u32x4(&An)[3][NF], u32x4(&Bn)[PF + 2], const int chunk) __attribute__((always_inline)) {
            constexpr int kw = decltype(kwc)::value;
            constexpr bool first = decltype(firstc)::value;
            const int c0 = chunk << 5;
            const bool last_chunk = chunk + 1 == nchunks;
            const bool hnext = !last_chunk || has_next;                       // a next chunk exists (this tile's or the next tile's first)
            // the tile of the next chunk
            const unsigned hblo = last_chunk ? n_blo : c_blo, hbhi = last_chunk ? n_bhi : c_bhi;
            const int hint = last_chunk ? n_int : c_int, hth0 = last_chunk ? n_h0 : c_h0, htw0 = last_chunk ? n_w0 : c_w0;
            const int hc0 = last_chunk ? 0 : c0 + 32;
            const int horg = (last_chunk ? n_toff : c_toff) + hc0 * 2;
            const bool snext = (kw < 2) || hnext;
            const bool s2next = (kw < 1) || hnext;
            const int w2base = seg_base((kw + 2) % 3, (kw < 1 || !last_chunk) ? c_col : n_col, (kw < 1) ? c0 : hc0);
            const uint32_t wbna = (uint32_t)(uintptr_t)wbase + (wsel ^ 1) * PS_WTILE + a_off0;
            char* const wb_self = wbase + wsel * PS_WTILE;
            const uint32_t hbn = (uint32_t)(uintptr_t)hbase + (kw < 2 ? hsel : hsel ^ 1) * PS_HBUF + b_off0[(kw + 1) % 3];
            char* const hdst = hbase + (hsel ^ 1) * PS_HBUF;

            pp_static_for<96>([&](auto ic) {
                constexpr int i = decltype(ic)::value, g = i / 8, pf = i % 8, kh = g / 4, f = g % 4;
                if constexpr (first && kh == 0) acc[f][pf] = bq[f];          // (the MFMA's C operand)
                mma_b128<T>(acc[f][pf], Ac[kh][f], Bc[pf + kh]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (i == 7) {
                    // the next segment's weight tile (issued behind the previous barrier) has landed; the six halo pieces a chunk's first segment issued behind its
                    // weights may stay in flight through the second segment's barrier
                    if constexpr (kw == 1) {
                        if (hnext) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (i >= 9 && (i - 9) % 2 == 0 && (i - 9) / 2 < 34) {
                    constexpr int s = (i - 9) / 2;                            // action slot 0 .. 33
                    // slots 0-17: [W k, R 2k, R 2k + 1] for k = 0..5 (weights of the segment after next into the buffer this segment's fragments came from; reads of the
                    // next segment's weight fragments); slots 18-33: [H j, R .., R ..] (halo pieces of the next chunk in a chunk's first segment; pixel-row fragments)
                    if constexpr (s < 18) {
                        if constexpr (s % 3 == 0) {
                            if (s2next) issue_w(std::integral_constant<int, s / 3>{}, w2base, wb_self);
                        } else {
                            if (snext) read_A(An, wbna, std::integral_constant<int, (s / 3) * 2 + (s % 3) - 1>{});
                        }
                    } else {
                        constexpr int u = s - 18;                             // 0 .. 15
                        if constexpr (u % 3 == 0) {
                            if constexpr (kw == 0) {
                                if (hnext) issue_halo(std::integral_constant<int, u / 3>{}, hblo, hbhi, horg, hint, hth0, htw0, hdst);
                            }
                        } else {
                            constexpr int r = (u / 3) * 2 + (u % 3) - 1;      // 0 .. 9 (u = 15 is H5: see below)
                            if (snext) read_B(Bn, hbn, std::integral_constant<int, r>{});
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            wsel ^= 1;
            if constexpr (kw == 2) hsel ^= 1;
        };
        using K0 = std::integral_constant<int, 0>;
        using K1 = std::integral_constant<int, 1>;
        using K2 = std::integral_constant<int, 2>;
        // chunks in pairs: the register sets alternate per segment, three segments per chunk
        segment(K0{}, std::true_type{}, A0, B0, A1, B1, 0);
        // the next tile's bias slice -> the other half of the bias region (lands under this tile's K loop)
        if (has_next) issue_bias(n_col, bbase + (bsel ^ 1) * (BN * 4));
        segment(K1{}, std::false_type{}, A1, B1, A0, B0, 0);
        segment(K2{}, std::false_type{}, A0, B0, A1, B1, 0);
        segment(K0{}, std::false_type{}, A1, B1, A0, B0, 1);
        segment(K1{}, std::false_type{}, A0, B0, A1, B1, 1);
        segment(K2{}, std::false_type{}, A1, B1, A0, B0, 1);
#pragma unroll 1
        for (int chunk = 2; chunk < nchunks; chunk += 2) {
            segment(K0{}, std::false_type{}, A0, B0, A1, B1, chunk);
            segment(K1{}, std::false_type{}, A1, B1, A0, B0, chunk);
            segment(K2{}, std::false_type{}, A0, B0, A1, B1, chunk);
            segment(K0{}, std::false_type{}, A1, B1, A0, B0, chunk + 1);
            segment(K1{}, std::false_type{}, A0, B0, A1, B1, chunk + 1);
            segment(K2{}, std::false_type{}, A1, B1, A0, B0, chunk + 1);
        }
        pp_epilogue_plain<NF, PF, EM, true, true>(a, acc, 0u, c_n, c_h0, c_w0, c_col, wm, wn, mbits);
        c_n = n_n; c_h0 = n_h0; c_w0 = n_w0; c_col = n_col; c_toff = n_toff; c_int = n_int; c_blo = n_blo; c_bhi = n_bhi;
        bsel ^= 1;
        __builtin_amdgcn_sched_barrier(0);
    }
}

bool conv_pps_eligible(const MisConvDesc* d) {
    if (!mis_sw(SW_CONV_PPS)) return false;
    if (!conv_pp_eligible(d) || conv_ppc_choice(d) != 4) return false;
    if (d->Cin % 64 != 0 || d->Cout % 128 != 0) return false;
    return true;
}

template <int EM> static int pps_launch_em(const MisConvDesc* d, hipStream_t stream) {
    ConvArgs a;
    a.tq = nullptr;
    a.N = d->N; a.D = 1; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.Cin0 = d->Cin; a.Cout0 = d->Cout0;
    a.x0 = SrcView{d->x0, d->x0_ld, 1, d->x0_H, d->x0_W};
    a.x1 = SrcView{nullptr, 0, 0, 0, 0};
    a.in_scale = nullptr; a.in_shift = nullptr;
    a.w = d->w; a.bias = d->bias; a.relu = d->relu; a.mask = d->mask; a.mask_ld = d->mask_ld; a.mask_bits = reinterpret_cast<const unsigned char*>(d->mask_bits);
    a.y0 = d->y0; a.y0_ld = d->y0_ld; a.y0_mode = d->y0_mode;
    a.y1 = d->y1; a.y1_ld = d->y1_ld; a.y1_mode = d->y1_mode;
    a.relu_bits = reinterpret_cast<unsigned char*>(d->relu_bits);
    g_conv_bits_fused = d->relu_bits != nullptr;
    a.gn_p = nullptr; a.gn_q = nullptr; a.gn_r = nullptr; a.gn_ld = 0; a.gn_relu = 0;
    a.tilesD = 2;
    a.tilesH = (d->H + PS_TH - 1) / PS_TH;
    a.tilesW = (d->W + PS_TW - 1) / PS_TW;
    const long long nsp = (long long)d->N * a.tilesH * a.tilesW;
    a.nCt = d->Cout / 128;
    MIS_REQUIRE(nsp * a.nCt < (1ll << 31), MIS_EUNSUPPORTED, "conv_igemm(pps): grid too large");
    a.nSp = (int)nsp;
    static std::atomic<unsigned long long> attr_done{0};
    if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&conv_pps_kernel<EM>), PS_LDS, "conv_igemm(pps)")) return rc;
    const long long total = nsp * a.nCt;
    hipLaunchKernelGGL(conv_pps_kernel<EM>, dim3((unsigned)(total > mis_persist_cus() ? mis_persist_cus() : total)), dim3(256), PS_LDS, stream, a);
    MIS_LAUNCH_CHECK("conv_igemm(pps)");
    return MIS_OK;
}

int launch_conv_pps(const MisConvDesc* d, hipStream_t stream, const char** tag) {
    if (d->mask_bits != nullptr) {
        *tag = "k3.2d.pps.bits";
        return pps_launch_em<PP_EM_BITS>(d, stream);
    }
    if (d->mask != nullptr) {
        *tag = "k3.2d.pps.mask";
        return pps_launch_em<PP_EM_MASK>(d, stream);
    }
    *tag = "k3.2d.pps";
    return pps_launch_em<PP_EM_NONE>(d, stream);
}
