// conv_ppc_kernel<8, 4> with the DMA issue and the R segment cut down (round 5, MIS_CONV_PPC2): see conv_pp.hip for the column-segment ping-pong structure this file modifies.
//
// The stamp table of conv_ppc_kernel<8, 4> (profiles/r05_ppc_stamps.txt) puts the DMA ISSUE at 21-24 % of a wave's cycles and the 22 fragment reads at 7.5 %; an M segment is
// stretched from 512 to ~750 cycles per 32 MFMAs by the partner's R instructions, and conv_pps.hip showed that a non-MFMA instruction costs the SIMD ~10 cycles whichever wave
// issues it.  So: fewer instructions.
//   * per-lane halo DMA offsets are tile-invariant and live in HJ = 5 registers; an interior tile's halo issue is a scalar origin + pp_dma16s (4 instead of ~30 instructions); weight
//     issues carry their tap / column / chunk offset in the instruction's scalar operand;
//   * the registers for that come out of the fragments: R reads the ten pixel rows and only FIVE of the twelve weight fragments; the other seven roll through a five-slot
//     rotation inside the M segment (fragment t = kh * 4 + f lives in slot t % 5 and is read behind MFMA group t - 5: four groups = 512 pipe cycles of lead, counted lgkmcnt):
//     peak 128 + 20 + 40 = 188 fragment / accumulator registers instead of 216;
//   * the current segment's weight tile therefore stays live through M, and the refill schedule changes: BOTH groups fetch their part of the NEXT segment's tile during the
//     current segment - group 1 at the start of its R, group 0 behind the first MFMA group of its M - into the buffer of the PREVIOUS segment (last read one slot earlier).
// MFMA order per accumulator unchanged: bit-identical to conv_ppc_kernel<8, 4> (tests/test_gpu_dispatch_parity.py).
#include "common.hpp"
#include "conv_args.hpp"
#include "conv_pp_common.hpp"
#include "dispatch_cfg.hpp"

// diagnostic build (-DMIS_PP_STAMPS, never shipped; scripts/ppc_stamps.sh): slots as in conv_pp.hip, read back with mis_debug_ppc2_stamps()
#ifdef MIS_PP_STAMPS
__device__ unsigned long long g_ppc2_stamps[256 * 8 * 8];
extern "C" int mis_debug_ppc2_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ppc2_stamps), sizeof(unsigned long long) * 256 * 8 * 8) == hipSuccess ? 0 : -1;
}
#define PP_STAMP(i)                                                  \
    {                                                                \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); \
        st_[i] += tn_ - tp_;                                         \
        tp_ = tn_;                                                   \
    }
#else
#define PP_STAMP(i)
#endif

namespace {
__device__ __forceinline__ void ppc2_dma4(__amdgpu_buffer_rsrc_t r, int voff, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (pp_lds_void_t*)lds_dst_wave_uniform, 4, voff, 0, 0, 0);
}
}   // namespace

template <int EM>
__global__ __launch_bounds__(512, 2) void conv_ppc2_kernel(const ConvArgs a) {
    using T = __bf16;
    constexpr int PF = 8, NF = 4;
    constexpr int WAVE_N = NF * 16, BN = 2 * WAVE_N, NV = 4 * NF;       // NF = 4: 128-column blocks; NF = 2: 64-column blocks (wave tile 128 px x 32 ch)
    constexpr int WWAVES = BN / 16;                                      // waves that carry a weight DMA instruction per tap (16 rows of 64 B each)
    constexpr int TH = 4 * PF, TW = 16, HH = TH + 2, HW = 18, HP = HH * HW;
    constexpr int HITEMS = HP * 4, HINSTR = (HITEMS + 63) / 64, HBUF = HINSTR * 1024, ROWB = HW * 64;
    constexpr int HJ = (HINSTR + 7) / 8;             // halo instructions per wave per chunk (PF 8: 5, PF 6: 4)
    constexpr int HJ0 = HJ < 3 ? HJ : 3, HJ1 = HJ - HJ0;
    constexpr int TAPB = BN * 64;                    // one tap's weight tile
    constexpr int WTILE = 3 * TAPB;                  // one column's weight tiles

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hbase = smem;                        // 2 x HBUF
    char* const wbase = smem + 2 * HBUF;             // 2 x WTILE
    char* const bbase = wbase + 2 * WTILE;           // 2 x BN floats
    char* const tbase = bbase + 2 * BN * 4;          // halo DMA offset table: 8 waves x 64 lanes x 32 bytes

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    const int li = lane & 15, lg = lane >> 4;

    const int total_tiles = a.nSp * a.nCt;
    const int tstride = (int)gridDim.x;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= total_tiles) return;                 // block-uniform
#ifdef PPT_DESYNC           // diagnostic build: blocks of one XCD start PPT_DESYNC x 64 cycles apart, so that their tile boundaries (store bursts) do not coincide
    for (int i = (int)((blockIdx.x >> 3) & 31); i > 0; --i) __builtin_amdgcn_s_sleep(PPT_DESYNC);
#endif
    const int nchunks = a.Cin >> 5;
    const int tpi = a.tilesH * a.tilesW;
    // tile order (a.tilesD, set by the launcher): 1 = column-tile major (all spatial tiles of one column tile, then the next: the weight tile is shared by every running block);
    // 2 = spatial major (the nCt column tiles of a spatial tile are neighbours in the tile order, i.e. run at the same time on the same XCD: its halo is fetched from HBM
    // once instead of once per column tile; a block still keeps its column tile from tile to tile because the persistent stride is a multiple of nCt)
    auto decode = [&](int t, int& tn, int& th0, int& tw0, int& tcol) {
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
        th0 = th * TH;
        tw0 = (r - th * a.tilesW) * TW;
        tcol = ct * BN;
    };

    const int a_off0 = (wn * WAVE_N + li) * 64 + ((lg ^ ((li >> 1) & 3)) << 4);
    int b_off0[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int px = li + kw;
        b_off0[kw] = (wm * PF * HW + px) * 64 + ((lg ^ ((px >> 1) & 3)) << 4);
    }
    int w_goff0;       // this wave's instruction of a tap tile: LDS rows 16*wave .. +15
    {
        const int slot = wave * 64 + lane;
        const int lrow = slot >> 2, pos = slot & 3;
        const int dc16 = pos ^ ((lrow >> 1) & 3);
        const int dwv = lrow / WAVE_N, j = lrow % WAVE_N;
        // MFMA row i = lg*4 + q of fragment f is output channel (f>>1)*32 + lg*8 + (f&1)*4 + q of the wave's slice: a lane holds 8 consecutive channels per fragment PAIR, and the
        // four lane groups of a pixel cover 64 contiguous bytes with ONE store instruction (pp_epilogue_plain)
        const int drow = dwv * WAVE_N + ((j >> 5) * 32) + ((j & 15) >> 2) * 8 + ((j >> 4) & 1) * 4 + (j & 3);
        w_goff0 = (drow * a.Cin + dc16 * 8) * 2;
    }
    const unsigned img_x = (unsigned)(((long long)a.H * a.W - 1) * a.x0.ld + a.Cin) * 2u;
    const char* const xb = reinterpret_cast<const char*>(a.x0.p);
    const __amdgpu_buffer_rsrc_t rw = pp_make_rsrc(a.w, (unsigned)((long long)9 * a.Cout * a.Cin * 2));

    // halo DMA j of this wave (instruction id = j*8 + wave).  The per-lane byte offsets of the HJ instructions are tile-invariant - (halo row * W + halo column) * ld +
    // swizzled chunk - and live in HJ registers: an INTERIOR tile's issue is a scalar origin + pp_dma16s (4 instructions instead of ~30); tiles on the image border keep
    // conv_ppc_kernel's per-lane recomputation with its bounds tests
    // (kept in LDS, not in registers: hipcc spills any further loop-resident VGPR of this kernel to SCRATCH, and a scratch reload waits on vmcnt - i.e. for every DMA in
    //  flight.  Table [wave][lane][8 dwords] behind the bias region, written once here; an R segment that issues halo pieces fetches its three / two offsets with one
    //  ds_read at its start - the latency hides under the weight issue or, in the group that has none, in the slack R now has)
    static_assert(HJ == 5, "");
    // (the table address of this lane is re-derived at each use - two v_mbcnt and a shift-add: held in a register it was the value hipcc spilled)
    auto tlane_now = [&]() {
        int l_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
        return (uint32_t)(uintptr_t)tbase + (uint32_t)((wave * 64 + l_) * 32);
    };
    {
        u32x4 t0;
        uint32_t t1[2];
#pragma unroll
        for (int j = 0; j < HJ; ++j) {
            const int id = j * 8 + wave;
            const int item = id * 64 + lane;
            const int p = item >> 2, pos = item & 3;
            const int py = p / HW, px = p - py * HW;
            const uint32_t v = (uint32_t)((id < HINSTR && item < HITEMS) ? ((py * a.W + px) * a.x0.ld + ((pos ^ ((px >> 1) & 3)) << 3)) * 2 : PP_OOB);
            if (j < 3) t0[j] = v;
            else t1[j - 3] = v;
        }
        t0[3] = 0u;
        *reinterpret_cast<u32x4*>(tbase + (wave * 64 + lane) * 32) = t0;
        *reinterpret_cast<u32x2*>(tbase + (wave * 64 + lane) * 32 + 16) = u32x2{t1[0], t1[1]};
    }
    // hrel = this lane's offset for piece j (from the table read of the segment)
    auto issue_halo = [&](auto jc, int hrel, int n, int h0, int w0, int c0, char* dst) {
        constexpr int j = decltype(jc)::value;
        const int id = j * 8 + wave;
        if (id >= HINSTR) return;                     // wave-uniform
        const __amdgpu_buffer_rsrc_t rx = pp_make_rsrc(xb + (size_t)n * a.H * a.W * a.x0.ld * 2, img_x);
        if (h0 >= 1 && h0 + TH + 1 <= a.H && w0 >= 1 && w0 + TW + 1 <= a.W) {          // block-uniform: every halo pixel lies inside the image
            pp_dma16s(rx, hrel, __builtin_amdgcn_readfirstlane((((h0 - 1) * a.W + (w0 - 1)) * a.x0.ld + c0) * 2), dst + id * 1024);
            return;
        }
        unsigned toff = (unsigned)((((h0 - 1) * a.W + (w0 - 1)) * a.x0.ld + c0) * 2);
        asm volatile("" : "+s"(toff));
        // the lane index is re-derived at every issue (two v_mbcnt, volatile so that it is not hoisted): with `lane` as input hipcc keeps id*64 + lane per instruction
        // across the tile loop, and at 256 VGPRs that meant SPILLING them - a scratch reload + s_waitcnt vmcnt(0) in front of two of the five halo issues of every
        // chunk, i.e. a full wait for the weight DMAs issued a few instructions earlier
        int item;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(item));
        item += id * 64;
        const int p = item >> 2, pos = item & 3;
        const int py = p / HW, px = p - py * HW;
        const unsigned rel = (unsigned)(((py * a.W + px) * a.x0.ld + ((pos ^ ((px >> 1) & 3)) << 3)) * 2);
        const bool ok = item < HITEMS && (unsigned)(h0 - 1 + py) < (unsigned)a.H && (unsigned)(w0 - 1 + px) < (unsigned)a.W;
        pp_dma16(rx, ok ? (int)(toff + rel) : PP_OOB, dst + id * 1024);
    };
    const __amdgpu_buffer_rsrc_t rb = pp_make_rsrc(a.bias != nullptr ? (const void*)a.bias : a.w, a.bias != nullptr ? (unsigned)a.Cout * 4u : 0u);
    auto issue_bias = [&](int col, char* dst) {
        if (wave < BN / 64) {
            int l;                                    // lane index re-derived (see issue_halo: a spilled `lane` here meant a scratch reload + vmcnt(0) right behind the tile's stores)
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            ppc2_dma4(rb, (col + wave * 64 + l) * 4, dst + wave * 256);
        }
    };
    // the three tap tiles (kh = 0..2) of filter column kw, column tile col, channels c0..c0+31: one instruction per tap per wave
    auto issue_weights = [&](int kw, int col, int c0, char* dst) {
        int soff = (int)((((long long)kw * a.Cout + col) * a.Cin + c0) * 2);
        asm volatile("" : "+s"(soff));
        const int tapstride = 3 * a.Cout * a.Cin * 2;          // tap index = kh*3 + kw
        if (wave >= WWAVES) return;                    // wave-uniform (NF = 2: the four waves of group 0 carry the whole tile)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) pp_dma16s(rw, w_goff0, soff + kh * tapstride, dst + kh * TAPB + wave * 1024);
    };

    int n, h0, w0, ncol0;
    decode(tile, n, h0, w0, ncol0);
    f32x4 acc[NF][PF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) acc[f][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue_bias(ncol0, bbase);
    issue_weights(0, ncol0, 0, wbase);
    // (both groups fetch their part of the NEXT segment's weight tile during the current one: group 1 at the start of its R, group 0 at the start of its M - the tile of
    //  the current segment stays live through M here, so nobody may refill it before the slot after that)
    {
        // (the table entries of this lane were written by this lane: no barrier needed)
        const u32x4 q0 = *reinterpret_cast<const u32x4*>(tbase + (wave * 64 + lane) * 32);
        const u32x2 q1 = *reinterpret_cast<const u32x2*>(tbase + (wave * 64 + lane) * 32 + 16);
        const uint32_t hv[5] = {q0[0], q0[1], q0[2], q1[0], q1[1]};
        pp_static_for<HJ>([&](auto jc) { issue_halo(jc, (int)hv[decltype(jc)::value], n, h0, w0, 0, hbase); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int wsel = 0, hsel = 0, bsel = 0;
#ifdef MIS_PP_STAMPS          // (diagnostic build: scripts/ppc_stamps.sh; slots as in conv_pp_kernel)
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = __builtin_amdgcn_s_memtime();
#endif
    if (grp == 1) __builtin_amdgcn_s_barrier();       // the stagger
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (; tile < total_tiles; tile += tstride) {
        const bool has_next = tile + tstride < total_tiles;
        int nn = n, nh0 = h0, nw0 = w0, ncolN = ncol0;
        if (has_next) decode(tile + tstride, nn, nh0, nw0, ncolN);
        // EM == PP_EM_BITS: the mask of this tile's epilogue - one 16-byte (8-byte) load per lane, issued here, ahead of the whole K loop (inside the chunk loop the
        // load's destination would be loop-carried and hipcc guards it with an s_waitcnt vmcnt(0) at the loop header, i.e. in front of every chunk's prefetches)
        u32x4 mbits = u32x4{0u, 0u, 0u, 0u};
        if constexpr (EM == PP_EM_BITS) mbits = pp_mask_bits_load<NF, PF>(a, n, h0, w0, ncol0, wm, wn);
        // the next tile's bias slice is read at that tile's START (first segment): fetched now, into the half of the bias region this tile's first segment is done with
        // by the time the DMA lands (issued behind that segment's barrier pair at the earliest - see the chunk-0 code)
        // The K loop: chunk 0 runs a copy of the segment code whose first filter row's MFMAs take the BIAS as their C operand (`first`), so that the accumulators are
        // neither zeroed nor biased by vector instructions anywhere (pp_epilogue_plain<..., BINIT>).
        auto run_chunk = [&](auto firstc, const int chunk) __attribute__((always_inline)) {
            constexpr bool first = decltype(firstc)::value;
            const int c0 = chunk << 5;
            const bool last_chunk = chunk + 1 == nchunks;
            const bool hnext = !last_chunk || has_next;
            const int hn = last_chunk ? nn : n, hh0 = last_chunk ? nh0 : h0, hw0 = last_chunk ? nw0 : w0, hc0 = last_chunk ? 0 : c0 + 32;
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
                const bool w2next = (kw < 1) || hnext;
                const int w2kw = (kw + 2) % 3;
                const int w2col = (kw < 1 || !last_chunk) ? ncol0 : ncolN;
                const int w2c0 = (kw < 1) ? c0 : hc0;
                // ================= R segment =================
#ifndef PPT_NO_DMA          // (PPT_NO_*: timing ablations of a diagnostic build, scripts/ppt_ablate.sh - results are garbage, never shipped)
                constexpr int NH = kw == 0 ? HJ0 : (kw == 1 ? HJ1 : 0);          // halo instructions issued in this segment (per wave; the last may be past the image)
                u32x4 hq = u32x4{0u, 0u, 0u, 0u};
                if constexpr (NH > 0) {
                    if (hnext) {
                        if constexpr (kw == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(hq) : "v"(tlane_now()));
                        else {
                            u32x2 h2;
                            asm volatile("ds_read_b64 %0, %1 offset:16" : "=v"(h2) : "v"(tlane_now()));
                            hq[0] = h2[0];
                            hq[1] = h2[1];
                        }
                    }
                }
                if (grp == 1 && wnext) issue_weights(wkw, wcol, wc0, wbn);
#endif
#ifndef PPT_NO_DMA
                if constexpr (NH > 0) {
                    if (hnext) {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hq)::"memory");
                        pp_static_for<NH>([&](auto jc) {
                            issue_halo(std::integral_constant<int, (kw == 0 ? 0 : HJ0) + decltype(jc)::value>{}, (int)hq[decltype(jc)::value], hn, hh0, hw0, hc0, hbn);
                        });
                    }
                }
#endif
                PP_STAMP(5)
                u32x4 As[5], Brow[PF + 2];          // weight fragments t = kh * 4 + f live in slot t % 5: five are read here, seven roll through the slots inside M
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
                pp_static_for<5>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    As[t] = pp_lds_read128<(t / 4) * TAPB + (t % 4) * 1024>(wb + a_off0);
                });
                pp_static_for<PF + 2>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    Brow[r] = pp_lds_read128<r * ROWB>(hb + b_off0[kw]);
                });
                PP_STAMP(6)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                PP_STAMP(7)
                // how many of this wave's youngest DMAs may stay in flight: the halo instructions of THIS segment (a wave whose last instruction id is past the image
                // issued one fewer - waiting for one more than necessary is harmless, so the count is the compile-time maximum only when it is exact)
                constexpr int KEEP = NH;
                if (grp == 1) {                                  // group 1: its weight DMAs for the next segment (issued one slot pair ago, in its M) must have landed
                    if (hnext && KEEP > 0) {
                        // wave-uniform: did this wave really issue KEEP halo instructions?
                        const int last_id = ((kw == 0 ? 0 : HJ0) + KEEP - 1) * 8 + wave;
                        if (last_id < HINSTR) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP > 0 ? KEEP - 1 : 0) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                PP_STAMP(0)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                PP_STAMP(1)
                // ================= M segment: 3 taps x NF x PF MFMAs =================
                #ifndef PPT_NO_PRIO
                __builtin_amdgcn_s_setprio(1);
#endif
                pp_static_for<12>([&](auto tc) {
                    constexpr int t = decltype(tc)::value, kh = t / 4, f = t % 4;
                    // fragment t was read into slot t % 5 behind group t - 5: everything but the reads issued after it (at most four) must have returned
                    if constexpr (t >= 5) {
                        constexpr int left = (11 - t) < 4 ? (11 - t) : 4;
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(As[t % 5]) : "n"(left) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf) {
                        if constexpr (first && kw == 0 && kh == 0) acc[f][pf] = bq[f];          // (folds into the MFMA's C operand)
                        mma_b128<T>(acc[f][pf], As[t % 5], Brow[pf + kh]);
                    }
                    if constexpr (t == 0) {
                        if (grp == 0 && wnext) issue_weights(wkw, wcol, wc0, wbn);
                    }
                    if constexpr (t + 5 < 12) {
                        // IN PLACE ("+v": the read's destination is the slot's own register tuple - left to itself the allocator gave the late fragments fresh registers,
                        // 36 instead of 20 for the weight fragments) and behind the group that last used the slot (the tie is also the write-after-read dependency)
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(As[t % 5]) : "v"(wb + a_off0), "n"(((t + 5) / 4) * TAPB + ((t + 5) % 4) * 1024));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                __builtin_amdgcn_s_setprio(0);
                // group 0: its part of the next segment's weight tile, issued behind this M's first MFMA group, is its YOUNGEST DMA: everything it has in flight
                // (the halo pieces of this segment's R are nearly two slots old) lands here
                if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                PP_STAMP(2)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                PP_STAMP(3)
                wsel ^= 1;
            });
            hsel ^= 1;
        };
        run_chunk(std::true_type{}, 0);
        // the next tile's bias slice -> the other half of the bias region (last read in the previous tile's chunk 0); it lands under the rest of this tile's K loop and is
        // read in the next tile's first segment (nchunks >= 2: at least three segments and their counted waits / barriers in between)
        if (has_next) issue_bias(ncolN, bbase + (bsel ^ 1) * (BN * 4));
#pragma unroll 1
        for (int chunk = 1; chunk < nchunks; ++chunk) run_chunk(std::false_type{}, chunk);
#if !defined(PPT_NO_EPI)
#ifdef PPT_EPI_PRIO
        __builtin_amdgcn_s_setprio(PPT_EPI_PRIO);
#endif
        pp_epilogue_plain<NF, PF, EM, true, true>(a, acc, 0u, n, h0, w0, ncol0, wm, wn, mbits);
#ifdef PPT_EPI_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#else                       // diagnostic build: no epilogue at all (the accumulators run on; they are summed into one conditional store after the tile loop)
#endif
        n = nn; h0 = nh0; w0 = nw0; ncol0 = ncolN;
        bsel ^= 1;
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(4)
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();       // pairs with group 1's last barrier
#ifdef MIS_PP_STAMPS
    if (lane == 0 && blockIdx.x < 256) {
        for (int i = 0; i < 8; ++i) g_ppc2_stamps[(blockIdx.x * 8 + wave) * 8 + i] = st_[i];
    }
#endif
#ifdef PPT_NO_EPI
    {
        f32x4 s4 = acc[0][0];
        pp_static_for<NF>([&](auto fc) { pp_static_for<PF>([&](auto pc) { s4 += acc[decltype(fc)::value][decltype(pc)::value]; }); });
        if (s4[0] == 123.456f) *reinterpret_cast<f32x4*>(a.y0) = s4;
    }
#endif
}


bool conv_ppc2_eligible(const MisConvDesc* d) {
    if (!mis_sw(SW_CONV_PPC2)) return false;
    if (!conv_pp_eligible(d) || conv_ppc_choice(d) != 4) return false;
    return true;
}

template <int EM> static int ppc2_launch_em(const MisConvDesc* d, hipStream_t stream) {
    constexpr int PF = 8, NF = 4, BN = 2 * NF * 16;
    constexpr int TH = 4 * PF, HINSTR = ((TH + 2) * 18 * 4 + 63) / 64;
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
    a.tilesH = (d->H + TH - 1) / TH;
    a.tilesW = (d->W + 15) / 16;
    const long long nsp = (long long)d->N * a.tilesH * a.tilesW;
    a.nCt = d->Cout / BN;
    a.tilesD = (256 % a.nCt == 0 && !mis_sw(SW_CONV_PPC_COLMAJOR)) ? 2 : 1;
    MIS_REQUIRE(nsp * a.nCt < (1ll << 31), MIS_EUNSUPPORTED, "conv_igemm(ppc2): grid too large");
    a.nSp = (int)nsp;
    const size_t lds = 2 * (size_t)HINSTR * 1024 + 2 * (size_t)3 * BN * 64 + 2 * (size_t)BN * 4 + 8 * 64 * 32;
    static std::atomic<unsigned long long> attr_done{0};
    if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&conv_ppc2_kernel<EM>), lds, "conv_igemm(ppc2)")) return rc;
    const long long total = nsp * a.nCt;
    hipLaunchKernelGGL(conv_ppc2_kernel<EM>, dim3((unsigned)(total > mis_persist_cus() ? mis_persist_cus() : total)), dim3(512), lds, stream, a);
    MIS_LAUNCH_CHECK("conv_igemm(ppc2)");
    return MIS_OK;
}

int launch_conv_ppc2(const MisConvDesc* d, hipStream_t stream, const char** tag) {
    if (d->mask_bits != nullptr) {
        *tag = "k3.2d.ppc2.bits";
        return ppc2_launch_em<PP_EM_BITS>(d, stream);
    }
    if (d->mask != nullptr) {
        *tag = "k3.2d.ppc2.mask";
        return ppc2_launch_em<PP_EM_MASK>(d, stream);
    }
    *tag = "k3.2d.ppc2";
    return ppc2_launch_em<PP_EM_NONE>(d, stream);
}
