// Ping-pong implicit-GEMM 3x3 convolution for the bf16 2-D layers (gfx950): forward and dgrad of nn.Conv2d(k3, p1) with Cout % 128 == 0
// (reference model/unet2d/layers.py:122-126), same arithmetic / operand packing / epilogues as conv_igemm.hip.
//
// Why another kernel: conv_igemm's 8 waves run in lockstep - all of them read fragments after a barrier, then all of them want the matrix pipe
// (PMC: MFMA pipe 36-50 % busy, 40 % of wave cycles parked at waitcnt / barrier).  Here the two waves that share a SIMD (wave w and w + 4) are
// staggered by one barrier and alternate strictly:
//
//       slot      0      1      2      3      4   ...          (a slot ends with an s_barrier of all 8 waves)
//       G0 (w<4)  R0     M0     R1     M1     R2               R = fragment ds_reads (+ LDS-DMA issue), M = 32 MFMAs
//       G1 (w>=4) -      R0     M0     R1     M1
//
// so that one wave per SIMD is always inside an MFMA cluster while its partner fetches the next fragments (the structure of the guide's 8-phase
// GEMM template, cdna_hip_programming.md §5 / T3+T4).  Nothing is staged through registers: the input HALO tile (18 x 18 pixels x 64 channels,
// once per K chunk, double-buffered) and the per-tap WEIGHT tile ([BN x 64 ch], double-buffered) both arrive by LDS-DMA (global_load_lds_dwordx4)
// with the XOR swizzle applied to the per-lane SOURCE address; zero padding comes from a zero page in global memory; prefetches stay in flight
// across barriers and are retired with counted s_waitcnt vmcnt.  Blocks are persistent (grid <= 256): the next tile's halo and first weight
// tile are fetched under the current tile's last taps, and a tile's epilogue runs in the R slot of the next tile's first step, i.e. under the
// partner group's MFMAs.
//
// LDS-DMA ordering rules used below (MI355X_MICROARCH.md, "Two waves per SIMD" item 7; cdna_hip_programming.md "Read a staged buffer one phase
// AFTER the wait that retires it"): a buffer is re-filled only after every wave's last ds_read of it has been waited for (lgkmcnt(0) before the
// slot's barrier) and a barrier passed; it is read only after EVERY issuing wave's vmcnt wait and a following barrier.
//   weights of step s+1 -> buffer (s+1)&1: last read (by G1) in the slot before step s starts; issued in the first R of step s; retired by each
//                          issuer at the end of step s (G0: after its last M, G1: after its last R - both in the last slot of step s).
//   halo of chunk c+1  -> buffer (c+1)&1: last read in chunk c-1; issued one instruction per wave in steps 0..5 of chunk c (the YOUNGEST
//                          outstanding op of the wave, so the per-step weight wait is vmcnt(1) in those steps); retired by vmcnt(0) in steps 6..8.
#include <stdlib.h>

#include "conv_pp_common.hpp"
#include "dispatch_cfg.hpp"

namespace {
constexpr int PP_TH = 16, PP_TW = 16, PP_HH = 18, PP_HW = 18, PP_HP = PP_HH * PP_HW;
constexpr int PP_HITEMS = PP_HP * 8;                 // 16-byte items of one halo chunk image (128 B per pixel)
constexpr int PP_HINSTR = (PP_HITEMS + 63) / 64;     // 41 wave-instructions (the last one half full: the image is padded to 41 KiB)
constexpr int PP_HBUF = PP_HINSTR * 1024;
constexpr int PP_ROWB = PP_HW * 128;                 // bytes per halo row

}   // namespace

// Diagnostic build (-DMIS_PP_STAMPS, never shipped): per wave, shader cycles spent (0) working in R segments, (1) parked at the barrier that ends an R segment,
// (2) working in M segments, (3) parked at the barrier that ends an M segment, (4) in the tile-end epilogue; R work is split further: (5) DMA issue incl. its scalar
// bookkeeping, (6) issuing the fragment reads, (7) waiting for them (lgkmcnt), (0) the rest (vmcnt drain); read back with mis_debug_pp_stamps().
#ifdef MIS_PP_STAMPS
__device__ unsigned long long g_pp_stamps[256 * 8 * 8];
extern "C" int mis_debug_pp_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pp_stamps), sizeof(unsigned long long) * 256 * 8 * 8) == hipSuccess ? 0 : -1;
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

// NF = 16-channel fragments per wave along the output channels: 8 -> 256-column blocks (wave tile 64 px x 128 ch, one segment = one 32-channel
// k-group), 4 -> 128-column blocks (wave tile 64 px x 64 ch, one segment = a whole 64-channel tap): 32 MFMAs per segment either way;
// 2 -> 64-column blocks (wave tile 64 px x 32 ch, one segment = a tap = 16 MFMAs) for the layers with 64 output channels.
template <int NF>
__global__ __launch_bounds__(512, 2) void conv_pp_kernel(const ConvArgs a) {
    using T = __bf16;
    constexpr int PF = 4, WAVE_N = NF * 16, BN = 2 * WAVE_N, NV = 4 * NF;
    constexpr int WTILE = BN * 128;                  // bytes of one tap's weight tile (BN rows x 64 channels)
    constexpr int SEGK = (NF == 8) ? 1 : 2;          // 32-channel k-groups per segment
    constexpr int SPS = 2 / SEGK;                    // segments per step (step = one tap of one 64-channel chunk)
    // Weight DMA: BN/8 instructions per step, WPW per wave (all 8 waves issue).  Issuing is the expensive part of an R segment (~100-185 cycles per
    // instruction next to ds_reads, ~60 inside an MFMA cluster), so the instructions are spread over the slots in which their buffer is free:
    //   NF = 8 (4 per wave, 4 slots per step): 2 in R0, 2 inside the following M0 - both groups, for the NEXT step;
    //   NF = 4 (2 per wave, 2 slots per step): group 0 in its R (next step), group 1 inside its M - for the step AFTER the next one: group 1's M of step s
    //          runs in the first slot of group 0's step s+1, which is when the buffer of step s (= that of step s+2) falls free.
    constexpr int WPW = (BN / 8) / 8;
    static_assert(WPW == 4 || WPW == 2 || WPW == 1, "");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hbase = smem;                        // 2 x PP_HBUF
    char* const wbase = smem + 2 * PP_HBUF;          // 2 x WTILE
    char* const bbase = wbase + 2 * WTILE;           // 2 x BN floats: the bias slice of the current / next tile's column tile (by tile parity)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    const int li = lane & 15, lg = lane >> 4;

    const int total_tiles = a.nSp * a.nCt;
    const int tstride = (int)gridDim.x;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= total_tiles) return;                 // block-uniform (the launcher never over-sizes the grid)
    const int nchunks = a.Cin >> 6;
    const int tpi = a.tilesH * a.tilesW;
    // cout-tile major, spatial minor: concurrently running blocks share the weight tile in L2, neighbours share halo rows
    auto decode = [&](int t, int& tn, int& th0, int& tw0, int& tcol) {
        const int ct = t / a.nSp;
        const int sp = t - ct * a.nSp;
        tn = sp / tpi;
        const int r = sp - tn * tpi;
        const int th = r / a.tilesW;
        th0 = th * PP_TH;
        tw0 = (r - th * a.tilesW) * PP_TW;
        tcol = ct * BN;
    };

    // ---- per-lane fragment offsets (the 16-byte chunk position is XORed with (row & 7) for weights, (halo column & 7) for pixels) ----
    // k-group 1 (channels 32..63) sits at chunk positions 4..7: chunk bit 2 is untouched by the XOR's low... it is XORed too, so the second k-group's
    // offset is the first one's with byte-offset bit 6 flipped - one v_xor per use instead of a second set of live registers
    int a_off0, b_off0[3];
    a_off0 = (wn * WAVE_N + li) * 128 + ((lg ^ (li & 7)) << 4);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int px = li + kw;
        b_off0[kw] = (wm * 4 * PP_HW + px) * 128 + ((lg ^ (px & 7)) << 4);
    }
    // ---- per-lane DMA source offsets.  An instruction fills 1 KiB of LDS linearly (lane l -> 16-byte slot 64*q + l): which (row, chunk) must land
    //      in a slot follows from inverting the read-side swizzle and, for weights, the row permutation that gives a lane NV consecutive channels.
    //      Weight instruction q = q0 + k (q0 = 4 * issuer index) covers LDS rows 8q .. 8q+7: its source row is that of k = 0 plus a lane-independent
    //      (2*(k&1)*NV + 4*(k>>1)) rows, which goes into the scalar part of the offset ----
    int w_goff0;
    {
        const int q = wave * WPW;
        const int slot = q * 64 + lane;
        const int lrow = slot >> 3, pos = slot & 7;
        const int dc16 = pos ^ (lrow & 7);
        const int dwv = lrow / WAVE_N, j = lrow % WAVE_N;
        const int drow = dwv * WAVE_N + ((j & 15) >> 2) * NV + (j >> 4) * 4 + (j & 3);
        w_goff0 = (drow * a.Cin + dc16 * 8) * 2;          // bytes from the first weight of (tap, column tile, K chunk)
    }
    // the input's buffer resource is rebuilt PER IMAGE (scalar ops only): 32-bit offsets then only have to span one image, whatever the batch size
    const unsigned img_x = (unsigned)(((long long)a.H * a.W - 1) * a.x0.ld + a.Cin) * 2u;
    const char* const xb = reinterpret_cast<const char*>(a.x0.p);
    const __amdgpu_buffer_rsrc_t rw = pp_make_rsrc(a.w, (unsigned)((long long)9 * a.Cout * a.Cin * 2));

    // Every instruction of an R segment costs the wave ~10 cycles (it shares the SIMD's issue port with its partner's MFMA cluster), so a halo DMA should be a handful of
    // instructions: where registers allow (128- / 64-column variants) the per-lane source offset and halo coordinates of the wave's six halo instructions are computed
    // ONCE; the 256-column variant (252 VGPRs) recomputes them at every issue instead.
    constexpr bool HPRE = (NF != 8);
    unsigned h_rel[HPRE ? 6 : 1];
    int h_coord[HPRE ? 6 : 1];                    // halo row | halo column << 8; -1: no item (tail of the last instruction / no such instruction)
    if constexpr (HPRE) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int id = j * 8 + wave;
            const int item = id * 64 + lane;
            const int p = item >> 3, pos = item & 7;
            const int py = p / PP_HW, px = p - py * PP_HW;
            h_rel[j] = (unsigned)(((py * a.W + px) * a.x0.ld + ((pos ^ (px & 7)) << 3)) * 2);
            h_coord[j] = (id < PP_HINSTR && item < PP_HITEMS) ? (py | (px << 8)) : -1;
        }
    }
    // one halo DMA of this wave: instruction id = j*8 + wave of the chunk image (n, h0, w0, channels c0..c0+63) into `dst`; false if id is past the image
    auto issue_halo = [&](auto jc, int n, int h0, int w0, int c0, char* dst) -> bool {
        constexpr int j = decltype(jc)::value;
        const int id = j * 8 + wave;
        if (id >= PP_HINSTR) return false;            // wave-uniform
        const __amdgpu_buffer_rsrc_t rx = pp_make_rsrc(xb + (size_t)n * a.H * a.W * a.x0.ld * 2, img_x);
        // byte offset of halo pixel (0, 0) = image pixel (h0 - 1, w0 - 1) within the image: "negative" (wraps) for tiles on the top / left border, where only valid items add to it
        unsigned toff = (unsigned)((((h0 - 1) * a.W + (w0 - 1)) * a.x0.ld + c0) * 2);
        if constexpr (HPRE) {
            const bool interior = h0 >= 1 && h0 + PP_TH + 1 <= a.H && w0 >= 1 && w0 + PP_TW + 1 <= a.W;        // block-uniform
            bool ok = h_coord[j] >= 0;
            if (!interior) {
                const int py = h_coord[j] & 0xff, px = h_coord[j] >> 8;
                ok = ok && (unsigned)(h0 - 1 + py) < (unsigned)a.H && (unsigned)(w0 - 1 + px) < (unsigned)a.W;
            }
            pp_dma16(rx, ok ? (int)(toff + h_rel[j]) : PP_OOB, dst + id * 1024);
        } else {
            asm volatile("" : "+s"(toff));                // opaque: keeps hipcc from pre-computing (and spilling) the offsets of a whole chunk's steps
            // lane -> 16-byte slot of the image -> (halo pixel, channel chunk): the chunk position in LDS is XORed with (halo column & 7)
            int item = id * 64 + lane;
            asm volatile("" : "+v"(item));                // recomputed at every issue instead of living in registers across the tile loop
            const int p = item >> 3, pos = item & 7;
            const int py = p / PP_HW, px = p - py * PP_HW;
            const unsigned rel = (unsigned)(((py * a.W + px) * a.x0.ld + ((pos ^ (px & 7)) << 3)) * 2);
            const bool ok = item < PP_HITEMS && (unsigned)(h0 - 1 + py) < (unsigned)a.H && (unsigned)(w0 - 1 + px) < (unsigned)a.W;
            pp_dma16(rx, ok ? (int)(toff + rel) : PP_OOB, dst + id * 1024);
        }
        return true;
    };
    // bias slice of a column tile -> LDS, 4 bytes per lane (waves 0 .. BN/64-1; without a bias the zero-sized buffer reads as zeros)
    const __amdgpu_buffer_rsrc_t rb = pp_make_rsrc(a.bias != nullptr ? (const void*)a.bias : a.w, a.bias != nullptr ? (unsigned)a.Cout * 4u : 0u);
    auto issue_bias = [&](int col, char* dst) {
        if (wave < BN / 64)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (pp_lds_void_t*)(dst + wave * 256), 4, (col + wave * 64 + lane) * 4, 0, 0, 0);
    };
    // instructions [k0, k1) of this wave's WPW for the weight tile (tap, column tile col, K chunk c0)
    auto issue_weights = [&](auto k0c, auto k1c, int tap, int col, int c0, char* dst) {
        constexpr int k0 = decltype(k0c)::value, k1 = decltype(k1c)::value;
        int soff = (int)((((long long)tap * a.Cout + col) * a.Cin + c0) * 2);
        asm volatile("" : "+s"(soff));
        const int q0 = wave * WPW;
#pragma unroll
        for (int k = k0; k < k1; ++k) pp_dma16(rw, (soff + (2 * (k & 1) * NV + 4 * (k >> 1)) * a.Cin * 2) + w_goff0, dst + (q0 + k) * 1024);
    };
    using I0 = std::integral_constant<int, 0>;
    using IH = std::integral_constant<int, WPW / 2>;
    using IW = std::integral_constant<int, WPW>;

    int n, h0, w0, ncol0;
    decode(tile, n, h0, w0, ncol0);
    f32x4 acc[NF][PF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) acc[f][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue_bias(ncol0, bbase);
    // ---- prologue: first weight tile and first halo chunk ----
    issue_weights(I0{}, IW{}, 0, ncol0, 0, wbase);
    if constexpr (NF != 8) {
        if (grp == 1) issue_weights(I0{}, IW{}, 3, ncol0, 0, wbase + WTILE);      // group 1's share of step 1 = tap (kh 1, kw 0) (in the loop it issues two steps ahead)
    }
    pp_static_for<6>([&](auto jc) { (void)issue_halo(jc, n, h0, w0, 0, hbase); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int wsel = 0, hsel = 0, bsel = 0;
#ifdef MIS_PP_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = __builtin_amdgcn_s_memtime();
#endif
    if (grp == 1) __builtin_amdgcn_s_barrier();       // the stagger: group 1 runs one slot behind group 0
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (; tile < total_tiles; tile += tstride) {
        const bool has_next = tile + tstride < total_tiles;
        int nn = n, nh0 = h0, nw0 = w0, ncolN = ncol0;
        if (has_next) decode(tile + tstride, nn, nh0, nw0, ncolN);
#pragma unroll 1
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int c0 = chunk << 6;
            const bool last_chunk = chunk + 1 == nchunks;
            // what the halo prefetch of this chunk's steps fetches: the next chunk of this tile, else chunk 0 of the next tile, else nothing
            const bool hnext = !last_chunk || has_next;
            const int hn = last_chunk ? nn : n, hh0 = last_chunk ? nh0 : h0, hw0 = last_chunk ? nw0 : w0, hc0 = last_chunk ? 0 : c0 + 64;
            const uint32_t hb = (uint32_t)(uintptr_t)hbase + hsel * PP_HBUF;      // LDS byte addresses (asm reads)
            char* hbn = hbase + (hsel ^ 1) * PP_HBUF;
            // Taps are walked COLUMN-major (kw outer, kh inner): the pixel fragment of tile row r for tap (kh, kw) is the fragment of row r + 1 for tap (kh - 1, kw), so
            // in the 128- / 64-column variants the three taps of a filter column share six row fragments held in registers (4 + 1 + 1 reads instead of 3 x 4 per
            // k-group: a quarter fewer ds_reads in the variant that is bound by its R segments).  The 256-column variant has no registers to spare and re-reads.
            constexpr bool BREUSE = (NF != 8);
            u32x4 Brow[BREUSE ? SEGK : 1][BREUSE ? 6 : 1];
            pp_static_for<9>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int kw = t / 3, kh = t % 3;
                auto tapidx = [](int tt) { return (tt % 3) * 3 + tt / 3; };       // loop position -> tap index (kh * 3 + kw) of the packed weights
                const uint32_t wb = (uint32_t)(uintptr_t)wbase + wsel * WTILE;
                char* wbn = wbase + (wsel ^ 1) * WTILE;
                // the step after this one: next tap; after the last tap the next chunk's (or the next tile's) first tap
                const bool wnext = (t < 8) || hnext;
                const int wtap = (t < 8) ? tapidx(t + 1) : 0;
                const int wcol = (t < 8 || !last_chunk) ? ncol0 : ncolN;
                const int wc0 = (t < 8) ? c0 : hc0;
                // ... and the one after that (NF = 4, group 1)
                const bool w2next = (t < 7) || hnext;
                const int w2tap = tapidx((t + 2) % 9);
                const int w2col = (t < 7 || !last_chunk) ? ncol0 : ncolN;
                const int w2c0 = (t < 7) ? c0 : hc0;
                char* wb_self = wbase + wsel * WTILE;     // the buffer this step reads = the buffer of the step after the next one
                bool hi = false;
                pp_static_for<SPS>([&](auto sc) {
                    constexpr int sg = decltype(sc)::value;
                    // ================= R segment =================
                    if constexpr (sg == 0) {
                        if constexpr (NF == 8) {
                            if (wnext) issue_weights(I0{}, IH{}, wtap, wcol, wc0, wbn);
                        } else {
                            if (grp == 0 && wnext) issue_weights(I0{}, IW{}, wtap, wcol, wc0, wbn);
                        }
                    }
                    if constexpr (sg == SPS - 1 && t < 6) {
                        if (hnext) hi = issue_halo(std::integral_constant<int, t>{}, hn, hh0, hw0, hc0, hbn);
                    }
                    PP_STAMP(5)
                    u32x4 A[SEGK][NF], B[SEGK][PF];
                    pp_static_for<SEGK>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        constexpr int kg = sg * SEGK + s;
                        pp_static_for<NF>([&](auto fc) {
                            constexpr int f = decltype(fc)::value;
                            A[s][f] = pp_lds_read128<f * 2048>(wb + (a_off0 ^ (kg << 6)));
                        });
                        if constexpr (BREUSE) {
                            // rows needed by this tap: kh .. kh + 3; rows 0..3 are read at kh == 0, row 4 at kh == 1, row 5 at kh == 2
                            pp_static_for<6>([&](auto rc) {
                                constexpr int r = decltype(rc)::value;
                                if constexpr ((kh == 0 && r < 4) || (kh == 1 && r == 4) || (kh == 2 && r == 5))
                                    Brow[s][r] = pp_lds_read128<r * PP_ROWB>(hb + (b_off0[kw] ^ (kg << 6)));
                            });
                        } else {
                            pp_static_for<PF>([&](auto pc) {
                                constexpr int pf = decltype(pc)::value;
                                B[s][pf] = pp_lds_read128<(pf + kh) * PP_ROWB>(hb + (b_off0[kw] ^ (kg << 6)));
                            });
                        }
                    });
                    PP_STAMP(6)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // fragments in registers; also: this wave's reads of both buffers are over
                    __builtin_amdgcn_sched_barrier(0);                      // no consumer may move above the wait (the asm reads are opaque to hipcc)
                    PP_STAMP(7)
                    if constexpr (sg == SPS - 1) {
                        // group 1's last slot of the step: its weight DMAs for the next step (issued one slot-pair ago) must have landed; a halo DMA issued in this
                        // segment is the youngest op and stays in flight
                        if (grp == 1) {
                            if (hi) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    PP_STAMP(0)
                    __builtin_amdgcn_s_barrier();
                    PP_STAMP(1)
                    __builtin_amdgcn_sched_barrier(0);
                    // ================= M segment =================
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int s = 0; s < SEGK; ++s)
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
#pragma unroll
                            for (int pf = 0; pf < PF; ++pf) mma_b128<T>(acc[f][pf], A[s][f], BREUSE ? Brow[s][pf + kh] : B[s][pf]);
                            if (s == 0 && f == 1) {      // the second half of the weight DMAs goes out from inside the MFMA cluster
                                if constexpr (NF == 8 && sg == 0) {
                                    if (wnext) issue_weights(IH{}, IW{}, wtap, wcol, wc0, wbn);
                                }
                                if constexpr (NF != 8) {
                                    if (grp == 1 && w2next) issue_weights(I0{}, IW{}, w2tap, w2col, w2c0, wb_self);
                                }
                            }
                        }
                    __builtin_amdgcn_s_setprio(0);
                    if constexpr (sg == SPS - 1) {
                        if (grp == 0) {                                      // group 0's last slot of the step
                            if (hi) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    PP_STAMP(2)
                    __builtin_amdgcn_s_barrier();
                    PP_STAMP(3)
                    __builtin_amdgcn_sched_barrier(0);
                });
                wsel ^= 1;
            });
            hsel ^= 1;
        }
        // this wave's tile is complete: store it and re-arm the accumulators.  In program order this sits at the head of the wave's next R slot, i.e. it runs
        // under the partner group's MFMAs (group 1 is still in its last M slot of the tile when group 0 gets here, and vice versa one slot later).
        // The next tile's bias slice goes to the OTHER half of the bias region (last read one tile ago); it is retired by the step drains long before its epilogue.
        pp_epilogue<NF>(a, acc, (uint32_t)(uintptr_t)bbase + bsel * (BN * 4), n, h0, w0, ncol0, wm, wn, li, lg);
        n = nn; h0 = nh0; w0 = nw0; ncol0 = ncolN;
        bsel ^= 1;
        if (has_next) issue_bias(ncol0, bbase + bsel * (BN * 4));
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(4)
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();       // pairs with group 1's last barrier
#ifdef MIS_PP_STAMPS
    if (lane == 0 && blockIdx.x < 256) {
        for (int i = 0; i < 8; ++i) g_pp_stamps[(blockIdx.x * 8 + wave) * 8 + i] = st_[i];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// Column-segment ping-pong kernel (end of round 2; the default for Cout % 128 == 0): 512 pixels (32 rows x 16 columns) x 128 channels per block, wave tile
// 128 px x 64 ch (PF = 8 pixel-row fragments x NF = 4 channel fragments), K chunks of 32 channels = 64 bytes per pixel / per weight row in LDS - that is what lets two
// 34 x 18-pixel halo buffers (2 x 39 KiB) and two 3-tap weight buffers (2 x 24 KiB) fit.  One segment = one FILTER COLUMN (3 taps) of one chunk = 3 x 4 x 8 = 96 MFMAs
// between two barriers, fed by 12 weight fragments + 10 pixel-row fragments (the three taps of a column shift the row: PF + 2 rows serve all of them): 0.23 ds_read_b128 per
// MFMA against 0.375 in conv_pp_kernel, 3 weight + at most 3 halo DMAs per wave per segment, and a third of the barriers per MFMA.  Same stagger, DMA ordering rules and
// epilogue as conv_pp_kernel.  Measured against conv_pp_kernel<8 / 4> in one process: in isolation on N(0,1) operands (scripts/bench_conv_layers.py) every 3x3 layer of the
// benchmark net is 3-12 % faster (18.07 vs 19.39 ms over the forward launches of a step; deep layers 1.59-1.64 against 1.49-1.55 PFLOP/s); inside the live train step
// (bench.py --layers, post-ReLU operands, half of the launches in the masked dgrad form) every layer is equal or faster, 14.35 vs 14.87 ms per step over the 28 launches -
// of which the step keeps 0.2 ms (36.96 vs 37.16): the rest comes back as lower clocks elsewhere (MI355X_MICROARCH.md, DVFS give-back).  With the first swizzle guess (2-way
// LDS bank conflicts, see below) it was only level with conv_pp_kernel in the live step.  What did NOT matter (diagnostic builds, scripts/ppt_ablate.sh): one tap per segment on the same tile
// (+6 % over conv_pp_kernel<4>, par with <8>: fragment reads per MFMA are not the limiter any more), a third weight buffer with two segments of prefetch distance (-0.5 %:
// not DMA latency), 32x32x16 MFMAs at the same pipe time (-2 %: not the vector issue port); removing the DMA issue altogether: +15 %; the read / DMA / barrier path alone
// takes 57-67 % of the kernel's time and overlaps the matrix pipe only partly.
// LDS swizzle for 64-byte rows: the 16-byte chunk position is XORed with (column >> 1) & 3 (pixels) / (row >> 1) & 3 (weights).  ds_read_b128 is served in four groups of 16
// lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) - and this is the function (found by enumeration over those groups, every tap
// offset and row alignment) that gives each group 16 distinct 16-byte bank quads; the first guess, (column >> 2) & 3, measured 2-way conflicts (SQ_LDS_BANK_CONFLICT = 47 % of the
// LDS cycles).
// DMA schedule per wave: weights 3 instructions per segment (group 0 in its R for the next segment, group 1 in its M for the segment after the next one, as in conv_pp_kernel);
// halo of the next chunk: instructions j = 0..2 in the R of column 0, j = 3.. in the R of column 1, none in column 2 (so that everything has had a slot pair to land before the
// chunk's last barrier); every wait is a counted vmcnt that leaves exactly the halo instructions issued in the same segment in flight.
template <int PF, int NF, int EM>
__global__ __launch_bounds__(512, 2) void conv_ppc_kernel(const ConvArgs a) {
    using T = __bf16;
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

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    const int li = lane & 15, lg = lane >> 4;

    const int total_tiles = a.nSp * a.nCt;
    const int tstride = (int)gridDim.x;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= total_tiles) return;                 // block-uniform
    // dynamic tile queue (conv_pp_common.hpp TileQ): this tile is the block's static one; `nxt` = the tile after it, drawn below; wave 0's lane 0 does the drawing
    const TileQ tq = tq_init(a.tq, total_tiles, tstride, (int)blockIdx.x);
    const bool dyn = tq.ctr != nullptr;
    const bool drawer = dyn && tid == 0;
    unsigned tk = 1u;
    if (drawer) tq_draw(tk, tq.ctr);                 // (older than every DMA of the prologue: the vmcnt(0) below covers it)
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

    // halo DMA j of this wave (instruction id = j*8 + wave); the per-lane offset is recomputed at every issue (registers go to the fragments)
    auto issue_halo = [&](auto jc, int n, int h0, int w0, int c0, char* dst) {
        constexpr int j = decltype(jc)::value;
        const int id = j * 8 + wave;
        if (id >= HINSTR) return;                     // wave-uniform
        const __amdgpu_buffer_rsrc_t rx = pp_make_rsrc(xb + (size_t)n * a.H * a.W * a.x0.ld * 2, img_x);
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
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (pp_lds_void_t*)(dst + wave * 256), 4, (col + wave * 64 + l) * 4, 0, 0, 0);
        }
    };
    // the three tap tiles (kh = 0..2) of filter column kw, column tile col, channels c0..c0+31: one instruction per tap per wave
    auto issue_weights = [&](int kw, int col, int c0, char* dst) {
        int soff = (int)((((long long)kw * a.Cout + col) * a.Cin + c0) * 2);
        asm volatile("" : "+s"(soff));
        const int tapstride = 3 * a.Cout * a.Cin * 2;          // tap index = kh*3 + kw
        if (wave >= WWAVES) return;                    // wave-uniform (NF = 2: the four waves of group 0 carry the whole tile)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) pp_dma16(rw, soff + kh * tapstride + w_goff0, dst + kh * TAPB + wave * 1024);
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
    if (grp == 1) issue_weights(1, ncol0, 0, wbase + WTILE);          // segment 1 (in the loop group 1 issues two segments ahead)
    pp_static_for<HJ>([&](auto jc) { issue_halo(jc, n, h0, w0, 0, hbase); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t mailbox = (uint32_t)(uintptr_t)(bbase + 2 * BN * 4);
    if (drawer) tq_post(tq, tk, mailbox);
    __syncthreads();
    int nxt = tile + tstride;
    if (dyn) {
        const unsigned k0 = tq_take(mailbox);
        nxt = tq_tile(tq, k0, tstride);
    }

    int wsel = 0, hsel = 0, bsel = 0;
#ifdef MIS_PP_STAMPS          // (diagnostic build: scripts/ppc_stamps.sh; slots as in conv_pp_kernel)
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = __builtin_amdgcn_s_memtime();
#endif
    if (grp == 1) __builtin_amdgcn_s_barrier();       // the stagger
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (;;) {
        const bool has_next = (unsigned)nxt < (unsigned)total_tiles;
        int nn = n, nh0 = h0, nw0 = w0, ncolN = ncol0;
        if (has_next) decode(nxt, nn, nh0, nw0, ncolN);
        // the tile after `nxt`: drawn now (the oldest entry of this wave's vmcnt queue for the whole tile: the kw = 2 wait of chunk 0, vmcnt(0), returns it), posted after
        // chunk 0, taken by every wave at the end of the tile - at least three segments' barriers later (Cin >= 64: two chunks)
        tk = 1u;
        if (drawer && has_next) tq_draw(tk, tq.ctr);
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
                if (grp == 0 && wnext) issue_weights(wkw, wcol, wc0, wbn);
#endif
                constexpr int NH = kw == 0 ? HJ0 : (kw == 1 ? HJ1 : 0);          // halo instructions issued in this segment (per wave; the last may be past the image)
#ifndef PPT_NO_DMA
                if constexpr (NH > 0) {
                    if (hnext) pp_static_for<NH>([&](auto jc) { issue_halo(std::integral_constant<int, (kw == 0 ? 0 : HJ0) + decltype(jc)::value>{}, hn, hh0, hw0, hc0, hbn); });
                }
#endif
                PP_STAMP(5)
                u32x4 A[3][NF], Brow[PF + 2];
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
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
#ifndef PPT_NO_MFMA
#pragma unroll
                        for (int pf = 0; pf < PF; ++pf) {
                            if constexpr (first && kw == 0) {
                                if (kh == 0) acc[f][pf] = bq[f];          // (folds into the MFMA's C operand: no copy is emitted when the compiler keeps bq in place)
                            }
                            mma_b128<T>(acc[f][pf], A[kh][f], Brow[pf + kh]);
                        }
#endif
#ifndef PPT_NO_DMA
                        if (kh == 0 && f == 0) {
                            if (grp == 1 && w2next) issue_weights(w2kw, w2col, w2c0, wb_self);
                        }
#endif
                    }
                __builtin_amdgcn_s_setprio(0);
                if (grp == 0) {                                  // group 0: the weight DMAs it issued in this segment's R
                    if (hnext && KEEP > 0) {
                        const int last_id = ((kw == 0 ? 0 : HJ0) + KEEP - 1) * 8 + wave;
                        if (last_id < HINSTR) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP > 0 ? KEEP - 1 : 0) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
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
        if (drawer && has_next) tq_post(tq, tk, mailbox);
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
        if (!has_next) break;
        tile = nxt;
        nxt = dyn ? tq_tile(tq, tq_take(mailbox), tstride) : tile + tstride;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();       // pairs with group 1's last barrier
#ifdef MIS_PP_STAMPS
    if (lane == 0 && blockIdx.x < 256) {
        for (int i = 0; i < 8; ++i) g_pp_stamps[(blockIdx.x * 8 + wave) * 8 + i] = st_[i];
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

template <int PF, int NF, int EM> static int pp_launch_col_em(const MisConvDesc* d, hipStream_t stream);
template <int PF, int NF> static int pp_launch_col(const MisConvDesc* d, hipStream_t stream) {      // one instantiation per epilogue mask path (pp_epilogue_plain)
    if (d->mask_bits != nullptr) return pp_launch_col_em<PF, NF, PP_EM_BITS>(d, stream);
    if (d->mask != nullptr) return pp_launch_col_em<PF, NF, PP_EM_MASK>(d, stream);
    return pp_launch_col_em<PF, NF, PP_EM_NONE>(d, stream);
}
template <int PF, int NF, int EM> static int pp_launch_col_em(const MisConvDesc* d, hipStream_t stream) {
    constexpr int BN = 2 * NF * 16;
    constexpr int TH = 4 * PF, HINSTR = ((TH + 2) * 18 * 4 + 63) / 64;
    ConvArgs a;
    a.tq = nullptr;
    a.N = d->N; a.D = 1; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.Cin0 = d->Cin; a.Cout0 = d->Cout0;
    a.x0 = SrcView{d->x0, d->x0_ld, 1, d->x0_H, d->x0_W};
    a.x1 = SrcView{nullptr, 0, 0, 0, 0};
    a.in_scale = nullptr; a.in_shift = nullptr;
    a.w = d->w; a.bias = d->bias; a.relu = d->relu; a.mask = d->mask; a.mask_ld = d->mask_ld; a.relu_bits = nullptr; a.mask_bits = reinterpret_cast<const unsigned char*>(d->mask_bits);
    a.y0 = d->y0; a.y0_ld = d->y0_ld; a.y0_mode = d->y0_mode;
    a.y1 = d->y1; a.y1_ld = d->y1_ld; a.y1_mode = d->y1_mode;
    a.relu_bits = reinterpret_cast<unsigned char*>(d->relu_bits);      // written from the epilogue (pp_epilogue_plain)
    g_conv_bits_fused = d->relu_bits != nullptr;
    a.tilesH = (d->H + TH - 1) / TH;
    a.tilesW = (d->W + 15) / 16;
    const long long nsp = (long long)d->N * a.tilesH * a.tilesW;
    a.nCt = d->Cout / BN;
    // spatial-major tile order when the persistent stride keeps a block on its column tile (256 % nCt == 0); MIS_CONV_PPC_COLMAJOR=1: the other order (A/B switch)
    a.tilesD = (256 % a.nCt == 0 && !mis_sw(SW_CONV_PPC_COLMAJOR)) ? 2 : 1;
    MIS_REQUIRE(nsp * a.nCt < (1ll << 31), MIS_EUNSUPPORTED, "conv_igemm(ppc): grid too large");
    a.nSp = (int)nsp;
    const size_t lds = 2 * (size_t)HINSTR * 1024 + 2 * (size_t)3 * BN * 64 + 2 * (size_t)BN * 4 + 16;          // (+ the tile queue's mailbox)
    static std::atomic<unsigned long long> attr_done{0};
    if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&conv_ppc_kernel<PF, NF, EM>), lds, "conv_igemm(ppc)")) return rc;
    const long long total = nsp * a.nCt;
    a.tq = d->Cin >= 64 ? mis_tile_queue(stream) : nullptr;          // (one K chunk: no barrier between the ticket's post and its take)
    hipLaunchKernelGGL((conv_ppc_kernel<PF, NF, EM>), dim3((unsigned)(total > mis_persist_cus() ? mis_persist_cus() : total)), dim3(512), lds, stream, a);
    MIS_LAUNCH_CHECK("conv_igemm(ppc)");
    return MIS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// 64 -> 64 channels (down_conv.0.second, up_conv.3.second and their dgrads: the layers with the lowest FLOP per byte): ping-pong with the filter in REGISTERS.
// A wave owns 64 px x 32 ch; its share of the whole 3x3 filter - 9 taps x 2 k-groups x 2 fragments = 144 VGPRs - is loaded once per kernel, so a segment only reads
// pixel fragments: one filter COLUMN (kw) and one k-group per segment = six row fragments (rows r .. r+3 serve tap kh = r) for 24 MFMAs, no weight traffic at all,
// one LDS-DMA stream (the halo tile of the next 16 x 16 tile, double-buffered).  Meant to replace conv64_ws_kernel (whole filter in LDS, 8 waves in lockstep: 38 % MFMA
// busy) - but it measures 0.700 vs 0.654 ms on 64->64 at 512^2 (884 vs 946 TFLOP/s): with 6 fragment reads per 24 MFMAs this layer is evidently not bound by the R/M
// structure but by its memory traffic (2.1-2.8 GB per launch, 16-byte store pieces here against 32-byte ones there).  Kept selectable (MIS_CONV_RS64=1) and tested.
__global__ __launch_bounds__(512, 2) void conv_pp_rs64_kernel(const ConvArgs a) {
    using T = __bf16;
    constexpr int NF = 2, PF = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hbase = smem;                        // 2 x PP_HBUF
    char* const bbase = smem + 2 * PP_HBUF;          // 64 bias floats (zeros without a bias)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    const int li = lane & 15, lg = lane >> 4;

    const int total_tiles = a.nSp;
    const int tstride = (int)gridDim.x;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= total_tiles) return;
    const int tpi = a.tilesH * a.tilesW;
    auto decode = [&](int t, int& tn, int& th0, int& tw0) {
        tn = t / tpi;
        const int r = t - tn * tpi;
        const int th = r / a.tilesW;
        th0 = th * PP_TH;
        tw0 = (r - th * a.tilesW) * PP_TW;
    };

    // ---- the filter: fragment (tap, kg, f) of this wave.  MFMA row i of fragment f must be output channel wn*32 + (i>>2)*8 + f*4 + (i&3), so that a lane ends up with
    //      8 consecutive channels (rows lg*4 .. lg*4+3 of both fragments); lane (li, lg) supplies row li, input channels (kg*4 + lg)*8 .. +7 ----
    u32x4 Wr[9][2][2];
    {
        const T* wp = reinterpret_cast<const T*>(a.w);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kg = 0; kg < 2; ++kg)
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int row = wn * 32 + (li >> 2) * 8 + f * 4 + (li & 3);
                    Wr[tap][kg][f] = *reinterpret_cast<const u32x4*>(wp + ((size_t)tap * 64 + row) * 64 + (kg * 4 + lg) * 8);
                }
    }
    if (tid < 64) reinterpret_cast<float*>(bbase)[tid] = a.bias != nullptr ? a.bias[tid] : 0.f;

    int b_off0[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int px = li + kw;
        b_off0[kw] = (wm * 4 * PP_HW + px) * 128 + ((lg ^ (px & 7)) << 4);
    }
    const unsigned img_x = (unsigned)(((long long)a.H * a.W - 1) * a.x0.ld + a.Cin) * 2u;
    const char* const xb = reinterpret_cast<const char*>(a.x0.p);
    // halo DMA j (0..5) of this wave: instruction id = j*8 + wave of the image of tile (n, h0, w0)
    auto issue_halo = [&](auto jc, int n, int h0, int w0, char* dst) {
        constexpr int j = decltype(jc)::value;
        const int id = j * 8 + wave;
        if (id >= PP_HINSTR) return;                  // wave-uniform
        const __amdgpu_buffer_rsrc_t rx = pp_make_rsrc(xb + (size_t)n * a.H * a.W * a.x0.ld * 2, img_x);
        unsigned toff = (unsigned)((((h0 - 1) * a.W + (w0 - 1)) * a.x0.ld) * 2);
        asm volatile("" : "+s"(toff));
        int item = id * 64 + lane;
        asm volatile("" : "+v"(item));                // recomputed per issue: the filter leaves no registers for per-lane tables
        const int p = item >> 3, pos = item & 7;
        const int py = p / PP_HW, px = p - py * PP_HW;
        const unsigned rel = (unsigned)(((py * a.W + px) * a.x0.ld + ((pos ^ (px & 7)) << 3)) * 2);
        const bool ok = item < PP_HITEMS && (unsigned)(h0 - 1 + py) < (unsigned)a.H && (unsigned)(w0 - 1 + px) < (unsigned)a.W;
        pp_dma16(rx, ok ? (int)(toff + rel) : PP_OOB, dst + id * 1024);
    };

    int n, h0, w0;
    decode(tile, n, h0, w0);
    f32x4 acc[NF][PF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) acc[f][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
    pp_static_for<6>([&](auto jc) { issue_halo(jc, n, h0, w0, hbase); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int hsel = 0;
    if (grp == 1) __builtin_amdgcn_s_barrier();       // the stagger
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (; tile < total_tiles; tile += tstride) {
        const bool has_next = tile + tstride < total_tiles;
        int nn = n, nh0 = h0, nw0 = w0;
        if (has_next) decode(tile + tstride, nn, nh0, nw0);
        const uint32_t hb = (uint32_t)(uintptr_t)hbase + hsel * PP_HBUF;
        char* hbn = hbase + (hsel ^ 1) * PP_HBUF;
        pp_static_for<6>([&](auto sc) {                // segment = (filter column kw, k-group kg)
            constexpr int sg = decltype(sc)::value;
            constexpr int kw = sg >> 1, kg = sg & 1;
            // ================= R segment =================
            if constexpr (sg < 3) {                    // the next tile's halo: two instructions per wave in each of the first three segments
                if (has_next) {
                    issue_halo(std::integral_constant<int, 2 * sg>{}, nn, nh0, nw0, hbn);
                    issue_halo(std::integral_constant<int, 2 * sg + 1>{}, nn, nh0, nw0, hbn);
                }
            }
            u32x4 Brow[6];
            pp_static_for<6>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                Brow[r] = pp_lds_read128<r * PP_ROWB>(hb + (b_off0[kw] ^ (kg << 6)));
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (sg == 5) {
                if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // group 1's last slot of the tile: the next halo has landed
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ================= M segment: 3 taps x 2 fragments x 4 pixel rows =================
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf) mma_b128<T>(acc[f][pf], Wr[kh * 3 + kw][kg][f], Brow[pf + kh]);
            __builtin_amdgcn_s_setprio(0);
            if constexpr (sg == 5) {
                if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // group 0's last slot of the tile
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        });
        pp_epilogue<NF>(a, acc, (uint32_t)(uintptr_t)bbase, n, h0, w0, 0, wm, wn, li, lg);
        hsel ^= 1;
        n = nn; h0 = nh0; w0 = nw0;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
}

static int pp_launch_rs64(const MisConvDesc* d, hipStream_t stream) {
    ConvArgs a;
    a.tq = nullptr;
    a.N = d->N; a.D = 1; a.H = d->H; a.W = d->W; a.Cin = 64; a.Cout = 64; a.Cin0 = 64; a.Cout0 = 64;
    a.x0 = SrcView{d->x0, d->x0_ld, 1, d->x0_H, d->x0_W};
    a.x1 = SrcView{nullptr, 0, 0, 0, 0};
    a.in_scale = nullptr; a.in_shift = nullptr;
    a.w = d->w; a.bias = d->bias; a.relu = d->relu; a.mask = d->mask; a.mask_ld = d->mask_ld; a.relu_bits = nullptr; a.mask_bits = reinterpret_cast<const unsigned char*>(d->mask_bits);
    a.y0 = d->y0; a.y0_ld = d->y0_ld; a.y0_mode = d->y0_mode;
    a.y1 = nullptr; a.y1_ld = 0; a.y1_mode = 0;
    a.tilesD = 1;
    a.tilesH = (d->H + PP_TH - 1) / PP_TH;
    a.tilesW = (d->W + PP_TW - 1) / PP_TW;
    const long long nsp = (long long)d->N * a.tilesH * a.tilesW;
    MIS_REQUIRE(nsp < (1ll << 31), MIS_EUNSUPPORTED, "conv_igemm(rs64): grid too large");
    a.nSp = (int)nsp;
    a.nCt = 1;
    const size_t lds = 2 * (size_t)PP_HBUF + 256;
    static std::atomic<unsigned long long> attr_done{0};
    if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&conv_pp_rs64_kernel), lds, "conv_igemm(rs64)")) return rc;
    hipLaunchKernelGGL(conv_pp_rs64_kernel, dim3((unsigned)(nsp > mis_persist_cus() ? mis_persist_cus() : nsp)), dim3(512), lds, stream, a);
    MIS_LAUNCH_CHECK("conv_igemm(rs64)");
    return MIS_OK;
}

// ---------------------------------------------------------------------------------------------------------
bool conv_pp_eligible(const MisConvDesc* d) {
    if (d->dtype != MIS_BF16 || d->is3d || d->ksize != 3) return false;
    if (d->x1 != nullptr || d->in_scale != nullptr) return false;
    if (d->x0_H != d->H || d->x0_W != d->W || d->D != 1) return false;
    if (d->Cin % 64 != 0 || d->Cout % 64 != 0) return false;
    if (d->Cout0 % 64 != 0) return false;              // a wave's 64 (128-column blocks) / 128 (256-column blocks) columns go to ONE destination
    if (d->bias != nullptr && (d->y0_mode == MIS_OUT_SHUFFLE2 || (d->y1 != nullptr && d->y1_mode == MIS_OUT_SHUFFLE2))) return false;   // bias is indexed by GEMM column here
    // 32-bit buffer offsets, computed in (signed) int by the kernels: ONE image of the input view and the packed weights must each span less than 2 GiB
    if ((((long long)d->H * d->W - 1) * d->x0_ld + d->Cin) * 2 >= (1ll << 31) - 65536) return false;
    if ((long long)9 * d->Cout * d->Cin * 2 >= (1ll << 31) - 65536) return false;
    return true;
}

template <int NF> static int pp_launch(const MisConvDesc* d, hipStream_t stream) {
    constexpr int BN = 2 * NF * 16;
    ConvArgs a;
    a.tq = nullptr;
    a.N = d->N; a.D = 1; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.Cin0 = d->Cin; a.Cout0 = d->Cout0;
    a.x0 = SrcView{d->x0, d->x0_ld, 1, d->x0_H, d->x0_W};
    a.x1 = SrcView{nullptr, 0, 0, 0, 0};
    a.in_scale = nullptr; a.in_shift = nullptr;
    a.w = d->w; a.bias = d->bias; a.relu = d->relu; a.mask = d->mask; a.mask_ld = d->mask_ld; a.relu_bits = nullptr; a.mask_bits = reinterpret_cast<const unsigned char*>(d->mask_bits);
    a.y0 = d->y0; a.y0_ld = d->y0_ld; a.y0_mode = d->y0_mode;
    a.y1 = d->y1; a.y1_ld = d->y1_ld; a.y1_mode = d->y1_mode;
    a.tilesD = 1;
    a.tilesH = (d->H + PP_TH - 1) / PP_TH;
    a.tilesW = (d->W + PP_TW - 1) / PP_TW;
    const long long nsp = (long long)d->N * a.tilesH * a.tilesW;
    a.nCt = d->Cout / BN;
    MIS_REQUIRE(nsp * a.nCt < (1ll << 31), MIS_EUNSUPPORTED, "conv_igemm(pp): grid too large");
    a.nSp = (int)nsp;
    const size_t lds = 2 * (size_t)PP_HBUF + 2 * (size_t)BN * 128 + 2 * (size_t)BN * 4;
    static std::atomic<unsigned long long> attr_done{0};
    if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&conv_pp_kernel<NF>), lds, "conv_igemm(pp)")) return rc;
    const long long total = nsp * a.nCt;
    hipLaunchKernelGGL((conv_pp_kernel<NF>), dim3((unsigned)(total > mis_persist_cus() ? mis_persist_cus() : total)), dim3(512), lds, stream, a);
    MIS_LAUNCH_CHECK("conv_igemm(pp)");
    return MIS_OK;
}

// pp_epilogue_plain: plain or pixel-unshuffled NHWC destinations, and ONE image of each destination view / of the mask spans less than 4 GiB - 64 KiB (32-bit buffer offsets)
static bool ppc_plain_ok(const MisConvDesc* d) {
    const long long img = (long long)d->H * d->W;
    const long long lim = (1ll << 32) - 65536;
    auto view_ok = [&](int mode, int ld, int cview) {
        if (mode == MIS_OUT_PLAIN) return ((img - 1) * ld + cview) * 2 < lim;
        if (mode == MIS_OUT_UNSHUFFLE2) return d->H % 2 == 0 && d->W % 2 == 0 && ((img / 4 - 1) * ld + 4ll * cview) * 2 < lim;
        return false;
    };
    if (!view_ok(d->y0_mode, d->y0_ld, d->Cout0)) return false;
    if (d->Cout0 < d->Cout && (d->y1 == nullptr || !view_ok(d->y1_mode, d->y1_ld, d->Cout - d->Cout0))) return false;
    if (d->mask != nullptr && ((img - 1) * d->mask_ld + d->Cout) * 2 >= lim) return false;
    if ((d->relu_bits != nullptr || d->mask_bits != nullptr) && (long long)rb_bytes(d->N, d->H, d->W, d->Cout) >= lim) return false;      // ReLU bits: one 32-bit-offset resource
    return true;
}

// 64-column blocks of the column-segment kernel by default: Cout = 64 (or any multiple of 64 that is not one of 128) with at least 128 input channels, on grids whose
// 32-row tiles waste at most 15 % (measured in one process, scripts/bench_conv_layers.py: 128->64 at 512^2 1.249 vs 1.379 ms for bn64.persist.dma, at 256^2 0.320 vs 0.336;
// 64->64 stays on the weight-stationary kernel, 0.633 vs 0.751 ms)
bool conv_ppc64_auto(const MisConvDesc* d) {
    return conv_pp_eligible(d) && d->Cout % 128 != 0 && (d->Cin >= 128 || !mis_sw(SW_CONV_NOPPD)) && !mis_sw(SW_CONV_NOPPC) && ppc_plain_ok(d) && ((d->H + 31) / 32) * 32 * 100 <= d->H * 115;
}

bool conv_pp_rs64_eligible(const MisConvDesc* d) {
    return conv_pp_eligible(d) && d->Cin == 64 && d->Cout == 64 && d->Cout0 == 64 && d->y0_mode == MIS_OUT_PLAIN && mis_sw(SW_CONV_RS64);      // opt-in: measured 0.700 vs 0.654 ms against conv64_ws_kernel
}

// which column-segment configuration launch_conv_pp takes for a descriptor conv_pp_eligible accepts: 4 / 2 = conv_ppc_kernel<8, 4 / 2>, 0 = none (conv_pp_kernel / rs64)
//   Cout % 128 == 0: 128-column blocks, unless the 32-row tiles waste more than 15 % of the rows (MIS_CONV_PPC=1 takes it regardless) or MIS_CONV_NOPPC is set
//   (the parity tests reach every kernel on small grids that way); else 64-column blocks where conv_ppc64_auto says so (MIS_CONV_PPC64=1: wherever eligible)
int conv_ppc_choice(const MisConvDesc* d) {
    if (conv_pp_rs64_eligible(d)) return 0;
    if (d->Cout % 128 == 0 && !mis_sw(SW_CONV_NOPPC) && ppc_plain_ok(d) && (((d->H + 31) / 32) * 32 * 100 <= d->H * 115 || mis_sw(SW_CONV_PPC))) return 4;
    if (d->Cout % 64 == 0 && (mis_sw(SW_CONV_PPC64) || (conv_ppc64_auto(d) && !mis_sw(SW_CONV_PP64))) && ppc_plain_ok(d)) return 2;
    return 0;
}

int launch_conv_pp(const MisConvDesc* d, hipStream_t stream, const char** tag) {
    const int ppc = conv_ppc_choice(d);
    // (the tag names the instantiation: one per epilogue mask path, ".mask" = bf16 mask tensor, ".bits" = ReLU bits)
#ifdef MIS_EXPERIMENTS          // (make EXPERIMENTS=1: the two round-5 experiments on this kernel, csrc/experiments/ - slower, kept as evidence, not part of the shipped library)
    if (ppc == 4 && conv_pps_eligible(d)) return launch_conv_pps(d, stream, tag);
    if (ppc == 4 && conv_ppc2_eligible(d)) return launch_conv_ppc2(d, stream, tag);
#endif
    if (ppc == 4) {
        *tag = d->mask_bits != nullptr ? "k3.2d.ppc8.bits" : (d->mask != nullptr ? "k3.2d.ppc8.mask" : "k3.2d.ppc8");
        return pp_launch_col<8, 4>(d, stream);
    }
    if (ppc == 2 && !mis_sw(SW_CONV_NOPPD)) return launch_conv_ppd(d, stream, tag);          // (Cin % 64 == 0: conv_pp_eligible)
    if (ppc == 2) {
        *tag = d->mask_bits != nullptr ? "k3.2d.ppc8n2.bits" : (d->mask != nullptr ? "k3.2d.ppc8n2.mask" : "k3.2d.ppc8n2");
        return pp_launch_col<8, 2>(d, stream);
    }
    MIS_REQUIRE(d->mask_bits == nullptr, MIS_EUNSUPPORTED, "conv_igemm(pp): ReLU bits are read by the column-segment kernels only");      // (dispatch never sends one here)
    if (conv_pp_rs64_eligible(d)) {
        *tag = "k3.2d.rs64";
        return pp_launch_rs64(d, stream);
    }
    if (d->Cout % 256 == 0 && d->Cout0 % 128 == 0) {
        const int no256 = mis_sw(SW_CONV_PP_NO256);
        if (!no256) {
            *tag = "k3.2d.pp256";
            return pp_launch<8>(d, stream);
        }
    }
    if (d->Cout % 128 == 0) {
        *tag = "k3.2d.pp128";
        return pp_launch<4>(d, stream);
    }
    *tag = "k3.2d.pp64";
    return pp_launch<2>(d, stream);
}
