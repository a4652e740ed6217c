// fp32 weight gradient of the 3x3 / 3x3x3 layers on v_mfma_f32_16x16x4_f32 for gfx950 - round 5.
//
//   dW[tap][ci][co] = sum_pixels X[pixel + tap][ci] * dY[pixel][co]          (X = the plain operand the forward convolution read: the materialised GroupNorm output)
//
// The round-1 kernel (wgrad.hip: 32 ci x 64 co blocks, 128-pixel tiles staged through registers, two barriers per tile, lock-step waves) kept the matrix pipe 58 %
// busy.  An f32 MFMA holds the pipe for 32 cycles, so the arithmetic itself needs almost nothing from the rest of the CU: what it needs is never to wait.  This kernel is
// the streaming structure of wgrad_pp_stream_kernel rebuilt around that fact:
//   * a block owns 64 ci x 64 co x the nine in-plane taps (3-D: the depth slice kd of the filter joins the block's identity; X comes from plane z + kd - 1) and walks DOWN
//     16-pixel-wide strips; its operands stream through a ring of five ROW slots (X row 18 px x 256 B + dY row 16 px x 256 B = 9 KiB) filled by LDS-DMA three rows
//     ahead (buffer_load ... lds; image borders and planes outside the volume = out-of-range offsets = zeros), one counted vmcnt wait and ONE barrier per row;
//   * wave w owns all 64 ci x output channels [16 w, 16 w + 16): the A operand of a 4-pixel k-step and tap is ONE ds_read_b128 per lane (lane (i, g): channels
//     4i .. 4i + 3 of pixel g - with 256-byte pixels the read is conflict-free without any swizzle, and every tap / k-step shift is an immediate), its four components are
//     the A operands of four MFMAs (row tile r = channels 4i + r); B is one ds_read_b32;
//   * the reads of sub-step (k-step, filter row) q + 1 are issued before the twelve MFMAs of sub-step q, into the other register set; the row's barrier sits before its
//     LAST sub-step, whose fragments are then already in registers - behind it the next row is visible, its first fragments are read while twelve MFMAs are still to
//     issue, and the oldest slot is free for the row three ahead;
//   * consecutive MFMAs never share an accumulator (36 tiles per wave);
//   * 256-thread blocks, two per CU; split-K over contiguous row ranges, one fp32 slab per block and (ci, co, kd) tile, reduced by the fixed-order kernels of wgrad.hip.
//
// Replaces the weight part of aten::convolution_backward for nn.Conv3d / nn.Conv2d(k3, p1) without bias (reference model/unet3d/buildingblocks.py:64-66).
#include "common.hpp"
#include "conv_pp_common.hpp"
#include "dispatch_cfg.hpp"
#include "wgrad_args.hpp"

struct WfArgs {
    const float* x;
    const float* dy;
    float* partial;
    int x_ld, dy_ld;
    int N, D, H, W, Cin, Cout;
    int nstrips, nCo, KD, base;                  // base = nCi * nCo * KD blocks per split
    long long U, upb;                            // row units (planes x strips x H) in all / per split
};

namespace {
constexpr int WF_TW = 16;
constexpr int WF_QROW = 4 * 1024;                // dY row image: 16 px x 256 B
constexpr int WF_R = 5;                          // ring slots
constexpr int WF_D = 3;                          // prefetch distance in rows
// X row image.  64 input channels per block: 18 px x 256 B = 4608 B in five DMA instructions (the last half full).  C32 (round 6: 32 input channels per block - the
// weight gradient of encoders.0 SingleConv2, whose operand has 32 real channels, reference buildingblocks.py:202-211): 18 px x 128 B = 2304 B; instruction q = wave q covers
// pixels 8q .. 8q + 7, so wave 2 lands pixels 16, 17 and zeros, wave 3 zeros only - every wave issues two instructions per element and the counted waits are uniform
template <bool C32> struct WfGeom {
    static constexpr int XPB = C32 ? 128 : 256;              // bytes of a pixel in the X row image
    static constexpr int XROW = C32 ? 4 * 1024 : 5 * 1024;
    static constexpr int SLOT = XROW + WF_QROW;
    static constexpr int LDS = WF_R * SLOT;                  // 46,080 / 40,960
    static constexpr int CIB = C32 ? 32 : 64;                // input channels per block
    static constexpr int NR = C32 ? 2 : 4;                   // row tiles (MFMAs) per fragment read
};


template <int OFF> __device__ __forceinline__ u32x2 wf_read64(uint32_t addr) {
    u32x2 r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF> __device__ __forceinline__ uint32_t wf_read32(uint32_t addr) {
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
__device__ __forceinline__ float wf_f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ float wf_f(const u32x4& v, int t) {
    const uint32_t u = v[t];
    return __uint_as_float(u);
}
__device__ __forceinline__ float wf_f(const u32x2& v, int t) {
    const uint32_t u = v[t];
    return __uint_as_float(u);
}
}   // namespace

template <bool IS3D, bool C32>
__global__ __launch_bounds__(256, 2) void wgrad_f32_stream_kernel(const WfArgs a) {
    using GEO = WfGeom<C32>;
    constexpr int WF_XROW = GEO::XROW, WF_SLOT = GEO::SLOT, XPB = GEO::XPB, NR = GEO::NR;
    using AFrag = std::conditional_t<C32, u32x2, u32x4>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int pair = v % a.base;
    const int split = v / a.base;
    const int kd = IS3D ? pair % 3 : 0;
    const int pc = IS3D ? pair / 3 : pair;
    const int ci_t = pc / a.nCo, co_t = pc - ci_t * a.nCo;
    const int ci0 = ci_t * GEO::CIB, co0 = co_t * 64;
    const long long u_begin = (long long)split * a.upb;
    long long u_end = u_begin + a.upb;
    if (u_end > a.U) u_end = a.U;

    // segment starting at row unit u of a range that ends at u_end: plane, first column, first row, rows
    auto segment = [&](long long u, int& plane, int& w0, int& ya, int& R) {
        const long long ps = u / a.H;
        ya = (int)(u - ps * a.H);
        plane = (int)(ps / a.nstrips);
        w0 = (int)(ps - (long long)plane * a.nstrips) * WF_TW;
        long long r = a.H - ya;
        if (r > u_end - u) r = u_end - u;
        R = (int)r;
    };
    int total = 0;                               // stream elements of this block: R + 2 per segment
    for (long long u = u_begin; u < u_end;) {
        int p_, w_, y_, R_;
        segment(u, p_, w_, y_, R_);
        total += R_ + 2;
        u += R_;
    }

    // ---- DMA lane parts: X instruction q covers slots q*64 .. q*64 + 63 of the row image (slot = pixel * 16 + 16-byte chunk), waves 0-3: q = wave, wave 0 also q = 4 ----
    const int xs = wave * 64 + lane;
    const int xpx = C32 ? ((xs >> 3) < 18 ? (xs >> 3) : 0x40000000) : xs >> 4;            // 0 .. 15 (C32: 0 .. 17, the rest out of range)
    const unsigned xrel = C32 ? (unsigned)(((xs >> 3) * a.x_ld + ((xs & 7) << 2)) * 4) : (unsigned)((xpx * a.x_ld + ((xs & 15) << 2)) * 4);
    const int xpx4 = lane < 32 ? 16 + (lane >> 4) : 0x40000000;                           // wave 0's fifth instruction: pixels 16, 17
    const unsigned xrel4 = (unsigned)(((16 + (lane >> 4)) * a.x_ld + ((lane & 15) << 2)) * 4);
    const int qpx = xs >> 4;
    const unsigned qrel = (unsigned)((qpx * a.dy_ld + ((xs & 15) << 2)) * 4);
    const unsigned img_x = (unsigned)((((size_t)a.H * a.W - 1) * a.x_ld + ci0 + GEO::CIB) * 4), img_q = (unsigned)((((size_t)a.H * a.W - 1) * a.dy_ld + a.Cout) * 4);
    const unsigned xrow = (unsigned)(a.W * a.x_ld * 4), qrow = (unsigned)(a.W * a.dy_ld * 4);

    // ---- the issue cursor runs WF_D elements ahead of the steps ----
    long long pu = u_begin;
    int pplane, pw0, pya, pR, pj = 0, pslot = 0, issued = 0;
    bool pzok = true;
    __amdgpu_buffer_rsrc_t prx, prq;
    unsigned ptoff, pqoff;
    auto cursor_segment = [&]() {
        segment(pu, pplane, pw0, pya, pR);
        int xplane = pplane;
        if constexpr (IS3D) {
            const int z = pplane % a.D + kd - 1;
            pzok = (unsigned)z < (unsigned)a.D;
            xplane = pzok ? pplane + kd - 1 : pplane;
        }
        prx = pp_make_rsrc(a.x + (size_t)xplane * a.H * a.W * a.x_ld, img_x);
        prq = pp_make_rsrc(a.dy + (size_t)pplane * a.H * a.W * a.dy_ld, img_q);
        ptoff = (unsigned)((((pya - 1) * a.W + (pw0 - 1)) * a.x_ld + ci0) * 4);
        pqoff = (unsigned)((((pya - 2) * a.W + pw0) * a.dy_ld + co0) * 4);
    };
    cursor_segment();
    auto issue_next = [&]() {
        if (issued >= total) __builtin_amdgcn_s_sleep(2);          // nothing left to issue: keep the reads behind the barrier away from it all the same (wgrad_pp.hip)
        if (issued < total) {
            char* const slot = smem + pslot * WF_SLOT;
            {
                const bool ok = pzok && (unsigned)(pw0 - 1 + xpx) < (unsigned)a.W;
                pp_dma16(prx, ok ? (int)(ptoff + xrel) : PP_OOB, slot + wave * 1024);
            }
            if constexpr (!C32) {
                if (wave == 0) {
                    const bool ok = pzok && (unsigned)(pw0 - 1 + xpx4) < (unsigned)a.W;
                    pp_dma16(prx, ok ? (int)(ptoff + xrel4) : PP_OOB, slot + 4 * 1024);
                }
            }
            {
                const bool ok = pj >= 2 && pw0 + qpx < a.W;
                pp_dma16(prq, ok ? (int)(pqoff + qrel) : PP_OOB, slot + WF_XROW + wave * 1024);
            }
            ++issued;
            pslot = pslot == WF_R - 1 ? 0 : pslot + 1;
            ptoff += xrow;
            pqoff += qrow;
            if (++pj == pR + 2) {
                pj = 0;
                pu += pR;
                if (pu < u_end) cursor_segment();
            }
        }
    };

    f32x4 acc[9][NR];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[t][r] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment lane parts: A = pixel lg (+ k-step / tap shift as immediate), 16-byte chunk li; B = pixel lg, channel 16 * wave + li
    const uint32_t a_lane = lds0 + (uint32_t)(lg * XPB + li * (C32 ? 8 : 16));      // (C32: 8 bytes per lane = channels 2 li, 2 li + 1: 256 contiguous bytes per half-wave)
    const uint32_t b_lane = lds0 + (uint32_t)(WF_XROW + lg * 256 + (wave * 16 + li) * 4);

    for (int i = 0; i < WF_D; ++i) issue_next();
    // element 0 must have landed; WF_D - 1 younger ones may stay in flight (wave 0 issues three instructions per element, the others two)
    if (total >= WF_D) {
        if (!C32 && wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (WF_D - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (WF_D - 1)) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_sleep(2);                 // (see wgrad_pp.hip: keep the first reads away from the publishing barrier)

    // reads of sub-step q = s * 3 + kh of the element whose X rows (kh = 0, 1, 2) are at xa[0..2] and whose dY row is at qa; B of k-step s joins sub-step s * 3
    auto read_sub = [&](AFrag(&A)[3], uint32_t& B, const uint32_t (&xa)[3], uint32_t qa, auto qc) {
        constexpr int q = decltype(qc)::value, s = q / 3, kh = q % 3;
        pp_static_for<3>([&](auto kwc) {
            constexpr int kw = decltype(kwc)::value;
            if constexpr (C32) A[kw] = wf_read64<(4 * s + kw) * XPB>(xa[kh]);
            else A[kw] = pp_lds_read128<(4 * s + kw) * XPB>(xa[kh]);
        });
        if constexpr (kh == 0) B = wf_read32<s * 1024>(qa);
    };
    auto wait_sub = [&](AFrag(&A)[3], uint32_t& B) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(B)::"memory"); };
    // the MFMAs of filter columns [K0, K1) of a sub-step (4 per column)
    auto mfma_cols = [&](const AFrag(&A)[3], uint32_t B, auto khc, auto k0c, auto k1c) {
        constexpr int kh = decltype(khc)::value, K0 = decltype(k0c)::value, K1 = decltype(k1c)::value;
        const float b = wf_f(B);
#pragma unroll
        for (int kw = K0; kw < K1; ++kw)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[kh * 3 + kw][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf_f(A[kw], r), b, acc[kh * 3 + kw][r], 0, 0, 0);
    };
    auto mfma_sub = [&](const AFrag(&A)[3], uint32_t B, auto khc) { mfma_cols(A, B, khc, std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}); };

    AFrag A0[3], A1[3];
    uint32_t B0 = 0u, B1 = 0u;                   // B of even / odd k-steps
    uint32_t xa[3] = {a_lane, a_lane, a_lane};   // X rows of elements e - 2, e - 1, e
    bool pref = false;                           // sub-step 0 of the current element was read during the previous element's last sub-step
    int slot = 0;

    long long u = u_begin;
    int e = 0;
#pragma unroll 1
    while (u < u_end) {
        int plane_, w0_, ya_, R;
        segment(u, plane_, w0_, ya_, R);
        u += R;
#pragma unroll 1
        for (int jj = 0; jj < R + 2; ++jj, ++e) {
            const bool comp = jj >= 2;                       // block-uniform
            const bool next_comp = jj + 1 >= 2 && jj + 1 < R + 2;
            xa[0] = xa[1];
            xa[1] = xa[2];
            xa[2] = a_lane + (uint32_t)(slot * WF_SLOT);
            const uint32_t qa = b_lane + (uint32_t)(slot * WF_SLOT);
            const int nslot = slot == WF_R - 1 ? 0 : slot + 1;
            if (comp) {
                if (!pref) {
                    read_sub(A0, B0, xa, qa, std::integral_constant<int, 0>{});
                    wait_sub(A0, B0);
                }
                __builtin_amdgcn_sched_barrier(0);
                pp_static_for<11>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    constexpr int s = q / 3, kh = q % 3, s1 = (q + 1) / 3;
                    using K0 = std::integral_constant<int, 0>;
                    using K1 = std::integral_constant<int, 1>;
                    using K3 = std::integral_constant<int, 3>;
                    // the sub-step's first four MFMAs, THEN the fragment reads of sub-step q + 1 into the other set (B of k-step s1 into B[s1 & 1]: k-step s still reads
                    // B[s & 1]), then its other eight: a sub-step boundary is one s_waitcnt - with the reads at the boundary the pipe ran dry for a few cycles twelve times per row
                    if constexpr (q % 2 == 0) mfma_cols(A0, s % 2 == 0 ? B0 : B1, std::integral_constant<int, kh>{}, K0{}, K1{});
                    else mfma_cols(A1, s % 2 == 0 ? B0 : B1, std::integral_constant<int, kh>{}, K0{}, K1{});
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (q % 2 == 0) {
                        if constexpr (s1 % 2 == 0) read_sub(A1, B0, xa, qa, std::integral_constant<int, q + 1>{});
                        else read_sub(A1, B1, xa, qa, std::integral_constant<int, q + 1>{});
                    } else {
                        if constexpr (s1 % 2 == 0) read_sub(A0, B0, xa, qa, std::integral_constant<int, q + 1>{});
                        else read_sub(A0, B1, xa, qa, std::integral_constant<int, q + 1>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (q % 2 == 0) mfma_cols(A0, s % 2 == 0 ? B0 : B1, std::integral_constant<int, kh>{}, K1{}, K3{});
                    else mfma_cols(A1, s % 2 == 0 ? B0 : B1, std::integral_constant<int, kh>{}, K1{}, K3{});
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (q % 2 == 0) wait_sub(A1, s1 % 2 == 0 ? B0 : B1);
                    else wait_sub(A0, s1 % 2 == 0 ? B0 : B1);
                });
            }
            // ---- the next element becomes visible; the slot of element e - 2 is handed to element e + WF_D ----
            if (e + 1 < total) {
                if (e + WF_D < total) {          // the issue cursor is still running: WF_D - 2 younger elements stay in flight
                    if (!C32 && wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (WF_D - 2)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (WF_D - 2)) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            // sub-step 11 (k-step 3, kh = 2; fragments in A1, B of k-step 3 in B1): its first four MFMAs go out right behind the barrier, the DMA issue of the row three
            // ahead (cursor bookkeeping: ~50 scalar / vector instructions) and the next element's first fragment reads follow INSIDE the cluster - the first version issued
            // both between the barrier and the first MFMA, which left the pipe idle for 5 % of every row (PMC: 94.3 % busy)
            if (comp) {
                mfma_cols(A1, B1, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
            }
            if (e + 1 < total) {
                issue_next();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (comp) {
                mfma_cols(A1, B1, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
                __builtin_amdgcn_sched_barrier(0);
                if (next_comp) {
                    const uint32_t xn[3] = {xa[1], xa[2], a_lane + (uint32_t)(nslot * WF_SLOT)};
                    read_sub(A0, B0, xn, b_lane + (uint32_t)(nslot * WF_SLOT), std::integral_constant<int, 0>{});
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_cols(A1, B1, std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
                __builtin_amdgcn_sched_barrier(0);
                wait_sub(A0, B0);
            }
            pref = comp && next_comp;
            slot = nslot;
        }
    }

    // ---- this block's slab: partial[split][kd * 9 + tap][ci][co]; lane (li, lg) holds ci = 16 lg + 4 i + r (C32: 8 lg + 2 i + r), co = 16 wave + li of row tile r ----
    float* out = a.partial + (size_t)split * (IS3D ? 27 : 9) * a.Cin * a.Cout;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ci = ci0 + (C32 ? 8 * lg + 2 * i + r : 16 * lg + 4 * i + r), co = co0 + 16 * wave + li;
                out[((size_t)(kd * 9 + tap) * a.Cin + ci) * a.Cout + co] = acc[tap][r][i];
            }
}

// ---- plan / launch ----
bool wgrad_f32_eligible(const MisWgradDesc* d) {
    if (mis_sw(SW_WGRAD_F32_NOPP)) return false;
    if (d->dtype != MIS_F32 || d->ksize != 3) return false;
    if (d->x1 != nullptr || d->in_scale != nullptr || d->dbias != nullptr || d->dw_per_sample != nullptr || d->dbias_per_sample != nullptr || d->dw_layout != 0) return false;
    if (d->Cin0 != d->Cin || d->Cin % 32 != 0 || d->Cout % 64 != 0 || d->x0_ld % 4 != 0 || d->dy_ld % 4 != 0) return false;
    if (d->x0_D != d->D || d->x0_H != d->H || d->x0_W != d->W) return false;
    if (!d->is3d && d->D != 1) return false;
    const size_t ld = d->x0_ld > d->dy_ld ? d->x0_ld : d->dy_ld;
    if (((size_t)(d->H + 2) * d->W * ld + 64) * 4 >= 0x7FFF0000ull) return false;           // one plane per buffer resource; row offsets may run one row past either end
    return true;
}

static void wf_plan(const MisWgradDesc* d, int* nsplit, long long* U, long long* upb, int* base, int* nstrips) {
    const int KD = d->is3d ? 3 : 1;
    *nstrips = (d->W + WF_TW - 1) / WF_TW;
    *base = (d->Cin / (d->Cin % 64 != 0 ? 32 : 64)) * (d->Cout / 64) * KD;           // (Cin = 32 mod 64: 32-channel blocks)
    *U = (long long)d->N * d->D * *nstrips * d->H;
    // two blocks per CU: grids that fill whole rounds of 512 block slots; among equally full ones the coarsest split (fewest slabs), but at least eight rows per block
    const int slots = 2 * mis_persist_cus();
    long long maxs = *U / 8;
    if (maxs < 1) maxs = 1;
    const long long slab = (long long)(d->is3d ? 27 : 9) * d->Cin * d->Cout * 4;
    while (maxs > 1 && maxs * slab > (512ll << 20)) --maxs;
    if (maxs > 4096) maxs = 4096;
    // ... and at least `minr` rounds (MIS_WGRAD_F32_ROUNDS, default 1).  Measured at cfg4's decoder layers: 1 / 2 / 4 / 8 rounds = 46.0 / 46.1 / 46.4 / 48.7 ms over the six
    // launches - there is no tail to shorten, more blocks only add slab traffic and prologues
    const int minr = mis_sw(SW_WGRAD_F32_ROUNDS) > 0 ? mis_sw(SW_WGRAD_F32_ROUNDS) : 1;
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= maxs; ++s) {
        const long long grid = (long long)*base * s;
        const long long rounds = (grid + slots - 1) / slots;
        const long long rows = (*U + s - 1) / s;                 // the longest block's rows
        double eff = (double)*U / ((double)rounds * slots / *base * rows);      // useful rows per (block slot x rounds) of the longest block
        if (rounds < minr && s < maxs) eff *= 0.5;
        if (eff > best_eff * 1.02) {
            best_eff = eff;
            best = s;
        }
        if (grid >= (long long)(minr + 3) * slots) break;
    }
    *upb = (*U + best - 1) / best;
    *nsplit = (int)((*U + *upb - 1) / *upb);
}

int wgrad_f32_nsplit(const MisWgradDesc* d) {
    int ns, base, nstrips;
    long long U, upb;
    wf_plan(d, &ns, &U, &upb, &base, &nstrips);
    return ns;
}

int launch_wgrad_f32(const MisWgradDesc* d, float* partial, hipStream_t stream, const char** tag) {
    WfArgs a;
    int ns;
    wf_plan(d, &ns, &a.U, &a.upb, &a.base, &a.nstrips);
    a.x = reinterpret_cast<const float*>(d->x0);
    a.dy = reinterpret_cast<const float*>(d->dy);
    a.partial = partial;
    a.x_ld = d->x0_ld; a.dy_ld = d->dy_ld;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
    a.nCo = d->Cout / 64;
    a.KD = d->is3d ? 3 : 1;
    const long long grid = (long long)a.base * ns;
    MIS_REQUIRE(grid < (1ll << 31), MIS_EUNSUPPORTED, "wgrad_f32: grid too large");
    if (d->Cin % 64 != 0) {
        if (d->is3d) {
            *tag = "k3.3d.f32s32";
            hipLaunchKernelGGL((wgrad_f32_stream_kernel<true, true>), dim3((unsigned)grid), dim3(256), WfGeom<true>::LDS, stream, a);
        } else {
            *tag = "k3.2d.f32s32";
            hipLaunchKernelGGL((wgrad_f32_stream_kernel<false, true>), dim3((unsigned)grid), dim3(256), WfGeom<true>::LDS, stream, a);
        }
    } else if (d->is3d) {
        *tag = "k3.3d.f32s";
        hipLaunchKernelGGL((wgrad_f32_stream_kernel<true, false>), dim3((unsigned)grid), dim3(256), WfGeom<false>::LDS, stream, a);
    } else {
        *tag = "k3.2d.f32s";
        hipLaunchKernelGGL((wgrad_f32_stream_kernel<false, false>), dim3((unsigned)grid), dim3(256), WfGeom<false>::LDS, stream, a);
    }
    MIS_LAUNCH_CHECK("wgrad_f32");
    return MIS_OK;
}
