// fp32 3x3x3 convolution (forward / dgrad) of the fused 3-D engine on v_mfma_f32_16x16x4_f32 for gfx950 - round 5.
//
// Why a kernel of its own: an f32 MFMA occupies the matrix pipe for 32 cycles per 2048 FLOP (1/16 of the bf16 rate), so per MFMA there is time for everything
// else eight times over - the lock-step kernel of round 1 (conv_igemm.hip) still left the pipe idle 20 % of the time (PMC: 78-84 % busy), because its waves
// read their fragments AFTER each barrier and the two blocks of a CU fall into step with each other: both read, both wait, both compute.  Here a wave never waits
// for an LDS read or a global load while it has MFMAs left:
//   * operands arrive by LDS-DMA only (buffer_load ... lds: no staging registers, zero padding = out-of-range offsets), so the input must be a plain tensor -
//     the fp32 engine now materialises the GroupNorm output once per SingleConv like the bf16 engines (mis_gn_apply), concat and nearest upsample included;
//   * output tile = 8 rows x 16 columns of ONE depth plane x BN (64 | 128) channels; K loop = (32-channel chunk, depth slice dz) groups of nine taps; per group one
//     halo image of plane z + dz - 1 (10 x 18 px x 128 B, double-buffered, fetched in six pieces per wave under the previous group's taps), per tap one weight
//     tile [BN x 128 B] (double-buffered, fetched a tap ahead);
//   * a tap = two half-steps (16 of the chunk's 32 channels each) of NF*PF*4 MFMAs; the fragments of half-step h + 1 are read into the OTHER register set before
//     the MFMAs of half-step h issue, and the block's one barrier per tap sits BETWEEN the two half-steps: behind it the next tap's weight tile is visible, the
//     current one is free for the tap after next, and the first fragments of the next tap are already in flight while 32-64 MFMAs are still to issue;
//   * the MFMAs of a half-step walk the accumulators round-robin (k outer), so none waits for its own result (40-cycle dependent latency against 32 of issue);
//   * 256-thread blocks, two per CU: barrier skew and tile prologues / epilogues of one block are covered by the other.
// LDS images (both conflict-free for ds_read_b128 by enumeration over its lane groups, MI355X_MICROARCH.md §LDS): halo [pixel][128 B] with the 16-byte chunk position
// XORed with (halo column & 7); weights [row][128 B] with the position XORed with (row & 7), rows permuted so that a lane ends up with 4*NF consecutive channels.
//
// Replaces aten::convolution / convolution_backward (input part) for nn.Conv3d(k3, p1, bias=False) (reference model/unet3d/buildingblocks.py:64-66).
#include "common.hpp"
#include "conv_args.hpp"
#include "conv_pp_common.hpp"
#include "dispatch_cfg.hpp"

struct F3Args {
    const float* x;
    const float* w;
    float* y;
    const float* mask;               // ReLU mask applied to the output (out = mask > 0 ? out : 0) or nullptr
    int x_ld, y_ld, mask_ld;
    int N, D, H, W, Cin, Cout;
    int tilesH, tilesW, nCt;
    int relu;
    // per-channel sums of the tile in the epilogue (round 6; MisConvDesc.st_mode): 1 = sum out, sum out * x (the two reductions of the GroupNorm backward, x = the tensor the
    // GroupNorm read: columns [0, sc0) from sx0, the rest from sx1 - on the half grid when sup), 2 = sum out, sum out^2 (the statistics of the next GroupNorm);
    // spart[spatial tile][row half wm][S1 | S2][Cout]
    int smode;
    const float* sx0;
    const float* sx1;
    int sx0_ld, sx1_ld, sc0, sup;
    float* spart;
};

namespace {
constexpr int F3_TH = 8, F3_TW = 16, F3_HW = 18, F3_HP = 10 * 18;
constexpr int F3_HINSTR = 23;                    // 180 px x 128 B = 23,040 B: 22.5 DMA instructions
constexpr int F3_HBUF = F3_HINSTR * 1024;


// LDS-DMA with a scalar offset on top of the per-lane one (the weight tiles: nothing here relies on the range check).  A __device__ function, not a call in the
// kernel body: the builtin does not exist for the host pass, and a kernel TEMPLATE whose body names it there is silently not instantiated (undefined stub at load time)
__device__ __forceinline__ void f3_dma16s(__amdgpu_buffer_rsrc_t r, int voff, int soff, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (pp_lds_void_t*)lds_dst_wave_uniform, 16, voff, soff, 0, 0);
}

template <int OFF> __device__ __forceinline__ u32x4 f3_read(uint32_t addr) { return pp_lds_read128<OFF>(addr); }

// the wait that ends a fragment prefetch: ties the registers to the statement so that no use (and no copy) of them can move above it
template <int N> __device__ __forceinline__ void f3_wait_lgkm(u32x4 (&r)[N]) {
    if constexpr (N == 5)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4])::"memory");
    else if constexpr (N == 6)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5])::"memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])::"memory");
}

// sum over the 16 lanes of a DPP row (every lane ends up with the total; fixed order)
template <int CTRL> __device__ __forceinline__ float f3_dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float f3_row16_sum(float v) {
    v = f3_dpp_add<0xB1>(v);          // quad_perm [1, 0, 3, 2]
    v = f3_dpp_add<0x4E>(v);          // quad_perm [2, 3, 0, 1]
    v = f3_dpp_add<0x141>(v);         // row_half_mirror
    return f3_dpp_add<0x140>(v);      // row_mirror
}
__device__ __forceinline__ float f3_f(const u32x4& v, int t) {
    const uint32_t u = v[t];
    return __uint_as_float(u);
}
}   // namespace

template <int NF>
__global__ __launch_bounds__(256, 2) void conv3d_f32_kernel(const F3Args a) {
    constexpr int PF = 4, WAVE_N = NF * 16, BN = 2 * WAVE_N, NV = 4 * NF;
    constexpr int WT = BN * 128;                 // bytes of one tap's weight tile
    constexpr int WPW = BN / 32;                 // weight DMA instructions per wave and tap (BN * 8 slots / 64 lanes / 4 waves)
    constexpr int NFR = NF + PF;                 // fragments per half-step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds_w = (uint32_t)(uintptr_t)smem;                  // two weight tiles first, then two halo images
    const uint32_t lds_h = lds_w + 2 * WT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;

    // tile: column tile fastest (the column tiles of a spatial tile share its halo in L2), then x, y, plane
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = v % a.nCt;
    int sp = v / a.nCt;
    const int tx = sp % a.tilesW;
    sp /= a.tilesW;
    const int ty = sp % a.tilesH;
    const int pz = sp / a.tilesH;                // plane = n * D + z
    const int z = pz % a.D;
    const int y0 = ty * F3_TH, x0 = tx * F3_TW, n0 = ct * BN;
    const int dz_lo = z == 0 ? 1 : 0, dz_hi = z == a.D - 1 ? 1 : 2;     // depth slices inside the volume (D == 1: only the centre)
    const int nch = a.Cin >> 5;
    const int G = (dz_hi - dz_lo + 1) * nch;

    // ---- DMA lane parts (the tile is fixed for the block's lifetime) ----
    int hoff[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int s = (wave + 4 * k) * 64 + lane;
        const int p = s >> 3, pos = s & 7;
        const int r = p / F3_HW, c = p - r * F3_HW;
        const int yy = y0 - 1 + r, xx = x0 - 1 + c;
        const bool ok = p < F3_HP && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        hoff[k] = ok ? ((yy * a.W + xx) * a.x_ld + ((pos ^ (c & 7)) << 2)) * 4 : PP_OOB;
    }
    int woff[WPW];
#pragma unroll
    for (int k = 0; k < WPW; ++k) {
        const int s = (wave * WPW + k) * 64 + lane;
        const int lrow = s >> 3, pos = s & 7;
        const int wv = lrow / WAVE_N, j = lrow % WAVE_N;
        const int ff = j >> 4, aa = (j & 15) >> 2, bb = j & 3;
        const int co_l = wv * WAVE_N + aa * NV + ff * 4 + bb;
        woff[k] = (co_l * a.Cin + ((pos ^ (lrow & 7)) << 2)) * 4;
    }
    const __amdgpu_buffer_rsrc_t rw = pp_make_rsrc(a.w, (unsigned)((size_t)27 * a.Cout * a.Cin * 4));
    const unsigned plane_bytes = (unsigned)((((size_t)a.H * a.W - 1) * a.x_ld + 32) * 4);
    const size_t plane_stride = (size_t)a.H * a.W * a.x_ld;

    // group g -> (chunk, dz), dz fastest: per output element the products are summed in the order of the lock-step kernel this one replaces (chunk, kd, kh, kw, channel) -
    // bit-identical outputs.  Per group two scalars: the weight offset of its tap 0 (`ws`) and the buffer resource of its halo plane / chunk (`rx`);
    // those of group g + 1 are derived by counters inside group g's first MFMA cluster, so that no address arithmetic stands between a barrier and an MFMA.
    const unsigned tapstride = (unsigned)a.Cout * a.Cin * 4u;
    auto ws_of = [&](int dz, int c0) { return (unsigned)((((unsigned)(dz * 9) * a.Cout + n0) * a.Cin + c0) * 4u); };
    auto rx_of = [&](int dz, int c0) { return pp_make_rsrc(a.x + (size_t)(pz + dz - 1) * plane_stride + c0, plane_bytes); };
    auto issue_halo_piece = [&](const __amdgpu_buffer_rsrc_t& rx, int buf, int k) {
        if (wave + 4 * k < F3_HINSTR) pp_dma16(rx, hoff[k], smem + 2 * WT + buf * F3_HBUF + (wave + 4 * k) * 1024);          // (wave-uniform)
    };
    auto issue_w = [&](unsigned ws, int tap, int par) {
        const int soff = (int)(ws + (unsigned)tap * tapstride);
#pragma unroll
        for (int k = 0; k < WPW; ++k)
            f3_dma16s(rw, woff[k], soff, smem + par * WT + (wave * WPW + k) * 1024);
    };

    // ---- fragment addresses ----
    uint32_t a_lane[2], b_lane[3][2];
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
        const int ch = kg * 4 + lg;
        a_lane[kg] = (uint32_t)((wn * WAVE_N + li) * 128 + ((ch ^ (li & 7)) << 4));
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int c = li + kw;
            b_lane[kw][kg] = (uint32_t)(((wm * PF) * F3_HW + c) * 128 + ((ch ^ (c & 7)) << 4));
        }
    }

    f32x4 acc[NF][PF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) acc[f][pf] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment reads of half-step (tap, kg) of the group whose halo image is at hb and whose weight tile is at wb
    auto read_frags = [&](u32x4(&r)[NFR], uint32_t wb, uint32_t hb, auto tapc, auto kgc) {
        constexpr int tap = decltype(tapc)::value, kg = decltype(kgc)::value;
        constexpr int kh = tap / 3, kw = tap % 3;
        pp_static_for<NF>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            r[f] = f3_read<f * 16 * 128>(wb + a_lane[kg]);
        });
        pp_static_for<PF>([&](auto pc) {
            constexpr int pf = decltype(pc)::value;
            r[NF + pf] = f3_read<(pf + kh) * F3_HW * 128>(hb + b_lane[kw][kg]);
        });
    };
    // the MFMAs of k sub-steps [T0, T1) of a half-step: accumulators round-robin
    auto mfmas = [&](const u32x4(&r)[NFR], auto t0c, auto t1c) {
        constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
#pragma unroll
        for (int t = T0; t < T1; ++t)
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) acc[f][pf] = __builtin_amdgcn_mfma_f32_16x16x4f32(f3_f(r[f], t), f3_f(r[NF + pf], t), acc[f][pf], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I4 = std::integral_constant<int, 4>;

    // ---- prologue: halo 0, weight tiles 0 and 1 ----
    int dz = dz_lo, c0 = 0;                      // the CURRENT group's depth slice / first channel
    unsigned ws = ws_of(dz, 0);
    {
        const __amdgpu_buffer_rsrc_t rx0 = rx_of(dz, 0);
#pragma unroll
        for (int k = 0; k < 6; ++k) issue_halo_piece(rx0, 0, k);
    }
    issue_w(ws, 0, 0);
    issue_w(ws, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // (a read that starts within a few dozen cycles of the barrier can still see the bytes an LDS-DMA of ANOTHER wave has just landed as their old contents -
    //  wgrad_pp.hip; every later barrier of this kernel is followed by DMA issues and MFMAs before the first dependent read)
    __builtin_amdgcn_s_sleep(2);

    u32x4 s0[NFR], s1[NFR];
    read_frags(s0, lds_w, lds_h, I0{}, I0{});
    f3_wait_lgkm(s0);
    __builtin_amdgcn_sched_barrier(0);

    int par = 0;                                 // slot of the current tap's weight tile
    unsigned ws_next = 0;
    __amdgpu_buffer_rsrc_t rx_next = rx_of(dz, 0);
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        const uint32_t hb = lds_h + (g & 1) * F3_HBUF, hb_next = lds_h + ((g & 1) ^ 1) * F3_HBUF;
        const bool more = g + 1 < G;
        pp_static_for<9>([&](auto tc) {
            constexpr int tap = decltype(tc)::value;
            const uint32_t wb = lds_w + par * WT, wb_next = lds_w + (par ^ 1) * WT;
            // ---- first half: the second half's fragments fly under its MFMAs ----
            read_frags(s1, wb, hb, tc, I1{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(s0, I0{}, I1{});
            if constexpr (tap == 0) {
                __builtin_amdgcn_sched_barrier(0);
                ++dz;                            // the next group's scalars (harmless values past the last group)
                if (dz > dz_hi) {
                    dz = dz_lo;
                    c0 += 32;
                }
                ws_next = ws_of(dz, more ? c0 : 0);
                rx_next = rx_of(dz, more ? c0 : 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            mfmas(s0, I1{}, I4{});
            __builtin_amdgcn_sched_barrier(0);
            f3_wait_lgkm(s1);
            // the next tap's weight tile (issued behind the previous barrier) and every halo piece issued so far
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- behind the barrier: this tap's weight tile is free -> the tap after next; a piece of the next group's halo; the next tap's first fragments -
            //      all of it inside the MFMA cluster of the second half, whose fragments are in registers already ----
            if constexpr (tap < 7) {
                issue_w(ws, tap + 2, par);
            } else {
                if (more) issue_w(ws_next, tap - 7, par);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas(s1, I0{}, I1{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (tap < 6) {
                if (more) issue_halo_piece(rx_next, (g & 1) ^ 1, tap);
            }
            if constexpr (tap < 8) {
                read_frags(s0, wb_next, hb, std::integral_constant<int, tap + 1>{}, I0{});
            } else {
                if (more) read_frags(s0, wb_next, hb_next, I0{}, I0{});
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas(s1, I1{}, I4{});
            __builtin_amdgcn_sched_barrier(0);
            f3_wait_lgkm(s0);
            par ^= 1;
        });
        ws = ws_next;
    }

    // ---- epilogue: lane (li, lg) holds, per pixel row pf, NV consecutive channels of pixel (y0 + wm*4 + pf, x0 + li) ----
    const int x = x0 + li;
    const int col = n0 + wn * WAVE_N + lg * NV;
    float st1[NV], st2[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) st1[i] = st2[i] = 0.f;
    // the x source of this wave's columns (wave-uniform: sc0 is a multiple of 64 >= WAVE_N)
    const bool sfrom1 = a.smode == 1 && a.sx1 != nullptr && n0 + wn * WAVE_N >= a.sc0;
    const float* const sxp = sfrom1 ? a.sx1 + (col - a.sc0) : a.sx0 + col;
    const int sxld = sfrom1 ? a.sx1_ld : a.sx0_ld;
    const bool shalf = sfrom1 && a.sup;
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const int y = y0 + wm * PF + pf;
        if (y < a.H && x < a.W) {
            const size_t pix = ((size_t)pz * a.H + y) * a.W + x;
            float* dst = a.y + pix * a.y_ld + col;
            f32x4 sxv[NF];
            if (a.smode == 1) {
                const size_t spix = shalf ? ((size_t)((pz / a.D) * (a.D >> 1) + (z >> 1)) * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1) : pix;
#pragma unroll
                for (int f = 0; f < NF; ++f) sxv[f] = *reinterpret_cast<const f32x4*>(sxp + spix * sxld + f * 4);
            }
            f32x4 mk[NF];
            if (a.mask != nullptr) {
#pragma unroll
                for (int f = 0; f < NF; ++f) mk[f] = *reinterpret_cast<const f32x4*>(a.mask + pix * a.mask_ld + col + f * 4);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                f32x4 o = acc[f][pf];
                if (a.relu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = fmaxf(o[i], 0.f);
                }
                if (a.mask != nullptr) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = mk[f][i] > 0.f ? o[i] : 0.f;
                }
                *reinterpret_cast<f32x4*>(dst + f * 4) = o;
                if (a.smode != 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        st1[f * 4 + i] += o[i];
                        st2[f * 4 + i] = fmaf(o[i], a.smode == 1 ? sxv[f][i] : o[i], st2[f * 4 + i]);
                    }
                }
            }
        }
    }
    if (a.smode != 0) {
        // sum over the 16 pixel columns (lanes li of one lg: butterfly inside 16-lane groups), fixed order; lane li == 0 stores the wave's NV channels of both sums
        // (DPP adds: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror - one VALU instruction per step and value instead of a trip through the LDS crossbar)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            st1[i] = f3_row16_sum(st1[i]);
            st2[i] = f3_row16_sum(st2[i]);
        }
        if (li == 0) {
            float* const row = a.spart + ((size_t)(v / a.nCt) * 2 + wm) * 2 * a.Cout + col;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                *reinterpret_cast<f32x4*>(row + f * 4) = f32x4{st1[f * 4], st1[f * 4 + 1], st1[f * 4 + 2], st1[f * 4 + 3]};
                *reinterpret_cast<f32x4*>(row + a.Cout + f * 4) = f32x4{st2[f * 4], st2[f * 4 + 1], st2[f * 4 + 2], st2[f * 4 + 3]};
            }
        }
    }
}

bool conv3d_f32_eligible(const MisConvDesc* d) {
    if (d->dtype != MIS_F32 || !d->is3d || d->ksize != 3) return false;
    if (d->x1 != nullptr || d->in_scale != nullptr || d->bias != nullptr || d->mask_bits != nullptr || d->relu_bits != nullptr || d->gn_p != nullptr) return false;
    if (d->y1 != nullptr || d->Cout0 != d->Cout || d->Cin0 != d->Cin || d->y0_mode != MIS_OUT_PLAIN) return false;
    if (d->x0_D != d->D || d->x0_H != d->H || d->x0_W != d->W) return false;
    if (d->Cin % 32 != 0 || d->Cout % 32 != 0 || d->x0_ld % 4 != 0 || d->y0_ld % 4 != 0 || (d->mask != nullptr && d->mask_ld % 4 != 0)) return false;
    if (((size_t)d->H * d->W * d->x0_ld + 64) * 4 >= 0xFFFF0000ull) return false;             // one input plane per buffer resource, 32-bit offsets
    if ((size_t)27 * d->Cout * d->Cin * 4 >= 0xFFFF0000ull) return false;
    return true;
}

// The partial rows of the statistics epilogue -> S1 / S2 [N][Cout], in DOUBLE (the GroupNorm backward subtracts nearly equal sums: an fp32 reduction over 32 k rows cost the
// deconv golden 1e-3 on the gradients behind it) and in a fixed order.  Stage 1: block (16 columns, sample, row slice z of CSR_Z) x 16 row lanes, four independent loads in
// flight per thread; stage 2: one thread per (sample, column) adds the CSR_Z slice sums.
constexpr int CSR_Z = 64;
__global__ __launch_bounds__(256) void conv_stats_reduce1_kernel(const float* __restrict__ part, long long rows, int C2, double* __restrict__ ws /*[N][CSR_Z][C2]*/) {
    __shared__ double red[16][17];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl, n = blockIdx.y, z = blockIdx.z;
    const long long per = (rows + CSR_Z - 1) / CSR_Z;
    const long long r0 = (long long)z * per, r1 = r0 + per < rows ? r0 + per : rows;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (c < C2) {
        const float* p = part + (size_t)n * rows * C2 + c;
        long long r = r0 + g;
        for (; r + 48 < r1; r += 64) {
            const float v0 = p[(size_t)r * C2], v1 = p[(size_t)(r + 16) * C2], v2 = p[(size_t)(r + 32) * C2], v3 = p[(size_t)(r + 48) * C2];
            a0 += (double)v0; a1 += (double)v1; a2 += (double)v2; a3 += (double)v3;
        }
        for (; r < r1; r += 16) a0 += (double)p[(size_t)r * C2];
    }
    red[g][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && c < C2) {
        double t = red[0][cl];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][cl];
        ws[((size_t)n * CSR_Z + z) * C2 + c] = t;
    }
}
__global__ __launch_bounds__(256) void conv_stats_reduce2_kernel(const double* __restrict__ ws, int N, int C2, int Cout, float* __restrict__ S1, float* __restrict__ S2) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * C2) return;
    const int n = idx / C2, c = idx - n * C2;
    double t = 0.0;
#pragma unroll 8
    for (int z = 0; z < CSR_Z; ++z) t += ws[((size_t)n * CSR_Z + z) * C2 + c];
    if (c < Cout) S1[(size_t)n * Cout + c] = (float)t;
    else S2[(size_t)n * Cout + (c - Cout)] = (float)t;
}

extern "C" size_t mis_conv_stats_reduce_workspace_bytes(int N, int Cout) { return (size_t)N * CSR_Z * 2 * Cout * sizeof(double); }

extern "C" int mis_conv_stats_reduce(const float* part, int N, long long rows, int Cout, void* workspace, float* S1, float* S2, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(part && workspace && S1 && S2 && N > 0 && N <= 65535 && rows > 0 && Cout > 0 && Cout % 16 == 0, MIS_EINVAL, "conv_stats_reduce: bad argument");
    MIS_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, MIS_EINVAL, "conv_stats_reduce: workspace alignment");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(conv_stats_reduce1_kernel, dim3((unsigned)(2 * Cout / 16), (unsigned)N, CSR_Z), dim3(256), 0, s, part, rows, 2 * Cout, reinterpret_cast<double*>(workspace));
    MIS_LAUNCH_CHECK("conv_stats_reduce1");
    hipLaunchKernelGGL(conv_stats_reduce2_kernel, dim3((unsigned)((N * 2 * Cout + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const double*>(workspace), N, 2 * Cout, Cout, S1, S2);
    MIS_LAUNCH_CHECK("conv_stats_reduce2");
    return MIS_OK;
}

// rows per sample of the statistics partials (MisConvDesc.st_part): one per (spatial tile, row half of the tile)
long long conv3d_f32_stats_rows(const MisConvDesc* d) { return (long long)d->D * ((d->H + F3_TH - 1) / F3_TH) * ((d->W + F3_TW - 1) / F3_TW) * 2; }

int launch_conv3d_f32(const MisConvDesc* d, hipStream_t stream, const char** tag) {
    F3Args a;
    a.x = reinterpret_cast<const float*>(d->x0);
    a.w = reinterpret_cast<const float*>(d->w);
    a.y = reinterpret_cast<float*>(d->y0);
    a.mask = reinterpret_cast<const float*>(d->mask);
    a.x_ld = d->x0_ld; a.y_ld = d->y0_ld; a.mask_ld = d->mask_ld;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
    a.tilesH = (d->H + F3_TH - 1) / F3_TH;
    a.tilesW = (d->W + F3_TW - 1) / F3_TW;
    a.relu = d->relu;
    a.smode = d->st_mode;
    a.sx0 = reinterpret_cast<const float*>(d->st_x0); a.sx1 = reinterpret_cast<const float*>(d->st_x1);
    a.sx0_ld = d->st_x0_ld; a.sx1_ld = d->st_x1_ld; a.sc0 = d->st_x1 != nullptr ? d->st_c0 : d->Cout; a.sup = d->st_up;
    a.spart = d->st_part;
    if (d->st_mode != 0) {
        MIS_REQUIRE((d->st_mode == 1 || d->st_mode == 2) && d->st_part != nullptr, MIS_EINVAL, "conv3d_f32: st_mode %d / st_part", d->st_mode);
        if (d->st_mode == 1) {
            MIS_REQUIRE(d->st_x0 != nullptr && d->st_x0_ld % 4 == 0, MIS_EINVAL, "conv3d_f32: st_x0");
            MIS_REQUIRE(d->st_x1 == nullptr || (d->st_x1_ld % 4 == 0 && d->st_c0 > 0 && d->st_c0 < d->Cout && d->st_c0 % 64 == 0), MIS_EINVAL, "conv3d_f32: st_x1 / st_c0");
            MIS_REQUIRE(d->st_x1 == nullptr || !d->st_up || (d->D % 2 == 0 && d->H % 2 == 0 && d->W % 2 == 0), MIS_EUNSUPPORTED, "conv3d_f32: st_up needs an even grid");
        }
    }
    // 128-column tiles unless they leave the chip underfilled (two 256-thread blocks per CU = 512 slots): the 16^3 level of cfg4 has 64 spatial tiles - 128 / 256 blocks of
    // 128 columns ran at 72 TFLOP/s; 64-column tiles double the blocks (same summation order per output element: bit-identical)
    const long long sp = (long long)d->N * d->D * a.tilesH * a.tilesW;
    const bool wide = d->Cout % 128 == 0 && (sp * (d->Cout / 128) >= 384 || mis_sw(SW_CONV3D_F32_WIDE));          // (MIS_CONV3D_F32_WIDE=1: 128 columns whatever the grid - tests)
    // Cout = 32 (mod 64): 32-column tiles (round 6: the dgrad of encoders.0 SingleConv2, whose input has 32 real channels - buildingblocks.py:202-211 - ran on 64 columns,
    // half of them weights of zero)
    const bool narrow = d->Cout % 64 != 0;
    a.nCt = d->Cout / (wide ? 128 : narrow ? 32 : 64);
    const long long grid = (long long)d->N * d->D * a.tilesH * a.tilesW * a.nCt;
    MIS_REQUIRE(grid < (1ll << 31), MIS_EUNSUPPORTED, "conv3d_f32: grid too large");
    static std::atomic<unsigned long long> attr_done[3] = {{0}, {0}, {0}};
    if (wide) {
        *tag = "k3.3d.f32pp128";
        constexpr size_t lds = (size_t)2 * 128 * 128 + 2 * F3_HBUF;
        if (const int rc = mis_set_dyn_lds(attr_done[1], reinterpret_cast<const void*>(&conv3d_f32_kernel<4>), lds, "conv3d_f32<4>")) return rc;
        hipLaunchKernelGGL(conv3d_f32_kernel<4>, dim3((unsigned)grid), dim3(256), lds, stream, a);
    } else if (narrow) {
        *tag = "k3.3d.f32pp32";
        constexpr size_t lds = (size_t)2 * 32 * 128 + 2 * F3_HBUF;
        if (const int rc = mis_set_dyn_lds(attr_done[2], reinterpret_cast<const void*>(&conv3d_f32_kernel<1>), lds, "conv3d_f32<1>")) return rc;
        hipLaunchKernelGGL(conv3d_f32_kernel<1>, dim3((unsigned)grid), dim3(256), lds, stream, a);
    } else {
        *tag = "k3.3d.f32pp64";
        constexpr size_t lds = (size_t)2 * 64 * 128 + 2 * F3_HBUF;
        if (const int rc = mis_set_dyn_lds(attr_done[0], reinterpret_cast<const void*>(&conv3d_f32_kernel<2>), lds, "conv3d_f32<2>")) return rc;
        hipLaunchKernelGGL(conv3d_f32_kernel<2>, dim3((unsigned)grid), dim3(256), lds, stream, a);
    }
    MIS_LAUNCH_CHECK("conv3d_f32");
    return MIS_OK;
}
