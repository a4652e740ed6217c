// Ping-pong 1x1 GEMM for the bf16 2-D layers (gfx950): y[pixel][col] = sum_ci x[pixel][ci] * w[col][ci] - the two GEMMs of nn.ConvTranspose2d(k2, s2) (reference
// model/unet2d/layers.py:165: forward = Cin -> 4*Cq columns stored pixel-shuffled into the concat buffer, dgrad = 4*Cq -> Cin on the pixel-unshuffled gradient with the
// ReLU mask of the layer below), same operand packing and epilogue arithmetic as conv_igemm.hip's ksize-1 path.
//
// Why another kernel: the generic kernel runs these GEMMs at 3.3-3.6 TB/s where they are bandwidth-bound (the two shallow levels: 1.6 / 0.8 GB per launch; a copy kernel
// reaches 5.4 TB/s on the same box, scripts/bw_probe.py) and at 0.4-0.65 PFLOP/s where they are not (the deep levels: lock-step waves, one tile per block).  This is the
// column-segment kernel's machinery (conv_pp.hip) without the filter window: persistent blocks, 512 pixels (32 rows x 16 columns) x 128 columns per block, wave tile
// 128 px x 64 columns (PF 8 x NF 4), two wave groups staggered by one barrier (R = 12 fragment reads + DMA issue, M = 32 MFMAs), everything staged by LDS-DMA.
// A 1x1 GEMM reuses a staged pixel only for the block's 128 columns (a 3x3 conv: 9 taps x 128), so there is five times the DMA traffic per MFMA: the K loop is a flat
// sequence of (tile, 32-channel chunk) steps over a THREE-stage LDS ring (40 KiB per stage: 512 px x 64 B + 128 weight rows x 64 B), each step's stage issued two steps
// ahead - across tile boundaries - and retired by a counted vmcnt that leaves exactly the step's own five instructions in flight.
//
// Ordering (same rules as conv_pp.hip): stage (g+2) % 3 was last read in step g-1, whose reads both groups finished (lgkmcnt(0)) before the barriers that precede
// their R(g); a wave's DMAs of step g+1 are retired by its vmcnt in R(g), published by that segment's barrier, and first read in R(g+1), two barriers later.
#include <stdlib.h>

#include "conv_pp_common.hpp"
#include "dispatch_cfg.hpp"

namespace {
constexpr int G1_PF = 8, G1_NF = 4, G1_WAVE_N = 64, G1_BN = 128, G1_TH = 32, G1_TW = 16;
constexpr int G1_PXB = G1_TH * G1_TW * 64;          // pixel tile of one 32-channel chunk: 512 px x 64 B (32 DMA instructions: one per tile row)
constexpr int G1_WB = G1_BN * 64;                   // weight tile: 128 columns x 64 B (8 DMA instructions: one per wave)
constexpr int G1_STAGE = G1_PXB + G1_WB;            // 40 KiB
constexpr int G1_NST = 3;
constexpr int G1_PER_WAVE = 5;                      // DMA instructions per wave and step
}   // namespace

template <int EM>
__global__ __launch_bounds__(512, 2) void gemm1_pp_kernel(const ConvArgs a) {
    using T = __bf16;
    constexpr int PF = G1_PF, NF = G1_NF, BN = G1_BN, WAVE_N = G1_WAVE_N, NV = 4 * NF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sbase = smem;                                // G1_NST stages: [pixel tile | weight tile]
    char* const bbase = smem + G1_NST * G1_STAGE;            // 2 x BN floats: bias slices by tile parity
    char* const mlds = bbase + 2 * BN * 4;                   // EM == PP_EM_BITS: 8 x 1 KiB, each wave's ReLU bits of the current tile (16 bytes per lane), staged by ONE
                                                             // LDS-DMA per wave during the tile's first step (a register load inside this flat step loop would be
                                                             // loop-carried, and hipcc guards such a load with s_waitcnt vmcnt(0) at the loop header)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    const int li = lane & 15, lg = lane >> 4;

    const int total_tiles = a.nSp * a.nCt;
    const int tstride = (int)gridDim.x;
    const int tile0 = xcd_remap(blockIdx.x, gridDim.x);
    if (tile0 >= total_tiles) return;                        // block-uniform
    const int nch = a.Cin >> 5;
    const int ntiles_mine = (total_tiles - tile0 + tstride - 1) / tstride;
    const int G = ntiles_mine * nch;                         // this block's steps
    const int tpi = a.tilesH * a.tilesW;
    // spatial major: the nCt column tiles of a spatial tile are neighbours in the tile order (they share the pixel tile in L2); the persistent stride is a multiple of nCt
    // whenever 256 % nCt == 0, so a block keeps its column tile
    auto decode = [&](int t, int& tn, int& th0, int& tw0, int& tcol) {
        const int sp = t / a.nCt, ct = t - sp * a.nCt;
        tn = sp / tpi;
        const int r = sp - tn * tpi;
        const int th = r / a.tilesW;
        th0 = th * G1_TH;
        tw0 = (r - th * a.tilesW) * G1_TW;
        tcol = ct * BN;
    };

    // ---- fragment offsets inside a stage (64-byte rows, 16-byte chunk position XORed with (row >> 1) & 3 / (pixel column >> 1) & 3: conv_pp.hip) ----
    const int a_off0 = G1_PXB + (wn * WAVE_N + li) * 64 + ((lg ^ ((li >> 1) & 3)) << 4);
    const int b_off0 = (wm * PF * G1_TW + li) * 64 + ((lg ^ ((li >> 1) & 3)) << 4);
    // ---- DMA source offsets.  Pixel instruction r = tile row r: lane -> (column lane >> 2, chunk position lane & 3).  Weight instruction = LDS rows 16*wave .. +15. ----
    const int pcol = lane >> 2;
    const unsigned prel = (unsigned)((pcol * a.x0.ld + (((lane & 3) ^ ((pcol >> 1) & 3)) << 3)) * 2);
    int w_goff0;
    {
        const int slot = wave * 64 + lane;
        const int lrow = slot >> 2, pos = slot & 3;
        const int dc16 = pos ^ ((lrow >> 1) & 3);
        const int dwv = lrow / WAVE_N, j = lrow % WAVE_N;
        const int drow = dwv * WAVE_N + ((j >> 5) * 32) + ((j & 15) >> 2) * 8 + ((j >> 4) & 1) * 4 + (j & 3);      // channel order of pp_epilogue_plain
        w_goff0 = (drow * a.Cin + dc16 * 8) * 2;
    }
    const unsigned img_x = (unsigned)(((long long)a.H * a.W - 1) * a.x0.ld + a.Cin) * 2u;
    const char* const xb = reinterpret_cast<const char*>(a.x0.p);
    const __amdgpu_buffer_rsrc_t rw = pp_make_rsrc(a.w, (unsigned)((long long)a.Cout * a.Cin * 2));
    const int shuffled = a.y0_mode == MIS_OUT_SHUFFLE2;
    const int cq = a.Cout >> 2;
    const __amdgpu_buffer_rsrc_t rb = pp_make_rsrc(a.bias != nullptr ? (const void*)a.bias : a.w, a.bias != nullptr ? (unsigned)(shuffled ? cq : a.Cout) * 4u : 0u);

    // the five DMA instructions of this wave for step (tile (n, h0, w0, col), chunk c0) into `stage`
    auto issue_step = [&](int n, int h0, int w0, int col, int c0, char* stage) {
        const __amdgpu_buffer_rsrc_t rx = pp_make_rsrc(xb + (size_t)n * a.H * a.W * a.x0.ld * 2, img_x);
        const bool colok = w0 + pcol < a.W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wave + 8 * j;
            const unsigned toff = (unsigned)((((h0 + r) * a.W + w0) * a.x0.ld + c0) * 2);
            pp_dma16(rx, (colok && h0 + r < a.H) ? (int)(toff + prel) : PP_OOB, stage + r * 1024);
        }
        pp_dma16(rw, (int)(((long long)col * a.Cin + c0) * 2) + w_goff0, stage + G1_PXB + wave * 1024);
    };
    // bias slice of a column tile -> LDS, 4 bytes per lane (waves 0, 1); pixel-shuffled outputs index the bias by the real channel c = column % Cq
    auto issue_bias = [&](int col, char* dst) {
        if (wave < BN / 64) {
            int l;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            int c = col + wave * 64 + l;
            if (shuffled) c = c % cq;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (pp_lds_void_t*)(dst + wave * 256), 4, c * 4, 0, 0, 0);
        }
    };

    f32x4 acc[NF][PF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) acc[f][pf] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- the step sequence: cursor P = the step whose stage is issued next (two ahead of the step being computed) ----
    int ptile = tile0, pchunk = 0, pn, ph0, pw0, pcol0;
    decode(ptile, pn, ph0, pw0, pcol0);
    auto advance_p = [&]() {
        if (++pchunk == nch) {
            pchunk = 0;
            ptile += tstride;
            if (ptile < total_tiles) decode(ptile, pn, ph0, pw0, pcol0);
        }
    };
    int n, h0, w0, ncol0;
    decode(tile0, n, h0, w0, ncol0);
    issue_bias(ncol0, bbase);
    issue_step(pn, ph0, pw0, pcol0, 0, sbase);               // step 0
    advance_p();
    if (G > 1) {
        issue_step(pn, ph0, pw0, pcol0, pchunk << 5, sbase + G1_STAGE);      // step 1
        advance_p();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int tile = tile0, chunk = 0, bsel = 0;
    auto issue_bits = [&]() {          // this wave's bits of tile (n, h0, w0, ncol0) -> its KiB of mlds
        int l_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
        const int bo = pp_bits_voff<NF, PF>(a, n, h0, w0, ncol0, wm, wn, l_ & 15, l_ >> 4);
        pp_dma16(pp_make_rsrc(a.mask_bits, (unsigned)rb_bytes(a.N, a.H, a.W, a.Cout)), bo, mlds + wave * 1024);
    };
    if (grp == 1) __builtin_amdgcn_s_barrier();              // the stagger
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        const int slot = g % G1_NST;
        const uint32_t sb = (uint32_t)(uintptr_t)sbase + slot * G1_STAGE;
        // ================= R segment =================
        const bool more = g + 2 < G;
        if (more) {
            issue_step(pn, ph0, pw0, pcol0, pchunk << 5, sbase + ((g + 2) % G1_NST) * G1_STAGE);
            advance_p();
        }
        u32x4 A[NF], Brow[PF];
        pp_static_for<NF>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            A[f] = pp_lds_read128<f * 1024>(sb + a_off0);
        });
        pp_static_for<PF>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            Brow[r] = pp_lds_read128<r * (G1_TW * 64)>(sb + b_off0);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // step g+1's stage (issued one step ago) has landed once at most THIS step's five instructions are outstanding
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G1_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ================= M segment: NF x PF MFMAs =================
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) mma_b128<T>(acc[f][pf], A[f], Brow[pf]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (++chunk == nch) {
            // this wave's tile is complete: epilogue (bias add, ReLU / mask, stores, accumulators re-armed), then the next tile's bookkeeping
            u32x4 mbits = u32x4{0u, 0u, 0u, 0u};
            if constexpr (EM == PP_EM_BITS) {
                int l_;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
                mbits = pp_lds_read128<0>((uint32_t)(uintptr_t)mlds + wave * 1024 + l_ * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            pp_epilogue_plain<NF, PF, EM>(a, acc, (uint32_t)(uintptr_t)bbase + bsel * (BN * 4), n, h0, w0, ncol0, wm, wn, mbits);
            chunk = 0;
            tile += tstride;
            bsel ^= 1;
            if (tile < total_tiles) decode(tile, n, h0, w0, ncol0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (chunk == 1) {
            if constexpr (EM == PP_EM_BITS) issue_bits();        // (retired by the next step's counted wait: it is older than that step's five instructions)
            // during a tile's first step: the NEXT tile's bias slice -> the other half of the bias region (its last reader, the previous tile's epilogue, is behind us)
            const int nt = tile + tstride;
            if (nt < total_tiles) {
                int bn_, bh_, bw_, bcol_;
                decode(nt, bn_, bh_, bw_, bcol_);
                issue_bias(bcol_, bbase + (bsel ^ 1) * (BN * 4));
            }
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();              // pairs with group 1's last barrier
}

// ---------------------------------------------------------------------------------------------------------
bool gemm1_pp_eligible(const MisConvDesc* d) {
    if (d->dtype != MIS_BF16 || d->is3d || d->ksize != 1 || d->D != 1) return false;
    if (d->x1 != nullptr || d->in_scale != nullptr) return false;
    if (d->x0_H != d->H || d->x0_W != d->W) return false;
    if (d->Cin % 64 != 0 || d->Cout % 128 != 0) return false;             // (Cin: at least two 32-channel steps per tile - the bias prefetch of the next tile rides in the first)
    if (d->Cout0 != d->Cout || d->y1 != nullptr) return false;
    const long long img = (long long)d->H * d->W, lim = (1ll << 32) - 65536;
    if (d->y0_mode == MIS_OUT_PLAIN) {
        if (((img - 1) * d->y0_ld + d->Cout) * 2 >= lim) return false;
    } else if (d->y0_mode == MIS_OUT_SHUFFLE2) {
        const int cq = d->Cout / 4;
        if (cq % 64 != 0 || d->mask != nullptr || d->mask_bits != nullptr || d->relu_bits != nullptr) return false;
        if (((4 * img - 1) * d->y0_ld + cq) * 2 >= lim) return false;
    } else {
        return false;
    }
    if (d->relu_bits != nullptr) return false;                              // (no forward form of the net asks a 1x1 GEMM for ReLU bits)
    if (d->mask != nullptr && ((img - 1) * d->mask_ld + d->Cout) * 2 >= lim) return false;
    if (d->mask_bits != nullptr && (long long)rb_bytes(d->N, d->H, d->W, d->Cout) >= lim) return false;
    if (((img - 1) * d->x0_ld + d->Cin) * 2 >= (1ll << 31) - 65536) return false;
    if ((long long)d->Cout * d->Cin * 2 >= (1ll << 31) - 65536) return false;
    return ((d->H + 31) / 32) * 32 * 100 <= d->H * 115;                     // 32-row tiles must fit the grid
}

template <int EM> static int g1_launch(const MisConvDesc* d, hipStream_t stream) {
    ConvArgs a;
    a.tq = nullptr;
    a.N = d->N; a.D = 1; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.Cin0 = d->Cin; a.Cout0 = d->Cout0;
    a.x0 = SrcView{d->x0, d->x0_ld, 1, d->x0_H, d->x0_W};
    a.x1 = SrcView{nullptr, 0, 0, 0, 0};
    a.in_scale = nullptr; a.in_shift = nullptr;
    a.w = d->w; a.bias = d->bias; a.relu = d->relu; a.mask = d->mask; a.mask_ld = d->mask_ld;
    a.relu_bits = nullptr; a.mask_bits = reinterpret_cast<const unsigned char*>(d->mask_bits);
    a.y0 = d->y0; a.y0_ld = d->y0_ld; a.y0_mode = d->y0_mode;
    a.y1 = nullptr; a.y1_ld = 0; a.y1_mode = 0;
    a.tilesD = 1;
    a.tilesH = (d->H + G1_TH - 1) / G1_TH;
    a.tilesW = (d->W + G1_TW - 1) / G1_TW;
    const long long nsp = (long long)d->N * a.tilesH * a.tilesW;
    a.nCt = d->Cout / G1_BN;
    MIS_REQUIRE(nsp * a.nCt < (1ll << 31), MIS_EUNSUPPORTED, "conv_igemm(k1 pp): grid too large");
    a.nSp = (int)nsp;
    a.order = 0; a.zg = 0;
    const size_t lds = (size_t)G1_NST * G1_STAGE + 2 * (size_t)G1_BN * 4 + (EM == PP_EM_BITS ? 8192 : 0);
    static std::atomic<unsigned long long> attr_done{0};
    if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&gemm1_pp_kernel<EM>), lds, "conv_igemm(k1 pp)")) return rc;
    const long long total = nsp * a.nCt;
    hipLaunchKernelGGL((gemm1_pp_kernel<EM>), dim3((unsigned)(total > mis_persist_cus() ? mis_persist_cus() : total)), dim3(512), lds, stream, a);
    MIS_LAUNCH_CHECK("conv_igemm(k1 pp)");
    return MIS_OK;
}

int launch_gemm1_pp(const MisConvDesc* d, hipStream_t stream, const char** tag) {
    if (d->mask_bits != nullptr) {
        *tag = "k1.2d.pp.bits";
        return g1_launch<PP_EM_BITS>(d, stream);
    }
    if (d->mask != nullptr) {
        *tag = "k1.2d.pp.mask";
        return g1_launch<PP_EM_MASK>(d, stream);
    }
    *tag = "k1.2d.pp";
    return g1_launch<PP_EM_NONE>(d, stream);
}
