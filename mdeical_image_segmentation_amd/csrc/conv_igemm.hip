// Implicit-GEMM convolution on MFMA for gfx950 (NHWC / NDHWC, im2col-free).
//
//   D[channel][pixel] = sum_{tap, ci} Wp[tap][channel][ci] * X[pixel + tap][ci]
//
// The MFMA "A" operand is the weight tile (rows = output channels), "B" is the pixel tile, so that a lane ends up
// holding 4*NF consecutive output channels of ONE pixel and the epilogue stores 16..64 contiguous bytes per lane.
// Per K chunk (128 bytes of input channels per pixel: 64 bf16 / 32 f32) the block stages the input HALO tile once
// into LDS and then walks the taps, streaming only the [BN x 128 B] weight tile of each tap (double-buffered LDS,
// next tile prefetched into registers while the MFMAs of the current tap issue).  Both LDS images are XOR-swizzled
// in 16-byte chunks (chunk ^= row & 7) so every ds_read_b128 of a fragment is bank-conflict free.
//
// Replaces: nn.Conv2d(k3,p1)+bias+ReLU fwd / dgrad (reference model/unet2d/layers.py:122-126),
//           nn.Conv3d(k3,p1) after GroupNorm (model/unet3d/buildingblocks.py:64-66,87-92),
//           nn.ConvTranspose2d(k2,s2) fwd / dgrad as a 1x1 GEMM + pixel (un)shuffle (layers.py:165),
//           torch.cat / F.interpolate(nearest) as input addressing (layers.py:186-190, buildingblocks.py:546-548,671-673).
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "conv_args.hpp"
#include "dispatch_cfg.hpp"
#include "relu_bits.hpp"

#ifndef MIS_WDMA_EXPLICIT_DRAIN
#define MIS_WDMA_EXPLICIT_DRAIN 1
#endif

template <int TD_, int TH_, int TW_, int KS_, bool IS3D_> struct Geom {
    static constexpr int TD = TD_, TH = TH_, TW = TW_, KS = KS_;
    static constexpr bool IS3D = IS3D_;
    static constexpr int PAD = KS / 2;
    static constexpr int PD = IS3D ? PAD : 0;
    static constexpr int KD = IS3D ? KS : 1;
    static constexpr int HD = TD + 2 * PD, HH = TH + 2 * PAD, HW = TW + 2 * PAD;
    static constexpr int HP = HD * HH * HW;
    static constexpr int M = TD * TH * TW;
    static constexpr int TAPS = KD * KS * KS;
    // 2-D tiles with 16-pixel rows: a fragment's 16 pixels are consecutive halo pixels, so a LINEAR image with a 160-byte
    // pixel stride is bank-conflict free for ds_read_b128 and every tap shift becomes an immediate offset (no address VALU
    // in the MFMA loop).  Other tiles keep the 128-byte XOR-swizzled image.
    static constexpr bool LIN = (!IS3D) && (TW == 16);
    static constexpr int HSTR = LIN ? 160 : 128;
};

// Halo tile of one K chunk: global (16 B per item) -> registers (load) ... -> swizzled LDS (store).  The two halves
// are separate so that the global loads of chunk c+1 fly while the last tap of chunk c computes.  Out-of-bounds pixels
// are zero (the conv's zero padding applies AFTER the optional per-(n,c) affine, exactly like GroupNorm -> Conv).
template <typename T, typename G, int NT> struct HaloStager {
    static constexpr int EPC = Tr<T>::EPC;
    static constexpr int ITEMS = G::HP * 8;
    static constexpr int HI = (ITEMS + NT - 1) / NT;
    u32x4 v[HI];
    uint32_t okmask;

    __device__ __forceinline__ void load(const ConvArgs& a, int n, int d0, int h0, int w0, int c0, int tid) {
        const bool first = c0 < a.Cin0;
        const SrcView s = first ? a.x0 : a.x1;
        const int cl = first ? c0 : c0 - a.Cin0;
        const int shd = (s.D != a.D), shh = (s.H != a.H), shw = (s.W != a.W);   // exact 2x nearest-upsample addressing
        const T* base = reinterpret_cast<const T*>(s.p) + (size_t)n * s.D * s.H * s.W * s.ld + cl;
        okmask = 0u;
#pragma unroll
        for (int b = 0; b < HI; ++b) {
            const int it = b * NT + tid;
            v[b] = u32x4{0u, 0u, 0u, 0u};
            if (it < ITEMS) {
                const int p = it >> 3, c16 = it & 7;
                const int pz = p / (G::HH * G::HW);
                const int pr = p - pz * (G::HH * G::HW);
                const int py = pr / G::HW;
                const int px = pr - py * G::HW;
                const int z = d0 + pz - G::PD, y = h0 + py - G::PAD, x = w0 + px - G::PAD;
                if (z >= 0 && z < a.D && y >= 0 && y < a.H && x >= 0 && x < a.W) {
                    const int off = (((z >> shd) * s.H + (y >> shh)) * s.W + (x >> shw)) * s.ld + c16 * EPC;
                    v[b] = *reinterpret_cast<const u32x4*>(base + off);
                    okmask |= (1u << b);
                }
            }
        }
    }

    __device__ __forceinline__ void store(char* halo, const ConvArgs& a, int n, int c0, int tid) {
#pragma unroll
        for (int b = 0; b < HI; ++b) {
            const int it = b * NT + tid;
            if (it < ITEMS) {
                const int p = it >> 3, c16 = it & 7;
                u32x4 val = v[b];
                if (a.in_scale != nullptr && ((okmask >> b) & 1u)) {
                    const float* sc = a.in_scale + (size_t)n * a.Cin + c0 + c16 * EPC;
                    const float* sh = a.in_shift + (size_t)n * a.Cin + c0 + c16 * EPC;
                    float f[EPC];
                    unpack_chunk<T>(val, f);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
                    val = pack_chunk<T>(f);
                }
                // non-LIN images: the chunk position is XORed with (halo COLUMN & 7).  ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...
                // (MI355X_MICROARCH.md, LDS); a fragment of the 8-wide 3-D tiles is two halo rows of 8 pixels, and for it XOR with the linear pixel index (the
                // round-1 choice) has 2-way conflicts (PMC: 24-39 % of the LDS cycles of the 3-D kernels) while XOR with the column is conflict-free for every tap
                // offset and row (enumerated over those groups)
                lds_write_b128(halo, G::LIN ? (p * G::HSTR + c16 * 16) : (p * 128 + ((c16 ^ ((p % G::HW) & 7)) << 4)), val);
            }
        }
    }
};

template <typename T, int NV>
__device__ __forceinline__ void store_run(T* dst, const float* v) {
    // NV consecutive channels, dst is 16-byte aligned
    constexpr int EPC = Tr<T>::EPC;
#pragma unroll
    for (int i = 0; i < NV; i += EPC) {
        u32x4 c = pack_chunk<T>(v + i);
        *reinterpret_cast<u32x4*>(dst + i) = c;
    }
}

// NT threads (256: 4 waves, two blocks per CU; 512: 8 waves sharing one weight tile, one block per CU),
// TPS taps per barrier interval (3 = one filter row: 96 MFMAs per wave between barriers instead of 32),
// PF pixel fragments per wave, PERSIST: the block walks tiles v, v+G, v+2G, ... and fetches the next tile's halo and first
// weight tile under the last MFMA cluster of the current tile (only the first tile of a block pays the load latency).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glob_void_t;

// WDMA: the weight tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPRs, no
// ds_write pass); the image stays XOR-swizzled by permuting the per-lane SOURCE address (the DMA destination is lane-linear).
template <typename T, typename G, int WN, int NF, int NT, int TPS, int PF, bool PERSIST, bool WDMA>
__global__ __launch_bounds__(NT, 2) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int WM = (NT / 64) / WN;
    constexpr int WAVE_M = PF * 16;      // pixels per wave
    static_assert(G::TAPS % TPS == 0, "taps per step must divide the tap count");
    constexpr int NSTEPS = G::TAPS / TPS;
    static_assert(G::M == WM * WAVE_M, "tile pixels must be PF*16 per M-wave");
    constexpr int WAVE_N = NF * 16;      // channels per wave
    constexpr int BN = WN * WAVE_N;      // channels per block
    constexpr int EPC = Tr<T>::EPC;
    constexpr int CK = Tr<T>::CK;
    constexpr int WI = TPS * BN * 8 / NT;  // 16-byte weight items per thread per step
    constexpr int WTILE = BN * 128;        // bytes of one tap's weight tile
    constexpr int NV = 4 * NF;           // consecutive output channels per lane
    static_assert(WI >= 1, "");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* wbuf = smem + G::HP * G::HSTR;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;

    const int total_tiles = a.nSp * a.nCt;
    const int tstride = PERSIST ? (int)gridDim.x : total_tiles;     // non-persistent: exactly one tile per block
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= total_tiles) return;   // block-uniform

    // cout-tile major, spatial minor: concurrently running blocks share the weight tile, neighbours share halos
    int n, d0, h0, w0, ncol0;
    auto decode = [&](int t, int& tn, int& td0, int& th0, int& tw0, int& tcol) {
        const int ct = t / a.nSp;
        const int sp = t - ct * a.nSp;
        const int tpi = a.tilesD * a.tilesH * a.tilesW;
        tn = sp / tpi;
        int r = sp - tn * tpi;
        const int td = r / (a.tilesH * a.tilesW);
        r -= td * (a.tilesH * a.tilesW);
        const int th = r / a.tilesW;
        const int tw = r - th * a.tilesW;
        td0 = td * G::TD;
        th0 = th * G::TH;
        tw0 = tw * G::TW;
        tcol = ct * BN;
    };
    decode(tile, n, d0, h0, w0, ncol0);

    int hb[PF], hx[PF];       // halo pixel index / halo column (without the tap offset) of this lane's pixel in fragment pf
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const int m = wm * WAVE_M + pf * 16 + li;
        const int dz = m / (G::TH * G::TW);
        const int hy = (m / G::TW) % G::TH;
        const int wx = m % G::TW;
        hb[pf] = (dz * G::HH + hy) * G::HW + wx;
        hx[pf] = wx;
    }

    f32x4 acc[NF][PF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) acc[f][pf] = f32x4{0.f, 0.f, 0.f, 0.f};

    int w_goff[WI], w_loff[WI], w_tapl[WI];
#pragma unroll
    for (int k = 0; k < WI; ++k) {
        const int itf = tid + k * NT;
        const int tapl = itf / (BN * 8);
        const int it = itf - tapl * (BN * 8);
        w_tapl[k] = tapl;
        const int row = it >> 3, c16 = it & 7;
        const int wv = row / WAVE_N, chw = row % WAVE_N;
        const int aa = chw / NV, ff = (chw >> 2) % NF, bb = chw & 3;
        const int lrow = wv * WAVE_N + ff * 16 + aa * 4 + bb;
        w_goff[k] = row * a.Cin + c16 * EPC;                 // + ncol0 * Cin per tile
        w_loff[k] = tapl * WTILE + lrow * 128 + ((c16 ^ (lrow & 7)) << 4);
        if constexpr (WDMA) {
            // DMA instruction q of this step fills LDS bytes [q*1024, q*1024+1024): lane l -> 16-byte slot q*64 + l.
            // Which (weight row, K chunk) must land there follows from inverting the read-side swizzle / row permutation.
            const int q = wave * WI + k;
            const int slot = q * 64 + lane;
            const int dtap = slot / (BN * 8);
            const int rem = slot - dtap * (BN * 8);
            const int lr = rem >> 3, pos = rem & 7;
            const int dc16 = pos ^ (lr & 7);
            const int dwv = lr / WAVE_N, j = lr % WAVE_N;
            const int dff = j / 16, daa = (j % 16) / 4, dbb = j & 3;
            const int drow = dwv * WAVE_N + daa * NV + dff * 4 + dbb;
            w_tapl[k] = dtap;
            w_goff[k] = drow * a.Cin + dc16 * EPC;
            w_loff[k] = q * 1024;                                // wave-uniform LDS byte offset of the instruction
        }
    }
    const T* wp = reinterpret_cast<const T*>(a.w);
    const size_t tap_stride = (size_t)a.Cout * a.Cin;
    const int nchunks = a.Cin / CK;

    u32x4 wreg[WDMA ? 1 : WI];
    auto w_fetch = [&](const T* wsrc, char* wdst) {   // next step's weight tile: registers, or straight into LDS buffer wdst
#pragma unroll
        for (int k = 0; k < WI; ++k) {
            const T* g = wsrc + (size_t)w_tapl[k] * tap_stride + w_goff[k];
            if constexpr (WDMA) {
                __builtin_amdgcn_global_load_lds((glob_void_t*)g, (lds_void_t*)(wdst + __builtin_amdgcn_readfirstlane(w_loff[k])), 16, 0, 0);
            } else {
                wreg[k] = *reinterpret_cast<const u32x4*>(g);
            }
        }
    };
    w_fetch(wp + (size_t)ncol0 * a.Cin, wbuf);
    HaloStager<T, G, NT> hs;
    hs.load(a, n, d0, h0, w0, 0, tid);
    hs.store(halo, a, n, 0, tid);
    int cur = 0;

#pragma unroll 1
    for (; tile < total_tiles; tile += tstride) {
        const bool has_next = PERSIST && (tile + tstride < total_tiles);
        int nn = n, nd0 = d0, nh0 = h0, nw0 = w0, ncolN = ncol0;
        if (has_next) decode(tile + tstride, nn, nd0, nh0, nw0, ncolN);

        // one step = TPS taps: W regs -> LDS, barrier, issue the next global loads (kept ABOVE the MFMA cluster), MFMAs
        auto step_fn = [&](int step, int c0, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            char* wb0 = wbuf + cur * (TPS * WTILE);
            if constexpr (!WDMA) {
#pragma unroll
                for (int k = 0; k < WI; ++k) lds_write_b128(wb0, w_loff[k], wreg[k]);
            }
            if constexpr (WDMA && (NF == 8 || MIS_WDMA_EXPLICIT_DRAIN)) {
                // the 256-column instantiations: drain this wave's LDS-DMA weight loads explicitly - with their register pressure hipcc no longer places
                // the vmcnt(0) ahead of the barrier on its own (observed as run-to-run differences in the low bits of the deep layers' outputs)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();   // WDMA: hipcc drains the in-flight LDS-DMA (vmcnt(0)) before this barrier
            {
                int nstep = step + 1, nc0 = c0, col = ncol0;
                int hn = n, hd0 = d0, hh0 = h0, hw0 = w0;        // whose halo to fetch (LAST only)
                if (LAST) {
                    nstep = 0;
                    nc0 = c0 + CK;
                    if (nc0 >= a.Cin) {       // tile finished: next tile's chunk 0, or (very last step) a harmless re-load
                        nc0 = 0;
                        col = ncolN;
                        hn = nn; hd0 = nd0; hh0 = nh0; hw0 = nw0;
                    }
                }
                w_fetch(wp + (size_t)(nstep * TPS) * tap_stride + (size_t)col * a.Cin + nc0, wbuf + (cur ^ 1) * (TPS * WTILE));
                if (LAST) hs.load(a, hn, hd0, hh0, hw0, nc0, tid);   // flies under this step's MFMAs
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tl = 0; tl < TPS; ++tl) {
                const int tap = step * TPS + tl;
                const char* wb = wb0 + tl * WTILE;
                int tapoff, tapkw = 0;
                if constexpr (G::KS == 1) {
                    tapoff = 0;
                } else {
                    const int kd = tap / (G::KS * G::KS);
                    const int kr = tap - kd * (G::KS * G::KS);
                    const int kh = kr / G::KS, kw = kr - kh * G::KS;
                    tapoff = (kd * G::HH + kh) * G::HW + kw;
                    tapkw = kw;
                }
                // (explicitly double-buffering these fragment reads behind sched_barriers, or weaving them with
                //  sched_group_barrier, measured 2-4 % SLOWER than letting hipcc interleave ds_reads and MFMAs itself)
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) {
                    u32x4 A[NF], B[PF];
                    const int ch = kg * 4 + lg;
#pragma unroll
                    for (int f = 0; f < NF; ++f)
                        A[f] = lds_read_b128(wb, (wn * WAVE_N + f * 16 + li) * 128 + ((ch ^ (li & 7)) << 4));
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf) {
                        if constexpr (G::LIN) {
                            B[pf] = lds_read_b128(halo, (hb[pf] + tapoff) * G::HSTR + ch * 16);
                        } else {
                            const int p = hb[pf] + tapoff;
                            B[pf] = lds_read_b128(halo, p * 128 + ((ch ^ ((hx[pf] + tapkw) & 7)) << 4));
                        }
                    }
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int pf = 0; pf < PF; ++pf) mma_b128<T>(acc[f][pf], A[f], B[pf]);
                }
            }
            cur ^= 1;
        };

#pragma unroll 1
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int c0 = chunk * CK;
#pragma unroll 1
            for (int st = 0; st < NSTEPS - 1; ++st) step_fn(st, c0, std::false_type{});
            step_fn(NSTEPS - 1, c0, std::true_type{});
            if (chunk + 1 < nchunks) {
                __syncthreads();   // every wave is done reading this chunk's halo
                hs.store(halo, a, n, c0 + CK, tid);
            }
        }

        // ---- epilogue: lane (li, lg) holds, per pixel fragment, NV consecutive columns ---------------------
        {
            const int colw = ncol0 + wn * WAVE_N;          // wave-uniform first column
            const int col = colw + lg * NV;                // this lane's first column
            const bool to0 = colw < a.Cout0;
            T* ybase = reinterpret_cast<T*>(to0 ? a.y0 : a.y1);
            const int yld = to0 ? a.y0_ld : a.y1_ld;
            const int ymode = to0 ? a.y0_mode : a.y1_mode;
            const int cview = to0 ? a.Cout0 : a.Cout - a.Cout0;    // columns routed to this output
            const int lcol = to0 ? col : col - a.Cout0;
            int bcol = col;                                         // bias index
            int ab = 0, cq = 0;
            if (ymode == MIS_OUT_SHUFFLE2) {
                cq = cview >> 2;
                ab = lcol / cq;
                bcol = (to0 ? 0 : a.Cout0) + (lcol - ab * cq);      // bias is per real output channel c
            }
            float bv[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) bv[i] = 0.f;
            if (a.bias != nullptr) {
#pragma unroll
                for (int i = 0; i < NV; ++i) bv[i] = a.bias[bcol + i];
            }
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) {
                const int m = wm * WAVE_M + pf * 16 + li;
                const int dz = m / (G::TH * G::TW);
                const int hy = (m / G::TW) % G::TH;
                const int wx = m % G::TW;
                const int z = d0 + dz, y = h0 + hy, x = w0 + wx;
                float o[NV];
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[f * 4 + q] = acc[f][pf][q] + bv[f * 4 + q];
                        acc[f][pf][q] = 0.f;
                    }
                if (z < a.D && y < a.H && x < a.W) {
                    if (a.relu) {
#pragma unroll
                        for (int i = 0; i < NV; ++i) o[i] = fmaxf(o[i], 0.f);
                    }
                    const size_t pix = (((size_t)n * a.D + z) * a.H + y) * a.W + x;
                    if (a.mask != nullptr) {
                        const T* mp = reinterpret_cast<const T*>(a.mask) + pix * a.mask_ld + col;
#pragma unroll
                        for (int i = 0; i < NV; i += EPC) {
                            float mf[EPC];
                            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(mp + i), mf);
#pragma unroll
                            for (int e = 0; e < EPC; ++e) o[i + e] = (mf[e] > 0.f) ? o[i + e] : 0.f;
                        }
                    }
                    if constexpr (sizeof(T) == 2) {
                        if (a.mask_bits != nullptr) {          // ReLU bits instead of the bf16 mask tensor (relu_bits.hpp): one byte per 8 columns of this pixel
#pragma unroll
                            for (int i = 0; i < NV; i += 8) {
                                const unsigned b = a.mask_bits[rb_byte_offset((a.H + 7) >> 3, a.W, a.Cout >> 6, n, y, x, (col + i) >> 3)];
#pragma unroll
                                for (int e = 0; e < 8; ++e) o[i + e] = ((b >> e) & 1u) ? o[i + e] : 0.f;
                            }
                        }
                    }
                    T* dst;
                    if (ymode == MIS_OUT_PLAIN) {
                        dst = ybase + pix * yld + lcol;
                    } else if (ymode == MIS_OUT_SHUFFLE2) {
                        const int oy = 2 * y + (ab >> 1), ox = 2 * x + (ab & 1);
                        const size_t opix = ((size_t)n * (2 * a.H) + oy) * (size_t)(2 * a.W) + ox;
                        dst = ybase + opix * yld + (lcol - ab * cq);
                    } else {   // MIS_OUT_UNSHUFFLE2
                        const int oh = a.H >> 1, ow = a.W >> 1;
                        const size_t opix = ((size_t)n * oh + (y >> 1)) * ow + (x >> 1);
                        dst = ybase + opix * yld + ((y & 1) * 2 + (x & 1)) * cview + lcol;
                    }
                    store_run<T, NV>(dst, o);
                }
            }
        }
        if (has_next) {
            __syncthreads();   // every wave is done reading this tile's last halo
            hs.store(halo, a, nn, 0, tid);
            n = nn; d0 = nd0; h0 = nh0; w0 = nw0; ncol0 = ncolN;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Weight-stationary 3x3 kernel for the 64 -> 64 channel bf16 layers at full resolution (down_conv.0.second, up_conv.3.second and
// their dgrads: the layers with the lowest FLOP per byte of the net).  All 9 taps of the [64 x 64] filter (72 KiB) are loaded
// into LDS ONCE per block; a persistent block then only streams 32x16-pixel halo tiles: two barriers per tile and no weight
// traffic.  The next tile's halo is fetched into registers at the top of the current tile's MFMA loop, so its HBM latency hides
// under a whole tile of compute.  With only 288 MFMAs per wave and tile, the VALU work around them decides the speed, so every
// tile-invariant quantity lives in registers: per-item global byte offsets (added to a scalar tile base: saddr loads, no
// per-item address arithmetic), LDS offsets, the output offsets; interior tiles skip all bounds checks; the bias is the
// accumulator's initial value; ReLU and the ReLU mask are applied on packed bf16 pairs.
// Halo image: 128-byte pixels, 16-byte chunk position XORed with (column & 7) - a fragment's 16 pixels are 16 consecutive
// columns of one halo row, which makes every ds_read_b128 conflict-free while a tap's row shift stays an immediate offset.
// ---------------------------------------------------------------------------------------------------------
struct WS64 {
    static constexpr int TH = 32, TW = 16, HH = 34, HW = 18, HP = HH * HW;
    static constexpr int NT = 512;
    static constexpr int WBYTES = 9 * 64 * 128;           // 73,728: weights first (tap offsets mostly fit ds immediates)
    static constexpr int HBYTES = HP * 128;               // 78,336
    static constexpr int ITEMS = HP * 8;
    static constexpr int HI = (ITEMS + NT - 1) / NT;      // 10
    static constexpr int LAST = ITEMS - (HI - 1) * NT;    // threads with a valid last item
};

typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {        // max as signed 16-bit: negative floats have the sign bit set
    const s16x2_t z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, v), z));
}
__device__ __forceinline__ uint32_t mask_bf16x2(uint32_t v, uint32_t m) {   // v where the bf16 mask value is > 0, else 0
    typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
    const s16x2_t z = {0, 0};
    const u16x2_t one = {1, 1};
    const u16x2_t pos = __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, m), z)), one);
    return __builtin_bit_cast(uint32_t, (u16x2_t)(__builtin_bit_cast(u16x2_t, v) * pos));
}

// v where bit 2q / 2q + 1 of byte R (0..3) of the ReLU-bits word mw is set (low / high half), else 0
__device__ __forceinline__ uint32_t maskbits_bf16x2(uint32_t v, uint32_t mw, int R, int Q) {      // (R, Q: constants after unrolling)
    const uint32_t sh = (uint32_t)((R & 1) * 8 + 2 * Q) * 0x00010001u + 0x00010000u;      // shift the byte's 16-bit half by (s, s + 1) into the two result halves
    uint32_t t;
    if (R < 2) asm("v_pk_lshrrev_b16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "s"(sh), "v"(mw));
    else asm("v_pk_lshrrev_b16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(t) : "s"(sh), "v"(mw));
    t &= 0x00010001u;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(t) : "v"(v), "v"(t));
    return t;
}

__global__ __launch_bounds__(512, 2) void conv64_ws_kernel(const ConvArgs a) {
    using T = __bf16;
    constexpr int EPC = 8, NF = 4, PF = 4, NV = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wlds = smem;
    char* halo = smem + WS64::WBYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wm = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;

    const int total_tiles = a.nSp;
    const int tstride = (int)gridDim.x;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= total_tiles) return;

    // ---- the whole filter, once: [tap][row'][128 B], row' permuted so that a lane ends up with 16 consecutive channels ----
    {
        const T* wp = reinterpret_cast<const T*>(a.w);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int it = tid + k * WS64::NT;                 // 4608 items = 9 taps x 64 rows x 8 chunks
            const int tap = it >> 9, r = (it >> 3) & 63, c16 = it & 7;
            // output channel r = (f >> 1) * 32 + lg * 8 + (f & 1) * 4 + q sits in MFMA row lg * 4 + q of fragment f (the channel order of the column-segment kernels, conv_pp.hip:
            // a lane holds channels lg*8..+7 and 32 + lg*8..+7, each store instruction covers 64 contiguous bytes per pixel, and the lane's ReLU bits are its own bytes)
            const int ff = (r >> 5) * 2 + ((r >> 2) & 1), aa = (r >> 3) & 3, bb = r & 3;
            const int lrow = ff * 16 + aa * 4 + bb;
            const u32x4 v = *reinterpret_cast<const u32x4*>(wp + ((size_t)tap * 64 + r) * 64 + c16 * EPC);
            lds_write_b128(wlds, tap * 8192 + lrow * 128 + ((c16 ^ (lrow & 7)) << 4), v);
        }
    }

    // ---- tile-invariant per-thread staging tables ----
    int rel[WS64::HI], lo[WS64::HI];
#pragma unroll
    for (int b = 0; b < WS64::HI; ++b) {
        int it = b * WS64::NT + tid;
        if (it >= WS64::ITEMS) it = WS64::ITEMS - 1;
        const int p = it >> 3, c16 = it & 7;
        const int py = p / WS64::HW, px = p - py * WS64::HW;
        rel[b] = ((py * a.W + px) * a.x0.ld + c16 * EPC) * 2;           // bytes from the tile's halo origin
        lo[b] = p * 128 + ((c16 ^ (px & 7)) << 4);
    }
    const bool last_ok = tid < WS64::LAST;

    const int tpi = a.tilesH * a.tilesW;
    auto decode = [&](int t, int& tn, int& th0, int& tw0) {
        tn = t / tpi;
        const int r = t - tn * tpi;
        const int th = r / a.tilesW;
        th0 = th * WS64::TH;
        tw0 = (r - th * a.tilesW) * WS64::TW;
    };
    u32x4 hv[WS64::HI];
    auto h_load = [&](int n, int h0, int w0) {
        // halo origin = pixel (h0 - 1, w0 - 1); may lie before the buffer for border tiles (those items are never dereferenced)
        const char* gbase = reinterpret_cast<const char*>(a.x0.p) + (((long long)n * a.H + (h0 - 1)) * a.W + (w0 - 1)) * (long long)a.x0.ld * 2;
        const bool interior = h0 >= 1 && h0 + WS64::TH + 1 <= a.H && w0 >= 1 && w0 + WS64::TW + 1 <= a.W;     // block-uniform
        if (interior) {
#pragma unroll
            for (int b = 0; b < WS64::HI - 1; ++b) hv[b] = *reinterpret_cast<const u32x4*>(gbase + (uint32_t)rel[b]);
            hv[WS64::HI - 1] = u32x4{0u, 0u, 0u, 0u};
            if (last_ok) hv[WS64::HI - 1] = *reinterpret_cast<const u32x4*>(gbase + (uint32_t)rel[WS64::HI - 1]);
        } else {
#pragma unroll
            for (int b = 0; b < WS64::HI; ++b) {
                const int p = (b * WS64::NT + tid) >> 3;              // border tiles only: recompute the item's halo coordinates
                const int py = p / WS64::HW, px = p - py * WS64::HW;
                const bool ok = (uint32_t)(h0 - 1 + py) < (uint32_t)a.H && (uint32_t)(w0 - 1 + px) < (uint32_t)a.W && (b < WS64::HI - 1 || last_ok);
                hv[b] = u32x4{0u, 0u, 0u, 0u};
                if (ok) hv[b] = *reinterpret_cast<const u32x4*>(gbase + (uint32_t)rel[b]);
            }
        }
    };
    auto h_store = [&]() {
#pragma unroll
        for (int b = 0; b < WS64::HI - 1; ++b) lds_write_b128(halo, lo[b], hv[b]);
        if (last_ok) lds_write_b128(halo, lo[WS64::HI - 1], hv[WS64::HI - 1]);
    };

    int n, h0, w0;
    decode(tile, n, h0, w0);
    h_load(n, h0, w0);
    h_store();

    // fragment addresses: B (pixels) = column part per (kw, kg) in a register + (pf + kh) * row stride as an immediate
    int bcol[3][2], acol[2];
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
        const int ch = kg * 4 + lg;
        acol[kg] = li * 128 + ((ch ^ (li & 7)) << 4);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int px = li + kw;
            bcol[kw][kg] = (wm * 4 * WS64::HW + px) * 128 + ((ch ^ (px & 7)) << 4);
        }
    }
    constexpr int ROWB = WS64::HW * 128;   // 2304

    // lane (li, lg) owns, per pixel row pf, channels lg*8 .. +7 (fragments 0, 1) and 32 + lg*8 .. +7 (fragments 2, 3) of pixel (h0 + wm*4 + pf, w0 + li)
    const int col = lg * 8;
    const uint32_t orel = (uint32_t)(((wm * 4) * a.W + li) * a.y0_ld + col) * 2u;       // bytes from the tile's first output pixel
    const uint32_t mrel = (uint32_t)(((wm * 4) * a.W + li) * a.mask_ld + col) * 2u;
    f32x4 bv[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        bv[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr) bv[f] = *reinterpret_cast<const f32x4*>(a.bias + (f >> 1) * 32 + col + (f & 1) * 4);
    }
    f32x4 acc[NF][PF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) acc[f][pf] = bv[f];           // the bias is the accumulator's initial value

#ifdef MIS_WS64_STAMPS
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define STAMP(i) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); st[i] += tn_ - tprev; tprev = tn_; }
#else
#define STAMP(i)
#endif
#pragma unroll 1
    for (; tile < total_tiles; tile += tstride) {
        const bool has_next = tile + tstride < total_tiles;
        int nn = n, nh0 = h0, nw0 = w0;
        if (has_next) decode(tile + tstride, nn, nh0, nw0);
        __syncthreads();                       // this tile's halo (and, first time, the filter) is in LDS
        STAMP(0)
        if (has_next) h_load(nn, nh0, nw0);    // in flight during the 288 MFMAs below
        // ReLU bits of this lane's 4 rows (relu_bits.hpp): bytes [lg][i][(wm & 1) * 4 .. + 3] of the record of (n, row block, column, the one 64-channel block) - two dwords,
        // in flight during the MFMAs as well
        uint32_t mbw[2] = {0u, 0u};
        if (a.mask_bits != nullptr && w0 + li < a.W && h0 + wm * 4 < a.H) {
            const uint32_t* bp = reinterpret_cast<const uint32_t*>(a.mask_bits + ((((size_t)n * ((a.H + 7) >> 3) + ((h0 + wm * 4) >> 3)) * a.W + (w0 + li)) * 64 + lg * 16 + (wm & 1) * 4));
            mbw[0] = bp[0];
            mbw[1] = bp[2];
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(1)
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
            const char* hrow = halo + kh * ROWB;
            const char* wrow = wlds + kh * (3 * 8192);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) {
                    u32x4 A[NF], B[PF];
#pragma unroll
                    for (int f = 0; f < NF; ++f) A[f] = lds_read_b128(wrow, acol[kg] + kw * 8192 + f * 2048);
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf) B[pf] = lds_read_b128(hrow, bcol[kw][kg] + pf * ROWB);
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int pf = 0; pf < PF; ++pf) mma_b128<T>(acc[f][pf], A[f], B[pf]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(2)
        __syncthreads();                       // every wave is done reading this halo
        STAMP(3)
        if (has_next) h_store();
        __builtin_amdgcn_sched_barrier(0);
        STAMP(4)
        // ---- epilogue ----
        {
            const long long opix = ((long long)n * a.H + h0) * a.W + w0;
            char* obase = reinterpret_cast<char*>(a.y0) + opix * a.y0_ld * 2;
            const char* mbase = reinterpret_cast<const char*>(a.mask) + opix * a.mask_ld * 2;
            const bool full = h0 + WS64::TH <= a.H && w0 + WS64::TW <= a.W;      // block-uniform
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) {
                u32x4 o[2];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    o[f >> 1][(f & 1) * 2] = pack_bf16x2(acc[f][pf][0], acc[f][pf][1]);
                    o[f >> 1][(f & 1) * 2 + 1] = pack_bf16x2(acc[f][pf][2], acc[f][pf][3]);
                    acc[f][pf] = bv[f];
                }
                if (full || (h0 + wm * 4 + pf < a.H && w0 + li < a.W)) {
                    if (a.relu) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[i >> 2][i & 3] = relu_bf16x2(o[i >> 2][i & 3]);
                    }
                    if (a.mask != nullptr) {
                        const u32x4* mp = reinterpret_cast<const u32x4*>(mbase + (mrel + (uint32_t)(pf * a.W * a.mask_ld * 2)));
                        const u32x4 m0 = mp[0], m1 = mp[4];           // (pieces are 64 bytes apart)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            o[0][i] = mask_bf16x2(o[0][i], m0[i]);
                            o[1][i] = mask_bf16x2(o[1][i], m1[i]);
                        }
                    }
                    if (a.mask_bits != nullptr) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int q = 0; q < 4; ++q) o[i][q] = maskbits_bf16x2(o[i][q], mbw[i], pf, q);
                    }
                    u32x4* dst = reinterpret_cast<u32x4*>(obase + (orel + (uint32_t)(pf * a.W * a.y0_ld * 2)));
                    dst[0] = o[0];
                    dst[4] = o[1];
                }
            }
        }
        n = nn; h0 = nh0; w0 = nw0;
        __builtin_amdgcn_sched_barrier(0);
        STAMP(5)
    }
#ifdef MIS_WS64_STAMPS
    if (a.y1 != nullptr && lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.y1) + ((size_t)blockIdx.x * 8 + wm) * 8;
        for (int i = 0; i < 6; ++i) o[i] = st[i];
        o[6] = __builtin_amdgcn_s_memrealtime();
        o[7] = __builtin_amdgcn_s_memtime();
    }
#endif
}

static int launch_ws64(const MisConvDesc* d, hipStream_t stream) {
    ConvArgs a;
    a.tq = nullptr;
    a.N = d->N; a.D = 1; a.H = d->H; a.W = d->W; a.Cin = 64; a.Cout = 64; a.Cin0 = 64; a.Cout0 = 64;
    a.x0 = SrcView{d->x0, d->x0_ld, 1, d->x0_H, d->x0_W};
    a.x1 = SrcView{nullptr, 0, 0, 0, 0};
    a.in_scale = nullptr; a.in_shift = nullptr;
    a.w = d->w; a.bias = d->bias; a.relu = d->relu; a.mask = d->mask; a.mask_ld = d->mask_ld; a.relu_bits = nullptr; a.mask_bits = reinterpret_cast<const unsigned char*>(d->mask_bits);
    a.y0 = d->y0; a.y0_ld = d->y0_ld; a.y0_mode = MIS_OUT_PLAIN;
    a.y1 = nullptr; a.y1_ld = 0; a.y1_mode = 0;
#ifdef MIS_WS64_STAMPS
    a.y1 = d->y1;
#endif
    a.tilesD = 1;
    a.tilesH = (d->H + WS64::TH - 1) / WS64::TH;
    a.tilesW = (d->W + WS64::TW - 1) / WS64::TW;
    const long long nsp = (long long)d->N * a.tilesH * a.tilesW;
    MIS_REQUIRE(nsp < (1ll << 31), MIS_EUNSUPPORTED, "conv_igemm: grid too large");
    // per-thread byte offsets inside one halo / output tile are kept in 32 bits
    MIS_REQUIRE((long long)(WS64::HH + 1) * d->W * (d->x0_ld > d->y0_ld ? d->x0_ld : d->y0_ld) * 2 < (1ll << 31), MIS_EUNSUPPORTED,
                "conv_igemm: row too long for the 64-channel weight-stationary kernel");
    a.nSp = (int)nsp;
    a.nCt = 1;
    const size_t lds = (size_t)WS64::WBYTES + WS64::HBYTES;
    static std::atomic<unsigned long long> attr_done{0};
    if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&conv64_ws_kernel), lds, "conv_igemm(ws64)")) return rc;
    hipLaunchKernelGGL(conv64_ws_kernel, dim3((unsigned)(nsp > mis_persist_cus() ? mis_persist_cus() : nsp)), dim3(WS64::NT), lds, stream, a);
    MIS_LAUNCH_CHECK("conv_igemm(ws64)");
    return MIS_OK;
}

// ---------------------------------------------------------------------------------------------------------
template <typename T, typename G, int WN, int NF, int NT = 256, int TPS = 1, int PF = 4, bool PERSIST = false, bool WDMA = false>
static int launch_cfg(const MisConvDesc* d, hipStream_t stream) {
    constexpr int BN = WN * NF * 16;
    ConvArgs a;
    a.tq = nullptr;
    a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
    a.Cin0 = d->Cin0; a.Cout0 = d->Cout0;
    a.x0 = SrcView{d->x0, d->x0_ld, d->x0_D, d->x0_H, d->x0_W};
    a.x1 = SrcView{d->x1, d->x1_ld, d->x1_D, d->x1_H, d->x1_W};
    a.in_scale = d->in_scale; a.in_shift = d->in_shift;
    a.w = d->w; a.bias = d->bias; a.relu = d->relu; a.mask = d->mask; a.mask_ld = d->mask_ld; a.relu_bits = nullptr; a.mask_bits = reinterpret_cast<const unsigned char*>(d->mask_bits);
    a.y0 = d->y0; a.y0_ld = d->y0_ld; a.y0_mode = d->y0_mode;
    a.y1 = d->y1; a.y1_ld = d->y1_ld; a.y1_mode = d->y1_mode;
    a.tilesD = (d->D + G::TD - 1) / G::TD;
    a.tilesH = (d->H + G::TH - 1) / G::TH;
    a.tilesW = (d->W + G::TW - 1) / G::TW;
    const long long nsp = (long long)d->N * a.tilesD * a.tilesH * a.tilesW;
    a.nCt = d->Cout / BN;
    MIS_REQUIRE(nsp * a.nCt < (1ll << 31), MIS_EUNSUPPORTED, "conv_igemm: grid too large");
    a.nSp = (int)nsp;
    const size_t lds = (size_t)G::HP * G::HSTR + 2 * (size_t)TPS * BN * 128;
    static std::atomic<unsigned long long> attr_done{0};
    if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&conv_igemm_kernel<T, G, WN, NF, NT, TPS, PF, PERSIST, WDMA>), lds, "conv_igemm"))
        return rc;
    hipLaunchKernelGGL((conv_igemm_kernel<T, G, WN, NF, NT, TPS, PF, PERSIST, WDMA>), dim3((unsigned)((PERSIST && nsp * a.nCt > 256) ? 256 : nsp * a.nCt)), dim3(NT), lds, stream, a);
    MIS_LAUNCH_CHECK("conv_igemm");
    return MIS_OK;
}

// name of the kernel configuration the last mis_conv_igemm call of this thread ran (tests assert that a parity case reaches the branch it is meant for)
static thread_local const char* g_conv_last = "";
thread_local bool g_conv_bits_fused = false;       // set by a launcher whose kernel writes MisConvDesc::relu_bits from its epilogue (conv_pp.hip)
extern "C" const char* mis_conv_last_dispatch(void) { return g_conv_last; }
#define RUN(tag, ...)          \
    do {                       \
        g_conv_last = (tag);   \
        return __VA_ARGS__;    \
    } while (0)

template <typename T> static int dispatch(const MisConvDesc* d, hipStream_t s) {
    const bool is3d = d->is3d != 0;
    const bool wide = (d->Cout % 128) == 0;
    if (d->ksize == 3) {
        if (!is3d) {
            const int v2 = !mis_sw(SW_CONV_V1);
            const int v3 = mis_sw(SW_CONV_V3);
            if constexpr (sizeof(T) == 2) {
                // bf16, Cout % 128 == 0, single source: the ping-pong kernel (conv_pp.hip)
                const bool pp = !mis_sw(SW_CONV_NOPP);      // (the parity tests reach the pre-ping-pong configurations through mis_dispatch_override)
                // 64-column layers stay on the weight-stationary / bn64 configurations: the ping-pong kernel with 64-column blocks (wave tile 64 px x 32 ch: 12 fragment
                // reads per 16 MFMAs) is bound by its R segments - measured 629 vs 959 TFLOP/s (64->64 at 512^2) and 784 vs 916 (128->64); MIS_CONV_PP64=1 selects it
                const bool pp64 = mis_sw(SW_CONV_PP64) || mis_sw(SW_CONV_PPC64);
                if (pp && conv_pp_eligible(d) && (d->Cout % 128 == 0 || pp64 || conv_ppc64_auto(d) || conv_pp_rs64_eligible(d)) && (d->mask_bits == nullptr || conv_ppc_choice(d) != 0)) {      // (rs64: opt-in, MIS_CONV_RS64=1)
                    const char* tag = "";
                    const int rc = launch_conv_pp(d, s, &tag);
                    g_conv_last = tag;
                    return rc;
                }
            }
            if constexpr (sizeof(T) == 2) {
                // deep layers: 256 output columns per block (wave tile 128 ch x 64 px, one tap per barrier): every staged halo pixel and every pixel
                // fragment read from LDS feeds twice the MFMAs (+7...12 % per layer for Cin >= 256 despite 19 spilled VGPRs)
                const int k3w = !mis_sw(SW_CONV_K3_NO256);
                const int k3min = mis_sw(SW_CONV_K3_256_MINCIN);
                if (k3w && v2 && d->Cout % 256 == 0 && d->Cin >= k3min) RUN("k3.2d.bn256.dma", launch_cfg<T, Geom<1, 16, 16, 3, false>, 2, 8, 512, 1, 4, false, true>(d, s));
            }
            if (wide && v3 && sizeof(T) == 2) RUN("k3.2d.bn128.v3", launch_cfg<T, Geom<1, 32, 16, 3, false>, 2, 4, 512, 1, 8>(d, s));   // 8 waves, wave tile 128 px x 64 ch
            // persistent tiles pay off when a tile has few K steps (prologue latency dominates); deep layers run ~5 % faster without
            const bool shallow = d->Cin <= 2 * (int)Tr<T>::CK;
            if (wide && v2 && shallow) RUN("k3.2d.bn128.persist.dma", launch_cfg<T, Geom<1, 16, 16, 3, false>, 2, 4, 512, 3, 4, true, true>(d, s));
            const int dma = !mis_sw(SW_CONV_NODMA);
            if (wide && v2 && dma) RUN("k3.2d.bn128.dma", launch_cfg<T, Geom<1, 16, 16, 3, false>, 2, 4, 512, 3, 4, false, true>(d, s));
            if (wide && v2) RUN("k3.2d.bn128.reg", launch_cfg<T, Geom<1, 16, 16, 3, false>, 2, 4, 512, 3, 4, false>(d, s));   // 8 waves, one filter row per barrier
            if (wide) RUN("k3.2d.bn128.v1", launch_cfg<T, Geom<1, 8, 16, 3, false>, 2, 4>(d, s));
            if constexpr (sizeof(T) == 2) {
                const int ws = !mis_sw(SW_CONV_NOWS64);
                const bool plain = d->x1 == nullptr && d->in_scale == nullptr && d->y0_mode == MIS_OUT_PLAIN && d->x0_H == d->H && d->x0_W == d->W;
                if (ws && plain && d->Cin == 64 && d->Cout == 64 && (long long)d->N * d->H * d->W >= 256ll * 512 && (d->mask == nullptr || d->mask_ld % 8 == 0)) RUN("k3.2d.ws64", launch_ws64(d, s));
            }
            if (v2 && (long long)d->H * d->W >= 64 * 64) RUN("k3.2d.bn64.persist.dma", launch_cfg<T, Geom<1, 32, 16, 3, false>, 1, 4, 512, 3, 4, true, true>(d, s));
            RUN("k3.2d.bn64.v1", launch_cfg<T, Geom<1, 16, 16, 3, false>, 1, 4>(d, s));
        }
        if constexpr (sizeof(T) == 2) {
            // bf16, single source, no operand affine (the engines hand over the normalised tensor): the column-segment ping-pong kernel (conv3d_pp.hip)
            if (!mis_sw(SW_CONV3D_NOPP) && conv3d_pp_eligible(d)) {
                const char* tag = "";
                const int rc = launch_conv3d_pp(d, s, &tag);
                g_conv_last = tag;
                return rc;
            }
        }
        if constexpr (sizeof(T) == 4) {
            // fp32, single plain source (the fp32 engine materialises the GroupNorm output since round 5): the all-DMA kernel with prefetched fragments (conv3d_f32.hip)
            if (!mis_sw(SW_CONV3D_F32_NOPP) && conv3d_f32_eligible(d)) {
                const char* tag = "";
                const int rc = launch_conv3d_f32(d, s, &tag);
                g_conv_last = tag;
                return rc;
            }
        }
        if (wide) RUN("k3.3d.bn128", launch_cfg<T, Geom<4, 4, 8, 3, true>, 2, 4>(d, s));
        // bf16, Cout not a multiple of 128 (the 64- and 192-column layers at full resolution): 8 waves on a 4x8x8 voxel tile, LDS-DMA weight
        // tiles, 3 taps per barrier (+6...11 % over the 4-wave 4x4x8 config; the same tile with 1 tap per barrier or 4 waves was slower)
        const int bn64v2 = !mis_sw(SW_CONV3D_BN64V1);
        if (bn64v2 && sizeof(T) == 2) RUN("k3.3d.bn64.dma", launch_cfg<T, Geom<4, 8, 8, 3, true>, 2, 2, 512, 3, 4, false, true>(d, s));
        RUN("k3.3d.bn64.v1", launch_cfg<T, Geom<4, 4, 8, 3, true>, 2, 2>(d, s));
    }
    if (!is3d) {
        const int k1v2 = !mis_sw(SW_CONV_K1V1);
        const int k1p = !mis_sw(SW_CONV_K1NOPERSIST);
        if constexpr (sizeof(T) == 2) {
            // the transposed convolution's GEMMs (plain / pixel-shuffled destination, Cin % 64 == 0, Cout % 128 == 0): the ping-pong 1x1 kernel (gemm1_pp.hip)
            if (!mis_sw(SW_GEMM1_NOPP) && gemm1_pp_eligible(d)) {
                const char* tag = "";
                const int rc = launch_gemm1_pp(d, s, &tag);
                g_conv_last = tag;
                return rc;
            }
        }
        if constexpr (sizeof(T) == 2) {
            // deep 1x1 GEMMs (transposed-conv forward / dgrad): 256 output columns per block = twice the MFMA work per staged pixel tile and barrier
            const int k1nf8 = !mis_sw(SW_CONV_K1_NO256);
            if (k1nf8 && k1v2 && d->Cout % 256 == 0 && d->Cin >= 4 * (int)Tr<T>::CK)
                RUN("k1.2d.bn256.dma", launch_cfg<T, Geom<1, 16, 16, 1, false>, 2, 8, 512, 1, 4, false, true>(d, s));
        }
        if (wide && k1v2 && k1p && d->Cin <= 4 * (int)Tr<T>::CK) RUN("k1.2d.bn128.persist.dma", launch_cfg<T, Geom<1, 16, 16, 1, false>, 2, 4, 512, 1, 4, true, true>(d, s));
        if (wide && k1v2) RUN("k1.2d.bn128.dma", launch_cfg<T, Geom<1, 16, 16, 1, false>, 2, 4, 512, 1, 4, false, true>(d, s));
        if (wide) RUN("k1.2d.bn128.v1", launch_cfg<T, Geom<1, 8, 16, 1, false>, 2, 4>(d, s));
        RUN("k1.2d.bn64", launch_cfg<T, Geom<1, 16, 16, 1, false>, 1, 4>(d, s));
    }
    if (wide) RUN("k1.3d.bn128", launch_cfg<T, Geom<4, 4, 8, 1, true>, 2, 4>(d, s));
    RUN("k1.3d.bn64", launch_cfg<T, Geom<4, 4, 8, 1, true>, 2, 2>(d, s));
}

extern "C" long long mis_conv_stats_rows(const MisConvDesc* d) {
    if (d == nullptr || d->dtype != MIS_F32 || !d->is3d || d->ksize != 3 || mis_sw(SW_CONV3D_F32_NOPP) || !conv3d_f32_eligible(d)) return 0;
    return conv3d_f32_stats_rows(d);
}

extern "C" int mis_conv_igemm(const MisConvDesc* d, void* stream) {
    (void)hipGetLastError();   // drop any stale (already handled) error of this thread
    MIS_REQUIRE(d != nullptr, MIS_EINVAL, "conv_igemm: null descriptor");
    MIS_REQUIRE(d->dtype == MIS_F32 || d->dtype == MIS_BF16, MIS_EINVAL, "conv_igemm: bad dtype %d", d->dtype);
    const int CK = d->dtype == MIS_BF16 ? 64 : 32;
    const int EPC = d->dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(d->ksize == 3 || d->ksize == 1, MIS_EUNSUPPORTED, "conv_igemm: ksize %d", d->ksize);
    MIS_REQUIRE(d->N > 0 && d->D > 0 && d->H > 0 && d->W > 0, MIS_EINVAL, "conv_igemm: empty grid");
    MIS_REQUIRE(d->is3d || d->D == 1, MIS_EINVAL, "conv_igemm: D must be 1 for a 2-D op");
    // (the 3-D ping-pong kernel walks 32-channel K chunks in bf16: encoders.0 SingleConv2 of UNet3D reads its 32 real input channels out of a 64-channel buffer)
    const bool pp3 = d->dtype == MIS_BF16 && d->is3d && d->ksize == 3 && !mis_sw(SW_CONV3D_NOPP) && conv3d_pp_eligible(d);
    MIS_REQUIRE(d->Cin > 0 && (d->Cin % CK == 0 || (pp3 && d->Cin % 32 == 0)), MIS_EUNSUPPORTED, "conv_igemm: Cin %d must be a multiple of %d", d->Cin, CK);
    // (round 6: the fp32 3x3x3 all-DMA kernel has a 32-column tile - the dgrad of encoders.0 SingleConv2 writes the layer's 32 real input channels)
    const bool f3 = d->dtype == MIS_F32 && d->is3d && d->ksize == 3 && !mis_sw(SW_CONV3D_F32_NOPP) && conv3d_f32_eligible(d);
    MIS_REQUIRE(d->Cout > 0 && (d->Cout % 64 == 0 || (f3 && d->Cout % 32 == 0)), MIS_EUNSUPPORTED, "conv_igemm: Cout %d must be a multiple of 64", d->Cout);
    MIS_REQUIRE(d->x0 != nullptr && d->w != nullptr && d->y0 != nullptr, MIS_EINVAL, "conv_igemm: null pointer");
    MIS_REQUIRE(d->Cin0 > 0 && d->Cin0 <= d->Cin && (d->Cin0 % CK == 0 || (pp3 && d->Cin0 == d->Cin)), MIS_EINVAL, "conv_igemm: Cin0 %d", d->Cin0);
    MIS_REQUIRE(d->Cin0 == d->Cin || d->x1 != nullptr, MIS_EINVAL, "conv_igemm: x1 missing");
    MIS_REQUIRE(d->st_mode == 0 || f3, MIS_EUNSUPPORTED, "conv_igemm: st_mode needs the fp32 3x3x3 all-DMA path (mis_conv_stats_rows(d) == 0 for this descriptor)");
    const bool gnb = d->gn_p != nullptr;          // GroupNorm backward in the epilogue: only the 3-D ping-pong kernels carry it
    MIS_REQUIRE(!gnb || pp3, MIS_EUNSUPPORTED, "conv_igemm: gn_p needs the bf16 3x3x3 ping-pong path (single source, q / r / mask given, no bias / ReLU)");
    MIS_REQUIRE(d->Cout0 > 0 && d->Cout0 <= d->Cout && (d->Cout0 % 64 == 0 || ((gnb || f3) && d->Cout0 % 32 == 0)), MIS_EINVAL, "conv_igemm: Cout0 %d", d->Cout0);
    MIS_REQUIRE(d->Cout0 == d->Cout || d->y1 != nullptr || gnb, MIS_EINVAL, "conv_igemm: y1 missing");
    MIS_REQUIRE(d->x0_ld % EPC == 0 && d->y0_ld % EPC == 0, MIS_EINVAL, "conv_igemm: ld must keep 16-byte alignment");
    MIS_REQUIRE(d->x1 == nullptr || d->x1_ld % EPC == 0, MIS_EINVAL, "conv_igemm: x1_ld alignment");
    MIS_REQUIRE(d->y1 == nullptr || d->y1_ld % EPC == 0, MIS_EINVAL, "conv_igemm: y1_ld alignment");
    MIS_REQUIRE(d->mask == nullptr || d->mask_ld % EPC == 0, MIS_EINVAL, "conv_igemm: mask_ld alignment");
    MIS_REQUIRE(d->x0_D > 0 && d->x0_H > 0 && d->x0_W > 0, MIS_EINVAL, "conv_igemm: x0 grid");
    MIS_REQUIRE(d->x1 == nullptr || (d->x1_D > 0 && d->x1_H > 0 && d->x1_W > 0), MIS_EINVAL, "conv_igemm: x1 grid");
    MIS_REQUIRE((d->in_scale == nullptr) == (d->in_shift == nullptr), MIS_EINVAL, "conv_igemm: in_scale/in_shift");
    {
        const SrcView srcs[2] = {{d->x0, d->x0_ld, d->x0_D, d->x0_H, d->x0_W}, {d->x1, d->x1_ld, d->x1_D, d->x1_H, d->x1_W}};
        for (int i = 0; i < 2; ++i) {
            if (srcs[i].p == nullptr) continue;
            const bool okD = srcs[i].D == d->D || (srcs[i].D * 2 == d->D);
            const bool okH = srcs[i].H == d->H || (srcs[i].H * 2 == d->H);
            const bool okW = srcs[i].W == d->W || (srcs[i].W * 2 == d->W);
            MIS_REQUIRE(okD && okH && okW, MIS_EUNSUPPORTED,
                        "conv_igemm: source grid must equal the pixel grid or be exactly half of it");
        }
    }
    const int modes[2] = {d->y0_mode, d->y1_mode};
    const int views[2] = {d->Cout0, d->Cout - d->Cout0};
    for (int i = 0; i < 2; ++i) {
        if (i == 1 && d->y1 == nullptr) break;
        MIS_REQUIRE(modes[i] >= 0 && modes[i] <= 2, MIS_EINVAL, "conv_igemm: output mode");
        if (modes[i] != MIS_OUT_PLAIN) MIS_REQUIRE(!d->is3d, MIS_EUNSUPPORTED, "conv_igemm: (un)shuffle is 2-D only");
        if (modes[i] == MIS_OUT_SHUFFLE2)
            MIS_REQUIRE(views[i] % 256 == 0, MIS_EUNSUPPORTED, "conv_igemm: shuffle needs 4*Cq columns with Cq %% 64 == 0");
        if (modes[i] == MIS_OUT_UNSHUFFLE2)
            MIS_REQUIRE(d->H % 2 == 0 && d->W % 2 == 0, MIS_EUNSUPPORTED, "conv_igemm: unshuffle needs even H, W");
    }
    if (d->relu_bits != nullptr || d->mask_bits != nullptr) {
        MIS_REQUIRE(d->dtype == MIS_BF16 && !d->is3d && d->D == 1, MIS_EUNSUPPORTED, "conv_igemm: ReLU bits are a bf16 2-D feature");
        MIS_REQUIRE(d->mask_bits == nullptr || d->mask == nullptr, MIS_EINVAL, "conv_igemm: mask and mask_bits are alternatives");
        MIS_REQUIRE(d->relu_bits == nullptr || (d->relu && d->Cout0 == d->Cout && d->y0_mode == MIS_OUT_PLAIN), MIS_EUNSUPPORTED,
                    "conv_igemm: relu_bits needs relu = 1 and one plain destination");
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    g_conv_bits_fused = false;
    const int rc = d->dtype == MIS_BF16 ? dispatch<__bf16>(d, s) : dispatch<float>(d, s);
    // a kernel without the fused producer (everything but the column-segment kernels): the stand-alone pass over the stored output
    if (rc == MIS_OK && d->relu_bits != nullptr && !g_conv_bits_fused) return mis_relu_bits(d->y0, d->y0_ld, d->N, d->H, d->W, d->Cout, d->relu_bits, stream);
    return rc;
}
