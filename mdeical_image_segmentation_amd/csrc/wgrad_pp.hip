// Ping-pong weight-gradient kernel for the bf16 2-D 3x3 layers (gfx950):  dW[tap][ci][co] = sum_pixels X[pixel + tap][ci] * dY[pixel][co]
// (the weight part of aten::convolution_backward for nn.Conv2d(k3, p1), reference model/unet2d/layers.py:122,125).
//
// Same arithmetic, fragment mapping and slab format as wgrad_kernel (wgrad.hip): a block owns one (64 ci x 64 co) tile for all 9 taps, the MFMA K
// dimension is the pixel axis, both operands are read from [pixel][channel] LDS images with the transposing ds_read_b64_tr_b16, fp32 partial slabs
// are reduced in fixed order by the kernels of wgrad.hip.  What changes is the schedule (the conv_pp.hip structure):
//   * 8 waves per block; a step stages one 16 x 16-pixel tile: the input halo (18 x 18 px x 64 ci) and the dY tile (256 px x 64 co), both by
//     buffer_load ... lds (no register staging; zero padding / ragged edges = out-of-range buffer offsets, which read as zero), double-buffered,
//     the next tile's 73 DMA instructions spread over the current tile's R segments and retired by one vmcnt(0) in the last slot of the step;
//   * every wave accumulates 16 ci x 32 co x 9 taps (72 accumulator registers, which leaves room for 36-MFMA segments); wave group 0 (waves 0-3) owns
//     input channels 0-31 of the tile, group 1 (waves 4-7) channels 32-63, both walk all 256 pixels - the groups are staggered by one barrier so
//     that one wave per SIMD is always inside an MFMA cluster while its partner reads fragments and issues DMAs;
//   * blocks are persistent over a contiguous range of pixel tiles (split-K): <= 256 blocks, one slab per block.
// LDS images: 128-byte pixel rows, the 16-byte chunk position XORed with (column & 7) (halo column for X, tile column for dY) - 8 consecutive pixels
// of a row then cover all 64 banks once for the transposing reads, a filter-row shift stays an immediate offset and the DMA fill stays lane-linear
// (the XOR goes into the per-lane SOURCE offset).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "dispatch_cfg.hpp"
#include "wgrad_args.hpp"

typedef __attribute__((address_space(3))) void wp_lds_void_t;
typedef __attribute__((address_space(3))) s16x4 wp_lds_s16x4;
typedef __attribute__((address_space(3))) char wp_lds_char_t;

namespace {
constexpr int WP_TH = 16, WP_TW = 16, WP_HW = 18, WP_HP = 18 * 18;
constexpr int WP_PITEMS = WP_HP * 8;                  // 16-byte items of the halo image
constexpr int WP_PINSTR = (WP_PITEMS + 63) / 64;      // 41 wave-instructions (the image is padded to 41 KiB)
constexpr int WP_PBUF = WP_PINSTR * 1024;
constexpr int WP_QINSTR = 32;                         // 256 px x 8 items / 64
constexpr int WP_QBUF = WP_QINSTR * 1024;
constexpr int WP_STAGE = WP_PBUF + WP_QBUF;
constexpr int WP_NINSTR = WP_PINSTR + WP_QINSTR;      // 73 per tile
constexpr int WP_PER_WAVE = (WP_NINSTR + 7) / 8;      // 10 (the last round only reaches wave 0)
constexpr int WP_ROWB = WP_HW * 128;
constexpr int WP_OOB = (int)0xFFFFFF00u;             // past every per-image buffer this kernel accepts (< 4 GiB - 256)

template <typename F, int... I> __device__ __forceinline__ void wp_static_for_impl(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void wp_static_for(F&& f) { wp_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ void wp_dma16(__amdgpu_buffer_rsrc_t r, int voff, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (wp_lds_void_t*)lds_dst_wave_uniform, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wp_make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// Transposing LDS read as INLINE ASM: through the builtin, hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in front of every such read while an LDS-DMA is in flight
// (it cannot tell the read from the DMA's destination), which serialises the whole prefetch pipeline (measured: 350 instead of 800+ TFLOP/s).  The asm form is
// invisible to that analysis; in exchange NOTHING waits for the result automatically: every use sits behind the explicit lgkmcnt(0) + sched_barrier of its R segment
// (cdna_hip_programming.md §5.4 rule 18, §5.7).
template <int OFF> __device__ __forceinline__ s16x4 wp_tr_read(uint32_t lds_addr) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF));
    return r;
}
// fragment for lane (i = lane & 15, g = lane >> 4): 8 pixels (k = 8g + 4s + e) of channel i of a 16-channel block: two transposing reads (pixel quads s = 0, 1)
template <int OFF> __device__ __forceinline__ bf16x8_t wp_frag(uint32_t addr_s0, uint32_t addr_s1) {
    const s16x4 lo = wp_tr_read<OFF>(addr_s0);
    const s16x4 hi = wp_tr_read<OFF>(addr_s1);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, r);
}
__device__ __forceinline__ float wp_sum8(const bf16x8_t& v) {     // fixed-order sum of the 8 bf16 values
    const u32x4 w = __builtin_bit_cast(u32x4, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t u = w[i];
        s += __uint_as_float(u << 16);
        s += __uint_as_float(u & 0xffff0000u);
    }
    return s;
}
}   // namespace

// KSS = 32-pixel k-steps per segment: a segment = KSS k-steps x all 9 taps = 18 * KSS MFMAs per wave.
// Wave w: input channels 16*(w>>1) .. +15 (one A fragment per tap), output channels 32*(w&1) .. +31 (two B fragments): 9 x 1 x 2 accumulator tiles = 72
// registers.  Both wave groups (w < 4 / w >= 4) walk ALL 256 pixels of the staged tile (8 k-steps) for their half of the input channels, so a block
// writes ONE slab.  Per segment a wave reads 18 * KSS A fragments and 2 * KSS B fragments (two transposing reads each).
template <int KSS>
__global__ __launch_bounds__(512, 2) void wgrad_pp_kernel(const WgArgs a) {
    constexpr int NS = 8 / KSS;                   // segments per step (step = one 16 x 16 pixel tile = 8 k-steps)
    constexpr int DPS = (WP_PER_WAVE + NS - 2) / (NS - 1);   // DMA instructions per wave and R segment (none in the last segment of a step)
    static_assert(DPS * (NS - 1) >= WP_PER_WAVE, "");
    constexpr int NPJ = (WP_PINSTR + 7) / 8;      // halo-image DMA instructions per wave (6; ids wave + 8j)
    constexpr int NQJ = WP_QINSTR / 8;            // dY-tile DMA instructions per wave (4; tile rows 2*(wave + 8j)/2 ...)
    static_assert(NPJ + NQJ == WP_PER_WAVE, "");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wi = wave >> 1, wj = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int q = li >> 2, pp = li & 3;

    const int npairs = a.nCi * a.nCo;
    int v = xcd_remap(blockIdx.x, gridDim.x);
    const int pair = v % npairs;
    const int split = v / npairs;
    const int ci_t = pair / a.nCo, co_t = pair - ci_t * a.nCo;
    const int ci0 = ci_t * 64, co0 = co_t * 64;
    const int t_begin = split * a.tps;
    int t_end = t_begin + a.tps;
    if (t_end > a.ntiles) t_end = a.ntiles;
    const int tpi = a.tilesH * a.tilesW;
    auto tile_coords = [&](int t, int& n, int& h0, int& w0) {
        n = t / tpi;
        const int r = t - n * tpi;
        const int th = r / a.tilesW;
        h0 = th * WP_TH;
        w0 = (r - th * a.tilesW) * WP_TW;
    };

    // ---- per-lane fragment offsets: pixel quad s of lane group lg is quad u = (lg&1) + 2s + 4(lg>>1) of the k-step's 2 x 16 pixels ----
    int qoff[2], poff[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int u = (lg & 1) + 2 * s + 4 * (lg >> 1);
        const int hy = u >> 2, wx = (u & 3) * 4 + q;      // this lane's pixel (row within the k-step, column) of the quad
        qoff[s] = (hy * 16 + wx) * 128 + (((wj * 4 + (pp >> 1)) ^ (wx & 7)) << 4) + (pp & 1) * 8;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int px = wx + kw;
            poff[s][kw] = (hy * WP_HW + px) * 128 + (((wi * 2 + (pp >> 1)) ^ (px & 7)) << 4) + (pp & 1) * 8;
        }
    }

    // ---- per-lane DMA source offsets, computed once (a DMA issue in the loop is then 2-10 VALU ops) ----
    // halo image of X, instruction id = wave + 8j: lane -> 16-byte slot -> (halo pixel, chunk position); the chunk is XORed with (halo column & 7)
    unsigned prel[NPJ];           // byte offset relative to halo pixel (0, 0), chunk included
    int pcoord[NPJ];              // halo row | halo column << 8; -1 = no item (tail of the last instruction / no such instruction)
#pragma unroll
    for (int j = 0; j < NPJ; ++j) {
        const int id = wave + 8 * j;
        const int item = id * 64 + lane;
        const int p = item >> 3, pos = item & 7;
        const int py = p / WP_HW, px = p - py * WP_HW;
        const bool have = id < WP_PINSTR && item < WP_PITEMS;
        prel[j] = (unsigned)(((py * a.W + px) * a.x0.ld + ((pos ^ (px & 7)) << 3)) * 2);
        pcoord[j] = have ? (py | (px << 8)) : -1;
    }
    // dY tile, instruction i = wave + 8j covers tile row i >> 1, columns 8 * (i & 1) .. + 7: lane -> (column 8*(i&1) + (lane >> 3), chunk position lane & 7)
    const unsigned qlane = (unsigned)((((lane >> 3)) * a.dy_ld + (((lane & 7) ^ (lane >> 3)) << 3)) * 2);

    // buffer resources are rebuilt PER IMAGE (scalar ops only): 32-bit offsets then only have to span one image, whatever the batch size
    const unsigned img_x = (unsigned)(((long long)a.H * a.W - 1) * a.x0.ld + a.Cin) * 2u, img_q = (unsigned)(((long long)a.H * a.W - 1) * a.dy_ld + a.Cout) * 2u;
    const char* const xb = reinterpret_cast<const char*>(a.x0.p);
    const char* const qb = reinterpret_cast<const char*>(a.dy);

    // DMA instruction jj (0..9) of this wave for the tile (n, h0, w0): jj < NPJ -> halo image, else dY tile
    auto issue = [&](auto jc, int n, int h0, int w0, char* stage) {
        constexpr int jj = decltype(jc)::value;
        if constexpr (jj < NPJ) {
            const int id = wave + 8 * jj;
            if (id >= WP_PINSTR) return;                      // wave-uniform
            const __amdgpu_buffer_rsrc_t rx = wp_make_rsrc(xb + (size_t)n * a.H * a.W * a.x0.ld * 2, img_x);
            // byte offset of halo pixel (0,0) = image pixel (h0-1, w0-1) within the image: "negative" (wraps) on the top / left border, where only valid items add to it
            const unsigned toff = (unsigned)((((h0 - 1) * a.W + (w0 - 1)) * a.x0.ld + ci0) * 2);
            const bool interior = h0 >= 1 && h0 + WP_TH + 1 <= a.H && w0 >= 1 && w0 + WP_TW + 1 <= a.W;        // block-uniform
            bool ok = pcoord[jj] >= 0;
            if (!interior) {
                const int py = pcoord[jj] & 0xff, px = pcoord[jj] >> 8;
                ok = ok && (unsigned)(h0 - 1 + py) < (unsigned)a.H && (unsigned)(w0 - 1 + px) < (unsigned)a.W;
            }
            wp_dma16(rx, ok ? (int)(toff + prel[jj]) : WP_OOB, stage + id * 1024);
        } else {
            const int i = wave + 8 * (jj - NPJ);
            const __amdgpu_buffer_rsrc_t rq = wp_make_rsrc(qb + (size_t)n * a.H * a.W * a.dy_ld * 2, img_q);
            const int hy = i >> 1, wx0 = (i & 1) * 8;
            const unsigned toff = (unsigned)((((h0 + hy) * a.W + (w0 + wx0)) * a.dy_ld + co0) * 2);
            const bool ok = (h0 + hy) < a.H && (w0 + wx0 + (lane >> 3)) < a.W;
            wp_dma16(rq, ok ? (int)(toff + qlane) : WP_OOB, stage + WP_PBUF + i * 1024);
        }
    };

    f32x4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int fj = 0; fj < 2; ++fj) acc[t][fj] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = (a.bias_partial != nullptr) && (ci_t == 0) && (wi == 0);     // wave-uniform: waves 0 and 1
    float bsum[2] = {0.f, 0.f};

    int n, h0, w0;
    tile_coords(t_begin, n, h0, w0);
    wp_static_for<WP_PER_WAVE>([&](auto jc) { issue(jc, n, h0, w0, smem); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int sel = 0;
    if (grp == 1) __builtin_amdgcn_s_barrier();       // the stagger: group 1 runs one slot behind group 0
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        const bool has_next = t + 1 < t_end;
        int nn = n, nh0 = h0, nw0 = w0;
        if (has_next) tile_coords(t + 1, nn, nh0, nw0);
        const uint32_t lds0 = (uint32_t)(uintptr_t)smem;                  // LDS byte address of the dynamic segment
        const uint32_t lds_p = lds0 + sel * WP_STAGE;
        const uint32_t lds_q = lds0 + sel * WP_STAGE + WP_PBUF;
        char* nstage = smem + (sel ^ 1) * WP_STAGE;
        wp_static_for<NS>([&](auto sgc) {
            constexpr int sg = decltype(sgc)::value;
            // ================= R segment =================
            // (a block's last tile issues nothing: keep its first fragment reads ~64 cycles away from the barrier behind the other group's vmcnt(0) - see the note in
            // wgrad_pp_stream_kernel's prologue)
            if constexpr (sg == 0) {
                if (!has_next) __builtin_amdgcn_s_sleep(1);
            }
            if constexpr (sg < NS - 1) {
                if (has_next) {
                    wp_static_for<DPS>([&](auto dc) {
                        constexpr int jj = sg * DPS + decltype(dc)::value;
                        if constexpr (jj < WP_PER_WAVE) issue(std::integral_constant<int, jj>{}, nn, nh0, nw0, nstage);
                    });
                }
            }
            bf16x8_t B[KSS][2], A[KSS][9];
            wp_static_for<KSS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int ks = sg * KSS + s;
                wp_static_for<2>([&](auto fjc) {
                    constexpr int fj = decltype(fjc)::value;
                    B[s][fj] = wp_frag<ks * (2 * 16 * 128)>(lds_q + (qoff[0] ^ (fj << 5)), lds_q + (qoff[1] ^ (fj << 5)));
                });
                wp_static_for<9>([&](auto tc) {
                    constexpr int tap = decltype(tc)::value;
                    constexpr int kh = tap / 3, kw = tap % 3;
                    A[s][tap] = wp_frag<(ks * 2 + kh) * WP_ROWB>(lds_p + poff[0][kw], lds_p + poff[1][kw]);
                });
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);               // nothing that consumes a fragment may move above the wait (the asm reads are opaque to hipcc)
            if (do_bias) {
#pragma unroll
                for (int s = 0; s < KSS; ++s) {
                    bsum[0] += wp_sum8(B[s][0]);
                    bsum[1] += wp_sum8(B[s][1]);
                }
            }
            if constexpr (sg == NS - 1) {
                if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // group 1's last slot of the step: its DMAs for the next tile have landed
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ================= M segment =================
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < KSS; ++s)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int fj = 0; fj < 2; ++fj)
                        acc[tap][fj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][tap], B[s][fj], acc[tap][fj], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if constexpr (sg == NS - 1) {
                if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // group 0's last slot of the step
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        });
        sel ^= 1;
        n = nn; h0 = nh0; w0 = nw0;
    }

    // ---- the block's partial slab: partial[split][tap][ci][co]; bias column sums likewise ----
    float* out = a.partial + (size_t)split * a.TT * a.Cin * a.Cout;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int fj = 0; fj < 2; ++fj) {
            const int co = co0 + (wj * 2 + fj) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + wi * 16 + lg * 4 + r;
                out[((size_t)tap * a.Cin + ci) * a.Cout + co] = acc[tap][fj][r];
            }
        }
    if (do_bias) {
#pragma unroll
        for (int fj = 0; fj < 2; ++fj) {
            float s = bsum[fj];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lg == 0) a.bias_partial[(size_t)split * a.Cout + co0 + (wj * 2 + fj) * 16 + li] = s;
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();       // pairs with group 1's last barrier
}

// ---------------------------------------------------------------------------------------------------------
// Wide variant for Cout % 128 == 0: a block owns 64 ci x 128 co, a wave 16 ci x 64 co x 9 taps (ONE A fragment per tap against FOUR dY fragments: 36 MFMAs per
// 32-pixel k-step from 26 transposing reads instead of 44 - the 64 x 64 kernel above is bound by its R segments, i.e. by LDS reads per MFMA).  A step stages a
// 16 x 8-pixel tile: halo (18 x 10 px x 64 ci, 128-byte rows) + dY (128 px x 128 co, 256-byte rows with the chunk position XORed with (column & 7) << 1, which keeps
// a pixel's two 16-byte halves of a fragment apart from its neighbours' over all 64 banks).  Group 0 = output channels 0-63, group 1 = 64-127; a segment = one k-step.
// Template W3<SPLIT>: SPLIT = false is the layout described above (Cout % 128 == 0).  SPLIT = true serves the layers with 64 output channels per tile with the SAME
// wave tile (16 ci x 64 co): a block owns 64 ci x 64 co, stages 16 x 16-pixel tiles (dY with 128-byte rows), group 0 accumulates tile rows 0-7 and group 1 rows 8-15 into
// their own accumulators, and each group writes its own slab (2 slabs per block).
template <bool SPLIT> struct W3 {
    static constexpr int TH = SPLIT ? 16 : 8, PH = TH + 2;
    static constexpr int CO = SPLIT ? 64 : 128;                       // output channels per block
    static constexpr int QROW = CO * 2;                               // bytes per dY pixel row
    static constexpr int PITEMS = PH * WP_HW * 8;
    static constexpr int PINSTR = (PITEMS + 63) / 64;                 // 41 / 23
    static constexpr int PBUF = PINSTR * 1024;
    static constexpr int QINSTR = TH * 16 * (QROW / 16) / 64;         // 32 either way
    static constexpr int QBUF = QINSTR * 1024;
    static constexpr int STAGE = PBUF + QBUF;
    static constexpr int NPJ = (PINSTR + 7) / 8;                      // halo instructions per wave: 6 / 3
    static constexpr int NQJ = QINSTR / 8;                            // 4
    static constexpr int PER_WAVE = NPJ + NQJ;
};

// IS3D (round 3): the weight gradient of a 3x3x3 layer = three 3x3 weight gradients, one per depth slice kd of the filter, each against the input plane z + kd - 1:
// kd joins (ci tile, co tile) in the block's identity (the three blocks of a pair walk the same dY tiles and input planes one apart, and sit next to each other in the
// launch order, i.e. on one XCD), a "pixel tile" is a 2-D tile of one depth plane (NDHWC: the planes of a batch are one contiguous sequence of H x W images), planes
// outside the volume read as zero through out-of-range DMA offsets, and the slab row is kd*9 + tap of TT = 27.
template <bool SPLIT, bool IS3D = false>
__global__ __launch_bounds__(512, 2) void wgrad_pp_wide_kernel(const WgArgs a) {
    using G = W3<SPLIT>;
    constexpr int NS = 4;                              // segments per step = the k-steps of a group's 128 pixels
    constexpr int DPS = (G::PER_WAVE + NS - 2) / (NS - 1);   // DMA instructions per wave and R segment (none in the last one)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wi = wave & 3, wj = SPLIT ? 0 : (wave >> 2);
    const int li = lane & 15, lg = lane >> 4;
    const int q = li >> 2, pp = li & 3;

    constexpr int KDN = IS3D ? 3 : 1;
    const int npairs = a.nCi * a.nCo * KDN;            // nCo counts G::CO-column tiles
    int v = xcd_remap(blockIdx.x, gridDim.x);
    const int pair = v % npairs;
    const int split = v / npairs;
    const int kd = IS3D ? pair % 3 : 0;
    const int pc = IS3D ? pair / 3 : pair;
    const int ci_t = pc / a.nCo, co_t = pc - ci_t * a.nCo;
    const int ci0 = ci_t * 64, co0 = co_t * G::CO;
    int t_begin = split * a.tps;
    int t_end = t_begin + a.tps;
    if (t_end > a.ntiles) t_end = a.ntiles;
    if (a.spb > 0) {               // per-sample split-K: this split's tile range lies inside ONE sample
        const int smp = split / a.spb;
        t_begin = smp * a.tpsamp + (split - smp * a.spb) * a.tps;
        t_end = t_begin + a.tps;
        if (t_end > (smp + 1) * a.tpsamp) t_end = (smp + 1) * a.tpsamp;
    }
    const int tpi = a.tilesH * a.tilesW;
    auto tile_coords = [&](int t, int& n, int& h0, int& w0) {
        n = t / tpi;
        const int r = t - n * tpi;
        const int th = r / a.tilesW;
        h0 = th * G::TH;
        w0 = (r - th * a.tilesW) * WP_TW;
    };
    auto src_plane = [&](int n, int& xn, bool& zok) {      // once per tile (the modulo is scalar work)
        xn = n;
        zok = true;
        if constexpr (IS3D) {
            const int z = n % a.D + kd - 1;
            zok = (unsigned)z < (unsigned)a.D;
            xn = zok ? n + kd - 1 : n;
        }
    };

    int qoff[2], poff[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int u = (lg & 1) + 2 * s + 4 * (lg >> 1);
        const int hy = u >> 2, wx = (u & 3) * 4 + q;
        if (SPLIT) qoff[s] = (hy * 16 + wx) * 128 + (((pp >> 1) ^ (wx & 7)) << 4) + (pp & 1) * 8;                       // fragment fj: ^ (fj << 5)
        else qoff[s] = (hy * 16 + wx) * 256 + (((wj * 8 + (pp >> 1)) ^ ((wx & 7) << 1)) << 4) + (pp & 1) * 8;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int px = wx + kw;
            poff[s][kw] = (hy * WP_HW + px) * 128 + (((wi * 2 + (pp >> 1)) ^ (px & 7)) << 4) + (pp & 1) * 8;
        }
    }
    unsigned prel[G::NPJ];
    int pcoord[G::NPJ];
#pragma unroll
    for (int j = 0; j < G::NPJ; ++j) {
        const int id = wave + 8 * j;
        const int item = id * 64 + lane;
        const int p = item >> 3, pos = item & 7;
        const int py = p / WP_HW, px = p - py * WP_HW;
        const bool have = id < G::PINSTR && item < G::PITEMS;
        prel[j] = (unsigned)(((py * a.W + px) * a.x0.ld + ((pos ^ (px & 7)) << 3)) * 2);
        pcoord[j] = have ? (py | (px << 8)) : -1;
    }
    // dY instruction i = wave + 8j.  SPLIT: tile row i >> 1, columns 8 * (i & 1) + (lane >> 3), chunk position lane & 7 (XOR column & 7); (i & 1) == (wave & 1) for every j.
    //                              else: tile row i >> 2, columns 4 * (i & 3) + (lane >> 4), chunk position lane & 15 (XOR (column & 7) << 1); (i & 3) == (wave & 3).
    const int qcol = SPLIT ? 8 * (wave & 1) + (lane >> 3) : 4 * (wave & 3) + (lane >> 4);
    const unsigned qlane = SPLIT ? (unsigned)((qcol * a.dy_ld + (((lane & 7) ^ (qcol & 7)) << 3)) * 2)
                                 : (unsigned)((qcol * a.dy_ld + (((lane & 15) ^ ((qcol & 7) << 1)) << 3)) * 2);

    const unsigned img_x = (unsigned)(((long long)a.H * a.W - 1) * a.x0.ld + a.Cin) * 2u, img_q = (unsigned)(((long long)a.H * a.W - 1) * a.dy_ld + a.Cout) * 2u;
    const char* const xb = reinterpret_cast<const char*>(a.x0.p);
    const char* const qb = reinterpret_cast<const char*>(a.dy);
    // IS3D: n is the plane index (sample * D + z) of the dY tile; its input plane is xn = n + kd - 1 when that lies inside the volume (zok), else it reads as zeros
    auto issue = [&](auto jc, int n, int h0, int w0, char* stage, int xn, bool zok) {
        constexpr int jj = decltype(jc)::value;
        if constexpr (jj < G::NPJ) {
            const int id = wave + 8 * jj;
            if (id >= G::PINSTR) return;
            const __amdgpu_buffer_rsrc_t rx = wp_make_rsrc(xb + (size_t)xn * a.H * a.W * a.x0.ld * 2, img_x);
            const unsigned toff = (unsigned)((((h0 - 1) * a.W + (w0 - 1)) * a.x0.ld + ci0) * 2);
            const bool interior = h0 >= 1 && h0 + G::TH + 1 <= a.H && w0 >= 1 && w0 + WP_TW + 1 <= a.W;
            bool ok = pcoord[jj] >= 0 && zok;
            if (!interior) {
                const int py = pcoord[jj] & 0xff, px = pcoord[jj] >> 8;
                ok = ok && (unsigned)(h0 - 1 + py) < (unsigned)a.H && (unsigned)(w0 - 1 + px) < (unsigned)a.W;
            }
            wp_dma16(rx, ok ? (int)(toff + prel[jj]) : WP_OOB, stage + id * 1024);
        } else {
            const int i = wave + 8 * (jj - G::NPJ);
            const __amdgpu_buffer_rsrc_t rq = wp_make_rsrc(qb + (size_t)n * a.H * a.W * a.dy_ld * 2, img_q);
            const int hy = SPLIT ? (i >> 1) : (i >> 2);
            const unsigned toff = (unsigned)((((h0 + hy) * a.W + w0) * a.dy_ld + co0) * 2);
            const bool ok = (h0 + hy) < a.H && (w0 + qcol) < a.W;
            wp_dma16(rq, ok ? (int)(toff + qlane) : WP_OOB, stage + G::PBUF + i * 1024);
        }
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) acc[t][fj] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient / column sums of dY: wave wi of a group sums dY fragment fj = wi with ONE extra MFMA per k-step against an all-ones A operand (every row of the
    // result is the column sum) - the VALU version (8 unpack + add per fragment on one wave per group) stretched that wave's R segment and with it every barrier
    const bool do_bias = (a.bias_partial != nullptr) && (ci_t == 0) && (!IS3D || kd == 1);     // block-uniform (3-D: the centre-slice block, whose input plane always exists)
    f32x4 bacc = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones8 = __builtin_bit_cast(bf16x8_t, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});

    int n, h0, w0, xn;
    bool zok;
    tile_coords(t_begin, n, h0, w0);
    src_plane(n, xn, zok);
    wp_static_for<G::PER_WAVE>([&](auto jc) { issue(jc, n, h0, w0, smem, xn, zok); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int sel = 0;
    if (grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        const bool has_next = t + 1 < t_end;
        int nn = n, nh0 = h0, nw0 = w0, nxn = xn;
        bool nzok = zok;
        if (has_next) {
            tile_coords(t + 1, nn, nh0, nw0);
            src_plane(nn, nxn, nzok);
        }
        const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
        const uint32_t lds_p = lds0 + sel * G::STAGE + (SPLIT ? grp * (8 * WP_ROWB) : 0);              // SPLIT: this group's 8 tile rows (+ 2 halo rows)
        const uint32_t lds_q = lds0 + sel * G::STAGE + G::PBUF + (SPLIT ? grp * (8 * 16 * 128) : 0);
        char* nstage = smem + (sel ^ 1) * G::STAGE;
        wp_static_for<NS>([&](auto sgc) {
            constexpr int ks = decltype(sgc)::value;
            // ================= R segment =================
            // (a block's last tile issues nothing: keep its first fragment reads ~64 cycles away from the barrier behind the other group's vmcnt(0) - see the note in
            // wgrad_pp_stream_kernel's prologue)
            if constexpr (ks == 0) {
                if (!has_next) __builtin_amdgcn_s_sleep(1);
            }
            if constexpr (ks < NS - 1) {
                if (has_next) {
                    wp_static_for<DPS>([&](auto dc) {
                        constexpr int jj = ks * DPS + decltype(dc)::value;
#ifdef WPT_SAME_TILE        // diagnostic: every prefetch re-reads the tile the block started with - the same DMA count from L2-resident lines (loaded-latency share of a step)
                        if constexpr (jj < G::PER_WAVE) issue(std::integral_constant<int, jj>{}, n, h0, w0, nstage, xn, zok);
#else
                        if constexpr (jj < G::PER_WAVE) issue(std::integral_constant<int, jj>{}, nn, nh0, nw0, nstage, nxn, nzok);
#endif
                    });
                }
            }
            bf16x8_t B[4], A[9];
            wp_static_for<4>([&](auto fjc) {
                constexpr int fj = decltype(fjc)::value;
                B[fj] = wp_frag<ks * (2 * 16 * G::QROW)>(lds_q + (qoff[0] ^ (fj << 5)), lds_q + (qoff[1] ^ (fj << 5)));
            });
            wp_static_for<9>([&](auto tc) {
                constexpr int tap = decltype(tc)::value;
                constexpr int kh = tap / 3, kw = tap % 3;
                A[tap] = wp_frag<(ks * 2 + kh) * WP_ROWB>(lds_p + poff[0][kw], lds_p + poff[1][kw]);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ks == NS - 1) {
                if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ================= M segment =================
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int fj = 0; fj < 4; ++fj) acc[tap][fj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[tap], B[fj], acc[tap][fj], 0, 0, 0);
            if (do_bias) bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, wi == 0 ? B[0] : (wi == 1 ? B[1] : (wi == 2 ? B[2] : B[3])), bacc, 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if constexpr (ks == NS - 1) {
                if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        });
        sel ^= 1;
#ifndef WPT_SAME_TILE
        n = nn; h0 = nh0; w0 = nw0; xn = nxn; zok = nzok;
#endif
    }

    const int se = SPLIT ? 2 * split + grp : split;            // SPLIT: one slab per wave group
    float* out = a.partial + (size_t)se * a.TT * a.Cin * a.Cout;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) {
            const int co = co0 + (wj * 4 + fj) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + wi * 16 + lg * 4 + r;
                out[((size_t)(kd * 9 + tap) * a.Cin + ci) * a.Cout + co] = acc[tap][fj][r];
            }
        }
    if (do_bias && lg == 0) a.bias_partial[(size_t)se * a.Cout + co0 + (wj * 4 + wi) * 16 + li] = bacc[0];      // row 0 of the ones x dY product: fragment fj = wi
    if (grp == 0) __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------------------------
// Row variant (round 3): the same block / wave tiles as wgrad_pp_wide_kernel, but a k-step is ONE ROW of 32 pixels (tile 32 x 4 pixels; SPLIT: 32 x 8, four rows per
// wave group).  The x fragment of tap (kh, kw) at k-step ks is the fragment of halo row ks + kh shifted by kw columns - so the nine fragments of a k-step are three rows x
// three shifts, and two of the three rows were already read for the previous k-step: a fragment stays in registers for the three k-steps that use it (kh = 2, 1, 0) and a
// k-step after the first reads 3 new x fragments instead of 9 (+ 4 dY fragments): 14 transposing reads per 36 MFMAs instead of 26, 17 on average over a tile.  The wide
// kernel is bound by its R segments (PMC r02: MFMA pipe 62 % / 50 % busy); this is the weight-gradient analogue of the column-segment recipe of conv_ppc_kernel.
// LDS per stage: halo (6 | 10) x 34 px x 128 B + dY 128 px x 256 B | 256 px x 128 B = 58 | 75 KiB, double-buffered.  Needs W % 32 == 0 to waste nothing (the
// dispatcher keeps the 16-wide tiles otherwise).
template <bool SPLIT> struct W4 {
    static constexpr int TW = 32, HW = TW + 2;
    static constexpr int TH = SPLIT ? 8 : 4, PH = TH + 2;
    static constexpr int CO = SPLIT ? 64 : 128;
    static constexpr int QROW = CO * 2;                               // bytes per dY pixel row
    static constexpr int ROWB = HW * 128;                             // bytes per halo row
    static constexpr int PITEMS = PH * HW * 8;
    static constexpr int PINSTR = (PITEMS + 63) / 64;                 // 26 / 43
    static constexpr int PBUF = PINSTR * 1024;
    static constexpr int QINSTR = TH * TW * (QROW / 16) / 64;         // 32 either way
    static constexpr int QBUF = QINSTR * 1024;
    static constexpr int STAGE = PBUF + QBUF;
    static constexpr int NPJ = (PINSTR + 7) / 8;                      // halo instructions per wave: 4 / 6
    static constexpr int NQJ = QINSTR / 8;                            // 4
    static constexpr int PER_WAVE = NPJ + NQJ;
};

template <bool SPLIT, bool IS3D>
__global__ __launch_bounds__(512, 2) void wgrad_pp_row_kernel(const WgArgs a) {
    using G = W4<SPLIT>;
    constexpr int NS = 4;                              // segments per step = the rows (k-steps) of a group's 4 x 32 pixels
    constexpr int DPS = (G::PER_WAVE + NS - 2) / (NS - 1);   // DMA instructions per wave and R segment (none in the last one)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wi = wave & 3, wj = SPLIT ? 0 : (wave >> 2);
    const int li = lane & 15, lg = lane >> 4;
    const int q = li >> 2, pp = li & 3;

    constexpr int KDN = IS3D ? 3 : 1;
    const int npairs = a.nCi * a.nCo * KDN;
    int v = xcd_remap(blockIdx.x, gridDim.x);
    const int pair = v % npairs;
    const int split = v / npairs;
    const int kd = IS3D ? pair % 3 : 0;
    const int pc = IS3D ? pair / 3 : pair;
    const int ci_t = pc / a.nCo, co_t = pc - ci_t * a.nCo;
    const int ci0 = ci_t * 64, co0 = co_t * G::CO;
    int t_begin = split * a.tps;
    int t_end = t_begin + a.tps;
    if (t_end > a.ntiles) t_end = a.ntiles;
    if (a.spb > 0) {               // per-sample split-K: this split's tile range lies inside ONE sample
        const int smp = split / a.spb;
        t_begin = smp * a.tpsamp + (split - smp * a.spb) * a.tps;
        t_end = t_begin + a.tps;
        if (t_end > (smp + 1) * a.tpsamp) t_end = (smp + 1) * a.tpsamp;
    }
    const int tpi = a.tilesH * a.tilesW;
    auto tile_coords = [&](int t, int& n, int& h0, int& w0) {
        n = t / tpi;
        const int r = t - n * tpi;
        const int th = r / a.tilesW;
        h0 = th * G::TH;
        w0 = (r - th * a.tilesW) * G::TW;
    };
    auto src_plane = [&](int n, int& xn, bool& zok) {
        xn = n;
        zok = true;
        if constexpr (IS3D) {
            const int z = n % a.D + kd - 1;
            zok = (unsigned)z < (unsigned)a.D;
            xn = zok ? n + kd - 1 : n;
        }
    };

    // fragment offsets: pixel quad s of lane group lg is quad u = (lg&1) + 2s + 4(lg>>1) of the k-step's 32 pixels = columns 4u .. 4u+3 of the row (lane groups
    // 0,1 and 2,3 each address 8 consecutive pixels: all 64 banks once, as in the wide kernel)
    int qoff[2], poff[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int u = (lg & 1) + 2 * s + 4 * (lg >> 1);
        const int wx = u * 4 + q;
        if (SPLIT) qoff[s] = wx * 128 + (((pp >> 1) ^ (wx & 7)) << 4) + (pp & 1) * 8;                       // fragment fj: ^ (fj << 5)
        else qoff[s] = wx * 256 + (((wj * 8 + (pp >> 1)) ^ ((wx & 7) << 1)) << 4) + (pp & 1) * 8;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int px = wx + kw;
            poff[s][kw] = px * 128 + (((wi * 2 + (pp >> 1)) ^ (px & 7)) << 4) + (pp & 1) * 8;
        }
    }
    unsigned prel[G::NPJ];
    int pcoord[G::NPJ];
#pragma unroll
    for (int j = 0; j < G::NPJ; ++j) {
        const int id = wave + 8 * j;
        const int item = id * 64 + lane;
        const int p = item >> 3, pos = item & 7;
        const int py = p / G::HW, px = p - py * G::HW;
        const bool have = id < G::PINSTR && item < G::PITEMS;
        prel[j] = (unsigned)(((py * a.W + px) * a.x0.ld + ((pos ^ (px & 7)) << 3)) * 2);
        pcoord[j] = have ? (py | (px << 8)) : -1;
    }
    // dY instruction i = wave + 8j.  SPLIT: tile row i >> 2, column 8 * (i & 3) + (lane >> 3), chunk position lane & 7 (XOR column & 7); (i & 3) == (wave & 3), row = (wave >> 2) + 2j.
    //                              else: tile row i >> 3 = j, column 4 * (i & 7) + (lane >> 4) = 4 * wave + (lane >> 4), chunk position lane & 15 (XOR (column & 7) << 1).
    const int qcol = SPLIT ? 8 * (wave & 3) + (lane >> 3) : 4 * wave + (lane >> 4);
    const unsigned qlane = SPLIT ? (unsigned)((qcol * a.dy_ld + (((lane & 7) ^ (qcol & 7)) << 3)) * 2)
                                 : (unsigned)((qcol * a.dy_ld + (((lane & 15) ^ ((qcol & 7) << 1)) << 3)) * 2);

    const unsigned img_x = (unsigned)(((long long)a.H * a.W - 1) * a.x0.ld + a.Cin) * 2u, img_q = (unsigned)(((long long)a.H * a.W - 1) * a.dy_ld + a.Cout) * 2u;
    const char* const xb = reinterpret_cast<const char*>(a.x0.p);
    const char* const qb = reinterpret_cast<const char*>(a.dy);
    auto issue = [&](auto jc, int n, int h0, int w0, char* stage, int xn, bool zok) {
        constexpr int jj = decltype(jc)::value;
        if constexpr (jj < G::NPJ) {
            const int id = wave + 8 * jj;
            if (id >= G::PINSTR) return;
            const __amdgpu_buffer_rsrc_t rx = wp_make_rsrc(xb + (size_t)xn * a.H * a.W * a.x0.ld * 2, img_x);
            const unsigned toff = (unsigned)((((h0 - 1) * a.W + (w0 - 1)) * a.x0.ld + ci0) * 2);
            const bool interior = h0 >= 1 && h0 + G::TH + 1 <= a.H && w0 >= 1 && w0 + G::TW + 1 <= a.W;
            bool ok = pcoord[jj] >= 0 && zok;
            if (!interior) {
                const int py = pcoord[jj] & 0xff, px = pcoord[jj] >> 8;
                ok = ok && (unsigned)(h0 - 1 + py) < (unsigned)a.H && (unsigned)(w0 - 1 + px) < (unsigned)a.W;
            }
            wp_dma16(rx, ok ? (int)(toff + prel[jj]) : WP_OOB, stage + id * 1024);
        } else {
            const int i = wave + 8 * (jj - G::NPJ);
            const __amdgpu_buffer_rsrc_t rq = wp_make_rsrc(qb + (size_t)n * a.H * a.W * a.dy_ld * 2, img_q);
            const int hy = SPLIT ? (i >> 2) : (i >> 3);
            const unsigned toff = (unsigned)((((h0 + hy) * a.W + w0) * a.dy_ld + co0) * 2);
            const bool ok = (h0 + hy) < a.H && (w0 + qcol) < a.W;
            wp_dma16(rq, ok ? (int)(toff + qlane) : WP_OOB, stage + G::PBUF + i * 1024);
        }
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) acc[t][fj] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient / column sums of dY: wave wi of a group sums dY fragment fj = wi with ONE extra MFMA per k-step against an all-ones A operand (every row of the
    // result is the column sum) - the VALU version (8 unpack + add per fragment on one wave per group) stretched that wave's R segment and with it every barrier
    const bool do_bias = (a.bias_partial != nullptr) && (ci_t == 0) && (!IS3D || kd == 1);     // block-uniform (3-D: the centre-slice block, whose input plane always exists)
    f32x4 bacc = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones8 = __builtin_bit_cast(bf16x8_t, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});

    int n, h0, w0, xn;
    bool zok;
    tile_coords(t_begin, n, h0, w0);
    src_plane(n, xn, zok);
    wp_static_for<G::PER_WAVE>([&](auto jc) { issue(jc, n, h0, w0, smem, xn, zok); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int sel = 0;
    if (grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        const bool has_next = t + 1 < t_end;
        int nn = n, nh0 = h0, nw0 = w0, nxn = xn;
        bool nzok = zok;
        if (has_next) {
            tile_coords(t + 1, nn, nh0, nw0);
            src_plane(nn, nxn, nzok);
        }
        const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
        const uint32_t lds_p = lds0 + sel * G::STAGE + (SPLIT ? grp * (4 * G::ROWB) : 0);               // SPLIT: this group's 4 tile rows (+ 2 halo rows)
        const uint32_t lds_q = lds0 + sel * G::STAGE + G::PBUF + (SPLIT ? grp * (4 * G::TW * G::QROW) : 0);
        char* nstage = smem + (sel ^ 1) * G::STAGE;
        bf16x8_t A[6][3];                              // halo row r, shift kw: live for the k-steps r - 2 .. r
        wp_static_for<NS>([&](auto sgc) {
            constexpr int ks = decltype(sgc)::value;
            // ================= R segment =================
            // (a block's last tile issues nothing: keep its first fragment reads ~64 cycles away from the barrier behind the other group's vmcnt(0) - see the note in
            // wgrad_pp_stream_kernel's prologue)
            if constexpr (ks == 0) {
                if (!has_next) __builtin_amdgcn_s_sleep(1);
            }
#ifndef WPT_NO_DMA          // (WPT_NO_*: timing ablations of a diagnostic build, scripts/wgrad_ablate.sh - results are garbage, never shipped)
            if constexpr (ks < NS - 1) {
                if (has_next) {
                    wp_static_for<DPS>([&](auto dc) {
                        constexpr int jj = ks * DPS + decltype(dc)::value;
#ifdef WPT_SAME_TILE        // diagnostic: every prefetch re-reads the tile the block started with - the same DMA count from L2-resident lines (loaded-latency share of a step)
                        if constexpr (jj < G::PER_WAVE) issue(std::integral_constant<int, jj>{}, n, h0, w0, nstage, xn, zok);
#else
                        if constexpr (jj < G::PER_WAVE) issue(std::integral_constant<int, jj>{}, nn, nh0, nw0, nstage, nxn, nzok);
#endif
                    });
                }
            }
#endif
            bf16x8_t B[4];
            wp_static_for<4>([&](auto fjc) {
                constexpr int fj = decltype(fjc)::value;
                B[fj] = wp_frag<ks * (G::TW * G::QROW)>(lds_q + (qoff[0] ^ (fj << 5)), lds_q + (qoff[1] ^ (fj << 5)));
            });
            wp_static_for<3>([&](auto rc) {            // new halo rows of this k-step: 0, 1, 2 for the first, ks + 2 afterwards
                constexpr int rr = decltype(rc)::value;
                if constexpr (ks == 0 || rr == 2) {
                    constexpr int r = ks + rr;
                    wp_static_for<3>([&](auto kc) {
                        constexpr int kw = decltype(kc)::value;
                        A[r][kw] = wp_frag<r * G::ROWB>(lds_p + poff[0][kw], lds_p + poff[1][kw]);
                    });
                }
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ks == NS - 1) {
                if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ================= M segment =================
            __builtin_amdgcn_s_setprio(1);
#ifdef WPT_NO_MFMA
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks][0], B[0], acc[0][0], 0, 0, 0);
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw)
                    for (int fj = 0; fj < 4; ++fj) asm volatile("" ::"v"(A[ks + kh][kw]), "v"(B[fj]));
#else
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int fj = 0; fj < 4; ++fj)
                        acc[kh * 3 + kw][fj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks + kh][kw], B[fj], acc[kh * 3 + kw][fj], 0, 0, 0);
#endif
            if (do_bias) bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, wi == 0 ? B[0] : (wi == 1 ? B[1] : (wi == 2 ? B[2] : B[3])), bacc, 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if constexpr (ks == NS - 1) {
                if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        });
        sel ^= 1;
#ifndef WPT_SAME_TILE
        n = nn; h0 = nh0; w0 = nw0; xn = nxn; zok = nzok;
#endif
    }

    const int se = SPLIT ? 2 * split + grp : split;            // SPLIT: one slab per wave group
    float* out = a.partial + (size_t)se * a.TT * a.Cin * a.Cout;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) {
            const int co = co0 + (wj * 4 + fj) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + wi * 16 + lg * 4 + r;
                out[((size_t)(kd * 9 + tap) * a.Cin + ci) * a.Cout + co] = acc[tap][fj][r];
            }
        }
    if (do_bias && lg == 0) a.bias_partial[(size_t)se * a.Cout + co0 + (wj * 4 + wi) * 16 + li] = bacc[0];      // row 0 of the ones x dY product: fragment fj = wi
    if (grp == 0) __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------------------------
// Streaming variant of the row kernel (end of round 3; H % 4 == 0 [64-column tiles: 8], W % 32 == 0): the same block / wave tiles, fragments and arithmetic as
// wgrad_pp_row_kernel<false>, but a block walks DOWN a 32-pixel-wide strip of the image and its operands stream through a ring of ROW slots instead of two tile stages.
// Why: with every prefetch redirected to L2-resident lines (diagnostic build WPT_SAME_TILE) the row kernel's launches of a train step take 8.27 instead of 9.59 ms - a
// seventh of its time is loaded HBM latency that a prefetch distance of ONE 32 x 4-pixel tile (58 KB, issued over the three segments before the wait) cannot cover, and
// its 6-row halo per 4-row tile fetches every input row 1.5 times.
//   * stream element j of a strip segment of R output rows [ya, ya + R): input row ya - 1 + j (34 px x 64 ci, 128-byte pixels) + dY row ya + j - 2 (32 px x 128 co,
//     256-byte pixels; none for j < 2), j = 0 .. R + 1 - 13 DMA instructions (waves 0-4: 2, waves 5-7: 1), one 13 KiB slot of a 12-slot ring;
//   * step j reads element j's three input fragments (kw = 0..2) into the register row j % 3 and, from j = 2 on, its four dY fragments, and runs k-step j - 2: the
//     36 MFMAs of output row ya + j - 2 against register rows (j - 2, j - 1, j) = taps kh = 0, 1, 2.  Every step is the row kernel's LIGHT k-step (14 transposing reads);
//     each input row is fetched once per strip segment (R + 2 rows per R);
//   * each wave's only vector-memory traffic is this stream, so its in-order vmcnt counter works as a FIFO: element j + 11 is issued at step j, and the wait before
//     step j + 1 leaves ten elements (130 KB per CU) in flight.
// Tile order: the split-K plan is the row kernel's (32 x 4-pixel tiles, a.tps per block) with the tiles of an image numbered column-major, so a block's range is a run
// of whole or partial strips; each run restarts the register rows (two MFMA-free warm-up steps).
// Template WS<SPLIT>: SPLIT = false is the layout described above (block 64 ci x 128 co, ONE stream for the block, all eight waves issue it).  SPLIT = true serves the layers
// with 64 output channels per tile the way W4<true> does - a block owns 64 ci x 64 co, the two wave groups accumulate DIFFERENT pixels into their own accumulators and write
// their own slabs - as two independent streams: group g walks the g-th half of the block's tile range through its own ring of 8 slots (input row + dY row 32 px x 64 co =
// 9 KiB), issued by its own four waves (wave 0 of the group: 3 instructions per element, the others 2).  The groups only share the barriers; the one whose stream is shorter
// (a strip boundary more or less) idles through the difference.
// IS3D: as in the tile-staged kernels - kd joins the block's identity, a "plane" is one depth slice of one sample, the input row comes from plane + kd - 1 (zero outside
// the volume), the slab row is kd*9 + tap.
namespace {
constexpr int WS_HROW = 5 * 1024;                   // input-row image: 34 px x 128 B = 4352 B in 5 DMA instructions (the last a quarter full)
template <bool SPLIT> struct WS {
    static constexpr int QROW = SPLIT ? 4 * 1024 : 8 * 1024;        // dY-row image: 32 px x 128 | 256 B
    static constexpr int SLOT = WS_HROW + QROW;
    static constexpr int NSLOT = SPLIT ? 8 : 12;                    // per stream
    static constexpr int D = NSLOT - 1;                             // prefetch distance in elements
    static constexpr int LDS = (SPLIT ? 2 : 1) * NSLOT * SLOT;      // 147,456 | 159,744
};
}   // namespace

template <bool SPLIT, bool IS3D>
__global__ __launch_bounds__(512, 2) void wgrad_pp_stream_kernel(const WgArgs a) {
    using G = WS<SPLIT>;
    constexpr int WS_SLOT = G::SLOT, WS_NSLOT = G::NSLOT, WS_D = G::D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wi = wave & 3, wj = SPLIT ? 0 : (wave >> 2);
    const int li = lane & 15, lg = lane >> 4;
    const int q = li >> 2, pp = li & 3;
    char* const ring = smem + (SPLIT ? grp * (WS_NSLOT * WS_SLOT) : 0);

    constexpr int KDN = IS3D ? 3 : 1;
    const int npairs = a.nCi * a.nCo * KDN;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int pair = v % npairs;
    const int split = v / npairs;
    const int kd = IS3D ? pair % 3 : 0;
    const int pc = IS3D ? pair / 3 : pair;
    const int ci_t = pc / a.nCo, co_t = pc - ci_t * a.nCo;
    const int ci0 = ci_t * 64, co0 = co_t * (SPLIT ? 64 : 128);
    int t_begin = split * a.tps;
    int t_end = t_begin + a.tps;
    if (t_end > a.ntiles) t_end = a.ntiles;
    if (a.spb > 0) {               // per-sample split-K: this split's tile range lies inside ONE sample
        const int smp = split / a.spb;
        t_begin = smp * a.tpsamp + (split - smp * a.spb) * a.tps;
        t_end = t_begin + a.tps;
        if (t_end > (smp + 1) * a.tpsamp) t_end = (smp + 1) * a.tpsamp;
    }
    // streams run over 32 x 4-pixel SUB-tiles, column-major within a plane: the plan's tile t (32 x 4, or 32 x 8 for SPLIT) = sub-tile t, or 2t and 2t + 1
    const int sth = SPLIT ? 2 * a.tilesH : a.tilesH;            // sub-tile rows per plane
    const int stpi = sth * a.tilesW;
    const int cnt = t_end - t_begin;                            // SPLIT: sub-tiles per group
    auto group_range = [&](int g, int& u0, int& u1) {
        if (SPLIT) {
            u0 = 2 * t_begin + g * cnt;
            u1 = u0 + cnt;
        } else {
            u0 = t_begin;
            u1 = t_end;
        }
    };
    // segment starting at sub-tile u of a stream that ends at u_end: plane, first column, first row, rows
    auto segment = [&](int u, int u_end, int& n, int& w0, int& ya, int& R) {
        n = u / stpi;
        const int r = u - n * stpi;
        const int tw = r / sth, th = r - tw * sth;
        w0 = tw * 32;
        ya = th * 4;
        int nt = sth - th;
        if (nt > u_end - u) nt = u_end - u;
        R = nt * 4;
    };
    auto stream_total = [&](int g) {
        int u0, u1, tot = 0;
        group_range(g, u0, u1);
        for (int u = u0; u < u1;) {
            int n_, w_, y_, R_;
            segment(u, u1, n_, w_, y_, R_);
            tot += R_ + 2;
            u += R_ >> 2;
        }
        return tot;
    };
    int u_begin, u_end;
    group_range(grp, u_begin, u_end);
    const int total = stream_total(grp);                 // stream elements (= steps) of this wave's stream
    int steps_all = total;                                // ... and of the longer of the block's streams (the barriers are shared)
    if (SPLIT) {
        const int other = stream_total(grp ^ 1);
        if (other > steps_all) steps_all = other;
    }

    // fragment offsets inside a slot (wgrad_pp_row_kernel: one row)
    int qoff[2], poff[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int u = (lg & 1) + 2 * s + 4 * (lg >> 1);
        const int wx = u * 4 + q;
        if (SPLIT) qoff[s] = WS_HROW + wx * 128 + (((pp >> 1) ^ (wx & 7)) << 4) + (pp & 1) * 8;                       // fragment fj: ^ (fj << 5)
        else qoff[s] = WS_HROW + wx * 256 + (((wj * 8 + (pp >> 1)) ^ ((wx & 7) << 1)) << 4) + (pp & 1) * 8;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int px = wx + kw;
            poff[s][kw] = px * 128 + (((wi * 2 + (pp >> 1)) ^ (px & 7)) << 4) + (pp & 1) * 8;
        }
    }
    // DMA lane parts.  Input row, instruction i: item = i*64 + lane -> halo column item >> 3, chunk position item & 7 (XOR column & 7).  !SPLIT: instruction `wave` for
    // waves 0-4; SPLIT: instruction wi, and the quarter-full fifth one (columns 32, 33) by the group's wave 0.
    // dY row: !SPLIT: instruction `wave`, column 4*wave + (lane >> 4), chunk position lane & 15 (XOR (column & 7) << 1); SPLIT: instruction wi, column 8*wi + (lane >> 3),
    // chunk position lane & 7 (XOR column & 7).
    const int hinstr = SPLIT ? wi : wave;
    const int hitem = hinstr * 64 + lane;
    const int hpx = hitem < 34 * 8 ? (hitem >> 3) : 0x40000000;
    const unsigned hrel = (unsigned)((((hitem >> 3) * a.x0.ld) + (((hitem & 7) ^ ((hitem >> 3) & 7)) << 3)) * 2);
    const int hpx4 = lane < 16 ? 32 + (lane >> 3) : 0x40000000;                    // SPLIT, wave 0 of a group: the fifth instruction
    const unsigned hrel4 = (unsigned)((((32 + (lane >> 3)) * a.x0.ld) + (((lane & 7) ^ ((32 + (lane >> 3)) & 7)) << 3)) * 2);
    const int qcol = SPLIT ? 8 * wi + (lane >> 3) : 4 * wave + (lane >> 4);
    const unsigned qrel = SPLIT ? (unsigned)((qcol * a.dy_ld + (((lane & 7) ^ (qcol & 7)) << 3)) * 2)
                                : (unsigned)((qcol * a.dy_ld + (((lane & 15) ^ ((qcol & 7) << 1)) << 3)) * 2);
    const int qinstr = SPLIT ? wi : wave;
    const unsigned img_x = (unsigned)(((long long)a.H * a.W - 1) * a.x0.ld + a.Cin) * 2u, img_q = (unsigned)(((long long)a.H * a.W - 1) * a.dy_ld + a.Cout) * 2u;
    const char* const xb = reinterpret_cast<const char*>(a.x0.p);
    const char* const qb = reinterpret_cast<const char*>(a.dy);
    // one stream element -> slot.  Rows above / below the image are negative / past-the-end offsets of the per-plane resource (read as zero), the columns left / right of
    // it are tested (hpx), planes outside the volume (3-D: zok) and the elements without a dY row (j < 2) issue out of range, so that every element costs a wave the same
    // number of vmcnt events.  Everything scalar comes from the issue cursor below (resources per segment, byte offsets advanced by one row per element).
    auto issue = [&](const __amdgpu_buffer_rsrc_t& rx, const __amdgpu_buffer_rsrc_t& rq, unsigned toff, unsigned qoffs, int w0, bool zok, bool has_q, char* slot) {
        if (SPLIT || wave < 5) {
            const bool ok = zok && (unsigned)(w0 - 1 + hpx) < (unsigned)a.W;
            wp_dma16(rx, ok ? (int)(toff + hrel) : WP_OOB, slot + hinstr * 1024);
        }
        if (SPLIT && wi == 0) {
            const bool ok = zok && (unsigned)(w0 - 1 + hpx4) < (unsigned)a.W;
            wp_dma16(rx, ok ? (int)(toff + hrel4) : WP_OOB, slot + 4 * 1024);
        }
        wp_dma16(rq, has_q ? (int)(qoffs + qrel) : WP_OOB, slot + WS_HROW + qinstr * 1024);
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) acc[t][fj] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = (a.bias_partial != nullptr) && (ci_t == 0) && (!IS3D || kd == 1);     // block-uniform
    f32x4 bacc = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones8 = __builtin_bit_cast(bf16x8_t, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});

    // the issue cursor (sub-tile, segment, element) runs WS_D elements ahead of the steps
    int pu = u_begin, pn, pw0, pya, pR, pj = 0, pslot = 0, issued = 0;
    bool pzok = true;
    __amdgpu_buffer_rsrc_t prx, prq;
    unsigned ptoff, pqoff;                             // byte offsets of element pj's input row / dY row (minus the lane parts) inside plane pn
    const unsigned xrow = (unsigned)(a.W * a.x0.ld * 2), qrow = (unsigned)(a.W * a.dy_ld * 2);
    auto cursor_segment = [&]() {
        segment(pu, u_end, pn, pw0, pya, pR);
        int xn = pn;
        if constexpr (IS3D) {
            const int z = pn % a.D + kd - 1;
            pzok = (unsigned)z < (unsigned)a.D;
            xn = pzok ? pn + kd - 1 : pn;
        }
        prx = wp_make_rsrc(xb + (size_t)xn * a.H * a.W * a.x0.ld * 2, img_x);
        prq = wp_make_rsrc(qb + (size_t)pn * a.H * a.W * a.dy_ld * 2, img_q);
        ptoff = (unsigned)((((pya - 1) * a.W + (pw0 - 1)) * a.x0.ld + ci0) * 2);
        pqoff = (unsigned)((((pya - 2) * a.W + pw0) * a.dy_ld + co0) * 2);
    };
    cursor_segment();
    auto issue_next = [&]() {
        if (issued >= total) __builtin_amdgcn_s_sleep(2);          // (see the note behind the prologue's barrier)
        if (issued < total) {
            issue(prx, prq, ptoff, pqoff, pw0, pzok, pj >= 2, ring + pslot * WS_SLOT);
            ++issued;
            pslot = pslot == WS_NSLOT - 1 ? 0 : pslot + 1;
            ptoff += xrow;
            pqoff += qrow;
            if (++pj == pR + 2) {
                pj = 0;
                pu += pR >> 2;
                if (pu < u_end) cursor_segment();
            }
        }
    };
    for (int i = 0; i < WS_D; ++i) issue_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // Measured on this kernel (2 x 64 x 64 case, where nothing is issued between this barrier and the first reads): fragment reads that start within a few dozen cycles
    // of the barrier can still see the OLD contents of bytes whose LDS-DMA another wave has just waited for with vmcnt(0) - the counter runs slightly ahead of the LDS
    // write becoming visible to other waves.  Every other path of the ping-pong kernels has a DMA issue (30+ instructions) between such a barrier and the first
    // dependent read; where this kernel has none (here, and the steps at the end of the stream that issue nothing) it sleeps ~128 cycles instead.
    __builtin_amdgcn_s_sleep(2);

    int slot = 0, e = 0;                               // the element / step about to run
    if (grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    bf16x8_t A[3][3];                                  // register rows: input row of element j lives in A[j % 3] for steps j, j + 1, j + 2
    // one step; PH = j % 3 (compile time), mm = this step carries a k-step (j >= 2)
    auto step = [&](auto phc, const bool mm) __attribute__((always_inline)) {
        constexpr int PH = decltype(phc)::value;
        // ================= R segment =================
        issue_next();                                  // element e + WS_D -> the slot element e - 1 left (every reader of this ring is past it)
        const uint32_t sb = (uint32_t)(uintptr_t)ring + slot * WS_SLOT;
        bf16x8_t B[4];
        // the step's 14 transposing reads AND their wait are ONE asm statement: the register rows outlive the step and are written by several copies of this code (warm-up
        // steps, loop body, tail), so hipcc has to reconcile their registers with copies - and a copy placed between a stand-alone read and its s_waitcnt (which the
        // compiler cannot see belong together) would move a register the LDS has not written yet
        {
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            s16x4 r[14];
            asm volatile(
                "ds_read_b64_tr_b16 %0, %14\n\tds_read_b64_tr_b16 %1, %15\n\t"
                "ds_read_b64_tr_b16 %2, %16\n\tds_read_b64_tr_b16 %3, %17\n\t"
                "ds_read_b64_tr_b16 %4, %18\n\tds_read_b64_tr_b16 %5, %19\n\t"
                "ds_read_b64_tr_b16 %6, %20\n\tds_read_b64_tr_b16 %7, %21\n\t"
                "ds_read_b64_tr_b16 %8, %22\n\tds_read_b64_tr_b16 %9, %23\n\t"
                "ds_read_b64_tr_b16 %10, %24\n\tds_read_b64_tr_b16 %11, %25\n\t"
                "ds_read_b64_tr_b16 %12, %26\n\tds_read_b64_tr_b16 %13, %27\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]),
                  "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13])
                : "v"(sb + qoff[0]), "v"(sb + qoff[1]), "v"(sb + (qoff[0] ^ 32)), "v"(sb + (qoff[1] ^ 32)), "v"(sb + (qoff[0] ^ 64)), "v"(sb + (qoff[1] ^ 64)), "v"(sb + (qoff[0] ^ 96)),
                  "v"(sb + (qoff[1] ^ 96)), "v"(sb + poff[0][0]), "v"(sb + poff[1][0]),
                  "v"(sb + poff[0][1]), "v"(sb + poff[1][1]), "v"(sb + poff[0][2]), "v"(sb + poff[1][2])
                : "memory");
            auto mk = [](const s16x4& lo, const s16x4& hi) {
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                return __builtin_bit_cast(bf16x8_t, v);
            };
#pragma unroll
            for (int fj = 0; fj < 4; ++fj) B[fj] = mk(r[2 * fj], r[2 * fj + 1]);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) A[PH][kw] = mk(r[8 + 2 * kw], r[9 + 2 * kw]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // element e + 1 must be in LDS before the next step reads it: everything but the WS_D - 1 youngest elements (this wave: 1, 2 or 3 instructions each); near the
        // end of the stream fewer are in flight and the wait is for all of them
        const bool full = e + WS_D < total;            // element e + WS_D was issued in this step
        auto wait_stream = [&]() {
            if (!full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (SPLIT) {
                if (wi == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (WS_D - 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (WS_D - 1)) : "memory");
            } else {
                if (wave < 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (WS_D - 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WS_D - 1) : "memory");
            }
        };
        if (grp == 1) wait_stream();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ================= M segment =================
        if (mm) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int fj = 0; fj < 4; ++fj)
                        acc[kh * 3 + kw][fj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[(PH + 1 + kh) % 3][kw], B[fj], acc[kh * 3 + kw][fj], 0, 0, 0);
            if (do_bias) bacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, wi == 0 ? B[0] : (wi == 1 ? B[1] : (wi == 2 ? B[2] : B[3])), bacc, 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        if (grp == 0) wait_stream();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        slot = slot == WS_NSLOT - 1 ? 0 : slot + 1;
        ++e;
    };

#pragma unroll 1
    for (int u = u_begin; u < u_end;) {
        int n, w0, ya, R;
        segment(u, u_end, n, w0, ya, R);
        u += R >> 2;
        // R + 2 steps, the first two without MFMAs; register rows rotate with period 3
        step(std::integral_constant<int, 0>{}, false);
        step(std::integral_constant<int, 1>{}, false);
        step(std::integral_constant<int, 2>{}, true);
        int j = 3;
#pragma unroll 1
        for (; j + 3 <= R + 2; j += 3) {
            step(std::integral_constant<int, 0>{}, true);
            step(std::integral_constant<int, 1>{}, true);
            step(std::integral_constant<int, 2>{}, true);
        }
        if (j < R + 2) step(std::integral_constant<int, 0>{}, true);
        if (j + 1 < R + 2) step(std::integral_constant<int, 1>{}, true);
    }
    if constexpr (SPLIT) {
        // the other group's stream is a few steps longer (a strip boundary more): keep its barriers company
#pragma unroll 1
        for (; e < steps_all; ++e) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
    }

    const int se = SPLIT ? 2 * split + grp : split;            // SPLIT: one slab per wave group
    float* out = a.partial + (size_t)se * a.TT * a.Cin * a.Cout;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) {
            const int co = co0 + (wj * 4 + fj) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + wi * 16 + lg * 4 + r;
                out[((size_t)(kd * 9 + tap) * a.Cin + ci) * a.Cout + co] = acc[tap][fj][r];
            }
        }
    if (do_bias && lg == 0) a.bias_partial[(size_t)se * a.Cout + co0 + (wj * 4 + wi) * 16 + li] = bacc[0];
    if (grp == 0) __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------------------------
// Ping-pong weight gradient of the bf16 1x1 GEMMs (end of round 3):  dW[ci][co] = sum_pixels X[pixel][ci] * dY[pixel][co]  - the weight part of the transposed
// convolutions' backward (reference model/unet2d/layers.py:165; dY = the pixel-unshuffled gradient, Cout = 4*Cq).  The round-1 kernel (wgrad_kernel, 128 x 128 channel
// tiles, lock-step waves, register-staged operands) ran these at 0.41-0.48 PFLOP/s.  With one tap there is no fragment reuse across taps: a wave tile of 64 ci x 64 co needs
// 16 transposing reads per 16 MFMAs, so the kernel lives at the LDS read bandwidth (64 KB per 32-pixel k-step and CU = 512 cycles, exactly the MFMA time) - what can be
// removed is everything else: operands by LDS-DMA into a three-slot ring two elements ahead (one stream per wave, counted vmcnt), the two wave groups staggered by a barrier.
// Block = 128 ci x 256 co (for the 128 -> 256 layer the whole dW: each operand byte is fetched once), wave (wi = wave & 1, wj = wave >> 1) owns 64 ci x 64 co; a stream
// element = 64 consecutive pixels of the flattened (n, y, x) index (a 1x1 GEMM knows no image borders) = two k-steps: X 64 px x 128 ci + dY 64 px x 256 co = 48 KiB as
// twelve [32 px][64 ch] images with 128-byte pixels, the chunk position XORed with (pixel & 7) (the dY layout of wgrad_pp_wide_kernel<true>).
namespace {
constexpr int WK_XB = 16 * 1024, WK_QB = 32 * 1024, WK_SLOT = WK_XB + WK_QB, WK_NSLOT = 3;
}   // namespace

__global__ __launch_bounds__(512, 2) void wgrad1_pp_kernel(const WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wi = wave & 1, wj = wave >> 1;
    const int li = lane & 15, lg = lane >> 4;
    const int q = li >> 2, pp = li & 3;

    const int npairs = a.nCi * a.nCo;                 // 128-ci x 256-co tiles
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int pair = v % npairs;
    const int split = v / npairs;
    const int ci_t = pair / a.nCo, co_t = pair - ci_t * a.nCo;
    const int ci0 = ci_t * 128, co0 = co_t * 256;
    int e_begin = split * a.tps;                      // elements of 64 pixels
    int e_end = e_begin + a.tps;
    if (e_end > a.ntiles) e_end = a.ntiles;
    const int total = e_end - e_begin;
    if (total <= 0) return;                           // block-uniform (never: every split is non-empty)

    // fragment offsets inside a [32 px][64 ch] image; the 16-channel block f of a fragment: ^ (f << 5)
    int fo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int u = (lg & 1) + 2 * s + 4 * (lg >> 1);
        const int wx = u * 4 + q;
        fo[s] = wx * 128 + (((pp >> 1) ^ (wx & 7)) << 4) + (pp & 1) * 8;
    }
    // DMA: one instruction = 8 pixels x 128 bytes of one image: lane -> pixel lane >> 3, chunk position lane & 7 (XOR pixel & 7)
    const int pl = lane >> 3;
    const unsigned xrel = (unsigned)((pl * a.x0.ld + (((lane & 7) ^ pl) << 3)) * 2);
    const unsigned qrel = (unsigned)((pl * a.dy_ld + (((lane & 7) ^ pl) << 3)) * 2);
    const char* const xb = reinterpret_cast<const char*>(a.x0.p);
    const char* const qb = reinterpret_cast<const char*>(a.dy);
    // element el (64 pixels from pixel 64*el) -> slot: X instructions wave, wave + 8 (id = (ks*2 + blk)*4 + pg), dY instructions wave + 8k (id = (ks*4 + blk)*4 + pg);
    // the image / pixel-group part of the address is the instruction's scalar offset (the lane part is used as it is)
    auto issue = [&](int el, char* slot) {
        const size_t p0 = (size_t)el * 64;
        const __amdgpu_buffer_rsrc_t rx = wp_make_rsrc(xb + p0 * a.x0.ld * 2, (unsigned)(64 * a.x0.ld * 2));
        const __amdgpu_buffer_rsrc_t rq = wp_make_rsrc(qb + p0 * a.dy_ld * 2, (unsigned)(64 * a.dy_ld * 2));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int id = wave + 8 * k;
            const int ks = id >> 3, blk = (id >> 2) & 1, pg = id & 3;
            const int so = ((ks * 32 + pg * 8) * a.x0.ld + ci0 + blk * 64) * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (wp_lds_void_t*)(slot + (ks * 2 + blk) * 4096 + pg * 1024), 16, (int)xrel, so, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int id = wave + 8 * k;
            const int ks = id >> 4, blk = (id >> 2) & 3, pg = id & 3;
            const int so = ((ks * 32 + pg * 8) * a.dy_ld + co0 + blk * 64) * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (wp_lds_void_t*)(slot + WK_XB + (ks * 4 + blk) * 4096 + pg * 1024), 16, (int)qrel, so, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) acc[f][fj] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient / column sums of dY: the waves with wi == 0 of the ci_t == 0 blocks, one ones-MFMA per dY fragment and k-step (every row of the result is the sum)
    const bool do_bias = (a.bias_partial != nullptr) && (ci_t == 0) && (wi == 0);
    f32x4 bacc[4];
#pragma unroll
    for (int fj = 0; fj < 4; ++fj) bacc[fj] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones8 = __builtin_bit_cast(bf16x8_t, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});

    issue(e_begin, smem);
    if (total > 1) issue(e_begin + 1, smem + WK_SLOT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_s_sleep(2);                       // (wgrad_pp_stream_kernel: keep the first reads away from the publishing barrier)
    if (grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int slot = 0;
#pragma unroll 1
    for (int e = 0; e < total; ++e) {
        // ================= R segment =================
        const bool more = e + 2 < total;
        if (more) issue(e_begin + e + 2, smem + (slot == 0 ? 2 : slot - 1) * WK_SLOT);      // the slot element e - 1 left (both groups are past its reads)
        else __builtin_amdgcn_s_sleep(1);
        const uint32_t sa = (uint32_t)(uintptr_t)smem + slot * WK_SLOT + wi * 4096;
        const uint32_t sq = (uint32_t)(uintptr_t)smem + slot * WK_SLOT + WK_XB + wj * 4096;
        bf16x8_t A[2][4], B[2][4];
        wp_static_for<2>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            wp_static_for<4>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                A[ks][f] = wp_frag<ks * 8192>(sa + (fo[0] ^ (f << 5)), sa + (fo[1] ^ (f << 5)));
                B[ks][f] = wp_frag<ks * 16384>(sq + (fo[0] ^ (f << 5)), sq + (fo[1] ^ (f << 5)));
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // element e + 1 must have landed before the next step reads it: everything but this step's six instructions
        if (grp == 1) {
            if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ================= M segment =================
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int fj = 0; fj < 4; ++fj) acc[f][fj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks][f], B[ks][fj], acc[f][fj], 0, 0, 0);
        if (do_bias) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int fj = 0; fj < 4; ++fj) bacc[fj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, B[ks][fj], bacc[fj], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) {
            if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        slot = slot == WK_NSLOT - 1 ? 0 : slot + 1;
    }

    float* out = a.partial + (size_t)split * a.Cin * a.Cout;       // TT = 1
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) {
            const int co = co0 + wj * 64 + fj * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + wi * 64 + f * 16 + lg * 4 + r;
                out[(size_t)ci * a.Cout + co] = acc[f][fj][r];
            }
        }
    if (do_bias && lg == 0) {
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) a.bias_partial[(size_t)split * a.Cout + co0 + wj * 64 + fj * 16 + li] = bacc[fj][0];
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------------------------
// the 1x1 form (wgrad1_pp_kernel): 2-D, plain single-source operand, 128-ci x 256-co tiles, whole 64-pixel elements
static bool wgrad1_pp_ok(const MisWgradDesc* d) {
    if (d->dtype != MIS_BF16 || d->ksize != 1 || d->is3d || d->D != 1 || mis_sw(SW_WGRAD_K1_NOPP)) return false;
    if (d->x1 != nullptr || d->in_scale != nullptr || d->dw_per_sample != nullptr) return false;
    if (d->x0_H != d->H || d->x0_W != d->W) return false;
    if (d->Cin % 128 != 0 || d->Cout % 256 != 0) return false;
    const long long P = (long long)d->N * d->H * d->W;
    if (P % 64 != 0 || P / 64 >= (1ll << 30)) return false;
    if ((long long)64 * d->x0_ld * 2 >= (1ll << 31) || (long long)64 * d->dy_ld * 2 >= (1ll << 31)) return false;
    return true;
}

bool wgrad_pp_eligible(const MisWgradDesc* d) {
    if (d->ksize == 1) return wgrad1_pp_ok(d);
    if (d->dtype != MIS_BF16 || d->ksize != 3) return false;
    if (d->is3d) {
        if (mis_sw(SW_WGRAD3D_NOPP) || d->x0_D != d->D) return false;
    } else if (d->D != 1) {
        return false;
    }
    if (d->x1 != nullptr || d->in_scale != nullptr) return false;
    if (d->x0_H != d->H || d->x0_W != d->W) return false;
    if (d->Cin % 64 != 0 || d->Cout % 64 != 0) return false;
    // 32-bit buffer offsets within ONE image / depth plane, computed in (signed) int: below 2 GiB
    if ((((long long)d->H * d->W - 1) * d->x0_ld + d->Cin) * 2 >= (1ll << 31) - 65536) return false;
    if ((((long long)d->H * d->W - 1) * d->dy_ld + d->Cout) * 2 >= (1ll << 31) - 65536) return false;
    if (mis_sw(SW_WGRAD_NO_TR)) return false;
    return true;
}

// kernel choice: 0 = wide (Cout % 128 == 0), 1 = pixel-split wide tile (64-column tiles), 2 = the 64 x 64 kernel with 16 x 32 wave tiles (MIS_WGRAD_PP_NOWIDE=1; 2-D only),
// 3 / 4 = the row variants of 0 / 1 (32-pixel-wide tiles with x-fragment reuse across k-steps): taken when the 32-wide tiles pad the W axis by at most an eighth
// (MIS_WGRAD_PP_ROW=1: always, MIS_WGRAD_PP_NOROW=1: never)
static int wp_kind(const MisWgradDesc* d) {
    if (d->ksize == 1) return 6;                     // wgrad1_pp_kernel (wgrad_pp_eligible has checked the shape)
    if (!d->is3d && mis_sw(SW_WGRAD_PP_NOWIDE)) return 2;
    const int base = d->Cout % 128 == 0 ? 0 : 1;
    const bool fits = ((d->W + 31) / 32) * 32 * 8 <= d->W * 9;
    if (!mis_sw(SW_WGRAD_PP_NOROW) && (fits || mis_sw(SW_WGRAD_PP_ROW))) return base + 3;
    return base;
}

// the streaming forms of kinds 3 / 4 (wgrad_pp_stream_kernel<false / true>): whole 32 x 4 / 32 x 8-pixel tiles; MIS_WGRAD_PP_NOSTREAM=1 keeps the tile-staged row kernels
static bool wp_stream_ok(const MisWgradDesc* d) {
    const int kind = wp_kind(d);
    if (kind != 3 && kind != 4) return false;
    return d->H % (kind == 4 ? 8 : 4) == 0 && d->W % 32 == 0 && !mis_sw(SW_WGRAD_PP_NOSTREAM);
}

static bool wp_per_sample(const MisWgradDesc* d) { return d->dw_per_sample != nullptr && wp_kind(d) != 2; }

// spb: splits per sample when the descriptor asks for per-sample weight gradients (the split ranges then never straddle a sample), else 0
static void wp_plan(const MisWgradDesc* d, int* ntiles, int* tps, int* nsb, int* tilesH, int* tilesW, int* spb = nullptr, int* tpsamp = nullptr) {
    const int kind = wp_kind(d);
    if (kind == 6) {                                 // "tiles" = stream elements of 64 pixels; one persistent block per CU
        const long long nt = (long long)d->N * d->H * d->W / 64;
        const long long npairs = (long long)(d->Cin / 128) * (d->Cout / 256);
        long long want = mis_persist_cus() / npairs;
        if (want < 1) want = 1;
        if (want > nt) want = nt;
        *ntiles = (int)nt;
        *tilesH = 1;
        *tilesW = 1;
        if (spb != nullptr) *spb = 0;
        if (tpsamp != nullptr) *tpsamp = (int)nt;
        *tps = (int)((nt + want - 1) / want);
        *nsb = (int)((nt + *tps - 1) / *tps);
        return;
    }
    const bool wide = kind == 0 || kind == 3;
    const int th = kind == 3 ? 4 : (kind == 0 || kind == 4) ? 8 : 16;
    const int tw = kind >= 3 ? 32 : WP_TW;
    *tilesH = (d->H + th - 1) / th;
    *tilesW = (d->W + tw - 1) / tw;
    const long long per = (long long)(d->is3d ? d->D : 1) * *tilesH * *tilesW;      // tiles per sample (3-D: one 2-D tile grid per depth plane)
    const long long nt = (long long)d->N * per;
    const long long npairs = (long long)(d->Cin / 64) * (d->Cout / (wide ? 128 : 64)) * (d->is3d ? 3 : 1);
    const int P = mis_persist_cus();              // 256 unless MIS_PERSIST_CUS leaves CUs to a concurrent collective (dispatch_cfg.hpp)
    long long want = P / npairs;                  // one persistent block per CU
    if (want < 1) want = 1;
    if (want > nt) want = nt;
    *ntiles = (int)nt;
    if (spb != nullptr) *spb = 0;
    if (tpsamp != nullptr) *tpsamp = (int)per;
    if (wp_per_sample(d)) {
        // splits per sample: the count that minimises (rounds of 256 blocks) / (splits) - e.g. 72 pairs x 2 samples: 3 per sample (432 blocks, two rounds of a third
        // of the tiles each) matches the 3 splits of the batch-wide plan, where 1 per sample would leave 112 CUs idle; ties go to the smaller count (fewer slabs)
        long long k = 1;
        double best = 1e30;
        for (long long c = 1; c <= 4 && c <= per; ++c) {
            const long long blocks = npairs * c * d->N;
            const double cost = (double)((blocks + P - 1) / P) / (double)(c * d->N);
            if (cost < best * 0.999) {
                best = cost;
                k = c;
            }
        }
        if (want / d->N > k) k = want / d->N;     // few pairs: one persistent block per CU, as in the batch-wide plan
        if (k > per) k = per;
        *tps = (int)((per + k - 1) / k);
        k = (per + *tps - 1) / *tps;              // every split non-empty
        *nsb = (int)(k * d->N);
        if (spb != nullptr) *spb = (int)k;
        return;
    }
    *tps = (int)((nt + want - 1) / want);
    *nsb = (int)((nt + *tps - 1) / *tps);
}

int wgrad_pp_nsplit(const MisWgradDesc* d) {
    int ntiles, tps, nsb, th, tw;
    wp_plan(d, &ntiles, &tps, &nsb, &th, &tw);
    const int kind = wp_kind(d);
    return (kind == 1 || kind == 4) ? 2 * nsb : nsb;
}

int wgrad_pp_splits_per_sample(const MisWgradDesc* d) {
    int ntiles, tps, nsb, th, tw, spb = 0;
    wp_plan(d, &ntiles, &tps, &nsb, &th, &tw, &spb);
    return spb;
}

template <bool SPLIT, bool IS3D> static int wp_launch_wide(const WgArgs& a, long long grid, hipStream_t stream, const char* what) {
    const size_t lds = 2 * (size_t)W3<SPLIT>::STAGE;
    static std::atomic<unsigned long long> attr_done{0};
    if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&wgrad_pp_wide_kernel<SPLIT, IS3D>), lds, what)) return rc;
    hipLaunchKernelGGL((wgrad_pp_wide_kernel<SPLIT, IS3D>), dim3((unsigned)grid), dim3(512), lds, stream, a);
    return MIS_OK;
}

template <bool SPLIT, bool IS3D> static int wp_launch_row(const WgArgs& a, long long grid, hipStream_t stream, const char* what) {
    const size_t lds = 2 * (size_t)W4<SPLIT>::STAGE;
    static std::atomic<unsigned long long> attr_done{0};
    if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&wgrad_pp_row_kernel<SPLIT, IS3D>), lds, what)) return rc;
    hipLaunchKernelGGL((wgrad_pp_row_kernel<SPLIT, IS3D>), dim3((unsigned)grid), dim3(512), lds, stream, a);
    return MIS_OK;
}

int launch_wgrad_pp(const MisWgradDesc* d, float* partial, float* bias_partial, hipStream_t stream, const char** tag) {
    const bool is3d = d->is3d != 0;
    WgArgs a;
    a.N = d->N; a.D = is3d ? d->D : 1; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.Cin0 = d->Cin;
    a.x0 = WSrc{d->x0, d->x0_ld, a.D, d->x0_H, d->x0_W};
    a.x1 = WSrc{nullptr, 0, 0, 0, 0};
    a.in_scale = nullptr; a.in_shift = nullptr;
    a.dy = d->dy; a.dy_ld = d->dy_ld; a.partial = partial; a.bias_partial = bias_partial;
    int nsb;
    const int kind = wp_kind(d);
    wp_plan(d, &a.ntiles, &a.tps, &nsb, &a.tilesH, &a.tilesW, &a.spb, &a.tpsamp);
    MIS_REQUIRE((long long)d->N * a.D * a.tilesH * a.tilesW < (1ll << 30), MIS_EUNSUPPORTED, "wgrad(pp): too many pixel tiles");
    if (kind == 6) {
        a.tilesD = 1; a.nsplit = nsb;
        a.nCi = d->Cin / 128; a.nCo = d->Cout / 256; a.KDn = 1; a.TT = 1;
        const long long grid1 = (long long)a.nCi * a.nCo * nsb;
        *tag = "k1.2d.ppg";
        static std::atomic<unsigned long long> attr_done{0};
        const size_t lds = (size_t)WK_NSLOT * WK_SLOT;
        if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&wgrad1_pp_kernel), lds, "wgrad(k1 pp)")) return rc;
        hipLaunchKernelGGL(wgrad1_pp_kernel, dim3((unsigned)grid1), dim3(512), lds, stream, a);
        MIS_LAUNCH_CHECK("wgrad(k1 pp)");
        return MIS_OK;
    }
    const bool wide = kind == 0 || kind == 3;
    a.tilesD = a.D; a.nsplit = (kind == 1 || kind == 4) ? 2 * nsb : nsb;
    a.nCi = d->Cin / 64; a.nCo = d->Cout / (wide ? 128 : 64); a.KDn = is3d ? 3 : 1; a.TT = is3d ? 27 : 9;
    const long long grid = (long long)a.nCi * a.nCo * a.KDn * nsb;
    MIS_REQUIRE(grid < (1ll << 31), MIS_EUNSUPPORTED, "wgrad(pp): grid too large");
    if (kind == 0) {
        *tag = is3d ? "k3.3d.ppw" : "k3.2d.ppw";
        if (const int rc = is3d ? wp_launch_wide<false, true>(a, grid, stream, "wgrad(ppw3)") : wp_launch_wide<false, false>(a, grid, stream, "wgrad(ppw)")) return rc;
    } else if (kind == 1) {
        *tag = is3d ? "k3.3d.pps" : "k3.2d.pps";
        if (const int rc = is3d ? wp_launch_wide<true, true>(a, grid, stream, "wgrad(pps3)") : wp_launch_wide<true, false>(a, grid, stream, "wgrad(pps)")) return rc;
    } else if ((kind == 3 || kind == 4) && wp_stream_ok(d)) {
        *tag = kind == 3 ? (is3d ? "k3.3d.ppst" : "k3.2d.ppst") : (is3d ? "k3.3d.ppss" : "k3.2d.ppss");
        auto go = [&](auto kernel, size_t lds) -> int {
            static std::atomic<unsigned long long> attr_done{0};
            if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(kernel), lds, "wgrad(ppst)")) return rc;
            hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(512), lds, stream, a);
            return MIS_OK;
        };
        int rc;
        if (kind == 3) rc = is3d ? go(&wgrad_pp_stream_kernel<false, true>, (size_t)WS<false>::LDS) : go(&wgrad_pp_stream_kernel<false, false>, (size_t)WS<false>::LDS);
        else rc = is3d ? go(&wgrad_pp_stream_kernel<true, true>, (size_t)WS<true>::LDS) : go(&wgrad_pp_stream_kernel<true, false>, (size_t)WS<true>::LDS);
        if (rc) return rc;
    } else if (kind == 3) {
        *tag = is3d ? "k3.3d.ppwr" : "k3.2d.ppwr";
        if (const int rc = is3d ? wp_launch_row<false, true>(a, grid, stream, "wgrad(ppwr3)") : wp_launch_row<false, false>(a, grid, stream, "wgrad(ppwr)")) return rc;
    } else if (kind == 4) {
        *tag = is3d ? "k3.3d.ppsr" : "k3.2d.ppsr";
        if (const int rc = is3d ? wp_launch_row<true, true>(a, grid, stream, "wgrad(ppsr3)") : wp_launch_row<true, false>(a, grid, stream, "wgrad(ppsr)")) return rc;
    } else {
        *tag = "k3.2d.pp";
        const size_t lds = 2 * (size_t)WP_STAGE;
        const bool kss1 = mis_sw(SW_WGRAD_PP_KSS1) != 0;     // default: two k-steps per segment (36 MFMAs per wave between barriers)
        if (kss1) {
            static std::atomic<unsigned long long> attr_done{0};
            if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&wgrad_pp_kernel<1>), lds, "wgrad(pp)")) return rc;
            hipLaunchKernelGGL((wgrad_pp_kernel<1>), dim3((unsigned)grid), dim3(512), lds, stream, a);
        } else {
            static std::atomic<unsigned long long> attr_done{0};
            if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&wgrad_pp_kernel<2>), lds, "wgrad(pp)")) return rc;
            hipLaunchKernelGGL((wgrad_pp_kernel<2>), dim3((unsigned)grid), dim3(512), lds, stream, a);
        }
    }
    MIS_LAUNCH_CHECK("wgrad(pp)");
    return MIS_OK;
}
