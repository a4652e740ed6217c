// Shared pieces of the ping-pong implicit-GEMM convolution kernels (conv_pp.hip: 2-D 3x3; conv3d_pp.hip: 3x3x3): compile-time loops, the LDS-DMA / LDS-read
// primitives and the tile epilogue.  See conv_pp.hip for the structure they serve.
#pragma once
#include <type_traits>
#include <utility>

#include "conv_args.hpp"
#include "relu_bits.hpp"

typedef __attribute__((address_space(3))) void pp_lds_void_t;
typedef __attribute__((address_space(1))) const void pp_glob_void_t;

namespace {
template <typename F, int... I> __device__ __forceinline__ void pp_static_for_impl(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void pp_static_for(F&& f) { pp_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// One LDS-DMA instruction: 64 lanes x 16 bytes from buffer `r` at per-lane byte offset `voff` to LDS [dst, dst + 1 KiB) (dst wave-uniform).
// buffer_load ... lds: 32-bit offsets (no 64-bit address registers), and an offset at or past num_records reads as ZERO - that is the conv's zero
// padding and the tail of the last halo instruction (PP_OOB is past every buffer this kernel accepts: one image of the input, < 4 GiB - 64 KiB, or the packed weights).
constexpr int PP_OOB = (int)0xFFFFFF00u;
__device__ __forceinline__ void pp_dma16(__amdgpu_buffer_rsrc_t r, int voff, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (pp_lds_void_t*)lds_dst_wave_uniform, 16, voff, 0, 0, 0);
}
// ... with a SCALAR offset on top of the per-lane one (round 5): lets a kernel keep its per-lane offsets tile-invariant in registers and move the tile / chunk origin through
// an SGPR - a DMA issue is then s_add + s_mov m0 + buffer_load instead of ~25 vector instructions.  The range check does not need the scalar part here: lanes that must
// read zeros carry PP_OOB in voff, valid lanes are in range by construction.  (A __device__ function: a kernel template that names the builtin itself may lose its host stub.)
__device__ __forceinline__ void pp_dma16s(__amdgpu_buffer_rsrc_t r, int voff, int soff, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (pp_lds_void_t*)lds_dst_wave_uniform, 16, voff, soff, 0, 0);
}
// ---- dynamic tile queue (round 5; host side: dispatch_cfg.hpp mis_tile_queue) -------------------------------------------------------------------------------
// Static schedule: block b runs tiles xcd_remap(b) + r * grid, r = 0, 1, ... - a block whose CU is held by another kernel (an RCCL all-reduce on the side stream) starts
// when the first blocks END, and the launch takes twice as long (measured with a CU-holding dummy kernel, scripts/hog_probe.sh: +40-63 % with 8 of 256 CUs taken).
// Dynamic schedule: round 0 stays static; XCD x's rounds r >= 1 (the SAME tiles as before: base_x + j + r * grid, j < q_x - the L2 locality of the tile order is kept) are
// handed out by a ticket counter per XCD: ticket k -> (r = 1 + k / q_x, j = k % q_x).  Tickets rise, so does the tile index: the first ticket past the end ends the block,
// and every block draws exactly one such ticket - the counter's last ticket is (tiles of rounds >= 1) + q_x - 1, and whoever draws it stores 0 (nobody draws after it).
// The draw is ONE lane's global_atomic_add issued a whole tile before its value is needed (the next-but-one tile); it travels through the block's LDS mailbox.
struct TileQ {
    unsigned* ctr;      // this XCD's counter (nullptr: static stride)
    int qx, basex;      // blocks of this XCD, its first tile of round 0
    unsigned last;      // the last ticket that will ever be drawn from ctr in this launch
};
__device__ __forceinline__ TileQ tq_init(unsigned* tq, int total, int grid, int bid) {
    TileQ q;
    const int xq = grid >> 3, xr = grid & 7, xcd = bid & 7;
    q.qx = xq + (xcd < xr ? 1 : 0);
    q.basex = (xcd < xr) ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    q.ctr = tq != nullptr ? tq + xcd * 16 : nullptr;
    const int R = total / grid, rem = total - R * grid;          // (total >= grid: the launchers never over-size the grid)
    int tail = rem - q.basex;
    tail = tail < 0 ? 0 : (tail > q.qx ? q.qx : tail);
    q.last = (unsigned)((R - 1) * q.qx + tail + q.qx - 1);
    return q;
}
__device__ __forceinline__ int tq_tile(const TileQ& q, unsigned k, int grid) {
    // a ticket past the launch's last one must never become a tile index.  It cannot be drawn from clean counters; a counter that did NOT start at zero (a launch that
    // never finished, two replays of one captured graph running concurrently) means tiles of this launch go unwritten - that is made LOUD: the word next to
    // the counter takes the ticket, mis_tile_queue_errors() (ops.tile_queue_check: tests, bench.py, smoke, every graph replay check) reports it and raises
    if (k > q.last) {
        __hip_atomic_store(q.ctr + 1, k | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0x7fffffff;
    }
    const int r = (int)(k / (unsigned)q.qx);
    return q.basex + ((int)k - r * q.qx) + (r + 1) * grid;
}
// issue only (ONE lane must be active): tk = 1 on entry, the ticket once the wave has waited for vmcnt to drain past this instruction
__device__ __forceinline__ void tq_draw(unsigned& tk, const unsigned* ctr) {
    unsigned zero = 0u;
    // s_nop 4: the counter address may have just been restored from a spill lane (v_readlane writes the SGPR pair) and a VMEM instruction reading an SGPR the VALU wrote
    // needs five wait states - the hazard recogniser does not look inside an asm block (without it the atomic went out with a stale high dword and faulted)
    asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %0, %2 sc0" : "+v"(tk) : "v"(zero), "s"(ctr) : "memory");
}
// ... the drawn ticket -> the block's mailbox (one LDS word); the last ticket of the launch resets the counter.  Same single lane, after its vmcnt wait.
__device__ __forceinline__ void tq_post(const TileQ& q, unsigned tk, uint32_t mailbox) {
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(mailbox), "v"(tk) : "memory");          // (written before this wave's next barrier, whatever the compiler waits for)
    if (tk == q.last) __hip_atomic_store(q.ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned tq_take(uint32_t mailbox) {          // every wave, a barrier after tq_post
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(mailbox) : "memory");
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

// LDS reads as INLINE ASM: hipcc (ROCm 7.2) may put an s_waitcnt vmcnt(N) in front of an LDS read it cannot tell apart from the destination of an LDS-DMA
// in flight - which then waits for the prefetch that was just issued and serialises the pipeline (seen here as soon as a second kind of LDS-DMA, the 4-byte
// bias fetch, joined the kernel; in wgrad_pp.hip with the transposing-read builtin).  The asm form is invisible to that analysis; in exchange NOTHING waits for the
// result automatically: every use sits behind an explicit s_waitcnt lgkmcnt(0) + sched_barrier (cdna_hip_programming.md §5.4 rule 18, §5.7).
template <int OFF> __device__ __forceinline__ u32x4 pp_lds_read128(uint32_t lds_addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF));
    return r;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pp_make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);   // raw buffer, stride 0, 32-bit data format (gfx9 family)
}
}   // namespace

// ---- epilogue: lane (li, lg) holds, per pixel row pf of its wave, NV = 4*NF consecutive output channels of pixel (h0 + wm*4 + pf, w0 + li) ----
// `bias_lds`: the block's BN bias values of this tile's column tile (zeros without a bias), staged in LDS by DMA: a global bias load here would sit behind the
// previous stores in the in-order vmcnt queue and expose their latency once per tile.  The accumulators are re-armed with zeros.
template <int NF, int PF = 4>
__device__ __forceinline__ void pp_epilogue(const ConvArgs& a, f32x4 (&acc)[NF][PF], uint32_t bias_lds, int n, int h0, int w0, int ncol0, int wm, int wn, int li,
                                            int lg) {
    using T = __bf16;
    constexpr int NV = 4 * NF, WAVE_N = NF * 16, EPC = 8;
    const int colw = ncol0 + wn * WAVE_N;          // wave-uniform first column
    const int col = colw + lg * NV;                // this lane's first column
    const bool to0 = colw < a.Cout0;
    T* ybase = reinterpret_cast<T*>(to0 ? a.y0 : a.y1);
    const int yld = to0 ? a.y0_ld : a.y1_ld;
    const int ymode = to0 ? a.y0_mode : a.y1_mode;
    const int cview = to0 ? a.Cout0 : a.Cout - a.Cout0;
    const int lcol = to0 ? col : col - a.Cout0;
    int ab = 0, cq = 0;
    if (ymode == MIS_OUT_SHUFFLE2) {
        cq = cview >> 2;
        ab = lcol / cq;
    }
    // The ReLU-mask rows are loaded where they are used: one exposed global round trip per pixel row (the masked dgrad form of a layer runs 5-22 % slower than its
    // forward form in isolation).  Two attempts to hide them lost: fetching two rows ahead made hipcc's register scoreboard insert vmcnt waits in front of the first
    // fragment reads of every chunk (serialises the weight prefetch); fetching 4 or 8 rows of a tile ahead of the stores is 3-7 % faster in isolation and 1-3 % SLOWER in
    // the live train step (conv_ppc_kernel, scripts/ab_step.py, one process: 37.76 / 39.01 / 38.31 ms per step for 0 / 4 / 8 rows ahead).
    constexpr int MC = NV / EPC;
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const int y = h0 + wm * PF + pf, x = w0 + li;
        float o[NV];
        {
            u32x4 braw[NF];
            pp_static_for<NF>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                braw[f] = pp_lds_read128<f * 16>(bias_lds + (wn * WAVE_N + lg * NV) * 4);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t u = braw[f][q];
                    o[f * 4 + q] = acc[f][pf][q] + __uint_as_float(u);
                    acc[f][pf][q] = 0.f;
                }
        }
        if (y < a.H && x < a.W) {
            if (a.relu) {
#pragma unroll
                for (int i = 0; i < NV; ++i) o[i] = fmaxf(o[i], 0.f);
            }
            const size_t pix = ((size_t)n * a.H + y) * a.W + x;
            if (a.mask != nullptr) {
                const T* mp = reinterpret_cast<const T*>(a.mask) + pix * a.mask_ld + col;
#pragma unroll
                for (int i = 0; i < MC; ++i) {
                    float mf[EPC];
                    unpack_chunk<T>(*reinterpret_cast<const u32x4*>(mp + i * EPC), mf);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) o[i * EPC + e] = (mf[e] > 0.f) ? o[i * EPC + e] : 0.f;
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
#ifndef PPT_NO_STORE
#pragma unroll
            for (int i = 0; i < NV; i += EPC) *reinterpret_cast<u32x4*>(dst + i) = pack_chunk<T>(o + i);
#else
            if (o[0] == 123.456f) *reinterpret_cast<u32x4*>(dst) = pack_chunk<T>(o);          // diagnostic build: keeps the arithmetic, drops the stores
#endif
        }
    }
}

// ---- epilogue of the column-segment kernels (plain NHWC destinations only): the same arithmetic as pp_epilogue - accumulator + bias, ReLU, ReLU mask of another tensor,
// round to bf16 - in a third of the instructions.  Measured on conv_ppc_kernel<8, 4> (scripts/ppt_ablate.sh, PPT_NO_EPI): the epilogue's VECTOR instructions, not its stores, were
// the per-tile overhead that made the short-K layers slow (128->128 at 256^2: 0.600 ms with, 0.465 ms without the epilogue; removing only the stores changed nothing) - every
// instruction of an R slot costs ~10 cycles next to the partner's MFMA cluster, and hipcc's code for pp_epilogue is ~110 of them per pixel row (one v_cvt per VALUE plus shift / or,
// canonicalising v_max pairs, 64-bit address arithmetic per row).  Here, per pixel row of NV values: NV/2 v_pk_add_f32, NV/2 v_cvt_pk_bf16_f32 (two values each), NV/2
// v_pk_max_i16 (ReLU on the PACKED bf16 pair: as int16 a negative float is a negative integer; the lower bound is 0x8000 = "no ReLU" otherwise), and buffer stores whose address is a
// per-lane offset computed once + a scalar row offset; the ReLU mask is applied to the packed pair too (min(m, 1) -> max(., 0) -> multiply, all 16-bit packed).  The bias comes
// out of LDS once per tile.  NaNs are not preserved through the packed ReLU (a negative-signed NaN becomes 0), as with fmaxf.
typedef float pp_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pp_bf16x2 __attribute__((ext_vector_type(2)));
typedef short pp_i16x2 __attribute__((ext_vector_type(2)));

// ReLU bits (relu_bits.hpp) of a lane's part of a tile, 8-row wave tiles only: NF = 4: the 16 bytes [lg][0..1][row 0..7] of record (n, row block, x, 64-channel block);
// NF = 2 (32-channel wave slice): the 8 bytes [lg][half][0..7].  Returns the per-lane byte offset into the bit tensor (32-bit: the launchers check its size), or PP_OOB.
template <int NF, int PF>
__device__ __forceinline__ int pp_bits_voff(const ConvArgs& a, int n, int h0, int w0, int ncol0, int wm, int wn, int li, int lg) {
    const int colw = ncol0 + wn * NF * 16;
    const int x = w0 + li, yb = (h0 + wm * PF) >> 3, H8 = (a.H + 7) >> 3;
    const unsigned rec = (unsigned)(((n * H8 + yb) * a.W + x) * (a.Cout >> 6) + (colw >> 6));
    return (x < a.W && yb < H8) ? (int)(rec * 64u + (unsigned)(lg * 16 + (NF == 2 ? ((colw >> 5) & 1) * 8 : 0))) : PP_OOB;
}

// EM (compile time: each kernel instantiation carries ONE mask path - with all of them in one body the 256-VGPR kernels spilled, scalars first):
//   PP_EM_NONE no mask (the forward form; writes a.relu_bits when asked to), PP_EM_MASK a.mask (bf16 tensor), PP_EM_BITS a.mask_bits (ReLU bits)
//   PP_EM_GN (conv3d_pp.hip): out = [a.gn_relu: (x > 0) *] (p * acc + q * x + r) with x = a.mask and p / q / r per (sample, column) staged in LDS at `pqr_lds`
//   ([3][BN] floats: p, q, r of the tile's sample and column tile) - the GroupNorm backward of a 'gcr' SingleConv continued from its dgrad (MisConvDesc.gn_p)
constexpr int PP_EM_NONE = 0, PP_EM_MASK = 1, PP_EM_BITS = 2, PP_EM_GN = 3;
// this lane's ReLU-bits bytes of tile (n, h0, w0, ncol0) (see pp_bits_voff): the kernels issue this a K chunk before the tile's epilogue, which takes the result as `mb`
template <int NF, int PF>
__device__ __forceinline__ u32x4 pp_mask_bits_load(const ConvArgs& a, int n, int h0, int w0, int ncol0, int wm, int wn) {
    int lane_;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
    const int bvoff = pp_bits_voff<NF, PF>(a, n, h0, w0, ncol0, wm, wn, lane_ & 15, lane_ >> 4);
    const __amdgpu_buffer_rsrc_t rbm = pp_make_rsrc(a.mask_bits, (unsigned)rb_bytes(a.N, a.H, a.W, a.Cout));
    u32x4 mb = u32x4{0u, 0u, 0u, 0u};
    if constexpr (NF == 4) {
        mb = __builtin_amdgcn_raw_buffer_load_b128(rbm, bvoff, 0, 0);
    } else {
        const u32x2 m2 = __builtin_amdgcn_raw_buffer_load_b64(rbm, bvoff, 0, 0);
        mb[0] = m2[0];
        mb[1] = m2[1];
    }
    return mb;
}

// BINIT (conv_ppc_kernel): the bias went into the accumulators as the C operand of the tile's first MFMAs and the next tile's first MFMAs overwrite them - the epilogue
// neither adds the bias nor re-arms the accumulators (192 of its ~330 vector instructions per tile and wave).
template <int NF, int PF, int EM, bool RB = true, bool BINIT = false>
__device__ __forceinline__ void pp_epilogue_plain(const ConvArgs& a, f32x4 (&acc)[NF][PF], uint32_t bias_lds, int n, int h0, int w0, int ncol0, int wm, int wn,
                                                  const u32x4& mb_pre, uint32_t pqr_lds = 0u, int pqr_stride = 0) {
    constexpr int NV = 4 * NF, WAVE_N = NF * 16, NS = NF / 2;          // NS 16-byte stores per pixel
    constexpr bool gn = EM == PP_EM_GN;
    if constexpr (gn) {
        // padding columns of the operand (a.Cout0 < a.Cout without a second destination): nothing to compute or store for this wave (wave-uniform)
        if (ncol0 + wn * WAVE_N >= a.Cout0 && a.y1 == nullptr) {
            if constexpr (!BINIT) {
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int pf = 0; pf < PF; ++pf) acc[f][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            return;
        }
    }
    constexpr bool BITS = RB && PF == 8 && (NF == 4 || NF == 2);        // ReLU bits: the 2-D instantiations (8-row wave tiles); RB = false: the 3-D kernels
    static_assert(EM != PP_EM_BITS || BITS, "");
    static_assert(NF % 2 == 0, "");
    // lane coordinates re-derived here (volatile: not hoisted), so that nothing per-lane of the epilogue is held - or, at 256 VGPRs, spilled - across the tile loop
    int lane_;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));
    const int li = lane_ & 15, lg = lane_ >> 4;
    const int colw = ncol0 + wn * WAVE_N;          // wave-uniform first column
    const bool to0 = colw < a.Cout0;
    const char* const ybase = reinterpret_cast<const char*>(to0 ? a.y0 : a.y1);
    const int yld = to0 ? a.y0_ld : a.y1_ld;
    const int cview = to0 ? a.Cout0 : a.Cout - a.Cout0;
    const int lcolw = to0 ? colw : colw - a.Cout0;
    u32x4 braw[NF];
    if constexpr (!BINIT) {
        pp_static_for<NF>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            braw[f] = pp_lds_read128<(f >> 1) * 128 + (f & 1) * 16>(bias_lds + (wn * WAVE_N + lg * 8) * 4);
        });
    }
    // one buffer resource per image (plane): 32-bit offsets span one image of the destination / the mask (the launchers check < 4 GiB - 64 KiB).  A pixel-UNSHUFFLED
    // destination (MIS_OUT_UNSHUFFLE2: pixel (y, x) -> pixel (y/2, x/2) of a half-resolution image, channel block (y&1)*2 + (x&1)) is the same store with other constants:
    // the x part goes into the per-lane offset, the y part into the scalar row offset
    const size_t img = (size_t)a.H * a.W;
    const int ymode = to0 ? a.y0_mode : a.y1_mode;                              // wave-uniform
    const bool uns = ymode == MIS_OUT_UNSHUFFLE2;
    // a pixel-SHUFFLED destination (MIS_OUT_SHUFFLE2, the transposed convolution's forward GEMM: column ab*Cq + c of pixel (y, x) -> channel c of pixel (2y + a, 2x + b) of an
    // image of twice the size; Cq % 64 == 0, so a wave's 64 columns share one (a, b)): again the same store with other constants
    const bool shf = ymode == MIS_OUT_SHUFFLE2;
    const int cqs = cview >> 2;
    const int abq = shf ? lcolw / cqs : 0;
    const int lcols = shf ? lcolw - abq * cqs : lcolw;
    const int ow = a.W >> 1;
    const size_t yimg = uns ? (size_t)(a.H >> 1) * ow : (shf ? 4 * img : img);
    const __amdgpu_buffer_rsrc_t ry = pp_make_rsrc(ybase + (size_t)n * yimg * yld * 2, (unsigned)(((yimg - 1) * yld + (uns ? 4 * cview : (shf ? cqs : cview))) * 2));
    const int x = w0 + li;
    const int xpart = uns ? (x >> 1) * yld + (x & 1) * cview : (shf ? (2 * x + (abq & 1)) * yld : x * yld);
    const int yv = x < a.W ? (xpart + lcols + lg * 8) * 2 : PP_OOB;          // a store past num_records is dropped: the ragged right edge
    constexpr bool masked = EM == PP_EM_MASK || gn;          // a row of a.mask is loaded per pixel row (the ReLU mask, or the GroupNorm's input x)
    const int mcols = (gn && a.y1 == nullptr) ? a.Cout0 : a.Cout;      // channels of the tensor behind a.mask
    const __amdgpu_buffer_rsrc_t rm = pp_make_rsrc(reinterpret_cast<const char*>(masked ? a.mask : a.y0) + (masked ? (size_t)n * img * a.mask_ld * 2 : 0),
                                                   masked ? (unsigned)(((img - 1) * a.mask_ld + mcols) * 2) : 0u);      // (dead unless a.mask is read)
    const int mv = x < a.W ? (x * a.mask_ld + colw + lg * 8) * 2 : PP_OOB;
    const unsigned yrow = (unsigned)(uns ? ow : (shf ? 2 * a.W : a.W)) * yld * 2, mrow = (unsigned)a.W * a.mask_ld * 2;
    const unsigned yodd = uns ? (unsigned)cview * 4u : 0u;                    // byte offset of the odd rows' channel blocks
    const int ush = uns ? 1 : 0;                                              // row y -> row y >> ush of the destination (+ yodd for odd y when unshuffling)
    const int ysh = shf ? 1 : 0, yadd = shf ? (abq >> 1) : 0;                 // ... or row 2y + a when shuffling
    const int yr0 = h0 + wm * PF;
    const uint32_t lowb = a.relu ? 0u : 0x80008000u;       // lower bound of the packed ReLU; a SCALAR operand of the asm below (no register held across the tile loop)
    // ReLU bits instead of the bf16 mask: this lane's rows of the tile are ONE load, issued by the kernel a K chunk ago (NF = 4: 16 bytes, byte i*8 + r = row r of store
    // piece i; NF = 2: 8 bytes)
    const u32x4 mb = mb_pre;
    u32x4 ob = u32x4{0u, 0u, 0u, 0u};                       // ... and the ones this tile produces, assembled row by row
    constexpr bool bmask = EM == PP_EM_BITS;
    const bool bout = BITS && EM == PP_EM_NONE && a.relu_bits != nullptr;
    int bvoff = 0;
    if constexpr (BITS) {
        if (bout) bvoff = pp_bits_voff<NF, PF>(a, n, h0, w0, ncol0, wm, wn, li, lg);
    }
    pp_f32x2 bias2[NF][2];
    if constexpr (!BINIT) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t u0 = braw[f][2 * h], u1 = braw[f][2 * h + 1];
                bias2[f][h] = pp_f32x2{__uint_as_float(u0), __uint_as_float(u1)};
            }
    }
    // PP_EM_GN: p, q, r of this lane's NV columns (pairs, in the order of the accumulator pairs) out of LDS, once per tile
    pp_f32x2 gp2[gn ? NF : 1][2], gq2[gn ? NF : 1][2], gr2[gn ? NF : 1][2];
    if constexpr (gn) {
        u32x4 praw[3][NF];
        pp_static_for<NF>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            const uint32_t ad = pqr_lds + (wn * WAVE_N + lg * 8) * 4;
            praw[0][f] = pp_lds_read128<(f >> 1) * 128 + (f & 1) * 16>(ad);
            praw[1][f] = pp_lds_read128<(f >> 1) * 128 + (f & 1) * 16>(ad + pqr_stride);
            praw[2][f] = pp_lds_read128<(f >> 1) * 128 + (f & 1) * 16>(ad + 2 * pqr_stride);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                gp2[f][h] = pp_f32x2{__uint_as_float(praw[0][f][2 * h]), __uint_as_float(praw[0][f][2 * h + 1])};
                gq2[f][h] = pp_f32x2{__uint_as_float(praw[1][f][2 * h]), __uint_as_float(praw[1][f][2 * h + 1])};
                gr2[f][h] = pp_f32x2{__uint_as_float(praw[2][f][2 * h]), __uint_as_float(praw[2][f][2 * h + 1])};
            }
    }
    // pixel rows whose mask loads are in flight together (the 2-D net masks with ReLU bits).  Round 6, PP_EM_GN below 256 VGPRs: ALL rows of the wave tile at once (the
    // 64-column 3-D dgrads: 10 or 8 loads, 32-40 registers that the dead fragments leave free) - the epilogue then waits out ONE memory latency per tile instead of
    // PF / MG of them (PMC r05: conv3d_ppc_kernel<10, 2, 3> 48 % pipe busy, 47.7 % of its wave cycles waiting, against 58.9 % for the same kernel without the x rows)
    constexpr int MG0 = PF % 4 == 0 ? 2 : (PF % 5 == 0 ? 5 : (PF % 2 == 0 ? 2 : 1));
#ifdef PP_GN_MG0          // (A/B build: the round-5 grouping)
    constexpr int MG = MG0;
#else
    constexpr int MG = (gn && PF * NS <= 10 && NF * PF < 32) ? PF : MG0;
#endif
#pragma unroll
    for (int pg = 0; pg < PF; pg += MG) {
        u32x4 mk[masked ? MG : 1][NS];
        if constexpr (masked) {
#pragma unroll
            for (int r = 0; r < MG; ++r)
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    mk[r][i] = __builtin_amdgcn_raw_buffer_load_b128(rm, mv + i * 64, __builtin_amdgcn_readfirstlane((int)((unsigned)(yr0 + pg + r) * mrow)), 0);      // rows past H: offset past num_records reads 0
        }
#pragma unroll
        for (int r = 0; r < MG; ++r) {
            const int pf = pg + r;
            const int y = yr0 + pf;                // wave-uniform
            u32x4 d[NS];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    pp_f32x2 s2 = pp_f32x2{acc[f][pf][2 * h], acc[f][pf][2 * h + 1]};
                    if constexpr (!BINIT) s2 += bias2[f][h];
                    if constexpr (gn) {
                        // v = fma(p, acc, fma(q, x, r)): mis_gn_bwd_apply's expression on the fp32 accumulator (that pass reads a bf16-rounded dL/dxn instead)
                        const uint32_t xw = mk[r][f / 2][(f & 1) * 2 + h];
                        const pp_f32x2 x2 = pp_f32x2{__uint_as_float(xw << 16), __uint_as_float(xw & 0xffff0000u)};
                        s2 = __builtin_elementwise_fma(gp2[f][h], s2, __builtin_elementwise_fma(gq2[f][h], x2, gr2[f][h]));
                    }
                    uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(s2, pp_bf16x2));
                    asm("v_pk_max_i16 %0, %1, %2" : "=v"(pk) : "v"(pk), "s"(lowb));
                    d[f / 2][(f & 1) * 2 + h] = pk;
                }
#ifndef PPT_EPI_NOZERO
                if constexpr (!BINIT) acc[f][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
            }
            if constexpr (masked) if (!gn || a.gn_relu) {
#pragma unroll
                for (int i = 0; i < NS; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t t = mk[r][i][q], du = d[i][q];          // per 16-bit half: 1 where the mask value is a positive float, else 0 (hipcc's own lowering of this is
                        asm("v_pk_min_i16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_max_i16 %0, %0, 0 op_sel_hi:[1,0]\n\tv_pk_mul_lo_u16 %0, %0, %2" : "=&v"(t) : "v"(t), "v"(du));      // compare + select + permute)
                        d[i][q] = t;
                    }
            }
            if constexpr (BITS) {
                if constexpr (bmask) {
                    // dword q of piece i holds channels 2q, 2q + 1 of the piece: their bits are bits 2q, 2q + 1 of byte i*8 + pf.  v_pk_lshrrev_b16 shifts the byte's
                    // 16-bit half by (s, s + 1) into the two result halves; & 0x00010001 leaves the multipliers (1 or 0 per half)
#pragma unroll
                    for (int i = 0; i < NS; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const uint32_t mw = mb[i * 2 + (pf >> 2)], du = d[i][q];
                            const uint32_t sh = (uint32_t)((pf & 1) * 8 + 2 * q) * 0x00010001u + 0x00010000u;
                            uint32_t t;
                            if ((pf & 3) < 2) asm("v_pk_lshrrev_b16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "s"(sh), "v"(mw));
                            else asm("v_pk_lshrrev_b16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(t) : "s"(sh), "v"(mw));
                            t &= 0x00010001u;
                            asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(t) : "v"(du), "v"(t));
                            d[i][q] = t;
                        }
                }
                if (bout) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        uint32_t t[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(t[q]) : "v"(d[i][q]));      // per half: 1 if the stored value is > 0 (it is >= 0)
                        uint32_t m = t[0] | (t[1] << 2) | (t[2] << 4) | (t[3] << 6);      // even channels at bits 0,2,4,6; odd ones at 16,18,20,22
                        m |= m >> 15;
                        const uint32_t prev = ob[i * 2 + (pf >> 2)];
                        ob[i * 2 + (pf >> 2)] = (pf & 3) == 0 ? (m & 0xffu) : (prev | ((m & 0xffu) << ((pf & 3) * 8)));
                    }
                }
            }
#ifdef PPT_EPI_NOSTORE
            if (y < a.H && d[0][0] == 0x12345678u) {
#else
            if (y < a.H) {
#endif
                // (readfirstlane: the offset is wave-uniform, but hipcc evaluates the selects on the VALU and would wrap every store in a waterfall loop)
                const int srow = __builtin_amdgcn_readfirstlane((int)((unsigned)(((y >> ush) << ysh) + yadd) * yrow + (unsigned)(y & ush) * yodd));
#pragma unroll
                for (int i = 0; i < NS; ++i) __builtin_amdgcn_raw_buffer_store_b128(d[i], ry, yv + i * 64, srow, 0);
            }
        }
    }
    if constexpr (BITS) {
        if (bout) {          // rows past H of the last row block carry garbage bits: never read
            const __amdgpu_buffer_rsrc_t rbo = pp_make_rsrc(a.relu_bits, (unsigned)rb_bytes(a.N, a.H, a.W, a.Cout));
            if constexpr (NF == 4) __builtin_amdgcn_raw_buffer_store_b128(ob, rbo, bvoff, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b64(u32x2{ob[0], ob[1]}, rbo, bvoff, 0, 0);
        }
    }
}
