/* misamd.h - C ABI of the MI355X-native U-Net hot path (libmisamd.so).
 *
 * The reference (a-green-hand-jack/mdeical_image_segmentation) is pure Python: it has no FFI
 * of its own.  Every entry point below replaces one or more ATen ops that the reference's
 * nn.Modules invoke on the hot path; the reference call site is cited per function.  The
 * Python host side (mdeical_image_segmentation_amd/) binds these with ctypes, passing raw
 * device pointers (torch tensors are only the allocator) and the current HIP stream.
 *
 * Conventions
 *   - activations are channels-last: NHWC (2-D) / NDHWC (3-D); a tensor view is
 *     (pointer to channel 0 of pixel 0, ld = elements between consecutive pixels, C)
 *     so that producers can write straight into a channel slice of a concat buffer;
 *   - dtype: MIS_F32 (exact f32 MFMA, the parity mode) or MIS_BF16 (bf16 storage, f32 accumulate);
 *   - every function enqueues on `stream` and never synchronises or allocates;
 *   - return value 0 = ok, negative = error; mis_last_error() gives the thread-local message.
 */
#ifndef MISAMD_H
#define MISAMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIS_F32 0
#define MIS_BF16 1

#define MIS_OK 0
#define MIS_EINVAL (-1)
#define MIS_EUNSUPPORTED (-2)
#define MIS_EHIP (-3)

/* epilogue store modes of mis_conv_igemm */
#define MIS_OUT_PLAIN 0      /* y[pixel][c]                                                    */
#define MIS_OUT_SHUFFLE2 1   /* GEMM column (ab*Cq + c) of pixel (h,w) -> pixel (2h+a, 2w+b), channel c */
#define MIS_OUT_UNSHUFFLE2 2 /* channel c of pixel (h,w) -> pixel (h/2,w/2), channel ((h&1)*2+(w&1))*C + c */

const char* mis_last_error(void);
int mis_version(void);

/* Layout of the structs of this header AS THE LOADED BUILD SEES THEM (round 5): callers that mirror them by hand in another language (the ctypes / numpy mirrors of
 * mdeical_image_segmentation_amd/_lib.py and ops.py) compare sizeof, field names, offsets and sizes against this at test time - a field added on one side only is otherwise
 * a silent wild pointer on the device.  mis_abi_layout fills up to max_fields entries of names / offsets / sizes (any of them may be NULL) and *size = sizeof(struct);
 * returns the struct's number of fields (in declaration order) or MIS_EINVAL for an unknown name.  mis_abi_struct_name(i), i < mis_abi_struct_count(): the covered structs. */
int mis_abi_struct_count(void);
const char* mis_abi_struct_name(int index);
int mis_abi_layout(const char* struct_name, size_t* size, const char** names, size_t* offsets, size_t* sizes, int max_fields);

/* Implicit-GEMM convolution on MFMA: rows = output pixels, K = taps x Cin, columns = Cout.
 * Replaces nn.Conv2d(k3,p1) fwd and its dgrad (model/unet2d/layers.py:122,125), nn.Conv3d(k3,p1)
 * (model/unet3d/buildingblocks.py:64-66), ConvTranspose2d(k2,s2) fwd/dgrad as a 1x1 GEMM with a
 * pixel-shuffle store (model/unet2d/layers.py:165) and the channel concat / nearest upsample
 * (layers.py:186-190, buildingblocks.py:546-548,671-673) which become addressing. */
typedef struct MisConvDesc {
    int dtype;               /* MIS_F32 | MIS_BF16 */
    int ksize;               /* 3 (3x3 or 3x3x3, pad 1, stride 1) or 1 */
    int N, D, H, W;          /* pixel grid walked by the GEMM rows (D = 1 for 2-D) */
    int is3d;                /* 1: volumetric op (3x3x3 taps for ksize 3) even when D == 1; 0: 2-D (D must be 1) */
    int Cin, Cout;           /* per-tap K and GEMM N; both multiples of the K chunk (64 bf16 / 32 f32) resp. 64 */
    /* input: channels [0,Cin0) from x0, [Cin0,Cin) from x1 (x1 may be NULL when Cin0 == Cin).
     * A source whose grid is exactly half of (D,H,W) along an axis is read with nearest addressing src = dst >> 1
     * (F.interpolate(mode='nearest') of the reference for the 2x case; other ratios are MIS_EUNSUPPORTED). */
    const void* x0; int x0_ld; int x0_D, x0_H, x0_W;
    const void* x1; int x1_ld; int x1_D, x1_H, x1_W;
    int Cin0;
    /* optional per-(n, channel) affine applied to in-bounds input while staging (GroupNorm folded
     * to a*x+b, buildingblocks.py:87-92): in_scale[n*Cin + c], in_shift[n*Cin + c]; NULL = none */
    const float* in_scale; const float* in_shift;
    const void* w;           /* packed [tap][Cout][Cin] in dtype (mis_pack_conv_weight) */
    const float* bias;       /* [Cout] or NULL; with MIS_OUT_SHUFFLE2 indexed by c (= col % Cq) */
    int relu;                /* epilogue max(x,0) */
    const void* mask; int mask_ld; /* optional: out *= (mask[pixel][col] > 0), same grid as the rows (ReLU backward) */
    /* outputs: columns [0,Cout0) -> y0, [Cout0,Cout) -> y1 (y1 may be NULL when Cout0 == Cout) */
    void* y0; int y0_ld; int y0_mode;
    void* y1; int y1_ld; int y1_mode;
    int Cout0;
    /* ReLU bits (bf16, 2-D, Cout % 64 == 0; layout: csrc/relu_bits.hpp, size mis_relu_bits_bytes(N, H, W, Cout)): one bit per output element.
     * relu_bits (out, with relu = 1, one plain destination): bit = (stored output > 0), written from the epilogue - the mask the ReLU backward of this layer needs.
     * mask_bits (in, instead of `mask`): out *= bit of a tensor of the output's shape - the same result as `mask` with the bf16 tensor the bits were taken from,
     * at 1/16 of its traffic.  Both NULL by default. */
    void* relu_bits;
    const void* mask_bits;
    /* GroupNorm backward in the epilogue (round 4; bf16 3x3x3 on the ping-pong kernels, single source): with gn_p != NULL the launch is the dgrad of a 'gcr' SingleConv
     * CONTINUED through the GroupNorm in front of the convolution and the ReLU that produced its input (model/unet3d/buildingblocks.py:87-92):
     *     out = [gn_relu: (x > 0) *] ( gn_p[n][col] * acc + gn_q[n][col] * x + gn_r[n][col] ),     x = mask[pixel][col]  (`mask` / `mask_ld`: the tensor the GroupNorm read)
     * with p / q / r the [N][gn_ld] fp32 tables of mis_gn_bwd_finalize - dL/d(normalised operand) is never written and mis_gn_bwd_apply's pass (3 tensors) disappears.
     * Columns [Cout0, Cout) are DROPPED when y1 == NULL (padding channels of the operand: here Cout0 may be any multiple of 32). */
    const float* gn_p; const float* gn_q; const float* gn_r; int gn_ld; int gn_relu;
    /* Per-channel sums of the output in the epilogue (round 6; fp32 3x3x3 on conv3d_f32.hip - mis_conv_stats_rows(d) > 0 says whether the launch this descriptor would
     * take carries them): st_mode 1 = the two reductions of the GroupNorm backward behind a dgrad (model/unet3d/buildingblocks.py:87-92), S1 = sum out, S2 = sum out * x with
     * x the tensor that GroupNorm read - columns [0, st_c0) from st_x0, [st_c0, Cout) from st_x1 (st_up: on the half grid, read at >> 1; st_x1 NULL: all from st_x0) -
     * instead of mis_gn_bwd_stats' pass over both tensors; st_mode 2 = S1 = sum out, S2 = sum out^2 of the stored (post-ReLU) output, the statistics the NEXT GroupNorm needs
     * (mis_chanstats' pass).  The kernel writes one partial row per (spatial tile, row half): st_part[n][row][S1 columns | S2 columns] with
     * mis_conv_stats_rows(d) rows per sample and 2 * Cout floats per row; the caller reduces the rows with mis_conv_stats_reduce (double precision, fixed order). 0: off. */
    int st_mode;
    const void* st_x0; int st_x0_ld;
    const void* st_x1; int st_x1_ld;
    int st_c0; int st_up;
    float* st_part;
} MisConvDesc;
int mis_conv_igemm(const MisConvDesc* d, void* stream);
/* rows per sample of MisConvDesc.st_part for this descriptor, or 0 when the kernel it dispatches to has no statistics epilogue (st_mode is then refused) */
long long mis_conv_stats_rows(const MisConvDesc* d);
/* st_part (N x rows x 2 * Cout floats, as the launch left it) -> S1, S2 [N][Cout]: rows added in double precision, fixed order */
size_t mis_conv_stats_reduce_workspace_bytes(int N, int Cout);
int mis_conv_stats_reduce(const float* part, int N, long long rows, int Cout, void* workspace, float* S1, float* S2, void* stream);
/* Name of the kernel configuration the calling thread's last mis_conv_igemm ran, e.g. "k3.2d.bn256.dma" (diagnostic: the parity tests assert
 * that each case reaches the dispatch branch it is written for). */
const char* mis_conv_last_dispatch(void);
/* Kernel-selection switches (A/B arms and fall-back configurations of mis_conv_igemm / mis_wgrad and a few others, e.g. "MIS_CONV_NOPP", "MIS_WGRAD_NOPP",
 * "MIS_CONV3D_PF"): each is read from the environment variable of the same name ONCE per process; this entry point overrides one of them at run time
 * (value >= 0), or returns it to the environment's value / its default (value < 0); name == NULL resets all.  mis_dispatch_switch returns the current value.
 * Unknown names: MIS_EINVAL.  This is how the parity tests reach every kernel configuration in one process; the launch path itself never calls getenv. */
int mis_dispatch_override(const char* name, int value);
int mis_dispatch_switch(const char* name);
/* Diagnostic: the eight per-XCD ticket counters of the persistent kernels' tile queue on `stream` (csrc/dispatch_cfg.hpp), after a device synchronisation -
 * all zero between launches (the kernel that draws a counter's last ticket resets it).  Returns -1 when the stream has no counter block (MIS_TILEQ_OFF=1). */
int mis_debug_tile_queue(void* stream, unsigned* out8);
/* Diagnostic: plant `value` in counter `xcd` (0..7) of `stream`'s block - the state an unfinished launch would leave (tests: the next launch must report it). -1: no block. */
int mis_debug_tile_queue_poke(void* stream, int xcd, unsigned value);
/* Tile queue housekeeping (round 6).  mis_tile_queue_init: allocate and zero the current device's counter pool now (library load time) instead of inside the first
 * launch - nothing on the launch path allocates or synchronises, and a launch captured before any eager one still finds the pool.  mis_tile_queue_reset: zero the counters
 * of `stream` (or of the capture running on it: a memset node) - the engines call it at the start of every train step.  mis_tile_queue_errors: device-synchronising check -
 * the number of counters that handed out a ticket past their launch's last one since the previous call (such a launch left output tiles unwritten; cannot happen on
 * clean counters), 0 = clean; the message is in mis_last_error(). */
int mis_tile_queue_init(void);
int mis_tile_queue_reset(void* stream);
int mis_tile_queue_errors(void);
/* 1 when the library was built with `make EXPERIMENTS=1` (csrc/experiments: kernel variants that lost their A/B, selectable through MIS_CONV_PPS / MIS_CONV_PPC2) */
int mis_build_has_experiments(void);

/* Weight-gradient GEMM: dW[tap][ci][co] = sum_pixels x[pixel+tap][ci] * dy[pixel][co]   (split-K over pixel tiles,
 * fp32 partial slabs + deterministic reduction).  Replaces the weight part of convolution_backward for
 * Conv2d/Conv3d(k3,p1) and ConvTranspose2d(k2,s2) (as ksize 1 on the un-shuffled gradient). */
typedef struct MisWgradDesc {
    int dtype;
    int ksize;               /* 3 or 1 */
    int N, D, H, W;
    int is3d;                /* as in MisConvDesc */
    int Cin, Cout;           /* x channels, dy channels */
    const void* x0; int x0_ld; int x0_D, x0_H, x0_W;
    const void* x1; int x1_ld; int x1_D, x1_H, x1_W;
    int Cin0;
    const float* in_scale; const float* in_shift;
    const void* dy; int dy_ld;
    float* workspace; size_t workspace_bytes;   /* >= mis_wgrad_workspace_bytes(d) */
    float* dw;               /* output, fp32 */
    int dw_layout;           /* 0: [Cout][Cin][taps] (nn.Conv weight), 1: [Cin][Cq][4] with dy column = ab*Cq + c (nn.ConvTranspose2d k2) */
    float alpha;             /* dw = alpha * sum */
    float* dbias;            /* optional: bias gradient = alpha * column sums of dy ([Cout], or [Cout/4] folded over (a,b) for layout 1) */
    void* reduce_stream;     /* optional second HIP stream: the slab reduction (HBM-bound) is enqueued there, ordered after the MFMA kernel by
                              * an event, so that it runs under the caller's next kernel on `stream`.  The caller joins reduce_stream before
                              * reading dw / dbias and before reusing the workspace. NULL: everything on `stream`. */
    float* dw_per_sample;    /* optional [N][Cout][Cin][taps] (layout 0, bf16 3x3 / 3x3x3 ping-pong path only): the weight gradient of every SAMPLE (dw = their sum).  The
                              * split-K ranges then never straddle a sample.  Why: a GroupNorm in front of the convolution needs sum_v dyn * x per (sample, channel) for
                              * its backward - which equals sum_{co,tap} W[co][c][tap] * dw_n[co][c][tap] once x is expressed through the normalised operand
                              * (mis_gn_bwd_stats_from_dw): the pass over dyn and x that mis_gn_bwd_stats makes is not needed. */
    float* dbias_per_sample; /* optional [N][Cout]: column sums of dy per sample (with dw_per_sample) */
    struct MisWgradReduceItem* defer;   /* optional: run the MFMA kernel only and describe the slab reduction that is left in *defer (host memory); the caller then reduces a
                              * GROUP of layers with one mis_wgrad_reduce_batch call - two launches per group instead of three small kernels per layer.  The workspace must
                              * stay untouched until that call; not with dw_per_sample / reduce_stream. */
} MisWgradDesc;
/* What a deferred mis_wgrad left to do: sum `nsplit` fp32 slabs [TT][Cin][Cout] at `partial` in a fixed order into dw (layout as MisWgradDesc.dw_layout, times alpha) and,
 * if dbias != NULL, the nsplit rows of column sums at bias_partial into dbias. */
typedef struct MisWgradReduceItem {
    float* partial; float* dw; const float* bias_partial; float* dbias;
    int nsplit, TT, Cin, Cout, dw_layout; float alpha;
} MisWgradReduceItem;
size_t mis_wgrad_workspace_bytes(const MisWgradDesc* d);
int mis_wgrad(const MisWgradDesc* d, void* stream);
/* Reduce the slabs of n <= 16 deferred weight gradients (host array `items`): same arithmetic and summation order per layer as the undeferred call. */
int mis_wgrad_reduce_batch(const MisWgradReduceItem* items, int n, void* stream);
/* Diagnostic twins of mis_conv_last_dispatch for mis_wgrad: configuration name and split-K factor of the calling thread's last call. */
const char* mis_wgrad_last_dispatch(void);
int mis_wgrad_last_nsplit(void);

/* First layer (Cin = 1..4, fp32 NCHW image in, NHWC out): direct conv, forward and weight/bias gradient.
 * model/unet2d/layers.py:122 for down_conv.0.first. */
int mis_conv3x3_first_fwd(int dtype, const float* x_nchw, int N, int Cin, int H, int W, const float* w /*[64][Cin][3][3]*/,
                          const float* bias, void* y, int y_ld, int Cout, void* stream);
/* ... the same with the ReLU bits of the output written from the epilogue (bf16; relu_bits may be NULL) */
int mis_conv3x3_first_fwd_rb(int dtype, const float* x_nchw, int N, int Cin, int H, int W, const float* w, const float* bias, void* y, int y_ld, int Cout,
                             void* relu_bits, void* stream);
/* ReLU bits of a bf16 NHWC tensor (N, H, W, C), C % 64 == 0: bits = (y > 0), in the layout of csrc/relu_bits.hpp (stand-alone producer; the convolutions write
 * them from their epilogues). */
size_t mis_relu_bits_bytes(int N, int H, int W, int C);
int mis_relu_bits(const void* y, int y_ld, int N, int H, int W, int C, void* bits, void* stream);
size_t mis_conv3x3_first_wgrad_workspace_bytes(int N, int Cin, int H, int W, int Cout);
int mis_conv3x3_first_wgrad(int dtype, const float* x_nchw, int N, int Cin, int H, int W, const void* dy, int dy_ld, int Cout,
                            float* workspace, float* dw, float* db, void* stream);

/* Column sums: out[c] = alpha * sum_pixels x[pixel][c] (+ fold: C = fold*Cq columns summed onto Cq outputs).
 * Bias gradients of every conv (convolution_backward's bias part). */
size_t mis_colsum_workspace_bytes(long long npix, int C);
int mis_colsum(int dtype, const void* x, int ld, long long npix, int C, int fold, float alpha, float* workspace, float* out,
               void* stream);

/* MaxPool 2x2 (2-D) / 2x2x2 (3-D), stride 2, floor.  model/unet2d/layers.py:147; model/unet3d/buildingblocks.py:409-418. */
int mis_maxpool2_fwd(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, int D, int H, int W, int C, void* stream);
/* dx = relu_mask(x) * (scatter(dy to the first max of each window) + (add ? add : 0)); dx may alias add. */
int mis_maxpool2_bwd(int dtype, const void* x, int x_ld, const void* dy, int dy_ld, const void* add, int add_ld, void* dx, int dx_ld,
                     int N, int D, int H, int W, int C, int relu_mask, void* stream);
/* 2-D max-pool with "pool bits": the forward pass also writes one byte per POOLED element (N, H/2, W/2, C; even H, W) = [bit k: window position k = kh*2 + kw is
 * the arg-max, first maximum in scan order] | [bit 4 + k: the input at position k is > 0]; the backward pass takes those bytes instead of the input tensor:
 * dx = (x > 0) * (add + scatter(dy)), the same result as mis_maxpool2_bwd(relu_mask = 1) without reading x. */
int mis_maxpool2_fwd_pb(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, int H, int W, int C, void* pbits, void* stream);
int mis_maxpool2_bwd_pb(int dtype, const void* pbits, const void* dy, int dy_ld, const void* add, int add_ld, void* dx, int dx_ld, int N, int H, int W, int C,
                        void* stream);

/* Weight repack: fp32 master in the reference layout -> packed operand layouts of mis_conv_igemm. */
/* conv: w [Cout][Cin][taps] -> fwd pack [tap][Cout][Cin] and (optional) dgrad pack [tap'][Cin][Cout] with tap' mirrored */
/* All weight repacks of a network in ONE launch (the per-layer calls cost a launch each: 22 per 2-D train step).  `items_dev`: DEVICE array of n entries;
 * kind 0 = mis_pack_conv_weight(w, rows = Cout, cols = Cin, taps), kind 1 = mis_pack_convt_weight(w, rows = Cin, cols = Cq); max_rows / max_cols = the largest
 * rows / cols over the entries (grid extent; smaller entries' surplus blocks exit). */
typedef struct MisPackItem {
    const float* w;
    void* w_fwd;
    void* w_dgrad;     /* may be NULL for kind 0 */
    int rows, cols, taps, kind;
} MisPackItem;
int mis_pack_batch(int dtype, const MisPackItem* items_dev, int n, int max_rows, int max_cols, void* stream);
/* ... with a compact grid: entry i owns blocks [blk0, blk0 + nbx * nby) (blk0 ascending from 0; nbx = ceil(cols / 32), nby = ceil(rows / 32); for kind 0 a block (by, bx)
 * packs the 32 x 32 tile at (row by * 32, column bx * 32)); total_blocks = the sum. */
typedef struct MisPackItem2 {
    const float* w;
    void* w_fwd;
    void* w_dgrad;
    int rows, cols, taps, kind;
    int blk0, nbx;
} MisPackItem2;
int mis_pack_batch2(int dtype, const MisPackItem2* items_dev, int n, int total_blocks, void* stream);
int mis_pack_conv_weight(int dtype, const float* w, int Cout, int Cin, int taps, void* w_fwd, void* w_dgrad, void* stream);
/* convT k2s2: w [Cin][Cq][2][2] -> fwd pack [1][4*Cq][Cin] (row = ab*Cq + c) and dgrad pack [1][Cin][4*Cq] */
int mis_pack_convt_weight(int dtype, const float* w, int Cin, int Cq, void* w_fwd, void* w_dgrad, void* stream);

/* 1x1 segmentation head + loss + argmax (+ backward in the same pass).
 * unet.py:89,127 (final_conv), unet.py:1184-1188 (CE | BCEWithLogits), argmax sites metrics.py:97 / predictor.py:167. */
typedef struct MisHeadDesc {
    int dtype;
    int loss;                /* 0 = cross entropy (labels int64 [N,spatial]), 1 = BCE with logits (targets f32 [N,C,spatial]),
                                2 = BCE + Dice (3-D, losses.py:167-178), -1 = no loss (inference),
                                3 = backward only from an external dL/dlogits given in `labels` (fp32 [N,C,spatial]) */
    long long npix_per_image; int N; int Cfeat; int C;     /* Cfeat = 64 */
    const void* y; int y_ld;       /* features (post-ReLU) */
    const float* w; const float* b;/* [C][Cfeat], [C] */
    const void* labels;
    float* logits;           /* [N][C][spatial] fp32 (reference layout), may be NULL */
    uint8_t* argmax;         /* [N][spatial] or NULL (C == 1: logits > 0) */
    float* workspace; size_t workspace_bytes;
    float* loss_out;         /* [1]; for loss 2: [2 + 3*C] = loss, bce mean, then the per-class sums I_c, P_c, T_c */
    /* backward (all NULL for forward only) */
    void* dy; int dy_ld;     /* dL/dfeatures * (y > 0) */
    float* dw; float* db;
    float grad_scale;        /* dL/dloss */
    float alpha, beta;       /* BCEDice weights */
    int phase;               /* loss 2 with backward: 0 = everything in one call; 1 = the forward part only (logits, arg-max, loss_out incl. the per-class sums);
                                2 = the gradient part only, from loss_out[1 .. 1 + 3*C] AS THE CALLER LEFT THEM - between the two the caller may sum I_c, P_c, T_c
                                over the ranks of a data-parallel job (36 bytes for C = 3), which makes the Dice term that of the GLOBAL batch, as the reference
                                computes it on the gathered batch (model/unet3d/trainer.py:312-318).  In phase 2 pass beta x world_size when grad_scale carries
                                1 / world_size: the Dice gradient from global sums is already the global derivative. */
} MisHeadDesc;
size_t mis_head_workspace_bytes(const MisHeadDesc* d);
int mis_head_loss(const MisHeadDesc* d, void* stream);
/* The last 3x3 convolution of the 2-D U-Net (up_conv.3.second: Conv2d(64k, 64, 3, p1) + bias + ReLU, layers.py:122-126) with the head, the loss and their backward in its
 * EPILOGUE (round 4, csrc/conv_ppd_head.hip): `conv` describes the convolution exactly as for mis_conv_igemm, `head` the head exactly as for mis_head_loss with
 * head->dy == conv->y0 (and equal ld): the 64-channel feature map is never written - dL/dfeatures takes its place - and mis_head_loss's pass over it disappears.
 * Eligible: bf16, 2-D, Cin % 64 == 0, Cout == 64, bias + ReLU, grids that fill 32-row tiles, C == 2 with cross entropy or C == 1 with BCE, training form (dy / dw / db
 * given).  mis_conv3x3_head_fused_eligible returns 1 / 0; the call itself returns MIS_EUNSUPPORTED otherwise (the caller then runs the two entry points separately). */
int mis_conv3x3_head_fused_eligible(const MisConvDesc* conv, const MisHeadDesc* head);
int mis_conv3x3_head_fused(const MisConvDesc* conv, const MisHeadDesc* head, void* stream);

/* Global grad-norm clip + AdamW on flat fp32 buffers (HF Trainer: clip_grad_norm_(1.0) then torch.optim.AdamW). */
size_t mis_adamw_workspace_bytes(long long n);
int mis_sumsq(const float* g, long long n, float* workspace /* partials */, void* stream);
int mis_adamw_step(float* p, const float* g, float* m, float* v, long long n, const float* sumsq_partials, int npartials,
                   float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   float* gradnorm_out, void* stream);
int mis_sumsq_npartials(long long n);
/* the same step with the optimizer state on the device, for captured (hipGraph) train steps: *step_dev is advanced by the kernel when `advance` != 0
 * (once per optimizer step: the first of the decay / no-decay calls), lr is read from lr_dev[0]; hyper_ws: 3 floats of scratch per call */
int mis_adamw_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* sumsq_partials, int npartials, float max_norm,
                       const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, int* step_dev, int advance, float* hyper_ws,
                       float* gradnorm_out, void* stream);

/* GroupNorm statistics (per sample, per channel sums) and finalisation to a*x+b. buildingblocks.py:87-92 */
size_t mis_chanstats_workspace_bytes(int N, long long npix, int C);
int mis_chanstats(int dtype, const void* x, int ld, int N, long long npix, int C, float* workspace, float* sum, float* sumsq,
                  void* stream);

/* GroupNorm folded around the 3-D convolutions (model/unet3d/buildingblocks.py:81-92, eps 1e-5).
 * fwd: per-channel sums (mis_chanstats) of up to two sources (encoder features | nearest-upsampled features, mult 8)
 *      -> per (n,c) scale/shift consumed by mis_conv_igemm / mis_wgrad (in_scale / in_shift), + mean/rstd per (n,g). */
int mis_gn_fwd_finalize(const float* sum0, const float* sq0, int C0, float mult0, const float* sum1, const float* sq1, int C1, float mult1,
                        int N, int G, double count, const float* gamma, const float* beta, float eps, int Cpad, float* scale, float* shift,
                        float* mean, float* rstd, void* stream);
/* fwd apply (bf16 engines): y[n][v][c_off + c] = round(fma(x[n][src(v)][c], scale[n*Ctot + c_off + c], shift[...])) for the Cs channels of ONE source of the
 * (virtual concat) input - the GroupNorm output of model/unet3d/buildingblocks.py:87-92 written once per SingleConv, with the arithmetic and rounding mis_conv_igemm applies
 * when it folds the affine into operand staging; up != 0: the source lives on the half grid (F.interpolate nearest 2x, buildingblocks.py:671-673).  y = channel 0 of the
 * destination tensor (row stride y_ld >= Ctot).  It feeds the ping-pong kernels, whose operands arrive by LDS-DMA (no ALU on that path). */
int mis_gn_apply(int dtype, const void* x, int x_ld, int Cs, int up, int N, int D, int H, int W, const float* scale, const float* shift, int Ctot, int c_off,
                 void* y, int y_ld, void* stream);
/* bwd: S1 = sum dy, S2 = sum dy*x for one source (up != 0: source on the half grid, dy summed over the 8 children) */
size_t mis_gn_bwd_stats_workspace_bytes(int N, int Cs);
int mis_gn_bwd_stats(int dtype, const void* dy, int dy_ld, const void* x, int x_ld, int Cs, int up, int N, int D, int H, int W,
                     float* workspace, float* S1, float* S2, int Ctot, int c_off, void* stream);
/* The same S1 / S2 WITHOUT reading dyn and x (bf16 engines whose conv operand is the materialised xn = scale * x + shift): from the per-sample weight gradients
 * (MisWgradDesc::dw_per_sample) and the sums of gy = dL/d(conv output) over the whole volume (MisWgradDesc::dbias_per_sample) and its boundary faces / edges /
 * corners - S1 = sum_{co,tap} W * G_tap, T = sum_{co,tap} W * dW_n, S2 = (T - shift * S1) / scale (see csrc/groupnorm.hip).  w, dw_per_sample: [Cout][Cw][27]
 * (reference layout, Cw >= Cs: Cw may include channel padding), scale / shift rows of stride sld, mean [N][groups]; writes S1, S2 [N][Cs]. */
size_t mis_gn_bwd_stats_from_dw_workspace_bytes(int N, int Cout);
int mis_gn_bwd_stats_from_dw(int dtype, const void* gy, int gy_ld, int N, int D, int H, int W, int Cout, const float* w, const float* dw_per_sample, int Cw,
                             const float* gy_colsum_per_sample, const float* scale, const float* shift, int sld, const float* mean, int groups, int Cs,
                             float* workspace, float* S1, float* S2, void* stream);
/* Conditioning guard of mis_gn_bwd_stats_from_dw: flags[l] = 1 when any channel c of GroupNorm layer l has |gamma[c]| < ratio * |beta[c]| (gamma = params + gamma_off[l],
 * beta = params + beta_off[l], count[l] channels; host arrays, nlayers <= 32).  There the normalised bf16 operand carries too little of x for the route through the weight
 * gradient (its error grows like |beta / gamma| * 2^-9): the caller sends such a layer through mis_gn_bwd_stats instead.  One launch, device-side flags, no synchronisation. */
int mis_gn_cond(const float* params, const unsigned long long* gamma_off, const unsigned long long* beta_off, const int* count, int nlayers, float ratio, int* flags,
                void* stream);

int mis_gn_bwd_finalize(const float* S1, const float* S2, const float* mean, const float* rstd, const float* gamma, int N, int C, int G,
                        double count, float* p, float* q, float* r, float* dgamma, float* dbeta, void* stream);
/* dx = (p*sum_children(dy) + mult*(q*x + r)) [* (x > 0)] [+ add] for one source */
int mis_gn_bwd_apply(int dtype, const void* dy, int dy_ld, const void* x, int x_ld, int Cs, int up, int N, int D, int H, int W, const float* p,
                     const float* q, const float* r, int Ctot, int c_off, int relu_mask, const void* add, int add_ld, void* dx, int dx_ld,
                     void* stream);
/* first 3-D layer: fp32 single-channel volume, GroupNorm(1) affine per sample, Conv3d 1 -> Cout (<= 64) k3 p1, ReLU */
int mis_first3d_fwd(int dtype, const float* x, const float* scale, const float* shift, int sstride, int N, int D, int H, int W, const float* w,
                    int Cout, void* y, int y_ld, int Cpad, void* stream);
size_t mis_first3d_bwd_workspace_bytes(void);
int mis_first3d_bwd(int dtype, const float* x, const float* mean /*[N]*/, const float* rstd /*[N]*/, const float* gamma /*[1]*/,
                    const float* beta /*[1]*/, int N, int D, int H, int W, const void* dy, int dy_ld, int Cpad, const float* w, int Cout,
                    float* workspace, float* dw, float* dgamma /*[1]*/, float* dbeta /*[1]*/, float* dxn /*optional (N,D,H,W) or NULL*/, void* stream);
int mis_relu_mask(int dtype, const void* dy, int dy_ld, const void* y, int y_ld, void* dx, int dx_ld, long long npix, int C, void* stream);
/* y = alpha * x * (mask > 0) on channels-last views: training-mode nn.Dropout inside a create_conv layer ('d', model/unet3d/buildingblocks.py:105-106), forward and
 * (with dy for x) backward; the caller draws the mask */
int mis_mask_scale(int dtype, const void* x, int x_ld, const void* mask, int mask_ld, void* y, int y_ld, long long npix, int C, float alpha, void* stream);
/* min and max of n fp32 values -> out[0], out[1] (two-stage, workspace >= 2 * 1024 floats): data-derived bounds of `Normalize`
 * (augment/unet3d_augment/transforms.py:575-586: np.min / np.max of the volume or of one channel) */
int mis_minmax(const float* x, long long n, float* workspace, float* out, void* stream);

/* 'deconv' upsampling of the 3-D decoders: ConvTranspose3d(k3, s2, p1, no bias) then nearest resize 2n-1 -> 2n
 * (model/unet3d/buildingblocks.py:676-728).  The contraction is mis_conv_igemm (ksize 1, 27*C columns: cols[i][k*C+c], k = (kd*3+kh)*3+kw);
 * col2im gathers cols (N,d,h,w,27*C) into u (N,2d,2h,2w,C; row stride u_ld); im2col is its adjoint: gu -> gcols (N,d,h,w,27*C). */
int mis_convt3_col2im(int dtype, const void* cols, void* u, int u_ld, int N, int d, int h, int w, int C, void* stream);
int mis_convt3_im2col(int dtype, const void* gu, int gu_ld, void* gcols, int N, int d, int h, int w, int C, void* stream);

/* BatchNorm2d of the `unetConv2` block (model/unet2d/layers.py:17-25: Conv2d(bias) -> BatchNorm2d -> ReLU; eps 1e-5, momentum 0.1).
 * fwd: mis_chanstats(z) -> mis_bn_fwd_finalize (batch statistics; running_mean/var updated in place with the unbiased variance;
 *      training == 0 uses the running statistics) -> per (n,c) scale/shift [N][C] -> mis_affine_act (y = relu(z*scale+shift)).
 * bwd: g = dy*(y>0) (mis_relu_mask), S1 = sum g, S2 = sum g*z (mis_gn_bwd_stats) -> mis_bn_bwd_finalize -> p,q,r [N][C] for
 *      mis_gn_bwd_apply (dz = p*g + q*z + r) and dgamma, dbeta. count_total = N*H*W. */
int mis_bn_fwd_finalize(const float* sum, const float* sumsq, int N, int C, double count_total, const float* gamma, const float* beta,
                        float eps, float momentum, float* running_mean, float* running_var, int training, float* scale, float* shift,
                        float* mean, float* rstd, void* stream);
int mis_bn_bwd_finalize(const float* S1, const float* S2, const float* mean, const float* rstd, const float* gamma, int N, int C,
                        double count_total, int training, float* p, float* q, float* r, float* dgamma, float* dbeta, void* stream);
/* fused form of the first and last backward passes: the ReLU mask is recomputed from z (fma(z, scale, shift) > 0, the forward's expression), so no
 * g tensor is written:  S1 = sum m*dy, S2 = sum m*dy*z;  dz = p*(m*dy) + q*z + r.  scale/shift/p/q/r: [N][C] fp32. */
size_t mis_bn_bwd_stats_workspace_bytes(int N, int C);
int mis_bn_bwd_stats(int dtype, const void* dy, int dy_ld, const void* z, int z_ld, int N, long long npix, int C, const float* scale,
                     const float* shift, float* workspace, float* S1, float* S2, void* stream);
int mis_bn_bwd_apply(int dtype, const void* dy, int dy_ld, const void* z, int z_ld, int N, long long npix, int C, const float* scale,
                     const float* shift, const float* p, const float* q, const float* r, void* dz, int dz_ld, void* stream);
int mis_affine_act(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, long long npix, int C, const float* scale,
                   const float* shift, int relu, void* stream);

/* On-device 3-D augmentation (augment/unet3d_augment/transforms.py:25-133,495-523,608-619). Volumes are (nvol, D, H, W),
 * 4- or 8-byte elements for the geometric ops (raw fp32 / int64 labels), fp32 for the intensity ops; src != dst. */
int mis_aug_flip_rot90(const void* src, void* dst, long long nvol, int D, int H, int W, int flipmask /*bit0 D, bit1 H, bit2 W*/, int k,
                       int elem_size, void* stream);
/* CropToFixed (augment/unet3d_augment/transforms.py:194-247) on (nslices, H, W) planes: dst (nslices, CH, CW)[i][j] = src[r(y0+i)][r(x0+j)],
 * r = numpy.pad(mode='reflect') index rule; elem_size 4 or 8 (raw fp32 / int64 labels). */
int mis_aug_crop_reflect(const void* src, void* dst, long long nslices, int H, int W, int y0, int x0, int CH, int CW, int elem_size, void* stream);
int mis_aug_rotate0(const void* src, void* dst, long long nvol, int D, int H, int W, int a0, int a1, const double* m4 /*host*/,
                    const double* off2 /*host*/, int elem_size, void* stream);
/* ... with scipy.ndimage's other boundary modes (RandomRotate(mode=...), order 0): mode 0 'reflect' / 'grid-mirror', 1 'constant', 2 'nearest', 3 'mirror', 4 'wrap',
 * 5 'grid-wrap', 6 'grid-constant'; cval_bits = the bit pattern of the fill value in the element type (low elem_size bytes). */
int mis_aug_rotate0_mode(const void* src, void* dst, long long nvol, int D, int H, int W, int a0, int a1, const double* m4, const double* off2, int elem_size, int mode,
                         unsigned long long cval_bits, void* stream);

/* order-3 (cubic spline) variant, fp32 volumes: workspace = nvol*D*H*W doubles (the float64 spline coefficients scipy keeps) */
size_t mis_aug_rotate3_workspace_bytes(long long nvol, int D, int H, int W);
int mis_aug_rotate3(const float* src, float* dst, double* workspace, long long nvol, int D, int H, int W, int a0, int a1,
                    const double* m4 /*host*/, const double* off2 /*host*/, void* stream);
/* ... and for the other spline orders of scipy.ndimage.rotate: order 1 (linear), 2, 4, 5 (order 3 forwards to mis_aug_rotate3); mode 'reflect'; same workspace */
int mis_aug_rotate_spline(const float* src, float* dst, double* workspace, long long nvol, int D, int H, int W, int a0, int a1, const double* m4,
                          const double* off2, int order, void* stream);
/* ElasticDeformation (transforms.py:138-191): one axis of scipy.ndimage.gaussian_filter(mode='reflect') on float64 fields (weights = the
 * normalised kernel of radius int(4*sigma+0.5), on the device), and scipy.ndimage.map_coordinates(order 0 | 3, mode='reflect') at the voxel
 * grid displaced by alpha * (fz, fy, fx) (fz may be NULL). */
int mis_aug_gauss1d(const double* src, double* dst, long long nvol, int D, int H, int W, int axis, const double* weights, int radius, void* stream);
/* GaussianBlur3D (transforms.py:708-718: skimage.filters.gaussian(x, sigma) = scipy.ndimage.gaussian_filter(fp32 volume, sigma, mode='nearest', truncate=4)):
 * the same separable pass on fp32 volumes, rounded to fp32 after every axis as scipy does; mode 0 = 'reflect', 1 = 'nearest'. */
int mis_aug_gauss1d_f32(const float* src, float* dst, long long nvol, int D, int H, int W, int axis, const double* weights, int radius, int mode, void* stream);
int mis_aug_map_coordinates(const void* src, void* dst, double* workspace, long long nvol, int D, int H, int W, const double* fz, const double* fy,
                            const double* fx, double alpha, int order, int elem_size, void* stream);
int mis_aug_pointwise(const float* src, float* dst, long long n, float a, float b, int do_clip, float lo, float hi, float noise_std,
                      unsigned long long seed, void* stream);
int mis_aug_contrast(const float* src, float* dst, long long n, float mean, float alpha, void* stream);

/* 2-D sample pipeline of the reference's datasets (dataset/unet2d_dataset/MYDataset.py:127-157, albumentations 1.4.10 restated - the library is absent:
 * parity unpinned): Resize(OH, OW, nearest) -> HorizontalFlip -> VerticalFlip -> rot90 (k counter-clockwise quarter turns) -> Transpose ->
 * brightness/contrast table clip(v*alpha + beta*255) on the image -> CHW float / 255.  img: uint8 (H, W, C) in HBM, mask: uint8 (H, W) or NULL;
 * out_img fp32 (C, FH, FW), out_mask fp32 (FH, FW); (FH, FW) = (OH, OW), swapped when exactly one of (rot_k odd, transpose) holds. */
int mis_aug2d_u8(const unsigned char* img, const unsigned char* mask, int H, int W, int C, int OH, int OW, int hflip, int vflip, int rot_k, int transpose,
                 int use_bc, float alpha, float beta, float* out_img, float* out_mask, void* stream);

/* Stand-alone BCE + Dice loss on logits (model/unet3d/losses.py:7-33,83-129,167-178): x, t fp32 (N, C, S).
 * fwd: out[0] = alpha*mean(BCEWithLogits) + beta*(1 - mean_c dice_c), out[1] = the BCE mean, out[2+4c..] = {sum s*t, sum s^2, sum t^2, dice_c}
 *      (s = sigmoid(x)); out must hold 2 + 4*C floats and is the `sums` input of bwd.
 * bwd: dx = grad_out[0] * dLoss/dx (grad_out: device scalar). */
size_t mis_bcedice_workspace_bytes(int C);
int mis_bcedice_fwd(const float* x, const float* t, int N, int C, long long S, float alpha, float beta, int normalize /*1: s = sigmoid(x), 0: s = x*/,
                    void* workspace, float* out, void* stream);
int mis_bcedice_bwd(const float* x, const float* t, int N, int C, long long S, float alpha, float beta, int normalize, const float* sums,
                    const float* grad_out, float* dx, void* stream);

/* Further criteria of the reference's 3-D loss factory (model/unet3d/losses.py:309-346): nn.CrossEntropyLoss(ignore_index) on logits fp32 (N, C, S) with
 * int64 labels (N, S), mean over the non-ignored voxels (out[2] = {loss, count}; all ignored -> nan like torch); nn.MSELoss / L1Loss / SmoothL1Loss
 * (kind 0 / 1 / 2, beta 1), mean over n elements.  bwd: dx = grad_out[0] * dLoss/dx.  workspace: mis_loss_workspace_bytes(). */
size_t mis_loss_workspace_bytes(void);
int mis_ce3d_fwd(const float* logits, const long long* labels, int N, int C, long long S, long long ignore_index, void* workspace, float* out, void* stream);
int mis_ce3d_bwd(const float* logits, const long long* labels, int N, int C, long long S, long long ignore_index, const float* fwd_out,
                 const float* grad_out, float* dx, void* stream);
int mis_pointloss_fwd(int kind, const float* x, const float* t, long long n, void* workspace, float* out, void* stream);
int mis_pointloss_bwd(int kind, const float* x, const float* t, long long n, const float* grad_out, float* dx, void* stream);

/* UNet 3+ skip paths (model/unet2d/unet.py:136-446): nn.MaxPool2d(k, k, ceil_mode=True) and nn.Upsample(scale_factor=s, mode='bilinear')
 * (align_corners=False), NHWC, forward and backward.  Pool output grid = ceil(H/k) x ceil(W/k); bilinear output grid = H*s x W*s. */
int mis_maxpoolk_fwd(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, int H, int W, int C, int k, void* stream);
int mis_maxpoolk_bwd(int dtype, const void* x, int x_ld, const void* dy, int dy_ld, void* dx, int dx_ld, int N, int H, int W, int C, int k, void* stream);
int mis_bilinear_up_fwd(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, int H, int W, int C, int scale, void* stream);
size_t mis_bilinear_up_bwd_workspace_bytes(int N, int H, int W, int C, int scale);
int mis_bilinear_up_bwd(int dtype, const void* dy, int dy_ld, void* dx, int dx_ld, int N, int H, int W, int C, int scale, float* workspace, void* stream);

/* Classification-guided module of UNet_3Plus_DeepSup_CGM (model/unet2d/unet.py:998-1003, 1012-1038, 1147-1153): per sample
 * cls[n][k] = sigmoid(max over pixels of (w[k] . x[n][pix] + b[k])), k = 0, 1;  gate[n] = float(argmax_k cls[n][k]) (first maximum);
 * mis_scale_sigmoid: out = sigmoid(x * gate[n]) (gy == NULL) or its backward gy * y(1-y) * gate[n].  x of mis_cgm_gate: (N, npix, C) channels-last. */
int mis_cgm_gate(int dtype, const void* x, int x_ld, int N, long long npix, int C, const float* w, const float* b, float* cls, float* gate, void* stream);
int mis_scale_sigmoid(const float* x, const float* gy, const float* gate, int N, long long per_sample, float* out, void* stream);

/* conv3x3(pad 1)(bilinear_upsample_s(x)) of the UNet 3+ decoder-to-decoder branches (model/unet2d/unet.py:190-192, 229-236, 273-285, 322-339) without
 * the upsampled tensor: the channel contraction runs on the low-resolution grid (mis_conv_igemm, ksize 1, column tap*C + co of z), then
 *   fwd: y[N][h*s][w*s][C] = bias + sum over the 3x3 taps inside the upsampled image of bilinear(z[..][tap*C + c]) at (o + tap - 1)
 *   bwd: dz[N][h][w][9*C]  = the adjoint of that gather applied to dy
 * z / dz are dense (row stride 9*C); y / dy have a row stride.  C % (16 bytes / element) == 0, 1 <= scale <= 32. */
size_t mis_upconv_gather_fwd_workspace_bytes(int dtype, int N, int h, int w, int scale, int C);
int mis_upconv_gather_fwd(int dtype, const void* z, void* y, int y_ld, const float* bias, int N, int h, int w, int scale, int C, void* workspace /* NULL: single pass */,
                          void* stream);
int mis_upconv_gather_bwd(int dtype, const void* dy, int dy_ld, void* dz, int N, int h, int w, int scale, int C, void* stream);

/* SegmentationLoss of the UNet 3+ path (model/unet2d/loss.py:21-70): F1Loss + MSSSIMLoss (pytorch_msssim 1.0.0 MS_SSIM, data_range 1, 5 scales,
 * 11-tap Gaussian) + IoULoss on single-channel (logits, targets) fp32 (N, H, W); min(H, W) > 160.
 * fwd: out[0] = w_f1*F1 + w_msssim*MSSSIM + w_iou*IoU (SegmentationLoss: all 1), out[1..3] = the three terms, out[4..5] internal;
 *      the workspace keeps the state for bwd.
 * bwd (same workspace / out): dlogits = grad_out[0] * dLoss/dlogits. */
size_t mis_segloss_workspace_bytes(int N, int H, int W);
int mis_segloss_fwd(const float* logits, const float* target, int N, int H, int W, float w_f1, float w_msssim, float w_iou, void* workspace,
                    float* out /*[8]*/, void* stream);
int mis_segloss_bwd(const float* target, int N, int H, int W, void* workspace, const float* out, const float* grad_out, float* dlogits, int with_msssim, void* stream);

/* Residual 3-D U-Net pieces (model/unet3d/buildingblocks.py:255-325 ResNetBlock; model.py:197-232): y = [relu](a + b) for the residual join and the
 * decoder's sum-joining; the first block's 1x1x1 conv from ONE input channel on the raw fp32 volume, y[v][c] = w[c]*x[v] + b[c], and its
 * weight / bias gradients dw[c] = sum x*dy, db[c] = sum dy. */
int mis_add_act(int dtype, const void* a, int a_ld, const void* b, int b_ld, void* y, int y_ld, long long npix, int C, int relu, void* stream);
int mis_expand1_fwd(int dtype, const float* x, const float* w, const float* b, void* y, int y_ld, long long nvox, int C, void* stream);
size_t mis_expand1_bwd_workspace_bytes(int C);
int mis_expand1_bwd(int dtype, const float* x, const void* dy, int dy_ld, long long nvox, int C, float* workspace, float* dw, float* db, void* stream);

/* Squeeze & excitation of the residual 3-D blocks: `ResNetBlockSE` with se_module 'scse' (model/unet3d/buildingblocks.py:326-362; model/unet3d/se.py:18-116
 * ChannelSELayer3D with reduction_ratio 1, SpatialSELayer3D, ChannelSpatialSELayer3D = elementwise max of the two).  e, y, g, de: (N, S, C) channels-last
 * with a row stride, C % 64 == 0; everything else fp32.
 *   fwd: mis_chanstats(e) -> mis_se_fc_fwd (mean, z1 = W1 mean + b1, a = sigmoid(W2 relu(z1) + b2), all [N][C]) -> mis_se_apply_fwd
 *        (bgate[n][v] = sigmoid(w . e[n][v] + b0), y = max(e*a, e*bgate)).
 *   bwd: mis_se_bwd_reduce (dq [N][S], da [N][C], dw [C], db0 [1]) -> mis_se_fc_bwd (dW1, db1, dW2, db2 [C][C] / [C], cross [N][C]) ->
 *        mis_se_bwd_apply (de = [e > 0] * (g * (selected gate) + cross + dq * w); e is a ReLU output and its mask is fused; de may alias g).
 * workspace: mis_se_bwd_workspace_bytes(N, C), shared by mis_se_bwd_reduce and mis_se_fc_bwd of one block. */
int mis_se_fc_fwd(const float* chan_sum, double count, const float* W1, const float* b1, const float* W2, const float* b2, int N, int C, float* mean,
                  float* z1, float* a, void* stream);
int mis_se_apply_fwd(int dtype, const void* e, int e_ld, int N, long long S, int C, const float* a, const float* w, const float* b0, float* bgate,
                     void* y, int y_ld, void* stream);
size_t mis_se_bwd_workspace_bytes(int N, int C);
int mis_se_bwd_reduce(int dtype, const void* g, int g_ld, const void* e, int e_ld, int N, long long S, int C, const float* a, const float* bgate,
                      float* workspace, float* dq, float* da, float* dw, float* db0, void* stream);
int mis_se_fc_bwd(const float* da, const float* a, const float* z1, const float* mean, const float* W1, const float* W2, int N, int C, double count,
                  float* workspace, float* dW1, float* db1, float* dW2, float* db2, float* cross, void* stream);
int mis_se_bwd_apply(int dtype, const void* g, int g_ld, const void* e, int e_ld, int N, long long S, int C, const float* a, const float* bgate,
                     const float* dq, const float* w, const float* cross, void* de, int de_ld, void* stream);

/* The three layers of model/unet3d/se.py called ON THEIR OWN (ChannelSELayer3D :18-53, SpatialSELayer3D :56-98, ChannelSpatialSELayer3D :101-116; the reference's
 * ResNetBlockSE reaches them with se_module 'cse' / 'sse' / 'scse', buildingblocks.py:346-352): the passes above with
 *   mode 0: y = max(e*a, e*bgate)    1: y = e*a (cSE; w, b0, bgate, dq unused, may be null)    2: y = e*bgate (sSE; a, cross unused, may be null)
 * and relu_mask = 0: the input is any tensor, not a ReLU output (de is not masked; where e == 0 both products tie and de = g (a + bgate) / 2, torch.max's rule).
 * Any reduction ratio: the caller zero-pads W1 [C/r][C] and W2 [C][C/r] to C x C (mis_se_fc_fwd / mis_se_fc_bwd are unchanged). */
int mis_se_layer_fwd(int dtype, const void* e, int e_ld, int N, long long S, int C, const float* a, const float* w, const float* b0, float* bgate, void* y,
                     int y_ld, int mode, void* stream);
int mis_se_layer_bwd_reduce(int dtype, const void* g, int g_ld, const void* e, int e_ld, int N, long long S, int C, const float* a, const float* bgate,
                            float* workspace, float* dq, float* da, float* dw, float* db0, int mode, void* stream);
int mis_se_layer_bwd_apply(int dtype, const void* g, int g_ld, const void* e, int e_ld, int N, long long S, int C, const float* a, const float* bgate,
                           const float* dq, const float* w, const float* cross, void* de, int de_ld, int mode, int relu_mask, void* stream);

/* Evaluation metrics of the 2-D trainer (trainer/metrcis.py:61-109,153-168 `compute_metrics`): sigmoid with +1e-6 in the denominator,
 * threshold = global mean probability, per-sample IoU / Dice, mean over samples.  values, labels: fp32 (N, npix); out[3] = {iou, dice, threshold}.
 * values_are_logits = 0 with a given threshold gives compute_iou / compute_dice on ready-made predictions. */
size_t mis_seg_metrics_workspace_bytes(int N, long long npix);
int mis_seg_metrics(const float* values, const float* labels, int N, long long npix, int values_are_logits, int auto_threshold, float threshold,
                    void* workspace, float* out, void* stream);

/* MeanIoU of the 3-D validation loop (model/unet3d/metrics.py:33-103, expand_as_one_hot model/unet3d/utils.py:222-254): counts[n][c] = {sum(pred & tgt),
 * sum(pred | tgt)} with pred = one-hot of the first channel maximum (C == 1: prob > 0.5), tgt = one-hot fp32 (N, C, S) cast to uint8 or int64 labels
 * (N, S); ignore_index zeroes pred and tgt where the target equals it.  probs fp32 (N, C, S), C <= 16; counts: N*C*2 uint64 (zeroed here). */
int mis_iou3d_counts(const float* probs, const void* target, int target_is_labels, int N, int C, long long S, int has_ignore, long long ignore_index,
                     unsigned long long* counts, void* stream);

/* Patch-tiled volume prediction (model/unet3d/predictor.py:85-168; dataset/unet3d_dataset/utils.py:85-125,314-361).
 * gather: patches[p] (C, PD, PH, PW) = the volume window starting at origins[p] - halo, np.pad(mode='reflect') outside the volume
 *         (origins: device int32 (NP, 3) positions of the patch INTERIORS in the unpadded volume; PD = interior + 2*hd, ...);
 * accumulate: halo cut off, map[c][origin + v] += act(pred[c][halo + v]), norm[origin + v] += 1 (uint8, one launch per patch = the
 *         reference's accumulation order); channel >= 0 keeps only that channel (prediction_channel);
 * finalize: prob = map / norm and/or seg = argmax_c (first maximum), uint16. */
int mis_patch_gather_reflect(const float* vol, int C, int D, int H, int W, const int* origins, int NP, int PD, int PH, int PW, int hd, int hh,
                             int hw, float* patches, void* stream);
int mis_patch_accumulate(const float* pred, int C, int PD, int PH, int PW, int hd, int hh, int hw, int activation, int channel, int oz, int oy,
                         int ox, float* map, unsigned char* norm, int D, int H, int W, void* stream);
int mis_pred_finalize(const float* map, const unsigned char* norm, int C, long long nvox, float* prob, unsigned short* seg, void* stream);

/* Stand-alone / general-shape 3-D building blocks (csrc/blocks3d.hip): what the fused engines refuse.  Replaces, per call, the torch modules that
 * model/unet3d/buildingblocks.py:14-113 `create_conv` strings together for an arbitrary order ('gcr', 'cge', 'cl', 'crg', ...), the Encoder's
 * MaxPool3d / AvgPool3d (:409-418) and `F.interpolate(x, size=encoder_features.size()[2:], mode='nearest')` (:671-673).
 * Channels-last (N, D, H, W, ld) tensors; C = the channels touched (a multiple of 16 bytes except for the gathers, which are scalar).
 *   act: 0 none, 1 ReLU, 2 LeakyReLU(slope), 3 ELU(alpha = slope);  scale / shift: [N][C] fp32 or both NULL (activation only).
 *   mis_norm_act_fwd: y = act(scale*x + shift);  mis_norm_act_bwd: dz = dy * act'(scale*x + shift)  (gradient w.r.t. the affine's output)
 *   mis_gn_*_finalize_ld: GroupNorm finalisation as mis_gn_fwd_finalize / mis_gn_bwd_finalize, for per-channel arrays with row stride ld >= C whose
 *     padding entries [C, ld) are written as 0 (sum / sq / S1 / S2 / scale / shift / p / q / r all [N][ld]).
 *   mis_pool3d_*: mode 0 = max, 1 = average; window = stride = (kd, kh, kw), floor mode (output D/kd x H/kh x W/kw); the max gradient goes to the
 *     first maximum in (d, h, w) order, as torch's max_pool3d does.
 *   mis_gather3d_fwd: y[n,d,h,w,c] = x[n,mD[d],mH[h],mW[w],c], c < C (x, y point at the first channel of the slices); mD/mH/mW = DEVICE int arrays.
 *   mis_gather3d_bwd: dx[n,s..] = sum of dy over the destination indices [rX[2s], rX[2s+1]) per axis (DEVICE int pairs), fixed order. */
int mis_norm_act_fwd(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, long long npix, int C, const float* scale, const float* shift, int act,
                     float slope, void* stream);
int mis_norm_act_bwd(int dtype, const void* dy, int dy_ld, const void* x, int x_ld, void* dz, int dz_ld, int N, long long npix, int C, const float* scale,
                     const float* shift, int act, float slope, void* stream);
int mis_gn_fwd_finalize_ld(const float* sum, const float* sq, int N, int C, int ld, int G, double count, const float* gamma, const float* beta, float eps,
                           float* scale, float* shift, float* mean, float* rstd, void* stream);
int mis_gn_bwd_finalize_ld(const float* S1, const float* S2, const float* mean, const float* rstd, const float* gamma, int N, int C, int ld, int G,
                           double count, float* p, float* q, float* r, float* dgamma, float* dbeta, void* stream);
int mis_pool3d_fwd(int dtype, int mode, int kd, int kh, int kw, const void* x, int x_ld, void* y, int y_ld, int N, int D, int H, int W, int C, void* stream);
int mis_pool3d_bwd(int dtype, int mode, int kd, int kh, int kw, const void* x, int x_ld, const void* dy, int dy_ld, void* dx, int dx_ld, int N, int D, int H,
                   int W, int C, void* stream);
int mis_gather3d_fwd(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, int sD, int sH, int sW, int dD, int dH, int dW, int C, const int* mD,
                     const int* mH, const int* mW, void* stream);
int mis_gather3d_bwd(int dtype, const void* dy, int dy_ld, void* dx, int dx_ld, int N, int sD, int sH, int sW, int dD, int dH, int dW, int C, const int* rD,
                     const int* rH, const int* rW, void* stream);

/* layout helpers */
int mis_nchw_to_nhwc(int dtype_out, const float* x, void* y, int y_ld, int N, int C, long long spatial, void* stream);
int mis_nhwc_to_nchw(int dtype_in, const void* x, int x_ld, float* y, int N, int C, long long spatial, void* stream);

/* test hooks: raw MFMA / LDS-transpose fragment probes (tests/test_gpu_fragments.py) */
int mis_probe_mfma(int which, const float* a, const float* b, float* c, void* stream);

/* numpy legacy RandomState.normal() on the device (csrc/mt19937.hip): the Gaussian-noise FIELD of AdditiveGaussianNoise, bit-comparable with the
 * reference's `m + random_state.normal(0, std, size=m.shape)` (augment/unet3d_augment/transforms.py:608-619).
 * mis_mt19937_words: the next n 32-bit outputs of MT19937 from (key[624], pos), both updated in place (device memory).
 * mis_legacy_normal: out[i] = (float)((double)in[i] + scale * g_i), g = numpy's legacy_gauss over `words` (4 words per polar attempt; has_gauss / gauss0 = the
 * generator's cache on entry); result5 (device) = {attempts consumed, cache filled?, cached value (double bits), -, 1 = success / 0 = words ran out}. */
int mis_mt19937_words(unsigned int* key_io, int* pos_io, unsigned int* out, long long n, void* stream);
int mis_legacy_normal(const unsigned int* words, long long nattempts, const float* in, float* out, long long count, double scale, int has_gauss, double gauss0,
                      unsigned long long* result5, void* stream);
/* The same stream from many workgroups (round 3).  mis_mt_jump: states = [S][624] generator keys; for b < njumps, states[dst0 + b] = the key J words after
 * states[src0 + b] (per_jump_src) or after states[src0] (all jumps from one key), where g_words = the 624-word bit mask of t^(J-1) mod phi - one mask, or one per
 * jump (per_jump_mask: g_words[b*624..]; host side: augment/unet3d_augment/mt_jump.py; MT19937 is linear over GF(2)); `parts` workgroups share the sum of a jump.
 * mis_mt_generate: chunk c (one workgroup) walks the stream positions [c*J, (c+1)*J) from states[c]; tempered words of positions [lo, hi) go to out[position - lo];
 * raw_block >= 0 (or used_dev, below): the untempered 624 words of that block go to raw_out (numpy's key array).  mis_legacy_normal_par = mis_legacy_normal over many workgroups
 * (count accepted attempts per block, scan, write), same result5 protocol; workspace >= mis_legacy_normal_par_workspace_bytes(nattempts). */
int mis_mt_jump(unsigned int* states, int src0, int dst0, int njumps, const unsigned int* g_words, int per_jump_src, int per_jump_mask, int parts, void* stream);
int mis_mt_generate(const unsigned int* states, int nchunks, long long J, long long lo, long long hi, unsigned int* out, long long raw_block, unsigned int* raw_out,
                    const unsigned long long* used_dev /* optional: result5 of mis_legacy_normal_par still on the device - the raw block is then the one in which numpy's
                    position ends after used_dev[0] attempts from position pos0 (nothing is written for 0 attempts) */, long long pos0, void* stream);
size_t mis_legacy_normal_par_workspace_bytes(long long nattempts);
int mis_legacy_normal_par(const unsigned int* words, long long nattempts, const float* in, float* out, long long count, double scale, int has_gauss, double gauss0,
                          void* workspace, unsigned long long* result5, void* stream);

/* ---- data-parallel gradient exchange: one RCCL communicator per process (csrc/comm.cpp) ----------------------------------------------
 * Replaces nn.DataParallel in the reference's 3-D trainer (model/unet3d/trainer.py:23-24) / the DDP wrapper HF Trainer adds under torchrun (train.py).
 * librccl is bound at run time; mis_comm_unique_id (rank 0) -> carry the 128 bytes to every rank -> mis_comm_init on the rank's HIP device ->
 * mis_allreduce_bucket(buf, n, stream): in-place fp32 SUM over all ranks, enqueued on `stream`, never synchronises -> mis_comm_finalize. */
int mis_comm_unique_id(void* out128);
int mis_comm_init(const void* unique_id128, int rank, int world);
int mis_comm_world(void);      /* ranks of the live communicator, 0 if none */
int mis_allreduce_bucket(float* buf, long long n, void* stream);
int mis_comm_finalize(void);

#ifdef __cplusplus
}
#endif
#endif
