// Kernel-argument structs shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_pp.hip).
#pragma once
#include "common.hpp"

struct SrcView {
    const void* p;
    int ld, D, H, W;
};

struct ConvArgs {
    int N, D, H, W, Cin, Cout, Cin0, Cout0;
    SrcView x0, x1;
    const float* in_scale;
    const float* in_shift;
    const void* w;
    const float* bias;
    int relu;
    const void* mask;
    int mask_ld;
    void* y0;
    int y0_ld, y0_mode;
    void* y1;
    int y1_ld, y1_mode;
    int tilesD, tilesH, tilesW, nSp, nCt;
    unsigned char* relu_bits;          // ReLU bits of the output (relu_bits.hpp) or nullptr
    const unsigned char* mask_bits;    // ReLU bits applied to the output instead of `mask`, or nullptr
    int order, zg;     // conv3d_pp.hip: tile order (1 = column-tile major, 2 = spatial major) and depth-group size of the plane walk
    const float* gn_p;                 // conv3d_pp.hip, PP_EM_GN: GroupNorm backward in the epilogue (MisConvDesc.gn_p ...), x = mask
    const float* gn_q;
    const float* gn_r;
    int gn_ld, gn_relu;
    unsigned* tq;                      // dynamic tile queue of the persistent kernels (dispatch_cfg.hpp mis_tile_queue) or nullptr = static stride
};

extern thread_local bool g_conv_bits_fused;            // conv_igemm.hip: the launched kernel writes relu_bits itself
// conv_pp.hip: the ping-pong 3x3 kernel for the bf16 2-D layers (returns MIS_OK after the launch, or an error); `eligible` says whether a
// descriptor can take that path at all (dispatch in conv_igemm.hip decides)
bool conv_pp_eligible(const MisConvDesc* d);
bool conv_ppc64_auto(const MisConvDesc* d);            // 64-column blocks of the column-segment kernel chosen by default (Cin >= 128)
bool conv_pp_rs64_eligible(const MisConvDesc* d);      // 64 -> 64 channels: the register-stationary ping-pong kernel
int conv_ppc_choice(const MisConvDesc* d);            // 4 / 2: launch_conv_pp runs conv_ppc_kernel<8, 4 / 2> (the kernels that read / write ReLU bits in their epilogue); 0: another kernel
int launch_conv_pp(const MisConvDesc* d, hipStream_t stream, const char** tag);
// conv_ppd.hip: the deep-prefetch form of conv_ppc_kernel<8, 2> (conv_ppc_choice(d) == 2 descriptors with Cin % 64 == 0)
int launch_conv_ppd(const MisConvDesc* d, hipStream_t stream, const char** tag);
// conv3d_pp.hip: the same structure for the bf16 3x3x3 layers (single source, no operand affine)
bool conv3d_pp_eligible(const MisConvDesc* d);
int launch_conv3d_pp(const MisConvDesc* d, hipStream_t stream, const char** tag);
// gemm1_pp.hip: the ping-pong 1x1 GEMM for the bf16 2-D transposed-convolution GEMMs (plain or pixel-shuffled destination, optional mask / ReLU bits)
bool gemm1_pp_eligible(const MisConvDesc* d);
int launch_gemm1_pp(const MisConvDesc* d, hipStream_t stream, const char** tag);
// conv3d_f32.hip: the fp32 3x3x3 layers of the fused 3-D engine (single plain source: the materialised GroupNorm output), all-DMA / fragment-prefetched
bool conv3d_f32_eligible(const MisConvDesc* d);
int launch_conv3d_f32(const MisConvDesc* d, hipStream_t stream, const char** tag);
long long conv3d_f32_stats_rows(const MisConvDesc* d);
// conv_pps.hip (round-5 experiment, MIS_CONV_PPS=1): the 128-column 3x3 layers with ONE wave per SIMD and both fragment sets in registers
bool conv_pps_eligible(const MisConvDesc* d);
int launch_conv_pps(const MisConvDesc* d, hipStream_t stream, const char** tag);
// conv_ppc2.hip (round 5, MIS_CONV_PPC2): conv_ppc_kernel<8, 4> with register-resident DMA offsets and weight fragments rolling through the M segment
bool conv_ppc2_eligible(const MisConvDesc* d);
int launch_conv_ppc2(const MisConvDesc* d, hipStream_t stream, const char** tag);
