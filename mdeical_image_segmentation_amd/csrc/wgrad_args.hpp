// Kernel-argument structs shared by the weight-gradient kernels (wgrad.hip, wgrad_pp.hip).
#pragma once
#include "common.hpp"

struct WSrc {
    const void* p;
    int ld, D, H, W;
};

struct WgArgs {
    int N, D, H, W, Cin, Cout, Cin0;
    WSrc x0, x1;
    const float* in_scale;
    const float* in_shift;
    const void* dy;
    int dy_ld;
    float* partial;
    float* bias_partial;   // [nsplit][Cout] column sums of dy (bias gradient), or nullptr
    int tilesD, tilesH, tilesW, ntiles, nsplit, tps;
    int nCi, nCo, KDn, TT;
    int spb, tpsamp;       // per-sample split-K (wgrad_pp.hip, MisWgradDesc::dw_per_sample): spb > 0 = splits per sample, each over tps tiles of the sample's tpsamp; 0 = one tile range over the batch
};

// wgrad_pp.hip: ping-pong weight-gradient kernel for the bf16 2-D 3x3 layers.  `wgrad_pp_eligible` = the descriptor can take that path;
// `wgrad_pp_nsplit` = number of fp32 partial slabs it writes (one per persistent block); the launch fills a.partial / a.bias_partial
// exactly like wgrad_kernel (slab [split][tap][ci][co], bias [split][co]) so the same reduction kernels finish the job.
bool wgrad_pp_eligible(const MisWgradDesc* d);
int wgrad_pp_nsplit(const MisWgradDesc* d);
int wgrad_pp_splits_per_sample(const MisWgradDesc* d);      // > 0 when d asks for per-sample weight gradients: slabs [n * k, (n + 1) * k) belong to sample n, k = this x (slabs per split)
int launch_wgrad_pp(const MisWgradDesc* d, float* partial, float* bias_partial, hipStream_t stream, const char** tag);
// wgrad_f32.hip: streaming fp32 weight gradient of the 3x3 / 3x3x3 layers (plain single-source operand, no bias gradient); slabs as wgrad_kernel writes them
bool wgrad_f32_eligible(const MisWgradDesc* d);
int wgrad_f32_nsplit(const MisWgradDesc* d);
int launch_wgrad_f32(const MisWgradDesc* d, float* partial, hipStream_t stream, const char** tag);
