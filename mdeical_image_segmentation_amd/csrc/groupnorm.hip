// GroupNorm around the 3-D convolutions (reference model/unet3d/buildingblocks.py:81-92: 'gcr' = GroupNorm -> Conv3d -> ReLU,
// eps 1e-5, num_groups 8 or 1 when C < 8) for gfx950.
//
// Forward: GroupNorm is folded to a per-(sample, channel) affine a*x+b that conv_igemm / wgrad apply while staging
// their input tile, so the normalised tensor is never written.  This file computes a and b from per-channel sums
// (mis_chanstats), including the virtual concat (encoder features | nearest-upsampled decoder features) where a
// group may straddle both sources.
// Backward: with dy = dL/d(GN output) from the conv's dgrad,
//   S1[n,c] = sum_v dy, S2[n,c] = sum_v dy*x                                (gn_bwd_stats, 2-stage, fixed order)
//   dbeta = sum_n S1, dgamma = sum_n rstd*(S2 - mean*S1)                     (gn_bwd_finalize)
//   dx = p[n,c]*dy + q[n,c]*x + r[n,c]  with  p = rstd*gamma, q = -rstd^2*B/m, r = -q*mean - rstd*A/m,
//        A = sum_{c in g} gamma*S1, B = sum_{c in g} gamma*rstd*(S2 - mean*S1)     (gn_bwd_apply; optional ReLU mask of x,
//   optional accumulate, and for the upsampled source the 8 children of a coarse voxel are summed here)
#include <stdlib.h>

#include "common.hpp"
#include "dispatch_cfg.hpp"

// ---------------------------------------------------------------------------------------------------------
// forward finalize: per (n, g) mean / rstd from per-channel sums of up to two sources; per (n, c) scale / shift
// ---------------------------------------------------------------------------------------------------------
__global__ void gn_fwd_finalize_kernel(const float* __restrict__ sum0, const float* __restrict__ sq0, int C0, float mult0,
                                       const float* __restrict__ sum1, const float* __restrict__ sq1, int C1, float mult1, int N, int G,
                                       double count, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                       int Cpad, float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                       float* __restrict__ rstd_out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * G) return;
    const int n = idx / G, g = idx - n * G;
    const int C = C0 + C1, cpg = C / G;
    double s = 0.0, q = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        if (c < C0) {
            s += (double)sum0[n * C0 + c] * mult0;
            q += (double)sq0[n * C0 + c] * mult0;
        } else {
            s += (double)sum1[n * C1 + (c - C0)] * mult1;
            q += (double)sq1[n * C1 + (c - C0)] * mult1;
        }
    }
    const double m = count * cpg;
    const double mean = s / m;
    double var = q / m - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    mean_out[idx] = (float)mean;
    rstd_out[idx] = (float)rstd;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        const double a = rstd * (double)gamma[c];
        scale[n * Cpad + c] = (float)a;
        shift[n * Cpad + c] = (float)((double)beta[c] - mean * a);
    }
    if (g == G - 1)   // zero the channel padding (a padded channel must stay exactly 0 after the affine)
        for (int c = C; c < Cpad; ++c) {
            scale[n * Cpad + c] = 0.f;
            shift[n * Cpad + c] = 0.f;
        }
}

extern "C" int mis_gn_fwd_finalize(const float* sum0, const float* sq0, int C0, float mult0, const float* sum1, const float* sq1, int C1,
                                   float mult1, int N, int G, double count, const float* gamma, const float* beta, float eps, int Cpad,
                                   float* scale, float* shift, float* mean, float* rstd, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(sum0 && sq0 && gamma && beta && scale && shift && mean && rstd, MIS_EINVAL, "gn_fwd_finalize: null pointer");
    MIS_REQUIRE(N > 0 && G > 0 && C0 > 0 && C1 >= 0 && (C0 + C1) % G == 0 && Cpad >= C0 + C1, MIS_EINVAL, "gn_fwd_finalize: bad sizes");
    MIS_REQUIRE(C1 == 0 || (sum1 && sq1), MIS_EINVAL, "gn_fwd_finalize: source 1 missing");
    hipLaunchKernelGGL(gn_fwd_finalize_kernel, dim3((N * G + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), sum0, sq0, C0, mult0,
                       sum1, sq1, C1, mult1, N, G, count, gamma, beta, eps, Cpad, scale, shift, mean, rstd);
    MIS_LAUNCH_CHECK("gn_fwd_finalize");
    return MIS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// forward apply (round 3, bf16 engines): the normalised tensor of ONE source of the (virtual concat) input, written once per SingleConv -
//   y[n][v][c_off + c] = round(fma(x[n][src(v)][c], scale[n][c_off + c], shift[n][c_off + c])),   src(v) = v, or (z>>1, y>>1, x>>1) for the nearest-upsampled source
// - the SAME arithmetic and rounding conv_igemm / wgrad apply while staging (HaloStager::store), so the ping-pong kernels (conv3d_pp.hip, wgrad_pp.hip), whose
// operands arrive by LDS-DMA, see a plain single-source tensor.  One 16-byte chunk per thread; the 8 readers of a coarse voxel hit L2.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, int x_ld, int Cs, int up, int D, int H, int W, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int Ctot, int c_off, T* __restrict__ y, int y_ld) {
    // a thread keeps ONE 16-byte channel chunk (its scale / shift stay in registers) and walks voxels; blockIdx.y = sample; 32-bit index arithmetic
    // (a sample has fewer than 2^31 voxels), four voxels in flight per thread
    constexpr int EPC = Tr<T>::EPC, UN = 4;
    const int nch = Cs / EPC;
    const int vpb = 256 / nch;                              // voxels per block and pass (nch <= 256)
    const int tid = threadIdx.x;
    const int ch = tid % nch, vl = tid / nch;
    if (vl >= vpb) return;
    const int n = blockIdx.y;
    const unsigned np = (unsigned)D * H * W;
    const int sH = up ? H / 2 : H, sW = up ? W / 2 : W;
    const size_t snp = (size_t)(up ? D / 2 : D) * sH * sW;
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = scale[(size_t)n * Ctot + c_off + ch * EPC + e];
        sh[e] = shift[(size_t)n * Ctot + c_off + ch * EPC + e];
    }
    const T* xb = x + (size_t)n * snp * x_ld + (size_t)ch * EPC;
    T* yb = y + (size_t)n * np * y_ld + c_off + (size_t)ch * EPC;
    const unsigned stride = gridDim.x * (unsigned)vpb;
    for (unsigned v0 = blockIdx.x * (unsigned)vpb + vl; v0 < np; v0 += UN * stride) {
        u32x4 raw[UN];
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            const unsigned v = v0 + k * stride;
            raw[k] = u32x4{0u, 0u, 0u, 0u};
            if (v < np) {
                unsigned sv = v;
                if (up) {
                    const unsigned xx = v % (unsigned)W, t = v / (unsigned)W;
                    const unsigned yy = t % (unsigned)H, zz = t / (unsigned)H;
                    sv = ((zz >> 1) * (unsigned)sH + (yy >> 1)) * (unsigned)sW + (xx >> 1);
                }
                raw[k] = *reinterpret_cast<const u32x4*>(xb + (size_t)sv * x_ld);
            }
        }
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            const unsigned v = v0 + k * stride;
            if (v < np) {
                float f[EPC];
                unpack_chunk<T>(raw[k], f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
                *reinterpret_cast<u32x4*>(yb + (size_t)v * y_ld) = pack_chunk<T>(f);
            }
        }
    }
}

extern "C" int mis_gn_apply(int dtype, const void* x, int x_ld, int Cs, int up, int N, int D, int H, int W, const float* scale, const float* shift, int Ctot,
                            int c_off, void* y, int y_ld, void* stream) {
    (void)hipGetLastError();
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "gn_apply: bad dtype");
    MIS_REQUIRE(x && scale && shift && y, MIS_EINVAL, "gn_apply: null pointer");
    MIS_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cs > 0 && Cs % EPC == 0 && x_ld % EPC == 0 && y_ld % EPC == 0 && c_off % EPC == 0 && c_off >= 0 &&
                    c_off + Cs <= Ctot && Ctot <= y_ld,
                MIS_EINVAL, "gn_apply: sizes / alignment");
    MIS_REQUIRE(!up || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0), MIS_EUNSUPPORTED, "gn_apply: upsampled source needs an even grid");
    MIS_REQUIRE(Cs / EPC <= 256 && (long long)D * H * W < (1ll << 31) && N <= 65535, MIS_EUNSUPPORTED, "gn_apply: at most 256 chunks per voxel, 2^31 voxels per sample");
    const int vpb = 256 / (Cs / EPC);
    long long blocks = ((long long)D * H * W + 4 * vpb - 1) / (4 * vpb);
    const long long cap = (4096 + N - 1) / N;
    if (blocks > cap) blocks = cap;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(gn_apply_kernel<__bf16>, dim3((unsigned)blocks, N), dim3(256), 0, s, (const __bf16*)x, x_ld, Cs, up, D, H, W, scale, shift, Ctot, c_off,
                           (__bf16*)y, y_ld);
    else
        hipLaunchKernelGGL(gn_apply_kernel<float>, dim3((unsigned)blocks, N), dim3(256), 0, s, (const float*)x, x_ld, Cs, up, D, H, W, scale, shift, Ctot, c_off,
                           (float*)y, y_ld);
    MIS_LAUNCH_CHECK("gn_apply");
    return MIS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// backward statistics for ONE source of the (virtual concat) input:  S1 = sum dy, S2 = sum dy * x
// grid (D,H,W) is the conv's pixel grid; with up != 0 the source lives on the half grid and each coarse voxel first sums
// dy over its 2x2x2 children.  Stage 1 writes slab partials, stage 2 (gn_bwd_finalize) adds them in a fixed order.
// ---------------------------------------------------------------------------------------------------------
constexpr int GB_SLABS = 128;

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const T* __restrict__ dy, int dy_ld, const T* __restrict__ x, int x_ld, int Cs, int up,
                                                           int D, int H, int W, float* __restrict__ part /*[N][slabs][2][Cs]*/) {
    constexpr int EPC = Tr<T>::EPC;
    __shared__ float red[2][256 * 8];
    const int nchunks = Cs / EPC;
    const int chb = nchunks < 256 ? nchunks : 256;
    const int rows = 256 / chb;
    const int tid = threadIdx.x, cl = tid % chb, r = tid / chb;
    const int chunk = blockIdx.y * chb + cl;
    const int n = blockIdx.z, nslabs = gridDim.x;
    const int sD = up ? D / 2 : D, sH = up ? H / 2 : H, sW = up ? W / 2 : W;
    const long long snp = (long long)sD * sH * sW;
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
    if (chunk < nchunks && r < rows) {
        const T* xb = x + (size_t)n * snp * x_ld + (size_t)chunk * EPC;
        const T* db = dy + (size_t)n * D * H * W * dy_ld + (size_t)chunk * EPC;
        for (long long p = (long long)blockIdx.x * rows + r; p < snp; p += (long long)nslabs * rows) {
            float xf[EPC], g[EPC];
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(xb + p * x_ld), xf);
            if (!up) {
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(db + p * dy_ld), g);
            } else {
                const int xx = (int)(p % sW);
                const long long t = p / sW;
                const int yy = (int)(t % sH), zz = (int)(t / sH);
#pragma unroll
                for (int e = 0; e < EPC; ++e) g[e] = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const size_t pix = ((size_t)(2 * zz + (k >> 2)) * H + 2 * yy + ((k >> 1) & 1)) * W + 2 * xx + (k & 1);
                    float c[EPC];
                    unpack_chunk<T>(*reinterpret_cast<const u32x4*>(db + pix * dy_ld), c);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) g[e] += c[e];
                }
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                s1[e] += g[e];
                s2[e] = fmaf(g[e], xf[e], s2[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        red[0][tid * EPC + e] = s1[e];
        red[1][tid * EPC + e] = s2[e];
    }
    __syncthreads();
    if (r == 0 && chunk < nchunks) {
        for (int k = 1; k < rows; ++k)
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                s1[e] += red[0][(k * chb + cl) * EPC + e];
                s2[e] += red[1][(k * chb + cl) * EPC + e];
            }
        float* o = part + (((size_t)n * nslabs + blockIdx.x) * 2) * Cs + (size_t)chunk * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            o[e] = s1[e];
            o[Cs + e] = s2[e];
        }
    }
}

__global__ void gn_bwd_statsum_kernel(const float* __restrict__ part, int nslabs, int N, int Cs, float* __restrict__ S1, float* __restrict__ S2,
                                      int Ctot, int c_off) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * Cs) return;
    const int n = idx / Cs, c = idx - n * Cs;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < nslabs; ++k) {
        a += (double)part[(((size_t)n * nslabs + k) * 2) * Cs + c];
        b += (double)part[(((size_t)n * nslabs + k) * 2 + 1) * Cs + c];
    }
    S1[n * Ctot + c_off + c] = (float)a;
    S2[n * Ctot + c_off + c] = (float)b;
}

extern "C" size_t mis_gn_bwd_stats_workspace_bytes(int N, int Cs) { return (size_t)N * GB_SLABS * 2 * Cs * sizeof(float); }

extern "C" int mis_gn_bwd_stats(int dtype, const void* dy, int dy_ld, const void* x, int x_ld, int Cs, int up, int N, int D, int H, int W,
                                float* workspace, float* S1, float* S2, int Ctot, int c_off, void* stream) {
    (void)hipGetLastError();
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(dy && x && workspace && S1 && S2, MIS_EINVAL, "gn_bwd_stats: null pointer");
    MIS_REQUIRE(Cs > 0 && Cs % EPC == 0 && dy_ld % EPC == 0 && x_ld % EPC == 0 && c_off % EPC == 0, MIS_EINVAL, "gn_bwd_stats: alignment");
    MIS_REQUIRE(!up || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0), MIS_EUNSUPPORTED, "gn_bwd_stats: upsampled source needs an even grid");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nchunks = Cs / EPC;
    const int chb = nchunks < 256 ? nchunks : 256;
    const int ngroups = (nchunks + chb - 1) / chb;
    const int rows = 256 / chb;
    const long long snp = (long long)(up ? D / 2 : D) * (up ? H / 2 : H) * (up ? W / 2 : W);
    long long slabs = (snp + rows * 8 - 1) / (rows * 8);
    if (slabs > GB_SLABS) slabs = GB_SLABS;
    if (slabs < 1) slabs = 1;
    const size_t coff = (size_t)c_off * (dtype == MIS_BF16 ? 2 : 4);
    const char* dyp = reinterpret_cast<const char*>(dy) + coff;
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(gn_bwd_stats_kernel<__bf16>, dim3((unsigned)slabs, ngroups, N), dim3(256), 0, s, (const __bf16*)dyp, dy_ld, (const __bf16*)x,
                           x_ld, Cs, up, D, H, W, workspace);
    else
        hipLaunchKernelGGL(gn_bwd_stats_kernel<float>, dim3((unsigned)slabs, ngroups, N), dim3(256), 0, s, (const float*)dyp, dy_ld, (const float*)x,
                           x_ld, Cs, up, D, H, W, workspace);
    MIS_LAUNCH_CHECK("gn_bwd_stats");
    hipLaunchKernelGGL(gn_bwd_statsum_kernel, dim3((N * Cs + 255) / 256), dim3(256), 0, s, (const float*)workspace, (int)slabs, N, Cs, S1, S2, Ctot,
                       c_off);
    MIS_LAUNCH_CHECK("gn_bwd_statsum");
    return MIS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Backward statistics WITHOUT a pass over dyn and x (round 3, bf16 engines with the materialised operand xn = a*x + b):
//   T[n,c]  = sum_v dyn[n,v,c] * xn[n,v,c] = sum_{co,tap} W[co][c][tap] * dW_n[n][co][c][tap]        (dW_n = the weight gradient of sample n, zero padding included)
//   S1[n,c] = sum_v dyn[n,v,c]             = sum_{co,tap} W[co][c][tap] * G[n][tap][co],   G = sum of g_y over the output voxels whose tap lands inside the volume
//   S2[n,c] = sum_v dyn * x                = (T - b * S1) / a
// G needs the total of g_y per (sample, channel) - the weight-gradient kernel's fused column sums - and its sums over the 6 faces, 12 edges and 8 corners of the
// volume (inclusion - exclusion over the excluded boundary planes of a tap): 4 % of the voxels.  W is taken as the kernels see it (rounded to the storage type).
// A channel whose scale a is exactly 0 (gamma = 0) carries no information about x in xn: its S2 is set to mean * S1 (dgamma of that channel reads 0; dx is exact
// regardless: it only needs gamma * (S2 - mean * S1) = T - beta * S1).  mis_gn_bwd_stats stays the exact route for that corner and for fp32.
// ---------------------------------------------------------------------------------------------------------
constexpr int GB_BSLABS = 16;

// region r = sz*9 + sy*3 + sx, s = 0: the whole axis, 1: its first index, 2: its last; r = 0 (everything) is not computed here.  grid (26, N, GB_BSLABS)
template <typename T>
__global__ __launch_bounds__(256) void gn_border_sums_kernel(const T* __restrict__ gy, int gy_ld, int C, int D, int H, int W, float* __restrict__ part /*[N][27][slabs][C]*/) {
    constexpr int EPC = Tr<T>::EPC;
    __shared__ float red[256 * 8];
    const int r = blockIdx.x + 1, n = blockIdx.y, slab = blockIdx.z, nslabs = gridDim.z;
    const int sel[3] = {r / 9, (r / 3) % 3, r % 3};
    const int L[3] = {D, H, W};
    int ext[3], org[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        ext[a] = sel[a] == 0 ? L[a] : 1;
        org[a] = sel[a] == 2 ? L[a] - 1 : 0;
    }
    const long long nvox = (long long)ext[0] * ext[1] * ext[2];
    const int nch = C / EPC;                       // <= 256
    const int tid = threadIdx.x, ch = tid % nch, slot = tid / nch, nslots = 256 / nch;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    if (slot < nslots) {
        const T* base = gy + (size_t)n * D * H * W * gy_ld + (size_t)ch * EPC;
        for (long long i = (long long)slab * nslots + slot; i < nvox; i += (long long)nslabs * nslots) {
            const int x = org[2] + (int)(i % ext[2]);
            const long long t = i / ext[2];
            const int y = org[1] + (int)(t % ext[1]), z = org[0] + (int)(t / ext[1]);
            float f[EPC];
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + (((size_t)z * H + y) * W + x) * gy_ld), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += f[e];
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[tid * EPC + e] = acc[e];
    __syncthreads();
    if (slot == 0) {
        for (int k = 1; k < nslots; ++k)
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += red[(k * nch + ch) * EPC + e];
        float* o = part + (((size_t)n * 27 + r) * nslabs + slab) * C + (size_t)ch * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = acc[e];
    }
}

// R[n][r][co] = sum over the slabs of region r (fixed order).  grid (26, N)
__global__ __launch_bounds__(256) void gn_border_reduce_kernel(const float* __restrict__ part, int nslabs, int C, float* __restrict__ R /*[N][27][C]*/) {
    const int r = blockIdx.x + 1, n = blockIdx.y;
    for (int co = threadIdx.x; co < C; co += 256) {
        const float* p = part + (((size_t)n * 27 + r) * nslabs) * C + co;
        double v = 0.0;
        for (int k = 0; k < nslabs; ++k) v += (double)p[(size_t)k * C];
        R[((size_t)n * 27 + r) * C + co] = (float)v;
    }
}

// G[n][tap][co] = total - faces + edges - corner for the boundary planes tap excludes.  grid (27, N), 256 threads over co
__global__ __launch_bounds__(256) void gn_border_finalize_kernel(const float* __restrict__ R, const float* __restrict__ tot /*[N][C]*/, int C, float* __restrict__ G /*[N][27][C]*/) {
    const int tap = blockIdx.x, n = blockIdx.y;
    const int d[3] = {tap / 9 - 1, (tap / 3) % 3 - 1, tap % 3 - 1};
    for (int co = threadIdx.x; co < C; co += 256) {
        double g = 0.0;
        for (int m = 0; m < 8; ++m) {              // subsets of the axes whose boundary plane is excluded
            int r = 0, bits = 0;
            bool ok = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                int s_ = 0;
                if ((m >> a) & 1) {
                    if (d[a] == 0) ok = false;
                    s_ = d[a] < 0 ? 1 : 2;
                    ++bits;
                }
                r = r * 3 + s_;
            }
            if (!ok) continue;
            const double v = r == 0 ? (double)tot[(size_t)n * C + co] : (double)R[((size_t)n * 27 + r) * C + co];
            g += (bits & 1) ? -v : v;
        }
        G[((size_t)n * 27 + tap) * C + co] = (float)g;
    }
}

// grid (Cs, N); one block per (n, c): reductions over (co, tap) in a fixed order
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_from_dw_kernel(const float* __restrict__ w /*[Cout][Cw][27]*/, const float* __restrict__ dwn /*[N][Cout][Cw][27]*/, int Cout,
                                                               int Cw, const float* __restrict__ G /*[N][27][Cout]*/, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int sld, const float* __restrict__ mean, int groups, int cpg,
                                                               float* __restrict__ S1, float* __restrict__ S2, int Ctot) {
    __shared__ double rT[256], rS[256];
    const int c = blockIdx.x, n = blockIdx.y;
    double t_ = 0.0, s_ = 0.0;
    const int total = Cout * 27;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int co = i / 27, tap = i - co * 27;
        float wv = w[((size_t)co * Cw + c) * 27 + tap];
        if constexpr (sizeof(T) == 2) wv = (float)(__bf16)wv;        // the packed operand the dgrad kernel multiplies with
        t_ += (double)wv * (double)dwn[(((size_t)n * Cout + co) * Cw + c) * 27 + tap];
        s_ += (double)wv * (double)G[((size_t)n * 27 + tap) * Cout + co];
    }
    rT[threadIdx.x] = t_;
    rS[threadIdx.x] = s_;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            rT[threadIdx.x] += rT[threadIdx.x + o];
            rS[threadIdx.x] += rS[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double a = scale[(size_t)n * sld + c], b = shift[(size_t)n * sld + c];
        const double s1 = rS[0], tt = rT[0];
        const double mu = mean[n * groups + c / cpg];
        S1[(size_t)n * Ctot + c] = (float)s1;
        S2[(size_t)n * Ctot + c] = (float)(a != 0.0 ? (tt - b * s1) / a : mu * s1);
    }
}

extern "C" size_t mis_gn_bwd_stats_from_dw_workspace_bytes(int N, int Cout) { return ((size_t)N * 27 * GB_BSLABS * Cout + 2 * (size_t)N * 27 * Cout) * sizeof(float); }

extern "C" int mis_gn_bwd_stats_from_dw(int dtype, const void* gy, int gy_ld, int N, int D, int H, int W, int Cout, const float* w, const float* dw_per_sample, int Cw,
                                        const float* gy_colsum_per_sample, const float* scale, const float* shift, int sld, const float* mean, int groups, int Cs,
                                        float* workspace, float* S1, float* S2, void* stream) {
    (void)hipGetLastError();
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "gn_bwd_stats_from_dw: bad dtype");
    MIS_REQUIRE(gy && w && dw_per_sample && gy_colsum_per_sample && scale && shift && mean && workspace && S1 && S2, MIS_EINVAL, "gn_bwd_stats_from_dw: null pointer");
    MIS_REQUIRE(N > 0 && N <= 65535 && D > 0 && H > 0 && W > 0 && Cout > 0 && Cout % EPC == 0 && Cout / EPC <= 256 && gy_ld % EPC == 0 && Cs > 0 && Cs <= Cw &&
                    Cs <= sld && groups > 0 && Cs % groups == 0,
                MIS_EINVAL, "gn_bwd_stats_from_dw: sizes");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* part = workspace;
    float* R = workspace + (size_t)N * 27 * GB_BSLABS * Cout;
    float* G = R + (size_t)N * 27 * Cout;
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(gn_border_sums_kernel<__bf16>, dim3(26, N, GB_BSLABS), dim3(256), 0, s, (const __bf16*)gy, gy_ld, Cout, D, H, W, part);
    else
        hipLaunchKernelGGL(gn_border_sums_kernel<float>, dim3(26, N, GB_BSLABS), dim3(256), 0, s, (const float*)gy, gy_ld, Cout, D, H, W, part);
    MIS_LAUNCH_CHECK("gn_border_sums");
    hipLaunchKernelGGL(gn_border_reduce_kernel, dim3(26, N), dim3(256), 0, s, (const float*)part, GB_BSLABS, Cout, R);
    MIS_LAUNCH_CHECK("gn_border_reduce");
    hipLaunchKernelGGL(gn_border_finalize_kernel, dim3(27, N), dim3(256), 0, s, (const float*)R, gy_colsum_per_sample, Cout, G);
    MIS_LAUNCH_CHECK("gn_border_finalize");
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(gn_stats_from_dw_kernel<__bf16>, dim3(Cs, N), dim3(256), 0, s, w, dw_per_sample, Cout, Cw, (const float*)G, scale, shift, sld, mean, groups,
                           Cs / groups, S1, S2, Cs);
    else
        hipLaunchKernelGGL(gn_stats_from_dw_kernel<float>, dim3(Cs, N), dim3(256), 0, s, w, dw_per_sample, Cout, Cw, (const float*)G, scale, shift, sld, mean, groups,
                           Cs / groups, S1, S2, Cs);
    MIS_LAUNCH_CHECK("gn_stats_from_dw");
    return MIS_OK;
}

// ---- conditioning guard of the statistics-from-dW route.  That route recovers sum dyn * x from T = sum dyn * xn with xn = round_bf16(a * x + b): when |gamma| is
// small against |beta| the stored operand carries little of x, and dgamma = (T - beta * S1) / gamma amplifies its rounding error by |beta / gamma| (2^-9 per element at
// gamma = 1, beta = 0: the one-ulp bar of tests/test_gpu_engine3d.py).  One launch over all GroupNorm layers: flags[l] = 1 when any channel of layer l has
// |gamma| < ratio * |beta|; the engine reads the flags back asynchronously and sends flagged layers through mis_gn_bwd_stats (the pass over dyn and x).
struct GnCondBatch {
    int n;
    unsigned long long goff[32], boff[32];
    int cnt[32];
};
__global__ __launch_bounds__(256) void gn_cond_kernel(const float* __restrict__ params, const GnCondBatch b, float ratio, int* __restrict__ flags) {
    const int l = blockIdx.x;
    __shared__ int any;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    const float* g = params + b.goff[l];
    const float* be = params + b.boff[l];
    int bad = 0;
    for (int c = threadIdx.x; c < b.cnt[l]; c += 256) bad |= (fabsf(g[c]) < ratio * fabsf(be[c])) ? 1 : 0;
    if (bad) atomicOr(&any, 1);
    __syncthreads();
    if (threadIdx.x == 0) flags[l] = any;
}

extern "C" int mis_gn_cond(const float* params, const unsigned long long* gamma_off, const unsigned long long* beta_off, const int* count, int nlayers, float ratio,
                           int* flags, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(params && gamma_off && beta_off && count && flags && nlayers > 0 && nlayers <= 32, MIS_EINVAL, "gn_cond: arguments (1..32 layers)");
    GnCondBatch b;
    b.n = nlayers;
    for (int i = 0; i < nlayers; ++i) {
        b.goff[i] = gamma_off[i];
        b.boff[i] = beta_off[i];
        b.cnt[i] = count[i];
    }
    hipLaunchKernelGGL(gn_cond_kernel, dim3(nlayers), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), params, b, ratio, flags);
    MIS_LAUNCH_CHECK("gn_cond");
    return MIS_OK;
}

// per (n, g): A, B -> per (n, c): p, q, r ; dgamma, dbeta.   xmult[c] = 1 for same-grid channels, 8 for upsampled
// (S1/S2 were summed over the SOURCE voxels with dy pre-summed over children, which is exactly the full-grid sum).
__global__ void gn_bwd_finalize_kernel(const float* __restrict__ S1, const float* __restrict__ S2, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, const float* __restrict__ gamma, int N, int C, int G, double count,
                                       float* __restrict__ p, float* __restrict__ q, float* __restrict__ r, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int cpg = C / G;
    if (idx < N * G) {
        const int n = idx / G, g = idx - n * G;
        const double mu = mean[idx], rs = rstd[idx];
        double A = 0.0, B = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            const double s1 = S1[n * C + c], s2 = S2[n * C + c];
            A += (double)gamma[c] * s1;
            B += (double)gamma[c] * rs * (s2 - mu * s1);
        }
        const double m = count * cpg;
        const double qq = -rs * rs * B / m;               // coefficient of x (B already carries one rstd)
        const double rr = -qq * mu - rs * A / m;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            p[n * C + c] = (float)(rs * (double)gamma[c]);
            q[n * C + c] = (float)qq;
            r[n * C + c] = (float)rr;
        }
    }
    if (idx < C) {
        const int c = idx, g = c / cpg;
        double dg = 0.0, db = 0.0;
        for (int n = 0; n < N; ++n) {
            const double mu = mean[n * G + g], rs = rstd[n * G + g];
            const double s1 = S1[n * C + c], s2 = S2[n * C + c];
            dg += rs * (s2 - mu * s1);
            db += s1;
        }
        dgamma[c] = (float)dg;
        dbeta[c] = (float)db;
    }
}

extern "C" int mis_gn_bwd_finalize(const float* S1, const float* S2, const float* mean, const float* rstd, const float* gamma, int N, int C,
                                   int G, double count, float* p, float* q, float* r, float* dgamma, float* dbeta, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(S1 && S2 && mean && rstd && gamma && p && q && r && dgamma && dbeta, MIS_EINVAL, "gn_bwd_finalize: null pointer");
    MIS_REQUIRE(N > 0 && C > 0 && G > 0 && C % G == 0, MIS_EINVAL, "gn_bwd_finalize: sizes");
    const int nthreads = (N * G > C) ? N * G : C;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((nthreads + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), S1, S2, mean, rstd,
                       gamma, N, C, G, count, p, q, r, dgamma, dbeta);
    MIS_LAUNCH_CHECK("gn_bwd_finalize");
    return MIS_OK;
}

// dx[source voxel][c] = (p*sum_children(dy) + mult*(q*x + r)) [* (x > 0)] [+ add]      (mult = 1 or 8)
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ dy, int dy_ld, const T* __restrict__ x, int x_ld, int Cs, int up,
                                                           int N, int D, int H, int W, const float* __restrict__ p, const float* __restrict__ q,
                                                           const float* __restrict__ r, int Ctot, int c_off, int relu_mask, const T* add, int add_ld,
                                                           T* dx, int dx_ld) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = Cs / EPC;
    const int sD = up ? D / 2 : D, sH = up ? H / 2 : H, sW = up ? W / 2 : W;
    const long long snp = (long long)sD * sH * sW;
    const long long total = (long long)N * snp * nch;
    const float mult = up ? 8.f : 1.f;
    // two chunks per thread and iteration, every load of both issued before the arithmetic (round 4: one dependent load pair per iteration left the pass latency-bound -
    // 178 us per call on average at 2 x 160^3)
    constexpr int U = 2;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += U * stride) {
        u32x4 xr[U], gr[U][8], ar[U];
        int chs[U], ns[U];
        long long sps[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u * stride;
            if (i >= total) break;
            const int ch = (int)(i % nch);
            const long long pp = i / nch;
            const int n = (int)(pp / snp);
            const long long sp = pp - (long long)n * snp;
            chs[u] = ch; ns[u] = n; sps[u] = sp;
            xr[u] = *reinterpret_cast<const u32x4*>(x + ((size_t)n * snp + sp) * x_ld + (size_t)ch * EPC);
            const T* db = dy + (size_t)n * D * H * W * dy_ld + (size_t)ch * EPC;
            if (!up) {
                gr[u][0] = *reinterpret_cast<const u32x4*>(db + (size_t)sp * dy_ld);
            } else {
                const int xx = (int)(sp % sW);
                const long long t = sp / sW;
                const int yy = (int)(t % sH), zz = (int)(t / sH);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const size_t pix = ((size_t)(2 * zz + (k >> 2)) * H + 2 * yy + ((k >> 1) & 1)) * W + 2 * xx + (k & 1);
                    gr[u][k] = *reinterpret_cast<const u32x4*>(db + pix * dy_ld);
                }
            }
            if (add != nullptr) ar[u] = *reinterpret_cast<const u32x4*>(add + ((size_t)n * snp + sp) * add_ld + (size_t)ch * EPC);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = i0 + u * stride;
            if (i >= total) break;
            const int ch = chs[u], n = ns[u];
            const long long sp = sps[u];
            float xf[EPC], g[EPC];
            unpack_chunk<T>(xr[u], xf);
            if (!up) {
                unpack_chunk<T>(gr[u][0], g);
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e) g[e] = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float c[EPC];
                    unpack_chunk<T>(gr[u][k], c);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) g[e] += c[e];
                }
            }
            const float* pc = p + (size_t)n * Ctot + c_off + ch * EPC;
            const float* qc = q + (size_t)n * Ctot + c_off + ch * EPC;
            const float* rc = r + (size_t)n * Ctot + c_off + ch * EPC;
            float o[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float v = fmaf(pc[e], g[e], mult * fmaf(qc[e], xf[e], rc[e]));
                if (relu_mask && !(xf[e] > 0.f)) v = 0.f;
                o[e] = v;
            }
            if (add != nullptr) {
                float af[EPC];
                unpack_chunk<T>(ar[u], af);
#pragma unroll
                for (int e = 0; e < EPC; ++e) o[e] += af[e];
            }
            *reinterpret_cast<u32x4*>(dx + ((size_t)n * snp + sp) * dx_ld + (size_t)ch * EPC) = pack_chunk<T>(o);
        }
    }
}

extern "C" int mis_gn_bwd_apply(int dtype, const void* dy, int dy_ld, const void* x, int x_ld, int Cs, int up, int N, int D, int H, int W,
                                const float* p, const float* q, const float* r, int Ctot, int c_off, int relu_mask, const void* add, int add_ld,
                                void* dx, int dx_ld, void* stream) {
    (void)hipGetLastError();
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(dy && x && p && q && r && dx, MIS_EINVAL, "gn_bwd_apply: null pointer");
    MIS_REQUIRE(Cs > 0 && Cs % EPC == 0 && dy_ld % EPC == 0 && x_ld % EPC == 0 && dx_ld % EPC == 0 && c_off % EPC == 0, MIS_EINVAL,
                "gn_bwd_apply: alignment");
    MIS_REQUIRE(add == nullptr || add_ld % EPC == 0, MIS_EINVAL, "gn_bwd_apply: add_ld");
    MIS_REQUIRE(!up || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0), MIS_EUNSUPPORTED, "gn_bwd_apply: upsampled source needs an even grid");
    const long long snp = (long long)(up ? D / 2 : D) * (up ? H / 2 : H) * (up ? W / 2 : W);
    long long blocks = ((long long)N * snp * (Cs / EPC) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t coff = (size_t)c_off * (dtype == MIS_BF16 ? 2 : 4);
    const char* dyp = reinterpret_cast<const char*>(dy) + coff;
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(gn_bwd_apply_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)dyp, dy_ld, (const __bf16*)x, x_ld, Cs, up,
                           N, D, H, W, p, q, r, Ctot, c_off, relu_mask, (const __bf16*)add, add_ld, (__bf16*)dx, dx_ld);
    else
        hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)dyp, dy_ld, (const float*)x, x_ld, Cs, up, N,
                           D, H, W, p, q, r, Ctot, c_off, relu_mask, (const float*)add, add_ld, (float*)dx, dx_ld);
    MIS_LAUNCH_CHECK("gn_bwd_apply");
    return MIS_OK;
}

// =========================================================================================================
// First 3-D layer: x fp32 [N][D][H][W] (one channel), GroupNorm(1 group) folded to a per-sample affine, Conv3d 1 -> Cout
// (Cout <= 64, multiple of 8) 3x3x3 p1 no bias, ReLU.  reference: encoders.0 SingleConv1 (buildingblocks.py:202-211 with in=1).
// 8 output channels per lane, Cout/8 lanes per voxel.  Output buffer may be wider than Cout: padding channels get 0.
// =========================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void first3d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                          int sstride, int N, int D, int H, int W, const float* __restrict__ w /*[Cout][27]*/, int Cout,
                                                          T* y, int y_ld, int Cpad) {
    __shared__ float wl[27 * 64];
    const int tid = threadIdx.x;
    for (int i = tid; i < 27 * 64; i += 256) {
        const int co = i & 63, tap = i >> 6;
        wl[i] = co < Cout ? w[co * 27 + tap] : 0.f;
    }
    __syncthreads();
    const int lpv = Cpad / 8;                 // lanes per voxel (4 or 8)
    const int vpb = 256 / lpv;
    const int cg = tid % lpv;
    const long long DHW = (long long)D * H * W, total = (long long)N * DHW;
    for (long long v = (long long)blockIdx.x * vpb + tid / lpv; v < total; v += (long long)gridDim.x * vpb) {
        const int n = (int)(v / DHW);
        const long long rem = v - (long long)n * DHW;
        const int xx = (int)(rem % W);
        const long long t = rem / W;
        const int yy = (int)(t % H), zz = (int)(t / H);
        const float a = scale[n * sstride], b = shift[n * sstride];
        const float* xp = x + (long long)n * DHW;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int sz = zz + tap / 9 - 1, sy = yy + (tap / 3) % 3 - 1, sx = xx + tap % 3 - 1;
            float xv = 0.f;
            if (sz >= 0 && sz < D && sy >= 0 && sy < H && sx >= 0 && sx < W) xv = fmaf(xp[((long long)sz * H + sy) * W + sx], a, b);
            const float* wr = &wl[tap * 64 + cg * 8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv, wr[j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
        T* dst = y + (size_t)v * y_ld + cg * 8;
        if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<u32x4*>(dst) = pack_chunk<__bf16>(acc);
        } else {
            *reinterpret_cast<u32x4*>(dst) = pack_chunk<float>(acc);
            *reinterpret_cast<u32x4*>(dst + 4) = pack_chunk<float>(acc + 4);
        }
    }
}

// Tiled variant for Cout = 32 (the reference's default first layer): one block = a 1 x 4 x 64 voxel tile whose normalised input halo
// (3 x 6 x 66 floats, zero outside the volume) is staged once in LDS; a thread owns 4 consecutive voxels x 8 channels, so one row of 6
// halo values feeds 3 taps x 4 voxels and every weight vector read from LDS is used 4 times: 864 FMAs per thread against ~60 LDS reads,
// no per-tap global loads or bounds checks.  Padding channels [32, Cpad) are written as zeros.
template <typename T>
__global__ __launch_bounds__(256) void first3d_fwd_tiled_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                                int sstride, int N, int D, int H, int W, const float* __restrict__ w /*[32][27]*/, T* y,
                                                                int y_ld, int Cpad) {
    constexpr int TH = 4, TW = 64, HH = TH + 2, HW = TW + 2, CO = 32;
    __shared__ __attribute__((aligned(16))) float wl[27 * CO];
    __shared__ __attribute__((aligned(16))) float xs[3][HH][HW + 2];      // +2: rows stay 16-byte aligned (68 floats)
    const int tid = threadIdx.x;
    for (int i = tid; i < 27 * CO; i += 256) wl[i] = w[(i % CO) * 27 + i / CO];       // [tap][co]
    const int tilesW = (W + TW - 1) / TW, tilesH = (H + TH - 1) / TH;
    long long b = blockIdx.x;
    const int tw = (int)(b % tilesW);
    b /= tilesW;
    const int th = (int)(b % tilesH);
    b /= tilesH;
    const int z = (int)(b % D);
    const int n = (int)(b / D);
    const int h0 = th * TH, w0 = tw * TW;
    const float a = scale[n * sstride], sh = shift[n * sstride];
    const float* xp = x + (long long)n * D * H * W;
    for (int i = tid; i < 3 * HH * HW; i += 256) {
        const int px = i % HW, r = i / HW;
        const int py = r % HH, pz = r / HH;
        const int sz = z + pz - 1, sy = h0 + py - 1, sx = w0 + px - 1;
        float v = 0.f;
        if (sz >= 0 && sz < D && sy >= 0 && sy < H && sx >= 0 && sx < W) v = fmaf(xp[((long long)sz * H + sy) * W + sx], a, sh);
        xs[pz][py][px] = v;
    }
    __syncthreads();
    const int cg = tid & 3, gq = tid >> 2;          // 4 channel groups of 8, 64 voxel groups
    const int r = gq >> 4, wq = (gq & 15) * 4;      // tile row, first of 4 voxels
    float acc[4][8];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[v][j] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            float xr[6];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(&xs[kd][r + kh][wq]);
            xr[0] = lo[0]; xr[1] = lo[1]; xr[2] = lo[2]; xr[3] = lo[3];
            xr[4] = xs[kd][r + kh][wq + 4];
            xr[5] = xs[kd][r + kh][wq + 5];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float* wr = &wl[((kd * 3 + kh) * 3 + kw) * CO + cg * 8];
                const f32x4 w0v = *reinterpret_cast<const f32x4*>(wr), w1v = *reinterpret_cast<const f32x4*>(wr + 4);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float xv = xr[v + kw];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[v][j] = fmaf(xv, w0v[j], acc[v][j]);
                        acc[v][4 + j] = fmaf(xv, w1v[j], acc[v][4 + j]);
                    }
                }
            }
        }
    const int yy = h0 + r;
    if (yy < H) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int xx = w0 + wq + v;
            if (xx >= W) break;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[v][j] = fmaxf(acc[v][j], 0.f);
            T* dst = y + ((((size_t)n * D + z) * H + yy) * W + xx) * y_ld + cg * 8;
            float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<u32x4*>(dst) = pack_chunk<__bf16>(acc[v]);
                if (Cpad > CO) *reinterpret_cast<u32x4*>(dst + CO) = pack_chunk<__bf16>(zero);
            } else {
                *reinterpret_cast<u32x4*>(dst) = pack_chunk<float>(acc[v]);
                *reinterpret_cast<u32x4*>(dst + 4) = pack_chunk<float>(acc[v] + 4);
                if (Cpad > CO) {
                    *reinterpret_cast<u32x4*>(dst + CO) = pack_chunk<float>(zero);
                    *reinterpret_cast<u32x4*>(dst + CO + 4) = pack_chunk<float>(zero);
                }
            }
        }
    }
}

// MFMA variant of the tiled kernel (end of round 3): the same 1 x 4 x 64 voxel tile and LDS halo, but the 27 x 32 multiply-adds of a voxel run on the matrix pipe in full
// fp32 (v_mfma_f32_16x16x4_f32: D[co][voxel] += W[co][tap] * X[tap][voxel], seven k-steps of four taps, tap 27 = a zero weight).  The tiled kernel is bound by its 864
// vector FMAs per thread (0.84 ms at 2 x 160^3 for 0.5 GB of output); here a wave keeps the whole filter in 14 registers (lane (co, tap group)), reads ONE halo value
// per lane and k-step and issues 14 MFMAs per 16 voxels.  Arithmetic: exact fp32 products, fp32 accumulation in a different (fixed) order than the fmaf chain.
template <typename T>
__global__ __launch_bounds__(256) void first3d_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                               int sstride, int N, int D, int H, int W, const float* __restrict__ w /*[32][27]*/, T* y,
                                                               int y_ld, int Cpad) {
    constexpr int TH = 4, TW = 64, HH = TH + 2, HW = TW + 2, XS = HW + 2, CO = 32;
    __shared__ __attribute__((aligned(16))) float xs[3][HH][XS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int tilesW = (W + TW - 1) / TW, tilesH = (H + TH - 1) / TH;
    long long b = blockIdx.x;
    const int tw = (int)(b % tilesW);
    b /= tilesW;
    const int th = (int)(b % tilesH);
    b /= tilesH;
    const int z = (int)(b % D);
    const int n = (int)(b / D);
    const int h0 = th * TH, w0 = tw * TW;
    const float a = scale[n * sstride], sh = shift[n * sstride];
    const float* xp = x + (long long)n * D * H * W;
    for (int i = tid; i < 3 * HH * HW; i += 256) {
        const int px = i % HW, r = i / HW;
        const int py = r % HH, pz = r / HH;
        const int sz = z + pz - 1, sy = h0 + py - 1, sx = w0 + px - 1;
        float v = 0.f;
        if (sz >= 0 && sz < D && sy >= 0 && sy < H && sx >= 0 && sx < W) v = fmaf(xp[((long long)sz * H + sy) * W + sx], a, sh);
        xs[pz][py][px] = v;
    }
    // A operand (weights): lane (i = li, k = lg) of k-step ks and fragment f holds W[co = f*16 + li][tap = 4*ks + lg] (0 for tap 27)
    float wa[7][2];
    int toff[7];           // byte offset of tap 4*ks + lg inside the halo (tap 27 reads tap 26's value: its weight is zero)
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
        const int t = 4 * ks + lg;
        const int tt = t < 27 ? t : 26;
#pragma unroll
        for (int f = 0; f < 2; ++f) wa[ks][f] = t < 27 ? w[(f * 16 + li) * 27 + t] : 0.f;
        toff[ks] = (((tt / 9) * HH + (tt / 3) % 3) * XS + tt % 3) * 4;
    }
    __syncthreads();
    const int r = wave;                                   // tile row of this wave; its 64 voxels = four groups of 16 along x
    const char* const xb = reinterpret_cast<const char*>(&xs[0][0][0]) + (r * XS + li) * 4;
    const int yy = h0 + r;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        float bv[7];
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) bv[ks] = *reinterpret_cast<const float*>(xb + toff[ks] + g * 64);      // X[tap][voxel g*16 + li]
#pragma unroll
        for (int ks = 0; ks < 7; ++ks)
#pragma unroll
            for (int f = 0; f < 2; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ks][f], bv[ks], acc[f], 0, 0, 0);
        // D: lane (li, lg), register q = output channel f*16 + 4*lg + q of voxel g*16 + li
        const int xx = w0 + g * 16 + li;
        if (yy < H && xx < W) {
            T* dst = y + ((((size_t)n * D + z) * H + yy) * W + xx) * y_ld + 4 * lg;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = fmaxf(acc[f][q], 0.f);
                if constexpr (sizeof(T) == 2) {
                    const u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                    *reinterpret_cast<u32x2*>(dst + f * 16) = pk;
                    if (Cpad > CO) *reinterpret_cast<u32x2*>(dst + CO + f * 16) = u32x2{0u, 0u};
                } else {
                    *reinterpret_cast<f32x4*>(dst + f * 16) = f32x4{o[0], o[1], o[2], o[3]};
                    if (Cpad > CO) *reinterpret_cast<f32x4*>(dst + CO + f * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }
}

extern "C" int mis_first3d_fwd(int dtype, const float* x, const float* scale, const float* shift, int sstride, int N, int D, int H, int W,
                               const float* w, int Cout, void* y, int y_ld, int Cpad, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(x && scale && shift && w && y, MIS_EINVAL, "first3d_fwd: null pointer");
    MIS_REQUIRE(Cout > 0 && Cout <= Cpad && (Cpad == 32 || Cpad == 64) && y_ld >= Cpad && y_ld % 8 == 0, MIS_EUNSUPPORTED,
                "first3d_fwd: Cout %d / Cpad %d", Cout, Cpad);
    const long long total = (long long)N * D * H * W;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int tiled = !mis_sw(SW_FIRST3D_UNTILED);
    const long long tiles = (long long)N * D * ((H + 3) / 4) * ((W + 63) / 64);
    if (tiled && Cout == 32 && tiles < (1ll << 31) && !mis_sw(SW_FIRST3D_NOMFMA)) {
        if (dtype == MIS_BF16)
            hipLaunchKernelGGL(first3d_fwd_mfma_kernel<__bf16>, dim3((unsigned)tiles), dim3(256), 0, s, x, scale, shift, sstride, N, D, H, W, w, (__bf16*)y, y_ld, Cpad);
        else
            hipLaunchKernelGGL(first3d_fwd_mfma_kernel<float>, dim3((unsigned)tiles), dim3(256), 0, s, x, scale, shift, sstride, N, D, H, W, w, (float*)y, y_ld, Cpad);
        MIS_LAUNCH_CHECK("first3d_fwd(mfma)");
        return MIS_OK;
    }
    if (tiled && Cout == 32 && tiles < (1ll << 31)) {
        if (dtype == MIS_BF16)
            hipLaunchKernelGGL(first3d_fwd_tiled_kernel<__bf16>, dim3((unsigned)tiles), dim3(256), 0, s, x, scale, shift, sstride, N, D, H, W, w, (__bf16*)y,
                               y_ld, Cpad);
        else
            hipLaunchKernelGGL(first3d_fwd_tiled_kernel<float>, dim3((unsigned)tiles), dim3(256), 0, s, x, scale, shift, sstride, N, D, H, W, w, (float*)y, y_ld,
                               Cpad);
        MIS_LAUNCH_CHECK("first3d_fwd(tiled)");
        return MIS_OK;
    }
    const int vpb = 256 / (Cpad / 8);
    long long blocks = (total + vpb - 1) / vpb;
    if (blocks > 8192) blocks = 8192;
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(first3d_fwd_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, x, scale, shift, sstride, N, D, H, W, w, Cout, (__bf16*)y,
                           y_ld, Cpad);
    else
        hipLaunchKernelGGL(first3d_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, x, scale, shift, sstride, N, D, H, W, w, Cout, (float*)y, y_ld,
                           Cpad);
    MIS_LAUNCH_CHECK("first3d_fwd");
    return MIS_OK;
}

// backward of the first 3-D layer given dy = dL/d(pre-activation) [voxel][Cpad].  With xn = gamma*xh + beta inside the volume (xh = the
// per-sample standardised input) and 0 outside, everything the layer needs follows from two correlation sums per (co, tap):
//   G[co][tap] = sum_v xh[v+tap] * dy[v][co],   S[co][tap] = sum_v [v+tap inside] * dy[v][co]           blockIdx.y = kd slab (9 taps)
//   dW = gamma*G + beta*S,   dgamma = sum_{co,tap} W*G,   dbeta = sum_{co,tap} W*S
// (dgamma/dbeta are sum_v dxn*xh and sum_v dxn with dxn = the transposed conv of dy - which therefore never has to be formed;
//  the optional first3d_dgrad_kernel below still produces it for callers that want dL/d(normalised input) itself).
constexpr int F3_BLOCKS = 512;
template <typename T>
__global__ __launch_bounds__(256) void first3d_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            int N, int D, int H, int W, const T* __restrict__ dy, int dy_ld, int Cpad,
                                                            float* __restrict__ partial /*[blocks][3][2][9][64]*/) {
    __shared__ float red[4][9 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lpv = Cpad / 8, vpb = 256 / lpv, cg = tid % lpv;
    const int kd = blockIdx.y;
    const long long DHW = (long long)D * H * W, total = (long long)N * DHW;
    float accG[9][8], accS[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) accG[t][j] = accS[t][j] = 0.f;
    for (long long v = (long long)blockIdx.x * vpb + tid / lpv; v < total; v += (long long)gridDim.x * vpb) {
        const int n = (int)(v / DHW);
        const long long rem = v - (long long)n * DHW;
        const int xx = (int)(rem % W);
        const long long t = rem / W;
        const int yy = (int)(t % H), zz = (int)(t / H);
        const float a = rstd[n], b = -mean[n] * rstd[n];
        const float* xp = x + (long long)n * DHW;
        float g[8];
        const T* src = dy + (size_t)v * dy_ld + cg * 8;
        if constexpr (sizeof(T) == 2) {
            unpack_chunk<__bf16>(*reinterpret_cast<const u32x4*>(src), g);
        } else {
            unpack_chunk<float>(*reinterpret_cast<const u32x4*>(src), g);
            unpack_chunk<float>(*reinterpret_cast<const u32x4*>(src + 4), g + 4);
        }
        const int sz = zz + kd - 1;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int sy = yy + tp / 3 - 1, sx = xx + tp % 3 - 1;
            float xv = 0.f, in = 0.f;
            if (sz >= 0 && sz < D && sy >= 0 && sy < H && sx >= 0 && sx < W) {
                xv = fmaf(xp[((long long)sz * H + sy) * W + sx], a, b);
                in = 1.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                accG[tp][j] = fmaf(xv, g[j], accG[tp][j]);
                accS[tp][j] = fmaf(in, g[j], accS[tp][j]);
            }
        }
    }
    // lanes with equal (lane % lpv) hold the same channels
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        float (*acc)[8] = which == 0 ? accG : accS;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float vsum = acc[t][j];
                for (int o = lpv; o < 64; o <<= 1) vsum += __shfl_xor(vsum, o, 64);
                acc[t][j] = vsum;
            }
        __syncthreads();
        for (int i = tid; i < 4 * 9 * 64; i += 256) (&red[0][0])[i] = 0.f;
        __syncthreads();
        if (lane < lpv) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[wave][t * 64 + lane * 8 + j] = acc[t][j];
        }
        __syncthreads();
        float* out = partial + (((size_t)blockIdx.x * 3 + kd) * 2 + which) * 576;
        for (int i = tid; i < 576; i += 256) out[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
}

// Tiled variant for Cout = 32: a persistent block walks 1 x 4 x 64 voxel tiles; the standardised input row halo (6 x 66 values of the
// plane z + kd - 1, plus the matching inside-the-volume flags) is staged in LDS, a thread owns 4 consecutive voxels x 8 channels of dy,
// so one halo row of 6 values feeds 3 taps x 4 voxels: no per-tap global loads or bounds checks.  Same partial layout as above.
template <typename T>
__global__ __launch_bounds__(256) void first3d_wgrad_tiled_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  int N, int D, int H, int W, const T* __restrict__ dy, int dy_ld,
                                                                  float* __restrict__ partial /*[blocks][3][2][9][64]*/) {
    constexpr int TH = 4, TW = 64, HH = TH + 2, HW = TW + 2;
    __shared__ __attribute__((aligned(16))) float xs[HH][HW + 2], ins[HH][HW + 2];
    __shared__ float red[4][9 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kd = blockIdx.y;
    const int cg = tid & 3, gq = tid >> 2;
    const int r = gq >> 4, wq = (gq & 15) * 4;
    const int tilesW = (W + TW - 1) / TW, tilesH = (H + TH - 1) / TH;
    const long long ntiles = (long long)N * D * tilesH * tilesW;
    float accG[9][8], accS[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) accG[t][j] = accS[t][j] = 0.f;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        long long b = tile;
        const int tw = (int)(b % tilesW);
        b /= tilesW;
        const int th = (int)(b % tilesH);
        b /= tilesH;
        const int z = (int)(b % D);
        const int n = (int)(b / D);
        const int sz = z + kd - 1;
        if (sz < 0 || sz >= D) continue;                 // block-uniform: this tap plane lies outside the volume
        const int h0 = th * TH, w0 = tw * TW;
        const float a = rstd[n], sh = -mean[n] * rstd[n];
        const float* xp = x + ((long long)n * D + sz) * H * W;
        __syncthreads();                                 // the previous tile's readers are done
        for (int i = tid; i < HH * HW; i += 256) {
            const int px = i % HW, py = i / HW;
            const int sy = h0 + py - 1, sx = w0 + px - 1;
            const bool in = sy >= 0 && sy < H && sx >= 0 && sx < W;
            xs[py][px] = in ? fmaf(xp[(long long)sy * W + sx], a, sh) : 0.f;
            ins[py][px] = in ? 1.f : 0.f;
        }
        float g[4][8];
        const int yy = h0 + r;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int xx = w0 + wq + v;
#pragma unroll
            for (int j = 0; j < 8; ++j) g[v][j] = 0.f;
            if (yy < H && xx < W) {
                const T* src = dy + ((((size_t)n * D + z) * H + yy) * W + xx) * dy_ld + cg * 8;
                if constexpr (sizeof(T) == 2) {
                    unpack_chunk<__bf16>(*reinterpret_cast<const u32x4*>(src), g[v]);
                } else {
                    unpack_chunk<float>(*reinterpret_cast<const u32x4*>(src), g[v]);
                    unpack_chunk<float>(*reinterpret_cast<const u32x4*>(src + 4), g[v] + 4);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            float xr[6], ir[6];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(&xs[r + kh][wq]);
            const f32x4 li4 = *reinterpret_cast<const f32x4*>(&ins[r + kh][wq]);
            xr[0] = lo[0]; xr[1] = lo[1]; xr[2] = lo[2]; xr[3] = lo[3];
            ir[0] = li4[0]; ir[1] = li4[1]; ir[2] = li4[2]; ir[3] = li4[3];
            xr[4] = xs[r + kh][wq + 4]; xr[5] = xs[r + kh][wq + 5];
            ir[4] = ins[r + kh][wq + 4]; ir[5] = ins[r + kh][wq + 5];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        accG[kh * 3 + kw][j] = fmaf(xr[v + kw], g[v][j], accG[kh * 3 + kw][j]);
                        accS[kh * 3 + kw][j] = fmaf(ir[v + kw], g[v][j], accS[kh * 3 + kw][j]);
                    }
        }
    }
    // lanes with equal (lane & 3) hold the same channels
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        float (*acc)[8] = which == 0 ? accG : accS;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float vsum = acc[t][j];
                for (int o = 4; o < 64; o <<= 1) vsum += __shfl_xor(vsum, o, 64);
                acc[t][j] = vsum;
            }
        __syncthreads();
        for (int i = tid; i < 4 * 9 * 64; i += 256) (&red[0][0])[i] = 0.f;
        __syncthreads();
        if (lane < 4) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[wave][t * 64 + lane * 8 + j] = acc[t][j];
        }
        __syncthreads();
        float* out = partial + (((size_t)blockIdx.x * 3 + kd) * 2 + which) * 576;
        for (int i = tid; i < 576; i += 256) out[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
}

// MFMA variant of the tiled kernel (end of round 3; Cout = 32): G[co][tap] = sum_v dY[v][co] * xh[v + tap] and S (the same with the inside-the-volume flags) are GEMMs with
// K = voxels - D[co][tap] += A[co][voxel] * B[voxel][tap] in full fp32 on the matrix pipe (v_mfma_f32_16x16x4_f32), all 27 taps of a tile at once (two 16-column
// fragments, columns 27-31 unused).  A persistent block stages a 1 x 4 x 64 voxel tile: the standardised input halo of the three planes z - 1 .. z + 1 and its flags
// (3 x 6 x 66 each) and dY as fp32 [voxel][32 co] (33-word rows: conflict-free); a wave owns one tile row = 16 k-steps of 4 voxels, per k-step two dY reads, four halo /
// flag reads and eight MFMAs.  The tiled kernel needs 2 x 216 vector FMAs per voxel and thread group (0.96 ms at 2 x 160^3); same partial layout, same reduction.
template <typename T>
__global__ __launch_bounds__(256) void first3d_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 int N, int D, int H, int W, const T* __restrict__ dy, int dy_ld,
                                                                 float* __restrict__ partial /*[blocks][3][2][9][64]*/) {
    constexpr int TH = 4, TW = 64, HH = TH + 2, HW = TW + 2, XS = HW + 2, DS = 33;
    __shared__ __attribute__((aligned(16))) float xs[3][HH][XS], ins[3][HH][XS];
    __shared__ __attribute__((aligned(16))) float dys[256 * DS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int cg = tid & 3, gq = tid >> 2;
    const int sr = gq >> 4, wq = (gq & 15) * 4;          // staging: this thread's tile row and first of 4 voxels
    const int tilesW = (W + TW - 1) / TW, tilesH = (H + TH - 1) / TH;
    const long long ntiles = (long long)N * D * tilesH * tilesW;
    // B operand addresses: lane (k = lg, j = li) of tap fragment b reads halo value [kd][row + kh][4*ks + lg + kw] of tap t = b*16 + li (t > 26: tap 26 again, unused)
    int boff[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int t = b * 16 + li, tt = t < 27 ? t : 26;
        boff[b] = ((tt / 9) * HH + (tt / 3) % 3 + wave) * XS + tt % 3 + lg;
    }
    const int aoff = (wave * 64 + lg) * DS + li;          // A operand: lane (i = li, k = lg) reads dys[voxel wave*64 + 4*ks + lg][f*16 + li]
    f32x4 accG[2][2], accS[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int b = 0; b < 2; ++b) accG[f][b] = accS[f][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        long long bb = tile;
        const int tw = (int)(bb % tilesW);
        bb /= tilesW;
        const int th = (int)(bb % tilesH);
        bb /= tilesH;
        const int z = (int)(bb % D);
        const int n = (int)(bb / D);
        const int h0 = th * TH, w0 = tw * TW;
        const float a = rstd[n], sh = -mean[n] * rstd[n];
        const float* xp = x + (long long)n * D * H * W;
        __syncthreads();                                 // the previous tile's readers are done
        for (int i = tid; i < 3 * HH * HW; i += 256) {
            const int px = i % HW, r = i / HW;
            const int py = r % HH, pz = r / HH;
            const int sz = z + pz - 1, sy = h0 + py - 1, sx = w0 + px - 1;
            const bool in = sz >= 0 && sz < D && sy >= 0 && sy < H && sx >= 0 && sx < W;
            xs[pz][py][px] = in ? fmaf(xp[((long long)sz * H + sy) * W + sx], a, sh) : 0.f;
            ins[pz][py][px] = in ? 1.f : 0.f;
        }
        {
            const int yy = h0 + sr;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int xx = w0 + wq + v;
                float g[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = 0.f;
                if (yy < H && xx < W) {
                    const T* src = dy + ((((size_t)n * D + z) * H + yy) * W + xx) * dy_ld + cg * 8;
                    if constexpr (sizeof(T) == 2) {
                        unpack_chunk<__bf16>(*reinterpret_cast<const u32x4*>(src), g);
                    } else {
                        unpack_chunk<float>(*reinterpret_cast<const u32x4*>(src), g);
                        unpack_chunk<float>(*reinterpret_cast<const u32x4*>(src + 4), g + 4);
                    }
                }
                float* dst = &dys[(sr * 64 + wq + v) * DS + cg * 8];
#pragma unroll
                for (int j = 0; j < 8; ++j) dst[j] = g[j];
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const float a0 = dys[aoff + ks * 4 * DS], a1 = dys[aoff + ks * 4 * DS + 16];
            const float* xf = &xs[0][0][0];
            const float* nf = &ins[0][0][0];
            const float g0 = xf[boff[0] + 4 * ks], g1 = xf[boff[1] + 4 * ks], s0 = nf[boff[0] + 4 * ks], s1 = nf[boff[1] + 4 * ks];
            accG[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, g0, accG[0][0], 0, 0, 0);
            accG[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, g1, accG[0][1], 0, 0, 0);
            accG[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, g0, accG[1][0], 0, 0, 0);
            accG[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, g1, accG[1][1], 0, 0, 0);
            accS[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, s0, accS[0][0], 0, 0, 0);
            accS[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, s1, accS[0][1], 0, 0, 0);
            accS[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, s0, accS[1][0], 0, 0, 0);
            accS[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, s1, accS[1][1], 0, 0, 0);
        }
    }
    // D: lane (li, lg), register q = [co = f*16 + 4*lg + q][tap = b*16 + li].  Per wave -> LDS, summed over the four waves in a fixed order, written in the partial layout
    __syncthreads();
    float* red = dys;                                    // [wave][which][tap 27][co 32]
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int t = b * 16 + li;
            if (t < 27) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    red[((wave * 2 + 0) * 27 + t) * 32 + f * 16 + 4 * lg + q] = accG[f][b][q];
                    red[((wave * 2 + 1) * 27 + t) * 32 + f * 16 + 4 * lg + q] = accS[f][b][q];
                }
            }
        }
    __syncthreads();
    for (int i = tid; i < 3 * 2 * 576; i += 256) {
        const int co = i & 63, tp = (i >> 6) % 9, which = (i / 576) & 1, kd = i / 1152;
        float v = 0.f;
        if (co < 32) {
            const int t = kd * 9 + tp;
            for (int wv = 0; wv < 4; ++wv) v += red[((wv * 2 + which) * 27 + t) * 32 + co];
        }
        partial[(((size_t)blockIdx.x * 3 + kd) * 2 + which) * 576 + tp * 64 + co] = v;
    }
}

// dW = gamma*G + beta*S in the reference layout; G and S totals go to gs[2][3*576] for the finalize kernel
__global__ void first3d_wgrad_reduce_kernel(const float* __restrict__ partial, int nblocks, int Cout, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, float* __restrict__ dw, float* __restrict__ gs) {
    // 16 lanes per (kd, tp, co): each sums every 16th block partial in a fixed order, then four shuffles (one lane per output walked all the partials by itself: 207 us)
    const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
    const int idx = gidx >> 4, part = gidx & 15;
    const bool live = idx < 3 * 576;
    const int kd = live ? idx / 576 : 0, r = live ? idx - kd * 576 : 0;
    const int tp = r >> 6, co = r & 63;
    float G = 0.f, S = 0.f;
    if (live && co < Cout) {
        for (int b = part; b < nblocks; b += 16) {
            G += partial[(((size_t)b * 3 + kd) * 2 + 0) * 576 + r];
            S += partial[(((size_t)b * 3 + kd) * 2 + 1) * 576 + r];
        }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        G += __shfl_xor(G, o, 64);
        S += __shfl_xor(S, o, 64);
    }
    if (!live || part != 0) return;
    if (co < Cout) dw[co * 27 + kd * 9 + tp] = fmaf(gamma[0], G, beta[0] * S);
    gs[idx] = G;
    gs[3 * 576 + idx] = S;
}

__global__ __launch_bounds__(256) void first3d_gn_finalize_kernel(const float* __restrict__ gs, const float* __restrict__ w /*[Cout][27]*/, int Cout,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double red[2][4];
    double dg = 0.0, db = 0.0;
    for (int idx = threadIdx.x; idx < 3 * 576; idx += 256) {
        const int kd = idx / 576, r = idx - kd * 576;
        const int tp = r >> 6, co = r & 63;
        if (co < Cout) {
            const double wv = w[co * 27 + kd * 9 + tp];
            dg += wv * (double)gs[idx];
            db += wv * (double)gs[3 * 576 + idx];
        }
    }
    dg = wave_sum_d(dg);
    db = wave_sum_d(db);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = dg;
        red[1][threadIdx.x >> 6] = db;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        dgamma[0] = (float)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        dbeta[0] = (float)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void first3d_dgrad_kernel(const T* __restrict__ dy, int dy_ld, int Cpad, int N, int D, int H, int W,
                                                            const float* __restrict__ w /*[Cout][27]*/, int Cout, float* __restrict__ dxn) {
    constexpr int EPC = Tr<T>::EPC;
    __shared__ float wl[27 * 64];
    const int tid = threadIdx.x;
    for (int i = tid; i < 27 * 64; i += 256) {
        const int co = i & 63, tap = i >> 6;
        wl[i] = co < Cout ? w[co * 27 + tap] : 0.f;
    }
    __syncthreads();
    const long long DHW = (long long)D * H * W, total = (long long)N * DHW;
    for (long long v = (long long)blockIdx.x * 256 + tid; v < total; v += (long long)gridDim.x * 256) {
        const int n = (int)(v / DHW);
        const long long rem = v - (long long)n * DHW;
        const int xx = (int)(rem % W);
        const long long t = rem / W;
        const int yy = (int)(t % H), zz = (int)(t / H);
        float s = 0.f;
        for (int tap = 0; tap < 27; ++tap) {
            // y[v'] used x[v' + tap - 1]  =>  x[v] contributes to y[v - (tap - 1)]
            const int sz = zz - (tap / 9 - 1), sy = yy - ((tap / 3) % 3 - 1), sx = xx - (tap % 3 - 1);
            if (sz < 0 || sz >= D || sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
            const T* src = dy + ((size_t)n * DHW + ((size_t)sz * H + sy) * W + sx) * dy_ld;
            for (int c = 0; c < Cout; c += EPC) {
                float g[EPC];
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(src + c), g);
#pragma unroll
                for (int e = 0; e < EPC; ++e) s = fmaf(g[e], wl[tap * 64 + c + e], s);
            }
        }
        dxn[v] = s;
    }
}

extern "C" size_t mis_first3d_bwd_workspace_bytes(void) { return ((size_t)F3_BLOCKS * 3 * 2 * 576 + 2 * 3 * 576) * sizeof(float); }

extern "C" int mis_first3d_bwd(int dtype, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int N, int D,
                               int H, int W, const void* dy, int dy_ld, int Cpad, const float* w, int Cout, float* workspace, float* dw,
                               float* dgamma, float* dbeta, float* dxn, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(x && mean && rstd && gamma && beta && dy && w && workspace && dw && dgamma && dbeta, MIS_EINVAL, "first3d_bwd: null pointer");
    MIS_REQUIRE(Cout > 0 && Cout <= Cpad && Cout % 8 == 0 && (Cpad == 32 || Cpad == 64) && dy_ld % 8 == 0, MIS_EUNSUPPORTED, "first3d_bwd: sizes");
    const long long total = (long long)N * D * H * W;
    const int vpb = 256 / (Cpad / 8);
    long long blocks = (total + vpb - 1) / vpb;
    if (blocks > F3_BLOCKS) blocks = F3_BLOCKS;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* gs = workspace + (size_t)F3_BLOCKS * 3 * 2 * 576;
    const int tiled = !mis_sw(SW_FIRST3D_UNTILED);
    if (tiled && Cout == 32 && !mis_sw(SW_FIRST3D_NOMFMA)) {
        const long long tiles = (long long)N * D * ((H + 3) / 4) * ((W + 63) / 64);
        blocks = tiles < F3_BLOCKS ? tiles : F3_BLOCKS;
        if (dtype == MIS_BF16)
            hipLaunchKernelGGL(first3d_wgrad_mfma_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, x, mean, rstd, N, D, H, W, (const __bf16*)dy, dy_ld, workspace);
        else
            hipLaunchKernelGGL(first3d_wgrad_mfma_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, x, mean, rstd, N, D, H, W, (const float*)dy, dy_ld, workspace);
    } else if (tiled && Cout == 32) {
        const long long tiles = (long long)N * D * ((H + 3) / 4) * ((W + 63) / 64);
        blocks = tiles < F3_BLOCKS ? tiles : F3_BLOCKS;
        if (dtype == MIS_BF16)
            hipLaunchKernelGGL(first3d_wgrad_tiled_kernel<__bf16>, dim3((unsigned)blocks, 3), dim3(256), 0, s, x, mean, rstd, N, D, H, W, (const __bf16*)dy,
                               dy_ld, workspace);
        else
            hipLaunchKernelGGL(first3d_wgrad_tiled_kernel<float>, dim3((unsigned)blocks, 3), dim3(256), 0, s, x, mean, rstd, N, D, H, W, (const float*)dy,
                               dy_ld, workspace);
    } else if (dtype == MIS_BF16)
        hipLaunchKernelGGL(first3d_wgrad_kernel<__bf16>, dim3((unsigned)blocks, 3), dim3(256), 0, s, x, mean, rstd, N, D, H, W, (const __bf16*)dy, dy_ld,
                           Cpad, workspace);
    else
        hipLaunchKernelGGL(first3d_wgrad_kernel<float>, dim3((unsigned)blocks, 3), dim3(256), 0, s, x, mean, rstd, N, D, H, W, (const float*)dy, dy_ld,
                           Cpad, workspace);
    MIS_LAUNCH_CHECK("first3d_wgrad");
    hipLaunchKernelGGL(first3d_wgrad_reduce_kernel, dim3((3 * 576 * 16 + 255) / 256), dim3(256), 0, s, (const float*)workspace, (int)blocks, Cout, gamma,
                       beta, dw, gs);
    MIS_LAUNCH_CHECK("first3d_wgrad_reduce");
    hipLaunchKernelGGL(first3d_gn_finalize_kernel, dim3(1), dim3(256), 0, s, (const float*)gs, w, Cout, dgamma, dbeta);
    MIS_LAUNCH_CHECK("first3d_gn_finalize");
    if (dxn != nullptr) {       // optional: dL/d(normalised input) itself
        long long dblocks = (total + 255) / 256;
        if (dblocks > 8192) dblocks = 8192;
        if (dtype == MIS_BF16)
            hipLaunchKernelGGL(first3d_dgrad_kernel<__bf16>, dim3((unsigned)dblocks), dim3(256), 0, s, (const __bf16*)dy, dy_ld, Cpad, N, D, H, W, w, Cout,
                               dxn);
        else
            hipLaunchKernelGGL(first3d_dgrad_kernel<float>, dim3((unsigned)dblocks), dim3(256), 0, s, (const float*)dy, dy_ld, Cpad, N, D, H, W, w, Cout,
                               dxn);
        MIS_LAUNCH_CHECK("first3d_dgrad");
    }
    return MIS_OK;
}

// elementwise ReLU-mask: dx = dy * (y > 0)  (used where the mask cannot ride in a conv epilogue)
template <typename T>
__global__ void relu_mask_kernel(const T* __restrict__ dy, int dy_ld, const T* __restrict__ y, int y_ld, T* __restrict__ dx, int dx_ld, long long npix,
                                 int C) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC;
    const long long total = npix * nch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % nch);
        const long long p = i / nch;
        float g[EPC], f[EPC];
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(dy + p * dy_ld + (size_t)ch * EPC), g);
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(y + p * y_ld + (size_t)ch * EPC), f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) g[e] = f[e] > 0.f ? g[e] : 0.f;
        *reinterpret_cast<u32x4*>(dx + p * dx_ld + (size_t)ch * EPC) = pack_chunk<T>(g);
    }
}
extern "C" int mis_relu_mask(int dtype, const void* dy, int dy_ld, const void* y, int y_ld, void* dx, int dx_ld, long long npix, int C, void* stream) {
    (void)hipGetLastError();
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(dy && y && dx && npix > 0 && C % EPC == 0 && dy_ld % EPC == 0 && y_ld % EPC == 0 && dx_ld % EPC == 0, MIS_EINVAL, "relu_mask: bad argument");
    long long blocks = (npix * (C / EPC) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(relu_mask_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, (const __bf16*)dy, dy_ld, (const __bf16*)y, y_ld, (__bf16*)dx, dx_ld, npix, C);
    else
        hipLaunchKernelGGL(relu_mask_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)dy, dy_ld, (const float*)y, y_ld, (float*)dx, dx_ld, npix, C);
    MIS_LAUNCH_CHECK("relu_mask");
    return MIS_OK;
}
