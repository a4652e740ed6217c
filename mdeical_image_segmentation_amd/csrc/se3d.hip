// Concurrent spatial-and-channel squeeze & excitation after a residual 3-D block - the reference's `ResNetBlockSE` with se_module 'scse'
// (model/unet3d/buildingblocks.py:326-362, model/unet3d/se.py:18-116, reduction_ratio 1) for gfx950, channels-last (N, S = D*H*W, C):
//   cSE : m[n][c] = mean_v e ; a = sigmoid(W2 relu(W1 m + b1) + b2)                        (per sample, C x C matrices: mis_se_fc_fwd)
//   sSE : b[n][v] = sigmoid(sum_c w[c] e[n][v][c] + b0)                                    (1x1x1 conv to one channel)
//   out = max(e * a[n][c], e * b[n][v])                                                    (mis_se_apply_fwd: one pass, writes b)
// Backward (g = dL/dout, sA = [e*a >= e*b]):
//   da[n][c] = sum_v g sA e ;  dq[n][v] = b(1-b) sum_c g (1-sA) e ;  dw[c] = sum_{n,v} dq e ;  db0 = sum dq          (mis_se_bwd_reduce)
//   dz2 = da a(1-a), dW2 = dz2 h^T, dh = W2^T dz2, dz1 = dh [z1>0], dW1 = dz1 m^T, cross[n][c] = (W1^T dz1)[c] / S      (mis_se_fc_bwd)
//   de = [e>0] * ( g (sA a + (1-sA) b) + cross[n][c] + dq[n][v] w[c] )                     (mis_se_bwd_apply; e is a ReLU output: the mask is the
//        block's own ReLU backward, fused here).  Ties e*a == e*b occur where e == 0, which the mask removes, so torch's half/half rule is moot.
// Stand-alone layers (round 5: se.py:18-53 ChannelSELayer3D, :56-98 SpatialSELayer3D, :101-116 ChannelSpatialSELayer3D called on their own): the `mis_se_layer_*`
// entry points run the same three passes with `mode` = 0 max(cSE, sSE) / 1 cSE alone (out = e a) / 2 sSE alone (out = e b) and without the ReLU mask (`relu_mask` 0:
// the input may be negative; a tie e a == e b at e == 0 then takes torch.max's half / half rule, de = g (a + b) / 2 - the reductions carry the factor e and see nothing).
// The C x C matrices serve any reduction ratio: the caller pads W1 [C/r][C] / W2 [C][C/r] with zero rows / columns.
// HBM-bound passes: G = min(C/EPC, 64) lanes share a voxel (one 16-byte chunk per lane and step), channel sums of a voxel by xor-shuffles,
// per-channel sums over voxels in registers -> per-group partial slabs -> fixed-order second stage (bitwise reproducible).
#include "common.hpp"

constexpr int SE_MAXK = 4;          // chunks per lane: C <= 64 lanes * 4 chunks * EPC  (2048 bf16 / 1024 f32 channels)
constexpr int SE_PARTS = 512;       // blocks (partial slabs) per sample in the reducing backward pass

struct SeGeom {
    int nch, G, K, gpb;             // chunks per voxel, lanes per voxel, chunks per lane, voxel groups per 256-thread block
};
static inline SeGeom se_geom(int C, int EPC) {
    SeGeom g;
    g.nch = C / EPC;
    g.G = g.nch < 64 ? g.nch : 64;
    g.K = g.nch / g.G;
    g.gpb = 256 / g.G;
    return g;
}

__device__ __forceinline__ float se_group_sum(float v, int G) {
    for (int off = G >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float se_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// ---- cSE fully connected part: one block per sample -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void se_fc_fwd_kernel(const float* __restrict__ sum, float inv_count, const float* __restrict__ W1,
                                                        const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2, int C,
                                                        float* __restrict__ mean, float* __restrict__ z1, float* __restrict__ a) {
    extern __shared__ float sm[];          // m[C], h[C]
    float* m = sm;
    float* h = sm + C;
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        m[c] = sum[(size_t)n * C + c] * inv_count;
        mean[(size_t)n * C + c] = m[c];
    }
    __syncthreads();
    for (int r = threadIdx.x; r < C; r += 256) {
        float acc = b1[r];
        for (int c = 0; c < C; ++c) acc = fmaf(W1[(size_t)r * C + c], m[c], acc);
        z1[(size_t)n * C + r] = acc;
        h[r] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < C; r += 256) {
        float acc = b2[r];
        for (int c = 0; c < C; ++c) acc = fmaf(W2[(size_t)r * C + c], h[c], acc);
        a[(size_t)n * C + r] = 1.f / (1.f + expf(-acc));
    }
}

// gradients of the C x C matrices: grid = C rows; sums over the N samples in a fixed order.  dz2 / dz1 are recomputed per block (N*C values).
__global__ __launch_bounds__(256) void se_fc_bwd_kernel(const float* __restrict__ da, const float* __restrict__ a, const float* __restrict__ z1,
                                                        const float* __restrict__ mean, const float* __restrict__ W1, const float* __restrict__ W2, int N,
                                                        int C, float inv_count, float* __restrict__ dz1_ws, float* __restrict__ dW1,
                                                        float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2,
                                                        float* __restrict__ cross, int phase) {
    // phase 0: dz2 -> dW2, db2, dz1 (workspace);  phase 1: dW1, db1, cross
    const int r = blockIdx.x;
    if (phase == 0) {
        // row r of dW2 = sum_n dz2[n][r] * h[n][:] ;  dz1[n][r] = [z1>0] sum_i W2[i][r] dz2[n][i]
        float bsum = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) {
            float acc = 0.f;
            for (int n = 0; n < N; ++n) {
                const float av = a[(size_t)n * C + r];
                const float dz2 = da[(size_t)n * C + r] * av * (1.f - av);
                acc = fmaf(dz2, fmaxf(z1[(size_t)n * C + c], 0.f), acc);
            }
            dW2[(size_t)r * C + c] = acc;
        }
        if (threadIdx.x == 0) {
            for (int n = 0; n < N; ++n) {
                const float av = a[(size_t)n * C + r];
                bsum += da[(size_t)n * C + r] * av * (1.f - av);
            }
            db2[r] = bsum;
        }
        __shared__ float red[256];
        for (int n = 0; n < N; ++n) {
            float part = 0.f;
            for (int i = threadIdx.x; i < C; i += 256) {
                const float av = a[(size_t)n * C + i];
                part = fmaf(W2[(size_t)i * C + r], da[(size_t)n * C + i] * av * (1.f - av), part);
            }
            red[threadIdx.x] = part;
            __syncthreads();
            if (threadIdx.x == 0) {
                float s = 0.f;
                for (int k = 0; k < 256; ++k) s += red[k];
                dz1_ws[(size_t)n * C + r] = z1[(size_t)n * C + r] > 0.f ? s : 0.f;
            }
            __syncthreads();
        }
    } else {
        for (int c = threadIdx.x; c < C; c += 256) {
            float acc = 0.f;
            for (int n = 0; n < N; ++n) acc = fmaf(dz1_ws[(size_t)n * C + r], mean[(size_t)n * C + c], acc);
            dW1[(size_t)r * C + c] = acc;
        }
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += dz1_ws[(size_t)n * C + r];
            db1[r] = s;
        }
        // cross[n][r] = inv_count * sum_i W1[i][r] dz1[n][i]
        __shared__ float red[256];
        for (int n = 0; n < N; ++n) {
            float part = 0.f;
            for (int i = threadIdx.x; i < C; i += 256) part = fmaf(W1[(size_t)i * C + r], dz1_ws[(size_t)n * C + i], part);
            red[threadIdx.x] = part;
            __syncthreads();
            if (threadIdx.x == 0) {
                float s = 0.f;
                for (int k = 0; k < 256; ++k) s += red[k];
                cross[(size_t)n * C + r] = s * inv_count;
            }
            __syncthreads();
        }
    }
}

// ---- main passes -------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void se_apply_fwd_kernel(const T* __restrict__ e, int e_ld, long long S, int C, SeGeom sg, const float* __restrict__ a,
                                                           const float* __restrict__ w, const float* __restrict__ b0, float* __restrict__ bgate,
                                                           T* __restrict__ y, int y_ld, int mode) {
    constexpr int EPC = Tr<T>::EPC;
    const int n = blockIdx.y;
    const int lg = threadIdx.x % sg.G, grp = threadIdx.x / sg.G;
    float av[SE_MAXK][EPC], wv[SE_MAXK][EPC];
#pragma unroll
    for (int k = 0; k < SE_MAXK; ++k)
        if (k < sg.K)
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                av[k][i] = mode != 2 ? a[(size_t)n * C + (lg + k * sg.G) * EPC + i] : 0.f;
                wv[k][i] = mode != 1 ? w[(lg + k * sg.G) * EPC + i] : 0.f;
            }
    const float bias = mode != 1 ? b0[0] : 0.f;
    const T* eb = e + (size_t)n * S * e_ld;
    T* yb = y + (size_t)n * S * y_ld;
    for (long long v = (long long)blockIdx.x * sg.gpb + grp; v < S; v += (long long)gridDim.x * sg.gpb) {
        float f[SE_MAXK][EPC];
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < SE_MAXK; ++k)
            if (k < sg.K) {
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(eb + v * e_ld + (size_t)(lg + k * sg.G) * EPC), f[k]);
#pragma unroll
                for (int i = 0; i < EPC; ++i) q = fmaf(wv[k][i], f[k][i], q);
            }
        q = se_group_sum(q, sg.G) + bias;
        const float bg = se_sigmoid(q);
        if (lg == 0 && mode != 1) bgate[(size_t)n * S + v] = bg;
#pragma unroll
        for (int k = 0; k < SE_MAXK; ++k)
            if (k < sg.K) {
                float o[EPC];
#pragma unroll
                for (int i = 0; i < EPC; ++i) {
                    const float pa = f[k][i] * av[k][i], pb = f[k][i] * bg;
                    o[i] = mode == 0 ? fmaxf(pa, pb) : (mode == 1 ? pa : pb);
                }
                *reinterpret_cast<u32x4*>(yb + v * y_ld + (size_t)(lg + k * sg.G) * EPC) = pack_chunk<T>(o);
            }
    }
}

// partial slabs: part_da[n][slot][C], part_dw[n][slot][C], part_db0[n][slot]; slot = blockIdx.x  (< SE_PARTS)
template <typename T>
__global__ __launch_bounds__(256) void se_bwd_reduce_kernel(const T* __restrict__ g, int g_ld, const T* __restrict__ e, int e_ld, long long S, int C,
                                                            SeGeom sg, const float* __restrict__ a, const float* __restrict__ bgate,
                                                            float* __restrict__ dq, float* __restrict__ part_da, float* __restrict__ part_dw,
                                                            float* __restrict__ part_db0, int nslots, int mode) {
    constexpr int EPC = Tr<T>::EPC;
    const int n = blockIdx.y;
    const int lg = threadIdx.x % sg.G, grp = threadIdx.x / sg.G;
    float av[SE_MAXK][EPC], da[SE_MAXK][EPC], dw[SE_MAXK][EPC];
#pragma unroll
    for (int k = 0; k < SE_MAXK; ++k)
#pragma unroll
        for (int i = 0; i < EPC; ++i) {
            av[k][i] = (k < sg.K && mode != 2) ? a[(size_t)n * C + (lg + k * sg.G) * EPC + i] : 0.f;
            da[k][i] = dw[k][i] = 0.f;
        }
    float db0 = 0.f;
    const T* eb = e + (size_t)n * S * e_ld;
    const T* gb = g + (size_t)n * S * g_ld;
    for (long long v = (long long)blockIdx.x * sg.gpb + grp; v < S; v += (long long)gridDim.x * sg.gpb) {
        const float bg = mode != 1 ? bgate[(size_t)n * S + v] : 0.f;
        float f[SE_MAXK][EPC];
        float dbv = 0.f;
#pragma unroll
        for (int k = 0; k < SE_MAXK; ++k)
            if (k < sg.K) {
                float gg[EPC];
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(eb + v * e_ld + (size_t)(lg + k * sg.G) * EPC), f[k]);
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(gb + v * g_ld + (size_t)(lg + k * sg.G) * EPC), gg);
#pragma unroll
                for (int i = 0; i < EPC; ++i) {
                    const float ge = gg[i] * f[k][i];
                    const bool sa = mode == 0 ? (f[k][i] * av[k][i] >= f[k][i] * bg) : (mode == 1);
                    da[k][i] += sa ? ge : 0.f;
                    dbv += sa ? 0.f : ge;
                }
            }
        dbv = se_group_sum(dbv, sg.G);
        const float dqv = dbv * bg * (1.f - bg);
        if (lg == 0) {
            if (mode != 1) dq[(size_t)n * S + v] = dqv;
            db0 += dqv;
        }
#pragma unroll
        for (int k = 0; k < SE_MAXK; ++k)
            if (k < sg.K)
#pragma unroll
                for (int i = 0; i < EPC; ++i) dw[k][i] = fmaf(dqv, f[k][i], dw[k][i]);
    }
    // block-level sums over the voxel groups (fixed order), then one partial slab per block
    extern __shared__ float red[];                      // [gpb][C]
    float* pa = part_da + ((size_t)n * nslots + blockIdx.x) * C;
    float* pw = part_dw + ((size_t)n * nslots + blockIdx.x) * C;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
        for (int k = 0; k < SE_MAXK; ++k)
            if (k < sg.K)
#pragma unroll
                for (int i = 0; i < EPC; ++i) red[(size_t)grp * C + (lg + k * sg.G) * EPC + i] = round == 0 ? da[k][i] : dw[k][i];
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float sum = 0.f;
            for (int gi = 0; gi < sg.gpb; ++gi) sum += red[(size_t)gi * C + c];
            (round == 0 ? pa : pw)[c] = sum;
        }
        __syncthreads();
    }
    if (lg == 0) red[grp] = db0;
    __syncthreads();
    if (threadIdx.x == 0) {
        float sum = 0.f;
        for (int gi = 0; gi < sg.gpb; ++gi) sum += red[gi];
        part_db0[(size_t)n * nslots + blockIdx.x] = sum;
    }
}

// da[n][c] = sum_slots ; dw[c] = sum_n sum_slots ; db0 = sum_n sum_slots   (fixed order, one thread per output)
__global__ __launch_bounds__(256) void se_bwd_finish_kernel(const float* __restrict__ part_da, const float* __restrict__ part_dw,
                                                            const float* __restrict__ part_db0, int N, int C, int nslots, float* __restrict__ da,
                                                            float* __restrict__ dw, float* __restrict__ db0) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < N * C) {
        const int n = idx / C, c = idx - n * C;
        double s = 0.0;
        for (int k = 0; k < nslots; ++k) s += (double)part_da[((size_t)n * nslots + k) * C + c];
        da[idx] = (float)s;
    }
    if (idx < C) {
        double s = 0.0;
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < nslots; ++k) s += (double)part_dw[((size_t)n * nslots + k) * C + idx];
        dw[idx] = (float)s;
    }
    if (idx == 0) {
        double s = 0.0;
        for (int k = 0; k < N * nslots; ++k) s += (double)part_db0[k];
        db0[0] = (float)s;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void se_bwd_apply_kernel(const T* g, int g_ld, const T* __restrict__ e, int e_ld, long long S, int C,
                                                           SeGeom sg, const float* __restrict__ a, const float* __restrict__ bgate,
                                                           const float* __restrict__ dq, const float* __restrict__ w, const float* __restrict__ cross,
                                                           T* de, int de_ld, int mode, int relu_mask) {
    constexpr int EPC = Tr<T>::EPC;
    const int n = blockIdx.y;
    const int lg = threadIdx.x % sg.G, grp = threadIdx.x / sg.G;
    float av[SE_MAXK][EPC], wv[SE_MAXK][EPC], cr[SE_MAXK][EPC];
#pragma unroll
    for (int k = 0; k < SE_MAXK; ++k)
        if (k < sg.K)
#pragma unroll
            for (int i = 0; i < EPC; ++i) {
                const int c = (lg + k * sg.G) * EPC + i;
                av[k][i] = mode != 2 ? a[(size_t)n * C + c] : 0.f;
                wv[k][i] = mode != 1 ? w[c] : 0.f;
                cr[k][i] = mode != 2 ? cross[(size_t)n * C + c] : 0.f;
            }
    const T* eb = e + (size_t)n * S * e_ld;
    const T* gb = g + (size_t)n * S * g_ld;
    T* ob = de + (size_t)n * S * de_ld;
    for (long long v = (long long)blockIdx.x * sg.gpb + grp; v < S; v += (long long)gridDim.x * sg.gpb) {
        const float bg = mode != 1 ? bgate[(size_t)n * S + v] : 0.f, dqv = mode != 1 ? dq[(size_t)n * S + v] : 0.f;
#pragma unroll
        for (int k = 0; k < SE_MAXK; ++k)
            if (k < sg.K) {
                float f[EPC], gg[EPC], o[EPC];
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(eb + v * e_ld + (size_t)(lg + k * sg.G) * EPC), f);
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(gb + v * g_ld + (size_t)(lg + k * sg.G) * EPC), gg);
#pragma unroll
                for (int i = 0; i < EPC; ++i) {
                    const bool sa = mode == 0 ? (f[i] * av[k][i] >= f[i] * bg) : (mode == 1);
                    float gate = sa ? av[k][i] : bg;
                    if (mode == 0 && !relu_mask && f[i] == 0.f) gate = 0.5f * (av[k][i] + bg);          // torch.max's tie rule (both products are 0)
                    const float d = gg[i] * gate + cr[k][i] + dqv * wv[k][i];
                    o[i] = (!relu_mask || f[i] > 0.f) ? d : 0.f;
                }
                *reinterpret_cast<u32x4*>(ob + v * de_ld + (size_t)(lg + k * sg.G) * EPC) = pack_chunk<T>(o);
            }
    }
}

static int se_check(const char* what, int dtype, int N, long long S, int C, SeGeom* sg) {
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "%s: bad dtype %d", what, dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(N > 0 && S > 0 && C > 0 && C % 64 == 0, MIS_EINVAL, "%s: N %d, S %lld, C %d (C %% 64 == 0)", what, N, S, C);
    *sg = se_geom(C, EPC);
    MIS_REQUIRE((sg->G & (sg->G - 1)) == 0 && sg->nch % sg->G == 0 && sg->K <= SE_MAXK, MIS_EUNSUPPORTED,
                "%s: C %d needs a power-of-two lane group and at most %d chunks per lane", what, C, SE_MAXK);
    return MIS_OK;
}

static unsigned se_blocks(long long S, const SeGeom& sg) {
    long long b = (S + sg.gpb * 4 - 1) / (sg.gpb * 4);
    if (b > SE_PARTS) b = SE_PARTS;
    return (unsigned)(b < 1 ? 1 : b);
}

extern "C" int mis_se_fc_fwd(const float* chan_sum, double count, const float* W1, const float* b1, const float* W2, const float* b2, int N, int C,
                             float* mean, float* z1, float* a, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(chan_sum && W1 && b1 && W2 && b2 && mean && z1 && a, MIS_EINVAL, "se_fc_fwd: null pointer");
    MIS_REQUIRE(N > 0 && C > 0 && C <= 4096 && count > 0.0, MIS_EINVAL, "se_fc_fwd: sizes");
    hipLaunchKernelGGL(se_fc_fwd_kernel, dim3(N), dim3(256), 2 * C * sizeof(float), reinterpret_cast<hipStream_t>(stream), chan_sum, (float)(1.0 / count), W1,
                       b1, W2, b2, C, mean, z1, a);
    MIS_LAUNCH_CHECK("se_fc_fwd");
    return MIS_OK;
}

static int se_apply_fwd_impl(int dtype, const void* e, int e_ld, int N, long long S, int C, const float* a, const float* w, const float* b0, float* bgate, void* y,
                             int y_ld, int mode, void* stream) {
    (void)hipGetLastError();
    SeGeom sg;
    if (int rc = se_check("se_apply_fwd", dtype, N, S, C, &sg)) return rc;
    MIS_REQUIRE(mode >= 0 && mode <= 2, MIS_EINVAL, "se_apply_fwd: mode %d (0 scSE, 1 cSE, 2 sSE)", mode);
    MIS_REQUIRE(e && (a || mode == 2) && ((w && b0 && bgate) || mode == 1) && y && e != y && e_ld >= C && y_ld >= C, MIS_EINVAL, "se_apply_fwd: pointers / strides");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // the forward has no per-slot partials: use a wide grid
    long long b = (S + sg.gpb - 1) / sg.gpb;
    if (b > 8192) b = 8192;
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(se_apply_fwd_kernel<__bf16>, dim3((unsigned)b, N), dim3(256), 0, st, (const __bf16*)e, e_ld, S, C, sg, a, w, b0, bgate, (__bf16*)y, y_ld, mode);
    else
        hipLaunchKernelGGL(se_apply_fwd_kernel<float>, dim3((unsigned)b, N), dim3(256), 0, st, (const float*)e, e_ld, S, C, sg, a, w, b0, bgate, (float*)y, y_ld, mode);
    MIS_LAUNCH_CHECK("se_apply_fwd");
    return MIS_OK;
}
extern "C" int mis_se_apply_fwd(int dtype, const void* e, int e_ld, int N, long long S, int C, const float* a, const float* w, const float* b0,
                                float* bgate, void* y, int y_ld, void* stream) {
    return se_apply_fwd_impl(dtype, e, e_ld, N, S, C, a, w, b0, bgate, y, y_ld, 0, stream);
}
extern "C" int mis_se_layer_fwd(int dtype, const void* e, int e_ld, int N, long long S, int C, const float* a, const float* w, const float* b0, float* bgate,
                                void* y, int y_ld, int mode, void* stream) {
    return se_apply_fwd_impl(dtype, e, e_ld, N, S, C, a, w, b0, bgate, y, y_ld, mode, stream);
}

extern "C" size_t mis_se_bwd_workspace_bytes(int N, int C) { return ((size_t)2 * N * SE_PARTS * C + (size_t)N * SE_PARTS + (size_t)N * C) * sizeof(float); }

/* g: dL/d(out), may already carry the [e > 0] mask.  Writes dq (N*S), da (N*C), dw (C), db0 (1). */
static int se_bwd_reduce_impl(int dtype, const void* g, int g_ld, const void* e, int e_ld, int N, long long S, int C, const float* a, const float* bgate,
                              float* workspace, float* dq, float* da, float* dw, float* db0, int mode, void* stream) {
    (void)hipGetLastError();
    SeGeom sg;
    if (int rc = se_check("se_bwd_reduce", dtype, N, S, C, &sg)) return rc;
    MIS_REQUIRE(mode >= 0 && mode <= 2, MIS_EINVAL, "se_bwd_reduce: mode %d", mode);
    MIS_REQUIRE(g && e && (a || mode == 2) && ((bgate && dq) || mode == 1) && workspace && da && dw && db0 && g_ld >= C && e_ld >= C, MIS_EINVAL,
                "se_bwd_reduce: pointers / strides");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned bx = se_blocks(S, sg);
    const int nslots = (int)bx;
    const size_t lds = (size_t)sg.gpb * C * sizeof(float);
    float* pa = workspace;
    float* pw = pa + (size_t)N * SE_PARTS * C;
    float* pb = pw + (size_t)N * SE_PARTS * C;
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(se_bwd_reduce_kernel<__bf16>, dim3(bx, N), dim3(256), lds, st, (const __bf16*)g, g_ld, (const __bf16*)e, e_ld, S, C, sg, a, bgate, dq,
                           pa, pw, pb, nslots, mode);
    else
        hipLaunchKernelGGL(se_bwd_reduce_kernel<float>, dim3(bx, N), dim3(256), lds, st, (const float*)g, g_ld, (const float*)e, e_ld, S, C, sg, a, bgate, dq, pa,
                           pw, pb, nslots, mode);
    MIS_LAUNCH_CHECK("se_bwd_reduce");
    hipLaunchKernelGGL(se_bwd_finish_kernel, dim3((N * C + 255) / 256), dim3(256), 0, st, (const float*)pa, (const float*)pw, (const float*)pb, N, C, nslots,
                       da, dw, db0);
    MIS_LAUNCH_CHECK("se_bwd_finish");
    return MIS_OK;
}
extern "C" int mis_se_bwd_reduce(int dtype, const void* g, int g_ld, const void* e, int e_ld, int N, long long S, int C, const float* a,
                                 const float* bgate, float* workspace, float* dq, float* da, float* dw, float* db0, void* stream) {
    return se_bwd_reduce_impl(dtype, g, g_ld, e, e_ld, N, S, C, a, bgate, workspace, dq, da, dw, db0, 0, stream);
}
extern "C" int mis_se_layer_bwd_reduce(int dtype, const void* g, int g_ld, const void* e, int e_ld, int N, long long S, int C, const float* a, const float* bgate,
                                       float* workspace, float* dq, float* da, float* dw, float* db0, int mode, void* stream) {
    return se_bwd_reduce_impl(dtype, g, g_ld, e, e_ld, N, S, C, a, bgate, workspace, dq, da, dw, db0, mode, stream);
}

extern "C" int mis_se_fc_bwd(const float* da, const float* a, const float* z1, const float* mean, const float* W1, const float* W2, int N, int C,
                             double count, float* workspace, float* dW1, float* db1, float* dW2, float* db2, float* cross, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(da && a && z1 && mean && W1 && W2 && workspace && dW1 && db1 && dW2 && db2 && cross, MIS_EINVAL, "se_fc_bwd: null pointer");
    MIS_REQUIRE(N > 0 && C > 0 && count > 0.0, MIS_EINVAL, "se_fc_bwd: sizes");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* dz1 = workspace + (size_t)2 * N * SE_PARTS * C + (size_t)N * SE_PARTS;       // the tail of the mis_se_bwd workspace
    for (int phase = 0; phase < 2; ++phase)
        hipLaunchKernelGGL(se_fc_bwd_kernel, dim3(C), dim3(256), 0, st, da, a, z1, mean, W1, W2, N, C, (float)(1.0 / count), dz1, dW1, db1, dW2, db2, cross,
                           phase);
    MIS_LAUNCH_CHECK("se_fc_bwd");
    return MIS_OK;
}

static int se_bwd_apply_impl(int dtype, const void* g, int g_ld, const void* e, int e_ld, int N, long long S, int C, const float* a, const float* bgate,
                             const float* dq, const float* w, const float* cross, void* de, int de_ld, int mode, int relu_mask, void* stream) {
    (void)hipGetLastError();
    SeGeom sg;
    if (int rc = se_check("se_bwd_apply", dtype, N, S, C, &sg)) return rc;
    MIS_REQUIRE(mode >= 0 && mode <= 2, MIS_EINVAL, "se_bwd_apply: mode %d", mode);
    MIS_REQUIRE(g && e && ((a && cross) || mode == 2) && ((bgate && dq && w) || mode == 1) && de && g_ld >= C && e_ld >= C && de_ld >= C, MIS_EINVAL,
                "se_bwd_apply: pointers / strides");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    long long b = (S + sg.gpb - 1) / sg.gpb;
    if (b > 8192) b = 8192;
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(se_bwd_apply_kernel<__bf16>, dim3((unsigned)b, N), dim3(256), 0, st, (const __bf16*)g, g_ld, (const __bf16*)e, e_ld, S, C, sg, a, bgate,
                           dq, w, cross, (__bf16*)de, de_ld, mode, relu_mask);
    else
        hipLaunchKernelGGL(se_bwd_apply_kernel<float>, dim3((unsigned)b, N), dim3(256), 0, st, (const float*)g, g_ld, (const float*)e, e_ld, S, C, sg, a, bgate, dq,
                           w, cross, (float*)de, de_ld, mode, relu_mask);
    MIS_LAUNCH_CHECK("se_bwd_apply");
    return MIS_OK;
}
extern "C" int mis_se_bwd_apply(int dtype, const void* g, int g_ld, const void* e, int e_ld, int N, long long S, int C, const float* a,
                                const float* bgate, const float* dq, const float* w, const float* cross, void* de, int de_ld, void* stream) {
    return se_bwd_apply_impl(dtype, g, g_ld, e, e_ld, N, S, C, a, bgate, dq, w, cross, de, de_ld, 0, 1, stream);
}
extern "C" int mis_se_layer_bwd_apply(int dtype, const void* g, int g_ld, const void* e, int e_ld, int N, long long S, int C, const float* a, const float* bgate,
                                      const float* dq, const float* w, const float* cross, void* de, int de_ld, int mode, int relu_mask, void* stream) {
    return se_bwd_apply_impl(dtype, g, g_ld, e, e_ld, N, S, C, a, bgate, dq, w, cross, de, de_ld, mode, relu_mask, stream);
}
