// Building blocks of the UNet 3+ decoders (reference model/unet2d/unet.py:136-446): nn.MaxPool2d(k, k, ceil_mode=True) for the
// encoder-to-decoder skip paths and nn.Upsample(scale_factor=s, mode='bilinear') (align_corners=False) for the decoder-to-decoder paths,
// NHWC, forward and backward.  HBM-bound index kernels, one 16-byte channel chunk per thread.
//   max-pool: windows do not overlap (stride = k) and are clipped at the border; the first maximum in row-major scan order wins
//   (PyTorch's CPU kernel); backward re-finds it and writes the whole window (every input pixel belongs to exactly one window).
//   bilinear: src = max(0, (dst + 0.5) / s - 0.5), i0 = floor(src), i1 = min(i0 + 1, n - 1), lambda = src - i0.  Backward is the
//   adjoint, computed separably as two deterministic gathers (along W into a scratch tensor, then along H).
#include "common.hpp"

template <typename T>
__global__ __launch_bounds__(256) void maxpoolk_fwd_kernel(const T* __restrict__ x, int x_ld, T* __restrict__ y, int y_ld, int N, int H, int W, int C, int k) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC, OH = (H + k - 1) / k, OW = (W + k - 1) / k;
    const long long total = (long long)N * OH * OW * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long r = i / nch;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        float best[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) best[e] = -INFINITY;
        for (int dy = 0; dy < k && oy * k + dy < H; ++dy)
            for (int dx = 0; dx < k && ox * k + dx < W; ++dx) {
                float f[EPC];
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(x + (((size_t)n * H + oy * k + dy) * W + ox * k + dx) * x_ld + (size_t)ch * EPC), f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) best[e] = f[e] > best[e] ? f[e] : best[e];
            }
        *reinterpret_cast<u32x4*>(y + (((size_t)n * OH + oy) * OW + ox) * y_ld + (size_t)ch * EPC) = pack_chunk<T>(best);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpoolk_bwd_kernel(const T* __restrict__ x, int x_ld, const T* __restrict__ dy, int dy_ld, T* __restrict__ dx, int dx_ld,
                                                           int N, int H, int W, int C, int k) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC, OH = (H + k - 1) / k, OW = (W + k - 1) / k;
    const long long total = (long long)N * OH * OW * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long r = i / nch;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        float best[EPC];
        int arg[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            best[e] = -INFINITY;
            arg[e] = 0;
        }
        for (int wy = 0; wy < k && oy * k + wy < H; ++wy)
            for (int wx = 0; wx < k && ox * k + wx < W; ++wx) {
                float f[EPC];
                unpack_chunk<T>(*reinterpret_cast<const u32x4*>(x + (((size_t)n * H + oy * k + wy) * W + ox * k + wx) * x_ld + (size_t)ch * EPC), f);
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    if (f[e] > best[e]) {
                        best[e] = f[e];
                        arg[e] = wy * k + wx;
                    }
            }
        float g[EPC];
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(dy + (((size_t)n * OH + oy) * OW + ox) * dy_ld + (size_t)ch * EPC), g);
        for (int wy = 0; wy < k && oy * k + wy < H; ++wy)
            for (int wx = 0; wx < k && ox * k + wx < W; ++wx) {
                float o[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) o[e] = (arg[e] == wy * k + wx) ? g[e] : 0.f;
                *reinterpret_cast<u32x4*>(dx + (((size_t)n * H + oy * k + wy) * W + ox * k + wx) * dx_ld + (size_t)ch * EPC) = pack_chunk<T>(o);
            }
    }
}

__device__ __forceinline__ void bil_src(int dst, float inv_scale, int n, int& i0, int& i1, float& lam) {
    float src = ((float)dst + 0.5f) * inv_scale - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > n - 1) i0 = n - 1;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    lam = src - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* __restrict__ x, int x_ld, T* __restrict__ y, int y_ld, int N, int H, int W, int C, int s) {
    constexpr int EPC = Tr<T>::EPC;
    const int nch = C / EPC, OH = H * s, OW = W * s;
    const float inv = 1.0f / (float)s;
    const long long total = (long long)N * OH * OW * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long r = i / nch;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        int y0, y1, x0, x1;
        float ly, lx;
        bil_src(oy, inv, H, y0, y1, ly);
        bil_src(ox, inv, W, x0, x1, lx);
        float a[EPC], b[EPC], c[EPC], d[EPC], o[EPC];
        const T* base = x + (size_t)n * H * W * x_ld + (size_t)ch * EPC;
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + ((size_t)y0 * W + x0) * x_ld), a);
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + ((size_t)y0 * W + x1) * x_ld), b);
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + ((size_t)y1 * W + x0) * x_ld), c);
        unpack_chunk<T>(*reinterpret_cast<const u32x4*>(base + ((size_t)y1 * W + x1) * x_ld), d);
        const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = hy * (hx * a[e] + lx * b[e]) + ly * (hx * c[e] + lx * d[e]);     // PyTorch's evaluation order
        *reinterpret_cast<u32x4*>(y + (((size_t)n * OH + oy) * OW + ox) * y_ld + (size_t)ch * EPC) = pack_chunk<T>(o);
    }
}

// adjoint along one axis: out[j] = sum_{o : o references j} w(o, j) * in[o], other axis untouched.
// axis 1 = W: in (N, R, n*s, C) -> out (N, R, n, C);  axis 0 = H: in (N, n*s, R, C) -> out (N, n, R, C).  fp32 scratch in / T or fp32 out.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void bilinear_adj_kernel(const TI* __restrict__ in, int in_ld, TO* __restrict__ out, int out_ld, int N, int R, int n, int C,
                                                           int s, int axis) {
    constexpr int EI = Tr<TI>::EPC, EO = Tr<TO>::EPC;
    constexpr int EPC = EI < EO ? EI : EO;      // channels per thread (4 when either side is fp32)
    const int nch = C / EPC;
    const float inv = 1.0f / (float)s;
    const long long total = (long long)N * R * n * nch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        long long r = i / nch;
        int j, rr;
        if (axis == 1) {
            j = (int)(r % n);
            r /= n;
            rr = (int)(r % R);
        } else {
            rr = (int)(r % R);
            r /= R;
            j = (int)(r % n);
        }
        const int nb = (int)(r / (axis == 1 ? R : n));
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        int lo = s * (j - 1) - 1, hi = s * (j + 1) + s;
        if (lo < 0) lo = 0;
        if (hi > n * s - 1) hi = n * s - 1;
        for (int o = lo; o <= hi; ++o) {
            int i0, i1;
            float lam;
            bil_src(o, inv, n, i0, i1, lam);
            float wgt = 0.f;
            if (i0 == j) wgt += 1.f - lam;
            if (i1 == j) wgt += lam;
            if (wgt == 0.f) continue;
            const size_t pix = axis == 1 ? (((size_t)nb * R + rr) * (size_t)(n * s) + o) : (((size_t)nb * (size_t)(n * s) + o) * R + rr);
            const TI* src = in + pix * in_ld + (size_t)ch * EPC;
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = fmaf(wgt, ld_elem<TI>(src + e), acc[e]);
        }
        const size_t opix = axis == 1 ? (((size_t)nb * R + rr) * n + j) : (((size_t)nb * n + j) * R + rr);
        TO* dst = out + opix * out_ld + (size_t)ch * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) st_elem<TO>(dst + e, acc[e]);
    }
}

static unsigned pu_grid(long long total) {
    long long b = (total + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (unsigned)b;
}

static int pu_check(const char* what, int dtype, const void* a, const void* b, int N, int H, int W, int C, int k, int lda, int ldb) {
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "%s: bad dtype %d", what, dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(a && b, MIS_EINVAL, "%s: null pointer", what);
    MIS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % EPC == 0 && lda % EPC == 0 && ldb % EPC == 0 && lda >= C && ldb >= C, MIS_EINVAL, "%s: sizes", what);
    MIS_REQUIRE(k >= 1 && k <= 32, MIS_EUNSUPPORTED, "%s: window / scale %d", what, k);
    return MIS_OK;
}

extern "C" int mis_maxpoolk_fwd(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, int H, int W, int C, int k, void* stream) {
    (void)hipGetLastError();
    if (int rc = pu_check("maxpoolk_fwd", dtype, x, y, N, H, W, C, k, x_ld, y_ld)) return rc;
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    const unsigned g = pu_grid((long long)N * ((H + k - 1) / k) * ((W + k - 1) / k) * (C / EPC));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(maxpoolk_fwd_kernel<__bf16>, dim3(g), dim3(256), 0, s, (const __bf16*)x, x_ld, (__bf16*)y, y_ld, N, H, W, C, k);
    else
        hipLaunchKernelGGL(maxpoolk_fwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, x_ld, (float*)y, y_ld, N, H, W, C, k);
    MIS_LAUNCH_CHECK("maxpoolk_fwd");
    return MIS_OK;
}

extern "C" int mis_maxpoolk_bwd(int dtype, const void* x, int x_ld, const void* dy, int dy_ld, void* dx, int dx_ld, int N, int H, int W, int C, int k,
                                void* stream) {
    (void)hipGetLastError();
    if (int rc = pu_check("maxpoolk_bwd", dtype, x, dy, N, H, W, C, k, x_ld, dy_ld)) return rc;
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(dx && dx_ld % EPC == 0 && dx_ld >= C, MIS_EINVAL, "maxpoolk_bwd: dx");
    const unsigned g = pu_grid((long long)N * ((H + k - 1) / k) * ((W + k - 1) / k) * (C / EPC));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(maxpoolk_bwd_kernel<__bf16>, dim3(g), dim3(256), 0, s, (const __bf16*)x, x_ld, (const __bf16*)dy, dy_ld, (__bf16*)dx, dx_ld, N, H, W, C, k);
    else
        hipLaunchKernelGGL(maxpoolk_bwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, x_ld, (const float*)dy, dy_ld, (float*)dx, dx_ld, N, H, W, C, k);
    MIS_LAUNCH_CHECK("maxpoolk_bwd");
    return MIS_OK;
}

extern "C" int mis_bilinear_up_fwd(int dtype, const void* x, int x_ld, void* y, int y_ld, int N, int H, int W, int C, int scale, void* stream) {
    (void)hipGetLastError();
    if (int rc = pu_check("bilinear_up_fwd", dtype, x, y, N, H, W, C, scale, x_ld, y_ld)) return rc;
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    const unsigned g = pu_grid((long long)N * H * scale * W * scale * (C / EPC));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(bilinear_fwd_kernel<__bf16>, dim3(g), dim3(256), 0, s, (const __bf16*)x, x_ld, (__bf16*)y, y_ld, N, H, W, C, scale);
    else
        hipLaunchKernelGGL(bilinear_fwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, x_ld, (float*)y, y_ld, N, H, W, C, scale);
    MIS_LAUNCH_CHECK("bilinear_up_fwd");
    return MIS_OK;
}

extern "C" size_t mis_bilinear_up_bwd_workspace_bytes(int N, int H, int W, int C, int scale) {
    return (size_t)N * H * scale * W * C * sizeof(float);       // (N, H*s, W, C) fp32 after the pass along W
}

// dy (N, H*s, W*s, C) -> dx (N, H, W, C)
extern "C" int mis_bilinear_up_bwd(int dtype, const void* dy, int dy_ld, void* dx, int dx_ld, int N, int H, int W, int C, int scale, float* workspace,
                                   void* stream) {
    (void)hipGetLastError();
    if (int rc = pu_check("bilinear_up_bwd", dtype, dy, dx, N, H, W, C, scale, dy_ld, dx_ld)) return rc;
    MIS_REQUIRE(workspace != nullptr, MIS_EINVAL, "bilinear_up_bwd: workspace");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g1 = pu_grid((long long)N * H * scale * W * (C / 4)), g2 = pu_grid((long long)N * H * W * (C / 4));
    if (dtype == MIS_BF16) {
        hipLaunchKernelGGL((bilinear_adj_kernel<__bf16, float>), dim3(g1), dim3(256), 0, s, (const __bf16*)dy, dy_ld, workspace, C, N, H * scale, W, C, scale, 1);
        hipLaunchKernelGGL((bilinear_adj_kernel<float, __bf16>), dim3(g2), dim3(256), 0, s, (const float*)workspace, C, (__bf16*)dx, dx_ld, N, W, H, C, scale, 0);
    } else {
        hipLaunchKernelGGL((bilinear_adj_kernel<float, float>), dim3(g1), dim3(256), 0, s, (const float*)dy, dy_ld, workspace, C, N, H * scale, W, C, scale, 1);
        hipLaunchKernelGGL((bilinear_adj_kernel<float, float>), dim3(g2), dim3(256), 0, s, (const float*)workspace, C, (float*)dx, dx_ld, N, W, H, C, scale, 0);
    }
    MIS_LAUNCH_CHECK("bilinear_up_bwd");
    return MIS_OK;
}

// ---- classification-guided module of UNet_3Plus_DeepSup_CGM (model/unet2d/unet.py:998-1003, 1012-1038, 1147-1153) --------------------------------
// cls = Sigmoid(AdaptiveMaxPool2d(1)(Conv2d(C, 2, 1)(x))) per sample, gate = float(argmax over the two classes) (first maximum on ties, after the
// sigmoid exactly like the reference: two saturated scores tie); the five segmentation maps are then sigmoid(d * gate).
template <typename T>
__global__ __launch_bounds__(256) void cgm_gate_kernel(const T* __restrict__ x, int x_ld, long long npix, int C, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ cls, float* __restrict__ gate) {
    constexpr int EPC = Tr<T>::EPC;
    const int n = blockIdx.x;
    const T* xb = x + (size_t)n * npix * x_ld;
    float m0 = -INFINITY, m1 = -INFINITY;
    for (long long p = threadIdx.x; p < npix; p += 256) {
        float a0 = 0.f, a1 = 0.f;
        for (int c = 0; c < C; c += EPC) {
            float f[EPC];
            unpack_chunk<T>(*reinterpret_cast<const u32x4*>(xb + p * x_ld + c), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                a0 = fmaf(f[e], w[c + e], a0);
                a1 = fmaf(f[e], w[C + c + e], a1);
            }
        }
        m0 = fmaxf(m0, a0 + b[0]);
        m1 = fmaxf(m1, a1 + b[1]);
    }
    __shared__ float r0[256], r1[256];
    r0[threadIdx.x] = m0;
    r1[threadIdx.x] = m1;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 256; ++k) {
            m0 = fmaxf(m0, r0[k]);
            m1 = fmaxf(m1, r1[k]);
        }
        const float s0 = 1.f / (1.f + expf(-m0)), s1 = 1.f / (1.f + expf(-m1));
        cls[2 * n] = s0;
        cls[2 * n + 1] = s1;
        gate[n] = s1 > s0 ? 1.f : 0.f;
    }
}

__global__ __launch_bounds__(256) void scale_sigmoid_kernel(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ gate,
                                                            long long per_sample, long long total, float* __restrict__ out, int backward) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float g = gate[i / per_sample];
        const float y = 1.f / (1.f + expf(-x[i] * g));
        out[i] = backward ? gy[i] * y * (1.f - y) * g : y;
    }
}

extern "C" int mis_cgm_gate(int dtype, const void* x, int x_ld, int N, long long npix, int C, const float* w, const float* b, float* cls, float* gate,
                            void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(dtype == MIS_F32 || dtype == MIS_BF16, MIS_EINVAL, "cgm_gate: bad dtype %d", dtype);
    const int EPC = dtype == MIS_BF16 ? 8 : 4;
    MIS_REQUIRE(x && w && b && cls && gate && N > 0 && npix > 0 && C > 0 && C % EPC == 0 && x_ld >= C && x_ld % EPC == 0, MIS_EINVAL, "cgm_gate: arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIS_BF16)
        hipLaunchKernelGGL(cgm_gate_kernel<__bf16>, dim3(N), dim3(256), 0, s, (const __bf16*)x, x_ld, npix, C, w, b, cls, gate);
    else
        hipLaunchKernelGGL(cgm_gate_kernel<float>, dim3(N), dim3(256), 0, s, (const float*)x, x_ld, npix, C, w, b, cls, gate);
    MIS_LAUNCH_CHECK("cgm_gate");
    return MIS_OK;
}

/* forward: out = sigmoid(x * gate[n]);  backward (gy != NULL): out = gy * y (1 - y) * gate[n] with y recomputed from x */
extern "C" int mis_scale_sigmoid(const float* x, const float* gy, const float* gate, int N, long long per_sample, float* out, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(x && gate && out && N > 0 && per_sample > 0, MIS_EINVAL, "scale_sigmoid: arguments");
    const long long total = (long long)N * per_sample;
    hipLaunchKernelGGL(scale_sigmoid_kernel, dim3(pu_grid(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, gy, gate, per_sample, total, out,
                       gy != nullptr ? 1 : 0);
    MIS_LAUNCH_CHECK("scale_sigmoid");
    return MIS_OK;
}
