// Shared device/host helpers for libmisamd (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "misamd.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

void mis_set_error(const char* fmt, ...);

#define MIS_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            mis_set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)

#define MIS_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            mis_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return MIS_EHIP;                                                          \
        }                                                                             \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: set it once per (kernel, device), thread-safe
// (`done` = one bit per device ordinal, a function-local static of the launching template).
static inline int mis_set_dyn_lds(std::atomic<unsigned long long>& done, const void* fn, size_t bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        mis_set_error("%s: hipGetDevice failed", what);
        return MIS_EHIP;
    }
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return MIS_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        mis_set_error("%s: cannot raise the dynamic LDS limit to %zu bytes: %s", what, bytes, hipGetErrorString(e));
        return MIS_EHIP;
    }
    done.fetch_or(bit, std::memory_order_release);
    return MIS_OK;
}

// ---- element traits: one K chunk is always 128 bytes of channels per pixel -------------------------
template <typename T> struct Tr;
template <> struct Tr<float> {
    static constexpr int EPC = 4;   // elements per 16-byte chunk
    static constexpr int CK = 32;   // channels per K chunk
};
template <> struct Tr<__bf16> {
    static constexpr int EPC = 8;
    static constexpr int CK = 64;
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b16) { return __uint_as_float(b16 << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;   // v_cvt_pk_bf16_f32, RNE, NaN preserved
    return (uint32_t)__builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) { return f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16); }

template <typename T> __device__ __forceinline__ float ld_elem(const T* p);
template <> __device__ __forceinline__ float ld_elem<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_elem<__bf16>(const __bf16* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st_elem(T* p, float v);
template <> __device__ __forceinline__ void st_elem<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_elem<__bf16>(__bf16* p, float v) { *p = (__bf16)v; }

// unpack a 16-byte chunk into EPC floats / pack back
template <typename T> __device__ __forceinline__ void unpack_chunk(const u32x4& c, float* f);
template <> __device__ __forceinline__ void unpack_chunk<float>(const u32x4& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t u = c[i];   // (bit_cast straight from a vector element mis-compiles: copy to a scalar first)
        f[i] = __uint_as_float(u);
    }
}
template <> __device__ __forceinline__ void unpack_chunk<__bf16>(const u32x4& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf16_bits_to_f32(c[i] & 0xffffu);
        f[2 * i + 1] = bf16_bits_to_f32(c[i] >> 16);
    }
}
template <typename T> __device__ __forceinline__ u32x4 pack_chunk(const float* f);
template <> __device__ __forceinline__ u32x4 pack_chunk<float>(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __float_as_uint(f[i]);
    return c;
}
template <> __device__ __forceinline__ u32x4 pack_chunk<__bf16>(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    return c;
}

// One 16x16 MFMA "b128 step": both operands are 16 bytes per lane of K-contiguous data.
//   bf16: one v_mfma_f32_16x16x32_bf16 (lane group g supplies k = 8g..8g+7)
//   f32 : four v_mfma_f32_16x16x4_f32; lane group g supplies k = 4g+t at step t (any k order is fine as long as
//         A and B use the same one, which they do)
template <typename T> __device__ __forceinline__ void mma_b128(f32x4& acc, const u32x4& a, const u32x4& b) {
    if constexpr (sizeof(T) == 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t ua = a[i], ub = b[i];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ua), __uint_as_float(ub), acc, 0, 0, 0);
        }
    }
}

__device__ __forceinline__ u32x4 lds_read_b128(const char* base, int byte_off) {
    return *reinterpret_cast<const u32x4*>(base + byte_off);
}
__device__ __forceinline__ void lds_write_b128(char* base, int byte_off, const u32x4& v) {
    *reinterpret_cast<u32x4*>(base + byte_off) = v;
}

// Bijective XCD-aware remap: consecutive virtual ids land on the same XCD (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// wave / block reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
