// numpy's legacy RandomState.normal() stream on the device: MT19937 + the polar Box-Muller `legacy_gauss` with its cached second value, so that
// AdditiveGaussianNoise (reference augment/unet3d_augment/transforms.py:608-619: `m + random_state.normal(0, std, size=m.shape)`) adds THE SAME noise
// field as the reference, not merely one with the same statistics.
//
// Both kernels run ONE workgroup: the Mersenne-Twister recurrence is serial in blocks of 624 words (each block: three data-parallel phases), and the rejection
// loop of the polar method makes the position of sample i in the word stream depend on every earlier rejection (resolved with a block-wide prefix sum per chunk).
// ~10 ms per 128^3 volume: this is the bit-comparable mode; the default noise path stays the counter-based generator of augment.hip.
//   mis_mt19937_words : words [0, n) that numpy would draw next from state (key[624], pos)         (randomkit / numpy/random/src/mt19937)
//   mis_legacy_normal : out[i] = (float)((double)in[i] + scale * gauss_i), gauss = numpy legacy_gauss over those words; reports how many words were consumed
//                       and the value left in the gauss cache, so that the host RandomState can be advanced to the state the reference would be in.
#include "common.hpp"

namespace {
constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MT_UPPER = 0x80000000u, MT_LOWER = 0x7fffffffu, MT_A = 0x9908b0dfu;

__device__ __forceinline__ uint32_t mt_twist(uint32_t cur, uint32_t nxt, uint32_t far) {
    const uint32_t y = (cur & MT_UPPER) | (nxt & MT_LOWER);
    return far ^ (y >> 1) ^ ((y & 1u) ? MT_A : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
}   // namespace

// one workgroup of 640 threads; key_io: in = the generator's key, out = the key after the last regeneration; pos_io likewise (numpy's `pos`, 0..624)
__global__ __launch_bounds__(640) void mt19937_words_kernel(uint32_t* __restrict__ key_io, int* __restrict__ pos_io, uint32_t* __restrict__ out, long long n) {
    __shared__ uint32_t mt[MT_N];
    const int t = threadIdx.x;
    if (t < MT_N) mt[t] = key_io[t];
    int pos = *pos_io;
    __syncthreads();
    long long done = 0;
    while (done < n) {
        if (pos >= MT_N) {      // regenerate the block: phases [0,227), [227,454), [454,624) - each reads only finished / untouched words
            uint32_t v = 0;
            if (t < MT_N - MT_M) v = mt_twist(mt[t], mt[t + 1], mt[t + MT_M]);
            __syncthreads();
            if (t < MT_N - MT_M) mt[t] = v;
            __syncthreads();
            if (t >= MT_N - MT_M && t < 2 * (MT_N - MT_M)) v = mt_twist(mt[t], mt[t + 1], mt[t - (MT_N - MT_M)]);
            __syncthreads();
            if (t >= MT_N - MT_M && t < 2 * (MT_N - MT_M)) mt[t] = v;
            __syncthreads();
            if (t >= 2 * (MT_N - MT_M) && t < MT_N - 1) v = mt_twist(mt[t], mt[t + 1], mt[t - (MT_N - MT_M)]);
            __syncthreads();
            if (t >= 2 * (MT_N - MT_M) && t < MT_N - 1) mt[t] = v;
            __syncthreads();
            if (t == MT_N - 1) mt[t] = mt_twist(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
            __syncthreads();
            pos = 0;
        }
        const int avail = MT_N - pos;
        const long long take = (n - done) < avail ? (n - done) : avail;
        if (t < take) out[done + t] = mt_temper(mt[pos + t]);
        done += take;
        pos += (int)take;
    }
    __syncthreads();
    if (t < MT_N) key_io[t] = mt[t];
    if (t == 0) *pos_io = pos;
}

// one workgroup of 1024 threads.  words: the 32-bit stream; an ATTEMPT = 4 words = two doubles in [0,1) (numpy: (a>>5, b>>6) -> (a*2^26 + b) / 2^53) = (x1, x2);
// accepted when 0 < r2 = x1^2 + x2^2 < 1; it yields f*x2 (returned first) and f*x1 (cached, returned next), f = sqrt(-2 ln r2 / r2).
// result[0] = attempts consumed, result[1] = 1 if a value is left in the cache, result[2..3] = that value (double bits), result[4] = 1 on success (0: stream too short)
__global__ __launch_bounds__(1024) void legacy_normal_kernel(const uint32_t* __restrict__ words, long long nattempts, const float* __restrict__ in, float* __restrict__ out,
                                                              long long count, double scale, int has_gauss, double gauss0, unsigned long long* __restrict__ result) {
    constexpr int PER = 4;
    __shared__ long long s_base;
    __shared__ int s_cnt[1024];
    __shared__ unsigned long long s_res[4];
    const int t = threadIdx.x;
    if (t == 0) {
        s_base = 0;
        s_res[0] = 0; s_res[1] = 0; s_res[2] = 0; s_res[3] = 0;
    }
    // a value cached by an earlier call is the first sample
    long long first = 0;
    if (has_gauss && count > 0) {
        if (t == 0) out[0] = (float)((double)in[0] + scale * gauss0);
        first = 1;
    }
    const long long need = count - first;               // samples still to draw
    const long long pairs = (need + 1) / 2;
    __syncthreads();
    for (long long a0 = 0; a0 < nattempts && s_base < pairs; a0 += 1024LL * PER) {
        double v0[PER], v1[PER];
        int ok[PER], c = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const long long j = a0 + (long long)t * PER + k;
            ok[k] = 0;
            if (j < nattempts) {
                const uint32_t w0 = words[4 * j], w1 = words[4 * j + 1], w2 = words[4 * j + 2], w3 = words[4 * j + 3];
                const double d1 = ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) / 9007199254740992.0;
                const double d2 = ((double)(w2 >> 5) * 67108864.0 + (double)(w3 >> 6)) / 9007199254740992.0;
                const double x1 = 2.0 * d1 - 1.0, x2 = 2.0 * d2 - 1.0;
                const double r2 = x1 * x1 + x2 * x2;
                if (r2 < 1.0 && r2 != 0.0) {
                    const double f = sqrt(-2.0 * log(r2) / r2);
                    v0[k] = f * x2;
                    v1[k] = f * x1;
                    ok[k] = 1;
                    ++c;
                }
            }
        }
        s_cnt[t] = c;
        __syncthreads();
        // inclusive scan of the per-thread counts (Hillis-Steele over 1024 entries)
        for (int off = 1; off < 1024; off <<= 1) {
            const int add = (t >= off) ? s_cnt[t - off] : 0;
            __syncthreads();
            s_cnt[t] += add;
            __syncthreads();
        }
        const long long base = s_base;
        long long p = base + s_cnt[t] - c;               // index of this thread's first accepted pair
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (ok[k]) {
                if (p < pairs) {
                    const long long i0 = first + 2 * p, i1 = i0 + 1;
                    out[i0] = (float)((double)in[i0] + scale * v0[k]);
                    if (i1 < count) out[i1] = (float)((double)in[i1] + scale * v1[k]);
                    if (p == pairs - 1) {               // the last pair: everything after it stays in the generator
                        s_res[0] = (unsigned long long)(a0 + (long long)t * PER + k + 1);
                        s_res[1] = (i1 >= count) ? 1ull : 0ull;
                        s_res[2] = (unsigned long long)__double_as_longlong(v1[k]);
                        s_res[3] = 1ull;
                    }
                }
                ++p;
            }
        }
        __syncthreads();
        if (t == 1023) s_base = base + s_cnt[1023];
        __syncthreads();
    }
    if (t == 0) {
        if (pairs == 0) {       // nothing drawn (count == 0, or the cached value covered a single sample)
            result[0] = 0; result[1] = (has_gauss && count == 0) ? 1ull : 0ull;
            result[2] = (unsigned long long)__double_as_longlong(gauss0); result[4] = 1ull;
        } else {
            result[0] = s_res[0]; result[1] = s_res[1]; result[2] = s_res[2]; result[4] = s_res[3];
        }
    }
}

extern "C" int mis_mt19937_words(unsigned int* key_io, int* pos_io, unsigned int* out, long long n, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(key_io != nullptr && pos_io != nullptr && (out != nullptr || n == 0) && n >= 0, MIS_EINVAL, "mt19937_words: bad arguments");
    hipLaunchKernelGGL(mt19937_words_kernel, dim3(1), dim3(640), 0, reinterpret_cast<hipStream_t>(stream), key_io, pos_io, out, n);
    MIS_LAUNCH_CHECK("mt19937_words");
    return MIS_OK;
}

extern "C" int mis_legacy_normal(const unsigned int* words, long long nattempts, const float* in, float* out, long long count, double scale, int has_gauss,
                                 double gauss0, unsigned long long* result5, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(words != nullptr && result5 != nullptr && count >= 0 && nattempts >= 0 && (count == 0 || (in != nullptr && out != nullptr)), MIS_EINVAL,
                "legacy_normal: bad arguments");
    hipLaunchKernelGGL(legacy_normal_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), words, nattempts, in, out, count, scale, has_gauss, gauss0,
                       result5);
    MIS_LAUNCH_CHECK("legacy_normal");
    return MIS_OK;
}
