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

// ---------------------------------------------------------------------------------------------------------
// Round 3: the same stream from MANY workgroups.  MT19937 is linear over GF(2): the key J words ahead is a GF(2) convolution of the next ~20.6 k words with the bit
// mask g = t^(J-1) mod phi (host side: augment/unet3d_augment/mt_jump.py) - x_{J+m} = XOR_{i : g_i} x_{i+1+m}.  states[c] = the key at the start of chunk c (J words
// per chunk); level k of a doubling scheme computes states[i + 2^k] from states[i] for all i < 2^k at once (one workgroup per jump: the stream segment lives in LDS,
// the 624 outputs are the lanes), then every chunk's words are generated by its own workgroup.  ~0.1 ms per level, instead of 10 ms of one serial workgroup per 128^3.
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int MT_B = MT_N - MT_M;                  // 227: words that can be produced in parallel (word n needs words <= n - 227)
constexpr int MT_JW = 33 * MT_N;                   // words of the stream a jump looks at: 19936 + 1 + 623 < 20592
}

// blockIdx.x = jump b (source states[src0 + (per_jump_src ? b : 0)], mask g + b * 624 when per_jump_mask, destination states[dst0 + b]); blockIdx.y = part of the
// mask (the 624 mask words are cut into gridDim.y ranges: every part XORs its share into the destination, which the host has zeroed - XOR is exact in any order)
__global__ __launch_bounds__(640) void mt_jump_kernel(uint32_t* __restrict__ states, int src0, int dst0, const uint32_t* __restrict__ g, int per_jump_src,
                                                      int per_jump_mask) {
    extern __shared__ uint32_t xs[];               // MT_JW words
    const int t = threadIdx.x;
    const uint32_t* src = states + (size_t)(src0 + (per_jump_src ? (int)blockIdx.x : 0)) * MT_N;
    const uint32_t* gm = g + (per_jump_mask ? (size_t)blockIdx.x * MT_N : 0);
    const int w0 = (int)(((long long)MT_N * blockIdx.y) / gridDim.y), w1 = (int)(((long long)MT_N * (blockIdx.y + 1)) / gridDim.y);
    const int need = 32 * w1 + 1 + MT_N;           // stream words this part looks at: indices < 32 * w1 + 1 + 624 <= MT_JW
    if (t < MT_N) xs[t] = src[t];
    __syncthreads();
    for (int n = MT_N; n < need; n += MT_B) {      // x[n + t] = x[n + t - 227] ^ twist(x[n + t - 624], x[n + t - 623])
        if (t < MT_B && n + t < MT_JW) xs[n + t] = mt_twist(xs[n + t - MT_N], xs[n + t - MT_N + 1], xs[n + t - MT_B]);
        __syncthreads();
    }
    if (t < MT_N) {
        uint32_t acc = 0;
        for (int w = w0; w < w1; ++w) {
            uint32_t bits = gm[w];                 // the same word for every lane: uniform control flow
            const uint32_t* base = xs + 32 * w + 1 + t;
            while (bits) {
                const int b = __builtin_ctz(bits);
                acc ^= base[b];
                bits &= bits - 1;
            }
        }
        atomicXor(&states[(size_t)(dst0 + blockIdx.x) * MT_N + t], acc);
    }
}

// chunk c = blockIdx.x produces the stream positions [c*J, (c+1)*J) (positions count from the first word of states[0]); tempered words with lo <= position < hi go
// to out[position - lo].  `raw_block` >= 0: additionally the UNTEMPERED 624 words of block raw_block (= positions 624*raw_block ..) go to raw_out (numpy's key array
// once the stream has been consumed up to somewhere inside that block).
__global__ __launch_bounds__(256) void mt_generate_kernel(const uint32_t* __restrict__ states, long long J, long long lo, long long hi, uint32_t* __restrict__ out,
                                                          long long raw_block, uint32_t* __restrict__ raw_out, const unsigned long long* __restrict__ used_dev,
                                                          long long pos0) {
    __shared__ uint32_t ring[1024];                // x_n at ring[n & 1023]: a batch of 227 writes never lands on a slot the same batch reads (they are >= 798 words old)
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * J;
    if (used_dev != nullptr) {                     // the raw block follows from the number of attempts the normal kernels consumed (result5[0], still on the device):
        const long long used = (long long)used_dev[0];      // the block in which numpy's position ends - an exhausted block stays current (pos = 624)
        const long long end = pos0 + 4 * used;
        raw_block = used == 0 ? -1 : (end % MT_N == 0 ? end / MT_N - 1 : end / MT_N);
    }
    const long long raw0 = raw_block >= 0 ? raw_block * MT_N : -1;
    const bool want_raw = raw0 >= base && raw0 < base + J;
    if ((base >= hi || base + J <= lo) && !want_raw) return;        // block-uniform
    const uint32_t* key = states + (size_t)blockIdx.x * MT_N;
    for (int k = t; k < MT_N; k += 256) {
        const uint32_t v = key[k];
        ring[k] = v;
        const long long p = base + k;
        if (p >= lo && p < hi) out[p - lo] = mt_temper(v);
        if (want_raw && p >= raw0 && p < raw0 + MT_N) raw_out[p - raw0] = v;
    }
    __syncthreads();
    const long long stop = (hi - base < J ? hi - base : J);          // nothing past hi is needed ...
    const long long stop_raw = want_raw ? raw0 - base + MT_N : 0;    // ... except the raw block
    const long long end = stop > stop_raw ? stop : stop_raw;
    for (long long n = MT_N; n < end; n += MT_B) {
        if (t < MT_B && n + t < J) {
            const long long i = n + t;
            const uint32_t v = mt_twist(ring[(i - MT_N) & 1023], ring[(i - MT_N + 1) & 1023], ring[(i - MT_B) & 1023]);
            ring[i & 1023] = v;
            const long long p = base + i;
            if (p >= lo && p < hi) out[p - lo] = mt_temper(v);
            if (want_raw && p >= raw0 && p < raw0 + MT_N) raw_out[p - raw0] = v;
        }
        __syncthreads();
    }
}

extern "C" int mis_mt_jump(unsigned int* states, int src0, int dst0, int njumps, const unsigned int* g_words, int per_jump_src, int per_jump_mask, int parts,
                           void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(states != nullptr && g_words != nullptr && src0 >= 0 && dst0 >= 0 && njumps > 0 && parts >= 1 && parts <= 64, MIS_EINVAL, "mt_jump: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    MIS_REQUIRE(hipMemsetAsync(states + (size_t)dst0 * MT_N, 0, (size_t)njumps * MT_N * sizeof(uint32_t), s) == hipSuccess, MIS_EHIP, "mt_jump: memset failed");
    const size_t lds = (size_t)MT_JW * sizeof(uint32_t);
    static std::atomic<unsigned long long> attr_done{0};
    if (const int rc = mis_set_dyn_lds(attr_done, reinterpret_cast<const void*>(&mt_jump_kernel), lds, "mt_jump")) return rc;
    hipLaunchKernelGGL(mt_jump_kernel, dim3((unsigned)njumps, (unsigned)parts), dim3(640), lds, s, states, src0, dst0, g_words, per_jump_src, per_jump_mask);
    MIS_LAUNCH_CHECK("mt_jump");
    return MIS_OK;
}

extern "C" int mis_mt_generate(const unsigned int* states, int nchunks, long long J, long long lo, long long hi, unsigned int* out, long long raw_block,
                               unsigned int* raw_out, const unsigned long long* used_dev, long long pos0, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(states != nullptr && nchunks > 0 && J >= MT_N && J % MT_N == 0 && lo >= 0 && hi >= lo && (out != nullptr || hi == lo) &&
                    ((raw_block < 0 && used_dev == nullptr) || raw_out != nullptr),
                MIS_EINVAL, "mt_generate: bad arguments");
    hipLaunchKernelGGL(mt_generate_kernel, dim3((unsigned)nchunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), states, J, lo, hi, out, raw_block, raw_out,
                       used_dev, pos0);
    MIS_LAUNCH_CHECK("mt_generate");
    return MIS_OK;
}

// ---- numpy's legacy_gauss over a word stream, many workgroups: (A) accepted attempts per block of 1024, (B) exclusive scan of the block counts, (C) every block writes
//      its accepted pairs at their global positions; same arithmetic as legacy_normal_kernel (which stays for short streams), same result5 protocol ----
namespace {
constexpr int LN_APB = 1024;                       // attempts per block (256 threads x 4)
__device__ __forceinline__ bool ln_attempt(const uint32_t* __restrict__ words, long long j, double& x1, double& x2, double& r2) {
    const uint32_t w0 = words[4 * j], w1 = words[4 * j + 1], w2 = words[4 * j + 2], w3 = words[4 * j + 3];
    const double d1 = ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) / 9007199254740992.0;
    const double d2 = ((double)(w2 >> 5) * 67108864.0 + (double)(w3 >> 6)) / 9007199254740992.0;
    x1 = 2.0 * d1 - 1.0;
    x2 = 2.0 * d2 - 1.0;
    r2 = x1 * x1 + x2 * x2;
    return r2 < 1.0 && r2 != 0.0;
}
}   // namespace

__global__ __launch_bounds__(256) void ln_count_kernel(const uint32_t* __restrict__ words, long long nattempts, int* __restrict__ counts) {
    __shared__ int red[256];
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long j = (long long)blockIdx.x * LN_APB + threadIdx.x * 4 + k;
        double x1, x2, r2;
        if (j < nattempts && ln_attempt(words, j, x1, x2, r2)) ++c;
    }
    red[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(1024) void ln_scan_kernel(const int* __restrict__ counts, int nblocks, long long* __restrict__ offsets) {
    __shared__ long long part[1024];
    const int t = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    long long s = 0;
    for (int i = t * per; i < (t + 1) * per && i < nblocks; ++i) s += counts[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const long long add = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    long long run = part[t] - s;                   // exclusive prefix of this thread's range
    for (int i = t * per; i < (t + 1) * per && i < nblocks; ++i) {
        offsets[i] = run;
        run += counts[i];
    }
    if (t == 1023) offsets[nblocks] = part[1023];  // total
}

__global__ __launch_bounds__(256) void ln_write_kernel(const uint32_t* __restrict__ words, long long nattempts, const long long* __restrict__ offsets, int nblocks,
                                                        const float* __restrict__ in, float* __restrict__ out, long long count, double scale, int has_gauss,
                                                        double gauss0, unsigned long long* __restrict__ result) {
    __shared__ int scan[256];
    const int t = threadIdx.x;
    const long long first = (has_gauss && count > 0) ? 1 : 0;
    const long long pairs = (count - first + 1) / 2;
    if (blockIdx.x == 0 && t == 0) {
        if (first) out[0] = (float)((double)in[0] + scale * gauss0);
        if (pairs == 0) {       // nothing drawn (count == 0, or the cached value covered a single sample)
            result[0] = 0; result[1] = (has_gauss && count == 0) ? 1ull : 0ull;
            result[2] = (unsigned long long)__double_as_longlong(gauss0); result[4] = 1ull;
        } else if (offsets[nblocks] < pairs) {
            result[4] = 0ull;   // the word stream was too short: the caller retries with more
        }
    }
    if (pairs == 0 || offsets[blockIdx.x] >= pairs) return;         // block-uniform
    double v0[4], v1[4];
    int ok[4], c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long j = (long long)blockIdx.x * LN_APB + t * 4 + k;
        double x1, x2, r2;
        ok[k] = 0;
        if (j < nattempts && ln_attempt(words, j, x1, x2, r2)) {
            const double f = sqrt(-2.0 * log(r2) / r2);
            v0[k] = f * x2;
            v1[k] = f * x1;
            ok[k] = 1;
            ++c;
        }
    }
    scan[t] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int add = (t >= off) ? scan[t - off] : 0;
        __syncthreads();
        scan[t] += add;
        __syncthreads();
    }
    long long p = offsets[blockIdx.x] + scan[t] - c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (ok[k]) {
            if (p < pairs) {
                const long long i0 = first + 2 * p, i1 = i0 + 1;
                out[i0] = (float)((double)in[i0] + scale * v0[k]);
                if (i1 < count) out[i1] = (float)((double)in[i1] + scale * v1[k]);
                if (p == pairs - 1) {               // the last pair: everything after it stays in the generator
                    result[0] = (unsigned long long)((long long)blockIdx.x * LN_APB + t * 4 + k + 1);
                    result[1] = (i1 >= count) ? 1ull : 0ull;
                    result[2] = (unsigned long long)__double_as_longlong(v1[k]);
                    result[4] = 1ull;
                }
            }
            ++p;
        }
    }
}

extern "C" size_t mis_legacy_normal_par_workspace_bytes(long long nattempts) {
    const long long nb = (nattempts + LN_APB - 1) / LN_APB;
    return (size_t)(nb + 1) * sizeof(long long) + (size_t)(nb + 2) * sizeof(int);
}

extern "C" int mis_legacy_normal_par(const unsigned int* words, long long nattempts, const float* in, float* out, long long count, double scale, int has_gauss,
                                     double gauss0, void* workspace, unsigned long long* result5, void* stream) {
    (void)hipGetLastError();
    MIS_REQUIRE(words != nullptr && result5 != nullptr && workspace != nullptr && count >= 0 && nattempts > 0 && (count == 0 || (in != nullptr && out != nullptr)),
                MIS_EINVAL, "legacy_normal_par: bad arguments");
    const long long nb = (nattempts + LN_APB - 1) / LN_APB;
    MIS_REQUIRE(nb < (1ll << 30), MIS_EUNSUPPORTED, "legacy_normal_par: too many attempts");
    long long* offsets = reinterpret_cast<long long*>(workspace);
    int* counts = reinterpret_cast<int*>(offsets + nb + 1);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ln_count_kernel, dim3((unsigned)nb), dim3(256), 0, s, words, nattempts, counts);
    MIS_LAUNCH_CHECK("ln_count");
    hipLaunchKernelGGL(ln_scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)counts, (int)nb, offsets);
    MIS_LAUNCH_CHECK("ln_scan");
    hipLaunchKernelGGL(ln_write_kernel, dim3((unsigned)nb), dim3(256), 0, s, words, nattempts, (const long long*)offsets, (int)nb, in, out, count, scale, has_gauss,
                       gauss0, result5);
    MIS_LAUNCH_CHECK("ln_write");
    return MIS_OK;
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
