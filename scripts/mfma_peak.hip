// Diagnostic (GPU box; built and run by scripts/mfma_peak.sh): what the matrix pipe of THIS part delivers when nothing but MFMAs is issued - register operands only, no LDS, no
// memory - for the three instructions the engines use, at 1 and 2 waves per SIMD, with the shader clock and socket power read from the card's hwmon files in the middle of
// each run.  It answers two questions the roofline keeps raising: (1) how far below the nominal dense peak (2.5 PFLOP/s bf16, 157.3 TFLOP/s f32 at 2.4 GHz) a pure MFMA stream
// lands under this part's clock / power management, (2) whether back-to-back f32 MFMAs retire at one per 32 cycles.  Every wave runs a fixed trip count: the grid drains.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cctype>
#include <climits>
#include <cstdlib>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include <glob.h>

typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
// RANDOM: the operands are fresh pseudo-random values in (-2, 2) every iteration (an LCG on the raw bits, sign and mantissa random, exponent pinned): the matrix pipe then
// toggles like it does on real activations - with constant operands it draws a quarter of the power and never meets the part's power management
__device__ __forceinline__ unsigned lcg(unsigned& st) {
    st = st * 1664525u + 1013904223u;
    return st;
}
__device__ __forceinline__ unsigned rnd_bf16_pair(unsigned& st) { return (lcg(st) & 0x807f807fu) | 0x3f803f80u; }          // two bf16 in +-[1, 2)
__device__ __forceinline__ float rnd_f32(unsigned& st) { return __uint_as_float((lcg(st) & 0x807fffffu) | 0x3f800000u); }   // +-[1, 2)

template <int KIND, bool RANDOM>
__global__ __launch_bounds__(512) void mfma_loop(int iters, float* sink) {
    const float seed = (float)(threadIdx.x & 7) * 0.125f;
    unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    // eight operand pairs in registers, set up ONCE: no vector instruction besides the MFMAs inside the loop.  RANDOM: eight different pseudo-random pairs, so that consecutive
    // MFMAs see different operands (the pipe's inputs toggle as on real data); otherwise all eight hold the same constants
    if constexpr (KIND == 0) {
        f32x4 acc[8];
        float A[8], B[8];
        for (int i = 0; i < 8; ++i) {
            acc[i] = f32x4{seed, seed, seed, seed};
            A[i] = RANDOM ? rnd_f32(st) : seed + 1.f;
            B[i] = RANDOM ? rnd_f32(st) : seed + 2.f;
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(i + r) & 7], B[(3 * i + r) & 7], acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
        if (s == 123.456f) sink[0] = s;
    } else if constexpr (KIND == 1) {
        f32x4 acc[8];
        bf16x8 A[8], B[8];
        for (int i = 0; i < 8; ++i) {
            acc[i] = f32x4{seed, seed, seed, seed};
            u32x4_t ua, ub;
            for (int k = 0; k < 4; ++k) {
                ua[k] = RANDOM ? rnd_bf16_pair(st) : 0x3f803f80u;
                ub[k] = RANDOM ? rnd_bf16_pair(st) : 0x3fc03fc0u;
            }
            A[i] = __builtin_bit_cast(bf16x8, ua);
            B[i] = __builtin_bit_cast(bf16x8, ub);
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[(i + r) & 7], B[(3 * i + r) & 7], acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
        if (s == 123.456f) sink[0] = s;
    } else {
        f32x16 acc[4];
        bf16x8 A[8], B[8];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = seed;
        for (int i = 0; i < 8; ++i) {
            u32x4_t ua, ub;
            for (int k = 0; k < 4; ++k) {
                ua[k] = RANDOM ? rnd_bf16_pair(st) : 0x3f803f80u;
                ub[k] = RANDOM ? rnd_bf16_pair(st) : 0x3fc03fc0u;
            }
            A[i] = __builtin_bit_cast(bf16x8, ua);
            B[i] = __builtin_bit_cast(bf16x8, ub);
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[(i + 2 * r) & 7], B[(3 * i + r) & 7], acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
        if (s == 123.456f) sink[0] = s;
    }
}

static std::string hwmon_dir() {
    // the hwmon directory of the card HIP device 0 is (several cards of the host may expose one): match the PCI address
    char bus[64] = {0};
    (void)hipDeviceGetPCIBusId(bus, sizeof(bus), 0);
    std::string want(bus);
    for (auto& c : want) c = (char)tolower(c);
    glob_t g;
    std::string out;
    if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", 0, nullptr, &g) == 0) {
        for (size_t i = 0; i < g.gl_pathc; ++i) {
            std::string f = g.gl_pathv[i];
            std::string card = f.substr(0, f.find("/hwmon/"));
            char real[4096];
            if (realpath(card.c_str(), real) != nullptr) {
                std::string r(real);
                if (want.empty() || r.size() >= want.size() && r.compare(r.size() - want.size(), want.size(), want) == 0 || g.gl_pathc == 1) {
                    out = f.substr(0, f.rfind('/'));
                    if (r.size() >= want.size() && r.compare(r.size() - want.size(), want.size(), want) == 0) break;
                }
            }
        }
    }
    globfree(&g);
    return out;
}
static double read_num(const std::string& p) {
    std::ifstream f(p);
    double v = 0;
    f >> v;
    return v;
}

template <int KIND, bool RANDOM> static void run(const char* name, double flop_per_mfma, int mfma_per_iter, int threads, const std::string& hw, float* sink) {
    const int blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int iters = 20000;
    hipLaunchKernelGGL((mfma_loop<KIND, RANDOM>), dim3(blocks), dim3(threads), 0, 0, 2000, sink);          // warm-up
    hipDeviceSynchronize();
    // size the run to ~300 ms
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<KIND, RANDOM>), dim3(blocks), dim3(threads), 0, 0, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * 400.0 / ms);
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<KIND, RANDOM>), dim3(blocks), dim3(threads), 0, 0, iters, sink);
    hipEventRecord(e1);
    // hwmon readings every 10 ms from the launch until the kernel has finished; the middle half of them is averaged (the ends see the ramp and the idle tail)
    std::vector<double> fm, pw;
    while (hipEventQuery(e1) == hipErrorNotReady) {
        if (!hw.empty()) {
            fm.push_back(read_num(hw + "/freq1_input") / 1e6);
            pw.push_back(read_num(hw + "/power1_input") / 1e6);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    double mhz = 0, watt = 0;
    int n = 0;
    for (size_t k = fm.size() / 4; k < fm.size() - fm.size() / 4; ++k) {
        mhz += fm[k];
        watt += pw[k];
        ++n;
    }
    const double waves = (double)blocks * threads / 64.0;
    const double tflops = waves * (double)iters * mfma_per_iter * flop_per_mfma / (ms * 1e-3) / 1e12;
    if (n) { mhz /= n; watt /= n; }
    const double cyc = n ? (ms * 1e-3 * mhz * 1e6) / ((double)iters * mfma_per_iter * (threads / 256.0)) : 0.0;          // shader cycles per MFMA and SIMD
    printf("%-28s %d waves/SIMD  %7.1f ms  %8.1f TFLOP/s  sclk %6.0f MHz  socket %6.0f W (%d readings)  %5.2f cycles per MFMA and SIMD at that clock\n", name, threads / 256, ms, tflops, mhz, watt, n, cyc);
}

int main() {
    float* sink = nullptr;
    hipMalloc(&sink, 64);
    const std::string hw = hwmon_dir();
    printf("# hwmon: %s\n", hw.c_str());
    printf("# 256 blocks (one per CU), register operands only (eight operand pairs held in registers: all equal constants, or eight different pseudo-random pairs - no vector instruction but the MFMAs in the loop); nominal dense peaks at 2.4 GHz: f32 157.3 TFLOP/s (32 cycles per 16x16x4 MFMA), bf16 2516 TFLOP/s (16 cycles per 16x16x32)\n");
    for (int threads : {256, 512}) {
        run<0, false>("f32_16x16x4 constant", 2.0 * 16 * 16 * 4, 32, threads, hw, sink);
        run<0, true>("f32_16x16x4 random", 2.0 * 16 * 16 * 4, 32, threads, hw, sink);
        run<1, false>("bf16_16x16x32 constant", 2.0 * 16 * 16 * 32, 32, threads, hw, sink);
        run<1, true>("bf16_16x16x32 random", 2.0 * 16 * 16 * 32, 32, threads, hw, sink);
        run<2, false>("bf16_32x32x16 constant", 2.0 * 32 * 32 * 16, 16, threads, hw, sink);
        run<2, true>("bf16_32x32x16 random", 2.0 * 32 * 32 * 16, 16, threads, hw, sink);
    }
    return 0;
}
