// The dispatch-switch table behind dispatch_cfg.hpp (see there).  A switch that is present in the environment with a numeric value takes that value, present
// with any other text counts as 1, absent = its default.
#include "dispatch_cfg.hpp"

#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "common.hpp"

namespace {
struct SwDef {
    const char* name;
    int dflt;
};
// order = enum MisSwitch
const SwDef g_defs[SW_COUNT] = {
    {"MIS_CONV_V1", 0}, {"MIS_CONV_V3", 0}, {"MIS_CONV_NOPP", 0}, {"MIS_CONV_PP64", 0}, {"MIS_CONV_PPC64", 0}, {"MIS_CONV_K3_NO256", 0},
    {"MIS_CONV_K3_256_MINCIN", 256}, {"MIS_CONV_NODMA", 0}, {"MIS_CONV_NOWS64", 0},
    {"MIS_CONV3D_BN64V1", 0}, {"MIS_CONV_K1V1", 0}, {"MIS_CONV_K1NOPERSIST", 0}, {"MIS_CONV_K1_NO256", 0},
    {"MIS_CONV_PPC_COLMAJOR", 0}, {"MIS_CONV_RS64", 0}, {"MIS_CONV_NOPPC", 0}, {"MIS_CONV_PPC", 0}, {"MIS_CONV_PP_NO256", 0},
    {"MIS_CONV3D_NOPP", 0}, {"MIS_CONV3D_PF", 0}, {"MIS_CONV3D_ZG", 4}, {"MIS_CONV3D_COLMAJOR", 0},
    {"MIS_WGRAD_K1_NARROW", 0}, {"MIS_WGRAD_NO_TR", 0}, {"MIS_WGRAD_BLOCKS", 1024}, {"MIS_WGRAD_NOPP", 0}, {"MIS_WGRAD_PP_NOWIDE", 0},
    {"MIS_WGRAD_PP_KSS1", 0}, {"MIS_WGRAD3D_NOPP", 0}, {"MIS_WGRAD_PP_ROW", 0}, {"MIS_WGRAD_PP_NOROW", 0},
    {"MIS_FIRST2D_UNTILED", 0}, {"MIS_FIRST3D_UNTILED", 0}, {"MIS_UPCONV_BWD_GENERIC", 0}, {"MIS_GEMM1_NOPP", 0}, {"MIS_CONV_NOPPD", 0}, {"MIS_WGRAD_PP_NOSTREAM", 0}, {"MIS_WGRAD_K1_NOPP", 0}, {"MIS_FIRST3D_NOMFMA", 0}, {"MIS_PERSIST_CUS", 256}, {"MIS_HEAD_UNFUSED", 0},
    {"MIS_CONV3D_F32_NOPP", 0}, {"MIS_WGRAD_F32_NOPP", 0}, {"MIS_WGRAD_F32_ROUNDS", 1}, {"MIS_CONV_PPS", 0}, {"MIS_CONV_PPC2", 0}, {"MIS_TILEQ_OFF", 0}, {"MIS_CONV3D_NOPF10N4", 0}, {"MIS_CONV3D_F32_WIDE", 0},
};
std::atomic<int> g_val[SW_COUNT];
std::once_flag g_once;

int from_env(int k) {
    const char* e = getenv(g_defs[k].name);
    if (e == nullptr) return g_defs[k].dflt;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    return (end != e && *end == '\0') ? (int)v : 1;
}
void parse_all() {
    for (int k = 0; k < SW_COUNT; ++k) g_val[k].store(from_env(k), std::memory_order_relaxed);
}
}   // namespace

int mis_sw(MisSwitch k) {
    std::call_once(g_once, parse_all);
    return g_val[k].load(std::memory_order_relaxed);
}

int mis_persist_cus() {
    const int v = mis_sw(SW_PERSIST_CUS);
    return v < 8 ? 8 : (v > 256 ? 256 : v);
}

// value >= 0: the switch takes `value` for the rest of the process (or until the next override); value < 0: back to the environment's value / the default.
// name == NULL: reset every switch.  Unknown name: MIS_EINVAL.
extern "C" int mis_dispatch_override(const char* name, int value) {
    std::call_once(g_once, parse_all);
    if (name == nullptr) {
        parse_all();
        return MIS_OK;
    }
    for (int k = 0; k < SW_COUNT; ++k)
        if (strcmp(name, g_defs[k].name) == 0) {
            g_val[k].store(value >= 0 ? value : from_env(k), std::memory_order_relaxed);
            return MIS_OK;
        }
    mis_set_error("mis_dispatch_override: unknown switch '%s'", name);
    return MIS_EINVAL;
}

// 1 when the library was built with `make EXPERIMENTS=1` (csrc/experiments/*.hip: measured-and-lost kernel variants kept as evidence; MIS_CONV_PPS / MIS_CONV_PPC2 select them)
extern "C" int mis_build_has_experiments(void) {
#ifdef MIS_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int mis_dispatch_switch(const char* name) {
    std::call_once(g_once, parse_all);
    if (name != nullptr)
        for (int k = 0; k < SW_COUNT; ++k)
            if (strcmp(name, g_defs[k].name) == 0) return g_val[k].load(std::memory_order_relaxed);
    mis_set_error("mis_dispatch_switch: unknown switch '%s'", name ? name : "(null)");
    return MIS_EINVAL;
}

// ---- tile-queue counter blocks (dispatch_cfg.hpp) --------------------------------------------------------------------------------------------------------
namespace {
constexpr int TQ_SLOTS = 256, TQ_BYTES = 512, TQ_MAXDEV = 16;
std::mutex g_tq_mu;
struct TqDev {
    unsigned char* pool = nullptr;          // TQ_SLOTS counter blocks in this device's memory, zeroed once
    bool failed = false;
    std::unordered_map<unsigned long long, int> slot;          // stream handle (eager launches) or capture id (captured ones) -> block
};
TqDev g_tq[TQ_MAXDEV];

// the pool of this device: one allocation for every slot, zeroed once.  Never from inside a stream capture (an allocation / device synchronisation there is a
// capture-unsafe call) - mis_tile_queue_init() makes it at library load time (ops.load), so that the launch path only ever looks it up.
bool tq_make_pool(TqDev& t) {
    if (t.pool != nullptr) return true;
    if (t.failed) return false;
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)TQ_SLOTS * TQ_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (hipMemset(p, 0, (size_t)TQ_SLOTS * TQ_BYTES) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        t.failed = true;
        return false;
    }
    t.pool = static_cast<unsigned char*>(p);
    return true;
}
}   // namespace

extern "C" int mis_tile_queue_init(void) {
    (void)hipGetLastError();
    if (mis_sw(SW_TILEQ_OFF)) return MIS_OK;
    int dev = 0;
    MIS_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < TQ_MAXDEV, MIS_EHIP, "tile_queue_init: no current device");
    std::lock_guard<std::mutex> lk(g_tq_mu);
    MIS_REQUIRE(tq_make_pool(g_tq[dev]), MIS_EHIP, "tile_queue_init: could not allocate the counter pool of device %d", dev);
    return MIS_OK;
}

unsigned* mis_tile_queue(void* stream) {
    if (mis_sw(SW_TILEQ_OFF)) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TQ_MAXDEV) return nullptr;
    // a launch that is being CAPTURED gets the block of its capture (one per captured graph: two graphs replayed concurrently on different streams no longer share
    // counters - ADVICE r5; replays of ONE graph are ordered by the stream they are launched on, as before); an eager launch the block of its stream
    hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    const bool capturing = hipStreamGetCaptureInfo(reinterpret_cast<hipStream_t>(stream), &cst, &cid) == hipSuccess && cst == hipStreamCaptureStatusActive;
    if (!capturing) (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_tq_mu);
    TqDev& t = g_tq[dev];
    if (t.pool == nullptr && (capturing || !tq_make_pool(t))) return nullptr;          // (no pool yet and a capture is running: the static stride, nothing is allocated)
    const unsigned long long key = capturing ? ((1ull << 63) | cid) : (unsigned long long)reinterpret_cast<uintptr_t>(stream);
    auto it = t.slot.find(key);
    if (it == t.slot.end()) {
        if ((int)t.slot.size() >= TQ_SLOTS) return nullptr;          // (more than 256 streams / captured graphs on one device: the further ones run the static stride)
        it = t.slot.emplace(key, (int)t.slot.size()).first;
    }
    return reinterpret_cast<unsigned*>(t.pool + (size_t)it->second * TQ_BYTES);
}

// Start-of-step reset of the stream's (or the running capture's) counter block: 512 bytes, stream-ordered, a memset node inside a captured graph.  The counters reset
// themselves when a launch completes; this makes a step independent of whatever an EARLIER launch left behind (an aborted kernel, a replay that was cut short).
extern "C" int mis_tile_queue_reset(void* stream) {
    (void)hipGetLastError();
    unsigned* q = mis_tile_queue(stream);
    if (q == nullptr) return MIS_OK;
    // the error words (word 1 of each counter line) survive: they are only cleared by mis_tile_queue_errors
    MIS_REQUIRE(hipMemset2DAsync(q, 64, 0, 4, 8, reinterpret_cast<hipStream_t>(stream)) == hipSuccess, MIS_EHIP, "tile_queue_reset: hipMemset2DAsync failed");
    return MIS_OK;
}

// Tickets past a launch's last one that a kernel has seen since the last call (conv_pp_common.hpp tq_tile: such a launch left output tiles unwritten): synchronises the
// device, scans every counter block of its pool, clears the words, returns how many were set (0 = clean) and leaves a message for mis_last_error().
extern "C" int mis_tile_queue_errors(void) {
    (void)hipGetLastError();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TQ_MAXDEV) return 0;
    std::lock_guard<std::mutex> lk(g_tq_mu);
    TqDev& t = g_tq[dev];
    if (t.pool == nullptr) return 0;
    static unsigned h[TQ_SLOTS * TQ_BYTES / 4];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, t.pool, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) {
        mis_set_error("tile_queue_errors: could not read the counter pool: %s", hipGetErrorString(hipGetLastError()));
        return -1;
    }
    int bad = 0, slot0 = -1;
    unsigned tk0 = 0;
    for (int s = 0; s < TQ_SLOTS; ++s)
        for (int x = 0; x < 8; ++x) {
            const unsigned v = h[s * (TQ_BYTES / 4) + x * 16 + 1];
            if (v != 0u) {
                if (bad++ == 0) { slot0 = s; tk0 = v & 0x7fffffffu; }
                (void)hipMemset(t.pool + (size_t)s * TQ_BYTES + (x * 16 + 1) * 4, 0, 4);
            }
        }
    if (bad) mis_set_error("tile queue: %d counter(s) handed out a ticket past their launch's last one (first: block %d, ticket %u) - a launch started on counters that "
                           "were not zero and left output tiles unwritten", bad, slot0, tk0);
    return bad;
}

// diagnostic / tests: the eight counters of `stream`'s block, read back after a device synchronisation (all zero between launches); -1 if the stream has no block
extern "C" int mis_debug_tile_queue(void* stream, unsigned* out8) {
    unsigned* q = mis_tile_queue(stream);
    if (q == nullptr || out8 == nullptr) return -1;
    unsigned h[128];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, q, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    for (int x = 0; x < 8; ++x) out8[x] = h[x * 16];
    return 0;
}

// diagnostic / tests: store `value` to counter `xcd` of `stream`'s block - what a launch that never finished would have left behind (the loud-failure test plants it)
extern "C" int mis_debug_tile_queue_poke(void* stream, int xcd, unsigned value) {
    unsigned* q = mis_tile_queue(stream);
    if (q == nullptr || xcd < 0 || xcd >= 8) return -1;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(q + xcd * 16, &value, 4, hipMemcpyHostToDevice) != hipSuccess) return -1;
    return 0;
}

