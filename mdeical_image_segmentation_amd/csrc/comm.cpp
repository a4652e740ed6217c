// RCCL communicator behind the C ABI (SURVEY.md §8b "one RCCL communicator handle per process created/destroyed by explicit init/finalize calls"; group
// `allreduce_bucket`): the gradient exchange of the data-parallel train step - one in-place fp32 SUM all-reduce per bucket of the flat gradient buffer over
// xGMI - callable from any host language.  Replaces the reference's nn.DataParallel gather/scatter (model/unet3d/trainer.py:23-24) and whatever HF Trainer does
// under torchrun (train.py:140-150 is launched one process per GPU).
//
// librccl is bound at RUN time (dlopen + dlsym: librccl.so.1 / librccl.so - the copy already mapped by the host process wins, e.g. the one PyTorch-ROCm ships), so
// libmisamd.so itself has no link-time RCCL dependency and loads on machines without it.  The unique id (128 bytes) is created by rank 0 with
// mis_comm_unique_id() and must be carried to the other ranks by the host (torch.distributed store, MPI, a file: any channel).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include <mutex>

#include "misamd.h"

void mis_set_error(const char* fmt, ...);

namespace {
struct NcclUniqueId {
    char internal[128];
};
typedef void* NcclComm;
typedef int (*GetUniqueIdFn)(NcclUniqueId*);
typedef int (*CommInitRankFn)(NcclComm*, int, NcclUniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*CommDestroyFn)(NcclComm);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn error_string = nullptr;
};
Rccl g_rccl;
NcclComm g_comm = nullptr;
int g_rank = -1, g_world = 0;
std::mutex g_mu;

int load_rccl() {
    if (g_rccl.handle != nullptr) return MIS_OK;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);       // a copy the process already mapped
        if (h != nullptr) break;
    }
    for (int i = 0; h == nullptr && i < 2; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) {
        mis_set_error("comm: cannot load librccl.so (%s)", dlerror());
        return MIS_EUNSUPPORTED;
    }
    g_rccl.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(h, "ncclCommInitRank"));
    g_rccl.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(h, "ncclAllReduce"));
    g_rccl.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(h, "ncclCommDestroy"));
    g_rccl.error_string = reinterpret_cast<GetErrorStringFn>(dlsym(h, "ncclGetErrorString"));
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy) {
        mis_set_error("comm: librccl.so lacks the ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy entry points");
        return MIS_EUNSUPPORTED;
    }
    g_rccl.handle = h;
    return MIS_OK;
}

int fail(const char* what, int rc) {
    mis_set_error("comm: %s failed: %s (%d)", what, g_rccl.error_string ? g_rccl.error_string(rc) : "?", rc);
    return MIS_EHIP;
}
}   // namespace

extern "C" int mis_comm_unique_id(void* out128) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (out128 == nullptr) {
        mis_set_error("comm: null output");
        return MIS_EINVAL;
    }
    if (const int rc = load_rccl()) return rc;
    NcclUniqueId id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc != 0) return fail("ncclGetUniqueId", rc);
    memcpy(out128, id.internal, sizeof(id.internal));
    return MIS_OK;
}

extern "C" int mis_comm_init(const void* unique_id128, int rank, int world) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (unique_id128 == nullptr || world < 1 || rank < 0 || rank >= world) {
        mis_set_error("comm: bad arguments (rank %d of %d)", rank, world);
        return MIS_EINVAL;
    }
    if (g_comm != nullptr) {
        mis_set_error("comm: a communicator already exists in this process (call mis_comm_finalize first)");
        return MIS_EINVAL;
    }
    if (const int rc = load_rccl()) return rc;
    NcclUniqueId id;
    memcpy(id.internal, unique_id128, sizeof(id.internal));
    const int rc = g_rccl.comm_init_rank(&g_comm, world, id, rank);      // binds to the CURRENT HIP device of the calling thread
    if (rc != 0) {
        g_comm = nullptr;
        return fail("ncclCommInitRank", rc);
    }
    g_rank = rank;
    g_world = world;
    return MIS_OK;
}

extern "C" int mis_comm_world(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_comm != nullptr ? g_world : 0;
}

// (holds g_mu like init / finalize: a finalize on another thread cannot destroy the communicator under an all-reduce that is being enqueued - ADVICE r2;
//  the call only ENQUEUES on `stream`, so the lock is held for microseconds)
extern "C" int mis_allreduce_bucket(float* buf, long long n, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_comm == nullptr) {
        mis_set_error("comm: no communicator (mis_comm_init)");
        return MIS_EINVAL;
    }
    if (buf == nullptr || n < 0) {
        mis_set_error("comm: bad bucket");
        return MIS_EINVAL;
    }
    if (n == 0) return MIS_OK;
    const int rc = g_rccl.all_reduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, g_comm, reinterpret_cast<hipStream_t>(stream));
    if (rc != 0) return fail("ncclAllReduce", rc);
    return MIS_OK;
}

extern "C" int mis_comm_finalize(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_comm == nullptr) return MIS_OK;
    const int rc = g_rccl.comm_destroy(g_comm);
    g_comm = nullptr;
    g_rank = -1;
    g_world = 0;
    if (rc != 0) return fail("ncclCommDestroy", rc);
    return MIS_OK;
}
