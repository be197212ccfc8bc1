// Gradient exchange from inside the library ("mmdgan_allreduce_bucket (RCCL)" and the communicator are part of
// the handle): RCCL is bound at run time with dlopen / dlsym - the library has no link-time dependency on it and loads on a
// machine without it (then these entries return MMDGAN_E_UNSUPPORTED).  In a process that has imported torch, "librccl.so.1"
// resolves to the copy torch already mapped (same SONAME), so both paths share one RCCL.
// A recorded step (mmdgan_plan_*) takes these collectives as ordinary nodes: a data-parallel step then replays from ONE C
// call, without the segment cuts a torch.distributed collective needs.
#include <dlfcn.h>

#include "common.h"

namespace mmdgan {
namespace {
typedef struct { char internal[128]; } UniqueId;         // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *Comm;                                      // ncclComm_t
constexpr int kNcclFloat32 = 7, kNcclSum = 0;            // ncclFloat32, ncclSum (rccl.h)
struct Api {
    void *lib = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Api &api() {
    static Api a;
    static bool tried = false;
    if (tried) return a;
    tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (a.lib) break;
    }
    if (!a.lib) return a;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce;
    return a;
}
// one communicator per thread-current handle would need the handle's definition here; the exchange is per process
// (one replica = one process = one GPU), so the communicator is process-wide and the handle only records its use
Comm g_comm = nullptr;
int g_nranks = 0;
unsigned g_generation = 1;
}  // namespace
unsigned comm_generation() { return g_generation; }
}  // namespace mmdgan

using namespace mmdgan;

extern "C" int mmdgan_comm_unique_id(void *out128) {
    MMDGAN_REQUIRE(out128, "comm_unique_id: null pointer");
    Api &a = api();
    if (!a.ok) { set_error("comm_unique_id: librccl is not available"); return MMDGAN_E_UNSUPPORTED; }
    UniqueId id;
    const int rc = a.GetUniqueId(&id);
    if (rc != 0) { set_error("ncclGetUniqueId: %s", a.GetErrorString ? a.GetErrorString(rc) : "error"); return MMDGAN_E_LAUNCH; }
    memcpy(out128, &id, sizeof(id));
    return MMDGAN_OK;
}

extern "C" int mmdgan_comm_init(const void *id128, int nranks, int rank) {
    MMDGAN_REQUIRE(id128 && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad arguments (nranks %d, rank %d)", nranks, rank);
    MMDGAN_REQUIRE(!g_comm, "comm_init: a communicator exists already (mmdgan_comm_destroy first)");
    Api &a = api();
    if (!a.ok) { set_error("comm_init: librccl is not available"); return MMDGAN_E_UNSUPPORTED; }
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    const int rc = a.CommInitRank(&g_comm, nranks, id, rank);
    if (rc != 0) { g_comm = nullptr; set_error("ncclCommInitRank: %s", a.GetErrorString ? a.GetErrorString(rc) : "error"); return MMDGAN_E_LAUNCH; }
    g_nranks = nranks;
    ++g_generation;
    return MMDGAN_OK;
}

extern "C" int mmdgan_comm_destroy(void) {
    // recorded plans hold the communicator by value: they refuse to replay once the generation has moved on
    if (g_comm) { (void)api().CommDestroy(g_comm); g_comm = nullptr; g_nranks = 0; ++g_generation; }
    return MMDGAN_OK;
}

extern "C" int mmdgan_comm_size(void) { return g_nranks; }

// in-place SUM all-reduce of `count` floats on `stream` (averaging is Adam's grad_scale = 1/world)
extern "C" int mmdgan_allreduce_bucket(float *buf, size_t count, void *stream) {
    MMDGAN_REQUIRE(buf && count >= 1, "allreduce_bucket: bad arguments");
    MMDGAN_REQUIRE(g_comm, "allreduce_bucket: no communicator (mmdgan_comm_init)");
    Api &a = api();
    Comm comm = g_comm;
    hipStream_t st = (hipStream_t)stream;
    const int rc = a.AllReduce(buf, buf, count, kNcclFloat32, kNcclSum, comm, st);
    if (rc != 0) { set_error("ncclAllReduce: %s", a.GetErrorString ? a.GetErrorString(rc) : "error"); return MMDGAN_E_LAUNCH; }
    if (plan_recording()) {
        auto fn = a.AllReduce;
        plan_note_collective();
        plan_push([=]() { (void)fn(buf, buf, count, kNcclFloat32, kNcclSum, comm, st); });
    }
    return MMDGAN_OK;
}
