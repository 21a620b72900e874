// Data-parallel gradient exchange of the C ABI: all_reduce(SUM) over RCCL (xGMI inside a node).
//
// padertorch/train/trainer.py:396-442 replicates the model over the devices of one process and SUMS the replicas' gradients
// (`:426-428`: accumulated, not averaged).  Here every GPU has its own process; a host that binds only
// this C ABI exchanges its flat gradient bucket through these four entry points (the Python Trainer of this package uses
// torch.distributed's "nccl" backend, which is the same RCCL).  librccl is opened on first use (dlopen): a single-GPU host
// never needs it, and libptmi.so has no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "common.h"

namespace ptmi {
namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*get_version)(int*) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        // a copy that is already in the process (torch ships one) is reused by soname; otherwise the ROCm installation's
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x.handle) break;
        }
        if (!x.handle) return x;
        x.get_unique_id = reinterpret_cast<decltype(x.get_unique_id)>(dlsym(x.handle, "ncclGetUniqueId"));
        x.comm_init_rank = reinterpret_cast<decltype(x.comm_init_rank)>(dlsym(x.handle, "ncclCommInitRank"));
        x.all_reduce = reinterpret_cast<decltype(x.all_reduce)>(dlsym(x.handle, "ncclAllReduce"));
        x.comm_destroy = reinterpret_cast<decltype(x.comm_destroy)>(dlsym(x.handle, "ncclCommDestroy"));
        x.get_version = reinterpret_cast<decltype(x.get_version)>(dlsym(x.handle, "ncclGetVersion"));
        x.ok = x.get_unique_id && x.comm_init_rank && x.all_reduce && x.comm_destroy;
        return x;
    }();
    return r;
}

// RCCL's result codes are reported as negative numbers below the library's own (-100 - ncclResult_t)
inline int rc(ncclResult_t r) { return r == ncclSuccess ? PTMI_OK : -100 - (int)r; }

}  // namespace
}  // namespace ptmi

using namespace ptmi;

struct ptmi_comm {
    ncclComm_t comm;
    int world, rank;
};

extern "C" {

int32_t ptmi_comm_rccl_version(void) {
    Rccl& r = rccl();
    int v = 0;
    if (!r.ok || !r.get_version || r.get_version(&v) != ncclSuccess) return 0;
    return v;
}

int ptmi_comm_unique_id(uint8_t* id_out) {
    PTMI_RETURN_IF(!id_out, PTMI_E_INVALID);
    Rccl& r = rccl();
    PTMI_RETURN_IF(!r.ok, PTMI_E_UNSUPPORTED);
    static_assert(sizeof(ncclUniqueId) == PTMI_COMM_ID_BYTES, "ptmi.h's id size must be RCCL's");
    ncclUniqueId id;
    const ncclResult_t e = r.get_unique_id(&id);
    if (e != ncclSuccess) return rc(e);
    memcpy(id_out, &id, sizeof(id));
    return PTMI_OK;
}

int ptmi_comm_create(ptmi_comm** comm, int32_t world_size, int32_t rank, const uint8_t* id) {
    PTMI_RETURN_IF(!comm || !id || world_size < 1 || rank < 0 || rank >= world_size, PTMI_E_INVALID);
    Rccl& r = rccl();
    PTMI_RETURN_IF(!r.ok, PTMI_E_UNSUPPORTED);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    const ncclResult_t e = r.comm_init_rank(&c, world_size, uid, rank);        // binds the CURRENT device, like every RCCL communicator
    if (e != ncclSuccess) return rc(e);
    *comm = new ptmi_comm{c, world_size, rank};
    return PTMI_OK;
}

int ptmi_allreduce_sum(ptmi_comm* comm, float* buffer, int64_t n, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!comm || (!buffer && n > 0) || n < 0, PTMI_E_INVALID);
    if (n == 0) return PTMI_OK;
    // in place, fp32, SUM - no division by the world size (trainer.py:426-428)
    return rc(rccl().all_reduce(buffer, buffer, (size_t)n, ncclFloat32, ncclSum, comm->comm, static_cast<hipStream_t>(stream)));
}

int ptmi_comm_destroy(ptmi_comm* comm) {
    if (!comm) return PTMI_OK;
    const ncclResult_t e = rccl().comm_destroy(comm->comm);
    delete comm;
    return rc(e);
}

}  // extern "C"
