// Native RCCL call site for the data-parallel gradient exchange (SURVEY.md section 8b lists pasn_allreduce; section 8e: ONE all-reduce of
// the flat fp32 gradient bucket per optimizer step, over xGMI).  The reference has no distributed code at all, so
// there is nothing to mirror: this is the exchange of protoasnet_amd/dp.py without torch.distributed in the data path.
//
// librccl.so is resolved at RUN TIME (dlopen), not at link time: the process already holds torch's copy of the library, and a second
// link-time copy from /opt/rocm would put two RCCLs with the same symbol names into one process.  dlopen of the same soname returns
// the handle torch loaded, so there is exactly one.  A process that never calls pasn_comm_* never touches RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only
#include <string.h>

#include <mutex>

#include "common.h"

namespace pasn {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    const char* (*GetErrorString)(ncclResult_t);
    bool ok = false;
};

static RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.AllReduce && api.CommDestroy && api.GetErrorString;
    });
    return api;
}

static int rccl_fail(const char* what, ncclResult_t r) {
    set_error(std::string(what) + ": " + rccl().GetErrorString(r));
    return PASN_ERR_LAUNCH;
}

}  // namespace pasn

using namespace pasn;

extern "C" int pasn_comm_unique_id(void* id_out) {
    PASN_REQUIRE(id_out, "null pointer");
    PASN_REQUIRE(rccl().ok, "librccl.so could not be loaded");
    static_assert(sizeof(ncclUniqueId) == PASN_COMM_ID_BYTES, "ncclUniqueId size");
    const ncclResult_t r = rccl().GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_out));
    return r == ncclSuccess ? PASN_OK : rccl_fail("ncclGetUniqueId", r);
}

extern "C" int pasn_comm_init(const void* id, int world_size, int rank, void** comm_out) {
    PASN_REQUIRE(id && comm_out && world_size > 0 && rank >= 0 && rank < world_size, "bad arguments");
    PASN_REQUIRE(rccl().ok, "librccl.so could not be loaded");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = rccl().CommInitRank(&comm, world_size, uid, rank);  // on the calling thread's current HIP device
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    *comm_out = comm;
    return PASN_OK;
}

extern "C" int pasn_allreduce(void* comm, void* buf, size_t count, int dtype, void* stream) {
    PASN_REQUIRE(comm && buf && count > 0, "bad arguments");
    PASN_REQUIRE(dtype == PASN_F32 || dtype == PASN_BF16, "dtype must be PASN_F32 or PASN_BF16");
    // in place, sum, asynchronous on the caller's stream (the gradient bucket is fp32; bf16 is there for activations-sized exchanges)
    const ncclResult_t r = rccl().AllReduce(buf, buf, count, dtype == PASN_F32 ? ncclFloat32 : ncclBfloat16, ncclSum,
                                            static_cast<ncclComm_t>(comm), (hipStream_t)stream);
    return r == ncclSuccess ? PASN_OK : rccl_fail("ncclAllReduce", r);
}

extern "C" int pasn_comm_destroy(void* comm) {
    if (!comm) return PASN_OK;
    const ncclResult_t r = rccl().CommDestroy(static_cast<ncclComm_t>(comm));
    return r == ncclSuccess ? PASN_OK : rccl_fail("ncclCommDestroy", r);
}
