// comm.hip -- the one collective of the path, natively over RCCL (xGMI) for C++ hosts.
// The reference has no communication layer at all; multi-GPU StoCS shards
// independent trial streams / candidate batches one per GPU and combines them with ONE 8-byte max
// all-reduce of the packed (score, global candidate id) key -- the arg-max of compute_best_transform
// (reference src/stocs.cpp:982-1004) across ranks, lowest id winning ties -- followed by a 64-byte
// broadcast of the winner's camera-frame pose.  Latency-bound (<= 600 B per GPU): xGMI link bandwidth
// is irrelevant here.
// librccl is opened lazily with dlopen so that libstocs_hip.so keeps working where RCCL is absent;
// Python callers use torch.distributed (backend "nccl" == RCCL) through model_matching_amd/dist.py.
#include <dlfcn.h>

#include <mutex>
#include <rccl/rccl.h>
#include <string.h>

#include "stocs_ctx.h"

namespace stocs {

struct Rccl {
    void* lib;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    const char* (*GetErrorString)(ncclResult_t);
};

static Rccl* rccl() {
    static Rccl r;
    static int state = 0;  // 0 untried, 1 ok, -1 failed
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (state == 0) {
        r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        state = -1;
        if (r.lib) {
            r.GetUniqueId = (ncclResult_t(*)(ncclUniqueId*))dlsym(r.lib, "ncclGetUniqueId");
            r.CommInitRank = (ncclResult_t(*)(ncclComm_t*, int, ncclUniqueId, int))dlsym(r.lib, "ncclCommInitRank");
            r.CommDestroy = (ncclResult_t(*)(ncclComm_t))dlsym(r.lib, "ncclCommDestroy");
            r.AllReduce = (ncclResult_t(*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t))dlsym(r.lib, "ncclAllReduce");
            r.Broadcast = (ncclResult_t(*)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t))dlsym(r.lib, "ncclBroadcast");
            r.GetErrorString = (const char* (*)(ncclResult_t))dlsym(r.lib, "ncclGetErrorString");
            if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Broadcast) state = 1;
        }
    }
    return state == 1 ? &r : NULL;
}

#define STOCS_NCCL_CHECK(expr)                                                                        \
    do {                                                                                              \
        ncclResult_t _r = (expr);                                                                     \
        if (_r != ncclSuccess) {                                                                      \
            set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, R->GetErrorString ? R->GetErrorString(_r) : "rccl error"); \
            return STOCS_ERR_HIP;                                                                     \
        }                                                                                             \
    } while (0)

}  // namespace stocs

using namespace stocs;

struct stocs_comm {
    ncclComm_t comm;
    int rank, nranks;
    unsigned long long* d_key;  // 8 bytes
    float* d_pose;              // 16 floats
};

extern "C" {

int stocs_comm_unique_id(void* id128) {
    if (!id128) return STOCS_ERR_INVALID;
    Rccl* R = rccl();
    if (!R) { set_error("librccl.so could not be loaded"); return STOCS_ERR_NO_DEVICE; }
    ncclUniqueId id;
    STOCS_NCCL_CHECK(R->GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return STOCS_OK;
}

int stocs_comm_create(const void* id128, int nranks, int rank, int device, stocs_comm** out) {
    if (!id128 || !out || nranks <= 0 || rank < 0 || rank >= nranks) return STOCS_ERR_INVALID;
    Rccl* R = rccl();
    if (!R) { set_error("librccl.so could not be loaded"); return STOCS_ERR_NO_DEVICE; }
    if (device >= 0) STOCS_HIP_CHECK(hipSetDevice(device));
    stocs_comm* c = new stocs_comm();
    c->rank = rank; c->nranks = nranks; c->d_key = NULL; c->d_pose = NULL; c->comm = NULL;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = R->CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { set_error("ncclCommInitRank failed: %s", R->GetErrorString ? R->GetErrorString(r) : "?"); delete c; return STOCS_ERR_HIP; }
    if (dev_malloc((void**)&c->d_key, 8) != hipSuccess || dev_malloc((void**)&c->d_pose, 64) != hipSuccess) {
        set_error("hipMalloc failed");
        (void)stocs_comm_destroy(c);   // gives back the communicator and whichever buffer exists
        return STOCS_ERR_HIP;
    }
    *out = c;
    return STOCS_OK;
}

int stocs_comm_destroy(stocs_comm* c) {
    if (!c) return STOCS_OK;
    Rccl* R = rccl();
    if (R && c->comm) R->CommDestroy(c->comm);
    if (c->d_key) (void)hipFree(c->d_key);
    if (c->d_pose) (void)hipFree(c->d_pose);
    delete c;
    return STOCS_OK;
}

// key_inout: this rank's stocs_pack_best(best_lcp, global id) (0 = nothing); pose16_inout: this rank's best
// camera-frame pose.  global ids must encode the owner as  id / ids_per_rank == rank.
int stocs_allreduce_best(stocs_comm* c, void* hip_stream, uint64_t* key_inout, float* pose16_inout, uint32_t ids_per_rank) {
    if (!c || !key_inout || !pose16_inout || ids_per_rank == 0) return STOCS_ERR_INVALID;
    Rccl* R = rccl();
    if (!R) { set_error("librccl.so could not be loaded"); return STOCS_ERR_NO_DEVICE; }
    hipStream_t st = (hipStream_t)hip_stream;
    STOCS_HIP_CHECK(hipMemcpyAsync(c->d_key, key_inout, 8, hipMemcpyHostToDevice, st));
    STOCS_NCCL_CHECK(R->AllReduce(c->d_key, c->d_key, 1, ncclUint64, ncclMax, c->comm, st));   // 8 bytes over xGMI
    uint64_t best = 0;
    STOCS_HIP_CHECK(hipMemcpyAsync(&best, c->d_key, 8, hipMemcpyDeviceToHost, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    if ((best >> 32) == 0) best = 0;   // score bits 0: a key packed by hand from a zero score is "no pose" too (stocs.cpp:987-998)
    *key_inout = best;
    if (best == 0) { memset(pose16_inout, 0, 64); return STOCS_OK; }   // no pose anywhere (Q18)
    float lcp; uint32_t gid;
    stocs_unpack_best(best, &lcp, &gid);
    const int owner = (int)(gid / ids_per_rank);
    if (owner < 0 || owner >= c->nranks) { set_error("winning id %u does not map to a rank", gid); return STOCS_ERR_INVALID; }
    STOCS_HIP_CHECK(hipMemcpyAsync(c->d_pose, pose16_inout, 64, hipMemcpyHostToDevice, st));
    STOCS_NCCL_CHECK(R->Broadcast(c->d_pose, c->d_pose, 16, ncclFloat32, owner, c->comm, st));   // 64 bytes from the owner
    STOCS_HIP_CHECK(hipMemcpyAsync(pose16_inout, c->d_pose, 64, hipMemcpyDeviceToHost, st));
    STOCS_HIP_CHECK(hipStreamSynchronize(st));
    return STOCS_OK;
}

}  // extern "C"
