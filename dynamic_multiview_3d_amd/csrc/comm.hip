// RCCL behind the C ABI: the gradient exchange of the data-parallel train step.
//
// The reference is single-device (multi_view_model/train.py:21,35); the batch shards naturally over GPUs, and the one
// exchange step is a SUM over ranks of the flat fp32 gradient buffer -- or, in the sharded-optimiser form, a reduce-scatter of
// it followed by an all-gather of the updated parameters.  RCCL is resolved at run time (dlsym in the process image): the host
// program has already loaded ONE HIP runtime and its RCCL (PyTorch-ROCm ships both), and linking a second copy of either is
// what breaks multi-process GPU programs.  No RCCL symbol in the process -> MV3D_E_UNSUPPORTED, loudly.
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void* ncclComm_p;
enum { kNcclSuccess = 0, kNcclFloat32 = 7, kNcclSum = 0 };

struct Rccl {
    int (*GetUniqueId)(ncclUniqueId_t*) = nullptr;
    int (*CommInitRank)(ncclComm_p*, int, ncclUniqueId_t, int) = nullptr;
    int (*CommDestroy)(ncclComm_p) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_p, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        void* h = RTLD_DEFAULT;
        if (!dlsym(h, "ncclCommInitRank")) {
            // not in the image yet: take the library the HIP runtime in use was shipped with (same directory), then the default search path
            Dl_info info;
            void* lib = nullptr;
            if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
                char path[1024];
                strncpy(path, info.dli_fname, sizeof(path) - 16);
                path[sizeof(path) - 16] = 0;
                char* slash = strrchr(path, '/');
                if (slash) { strcpy(slash + 1, "librccl.so"); lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL); }
            }
            if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (lib) h = lib;
        }
#define MV3D_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name))
        MV3D_SYM(GetUniqueId, "ncclGetUniqueId"); MV3D_SYM(CommInitRank, "ncclCommInitRank"); MV3D_SYM(CommDestroy, "ncclCommDestroy");
        MV3D_SYM(AllReduce, "ncclAllReduce"); MV3D_SYM(ReduceScatter, "ncclReduceScatter"); MV3D_SYM(AllGather, "ncclAllGather");
        MV3D_SYM(GetErrorString, "ncclGetErrorString");
#undef MV3D_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.ReduceScatter && r.AllGather;
    }
    return r;
}

int nccl_fail(const char* what, int rc) {
    Rccl& r = rccl();
    return mv3d::fail(MV3D_E_HIP, "%s: RCCL error %d (%s)", what, rc, r.GetErrorString ? r.GetErrorString(rc) : "?");
}

}  // namespace

struct mv3d_comm { ncclComm_p comm; int rank, world; };

extern "C" {

int mv3d_comm_available(void) { return rccl().ok ? 1 : 0; }

int mv3d_comm_unique_id(void* id128) {
    if (!id128) return mv3d::fail(MV3D_E_INVAL, "mv3d_comm_unique_id: null buffer");
    if (!rccl().ok) return mv3d::fail(MV3D_E_UNSUPPORTED, "mv3d_comm_unique_id: no RCCL in this process (librccl.so not found)");
    ncclUniqueId_t id;
    int rc = rccl().GetUniqueId(&id);
    if (rc != kNcclSuccess) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, &id, sizeof(id));
    return MV3D_OK;
}

int mv3d_comm_init(mv3d_comm** out, int rank, int world, const void* id128) {
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return mv3d::fail(MV3D_E_INVAL, "mv3d_comm_init: bad arguments");
    if (!rccl().ok) return mv3d::fail(MV3D_E_UNSUPPORTED, "mv3d_comm_init: no RCCL in this process (librccl.so not found)");
    ncclUniqueId_t id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_p c = nullptr;
    int rc = rccl().CommInitRank(&c, world, id, rank);      // binds to the calling thread's current HIP device
    if (rc != kNcclSuccess) return nccl_fail("ncclCommInitRank", rc);
    *out = new mv3d_comm{c, rank, world};
    return MV3D_OK;
}

int mv3d_comm_destroy(mv3d_comm* c) {
    if (!c) return MV3D_OK;
    int rc = rccl().CommDestroy(c->comm);
    delete c;
    return rc == kNcclSuccess ? MV3D_OK : nccl_fail("ncclCommDestroy", rc);
}

int mv3d_comm_allreduce_sum(mv3d_comm* c, void* buf, int64_t count, void* stream) {
    if (!c || !buf || count <= 0) return mv3d::fail(MV3D_E_INVAL, "mv3d_comm_allreduce_sum: bad arguments");
    int rc = rccl().AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, c->comm, reinterpret_cast<hipStream_t>(stream));
    return rc == kNcclSuccess ? MV3D_OK : nccl_fail("ncclAllReduce", rc);
}

int mv3d_comm_reduce_scatter_sum(mv3d_comm* c, const void* send, void* recv, int64_t recv_count, void* stream) {
    if (!c || !send || !recv || recv_count <= 0) return mv3d::fail(MV3D_E_INVAL, "mv3d_comm_reduce_scatter_sum: bad arguments");
    int rc = rccl().ReduceScatter(send, recv, (size_t)recv_count, kNcclFloat32, kNcclSum, c->comm, reinterpret_cast<hipStream_t>(stream));
    return rc == kNcclSuccess ? MV3D_OK : nccl_fail("ncclReduceScatter", rc);
}

int mv3d_comm_allgather(mv3d_comm* c, const void* send, void* recv, int64_t send_count, void* stream) {
    if (!c || !send || !recv || send_count <= 0) return mv3d::fail(MV3D_E_INVAL, "mv3d_comm_allgather: bad arguments");
    int rc = rccl().AllGather(send, recv, (size_t)send_count, kNcclFloat32, c->comm, reinterpret_cast<hipStream_t>(stream));
    return rc == kNcclSuccess ? MV3D_OK : nccl_fail("ncclAllGather", rc);
}

}  // extern "C"
