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
#include <algorithm>

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

// ---- mesh-direct exchange: peers' buffers mapped by hipIpc, slices pulled over xGMI point to point ----------------------
// xGMI is a full mesh of point-to-point links (7 x ~153 GB/s per GPU): a ring collective is bound by ONE link, while a rank that
// pulls slice r of every peer's buffer directly uses all seven at once.  These entry points are the data plane of that form;
// the handshake (handles over the control plane, who-reads-what-when) lives with the caller (parallel.MeshComm).
int mv3d_ipc_export(const void* ptr, void* handle64, int64_t* offset) {
    if (!ptr || !handle64 || !offset) return mv3d::fail(MV3D_E_INVAL, "mv3d_ipc_export: null argument");
    void* base = nullptr; size_t size = 0;
    hipError_t e = hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &size, const_cast<void*>(ptr));
    if (e != hipSuccess) return mv3d::fail(MV3D_E_HIP, "mv3d_ipc_export: hipMemGetAddressRange: %s", hipGetErrorString(e));
    hipIpcMemHandle_t h;
    static_assert(sizeof(h) <= 64, "handle buffer");
    e = hipIpcGetMemHandle(&h, base);
    if (e != hipSuccess) return mv3d::fail(MV3D_E_HIP, "mv3d_ipc_export: hipIpcGetMemHandle: %s", hipGetErrorString(e));
    memset(handle64, 0, 64);
    memcpy(handle64, &h, sizeof(h));
    *offset = (int64_t)((const char*)ptr - (const char*)base);
    return MV3D_OK;
}
int mv3d_ipc_open(const void* handle64, void** base) {
    if (!handle64 || !base) return mv3d::fail(MV3D_E_INVAL, "mv3d_ipc_open: null argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    hipError_t e = hipIpcOpenMemHandle(base, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return mv3d::fail(MV3D_E_HIP, "mv3d_ipc_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    return MV3D_OK;
}
int mv3d_ipc_close(void* base) {
    if (!base) return MV3D_OK;
    hipError_t e = hipIpcCloseMemHandle(base);
    return e == hipSuccess ? MV3D_OK : mv3d::fail(MV3D_E_HIP, "mv3d_ipc_close: %s", hipGetErrorString(e));
}

}  // extern "C"

namespace {
struct MeshSrcs { const float* p[MV3D_MESH_MAX_RANKS]; int n; };
// dst[i] = srcs[0][i] + srcs[1][i] + ... in rank order: every rank that reduces a slice adds in the same order
__global__ __launch_bounds__(256) void mesh_reduce_kernel(const MeshSrcs s, float* __restrict__ dst, int64_t count) {
    const int64_t nvec = count >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float4 a = reinterpret_cast<const float4*>(s.p[0])[i];
        for (int r = 1; r < s.n; ++r) {
            const float4 b = reinterpret_cast<const float4*>(s.p[r])[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4*>(dst)[i] = a;
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        const int64_t i = (nvec << 2) + threadIdx.x;
        float a = s.p[0][i];
        for (int r = 1; r < s.n; ++r) a += s.p[r][i];
        dst[i] = a;
    }
}
}  // namespace

extern "C" {

int mv3d_mesh_reduce_sum(const void* const* srcs, int nsrc, void* dst, int64_t count, void* stream) {
    if (!srcs || nsrc < 1 || nsrc > MV3D_MESH_MAX_RANKS || !dst || count <= 0) return mv3d::fail(MV3D_E_INVAL, "mv3d_mesh_reduce_sum: bad arguments (at most %d ranks)", MV3D_MESH_MAX_RANKS);
    MeshSrcs s = {};
    s.n = nsrc;
    for (int r = 0; r < nsrc; ++r) {
        if (!srcs[r] || ((uintptr_t)srcs[r] & 15)) return mv3d::fail(MV3D_E_INVAL, "mv3d_mesh_reduce_sum: source %d null or not 16-byte aligned", r);
        s.p[r] = (const float*)srcs[r];
    }
    if ((uintptr_t)dst & 15) return mv3d::fail(MV3D_E_INVAL, "mv3d_mesh_reduce_sum: destination not 16-byte aligned");
    const int blocks = (int)std::min<int64_t>(mv3d::cdiv64(count / 4 + 1, 256), 2048);
    return mv3d::dispatch(stream, mv3d::OpInfo{"mesh_reduce", 0.0, 4.0 * count * (nsrc + 1)}, [=](hipStream_t st) {
        mesh_reduce_kernel<<<blocks, 256, 0, st>>>(s, (float*)dst, count);
        return mv3d::launched("mesh_reduce_kernel");
    });
}

int mv3d_mesh_copy(void* dst, const void* src, int64_t count, void* stream) {
    if (!dst || !src || count <= 0) return mv3d::fail(MV3D_E_INVAL, "mv3d_mesh_copy: bad arguments");
    hipError_t e = hipMemcpyAsync(dst, src, (size_t)count * 4, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream));
    return e == hipSuccess ? MV3D_OK : mv3d::fail(MV3D_E_HIP, "mv3d_mesh_copy: %s", hipGetErrorString(e));
}

}  // extern "C"
