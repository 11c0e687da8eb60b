// Data-parallel exchange of the training step behind the C ABI: the two collectives SURVEY.md section 8(e) names -- the
// token count that normalises the loss (model/img2seq.py:69-71 takes the mean over ALL unmasked tokens of the global batch)
// and the bucketed gradient sum -- as ncclAllReduce on RCCL (xGMI inside a node), issued on a caller-provided side stream
// behind a caller-provided "gradients ready" event.  The reference has no multi-device path (one sess.run per step,
// img2seq.py:169); this is the layer a binding adds around that call.
//
// RCCL is bound at first use with dlopen, not at link time: liblxo.so loads (and the single-GPU path runs) on a box without
// librccl, and a process that already carries an RCCL (PyTorch's) shares that copy instead of loading a second one.
#include "lxo.h"
#include "api_util.h"
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

namespace {
// the slice of rccl.h this file needs (opaque handles, the two enums by value: ncclFloat32 = 7, ncclBfloat16 = 9, ncclInt32 = 2, ncclSum = 0)
struct NcclId { char internal[LXO_COMM_ID_BYTES]; };
typedef void* NcclComm;
struct Rccl {
    void* lib;
    int (*GetUniqueId)(NcclId*);
    int (*CommInitRank)(NcclComm*, int, NcclId, int);
    int (*CommDestroy)(NcclComm);
    int (*CommCount)(NcclComm, int*);
    int (*CommUserRank)(NcclComm, int*);
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
    const char* (*GetErrorString)(int);
};
Rccl g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
thread_local char g_comm_err[256] = "";

int rccl_bind() {
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { snprintf(g_comm_err, sizeof(g_comm_err), "librccl.so not found (%s)", dlerror()); return -20; }
    Rccl r; r.lib = h;
#define SYM(field, name) *(void**)(&r.field) = dlsym(h, name); if (!r.field) { snprintf(g_comm_err, sizeof(g_comm_err), "librccl: missing symbol %s", name); return -21; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(CommCount, "ncclCommCount") SYM(CommUserRank, "ncclCommUserRank") SYM(AllReduce, "ncclAllReduce") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl = r;
    return 0;
}
int nccl_fail(int rc, const char* what) {
    snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s (nccl code %d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?", rc);
    return -22;
}
struct Comm { NcclComm c; int rank, world; };
}  // namespace

extern "C" const char* lxo_comm_last_error(void) { return g_comm_err; }

extern "C" int lxo_comm_unique_id(void* id_out) {
    if (!id_out) return -1;
    RC(rccl_bind());
    NcclId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return nccl_fail(rc, "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int lxo_comm_init(const void* unique_id, int rank, int world, void** comm_out) {
    if (!unique_id || !comm_out || world < 1 || rank < 0 || rank >= world) { snprintf(g_comm_err, sizeof(g_comm_err), "lxo_comm_init: bad arguments"); return -1; }
    RC(rccl_bind());
    NcclId id; memcpy(&id, unique_id, sizeof(id));
    Comm* c = new Comm{nullptr, rank, world};
    const int rc = g_rccl.CommInitRank(&c->c, world, id, rank);      // binds the communicator to the CURRENT HIP device of this thread
    if (rc != 0) { delete c; return nccl_fail(rc, "ncclCommInitRank"); }
    *comm_out = c;
    return 0;
}

extern "C" int lxo_comm_info(void* comm, int* rank, int* world) {
    if (!comm) return -1;
    Comm* c = static_cast<Comm*>(comm);
    int n = 0, r = 0;
    int rc = g_rccl.CommCount(c->c, &n); if (rc != 0) return nccl_fail(rc, "ncclCommCount");
    rc = g_rccl.CommUserRank(c->c, &r); if (rc != 0) return nccl_fail(rc, "ncclCommUserRank");
    if (rank) *rank = r;
    if (world) *world = n;
    return 0;
}

// sum over ranks, in place, of `count` elements (LXO_F32, LXO_BF16 or LXO_I32) on `side_stream`.  ready_event (nullable): the
// event the producer stream recorded behind the kernels that finalised this bucket; the side stream waits for it first.
// A caller that orders buckets on the host instead (hipEventSynchronize, then this call: a stream that waits for a
// compute-stream event slows the compute stream's launch chains on this runtime) passes NULL.
extern "C" int lxo_allreduce_bucket(void* comm, void* ptr, long long count, int dtype, void* side_stream, void* ready_event) {
    if (!comm || !ptr || count < 0) return -1;
    if (count == 0) return 0;
    Comm* c = static_cast<Comm*>(comm);
    const int nt = dtype == LXO_F32 ? 7 : (dtype == LXO_BF16 ? 9 : (dtype == LXO_I32 ? 2 : -1));
    if (nt < 0) { snprintf(g_comm_err, sizeof(g_comm_err), "lxo_allreduce_bucket: dtype %d", dtype); return -1; }
    hipStream_t st = (hipStream_t)side_stream;
    if (ready_event) HIPRC(hipStreamWaitEvent(st, (hipEvent_t)ready_event, 0));
    const int rc = g_rccl.AllReduce(ptr, ptr, (size_t)count, nt, 0 /* ncclSum */, c->c, st);
    if (rc != 0) return nccl_fail(rc, "ncclAllReduce");
    return 0;
}

extern "C" int lxo_comm_destroy(void* comm) {
    if (!comm) return 0;
    Comm* c = static_cast<Comm*>(comm);
    const int rc = g_rccl.CommDestroy ? g_rccl.CommDestroy(c->c) : 0;
    delete c;
    if (rc != 0) return nccl_fail(rc, "ncclCommDestroy");
    return 0;
}
