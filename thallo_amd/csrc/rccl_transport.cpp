// rccl_transport.cpp -- the all-gather / all-reduce of the multi-GPU step on RCCL, INSIDE the library ("take Python out of the PCG loop").
//
// The reference has no counterpart (single device: API/src/util.t:769-772).  solver_dist.cpp issues one all-gather per PCG iteration for the row-slab,
// vertex-range and camera-shard forms (plus one all-reduce of the point block for bundle adjustment); round 2 routed each of them through the caller's
// callback -- in this repo a ctypes callback into torch.distributed, i.e. the Python interpreter once or twice per PCG iteration.  With a communicator
// of its own the Plan enqueues ncclAllGather / ncclAllReduce on its stream itself: the loop has no host-language hop, and a captured hipGraph of
// Thallo_ProblemStep holds the collectives as plain kernel nodes.
//
// RCCL is bound with dlopen at first use, not linked: libThallo.so must load on machines without it (the build container, single-GPU hosts), and an
// application that already has an RCCL in its process (PyTorch ships one) must end up with THAT instance, not a second one: dlopen by soname returns
// the copy that is already mapped.
#include "rccl_transport.hpp"
#include <dlfcn.h>
#include <cstdio>
#include <cstring>

namespace thallo {

namespace {
// the six entry points used, with the ABI of rccl.h (ncclResult_t = int, ncclComm_t = opaque pointer, ncclUniqueId = 128 bytes by value)
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueId_t)(UniqueId*);
typedef int (*CommInitRank_t)(void** comm, int nranks, UniqueId id, int rank);
typedef int (*CommDestroy_t)(void* comm);
typedef int (*AllGather_t)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t s);
typedef int (*AllReduce_t)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s);
typedef const char* (*GetErrorString_t)(int);
typedef int (*CommQuery_t)(void* comm, int* out);      // ncclCommCount / ncclCommCuDevice / ncclCommUserRank
constexpr int kChar = 0, kFloat = 7, kSum = 0;      // ncclChar, ncclFloat32, ncclSum (rccl.h:448-466)

struct Api {
    void* handle = nullptr;
    GetUniqueId_t get_unique_id = nullptr; CommInitRank_t comm_init_rank = nullptr; CommDestroy_t comm_destroy = nullptr;
    AllGather_t all_gather = nullptr; AllReduce_t all_reduce = nullptr; GetErrorString_t error_string = nullptr;
    CommQuery_t comm_count = nullptr, comm_device = nullptr, comm_rank = nullptr;      // (optional: diagnostics)
    bool tried = false;
    const char* why = "";
};
Api& api()
{
    static Api a;
    if (a.tried) return a;
    a.tried = true;
    // A copy the process has already mapped comes first (PyTorch ships its own, soname librccl.so.1: glibc matches mapped libraries by soname, so asking for
    // "librccl.so" first could load a SECOND RCCL from the system path next to it -- ADVICE r3); only then the loader's search path.
    a.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
        if (a.handle) break;
        a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!a.handle) { a.why = "librccl.so not found"; return a; }
    a.get_unique_id = (GetUniqueId_t)dlsym(a.handle, "ncclGetUniqueId");
    a.comm_init_rank = (CommInitRank_t)dlsym(a.handle, "ncclCommInitRank");
    a.comm_destroy = (CommDestroy_t)dlsym(a.handle, "ncclCommDestroy");
    a.all_gather = (AllGather_t)dlsym(a.handle, "ncclAllGather");
    a.all_reduce = (AllReduce_t)dlsym(a.handle, "ncclAllReduce");
    a.error_string = (GetErrorString_t)dlsym(a.handle, "ncclGetErrorString");
    a.comm_count = (CommQuery_t)dlsym(a.handle, "ncclCommCount"); a.comm_device = (CommQuery_t)dlsym(a.handle, "ncclCommCuDevice"); a.comm_rank = (CommQuery_t)dlsym(a.handle, "ncclCommUserRank");
    if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_gather || !a.all_reduce) { a.why = "librccl.so lacks an entry point"; a.handle = nullptr; }
    return a;
}
int fail(const char* what, int rc)
{
    Api& a = api();
    set_error("RCCL: %s failed (%d: %s)", what, rc, a.error_string ? a.error_string(rc) : "?");
    return -1;
}
}  // namespace

bool rccl_available(const char** why) { Api& a = api(); if (why) *why = a.why; return a.handle != nullptr; }

int rccl_unique_id(unsigned char* out128)
{
    Api& a = api();
    if (!a.handle) { set_error("RCCL: %s", a.why); return -1; }
    UniqueId id; memset(&id, 0, sizeof(id));
    const int rc = a.get_unique_id(&id);
    if (rc) return fail("ncclGetUniqueId", rc);
    memcpy(out128, id.internal, 128);
    return 0;
}

RcclComm* rccl_comm_create(const unsigned char* id128, int rank, int world)
{
    Api& a = api();
    if (!a.handle) { set_error("RCCL: %s", a.why); return nullptr; }
    if (!id128 || world < 1 || rank < 0 || rank >= world) { set_error("RCCL: rank %d of %d", rank, world); return nullptr; }
    UniqueId id; memcpy(id.internal, id128, 128);
    void* comm = nullptr;
    const int rc = a.comm_init_rank(&comm, world, id, rank);        // collective: every rank of the id calls it, each with its GPU current
    if (rc || !comm) { fail("ncclCommInitRank", rc); return nullptr; }
    RcclComm* c = new RcclComm(); c->comm = comm; c->rank = rank; c->world = world;
    return c;
}

void rccl_comm_destroy(RcclComm* c)
{
    if (!c) return;
    Api& a = api();
    if (a.handle && c->comm) a.comm_destroy(c->comm);
    delete c;
}

// what the communicator itself says: ranks in it, this rank's device, this rank's number (-1 where RCCL has no answer).  First contact with a real node: a world of
// 1 here while torch.distributed says N means the ranks never met
void rccl_comm_query(RcclComm* c, int out[3])
{
    out[0] = out[1] = out[2] = -1;
    Api& a = api();
    if (!c || !a.handle || !c->comm) return;
    if (a.comm_count) { int v = -1; if (a.comm_count(c->comm, &v) == 0) out[0] = v; }
    if (a.comm_device) { int v = -1; if (a.comm_device(c->comm, &v) == 0) out[1] = v; }
    if (a.comm_rank) { int v = -1; if (a.comm_rank(c->comm, &v) == 0) out[2] = v; }
}

int rccl_allgather(RcclComm* c, const void* send, void* recv, long bytes_per_rank, hipStream_t s)
{
    Api& a = api();
    if (!c || !a.handle) { set_error("RCCL: no communicator"); return -1; }
    const int rc = a.all_gather(send, recv, (size_t)bytes_per_rank, kChar, c->comm, s);
    return rc ? fail("ncclAllGather", rc) : 0;
}

int rccl_allreduce_sum(RcclComm* c, float* buf, long count, hipStream_t s)
{
    Api& a = api();
    if (!c || !a.handle) { set_error("RCCL: no communicator"); return -1; }
    const int rc = a.all_reduce(buf, buf, (size_t)count, kFloat, kSum, c->comm, s);
    return rc ? fail("ncclAllReduce", rc) : 0;
}

}  // namespace thallo
