// rccl_hook.cpp -- the cross-rank sum of the sharded ADMM path as a NATIVE all-reduce hook (liblorads_rccl.so).
//
// The HIP library takes its cross-rank sum through one callback, lorads_hip_allreduce_fn (include/lorads_hip.h:63): once
// per ADMM iteration, between two of its kernel launches, on a device buffer of m + 2 (+ objective partials) doubles.
// A host program written in Python can implement that callback with torch.distributed, but then every ADMM iteration
// runs ~50-100 us of interpreter + ProcessGroup code in the middle of a ~170 us launch chain, and two cross-stream event
// hops around RCCL's own stream: the sharded iteration becomes host-bound as soon as the host is a little slow (far NUMA
// node, busy sibling core: 0.18 -> 0.25 ms per iteration, profiles/tools/numa_probe.sh).  This hook is plain C: one
// ncclAllReduce enqueued IN STREAM ORDER on the library's own stream (RCCL over xGMI, one process per GPU), nothing else.
//
// RCCL is not linked: the functions are taken with dlsym from the librccl the process names (bench.py passes the copy
// PyTorch has already loaded, so there is one RCCL in the process).  The communicator is created from a unique id the
// ranks exchange by whatever rendezvous they already have (bench.py: torch.distributed broadcast).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace {

struct Api {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
} api;

thread_local std::string g_err;

struct Comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr; // the HIP library's own stream: the collective is ordered with its kernels
    double *scratch = nullptr;    // for the few host-buffer sums (rank agreement on a decision: 1-2 doubles)
    size_t scratch_len = 0;
};

int fail(const char *what, ncclResult_t r) {
    g_err = std::string(what) + ": " + (api.GetErrorString ? api.GetErrorString(r) : "RCCL error");
    return 1;
}

} // namespace

extern "C" {

const char *lorads_rccl_last_error() { return g_err.c_str(); }

// dlopen `libpath` (or the default search path when null/empty) and bind the five entry points
int lorads_rccl_open(const char *libpath) {
    if (api.lib) return 0;
    void *h = dlopen((libpath && *libpath) ? libpath : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { g_err = std::string("dlopen: ") + dlerror(); return 1; }
#define BIND(field, name)                                                                          \
    api.field = (decltype(api.field))dlsym(h, name);                                               \
    if (!api.field) { g_err = std::string("dlsym ") + name + ": " + dlerror(); dlclose(h); return 1; }
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(AllReduce, "ncclAllReduce");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    api.lib = h;
    return 0;
}

int lorads_rccl_unique_id(char out[NCCL_UNIQUE_ID_BYTES]) {
    if (!api.lib) { g_err = "lorads_rccl_open first"; return 1; }
    ncclUniqueId id;
    ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
    memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

// collective over all ranks; `hip_stream` = lorads_hip_stream(ctx).  Returns the handle to pass as the hook's `user`.
void *lorads_rccl_comm_create(const char id_bytes[NCCL_UNIQUE_ID_BYTES], int rank, int world, void *hip_stream) {
    if (!api.lib) { g_err = "lorads_rccl_open first"; return nullptr; }
    ncclUniqueId id;
    memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
    Comm *c = new Comm;
    c->stream = (hipStream_t)hip_stream;
    ncclResult_t r = api.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { fail("ncclCommInitRank", r); delete c; return nullptr; }
    c->scratch_len = 64;
    if (hipMalloc((void **)&c->scratch, sizeof(double) * c->scratch_len) != hipSuccess) {
        g_err = "hipMalloc of the scratch buffer failed";
        api.CommDestroy(c->comm);
        delete c;
        return nullptr;
    }
    return c;
}

// lorads_hip_allreduce_fn.  Device buffers: one in-place ncclAllReduce on the library's stream, no synchronisation (the
// library is told so with lorads_hip_set_allreduce_stream_ordered).  Host buffers (a couple of doubles): through the
// scratch buffer, synchronised.
int lorads_rccl_allreduce_hook(void *user, double *buf, int32_t count, int32_t on_device) {
    Comm *c = (Comm *)user;
    if (!c || count < 0) { g_err = "bad hook arguments"; return 1; }
    if (count == 0) return 0;
    if (on_device) {
        ncclResult_t r = api.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, c->comm, c->stream);
        return r == ncclSuccess ? 0 : fail("ncclAllReduce", r);
    }
    if ((size_t)count > c->scratch_len) {
        double *p = nullptr;
        if (hipMalloc((void **)&p, sizeof(double) * (size_t)count) != hipSuccess) { g_err = "hipMalloc failed"; return 1; }
        hipFree(c->scratch);
        c->scratch = p;
        c->scratch_len = (size_t)count;
    }
    if (hipMemcpyAsync(c->scratch, buf, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
        g_err = "copy to the device failed";
        return 1;
    }
    ncclResult_t r = api.AllReduce(c->scratch, c->scratch, (size_t)count, ncclDouble, ncclSum, c->comm, c->stream);
    if (r != ncclSuccess) return fail("ncclAllReduce", r);
    if (hipMemcpyAsync(buf, c->scratch, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        g_err = "copy from the device failed";
        return 1;
    }
    return 0;
}

void lorads_rccl_comm_destroy(void *h) {
    Comm *c = (Comm *)h;
    if (!c) return;
    if (c->comm) api.CommDestroy(c->comm);
    hipFree(c->scratch);
    delete c;
}

} // extern "C"
