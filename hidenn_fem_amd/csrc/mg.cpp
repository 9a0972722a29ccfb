// In-library multi-GPU collectives (SURVEY section 8b "hfem_mg_*", section 8e): one RCCL communicator per rank
// (one process per GPU), created from a caller-broadcast unique id, whose collectives are ENQUEUED ON THE CALLER'S
// STREAM right after the energy kernel -- so a whole multi-GPU step (energy -> pack -> all_gather -> unpack ->
// optimiser) is a stream-ordered sequence of launches with no Python between them and can be captured into one
// hipGraph.  The reference has no distributed code; this is the one exchange step its path needs (the energy is a
// plain sum over elements, /root/reference/src/loss.py:85-88).
//
// RCCL is bound at run time (dlopen): the library loads and every other entry point works where RCCL is absent;
// hfem_mg_* then return an error.  Load order (ensure_loaded): the explicit path given to hfem_mg_load (hidenn_fem_amd
// passes the librccl.so bundled with torch, so that both use ONE RCCL instance), then $HFEM_RCCL_PATH, then a librccl
// already mapped into the process (RTLD_NOLOAD), then the ROCm installation.  No HIP kernel in this file.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include <hip/hip_runtime_api.h>

#include "hfem_common.h"

namespace {

struct UniqueId { char internal[128]; };          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void *Comm;                               // ncclComm_t
constexpr int kDouble = 8, kSum = 0;              // ncclFloat64, ncclSum

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string path;
} g_rccl;
std::mutex g_mu;

int load_from(const char *path, int flags) {
    void *h = dlopen(path, flags);
    if (!h) return -1;
    Rccl r;
    r.h = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.AllGather) { dlclose(h); return -1; }
    r.path = path;
    g_rccl = r;
    return 0;
}

int ensure_loaded(const char *path) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rccl.h) return 0;
    if (path && *path && load_from(path, RTLD_NOW | RTLD_LOCAL) == 0) return 0;
    const char *env = getenv("HFEM_RCCL_PATH");
    if (env && *env && load_from(env, RTLD_NOW | RTLD_LOCAL) == 0) return 0;
    for (const char *cand : {"librccl.so", "librccl.so.1"})           // already mapped (torch's): share that instance
        if (load_from(cand, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD) == 0) return 0;
    for (const char *cand : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"})
        if (load_from(cand, RTLD_NOW | RTLD_LOCAL) == 0) return 0;
    hfem::set_error("hfem_mg: could not load RCCL (librccl.so); set HFEM_RCCL_PATH or call hfem_mg_load(path)");
    return -1;
}

int rccl_status(int rc, const char *what) {
    if (rc == 0) return 0;
    hfem::set_error(std::string(what) + ": RCCL error " + std::to_string(rc) + " (" +
                    (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?") + ")");
    return 1000 + rc;
}

}  // namespace

struct hfem_mg_comm {
    Comm comm = nullptr;
    int device = -1, rank = 0, world = 1;
};

extern "C" int hfem_mg_load(const char *librccl_path) { return ensure_loaded(librccl_path); }

extern "C" int hfem_mg_unique_id(void *id_out) {
    HFEM_ARG_CHECK(id_out, "null pointer");
    if (int rc = ensure_loaded(nullptr)) return rc;
    UniqueId id;
    if (int rc = rccl_status(g_rccl.GetUniqueId(&id), "ncclGetUniqueId")) return rc;
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int hfem_mg_comm_create(int device, int rank, int world, const void *id, hfem_mg_comm **out) {
    HFEM_ARG_CHECK(out && id, "null pointer");
    HFEM_ARG_CHECK(world >= 1 && rank >= 0 && rank < world, "bad rank / world size");
    *out = nullptr;
    if (int rc = ensure_loaded(nullptr)) return rc;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { hfem::set_error(std::string("hfem_mg_comm_create: hipSetDevice: ") + hipGetErrorString(e)); return (int)e; }
    UniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    hfem_mg_comm *c = new hfem_mg_comm;
    c->device = device; c->rank = rank; c->world = world;
    if (int rc = rccl_status(g_rccl.CommInitRank(&c->comm, world, uid, rank), "ncclCommInitRank")) { delete c; return rc; }
    *out = c;
    return 0;
}

extern "C" int hfem_mg_comm_destroy(hfem_mg_comm *c) {
    if (!c) return 0;
    int rc = 0;
    if (c->comm && g_rccl.CommDestroy) rc = rccl_status(g_rccl.CommDestroy(c->comm), "ncclCommDestroy");
    delete c;
    return rc;
}

extern "C" int hfem_mg_allreduce_sum(hfem_mg_comm *c, const double *send, double *recv, int64_t count, void *stream) {
    HFEM_ARG_CHECK(c && c->comm && send && recv && count >= 0, "bad arguments");
    if (count == 0) return 0;
    (void)hipSetDevice(c->device);
    return rccl_status(g_rccl.AllReduce(send, recv, (size_t)count, kDouble, kSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int hfem_mg_allgather(hfem_mg_comm *c, const double *send, double *recv, int64_t count, void *stream) {
    HFEM_ARG_CHECK(c && c->comm && send && recv && count >= 0, "bad arguments");
    if (count == 0) return 0;
    (void)hipSetDevice(c->device);
    return rccl_status(g_rccl.AllGather(send, recv, (size_t)count, kDouble, c->comm, (hipStream_t)stream), "ncclAllGather");
}
