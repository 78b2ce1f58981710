// dl_comm.hip -- exchange of per-walker results between the GPUs of one node: a direct binding of RCCL (include/desilike_amd.h, dl_comm_*).
//
// The reference's only parallelism is over parameter points: ``vmap(..., backend='mpi')`` scatters the points, loops locally and gathers the results
// (desilike/base.py:310-335), and the samplers broadcast the log-posteriors to every rank (desilike/samplers/base.py:196-200).  Here: one process per GPU, every
// rank evaluates its contiguous share and ONE ncclAllGather (enqueued on the caller's HIP stream: no host synchronisation) hands every rank all results.
// RCCL is bound at run time (dlopen): the library has no link-time dependency on it, loads on machines without RCCL, and shares the copy the process already
// holds (PyTorch bundles its own librccl.so; two different RCCL builds in one process are avoided by preferring the one that is already loaded).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>

#include "../../include/desilike_amd.h"
#include "dl_kernels.h"

namespace {

// the subset of rccl.h this file needs (ABI of NCCL 2.x / RCCL: stable since NCCL 2.0)
typedef struct ncclComm* dlNcclComm;
typedef struct { char internal[DL_COMM_ID_BYTES]; } dlNcclUniqueId;
enum { dlNcclSuccess = 0 };
enum { dlNcclInt32 = 2, dlNcclFloat64 = 8 };

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(dlNcclUniqueId*) = nullptr;
    int (*CommInitRank)(dlNcclComm*, int, dlNcclUniqueId, int) = nullptr;
    int (*CommDestroy)(dlNcclComm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, dlNcclComm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, dlNcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    std::string path;
};

RcclApi g_rccl;
std::mutex g_rccl_mutex;

int fail(const std::string& msg) {
    dl_set_last_error(msg.c_str());
    return 1;
}

// Load RCCL once.  Order: the explicit path, the copy already mapped into the process (RTLD_NOLOAD), the default search path, /opt/rocm/lib.
int load_rccl(const char* path) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return 0;
    void* h = nullptr;
    std::string used;
    if (path && *path) {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        used = path;
        if (!h) return fail(std::string("dl_comm: cannot load ") + path + ": " + dlerror());
    }
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (int pass = 0; pass < 2 && !h; ++pass)
        for (const char* name : names) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) { used = name; break; }
        }
    if (!h) { h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL); used = "/opt/rocm/lib/librccl.so.1"; }
    if (!h) return fail("dl_comm: RCCL (librccl.so.1) not found: pass its path to dl_comm_unique_id / dl_comm_create");
    RcclApi api;
    api.handle = h;
    api.path = used;
#define DL_SYM(field, name)                                                     \
    *(void**)(&api.field) = dlsym(h, name);                                     \
    if (!api.field) return fail(std::string("dl_comm: symbol ") + name + " missing in " + used);
    DL_SYM(GetUniqueId, "ncclGetUniqueId")
    DL_SYM(CommInitRank, "ncclCommInitRank")
    DL_SYM(CommDestroy, "ncclCommDestroy")
    DL_SYM(AllGather, "ncclAllGather")
    DL_SYM(Broadcast, "ncclBroadcast")
    DL_SYM(GetErrorString, "ncclGetErrorString")
    DL_SYM(GetVersion, "ncclGetVersion")
#undef DL_SYM
    g_rccl = api;
    return 0;
}

int check(int rc, const char* what) {
    if (rc == dlNcclSuccess) return 0;
    return fail(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}

}  // namespace

struct dl_comm {
    dlNcclComm comm = nullptr;
    int device = 0, rank = 0, world = 1;
};

extern "C" {

int dl_comm_unique_id(char* id, const char* rccl_library_path) {
    if (!id) return fail("dl_comm_unique_id: null argument");
    if (load_rccl(rccl_library_path)) return 1;
    dlNcclUniqueId uid;
    std::memset(&uid, 0, sizeof(uid));
    if (check(g_rccl.GetUniqueId(&uid), "ncclGetUniqueId")) return 1;
    std::memcpy(id, uid.internal, DL_COMM_ID_BYTES);
    return 0;
}

int dl_comm_create(dl_comm** out, int device, int rank, int world, const char* id, const char* rccl_library_path) {
    if (!out || !id) return fail("dl_comm_create: null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail("dl_comm_create: rank / world out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("dl_comm_create: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail("dl_comm_create: device ordinal out of range");
    if (load_rccl(rccl_library_path)) return 1;
    if (hipSetDevice(device) != hipSuccess) return fail("dl_comm_create: hipSetDevice failed");
    dlNcclUniqueId uid;
    std::memcpy(uid.internal, id, DL_COMM_ID_BYTES);
    dl_comm* comm = new dl_comm();
    comm->device = device; comm->rank = rank; comm->world = world;
    if (check(g_rccl.CommInitRank(&comm->comm, world, uid, rank), "ncclCommInitRank")) { delete comm; return 1; }
    *out = comm;
    return 0;
}

void dl_comm_destroy(dl_comm* comm) {
    if (!comm) return;
    if (comm->comm && g_rccl.CommDestroy) { (void)hipSetDevice(comm->device); (void)g_rccl.CommDestroy(comm->comm); }
    delete comm;
}

int64_t dl_comm_info(const dl_comm* comm, const char* key) {
    if (!key) return -1;
    std::string k(key);
    if (k == "rccl_version") { int v = -1; if (g_rccl.GetVersion) (void)g_rccl.GetVersion(&v); return v; }
    if (!comm) return -1;
    if (k == "rank") return comm->rank;
    if (k == "world") return comm->world;
    if (k == "device") return comm->device;
    return -1;
}

int dl_comm_allgather_f64(dl_comm* comm, const double* send_dev, double* recv_dev, int64_t count, void* hip_stream) {
    if (!comm || !comm->comm) return fail("dl_comm_allgather_f64: null communicator");
    if (count < 0 || (count > 0 && (!send_dev || !recv_dev))) return fail("dl_comm_allgather_f64: invalid argument");
    if (count == 0) return 0;
    if (hipSetDevice(comm->device) != hipSuccess) return fail("dl_comm_allgather_f64: hipSetDevice failed");
    return check(g_rccl.AllGather(send_dev, recv_dev, (size_t)count, dlNcclFloat64, comm->comm, (hipStream_t)hip_stream), "ncclAllGather");
}

int dl_comm_broadcast_f64(dl_comm* comm, double* buf_dev, int64_t count, int root, void* hip_stream) {
    if (!comm || !comm->comm) return fail("dl_comm_broadcast_f64: null communicator");
    if (count < 0 || (count > 0 && !buf_dev) || root < 0 || root >= comm->world) return fail("dl_comm_broadcast_f64: invalid argument");
    if (count == 0) return 0;
    if (hipSetDevice(comm->device) != hipSuccess) return fail("dl_comm_broadcast_f64: hipSetDevice failed");
    return check(g_rccl.Broadcast(buf_dev, buf_dev, (size_t)count, dlNcclFloat64, root, comm->comm, (hipStream_t)hip_stream), "ncclBroadcast");
}

}  // extern "C"
