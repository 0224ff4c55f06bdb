// RCCL collectives behind the C ABI: for hosts that bind libvar_hip.so without
// torch.distributed.  One communicator per context (= per process and GPU), created from an RCCL unique id that the
// host distributes over its own channel (rank 0 calls var_comm_unique_id).  The two exchanges of the data-parallel
// step: sum-all-reduce of the flat gradient arena (+ loss slot) over xGMI, and the all-gather of the local
// embeddings for the in-batch-negatives extension of BASELINE config 3.  librccl is opened lazily with dlopen
// (whichever copy the process already holds -- PyTorch ships one -- is reused), so that the library has no
// load-time dependency on it and single-GPU users never touch it.
#include <dlfcn.h>

#include "var_common.h"

namespace {
struct UniqueId { char internal[128]; };               // ncclUniqueId (NCCL_UNIQUE_ID_BYTES)
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, Comm, hipStream_t);
typedef const char* (*ErrStrFn)(int);
constexpr int kFloat = 7, kSum = 0;                    // ncclFloat32, ncclSum

struct Rccl {
    void* h = nullptr;
    GetUniqueIdFn get_id = nullptr; CommInitRankFn init = nullptr; CommDestroyFn destroy = nullptr;
    AllReduceFn allreduce = nullptr; AllGatherFn allgather = nullptr; ErrStrFn errstr = nullptr;
};
Rccl g_rccl;

int load_rccl(var_ctx* c) {
    if (g_rccl.h) return VAR_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { VAR_SET_ERR(c, "var_comm: cannot open librccl.so (%s)", dlerror()); return VAR_ERR_STATE; }
    Rccl r;
    r.h = h;
    r.get_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    r.init = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    r.destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    r.allreduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    r.allgather = (AllGatherFn)dlsym(h, "ncclAllGather");
    r.errstr = (ErrStrFn)dlsym(h, "ncclGetErrorString");
    if (!r.get_id || !r.init || !r.destroy || !r.allreduce || !r.allgather) {
        VAR_SET_ERR(c, "var_comm: librccl lacks an expected symbol");
        return VAR_ERR_STATE;
    }
    g_rccl = r;
    return VAR_OK;
}

#define RCCL_CHECK(c, expr)                                                                                  \
    do {                                                                                                     \
        int e_ = (expr);                                                                                     \
        if (e_ != 0) {                                                                                       \
            VAR_SET_ERR(c, "%s failed: %s", #expr, g_rccl.errstr ? g_rccl.errstr(e_) : "rccl error");        \
            return VAR_ERR_HIP;                                                                              \
        }                                                                                                    \
    } while (0)
}  // namespace

void comm_free(var_ctx* c) {
    if (c->comm && g_rccl.destroy) (void)g_rccl.destroy((Comm)c->comm);
    c->comm = nullptr;
}

extern "C" {

int var_comm_unique_id(var_ctx* c, void* id128) {
    if (!c || !id128) return VAR_ERR_ARG;
    int r = load_rccl(c);
    if (r != VAR_OK) return r;
    RCCL_CHECK(c, g_rccl.get_id((UniqueId*)id128));
    return VAR_OK;
}

int var_comm_init(var_ctx* c, int rank, int nranks, const void* id128) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) { VAR_SET_ERR(c, "var_comm_init: bad argument"); return VAR_ERR_ARG; }
    int r = load_rccl(c);
    if (r != VAR_OK) return r;
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    comm_free(c);
    UniqueId id = *(const UniqueId*)id128;
    Comm comm = nullptr;
    RCCL_CHECK(c, g_rccl.init(&comm, nranks, id, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_size = nranks;
    return VAR_OK;
}

int var_comm_destroy(var_ctx* c) {
    if (!c) return VAR_ERR_ARG;
    comm_free(c);
    return VAR_OK;
}

int var_allreduce_grads(var_ctx* c, void* stream, float* flat_grad, long n) {
    if (!c || !flat_grad || n < 1) return VAR_ERR_ARG;
    if (!c->comm) { VAR_SET_ERR(c, "var_allreduce_grads: var_comm_init first"); return VAR_ERR_STATE; }
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    RCCL_CHECK(c, g_rccl.allreduce(flat_grad, flat_grad, (size_t)n, kFloat, kSum, (Comm)c->comm, (hipStream_t)stream));
    return VAR_OK;
}

int var_allgather_emb(var_ctx* c, void* stream, const float* local, float* global, long n_local) {
    if (!c || !local || !global || n_local < 1) return VAR_ERR_ARG;
    if (!c->comm) { VAR_SET_ERR(c, "var_allgather_emb: var_comm_init first"); return VAR_ERR_STATE; }
    VAR_HIP_CHECK(c, hipSetDevice(c->device));
    RCCL_CHECK(c, g_rccl.allgather(local, global, (size_t)n_local, kFloat, (Comm)c->comm, (hipStream_t)stream));
    return VAR_OK;
}

}  // extern "C"
