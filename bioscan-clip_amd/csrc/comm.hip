// RCCL entry points of the C ABI: the two exchange steps of the global-batch training step, as a caller
// without torch.distributed would drive them.
//
//   all-gather of a modality's [B, D] embeddings (and of the int64 labels)   -- semantic model: gather_features,
//       reference bioscanclip/model/loss_func.py:58-91 (torch.distributed.all_gather at :84-89 / :122)
//   all-reduce (SUM) of a flat f32 trainable-gradient buffer                  -- the gradient synchronisation the reference's
//       scripts/train_cl.py never does and north_star defines
//
// Stream discipline is explicit, which is the point of having them here: every call takes the communication stream, an optional
// event to wait for (recorded by the caller on the stream that PRODUCED the payload -- a tower's stream right after its
// l2-normalise, or the stream an encoder's backward ran on) and an optional event the call records behind the collective (the
// consumer -- the loss, the optimizer -- waits for that one).  Nothing synchronises the host, so a modality's all-gather runs on
// RCCL's stream beside the other towers' GEMMs.  xGMI payloads here are tiny (0.8 MB per modality, 5.9 MB of gradients): one
// collective per tensor, latency-bound.
//
// RCCL is bound at first use with dlopen, not at link time: libbsclip_hip.so loads on a box without RCCL, and a process that
// already carries torch's bundled librccl keeps a single copy of its symbols.
#include <dlfcn.h>

#include <mutex>
#include <string.h>
#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

// Resolution order: (1) an RCCL the process already carries (torch bundles one, SONAME librccl.so.1, found through torch's
// RPATH): its symbols through RTLD_DEFAULT, or the loaded object itself through RTLD_NOLOAD -- never a second copy; (2) only
// then a path search.  No RTLD_GLOBAL: the handle is private to this table.  Initialisation runs once (std::call_once):
// collectives are issued from the host thread and from autograd's device thread.
const char* load_rccl_once() {
    void* h = nullptr;
    const bool resident = dlsym(RTLD_DEFAULT, "ncclAllGather") != nullptr;
    if (!resident) {
        const char* loaded[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : loaded) {
            h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (h) break;
        }
        if (!h) {
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
            for (const char* n : names) {
                h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (h) break;
            }
        }
        if (!h) return "librccl.so not found (dlopen)";
    }
    void* from = resident ? RTLD_DEFAULT : h;
#define BIND(field, sym)                                                       \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(from, sym)); \
    if (!g_rccl.field) return "RCCL symbol missing: " sym;
    BIND(GetUniqueId, "ncclGetUniqueId")
    BIND(CommInitRank, "ncclCommInitRank")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(AllGather, "ncclAllGather")
    BIND(AllReduce, "ncclAllReduce")
    BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    g_rccl.handle = resident ? reinterpret_cast<void*>(1) : h;
    return nullptr;
}

std::once_flag g_rccl_once;
const char* g_rccl_err = nullptr;
const char* load_rccl() {
    std::call_once(g_rccl_once, [] { g_rccl_err = load_rccl_once(); });
    return g_rccl_err;
}

#define RCCL_READY()                                              \
    do {                                                          \
        const char* err__ = load_rccl();                          \
        BSCLIP_REQUIRE(err__ == nullptr, "RCCL unavailable: %s", err__); \
    } while (0)
#define RCCL_CALL(expr)                                                                          \
    do {                                                                                         \
        ncclResult_t r__ = (expr);                                                               \
        if (r__ != ncclSuccess) {                                                                \
            bsclip_set_error("%s:%d RCCL: %s", __FILE__, __LINE__, g_rccl.GetErrorString(r__)); \
            return BSCLIP_ERR_LAUNCH;                                                            \
        }                                                                                        \
    } while (0)

int wait_then(hipStream_t s, void* wait_event) {
    if (wait_event && hipStreamWaitEvent(s, static_cast<hipEvent_t>(wait_event), 0) != hipSuccess) {
        bsclip_set_error("hipStreamWaitEvent failed");
        return BSCLIP_ERR_LAUNCH;
    }
    return BSCLIP_OK;
}
int record_after(hipStream_t s, void* done_event) {
    if (done_event && hipEventRecord(static_cast<hipEvent_t>(done_event), s) != hipSuccess) {
        bsclip_set_error("hipEventRecord failed");
        return BSCLIP_ERR_LAUNCH;
    }
    return BSCLIP_OK;
}

}  // namespace

extern "C" int bsclip_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

extern "C" int bsclip_comm_unique_id(void* id_out) {
    BSCLIP_REQUIRE(id_out, "bsclip_comm_unique_id: null pointer");
    RCCL_READY();
    RCCL_CALL(g_rccl.GetUniqueId(static_cast<ncclUniqueId*>(id_out)));
    return BSCLIP_OK;
}

extern "C" int bsclip_comm_init(void** comm_out, const void* unique_id, int rank, int world) {
    BSCLIP_REQUIRE(comm_out && unique_id && world >= 1 && rank >= 0 && rank < world, "bsclip_comm_init: rank %d of %d", rank, world);
    RCCL_READY();
    ncclUniqueId id;
    __builtin_memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    RCCL_CALL(g_rccl.CommInitRank(&c, world, id, rank));
    *comm_out = c;
    return BSCLIP_OK;
}

extern "C" int bsclip_comm_destroy(void* comm) {
    BSCLIP_REQUIRE(comm, "bsclip_comm_destroy: null communicator");
    RCCL_READY();
    RCCL_CALL(g_rccl.CommDestroy(static_cast<ncclComm_t>(comm)));
    return BSCLIP_OK;
}

extern "C" int bsclip_allgather_embeddings(void* comm, const float* local, float* gathered, int64_t count, void* comm_stream,
                                           void* wait_event, void* done_event) {
    BSCLIP_REQUIRE(comm && local && gathered && count > 0, "bsclip_allgather_embeddings: null/empty input");
    RCCL_READY();
    hipStream_t s = static_cast<hipStream_t>(comm_stream);
    if (int rc = wait_then(s, wait_event)) return rc;
    RCCL_CALL(g_rccl.AllGather(local, gathered, (size_t)count, ncclFloat32, static_cast<ncclComm_t>(comm), s));
    return record_after(s, done_event);
}

extern "C" int bsclip_allgather_labels(void* comm, const int64_t* local, int64_t* gathered, int64_t count, void* comm_stream,
                                       void* wait_event, void* done_event) {
    BSCLIP_REQUIRE(comm && local && gathered && count > 0, "bsclip_allgather_labels: null/empty input");
    RCCL_READY();
    hipStream_t s = static_cast<hipStream_t>(comm_stream);
    if (int rc = wait_then(s, wait_event)) return rc;
    RCCL_CALL(g_rccl.AllGather(local, gathered, (size_t)count, ncclInt64, static_cast<ncclComm_t>(comm), s));
    return record_after(s, done_event);
}

extern "C" int bsclip_allreduce_grads(void* comm, float* grads, int64_t count, void* comm_stream, void* wait_event,
                                      void* done_event) {
    BSCLIP_REQUIRE(comm && grads && count > 0, "bsclip_allreduce_grads: null/empty input");
    RCCL_READY();
    hipStream_t s = static_cast<hipStream_t>(comm_stream);
    if (int rc = wait_then(s, wait_event)) return rc;
    RCCL_CALL(g_rccl.AllReduce(grads, grads, (size_t)count, ncclFloat32, ncclSum, static_cast<ncclComm_t>(comm), s));
    return record_after(s, done_event);
}
