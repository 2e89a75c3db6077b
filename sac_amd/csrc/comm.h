// Per-stage merge over the GPUs of a node, behind the C ABI (include/sac_hip.h "multi-GPU merge").
//
// The reference has no counterpart: its fan-out is one Julia process per parameter point / replica with files as the
// hand-off (/root/reference/sac/cross_validation/run_cv.sh:5-9, sac/qsub_files/edge/submit.sh:25-51).  Here one process
// per GPU holds a block of the chains of ONE problem; after every fixed-Theta stage the ranks' partial sums are merged
// by exactly ONE ncclAllReduce(sum) over a fused device buffer [histograms | per-delta averages | scalar sums | chain
// count] and ONE ncclAllReduce(min) (chi2_min), both on the handle's stream and in place on device memory -- no host
// staging of the payload; the host reads back only the few merged scalars the schedule predicates need
// (free:1087, 1128, 1249-1252; peak:1343, 1545; edge:1680-1688, 1718).
//
// RCCL is resolved at run time (dlopen) the first time a communicator is asked for: single-GPU users never load it and
// libsac_hip.so carries no link-time dependency on it.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace saccomm {

// the slice of the NCCL/RCCL API used (values as in rccl.h 2.x: ncclFloat64 = 8, ncclSum = 0, ncclMin = 3)
struct UniqueId { char internal[128]; };
typedef void *comm_t;
constexpr int kFloat64 = 8, kSum = 0, kMin = 3;

struct Api {
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(comm_t *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    void *lib = nullptr;
    char why[256] = "";
    char path[1024] = "";     // what was loaded
};

inline Api &api()
{
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        // The RCCL that belongs to the HIP runtime THIS library runs on: the handle's stream and the merge buffers are
        // objects of that runtime, and a process may hold a second one (PyTorch ships its own libamdhip64 + librccl, whose
        // soname is also librccl.so.1 -- asking for the bare soname would return that copy and hand it a foreign stream).
        // So: SAC_RCCL_LIB, then librccl.so.1 next to the libamdhip64 this library is linked against, then the usual names.
        // RTLD_LOCAL | RTLD_DEEPBIND: the copy binds to its own dependencies, whatever else is loaded.
        const char *env = std::getenv("SAC_RCCL_LIB");
        char sibling[1024] = "";
        Dl_info info;
        if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
            const char *slash = std::strrchr(info.dli_fname, '/');
            if (slash && (size_t)(slash - info.dli_fname) + 16 < sizeof(sibling)) {
                std::memcpy(sibling, info.dli_fname, (size_t)(slash - info.dli_fname) + 1);
                std::strcat(sibling, "librccl.so.1");
            }
        }
        const char *names[] = {env, sibling, "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
            if (a.lib) { std::snprintf(a.path, sizeof(a.path), "%s", n); break; }
            std::snprintf(a.why, sizeof(a.why), "%s", dlerror());
        }
        if (!a.lib) return;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.lib, "ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
        if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString) {
            std::snprintf(a.why, sizeof(a.why), "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
            a.lib = nullptr;
        }
    });
    return a;
}

// One handle's communicator + its fused merge buffers (device) and the merged scalar tail (host).
struct Comm {
    comm_t comm = nullptr;
    int rank = 0, world = 1;
    double *dSum = nullptr;         // [n_sum]: payload then scalar slots; all-reduced (sum) in place
    double *dMin = nullptr;         // [n_min]: all-reduced (min) in place
    size_t n_sum = 0, n_min = 0;
    double *hTail = nullptr;        // pinned host copy of the scalar slots + the min slots
    size_t n_tail = 0;
    bool merged = false;            // the buffers hold the merge of the handle's last stage
    long long merges = 0, allreduce_calls = 0;   // bookkeeping the tests read (exactly 2 collectives per stage)
};

inline void destroy(Comm *c)
{
    if (!c) return;
    if (c->comm && api().CommDestroy) api().CommDestroy(c->comm);
    if (c->dSum) (void)hipFree(c->dSum);
    if (c->dMin) (void)hipFree(c->dMin);
    if (c->hTail) (void)hipHostFree(c->hTail);
    delete c;
}

// Creates the communicator (collective over all ranks of the job) and the buffers.  Returns nullptr and a message on failure.
inline Comm *create(const void *id128, int rank, int world, size_t n_payload, size_t n_scalars, size_t n_min, char *msg, size_t msg_len)
{
    Api &a = api();
    if (!a.lib) { std::snprintf(msg, msg_len, "RCCL is not available: %s", a.why); return nullptr; }
    if (!id128 || world < 1 || rank < 0 || rank >= world) { std::snprintf(msg, msg_len, "bad rank %d of %d", rank, world); return nullptr; }
    Comm *c = new Comm();
    c->rank = rank;
    c->world = world;
    c->n_sum = n_payload + n_scalars;
    c->n_min = n_min;
    c->n_tail = n_scalars + n_min;
    UniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    const int rc = a.CommInitRank(&c->comm, world, id, rank);
    if (rc != 0) {
        std::snprintf(msg, msg_len, "ncclCommInitRank(rank %d of %d): %s", rank, world, a.GetErrorString(rc));
        c->comm = nullptr;
        destroy(c);
        return nullptr;
    }
    if (hipMalloc((void **)&c->dSum, c->n_sum * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&c->dMin, (c->n_min ? c->n_min : 1) * sizeof(double)) != hipSuccess ||
        hipHostMalloc((void **)&c->hTail, (c->n_tail ? c->n_tail : 1) * sizeof(double)) != hipSuccess) {
        std::snprintf(msg, msg_len, "allocating the merge buffers (%zu doubles) failed", c->n_sum);
        destroy(c);
        return nullptr;
    }
    return c;
}

// The two collectives of a stage + the read-back of the merged scalar slots.  The caller has filled dSum / dMin on `stream`.
// with_payload = false (a stage that sampled no histogram): only the scalar slots [n_payload, n_sum) are reduced.
inline int merge(Comm *c, size_t n_payload, bool with_payload, hipStream_t stream, char *msg, size_t msg_len)
{
    Api &a = api();
    double *first = with_payload ? c->dSum : c->dSum + n_payload;
    const size_t count = with_payload ? c->n_sum : c->n_sum - n_payload;
    int rc = a.AllReduce(first, first, count, kFloat64, kSum, c->comm, stream);
    if (rc == 0 && c->n_min) rc = a.AllReduce(c->dMin, c->dMin, c->n_min, kFloat64, kMin, c->comm, stream);
    if (rc != 0) { std::snprintf(msg, msg_len, "ncclAllReduce: %s", a.GetErrorString(rc)); return -1; }
    c->allreduce_calls += c->n_min ? 2 : 1;
    const size_t n_scal = c->n_sum - n_payload;
    if (hipMemcpyAsync(c->hTail, c->dSum + n_payload, n_scal * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        (c->n_min && hipMemcpyAsync(c->hTail + n_scal, c->dMin, c->n_min * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess) ||
        hipStreamSynchronize(stream) != hipSuccess) {
        std::snprintf(msg, msg_len, "reading back the merged scalars failed: %s", hipGetErrorString(hipGetLastError()));
        return -1;
    }
    c->merged = true;
    c->merges += 1;
    return 0;
}

}  // namespace saccomm
