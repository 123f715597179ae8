// csrc/kws_comm.hip -- data-parallel gradient exchange behind the C ABI: RCCL over xGMI, one communicator per GPU process.
//
// The reference trains in one process (train.py:81-92, model.fit(workers=1)); this component is new (SURVEY section 5,
// "Distributed communication backend").  The exchange of a train step is ONE sum-all-reduce of the flat fp32 gradient
// buffer (134 932 floats = 540 KB for simple_cnn), issued as two buckets:
//   early bucket  grads[split, n)   conv4 + BN4 + dense + head, 82 % of the bytes, final first -> inside the train step
//                                   (kws_train_args.comm) it is enqueued on the MODEL's side stream right behind conv4's weight
//                                   gradient, i.e. while conv3 .. conv1 backward still run on the caller's stream
//   late bucket   grads[0, split)   + the BatchNormalization moving statistics (weighted mean over ranks), one RCCL group on
//                                   the caller's stream behind the backward pass.
// The communicator owns NO stream: a collective on a stream of its own stalled the device by ~1.1 ms per step (0.65 -> 1.79 ms
// with a one-rank communicator; the same call on a stream that already carries the step's kernels costs nothing: tools/commbench.py),
// so every collective is enqueued on a stream the step already uses.
//
// RCCL is bound at run time (dlopen): a process that already holds a librccl.so.1 (PyTorch ships one) shares that
// instance, a torch-free host loads the system one from the ROCm library path.  libkws_hip.so itself has no link-time
// dependency on RCCL, so single-GPU users never load it.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "kws_common.h"

using namespace kws;

namespace {

// the slice of the NCCL / RCCL C API this file uses (stable since NCCL 2.10)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[KWS_COMM_ID_BYTES]; } ncclUniqueId;
enum { ncclSuccessV = 0 };
enum { ncclInt8 = 0, ncclInt32 = 2, ncclInt64 = 4, ncclFloat32 = 7, ncclFloat64 = 8 };
enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    std::string error;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)                                   // an instance the process already holds wins
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!r.handle)
            for (const char *n : names)
                if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.handle) { r.error = std::string("cannot load librccl.so.1: ") + (dlerror() ? dlerror() : "not found"); return; }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("librccl has no symbol ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    });
    return r.error.empty() ? &r : nullptr;
}

int rccl_fail(const char *what, int rc)
{
    Rccl *r = rccl();
    return fail(KWS_ERR_COMM, "%s failed: %s", what, (r && r->GetErrorString) ? r->GetErrorString(rc) : "RCCL error");
}

#define KWS_RCCL_CHECK(what, expr)                       \
    do {                                                 \
        const int rc__ = (expr);                         \
        if (rc__ != ncclSuccessV) return rccl_fail(what, rc__); \
    } while (0)

__global__ void scale_kernel(float *__restrict__ x, long n, float a)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] *= a;
}

}  // namespace

struct kws_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    bool timing = false;
    hipEvent_t t[4] = {nullptr, nullptr, nullptr, nullptr};   // early bucket begin / end, late bucket begin / end
    bool timed_early = false, timed_late = false;
};

extern "C" {

int kws_comm_unique_id(void *id)
{
    if (!id) return fail(KWS_ERR_INVALID, "null argument");
    Rccl *r = rccl();
    if (!r) return fail(KWS_ERR_COMM, "RCCL is not available (librccl.so.1 could not be loaded)");
    ncclUniqueId u;
    KWS_RCCL_CHECK("ncclGetUniqueId", r->GetUniqueId(&u));
    std::memcpy(id, u.internal, KWS_COMM_ID_BYTES);
    return KWS_OK;
}

int kws_comm_init(int rank, int world, const void *unique_id, kws_comm **out)
{
    if (!out || !unique_id) return fail(KWS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(KWS_ERR_INVALID, "rank %d outside a world of %d", rank, world);
    Rccl *r = rccl();
    if (!r) return fail(KWS_ERR_COMM, "RCCL is not available (librccl.so.1 could not be loaded)");
    int dev = 0;
    KWS_HIP_CHECK(hipGetDevice(&dev));
    auto *c = new kws_comm();
    c->rank = rank; c->world = world; c->device = dev;
    ncclUniqueId u;
    std::memcpy(u.internal, unique_id, KWS_COMM_ID_BYTES);
    int rc = r->CommInitRank(&c->comm, world, u, rank);
    if (rc != ncclSuccessV) { delete c; return rccl_fail("ncclCommInitRank", rc); }
    hipError_t e = hipSuccess;
    for (auto &t : c->t)
        if (e == hipSuccess) e = hipEventCreate(&t);
    if (e != hipSuccess) {
        kws_comm_destroy(c);
        return fail(KWS_ERR_HIP, "communicator timing events: %s", hipGetErrorString(e));
    }
    *out = c;
    return KWS_OK;
}

void kws_comm_destroy(kws_comm *c)
{
    if (!c) return;
    Rccl *r = rccl();
    // the communicator, its events and the work still in flight belong to c->device, which need not be the caller's current one
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != c->device) (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->comm && r) (void)r->CommDestroy(c->comm);
    for (auto &t : c->t)
        if (t) (void)hipEventDestroy(t);
    (void)hipGetLastError();
    if (cur >= 0 && cur != c->device) (void)hipSetDevice(cur);
    delete c;
}

int kws_comm_info(const kws_comm *c, int *rank, int *world, int *rccl_version)
{
    if (!c) return fail(KWS_ERR_INVALID, "null argument");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (rccl_version) {
        *rccl_version = 0;
        Rccl *r = rccl();
        if (r && r->GetVersion) (void)r->GetVersion(rccl_version);
    }
    return KWS_OK;
}

}  // extern "C"

namespace kws {

// early bucket: one all-reduce on `s` (the train step passes its side stream)
int comm_allreduce_early(kws_comm *c, float *buf, int64_t n, hipStream_t s)
{
    if (!c || !buf || n <= 0) return KWS_OK;
    Rccl *r = rccl();
    if (!r) return fail(KWS_ERR_COMM, "RCCL is not available");
    c->timed_early = false;
    if (c->timing) KWS_HIP_CHECK(hipEventRecord(c->t[0], s));
    KWS_RCCL_CHECK("ncclAllReduce (early bucket)", r->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, s));
    if (c->timing) { KWS_HIP_CHECK(hipEventRecord(c->t[1], s)); c->timed_early = true; }
    return KWS_OK;
}

// late bucket + state * weight, one RCCL group on `s`
int comm_allreduce_late(kws_comm *c, float *grads, int64_t n, float *state, int64_t n_state, float state_weight, hipStream_t s)
{
    if (!c) return fail(KWS_ERR_INVALID, "null communicator");
    Rccl *r = rccl();
    if (!r) return fail(KWS_ERR_COMM, "RCCL is not available");
    c->timed_late = false;
    if (n <= 0 && n_state <= 0) return KWS_OK;
    if (c->timing) KWS_HIP_CHECK(hipEventRecord(c->t[2], s));
    if (n_state > 0 && state_weight != 1.0f)
        hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n_state + 255) / 256)), dim3(256), 0, s, state, (long)n_state, state_weight);
    KWS_RCCL_CHECK("ncclGroupStart", r->GroupStart());
    const int rc1 = n > 0 ? r->AllReduce(grads, grads, (size_t)n, ncclFloat32, ncclSum, c->comm, s) : ncclSuccessV;
    const int rc2 = n_state > 0 ? r->AllReduce(state, state, (size_t)n_state, ncclFloat32, ncclSum, c->comm, s) : ncclSuccessV;
    const int rc3 = r->GroupEnd();
    if (rc1 != ncclSuccessV) return rccl_fail("ncclAllReduce (late bucket)", rc1);
    if (rc2 != ncclSuccessV) return rccl_fail("ncclAllReduce (BatchNormalization statistics)", rc2);
    if (rc3 != ncclSuccessV) return rccl_fail("ncclGroupEnd", rc3);
    if (c->timing) { KWS_HIP_CHECK(hipEventRecord(c->t[3], s)); c->timed_late = true; }
    KWS_LAUNCH_CHECK("gradient exchange");
    return KWS_OK;
}

}  // namespace kws

extern "C" {

int kws_allreduce_grads(kws_comm *c, float *grads, int64_t n, int64_t split, float *state, int64_t n_state, float state_weight, void *stream)
{
    if (!c || !grads) return fail(KWS_ERR_INVALID, "null argument");
    if (n < 0 || split < 0 || split > n || n_state < 0 || (n_state > 0 && !state)) return fail(KWS_ERR_INVALID, "bad bucket sizes");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the same two collectives, in the same order, as a train step with kws_train_args.comm issues: ranks may mix the two forms
    // (a rank whose shard of a partial batch is empty calls this on a cleared gradient buffer)
    const bool two = split > 0 && split < n;
    if (two)
        if (int rc = comm_allreduce_early(c, grads + split, n - split, s)) return rc;
    return comm_allreduce_late(c, grads, two ? split : n, state, n_state, state_weight, s);
}

int kws_comm_allreduce(kws_comm *c, void *buf, int64_t n, int dtype, int op, void *stream)
{
    if (!c || (!buf && n > 0)) return fail(KWS_ERR_INVALID, "null argument");
    if (n < 0) return fail(KWS_ERR_INVALID, "negative count");
    if (n == 0) return KWS_OK;
    Rccl *r = rccl();
    if (!r) return fail(KWS_ERR_COMM, "RCCL is not available");
    int dt, ro;
    switch (dtype) {
    case KWS_DT_F32: dt = ncclFloat32; break;
    case KWS_DT_F64: dt = ncclFloat64; break;
    case KWS_DT_I32: dt = ncclInt32; break;
    case KWS_DT_I64: dt = ncclInt64; break;
    default: return fail(KWS_ERR_INVALID, "unknown dtype %d", dtype);
    }
    switch (op) {
    case KWS_OP_SUM: ro = ncclSum; break;
    case KWS_OP_MAX: ro = ncclMax; break;
    case KWS_OP_AVG: ro = ncclAvg; break;
    default: return fail(KWS_ERR_INVALID, "unknown reduction %d", op);
    }
    KWS_RCCL_CHECK("ncclAllReduce", r->AllReduce(buf, buf, (size_t)n, dt, ro, c->comm, static_cast<hipStream_t>(stream)));
    return KWS_OK;
}

int kws_comm_broadcast(kws_comm *c, void *buf, int64_t nbytes, int root, void *stream)
{
    if (!c || (!buf && nbytes > 0)) return fail(KWS_ERR_INVALID, "null argument");
    if (nbytes < 0) return fail(KWS_ERR_INVALID, "negative count");
    if (root < 0 || root >= c->world) return fail(KWS_ERR_INVALID, "root %d outside a world of %d", root, c->world);
    if (nbytes == 0) return KWS_OK;
    Rccl *r = rccl();
    if (!r) return fail(KWS_ERR_COMM, "RCCL is not available");
    KWS_RCCL_CHECK("ncclBroadcast", r->Broadcast(buf, buf, (size_t)nbytes, ncclInt8, root, c->comm, static_cast<hipStream_t>(stream)));
    return KWS_OK;
}

int kws_comm_timing(kws_comm *c, int on)
{
    if (!c) return fail(KWS_ERR_INVALID, "null argument");
    c->timing = on != 0;
    c->timed_early = c->timed_late = false;
    return KWS_OK;
}

int kws_comm_last_us(kws_comm *c, float *early_us, float *late_us)
{
    if (!c) return fail(KWS_ERR_INVALID, "null argument");
    if (early_us) *early_us = -1.f;
    if (late_us) *late_us = -1.f;
    float ms = 0.f;
    if (c->timed_early && early_us) {
        KWS_HIP_CHECK(hipEventSynchronize(c->t[1]));
        KWS_HIP_CHECK(hipEventElapsedTime(&ms, c->t[0], c->t[1]));
        *early_us = ms * 1e3f;
    }
    if (c->timed_late && late_us) {
        KWS_HIP_CHECK(hipEventSynchronize(c->t[3]));
        KWS_HIP_CHECK(hipEventElapsedTime(&ms, c->t[2], c->t[3]));
        *late_us = ms * 1e3f;
    }
    return KWS_OK;
}

}  // extern "C"
