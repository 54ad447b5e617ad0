// b2x_comm.cpp — the sum-MPO communicator of the C ABI: RCCL collectives on device-resident fp64 vectors.
//
// Replaces, for the H.psi path, the MPI bodies of ParallelCommunicator<S> (src/core/parallel_rule.hpp:55, 74, 128):
//   allreduce_sum(double*, size_t)      src/core/parallel_mpi.hpp:300-309   (MPI_Allreduce, MPI_SUM, in place)
//   broadcast(double*, size_t, owner)   src/core/parallel_mpi.hpp:133-141
//   barrier()                           src/core/parallel_mpi.hpp:125-132
// One process per GPU.  The rendezvous needs no MPI: rank 0 writes the 128-byte ncclUniqueId to a file every rank can
// see (single node: any local directory), the others wait for it.  RCCL is loaded on first use (dlopen), so a serial
// run never maps it and a process that already carries an RCCL (PyTorch) shares that copy.
#include "../../include/b2x.h"
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string>
#include <thread>
#include <unistd.h>

extern int b2x_set_error(int code, const std::string &msg); // b2x_capi.cpp

namespace {

// the slice of rccl.h this file needs (declared here so that libb2x.so has no link-time dependency on RCCL)
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclFloat64 = 8 }; // ncclDataType_t: ncclDouble
enum { ncclSum = 0 };     // ncclRedOp_t

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
    bool load() {
        if (h)
            return true;
        const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
                break;
        if (!h) {
            err = std::string("cannot load RCCL: ") + dlerror();
            return false;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(h, n);
            if (!p)
                err = std::string("RCCL symbol missing: ") + n;
            return p;
        };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        Broadcast = (decltype(Broadcast))sym("ncclBroadcast");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        return GetUniqueId && CommInitRank && CommDestroy && AllReduce && Broadcast && GetErrorString;
    }
};
Rccl g_rccl;

} // namespace

struct b2x_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
    hipStream_t stream = nullptr; // the communicator's own stream: collectives run beside the caller's compute stream
    hipEvent_t ready = nullptr, done = nullptr;
    double *token = nullptr; // one element: barrier
};

#define NCHK(expr, what)                                                                                               \
    do {                                                                                                               \
        int r_ = (expr);                                                                                               \
        if (r_ != ncclSuccess)                                                                                         \
            return b2x_set_error(B2X_ERR_DEVICE, std::string(what) + ": " + g_rccl.GetErrorString(r_));              \
    } while (0)
#define HCHK(expr)                                                                                                     \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            return b2x_set_error(B2X_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)

extern "C" {

int b2x_comm_unique_id(void *id128) {
    if (!id128)
        return b2x_set_error(B2X_ERR_INVALID, "b2x_comm_unique_id: null argument");
    if (!g_rccl.load())
        return b2x_set_error(B2X_ERR_DEVICE, g_rccl.err);
    ncclUniqueId id;
    NCHK(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id128, id.internal, sizeof(id.internal));
    return B2X_OK;
}

int b2x_comm_init_id(b2x_comm **out, int rank, int size, const void *id128) {
    if (!out || !id128 || size < 1 || rank < 0 || rank >= size)
        return b2x_set_error(B2X_ERR_INVALID, "b2x_comm_init: bad rank / size / id");
    if (!g_rccl.load())
        return b2x_set_error(B2X_ERR_DEVICE, g_rccl.err);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return b2x_set_error(B2X_ERR_DEVICE, "b2x_comm_init: no HIP device");
    ncclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    b2x_comm *c = new b2x_comm();
    c->rank = rank, c->size = size;
    int r = g_rccl.CommInitRank(&c->comm, size, id, rank); // binds the communicator to the CURRENT device
    if (r != ncclSuccess) {
        delete c;
        return b2x_set_error(B2X_ERR_DEVICE, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipMalloc((void **)&c->token, sizeof(double));
    if (e == hipSuccess)
        e = hipMemset(c->token, 0, sizeof(double));
    if (e != hipSuccess) { // give back whatever was created before the failing step
        (void)g_rccl.CommDestroy(c->comm);
        if (c->token)
            (void)hipFree(c->token);
        if (c->done)
            (void)hipEventDestroy(c->done);
        if (c->ready)
            (void)hipEventDestroy(c->ready);
        if (c->stream)
            (void)hipStreamDestroy(c->stream);
        delete c;
        return b2x_set_error(B2X_ERR_DEVICE, std::string("b2x_comm_init: ") + hipGetErrorString(e));
    }
    *out = c;
    return B2X_OK;
}

// File rendezvous.  The file is self-identifying: magic, the caller's session nonce, the 128-byte id.  Rank 0 removes
// whatever is at `id_file` (a file left by a crashed earlier run), writes a private file and renames it into place; the
// other ranks poll and accept only a complete file that carries THEIR nonce — a stale file of another session is ignored
// until the timeout (B2X_COMM_TIMEOUT_S, default 120 s) and then reported as such.  Rank 0 removes the file again once
// the communicator exists (ncclCommInitRank returns when every rank has joined).  nonce 0 = "any session": safe only
// when the caller guarantees that no file of an earlier run can be at that path.
static const char kIdMagic[8] = {'B', '2', 'X', 'I', 'D', '0', '0', '1'};
struct IdFile {
    char magic[8];
    uint64_t nonce;
    char id[128];
};

int b2x_comm_init_session(b2x_comm **out, int rank, int size, const char *id_file, uint64_t nonce) {
    if (!out || size < 1 || rank < 0 || rank >= size || (size > 1 && (!id_file || !id_file[0])))
        return b2x_set_error(B2X_ERR_INVALID, "b2x_comm_init: bad rank / size / id_file");
    IdFile rec;
    memcpy(rec.magic, kIdMagic, 8), rec.nonce = nonce;
    if (size == 1 || rank == 0) {
        int rc = b2x_comm_unique_id(rec.id);
        if (rc != B2X_OK)
            return rc;
    }
    if (size > 1 && rank == 0) {
        (void)unlink(id_file); // a leftover of an earlier run must not be taken for this session's id
        const std::string tmp = std::string(id_file) + ".tmp." + std::to_string((long)getpid());
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&rec, 1, sizeof(rec), f) != sizeof(rec)) {
            if (f)
                fclose(f);
            return b2x_set_error(B2X_ERR_STATE, "b2x_comm_init: cannot write " + tmp);
        }
        fclose(f);
        if (rename(tmp.c_str(), id_file) != 0)
            return b2x_set_error(B2X_ERR_STATE, std::string("b2x_comm_init: cannot publish ") + id_file);
    } else if (size > 1) {
        const char *te = getenv("B2X_COMM_TIMEOUT_S");
        const int tries_max = (te ? std::max(1, atoi(te)) : 120) * 10;
        bool got = false, stale = false;
        for (int tries = 0; tries < tries_max && !got; tries++) {
            FILE *f = fopen(id_file, "rb");
            if (f) {
                IdFile r;
                // (rank 0 renames a complete file into place: a short or foreign file is never this session's)
                if (fread(&r, 1, sizeof(r), f) == sizeof(r) && memcmp(r.magic, kIdMagic, 8) == 0 &&
                    (nonce == 0 || r.nonce == nonce))
                    rec = r, got = true;
                else
                    stale = true;
                fclose(f);
            }
            if (!got)
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        if (!got)
            return b2x_set_error(B2X_ERR_STATE, std::string("b2x_comm_init: no id of this session from rank 0 in ") + id_file +
                                                    (stale ? " (a file of another session is there: stale id rejected)" : ""));
    }
    int rc = b2x_comm_init_id(out, rank, size, rec.id);
    if (size > 1 && rank == 0)
        (void)unlink(id_file); // every rank has joined (or the attempt failed): the id is spent either way
    return rc;
}

int b2x_comm_init(b2x_comm **out, int rank, int size, const char *id_file) {
    return b2x_comm_init_session(out, rank, size, id_file, 0);
}

int b2x_comm_rank(const b2x_comm *c, int *rank, int *size) {
    if (!c)
        return b2x_set_error(B2X_ERR_INVALID, "b2x_comm_rank: null communicator");
    if (rank)
        *rank = c->rank;
    if (size)
        *size = c->size;
    return B2X_OK;
}

// The collective runs on the communicator's own stream: it starts when everything queued on `stream` so far has finished
// and `stream` continues only after it — work the caller queues on OTHER streams meanwhile overlaps with it.
static int bracket(b2x_comm *c, hipStream_t st, int rc_coll) {
    if (rc_coll != ncclSuccess)
        return b2x_set_error(B2X_ERR_DEVICE, std::string("RCCL collective: ") + g_rccl.GetErrorString(rc_coll));
    HCHK(hipEventRecord(c->done, c->stream));
    HCHK(hipStreamWaitEvent(st, c->done, 0));
    return B2X_OK;
}

int b2x_allreduce_sum(b2x_comm *c, double *dev, size_t n, void *stream) {
    if (!c || (!dev && n))
        return b2x_set_error(B2X_ERR_INVALID, "b2x_allreduce_sum: null argument");
    if (n == 0)
        return B2X_OK;
    hipStream_t st = (hipStream_t)stream;
    HCHK(hipEventRecord(c->ready, st));
    HCHK(hipStreamWaitEvent(c->stream, c->ready, 0));
    return bracket(c, st, g_rccl.AllReduce(dev, dev, n, ncclFloat64, ncclSum, c->comm, c->stream));
}

int b2x_broadcast(b2x_comm *c, double *dev, size_t n, int root, void *stream) {
    if (!c || (!dev && n) || root < 0 || root >= c->size)
        return b2x_set_error(B2X_ERR_INVALID, "b2x_broadcast: bad argument");
    if (n == 0)
        return B2X_OK;
    hipStream_t st = (hipStream_t)stream;
    HCHK(hipEventRecord(c->ready, st));
    HCHK(hipStreamWaitEvent(c->stream, c->ready, 0));
    return bracket(c, st, g_rccl.Broadcast(dev, dev, n, ncclFloat64, root, c->comm, c->stream));
}

int b2x_barrier(b2x_comm *c) {
    if (!c)
        return b2x_set_error(B2X_ERR_INVALID, "b2x_barrier: null communicator");
    int r = g_rccl.AllReduce(c->token, c->token, 1, ncclFloat64, ncclSum, c->comm, c->stream);
    if (r != ncclSuccess)
        return b2x_set_error(B2X_ERR_DEVICE, std::string("RCCL barrier: ") + g_rccl.GetErrorString(r));
    HCHK(hipStreamSynchronize(c->stream));
    return B2X_OK;
}

int b2x_comm_destroy(b2x_comm *c) {
    if (!c)
        return B2X_OK;
    (void)hipStreamSynchronize(c->stream);
    if (c->comm)
        (void)g_rccl.CommDestroy(c->comm);
    (void)hipEventDestroy(c->ready), (void)hipEventDestroy(c->done), (void)hipStreamDestroy(c->stream);
    (void)hipFree(c->token);
    delete c;
    return B2X_OK;
}

} // extern "C"
