// fake_rccl.cpp -- a TEST DOUBLE of the ten RCCL entry points that biolith_amd/csrc/comm_rccl.hpp resolves with dlsym.
//
// THIS IS NOT RCCL.  It exists so that the `world > 1` branch of bl_gather_draws -- block offsets, the stream / event ordering
// between a rank's sampler launch and its contribution, the "v" form (grouped ncclBroadcast) for unequal chain counts,
// want_result = False on the non-root ranks, the always-closed group on an error -- executes on a ONE-GPU box, where the real
// library refuses two ranks on one device.  The product never loads it: only tests set BIOLITH_RCCL_LIB (comm_rccl.hpp:56) to
// this file's build, in a process of their own.  What it does NOT show: anything about RCCL itself (rings, xGMI, IPC handles,
// multi-process rendezvous) -- an N > 1 run on hardware remains the driver's.
//
// Semantics kept from the real library, as far as the caller can tell:
//   * ncclCommInitAll(comms, n, devs): n communicators of one clique, in ONE process; device ids may repeat (the point of the double);
//   * ncclAllGather / ncclBroadcast between ncclGroupStart / ncclGroupEnd: recorded, and carried out at ncclGroupEnd as
//     hipMemcpyAsync device-to-device copies on the RECEIVING rank's stream, each behind an event recorded on the SENDING rank's
//     stream at that moment (so a copy reads a rank's block only after everything that rank had queued in front of the collective);
//   * a collective posted outside a group with more than one rank in one process would deadlock the real library: refused;
//   * ncclCommInitRank: one-rank worlds only (the double is single-process);
//   * FAKE_RCCL_FAIL_CALL=n (environment, read at every call): the n-th collective call from now on fails with ncclInternalError
//     and the group it sits in is discarded at ncclGroupEnd -- the thread must stay usable.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {

struct Clique;
struct FakeComm {
    int rank = 0, world = 1, device = 0;
    Clique *clique = nullptr;
    hipEvent_t ready = nullptr; // "everything this rank queued in front of the collective is done"
};
struct Clique {
    int world = 0, alive = 0;
    std::vector<FakeComm *> members;
};
struct Op {
    int kind; // 0 all-gather, 1 broadcast
    const char *send;
    char *recv;
    size_t bytes;
    int root;
    FakeComm *comm;
    hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local bool g_poisoned = false;
thread_local std::vector<Op> g_ops;
std::mutex g_mu;
long g_calls = 0; // collective calls seen since the failure counter was last armed
long g_armed_at = -1;

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

bool inject_failure()
{
    std::lock_guard<std::mutex> lk(g_mu);
    const char *e = getenv("FAKE_RCCL_FAIL_CALL");
    if (!e || !*e) { g_armed_at = -1; return false; }
    const long n = atol(e);
    if (g_armed_at < 0) { g_armed_at = g_calls; }
    g_calls++;
    if (g_calls - g_armed_at == n) { unsetenv("FAKE_RCCL_FAIL_CALL"); g_armed_at = -1; return true; }
    return false;
}

ncclResult_t run_ops(std::vector<Op> &ops)
{
    // by clique, in posting order
    std::map<Clique *, std::vector<Op *>> by;
    for (Op &o : ops) by[o.comm->clique].push_back(&o);
    for (auto &kv : by) {
        Clique *cl = kv.first;
        std::vector<Op *> &v = kv.second;
        // per rank, its ops in posting order
        std::vector<std::vector<Op *>> per(cl->world);
        for (Op *o : v) per[o->comm->rank].push_back(o);
        const size_t n = per[0].size();
        for (int r = 0; r < cl->world; r++)
            if (per[r].size() != n || n == 0) return ncclInvalidUsage; // every rank of the clique posts the same sequence (one process drives them all)
        // the senders' streams: mark "ready" behind whatever each rank had queued
        for (int r = 0; r < cl->world; r++) {
            FakeComm *c = cl->members[r];
            if (hipSetDevice(c->device) != hipSuccess) return ncclUnhandledCudaError;
            if (hipEventRecord(c->ready, per[r][0]->stream) != hipSuccess) return ncclUnhandledCudaError;
        }
        for (size_t k = 0; k < n; k++) {
            const int kind = per[0][k]->kind, root = per[0][k]->root;
            const size_t bytes = per[0][k]->bytes;
            for (int r = 0; r < cl->world; r++)
                if (per[r][k]->kind != kind || per[r][k]->bytes != bytes || per[r][k]->root != root) return ncclInvalidArgument;
            for (int dst = 0; dst < cl->world; dst++) {
                Op *d = per[dst][k];
                if (hipSetDevice(d->comm->device) != hipSuccess) return ncclUnhandledCudaError;
                if (kind == 0) {
                    for (int src = 0; src < cl->world; src++) {
                        if (hipStreamWaitEvent(d->stream, cl->members[src]->ready, 0) != hipSuccess) return ncclUnhandledCudaError;
                        if (hipMemcpyAsync(d->recv + (size_t)src * bytes, per[src][k]->send, bytes, hipMemcpyDeviceToDevice, d->stream) != hipSuccess)
                            return ncclUnhandledCudaError;
                    }
                } else {
                    if (root < 0 || root >= cl->world) return ncclInvalidArgument;
                    if (hipStreamWaitEvent(d->stream, cl->members[root]->ready, 0) != hipSuccess) return ncclUnhandledCudaError;
                    if (hipMemcpyAsync(d->recv, per[root][k]->send, bytes, hipMemcpyDeviceToDevice, d->stream) != hipSuccess)
                        return ncclUnhandledCudaError;
                }
            }
        }
    }
    return ncclSuccess;
}

ncclResult_t post(Op op)
{
    if (inject_failure()) { g_poisoned = g_depth > 0; return ncclInternalError; }
    if (g_depth == 0) {
        if (op.comm->world > 1) return ncclInvalidUsage; // one thread driving several ranks must group its calls
        std::vector<Op> one{op};
        return run_ops(one);
    }
    g_ops.push_back(op);
    return ncclSuccess;
}

FakeComm *new_comm(Clique *cl, int rank, int device)
{
    FakeComm *c = new FakeComm();
    c->rank = rank; c->world = cl->world; c->device = device; c->clique = cl;
    (void)hipSetDevice(device);
    (void)hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    cl->members[rank] = c;
    cl->alive++;
    return c;
}

} // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version) { *version = 99999; return ncclSuccess; } // (tests: "the double is what got loaded")

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    memcpy(id, "fake-rccl", 9);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId, int rank)
{
    if (nranks != 1 || rank != 0) return ncclInvalidUsage; // single-process double
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ncclUnhandledCudaError;
    Clique *cl = new Clique();
    cl->world = 1; cl->members.assign(1, nullptr);
    *comm = reinterpret_cast<ncclComm_t>(new_comm(cl, 0, dev));
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    if (ndev <= 0) return ncclInvalidArgument;
    Clique *cl = new Clique();
    cl->world = ndev; cl->members.assign(ndev, nullptr);
    for (int i = 0; i < ndev; i++) comms[i] = reinterpret_cast<ncclComm_t>(new_comm(cl, i, devlist ? devlist[i] : i));
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (!c) return ncclSuccess;
    (void)hipSetDevice(c->device);
    if (c->ready) (void)hipEventDestroy(c->ready);
    Clique *cl = c->clique;
    delete c;
    if (--cl->alive == 0) delete cl;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    return post(Op{0, (const char *)sendbuff, (char *)recvbuff, sendcount * type_bytes(datatype), 0, reinterpret_cast<FakeComm *>(comm), stream});
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream)
{
    return post(Op{1, (const char *)sendbuff, (char *)recvbuff, count * type_bytes(datatype), root, reinterpret_cast<FakeComm *>(comm), stream});
}

ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    if (g_poisoned) { g_poisoned = false; return ncclSuccess; } // a call of this group failed: nothing of it is carried out
    return ops.empty() ? ncclSuccess : run_ops(ops);
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake-rccl: HIP call failed";
    case ncclInternalError: return "fake-rccl: injected failure";
    case ncclInvalidArgument: return "fake-rccl: invalid argument";
    case ncclInvalidUsage: return "fake-rccl: invalid usage";
    default: return "fake-rccl: error";
    }
}

} // extern "C"
