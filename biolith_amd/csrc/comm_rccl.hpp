// comm_rccl.hpp -- the path's only collective: the gather of posterior draws over RCCL / xGMI.
// Included at the end of biolith_hip.hip (it needs the handle's layout).
//
// Reference counterpart: the implicit device->host gather of `mcmc.get_samples()` after
// `chain_method="parallel"` sampling (biolith/utils/fit.py:109-113, 132).  Chains are the only unit the path
// shards over, so there is exactly ONE exchange, after sampling: every rank contributes the result block of its
// own chains (draws + per-draw extras + adaptation results, contiguous in the run slab) and receives everybody's.
//
// librccl is resolved at first use with dlopen (573 MB: not worth mapping for single-GPU fits), from the directory of the
// libamdhip64 this library is bound to (see rccl_api()).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <mutex>

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

static void rccl_load(RcclApi &api);

RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once; // (two threads may create communicators at the same time)
    std::call_once(once, [] { rccl_load(api); });
    return &api;
}

static void rccl_load(RcclApi &api)
{
    // The RCCL that belongs to the HIP runtime THIS library is bound to: a process may hold two ROCm stacks (the system's
    // under /opt/rocm and the one a torch wheel bundles; which libamdhip64 this library got depends on load order), and an
    // RCCL bound to the other runtime finds "no ROCm-capable device".  So: look next to our own libamdhip64 first.
    std::string beside, beside1;
    Dl_info info;
    if (dladdr((void *)static_cast<hipError_t (*)(void **, size_t)>(&hipMalloc), &info) && info.dli_fname) {
        const std::string hip = info.dli_fname;
        const size_t slash = hip.rfind('/');
        if (slash != std::string::npos) { beside1 = hip.substr(0, slash + 1) + "librccl.so.1"; beside = hip.substr(0, slash + 1) + "librccl.so"; }
    }
    const char *names[] = {getenv("BIOLITH_RCCL_LIB"), beside1.c_str(), beside.c_str(), "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
        api.error = dlerror();
    }
    if (!api.handle) return;
    bool ok = true;
    auto sym = [&](const char *name) { void *p = dlsym(api.handle, name); if (!p) { ok = false; api.error = std::string("missing symbol ") + name; } return p; };
    api.GetVersion = (decltype(api.GetVersion))sym("ncclGetVersion");
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(api.handle); api.handle = nullptr; }
}

} // namespace

struct bl_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0, device = 0;
    hipStream_t stream = nullptr;
    void *d_recv = nullptr;
    size_t recv_bytes = 0;
    double init_ms = 0.0;
};

#define BL_NCCL(api, call)                                                                                       \
    do {                                                                                                         \
        ncclResult_t r__ = (call);                                                                               \
        if (r__ != ncclSuccess)                                                                                  \
            return bl_fail(BL_ERR_COMM, "%s failed: %s (%s:%d)", #call, (api)->GetErrorString(r__), __FILE__, __LINE__); \
    } while (0)

static int need_rccl(RcclApi **out)
{
    // RCCL checks hipGetLastError() after its own launches: a stale, harmless last-error of this thread (hipErrorNotReady
    // from an event query, say) would make it report "unhandled cuda error"
    (void)hipGetLastError();
    RcclApi *api = rccl_api();
    if (!api->handle) return bl_fail(BL_ERR_COMM, "librccl could not be loaded: %s", api->error.c_str());
    *out = api;
    return BL_OK;
}

extern "C" int bl_comm_rccl_version(int *version)
{
    if (!version) return bl_fail(BL_ERR_INVALID, "NULL argument");
    RcclApi *api;
    if (int rc = need_rccl(&api)) return rc;
    BL_NCCL(api, api->GetVersion(version));
    return BL_OK;
}

extern "C" int bl_comm_unique_id(uint8_t *id)
{
    if (!id) return bl_fail(BL_ERR_INVALID, "NULL argument");
    RcclApi *api;
    if (int rc = need_rccl(&api)) return rc;
    ncclUniqueId u;
    BL_NCCL(api, api->GetUniqueId(&u));
    static_assert(sizeof u == BL_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof u);
    return BL_OK;
}

static int comm_finish(bl_comm *c)
{
    BL_HIP(hipSetDevice(c->device));
    BL_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    return BL_OK;
}

extern "C" int bl_comm_init_rank(const uint8_t *id, int world, int rank, int device, bl_comm **out)
{
    if (!id || !out || world <= 0 || rank < 0 || rank >= world) return bl_fail(BL_ERR_INVALID, "bl_comm_init_rank: bad argument");
    *out = nullptr;
    RcclApi *api;
    if (int rc = need_rccl(&api)) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bl_fail(BL_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return bl_fail(BL_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    BL_HIP(hipSetDevice(device));
    const auto t0 = std::chrono::steady_clock::now();
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    BL_NCCL(api, api->CommInitRank(&comm, world, u, rank));
    bl_comm *c = new bl_comm();
    c->comm = comm; c->world = world; c->rank = rank; c->device = device;
    if (int rc = comm_finish(c)) { api->CommDestroy(comm); delete c; return rc; }
    c->init_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = c;
    return BL_OK;
}

extern "C" int bl_comm_init_all(int ndev, const int *devices, bl_comm **out)
{
    if (ndev <= 0 || ndev > 64 || !devices || !out) return bl_fail(BL_ERR_INVALID, "bl_comm_init_all: bad argument");
    for (int i = 0; i < ndev; i++) out[i] = nullptr;
    // (RCCL refuses two ranks on one device; said here in plain words.  BIOLITH_TEST_ALLOW_DUP_DEVICES=1 -- tests only: tests/fake_rccl,
    // a double of the collective that copies between the ranks' buffers and so can take a device twice -- lifts the refusal.  It is
    // NOT keyed on BIOLITH_RCCL_LIB: that variable is also how a real librccl in another place is named, and a real one keeps the refusal.)
    const char *allow_dup = getenv("BIOLITH_TEST_ALLOW_DUP_DEVICES");
    for (int i = 0; i < ndev && !(allow_dup && allow_dup[0] == '1'); i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) return bl_fail(BL_ERR_INVALID, "bl_comm_init_all: device %d named twice (one rank per GPU)", devices[i]);
    RcclApi *api;
    if (int rc = need_rccl(&api)) return rc;
    int nvis = 0;
    if (hipGetDeviceCount(&nvis) != hipSuccess || nvis <= 0) return bl_fail(BL_ERR_NO_DEVICE, "no HIP device visible");
    for (int i = 0; i < ndev; i++)
        if (devices[i] < 0 || devices[i] >= nvis) return bl_fail(BL_ERR_INVALID, "device %d out of range (%d visible)", devices[i], nvis);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<ncclComm_t> comms(ndev, nullptr);
    BL_NCCL(api, api->CommInitAll(comms.data(), ndev, devices));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < ndev; i++) {
        bl_comm *c = new bl_comm();
        c->comm = comms[i]; c->world = ndev; c->rank = i; c->device = devices[i]; c->init_ms = ms;
        out[i] = c;
        if (int rc = comm_finish(c)) {
            for (int j = 0; j <= i; j++) { bl_comm_destroy(out[j]); out[j] = nullptr; }
            for (int j = i + 1; j < ndev; j++) api->CommDestroy(comms[j]);
            return rc;
        }
    }
    return BL_OK;
}

extern "C" int bl_comm_info(const bl_comm *c, int *world, int *rank, int *device, double *init_ms)
{
    if (!c) return bl_fail(BL_ERR_INVALID, "NULL argument");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    if (device) *device = c->device;
    if (init_ms) *init_ms = c->init_ms;
    return BL_OK;
}

extern "C" int bl_comm_destroy(bl_comm *c)
{
    if (!c) return BL_OK;
    hipSetDevice(c->device);
    if (c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
    if (c->d_recv) hipFree(c->d_recv);
    RcclApi *api = rccl_api();
    if (c->comm && api->handle) api->CommDestroy(c->comm);
    delete c;
    return BL_OK;
}

// ---- host-only half of the gather: where every rank's block lies in the gathered buffer, and how a block unpacks ----
// (exported so that the N > 1 arithmetic is testable on a machine without GPUs: tests/test_gather_layout.py)
extern "C" int bl_result_block_layout(int chains, int num_samples, int D, uint64_t *offsets /*[9]*/)
{
    if (chains <= 0 || num_samples < 0 || D <= 0 || !offsets) return bl_fail(BL_ERR_INVALID, "bl_result_block_layout: bad argument");
    const RunLayout L = result_layout((size_t)chains, num_samples > 0 ? (size_t)num_samples : 1, (size_t)D);
    const size_t o[9] = {L.draws, L.div, L.steps, L.acc, L.pot, L.eps, L.minv, L.nleap, L.end};
    for (int i = 0; i < 9; i++) offsets[i] = (uint64_t)o[i];
    return BL_OK;
}

// byte offset of every rank's block in the gathered buffer (boff[world] = its size); *equal = all ranks ran the same chain count
static int gather_block_offsets(int world, const int32_t *chains_per_rank, size_t S, size_t D, std::vector<size_t> &boff, bool *equal)
{
    const size_t Sa = S > 0 ? S : 1;
    boff.assign((size_t)world + 1, 0);
    *equal = true;
    for (int r = 0; r < world; r++) {
        if (chains_per_rank[r] <= 0) return bl_fail(BL_ERR_INVALID, "bl_gather_draws: rank %d has no chains", r);
        boff[r + 1] = boff[r] + result_layout((size_t)chains_per_rank[r], Sa, D).end;
        *equal = *equal && chains_per_rank[r] == chains_per_rank[0];
    }
    return BL_OK;
}

// Scatter one rank's result block (the run slab's head, laid out by result_layout) into the caller's arrays at chain `c0`.
static void unpack_block(const char *blk, size_t Cr, size_t S, size_t D, size_t c0, bl_nuts_output *out)
{
    const RunLayout L = result_layout(Cr, S > 0 ? S : 1, D);
    if (out->draws && S) memcpy(out->draws + c0 * S * D, blk + L.draws, Cr * S * D * 4);
    if (out->diverging && S) memcpy(out->diverging + c0 * S, blk + L.div, Cr * S);
    if (out->num_steps && S) memcpy(out->num_steps + c0 * S, blk + L.steps, Cr * S * 4);
    if (out->accept_prob && S) memcpy(out->accept_prob + c0 * S, blk + L.acc, Cr * S * 4);
    if (out->potential_energy && S) memcpy(out->potential_energy + c0 * S, blk + L.pot, Cr * S * 4);
    if (out->step_size) memcpy(out->step_size + c0, blk + L.eps, Cr * 4);
    if (out->inv_mass) memcpy(out->inv_mass + c0 * D, blk + L.minv, Cr * D * 4);
    if (out->n_leapfrog) memcpy(out->n_leapfrog + c0 * 2, blk + L.nleap, Cr * 16);
}

static void unpack_gathered(const char *host, int world, const int32_t *chains_per_rank, const std::vector<size_t> &boff, size_t S, size_t D,
                            bl_nuts_output *out)
{
    size_t c0 = 0;
    for (int r = 0; r < world; r++) {
        unpack_block(host + boff[r], (size_t)chains_per_rank[r], S, D, c0, out);
        c0 += (size_t)chains_per_rank[r];
    }
}

extern "C" int bl_gather_draws(bl_comm *const *comms, bl_dataset *const *dss, int n_local, const int32_t *chains_per_rank,
                               bl_nuts_output *out)
{
    if (!comms || !dss || n_local <= 0 || !chains_per_rank) return bl_fail(BL_ERR_INVALID, "bl_gather_draws: bad argument");
    RcclApi *api;
    if (int rc = need_rccl(&api)) return rc;
    const int world = comms[0] ? comms[0]->world : 0;
    if (world <= 0) return bl_fail(BL_ERR_INVALID, "bl_gather_draws: NULL communicator");
    const size_t S = dss[0] ? (size_t)dss[0]->S : 0, D = dss[0] ? (size_t)dss[0]->D : 0;
    for (int i = 0; i < n_local; i++) {
        const bl_comm *c = comms[i];
        const bl_dataset *ds = dss[i];
        if (!c || !ds) return bl_fail(BL_ERR_INVALID, "bl_gather_draws: NULL handle");
        if (c->world != world || c->rank < 0 || c->rank >= world) return bl_fail(BL_ERR_INVALID, "bl_gather_draws: communicators of different worlds");
        if (!ds->have_run || ds->in_flight) return bl_fail(BL_ERR_BUSY, "bl_gather_draws: rank %d has no finished NUTS launch (bl_nuts_wait first)", c->rank);
        if (ds->device != c->device) return bl_fail(BL_ERR_INVALID, "bl_gather_draws: rank %d's dataset lives on device %d, its communicator on %d", c->rank, ds->device, c->device);
        if ((size_t)ds->S != S || (size_t)ds->D != D) return bl_fail(BL_ERR_INVALID, "bl_gather_draws: ranks disagree on num_samples / D");
        if (ds->C != chains_per_rank[c->rank])
            return bl_fail(BL_ERR_INVALID, "bl_gather_draws: rank %d ran %d chains, chains_per_rank says %d", c->rank, ds->C, chains_per_rank[c->rank]);
    }
    // every rank's block size follows from its chain count (the same carve as the launch), so nothing is negotiated
    std::vector<size_t> boff;
    bool equal = true;
    if (int rc = gather_block_offsets(world, chains_per_rank, S, D, boff, &equal)) return rc;
    for (int i = 0; i < n_local; i++) {
        bl_comm *c = comms[i];
        BL_HIP(hipSetDevice(c->device));
        if (boff[world] > c->recv_bytes) {
            if (c->d_recv) hipFree(c->d_recv);
            c->d_recv = nullptr; c->recv_bytes = 0;
            BL_HIP(hipMalloc(&c->d_recv, boff[world]));
            c->recv_bytes = boff[world];
        }
        BL_HIP(hipStreamWaitEvent(c->stream, dss[i]->ev1, 0)); // the kernel's results are complete before they travel
    }
    // ONE collective: ncclAllGather of the blocks (equal chain counts: the usual case), or its "v" form as grouped
    // broadcasts when the chains do not divide evenly
    // (an error inside the group must not leave this thread in RCCL's group mode: remember the first one, always close the group)
    BL_NCCL(api, api->GroupStart());
    ncclResult_t first_err = ncclSuccess;
    const char *first_what = "";
    for (int i = 0; i < n_local && first_err == ncclSuccess; i++) {
        bl_comm *c = comms[i];
        const char *send = (const char *)dss[i]->d_run;
        if (equal) {
            first_err = api->AllGather(send, c->d_recv, boff[1], ncclChar, c->comm, c->stream);
            first_what = "ncclAllGather";
        } else {
            for (int r = 0; r < world && first_err == ncclSuccess; r++) {
                first_err = api->Broadcast(send, (char *)c->d_recv + boff[r], boff[r + 1] - boff[r], ncclChar, r, c->comm, c->stream);
                first_what = "ncclBroadcast";
            }
        }
    }
    const ncclResult_t end_err = api->GroupEnd();
    if (first_err != ncclSuccess) return bl_fail(BL_ERR_COMM, "%s failed: %s", first_what, api->GetErrorString(first_err));
    if (end_err != ncclSuccess) return bl_fail(BL_ERR_COMM, "ncclGroupEnd failed: %s", api->GetErrorString(end_err));
    for (int i = 0; i < n_local; i++) {
        BL_HIP(hipSetDevice(comms[i]->device));
        BL_HIP(hipStreamSynchronize(comms[i]->stream));
    }
    if (out) {
        // (every rank holds all blocks; the first local one hands them to the host)
        bl_comm *c = comms[0];
        std::vector<char> host(boff[world]);
        BL_HIP(hipSetDevice(c->device));
        BL_HIP(hipMemcpy(host.data(), c->d_recv, boff[world], hipMemcpyDeviceToHost));
        unpack_gathered(host.data(), world, chains_per_rank, boff, S, D, out);
    }
    return BL_OK;
}

// The unpacking of a gathered buffer alone (host memory in, host arrays out; no GPU, no RCCL): `gathered` holds the
// world blocks back to back, as bl_gather_draws receives them.
extern "C" int bl_gather_unpack(const void *gathered, uint64_t gathered_bytes, int world, const int32_t *chains_per_rank,
                                int num_samples, int D, bl_nuts_output *out)
{
    if (!gathered || world <= 0 || !chains_per_rank || num_samples < 0 || D <= 0 || !out) return bl_fail(BL_ERR_INVALID, "bl_gather_unpack: bad argument");
    std::vector<size_t> boff;
    bool equal = true;
    if (int rc = gather_block_offsets(world, chains_per_rank, (size_t)num_samples, (size_t)D, boff, &equal)) return rc;
    if ((uint64_t)boff[world] != gathered_bytes)
        return bl_fail(BL_ERR_INVALID, "bl_gather_unpack: %llu bytes given, the %d blocks take %llu", (unsigned long long)gathered_bytes, world, (unsigned long long)boff[world]);
    unpack_gathered((const char *)gathered, world, chains_per_rank, boff, (size_t)num_samples, (size_t)D, out);
    return BL_OK;
}
