// biolith_hip.hip -- host side of the C-ABI in include/biolith_hip.h (+ the small non-templated kernels).
//
// Everything numpyro did behind `mcmc.run` (biolith/utils/fit.py:128-130) for the occu model is
// reached through these entry points.  No CPU fallback exists: without a HIP device every compute
// entry point returns BL_ERR_NO_DEVICE.
#include "../../include/biolith_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "logp_kernel.hpp"
#include "re_kernel.hpp"
#include "nuts_kernel.hpp"

// ------------------------------------------------------------------ errors ----
static thread_local std::string g_err;
static thread_local char g_kernel_name[160]; // the sampler instantiation the last launch of this thread picked (kernels_inst.hip: bl_launch)
static int bl_fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define BL_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return bl_fail(BL_ERR_NO_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),   \
                           __FILE__, __LINE__);                                                        \
    } while (0)

// Scratch device allocations of one call: freed on every return path.
struct DevScratch {
    std::vector<void *> ptrs;
    ~DevScratch() { for (void *q : ptrs) hipFree(q); }
    hipError_t alloc(void **out, size_t bytes)
    {
        const hipError_t e = hipMalloc(out, bytes ? bytes : 1);
        if (e == hipSuccess) ptrs.push_back(*out);
        return e;
    }
};

// ------------------------------------------- kernel instantiation dispatch ----
#if defined(BL_STAMPS) || defined(BL_ONLY33) /* diagnostic / experimental builds: one instantiation */
#define BL_KK_LIST(X) X(3, 3)
#else
#define BL_K_LIST(X, a) X(a, 1) X(a, 2) X(a, 3) X(a, 4) X(a, 8) X(a, 16)
#define BL_KK_LIST(X) BL_K_LIST(X, 1) BL_K_LIST(X, 2) BL_K_LIST(X, 3) BL_K_LIST(X, 4) BL_K_LIST(X, 8) BL_K_LIST(X, 16)
#endif
#define BL_DECL(ks, ko)                                                                                        \
    extern "C" int bl_launch_nuts_##ks##_##ko(const BlNutsParams *, int, int, int, int, hipStream_t);          \
    extern "C" int bl_launch_logp_##ks##_##ko(const BlLogpParams *, int, int, int, int, hipStream_t);
BL_KK_LIST(BL_DECL)

typedef int (*nuts_launch_fn)(const BlNutsParams *, int, int, int, int, hipStream_t);
typedef int (*logp_launch_fn)(const BlLogpParams *, int, int, int, int, hipStream_t);
struct KernelEntry {
    int ks, ko;
    nuts_launch_fn nuts;
    logp_launch_fn logp;
};
#define BL_ENTRY(ks, ko) {ks, ko, bl_launch_nuts_##ks##_##ko, bl_launch_logp_##ks##_##ko},
static const KernelEntry g_kernels[] = {BL_KK_LIST(BL_ENTRY)};

static int pad_covs(int k)
{
    if (k <= 1) return 1;
    if (k <= 4) return k;
    if (k <= 8) return 8;
    return 16;
}
static const KernelEntry *find_kernels(int KS, int KO)
{
    for (const auto &e : g_kernels)
        if (e.ks == KS && e.ko == KO) return &e;
    return nullptr;
}

// ------------------------------------------------------------- host RNG ----
// xoshiro128++ 1.0 with jump(); stream (chain, s) = base advanced by chain*64+s jumps.
static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
static uint32_t xo_next(uint32_t s[4])
{
    const uint32_t result = rotl32(s[0] + s[3], 7) + s[0];
    const uint32_t t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl32(s[3], 11);
    return result;
}
static void xo_jump(uint32_t s[4])
{
    static const uint32_t JUMP[4] = {0x8764000bu, 0xf542d2d3u, 0x6fa035c3u, 0x77f2db5bu};
    uint32_t t[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 32; b++) {
            if (JUMP[i] & (1u << b)) { t[0] ^= s[0]; t[1] ^= s[1]; t[2] ^= s[2]; t[3] ^= s[3]; }
            xo_next(s);
        }
    memcpy(s, t, sizeof t);
}
static uint64_t splitmix64(uint64_t &x)
{
    uint64_t z = (x += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static void rng_streams(uint64_t seed, int chain, int nstreams, uint32_t *out)
{
    uint64_t sm = seed;
    const uint64_t a = splitmix64(sm), b = splitmix64(sm);
    uint32_t s[4] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    if (!(s[0] | s[1] | s[2] | s[3])) s[0] = 1;
    for (long k = 0; k < (long)chain * BL_RNG_STREAMS_PER_CHAIN; k++) xo_jump(s);
    for (int k = 0; k < nstreams; k++) {
        memcpy(out + 4 * k, s, sizeof s);
        xo_jump(s);
    }
}

// Streams of `nchains` consecutive chains at `stride` streams per chain (stream (c, s) = base advanced by c * stride + s
// jumps, like rng_streams with its stride of 64): out[(c - first_chain) * stride + s].
static void rng_streams_strided(uint64_t seed, int first_chain, int stride, int nchains, uint32_t *out)
{
    uint64_t sm = seed;
    const uint64_t a = splitmix64(sm), b = splitmix64(sm);
    uint32_t s[4] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    if (!(s[0] | s[1] | s[2] | s[3])) s[0] = 1;
    for (long long k = 0; k < (long long)first_chain * stride; k++) xo_jump(s);
    for (long long k = 0; k < (long long)nchains * stride; k++) {
        memcpy(out + 4 * k, s, sizeof s);
        xo_jump(s);
    }
}

static int adaptation_schedule(int num_steps, int32_t *starts, int32_t *ends, int cap)
{
    // numpyro.infer.hmc_util.build_adaptation_schedule
    int n = 0;
    auto push = [&](int s, int e) { if (n < cap) { starts[n] = s; ends[n] = e; } n++; };
    if (num_steps <= 0) return 0;
    if (num_steps < 20) { push(0, num_steps - 1); return n; }
    int start_buffer = 75, end_buffer = 50, init_window = 25;
    if (start_buffer + end_buffer + init_window > num_steps) {
        start_buffer = (int)(0.15 * num_steps);
        end_buffer = (int)(0.1 * num_steps);
        init_window = num_steps - start_buffer - end_buffer;
    }
    push(0, start_buffer - 1);
    const int end_window_start = num_steps - end_buffer;
    int next_size = init_window, next_start = start_buffer;
    while (next_start < end_window_start) {
        int cur_start = next_start, cur_size = next_size;
        if (3 * cur_size <= end_window_start - cur_start) next_size = 2 * cur_size;
        else cur_size = end_window_start - cur_start;
        next_start = cur_start + cur_size;
        push(cur_start, next_start - 1);
    }
    push(end_window_start, num_steps - 1);
    return n;
}

// ----------------------------------------------------------------- handle ----
struct bl_dataset {
    int device = 0;
    int model = 0;          // 0 occu, 1 occu_rn, 2 occu with false positives, 3 occu_cop, 4 nmixture, 6 occu with random effects
    BlReModel re{};         // model 6 (re_kernel.hpp); D is then the full coordinate count
    float *d_restate = nullptr; // model 6: sampler state [C][k][RE_SLOTS][dl_max]
    size_t restate_bytes = 0;
    float *d_tab = nullptr; // nmixture: B[t][n][site] = sum_j m log C(n, y_j), -inf below the largest count
    int ko_layout = 0;      // KO the record layout helpers are called with (KO, or KO + 1 for occu_cop's wider visits)
    int max_abundance = 0;  // occu_rn only
    int fp_mode = 0;        // model 2: BL_FP_CONSTANT / BL_FP_UNOCCUPIED
    double fp_a = 2.0, fp_b = 5.0;  // model 2: Beta prior of the false-positive rate
    bl_dims dims{};
    int Ks = 0, Ko = 0, KS = 0, KO = 0, D = 0;
    int nsp = 1;            // species sampled jointly by this handle (models 0 and 2)
    int n_stride = 0, n_rows = 0;
    float *d_rows = nullptr;
    BlDevData dd{};
    const KernelEntry *kern = nullptr;
    bl_normal_prior pb{}, pa{};
    int fam_b = 0, fam_a = 0; // BL_PRIOR_NORMAL / BL_PRIOR_LAPLACE (bl_dataset_set_prior_family)
    // raw nan_to_num'd obs covariates, site-fastest [V][Ko][n_stride], for prob_detection (lazy upload)
    std::vector<float> h_wraw;
    float *d_wraw = nullptr;
    // occu_cop: raw session durations, site-fastest [V][n_stride], for the predictive counts (lazy upload)
    std::vector<float> h_dur;
    float *d_dur = nullptr;
    // ---- last NUTS launch ----
    bool in_flight = false, have_run = false;
    int C = 0, S = 0, W = 0, k = 0, nloc = 0, lds_ld = 0, lds_bytes = 0, staged = 0, nvp = 0, ncw = 0, lane_grp = 0;
    char kernel_name[160] = {0}; // the sampler instantiation of the last launch, as rocprofv3 names it
    char env_overrides[512] = {0}; // BIOLITH_HIP_* knobs that were set when the last launch read the environment (bl_env_snapshot)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void *d_run = nullptr;  // one slab for all run buffers
    size_t run_bytes = 0;
    float *d_draws = nullptr, *d_acc = nullptr, *d_pot = nullptr, *d_eps = nullptr, *d_minv = nullptr, *d_init = nullptr;
    unsigned char *d_div = nullptr;
    int *d_steps = nullptr, *d_status = nullptr;
    long long *d_nleap = nullptr, *d_dbg = nullptr;
    int *d_loc = nullptr;
    uint32_t *d_rng = nullptr;
    float *d_scores = nullptr; // occu_cs: scores, site-fastest [T J][n_stride]
    unsigned long long *d_xchg = nullptr;
    size_t xchg_bytes = 0;
    int *h_abort = nullptr, *d_abort = nullptr;
};


// ------------------------------------------------------------------------------------------ environment knobs ----
// Every environment variable the launch path looks at, in ONE table (INTEGRATION.md lists them with their meaning).  None is needed in
// production: they force a geometry or a kernel form for tests, A/B runs and measurement.  They are read in ONE place --
// bl_env_snapshot(), at the top of the public entries that can reach one (dataset creation, bl_logp_grad, bl_nuts_launch) -- so a launch
// sees one consistent set, and bl_env_overrides() / bl_nuts_env_overrides() report the ones that were set (bench.py prints them: a
// stray variable must not change a geometry silently).  A snapshot per entry rather than per process: the parity tests switch forms
// between two launches of one process (tests/test_gpu_kernel_forms.py, test_gpu_re.py).
#define BL_ENV_TABLE(X) X(DYN_G) X(GRP_VISITS) X(OCCU_G) X(OCCU_GT) X(SINGLE) X(CWAVES) X(NO_WIDE) X(RN_K) X(WIDE_K) X(NMIX_LDS) X(RE_EFF) \
    X(RE_NO_SPLIT) X(RE_LDS_ROWS) X(RE_LDS_TIER) X(RE_WGS) X(NO_LOCAL) X(PITCH) X(GRP_KERNEL) X(SPIN_US) X(POLL_SLEEP) X(FIRST_DELAY) X(GENERAL)
enum bl_env_knob {
#define X(n) BL_ENV_##n,
    BL_ENV_TABLE(X)
#undef X
    BL_ENV_COUNT
};
static const char *const BL_ENV_NAMES[BL_ENV_COUNT] = {
#define X(n) "BIOLITH_HIP_" #n,
    BL_ENV_TABLE(X)
#undef X
};
struct bl_env_config {
    bool set[BL_ENV_COUNT] = {};
    char val[BL_ENV_COUNT][32] = {};
};
static bl_env_config g_env;
static void bl_env_snapshot()
{
    for (int i = 0; i < BL_ENV_COUNT; i++) {
        const char *e = getenv(BL_ENV_NAMES[i]);
        g_env.set[i] = e != nullptr;
        snprintf(g_env.val[i], sizeof g_env.val[i], "%s", e ? e : "");
    }
}
// the knob's value at the last snapshot, or NULL when it was not set
static const char *bl_env(bl_env_knob k) { return g_env.set[k] ? g_env.val[k] : nullptr; }
static std::string bl_env_active()
{
    std::string out;
    for (int i = 0; i < BL_ENV_COUNT; i++)
        if (g_env.set[i]) out += (out.empty() ? "" : ",") + std::string(BL_ENV_NAMES[i]) + "=" + g_env.val[i];
    return out;
}
extern "C" int bl_env_force_general(void) { const char *e = bl_env(BL_ENV_GENERAL); return e && e[0] == '1'; } // (kernels_inst.hip)
extern "C" int bl_env_overrides(char *buf, int n)
{
    if (!buf || n <= 0) return bl_fail(BL_ERR_INVALID, "NULL argument");
    bl_env_snapshot();
    snprintf(buf, (size_t)n, "%s", bl_env_active().c_str());
    return BL_OK;
}

static int set_device(const bl_dataset *ds)
{
    BL_HIP(hipSetDevice(ds->device));
    return BL_OK;
}

extern "C" int bl_abi_version(void) { return BL_ABI_VERSION; }
extern "C" const char *bl_last_error(void) { return g_err.c_str(); }

extern "C" int bl_device_count(int *count)
{
    if (!count) return bl_fail(BL_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return bl_fail(BL_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return BL_OK;
}

// ------------------------------------------------------------------ posterior predictive ----
// One generator per (draw, period, site): xoshiro128++ keyed by splitmix64 of the flat index, so the
// sample does not depend on the launch geometry or on how the draws are chunked.
__device__ inline unsigned long long bl_splitmix(unsigned long long &x)
{
    unsigned long long z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct BlPredRng {
    unsigned s0, s1, s2, s3;
    __device__ BlPredRng(unsigned long long seed, unsigned long long index)
    {
        unsigned long long x = seed ^ (index * 0xD1342543DE82EF95ull);
        const unsigned long long a = bl_splitmix(x), b = bl_splitmix(x);
        s0 = (unsigned)a; s1 = (unsigned)(a >> 32); s2 = (unsigned)b; s3 = (unsigned)(b >> 32) | 1u;
    }
    __device__ float uniform() // [0, 1)
    {
        const unsigned r0 = s0 + s3, r = ((r0 << 7) | (r0 >> 25)) + s0, t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = (s3 << 11) | (s3 >> 21);
        return (float)(r >> 8) * 5.9604644775390625e-08f;
    }
};
// occu (occu.py:207-241 with obs=None):  z ~ Bernoulli(psi),  y_j ~ Bernoulli(z * p_j)
// occu_rn (occu_rn.py:192-221):          N ~ Categorical(Poisson(lambda) pmf on 0..K),  y_j ~ Bernoulli(1 - (1 - r_j)^N)
__global__ void bl_predict_kernel(const float *__restrict__ rows, const float *__restrict__ wraw, int n_stride, int N, int T, int J,
                                  int Ks, int Ko, int D, const float *__restrict__ draws, int n0, int n1,
                                  unsigned long long seed, int model, int max_abundance, int fp_mode,
                                  unsigned char *__restrict__ latent, unsigned char *__restrict__ y, int o_u, int o_v, int o_e, int o_fp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float x[BL_MAX_COVS];
    for (int k = 0; k < Ks; k++) x[k] = rows[(size_t)k * n_stride + i];
    for (int n = n0 + blockIdx.y; n < n1; n += gridDim.y) {
        const float *th = draws + (size_t)n * D;
        const float *al = th + Ks + 1;
        float eta = th[0];
        for (int k = 0; k < Ks; k++) eta = fmaf(x[k], th[k + 1], eta);
        if (o_u >= 0) eta += th[o_u + i]; // random effects (model 6; offsets into a draw, -1 = absent): occu.py:198-202, 221-228
        // false-positive rate (model 2): acts on every site ("constant") or on unoccupied ones only
        // (o_fp: where phi = logit(rate) sits in a draw -- the last coordinate, or right behind the coefficients with random effects)
        const float fpr = (model == 2 || (model == 1 && fp_mode == BL_FP_CONSTANT && o_fp >= 0)) ? 1.0f / (1.0f + __expf(-th[o_fp])) : 0.0f;
        const float f_c = fp_mode == BL_FP_CONSTANT ? fpr : 0.0f, f_u = fp_mode == BL_FP_UNOCCUPIED ? fpr : 0.0f;
        for (int t = 0; t < T; t++) {
            BlPredRng rng(seed, ((unsigned long long)n * T + t) * N + i);
            int zn;
            if (model == 1) {
                // inversion over the (renormalised) truncated Poisson pmf, float64 recursion p_n = p_{n-1} lambda / n
                const double lam = exp((double)eta);
                double p = exp(-lam), tot = 0.0;
                for (int m = 0; m <= max_abundance; m++) { tot += p; p *= lam / (double)(m + 1); }
                const double target = (double)rng.uniform() * tot;
                p = exp(-lam);
                double cum = 0.0;
                zn = max_abundance;
                for (int m = 0; m <= max_abundance; m++) {
                    cum += p;
                    if (target < cum) { zn = m; break; }
                    p *= lam / (double)(m + 1);
                }
            } else {
                const float psi = 1.0f / (1.0f + __expf(-eta));
                zn = rng.uniform() < psi ? 1 : 0;
            }
            if (latent) latent[((size_t)(n - n0) * T + t) * N + i] = (unsigned char)zn;
            if (!y) continue;
            for (int j = 0; j < J; j++) {
                const int v = t * J + j;
                float nu = al[0];
                for (int k = 0; k < Ko; k++) nu = fmaf(wraw[((size_t)v * Ko + k) * n_stride + i], al[k + 1], nu);
                if (o_v >= 0) nu += th[o_v + i];
                if (o_e >= 0) nu += th[o_e + (size_t)i * T * J + v];
                const float r = 1.0f / (1.0f + __expf(-nu));
                float pd = model == 1 ? 1.0f - __powf(1.0f - r, (float)zn) : (float)zn * r;
                if (model == 2) pd = 1.0f - (1.0f - pd) * (1.0f - f_c) * (1.0f - (zn ? 0.0f : f_u));
                if (model == 1) pd = 1.0f - (1.0f - pd) * (1.0f - fpr); // (Royle-Nichols with a false-positive rate: occu_rn.py:214-221; fpr = 0 without)
                const float u = rng.uniform();
                y[(((size_t)(n - n0) * J + j) * T + t) * N + i] = (u < pd) ? 1 : 0;
            }
        }
    }
}

extern "C" int bl_predict(bl_dataset *ds, int n_draws, const float *draws, uint64_t seed, uint8_t *latent, uint8_t *y)
{
    if (!ds || !draws || n_draws <= 0 || (!latent && !y)) return bl_fail(BL_ERR_INVALID, "bl_predict: bad argument");
    if (ds->nsp > 1) return bl_fail(BL_ERR_UNSUPPORTED, "bl_predict: a joint-species handle samples; predict from one handle per species");
    if (ds->model == 8) return bl_fail(BL_ERR_UNSUPPORTED, "bl_predict: not built for the dynamic occupancy model");
    if (ds->model == 3 || ds->model == 4)
        return bl_fail(BL_ERR_UNSUPPORTED, "bl_predict: the count models (occu_cop, nmixture) use bl_predict_counts");
    if (ds->model == 6 && ds->re.kind == 1)
        return bl_fail(BL_ERR_UNSUPPORTED, "bl_predict: not built for occu_cs (its observed site is a continuous score)");
    if (ds->model == 6 && (ds->re.kind == 3 || ds->re.kind >= 6))
        return bl_fail(BL_ERR_UNSUPPORTED, "bl_predict: the count models' sampled sites are counts (bl_predict_counts)");
    if (ds->in_flight) return bl_fail(BL_ERR_BUSY, "a NUTS launch is in flight on this handle");
    int rc = set_device(ds);
    if (rc) return rc;
    const int N = ds->dims.n_sites, T = ds->dims.n_periods, J = ds->dims.n_replicates, D = ds->D;
    float *d_draws = nullptr;
    unsigned char *d_lat = nullptr, *d_y = nullptr;
    DevScratch scratch;
    BL_HIP(scratch.alloc((void **)&d_draws, (size_t)n_draws * D * 4));
    BL_HIP(hipMemcpy(d_draws, draws, (size_t)n_draws * D * 4, hipMemcpyHostToDevice));
    const size_t per_draw = (size_t)T * N * (y ? (size_t)J : 1); // bytes of the larger output
    int chunk = (int)((256u << 20) / (per_draw ? per_draw : 1));
    if (chunk < 1) chunk = 1;
    if (chunk > n_draws) chunk = n_draws;
    if (latent) BL_HIP(scratch.alloc((void **)&d_lat, (size_t)chunk * T * N));
    if (y) BL_HIP(scratch.alloc((void **)&d_y, (size_t)chunk * J * T * N));
    if (!ds->d_wraw) {
        BL_HIP(hipMalloc((void **)&ds->d_wraw, ds->h_wraw.size() * 4));
        BL_HIP(hipMemcpy(ds->d_wraw, ds->h_wraw.data(), ds->h_wraw.size() * 4, hipMemcpyHostToDevice));
    }
    const dim3 block(256);
    for (int n0 = 0; n0 < n_draws; n0 += chunk) {
        const int n1 = (n0 + chunk < n_draws) ? n0 + chunk : n_draws;
        const dim3 grid((N + 255) / 256, (n1 - n0) < 1024 ? (n1 - n0) : 1024);
        hipLaunchKernelGGL(bl_predict_kernel, grid, block, 0, nullptr, ds->d_rows, ds->d_wraw, ds->n_stride, N, T, J, ds->Ks, ds->Ko, D,
                           d_draws, n0, n1, (unsigned long long)seed,
                           // random-effects handles: Royle-Nichols (kind 4) / false positives (kind 2) run those branches with the effects
                           ds->model == 6 && (ds->re.kind == 4 || ds->re.kind == 5) ? 1 : (ds->model == 6 && ds->re.kind == 2 ? 2 : ds->model),
                           ds->max_abundance, ds->model == 6 && (ds->re.kind == 2 || ds->re.kind == 5) ? ds->re.fp_mode : ds->fp_mode, d_lat, d_y,
                           ds->model == 6 ? ds->re.o_u : -1, ds->model == 6 ? ds->re.o_v : -1, ds->model == 6 ? ds->re.o_e : -1,
                           ds->model == 6 ? ((ds->re.kind == 2 || ds->re.kind == 5) ? ds->re.o_fp : -1) : D - 1);
        BL_HIP(hipGetLastError());
        if (latent) BL_HIP(hipMemcpy(latent + (size_t)n0 * T * N, d_lat, (size_t)(n1 - n0) * T * N, hipMemcpyDeviceToHost));
        if (y) BL_HIP(hipMemcpy(y + (size_t)n0 * J * T * N, d_y, (size_t)(n1 - n0) * J * T * N, hipMemcpyDeviceToHost));
    }
    return BL_OK;
}

// ---- predictive scores of the continuous-score model (occu_cs.py:196-232 with obs=None) ----
// z ~ Bernoulli(psi);  f_j ~ Bernoulli(z p_j);  s_j ~ Normal(mu_f, sigma_f)   (draw = [beta, alpha, mu0, log(mu1 - mu0), log sigma0, log sigma1])
__global__ void bl_predict_scores_kernel(const float *__restrict__ rows, const float *__restrict__ wraw, int n_stride, int N, int T, int J,
                                         int Ks, int Ko, int D, const float *__restrict__ draws, int n0, int n1, unsigned long long seed,
                                         unsigned char *__restrict__ latent, unsigned char *__restrict__ f_out, float *__restrict__ s_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float x[BL_MAX_COVS];
    for (int k = 0; k < Ks; k++) x[k] = rows[(size_t)k * n_stride + i];
    for (int n = n0 + blockIdx.y; n < n1; n += gridDim.y) {
        const float *th = draws + (size_t)n * D, *al = th + Ks + 1, *ex = th + Ks + Ko + 2;
        const float mu0 = ex[0], mu1 = ex[0] + __expf(ex[1]), sg0 = __expf(ex[2]), sg1 = __expf(ex[3]);
        float eta = th[0];
        for (int k = 0; k < Ks; k++) eta = fmaf(x[k], th[k + 1], eta);
        const float psi = 1.0f / (1.0f + __expf(-eta));
        for (int t = 0; t < T; t++) {
            BlPredRng rng(seed, ((unsigned long long)n * T + t) * N + i);
            const int zn = rng.uniform() < psi ? 1 : 0;
            if (latent) latent[((size_t)(n - n0) * T + t) * N + i] = (unsigned char)zn;
            for (int j = 0; j < J; j++) {
                const int v = t * J + j;
                float nu = al[0];
                for (int k = 0; k < Ko; k++) nu = fmaf(wraw[((size_t)v * Ko + k) * n_stride + i], al[k + 1], nu);
                const int fn = rng.uniform() < (float)zn / (1.0f + __expf(-nu)) ? 1 : 0;
                // Box-Muller, one normal per replicate
                const float u1 = fmaxf(rng.uniform(), 5.9604645e-08f), u2 = rng.uniform();
                const float g = sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
                const size_t o = (((size_t)(n - n0) * J + j) * T + t) * N + i;
                if (f_out) f_out[o] = (unsigned char)fn;
                if (s_out) s_out[o] = fn ? fmaf(sg1, g, mu1) : fmaf(sg0, g, mu0);
            }
        }
    }
}

extern "C" int bl_predict_scores(bl_dataset *ds, int n_draws, const float *draws, uint64_t seed, uint8_t *latent, uint8_t *f, float *s)
{
    if (!ds || !draws || n_draws <= 0 || (!latent && !f && !s)) return bl_fail(BL_ERR_INVALID, "bl_predict_scores: bad argument");
    if (!(ds->model == 6 && ds->re.kind == 1)) return bl_fail(BL_ERR_UNSUPPORTED, "bl_predict_scores: an occu_cs dataset is required");
    if (ds->in_flight) return bl_fail(BL_ERR_BUSY, "a NUTS launch is in flight on this handle");
    int rc = set_device(ds);
    if (rc) return rc;
    const int N = ds->dims.n_sites, T = ds->dims.n_periods, J = ds->dims.n_replicates, D = ds->D;
    float *d_draws = nullptr, *d_s = nullptr;
    unsigned char *d_lat = nullptr, *d_f = nullptr;
    DevScratch scratch;
    BL_HIP(scratch.alloc((void **)&d_draws, (size_t)n_draws * D * 4));
    BL_HIP(hipMemcpy(d_draws, draws, (size_t)n_draws * D * 4, hipMemcpyHostToDevice));
    const size_t per_draw = (size_t)T * N * J * 4;
    int chunk = (int)((256u << 20) / (per_draw ? per_draw : 1));
    if (chunk < 1) chunk = 1;
    if (chunk > n_draws) chunk = n_draws;
    if (latent) BL_HIP(scratch.alloc((void **)&d_lat, (size_t)chunk * T * N));
    if (f) BL_HIP(scratch.alloc((void **)&d_f, (size_t)chunk * J * T * N));
    if (s) BL_HIP(scratch.alloc((void **)&d_s, (size_t)chunk * J * T * N * 4));
    if (!ds->d_wraw) {
        BL_HIP(hipMalloc((void **)&ds->d_wraw, ds->h_wraw.size() * 4));
        BL_HIP(hipMemcpy(ds->d_wraw, ds->h_wraw.data(), ds->h_wraw.size() * 4, hipMemcpyHostToDevice));
    }
    const dim3 block(256);
    for (int n0 = 0; n0 < n_draws; n0 += chunk) {
        const int n1 = (n0 + chunk < n_draws) ? n0 + chunk : n_draws;
        const dim3 grid((N + 255) / 256, (n1 - n0) < 1024 ? (n1 - n0) : 1024);
        hipLaunchKernelGGL(bl_predict_scores_kernel, grid, block, 0, nullptr, ds->d_rows, ds->d_wraw, ds->n_stride, N, T, J, ds->Ks, ds->Ko, D,
                           d_draws, n0, n1, (unsigned long long)seed, d_lat, d_f, d_s);
        BL_HIP(hipGetLastError());
        if (latent) BL_HIP(hipMemcpy(latent + (size_t)n0 * T * N, d_lat, (size_t)(n1 - n0) * T * N, hipMemcpyDeviceToHost));
        if (f) BL_HIP(hipMemcpy(f + (size_t)n0 * J * T * N, d_f, (size_t)(n1 - n0) * J * T * N, hipMemcpyDeviceToHost));
        if (s) BL_HIP(hipMemcpy(s + (size_t)n0 * J * T * N, d_s, (size_t)(n1 - n0) * J * T * N * 4, hipMemcpyDeviceToHost));
    }
    return BL_OK;
}

// ---- predictive counts of the count models (occu_cop, nmixture) ----
// Poisson(lam): inversion by sequential search for lam < 10, else Hoermann's PTRS transformed rejection
// ("The transformed rejection method for generating Poisson random variables", 1993); both exact.
__device__ inline int bl_poisson(BlPredRng &rng, double lam)
{
    if (!(lam > 0.0)) return 0;
    if (lam < 10.0) {
        const double enlam = exp(-lam);
        int k = 0;
        double prod = (double)rng.uniform();
        while (prod > enlam && k < 1000) { prod *= (double)rng.uniform(); k++; }
        return k;
    }
    const double slam = sqrt(lam), loglam = log(lam);
    const double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
    for (int it = 0; it < 1000; it++) {
        const double U = (double)rng.uniform() - 0.5, V = (double)rng.uniform();
        const double us = 0.5 - fabs(U);
        const double kf = floor((2.0 * a / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return (int)kf;
        if (kf < 0.0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lam + kf * loglam - lgamma(kf + 1.0)) return (int)kf;
    }
    return (int)lam;
}
// occu_cop (occu_cop.py:222-255, obs withheld):  z ~ Bernoulli(psi),  y_j ~ Poisson(dur_j (z lambda_j + (1 - z) f_u + f_c))
// nmixture (nmixture.py:183-220, obs withheld):  N ~ Poisson(lambda) restricted to 0..K,  y_j ~ Binomial(N, p_j)
__global__ void bl_predict_counts_kernel(const float *__restrict__ rows, const float *__restrict__ wraw, const float *__restrict__ dur,
                                         int n_stride, int N, int T, int J, int Ks, int Ko, int D,
                                         const float *__restrict__ draws, int n0, int n1, unsigned long long seed, int model,
                                         int max_abundance, int fp_mode, int *__restrict__ latent, int *__restrict__ y,
                                         int o_u, int o_v, int o_e, int o_fp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float x[BL_MAX_COVS];
    for (int k = 0; k < Ks; k++) x[k] = rows[(size_t)k * n_stride + i];
    for (int n = n0 + blockIdx.y; n < n1; n += gridDim.y) {
        const float *th = draws + (size_t)n * D;
        const float *al = th + Ks + 1;
        float eta = th[0];
        for (int k = 0; k < Ks; k++) eta = fmaf(x[k], th[k + 1], eta);
        if (o_u >= 0) eta += th[o_u + i]; // random effects (offsets into a draw, -1 = absent): nmixture.py:166-172, 199-214
        const float f = (model == 3 && fp_mode) ? __expf(th[o_fp]) : 0.0f; // (the last coordinate, or right behind the coefficients with random effects)
        const float f_c = fp_mode == BL_FP_CONSTANT ? f : 0.0f, f_u = fp_mode == BL_FP_UNOCCUPIED ? f : 0.0f;
        for (int t = 0; t < T; t++) {
            BlPredRng rng(seed, ((unsigned long long)n * T + t) * N + i);
            int zn;
            if (model == 4) { // inversion over the renormalised truncated Poisson pmf (float64 recursion)
                const double lam = exp((double)eta);
                double p = exp(-lam), tot = 0.0;
                for (int m = 0; m <= max_abundance; m++) { tot += p; p *= lam / (double)(m + 1); }
                const double target = (double)rng.uniform() * tot;
                p = exp(-lam);
                double cum = 0.0;
                zn = max_abundance;
                for (int m = 0; m <= max_abundance; m++) {
                    cum += p;
                    if (target < cum) { zn = m; break; }
                    p *= lam / (double)(m + 1);
                }
            } else {
                zn = rng.uniform() < 1.0f / (1.0f + __expf(-eta)) ? 1 : 0;
            }
            if (latent) latent[((size_t)(n - n0) * T + t) * N + i] = zn;
            if (!y) continue;
            for (int j = 0; j < J; j++) {
                const int v = t * J + j;
                float nu = al[0];
                for (int k = 0; k < Ko; k++) nu = fmaf(wraw[((size_t)v * Ko + k) * n_stride + i], al[k + 1], nu);
                if (o_v >= 0) nu += th[o_v + i];
                if (o_e >= 0) nu += th[o_e + (size_t)i * T * J + v];
                int cnt = 0;
                if (model == 4) {
                    const float p = 1.0f / (1.0f + __expf(-nu));
                    for (int m = 0; m < zn; m++) cnt += rng.uniform() < p ? 1 : 0; // Binomial(N, p), N <= 127
                } else {
                    const double rate = (double)dur[(size_t)v * n_stride + i] * ((zn ? (double)__expf(nu) : (double)f_u) + (double)f_c);
                    cnt = bl_poisson(rng, rate);
                }
                y[(((size_t)(n - n0) * J + j) * T + t) * N + i] = cnt;
            }
        }
    }
}

extern "C" int bl_predict_counts(bl_dataset *ds, int n_draws, const float *draws, uint64_t seed, int32_t *latent, int32_t *y)
{
    if (!ds || !draws || n_draws <= 0 || (!latent && !y)) return bl_fail(BL_ERR_INVALID, "bl_predict_counts: bad argument");
    const bool nmix_re = ds->model == 6 && ds->re.kind == 3; // the N-mixture model with random effects
    const bool cop_re = ds->model == 6 && ds->re.kind >= 6;  // occu_cop with random effects (and a false-positive rate: kind 7)
    if (ds->model != 3 && ds->model != 4 && !nmix_re && !cop_re)
        return bl_fail(BL_ERR_UNSUPPORTED, "bl_predict_counts: for the count models (occu_cop, nmixture); use bl_predict");
    if (ds->in_flight) return bl_fail(BL_ERR_BUSY, "a NUTS launch is in flight on this handle");
    int rc = set_device(ds);
    if (rc) return rc;
    const int N = ds->dims.n_sites, T = ds->dims.n_periods, J = ds->dims.n_replicates, D = ds->D;
    float *d_draws = nullptr;
    int *d_lat = nullptr, *d_y = nullptr;
    DevScratch scratch;
    BL_HIP(scratch.alloc((void **)&d_draws, (size_t)n_draws * D * 4));
    BL_HIP(hipMemcpy(d_draws, draws, (size_t)n_draws * D * 4, hipMemcpyHostToDevice));
    const size_t per_draw = (size_t)T * N * 4 * (y ? (size_t)J : 1);
    int chunk = (int)((256u << 20) / (per_draw ? per_draw : 1));
    if (chunk < 1) chunk = 1;
    if (chunk > n_draws) chunk = n_draws;
    if (latent) BL_HIP(scratch.alloc((void **)&d_lat, (size_t)chunk * T * N * 4));
    if (y) BL_HIP(scratch.alloc((void **)&d_y, (size_t)chunk * J * T * N * 4));
    if (!ds->d_wraw) {
        BL_HIP(hipMalloc((void **)&ds->d_wraw, ds->h_wraw.size() * 4));
        BL_HIP(hipMemcpy(ds->d_wraw, ds->h_wraw.data(), ds->h_wraw.size() * 4, hipMemcpyHostToDevice));
    }
    if ((ds->model == 3 || cop_re) && !ds->d_dur) {
        BL_HIP(hipMalloc((void **)&ds->d_dur, ds->h_dur.size() * 4));
        BL_HIP(hipMemcpy(ds->d_dur, ds->h_dur.data(), ds->h_dur.size() * 4, hipMemcpyHostToDevice));
    }
    const dim3 block(256);
    for (int n0 = 0; n0 < n_draws; n0 += chunk) {
        const int n1 = (n0 + chunk < n_draws) ? n0 + chunk : n_draws;
        const dim3 grid((N + 255) / 256, (n1 - n0) < 1024 ? (n1 - n0) : 1024);
        hipLaunchKernelGGL(bl_predict_counts_kernel, grid, block, 0, nullptr, ds->d_rows, ds->d_wraw, ds->d_dur, ds->n_stride, N, T, J,
                           ds->Ks, ds->Ko, D, d_draws, n0, n1, (unsigned long long)seed, nmix_re ? 4 : (cop_re ? 3 : ds->model), ds->max_abundance, ds->fp_mode,
                           d_lat, d_y, (nmix_re || cop_re) ? ds->re.o_u : -1, (nmix_re || cop_re) ? ds->re.o_v : -1, (nmix_re || cop_re) ? ds->re.o_e : -1,
                           cop_re && ds->fp_mode ? ds->re.o_fp : D - 1);
        BL_HIP(hipGetLastError());
        if (latent) BL_HIP(hipMemcpy(latent + (size_t)n0 * T * N, d_lat, (size_t)(n1 - n0) * T * N * 4, hipMemcpyDeviceToHost));
        if (y) BL_HIP(hipMemcpy(y + (size_t)n0 * J * T * N, d_y, (size_t)(n1 - n0) * J * T * N * 4, hipMemcpyDeviceToHost));
    }
    return BL_OK;
}

extern "C" int bl_rng_streams(uint64_t seed, int chain, int nstreams, uint32_t *out)
{
    if (!out || chain < 0 || nstreams < 0) return bl_fail(BL_ERR_INVALID, "bl_rng_streams: bad argument");
    rng_streams(seed, chain, nstreams, out);
    return BL_OK;
}

extern "C" int bl_adaptation_schedule(int num_warmup, int32_t *starts, int32_t *ends, int capacity)
{
    if (!starts || !ends || capacity <= 0) return -1;
    return adaptation_schedule(num_warmup, starts, ends, capacity);
}

struct ModelOpts {
    int model = 0, max_abundance = 0, fp_mode = 0;
    double fp_a = 2.0, fp_b = 5.0;           // model 2: Beta(a, b); model 3: Exponential(rate = fp_a)
    const float *session_duration = nullptr; // model 3: [N][T][J]
    bool vector_kernels = false;             // the rows feed re_kernel.hpp (random effects): no 64-lane limit on species x coefficients
};
static int dataset_create_impl(const ModelOpts &mo, const bl_dims *dims, const float *site_covs, const float *obs_covs,
                               const float *obs, const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha,
                               int device, bl_dataset **out);

extern "C" int bl_dataset_create(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                                 const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha, int device,
                                 bl_dataset **out)
{
    return dataset_create_impl(ModelOpts{}, dims, site_covs, obs_covs, obs, prior_beta, prior_alpha, device, out);
}

extern "C" int bl_dataset_create_rn(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                                    int max_abundance, const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha,
                                    int device, bl_dataset **out)
{
    if (max_abundance < 1 || max_abundance >= BL_RN_NB)
        return bl_fail(BL_ERR_UNSUPPORTED, "max_abundance=%d outside 1..%d", max_abundance, BL_RN_NB - 1);
    ModelOpts mo; mo.model = 1; mo.max_abundance = max_abundance;
    return dataset_create_impl(mo, dims, site_covs, obs_covs, obs, prior_beta, prior_alpha, device, out);
}

extern "C" int bl_dataset_create_dyn(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                                     const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha, int device, bl_dataset **out)
{
    if (dims && dims->n_site_covs > BL_DYN_MAX_KS)
        return bl_fail(BL_ERR_UNSUPPORTED, "dynamic occupancy: at most %d site covariates (three coefficient blocks), got %d", BL_DYN_MAX_KS, dims->n_site_covs);
    ModelOpts mo; mo.model = 8; // (6 names the random-effects / occu_cs handles)
    return dataset_create_impl(mo, dims, site_covs, obs_covs, obs, prior_beta, prior_alpha, device, out);
}

extern "C" int bl_dataset_create_fp(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                                    int fp_mode, const bl_beta_prior *prior_fp, const bl_normal_prior *prior_beta,
                                    const bl_normal_prior *prior_alpha, int device, bl_dataset **out)
{
    if (fp_mode != BL_FP_CONSTANT && fp_mode != BL_FP_UNOCCUPIED)
        return bl_fail(BL_ERR_INVALID, "fp_mode must be BL_FP_CONSTANT or BL_FP_UNOCCUPIED");
    ModelOpts mo; mo.model = 2; mo.fp_mode = fp_mode;
    if (prior_fp) { mo.fp_a = prior_fp->a; mo.fp_b = prior_fp->b; }
    if (!(mo.fp_a > 0.0) || !(mo.fp_b > 0.0) || !std::isfinite(mo.fp_a) || !std::isfinite(mo.fp_b))
        return bl_fail(BL_ERR_INVALID, "Beta prior needs finite a, b > 0");
    return dataset_create_impl(mo, dims, site_covs, obs_covs, obs, prior_beta, prior_alpha, device, out);
}

extern "C" int bl_dataset_create_nmix(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                                      int max_abundance, const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha,
                                      int device, bl_dataset **out)
{
    if (max_abundance < 1 || max_abundance >= BL_RN_NB)
        return bl_fail(BL_ERR_UNSUPPORTED, "max_abundance=%d outside 1..%d", max_abundance, BL_RN_NB - 1);
    ModelOpts mo; mo.model = 4; mo.max_abundance = max_abundance;
    return dataset_create_impl(mo, dims, site_covs, obs_covs, obs, prior_beta, prior_alpha, device, out);
}

extern "C" int bl_dataset_create_cop(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                                     const float *session_duration, int fp_mode, double prior_fp_rate,
                                     const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha, int device,
                                     bl_dataset **out)
{
    if (fp_mode != 0 && fp_mode != BL_FP_CONSTANT && fp_mode != BL_FP_UNOCCUPIED)
        return bl_fail(BL_ERR_INVALID, "fp_mode must be 0, BL_FP_CONSTANT or BL_FP_UNOCCUPIED");
    if (!session_duration) return bl_fail(BL_ERR_INVALID, "session_duration is NULL");
    ModelOpts mo; mo.model = 3; mo.fp_mode = fp_mode; mo.session_duration = session_duration;
    mo.fp_a = fp_mode ? prior_fp_rate : 1.0;
    if (!(mo.fp_a > 0.0) || !std::isfinite(mo.fp_a)) return bl_fail(BL_ERR_INVALID, "Exponential prior needs a finite rate > 0");
    return dataset_create_impl(mo, dims, site_covs, obs_covs, obs, prior_beta, prior_alpha, device, out);
}

static int dataset_create_impl(const ModelOpts &mo, const bl_dims *dims, const float *site_covs, const float *obs_covs,
                               const float *obs, const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha,
                               int device, bl_dataset **out)
{
    const int model = mo.model, max_abundance = mo.max_abundance;
    if (!dims || !out) return bl_fail(BL_ERR_INVALID, "dims/out is NULL");
    bl_env_snapshot();
    *out = nullptr;
    const int S = dims->n_species, N = dims->n_sites, T = dims->n_periods, J = dims->n_replicates;
    const int Ks = dims->n_site_covs, Ko = dims->n_obs_covs;
    // shape checks mirror the asserts at biolith/models/occu.py:103-133
    if (N <= 0 || T <= 0 || J <= 0 || Ks < 0 || Ko < 0) return bl_fail(BL_ERR_INVALID, "non-positive dimension");
    // Several species in ONE handle = one chain over all species' coefficients (the species plate of occu.py:182-186 under one
    // NUTS, with a false-positive rate shared across species, occu.py:146-157): the LDS-staged occu / false-positive forms only.
    if (S < 1) return bl_fail(BL_ERR_INVALID, "n_species=%d", S);
    if (model == 8 && S != 1) return bl_fail(BL_ERR_UNSUPPORTED, "dynamic occupancy: one species per dataset");
    if (S > 1 && model != 0 && model != 2)
        return bl_fail(BL_ERR_UNSUPPORTED, "n_species=%d: joint sampling of several species is built for occu with or without false positives; "
                       "this model takes one species per dataset", S);
    if (Ks > BL_MAX_COVS || Ko > BL_MAX_COVS)
        return bl_fail(BL_ERR_UNSUPPORTED, "more than %d covariates per side (Ks=%d, Ko=%d)", BL_MAX_COVS, Ks, Ko);
    if ((Ks > 0 && !site_covs) || (Ko > 0 && !obs_covs) || !obs) return bl_fail(BL_ERR_INVALID, "NULL data pointer");
    const bl_normal_prior pb = prior_beta ? *prior_beta : bl_normal_prior{0.0, 1.0};
    const bl_normal_prior pa = prior_alpha ? *prior_alpha : bl_normal_prior{0.0, 1.0};
    if (!(pb.scale > 0.0) || !(pa.scale > 0.0) || !std::isfinite(pb.loc) || !std::isfinite(pa.loc))
        return bl_fail(BL_ERR_INVALID, "Normal prior needs finite loc and scale > 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return bl_fail(BL_ERR_NO_DEVICE, "no HIP device visible: the occupancy engine has no CPU fallback");
    if (device < 0 || device >= ndev) return bl_fail(BL_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);

    bl_dataset *ds = new bl_dataset();
    const int has_extra = (model == 2 || (model == 3 && mo.fp_mode != 0)) ? 1 : 0; // trailing false-positive coordinate
    ds->device = device; ds->dims = *dims; ds->Ks = Ks; ds->Ko = Ko; ds->D = S * (Ks + Ko + 2) + has_extra;
    if (model == 8) ds->D = 3 * (Ks + 1) + Ko + 1; // dynamic occupancy: [b_psi | b_gamma | b_eps | alpha]
    ds->nsp = S;
    ds->model = model; ds->max_abundance = max_abundance;
    ds->fp_mode = mo.fp_mode; ds->fp_a = mo.fp_a; ds->fp_b = mo.fp_b;
    ds->KS = pad_covs(Ks); ds->KO = pad_covs(Ko);
    ds->kern = find_kernels(ds->KS, ds->KO);
    ds->pb = pb; ds->pa = pa;
    if (!ds->kern) { delete ds; return bl_fail(BL_ERR_UNSUPPORTED, "no kernel for capacity (%d,%d)", pad_covs(Ks), pad_covs(Ko)); }
    const int V = T * J, KS = ds->KS, KO = ds->KO;
    if (S > 1 && !mo.vector_kernels && (S * BL_SP_COEF(KS, KO) + 2 > 64 || S * BL_SP_PART(KS, KO) * BL_CWAVES_MAX > BL_PART_FLOATS || ds->D + 4 > 64)) {
        delete ds;
        return bl_fail(BL_ERR_UNSUPPORTED, "n_species=%d x (Ks=%d, Ko=%d): too many coordinates for one joint chain (sample the species one by one)", S, Ks, Ko);
    }
    // floats per visit: (c, c w_1 .. c w_KO), or (y, dur | m, w..) for the count models; occu_rn keeps one more, zero here, that its
    // kernel rewrites at every evaluation (rn_device.hpp)
    const int vw = KO + 1 + ((model == 1 || model == 3 || model == 4) ? 1 : 0);
    ds->ko_layout = vw - 1;
    const int n_stride = (N + 63) / 64 * 64;
    const int sp_rows = V * vw + 2 * T;          // rows of one species' block (visits, ka, kb); the site covariates come once
    const int n_rows = KS + S * sp_rows;
    ds->n_stride = n_stride; ds->n_rows = n_rows;

    // ---- pack: mask (occu.py:136-142, modeling.py:15-17), NaN->0, sign folding, site-fastest rows ----
    std::vector<float> rows((size_t)n_rows * n_stride, 0.0f);
    ds->h_wraw.assign((size_t)V * (Ko > 0 ? Ko : 1) * n_stride, 0.0f);
    if (model == 3) ds->h_dur.assign((size_t)V * n_stride, 0.0f);
    const double LN2 = 0.69314718055994530942, LOG_TINY = -87.33654475055310898657;
    const int row_wc = KS, row_ka = KS + V * vw, row_kb = row_ka + T;
    double cop_const = 0.0; // occu_cop: sum over unmasked visits of y log(dur) - lgamma(y + 1)
    // nmixture: B[t][n][site] = sum over unmasked visits of log C(n, y), -inf for n below the largest count
    // (nmixture.py:150-155, 186-190 and numpyro's BinomialProbs.log_prob); one spare row for pair reads
    const int Kn = max_abundance;
    std::vector<float> tab(model == 4 ? ((size_t)T * (Kn + 1) + 1) * n_stride : 0, 0.0f);
    for (int i = 0; i < N; i++) {
        bool site_nan = false;
        for (int k = 0; k < Ks; k++) {
            float x = site_covs[(size_t)i * Ks + k];
            if (std::isnan(x)) { site_nan = true; x = 0.0f; }
            rows[(size_t)k * n_stride + i] = x;
        }
        for (int t = 0; t < T && model == 4; t++) {
            std::vector<double> ys;
            for (int j = 0; j < J; j++) {
                const int v = t * J + j;
                const size_t o = ((size_t)i * T + t) * J + j;
                bool cov_nan = site_nan;
                const size_t r0 = (size_t)(row_wc + v * vw);
                for (int k = 0; k < Ko; k++) {
                    float x = obs_covs[o * Ko + k];
                    if (std::isnan(x)) { cov_nan = true; x = 0.0f; }
                    ds->h_wraw[((size_t)v * Ko + k) * n_stride + i] = x;
                    rows[(r0 + 2 + k) * n_stride + i] = x;
                }
                const float y = obs[o];
                if (cov_nan || !std::isfinite(y)) {
                    for (int k = 0; k < Ko; k++) rows[(r0 + 2 + k) * n_stride + i] = 0.0f;
                    continue;
                }
                rows[r0 * n_stride + i] = y;          // m y
                rows[(r0 + 1) * n_stride + i] = 1.0f; // m
                ys.push_back((double)y);
            }
            double ymax = 0.0;
            for (double y : ys) ymax = std::max(ymax, y);
            for (int n = 0; n <= Kn; n++) {
                double b = 0.0;
                if ((double)n < ymax) b = -INFINITY;
                else for (double y : ys) b += std::lgamma(n + 1.0) - std::lgamma(y + 1.0) - std::lgamma(n - y + 1.0);
                tab[((size_t)t * (Kn + 1) + n) * n_stride + i] = (float)b;
            }
            rows[(size_t)(row_ka + t) * n_stride + i] = (float)ymax;
            rows[(size_t)(row_kb + t) * n_stride + i] = (float)ys.size();
        }
        for (int t = 0; t < T && model == 3; t++) {
            // occu_cop.py:150-156,236-255: visit = (y_m, d_m, w_1..w_Ko), masked visits contribute nothing
            double ysum = 0.0, dsum = 0.0;
            for (int j = 0; j < J; j++) {
                const int v = t * J + j;
                const size_t o = ((size_t)i * T + t) * J + j;
                bool cov_nan = site_nan;
                const size_t r0 = (size_t)(row_wc + v * vw);
                for (int k = 0; k < Ko; k++) {
                    float x = obs_covs[o * Ko + k];
                    if (std::isnan(x)) { cov_nan = true; x = 0.0f; }
                    ds->h_wraw[((size_t)v * Ko + k) * n_stride + i] = x;
                    rows[(r0 + 2 + k) * n_stride + i] = x;
                }
                const float y = obs[o], dur = mo.session_duration[o];
                ds->h_dur[(size_t)v * n_stride + i] = dur;
                if (cov_nan || !std::isfinite(y)) {
                    for (int k = 0; k < Ko; k++) rows[(r0 + 2 + k) * n_stride + i] = 0.0f;
                    continue;
                }
                rows[r0 * n_stride + i] = y;
                rows[(r0 + 1) * n_stride + i] = dur;
                ysum += y; dsum += dur;
                cop_const += (y > 0.0f ? (double)y * std::log((double)dur) : 0.0) - std::lgamma((double)y + 1.0);
            }
            rows[(size_t)(row_ka + t) * n_stride + i] = (float)ysum;
            rows[(size_t)(row_kb + t) * n_stride + i] = (float)dsum;
        }
        for (int sp = 0; sp < S && model != 3 && model != 4; sp++)
        for (int t = 0; t < T; t++) {
            const size_t sp_off = (size_t)sp * sp_rows;     // this species' block of rows
            const float *obs_sp = obs + (size_t)sp * N * T * J;
            int n_masked = 0, n_det = 0;
            for (int j = 0; j < J; j++) {
                const int v = t * J + j;
                const size_t o = ((size_t)i * T + t) * J + j;
                bool cov_nan = site_nan;
                float w[BL_MAX_COVS];
                for (int k = 0; k < Ko; k++) {
                    float x = obs_covs[o * Ko + k];
                    if (std::isnan(x)) { cov_nan = true; x = 0.0f; }
                    w[k] = x;
                    ds->h_wraw[((size_t)v * Ko + k) * n_stride + i] = x;
                }
                const float y = obs_sp[o];
                float c = 0.0f;
                if (cov_nan || !std::isfinite(y)) n_masked++;
                else if (y != 0.0f) { c = 1.0f; n_det++; }
                else c = -1.0f;
                const size_t r0 = sp_off + (size_t)(row_wc + v * vw);
                rows[r0 * n_stride + i] = c;
                for (int k = 0; k < Ko; k++) rows[(r0 + 1 + k) * n_stride + i] = c * w[k];
            }
            rows[(sp_off + (size_t)(row_ka + t)) * n_stride + i] = (float)(n_masked * LN2);
            rows[(sp_off + (size_t)(row_kb + t)) * n_stride + i] = (float)(n_det * LOG_TINY);
        }
    }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void **)&ds->d_rows, rows.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(ds->d_rows, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && model == 4) e = hipMalloc((void **)&ds->d_tab, tab.size() * sizeof(float));
    if (e == hipSuccess && model == 4) e = hipMemcpy(ds->d_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ds->h_abort, 64, hipHostMallocMapped);
    if (e == hipSuccess) { *ds->h_abort = 0; e = hipHostGetDevicePointer((void **)&ds->d_abort, ds->h_abort, 0); }
    if (e == hipSuccess) e = hipEventCreate(&ds->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ds->ev1);
    if (e != hipSuccess) {
        const int rc = bl_fail(BL_ERR_NO_DEVICE, "dataset upload failed: %s", hipGetErrorString(e));
        bl_dataset_destroy(ds);
        return rc;
    }
    BlDevData &dd = ds->dd;
    dd.rows = ds->d_rows; dd.n_sites = N; dd.n_stride = n_stride; dd.T = T; dd.J = J;
    dd.Ks = Ks; dd.Ko = Ko; dd.KS = KS; dd.KO = KO;
    dd.loc_b = (float)pb.loc; dd.isc2_b = (float)(1.0 / (pb.scale * pb.scale));
    dd.loc_a = (float)pa.loc; dd.isc2_a = (float)(1.0 / (pa.scale * pa.scale));
    dd.prior_const = S * ((Ks + 1) * std::log(pb.scale) + (Ko + 1) * std::log(pa.scale) + (Ks + Ko + 2) * 0.91893853320467274178);
    dd.n_species = S;
    dd.dyn = model == 8 ? 1 : 0;
    if (model == 8) dd.prior_const = 3 * (Ks + 1) * std::log(pb.scale) + (Ko + 1) * std::log(pa.scale) + ds->D * 0.91893853320467274178;
    dd.has_fp = has_extra ? model : 0; dd.fp_a = (float)mo.fp_a; dd.fp_b = (float)mo.fp_b;
    if (model == 3) {
        dd.prior_const -= cop_const;                        // the parameter-free part of the Poisson log-pmf
        if (has_extra) dd.prior_const -= std::log(mo.fp_a); // phi = log f, f ~ Exponential(r): energy r e^phi - phi - log r
    }
    if (model == 2) // phi = logit f, f ~ Beta(a, b): the energy a softplus(-phi) + b softplus(phi) carries + log B(a, b)
        dd.prior_const += std::lgamma(mo.fp_a) + std::lgamma(mo.fp_b) - std::lgamma(mo.fp_a + mo.fp_b);
    *out = ds;
    return BL_OK;
}

extern "C" int bl_dataset_set_prior_family(bl_dataset *ds, int family_beta, int family_alpha)
{
    if (!ds) return bl_fail(BL_ERR_INVALID, "NULL argument");
    if ((family_beta != BL_PRIOR_NORMAL && family_beta != BL_PRIOR_LAPLACE) || (family_alpha != BL_PRIOR_NORMAL && family_alpha != BL_PRIOR_LAPLACE))
        return bl_fail(BL_ERR_INVALID, "prior family must be BL_PRIOR_NORMAL or BL_PRIOR_LAPLACE");
    if (ds->in_flight) return bl_fail(BL_ERR_BUSY, "a NUTS launch is in flight on this handle");
    BlDevData &dd = ds->dd;
    // constant of the potential: log(scale) + log(2 pi) / 2 per Normal coefficient, log(2 scale) per Laplace one
    const double HL2PI = 0.91893853320467274178;
    auto konst = [&](int fam, double scale) { return fam == BL_PRIOR_LAPLACE ? std::log(2.0 * scale) : std::log(scale) + HL2PI; };
    const double before = ds->nsp * ((ds->Ks + 1) * konst(ds->fam_b, ds->pb.scale) + (ds->Ko + 1) * konst(ds->fam_a, ds->pa.scale));
    const double after = ds->nsp * ((ds->Ks + 1) * konst(family_beta, ds->pb.scale) + (ds->Ko + 1) * konst(family_alpha, ds->pa.scale));
    dd.prior_const += after - before;
    ds->fam_b = family_beta; ds->fam_a = family_alpha;
    dd.isc2_b = family_beta == BL_PRIOR_LAPLACE ? 0.0f : (float)(1.0 / (ds->pb.scale * ds->pb.scale));
    dd.l1_b = family_beta == BL_PRIOR_LAPLACE ? (float)(1.0 / ds->pb.scale) : 0.0f;
    dd.isc2_a = family_alpha == BL_PRIOR_LAPLACE ? 0.0f : (float)(1.0 / (ds->pa.scale * ds->pa.scale));
    dd.l1_a = family_alpha == BL_PRIOR_LAPLACE ? (float)(1.0 / ds->pa.scale) : 0.0f;
    if (ds->model == 6) {
        BlReModel &m = ds->re;
        m.u_const += after - before;
        m.isc2_b = dd.isc2_b; m.l1_b = dd.l1_b; m.isc2_a = dd.isc2_a; m.l1_a = dd.l1_a;
    }
    return BL_OK;
}

extern "C" int bl_dataset_destroy(bl_dataset *ds)
{
    if (!ds) return BL_OK;
    hipSetDevice(ds->device);
    if (ds->in_flight) { *ds->h_abort = 1; hipStreamSynchronize(ds->stream); }
    if (ds->d_rows) hipFree(ds->d_rows);
    if (ds->d_wraw) hipFree(ds->d_wraw);
    if (ds->d_tab) hipFree(ds->d_tab);
    if (ds->d_dur) hipFree(ds->d_dur);
    if (ds->d_run) hipFree(ds->d_run);
    if (ds->d_xchg) hipFree(ds->d_xchg);
    if (ds->d_restate) hipFree(ds->d_restate);
    if (ds->d_scores) hipFree(ds->d_scores);
    if (ds->h_abort) hipHostFree(ds->h_abort);
    if (ds->ev0) hipEventDestroy(ds->ev0);
    if (ds->ev1) hipEventDestroy(ds->ev1);
    delete ds;
    return BL_OK;
}

extern "C" int bl_dataset_param_dim(const bl_dataset *ds, int *D)
{
    if (!ds || !D) return bl_fail(BL_ERR_INVALID, "NULL argument");
    *D = ds->D;
    return BL_OK;
}

// Workgroups per chain / LDS staging decision.  One site per thread is the latency optimum
// (DESIGN.md "geometry"); fall back to several sites per thread, then to un-staged HBM rows.
// Dynamic occupancy: lanes that share one site pair (dyn_device.hpp) -- as many (a power of two, <= 8, <= the seasons) as the lanes of
// the workgroups one XCD offers a chain allow at one pair per lane group.
static int dyn_lanes_per_pair(const bl_dataset *ds, int chains)
{
    const int per_xcd = ((chains > 0 ? chains : 1) + 7) / 8;
    const int kmax = std::max(1, 32 / per_xcd);
    const long long npairs = (ds->dims.n_sites + 1) / 2;
    int G = 1;
    while (2 * G <= 8 && 2 * G <= ds->dims.n_periods) G *= 2;
    while (G > 1 && npairs * G > (long long)kmax * 4 * 64) G >>= 1; // (round 4: up to four compute waves per workgroup)
    if (const char *e = bl_env(BL_ENV_DYN_G)) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) G = v; } // A/B knob
    return G;
}

// Plain occupancy model and its false-positive form: lanes that share one site pair (occu_device.hpp: bl_eval_sites_grp), as the code
// log2(period lanes) | log2(visit lanes) << 4; 0 = one pair per lane.  More lanes per pair while a lane would still walk more than
// BL_GRP_VISITS visits of its pair and the lanes one XCD offers the chain (4 compute waves on each of its <= kmax workgroups) allow it;
// the periods are split first (nothing to exchange between those lanes), then the visits of a period (one DPP fold per period).
#define BL_GRP_VISITS 6
#ifndef BL_WIDE_KMAX
#define BL_WIDE_KMAX 128   // workgroups of a chain in the wide geometry when lane groups ask for more than the slice needs (measured: tools/time_wide.py)
#endif
static int occu_lane_group(const bl_dataset *ds, int chains, int want_k, int kcap = 0)
{
    if (ds->model != 0 && ds->model != 2 && ds->model != 3 && ds->model != 4) return 0; // (occu, false positives, occu_cop, nmixture)
    const int T = ds->dims.n_periods, J = ds->dims.n_replicates;
    const long long V = (long long)T * J, npairs = (ds->dims.n_sites + 1) / 2;
    const int per_xcd = ((chains > 0 ? chains : 1) + 7) / 8;
    int kmax = std::max(1, 32 / per_xcd);
    if (kcap > 0) kmax = kcap; // (wide geometry: the chain's workgroups span XCDs -- the lanes of up to 256 / chains workgroups)
    if (want_k > 0) kmax = std::min(kmax, want_k);
    int target = BL_GRP_VISITS;
    if (const char *e = bl_env(BL_ENV_GRP_VISITS)) { const int v = atoi(e); if (v >= 1) target = v; } // A/B knob
    int lg = 0;
    while (lg < 4 && ((V + (1 << lg) - 1) >> lg) > target && (npairs << (lg + 1)) <= (long long)kmax * 4 * 64) lg++;
    if (const char *e = bl_env(BL_ENV_OCCU_G)) { // tests / A/B: force the lanes per pair (1, 2, 4, 8, 16)
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) { lg = 0; while ((1 << lg) < v) lg++; }
    }
    int lgt = 0;
    while (lgt < lg && (2 << lgt) <= T) lgt++;
    if (const char *e = bl_env(BL_ENV_OCCU_GT)) { const int v = atoi(e); if (v >= 0 && v <= lg) lgt = v; } // tests: log2 of the period lanes
    int lgj = lg - lgt;
    while (lgj > 0 && (1 << (lgj - 1)) >= J) lgj--; // (visit lanes beyond a period's visits would only idle)
    return lgt | (lgj << 4);
}

// Small problems (plain model and false positives): the whole chain on ONE workgroup of BL_CWAVES_SINGLE compute waves -- no exchange
// through L2 at all (nuts_kernel.hpp: the k == 1 path), the site pairs shared among as many lanes as the 448 offer.  Taken when a lane
// is then left with at most BL_SINGLE_VISITS visits of its pair (400 x 16, 8 visits per lane: 2.00 us on one workgroup against 1.86 on five) and the records fit one CU's LDS.  Returns the lane-group code or -1.
#define BL_SINGLE_VISITS 7
static int occu_single_workgroup(const bl_dataset *ds, int want_k)
{
    if ((ds->model != 0 && ds->model != 2) || want_k > 0 || ds->nsp > 1) return -1; // (several species: the partial table is sized for 4 waves)
    if (const char *e = bl_env(BL_ENV_SINGLE)) { if (e[0] == '0') return -1; } // A/B knob
    const int T = ds->dims.n_periods, J = ds->dims.n_replicates;
    const long long V = (long long)T * J, npairs = (ds->dims.n_sites + 1) / 2, lanes = BL_CWAVES_SINGLE * 64;
    if (npairs > lanes) return -1;
    int lg = 0;
    while (lg < 4 && (npairs << (lg + 1)) <= lanes && ((V + (1 << lg) - 1) >> lg) > 1) lg++;
    if (((V + (1 << lg) - 1) >> lg) > BL_SINGLE_VISITS) return -1;
    if (const char *e = bl_env(BL_ENV_OCCU_G)) { // tests / A/B: force the lanes per pair
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) { lg = 0; while ((1 << lg) < v) lg++; }
    }
    int lgt = 0;
    while (lgt < lg && (2 << lgt) <= T) lgt++;
    if (const char *e = bl_env(BL_ENV_OCCU_GT)) { const int v = atoi(e); if (v >= 0 && v <= lg) lgt = v; }
    int lgj = lg - lgt;
    while (lgj > 0 && (1 << (lgj - 1)) >= J) lgj--;
    return lgt | (lgj << 4);
}

static void choose_geometry(const bl_dataset *ds, int chains, int want_k, int *k_out, int *nloc_out, int *ld_out,
                            int *lds_bytes_out, int *staged_out, int *ncw_out, int *wide_out, int *grp_out = nullptr)
{
    const int N = ds->dims.n_sites;
    // chains are dealt to the 8 XCDs (32 CUs each); a chain's k workgroups share one XCD, one per CU
    const int per_xcd = ((chains > 0 ? chains : 1) + 7) / 8;
    int kmax = 32 / per_xcd;
    if (kmax < 1) kmax = 1;
    // occu / false positives: a lane evaluates a PAIR of sites (packed f32 math); one pair per lane is the
    // latency optimum.  With 3 compute waves every wave, the control wave included, owns a SIMD; that is
    // used whenever one pair per lane still fits the chain's workgroup budget, else 4 compute waves.
    // occu_rn (rn_device.hpp): 7 compute waves; its cost is per item of the sums over N, not per lane, so a chain takes
    // all the workgroups its XCD offers once it has more than a wave of sites for each.
    int ncw = BL_CWAVES_RN, per_wg = 64, grp = 0;
    if (ds->model != 1) {
        grp = occu_lane_group(ds, chains, want_k);
        const int G = 1 << ((grp & 15) + (grp >> 4)); // lanes per site pair
        ncw = (((long long)N * G + 2 * 3 * 64 - 1) / (2 * 3 * 64) <= kmax) ? 3 : 4;
        // A/B knob: only the counts this library was built with (3, 4, and BL_OCCU_CWX in a variant build of the plain model); anything
        // else is ignored rather than turned into a launch that no instantiation serves
        if (const char *e = bl_env(BL_ENV_CWAVES)) {
            const int v = atoi(e);
            bool built = v == 3 || v == 4;
#ifdef BL_OCCU_CWX
            built = built || (ds->model == 0 && v == BL_OCCU_CWX);
#endif
            if (built) ncw = v;
        }
        per_wg = std::max(2, 2 * ncw * 64 / G);
        if (ds->model == 8) { // one site pair per lane group; three compute waves while the chain's workgroups offer the lanes, else four
            const int Gd = dyn_lanes_per_pair(ds, chains);
            ncw = (((long long)N * Gd + 2 * 3 * 64 - 1) / (2 * 3 * 64) <= kmax) ? 3 : 4;
            if (const char *e = bl_env(BL_ENV_CWAVES)) { const int v = atoi(e); if (v == 3 || v == 4) ncw = v; }
            per_wg = std::max(2, 2 * ncw * 64 / Gd);
        }
    }
    int k = want_k > 0 ? want_k : (N + per_wg - 1) / per_wg;
    if (k > kmax) k = kmax;
    if (k < 1) k = 1;
    const int single_grp = occu_single_workgroup(ds, want_k);
    if (single_grp >= 0 && (long long)((N + 1) / 2) * bl_record_stride(ds->dims.n_periods, ds->dims.n_replicates, ds->KS, ds->ko_layout) * 4 * ds->nsp
                               <= BL_LDS_TOTAL - BL_OFF_DATA) {
        k = 1; ncw = BL_CWAVES_SINGLE; grp = single_grp; // one workgroup per chain: no exchange
    }
    // behind the records: occu_rn's tables; the dynamic model's lane-private columns (sized for 4 compute waves: ncw is settled below)
    const int rn_scratch = ds->model == 1 ? bl_rn_scratch_bytes(BL_CWAVES_RN) : (ds->model == 8 ? bl_dyn_scratch_bytes(ds->dims.n_periods, 4) : 0);
    const int lds_cap = BL_LDS_TOTAL - BL_OFF_DATA - rn_scratch;
    // LDS keeps one record of `stride` floats per PAIR of sites (occu_device.hpp)
    const int stride = bl_record_stride(ds->dims.n_periods, ds->dims.n_replicates, ds->KS, ds->ko_layout);
    auto fits = [&](int kk, int *nloc) { // (every species has a record region of its own)
        *nloc = (N + kk - 1) / kk;
        return (long long)((*nloc + 1) / 2) * stride * 4 * ds->nsp <= lds_cap;
    };
    int nloc;
    bool ok = fits(k, &nloc);
    // (an explicit count that does not fit is raised too for the models that have no HBM-row form -- fit()'s retry at k/2 after an engine
    // timeout must not turn into a refusal -- but kept for the plain model, whose HBM-row form serves any count)
    if (!ok && (want_k <= 0 || ds->model != 0))
        for (int kk = k + 1; kk <= kmax && !ok; kk++) { ok = fits(kk, &nloc); if (ok) k = kk; }
    // Wide geometry: the slice does not fit the LDS of one XCD's workgroups, but it does when the chain takes more
    // CUs than one XCD has.  The exchange then crosses XCDs (the kernel's placement census makes it take the fabric
    // form), which costs a few thousand cycles per tick -- little next to re-reading the rows from HBM every tick.
    int wide = 0;
    if (!ok && want_k <= 0) {
        const char *e = bl_env(BL_ENV_NO_WIDE);
        const int kwide = 256 / (chains > 0 ? chains : 1);
        if (!(e && e[0] == '1'))
            for (int kk = kmax + 1; kk <= kwide && !ok; kk++) { ok = fits(kk, &nloc); if (ok) { k = kk; wide = 1; } }
    }
    if (ok && !wide && ds->model == 1 && want_k <= 0) { // A/B knob: occu_rn over more workgroups than one XCD offers a chain
        if (const char *e = bl_env(BL_ENV_RN_K)) {
            const int kk = atoi(e), kwide = 256 / (chains > 0 ? chains : 1);
            int nl;
            if (kk > kmax && kk <= kwide && fits(kk, &nl)) { k = kk; nloc = nl; wide = 1; }
        }
    }
    if (ds->model == 0 && want_k <= 0) { // A/B knob: the plain model over kk workgroups across XCDs although one XCD's would do
        if (const char *e = bl_env(BL_ENV_WIDE_K)) {
            const int kk = atoi(e), kwide = 256 / (chains > 0 ? chains : 1);
            int nl;
            if (kk > 0 && kk <= kwide && fits(kk, &nl)) { k = kk; nloc = nl; wide = 1; ok = true; }
        }
    }
    // Wide geometry and many visits per site (the top rows of the reference's benchmark grid: 6 400 x 64, 12 800 x 90): the lanes of
    // one XCD no longer cap the lane groups -- a site pair is shared by as many lanes as BL_GRP_VISITS asks for and the chain takes
    // the workgroups that offer them (four compute waves each), up to 256 / chains.  More workgroups make the exchange longer (every
    // workgroup gathers every record over the fabric); BL_WIDE_KMAX bounds them where that costs more than the shorter visit loops save.
    if (wide && ok && (ds->model == 0 || ds->model == 2) && want_k <= 0 && !bl_env(BL_ENV_WIDE_K)) {
        const int kwide = std::min(256 / (chains > 0 ? chains : 1), BL_WIDE_KMAX);
        if (kwide > k) {
            const int g2 = occu_lane_group(ds, chains, 0, kwide);
            const int G2 = 1 << ((g2 & 15) + (g2 >> 4));
            int kk = (int)((((long long)N + 1) / 2 * G2 + 4 * 64 - 1) / (4 * 64));
            kk = std::min(std::max(kk, k), kwide);
            int nl;
            if (fits(kk, &nl)) { k = kk; nloc = nl; grp = g2; }
        }
    } else if (wide && ok && (ds->model == 0 || ds->model == 2) && want_k <= 0) {
        grp = occu_lane_group(ds, chains, 0, k); // (forced k: the groups its lanes allow)
    }
    if (!ok) fits(k, &nloc);
    if (!ok && ds->model == 0) ncw = 4; // HBM-row form is built for 4 compute waves only
    if (wide) ncw = ds->model == 1 ? BL_CWAVES_RN : 4; // full slices: all four SIMDs evaluate
    if (!ok) grp = 0; // (the HBM-row form keeps one pair per lane)
    if (grp_out) *grp_out = grp;
    *k_out = k; *nloc_out = nloc; *ld_out = stride; *staged_out = ok ? 1 : 0; *ncw_out = ncw; *wide_out = wide;
    *lds_bytes_out = ok ? BL_OFF_DATA + ((nloc + 1) / 2) * stride * 4 * ds->nsp + rn_scratch : BL_OFF_DATA;
}

// nmixture: the columns of the table B[t][n][site] that belong to a workgroup's sites go to LDS behind the records when they fit
// (T (K + 1) rows of 2 ceil(nloc / 2) floats); returns 1 and grows *lds_bytes if so.  BIOLITH_HIP_NMIX_LDS=0 keeps them in HBM / L2 (A/B).
static int nmix_table_in_lds(const bl_dataset *ds, int staged, int nloc, int *lds_bytes)
{
    if (ds->model != 4 || !staged) return 0;
    if (const char *e = bl_env(BL_ENV_NMIX_LDS)) { if (e[0] == '0') return 0; }
    const long long tb = (long long)ds->dims.n_periods * (ds->max_abundance + 1) * (2 * ((nloc + 1) / 2)) * 4;
    if (*lds_bytes + tb > BL_LDS_TOTAL) return 0;
    *lds_bytes += (int)tb;
    return 1;
}

// ------------------------------------------------------------- K1 logp ----
__global__ void bl_logp_final_kernel(BlDevData dd, int k, const double *theta, const double *partial, double *U, double *grad)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int Dsp = dd.Ks + dd.Ko + 2, nsp = dd.n_species > 0 ? dd.n_species : 1;
    const int D = dd.dyn ? 3 * (dd.Ks + 1) + dd.Ko + 1 : nsp * Dsp + (dd.has_fp ? 1 : 0);
    const int lj = lane < nsp * Dsp ? lane % Dsp : lane; // index within the lane's species
    double acc = 0.0;
    if (lane <= D)
        for (int m = 0; m < k; m++) acc += partial[((size_t)b * k + m) * 64 + lane];
    double pr = 0.0;
    if (lane < D && dd.has_fp == 3 && lane == D - 1) {
        // phi = log f, f ~ Exponential(rate r), Jacobian included: r e^phi - phi  (log r is in prior_const)
        const double phi = theta[(size_t)b * D + lane], rf = dd.fp_a * exp(phi);
        pr = rf - phi;
        grad[(size_t)b * D + lane] = -acc + rf - 1.0;
    } else if (lane < D && dd.has_fp && lane == D - 1) {
        // phi = logit f, f ~ Beta(a, b), Jacobian included: a softplus(-phi) + b softplus(phi)
        const double phi = theta[(size_t)b * D + lane], l = log1p(exp(-fabs(phi)));
        const double sig = 1.0 / (1.0 + exp(-phi));
        pr = dd.fp_a * (fmax(-phi, 0.0) + l) + dd.fp_b * (fmax(phi, 0.0) + l);
        grad[(size_t)b * D + lane] = -acc + (dd.fp_a + dd.fp_b) * sig - dd.fp_a;
    } else if (lane < D) {
        const bool is_b = dd.dyn ? lane < 3 * (dd.Ks + 1) : lj <= dd.Ks;
        const double loc = is_b ? dd.loc_b : dd.loc_a, isc2 = is_b ? dd.isc2_b : dd.isc2_a, l1 = is_b ? dd.l1_b : dd.l1_a;
        const double dth = theta[(size_t)b * D + lane] - loc;
        pr = 0.5 * dth * dth * isc2 + fabs(dth) * l1; // Normal or Laplace (one of isc2, l1 is 0)
        grad[(size_t)b * D + lane] = -acc + dth * isc2 + ((dth > 0.0) - (dth < 0.0)) * l1;
    }
    const double prior = bl_wave_sum_d(pr);
    if (lane == D) U[b] = -acc + prior + dd.prior_const;
}

// Launch of the random-effects / occu_cs NUTS kernel instantiated for this run's capacity, model kind and LDS residency.
template <int MK, int KIND, bool LROWS, int LT>
static hipError_t re_nuts_launch_inst(const BlReRun &run, int grid, size_t lds, hipStream_t st)
{
    if (lds) {
        const hipError_t e = hipFuncSetAttribute((const void *)bl_re_nuts_kernel<MK, KIND, LROWS, LT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((bl_re_nuts_kernel<MK, KIND, LROWS, LT>), dim3(grid), dim3(BL_RE_NT), lds, st, run);
    snprintf(g_kernel_name, sizeof g_kernel_name, "bl_re_nuts_kernel<%d, %d, %s, %d, 0>", MK, KIND, LROWS ? "true" : "false", LT); // (as rocprofv3 prints it)
    return hipGetLastError();
}
// the bench form's instantiations with the model's effects as compile-time facts (re_kernel.hpp: EFF)
template <int EFF>
static hipError_t re_nuts_launch_eff(const BlReRun &run, int grid, size_t lds, hipStream_t st)
{
    if (lds) {
        const hipError_t e = hipFuncSetAttribute((const void *)bl_re_nuts_kernel<4, 0, true, 2, EFF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((bl_re_nuts_kernel<4, 0, true, 2, EFF>), dim3(grid), dim3(BL_RE_NT), lds, st, run);
    snprintf(g_kernel_name, sizeof g_kernel_name, "bl_re_nuts_kernel<4, 0, true, 2, %d>", EFF);
    return hipGetLastError();
}
template <int MK, int KIND>
static hipError_t re_nuts_dispatch_lds(const BlReRun &run, int grid, size_t lds, hipStream_t st)
{
    const int lt = run.m.lds_hot;
    if constexpr (MK == 4 && KIND == 0) {
        const char *e = bl_env(BL_ENV_RE_EFF); // A/B knob: 0 = the general kernel
        if (run.m.lds_rows && lt == 2 && run.m.n_species == 1 && !(e && e[0] == '0')) {
            // one period as a fact too, for site effects alone (same trajectories, one box: 8.91 -> 8.85 us on the bench shape; with
            // observation effects the same fact measured 0.1-2 % slower on four shapes -- profiles/r04/i_ab_re_t1_all.txt -- and is not
            // instantiated); knob 1: without it
            const int eff = (run.m.site_re ? 1 : 0) | (run.m.obs_re ? 2 : 0);
            if (eff == 1) return run.m.T == 1 && !(e && e[0] == '1') ? re_nuts_launch_eff<5>(run, grid, lds, st) : re_nuts_launch_eff<1>(run, grid, lds, st);
            if (eff == 2) return re_nuts_launch_eff<2>(run, grid, lds, st);
            if (eff == 3) return re_nuts_launch_eff<3>(run, grid, lds, st);
        }
    }
    if (run.m.lds_rows)
        return lt == 2 ? re_nuts_launch_inst<MK, KIND, true, 2>(run, grid, lds, st)
                       : (lt == 1 ? re_nuts_launch_inst<MK, KIND, true, 1>(run, grid, lds, st) : re_nuts_launch_inst<MK, KIND, true, 0>(run, grid, lds, st));
    return lt == 2 ? re_nuts_launch_inst<MK, KIND, false, 2>(run, grid, lds, st)
                   : (lt == 1 ? re_nuts_launch_inst<MK, KIND, false, 1>(run, grid, lds, st) : re_nuts_launch_inst<MK, KIND, false, 0>(run, grid, lds, st));
}
static hipError_t re_nuts_dispatch(int mk, const BlReRun &run, int grid, size_t lds, hipStream_t st)
{
    if (run.m.kind == 2) return mk == 4 ? re_nuts_dispatch_lds<4, 2>(run, grid, lds, st) : re_nuts_dispatch_lds<16, 2>(run, grid, lds, st);
    if (run.m.kind == 3) return mk == 4 ? re_nuts_dispatch_lds<4, 3>(run, grid, lds, st) : re_nuts_dispatch_lds<16, 3>(run, grid, lds, st);
    if (run.m.kind == 4) return mk == 4 ? re_nuts_dispatch_lds<4, 4>(run, grid, lds, st) : re_nuts_dispatch_lds<16, 4>(run, grid, lds, st);
    if (run.m.kind == 5) return mk == 4 ? re_nuts_dispatch_lds<4, 5>(run, grid, lds, st) : re_nuts_dispatch_lds<16, 5>(run, grid, lds, st);
    if (run.m.kind == 6) return mk == 4 ? re_nuts_dispatch_lds<4, 6>(run, grid, lds, st) : re_nuts_dispatch_lds<16, 6>(run, grid, lds, st);
    if (run.m.kind == 7) return mk == 4 ? re_nuts_dispatch_lds<4, 7>(run, grid, lds, st) : re_nuts_dispatch_lds<16, 7>(run, grid, lds, st);
    if (mk == 4) return run.m.kind == 1 ? re_nuts_dispatch_lds<4, 1>(run, grid, lds, st) : re_nuts_dispatch_lds<4, 0>(run, grid, lds, st);
    return run.m.kind == 1 ? re_nuts_dispatch_lds<16, 1>(run, grid, lds, st) : re_nuts_dispatch_lds<16, 0>(run, grid, lds, st);
}

// Geometry of the random-effects kernels for slices of `nloc` sites: threads sharing a site's visits, what lives in LDS.
// with_hot: also the sampler's hot (or hot and warm) vectors (NUTS only).  Returns the dynamic LDS bytes.
static size_t re_geometry(BlReModel &m, int nloc, int with_hot, int dl_max)
{
    int tps = 1;
    while (tps < 64 && 2 * tps * nloc <= BL_RE_NT && 2 * tps <= m.J) tps *= 2; // spare threads share a site's visits
    m.tps = tps;
    // two classes of waves (re_kernel.hpp: BlReSiteMap): more than four waves' worth of sites in one round -> the first four waves
    // full, the remaining sites on the other four waves with as many threads per site as fit
    m.w_a = BL_RE_NW; m.n_a = nloc; m.tps_b = tps;
    {
        const int lanes = nloc * tps, half = BL_RE_NT / 2;
        if (lanes > half && lanes <= BL_RE_NT) {
            const int n_a = half / tps, rest = nloc - n_a;
            int tb = tps;
            while (tb < 64 && 2 * tb * rest <= half && 2 * tb <= m.J) tb *= 2;
            const char *e = bl_env(BL_ENV_RE_NO_SPLIT); // (measurement)
            if (tb > tps && !(e && e[0] == '1')) { m.w_a = BL_RE_NW / 2; m.n_a = n_a; m.tps_b = tb; }
        }
    }
    // a workgroup's own LDS copy of its rows, and of the five vectors every leapfrog touches, when they fit (160 KB per CU,
    // one workgroup per CU; a few KB go to the reduction scratch)
    // (160 KB per workgroup; the kernels' static arrays -- reduction scratch, the exchange's [32][NRED] staging, per-species sums -- take up
    // to 10.3 KB at capacity 16)
    const size_t row_bytes = (size_t)m.n_rows * nloc * 4, budget = (size_t)148 * 1024;
    const size_t hot_bytes = (size_t)RE_HOT * dl_max * 4, warm_bytes = (size_t)RE_WARM * dl_max * 4;
    m.lds_rows = row_bytes <= budget ? 1 : 0;
    if (const char *e = bl_env(BL_ENV_RE_LDS_ROWS)) m.lds_rows = std::min(m.lds_rows, atoi(e)); // (tests: every instantiation)
    const size_t used = m.lds_rows ? row_bytes : 0;
    m.lds_hot = !with_hot ? 0 : (used + warm_bytes <= budget ? 2 : (used + hot_bytes <= budget ? 1 : 0));
    if (const char *e = bl_env(BL_ENV_RE_LDS_TIER)) m.lds_hot = std::min(m.lds_hot, atoi(e)); // (measurement)
    return used + (m.lds_hot == 2 ? warm_bytes : (m.lds_hot == 1 ? hot_bytes : 0));
}

static int create_re_impl(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                          int site_random_effects, int obs_random_effects, double prior_site_re_sd_scale,
                          double prior_obs_re_sd_scale, int fp_mode, double fp_a, double fp_b, int count_model, int count_K, const bl_normal_prior *prior_beta,
                          const bl_normal_prior *prior_alpha, int device, bl_dataset **out, const float *session_duration = nullptr);

// occu_cop(site_random_effects / obs_random_effects = True [, false_positives_constant / _unoccupied = True]): occu_cop.py:158-170,
// 183-186, 204-210, 229-248.  theta = [beta, alpha, (phi = log rate_fp), (log sds), site_re_occ [N], site_re_det [N], obs_re [N][T][J]];
// `counts` / `session_duration` / fp_mode / prior_fp_rate as bl_dataset_create_cop; one species.
extern "C" int bl_dataset_create_cop_re(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *counts,
                                        const float *session_duration, int fp_mode, double prior_fp_rate, int site_random_effects,
                                        int obs_random_effects, double prior_site_re_sd_scale, double prior_obs_re_sd_scale,
                                        const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha, int device, bl_dataset **out)
{
    if (fp_mode != 0 && fp_mode != BL_FP_CONSTANT && fp_mode != BL_FP_UNOCCUPIED)
        return bl_fail(BL_ERR_INVALID, "fp_mode must be 0, BL_FP_CONSTANT or BL_FP_UNOCCUPIED");
    if (fp_mode && (!(prior_fp_rate > 0.0) || !std::isfinite(prior_fp_rate))) return bl_fail(BL_ERR_INVALID, "Exponential prior needs a finite rate > 0");
    if (!session_duration) return bl_fail(BL_ERR_INVALID, "session_duration is NULL");
    if (dims && dims->n_species != 1)
        return bl_fail(BL_ERR_UNSUPPORTED, "occu_cop with random effects: one species per dataset (n_species=%d)", dims->n_species);
    return create_re_impl(dims, site_covs, obs_covs, counts, site_random_effects, obs_random_effects, prior_site_re_sd_scale,
                          prior_obs_re_sd_scale, fp_mode, fp_mode ? prior_fp_rate : 1.0, 0.0, 3, 0, prior_beta, prior_alpha, device, out, session_duration);
}

extern "C" int bl_dataset_create_re(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                                    int site_random_effects, int obs_random_effects, double prior_site_re_sd_scale,
                                    double prior_obs_re_sd_scale, const bl_normal_prior *prior_beta,
                                    const bl_normal_prior *prior_alpha, int device, bl_dataset **out)
{
    return create_re_impl(dims, site_covs, obs_covs, obs, site_random_effects, obs_random_effects, prior_site_re_sd_scale,
                          prior_obs_re_sd_scale, 0, 0.0, 0.0, 0, 0, prior_beta, prior_alpha, device, out);
}

// occu_rn(false_positives_constant = True [, site_random_effects / obs_random_effects = True]): occu_rn.py:133-138, 214-221 --
// y ~ Bernoulli(1 - (1 - p)(1 - f)), f ~ Beta(a, b).  theta = [beta, alpha, phi = logit f, (log sds), (effects)]; one species.  The
// model runs on the random-effects kernels also WITHOUT effects (kind 5; the work-proportional kernel of the plain model has no such term).
extern "C" int bl_dataset_create_rn_fp(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                                       int max_abundance, int site_random_effects, int obs_random_effects,
                                       double prior_site_re_sd_scale, double prior_obs_re_sd_scale, const bl_beta_prior *prior_fp,
                                       const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha, int device, bl_dataset **out)
{
    if (max_abundance < 1 || max_abundance >= BL_RN_NB)
        return bl_fail(BL_ERR_UNSUPPORTED, "max_abundance=%d outside 1..%d", max_abundance, BL_RN_NB - 1);
    if (dims && dims->n_species != 1)
        return bl_fail(BL_ERR_UNSUPPORTED, "Royle-Nichols with a false-positive rate: one species per dataset (n_species=%d)", dims->n_species);
    const double a = prior_fp ? prior_fp->a : 2.0, b = prior_fp ? prior_fp->b : 5.0;
    if (!(a > 0.0) || !(b > 0.0) || !std::isfinite(a) || !std::isfinite(b)) return bl_fail(BL_ERR_INVALID, "Beta prior needs finite a, b > 0");
    return create_re_impl(dims, site_covs, obs_covs, obs, site_random_effects, obs_random_effects,
                          site_random_effects ? prior_site_re_sd_scale : 1.0, obs_random_effects ? prior_obs_re_sd_scale : 1.0,
                          BL_FP_CONSTANT, a, b, 1, max_abundance, prior_beta, prior_alpha, device, out);
}

// occu(site_random_effects / obs_random_effects = True, false_positives_constant / _unoccupied = True): occu.py:146-157 with
// :170-173, 191-196.  theta = [beta, alpha, phi = logit(rate), (log sds), (effects)]; one species.
extern "C" int bl_dataset_create_re_fp(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                                       int site_random_effects, int obs_random_effects, double prior_site_re_sd_scale,
                                       double prior_obs_re_sd_scale, int fp_mode, const bl_beta_prior *prior_fp,
                                       const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha, int device, bl_dataset **out)
{
    if (fp_mode != BL_FP_CONSTANT && fp_mode != BL_FP_UNOCCUPIED)
        return bl_fail(BL_ERR_INVALID, "fp_mode must be BL_FP_CONSTANT or BL_FP_UNOCCUPIED");
    const double a = prior_fp ? prior_fp->a : 2.0, b = prior_fp ? prior_fp->b : 5.0;
    if (!(a > 0.0) || !(b > 0.0) || !std::isfinite(a) || !std::isfinite(b)) return bl_fail(BL_ERR_INVALID, "Beta prior needs finite a, b > 0");
    if (dims && dims->n_species != 1)
        return bl_fail(BL_ERR_UNSUPPORTED, "random effects with a false-positive rate: one species per dataset (n_species=%d)", dims->n_species);
    return create_re_impl(dims, site_covs, obs_covs, obs, site_random_effects, obs_random_effects, prior_site_re_sd_scale,
                          prior_obs_re_sd_scale, fp_mode, a, b, 0, 0, prior_beta, prior_alpha, device, out);
}

// nmixture(site_random_effects / obs_random_effects = True): nmixture.py:139-141, 166-172, 199-214.  theta = [beta, alpha, (log sds),
// site_re_abu [N], site_re_det [N], obs_re [N][T][J]]; `counts` as bl_dataset_create_nmix; one species.
extern "C" int bl_dataset_create_nmix_re(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *counts,
                                         int max_abundance, int site_random_effects, int obs_random_effects,
                                         double prior_site_re_sd_scale, double prior_obs_re_sd_scale, const bl_normal_prior *prior_beta,
                                         const bl_normal_prior *prior_alpha, int device, bl_dataset **out)
{
    if (max_abundance < 1 || max_abundance >= BL_RN_NB)
        return bl_fail(BL_ERR_UNSUPPORTED, "max_abundance=%d outside 1..%d", max_abundance, BL_RN_NB - 1);
    if (dims && dims->n_species != 1)
        return bl_fail(BL_ERR_UNSUPPORTED, "N-mixture with random effects: one species per dataset (n_species=%d)", dims->n_species);
    return create_re_impl(dims, site_covs, obs_covs, counts, site_random_effects, obs_random_effects, prior_site_re_sd_scale,
                          prior_obs_re_sd_scale, 0, 0.0, 0.0, 4, max_abundance, prior_beta, prior_alpha, device, out);
}

// occu_rn(site_random_effects / obs_random_effects = True): occu_rn.py:151-154, 172-184, 199-212.  theta = [beta, alpha, (log sds),
// site_re_abu [N], site_re_det [N], obs_re [N][T][J]]; one species.
extern "C" int bl_dataset_create_rn_re(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                                       int max_abundance, int site_random_effects, int obs_random_effects,
                                       double prior_site_re_sd_scale, double prior_obs_re_sd_scale, const bl_normal_prior *prior_beta,
                                       const bl_normal_prior *prior_alpha, int device, bl_dataset **out)
{
    if (max_abundance < 1 || max_abundance >= BL_RN_NB)
        return bl_fail(BL_ERR_UNSUPPORTED, "max_abundance=%d outside 1..%d", max_abundance, BL_RN_NB - 1);
    if (dims && dims->n_species != 1)
        return bl_fail(BL_ERR_UNSUPPORTED, "Royle-Nichols with random effects: one species per dataset (n_species=%d)", dims->n_species);
    return create_re_impl(dims, site_covs, obs_covs, obs, site_random_effects, obs_random_effects, prior_site_re_sd_scale,
                          prior_obs_re_sd_scale, 0, 0.0, 0.0, 1, max_abundance, prior_beta, prior_alpha, device, out);
}

static int create_re_impl(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                          int site_random_effects, int obs_random_effects, double prior_site_re_sd_scale,
                          double prior_obs_re_sd_scale, int fp_mode, double fp_a, double fp_b, int count_model, int count_K, const bl_normal_prior *prior_beta,
                          const bl_normal_prior *prior_alpha, int device, bl_dataset **out, const float *session_duration /* occu_cop only */)
{
    bl_env_snapshot();
    if (!site_random_effects && !obs_random_effects && !(count_model == 1 && fp_mode))
        return bl_fail(BL_ERR_INVALID, "bl_dataset_create_re: neither random effect requested (use bl_dataset_create)");
    if (dims && (dims->n_site_covs > BL_RE_MAXK || dims->n_obs_covs > BL_RE_MAXK))
        return bl_fail(BL_ERR_UNSUPPORTED, "random-effects kernels are built for at most %d covariates per side (Ks=%d, Ko=%d)",
                       BL_RE_MAXK, dims->n_site_covs, dims->n_obs_covs);
    if ((site_random_effects && !(prior_site_re_sd_scale > 0.0)) || (obs_random_effects && !(prior_obs_re_sd_scale > 0.0)))
        return bl_fail(BL_ERR_INVALID, "HalfNormal prior needs scale > 0");
    if (dims && dims->n_species > BL_RE_SMAX)
        return bl_fail(BL_ERR_UNSUPPORTED, "random effects: at most %d species under one chain (n_species=%d)", BL_RE_SMAX, dims->n_species);
    // the rows of the plain model (mask, NaN -> 0, sign folding; one block of visit rows per species) are exactly what the
    // random-effects site pass reads
    // (N-mixture: the count model's rows -- visit = (m y, m, w..) -- and its table of log-binomial sums)
    ModelOpts mo; mo.vector_kernels = true;
    if (count_model) { mo.model = count_model; mo.max_abundance = count_K; } // 4: N-mixture, 1: Royle-Nichols, 3: occu_cop (no false positives)
    if (count_model == 3) { mo.fp_mode = fp_mode; mo.fp_a = fp_mode ? fp_a : 1.0; mo.session_duration = session_duration; } // (fp_a: the Exponential prior's rate)
    int rc = dataset_create_impl(mo, dims, site_covs, obs_covs, obs, prior_beta, prior_alpha, device, out);
    if (rc) return rc;
    bl_dataset *ds = *out;
    const int S = dims->n_species, N = dims->n_sites, T = dims->n_periods, J = dims->n_replicates, Ks = ds->Ks, Ko = ds->Ko;
    // draws: [species 0: beta, alpha | species 1: ... | log site_re_sd | log obs_re_sd | site_re_occ [S][N] | site_re_det [S][N] | obs_re [S][N][T][J]]
    const long long Dll = (long long)S * (Ks + Ko + 2) + (fp_mode ? 1 : 0) + (site_random_effects ? 1 + 2LL * S * N : 0) + (obs_random_effects ? 1 + (long long)S * N * T * J : 0);
    if (Dll > (1LL << 24)) { bl_dataset_destroy(ds); *out = nullptr; return bl_fail(BL_ERR_UNSUPPORTED, "%lld coordinates", Dll); }
    BlReModel &m = ds->re;
    m.rows = ds->dd.rows; m.n_sites = N; m.n_stride = ds->dd.n_stride; m.T = T; m.J = J; m.Ks = Ks; m.Ko = Ko; m.KS = ds->KS; m.KO = ds->KO;
    m.site_re = site_random_effects ? 1 : 0; m.obs_re = obs_random_effects ? 1 : 0;
    m.n_species = S; m.G0s = Ks + Ko + 2; m.sp = 0; m.cb = 0; m.rv0 = ds->KS; m.sp_rows = T * J * (ds->KO + 1 + (count_model ? 1 : 0)) + 2 * T;
    m.G0 = S * m.G0s; m.G = m.G0 + (fp_mode ? 1 : 0) + m.site_re + m.obs_re; m.D = (int)Dll;
    int at = m.G0;
    m.kind = count_model == 4 ? 3 : (count_model == 1 ? (fp_mode ? 5 : 4) : (count_model == 3 ? (fp_mode ? 7 : 6) : (fp_mode ? 2 : 0))); m.fp_mode = fp_mode; m.fp_a = (float)fp_a; m.fp_b = (float)fp_b;
    m.tab = ds->d_tab; m.tab_ld = ds->n_stride; m.max_abundance = count_K;
    m.o_fp = fp_mode ? at++ : -1;       // phi = logit(false-positive rate): right behind the regression coefficients
    m.o_phi_s = m.site_re ? at++ : -1;
    m.o_phi_o = m.obs_re ? at++ : -1;
    m.o_u = m.o_v = m.o_e = -1;
    if (m.site_re) { m.o_u = at; m.o_v = at + S * N; at += 2 * S * N; }
    if (m.obs_re) { m.o_e = at; at += S * N * T * J; }
    m.loc_b = ds->dd.loc_b; m.isc2_b = ds->dd.isc2_b; m.loc_a = ds->dd.loc_a; m.isc2_a = ds->dd.isc2_a;
    m.l1_b = 0.0f; m.l1_a = 0.0f;
    m.hn_is2_s = site_random_effects ? (float)(1.0 / (prior_site_re_sd_scale * prior_site_re_sd_scale)) : 0.0f;
    m.hn_is2_o = obs_random_effects ? (float)(1.0 / (prior_obs_re_sd_scale * prior_obs_re_sd_scale)) : 0.0f;
    const double HL2PI = 0.91893853320467274178, HN0 = 0.5 * std::log(2.0 / 3.14159265358979323846);
    m.u_const = ds->dd.prior_const;
    if (fp_mode && count_model != 3) m.u_const += std::lgamma(fp_a) + std::lgamma(fp_b) - std::lgamma(fp_a + fp_b); // + log B(a, b)  (occu_cop: -log rate is in prior_const)
    if (m.site_re) m.u_const += -HN0 + std::log(prior_site_re_sd_scale) + 2.0 * S * N * HL2PI;
    if (m.obs_re) m.u_const += -HN0 + std::log(prior_obs_re_sd_scale) + (double)S * N * T * J * HL2PI;
    m.n_total = N; m.s0 = 0; m.x_u = m.o_u; m.x_v = m.o_v; m.x_e = m.o_e;
    m.n_rows = ds->KS + m.sp_rows; // what ONE workgroup stages: the site covariates and its species' block
    re_geometry(m, N, 0, 0); // one workgroup over all sites (bl_logp_grad); bl_nuts_launch picks its own slices
    ds->model = 6; ds->D = m.D;
    return BL_OK;
}

// occu_cs (biolith/models/occu_cs.py): the continuous-score model runs on the random-effects kernels' framework (a chain over
// several workgroups, vectors in device memory) with four extra fixed coordinates and no effects.
extern "C" int bl_dataset_create_cs(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *scores,
                                    const double *prior_mu /*[4]: loc, scale of mu0; of mu1's base*/,
                                    const double *prior_sigma /*[4]: concentration, rate of sigma0; of sigma1*/,
                                    const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha, int device, bl_dataset **out)
{
    if (!dims || !scores || !prior_mu || !prior_sigma || !out) return bl_fail(BL_ERR_INVALID, "NULL argument");
    if (dims->n_site_covs > BL_RE_MAXK || dims->n_obs_covs > BL_RE_MAXK)
        return bl_fail(BL_ERR_UNSUPPORTED, "occu_cs kernels are built for at most %d covariates per side (Ks=%d, Ko=%d)", BL_RE_MAXK,
                       dims->n_site_covs, dims->n_obs_covs);
    if (!(prior_mu[1] > 0.0) || !(prior_mu[3] > 0.0) || !(prior_sigma[0] > 0.0) || !(prior_sigma[1] > 0.0) || !(prior_sigma[2] > 0.0) || !(prior_sigma[3] > 0.0))
        return bl_fail(BL_ERR_INVALID, "occu_cs priors need positive scales / concentrations / rates");
    const size_t n = (size_t)dims->n_species * dims->n_sites * dims->n_periods * dims->n_replicates;
    // the plain model's rows with "has a score" as the observation: c = 1 for a scored replicate, 0 for a masked one
    std::vector<float> has(n);
    for (size_t i = 0; i < n; i++) has[i] = std::isfinite(scores[i]) ? 1.0f : NAN;
    int rc = dataset_create_impl(ModelOpts{}, dims, site_covs, obs_covs, has.data(), prior_beta, prior_alpha, device, out);
    if (rc) return rc;
    bl_dataset *ds = *out;
    const int N = dims->n_sites, V = dims->n_periods * dims->n_replicates, Ks = ds->Ks, Ko = ds->Ko;
    std::vector<float> sc((size_t)V * ds->n_stride, 0.0f);
    for (int i = 0; i < N; i++)
        for (int v = 0; v < V; v++) {
            const float x = scores[(size_t)i * V + v];
            sc[(size_t)v * ds->n_stride + i] = std::isfinite(x) ? x : 0.0f;
        }
    if (hipMalloc((void **)&ds->d_scores, sc.size() * 4) != hipSuccess ||
        hipMemcpy(ds->d_scores, sc.data(), sc.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        bl_dataset_destroy(ds); *out = nullptr;
        return bl_fail(BL_ERR_NO_DEVICE, "occu_cs: score upload failed");
    }
    BlReModel &m = ds->re;
    m.rows = ds->dd.rows; m.n_sites = N; m.n_stride = ds->dd.n_stride; m.T = dims->n_periods; m.J = dims->n_replicates;
    m.Ks = Ks; m.Ko = Ko; m.KS = ds->KS; m.KO = ds->KO;
    m.site_re = 0; m.obs_re = 0; m.kind = 1; m.scores = ds->d_scores;
    m.G0 = Ks + Ko + 2; m.G = m.G0 + 4; m.D = m.G;
    m.n_species = 1; m.G0s = m.G0; m.sp = 0; m.cb = 0; m.rv0 = ds->KS; m.sp_rows = V * (ds->KO + 1) + 2 * dims->n_periods;
    m.o_phi_s = m.o_phi_o = m.o_u = m.o_v = m.o_e = -1;
    m.loc_b = ds->dd.loc_b; m.isc2_b = ds->dd.isc2_b; m.loc_a = ds->dd.loc_a; m.isc2_a = ds->dd.isc2_a; m.l1_b = 0.0f; m.l1_a = 0.0f;
    m.hn_is2_s = m.hn_is2_o = 0.0f;
    for (int k = 0; k < 4; k++) { m.cs_mu[k] = (float)prior_mu[k]; m.cs_sg[k] = (float)prior_sigma[k]; }
    // constants of the potential: the coefficients' (dd.prior_const), the two Normal normalisers of mu0 / mu1, the Gammas'
    const double HL2PI = 0.91893853320467274178;
    m.u_const = ds->dd.prior_const + std::log(prior_mu[1]) + std::log(prior_mu[3]) + 2.0 * HL2PI;
    for (int f = 0; f < 2; f++) m.u_const += -prior_sigma[2 * f] * std::log(prior_sigma[2 * f + 1]) + std::lgamma(prior_sigma[2 * f]);
    m.n_total = N; m.s0 = 0; m.x_u = m.x_v = m.x_e = -1;
    m.n_rows = ds->n_rows;
    re_geometry(m, N, 0, 0);
    ds->model = 6; ds->D = m.D;
    return BL_OK;
}

static int re_logp_grad(bl_dataset *ds, int B, const double *theta, double *U, double *grad)
{
    const size_t D = ds->D;
    std::vector<float> th32((size_t)B * D);
    for (size_t i = 0; i < th32.size(); i++) th32[i] = (float)theta[i];
    float *d_th = nullptr, *d_work = nullptr;
    double *d_U = nullptr, *d_grad = nullptr;
    DevScratch scratch;
    BL_HIP(scratch.alloc((void **)&d_th, th32.size() * 4));
    BL_HIP(scratch.alloc((void **)&d_work, (size_t)B * 2 * D * 4));
    BL_HIP(scratch.alloc((void **)&d_U, (size_t)B * 8));
    BL_HIP(scratch.alloc((void **)&d_grad, (size_t)B * D * 8));
    BL_HIP(hipMemcpy(d_th, th32.data(), th32.size() * 4, hipMemcpyHostToDevice));
    BlReModel m = ds->re;
    const size_t lds = re_geometry(m, m.n_sites, 0, 0);
    // capacity 4 (the common case: short register arrays) or 16 covariates per side
    if (m.Ks <= 4 && m.Ko <= 4) {
        if (lds) BL_HIP(hipFuncSetAttribute((const void *)bl_re_logp_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(bl_re_logp_kernel<4>, dim3(B), dim3(BL_RE_NT), lds, nullptr, m, B, d_th, d_work, d_U, d_grad);
    } else {
        if (lds) BL_HIP(hipFuncSetAttribute((const void *)bl_re_logp_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(bl_re_logp_kernel<16>, dim3(B), dim3(BL_RE_NT), lds, nullptr, m, B, d_th, d_work, d_U, d_grad);
    }
    BL_HIP(hipGetLastError());
    BL_HIP(hipMemcpy(U, d_U, (size_t)B * 8, hipMemcpyDeviceToHost));
    BL_HIP(hipMemcpy(grad, d_grad, (size_t)B * D * 8, hipMemcpyDeviceToHost));
    return BL_OK;
}

extern "C" int bl_logp_grad(bl_dataset *ds, int B, const double *theta, double *U, double *grad, int staged)
{
    if (!ds || !theta || !U || !grad || B <= 0) return bl_fail(BL_ERR_INVALID, "bl_logp_grad: bad argument");
    bl_env_snapshot();
    if (ds->in_flight) return bl_fail(BL_ERR_BUSY, "a NUTS launch is in flight on this handle");
    int rc = set_device(ds);
    if (rc) return rc;
    if (ds->model == 6) return re_logp_grad(ds, B, theta, U, grad);
    const int D = ds->D;
    int k, nloc, ld, lds_bytes, can_stage, ncw, wide, grp;
    choose_geometry(ds, 1, 0, &k, &nloc, &ld, &lds_bytes, &can_stage, &ncw, &wide, &grp);
    const int use_staged = staged && can_stage;
    if (!use_staged) { lds_bytes = BL_OFF_DATA; ncw = 4; grp = 0; }
    std::vector<float> th32((size_t)B * D);
    for (size_t i = 0; i < th32.size(); i++) th32[i] = (float)theta[i];
    float *d_th32 = nullptr;
    double *d_th = nullptr, *d_partial = nullptr, *d_U = nullptr, *d_grad = nullptr;
    DevScratch scratch;
    BL_HIP(scratch.alloc((void **)&d_th32, th32.size() * 4));
    BL_HIP(scratch.alloc((void **)&d_th, (size_t)B * D * 8));
    BL_HIP(scratch.alloc((void **)&d_partial, (size_t)B * k * 64 * 8));
    BL_HIP(scratch.alloc((void **)&d_U, (size_t)B * 8));
    BL_HIP(scratch.alloc((void **)&d_grad, (size_t)B * D * 8));
    BL_HIP(hipMemcpy(d_th32, th32.data(), th32.size() * 4, hipMemcpyHostToDevice));
    BL_HIP(hipMemcpy(d_th, theta, (size_t)B * D * 8, hipMemcpyHostToDevice));
    BL_HIP(hipMemset(d_partial, 0, (size_t)B * k * 64 * 8));
    BlLogpParams p{};
    p.dd = ds->dd; p.k = k; p.nloc = nloc; p.rec_stride = ld; p.B = B; p.theta = d_th32; p.partial = d_partial;
    p.max_abundance = ds->max_abundance;
    p.rn_off = BL_OFF_DATA + ((nloc + 1) / 2) * ld * 4; // (occu_rn: one species)
    p.lane_grp = ds->model == 8 ? dyn_lanes_per_pair(ds, 1) : grp;
    p.nmix_lds = nmix_table_in_lds(ds, use_staged, nloc, &lds_bytes);
    p.fp_mode = ds->fp_mode;
    p.nmix_tab = ds->d_tab;
    p.ncw = ncw;
    p.n_species = ds->nsp; p.sp_lds = ((nloc + 1) / 2) * ld;
    if (ds->nsp > 1 && !use_staged)
        return bl_fail(BL_ERR_UNSUPPORTED, "joint sampling of %d species needs the LDS-staged path (slice too large, or staged=0 requested)", ds->nsp);
    if (ds->model != 0 && !use_staged)
        return bl_fail(BL_ERR_UNSUPPORTED, "occu_rn / false-positive models need the LDS-staged path (slice too large, or staged=0 requested)");
    const int lrc = ds->kern->logp(&p, k, lds_bytes, use_staged, ds->model, nullptr);
    if (lrc != 0) return bl_fail(BL_ERR_NO_DEVICE, "logp kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
    hipLaunchKernelGGL(bl_logp_final_kernel, dim3(B), dim3(64), 0, nullptr, ds->dd, k, d_th, d_partial, d_U, d_grad);
    BL_HIP(hipGetLastError());
    BL_HIP(hipMemcpy(U, d_U, (size_t)B * 8, hipMemcpyDeviceToHost));
    BL_HIP(hipMemcpy(grad, d_grad, (size_t)B * D * 8, hipMemcpyDeviceToHost));
    return BL_OK;
}

// ------------------------------------------------------------------- NUTS ----
static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// Head of the run slab = the RESULT BLOCK of a launch: everything bl_nuts_fetch returns, contiguous, so that the gather
// over RCCL (comm_rccl.hpp) ships it as one buffer.  Offsets are a function of (chains, draws kept, D) only.
struct RunLayout { size_t draws, div, steps, acc, pot, eps, minv, nleap, end; };
static RunLayout result_layout(size_t C, size_t Sa, size_t D)
{
    RunLayout L{};
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
    L.draws = carve(C * Sa * D * 4); L.div = carve(C * Sa); L.steps = carve(C * Sa * 4); L.acc = carve(C * Sa * 4);
    L.pot = carve(C * Sa * 4); L.eps = carve(C * 4); L.minv = carve(C * D * 4); L.nleap = carve(C * 16);
    L.end = off;
    return L;
}

// Random-effects model: k workgroups per chain, each with a slice of the sites; sampler state in device memory / LDS
// (re_kernel.hpp).  Outputs land in the same run-slab fields as the other models', so poll / wait / fetch are shared.
static int re_nuts_launch(bl_dataset *ds, const bl_nuts_config *cfg, hipStream_t st, int max_depth)
{
    const int C = cfg->num_chains, S = cfg->num_samples, W = cfg->num_warmup;
    const size_t D = ds->D, Sa = S > 0 ? S : 1;
    const BlReModel &g = ds->re;
    const int N = g.n_sites, V = g.T * g.J, per_site = (g.site_re ? 2 : 0) + (g.obs_re ? V : 0);
    // workgroups per chain: about a thousand coordinates, or eight thousand visits, each; all C k of them must be resident
    // at once (one per CU: they spin on each other's sums)
    int k = cfg->wgs_per_chain;
    if (k <= 0) {
        const long long work = std::max<long long>((long long)D, (long long)N * V / 8);
        k = (int)((2 * work + BL_RE_NT - 1) / BL_RE_NT);
        if (const char *e = bl_env(BL_ENV_RE_WGS)) k = atoi(e);
    }
    int ncu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ds->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
    k = std::max(1, std::min(std::min(k, 32), std::min(ncu * 7 / 8 / C, N * g.n_species))); // (an eighth of the CUs spare: nobody may wait for a CU)
    // several species: the chain's workgroups are n_species groups of kps, each group slicing the sites of its species
    const int nsp = g.n_species;
    int kps = std::max(1, k / nsp);
    const int nloc = (N + kps - 1) / kps;
    kps = (N + nloc - 1) / nloc; // no empty slice
    k = nsp * kps;
    if (k > 32) return bl_fail(BL_ERR_UNSUPPORTED, "n_species=%d needs %d workgroups per chain (at most 32)", nsp, k);
    if (C * k > ncu) return bl_fail(BL_ERR_UNSUPPORTED, "num_chains=%d random-effects chains need %d resident workgroups (%d CUs)", C, C * k, ncu);
    const int dl_max = g.G + nloc * per_site;

    const RunLayout RL = result_layout((size_t)C, Sa, D);
    size_t off = RL.end;
    auto carve = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
    const size_t o_draws = RL.draws, o_div = RL.div, o_steps = RL.steps, o_acc = RL.acc, o_pot = RL.pot, o_eps = RL.eps,
                 o_minv = RL.minv, o_nleap = RL.nleap, o_status = carve(16),
                 o_rng = carve((size_t)C * k * (dl_max + 2) * 16), o_init = carve((size_t)C * D * 4), o_dbg = carve(BL_DBG_SLOTS * 8),
                 o_xchg = carve((size_t)C * 2 * k * BL_RE_NRED_MAX * 8), o_loc = carve((size_t)C * 4), o_run = carve(sizeof(BlReRun));
    if (off > ds->run_bytes) {
        if (ds->d_run) hipFree(ds->d_run);
        ds->d_run = nullptr; ds->run_bytes = 0;
        BL_HIP(hipMalloc(&ds->d_run, off));
        ds->run_bytes = off;
    }
    const size_t state_bytes = (size_t)C * k * RE_SLOTS * dl_max * 4;
    if (state_bytes > ds->restate_bytes) {
        if (ds->d_restate) hipFree(ds->d_restate);
        ds->d_restate = nullptr; ds->restate_bytes = 0;
        BL_HIP(hipMalloc((void **)&ds->d_restate, state_bytes));
        ds->restate_bytes = state_bytes;
    }
    char *base = (char *)ds->d_run;
    ds->d_draws = (float *)(base + o_draws); ds->d_div = (unsigned char *)(base + o_div); ds->d_steps = (int *)(base + o_steps);
    ds->d_acc = (float *)(base + o_acc); ds->d_pot = (float *)(base + o_pot); ds->d_eps = (float *)(base + o_eps);
    ds->d_minv = (float *)(base + o_minv); ds->d_nleap = (long long *)(base + o_nleap); ds->d_status = (int *)(base + o_status);
    ds->d_rng = (uint32_t *)(base + o_rng); ds->d_init = (float *)(base + o_init); ds->d_dbg = (long long *)(base + o_dbg);
    ds->d_loc = (int *)(base + o_loc);

    // RNG: one stream per coordinate (the model's order), then the scalar and the direction stream; chains are D + 2 streams
    // apart (>= 64).  Each workgroup gets the streams of ITS coordinates in its own order; the fixed effects' and the two
    // scalar streams are replicated (every workgroup advances its copy identically).
    // (With at most 61 coordinates -- occu_cs, occu_rn with a false-positive rate and no effects -- the layout is the small models':
    // chains 64 streams apart, the scalar stream 63 and the direction stream 62, as the oracle's orc_nuts_run lays them out.)
    const bool small_layout = D + 2 <= (size_t)BL_RNG_STREAMS_PER_CHAIN - 1;
    const int stride = small_layout ? BL_RNG_STREAMS_PER_CHAIN : (int)D + 2;
    const int ext_scalar = small_layout ? BL_RNG_STREAMS_PER_CHAIN - 1 : (int)D, ext_dir = small_layout ? BL_RNG_STREAMS_PER_CHAIN - 2 : (int)D + 1;
    std::vector<uint32_t> all((size_t)C * stride * 4), rs((size_t)C * k * (dl_max + 2) * 4, 0u);
    rng_streams_strided(cfg->seed, cfg->chain_offset, stride, C, all.data());
    for (int c = 0; c < C; c++)
        for (int w = 0; w < k; w++) {
            const int sp = w / kps, s0 = (w - sp * kps) * nloc, cnt = std::min(nloc, N - s0);
            uint32_t *dst = rs.data() + ((size_t)c * k + w) * (dl_max + 2) * 4;
            const uint32_t *src = all.data() + (size_t)c * stride * 4;
            auto put = [&](int local, int ext) { memcpy(dst + (size_t)local * 4, src + (size_t)ext * 4, 16); };
            int at = 0;
            for (; at < g.G; at++) put(at, at);
            if (g.site_re) {
                for (int i = 0; i < cnt; i++) put(at + i, g.o_u + sp * N + s0 + i);
                for (int i = 0; i < cnt; i++) put(at + cnt + i, g.o_v + sp * N + s0 + i);
                at += 2 * cnt;
            }
            if (g.obs_re)
                for (int v = 0; v < V; v++)
                    for (int i = 0; i < cnt; i++) put(at + v * cnt + i, g.o_e + (sp * N + s0 + i) * V + v);
            put(dl_max, ext_scalar);   // scalar stream
            put(dl_max + 1, ext_dir);  // direction stream
        }
    BL_HIP(hipMemcpyAsync(ds->d_rng, rs.data(), rs.size() * 4, hipMemcpyHostToDevice, st));
    std::vector<float> it32;
    if (cfg->init_theta) {
        it32.resize((size_t)C * D);
        for (size_t i = 0; i < it32.size(); i++) it32[i] = (float)cfg->init_theta[i];
        BL_HIP(hipMemcpyAsync(ds->d_init, it32.data(), it32.size() * 4, hipMemcpyHostToDevice, st));
    }
    BlReRun run{};
    run.m = g;
    const size_t lds = re_geometry(run.m, nloc, 1, dl_max);
    const bool cap4 = g.Ks <= 4 && g.Ko <= 4;
    run.num_chains = C; run.num_warmup = W; run.num_samples = S; run.max_depth = max_depth;
    run.k = k; run.nloc = nloc; run.dl_max = dl_max;
    run.xchg = (unsigned long long *)(base + o_xchg);
    { const char *e1 = bl_env(BL_ENV_NO_LOCAL); run.allow_local = (e1 && e1[0] == '1') ? 0 : 1; }
    run.xcd_local = ds->d_loc;
    run.target_accept = (float)(cfg->target_accept > 0.0 ? cfg->target_accept : 0.8);
    int32_t ws[32], we[32];
    run.nwin = adaptation_schedule(W, ws, we, 32);
    if (run.nwin > 32) return bl_fail(BL_ERR_INVALID, "adaptation schedule too long");
    for (int i = 0; i < 32; i++) run.win_end[i] = i < run.nwin ? we[i] : 0x7fffffff;
    run.state = ds->d_restate; run.rng = ds->d_rng;
    run.init_theta = cfg->init_theta ? ds->d_init : nullptr;
    run.abort_flag = ds->d_abort;
    run.draws = ds->d_draws; run.diverging = ds->d_div; run.num_steps = ds->d_steps; run.accept_prob = ds->d_acc;
    run.potential = ds->d_pot; run.step_size = ds->d_eps; run.inv_mass = ds->d_minv; run.nleap = ds->d_nleap; run.status = ds->d_status;
    run.dbg = ds->d_dbg;
    BlReRun *d_runp = (BlReRun *)(base + o_run);
    BL_HIP(hipMemcpyAsync(d_runp, &run, sizeof run, hipMemcpyHostToDevice, st));
    BL_HIP(hipStreamSynchronize(st));
    *ds->h_abort = 0;
    BL_HIP(hipEventRecord(ds->ev0, st));
    BL_HIP(hipMemsetAsync(ds->d_status, 0, 16, st));
    BL_HIP(hipMemsetAsync(ds->d_dbg, 0, BL_DBG_SLOTS * 8, st));
    BL_HIP(hipMemsetAsync(run.xchg, 0, (size_t)C * 2 * k * BL_RE_NRED_MAX * 8, st));
    BL_HIP(hipMemsetAsync(ds->d_loc, 0, (size_t)C * 4, st));
    // XCD-aware mapping in the kernel (surplus blocks exit at once)
    BL_HIP(re_nuts_dispatch(cap4 ? 4 : 16, run, 8 * k * ((C + 7) / 8), lds, st));
    snprintf(ds->kernel_name, sizeof ds->kernel_name, "%s", g_kernel_name);
    BL_HIP(hipEventRecord(ds->ev1, st));
    ds->stream = st; ds->in_flight = true; ds->have_run = true;
    ds->C = C; ds->S = S; ds->W = W; ds->k = k; ds->nloc = nloc; ds->lds_ld = 0; ds->lds_bytes = (int)lds; ds->staged = run.m.lds_rows | (run.m.lds_hot << 1); ds->nvp = 0; ds->ncw = BL_RE_NW; ds->lane_grp = 0;
    return BL_OK;
}

extern "C" int bl_nuts_launch(bl_dataset *ds, const bl_nuts_config *cfg, void *stream)
{
    if (!ds || !cfg) return bl_fail(BL_ERR_INVALID, "NULL argument");
    if (ds->in_flight) return bl_fail(BL_ERR_BUSY, "a NUTS launch is already in flight on this handle");
    bl_env_snapshot(); // the one read of the environment for this launch
    snprintf(ds->env_overrides, sizeof ds->env_overrides, "%s", bl_env_active().c_str());
    const int C = cfg->num_chains, S = cfg->num_samples, W = cfg->num_warmup, D = ds->D;
    if (C <= 0 || S < 0 || W < 0 || S + W <= 0) return bl_fail(BL_ERR_INVALID, "num_chains/num_samples/num_warmup out of range");
    if (C > 256) return bl_fail(BL_ERR_UNSUPPORTED, "num_chains=%d > 256 per device launch", C);
    const int max_depth = cfg->max_tree_depth > 0 ? cfg->max_tree_depth : 10;
    if (max_depth > BL_MAX_DEPTH) return bl_fail(BL_ERR_INVALID, "max_tree_depth > %d", BL_MAX_DEPTH);
    if (cfg->chain_offset < 0) return bl_fail(BL_ERR_INVALID, "chain_offset < 0");
    int rc = set_device(ds);
    if (rc) return rc;

    if (ds->model == 6) return re_nuts_launch(ds, cfg, (hipStream_t)stream, max_depth);

    int k, nloc, ld, lds_bytes, staged;
    int ncw, wide, grp;
    choose_geometry(ds, C, cfg->wgs_per_chain, &k, &nloc, &ld, &lds_bytes, &staged, &ncw, &wide, &grp);
    const int nvp = (D + 4 <= 16) ? 16 : (D + 4 <= 32 ? 32 : 64);

    // ---- (re)allocate run slab ----
    const size_t Sa = S > 0 ? S : 1;
    const RunLayout RL = result_layout((size_t)C, Sa, (size_t)D);
    size_t off = RL.end;
    auto carve = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
    const size_t o_draws = RL.draws, o_div = RL.div, o_steps = RL.steps, o_acc = RL.acc, o_pot = RL.pot, o_eps = RL.eps,
                 o_minv = RL.minv, o_nleap = RL.nleap, o_status = carve(16),
                 o_rng = carve((size_t)C * BL_RNG_STREAMS_PER_CHAIN * 16), o_init = carve((size_t)C * D * 4), o_dbg = carve(BL_DBG_SLOTS * 8), o_loc = carve((size_t)C * 4),
                 o_cold = carve(sizeof(BlNutsCold));
    if (off > ds->run_bytes) {
        if (ds->d_run) hipFree(ds->d_run);
        ds->d_run = nullptr; ds->run_bytes = 0;
        BL_HIP(hipMalloc(&ds->d_run, off));
        ds->run_bytes = off;
    }
    char *base = (char *)ds->d_run;
    ds->d_draws = (float *)(base + o_draws); ds->d_div = (unsigned char *)(base + o_div); ds->d_steps = (int *)(base + o_steps);
    ds->d_acc = (float *)(base + o_acc); ds->d_pot = (float *)(base + o_pot); ds->d_eps = (float *)(base + o_eps);
    ds->d_minv = (float *)(base + o_minv); ds->d_nleap = (long long *)(base + o_nleap); ds->d_status = (int *)(base + o_status);
    ds->d_rng = (uint32_t *)(base + o_rng); ds->d_init = (float *)(base + o_init); ds->d_dbg = (long long *)(base + o_dbg); ds->d_loc = (int *)(base + o_loc);
    int pitch = nvp; // granules between workgroup records
    { const char *e = bl_env(BL_ENV_PITCH); if (e && atoi(e) >= nvp && atoi(e) <= 4096) pitch = atoi(e); }
    const size_t xb = align256((size_t)C * BL_XCHG_SLOTS * k * pitch * 8);
    if (xb > ds->xchg_bytes) {
        if (ds->d_xchg) hipFree(ds->d_xchg);
        ds->d_xchg = nullptr; ds->xchg_bytes = 0;
        BL_HIP(hipMalloc((void **)&ds->d_xchg, xb));
        ds->xchg_bytes = xb;
    }

    hipStream_t st = (hipStream_t)stream;
    // ---- inputs: RNG streams (host-jumped), optional init ----
    std::vector<uint32_t> rs((size_t)C * BL_RNG_STREAMS_PER_CHAIN * 4);
    for (int c = 0; c < C; c++)
        rng_streams(cfg->seed, cfg->chain_offset + c, BL_RNG_STREAMS_PER_CHAIN, rs.data() + (size_t)c * BL_RNG_STREAMS_PER_CHAIN * 4);
    BL_HIP(hipMemcpyAsync(ds->d_rng, rs.data(), rs.size() * 4, hipMemcpyHostToDevice, st));
    if (cfg->init_theta) {
        std::vector<float> it32((size_t)C * D);
        for (size_t i = 0; i < it32.size(); i++) it32[i] = (float)cfg->init_theta[i];
        BL_HIP(hipMemcpyAsync(ds->d_init, it32.data(), it32.size() * 4, hipMemcpyHostToDevice, st));
    }
    // rarely-read constants + output pointers: one device-memory block
    BlNutsCold cold{};
    cold.num_warmup = W; cold.num_samples = S;
    cold.target_accept = (float)(cfg->target_accept > 0.0 ? cfg->target_accept : 0.8);
    int32_t ws[32], we[32];
    cold.nwin = adaptation_schedule(W, ws, we, 32);
    if (cold.nwin > 32) return bl_fail(BL_ERR_INVALID, "adaptation schedule too long");
    for (int i = 0; i < 32; i++) cold.win_end[i] = i < cold.nwin ? we[i] : 0x7fffffff;
    cold.loc_b = ds->dd.loc_b; cold.isc2_b = ds->dd.isc2_b; cold.loc_a = ds->dd.loc_a; cold.isc2_a = ds->dd.isc2_a;
    cold.l1_b = ds->dd.l1_b; cold.l1_a = ds->dd.l1_a;
    cold.prior_const = ds->dd.prior_const;
    cold.fp_a = (float)ds->fp_a; cold.fp_b = (float)ds->fp_b;
    cold.rng = ds->d_rng;
    cold.init_theta = cfg->init_theta ? ds->d_init : nullptr;
    cold.abort_flag = ds->d_abort;
    cold.draws = ds->d_draws; cold.diverging = ds->d_div; cold.num_steps = ds->d_steps; cold.accept_prob = ds->d_acc;
    cold.potential = ds->d_pot; cold.step_size = ds->d_eps; cold.inv_mass = ds->d_minv; cold.nleap = ds->d_nleap;
    cold.status = ds->d_status; cold.xcd_local = ds->d_loc; cold.dbg = ds->d_dbg;
    BlNutsCold *d_cold = (BlNutsCold *)(base + o_cold);
    BL_HIP(hipMemcpyAsync(d_cold, &cold, sizeof cold, hipMemcpyHostToDevice, st));
    BL_HIP(hipStreamSynchronize(st)); // staging buffers die with this scope
    *ds->h_abort = 0;

    BlNutsParams p{};
    p.rows = ds->dd.rows; p.n_sites = ds->dd.n_sites; p.n_stride = ds->dd.n_stride;
    p.T = ds->dd.T; p.J = ds->dd.J; p.Ks = ds->dd.Ks; p.Ko = ds->dd.Ko;
    p.num_chains = C;
    p.k = k; p.nloc = nloc; p.rec_stride = ld; p.nvp = nvp;
    p.max_depth = max_depth;
    p.max_abundance = ds->max_abundance;
    p.rn_off = BL_OFF_DATA + ((nloc + 1) / 2) * ld * 4; // (occu_rn: one species)
    p.lane_grp = ds->model == 8 ? dyn_lanes_per_pair(ds, C) : grp;
    p.nmix_lds = nmix_table_in_lds(ds, staged, nloc, &lds_bytes);
    // the GRP instantiation (lane groups; the exchange-free path of a one-workgroup chain) only where it is needed: the headline's kernel
    // stays the one-pair-per-lane form alone
    // (a chain of ONE workgroup always runs on BL_CWAVES_SINGLE compute waves: that kernel alone has the exchange-free path)
    p.grp_kernel = ((ds->model == 0 || ds->model == 2) && staged && (grp > 0 || ncw == BL_CWAVES_SINGLE)) ? 1 : 0;
    if (ncw == BL_CWAVES_SINGLE && k != 1) return bl_fail(BL_ERR_INVALID, "internal: %d compute waves are the one-workgroup form, k = %d", ncw, k);
    if (const char *e = bl_env(BL_ENV_GRP_KERNEL)) { if (e[0] == '1' && (ds->model == 0 || ds->model == 2) && staged) p.grp_kernel = 1; } // A/B knob: the GRP form although it is not needed
    p.fp_mode = ds->fp_mode;
    p.nmix_tab = ds->d_tab;
    p.ncw = ncw;
    p.n_species = ds->nsp; p.sp_lds = ((nloc + 1) / 2) * ld;
    if (ds->nsp > 1 && !staged)
        return bl_fail(BL_ERR_UNSUPPORTED, "joint sampling of %d species: the dataset slice does not fit the LDS-staged path", ds->nsp);
    p.wide = wide;
    p.xchg = ds->d_xchg;
    p.cold = d_cold;
    p.spin_limit = 5000000u; // microseconds (BIOLITH_HIP_SPIN_US overrides: tests of the bound)
    { const char *e = bl_env(BL_ENV_SPIN_US); if (e && atoi(e) > 0) p.spin_limit = (unsigned)atoi(e); }
    {   // developer knobs (A/B measurements): exchange form and poll spacing
        const char *e1 = bl_env(BL_ENV_NO_LOCAL), *e2 = bl_env(BL_ENV_POLL_SLEEP);
        p.allow_local = (e1 && e1[0] == '1') ? 0 : 1;
        p.poll_sleep = e2 ? atoi(e2) : 3;
        if (p.poll_sleep < 0 || p.poll_sleep > 127) p.poll_sleep = 3;
        const char *e3 = bl_env(BL_ENV_FIRST_DELAY);
        p.pitch = pitch;
        p.first_delay = e3 ? atoi(e3) : 0;
        if (p.first_delay < 0 || p.first_delay > 127) p.first_delay = 0;
    }

    // timed region: state re-init (guide G16 "re-initialise every call") + the persistent kernel
    BL_HIP(hipEventRecord(ds->ev0, st));
    BL_HIP(hipMemsetAsync(ds->d_xchg, 0, xb, st));
    BL_HIP(hipMemsetAsync(ds->d_status, 0, 16, st));
    BL_HIP(hipMemsetAsync(ds->d_dbg, 0, BL_DBG_SLOTS * 8, st));
    BL_HIP(hipMemsetAsync(ds->d_loc, 0, (size_t)C * 4, st));
    // XCD-aware mapping in the kernel (surplus blocks exit at once), or consecutive blocks per chain in the wide geometry
    const int grid = wide ? C * k : 8 * k * ((C + 7) / 8);
    if (ds->model != 0 && !staged)
        return bl_fail(BL_ERR_UNSUPPORTED, "occu_rn / false-positive model: the dataset slice does not fit the LDS-staged path");
    const int lrc = ds->kern->nuts(&p, grid, lds_bytes, staged, ds->model, st);
    if (lrc != 0) return bl_fail(BL_ERR_NO_DEVICE, "NUTS kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
    snprintf(ds->kernel_name, sizeof ds->kernel_name, "%s", g_kernel_name);
    BL_HIP(hipEventRecord(ds->ev1, st));
    ds->stream = st; ds->in_flight = true; ds->have_run = true;
    ds->C = C; ds->S = S; ds->W = W; ds->k = k; ds->nloc = nloc; ds->lds_ld = ld; ds->lds_bytes = lds_bytes; ds->staged = staged; ds->nvp = nvp; ds->ncw = ncw; ds->lane_grp = p.lane_grp;
    return BL_OK;
}

extern "C" int bl_nuts_poll(bl_dataset *ds, int *done)
{
    if (!ds || !done) return bl_fail(BL_ERR_INVALID, "NULL argument");
    if (!ds->in_flight) { *done = 1; return BL_OK; }
    hipSetDevice(ds->device);
    hipError_t e = hipEventQuery(ds->ev1);
    if (e == hipSuccess) { *done = 1; return BL_OK; }
    if (e == hipErrorNotReady) {
        (void)hipGetLastError(); // "not ready" is an answer, not an error: do not leave it behind as this thread's last error
        *done = 0;
        return BL_OK;
    }
    return bl_fail(BL_ERR_NO_DEVICE, "hipEventQuery: %s", hipGetErrorString(e));
}

extern "C" int bl_nuts_abort(bl_dataset *ds)
{
    if (!ds) return bl_fail(BL_ERR_INVALID, "NULL argument");
    if (ds->h_abort) __atomic_store_n(ds->h_abort, 1, __ATOMIC_SEQ_CST);
    return BL_OK;
}

extern "C" int bl_nuts_wait(bl_dataset *ds)
{
    if (!ds) return bl_fail(BL_ERR_INVALID, "NULL argument");
    if (!ds->have_run) return bl_fail(BL_ERR_INVALID, "no NUTS launch on this handle");
    int rc = set_device(ds);
    if (rc) return rc;
    if (ds->in_flight) {
        BL_HIP(hipEventSynchronize(ds->ev1));
        ds->in_flight = false;
    }
    int status = 0;
    BL_HIP(hipMemcpy(&status, ds->d_status, 4, hipMemcpyDeviceToHost));
    if (status == BL_ERR_TIMEOUT) return bl_fail(BL_ERR_TIMEOUT, "in-kernel exchange spin bound hit (a workgroup was not resident?)");
    if (status == BL_ERR_ABORTED) return bl_fail(BL_ERR_ABORTED, "sampling aborted on request");
    if (status != 0) return bl_fail(BL_ERR_NO_DEVICE, "kernel reported status %d", status);
    return BL_OK;
}

extern "C" int bl_nuts_fetch(bl_dataset *ds, bl_nuts_output *out)
{
    if (!ds || !out) return bl_fail(BL_ERR_INVALID, "NULL argument");
    if (!ds->have_run || ds->in_flight) return bl_fail(BL_ERR_BUSY, "no finished NUTS launch to fetch");
    int rc = set_device(ds);
    if (rc) return rc;
    const size_t C = ds->C, S = ds->S, D = ds->D;
    if (out->draws && S) BL_HIP(hipMemcpy(out->draws, ds->d_draws, C * S * D * 4, hipMemcpyDeviceToHost));
    if (out->diverging && S) BL_HIP(hipMemcpy(out->diverging, ds->d_div, C * S, hipMemcpyDeviceToHost));
    if (out->num_steps && S) BL_HIP(hipMemcpy(out->num_steps, ds->d_steps, C * S * 4, hipMemcpyDeviceToHost));
    if (out->accept_prob && S) BL_HIP(hipMemcpy(out->accept_prob, ds->d_acc, C * S * 4, hipMemcpyDeviceToHost));
    if (out->potential_energy && S) BL_HIP(hipMemcpy(out->potential_energy, ds->d_pot, C * S * 4, hipMemcpyDeviceToHost));
    if (out->step_size) BL_HIP(hipMemcpy(out->step_size, ds->d_eps, C * 4, hipMemcpyDeviceToHost));
    if (out->inv_mass) BL_HIP(hipMemcpy(out->inv_mass, ds->d_minv, C * D * 4, hipMemcpyDeviceToHost));
    if (out->n_leapfrog) BL_HIP(hipMemcpy(out->n_leapfrog, ds->d_nleap, C * 16, hipMemcpyDeviceToHost));
    return BL_OK;
}

extern "C" int bl_nuts_run(bl_dataset *ds, const bl_nuts_config *cfg, bl_nuts_output *out)
{
    int rc = bl_nuts_launch(ds, cfg, nullptr);
    if (rc) return rc;
    rc = bl_nuts_wait(ds);
    if (rc) return rc;
    return out ? bl_nuts_fetch(ds, out) : BL_OK;
}

extern "C" int bl_nuts_elapsed_ms(bl_dataset *ds, float *ms)
{
    if (!ds || !ms) return bl_fail(BL_ERR_INVALID, "NULL argument");
    if (!ds->have_run || ds->in_flight) return bl_fail(BL_ERR_BUSY, "no finished NUTS launch");
    BL_HIP(hipEventElapsedTime(ms, ds->ev0, ds->ev1));
    return BL_OK;
}

extern "C" int bl_nuts_device_draws(bl_dataset *ds, void **dev_ptr, size_t *bytes)
{
    if (!ds || !dev_ptr || !bytes) return bl_fail(BL_ERR_INVALID, "NULL argument");
    if (!ds->have_run) return bl_fail(BL_ERR_INVALID, "no NUTS launch on this handle");
    *dev_ptr = ds->d_draws;
    *bytes = (size_t)ds->C * ds->S * ds->D * 4;
    return BL_OK;
}

extern "C" int bl_nuts_debug_counters(bl_dataset *ds, int64_t *out, int n)
{
    if (!ds || !out || n <= 0 || n > BL_DBG_SLOTS) return bl_fail(BL_ERR_INVALID, "bad argument");
    if (!ds->have_run || ds->in_flight) return bl_fail(BL_ERR_BUSY, "no finished NUTS launch");
    BL_HIP(hipMemcpy(out, ds->d_dbg, (size_t)n * 8, hipMemcpyDeviceToHost));
    return BL_OK;
}

extern "C" int bl_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0) return bl_fail(BL_ERR_INVALID, "bl_host_alloc: NULL / empty");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bl_fail(BL_ERR_NO_DEVICE, "no HIP device visible");
    BL_HIP(hipHostMalloc(out, bytes, hipHostMallocPortable));
    return BL_OK;
}
extern "C" int bl_host_free(void *ptr)
{
    if (ptr) BL_HIP(hipHostFree(ptr));
    return BL_OK;
}

// (kernels_inst.hip: bl_launch hands over the name of the instantiation it launches)
extern "C" void bl_note_kernel_name(const char *name)
{
    snprintf(g_kernel_name, sizeof g_kernel_name, "%s", name ? name : "");
}

extern "C" int bl_nuts_kernel_name(bl_dataset *ds, char *buf, int n)
{
    if (!ds || !buf || n <= 0) return bl_fail(BL_ERR_INVALID, "bl_nuts_kernel_name: bad argument");
    if (!ds->have_run) return bl_fail(BL_ERR_INVALID, "no NUTS launch on this handle");
    snprintf(buf, (size_t)n, "%s", ds->kernel_name);
    return BL_OK;
}

extern "C" int bl_nuts_lane_group(bl_dataset *ds, int *period_lanes, int *visit_lanes)
{
    if (!ds) return bl_fail(BL_ERR_INVALID, "NULL argument");
    if (!ds->have_run) return bl_fail(BL_ERR_INVALID, "no NUTS launch on this handle");
    int gt = 1, gj = 1;
    if (ds->model == 8) gt = ds->lane_grp > 0 ? ds->lane_grp : 1;                                    // dynamic occupancy: the seasons
    else if (ds->model == 0 || ds->model == 2 || ds->model == 3 || ds->model == 4) { gt = 1 << (ds->lane_grp & 15); gj = 1 << (ds->lane_grp >> 4); }
    if (period_lanes) *period_lanes = gt;
    if (visit_lanes) *visit_lanes = gj;
    return BL_OK;
}

extern "C" int bl_nuts_env_overrides(bl_dataset *ds, char *buf, int n)
{
    if (!ds || !buf || n <= 0) return bl_fail(BL_ERR_INVALID, "NULL argument");
    if (!ds->have_run) return bl_fail(BL_ERR_INVALID, "no NUTS launch on this handle");
    snprintf(buf, (size_t)n, "%s", ds->env_overrides);
    return BL_OK;
}

extern "C" int bl_nuts_geometry(bl_dataset *ds, int *wgs_per_chain, int *threads_per_wg, int *lds_bytes, int *lds_staged,
                                int *chains_on_l2_local_exchange)
{
    if (!ds) return bl_fail(BL_ERR_INVALID, "NULL argument");
    if (!ds->have_run) return bl_fail(BL_ERR_INVALID, "no NUTS launch on this handle");
    if (wgs_per_chain) *wgs_per_chain = ds->k;
    if (threads_per_wg) *threads_per_wg = ds->model == 6 ? BL_RE_NT : 64 * (ds->ncw + 1);
    if (lds_bytes) *lds_bytes = ds->lds_bytes;
    if (lds_staged) *lds_staged = ds->staged;
    if (chains_on_l2_local_exchange) {
        *chains_on_l2_local_exchange = 0;
        if (!ds->in_flight && ds->d_loc) {
            std::vector<int> loc(ds->C, 0);
            BL_HIP(hipMemcpy(loc.data(), ds->d_loc, (size_t)ds->C * 4, hipMemcpyDeviceToHost));
            for (int v : loc) *chains_on_l2_local_exchange += v;
        }
    }
    return BL_OK;
}

// ------------------------------------------------- deterministic sites ----
// psi[n][t][i] = sigmoid(beta0 + x_i . beta)      (occu.py:198-207; constant over t)
// With random effects (model 6; offsets o_u / o_v / o_e into a draw, -1 = absent, external coordinate order) the site's
// occupancy effect joins eta, its detection effect and the replicate's effect join nu (occu.py:198-202, 221-228).
__global__ void bl_psi_kernel(const float *__restrict__ rows, int n_stride, int N, int T, int Ks, int D,
                              const float *__restrict__ draws, int n0, int n1, float *__restrict__ psi, int model, int o_u)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float x[BL_MAX_COVS];
    for (int k = 0; k < Ks; k++) x[k] = rows[(size_t)k * n_stride + i];
    for (int n = n0 + blockIdx.y; n < n1; n += gridDim.y) {
        const float *th = draws + (size_t)n * D;
        float eta = th[0];
        for (int k = 0; k < Ks; k++) eta = fmaf(x[k], th[k + 1], eta);
        if (o_u >= 0) eta += th[o_u + i];
        // occu: psi = sigmoid(eta) (occu.py:207); occu_rn: abundance = exp(eta) (occu_rn.py:192)
        const float v = (model == 1 || model == 4) ? __expf(eta) : 1.0f / (1.0f + __expf(-eta));
        for (int t = 0; t < T; t++) psi[((size_t)(n - n0) * T + t) * N + i] = v;
    }
}
// prob_detection[n][j][t][i] = sigmoid(alpha0 + w_itj . alpha)   (occu.py:221-228)
__global__ void bl_pdet_kernel(const float *__restrict__ wraw, int n_stride, int N, int T, int J, int Ks, int Ko, int D,
                               const float *__restrict__ draws, int n0, int n1, float *__restrict__ out, int model, int o_v, int o_e)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    for (int n = n0 + blockIdx.y; n < n1; n += gridDim.y) {
        const float *al = draws + (size_t)n * D + Ks + 1;
        const float *th = draws + (size_t)n * D;
        for (int t = 0; t < T; t++)
            for (int j = 0; j < J; j++) {
                const int v = t * J + j;
                float nu = al[0];
                for (int k = 0; k < Ko; k++) nu = fmaf(wraw[((size_t)v * Ko + k) * n_stride + i], al[k + 1], nu);
                if (o_v >= 0) nu += th[o_v + i];
                if (o_e >= 0) nu += th[o_e + (size_t)i * T * J + v];
                // occu / occu_rn: prob_detection = sigmoid(nu); occu_cop: rate_detection = exp(nu) (occu_cop.py:236-243)
                out[(((size_t)(n - n0) * J + j) * T + t) * N + i] = model == 3 ? __expf(nu) : 1.0f / (1.0f + __expf(-nu));
            }
    }
}

extern "C" int bl_deterministic(bl_dataset *ds, int n_draws, const float *draws, float *psi, float *prob_detection)
{
    if (!ds || !draws || n_draws <= 0) return bl_fail(BL_ERR_INVALID, "bl_deterministic: bad argument");
    if (ds->nsp > 1) return bl_fail(BL_ERR_UNSUPPORTED, "bl_deterministic: a joint-species handle samples; take the sites from one handle per species");
    if (ds->model == 8) return bl_fail(BL_ERR_UNSUPPORTED, "bl_deterministic: the dynamic occupancy model's sites (psi, gamma, eps) are formed by the caller from the draws");
    if (ds->in_flight) return bl_fail(BL_ERR_BUSY, "a NUTS launch is in flight on this handle");
    int rc = set_device(ds);
    if (rc) return rc;
    const int N = ds->dims.n_sites, T = ds->dims.n_periods, J = ds->dims.n_replicates, D = ds->D;
    float *d_draws = nullptr, *d_out = nullptr;
    DevScratch scratch;
    BL_HIP(scratch.alloc((void **)&d_draws, (size_t)n_draws * D * 4));
    BL_HIP(hipMemcpy(d_draws, draws, (size_t)n_draws * D * 4, hipMemcpyHostToDevice));
    // chunk the draws so the device staging buffer stays <= 256 MiB
    const size_t per_draw = (size_t)T * N * 4 * (prob_detection ? (size_t)J : 1);
    int chunk = (int)((256u << 20) / (per_draw ? per_draw : 1));
    if (chunk < 1) chunk = 1;
    if (chunk > n_draws) chunk = n_draws;
    BL_HIP(scratch.alloc((void **)&d_out, (size_t)chunk * per_draw));
    if (prob_detection && !ds->d_wraw) {
        BL_HIP(hipMalloc((void **)&ds->d_wraw, ds->h_wraw.size() * 4));
        BL_HIP(hipMemcpy(ds->d_wraw, ds->h_wraw.data(), ds->h_wraw.size() * 4, hipMemcpyHostToDevice));
    }
    const dim3 block(256);
    for (int n0 = 0; n0 < n_draws; n0 += chunk) {
        const int n1 = (n0 + chunk < n_draws) ? n0 + chunk : n_draws;
        const dim3 grid((N + 255) / 256, (n1 - n0) < 1024 ? (n1 - n0) : 1024);
        if (psi) {
            hipLaunchKernelGGL(bl_psi_kernel, grid, block, 0, nullptr, ds->d_rows, ds->n_stride, N, T, ds->Ks, D, d_draws, n0, n1, d_out,
                               ds->model == 6 && ds->re.kind >= 3 && ds->re.kind <= 5 ? 4 : ds->model /* N-mixture / Royle-Nichols with effects: abundance = exp(eta + u) */, ds->model == 6 ? ds->re.o_u : -1);
            BL_HIP(hipGetLastError());
            BL_HIP(hipMemcpy(psi + (size_t)n0 * T * N, d_out, (size_t)(n1 - n0) * T * N * 4, hipMemcpyDeviceToHost));
        }
        if (prob_detection) {
            hipLaunchKernelGGL(bl_pdet_kernel, grid, block, 0, nullptr, ds->d_wraw, ds->n_stride, N, T, J, ds->Ks, ds->Ko, D, d_draws, n0, n1, d_out,
                               ds->model == 6 && ds->re.kind >= 6 ? 3 : ds->model /* occu_cop with effects: rate_detection = exp(nu) */,
                               ds->model == 6 ? ds->re.o_v : -1, ds->model == 6 ? ds->re.o_e : -1);
            BL_HIP(hipGetLastError());
            BL_HIP(hipMemcpy(prob_detection + (size_t)n0 * J * T * N, d_out, (size_t)(n1 - n0) * J * T * N * 4, hipMemcpyDeviceToHost));
        }
    }
    return BL_OK;
}

// ------------------------------------------------------- multi-GPU: the gather of the draws over RCCL ----
#include "comm_rccl.hpp"
