// re_kernel.hpp -- occupancy model with site / observation random effects (biolith/models/occu.py:170-173, 191-196,
// 215-218): log-density + gradient, and NUTS over all D = Ks + Ko + 2 (+1 +1) + 2 N + N T J coordinates.
//
// Shape of the problem: D runs to thousands, so the lane-per-coordinate design of nuts_kernel.hpp does not apply.  Here a
// chain is k workgroups, each with a contiguous slice of the sites and those sites' effects; the fixed effects and log sds
// are replicated in every workgroup (same inputs, same arithmetic, same values).  Inside a workgroup every vector of the sampler (position, momentum, gradient, tree edges, proposals, checkpoints,
// mass matrix, Welford moments) is an array of length D in device memory (L2-resident: 40 D floats per chain), threads
// own coordinates d = tid, tid + NT, ...; scalars of the sampler are kept redundantly by every thread (all threads see the
// same reduced sums in LDS and take the same decisions), so nothing is broadcast.  Per leapfrog: two vector passes, one
// site pass, two block reductions -- each followed, when k > 1, by an exchange of the workgroups' partial sums through
// self-tagged 8-byte granules in device memory ({epoch, float}: the exchange of nuts_kernel.hpp, agent-scope atomics).  The arithmetic follows the oracle (oracle/occu_oracle.c: potential_grad_re,
// orc_nuts_run) statement by statement; RNG: one xoshiro stream per coordinate (external order) + the two scalar streams.
//
// The site data are the sign-folded rows the plain occu kernels read (occu_device.hpp: bl_eval_sites_hbm): a visit's
// record is (c, c w_1 .. c w_KO), c = +1 detection / -1 non-detection / 0 masked, so a random effect enters a visit's
// u = c nu as c (v_i + e_itj).
#pragma once
#include "nuts_kernel.hpp"

#define BL_RE_NT 512             // threads per workgroup (measured: 256 / 512 / 1024 -> 512 with 16-32 workgroups per chain)
#define BL_RE_NW (BL_RE_NT / 64)
#define BL_RE_MAXK 16            // covariates per side: the kernels are instantiated for capacities 4 and 16 (template parameter MK)
// Layout of the first block reduction for capacity MK: [0] log-lik, [1 .. MK+1] d/d beta, [MK+2 .. 2MK+2] d/d alpha, then from
// OX = 2MK+3: sum u^2+v^2 (occu_cs: d/d mu0), sum e^2 (d/d mu1), prior quadratic, abort request, XCC census (2), occu_cs d/d log sigma (2)
#define BL_RE_OA(MK) ((MK) + 2)
#define BL_RE_OX(MK) (2 * (MK) + 3)
#define BL_RE_NV1(MK) (2 * (MK) + 11)
#define BL_RE_NRED_OF(MK) (BL_RE_NV1(MK) > 26 ? BL_RE_NV1(MK) : 26) // widest block reduction (the second one: 3 + 2 x 10 checkpoints)
#define BL_RE_NRED_MAX BL_RE_NRED_OF(BL_RE_MAXK)
#define BL_RE_SMAX 8             // species under one chain (each needs at least one of the chain's <= 32 workgroups)
#define BL_RE_VB 4               // visits whose loads are issued together in the site pass

struct BlReModel {
    const float *rows;
    int n_sites, n_stride, T, J, Ks, Ko, KS, KO;
    int site_re, obs_re;
    int kind;                             // 0: occu with random effects; 1: occu_cs (continuous scores, no effects: D = G = G0 + 4);
                                          // 2: occu with random effects AND a false-positive rate (occu.py:146-157 with :170-173, 191-196)
    const float *scores;                  // kind 1: the replicates' scores, site-fastest [T J][n_stride] (0 where masked)
    float cs_mu[4], cs_sg[4];             // kind 1: Normal(loc, scale) of mu0 and of mu1's base; Gamma(concentration, rate) of sigma0, sigma1
    int G0, G, D;                         // fixed effects (of all species), + log sds, all coordinates
    // several species under one chain (occu.py:182-196: beta, alpha and the effects inside the species plate; 170-173: the sds outside
    // it, i.e. shared): a workgroup's slice belongs to ONE species -- its G0s coefficients start at cb, its visit rows at row rv0
    int n_species, G0s, sp, cb, rv0, sp_rows;
    int o_phi_s, o_phi_o, o_u, o_v, o_e;  // offsets (internal order = external order except inside the obs_re block)
    float loc_b, isc2_b, loc_a, isc2_a;   // Normal priors of beta / alpha (isc2 = 0 for a Laplace prior)
    float l1_b, l1_a;                     // Laplace priors: 1 / scale (0 for a Normal prior)
    float hn_is2_s, hn_is2_o;             // 1 / scale^2 of the HalfNormal priors of site_re_sd / obs_re_sd
    double u_const;                       // the constant part of the potential
    int tps;                              // threads that share one site in the site pass (power of two, <= 64)
    // two classes of waves in the site pass (BlReSiteMap): waves [0, w_a) take sites [0, n_a) with `tps` threads per site, waves
    // [w_a, NW) take the rest with tps_b (w_a = NW: every wave in the first class)
    int w_a, n_a, tps_b;
    // a workgroup works on a LOCAL copy of this struct: n_sites = its slice, rows advanced to its first site, D / o_* the
    // local layout; what refers to the whole dataset stays in the fields below
    int n_total;                          // sites of the dataset
    int s0;                               // first site of the slice
    int x_u, x_v, x_e;                    // external offsets of site_re_occ / site_re_det / obs_re in a draw
    int n_rows;                           // rows of the dataset (KS + T J (KO + 1) + 2 T)
    int lds_rows;                         // 1: every workgroup keeps its own copy of the rows in LDS (n_rows * n_sites floats)
    int lds_hot;                          // 1 / 2: the first RE_HOT / RE_WARM vectors of the chain live in LDS (after the rows), not in device memory
    // kind 2 only (kept at the end: the other kinds' kernels never load them)
    int fp_mode, o_fp;                    // 1 = the rate acts on every site ("constant"), 2 = on unoccupied sites; phi = logit(rate) at o_fp = G0
    float fp_a, fp_b;                     // Beta(a, b) prior of the rate
    // kinds 3, 4 and 5 (max_abundance; the table is kind 3's): N-mixture / Royle-Nichols with random effects / Royle-Nichols with a false-positive rate (nmixture.py:139-141, 166-172, 199-214) -- the count model's rows
    // (visit = (m y, m, w..)) and its data-only table tab[t][n][site] = sum_j m log C(n, y_j) (-inf below the largest count)
    const float *tab;
    int tab_ld, max_abundance;
};

// The dataset's rows for this workgroup: staged into dynamic LDS once when they fit, else read from device memory (L2).
__device__ __forceinline__ const float *bl_re_rows(const BlReModel &m, float *lds, int &ns)
{
    if (!m.lds_rows) { ns = m.n_stride; return m.rows; }
    const int N = m.n_sites;
    for (int r = 0; r < m.n_rows; r++)
        for (int i = threadIdx.x; i < N; i += BL_RE_NT) lds[r * N + i] = m.rows[(size_t)r * m.n_stride + i];
    __syncthreads();
    ns = N;
    return lds;
}

// slots of the per-chain state block, each D floats, ordered by how often a leapfrog touches them: the first RE_HOT, or the
// first RE_WARM, live in the workgroup's LDS when they fit (BlReModel::lds_hot = 1 / 2), the rest in device memory (L2)
enum {
    RE_CZ = 0, RE_CR, RE_CG,                       // leaf in flight: position, momentum (half step, then full), gradient
    RE_MINV, RE_SRSUM,                             // diagonal mass matrix; the subtree's momentum sum
    RE_HOT,
    RE_RSUM = RE_HOT, RE_RL, RE_RR,                // the tree's momentum sum; the momenta at the tree's edges
    RE_SZP, RE_SGP,                                // the subtree's proposal
    RE_CKR,                                        // BL_MAX_DEPTH checkpoints of r, then BL_MAX_DEPTH of the running sum
    RE_WARM = RE_CKR + 2 * BL_MAX_DEPTH,
    RE_TH = RE_WARM, RE_GR,                        // position / gradient the transition started from
    RE_ZL, RE_GL, RE_ZR, RE_GRR,                   // positions / gradients at the tree's edges
    RE_ZP, RE_GP,                                  // the tree's proposal
    RE_WFMEAN, RE_WFM2,                            // Welford moments
    RE_SLOTS
};

struct BlReRun {
    BlReModel m;
    int num_chains, num_warmup, num_samples, max_depth, nwin;
    int win_end[32];
    float target_accept;
    int k, nloc, dl_max;        // workgroups per chain, sites per workgroup, coordinates of the largest slice
    int allow_local;            // 0: always the placement-independent exchange
    int *xcd_local;             // [C] 1 if the chain ran on the L2-local exchange
    unsigned long long *xchg;   // [C][2][k][NRED of the kernel's capacity] exchange granules (k > 1)
    float *state;               // [C][k][RE_SLOTS][dl_max]
    uint32_t *rng;              // [C][k][dl_max + 2][4]: a workgroup's streams in ITS coordinate order, then scalar, direction
    const float *init_theta;    // [C][D] external order, or NULL
    const int *abort_flag;
    float *draws;               // [C][S][D] external order
    unsigned char *diverging; int *num_steps; float *accept_prob, *potential, *step_size, *inv_mass;
    long long *nleap; int *status;
    long long *dbg;             // [32] section cycle counters (BL_STAMPS diagnostic builds; chain 0, thread 0)
};
#ifdef BL_STAMPS
#define BL_RE_T(i) { const long long now_ = (long long)clock64(); st_acc[i] += now_ - st_prev; st_prev = now_; }
#else
#define BL_RE_T(i)
#endif

// local coordinate of a workgroup -> external (oracle / caller) coordinate.  Local order: fixed effects and log sds, then
// the slice's site_re_occ, site_re_det, then obs_re site-fastest; external order: the model's (obs_re replicate-fastest).
__device__ __forceinline__ int bl_re_ext(const BlReModel &m, int d)
{
    if (d < m.G) return d;
    const int cnt = m.n_sites;
    if (m.site_re && d < m.o_u + cnt) return m.x_u + m.s0 + (d - m.o_u);
    if (m.site_re && d < m.o_v + cnt) return m.x_v + m.s0 + (d - m.o_v);
    const int r = d - m.o_e, v = r / cnt, i = r - v * cnt;
    return m.x_e + (m.s0 + i) * (m.T * m.J) + v;
}

// The slice [s0, s0 + cnt) of species sp's sites as a model of its own (see BlReModel).
__device__ __forceinline__ BlReModel bl_re_slice(const BlReModel &g, int sp, int s0, int cnt)
{
    BlReModel m = g;
    m.rows = g.rows + s0; m.n_sites = cnt; m.s0 = s0;
    if (g.kind == 3) m.tab = g.tab + s0;
    m.sp = sp; m.cb = sp * g.G0s; m.rv0 = g.KS + sp * g.sp_rows;
    if (g.kind == 1) m.scores = g.scores + s0;
    int at = g.G;
    if (g.site_re) { m.o_u = at; m.o_v = at + cnt; at += 2 * cnt; m.x_u = g.x_u + sp * g.n_total; m.x_v = g.x_v + sp * g.n_total; }
    if (g.obs_re) { m.o_e = at; at += cnt * g.T * g.J; m.x_e = g.x_e + sp * g.n_total * g.T * g.J; }
    m.D = at;
    return m;
}

// Sum NV per-thread values over the workgroup: DPP wave sums, then a fixed-order f64 sum of the wave partials.
// out[] (LDS) is valid for every thread on return.
template <int M, int NV>
__device__ __forceinline__ void bl_re_wave_sums(const float (&v)[NV], float *dst, int n_used)
{
    float t[M];
#pragma unroll
    for (int k = 0; k < M; k++) t[k] = k < NV ? v[k < NV ? k : 0] : 0.0f;
    bl_wave_sum_vec_l63<M>(t); // M interleaved DPP chains, totals in lane 63
    if ((threadIdx.x & 63) == 63) {
#pragma unroll
        for (int k = 0; k < M; k++)
            if (k < n_used) dst[k] = t[k];
    }
}
template <int NV, int NRED>
__device__ __forceinline__ void bl_re_block_sum(float (&v)[NV], float *scr /*[NW][NRED]*/, double *out /*[NRED]*/, int n_used = NV,
                                                bool last_barrier = true)
{
    const int tid = threadIdx.x, wave = tid >> 6;
    float *dst = scr + wave * NRED;
    // (workgroup-uniform choice of the chain count: most leaves close few checkpoints)
    if (NV <= 5 || n_used <= 5) bl_re_wave_sums<(NV < 5 ? (NV < 3 ? 3 : NV) : 5)>(v, dst, n_used);
    else if (NV <= 9 || n_used <= 9) bl_re_wave_sums<(NV < 9 ? NV : 9)>(v, dst, n_used);
    else if (NV <= 17 || n_used <= 17) bl_re_wave_sums<(NV < 17 ? NV : 17)>(v, dst, n_used);
    else if (NV <= 26 || n_used <= 26) bl_re_wave_sums<(NV < 26 ? NV : 26)>(v, dst, n_used);
    else bl_re_wave_sums<NV>(v, dst, n_used);
    __syncthreads();
    if (tid < n_used) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < BL_RE_NW; w++) s += (double)scr[w * NRED + tid];
        out[tid] = s;
    }
    if (last_barrier) __syncthreads(); // (an exchange that follows only touches out[tid] from the thread that wrote it)
}

// Exchange between the k workgroups of a chain: each publishes its nv block sums as {epoch, float} granules, reads all
// k x nv of them and forms the totals in fixed order (f64) -- every workgroup gets bit-identical totals in out[].  Two
// parities of slots: a workgroup can be at most one exchange ahead of the slowest.  Returns false on the spin bound.
struct BlReXchg {
    unsigned long long *buf; // [C][2][k][NRED]
    int k, wg, chain;
    int n_species;           // > 1: workgroups [s kps, (s + 1) kps) belong to species s, kps = k / n_species
    unsigned epoch, spin_limit;
    bool local;              // all k workgroups proven to sit on one XCD: stores may stay in that XCD's L2 (nuts_kernel.hpp)
};
// The two halves can be called apart: what a workgroup does between them overlaps the hand-off.
template <int NRED>
__device__ __forceinline__ void bl_re_publish(BlReXchg &x, const double *out, int nv)
{
    if (x.k == 1) return;
    const int tid = threadIdx.x;
    x.epoch++;
    unsigned long long *base = x.buf + (((size_t)x.chain * 2 + (x.epoch & 1u)) * x.k) * NRED;
    if (tid < nv) {
        const unsigned long long gr = ((unsigned long long)x.epoch << 32) | (unsigned long long)__float_as_uint((float)out[tid]);
        if (x.local) // the line stays in this XCD's L2, where every consumer of this chain polls it
            __hip_atomic_store(base + (size_t)x.wg * NRED + tid, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else         // write-through: visible to any XCD
            __hip_atomic_store(base + (size_t)x.wg * NRED + tid, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// out_sp (several species, first exchange of a leapfrog only): [n_species][NRED], each species' sums over ITS workgroups.
template <int NRED>
__device__ __forceinline__ bool bl_re_collect(BlReXchg &x, double *out, float *scr2 /*[k][NRED]*/, int *lds_flag, int nv, double *out_sp = nullptr)
{
    if (x.k == 1) return true;
    const int tid = threadIdx.x;
    unsigned long long *base = x.buf + (((size_t)x.chain * 2 + (x.epoch & 1u)) * x.k) * NRED;
    for (int t = tid; t < x.k * nv; t += BL_RE_NT) {
        const int w = t / nv, v = t - w * nv;
        BlSpinBound bound;
        unsigned long long gr;
        while (true) {
            gr = __hip_atomic_load(base + (size_t)w * NRED + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(gr >> 32) == x.epoch) break;
            if (bound.expired(x.spin_limit)) { *lds_flag = 1; break; }
            if (!x.local) __builtin_amdgcn_s_sleep(1);
        }
        scr2[w * NRED + v] = __uint_as_float((unsigned)gr);
    }
    __syncthreads();
    if (tid < nv) {
        double t = 0.0;
        for (int w = 0; w < x.k; w++) t += (double)scr2[w * NRED + tid];
        out[tid] = t;
    } else if (out_sp && tid < nv * (1 + x.n_species)) {
        const int sp = tid / nv - 1, v = tid - (sp + 1) * nv, kps = x.k / x.n_species;
        double t = 0.0;
        for (int w = sp * kps; w < (sp + 1) * kps; w++) t += (double)scr2[w * NRED + v];
        out_sp[sp * NRED + v] = t;
    }
    __syncthreads();
    return *lds_flag == 0;
}
// One species (no per-species sums): the first wave alone collects, the way the main kernel's control wave does -- the 64 / nvp
// lane groups (nvp = 16, 32 or 64 value slots) each poll a share of the k workgroups' granules, up to eight loads in flight per
// lane, add them in f64 and are folded across groups by v_permlane16_swap / v_permlane32_swap; one barrier hands the totals to
// the other waves.  (The general form above stages k x nv floats in LDS between two barriers and adds them serially: 2.5-3.2 k
// cycles per exchange against ~1.7 k; it remains for the per-species sums of more than three species.)  Same fixed order in every
// workgroup => bit-identical totals.
// (bl_re_wave_poll: the first wave's lanes < nv get the sums over workgroups [w_lo, w_lo + w_cnt); out_sp: several species, first
// exchange of a leapfrog -- one poll per species' group of workgroups, the totals are their sums)
template <int NRED>
__device__ __forceinline__ double bl_re_wave_poll(const BlReXchg &x, int nv, int w_lo, int w_cnt, bool &timed_out)
{
    const int lane = threadIdx.x & 63;
    const int nvp = nv <= 16 ? 16 : (nv <= 32 ? 32 : 64), G = 64 / nvp;
    const int c = min(lane & (nvp - 1), nv - 1), sub = lane / nvp;
    const unsigned long long *base = x.buf + (((size_t)x.chain * 2 + (x.epoch & 1u)) * x.k + w_lo) * NRED + c;
    double acc = 0.0;
    for (int p0 = 0; p0 < w_cnt && !timed_out; p0 += 8 * G) {
        unsigned long long v[8];
        BlSpinBound bound;
        while (true) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int w = min(p0 + q * G + sub, w_cnt - 1); // (beyond the range: its last record again, dropped below)
                v[q] = __hip_atomic_load(base + (size_t)w * NRED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            unsigned bad = 0u;
#pragma unroll
            for (int q = 0; q < 8; q++) bad |= (unsigned)(v[q] >> 32) ^ x.epoch;
            if (__all(bad == 0u)) break;
            if (bound.expired(x.spin_limit)) { timed_out = true; break; }
            if (!x.local) __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) acc += (p0 + q * G + sub < w_cnt) ? (double)__uint_as_float((unsigned)v[q]) : 0.0;
    }
    if (nvp <= 16) acc = bl_fold_rows16_d(acc);
    if (nvp <= 32) acc = bl_fold_halves32_d(acc);
    return acc;
}
template <int NRED>
__device__ __forceinline__ bool bl_re_collect_wave(BlReXchg &x, double *out, int *lds_flag, int nv, double *out_sp = nullptr)
{
    if (x.k == 1) return true;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        bool timed_out = false;
        double tot = 0.0;
        if (out_sp) {
            const int kps = x.k / x.n_species;
            for (int sp = 0; sp < x.n_species; sp++) {
                const double t = bl_re_wave_poll<NRED>(x, nv, sp * kps, kps, timed_out);
                if (lane < nv) out_sp[sp * NRED + lane] = t;
                tot += t;
            }
        } else tot = bl_re_wave_poll<NRED>(x, nv, 0, x.k, timed_out);
        if (timed_out && lane == 0) *lds_flag = 1;
        if (lane < nv) out[lane] = tot;
    }
    __syncthreads();
    return *lds_flag == 0;
}
template <int NRED>
__device__ __forceinline__ bool bl_re_exchange(BlReXchg &x, double *out, float *scr2 /*[k][NRED]*/, int *lds_flag, int nv)
{
    bl_re_publish<NRED>(x, out, nv);
    return bl_re_collect_wave<NRED>(x, out, lds_flag, nv);
}

// Calls f(integral_constant<KB>, args...) for the smallest compiled covariate count KB in {1, 2, 4, 8, 16} (<= MK) that holds Ko.
template <int MK, class F, class... A>
__device__ __forceinline__ void bl_re_tiered(int Ko, F &f, A... args)
{
    if (Ko <= 1) f(std::integral_constant<int, 1>{}, args...);
    else if (Ko <= 2) f(std::integral_constant<int, 2>{}, args...);
    else if (MK <= 4 || Ko <= 4) f(std::integral_constant<int, (MK < 4 ? MK : 4)>{}, args...);
    else if (MK <= 8 || Ko <= 8) f(std::integral_constant<int, (MK < 8 ? MK : 8)>{}, args...);
    else f(std::integral_constant<int, MK>{}, args...);
}

// Which sites a thread works on in the site pass.  Threads of a site sit S = 64 / tps lanes apart in one wave: for a given visit the
// S neighbouring lanes read S neighbouring sites (one segment of a row), and the visits are pooled by xor-shuffles over the upper
// lane bits.  The pass is bound by VALU issue and a SIMD runs two of the workgroup's waves, so when the sites fill more than four
// waves but not eight, the first four waves take full loads and the remaining sites are spread over the other four with more
// threads per site (a shorter instruction stream): 313 sites x 10 visits: 4 x 64 sites with one thread each + 57 sites with four.
struct BlReSiteMap {
    int tps, S, sub, first, cnt, grp, ngrp, rounds;
};
__device__ __forceinline__ BlReSiteMap bl_re_site_map(const BlReModel &m)
{
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool second = wave >= m.w_a;
    BlReSiteMap q;
    q.tps = second ? m.tps_b : m.tps;
    q.S = 64 / q.tps;
    q.sub = lane / q.S;
    q.first = second ? m.n_a : 0;
    q.cnt = second ? m.n_sites - m.n_a : min(m.n_a, m.n_sites);
    const int wv = second ? wave - m.w_a : wave, nw = second ? BL_RE_NW - m.w_a : m.w_a;
    q.grp = wv * q.S + (lane & (q.S - 1));
    q.ngrp = nw * q.S;
    q.rounds = (q.cnt + q.ngrp - 1) / q.ngrp;
    return q;
}

// Site pass at position z: per-thread partials of the log-likelihood and of its gradient w.r.t. beta / alpha, and the
// random effects' full potential gradient written to g (the effect's own likelihood term + its Normal(0, sd) prior).
// part[0] = ll, part[1 .. 5] = d/d beta, part[6 .. 10] = d/d alpha.
// rows / ns: the dataset's rows in device memory (stride n_stride), or the workgroup's LDS copy of them (stride n_sites).
// FP (kind 2): the visits' terms gain the false-positive rate as in bl_eval_sites_fp (occu_device.hpp) -- a detection costs
// log sigma(u) + log(1 + f1 e^-u), a non-detection log sigma(u) + log(1 - f1), the z = 0 branch n_det log f + n_nondet log(1 - f) --
// and gphi accumulates d ll / d phi (chain rule through f = sigmoid(phi) included).
template <int MK, bool FP = false>
__device__ __forceinline__ void bl_re_site_pass(const BlReModel &m, const float *__restrict__ rows, int ns, int rv /* first visit row in rows */,
                                                const float *__restrict__ z, float *__restrict__ g, float (&part)[2 * MK + 3], float *gphi = nullptr)
{
    BlFpScalars fp{};
    if constexpr (FP) fp = bl_fp_scalars(z[m.o_fp], m.fp_mode == 1);
    float gp = 0.0f;
    const float Jf = (float)m.J;
    const BlReSiteMap sm = bl_re_site_map(m);
    const int tps = sm.tps, S = sm.S, sub = sm.sub;
    const int N = m.n_sites, T = m.T, J = m.J, Ks = m.Ks, Ko = m.Ko, vw = m.KO + 1;
    // (every load of the pass is unconditional, its index clamped into the array, and the value selected afterwards: a load
    // under a wave-uniform condition `k <= Ko` becomes a branch around it -- twenty basic blocks per batch of visits -- and
    // the loads of a batch are no longer in flight together)
    float beta[MK + 1], alpha[MK + 1];
#pragma unroll
    for (int k = 0; k <= MK; k++) {
        const float b = z[m.cb + min(k, Ks)], a = z[m.cb + Ks + 1 + min(k, Ko)];
        beta[k] = k <= Ks ? b : 0.0f;
        alpha[k] = k <= Ko ? a : 0.0f;
    }
    const float isd2_s = m.site_re ? bl_exp(-2.0f * z[m.o_phi_s]) : 0.0f;
    const float isd2_o = m.obs_re ? bl_exp(-2.0f * z[m.o_phi_o]) : 0.0f;
#pragma unroll
    for (int k = 0; k < 2 * MK + 3; k++) part[k] = 0.0f;
    const int row_ka = rv + T * J * vw, row_kb = row_ka + T;
    for (int rd = 0; rd < sm.rounds; rd++) {
        const int i_raw = rd * sm.ngrp + sm.grp;
        const bool live = i_raw < sm.cnt;
        const int i = sm.first + (live ? i_raw : sm.cnt - 1); // idle groups shadow the class's last site (their results are dropped)
        float x[MK];
        float eta = beta[0];
#pragma unroll
        for (int k = 0; k < MK; k++) {
            const float xk = rows[min(k, Ks) * ns + i]; // (k == Ks: the first visit row, discarded)
            x[k] = k < Ks ? xk : 0.0f;
            eta = fmaf(x[k], beta[k + 1], eta);
        }
        const float ui_ = z[m.site_re ? m.o_u + i : 0], vi_ = z[m.site_re ? m.o_v + i : 0];
        const float ui = m.site_re ? ui_ : 0.0f, vi = m.site_re ? vi_ : 0.0f;
        eta += ui;
        const float ee = bl_exp(-fabsf(eta)), lop = bl_log(1.0f + ee);
        const float log_psi = fminf(eta, 0.0f) - lop, log_1mpsi = fminf(-eta, 0.0f) - lop;
        // psi = sel / (1 + e^-|eta|) enters only d l / d eta = q - psi, written as ONE fused multiply-add in every instantiation (left to
        // the compiler, `q - sel * rc` is contracted when one period is a compile-time fact -- product and difference in one basic
        // block -- and not behind a period loop: the forms' draws would differ in the last bit)
        const float psi_sel = eta > 0.0f ? 1.0f : ee, psi_rc = bl_rcp(1.0f + ee);
        float dl_deta = 0.0f, dl_dv = 0.0f;
        for (int t = 0; t < T; t++) {
            float a = 0.0f, ga[MK + 1], gf = 0.0f;
#pragma unroll
            for (int k = 0; k <= MK; k++) ga[k] = 0.0f;
            // visits in batches of BL_RE_VB: all loads of a batch are issued before the first is used (a lone workgroup per
            // CU has little else to hide memory latency behind); a visit past J re-reads the last one and is dropped.  The
            // batch is compiled for a few covariate counts KB >= Ko (1, 2, 4, 8, 16 up to MK): one uniform branch per batch.
            auto batch = [&](auto kb_, int jb) {
                constexpr int KB = decltype(kb_)::value;
                float w[BL_RE_VB][KB + 1], eo[BL_RE_VB];
#pragma unroll
                for (int b = 0; b < BL_RE_VB; b++) {
                    const int jj = min(jb + b * tps, J - 1), v = t * J + jj;
                    const int r0 = (rv + v * vw) * ns + i;
#pragma unroll
                    for (int k = 0; k <= KB; k++) {
                        const float wk = rows[r0 + min(k, Ko) * ns];
                        w[b][k] = k <= Ko ? wk : 0.0f;
                    }
                    const float ev = z[m.obs_re ? m.o_e + v * N + i : 0];
                    eo[b] = m.obs_re ? ev : 0.0f;
                }
#pragma unroll
                for (int b = 0; b < BL_RE_VB; b++) {
                    const float ok = jb + b * tps < J ? 1.0f : 0.0f;
                    float u = w[b][0] * alpha[0];
#pragma unroll
                    for (int k = 1; k <= KB; k++) u = fmaf(w[b][k], alpha[k], u);
                    u = fmaf(w[b][0], vi + eo[b], u);
                    const float e = bl_exp(-fabsf(u)), op = 1.0f + e;
                    a = fmaf(ok, fminf(u, 0.0f) - bl_log(op), a);
                    float s = ok * (u > 0.0f ? e : 1.0f) * bl_rcp(op); // sigma(-u): d log sigma(u) / du
                    if constexpr (FP) {
                        // a detection where the rate acts on occupied sites: + log(1 + f1 e^-u), d/du -= f1 e^-u / (1 + f1 e^-u)
                        const float det = (w[b][0] > 0.0f ? ok : 0.0f) * fp.z1;
                        const float td = bl_exp(-fmaxf(u, -87.0f)) * det;
                        const float opf = fmaf(td, fp.f1, 1.0f);
                        a += bl_log(opf);
                        const float tr = td * bl_rcp(opf);
                        gf += tr;
                        s = fmaf(tr, -fp.f1, s);
                    }
#pragma unroll
                    for (int k = 0; k <= KB; k++) ga[k] = fmaf(s, w[b][k], ga[k]);
                }
            };
            // (measured 2-4 % slower: batches compiled for the rows' exact capacity KO, loading every row unconditionally without the
            // clamped index and the select)
            // (measured equal: a shorter last batch -- lengths 1 .. 3 compiled beside BL_RE_VB -- instead of a full one with dropped visits)
            for (int jb = sub; jb < J; jb += tps * BL_RE_VB) bl_re_tiered<MK>(Ko, batch, jb);
            for (int msk = S; msk < 64; msk <<= 1) { // the site's threads pool their visits
                a += __shfl_xor(a, msk);
                if constexpr (FP) gf += __shfl_xor(gf, msk);
#pragma unroll
                for (int k = 0; k <= MK; k++) ga[k] += __shfl_xor(ga[k], msk);
            }
            const float ka = rows[(row_ka + t) * ns + i];
            float kb = rows[(row_kb + t) * ns + i];
            float nnon = 0.0f, ndet = 0.0f;
            if constexpr (FP) { // counts of this (site, period) from the rows' ka = n_masked ln2 and kb = n_det log(tiny)
                ndet = __builtin_rintf(kb * (1.0f / -87.33654475f));
                nnon = Jf - ndet - __builtin_rintf(ka * (1.0f / BL_LN2));
                a = fmaf(nnon, fp.l1f1, a);
                kb = fmaf(ndet, fp.lf, nnon * fp.l1f);
            }
            const float A = log_psi + a + ka, B = log_1mpsi + kb;
            const float l = bl_logaddexp(A, B);
            const float q = bl_exp(A - l);
            if constexpr (FP) {
                if (live && sub == 0) { // d/dphi: the z = 1 branch (only when the rate acts there) and the z = 0 branch, each times f (1 - f)
                    const float d1 = fmaf(gf, fp.ff1, nnon * (-fp.f * fp.z1)), d0 = fmaf(ndet, 1.0f - fp.f, nnon * -fp.f);
                    gp += fmaf(q, d1 - d0, d0);
                }
            }
            if (live && sub == 0) {
                part[0] += l;
#pragma unroll
                for (int k = 0; k <= MK; k++) part[MK + 2 + k] = fmaf(q, ga[k], part[MK + 2 + k]);
            }
            dl_deta += fmaf(-psi_sel, psi_rc, q);
            dl_dv = fmaf(q, ga[0], dl_dv);
            if (m.obs_re && live) {
                // each replicate's own effect: d U / d e = -q d a / d nu + e / sd^2 (d a / d nu recomputed: nothing was kept per visit)
                auto batch_e = [&](auto kb_, int jb) {
                    constexpr int KB = decltype(kb_)::value;
                    float w[BL_RE_VB][KB + 1], eo[BL_RE_VB];
#pragma unroll
                    for (int b = 0; b < BL_RE_VB; b++) {
                        const int jj = min(jb + b * tps, J - 1), v = t * J + jj;
                        const int r0 = (rv + v * vw) * ns + i;
#pragma unroll
                        for (int k = 0; k <= KB; k++) {
                            const float wk = rows[r0 + min(k, Ko) * ns];
                            w[b][k] = k <= Ko ? wk : 0.0f;
                        }
                        eo[b] = z[m.o_e + v * N + i];
                    }
#pragma unroll
                    for (int b = 0; b < BL_RE_VB; b++) {
                        if (jb + b * tps < J) {
                            float u = w[b][0] * alpha[0];
#pragma unroll
                            for (int k = 1; k <= KB; k++) u = fmaf(w[b][k], alpha[k], u);
                            u = fmaf(w[b][0], vi + eo[b], u);
                            const float e = bl_exp(-fabsf(u));
                            float s = (u > 0.0f ? e : 1.0f) * bl_rcp(1.0f + e);
                            if constexpr (FP) {
                                const float td = w[b][0] > 0.0f ? bl_exp(-fmaxf(u, -87.0f)) * fp.z1 : 0.0f;
                                s = fmaf(td * bl_rcp(fmaf(td, fp.f1, 1.0f)), -fp.f1, s);
                            }
                            g[m.o_e + (t * J + jb + b * tps) * N + i] = fmaf(eo[b], isd2_o, -q * s * w[b][0]);
                        }
                    }
                };
                for (int jb = sub; jb < J; jb += tps * BL_RE_VB) bl_re_tiered<MK>(Ko, batch_e, jb);
            }
        }
        if (live && sub == 0) {
            part[1] += dl_deta;
#pragma unroll
            for (int k = 0; k < MK; k++) part[2 + k] = fmaf(dl_deta, x[k], part[2 + k]);
            if (m.site_re) {
                g[m.o_u + i] = fmaf(ui, isd2_s, -dl_deta);
                g[m.o_v + i] = fmaf(vi, isd2_s, -dl_dv);
            }
        }
    }
    if constexpr (FP) *gphi = gp;
}

// ---- N-mixture with random effects (kind 3; biolith/models/nmixture.py:139-141, 166-172, 199-214) ----
// site_re_abu_i joins the abundance predictor, site_re_det_i and obs_re_itj the detection predictor; the layout of the coordinates,
// the sds and the effects' priors are the occupancy model's (kind 0), so only this site pass differs.  With nu = logit p:
//   l = sum_j m y_j nu_j - lambda + logsumexp_n [ n (eta + c) - lgamma(n+1) + B_n ],   c = sum_j m log(1 - p_j),
//   d l / d eta = E[n] - lambda,   d l / d nu_j = m (y_j - E[n] p_j)          (bl_eval_sites_nmix, occu_device.hpp).
// One thread per site (this model's sums over n are cheap -- no per-visit recursion); rows: the count model's, visit = (m y, m, w..).
template <int MK>
__device__ __forceinline__ void bl_nmix_re_site_pass(const BlReModel &m, const float *__restrict__ rows, int ns, int rv,
                                                     const float *__restrict__ z, float *__restrict__ g, float (&part)[2 * MK + 3])
{
    const int N = m.n_sites, T = m.T, J = m.J, Ks = m.Ks, Ko = m.Ko, vw = m.KO + 2, K = m.max_abundance;
    float beta[MK + 1], alpha[MK + 1];
#pragma unroll
    for (int k = 0; k <= MK; k++) {
        const float b = z[m.cb + min(k, Ks)], a = z[m.cb + Ks + 1 + min(k, Ko)];
        beta[k] = k <= Ks ? b : 0.0f;
        alpha[k] = k <= Ko ? a : 0.0f;
    }
    const float isd2_s = m.site_re ? bl_exp(-2.0f * z[m.o_phi_s]) : 0.0f;
    const float isd2_o = m.obs_re ? bl_exp(-2.0f * z[m.o_phi_o]) : 0.0f;
#pragma unroll
    for (int k = 0; k < 2 * MK + 3; k++) part[k] = 0.0f;
    for (int i = threadIdx.x; i < N; i += BL_RE_NT) {
        float x[MK];
        float eta = beta[0];
#pragma unroll
        for (int k = 0; k < MK; k++) {
            const float xk = rows[min(k, Ks) * ns + i];
            x[k] = k < Ks ? xk : 0.0f;
            eta = fmaf(x[k], beta[k + 1], eta);
        }
        const float ui = m.site_re ? z[m.o_u + i] : 0.0f, vi = m.site_re ? z[m.o_v + i] : 0.0f;
        eta += ui;
        const float lam = bl_exp(fminf(eta, 80.0f));
        float dl_deta = 0.0f, dl_dv = 0.0f;
        for (int t = 0; t < T; t++) {
            float a = 0.0f, c = 0.0f, gy[MK + 1], gp[MK + 1];
#pragma unroll
            for (int k = 0; k <= MK; k++) { gy[k] = 0.0f; gp[k] = 0.0f; }
            for (int j = 0; j < J; j++) {
                const int v = t * J + j, r0 = (rv + v * vw) * ns + i;
                const float ym = rows[r0], mk = rows[r0 + ns];
                float w[MK];
                float nu = alpha[0] + vi + (m.obs_re ? z[m.o_e + v * N + i] : 0.0f);
#pragma unroll
                for (int k = 0; k < MK; k++) {
                    const float wk = rows[r0 + (2 + min(k, max(Ko, 1) - 1)) * ns];
                    w[k] = k < Ko ? wk : 0.0f;
                    nu = fmaf(w[k], alpha[k + 1], nu);
                }
                const float e = bl_exp(-fabsf(nu)), op = 1.0f + e;
                const float sp = fmaxf(nu, 0.0f) + bl_log(op);              // softplus(nu)
                const float p = (nu > 0.0f ? 1.0f : e) * bl_rcp(op) * mk;   // m sigmoid(nu)
                a = fmaf(ym, nu, a);
                c = fmaf(mk, -sp, c);
                gy[0] += ym; gp[0] += p;
#pragma unroll
                for (int k = 0; k < MK; k++) { gy[k + 1] = fmaf(ym, w[k], gy[k + 1]); gp[k + 1] = fmaf(p, w[k], gp[k + 1]); }
            }
            // logsumexp over n of  n (eta + c) - lgamma(n+1) + B_n:  the maximum and the last term within 25 nats of the running
            // maximum (the terms are concave in n above the largest count), then the sums up to there
            const float slope = eta + c;
            const float *b = m.tab + (size_t)t * (K + 1) * m.tab_ld + i;
            float mx = -INFINITY;
            int hi = 0;
            for (int n = 0; n <= K; n++) {
                const float tn = fmaf((float)n, slope, b[(size_t)n * m.tab_ld] - BL_LGAMMA1P[n]);
                mx = fmaxf(mx, tn);
                hi = tn >= mx - 25.0f ? n : hi;
            }
            float S = 0.0f, S1 = 0.0f;
            for (int n = 0; n <= hi; n++) {
                const float en = bl_exp(fmaf((float)n, slope, b[(size_t)n * m.tab_ld] - BL_LGAMMA1P[n]) - mx);
                S += en;
                S1 = fmaf((float)n, en, S1);
            }
            const float En = S1 * bl_rcp(S);
            part[0] += a - lam + mx + bl_log(S);
            dl_deta += En - lam;
#pragma unroll
            for (int k = 0; k <= MK; k++) part[MK + 2 + k] += gy[k] - En * gp[k];
            dl_dv += gy[0] - En * gp[0];
            if (m.obs_re) { // each replicate's own effect: d U / d e = -(m y - E[n] m p) + e / sd^2 (p recomputed: nothing was kept per visit)
                for (int j = 0; j < J; j++) {
                    const int v = t * J + j, r0 = (rv + v * vw) * ns + i;
                    const float ym = rows[r0], mk = rows[r0 + ns], ev = z[m.o_e + v * N + i];
                    float nu = alpha[0] + vi + ev;
#pragma unroll
                    for (int k = 0; k < MK; k++) {
                        const float wk = rows[r0 + (2 + min(k, max(Ko, 1) - 1)) * ns];
                        nu = fmaf(k < Ko ? wk : 0.0f, alpha[k + 1], nu);
                    }
                    const float e = bl_exp(-fabsf(nu));
                    const float p = (nu > 0.0f ? 1.0f : e) * bl_rcp(1.0f + e) * mk;
                    g[m.o_e + v * N + i] = fmaf(ev, isd2_o, -(ym - En * p));
                }
            }
        }
        part[1] += dl_deta;
#pragma unroll
        for (int k = 0; k < MK; k++) part[2 + k] = fmaf(dl_deta, x[k], part[2 + k]);
        if (m.site_re) {
            g[m.o_u + i] = fmaf(ui, isd2_s, -dl_deta);
            g[m.o_v + i] = fmaf(vi, isd2_s, -dl_dv);
        }
    }
}

// ---- Royle-Nichols with random effects (kind 4; biolith/models/occu_rn.py:151-154, 172-184, 199-212) ----
// site_re_abu_i joins the abundance predictor, site_re_det_i and obs_re_itj the detection predictor.  Per (site, period)
//   l = logsumexp_n [ log pi_n + sum_j m_j log P(y_j | n) ],  pi_n = the Poisson weights renormalised over 0..K,
//   P(y = 1 | n) = 1 - q^n = r b_n  (b_n = b_{n-1} q + 1, no cancellation),  both Bernoulli branches clamped as numpyro clamps them
//   (log P(1 | 0) = log tiny; log P(0 | n) = max(n log q, log eps); a clamped factor has no gradient) -- rn_device.hpp's arithmetic,
// one thread per site with the K + 1 terms of a (site, period) in a private column: the effects make every site's predictors its own,
// and this form serves the reference's sizes (its tests: 100 sites) -- the work-proportional kernel of the plain model is not reused.
// rows: the plain model's (visit = (c, c w_1..w_KO, spare), c = +1 detection, -1 none, 0 masked).
// FP (kind 5; occu_rn.py:133-138, 214-221): y ~ Bernoulli(1 - (1 - p)(1 - f)), f = sigmoid(phi) at z[m.o_fp]:  P(y = 1 | n) = f + (1 - f) r b_n,
// P(y = 0 | n) = q^n (1 - f); *gphi accumulates d ll / d phi (chain rule through f included).
template <int MK, bool FP = false>
__device__ __forceinline__ void bl_rn_re_site_pass(const BlReModel &m, const float *__restrict__ rows, int ns, int rv,
                                                   const float *__restrict__ z, float *__restrict__ g, float (&part)[2 * MK + 3],
                                                   float *gphi = nullptr)
{
    const int N = m.n_sites, T = m.T, J = m.J, Ks = m.Ks, Ko = m.Ko, vw = m.KO + 2, K = m.max_abundance;
    constexpr float LOG_EPS = -15.9423847f, LOG_TINY = -87.3365448f, ONE_M_EPS = 0.99999988f, TINY = 1.17549435e-38f;
    float beta[MK + 1], alpha[MK + 1];
#pragma unroll
    for (int k = 0; k <= MK; k++) {
        const float b = z[m.cb + min(k, Ks)], a = z[m.cb + Ks + 1 + min(k, Ko)];
        beta[k] = k <= Ks ? b : 0.0f;
        alpha[k] = k <= Ko ? a : 0.0f;
    }
    const float isd2_s = m.site_re ? bl_exp(-2.0f * z[m.o_phi_s]) : 0.0f;
    const float isd2_o = m.obs_re ? bl_exp(-2.0f * z[m.o_phi_o]) : 0.0f;
    float fpr = 0.0f, gq = 1.0f, lgq = 0.0f, dfp = 0.0f; // the rate f, 1 - f, log(1 - f); d ll / d f
    if constexpr (FP) {
        const float phi = z[m.o_fp], e = bl_exp(-fabsf(phi)), op = 1.0f + e, iop = bl_rcp(op);
        fpr = (phi > 0.0f ? 1.0f : e) * iop;
        gq = (phi > 0.0f ? e : 1.0f) * iop;
        lgq = -(fmaxf(phi, 0.0f) + bl_log(op));
    }
#pragma unroll
    for (int k = 0; k < 2 * MK + 3; k++) part[k] = 0.0f;
    float term[BL_RN_NB];
    for (int i = threadIdx.x; i < N; i += BL_RE_NT) {
        float x[MK];
        float eta = beta[0];
#pragma unroll
        for (int k = 0; k < MK; k++) {
            const float xk = rows[min(k, Ks) * ns + i];
            x[k] = k < Ks ? xk : 0.0f;
            eta = fmaf(x[k], beta[k + 1], eta);
        }
        const float ui = m.site_re ? z[m.o_u + i] : 0.0f, vi = m.site_re ? z[m.o_v + i] : 0.0f;
        eta += ui;
        // the renormalised weights: log pi_n = n eta - lgamma(n + 1) - log Z,  d log Z / d eta = E_pi[n]
        float m0 = -INFINITY;
        for (int n = 0; n <= K; n++) m0 = fmaxf(m0, fmaf((float)n, eta, -BL_LGAMMA1P[n]));
        float sz = 0.0f, s1 = 0.0f;
        for (int n = 0; n <= K; n++) {
            const float e = bl_exp(fmaf((float)n, eta, -BL_LGAMMA1P[n]) - m0);
            sz += e;
            s1 = fmaf((float)n, e, s1);
        }
        const float logz = m0 + bl_log(sz), en_prior = s1 * bl_rcp(sz);
        float dl_deta = 0.0f, dl_dv = 0.0f;
        for (int t = 0; t < T; t++) {
            for (int n = 0; n <= K; n++) term[n] = fmaf((float)n, eta, -BL_LGAMMA1P[n]) - logz;
            for (int j = 0; j < J; j++) {
                const int v = t * J + j, r0 = (rv + v * vw) * ns + i;
                const float c = rows[r0];
                if (c == 0.0f) continue;
                float sc = alpha[0] * c;
#pragma unroll
                for (int k = 0; k < MK; k++) {
                    const float wk = rows[r0 + (1 + min(k, max(Ko, 1) - 1)) * ns];
                    sc = fmaf(k < Ko ? wk : 0.0f, alpha[k + 1], sc);
                }
                const float nu = fmaf(c, sc, vi + (m.obs_re ? z[m.o_e + v * N + i] : 0.0f));
                const float e = bl_exp(-fabsf(nu)), op = 1.0f + e, iop = bl_rcp(op);
                const float r = (nu > 0.0f ? 1.0f : e) * iop, q = (nu > 0.0f ? e : 1.0f) * iop;
                const float lq = -(fmaxf(nu, 0.0f) + bl_log(op)); // log(1 - r)
                if (c < 0.0f) {
                    if constexpr (FP) term[0] += lgq;
                    for (int n = 1; n <= K; n++) term[n] += fmaxf(fmaf((float)n, lq, lgq), LOG_EPS);
                } else {
                    term[0] += FP ? bl_log(fmaxf(fminf(fpr, ONE_M_EPS), TINY)) : LOG_TINY;
                    float b = 1.0f;
                    for (int n = 1; n <= K; n++) {
                        term[n] += bl_log(fmaxf(fminf(fmaf(gq * r, b, fpr), ONE_M_EPS), TINY));
                        b = fmaf(b, q, 1.0f);
                    }
                }
            }
            float mx = -INFINITY;
            for (int n = 0; n <= K; n++) mx = fmaxf(mx, term[n]);
            float S = 0.0f, S1 = 0.0f;
            for (int n = 0; n <= K; n++) {
                const float e = bl_exp(term[n] - mx);
                term[n] = e; // (unnormalised posterior weight of n)
                S += e;
                S1 = fmaf((float)n, e, S1);
            }
            const float inv = bl_rcp(S);
            part[0] += mx + bl_log(S);
            dl_deta += S1 * inv - en_prior;
            for (int j = 0; j < J; j++) {
                const int v = t * J + j, r0 = (rv + v * vw) * ns + i;
                const float c = rows[r0], ev = m.obs_re ? z[m.o_e + v * N + i] : 0.0f;
                float w[MK];
                float sc = alpha[0] * c;
#pragma unroll
                for (int k = 0; k < MK; k++) {
                    const float wk = rows[r0 + (1 + min(k, max(Ko, 1) - 1)) * ns];
                    w[k] = k < Ko ? wk : 0.0f; // (c w_k)
                    sc = fmaf(w[k], alpha[k + 1], sc);
                }
                float dnu = 0.0f;
                if (c != 0.0f) {
                    const float nu = fmaf(c, sc, vi + ev);
                    const float e = bl_exp(-fabsf(nu)), op = 1.0f + e, iop = bl_rcp(op);
                    const float r = (nu > 0.0f ? 1.0f : e) * iop, q = (nu > 0.0f ? e : 1.0f) * iop;
                    if (c < 0.0f) { // d (n log q) / d nu = -n r, where the floor does not hold  (FP: d log(q^n (1 - f)) / d f = -1 / (1 - f))
                        const float lq = -(fmaxf(nu, 0.0f) + bl_log(op));
                        float acc = 0.0f, acc0 = 0.0f;
                        for (int n = 1; n <= K; n++) {
                            const bool open_ = fmaf((float)n, lq, lgq) > LOG_EPS;
                            acc = fmaf(open_ ? (float)n : 0.0f, term[n], acc);
                            if constexpr (FP) acc0 += open_ ? term[n] : 0.0f;
                        }
                        dnu = -r * acc * inv;
                        if constexpr (FP) dfp -= (acc0 + (lgq > LOG_EPS ? term[0] : 0.0f)) * inv * bl_rcp(gq);
                    } else if constexpr (!FP) { // d log(1 - q^n) / d nu = n q^n / b_n = n (1 / b_n - r), where neither clamp holds
                        float b = 1.0f, acc = 0.0f;
                        for (int n = 1; n <= K; n++) {
                            const float p = r * b;
                            acc = fmaf((p <= ONE_M_EPS && p >= TINY) ? (float)n * (bl_rcp(b) - r) : 0.0f, term[n], acc);
                            b = fmaf(b, q, 1.0f);
                        }
                        dnu = acc * inv;
                    } else {        // P = f + (1 - f) r b_n:  d log P / d nu = n r q^n (1 - f) / P,  d log P / d f = q^n / P,  q^n = 1 - r b_n
                        float b = 1.0f, acc = 0.0f, accf = (fpr <= ONE_M_EPS && fpr >= TINY) ? term[0] * bl_rcp(fpr) : 0.0f; // (n = 0: P = f, q^0 = 1)
                        for (int n = 1; n <= K; n++) {
                            const float p = fmaf(gq * r, b, fpr), qn = fmaf(-r, b, 1.0f);
                            const float wp = (p <= ONE_M_EPS && p >= TINY) ? term[n] * bl_rcp(p) : 0.0f;
                            acc = fmaf((float)n * qn, wp, acc);
                            accf = fmaf(qn, wp, accf);
                            b = fmaf(b, q, 1.0f);
                        }
                        dnu = acc * r * gq * inv;
                        dfp += accf * inv;
                    }
                }
                part[MK + 2] += dnu;
#pragma unroll
                for (int k = 0; k < MK; k++) part[MK + 3 + k] = fmaf(dnu * c, w[k], part[MK + 3 + k]); // w_k = c (c w_k)
                dl_dv += dnu;
                if (m.obs_re) g[m.o_e + v * N + i] = fmaf(ev, isd2_o, -dnu);
            }
        }
        part[1] += dl_deta;
#pragma unroll
        for (int k = 0; k < MK; k++) part[2 + k] = fmaf(dl_deta, x[k], part[2 + k]);
        if (m.site_re) {
            g[m.o_u + i] = fmaf(ui, isd2_s, -dl_deta);
            g[m.o_v + i] = fmaf(vi, isd2_s, -dl_dv);
        }
    }
    if constexpr (FP) *gphi = dfp * fpr * gq; // (d f / d phi = f (1 - f))
}

// ---- occu_cop with random effects (kind 6; biolith/models/occu_cop.py:183-186, 204-210, 229-243) ----
// site_re_occ_i joins the occupancy predictor, site_re_det_i and obs_re_itj the log detection rate; no false-positive rates.  Per
// (site, period), with the parameter-free part of the Poisson log-pmf in the potential's constant (the count model's rows:
// visit = (y, dur, w..), masked visits y = dur = 0; ka = sum y):
//   a = sum_j (y_j nu_j - dur_j e^nu_j),   l = log psi + a  if any count is positive (Poisson(0) gives it probability 0 at z = 0),
//   else logaddexp(log psi + a, log(1 - psi));   q = P(z = 1 | y);   d l / d eta = q - psi,   d l / d nu_j = q (y_j - dur_j e^nu_j).
// One thread per site (bl_eval_sites_cop's arithmetic without the false-positive terms).
// FP (kind 7; occu_cop.py:158-170, 244-248): a false-positive rate f = e^phi (phi at z[m.o_fp]) on every site ("constant") or on the
// unoccupied ones: rate_1 = dur (e^nu + f_c), rate_0 = dur (f_u + f_c); both branches are then live, y log dur - lgamma stays in the constant.
template <int MK, bool FP = false>
__device__ __forceinline__ void bl_cop_re_site_pass(const BlReModel &m, const float *__restrict__ rows, int ns, int rv,
                                                    const float *__restrict__ z, float *__restrict__ g, float (&part)[2 * MK + 3],
                                                    float *gphi = nullptr)
{
    const int N = m.n_sites, T = m.T, J = m.J, Ks = m.Ks, Ko = m.Ko, vw = m.KO + 2, V = T * J;
    float beta[MK + 1], alpha[MK + 1];
#pragma unroll
    for (int k = 0; k <= MK; k++) {
        const float b = z[m.cb + min(k, Ks)], a = z[m.cb + Ks + 1 + min(k, Ko)];
        beta[k] = k <= Ks ? b : 0.0f;
        alpha[k] = k <= Ko ? a : 0.0f;
    }
    const float isd2_s = m.site_re ? bl_exp(-2.0f * z[m.o_phi_s]) : 0.0f;
    const float isd2_o = m.obs_re ? bl_exp(-2.0f * z[m.o_phi_o]) : 0.0f;
    float fr = 0.0f, f_c = 0.0f, f0 = 0.0f, lf0 = 0.0f, dfp = 0.0f; // the rate, its part on occupied sites, the unoccupied sites' rate and its log; d ll / d f
    if constexpr (FP) {
        fr = bl_exp(fminf(z[m.o_fp], 80.0f));
        f_c = m.fp_mode == 1 ? fr : 0.0f;
        f0 = fr; // (f_u + f_c: one of the two is the rate, the other 0)
        lf0 = bl_log(f0);
    }
#pragma unroll
    for (int k = 0; k < 2 * MK + 3; k++) part[k] = 0.0f;
    for (int i = threadIdx.x; i < N; i += BL_RE_NT) {
        float x[MK];
        float eta = beta[0];
#pragma unroll
        for (int k = 0; k < MK; k++) {
            const float xk = rows[min(k, Ks) * ns + i];
            x[k] = k < Ks ? xk : 0.0f;
            eta = fmaf(x[k], beta[k + 1], eta);
        }
        const float ui = m.site_re ? z[m.o_u + i] : 0.0f, vi = m.site_re ? z[m.o_v + i] : 0.0f;
        eta += ui;
        const float ee = bl_exp(-fabsf(eta)), ope = 1.0f + ee, le = bl_log(ope);
        const float psi = (eta > 0.0f ? 1.0f : ee) * bl_rcp(ope);
        const float log_psi = fminf(eta, 0.0f) - le, log_1mpsi = -fmaxf(eta, 0.0f) - le;
        float dl_deta = 0.0f, dl_dv = 0.0f;
        for (int t = 0; t < T; t++) {
            float a = 0.0f, df1 = 0.0f, gy[MK + 1], gd[MK + 1];
#pragma unroll
            for (int k = 0; k <= MK; k++) { gy[k] = 0.0f; gd[k] = 0.0f; }
            for (int j = 0; j < J; j++) {
                const int v = t * J + j, r0 = (rv + v * vw) * ns + i;
                const float y = rows[r0], dur = rows[r0 + ns];
                float w[MK];
                float nu = alpha[0] + vi + (m.obs_re ? z[m.o_e + v * N + i] : 0.0f);
#pragma unroll
                for (int k = 0; k < MK; k++) {
                    const float wk = rows[r0 + (2 + min(k, max(Ko, 1) - 1)) * ns];
                    w[k] = k < Ko ? wk : 0.0f;
                    nu = fmaf(w[k], alpha[k + 1], nu);
                }
                const float lam = bl_exp(fminf(nu, 80.0f)), rate = dur * lam;
                if constexpr (!FP) {
                    a += fmaf(y, nu, -rate);
                    gy[0] += y; gd[0] += rate;
#pragma unroll
                    for (int k = 0; k < MK; k++) { gy[k + 1] = fmaf(y, w[k], gy[k + 1]); gd[k + 1] = fmaf(rate, w[k], gd[k + 1]); }
                } else { // rate_1 = dur (lam + f_c):  d / d nu = (y / (lam + f_c) - dur) lam,  d / d f_c = y / (lam + f_c) - dur
                    const float l1 = lam + f_c, il1 = bl_rcp(l1);
                    a += fmaf(y, bl_log(l1), -dur * l1);
                    const float yw = y * lam * il1; // (the weight y takes in the nu-gradient)
                    gy[0] += yw; gd[0] += rate;
#pragma unroll
                    for (int k = 0; k < MK; k++) { gy[k + 1] = fmaf(yw, w[k], gy[k + 1]); gd[k + 1] = fmaf(rate, w[k], gd[k + 1]); }
                    df1 += m.fp_mode == 1 ? fmaf(y, il1, -dur) : 0.0f;
                }
            }
            const float ka = rows[(rv + V * vw + t) * ns + i], kb = rows[(rv + V * vw + T + t) * ns + i]; // sum y, sum dur over the valid visits
            const float A = log_psi + a;
            float l = A, q = 1.0f;
            if (FP || !(ka > 0.0f)) { // (no rate: the unoccupied branch is possible only with no count at all)
                const float B = log_1mpsi + (FP ? fmaf(ka, lf0, -kb * f0) : 0.0f);
                const float d = A - B, e = bl_exp(-fabsf(d)), op = 1.0f + e;
                l = fmaxf(A, B) + bl_log(op);
                q = (d > 0.0f ? 1.0f : e) * bl_rcp(op);
            }
            if constexpr (FP) dfp += q * df1 + (1.0f - q) * (ka * bl_rcp(f0) - kb); // d a0 / d f = sum y / f0 - sum dur
            part[0] += l;
            dl_deta += q - psi;
#pragma unroll
            for (int k = 0; k <= MK; k++) part[MK + 2 + k] = fmaf(q, gy[k] - gd[k], part[MK + 2 + k]);
            dl_dv = fmaf(q, gy[0] - gd[0], dl_dv);
            if (m.obs_re) { // each replicate's own effect (its rate recomputed: nothing was kept per visit)
                for (int j = 0; j < J; j++) {
                    const int v = t * J + j, r0 = (rv + v * vw) * ns + i;
                    const float y = rows[r0], dur = rows[r0 + ns], ev = z[m.o_e + v * N + i];
                    float nu = alpha[0] + vi + ev;
#pragma unroll
                    for (int k = 0; k < MK; k++) {
                        const float wk = rows[r0 + (2 + min(k, max(Ko, 1) - 1)) * ns];
                        nu = fmaf(k < Ko ? wk : 0.0f, alpha[k + 1], nu);
                    }
                    const float lam = bl_exp(fminf(nu, 80.0f));
                    g[m.o_e + v * N + i] = fmaf(ev, isd2_o, -q * (y * (FP ? lam * bl_rcp(lam + f_c) : 1.0f) - dur * lam));
                }
            }
        }
        part[1] += dl_deta;
#pragma unroll
        for (int k = 0; k < MK; k++) part[2 + k] = fmaf(dl_deta, x[k], part[2 + k]);
        if (m.site_re) {
            g[m.o_u + i] = fmaf(ui, isd2_s, -dl_deta);
            g[m.o_v + i] = fmaf(vi, isd2_s, -dl_dv);
        }
    }
    if constexpr (FP) *gphi = dfp * fr; // (d f / d phi = f)
}

// ---- occu_cs (biolith/models/occu_cs.py:120-232): s ~ Normal(mu_f, sigma_f), f ~ Bernoulli(z p), z ~ Bernoulli(psi); z and every
// f summed out (the f of different replicates are independent given z).  Coordinates: [beta, alpha, mu0, x1 = log(mu1 - mu0),
// log sigma0, log sigma1].  Site pass: part[0] = ll, [1..5] d/d beta, [6..10] d/d alpha, [11..14] d/d (mu0, mu1, log sigma0, log sigma1).
template <int MK>
__device__ __forceinline__ void bl_cs_site_pass(const BlReModel &m, const float *__restrict__ rows, int ns, const float *__restrict__ z,
                                                float (&part)[2 * MK + 7])
{
    const BlReSiteMap sm = bl_re_site_map(m);
    const int tps = sm.tps, S = sm.S, sub = sm.sub;
    const int T = m.T, J = m.J, Ks = m.Ks, Ko = m.Ko, vw = m.KO + 1;
    // (every load of the pass is unconditional, its index clamped into the array, and the value selected afterwards: a load
    // under a wave-uniform condition `k <= Ko` becomes a branch around it -- twenty basic blocks per batch of visits -- and
    // the loads of a batch are no longer in flight together)
    float beta[MK + 1], alpha[MK + 1];
#pragma unroll
    for (int k = 0; k <= MK; k++) {
        const float b = z[min(k, Ks)], a = z[Ks + 1 + min(k, Ko)];
        beta[k] = k <= Ks ? b : 0.0f;
        alpha[k] = k <= Ko ? a : 0.0f;
    }
    const float mu0 = z[m.G0], mu1 = mu0 + bl_exp(z[m.G0 + 1]), ls0 = z[m.G0 + 2], ls1 = z[m.G0 + 3];
    const float is0 = bl_exp(-ls0), is1 = bl_exp(-ls1);
    const float TINY = 1.1754944e-38f, PMAX = 1.0f - 1.1920929e-07f, HL2PI = 0.9189385f;
    const float l_f1_z0 = -87.33654475f, l_f0_z0 = -1.1754944e-38f; // log(tiny), log1p(-tiny): numpyro clamps P(f = 1 | z = 0) = 0 to tiny
#pragma unroll
    for (int k = 0; k < 2 * MK + 7; k++) part[k] = 0.0f;
    for (int rd = 0; rd < sm.rounds; rd++) {
        const int i_raw = rd * sm.ngrp + sm.grp;
        const bool live = i_raw < sm.cnt;
        const int i = sm.first + (live ? i_raw : sm.cnt - 1); // idle groups shadow the class's last site (their results are dropped)
        float x[MK];
        float eta = beta[0];
#pragma unroll
        for (int k = 0; k < MK; k++) {
            const float xk = rows[min(k, Ks) * ns + i]; // (k == Ks: the first visit row, discarded)
            x[k] = k < Ks ? xk : 0.0f;
            eta = fmaf(x[k], beta[k + 1], eta);
        }
        const float ee = bl_exp(-fabsf(eta)), lop = bl_log(1.0f + ee);
        const float log_psi = fminf(eta, 0.0f) - lop, log_1mpsi = fminf(-eta, 0.0f) - lop;
        const float psi = (eta > 0.0f ? 1.0f : ee) * bl_rcp(1.0f + ee);
        float dl_deta = 0.0f;
        for (int t = 0; t < T; t++) {
            // r[0] = a1, r[1] = a0, r[2..6] = d a1 / d alpha, r[7..10] = d a1 / d (mu0, mu1, ls0, ls1), r[11..14] = the same of a0
            constexpr int RB = MK + 3; // r[RB .. RB+3] = d a1 / d (mu0, mu1, ls0, ls1), r[RB+4 .. RB+7] = the same of a0
            float r[MK + 11];
#pragma unroll
            for (int k = 0; k < MK + 11; k++) r[k] = 0.0f;
            for (int j = sub; j < J; j += tps) {
                const int v = t * J + j;
                const int r0 = (m.KS + v * vw) * ns + i;
                float w[MK + 1];
#pragma unroll
                for (int k = 0; k <= MK; k++) {
                    const float wk = rows[r0 + min(k, Ko) * ns];
                    w[k] = k <= Ko ? wk : 0.0f;
                }
                const float c = w[0]; // 1: the replicate has a score, 0: masked
                const float sc = m.scores[(size_t)v * m.n_stride + i];
                float nu = c * alpha[0];
#pragma unroll
                for (int k = 1; k <= MK; k++) nu = fmaf(w[k], alpha[k], nu);
                const float e0 = (sc - mu0) * is0, e1 = (sc - mu1) * is1;
                const float lphi0 = fmaf(-0.5f * e0, e0, -ls0 - HL2PI), lphi1 = fmaf(-0.5f * e1, e1, -ls1 - HL2PI);
                const float en = bl_exp(-fabsf(nu)), ron = bl_rcp(1.0f + en);
                const float p = (nu > 0.0f ? 1.0f : en) * ron;
                const bool inside = p > TINY && p < PMAX;
                const float pc = fminf(fmaxf(p, TINY), PMAX);
                // z = 1
                const float t1 = bl_log(pc) + lphi1, t0 = bl_log(1.0f - pc) + lphi0;
                const float L1 = bl_logaddexp(t0, t1), w1 = bl_exp(t1 - L1), w0 = bl_exp(t0 - L1);
                const float dnu = inside ? fmaf(w1, 1.0f - p, -w0 * p) : 0.0f;
                // z = 0
                const float u1 = l_f1_z0 + lphi1, u0 = l_f0_z0 + lphi0;
                const float L0 = bl_logaddexp(u0, u1), v1 = bl_exp(u1 - L0), v0 = bl_exp(u0 - L0);
                r[0] = fmaf(c, L1, r[0]); r[1] = fmaf(c, L0, r[1]);
#pragma unroll
                for (int k = 0; k <= MK; k++) r[2 + k] = fmaf(dnu, w[k], r[2 + k]); // (w carries the mask)
                r[RB] = fmaf(c * w0, e0 * is0, r[RB]); r[RB + 1] = fmaf(c * w1, e1 * is1, r[RB + 1]);
                r[RB + 2] = fmaf(c * w0, fmaf(e0, e0, -1.0f), r[RB + 2]); r[RB + 3] = fmaf(c * w1, fmaf(e1, e1, -1.0f), r[RB + 3]);
                r[RB + 4] = fmaf(c * v0, e0 * is0, r[RB + 4]); r[RB + 5] = fmaf(c * v1, e1 * is1, r[RB + 5]);
                r[RB + 6] = fmaf(c * v0, fmaf(e0, e0, -1.0f), r[RB + 6]); r[RB + 7] = fmaf(c * v1, fmaf(e1, e1, -1.0f), r[RB + 7]);
            }
            for (int msk = S; msk < 64; msk <<= 1) {
#pragma unroll
                for (int k = 0; k < MK + 11; k++) r[k] += __shfl_xor(r[k], msk);
            }
            const float A = log_psi + r[0], B = log_1mpsi + r[1];
            const float l = bl_logaddexp(A, B);
            const float q = bl_exp(A - l), q0 = 1.0f - q;
            if (live && sub == 0) {
                part[0] += l;
#pragma unroll
                for (int k = 0; k <= MK; k++) part[MK + 2 + k] = fmaf(q, r[2 + k], part[MK + 2 + k]);
#pragma unroll
                for (int k = 0; k < 4; k++) part[2 * MK + 3 + k] += fmaf(q, r[RB + k], q0 * r[RB + 4 + k]);
            }
            dl_deta += q - psi;
        }
        if (live && sub == 0) {
            part[1] += dl_deta;
#pragma unroll
            for (int k = 0; k < MK; k++) part[2 + k] = fmaf(dl_deta, x[k], part[2 + k]);
        }
    }
}

// occu_cs: potential gradient of coordinate G0 + e (e = 0 mu0, 1 x1, 2 log sigma0, 3 log sigma1) from the reduced sums
// (red[OX], red[OX+1], red[OX+6], red[OX+7] = d ll / d (mu0, mu1, log sigma0, log sigma1)) and the priors: mu0 ~ Normal(l0, s0);
// mu1 ~ Normal(l1, s1) truncated below at mu0, in x1 = log(mu1 - mu0); sigma_f ~ Gamma(a, b) in log sigma_f.
template <int MK>
__device__ __forceinline__ float bl_cs_extra_grad(const BlReModel &m, int e, const float *z, const double *red)
{
    constexpr int OX = BL_RE_OX(MK);
    const float mu0 = z[m.G0], ex1 = bl_exp(z[m.G0 + 1]), mu1 = mu0 + ex1;
    if (e >= 2) {
        // (selects, not m.cs_sg[2 * (e - 2)]: a dynamically indexed member would put the whole model struct into scratch memory
        // -- 224 bytes per lane, read back on every use of the model: 12-16 % of a leapfrog, measured)
        const float a = e == 2 ? m.cs_sg[0] : m.cs_sg[2], b = e == 2 ? m.cs_sg[1] : m.cs_sg[3];
        return -((float)red[OX + 4 + e] + a - b * bl_exp(z[m.G0 + e]));
    }
    const float l0 = m.cs_mu[0], s0 = m.cs_mu[1], l1 = m.cs_mu[2], s1 = m.cs_mu[3];
    const float z1 = (mu1 - l1) / s1, dmu1 = (float)red[OX + 1] - z1 / s1;
    if (e == 1) return -(dmu1 * ex1 + 1.0f);
    const float z0 = (mu0 - l0) / s0, zl = (mu0 - l1) / s1;
    const float hazard = bl_exp(-0.5f * zl * zl - 0.9189385f) / (0.5f * erfcf(zl * 0.70710678f)) / s1; // d/d mu0 of -log(1 - Phi(zl))
    return -((float)red[OX] - z0 / s0 + dmu1 + hazard);
}
// occu_cs: the four extra coordinates' share of the potential (their priors, Jacobians, the truncation's normaliser)
__device__ __forceinline__ double bl_cs_extra_potential(const BlReModel &m, const float *z)
{
    const double mu0 = z[m.G0], x1 = z[m.G0 + 1], mu1 = mu0 + exp(x1);
    const double l0 = m.cs_mu[0], s0 = m.cs_mu[1], l1 = m.cs_mu[2], s1 = m.cs_mu[3];
    const double z0 = (mu0 - l0) / s0, z1 = (mu1 - l1) / s1, zl = (mu0 - l1) / s1;
    double U = 0.5 * z0 * z0 + 0.5 * z1 * z1 + log(0.5 * erfc(zl * 0.70710678118654752)) - x1;
    for (int f = 0; f < 2; f++) {
        const double a = m.cs_sg[2 * f], b = m.cs_sg[2 * f + 1], ls = z[m.G0 + 2 + f];
        U += -(a - 1.0) * ls + b * exp(ls) - ls;
    }
    return U; // (constants: m.u_const)
}

// Potential gradient of a fixed effect / log sd coordinate d < G at position z, from the reduced sums of the site pass
// (red[0 .. 2MK+2]) and of the effects' squares (red[OX] = sum u^2 + v^2, red[OX+1] = sum e^2).
template <int MK, bool FP = false>
__device__ __forceinline__ float bl_re_global_grad(const BlReModel &m, int d, float zd, const double *red, const double *red_sp = nullptr,
                                                   int nred = 0)
{
    constexpr int OA = BL_RE_OA(MK), OX = BL_RE_OX(MK);
    if (d < m.G0) {
        // (several species: red_sp holds each species' own gradient sums, NRED apart; one species: red itself)
        const int sd = d / m.G0s, j = d - sd * m.G0s;
        const bool is_b = j <= m.Ks;
        const float loc = is_b ? m.loc_b : m.loc_a, isc2 = is_b ? m.isc2_b : m.isc2_a, l1 = is_b ? m.l1_b : m.l1_a;
        const double gl = (m.n_species > 1 ? red_sp + sd * nred : red)[is_b ? 1 + j : OA + (j - m.Ks - 1)];
        const float dth = zd - loc;
        return (float)(-gl) + fmaf(dth, isc2, dth > 0.0f ? l1 : (dth < 0.0f ? -l1 : 0.0f));
    }
    if (m.kind == 1) return 0.0f; // (occu_cs: its four extra coordinates are handled by bl_cs_extra_grad)
    if (FP && d == m.o_fp && m.kind == 7) // phi = log f, f ~ Exponential(rate = fp_a), Jacobian included: energy rate e^phi - phi
        return (float)(-red[OX + 6]) + m.fp_a * bl_exp(fminf(zd, 80.0f)) - 1.0f;
    if (FP && d == m.o_fp) {
        // phi = logit f, f ~ Beta(a, b), Jacobian included: energy a softplus(-phi) + b softplus(phi); red[OX + 6] = d ll / d phi
        const float e = bl_exp(-fabsf(zd)), sig = (zd > 0.0f ? 1.0f : e) * bl_rcp(1.0f + e);
        return (float)(-red[OX + 6]) + (m.fp_a + m.fp_b) * sig - m.fp_a;
    }
    const bool site = m.site_re && d == m.o_phi_s;
    const float isd2 = bl_exp(-2.0f * zd), sd2 = bl_exp(2.0f * zd);
    const float cnt = (float)m.n_species * (site ? 2.0f * (float)m.n_total : (float)m.n_total * (float)(m.T * m.J));
    const float ssq = (float)red[site ? OX : OX + 1];
    // U = sd^2 / (2 s^2) - phi + sum_k (x_k^2 / (2 sd^2) + phi):   dU/dphi = sd^2 / s^2 - 1 - ssq / sd^2 + cnt
    return sd2 * (site ? m.hn_is2_s : m.hn_is2_o) - 1.0f - ssq * isd2 + cnt;
}

// Potential at z from the reduced sums (f64): red[0] = log-lik, pe2 = sum over fixed effects of ((z - loc) / scale)^2
template <bool FP = false>
__device__ __forceinline__ double bl_re_potential(const BlReModel &m, const float *z, const double *red, double pe2, int OX)
{
    double U = -red[0] + 0.5 * pe2 + m.u_const;
    if constexpr (FP) {
        const double phi = z[m.o_fp], l = log1p(exp(-fabs(phi)));
        if (m.kind == 7) U += (double)m.fp_a * exp(fmin(phi, 80.0)) - phi; // (Exponential prior of occu_cop's rate; - log rate is in u_const)
        else U += m.fp_a * (fmax(-phi, 0.0) + l) + m.fp_b * (fmax(phi, 0.0) + l);
    }
    if (m.site_re) {
        const float phi = z[m.o_phi_s];
        U += 0.5 * (double)(bl_exp(2.0f * phi) * m.hn_is2_s) - (double)phi + 0.5 * red[OX] * (double)bl_exp(-2.0f * phi) + 2.0 * m.n_species * m.n_total * (double)phi;
    }
    if (m.obs_re) {
        const float phi = z[m.o_phi_o];
        U += 0.5 * (double)(bl_exp(2.0f * phi) * m.hn_is2_o) - (double)phi + 0.5 * red[OX + 1] * (double)bl_exp(-2.0f * phi)
             + (double)m.n_species * m.n_total * (m.T * m.J) * (double)phi;
    }
    return U;
}

// sum of squares of the random effects owned by this thread at position z -> ss[0] (site), ss[1] (obs)
__device__ __forceinline__ void bl_re_effect_squares(const BlReModel &m, const float *z, float (&ss)[2])
{
    ss[0] = 0.0f; ss[1] = 0.0f;
    for (int d = m.G + threadIdx.x; d < m.D; d += BL_RE_NT) {
        const float x = z[d];
        if (m.obs_re && d >= m.o_e) ss[1] = fmaf(x, x, ss[1]);
        else ss[0] = fmaf(x, x, ss[0]);
    }
}

// this thread's share of twice the fixed effects' prior energy: ((z - loc) / scale)^2 (Normal) or 2 |z - loc| / scale (Laplace)
__device__ __forceinline__ float bl_re_prior_quad(const BlReModel &m, const float *z)
{
    float pe = 0.0f;
    for (int d = threadIdx.x; d < m.G0; d += BL_RE_NT) {
        const bool is_b = d % m.G0s <= m.Ks;
        const float t = z[d] - (is_b ? m.loc_b : m.loc_a);
        pe = fmaf(t * t, is_b ? m.isc2_b : m.isc2_a, fmaf(2.0f * fabsf(t), is_b ? m.l1_b : m.l1_a, pe));
    }
    return pe;
}

// ---- parity hook: U and dU/dtheta for B positions (external order in, external order out), one workgroup each ----
// Several species: the workgroup takes them one after the other (a species' slice = all sites, its own local vectors in `work`).
template <int MK>
__global__ void __launch_bounds__(BL_RE_NT) bl_re_logp_kernel(const BlReModel gm, int B, const float *__restrict__ theta,
                                                              float *__restrict__ work /*[B][2][D of one species' slice]*/,
                                                              double *__restrict__ U, double *__restrict__ grad)
{
    constexpr int NRED = BL_RE_NRED_OF(MK), OX = BL_RE_OX(MK), NV1 = BL_RE_NV1(MK);
    __shared__ float scr[BL_RE_NW * NRED];
    __shared__ double red[NRED], red_tot[NRED], red_sp[BL_RE_SMAX * NRED];
    const int b = blockIdx.x, tid = threadIdx.x, S = gm.n_species;
    if (b >= B) return;
    extern __shared__ float bl_re_lds[];
    const bool stage = gm.lds_rows && S == 1;
    if (tid < NRED) red_tot[tid] = 0.0;
    for (int sp = 0; sp < S; sp++) {
        const BlReModel m = bl_re_slice(gm, sp, 0, gm.n_sites);
        const int D = m.D;
        float *z = work + (size_t)b * 2 * D, *g = z + D;
        __syncthreads(); // (the previous species' vectors are done with)
        for (int d = tid; d < D; d += BL_RE_NT) z[d] = theta[(size_t)b * gm.D + bl_re_ext(m, d)];
        int ns = m.n_stride, rv = m.rv0;
        const float *rows = m.rows;
        if (stage) { rows = bl_re_rows(m, bl_re_lds, ns); rv = m.KS; }
        __syncthreads();
        float v[NV1];
        if (m.kind == 1) {
            float part[2 * MK + 7];
            bl_cs_site_pass<MK>(m, rows, ns, z, part);
#pragma unroll
            for (int k = 0; k < OX; k++) v[k] = part[k];
            v[OX] = part[OX]; v[OX + 1] = part[OX + 1]; v[OX + 6] = part[OX + 2]; v[OX + 7] = part[OX + 3];
        } else {
            float part[2 * MK + 3], ss[2], gphi = 0.0f;
            if (m.kind == 2) bl_re_site_pass<MK, true>(m, rows, ns, rv, z, g, part, &gphi);
            else if (m.kind == 3) bl_nmix_re_site_pass<MK>(m, rows, ns, rv, z, g, part);
            else if (m.kind == 4) bl_rn_re_site_pass<MK>(m, rows, ns, rv, z, g, part);
            else if (m.kind == 5) bl_rn_re_site_pass<MK, true>(m, rows, ns, rv, z, g, part, &gphi);
            else if (m.kind == 6) bl_cop_re_site_pass<MK>(m, rows, ns, rv, z, g, part);
            else if (m.kind == 7) bl_cop_re_site_pass<MK, true>(m, rows, ns, rv, z, g, part, &gphi);
            else bl_re_site_pass<MK>(m, rows, ns, rv, z, g, part);
            bl_re_effect_squares(m, z, ss);
#pragma unroll
            for (int k = 0; k < OX; k++) v[k] = part[k];
            v[OX] = ss[0]; v[OX + 1] = ss[1]; v[OX + 6] = gphi; v[OX + 7] = 0.0f;
        }
        v[OX + 2] = sp == 0 ? bl_re_prior_quad(m, z) : 0.0f;
        v[OX + 3] = 0.0f; v[OX + 4] = 0.0f; v[OX + 5] = 0.0f;
        bl_re_block_sum<NV1, NRED>(v, scr, red);
        if (tid < NV1) { red_sp[sp * NRED + tid] = red[tid]; red_tot[tid] += red[tid]; }
        __syncthreads();
        for (int d = m.G + tid; d < D; d += BL_RE_NT) grad[(size_t)b * gm.D + bl_re_ext(m, d)] = (double)g[d];
        if (sp == S - 1) { // fixed effects of every species and the log sds, from the species' and the total sums
            for (int d = tid; d < m.G; d += BL_RE_NT)
                grad[(size_t)b * gm.D + d] = (double)((m.kind == 1 && d >= m.G0) ? bl_cs_extra_grad<MK>(m, d - m.G0, z, red_tot)
                                                      : ((m.kind == 2 || m.kind == 5 || m.kind == 7) ? bl_re_global_grad<MK, true>(m, d, z[d], red_tot, red_sp, NRED)
                                                                     : bl_re_global_grad<MK>(m, d, z[d], red_tot, red_sp, NRED)));
            if (tid == 0) U[b] = ((m.kind == 2 || m.kind == 5 || m.kind == 7) ? bl_re_potential<true>(m, z, red_tot, red_tot[OX + 2], OX) : bl_re_potential(m, z, red_tot, red_tot[OX + 2], OX))
                                 + (m.kind == 1 ? bl_cs_extra_potential(m, z) : 0.0);
        }
    }
}

// ------------------------------------------------------------------------------------------------ NUTS ----
// KIND: BlReModel::kind; LROWS / LT: BlReModel::lds_rows / lds_hot as compile-time facts, so that every access to the rows and to
// the sampler's vectors is a DS or a GLOBAL instruction (one generic pointer for both made all of them FLAT: 272 flat loads, 230 spilled
// SGPRs of 64-bit bases)
// EFF (round 4, the bench form only): which effects the model has, as a compile-time fact -- 0: read from the model at run time (the
// general kernel); 1: site effects only, one species; 2: observation effects only, one species; 3: both, one species; + 4: one period.  The workgroup's
// local copy of the model gets those fields as CONSTANTS, and every `if (m.obs_re)` / species loop behind them folds away (what a kernel
// merely carries costs the rest: profiles/NOTES.md).
template <int MK, int KIND, bool LROWS, int LT, int EFF = 0>
__global__ void __launch_bounds__(BL_RE_NT) bl_re_nuts_kernel(const BlReRun R)
{
    constexpr int NRED = BL_RE_NRED_OF(MK), OX = BL_RE_OX(MK), NV1 = BL_RE_NV1(MK);
    __shared__ float scr[BL_RE_NW * NRED];
    __shared__ double red[NRED], red2[NRED], red_sp[BL_RE_SMAX * NRED];
    __shared__ float scr2[32 * NRED];
    __shared__ int xflag;
    // XCD-aware mapping (speed only, as in nuts_kernel.hpp): blocks b and b + 8 share an XCD under the observed round-robin
    // dealing, so chain c takes blocks with b % 8 == c % 8 and its k workgroups share one L2; the first exchange checks it
    const int label = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int chain = label + 8 * (slot / R.k), wg = slot % R.k, tid = threadIdx.x;
    if (chain >= R.num_chains) return;
    // several species: the chain's k workgroups are n_species groups of kps, each group slicing the sites of its species
    const int kps = R.k / R.m.n_species, sp = wg / kps, s0 = (wg - sp * kps) * R.nloc;
    BlReModel m_ = bl_re_slice(R.m, sp, s0, min(R.nloc, R.m.n_sites - s0)); // this workgroup's slice as a model of its own
    if constexpr (EFF != 0) { m_.site_re = (EFF & 1) ? 1 : 0; m_.obs_re = (EFF & 2) ? 1 : 0; m_.n_species = 1; m_.sp = 0; }
    if constexpr ((EFF & 4) != 0) m_.T = 1; // one period (the shape of simulate()'s defaults and of the bench)
    const BlReModel &m = m_;
    const int D = m.D, G = m.G;
    const bool lead = wg == 0; // the fixed effects / log sds are replicated; workgroup 0 accounts for them in every sum and output
    float *sv = R.state + ((size_t)chain * R.k + wg) * RE_SLOTS * R.dl_max;
    extern __shared__ float bl_re_lds[];
    float *const hot_lds = bl_re_lds + (LROWS ? m.n_rows * R.nloc : 0);
    // the RE_HOT vectors (H), the warm ones up to RE_WARM (Wm), the others (V)
    auto H = [&](int slot) -> float * { if constexpr (LT >= 1) return hot_lds + slot * R.dl_max; else return sv + (size_t)slot * R.dl_max; };
    auto Wm = [&](int slot) -> float * { if constexpr (LT >= 2) return hot_lds + slot * R.dl_max; else return sv + (size_t)slot * R.dl_max; };
    auto V = [&](int slot) -> float * { return sv + (size_t)slot * R.dl_max; };
    uint32_t *rng_base = R.rng + ((size_t)chain * R.k + wg) * (R.dl_max + 2) * 4;
    BlReXchg xc{R.xchg, R.k, wg, chain, R.m.n_species, 0u, 5000000u /* microseconds */, false};
    const float xcc = (float)bl_xcc_id();
    if (tid == 0) xflag = 0;
    const int S = R.num_samples, W = R.num_warmup, total = W + S;
    int rows_ns = m.n_stride, rows_rv = m.rv0;
    const float *rows = m.rows;
    if constexpr (LROWS) { // the workgroup's own copy of its rows: the site covariates, then its species' block
        const int N = m.n_sites;
        for (int r = 0; r < m.n_rows; r++) {
            const int rg = r < m.KS ? r : m.rv0 + (r - m.KS);
            for (int i = tid; i < N; i += BL_RE_NT) bl_re_lds[r * N + i] = m.rows[(size_t)rg * m.n_stride + i];
        }
        __syncthreads();
        rows = bl_re_lds; rows_ns = N; rows_rv = m.KS;
    }

    BlRng rng_u, rng_dir; // every thread carries its own copy of the two scalar streams and advances it identically
    {
        const uint32_t *a = rng_base + (size_t)R.dl_max * 4, *b = a + 4;
        rng_u.s0 = a[0]; rng_u.s1 = a[1]; rng_u.s2 = a[2]; rng_u.s3 = a[3];
        rng_dir.s0 = b[0]; rng_dir.s1 = b[1]; rng_dir.s2 = b[2]; rng_dir.s3 = b[3];
    }
    auto rng_load = [&](int d) { const uint32_t *s = rng_base + (size_t)d * 4; BlRng r; r.s0 = s[0]; r.s1 = s[1]; r.s2 = s[2]; r.s3 = s[3]; return r; };
    auto rng_store = [&](int d, const BlRng &r) { uint32_t *s = rng_base + (size_t)d * 4; s[0] = r.s0; s[1] = r.s1; s[2] = r.s2; s[3] = r.s3; };

    // sampler scalars (identical in every thread)
    float eps = 1.0f, epsdir = 1.0f;
    int depth = 0, snprop = 0, it = 0;
    bool sturn = false, sdiv = false, going_right = false;
    double E0 = 0.0, Ucur = 0.0, sUp = 0.0, Up = 0.0;
    float swt = 0.f, ssumacc = 0.f, wt = 0.f, sumacc = 0.f;
    int nprop = 0;
    float da_prox = 2.302585093f, da_gavg = 0.f, da_xt = 0.f, da_xavg = 0.f; // log(10 * step_size0)
    int da_t = 0, win_idx = 0, wf_n = 0;
    long long nleap_w = 0, nleap_s = 0;
    int flag = 0;
    float abort_req = 0.0f; // (non-zero only in thread 0 of workgroup 0)

    // evaluate the potential and its gradient at H(RE_CZ) into H(RE_CG); returns U (same value in every thread).
    // The gradient of a fixed effect / log sd is written by the thread that owns the coordinate, after the reduction.
#ifdef BL_STAMPS
    long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = (long long)clock64(), st_leaves = 0;
#endif
    // evaluate_a: site pass at H(RE_CZ) -- the random effects' gradients are complete after it -- and the workgroup's sums published
    // to the chain; evaluate_b: the chain's sums collected, the gradients of the fixed effects / log sds written by the threads
    // that own those coordinates; returns U (same value in every thread).
    bool ev_first = false;
    int ev_nv = 0;
    auto evaluate_a = [&]() {
        float v[NV1];
        const float *z = H(RE_CZ);
        float *g = H(RE_CG);
        BL_RE_T(7)
        if constexpr (KIND == 1) { // occu_cs: no effects; four more gradient sums
            float part[2 * MK + 7];
            bl_cs_site_pass<MK>(m, rows, rows_ns, z, part);
#pragma unroll
            for (int k = 0; k < OX; k++) v[k] = part[k];
            v[OX] = part[OX]; v[OX + 1] = part[OX + 1]; v[OX + 6] = part[OX + 2]; v[OX + 7] = part[OX + 3];
        } else {
            float part[2 * MK + 3], ss[2], gphi = 0.0f;
            if constexpr (KIND == 2) bl_re_site_pass<MK, true>(m, rows, rows_ns, rows_rv, z, g, part, &gphi);
            else if constexpr (KIND == 3) bl_nmix_re_site_pass<MK>(m, rows, rows_ns, rows_rv, z, g, part);
            else if constexpr (KIND == 4) bl_rn_re_site_pass<MK>(m, rows, rows_ns, rows_rv, z, g, part);
            else if constexpr (KIND == 5) bl_rn_re_site_pass<MK, true>(m, rows, rows_ns, rows_rv, z, g, part, &gphi);
            else if constexpr (KIND == 6) bl_cop_re_site_pass<MK>(m, rows, rows_ns, rows_rv, z, g, part);
            else if constexpr (KIND == 7) bl_cop_re_site_pass<MK, true>(m, rows, rows_ns, rows_rv, z, g, part, &gphi);
            else bl_re_site_pass<MK>(m, rows, rows_ns, rows_rv, z, g, part);
            BL_RE_T(8)
            bl_re_effect_squares(m, z, ss);
#pragma unroll
            for (int k = 0; k < OX; k++) v[k] = part[k];
            v[OX] = ss[0]; v[OX + 1] = ss[1]; v[OX + 6] = gphi; v[OX + 7] = 0.0f;
        }
        v[OX + 2] = lead ? bl_re_prior_quad(m, z) : 0.0f;
        // the host's abort request rides in the sums, so that every workgroup of the chain sees it at the same leapfrog
        v[OX + 3] = abort_req;
        // XCD census (first exchange): k sum(x^2) == (sum x)^2 iff every workgroup reports the same XCC id
        v[OX + 4] = tid == 0 ? xcc : 0.0f; v[OX + 5] = tid == 0 ? xcc * xcc : 0.0f;
        BL_RE_T(0)
        ev_first = xc.epoch == 0u;
        ev_nv = KIND == 1 ? NV1 : ((KIND == 2 || KIND == 5 || KIND == 7) ? OX + 7 : (ev_first ? OX + 6 : OX + 4));
        bl_re_block_sum<NV1, NRED>(v, scr, red, ev_nv, xc.k == 1);
        BL_RE_T(9)
        bl_re_publish<NRED>(xc, red, ev_nv);
    };
    auto evaluate_b = [&]() -> double {
        const float *z = H(RE_CZ);
        float *g = H(RE_CG);
        // (several species: per-species sums by one poll per species up to three species, else the general form)
        if (!(m.n_species <= 3 ? bl_re_collect_wave<NRED>(xc, red, &xflag, ev_nv, m.n_species > 1 ? red_sp : nullptr)
                               : bl_re_collect<NRED>(xc, red, scr2, &xflag, ev_nv, red_sp))) flag = 4;
        BL_RE_T(10)
        if (ev_first && R.allow_local) xc.local = ((double)R.k * red[OX + 5] == red[OX + 4] * red[OX + 4]); // exact: small integers
        if (red[OX + 3] > 0.0) flag = 5;
        for (int d = tid; d < G; d += BL_RE_NT)
            g[d] = (KIND == 1 && d >= m.G0) ? bl_cs_extra_grad<MK>(m, d - m.G0, z, red) : bl_re_global_grad<MK, KIND == 2 || KIND == 5 || KIND == 7>(m, d, z[d], red, red_sp, NRED);
        double U = bl_re_potential<KIND == 2 || KIND == 5 || KIND == 7>(m, z, red, red[OX + 2], OX);
        if constexpr (KIND == 1) U += bl_cs_extra_potential(m, z);
        BL_RE_T(1)
        return U;
    };

    // momentum r ~ N(0, M), fresh tree, first doubling, and the first leaf's half step (start in CZ / CR / CG)
    auto new_transition = [&]() {
        // the abort flag is host memory (a PCIe round trip per read): sampled once per transition, by one thread, and consumed by
        // the next leaf's sums
        if (lead && tid == 0 && R.abort_flag) abort_req = *(volatile const int *)R.abort_flag ? 1.0f : 0.0f;
        going_right = (bl_rng_next(rng_dir) >> 31) != 0u;
        epsdir = going_right ? eps : -eps;
        float kin[1] = {0.0f};
        for (int d = tid; d < D; d += BL_RE_NT) {
            const float th = V(RE_TH)[d], gr = V(RE_GR)[d], mi = H(RE_MINV)[d];
            BlRng r = rng_load(d);
            const float z01 = bl_rng_normal(r);
            rng_store(d, r);
            const float r0 = z01 * __builtin_amdgcn_rsqf(mi);
            if (d >= G || lead) kin[0] = fmaf(mi * r0, r0, kin[0]);
            V(RE_ZL)[d] = th; Wm(RE_RL)[d] = r0; V(RE_GL)[d] = gr;
            V(RE_ZR)[d] = th; Wm(RE_RR)[d] = r0; V(RE_GRR)[d] = gr;
            V(RE_ZP)[d] = th; V(RE_GP)[d] = gr; Wm(RE_RSUM)[d] = r0;
            float rh, zn;
            bl_next_leaf(th, r0, gr, epsdir, mi, rh, zn);
            H(RE_CZ)[d] = zn; H(RE_CR)[d] = rh; H(RE_CG)[d] = gr;
        }
        bl_re_block_sum<1, NRED>(kin, scr, red2, 1, xc.k == 1); // (its barriers, or the exchange's, also publish the leaf start)
        if (!bl_re_exchange<NRED>(xc, red2, scr2, &xflag, 1)) flag = 4;
        E0 = Ucur + 0.5 * red2[0];
        Up = Ucur; wt = 0.f; sumacc = 0.f; nprop = 0; depth = 0;
        snprop = 0; sturn = false; sdiv = false;
    };

    // ---- initial position: init_to_uniform(radius = 2) from each coordinate's own stream (always consumed) ----
    for (int d = tid; d < D; d += BL_RE_NT) {
        BlRng r = rng_load(d);
        const float u0 = bl_rng_uniform(r);
        rng_store(d, r);
        H(RE_CZ)[d] = R.init_theta ? R.init_theta[(size_t)chain * R.m.D + bl_re_ext(m, d)] : 4.0f * u0 - 2.0f;
        H(RE_MINV)[d] = 1.0f; V(RE_WFMEAN)[d] = 0.0f; V(RE_WFM2)[d] = 0.0f;
    }
    __syncthreads();
    evaluate_a();
    Ucur = evaluate_b();
    __syncthreads(); // every coordinate's gradient is in place
    for (int d = tid; d < D; d += BL_RE_NT) { V(RE_TH)[d] = H(RE_CZ)[d]; V(RE_GR)[d] = H(RE_CG)[d]; }
    if (total <= 0) flag = 1;
    else new_transition();

    while (flag == 0) {
        // ---- the leaf's position is in CZ, its half-step momentum in CR (hmc_util velocity Verlet) ----
        evaluate_a();
        if (it < W) nleap_w++; else nleap_s++;
        // ---- second half step; kinetic energy; subtree momentum sum; U-turn dot products for every checkpoint this
        //      leaf closes (_leaf_idx_to_ckpt_idxs) and for the whole tree in case the subtree ends here ----
        const int leaf_idx = snprop;
        const int idx_max = __popc((unsigned)leaf_idx >> 1);
        const int idx_min = idx_max - (int)__builtin_ctz(~(unsigned)leaf_idx) + 1;
        const bool odd = (leaf_idx & 1) != 0;
        float acc[26]; // 0: kinetic; 1, 2: tree; 3 + 2 q, 4 + 2 q: checkpoint idx_min + q
#pragma unroll
        for (int k = 0; k < 26; k++) acc[k] = 0.0f;
        const int nck = odd ? idx_max - idx_min + 1 : 0;
        // One pass costs its dependent latency (~1.4 k cycles) whatever the coordinate count, so every load is issued before the
        // first store (the compiler must assume the vectors alias and will not move a load above a store), and the checkpoint
        // loop is compiled for QN = 0 (even leaf), 1, 2, 4 or all checkpoints, loaded unconditionally (index clamped, weight 0).
        auto second_half = [&](auto qn_, int d) {
            constexpr int QN = decltype(qn_)::value;
            const float mi = H(RE_MINV)[d], cr0 = H(RE_CR)[d], cg = H(RE_CG)[d], srs0 = H(RE_SRSUM)[d];
            const float r_other = Wm(going_right ? RE_RL : RE_RR)[d], rsum = Wm(RE_RSUM)[d];
            float ck[QN > 0 ? QN : 1], cs[QN > 0 ? QN : 1];
#pragma unroll
            for (int q = 0; q < QN; q++) {
                const int i = min(idx_min + q, idx_max);
                ck[q] = Wm(RE_CKR + i)[d];
                cs[q] = Wm(RE_CKR + BL_MAX_DEPTH + i)[d];
            }
            const float mc = (d >= G || lead) ? mi : 0.0f; // weight of this coordinate in the chain-wide sums
            const float cr = bl_leaf_momentum(cr0, epsdir, cg);
            acc[0] = fmaf(mc * cr, cr, acc[0]);
            const float srs = leaf_idx == 0 ? cr : srs0 + cr;
#pragma unroll
            for (int q = 0; q < QN; q++) {
                const float mq = idx_min + q <= idx_max ? mc : 0.0f;
                const float s_i = srs - cs[q] + ck[q];
                const float rho = s_i - 0.5f * (ck[q] + cr);
                acc[3 + 2 * q] = fmaf(mq * ck[q], rho, acc[3 + 2 * q]);
                acc[4 + 2 * q] = fmaf(mq * cr, rho, acc[4 + 2 * q]);
            }
            const float rl = going_right ? r_other : cr, rr = going_right ? cr : r_other;
            const float rho_t = (rsum + srs) - 0.5f * (rl + rr);
            acc[1] = fmaf(mc * rl, rho_t, acc[1]);
            acc[2] = fmaf(mc * rr, rho_t, acc[2]);
            H(RE_CR)[d] = cr;
            H(RE_SRSUM)[d] = srs;
            if (QN == 0) { // even leaf: it opens checkpoint idx_max
                Wm(RE_CKR + idx_max)[d] = cr;
                Wm(RE_CKR + BL_MAX_DEPTH + idx_max)[d] = srs;
            }
        };
        // (Measured and dropped: two coordinates per thread and pass for slices of more than NT coordinates, both loaded before either
        // is stored, here and in the vector loop after the decisions: 10 % slower at 634 and 764 coordinates per slice.)
        auto second_half_all = [&](auto qn_) {
            for (int d = tid; d < D; d += BL_RE_NT) second_half(qn_, d);
        };
        // (Measured and dropped: running this loop for the random effects between evaluate_a and evaluate_b -- their gradients are
        // complete after the site pass -- to overlap the exchange's hand-off.  7 % SLOWER: collecting granules that are already
        // there still costs one L2 round trip + the sums (~2 k of the exchange's 2.4 k cycles), while the extra pass of this loop
        // for the handful of fixed effects costs its full latency again, ~1.4 k.)
        // (Measured and dropped, round 2: TWO exchanges in flight -- the second exchange carrying the random effects' dot products only,
        // published right behind the first, and the first wave adding the replicated coordinates' share, identical in every workgroup,
        // after collecting both: one hand-off and one block sum less on the critical path, but the extra pass of the first wave over
        // its replicated coordinates and its in-wave reduction cost more: 4-7 % slower, bench line 9.8 -> 10.3 us.  And its
        // leaner form -- ONE block sum for both payloads, the two records published together and collected by the first two waves
        // side by side, four barriers per leapfrog instead of five: -6 ... +3 % over the five shapes, -1 % on the bench line, i.e. even.)
        const double Un = evaluate_b();
        if (!odd) second_half_all(std::integral_constant<int, 0>{});
        else if (nck <= 1) second_half_all(std::integral_constant<int, 1>{});
        else if (nck <= 2) second_half_all(std::integral_constant<int, 2>{});
        else if (nck <= 4) second_half_all(std::integral_constant<int, 4>{});
        else second_half_all(std::integral_constant<int, BL_MAX_DEPTH>{});
        BL_RE_T(2)
        bl_re_block_sum<26, NRED>(acc, scr, red2, 3 + 2 * nck, xc.k == 1);
        BL_RE_T(11)
        if (!bl_re_exchange<NRED>(xc, red2, scr2, &xflag, 3 + 2 * nck)) flag = 4;
        BL_RE_T(3)
        // ---- decisions (_build_basetree tail, _iterative_build_subtree, _combine_tree, _double_tree) ----
        double dE = (Un + 0.5 * red2[0]) - E0;
        if (dE != dE) dE = (double)INFINITY;
        sdiv = dE > 1000.0;
        snprop = leaf_idx + 1;
        const float fdE = (float)dE;
        const float lw = -fdE, lacc = fdE <= 0.0f ? 1.0f : bl_exp(-fdE);
        bool take = true; // this leaf becomes the subtree's proposal
        if (leaf_idx == 0) { swt = lw; ssumacc = lacc; }
        else {
            float pr, lse;
            bl_merge_weights(swt, lw, lse, pr);
            const float u = bl_rng_uniform(rng_u);
            take = u < pr;
            swt = lse; ssumacc += lacc;
        }
        if (take) sUp = Un;
        if (odd) {
#pragma unroll
            for (int q = 0; q < BL_MAX_DEPTH; q++)
                if (idx_min + q <= idx_max) sturn = sturn || red2[3 + 2 * q] <= 0.0 || red2[4 + 2 * q] <= 0.0;
        }
        const bool sub_done = !(snprop < (1 << depth) && !sturn && !sdiv);
        bool take2 = false, cont = false, trans_end = false;
        const bool was_right = going_right;
        int nprop_out = 0; float accp = 0.f;
        if (sub_done) {
            const bool turning = sturn || red2[1] <= 0.0 || red2[2] <= 0.0;
            float pr = fminf(1.0f, bl_exp(swt - wt));
            if (sturn || sdiv) pr = 0.0f;
            const float u = bl_rng_uniform(rng_u);
            take2 = u < pr;
            if (take2) Up = sUp;
            wt = bl_logaddexp(wt, swt);
            sumacc += ssumacc;
            nprop += snprop;
            depth++;
            cont = depth < R.max_depth && !turning && !sdiv;
            trans_end = !cont;
            if (cont) {
                going_right = (bl_rng_next(rng_dir) >> 31) != 0u;
                epsdir = going_right ? eps : -eps;
                snprop = 0; sturn = false;
            }
        }
        // ---- transition end: scalar part of the warmup adapter (hmc_util.warmup_adapter.update_fn) ----
        bool wf_update = false, wf_close = false;
        const int it_done = it;
        const bool div_out = sdiv;
        if (trans_end) {
            nprop_out = nprop;
            accp = sumacc * bl_rcp((float)nprop);
            Ucur = Up;
            if (it < W) {
                const float gdiff = R.target_accept - accp;
                da_t += 1;
                const float tt = (float)da_t, rt10 = bl_rcp(tt + 10.0f);
                da_gavg = (1.0f - rt10) * da_gavg + gdiff * rt10;
                da_xt = da_prox - __builtin_amdgcn_sqrtf(tt) * 20.0f * da_gavg;
                const float wgt = __builtin_amdgcn_exp2f(-0.75f * __builtin_amdgcn_logf(tt));
                da_xavg = (1.0f - wgt) * da_xavg + wgt * da_xt;
                eps = bl_exp((it == W - 1) ? da_xavg : da_xt);
                eps = fminf(fmaxf(eps, 1.1754944e-38f), 3.4028235e+38f);
                const bool middle = win_idx > 0 && win_idx < R.nwin - 1;
                if (middle) { wf_n += 1; wf_update = true; }
                const bool at_end = it == R.win_end[win_idx];
                if (at_end) win_idx++;
                wf_close = at_end && middle;
            }
            it++;
            if (it >= total && flag == 0) flag = 1;
        }
        // ---- vector part of the same decisions, and -- unless the transition ends -- the next leaf's half step ----
        BL_RE_T(4)
        const float wfn = (float)wf_n;
        for (int d = tid; d < D; d += BL_RE_NT) {
            const float cz = H(RE_CZ)[d], cg = H(RE_CG)[d], cr = H(RE_CR)[d], mi = H(RE_MINV)[d]; // (loads before the first store)
            if (take) { Wm(RE_SZP)[d] = cz; Wm(RE_SGP)[d] = cg; }
            if (!sub_done) { // the subtree goes on from this leaf
                float rh, zn;
                bl_next_leaf(cz, cr, cg, epsdir, mi, rh, zn);
                H(RE_CZ)[d] = zn; H(RE_CR)[d] = rh;
                continue;
            }
            V(was_right ? RE_ZR : RE_ZL)[d] = cz; Wm(was_right ? RE_RR : RE_RL)[d] = cr; V(was_right ? RE_GRR : RE_GL)[d] = cg;
            Wm(RE_RSUM)[d] += H(RE_SRSUM)[d];
            if (take2) { V(RE_ZP)[d] = take ? cz : Wm(RE_SZP)[d]; V(RE_GP)[d] = take ? cg : Wm(RE_SGP)[d]; }
            if (cont) { // next doubling: its first leaf starts from the tree edge on the chosen side
                const float ez = V(going_right ? RE_ZR : RE_ZL)[d], er = Wm(going_right ? RE_RR : RE_RL)[d], eg = V(going_right ? RE_GRR : RE_GL)[d];
                float rh, zn;
                bl_next_leaf(ez, er, eg, epsdir, mi, rh, zn);
                H(RE_CZ)[d] = zn; H(RE_CR)[d] = rh; H(RE_CG)[d] = eg;
                continue;
            }
            const float th = V(RE_ZP)[d];
            V(RE_TH)[d] = th; V(RE_GR)[d] = V(RE_GP)[d];
            if (wf_update) {
                const float mean0 = V(RE_WFMEAN)[d], dpre = th - mean0;
                const float mean = mean0 + dpre * bl_rcp(wfn);
                V(RE_WFMEAN)[d] = mean;
                V(RE_WFM2)[d] += dpre * (th - mean);
            }
            if (wf_close) {
                const float var = V(RE_WFM2)[d] * bl_rcp(wfn - 1.0f), rn5 = bl_rcp(wfn + 5.0f);
                H(RE_MINV)[d] = wfn * rn5 * var + 1e-3f * 5.0f * rn5;
                V(RE_WFMEAN)[d] = 0.0f; V(RE_WFM2)[d] = 0.0f;
            }
            if (it_done >= W && (d >= G || lead)) R.draws[((size_t)chain * S + (it_done - W)) * R.m.D + bl_re_ext(m, d)] = th;
        }
        if (trans_end) {
            if (wf_close) {
                wf_n = 0; da_xt = 0.f; da_xavg = 0.f; da_gavg = 0.f; da_t = 0;
                da_prox = bl_log(10.0f * eps);
            }
            if (it_done >= W && tid == 0 && lead) {
                const size_t s = (size_t)chain * S + (it_done - W);
                R.num_steps[s] = nprop_out; R.accept_prob[s] = accp;
                R.diverging[s] = div_out ? 1 : 0; R.potential[s] = (float)Ucur;
            }
            BL_RE_T(5)
            if (flag == 0) new_transition(); // (own coordinates only: TH / GR / MINV were written by this thread above)
            BL_RE_T(6)
        }
        BL_RE_T(5)
        __syncthreads(); // the next leaf's position is visible to the site pass
#ifdef BL_STAMPS
        st_leaves++;
#endif
    }
#ifdef BL_STAMPS
    if (R.dbg && chain == 0 && tid == 0 && lead) {
        for (int i = 0; i < 16; i++) R.dbg[i] = st_acc[i];
        R.dbg[16] = st_leaves;
    }
#endif
    for (int d = tid; d < D; d += BL_RE_NT)
        if (d >= G || lead) R.inv_mass[(size_t)chain * R.m.D + bl_re_ext(m, d)] = H(RE_MINV)[d];
    if (tid == 0 && flag > 1) atomicMax(R.status, flag);
    if (tid == 0 && lead) {
        if (R.xcd_local) R.xcd_local[chain] = xc.local ? 1 : 0;
        R.step_size[chain] = eps;
        R.nleap[chain * 2 + 0] = nleap_w; R.nleap[chain * 2 + 1] = nleap_s;
    }
}
