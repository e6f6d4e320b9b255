// nuts_kernel.hpp -- persistent multi-CU NUTS kernel (one launch = all chains, warmup + sampling).
//
// Replaces numpyro.infer.MCMC(NUTS(occu)).run(...) as biolith/utils/fit.py:92-130 invokes it.
// The sampler follows NumPyro's algorithm: iterative tree doubling with
// checkpointed generalised U-turn checks, multinomial proposals (uniform inside a subtree, biased
// between subtrees), dual-averaging step size, windowed Welford diagonal mass matrix.
//
// Mapping to the chip
//   * one chain = k cooperating workgroups, one per CU; a workgroup = CW compute waves + 1 control
//     wave (CW = 3: every wave owns a SIMD; CW = 4 for larger slices, occu_device.hpp).  It owns a
//     contiguous slice of sites, staged ONCE into LDS as pair records and evaluated two sites per
//     lane (occu_rn: one);
//   * blockIdx -> (chain, member) is XCD-aware: blocks b and b+8 share an XCD under the observed
//     round-robin dealing, so chain c takes the blocks with b % 8 == c % 8 and its k workgroups
//     share one L2.  That is a SPEED arrangement only: every workgroup reads HW_REG_XCC_ID and the
//     chain switches to the L2-local exchange (below) only if the first, placement-independent
//     exchange proves that all k workgroups really sit on one XCD;
//   * every leapfrog ("tick"): the compute waves evaluate their sites' log-lik + gradient from LDS
//     -> one interleaved DPP wave reduction -> LDS; the control wave forms the workgroup partial
//     (D gradients + log-lik, f32) and
//   * all-gathers the k partials through 8-byte {epoch, value} granules (guide G16, form R2:
//     the data is the flag), double-buffered by epoch parity, every spin bounded:
//       - placement-independent form: ONE sc1 (write-through) store per granule, relaxed
//         agent-scope (sc1) polls;
//       - L2-local form (verified same-XCD chains only): workgroup-scope stores keep the line in
//         the XCD's L2, sc1 polls bypass L1 and hit that L2: several times shorter hop;
//     every workgroup sums the k records in the same fixed order in f64, so all k copies of the
//     chain state stay bit-identical without any broadcast;
//   * the control wave (lane d = dimension d) then writes the SPECULATIVE next position -- the next
//     leaf of the subtree or the first leaf of the next doubling, a few FMAs from the gathered
//     gradient -- and releases the compute waves; NumPyro's per-leaf decisions (finish the
//     velocity-Verlet step, energy error, checkpointed U-turn tests, tree edges, transition end with
//     dual averaging / Welford windows / new momentum) and the multinomial bookkeeping run BESIDE
//     that evaluation.  If the decided position is not bit-equal to the guess the evaluation is
//     dropped and redone (transition ends), so correctness never depends on the guess.  occu_rn
//     and the HBM-row form (long evaluations) decide first and overlap only the
//     bookkeeping.  Direction bits and transition uniforms come from two separate xoshiro streams
//     so that neither form reorders a stream (the oracle draws from the same two).
// No host round trip, no kernel boundary and no HBM traffic inside the sampling loop.
#pragma once
#include "occu_device.hpp"

// In-kernel phase stamps (guide section 7 "In-kernel stamps"): diagnostic builds only
// (make stamps); the shipped kernel executes none of this.
#ifdef BL_STAMPS
#define BL_STAMP_DECL long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long st_prev = (long long)clock64(); long long st_rt0 = (long long)wall_clock64(); long long st_spins = 0; \
    long long st_kn[3] = {0, 0, 0}, st_kc[3] = {0, 0, 0}; int st_kind = 0; long long st_sub[6] = {0, 0, 0, 0, 0, 0}; long long st_t = 0;
#define BL_STAMP(i) { const long long st_now = (long long)clock64(); st_acc[i] += st_now - st_prev; st_prev = st_now; }
// critical-control time split by what the tick decided: 0 next leaf of the subtree, 1 next doubling, 2 transition end / init
#define BL_STAMP_KIND(k) st_kind = (k);
#define BL_STAMP_CRIT { const long long st_now = (long long)clock64(); st_kn[st_kind]++; st_kc[st_kind] += st_now - st_prev; }
#define BL_COUNT_SPINS(n) st_spins += (long long)(n);
// sub-steps of a decision: BL_SUB0 starts the clock, BL_SUB(i) charges the time since the last mark to slot i
#define BL_SUB0 st_t = (long long)clock64();
#define BL_SUB(i) { const long long st_n2 = (long long)clock64(); st_sub[i] += st_n2 - st_t; st_t = st_n2; }
#else
#define BL_SUB0
#define BL_SUB(i)
#define BL_COUNT_SPINS(n)
#define BL_STAMP_DECL
#define BL_STAMP(i)
#define BL_STAMP_KIND(k)
#define BL_STAMP_CRIT
#endif

// Rarely-read launch constants and output pointers live in device memory (keeps the kernel's
// SGPR budget for the loop).
struct BlNutsCold {
    int num_warmup, num_samples, nwin;
    float target_accept;
    int win_end[32];               // numpyro adaptation windows (inclusive ends)
    float loc_b, isc2_b, loc_a, isc2_a;  // Normal prior loc, 1/scale^2 (0 for a Laplace prior)
    float l1_b, l1_a;                    // Laplace prior: 1/scale (0 for a Normal prior)
    double prior_const;            // sum_k log(scale_k) + D/2 log(2 pi)  (+ log B(a,b) for MODEL 2)
    float fp_a, fp_b;              // MODEL 2: Beta(a, b) prior of the false-positive rate; MODEL 3: Exponential(rate = fp_a)
    const uint32_t *rng;           // [C][64][4] xoshiro states (host-jumped)
    const float *init_theta;       // [C][D] or null -> Uniform(-2,2)
    const int *abort_flag;         // host-mapped
    float *draws;                  // [C][S][D]
    unsigned char *diverging;      // [C][S]
    int *num_steps;                // [C][S]
    float *accept_prob;            // [C][S]
    float *potential;              // [C][S]
    float *step_size;              // [C]
    float *inv_mass;               // [C][D]
    long long *nleap;              // [C][2]
    int *status;                   // [1]
    int *xcd_local;                // [C] 1 if the chain ran on the L2-local exchange
    long long *dbg;                // [BL_DBG_SLOTS] phase cycle counters (diagnostic BL_STAMPS builds only); from 32: chain 0's site-evaluation
                                   // cycles per compute wave, [workgroup][wave] at 8 per workgroup
};
#define BL_DBG_SLOTS 544           // 32 + 64 workgroups x 8 compute waves

struct BlNutsParams {
    const float *rows;             // HBM data matrix [n_rows][n_stride]
    int n_sites, n_stride, T, J, Ks, Ko;
    int num_chains;
    int k;                         // workgroups per chain
    int nloc;                      // sites per workgroup
    int rec_stride;                // floats per LDS pair record
    int nvp;                       // granules per workgroup record: 16, 32 or 64 (>= D+4)
    int ncw;                       // compute waves per workgroup: selects the CW instantiation (host side)
    int grp_kernel;                // 1: launch the GRP instantiation (lane groups chosen, or a chain of one workgroup); host side
    int wide;                      // 1: the chain's workgroups span XCDs (blocks chain*k .. chain*k + k - 1), fabric exchange
    const float *nmix_tab;         // MODEL 4: B[t][n][site] = sum_j m log C(n, y_j) (-inf below the largest count), row length n_stride
    int fp_mode;                   // MODEL 2 / 3: 0 none, 1 = rate acts on every site ("constant"), 2 = on unoccupied sites only
    int max_depth;
    int max_abundance;             // occu_rn only (occu_rn.py:26)
    int rn_off;                    // occu_rn: byte offset in LDS of its scratch (lgamma table + wave-private tables), behind the records;
                                   // dynamic occupancy (MODEL 8): of its lane-private columns (dyn_device.hpp)
    int nmix_lds;                  // MODEL 4: 1 = the table's columns of a workgroup's sites are staged in LDS at rn_off (row length 2 ceil(nloc / 2))
    int lane_grp;                  // lanes that share one site pair -- dynamic occupancy: 1, 2, 4, 8 (dyn_device.hpp); occu / false positives:
                                   // log2(period lanes) | log2(visit lanes) << 4, 0 = one pair per lane (occu_device.hpp: bl_eval_sites_grp)
    int allow_local;               // 0: always use the placement-independent exchange
    int poll_sleep;                // s_sleep(1) repeats between re-polls (fabric form)
    int first_delay;               // s_sleep(1) repeats between publishing and the first poll
    int pitch;                     // granules between consecutive workgroup records (>= nvp)
    int n_species;                 // > 1: ONE chain over all species' coefficients (occu.py:182-186); layouts in occu_device.hpp (BL_SP_*)
    int sp_lds;                    // floats between two species' record regions in LDS
    unsigned spin_limit;           // bound of an exchange's wait in MICROSECONDS of wall time (a peer that is late -- late-resident,
                                   // descheduled by a co-tenant or a profiler -- is not an error for a while; one that vanished is)
    unsigned long long *xchg;      // [C][BL_XCHG_SLOTS][k][pitch] granules, zeroed before every launch
    const BlNutsCold *cold;
};

// Control-wave state that changes only at subtree / transition boundaries (LDS, control wave only).
struct BlCtlScalars {
    double U, Up;                  // potential at the current position / at the tree's proposal
    float wt, sumacc, eps;         // tree log-weight, sum of accept probs, step size
    float da_xt, da_xavg, da_gavg, da_prox;  // dual averaging
    int nprop, diverged, da_t, wf_n, win_idx, it;
    long long nleap_w, nleap_s;
};
enum { SV_TH = 0, SV_GR, SV_ZL, SV_RL, SV_GL, SV_ZR, SV_RR, SV_GRR, SV_ZP, SV_GP, SV_RSUM, SV_WFMEAN, SV_WFM2, SV_MOMZ };

__device__ __forceinline__ float bl_exp(float x) { return __builtin_amdgcn_exp2f(x * BL_LOG2E); }
__device__ __forceinline__ float bl_log(float x) { return BL_LN2 * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float bl_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float bl_logaddexp(float a, float b)
{
    const float m = fmaxf(a, b);
    if (m == -INFINITY) return m;
    return m + bl_log(1.0f + bl_exp(-fabsf(a - b)));
}
// logaddexp(a, b) and sigmoid(b - a) = weight of b, from ONE exp / log / rcp
__device__ __forceinline__ void bl_merge_weights(float a, float b, float &lse, float &pb)
{
    const float m = fmaxf(a, b);
    if (m == -INFINITY) { lse = m; pb = 0.0f; return; }
    const float d = b - a;
    const float e = bl_exp(-fabsf(d)), op = 1.0f + e;
    lse = m + bl_log(op);
    pb = (d > 0.0f ? 1.0f : e) * bl_rcp(op); // NaN d (inf - inf) -> e NaN -> pb NaN -> "u < pb" false
}

// numpyro hmc_util._is_turning (diagonal mass); lane d = dim d, lanes >= D hold zeros.
// (D <= 0: "at most 8 dimensions", decided once per launch -- the fixed three-level sum)
__device__ __forceinline__ bool bl_is_turning(float minv, float rl, float rr, float rsum, int D)
{
    const float rho = rsum - 0.5f * (rl + rr);
    float dl = minv * rl * rho, dr = minv * rr * rho;
    if (D <= 0) bl_low_sum2_w8(dl, dr); else bl_low_sum2(dl, dr, D);
    return (dl <= 0.0f) || (dr <= 0.0f);
}

// Velocity-Verlet pieces, written once so that the speculative and the deciding code paths round identically:
// end-of-leaf momentum r' = r_half - (eps/2) g', and the start of the next leaf from (z, r, g):
// r_half = r - (eps/2) g ,  z' = z + eps M^-1 r_half        (epsdir = +-eps)
__device__ __forceinline__ float bl_leaf_momentum(float rh, float epsdir, float g) { return fmaf(-0.5f * epsdir, g, rh); }
__device__ __forceinline__ void bl_next_leaf(float z, float r, float g, float epsdir, float minv, float &rh_out, float &z_out)
{
    const float rh = fmaf(-0.5f * epsdir, g, r);
    rh_out = rh;
    z_out = fmaf(epsdir * minv, rh, z);
}

__device__ __forceinline__ unsigned bl_xcc_id()
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xFu;
}

__device__ __forceinline__ unsigned long long bl_poll_load(const unsigned char *base, unsigned off)
{
    return __hip_atomic_load(reinterpret_cast<const unsigned long long *>(base + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// Bounded wait of the exchange polls: the wall clock (100 MHz) is looked at every 256th unsuccessful round only
struct BlSpinBound {
    unsigned spins = 0;
    long long deadline = 0;
    __device__ __forceinline__ bool expired(unsigned limit_us)
    {
        if ((++spins & 255u) != 0u) return false;
        const long long now = (long long)wall_clock64();
        if (deadline == 0) { deadline = now + (long long)limit_us * 100; return false; }
        return now > deadline;
    }
};

// Slots of the exchange, by epoch: a workgroup whose decisions drop the evaluation in flight has already published it and goes on
// to publish the corrected one without reading anybody's record in between, so it can be two epochs ahead of a peer that is still
// reading -- four slots keep those apart (two sufficed while dropped evaluations were not published).
#define BL_XCHG_SLOTS 4u
#ifndef BL_COMPUTE_PRIO
#define BL_COMPUTE_PRIO 3 // s_setprio of the compute waves (not occu_rn's); 0: left at the launch value (A/B)
#endif
#ifndef BL_POLL_NB
#define BL_POLL_NB 2 // batches of 8 polls per lane in flight at once when a chain has more workgroups than one batch covers (A/B: 1 = one after the other)
#endif
#define SMALL_D_NVP(LEAN, MODEL, KS, KO) ((LEAN) && ((MODEL) == 0 || (MODEL) == 1) && (KS) + (KO) + 2 <= 8)
#ifndef BL_GRP_FORM
#define BL_GRP_FORM 2 // what the sampler's GRP instantiation carries: 2 = the lane-group evaluator alone, 1 = both evaluators (A/B)
#endif
// GRP: the instantiation for lane groups and for chains of ONE workgroup (occu_device.hpp: bl_phase_a); without it a chain of one
// workgroup goes through the exchange like any other (as until round 3)
// JSEL: the plain model's one-pair-per-lane form this instantiation carries (occu_device.hpp: bl_eval_sites; -1 = all of them)
// LEAN: ONE species, ONE period, at most one site pair per compute lane, and every workgroup's record polled in one batch
// (k <= 8 x 64 / nvp) -- by far the commonest launch of the plain model, the headline among them.  The kernel then carries neither the
// species loops nor the several-batches poll, its evaluation is straight-line code (no loop over periods or pairs), and with <= 8
// coefficients the sums over the dimensions and the granule count are compile-time facts too.  Same reason as JSEL: what a kernel merely
// carries costs the rest (same-box A/Bs: 2.27 -> 2.15 -> 2.12 -> 2.03 us per leapfrog, profiles/r04/e_ab_grp_instantiation.txt).
template <int KS, int KO, bool LDS, int MODEL, int CW, bool GRP = false, int JSEL = -1, bool LEAN = false>
__global__ void __launch_bounds__(64 * (CW + 1)) bl_nuts_kernel(const BlNutsParams p)
{
    // a chain of ONE workgroup runs on BL_CWAVES_SINGLE compute waves and nothing else does (biolith_hip.hip: choose_geometry), so
    // which of the two paths a kernel has is a compile-time fact: the one-workgroup kernels carry no publish / poll, the others no
    // k == 1 branches (stacked 2.525 -> 2.513 us, dynamic 4.99 -> 4.95: profiles/r04/i_ab_split_single.txt)
    constexpr bool multi_wg = !(GRP && CW == BL_CWAVES_SINGLE);
    // XCD-aware mapping (speed only; see header): label = b % 8 names a set of blocks that share an XCD.
    // Wide geometry (slices that only fit LDS when a chain takes more than one XCD's CUs): consecutive blocks, any XCD.
    const int label = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int chain = p.wide ? (int)blockIdx.x / p.k : label + 8 * (slot / p.k);
    const int member = p.wide ? (int)blockIdx.x % p.k : slot % p.k;
    if (chain >= p.num_chains) return; // whole block leaves before any barrier or exchange
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // (one species: a compile-time fact in the lean kernels and for the models that are never sampled jointly)
    const int Ks = p.Ks, Ko = p.Ko, nsp = (LEAN || (MODEL != 0 && MODEL != 2)) ? 1 : p.n_species;
    const int Dsp = Ks + Ko + 2;    // coordinates of one species
    const int D = bl_model_dim<MODEL>(Ks, Ko, p.fp_mode) + (nsp - 1) * Dsp;
    // (lean, one pair per lane: one period, a compile-time fact -- the evaluation is then straight-line code; the lane-group and the
    // Royle-Nichols kernels keep their loops)
    constexpr bool ONE1 = LEAN && !GRP && MODEL == 0;
    const int T = ONE1 ? 1 : p.T, J = p.J;
    const int s0 = member * p.nloc;
    int cnt = p.n_sites - s0;
    cnt = cnt < 0 ? 0 : (cnt > p.nloc ? p.nloc : cnt);
    const float *grows = nullptr;
    int ld = p.rec_stride;
    if constexpr (MODEL == 1) { // occu_rn: the order its records are staged in, and the waves' shares of the sites (rn_device.hpp)
        bl_rn_split_init<KS, KO, CW>(p.rows, p.n_stride, s0, cnt, T, J, p.rn_off);
        __syncthreads();
    }
    if constexpr (LDS) {
        for (int sp = 0; sp < nsp; sp++)
            bl_stage_records(p.rows, p.n_stride, s0, cnt, T, J, KS, bl_layout_ko<MODEL>(KO), p.rec_stride, 64 * (CW + 1), sp, sp * p.sp_lds, MODEL == 1 ? bl_rn_order<CW>(p.rn_off, cnt) : nullptr);
    } else {
        grows = p.rows + s0;
        ld = p.n_stride;
    }
    if constexpr (MODEL == 1) bl_rn_fill_lgamma(p.rn_off, p.max_abundance, 64 * (CW + 1)); // (the barrier below publishes it)
    if constexpr (MODEL == 4) {
        if (p.nmix_lds) bl_stage_nmix_tab(p.nmix_tab + s0, p.n_stride, cnt, 2 * ((p.nloc + 1) / 2), T * (p.max_abundance + 1), p.rn_off, 64 * (CW + 1));
    }
    float *sh_coef = bl_lds_f(BL_OFF_COEF);
    int *sh_flag = bl_lds_i(BL_OFF_FLAG);
    float *sh_ckr = bl_lds_f(BL_OFF_CKR), *sh_ckrs = bl_lds_f(BL_OFF_CKRS);
    float *sv = bl_lds_f(BL_OFF_SV) + lane;  // state vector slot s of this lane: sv[s * 64]
    BlCtlScalars *ss = reinterpret_cast<BlCtlScalars *>(bl_smem_raw + BL_OFF_SS);
    if (tid < 64) sh_coef[tid] = 0.0f;
    __syncthreads();

    // ------------------------------------------------ replicated chain state (control wave) ----
    const BlNutsCold *cold = p.cold;
    const bool act = lane < D;
    // the false-positive coordinate (logit rate, Beta prior: MODEL 2; log rate, Exponential prior: MODEL 3), not Normal
    const bool is_phi = (MODEL == 2 || (MODEL == 3 && p.fp_mode != 0)) && lane == D - 1;
    // lane d = coordinate d of theta = [species 0: beta, alpha | species 1: ... | (phi)]: species and index within it
    const int lsp = lane < nsp * Dsp ? lane / Dsp : 0, lj = lane - lsp * Dsp;
    // where lane d's coefficient lives in the LDS coefficient block, and its partial in a wave's row of the partial table;
    // the log-lik (lane D) and the shared phi are sums over the species' slots (part_all)
    int my_pos = lane < nsp * Dsp ? lsp * BL_SP_COEF(KS, KO) + bl_coef_pos(lj, Ks, Ko, KS, KO) : nsp * BL_SP_COEF(KS, KO) + 1;
    int part_pos = lane < nsp * Dsp ? lsp * BL_SP_PART(KS, KO) + bl_coef_pos(lj, Ks, Ko, KS, KO) : (is_phi ? KS + KO + 3 : KS + KO + 2);
    bool beta_lane = lj <= Ks;          // this coordinate takes beta's prior (else alpha's)
    if constexpr (MODEL == 8) {         // dynamic occupancy: [b_psi | b_gamma | b_eps | alpha], its own layout (dyn_device.hpp)
        my_pos = part_pos = bl_dyn_pos(lane < D ? lane : D, Ks, Ko, KS, KO);
        beta_lane = lane < 3 * (Ks + 1);
    }
    const bool part_all = MODEL != 8 && lane >= nsp * Dsp;
    const int part_rs = nsp > 1 ? nsp * BL_SP_PART(KS, KO) : BL_PART_STRIDE;
    // ---- loop-carried registers: only what the per-leaf path touches ----
    BlRng rng_d, rng_u, rng_dir;        // per-dimension stream, transition uniforms, direction bits
    float minv = act ? 1.0f : 0.0f;     // diagonal inverse mass (0 in idle lanes keeps sums clean)
    float eps = 1.0f, epsdir = 0.f;
    double E0 = 0.0;                    // energy at the start of the transition
    int depth = 0;
    // subtree under construction
    double sUp = 0.0;
    float szp = 0, sgp = 0, srsum = 0, swt = 0, ssumacc = 0;
    int snprop = 0;
    bool sdiv = false, sturn = false, going_right = false;
    // leaf in flight
    float cz = 0, rh = 0;
    // bookkeeping deferred from the last leaf (runs while the compute waves evaluate the next one)
    bool pend = false, pend_first = false, pend_end = false, pend_sturn = false, pend_sdiv = false;
    float pend_dE = 0.f, pend_z = 0.f, pend_g = 0.f;
    double pend_U = 0.0;
    int pend_snprop = 0;
    bool init_pending = true;           // evaluating the initial position
    // momentum of the NEXT transition, drawn one transition ahead (same order on the per-dimension streams), so that a
    // transition's end finds it ready: z ~ N(0,1), r0 = z / sqrt(M^-1), kinetic energy sum(M^-1 r0^2) / 2
    float mom_r0 = 0.f;                 // (its z is kept in LDS: only a closing adaptation window needs it again)
    double mom_kin = 0.0;
    // what a finished transition still owes (outputs, Welford moments, the fresh tree, the next momentum): written out beside
    // the evaluation of the new transition's first leaf, which the decisions release first.  Everything it needs is still in
    // the control wave's LDS block by then; only these bits travel in a register:
    // 1 pending, 2 warmup transition, 4 add the draw to the Welford moments, 8 the adaptation window ended, 16 emit a draw
    int end_kind = 0;
    bool local = false;                 // L2-local exchange proven safe for this chain
    float prior_loc = 0.f, prior_isc2 = 0.f, prior_l1 = 0.f;
    double prior_const = 0.0;
    int S = 0, W = 0, total = 0, nwin = 0, win_end_cur = 0;
    float target_accept = 0.8f;

    // exchange addressing (per lane, fixed for the whole launch): byte offsets of the <= 8 granules this
    // lane polls per round; record indices beyond k are clamped to k-1 (duplicates carry valid
    // tags and are ignored in the sum, so a round needs no predication)
    const int nvp = SMALL_D_NVP(LEAN, MODEL, KS, KO) ? 16 : p.nvp, G = 64 / nvp; // (<= 8 coefficients: D + 4 <= 16 granules, a compile-time fact when lean)
    const int c_idx = lane & (nvp - 1), sub = lane / nvp;
    const unsigned rec_bytes = (unsigned)(p.k * p.pitch * 8);
    const unsigned char *xbase = reinterpret_cast<const unsigned char *>(p.xchg) + (size_t)chain * BL_XCHG_SLOTS * rec_bytes;
    unsigned poff[8];
    float pval[8]; // 1 if that load is a real (unclamped) record of this lane
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int w = q * G + sub;
        poff[q] = (unsigned)(((w < p.k ? w : p.k - 1) * p.pitch + c_idx) * 8);
        pval[q] = (w < p.k) ? 1.0f : 0.0f;
    }
    const int nq = (p.k + G - 1) / G; // loads per round actually needed (<= 8 when k <= 8 G)
    const bool one_batch = LEAN || p.k <= 8 * G;

    if (wave == 0) {
        S = cold->num_samples; W = cold->num_warmup; total = W + S;
        // what a warmup transition's end needs of the cold block, in registers: read there they are three device-memory loads one
        // behind the other (the last one's index behind an LDS read) between a transition's last gradient and the next one's first position
        target_accept = cold->target_accept; nwin = cold->nwin; win_end_cur = cold->win_end[0];
        prior_loc = beta_lane ? cold->loc_b : cold->loc_a;
        prior_isc2 = act ? (beta_lane ? cold->isc2_b : cold->isc2_a) : 0.0f;
        prior_l1 = act ? (beta_lane ? cold->l1_b : cold->l1_a) : 0.0f;
        if (is_phi) { prior_loc = cold->fp_a; prior_isc2 = cold->fp_b; prior_l1 = 0.0f; }
        prior_const = cold->prior_const;
        const uint32_t *rs = cold->rng + ((size_t)chain * BL_NSTREAM + lane) * 4;
        rng_d.s0 = rs[0]; rng_d.s1 = rs[1]; rng_d.s2 = rs[2]; rng_d.s3 = rs[3];
        const uint32_t *ru = cold->rng + ((size_t)chain * BL_NSTREAM + BL_SCALAR_STREAM) * 4;
        rng_u.s0 = ru[0]; rng_u.s1 = ru[1]; rng_u.s2 = ru[2]; rng_u.s3 = ru[3];
        const uint32_t *rd = cold->rng + ((size_t)chain * BL_NSTREAM + BL_DIR_STREAM) * 4;
        rng_dir.s0 = rd[0]; rng_dir.s1 = rd[1]; rng_dir.s2 = rd[2]; rng_dir.s3 = rd[3];
        const float u0 = bl_rng_uniform(rng_d); // init_to_uniform(radius=2), fit.py:93
        const float *init = cold->init_theta;
        cz = act ? (init ? init[chain * D + lane] : 4.0f * u0 - 2.0f) : 0.0f;
        if (act) sh_coef[my_pos] = cz;
        const float z_first = bl_rng_normal(rng_d);  // the first transition's momentum (M^-1 = 1)
        mom_r0 = act ? z_first : 0.0f;
        mom_kin = (double)(0.5f * bl_wave_sum(mom_r0 * mom_r0));
#pragma unroll
        for (int s = 0; s < 16; s++) sv[s * 64] = 0.0f;
        sv[SV_MOMZ * 64] = z_first;
        if (lane == 0) {
            sh_flag[0] = 0; sh_flag[1] = 0; sh_flag[2] = 0; // run status; L2-local exchange proven; (spare)
            for (int w = 0; w < 16; w++) bl_lds_i(BL_OFF_TAG)[w] = 0; // no row of the partial table belongs to an evaluation yet
            BlCtlScalars z{};
            z.eps = 1.0f; z.da_prox = 2.302585093f; // log(10 * step_size0)
            *ss = z;
        }
    }
    __syncthreads();

    // The tree in progress -- its two edges (z, r, g), the momentum sum, the proposal and the weight / acceptance counters -- lives in
    // the control wave's REGISTERS since round 3 (13 values; until then in its private LDS block, where every subtree's end and every
    // transition's end fetched them one dependent round trip after the other: two to four per such tick).  LDS keeps what only the
    // warmup adapter and the outputs touch.
    // (0: the sums over the dimensions take the fixed three-level form; a compile-time fact in the lean kernels of <= 8 coefficients)
    constexpr bool SMALL_D = SMALL_D_NVP(LEAN, MODEL, KS, KO);
    const int Dred = (SMALL_D || D <= 8) ? 0 : D;
    float ck_last = 0.f, cks_last = 0.f;
    float t_zl = 0.f, t_rl = 0.f, t_gl = 0.f, t_zr = 0.f, t_rr = 0.f, t_gr = 0.f, t_rsum = 0.f, t_zp = 0.f, t_gp = 0.f;
    float t_wt = 0.f, t_sumacc = 0.f;
    double t_Up = 0.0;
    int t_nprop = 0, t_it = 0;
    auto mom_refresh = [&]() { // (after M^-1 changed: the same z under the new metric)
        mom_r0 = act ? sv[SV_MOMZ * 64] * __builtin_amdgcn_rsqf(minv) : 0.0f;
        mom_kin = (double)(0.5f * bl_wave_sum(minv * mom_r0 * mom_r0));
    };
    // The rest of a transition's end (everything the next position did not need), see end_kind above.  The tree's proposal
    // (t_zp, t_gp, t_Up), its counters (t_nprop, t_sumacc), t_it and the started transition's momentum (mom_r0) are
    // untouched since the decisions.
    auto end_deferred = [&]() {
        if (end_kind == 0) return;
        const float th = t_zp, gr = t_gp, r0 = mom_r0;
        const double U = t_Up;
        const int nprop = t_nprop, it = t_it;
        if (end_kind & 2) {
            ss->nleap_w += nprop;
            if (end_kind & 4) { // Welford moments of the draw (a window's last draw was added by the decisions themselves)
                const int wf_n = ss->wf_n + 1;
                const float wf_mean0 = sv[SV_WFMEAN * 64];
                const float dpre = th - wf_mean0;
                const float wf_mean = wf_mean0 + dpre * bl_rcp((float)wf_n);
                sv[SV_WFMEAN * 64] = wf_mean;
                sv[SV_WFM2 * 64] += dpre * (th - wf_mean);
                ss->wf_n = wf_n;
            }
            if (end_kind & 8) {
                const int wi = ss->win_idx + 1;
                ss->win_idx = wi;
                win_end_cur = cold->win_end[wi < 31 ? wi : 31]; // (beside the next transition's first evaluation)
            }
        } else if (end_kind & 16) {
            ss->nleap_s += nprop;
            if (member == 0) {
                const size_t s = (size_t)chain * S + (it - W);
                if (act) cold->draws[s * D + lane] = th;
                if (lane == 0) {
                    cold->num_steps[s] = nprop;
                    cold->accept_prob[s] = t_sumacc * bl_rcp((float)nprop);
                    cold->diverging[s] = pend_sdiv ? 1 : 0;
                    cold->potential[s] = (float)U;
                }
            }
        }
        if (end_kind & (2 | 16)) t_it = it + 1;
        end_kind = 0;
        // the fresh tree of the transition that has already started
        t_zl = th; t_rl = r0; t_gl = gr;
        t_zr = th; t_rr = r0; t_gr = gr;
        t_rsum = r0;
        t_wt = 0.f; t_sumacc = 0.f; t_nprop = 0;
        // and the momentum of the one after it
        const float z = bl_rng_normal(rng_d);
        sv[SV_MOMZ * 64] = z;
        mom_r0 = act ? z * __builtin_amdgcn_rsqf(minv) : 0.0f;
        mom_kin = (double)(0.5f * bl_wave_sum(minv * mom_r0 * mom_r0));
    };

    // Deferred bookkeeping of the last finished leaf (hmc_util._combine_tree: proposal + weights).
    // Nothing in here can change where the next leaf is evaluated.
    auto run_deferred = [&]() {
        if (!pend) return;
        pend = false;
        const float lw = -pend_dE;
        const float lacc = pend_dE <= 0.0f ? 1.0f : bl_exp(-pend_dE); // min(1, exp(-dE))
        if (pend_first) {
            szp = pend_z; sgp = pend_g; sUp = pend_U; swt = lw; ssumacc = lacc;
        } else {
            // biased=False: uniform transition kernel inside the subtree
            float pr, lse;
            bl_merge_weights(swt, lw, lse, pr); // pr = expit(lw - swt)
            const float u = bl_rng_uniform(rng_u);
            if (u < pr) { szp = pend_z; sgp = pend_g; sUp = pend_U; }
            swt = lse;
            ssumacc += lacc;
        }
        if (pend_end) {
            // biased=True: merge the finished subtree into the tree
            const float wt = t_wt;
            float pr = fminf(1.0f, bl_exp(swt - wt));
            if (pend_sturn || pend_sdiv) pr = 0.0f;
            const float u = bl_rng_uniform(rng_u);
            if (u < pr) { t_zp = szp; t_gp = sgp; t_Up = sUp; }
            t_wt = bl_logaddexp(wt, swt);
            t_sumacc += ssumacc;
            t_nprop += pend_snprop;
        }
    };

    unsigned epoch = 0;
    // Result of the last exchange, waiting for its decisions (made while the compute waves already
    // evaluate the position those decisions are EXPECTED to choose -- see "speculative position" below).
    bool have_pending = false, p_timed_out = false;
    double p_acc = 0.0;                 // lane d: d log-lik / d theta_d summed over the chain's workgroups; lane D: log-lik
    float p_cg = 0.f, p_pe2 = 0.f;      // gradient of the potential / twice the prior energy at the evaluated position
    float cz_spec = 0.f;                // position written to LDS for the evaluation in flight
    int flag = 0;
    BL_STAMP_DECL
    // The NumPyro decisions that follow one gathered gradient (hmc_util: _build_basetree tail, _iterative_build_subtree
    // step, _double_tree / _combine_tree, and at transition end the warmup adapter): takes the stashed exchange
    // result, leaves the next position to evaluate in cz and the run status in flag.
    // Speculative overlap pays while a dropped evaluation is cheap: the LDS-staged occu / false-positive / occu_cop
    // forms (<= ~3 site pairs per lane) and nmixture (measured: 8.0 -> 7.6 us).  occu_rn's evaluation (sums over N) dominates its tick and the HBM-row form serves huge
    // slices: there the decisions are taken right after the exchange and only the bookkeeping overlaps.
    constexpr bool SPEC = LDS && MODEL != 1;
    // occu_rn (round 5): speculate where the guess cannot be wrong short of a divergence -- behind an EVEN leaf of a subtree (no U-turn
    // test is made there and the subtree cannot be full: hmc_util._leaf_idx_to_ckpt_idxs, _iterative_build_subtree) the next leaf is the
    // next leaf, so its evaluation starts at once and the decisions about the even leaf run beside it; everywhere else the decisions
    // come first, as before (a dropped occu_rn evaluation costs more than they do).  Round 6 (BL_RN_HYBRID 2): behind EVERY leaf that is
    // not its subtree's last -- an odd one can end the subtree by a checkpointed U-turn test, which drops the evaluation, but that is one
    // leaf in forty where the decisions in front of the release were 2 800 cycles on one in five: 7.37 -> 7.30 us per leapfrog, draws,
    // trees and adaptation bit-identical (profiles/r06/s_time_rn_hybrid.txt, tools/ab_rn_hybrid.py).  1: even leaves only; 0: none (A/B).
#ifndef BL_RN_HYBRID
#define BL_RN_HYBRID 2
#endif
    constexpr bool HYBRID = BL_RN_HYBRID && LDS && MODEL == 1;
    auto decide = [&]() {
        have_pending = false;
        const double acc = p_acc;
        const float cg = p_cg, pe2 = p_pe2;
        const double ll_tot = bl_readlane_d(acc, D);
        const bool abort_req = bl_readlane_d(acc, D + 1) != 0.0;
        flag = 0;
        if (p_timed_out) flag = 4;     // BL_ERR_TIMEOUT
        else if (abort_req) flag = 5;  // BL_ERR_ABORTED
        bool new_transition = false;
        BL_STAMP_KIND(0)
        if (flag == 0) {
            if (init_pending) {
                BL_STAMP_KIND(2)
                BL_SUB0
                // initial evaluation done
                const double Un = -ll_tot + (double)(0.5f * bl_wave_sum(pe2)) + prior_const;
                init_pending = false;
                new_transition = true;
                // (no transition behind it: the deferred part only installs the start state and the first tree)
                t_zp = cz; t_gp = cg; t_Up = Un;
                end_kind = 1;
            } else {
                // ------- CRITICAL: finish the leaf (_build_basetree), decide where the next one goes -------
                const float cr = bl_leaf_momentum(rh, epsdir, cg);
                float s_prior = pe2, s_kin = minv * cr * cr;
                if (SMALL_D || Dred <= 0) bl_low_sum2_w8(s_prior, s_kin); else bl_low_sum2(s_prior, s_kin, D);
                const double Un = -ll_tot + (double)(0.5f * s_prior) + prior_const;
                const double Kn = (double)(0.5f * s_kin);
                double dE = (Un + Kn) - E0;
                if (dE != dE) dE = (double)INFINITY;
                const int leaf_idx = snprop;
                sdiv = dE > 1000.0;
                srsum = (leaf_idx == 0) ? cr : srsum + cr;
                snprop = leaf_idx + 1;
                // its multinomial / weight bookkeeping is deferred
                pend = true; pend_first = (leaf_idx == 0); pend_end = false;
                pend_dE = (float)dE; pend_U = Un; pend_z = cz; pend_g = cg;
                // checkpointed U-turn (_leaf_idx_to_ckpt_idxs / _is_iterative_turning)
                const int idx_max = __popc((unsigned)leaf_idx >> 1);
                const int idx_min = idx_max - (int)__builtin_ctz(~(unsigned)leaf_idx) + 1;
                if ((leaf_idx & 1) == 0) {
                    sh_ckr[idx_max * 64 + lane] = cr;
                    sh_ckrs[idx_max * 64 + lane] = srsum;
                    ck_last = cr; cks_last = srsum; // (the checkpoint the NEXT leaf's first test compares with: kept in registers too)
                } else {
                    // the first test is against the checkpoint the previous (even) leaf has just written -- popc((L - 1) >> 1) = popc(L >> 1)
                    // for odd L -- so it needs no LDS round trip; half of the odd leaves have no other test
                    sturn = bl_is_turning(minv, ck_last, cr, srsum - cks_last + ck_last, SMALL_D ? 0 : Dred);
                    for (int i = idx_max - 1; i >= idx_min && !sturn; i--) {
                        const float ck = sh_ckr[i * 64 + lane];
                        const float srs = srsum - sh_ckrs[i * 64 + lane] + ck;
                        sturn = bl_is_turning(minv, ck, cr, srs, SMALL_D ? 0 : Dred);
                    }
                }
                if (snprop < (1 << depth) && !sturn && !sdiv) {
                    // next leaf continues from this one
                    bl_next_leaf(cz, cr, cg, epsdir, minv, rh, cz);
                } else {
                    // ---------- subtree complete: extend the tree edge, tree-level U-turn ----------
                    BL_STAMP_KIND(1)
                    pend_end = true; pend_sturn = sturn; pend_sdiv = sdiv; pend_snprop = snprop;
                    // the edge this subtree extends takes its last leaf; the opposite edge stays
                    const float o_z = going_right ? t_zl : t_zr, r_other = going_right ? t_rl : t_rr, o_g = going_right ? t_gl : t_gr;
                    if (going_right) { t_zr = cz; t_rr = cr; t_gr = cg; } else { t_zl = cz; t_rl = cr; t_gl = cg; }
                    const float rsum = t_rsum + srsum;
                    t_rsum = rsum;
                    // numpyro _combine_tree (biased): turning = new_tree.turning | _is_turning(edges, r_sum)
                    const bool turning = sturn || bl_is_turning(minv, going_right ? r_other : cr, going_right ? cr : r_other, rsum, Dred);
                    depth++;
                    if (depth < p.max_depth && !turning && !sdiv) {
                        // next doubling
                        const bool was_right = going_right;
                        going_right = (bl_rng_next(rng_dir) >> 31) != 0u;
                        epsdir = going_right ? eps : -eps;
                        snprop = 0; sturn = false; sdiv = false;
                        // the edge the next doubling starts from: the one just extended (this leaf, in registers) or the other one
                        float ez = cz, er = cr, eg = cg;
                        if (going_right != was_right) { ez = o_z; er = r_other; eg = o_g; }
                        bl_next_leaf(ez, er, eg, epsdir, minv, rh, cz);
                    } else {
                        // ---------------- transition complete ----------------
                        // Only what the next position needs is done here -- the proposal, and during warmup the new step
                        // size (and metric, when a window closes); outputs, adaptation state, the fresh tree and the next
                        // momentum follow beside the evaluation of that position (end_deferred).
                        BL_STAMP_KIND(2)
                        BL_SUB0
                        run_deferred();
                        BL_SUB(0)
                        const int it = t_it;
                        end_kind = it < W ? (1 | 2) : (1 | 16);
                        if (it < W) {
                            // warmup_adapter.update_fn: dual averaging (t0=10, kappa=.75, gamma=.05)
                            const float accp = t_sumacc * bl_rcp((float)t_nprop);
                            const float g = target_accept - accp;
                            const int da_t = ss->da_t + 1;
                            const float tt = (float)da_t;
                            const float rt10 = bl_rcp(tt + 10.0f);
                            const float da_gavg = (1.0f - rt10) * ss->da_gavg + g * rt10;
                            const float da_xt = ss->da_prox - __builtin_amdgcn_sqrtf(tt) * 20.0f * da_gavg;
                            const float wgt = __builtin_amdgcn_exp2f(-0.75f * __builtin_amdgcn_logf(tt));
                            const float da_xavg = (1.0f - wgt) * ss->da_xavg + wgt * da_xt;
                            eps = bl_exp((it == W - 1) ? da_xavg : da_xt);
                            eps = fminf(fmaxf(eps, 1.1754944e-38f), 3.4028235e+38f);
                            ss->da_t = da_t; ss->da_gavg = da_gavg; ss->da_xt = da_xt; ss->da_xavg = da_xavg;
                            const int win_idx = ss->win_idx;
                            const bool middle = win_idx > 0 && win_idx < nwin - 1;
                            const bool at_end = it == win_end_cur;
                            if (middle) end_kind |= 4;
                            if (at_end) end_kind |= 8;
                            if (at_end && middle) {
                                // a window closes: the new metric enters the very next leaf, so this draw's moments are
                                // added here and now (a handful of times per run)
                                const float th = t_zp;
                                const int wf_n = ss->wf_n + 1;
                                const float wf_mean0 = sv[SV_WFMEAN * 64];
                                const float dpre = th - wf_mean0;
                                const float wf_mean = wf_mean0 + dpre * bl_rcp((float)wf_n);
                                const float wf_m2 = sv[SV_WFM2 * 64] + dpre * (th - wf_mean);
                                const float n = (float)wf_n;
                                const float var = wf_m2 * bl_rcp(n - 1.0f), rn5 = bl_rcp(n + 5.0f);
                                minv = act ? (n * rn5 * var + 1e-3f * 5.0f * rn5) : 0.0f;
                                sv[SV_WFMEAN * 64] = 0.f; sv[SV_WFM2 * 64] = 0.f; ss->wf_n = 0;
                                end_kind &= ~4; // (done)
                                ss->da_xt = 0.f; ss->da_xavg = 0.f; ss->da_gavg = 0.f; ss->da_t = 0;
                                ss->da_prox = bl_log(10.0f * eps);
                                mom_refresh();
                            }
                        }
                        if (it + 1 >= total) flag = 1; // done
                        else new_transition = true;
                        BL_SUB(1)
                    }
                }
            }
            if (new_transition) {
                // a fresh tree from the proposal (t_zp, t_gp, t_Up) with the momentum drawn ahead; its first doubling, first leaf
                const float th = t_zp, gr = t_gp;
                const float r0 = mom_r0;
                E0 = t_Up + mom_kin;
                depth = 0;
                going_right = (bl_rng_next(rng_dir) >> 31) != 0u;
                epsdir = going_right ? eps : -eps;
                snprop = 0; sturn = false; sdiv = false;
                bl_next_leaf(th, r0, gr, epsdir, minv, rh, cz);
                BL_SUB(2)
            }
        }
        if (flag != 0) end_deferred();
        if (flag != 0 && member == 0) {
            if (flag > 1 && lane == 0) atomicMax(cold->status, flag);
            if (act) cold->inv_mass[chain * D + lane] = minv;
            if (lane == 0) {
                cold->step_size[chain] = eps;
                cold->nleap[chain * 2 + 0] = ss->nleap_w;
                cold->nleap[chain * 2 + 1] = ss->nleap_s;
                cold->xcd_local[chain] = local ? 1 : 0;
            }
        }
    };

    // Two loops, one per role, meeting at the same two workgroup barriers per tick: the register allocator then sees the
    // compute waves' evaluation and the control wave's sampler state as DISJOINT live ranges (one shared loop keeps every
    // loop-carried register of either role alive through the other role's code: spills on both sides).
    if (wave > 0) {
        // A compute wave that shares its SIMD with the control wave (four or more compute waves: five waves on four SIMDs) is issued ahead
        // of it: the decisions have slack beside the evaluation, the evaluation has none (s_setprio: 0 ... 3, 0 at launch).  Same box,
        // priority 0 -> 3: stacked 2.343 -> 2.252 us per leapfrog, dynamic 3.705 -> 3.617, headline 1.976 -> 1.955; occu_rn -- whose
        // decisions are on the critical path -- 8.34 -> 8.45, and the one-workgroup form (simulate()'s defaults) 1.92 -> 1.96, so not there
        // (profiles/r05/p_ab_compute_priority.txt).
        if constexpr (BL_COMPUTE_PRIO > 0 && MODEL != 1 && multi_wg) __builtin_amdgcn_s_setprio(BL_COMPUTE_PRIO);
        unsigned epoch_c = 0;  // evaluations so far = the exchange epoch of the one in flight (the control wave counts the same)
        const int lane_grp = MODEL == 1 ? bl_rn_npos(p.rn_off) : p.lane_grp; // (occu_rn: its waves' shares of the sites, rn_device.hpp)
        while (true) {
            // The loop control of this tick (run status, "the exchange is L2-local") is read in ONE ds_read_b64 that is issued here
            // and looked at after the evaluation: read and tested first -- two dependent LDS round trips, as it was until round 3 --
            // it stood in front of every evaluation.  The price: one evaluation nobody needs when the run ends.
            typedef int BlInt2 __attribute__((ext_vector_type(2)));
            const BlInt2 ctl = *(__attribute__((address_space(3))) const volatile BlInt2 *)sh_flag;
            // ------------------------------------- phase A: compute waves, site log-lik ----
#ifdef BL_STAMPS
            const long long st_a0 = (long long)clock64();
#endif
            bl_phase_a<KS, KO, LDS, MODEL, CW, GRP ? BL_GRP_FORM : 0, JSEL, ONE1>(tid - 64, wave - 1, grows, ld, cnt, T, J, p.max_abundance, p.fp_mode, p.nmix_tab + s0, (MODEL == 4 && p.nmix_lds) ? 2 * ((p.nloc + 1) / 2) : p.n_stride, nsp, p.sp_lds, p.rn_off, lane_grp, p.nmix_lds);
#ifdef BL_STAMPS
            st_sub[5] += (long long)clock64() - st_a0; st_sub[4]++;
#endif
            if (ctl.x != 0) break;
            const bool local_c = ctl.y != 0;
            // ---- the LAST compute wave to finish publishes the workgroup's partial to the chain at once: the hand-off to the
            // other workgroups then runs beside the control wave's decisions about the previous leaf (which outlast phase A on
            // every doubling and transition-end tick) instead of after them.  Every evaluation is published, also one the
            // decisions are about to drop (all workgroups drop the same ones and skip that epoch's poll).
            // Who is last is found with ONE LDS round trip (round 3; an arrival counter -- atomic with return, then the rows -- took two):
            // behind its row of the partial table a wave writes the evaluation's number as the row's tag, then reads every row and every
            // tag in one go.  A wave's LDS instructions execute in order and the LDS serves the waves' instructions one after another,
            // so the wave whose reads see all tags current has read rows that are complete -- the last one always does; when two finish
            // together both may, and both publish the same granule (fixed wave order of the sum), which is harmless.
            // (The tags are read BEFORE the rows: tags all current at that point => every row was complete at that point.  Volatile
            // accesses through LDS-address-space pointers: program order kept, ds_read / ds_write -- a volatile generic pointer makes
            // them flat_load / flat_store with system coherence bits.)
            epoch_c++;
            if (multi_wg) { // (one workgroup per chain: nothing to publish -- the control wave reads the rows from LDS behind the barrier)
            typedef __attribute__((address_space(3))) volatile unsigned BlLdsU;
            typedef __attribute__((address_space(3))) const volatile float BlLdsF;
            BlLdsU *tags = (BlLdsU *)bl_lds_i(BL_OFF_TAG);
            BlLdsF *part = (BlLdsF *)(bl_lds_f(BL_OFF_PART) + part_pos);
            asm volatile("" ::: "memory"); // (compiler only: the row's stores stay in front of the tag's)
            if (lane == 0) tags[wave - 1] = epoch_c;
            unsigned stale = 0u;
#pragma unroll
            for (int w = 0; w < CW; w++) stale |= tags[w] ^ epoch_c;
            float row[CW];
#pragma unroll
            for (int w = 0; w < CW; w++) row[w] = part[w * part_rs];
            if (__builtin_amdgcn_readfirstlane(stale) == 0u) {
                // workgroup partial (fixed wave order): lanes < D their gradient component, lane D the log-lik
                float comp = row[0];
#pragma unroll
                for (int w = 1; w < CW; w++) comp += row[w];
                for (int sp = 1; sp < nsp; sp++) { // several species: the log-lik and the shared coordinate add the other species' slots
#pragma unroll
                    for (int w = 0; w < CW; w++) comp += part_all ? (float)part[w * part_rs + sp * BL_SP_PART(KS, KO)] : 0.0f;
                }
                if (lane > D) comp = 0.0f;
                if (lane == D + 1 && member == 0 && (epoch_c & 255u) == 0u)
                    comp = (__hip_atomic_load(cold->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) ? 1.0f : 0.0f;
                if (epoch_c == 1u) { // placement census: all k XCC ids equal  <=>  k * sum(x^2) == (sum x)^2
                    const float xcc = (float)bl_xcc_id(); // (read here, once: as a loop-carried value it was spilled to scratch and reloaded on every publish)
                    if (lane == D + 2) comp = xcc;
                    if (lane == D + 3) comp = xcc * xcc;
                }
                const unsigned char *rbase = xbase + (epoch_c & (BL_XCHG_SLOTS - 1u)) * rec_bytes;
                const unsigned long long granule = ((unsigned long long)epoch_c << 32) | __float_as_uint(comp);
                if (lane < nvp) {
                    // (the granule's offset is formed HERE, from an opaque copy of the lane id: hoisted out of the loop, the 64-bit
                    // address was spilled to scratch by the fullest instantiations and reloaded on every publish)
                    int lane_o = lane;
                    asm volatile("" : "+v"(lane_o));
                    const unsigned store_off = (unsigned)((member * p.pitch + (lane_o & (nvp - 1))) * 8);
                    unsigned long long *dst = reinterpret_cast<unsigned long long *>(const_cast<unsigned char *>(rbase) + store_off);
                    if (local_c) // line stays in this XCD's L2, where every consumer of this chain polls it
                        __hip_atomic_store(dst, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else         // write-through: visible to any XCD
                        __hip_atomic_store(dst, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            }
            // (Measured and dropped: no barrier here -- the control wave polling straight after its decisions, a counter of coefficient
            // reads guarding the one write that the barrier ordered: 2 % slower, the poll loop competes with the evaluation's tail.)
            __syncthreads(); // (the control wave's decisions are done)
            __syncthreads(); // the next position is in LDS
        }
    } else {
    int run_status = 0;
    while (true) {
        bool redo = false; // the evaluation in flight is not the one the sampler needs next
        if ((SPEC || HYBRID) && have_pending) {
            end_deferred(); // (only if the guess at a transition's end happened to be right: otherwise done in the branch below)
            decide();
            // was the position being evaluated right now the one just chosen?  (bit-equal or redo: correctness
            // never depends on the guess)
            redo = flag != 0 || __any(act && !(cz == cz_spec));
            // this leaf's proposal / weight bookkeeping, still under phase A.  (Measured and dropped, round 3: the same call in the shadow of
            // the first poll round -- every tick: 3 % slower, doubling ticks only: 2 % slower, the round's check waits for it -- and lagging
            // one tick, at the head of the next tick's decisions: no change, what a doubling tick sheds its successor picks up.)
            if (flag == 0) run_deferred();
            BL_STAMP_CRIT
            BL_STAMP(0)
        } else if (!SPEC) {
            end_deferred();
            run_deferred(); // non-speculative form (HYBRID: a tick whose decisions came first): only the bookkeeping overlaps the evaluation
            BL_STAMP(0)
        } else {
            end_deferred(); // beside the evaluation of the new transition's first leaf
            BL_STAMP(0)
        }
        // What the exchange of the evaluation in flight does not need is done BEFORE the barrier -- on a leaf tick the decisions
        // are over before the site evaluation is, and this ran in the shadow of nothing: the first poll then goes out the moment the
        // barrier opens (the prior at the position in flight, which kind of leaf comes next, the peeked direction, the other edge).
        float pe2 = 0.0f, pg = 0.0f;
        int spec_kind = 0;           // 0 none, 1 next leaf of the subtree, 2 / 3 next doubling from this leaf / from the other edge
        float spec_ed = epsdir, spec_other = 0.0f;
        if (!redo) {
        // ------------------------------------------------ potential at cz (lane d) ----
        // prior of lane d: pe2 = 2 x its energy, pg = d energy / d theta_d
        //   Normal(loc, scale):            pe2 = (theta-loc)^2 / scale^2 ,  pg = (theta-loc) / scale^2
        //   Laplace(loc, scale):           pe2 = 2 |theta-loc| / scale ,    pg = sign(theta-loc) / scale   (one of isc2, l1 is 0)
        //   phi = logit f, f ~ Beta(a,b):  energy = a softplus(-phi) + b softplus(phi)  (Jacobian included),
        //                                  pg = (a+b) sigmoid(phi) - a        (prior_loc = a, prior_isc2 = b)
        const float dth = cz - prior_loc;
        pe2 = fmaf(dth * dth, prior_isc2, 2.0f * fabsf(dth) * prior_l1);
        pg = fmaf(dth, prior_isc2, dth > 0.0f ? prior_l1 : (dth < 0.0f ? -prior_l1 : 0.0f));
        if constexpr (MODEL == 2) {
            if (is_phi) {
                const float e = bl_exp(-fabsf(cz)), op = 1.0f + e, l = bl_log(op);
                const float sig = (cz > 0.0f ? 1.0f : e) * bl_rcp(op);
                pe2 = 2.0f * (prior_loc * (fmaxf(-cz, 0.0f) + l) + prior_isc2 * (fmaxf(cz, 0.0f) + l));
                pg = (prior_loc + prior_isc2) * sig - prior_loc;
            }
        }
        if constexpr (MODEL == 3) {
            //   phi = log f, f ~ Exponential(rate r):  energy = r e^phi - phi  (Jacobian included),  pg = r e^phi - 1
            if (is_phi) {
                const float rf = prior_loc * bl_exp(fminf(cz, 80.0f));
                pe2 = 2.0f * (rf - cz);
                pg = rf - 1.0f;
            }
        }
        // Everything of the speculative position that does not need the gathered gradient is prepared here, in the
        // shadow of the store -> poll latency: which kind of leaf comes next, the peeked direction bit, and -- when
        // the next doubling starts from the tree's OTHER edge -- the whole position (it does not depend on this leaf).
        spec_ed = epsdir;
        if constexpr (HYBRID) {
            // the leaf in flight has index snprop: not the subtree's last (which is odd whenever there is more than one leaf), and with
            // BL_RN_HYBRID 1 even as well
            if (!init_pending && ((snprop & 1) == 0 || BL_RN_HYBRID == 2) && snprop + 1 < (1 << depth)) spec_kind = 1;
        }
        if constexpr (SPEC) {
            if (!init_pending) {
                if (snprop + 1 < (1 << depth)) spec_kind = 1;
                else if (depth + 1 < p.max_depth) {
                    BlRng peek = rng_dir; // the real draw is made by the decisions
                    const bool gr2 = (bl_rng_next(peek) >> 31) != 0u;
                    spec_ed = gr2 ? eps : -eps;
                    spec_kind = 2;
                    if (gr2 != going_right) { // that edge is in LDS, untouched by the subtree in progress
                        float rh2;
                        bl_next_leaf(gr2 ? t_zr : t_zl, gr2 ? t_rr : t_rl, gr2 ? t_gr : t_gl, spec_ed, minv, rh2, spec_other);
                        spec_kind = 3;
                    }
                }
            }
        }
        }
        __syncthreads();
        BL_STAMP(1)

        epoch++; // (the evaluation just finished was published by the compute waves under this epoch, wanted or not)
        if (redo) {
            // the decisions chose another position (transition end, U-turn, divergence, ...) or the run is over:
            // that epoch is skipped by every workgroup alike; hand the compute waves the right position
            if (act) sh_coef[my_pos] = cz;
            if (lane == 0) sh_flag[0] = flag;
            run_status = flag;
        } else {
            // ------------------------------ all-gather of the k partials (G16 / R2) ----
            const unsigned char *rbase = xbase + (epoch & (BL_XCHG_SLOTS - 1u)) * rec_bytes;
            BL_STAMP(2)
            // a first poll that misses costs a whole extra round trip: give the slowest peers' stores
            // a moment to land before looking
            for (int z = 0; z < p.first_delay; z++) __builtin_amdgcn_s_sleep(1);
            double acc = 0.0;
            bool timed_out = false;
            if (!multi_wg) {
                // ONE workgroup per chain (round 4; small problems -- simulate()'s defaults, the size of the reference's own tests): there is
                // nobody to exchange with, so the hand-off through L2 (a store, a round trip, the sums: ~1 500 cycles of a 5 000-cycle tick)
                // is skipped and the waves' rows are added straight from LDS -- complete and visible behind the barrier above.  Same fixed
                // wave order as the publishing wave's sum.
                const float *part = bl_lds_f(BL_OFF_PART) + part_pos;
                float comp = part[0];
#pragma unroll
                for (int w = 1; w < CW; w++) comp += part[w * part_rs];
                for (int sp = 1; sp < nsp; sp++) {
#pragma unroll
                    for (int w = 0; w < CW; w++) comp += part_all ? part[w * part_rs + sp * BL_SP_PART(KS, KO)] : 0.0f;
                }
                if (lane > D) comp = 0.0f;
                if (lane == D + 1 && (epoch & 255u) == 0u) // the host's abort request (fit(timeout=...))
                    comp = (__hip_atomic_load(cold->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) ? 1.0f : 0.0f;
                acc = (double)comp;
            } else if (one_batch) {
                // common shape: one round of <= 8 loads per lane covers all k records.  (Measured and dropped: two rounds in flight
                // half a round trip apart, the first complete one taken -- 4-11 % slower: the waits on the second round's loads
                // serialise behind the first's.)
                unsigned long long v[8];
                BlSpinBound bound;
                while (true) {
#ifdef BL_STAMPS
                    const long long st_r0 = (long long)clock64();
#endif
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        if (q < nq) { // wave-uniform: skip load pairs that would only re-read record k-1
                            v[q] = bl_poll_load(rbase, poff[q]);
                            v[q + 1] = bl_poll_load(rbase, poff[q + 1]);
                        } else {
                            v[q] = v[0];
                            v[q + 1] = v[0];
                        }
                    }
                    unsigned bad = 0u;
#pragma unroll
                    for (int q = 0; q < 8; q++) bad |= ((unsigned)(v[q] >> 32)) ^ epoch;
                    const bool all_in = __all(bad == 0u);
#ifdef BL_STAMPS
                    st_sub[3] += (long long)clock64() - st_r0; // cycles in poll rounds (loads issued -> tags checked)
#endif
                    if (all_in) break;
                    if (bound.expired(p.spin_limit)) { timed_out = true; break; }
                    if (!local)
                        for (int z = 0; z < p.poll_sleep; z++) __builtin_amdgcn_s_sleep(1);
                }
                BL_COUNT_SPINS(bound.spins)
#pragma unroll
                for (int q = 0; q < 8; q++) acc += (double)(pval[q] * __uint_as_float((unsigned)v[q]));
            } else {
                // Many workgroups (the wide geometry: a chain across XCDs, up to 128 records): the loads of up to BL_POLL_NB batches
                // of 8 per lane are ALL in flight before the first tag is looked at -- one round trip over the fabric for the lot.
                // (Until round 5 the batches were polled one after the other, a round trip each: the exchange of the benchmark grid's
                // top rows grew with k faster than the visit loops shrank -- profiles/r05/d_time_wide_before.txt.)  Same lanes, same
                // records, same order of the f64 sums as the batch-by-batch form.
                // (occu_rn keeps one batch at a time: its kernel sits at the 256-register budget -- tests/test_kernel_resources.py -- and its
                // chains do not leave their XCD)
                constexpr int NB = MODEL == 1 ? 1 : BL_POLL_NB;
                for (int p0 = 0; p0 < p.k && !timed_out; p0 += NB * 8 * G) {
                    unsigned long long v[NB][8];
                    BlSpinBound bound;
                    while (true) {
                        unsigned bad = 0u;
#pragma unroll
                        for (int b = 0; b < NB; b++) {
                            if (b == 0 || p0 + b * 8 * G < p.k) { // wave-uniform: batches past the last record are not loaded
#pragma unroll
                                for (int q = 0; q < 8; q++) {
                                    const int w = p0 + (b * 8 + q) * G + sub;
                                    v[b][q] = bl_poll_load(rbase, (unsigned)(((w < p.k ? w : p.k - 1) * p.pitch + c_idx) * 8));
                                }
                            } else {
#pragma unroll
                                for (int q = 0; q < 8; q++) v[b][q] = v[0][0];
                            }
                        }
#pragma unroll
                        for (int b = 0; b < NB; b++)
#pragma unroll
                            for (int q = 0; q < 8; q++) bad |= ((unsigned)(v[b][q] >> 32)) ^ epoch;
                        if (__all(bad == 0u)) break;
                        if (bound.expired(p.spin_limit)) { timed_out = true; break; }
                        if (!local)
                            for (int z = 0; z < p.poll_sleep; z++) __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int b = 0; b < NB; b++)
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            const int w = p0 + (b * 8 + q) * G + sub;
                            acc += (w < p.k) ? (double)__uint_as_float((unsigned)v[b][q]) : 0.0;
                        }
                }
            }
            // fold the 64/nvp lane groups (each summed a different subset of the workgroups): element-wise
            // across rows, by gfx950's v_permlane16_swap / v_permlane32_swap (one VALU op per 32-bit half)
            if (multi_wg) {
                if (nvp <= 16) acc = bl_fold_rows16_d(acc);
                if (nvp <= 32) acc = bl_fold_halves32_d(acc);
            }
            BL_STAMP(3)
            if (epoch == 1u && p.allow_local) {
                const double sx = bl_readlane_d(acc, D + 2), sxx = bl_readlane_d(acc, D + 3);
                local = ((double)p.k * sxx == sx * sx); // exact: small integers
                if (lane == 0) sh_flag[1] = local ? 1 : 0; // (the compute waves' stores from the next evaluation on)
            }
            const float cg = act ? (-(float)acc + pg) : 0.0f;
            p_acc = acc; p_cg = cg; p_pe2 = pe2; p_timed_out = timed_out;
            have_pending = true;
            // ---------------------------------------------------- speculative position ----
            // Where the next leaf goes depends on this one's energy / U-turn tests, but almost always it is
            // "the next leaf of the subtree" or, when the subtree is full, "the first leaf of the next doubling".
            // Both follow from this gradient in a handful of FMAs, so the compute waves start on that position
            // at once and the tests run beside them (loop head).  A wrong guess costs nothing extra: it happens
            // when a transition ends, and the decisions of that tick outlast a phase A anyway.
            cz_spec = __builtin_nanf("");
            if (!SPEC && !(HYBRID && spec_kind != 0 && !timed_out)) {
                // Long site evaluations (occu_rn; many site pairs per lane): a dropped evaluation would cost more than
                // the overlap saves, so the decisions are taken here and the compute waves get the decided position.
                decide();
                if (act) sh_coef[my_pos] = cz;
                if (lane == 0) sh_flag[0] = flag;
                run_status = flag;
                BL_STAMP_CRIT
            } else {
            if (spec_kind != 0 && !timed_out) {
                if (spec_kind == 3) cz_spec = spec_other;
                else {
                    const float cr = bl_leaf_momentum(rh, epsdir, cg);
                    float rh2;
                    bl_next_leaf(cz, cr, cg, spec_ed, minv, rh2, cz_spec); // spec_ed = epsdir for the next leaf of the subtree
                }
            }
            if (act && cz_spec == cz_spec) sh_coef[my_pos] = cz_spec;
            if (lane == 0) sh_flag[0] = 0;
            run_status = 0;
            }
            BL_STAMP(4)
        }
        __syncthreads();
        BL_STAMP(5)
        if (run_status != 0) break; // (what this wave just wrote to sh_flag[0]: no need to read it back)
    }
    }
#ifdef BL_STAMPS
    if (cold->dbg && chain == 0 && member == 0 && tid == 0) {
        for (int i = 0; i < 8; i++) cold->dbg[i] = st_acc[i];
        cold->dbg[8] = (long long)epoch;
        cold->dbg[9] = (long long)wall_clock64() - st_rt0;
        cold->dbg[10] = st_spins;
        for (int i = 0; i < 3; i++) { cold->dbg[11 + i] = st_kn[i]; }
        cold->dbg[14] = st_kc[0]; cold->dbg[15] = st_kc[1]; cold->dbg[7] = st_kc[2];
        for (int i = 0; i < 4; i++) cold->dbg[16 + i] = st_sub[i];
    }
    if (cold->dbg && chain == 0 && member < 32 && wave >= 1 && wave <= 8 && (tid & 63) == 0) {
        cold->dbg[32 + member * 8 + (wave - 1)] = st_sub[5];
        unsigned hw; // where the hardware put this wave: HW_ID bits 5:4 = SIMD, 11:8 = CU, 3:0 = wave slot
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        cold->dbg[32 + 256 + member * 8 + (wave - 1)] = (long long)hw;
    }
    if (cold->dbg && chain == 0 && member < 32 && wave == 0 && tid == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        cold->dbg[32 + 256 + member * 8 + 7] = (long long)hw; // (the control wave's, in the eighth slot)
    }
    if (cold->dbg && chain == 0 && member == 0 && tid == 64) { // first compute wave: site-evaluation passes and their cycles
        cold->dbg[20] = st_sub[4]; cold->dbg[21] = st_sub[5];
        if constexpr (MODEL == 1) for (int i = 0; i < 8; i++) cold->dbg[22 + i] = bl_rn_dbg[i];
    }
#endif
}
