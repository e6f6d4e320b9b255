// nuts_kernel.hpp -- persistent multi-CU NUTS kernel (one launch = all chains, warmup + sampling).
//
// Replaces numpyro.infer.MCMC(NUTS(occu)).run(...) as biolith/utils/fit.py:92-130 invokes it.
// The sampler follows NumPyro's algorithm: iterative tree doubling with
// checkpointed generalised U-turn checks, multinomial proposals (uniform inside a subtree, biased
// between subtrees), dual-averaging step size, windowed Welford diagonal mass matrix.
//
// Mapping to the chip
//   * one chain = k cooperating workgroups (512 threads each, one per CU); a workgroup owns a
//     contiguous slice of sites, staged ONCE into LDS as per-site records (occu_device.hpp);
//   * blockIdx -> (chain, member) is XCD-aware: blocks b and b+8 share an XCD under the observed
//     round-robin dealing, so chain c takes the blocks with b % 8 == c % 8 and its k workgroups
//     share one L2.  That is a SPEED arrangement only: every workgroup reads HW_REG_XCC_ID and the
//     chain switches to the L2-local exchange (below) only if the first, placement-independent
//     exchange proves that all k workgroups really sit on one XCD;
//   * every leapfrog ("tick"): all 8 waves evaluate their sites' log-lik + gradient from LDS ->
//     one interleaved DPP wave reduction -> LDS -> workgroup partial (D grads f32, log-lik hi+lo);
//   * the k partials are all-gathered through 8-byte {epoch, value} granules (guide G16, form R2:
//     the data is the flag), double-buffered by epoch parity, every spin bounded:
//       - placement-independent form: ONE sc1 (write-through) store per granule, relaxed
//         agent-scope (sc1) polls;
//       - L2-local form (verified same-XCD chains only): workgroup-scope stores keep the line in
//         the XCD's L2, sc1 polls bypass L1 and hit that L2: several times shorter hop;
//     two poll rounds are kept in flight, spaced a fraction of a round trip apart;
//     every workgroup sums the k records in the same fixed order in f64, so all k copies of the
//     chain state stay bit-identical without any broadcast;
//   * wave 0 of every workgroup then advances the (replicated) NUTS state machine by one leaf,
//     lane d holding dimension d, and publishes the next position to its workgroup through LDS.
// No host round trip, no kernel boundary and no HBM traffic inside the sampling loop.
#pragma once
#include "occu_device.hpp"

// In-kernel phase stamps (guide section 7 "In-kernel stamps"): diagnostic builds only
// (make stamps); the shipped kernel executes none of this.
#ifdef BL_STAMPS
#define BL_STAMP_DECL long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long st_prev = (long long)clock64(); long long st_rt0 = (long long)wall_clock64(); long long st_spins = 0;
#define BL_STAMP(i) { const long long st_now = (long long)clock64(); st_acc[i] += st_now - st_prev; st_prev = st_now; }
#define BL_COUNT_SPINS(n) st_spins += (long long)(n);
#else
#define BL_COUNT_SPINS(n)
#define BL_STAMP_DECL
#define BL_STAMP(i)
#endif

// Rarely-read launch constants and output pointers live in device memory (keeps the kernel's
// SGPR budget for the loop).
struct BlNutsCold {
    int num_warmup, num_samples, nwin;
    float target_accept;
    int win_end[32];               // numpyro adaptation windows (inclusive ends)
    float loc_b, isc2_b, loc_a, isc2_a;  // Normal prior loc, 1/scale^2
    double prior_const;            // sum_k log(scale_k) + D/2 log(2 pi)
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
    long long *dbg;                // [16] phase cycle counters (diagnostic BL_STAMPS builds only)
};

struct BlNutsParams {
    const float *rows;             // HBM data matrix [n_rows][n_stride]
    int n_sites, n_stride, T, J, Ks, Ko;
    int num_chains;
    int k;                         // workgroups per chain
    int nloc;                      // sites per workgroup
    int rec_stride;                // floats per LDS site record
    int nvp;                       // granules per workgroup record: 16, 32 or 64 (>= D+4)
    int max_depth;
    int allow_local;               // 0: always use the placement-independent exchange
    int poll_sleep;                // s_sleep(1) repeats between the two in-flight poll rounds (fabric form)
    unsigned spin_limit;
    unsigned long long *xchg;      // [C][2][k][nvp] granules, zeroed before every launch
    const BlNutsCold *cold;
};

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
__device__ __forceinline__ bool bl_is_turning(float minv, float rl, float rr, float rsum)
{
    const float rho = rsum - 0.5f * (rl + rr);
    float dl = minv * rl * rho, dr = minv * rr * rho;
    bl_wave_sum2(dl, dr);
    return (dl <= 0.0f) || (dr <= 0.0f);
}

__device__ __forceinline__ unsigned bl_xcc_id()
{
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xFu;
}

// One poll round: every lane loads its granule of 8 records (sc1: bypasses L1).  Record indices
// beyond k are clamped to k-1: the duplicate loads carry valid tags and are ignored in the sum,
// so the round needs no predication.
#define BL_POLL_ISSUE(buf)                                                                                       \
    _Pragma("unroll") for (int q = 0; q < 8; q++) {                                                              \
        const int w_ = min(p0 + q * G + sub, p.k - 1);                                                           \
        buf[q] = __hip_atomic_load(rec + (size_t)w_ * nvp + c_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  \
    }
#define BL_POLL_CHECK(buf, okv)                                                                                  \
    {                                                                                                            \
        unsigned bad_ = 0u;                                                                                      \
        _Pragma("unroll") for (int q = 0; q < 8; q++) bad_ |= ((unsigned)(buf[q] >> 32)) ^ epoch;                \
        okv = __all(bad_ == 0u);                                                                                 \
    }

template <int KS, int KO, bool LDS>
__global__ void __launch_bounds__(BL_THREADS) bl_nuts_kernel(const BlNutsParams p)
{
    // XCD-aware mapping (speed only; see header): label = b % 8 names a set of blocks that share an XCD
    const int label = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int chain = label + 8 * (slot / p.k), member = slot % p.k;
    if (chain >= p.num_chains) return; // whole block leaves before any barrier or exchange
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Ks = p.Ks, Ko = p.Ko, D = Ks + Ko + 2;
    const int T = p.T, J = p.J;
    const int s0 = member * p.nloc;
    int cnt = p.n_sites - s0;
    cnt = cnt < 0 ? 0 : (cnt > p.nloc ? p.nloc : cnt);
    const float *grows = nullptr;
    int ld = p.rec_stride;
    if constexpr (LDS) {
        bl_stage_records(p.rows, p.n_stride, s0, cnt, T, J, KS, KO, p.rec_stride);
    } else {
        grows = p.rows + s0;
        ld = p.n_stride;
    }
    float *sh_coef = bl_lds_f(BL_OFF_COEF);
    int *sh_flag = bl_lds_i(BL_OFF_FLAG);
    float *sh_ckr = bl_lds_f(BL_OFF_CKR), *sh_ckrs = bl_lds_f(BL_OFF_CKRS);
    if (tid < 64) sh_coef[tid] = 0.0f;
    __syncthreads();

    // ------------------------------------------------ replicated chain state (wave 0) ----
    const BlNutsCold *cold = p.cold;
    const bool act = lane < D;
    // where lane d's coefficient / partial lives in the padded LDS layouts; lanes >= D read the log-lik
    const int my_pos = act ? bl_coef_pos(lane, Ks, KS) : KS + KO + 2;
    int S = 0, W = 0, total = 0;
    BlRng rng_d, rng_s; // per-dimension stream, shared scalar stream
    float th = 0.f, gr = 0.f;           // current position / gradient of U
    double U = 0.0;                     // current potential
    float minv = act ? 1.0f : 0.0f;     // diagonal inverse mass (0 in idle lanes keeps sums clean)
    float eps = 1.0f;
    // dual averaging + Welford
    float da_xt = 0.f, da_xavg = 0.f, da_gavg = 0.f, da_prox = 2.302585093f;
    int da_t = 0, wf_n = 0, win_idx = 0;
    float wf_mean = 0.f, wf_m2 = 0.f;
    // tree
    double E0 = 0.0, Up = 0.0;
    float zl = 0, rl = 0, gl = 0, zr = 0, rr = 0, gR = 0, zp = 0, gp = 0, rsum = 0, wt = 0, sumacc = 0;
    int depth = 0, nprop = 0;
    bool diverged = false;
    // subtree under construction
    double sUp = 0.0;
    float szp = 0, sgp = 0, srsum = 0, swt = 0, ssumacc = 0;
    int snprop = 0;
    bool sdiv = false, sturn = false, going_right = false;
    float epsdir = 0.f;
    // leaf in flight
    float cz = 0, rh = 0;
    int it = -1; // -1: evaluating the initial position
    long long nleap_w = 0, nleap_s = 0;
    bool local = false; // L2-local exchange proven safe for this chain
    float prior_loc = 0.f, prior_isc2 = 0.f;
    double prior_const = 0.0;
    const float xcc = (float)bl_xcc_id();

    if (wave == 0) {
        S = cold->num_samples; W = cold->num_warmup; total = W + S;
        prior_loc = (lane <= Ks) ? cold->loc_b : cold->loc_a;
        prior_isc2 = act ? ((lane <= Ks) ? cold->isc2_b : cold->isc2_a) : 0.0f;
        prior_const = cold->prior_const;
        const uint32_t *rs = cold->rng + ((size_t)chain * BL_NSTREAM + lane) * 4;
        rng_d.s0 = rs[0]; rng_d.s1 = rs[1]; rng_d.s2 = rs[2]; rng_d.s3 = rs[3];
        const uint32_t *rc = cold->rng + ((size_t)chain * BL_NSTREAM + BL_SCALAR_STREAM) * 4;
        rng_s.s0 = rc[0]; rng_s.s1 = rc[1]; rng_s.s2 = rc[2]; rng_s.s3 = rc[3];
        const float u0 = bl_rng_uniform(rng_d); // init_to_uniform(radius=2), fit.py:93
        const float *init = cold->init_theta;
        cz = act ? (init ? init[chain * D + lane] : 4.0f * u0 - 2.0f) : 0.0f;
        if (act) sh_coef[my_pos] = cz;
        if (lane == 0) sh_flag[0] = 0;
    }
    __syncthreads();

    unsigned epoch = 0;
    BL_STAMP_DECL
    while (true) {
        // ------------------------------------------- phase A: all waves, site log-lik ----
        float beta[KS + 1], alpha[KO + 1];
        bl_load_coefs<KS, KO>(beta, alpha);
        BL_STAMP(6)
        float ll = 0.0f, gb[KS + 1], ga[KO + 1];
#pragma unroll
        for (int k = 0; k <= KS; k++) gb[k] = 0.0f;
#pragma unroll
        for (int k = 0; k <= KO; k++) ga[k] = 0.0f;
        bl_eval_sites<KS, KO, LDS>(grows, ld, cnt, T, J, beta, alpha, ll, gb, ga);
        BL_STAMP(7)
        bl_wave_partials_to_lds<KS, KO>(ll, gb, ga);
        BL_STAMP(0)
        __syncthreads();
        BL_STAMP(1)

        if (wave == 0) {
            epoch++;
            // ---------------------------------- workgroup partial (fixed wave order) ----
            const float *part = bl_lds_f(BL_OFF_PART) + my_pos;
            float comp = 0.0f; // lanes < D: their gradient component; lane D: the log-lik
#pragma unroll
            for (int w = 0; w < BL_WAVES; w++) comp += part[w * BL_PART_STRIDE];
            if (lane > D) comp = 0.0f;
            if (lane == D + 1 && member == 0 && (epoch & 255u) == 0u)
                comp = (__hip_atomic_load(cold->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) ? 1.0f : 0.0f;
            if (epoch == 1u) { // placement census: all k XCC ids equal  <=>  k * sum(x^2) == (sum x)^2
                if (lane == D + 2) comp = xcc;
                if (lane == D + 3) comp = xcc * xcc;
            }

            // ------------------------------ all-gather of the k partials (G16 / R2) ----
            const int nvp = p.nvp, G = 64 / nvp;
            unsigned long long *rec = p.xchg + ((size_t)(chain * 2 + (epoch & 1u)) * p.k) * nvp;
            const unsigned long long granule = ((unsigned long long)epoch << 32) | __float_as_uint(comp);
            if (lane < nvp) {
                if (local) // line stays in this XCD's L2, where every consumer of this chain polls it
                    __hip_atomic_store(rec + (size_t)member * nvp + lane, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else       // write-through: visible to any XCD
                    __hip_atomic_store(rec + (size_t)member * nvp + lane, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            BL_STAMP(2)
            const int c_idx = lane & (nvp - 1), sub = lane / nvp;
            const int nsleep = local ? 1 : p.poll_sleep;
            double acc = 0.0;
            bool timed_out = false;
            for (int p0 = 0; p0 < p.k; p0 += 8 * G) {
                unsigned long long va[8], vb[8];
                unsigned spins = 0;
                bool ok;
                BL_POLL_ISSUE(va)
                while (true) {
                    for (int z = 0; z < nsleep; z++) __builtin_amdgcn_s_sleep(1);
                    BL_POLL_ISSUE(vb)
                    BL_POLL_CHECK(va, ok)
                    if (ok) break;
                    for (int z = 0; z < nsleep; z++) __builtin_amdgcn_s_sleep(1);
                    BL_POLL_ISSUE(va)
                    BL_POLL_CHECK(vb, ok)
                    if (ok) {
#pragma unroll
                        for (int q = 0; q < 8; q++) va[q] = vb[q];
                        break;
                    }
                    if (++spins > p.spin_limit) { timed_out = true; break; }
                }
                BL_COUNT_SPINS(spins)
                if (timed_out) break;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int w = p0 + q * G + sub;
                    acc += (w < p.k) ? (double)__uint_as_float((unsigned)va[q]) : 0.0;
                }
            }
            for (int off = nvp; off < 64; off <<= 1) {
                const unsigned long long b = (unsigned long long)__double_as_longlong(acc);
                const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)b, off);
                const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(b >> 32), off);
                acc += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            }
            BL_STAMP(3)
            const double ll_tot = bl_readlane_d(acc, D);
            const bool abort_req = bl_readlane_d(acc, D + 1) != 0.0;
            if (epoch == 1u && p.allow_local) {
                const double sx = bl_readlane_d(acc, D + 2), sxx = bl_readlane_d(acc, D + 3);
                local = ((double)p.k * sxx == sx * sx); // exact: small integers
            }
            int flag = 0;
            if (timed_out) flag = 4;       // BL_ERR_TIMEOUT
            else if (abort_req) flag = 5;  // BL_ERR_ABORTED

            // ------------------------------------------------ potential at cz (lane d) ----
            const float dth = cz - prior_loc;
            const float cg = act ? (-(float)acc + dth * prior_isc2) : 0.0f;

            bool new_transition = false;
            if (flag == 0) {
                if (it < 0) {
                    // initial evaluation done
                    const double Un = -ll_tot + (double)(0.5f * bl_wave_sum(dth * dth * prior_isc2)) + prior_const;
                    th = cz; gr = cg; U = Un;
                    it = 0;
                    new_transition = true;
                } else {
                    if (it < W) nleap_w++; else nleap_s++;
                    // ---------------- finish the leaf (_build_basetree) ----------------
                    const float cr = rh - 0.5f * epsdir * cg;
                    float s_prior = dth * dth * prior_isc2, s_kin = minv * cr * cr;
                    bl_wave_sum2(s_prior, s_kin);
                    const double Un = -ll_tot + (double)(0.5f * s_prior) + prior_const;
                    const double Kn = (double)(0.5f * s_kin);
                    double dE = (Un + Kn) - E0;
                    if (dE != dE) dE = (double)INFINITY;
                    const float dEf = (float)dE;
                    const float lw = -dEf;
                    const bool ldiv = dE > 1000.0;
                    const float lacc = dEf <= 0.0f ? 1.0f : bl_exp(-dEf);
                    const int leaf_idx = snprop;
                    if (leaf_idx == 0) {
                        szp = cz; sgp = cg; sUp = Un; swt = lw; srsum = cr;
                        sdiv = ldiv; ssumacc = lacc; snprop = 1;
                    } else {
                        // _combine_tree(..., biased=False): uniform transition kernel
                        float pr, lse;
                        bl_merge_weights(swt, lw, lse, pr); // pr = expit(lw - swt)
                        const float u = bl_rng_uniform(rng_s);
                        if (u < pr) { szp = cz; sgp = cg; sUp = Un; }
                        swt = lse;
                        sdiv = ldiv;
                        ssumacc += lacc;
                        srsum += cr;
                        snprop++;
                    }
                    // checkpointed U-turn (_leaf_idx_to_ckpt_idxs / _is_iterative_turning)
                    const int idx_max = __popc((unsigned)leaf_idx >> 1);
                    const int idx_min = idx_max - (int)__builtin_ctz(~(unsigned)leaf_idx) + 1;
                    if ((leaf_idx & 1) == 0) {
                        sh_ckr[idx_max * 64 + lane] = cr;
                        sh_ckrs[idx_max * 64 + lane] = srsum;
                    } else {
                        for (int i = idx_max; i >= idx_min && !sturn; i--) {
                            const float ck = sh_ckr[i * 64 + lane];
                            const float srs = srsum - sh_ckrs[i * 64 + lane] + ck;
                            sturn = bl_is_turning(minv, ck, cr, srs);
                        }
                    }
                    if (snprop < (1 << depth) && !sturn && !sdiv) {
                        // next leaf continues from this one
                        rh = cr - 0.5f * epsdir * cg;
                        cz = cz + epsdir * minv * rh;
                    } else {
                        // ---------- _combine_tree(tree, subtree, biased=True) ----------
                        if (going_right) { zr = cz; rr = cr; gR = cg; }
                        else { zl = cz; rl = cr; gl = cg; }
                        rsum += srsum;
                        float pr = fminf(1.0f, bl_exp(swt - wt));
                        if (sturn || sdiv) pr = 0.0f;
                        const bool turning = bl_is_turning(minv, rl, rr, rsum);
                        const float u = bl_rng_uniform(rng_s);
                        if (u < pr) { zp = szp; gp = sgp; Up = sUp; }
                        depth++;
                        wt = bl_logaddexp(wt, swt);
                        diverged = sdiv;
                        sumacc += ssumacc;
                        nprop += snprop;
                        if (depth < p.max_depth && !turning && !diverged) {
                            // next doubling
                            going_right = (bl_rng_next(rng_s) >> 31) != 0u;
                            epsdir = going_right ? eps : -eps;
                            snprop = 0; sturn = false; sdiv = false;
                            const float ez = going_right ? zr : zl, er = going_right ? rr : rl, eg = going_right ? gR : gl;
                            rh = er - 0.5f * epsdir * eg;
                            cz = ez + epsdir * minv * rh;
                        } else {
                            // ---------------- transition complete ----------------
                            const float accp = sumacc * bl_rcp((float)nprop);
                            th = zp; gr = gp; U = Up;
                            if (it < W) {
                                // warmup_adapter.update_fn: dual averaging (t0=10, kappa=.75, gamma=.05)
                                const float g = cold->target_accept - accp;
                                da_t += 1;
                                const float tt = (float)da_t;
                                const float rt10 = bl_rcp(tt + 10.0f);
                                da_gavg = (1.0f - rt10) * da_gavg + g * rt10;
                                da_xt = da_prox - __builtin_amdgcn_sqrtf(tt) * 20.0f * da_gavg;
                                const float wgt = __builtin_amdgcn_exp2f(-0.75f * __builtin_amdgcn_logf(tt));
                                da_xavg = (1.0f - wgt) * da_xavg + wgt * da_xt;
                                eps = bl_exp((it == W - 1) ? da_xavg : da_xt);
                                eps = fminf(fmaxf(eps, 1.1754944e-38f), 3.4028235e+38f);
                                const bool middle = win_idx > 0 && win_idx < cold->nwin - 1;
                                if (middle) {
                                    wf_n += 1;
                                    const float dpre = th - wf_mean;
                                    wf_mean += dpre * bl_rcp((float)wf_n);
                                    wf_m2 += dpre * (th - wf_mean);
                                }
                                const bool at_end = it == cold->win_end[win_idx];
                                if (at_end) win_idx++;
                                if (at_end && middle) {
                                    const float n = (float)wf_n;
                                    const float var = wf_m2 * bl_rcp(n - 1.0f), rn5 = bl_rcp(n + 5.0f);
                                    minv = act ? (n * rn5 * var + 1e-3f * 5.0f * rn5) : 0.0f;
                                    wf_mean = 0.f; wf_m2 = 0.f; wf_n = 0;
                                    da_xt = 0.f; da_xavg = 0.f; da_gavg = 0.f; da_t = 0;
                                    da_prox = bl_log(10.0f * eps);
                                }
                            } else if (member == 0) {
                                const size_t s = (size_t)chain * S + (it - W);
                                if (act) cold->draws[s * D + lane] = th;
                                if (lane == 0) {
                                    cold->num_steps[s] = nprop;
                                    cold->accept_prob[s] = accp;
                                    cold->diverging[s] = diverged ? 1 : 0;
                                    cold->potential[s] = (float)U;
                                }
                            }
                            it++;
                            if (it >= total) flag = 1; // done
                            else new_transition = true;
                        }
                    }
                }
                if (new_transition) {
                    // sample momentum r = N(0,1)/sqrt(M^-1); start a fresh tree and its first doubling
                    const float z01 = bl_rng_normal(rng_d);
                    const float r0 = act ? z01 * __builtin_amdgcn_rsqf(minv) : 0.0f;
                    E0 = U + (double)(0.5f * bl_wave_sum(minv * r0 * r0));
                    zl = th; rl = r0; gl = gr; zr = th; rr = r0; gR = gr;
                    zp = th; gp = gr; Up = U;
                    wt = 0.f; rsum = r0; sumacc = 0.f; nprop = 0; depth = 0; diverged = false;
                    going_right = (bl_rng_next(rng_s) >> 31) != 0u;
                    epsdir = going_right ? eps : -eps;
                    snprop = 0; sturn = false; sdiv = false;
                    rh = r0 - 0.5f * epsdir * gr;
                    cz = th + epsdir * minv * rh;
                }
            }
            if (act) sh_coef[my_pos] = cz;
            if (lane == 0) sh_flag[0] = flag;
            if (flag != 0 && member == 0) {
                if (flag > 1 && lane == 0) atomicMax(cold->status, flag);
                if (act) cold->inv_mass[chain * D + lane] = minv;
                if (lane == 0) {
                    cold->step_size[chain] = eps;
                    cold->nleap[chain * 2 + 0] = nleap_w;
                    cold->nleap[chain * 2 + 1] = nleap_s;
                    cold->xcd_local[chain] = local ? 1 : 0;
                }
            }
            BL_STAMP(4)
        }
        __syncthreads();
        BL_STAMP(5)
        if (sh_flag[0] != 0) break;
    }
#ifdef BL_STAMPS
    if (cold->dbg && chain == 0 && member == 0 && tid == 0) {
        for (int i = 0; i < 8; i++) cold->dbg[i] = st_acc[i];
        cold->dbg[8] = (long long)epoch;
        cold->dbg[9] = (long long)wall_clock64() - st_rt0;
        cold->dbg[10] = st_spins;
    }
#endif
}
