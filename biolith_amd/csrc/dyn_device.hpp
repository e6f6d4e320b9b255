// dyn_device.hpp -- dynamic (multi-season) occupancy: site evaluation for gfx950.  BUILDER-DEFINED MODEL, NO REFERENCE COUNTERPART.
//
// BASELINE.json configs[4] asks for "multi-season dynamic occupancy (colonisation / extinction), forward-algorithm HIP kernel";
// timmh/biolith has no such model (its periods share one psi, biolith/models/occu.py:198-210).  The model
// built here is the standard one (MacKenzie et al. 2003) put together from the reference's own pieces -- LinearRegression
// predictors with Normal priors (regression/linear.py:28-66), the occupancy model's detection layer, masking and numpyro clamp
// (occu.py:136-142, 221-242):
//     z_i1 ~ Bernoulli(psi_i),                    logit psi_i   = x_i b_psi
//     z_i,t+1 | z_it = 0 ~ Bernoulli(gamma_i),    logit gamma_i = x_i b_gamma      (colonisation)
//     z_i,t+1 | z_it = 1 ~ Bernoulli(1 - eps_i),  logit eps_i   = x_i b_eps        (extinction)
//     y_itj | z_it ~ Bernoulli(z_it p_itj),       logit p_itj   = w_itj alpha      (a detection at z = 0 costs log tiny_f32)
// theta = [b_psi | b_gamma | b_eps (Ks + 1 each) | alpha (Ko + 1)].  Its parity is against the builder's own oracle only
// (oracle/occu_oracle.c: potential_grad_dyn, pinned by brute force over the 2^T paths and by central differences).
//
// Data: the plain occupancy model's rows and LDS pair records, unchanged (per period: the sign-folded visits, ka, kb).
// Per site pair and lane, everything in registers except 2 T pairs of floats in a lane-private LDS column:
//   pass 1   per period: a_t = sum_j log sigma(u_j) + ka  (log P(y_t | z_t = 1)), kb_t (log P(y_t | z_t = 0)); the scaled FORWARD
//            recursion  pi_1 = psi,  phi_t = P(z_t = 1 | y_1..t) = sigmoid(log pi_t + a_t - log(1 - pi_t) - kb_t),
//            pi_t+1 = phi_t (1 - eps) + (1 - phi_t) gamma;  the log-likelihood is the sum of the normalisers;  (phi_t, pi_t+1) -> LDS;
//   backward rho_T = phi_T,  xi_t(a, b) = rho_t+1(b) phi_t(a) P(b | a) / pi_t+1(b),  rho_t = xi(1,1) + xi(1,0):
//            d/d eta_gamma = sum_t xi(0,1) (1 - gamma) - xi(0,0) gamma,  d/d eta_eps = sum_t xi(1,0) (1 - eps) - xi(1,1) eps,
//            d/d eta_psi = rho_1 - psi;  rho_t -> LDS;
//   pass 2   per period and visit: d/d alpha += rho_t sigma(-u_j) (c, c w)_j   (the visits are evaluated a second time: keeping
//            T (Ko + 1) sums per site would not fit registers for general T).
// A site pair is worked on by a GROUP of G = 1, 2, 4 or 8 neighbouring lanes (host: as many as the chain's lanes allow, <= T): lane g
// of the group takes the periods t = g, g + G, ... in both visit passes -- they are nine tenths of the transcendentals -- and the
// recursions between them, which need every period in order, are run by all G lanes alike on the visit sums exchanged through LDS.
#pragma once
#ifndef BL_DYN_SCAN
#define BL_DYN_SCAN 1 // the two-scans form (bl_eval_sites_dyn_scan) where a lane owns ONE period; 0: the scaled recursions everywhere (A/B)
#endif

// bytes of LDS behind the staged records: two float2 per period and compute lane (the lane's forward / smoothed columns), and one
// more per period and lane GROUP for the periods' visit sums that the lanes of a group hand each other (host: choose_geometry)
// (round 4: two more float2 per period and lane group -- the scaled likelihoods of bl_eval_sites_dyn_scaled; sized for one lane per group)
__host__ __device__ inline int bl_dyn_scratch_bytes(int T, int cw) { return T * 4 * 8 * cw * 64; }
// coefficient / partial-sum layout of MODEL 8 (LDS coefficient block and a wave's row of the partial table alike):
// block b in {psi 0, gamma 1, eps 2}: coefficient k at b (KS + 1) + k;  alpha_k at 3 (KS + 1) + k;  the log-lik at 3 (KS + 1) + KO + 1
#define BL_DYN_OA(KS) (3 * ((KS) + 1))
#define BL_DYN_LL(KS, KO) (3 * ((KS) + 1) + (KO) + 1)

__device__ __forceinline__ bl_f2 bl_sigmoid2(bl_f2 x)
{
    const bl_f2 e = bl_exp2_2(__builtin_elementwise_abs(x) * bl2(-BL_LOG2E));
    return bl_sel_pos_one(x, e) * bl_rcp_2(e + bl2(1.0f));
}

// Accumulates over this thread's site PAIRS m = ct, ct + CT, ...: ll, gb[b][k] = d ll / d (block b's coefficient k), ga[k] = d ll / d alpha_k
template <int KS, int KO, int CT>
__device__ __forceinline__ void bl_eval_sites_dyn(int ct, int pstride, int cnt, int T, int J, int G, int scratch_off,
                                                  const float (&bpsi)[KS + 1], const float (&bgam)[KS + 1], const float (&beps)[KS + 1],
                                                  const float (&alpha)[KO + 1],
                                                  float &ll, float (&gb)[3][KS + 1], float (&ga)[KO + 1])
{
    constexpr int XQ = (KS + 3) & ~3;
    const int pb = bl_period_block(J, KO);
    const float *data = bl_lds_f(BL_OFF_DATA);
    float2 *col = reinterpret_cast<float2 *>(bl_smem_raw + scratch_off) + ct; // element (t, which) of this lane: col[(2 t + which) * CT]
    const int sub = ct & (G - 1), slot = ct / G, nslots = CT / G;              // lane `sub` of group `slot`
    float2 *acol = reinterpret_cast<float2 *>(bl_smem_raw + scratch_off) + 2 * T * CT + slot; // period t's visit sum of this group: acol[t * nslots]
    const float first = sub == 0 ? 1.0f : 0.0f;                               // (per-site sums are counted by the group's first lane)
    const int npairs = (cnt + 1) >> 1;
    const int rounds = (npairs + nslots - 1) / nslots;
    bl_f2 ll2 = bl2(0.0f), gb2[3][KS + 1], ga2[KO + 1];
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int k = 0; k <= KS; k++) gb2[b][k] = bl2(0.0f);
#pragma unroll
    for (int k = 0; k <= KO; k++) ga2[k] = bl2(0.0f);
    for (int rd = 0; rd < rounds; rd++) {
        // (every lane of a wave stays in the loop -- the groups exchange through LDS between wave-level fences; a group without a
        // pair in the last round re-evaluates the slice's last pair and is masked out)
        const int m_raw = rd * nslots + slot;
        const int m = min(m_raw, npairs - 1);
        const float live = m_raw < npairs ? 1.0f : 0.0f;
        const float2 *rec = reinterpret_cast<const float2 *>(data + (size_t)m * pstride);
        const bl_f2 vmask = bl_f2{live * first, (2 * m + 1 < cnt) ? live * first : 0.0f}; // odd slice: the last pair's second site is a dummy
        bl_f2 x[KS > 0 ? KS : 1];
        bl_f2 e_psi = bl2(bpsi[0]), e_gam = bl2(bgam[0]), e_eps = bl2(beps[0]);
#pragma unroll
        for (int k = 0; k < KS; k++) {
            const float2 v = rec[k];
            x[k] = bl_f2{v.x, v.y};
            e_psi = bl_fma2(x[k], bl2(bpsi[k + 1]), e_psi);
            e_gam = bl_fma2(x[k], bl2(bgam[k + 1]), e_gam);
            e_eps = bl_fma2(x[k], bl2(beps[k + 1]), e_eps);
        }
        const bl_f2 gam = bl_sigmoid2(e_gam), eps = bl_sigmoid2(e_eps);
        // log psi, log(1 - psi) in their exact forms (a site far in a tail of psi keeps its relative precision)
        const bl_f2 ee = bl_exp2_2(__builtin_elementwise_abs(e_psi) * bl2(-BL_LOG2E)), op = ee + bl2(1.0f);
        const bl_f2 lop = bl_log2_2(op) * bl2(BL_LN2);
        const bl_f2 psi = bl_sel_pos_one(e_psi, ee) * bl_rcp_2(op);
        bl_f2 lpi = __builtin_elementwise_min(e_psi, bl2(0.0f)) - lop;           // log pi_1
        bl_f2 l1m = __builtin_elementwise_min(-e_psi, bl2(0.0f)) - lop;          // log(1 - pi_1)
        const bl_f2 stay = bl2(1.0f) - eps - gam;                                // pi_t+1 = gamma + phi_t (1 - eps - gamma)
        bl_f2 lsite = bl2(0.0f), phi = bl2(0.0f);
        // ---- pass 1: the periods' visit sums (this lane's share of the periods), handed to the group through LDS ----
        for (int t = sub; t < T; t += G) {
            const float2 *pp = rec + XQ + t * pb;
            const float2 a_ = pp[J * (KO + 1)];
            bl_f2 a = bl_f2{a_.x, a_.y};                       // ka: cancels the log sigma(0) of the masked visits
#pragma unroll 2
            for (int j = 0; j < J; j++) {
                const float2 v0 = pp[j * (KO + 1)];
                bl_f2 u = bl_f2{v0.x, v0.y} * bl2(alpha[0]);
#pragma unroll
                for (int k = 1; k <= KO; k++) {
                    const float2 v = pp[j * (KO + 1) + k];
                    u = bl_fma2(bl_f2{v.x, v.y}, bl2(alpha[k]), u);
                }
                a = bl_fma2(bl_log2_2(bl_expneg_2(u) + bl2(1.0f)), bl2(-BL_LN2), a); // log sigma(u) = -log(1 + e^-u)
            }
            acol[t * nslots] = make_float2(a.x, a.y);
        }
        if (G > 1) bl_wave_lds_fence();
        // ---- the forward recursion (every lane of the group, all periods) ----
        for (int t = 0; t < T; t++) {
            const float2 a_ = acol[t * nslots], kb_ = (rec + XQ + t * pb)[J * (KO + 1) + 1];
            const bl_f2 a = bl_f2{a_.x, a_.y}, kb = bl_f2{kb_.x, kb_.y};
            const bl_f2 A = lpi + a, B = l1m + kb;
            const bl_f2 d = A - B;
            const bl_f2 e_d = bl_exp2_2(__builtin_elementwise_abs(d) * bl2(-BL_LOG2E)), op_d = e_d + bl2(1.0f);
            lsite += bl_fma2(bl_log2_2(op_d), bl2(BL_LN2), __builtin_elementwise_max(A, B));
            phi = bl_sel_pos_one(d, e_d) * bl_rcp_2(op_d);     // P(z_t = 1 | y_1..t)
            const bl_f2 pin = bl_fma2(phi, stay, gam);         // P(z_t+1 = 1 | y_1..t)
            col[(2 * t) * CT] = make_float2(phi.x, phi.y);
            col[(2 * t + 1) * CT] = make_float2(pin.x, pin.y);
            lpi = bl_log2_2(pin) * bl2(BL_LN2);
            l1m = bl_log2_2(bl2(1.0f) - pin) * bl2(BL_LN2);
        }
        // ---- backward: the smoothed marginals rho_t (left in LDS over phi_t) and the transitions' gradient sums ----
        bl_f2 rho = phi, d_gam = bl2(0.0f), d_eps = bl2(0.0f);
        for (int t = T - 2; t >= 0; t--) {
            const float2 f_ = col[(2 * t) * CT], p_ = col[(2 * t + 1) * CT];
            const bl_f2 f = bl_f2{f_.x, f_.y}, p1 = bl_f2{p_.x, p_.y};
            // w1 = rho / pi_t+1, w0 = (1 - rho) / (1 - pi_t+1)   (0 where the denominator is: that state cannot be reached)
            const bl_f2 r1 = bl_rcp_2(p1), r0 = bl_rcp_2(bl2(1.0f) - p1);
            const bl_f2 w1 = bl_f2{p1.x > 0.0f ? rho.x * r1.x : 0.0f, p1.y > 0.0f ? rho.y * r1.y : 0.0f};
            const bl_f2 w0 = bl_f2{p1.x < 1.0f ? (1.0f - rho.x) * r0.x : 0.0f, p1.y < 1.0f ? (1.0f - rho.y) * r0.y : 0.0f};
            const bl_f2 nf = bl2(1.0f) - f;
            const bl_f2 xi11 = w1 * f * (bl2(1.0f) - eps), xi01 = w1 * nf * gam;
            const bl_f2 xi10 = w0 * f * eps, xi00 = w0 * nf * (bl2(1.0f) - gam);
            d_gam += xi01 * (bl2(1.0f) - gam) - xi00 * gam;
            d_eps += xi10 * (bl2(1.0f) - eps) - xi11 * eps;
            col[(2 * (t + 1)) * CT] = make_float2(rho.x, rho.y);
            rho = xi11 + xi10;
        }
        // ---- pass 2: d/d alpha = sum_t rho_t sum_j sigma(-u_j) (c, c w)_j   (this lane's share of the periods) ----
        for (int t = sub; t < T; t += G) {
            const float2 *pp = rec + XQ + t * pb;
            bl_f2 rt = rho;                                     // t = 0: still in registers
            if (t > 0) { const float2 r_ = col[(2 * t) * CT]; rt = bl_f2{r_.x, r_.y}; }
            rt *= bl2(live);
#pragma unroll 2
            for (int j = 0; j < J; j++) {
                bl_f2 w[KO + 1];
#pragma unroll
                for (int k = 0; k <= KO; k++) {
                    const float2 v = pp[j * (KO + 1) + k];
                    w[k] = bl_f2{v.x, v.y};
                }
                bl_f2 u = w[0] * bl2(alpha[0]);
#pragma unroll
                for (int k = 1; k <= KO; k++) u = bl_fma2(w[k], bl2(alpha[k]), u);
                const bl_f2 tt = bl_expneg_2(u);
                const bl_f2 s = rt * tt * bl_rcp_2(tt + bl2(1.0f)); // rho_t sigma(-u)   (a masked visit or the dummy site: w = 0)
#pragma unroll
                for (int k = 0; k <= KO; k++) ga2[k] = bl_fma2(s, w[k], ga2[k]);
            }
        }
        if (G > 1) bl_wave_lds_fence(); // (the next round's visit sums overwrite this round's)
        ll2 = bl_fma2(lsite, vmask, ll2);
        const bl_f2 dv[3] = {(rho - psi) * vmask, d_gam * vmask, d_eps * vmask};
#pragma unroll
        for (int b = 0; b < 3; b++) {
            gb2[b][0] += dv[b];
#pragma unroll
            for (int k = 0; k < KS; k++) gb2[b][k + 1] = bl_fma2(dv[b], x[k], gb2[b][k + 1]);
        }
    }
    ll += ll2.x + ll2.y;
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int k = 0; k <= KS; k++) gb[b][k] += gb2[b][k].x + gb2[b][k].y;
#pragma unroll
    for (int k = 0; k <= KO; k++) ga[k] += ga2[k].x + ga2[k].y;
}

// Round 4: the same sums for at most TWO periods per lane of a group (T <= 2 G: the host picks G accordingly), restructured around
// what the first form spent its time on -- (a) the visits were evaluated twice (log-likelihood, then the gradient once the smoothed
// marginals were known): a lane now keeps its periods' gradient sums g_t[k] = sum_j sigma(-u_j) (c, c w)_j in registers and scales
// them by rho_t at the end, d/d alpha = sum_t rho_t g_t; (b) every step of the forward recursion was a chain of six dependent
// transcendentals (log pi, log(1 - pi), exp, log, rcp, ...): the recursion now runs on SCALED likelihoods, the standard normalised
// forward algorithm -- E1_t = exp(a_t - m_t), E0_t = exp(kb_t - m_t), m_t = max(a_t, kb_t), computed by the period's lane beside its
// visit pass -- so that a step is  c = pi E1 + (1 - pi) E0,  phi = pi E1 / c,  pi' = gamma + phi (1 - eps - gamma): ONE reciprocal on
// the chain; the log-likelihood sum_t (m_t + log c_t) is added by the lane that owns period t (its own one or two logs, off the chain).
// The backward recursion is the first form's.  Same records, same scratch region (two more float2 per period and group).
template <int KS, int KO, int CT>
__device__ __forceinline__ void bl_eval_sites_dyn_scaled(int ct, int pstride, int cnt, int T, int J, int G, int scratch_off,
                                                         const float (&bpsi)[KS + 1], const float (&bgam)[KS + 1], const float (&beps)[KS + 1],
                                                         const float (&alpha)[KO + 1],
                                                         float &ll, float (&gb)[3][KS + 1], float (&ga)[KO + 1])
{
    constexpr int XQ = (KS + 3) & ~3;
    const int pb = bl_period_block(J, KO);
    const float *data = bl_lds_f(BL_OFF_DATA);
    float2 *col = reinterpret_cast<float2 *>(bl_smem_raw + scratch_off) + ct; // lane-private: element (t, which): col[(2 t + which) * CT]
    const int sub = ct & (G - 1), slot = ct / G, nslots = CT / G;
    float2 *ecol = reinterpret_cast<float2 *>(bl_smem_raw + scratch_off) + 2 * T * CT + slot; // the group's (E1_t, E0_t): ecol[(2 t + which) * nslots]
    const float first = sub == 0 ? 1.0f : 0.0f; // (per-site sums, formed by every lane of the group alike, are counted by its first lane)
    const int npairs = (cnt + 1) >> 1;
    const int rounds = (npairs + nslots - 1) / nslots;
    bl_f2 ll2 = bl2(0.0f), gb2[3][KS + 1], ga2[KO + 1];
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int k = 0; k <= KS; k++) gb2[b][k] = bl2(0.0f);
#pragma unroll
    for (int k = 0; k <= KO; k++) ga2[k] = bl2(0.0f);
    for (int rd = 0; rd < rounds; rd++) {
        const int m_raw = rd * nslots + slot;
        const int m = min(m_raw, npairs - 1);
        const float live = m_raw < npairs ? 1.0f : 0.0f;
        const float2 *rec = reinterpret_cast<const float2 *>(data + (size_t)m * pstride);
        const float second = (2 * m + 1 < cnt) ? live : 0.0f; // odd slice: the last pair's second site is a dummy
        const bl_f2 vm_own = bl_f2{live, second}, vm_site = bl_f2{live * first, second * first};
        bl_f2 x[KS > 0 ? KS : 1];
        bl_f2 e_psi = bl2(bpsi[0]), e_gam = bl2(bgam[0]), e_eps = bl2(beps[0]);
#pragma unroll
        for (int k = 0; k < KS; k++) {
            const float2 v = rec[k];
            x[k] = bl_f2{v.x, v.y};
            e_psi = bl_fma2(x[k], bl2(bpsi[k + 1]), e_psi);
            e_gam = bl_fma2(x[k], bl2(bgam[k + 1]), e_gam);
            e_eps = bl_fma2(x[k], bl2(beps[k + 1]), e_eps);
        }
        const bl_f2 gam = bl_sigmoid2(e_gam), eps = bl_sigmoid2(e_eps), psi = bl_sigmoid2(e_psi);
        const bl_f2 stay = bl2(1.0f) - eps - gam;                                // pi_t+1 = gamma + phi_t (1 - eps - gamma)
        // ---- this lane's periods t = sub, sub + G: visits (log-likelihood and gradient sums at once), scaled likelihoods -> LDS ----
        bl_f2 gt[2][KO + 1], own_m[2], own_c[2], own_rho[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int t = sub + i * G;
#pragma unroll
            for (int k = 0; k <= KO; k++) gt[i][k] = bl2(0.0f);
            own_m[i] = bl2(0.0f); own_c[i] = bl2(1.0f); own_rho[i] = bl2(0.0f);
            if (t < T) {
                const float2 *pp = rec + XQ + t * pb;
                const float2 a_ = pp[J * (KO + 1)], kb_ = pp[J * (KO + 1) + 1];
                bl_f2 a = bl_f2{a_.x, a_.y};                   // ka: cancels the log sigma(0) of the masked visits
                const bl_f2 kb = bl_f2{kb_.x, kb_.y};
#pragma unroll 2
                for (int j = 0; j < J; j++) {
                    bl_f2 w[KO + 1];
#pragma unroll
                    for (int k = 0; k <= KO; k++) {
                        const float2 v = pp[j * (KO + 1) + k];
                        w[k] = bl_f2{v.x, v.y};
                    }
                    bl_visit2<KO>(w, alpha, a, gt[i]);
                }
                const bl_f2 mm = __builtin_elementwise_max(a, kb);
                const bl_f2 E1 = bl_exp2_2((a - mm) * bl2(BL_LOG2E)), E0 = bl_exp2_2((kb - mm) * bl2(BL_LOG2E));
                ecol[(2 * t) * nslots] = make_float2(E1.x, E1.y);
                ecol[(2 * t + 1) * nslots] = make_float2(E0.x, E0.y);
                own_m[i] = mm;
            }
        }
        if (G > 1) bl_wave_lds_fence();
        // ---- the scaled forward recursion (every lane of the group, all periods) ----
        // One step's dependent chain is  c = E0 + pi (E1 - E0)  ->  1 / c  ->  phi = (pi E1) / c  ->  pi' = gamma + phi stay ; the next
        // step's scaled likelihoods are loaded a step ahead (the compiler will not move an LDS load across the stores of the loop, so the
        // order is written out here: without it every step paid an LDS round trip on top of its chain).
        bl_f2 pi = psi, phi = bl2(0.0f);
        float2 e1n = ecol[0], e0n = ecol[nslots];
        for (int t = 0; t < T; t++) {
            const bl_f2 E1 = bl_f2{e1n.x, e1n.y}, E0 = bl_f2{e0n.x, e0n.y};
            const int tn = min(t + 1, T - 1);
            e1n = ecol[(2 * tn) * nslots]; e0n = ecol[(2 * tn + 1) * nslots];
            const bl_f2 a1 = pi * E1;
            const bl_f2 c = __builtin_elementwise_max(bl_fma2(pi, E1 - E0, E0), bl2(1e-37f)); // P(y_t | y_1..t-1) / e^m_t  (> 0 unless pi left (0, 1) in f32)
            phi = a1 * bl_rcp_2(c);                            // P(z_t = 1 | y_1..t)
            const bl_f2 pin = bl_fma2(phi, stay, gam);         // P(z_t+1 = 1 | y_1..t)
            col[(2 * t) * CT] = make_float2(phi.x, phi.y);
            col[(2 * t + 1) * CT] = make_float2(pin.x, pin.y);
            if (t == sub) own_c[0] = c;
            if (t == sub + G) own_c[1] = c;
            pi = pin;
        }
        // ---- backward: the smoothed marginals rho_t and the transitions' gradient sums ----
        // With r1 = 1 / pi_t+1, r0 = 1 / (1 - pi_t+1) (0 where that state cannot be reached) the pairwise marginals are
        //   xi(1,1) = rho r1 f (1 - eps),  xi(0,1) = rho r1 (1 - f) gamma,  xi(1,0) = (1 - rho) r0 f eps,  xi(0,0) = (1 - rho) r0 (1 - f)(1 - gamma),
        // so  rho_t = xi(1,1) + xi(1,0) = A0 + rho (A1 - A0)  with A1 = r1 f (1 - eps), A0 = r0 f eps -- ONE fma on the dependent chain --
        // and with  Dl = rho r1 - (1 - rho) r0 :  d/d eta_gamma += (1 - f) gamma (1 - gamma) Dl,  d/d eta_eps -= f eps (1 - eps) Dl
        // (the first form's xi(0,1)(1 - gamma) - xi(0,0) gamma and xi(1,0)(1 - eps) - xi(1,1) eps, factored).  Everything but the fma hangs
        // off (f, pi_t+1), which are loaded a step ahead.
        bl_f2 rho = phi, d_gam = bl2(0.0f), d_eps = bl2(0.0f);
        if (T - 1 == sub) own_rho[0] = rho;
        if (T - 1 == sub + G) own_rho[1] = rho;
        const bl_f2 g1g = gam * (bl2(1.0f) - gam), e1e = eps * (bl2(1.0f) - eps);
        {
            // (volatile LDS-address-space loads: the loop has no store, so the compiler is otherwise free to sink the "a step ahead" loads
            // to right in front of their use -- it did: `ds_read2st64_b64 ... s_waitcnt lgkmcnt(0)` at the head of every step)
            typedef __attribute__((address_space(3))) const volatile bl_f2 BlLdsV2;
            const int t0 = max(T - 2, 0);
            bl_f2 fn = *(BlLdsV2 *)&col[(2 * t0) * CT], pn = *(BlLdsV2 *)&col[(2 * t0 + 1) * CT];
            for (int t = T - 2; t >= 0; t--) {
                const bl_f2 f = fn, p1 = pn;
                const int tp = max(t - 1, 0);
                fn = *(BlLdsV2 *)&col[(2 * tp) * CT]; pn = *(BlLdsV2 *)&col[(2 * tp + 1) * CT];
                const bl_f2 q1 = bl_rcp_2(p1), q0 = bl_rcp_2(bl2(1.0f) - p1);
                const bl_f2 r1 = bl_f2{p1.x > 0.0f ? q1.x : 0.0f, p1.y > 0.0f ? q1.y : 0.0f};
                const bl_f2 r0 = bl_f2{p1.x < 1.0f ? q0.x : 0.0f, p1.y < 1.0f ? q0.y : 0.0f};
                const bl_f2 A1 = r1 * f * (bl2(1.0f) - eps), A0 = r0 * f * eps;
                const bl_f2 Dl = bl_fma2(rho, r1 + r0, -r0);
                d_gam = bl_fma2((bl2(1.0f) - f) * g1g, Dl, d_gam);
                d_eps = bl_fma2(f * e1e, -Dl, d_eps);
                rho = bl_fma2(rho, A1 - A0, A0);
                if (t == sub) own_rho[0] = rho;
                if (t == sub + G) own_rho[1] = rho;
            }
        }
        if (G > 1) bl_wave_lds_fence(); // (the next round's scaled likelihoods overwrite this round's)
        // ---- this lane's periods: log-likelihood terms m_t + log c_t, d/d alpha = rho_t g_t ----
        bl_f2 lown = bl2(0.0f);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float valid = sub + i * G < T ? 1.0f : 0.0f;
            lown += bl_fma2(bl_log2_2(own_c[i]), bl2(BL_LN2), own_m[i]) * bl2(valid);
            const bl_f2 rt = own_rho[i] * vm_own; // (a masked visit or the dummy site: its w = 0 anyway)
#pragma unroll
            for (int k = 0; k <= KO; k++) ga2[k] = bl_fma2(rt, gt[i][k], ga2[k]);
        }
        ll2 = bl_fma2(lown, vm_own, ll2);
        const bl_f2 dv[3] = {(rho - psi) * vm_site, d_gam * vm_site, d_eps * vm_site};
#pragma unroll
        for (int b = 0; b < 3; b++) {
            gb2[b][0] += dv[b];
#pragma unroll
            for (int k = 0; k < KS; k++) gb2[b][k + 1] = bl_fma2(dv[b], x[k], gb2[b][k + 1]);
        }
    }
    ll += ll2.x + ll2.y;
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int k = 0; k <= KS; k++) gb[b][k] += gb2[b][k].x + gb2[b][k].y;
#pragma unroll
    for (int k = 0; k <= KO; k++) ga[k] += ga2[k].x + ga2[k].y;
}

// Round 5: ONE period per lane (T == G in {2, 4, 8}: BASELINE.json configs[4], 8 seasons on 8 lanes) and no recursion at all -- the
// forward-backward algorithm as two SCANS over the lanes of a group.  With the scaled likelihoods L_t = (E0_t, E1_t) and the transition
// matrix A = [[1 - gamma, gamma], [eps, 1 - eps]] every period is a 2 x 2 matrix, X_0 = diag(L_0), X_t = A diag(L_t):
//     alpha_t = pi X_0 X_1 ... X_t            (unnormalised forward vector;  pi = (1 - psi, psi))
//     beta_t  = X_t+1 ... X_T-1 (1, 1)'       (backward vector)
// so the INCLUSIVE PREFIX products P_t = X_0 ... X_t and SUFFIX products S_t = X_t ... X_T-1 are all a lane needs, and both are
// log2(G) Hillis-Steele steps of a 2 x 2 product with the lane 1, 2, 4 places to the left / right (DPP row_shr / row_shl; lanes
// without a partner multiply by the identity).  The two recursions of the other forms are 2 T - 1 dependent steps (a reciprocal or
// two on each) that every lane of a group executes alike; here a lane does 6 dependent matrix products, no reciprocal but its own
// normaliser's, and nothing goes through LDS.
//   rho_t(1) = alpha_t(1) beta_t(1) / Z_t,   xi_t(a, b) = alpha_t(a) Q_t+1(a, b) / Z_t,   Q_u(a, b) = X_u(a, b) beta_u(b),
//   Z_t = sum_a alpha_t(a) sum_b Q_t+1(a, b)    -- every lane normalises by ITS Z_t, whatever power of two its products carry:
// the products are rescaled once, after the second step (entries of a product of 4 periods' matrices: an exact power of two, by
// the exponent of the largest entry), and only the prefix keeps count of it -- the log-likelihood is the last lane's
// log(alpha_T-1 . 1) + ln 2 x that count + sum_t m_t.  Every lane adds ITS period's share of the gradients (the transition t -> t + 1,
// its rho_t g_t, lane 0 the initial state's): the wave sums them anyway.  Same records; the scratch region behind them is not touched.
struct BlM22 { bl_f2 a, b, c, d; }; // [[a, b], [c, d]], both sites of the pair
__device__ __forceinline__ BlM22 bl_m22_mul(const BlM22 &L, const BlM22 &R)
{
    BlM22 r;
    r.a = bl_fma2(L.b, R.c, L.a * R.a); r.b = bl_fma2(L.b, R.d, L.a * R.b);
    r.c = bl_fma2(L.d, R.c, L.c * R.a); r.d = bl_fma2(L.d, R.d, L.c * R.b);
    return r;
}
template <int CTRL> __device__ __forceinline__ float bl_dpp_row(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// the neighbour's value (row_shr / row_shl by CTRL & 15 lanes) where this lane has a partner inside its group, else `id`
template <int CTRL> __device__ __forceinline__ bl_f2 bl_dpp_or(bl_f2 v, bool has, float id)
{
    const float x = bl_dpp_row<CTRL>(v.x), y = bl_dpp_row<CTRL>(v.y);
    return bl_f2{has ? x : id, has ? y : id};
}
template <int CTRL> __device__ __forceinline__ BlM22 bl_m22_dpp_or_identity(const BlM22 &m, bool has)
{
    BlM22 r;
    r.a = bl_dpp_or<CTRL>(m.a, has, 1.0f); r.b = bl_dpp_or<CTRL>(m.b, has, 0.0f);
    r.c = bl_dpp_or<CTRL>(m.c, has, 0.0f); r.d = bl_dpp_or<CTRL>(m.d, has, 1.0f);
    return r;
}
// m / 2^e with e = the exponent of m's largest entry (per site; exact); returns e as a float (0 for an all-zero matrix)
__device__ __forceinline__ bl_f2 bl_m22_rescale(BlM22 &m)
{
    const bl_f2 mx = __builtin_elementwise_max(__builtin_elementwise_max(m.a, m.b), __builtin_elementwise_max(m.c, m.d));
    const int ex = mx.x > 0.0f ? __builtin_amdgcn_frexp_expf(mx.x) : 0, ey = mx.y > 0.0f ? __builtin_amdgcn_frexp_expf(mx.y) : 0;
    auto sc = [&](bl_f2 v) { return bl_f2{__builtin_amdgcn_ldexpf(v.x, -ex), __builtin_amdgcn_ldexpf(v.y, -ey)}; };
    m.a = sc(m.a); m.b = sc(m.b); m.c = sc(m.c); m.d = sc(m.d);
    return bl_f2{(float)ex, (float)ey};
}

// GC, JC > 0: the lanes per pair (= periods) and the visits per period as compile-time facts (8 and 4: BASELINE.json configs[4])
template <int KS, int KO, int CT, int GC = 0, int JC = 0>
__device__ __forceinline__ void bl_eval_sites_dyn_scan(int ct, int pstride, int cnt, int T, int J_rt, int G_rt,
                                                       const float (&bpsi)[KS + 1], const float (&bgam)[KS + 1], const float (&beps)[KS + 1],
                                                       const float (&alpha)[KO + 1],
                                                       float &ll, float (&gb)[3][KS + 1], float (&ga)[KO + 1])
{
    constexpr int XQ = (KS + 3) & ~3;
    constexpr int UNR = JC > 0 ? JC : 2; // the visit loop's unrolling
    const int J = JC > 0 ? JC : J_rt, G = GC > 0 ? GC : G_rt;
    const int pb = bl_period_block(J, KO);
    const float *data = bl_lds_f(BL_OFF_DATA);
    const int sub = ct & (G - 1), slot = ct / G, nslots = CT / G; // lane `sub` of group `slot` owns period t = sub
    const int npairs = (cnt + 1) >> 1;
    const int rounds = (npairs + nslots - 1) / nslots;
    const bool is_first = sub == 0, is_last = sub == G - 1;
    bl_f2 ll2 = bl2(0.0f), gb2[3][KS + 1], ga2[KO + 1];
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int k = 0; k <= KS; k++) gb2[b][k] = bl2(0.0f);
#pragma unroll
    for (int k = 0; k <= KO; k++) ga2[k] = bl2(0.0f);
    for (int rd = 0; rd < rounds; rd++) {
        const int m_raw = rd * nslots + slot;
        const int m = min(m_raw, npairs - 1);
        const float live = m_raw < npairs ? 1.0f : 0.0f;
        const float2 *rec = reinterpret_cast<const float2 *>(data + (size_t)m * pstride);
        const float second = (2 * m + 1 < cnt) ? live : 0.0f; // odd slice: the last pair's second site is a dummy
        const bl_f2 vm = bl_f2{live, second};
        bl_f2 x[KS > 0 ? KS : 1];
        bl_f2 e_psi = bl2(bpsi[0]), e_gam = bl2(bgam[0]), e_eps = bl2(beps[0]);
#pragma unroll
        for (int k = 0; k < KS; k++) {
            const float2 v = rec[k];
            x[k] = bl_f2{v.x, v.y};
            e_psi = bl_fma2(x[k], bl2(bpsi[k + 1]), e_psi);
            e_gam = bl_fma2(x[k], bl2(bgam[k + 1]), e_gam);
            e_eps = bl_fma2(x[k], bl2(beps[k + 1]), e_eps);
        }
        const bl_f2 gam = bl_sigmoid2(e_gam), eps = bl_sigmoid2(e_eps), psi = bl_sigmoid2(e_psi);
        // ---- this lane's period: visits (log-likelihood and gradient sums at once), scaled likelihoods ----
        bl_f2 gt[KO + 1];
#pragma unroll
        for (int k = 0; k <= KO; k++) gt[k] = bl2(0.0f);
        const float2 *pp = rec + XQ + sub * pb;
        const float2 a_ = pp[J * (KO + 1)], kb_ = pp[J * (KO + 1) + 1];
        bl_f2 a = bl_f2{a_.x, a_.y};                   // ka: cancels the log sigma(0) of the masked visits
        const bl_f2 kb = bl_f2{kb_.x, kb_.y};
#pragma unroll UNR
        for (int j = 0; j < J; j++) {
            bl_f2 w[KO + 1];
#pragma unroll
            for (int k = 0; k <= KO; k++) {
                const float2 v = pp[j * (KO + 1) + k];
                w[k] = bl_f2{v.x, v.y};
            }
            bl_visit2<KO>(w, alpha, a, gt);
        }
        const bl_f2 mm = __builtin_elementwise_max(a, kb);
        const bl_f2 E1 = bl_exp2_2((a - mm) * bl2(BL_LOG2E)), E0 = bl_exp2_2((kb - mm) * bl2(BL_LOG2E));
        // ---- this period's matrix: X_0 = diag(L_0), X_t = A diag(L_t) ----
        BlM22 X;
        X.a = is_first ? E0 : (bl2(1.0f) - gam) * E0; X.b = is_first ? bl2(0.0f) : gam * E1;
        X.c = is_first ? bl2(0.0f) : eps * E0;        X.d = is_first ? E1 : (bl2(1.0f) - eps) * E1;
        // ---- inclusive prefix products P_t = X_0 ... X_t and suffix products S_t = X_t ... X_T-1 over the group's lanes ----
        BlM22 P = X, S = X;
        bl_f2 eP = bl2(0.0f);                          // P's true value is P x 2^eP
#define BL_DYN_SCAN_STEP(s)                                                                          \
        if (G > (s)) {                                                                               \
            const bool hl = sub >= (s), hr = sub + (s) < G;                                          \
            const BlM22 Lp = bl_m22_dpp_or_identity<0x110 + (s)>(P, hl);                             \
            eP += bl_dpp_or<0x110 + (s)>(eP, hl, 0.0f);                                              \
            P = bl_m22_mul(Lp, P);                                                                   \
            const BlM22 Rs = bl_m22_dpp_or_identity<0x100 + (s)>(S, hr);                             \
            S = bl_m22_mul(S, Rs);                                                                   \
        }
        BL_DYN_SCAN_STEP(1)
        BL_DYN_SCAN_STEP(2)
        if (G > 4) { eP += bl_m22_rescale(P); (void)bl_m22_rescale(S); }
        BL_DYN_SCAN_STEP(4)
#undef BL_DYN_SCAN_STEP
        // ---- beta_t = (row sums of S_t+1), Q_t+1 = X_t+1 (.) beta_t+1 from the lane to the right; alpha_t = pi P_t ----
        const bl_f2 rs0 = S.a + S.b, rs1 = S.c + S.d;
        const bl_f2 be0 = bl_dpp_or<0x101>(rs0, !is_last, 1.0f), be1 = bl_dpp_or<0x101>(rs1, !is_last, 1.0f); // this lane's beta_t
        BlM22 Qo; // what this lane hands to the left: X_t (.) beta_t, column-wise
        Qo.a = X.a * be0; Qo.b = X.b * be1; Qo.c = X.c * be0; Qo.d = X.d * be1;
        const BlM22 Q = bl_m22_dpp_or_identity<0x101>(Qo, !is_last); // Q_t+1 (the last lane: the identity -- beta = 1, no transition)
        const bl_f2 al0 = bl_fma2(psi, P.c, (bl2(1.0f) - psi) * P.a), al1 = bl_fma2(psi, P.d, (bl2(1.0f) - psi) * P.b);
        const bl_f2 Z = __builtin_elementwise_max(bl_fma2(al1, Q.c + Q.d, al0 * (Q.a + Q.b)), bl2(1e-37f));
        const bl_f2 rz = bl_rcp_2(Z);
        const bl_f2 rho = al1 * (Q.c + Q.d) * rz;      // P(z_t = 1 | y)
        const bl_f2 t0 = al0 * rz, t1 = al1 * rz;
        const bl_f2 nx = is_last ? bl2(0.0f) : vm;     // (the last period starts no transition)
        const bl_f2 d_gam = (t0 * Q.b * (bl2(1.0f) - gam) - t0 * Q.a * gam) * nx;   // xi(0,1)(1 - gamma) - xi(0,0) gamma
        const bl_f2 d_eps = (t1 * Q.c * (bl2(1.0f) - eps) - t1 * Q.d * eps) * nx;   // xi(1,0)(1 - eps) - xi(1,1) eps
        const bl_f2 d_psi = is_first ? (rho - psi) * vm : bl2(0.0f);
        // ---- log-likelihood: every lane its period's scale m_t, the last lane log(alpha_T-1 . 1) and the prefix's power of two ----
        bl_f2 lown = mm;
        if (is_last) lown += (bl_log2_2(__builtin_elementwise_max(al0 + al1, bl2(1e-37f))) + eP) * bl2(BL_LN2);
        ll2 = bl_fma2(lown, vm, ll2);
        const bl_f2 rt = rho * vm;
#pragma unroll
        for (int k = 0; k <= KO; k++) ga2[k] = bl_fma2(rt, gt[k], ga2[k]);
        const bl_f2 dv[3] = {d_psi, d_gam, d_eps};
#pragma unroll
        for (int b = 0; b < 3; b++) {
            gb2[b][0] += dv[b];
#pragma unroll
            for (int k = 0; k < KS; k++) gb2[b][k + 1] = bl_fma2(dv[b], x[k], gb2[b][k + 1]);
        }
    }
    ll += ll2.x + ll2.y;
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int k = 0; k <= KS; k++) gb[b][k] += gb2[b][k].x + gb2[b][k].y;
#pragma unroll
    for (int k = 0; k <= KO; k++) ga[k] += ga2[k].x + ga2[k].y;
}

// Wave reduction of (gb[3], ga, ll) -> this wave's row of the LDS partial table in the MODEL 8 layout (one interleaved DPP butterfly)
template <int KS, int KO>
__device__ __forceinline__ void bl_wave_partials_dyn(int cwave, float ll, const float (&gb)[3][KS + 1], const float (&ga)[KO + 1])
{
    constexpr int NV = BL_DYN_LL(KS, KO) + 1;
    static_assert(NV <= BL_PART_STRIDE, "MODEL 8: a wave's partial sums must fit its row of the table");
    const int lane = threadIdx.x & 63;
    float v[NV];
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int k = 0; k <= KS; k++) v[b * (KS + 1) + k] = gb[b][k];
#pragma unroll
    for (int k = 0; k <= KO; k++) v[BL_DYN_OA(KS) + k] = ga[k];
    v[BL_DYN_LL(KS, KO)] = ll;
    if constexpr (NV > 16 && NV <= 20) {
        // (round 4) the first sixteen as a reduce-scatter (occu_device.hpp: 35 instructions instead of the butterfly's 96), the rest one by one
        float head[16];
#pragma unroll
        for (int k = 0; k < 16; k++) head[k] = v[k];
        const float y = bl_wave_reduce_scatter<16>(head);
        const int g = lane >> 2;
        const int k = (int)(BlScatter<16>::slots() >> (4 * g)) & 15;
        float *part = bl_lds_f(BL_OFF_PART) + cwave * BL_PART_STRIDE;
        if ((lane & 3) == 0 && ((BlScatter<16>::stores() >> g) & 1u)) part[k] = y;
#pragma unroll
        for (int i = 16; i < NV; i++) {
            const float t = bl_wave_sum(v[i]);
            if (lane == 0) part[i] = t;
        }
    } else {
    bl_wave_sum_vec_l63<NV>(v);
    if (lane == 63) {
        float *part = bl_lds_f(BL_OFF_PART) + cwave * BL_PART_STRIDE;
#pragma unroll
        for (int k = 0; k < NV; k++) part[k] = v[k];
    }
    }
}
