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

// bytes of LDS behind the staged records: two float2 per period and compute lane (the lane's forward / smoothed columns), and one
// more per period and lane GROUP for the periods' visit sums that the lanes of a group hand each other (host: choose_geometry)
__host__ __device__ inline int bl_dyn_scratch_bytes(int T, int cw) { return T * 3 * 8 * cw * 64; }
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
    bl_wave_sum_vec_l63<NV>(v);
    if (lane == 63) {
        float *part = bl_lds_f(BL_OFF_PART) + cwave * BL_PART_STRIDE;
#pragma unroll
        for (int k = 0; k < NV; k++) part[k] = v[k];
    }
}
