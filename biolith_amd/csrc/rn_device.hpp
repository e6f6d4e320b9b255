// rn_device.hpp -- Royle-Nichols (occu_rn) site evaluation for gfx950, WORK-PROPORTIONAL form.
//
// biolith/models/occu_rn.py:179-222 + utils/distributions.py:31-40 with N summed out:
//   lambda = exp(eta);  pi_n = Poisson(lambda)(n) renormalised over n <= K   (K = max_abundance)
//   P(y=1 | n) = 1 - q^n = r * b_n ,  q = 1 - r ,  b_n = 1 + q + ... + q^(n-1)  (b_n = b_(n-1) q + 1: no cancellation);
//   non-detections contribute n log q: rank-1 in n.
//   l = logsumexp_n [ n (eta + sum_nondet log q) - lgamma(n+1) + sum_det (log r + log b_n) ] - log Z
//
// Rounds 1-2 gave every site one lane that ran the sum over n up to its WAVE's largest cutoff (20-35 at BASELINE.json's
// config 4, where a site needs n <= 7 on average) with a 128-entry per-lane table, one wave per SIMD.  Here the sum over n
// of a (site, period) is cut into ITEMS of BL_RN_CH = 8 consecutive n and a wave evaluates its sites in two steps:
//
//   P1  one lane per site: the visits (log r, log q, the rank-1 gradient pieces; each visit's q or log2 q is left in a spare
//       slot of the site's LDS record for the items), the truncated-Poisson normaliser (closed form e^lambda whenever the
//       tail beyond K is below 3e-11, else a loop), and the site's item count from the two concave bounds of its terms;
//       a record {a, clr, term0, thr, cnon, n*} per site goes to a wave-private LDS table.
//   P2  one lane per ITEM (site, chunk c: n = 8c+1 .. 8c+8): a wave prefix sum of the sites' item counts deals the items
//       to lanes, sites packed whole into rounds of <= 64 lanes.  The item's b_n / q^n recursions start from
//       b_(8c) = b_8 (1 + q^8 + ... + q^(8(c-1))), q^(8c) (products of positives: no cancellation, no transcendental);
//       the items of a site combine their maxima and sums through a wave-private LDS table; every item then adds its
//       share of the detections' gradient straight into its lane's accumulators (they are summed over the wave at
//       the end anyway), so no per-site reduction of gradients exists.  numpyro's floor of a non-detection is applied by
//       the items that hold a term it can matter for (a bound per term, no per-site flags).
//       The item for chunk 0 also carries the n = 0 term and hands (max, sum, sum n w) back to the site's lane.
//
// Round 6 (from per-wave and per-SIMD stamps, profiles/r06): a (site, period) without a detection takes NO item -- its sum over n is closed
// (e^(e^a)) whenever the truncation at K and numpyro's floor provably cannot matter; a workgroup stages its sites with a detection first
// and, when they fit CW - 1 waves, gives the others to its LAST compute wave -- the one that shares a SIMD with the first -- whose stream
// is then the visit pass alone; and the first non-detection to reach numpyro's floor is stated by every item without a branch (the loop
// over the visits remains for a second one).
//
// Work is then sum_sites ceil(cutoff / 8) items of 8 terms instead of 64 x (wave's largest cutoff) per wave, the per-lane
// table is 8 registers, and a workgroup runs 5 compute waves + the control wave (six waves on four SIMDs): a wave's instruction
// stream is the same however few of its lanes are busy, so the count is the one that fills the lanes (31 sites, 46 items per wave at
// BASELINE.json's config 4) while a SIMD still has a second wave to hide the exp / log / rcp and LDS latencies behind.
#pragma once

#ifndef BL_CWAVES_RN
#define BL_CWAVES_RN 5                 // compute waves per workgroup of the occu_rn kernels (measured at config 4, k = 32: 3 -> 15.5, 4 -> 12.9,
                                       // 5 -> 9.5, 6 -> 10.0, 7 -> 9.9, 11 -> 13.5, 15 -> 16.0 us per leapfrog)
#endif
#define BL_RN_CH 8                     // n-terms per item
// Items of a site whose chunk tests / item-map stores are made without a loop.  Same-box A/B of both at once (profiles/r06/o_time_rn_unroll.txt,
// us per leapfrog of the slowest chain): 3: 7.55 | 4: 7.46 | 5: 7.61 | 6: 7.58 -- a wave that holds a site of four or five items pays ~400
// cycles per loop round (o_stamps_rn_slow_and_typical_wave.txt) and is the slowest of its chain, but every further unconditional test
// costs all 128 main waves; tests 4 with stores 3 measured 7.42 and is what ships.
#ifndef BL_RN_TWO_PER_ROUND
#define BL_RN_TWO_PER_ROUND 1          // the rare loops (chunk count beyond the unrolled tests, item map's tail, the recursions' start values) take two steps a
                                       // round (0: A/B; 7.44 -> 7.38 us per leapfrog, profiles/r06/q_time_rn_rare_loops.txt)
#endif
#ifndef BL_RN_UNR_TESTS
#define BL_RN_UNR_TESTS 4
#endif
#ifndef BL_RN_UNR_STORES
#define BL_RN_UNR_STORES 3
#endif
#define BL_RN_LGT 144                  // floats of the shifted lgamma table: lgt[i] = lgamma(i + 2) = the entry of n = i + 1
#define BL_RN_WAVE_FLOATS 1344         // wave-private scratch: site records 64 x 12, site results 64 x 4, item map 64, combine 64 x 4
#define BL_RN_GA 10                    // visits whose b_n recursions run side by side in pass A1 (one log per group and n)
#define BL_RN_G0 5                     // visits evaluated side by side in pass A0
// The workgroup's ORDER of its sites (round 6): [sites with a detection, in their own order | sites without one], built once per launch
// (bl_rn_split_init), and how many of the first kind there are.  Why: five compute waves and the control wave are six waves on four SIMDs,
// so two compute waves share one -- the first and the last -- and the last one's evaluation used to end 2 400 cycles after everybody
// else's (profiles/r06/c_stamps_rn_waves.txt: 14.7 k against 11.7 k cycles, in every workgroup): the tick waited for it.  A site WITHOUT a
// detection needs no sum over n at all (closed form below), so those sites -- a third of BASELINE.json's config 4 -- are what the last
// wave takes: a short stream beside the first wave's full one, while the sites with detections are shared by the other CW - 1 waves.
#define BL_RN_ORD_MAX 508              // sites per workgroup up to which the order is kept (beyond: the plain equal shares)
#define BL_RN_HDR (4 + BL_RN_ORD_MAX)  // ints behind the lgamma table: [0] sites with a detection when the split is on, else -1; [4 ...] the order
// bytes of LDS behind the staged records that the occu_rn kernels need (host: choose_geometry)
__host__ __device__ inline int bl_rn_scratch_bytes(int cw) { return 4 * (BL_RN_LGT + BL_RN_HDR + cw * BL_RN_WAVE_FLOATS); }

#ifdef BL_STAMPS
static __device__ long long bl_rn_dbg[16];
#define BL_RN_T(i) { const long long now_ = (long long)clock64(); if (st_on) bl_rn_dbg[i] += now_ - st_prev; st_prev = now_; }
#else
#define BL_RN_T(i)
#endif

// lgt[i] = lgamma(i + 2) for n = i + 1 <= K, a huge value beyond (that term's weight underflows to exactly 0: the model's
// bound n <= max_abundance).  Every thread of the workgroup calls this once, before the first evaluation and a barrier.
__device__ __forceinline__ void bl_rn_fill_lgamma(int lds_off, int K, int nthreads)
{
    float *lgt = bl_lds_f(lds_off);
    for (int i = threadIdx.x; i < BL_RN_LGT; i += nthreads) lgt[i] = (i + 1 <= K && i + 1 < 128) ? BL_LGAMMA1P[i + 1] : 3.0e38f;
}

// wave-uniform maximum of a non-negative per-lane integer (inactive lanes count as 0)
__device__ __forceinline__ int bl_wave_max_u(int x)
{
#define BL_MAXSTEP(ctrl, rm) x = max(x, __builtin_amdgcn_update_dpp(0, x, ctrl, rm, 0xF, false));
    BL_MAXSTEP(0xB1, 0xF) BL_MAXSTEP(0x4E, 0xF) BL_MAXSTEP(0x141, 0xF) BL_MAXSTEP(0x140, 0xF)
    BL_MAXSTEP(0x142, 0xA) BL_MAXSTEP(0x143, 0xC)
#undef BL_MAXSTEP
    return __builtin_amdgcn_readlane(x, 63);
}
// inclusive prefix sum over the 64 lanes: Hillis-Steele inside the rows of 16 (row_shr 1, 2, 4, 8), then the rows' totals
__device__ __forceinline__ int bl_wave_iscan(int x)
{
#define BL_SCANSTEP(ctrl, rm) x += __builtin_amdgcn_update_dpp(0, x, ctrl, rm, 0xF, false);
    BL_SCANSTEP(0x111, 0xF) BL_SCANSTEP(0x112, 0xF) BL_SCANSTEP(0x114, 0xF) BL_SCANSTEP(0x118, 0xF)
    BL_SCANSTEP(0x142, 0xA) BL_SCANSTEP(0x143, 0xC)
#undef BL_SCANSTEP
    return x;
}
// the compiler may not move LDS accesses across this point; a wave's LDS operations complete in order
__device__ __forceinline__ void bl_wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Once per launch, by the workgroup's first wave, IN FRONT of the staging of the records (a barrier between the two): the order of the
// sites (see BL_RN_HDR; bl_stage_records places site ord[i] of the slice at position i, so the evaluator never looks at the order again)
// and whether the split is on -- it is when there are sites of both kinds, the sites with a detection fit CW - 1 waves at two lanes
// per site (<= 32 each: the two-lane visit pass), and the order fits its table.  Read from the HBM rows: row KS + v (KO + 2) holds
// visit v's sign c (> 0: a detection) of every site.
template <int KS, int KO, int CW>
__device__ __forceinline__ void bl_rn_split_init(const float *__restrict__ rows, int n_stride, int s0, int cnt, int T, int J, int rn_off)
{
    if (threadIdx.x >= 64) return;
    constexpr int VW = KO + 2;
    const int lane = threadIdx.x, V = T * J;
    int *hdr = reinterpret_cast<int *>(bl_lds_f(rn_off) + BL_RN_LGT);
    int *ord = hdr + 4;
    if (cnt > BL_RN_ORD_MAX || CW < 3) {
        if (lane == 0) hdr[0] = -1;
        return;
    }
    auto detected = [&](int i) -> bool {
        bool det = false;
        for (int v = 0; v < V; v++) det = det || rows[(size_t)(KS + v * VW) * n_stride + s0 + i] > 0.0f;
        return det;
    };
    int npos = 0;
    for (int base = 0; base < cnt; base += 64) {
        const int i = base + lane;
        npos += (int)__popcll(__ballot(i < cnt && detected(i)));
    }
    int p1 = 0, p0 = npos;
    for (int base = 0; base < cnt; base += 64) {
        const int i = base + lane;
        const bool in = i < cnt, det = in && detected(i);
        const unsigned long long m1 = __ballot(det), m0 = __ballot(in && !det);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (det) ord[p1 + (int)__popcll(m1 & below)] = i;
        else if (in) ord[p0 + (int)__popcll(m0 & below)] = i;
        p1 += (int)__popcll(m1);
        p0 += (int)__popcll(m0);
    }
    const bool split = npos > 0 && npos < cnt && (npos + CW - 2) / (CW - 1) <= 32;
    if (lane == 0) hdr[0] = split ? npos : -1;
}
// what bl_rn_split_init decided (behind the barrier that follows it): the sites with a detection when the split is on, else -1
__device__ __forceinline__ int bl_rn_npos(int rn_off)
{
    return __builtin_amdgcn_readfirstlane(reinterpret_cast<const int *>(bl_lds_f(rn_off) + BL_RN_LGT)[0]);
}
// the order to stage the records in, or NULL when none was built (bl_stage_records)
template <int CW> __device__ __forceinline__ const int *bl_rn_order(int rn_off, int cnt)
{
    return (cnt > BL_RN_ORD_MAX || CW < 3) ? nullptr : reinterpret_cast<const int *>(bl_lds_f(rn_off) + BL_RN_LGT) + 4;
}

// lower bound of max_n (n a - lgamma(n+1)) over 1 <= n <= K: evaluate at the Poisson mode with
// lgamma(n+1) <= (n + 1/2) ln n - n + 1  (n >= 1)
__device__ __forceinline__ float bl_rn_mode_lb(float a, float K)
{
    const float n1 = fminf(fmaxf(floorf(bl_exp_f(fminf(a, 80.0f))), 1.0f), K);
    return fmaf(n1, a, -(fmaf(n1 + 0.5f, BL_LN2 * __builtin_amdgcn_logf(n1), 1.0f - n1)));
}

// Accumulates over this WAVE's share of the workgroup's sites (cwave of CW compute waves, contiguous shares):
//   ll += sum_t l_it ,  gb[k] += d ll / d beta_k ,  ga[k] += d ll / d alpha_k        (per lane; the caller sums the wave)
// LDS records: the plain model's, one float wider per visit (bl_layout_ko<1>: [c, c w_1 .. c w_KO, v]).  The last slot is DYNAMIC:
// at every evaluation the site's lane leaves there what the item lanes need of the visit -- q_j > 0 (a detection), log2 q_j < 0
// (a non-detection) or 0 (masked) -- so that the items read one float per visit instead of repeating the dot product and its exp / rcp.
// J10: at most ten visits per period AND one period as compile-time facts (the sampler's instantiation for that case -- config 4 --
// carries the one-group paths alone and no loop over periods)
// n_pos: the workgroup's sites with a detection when the waves' shares are split (bl_rn_npos, read ONCE per launch by the caller: read
// here, at every evaluation, it was an LDS round trip in front of everything else), -1 when they are not.
template <int KS, int KO, int CW, bool J10 = false>
__device__ __forceinline__ void bl_eval_sites_rn(int cwave, int pstride, int cnt, int T, int J, int K, int rn_off, int n_pos,
                                                 const float (&beta)[KS + 1], const float (&alpha)[KO + 1],
                                                 float &ll, float (&gb)[KS + 1], float (&ga)[KO + 1])
{
    constexpr int XQ = (KS + 3) & ~3;
    constexpr int VW = KO + 2;                 // floats per visit in the record
    const int lane = threadIdx.x & 63;
    const int pb = bl_period_block(J, KO + 1);
    float *data = bl_lds_f(BL_OFF_DATA);
    const float *lgt = bl_lds_f(rn_off);
    float *wsc = bl_lds_f(rn_off) + BL_RN_LGT + BL_RN_HDR + cwave * BL_RN_WAVE_FLOATS;
    constexpr int SR = 12;                                          // floats of a site's record
    float *srec = wsc;                                              // [64][12] site lane -> item lanes (8), parked for the site lane itself (2)
    float *sres = wsc + 768;                                        // [64][4]  chunk-0 item lane -> site lane
    unsigned *imap = reinterpret_cast<unsigned *>(wsc + 1024);      // [64]     item -> site lane | chunk << 8 | first lane << 16 | chunks << 24
    float *comb = wsc + 1088;                                       // [64][4]  items of one site: max, sum, sum n w
    const float LOG_TINY = -87.33654475f;
    const float FL = -15.942385f; // log(finfo(float32).eps)
    const float Kf = (float)K;
    const int cK = (K - 1) / BL_RN_CH;         // last chunk that holds an n <= K
#ifdef BL_STAMPS
#ifndef BL_RN_STAMP_WG
#define BL_RN_STAMP_WG 0     // the workgroup (of chain 0) and the compute wave whose stages are stamped (tools/stamps_rn.py)
#define BL_RN_STAMP_WAVE 0
#endif
    const bool st_on = blockIdx.x == 8 * BL_RN_STAMP_WG && threadIdx.x == 64 * (BL_RN_STAMP_WAVE + 1); // (chain 0, member BL_RN_STAMP_WG: the XCD-aware block mapping of nuts_kernel.hpp)
    long long st_prev = (long long)clock64();
#endif
    // this wave's share of the workgroup's sites (staged in the order of bl_rn_split_init): the sites with a detection in equal shares
    // over the first CW - 1 waves and the others to the last wave, or -- no split -- equal shares of all for everybody
    int w0, w1;
    if (n_pos >= 0) {
        const int spw = (n_pos + CW - 2) / (CW - 1);
        w0 = cwave < CW - 1 ? cwave * spw : n_pos;
        w1 = cwave < CW - 1 ? min(n_pos, w0 + spw) : cnt;
    } else {
        const int spw = (cnt + CW - 1) / CW;
        w0 = cwave * spw;
        w1 = min(cnt, w0 + spw);
    }
    for (int r0 = w0; r0 < w1; r0 += 64) {
        const int ns = min(64, w1 - r0);           // sites of this round (wave-uniform), one per lane
        // A round of <= 32 sites gives every site TWO lanes for its visits (l and l + 32 take half of them each and fold their
        // sums): the visits are the longest stretch of P1 and half of the lanes would idle.  Everything else is the site lane's.
        const bool two = ns <= 32;                 // wave-uniform
        const int sl0 = two ? (lane & 31) : lane, half = two ? (lane >> 5) : 0;
        const bool has = lane < ns;                // this lane is a site's lane
        const int i = r0 + min(sl0, ns - 1);       // lanes beyond the round re-evaluate its last site and are masked out
        const float live = has ? 1.0f : 0.0f;
        float *rec = data + (size_t)(i >> 1) * pstride + (i & 1); // element e of this site: rec[2 e]
        float eta = beta[0];
#pragma unroll
        for (int k = 0; k < KS; k++) eta = fmaf(rec[2 * k], beta[k + 1], eta); // (x is read again for the gradient: not kept across P2)
        // ---- truncated-Poisson prior, p_n = n eta - lgamma(n+1), n <= K: log Z and E[n] ----
        // Z = e^lambda (1 - tail), tail = P(Poisson(lambda) > K) <= 2 e^(p_(K+1) - lambda) while lambda <= (K+2)/2: below 3e-11
        // the closed forms log Z = lambda, E[n] = lambda are exact in float32.  Otherwise (lambda near or beyond K) the sums.
        const float lam = bl_exp_f(fminf(eta, 80.0f));
        const float pK = fmaf(Kf, eta, -lgt[K - 1]);
        const float pK1 = pK + eta - BL_LN2 * __builtin_amdgcn_logf(Kf + 1.0f);
        const bool closed = (lam <= 0.5f * (Kf + 2.0f)) && (pK1 - lam <= -25.0f);
        float log_z = lam, en_prior = lam;
        if (__any(has && !closed)) {
            // shift: an upper bound of max_n p_n tight to a few nats -- lambda while lambda <= K (Stirling: p_n <= lambda - 0.9
            // for n >= 1, p_0 = 0), else p_K (p_n still rising at K)
            const float mzs = lam <= Kf ? lam : pK;
            const float mz_lb = fmaxf(bl_rn_mode_lb(eta, Kf), 0.0f);
            // no term with p_n < mz_lb - 20 matters: for n > lambda, p_n <= lambda - 0.9 - (n - lambda)^2 / (n + lambda) (Stirling, and
            // ln x >= 2 (x - 1) / (x + 1) for x >= 1), below that once n - lambda > (Tq + sqrt(Tq^2 + 8 lambda Tq)) / 2
            const float Tq = lam - 0.9f - (mz_lb - 20.0f);
            const float dmax = 0.5f * (Tq + __builtin_amdgcn_sqrtf(fmaf(Tq, Tq, 8.0f * lam * Tq)));
            const int Klp = min(K, bl_wave_max_u((has && !closed) ? (int)fminf(lam + dmax + 2.0f, 1.0e6f) : 0));
            float sz = bl_exp_f(-mzs), b1 = 0.0f; // n = 0
            for (int n = 1; n <= Klp; n++) {
                const float e = bl_exp_f(fmaf((float)n, eta, -lgt[n - 1]) - mzs);
                sz += e;
                b1 = fmaf((float)n, e, b1);
            }
            if (!closed) {
                log_z = mzs + BL_LN2 * __builtin_amdgcn_logf(sz);
                en_prior = b1 * __builtin_amdgcn_rcpf(sz);
            }
        }
        float ll_s = 0.0f, deta = 0.0f;
        BL_RN_T(0)
        const int Tn = J10 ? 1 : T; // (J10: also ONE period, a compile-time fact -- the loop folds away)
        for (int t = 0; t < Tn; t++) {
            float *pv = rec + 2 * (XQ + t * pb);
            float cnon = 0.0f, clr = 0.0f, ndet = 0.0f;
            float lqmin = 0.0f; // smallest log q over the non-detections, the visit it belongs to, and the second smallest (numpyro's floor: below)
            float lq2nd = 0.0f;
            int jmin = 0;
            float Rv[KO + 1];
#pragma unroll
            for (int k = 0; k <= KO; k++) Rv[k] = 0.0f;
            // ---- A0 (lane = site, or half a site): visits, no sum over n yet.  u = c nu;  log sigma(u) = log r (detection) or log q (non-detection) ----
            // BL_RN_G0 visits side by side and branch-free.  A visit past the lane's share re-reads the last one as masked.
            const int Jh = two ? (J + 1) >> 1 : J, jbeg = half * Jh;
            for (int j0 = 0; j0 < Jh; j0 += BL_RN_G0) {
#pragma unroll
                for (int g = 0; g < BL_RN_G0; g++) {
                    const bool valid = (j0 + g < Jh) && (jbeg + j0 + g < J) && sl0 < ns;
                    const int j = min(jbeg + j0 + g, J - 1);
                    float w[KO + 1];
#pragma unroll
                    for (int k = 0; k <= KO; k++) w[k] = pv[2 * (j * VW + k)];
                    const float c = valid ? w[0] : 0.0f;
                    float u = w[0] * alpha[0];
#pragma unroll
                    for (int k = 1; k <= KO; k++) u = fmaf(w[k], alpha[k], u);
                    const float e = __builtin_amdgcn_exp2f(-fabsf(u) * BL_LOG2E), op = 1.0f + e;
                    const float logsig = fminf(u, 0.0f) - BL_LN2 * __builtin_amdgcn_logf(op);
                    const float sneg = (u > 0.0f ? e : 1.0f) * __builtin_amdgcn_rcpf(op); // sigma(-u): q of a detection, r of a non-detection
                    const float ld = c > 0.0f ? logsig : 0.0f, ln = c < 0.0f ? logsig : 0.0f;
                    const float sm = c < 0.0f ? sneg : 0.0f;
                    clr += ld;
                    ndet += c > 0.0f ? 1.0f : 0.0f;
                    cnon += ln;
                    {   // (a visit that ties with the smallest counts as the second smallest)
                        const bool lower = ln < lqmin;
                        lq2nd = lower ? lqmin : fminf(lq2nd, ln);
                        jmin = lower ? j : jmin;
                        lqmin = fminf(lqmin, ln);
                    }
                    // d/dnu of n log q is -n r: rank-1; dnu * (1, w) = r E[n] * (c, c w)
#pragma unroll
                    for (int k = 0; k <= KO; k++) Rv[k] = fmaf(sm, w[k], Rv[k]);
                    // the visit's dynamic slot: q (detection), log2 q (non-detection), 0 (masked)
                    if (valid) pv[2 * (j * VW + KO + 1)] = c > 0.0f ? sneg : ln * BL_LOG2E;
                }
            }
            if (two) { // fold the two halves of every site (v_permlane32_swap: x[l], x[l ^ 32] side by side in every lane)
                auto both = [](float v, bool take_min) -> float {
                    const unsigned b = __float_as_uint(v);
                    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
                    const float lo = __uint_as_float(r[0]), hi = __uint_as_float(r[1]);
                    return take_min ? fminf(lo, hi) : lo + hi;
                };
                cnon = both(cnon, false); clr = both(clr, false); ndet = both(ndet, false);
                {   // the smallest two of the halves' four, and the smallest one's visit
                    const auto rm = __builtin_amdgcn_permlane32_swap(__float_as_uint(lqmin), __float_as_uint(lqmin), false, false);
                    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(lq2nd), __float_as_uint(lq2nd), false, false);
                    const auto rj = __builtin_amdgcn_permlane32_swap((unsigned)jmin, (unsigned)jmin, false, false);
                    const bool up = lane >= 32;                       // r[0]: lane l's value, r[1]: lane l + 32's, in both lanes
                    const float om = __uint_as_float(up ? rm[0] : rm[1]), o2 = __uint_as_float(up ? r2[0] : r2[1]);
                    const int oj = (int)(up ? rj[0] : rj[1]);
                    const bool lower = om < lqmin;
                    lq2nd = lower ? fminf(lqmin, o2) : fminf(lq2nd, om);
                    jmin = lower ? oj : jmin;
                    lqmin = fminf(lqmin, om);
                }
#pragma unroll
                for (int k = 0; k <= KO; k++) Rv[k] = both(Rv[k], false);
            }
            BL_RN_T(1)
            const float a = eta + cnon;
            const float term0 = ndet * LOG_TINY; // n = 0: detections impossible -> numpyro's clamp tiny
            // numpyro floors a non-detection's log(1 - P) = n log q at log(eps_f32) = -15.94 (Bernoulli probabilities are clamped to
            // [tiny, 1 - eps]): the visit's term is max(n log q_j, FL).  The first visit to floor does so at n* = FL / lqmin; up to there
            // the non-detections' sum is n cnon (rank one in n: it rides in `a`), beyond it is non-increasing, hence <= n* cnon.
            // Every term_n is therefore bounded above by the Poisson part p_n = n eta - lgamma(n+1) plus that share (the detections'
            // log r + log b_n = log(1 - q^n) is <= 0):
            //     g(n) = max( n a - lgamma(n+1) ,  p_n + n* cnon ) ,     both branches concave in n,
            // and the best term is at least m_lb = max(term_0, term_1, term at the mode of n a - lgamma(n+1), with log b >= 0).
            // The site keeps n up to the last g(n) >= thr = m_lb - 20 (what is dropped is below 2e-9 of the sum, per term).
            const float ea = bl_exp_f(fminf(a, 80.0f));
            const float nstar = lqmin < 0.0f ? FL * __builtin_amdgcn_rcpf(lqmin) : 0.0f; // cnon = 0 when there is none
            const float shiftB = cnon * nstar;
            // ---- NO detection in this (site, period): every term is p_n + sum_j max(n log q_j, FL) and, the floor aside, the sum over n is
            // closed: sum_n e^(n a) / n! = e^(e^a), l = e^a - log Z, E[n | y] = e^a -- no items.  The truncation at K is covered by
            // `closed` (e^a <= lambda: its tail beyond K is the smaller one); the floor RAISES terms beyond n* only, by at most
            // sum_(n > n*) e^(p_n + n* cnon) <= e^(lambda + n* cnon) against a sum of at least 1 (the n = 0 term): dropped when that is
            // below e^-20.  (A site that fails either test takes its item like any other.) ----
            const bool zero_closed = has && ndet == 0.0f && closed && (lqmin == 0.0f || lam + shiftB <= -20.0f);
            float thr = 0.0f;
            int nch = (has && !zero_closed) ? 1 : 0;
            if (__any(nch > 0)) {
            const float n1 = fminf(fmaxf(floorf(ea), 1.0f), Kf);
            const float mode_lb = fmaf(n1, a, -(fmaf(n1 + 0.5f, BL_LN2 * __builtin_amdgcn_logf(n1), 1.0f - n1))); // (bl_rn_mode_lb)
            thr = fmaxf(fmaxf(term0, a + clr), mode_lb + clr) - 20.0f;
            // ---- the site's item count: chunk c (n = 8c+1 ..) is needed while some n >= 8c+1 has g(n) >= thr.  Per branch: its mode
            // is at or beyond 8c+1 (floor(e^a), whose value is >= thr; floor(lambda), with p_n <= lambda - 0.9 as the test) or, past the
            // mode where the branch falls, its value AT 8c+1 still reaches thr.  Monotone in c, so the chunks are counted until no
            // site of the wave needs another (two or three steps at the posterior). ----
            {
                const bool okB = lam - 0.9f + shiftB >= thr;
                auto need_chunk = [&](int c) -> bool {
                    const float tn = (float)(c * BL_RN_CH + 1), lg = lgt[c * BL_RN_CH];
                    return (ea >= tn) || (fmaf(tn, a, -lg) >= thr) || (okB && lam >= tn) || (fmaf(tn, eta, -lg) + shiftB >= thr);
                };
                // (the first BL_RN_UNR_TESTS - 1 further chunks are tested unconditionally -- their table reads and compares go out together --
                // the rest by the loop: a round of it is an LDS read, a wave-wide `any` and a branch, ~400 cycles beside the other wave)
                bool need = nch > 0;
#pragma unroll
                for (int c = 1; c < BL_RN_UNR_TESTS; c++) {
                    need = need && cK >= c && need_chunk(c);
                    nch += need ? 1 : 0;
                }
#if BL_RN_TWO_PER_ROUND
                for (int c = BL_RN_UNR_TESTS; c <= cK && __any(need); c += 2) { // (two chunks a round: their table reads go out together)
                    const bool n1 = need && need_chunk(c);
                    const bool n2 = n1 && c + 1 <= cK && need_chunk(c + 1);    // (the table reaches beyond the last chunk: huge values there)
                    nch += (n1 ? 1 : 0) + (n2 ? 1 : 0);
                    need = n2;
                }
#else
                for (int c = BL_RN_UNR_TESTS; c <= cK && __any(need); c++) {
                    need = need && need_chunk(c);
                    nch += need ? 1 : 0;
                }
#endif
            }
            }
            // record for the item lanes.  (A site in closed form has none: this lane leaves its result where the site's first item would --
            // log2 of e^(e^a), the sum relative to it, sum n w_n relative to it -- and nothing is kept across the item round for it.)
            // The FIRST visit to floor -- the non-detection with the smallest log q, visit jmin -- is stated exactly by every item without a
            // branch (its log2 q rides in the record: max(n log q, FL) = n log q + max(0, FL - n log q)); what the items still test for,
            // and treat by the loop over the visits when it matters, is the SECOND visit to floor, at n2* = FL / (second smallest log q):
            // beyond it the other non-detections' sum is at most n2* (cnon - lqmin).  (Round 6: with the first floor behind the test, a
            // wave held such a site in every other workgroup and its evaluation took 14.8 k cycles where the others took 12.4 k --
            // profiles/r06/h_*; no site of config 4 needs the second floor at the posterior.)
            const float nstar2 = lq2nd < 0.0f ? FL * __builtin_amdgcn_rcpf(lq2nd) : 0.0f;
            *reinterpret_cast<float4 *>(srec + lane * SR) = make_float4(a, clr, term0, thr);
            if (zero_closed) *reinterpret_cast<float4 *>(sres + lane * 4) = make_float4(ea * BL_LOG2E, 1.0f, ea, 0.0f);
            *reinterpret_cast<float4 *>(srec + lane * SR + 4) = make_float4(cnon - lqmin, nstar2, lqmin * BL_LOG2E, __int_as_float(jmin));
            *reinterpret_cast<float2 *>(srec + lane * SR + 8) = make_float2(log_z, en_prior); // (parked for this lane itself)
            const bool any_item = __any(nch > 0);    // (a wave of sites without detections has none: no item round at all)
            const int P = any_item ? bl_wave_iscan(nch) : 0;
            BL_RN_T(2)
            // ---- P2: items.  Sites are packed whole into rounds of <= 64 lanes (a site has <= 16 items) ----
            int first = any_item ? 0 : ns, base = 0;
            while (first < ns) {
                const unsigned long long fit = __ballot(P - base <= 64);    // a prefix of the lanes (P is non-decreasing)
                const int last = min(ns, (int)__popcll(fit));               // sites [first, last) go into this round
                const int nitems = __builtin_amdgcn_readlane(P, last - 1) - base;
                const bool mine = lane >= first && lane < last;
                const int start = P - nch - base;
                // (a site's first BL_RN_UNR_STORES items without a loop -- their stores go out together; more: the rare tail)
#pragma unroll
                for (int c = 0; c < BL_RN_UNR_STORES; c++)
                    if (mine && c < nch) imap[start + c] = (unsigned)lane | ((unsigned)c << 8) | ((unsigned)start << 16) | ((unsigned)nch << 24);
                if (__any(mine && nch > BL_RN_UNR_STORES))
                    for (int c = BL_RN_UNR_STORES; __any(mine && c < nch); c += 1 + BL_RN_TWO_PER_ROUND) {
                        if (mine && c < nch) imap[start + c] = (unsigned)lane | ((unsigned)c << 8) | ((unsigned)start << 16) | ((unsigned)nch << 24);
                        if (BL_RN_TWO_PER_ROUND && mine && c + 1 < nch) imap[start + c + 1] = (unsigned)lane | ((unsigned)(c + 1) << 8) | ((unsigned)start << 16) | ((unsigned)nch << 24);
                    }
                bl_wave_lds_fence(); // (also: the sites' dynamic visit slots are written)
                const bool item = lane < nitems;
                const unsigned im = imap[item ? lane : 0];
                const int sl = (int)(im & 0xFFu), ch = (int)((im >> 8) & 0xFFu), st = (int)((im >> 16) & 0xFFu), nc = (int)(im >> 24);
                const int is = r0 + sl;
                const float *ipv = data + (size_t)(is >> 1) * pstride + (is & 1) + 2 * (XQ + t * pb);
                const float4 s0v = *reinterpret_cast<const float4 *>(srec + sl * SR);
                const float4 s1v = *reinterpret_cast<const float4 *>(srec + sl * SR + 4);
                const float i_a = s0v.x, i_clr = s0v.y, i_term0 = s0v.z, i_thr = s0v.w;
                const float i_lq1 = s1v.z;                       // log2 q of the site's first visit to floor (0: no non-detection)
                const bool deep = __any(ch > 0);                // some item is not a chunk 0: the recursions' starting values are needed
                const bool multi = __any(nc > 1);               // some site has several items: combine through LDS
                const float n0f = (float)(ch * BL_RN_CH);
                BL_RN_T(3)
                // LP[i] (log2 units): term of n = 8 ch + 1 + i;  starts as the part that needs no recursion, p_n + n cnon + clr
                float LP[BL_RN_CH];
                {
                    const float4 l0 = *reinterpret_cast<const float4 *>(lgt + ch * BL_RN_CH);
                    const float4 l1 = *reinterpret_cast<const float4 *>(lgt + ch * BL_RN_CH + 4);
                    const float lg[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
                    for (int q = 0; q < BL_RN_CH; q++) {
                        const float nf = n0f + (float)(q + 1);
                        LP[q] = fmaf(fmaf(nf, i_a, -lg[q]) + i_clr, BL_LOG2E, fmaxf(fmaf(-nf, i_lq1, -23.0f), 0.0f)); // (+ the first floor; log2(eps_f32) = -23)
                    }
                }
                // ---- A1: LP[n] += sum over detection visits of log2 b_n ----
                // Visits are taken BL_RN_GA at a time, their b_n recursions side by side as five packed pairs, and ONE log per n for
                // the group: sum_j log b_jn = log prod_j b_jn (b <= n <= 127, so ten factors stay far inside float32).  A
                // non-detection visit gets q = 0, hence b = 1: its factor is 1 and it needs no mask.
                // J <= 10 (one group): the reciprocals 1 / b_jn, which pass C needs, are formed here and kept (80 registers), the
                // product runs over them, and C is left with ten dot products -- no second recursion.
                static_assert(BL_RN_GA == 10, "the product tree below is written for five pairs");
                const bool one_group = J10 || J <= BL_RN_GA; // wave-uniform (a compile-time fact with J10)
                bl_f2 rb[BL_RN_CH][5], q2k[5];
                if (one_group) {
#pragma unroll
                    for (int g = 0; g < BL_RN_GA; g++) {
                        float qv = 0.0f;
                        if (g < J) qv = fmaxf(ipv[2 * (g * VW + KO + 1)], 0.0f); // wave-uniform guard
                        if (g & 1) q2k[g >> 1].y = qv; else q2k[g >> 1].x = qv;
                    }
                    bl_f2 b2[5];
#pragma unroll
                    for (int g = 0; g < 5; g++) b2[g] = bl2(0.0f);
                    if (deep) { // b_(8 ch) = b_8 B,  b_8 = (1+q)(1+q^2)(1+q^4),  B = 1 + q^8 + ... + q^(8 (ch-1))  (0 for ch = 0)
                        bl_f2 q8[5], B[5];
#pragma unroll
                        for (int g = 0; g < 5; g++) {
                            const bl_f2 qq = q2k[g] * q2k[g], q4 = qq * qq;
                            q8[g] = q4 * q4;
                            b2[g] = ((q2k[g] + bl2(1.0f)) * (qq + bl2(1.0f))) * (q4 + bl2(1.0f));
                            B[g] = bl2(0.0f);
                        }
#if BL_RN_TWO_PER_ROUND
                        for (int s = 0; __any(s < ch); s += 2) { // (two steps a round: a round costs a wave-wide `any` and a branch)
                            if (s < ch) { // (exec-masked: the lanes with fewer steps sit this one out)
#pragma unroll
                                for (int g = 0; g < 5; g++) B[g] = bl_fma2(B[g], q8[g], bl2(1.0f));
                            }
                            if (s + 1 < ch) {
#pragma unroll
                                for (int g = 0; g < 5; g++) B[g] = bl_fma2(B[g], q8[g], bl2(1.0f));
                            }
                        }
#else
                        for (int s = 0; __any(s < ch); s++) {
                            if (s < ch) { // (exec-masked: the lanes with fewer steps sit this one out)
#pragma unroll
                                for (int g = 0; g < 5; g++) B[g] = bl_fma2(B[g], q8[g], bl2(1.0f));
                            }
                        }
#endif
#pragma unroll
                        for (int g = 0; g < 5; g++) b2[g] = b2[g] * B[g];
                    }
#pragma unroll
                    for (int n = 0; n < BL_RN_CH; n++) {
                        float rp[5]; // 1 / (b_x b_y) of every pair: ONE reciprocal serves both visits (1 / b_x = b_y rp), and prod_j 1 / b_j = prod rp
#pragma unroll
                        for (int g = 0; g < 5; g++) {
                            b2[g] = bl_fma2(b2[g], q2k[g], bl2(1.0f));
                            rp[g] = __builtin_amdgcn_rcpf(b2[g].x * b2[g].y);
                            rb[n][g] = bl_f2{b2[g].y, b2[g].x} * bl2(rp[g]);
                        }
                        LP[n] -= __builtin_amdgcn_logf(((rp[0] * rp[1]) * (rp[2] * rp[3])) * rp[4]);
                    }
                } else
                for (int j0 = 0; j0 < J; j0 += BL_RN_GA) {
                    bl_f2 b2[5], q2[5];
                    float qmax = 0.0f;
#pragma unroll
                    for (int g = 0; g < BL_RN_GA; g++) {
                        float qv = 0.0f;
                        if (j0 + g < J) qv = fmaxf(ipv[2 * ((j0 + g) * VW + KO + 1)], 0.0f); // wave-uniform guard
                        if (g & 1) q2[g >> 1].y = qv; else q2[g >> 1].x = qv;
                        qmax = fmaxf(qmax, qv);
                    }
                    if (!__any(qmax > 0.0f)) continue; // no lane of the wave has a detection in this group
#pragma unroll
                    for (int g = 0; g < 5; g++) b2[g] = bl2(0.0f);
                    if (deep) {
                        bl_f2 q8[5], B[5];
#pragma unroll
                        for (int g = 0; g < 5; g++) {
                            const bl_f2 qq = q2[g] * q2[g], q4 = qq * qq;
                            q8[g] = q4 * q4;
                            b2[g] = ((q2[g] + bl2(1.0f)) * (qq + bl2(1.0f))) * (q4 + bl2(1.0f));
                            B[g] = bl2(0.0f);
                        }
                        for (int s = 0; __any(s < ch); s++) {
                            if (s < ch) { // (exec-masked: the lanes with fewer steps sit this one out)
#pragma unroll
                                for (int g = 0; g < 5; g++) B[g] = bl_fma2(B[g], q8[g], bl2(1.0f));
                            }
                        }
#pragma unroll
                        for (int g = 0; g < 5; g++) b2[g] = b2[g] * B[g];
                    }
#pragma unroll
                    for (int n = 0; n < BL_RN_CH; n++) {
#pragma unroll
                        for (int g = 0; g < 5; g++) b2[g] = bl_fma2(b2[g], q2[g], bl2(1.0f));
                        const bl_f2 pp = ((b2[0] * b2[1]) * (b2[2] * b2[3])) * b2[4];
                        LP[n] += __builtin_amdgcn_logf(pp.x * pp.y);
                    }
                }
                BL_RN_T(4)
                // ---- A1b: numpyro's floor, second visit onwards (the first is in LP already).  Beyond n2* the TRUE term is at most
                // LP[n] + (cnon - lqmin) (n2* - n): only an item with such an n within 20 nats of m_lb needs LP[n] += sum_j max(0, FL - n log q_j)
                // over the OTHER visits, and then only for those whose floor one of its n reaches (taking the floor where it does not
                // matter is still the model, so a visit picked by any lane is corrected in all). ----
                unsigned long long floored = 0ull; // wave-uniform: visits (bit j & 63) that took the correction
                {
                    const float i_cnon1 = s1v.x, i_nstar2 = s1v.y;
                    bool rel = false;
#pragma unroll
                    for (int n = 0; n < BL_RN_CH; n++) {
                        const float nf = n0f + (float)(n + 1);
                        rel = rel || (nf > i_nstar2 && fmaf(LP[n], BL_LN2, i_cnon1 * (i_nstar2 - nf)) >= i_thr);
                    }
                    rel = rel && item && i_nstar2 > 0.0f;
                    if (__any(rel)) {
                        const float nmax = n0f + (float)BL_RN_CH;
                        const int i_j1 = __float_as_int(s1v.w);
                        for (int j = 0; j < J; j++) {
                            const float lq2 = j == i_j1 ? 0.0f : fminf(ipv[2 * (j * VW + KO + 1)], 0.0f); // log2 q of a non-detection other than the first to floor, else 0
                            if (!__any(rel && nmax * lq2 < -23.0f)) continue;          // log2(eps_f32) = -23
                            floored |= 1ull << (j & 63);
#pragma unroll
                            for (int n = 0; n < BL_RN_CH; n++) LP[n] += fmaxf(fmaf(-(n0f + (float)(n + 1)), lq2, -23.0f), 0.0f);
                        }
                    }
                }
                // ---- B: the item's maximum and sums, then the site's over its items (one exchange through LDS) ----
                float m_it = ch == 0 ? i_term0 * BL_LOG2E : -3.0e38f; // the n = 0 term rides with chunk 0
#pragma unroll
                for (int n = 0; n < BL_RN_CH; n++) m_it = fmaxf(m_it, LP[n]);
                float S = ch == 0 ? __builtin_amdgcn_exp2f(fmaf(i_term0, BL_LOG2E, -m_it)) : 0.0f, a1 = 0.0f;
#pragma unroll
                for (int n = 0; n < BL_RN_CH; n++) {
                    const float wn = __builtin_amdgcn_exp2f(LP[n] - m_it);
                    const float nw = (n0f + (float)(n + 1)) * wn;
                    LP[n] = nw; // n x the posterior weight of N = n, relative to the item's best term
                    S += wn;
                    a1 += nw;
                }
                const float a1_it = a1;
                float m_s = m_it, f_it = 1.0f; // f_it: the item's weights relative to the site's best term
                if (multi) *reinterpret_cast<float4 *>(comb + lane * 4) = make_float4(m_it, S, a1, 0.0f);
                // (pass C's dot products need the item's own weights only: they run here, in the shadow of that LDS round trip)
                bl_f2 h[5];
                if (one_group) {
#pragma unroll
                    for (int g = 0; g < 5; g++) h[g] = bl2(0.0f);
#pragma unroll
                    for (int n = 0; n < BL_RN_CH; n++) {
#pragma unroll
                        for (int g = 0; g < 5; g++) h[g] = bl_fma2(bl2(LP[n]), rb[n][g], h[g]);
                    }
                }
                if (multi) {
                    bl_wave_lds_fence();
                    // The first four items of the site (n <= 32: all but one site in two hundred at config 4) are read in ONE go -- four
                    // float4 loads in flight, then the maximum and the sums from registers; until round 5 every item was a round trip of its
                    // own in each of two loops (floors + B + combine: 3 809 -> of the 14.7 k-cycle evaluation).  Same order of the sums.
                    float4 cv[4];
#pragma unroll
                    for (int s = 0; s < 4; s++) cv[s] = *reinterpret_cast<const float4 *>(comb + (st + min(s, nc - 1)) * 4);
                    const bool more = __any(nc > 4);
#pragma unroll
                    for (int s = 0; s < 4; s++) m_s = fmaxf(m_s, cv[s].x); // (an item read twice changes no maximum)
                    if (more)
                        for (int s = 4; __any(s < nc); s++) m_s = fmaxf(m_s, comb[(st + min(s, nc - 1)) * 4]);
                    f_it = __builtin_amdgcn_exp2f(m_it - m_s);
                    S = 0.0f; a1 = 0.0f;
#pragma unroll
                    for (int s = 0; s < 4; s++) { // (fixed order: every item of a site forms the same sums)
                        const float f = s < nc ? __builtin_amdgcn_exp2f(cv[s].x - m_s) : 0.0f;
                        S = fmaf(f, cv[s].y, S);
                        a1 = fmaf(f, cv[s].z, a1);
                    }
                    if (more)
                        for (int s = 4; __any(s < nc); s++) {
                            const float4 v = *reinterpret_cast<const float4 *>(comb + (st + min(s, nc - 1)) * 4);
                            const float f = s < nc ? __builtin_amdgcn_exp2f(v.x - m_s) : 0.0f;
                            S = fmaf(f, v.y, S);
                            a1 = fmaf(f, v.z, a1);
                        }
                }
                if (item && ch == 0) *reinterpret_cast<float4 *>(sres + sl * 4) = make_float4(m_s, S, a1, 0.0f);
                const float rs = item ? f_it * __builtin_amdgcn_rcpf(S) : 0.0f;
                BL_RN_T(5)
                // floored visits: d/du max(n log sigma(u), FL) vanishes for the floored n -- take their n w_n back out of the rank-1 part.
                // The first visit to floor: its share is formed by every item (LP holds n w_n here) and is zero wherever no n of the item
                // lies beyond n*; the visit's covariates are fetched only by a wave in which some item has one.
                {
                    float hf1 = 0.0f;
#pragma unroll
                    for (int n = 0; n < BL_RN_CH; n++) hf1 += ((n0f + (float)(n + 1)) * i_lq1 < -23.0f) ? LP[n] : 0.0f;
                    // (hf1 rs = the posterior mass, times n, that the item holds beyond n*: below 2e-9 it is dropped like every term the
                    // sites' thresholds drop)
                    if (__any(hf1 * rs > 2.0e-9f)) {
                        const float *wv = ipv + 2 * (__float_as_int(*(srec + sl * SR + 7)) * VW);
                        const float dnu = -(1.0f - __builtin_amdgcn_exp2f(i_lq1)) * hf1 * rs; // r = 1 - q of that non-detection
#pragma unroll
                        for (int k = 0; k <= KO; k++) ga[k] = fmaf(dnu, wv[2 * k], ga[k]);
                    }
                }
                if (floored != 0ull) {
                    const int i_j1 = __float_as_int(*(srec + sl * SR + 7));
                    for (int j = 0; j < J; j++) {
                        if (!((floored >> (j & 63)) & 1ull)) continue;
                        const float *wv = ipv + 2 * (j * VW);
                        const float lq2 = j == i_j1 ? 0.0f : fminf(wv[2 * (KO + 1)], 0.0f);
                        const float sm = 1.0f - __builtin_amdgcn_exp2f(lq2); // r = sigma(-u) of a non-detection (a floor is reached where q <= 0.88)
                        float hf = 0.0f;
#pragma unroll
                        for (int n = 0; n < BL_RN_CH; n++) hf += ((n0f + (float)(n + 1)) * lq2 < -23.0f) ? LP[n] : 0.0f;
                        const float dnu = -sm * hf * rs;
#pragma unroll
                        for (int k = 0; k <= KO; k++) ga[k] = fmaf(dnu, wv[2 * k], ga[k]);
                    }
                }
                // ---- C: detection visits' d/dnu = sum_n w_n n q^n / b_n   (d log(1 - q^n) / dnu = n r q^n / (1 - q^n)), and
                // q^n / b_n = 1 / b_n - r  (1 - q^n = r b_n):  sum_n (n w_n) / b_jn - r_j sum_n n w_n  -- no q^n recursion ----
                if (one_group) {
#pragma unroll
                    for (int g = 0; g < BL_RN_GA; g++) {
                        if (g < J) { // wave-uniform
                            const float *wv = ipv + 2 * (g * VW);
                            // (q_j is read again rather than kept: ten registers less across the combine -- the kernel sits at its 256-register budget)
                            const float qv = fmaxf(wv[2 * (KO + 1)], 0.0f), hv = (g & 1) ? h[g >> 1].y : h[g >> 1].x;
                            const float dnu = qv > 0.0f ? fmaf(qv - 1.0f, a1_it, hv) * rs : 0.0f;
#pragma unroll
                            for (int k = 0; k <= KO; k++) ga[k] = fmaf(dnu, wv[2 * k], ga[k]);
                        }
                    }
                } else
                for (int j0 = 0; j0 < J; j0 += 2) {
                    const float *wv0 = ipv + 2 * (j0 * VW), *wv1 = ipv + 2 * (min(j0 + 1, J - 1) * VW);
                    bl_f2 q2;
                    q2.x = fmaxf(wv0[2 * (KO + 1)], 0.0f);
                    q2.y = j0 + 1 < J ? fmaxf(wv1[2 * (KO + 1)], 0.0f) : 0.0f;
                    if (!__any(fmaxf(q2.x, q2.y) > 0.0f)) continue;
                    bl_f2 b = bl2(0.0f), h = bl2(0.0f);
                    if (deep) {
                        const bl_f2 qq = q2 * q2, q4 = qq * qq, q8 = q4 * q4;
                        const bl_f2 b8 = ((q2 + bl2(1.0f)) * (qq + bl2(1.0f))) * (q4 + bl2(1.0f));
                        bl_f2 B = bl2(0.0f);
                        for (int s = 0; __any(s < ch); s++)
                            if (s < ch) B = bl_fma2(B, q8, bl2(1.0f));
                        b = b8 * B;
                    }
#pragma unroll
                    for (int n = 0; n < BL_RN_CH; n++) {
                        b = bl_fma2(b, q2, bl2(1.0f));
                        h = bl_fma2(bl2(LP[n]), bl_rcp_2(b), h);
                    }
                    const float dnu0 = q2.x > 0.0f ? fmaf(q2.x - 1.0f, a1_it, h.x) * rs : 0.0f;
                    const float dnu1 = q2.y > 0.0f ? fmaf(q2.y - 1.0f, a1_it, h.y) * rs : 0.0f;
#pragma unroll
                    for (int k = 0; k <= KO; k++) ga[k] = fmaf(dnu0, wv0[2 * k], ga[k]);
#pragma unroll
                    for (int k = 0; k <= KO; k++) ga[k] = fmaf(dnu1, wv1[2 * k], ga[k]);
                }
                BL_RN_T(6)
                first = last;
                base += nitems;
                bl_wave_lds_fence(); // (the next round rewrites the item map and the combine table)
            }
            // ---- back in the site's lane: log-lik, E[n | y], the non-detections' rank-1 gradient ----
            float4 res = *reinterpret_cast<const float4 *>(sres + lane * 4);
            if (!has) res = make_float4(0.0f, 1.0f, 0.0f, 0.0f); // (no item wrote this lane's slot)
            const float en_post = res.z * __builtin_amdgcn_rcpf(res.y);
            const float2 parked = *reinterpret_cast<const float2 *>(srec + lane * SR + 8);  // log Z, E[n] of the prior
            ll_s += BL_LN2 * (res.x + __builtin_amdgcn_logf(res.y)) - parked.x;
            deta += en_post - parked.y;
            const float enl = live * en_post;
#pragma unroll
            for (int k = 0; k <= KO; k++) ga[k] = fmaf(enl, Rv[k], ga[k]);
            BL_RN_T(7)
        }
        deta *= live;
        ll = fmaf(live, ll_s, ll);
        gb[0] += deta;
#pragma unroll
        for (int k = 0; k < KS; k++) gb[k + 1] = fmaf(deta, rec[2 * k], gb[k + 1]);
    }
}
