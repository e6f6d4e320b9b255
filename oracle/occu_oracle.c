/*
 * occu_oracle.c -- CPU ORACLE (test infrastructure, NOT a product path).
 *
 * Plain-C, float64 restatement of the hot path of timmh/biolith's
 *     fit(occu, ...)            biolith/utils/fit.py:16-135
 * i.e. NUTS over the z-marginalised occupancy log-density of
 *     occu()                    biolith/models/occu.py:136-242  (incl. false positives, :146-157, 229-241, and site /
 *                               observation random effects, :170-173, 191-196, 215-218)
 *     occu_rn()                 biolith/models/occu_rn.py:123-222 + utils/distributions.py:6-40
 *     occu_cop()                biolith/models/occu_cop.py:150-255
 *     nmixture()                biolith/models/nmixture.py:150-220
 *     LinearRegression          biolith/regression/linear.py:28-66  (Normal or Laplace coefficient priors: the two families
 *                               biolith/utils/grid_search.py:366-371 passes)
 *     mask_missing_obs          biolith/utils/modeling.py:8-19
 * and, with NO reference counterpart (BASELINE.json configs[4] names a model the reference does not have),
 * the builder-defined dynamic occupancy model (potential_grad_dyn: initial occupancy, colonisation, extinction; forward algorithm).
 * plus the sampler those lines delegate to, which is NOT in /root/reference:
 * numpyro (>=0.18, un-pinned; pyproject.toml:22-28) numpyro.infer.hmc /
 * hmc_util (NUTS with iterative tree building, dual averaging, Welford
 * diagonal mass adaptation) restated from its published algorithm
 * (SURVEY.md Appendix B).
 *
 * PARITY STATUS (round 5).  The LOG-DENSITIES are PINNED to outputs of the reference itself: its own model functions
 * (occu.py:136-242, occu_rn.py:123-222, occu_cop.py:150-255, nmixture.py:150-220 with regression/linear.py, utils/modeling.py,
 * utils/distributions.py) were executed in the build container under a functional NumPy shim of the numpyro / jax names they use
 * (tests/golden/make_reference_logjoint.py, committed with its 34 x 5 log-joint values, tests/golden/reference_logjoint_*.json);
 * potential_grad equals them to 2e-15 relative (tests/test_reference_logjoint.py), its gradient their central differences.  The
 * SIMULATORS are pinned bit for bit (tests/golden/make_golden.py).  The SAMPLER stays "parity unpinned" against numpyro's trees --
 * numpyro / jax are not installable here and the reference holds no golden vector for it -- and is held
 * instead to the DISTRIBUTION it must sample: numerical integration of small posteriors (tests/quadrature.py) and simulation-based
 * calibration under the reference's generative models (tests/sbc.py), besides the reference's statistical-recovery tolerances
 * (occu.py:440-456) and the published adaptation-window schedules; the oracle's own draws for fixed seeds
 * (tests/golden/oracle_first_draws.json) guard it against drift.  What remains upstream-assumed is listed in DESIGN.md section 3.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's library.  The product (biolith_amd) never does.
 *
 * Deliberate, documented deviations from the reference arithmetic:
 *   - float64 throughout (reference: float32, utils/data.py:135-140);
 *   - the z=1 branch uses exact log-sigmoid forms; numpyro clamps Bernoulli
 *     probs to [tiny, 1-eps] which only matters for |nu| > 15.9;
 *   - the z=0 branch keeps numpyro's clamp: P(y=1|z=0) = tiny(float32), so a
 *     detection contributes log(tiny) = -87.3365 (SURVEY.md App. A note ii);
 *   - the tree-doubling loop stops when the new subtree turned internally OR the whole-tree criterion
 *     fires (numpyro _combine_tree; SURVEY.md App. B.2 states only the latter);
 *   - RNG is xoshiro128++ (jump-separated streams), not JAX threefry:
 *     draw-level equality with the reference is impossible by construction.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_NSTREAM 64          /* RNG streams per chain                     */
#define ORC_SCALAR_STREAM 63    /* stream of the multinomial-transition uniforms  */
#define ORC_DIR_STREAM 62       /* stream of the tree-doubling direction bits     */
#define ORC_MAX_DEPTH 10        /* numpyro NUTS max_tree_depth default        */
#define ORC_MAX_D 61            /* fixed-effect models: stack arrays of this size            */
#define ORC_BIG_D 65536         /* random-effects model: heap vectors                        */

static const double LOG_TINY_F32 = -87.33654475055310898657; /* log(1.1754943508222875e-38) */
static const double HALF_LOG_2PI = 0.91893853320467274178;

/* ------------------------------------------------------------------ RNG -- */
/* xoshiro128++ 1.0 (Blackman & Vigna, public domain algorithm).             */
static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

uint32_t orc_rng_next(uint32_t s[4])
{
    const uint32_t result = rotl32(s[0] + s[3], 7) + s[0];
    const uint32_t t = s[1] << 9;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl32(s[3], 11);
    return result;
}

/* advance by 2^64 draws */
void orc_rng_jump(uint32_t s[4])
{
    static const uint32_t JUMP[4] = {0x8764000bu, 0xf542d2d3u, 0x6fa035c3u, 0x77f2db5bu};
    uint32_t a = 0, b = 0, c = 0, d = 0;
    for (int i = 0; i < 4; i++)
        for (int bit = 0; bit < 32; bit++) {
            if (JUMP[i] & (1u << bit)) { a ^= s[0]; b ^= s[1]; c ^= s[2]; d ^= s[3]; }
            orc_rng_next(s);
        }
    s[0] = a; s[1] = b; s[2] = c; s[3] = d;
}

static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

/* Stream (chain, s) = base state advanced by (chain*stride + s) jumps; stride = ORC_NSTREAM for the models whose
 * parameters fit one stream each below the two scalar streams, max(ORC_NSTREAM, D + 2) for the random-effects model. */
void orc_rng_streams_strided(uint64_t seed, int chain, int stride, int nstreams, uint32_t *out /*[nstreams][4]*/)
{
    uint64_t sm = seed;
    uint64_t a = splitmix64(&sm), b = splitmix64(&sm);
    uint32_t s[4] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    if (!(s[0] | s[1] | s[2] | s[3])) s[0] = 1;
    for (long k = 0; k < (long)chain * stride; k++) orc_rng_jump(s);
    for (int k = 0; k < nstreams; k++) {
        memcpy(out + 4 * k, s, sizeof s);
        orc_rng_jump(s);
    }
}
void orc_rng_streams(uint64_t seed, int chain, int nstreams, uint32_t *out /*[nstreams][4]*/)
{
    uint64_t sm = seed;
    uint64_t a = splitmix64(&sm), b = splitmix64(&sm);
    uint32_t s[4] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    if (!(s[0] | s[1] | s[2] | s[3])) s[0] = 1;
    for (long k = 0; k < (long)chain * ORC_NSTREAM; k++) orc_rng_jump(s);
    for (int k = 0; k < nstreams; k++) {
        memcpy(out + 4 * k, s, sizeof s);
        orc_rng_jump(s);
    }
}

/* uniform in (0,1), exactly representable in float32 */
static inline double rng_uniform(uint32_t s[4])
{
    return ((double)(orc_rng_next(s) >> 9) + 0.5) * (1.0 / 8388608.0);
}

/* one standard normal per call (Box-Muller, cosine branch; sine discarded) */
static inline double rng_normal(uint32_t s[4])
{
    double u1 = rng_uniform(s), u2 = rng_uniform(s);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}

double orc_rng_uniform(uint32_t s[4]) { return rng_uniform(s); }
double orc_rng_normal(uint32_t s[4]) { return rng_normal(s); }

/* --------------------------------------------------------- log-density -- */
typedef struct {
    int N, T, J, Ks, Ko, D;
    double *X;       /* [N][Ks]        NaN -> 0      (occu.py:142) */
    double *W;       /* [N][T][J][Ko]  NaN -> 0      (occu.py:141) */
    signed char *C;  /* [N][T][J] +1 detection, -1 non-detection, 0 masked (occu.py:136-140, modeling.py:15-17) */
    double loc_b, scale_b, loc_a, scale_a; /* Normal priors, occu.py:28-29 */
    int laplace_b, laplace_a;              /* 1: that prior is Laplace(loc, scale) instead (grid_search.py:366-371) */
    int model;         /* 0 = occu (occu.py), 1 = occu_rn (occu_rn.py), 2 = occu with a false-positive rate */
    int max_abundance; /* occu_rn.py:26 */
    int rn_no_cut;     /* 1: potential_grad_rn sums every n <= max_abundance (tests); 0: it stops where the terms have died out */
    int fp_mode;       /* model 2/3: 1 = false_positives_constant, 2 = false_positives_unoccupied (occu.py:146-157); model 3 also 0 */
    double fp_a, fp_b; /* Beta(a, b) prior of that rate (occu.py:32-33); model 3: Exponential(rate = fp_a) (occu_cop.py:31-32) */
    int site_re, obs_re;            /* model 6: occu with random effects (occu.py:170-173, 191-196, 215-218) */
    double re_scale_site, re_scale_obs; /* HalfNormal(scale) priors of site_re_sd / obs_re_sd (occu.py:38-39) */
    double cs_mu[4];   /* model 7 (occu_cs): Normal(loc, scale) of mu0, then of the base of mu1 (truncated below at mu0), occu_cs.py:143-148 */
    double cs_sg[4];   /* model 7: Gamma(concentration, rate) of sigma0, then of sigma1 (occu_cs.py:149-152) */
    int S;             /* species sampled JOINTLY (0 reads as 1): theta = [species 0: beta, alpha | species 1: ... | (phi)], the species
                          plate of occu.py:182-186 under one NUTS; C then holds S blocks of [N][T][J].  Models 0 and 2. */
    int skip_phi_prior; /* (internal: the shared false-positive rate's prior is counted once) */
    double *Cnt;       /* model 3: counts [N][T][J] (valid where C != 0)   occu_cop.py:250-254; model 7: the scores */
    double *Dur;       /* model 3: session_duration [N][T][J]              occu_cop.py:176 */
} orc_data;

/* log prior density of regression coefficient k (beta for k <= Ks, else alpha) at v, and its derivative added to *g:
 * Normal(loc, scale) (occu.py:28-29) or Laplace(loc, scale) (the other family grid_search_priors tries,
 * utils/grid_search.py:366-371): -|v - loc| / scale - log(2 scale), derivative -sign(v - loc) / scale (0 at the kink). */
static inline double prior_coef(const orc_data *d, int k, double v, double *g);
static inline double softplus(double x) { return x > 0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
static inline double expit(double x)
{
    if (x >= 0) return 1.0 / (1.0 + exp(-x));
    double e = exp(x);
    return e / (1.0 + e);
}
static inline double logaddexp(double a, double b)
{
    double m = a > b ? a : b;
    if (isinf(m) && m < 0) return m;
    return m + log1p(exp(-fabs(a - b)));
}

static inline double prior_coef(const orc_data *d, int k, double v, double *g)
{
    const int is_b = k <= d->Ks;
    const double loc = is_b ? d->loc_b : d->loc_a, sc = is_b ? d->scale_b : d->scale_a;
    const int laplace = is_b ? d->laplace_b : d->laplace_a;
    const double zz = (v - loc) / sc;
    if (laplace) {
        *g += -((zz > 0.0) - (zz < 0.0)) / sc;
        return -fabs(zz) - log(2.0 * sc);
    }
    *g += -zz / sc;
    return -0.5 * zz * zz - log(sc) - HALF_LOG_2PI;
}
void orc_data_set_prior_family(orc_data *d, int laplace_beta, int laplace_alpha) { d->laplace_b = laplace_beta != 0; d->laplace_a = laplace_alpha != 0; }

orc_data *orc_data_create(int N, int T, int J, int Ks, int Ko,
                          const double *site_covs /*[N][Ks]*/,
                          const double *obs_covs /*[N][T][J][Ko]*/,
                          const double *obs /*[N][T][J], NaN = missing*/,
                          double loc_b, double scale_b, double loc_a, double scale_a)
{
    orc_data *d = (orc_data *)calloc(1, sizeof *d);
    d->N = N; d->T = T; d->J = J; d->Ks = Ks; d->Ko = Ko; d->D = Ks + 1 + Ko + 1;
    d->loc_b = loc_b; d->scale_b = scale_b; d->loc_a = loc_a; d->scale_a = scale_a;
    d->X = (double *)malloc(sizeof(double) * (size_t)N * (Ks > 0 ? Ks : 1));
    d->W = (double *)malloc(sizeof(double) * (size_t)N * T * J * (Ko > 0 ? Ko : 1));
    d->C = (signed char *)malloc((size_t)N * T * J);
    for (int i = 0; i < N; i++) {
        int site_nan = 0;
        for (int k = 0; k < Ks; k++) {
            double v = site_covs[(size_t)i * Ks + k];
            if (isnan(v)) { site_nan = 1; v = 0.0; }
            d->X[(size_t)i * Ks + k] = v;
        }
        for (int v = 0; v < T * J; v++) {
            size_t o = (size_t)i * T * J + v;
            int cov_nan = site_nan;
            for (int k = 0; k < Ko; k++) {
                double w = obs_covs[o * Ko + k];
                if (isnan(w)) { cov_nan = 1; w = 0.0; }
                d->W[o * Ko + k] = w;
            }
            double y = obs[o];
            /* mask = isfinite(obs) after covariate-NaN poisoning */
            if (cov_nan || !isfinite(y)) d->C[o] = 0;
            else d->C[o] = (y != 0.0) ? 1 : -1; /* TODO in reference: non-binary obs unchecked (occu.py:110-111) */
        }
    }
    return d;
}

void orc_data_destroy(orc_data *d)
{
    if (!d) return;
    free(d->X); free(d->W); free(d->C); free(d->Cnt); free(d->Dur); free(d);
}

int orc_data_dim(const orc_data *d) { return d->D; }
void orc_data_set_model(orc_data *d, int model, int max_abundance) { d->model = model; d->max_abundance = max_abundance; }
/* occu with false positives: theta gains a trailing coordinate phi = logit(rate) (numpyro's
 * unconstrained space for a unit-interval site: biject_to(unit_interval) = sigmoid). */
void orc_data_set_fp(orc_data *d, int fp_mode, double a, double b)
{
    d->model = 2; d->fp_mode = fp_mode; d->fp_a = a; d->fp_b = b;
    d->D = d->Ks + 1 + d->Ko + 1 + 1;
}

/* Count occupancy model (biolith/models/occu_cop.py:17-255): obs are counts, session_duration the exposure.
 * fp_mode 0 / 1 / 2 = no false-positive rate / rate_fp_constant / rate_fp_unoccupied with an
 * Exponential(rate) prior; then theta gains phi = log(rate) (biject_to(positive) = exp). */
void orc_data_set_cop(orc_data *d, const double *obs /*[N][T][J]*/, const double *dur /*[N][T][J]*/, int fp_mode, double rate)
{
    const size_t n = (size_t)d->N * d->T * d->J;
    d->model = 3; d->fp_mode = fp_mode; d->fp_a = rate;
    d->D = d->Ks + 1 + d->Ko + 1 + (fp_mode ? 1 : 0);
    d->Cnt = (double *)malloc(sizeof(double) * n);
    d->Dur = (double *)malloc(sizeof(double) * n);
    for (size_t o = 0; o < n; o++) { d->Cnt[o] = d->C[o] ? obs[o] : 0.0; d->Dur[o] = dur[o]; }
}

/* log Poisson(y; rate), numpyro Poisson.log_prob: xlogy(y, rate) - lgamma(y+1) - rate  (rate 0: 0 or -inf) */
static double poisson_logpmf(double y, double rate, double *dlog_drate)
{
    if (rate <= 0.0) { *dlog_drate = 0.0; return y > 0.0 ? -INFINITY : 0.0; }
    *dlog_drate = y / rate - 1.0;
    return (y > 0.0 ? y * log(rate) : 0.0) - lgamma(y + 1.0) - rate;
}

static double potential_grad_cop(const orc_data *d, const double *th, double *grad)
{
    const int N = d->N, T = d->T, J = d->J, Ks = d->Ks, Ko = d->Ko, D = d->D;
    const double *beta = th, *alpha = th + Ks + 1;
    const double f = d->fp_mode ? exp(th[D - 1]) : 0.0;
    const double f_c = d->fp_mode == 1 ? f : 0.0, f_u = d->fp_mode == 2 ? f : 0.0;
    double gl[ORC_MAX_D], ga1[ORC_MAX_D];
    for (int k = 0; k < D; k++) gl[k] = 0.0;
    double ll = 0.0;
    for (int i = 0; i < N; i++) {
        const double *x = d->X + (size_t)i * Ks;
        double eta = beta[0];
        for (int k = 0; k < Ks; k++) eta += x[k] * beta[k + 1];
        const double log_psi = -softplus(-eta), log_1mpsi = -softplus(eta), psi = expit(eta);
        double dl_deta = 0.0;
        for (int t = 0; t < T; t++) {
            double a1 = 0.0, a0 = 0.0, df1 = 0.0, df0 = 0.0;
            for (int k = 0; k <= Ko; k++) ga1[k] = 0.0;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                if (!d->C[o]) continue;                       /* mask_missing_obs, occu_cop.py:150-156,249 */
                const double *w = d->W + o * Ko;
                double nu = alpha[0];
                for (int k = 0; k < Ko; k++) nu += w[k] * alpha[k + 1];
                const double lam = exp(nu), dur = d->Dur[o], y = d->Cnt[o];
                /* l_det = z lambda + (1 - z) f_u + f_c  (occu_cop.py:244-248) */
                double dl1, dl0;
                a1 += poisson_logpmf(y, dur * (lam + f_c), &dl1);
                a0 += poisson_logpmf(y, dur * (f_u + f_c), &dl0);
                const double r = dl1 * dur * lam;             /* d/dnu */
                ga1[0] += r;
                for (int k = 0; k < Ko; k++) ga1[k + 1] += r * w[k];
                if (d->fp_mode == 1) df1 += dl1 * dur;        /* d rate1 / d f_c */
                if (d->fp_mode) df0 += dl0 * dur;             /* d rate0 / d f */
            }
            const double A = log_psi + a1, B = log_1mpsi + a0;
            const double l = logaddexp(A, B);
            const double q = exp(A - l);
            ll += l;
            dl_deta += q - psi;
            for (int k = 0; k <= Ko; k++) gl[Ks + 1 + k] += q * ga1[k];
            if (d->fp_mode) gl[D - 1] += (q * df1 + (1.0 - q) * df0) * f;
        }
        gl[0] += dl_deta;
        for (int k = 0; k < Ks; k++) gl[k + 1] += dl_deta * x[k];
    }
    for (int k = 0; k < Ks + Ko + 2; k++) {
        ll += prior_coef(d, k, th[k], &gl[k]);
    }
    if (d->fp_mode) { /* Exponential(rate) log-density of f plus the Jacobian log f = phi */
        ll += log(d->fp_a) - d->fp_a * f + th[D - 1];
        gl[D - 1] += -d->fp_a * f + 1.0;
    }
    for (int k = 0; k < D; k++) grad[k] = -gl[k];
    return -ll;
}

/* ---- N-mixture model (biolith/models/nmixture.py:17-220, Royle 2004) ----
 * abundance lambda_i = exp(beta0 + x_i beta); N_it enumerated over 0..K with weights Poisson(lambda).pmf(n), set to
 * zero below the largest count of the (site, period) (nmixture.py:150-155,185-190).  The model adds
 * factor(logsumexp(logits)) next to Categorical(logits) (nmixture.py:191-196), which cancels the Categorical's
 * normalisation: the weights are the raw truncated Poisson pmf, NOT renormalised (unlike occu_rn).
 * y_itj ~ Binomial(N_it, p_itj), p = sigmoid(alpha0 + w alpha), with numpyro's BinomialProbs.log_prob (no clamp).
 * obs are counts (d->Cnt).  Written term by term from the model text. */
void orc_data_set_nmix(orc_data *d, const double *obs /*[N][T][J]*/, int max_abundance)
{
    const size_t n = (size_t)d->N * d->T * d->J;
    d->model = 4; d->max_abundance = max_abundance;
    d->Cnt = (double *)malloc(sizeof(double) * n);
    for (size_t o = 0; o < n; o++) d->Cnt[o] = d->C[o] ? obs[o] : 0.0;
}

static double potential_grad_nmix(const orc_data *d, const double *th, double *grad)
{
    const int N = d->N, T = d->T, J = d->J, Ks = d->Ks, Ko = d->Ko, D = d->D, K = d->max_abundance;
    const double *beta = th, *alpha = th + Ks + 1;
    double gl[ORC_MAX_D];
    for (int k = 0; k < D; k++) gl[k] = 0.0;
    double ll = 0.0;
    double *term = (double *)malloc(sizeof(double) * (K + 1));
    double *pj = (double *)malloc(sizeof(double) * (J > 0 ? J : 1));
    for (int i = 0; i < N; i++) {
        const double *x = d->X + (size_t)i * Ks;
        double eta = beta[0];
        for (int k = 0; k < Ks; k++) eta += x[k] * beta[k + 1];
        const double lam = exp(eta);
        double dl_deta = 0.0;
        for (int t = 0; t < T; t++) {
            int min_count = 0;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                pj[j] = 0.0;
                if (!d->C[o]) continue;
                const double *w = d->W + o * Ko;
                double nu = alpha[0];
                for (int k = 0; k < Ko; k++) nu += w[k] * alpha[k + 1];
                pj[j] = expit(nu);
                if ((int)d->Cnt[o] > min_count) min_count = (int)d->Cnt[o];
            }
            double mx = -INFINITY;
            for (int n = 0; n <= K; n++) {
                if (n < min_count) { term[n] = -INFINITY; continue; }
                double v = (n > 0 ? n * eta : 0.0) - lam - lgamma(n + 1.0);            /* Poisson(lambda).log_prob(n) */
                for (int j = 0; j < J; j++) {
                    size_t o = ((size_t)i * T + t) * J + j;
                    if (!d->C[o]) continue;
                    const double y = d->Cnt[o], p = pj[j];
                    v += lgamma(n + 1.0) - lgamma(y + 1.0) - lgamma(n - y + 1.0)
                         + (y > 0 ? y * log(p) : 0.0) + (n - y > 0 ? (n - y) * log1p(-p) : 0.0);  /* Binomial(n, p).log_prob(y) */
                }
                term[n] = v;
                if (v > mx) mx = v;
            }
            if (!(mx > -INFINITY)) { ll += -INFINITY; continue; }  /* max_abundance below an observed count */
            double S = 0.0, En = 0.0;
            for (int n = 0; n <= K; n++) { const double e = exp(term[n] - mx); S += e; En += n * e; }
            ll += mx + log(S);
            En /= S;                                                 /* posterior mean of N */
            dl_deta += En - lam;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                if (!d->C[o]) continue;
                const double *w = d->W + o * Ko;
                const double r = d->Cnt[o] - En * pj[j];             /* d/dnu of y log p + (n - y) log(1 - p), averaged over n */
                gl[Ks + 1] += r;
                for (int k = 0; k < Ko; k++) gl[Ks + 2 + k] += r * w[k];
            }
        }
        gl[0] += dl_deta;
        for (int k = 0; k < Ks; k++) gl[k + 1] += dl_deta * x[k];
    }
    free(term); free(pj);
    for (int k = 0; k < D; k++) {
        ll += prior_coef(d, k, th[k], &gl[k]);
    }
    for (int k = 0; k < D; k++) grad[k] = -gl[k];
    return -ll;
}

/* ---- dynamic (multi-season) occupancy: BUILDER-DEFINED, NO REFERENCE COUNTERPART ----
 * BASELINE.json configs[4] asks for "multi-season dynamic occupancy (colonisation / extinction), forward-algorithm kernel"; the
 * reference has no such model (its periods share one psi, occu.py:198-210).  The model built here is the
 * standard one (MacKenzie et al. 2003), stated with the reference's own pieces (LinearRegression with Normal priors, the
 * occupancy model's detection layer and its numpyro clamp):
 *     z_i1 ~ Bernoulli(psi_i),                    logit psi_i   = x_i b_psi
 *     z_i,t+1 | z_it = 0 ~ Bernoulli(gamma_i),    logit gamma_i = x_i b_gamma      (colonisation)
 *     z_i,t+1 | z_it = 1 ~ Bernoulli(1 - eps_i),  logit eps_i   = x_i b_eps        (extinction)
 *     y_itj | z_it ~ Bernoulli(z_it p_itj),       logit p_itj   = w_itj alpha      (a detection at z = 0 costs log tiny_f32, as in occu)
 * theta = [b_psi (Ks+1) | b_gamma (Ks+1) | b_eps (Ks+1) | alpha (Ko+1)], every coefficient Normal(loc, scale) as in occu.
 * The latent paths z_i. are summed out by the scaled forward recursion; the gradient comes from the smoothed marginals
 * rho_t = P(z_t = 1 | y) and the pairwise xi_t(a, b) = P(z_t = a, z_t+1 = b | y) of the backward pass.
 * Pinned by: brute force over all 2^T paths (oracle.literal_log_joint_dyn), central differences.                         */
void orc_data_set_dyn(orc_data *d)
{
    d->model = 8;
    d->D = 3 * (d->Ks + 1) + d->Ko + 1;
}

static double potential_grad_dyn(const orc_data *d, const double *th, double *grad)
{
    const int N = d->N, T = d->T, J = d->J, Ks = d->Ks, Ko = d->Ko, D = d->D, B = Ks + 1;
    const double *bp = th, *bg = th + B, *be = th + 2 * B, *alpha = th + 3 * B;
    double gl[ORC_MAX_D];
    for (int k = 0; k < D; k++) gl[k] = 0.0;
    double *la = (double *)malloc(sizeof(double) * T), *lb = (double *)malloc(sizeof(double) * T);
    double *phi = (double *)malloc(sizeof(double) * T), *pi = (double *)malloc(sizeof(double) * (T + 1));
    double *ga = (double *)malloc(sizeof(double) * T * (Ko + 1));
    double ll = 0.0;
    for (int i = 0; i < N; i++) {
        const double *x = d->X + (size_t)i * Ks;
        double e_psi = bp[0], e_gam = bg[0], e_eps = be[0];
        for (int k = 0; k < Ks; k++) { e_psi += x[k] * bp[k + 1]; e_gam += x[k] * bg[k + 1]; e_eps += x[k] * be[k + 1]; }
        const double psi = expit(e_psi), gam = expit(e_gam), eps = expit(e_eps);
        for (int t = 0; t < T; t++) {
            double a = 0.0;
            int ndet = 0;
            double *g = ga + (size_t)t * (Ko + 1);
            for (int k = 0; k <= Ko; k++) g[k] = 0.0;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                int c = d->C[o];
                if (!c) continue;
                const double *w = d->W + o * Ko;
                double nu = alpha[0];
                for (int k = 0; k < Ko; k++) nu += w[k] * alpha[k + 1];
                double r;
                if (c > 0) { a -= softplus(-nu); r = expit(-nu); ndet++; }
                else       { a -= softplus(nu);  r = -expit(nu); }
                g[0] += r;
                for (int k = 0; k < Ko; k++) g[k + 1] += r * w[k];
            }
            la[t] = a;                       /* log P(y_t | z_t = 1) */
            lb[t] = ndet * LOG_TINY_F32;     /* log P(y_t | z_t = 0) */
        }
        /* forward: pi_t = P(z_t = 1 | y_1..t-1), phi_t = P(z_t = 1 | y_1..t) */
        pi[0] = psi;
        for (int t = 0; t < T; t++) {
            const double A = log(pi[t]) + la[t], Bz = log1p(-pi[t]) + lb[t];
            const double l = logaddexp(A, Bz);
            ll += l;
            phi[t] = exp(A - l);
            pi[t + 1] = phi[t] * (1.0 - eps) + (1.0 - phi[t]) * gam;
        }
        /* backward: rho = P(z_t = 1 | y), xi(a, b) = P(z_t = a, z_t+1 = b | y) = rho_t+1(b) phi_t(a) P(b | a) / pi_t+1(b) */
        double rho = phi[T - 1], d_gam = 0.0, d_eps = 0.0;
        {
            const double *g = ga + (size_t)(T - 1) * (Ko + 1);
            for (int k = 0; k <= Ko; k++) gl[3 * B + k] += rho * g[k];
        }
        for (int t = T - 2; t >= 0; t--) {
            const double p1 = pi[t + 1], f = phi[t];
            const double w1 = p1 > 0.0 ? rho / p1 : 0.0, w0 = p1 < 1.0 ? (1.0 - rho) / (1.0 - p1) : 0.0;
            const double xi11 = w1 * f * (1.0 - eps), xi01 = w1 * (1.0 - f) * gam;
            const double xi10 = w0 * f * eps, xi00 = w0 * (1.0 - f) * (1.0 - gam);
            d_gam += xi01 * (1.0 - gam) - xi00 * gam;
            d_eps += xi10 * (1.0 - eps) - xi11 * eps;
            rho = xi11 + xi10;
            const double *g = ga + (size_t)t * (Ko + 1);
            for (int k = 0; k <= Ko; k++) gl[3 * B + k] += rho * g[k];
        }
        const double d_psi = rho - psi;
        gl[0] += d_psi; gl[B] += d_gam; gl[2 * B] += d_eps;
        for (int k = 0; k < Ks; k++) { gl[1 + k] += d_psi * x[k]; gl[B + 1 + k] += d_gam * x[k]; gl[2 * B + 1 + k] += d_eps * x[k]; }
    }
    free(la); free(lb); free(phi); free(pi); free(ga);
    for (int k = 0; k < D; k++) { /* the three site-side blocks take beta's prior, the detection block alpha's */
        const int is_b = k < 3 * B;
        const double loc = is_b ? d->loc_b : d->loc_a, sc = is_b ? d->scale_b : d->scale_a;
        const int laplace = is_b ? d->laplace_b : d->laplace_a;
        const double zz = (th[k] - loc) / sc;
        if (laplace) { gl[k] += -((zz > 0.0) - (zz < 0.0)) / sc; ll += -fabs(zz) - log(2.0 * sc); }
        else { gl[k] += -zz / sc; ll += -0.5 * zz * zz - log(sc) - HALF_LOG_2PI; }
    }
    for (int k = 0; k < D; k++) grad[k] = -gl[k];
    return -ll;
}

/* U(theta) = -log p(theta, y) and its gradient.  theta = [beta_0..beta_Ks, alpha_0..alpha_Ko]
 * (SURVEY.md Appendix A).  Includes the Normal prior normaliser.               */
static double potential_grad_occu(const orc_data *d, const double *th, double *grad)
{
    const int N = d->N, T = d->T, J = d->J, Ks = d->Ks, Ko = d->Ko, D = d->D;
    const double *beta = th, *alpha = th + Ks + 1;
    double gl[ORC_MAX_D];
    double ga[ORC_MAX_D];
    for (int k = 0; k < D; k++) gl[k] = 0.0;
    double ll = 0.0;
    for (int i = 0; i < N; i++) {
        const double *x = d->X + (size_t)i * Ks;
        double eta = beta[0];
        for (int k = 0; k < Ks; k++) eta += x[k] * beta[k + 1];  /* linear.py:59-66 */
        const double log_psi = -softplus(-eta), log_1mpsi = -softplus(eta), psi = expit(eta);
        double dl_deta = 0.0;
        for (int t = 0; t < T; t++) {
            double a = 0.0;
            int ndet = 0;
            for (int k = 0; k <= Ko; k++) ga[k] = 0.0;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                int c = d->C[o];
                if (!c) continue;
                const double *w = d->W + o * Ko;
                double nu = alpha[0];
                for (int k = 0; k < Ko; k++) nu += w[k] * alpha[k + 1];
                double r;
#ifdef ORC_BENCH_BUILD
                /* bench.py's cpu_baseline build (make native): the same two quantities from ONE exponential per visit --
                 * u = c nu: log sigma(u) = min(u, 0) - log1p(e), sigma(-u) = (u > 0 ? e : 1) / (1 + e), e = exp(-|u|) */
                const double u = c > 0 ? nu : -nu, e = exp(-fabs(u));
                a += (u < 0.0 ? u : 0.0) - log1p(e);
                r = (u > 0.0 ? e : 1.0) / (1.0 + e);
                if (c > 0) ndet++; else r = -r;
#else
                if (c > 0) { a -= softplus(-nu); r = expit(-nu); ndet++; }  /* y=1: log p,     d/dnu = 1-p */
                else       { a -= softplus(nu);  r = -expit(nu); }           /* y=0: log(1-p),  d/dnu = -p  */
#endif
                ga[0] += r;
                for (int k = 0; k < Ko; k++) ga[k + 1] += r * w[k];
            }
            /* sum over z in {1,0} (occu.py:208-210, enumerate=parallel) */
            const double A = log_psi + a;
            const double B = log_1mpsi + ndet * LOG_TINY_F32;
            const double l = logaddexp(A, B);
            const double q = exp(A - l);
            ll += l;
            dl_deta += q - psi;
            for (int k = 0; k <= Ko; k++) gl[Ks + 1 + k] += q * ga[k];
        }
        gl[0] += dl_deta;
        for (int k = 0; k < Ks; k++) gl[k + 1] += dl_deta * x[k];
    }
    /* priors: prior.expand([n+1]).to_event(1), linear.py:28 */
    for (int k = 0; k < D; k++) {
        ll += prior_coef(d, k, th[k], &gl[k]);
    }
    for (int k = 0; k < D; k++) grad[k] = -gl[k];
    return -ll;
}


/* ---- occu with a false-positive rate (biolith/models/occu.py:146-157, 229-241) ----
 * y_itj ~ Bernoulli(1 - (1 - z p)(1 - f_c)(1 - (1 - z) f_u)) with exactly one of f_c, f_u sampled
 * (the other is the constant 0), z enumerated, Bernoulli probs clamped to [tiny_f32, 1 - eps_f32] as
 * numpyro does.  Written branch by branch from the model text (no algebraic folding), so that it is
 * an independent check of the kernel's folded form.  theta = [beta, alpha, phi], rate = sigmoid(phi);
 * the potential carries the Beta(a, b) log-density and the log|d rate / d phi| Jacobian. */
static double potential_grad_occu_fp(const orc_data *d, const double *th, double *grad)
{
    const int N = d->N, T = d->T, J = d->J, Ks = d->Ks, Ko = d->Ko, D = d->D;
    const double *beta = th, *alpha = th + Ks + 1;
    const double phi = th[D - 1], f = expit(phi);
    const double f_c = d->fp_mode == 1 ? f : 0.0, f_u = d->fp_mode == 2 ? f : 0.0;
    const double TINY = 1.1754943508222875e-38, ONE_M_EPS = 1.0 - 1.1920928955078125e-07;
    double gl[ORC_MAX_D], ga1[ORC_MAX_D];
    for (int k = 0; k < D; k++) gl[k] = 0.0;
    double ll = 0.0;
    for (int i = 0; i < N; i++) {
        const double *x = d->X + (size_t)i * Ks;
        double eta = beta[0];
        for (int k = 0; k < Ks; k++) eta += x[k] * beta[k + 1];
        const double log_psi = -softplus(-eta), log_1mpsi = -softplus(eta), psi = expit(eta);
        double dl_deta = 0.0;
        for (int t = 0; t < T; t++) {
            double a1 = 0.0, a0 = 0.0, df1 = 0.0, df0 = 0.0; /* branch log-liks and their d/df */
            for (int k = 0; k <= Ko; k++) ga1[k] = 0.0;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                int c = d->C[o];
                if (!c) continue;
                const double *w = d->W + o * Ko;
                double nu = alpha[0];
                for (int k = 0; k < Ko; k++) nu += w[k] * alpha[k + 1];
                const double p = expit(nu);
                /* z = 1:  P1 = 1 - (1 - p)(1 - f_c);   z = 0:  P0 = 1 - (1 - f_c)(1 - f_u) */
                double P1 = 1.0 - (1.0 - p) * (1.0 - f_c), P0 = 1.0 - (1.0 - f_c) * (1.0 - f_u);
                int clip1 = 0, clip0 = 0;
                if (P1 < TINY) { P1 = TINY; clip1 = 1; } else if (P1 > ONE_M_EPS) { P1 = ONE_M_EPS; clip1 = 1; }
                if (P0 < TINY) { P0 = TINY; clip0 = 1; } else if (P0 > ONE_M_EPS) { P0 = ONE_M_EPS; clip0 = 1; }
                const double dl1 = c > 0 ? 1.0 / P1 : -1.0 / (1.0 - P1); /* d log Bern / dP */
                const double dl0 = c > 0 ? 1.0 / P0 : -1.0 / (1.0 - P0);
                a1 += c > 0 ? log(P1) : log1p(-P1);
                a0 += c > 0 ? log(P0) : log1p(-P0);
                if (!clip1) {
                    const double r = dl1 * (1.0 - f_c) * p * (1.0 - p); /* dP1/dnu */
                    ga1[0] += r;
                    for (int k = 0; k < Ko; k++) ga1[k + 1] += r * w[k];
                    if (d->fp_mode == 1) df1 += dl1 * (1.0 - p);        /* dP1/df_c */
                }
                if (!clip0) df0 += dl0;                                 /* dP0/df = 1 in both modes */
            }
            const double A = log_psi + a1, B = log_1mpsi + a0;
            const double l = logaddexp(A, B);
            const double q = exp(A - l);
            ll += l;
            dl_deta += q - psi;
            for (int k = 0; k <= Ko; k++) gl[Ks + 1 + k] += q * ga1[k];
            gl[D - 1] += (q * df1 + (1.0 - q) * df0) * f * (1.0 - f);
        }
        gl[0] += dl_deta;
        for (int k = 0; k < Ks; k++) gl[k + 1] += dl_deta * x[k];
    }
    for (int k = 0; k < D - 1; k++) {
        ll += prior_coef(d, k, th[k], &gl[k]);
    }
    /* Beta(a, b) log-density of the rate + Jacobian log f + log(1 - f) */
    if (!d->skip_phi_prior) {
        const double a = d->fp_a, b = d->fp_b;
        const double lf = -softplus(-phi), l1f = -softplus(phi);
        ll += (a - 1.0) * lf + (b - 1.0) * l1f - (lgamma(a) + lgamma(b) - lgamma(a + b)) + lf + l1f;
        gl[D - 1] += a * (1.0 - f) - b * f;
    }
    for (int k = 0; k < D; k++) grad[k] = -gl[k];
    return -ll;
}


/* ---- Royle-Nichols (biolith/models/occu_rn.py:123-222, utils/distributions.py:6-40) ----
 * abundance lambda_i = exp(beta0 + x_i beta); N_it ~ RightTruncatedPoisson(lambda_i, K) enumerated
 * (occu_rn.py:194-198: Categorical(logits = Poisson(lambda).log_prob(0..K)), i.e. renormalised);
 * y_itj ~ Bernoulli(1 - (1 - r_itj)^N), r = sigmoid(alpha0 + w alpha)  (occu_rn.py:209-222),
 * Bernoulli probs clamped to [tiny_f32, 1 - eps_f32] as numpyro does.  Returns U and dU/dtheta.
 * The gradient is obtained from posterior weights over N (exact differentiation of the logsumexp;
 * inside a clamped region the clamped factor has zero derivative, as under autodiff).            */
#define ORC_MAX_ABUNDANCE 255
/* The sums over n stop where nothing is left to add: every term is at most its prior part, lprior[n] - logz (the Bernoulli factors
 * are <= 1), and the prior part falls monotonically beyond the Poisson mode, so once n > lambda and lprior[n] - logz is more than
 * ORC_RN_CUT_NATS below the largest term found so far, every later term is too: what is dropped is below e^-45 = 3e-20 of the sum
 * per term, under the rounding of a double.  orc_data_set_rn_cut(d, 0) switches the stop off (tests: the two agree to 1e-13).
 * Per-visit quantities (log q, r) are formed once per (site, period), not once per n.  Same arithmetic per term as before. */
#define ORC_RN_CUT_NATS 45.0
static double potential_grad_rn(const orc_data *d, const double *th, double *grad)
{
    const int N = d->N, T = d->T, J = d->J, Ks = d->Ks, Ko = d->Ko, D = d->D, K = d->max_abundance;
    const double *beta = th, *alpha = th + Ks + 1;
    const double TINY = 1.1754943508222875e-38, ONE_M_EPS = 1.0 - 1.1920928955078125e-07;
    const int cut = !d->rn_no_cut;
    double gl[ORC_MAX_D];
    for (int k = 0; k < D; k++) gl[k] = 0.0;
    double ll = 0.0;
    double *term = (double *)malloc(sizeof(double) * (K + 1));
    double *lprior = (double *)malloc(sizeof(double) * (K + 1));
    double *lgam = (double *)malloc(sizeof(double) * (K + 1));
    double *nu = (double *)malloc(sizeof(double) * J), *dnu = (double *)malloc(sizeof(double) * J);
    double *lqv = (double *)malloc(sizeof(double) * J), *rv = (double *)malloc(sizeof(double) * J);
    for (int n = 0; n <= K; n++) lgam[n] = lgamma(n + 1.0);
    for (int i = 0; i < N; i++) {
        const double *x = d->X + (size_t)i * Ks;
        double eta = beta[0];
        for (int k = 0; k < Ks; k++) eta += x[k] * beta[k + 1];
        const double lam = exp(eta);
        /* renormalised truncated-Poisson prior and d log pi_n / d eta = n - E_pi[n] */
        double mz = -INFINITY;
        for (int n = 0; n <= K; n++) { lprior[n] = n * eta - lam - lgam[n]; if (lprior[n] > mz) mz = lprior[n]; }
        double sz = 0.0, en_prior = 0.0;
        for (int n = 0; n <= K; n++) { const double e = exp(lprior[n] - mz); sz += e; en_prior += n * e; }
        const double logz = mz + log(sz);
        en_prior /= sz;
        double dl_deta = 0.0;
        for (int t = 0; t < T; t++) {
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                const double *w = d->W + o * Ko;
                double v = alpha[0];
                for (int k = 0; k < Ko; k++) v += w[k] * alpha[k + 1];
                nu[j] = v; dnu[j] = 0.0;
                lqv[j] = -softplus(v);                             /* log(1 - r) */
                rv[j] = expit(v);
            }
            double m = -INFINITY;
            int nhi = K;                                           /* last n that is summed */
            for (int n = 0; n <= K; n++) {
                double tv = lprior[n] - logz;
                if (cut && (double)n > lam && tv < m - ORC_RN_CUT_NATS) { nhi = n - 1; break; }
                for (int j = 0; j < J; j++) {
                    const int c = d->C[((size_t)i * T + t) * J + j];
                    if (!c) continue;
                    double p = -expm1(n * lqv[j]);                 /* 1 - (1-r)^n */
                    if (p < TINY) p = TINY;
                    if (p > ONE_M_EPS) p = ONE_M_EPS;
                    tv += c > 0 ? log(p) : log1p(-p);
                }
                term[n] = tv;
                if (tv > m) m = tv;
            }
            double s = 0.0;
            for (int n = 0; n <= nhi; n++) s += exp(term[n] - m);
            const double l = m + log(s);
            ll += l;
            double en_post = 0.0;
            for (int n = 0; n <= nhi; n++) {
                const double wn = exp(term[n] - l);
                en_post += n * wn;
                for (int j = 0; j < J; j++) {
                    const int c = d->C[((size_t)i * T + t) * J + j];
                    if (!c || n == 0) continue;
                    const double r = rv[j];
                    const double sn = exp(n * lqv[j]);             /* (1-r)^n */
                    const double p = 1.0 - sn;
                    if (p < TINY || p > ONE_M_EPS) continue;       /* clamped: zero derivative */
                    /* d p / d nu = n r (1-r)^n */
                    dnu[j] += wn * (c > 0 ? n * r * sn / p : -n * r * sn / (1.0 - p));
                }
            }
            dl_deta += en_post - en_prior;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                const double *w = d->W + o * Ko;
                gl[Ks + 1] += dnu[j];
                for (int k = 0; k < Ko; k++) gl[Ks + 2 + k] += dnu[j] * w[k];
            }
        }
        gl[0] += dl_deta;
        for (int k = 0; k < Ks; k++) gl[k + 1] += dl_deta * x[k];
    }
    free(term); free(lprior); free(lgam); free(nu); free(dnu); free(lqv); free(rv);
    for (int k = 0; k < D; k++) {
        ll += prior_coef(d, k, th[k], &gl[k]);
    }
    for (int k = 0; k < D; k++) grad[k] = -gl[k];
    return -ll;
}
void orc_data_set_rn_cut(orc_data *d, int on) { d->rn_no_cut = on ? 0 : 1; }

/* ---- occu with site / observation random effects (biolith/models/occu.py:170-173, 191-196, 215-218) ----
 * site_re_sd ~ HalfNormal(s_site) and obs_re_sd ~ HalfNormal(s_obs) are sampled before the plates; per site
 * site_re_occ_i, site_re_det_i ~ Normal(0, site_re_sd) join the occupancy / detection predictors, per replicate
 * obs_re_itj ~ Normal(0, obs_re_sd) joins the detection predictor -- masked replicates keep their prior term.
 * theta = [beta, alpha, (log site_re_sd), (log obs_re_sd), (site_re_occ[N], site_re_det[N]), (obs_re[N][T][J])];
 * positive sites live on the log scale (numpyro: biject_to(positive) = exp), the potential carries the Jacobian. */
void orc_data_set_re(orc_data *d, int site_re, int obs_re, double scale_site, double scale_obs)
{
    d->model = 6; d->site_re = site_re != 0; d->obs_re = obs_re != 0;
    d->re_scale_site = scale_site; d->re_scale_obs = scale_obs;
    d->D = d->Ks + 1 + d->Ko + 1 + d->site_re + d->obs_re + (d->site_re ? 2 * d->N : 0) + (d->obs_re ? d->N * d->T * d->J : 0);
}

/* Random effects TOGETHER with a false-positive rate (occu.py:146-157 with :170-173, 191-196): theta gains phi = logit(rate) right
 * behind the regression coefficients, theta = [beta, alpha, phi, (log sds), (effects)]; one species. */
void orc_data_set_re_fp(orc_data *d, int fp_mode, double a, double b)
{
    d->fp_mode = fp_mode; d->fp_a = a; d->fp_b = b;
    d->D += 1;
}

static double potential_grad_re(const orc_data *d, const double *th, double *grad)
{
    const int N = d->N, T = d->T, J = d->J, Ks = d->Ks, Ko = d->Ko, D = d->D, G = Ks + 1 + Ko + 1;
    const double *beta = th, *alpha = th + Ks + 1;
    int o_phi_s = -1, o_phi_o = -1, o_u = -1, o_v = -1, o_e = -1, o_fp = -1, at = G;
    if (d->fp_mode) o_fp = at++;
    const double fpr = o_fp >= 0 ? expit(th[o_fp]) : 0.0;                  /* the false-positive rate */
    const double f_c = d->fp_mode == 1 ? fpr : 0.0, f_u = d->fp_mode == 2 ? fpr : 0.0;
    const double TINY = 1.1754943508222875e-38, ONE_M_EPS = 1.0 - 1.1920928955078125e-07;
    if (d->site_re) o_phi_s = at++;
    if (d->obs_re) o_phi_o = at++;
    if (d->site_re) { o_u = at; o_v = at + N; at += 2 * N; }
    if (d->obs_re) { o_e = at; at += N * T * J; }
    double *gl = grad; /* accumulate d log p / d theta in place, negate at the end */
    for (int k = 0; k < D; k++) gl[k] = 0.0;
    double ga[ORC_MAX_D];
    double ll = 0.0;
    for (int i = 0; i < N; i++) {
        const double *x = d->X + (size_t)i * Ks;
        double eta = beta[0];
        for (int k = 0; k < Ks; k++) eta += x[k] * beta[k + 1];
        if (d->site_re) eta += th[o_u + i];
        const double log_psi = -softplus(-eta), log_1mpsi = -softplus(eta), psi = expit(eta);
        double dl_deta = 0.0, dl_dv = 0.0;
        for (int t = 0; t < T; t++) {
            double a = 0.0, gsum;
            int ndet = 0;
            for (int k = 0; k <= Ko; k++) ga[k] = 0.0;
            /* first pass: the z = 1 branch and the per-visit d a / d nu (kept in ga and, per visit, below) */
            double rj[4096];
            if (J > 4096) return NAN;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                int c = d->C[o];
                rj[j] = 0.0;
                if (!c) continue;
                const double *w = d->W + o * Ko;
                double nu = alpha[0];
                for (int k = 0; k < Ko; k++) nu += w[k] * alpha[k + 1];
                if (d->site_re) nu += th[o_v + i];
                if (d->obs_re) nu += th[o_e + o];
                double r;
                if (c > 0) { a -= softplus(-nu); r = expit(-nu); ndet++; }
                else       { a -= softplus(nu);  r = -expit(nu); }
                rj[j] = r;
                ga[0] += r;
                for (int k = 0; k < Ko; k++) ga[k + 1] += r * w[k];
            }
            double A = log_psi + a;
            double B = log_1mpsi + ndet * LOG_TINY_F32;
            double df1 = 0.0, df0 = 0.0;
            if (d->fp_mode) {
                /* the visits again with the rate (potential_grad_occu_fp's arithmetic, numpyro's clamps included):
                 * z = 1: P1 = 1 - (1 - p)(1 - f_c);  z = 0: P0 = 1 - (1 - f_c)(1 - f_u) */
                double a1 = 0.0, a0 = 0.0;
                for (int k = 0; k <= Ko; k++) ga[k] = 0.0;
                for (int j = 0; j < J; j++) {
                    size_t o = ((size_t)i * T + t) * J + j;
                    int c = d->C[o];
                    rj[j] = 0.0;
                    if (!c) continue;
                    const double *w = d->W + o * Ko;
                    double nu = alpha[0];
                    for (int k = 0; k < Ko; k++) nu += w[k] * alpha[k + 1];
                    if (d->site_re) nu += th[o_v + i];
                    if (d->obs_re) nu += th[o_e + o];
                    const double p = expit(nu);
                    double P1 = 1.0 - (1.0 - p) * (1.0 - f_c), P0 = 1.0 - (1.0 - f_c) * (1.0 - f_u);
                    int clip1 = 0, clip0 = 0;
                    if (P1 < TINY) { P1 = TINY; clip1 = 1; } else if (P1 > ONE_M_EPS) { P1 = ONE_M_EPS; clip1 = 1; }
                    if (P0 < TINY) { P0 = TINY; clip0 = 1; } else if (P0 > ONE_M_EPS) { P0 = ONE_M_EPS; clip0 = 1; }
                    const double dl1 = c > 0 ? 1.0 / P1 : -1.0 / (1.0 - P1), dl0 = c > 0 ? 1.0 / P0 : -1.0 / (1.0 - P0);
                    a1 += c > 0 ? log(P1) : log1p(-P1);
                    a0 += c > 0 ? log(P0) : log1p(-P0);
                    if (!clip1) {
                        const double r = dl1 * (1.0 - f_c) * p * (1.0 - p);
                        rj[j] = r;
                        ga[0] += r;
                        for (int k = 0; k < Ko; k++) ga[k + 1] += r * w[k];
                        if (d->fp_mode == 1) df1 += dl1 * (1.0 - p);
                    }
                    if (!clip0) df0 += dl0;
                }
                A = log_psi + a1; B = log_1mpsi + a0;
            }
            const double l = logaddexp(A, B);
            const double q = exp(A - l);
            ll += l;
            dl_deta += q - psi;
            for (int k = 0; k <= Ko; k++) gl[Ks + 1 + k] += q * ga[k];
            gsum = q * ga[0];
            dl_dv += gsum;
            if (d->obs_re)
                for (int j = 0; j < J; j++) gl[o_e + ((size_t)i * T + t) * J + j] += q * rj[j];
            if (o_fp >= 0) gl[o_fp] += (q * df1 + (1.0 - q) * df0) * fpr * (1.0 - fpr);
        }
        gl[0] += dl_deta;
        for (int k = 0; k < Ks; k++) gl[k + 1] += dl_deta * x[k];
        if (d->site_re) { gl[o_u + i] += dl_deta; gl[o_v + i] += dl_dv; }
    }
    for (int k = 0; k < G; k++) {
        ll += prior_coef(d, k, th[k], &gl[k]);
    }
    if (o_fp >= 0) { /* Beta(a, b) log-density of the rate + the Jacobian log f + log(1 - f) */
        const double a = d->fp_a, b = d->fp_b, phi = th[o_fp];
        const double lf = -softplus(-phi), l1f = -softplus(phi);
        ll += (a - 1.0) * lf + (b - 1.0) * l1f - (lgamma(a) + lgamma(b) - lgamma(a + b)) + lf + l1f;
        gl[o_fp] += a * (1.0 - fpr) - b * fpr;
    }
    /* sd = exp(phi) ~ HalfNormal(s): log density + log |d sd / d phi|; its effects ~ Normal(0, sd) */
    for (int which = 0; which < 2; which++) {
        const int o_phi = which == 0 ? o_phi_s : o_phi_o;
        if (o_phi < 0) continue;
        const double s0 = which == 0 ? d->re_scale_site : d->re_scale_obs;
        const double phi = th[o_phi], sd = exp(phi), isd2 = exp(-2.0 * phi);
        if (!d->skip_phi_prior) { /* (several species: the shared sd's prior and Jacobian are counted once) */
            ll += 0.5 * log(2.0 / 3.14159265358979323846) - log(s0) - 0.5 * sd * sd / (s0 * s0) + phi;
            gl[o_phi] += -sd * sd / (s0 * s0) + 1.0;
        }
        const int first = which == 0 ? o_u : o_e, count = which == 0 ? 2 * N : N * T * J;
        for (int k = 0; k < count; k++) {
            const double e = th[first + k];
            ll += -0.5 * e * e * isd2 - phi - HALF_LOG_2PI;
            gl[first + k] += -e * isd2;
            gl[o_phi] += e * e * isd2 - 1.0;
        }
    }
    for (int k = 0; k < D; k++) grad[k] = -gl[k];
    return -ll;
}

/* ---- N-mixture with site / observation random effects (nmixture.py:139-141, 166-172, 199-214) ----
 * site_re_abu_i joins the abundance predictor, site_re_det_i and obs_re_itj the detection predictor; sds ~ HalfNormal on the log scale.
 * theta = [beta, alpha, (log site_re_sd), (log obs_re_sd), (site_re_abu[N], site_re_det[N]), (obs_re[N][T][J])]: the layout of the
 * occupancy model with random effects (potential_grad_re), whose treatment of the sds and the effects' priors is repeated here. */
void orc_data_set_nmix_re(orc_data *d, int site_re, int obs_re, double scale_site, double scale_obs)
{
    d->model = 9; d->site_re = site_re != 0; d->obs_re = obs_re != 0;
    d->re_scale_site = scale_site; d->re_scale_obs = scale_obs;
    d->D = d->Ks + 1 + d->Ko + 1 + d->site_re + d->obs_re + (d->site_re ? 2 * d->N : 0) + (d->obs_re ? d->N * d->T * d->J : 0);
}

static double potential_grad_nmix_re(const orc_data *d, const double *th, double *grad)
{
    const int N = d->N, T = d->T, J = d->J, Ks = d->Ks, Ko = d->Ko, D = d->D, K = d->max_abundance, G = Ks + 1 + Ko + 1;
    const double *beta = th, *alpha = th + Ks + 1;
    int o_phi_s = -1, o_phi_o = -1, o_u = -1, o_v = -1, o_e = -1, at = G;
    if (d->site_re) o_phi_s = at++;
    if (d->obs_re) o_phi_o = at++;
    if (d->site_re) { o_u = at; o_v = at + N; at += 2 * N; }
    if (d->obs_re) { o_e = at; at += N * T * J; }
    double *gl = grad;
    for (int k = 0; k < D; k++) gl[k] = 0.0;
    double ll = 0.0;
    double *term = (double *)malloc(sizeof(double) * (K + 1));
    double *pj = (double *)malloc(sizeof(double) * (J > 0 ? J : 1));
    for (int i = 0; i < N; i++) {
        const double *x = d->X + (size_t)i * Ks;
        double eta = beta[0];
        for (int k = 0; k < Ks; k++) eta += x[k] * beta[k + 1];
        if (d->site_re) eta += th[o_u + i];
        const double lam = exp(eta);
        double dl_deta = 0.0, dl_dv = 0.0;
        for (int t = 0; t < T; t++) {
            int min_count = 0;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                pj[j] = 0.0;
                if (!d->C[o]) continue;
                const double *w = d->W + o * Ko;
                double nu = alpha[0];
                for (int k = 0; k < Ko; k++) nu += w[k] * alpha[k + 1];
                if (d->site_re) nu += th[o_v + i];
                if (d->obs_re) nu += th[o_e + o];
                pj[j] = expit(nu);
                if ((int)d->Cnt[o] > min_count) min_count = (int)d->Cnt[o];
            }
            double mx = -INFINITY;
            for (int n = 0; n <= K; n++) {
                if (n < min_count) { term[n] = -INFINITY; continue; }
                double v = (n > 0 ? n * eta : 0.0) - lam - lgamma(n + 1.0);
                for (int j = 0; j < J; j++) {
                    size_t o = ((size_t)i * T + t) * J + j;
                    if (!d->C[o]) continue;
                    const double y = d->Cnt[o], p = pj[j];
                    v += lgamma(n + 1.0) - lgamma(y + 1.0) - lgamma(n - y + 1.0)
                         + (y > 0 ? y * log(p) : 0.0) + (n - y > 0 ? (n - y) * log1p(-p) : 0.0);
                }
                term[n] = v;
                if (v > mx) mx = v;
            }
            if (!(mx > -INFINITY)) { ll += -INFINITY; continue; }
            double S = 0.0, En = 0.0;
            for (int n = 0; n <= K; n++) { const double e = exp(term[n] - mx); S += e; En += n * e; }
            ll += mx + log(S);
            En /= S;
            dl_deta += En - lam;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                if (!d->C[o]) continue;
                const double *w = d->W + o * Ko;
                const double r = d->Cnt[o] - En * pj[j];
                gl[Ks + 1] += r;
                for (int k = 0; k < Ko; k++) gl[Ks + 2 + k] += r * w[k];
                dl_dv += r;
                if (d->obs_re) gl[o_e + o] += r;
            }
        }
        gl[0] += dl_deta;
        for (int k = 0; k < Ks; k++) gl[k + 1] += dl_deta * x[k];
        if (d->site_re) { gl[o_u + i] += dl_deta; gl[o_v + i] += dl_dv; }
    }
    free(term); free(pj);
    for (int k = 0; k < G; k++) ll += prior_coef(d, k, th[k], &gl[k]);
    for (int which = 0; which < 2; which++) { /* sd = exp(phi) ~ HalfNormal(s) + log-Jacobian; its effects ~ Normal(0, sd) */
        const int o_phi = which == 0 ? o_phi_s : o_phi_o;
        if (o_phi < 0) continue;
        const double s0 = which == 0 ? d->re_scale_site : d->re_scale_obs;
        const double phi = th[o_phi], sd = exp(phi), isd2 = exp(-2.0 * phi);
        ll += 0.5 * log(2.0 / 3.14159265358979323846) - log(s0) - 0.5 * sd * sd / (s0 * s0) + phi;
        gl[o_phi] += -sd * sd / (s0 * s0) + 1.0;
        const int first = which == 0 ? o_u : o_e, count = which == 0 ? 2 * N : N * T * J;
        for (int k = 0; k < count; k++) {
            const double e = th[first + k];
            ll += -0.5 * e * e * isd2 - phi - HALF_LOG_2PI;
            gl[first + k] += -e * isd2;
            gl[o_phi] += e * e * isd2 - 1.0;
        }
    }
    for (int k = 0; k < D; k++) grad[k] = -gl[k];
    return -ll;
}

/* ---- Royle-Nichols with site / observation random effects (occu_rn.py:151-154, 172-184, 199-212) ----
 * site_re_abu_i joins the abundance predictor, site_re_det_i and obs_re_itj the detection predictor; everything else -- the
 * renormalised truncated-Poisson weights, the clamped Bernoulli -- is potential_grad_rn's, the sds and the effects' priors are
 * potential_grad_re's.  theta = [beta, alpha, (log site_re_sd), (log obs_re_sd), (site_re_abu[N], site_re_det[N]), (obs_re[N][T][J])]. */
void orc_data_set_rn_re(orc_data *d, int site_re, int obs_re, double scale_site, double scale_obs)
{
    d->model = 10; d->site_re = site_re != 0; d->obs_re = obs_re != 0;
    d->re_scale_site = scale_site; d->re_scale_obs = scale_obs;
    d->D = d->Ks + 1 + d->Ko + 1 + d->site_re + d->obs_re + (d->site_re ? 2 * d->N : 0) + (d->obs_re ? d->N * d->T * d->J : 0);
}

/* ... TOGETHER with a false-positive rate (occu_rn.py:133-138, 214-221: y ~ Bernoulli(1 - (1 - p)(1 - f)), f ~ Beta(a, b)), with or
 * without the random effects: theta = [beta, alpha, phi = logit f, (log sds), (effects)]. */
void orc_data_set_rn_fp(orc_data *d, int site_re, int obs_re, double scale_site, double scale_obs, double a, double b)
{
    orc_data_set_rn_re(d, site_re, obs_re, scale_site, scale_obs);
    d->fp_mode = 1; d->fp_a = a; d->fp_b = b;
    d->D += 1;
}

static double potential_grad_rn_re(const orc_data *d, const double *th, double *grad)
{
    const int N = d->N, T = d->T, J = d->J, Ks = d->Ks, Ko = d->Ko, D = d->D, K = d->max_abundance, G = Ks + 1 + Ko + 1;
    const double *beta = th, *alpha = th + Ks + 1;
    const double TINY = 1.1754943508222875e-38, ONE_M_EPS = 1.0 - 1.1920928955078125e-07;
    int o_phi_s = -1, o_phi_o = -1, o_u = -1, o_v = -1, o_e = -1, o_fp = -1, at = G;
    if (d->fp_mode) o_fp = at++;
    if (d->site_re) o_phi_s = at++;
    if (d->obs_re) o_phi_o = at++;
    if (d->site_re) { o_u = at; o_v = at + N; at += 2 * N; }
    if (d->obs_re) { o_e = at; at += N * T * J; }
    const double fpr = o_fp >= 0 ? expit(th[o_fp]) : 0.0, gq = 1.0 - fpr; /* P(y = 1 | n) = 1 - q^n (1 - f) */
    double dfp = 0.0;                                                      /* d ll / d f */
    double *gl = grad;
    for (int k = 0; k < D; k++) gl[k] = 0.0;
    double ll = 0.0;
    double *term = (double *)malloc(sizeof(double) * (K + 1));
    double *lprior = (double *)malloc(sizeof(double) * (K + 1));
    double *nu = (double *)malloc(sizeof(double) * (J > 0 ? J : 1)), *dnu = (double *)malloc(sizeof(double) * (J > 0 ? J : 1));
    for (int i = 0; i < N; i++) {
        const double *x = d->X + (size_t)i * Ks;
        double eta = beta[0];
        for (int k = 0; k < Ks; k++) eta += x[k] * beta[k + 1];
        if (d->site_re) eta += th[o_u + i];
        const double lam = exp(eta);
        double mz = -INFINITY;
        for (int n = 0; n <= K; n++) { lprior[n] = n * eta - lam - lgamma(n + 1.0); if (lprior[n] > mz) mz = lprior[n]; }
        double sz = 0.0, en_prior = 0.0;
        for (int n = 0; n <= K; n++) { const double e = exp(lprior[n] - mz); sz += e; en_prior += n * e; }
        const double logz = mz + log(sz);
        en_prior /= sz;
        double dl_deta = 0.0, dl_dv = 0.0;
        for (int t = 0; t < T; t++) {
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                const double *w = d->W + o * Ko;
                double v = alpha[0];
                for (int k = 0; k < Ko; k++) v += w[k] * alpha[k + 1];
                if (d->site_re) v += th[o_v + i];
                if (d->obs_re) v += th[o_e + o];
                nu[j] = v; dnu[j] = 0.0;
            }
            double m = -INFINITY;
            for (int n = 0; n <= K; n++) {
                double tv = lprior[n] - logz;
                for (int j = 0; j < J; j++) {
                    const int c = d->C[((size_t)i * T + t) * J + j];
                    if (!c) continue;
                    const double lq = -softplus(nu[j]);
                    double p = o_fp >= 0 ? 1.0 - exp(n * lq) * gq : -expm1(n * lq); /* with a false-positive rate: 1 - q^n (1 - f) */
                    if (p < TINY) p = TINY;
                    if (p > ONE_M_EPS) p = ONE_M_EPS;
                    tv += c > 0 ? log(p) : log1p(-p);
                }
                term[n] = tv;
                if (tv > m) m = tv;
            }
            double s = 0.0;
            for (int n = 0; n <= K; n++) s += exp(term[n] - m);
            const double l = m + log(s);
            ll += l;
            double en_post = 0.0;
            for (int n = 0; n <= K; n++) {
                const double wn = exp(term[n] - l);
                en_post += n * wn;
                for (int j = 0; j < J; j++) {
                    const int c = d->C[((size_t)i * T + t) * J + j];
                    if (!c || (n == 0 && o_fp < 0)) continue;
                    const double r = expit(nu[j]), lq = -softplus(nu[j]);
                    const double sn = exp(n * lq);
                    if (o_fp >= 0) { /* p = 1 - sn (1 - f):  dp / dnu = n r sn (1 - f),  dp / df = sn */
                        const double p = 1.0 - sn * gq;
                        if (p < TINY || p > ONE_M_EPS) continue;
                        const double dlp = c > 0 ? 1.0 / p : -1.0 / (1.0 - p);
                        dnu[j] += wn * dlp * n * r * sn * gq;
                        dfp += wn * dlp * sn;
                        continue;
                    }
                    const double p = 1.0 - sn;
                    if (p < TINY || p > ONE_M_EPS) continue;
                    dnu[j] += wn * (c > 0 ? n * r * sn / p : -n * r * sn / (1.0 - p));
                }
            }
            dl_deta += en_post - en_prior;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                const double *w = d->W + o * Ko;
                gl[Ks + 1] += dnu[j];
                for (int k = 0; k < Ko; k++) gl[Ks + 2 + k] += dnu[j] * w[k];
                dl_dv += dnu[j];
                if (d->obs_re) gl[o_e + o] += dnu[j];
            }
        }
        gl[0] += dl_deta;
        for (int k = 0; k < Ks; k++) gl[k + 1] += dl_deta * x[k];
        if (d->site_re) { gl[o_u + i] += dl_deta; gl[o_v + i] += dl_dv; }
    }
    free(term); free(lprior); free(nu); free(dnu);
    for (int k = 0; k < G; k++) ll += prior_coef(d, k, th[k], &gl[k]);
    if (o_fp >= 0) { /* chain rule through f = expit(phi); Beta(a, b) log-density of the rate + the Jacobian log f + log(1 - f) */
        const double a = d->fp_a, b = d->fp_b, phi = th[o_fp];
        const double lf = -softplus(-phi), l1f = -softplus(phi);
        gl[o_fp] += dfp * fpr * gq;
        ll += (a - 1.0) * lf + (b - 1.0) * l1f - (lgamma(a) + lgamma(b) - lgamma(a + b)) + lf + l1f;
        gl[o_fp] += a * (1.0 - fpr) - b * fpr;
    }
    for (int which = 0; which < 2; which++) { /* sd = exp(phi) ~ HalfNormal(s) + log-Jacobian; its effects ~ Normal(0, sd) */
        const int o_phi = which == 0 ? o_phi_s : o_phi_o;
        if (o_phi < 0) continue;
        const double s0 = which == 0 ? d->re_scale_site : d->re_scale_obs;
        const double phi = th[o_phi], sd = exp(phi), isd2 = exp(-2.0 * phi);
        ll += 0.5 * log(2.0 / 3.14159265358979323846) - log(s0) - 0.5 * sd * sd / (s0 * s0) + phi;
        gl[o_phi] += -sd * sd / (s0 * s0) + 1.0;
        const int first = which == 0 ? o_u : o_e, count = which == 0 ? 2 * N : N * T * J;
        for (int k = 0; k < count; k++) {
            const double e = th[first + k];
            ll += -0.5 * e * e * isd2 - phi - HALF_LOG_2PI;
            gl[first + k] += -e * isd2;
            gl[o_phi] += e * e * isd2 - 1.0;
        }
    }
    for (int k = 0; k < D; k++) grad[k] = -gl[k];
    return -ll;
}

/* ---- occu_cop with site / observation random effects (occu_cop.py:183-186, 204-210, 229-243) ----
 * site_re_occ_i joins the occupancy predictor, site_re_det_i and obs_re_itj the log detection rate; no false-positive rates here.
 * theta = [beta, alpha, (log site_re_sd), (log obs_re_sd), (site_re_occ[N], site_re_det[N]), (obs_re[N][T][J])]: the layout, the
 * sds and the effects' priors of the occupancy model with random effects (potential_grad_re). */
void orc_data_set_cop_re(orc_data *d, int site_re, int obs_re, double scale_site, double scale_obs)
{
    /* (a false-positive rate set by orc_data_set_cop stays: theta = [beta, alpha, phi = log rate, (log sds), (effects)]) */
    d->model = 12; d->site_re = site_re != 0; d->obs_re = obs_re != 0;
    d->re_scale_site = scale_site; d->re_scale_obs = scale_obs;
    d->D = d->Ks + 1 + d->Ko + 1 + (d->fp_mode ? 1 : 0) + d->site_re + d->obs_re + (d->site_re ? 2 * d->N : 0) + (d->obs_re ? d->N * d->T * d->J : 0);
}

static double potential_grad_cop_re(const orc_data *d, const double *th, double *grad)
{
    const int N = d->N, T = d->T, J = d->J, Ks = d->Ks, Ko = d->Ko, D = d->D, G = Ks + 1 + Ko + 1;
    const double *beta = th, *alpha = th + Ks + 1;
    int o_phi_s = -1, o_phi_o = -1, o_u = -1, o_v = -1, o_e = -1, o_fp = -1, at = G;
    if (d->fp_mode) o_fp = at++;
    if (d->site_re) o_phi_s = at++;
    if (d->obs_re) o_phi_o = at++;
    if (d->site_re) { o_u = at; o_v = at + N; at += 2 * N; }
    if (d->obs_re) { o_e = at; at += N * T * J; }
    double *gl = grad;
    for (int k = 0; k < D; k++) gl[k] = 0.0;
    double ll = 0.0;
    const double f = o_fp >= 0 ? exp(th[o_fp]) : 0.0;
    const double f_c = d->fp_mode == 1 ? f : 0.0, f_u = d->fp_mode == 2 ? f : 0.0;
    double *rj = (double *)malloc(sizeof(double) * (J > 0 ? J : 1));
    for (int i = 0; i < N; i++) {
        const double *x = d->X + (size_t)i * Ks;
        double eta = beta[0];
        for (int k = 0; k < Ks; k++) eta += x[k] * beta[k + 1];
        if (d->site_re) eta += th[o_u + i];
        const double log_psi = -softplus(-eta), log_1mpsi = -softplus(eta), psi = expit(eta);
        double dl_deta = 0.0, dl_dv = 0.0;
        for (int t = 0; t < T; t++) {
            double a1 = 0.0, a0 = 0.0, df1 = 0.0, df0 = 0.0;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                rj[j] = 0.0;
                if (!d->C[o]) continue;
                const double *w = d->W + o * Ko;
                double nu = alpha[0];
                for (int k = 0; k < Ko; k++) nu += w[k] * alpha[k + 1];
                if (d->site_re) nu += th[o_v + i];
                if (d->obs_re) nu += th[o_e + o];
                const double lam = exp(nu), dur = d->Dur[o], y = d->Cnt[o];
                double dl1, dl0;
                a1 += poisson_logpmf(y, dur * (lam + f_c), &dl1);
                a0 += poisson_logpmf(y, dur * (f_u + f_c), &dl0);
                rj[j] = dl1 * dur * lam; /* d/dnu of the z = 1 branch */
                if (d->fp_mode == 1) df1 += dl1 * dur;
                if (d->fp_mode) df0 += dl0 * dur;
            }
            const double A = log_psi + a1, B = log_1mpsi + a0;
            const double l = logaddexp(A, B);
            const double q = exp(A - l);
            ll += l;
            dl_deta += q - psi;
            if (o_fp >= 0) gl[o_fp] += (q * df1 + (1.0 - q) * df0) * f;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                if (!d->C[o]) continue;
                const double *w = d->W + o * Ko;
                const double r = q * rj[j];
                gl[Ks + 1] += r;
                for (int k = 0; k < Ko; k++) gl[Ks + 2 + k] += r * w[k];
                dl_dv += r;
                if (d->obs_re) gl[o_e + o] += r;
            }
        }
        gl[0] += dl_deta;
        for (int k = 0; k < Ks; k++) gl[k + 1] += dl_deta * x[k];
        if (d->site_re) { gl[o_u + i] += dl_deta; gl[o_v + i] += dl_dv; }
    }
    free(rj);
    for (int k = 0; k < G; k++) ll += prior_coef(d, k, th[k], &gl[k]);
    if (o_fp >= 0) { /* Exponential(rate) log-density of f plus the Jacobian log f = phi */
        ll += log(d->fp_a) - d->fp_a * f + th[o_fp];
        gl[o_fp] += -d->fp_a * f + 1.0;
    }
    for (int which = 0; which < 2; which++) { /* sd = exp(phi) ~ HalfNormal(s) + log-Jacobian; its effects ~ Normal(0, sd) */
        const int o_phi = which == 0 ? o_phi_s : o_phi_o;
        if (o_phi < 0) continue;
        const double s0 = which == 0 ? d->re_scale_site : d->re_scale_obs;
        const double phi = th[o_phi], sd = exp(phi), isd2 = exp(-2.0 * phi);
        ll += 0.5 * log(2.0 / 3.14159265358979323846) - log(s0) - 0.5 * sd * sd / (s0 * s0) + phi;
        gl[o_phi] += -sd * sd / (s0 * s0) + 1.0;
        const int first = which == 0 ? o_u : o_e, count = which == 0 ? 2 * N : N * T * J;
        for (int k = 0; k < count; k++) {
            const double e = th[first + k];
            ll += -0.5 * e * e * isd2 - phi - HALF_LOG_2PI;
            gl[first + k] += -e * isd2;
            gl[o_phi] += e * e * isd2 - 1.0;
        }
    }
    for (int k = 0; k < D; k++) grad[k] = -gl[k];
    return -ll;
}

/* ---- continuous-score occupancy model (biolith/models/occu_cs.py:120-232) ----
 * s_itj ~ Normal(mu_f, sigma_f), f_itj ~ Bernoulli(z_it p_itj), z_it ~ Bernoulli(psi_i); z and every f enumerated (the f of
 * different replicates are independent given z, so the sum over them factorises per replicate).  Bernoulli probabilities of f
 * are clamped to [tiny_f32, 1 - eps_f32] as numpyro does (z = 0: P(f = 1) = tiny).  mu1 ~ TruncatedDistribution(Normal, low=mu0).
 * theta = [beta, alpha, mu0, x1, log sigma0, log sigma1] with mu1 = mu0 + exp(x1) (biject_to(greater_than(mu0))), positive
 * sites on the log scale; the potential carries the Jacobians and the truncation's normaliser. */
void orc_data_set_cs(orc_data *d, const double *scores /*[N][T][J], NaN = missing*/, const double *prior_mu /*[4]*/, const double *prior_sigma /*[4]*/)
{
    const size_t n = (size_t)d->N * d->T * d->J;
    d->model = 7;
    d->D = d->Ks + 1 + d->Ko + 1 + 4;
    free(d->Cnt);
    d->Cnt = (double *)malloc(sizeof(double) * n);
    for (size_t o = 0; o < n; o++) d->Cnt[o] = isfinite(scores[o]) ? scores[o] : 0.0;
    memcpy(d->cs_mu, prior_mu, sizeof d->cs_mu);
    memcpy(d->cs_sg, prior_sigma, sizeof d->cs_sg);
}

static double potential_grad_cs(const orc_data *d, const double *th, double *grad)
{
    const int N = d->N, T = d->T, J = d->J, Ks = d->Ks, Ko = d->Ko, D = d->D, G0 = Ks + Ko + 2;
    const double *beta = th, *alpha = th + Ks + 1;
    const double mu0 = th[G0], x1 = th[G0 + 1], ls0 = th[G0 + 2], ls1 = th[G0 + 3];
    const double ex1 = exp(x1), mu1 = mu0 + ex1, sg0 = exp(ls0), sg1 = exp(ls1);
    const double TINY = 1.1754943508222875e-38, EPS = 1.1920928955078125e-07;
    const double l_f1_z0 = log(TINY), l_f0_z0 = log1p(-TINY);
    double gl[ORC_MAX_D];
    for (int k = 0; k < D; k++) gl[k] = 0.0;
    double ll = 0.0, g_mu0 = 0.0, g_mu1 = 0.0, g_ls0 = 0.0, g_ls1 = 0.0;
    for (int i = 0; i < N; i++) {
        const double *x = d->X + (size_t)i * Ks;
        double eta = beta[0];
        for (int k = 0; k < Ks; k++) eta += x[k] * beta[k + 1];
        const double log_psi = -softplus(-eta), log_1mpsi = -softplus(eta), psi = expit(eta);
        double dl_deta = 0.0;
        for (int t = 0; t < T; t++) {
            /* per replicate and per branch of z: L = log sum_f P(f | z) Normal(s; mu_f, sigma_f), and its derivatives */
            double a1 = 0.0, a0 = 0.0, ga[ORC_MAX_D];
            double d1[4] = {0, 0, 0, 0}, d0[4] = {0, 0, 0, 0}; /* d / d (mu0, mu1, log sigma0, log sigma1) of a1, a0 */
            for (int k = 0; k <= Ko; k++) ga[k] = 0.0;
            for (int j = 0; j < J; j++) {
                size_t o = ((size_t)i * T + t) * J + j;
                if (!d->C[o]) continue;
                const double *w = d->W + o * Ko;
                double nu = alpha[0];
                for (int k = 0; k < Ko; k++) nu += w[k] * alpha[k + 1];
                const double sc = d->Cnt[o];
                const double e0 = (sc - mu0) / sg0, e1 = (sc - mu1) / sg1;
                const double lphi0 = -0.5 * e0 * e0 - ls0 - HALF_LOG_2PI, lphi1 = -0.5 * e1 * e1 - ls1 - HALF_LOG_2PI;
                const double p = expit(nu);
                const int inside = p > TINY && p < 1.0 - EPS;
                const double pc = p < TINY ? TINY : (p > 1.0 - EPS ? 1.0 - EPS : p);
                /* z = 1 */
                const double t1 = log(pc) + lphi1, t0 = log1p(-pc) + lphi0, L1 = logaddexp(t0, t1), w1 = exp(t1 - L1), w0 = exp(t0 - L1);
                a1 += L1;
                const double dnu = inside ? w1 * (1.0 - p) - w0 * p : 0.0;
                ga[0] += dnu;
                for (int k = 0; k < Ko; k++) ga[k + 1] += dnu * w[k];
                d1[0] += w0 * e0 / sg0; d1[1] += w1 * e1 / sg1; d1[2] += w0 * (e0 * e0 - 1.0); d1[3] += w1 * (e1 * e1 - 1.0);
                /* z = 0: P(f = 1) = 0 clamped to tiny */
                const double u1 = l_f1_z0 + lphi1, u0 = l_f0_z0 + lphi0, L0 = logaddexp(u0, u1), v1 = exp(u1 - L0), v0 = exp(u0 - L0);
                a0 += L0;
                d0[0] += v0 * e0 / sg0; d0[1] += v1 * e1 / sg1; d0[2] += v0 * (e0 * e0 - 1.0); d0[3] += v1 * (e1 * e1 - 1.0);
            }
            const double A = log_psi + a1, B = log_1mpsi + a0;
            const double l = logaddexp(A, B), q = exp(A - l);
            ll += l;
            dl_deta += q - psi;
            for (int k = 0; k <= Ko; k++) gl[Ks + 1 + k] += q * ga[k];
            g_mu0 += q * d1[0] + (1.0 - q) * d0[0]; g_mu1 += q * d1[1] + (1.0 - q) * d0[1];
            g_ls0 += q * d1[2] + (1.0 - q) * d0[2]; g_ls1 += q * d1[3] + (1.0 - q) * d0[3];
        }
        gl[0] += dl_deta;
        for (int k = 0; k < Ks; k++) gl[k + 1] += dl_deta * x[k];
    }
    for (int k = 0; k < G0; k++) ll += prior_coef(d, k, th[k], &gl[k]);
    /* mu0 ~ Normal(l0, s0); mu1 ~ Normal(l1, s1) truncated below at mu0, in x1 = log(mu1 - mu0) */
    {
        const double l0 = d->cs_mu[0], s0 = d->cs_mu[1], l1 = d->cs_mu[2], s1 = d->cs_mu[3];
        const double z0 = (mu0 - l0) / s0, z1 = (mu1 - l1) / s1, zl = (mu0 - l1) / s1;
        const double surv = 0.5 * erfc(zl / 1.41421356237309504880);
        ll += -0.5 * z0 * z0 - log(s0) - HALF_LOG_2PI;
        ll += -0.5 * z1 * z1 - log(s1) - HALF_LOG_2PI - log(surv) + x1;
        const double hazard = exp(-0.5 * zl * zl - HALF_LOG_2PI) / surv / s1; /* d/d mu0 of -log surv */
        const double dmu1 = g_mu1 - z1 / s1;
        gl[G0] += g_mu0 - z0 / s0 + dmu1 + hazard;
        gl[G0 + 1] += dmu1 * ex1 + 1.0;
    }
    /* sigma_f ~ Gamma(a, b) in log sigma_f */
    for (int f = 0; f < 2; f++) {
        const double a = d->cs_sg[2 * f], b = d->cs_sg[2 * f + 1], sg = f ? sg1 : sg0, ls = f ? ls1 : ls0;
        ll += a * log(b) - lgamma(a) + (a - 1.0) * ls - b * sg + ls;
        gl[G0 + 2 + f] += (f ? g_ls1 : g_ls0) + a - b * sg;
    }
    for (int k = 0; k < D; k++) grad[k] = -gl[k];
    return -ll;
}

/* Several species under one chain (biolith/models/occu.py:182-186: beta, alpha inside the species plate; 146-157: the
 * false-positive rate outside it, i.e. shared): the joint potential is the sum of the species' potentials, the shared rate's
 * prior counted once and its gradient summed.  obs_all [S][N][T][J]; cov_nan [N][T][J] != 0 where a covariate is NaN (the mask
 * of occu.py:136-139, which orc_data_create folded into species 0's C). */
void orc_data_set_species(orc_data *d, int S, const double *obs_all, const unsigned char *cov_nan)
{
    const size_t n = (size_t)d->N * d->T * d->J;
    const int Dp = d->Ks + d->Ko + 2;
    free(d->C);
    d->C = (signed char *)malloc((size_t)S * n);
    for (int s = 0; s < S; s++)
        for (size_t o = 0; o < n; o++) {
            const double y = obs_all[(size_t)s * n + o];
            d->C[(size_t)s * n + o] = (cov_nan[o] || !isfinite(y)) ? 0 : (y != 0.0 ? 1 : -1);
        }
    d->S = S;
    d->D = S * Dp + (d->model == 2 ? 1 : 0);
    if (d->model == 6) /* random effects inside the species plate (occu.py:182-196), their sds outside it (occu.py:170-173) */
        d->D = S * Dp + d->site_re + d->obs_re + S * ((d->site_re ? 2 * d->N : 0) + (d->obs_re ? d->N * d->T * d->J : 0));
}
/* Several species with random effects: theta = [species 0: beta, alpha | species 1: ... | (log site_re_sd) | (log obs_re_sd) |
 * site_re_occ [S][N] | site_re_det [S][N] | obs_re [S][N][T][J]].  The joint potential is the sum of the species' potentials
 * with the two sds -- and their HalfNormal priors and Jacobians -- shared. */
static double potential_grad_re_species(const orc_data *d, const double *th, double *grad)
{
    const int S = d->S, N = d->N, V = d->T * d->J, Dp = d->Ks + d->Ko + 2, nsd = d->site_re + d->obs_re;
    const size_t n = (size_t)N * V;
    const int D1 = Dp + nsd + (d->site_re ? 2 * N : 0) + (d->obs_re ? N * V : 0);
    const int o_sd = S * Dp, o_u = o_sd + nsd, o_v = o_u + (d->site_re ? S * N : 0), o_e = o_v + (d->site_re ? S * N : 0);
    double *ths = (double *)malloc(sizeof(double) * D1 * 2), *gs = ths + D1;
    double U = 0.0;
    for (int k = 0; k < d->D; k++) grad[k] = 0.0;
    for (int s = 0; s < S; s++) {
        orc_data ds = *d;
        ds.S = 1; ds.C = d->C + (size_t)s * n; ds.D = D1; ds.skip_phi_prior = s > 0;
        int at = 0;
        memcpy(ths, th + (size_t)s * Dp, sizeof(double) * Dp); at = Dp;
        memcpy(ths + at, th + o_sd, sizeof(double) * nsd); at += nsd;
        if (d->site_re) {
            memcpy(ths + at, th + o_u + (size_t)s * N, sizeof(double) * N); at += N;
            memcpy(ths + at, th + o_v + (size_t)s * N, sizeof(double) * N); at += N;
        }
        if (d->obs_re) memcpy(ths + at, th + o_e + (size_t)s * n, sizeof(double) * n);
        U += potential_grad_re(&ds, ths, gs);
        at = 0;
        memcpy(grad + (size_t)s * Dp, gs, sizeof(double) * Dp); at = Dp;
        for (int k = 0; k < nsd; k++) grad[o_sd + k] += gs[at + k];
        at += nsd;
        if (d->site_re) {
            memcpy(grad + o_u + (size_t)s * N, gs + at, sizeof(double) * N); at += N;
            memcpy(grad + o_v + (size_t)s * N, gs + at, sizeof(double) * N); at += N;
        }
        if (d->obs_re) memcpy(grad + o_e + (size_t)s * n, gs + at, sizeof(double) * n);
    }
    free(ths);
    return U;
}
static double potential_grad_species(const orc_data *d, const double *th, double *grad)
{
    const int S = d->S, Dp = d->Ks + d->Ko + 2, fp = d->model == 2;
    const size_t n = (size_t)d->N * d->T * d->J;
    double U = 0.0, gphi = 0.0;
    for (int s = 0; s < S; s++) {
        orc_data ds = *d;
        ds.S = 1; ds.C = d->C + (size_t)s * n; ds.D = Dp + fp; ds.skip_phi_prior = s > 0;
        double ths[ORC_MAX_D], gs[ORC_MAX_D];
        memcpy(ths, th + (size_t)s * Dp, sizeof(double) * Dp);
        if (fp) ths[Dp] = th[S * Dp];
        U += fp ? potential_grad_occu_fp(&ds, ths, gs) : potential_grad_occu(&ds, ths, gs);
        memcpy(grad + (size_t)s * Dp, gs, sizeof(double) * Dp);
        if (fp) gphi += gs[Dp];
    }
    if (fp) grad[S * Dp] = gphi;
    return U;
}

/* U(theta) = -log p(theta, y) and its gradient for the dataset's model. */
double orc_potential_grad(const orc_data *d, const double *th, double *grad)
{
    if (d->S > 1) return d->model == 6 ? potential_grad_re_species(d, th, grad) : potential_grad_species(d, th, grad);
    if (d->model == 12) return potential_grad_cop_re(d, th, grad);
    if (d->model == 10) return potential_grad_rn_re(d, th, grad);
    if (d->model == 9) return potential_grad_nmix_re(d, th, grad);
    if (d->model == 8) return potential_grad_dyn(d, th, grad);
    if (d->model == 7) return potential_grad_cs(d, th, grad);
    if (d->model == 6) return potential_grad_re(d, th, grad);
    if (d->model == 4) return potential_grad_nmix(d, th, grad);
    if (d->model == 3) return potential_grad_cop(d, th, grad);
    if (d->model == 2) return potential_grad_occu_fp(d, th, grad);
    return d->model == 1 ? potential_grad_rn(d, th, grad) : potential_grad_occu(d, th, grad);
}

/* ------------------------------------------------------------- sampler -- */
/* numpyro.infer.hmc_util.build_adaptation_schedule. Returns #windows. */
int orc_adaptation_schedule(int num_steps, int *starts, int *ends /* capacity >= 32 */)
{
    int n = 0;
    if (num_steps < 20) { starts[0] = 0; ends[0] = num_steps - 1; return 1; }
    int start_buffer = 75, end_buffer = 50, init_window = 25;
    if (start_buffer + end_buffer + init_window > num_steps) {
        start_buffer = (int)(0.15 * num_steps);
        end_buffer = (int)(0.1 * num_steps);
        init_window = num_steps - start_buffer - end_buffer;
    }
    starts[n] = 0; ends[n] = start_buffer - 1; n++;
    const int end_window_start = num_steps - end_buffer;
    int next_size = init_window, next_start = start_buffer;
    while (next_start < end_window_start) {
        int cur_start = next_start, cur_size = next_size;
        if (3 * cur_size <= end_window_start - cur_start) next_size = 2 * cur_size;
        else cur_size = end_window_start - cur_start;
        next_start = cur_start + cur_size;
        starts[n] = cur_start; ends[n] = next_start - 1; n++;
    }
    starts[n] = end_window_start; ends[n] = num_steps - 1; n++;
    return n;
}

/* Vectors of length D live in one heap block per run (D reaches thousands with random effects). */
typedef struct { double *z, *r, *g; } edge_t;

typedef struct {
    edge_t left, right;
    double *zp, *gp, Up, Ep;  /* proposal */
    double weight, *rsum, sum_accept;
    int depth, turning, diverging, nprop;
} tree_t;

static double *carve(double **pool, int D) { double *v = *pool; *pool += D; return v; }
static void edge_alloc(edge_t *e, double **pool, int D) { e->z = carve(pool, D); e->r = carve(pool, D); e->g = carve(pool, D); }
static void edge_copy(edge_t *dst, const edge_t *src, int D)
{
    memcpy(dst->z, src->z, sizeof(double) * D); memcpy(dst->r, src->r, sizeof(double) * D); memcpy(dst->g, src->g, sizeof(double) * D);
}
static void tree_alloc(tree_t *t, double **pool, int D)
{
    edge_alloc(&t->left, pool, D); edge_alloc(&t->right, pool, D);
    t->zp = carve(pool, D); t->gp = carve(pool, D); t->rsum = carve(pool, D);
}
static void tree_reset(tree_t *t)
{
    t->Up = t->Ep = t->weight = t->sum_accept = 0.0;
    t->depth = t->turning = t->diverging = t->nprop = 0;
}

static double kinetic(const double *minv, const double *r, int D)
{
    double k = 0.0;
    for (int i = 0; i < D; i++) k += minv[i] * r[i] * r[i];
    return 0.5 * k;
}

/* hmc_util._is_turning, diagonal mass */
static int is_turning(const double *minv, const double *rl, const double *rr, const double *rsum, int D)
{
    double dl = 0.0, dr = 0.0;
    for (int i = 0; i < D; i++) {
        double rho = rsum[i] - 0.5 * (rl[i] + rr[i]);
        dl += minv[i] * rl[i] * rho;
        dr += minv[i] * rr[i] * rho;
    }
    return (dl <= 0.0) || (dr <= 0.0);
}

static inline int popcount_u(unsigned n) { int c = 0; while (n) { c += n & 1u; n >>= 1; } return c; }
static inline int trailing_ones(unsigned n) { int c = 0; while (n & 1u) { c++; n >>= 1; } return c; }

typedef struct {
    /* dual averaging (t0=10, kappa=0.75, gamma=0.05) */
    double x_t, x_avg, g_avg, prox_center; int t;
    /* Welford */
    double *mean, *m2; int n;
    int window_idx;
} adapt_t;

static void ss_init(adapt_t *a, double prox) { a->x_t = 0; a->x_avg = 0; a->g_avg = 0; a->t = 0; a->prox_center = prox; }

/* One chain.  Outputs are post-warmup only (thinning 1). Returns 0. */
int orc_nuts_run(const orc_data *d, int num_warmup, int num_samples, uint64_t seed, int chain_id,
                 int max_tree_depth, double target_accept,
                 const double *init_theta /* NULL -> Uniform(-2,2) from the chain's streams */,
                 double *draws /*[S][D]*/, int *num_steps /*[S]*/, double *accept_prob /*[S]*/,
                 unsigned char *diverging /*[S]*/, double *potential /*[S]*/,
                 double *step_size_out, double *inv_mass_out /*[D]*/,
                 long long *n_leapfrog_total /* [2]: warmup, sampling */,
                 /* optional trace of EVERY transition incl. warmup (NULL ok): */
                 double *trace_theta /*[W+S][D]*/, int *trace_steps /*[W+S]*/, double *trace_eps /*[W+S]*/)
{
    const int D = d->D;
    if (D > ORC_BIG_D || max_tree_depth > ORC_MAX_DEPTH) return -1;
    if (d->model != 6 && d->model != 9 && d->model != 10 && d->model != 12 && D > ORC_MAX_D) return -1; /* the small models keep gradients in stack arrays of ORC_MAX_D */
    /* one stream per coordinate, then the two scalar streams: ORC_SCALAR_STREAM / ORC_DIR_STREAM of ORC_NSTREAM while
     * D <= ORC_MAX_D (the device layout of the small models); streams D (scalar) and D + 1 (direction) of D + 2 beyond */
    const int big = D > ORC_MAX_D;
    const int i_scalar = big ? D : ORC_SCALAR_STREAM, i_dir = big ? D + 1 : ORC_DIR_STREAM;
    const int nstreams = big ? D + 2 : ORC_NSTREAM;
    uint32_t (*st)[4] = (uint32_t (*)[4])malloc(sizeof(uint32_t) * 4 * (size_t)nstreams);
    orc_rng_streams_strided(seed, chain_id, nstreams > ORC_NSTREAM ? nstreams : ORC_NSTREAM, nstreams, &st[0][0]);
    uint32_t *sc = st[i_scalar], *sdir = st[i_dir];

    double *block = (double *)calloc((size_t)D * (40 + 2 * ORC_MAX_DEPTH), sizeof(double)), *pool = block;
    double *theta = carve(&pool, D), *grad = carve(&pool, D), U;
    for (int i = 0; i < D; i++) {
        double u = rng_uniform(st[i]); /* always consumed, keeps streams aligned with the device */
        theta[i] = init_theta ? init_theta[i] : 4.0 * u - 2.0; /* init_to_uniform(radius=2), fit.py:93 */
    }
    U = orc_potential_grad(d, theta, grad);
    long long nleap[2] = {0, 0};

    double *minv = carve(&pool, D);
    for (int i = 0; i < D; i++) minv[i] = 1.0;
    double step_size = 1.0;
    adapt_t ad; memset(&ad, 0, sizeof ad);
    ad.mean = carve(&pool, D); ad.m2 = carve(&pool, D);
    ss_init(&ad, log(10.0 * step_size));
    int wstart[40], wend[40];
    const int nwin = num_warmup > 0 ? orc_adaptation_schedule(num_warmup, wstart, wend) : 0;

    double *r_ckpt[ORC_MAX_DEPTH], *rsum_ckpt[ORC_MAX_DEPTH];
    for (int k = 0; k < ORC_MAX_DEPTH; k++) { r_ckpt[k] = carve(&pool, D); rsum_ckpt[k] = carve(&pool, D); }
    double *r0 = carve(&pool, D), *rh = carve(&pool, D), *srs = carve(&pool, D);
    tree_t tr, sub;
    tree_alloc(&tr, &pool, D); tree_alloc(&sub, &pool, D);
    edge_t cur, nw;
    edge_alloc(&cur, &pool, D); edge_alloc(&nw, &pool, D);
    const int total = num_warmup + num_samples;

    for (int it = 0; it < total; it++) {
        const double eps0 = step_size;
        /* momentum: r = eps / sqrt(M^-1) */
        for (int i = 0; i < D; i++) r0[i] = rng_normal(st[i]) / sqrt(minv[i]);
        const double E0 = U + kinetic(minv, r0, D);

        tree_reset(&tr);
        memcpy(tr.left.z, theta, sizeof(double) * D); memcpy(tr.left.r, r0, sizeof(double) * D); memcpy(tr.left.g, grad, sizeof(double) * D);
        edge_copy(&tr.right, &tr.left, D);
        memcpy(tr.zp, theta, sizeof(double) * D); memcpy(tr.gp, grad, sizeof(double) * D);
        tr.Up = U; tr.Ep = E0; tr.weight = 0.0; memcpy(tr.rsum, r0, sizeof(double) * D);

        while (tr.depth < max_tree_depth && !tr.turning && !tr.diverging) {
            const int going_right = (int)(orc_rng_next(sdir) >> 31);
            const double eps = going_right ? eps0 : -eps0;
            /* ---- _iterative_build_subtree ---- */
            tree_reset(&sub);
            int sub_turning = 0;
            const int max_prop = 1 << tr.depth;
            edge_copy(&cur, going_right ? &tr.right : &tr.left, D);
            while (sub.nprop < max_prop && !sub_turning && !sub.diverging) {
                /* velocity Verlet leaf (_build_basetree) */
                for (int i = 0; i < D; i++) {
                    rh[i] = cur.r[i] - 0.5 * eps * cur.g[i];
                    nw.z[i] = cur.z[i] + eps * minv[i] * rh[i];
                }
                const double Un = orc_potential_grad(d, nw.z, nw.g);
                nleap[it < num_warmup ? 0 : 1]++;
                for (int i = 0; i < D; i++) nw.r[i] = rh[i] - 0.5 * eps * nw.g[i];
                const double En = Un + kinetic(minv, nw.r, D);
                double dE = En - E0;
                if (isnan(dE)) dE = INFINITY;
                const double lw = -dE;
                const int ldiv = dE > 1000.0;
                const double lacc = dE <= 0.0 ? 1.0 : exp(-dE);
                const int leaf_idx = sub.nprop;
                if (leaf_idx == 0) {
                    edge_copy(&sub.left, &nw, D); edge_copy(&sub.right, &nw, D);
                    memcpy(sub.zp, nw.z, sizeof(double) * D); memcpy(sub.gp, nw.g, sizeof(double) * D);
                    sub.Up = Un; sub.Ep = En; sub.weight = lw;
                    memcpy(sub.rsum, nw.r, sizeof(double) * D);
                    sub.diverging = ldiv; sub.sum_accept = lacc; sub.nprop = 1;
                } else {
                    /* _combine_tree(sub, leaf, biased=False): uniform transition kernel */
                    edge_copy(going_right ? &sub.right : &sub.left, &nw, D);
                    const double p = expit(lw - sub.weight);
                    const double u = rng_uniform(sc);
                    if (u < p) {
                        memcpy(sub.zp, nw.z, sizeof(double) * D); memcpy(sub.gp, nw.g, sizeof(double) * D);
                        sub.Up = Un; sub.Ep = En;
                    }
                    sub.weight = logaddexp(sub.weight, lw);
                    sub.diverging = ldiv;
                    sub.sum_accept += lacc;
                    for (int i = 0; i < D; i++) sub.rsum[i] += nw.r[i];
                    sub.nprop++;
                }
                /* checkpoints / U-turn inside the subtree (_leaf_idx_to_ckpt_idxs, _is_iterative_turning) */
                const int idx_max = popcount_u((unsigned)leaf_idx >> 1);
                const int idx_min = idx_max - trailing_ones((unsigned)leaf_idx) + 1;
                if ((leaf_idx & 1) == 0) {
                    memcpy(r_ckpt[idx_max], nw.r, sizeof(double) * D);
                    memcpy(rsum_ckpt[idx_max], sub.rsum, sizeof(double) * D);
                } else {
                    for (int i = idx_max; i >= idx_min && !sub_turning; i--) {
                        for (int k = 0; k < D; k++) srs[k] = sub.rsum[k] - rsum_ckpt[i][k] + r_ckpt[i][k];
                        sub_turning = is_turning(minv, r_ckpt[i], nw.r, srs, D);
                    }
                }
                edge_copy(&cur, &nw, D);
            }
            sub.turning = sub_turning;
            /* ---- _combine_tree(tree, sub, biased=True) ---- */
            if (going_right) edge_copy(&tr.right, &sub.right, D); else edge_copy(&tr.left, &sub.left, D);
            for (int i = 0; i < D; i++) tr.rsum[i] += sub.rsum[i];
            double p = exp(sub.weight - tr.weight);
            if (p > 1.0) p = 1.0;
            if (sub.turning || sub.diverging) p = 0.0;
            /* the doubling loop also stops when the new subtree itself turned (numpyro _combine_tree,
             * biased branch: turning = new_tree.turning | _is_turning(...); Hoffman & Gelman Alg. 3: s <- s' * 1[...]) */
            const int turning = sub.turning || is_turning(minv, tr.left.r, tr.right.r, tr.rsum, D);
            const double u = rng_uniform(sc);
            if (u < p) {
                memcpy(tr.zp, sub.zp, sizeof(double) * D); memcpy(tr.gp, sub.gp, sizeof(double) * D);
                tr.Up = sub.Up; tr.Ep = sub.Ep;
            }
            tr.depth++;
            tr.weight = logaddexp(tr.weight, sub.weight);
            tr.turning = turning;
            tr.diverging = sub.diverging;
            tr.sum_accept += sub.sum_accept;
            tr.nprop += sub.nprop;
        }
        const double acc = tr.sum_accept / tr.nprop;
        memcpy(theta, tr.zp, sizeof(double) * D); memcpy(grad, tr.gp, sizeof(double) * D);
        U = tr.Up;

        /* warmup adaptation (hmc_util.warmup_adapter.update_fn) */
        if (it < num_warmup) {
            /* dual averaging on (target - accept_prob) */
            const double g = target_accept - acc;
            ad.t += 1;
            ad.g_avg = (1.0 - 1.0 / (ad.t + 10.0)) * ad.g_avg + g / (ad.t + 10.0);
            ad.x_t = ad.prox_center - sqrt((double)ad.t) / 0.05 * ad.g_avg;
            const double wt = pow((double)ad.t, -0.75);
            ad.x_avg = (1.0 - wt) * ad.x_avg + wt * ad.x_t;
            step_size = (it == num_warmup - 1) ? exp(ad.x_avg) : exp(ad.x_t);
            if (step_size < 1.1754943508222875e-38) step_size = 1.1754943508222875e-38;
            if (step_size > 3.4028234663852886e+38) step_size = 3.4028234663852886e+38;
            const int middle = ad.window_idx > 0 && ad.window_idx < nwin - 1;
            if (middle) {
                ad.n += 1;
                for (int i = 0; i < D; i++) {
                    const double dpre = theta[i] - ad.mean[i];
                    ad.mean[i] += dpre / ad.n;
                    ad.m2[i] += dpre * (theta[i] - ad.mean[i]);
                }
            }
            const int at_end = it == wend[ad.window_idx];
            if (at_end) ad.window_idx++;
            if (at_end && middle) {
                const double n = ad.n;
                for (int i = 0; i < D; i++) {
                    const double var = ad.m2[i] / (n - 1.0);
                    minv[i] = (n / (n + 5.0)) * var + 1e-3 * (5.0 / (n + 5.0));
                    ad.mean[i] = 0.0; ad.m2[i] = 0.0;
                }
                ad.n = 0;
                ss_init(&ad, log(10.0 * step_size));
            }
        } else {
            const int s = it - num_warmup;
            for (int i = 0; i < D; i++) draws[(size_t)s * D + i] = theta[i];
            num_steps[s] = tr.nprop; accept_prob[s] = acc;
            diverging[s] = (unsigned char)tr.diverging; potential[s] = U;
        }
        if (trace_theta) for (int i = 0; i < D; i++) trace_theta[(size_t)it * D + i] = theta[i];
        if (trace_steps) trace_steps[it] = tr.nprop;
        if (trace_eps) trace_eps[it] = eps0;
    }
    if (step_size_out) *step_size_out = step_size;
    if (inv_mass_out) memcpy(inv_mass_out, minv, sizeof(double) * D);
    if (n_leapfrog_total) { n_leapfrog_total[0] = nleap[0]; n_leapfrog_total[1] = nleap[1]; }
    free(block); free(st);
    return 0;
}
