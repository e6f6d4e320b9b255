/*
 * biolith_hip.h -- C-ABI of the MI355X-native occupancy-model NUTS engine.
 *
 * This is the drop-in boundary for ONE path of timmh/biolith:
 *     biolith.utils.fit(biolith.models.occu, ...)          (biolith/utils/fit.py:16-135)
 * Everything numpyro/jax did behind `mcmc.run(...)` (fit.py:128-130) happens behind these
 * entry points in hand-written HIP for gfx950.  Plain pointers and sizes only; no torch / C++
 * types cross the boundary; no exception crosses it (int status + bl_last_error()).
 *
 * Ownership: the caller allocates every output.  bl_dataset_create copies its inputs; the
 * library keeps nothing of the caller's after any call returns, except its own opaque handle.
 * A handle is not thread-safe; use one host thread (or process) per handle.
 */
#ifndef BIOLITH_HIP_H
#define BIOLITH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BL_ABI_VERSION 1

enum {
    BL_OK = 0,
    BL_ERR_INVALID = 1,      /* bad argument / unsupported shape (message says which)       */
    BL_ERR_NO_DEVICE = 2,    /* no HIP device / HIP runtime error                            */
    BL_ERR_UNSUPPORTED = 3,  /* model option outside the built path                          */
    BL_ERR_TIMEOUT = 4,      /* in-kernel spin bound hit (a cooperating workgroup vanished)  */
    BL_ERR_ABORTED = 5,      /* bl_nuts_abort() was honoured                                 */
    BL_ERR_BUSY = 6,         /* a launch is still in flight on this handle                   */
    BL_ERR_COMM = 7          /* RCCL could not be loaded, or a communicator call failed      */
};

#define BL_RNG_STREAMS_PER_CHAIN 64
#define BL_MAX_COVS 16 /* per side (site / observation) */

/* Shapes as the reference names them (biolith/models/occu.py:116-133). */
typedef struct bl_dims {
    int32_t n_species;    /* S : the "species" plate (occu.py:182).  S > 1 = ONE chain over all species' coefficients:
                           *     bl_dataset_create / _fp (S (Ks + Ko + 2) (+ 1) <= 60 coordinates) and _re (S <= 8); the other
                           *     models take one species per handle                                                       */
    int32_t n_sites;      /* N                                                               */
    int32_t n_periods;    /* T : stacked periods sharing psi (occu.py:198-210)               */
    int32_t n_replicates; /* J : visits per period                                           */
    int32_t n_site_covs;  /* Ks                                                              */
    int32_t n_obs_covs;   /* Ko                                                              */
} bl_dims;

/* prior.expand([K+1]).to_event(1) with prior = Normal(loc, scale)
 * (biolith/regression/linear.py:28, defaults occu.py:28-29). */
typedef struct bl_normal_prior {
    double loc;
    double scale;
} bl_normal_prior;

typedef struct bl_dataset bl_dataset;

int bl_abi_version(void);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char *bl_last_error(void);
int bl_device_count(int *count);

/*
 * Replaces what numpyro traces out of occu() on first call (fit.py:128 -> occu.py:102-242):
 * validates shapes (occu.py:103-133), builds the missing-data mask (occu.py:136-142,
 * utils/modeling.py:15-17), NaN->0 on covariates, and lays the data out site-fastest in HBM.
 *   site_covs [N][Ks]  obs_covs [N][T][J][Ko]  obs [S][N][T][J]   (row-major float32, NaN = missing)
 * exactly the arrays prepare_data() produces (utils/data.py:135-140).
 */
int bl_dataset_create(const bl_dims *dims, const float *site_covs, const float *obs_covs,
                      const float *obs, const bl_normal_prior *prior_beta,
                      const bl_normal_prior *prior_alpha, int device, bl_dataset **out);
/*
 * Same for the Royle-Nichols model biolith.models.occu_rn (models/occu_rn.py:20-222): latent
 * abundance N ~ RightTruncatedPoisson(exp(beta0 + x beta), max_abundance) (utils/distributions.py:6-40)
 * summed out in the kernel, detection 1 - (1 - sigmoid(alpha0 + w alpha))^N.  Built for
 * max_abundance <= 127 (covariates: up to BL_MAX_COVS per side, as for every model).  The handle then behaves exactly like an
 * occu handle (bl_logp_grad, bl_nuts_*); bl_deterministic's first output becomes
 * `abundance` = exp(beta0 + x beta) (occu_rn.py:192).
 */
int bl_dataset_create_rn(const bl_dims *dims, const float *site_covs, const float *obs_covs,
                         const float *obs, int max_abundance, const bl_normal_prior *prior_beta,
                         const bl_normal_prior *prior_alpha, int device, bl_dataset **out);
/*
 * Dynamic (multi-season) occupancy -- BUILDER-DEFINED, NO REFERENCE COUNTERPART: BASELINE.json configs[4] names a model
 * timmh/biolith does not have (its periods share one psi, models/occu.py:198-210).  Initial occupancy psi, colonisation gamma and
 * extinction eps, each logit-linear in the site covariates; detection as in occu; the latent paths summed out by the forward
 * recursion in the kernel (csrc/dyn_device.hpp).  theta = [b_psi | b_gamma | b_eps (n_site_covs + 1 each) | alpha (n_obs_covs + 1)],
 * every coefficient under the Normal priors given (prior_beta for the three site-side blocks).  n_site_covs <= 8; one species.
 * The handle serves bl_logp_grad and bl_nuts_*; deterministic sites are formed by the caller from the draws.
 */
int bl_dataset_create_dyn(const bl_dims *dims, const float *site_covs, const float *obs_covs,
                          const float *obs, const bl_normal_prior *prior_beta,
                          const bl_normal_prior *prior_alpha, int device, bl_dataset **out);
/*
 * Same for occu with a false-positive rate (models/occu.py:146-157, 229-241):
 *   P(y=1 | z) = 1 - (1 - z p)(1 - f_c)(1 - (1 - z) f_u),  exactly one of f_c / f_u sampled:
 *   fp_mode BL_FP_CONSTANT   = false_positives_constant=True   (site "prob_fp_constant",   prior occu.py:32)
 *   fp_mode BL_FP_UNOCCUPIED = false_positives_unoccupied=True (site "prob_fp_unoccupied", prior occu.py:33)
 * prior_fp is the Beta(a, b) prior of that rate (NULL = Beta(2, 5), the reference default).  The handle's
 * parameter vector gains one trailing coordinate phi = logit(rate) (NumPyro's unconstrained space for
 * a unit-interval site), so bl_dataset_param_dim = Ks + Ko + 3 and draws[..., D-1] is phi.
 */
typedef struct { double a, b; } bl_beta_prior;
enum { BL_FP_CONSTANT = 1, BL_FP_UNOCCUPIED = 2 };
int bl_dataset_create_fp(const bl_dims *dims, const float *site_covs, const float *obs_covs,
                         const float *obs, int fp_mode, const bl_beta_prior *prior_fp,
                         const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha,
                         int device, bl_dataset **out);
/*
 * Same for the count occupancy model biolith.models.occu_cop (models/occu_cop.py:17-255):
 *   y_itj ~ Poisson(session_duration_itj * (z lambda_itj + (1 - z) f_u + f_c)),  lambda = exp(alpha0 + w alpha),
 * obs holds counts, session_duration [N][T][J] the exposure (occu_cop.py:176, 250-254).  fp_mode 0: no
 * false-positive rate; BL_FP_CONSTANT / BL_FP_UNOCCUPIED: sites "rate_fp_constant" / "rate_fp_unoccupied"
 * with an Exponential(prior_fp_rate) prior (occu_cop.py:31-32, 158-170) sampled as a trailing coordinate
 * phi = log(rate).  bl_deterministic's second output becomes `rate_detection` = exp(alpha0 + w alpha)
 * (occu_cop.py:236-243).  Predictive draws: bl_predict_counts.
 */
int bl_dataset_create_cop(const bl_dims *dims, const float *site_covs, const float *obs_covs,
                          const float *obs, const float *session_duration, int fp_mode,
                          double prior_fp_rate, const bl_normal_prior *prior_beta,
                          const bl_normal_prior *prior_alpha, int device, bl_dataset **out);
/*
 * Same for the N-mixture model biolith.models.nmixture (models/nmixture.py:17-220): obs holds counts,
 * N ~ Poisson(exp(beta0 + x beta)) enumerated over 0..max_abundance (raw, un-renormalised weights: the model's
 * "N_i_trunc_norm" factor, nmixture.py:183-196; support cut below the largest count of each (site, period),
 * nmixture.py:150-155), y ~ Binomial(N, sigmoid(alpha0 + w alpha)).  bl_deterministic's outputs are `abundance`
 * and `prob_detection`.  Built for max_abundance <= 127; predictive draws: bl_predict_counts.
 */
int bl_dataset_create_nmix(const bl_dims *dims, const float *site_covs, const float *obs_covs,
                           const float *obs, int max_abundance, const bl_normal_prior *prior_beta,
                           const bl_normal_prior *prior_alpha, int device, bl_dataset **out);
/*
 * Same for biolith.models.occu with site_random_effects / obs_random_effects (models/occu.py:36-39, 170-173, 191-196,
 * 215-218): site_re_sd, obs_re_sd ~ HalfNormal(scale) sampled before the plates; per site site_re_occ, site_re_det ~
 * Normal(0, site_re_sd) join the occupancy and detection predictors; per replicate obs_re ~ Normal(0, obs_re_sd) joins
 * the detection predictor (masked replicates keep their prior term).  Coordinates, in NumPyro's
 * unconstrained space:  theta = [beta, alpha, (log site_re_sd), (log obs_re_sd), (site_re_occ[N], site_re_det[N]),
 * (obs_re[N][T][J])], D = bl_dataset_param_dim().  Several species (dims->n_species <= 8, obs [S][N][T][J]) are ONE chain, as in
 * the reference: beta, alpha and the effects sit inside the species plate (occu.py:182-196), the two sds outside it
 * (occu.py:170-173), i.e. shared:  theta = [species 0: beta, alpha | species 1: ... | (log site_re_sd) | (log obs_re_sd) |
 * site_re_occ [S][N] | site_re_det [S][N] | obs_re [S][N][T][J]]; a chain then runs on a multiple of S workgroups, each
 * slicing the sites of one species; bl_deterministic / bl_predict take one-species handles (cut the draws per species).
 * bl_logp_grad / bl_nuts_* work as for the other models (the sampler
 * runs k workgroups per chain -- site slices, partial sums exchanged through device memory; all num_chains x k must be
 * resident, so num_chains x k <= compute units -- with its vectors in device memory / LDS; RNG: one stream per coordinate, D + 2 per chain;
 * with D <= 61 the layout of the other models: 64 per chain, the two scalar streams last).
 * At most 16 covariates per side.  bl_deterministic adds the effects to both predictors; bl_predict draws z and y from them.
 */
int bl_dataset_create_re(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                         int site_random_effects, int obs_random_effects, double prior_site_re_sd_scale,
                         double prior_obs_re_sd_scale, const bl_normal_prior *prior_beta,
                         const bl_normal_prior *prior_alpha, int device, bl_dataset **out);
/* The same with a false-positive rate on top (occu.py:146-157 together with :170-173, 191-196; fp_mode / prior_fp as for
 * bl_dataset_create_fp): theta = [beta, alpha, phi = logit(rate), (log sds), (effects)].  One species per dataset. */
int bl_dataset_create_re_fp(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                            int site_random_effects, int obs_random_effects, double prior_site_re_sd_scale,
                            double prior_obs_re_sd_scale, int fp_mode, const bl_beta_prior *prior_fp,
                            const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha, int device, bl_dataset **out);
/* The N-mixture model with the same random effects (biolith/models/nmixture.py:139-141, 166-172, 199-214: site_re_abu joins the
 * abundance predictor, site_re_det and obs_re the detection predictor): `counts` / max_abundance as for bl_dataset_create_nmix,
 * theta = [beta, alpha, (log site_re_sd), (log obs_re_sd), (site_re_abu[N], site_re_det[N]), (obs_re[N][T][J])].  One species
 * per dataset.  bl_logp_grad / bl_nuts_* as above; bl_deterministic returns abundance = exp(eta + site_re_abu) and the
 * detection probability with its effects; bl_predict_counts draws N and the counts from them. */
int bl_dataset_create_nmix_re(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *counts,
                              int max_abundance, int site_random_effects, int obs_random_effects,
                              double prior_site_re_sd_scale, double prior_obs_re_sd_scale, const bl_normal_prior *prior_beta,
                              const bl_normal_prior *prior_alpha, int device, bl_dataset **out);
/* The Royle-Nichols model with the same random effects (biolith/models/occu_rn.py:151-154, 172-184, 199-212): `obs` / max_abundance
 * as for bl_dataset_create_rn, theta laid out as for bl_dataset_create_nmix_re.  One species per dataset.  bl_deterministic returns
 * abundance = exp(eta + site_re_abu) and the detection probability with its effects; bl_predict draws N and y from them. */
int bl_dataset_create_rn_re(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                            int max_abundance, int site_random_effects, int obs_random_effects,
                            double prior_site_re_sd_scale, double prior_obs_re_sd_scale, const bl_normal_prior *prior_beta,
                            const bl_normal_prior *prior_alpha, int device, bl_dataset **out);
/* ... and with a false-positive rate (occu_rn.py:133-138, 214-221: y ~ Bernoulli(1 - (1 - p)(1 - f)), f ~ Beta(a, b)), with or without the
 * random effects: theta = [beta, alpha, phi = logit f, (log sds), (effects)].  One species.  Runs on the random-effects kernels in
 * either case; bl_deterministic / bl_predict apply the effects and the rate. */
int bl_dataset_create_rn_fp(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *obs,
                            int max_abundance, int site_random_effects, int obs_random_effects,
                            double prior_site_re_sd_scale, double prior_obs_re_sd_scale, const bl_beta_prior *prior_fp,
                            const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha, int device, bl_dataset **out);
/* occu_cop with the same random effects (biolith/models/occu_cop.py:183-186, 204-210, 229-243: site_re_occ joins the occupancy
 * predictor, site_re_det and obs_re the log detection rate), with or without a false-positive rate (occu_cop.py:158-170, 244-248):
 * `counts` / `session_duration` / fp_mode / prior_fp_rate as for bl_dataset_create_cop, theta = [beta, alpha, (phi = log rate_fp),
 * (log sds), (effects)].  One species.  bl_deterministic returns psi and rate_detection with the effects; bl_predict_counts draws z
 * and the counts from them. */
int bl_dataset_create_cop_re(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *counts,
                             const float *session_duration, int fp_mode, double prior_fp_rate, int site_random_effects,
                             int obs_random_effects, double prior_site_re_sd_scale, double prior_obs_re_sd_scale,
                             const bl_normal_prior *prior_beta, const bl_normal_prior *prior_alpha, int device, bl_dataset **out);
/*
 * The continuous-score occupancy model biolith.models.occu_cs (models/occu_cs.py:17-232; Rhinehart et al. 2022): `scores`
 * [S=1][N][T][J] (NaN = missing) ~ Normal(mu_f, sigma_f) with f ~ Bernoulli(z p) and z ~ Bernoulli(psi) summed out.
 * prior_mu = {loc, scale of mu0; loc, scale of the Normal that mu1 follows truncated below at mu0}; prior_sigma =
 * {concentration, rate of sigma0's Gamma; of sigma1's}.  theta = [beta, alpha, mu0, log(mu1 - mu0), log sigma0, log sigma1]
 * (NumPyro's unconstrained space), D = Ks + Ko + 6.  Runs on the random-effects kernels' framework (bl_dataset_create_re);
 * bl_deterministic gives psi and prob_detection; predictive draws: bl_predict_scores.
 */
int bl_dataset_create_cs(const bl_dims *dims, const float *site_covs, const float *obs_covs, const float *scores,
                         const double *prior_mu, const double *prior_sigma, const bl_normal_prior *prior_beta,
                         const bl_normal_prior *prior_alpha, int device, bl_dataset **out);
/*
 * The other prior family biolith.utils.grid_search_priors tries for the regression coefficients
 * (utils/grid_search.py:366-371): Laplace(loc, scale) instead of Normal(loc, scale), per side, with the (loc, scale) the
 * dataset was created with.  Call between bl_dataset_create* and the first use; applies to every model.
 */
#define BL_PRIOR_NORMAL 0
#define BL_PRIOR_LAPLACE 1
int bl_dataset_set_prior_family(bl_dataset *ds, int family_beta, int family_alpha);
int bl_dataset_destroy(bl_dataset *ds);
/* Coordinates of theta.  occu / occu_rn / nmixture / occu_cop without a rate: D = Ks+1 + Ko+1, theta = [beta_0..beta_Ks,
 * alpha_0..alpha_Ko]; with a false-positive coordinate (bl_dataset_create_fp, occu_cop with a rate): + 1 (trailing phi);
 * S species under one chain: S (Ks + Ko + 2) (+ 1); dynamic occupancy: 3 (Ks + 1) + Ko + 1; occu_cs: Ks + Ko + 6; random
 * effects: Ks + Ko + 2 + the log sds + 2 N (site effects) + N T J (observation effects), per species where S > 1. */
int bl_dataset_param_dim(const bl_dataset *ds, int *D);

/*
 * Parity hook for the likelihood kernel: U = -log p(theta, y) (z marginalised, Normal prior
 * normaliser included) and dU/dtheta for B parameter vectors.  Replaces
 * jax.value_and_grad(potential_fn) over occu.py:136-242 + funsor enumeration (occu.py:208-210).
 * `staged` != 0 evaluates through the same LDS-staged code path the NUTS kernel uses.
 */
int bl_logp_grad(bl_dataset *ds, int B, const double *theta /*[B][D]*/, double *U /*[B]*/,
                 double *grad /*[B][D]*/, int staged);

/* MCMC(NUTS(model), num_samples, num_warmup, num_chains).run(PRNGKey(seed))  (fit.py:92-130) */
typedef struct bl_nuts_config {
    int32_t num_warmup;     /* fit.py:23 default 1000 */
    int32_t num_samples;    /* fit.py:22 default 1000 */
    int32_t num_chains;     /* chains run by THIS call (fit.py:25 default 5)                 */
    int32_t chain_offset;   /* global id of this call's first chain: selects its RNG streams */
    uint64_t seed;          /* fit.py:24 / fit.py:122                                       */
    int32_t max_tree_depth; /* numpyro NUTS default 10                                       */
    int32_t wgs_per_chain;  /* 0 = auto; workgroups (CUs) cooperating on one chain           */
    double target_accept;   /* numpyro NUTS default 0.8                                      */
    const double *init_theta; /* NULL = init_to_uniform(radius 2) (fit.py:93); else [C][D]  */
} bl_nuts_config;

/* Host pointers, caller-allocated; NULL = not wanted.  C = num_chains, S = num_samples. */
typedef struct bl_nuts_output {
    float *draws;            /* [C][S][D] post-warmup positions (mcmc.get_samples, fit.py:132) */
    uint8_t *diverging;      /* [C][S]   extra field "diverging" (diagnostics.py:34-38)        */
    int32_t *num_steps;      /* [C][S]   leapfrogs of each transition                          */
    float *accept_prob;      /* [C][S]                                                          */
    float *potential_energy; /* [C][S]                                                          */
    float *step_size;        /* [C]      adapted step size                                      */
    float *inv_mass;         /* [C][D]   adapted diagonal inverse mass                          */
    int64_t *n_leapfrog;     /* [C][2]   gradient evaluations: warmup, sampling                 */
} bl_nuts_output;

/* Blocking convenience: launch + wait + fetch. */
int bl_nuts_run(bl_dataset *ds, const bl_nuts_config *cfg, bl_nuts_output *out);

/* Asynchronous form (timeouts, overlap, timing).  `stream` is a hipStream_t or NULL.
 * The kernel is persistent and its workgroups exchange data while they run, so all of them must be resident
 * at once: launches on ONE device belong on one stream (they then run back to back), launches on different
 * devices overlap freely.  If the device is shared and a workgroup cannot become resident, the bounded spins
 * expire and bl_nuts_wait returns BL_ERR_TIMEOUT instead of hanging. */
int bl_nuts_launch(bl_dataset *ds, const bl_nuts_config *cfg, void *stream);
int bl_nuts_poll(bl_dataset *ds, int *done);
int bl_nuts_abort(bl_dataset *ds); /* fit(timeout=...) (fit.py:124-128): kernel exits at its next leapfrog */
int bl_nuts_wait(bl_dataset *ds);  /* returns BL_ERR_TIMEOUT / BL_ERR_ABORTED if the kernel reported it */
int bl_nuts_fetch(bl_dataset *ds, bl_nuts_output *out);
/* HIP-event time of the last launch (kernel + its memset), on the launch stream. */
int bl_nuts_elapsed_ms(bl_dataset *ds, float *ms);
/* Device address of the last launch's draws [C][S][D] float32 (for an RCCL gather). */
int bl_nuts_device_draws(bl_dataset *ds, void **dev_ptr, size_t *bytes);
/* Geometry the last launch used; the last field counts the chains whose workgroups were verified
 * (HW_REG_XCC_ID census) to share one XCD and therefore ran the L2-local exchange.  lds_staged: bit 0 = the
 * site data were staged in LDS; random-effects / occu_cs launches: bits 1-2 = how many of the sampler's
 * vectors lived in LDS (0 none, 1 the five of the leaf in flight, 2 every vector a leapfrog touches). */
int bl_nuts_geometry(bl_dataset *ds, int *wgs_per_chain, int *threads_per_wg, int *lds_bytes, int *lds_staged,
                     int *chains_on_l2_local_exchange);

/* Lanes that shared one site pair in the last launch (1, 1 = one pair per lane): period_lanes split the pair's periods, visit_lanes
 * the visits of a period -- the layout SURVEY.md 7.1 asks for when a site has many visits (simulate() defaults: 52 per site,
 * biolith/models/occu.py:251-252, 336; benchmarks/occu_spoccupancy.py:16-70: up to 90).  Environment knobs for tests and A/B
 * runs: BIOLITH_HIP_OCCU_G = 1 | 2 | 4 | 8 | 16 forces the lanes per pair, BIOLITH_HIP_OCCU_GT the log2 of the period lanes. */
int bl_nuts_lane_group(bl_dataset *ds, int *period_lanes, int *visit_lanes);

/* Name of the sampler instantiation the last launch ran, as a profiler prints it (e.g. "bl_nuts_kernel<3, 3, true, 0, 3, false, 5, true>":
 * capacities, LDS-staged, model, compute waves, lane-group form, visits-per-period form, lean form; or, for the random-effects models,
 * "bl_re_nuts_kernel<4, 0, true, 2, 5>": covariate capacity, model kind, rows in LDS, sampler-vector tier in LDS, effects / one period as
 * compile-time facts).  Measurement only: bench.py's roofline.kernel. */
int bl_nuts_kernel_name(bl_dataset *ds, char *buf, int n);

/* The BIOLITH_HIP_* environment knobs (tests, A/B runs, measurement; INTEGRATION.md lists every one) that were SET when the last
 * launch on this handle read the environment, as "NAME=value,NAME=value" ("" = none: the geometry and the kernel form are the
 * engine's own choice).  The launch path reads the environment in one place, once per launch.  bl_env_overrides: the same for the
 * environment as it is now (no handle).  Not part of the reference's interface (numpyro has no such knobs); bench.py prints it so
 * that a stray variable cannot change a measured geometry silently. */
int bl_nuts_env_overrides(bl_dataset *ds, char *buf, int n);
int bl_env_overrides(char *buf, int n);

/* Page-locked host memory for large outputs (bl_deterministic / bl_predict write into caller memory; into page-locked memory the
 * device copies at PCIe rate instead of staging through the runtime's bounce buffers).  The caller owns and frees it.  The
 * reference's counterpart is jax.device_get() of a deterministic site (utils/fit.py:132). */
int bl_host_alloc(size_t bytes, void **out);
int bl_host_free(void *ptr);

/* In-kernel phase cycle counters of the last launch; all zero unless the library is a diagnostic
 * BL_STAMPS build (make -C biolith_amd/csrc stamps).  Not part of the reference's interface. */
int bl_nuts_debug_counters(bl_dataset *ds, int64_t *out /*[n<=544]*/, int n);

/*
 * Deterministic sites (occu.py:207, 221-228), recomputed from draws on the device:
 *   psi            [n_draws][T][N]      = sigmoid(beta_0 + X beta)       (constant over T)
 *   prob_detection [n_draws][J][T][N]   = sigmoid(alpha_0 + W alpha)
 * draws is [n_draws][D] float32 on the host; outputs are host float32 (NULL = skip).
 */
int bl_deterministic(bl_dataset *ds, int n_draws, const float *draws, float *psi, float *prob_detection);

/*
 * Posterior predictive draws of the discrete sites, one per posterior draw -- what
 * numpyro.infer.Predictive(model_fn, posterior_samples)(key, site_covs, obs_covs) samples in
 * biolith/utils/predict.py:66-92 (obs withheld, so no observation is masked):
 *   occu     latent = z   [n_draws][T][N]  ~ Bernoulli(psi)                         (occu.py:208-210)
 *            y            [n_draws][J][T][N] ~ Bernoulli(z * prob_detection)        (occu.py:229-241)
 *   occu_rn  latent = N_i [n_draws][T][N]  ~ truncated Poisson(abundance) on 0..max_abundance (occu_rn.py:194-198)
 *            y            ~ Bernoulli(1 - (1 - prob_detection)^N_i)                 (occu_rn.py:211-221)
 *   false-positive handles: y ~ Bernoulli(1 - (1 - z p)(1 - f_c)(1 - (1 - z) f_u))   (occu.py:229-241)
 * uint8 on the host (NULL = skip).  The sample is a function of (seed, draw, period, site) only; it is
 * distributionally, not bitwise, the reference's (JAX threefry keys are not reproduced).
 */
int bl_predict(bl_dataset *ds, int n_draws, const float *draws, uint64_t seed, uint8_t *latent, uint8_t *y);

/*
 * The same for the count models, whose sampled sites do not fit a byte (predict.py:66-92):
 *   occu_cop  latent = z,  y ~ Poisson(session_duration * (z rate_detection + (1 - z) f_u + f_c))   (occu_cop.py:222-255)
 *   nmixture  latent = N_i ~ Poisson(abundance) restricted to 0..max_abundance,  y ~ Binomial(N_i, prob_detection)
 *             (nmixture.py:183-220; with obs withheld there is no largest-count cut-off)
 * int32 on the host, shapes as in bl_predict.  Poisson draws by inversion (rate < 10) or Hoermann's PTRS.
 */
int bl_predict_counts(bl_dataset *ds, int n_draws, const float *draws, uint64_t seed, int32_t *latent, int32_t *y);
/* Same for occu_cs (occu_cs.py:196-232 with obs withheld): z[n][T][N], f[n][J][T][N] (which score distribution a replicate
 * drew from) as bytes, s[n][J][T][N] the scores; any of the three may be NULL. */
int bl_predict_scores(bl_dataset *ds, int n_draws, const float *draws, uint64_t seed, uint8_t *latent, uint8_t *f, float *s);

/*
 * Multi-GPU: chain-parallel sampling and the gather of the draws.
 *
 * The reference's only multi-device strategy is chain_method="parallel" (biolith/utils/fit.py:109-113: one chain per
 * local device under pmap); its gather is the implicit device->host copy of mcmc.get_samples() (fit.py:132).  Here rank
 * r runs its own chains (bl_nuts_config.chain_offset selects their RNG streams), nothing is exchanged while sampling, and
 * ONE collective over RCCL / xGMI follows: every rank contributes the result block of its chains (draws, diverging,
 * num_steps, accept_prob, potential_energy, step_size, inv_mass, n_leapfrog -- contiguous in device memory) and receives
 * everybody's.  librccl is loaded on first use (BL_ERR_COMM if it cannot be).
 *
 * Two ways to make the communicators:
 *   process per GPU   rank 0 calls bl_comm_unique_id and hands the 128 bytes to the other ranks by any side channel
 *                     (a file, a socket, torch.distributed's store); every rank calls bl_comm_init_rank.
 *   one process       bl_comm_init_all(ndev, devices, comms) -- the in-process fit(..., devices=[...]).
 */
#define BL_COMM_ID_BYTES 128
typedef struct bl_comm bl_comm;
int bl_comm_rccl_version(int *version);
int bl_comm_unique_id(uint8_t *id /*[BL_COMM_ID_BYTES]*/);
int bl_comm_init_rank(const uint8_t *id /*[BL_COMM_ID_BYTES]*/, int world, int rank, int device, bl_comm **out);
int bl_comm_init_all(int ndev, const int *devices /*[ndev], distinct*/, bl_comm **out /*[ndev]*/);
/* world size, this communicator's rank and device, wall time its creation took (reported, never inside a timed region) */
int bl_comm_info(const bl_comm *comm, int *world, int *rank, int *device, double *init_ms);
int bl_comm_destroy(bl_comm *comm);
/*
 * All-gather of the last finished launch of every local dataset (bl_nuts_wait first).  comms / datasets: the n_local
 * ranks this process drives (1 for process-per-GPU; all of them after bl_comm_init_all), dataset i on communicator i's
 * device.  chains_per_rank[world]: chains each rank ran -- every rank can compute it (chains are dealt in contiguous
 * blocks), so block sizes are never negotiated.  Equal counts: one ncclAllGather; unequal: its "v" form, one grouped
 * set of broadcasts.  out (host, caller-allocated for sum(chains_per_rank) chains, fields may be NULL; out itself may be
 * NULL on ranks that do not want the result) receives the chains in rank order.
 */
int bl_gather_draws(bl_comm *const *comms, bl_dataset *const *datasets, int n_local, const int32_t *chains_per_rank,
                    bl_nuts_output *out);
/*
 * The host-only half of the gather (no GPU, no RCCL needed; what a machine without GPUs can test of the N > 1 path).
 * bl_result_block_layout: byte offsets of {draws, diverging, num_steps, accept_prob, potential_energy, step_size, inv_mass,
 * n_leapfrog, end} inside ONE rank's result block for `chains` chains of `num_samples` draws of D coordinates.
 * bl_gather_unpack: scatters a gathered buffer -- the world blocks back to back in rank order, exactly what
 * bl_gather_draws receives on the device -- into the caller's arrays, the chains in rank order.
 */
int bl_result_block_layout(int chains, int num_samples, int D, uint64_t *offsets /*[9]*/);
int bl_gather_unpack(const void *gathered, uint64_t gathered_bytes, int world, const int32_t *chains_per_rank,
                     int num_samples, int D, bl_nuts_output *out);

/* The engine's xoshiro128++ streams (host-side; no GPU needed): out[nstreams][4]. */
int bl_rng_streams(uint64_t seed, int chain, int nstreams, uint32_t *out);
/* numpyro build_adaptation_schedule restatement used by the kernel: returns window count. */
int bl_adaptation_schedule(int num_warmup, int32_t *starts, int32_t *ends, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* BIOLITH_HIP_H */
