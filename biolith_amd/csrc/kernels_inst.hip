// kernels_inst.hip -- one translation unit per padded covariate capacity pair (BL_KS, BL_KO).
// Built N times by the Makefile (-DBL_KS=.. -DBL_KO=..) so the instantiations compile in parallel.
// model 0 = occu (LDS-staged and HBM-row forms); model 1 = occu_rn and model 2 = occu with false
// positives, model 3 = occu_cop, model 4 = nmixture (LDS-staged form, capacities <= 4).
#include <cstdlib>
#include "logp_kernel.hpp"
#include "nuts_kernel.hpp"

#if !defined(BL_KS) || !defined(BL_KO)
#error "compile with -DBL_KS=<n> -DBL_KO=<n>"
#endif
#define BL_CAT3(a, b, c) a##_##b##_##c
#define BL_NAME(base, ks, ko) BL_CAT3(base, ks, ko)
#if 1 // every model at every capacity pair (the count models and occu_rn were once limited to <= 4 covariates per side)
#define BL_HAVE_RN 1
#else
#define BL_HAVE_RN 0
#endif

// Per-FORM instantiations of the sampler (JSEL / LEAN / lean lane-group and Royle-Nichols forms: nuts_kernel.hpp) exist for the capacity
// pairs up to 4 + 4 -- what the reference's own datasets, its benchmark grid and BASELINE.json's configs use; with 8 or 16 covariates on
// a side the general kernels serve (they carry every form at run time: 3-7 % slower on the shapes a form serves).  20 of the 36
// translation units are then 40 % smaller: the library builds from scratch in 5 minutes on 8 cores instead of 5.5 (bounded by biolith_hip.o) and is 46 MB, not 55.
#if BL_KS <= 4 && BL_KO <= 4
#define BL_FORMS_FULL 1
#else
#define BL_FORMS_FULL 0
#endif

// `name`: the instantiation as a profiler prints it (handed to the host: bl_nuts_kernel_name, bench.py's roofline.kernel)
extern "C" void bl_note_kernel_name(const char *name);
template <auto Kernel, typename P>
static int bl_launch(const char *name, const P *p, int grid, int threads, int lds_bytes, hipStream_t stream)
{
    bl_note_kernel_name(name);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(Kernel, dim3(grid), dim3(threads), lds_bytes, stream, *p);
    return (int)hipGetLastError();
}
#define BL_STR2(x) #x
#define BL_STR(x) BL_STR2(x)
#define BL_KHEAD(KERNEL, LDS, MODEL, CW) #KERNEL "<" BL_STR(BL_KS) ", " BL_STR(BL_KO) ", " #LDS ", " BL_STR(MODEL) ", " BL_STR(CW)
#define BL_TAIL_bl_nuts_kernel ", false, -1, false>"
#define BL_TAIL_bl_logp_kernel ">"

// Instantiations (CW = compute waves per workgroup, chosen by the host's choose_geometry):
//   occu and false positives, LDS-staged: CW 3 and 4, and BL_CWAVES_SINGLE (7) for chains of ONE workgroup (small problems: no exchange);  occu, HBM rows: CW 4;  occu_rn: CW 7 (BL_CWAVES_RN);  false positives, occu_cop, nmixture: CW 3 and 4.
#define BL_PICK(KERNEL, P, LDS, MODEL, CW) bl_launch<KERNEL<BL_KS, BL_KO, LDS, MODEL, CW>>(BL_KHEAD(KERNEL, LDS, MODEL, CW) BL_TAIL_##KERNEL, P, grid, 64 * (CW + 1), lds_bytes, stream)
// the sampler's GRP instantiation (lane groups / one workgroup per chain: nuts_kernel.hpp) of the plain and false-positive models
#define BL_PICK_GRP(P, MODEL, CW) bl_launch<bl_nuts_kernel<BL_KS, BL_KO, true, MODEL, CW, true>>(BL_KHEAD(bl_nuts_kernel, true, MODEL, CW) ", true, -1, false>", P, grid, 64 * (CW + 1), lds_bytes, stream)
// ... and its lean form (one species, one-batch poll; nuts_kernel.hpp LEAN)
#define BL_PICK_GRP_LEAN(P, MODEL, CW) bl_launch<bl_nuts_kernel<BL_KS, BL_KO, true, MODEL, CW, true, -1, true>>(BL_KHEAD(bl_nuts_kernel, true, MODEL, CW) ", true, -1, true>", P, grid, 64 * (CW + 1), lds_bytes, stream)
// ... and of one period (JSEL = 1 in a lane-group kernel: occu_device.hpp bl_eval_sites_grp<.., T1>)
#define BL_PICK_GRP_LEAN_T1(P, MODEL, CW) bl_launch<bl_nuts_kernel<BL_KS, BL_KO, true, MODEL, CW, true, 1, true>>(BL_KHEAD(bl_nuts_kernel, true, MODEL, CW) ", true, 1, true>", P, grid, 64 * (CW + 1), lds_bytes, stream)
// (BIOLITH_HIP_GENERAL=1: tests / A/B -- the general kernel although a per-form instantiation would serve; draws must not change by a bit:
// tests/test_gpu_kernel_forms.py)
extern "C" int bl_env_force_general(void); // biolith_hip.hip: BIOLITH_HIP_GENERAL=1 at the launch's one read of the environment (A/B, tests)
static inline bool bl_force_general() { return bl_env_force_general() != 0; }
// the dynamic model on one period per lane (JSEL = 1 in a MODEL 8 lane-group kernel: the two-scans form, dyn_device.hpp)
#define BL_PICK_DYN_SCAN(P, CW) bl_launch<bl_nuts_kernel<BL_KS, BL_KO, true, 8, CW, true, 1, false>>(BL_KHEAD(bl_nuts_kernel, true, 8, CW) ", true, 1, false>", P, grid, 64 * (CW + 1), lds_bytes, stream)
// ... and with eight periods on eight lanes at four visits each as compile-time facts (JSEL = 2)
#define BL_PICK_DYN_SCAN84(P, CW) bl_launch<bl_nuts_kernel<BL_KS, BL_KO, true, 8, CW, true, 2, false>>(BL_KHEAD(bl_nuts_kernel, true, 8, CW) ", true, 2, false>", P, grid, 64 * (CW + 1), lds_bytes, stream)
#define BL_IS_LEAN(P) (!bl_force_general() && (P)->n_species <= 1 && (P)->k <= 8 * (64 / (P)->nvp))
// ... and of one period per lane at four visits each (JSEL = 104: stacked periods, BASELINE.json configs[4]'s stand-in 2 000 x 8 x 4 --
// the lane's period is straight-line code: 2.56 -> 2.32 us per leapfrog there, profiles/r05/l_ab_stacked_own_period.txt)
#define BL_PICK_GRP_LEAN_OWN4(P, MODEL, CW) bl_launch<bl_nuts_kernel<BL_KS, BL_KO, true, MODEL, CW, true, 104, true>>(BL_KHEAD(bl_nuts_kernel, true, MODEL, CW) ", true, 104, true>", P, grid, 64 * (CW + 1), lds_bytes, stream)
#define BL_GRP_OWN4(P) ((P)->J == 4 && ((P)->lane_grp >> 4) == 0 && (1 << ((P)->lane_grp & 15)) == (P)->T && (P)->T > 1)
#if BL_FORMS_FULL
#define BL_PICK_GRP_ANY(P, CW) (!BL_IS_LEAN(P) ? BL_PICK_GRP(P, 0, CW) : ((P)->T == 1 && ((P)->lane_grp & 15) == 0) ? BL_PICK_GRP_LEAN_T1(P, 0, CW) : BL_GRP_OWN4(P) ? BL_PICK_GRP_LEAN_OWN4(P, 0, CW) : BL_PICK_GRP_LEAN(P, 0, CW))
#else
#define BL_PICK_GRP_ANY(P, CW) BL_PICK_GRP(P, 0, CW)
#endif
// the plain model, one pair per lane: one instantiation per visits-per-period form (1 .. 6, 8 unrolled; 0 = any J at run time)
#ifndef BL_J_LEAN
#define BL_J_LEAN true // (A/B: -DBL_J_LEAN=false)
#endif
#define BL_PICK_J(P, CW, JSEL) bl_launch<bl_nuts_kernel<BL_KS, BL_KO, true, 0, CW, false, JSEL, BL_J_LEAN>>(BL_KHEAD(bl_nuts_kernel, true, 0, CW) ", false, " BL_STR(JSEL) ", " BL_STR(BL_J_LEAN) ">", P, grid, 64 * (CW + 1), lds_bytes, stream)
#define BL_PICK_J_ANY(P, CW)                                   \
    switch ((P)->J) {                                          \
    case 1: return BL_PICK_J(P, CW, 1);                        \
    case 2: return BL_PICK_J(P, CW, 2);                        \
    case 3: return BL_PICK_J(P, CW, 3);                        \
    case 4: return BL_PICK_J(P, CW, 4);                        \
    case 5: return BL_PICK_J(P, CW, 5);                        \
    case 6: return BL_PICK_J(P, CW, 6);                        \
    case 8: return BL_PICK_J(P, CW, 8);                        \
    default: return BL_PICK_J(P, CW, 0);                       \
    }

extern "C" int BL_NAME(bl_launch_nuts, BL_KS, BL_KO)(const BlNutsParams *p, int grid, int lds_bytes, int staged, int model, hipStream_t stream)
{
    if (model == 1) { // occu_rn
#if BL_HAVE_RN
#if BL_FORMS_FULL
        // (lean and at most ten visits per period -- config 4 -- : the instantiation that carries the one-group paths alone)
        if (staged && p->ncw == BL_CWAVES_RN && BL_IS_LEAN(p) && p->J <= 10 && p->T == 1)
            return bl_launch<bl_nuts_kernel<BL_KS, BL_KO, true, 1, BL_CWAVES_RN, false, 10, true>>(BL_KHEAD(bl_nuts_kernel, true, 1, BL_CWAVES_RN) ", false, 10, true>", p, grid, 64 * (BL_CWAVES_RN + 1), lds_bytes, stream);
        if (staged && p->ncw == BL_CWAVES_RN && BL_IS_LEAN(p))
            return bl_launch<bl_nuts_kernel<BL_KS, BL_KO, true, 1, BL_CWAVES_RN, false, -1, true>>(BL_KHEAD(bl_nuts_kernel, true, 1, BL_CWAVES_RN) ", false, -1, true>", p, grid, 64 * (BL_CWAVES_RN + 1), lds_bytes, stream);
#endif
        if (staged && p->ncw == BL_CWAVES_RN) return BL_PICK(bl_nuts_kernel, p, true, 1, BL_CWAVES_RN);
#endif
        return (int)hipErrorNotSupported;
    }
    if (model == 2) {
#if BL_HAVE_RN
        if (staged && p->grp_kernel && p->ncw == 3) return BL_PICK_GRP(p, 2, 3);
        if (staged && p->grp_kernel && p->ncw == 4) return BL_PICK_GRP(p, 2, 4);
        if (staged && p->ncw == BL_CWAVES_SINGLE) return BL_PICK_GRP(p, 2, BL_CWAVES_SINGLE);
        if (staged && p->ncw == 3) return BL_PICK(bl_nuts_kernel, p, true, 2, 3);
        if (staged && p->ncw == 4) return BL_PICK(bl_nuts_kernel, p, true, 2, 4);
#endif
        return (int)hipErrorNotSupported;
    }
    if (model == 4) {
#if BL_HAVE_RN
        if (staged && p->ncw == 3) return BL_PICK(bl_nuts_kernel, p, true, 4, 3);
        if (staged && p->ncw == 4) return BL_PICK(bl_nuts_kernel, p, true, 4, 4);
#endif
        return (int)hipErrorNotSupported;
    }
    if (model == 8) { // dynamic occupancy (dyn_device.hpp): site-covariate capacities up to BL_DYN_MAX_KS
#if BL_KS <= BL_DYN_MAX_KS
        // (T <= 2 G: the instantiation that carries the scaled-likelihood form alone; else the one with the first form alone)
        // (one period per lane, T == G: the instantiation that carries the two-scans form alone -- dyn_device.hpp, round 5)
#if BL_FORMS_FULL
        if (BL_DYN_SCAN && !bl_force_general() && staged && p->ncw == 3 && p->T == 8 && p->lane_grp == 8 && p->J == 4) return BL_PICK_DYN_SCAN84(p, 3);
        if (BL_DYN_SCAN && !bl_force_general() && staged && p->ncw == 4 && p->T == 8 && p->lane_grp == 8 && p->J == 4) return BL_PICK_DYN_SCAN84(p, 4);
#endif
        if (BL_DYN_SCAN && staged && p->ncw == 3 && p->T == p->lane_grp && p->T > 1) return BL_PICK_DYN_SCAN(p, 3);
        if (BL_DYN_SCAN && staged && p->ncw == 4 && p->T == p->lane_grp && p->T > 1) return BL_PICK_DYN_SCAN(p, 4);
        if (staged && p->ncw == 3 && p->T <= 2 * p->lane_grp) return BL_PICK_GRP(p, 8, 3);
        if (staged && p->ncw == 4 && p->T <= 2 * p->lane_grp) return BL_PICK_GRP(p, 8, 4);
        if (staged && p->ncw == 3) return BL_PICK(bl_nuts_kernel, p, true, 8, 3);
        if (staged && p->ncw == 4) return BL_PICK(bl_nuts_kernel, p, true, 8, 4);
#endif
        return (int)hipErrorNotSupported;
    }
    if (model == 3) {
#if BL_HAVE_RN
        if (staged && p->ncw == 3) return BL_PICK(bl_nuts_kernel, p, true, 3, 3);
        if (staged && p->ncw == 4) return BL_PICK(bl_nuts_kernel, p, true, 3, 4);
#endif
        return (int)hipErrorNotSupported;
    }
    if (staged && p->grp_kernel && p->ncw == 3) return BL_PICK_GRP_ANY(p, 3);
    if (staged && p->grp_kernel && p->ncw == 4) return BL_PICK_GRP_ANY(p, 4);
    if (staged && p->ncw == BL_CWAVES_SINGLE) return BL_PICK_GRP_ANY(p, BL_CWAVES_SINGLE);
    // one species and a one-batch poll (k <= 8 x 64 / nvp): the lean per-form instantiations; else the kernel that carries everything
    // (and one period, at most one site pair per compute lane: nuts_kernel.hpp LEAN)
#if BL_FORMS_FULL
    const bool lean = (!BL_J_LEAN && !bl_force_general()) || (BL_IS_LEAN(p) && p->T == 1 && p->nloc <= 2 * 64 * p->ncw);
    if (staged && p->ncw == 3 && lean) { BL_PICK_J_ANY(p, 3) }
    if (staged && p->ncw == 4 && lean) { BL_PICK_J_ANY(p, 4) }
#endif
    if (staged && p->ncw == 3) return BL_PICK(bl_nuts_kernel, p, true, 0, 3);
    if (staged && p->ncw == 4) return BL_PICK(bl_nuts_kernel, p, true, 0, 4);
#ifdef BL_OCCU_CWX
    if (staged && p->ncw == BL_OCCU_CWX) return BL_PICK(bl_nuts_kernel, p, true, 0, BL_OCCU_CWX);
#endif
    if (!staged && p->ncw == 4) return BL_PICK(bl_nuts_kernel, p, false, 0, 4);
    return (int)hipErrorNotSupported;
}

extern "C" int BL_NAME(bl_launch_logp, BL_KS, BL_KO)(const BlLogpParams *p, int grid, int lds_bytes, int staged, int model, hipStream_t stream)
{
    if (model == 1) {
#if BL_HAVE_RN
        if (staged && p->ncw == BL_CWAVES_RN) return BL_PICK(bl_logp_kernel, p, true, 1, BL_CWAVES_RN);
#endif
        return (int)hipErrorNotSupported;
    }
    if (model == 2) {
#if BL_HAVE_RN
        if (staged && p->ncw == 3) return BL_PICK(bl_logp_kernel, p, true, 2, 3);
        if (staged && p->ncw == 4) return BL_PICK(bl_logp_kernel, p, true, 2, 4);
        if (staged && p->ncw == BL_CWAVES_SINGLE) return BL_PICK(bl_logp_kernel, p, true, 2, BL_CWAVES_SINGLE);
#endif
        return (int)hipErrorNotSupported;
    }
    if (model == 4) {
#if BL_HAVE_RN
        if (staged && p->ncw == 3) return BL_PICK(bl_logp_kernel, p, true, 4, 3);
        if (staged && p->ncw == 4) return BL_PICK(bl_logp_kernel, p, true, 4, 4);
#endif
        return (int)hipErrorNotSupported;
    }
    if (model == 8) {
#if BL_KS <= BL_DYN_MAX_KS
        if (staged && p->ncw == 3) return BL_PICK(bl_logp_kernel, p, true, 8, 3);
        if (staged && p->ncw == 4) return BL_PICK(bl_logp_kernel, p, true, 8, 4);
#endif
        return (int)hipErrorNotSupported;
    }
    if (model == 3) {
#if BL_HAVE_RN
        if (staged && p->ncw == 3) return BL_PICK(bl_logp_kernel, p, true, 3, 3);
        if (staged && p->ncw == 4) return BL_PICK(bl_logp_kernel, p, true, 3, 4);
#endif
        return (int)hipErrorNotSupported;
    }
    if (staged && p->ncw == 3) return BL_PICK(bl_logp_kernel, p, true, 0, 3);
    if (staged && p->ncw == 4) return BL_PICK(bl_logp_kernel, p, true, 0, 4);
    if (staged && p->ncw == BL_CWAVES_SINGLE) return BL_PICK(bl_logp_kernel, p, true, 0, BL_CWAVES_SINGLE);
#ifdef BL_OCCU_CWX
    if (staged && p->ncw == BL_OCCU_CWX) return BL_PICK(bl_logp_kernel, p, true, 0, BL_OCCU_CWX);
#endif
    if (!staged && p->ncw == 4) return BL_PICK(bl_logp_kernel, p, false, 0, 4);
    return (int)hipErrorNotSupported;
}
