// kernels_inst.hip -- one translation unit per padded covariate capacity pair (BL_KS, BL_KO).
// Built N times by the Makefile (-DBL_KS=.. -DBL_KO=..) so the instantiations compile in parallel.
// model 0 = occu (LDS-staged and HBM-row forms); model 1 = occu_rn and model 2 = occu with false
// positives (LDS-staged form, capacities <= 4).
#include "logp_kernel.hpp"
#include "nuts_kernel.hpp"

#if !defined(BL_KS) || !defined(BL_KO)
#error "compile with -DBL_KS=<n> -DBL_KO=<n>"
#endif
#define BL_CAT3(a, b, c) a##_##b##_##c
#define BL_NAME(base, ks, ko) BL_CAT3(base, ks, ko)
#if BL_KS <= 4 && BL_KO <= 4
#define BL_HAVE_RN 1
#else
#define BL_HAVE_RN 0
#endif

template <typename K, typename P>
static int bl_launch(K kernel, const P *p, int grid, int threads, int lds_bytes, hipStream_t stream)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds_bytes, stream, *p);
    return (int)hipGetLastError();
}

extern "C" int BL_NAME(bl_launch_nuts, BL_KS, BL_KO)(const BlNutsParams *p, int grid, int lds_bytes, int staged, int model, hipStream_t stream)
{
    if (model == 1) {
#if BL_HAVE_RN
        if (staged) return bl_launch(bl_nuts_kernel<BL_KS, BL_KO, true, 1>, p, grid, BL_THREADS_RN, lds_bytes, stream);
#endif
        return (int)hipErrorNotSupported;
    }
    if (model == 2) {
#if BL_HAVE_RN
        if (staged) return bl_launch(bl_nuts_kernel<BL_KS, BL_KO, true, 2>, p, grid, BL_THREADS, lds_bytes, stream);
#endif
        return (int)hipErrorNotSupported;
    }
    if (staged) return bl_launch(bl_nuts_kernel<BL_KS, BL_KO, true, 0>, p, grid, BL_THREADS, lds_bytes, stream);
    return bl_launch(bl_nuts_kernel<BL_KS, BL_KO, false, 0>, p, grid, BL_THREADS, lds_bytes, stream);
}

extern "C" int BL_NAME(bl_launch_logp, BL_KS, BL_KO)(const BlLogpParams *p, int grid, int lds_bytes, int staged, int model, hipStream_t stream)
{
    if (model == 1) {
#if BL_HAVE_RN
        if (staged) return bl_launch(bl_logp_kernel<BL_KS, BL_KO, true, 1>, p, grid, BL_THREADS_RN, lds_bytes, stream);
#endif
        return (int)hipErrorNotSupported;
    }
    if (model == 2) {
#if BL_HAVE_RN
        if (staged) return bl_launch(bl_logp_kernel<BL_KS, BL_KO, true, 2>, p, grid, BL_THREADS, lds_bytes, stream);
#endif
        return (int)hipErrorNotSupported;
    }
    if (staged) return bl_launch(bl_logp_kernel<BL_KS, BL_KO, true, 0>, p, grid, BL_THREADS, lds_bytes, stream);
    return bl_launch(bl_logp_kernel<BL_KS, BL_KO, false, 0>, p, grid, BL_THREADS, lds_bytes, stream);
}
