// kernels_inst.hip -- one translation unit per padded covariate capacity pair (BL_KS, BL_KO).
// Built N times by the Makefile (-DBL_KS=.. -DBL_KO=..) so the instantiations compile in parallel.
#include "logp_kernel.hpp"
#include "nuts_kernel.hpp"

#if !defined(BL_KS) || !defined(BL_KO)
#error "compile with -DBL_KS=<n> -DBL_KO=<n>"
#endif
#define BL_CAT3(a, b, c) a##_##b##_##c
#define BL_NAME(base, ks, ko) BL_CAT3(base, ks, ko)

template <typename K>
static hipError_t bl_set_lds(K kernel, int lds_bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}

extern "C" int BL_NAME(bl_launch_nuts, BL_KS, BL_KO)(const BlNutsParams *p, int grid, int lds_bytes, int staged, hipStream_t stream)
{
    hipError_t e;
    if (staged) {
        if ((e = bl_set_lds(bl_nuts_kernel<BL_KS, BL_KO, true>, lds_bytes)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL((bl_nuts_kernel<BL_KS, BL_KO, true>), dim3(grid), dim3(BL_THREADS), lds_bytes, stream, *p);
    } else {
        if ((e = bl_set_lds(bl_nuts_kernel<BL_KS, BL_KO, false>, lds_bytes)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL((bl_nuts_kernel<BL_KS, BL_KO, false>), dim3(grid), dim3(BL_THREADS), lds_bytes, stream, *p);
    }
    return (int)hipGetLastError();
}

extern "C" int BL_NAME(bl_launch_logp, BL_KS, BL_KO)(const BlLogpParams *p, int grid, int lds_bytes, int staged, hipStream_t stream)
{
    hipError_t e;
    if (staged) {
        if ((e = bl_set_lds(bl_logp_kernel<BL_KS, BL_KO, true>, lds_bytes)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL((bl_logp_kernel<BL_KS, BL_KO, true>), dim3(grid), dim3(BL_THREADS), lds_bytes, stream, *p);
    } else {
        if ((e = bl_set_lds(bl_logp_kernel<BL_KS, BL_KO, false>, lds_bytes)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL((bl_logp_kernel<BL_KS, BL_KO, false>), dim3(grid), dim3(BL_THREADS), lds_bytes, stream, *p);
    }
    return (int)hipGetLastError();
}
