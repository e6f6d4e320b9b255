// logp_kernel.hpp -- K1: batched occupancy log-density + gradient (parity hook behind bl_logp_grad).
// Uses exactly the device functions the persistent NUTS kernel uses (same LDS records, same
// per-site arithmetic, same wave/workgroup reduction order), one launch per batch.
#pragma once
#include "occu_device.hpp"

struct BlLogpParams {
    BlDevData dd;
    int k, nloc, rec_stride;
    int max_abundance;   // occu_rn only
    int rn_off;          // occu_rn / dynamic occupancy: byte offset in LDS of its scratch (see BlNutsParams)
    int nmix_lds;        // MODEL 4: the table staged in LDS (see BlNutsParams)
    int lane_grp;        // lanes per site pair (see BlNutsParams)
    int fp_mode;         // false-positive coordinate (see BlNutsParams)
    const float *nmix_tab; // MODEL 4 (see BlNutsParams)
    int ncw;             // compute waves per workgroup: selects the CW instantiation (host side)
    int n_species, sp_lds; // joint-species datasets (see BlNutsParams)
    int B;
    const float *theta;  // [B][D] float32 view of the caller's double theta
    double *partial;     // [B][k][64]: c < D grad of log-lik, c == D log-lik
};

template <int KS, int KO, bool LDS, int MODEL, int CW>
__global__ void __launch_bounds__(64 * (CW + 1)) bl_logp_kernel(const BlLogpParams p)
{
    const int member = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Ks = p.dd.Ks, Ko = p.dd.Ko, nsp = p.n_species, Dsp = Ks + Ko + 2;
    const int D = bl_model_dim<MODEL>(Ks, Ko, p.fp_mode) + (nsp - 1) * Dsp;
    const int s0 = member * p.nloc;
    int cnt = p.dd.n_sites - s0;
    cnt = cnt < 0 ? 0 : (cnt > p.nloc ? p.nloc : cnt);
    const float *grows = nullptr;
    int ld = p.rec_stride;
    if constexpr (MODEL == 1) { // occu_rn: the order its records are staged in, and the waves' shares of the sites (rn_device.hpp)
        bl_rn_split_init<KS, KO, CW>(p.dd.rows, p.dd.n_stride, s0, cnt, p.dd.T, p.dd.J, p.rn_off);
        __syncthreads();
    }
    if constexpr (LDS) {
        for (int sp = 0; sp < nsp; sp++)
            bl_stage_records(p.dd.rows, p.dd.n_stride, s0, cnt, p.dd.T, p.dd.J, KS, bl_layout_ko<MODEL>(KO), p.rec_stride, 64 * (CW + 1), sp, sp * p.sp_lds, MODEL == 1 ? bl_rn_order<CW>(p.rn_off, cnt) : nullptr);
    } else {
        grows = p.dd.rows + s0;
        ld = p.dd.n_stride;
    }
    if constexpr (MODEL == 1) bl_rn_fill_lgamma(p.rn_off, p.max_abundance, 64 * (CW + 1)); // (the barrier below publishes it)
    if constexpr (MODEL == 4) {
        if (p.nmix_lds) bl_stage_nmix_tab(p.nmix_tab + s0, p.dd.n_stride, cnt, 2 * ((p.nloc + 1) / 2), p.dd.T * (p.max_abundance + 1), p.rn_off, 64 * (CW + 1));
    }
    float *sh_coef = bl_lds_f(BL_OFF_COEF);
    if (tid < 64) sh_coef[tid] = 0.0f;
    __syncthreads();
    // (same lane -> coefficient / partial mapping as the NUTS kernel's control wave)
    const bool has_phi = MODEL != 8 && D > nsp * Dsp;
    const int lsp = lane < nsp * Dsp ? lane / Dsp : 0, lj = lane - lsp * Dsp;
    int my_pos = lane < nsp * Dsp ? lsp * BL_SP_COEF(KS, KO) + bl_coef_pos(lj, Ks, Ko, KS, KO) : nsp * BL_SP_COEF(KS, KO) + 1;
    int part_pos = lane < nsp * Dsp ? lsp * BL_SP_PART(KS, KO) + bl_coef_pos(lj, Ks, Ko, KS, KO)
                                    : ((has_phi && lane == D - 1) ? KS + KO + 3 : KS + KO + 2);
    if constexpr (MODEL == 8) my_pos = part_pos = bl_dyn_pos(lane < D ? lane : D, Ks, Ko, KS, KO); // dynamic occupancy (dyn_device.hpp)
    const bool part_all = MODEL != 8 && lane >= nsp * Dsp;
    const int part_rs = nsp > 1 ? nsp * BL_SP_PART(KS, KO) : BL_PART_STRIDE;
    const int lane_grp = MODEL == 1 ? bl_rn_npos(p.rn_off) : p.lane_grp; // (occu_rn: its waves' shares of the sites, rn_device.hpp)
    for (int b = 0; b < p.B; b++) {
        if (wave == 0 && lane < D) sh_coef[my_pos] = p.theta[(size_t)b * D + lane];
        __syncthreads();
        if (wave > 0) { // compute waves, exactly as in the NUTS kernel
            bl_phase_a<KS, KO, LDS, MODEL, CW>(tid - 64, wave - 1, grows, ld, cnt, p.dd.T, p.dd.J, p.max_abundance, p.fp_mode, p.nmix_tab + s0, (MODEL == 4 && p.nmix_lds) ? 2 * ((p.nloc + 1) / 2) : p.dd.n_stride, nsp, p.sp_lds, p.rn_off, lane_grp, p.nmix_lds);
        }
        __syncthreads();
        if (wave == 0) {
            const float *part = bl_lds_f(BL_OFF_PART);
            double acc = 0.0;
#pragma unroll
            for (int w = 0; w < CW; w++) acc += (double)part[w * part_rs + part_pos];
            for (int sp = 1; sp < nsp; sp++)
                for (int w = 0; w < CW; w++) acc += part_all ? (double)part[w * part_rs + part_pos + sp * BL_SP_PART(KS, KO)] : 0.0;
            double *out = p.partial + ((size_t)b * p.k + member) * 64;
            if (lane <= D) out[lane] = acc;
        }
        __syncthreads();
    }
}
