// occu_device.hpp -- device-side building blocks for the gfx950 occupancy engine.
//
//  * per-site marginal log-likelihood + analytic gradient (closed form of
//    biolith/models/occu.py:136-242 with z summed out)
//  * wave64 reductions on DPP (no LDS traffic)
//  * xoshiro128++ streams
//
// Written for wave64 / CDNA4 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BL_THREADS 512
#define BL_WAVES (BL_THREADS / 64)
#define BL_MAX_DEPTH 10
#define BL_NSTREAM 64
#define BL_SCALAR_STREAM 63

// ---- dynamic LDS carve (bytes; every offset a multiple of 16: guide G17) ----
#define BL_OFF_THETA 0      // 64 floats : theta being evaluated (lane d = dim d)
#define BL_OFF_FLAG 256     // 4 ints    : loop control
#define BL_OFF_LL 272       // 8 doubles : per-wave log-lik partial
#define BL_OFF_PART 336     // 8 x 64 floats : per-wave gradient partials
#define BL_OFF_CKR 2384     // 10 x 64 floats: r checkpoints      (numpyro r_ckpts)
#define BL_OFF_CKRS 4944    // 10 x 64 floats: r_sum checkpoints  (numpyro r_sum_ckpts)
#define BL_OFF_DATA 7680    // staged site data starts here
#define BL_LDS_TOTAL 163840

extern __shared__ __attribute__((aligned(16))) unsigned char bl_smem_raw[];

__device__ __forceinline__ float *bl_lds_f(int byte_off) { return reinterpret_cast<float *>(bl_smem_raw + byte_off); }
__device__ __forceinline__ double *bl_lds_d(int byte_off) { return reinterpret_cast<double *>(bl_smem_raw + byte_off); }
__device__ __forceinline__ int *bl_lds_i(int byte_off) { return reinterpret_cast<int *>(bl_smem_raw + byte_off); }

// Dataset as it lives in HBM: one [rows][n_stride] float matrix, site index fastest
// (the reference's own plate order, occu.py:176-178), so lane <-> site loads coalesce.
//   rows [0, KS)                     x_k                      site covariates (NaN->0)
//   rows [KS, KS + V*(KO+1))         visit v: c, c*w_1..c*w_KO  c = +1 detection / -1 non-detection / 0 masked
//   next T rows                      ka = n_masked * ln2       (cancels the log sigma(0) of masked visits)
//   next T rows                      kb = n_detections * log(tiny_f32)   (z=0 branch, numpyro clamp)
struct BlDevData {
    const float *rows;
    int n_sites, n_stride, T, J;
    int Ks, Ko;     // actual covariate counts (theta layout)
    int KS, KO;     // padded counts the rows were packed for (= kernel template capacity)
    float loc_b, isc2_b, loc_a, isc2_a;  // Normal prior loc, 1/scale^2
    double prior_const;                  // sum_k log(scale_k) + D/2 log(2 pi)
};

// ------------------------------------------------------------------ math ----
#define BL_LOG2E 1.4426950408889634f
#define BL_LN2 0.6931471805599453f

// ------------------------------------------------------------ DPP helpers ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float bl_dpp(float x)
{
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double bl_dpp_d(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float bl_readlane(float x, int l)
{
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(x), l));
}
__device__ __forceinline__ double bl_readlane_d(double x, int l)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Sum over the 64 lanes of a wave; result is wave-uniform (read from lane 63).
// quad xor1, quad xor2, half-row mirror, row mirror -> every lane holds its 16-lane row sum;
// row_bcast15 / row_bcast31 fold the four rows into lane 63.  Fixed order => reproducible.
__device__ __forceinline__ float bl_wave_sum(float x)
{
    x += bl_dpp<0xB1, 0xF>(x);
    x += bl_dpp<0x4E, 0xF>(x);
    x += bl_dpp<0x141, 0xF>(x);
    x += bl_dpp<0x140, 0xF>(x);
    x += bl_dpp<0x142, 0xA>(x);
    x += bl_dpp<0x143, 0xC>(x);
    return bl_readlane(x, 63);
}
__device__ __forceinline__ double bl_wave_sum_d(double x)
{
    x += bl_dpp_d<0xB1, 0xF>(x);
    x += bl_dpp_d<0x4E, 0xF>(x);
    x += bl_dpp_d<0x141, 0xF>(x);
    x += bl_dpp_d<0x140, 0xF>(x);
    x += bl_dpp_d<0x142, 0xA>(x);
    x += bl_dpp_d<0x143, 0xC>(x);
    return bl_readlane_d(x, 63);
}

// N independent sums with the butterfly steps outermost: the N DPP chains interleave, so no
// s_nop padding between a VALU write and the DPP read of the same register is needed.
template <int N>
__device__ __forceinline__ void bl_wave_sum_vec(float (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += bl_dpp<0xB1, 0xF>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += bl_dpp<0x4E, 0xF>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += bl_dpp<0x141, 0xF>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += bl_dpp<0x140, 0xF>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += bl_dpp<0x142, 0xA>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += bl_dpp<0x143, 0xC>(v[i]);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = bl_readlane(v[i], 63);
}
__device__ __forceinline__ void bl_wave_sum2(float &a, float &b)
{
    float v[2] = {a, b};
    bl_wave_sum_vec<2>(v);
    a = v[0];
    b = v[1];
}

// ------------------------------------------------------------------- RNG ----
// xoshiro128++ 1.0; identical sequence to oracle/occu_oracle.c (tests compare them).
struct BlRng {
    uint32_t s0, s1, s2, s3;
};
__device__ __forceinline__ uint32_t bl_rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
__device__ __forceinline__ uint32_t bl_rng_next(BlRng &r)
{
    const uint32_t result = bl_rotl(r.s0 + r.s3, 7) + r.s0;
    const uint32_t t = r.s1 << 9;
    r.s2 ^= r.s0;
    r.s3 ^= r.s1;
    r.s1 ^= r.s2;
    r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = bl_rotl(r.s3, 11);
    return result;
}
// uniform in (0,1), 23 random bits + 1/2 ulp offset: exact in float32
__device__ __forceinline__ float bl_rng_uniform(BlRng &r)
{
    return ((float)(bl_rng_next(r) >> 9) + 0.5f) * (1.0f / 8388608.0f);
}
// Box-Muller, cosine branch (v_cos_f32 takes revolutions)
__device__ __forceinline__ float bl_rng_normal(BlRng &r)
{
    const float u1 = bl_rng_uniform(r), u2 = bl_rng_uniform(r);
    return sqrtf(-2.0f * __logf(u1)) * __builtin_amdgcn_cosf(u2);
}

// ------------------------------------------------- site log-lik + gradient ----
// Source accessor: staged slice in LDS (row stride = ld floats) or the HBM matrix directly.
template <bool LDS>
__device__ __forceinline__ float bl_ld(const float *__restrict__ grows, int row, int ld, int i)
{
    if constexpr (LDS)
        return bl_lds_f(BL_OFF_DATA)[row * ld + i];
    else
        return grows[(size_t)row * ld + i];
}

template <int KS, int KO, bool LDS, int JC>
__device__ __forceinline__ void bl_eval_sites_j(const float *__restrict__ grows, int ld, int cnt, int T, int J,
                                                const float (&beta)[KS + 1], const float (&alpha)[KO + 1],
                                                float &ll, float (&gb)[KS + 1], float (&ga)[KO + 1])
{
    // JC > 0: J == JC known at compile time -> the visit loop is fully unrolled, its LDS reads are
    // issued together and the JC independent exp/log/rcp chains interleave (latency, not issue,
    // bounds this phase at 2 waves per SIMD).  JC == 0: runtime J.
    const int Jn = JC > 0 ? JC : J;
    const int V = T * Jn;
    const int row_wc = KS, row_ka = KS + V * (KO + 1), row_kb = row_ka + T;
    for (int i = threadIdx.x; i < cnt; i += BL_THREADS) {
        float x[KS > 0 ? KS : 1];
        float eta = beta[0];
#pragma unroll
        for (int k = 0; k < KS; k++) {
            x[k] = bl_ld<LDS>(grows, k, ld, i);
            eta = fmaf(x[k], beta[k + 1], eta);
        }
        // softplus(eta), psi = sigmoid(eta)
        const float e_eta = __builtin_amdgcn_exp2f(-fabsf(eta) * BL_LOG2E);
        const float op_eta = 1.0f + e_eta;
        const float sp = fmaxf(eta, 0.0f) + BL_LN2 * __builtin_amdgcn_logf(op_eta);
        const float psi = (eta > 0.0f ? 1.0f : e_eta) * __builtin_amdgcn_rcpf(op_eta);
        float dsum = 0.0f;
        for (int t = 0; t < T; t++) {
            float a = bl_ld<LDS>(grows, row_ka + t, ld, i);
            float g[KO + 1];
#pragma unroll
            for (int k = 0; k <= KO; k++) g[k] = 0.0f;
            auto visit = [&](int j) {
                const int r0 = row_wc + (t * Jn + j) * (KO + 1);
                float w[KO + 1];
#pragma unroll
                for (int k = 0; k <= KO; k++) w[k] = bl_ld<LDS>(grows, r0 + k, ld, i);
                float u = w[0] * alpha[0];
#pragma unroll
                for (int k = 1; k <= KO; k++) u = fmaf(w[k], alpha[k], u);
                const float e = __builtin_amdgcn_exp2f(-fabsf(u) * BL_LOG2E);
                const float op = 1.0f + e;
                a += fminf(u, 0.0f) - BL_LN2 * __builtin_amdgcn_logf(op);
                const float s = (u > 0.0f ? e : 1.0f) * __builtin_amdgcn_rcpf(op);
#pragma unroll
                for (int k = 0; k <= KO; k++) g[k] = fmaf(s, w[k], g[k]);
            };
            if constexpr (JC > 0) {
#pragma unroll
                for (int j = 0; j < JC; j++) visit(j);
            } else {
#pragma unroll 4
                for (int j = 0; j < Jn; j++) visit(j);
            }
            const float kb = bl_ld<LDS>(grows, row_kb + t, ld, i);
            // z=1 branch A = log psi + a ; z=0 branch B = log(1-psi) + n_det log(tiny)
            const float A = eta - sp + a, B = kb - sp;
            const float d = eta + a - kb; // = A - B
            const float e_d = __builtin_amdgcn_exp2f(-fabsf(d) * BL_LOG2E);
            const float op_d = 1.0f + e_d;
            ll += fmaxf(A, B) + BL_LN2 * __builtin_amdgcn_logf(op_d);
            const float q = (d > 0.0f ? 1.0f : e_d) * __builtin_amdgcn_rcpf(op_d); // P(z=1 | y, theta)
            dsum += q - psi;
#pragma unroll
            for (int k = 0; k <= KO; k++) ga[k] = fmaf(q, g[k], ga[k]);
        }
        gb[0] += dsum;
#pragma unroll
        for (int k = 0; k < KS; k++) gb[k + 1] = fmaf(dsum, x[k], gb[k + 1]);
    }
}

// Accumulates, over this thread's sites i = tid, tid+BL_THREADS, ... < cnt :
//   ll     += sum_t l_it                        (log-lik, z marginalised)
//   gb[k]  += d ll / d beta_k ,  ga[k] += d ll / d alpha_k
// Per visit (all in f32, stable forms):  u = c*alpha0 + sum_k (c w_k) alpha_k ,
//   log sigma(u) = min(u,0) - log(1+e^-|u|),  sigma(-u) = (u>0 ? e : 1)/(1+e),  e = e^-|u|
template <int KS, int KO, bool LDS>
__device__ __forceinline__ void bl_eval_sites(const float *__restrict__ grows, int ld, int cnt, int T, int J,
                                              const float (&beta)[KS + 1], const float (&alpha)[KO + 1],
                                              float &ll, float (&gb)[KS + 1], float (&ga)[KO + 1])
{
    switch (J) { // wave-uniform
    case 1: bl_eval_sites_j<KS, KO, LDS, 1>(grows, ld, cnt, T, J, beta, alpha, ll, gb, ga); break;
    case 2: bl_eval_sites_j<KS, KO, LDS, 2>(grows, ld, cnt, T, J, beta, alpha, ll, gb, ga); break;
    case 3: bl_eval_sites_j<KS, KO, LDS, 3>(grows, ld, cnt, T, J, beta, alpha, ll, gb, ga); break;
    case 4: bl_eval_sites_j<KS, KO, LDS, 4>(grows, ld, cnt, T, J, beta, alpha, ll, gb, ga); break;
    case 5: bl_eval_sites_j<KS, KO, LDS, 5>(grows, ld, cnt, T, J, beta, alpha, ll, gb, ga); break;
    case 6: bl_eval_sites_j<KS, KO, LDS, 6>(grows, ld, cnt, T, J, beta, alpha, ll, gb, ga); break;
    case 8: bl_eval_sites_j<KS, KO, LDS, 8>(grows, ld, cnt, T, J, beta, alpha, ll, gb, ga); break;
    default: bl_eval_sites_j<KS, KO, LDS, 0>(grows, ld, cnt, T, J, beta, alpha, ll, gb, ga); break;
    }
}

// Copy this workgroup's site slice [s0, s0+cnt) of every row into LDS (row stride ld).
__device__ __forceinline__ void bl_stage_rows(const float *__restrict__ rows, int n_rows, int n_stride, int s0, int cnt, int ld)
{
    float *dst = bl_lds_f(BL_OFF_DATA);
    for (int r = 0; r < n_rows; r++) {
        const float *src = rows + (size_t)r * n_stride + s0;
        for (int i = threadIdx.x; i < ld; i += BL_THREADS) dst[r * ld + i] = (i < cnt) ? src[i] : 0.0f;
    }
}

// theta (D floats in LDS, lane-d order) -> padded coefficient registers.
template <int KS, int KO>
__device__ __forceinline__ void bl_load_coefs(const float *th, int Ks, int Ko, float (&beta)[KS + 1], float (&alpha)[KO + 1])
{
#pragma unroll
    for (int k = 0; k <= KS; k++) beta[k] = (k <= Ks) ? th[k] : 0.0f;
#pragma unroll
    for (int k = 0; k <= KO; k++) alpha[k] = (k <= Ko) ? th[Ks + 1 + k] : 0.0f;
}

// Workgroup reduction of (ll, gb, ga) -> LDS per-wave partials.  Component layout (lane index in
// the control wave): c in [0, Ks] dll/dbeta, [Ks+1, D) dll/dalpha.  ll is kept in double.
template <int KS, int KO>
__device__ __forceinline__ void bl_wave_partials_to_lds(int Ks, int Ko, float ll, const float (&gb)[KS + 1], const float (&ga)[KO + 1])
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float v[KS + KO + 2];
#pragma unroll
    for (int k = 0; k <= KS; k++) v[k] = gb[k];
#pragma unroll
    for (int k = 0; k <= KO; k++) v[KS + 1 + k] = ga[k];
    bl_wave_sum_vec<KS + KO + 2>(v);
    const double llw = bl_wave_sum_d((double)ll);
    float *part = bl_lds_f(BL_OFF_PART) + wave * 64;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k <= KS; k++)
            if (k <= Ks) part[k] = v[k];
#pragma unroll
        for (int k = 0; k <= KO; k++)
            if (k <= Ko) part[Ks + 1 + k] = v[KS + 1 + k];
        bl_lds_d(BL_OFF_LL)[wave] = llw;
    }
}
