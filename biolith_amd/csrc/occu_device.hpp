// occu_device.hpp -- device-side building blocks for the gfx950 occupancy engine.
//
//  * per-site marginal log-likelihood + analytic gradient (closed form of
//    biolith/models/occu.py:136-242 with z summed out)
//  * wave64 reductions on DPP (no LDS traffic)
//  * xoshiro128++ streams
//
// Written for wave64 / CDNA4 only.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

// Workgroup = 1 control wave (wave 0: exchange + NUTS state machine) + CW compute waves that
// evaluate the sites; CW is a template parameter of the kernels (blockDim.x = 64 * (CW + 1)):
// 3 -> every wave has a SIMD to itself, so the control wave's decisions, which overlap the site
//      evaluation, do not steal issue slots from a compute wave, and one wave per SIMD may use the
//      whole 512-entry register file (the latency optimum while one site pair per lane still fits:
//      N <= 384 sites per workgroup);
// 4 -> all four SIMDs evaluate sites (larger slices, where the evaluation dominates the tick);
// BL_CWAVES_RN -> occu_rn (rn_device.hpp): several waves per SIMD hide each other's LDS / exp / log / rcp latencies.
// 7 (BL_CWAVES_SINGLE) -> a chain of ONE workgroup (small problems -- simulate()'s defaults, the reference's own test sizes): all of a CU's
//      eight wave slots at two per SIMD, so that 448 lanes share the site pairs (lane groups) and nothing leaves the CU: no exchange.
#define BL_CWAVES_SINGLE 7
#define BL_CWAVES_MAX 4   // (of the kernels that sample several species jointly: sizes their partial table)
#define BL_DIR_STREAM 62
#define BL_MAX_DEPTH 10
#define BL_NSTREAM 64
#define BL_SCALAR_STREAM 63

// ---- dynamic LDS carve (bytes; every offset a multiple of 16: guide G17) ----
#define BL_OFF_COEF 0       // 64 floats : coefficients being evaluated, PADDED layout (beta[0..KS], alpha[0..KO])
#define BL_OFF_FLAG 256     // 4 ints    : loop control
#define BL_OFF_TAG 272      // 16 ints   : per compute wave, the evaluation (epoch) its row of the partial table belongs to
#define BL_OFF_PART 336     // <= 16 compute waves x BL_PART_STRIDE floats : per-wave partial sums (padded layout + log-lik)
#define BL_PART_STRIDE 48
#define BL_OFF_CKR 3408     // 10 x 64 floats: r checkpoints      (numpyro r_ckpts)
#define BL_OFF_CKRS 5968    // 10 x 64 floats: r_sum checkpoints  (numpyro r_sum_ckpts)
#define BL_OFF_SV 8528      // 16 x 64 floats: control wave's rarely-touched per-dimension state (tree edges, proposal, ...)
#define BL_OFF_SS 12624     // 256 bytes     : control wave's rarely-touched scalars (BlCtlScalars)
#define BL_OFF_DATA 12880   // staged site records start here
#define BL_LDS_TOTAL 163840

extern __shared__ __attribute__((aligned(16))) unsigned char bl_smem_raw[];

__device__ __forceinline__ float *bl_lds_f(int byte_off) { return reinterpret_cast<float *>(bl_smem_raw + byte_off); }
__device__ __forceinline__ int *bl_lds_i(int byte_off) { return reinterpret_cast<int *>(bl_smem_raw + byte_off); }

// Dataset as it lives in HBM: one [rows][n_stride] float matrix, site index fastest
// (the reference's own plate order, occu.py:176-178), so lane <-> site loads coalesce.
//   rows [0, KS)                     x_k                      site covariates (NaN->0)
//   rows [KS, KS + V*(KO+1))         visit v: c, c*w_1..c*w_KO  c = +1 detection / -1 non-detection / 0 masked
//   next T rows                      ka = n_masked * ln2       (cancels the log sigma(0) of masked visits)
//   next T rows                      kb = n_detections * log(tiny_f32)   (z=0 branch, numpyro clamp)
//
// In LDS a workgroup keeps its sites as RECORDS (array of structures) so that a lane reads its data
// with a few ds_read_b128 at immediate offsets and no address arithmetic.  One site's elements:
//   [ x_0..x_KS-1 | pad to 4 ]  then per period t a block of pb floats:
//   [ visit 0: c, c*w_1..c*w_KO | visit 1 | ... | visit J-1 | ka | kb | pad to 4 ]
// and two consecutive sites are interleaved element by element into one PAIR record
// (a0 b0 a1 b1 ...), because a lane evaluates two sites with packed f32 math (see bl_eval_sites_lds).
// Pair stride = 4 * (odd number) floats: 16-byte aligned and ds_read_b128 conflict-free
// (any 16 lanes that are distinct mod 16 hit 16 distinct 4-bank slots).
struct BlDevData {
    const float *rows;
    int n_sites, n_stride, T, J;
    int Ks, Ko;     // actual covariate counts (theta layout)
    int KS, KO;     // padded counts the rows were packed for (= kernel template capacity)
    float loc_b, isc2_b, loc_a, isc2_a;  // Normal prior loc, 1/scale^2 (0 for a Laplace prior)
    float l1_b, l1_a;                    // Laplace(loc, scale) prior: 1/scale (0 for a Normal prior); energy = dth^2 isc2 / 2 + |dth| l1
    double prior_const;                  // sum_k log(scale_k) + D/2 log(2 pi)  (+ log B(a,b) with has_fp)
    int n_species;                       // species sampled jointly (theta = [species 0: beta, alpha | species 1: ... | (phi)]); 0 reads as 1
    int dyn;                             // 1: the dynamic occupancy model (theta = [b_psi | b_gamma | b_eps | alpha], dyn_device.hpp)
    int has_fp;                          // 0, or the model id (2: logit rate, Beta prior; 3: log rate, Exponential prior)
                                         // whose false-positive coordinate phi is theta's last entry
    float fp_a, fp_b;                    // its prior: Beta(a, b) / Exponential(rate = a)
};

__host__ __device__ inline int bl_round4(int x) { return (x + 3) & ~3; }
// floats per period block / per site record (host and device must agree)
__host__ __device__ inline int bl_period_block(int J, int KO) { return bl_round4(J * (KO + 1) + 2); }
// floats per PAIR record (two sites interleaved element-wise)
__host__ __device__ inline int bl_record_stride(int T, int J, int KS, int KO)
{
    int q = (bl_round4(KS) + T * bl_period_block(J, KO)) / 2;
    if ((q & 1) == 0) q++;
    return 4 * q;
}

// ------------------------------------------------------------------ math ----
#define BL_LOG2E 1.4426950408889634f
#define BL_LN2 0.6931471805599453f

__device__ __forceinline__ float bl_exp_f(float x) { return __builtin_amdgcn_exp2f(x * BL_LOG2E); }

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

// x[l] + x[l ^ 16] in every lane (rows 0+1 and 2+3 of 16 lanes), and x[l] + x[l ^ 32], for doubles.
// v_permlane16_swap(a, b): odd rows of a <-> even rows of b; with a = b = x the two results are
// {r0, r0, r2, r2} and {r1, r1, r3, r3}.  v_permlane32_swap likewise for the 32-lane halves.
__device__ __forceinline__ double bl_fold_rows16_d(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __longlong_as_double((long long)(((unsigned long long)h[0] << 32) | l[0])) +
           __longlong_as_double((long long)(((unsigned long long)h[1] << 32) | l[1]));
}
__device__ __forceinline__ double bl_fold_halves32_d(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __longlong_as_double((long long)(((unsigned long long)h[0] << 32) | l[0])) +
           __longlong_as_double((long long)(((unsigned long long)h[1] << 32) | l[1]));
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

// x += dpp(x) as ONE instruction (v_add_f32_dpp).  Written in assembly because the compiler's SLP
// vectoriser otherwise pairs the adds of neighbouring values into v_pk_add_f32 and feeds them through
// v_mov_b32 + v_mov_b32_dpp copies: 5 instructions (24 issue cycles) per pair and level instead of 2 (8).
// The assembler does not see through inline asm, so the caller guarantees the DPP read-after-VALU-write
// distance (2 wait states): interleave >= 3 independent chains, or put BL_DPP_GAP between levels.
// For the row_bcast steps the rows masked out by row_mask keep x (= x + 0), which is what a fold wants.
#define BL_DPP_ADD(x, MOD) asm volatile("v_add_f32_dpp %0, %0, %0 " MOD : "+v"(x))
#define BL_DPP_GAP asm volatile("s_nop 1")
#define BL_DPP_XOR1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define BL_DPP_XOR2 "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define BL_DPP_HALF "row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define BL_DPP_ROW "row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define BL_DPP_B15 "row_bcast:15 row_mask:0xa bank_mask:0xf"
#define BL_DPP_B31 "row_bcast:31 row_mask:0xc bank_mask:0xf"

// N independent sums with the butterfly steps outermost: the N DPP chains interleave, so no
// padding between a VALU write and the DPP read of the same register is needed (N >= 3).
// On return lane 63 of every v[i] holds the wave total (other lanes: partial sums).
template <int N>
__device__ __forceinline__ void bl_wave_sum_vec_l63(float (&v)[N])
{
    if constexpr (N >= 3) {
#pragma unroll
        for (int i = 0; i < N; i++) asm volatile("" : "+v"(v[i])); // (pinned: see bl_low_sum2)
        BL_DPP_GAP; // the values were just produced by ordinary VALU code
#pragma unroll
        for (int i = 0; i < N; i++) BL_DPP_ADD(v[i], BL_DPP_XOR1);
#pragma unroll
        for (int i = 0; i < N; i++) BL_DPP_ADD(v[i], BL_DPP_XOR2);
#pragma unroll
        for (int i = 0; i < N; i++) BL_DPP_ADD(v[i], BL_DPP_HALF);
#pragma unroll
        for (int i = 0; i < N; i++) BL_DPP_ADD(v[i], BL_DPP_ROW);
#pragma unroll
        for (int i = 0; i < N; i++) BL_DPP_ADD(v[i], BL_DPP_B15);
#pragma unroll
        for (int i = 0; i < N; i++) BL_DPP_ADD(v[i], BL_DPP_B31);
        BL_DPP_GAP;
    } else {
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
    }
}
// ---- NV <= 16 wave sums as a REDUCE-SCATTER: each total ends up in one lane group instead of all of them in lane 63 ----
// The butterfly above spends NV adds on each of its six levels (54 for the 9 values of the (3, 3) kernels).  Here every level halves
// the values a lane still carries -- the lane groups a level separates keep different halves:
//   1. row_half_mirror (banks 0<->1, 2<->3 of a row): banks 0, 2 keep v[0, H1), banks 1, 3 keep v[H1, NV)      NV  masked DPP adds
//   2. row_ror:8       (banks 0<->2, 1<->3):          banks 0, 1 keep the first H2 of theirs, banks 2, 3 the rest  H1
//   3. v_permlane16_swap (rows 0<->1, 2<->3): a swap + an add folds TWO values, even rows keep one, odd rows the other   ~H2
//   4. v_permlane32_swap (the wave's halves): likewise                                                              2
//   5. the two quad levels on the one value left                                                                     2
// -- 23 instead of 54 for NV = 9, and one LDS store per lane group instead of nine from lane 63.  (bank_mask selects groups of four
// lanes, which is why the quad levels come last.)  BlScatter<NV>::slot(bank, row): the index of the total the lanes of a (bank, row) hold,
// -1 where nothing is to be stored; lanes a level does not write keep stale values, which stay inside their own (bank, value) slots.
template <int NV>
struct BlScatter {
    static_assert(NV >= 3 && NV <= 16, "reduce-scatter form: 3..16 values");
    static constexpr int H1 = (NV + 1) / 2, H2 = (H1 + 1) / 2, H3 = (H2 + 1) / 2;
    static constexpr int slot(int b, int r)
    {
        const int c1 = b & 1, c2 = b >> 1, rp = r & 1, hh = r >> 1;
        const int xi = hh;                       // level 4: the lower half keeps x0, the upper x1 (a single x: the lower half stores it)
        if (xi >= H3) return -1;
        int ui = 2 * xi + rp;                    // level 3: even rows keep u[2 xi], odd rows u[2 xi + 1] (a single u: the even row stores it)
        if (2 * xi + 1 >= H2) { if (rp) return -1; ui = 2 * xi; }
        if (ui >= (c2 ? H1 - H2 : H2)) return -1;
        const int wj = (c2 ? H2 : 0) + ui;       // level 2
        if (wj >= (c1 ? NV - H1 : H1)) return -1;
        return (c1 ? H1 : 0) + wj;               // level 1
    }
    static constexpr unsigned long long slots() // 16 x 4 bits, indexed by (row << 2) | bank = lane >> 2
    {
        unsigned long long t = 0;
        for (int g = 0; g < 16; g++) { const int k = slot(g & 3, g >> 2); t |= (unsigned long long)(k < 0 ? 0 : k) << (4 * g); }
        return t;
    }
    static constexpr unsigned stores()          // bit g: the lane group g = lane >> 2 stores
    {
        unsigned m = 0;
        for (int g = 0; g < 16; g++) m |= (slot(g & 3, g >> 2) >= 0 ? 1u : 0u) << g;
        return m;
    }
};
#define BL_DPP_ADD_TO(dst, src, MOD) asm volatile("v_add_f32_dpp %0, %1, %1 " MOD : "+v"(dst) : "v"(src))
// On return: the lane holds the wave total of value BlScatter<NV>::slot(bank, row) (all four lanes of its bank alike).
template <int NV>
__device__ __forceinline__ float bl_wave_reduce_scatter(const float (&v)[NV])
{
    using S = BlScatter<NV>;
    float t[NV], w[S::H1], u[S::H2];
    // The assembler does not see through inline asm, so the DPP read-after-VALU-write distance (2 wait states) is this code's to keep:
    // every input is pinned to a register HERE (the compiler otherwise sinks the instruction that produces it to just in front of the
    // DPP add that reads it), then one gap; inside a level the adds read registers nothing writes any more.
#pragma unroll
    for (int i = 0; i < NV; i++) { t[i] = v[i]; asm volatile("" : "+v"(t[i])); }
#pragma unroll
    for (int i = 0; i < S::H1; i++) w[i] = t[i];
    BL_DPP_GAP;
#pragma unroll
    for (int i = 0; i < S::H1; i++) BL_DPP_ADD_TO(w[i], t[i], "row_half_mirror row_mask:0xf bank_mask:0x5");
#pragma unroll
    for (int i = 0; i < NV - S::H1; i++) BL_DPP_ADD_TO(w[i], t[S::H1 + i], "row_half_mirror row_mask:0xf bank_mask:0xa");
    BL_DPP_GAP;
#pragma unroll
    for (int i = 0; i < S::H2; i++) u[i] = w[i];
#pragma unroll
    for (int i = 0; i < S::H2; i++) BL_DPP_ADD_TO(u[i], w[i], "row_ror:8 row_mask:0xf bank_mask:0x3");
#pragma unroll
    for (int i = 0; i < S::H1 - S::H2; i++) BL_DPP_ADD_TO(u[i], w[S::H2 + i], "row_ror:8 row_mask:0xf bank_mask:0xc");
    BL_DPP_GAP;
    float x[S::H3];
#pragma unroll
    for (int i = 0; i < S::H3; i++) {
        const unsigned a = __float_as_uint(u[2 * i]), b = __float_as_uint(u[2 * i + 1 < S::H2 ? 2 * i + 1 : 2 * i]);
        const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        x[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[0]), __float_as_uint(x[S::H3 > 1 ? 1 : 0]), false, false);
    float y = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    y += bl_dpp<0xB1, 0xF>(y);
    y += bl_dpp<0x4E, 0xF>(y);
    return y;
}

__device__ __forceinline__ void bl_wave_sum2(float &a, float &b)
{
    float v[2] = {a, b};
    bl_wave_sum_vec_l63<2>(v);
    a = bl_readlane(v[0], 63);
    b = bl_readlane(v[1], 63);
}
// Two sums over lanes [0, n) only (all other lanes hold zeros): the butterfly stops as soon as
// lane 0's group covers n lanes; totals are read from lane 0.  n is wave-uniform.
__device__ __forceinline__ void bl_low_sum2(float &a, float &b, int n)
{
    // two chains only: a gap between levels provides the DPP read-after-write distance.  (The inputs are pinned first: without it
    // the compiler may sink the instruction producing one to behind the gap, next to the DPP add that reads it.)
    asm volatile("" : "+v"(a), "+v"(b));
    BL_DPP_GAP; BL_DPP_ADD(a, BL_DPP_XOR1); BL_DPP_ADD(b, BL_DPP_XOR1);
    BL_DPP_GAP; BL_DPP_ADD(a, BL_DPP_XOR2); BL_DPP_ADD(b, BL_DPP_XOR2);
    if (n > 4) { BL_DPP_GAP; BL_DPP_ADD(a, BL_DPP_HALF); BL_DPP_ADD(b, BL_DPP_HALF); }  // 8 lanes
    if (n > 8) { BL_DPP_GAP; BL_DPP_ADD(a, BL_DPP_ROW); BL_DPP_ADD(b, BL_DPP_ROW); }    // 16 lanes
    if (n > 16) {
        // rows 1..3 still to fold: the generic steps, totals in lane 63
        BL_DPP_GAP; BL_DPP_ADD(a, BL_DPP_B15); BL_DPP_ADD(b, BL_DPP_B15);
        BL_DPP_GAP; BL_DPP_ADD(a, BL_DPP_B31); BL_DPP_ADD(b, BL_DPP_B31);
        BL_DPP_GAP;
        a = bl_readlane(a, 63); b = bl_readlane(b, 63);
        return;
    }
    BL_DPP_GAP;
    a = bl_readlane(a, 0); b = bl_readlane(b, 0);
}

// The same for n <= 8 known beforehand (the (<= 3, <= 3)-covariate kernels of one species): three levels, no branches on n.
__device__ __forceinline__ void bl_low_sum2_w8(float &a, float &b)
{
    asm volatile("" : "+v"(a), "+v"(b));
    BL_DPP_GAP; BL_DPP_ADD(a, BL_DPP_XOR1); BL_DPP_ADD(b, BL_DPP_XOR1);
    BL_DPP_GAP; BL_DPP_ADD(a, BL_DPP_XOR2); BL_DPP_ADD(b, BL_DPP_XOR2);
    BL_DPP_GAP; BL_DPP_ADD(a, BL_DPP_HALF); BL_DPP_ADD(b, BL_DPP_HALF);
    BL_DPP_GAP;
    a = bl_readlane(a, 0); b = bl_readlane(b, 0);
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
    return __builtin_amdgcn_sqrtf(-2.0f * BL_LN2 * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
}

// ------------------------------------------------- site log-lik + gradient ----
// One visit (all f32, stable forms):  u = c*alpha0 + sum_k (c w_k) alpha_k ,
//   log sigma(u) = min(u,0) - log(1+e^-|u|),  sigma(-u) = (u>0 ? e : 1)/(1+e),  e = e^-|u|
template <int KO>
__device__ __forceinline__ void bl_visit(const float (&w)[KO + 1], const float (&alpha)[KO + 1], float &a, float (&g)[KO + 1])
{
    float u = w[0] * alpha[0];
#pragma unroll
    for (int k = 1; k <= KO; k++) u = fmaf(w[k], alpha[k], u);
    const float e = __builtin_amdgcn_exp2f(-fabsf(u) * BL_LOG2E);
    const float op = 1.0f + e;
    a += fminf(u, 0.0f) - BL_LN2 * __builtin_amdgcn_logf(op);
    const float s = (u > 0.0f ? e : 1.0f) * __builtin_amdgcn_rcpf(op);
#pragma unroll
    for (int k = 0; k <= KO; k++) g[k] = fmaf(s, w[k], g[k]);
}

// softplus(eta) and psi = sigmoid(eta)
__device__ __forceinline__ void bl_site_head(float eta, float &sp, float &psi)
{
    const float e_eta = __builtin_amdgcn_exp2f(-fabsf(eta) * BL_LOG2E);
    const float op_eta = 1.0f + e_eta;
    sp = fmaxf(eta, 0.0f) + BL_LN2 * __builtin_amdgcn_logf(op_eta);
    psi = (eta > 0.0f ? 1.0f : e_eta) * __builtin_amdgcn_rcpf(op_eta);
}

// Sum over z of one (site, period): z=1 branch A = log psi + a ; z=0 branch B = log(1-psi) + n_det log(tiny)
template <int KO>
__device__ __forceinline__ void bl_period_tail(float eta, float sp, float psi, float a, float kb, const float (&g)[KO + 1],
                                               float &ll, float &dsum, float (&ga)[KO + 1])
{
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

// ---- packed two-sites-per-lane forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) ----
// Non-packed f32 VALU ops issue at 4 cycles per wave64 on a CDNA SIMD and one site per lane needs
// ~160 of them, which made this phase VALU-issue-bound with two waves per SIMD.  Each lane therefore
// evaluates TWO sites: .x / .y of every value belong to the two sites of a pair, FMA/mul/add run
// packed (one issue slot for both sites), only exp/log/rcp/min/max/select are per component.
typedef float bl_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bl_f2 bl2(float a) { return bl_f2{a, a}; }
__device__ __forceinline__ bl_f2 bl_fma2(bl_f2 a, bl_f2 b, bl_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ bl_f2 bl_exp2_2(bl_f2 x) { return bl_f2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
__device__ __forceinline__ bl_f2 bl_log2_2(bl_f2 x) { return bl_f2{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)}; }
__device__ __forceinline__ bl_f2 bl_rcp_2(bl_f2 x) { return bl_f2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
// (u > 0 ? 1 : e)
__device__ __forceinline__ bl_f2 bl_sel_pos_one(bl_f2 u, bl_f2 e) { return bl_f2{u.x > 0.0f ? 1.0f : e.x, u.y > 0.0f ? 1.0f : e.y}; }
// (u > 0 ? e : 1)
__device__ __forceinline__ bl_f2 bl_sel_pos_e(bl_f2 u, bl_f2 e) { return bl_f2{u.x > 0.0f ? e.x : 1.0f, u.y > 0.0f ? e.y : 1.0f}; }

// exp(-max(u, -87)): the clamp keeps e^-u finite in f32; beyond |u| = 87 the log-lik goes flat,
// far outside anything a posterior visits (numpyro's own clamp_probs flattens it at 15.9 / 87.3).
__device__ __forceinline__ bl_f2 bl_expneg_2(bl_f2 u)
{
    return bl_exp2_2(__builtin_elementwise_max(u, bl2(-87.0f)) * bl2(-BL_LOG2E));
}

// One visit of both sites:  u = c*alpha0 + sum_k (c w_k) alpha_k ,  t = e^-u ,
//   log sigma(u) = -log(1+t) ,  sigma(-u) = t/(1+t)     (no branches, no selects)
template <int KO>
__device__ __forceinline__ void bl_visit2(const bl_f2 (&w)[KO + 1], const float (&alpha)[KO + 1], bl_f2 &a, bl_f2 (&g)[KO + 1])
{
    bl_f2 u = w[0] * bl2(alpha[0]);
#pragma unroll
    for (int k = 1; k <= KO; k++) u = bl_fma2(w[k], bl2(alpha[k]), u);
    const bl_f2 t = bl_expneg_2(u);
    const bl_f2 op = t + bl2(1.0f);
    a = bl_fma2(bl_log2_2(op), bl2(-BL_LN2), a);
    const bl_f2 s = t * bl_rcp_2(op);
#pragma unroll
    for (int k = 0; k <= KO; k++) g[k] = bl_fma2(s, w[k], g[k]);
}

// Accumulates, over this thread's site PAIRS m = ct, ct+CT, ... (CT compute threads):
//   ll += sum_t l_it ,  gb[k] += d ll / d beta_k ,  ga[k] += d ll / d alpha_k
// LDS pair records (element e of the two sites adjacent); JC > 0: J == JC at compile time -> the
// whole period block is read with ds_read_b128 at immediate offsets and the visits are unrolled
// (their exp/log/rcp chains interleave).  JC == 0: runtime J.
// ONE1 (the sampler's lean instantiations): one period and at most one pair per lane -- no loops, straight-line code
template <int KS, int KO, int JC, int CT, bool ONE1 = false>
__device__ __forceinline__ void bl_eval_sites_lds(int ct, int pstride, int cnt, int T, int J,
                                                  const float (&beta)[KS + 1], const float (&alpha)[KO + 1],
                                                  float &ll, float (&gb)[KS + 1], float (&ga)[KO + 1], int data_off = 0)
{
    constexpr int XQ = (KS + 3) & ~3;
    const int Jn = JC > 0 ? JC : J;
    const int pb = bl_period_block(Jn, KO);
    const float *data = bl_lds_f(BL_OFF_DATA) + data_off; // (data_off: the records of one species of a joint-species dataset)
    const int npairs = (cnt + 1) >> 1;
    bl_f2 ll2 = bl2(0.0f), gb2[KS + 1], ga2[KO + 1];
#pragma unroll
    for (int k = 0; k <= KS; k++) gb2[k] = bl2(0.0f);
#pragma unroll
    for (int k = 0; k <= KO; k++) ga2[k] = bl2(0.0f);
    auto one_pair = [&](int m) {
        const float4 *rec = reinterpret_cast<const float4 *>(data + (size_t)m * pstride);
        const bl_f2 vmask = bl_f2{1.0f, (2 * m + 1 < cnt) ? 1.0f : 0.0f}; // odd slice: the last pair's second site is a dummy
        bl_f2 x[XQ];
#pragma unroll
        for (int q = 0; q < XQ / 2; q++) {
            const float4 v = rec[q];
            x[2 * q] = bl_f2{v.x, v.y};
            x[2 * q + 1] = bl_f2{v.z, v.w};
        }
        bl_f2 eta = bl2(beta[0]);
#pragma unroll
        for (int k = 0; k < KS; k++) eta = bl_fma2(x[k], bl2(beta[k + 1]), eta);
        // softplus(eta), psi = sigmoid(eta)   (exact forms: once per site)
        const bl_f2 e_eta = bl_exp2_2(__builtin_elementwise_abs(eta) * bl2(-BL_LOG2E));
        const bl_f2 op_eta = e_eta + bl2(1.0f);
        const bl_f2 sp = bl_fma2(bl_log2_2(op_eta), bl2(BL_LN2), __builtin_elementwise_max(eta, bl2(0.0f)));
        const bl_f2 psi = bl_sel_pos_one(eta, e_eta) * bl_rcp_2(op_eta);
        bl_f2 dsum = bl2(0.0f), lsite = bl2(0.0f);
        auto one_period = [&](int t) {
            bl_f2 g[KO + 1];
#pragma unroll
            for (int k = 0; k <= KO; k++) g[k] = bl2(0.0f);
            bl_f2 a, kb;
            if constexpr (JC > 0) {
                constexpr int PBC = (JC * (KO + 1) + 2 + 3) & ~3;
                const float4 *pq = rec + XQ / 2 + t * (PBC / 2);
                bl_f2 blk[PBC];
#pragma unroll
                for (int q = 0; q < PBC / 2; q++) {
                    const float4 v = pq[q];
                    blk[2 * q] = bl_f2{v.x, v.y};
                    blk[2 * q + 1] = bl_f2{v.z, v.w};
                }
                a = blk[JC * (KO + 1)];
                kb = blk[JC * (KO + 1) + 1];
#pragma unroll
                for (int j = 0; j < JC; j++) {
                    bl_f2 w[KO + 1];
#pragma unroll
                    for (int k = 0; k <= KO; k++) w[k] = blk[j * (KO + 1) + k];
                    bl_visit2<KO>(w, alpha, a, g);
                }
            } else {
                const float2 *pp = reinterpret_cast<const float2 *>(data + (size_t)m * pstride) + XQ + t * pb;
                const float2 a_ = pp[Jn * (KO + 1)], kb_ = pp[Jn * (KO + 1) + 1];
                a = bl_f2{a_.x, a_.y};
                kb = bl_f2{kb_.x, kb_.y};
#pragma unroll 2
                for (int j = 0; j < Jn; j++) {
                    bl_f2 w[KO + 1];
#pragma unroll
                    for (int k = 0; k <= KO; k++) {
                        const float2 v = pp[j * (KO + 1) + k];
                        w[k] = bl_f2{v.x, v.y};
                    }
                    bl_visit2<KO>(w, alpha, a, g);
                }
            }
            // sum over z: z=1 branch A = log psi + a ; z=0 branch B = log(1-psi) + n_det log(tiny)
            const bl_f2 A = eta - sp + a, B = kb - sp;
            const bl_f2 d = eta + a - kb; // = A - B
            const bl_f2 e_d = bl_exp2_2(__builtin_elementwise_abs(d) * bl2(-BL_LOG2E));
            const bl_f2 op_d = e_d + bl2(1.0f);
            lsite += bl_fma2(bl_log2_2(op_d), bl2(BL_LN2), __builtin_elementwise_max(A, B));
            const bl_f2 q = bl_sel_pos_one(d, e_d) * bl_rcp_2(op_d); // P(z=1 | y, theta)
            dsum += q - psi;
#pragma unroll
            for (int k = 0; k <= KO; k++) ga2[k] = bl_fma2(q, g[k], ga2[k]); // dummy site: g == 0
        };
        if constexpr (ONE1) one_period(0); // (one period: no loop)
        else for (int t = 0; t < T; t++) one_period(t);
        ll2 = bl_fma2(lsite, vmask, ll2);
        dsum *= vmask;
        gb2[0] += dsum;
#pragma unroll
        for (int k = 0; k < KS; k++) gb2[k + 1] = bl_fma2(dsum, x[k], gb2[k + 1]);
    };
    if constexpr (ONE1) { if (ct < npairs) one_pair(ct); } // (at most one pair per lane: no loop)
    else for (int m = ct; m < npairs; m += CT) one_pair(m);
    ll += ll2.x + ll2.y;
#pragma unroll
    for (int k = 0; k <= KS; k++) gb[k] += gb2[k].x + gb2[k].y;
#pragma unroll
    for (int k = 0; k <= KO; k++) ga[k] += ga2[k].x + ga2[k].y;
}

// Same arithmetic straight from the HBM rows (slice too large for LDS); grows is already offset to
// this workgroup's first site, ld = n_stride.
template <int KS, int KO, int CT>
__device__ __forceinline__ void bl_eval_sites_hbm(int ct, const float *__restrict__ grows, int ld, int cnt, int T, int J,
                                                  const float (&beta)[KS + 1], const float (&alpha)[KO + 1],
                                                  float &ll, float (&gb)[KS + 1], float (&ga)[KO + 1])
{
    const int V = T * J;
    const int row_wc = KS, row_ka = KS + V * (KO + 1), row_kb = row_ka + T;
    for (int i = ct; i < cnt; i += CT) {
        float x[KS > 0 ? KS : 1];
        float eta = beta[0];
#pragma unroll
        for (int k = 0; k < KS; k++) {
            x[k] = grows[(size_t)k * ld + i];
            eta = fmaf(x[k], beta[k + 1], eta);
        }
        float sp, psi;
        bl_site_head(eta, sp, psi);
        float dsum = 0.0f;
        for (int t = 0; t < T; t++) {
            float a = grows[(size_t)(row_ka + t) * ld + i];
            const float kb = grows[(size_t)(row_kb + t) * ld + i];
            float g[KO + 1];
#pragma unroll
            for (int k = 0; k <= KO; k++) g[k] = 0.0f;
#pragma unroll 4
            for (int j = 0; j < J; j++) {
                const size_t r0 = (size_t)(row_wc + (t * J + j) * (KO + 1));
                float w[KO + 1];
#pragma unroll
                for (int k = 0; k <= KO; k++) w[k] = grows[(r0 + k) * ld + i];
                bl_visit<KO>(w, alpha, a, g);
            }
            bl_period_tail<KO>(eta, sp, psi, a, kb, g, ll, dsum, ga);
        }
        gb[0] += dsum;
#pragma unroll
        for (int k = 0; k < KS; k++) gb[k + 1] = fmaf(dsum, x[k], gb[k + 1]);
    }
}


// ------------------------------------------------- occu with false positives (MODEL 2) ----
// biolith/models/occu.py:146-157,229-241: P(y=1 | z) = 1 - (1 - z p)(1 - f_c)(1 - (1 - z) f_u) with one of
// f_c ("constant") / f_u ("unoccupied") sampled, the other 0.  Both are the same likelihood with
//   z = 1:  P = p + f1 (1 - p)   (f1 = f for "constant", 0 for "unoccupied"),      z = 0:  P = f
// and one extra unconstrained coordinate phi = logit f.  On the sign-folded records (u = c nu):
//   detection      log P     = log sigma(u) + log(1 + f1 e^-u)
//   non-detection  log(1-P)  = log sigma(u) + log(1 - f1)
//   z = 0 branch   n_det log f + n_nondet log(1 - f)     (replaces n_det log(tiny))
// so the occu visit arithmetic stays and each visit adds one log and one rcp.  The counts come from
// the record's ka = n_masked ln2 and kb = n_det log(tiny).  Accumulates d/dphi (chain rule through
// f = sigmoid(phi) included) into gphi.  Pair records, runtime J.
struct BlFpScalars {
    float f, f1, lf, l1f, l1f1, ff1; // f, f1, log f, log(1-f), log(1-f1), f (1-f)
    float z1;                        // 1 if f also acts on occupied sites ("constant")
};
__device__ __forceinline__ BlFpScalars bl_fp_scalars(float phi, int fp_z1)
{
    const float e = __builtin_amdgcn_exp2f(-fabsf(phi) * BL_LOG2E), op = 1.0f + e;
    const float l = BL_LN2 * __builtin_amdgcn_logf(op), r = __builtin_amdgcn_rcpf(op);
    BlFpScalars s;
    s.f = (phi > 0.0f ? 1.0f : e) * r;
    s.lf = fminf(phi, 0.0f) - l;    // -softplus(-phi)
    s.l1f = -fmaxf(phi, 0.0f) - l;  // -softplus(phi)
    s.ff1 = e * r * r;              // f (1 - f)
    s.z1 = fp_z1 ? 1.0f : 0.0f;
    s.f1 = fp_z1 ? s.f : 0.0f;
    s.l1f1 = fp_z1 ? s.l1f : 0.0f;
    return s;
}

template <int KS, int KO, int CT>
__device__ __forceinline__ void bl_eval_sites_fp(int ct, int pstride, int cnt, int T, int J, const BlFpScalars fp,
                                                 const float (&beta)[KS + 1], const float (&alpha)[KO + 1],
                                                 float &ll, float (&gb)[KS + 1], float (&ga)[KO + 1], float &gphi, int data_off = 0)
{
    constexpr int XQ = (KS + 3) & ~3;
    const int pb = bl_period_block(J, KO);
    const float *data = bl_lds_f(BL_OFF_DATA) + data_off;
    const int npairs = (cnt + 1) >> 1;
    const float Jf = (float)J;
    bl_f2 ll2 = bl2(0.0f), gb2[KS + 1], ga2[KO + 1], gp2 = bl2(0.0f);
#pragma unroll
    for (int k = 0; k <= KS; k++) gb2[k] = bl2(0.0f);
#pragma unroll
    for (int k = 0; k <= KO; k++) ga2[k] = bl2(0.0f);
    for (int m = ct; m < npairs; m += CT) {
        const float4 *rec = reinterpret_cast<const float4 *>(data + (size_t)m * pstride);
        const bl_f2 vmask = bl_f2{1.0f, (2 * m + 1 < cnt) ? 1.0f : 0.0f};
        bl_f2 x[XQ];
#pragma unroll
        for (int q = 0; q < XQ / 2; q++) {
            const float4 v = rec[q];
            x[2 * q] = bl_f2{v.x, v.y};
            x[2 * q + 1] = bl_f2{v.z, v.w};
        }
        bl_f2 eta = bl2(beta[0]);
#pragma unroll
        for (int k = 0; k < KS; k++) eta = bl_fma2(x[k], bl2(beta[k + 1]), eta);
        const bl_f2 e_eta = bl_exp2_2(__builtin_elementwise_abs(eta) * bl2(-BL_LOG2E));
        const bl_f2 op_eta = e_eta + bl2(1.0f);
        const bl_f2 sp = bl_fma2(bl_log2_2(op_eta), bl2(BL_LN2), __builtin_elementwise_max(eta, bl2(0.0f)));
        const bl_f2 psi = bl_sel_pos_one(eta, e_eta) * bl_rcp_2(op_eta);
        bl_f2 dsum = bl2(0.0f), lsite = bl2(0.0f), gpsite = bl2(0.0f);
        const float2 *pp0 = reinterpret_cast<const float2 *>(data + (size_t)m * pstride) + XQ;
        for (int t = 0; t < T; t++) {
            const float2 *pp = pp0 + t * pb;
            bl_f2 g[KO + 1];
#pragma unroll
            for (int k = 0; k <= KO; k++) g[k] = bl2(0.0f);
            const float2 a_ = pp[J * (KO + 1)], kb_ = pp[J * (KO + 1) + 1];
            bl_f2 a = bl_f2{a_.x, a_.y};
            bl_f2 gf = bl2(0.0f); // d a / d f1
            // counts of this (site, period): detections, non-detections (the rest of J is masked)
            const bl_f2 ndet = bl_f2{__builtin_rintf(kb_.x * (1.0f / -87.33654475f)), __builtin_rintf(kb_.y * (1.0f / -87.33654475f))};
            const bl_f2 nmask = bl_f2{__builtin_rintf(a_.x * (1.0f / BL_LN2)), __builtin_rintf(a_.y * (1.0f / BL_LN2))};
            const bl_f2 nnon = bl2(Jf) - ndet - nmask;
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
                const bl_f2 op = tt + bl2(1.0f);
                a = bl_fma2(bl_log2_2(op), bl2(-BL_LN2), a);
                // c = +1 -> 1, else 0; and 0 throughout when f does not act on occupied sites
                const bl_f2 det = __builtin_elementwise_max(w[0], bl2(0.0f)) * bl2(fp.z1);
                const bl_f2 td = tt * det;
                const bl_f2 opf = bl_fma2(td, bl2(fp.f1), bl2(1.0f));         // 1 + f1 e^-u on detections, else 1
                a = bl_fma2(bl_log2_2(opf), bl2(BL_LN2), a);
                const bl_f2 tr = td * bl_rcp_2(opf);                          // e^-u / (1 + f1 e^-u)
                gf += tr;
                const bl_f2 s = bl_fma2(tr, bl2(-fp.f1), tt * bl_rcp_2(op));  // d/du
#pragma unroll
                for (int k = 0; k <= KO; k++) g[k] = bl_fma2(s, w[k], g[k]);
            }
            a = bl_fma2(nnon, bl2(fp.l1f1), a);
            const bl_f2 kb = bl_fma2(ndet, bl2(fp.lf), nnon * bl2(fp.l1f));
            const bl_f2 A = eta - sp + a, B = kb - sp;
            const bl_f2 d = eta + a - kb;
            const bl_f2 e_d = bl_exp2_2(__builtin_elementwise_abs(d) * bl2(-BL_LOG2E));
            const bl_f2 op_d = e_d + bl2(1.0f);
            lsite += bl_fma2(bl_log2_2(op_d), bl2(BL_LN2), __builtin_elementwise_max(A, B));
            const bl_f2 q = bl_sel_pos_one(d, e_d) * bl_rcp_2(op_d); // P(z=1 | y, theta)
            dsum += q - psi;
#pragma unroll
            for (int k = 0; k <= KO; k++) ga2[k] = bl_fma2(q, g[k], ga2[k]);
            // d/dphi: z=1 branch (only when f acts there) and z=0 branch, each times f(1-f)
            const bl_f2 d1 = bl_fma2(gf, bl2(fp.ff1), nnon * bl2(-fp.f * fp.z1));
            const bl_f2 d0 = bl_fma2(ndet, bl2(1.0f - fp.f), nnon * bl2(-fp.f));
            gpsite += bl_fma2(q, d1 - d0, d0);
        }
        ll2 = bl_fma2(lsite, vmask, ll2);
        gp2 = bl_fma2(gpsite, vmask, gp2);
        dsum *= vmask;
        gb2[0] += dsum;
#pragma unroll
        for (int k = 0; k < KS; k++) gb2[k + 1] = bl_fma2(dsum, x[k], gb2[k + 1]);
    }
    ll += ll2.x + ll2.y;
    gphi += gp2.x + gp2.y;
#pragma unroll
    for (int k = 0; k <= KS; k++) gb[k] += gb2[k].x + gb2[k].y;
#pragma unroll
    for (int k = 0; k <= KO; k++) ga[k] += ga2[k].x + ga2[k].y;
}


// ------------------------------------------- lane groups over the visits of a site pair (MODEL 0 and 2) ----
// One site pair per lane (above) is the latency optimum only while a (site, period) has few visits: the lane walks them one after the
// other, and at biolith's own defaults (simulate(): 100 sites x 52 visits, occu.py:251-252, 336), on its benchmark grid
// (benchmarks/occu_spoccupancy.py:16-70: up to 90 visits) and with stacked periods (T > 1, occu.py:198-210) most of a chain's lanes
// idle meanwhile.  Here a pair is worked on by a GROUP of G = Gt x Gj neighbouring lanes (host: occu_lane_group):
//   * Gt lanes split the PERIODS (lane's periods t = sub_t, sub_t + Gt, ...): nothing to exchange at all -- a period's share of
//     d/d eta (q - psi), of the log-likelihood and of d/d alpha are plain addends of sums that the wave reduction forms anyway;
//   * Gj lanes split the VISITS of a period (contiguous chunks): the one per-period quantity every lane needs whole is
//     a = sum_j log sigma(u_j), folded across the Gj lanes with log2(Gj) DPP adds per site of the pair; each lane then forms the
//     posterior occupancy weight q itself (one exp / log / rcp: less than one visit) and scales ITS visits' gradient partials by it.
//     Per-(site, period) addends (log-lik, q - psi) are counted by the chunk's first lane, per-site ones by the group's first lane.
// grp = log2(Gt) | log2(Gj) << 4 (wave-uniform; 0 selects the one-pair-per-lane forms above).  Runtime J; the records are the same.
template <int CTRL>
__device__ __forceinline__ bl_f2 bl_dpp2(bl_f2 v) { return bl_f2{bl_dpp<CTRL, 0xF>(v.x), bl_dpp<CTRL, 0xF>(v.y)}; }
// sum over 2^lg neighbouring lanes (aligned groups), result in every lane of the group
__device__ __forceinline__ bl_f2 bl_group_sum2(bl_f2 v, int lg)
{
    if (lg >= 1) v += bl_dpp2<0xB1>(v);  // quad_perm [1,0,3,2]
    if (lg >= 2) v += bl_dpp2<0x4E>(v);  // quad_perm [2,3,0,1]
    if (lg >= 3) v += bl_dpp2<0x141>(v); // row_half_mirror
    if (lg >= 4) v += bl_dpp2<0x140>(v); // row_mirror
    return v;
}

// T1: one period (and hence no period lanes) as a compile-time fact -- simulate()'s defaults and the whole benchmark grid
#ifndef BL_GRP_UNROLL
#define BL_GRP_UNROLL 2 // visits per iteration of a lane's run-time visit loop (A/B: 4)
#endif
// JC > 0 with OWNT: J == JC visits per period, no visit lanes, and ONE period per lane (period lanes == periods) as compile-time facts
// -- stacked periods at a few visits each (2 000 x 8 x 4): the lane's period is straight-line code, its visits unrolled
template <int KS, int KO, int CT, bool FP, bool T1 = false, int JC = 0, bool OWNT = false>
__device__ __forceinline__ void bl_eval_sites_grp(int ct, int pstride, int cnt, int T, int J_rt, int grp, const BlFpScalars fp,
                                                  const float (&beta)[KS + 1], const float (&alpha)[KO + 1],
                                                  float &ll, float (&gb)[KS + 1], float (&ga)[KO + 1], float &gphi, int data_off = 0)
{
    constexpr int XQ = (KS + 3) & ~3;
    constexpr int UNR = (JC > 0 && OWNT) ? JC : BL_GRP_UNROLL; // the visit loop's unrolling
    const int J = JC > 0 ? JC : J_rt;
    const int lgt = grp & 15, lgj = (JC > 0 && OWNT) ? 0 : grp >> 4, lg = lgt + lgj;
    const int sub = ct & ((1 << lg) - 1), slot = ct >> lg, nslots = CT >> lg;
    const int sub_j = sub & ((1 << lgj) - 1), sub_t = sub >> lgj, Gt = 1 << lgt;
    const int jc = (J + (1 << lgj) - 1) >> lgj;              // visits per chunk
    const int j0 = (JC > 0 && OWNT) ? 0 : min(sub_j * jc, J), j1 = (JC > 0 && OWNT) ? JC : min(j0 + jc, J); // this lane's chunk of every period
    const bl_f2 firstj = bl2(sub_j == 0 ? 1.0f : 0.0f);
    const int pb = bl_period_block(J, KO);
    const float *data = bl_lds_f(BL_OFF_DATA) + data_off;
    const int npairs = (cnt + 1) >> 1;
    const float Jf = (float)J;
    bl_f2 ll2 = bl2(0.0f), gb2[KS + 1], ga2[KO + 1], gp2 = bl2(0.0f);
#pragma unroll
    for (int k = 0; k <= KS; k++) gb2[k] = bl2(0.0f);
#pragma unroll
    for (int k = 0; k <= KO; k++) ga2[k] = bl2(0.0f);
    for (int m = slot; m < npairs; m += nslots) { // (all lanes of a group share m: they enter and leave the loop together)
        const float4 *rec = reinterpret_cast<const float4 *>(data + (size_t)m * pstride);
        const bl_f2 vmask = bl_f2{1.0f, (2 * m + 1 < cnt) ? 1.0f : 0.0f}; // odd slice: the last pair's second site is a dummy
        bl_f2 x[XQ];
#pragma unroll
        for (int q = 0; q < XQ / 2; q++) {
            const float4 v = rec[q];
            x[2 * q] = bl_f2{v.x, v.y};
            x[2 * q + 1] = bl_f2{v.z, v.w};
        }
        bl_f2 eta = bl2(beta[0]);
#pragma unroll
        for (int k = 0; k < KS; k++) eta = bl_fma2(x[k], bl2(beta[k + 1]), eta);
        const bl_f2 e_eta = bl_exp2_2(__builtin_elementwise_abs(eta) * bl2(-BL_LOG2E));
        const bl_f2 op_eta = e_eta + bl2(1.0f);
        const bl_f2 sp = bl_fma2(bl_log2_2(op_eta), bl2(BL_LN2), __builtin_elementwise_max(eta, bl2(0.0f)));
        const bl_f2 psi = bl_sel_pos_one(eta, e_eta) * bl_rcp_2(op_eta);
        bl_f2 dsum = bl2(0.0f), lsite = bl2(0.0f), gpsite = bl2(0.0f);
        const float2 *pp0 = reinterpret_cast<const float2 *>(data + (size_t)m * pstride) + XQ;
        auto one_period = [&](int t) {
            const float2 *pp = pp0 + t * pb;
            bl_f2 g[KO + 1];
#pragma unroll
            for (int k = 0; k <= KO; k++) g[k] = bl2(0.0f);
            const float2 a_ = pp[J * (KO + 1)], kb_ = pp[J * (KO + 1) + 1];
            bl_f2 a = bl_f2{a_.x, a_.y} * firstj; // ka once per (site, period)
            bl_f2 gf = bl2(0.0f);                 // FP: d a / d f1 (this lane's visits)
#pragma unroll UNR
            for (int j = j0; j < j1; j++) {
                bl_f2 w[KO + 1];
#pragma unroll
                for (int k = 0; k <= KO; k++) {
                    const float2 v = pp[j * (KO + 1) + k];
                    w[k] = bl_f2{v.x, v.y};
                }
                if constexpr (!FP) {
                    bl_visit2<KO>(w, alpha, a, g);
                } else { // (bl_eval_sites_fp's visit)
                    bl_f2 u = w[0] * bl2(alpha[0]);
#pragma unroll
                    for (int k = 1; k <= KO; k++) u = bl_fma2(w[k], bl2(alpha[k]), u);
                    const bl_f2 tt = bl_expneg_2(u);
                    const bl_f2 op = tt + bl2(1.0f);
                    a = bl_fma2(bl_log2_2(op), bl2(-BL_LN2), a);
                    const bl_f2 det = __builtin_elementwise_max(w[0], bl2(0.0f)) * bl2(fp.z1);
                    const bl_f2 td = tt * det;
                    const bl_f2 opf = bl_fma2(td, bl2(fp.f1), bl2(1.0f));
                    a = bl_fma2(bl_log2_2(opf), bl2(BL_LN2), a);
                    const bl_f2 tr = td * bl_rcp_2(opf);
                    gf += tr;
                    const bl_f2 s = bl_fma2(tr, bl2(-fp.f1), tt * bl_rcp_2(op));
#pragma unroll
                    for (int k = 0; k <= KO; k++) g[k] = bl_fma2(s, w[k], g[k]);
                }
            }
            a = bl_group_sum2(a, lgj);
            bl_f2 kb = bl_f2{kb_.x, kb_.y}, ndet = bl2(0.0f), nnon = bl2(0.0f);
            if constexpr (FP) { // counts of this (site, period) from the record's ka = n_masked ln2, kb = n_det log(tiny)
                ndet = bl_f2{__builtin_rintf(kb_.x * (1.0f / -87.33654475f)), __builtin_rintf(kb_.y * (1.0f / -87.33654475f))};
                const bl_f2 nmask = bl_f2{__builtin_rintf(a_.x * (1.0f / BL_LN2)), __builtin_rintf(a_.y * (1.0f / BL_LN2))};
                nnon = bl2(Jf) - ndet - nmask;
                a = bl_fma2(nnon, bl2(fp.l1f1), a);
                kb = bl_fma2(ndet, bl2(fp.lf), nnon * bl2(fp.l1f));
            }
            const bl_f2 A = eta - sp + a, B = kb - sp;
            const bl_f2 d = eta + a - kb; // = A - B
            const bl_f2 e_d = bl_exp2_2(__builtin_elementwise_abs(d) * bl2(-BL_LOG2E));
            const bl_f2 op_d = e_d + bl2(1.0f);
            lsite = bl_fma2(bl_fma2(bl_log2_2(op_d), bl2(BL_LN2), __builtin_elementwise_max(A, B)), firstj, lsite);
            const bl_f2 q = bl_sel_pos_one(d, e_d) * bl_rcp_2(op_d); // P(z=1 | y, theta)
            dsum = bl_fma2(q - psi, firstj, dsum);
#pragma unroll
            for (int k = 0; k <= KO; k++) ga2[k] = bl_fma2(q, g[k], ga2[k]); // this lane's visits (dummy site: g == 0)
            if constexpr (FP) {
                // d/dphi is linear in gf: q gf ff1 from every lane, the counts' part once per (site, period)
                const bl_f2 d1c = nnon * bl2(-fp.f * fp.z1);
                const bl_f2 d0 = bl_fma2(ndet, bl2(1.0f - fp.f), nnon * bl2(-fp.f));
                gpsite += q * gf * bl2(fp.ff1) + bl_fma2(q, d1c - d0, d0) * firstj;
            }
        };
        if constexpr (T1) one_period(0); // (one period, no period lanes: no loop)
        else if constexpr (OWNT) one_period(sub_t); // (period lanes == periods: this lane's one)
        else for (int t = sub_t; t < T; t += Gt) one_period(t); // (the lanes that fold a period's sums share sub_t, hence this loop's trip count)
        ll2 = bl_fma2(lsite, vmask, ll2);
        if constexpr (FP) gp2 = bl_fma2(gpsite, vmask, gp2);
        dsum *= vmask;
        gb2[0] += dsum;
#pragma unroll
        for (int k = 0; k < KS; k++) gb2[k + 1] = bl_fma2(dsum, x[k], gb2[k + 1]);
    }
    ll += ll2.x + ll2.y;
    if constexpr (FP) gphi += gp2.x + gp2.y;
#pragma unroll
    for (int k = 0; k <= KS; k++) gb[k] += gb2[k].x + gb2[k].y;
#pragma unroll
    for (int k = 0; k <= KO; k++) ga[k] += ga2[k].x + ga2[k].y;
}


// ------------------------------------------------------- count occupancy (occu_cop, MODEL 3) ----
// biolith/models/occu_cop.py:17-255: y_itj ~ Poisson(dur_itj * (z lambda_itj + (1 - z) f_u + f_c)),
// lambda = exp(alpha0 + w alpha), z ~ Bernoulli(psi) summed out; at most one of the false-positive rates
// f_c ("constant") / f_u ("unoccupied") is sampled (fp_mode 1 / 2, else 0).  With f the sampled rate:
//   z = 1:  rate dur (lambda + f1)   (f1 = f for "constant", else 0),      z = 0:  rate dur f0  (f0 = f, or 0)
// The parameter-free part sum m (y log dur - lgamma(y+1)) is common to both branches and is added by
// the host to the constant of the potential.  What remains per visit (m folded into y_m, d_m):
//   z = 1:  y_m log(lambda + f1) - d_m (lambda + f1)     [ = y_m nu - d_m lambda  when f1 = 0: no log ]
//   z = 0:  Ysum log f - Dsum f                          [ f0 = 0:  0 if Ysum = 0, else -inf: Poisson(0) ]
// Records: visit = (y_m, d_m, w_1..w_KO) -- one float wider than an occu visit, so every layout helper
// is called with KO + 1 -- and per period ka = Ysum, kb = Dsum.  The rate enters as phi = log f
// (NumPyro's unconstrained coordinate of a positive site); d/dphi = f d/df is accumulated in gphi.
struct BlCopScalars {
    float f, f1, phi;  // sampled rate, its share in the z = 1 branch, log f
    float has0;        // 1 if the z = 0 branch has a positive rate
    float z1;          // 1 if f acts on occupied sites
};
__device__ __forceinline__ BlCopScalars bl_cop_scalars(float phi, int fp_mode)
{
    BlCopScalars s;
    s.phi = phi;
    s.f = fp_mode ? __builtin_amdgcn_exp2f(fminf(phi, 80.0f) * BL_LOG2E) : 0.0f;
    s.z1 = fp_mode == 1 ? 1.0f : 0.0f;
    s.f1 = fp_mode == 1 ? s.f : 0.0f;
    s.has0 = fp_mode ? 1.0f : 0.0f;
    return s;
}

// grp (round 4): lanes that share a site pair, as for the plain model (bl_eval_sites_grp: log2(period lanes) | log2(visit lanes) << 4; 0 = one pair
// per lane) -- the reference's own sizes for this model are simulate_cop()'s defaults, 100 sites x 52 visits (occu_cop.py:258-396).  The one
// per-period value every lane needs whole is the z = 1 branch's sum a; the z = 0 branch is data (Ysum, Dsum) and the rate.
template <int KS, int KO, int CT>
__device__ __forceinline__ void bl_eval_sites_cop(int ct, int pstride, int cnt, int T, int J, const BlCopScalars fp,
                                                  const float (&beta)[KS + 1], const float (&alpha)[KO + 1],
                                                  float &ll, float (&gb)[KS + 1], float (&ga)[KO + 1], float &gphi, int grp = 0)
{
    constexpr int XQ = (KS + 3) & ~3;
    constexpr int VW = KO + 2; // floats per visit
    const int pb = bl_period_block(J, KO + 1);
    const float *data = bl_lds_f(BL_OFF_DATA);
    const int npairs = (cnt + 1) >> 1;
    const bool with_f1 = fp.z1 != 0.0f; // wave-uniform
    const int lgt = grp & 15, lgj = grp >> 4, lg = lgt + lgj;
    const int sub = ct & ((1 << lg) - 1), slot = ct >> lg, nslots = CT >> lg;
    const int sub_j = sub & ((1 << lgj) - 1), sub_t = sub >> lgj, Gt = 1 << lgt;
    const int jc = (J + (1 << lgj) - 1) >> lgj;
    const int j0 = min(sub_j * jc, J), j1 = min(j0 + jc, J); // this lane's chunk of every period's visits
    const bl_f2 firstj = bl2(sub_j == 0 ? 1.0f : 0.0f);      // per-(site, period) addends are counted by the chunk's first lane
    bl_f2 ll2 = bl2(0.0f), gb2[KS + 1], ga2[KO + 1], gp2 = bl2(0.0f);
#pragma unroll
    for (int k = 0; k <= KS; k++) gb2[k] = bl2(0.0f);
#pragma unroll
    for (int k = 0; k <= KO; k++) ga2[k] = bl2(0.0f);
    for (int m = slot; m < npairs; m += nslots) {
        const float4 *rec = reinterpret_cast<const float4 *>(data + (size_t)m * pstride);
        const bl_f2 vmask = bl_f2{1.0f, (2 * m + 1 < cnt) ? 1.0f : 0.0f};
        bl_f2 x[XQ];
#pragma unroll
        for (int q = 0; q < XQ / 2; q++) {
            const float4 v = rec[q];
            x[2 * q] = bl_f2{v.x, v.y};
            x[2 * q + 1] = bl_f2{v.z, v.w};
        }
        bl_f2 eta = bl2(beta[0]);
#pragma unroll
        for (int k = 0; k < KS; k++) eta = bl_fma2(x[k], bl2(beta[k + 1]), eta);
        const bl_f2 e_eta = bl_exp2_2(__builtin_elementwise_abs(eta) * bl2(-BL_LOG2E));
        const bl_f2 op_eta = e_eta + bl2(1.0f);
        const bl_f2 sp = bl_fma2(bl_log2_2(op_eta), bl2(BL_LN2), __builtin_elementwise_max(eta, bl2(0.0f)));
        const bl_f2 psi = bl_sel_pos_one(eta, e_eta) * bl_rcp_2(op_eta);
        bl_f2 dsum = bl2(0.0f), lsite = bl2(0.0f), gpsite = bl2(0.0f);
        const float2 *pp0 = reinterpret_cast<const float2 *>(data + (size_t)m * pstride) + XQ;
        for (int t = sub_t; t < T; t += Gt) {
            const float2 *pp = pp0 + t * pb;
            bl_f2 g[KO + 1];
#pragma unroll
            for (int k = 0; k <= KO; k++) g[k] = bl2(0.0f);
            const float2 ys_ = pp[J * VW], ds_ = pp[J * VW + 1];
            const bl_f2 ysum = bl_f2{ys_.x, ys_.y}, dsm = bl_f2{ds_.x, ds_.y};
            bl_f2 a = bl2(0.0f), gf = bl2(0.0f);
#pragma unroll 2
            for (int j = j0; j < j1; j++) {
                const float2 y_ = pp[j * VW], d_ = pp[j * VW + 1];
                const bl_f2 ym = bl_f2{y_.x, y_.y}, dm = bl_f2{d_.x, d_.y};
                bl_f2 w[KO > 0 ? KO : 1];
                bl_f2 nu = bl2(alpha[0]);
#pragma unroll
                for (int k = 0; k < KO; k++) {
                    const float2 v = pp[j * VW + 2 + k];
                    w[k] = bl_f2{v.x, v.y};
                    nu = bl_fma2(w[k], bl2(alpha[k + 1]), nu);
                }
                nu = __builtin_elementwise_min(nu, bl2(80.0f)); // keeps lambda finite: 0 * inf on masked visits
                const bl_f2 lam = bl_exp2_2(nu * bl2(BL_LOG2E));
                bl_f2 s; // d/dnu
                if (with_f1) {
                    const bl_f2 tt = lam + bl2(fp.f1);
                    a = bl_fma2(bl_log2_2(tt) * bl2(BL_LN2), ym, bl_fma2(dm, -tt, a));
                    const bl_f2 r = bl_fma2(ym, bl_rcp_2(tt), -dm); // y / (lambda + f) - d
                    gf += r;
                    s = r * lam;
                } else {
                    a = bl_fma2(ym, nu, bl_fma2(dm, -lam, a));
                    s = bl_fma2(dm, -lam, ym);
                }
                g[0] += s;
#pragma unroll
                for (int k = 0; k < KO; k++) g[k + 1] = bl_fma2(s, w[k], g[k + 1]);
            }
            a = bl_group_sum2(a, lgj);
            // z = 0 branch: Ysum log f - Dsum f, or Poisson(0): 0 / -inf
            const bl_f2 b_f = bl_fma2(ysum, bl2(fp.phi), dsm * bl2(-fp.f));
            const bl_f2 b_0 = bl_f2{ysum.x > 0.0f ? -INFINITY : 0.0f, ysum.y > 0.0f ? -INFINITY : 0.0f};
            const bl_f2 kb = fp.has0 != 0.0f ? b_f : b_0;
            const bl_f2 A = eta - sp + a, B = kb - sp;
            const bl_f2 d = eta + a - kb;
            const bl_f2 e_d = bl_exp2_2(__builtin_elementwise_abs(d) * bl2(-BL_LOG2E));
            const bl_f2 op_d = e_d + bl2(1.0f);
            lsite = bl_fma2(bl_fma2(bl_log2_2(op_d), bl2(BL_LN2), __builtin_elementwise_max(A, B)), firstj, lsite);
            const bl_f2 q = bl_sel_pos_one(d, e_d) * bl_rcp_2(op_d); // P(z=1 | y, theta)
            dsum = bl_fma2(q - psi, firstj, dsum);
#pragma unroll
            for (int k = 0; k <= KO; k++) ga2[k] = bl_fma2(q, g[k], ga2[k]); // (this lane's visits)
            // d/dphi = f d/df:  z=1: f sum_j (y/(lambda+f) - d)  ("constant" only; linear in this lane's share gf);  z=0: Ysum - Dsum f
            const bl_f2 d1 = gf * bl2(fp.f * fp.z1);
            const bl_f2 d0 = bl_fma2(dsm, bl2(-fp.f), ysum) * bl2(fp.has0);
            gpsite += bl_fma2(q, d1, (bl2(1.0f) - q) * d0 * firstj);
        }
        ll2 = bl_fma2(lsite, vmask, ll2);
        gp2 = bl_fma2(gpsite, vmask, gp2);
        dsum *= vmask;
        gb2[0] += dsum;
#pragma unroll
        for (int k = 0; k < KS; k++) gb2[k + 1] = bl_fma2(dsum, x[k], gb2[k + 1]);
    }
    ll += ll2.x + ll2.y;
    gphi += gp2.x + gp2.y;
#pragma unroll
    for (int k = 0; k <= KS; k++) gb[k] += gb2[k].x + gb2[k].y;
#pragma unroll
    for (int k = 0; k <= KO; k++) ga[k] += ga2[k].x + ga2[k].y;
}


// lgamma(n + 1) for n < 128 (occu_rn, nmixture)
#define BL_RN_NB 128 // entries of the table: max_abundance <= 127
__device__ constexpr float BL_LGAMMA1P[128] = {0.000000000e+00f, 0.000000000e+00f, 6.931471806e-01f, 1.791759469e+00f, 3.178053830e+00f, 4.787491743e+00f, 6.579251212e+00f, 8.525161361e+00f, 1.060460290e+01f, 1.280182748e+01f, 1.510441257e+01f, 1.750230785e+01f, 1.998721450e+01f, 2.255216385e+01f, 2.519122118e+01f, 2.789927138e+01f, 3.067186011e+01f, 3.350507345e+01f, 3.639544521e+01f, 3.933988419e+01f, 4.233561646e+01f, 4.538013890e+01f, 4.847118135e+01f, 5.160667557e+01f, 5.478472940e+01f, 5.800360522e+01f, 6.126170176e+01f, 6.455753863e+01f, 6.788974314e+01f, 7.125703897e+01f, 7.465823635e+01f, 7.809222355e+01f, 8.155795946e+01f, 8.505446702e+01f, 8.858082754e+01f, 9.213617560e+01f, 9.571969454e+01f, 9.933061245e+01f, 1.029681986e+02f, 1.066317603e+02f, 1.103206397e+02f, 1.140342118e+02f, 1.177718814e+02f, 1.215330815e+02f, 1.253172711e+02f, 1.291239336e+02f, 1.329525750e+02f, 1.368027226e+02f, 1.406739236e+02f, 1.445657439e+02f, 1.484777670e+02f, 1.524095926e+02f, 1.563608363e+02f, 1.603311282e+02f, 1.643201123e+02f, 1.683274454e+02f, 1.723527971e+02f, 1.763958484e+02f, 1.804562914e+02f, 1.845338289e+02f, 1.886281734e+02f, 1.927390473e+02f, 1.968661817e+02f, 2.010093164e+02f, 2.051681995e+02f, 2.093425868e+02f, 2.135322415e+02f, 2.177369341e+02f, 2.219564418e+02f, 2.261905483e+02f, 2.304390436e+02f, 2.347017234e+02f, 2.389783896e+02f, 2.432688490e+02f, 2.475729141e+02f, 2.518904022e+02f, 2.562211356e+02f, 2.605649410e+02f, 2.649216498e+02f, 2.692910977e+02f, 2.736731243e+02f, 2.780675734e+02f, 2.824742927e+02f, 2.868931333e+02f, 2.913239501e+02f, 2.957666014e+02f, 3.002209486e+02f, 3.046868568e+02f, 3.091641936e+02f, 3.136528299e+02f, 3.181526396e+02f, 3.226634991e+02f, 3.271852877e+02f, 3.317178872e+02f, 3.362611820e+02f, 3.408150589e+02f, 3.453794071e+02f, 3.499541180e+02f, 3.545390855e+02f, 3.591342054e+02f, 3.637393756e+02f, 3.683544961e+02f, 3.729794689e+02f, 3.776141979e+02f, 3.822585888e+02f, 3.869125491e+02f, 3.915759882e+02f, 3.962488171e+02f, 4.009309483e+02f, 4.056222962e+02f, 4.103227765e+02f, 4.150323067e+02f, 4.197508056e+02f, 4.244781934e+02f, 4.292143919e+02f, 4.339593240e+02f, 4.387129142e+02f, 4.434750881e+02f, 4.482457727e+02f, 4.530248962e+02f, 4.578123880e+02f, 4.626081785e+02f, 4.674121996e+02f, 4.722243839e+02f, 4.770446655e+02f, 4.818729792e+02f, 4.867092611e+02f, 4.915534482e+02f};

// --------------------------------------------------------------- N-mixture (nmixture, MODEL 4) ----
// biolith/models/nmixture.py:150-220: N_it enumerated over 0..K with raw Poisson(lambda) weights (the model's
// "N_i_trunc_norm" factor cancels the Categorical's normalisation), cut off below the largest count of the
// (site, period); y_itj ~ Binomial(N_it, p_itj).  With nu = logit p, m the mask and c = sum_j m log(1 - p_j):
//   l = sum_j m y_j nu_j  - lambda + logsumexp_n [ n (eta + c) - lgamma(n+1) + B_n ]
//   B_n = sum_j m log C(n, y_j)   (data only: tabulated by the host, -inf below the largest count)
//   d l / d eta = E[n] - lambda ,   d l / d nu_j = m (y_j - E[n] p_j) ,   E[n] = posterior mean of N
// Records as for occu_cop: visit = (m y, m, w_1..w_KO) (layout KO + 1); B lives in HBM/L2 as
// tab[t][n][site] (site fastest: lanes read consecutive floats).  Two passes over n (max, then sums); the second
// stops at the last term that can matter.
// grp (round 4): lanes that share a site pair as for the plain model (simulate_nmixture()'s defaults are 100 sites x 52 visits,
// nmixture.py:223-369): the visit lanes fold c = sum_j m log(1 - p_j) (the slope of the terms in n); the sums over n are then formed by
// every lane of the group alike (same table entries), y's and p's gradient sums stay per lane and enter linearly.
template <int KS, int KO, int CT>
__device__ __forceinline__ void bl_eval_sites_nmix(int ct, int pstride, int cnt, int T, int J, int K,
                                                   const float *__restrict__ tab, int tab_ld,
                                                   const float (&beta)[KS + 1], const float (&alpha)[KO + 1],
                                                   float &ll, float (&gb)[KS + 1], float (&ga)[KO + 1], int grp = 0)
{
    constexpr int XQ = (KS + 3) & ~3;
    constexpr int VW = KO + 2;
    const int pb = bl_period_block(J, KO + 1);
    const float *data = bl_lds_f(BL_OFF_DATA);
    const int npairs = (cnt + 1) >> 1;
    const int lgt = grp & 15, lgj = grp >> 4, lg = lgt + lgj;
    const int sub = ct & ((1 << lg) - 1), slot = ct >> lg, nslots = CT >> lg;
    const int sub_j = sub & ((1 << lgj) - 1), sub_t = sub >> lgj, Gt = 1 << lgt;
    const int jc = (J + (1 << lgj) - 1) >> lgj;
    const int j0 = min(sub_j * jc, J), j1 = min(j0 + jc, J);
    const bl_f2 firstj = bl2(sub_j == 0 ? 1.0f : 0.0f);
    bl_f2 ll2 = bl2(0.0f), gb2[KS + 1], ga2[KO + 1];
#pragma unroll
    for (int k = 0; k <= KS; k++) gb2[k] = bl2(0.0f);
#pragma unroll
    for (int k = 0; k <= KO; k++) ga2[k] = bl2(0.0f);
    for (int m = slot; m < npairs; m += nslots) {
        const float4 *rec = reinterpret_cast<const float4 *>(data + (size_t)m * pstride);
        const bool second = 2 * m + 1 < cnt;
        const bl_f2 vmask = bl_f2{1.0f, second ? 1.0f : 0.0f};
        const float *t0 = tab + 2 * m, *t1 = tab + (second ? 2 * m + 1 : 2 * m); // dummy second site: re-read the first
        bl_f2 x[XQ];
#pragma unroll
        for (int q = 0; q < XQ / 2; q++) {
            const float4 v = rec[q];
            x[2 * q] = bl_f2{v.x, v.y};
            x[2 * q + 1] = bl_f2{v.z, v.w};
        }
        bl_f2 eta = bl2(beta[0]);
#pragma unroll
        for (int k = 0; k < KS; k++) eta = bl_fma2(x[k], bl2(beta[k + 1]), eta);
        const bl_f2 lam = bl_exp2_2(__builtin_elementwise_min(eta, bl2(80.0f)) * bl2(BL_LOG2E));
        bl_f2 dsum = bl2(0.0f), lsite = bl2(0.0f);
        const float2 *pp0 = reinterpret_cast<const float2 *>(data + (size_t)m * pstride) + XQ;
        for (int t = sub_t; t < T; t += Gt) {
            const float2 *pp = pp0 + t * pb;
            bl_f2 gy[KO + 1], gp[KO + 1];
#pragma unroll
            for (int k = 0; k <= KO; k++) { gy[k] = bl2(0.0f); gp[k] = bl2(0.0f); }
            bl_f2 a = bl2(0.0f), c = bl2(0.0f);
#pragma unroll 2
            for (int j = j0; j < j1; j++) {
                const float2 y_ = pp[j * VW], m_ = pp[j * VW + 1];
                const bl_f2 ym = bl_f2{y_.x, y_.y}, mk = bl_f2{m_.x, m_.y};
                bl_f2 w[KO > 0 ? KO : 1];
                bl_f2 nu = bl2(alpha[0]);
#pragma unroll
                for (int k = 0; k < KO; k++) {
                    const float2 v = pp[j * VW + 2 + k];
                    w[k] = bl_f2{v.x, v.y};
                    nu = bl_fma2(w[k], bl2(alpha[k + 1]), nu);
                }
                const bl_f2 e = bl_exp2_2(__builtin_elementwise_abs(nu) * bl2(-BL_LOG2E));
                const bl_f2 op = e + bl2(1.0f);
                const bl_f2 sp = bl_fma2(bl_log2_2(op), bl2(BL_LN2), __builtin_elementwise_max(nu, bl2(0.0f))); // softplus(nu)
                const bl_f2 p = bl_sel_pos_one(nu, e) * bl_rcp_2(op) * mk;                                        // m sigmoid(nu)
                a = bl_fma2(ym, nu, a);
                c = bl_fma2(mk, -sp, c);
                gy[0] += ym; gp[0] += p;
#pragma unroll
                for (int k = 0; k < KO; k++) { gy[k + 1] = bl_fma2(ym, w[k], gy[k + 1]); gp[k + 1] = bl_fma2(p, w[k], gp[k + 1]); }
            }
            // logsumexp over n of  n (eta + c) - lgamma(n+1) + B_n
            c = bl_group_sum2(c, lgj);
            const bl_f2 slope = eta + c;
            const float *b0 = t0 + (size_t)t * (K + 1) * tab_ld, *b1 = t1 + (size_t)t * (K + 1) * tab_ld;
            // pass 1: the maximum, and the last n whose term is within 25 nats of the running maximum -- a superset of
            // the terms within 25 nats of the final one, so pass 2 may stop there (what it skips is < e^-25 of the sum).
            // The terms are concave in n above the largest count (n slope, -lgamma(n+1) and every log C(n, y) are), so
            // once one falls 25 nats below the running maximum all later ones do: pass 1 stops there too.
            bl_f2 mx = bl2(-INFINITY);
            int hi0 = 0, hi1 = 0;
            for (int nb = 0; nb <= K; nb += 4) { // blocks of 4: their table loads are in flight together
                bl_f2 B[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int n = min(nb + q, K);
                    B[q] = bl_f2{b0[(size_t)n * tab_ld], b1[(size_t)n * tab_ld]};
                }
                bool in0 = false, in1 = false;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int n = min(nb + q, K); // the clamped tail repeats term K: harmless for max / hi
                    const bl_f2 tn = bl_fma2(bl2((float)n), slope, B[q] - bl2(BL_LGAMMA1P[n]));
                    mx = __builtin_elementwise_max(mx, tn);
                    in0 = tn.x >= mx.x - 25.0f; in1 = tn.y >= mx.y - 25.0f;
                    hi0 = in0 ? n : hi0;
                    hi1 = in1 ? n : hi1;
                }
                if (!in0 && !in1 && mx.x > -INFINITY && mx.y > -INFINITY) break; // the block's last term is out: so is the rest
            }
            const int n_hi = max(hi0, hi1);
            bl_f2 S = bl2(0.0f), S1 = bl2(0.0f);
#pragma unroll 4
            for (int n = 0; n <= n_hi; n++) {
                const bl_f2 B = bl_f2{b0[(size_t)n * tab_ld], b1[(size_t)n * tab_ld]};
                const bl_f2 tn = bl_fma2(bl2((float)n), slope, B - bl2(BL_LGAMMA1P[n])) - mx;
                const bl_f2 en = bl_exp2_2(tn * bl2(BL_LOG2E));
                S += en;
                S1 = bl_fma2(bl2((float)n), en, S1);
            }
            const bl_f2 En = S1 * bl_rcp_2(S);
            lsite += a + (mx - lam + bl_log2_2(S) * bl2(BL_LN2)) * firstj; // (a: this lane's visits)
            dsum = bl_fma2(En - lam, firstj, dsum);
#pragma unroll
            for (int k = 0; k <= KO; k++) ga2[k] += (gy[k] - En * gp[k]) * vmask;
        }
        ll2 = bl_fma2(lsite, vmask, ll2);
        dsum *= vmask;
        gb2[0] += dsum;
#pragma unroll
        for (int k = 0; k < KS; k++) gb2[k + 1] = bl_fma2(dsum, x[k], gb2[k + 1]);
    }
    ll += ll2.x + ll2.y;
#pragma unroll
    for (int k = 0; k <= KS; k++) gb[k] += gb2[k].x + gb2[k].y;
#pragma unroll
    for (int k = 0; k <= KO; k++) ga[k] += ga2[k].x + ga2[k].y;
}


#include "rn_device.hpp" // occu_rn: work-proportional site evaluation (items of 8 terms of the sum over N)
#include "dyn_device.hpp" // dynamic occupancy (builder-defined, BASELINE.json configs[4]): forward / backward recursions over the seasons
#define BL_DYN_MAX_KS 8   // MODEL 8 is instantiated for site-covariate capacities up to 8 (three coefficient blocks share the 64-float coefficient block)

// MODEL 0 = occu (occu.py); MODEL 1 = occu_rn (occu_rn.py) has its own entry, bl_eval_sites_rn (rn_device.hpp), called by bl_phase_a;
// MODEL 2 (occu with false positives), 3 (occu_cop), 4 (nmixture) are dispatched by bl_phase_a below
// JSEL: -1 = every form in one kernel, chosen by a switch on J (the parity hook); >= 0 = THE form of this kernel -- JSEL visits per period
// unrolled (1 .. 6, 8), or 0: any J at run time.  The sampler is instantiated per form (round 4): a kernel that carries all eight runs the
// one it needs 3 % slower (headline: 2.353 -> 2.278 us per leapfrog with J = 5 alone; profiles/r04/e_ab_grp_instantiation.txt).
template <int KS, int KO, bool LDS, int MODEL, int CT, int JSEL = -1, bool ONE1 = false>
__device__ __forceinline__ void bl_eval_sites(int ct, const float *__restrict__ grows, int ld_or_stride, int cnt, int T, int J, int K,
                                              const float (&beta)[KS + 1], const float (&alpha)[KO + 1],
                                              float &ll, float (&gb)[KS + 1], float (&ga)[KO + 1], int data_off = 0)
{
    static_assert(MODEL == 0, "bl_eval_sites: the plain occupancy model only");
    if constexpr (LDS && JSEL >= 0) {
        bl_eval_sites_lds<KS, KO, JSEL, CT, ONE1>(ct, ld_or_stride, cnt, T, J, beta, alpha, ll, gb, ga, data_off);
    } else if constexpr (LDS) {
        switch (J) { // wave-uniform
        case 1: bl_eval_sites_lds<KS, KO, 1, CT>(ct, ld_or_stride, cnt, T, J, beta, alpha, ll, gb, ga, data_off); break;
        case 2: bl_eval_sites_lds<KS, KO, 2, CT>(ct, ld_or_stride, cnt, T, J, beta, alpha, ll, gb, ga, data_off); break;
        case 3: bl_eval_sites_lds<KS, KO, 3, CT>(ct, ld_or_stride, cnt, T, J, beta, alpha, ll, gb, ga, data_off); break;
        case 4: bl_eval_sites_lds<KS, KO, 4, CT>(ct, ld_or_stride, cnt, T, J, beta, alpha, ll, gb, ga, data_off); break;
        case 5: bl_eval_sites_lds<KS, KO, 5, CT>(ct, ld_or_stride, cnt, T, J, beta, alpha, ll, gb, ga, data_off); break;
        case 6: bl_eval_sites_lds<KS, KO, 6, CT>(ct, ld_or_stride, cnt, T, J, beta, alpha, ll, gb, ga, data_off); break;
        case 8: bl_eval_sites_lds<KS, KO, 8, CT>(ct, ld_or_stride, cnt, T, J, beta, alpha, ll, gb, ga, data_off); break;
        default: bl_eval_sites_lds<KS, KO, 0, CT>(ct, ld_or_stride, cnt, T, J, beta, alpha, ll, gb, ga, data_off); break;
        }
    } else {
        bl_eval_sites_hbm<KS, KO, CT>(ct, grows, ld_or_stride, cnt, T, J, beta, alpha, ll, gb, ga);
    }
}

// Transpose this workgroup's site slice [s0, s0+cnt) of the HBM rows (coalesced reads along the
// site axis) into LDS pair records: element `pos` of site i lands at pair (i/2), float 2*pos + (i&1).
// Joint-species datasets (species > 0): the HBM rows hold the site covariates once and then one block of visit / ka / kb rows
// per species; every species gets a full record region of its own in LDS (lds_off floats from the first).
// ord (occu_rn, rn_device.hpp): position i takes site ord[i] of the slice instead of site i.
__device__ __forceinline__ void bl_stage_records(const float *__restrict__ rows, int n_stride, int s0, int cnt,
                                                 int T, int J, int KS, int KO, int pstride, int nthreads, int species = 0, int lds_off = 0,
                                                 const int *ord = nullptr)
{
    float *dst = bl_lds_f(BL_OFF_DATA) + lds_off;
    const int xq = bl_round4(KS), pb = bl_period_block(J, KO), V = T * J, vw = KO + 1;
    const int n_rows = KS + V * vw + 2 * T;
    const int species_rows = species * (V * vw + 2 * T);
    if (cnt & 1) { // dummy second site of the last pair: all zeros (its contributions are masked)
        float *last = dst + (size_t)(cnt >> 1) * pstride;
        for (int e = threadIdx.x; e < xq + T * pb; e += nthreads) last[2 * e + 1] = 0.0f;
    }
    for (int r = 0; r < n_rows; r++) {
        int pos;
        if (r < KS) pos = r;
        else if (r < KS + V * vw) {
            const int v = (r - KS) / vw, k = (r - KS) - v * vw, t = v / J, j = v - t * J;
            pos = xq + t * pb + j * vw + k;
        } else if (r < KS + V * vw + T) pos = xq + (r - KS - V * vw) * pb + J * vw;
        else pos = xq + (r - KS - V * vw - T) * pb + J * vw + 1;
        const float *src = rows + (size_t)(r < KS ? r : r + species_rows) * n_stride + s0;
        for (int i = threadIdx.x; i < cnt; i += nthreads) dst[(size_t)(i >> 1) * pstride + 2 * pos + (i & 1)] = src[ord ? ord[i] : i];
    }
}

// nmixture: this workgroup's columns of the table B[rows = T (K + 1)][site] (HBM / L2, row length n_stride; `tab` already points at
// the slice's first site) copied to LDS at byte offset `off`, row length `ld` floats (an even number >= cnt)
__device__ __forceinline__ void bl_stage_nmix_tab(const float *__restrict__ tab, int n_stride, int cnt, int ld, int rows, int off, int nthreads)
{
    float *dst = reinterpret_cast<float *>(bl_smem_raw + off);
    for (int r = 0; r < rows; r++)
        for (int i = threadIdx.x; i < cnt; i += nthreads) dst[(size_t)r * ld + i] = tab[(size_t)r * n_stride + i];
}

// Padded coefficient layout in LDS: beta_k at k (k <= KS), alpha_k at KS+1+k.  dim d of theta
// (d <= Ks: beta_d, else alpha_{d-Ks-1}) lives at:
// MODEL 2's extra coordinate phi (d = Ks+Ko+2) lives at KS+KO+3, after the log-lik slot of the partial rows.
__device__ __forceinline__ int bl_coef_pos(int d, int Ks, int Ko, int KS, int KO)
{
    return d <= Ks ? d : (d <= Ks + Ko + 1 ? KS + 1 + (d - Ks - 1) : KS + KO + 3);
}
// theta dimension of a model
// theta dimension of a model: MODEL 2 always carries the false-positive coordinate, MODEL 3 when fp_mode != 0
template <int MODEL> __device__ __forceinline__ int bl_model_dim(int Ks, int Ko, int fp_mode)
{
    if constexpr (MODEL == 8) return 3 * (Ks + 1) + Ko + 1; // [b_psi | b_gamma | b_eps | alpha]
    return Ks + Ko + 2 + ((MODEL == 2 || (MODEL == 3 && fp_mode != 0)) ? 1 : 0);
}
// MODEL 8: where coordinate d of theta lives in the coefficient block / a partial row (padded capacities KS, KO), and
// whether it takes beta's prior; d == D names the log-likelihood's slot
__device__ __forceinline__ int bl_dyn_pos(int d, int Ks, int Ko, int KS, int KO)
{
    const int B = Ks + 1;
    if (d < 3 * B) { const int blk = d / B; return blk * (KS + 1) + (d - blk * B); }
    if (d < 3 * B + Ko + 1) return BL_DYN_OA(KS) + (d - 3 * B);
    return BL_DYN_LL(KS, KO);
}
// visit width of the records in floats minus one: the KO every layout helper is called with
template <int MODEL> __device__ __forceinline__ constexpr int bl_layout_ko(int KO) { return KO + ((MODEL == 1 || MODEL == 3 || MODEL == 4) ? 1 : 0); }

template <int KS, int KO>
__device__ __forceinline__ void bl_load_coefs(float (&beta)[KS + 1], float (&alpha)[KO + 1], int coef_off = 0)
{
    const float *c = bl_lds_f(BL_OFF_COEF) + coef_off; // unused (padded) slots were zeroed once at kernel start
#pragma unroll
    for (int k = 0; k <= KS; k++) beta[k] = c[k];
#pragma unroll
    for (int k = 0; k <= KO; k++) alpha[k] = c[KS + 1 + k];
}

// Wave reduction of (gb, ga, ll) -> this wave's row of the LDS partial table, padded layout:
// [0..KS] d/dbeta, [KS+1..KS+KO+1] d/dalpha, [KS+KO+2] log-lik, [KS+KO+3] d/dphi (MODEL 2).  One interleaved DPP butterfly for
// all values (f32: the per-term rounding of the f32 site math dominates the error budget anyway);
// lane 63 holds the totals and stores them.  Cross-wave / cross-workgroup sums are done in f64.
template <int KS, int KO, bool EXTRA = false>
__device__ __forceinline__ void bl_wave_partials_to_lds(int cwave, float ll, const float (&gb)[KS + 1], const float (&ga)[KO + 1],
                                                        float gextra = 0.0f, int row_stride = BL_PART_STRIDE, int row_off = 0)
{
    constexpr int NV = KS + KO + 3 + (EXTRA ? 1 : 0);
    const int wave = cwave, lane = threadIdx.x & 63;
    float v[NV];
#pragma unroll
    for (int k = 0; k <= KS; k++) v[k] = gb[k];
#pragma unroll
    for (int k = 0; k <= KO; k++) v[KS + 1 + k] = ga[k];
    v[KS + KO + 2] = ll;
    if constexpr (EXTRA) v[KS + KO + 3] = gextra;
    if constexpr (NV <= 16) {
        // reduce-scatter: one lane of every (bank, row) group that ends up with a total stores it
        const float y = bl_wave_reduce_scatter<NV>(v);
        const int g = lane >> 2;
        const int k = (int)(BlScatter<NV>::slots() >> (4 * g)) & 15;
        if ((lane & 3) == 0 && ((BlScatter<NV>::stores() >> g) & 1u)) bl_lds_f(BL_OFF_PART)[wave * row_stride + row_off + k] = y;
    } else {
        bl_wave_sum_vec_l63<NV>(v);
        if (lane == 63) {
            float *part = bl_lds_f(BL_OFF_PART) + wave * row_stride + row_off;
#pragma unroll
            for (int k = 0; k < NV; k++) part[k] = v[k];
        }
    }
}

// Phase A of one evaluation for a compute thread `ct` of the workgroup: coefficients from LDS, the
// workgroup's site slice, wave partials into the LDS table.  Shared by the NUTS and logp kernels.
// Joint-species layouts (n_species > 1; LDS-staged occu and false-positive forms): species s keeps its coefficients at
// s * BL_SP_COEF(KS, KO) of the coefficient block (the shared false-positive coordinate after the last species), its records at
// s * sp_lds floats of the data region, and its partial sums at s * BL_SP_PART(KS, KO) of a wave's row (row stride
// n_species * BL_SP_PART).  One species: the round-1 layout (row stride BL_PART_STRIDE), unchanged.
#define BL_SP_COEF(KS, KO) ((KS) + (KO) + 2)
#define BL_SP_PART(KS, KO) ((KS) + (KO) + 4)
#define BL_PART_FLOATS 768 // floats between BL_OFF_PART and BL_OFF_CKR
// GRP (round 4): the kernel instantiation carries the lane-group evaluators of the plain / false-positive models (and, in the NUTS kernel,
// the one-workgroup path without an exchange).  The headline's instantiation does NOT: with both forms in one kernel the one-pair-per-lane
// evaluation ran 4.4 % slower and the k == 1 branches cost another 2.5 % (same-box A/B of variant libraries, profiles/NOTES.md) -- the
// host launches the GRP form only when it has chosen lane groups or a single workgroup.
// GRP = 0: the one-pair-per-lane evaluators only; 1: both, chosen at run time by lane_grp (the parity hook: one launch serves either);
// 2: the lane-group evaluator only (lane_grp = 0 is its one-lane group) -- the sampler's GRP instantiation, which then does not carry the
// eight unrolled one-pair forms either.
template <int KS, int KO, bool LDS, int MODEL, int CW, int GRP = 1, int JSEL = -1, bool ONE1 = false>
__device__ __forceinline__ void bl_phase_a(int ct, int cwave, const float *__restrict__ grows, int ld_or_stride, int cnt,
                                           int T, int J, int max_abundance, int fp_mode, const float *__restrict__ tab = nullptr,
                                           int tab_ld = 0, int n_species = 1, int sp_lds = 0, int rn_off = 0, int lane_grp = 0, int nmix_lds = 0)
{
    const int row_stride = n_species > 1 ? n_species * BL_SP_PART(KS, KO) : BL_PART_STRIDE;
    for (int sp = 0; sp < n_species; sp++) {
    float beta[KS + 1], alpha[KO + 1];
    bl_load_coefs<KS, KO>(beta, alpha, sp * BL_SP_COEF(KS, KO));
    // (-0.0f: the identity of IEEE addition, so the evaluators' closing `acc += a.x + a.y` folds to a plain add; with +0.0f the
    // compiler has to keep a `0 + x` per value -- 9 instructions per evaluation at (3, 3))
    float ll = -0.0f, gb[KS + 1], ga[KO + 1];
#pragma unroll
    for (int k = 0; k <= KS; k++) gb[k] = -0.0f;
#pragma unroll
    for (int k = 0; k <= KO; k++) ga[k] = -0.0f;
    if constexpr (MODEL == 2) {
        static_assert(LDS, "false-positive model: LDS records only");
        float gphi = -0.0f;
        const BlFpScalars fp = bl_fp_scalars(bl_lds_f(BL_OFF_COEF)[n_species * BL_SP_COEF(KS, KO) + 1], fp_mode == 1);
        if constexpr (GRP == 2) {
            bl_eval_sites_grp<KS, KO, CW * 64, true>(ct, ld_or_stride, cnt, T, J, lane_grp, fp, beta, alpha, ll, gb, ga, gphi, sp * sp_lds);
        } else {
            bool grouped = false;
            if constexpr (GRP == 1) {
                if (lane_grp > 0) {
                    bl_eval_sites_grp<KS, KO, CW * 64, true>(ct, ld_or_stride, cnt, T, J, lane_grp, fp, beta, alpha, ll, gb, ga, gphi, sp * sp_lds);
                    grouped = true;
                }
            }
            if (!grouped) bl_eval_sites_fp<KS, KO, CW * 64>(ct, ld_or_stride, cnt, T, J, fp, beta, alpha, ll, gb, ga, gphi, sp * sp_lds);
        }
        bl_wave_partials_to_lds<KS, KO, true>(cwave, ll, gb, ga, gphi, row_stride, sp * BL_SP_PART(KS, KO));
    } else if constexpr (MODEL == 4) {
        static_assert(LDS, "N-mixture model: LDS records only");
        // (round 4) the data-only table B[t][n][site] of this workgroup's sites staged in LDS behind the records when it fits (host:
        // nmix_lds; bl_stage_nmix_tab): the two passes over n then wait for LDS instead of L2 -- two instances, so that the staged one's
        // loads are DS instructions (one generic pointer for both would make them FLAT)
        if (nmix_lds) bl_eval_sites_nmix<KS, KO, CW * 64>(ct, ld_or_stride, cnt, T, J, max_abundance, reinterpret_cast<const float *>(bl_smem_raw + rn_off),
                                                           tab_ld, beta, alpha, ll, gb, ga, lane_grp);
        else bl_eval_sites_nmix<KS, KO, CW * 64>(ct, ld_or_stride, cnt, T, J, max_abundance, tab, tab_ld, beta, alpha, ll, gb, ga, lane_grp);
        bl_wave_partials_to_lds<KS, KO>(cwave, ll, gb, ga);
    } else if constexpr (MODEL == 8) {
        static_assert(LDS, "dynamic occupancy model: LDS records only");
        if constexpr (KS <= BL_DYN_MAX_KS) {
            const float *c = bl_lds_f(BL_OFF_COEF);
            float bq[3][KS + 1], gq[3][KS + 1];
#pragma unroll
            for (int b = 0; b < 3; b++)
#pragma unroll
                for (int k = 0; k <= KS; k++) { bq[b][k] = c[b * (KS + 1) + k]; gq[b][k] = 0.0f; }
#pragma unroll
            for (int k = 0; k <= KO; k++) alpha[k] = c[BL_DYN_OA(KS) + k];
            // (at most two periods per lane of a group: the one-visit-pass form on scaled likelihoods; else the first form)
            // GRP = 2 (the sampler's instantiation for T <= 2 G): the scaled form alone; 0: the first form alone (any T); 1: both (parity hook)
            // JSEL == 1 (the sampler's instantiation for T == G: one period per lane): the two-scans form alone
            // JSEL == 2: eight periods on eight lanes at four visits each as compile-time facts (BASELINE.json configs[4]: 3.86 -> 3.67 us)
            if constexpr (GRP == 2 && JSEL == 2) bl_eval_sites_dyn_scan<KS, KO, CW * 64, 8, 4>(ct, ld_or_stride, cnt, T, J, lane_grp, bq[0], bq[1], bq[2], alpha, ll, gq, ga);
            else if constexpr (GRP == 2 && JSEL == 1) bl_eval_sites_dyn_scan<KS, KO, CW * 64>(ct, ld_or_stride, cnt, T, J, lane_grp, bq[0], bq[1], bq[2], alpha, ll, gq, ga);
            else if constexpr (GRP == 2) bl_eval_sites_dyn_scaled<KS, KO, CW * 64>(ct, ld_or_stride, cnt, T, J, lane_grp, rn_off, bq[0], bq[1], bq[2], alpha, ll, gq, ga);
            else if constexpr (GRP == 0) bl_eval_sites_dyn<KS, KO, CW * 64>(ct, ld_or_stride, cnt, T, J, lane_grp, rn_off, bq[0], bq[1], bq[2], alpha, ll, gq, ga);
            else if (T == lane_grp && T > 1 && BL_DYN_SCAN) bl_eval_sites_dyn_scan<KS, KO, CW * 64>(ct, ld_or_stride, cnt, T, J, lane_grp, bq[0], bq[1], bq[2], alpha, ll, gq, ga);
            else if (T <= 2 * lane_grp) bl_eval_sites_dyn_scaled<KS, KO, CW * 64>(ct, ld_or_stride, cnt, T, J, lane_grp, rn_off, bq[0], bq[1], bq[2], alpha, ll, gq, ga);
            else bl_eval_sites_dyn<KS, KO, CW * 64>(ct, ld_or_stride, cnt, T, J, lane_grp, rn_off, bq[0], bq[1], bq[2], alpha, ll, gq, ga);
            bl_wave_partials_dyn<KS, KO>(cwave, ll, gq, ga);
        }
    } else if constexpr (MODEL == 1) {
        static_assert(LDS, "Royle-Nichols model: LDS records only");
        // (lane_grp carries bl_rn_npos for this model -- it has no lane groups: the caller reads it once per launch)
        bl_eval_sites_rn<KS, KO, CW, JSEL == 10>(cwave, ld_or_stride, cnt, T, J, max_abundance, rn_off, lane_grp, beta, alpha, ll, gb, ga); // (JSEL == 10: J <= 10)
        bl_wave_partials_to_lds<KS, KO>(cwave, ll, gb, ga);
    } else if constexpr (MODEL == 3) {
        static_assert(LDS, "count occupancy model: LDS records only");
        float gphi = -0.0f;
        const BlCopScalars fp = bl_cop_scalars(bl_lds_f(BL_OFF_COEF)[KS + KO + 3], fp_mode);
        bl_eval_sites_cop<KS, KO, CW * 64>(ct, ld_or_stride, cnt, T, J, fp, beta, alpha, ll, gb, ga, gphi, lane_grp);
        bl_wave_partials_to_lds<KS, KO, true>(cwave, ll, gb, ga, gphi);
    } else {
        if constexpr (LDS && GRP == 2) {
            float gphi = 0.0f; // (JSEL == 1 in a lane-group kernel: the one-period form)
            // (JSEL > 100 in a lane-group kernel: one period per lane and JSEL - 100 visits per period as compile-time facts)
            if constexpr (JSEL > 100) bl_eval_sites_grp<KS, KO, CW * 64, false, false, JSEL - 100, true>(ct, ld_or_stride, cnt, T, J, lane_grp, BlFpScalars{}, beta, alpha, ll, gb, ga, gphi, sp * sp_lds);
            else bl_eval_sites_grp<KS, KO, CW * 64, false, JSEL == 1>(ct, ld_or_stride, cnt, T, J, lane_grp, BlFpScalars{}, beta, alpha, ll, gb, ga, gphi, sp * sp_lds);
        } else {
            bool grouped = false;
            if constexpr (LDS && GRP == 1) {
                if (lane_grp > 0) { // lane groups over the visits (wave-uniform)
                    float gphi = 0.0f;
                    bl_eval_sites_grp<KS, KO, CW * 64, false>(ct, ld_or_stride, cnt, T, J, lane_grp, BlFpScalars{}, beta, alpha, ll, gb, ga, gphi, sp * sp_lds);
                    grouped = true;
                }
            }
            if (!grouped) bl_eval_sites<KS, KO, LDS, MODEL, CW * 64, JSEL, ONE1>(ct, grows, ld_or_stride, cnt, T, J, max_abundance, beta, alpha, ll, gb, ga, sp * sp_lds);
        }
        bl_wave_partials_to_lds<KS, KO>(cwave, ll, gb, ga, 0.0f, row_stride, sp * BL_SP_PART(KS, KO));
    }
    }
}
