// The scan kernels and their launch templates, per objective kind (vs_device_scan.h holds the device code).  Included
// by vs_scan_g.hip, vs_scan_poly.hip and vs_scan_lin.hip -- one translation unit per built-in separable objective, so
// that the ~100 kernel instances compile side by side -- and by vs_scan.hip for the shape type.
#pragma once
#include <cmath>

#include "vs_common.h"

#include "vs_device_scan.h"

struct scan_shape {
    int L, W;
};

unsigned vs_persistent_workgroups(vs_context* ctx, const void* kernel, int threads, size_t lds, uint64_t blocks, int l);

// one entry per objective kind: launch (or plan) the evaluation (fused = false) or the fused evaluate+reduce kernel
int vs_scan_launch_g(vs_context* ctx, const vs_eval_args& A, scan_shape s, unsigned* P_io, const double* shift_dev,
                     bool plan_only, bool fused);
int vs_scan_launch_poly(vs_context* ctx, const vs_eval_args& A, scan_shape s, unsigned* P_io, const double* shift_dev,
                        bool plan_only, bool fused);
int vs_scan_launch_lin(vs_context* ctx, const vs_eval_args& A, scan_shape s, unsigned* P_io, const double* shift_dev,
                       bool plan_only, bool fused);

// fused evaluate + both orders (k <= 32, one output; vs_scan_gram.hip): plan or launch; nbh = 16-row blocks per half
// (1: k <= 16, 2: k <= 32)
int vs_scan_gram_launch(vs_context* ctx, const vs_eval_args& A, int nbh, unsigned* P_io, const double* shift_dev,
                        double* first_partials, double* gram_partials, bool plan_only);

// the same beyond 32 columns: nb = 8 (65 .. 128 columns: eight waves, one workgroup per CU; vs_scan_gram8*.hip, one
// translation unit per objective kind) or nb = 4 (33 .. 64 columns: four waves, two workgroups per CU; vs_scan_gram4.hip)
int vs_scan_gram8_launch(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                         double* first_partials, double* gram_partials, bool plan_only);
int vs_scan_gram4_launch(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                         double* first_partials, double* gram_partials, bool plan_only);

// several outputs (2 .. lmax, k l <= 128) in the fused both-orders kernels: nb = 1, 2 (four waves, 16 / 32 index rows per
// half), 4 (four waves, 64) in vs_scan_gramm.hip, 8 (eight waves, 128) in vs_scan_gramm8_*.hip; vs_scan_gram_multi_lmax:
// the most outputs the instance of `nb` takes (0: none)
int vs_scan_gram_multi_lmax(int nb);
int vs_scan_gram_multi_launch(vs_context* ctx, const vs_eval_args& A, int nb, unsigned* P_io, const double* shift_dev,
                              double* first_partials, double* gram_partials, bool plan_only);
#define VS_GRAM_LMAX_SMALL 8   // nb = 1, 2
#define VS_GRAM_LMAX_4 4       // nb = 4: only with 16 or more columns (the host's choice, vs_generate.hip), i.e. up to 4 outputs
#define VS_GRAM_LMAX_8 4       // nb = 8: the 160 KiB of LDS hold the tile, two totals slabs and four (a, b) row pairs

#ifdef VS_SCAN_KIND
// DIV: the column code may meet a reciprocal of 0 (a parameter whose 1 + a Markstein's theorem does not cover) and then
// divides; compiled into the general column code only (GENV 0 and 3), where such objectives are routed
template <int KIND, bool DIV = false>
struct builtin_term {
    static __device__ __forceinline__ double pre(int c, const double* __restrict__ p) { return p[c]; }
    // G_FUNCTION: r = RN(1 / (1 + a)), stored behind the l*k parameters by vs_objective_create
    static __device__ __forceinline__ double pre2(int c, const double* __restrict__ p, int lk) {
        return KIND == VS_OBJ_G_FUNCTION ? p[c + lk] : 0.0;
    }
    static __device__ __forceinline__ double eval(double x, int, const double* __restrict__, double pc, double r) {
        if (KIND == VS_OBJ_G_FUNCTION) {
            // (|4x - 2| + a) / (1 + a) as the correctly rounded quotient, without the division: Markstein's sequence on
            // the correctly rounded reciprocal (two steps; the second is exact for every normal n / d -- the host
            // admits only such d).  Same bits as vs_term<G>'s division; 5 fp64 instructions instead of ~13.
            const double n = fabs(4.0 * x - 2.0) + pc, d = 1.0 + pc;
            if (DIV && r == 0.0) return n / d;   // wave-uniform: a column the theorem does not cover (host: r = 0)
            double q = n * r;
            q = fma(fma(-q, d, n), r, q);
            return fma(fma(-q, d, n), r, q);
        }
        return vs_term<KIND>(x, pc);
    }
};

// Two shapes are held to 128 registers (a handful of spilled dwords) for four waves per SIMD instead of two or
// three: 16 columns per wave without the fused sums (k <= 128: two workgroups of eight waves share a CU) and the
// fused kernel at 8 columns per wave (k <= 32; measured 0.395 vs 0.407 ms at n = 2^22)
#define VS_SCAN_MIN_WAVES(L, W, FUSED, MULTI) \
    (((L) == 16 && (W) == 8 && !(FUSED) && !(MULTI)) || ((L) == 8 && (W) == 4 && (FUSED)) ? 4 : 1)

// GENV: 1 / 2 = Halton indices below 2^32 with identity / linear scaling (straight-line column code);
//       0 = any generator and scaling (fused: except power scaling); 3 = fused, power scaling
// columns per wave whose four sums the fused kernel keeps in LDS instead of registers (vs_scan_body): what the
// four-waves-per-SIMD shape is short of, so that nothing goes to scratch
#define VS_SCAN_SUMS_IN_LDS(KIND, L, W, FUSED, GENV) ((FUSED) && (L) == 8 && (W) == 4 ? VS_SCAN_SLDS_84(KIND, GENV) : 0)
// (Halton with identity / linear scaling: one column -- two for the linear objective, whose kernel was one register
// short once the folds of the segment totals were compiled per wave: 0.268 ms either way --; the general generator
// paths and the one with pow() behind a call need more room)
#ifndef VS_SCAN_SLDS_84
#define VS_SCAN_SLDS_84(KIND, GENV) ((GENV) == 1 || (GENV) == 2 || (GENV) == 4 ? ((KIND) == VS_OBJ_LINEAR ? 2 : 1) : (GENV) == 0 ? 2 : 4)
#endif

// How a scan kernel comes to its Sobol points (vs_scan_block's SNIB; the general column code only): the persistent fused
// kernels keep the nibble tables of all their columns in LDS (W L 512 B: 16 KiB at 32 columns, 48 at 96) -- except the
// one whose terms already fill it (TLDS: 97 .. 128 columns) --, the evaluation kernels, one 64-row block per workgroup,
// pass them through a KiB per wave either way
#if defined(VS_SOBOL_BYTE_TABLES)   // A/B build (tools/sobol_lds.sh): round 4's byte tables in memory everywhere
#define VS_SCAN_SNIB(FUSED, TLDS, GENV) 0
#define VS_SOBOL_INSTANCE 0          // ... and no instance of the column code for Sobol alone
#define VS_SOBOL_GENV 0
#elif defined(VS_SCAN_SNIB_ALL1)    // A/B build: the one-KiB slot everywhere
#define VS_SCAN_SNIB(FUSED, TLDS, GENV) ((GENV) == 0 || (GENV) >= 3 ? 1 : 0)
#else
#define VS_SCAN_SNIB(FUSED, TLDS, GENV) ((GENV) == 0 || (GENV) >= 3 ? ((FUSED) && !(TLDS) ? 2 : 1) : 0)
#endif

template <int KIND, int L, int W, bool FUSED, bool TLDS, bool MULTI, int GENV>
__global__ void __launch_bounds__(VS_EVAL_ROWS* W)
__attribute__((amdgpu_waves_per_eu(VS_SCAN_MIN_WAVES(L, W, FUSED, MULTI))))
scan_kernel(vs_eval_args A, const double* shift_src, double* partials, uint32_t* counters) {
    extern __shared__ double scan_lds[];  // totals slabs [2 families][W][64] (+ term slabs, + sum slabs), vs_scan_lds
    vs_scan_body<builtin_term<KIND, GENV == 0 || GENV == 3>, KIND, L, W, FUSED, TLDS, MULTI, GENV, VS_SCAN_SUMS_IN_LDS(KIND, L, W, FUSED, GENV),
                 VS_SCAN_SNIB(FUSED, TLDS, GENV)>(
        A, scan_lds, shift_src, partials);
    // counters of the fold that follows (see reduce_first_kernel).  At the END of the kernel: a global
    // store in front of the row loop would make every load of design constants "possibly clobbered"
    // and keep it out of the scalar cache.
    if (FUSED && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        counters[0] = 0;
        counters[1] = 0xffffffffu;
        counters[2] = 0;
        counters[3] = 0;
        counters[4] = 0;
    }
}

template <int KIND, int L, int W, bool FUSED, bool TLDS, bool MULTI, int GENV>
static int launch_scan_g(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev, bool plan_only) {
    size_t lds = sizeof(double) * vs_scan_lds<L, W, TLDS, VS_SCAN_SUMS_IN_LDS(KIND, L, W, FUSED, GENV)>::doubles;
    // Sobol designs: LDS for the nibble tables (vs_sobol_pipe; the general column code of these kernels has no other way
    // to a Sobol point)
    if (A.D.generator == VS_GEN_SOBOL) lds += VS_SOBOL_LDS_BYTES(VS_SCAN_SNIB(FUSED, TLDS, GENV), L, W);
    static_assert(sizeof(double) * vs_scan_lds<L, W, TLDS, VS_SCAN_SUMS_IN_LDS(KIND, L, W, FUSED, GENV)>::doubles +
                          VS_SOBOL_LDS_BYTES(VS_SCAN_SNIB(FUSED, TLDS, GENV), L, W) <= VS_LDS_BYTES, "no room for the Sobol tables");
    const void* fn = (const void*)scan_kernel<KIND, L, W, FUSED, TLDS, MULTI, GENV>;
    if (lds > 64 * 1024) VS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t blocks = (A.rows + VS_EVAL_ROWS - 1) / VS_EVAL_ROWS;
    if (plan_only) {
        *P_io = FUSED ? vs_persistent_workgroups(ctx, fn, VS_EVAL_ROWS * W, lds, blocks, A.l) : (unsigned)blocks;
        return VS_OK;
    }
    {
        vs_launch_scope scope(ctx, FUSED ? VS_KERNEL_FUSED : VS_KERNEL_EVAL);
        // outputs: grid.y (fused; evaluation without the in-kernel output loop) or inside the block walk (MULTI)
        hipLaunchKernelGGL((scan_kernel<KIND, L, W, FUSED, TLDS, MULTI, GENV>), dim3(*P_io, MULTI ? 1u : (unsigned)A.l),
                           dim3(VS_EVAL_ROWS * W), lds, ctx->stream, A, shift_dev, ctx->partials, ctx->counters);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
}

// the specialised column code for the common designs: Halton with identity or linear scaling and -- in the fused
// kernels -- every index small enough for the three-instruction digit step (vs_radical_inverse_pair_horner: below 9.49e7
// indices in all, n up to 47 million rows); larger Halton designs there, Sobol, raw points and power scaling take the
// general one
// GENV 4, the column code for Sobol designs alone (identity or linear scaling, points from LDS): what takes it
#ifndef VS_SOBOL_INSTANCE
#define VS_SOBOL_INSTANCE 1
#define VS_SOBOL_GENV 4
#endif
static inline bool vs_sobol_instance_covers(const vs_context* ctx, const vs_eval_args& A) {
    return A.D.generator == VS_GEN_SOBOL && ctx->tune.scan_general == 0 &&
           (A.D.scale_mode == VS_SCALE_IDENTITY || A.D.scale_mode == VS_SCALE_LINEAR) &&
           (A.kind != VS_OBJ_G_FUNCTION || A.plain_recip != 0);   // else: the column code that can divide
}

template <int KIND, int L, int W, bool FUSED, bool TLDS, bool MULTI>
static int launch_scan_lw(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev, bool plan_only) {
    // (the kernels that write f-values run the four-instruction digit loop -- but from 97 columns on --: any design
    // below 2^32 indices)
    const bool halton = A.D.generator == VS_GEN_HALTON && (A.D.halton_horner != 0 || !VS_SCAN_SHORT_DIGITS(FUSED, L, W)) &&
                        ctx->tune.scan_general == 0 &&
                        (A.kind != VS_OBJ_G_FUNCTION || A.plain_recip != 0);   // else: the column code that can divide
    if (halton && A.D.scale_mode == VS_SCALE_IDENTITY)
        return launch_scan_g<KIND, L, W, FUSED, TLDS, MULTI, 1>(ctx, A, P_io, shift_dev, plan_only);
    if (halton && A.D.scale_mode == VS_SCALE_LINEAR)
        return launch_scan_g<KIND, L, W, FUSED, TLDS, MULTI, 2>(ctx, A, P_io, shift_dev, plan_only);
    if (VS_SOBOL_INSTANCE && vs_sobol_instance_covers(ctx, A))
        return launch_scan_g<KIND, L, W, FUSED, TLDS, MULTI, VS_SOBOL_GENV>(ctx, A, P_io, shift_dev, plan_only);
    if constexpr (FUSED) {   // the fused kernels' general variant carries no pow(): power scaling has its own instance
        if (A.D.scale_mode == VS_SCALE_POWER)
            return launch_scan_g<KIND, L, W, FUSED, TLDS, MULTI, 3>(ctx, A, P_io, shift_dev, plan_only);
    }
    return launch_scan_g<KIND, L, W, FUSED, TLDS, MULTI, 0>(ctx, A, P_io, shift_dev, plan_only);
}

template <int KIND, bool FUSED>
static int launch_scan_kind(vs_context* ctx, const vs_eval_args& A, scan_shape s, unsigned* P_io, const double* shift_dev,
                            bool plan_only) {
    // l > 1: the outputs share one generated row inside the kernel (MULTI), except at 32 columns per wave,
    // where the coordinates no longer fit the registers beside the terms (then: one grid.y plane per output)
    const bool multi = !FUSED && A.l > 1 && s.L < 32;
#define VS_SCAN_CASE(LL, WW, TL)                                                                                       \
    if (s.L == LL && s.W == WW) {                                                                                      \
        if constexpr (!FUSED && LL < 32) {                                                                             \
            if (multi) return launch_scan_lw<KIND, LL, WW, false, false, true>(ctx, A, P_io, shift_dev, plan_only);    \
        }                                                                                                              \
        return launch_scan_lw<KIND, LL, WW, FUSED, FUSED && TL, false>(ctx, A, P_io, shift_dev, plan_only);            \
    }
    VS_SCAN_CASE(8, 1, false)
    VS_SCAN_CASE(16, 1, false)
    VS_SCAN_CASE(12, 2, false)
    VS_SCAN_CASE(8, 4, false)
    VS_SCAN_CASE(12, 4, false)
    VS_SCAN_CASE(8, 8, false)
    VS_SCAN_CASE(12, 8, false)
    VS_SCAN_CASE(16, 8, true)
    if constexpr (!FUSED) {
        VS_SCAN_CASE(32, 8, false)
    }
#undef VS_SCAN_CASE
    return vs_fail(VS_ERR_ARG, "no scan kernel for L=%d W=%d", s.L, s.W);
}

// ---- fused evaluate + both orders: four waves x L columns, tile of 16 NBH index rows per half (vs_device_scan.h)
// Sobol points (vs_scan_block's SNIB; the general column code only): the one-KiB slot per wave -- whole tables beside the
// tiles would cost resident workgroups (k <= 32: 40 000 B per workgroup, four per CU; + 16 KiB: two) or do not fit (k > 32);
// the instance of 65 .. 128 index rows with several outputs has not even the slot's 8 KiB left and keeps the byte tables
#if defined(VS_SOBOL_BYTE_TABLES)
#define VS_GRAM_SNIB(GENV, NB, LMAX) 0
#else
#define VS_GRAM_SNIB(GENV, NB, LMAX) ((GENV) == 0 || (GENV) >= 3 ? ((NB) == 8 && (LMAX) > 1 ? 0 : 1) : 0)
#endif
template <int KIND, int L, int NBH, int GENV, int LMAX = 1>
__global__ void __launch_bounds__(VS_EVAL_ROWS * 4) __attribute__((amdgpu_waves_per_eu(4)))
scan_gram_kernel(vs_eval_args A, const double* shift_src, double* first_partials, double* gram_partials,
                 uint32_t* counters) {
    extern __shared__ double scan_lds[];   // [slabs][2][4][64] totals, then the tile [32 NBH + 2 LMAX][68]
    vs_scan_gram_body<builtin_term<KIND, GENV == 0 || GENV == 3>, KIND, L, NBH, GENV, LMAX, VS_GRAM_SNIB(GENV, NBH, LMAX)>(
        A, scan_lds, shift_src, first_partials, gram_partials);
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // counters of the folds that follow, at the END (see scan_kernel)
        counters[0] = 0;
        counters[1] = 0xffffffffu;
        counters[2] = 0;
        counters[3] = 0;
        counters[4] = 0;
    }
}

template <int KIND, int L, int NBH, int GENV, int LMAX = 1>
static int launch_scan_gram_g(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                              double* first_partials, double* gram_partials, bool plan_only) {
    size_t lds = sizeof(double) * VS_GT_LDS_DOUBLES(4, NBH, LMAX);
    if (A.D.generator == VS_GEN_SOBOL) lds += VS_SOBOL_LDS_BYTES(VS_GRAM_SNIB(GENV, NBH, LMAX), L, 4);
    const void* fn = (const void*)scan_gram_kernel<KIND, L, NBH, GENV, LMAX>;
    if (lds > 64 * 1024) VS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t blocks = (A.rows + VS_EVAL_ROWS - 1) / VS_EVAL_ROWS;
    if (plan_only) {
        // twice the resident set (as the first-order fused kernels; 0.61 vs 0.65 ms at k = 32, n = 2^22) -- but every
        // workgroup leaves a Gram partial for the folds: with fewer than 32 blocks each, the resident set alone
        unsigned P = vs_persistent_workgroups(ctx, fn, VS_EVAL_ROWS * 4, lds, blocks, 1);
        if (ctx->tune.fused_wgs_per_cu == 0 && blocks < 32ull * P) P = (P + 1) / 2;
        *P_io = P;
        return VS_OK;
    }
    {
        vs_launch_scope scope(ctx, VS_KERNEL_FUSED);
        hipLaunchKernelGGL((scan_gram_kernel<KIND, L, NBH, GENV, LMAX>), dim3(*P_io), dim3(VS_EVAL_ROWS * 4), lds, ctx->stream,
                           A, shift_dev, first_partials, gram_partials, ctx->counters);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
}

template <int KIND, int L, int NBH, int LMAX = 1>
static int launch_scan_gram_lw(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                               double* first_partials, double* gram_partials, bool plan_only) {
    const bool halton = A.D.generator == VS_GEN_HALTON && A.D.halton_horner != 0 && ctx->tune.scan_general == 0 &&
                        (A.kind != VS_OBJ_G_FUNCTION || A.plain_recip != 0);   // else: the column code that can divide
#define VS_GRAM_GENV(G) \
    return launch_scan_gram_g<KIND, L, NBH, G, LMAX>(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only)
    if (halton && A.D.scale_mode == VS_SCALE_IDENTITY) VS_GRAM_GENV(1);
    if (halton && A.D.scale_mode == VS_SCALE_LINEAR) VS_GRAM_GENV(2);
    if constexpr (LMAX == 1) {   // (several outputs: the general column code)
        if (VS_SOBOL_INSTANCE && vs_sobol_instance_covers(ctx, A)) VS_GRAM_GENV(VS_SOBOL_GENV);
    }
    if (A.D.scale_mode == VS_SCALE_POWER) VS_GRAM_GENV(3);
    VS_GRAM_GENV(0);
#undef VS_GRAM_GENV
}

template <int KIND>
static int launch_scan_gram_kind(vs_context* ctx, const vs_eval_args& A, int nbh, unsigned* P_io, const double* shift_dev,
                                 double* first_partials, double* gram_partials, bool plan_only) {
    if (nbh == 1) return launch_scan_gram_lw<KIND, 4, 1>(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only);
    if (nbh == 2) return launch_scan_gram_lw<KIND, 8, 2>(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only);
    return vs_fail(VS_ERR_ARG, "the fused both-orders kernels take k <= 32");
}

// ---- beyond 32 columns: NB waves x 16 columns, tile of 16 NB index rows per half, 256 VGPRs (vs_scan_gramn_body)
template <int KIND, int GENV, int NB, int NBL, int LMAX = 1>
__global__ void __launch_bounds__(VS_EVAL_ROWS * NB) __attribute__((amdgpu_waves_per_eu(2, 2)))
scan_gramn_kernel(vs_eval_args A, const double* shift_src, double* first_partials, double* gram_partials,
                  uint32_t* counters) {
    extern __shared__ double scan_lds[];   // [slabs][2][NB][64] totals, then the tile [32 NB + 2 LMAX][68]
    vs_scan_gramn_body<builtin_term<KIND, GENV == 0 || GENV == 3>, KIND, GENV, NB, NBL, LMAX, VS_GRAM_SNIB(GENV, NB, LMAX)>(
        A, scan_lds, shift_src, first_partials, gram_partials);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        counters[0] = 0;
        counters[1] = 0xffffffffu;
        counters[2] = 0;
        counters[3] = 0;
        counters[4] = 0;
    }
}

template <int KIND, int GENV, int NB, int NBL, int LMAX = 1>
static int launch_scan_gramn_g(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                               double* first_partials, double* gram_partials, bool plan_only) {
    // one output: NB = 8: 148 544 B, one workgroup per CU; NB = 4: 74 816 B, two.  Several: a second totals slab and
    // 2 (LMAX - 1) more tile rows (NB = 8, LMAX = 4: 160 032 B of the 163 840)
    size_t lds = sizeof(double) * VS_GT_LDS_DOUBLES(NB, NB, LMAX);
    static_assert(sizeof(double) * VS_GT_LDS_DOUBLES(NB, NB, LMAX) + VS_SOBOL_LDS_BYTES(VS_GRAM_SNIB(GENV, NB, LMAX), LMAX > 1 ? 8 : 16, NB) <=
                      VS_LDS_BYTES, "the tile does not fit the LDS of a CU");
    if (A.D.generator == VS_GEN_SOBOL) lds += VS_SOBOL_LDS_BYTES(VS_GRAM_SNIB(GENV, NB, LMAX), LMAX > 1 ? 8 : 16, NB);
    const void* fn = (const void*)scan_gramn_kernel<KIND, GENV, NB, NBL, LMAX>;
    const uint64_t blocks = (A.rows + VS_EVAL_ROWS - 1) / VS_EVAL_ROWS;
    if (plan_only) {
        const uint64_t cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        const uint64_t want = cus * (NB == 8 ? 1 : 2);   // all that are resident: every workgroup leaves a Gram partial
        *P_io = (unsigned)(blocks < want ? blocks : want);
        return VS_OK;
    }
    VS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        vs_launch_scope scope(ctx, VS_KERNEL_FUSED);
        hipLaunchKernelGGL((scan_gramn_kernel<KIND, GENV, NB, NBL, LMAX>), dim3(*P_io), dim3(VS_EVAL_ROWS * NB), lds,
                           ctx->stream, A, shift_dev, first_partials, gram_partials, ctx->counters);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
}

template <int KIND, int NB, int NBL, int LMAX = 1>
static int launch_scan_gramn_nbl(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                                 double* first_partials, double* gram_partials, bool plan_only) {
    const bool halton = A.D.generator == VS_GEN_HALTON && A.D.halton_horner != 0 && ctx->tune.scan_general == 0 &&
                        (A.kind != VS_OBJ_G_FUNCTION || A.plain_recip != 0);   // else: the column code that can divide
#define VS_GRAMN_GENV(G) \
    return launch_scan_gramn_g<KIND, G, NB, NBL, LMAX>(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only)
    if (halton && A.D.scale_mode == VS_SCALE_IDENTITY) VS_GRAMN_GENV(1);
    if (halton && A.D.scale_mode == VS_SCALE_LINEAR) VS_GRAMN_GENV(2);
    // (not the body for eight live blocks, k = 113 .. 128: at its 256 registers already, the 32 that hold a block's
    // Sobol bits ahead of the column loop make it spill -- the general column code's shorter live ranges do not)
    if constexpr (LMAX == 1 && !(NB == 8 && NBL == 8)) {
        if (VS_SOBOL_INSTANCE && vs_sobol_instance_covers(ctx, A)) VS_GRAMN_GENV(VS_SOBOL_GENV);
    }
    if (A.D.scale_mode == VS_SCALE_POWER) VS_GRAMN_GENV(3);
    VS_GRAMN_GENV(0);
#undef VS_GRAMN_GENV
}

// the instance for ceil(k l / 16) live blocks (NB / 2 < live <= NB)
template <int KIND, int NB, int LMAX = 1>
static int launch_scan_gramn_kind(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                                  double* first_partials, double* gram_partials, bool plan_only) {
    const int nbl = (A.D.k * (LMAX > 1 ? A.l : 1) + 15) / 16;
#define VS_GRAMN_NBL(N)                                                                                              \
    if (nbl == N) return launch_scan_gramn_nbl<KIND, NB, N, LMAX>(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only)
    if constexpr (NB == 8) {
        VS_GRAMN_NBL(5);
        VS_GRAMN_NBL(6);
        VS_GRAMN_NBL(7);
        VS_GRAMN_NBL(8);
    } else {
        VS_GRAMN_NBL(3);
        VS_GRAMN_NBL(4);
    }
#undef VS_GRAMN_NBL
    return vs_fail(VS_ERR_ARG, "k=%d, l=%d does not belong to the %d-block fused both-orders kernels", A.D.k, A.l, NB);
}

// several outputs, up to 64 index rows per half: the four-wave kernels of nb = 1, 2 (2 / 4 columns per wave) and 4 (8)
template <int KIND>
static int launch_scan_gram_multi_kind(vs_context* ctx, const vs_eval_args& A, int nb, unsigned* P_io, const double* shift_dev,
                                       double* first_partials, double* gram_partials, bool plan_only) {
    constexpr int LM = VS_GRAM_LMAX_SMALL;
    if (nb == 1) return launch_scan_gram_lw<KIND, 2, 1, LM>(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only);
    if (nb == 2) return launch_scan_gram_lw<KIND, 4, 2, LM>(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only);
    if (nb == 4)
        return launch_scan_gramn_kind<KIND, 4, VS_GRAM_LMAX_4>(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only);
    return vs_fail(VS_ERR_ARG, "nb=%d is not a four-wave fused both-orders kernel", nb);
}

#endif  // VS_SCAN_KIND
