// Separable objectives in segmented-scan form (vs_device_scan.h): the evaluation kernel for larger k
// and the fused evaluate + first-order-reduce kernel that never writes f-values.
//
// Reference counterparts (paths relative to the reference checkout):
//   Objective evaluation loops         varsens/saltelli.py:329-355
//   E_2, var_y, U_j, U_nj sums         varsens/saltelli.py:577,583,591-596 (fused kernel)
//
// Built with -ffp-contract=off (see vs_generate.hip).
#include <cmath>

#include "vs_common.h"

#include "vs_device_scan.h"

template <int KIND>
struct builtin_term {
    static __device__ __forceinline__ double pre(int c, const double* __restrict__ p) { return p[c]; }
    static __device__ __forceinline__ double eval(double x, int, const double* __restrict__, double pc) {
        return vs_term<KIND>(x, pc);
    }
};

// Two shapes are held to 128 registers (a handful of spilled dwords) for four waves per SIMD instead of two or
// three: 16 columns per wave without the fused sums (k <= 128: two workgroups of eight waves share a CU) and the
// fused kernel at 8 columns per wave (k <= 32; measured 0.395 vs 0.407 ms at n = 2^22)
#define VS_SCAN_MIN_WAVES(L, W, FUSED, MULTI) \
    (((L) == 16 && (W) == 8 && !(FUSED) && !(MULTI)) || ((L) == 8 && (W) == 4 && (FUSED)) ? 4 : 1)

// GENV: 1 / 2 = Halton indices below 2^32 with identity / linear scaling (straight-line column code);
//       0 = any generator and scaling
template <int KIND, int L, int W, bool FUSED, bool TLDS, bool MULTI, int GENV>
__global__ void __launch_bounds__(VS_EVAL_ROWS* W)
__attribute__((amdgpu_waves_per_eu(VS_SCAN_MIN_WAVES(L, W, FUSED, MULTI))))
scan_kernel(vs_eval_args A, const double* shift_src, double* partials, uint32_t* counters) {
    extern __shared__ double scan_lds[];  // totals slabs [2 families][W][64] (+ term slabs), vs_scan_lds
    vs_scan_body<builtin_term<KIND>, KIND, L, W, FUSED, TLDS, MULTI, GENV>(A, scan_lds, shift_src, partials);
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

// Ishigami (k = 3, not separable): thread = design row.  The 8 values of a row -- f(M_1), f(M_2), N_j[0..2],
// N_nj[0..2] -- from 6 sines; same operation order as eval_kernel<VS_OBJ_ISHIGAMI>, so the same bits.
__device__ __forceinline__ void ishigami_row(const vs_design_dev& D, uint64_t r, double pa_, double pb_, double (&v)[8]) {
    double s[2][2], p4[2];
    s[0][0] = sin(vs_m1(D, r, 0));
    s[1][0] = sin(vs_m2(D, r, 0));
    s[0][1] = sin(vs_m1(D, r, 1));
    s[1][1] = sin(vs_m2(D, r, 1));
    const double x3a = vs_m1(D, r, 2), x3b = vs_m2(D, r, 2);
    p4[0] = (x3a * x3a) * (x3a * x3a);
    p4[1] = (x3b * x3b) * (x3b * x3b);
    auto val = [&](int u0, int u1, int u2) {  // coordinate c taken from sample u_c in {0 = M_1, 1 = M_2}
        const double s1 = s[u0][0], s2 = s[u1][1];
        return (s1 + pa_ * (s2 * s2)) + (pb_ * p4[u2]) * s1;
    };
    v[0] = val(0, 0, 0);
    v[1] = val(1, 1, 1);
    v[2] = val(0, 1, 1);  // N_j[i]: M_2 except column i
    v[3] = val(1, 0, 1);
    v[4] = val(1, 1, 0);
    v[5] = val(1, 0, 0);  // N_nj[i]: M_1 except column i
    v[6] = val(0, 1, 0);
    v[7] = val(0, 0, 1);
}

// vs_eval for Ishigami: 8 coalesced 512-byte stores per wave
__global__ void __launch_bounds__(256)
ishigami_eval_kernel(vs_eval_args A) {
    const double pa_ = A.params[0], pb_ = A.params[1];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t lr = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; lr < A.rows; lr += stride) {
        double v[8];
        ishigami_row(A.D, A.row_begin + lr, pa_, pb_, v);
#pragma unroll
        for (int b = 0; b < 8; ++b) VS_STORE(A.f + (uint64_t)b * A.rows + lr, v[b]);
    }
}

// fused evaluate + first-order reduce for Ishigami: 4*3 + 5 register sums per thread
__global__ void __launch_bounds__(256)
fused_ishigami_kernel(vs_eval_args A, const double* shift_src, double* partials, uint32_t* counters) {
    __shared__ double wsum[4][17];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        counters[0] = 0;
        counters[1] = 0xffffffffu;
        counters[2] = 0;
        counters[3] = 0;
        counters[4] = 0;
    }
    const double pa_ = A.params[0], pb_ = A.params[1];
    const double shift = shift_src[0];
    double acc[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t lr = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; lr < A.rows; lr += stride) {
        double v[8];
        ishigami_row(A.D, A.row_begin + lr, pa_, pb_, v);
        const double a = v[0], b = v[1];
        const double dx = a - shift, dy = b - shift;
        acc[VS_P_SAB] = fma(a, b, acc[VS_P_SAB]);
        acc[VS_P_SA] += a;
        acc[VS_P_SB] += b;
        acc[VS_P_S1] += dx + dy;
        acc[VS_P_S2] = fma(dx, dx, acc[VS_P_S2]);
        acc[VS_P_S2] = fma(dy, dy, acc[VS_P_S2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double c = v[2 + i], d = v[5 + i];
            acc[VS_P_HEAD + 0 * 3 + i] = fma(a, c, acc[VS_P_HEAD + 0 * 3 + i]);
            acc[VS_P_HEAD + 1 * 3 + i] = fma(b, d, acc[VS_P_HEAD + 1 * 3 + i]);
            acc[VS_P_HEAD + 2 * 3 + i] = fma(a, d, acc[VS_P_HEAD + 2 * 3 + i]);
            acc[VS_P_HEAD + 3 * 3 + i] = fma(b, c, acc[VS_P_HEAD + 3 * 3 + i]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        const double v = vs_wave_sum_dpp(acc[i]);
        if (lane == 63) wsum[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 17)
        partials[(size_t)blockIdx.x * 17 + threadIdx.x] =
            ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + wsum[2][threadIdx.x]) + wsum[3][threadIdx.x];
}

// ---------------------------------------------------------------------------
// launch shapes (measured on MI355X, tools/scan_sweep.py; registers: 2L doubles of terms, 2(L/4 + 4) of
// suffixes, +4L sums in the fused kernel, whose terms move to LDS beyond 96 columns)
// ---------------------------------------------------------------------------
//   k <=  8: 1 wave  x  8 columns     k <= 32: 4 waves x  8     k <=  96: 8 waves x 12
//   k <= 16: 1 wave  x 16             k <= 48: 4 waves x 12     k <= 128: 8 waves x 16
//   k <= 24: 2 waves x 12             k <= 64: 8 waves x  8     k <= 256: 8 waves x 32 (evaluation only)
struct scan_shape {
    int L, W;
};

// variant (VS_SCAN_SHAPE / VS_FUSED_VARIANT = L*100+W, measurements only): another instantiated shape
static bool scan_shape_for(int k, bool fused, scan_shape* s, int variant = 0) {
    if (variant >= 100) {
        *s = {variant / 100, variant % 100};
        return s->L * s->W >= k;
    }
    if (k <= 8) *s = {8, 1};          // few columns: one wave per 64 rows, nothing to exchange
    else if (k <= 16) *s = {16, 1};
    else if (k <= 24) *s = {12, 2};
    else if (k <= 32) *s = {8, 4};
    else if (k <= 48) *s = {12, 4};
    else if (k <= 64) *s = {8, 8};
    else if (k <= 96) *s = {12, 8};
    else if (k <= 128) *s = {16, 8};
    else if (k <= 256 && !fused) *s = {32, 8};
    else return false;
    return true;
}

static bool scan_terms_in_lds(scan_shape s, bool fused) { return fused && s.L * s.W > 96; }

bool vs_scan_shape(int k, bool fused, int* L, int* W, int* tlds, size_t* lds_bytes) {
    scan_shape s;
    if (!scan_shape_for(k, fused, &s)) return false;
    const bool t = scan_terms_in_lds(s, fused);
    *L = s.L;
    *W = s.W;
    *tlds = t ? 1 : 0;
    const int slabs = t ? 1 : 2;
    *lds_bytes = sizeof(double) * ((size_t)slabs * 2 * s.W * VS_EVAL_ROWS + (t ? 2 * (size_t)s.L * s.W * VS_EVAL_ROWS : 0));
    return true;
}

bool vs_scan_covers(int kind, int k, bool fused) {
    scan_shape s;
    const bool separable = kind == VS_OBJ_G_FUNCTION || kind == VS_OBJ_POLYNOMIAL || kind == VS_OBJ_LINEAR;
    return separable && scan_shape_for(k, fused, &s);
}

// Workgroups of the persistent fused launch: twice what is resident at the kernel's occupancy (the
// second half takes over CU by CU as the first drains; measured better than an exact fit).
unsigned vs_persistent_workgroups(vs_context* ctx, const void* kernel, int threads, size_t lds, uint64_t blocks, int l) {
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    int resident = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kernel, threads, lds) != hipSuccess || resident < 1)
        resident = 1;
    int per_cu = 2 * resident;
    if (ctx->tune.fused_wgs_per_cu > 0) per_cu = ctx->tune.fused_wgs_per_cu;
    uint64_t P = (uint64_t)cus * per_cu;
    if (l > 1) P = (P + l - 1) / l;  // grid.y = l outputs share the chip
    if (P > blocks) P = blocks;
    return (unsigned)(P < 1 ? 1 : P);
}

template <int KIND, int L, int W, bool FUSED, bool TLDS, bool MULTI, int GENV>
static int launch_scan_g(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev, bool plan_only) {
    const size_t lds = sizeof(double) * vs_scan_lds<L, W, TLDS>::doubles;
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

// the specialised column code for the common designs: Halton (indices below 2^32: every design the scan
// kernels accept) with identity or linear scaling; Sobol, raw points and power scaling take the general one
template <int KIND, int L, int W, bool FUSED, bool TLDS, bool MULTI>
static int launch_scan_lw(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev, bool plan_only) {
    const bool halton = A.D.generator == VS_GEN_HALTON && ctx->tune.scan_general == 0;
    if (halton && A.D.scale_mode == VS_SCALE_IDENTITY)
        return launch_scan_g<KIND, L, W, FUSED, TLDS, MULTI, 1>(ctx, A, P_io, shift_dev, plan_only);
    if (halton && A.D.scale_mode == VS_SCALE_LINEAR)
        return launch_scan_g<KIND, L, W, FUSED, TLDS, MULTI, 2>(ctx, A, P_io, shift_dev, plan_only);
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

template <bool FUSED>
static int launch_scan(vs_context* ctx, const vs_eval_args& A, scan_shape s, unsigned* P_io, const double* shift_dev,
                       bool plan_only) {
    switch (A.kind) {
        case VS_OBJ_G_FUNCTION: return launch_scan_kind<VS_OBJ_G_FUNCTION, FUSED>(ctx, A, s, P_io, shift_dev, plan_only);
        case VS_OBJ_POLYNOMIAL: return launch_scan_kind<VS_OBJ_POLYNOMIAL, FUSED>(ctx, A, s, P_io, shift_dev, plan_only);
        case VS_OBJ_LINEAR: return launch_scan_kind<VS_OBJ_LINEAR, FUSED>(ctx, A, s, P_io, shift_dev, plan_only);
    }
    return vs_fail(VS_ERR_ARG, "objective kind %d is not separable", A.kind);
}

// vs_eval for a separable built-in in scan form (called by vs_eval, vs_generate.hip): one 64-row block
// per workgroup, all l outputs inside it
int vs_scan_eval(vs_context* ctx, const vs_eval_args& A) {
    scan_shape s;
    if (!scan_shape_for(A.D.k, false, &s, ctx->tune.scan_shape))
        return vs_fail(VS_ERR_ARG, "k=%d is outside the scan kernels", A.D.k);
    const uint64_t blocks = (A.rows + VS_EVAL_ROWS - 1) / VS_EVAL_ROWS;
    VS_REQUIRE(blocks <= 0x7fffffffull, "too many rows for one launch");
    unsigned P = (unsigned)blocks;
    return launch_scan<false>(ctx, A, s, &P, nullptr, false);
}

// The fused path.  Plan (P workgroups -> the caller sizes ctx->partials for [P][l][plen] records), then
// launch; the records are folded like reduce_first_kernel's.  shift_dev: the f-values of one design
// row, l planes of 2k+2 doubles; the variance shift of output o is shift_dev[o*(2k+2)] = f_o(M_1[row]).
int vs_scan_fused_plan(vs_context* ctx, const vs_eval_args& A, unsigned* P) {
    scan_shape s;
    if (!scan_shape_for(A.D.k, true, &s, ctx->tune.fused_variant))
        return vs_fail(VS_ERR_ARG, "k=%d is outside the fused kernels", A.D.k);
    return launch_scan<true>(ctx, A, s, P, nullptr, true);
}

int vs_scan_fused(vs_context* ctx, const vs_eval_args& A, unsigned P, const double* shift_dev) {
    scan_shape s;
    if (!scan_shape_for(A.D.k, true, &s, ctx->tune.fused_variant))
        return vs_fail(VS_ERR_ARG, "k=%d is outside the fused kernels", A.D.k);
    return launch_scan<true>(ctx, A, s, &P, shift_dev, false);
}

unsigned vs_fused_ishigami_workgroups(vs_context* ctx, uint64_t rows) {
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    uint64_t P = (uint64_t)cus * 12;   // 3 resident per CU (141 VGPRs), four rounds
    const uint64_t blocks = (rows + 255) / 256;
    if (P > blocks) P = blocks;
    return (unsigned)(P < 1 ? 1 : P);
}

int vs_fused_ishigami(vs_context* ctx, const vs_eval_args& A, unsigned P, const double* shift_dev) {
    vs_launch_scope scope(ctx, VS_KERNEL_FUSED);
    hipLaunchKernelGGL(fused_ishigami_kernel, dim3(P), dim3(256), 0, ctx->stream, A, shift_dev, ctx->partials,
                       ctx->counters);
    VS_HIP(hipGetLastError());
    return VS_OK;
}

// vs_eval for Ishigami (called by vs_eval, vs_generate.hip)
int vs_ishigami_eval(vs_context* ctx, const vs_eval_args& A) {
    const uint64_t blocks = (A.rows + 255) / 256;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    const uint64_t cap = (uint64_t)cus * 32;
    {
        vs_launch_scope scope(ctx, VS_KERNEL_EVAL);
        hipLaunchKernelGGL(ishigami_eval_kernel, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, ctx->stream, A);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
}
