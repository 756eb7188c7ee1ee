// Separable objectives in segmented-scan form (vs_device_scan.h): the evaluation kernel for larger k
// and the fused evaluate + first-order-reduce kernel that never writes f-values.
//
// Reference counterparts (paths relative to the reference checkout):
//   Objective evaluation loops         varsens/saltelli.py:329-355
//   E_2, var_y, U_j, U_nj sums         varsens/saltelli.py:577,583,591-596 (fused kernel)
//
// Built with -ffp-contract=off (see vs_generate.hip).
#include "vs_scan_impl.h"

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
        const vs_ftile ft = vs_f_tile(A.rows, 8, A.f_shift, lr);   // row-blocked f-values: the row's own (base, pitch)
#pragma unroll
        for (int b = 0; b < 8; ++b) VS_STORE(A.f + ft.base + (uint64_t)b * ft.pitch, v[b]);
    }
}

// fused evaluate + reduce for Ishigami: 4*3 + 5 register sums per thread; GRAM: also the second-order sums
// s_i s_j and t_i t_j (i <= j; s = c + d, t = c - d, varsens/saltelli.py:612-619) -- 12 more -- so that the default
// Varsens(...) call of BASELINE configs[2] is this one kernel.  Gram partial per workgroup: [S S^T, T T^T][16][16]
// (gram_fold_kernel's st = 2 layout, 3 x 3 entries live).
template <bool GRAM>
__global__ void __launch_bounds__(256)
fused_ishigami_kernel(vs_eval_args A, const double* shift_src, double* partials, double* gram_partials, uint32_t* counters) {
    constexpr int NS = GRAM ? 29 : 17;
    __shared__ double wsum[4][NS];
    const double pa_ = A.params[0], pb_ = A.params[1];
    const double shift = shift_src[0];
    double acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] = 0.0;
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
        if (GRAM) {
            double sv[3], tv[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                sv[i] = v[2 + i] + v[5 + i];
                tv[i] = v[2 + i] - v[5 + i];
            }
            int e = 17;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = i; j < 3; ++j) {
                    acc[GRAM ? e : 0] = fma(sv[i], sv[j], acc[GRAM ? e : 0]);
                    acc[GRAM ? e + 6 : 0] = fma(tv[i], tv[j], acc[GRAM ? e + 6 : 0]);
                    ++e;
                }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double v = vs_wave_sum_dpp(acc[i]);
        if (lane == 63) wsum[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        const double t = ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + wsum[2][threadIdx.x]) + wsum[3][threadIdx.x];
        if (threadIdx.x < 17) {
            partials[(size_t)blockIdx.x * 17 + threadIdx.x] = t;
        } else {
            // entry e - 17 of the upper triangles (0,0) (0,1) (0,2) (1,1) (1,2) (2,2), S first: both triangles of the
            // 16 x 16 block are read by the fold
            const int e = ((int)threadIdx.x - 17) % 6, pair = ((int)threadIdx.x - 17) / 6;
            const int i = e < 3 ? 0 : e < 5 ? 1 : 2, j = e < 3 ? e : e < 5 ? e - 2 : 2;
            double* g = gram_partials + ((size_t)blockIdx.x * 2 + pair) * 256;
            g[i * 16 + j] = t;
            g[j * 16 + i] = t;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // counters of the folds that follow, at the end (see scan_kernel)
        counters[0] = 0;
        counters[1] = 0xffffffffu;
        counters[2] = 0;
        counters[3] = 0;
        counters[4] = 0;
    }
}

// ---------------------------------------------------------------------------
// launch shapes (measured on MI355X, tools/scan_sweep.py; registers: 2L doubles of terms, 2(L/4 + 4) of
// suffixes, +4L sums in the fused kernel, whose terms move to LDS beyond 96 columns)
// ---------------------------------------------------------------------------
//   k <=  8: 1 wave  x  8 columns     k <= 32: 4 waves x  8     k <=  96: 8 waves x 12
//   k <= 16: 1 wave  x 16             k <= 48: 4 waves x 12     k <= 128: 8 waves x 16
//   k <= 24: 2 waves x 12             k <= 64: 8 waves x  8     k <= 256: 8 waves x 32 (evaluation only)
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

template <bool FUSED>
static int launch_scan(vs_context* ctx, const vs_eval_args& A, scan_shape s, unsigned* P_io, const double* shift_dev,
                       bool plan_only) {
    switch (A.kind) {
        case VS_OBJ_G_FUNCTION: return vs_scan_launch_g(ctx, A, s, P_io, shift_dev, plan_only, FUSED);
        case VS_OBJ_POLYNOMIAL: return vs_scan_launch_poly(ctx, A, s, P_io, shift_dev, plan_only, FUSED);
        case VS_OBJ_LINEAR: return vs_scan_launch_lin(ctx, A, s, P_io, shift_dev, plan_only, FUSED);
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

// gram_partials: null = first order only, else [P][2][16][16] for the second-order sums too
int vs_fused_ishigami(vs_context* ctx, const vs_eval_args& A, unsigned P, const double* shift_dev, double* gram_partials) {
    vs_launch_scope scope(ctx, VS_KERNEL_FUSED);
    if (gram_partials)
        hipLaunchKernelGGL(fused_ishigami_kernel<true>, dim3(P), dim3(256), 0, ctx->stream, A, shift_dev, ctx->partials,
                           gram_partials, ctx->counters);
    else
        hipLaunchKernelGGL(fused_ishigami_kernel<false>, dim3(P), dim3(256), 0, ctx->stream, A, shift_dev, ctx->partials,
                           (double*)nullptr, ctx->counters);
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
