// Kernel template compiled by hiprtc for a user SEPARABLE objective (vs_objective_create_separable):
//   f(x) = t_0(x_0) (+|*) t_1(x_1) (+|*) ... folded strictly left to right,
// text-included after vs_device_gen.h and after the generated definitions of
//   #define VS_USER_FOLD_KIND  VS_OBJ_LINEAR (sum) | VS_OBJ_POLYNOMIAL (product)
//   __device__ double vs_user_term(double x, int c, const double* p) { <body> }
// Phase 1 stores the terms t_c(M_1[r][c]), t_c(M_2[r][c]); phase 2 is the chain scheme of
// eval_kernel (prefix sharing, 8+8 chains per thread, per-wave static schedule for full passes),
// so a user model of this shape runs at the speed of the built-in g-function / linear kernels.

// By default (VS_USER_EVAL_SCAN) the evaluation takes the segmented-scan form of vs_device_scan.h
// instead (O(k) folds per row), and for k <= 128 a fused evaluate+reduce kernel vs_user_fused is built
// beside it (VS_USER_FUSED) and the fused evaluate + both-orders kernel vs_user_gram (VS_USER_GRAM);
// shapes VS_USER_*_L / _W / _TLDS / _NBH come from the host.  The chain-scheme kernel
// below serves contexts in strict evaluation order.
struct vs_user_term_t {
    static __device__ __forceinline__ double pre(int, const double* __restrict__) { return 0.0; }
    static __device__ __forceinline__ double pre2(int, const double* __restrict__, int) { return 0.0; }
    static __device__ __forceinline__ double eval(double x, int c, const double* __restrict__ p, double, double) {
        return vs_user_term(x, c, p);
    }
};

#if VS_USER_FUSED
extern "C" __global__ void __launch_bounds__(VS_EVAL_ROWS* VS_USER_FUSED_W)
vs_user_fused(vs_eval_args A, const double* shift_src, double* partials, unsigned* counters) {
    extern __shared__ double lds[];
    // (the last argument: Sobol points from nibble tables passed through a KiB of LDS per wave, which the host adds to
    // the launch for a Sobol design -- vs_sobol_points_lds, vs_device_scan.h)
    vs_scan_body<vs_user_term_t, VS_USER_FOLD_KIND, VS_USER_FUSED_L, VS_USER_FUSED_W, true, VS_USER_FUSED_TLDS, false, 0, 0, 1>(
        A, lds, shift_src, partials);
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // counters of the fold that follows (see scan_kernel)
        counters[0] = 0;
        counters[1] = 0xffffffffu;
        counters[2] = 0;
        counters[3] = 0;
        counters[4] = 0;
    }
}
#endif

#if VS_USER_GRAM   // k <= 128: evaluate + both orders in one kernel (vs_scan_gram_body / vs_scan_gramn_body, vs_device_scan.h)
#if VS_USER_GRAM_NBH <= 2
extern "C" __global__ void __launch_bounds__(VS_EVAL_ROWS * 4)
#else
extern "C" __global__ void __launch_bounds__(VS_EVAL_ROWS * VS_USER_GRAM_NBH) __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
vs_user_gram(vs_eval_args A, const double* shift_src, double* first_partials, double* gram_partials, unsigned* counters) {
    extern __shared__ double lds[];
#if VS_USER_GRAM_NBH <= 2
    vs_scan_gram_body<vs_user_term_t, VS_USER_FOLD_KIND, VS_USER_GRAM_L, VS_USER_GRAM_NBH, 0, 1, 1>(A, lds, shift_src,
                                                                                                   first_partials, gram_partials);
#else   // 33 .. 128 columns: NBH waves x 16 columns, body compiled for VS_USER_GRAM_NBL live 16-row blocks
    vs_scan_gramn_body<vs_user_term_t, VS_USER_FOLD_KIND, 0, VS_USER_GRAM_NBH, VS_USER_GRAM_NBL, 1, 1>(A, lds, shift_src,
                                                                                                      first_partials, gram_partials);
#endif
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // counters of the folds that follow (see scan_kernel)
        counters[0] = 0;
        counters[1] = 0xffffffffu;
        counters[2] = 0;
        counters[3] = 0;
        counters[4] = 0;
    }
}
#endif

#if VS_USER_EVAL_SCAN
extern "C" __global__ void __launch_bounds__(VS_EVAL_ROWS* VS_USER_EVAL_W) vs_user_eval(vs_eval_args A, double* scratch) {
    extern __shared__ double lds[];
    vs_scan_body<vs_user_term_t, VS_USER_FOLD_KIND, VS_USER_EVAL_L, VS_USER_EVAL_W, false, false, false, 0, 0, 1>(A, lds, nullptr,
                                                                                                                 nullptr);
}
#else
extern "C" __global__ void __launch_bounds__(VS_EVAL_ROWS* VS_EVAL_WAVES) vs_user_eval(vs_eval_args A, double* scratch) {
    extern __shared__ double lds[];
    constexpr int T = VS_EVAL_ROWS;
    constexpr int KIND = VS_USER_FOLD_KIND;
    const int k = A.D.k;
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const vs_u64 rows = A.rows;
    const vs_u64 lr = (vs_u64)blockIdx.x * T + lane;
    const bool valid = lr < rows;
    const vs_u64 r = A.row_begin + (valid ? lr : rows - 1);
    double* F1 = lds + lane;
    double* F2 = lds + (size_t)k * T + lane;
    // row-blocked f-values (vs_f_tile), one output: plane b of these 64 rows at f + b * fpitch
    const vs_ftile ft = vs_f_tile(rows, 2 * (vs_u64)k + 2, A.f_shift, (vs_u64)blockIdx.x * T);
    const vs_u64 fpitch = ft.pitch;
    double* f = A.f + ft.base + lane;

    const bool halton_fast = A.D.generator == VS_GEN_HALTON && A.D.bases[0].ndig != 0;
    const vs_u64 src2 = A.D.n + vs_perm_at(A.D, r);
    for (int c = q; c < k; c += VS_EVAL_WAVES) {
        double x1, x2;
        if (halton_fast) {
            vs_radical_inverse_pair((vs_u32)(A.D.first_index + r), (vs_u32)(A.D.first_index + src2), A.D.bases[c],
                                    x1, x2);
        } else {
            x1 = vs_unit_point(A.D, r, c);
            x2 = vs_unit_point(A.D, src2, c);
        }
        F1[(size_t)c * T] = vs_user_term(vs_apply_scale(A.D, x1, c), c, A.params);
        F2[(size_t)c * T] = vs_user_term(vs_apply_scale(A.D, x2, c), c, A.params);
    }
    __syncthreads();

    const double ident = KIND == VS_OBJ_LINEAR ? 0.0 : 1.0;
    double pa = ident, pb = ident;
    for (int i0 = 0; i0 < k; i0 += VS_PASS) {
        double cj[VS_CHAINS], dj[VS_CHAINS];
#pragma unroll
        for (int m = 0; m < VS_CHAINS; ++m) cj[m] = dj[m] = ident;
        const int cend = min(k, i0 + VS_PASS);
        int c = i0;
        if (cend - i0 == VS_PASS) {
            vs_chain_dispatch<KIND, false, VS_EVAL_WAVES>::run(q, F1, F2, i0, pa, pb, cj, dj);
            c = cend;
        } else {
            for (; c < cend; ++c) {  // partial pass: same schedule with run-time bounds
                const double ta = F1[(size_t)c * T], tb = F2[(size_t)c * T];
#pragma unroll
                for (int m = 0; m < VS_CHAINS; ++m) {
                    const int start = i0 + q + VS_EVAL_WAVES * m;
                    if (c > start) {
                        cj[m] = vs_fold<KIND>(cj[m], tb);
                        dj[m] = vs_fold<KIND>(dj[m], ta);
                    } else if (c == start) {
                        cj[m] = vs_fold<KIND>(pb, ta);
                        dj[m] = vs_fold<KIND>(pa, tb);
                    }
                }
                pa = vs_fold<KIND>(pa, ta);
                pb = vs_fold<KIND>(pb, tb);
            }
        }
        for (; c < k; ++c) {  // columns of later passes
            const double ta = F1[(size_t)c * T], tb = F2[(size_t)c * T];
#pragma unroll
            for (int m = 0; m < VS_CHAINS; ++m) {
                cj[m] = vs_fold<KIND>(cj[m], tb);
                dj[m] = vs_fold<KIND>(dj[m], ta);
            }
        }
#pragma unroll
        for (int m = 0; m < VS_CHAINS; ++m) {
            const int i = i0 + q + VS_EVAL_WAVES * m;
            if (i < k && valid) {
                VS_STORE(f + (vs_u64)(2 + i) * fpitch, cj[m]);
                VS_STORE(f + (vs_u64)(2 + k + i) * fpitch, dj[m]);
            }
        }
    }
    if (q == 0 && valid) {
        VS_STORE(f, pa);
        VS_STORE(f + fpitch, pb);
    }
}
#endif  // VS_USER_EVAL_SCAN
