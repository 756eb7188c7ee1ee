// Fused evaluate + first- and second-order reduce for the built-in separable objectives, k <= 32, one output:
// the scan kernels' generate -> evaluate walk writes s = c + d, t = c - d into an LDS tile and the same workgroup
// multiplies it on the fp64 matrix cores (vs_device_scan.h: vs_scan_gram_body) -- compute_varsens
// (varsens/saltelli.py:572-622) without a single f-value in HBM.
// Built with -ffp-contract=off (see vs_generate.hip).
#define VS_SCAN_KIND 0   // the kernel templates of vs_scan_impl.h; all three kinds are instantiated here
#include "vs_scan_impl.h"

int vs_scan_gram_launch(vs_context* ctx, const vs_eval_args& A, int nbh, unsigned* P_io, const double* shift_dev,
                        double* first_partials, double* gram_partials, bool plan_only) {
    switch (A.kind) {
        case VS_OBJ_G_FUNCTION:
            return launch_scan_gram_kind<VS_OBJ_G_FUNCTION>(ctx, A, nbh, P_io, shift_dev, first_partials, gram_partials, plan_only);
        case VS_OBJ_POLYNOMIAL:
            return launch_scan_gram_kind<VS_OBJ_POLYNOMIAL>(ctx, A, nbh, P_io, shift_dev, first_partials, gram_partials, plan_only);
        case VS_OBJ_LINEAR:
            return launch_scan_gram_kind<VS_OBJ_LINEAR>(ctx, A, nbh, P_io, shift_dev, first_partials, gram_partials, plan_only);
    }
    return vs_fail(VS_ERR_ARG, "objective kind %d is not separable", A.kind);
}

// ---- several outputs: vs_scan_gramm_g.hip, vs_scan_gramm_poly.hip, vs_scan_gramm_lin.hip
int vs_scan_gram_multi_launch_g(vs_context* ctx, const vs_eval_args& A, int nb, unsigned* P_io, const double* shift_dev,
                                double* first_partials, double* gram_partials, bool plan_only);
int vs_scan_gram_multi_launch_poly(vs_context* ctx, const vs_eval_args& A, int nb, unsigned* P_io, const double* shift_dev,
                                   double* first_partials, double* gram_partials, bool plan_only);
int vs_scan_gram_multi_launch_lin(vs_context* ctx, const vs_eval_args& A, int nb, unsigned* P_io, const double* shift_dev,
                                  double* first_partials, double* gram_partials, bool plan_only);

int vs_scan_gram_multi_lmax(int nb) {
    return nb == 8 ? VS_GRAM_LMAX_8 : nb == 4 ? VS_GRAM_LMAX_4 : (nb == 1 || nb == 2) ? VS_GRAM_LMAX_SMALL : 0;
}

int vs_scan_gram_multi_launch(vs_context* ctx, const vs_eval_args& A, int nb, unsigned* P_io, const double* shift_dev,
                              double* first_partials, double* gram_partials, bool plan_only) {
    VS_REQUIRE(A.l >= 2 && A.l <= vs_scan_gram_multi_lmax(nb), "l=%d outputs do not fit the %d-block kernel", A.l, nb);
    switch (A.kind) {
        case VS_OBJ_G_FUNCTION:
            return vs_scan_gram_multi_launch_g(ctx, A, nb, P_io, shift_dev, first_partials, gram_partials, plan_only);
        case VS_OBJ_POLYNOMIAL:
            return vs_scan_gram_multi_launch_poly(ctx, A, nb, P_io, shift_dev, first_partials, gram_partials, plan_only);
        case VS_OBJ_LINEAR:
            return vs_scan_gram_multi_launch_lin(ctx, A, nb, P_io, shift_dev, first_partials, gram_partials, plan_only);
    }
    return vs_fail(VS_ERR_ARG, "objective kind %d is not separable", A.kind);
}
