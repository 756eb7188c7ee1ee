// Dispatch of the eight-wave fused evaluate + both-orders kernels (65 .. 128 columns) over the objective kinds:
// vs_scan_gram8_g.hip, vs_scan_gram8_poly.hip, vs_scan_gram8_lin.hip.
#include "vs_scan_impl.h"

int vs_scan_gram8_launch_g(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                           double* first_partials, double* gram_partials, bool plan_only);
int vs_scan_gram8_launch_poly(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                              double* first_partials, double* gram_partials, bool plan_only);
int vs_scan_gram8_launch_lin(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                             double* first_partials, double* gram_partials, bool plan_only);

int vs_scan_gram8_launch(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                         double* first_partials, double* gram_partials, bool plan_only) {
    switch (A.kind) {
        case VS_OBJ_G_FUNCTION: return vs_scan_gram8_launch_g(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only);
        case VS_OBJ_POLYNOMIAL: return vs_scan_gram8_launch_poly(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only);
        case VS_OBJ_LINEAR: return vs_scan_gram8_launch_lin(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only);
    }
    return vs_fail(VS_ERR_ARG, "objective kind %d is not separable", A.kind);
}
