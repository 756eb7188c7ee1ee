// Fused evaluate + first- and second-order reduce for the built-in separable objectives with 33 .. 64 columns, one
// output: four waves x 16 columns write s = c + d, t = c - d of every 64-row block into a
// 69 KiB LDS tile and multiply it on the fp64 matrix cores, five accumulator tiles per wave (vs_device_scan.h:
// vs_scan_gramn_body) -- compute_varsens (varsens/saltelli.py:572-622) without f-values in HBM.
// Built with -ffp-contract=off (see vs_generate.hip).
#define VS_SCAN_KIND 0   // the kernel templates of vs_scan_impl.h; all three kinds are instantiated here
#include "vs_scan_impl.h"

int vs_scan_gram4_launch(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                         double* first_partials, double* gram_partials, bool plan_only) {
    switch (A.kind) {
        case VS_OBJ_G_FUNCTION:
            return launch_scan_gramn_kind<VS_OBJ_G_FUNCTION, 4>(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only);
        case VS_OBJ_POLYNOMIAL:
            return launch_scan_gramn_kind<VS_OBJ_POLYNOMIAL, 4>(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only);
        case VS_OBJ_LINEAR:
            return launch_scan_gramn_kind<VS_OBJ_LINEAR, 4>(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only);
    }
    return vs_fail(VS_ERR_ARG, "objective kind %d is not separable", A.kind);
}
