// Fused evaluate + first- and second-order reduce for the built-in separable objectives with 65 .. 128 columns, one
// output (BASELINE configs[4]): eight waves x 16 columns write s = c + d, t = c - d of every 64-row block into a
// 137 KiB LDS tile and multiply it on the fp64 matrix cores, nine accumulator tiles per wave (vs_device_scan.h:
// vs_scan_gramn_body) -- compute_varsens (varsens/saltelli.py:572-622) without the 34.6 GB of f-values.
// One translation unit per objective kind (16 kernels each: 5 .. 8 live blocks x 4 generator / scaling variants).
// Built with -ffp-contract=off (see vs_generate.hip).
#define VS_SCAN_KIND VS_OBJ_LINEAR
#include "vs_scan_impl.h"

int vs_scan_gram8_launch_lin(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                            double* first_partials, double* gram_partials, bool plan_only) {
    return launch_scan_gramn_kind<VS_SCAN_KIND, 8>(ctx, A, P_io, shift_dev, first_partials, gram_partials, plan_only);
}
