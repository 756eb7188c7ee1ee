// Scan kernels of one built-in separable objective (VS_OBJ_G_FUNCTION); see vs_scan_impl.h, vs_scan.hip.
// Built with -ffp-contract=off (see vs_generate.hip).
#define VS_SCAN_KIND VS_OBJ_G_FUNCTION
#include "vs_scan_impl.h"

int vs_scan_launch_g(vs_context* ctx, const vs_eval_args& A, scan_shape s, unsigned* P_io, const double* shift_dev,
                     bool plan_only, bool fused) {
    return fused ? launch_scan_kind<VS_SCAN_KIND, true>(ctx, A, s, P_io, shift_dev, plan_only)
                 : launch_scan_kind<VS_SCAN_KIND, false>(ctx, A, s, P_io, shift_dev, plan_only);
}
