// Fused evaluate + first- and second-order reduce for SEVERAL outputs of a built-in separable objective (VS_OBJ_LINEAR;
// 2 .. 8 outputs, k l <= 128 index rows; 2 .. 4 outputs beyond 64 rows): the generator runs once per design row, every
// output builds its terms from the same coordinates and writes s = c + d, t = c - d at index row column * l + output of
// the one LDS tile; the matrix cores multiply the whole tile, so sens_2[:, o, :, o'] for every pair of outputs comes out
// of the same pass (vs_device_scan.h: vs_scan_gram_body / vs_scan_gramn_body with LMAX > 1) -- compute_varsens
// (varsens/saltelli.py:572-622) of a multi-output model (varsens/tests/test_g_function.py:77-88) without f-values in
// HBM.  One translation unit per objective kind.  Built with -ffp-contract=off (see vs_generate.hip).
#define VS_SCAN_KIND VS_OBJ_LINEAR
#include "vs_scan_impl.h"

int vs_scan_gram_multi_launch_lin(vs_context* ctx, const vs_eval_args& A, int nb, unsigned* P_io, const double* shift_dev,
                                 double* first_partials, double* gram_partials, bool plan_only) {
    if (nb == 8)
        return launch_scan_gramn_kind<VS_SCAN_KIND, 8, VS_GRAM_LMAX_8>(ctx, A, P_io, shift_dev, first_partials, gram_partials,
                                                                      plan_only);
    return launch_scan_gram_multi_kind<VS_SCAN_KIND>(ctx, A, nb, P_io, shift_dev, first_partials, gram_partials, plan_only);
}
