// Second-order Gram sums (varsens/saltelli.py:612-619) -- placeholder replaced below.
#include "vs_common.h"

extern "C" int vs_reduce_second_order(vs_context* ctx, const double* f_dev, uint64_t rows, int k, int l,
                                      int flags, double* gram_dev) {
    (void)ctx; (void)f_dev; (void)rows; (void)k; (void)l; (void)flags; (void)gram_dev;
    return vs_fail(VS_ERR_STATE, "second-order kernel not built yet");
}

extern "C" int vs_finish_second_order(vs_context* ctx, const double* gram_dev, const double* first_dev,
                                      uint64_t n, int k, int l, double* sens2_dev, double* sens2n_dev) {
    (void)ctx; (void)gram_dev; (void)first_dev; (void)n; (void)k; (void)l; (void)sens2_dev; (void)sens2n_dev;
    return vs_fail(VS_ERR_STATE, "second-order kernel not built yet");
}
