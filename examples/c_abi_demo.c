/* The C ABI without Python: config 1 of the reference (Sobol g-function, a = 0,.5,3,9,99,99, k = 6,
 * n = 1024; varsens/tests/test_g_function.py:52-77) from a plain C99 program.
 *
 *   gcc -std=c99 -Iinclude examples/c_abi_demo.c -Lvarsens_amd -lvarsens_hip -Wl,-rpath,$PWD/varsens_amd -o /tmp/demo
 *
 * Prints E_2, var_y and the k first-order / total indices; exit code 0 when every call succeeded. */
#include <stdio.h>
#include <stdlib.h>

#include "varsens_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != 0) {                                                          \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, vs_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(void) {
    enum { K = 6, L = 1 };
    const uint64_t n = 1024;
    const double a[K] = {0.0, 0.5, 3.0, 9.0, 99.0, 99.0};
    vs_context* ctx = NULL;
    vs_design* design = NULL;
    vs_objective* objective = NULL;
    void *f = NULL, *acc = NULL, *gram = NULL, *first = NULL;
    uint32_t* perm = (uint32_t*)malloc(sizeof(uint32_t) * n);
    double out[VS_FIRST_ORDER_LEN(K, L)];
    vs_design_desc d;
    int j;

    if (vs_abi_version() != VS_ABI_VERSION) return 2;
    CHECK(vs_context_create(0, &ctx));
    CHECK(vs_seed1_permutation(perm, n));             /* numpy.random.seed(1); shuffle(M_2) */
    d.generator = VS_GEN_HALTON;
    d.k = K;
    d.n = n;
    d.first_index = 1 + 20 * K;                       /* ghalton skip of saltelli.py:83 */
    d.scale_mode = VS_SCALE_IDENTITY;
    d.flags = 0;
    d.lower = d.upper = NULL;
    d.perm = perm;
    d.raw = NULL;
    d.sobol_directions = NULL;
    d.perm_begin = 0;                                 /* the whole shuffle (a rank of a multi-GPU job passes its slice) */
    d.perm_count = n;
    CHECK(vs_design_create(ctx, &d, &design));
    CHECK(vs_objective_create(ctx, VS_OBJ_G_FUNCTION, K, L, a, K, &objective));
    CHECK(vs_alloc(ctx, sizeof(double) * 2 * (K + 1) * n * L, &f));
    CHECK(vs_alloc(ctx, sizeof(double) * VS_ACC_LEN(K, L, 1), &acc));
    CHECK(vs_alloc(ctx, sizeof(double) * VS_GRAM_LEN(K), &gram));
    CHECK(vs_alloc(ctx, sizeof(double) * VS_FIRST_ORDER_LEN(K, L), &first));
    CHECK(vs_eval(ctx, design, objective, 0, n, (double*)f));
    CHECK(vs_reduce_first_second_order(ctx, (const double*)f, n, K, L, 0, 0, 1, (double*)acc, (double*)gram));
    CHECK(vs_finish_first_order(ctx, (const double*)acc, n, K, L, 1, (double*)first));
    CHECK(vs_download(ctx, out, first, sizeof(out)));
    printf("E_2 %.15g\nvar_y %.15g\n", out[0], out[1]);
    for (j = 0; j < K; ++j) printf("sens[%d] %.15g   sens_t[%d] %.15g\n", j, out[2 + 2 * K + j], j, out[2 + 3 * K + j]);
    /* the same run in ONE kernel, no objective values in device memory (vs_eval_reduce_first_second_order) */
    {
        int ok = 0;
        CHECK(vs_eval_reduce_both_supported(ctx, design, objective, &ok));
        if (ok) {
            CHECK(vs_eval_reduce_first_second_order(ctx, design, objective, 0, n, 0, 1, (double*)acc, (double*)gram));
            CHECK(vs_finish_first_order(ctx, (const double*)acc, n, K, L, 1, (double*)first));
            CHECK(vs_download(ctx, out, first, sizeof(out)));
            for (j = 0; j < K; ++j) printf("fused_sens[%d] %.15g   fused_sens_t[%d] %.15g\n", j, out[2 + 2 * K + j], j, out[2 + 3 * K + j]);
        }
    }
    CHECK(vs_free(ctx, first));
    CHECK(vs_free(ctx, gram));
    CHECK(vs_free(ctx, acc));
    CHECK(vs_free(ctx, f));
    CHECK(vs_objective_destroy(ctx, objective));
    CHECK(vs_design_destroy(ctx, design));
    CHECK(vs_context_destroy(ctx));
    free(perm);
    return 0;
}
