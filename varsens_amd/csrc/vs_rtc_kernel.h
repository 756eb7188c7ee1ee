// Kernel template compiled by hiprtc for a user objective (vs_objective_create_source).
// Text-included after vs_device_gen.h and after the generated definition of
//   #define VS_USER_L <outputs>
//   __device__ void vs_user_objective(const vs_point& x, int k, const double* p, double* out) { <body> }
// Same structure as eval_kernel: four waves share 64 design rows; phase 1 stages the scaled
// coordinates of M_1[r], M_2[r] in LDS, phase 2 evaluates the 2k+2 points of the row, wave q
// taking blocks q, q+4, ...  A point is a column select between the two LDS columns.

// VS_USER_SCRATCH = 1: the tile does not fit LDS; it lives in a per-workgroup slice of an HBM
// scratch buffer, workgroups are persistent and tile reads are L2-scoped (see vs_tile_ld in
// vs_generate.hip).
extern "C" __global__ void __launch_bounds__(VS_EVAL_ROWS* VS_EVAL_WAVES) vs_user_eval(vs_eval_args A, double* scratch) {
    extern __shared__ double lds_tile[];
    const int T = VS_EVAL_ROWS;
    const int k = A.D.k;
    const int l = A.l;
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const vs_u64 rows = A.rows;
    double* lds = VS_USER_SCRATCH ? scratch + (size_t)blockIdx.x * 2 * k * T : lds_tile;
    const vs_u64 nrowblocks = (rows + T - 1) / T;
    for (vs_u64 rb = blockIdx.x; rb < nrowblocks; rb += gridDim.x) {
    const vs_u64 lr = rb * T + lane;
    const bool valid = lr < rows;
    const vs_u64 r = A.row_begin + (valid ? lr : rows - 1);
    double* X1 = lds + lane;
    double* X2 = lds + (size_t)k * T + lane;
    // row-blocked f-values (vs_f_tile): plane (o, blk) of these 64 rows is fpitch doubles behind the one before it
    const vs_ftile ft = vs_f_tile(rows, (2 * (vs_u64)k + 2) * (vs_u64)l, A.f_shift, rb * T);
    const vs_u64 fpitch = ft.pitch;
    double* f = A.f + ft.base + lane;
    const vs_u64 plane = (2 * (vs_u64)k + 2) * fpitch;

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
        X1[(size_t)c * T] = vs_apply_scale(A.D, x1, c);
        X2[(size_t)c * T] = vs_apply_scale(A.D, x2, c);
    }
    __syncthreads();

    const int npoints = 2 * k + 2;   // evaluation points of a design row = blocks of the f-values
    for (int blk = q; blk < npoints; blk += VS_EVAL_WAVES) {
        vs_point x;
        if (blk == 0) { x.a = X1; x.b = X1; x.swap = -1; }                    // M_1[r]
        else if (blk == 1) { x.a = X2; x.b = X2; x.swap = -1; }               // M_2[r]
        else if (blk < 2 + k) { x.a = X2; x.b = X1; x.swap = blk - 2; }       // N_j[i]: M_2, column i from M_1
        else { x.a = X1; x.b = X2; x.swap = blk - 2 - k; }                    // N_nj[i]: M_1, column i from M_2
        double out[VS_USER_L];
        for (int o = 0; o < VS_USER_L; ++o) out[o] = 0.0;
        vs_user_objective(x, k, A.params, out);
        if (valid)
            for (int o = 0; o < VS_USER_L; ++o)
                VS_STORE(f + ((vs_u64)o * plane + (vs_u64)blk * fpitch), out[o]);
    }
    if (!VS_USER_SCRATCH) break;
    __syncthreads();
    }
}
