// Second-order Gram sums of Varsens.compute_varsens (varsens/saltelli.py:612-622).
//
// With C = fN_j and D = fN_nj viewed as (K, rows) matrices, K = k*l, row index
// I = j*l + o, the reference needs D C^T (+ its transpose), D D^T and C C^T:
// four numpy.tensordot calls contracting the row axis.  Here: one pass over the
// 2K streams.  The Gram is cut into 32x32 tiles of index pairs (I-tile <= J-tile);
// a workgroup stages 64-row chunks of the four operand tiles (C_I, D_I, C_J, D_J)
// in LDS and every thread keeps a 2x2 register block of each of the four
// products (D_I.C_J, D_J.C_I, D_I.D_J, C_I.C_J).  fp64 vector fma bound for
// k >= 32, HBM bound below; reported separately from the first-order pass.
#include "vs_common.h"

#define VS2_TILE 32
#define VS2_CHUNK 64
#define VS2_LD (VS2_CHUNK + 1)  // padded leading dimension: conflict-free column reads
#define VS2_TILE_OUT (4 * VS2_TILE * VS2_TILE)

enum { CNT_NANFLAG = 0, CNT_FIRST_KEPT = 1, CNT_KEPT = 2, CNT2_FORCE = 8 };

struct vs_gram_args {
    const double* f;
    uint64_t rows;
    int k, l, K;
    int ntile;           // ceil(K / 32)
    uint32_t slab_wgs;   // grid.x
    double* partials;    // [slab_wgs][npairs][4][32][32]
    const uint8_t* keep;
    const uint32_t* counters;
};

__device__ __forceinline__ void pair_from_index(int p, int ntile, int& I, int& J) {
    // enumerate I <= J row by row
    I = 0;
    int rowlen = ntile;
    while (p >= rowlen) {
        p -= rowlen;
        ++I;
        --rowlen;
    }
    J = I + p;
}

template <bool MASKED>
__global__ void __launch_bounds__(256)
gram_kernel(vs_gram_args A) {
    extern __shared__ double tiles[];  // 4 x [32][VS2_LD]
    if (MASKED && A.counters[CNT_NANFLAG] == 0) return;
    double* CI = tiles;
    double* DI = CI + VS2_TILE * VS2_LD;
    double* CJ = DI + VS2_TILE * VS2_LD;
    double* DJ = CJ + VS2_TILE * VS2_LD;

    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    int I, J;
    pair_from_index(blockIdx.y, A.ntile, I, J);
    const int l = A.l, k = A.k, K = A.K;
    const uint64_t rows = A.rows;

    double dc[2][2] = {{0, 0}, {0, 0}}, cd[2][2] = {{0, 0}, {0, 0}}, dd[2][2] = {{0, 0}, {0, 0}},
           cc[2][2] = {{0, 0}, {0, 0}};

    const uint64_t nchunks = (rows + VS2_CHUNK - 1) / VS2_CHUNK;
    for (uint64_t ch = blockIdx.x; ch < nchunks; ch += A.slab_wgs) {
        const uint64_t r0 = ch * VS2_CHUNK;
        __syncthreads();  // previous chunk fully consumed
        // stage: 4 tiles x 32 index rows x 64 sample rows; thread -> (index row, sample row)
        for (int e = tid; e < VS2_TILE * VS2_CHUNK; e += 256) {
            const int ir = e >> 6, rr = e & 63;
            const uint64_t r = r0 + rr;
            const bool in = r < rows && (!MASKED || A.keep[r]);
            const int gi = I * VS2_TILE + ir, gj = J * VS2_TILE + ir;
            double ci = 0, di = 0, cj = 0, dj = 0;
            if (in && gi < K) {
                const int j = gi / l, o = gi - j * l;
                ci = A.f[((uint64_t)(2 + j) * rows + r) * l + o];
                di = A.f[((uint64_t)(2 + k + j) * rows + r) * l + o];
            }
            if (in && gj < K) {
                const int j = gj / l, o = gj - j * l;
                cj = A.f[((uint64_t)(2 + j) * rows + r) * l + o];
                dj = A.f[((uint64_t)(2 + k + j) * rows + r) * l + o];
            }
            CI[ir * VS2_LD + rr] = ci;
            DI[ir * VS2_LD + rr] = di;
            CJ[ir * VS2_LD + rr] = cj;
            DJ[ir * VS2_LD + rr] = dj;
        }
        __syncthreads();
#pragma unroll 4
        for (int rr = 0; rr < VS2_CHUNK; ++rr) {
            double ci[2], di[2], cj[2], dj[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                ci[u] = CI[(ti + 16 * u) * VS2_LD + rr];
                di[u] = DI[(ti + 16 * u) * VS2_LD + rr];
                cj[u] = CJ[(tj + 16 * u) * VS2_LD + rr];
                dj[u] = DJ[(tj + 16 * u) * VS2_LD + rr];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    dc[u][v] = fma(di[u], cj[v], dc[u][v]);
                    cd[u][v] = fma(ci[u], dj[v], cd[u][v]);
                    dd[u][v] = fma(di[u], dj[v], dd[u][v]);
                    cc[u][v] = fma(ci[u], cj[v], cc[u][v]);
                }
        }
    }
    double* out = A.partials + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * VS2_TILE_OUT;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int e = (ti + 16 * u) * VS2_TILE + (tj + 16 * v);
            out[0 * VS2_TILE * VS2_TILE + e] = dc[u][v];
            out[1 * VS2_TILE * VS2_TILE + e] = cd[u][v];
            out[2 * VS2_TILE * VS2_TILE + e] = dd[u][v];
            out[3 * VS2_TILE * VS2_TILE + e] = cc[u][v];
        }
}

// keep mask for the second-order pass (same rule as the first-order one)
__global__ void __launch_bounds__(256)
nan_mask2_kernel(const double* __restrict__ f, uint64_t rows, int k, int l, uint8_t* __restrict__ keep,
                 const uint32_t* counters) {
    if (counters[CNT_NANFLAG] == 0) return;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        bool bad = false;
        for (int b = 0; b < 2 * k + 2; ++b) {
            double x = f[((uint64_t)b * rows + r) * l];
            bad |= (x != x);
        }
        keep[r] = bad ? 0 : 1;
    }
}

// sum the slab partials (compensated) and scatter tile entries into the three
// row-major (K,K) matrices DC, DD, CC.
struct vs_gram_fold_args {
    const double* partials;
    uint32_t P;
    int K, ntile, npairs;
    double* gram;
    uint32_t* counters;
    int masked;
};

__global__ void __launch_bounds__(256)
gram_fold_kernel(vs_gram_fold_args A) {
    if (A.masked && A.counters[CNT_NANFLAG] == 0) return;
    const size_t width = (size_t)A.npairs * VS2_TILE_OUT;
    const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= width) return;
    double s = 0.0, comp = 0.0;
    for (uint32_t p = 0; p < A.P; ++p) {
        const double x = A.partials[(size_t)p * width + col];
        const double t = s + x;
        comp += (fabs(s) >= fabs(x)) ? (s - t) + x : (x - t) + s;
        s = t;
    }
    const double total = s + comp;
    if (!A.masked && total != total) atomicOr(&A.counters[CNT_NANFLAG], 1u);
    const int pair = (int)(col / VS2_TILE_OUT);
    const int rem = (int)(col % VS2_TILE_OUT);
    const int which = rem / (VS2_TILE * VS2_TILE);
    const int e = rem % (VS2_TILE * VS2_TILE);
    int I, J;
    pair_from_index(pair, A.ntile, I, J);
    const int gi = I * VS2_TILE + e / VS2_TILE, gj = J * VS2_TILE + e % VS2_TILE;
    const int K = A.K;
    if (gi >= K || gj >= K) return;
    double* DC = A.gram;
    double* DD = DC + (size_t)K * K;
    double* CC = DD + (size_t)K * K;
    switch (which) {
        case 0: DC[(size_t)gi * K + gj] = total; break;                       // D_gi . C_gj
        case 1: if (I != J) DC[(size_t)gj * K + gi] = total; break;            // C_gi . D_gj = DC[gj][gi]
        case 2: DD[(size_t)gi * K + gj] = total; if (I != J) DD[(size_t)gj * K + gi] = total; break;
        case 3: CC[(size_t)gi * K + gj] = total; if (I != J) CC[(size_t)gj * K + gi] = total; break;
    }
}

__global__ void __launch_bounds__(256)
finish_second_kernel(const double* __restrict__ gram, const double* __restrict__ first, uint64_t n, int k, int l,
                     double* __restrict__ s2, double* __restrict__ s2n) {
    const int K = k * l;
    const double* DC = gram;
    const double* DD = DC + (size_t)K * K;
    const double* CC = DD + (size_t)K * K;
    const double* E_2 = first;
    const double* var_y = first + l;
    const double div = 2.0 * ((double)n - 1.0);
    const size_t total = (size_t)K * K;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t I = e / K, J = e % K;
        const int p = (int)(J % l);  // output index of the LAST axis: E_2 / var_y broadcast over it
        double a = DC[I * K + J] + DC[J * K + I];   // saltelli.py:612-613
        a /= div;                                    // :614
        a -= E_2[p];                                 // :615
        a /= var_y[p];                               // :616
        s2[e] = a;
        double b = DD[I * K + J] + CC[I * K + J];    // :618-619
        b /= div;
        b -= E_2[p];
        b /= var_y[p];
        s2n[e] = b;
    }
}

extern "C" int vs_reduce_second_order(vs_context* ctx, const double* f_dev, uint64_t rows, int k, int l,
                                      int flags, double* gram_dev) {
    VS_REQUIRE(ctx && gram_dev, "null argument");
    VS_REQUIRE(k >= 1 && l >= 1, "k and l must be positive");
    VS_REQUIRE(rows == 0 || f_dev, "f_dev is null");
    VS_REQUIRE((uint64_t)k * l <= 4096, "k*l=%llu too large for the second-order pass (limit 4096)",
               (unsigned long long)k * l);
    const int K = k * l;
    const int ntile = (K + VS2_TILE - 1) / VS2_TILE;
    const int npairs = ntile * (ntile + 1) / 2;
    uint64_t nchunks = (rows + VS2_CHUNK - 1) / VS2_CHUNK;
    if (nchunks < 1) nchunks = 1;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    uint64_t want = ((uint64_t)cus * 2 + npairs - 1) / npairs;
    if (want < 1) want = 1;
    const uint32_t slab_wgs = (uint32_t)(nchunks < want ? nchunks : want);
    const size_t width = (size_t)npairs * VS2_TILE_OUT;
    int rc = vs_workspace_partials(ctx, sizeof(double) * (size_t)slab_wgs * width);
    if (rc) return rc;

    ctx->counters_host[8] = 0;
    VS_HIP(hipMemcpyAsync(ctx->counters + CNT_NANFLAG, ctx->counters_host + 8, sizeof(uint32_t),
                          hipMemcpyHostToDevice, ctx->stream));
    VS_HIP(hipMemsetAsync(gram_dev, 0, sizeof(double) * 3 * (size_t)K * K, ctx->stream));

    vs_gram_args A;
    A.f = f_dev;
    A.rows = rows;
    A.k = k;
    A.l = l;
    A.K = K;
    A.ntile = ntile;
    A.slab_wgs = slab_wgs;
    A.partials = ctx->partials;
    A.keep = nullptr;
    A.counters = ctx->counters;
    vs_gram_fold_args F;
    F.partials = ctx->partials;
    F.P = slab_wgs;
    F.K = K;
    F.ntile = ntile;
    F.npairs = npairs;
    F.gram = gram_dev;
    F.counters = ctx->counters;
    const size_t lds = sizeof(double) * 4 * VS2_TILE * VS2_LD;
    VS_HIP(hipFuncSetAttribute((const void*)gram_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    VS_HIP(hipFuncSetAttribute((const void*)gram_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid(slab_wgs, npairs);
    for (int pass = 0; pass < 2; ++pass) {
        const bool masked = pass == 1;
        if (masked) {
            if (!(flags & VS_REDUCE_NAN_TRIM) || rows == 0) break;
            rc = vs_workspace_keep(ctx, rows);
            if (rc) return rc;
            A.keep = ctx->keep;
            vs_launch_scope scope(ctx, VS_KERNEL_NAN_MASK);
            hipLaunchKernelGGL(nan_mask2_kernel, dim3((unsigned)((rows + 255) / 256 < 4096 ? (rows + 255) / 256 : 4096)),
                               dim3(256), 0, ctx->stream, f_dev, rows, k, l, ctx->keep, (const uint32_t*)ctx->counters);
        }
        {
            vs_launch_scope scope(ctx, VS_KERNEL_SECOND_ORDER);
            if (masked) hipLaunchKernelGGL(gram_kernel<true>, grid, dim3(256), lds, ctx->stream, A);
            else hipLaunchKernelGGL(gram_kernel<false>, grid, dim3(256), lds, ctx->stream, A);
        }
        VS_HIP(hipGetLastError());
        F.masked = masked ? 1 : 0;
        {
            vs_launch_scope scope(ctx, VS_KERNEL_SECOND_FOLD);
            hipLaunchKernelGGL(gram_fold_kernel, dim3((unsigned)((width + 255) / 256)), dim3(256), 0, ctx->stream, F);
        }
        VS_HIP(hipGetLastError());
    }
    return VS_OK;
}

extern "C" int vs_finish_second_order(vs_context* ctx, const double* gram_dev, const double* first_dev,
                                      uint64_t n, int k, int l, double* sens2_dev, double* sens2n_dev) {
    VS_REQUIRE(ctx && gram_dev && first_dev && sens2_dev && sens2n_dev, "null argument");
    VS_REQUIRE(k >= 1 && l >= 1 && n >= 2, "bad dimensions");
    const size_t total = (size_t)k * l * k * l;
    {
        vs_launch_scope scope(ctx, VS_KERNEL_FINISH);
        hipLaunchKernelGGL(finish_second_kernel, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)),
                           dim3(256), 0, ctx->stream, gram_dev, first_dev, n, k, l, sens2_dev, sens2n_dev);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
}
