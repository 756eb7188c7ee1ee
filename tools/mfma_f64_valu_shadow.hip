// How many v_fma_f64 fit in the shadow of a v_mfma_f64_16x16x4_f64 stream?  (second-order kernel:
// diagonal tiles on the vector pipe beside off-diagonal tiles on the matrix pipe)
// hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_valu_shadow.hip -o /tmp/shadow && /tmp/shadow <wgs/CU>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NV>
__global__ void __launch_bounds__(256) loop(double* out, int iters, double seed) {
    d4 acc[NM > 0 ? NM : 1];
    double v[NV > 0 ? NV : 1];
#pragma unroll
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = (d4){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = i;
    double a = seed * (1.0 + threadIdx.x * 1e-3), b = seed * (1.0 - threadIdx.x * 2e-3);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < (NM > NV ? NM : NV); ++i) {
            if (i < NM) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < (NM > 0 ? NV / NM : 1); ++r) {
                const int j = NM > 0 ? i * (NV / NM) + r : i;
                if (j < NV) v[j] = fma(a, b, v[j]);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV>
void run(int wgs, double* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop<NM, NV>), dim3(wgs), dim3(256), 0, 0, out, iters, 1.0 + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double waves = 4.0 * wgs;
    printf("MFMA/iter=%2d fma_f64/iter=%3d: %.3f ms  -> %.1f TF matrix + %.1f TF vector\n", NM, NV, best,
           waves * iters * NM * 2048.0 / best / 1e9, waves * iters * NV * 128.0 / best / 1e9);
}

int main(int argc, char** argv) {
    const int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 2;
    const int iters = 5000;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int wgs = p.multiProcessorCount * wgs_per_cu;
    double* out;
    hipMalloc(&out, sizeof(double) * wgs * 256);
    printf("workgroups per CU: %d (4 waves each)\n", wgs_per_cu);
    run<6, 0>(wgs, out, iters);
    run<0, 36>(wgs, out, iters);
    run<6, 12>(wgs, out, iters);
    run<6, 24>(wgs, out, iters);
    run<6, 36>(wgs, out, iters);
    run<6, 48>(wgs, out, iters);
    run<6, 72>(wgs, out, iters);
    return 0;
}
