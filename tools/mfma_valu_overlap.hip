// Do fp64 MFMA (v_mfma_f64_16x16x4_f64) and fp64 VALU (v_fma_f64) work overlap on gfx950, or do they
// queue for the same double-precision units?  Decides whether a kernel that evaluates the objective
// (fp64 VALU-bound) AND accumulates the second-order Gram sums (fp64 MFMA-bound) can beat the two
// kernels run one after the other.
//   mode 0: MFMA only        mode 1: FMA only        mode 2: both, interleaved inside every wave
//   mode 3: both, split by wave (waves 0-3 of a workgroup MFMA, waves 4-7 FMA: one of each per SIMD)
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int NACC = 8;    // independent MFMA chains per wave
constexpr int NFMA = 16;   // independent FMA chains per wave

template <int MODE>
__global__ void __launch_bounds__(512) overlap(double* out, int iters, int fma_per_mfma, double seed) {
    d4 acc[NACC];
    double v[NFMA];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NFMA; ++i) v[i] = seed * (i + 1);
    const double a = seed * (1.0 + threadIdx.x * 1e-3), b = seed * (1.0 - threadIdx.x * 2e-3);
    const double m = 1.0 + seed * 1e-9, c = seed * 1e-12;
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_fma = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        if (do_fma) {
            for (int r = 0; r < fma_per_mfma; ++r) {
#pragma unroll
                for (int i = 0; i < NFMA; ++i) v[i] = __builtin_fma(v[i], m, c);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < NFMA; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(double* out, int wgs, int iters, int fpm) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(overlap<MODE>, dim3(wgs), dim3(512), 0, 0, out, iters, fpm, 1.0 + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int wgs = p.multiProcessorCount;   // one 8-wave workgroup per CU: two waves per SIMD
    double* out;
    hipMalloc(&out, sizeof(double) * wgs * 512);
    for (int fpm : {1, 2, 4, 8}) {
        const float t0 = run<0>(out, wgs, iters, fpm), t1 = run<1>(out, wgs, iters, fpm);
        const float t2 = run<2>(out, wgs, iters, fpm), t3 = run<3>(out, wgs, iters, fpm);
        const double mf = (double)iters * NACC * 2048.0 * 8 * wgs, ff = (double)iters * fpm * NFMA * 128.0 * 8 * wgs;
        printf("fma/mfma-group=%d: mfma only %.3f ms (%.1f TF)  fma only %.3f ms (%.1f TF)  interleaved %.3f ms "
               "(sum %.3f, max %.3f)  split by wave %.3f ms (half the waves each: sum %.3f, max %.3f)\n",
               fpm, t0, mf / t0 / 1e9, t1, ff / t1 / 1e9, t2, t0 + t1, t0 > t1 ? t0 : t1, t3, (t0 + t1) / 2,
               (t0 > t1 ? t0 : t1) / 2);
    }
    return 0;
}
