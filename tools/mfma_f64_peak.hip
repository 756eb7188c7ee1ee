// Bare v_mfma_f64_16x16x4_f64 issue loop: what the fp64 matrix pipe sustains on this chip, and
// at which clock -- the MFMA-side roofline of the second-order Gram kernel.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma && /tmp/mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(double* out, long long* clk, int iters, double seed) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = seed * (1.0 + threadIdx.x * 1e-3), b = seed * (1.0 - threadIdx.x * 2e-3);  // seed 0: all-zero operands
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}

int main(int argc, char** argv) {
    const int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 1;
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    const double scale = argc > 3 ? atof(argv[3]) : 1.0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int wgs = p.multiProcessorCount * wgs_per_cu;
    double* out;
    long long* clk;
    hipMalloc(&out, sizeof(double) * wgs * 256);
    hipMalloc(&clk, sizeof(long long) * 2 * wgs);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    constexpr int NACC = 10;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(wgs), dim3(256), 0, 0, out, clk, iters, scale * (1.0 + rep));
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(2 * wgs);
        hipMemcpy(h.data(), clk, sizeof(long long) * 2 * wgs, hipMemcpyDeviceToHost);
        const double mf = (double)iters * NACC;
        const double flops = mf * 2048.0 * 4 * wgs;  // 4 waves per workgroup
        printf("wgs/CU=%d: %.3f ms, %.1f TFLOP/s fp64; per MFMA: %.1f shader clocks (clock64), %.2f ns wall => %.0f MHz effective\n",
               wgs_per_cu, ms, flops / ms / 1e9, h[0] / mf, h[1] * 10.0 / mf, (h[0] / mf) / (h[1] * 10.0 / mf) * 1e3);
    }
    return 0;
}
