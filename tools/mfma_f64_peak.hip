// Bare v_mfma_f64_16x16x4_f64 issue loop: what the fp64 matrix pipe sustains on this chip, by the number of
// independent accumulator chains per wave and of waves per SIMD -- the MFMA-side roofline of the second-order
// Gram kernels.
//
// The accumulator register class matters.  Build it twice on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_f64_peak.hip -o /tmp/mfma_v && /tmp/mfma_v
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_a && /tmp/mfma_a
// With the accumulators in VGPRs (what the Gram kernels get: llvm-objdump shows v[..] destinations) one wave
// with ONE dependent chain already sustains 69 TFLOP/s and anything more 77.6-78.2 (the data sheet's 78.6 at
// 2.4 GHz: one MFMA per 64 clocks per SIMD).  With the accumulators in AGPRs -- the compiler's choice for a
// 256-thread kernel that leaves it the register budget; 512 threads and more get VGPRs by themselves, so only
// the waves/SIMD=1 rows of the second build differ -- a wave stops at 35 TFLOP/s however many chains it has, and
// several 256-thread workgroups per CU reach 47-49 (round 1's version of this tool).  That "47-49 TFLOP/s
// measured issue ceiling" of rounds 1-2 was this artefact, not a property of the pipe.
// Operands are constants here, so the loop draws little power and runs at 2.4 GHz; the Gram kernels, streaming
// HBM beside the matrix pipe, were measured at 2.0-2.1 GHz (clock64 / wall_clock64 inside the kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int THREADS>
__global__ void __launch_bounds__(THREADS) mfma_loop(double* out, long long* clk, int iters, double seed) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    const double a = seed * (1.0 + threadIdx.x * 1e-3), b = seed * (1.0 - threadIdx.x * 2e-3);
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
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = w1 - w0;
    }
}

template <int NACC, int THREADS>
void run(double* out, long long* clk, int wgs, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<NACC, THREADS>), dim3(wgs), dim3(THREADS), 0, 0, out, clk, iters, 1.0 + rep);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    long long h[2];
    (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double flops = (double)iters * NACC * 2048.0 * (THREADS / 64) * wgs;
    printf("chains/wave=%2d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s fp64  (%.0f MHz shader clock)\n", NACC, THREADS / 256, best,
           flops / best / 1e9, (double)h[0] / (h[1] * 10.0) * 1e3);
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int wgs = p.multiProcessorCount;   // one workgroup per CU
    double* out;
    long long* clk;
    (void)hipMalloc(&out, 8 * wgs * 1024);
    (void)hipMalloc(&clk, 16);
    run<1, 256>(out, clk, wgs, 20000);
    run<2, 256>(out, clk, wgs, 20000);
    run<4, 256>(out, clk, wgs, 20000);
    run<10, 256>(out, clk, wgs, 8000);
    run<1, 512>(out, clk, wgs, 20000);
    run<4, 512>(out, clk, wgs, 8000);
    run<10, 512>(out, clk, wgs, 8000);
    run<10, 768>(out, clk, wgs, 8000);
    run<4, 1024>(out, clk, wgs, 8000);
    return 0;
}
