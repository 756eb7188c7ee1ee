// Dependent-issue latency of v_fma_f64: one wave per SIMD running 1, 2, 4 or 8 independent chains
// (the Halton digit loop has two), and the same with 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/valu_f64_latency.hip -o /tmp/lat && /tmp/lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int CHAINS>
__global__ void __launch_bounds__(256) loop(double* out, int iters, double seed) {
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x * 1e-3;
    const double b = 1.0000001, c = 0.3;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32 / CHAINS; ++r) {
            if (CHAINS == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[0]) : "v"(b), "v"(c));
            if (CHAINS == 2) asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(v[0]), "+v"(v[1]) : "v"(b), "v"(c));
            if (CHAINS == 4) asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                                          : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(b), "v"(c));
            if (CHAINS == 8) asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                                          "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                                          : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(b), "v"(c));
        }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[gridDim.x * blockDim.x] = t1 - t0;
}

template <int CHAINS>
void run(double* out, int wgs, int per_cu) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((loop<CHAINS>), dim3(wgs), dim3(256), 0, 0, out, iters, 1.0 + rep);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    long long clk;
    (void)hipMemcpy(&clk, (long long*)out + (size_t)wgs * 256, 8, hipMemcpyDeviceToHost);
    const double instr = (double)iters * 32;
    printf("%d waves/SIMD, %d independent chains per wave: %.1f clocks per instruction seen by a wave, %.2f ns per instruction and SIMD\n",
           per_cu, CHAINS, clk / instr, best * 1e6 / (instr * per_cu));
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    double* out;
    (void)hipMalloc(&out, sizeof(double) * ((size_t)p.multiProcessorCount * 8 * 256 + 8));
    for (int per_cu : {1, 4}) {
        const int wgs = p.multiProcessorCount * per_cu;
        run<1>(out, wgs, per_cu);
        run<2>(out, wgs, per_cu);
        run<4>(out, wgs, per_cu);
        run<8>(out, wgs, per_cu);
    }
    return 0;
}
