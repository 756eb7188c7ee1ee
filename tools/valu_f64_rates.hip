// Issue cost of the fp64 vector instructions the Halton digit loop is made of (cycles per wave
// instruction on one SIMD with 4 waves resident):
// hipcc --offload-arch=gfx950 -O3 tools/valu_f64_rates.hip -o /tmp/rates && /tmp/rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(x) x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) loop(double* out, int iters, double seed) {
    double a = seed + threadIdx.x * 1e-3, b = 1.0000001, c = 0.3, d = seed * 0.7 + threadIdx.x, e = a + 1, f = a + 2, g = a + 3,
           h = a + 4;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP8(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %3, %3, %1, %2\n v_fma_f64 %4, %4, %1, %2\n v_fma_f64 %5, %5, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 1) { REP8(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %3, %3, %1\n v_add_f64 %4, %4, %1\n v_add_f64 %5, %5, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 2) { REP8(asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %3, %3, %1\n v_mul_f64 %4, %4, %1\n v_mul_f64 %5, %5, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 3) { REP8(asm volatile("v_floor_f64 %0, %0\n v_floor_f64 %3, %3\n v_floor_f64 %4, %4\n v_floor_f64 %5, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 4) { REP8(asm volatile("v_rndne_f64 %0, %0\n v_rndne_f64 %3, %3\n v_rndne_f64 %4, %4\n v_rndne_f64 %5, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 5) { REP8(asm volatile("v_trunc_f64 %0, %0\n v_trunc_f64 %3, %3\n v_trunc_f64 %4, %4\n v_trunc_f64 %5, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
        if (OP == 6) { float x = (float)a, y = (float)d, z = (float)e, w = (float)f, m = 1.0000001f;
                       REP8(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(x), "+v"(y), "+v"(z), "+v"(w), "+v"(m));)
                       a += x; d += y; e += z; f += w; }
        if (OP == 7) { float x = (float)a, y = (float)d, z = (float)e, w = (float)f;
                       REP8(asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));)
                       a += x; d += y; e += z; f += w; }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[gridDim.x * blockDim.x] = t1 - t0;
}

template <int OP>
void run(const char* name, double* out, int wgs, int waves_per_simd) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((loop<OP>), dim3(wgs), dim3(256), 0, 0, out, iters, 1.0 + rep);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    long long clk;
    (void)hipMemcpy(&clk, (long long*)out + (size_t)wgs * 256, 8, hipMemcpyDeviceToHost);
    const double instr = (double)iters * 32;  // per wave
    const double total = instr * 4.0 * wgs;                    // wave instructions in the launch
    const double per_simd_ns = best * 1e6 / (total / (wgs / waves_per_simd * 4.0));
    printf("%-12s one wave sees %.1f shader clocks per instruction; %d waves per SIMD: %.2f ns per instruction and SIMD "
           "(%.1f clocks at 2.4 GHz), %.1f T lane-ops/s\n",
           name, clk / instr, waves_per_simd, per_simd_ns, per_simd_ns * 2.4, total * 64 / best / 1e9);
}

int main(int argc, char** argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 4;   // workgroups (4 waves each) per CU = waves per SIMD
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int wgs = p.multiProcessorCount * per_cu;
    double* out;
    (void)hipMalloc(&out, sizeof(double) * ((size_t)wgs * 256 + 8));
    run<0>("v_fma_f64", out, wgs, per_cu);
    run<1>("v_add_f64", out, wgs, per_cu);
    run<2>("v_mul_f64", out, wgs, per_cu);
    run<3>("v_floor_f64", out, wgs, per_cu);
    run<4>("v_rndne_f64", out, wgs, per_cu);
    run<5>("v_trunc_f64", out, wgs, per_cu);
    run<6>("v_fma_f32", out, wgs, per_cu);
    run<7>("v_floor_f32", out, wgs, per_cu);
    return 0;
}
