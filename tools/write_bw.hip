// HBM write bandwidth of a pure store stream (what bounds the evaluation kernel's 8 B per evaluation):
// hipcc --offload-arch=gfx950 -O3 tools/write_bw.hip -o /tmp/wbw && /tmp/wbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT, int VEC>
__global__ void __launch_bounds__(256) fill(double* p, size_t n, double v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * VEC;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < n; i += stride) {
        if (VEC == 2) {
            d2 x = {v, v + 1.0};
            if (NT) __builtin_nontemporal_store(x, (d2*)(p + i));
            else *(d2*)(p + i) = x;
        } else {
            if (NT) __builtin_nontemporal_store(v, p + i);
            else p[i] = v;
        }
    }
}

// the evaluation kernel's store shape: a workgroup owns ROWS sample rows and writes one ROWS*8-byte piece
// into each of the 2k+2 planes (planes are rows*8 bytes apart); wave w writes planes w, w+4, ..
template <int ROWS>
__global__ void __launch_bounds__(256) planar(double* p, size_t rows, int planes, double v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t r0 = (size_t)blockIdx.x * ROWS;
    for (int pl = wave; pl < planes; pl += 4)
#pragma unroll
        for (int c = 0; c < ROWS / 64; ++c)
            __builtin_nontemporal_store(v + pl, p + (size_t)pl * rows + r0 + c * 64 + lane);
}

template <int ROWS>
void run_planar(double* p, size_t rows, int planes) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((planar<ROWS>), dim3((unsigned)(rows / ROWS)), dim3(256), 0, 0, p, rows, planes, 1.0 + rep);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    printf("planar, %4d-byte pieces into %d planes: %.3f ms  %.2f TB/s\n", ROWS * 8, planes, best,
           rows * planes * 8.0 / best / 1e9);
}

template <bool NT, int VEC>
void run(const char* name, double* p, size_t n, int wgs) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((fill<NT, VEC>), dim3(wgs), dim3(256), 0, 0, p, n, 1.0 + rep);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    printf("%-28s wgs=%6d: %.3f ms  %.2f TB/s\n", name, wgs, best, n * 8.0 / best / 1e9);
}

int main() {
    const size_t n = (size_t)276824064;  // doubles = one evaluation pass at k=32, n=2^22 (2.21 GB)
    double* p;
    (void)hipMalloc(&p, n * 8);
    for (int wgs : {1024, 4096, 16384, 65536}) {
        run<true, 1>("nt 8-byte stores", p, n, wgs);
        run<false, 1>("plain 8-byte stores", p, n, wgs);
        run<true, 2>("nt 16-byte stores", p, n, wgs);
        run<false, 2>("plain 16-byte stores", p, n, wgs);
    }
    run_planar<64>(p, (size_t)1 << 22, 66);
    run_planar<128>(p, (size_t)1 << 22, 66);
    run_planar<256>(p, (size_t)1 << 22, 66);
    run_planar<512>(p, (size_t)1 << 22, 66);
    (void)hipMemsetAsync(p, 0, n * 8, 0);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    (void)hipMemsetAsync(p, 0, n * 8, 0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemsetAsync: %.3f ms %.2f TB/s\n", ms, n * 8.0 / ms / 1e9);
    return 0;
}
