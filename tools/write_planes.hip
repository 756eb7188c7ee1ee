// The evaluation kernels' store pattern at large k: ROWS*8-byte pieces into P planes that lie `rows` doubles apart.
// How does the achievable write rate depend on the number of planes and on the plane pitch?
// hipcc --offload-arch=gfx950 -O3 tools/write_planes.hip -o /tmp/wpl && /tmp/wpl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int ROWS, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) planar(double* p, size_t pitch, int planes, double v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t r0 = (size_t)blockIdx.x * ROWS;
    for (int pl = wave; pl < planes; pl += WAVES)
#pragma unroll
        for (int c = 0; c < ROWS / 64; ++c)
            __builtin_nontemporal_store(v + pl, p + (size_t)pl * pitch + r0 + c * 64 + lane);
}

template <int ROWS, int WAVES>
void run(double* p, size_t rows, size_t pitch, int planes) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((planar<ROWS, WAVES>), dim3((unsigned)(rows / ROWS)), dim3(64 * WAVES), 0, 0, p, pitch, planes, 1.0 + rep);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    printf("%4d planes, rows 2^%d, pitch rows+%zu, %4d-byte pieces, %d waves/wg: %8.3f ms  %.2f TB/s\n", planes,
           (int)__builtin_ctzll(rows), pitch - rows, ROWS * 8, WAVES, best, rows * planes * 8.0 / best / 1e9);
}

int main() {
    double* p;
    const size_t total = (size_t)258 * (((size_t)1 << 24) + 4096);
    if (hipMalloc(&p, total * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int lg : {20, 22, 24}) {
        const size_t rows = (size_t)1 << lg;
        for (int planes : {66, 130, 258}) {
            run<64, 4>(p, rows, rows, planes);
            run<64, 8>(p, rows, rows, planes);
            run<128, 8>(p, rows, rows, planes);
            run<256, 8>(p, rows, rows, planes);
            run<64, 8>(p, rows, rows + 32, planes);     // planes 256 bytes off a power-of-two pitch
            run<64, 8>(p, rows, rows + 2080, planes);
        }
    }
    return 0;
}
