#include <hip/hip_runtime.h>
#include <cstdio>
// The evaluation kernel's store pattern (64 rows per workgroup, one 512-byte piece per plane) into a ROW-BLOCKED layout
// [row block of RB rows][plane][RB rows] instead of the planar [plane][all rows]: how much would the layout buy?
// Measured: planar 5.3-5.65 TB/s (tools/write_xcd_map.hip); RB = 1024: 6.0-6.1; RB = 4096: 6.0-6.5; RB = 64 (one
// contiguous 33 KB region per workgroup): 5.8; a linear one-shot fill: 6.8.  DESIGN.md 7.
// hipcc --offload-arch=gfx950 -O3 tools/write_blocked_layout.hip -o /tmp/wb && /tmp/wb
template <int RB>
__global__ void __launch_bounds__(256) blocked(double* p, size_t rows, int planes, double v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t rb = blockIdx.x;                 // 64-row block
    const size_t big = rb / (RB / 64), sub = rb % (RB / 64);
    double* base = p + big * (size_t)planes * RB + sub * 64 + lane;
    for (int pl = wave; pl < planes; pl += 4) __builtin_nontemporal_store(v + pl, base + (size_t)pl * RB);
}
template <int RB> void run(double* p, size_t rows, int planes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((blocked<RB>), dim3((unsigned)(rows / 64)), dim3(256), 0, 0, p, rows, planes, 1.0 + rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    printf("planes=%3d rows=2^%d row block %5d: %.3f ms  %.2f TB/s\n", planes, (int)__builtin_ctzll(rows), RB, best, rows * planes * 8.0 / best / 1e9);
}
int main() {
    double* p; hipMalloc(&p, (size_t)258 * (1 << 24) * 8);
    for (int planes : {66, 258}) for (size_t rows : {(size_t)1 << 22, (size_t)1 << 24}) {
        run<64>(p, rows, planes); run<128>(p, rows, planes); run<256>(p, rows, planes); run<512>(p, rows, planes);
        run<1024>(p, rows, planes); run<4096>(p, rows, planes); run<65536>(p, rows, planes);
    }
    return 0;
}
