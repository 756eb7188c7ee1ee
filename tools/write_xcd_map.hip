#include <hip/hip_runtime.h>
#include <cstdio>
// Does an XCD-aware block map help the evaluation kernel's planar store pattern (64 rows per workgroup, one 512-byte
// piece into each of 2k+2 planes)?  map 0: block = blockIdx; 1: every XCD owns a contiguous eighth of the rows;
// 2, 3: 8 / 32 consecutive blocks per XCD inside groups of 64 / 256.  Measured: no -- 5.3-5.65 TB/s whatever the map
// (a linear one-shot fill reaches 6.8 TB/s, hipMemset 6.3-6.8: the planar layout, not the block order, sets the
// store ceiling; a row-blocked layout [row block][plane][rows] is the candidate remedy).
// hipcc --offload-arch=gfx950 -O3 tools/write_xcd_map.hip -o /tmp/wx && /tmp/wx
template <int MAP>
__global__ void __launch_bounds__(256) planar(double* p, size_t rows, int planes, double v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t nb = rows / 64;
    size_t rb = blockIdx.x;
    if (MAP == 1) { const size_t per = nb / 8; rb = (blockIdx.x % 8) * per + blockIdx.x / 8; }          // XCD x owns a contiguous eighth
    if (MAP == 2) { const size_t g = blockIdx.x / 64, w = blockIdx.x % 64; rb = g * 64 + (w % 8) * 8 + w / 8; }  // 8 consecutive blocks per XCD within groups of 64
    if (MAP == 3) { const size_t g = blockIdx.x / 256, w = blockIdx.x % 256; rb = g * 256 + (w % 8) * 32 + w / 8; } // 32 consecutive per XCD within groups of 256
    const size_t r0 = rb * 64;
    for (int pl = wave; pl < planes; pl += 4)
        __builtin_nontemporal_store(v + pl, p + (size_t)pl * rows + r0 + lane);
}
template <int MAP> void run(double* p, size_t rows, int planes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((planar<MAP>), dim3((unsigned)(rows / 64)), dim3(256), 0, 0, p, rows, planes, 1.0 + rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    printf("planes=%d rows=2^%d map %d: %.3f ms  %.2f TB/s\n", planes, (int)__builtin_ctzll(rows), MAP, best, rows * planes * 8.0 / best / 1e9);
}
int main() {
    double* p; hipMalloc(&p, (size_t)258 * (1 << 24) * 8);
    for (int planes : {66, 258}) for (size_t rows : {(size_t)1 << 22, (size_t)1 << 24}) {
        if (planes == 258 && rows > ((size_t)1 << 22)) { }
        run<0>(p, rows, planes); run<1>(p, rows, planes); run<2>(p, rows, planes); run<3>(p, rows, planes);
    }
    return 0;
}
