// How the shape of a wave-wide 16-byte load decides what HBM delivers: every pattern below moves the same bytes
// with the same number of loads in flight per wave (8 x 16 B per lane and unit), reading P planes of `rows`
// doubles exactly once.  What differs is how the 64 lanes of ONE instruction are spread:
//   8x128B: 8 planes x 128 contiguous bytes (the Gram kernels' 16-sample-row staging)   ~4.9 TB/s
//   4x256B: 4 planes x 256 B (32-sample-row staging)                                    ~6.7 TB/s
//   2x512B, 1x1KB: one or two planes                                                    ~6.5 / 7.0 TB/s
// (MI355X, 2-3 workgroups of 4 waves per CU; "depth 2" keeps two units in flight per wave.)  128-byte runs
// cap the stream however many loads are in flight; 256-byte runs are enough.  This is why the second-order
// kernels stage 32 sample rows per fetch (gram_diag_wide_kernel, gram_macro_kernel<.., WIDE>).
// hipcc --offload-arch=gfx950 -O3 tools/load_patterns.hip -o /tmp/ldpat && /tmp/ldpat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int PAT, int DEPTH>
__global__ void __launch_bounds__(256) k(const double* f, double* out, uint64_t rows, int planes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t W = (uint64_t)gridDim.x * 4;
    // unit u: PAT 0: samples [16u,16u+16) of planes [64g, 64g+64) ; PAT 1: samples [128u', ..+128) of 8 planes
    const int groups = planes / 64;
    const uint64_t units = (rows / 16) * groups;
    d2 raw[DEPTH][8];
    double s = 0;
    auto issue = [&](uint64_t u, int slot) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double* p;
            if (PAT == 0) {
                const uint64_t c = u / groups; const int g = (int)(u % groups);
                p = f + (uint64_t)(64 * g + 8 * j + (lane >> 3)) * rows + 16 * c + 2 * (lane & 7);
            } else if (PAT >= 2) {
                // PAT 2: 2 planes x 512 B per instruction (64 samples x 16 planes per unit); PAT 3: 4 planes x 256 B (32 samples x 32 planes)
                constexpr int PP = PAT == 2 ? 2 : 4;            // planes per instruction
                constexpr int UP = 8 * PP;                      // planes per unit
                constexpr int US = 1024 / UP;                   // samples per unit
                const int upg = 64 / UP;                        // units per 64-plane slab
                const uint64_t slab = u / (upg * groups); const int r = (int)(u % (upg * groups));
                const int lp = lane / (64 / PP), li = lane % (64 / PP);
                p = f + (uint64_t)(UP * r + PP * j + lp) * rows + US * slab + 2 * li;
            } else {
                // 8 consecutive units of PAT 0 hold 128 samples x 64 planes; unit u takes planes 8*(u%8).. of that slab
                const uint64_t slab = u / (8 * groups); const int r = (int)(u % (8 * groups));
                p = f + (uint64_t)(8 * r + j) * rows + 128 * slab + 2 * lane;
            }
            raw[slot][j] = __builtin_nontemporal_load((const d2*)p);
        }
    };
    uint64_t u = (uint64_t)blockIdx.x * 4 + wave;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (u + d * W < units) issue(u + d * W, d);
    for (; u < units; u += DEPTH * W) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (u + d * W < units) {
#pragma unroll
                for (int j = 0; j < 8; ++j) s += raw[d][j].x + raw[d][j].y;
                if (u + (d + DEPTH) * W < units) issue(u + (d + DEPTH) * W, d);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int PAT, int DEPTH>
void run(const double* f, double* out, uint64_t rows, int planes, int wgs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<PAT, DEPTH>), dim3(wgs), dim3(256), 0, 0, f, out, rows, planes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("planes=%d pattern %s depth %d wgs/CU=%d: %.3f ms %.0f GB/s\n", planes, PAT == 0 ? "8x128B" : PAT == 1 ? "1x1KB" : PAT == 2 ? "2x512B" : "4x256B", DEPTH, wgs / 256, best,
           8.0 * rows * planes / best / 1e6);
}
int main() {
    const uint64_t rows = 1ull << 21;
    for (int planes : {64, 256}) {
        double *f, *out;
        hipMalloc(&f, 8 * rows * planes); hipMalloc(&out, 8 * 256 * 256 * 8);
        hipMemset(f, 0, 8 * rows * planes);
        for (int w : {2, 3}) {
            run<0, 1>(f, out, rows, planes, 256 * w); run<3, 1>(f, out, rows, planes, 256 * w);
            run<2, 1>(f, out, rows, planes, 256 * w); run<1, 1>(f, out, rows, planes, 256 * w);
            run<3, 2>(f, out, rows, planes, 256 * w); run<2, 2>(f, out, rows, planes, 256 * w);
        }
        hipFree(f); hipFree(out);
    }
    return 0;
}
