// Prototype of a wave-private second-order kernel for 64 index rows per half (k = 64, one output): S S^T and T T^T of
// S = C + D, T = C - D on v_mfma_f64_16x16x4_f64 with the operands loaded from global memory STRAIGHT INTO THE MFMA
// FRAGMENT LAYOUT -- no LDS tile, no ds_write / ds_read, no workgroup barrier in the loop.
//
// Why it can work: in S S^T both MFMA operands are "16 index rows x 4 sample rows" fragments of the SAME matrix, lane
// (i = lane & 15, q = lane >> 4) holding S[16 I + i][sample row q-th of the step]; which four sample rows make a step is
// free (the sum over sample rows commutes).  So lane (i, q) loads 16 bytes (two sample rows) of plane 16 I + i: four
// lanes cover 64 contiguous bytes of a plane, a wave-wide load 16 planes x 64 B, and the pair of loads of a 16-row unit
// covers one 128-byte line per plane.  A wave keeps ALL 20 upper-triangle tiles (10 of S S^T, 10 of T T^T: 160 VGPRs)
// and walks its own units; two waves per SIMD, one workgroup of eight per CU, persistent.
// Per unit of 16 sample rows and wave: 16 loads of 16 B per lane (4 blocks x {c, d} x 2 halves), 32 v_add_f64 (s, t),
// 80 MFMAs.
//
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/gram_direct.hip -o /tmp/gd && /tmp/gd [log2rows=22] [fshift=11]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

#define PIN() __builtin_amdgcn_sched_barrier(0)
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NB = 4;                      // 16-row blocks per half
constexpr int NT = NB * (NB + 1) / 2;      // upper-triangle tiles per product: 10
constexpr int K = 16 * NB;

// bits: random mantissa bits per value (20: short mantissas, the products toggle fewer bits of the pipe; 52: what real
// f-values look like -- the matrix pipe draws more power and the shader clock settles lower)
__global__ void fill(double* f, size_t n, int bits) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
        f[i] = 1.0 + ldexp((double)(z >> (64 - bits)), -bits);
    }
}

// MODE 0: the product candidate.  MODE 1: no loads in the loop (operands stay what the prologue loaded): the MFMA +
// add floor of this shape.  MODE 2: loads only (no MFMAs; the adds keep the loads alive): the stream's own time.
template <int MODE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
gram_direct(const double* __restrict__ f, uint64_t rows, int fshift, uint32_t G, double* __restrict__ partials,
            long long* clk) {
    __shared__ double red[2 * NT * 256];   // 40 KiB: the workgroup's sum of the eight waves' tiles
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, q = lane >> 4;
    const uint64_t P = 2 * K + 2;
    const uint64_t units = rows >> 4;
    const uint64_t stride = (uint64_t)G * 8;
    const long long t0 = clock64(), w0 = wall_clock64();

    d4 accS[NT], accT[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) accS[t] = accT[t] = (d4){0.0, 0.0, 0.0, 0.0};
    d2 L[2][2][NB];   // [half of the unit][c, d][block]

    // half h of a unit: sample rows r0 + 8 h + 2 q, + 1 of planes 2 + 16 I + li (c) and 2 + K + 16 I + li (d)
    auto issue = [&](uint64_t unit, const int h) {
        const uint64_t r0 = unit << 4;
        const uint64_t rb = (r0 >> fshift) << fshift;
        const uint64_t left = rows - rb;
        const uint64_t pitch = left < ((uint64_t)1 << fshift) ? left : ((uint64_t)1 << fshift);
        const double* sb = f + rb * P + (r0 - rb) + 8 * h + 2 * pitch;            // wave-uniform
        const uint32_t voff = (uint32_t)li * (uint32_t)pitch + 2 * (uint32_t)q;   // per lane
#pragma unroll
        for (int src = 0; src < 2; ++src)
#pragma unroll
            for (int I = 0; I < NB; ++I)
                L[h][src][I] = __builtin_nontemporal_load((const d2*)(sb + (uint64_t)(src * K + 16 * I) * pitch + voff));
    };
    auto step = [&](const int h, const int e) {
        double s[NB], t[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            const double c = L[h][0][I][e], d = L[h][1][I][e];
            s[I] = c + d;
            t[I] = c - d;
        }
        if (MODE == 2) {
#pragma unroll
            for (int I = 0; I < NB; ++I) { accS[I][0] += s[I]; accT[I][0] += t[I]; }
            return;
        }
        int idx = 0;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = I; J < NB; ++J) {
                accS[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(s[I], s[J], accS[idx], 0, 0, 0);
                accT[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(t[I], t[J], accT[idx], 0, 0, 0);
                ++idx;
            }
    };

    uint64_t u = (uint64_t)blockIdx.x * 8 + wave;
    if (u < units) {
        issue(u, 0);
        issue(u, 1);
    }
    for (; u < units; u += stride) {
        // the last unit of a wave fetches itself again (never used): no branch around the loads
        const uint64_t un = u + stride < units ? u + stride : u;
        PIN();
        step(0, 0);
        PIN();
        step(0, 1);
        PIN();
        if (MODE != 1) issue(un, 0);
        PIN();
        step(1, 0);
        PIN();
        step(1, 1);
        PIN();
        if (MODE != 1) issue(un, 1);
    }

    // the workgroup's sum, wave after wave (a fixed order: the same bits run to run)
    for (int w = 0; w < 8; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double* ps = red + (t * 4 + v) * 64 + lane;
                    double* pt = red + ((NT + t) * 4 + v) * 64 + lane;
                    *ps = (w ? *ps : 0.0) + accS[t][v];
                    *pt = (w ? *pt : 0.0) + accT[t][v];
                }
        }
        __syncthreads();
    }
    double* out = partials + (size_t)blockIdx.x * 2 * NT * 256;
    for (int e = threadIdx.x; e < 2 * NT * 256; e += 512) out[e] = red[e];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = clock64() - t0;
        clk[1] = wall_clock64() - w0;
    }
}


// The load stream alone, by how the 64 lanes of one 16-byte load are spread over the unit (same bytes, same number of
// loads in flight per wave as the product candidate: two halves of 8):
//   MAP 0: lane = i + 16 q  (the MFMA fragment layout: 16 planes x 64 B, NEIGHBOURING LANES IN DIFFERENT PLANES)
//   MAP 1: lane = 4 i + q   (16 planes x 64 B, four neighbouring lanes contiguous)
//   MAP 2: lane = 8 p + r   (8 planes x 128 B: whole lines; a block of 16 planes = two loads)
//   MAP 3: lane = 16 p + r  (4 planes x 256 B; units of 32 sample rows, a block of 16 planes = four loads)
template <int MAP>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
stream_only(const double* __restrict__ f, uint64_t rows, int fshift, uint32_t G, double* __restrict__ partials) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t P = 2 * K + 2;
    constexpr int UROWS = MAP == 3 ? 32 : 16;
    const uint64_t units = rows / UROWS;
    const uint64_t stride = (uint64_t)G * 8;
    constexpr int NL = MAP == 3 ? 16 : 8;    // loads per half
    d2 L[2][NL];
    double acc = 0.0;
    auto issue = [&](uint64_t unit, const int h) {
        const uint64_t r0 = unit * UROWS;
        const uint64_t rb = (r0 >> fshift) << fshift;
        const uint64_t left = rows - rb;
        const uint64_t pitch = left < ((uint64_t)1 << fshift) ? left : ((uint64_t)1 << fshift);
        const double* sb = f + rb * P + (r0 - rb) + 2 * pitch;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            uint64_t plane, off;
            if (MAP == 0) {          // half h: rows 8 h + 2 q; load j: src = j >> 2, block j & 3
                plane = (j >> 2) * K + 16 * (j & 3) + (lane & 15);
                off = 8 * h + 2 * (lane >> 4);
            } else if (MAP == 1) {
                plane = (j >> 2) * K + 16 * (j & 3) + (lane >> 2);
                off = 8 * h + 2 * (lane & 3);
            } else if (MAP == 2) {   // half h: c (h = 0) or d (h = 1); load j: planes 8 j .. 8 j + 7, all 16 rows
                plane = h * K + 8 * j + (lane >> 3);
                off = 2 * (lane & 7);
            } else {                 // half h: c or d; load j: planes 4 j .. 4 j + 3, all 32 rows
                plane = h * K + 4 * j + (lane >> 4);
                off = 2 * (lane & 15);
            }
            L[h][j] = __builtin_nontemporal_load((const d2*)(sb + plane * pitch + off));
        }
    };
    auto eat = [&](const int h) {
#pragma unroll
        for (int j = 0; j < NL; ++j) acc += L[h][j][0] + L[h][j][1];
    };
    uint64_t u = (uint64_t)blockIdx.x * 8 + wave;
    if (u < units) {
        issue(u, 0);
        issue(u, 1);
    }
    for (; u < units; u += stride) {
        const uint64_t un = u + stride < units ? u + stride : u;
        PIN();
        eat(0);
        PIN();
        issue(un, 0);
        PIN();
        eat(1);
        PIN();
        issue(un, 1);
    }
    partials[(size_t)blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int MAP>
static void run_stream(const double* f, uint64_t rows, int fshift, uint32_t G, double* partials, double gb) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 8; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream_only<MAP>), dim3(G), dim3(512), 0, 0, f, rows, fshift, G, partials);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) best = ms < best ? ms : best;
    }
    static const char* names[] = {"i + 16 q: 16 planes x 64 B, lanes scattered", "4 i + q: 16 planes x 64 B, 4 lanes contiguous",
                                  "8 planes x 128 B", "4 planes x 256 B (32-row units, 16 loads per half)"};
    printf("  stream only, map %d (%s): %.4f ms = %.2f TB/s\n", MAP, names[MAP], best, gb / best);
}


// ---------------------------------------------------------------------------------------------------------------------
// The candidate that follows from the stream measurements above (whole 128-byte lines or nothing): every wave brings
// its unit of 16 sample rows x 128 planes into a PRIVATE 16 KiB LDS image by LDS-DMA (global_load_lds_dwordx4, sixteen
// pieces of 8 planes x 128 B: no vector registers, no ds_write), reads the MFMA fragments from it with ds_read_b128
// and forms s = c + d, t = c - d from the fragments.  No workgroup barrier in the loop; a wave owns all 20 tiles.
//   LDS image of a unit: [c, d][64 index rows][16 doubles], 128-byte rows, the eight 16-byte chunks of row u stored at
//   chunk position r ^ ((u >> 1) & 7) -- the swizzle is applied on the SOURCE address of the DMA (the LDS side of an
//   LDS-DMA is lane-linear) and makes the fragment reads conflict-free: a ds_read_b128 lane group {i: 0-3, 12-15 at
//   chunk q} + {i: 4-11 at chunk q + 1} meets 16 different 16-byte slots of the 256-byte bank row.
//   NBUF = 2, WAVES = 4: one wave per SIMD, the DMA of unit n + 1 runs beside the MFMAs of unit n.
//   NBUF = 1, WAVES = 8: two waves per SIMD, a wave's DMA of unit n + 1 starts when its last fragments of unit n are in
//   registers; the partner's MFMAs cover the wait.
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

template <int WAVES, int NBUF, int AUX, bool NODMA>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES / 4, WAVES / 4)))
gram_dma(const double* __restrict__ f, uint64_t rows, int fshift, uint32_t G, double* __restrict__ partials,
         long long* clk) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* const mybuf = lds + wave * NBUF * 2048;
    const uint64_t P = 2 * K + 2;
    const uint64_t units = rows >> 4;
    const uint64_t stride = (uint64_t)G * WAVES;
    const long long t0 = clock64(), w0 = wall_clock64();

    // DMA side: lane = 8 p + r': plane 8 g + p of the piece, LDS chunk position r' <- source chunk r' ^ sw(row)
    const int p = lane >> 3;
    const uint32_t chA = (uint32_t)((lane & 7) ^ (p >> 1)), chB = chA ^ 4u;   // pieces with g even / odd
    // fragment side: lane = i + 16 q reads chunk 4 h + q of row 16 I + i
    const int li = lane & 15, q = lane >> 4, sw = (li >> 1) & 7;
    const uint32_t fr0 = (uint32_t)(li * 16 + 2 * (q ^ sw)), fr1 = (uint32_t)(li * 16 + 2 * ((4 + q) ^ sw));

    d4 accS[NT], accT[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) accS[t] = accT[t] = (d4){0.0, 0.0, 0.0, 0.0};
    d2 F[2][2][NB];   // [half][c, d][block]

    auto dma = [&](uint64_t unit, double* buf) {
        const uint64_t r0 = unit << 4;
        const uint64_t rb = (r0 >> fshift) << fshift;
        const uint64_t left = rows - rb;
        const uint64_t pitch = left < ((uint64_t)1 << fshift) ? left : ((uint64_t)1 << fshift);
        const double* ub = f + rb * P + (r0 - rb) + 2 * pitch + (uint64_t)p * pitch;
#pragma unroll
        for (int src = 0; src < 2; ++src)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const double* gp = ub + (uint64_t)(src * K + 8 * g) * pitch + 2 * ((g & 1) ? chB : chA);
                __builtin_amdgcn_global_load_lds((glb_ptr)gp, (lds_ptr)(buf + (src * 8 + g) * 128), 16, 0, AUX);
            }
    };
    auto read = [&](const int h, const double* buf) {
        const double* rd = buf + (h ? fr1 : fr0);
#pragma unroll
        for (int src = 0; src < 2; ++src)
#pragma unroll
            for (int I = 0; I < NB; ++I) F[h][src][I] = *(const d2*)(rd + src * 1024 + I * 256);
    };
    auto step = [&](const int h, const int e) {
        double s[NB], t[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            const double c = F[h][0][I][e], d = F[h][1][I][e];
            s[I] = c + d;
            t[I] = c - d;
        }
        int idx = 0;
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = I; J < NB; ++J) {
                accS[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(s[I], s[J], accS[idx], 0, 0, 0);
                accT[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(t[I], t[J], accT[idx], 0, 0, 0);
                ++idx;
            }
    };
#define WAIT_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

    uint64_t u = (uint64_t)blockIdx.x * WAVES + wave;
    int b = 0;
    if (u < units) {
        dma(u, mybuf);
        WAIT_DMA();
        read(0, mybuf);
    }
    for (; u < units; u += stride) {
        const uint64_t un = u + stride < units ? u + stride : u;   // the last unit fetches itself again, unused
        double* const cur = mybuf + (NBUF == 2 ? b * 2048 : 0);
        double* const nxt = mybuf + (NBUF == 2 ? (b ^ 1) * 2048 : 0);
        if (NBUF == 2 && !NODMA) dma(un, nxt);   // its last reads were retired before the previous unit's last MFMAs
        read(1, cur);
        PIN();
        step(0, 0);
        PIN();
        step(0, 1);
        PIN();
        if (NBUF == 1 && !NODMA) {
            WAIT_LDS();                          // every fragment of this unit is in registers
            dma(un, nxt);
        }
        step(1, 0);
        PIN();
        if (!NODMA) WAIT_DMA();
        read(0, nxt);
        PIN();
        step(1, 1);
        PIN();
        b ^= 1;
    }
#undef WAIT_DMA
#undef WAIT_LDS

    __syncthreads();
    double* red = lds;
    for (int w = 0; w < WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double* ps = red + (t * 4 + v) * 64 + lane;
                    double* pt = red + ((NT + t) * 4 + v) * 64 + lane;
                    *ps = (w ? *ps : 0.0) + accS[t][v];
                    *pt = (w ? *pt : 0.0) + accT[t][v];
                }
        }
        __syncthreads();
    }
    double* out = partials + (size_t)blockIdx.x * 2 * NT * 256;
    for (int e = threadIdx.x; e < 2 * NT * 256; e += 64 * WAVES) out[e] = red[e];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = clock64() - t0;
        clk[1] = wall_clock64() - w0;
    }
}

template <int WAVES, int NBUF, int AUX, bool NODMA>
static float run_dma(const double* f, uint64_t rows, int fshift, uint32_t G, double* partials, long long* clk, int reps) {
    const size_t lds_bytes = (size_t)WAVES * NBUF * 16384;
    CK(hipFuncSetAttribute((const void*)gram_dma<WAVES, NBUF, AUX, NODMA>, hipFuncAttributeMaxDynamicSharedMemorySize,
                           (int)lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((gram_dma<WAVES, NBUF, AUX, NODMA>), dim3(G), dim3(64 * WAVES), lds_bytes, 0, f, rows, fshift, G,
                           partials, clk);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    long long h[2];
    CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    printf("  LDS-DMA, %d waves per CU, %d buffer(s) per wave, aux %d%s: best %.4f ms, mean %.4f ms, shader clock %.0f MHz\n",
           WAVES, NBUF, AUX, NODMA ? ", NO DMA IN THE LOOP (floor)" : "", best, sum / reps,
           (double)h[0] / (h[1] * 10.0) * 1e3);
    return best;
}


static void host_layout_index_fwd(uint64_t rows, int fshift, uint64_t P, uint64_t plane, uint64_t r, uint64_t& idx) {
    const uint64_t rb = (r >> fshift) << fshift;
    const uint64_t left = rows - rb;
    const uint64_t pitch = left < ((uint64_t)1 << fshift) ? left : ((uint64_t)1 << fshift);
    idx = rb * P + plane * pitch + (r - rb);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same idea at 128 index rows per half (BASELINE configs[4]), where a wave cannot own all 72 tiles: a workgroup of FOUR
// waves shares an LDS-DMA image (32 KiB per unit of 16 sample rows, two images per workgroup), TWO such workgroups per CU
// -- the two waves of a SIMD belong to different workgroups, so one workgroup's barrier (one per unit of 72 MFMAs per
// wave) does not stop the SIMD.  Waves 0, 1 take S S^T (block rows {0, 7, 1, 6} and {2, 5, 3, 4}: 18 tiles each), waves
// 2, 3 the same of T T^T; a wave forms only what it multiplies (s = c + d or t = c - d: one v_fma_f64 with a sign).
// Timing prototype for the question "what would gram_st_kernel<8, ..> gain from this structure": TIMING leaves the tiles
// in registers (one value per lane is written), VALIDATE adds them into a global matrix with atomics.
constexpr int K2 = 128, NB2 = 8;

template <int ROLE, bool VALIDATE>
__device__ __forceinline__ void dma128_body(const double* __restrict__ f, uint64_t rows, int fshift, uint32_t G,
                                            double* __restrict__ result, double* lds, const int wave, const double sgn) {
    constexpr int FIRSTB = ROLE == 0 ? 0 : 2;          // lowest block this role reads
    constexpr int ROWS[4] = {ROLE == 0 ? 0 : 2, ROLE == 0 ? 7 : 5, ROLE == 0 ? 1 : 3, ROLE == 0 ? 6 : 4};
    const int lane = threadIdx.x & 63;
    const uint64_t P = 2 * K2 + 2;
    const uint64_t units = rows >> 4;
    const int li = lane & 15, q = lane >> 4, sw = (li >> 1) & 7;
    const uint32_t fr0 = (uint32_t)(li * 16 + 2 * (q ^ sw)), fr1 = (uint32_t)(li * 16 + 2 * ((4 + q) ^ sw));
    d4 acc[18];
#pragma unroll
    for (int t = 0; t < 18; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    d2 H[2][NB2];   // [c, d][block] of one half-unit

    auto dma = [&](uint64_t unit, double* img) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int p = ln >> 3;
        const uint32_t chA = (uint32_t)((ln & 7) ^ (p >> 1)), chB = chA ^ 4u;
        const uint64_t r0 = unit << 4;
        const uint64_t rb = (r0 >> fshift) << fshift;
        const uint64_t left = rows - rb;
        const uint64_t pitch = left < ((uint64_t)1 << fshift) ? left : ((uint64_t)1 << fshift);
        const int src = wave >> 1, g0 = 8 * (wave & 1);
        const double* lb = f + rb * P + (r0 - rb) + (uint64_t)(2 + src * K2 + p) * pitch;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const double* gp = lb + (uint64_t)(8 * (g0 + g)) * pitch + 2 * ((g & 1) ? chB : chA);
            __builtin_amdgcn_global_load_lds((glb_ptr)gp, (lds_ptr)(img + src * 2048 + (g0 + g) * 128), 16, 0, 2);
        }
    };
    auto readh = [&](const int h, const double* img) {
        const double* rd = img + (h ? fr1 : fr0);
#pragma unroll
        for (int src = 0; src < 2; ++src)
#pragma unroll
            for (int I = FIRSTB; I < NB2; ++I) H[src][I] = *(const d2*)(rd + src * 2048 + I * 256);
    };
    auto step = [&](const int e) {
        double s[NB2];
#pragma unroll
        for (int I = FIRSTB; I < NB2; ++I) s[I] = fma(H[1][I][e], sgn, H[0][I][e]);
        PIN();
        int idx = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int J = ROWS[r]; J < NB2; ++J) {
                acc[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(s[ROWS[r]], s[J], acc[idx], 0, 0, 0);
                ++idx;
            }
    };
    double* const img0 = lds;
    uint64_t u = blockIdx.x;
    int b = 0;
    if (u < units) dma(u, img0);
    for (; u < units; u += G) {
        const uint64_t un = u + G < units ? u + G : u;
        double* const cur = img0 + b * 4096;
        double* const nxt = img0 + (b ^ 1) * 4096;
        __syncthreads();          // own DMA landed (vmcnt(0)), everybody's, and everybody is done with the other image
        dma(un, nxt);
        readh(0, cur);
        PIN();
        step(0);
        PIN();
        step(1);
        PIN();
        readh(1, cur);
        PIN();
        step(0);
        PIN();
        step(1);
        PIN();
        b ^= 1;
    }
    if (VALIDATE) {
        double* M = result + (sgn > 0.0 ? 0 : K2 * K2);
        int idx = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int J = ROWS[r]; J < NB2; ++J) {
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    atomicAdd(&M[(16 * ROWS[r] + (lane >> 4) + 4 * v) * K2 + 16 * J + (lane & 15)], acc[idx][v]);
                ++idx;
            }
    } else {
        double x = 0.0;
#pragma unroll
        for (int t = 0; t < 18; ++t) x += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
        result[(size_t)blockIdx.x * 256 + threadIdx.x] = x;
    }
}

template <bool VALIDATE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
gram_dma128(const double* __restrict__ f, uint64_t rows, int fshift, uint32_t G, double* __restrict__ result) {
    extern __shared__ double lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double sgn = wave < 2 ? 1.0 : -1.0;
    if (wave & 1) dma128_body<1, VALIDATE>(f, rows, fshift, G, result, lds, wave, sgn);
    else dma128_body<0, VALIDATE>(f, rows, fshift, G, result, lds, wave, sgn);
}

static int run_dma128(int log2rows, int fshift, int bits) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const uint32_t G = 2 * prop.multiProcessorCount;
    const uint64_t P = 2 * K2 + 2;
    double *f, *result;
    CK(hipFuncSetAttribute((const void*)gram_dma128<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)gram_dma128<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    {   // correctness, small shard with a short last row block
        const uint64_t rows = 3 * 2048 + 16 * 21;
        CK(hipMalloc(&f, rows * P * 8));
        CK(hipMalloc(&result, 2 * K2 * K2 * 8));
        CK(hipMemset(result, 0, 2 * K2 * K2 * 8));
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, f, rows * P, 52);
        hipLaunchKernelGGL((gram_dma128<true>), dim3(G), dim3(256), 65536, 0, f, rows, fshift, G, result);
        CK(hipDeviceSynchronize());
        std::vector<double> hf(rows * P), got(2 * K2 * K2);
        CK(hipMemcpy(hf.data(), f, hf.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(got.data(), result, got.size() * 8, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int a = 0; a < K2; a += 5)
            for (int b = a; b < K2; b += 3) {
                if ((b >> 4) < (a >> 4)) continue;
                long double ss = 0, tt = 0;
                for (uint64_t r = 0; r < rows; ++r) {
                    uint64_t ia, ib, ja, jb;
                    host_layout_index_fwd(rows, fshift, P, 2 + a, r, ia);
                    host_layout_index_fwd(rows, fshift, P, 2 + K2 + a, r, ja);
                    host_layout_index_fwd(rows, fshift, P, 2 + b, r, ib);
                    host_layout_index_fwd(rows, fshift, P, 2 + K2 + b, r, jb);
                    ss += (long double)(hf[ia] + hf[ja]) * (hf[ib] + hf[jb]);
                    tt += (long double)(hf[ia] - hf[ja]) * (hf[ib] - hf[jb]);
                }
                worst = fmax(worst, fabs(got[a * K2 + b] - (double)ss) / fabs((double)ss));
                worst = fmax(worst, fabs(got[K2 * K2 + a * K2 + b] - (double)tt) / fabs((double)ss));
            }
        printf("k = 128 candidate, correctness (rows = %llu): max rel err %.3e %s\n", (unsigned long long)rows, worst,
               worst < 1e-12 ? "OK" : "WRONG");
        CK(hipFree(f));
        CK(hipFree(result));
        if (!(worst < 1e-12)) return 2;
    }
    const uint64_t rows = (uint64_t)1 << log2rows;
    CK(hipMalloc(&f, rows * P * 8));
    CK(hipMalloc(&result, (size_t)G * 256 * 8));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, f, rows * P, bits);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    const int reps = 8;
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((gram_dma128<false>), dim3(G), dim3(256), 65536, 0, f, rows, fshift, G, result);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double gflop = 72.0 * 2048.0 * (rows / 4) / 1e9, gb = 2.0 * K2 * rows * 8 / 1e9;
    printf("k = 128, rows = 2^%d, %d mantissa bits: best %.3f ms, mean %.3f ms = %.1f TFLOP/s (%.3f of 78.6), %.2f TB/s\n",
           log2rows, bits, best, sum / reps, gflop / best, gflop / best / 78.6, gb / best);
    return 0;
}

static void host_layout_index(uint64_t rows, int fshift, uint64_t P, uint64_t plane, uint64_t r, uint64_t& idx) {
    const uint64_t rb = (r >> fshift) << fshift;
    const uint64_t left = rows - rb;
    const uint64_t pitch = left < ((uint64_t)1 << fshift) ? left : ((uint64_t)1 << fshift);
    idx = rb * P + plane * pitch + (r - rb);
}

template <int MODE>
static float run(const double* f, uint64_t rows, int fshift, uint32_t G, double* partials, long long* clk, int reps,
                 double* mhz) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((gram_direct<MODE>), dim3(G), dim3(512), 0, 0, f, rows, fshift, G, partials, clk);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    long long h[2];
    CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    *mhz = (double)h[0] / (h[1] * 10.0) * 1e3;
    printf("  mode %d: best %.4f ms, mean %.4f ms over %d launches, shader clock %.0f MHz\n", MODE, best, sum / reps, reps,
           *mhz);
    return best;
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'k') {   // ./gd k128 [log2rows=24] [fshift=11] [bits=52]
        return run_dma128(argc > 2 ? atoi(argv[2]) : 24, argc > 3 ? atoi(argv[3]) : 11, argc > 4 ? atoi(argv[4]) : 52);
    }
    const int log2rows = argc > 1 ? atoi(argv[1]) : 22;
    const int fshift = argc > 2 ? atoi(argv[2]) : 11;
    const int bits = argc > 3 ? atoi(argv[3]) : 52;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const uint32_t G = prop.multiProcessorCount;
    const uint64_t P = 2 * K + 2;
    double *f, *partials;
    long long* clk;
    CK(hipMalloc(&partials, (size_t)G * 2 * NT * 256 * 8));
    CK(hipMalloc(&clk, 16));

    // 1. correctness on a small shard whose last row block is short
    for (int which = 0; which < 3; ++which) {
        const uint64_t rows = 3 * 2048 + 16 * 37;
        CK(hipMalloc(&f, rows * P * 8));
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, f, rows * P, 52);
        if (which == 0) {
            hipLaunchKernelGGL((gram_direct<0>), dim3(G), dim3(512), 0, 0, f, rows, fshift, G, partials, clk);
        } else if (which == 1) {
            CK(hipFuncSetAttribute((const void*)gram_dma<4, 2, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            hipLaunchKernelGGL((gram_dma<4, 2, 0, false>), dim3(G), dim3(256), 131072, 0, f, rows, fshift, G, partials, clk);
        } else {
            CK(hipFuncSetAttribute((const void*)gram_dma<8, 1, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            hipLaunchKernelGGL((gram_dma<8, 1, 0, false>), dim3(G), dim3(512), 131072, 0, f, rows, fshift, G, partials, clk);
        }
        CK(hipDeviceSynchronize());
        std::vector<double> hf(rows * P), hp((size_t)G * 2 * NT * 256);
        CK(hipMemcpy(hf.data(), f, hf.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hp.data(), partials, hp.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> got(2 * K * K, 0.0), want(2 * K * K, 0.0);
        for (uint32_t g = 0; g < G; ++g) {
            int t = 0;
            for (int I = 0; I < NB; ++I)
                for (int J = I; J < NB; ++J, ++t)
                    for (int v = 0; v < 4; ++v)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int row = 16 * I + (lane >> 4) + 4 * v, col = 16 * J + (lane & 15);
                            got[row * K + col] += hp[((size_t)g * 2 * NT + t) * 256 + v * 64 + lane];
                            got[K * K + row * K + col] += hp[((size_t)g * 2 * NT + NT + t) * 256 + v * 64 + lane];
                        }
        }
        std::vector<double> S(K * rows), T(K * rows);
        for (int u = 0; u < K; ++u)
            for (uint64_t r = 0; r < rows; ++r) {
                uint64_t ic, id;
                host_layout_index(rows, fshift, P, 2 + u, r, ic);
                host_layout_index(rows, fshift, P, 2 + K + u, r, id);
                S[u * rows + r] = hf[ic] + hf[id];
                T[u * rows + r] = hf[ic] - hf[id];
            }
        double worst = 0;
        for (int a = 0; a < K; ++a)
            for (int b = a; b < K; ++b) {
                long double ss = 0, tt = 0;
                for (uint64_t r = 0; r < rows; ++r) {
                    ss += (long double)S[a * rows + r] * S[b * rows + r];
                    tt += (long double)T[a * rows + r] * T[b * rows + r];
                }
                want[a * K + b] = (double)ss;
                want[K * K + a * K + b] = (double)tt;
                // T T^T entries are sums of products of either sign: measure against the S S^T scale
                const double e1 = fabs(got[a * K + b] - (double)ss) / fabs((double)ss);
                const double e2 = fabs(got[K * K + a * K + b] - (double)tt) / fabs((double)ss);
                worst = fmax(worst, fmax(e1, e2));
            }
        printf("correctness of kernel %d (rows = %llu, last row block short): max rel err %.3e %s\n", which, (unsigned long long)rows, worst,
               worst < 1e-12 ? "OK" : "WRONG");
        CK(hipFree(f));
        if (!(worst < 1e-12)) return 2;
    }

    // 2. time at the bench size
    const uint64_t rows = (uint64_t)1 << log2rows;
    CK(hipMalloc(&f, rows * P * 8));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, f, rows * P, bits);
    printf("values with %d random mantissa bits\n", bits);
    CK(hipDeviceSynchronize());
    const double gb = 2.0 * K * rows * 8 / 1e9, gflop = 2.0 * NT * 2048.0 * (rows / 4) / 1e9;
    printf("k = %d, rows = 2^%d, row blocks of 2^%d: %.3f GB of operands, %.2f GFLOP issued\n", K, log2rows, fshift, gb,
           gflop);
    double mhz;
    run_stream<0>(f, rows, fshift, G, partials, gb);
    run_stream<1>(f, rows, fshift, G, partials, gb);
    run_stream<2>(f, rows, fshift, G, partials, gb);
    run_stream<3>(f, rows, fshift, G, partials, gb);
    for (int round = 0; round < 2; ++round) {
        if (argc > 4) {
            run_dma<4, 2, 0, false>(f, rows, fshift, G, partials, clk, 10);
            run_dma<4, 2, 2, false>(f, rows, fshift, G, partials, clk, 10);
            run_dma<8, 1, 0, false>(f, rows, fshift, G, partials, clk, 10);
        }
        run_dma<8, 1, 2, false>(f, rows, fshift, G, partials, clk, 10);
        run_dma<4, 2, 0, true>(f, rows, fshift, G, partials, clk, 10);
        run_dma<8, 1, 0, true>(f, rows, fshift, G, partials, clk, 10);
    }
    for (int round = 0; round < 1; ++round) {
        float a = run<0>(f, rows, fshift, G, partials, clk, 10, &mhz);
        printf("    product candidate: %.2f TB/s, %.1f TFLOP/s\n", gb / a, gflop / a);
        run<1>(f, rows, fshift, G, partials, clk, 10, &mhz);
        run<2>(f, rows, fshift, G, partials, clk, 10, &mhz);
    }
    return 0;
}
