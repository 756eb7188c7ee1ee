// sens_2 / sens_2n sums for 65 .. 128 index rows per half (varsens/saltelli.py:612-619): S S^T and T T^T of
// S = C + D, T = C - D on the fp64 matrix cores, one output, even row count, no NaN mask -- BASELINE configs[4]
// (k = 128, n = 2^24).  Replaces gram_macro_st_kernel for that shape; the tile ownership and the partials are
// the same (vs_second.hip: wave w owns block row w of S S^T and block row 7 - w of T T^T, nine 16x16 tiles).
//
// Why a second kernel (round 3, profiles/r03/gram_k128_n24_before.json): the old one ran four waves per SIMD at
// 128 VGPRs -- nine accumulator tiles (72) plus eight 16-byte loads in flight (32) leave nothing, so the compiler
// spilled ten dwords and reloaded them INSIDE the MFMA section: every tile opened with `scratch_load_dword;
// s_waitcnt vmcnt(0)`, i.e. waited for the HBM loads of the next stage it had issued a moment before, and every
// `ds_read_b128` was followed by `s_waitcnt lgkmcnt(0)`.  Counters: matrix pipe busy 0.57, waves parked on
// s_waitcnt / barrier 46 % of their cycles, 3.1 x the needed vector-memory reads (the reloads).
//
// Here:
//  * ONE workgroup of eight waves per CU = two waves per SIMD, 256 VGPRs each: no spills, and room for every
//    fragment of a sub-tile to be in flight before the first MFMA that needs it;
//  * the body is compiled once per wave index (switch on the wave, all tile / fragment addresses are
//    compile-time facts): straight-line code between barriers, the diagonal tiles reuse the A fragments;
//  * a stage is 32 sample rows (what one fetch brings: 4 index rows x 256 contiguous bytes per wave-wide load,
//    the shape that streams at 6.7 TB/s, tools/load_patterns.hip) in two 16-row sub-tiles of pitch 20;
//    two stages = all 160 KiB of LDS; ONE barrier per 72 MFMAs per wave (was: per 36);
//  * the next stage's LDS writes and the fetch after it sit between the two sub-tiles' MFMAs.
// FIRST (vs_reduce_first_second_order, config 5's default API call): a = fM_1 and b = fM_2 of the stage ride in
// the tile-row padding and meet the two A fragments of every wave (a.s, b.s, a.t, b.t of 16 index rows each) --
// 32 v_fma_f64 per wave and stage beside 72 MFMAs; the record takes a.c = (a.s + a.t)/2, a.d = (a.s - a.t)/2.
// The five scalar sums come from first_head_kernel as for gram_half_st_kernel.
#include "vs_second.h"

namespace {

constexpr int SUB = 2 * VS2_MACRO * VS2_WPITCH;   // doubles per sub-tile: 256 tile rows (128 s, 128 t) x pitch 20
constexpr int STAGE = 2 * SUB;                    // doubles per stage: 32 sample rows
constexpr int ST_ROWS = 2 * VS2_WCHUNK;           // sample rows per stage
constexpr size_t ST_LDS_BYTES = sizeof(double) * 2 * STAGE;   // two stages: 163 840 B

template <int W, int NBL, bool FULLK, bool FIRST>
__device__ __forceinline__ void st128_body(const vs_gram_args& A, double* tiles) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int K = A.K;
    static_assert(!FULLK || NBL == 8, "FULLK means K == 128");
    constexpr int nbl = NBL;                     // live 16-row blocks per half = (K + 15) / 16: a compile-time fact
    const uint64_t rows = A.rows;
    constexpr int TR = 7 - W;                    // block row of T T^T

    // staging: thread t holds sample rows 2p, 2p+1 (p = t & 15) of the stage for c_u and d_u, u = (t >> 4) + 32 j
    const int lrow = tid >> 4, piece = tid & 15;
    const uint64_t jstep = 32 * rows, dstep = (uint64_t)K * rows;
    double* wr = tiles + (piece >> 3) * SUB + lrow * VS2_WPITCH + 2 * (piece & 7);   // + 32 j rows: s; + 128 rows: t
    const double* rd = tiles + (lane & 15) * VS2_WPITCH + 2 * (lane >> 4);           // + 16 blk rows (+ 128: t)

    vs_d4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = (vs_d4){0.0, 0.0, 0.0, 0.0};

    // a workgroup takes two consecutive stages per visit (512 contiguous bytes per index row)
    const uint64_t nst = (rows + ST_ROWS - 1) / ST_ROWS;
    const uint64_t ngroups = (nst + 1) / 2;
    uint64_t mine = 0;
    for (uint64_t g = blockIdx.x; g < ngroups; g += gridDim.x) mine += (2 * g + 2 <= nst) ? 2 : nst - 2 * g;
    vs_d2 raw[8];
    // FIRST: threads 0..15 stage a (plane 0), 16..31 b (plane 1)
    const bool abth = FIRST && tid < 32;
    vs_d2 rawab = (vs_d2){0.0, 0.0};
    double pas = 0.0, pbs = 0.0, pat = 0.0, pbt = 0.0;

    auto fetch = [&](uint64_t st) {
        const uint64_t s0 = st * ST_ROWS;
        // rows is even and a thread's two sample rows start at an even index: both are inside or both outside
        const bool in_rows = s0 + 2 * piece < rows;
        // the eight addresses are rebuilt from the thread's row at every fetch (a handful of integer instructions)
        // instead of living in sixteen registers between fetches
        int lr = lrow;
        asm volatile("" : "+v"(lr));   // not loop-invariant as far as the compiler can tell
        if (FIRST && abth) {
            const double* qab = A.f + (lr >= 1 ? rows : 0) + 2 * piece + s0;   // tid < 32: lr = tid >> 4 is 0 (a) or 1 (b)
            rawab = in_rows ? __builtin_nontemporal_load((const vs_d2*)qab) : (vs_d2){0.0, 0.0};
        }
        const double* q0 = A.f + (uint64_t)(2 + lr) * rows + 2 * piece + s0;   // c_u, u = lr; plane of c_u is 2 + u, of d_u 2 + K + u
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool in = (FULLK || lr + 32 * (j & 3) < K) && in_rows;   // rows past K are zeros in the tile
            const double* q = q0 + (j & 3) * jstep + (j >= 4 ? dstep : 0);
            raw[j] = in ? __builtin_nontemporal_load((const vs_d2*)q) : (vs_d2){0.0, 0.0};
        }
    };
    auto stash = [&](int b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const vs_d2 c = raw[j], d = raw[4 + j];
            *(vs_d2*)(wr + b * STAGE + 32 * j * VS2_WPITCH) = c + d;
            *(vs_d2*)(wr + b * STAGE + (VS2_MACRO + 32 * j) * VS2_WPITCH) = c - d;
        }
        // sample pair p of a sub-tile goes into the padding of that sub-tile's row p: a at [16,17], b at [18,19]
        if (FIRST && abth)
            *(vs_d2*)(tiles + b * STAGE + ((tid & 15) >> 3) * SUB + (tid & 7) * VS2_WPITCH + 16 + 2 * (tid >> 4)) = rawab;
    };
    // A sub-tile's nine tiles go in three groups of three slots; the fragments of a group are read into
    // registers one group ahead of its MFMAs (explicit software pipeline, pinned with sched_barrier: the
    // compiler cannot move LDS reads across the LDS writes of `stash`, so left alone the reads of the second
    // sub-tile queue behind the eight ds_write_b128 of every wave -- measured: 0.2 ms of 2.6 at n = 2^22).
    // Every address is a compile-time offset from `rd + off`.
    struct frags {
        vs_d2 as0, as1, at0, at1;
        vs_d2 b0[3], b1[3];
    };
    auto slot_live = [&](int sl) {
        const bool ss = sl < 8 - W;
        const int blk = ss ? W + sl : sl - 1;
        return FULLK || ((ss ? W : TR) < nbl && blk < nbl);
    };
    auto load_group = [&](int off, int lo, int hi, frags& F) {
        const double* base = rd + off;
        bool need_s = false, need_t = false;   // slots below 8 - W are S S^T tiles (folds at compile time)
#pragma unroll
        for (int sl = lo; sl < hi; ++sl) {
            need_s |= slot_live(sl) && sl < 8 - W;
            need_t |= slot_live(sl) && sl >= 8 - W;
        }
        if (FIRST) {   // the first-order sums take the S fragments with slot 0's group, the T fragments with slot 8's
            need_s |= lo == 0 && W < nbl;
            need_t |= hi == 9 && TR < nbl;
        }
        if (need_s) {
            F.as0 = *(const vs_d2*)(base + 16 * W * VS2_WPITCH);
            F.as1 = *(const vs_d2*)(base + 16 * W * VS2_WPITCH + 8);
        }
        if (need_t) {
            F.at0 = *(const vs_d2*)(base + (VS2_MACRO + 16 * TR) * VS2_WPITCH);
            F.at1 = *(const vs_d2*)(base + (VS2_MACRO + 16 * TR) * VS2_WPITCH + 8);
        }
#pragma unroll
        for (int sl = lo; sl < hi; ++sl) {
            const bool ss = sl < 8 - W;
            const int blk = ss ? W + sl : sl - 1;        // block of the other operand
            if (slot_live(sl) && blk != (ss ? W : TR)) {  // the diagonal tiles take the A fragments twice
                const double* rb = base + ((ss ? 0 : VS2_MACRO) + 16 * blk) * VS2_WPITCH;
                F.b0[sl - lo] = *(const vs_d2*)rb;
                F.b1[sl - lo] = *(const vs_d2*)(rb + 8);
            }
        }
    };
    // FIRST: a, a, b, b of the sample rows 2q, 2q+1 (tile row q) and 8+2q, 8+2q+1 (row 4+q) of the lane group
    // q = lane >> 4, read with the fragments of the group before
    vs_d4 ab0 = (vs_d4){0.0, 0.0, 0.0, 0.0}, ab1 = ab0;
    auto load_ab = [&](int off) {
        const double* rdab = tiles + off + (lane >> 4) * VS2_WPITCH + 16;
        ab0 = *(const vs_d4*)rdab;
        ab1 = *(const vs_d4*)(rdab + 4 * VS2_WPITCH);
    };
    auto mfma_group = [&](int lo, int hi, const frags& F) {
#pragma unroll
        for (int sl = lo; sl < hi; ++sl) {
            const bool ss = sl < 8 - W;                  // S S^T tile (W, W + sl), else T T^T tile (TR, sl - 1)
            const int blk = ss ? W + sl : sl - 1;
            if (slot_live(sl)) {
                const vs_d2 a0 = ss ? F.as0 : F.at0, a1 = ss ? F.as1 : F.at1;
                const bool diag = blk == (ss ? W : TR);
                const vs_d2 f0 = diag ? a0 : F.b0[sl - lo], f1 = diag ? a1 : F.b1[sl - lo];
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], f0[0], acc[sl], 0, 0, 0);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], f0[1], acc[sl], 0, 0, 0);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], f1[0], acc[sl], 0, 0, 0);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], f1[1], acc[sl], 0, 0, 0);
            }
        }
    };
    // FIRST: the sub-tile's a, b against the wave's two A fragments (one group of every sub-tile holds each)
    auto first_sums = [&](int lo, int hi, const frags& F) {
        if (lo == 0 && W < nbl) {        // the group with slot 0 holds the S fragments; block rows past nbl are zeros
            pas = fma(ab0[0], F.as0[0], pas);
            pbs = fma(ab0[2], F.as0[0], pbs);
            pas = fma(ab0[1], F.as0[1], pas);
            pbs = fma(ab0[3], F.as0[1], pbs);
            pas = fma(ab1[0], F.as1[0], pas);
            pbs = fma(ab1[2], F.as1[0], pbs);
            pas = fma(ab1[1], F.as1[1], pas);
            pbs = fma(ab1[3], F.as1[1], pbs);
        }
        if (hi == 9 && TR < nbl) {       // the group with slot 8 holds the T fragments
            pat = fma(ab0[0], F.at0[0], pat);
            pbt = fma(ab0[2], F.at0[0], pbt);
            pat = fma(ab0[1], F.at0[1], pat);
            pbt = fma(ab0[3], F.at0[1], pbt);
            pat = fma(ab1[0], F.at1[0], pat);
            pbt = fma(ab1[2], F.at1[0], pbt);
            pat = fma(ab1[1], F.at1[1], pat);
            pbt = fma(ab1[3], F.at1[1], pbt);
        }
    };
#define ST_PIN() __builtin_amdgcn_sched_barrier(0)

    // gridDim.x as an opaque scalar: left to itself the compiler re-reads it from the dispatch packet inside the loop
    // (s_load + s_waitcnt lgkmcnt(0), which also waits for every LDS write of the stash in front of it)
    uint32_t G = A.slab_wgs;
    asm volatile("" : "+s"(G));
    auto stage_of = [&](uint64_t i) { return 2 * (blockIdx.x + (i / 2) * (uint64_t)G) + (i & 1); };

    if (mine > 0) {
        fetch(stage_of(0));
        stash(0);
        if (mine > 1) fetch(stage_of(1));
    }
    __syncthreads();
    // groups 0..5 of a stage: sub-tile g / 3, slots 3 (g % 3) .. 3 (g % 3) + 2; group g lives in F[g & 1]
    frags F[2];
    if (mine > 0) load_group(0, 0, 3, F[0]);
    for (uint64_t i = 0; i < mine; ++i) {
        const int b = (int)(i & 1);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int lo = 3 * (g % 3), off = b * STAGE + (g / 3) * SUB;
            if (g + 1 < 6) {
                load_group(b * STAGE + ((g + 1) / 3) * SUB, 3 * ((g + 1) % 3), 3 * ((g + 1) % 3) + 3, F[(g + 1) & 1]);
            } else {
                // every LDS read of this stage has been issued (group 5 sits in F[1]); the barrier -- it waits for
                // them -- frees the buffer for the stash of the next iteration and publishes the other one, whose
                // first group is read here: barrier and read latency hide behind the twelve MFMAs of group 5
                __syncthreads();
                if (i + 1 < mine) load_group((b ^ 1) * STAGE, 0, 3, F[0]);
            }
            if (FIRST && g % 3 == 0) load_ab(off);      // used with this group (S fragments) and the third (T)
            ST_PIN();
            mfma_group(lo, lo + 3, F[g & 1]);
            if (FIRST) first_sums(lo, lo + 3, F[g & 1]);
            ST_PIN();
            // the next stage goes into the other buffer (everybody left it at the last barrier) while both waves
            // of the SIMD have MFMAs queued; its registers are then free for the fetch after it, which has the
            // rest of this iteration and a third of the next to land
            if (g == 1 && i + 1 < mine) stash(b ^ 1);
            if (g == 3 && i + 2 < mine) fetch(stage_of(i + 2));
        }
    }
#undef ST_PIN

    if (FIRST) {
        // the four lane groups of a wave hold the same index rows over different samples
        pas += __shfl_xor(pas, 16, 64);
        pas += __shfl_xor(pas, 32, 64);
        pbs += __shfl_xor(pbs, 16, 64);
        pbs += __shfl_xor(pbs, 32, 64);
        pat += __shfl_xor(pat, 16, 64);
        pat += __shfl_xor(pat, 32, 64);
        pbt += __shfl_xor(pbt, 16, 64);
        pbt += __shfl_xor(pbt, 32, 64);
        double* red = tiles;   // the tiles are free: [s rows 0..127, t rows 128..255][2]
        if (lane < 16) {
            red[(16 * W + lane) * 2 + 0] = pas;
            red[(16 * W + lane) * 2 + 1] = pbs;
            red[(VS2_MACRO + 16 * TR + lane) * 2 + 0] = pat;
            red[(VS2_MACRO + 16 * TR + lane) * 2 + 1] = pbt;
        }
        __syncthreads();
        const int k = A.k;
        const size_t plen = VS_PLEN(k);
        double* rec = A.first_partials + (size_t)blockIdx.x * plen;
        for (size_t e = tid; e < plen; e += 512) {
            double v = 0.0;
            if (e >= VS_P_HEAD) {   // the five scalar sums come from first_head_kernel's records
                // record layout: head, ac[k], bd[k], ad[k], bc[k]
                const int slot = (int)((e - VS_P_HEAD) / k), j = (int)((e - VS_P_HEAD) % k);
                const int which = (slot == 0 || slot == 2) ? 0 : 1;
                const double xs = red[j * 2 + which], xt = red[(VS2_MACRO + j) * 2 + which];
                v = (slot == 0 || slot == 3) ? 0.5 * (xs + xt) : 0.5 * (xs - xt);
            }
            rec[e] = v;
        }
    }
    // a tile belongs to one wave: straight to this workgroup's partial [S S^T, T T^T][128][128] (all-zero and
    // mirror tiles stay unwritten; the fold never reads them)
    double* out = A.partials + (size_t)blockIdx.x * 2 * VS2_MTILE_OUT;
#pragma unroll
    for (int sl = 0; sl < 9; ++sl) {
        const bool ss = sl < 8 - W;
        const int blk = ss ? W + sl : sl - 1;
        const int rowblk = ss ? W : TR;
        const bool live = FULLK || (rowblk < nbl && blk < nbl);
        if (live) {
            double* o = out + (ss ? 0 : VS2_MTILE_OUT);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = (lane >> 4) + 4 * v, col = lane & 15;   // f64 C/D layout
                o[(rowblk * 16 + row) * VS2_MACRO + (blk * 16 + col)] = acc[sl][v];
            }
        }
    }
}

template <int NBL, bool FULLK, bool FIRST>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
gram_st128_kernel(vs_gram_args A) {
    extern __shared__ double tiles[];   // [2 stages][2 sub-tiles][256 tile rows][VS2_WPITCH]
    if (FIRST && blockIdx.x == 0 && threadIdx.x == 0) {
        // this kernel opens the call: reset the counters the two folds use (cf. reduce_first_kernel)
        uint32_t* c = const_cast<uint32_t*>(A.counters);
        c[0] = 0;
        c[1] = 0xffffffffu;
        c[2] = 0;
        c[3] = 0;
        c[4] = 0;
    }
    // one body per wave index: the s_barrier count is the same in all of them
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
        case 0: st128_body<0, NBL, FULLK, FIRST>(A, tiles); break;
        case 1: st128_body<1, NBL, FULLK, FIRST>(A, tiles); break;
        case 2: st128_body<2, NBL, FULLK, FIRST>(A, tiles); break;
        case 3: st128_body<3, NBL, FULLK, FIRST>(A, tiles); break;
        case 4: st128_body<4, NBL, FULLK, FIRST>(A, tiles); break;
        case 5: st128_body<5, NBL, FULLK, FIRST>(A, tiles); break;
        case 6: st128_body<6, NBL, FULLK, FIRST>(A, tiles); break;
        default: st128_body<7, NBL, FULLK, FIRST>(A, tiles); break;
    }
}

}  // namespace

// (NBL, FULLK) instances: 5 .. 8 live blocks per half; K == 128 also drops the row-below-K tests
#define VS_ST_INSTANCES(X) X(5, false) X(6, false) X(7, false) X(8, false) X(8, true)

int vs_gram_st128_prepare(vs_context* ctx) {
    (void)ctx;
#define VS_ST_SET(NB, FK)                                                                                      \
    VS_HIP(hipFuncSetAttribute((const void*)gram_st128_kernel<NB, FK, false>,                                  \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)ST_LDS_BYTES));                \
    VS_HIP(hipFuncSetAttribute((const void*)gram_st128_kernel<NB, FK, true>,                                   \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)ST_LDS_BYTES));
    VS_ST_INSTANCES(VS_ST_SET)
#undef VS_ST_SET
    return VS_OK;
}

// A.K in 65..128, A.l == 1, A.rows even, A.f 16-byte aligned (the caller checks); `workgroups` <= CUs
int vs_gram_st128_launch(vs_context* ctx, const vs_gram_args& A, bool with_first, unsigned workgroups) {
    const dim3 grid(workgroups), block(512);
    const int nbl = (A.K + 15) / 16;
    const bool full = A.K == VS2_MACRO;
    VS_REQUIRE(nbl >= 5 && nbl <= 8 && A.l == 1, "gram_st128_kernel takes 65 .. 128 index rows of one output");
    bool launched = false;
#define VS_ST_LAUNCH(NB, FK)                                                                                   \
    if (!launched && nbl == NB && full == FK) {                                                                \
        if (with_first) hipLaunchKernelGGL((gram_st128_kernel<NB, FK, true>), grid, block, ST_LDS_BYTES, ctx->stream, A);  \
        else hipLaunchKernelGGL((gram_st128_kernel<NB, FK, false>), grid, block, ST_LDS_BYTES, ctx->stream, A);           \
        launched = true;                                                                                       \
    }
    VS_ST_INSTANCES(VS_ST_LAUNCH)
#undef VS_ST_LAUNCH
    VS_HIP(hipGetLastError());
    return VS_OK;
}
