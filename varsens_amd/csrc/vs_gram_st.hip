// sens_2 / sens_2n sums for 33 .. 128 index rows per half (varsens/saltelli.py:612-619): S S^T and T T^T of
// S = C + D, T = C - D on the fp64 matrix cores, any number of outputs (index rows = k l), even row count, no NaN
// mask -- BASELINE configs[4] (k = 128, n = 2^24) and every k l in 33 .. 128.  Replaces gram_macro_st_kernel (65 .. 128 rows: NB = 8 blocks of
// 16 per half) and gram_half_st_kernel (33 .. 64: NB = 4) for those shapes; tile ownership and partials are theirs
// (vs_second.hip): a wave with block row R owns row R of S S^T (NB - R tiles) and row NB-1-R of T T^T (R + 1).
//
// Why (round 3, profiles/r03/gram_k128_n24_before.json, gram_k64_n22_before.json): the old kernels ran four waves
// per SIMD at 128 VGPRs.  At nine accumulator tiles (72 VGPRs) plus eight 16-byte loads in flight (32) the
// compiler spilled ten dwords and reloaded them INSIDE the MFMA section -- every tile opened with
// `scratch_load_dword; s_waitcnt vmcnt(0)`, i.e. waited for the HBM loads of the next stage issued a moment
// before -- and in both kernels every `ds_read_b128` was followed by `s_waitcnt lgkmcnt(0)` (tiles behind
// wave-uniform branches).  Counters: matrix pipe busy 0.57 / 0.64, waves parked on s_waitcnt or the barrier
// 46 % / 29 % of their cycles, 3.1 x the needed vector-memory reads at k = 128 (the reloads).
//
// Here:
//  * ONE workgroup of eight waves per CU = two waves per SIMD, 256 VGPRs each: no spills, room for the
//    fragments of the next tiles to be in flight while the current ones multiply;
//  * the body is compiled once per wave index (switch on the wave) and per number of live blocks: every tile
//    and fragment address is a compile-time fact, straight-line code, the diagonal tiles reuse the A fragments;
//  * a stage is 80 KiB of tile: 32 sample rows at NB = 8 (two 16-row sub-tiles of 256 tile rows, every wave
//    works on both), 64 at NB = 4 (four sub-tiles of 128 tile rows; waves 0..3 take the first two, 4..7 the
//    others); two stages = all 160 KiB of LDS; one barrier per 72 (NB = 8) or 40 (NB = 4) MFMAs per wave;
//  * a fetch is eight 16-byte loads per thread and stage, addresses rebuilt per fetch; a wave-wide load covers
//    4 index rows x 256 contiguous bytes at NB = 8 and 2 x 512 at NB = 4 (tools/load_patterns.hip: 4 x 256 streams
//    at 6.7 TB/s, 1 x 1024 at 7.0, 8 x 128 at 4.9);
//  * explicit software pipeline, pinned with sched_barrier: the fragments of a group of tiles are read one group
//    ahead of its MFMAs; the next stage's LDS writes and the fetch after it go between groups, where both waves
//    of a SIMD have MFMAs queued (the compiler cannot move LDS reads across those writes; left alone the reads
//    of the second sub-tile queue behind the eight ds_write_b128 of every wave); the barrier sits AHEAD of the
//    last group's MFMAs, so its wait and the first reads of the next stage hide behind them.
// FIRST (vs_reduce_first_second_order): a = fM_1 and b = fM_2 of the stage ride in the tile-row padding and meet
// the two A fragments of every wave (a.s, b.s, a.t, b.t of 16 index rows each) -- 16 v_fma_f64 per sub-tile
// beside 36 / 20 MFMAs; the record takes a.c = (a.s + a.t)/2, a.d = (a.s - a.t)/2.  The five scalar sums come
// from first_head_kernel.
// Measured (n = 2^24, k = 128): 9.8 ms (old: 13.95), both orders in one call 10.5 (19.2); matrix pipe busy 0.80 at
// 2.40 GHz.
// NB = 4 (k = 64, n = 2^22) stays at 0.97-1.0 ms (old kernel: 1.0; k = 40 .. 56: 0.63 .. 0.92, old 0.73 .. 0.96), at
// 2.11 GHz with the pipe busy 0.63 -- pipe floor 0.62 ms at that clock, HBM 0.61 ms at 7 TB/s.  Ablations
// (profiles/r03/gram_k64_ablation.txt): without the next stage's loads 0.76 ms (2.28 GHz), without loads, LDS writes
// and barrier 0.66 = the MFMA + fragment-read loop alone.  The 0.24 ms of the loads are NOT: the address
// arithmetic (wave-uniform offsets, one vector add per load: same time), the wait for the data (never stashing
// what was loaded: same time), prefetch distance (two stages in flight: same), code size (waves r and r + 4 share a
// body: same), one burst per stage (spread over the stage: same), the shape of a wave-wide load (2 x 512 B instead
// of 4 x 256 B: same), the workgroup-to-sample-range map (runs of 1 .. 256 visits per workgroup: same) -- or who
// issues them: with four extra waves issuing the loads and the compute waves touching neither vector memory nor
// LDS writes, the time is the same 1.0 ms.  The HBM stream itself takes the matrix pipe's throughput (clock 2.11
// instead of 2.28 GHz, and a lower busy fraction at that clock).
// Round 5: with one output the NB = 4 shapes run on gram_dma_kernel (vs_gram_dma.hip: no shared tile, every wave on its
// own LDS-DMA image, 0.80-0.82 ms at k = 64), several outputs included; the NB = 4 instances here are what VS_GRAM_DMA=0
// selects (tests compare the two).
#include "vs_second.h"

namespace {

constexpr int STAGE = 2 * 2 * VS2_MACRO * VS2_WPITCH;         // doubles per stage: 512 tile rows of pitch 20 = 80 KiB
constexpr size_t ST_LDS_BYTES = sizeof(double) * 2 * STAGE;   // two stages: 163 840 B

// NB: 16-row blocks per half the tile provides (8: 128 rows, 4: 64 rows); NBL <= NB of them are live (K rows);
// FULLK: K == 16 NB (no row-below-K tests)
// R: the wave's block row of S S^T (compile-time); grp: its wave group (NB = 4: which two sub-tiles of the stage it works
// on -- waves r and r + 4 share one body: half the code, which has to stay inside the instruction cache)
// ONE: one output (index row u is plane 2 + u for c_u, 2 + K + u for d_u: wave-uniform strides between a thread's loads);
// otherwise row m = j l + o lives in plane o (2k + 2) + 2 + j (+ k for d): the eight plane offsets of a thread are
// worked out once
template <int R, int NB, int NBL, bool FULLK, bool FIRST, bool ONE>
__device__ __forceinline__ void st_body(const vs_gram_args& A, double* tiles, const int grp) {
    static_assert(ONE || !FIRST, "the first-order sums ride along for one output only");
    static_assert(NB == 8 || NB == 4, "128 or 64 index rows per half");
    static_assert(!FULLK || NBL == NB, "FULLK means K == 16 NB");
    constexpr int NROWS = 16 * NB;                   // index rows per half in the tile
    constexpr int SUB = 2 * NROWS * VS2_WPITCH;      // doubles per sub-tile (16 sample rows): s rows, then t rows
    constexpr int NSUB = STAGE / SUB;                // sub-tiles per stage: 2 or 4
    constexpr int ST_ROWS = NSUB * VS2_WCHUNK;       // sample rows per stage: 32 or 64
    static_assert(R < NB, "block row");
    constexpr int TR = NB - 1 - R;                   // block row of T T^T
    constexpr int NSLOT = NB + 1;                    // accumulator tiles per wave
    // tiles per pipeline group: three.  (Where the nine-tile kernel also carries the first-order sums its 256 registers
    // ran out in round 3 and a thread constant went through scratch once per stage; the constants are now rebuilt where
    // they are used -- stash(), the epilogue -- and nothing is spilled at three tiles per group or at two;
    // VS_ST_GS_FIRST8 = 2 is the measurement knob for the latter.)
#ifndef VS_ST_GS_FIRST8
#define VS_ST_GS_FIRST8 3
#endif
    constexpr int GS = (FIRST && NB == 8) ? VS_ST_GS_FIRST8 : 3;
    constexpr int NGS = (NSLOT + GS - 1) / GS;       // groups per sub-tile: 3 (slots 0-2, 3-5, 6-8) or 2 (0-2, 3-4)
    constexpr int NG = 2 * NGS;                      // groups per stage and wave
    // One fetch in flight.  Two (a second register set, the stage loop unrolled by two so that the set is a
    // compile-time fact) were measured at NB = 4 and bought nothing: see the note on NB = 4 at the top.
    constexpr int DEPTH = 1;
    constexpr int nbl = NBL;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int K = A.K;
    const uint64_t rows = A.rows;

    // staging: a stage is ST_ROWS sample rows = PIECES pairs per index row; thread t holds pair p = t % PIECES of the
    // index rows u = t / PIECES + RSTEP j, j < 4, for c_u and for d_u: eight 16-byte loads.  A wave-wide load is
    // 4 index rows x 256 B at NB = 8 (32-sample stages) and 2 x 512 B at NB = 4 (64-sample stages)
    constexpr int PIECES = ST_ROWS / 2, RSTEP = 512 / PIECES;
    static_assert(4 * RSTEP == NROWS, "four row blocks per half");
    const int lrow = tid / PIECES, piece = tid % PIECES;
    uint32_t poff[ONE ? 1 : 8];   // several outputs: the plane of load j (dead rows: row 0, never loaded)
    if (!ONE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int u = lrow + RSTEP * (j & 3);
            const int mm = u < K ? u : 0;
            const int jj = mm / A.l, o = mm - jj * A.l;
            poff[ONE ? 0 : j] = (uint32_t)(o * (2 * A.k + 2) + (j >= 4 ? 2 + A.k : 2) + jj);
        }
    }
    // LDS write address of the stash: tiles + (piece >> 3) * SUB + lrow * VS2_WPITCH + 2 * (piece & 7) (+ RSTEP j rows:
    // s; + NROWS rows: t) -- rebuilt inside stash() (see there)
    const double* rd = tiles + 2 * grp * SUB + (lane & 15) * VS2_WPITCH + 2 * (lane >> 4);   // + 16 blk rows (+ NROWS: t)

    vs_d4 acc[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) acc[i] = (vs_d4){0.0, 0.0, 0.0, 0.0};

    // a workgroup takes 64 consecutive sample rows per visit (512 contiguous bytes per index row)
    constexpr int VISIT = 64 / ST_ROWS;              // stages per visit: 2 or 1
    const uint64_t nst = (rows + ST_ROWS - 1) / ST_ROWS;
    const uint64_t nvisits = (nst + VISIT - 1) / VISIT;
    // gridDim.x as an opaque scalar: left to itself the compiler re-reads it from the dispatch packet inside the loop
    // (s_load + s_waitcnt lgkmcnt(0), which also waits for every LDS write of the stash in front of it)
    uint32_t G = A.slab_wgs;
    asm volatile("" : "+s"(G));
    uint64_t mine = 0;
    for (uint64_t v = blockIdx.x; v < nvisits; v += G) mine += (VISIT * v + VISIT <= nst) ? VISIT : nst - VISIT * v;
#if defined(VS_ST_REPEAT) && VS_ST_REPEAT > 1
    // diagnostic build (tools/gram_cache_vs_hbm.sh): every workgroup walks its stages VS_ST_REPEAT times, so that a
    // design small enough for the 256 MB Infinity Cache gives a launch as long as a large one (sums x VS_ST_REPEAT:
    // timing only)
    const uint64_t mine_once = mine;
    mine *= VS_ST_REPEAT;
    auto stage_of = [&](uint64_t i) {
        i %= mine_once;
        return VISIT * (blockIdx.x + (i / VISIT) * (uint64_t)G) + (i % VISIT);
    };
#else
    auto stage_of = [&](uint64_t i) { return VISIT * (blockIdx.x + (i / VISIT) * (uint64_t)G) + (i % VISIT); };
#endif

    vs_d2 raw[DEPTH][8];
    // FIRST: a (plane 0) and b (plane 1) of the stage: threads 0 .. PIECES-1 and PIECES .. 2 PIECES-1 (lrow 0 and 1)
    const bool abth = FIRST && tid < 2 * PIECES;
    vs_d2 rawab[DEPTH];
#pragma unroll
    for (int z = 0; z < DEPTH; ++z) rawab[z] = (vs_d2){0.0, 0.0};
    double pas = 0.0, pbs = 0.0, pat = 0.0, pbt = 0.0;

    // load j of a thread: row block j & 3, d for j >= 4
    auto fetch = [&](uint64_t st, int set) {
        const uint64_t s0 = st * ST_ROWS;
        // the eight addresses are rebuilt from the thread's row at every fetch (a handful of integer instructions)
        // instead of living in sixteen registers between fetches
        int lr = lrow;
        asm volatile("" : "+v"(lr));   // not loop-invariant as far as the compiler can tell
        // rows is even and a thread's two sample rows start at an even index: both are inside or both outside
        const bool in_rows = s0 + 2 * piece < rows;
        // row-blocked f-values (vs_second.h): the stage's 32 / 64 sample rows lie in one row block; fb = plane 0 at the
        // stage's first row, fp = doubles between the planes of these rows
        VS2_FTILE(s0);
        const uint64_t jstep = (uint64_t)RSTEP * fp, dstep = (uint64_t)K * fp;   // one output: planes u -> u + RSTEP, c_u -> d_u
        if (FIRST && abth) {
            const double* qab = fb + (lr ? fp : 0) + 2 * piece;
            rawab[set] = in_rows ? __builtin_nontemporal_load((const vs_d2*)qab) : (vs_d2){0.0, 0.0};
        }
        const double* q0 = fb + (uint64_t)(2 + lr) * fp + 2 * piece;   // c_u, u = lr; plane of c_u is 2 + u, of d_u 2 + K + u
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int jb = j & 3;
            const bool in = (FULLK || lr + RSTEP * jb < K) && in_rows;   // rows past K are zeros in the tile
            const double* q = ONE ? q0 + jb * jstep + (j >= 4 ? dstep : 0) : fb + (uint64_t)poff[ONE ? 0 : j] * fp + 2 * piece;
            raw[set][j] = in ? __builtin_nontemporal_load((const vs_d2*)q) : (vs_d2){0.0, 0.0};
        }
    };
    auto stash = [&](int b, int set) {
        // The write address is rebuilt from the thread index at every stash (five integer instructions) instead of
        // living in a register between stashes: at nine accumulator tiles plus the first-order sums the 256 registers
        // ran out, the compiler kept this thread constant in scratch and reloaded it once per stage behind
        // `s_waitcnt vmcnt(0)` (round 3's code object: 11 spilled dwords, 20 B of scratch).
        int t = tid;
        asm volatile("" : "+v"(t));   // not loop-invariant as far as the compiler can tell
        const int lrow_w = t / PIECES, piece_w = t % PIECES;
        double* wr = tiles + (piece_w >> 3) * SUB + lrow_w * VS2_WPITCH + 2 * (piece_w & 7);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const vs_d2 c = raw[set][j], d = raw[set][4 + j];   // the c and d loads of the same rows and samples
            *(vs_d2*)(wr + b * STAGE + RSTEP * j * VS2_WPITCH) = c + d;
            *(vs_d2*)(wr + b * STAGE + (NROWS + RSTEP * j) * VS2_WPITCH) = c - d;
        }
        // sample pair p of a sub-tile goes into the padding of that sub-tile's row p: a at [16,17], b at [18,19]
        if (FIRST && t < 2 * PIECES)
            *(vs_d2*)(tiles + b * STAGE + (piece_w >> 3) * SUB + (piece_w & 7) * VS2_WPITCH + 16 + 2 * lrow_w) = rawab[set];
    };

    // fragments of one group of GS tiles; every address is a compile-time offset from `rd + off`
    struct frags {
        vs_d2 as0, as1, at0, at1;
        vs_d2 b0[GS], b1[GS];
    };
    auto slot_live = [&](int sl) {
        const bool ss = sl < NB - R;                 // S S^T tile (R, R + sl), else T T^T tile (TR, sl - 1)
        const int blk = ss ? R + sl : sl - 1;
        return sl < NSLOT && (FULLK || ((ss ? R : TR) < nbl && blk < nbl));
    };
    auto load_group = [&](int off, int lo, frags& F) {
        const double* base = rd + off;
        bool need_s = false, need_t = false;   // folds at compile time
#pragma unroll
        for (int sl = lo; sl < lo + GS; ++sl) {
            need_s |= slot_live(sl) && sl < NB - R;
            need_t |= slot_live(sl) && sl >= NB - R;
        }
        if (FIRST) {   // the first-order sums take the S fragments with slot 0's group, the T fragments with the last slot's
            need_s |= lo == 0 && R < nbl;
            need_t |= lo + GS >= NSLOT && TR < nbl;
        }
        if (need_s) {
            F.as0 = *(const vs_d2*)(base + 16 * R * VS2_WPITCH);
            F.as1 = *(const vs_d2*)(base + 16 * R * VS2_WPITCH + 8);
        }
        if (need_t) {
            F.at0 = *(const vs_d2*)(base + (NROWS + 16 * TR) * VS2_WPITCH);
            F.at1 = *(const vs_d2*)(base + (NROWS + 16 * TR) * VS2_WPITCH + 8);
        }
#pragma unroll
        for (int sl = lo; sl < lo + GS; ++sl) {
            const bool ss = sl < NB - R;
            const int blk = ss ? R + sl : sl - 1;        // block of the other operand
            if (slot_live(sl) && blk != (ss ? R : TR)) { // the diagonal tiles take the A fragments twice
                const double* rb = base + ((ss ? 0 : NROWS) + 16 * blk) * VS2_WPITCH;
                F.b0[sl - lo] = *(const vs_d2*)rb;
                F.b1[sl - lo] = *(const vs_d2*)(rb + 8);
            }
        }
    };
    // FIRST: a, a, b, b of the sample rows 2q, 2q+1 (tile row q) and 8+2q, 8+2q+1 (row 4+q) of the lane group
    // q = lane >> 4, read with the first group of a sub-tile
    vs_d4 ab0 = (vs_d4){0.0, 0.0, 0.0, 0.0}, ab1 = ab0;
    auto load_ab = [&](int off) {
        const double* rdab = tiles + 2 * grp * SUB + off + (lane >> 4) * VS2_WPITCH + 16;
        ab0 = *(const vs_d4*)rdab;
        ab1 = *(const vs_d4*)(rdab + 4 * VS2_WPITCH);
    };
    auto mfma_group = [&](int lo, const frags& F) {
#pragma unroll
        for (int sl = lo; sl < lo + GS; ++sl) {
            const bool ss = sl < NB - R;
            const int blk = ss ? R + sl : sl - 1;
            if (slot_live(sl)) {
                const vs_d2 a0 = ss ? F.as0 : F.at0, a1 = ss ? F.as1 : F.at1;
                const bool diag = blk == (ss ? R : TR);
                const vs_d2 f0 = diag ? a0 : F.b0[sl - lo], f1 = diag ? a1 : F.b1[sl - lo];
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], f0[0], acc[sl], 0, 0, 0);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], f0[1], acc[sl], 0, 0, 0);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], f1[0], acc[sl], 0, 0, 0);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], f1[1], acc[sl], 0, 0, 0);
            }
        }
    };
    auto first_sums = [&](int lo, const frags& F) {
        if (lo == 0 && R < nbl) {                // the group with slot 0 holds the S fragments; block rows past nbl are zeros
            pas = fma(ab0[0], F.as0[0], pas);
            pbs = fma(ab0[2], F.as0[0], pbs);
            pas = fma(ab0[1], F.as0[1], pas);
            pbs = fma(ab0[3], F.as0[1], pbs);
            pas = fma(ab1[0], F.as1[0], pas);
            pbs = fma(ab1[2], F.as1[0], pbs);
            pas = fma(ab1[1], F.as1[1], pas);
            pbs = fma(ab1[3], F.as1[1], pbs);
        }
        if (lo + GS >= NSLOT && TR < nbl) {      // the group with the last slot holds the T fragments
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

#ifdef VS_ST_GLDS_TIMING
    // TIMING EXPERIMENT (profiles/r05/gram_k64_glds_bound.txt; WRONG RESULTS): the stage's operands go
    // straight from global memory into LDS (global_load_lds_dwordx4: no vector registers, no ds_write) -- 64 wave-wide
    // pieces of 1 KiB, lane-linear, one after the other in the stage's buffer; nothing forms s and t, the fragment reads
    // and MFMAs are the product's.  What a kernel built around LDS-DMA could gain at most on the operand path.
    auto glds_stage = [&](uint64_t st, int b) {
        const uint64_t s0 = st * ST_ROWS;
        int lr = lrow;
        asm volatile("" : "+v"(lr));
        VS2_FTILE(s0);
        const uint64_t jstep = (uint64_t)RSTEP * fp, dstep = (uint64_t)K * fp;
        const double* q0 = fb + (uint64_t)(2 + lr) * fp + 2 * piece;
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* glb_ptr;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int jb = j & 3;
            // rows past K and past the shard's last row read row 0 of the stage instead (always inside the buffer)
            const bool in = (FULLK || lr + RSTEP * jb < K) && s0 + 2 * piece < rows;
            const double* q = in ? q0 + jb * jstep + (j >= 4 ? dstep : 0) : fb;
            __builtin_amdgcn_global_load_lds((glb_ptr)q, (lds_ptr)(tiles + b * STAGE + (j * 8 + wv) * 128), 16, 0, 0);
        }
    };
    if (mine > 0) glds_stage(stage_of(0), 0);
    __syncthreads();
#else
    // stage s travels in register set s % DEPTH
    if (mine > 0) {
        fetch(stage_of(0), 0);
        stash(0, 0);
        if (mine > 1) fetch(stage_of(1), 0);
    }
    __syncthreads();
#endif
    // groups 0 .. NG-1 of a stage: sub-tile g / NGS of the wave's two, slots GS (g % NGS) ..; group g lives in F[g & 1]
    frags F[2];
    if (mine > 0) load_group(0, 0, F[0]);
    auto stage_iter = [&](uint64_t i, const int b) {   // b: LDS buffer of the stage
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int lo = GS * (g % NGS), off = b * STAGE + (g / NGS) * SUB;
            if (g + 1 < NG) {
                load_group(b * STAGE + ((g + 1) / NGS) * SUB, GS * ((g + 1) % NGS), F[(g + 1) & 1]);
            } else {
                // every LDS read of this stage has been issued (the last group sits in F[(NG-1) & 1]); the barrier --
                // it waits for them -- frees the buffer for the stash of the next iteration and publishes the other
                // one, whose first group is read here: barrier and read latency hide behind the last group's MFMAs
                __syncthreads();
                if (i + 1 < mine) load_group((b ^ 1) * STAGE, 0, F[NG & 1]);
            }
            if (FIRST && g % NGS == 0) load_ab(off);      // used with this group (S fragments) and the sub-tile's last (T)
            ST_PIN();
            mfma_group(lo, F[g & 1]);
            if (FIRST) first_sums(lo, F[g & 1]);
            ST_PIN();
            // the next stage goes into the other buffer (everybody left it at the last barrier) while both waves
            // of the SIMD have MFMAs queued; its registers are then free for the fetch after it, which has the
            // rest of this iteration to land (positions measured: stash after group 0, fetch after group 1)
#ifdef VS_ST_GLDS_TIMING
            if (g == 0 && i + 1 < mine) glds_stage(stage_of(i + 1), b ^ 1);
#else
            if (g == 0 && i + 1 < mine) stash(b ^ 1, 0);
            if (g == 1 && i + 2 < mine) fetch(stage_of(i + 2), 0);
#endif
        }
    };
    for (uint64_t i = 0; i < mine; ++i) stage_iter(i, (int)(i & 1));
#undef ST_PIN

    // what the epilogue needs of the thread index is rebuilt here, so that nothing of it is carried (or spilled)
    // across the stage loop
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63;
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
        double* red = tiles + grp * 4 * NROWS;   // the tiles are free: [wave group][s rows, then t rows][2]
        if (lane_e < 16) {
            red[(16 * R + lane_e) * 2 + 0] = pas;
            red[(16 * R + lane_e) * 2 + 1] = pbs;
            red[(NROWS + 16 * TR + lane_e) * 2 + 0] = pat;
            red[(NROWS + 16 * TR + lane_e) * 2 + 1] = pbt;
        }
        __syncthreads();
        const int k = A.k;
        const size_t plen = VS_PLEN(k);
        constexpr int NGRP = NB == 8 ? 1 : 2, TPG = 512 / NGRP;   // one record per wave group
        double* rec = A.first_partials + ((size_t)blockIdx.x * NGRP + grp) * plen;
        for (size_t e = tid_e % TPG; e < plen; e += TPG) {
            double v = 0.0;
            if (e >= VS_P_HEAD) {   // the five scalar sums come from first_head_kernel's records
                // record layout: head, ac[k], bd[k], ad[k], bc[k]
                const int slot = (int)((e - VS_P_HEAD) / k), j = (int)((e - VS_P_HEAD) % k);
                const int which = (slot == 0 || slot == 2) ? 0 : 1;
                const double xs = red[j * 2 + which], xt = red[(NROWS + j) * 2 + which];
                v = (slot == 0 || slot == 3) ? 0.5 * (xs + xt) : 0.5 * (xs - xt);
            }
            rec[e] = v;
        }
    }
    // a tile belongs to one wave: straight to the partial of this workgroup (NB = 8) or wave group (NB = 4):
    // [S S^T, T T^T][NROWS][NROWS] (all-zero and mirror tiles stay unwritten; the fold never reads them)
    constexpr int NGRP_OUT = NB == 8 ? 1 : 2;
    double* out = A.partials + ((size_t)blockIdx.x * NGRP_OUT + grp) * 2 * NROWS * NROWS;
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
        const bool ss = sl < NB - R;
        const int blk = ss ? R + sl : sl - 1;
        const int rowblk = ss ? R : TR;
        if (slot_live(sl)) {
            double* o = out + (ss ? 0 : NROWS * NROWS);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = (lane_e >> 4) + 4 * v, col = lane_e & 15;   // f64 C/D layout
                o[(rowblk * 16 + row) * NROWS + (blk * 16 + col)] = acc[sl][v];
            }
        }
    }
}

template <int NB, int NBL, bool FULLK, bool FIRST, bool ONE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
gram_st_kernel(vs_gram_args A) {
    extern __shared__ double tiles[];   // [2 stages][sub-tiles][tile rows][VS2_WPITCH]
    if (FIRST && blockIdx.x == 0 && threadIdx.x == 0) {
        // this kernel opens the call: reset the counters the two folds use (cf. reduce_first_kernel)
        uint32_t* c = const_cast<uint32_t*>(A.counters);
        c[0] = 0;
        c[1] = 0xffffffffu;
        c[2] = 0;
        c[3] = 0;
        c[4] = 0;
    }
    // one body per block row: the s_barrier count is the same in all of them
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (NB == 8) {
        switch (wave) {
            case 0: st_body<0, NB, NBL, FULLK, FIRST, ONE>(A, tiles, 0); break;
            case 1: st_body<1, NB, NBL, FULLK, FIRST, ONE>(A, tiles, 0); break;
            case 2: st_body<2, NB, NBL, FULLK, FIRST, ONE>(A, tiles, 0); break;
            case 3: st_body<3, NB, NBL, FULLK, FIRST, ONE>(A, tiles, 0); break;
            case 4: st_body<4, NB, NBL, FULLK, FIRST, ONE>(A, tiles, 0); break;
            case 5: st_body<5, NB, NBL, FULLK, FIRST, ONE>(A, tiles, 0); break;
            case 6: st_body<6, NB, NBL, FULLK, FIRST, ONE>(A, tiles, 0); break;
            default: st_body<7, NB, NBL, FULLK, FIRST, ONE>(A, tiles, 0); break;
        }
    } else {
        switch (wave & 3) {
            case 0: st_body<0, NB, NBL, FULLK, FIRST, ONE>(A, tiles, wave >> 2); break;
            case 1: st_body<1, NB, NBL, FULLK, FIRST, ONE>(A, tiles, wave >> 2); break;
            case 2: st_body<2, NB, NBL, FULLK, FIRST, ONE>(A, tiles, wave >> 2); break;
            default: st_body<3, NB, NBL, FULLK, FIRST, ONE>(A, tiles, wave >> 2); break;
        }
    }
}

}  // namespace

// (NB, NBL, FULLK) instances: 3 .. 4 live blocks of the 64-row tile, 5 .. 8 of the 128-row tile; K == 16 NB also
// drops the row-below-K tests
#define VS_ST_INSTANCES(X) \
    X(4, 3, false) X(4, 4, false) X(4, 4, true) X(8, 5, false) X(8, 6, false) X(8, 7, false) X(8, 8, false) X(8, 8, true)

int vs_gram_st_prepare(vs_context* ctx) {
    (void)ctx;
#define VS_ST_SET(NB, NL, FK)                                                                                  \
    VS_HIP(hipFuncSetAttribute((const void*)gram_st_kernel<NB, NL, FK, false, true>,                           \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)ST_LDS_BYTES));                \
    VS_HIP(hipFuncSetAttribute((const void*)gram_st_kernel<NB, NL, FK, true, true>,                            \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)ST_LDS_BYTES));                \
    VS_HIP(hipFuncSetAttribute((const void*)gram_st_kernel<NB, NL, FK, false, false>,                          \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)ST_LDS_BYTES));
    VS_ST_INSTANCES(VS_ST_SET)
#undef VS_ST_SET
    return VS_OK;
}

// A.K = k l in 33..128, A.rows even, A.f 16-byte aligned (the caller checks); `workgroups` <= CUs; with_first: one output.
// Partials: K > 64: one per workgroup, [S S^T, T T^T][128][128]; else two (one per wave group), [..][64][64].
int vs_gram_st_launch(vs_context* ctx, const vs_gram_args& A, bool with_first, unsigned workgroups) {
    const dim3 grid(workgroups), block(512);
    const int nb = A.K > 64 ? 8 : 4;
    const int nbl = (A.K + 15) / 16;
    const bool full = A.K == 16 * nb;
    const bool one = A.l == 1;
    VS_REQUIRE(A.K > 32 && A.K <= 128 && A.K == A.k * A.l, "gram_st_kernel takes 33 .. 128 index rows per half");
    VS_REQUIRE(one || !with_first, "the first-order sums ride along for one output only");
    VS_REQUIRE(A.slab_wgs == workgroups, "slab_wgs must be the grid size");
    bool launched = false;
#define VS_ST_LAUNCH(NB, NL, FK)                                                                               \
    if (!launched && nb == NB && nbl == NL && full == FK) {                                                    \
        if (with_first) hipLaunchKernelGGL((gram_st_kernel<NB, NL, FK, true, true>), grid, block, ST_LDS_BYTES, ctx->stream, A);  \
        else if (one) hipLaunchKernelGGL((gram_st_kernel<NB, NL, FK, false, true>), grid, block, ST_LDS_BYTES, ctx->stream, A);   \
        else hipLaunchKernelGGL((gram_st_kernel<NB, NL, FK, false, false>), grid, block, ST_LDS_BYTES, ctx->stream, A);          \
        launched = true;                                                                                       \
    }
    VS_ST_INSTANCES(VS_ST_LAUNCH)
#undef VS_ST_LAUNCH
    if (!launched) return vs_fail(VS_ERR_STATE, "no gram_st_kernel instance for K=%d", A.K);
    VS_HIP(hipGetLastError());
    return VS_OK;
}
