// Second-order sums of Varsens.compute_varsens (varsens/saltelli.py:612-622).
//
// With C = fN_j and D = fN_nj viewed as (K, rows) matrices, K = k*l, row index j*l + o, the reference
// forms D C^T, C D^T, D D^T and C C^T (four numpy.tensordot calls contracting the row axis) and then
// uses only D C^T + C D^T (:612-613) and D D^T + C C^T (:618-619).  With S = C + D and T = C - D:
//     D C^T + C D^T = (S S^T - T T^T) / 2,      D D^T + C C^T = (S S^T + T T^T) / 2,
// two symmetric K x K products instead of the upper triangle of the 2K x 2K Gram matrix of [C; D]: half the
// matrix work.  This is the one GEMM-shaped piece of the path, so it runs on the fp64 matrix cores:
// v_mfma_f64_16x16x4_f64, D(16x16) += A(16x4) B(4x16), where A and B are both "16 index rows x 4 sample
// rows" fragments (one double per lane).  s and t form in registers on the way into LDS.  The interface
// returns SS = S S^T and TT = T T^T (include/varsens_hip.h); vs_finish_second_order takes half-sum and
// half-difference.
//
// Kernels by the number of index rows K per half:
//   K <= 32: one tile of 2K rows (s rows, then t rows), wave-private kernels -- every wave keeps the accumulator
//     tiles, skips the S x T ones and walks its own sample rows through a private LDS tile, no workgroup barrier
//     in the loop: gram_diag_wave_kernel (16-sample-row chunks; any shape) and gram_diag_wide_kernel
//     (32-sample-row units; one output, even row count, K > 16);
//   32 < K <= 128, one output, even row count, no mask: gram_st_kernel (vs_gram_st.hip; carries the first-order dot
//     products); the other shapes of that range: gram_macro_st_kernel -- a workgroup of eight waves, one 256-row tile
//     (128 s rows, 128 t rows), 72 tiles of S S^T and T T^T, nine per wave -- and, up to 64 rows with an even row
//     count, gram_half_st_kernel -- the tile holds two chunks, five tiles per wave;
//   K > 128: gram_macro_kernel over [C; D] -- 128-row macro blocks, a workgroup per pair (MI <= MJ) of them;
//     gram_convert_kernel turns its D C^T, D D^T, C C^T into SS, TT.
// One dense partial per workgroup goes to HBM; gram_fold_kernel sums the partials (compensated) and scatters.
// Roofline at k=32, rows=2^22: HBM 16*K*rows bytes = 2.15 GB => 0.31 ms at 7 TB/s; matrix pipe: 6 tiles x
// rows/4 MFMAs of 64 clocks per SIMD => 0.19 ms at 2.05 GHz (the [C; D] form: 10 tiles, 0.32 ms -- it was
// pipe-bound at 0.44 ms).  The pipe sustains 77.6 TFLOP/s at 2.4 GHz with VGPR accumulators
// (tools/mfma_f64_peak.hip; the "47-49 TFLOP/s" quoted in earlier rounds was the AGPR form); fp64 MFMA and
// v_fma_f64 share it (tools/mfma_valu_overlap.hip).  What HBM delivers depends on the shape of the wave-wide
// loads (tools/load_patterns.hip): 8 x 128-byte runs per instruction cap it at 4.9 TB/s, 4 x 256-byte runs
// reach 6.7 -- hence the 32-sample-row staging.  Reported separately from the first-order pass.
#include "vs_second.h"

__device__ __forceinline__ void pair_from_index(int p, int ntile, int& I, int& J) {
    // pairs 0..ntile-1 are the diagonal (I,I); then I < J row by row
    if (p < ntile) {
        I = J = p;
        return;
    }
    p -= ntile;
    I = 0;
    int rowlen = ntile - 1;
    while (p >= rowlen) {
        p -= rowlen;
        ++I;
        --rowlen;
    }
    J = I + 1 + p;
}

// plane of X[m] in the f-values: index row m < K is C (fN_j), m >= K is D (fN_nj); plane = output (2k+2) + block
__device__ __forceinline__ uint32_t x_plane(const vs_gram_args& A, int m) {
    const int K = A.K, l = A.l;
    const int mm = m < K ? m : m - K;
    const int j = mm / l, o = mm - j * l;
    const uint32_t block = (uint32_t)(m < K ? 2 : 2 + A.k) + (uint32_t)j;
    return (uint32_t)o * (2u * (uint32_t)A.k + 2u) + block;
}

// First-order record of a wave-private workgroup (FIRST): the 16 (wave, lane group) pairs each hold a share of
// every sum -- pa/pb per 16-row block and tile row lane&15 (a.s_u, b.s_u, a.t_u, b.t_u), the five scalar sums in
// the lanes with lane&15 == 0.
template <int NB>
__device__ __forceinline__ void gram_first_record(const vs_gram_args& A, double* tiles, const double (&pa)[NB],
                                                  const double (&pb)[NB], double t_sab, double t_sa, double t_sb,
                                                  double t_s1, double t_s2) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fq = lane >> 4;
    const int k = A.k;
    const int g = wave * 4 + fq, i = lane & 15;
    double* r2 = tiles;                          // [16][NB*16][2]
    double* r3 = tiles + 16 * (NB * 16) * 2;     // [16][5]
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        r2[((size_t)g * (NB * 16) + blk * 16 + i) * 2 + 0] = pa[blk];
        r2[((size_t)g * (NB * 16) + blk * 16 + i) * 2 + 1] = pb[blk];
    }
    if (i == 0) {
        r3[g * 5 + VS_P_SAB] = t_sab;
        r3[g * 5 + VS_P_SA] = t_sa;
        r3[g * 5 + VS_P_SB] = t_sb;
        r3[g * 5 + VS_P_S1] = t_s1;
        r3[g * 5 + VS_P_S2] = t_s2;
    }
    __syncthreads();
    // the tile rows are s_u = c_u + d_u (row u) and t_u = c_u - d_u (row K + u): a.c = (a.s + a.t) / 2, a.d = (a.s - a.t) / 2
    double* rec = A.first_partials + (size_t)blockIdx.x * VS_PLEN(k);
    for (int e = tid; e < 4 * k; e += 256) {
        const int slot = e / k, j = e - slot * k;   // record layout: head, ac[k], bd[k], ad[k], bc[k]
        const int which = (slot == 0 || slot == 2) ? 0 : 1;   // sums against a (0) or b (1)
        double xs = 0.0, xt = 0.0;
        for (int gg = 0; gg < 16; ++gg) {
            xs += r2[((size_t)gg * (NB * 16) + j) * 2 + which];
            xt += r2[((size_t)gg * (NB * 16) + A.K + j) * 2 + which];
        }
        const bool with_c = slot == 0 || slot == 3;
        rec[VS_P_HEAD + e] = with_c ? 0.5 * (xs + xt) : 0.5 * (xs - xt);
    }
    if (tid < VS_P_HEAD) {
        double sum = 0.0;
        for (int gg = 0; gg < 16; ++gg) sum += r3[gg * 5 + tid];
        rec[tid] = sum;
    }
}

// Up to 32 index rows per half (2K <= 64 tile rows), wave-private form: no workgroup barrier in the loop.  A wave
// owns 16-sample-row chunks (the four waves of a workgroup take adjacent chunks: 512 contiguous bytes per index
// row between them), keeps the accumulator tiles of the [s rows; t rows] tile in registers and stages its chunk
// through a private LDS tile [16 NB tile rows][16 sample rows], pitch 20 doubles:
//   global: lane L loads 16 B = sample rows 2(L&7), 2(L&7)+1 of c_u and of d_u, u = 8j + (L>>3): one
//           instruction = 8 index rows x 128 contiguous bytes, non-temporal;
//   LDS:    ds_write_b128 of s = c + d to tile row u and of t = c - d to tile row K + u, then per 16-row block two
//           ds_read_b128 fragments (tile row = lane&15, sample rows 2q, 2q+1 and 8+2q, 8+2q+1, q = lane>>4) --
//           pitch 20 is conflict-free for the b128 lane groups of gfx950 (MI355X_MICROARCH.md, LDS);
//   MFMA:   4 k-steps x the needed tiles (S x T tiles are skipped); A and B fragments come from the same
//           lane->sample map, and a sum over sample rows does not care in which order they are contracted.
// The next chunk's global loads are in flight while the MFMAs of the current one run.

//
// FIRST (vs_reduce_first_second_order; l == 1, no mask): the first-order sums of reduce_first_kernel ride
// along.  a = fM_1 and b = fM_2 are loaded in the fragment layout (the sample rows a lane's fragments hold),
// so each fragment element meets its a and b in two fmas: tile row u < K is s_u (-> a.s, b.s), row K + u is
// t_u (-> a.t, b.t); gram_first_record turns them into a.c, b.c, a.d, b.d.  Sixteen lanes hold the same a, b;
// lane&15 == 0 of each group carries the scalar sums (sab, sa, sb, shifted variance sums).
// Waves per SIMD (= resident workgroups per CU) by the number of live 16-row blocks: with few index rows a wave
// has little in flight (16 sample rows x 16 NB index rows), so the small shapes -- BASELINE configs[1], [2] have
// 12 and 6 index rows -- run more waves: they are bound by bytes in flight, not by the matrix pipe.
// Round 4: without the first-order sums the kernel ran 6 / 3 / 3 waves per SIMD at 2 / 3 / 4 live blocks -- 80 and 168
// registers, where the code objects showed 2 and 43-69 spilled dwords (the run-time tile set of four blocks keeps ten
// accumulator tiles): 5 / 3 / 2 waves now, nothing spilled; measured at n = 2^22 (profiles/r04/ab_wave_occupancy.txt):
// k=10 l=3 0.508 -> 0.391 ms, k=6 l=2 0.154 -> 0.140, k=16 l=2 0.400 -> 0.410.
#define VS2_WAVES_PER_SIMD(NB, FIRST) \
    ((FIRST) ? ((NB) == 1 ? 5 : (NB) == 2 ? 3 : 2) : ((NB) == 1 ? 8 : (NB) == 2 ? 5 : (NB) == 3 ? 3 : 2))
// LDS doubles per workgroup: the four wave tiles (16 NB rows each), and the epilogue's scratch
#define VS2_WAVE_LDS(NB, FIRST) \
    ((4 * 16 * (NB) * VS2_WPITCH) > 1024 + ((FIRST) ? 16 * (NB) * 16 * 2 + 80 : 0) \
         ? (4 * 16 * (NB) * VS2_WPITCH) : 1024 + ((FIRST) ? 16 * (NB) * 16 * 2 + 80 : 0))

// FULL: K == 8 NB exactly (k = 16, 32 with one output): blocks 0 .. NB/2-1 are all S, the rest all T, so the
// needed tiles are known at compile time and the others cost no registers.
template <int NB, bool MASKED, bool ALIGNED, bool FIRST, bool FULL>
__global__ void __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(VS2_WAVES_PER_SIMD(NB, FIRST), VS2_WAVES_PER_SIMD(NB, FIRST))))
gram_diag_wave_kernel(vs_gram_args A) {
    extern __shared__ double tiles[];  // [4 waves][16 NB][VS2_WPITCH] (VS2_WAVE_LDS doubles)
    if (MASKED && A.counters[CNT_NANFLAG] == 0) return;
    if (FIRST && blockIdx.x == 0 && threadIdx.x == 0) {
        // this kernel opens the call: reset the counters the two folds use (cf. reduce_first_kernel)
        uint32_t* c = const_cast<uint32_t*>(A.counters);
        c[0] = 0;            // NaN flag
        c[1] = 0xffffffffu;  // first kept row
        c[2] = 0;            // kept rows
        c[3] = 0;            // fold tickets
        c[4] = 0;
    }
    constexpr int NT = NB * (NB + 1) / 2;
    constexpr int NJ = 2 * NB;          // loads per chunk: NB pairs (c row u, d row u), u = 8j + (lane >> 3)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = A.K;
    const uint64_t rows = A.rows;
    double* tile = tiles + wave * (16 * NB * VS2_WPITCH);

    // index rows of the tile: s_u = c_u + d_u at row u, t_u = c_u - d_u at row K + u (u < K); rows from 2K on are
    // padding, zeroed once.  A lane loads c and d of the SAME u and sample rows, so s and t form in registers.
    const int lrow = lane >> 3, piece = lane & 7;
    uint32_t pl[NJ];    // planes of the rows this lane loads: [2j] = c row, [2j + 1] = d row
    bool ok[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int u = 8 * j + lrow;
        ok[j] = u < K;
        pl[2 * j] = x_plane(A, ok[j] ? u : 0);
        pl[2 * j + 1] = x_plane(A, K + (ok[j] ? u : 0));
    }
    for (int e = 2 * K * VS2_WPITCH + lane; e < 16 * NB * VS2_WPITCH; e += 64) tile[e] = 0.0;
    double* wr = tile + lrow * VS2_WPITCH + 2 * piece;                       // + 8j rows: s; + K rows more: t
    double* wrt = wr + K * VS2_WPITCH;
    // tiles of an S block times a T block are never needed (block b is all S when 16(b+1) <= K, all T when 16b >= K)
    bool need[NT];
    {
        int a = 0;
#pragma unroll
        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int bj = bi; bj < NB; ++bj) {
                need[a] = FULL ? (bi < NB / 2) == (bj < NB / 2) : !(16 * (bi + 1) <= K && 16 * bj >= K);
                ++a;
            }
    }
    const double* rd = tile + (lane & 15) * VS2_WPITCH + 2 * (lane >> 4);    // + 16b rows, + 8h

    vs_d4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = (vs_d4){0.0, 0.0, 0.0, 0.0};

    const uint64_t nfull = rows / VS2_WCHUNK;
    const uint64_t nch = nfull + ((rows % VS2_WCHUNK) ? 1 : 0);
    const uint64_t W = (uint64_t)gridDim.x * 4;
    // first-order state (FIRST only): planes 0 and 1 of the single output are a and b
    const int fq = lane >> 4;
    double shift = 0.0;
    if (FIRST) shift = vs_shift_estimate(A.f, rows);  // wave-wide call, same bits in every wave and in the fold
    vs_d2 an[2], bn[2];
    double pa[NB], pb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) pa[b] = pb[b] = 0.0;
    double t_sab = 0.0, t_sa = 0.0, t_sb = 0.0, t_s1 = 0.0, t_s2 = 0.0;
    vs_d2 raw[NJ];
    bool k0 = true, k1 = true;
    // fetch only issues the loads; padding / keep masks are applied where the values are consumed,
    // one iteration later, so nothing waits on them while the MFMAs of the current chunk run
    auto fetch = [&](uint64_t c) {
        const uint64_t s0 = c * VS2_WCHUNK;
        VS2_FTILE(s0);
        if (c < nfull) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const double* q = fb + (uint64_t)pl[j] * fp + 2 * piece;
                if (ALIGNED) {
                    raw[j] = __builtin_nontemporal_load((const vs_d2*)q);
                } else {  // odd row count: planes start 8 bytes off a 16-byte boundary
                    raw[j].x = __builtin_nontemporal_load(q);
                    raw[j].y = __builtin_nontemporal_load(q + 1);
                }
            }
        } else {  // the ragged last chunk
            const uint64_t s = s0 + 2 * piece;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const double* q = fb + (uint64_t)pl[j] * fp + 2 * piece;
                raw[j].x = s < rows ? q[0] : 0.0;
                raw[j].y = s + 1 < rows ? q[1] : 0.0;
            }
        }
        if (MASKED) {
            const uint64_t s = s0 + 2 * piece;
            k0 = s < rows && A.keep[s];
            k1 = s + 1 < rows && A.keep[s + 1];
        }
        if (FIRST) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint64_t s = s0 + 8 * h;
                const double* ap = fb + 2 * fq + 8 * h;   // planes 0 and 1 of the single output: a = fM_1, b = fM_2
                const double* bp = ap + fp;
                if (c < nfull && ALIGNED) {
                    an[h] = __builtin_nontemporal_load((const vs_d2*)ap);
                    bn[h] = __builtin_nontemporal_load((const vs_d2*)bp);
                } else {
                    const uint64_t e = s + 2 * fq;
                    an[h].x = e < rows ? ap[0] : 0.0;
                    an[h].y = e + 1 < rows ? ap[1] : 0.0;
                    bn[h].x = e < rows ? bp[0] : 0.0;
                    bn[h].y = e + 1 < rows ? bp[1] : 0.0;
                }
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            vs_d2 c = raw[2 * j], d = raw[2 * j + 1];
            if (MASKED) {
                c.x = k0 ? c.x : 0.0;
                c.y = k1 ? c.y : 0.0;
                d.x = k0 ? d.x : 0.0;
                d.y = k1 ? d.y : 0.0;
            }
            if (ok[j]) {   // rows u >= K of the s part ARE rows of the t part: nothing may be written for them
                *(vs_d2*)(wr + 8 * j * VS2_WPITCH) = c + d;
                *(vs_d2*)(wrt + 8 * j * VS2_WPITCH) = c - d;
            }
        }
    };

    uint64_t c = (uint64_t)blockIdx.x * 4 + wave;
    if (c < nch) fetch(c);
    for (; c < nch; c += W) {
        stash();
        vs_d2 ac_[2], bc_[2];
        if (FIRST) {
            ac_[0] = an[0]; ac_[1] = an[1];
            bc_[0] = bn[0]; bc_[1] = bn[1];
        }
        if (c + W < nch) fetch(c + W);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        vs_d2 fr[NB][2];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            fr[b][0] = *(const vs_d2*)(rd + 16 * b * VS2_WPITCH);
            fr[b][1] = *(const vs_d2*)(rd + 16 * b * VS2_WPITCH + 8);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                int a = 0;
#pragma unroll
                for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                    for (int bj = bi; bj < NB; ++bj) {
                        if (need[a]) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[bi][h][t], fr[bj][h][t], acc[a], 0, 0, 0);
                        ++a;
                    }
            }
        if (FIRST) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const double a = ac_[h][t], b = bc_[h][t];
#pragma unroll
                    for (int blk = 0; blk < NB; ++blk) {
                        pa[blk] = fma(a, fr[blk][h][t], pa[blk]);
                        pb[blk] = fma(b, fr[blk][h][t], pb[blk]);
                    }
                    const bool live = c < nfull || c * VS2_WCHUNK + 8 * h + 2 * fq + t < rows;
                    const double dx = live ? a - shift : 0.0;
                    const double dy = live ? b - shift : 0.0;
                    t_sab = fma(a, b, t_sab);
                    t_sa += a;
                    t_sb += b;
                    t_s1 += dx + dy;
                    t_s2 = fma(dx, dx, t_s2);
                    t_s2 = fma(dy, dy, t_s2);
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // add the four waves' accumulators through LDS (the tiles are free now), fixed order
    __syncthreads();
    double* red = tiles;
    double* out = A.partials + (size_t)blockIdx.x * VS2_TILE_OUT;   // one 64-row block
    int a = 0;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int bj = bi; bj < NB; ++bj) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = (lane >> 4) + 4 * v, col = lane & 15;   // f64 C/D layout
                red[(wave * 256) + row * 16 + col] = acc[a][v];
            }
            __syncthreads();
            {
                const double t = ((red[tid] + red[256 + tid]) + red[512 + tid]) + red[768 + tid];
                const int row = tid >> 4, col = tid & 15;
                out[(bi * 16 + row) * VS2_SUPER + (bj * 16 + col)] = t;
            }
            __syncthreads();
            ++a;
        }
    if (FIRST) gram_first_record<NB>(A, tiles, pa, pb, t_sab, t_sa, t_sb, t_s1, t_s2);
}

// Wide form of the wave-private kernel (one output, even row count, no NaN mask, 17 .. 32 index rows per half): the
// same tile of s and t rows, accumulators and fragments, but a wave stages 32 sample rows at a time and every
// load instruction reads 4 index rows x 256 contiguous bytes.  Measured on MI355X (tools/load_patterns.hip): a wave-wide 16-byte load
// that gathers 8 x 128 B runs from 8 planes sustains 4.9 TB/s however many are in flight; 4 x 256 B runs
// sustain 6.7 TB/s, 1 x 1 KB 7.0 TB/s.  With one output c_u is plane 2 + u and d_u plane 2 + K + u of f, so the
// lane's pointer advances by wave-uniform strides from one load to the next (no pointer per load in registers).
//   global: pair j, lane L: 16 B = sample rows 2(L&15), 2(L&15)+1 of c_u and of d_u, u = 4j + (L>>4);
//   LDS:    tile [16 NB index rows (+ a, b rows)][32 sample rows], pitch 36 doubles: the 16 lanes of an index
//           row write 256 contiguous bytes; a b128 fragment read (index row r = lane&15, lane group q = lane>>4:
//           dword 72 r + 4 q + 16 quarter) is served in the four 16-lane groups of MI355X_MICROARCH.md (LDS) --
//           {0-3, 12-15, 20-27}, ... -- and with 72 r mod 64 = 8 r the sixteen lanes of every group land in sixteen
//           different 4-bank sets: conflict-free.  Pitch 34 (rounds 1-2, "eight distinct groups per 8 lanes") put
//           rows r = 12, q = 0 and r = 11, q = 1 of one group on the same banks: SQ_LDS_BANK_CONFLICT was 63 % of the
//           LDS-active cycles (profiles/r03/gram_k32_n22_before.json); 32, 38 .. 42, 46, 48 conflict too, 36 and 44 do not;
//   MFMA:   four 8-sample-row quarters x 2 k-steps x the needed tiles per staged unit.
// 64 VGPRs of loads in flight per wave (16 KiB), two waves per SIMD: 128 KiB in flight per CU.
// FIRST: a = fM_1, b = fM_2 of the unit are one more load (planes 0, 1) into two more tile rows; every lane
// group reads them back in the fragment layout and meets each fragment element in two fmas (a.s, b.s, a.t, b.t;
// gram_first_record turns them into the record's a.c, b.c, a.d, b.d).  The five scalar sums of the record come from first_head_kernel, as on the macro route: they
// would cost 12 more VGPRs here, and the kernel sits at 256.
#define VS2_XCHUNK 32
#define VS2_XPITCH 36
#define VS2_WIDE_ROWS(NB, FIRST) (16 * (NB) + ((FIRST) ? 4 : 0))
#define VS2_WIDE_LDS(NB, FIRST) \
    ((4 * VS2_WIDE_ROWS(NB, FIRST) * VS2_XPITCH) > 1024 + ((FIRST) ? 16 * (NB) * 16 * 2 + 80 : 0) \
         ? (4 * VS2_WIDE_ROWS(NB, FIRST) * VS2_XPITCH) : 1024 + ((FIRST) ? 16 * (NB) * 16 * 2 + 80 : 0))

template <int NB, bool FIRST, bool FULL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
gram_diag_wide_kernel(vs_gram_args A) {
    extern __shared__ double tiles[];  // [4 waves][VS2_WIDE_ROWS][VS2_XPITCH] (VS2_WIDE_LDS doubles)
    if (FIRST && blockIdx.x == 0 && threadIdx.x == 0) {
        // this kernel opens the call: reset the counters the two folds use (cf. reduce_first_kernel)
        uint32_t* c = const_cast<uint32_t*>(A.counters);
        c[0] = 0;            // NaN flag
        c[1] = 0xffffffffu;  // first kept row
        c[2] = 0;            // kept rows
        c[3] = 0;            // fold tickets
        c[4] = 0;
    }
    constexpr int NT = NB * (NB + 1) / 2;
    constexpr int NJ = 4 * NB;          // loads per unit: 2 NB pairs (c row u, d row u), u = 4j + (lane >> 4)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = A.K;
    const uint64_t rows = A.rows;
    double* tile = tiles + wave * (VS2_WIDE_ROWS(NB, FIRST) * VS2_XPITCH);

    const int mrow = lane >> 4, li = lane & 15;
    // tile rows: s_u = c_u + d_u at row u, t_u = c_u - d_u at row K + u; rows from 2K on are padding, zeroed once.
    // One output: c_u is plane 2 + u, d_u plane 2 + K + u of f; rows u >= K read u = 0 and are not stored.
    // (per fetch: p0 = c row mrow of the unit's rows, jstride = four planes on, dstride = K planes on: the d rows)
    for (int e = 2 * K * VS2_XPITCH + lane; e < 16 * NB * VS2_XPITCH; e += 64) tile[e] = 0.0;
    double* wr = tile + mrow * VS2_XPITCH + 2 * li;                          // + 4j rows: s; + K rows more: t
    double* wrt = wr + K * VS2_XPITCH;
    bool need[NT];   // S block x T block tiles are never needed
    {
        int a = 0;
#pragma unroll
        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int bj = bi; bj < NB; ++bj) {
                need[a] = FULL ? (bi < NB / 2) == (bj < NB / 2) : !(16 * (bi + 1) <= K && 16 * bj >= K);
                ++a;
            }
    }
    const double* rd = tile + (lane & 15) * VS2_XPITCH + 2 * (lane >> 4);    // + 16b rows, + 8 per quarter
    // FIRST: lanes 0..15 load a (plane 0), 16..31 b (plane 1), the rest load them again into two spare rows
    double* wrab = tile + (16 * NB + mrow) * VS2_XPITCH + 2 * li;
    const double* rdab = tile + 16 * NB * VS2_XPITCH + 2 * (lane >> 4);      // a; b one tile row further

    vs_d4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = (vs_d4){0.0, 0.0, 0.0, 0.0};

    const uint64_t nfull = rows / VS2_XCHUNK;
    const uint64_t nun = nfull + ((rows % VS2_XCHUNK) ? 1 : 0);
    const uint64_t W = (uint64_t)gridDim.x * 4;
    double pa[NB], pb[NB];   // FIRST: a.x and b.x of tile row lane&15 of every block
#pragma unroll
    for (int b = 0; b < NB; ++b) pa[b] = pb[b] = 0.0;
    // The five scalar sums of the record (sab, sa, sb, shifted variance sums): with the compile-time tile set (FULL:
    // six accumulator tiles instead of ten) they fit beside everything else; otherwise first_head_kernel adds them.
    constexpr bool HEAD = FIRST && FULL;
    const int fq = lane >> 4;
    double shift = 0.0;
    if (HEAD) shift = vs_shift_estimate(A.f, rows);  // wave-wide call, same bits in every wave and in the fold
    double t_sab = 0.0, t_sa = 0.0, t_sb = 0.0, t_s1 = 0.0, t_s2 = 0.0;
    vs_d2 raw[NJ], rawab = (vs_d2){0.0, 0.0};
    auto fetch = [&](uint64_t u) {
        const uint64_t s0 = u * VS2_XCHUNK;
        VS2_FTILE(s0);
        const double* p0 = fb + (uint64_t)(2 + mrow) * fp + 2 * li;           // c row mrow
        const uint64_t jstride = 4 * fp, dstride = (uint64_t)K * fp;
        const double* pab = fb + (uint64_t)(mrow & 1) * fp + 2 * li;
        if (u < nfull) {
#pragma unroll
            for (int j = 0; j < NJ / 2; ++j) {
                const double* q = (4 * j + mrow < K ? p0 + j * jstride : p0);
                raw[2 * j] = __builtin_nontemporal_load((const vs_d2*)q);
                raw[2 * j + 1] = __builtin_nontemporal_load((const vs_d2*)(q + dstride));
            }
            if (FIRST) rawab = __builtin_nontemporal_load((const vs_d2*)pab);
        } else {  // the ragged last unit
            const uint64_t s = s0 + 2 * li;
#pragma unroll
            for (int j = 0; j < NJ / 2; ++j) {
                const double* q = (4 * j + mrow < K ? p0 + j * jstride : p0);
                raw[2 * j].x = s < rows ? q[0] : 0.0;
                raw[2 * j].y = s + 1 < rows ? q[1] : 0.0;
                raw[2 * j + 1].x = s < rows ? q[dstride] : 0.0;
                raw[2 * j + 1].y = s + 1 < rows ? q[dstride + 1] : 0.0;
            }
            if (FIRST) {
                rawab.x = s < rows ? pab[0] : 0.0;
                rawab.y = s + 1 < rows ? pab[1] : 0.0;
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < NJ / 2; ++j) {
            if (4 * j + mrow < K) {   // rows u >= K of the s part ARE rows of the t part
                *(vs_d2*)(wr + 4 * j * VS2_XPITCH) = raw[2 * j] + raw[2 * j + 1];
                *(vs_d2*)(wrt + 4 * j * VS2_XPITCH) = raw[2 * j] - raw[2 * j + 1];
            }
        }
        if (FIRST) *(vs_d2*)wrab = rawab;
    };

    uint64_t u = (uint64_t)blockIdx.x * 4 + wave;
    if (u < nun) fetch(u);
    for (; u < nun; u += W) {
        stash();
        if (u + W < nun) fetch(u + W);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {   // four 8-sample-row quarters of the unit: sample rows 8 hh + 2q, + 1
            vs_d2 fr[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) fr[b] = *(const vs_d2*)(rd + 16 * b * VS2_XPITCH + 8 * hh);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                int a = 0;
#pragma unroll
                for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                    for (int bj = bi; bj < NB; ++bj) {
                        if (need[a]) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[bi][t], fr[bj][t], acc[a], 0, 0, 0);
                        ++a;
                    }
            }
            if (FIRST) {
                const vs_d2 av = *(const vs_d2*)(rdab + 8 * hh);
                const vs_d2 bv = *(const vs_d2*)(rdab + VS2_XPITCH + 8 * hh);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const double a = av[t], b = bv[t];
#pragma unroll
                    for (int blk = 0; blk < NB; ++blk) {
                        pa[blk] = fma(a, fr[blk][t], pa[blk]);
                        pb[blk] = fma(b, fr[blk][t], pb[blk]);
                    }
                    if (HEAD) {   // sixteen lanes hold the same a, b: gram_first_record takes lane&15 == 0 of each group
                        const bool live = u < nfull || u * VS2_XCHUNK + 8 * hh + 2 * fq + t < rows;
                        const double dx = live ? a - shift : 0.0;
                        const double dy = live ? b - shift : 0.0;
                        t_sab = fma(a, b, t_sab);
                        t_sa += a;
                        t_sb += b;
                        t_s1 += dx + dy;
                        t_s2 = fma(dx, dx, t_s2);
                        t_s2 = fma(dy, dy, t_s2);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // add the four waves' accumulators through LDS (the tiles are free now), fixed order
    __syncthreads();
    double* red = tiles;
    double* out = A.partials + (size_t)blockIdx.x * VS2_TILE_OUT;
    int a = 0;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int bj = bi; bj < NB; ++bj) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = (lane >> 4) + 4 * v, col = lane & 15;   // f64 C/D layout
                red[(wave * 256) + row * 16 + col] = acc[a][v];
            }
            __syncthreads();
            {
                const double t = ((red[tid] + red[256 + tid]) + red[512 + tid]) + red[768 + tid];
                const int row = tid >> 4, col = tid & 15;
                out[(bi * 16 + row) * VS2_SUPER + (bj * 16 + col)] = t;
            }
            __syncthreads();
            ++a;
        }
    if (FIRST) gram_first_record<NB>(A, tiles, pa, pb, t_sab, t_sa, t_sb, t_s1, t_s2);
}

// More than 64 index rows (k*l > 32): 128-row macro blocks.  One workgroup of eight waves owns a pair
// (MI <= MJ) of macro blocks and a strided set of 16-sample-row chunks; the chunk of BOTH blocks (256 index
// rows x 16 sample rows, 32 KiB) is staged once in LDS -- double-buffered, one workgroup barrier per chunk --
// and feeds all 64 output tiles of the 128 x 128 product: 2*128*128 flop per 8*256 bytes = 16 flop/byte, twice
// the fp64 matrix pipe's break-even with HBM (64-row blocks sit exactly on it: the pair kernel they had was
// reading 4.7 TB/s at k = 128).  Wave w owns block row w: its A fragments come from index rows 16w.. of MI,
// its B fragments from the eight 16-row blocks of MJ; on a diagonal pair (MI == MJ) only the blocks bj >= w
// are needed (wave-uniform skip; two workgroups share a CU, so the shorter waves leave the pipe to others).
// LDS tile [256 index rows][16 sample rows], pitch 20 doubles, lane -> sample map and ds_read_b128 fragments
// exactly as in gram_diag_wave_kernel.

//
// FIRST (vs_reduce_first_second_order; one output, no mask): the first-order sums of reduce_first_kernel ride
// on the DIAGONAL pairs, where every index row of X appears exactly once: a = fM_1 and b = fM_2 of the chunk
// are staged in the padding of the LDS tile rows, each A fragment element meets them in two fmas (index row
// m < k is c_m -> a.c, b.c; m >= k is d_{m-k} -> a.d, b.d).  One first-order record per diagonal workgroup,
// folded like reduce_first_kernel's; sab, sa, sb and the shifted variance sums -- one pass over fM_1, fM_2 only,
// 1/(k+1) of the data -- come from first_head_kernel's records in the same fold.
// WIDE (one output, even row count): the workgroup stages TWO chunks per fetch -- thread t loads 16 B = sample rows
// 2(t&15), 2(t&15)+1 of the 32, of index rows (t>>4) + 32j of MI (j < 4) and of MJ, so that a wave-wide load
// reads 4 index rows x 256 contiguous bytes instead of 8 x 128 (4.9 -> 6.7 TB/s for the pattern alone,
// tools/load_patterns.hip) -- and the half of the threads that hold a chunk's sample rows write it to the tile.
// With one output the index row m is plane 2 + m: one pointer per macro block, wave-uniform strides between loads.
// The body is compiled twice, for diagonal and off-diagonal pairs (a workgroup-uniform branch in the kernel): a
// diagonal pair stages one macro block only and alone carries the first-order sums, so neither form pays the
// other's registers.
template <bool MASKED, bool ALIGNED, bool FIRST, bool WIDE, bool DIAG>
__device__ __forceinline__ void gram_macro_body(const vs_gram_args& A, const int MI, const int MJ, double* tiles) {
    constexpr bool diag = DIAG;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7 = block row of the output
    const int M = 2 * A.K;
    const uint64_t rows = A.rows;
    constexpr int BUF = 2 * VS2_MACRO * VS2_WPITCH;  // doubles per buffer
    constexpr bool first = FIRST && diag;   // this workgroup also carries first-order dot products

    // staging: thread t loads 16 B (sample rows 2p, 2p+1) of index rows lrow, 64+lrow of MI and of MJ
    // (WIDE: rows lrow + 32j, j < 4, of MI and of MJ; 16 pieces = 32 sample rows)
    constexpr int NLD = WIDE ? 8 : 4;
    const int lrow = WIDE ? tid >> 4 : tid >> 3, piece = WIDE ? tid & 15 : tid & 7;
    uint32_t pl[WIDE ? 1 : 4];   // (not WIDE) planes of the index rows this thread loads
    bool ok[NLD];
    if (WIDE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ok[j] = (j < 4 ? MI : MJ) * VS2_MACRO + (j & 3) * 32 + lrow < M && !(diag && j >= 4);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = (j < 2 ? MI : MJ) * VS2_MACRO + (j & 1) * 64 + lrow;
            ok[j] = m < M && !(diag && j >= 2);
            pl[j] = x_plane(A, ok[j] ? m : 0);
        }
    }
    // WIDE: the address of load j is rebuilt from the index row at every fetch (a handful of integer
    // instructions) instead of living in registers between fetches: index row m is plane 2 + m, rows past 2k
    // read index row 0 and are zeroed on the way into the tile; fb, fp: the fetch's chunk (VS2_FTILE)
    auto wide_ptr = [&](int j, int lr, const double* fb, uint64_t fp) {
        const int m = (j < 4 ? MI : MJ) * VS2_MACRO + (j & 3) * 32 + lr;
        return fb + 2 * piece + (uint64_t)(2 + (m < M ? m : 0)) * fp;
    };
    double* wr = tiles + lrow * VS2_WPITCH + 2 * (piece & 7);   // + 64j (WIDE: 32j) rows, + buffer
    // fragments: A = index row 16*wave + (lane&15) of MI's half; B = 16*bj + (lane&15) of MJ's half
    const double* rdA = tiles + (16 * wave + (lane & 15)) * VS2_WPITCH + 2 * (lane >> 4);
    const double* rdB = tiles + ((diag ? 0 : VS2_MACRO) + (lane & 15)) * VS2_WPITCH + 2 * (lane >> 4);
    const int bj0 = diag ? wave : 0;

    vs_d4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (vs_d4){0.0, 0.0, 0.0, 0.0};

    const uint64_t nfull = rows / VS2_WCHUNK;
    const uint64_t nch = nfull + ((rows % VS2_WCHUNK) ? 1 : 0);
    // a workgroup takes GROUP consecutive chunks per visit (512 contiguous bytes per index row)
    constexpr uint64_t GROUP = 4;
    const uint64_t ngroups = (nch + GROUP - 1) / GROUP;
    uint64_t mine = 0;   // chunks this workgroup walks (WIDE: always whole pairs; a chunk past the end is zeros)
    for (uint64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const uint64_t here = (g * GROUP + GROUP <= nch) ? GROUP : nch - g * GROUP;
        mine += WIDE ? (here + 1) & ~(uint64_t)1 : here;
    }
    auto chunk_of = [&](uint64_t i) {   // the i-th chunk of this workgroup
        const uint64_t g = blockIdx.x + (i / GROUP) * gridDim.x;
        return g * GROUP + (i % GROUP);
    };
    vs_d2 raw[NLD];
    bool k0 = true, k1 = true;
    // FIRST: threads 0..7 stage a (plane 0), 8..15 b (plane 1) of the chunk (WIDE: 0..15 and 16..31, of the pair
    // of chunks): sample pair p goes into the padding of tile row p (16 of 20 doubles per row are index-row
    // data): a at [16,17], b at [18,19]
    const bool abth = first && tid < (WIDE ? 32 : 16);
    // the a / b address and sample offset are rebuilt at every fetch (by the 16 or 32 threads that stage a and b: part
    // of wave 0) from an opaque copy of the thread index instead of living in three registers across the MFMA loop:
    // at the kernel's 128 registers they were the three dwords the allocator sent through scratch
    auto ab_lane = [&]() {
        int t = tid;
        asm volatile("" : "+v"(t));
        return t;
    };
    vs_d2 rawab = (vs_d2){0.0, 0.0};
    double pa = 0.0, pb = 0.0;
    auto fetch = [&](uint64_t c) {   // WIDE: c is the first chunk of a pair
        const uint64_t s0 = c * VS2_WCHUNK;
        VS2_FTILE(s0);
        const bool whole = WIDE ? s0 + 2 * VS2_WCHUNK <= rows : c < nfull;
        if (FIRST && abth) {
            const int t = ab_lane();
            const uint64_t sm = s0 + 2 * (WIDE ? t & 15 : t & 7);
            const double* abp = fb + (t >= (WIDE ? 16 : 8) ? fp : 0) + 2 * (WIDE ? t & 15 : t & 7);   // plane 0: a, plane 1: b
            if (whole && ALIGNED) {
                rawab = __builtin_nontemporal_load((const vs_d2*)abp);
            } else {
                rawab.x = sm < rows ? abp[0] : 0.0;
                rawab.y = sm + 1 < rows ? abp[1] : 0.0;
            }
        }
        int lr = lrow;
        if (WIDE) asm volatile("" : "+v"(lr));   // not loop-invariant as far as the compiler can tell
        if (whole) {
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                if (WIDE) {
                    if (diag && j >= 4) continue;
                    raw[j] = __builtin_nontemporal_load((const vs_d2*)wide_ptr(j, lr, fb, fp));
                    continue;
                }
                if (!ok[j]) continue;
                const double* q = fb + (uint64_t)pl[WIDE ? 0 : j] * fp + 2 * piece;
                if (ALIGNED) {
                    raw[j] = __builtin_nontemporal_load((const vs_d2*)q);
                } else {
                    raw[j].x = __builtin_nontemporal_load(q);
                    raw[j].y = __builtin_nontemporal_load(q + 1);
                }
            }
        } else {
            const uint64_t sm = s0 + 2 * piece;
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const double* q = WIDE ? wide_ptr(j, lr, fb, fp) : fb + (uint64_t)pl[WIDE ? 0 : j] * fp + 2 * piece;
                raw[j].x = (ok[j] && sm < rows) ? q[0] : 0.0;
                raw[j].y = (ok[j] && sm + 1 < rows) ? q[1] : 0.0;
            }
        }
        if (MASKED) {
            const uint64_t sm = s0 + 2 * piece;
            k0 = sm < rows && A.keep[sm];
            k1 = sm + 1 < rows && A.keep[sm + 1];
        }
    };
    auto stash = [&](int b, int half) {   // WIDE: the threads holding sample rows 16 half .. of the pair write
        if (WIDE && (piece >> 3) != half) return;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            vs_d2 v = raw[j];
            if (MASKED) {
                v.x = k0 ? v.x : 0.0;
                v.y = k1 ? v.y : 0.0;
            }
            if (!ok[j]) v = (vs_d2){0.0, 0.0};
            if (WIDE) {
                if (!(diag && j >= 4)) *(vs_d2*)(wr + b * BUF + ((j >> 2) * VS2_MACRO + (j & 3) * 32) * VS2_WPITCH) = v;
            } else {
                if (!(diag && j >= 2)) *(vs_d2*)(wr + b * BUF + 64 * j * VS2_WPITCH) = v;
            }
        }
        if (FIRST && abth) {
            const int t = ab_lane();   // (as in fetch: the tile address is not kept across the loop either)
            if (WIDE) *(vs_d2*)(tiles + b * BUF + (t & 7) * VS2_WPITCH + 16 + 2 * (t >> 4)) = rawab;
            else *(vs_d2*)(tiles + b * BUF + (t & 7) * VS2_WPITCH + 16 + 2 * (t >> 3)) = rawab;
        }
    };

    if (mine > 0) {
        fetch(chunk_of(0));
        stash(0, 0);
        if (!WIDE && mine > 1) fetch(chunk_of(1));
    }
    __syncthreads();
    for (uint64_t i = 0; i < mine; ++i) {
        const int b = (int)(i & 1);
        // the next chunk goes into the other buffer (everybody left it at the last barrier) ...
        if (i + 1 < mine) stash(b ^ 1, (int)((i + 1) & 1));
        if (WIDE) {   // ... the registers are free once the second chunk of the pair is in LDS: the next pair
            if (!(i & 1) && i + 2 < mine) fetch(chunk_of(i + 2));
        } else {
            if (i + 2 < mine) fetch(chunk_of(i + 2));
        }
        // ... while this one feeds the matrix pipe
        vs_d2 fa[2];
        fa[0] = *(const vs_d2*)(rdA + b * BUF);
        fa[1] = *(const vs_d2*)(rdA + b * BUF + 8);
        if (FIRST && first) {
            // samples 2q, 2q+1 (h = 0) and 8+2q, 8+2q+1 (h = 1) of the lane group q, as in the fragments
            const double* rdAB = tiles + b * BUF + (lane >> 4) * VS2_WPITCH + 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const vs_d4 ab = *(const vs_d4*)(rdAB + 4 * h * VS2_WPITCH);   // a, a, b, b of two sample rows
                pa = fma(ab[0], fa[h][0], pa);
                pb = fma(ab[2], fa[h][0], pb);
                pa = fma(ab[1], fa[h][1], pa);
                pb = fma(ab[3], fa[h][1], pb);
                // one a / b quadruple alive at a time, and none beside the B fragments of the tiles below (the kernel
                // sits at its 128 registers: unpinned, three dwords went through scratch)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int bj = 0; bj < 8; ++bj) {
            if (bj >= bj0) {
                const vs_d2 f0 = *(const vs_d2*)(rdB + b * BUF + 16 * bj * VS2_WPITCH);
                const vs_d2 f1 = *(const vs_d2*)(rdB + b * BUF + 16 * bj * VS2_WPITCH + 8);
                acc[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0][0], f0[0], acc[bj], 0, 0, 0);
                acc[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0][1], f0[1], acc[bj], 0, 0, 0);
                acc[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1][0], f1[0], acc[bj], 0, 0, 0);
                acc[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1][1], f1[1], acc[bj], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (FIRST && first) {
        // the four lane groups of a wave hold the same index rows over different samples
        pa += __shfl_xor(pa, 16, 64);
        pa += __shfl_xor(pa, 32, 64);
        pb += __shfl_xor(pb, 16, 64);
        pb += __shfl_xor(pb, 32, 64);
        double* red = tiles;   // the tiles are free: [128 index rows][2]
        if (lane < 16) {
            red[(16 * wave + lane) * 2 + 0] = pa;
            red[(16 * wave + lane) * 2 + 1] = pb;
        }
        __syncthreads();
        // this workgroup's first-order record: its own 128 index rows, zeros elsewhere
        const int k = A.k, K = A.K;
        const size_t plen = VS_PLEN(k);
        double* rec = A.first_partials + ((size_t)blockIdx.x * A.nsuper + MI) * plen;
        for (size_t e = tid; e < plen; e += 512) {
            double v = 0.0;
            if (e >= VS_P_HEAD) {   // the five scalar sums come from first_head_kernel's records
                // record layout: head, ac[k], bd[k], ad[k], bc[k]; ac/bc come from the c rows (m = j), ad/bd from d (m = K + j)
                const int slot = (int)((e - VS_P_HEAD) / k), j = (int)((e - VS_P_HEAD) % k);
                const int m = (slot == 0 || slot == 3) ? j : K + j;
                const int local = m - MI * VS2_MACRO;
                if (local >= 0 && local < VS2_MACRO) v = red[local * 2 + ((slot == 0 || slot == 2) ? 0 : 1)];
            }
            rec[e] = v;
        }
    }
    // a tile belongs to one wave: straight to this workgroup's partial (tiles a diagonal pair skips stay
    // unwritten; the fold never reads them)
    const int npairs = A.nsuper * (A.nsuper + 1) / 2;
    double* out = A.partials + ((size_t)blockIdx.x * npairs + blockIdx.y) * VS2_MTILE_OUT;
#pragma unroll
    for (int bj = 0; bj < 8; ++bj) {
        if (bj >= bj0) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = (lane >> 4) + 4 * v, col = lane & 15;   // f64 C/D layout
                out[(wave * 16 + row) * VS2_MACRO + (bj * 16 + col)] = acc[bj][v];
            }
        }
    }
}

template <bool MASKED, bool ALIGNED, bool FIRST, bool WIDE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
gram_macro_kernel(vs_gram_args A) {
    extern __shared__ double tiles[];  // [2 buffers][256][VS2_WPITCH]; FIRST: a, b of the chunk in the row padding
    if (MASKED && A.counters[CNT_NANFLAG] == 0) return;
    if (FIRST && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        // this kernel opens the call: reset the counters the two folds use (cf. reduce_first_kernel)
        uint32_t* c = const_cast<uint32_t*>(A.counters);
        c[0] = 0;
        c[1] = 0xffffffffu;
        c[2] = 0;
        c[3] = 0;
        c[4] = 0;
    }
    int MI, MJ;
    pair_from_index(blockIdx.y, A.nsuper, MI, MJ);   // nsuper = number of macro blocks here
    if (MI == MJ) gram_macro_body<MASKED, ALIGNED, FIRST, WIDE, true>(A, MI, MJ, tiles);
    else gram_macro_body<MASKED, ALIGNED, FIRST, WIDE, false>(A, MI, MJ, tiles);
}

// 33 .. 128 index rows per half (32 < k*l <= 128): the S/T form of the macro kernel.  ONE 256-row tile holds
// s_u = c_u + d_u (rows 0 .. 127, zero from K on) and t_u = c_u - d_u (rows 128 ..): a thread loads c and d of the same u
// and sample rows, so both form in registers -- the same loads as an off-diagonal pair of gram_macro_kernel -- and the
// workgroup computes the upper block triangles of S S^T and T T^T: 72 tiles of 16x16 instead of the 136 of the
// [C; D] Gram matrix.  Wave w owns block row w of S S^T (8 - w tiles) and block row 7 - w of T T^T (w + 1 tiles):
// nine accumulator tiles each, in slots whose meaning depends on w but whose register index does not
// (slot s: S S^T tile (w, w + s) while s <= 7 - w, then T T^T tile (7 - w, s - 1)).  Tiles of all-zero blocks
// (K < 128) are skipped.  Staging, double buffering and the WIDE fetch are gram_macro_kernel's.
// Partials: [workgroup][S S^T, T T^T][128][128].
template <bool MASKED, bool ALIGNED, bool WIDE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
gram_macro_st_kernel(vs_gram_args A) {
    extern __shared__ double tiles[];  // [2 buffers][256][VS2_WPITCH]
    if (MASKED && A.counters[CNT_NANFLAG] == 0) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = A.K;
    const int nbl = (K + 15) / 16;     // live 16-row blocks per half
    const uint64_t rows = A.rows;
    constexpr int BUF = 2 * VS2_MACRO * VS2_WPITCH;  // doubles per buffer

    // staging: loads j < NLD/2 are c rows u_j, loads j >= NLD/2 the d rows of the same u_j
    constexpr int NLD = WIDE ? 8 : 4, NU = NLD / 2;
    const int lrow = WIDE ? tid >> 4 : tid >> 3, piece = WIDE ? tid & 15 : tid & 7;
    constexpr int USTEP = WIDE ? 32 : 64;   // u_j = lrow + USTEP j
    uint32_t pl[WIDE ? 1 : NLD];   // (not WIDE) planes of the rows this thread loads
    if (!WIDE) {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const int u = lrow + USTEP * j;
            pl[j] = x_plane(A, u < K ? u : 0);
            pl[NU + j] = x_plane(A, K + (u < K ? u : 0));
        }
    }
    // WIDE (one output): c_u is plane 2 + u, d_u plane 2 + K + u; the address is rebuilt at every fetch (fb, fp: VS2_FTILE)
    auto wide_ptr = [&](int j, int lr, const double* fb, uint64_t fp) {
        const int u = lr + USTEP * (j % NU);
        return fb + 2 * piece + (uint64_t)(2 + (j >= NU ? K : 0) + (u < K ? u : 0)) * fp;
    };
    double* wr = tiles + lrow * VS2_WPITCH + 2 * (piece & 7);   // + USTEP j rows: s; + 128 rows: t; + buffer
    // fragments
    const double* rdS = tiles + (lane & 15) * VS2_WPITCH + 2 * (lane >> 4);                 // + 16 blk rows
    const double* rdT = rdS + VS2_MACRO * VS2_WPITCH;
    const int nss = wave < nbl ? nbl - wave : 0;               // live S S^T tiles: (wave, wave .. nbl-1)
    const int tr = 7 - wave;                                   // block row of T T^T

    vs_d4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = (vs_d4){0.0, 0.0, 0.0, 0.0};

    const uint64_t nfull = rows / VS2_WCHUNK;
    const uint64_t nch = nfull + ((rows % VS2_WCHUNK) ? 1 : 0);
    constexpr uint64_t GROUP = 4;
    const uint64_t ngroups = (nch + GROUP - 1) / GROUP;
    uint64_t mine = 0;   // chunks this workgroup walks (WIDE: whole pairs; a chunk past the end is zeros)
    for (uint64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const uint64_t here = (g * GROUP + GROUP <= nch) ? GROUP : nch - g * GROUP;
        mine += WIDE ? (here + 1) & ~(uint64_t)1 : here;
    }
    auto chunk_of = [&](uint64_t i) {
        const uint64_t g = blockIdx.x + (i / GROUP) * gridDim.x;
        return g * GROUP + (i % GROUP);
    };
    vs_d2 raw[NLD];
    bool k0 = true, k1 = true;
    auto fetch = [&](uint64_t c) {   // WIDE: c is the first chunk of a pair
        const uint64_t s0 = c * VS2_WCHUNK;
        VS2_FTILE(s0);
        const bool whole = WIDE ? s0 + 2 * VS2_WCHUNK <= rows : c < nfull;
        int lr = lrow;
        if (WIDE) asm volatile("" : "+v"(lr));   // not loop-invariant as far as the compiler can tell
        if (whole) {
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                if (lr + USTEP * (j % NU) >= K) continue;   // rows past K are zeros in the tile: nothing to load
                const double* q = WIDE ? wide_ptr(j, lr, fb, fp) : fb + (uint64_t)pl[WIDE ? 0 : j] * fp + 2 * piece;
                if (ALIGNED) {
                    raw[j] = __builtin_nontemporal_load((const vs_d2*)q);
                } else {
                    raw[j].x = __builtin_nontemporal_load(q);
                    raw[j].y = __builtin_nontemporal_load(q + 1);
                }
            }
        } else {
            const uint64_t sm = s0 + 2 * piece;
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const bool in = lr + USTEP * (j % NU) < K;
                const double* q = WIDE ? wide_ptr(j, lr, fb, fp) : fb + (uint64_t)pl[WIDE ? 0 : j] * fp + 2 * piece;
                raw[j].x = (in && sm < rows) ? q[0] : 0.0;
                raw[j].y = (in && sm + 1 < rows) ? q[1] : 0.0;
            }
        }
        if (MASKED) {
            const uint64_t sm = s0 + 2 * piece;
            k0 = sm < rows && A.keep[sm];
            k1 = sm + 1 < rows && A.keep[sm + 1];
        }
    };
    auto stash = [&](int b, int half) {   // WIDE: the threads holding sample rows 16 half .. of the pair write
        if (WIDE && (piece >> 3) != half) return;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            vs_d2 c = raw[j], d = raw[NU + j];
            if (MASKED) {
                c.x = k0 ? c.x : 0.0;
                c.y = k1 ? c.y : 0.0;
                d.x = k0 ? d.x : 0.0;
                d.y = k1 ? d.y : 0.0;
            }
            if (lrow + USTEP * j >= K) c = d = (vs_d2){0.0, 0.0};
            *(vs_d2*)(wr + b * BUF + USTEP * j * VS2_WPITCH) = c + d;
            *(vs_d2*)(wr + b * BUF + (VS2_MACRO + USTEP * j) * VS2_WPITCH) = c - d;
        }
    };

    if (mine > 0) {
        fetch(chunk_of(0));
        stash(0, 0);
        if (!WIDE && mine > 1) fetch(chunk_of(1));
    }
    __syncthreads();
    for (uint64_t i = 0; i < mine; ++i) {
        const int b = (int)(i & 1);
        if (i + 1 < mine) stash(b ^ 1, (int)((i + 1) & 1));
        if (WIDE) {
            if (!(i & 1) && i + 2 < mine) fetch(chunk_of(i + 2));
        } else {
            if (i + 2 < mine) fetch(chunk_of(i + 2));
        }
        vs_d2 fa[2];
        fa[0] = *(const vs_d2*)(rdS + b * BUF + 16 * wave * VS2_WPITCH);
        fa[1] = *(const vs_d2*)(rdS + b * BUF + 16 * wave * VS2_WPITCH + 8);
#pragma unroll
        for (int sl = 0; sl < 9; ++sl) {
            const bool ss = sl < 8 - wave;                 // wave-uniform
            if (sl == 8 - wave) {                          // from here on: row block 7 - wave of T T^T
                fa[0] = *(const vs_d2*)(rdT + b * BUF + 16 * tr * VS2_WPITCH);
                fa[1] = *(const vs_d2*)(rdT + b * BUF + 16 * tr * VS2_WPITCH + 8);
            }
            const int blk = ss ? wave + sl : sl - 1;       // block of the other operand
            const bool live = ss ? sl < nss : (tr < nbl && blk < nbl);
            if (live) {
                const double* rb = (ss ? rdS : rdT) + b * BUF + 16 * blk * VS2_WPITCH;
                const vs_d2 f0 = *(const vs_d2*)rb;
                const vs_d2 f1 = *(const vs_d2*)(rb + 8);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0][0], f0[0], acc[sl], 0, 0, 0);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0][1], f0[1], acc[sl], 0, 0, 0);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1][0], f1[0], acc[sl], 0, 0, 0);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1][1], f1[1], acc[sl], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // a tile belongs to one wave: straight to this workgroup's partial (all-zero and mirror tiles stay unwritten;
    // the fold never reads them)
    double* out = A.partials + (size_t)blockIdx.x * 2 * VS2_MTILE_OUT;
#pragma unroll
    for (int sl = 0; sl < 9; ++sl) {
        const bool ss = sl < 8 - wave;
        const int blk = ss ? wave + sl : sl - 1;
        const int rowblk = ss ? wave : tr;
        const bool live = ss ? sl < nss : (tr < nbl && blk < nbl);
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

// 33 .. 64 index rows per half, even row count: the S/T macro kernel with TWO chunks per tile.  With at
// most four live 16-row blocks per half the 256-row tile of gram_macro_st_kernel is half zeros and half of its waves
// hold one or two tiles per barrier; here the tile holds [s (64 rows); t (64 rows)] of chunk 2i in rows 0..127 and of
// chunk 2i+1 in rows 128..255 -- exactly what one WIDE fetch brings (32 sample rows: the threads with sample rows
// 0..15 write the first half of the tile, the others the second) -- and waves 0..3 / 4..7 each take one chunk:
// wave r of a group owns block row r of S S^T (4 - r tiles) and block row 3 - r of T T^T (r + 1 tiles), five
// accumulator tiles in w-independent slots as in gram_macro_st_kernel.  One barrier per 32 sample rows, 20 MFMAs per
// wave between barriers.  Two partials per workgroup (one per wave group): [S S^T, T T^T][64][64].
// Since round 3 this kernel serves the NaN-masked pass and several outputs only: the unmasked pass of one output --
// with or without the first-order dot products -- runs on gram_st_kernel (vs_gram_st.hip).
#define VS2_HALF 64
#define VS2_HTILE_OUT (VS2_HALF * VS2_HALF)
// ONE: one output -- c_u is plane 2 + u, d_u plane 2 + K + u, rebuilt at every fetch; several outputs keep a pointer per load.
template <bool MASKED, bool ONE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
gram_half_st_kernel(vs_gram_args A) {
    extern __shared__ double tiles[];  // [2 buffers][2 chunks][128][VS2_WPITCH]
    if (MASKED && A.counters[CNT_NANFLAG] == 0) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, r = wave & 3;   // chunk of the pair this wave works on; its block row of S S^T
    const int tr = 3 - r;                      // its block row of T T^T
    const int K = A.K;
    const int nbl = (K + 15) / 16;             // live 16-row blocks per half (3 or 4)
    const uint64_t rows = A.rows;
    constexpr int BUF = 2 * VS2_MACRO * VS2_WPITCH;  // doubles per buffer

    // staging: thread t holds sample rows 2(t&15), 2(t&15)+1 of the 32 of c_u and d_u, u = (t>>4) + 32j, j < 2
    const int lrow = tid >> 4, piece = tid & 15;
    uint32_t pl[ONE ? 1 : 4];   // (several outputs) planes of the rows this thread loads
    if (!ONE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = lrow + 32 * (j & 1);
            pl[j] = x_plane(A, (j >= 2 ? K : 0) + (u < K ? u : 0));
        }
    }
    auto ptr = [&](int j, int lr, const double* fb, uint64_t fp) {   // j < 2: c rows, else d rows; fb, fp: VS2_FTILE of the fetch
        const int u = lr + 32 * (j & 1);
        return fb + 2 * piece + (uint64_t)(ONE ? (uint32_t)(2 + (j >= 2 ? K : 0) + (u < K ? u : 0)) : pl[ONE ? 0 : j]) * fp;
    };
    double* wr = tiles + ((piece >> 3) * VS2_MACRO + lrow) * VS2_WPITCH + 2 * (piece & 7);   // + 32j rows: s; + 64: t
    const double* rdS = tiles + (grp * VS2_MACRO + (lane & 15)) * VS2_WPITCH + 2 * (lane >> 4);   // + 16 blk rows
    const double* rdT = rdS + VS2_HALF * VS2_WPITCH;

    vs_d4 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] = (vs_d4){0.0, 0.0, 0.0, 0.0};

    // a workgroup takes two consecutive pairs of chunks per visit (512 contiguous bytes per index row)
    const uint64_t npairs = (rows + 2 * VS2_WCHUNK - 1) / (2 * VS2_WCHUNK);
    const uint64_t ngroups = (npairs + 1) / 2;
    uint64_t mine = 0;   // pairs this workgroup walks
    for (uint64_t g = blockIdx.x; g < ngroups; g += gridDim.x) mine += (2 * g + 2 <= npairs) ? 2 : npairs - 2 * g;
    auto pair_of = [&](uint64_t i) { return 2 * (blockIdx.x + (i / 2) * gridDim.x) + (i & 1); };
    vs_d2 raw[4];
    bool k0 = true, k1 = true;
    auto fetch = [&](uint64_t pr) {
        const uint64_t s0 = pr * 2 * VS2_WCHUNK;
        VS2_FTILE(s0);
        const bool whole = s0 + 2 * VS2_WCHUNK <= rows;
        const uint64_t sm = s0 + 2 * piece;
        int lr = lrow;
        if (ONE) asm volatile("" : "+v"(lr));   // not loop-invariant as far as the compiler can tell
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = lr + 32 * (j & 1) < K;   // rows past K are zeros in the tile: nothing to load
            const double* q = ptr(j, lr, fb, fp);
            if (whole) {
                if (in) raw[j] = __builtin_nontemporal_load((const vs_d2*)q);
            } else {
                raw[j].x = (in && sm < rows) ? q[0] : 0.0;
                raw[j].y = (in && sm + 1 < rows) ? q[1] : 0.0;
            }
        }
        if (MASKED) {
            k0 = sm < rows && A.keep[sm];
            k1 = sm + 1 < rows && A.keep[sm + 1];
        }
    };
    auto stash = [&](int b) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            vs_d2 c = raw[j], d = raw[2 + j];
            if (MASKED) {
                c.x = k0 ? c.x : 0.0;
                c.y = k1 ? c.y : 0.0;
                d.x = k0 ? d.x : 0.0;
                d.y = k1 ? d.y : 0.0;
            }
            if (lrow + 32 * j >= K) c = d = (vs_d2){0.0, 0.0};
            *(vs_d2*)(wr + b * BUF + 32 * j * VS2_WPITCH) = c + d;
            *(vs_d2*)(wr + b * BUF + (VS2_HALF + 32 * j) * VS2_WPITCH) = c - d;
        }
    };

    if (mine > 0) {
        fetch(pair_of(0));
        stash(0);
        if (mine > 1) fetch(pair_of(1));
    }
    __syncthreads();
    for (uint64_t i = 0; i < mine; ++i) {
        const int b = (int)(i & 1);
        // the next pair goes into the other buffer (everybody left it at the last barrier) ...
        if (i + 1 < mine) stash(b ^ 1);
        if (i + 2 < mine) fetch(pair_of(i + 2));
        // ... while this one feeds the matrix pipe
        vs_d2 fa[2];
        fa[0] = *(const vs_d2*)(rdS + b * BUF + 16 * r * VS2_WPITCH);
        fa[1] = *(const vs_d2*)(rdS + b * BUF + 16 * r * VS2_WPITCH + 8);
#pragma unroll
        for (int sl = 0; sl < 5; ++sl) {
            const bool ss = sl < 4 - r;                 // wave-uniform
            if (sl == 4 - r) {                          // from here on: row block 3 - r of T T^T
                fa[0] = *(const vs_d2*)(rdT + b * BUF + 16 * tr * VS2_WPITCH);
                fa[1] = *(const vs_d2*)(rdT + b * BUF + 16 * tr * VS2_WPITCH + 8);
            }
            const int blk = ss ? r + sl : sl - 1;       // block of the other operand
            const bool live = (ss ? r : tr) < nbl && blk < nbl;
            if (live) {
                const double* rb = (ss ? rdS : rdT) + b * BUF + 16 * blk * VS2_WPITCH;
                const vs_d2 f0 = *(const vs_d2*)rb;
                const vs_d2 f1 = *(const vs_d2*)(rb + 8);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0][0], f0[0], acc[sl], 0, 0, 0);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0][1], f0[1], acc[sl], 0, 0, 0);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1][0], f1[0], acc[sl], 0, 0, 0);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[1][1], f1[1], acc[sl], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // a tile belongs to one wave: straight to the wave group's partial (all-zero and mirror tiles stay unwritten;
    // the fold never reads them)
    double* out = A.partials + ((size_t)blockIdx.x * 2 + grp) * 2 * VS2_HTILE_OUT;
#pragma unroll
    for (int sl = 0; sl < 5; ++sl) {
        const bool ss = sl < 4 - r;
        const int blk = ss ? r + sl : sl - 1;
        const int rowblk = ss ? r : tr;
        if (rowblk < nbl && blk < nbl) {
            double* o = out + (ss ? 0 : VS2_HTILE_OUT);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = (lane >> 4) + 4 * v, col = lane & 15;   // f64 C/D layout
                o[(rowblk * 16 + row) * VS2_HALF + (blk * 16 + col)] = acc[sl][v];
            }
        }
    }
}

// The five scalar sums of the first-order record (sab, sa, sb, shifted s1, s2: varsens/saltelli.py:577,583) for
// the one-pass route beyond 32 columns: a = fM_1, b = fM_2 only.  One full record per workgroup (zeros in the
// dot-product part), appended to the macro kernel's records.
#define VS2_HEAD_THREADS 1024
__global__ void __launch_bounds__(VS2_HEAD_THREADS)
first_head_kernel(const double* __restrict__ f, uint64_t rows, int k, int f_shift, double* __restrict__ records) {
    __shared__ double wsum[VS2_HEAD_THREADS / 64][5];
    const double shift = vs_shift_estimate(f, rows);   // wave-wide call; every wave and the fold get the same bits
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    auto add = [&](double a, double b) {
        const double dx = a - shift, dy = b - shift;
        v[VS_P_SAB] = fma(a, b, v[VS_P_SAB]);
        v[VS_P_SA] += a;
        v[VS_P_SB] += b;
        v[VS_P_S1] += dx + dy;
        v[VS_P_S2] = fma(dx, dx, v[VS_P_S2]);
        v[VS_P_S2] = fma(dy, dy, v[VS_P_S2]);
    };
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t t0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // row-blocked f-values (vs_f_tile), one output: a = fM_1 is plane 0, b = fM_2 plane 1 of the row's row block
    const uint64_t planes = 2 * (uint64_t)k + 2;
    if (rows % 2 == 0 && ((uintptr_t)f & 15) == 0) {   // 16-byte loads: two rows per thread and trip
        for (uint64_t r = 2 * t0; r < rows; r += 2 * stride) {
            const vs_ftile ft = vs_f_tile(rows, planes, f_shift, r);
            const vs_d2 a = __builtin_nontemporal_load((const vs_d2*)(f + ft.base));
            const vs_d2 b = __builtin_nontemporal_load((const vs_d2*)(f + ft.base + ft.pitch));
            add(a.x, b.x);
            add(a.y, b.y);
        }
    } else {
        for (uint64_t r = t0; r < rows; r += stride) {
            const vs_ftile ft = vs_f_tile(rows, planes, f_shift, r);
            add(__builtin_nontemporal_load(f + ft.base), __builtin_nontemporal_load(f + ft.base + ft.pitch));
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int z = 0; z < 5; ++z) {
        const double t = vs_wave_sum_dpp(v[z]);
        if (lane == 63) wsum[wave][z] = t;
    }
    __syncthreads();
    const size_t plen = VS_PLEN(k);
    double* rec = records + (size_t)blockIdx.x * plen;
    for (size_t e = threadIdx.x; e < plen; e += VS2_HEAD_THREADS) {
        double t = 0.0;
        if (e < VS_P_HEAD)
            for (int w = 0; w < VS2_HEAD_THREADS / 64; ++w) t += wsum[w][e];   // fixed order
        rec[e] = t;
    }
}

// NaN probe of the two blocks the Gram matrix never sees: fM_1 and fM_2 of output 0 (planes 0, 1).
// The reference drops row r from fN_j / fN_nj when ANY of the 2k+2 blocks holds a NaN at r
// (saltelli.py:474-492); a NaN in blocks 2..2k+1 poisons a Gram total and gram_fold_kernel raises the
// flag, a NaN in block 0 or 1 is caught here.  2/(2k+2) of the data: cheap beside the Gram pass.
__global__ void __launch_bounds__(256)
nan_probe_kernel(const double* __restrict__ f, uint64_t rows, uint64_t planes, int f_shift, uint32_t* counters) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        const vs_ftile ft = vs_f_tile(rows, planes, f_shift, r);   // the row's planes 0 and 1
        const double x = f[ft.base], y = f[ft.base + ft.pitch];
        bad |= (x != x) | (y != y);
    }
    if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(&counters[CNT_NANFLAG], 1u);
}

// keep mask for the second-order pass (same rule as the first-order one)
__global__ void __launch_bounds__(256)
nan_mask2_kernel(const double* __restrict__ f, uint64_t rows, int k, int l, int f_shift, uint8_t* __restrict__ keep,
                 const uint32_t* counters) {
    if (counters[CNT_NANFLAG] == 0) return;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t planes = (uint64_t)l * (2 * (uint64_t)k + 2);
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        bool bad = false;
        const vs_ftile ft = vs_f_tile(rows, planes, f_shift, r);
        for (int b = 0; b < 2 * k + 2; ++b) {
            double x = f[ft.base + (uint64_t)b * ft.pitch];  // planes of output 0
            bad |= (x != x);
        }
        keep[r] = bad ? 0 : 1;
    }
}

// sum the slab partials (compensated) and scatter the Gram entries into the three
// row-major (K,K) matrices DC = D C^T, DD = D D^T, CC = C C^T.
struct vs_gram_fold_args {
    const double* partials;
    uint32_t P;
    int K, nsuper, npairs;
    int super;   // index rows per block of the partials: 64 (gram_diag_wave_kernel) or 128 (gram_macro_kernel)
    int st;      // 1: the partials are sums over [S; T] (rows u and K + u): SS and TT go straight to `gram`;
                 // 0: sums over [C; D]: DC, DD, CC for gram_convert_kernel
    double* gram;
    uint32_t* counters;
    int masked;
    double* seg_sums;   // grid.y > 1: [grid.y][npairs * super^2] segment sums (two-stage fold)
};

#define VS2_FOLD_THREADS 1024
enum { CNT_GRAM_TICKET = 5 };   // zero between folds (the last block of a two-stage fold puts it back)

// block b folds 64 consecutive entries: 16 waves stride over the P partials (coalesced),
// then wave 0 combines them in fixed order and scatters.
// grid.y = segments (two-stage fold, as fold_first_kernel: thousands of partials behind the fused evaluate + both
// orders kernels): block (b, s) leaves the sums of segment s in seg_sums, the block that takes the last ticket adds
// them in segment order and scatters.  grid.y = 1: single stage.
struct gram_entry {
    bool live, mirror;
    int gi, gj;
};

__device__ __forceinline__ gram_entry gram_entry_of(const vs_gram_fold_args& A, size_t col) {
    const int SB = A.super;
    const size_t tile_out = (size_t)SB * SB;
    const int pair = (int)(col / tile_out);
    const int e = (int)(col % tile_out);
    int SI = 0, SJ = 0;
    if (A.st != 2) pair_from_index(pair, A.nsuper, SI, SJ);
    const int li = e / SB, lj = e % SB;
    const int K = A.K;
    gram_entry g;
    // st == 2 (gram_macro_st_kernel): tile 0 is S S^T, tile 1 T T^T, rows and columns u < K
    g.gi = A.st == 2 ? li + pair * K : SI * SB + li;
    g.gj = A.st == 2 ? lj + pair * K : SJ * SB + lj;
    // entries of mirror blocks (never computed) and of padding rows are skipped
    g.live = A.st == 2 ? (lj / 16) >= (li / 16) && li < K && lj < K
                       : !(SI == SJ && (lj / 16) < (li / 16)) && g.gi < 2 * K && g.gj < 2 * K &&
                             !(A.st && (g.gi < K) != (g.gj < K));   // S.T products are by-products
    g.mirror = !(SI == SJ && (li / 16) == (lj / 16));  // diagonal 16-blocks hold both triangles
    return g;
}

__device__ __forceinline__ void gram_scatter(const vs_gram_fold_args& A, const gram_entry& g, double total) {
    const int K = A.K, gi = g.gi, gj = g.gj;
    if (!A.masked && total != total) atomicOr(&A.counters[CNT_NANFLAG], 1u);
    if (A.st) {
        double* M = A.gram + (gi < K ? 0 : (size_t)K * K);   // SS, then TT
        const int i = gi < K ? gi : gi - K, j = gj < K ? gj : gj - K;
        M[(size_t)i * K + j] = total;
        if (g.mirror) M[(size_t)j * K + i] = total;
        return;
    }
    double* DC = A.gram;
    double* DD = DC + (size_t)K * K;
    double* CC = DD + (size_t)K * K;
    if (gi < K && gj < K) {
        CC[(size_t)gi * K + gj] = total;
        if (g.mirror) CC[(size_t)gj * K + gi] = total;
    } else if (gi >= K && gj >= K) {
        DD[(size_t)(gi - K) * K + (gj - K)] = total;
        if (g.mirror) DD[(size_t)(gj - K) * K + (gi - K)] = total;
    } else if (gi < K) {            // C_gi . D_gj
        DC[(size_t)(gj - K) * K + gi] = total;
    } else {                        // D_gi . C_gj
        DC[(size_t)(gi - K) * K + gj] = total;
    }
}

__global__ void __launch_bounds__(VS2_FOLD_THREADS)
gram_fold_kernel(vs_gram_fold_args A) {
    __shared__ double s_sum[VS2_FOLD_THREADS / 64][64];
    __shared__ double s_cmp[VS2_FOLD_THREADS / 64][64];
    __shared__ uint32_t s_last;
    if (A.masked && A.counters[CNT_NANFLAG] == 0) return;
    const size_t width = (size_t)A.npairs * A.super * A.super;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t col = (size_t)blockIdx.x * 64 + lane;   // width is a multiple of 64
    const gram_entry g = gram_entry_of(A, col);
    const uint32_t nseg = gridDim.y, per_seg = (A.P + nseg - 1) / nseg;
    const uint32_t p_lo = blockIdx.y * per_seg, p_hi = p_lo + per_seg < A.P ? p_lo + per_seg : A.P;
    double s = 0.0, comp = 0.0;
    if (g.live) {
        constexpr uint32_t STEP = VS2_FOLD_THREADS / 64, BATCH = 8;
        for (uint32_t p0 = p_lo + wave; p0 < p_hi; p0 += STEP * BATCH) {
            double x[BATCH];
#pragma unroll
            for (uint32_t i = 0; i < BATCH; ++i) {
                const uint32_t p = p0 + i * STEP;
                x[i] = p < p_hi ? A.partials[(size_t)p * width + col] : 0.0;
            }
#pragma unroll
            for (uint32_t i = 0; i < BATCH; ++i) {
                const double t = s + x[i];
                comp += (fabs(s) >= fabs(x[i])) ? (s - t) + x[i] : (x[i] - t) + s;
                s = t;
            }
        }
    }
    s_sum[wave][lane] = s;
    s_cmp[wave][lane] = comp;
    __syncthreads();
    if (wave == 0 && g.live) {
        double S = 0.0, Cc = 0.0;
        for (int w = 0; w < VS2_FOLD_THREADS / 64; ++w) {
            const double x = s_sum[w][lane];
            const double t = S + x;
            Cc += (fabs(S) >= fabs(x)) ? (S - t) + x : (x - t) + S;
            S = t;
            Cc += s_cmp[w][lane];
        }
        const double total = S + Cc;   // = X_gi . X_gj (nseg > 1: over this block's segment)
        if (nseg == 1) gram_scatter(A, g, total);
        else __hip_atomic_store(&A.seg_sums[(size_t)blockIdx.y * width + col], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (nseg == 1) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = atomicAdd(&A.counters[CNT_GRAM_TICKET], 1u);
        s_last = (t == gridDim.x * gridDim.y - 1) ? 1u : 0u;
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (!s_last) return;
    for (size_t c = threadIdx.x; c < width; c += VS2_FOLD_THREADS) {
        const gram_entry e = gram_entry_of(A, c);
        if (!e.live) continue;
        double S = 0.0, Cc = 0.0;
        for (uint32_t sg = 0; sg < nseg; ++sg) {
            const double x = __hip_atomic_load(&A.seg_sums[(size_t)sg * width + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double t = S + x;
            Cc += (fabs(S) >= fabs(x)) ? (S - t) + x : (x - t) + S;
            S = t;
        }
        gram_scatter(A, e, S + Cc);
    }
    if (threadIdx.x == 0) A.counters[CNT_GRAM_TICKET] = 0;   // ready for the next two-stage fold of this context
}

__global__ void __launch_bounds__(256)
finish_second_kernel(const double* __restrict__ gram, const double* __restrict__ first, uint64_t n, int k, int l,
                     double* __restrict__ s2, double* __restrict__ s2n) {
    const int K = k * l;
    const double* SS = gram;                     // (C + D)(C + D)^T
    const double* TT = SS + (size_t)K * K;       // (C - D)(C - D)^T
    const double* E_2 = first;
    const double* var_y = first + l;
    const double div = 2.0 * ((double)n - 1.0);
    const size_t total = (size_t)K * K;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t J = e % K;
        const int p = (int)(J % l);  // output index of the LAST axis: E_2 / var_y broadcast over it
        double a = 0.5 * (SS[e] - TT[e]);            // D C^T + C D^T, saltelli.py:612-613
        a /= div;                                    // :614
        a -= E_2[p];                                 // :615
        a /= var_y[p];                               // :616
        s2[e] = a;
        double b = 0.5 * (SS[e] + TT[e]);            // D D^T + C C^T, :618-619
        b /= div;
        b -= E_2[p];
        b /= var_y[p];
        s2n[e] = b;
    }
}

// Kernels that accumulate the Gram matrix of X = [C; D] (more than 128 index rows per half: see second_order_impl)
// leave DC = D C^T, DD, CC in the workspace; this turns them into the interface's SS and TT.
__global__ void __launch_bounds__(256)
gram_convert_kernel(const double* __restrict__ cd, int K, double* __restrict__ gram) {
    const double* DC = cd;
    const double* DD = DC + (size_t)K * K;
    const double* CC = DD + (size_t)K * K;
    const size_t total = (size_t)K * K;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t I = e / K, J = e % K;
        const double cross = DC[I * K + J] + DC[J * K + I];
        const double same = DD[e] + CC[e];
        gram[e] = same + cross;
        gram[total + e] = same - cross;
    }
}

// (MASKED, ALIGNED, FIRST, WIDE) instances of the [C; D] macro kernel the launches can pick
#define VS2_MACRO_INSTANCES(X) \
    X(false, false, false, false) X(false, true, false, false) X(true, false, false, false) X(true, true, false, false) \
    X(false, false, true, false) X(false, true, true, false) X(false, true, false, true) X(true, true, false, true)
// (MASKED, ALIGNED, WIDE) instances of the S/T macro kernel
#define VS2_ST_INSTANCES(X) \
    X(false, false, false) X(false, true, false) X(false, true, true) X(true, false, false) X(true, true, false) X(true, true, true)
// (NB, FIRST, FULL) instances of the wide wave-private kernel
#define VS2_WIDE_INSTANCES(X) X(3, false, false) X(4, false, false) X(4, false, true) X(3, true, false) X(4, true, true)

// Kernels that ask for more than 64 KiB of dynamic LDS need the limit raised; once per context (= per device).
static int gram_raise_lds_limits(vs_context* ctx) {
    if (ctx->gram_lds_raised) return VS_OK;
    const int lds_m = (int)(sizeof(double) * 2 * 2 * VS2_MACRO * VS2_WPITCH);   // macro kernels: two buffers of 256 tile rows
#define VS2_SET(...) VS_HIP(hipFuncSetAttribute((const void*)__VA_ARGS__, hipFuncAttributeMaxDynamicSharedMemorySize, lds_m));
#define VS2_SET_MACRO(MK, AL, FI, WI) VS2_SET(gram_macro_kernel<MK, AL, FI, WI>)
#define VS2_SET_ST(MK, AL, WI) VS2_SET(gram_macro_st_kernel<MK, AL, WI>)
    VS2_MACRO_INSTANCES(VS2_SET_MACRO)
    VS2_ST_INSTANCES(VS2_SET_ST)
    VS2_SET(gram_half_st_kernel<true, true>)
    VS2_SET(gram_half_st_kernel<true, false>)
#undef VS2_SET_ST
#undef VS2_SET_MACRO
#undef VS2_SET
#define VS2_SET_WIDE(NBV, FI, FU)                                                                                      \
    VS_HIP(hipFuncSetAttribute((const void*)gram_diag_wide_kernel<NBV, FI, FU>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)(sizeof(double) * VS2_WIDE_LDS(NBV, FI))));
    VS2_WIDE_INSTANCES(VS2_SET_WIDE)
#undef VS2_SET_WIDE
    int rc_st = vs_gram_st_prepare(ctx);
    if (rc_st) return rc_st;
    rc_st = vs_gram_dma_prepare(ctx);
    if (rc_st) return rc_st;
    ctx->gram_lds_raised = true;
    return VS_OK;
}

// with_first (macro kernel, one output, no NaN trimming): also this rank's first-order record into acc_dev
static int second_order_impl(vs_context* ctx, const double* f_dev, uint64_t rows, int k, int l, int flags,
                             double* gram_dev, bool with_first, int rank, int max_ranks, double* acc_dev) {
    VS_REQUIRE(ctx && gram_dev, "null argument");
    VS_HIP(hipSetDevice(ctx->device));
    VS_REQUIRE(k >= 1 && l >= 1, "k and l must be positive");
    VS_REQUIRE(rows == 0 || f_dev, "f_dev is null");
    VS_REQUIRE((uint64_t)k * l <= 4096, "k*l=%llu too large for the second-order pass (limit 4096)",
               (unsigned long long)k * l);
    const int K = k * l;
    // up to 64 index rows: one 64-row block, wave-private kernel; beyond: 128-row macro blocks
    const bool macro = 2 * K > VS2_SUPER;
    // 33 .. 128 index rows per half: the S/T macro kernel (one tile of 128 s rows + 128 t rows per workgroup, two
    // 128 x 128 partials: "2 pairs" below, one grid row); beyond: the [C; D] macro kernel over pairs of 128-row blocks
    // (49 .. 64 rows per half stay on the [C; D] kernel, measured: four live 16-row blocks leave half of the S/T
    // kernel's waves one or two tiles per barrier -- 1.49 vs 1.62-1.67 ms at n = 2^22)
    // 33 .. 64 rows per half, even row count: two chunks per tile (gram_half_st_kernel; with one output it also carries
    // the first-order dot products); odd row counts: 33 .. 48 the S/T kernel above, 49 .. 64 the [C; D] kernel
    const bool half_st = macro && K <= VS2_HALF && rows % 2 == 0 && ((uintptr_t)f_dev & 15) == 0 &&
                         ctx->tune.gram_wide != 0 && ctx->tune.gram_st != 0;
    // 33 .. 128 rows per half, one output, even row count: gram_st_kernel (vs_gram_st.hip) takes the unmasked pass
    // (one workgroup per CU; with_first: it carries the first-order dot products); the masked pass and the other
    // shapes of this range stay on gram_macro_st_kernel / gram_half_st_kernel
    const bool st128 = macro && K > VS2_HALF && K <= VS2_MACRO && rows % 2 == 0 && ((uintptr_t)f_dev & 15) == 0 &&
                       ctx->tune.gram_wide != 0 && ctx->tune.gram_st != 0;
    const bool macro_st = macro && !half_st && K <= VS2_MACRO && !(K > 48 && K <= 64) && (!with_first || st128) &&
                          ctx->tune.gram_st != 0;
    const int SB = half_st ? VS2_HALF : macro ? VS2_MACRO : VS2_SUPER;
    const int nsuper = (macro_st || half_st) ? 1 : (2 * K + SB - 1) / SB;
    const int npairs = (macro_st || half_st) ? 2 : nsuper * (nsuper + 1) / 2;
    const int grid_rows = (macro_st || half_st) ? 1 : npairs;
    uint64_t nchunks = (rows + VS2_WCHUNK - 1) / VS2_WCHUNK;
    if (nchunks < 1) nchunks = 1;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    // 16-byte loads need an even row count: planar blocks start at multiples of `rows` doubles
    const bool aligned = rows % 2 == 0 && ((uintptr_t)f_dev & 15) == 0;
    // workgroups per CU: 3 wave-private ones (measured, tools/gram_only.py); macro ones: 2 are resident (LDS:
    // 80 KiB each), 6 are launched -- the diagonal pairs finish early and the dispatcher evens it out (measured)
    const int nb_wave = (2 * K + 15) / 16;   // live 16-row blocks of the wave-private kernel (2K <= 64)
    const int per_cu = ctx->tune.gram_wgs_per_cu > 0 ? ctx->tune.gram_wgs_per_cu
                       : ((macro_st || half_st) ? 2 : macro ? 6 : (nb_wave <= 4 ? VS2_WAVES_PER_SIMD(nb_wave, false) : 3));
    // grid = want x npairs workgroups: never more than are resident at once (a single workgroup left over
    // would run a whole extra round on an empty chip)
    uint64_t want = macro ? ((uint64_t)cus * per_cu) / grid_rows : ((uint64_t)cus * per_cu + npairs - 1) / npairs;
    if (want < 1) want = 1;
    const uint64_t units = macro ? (nchunks + 3) / 4 : nchunks;   // the macro kernel walks groups of 4 chunks
    const uint32_t slab_wgs = (uint32_t)(units < want ? units : want);
    const bool st_new = st128 || half_st;   // the unmasked pass runs on gram_st_kernel (any number of outputs)
    // ... or, second order alone with one output and 33 .. 64 index rows, on gram_dma_kernel (vs_gram_dma.hip)
    // (one output: with the first-order dot products on board where they fit; several outputs: second order alone)
    const bool dma = half_st && ctx->tune.gram_dma != 0 && (!with_first || (l == 1 && vs_gram_dma_takes_first(K)));
    // ... and, as a measurement knob only (VS_GRAM_DMA=2), 17 .. 32 index rows (sixteen waves per CU on images of 32 rows
    // per half; partials [S S^T, T T^T][32][32]): HBM-bound there and level with gram_diag_wide_kernel -- k = 32, n = 2^22:
    // 0.338-0.348 against 0.332-0.398 ms, k = 24: 0.275 against 0.255 (profiles/r05/gram_dma_ab.txt) -- so not the default
    const bool dma32 = !macro && K > 16 && l == 1 && aligned && !with_first && ctx->tune.gram_dma == 2;
    const uint32_t dma32_wgs = (uint32_t)(nchunks < (uint64_t)cus ? nchunks : (uint64_t)cus);
    const uint32_t st_wgs = (uint32_t)(units < (uint64_t)cus ? units : (uint64_t)cus);   // one per CU (160 KiB of LDS)
    // the wide wave-private kernel (32 sample rows per unit, two workgroups per CU) takes the unmasked pass of
    // one-output problems with 33..64 index rows
    const bool wide = !macro && l == 1 && aligned && nb_wave >= 3 && ctx->tune.gram_wide != 0;
    const uint64_t wide_units = (rows + VS2_XCHUNK - 1) / VS2_XCHUNK;
    const uint64_t wide_want = (uint64_t)cus * 2;   // all that are resident (LDS)
    const uint32_t wide_wgs = (uint32_t)(wide_units < 1 ? 1 : wide_units < wide_want ? wide_units : wide_want);
    const size_t tile_out = (size_t)SB * SB;
    const size_t width = (size_t)npairs * tile_out;
    // workspace: [with_first: first-order records of the diagonal workgroups + folded record,] then the Gram partials
    const size_t plen = VS_PLEN(k);
    const uint32_t pslots_old = half_st ? 2 * slab_wgs : slab_wgs;   // partials per pair: the older kernels (masked pass)
    const uint32_t pslots_new = half_st ? 2 * st_wgs : st_wgs;        //   gram_st_kernel
    const uint32_t pslots = (dma && with_first) ? st_wgs : (st_new && with_first) ? pslots_new : pslots_old;   // first-order records per pair
    const uint32_t P_diag = with_first ? pslots * (uint32_t)nsuper : 0;                         // macro kernels' records
    const uint32_t P_head = with_first ? (uint32_t)((rows + 4095) / 4096 < (uint64_t)cus ? (rows + 4095) / 4096 : cus) : 0;
    const uint32_t P_first = P_diag + P_head;                                                   // + first_head_kernel's
    size_t first_doubles = with_first ? (size_t)P_first * plen + plen : 0;
    first_doubles = (first_doubles + 1) & ~(size_t)1;
    const uint32_t max_wgs = wide && wide_wgs > pslots_old ? wide_wgs : pslots_old;
    // ... and DC, DD, CC (3 K^2) behind them, for gram_convert_kernel
    const size_t cd_doubles = 3 * (size_t)K * K;
    int rc = vs_workspace_partials(ctx, sizeof(double) * (first_doubles + (size_t)max_wgs * width + cd_doubles));
    if (rc) return rc;
    double* cd_dev = ctx->partials + first_doubles + (size_t)max_wgs * width;

    VS_HIP(hipMemsetAsync(ctx->counters + CNT_NANFLAG, 0, sizeof(uint32_t), ctx->stream));
    VS_HIP(hipMemsetAsync(cd_dev, 0, sizeof(double) * cd_doubles, ctx->stream));

    vs_gram_args A;
    A.f = f_dev;
    A.f_shift = ctx->tune.f_shift;
    A.rows = rows;
    A.k = k;
    A.l = l;
    A.K = K;
    A.nsuper = nsuper;
    A.slab_wgs = slab_wgs;
    A.partials = ctx->partials + first_doubles;
    A.first_partials = with_first ? ctx->partials : nullptr;
    A.keep = nullptr;
    A.counters = ctx->counters;
    vs_gram_fold_args F;
    F.partials = A.partials;
    F.P = slab_wgs;
    F.K = K;
    F.nsuper = nsuper;
    F.npairs = npairs;
    F.super = SB;
    const bool st = !macro || macro_st || half_st;   // the kernels accumulate [S; T] sums (else [C; D] sums, converted at the end)
    F.st = (macro_st || half_st) ? 2 : st ? 1 : 0;
    F.gram = st ? gram_dev : cd_dev;
    if (st) VS_HIP(hipMemsetAsync(gram_dev, 0, sizeof(double) * VS_GRAM_LEN(K), ctx->stream));
    F.counters = ctx->counters;
    F.seg_sums = nullptr;
    const size_t lds_w = sizeof(double) * (nb_wave <= 4 ? (size_t)VS2_WAVE_LDS(nb_wave, false) : 0);   // wave-private tiles
    const size_t lds_m = sizeof(double) * 2 * 2 * VS2_MACRO * VS2_WPITCH;   // macro: two buffers of 256 rows
    const bool mwide = macro && l == 1 && aligned && ctx->tune.gram_wide != 0;   // two chunks per fetch, 256-byte runs
    rc = gram_raise_lds_limits(ctx);
    if (rc) return rc;
    VS_REQUIRE(!(with_first && macro_st) || st128, "no one-pass kernel for this shape");
    if ((flags & VS_REDUCE_NAN_TRIM) && rows > 0) {
        const uint64_t want_blocks = (rows + 255) / 256;   // a thread per row: blocks 0 and 1 of output 0
        vs_launch_scope scope(ctx, VS_KERNEL_NAN_MASK);
        hipLaunchKernelGGL(nan_probe_kernel, dim3((unsigned)(want_blocks < 2048 ? want_blocks : 2048)), dim3(256), 0,
                           ctx->stream, f_dev, rows, (uint64_t)l * (2 * (uint64_t)k + 2), ctx->tune.f_shift, ctx->counters);
        VS_HIP(hipGetLastError());
    }
    for (int pass = 0; pass < 2; ++pass) {
        const bool masked = pass == 1;
        if (masked) {
            if (!(flags & VS_REDUCE_NAN_TRIM) || rows == 0) break;
            rc = vs_workspace_keep(ctx, rows);
            if (rc) return rc;
            A.keep = ctx->keep;
            vs_launch_scope scope(ctx, VS_KERNEL_NAN_MASK);
            hipLaunchKernelGGL(nan_mask2_kernel, dim3((unsigned)((rows + 255) / 256 < 4096 ? (rows + 255) / 256 : 4096)),
                               dim3(256), 0, ctx->stream, f_dev, rows, k, l, ctx->tune.f_shift, ctx->keep, (const uint32_t*)ctx->counters);
        }
        {
            vs_launch_scope scope(ctx, VS_KERNEL_SECOND_ORDER);
            if (half_st && st_new && !masked && dma) {
                A.slab_wgs = F.P = st_wgs;   // one partial per workgroup
                rc = vs_gram_dma_launch(ctx, A, with_first, st_wgs);
                if (rc) return rc;
            } else if (half_st && st_new && !masked) {
                A.slab_wgs = st_wgs;
                F.P = pslots_new;
                rc = vs_gram_st_launch(ctx, A, with_first, st_wgs);
                if (rc) return rc;
            } else if (half_st) {
                A.slab_wgs = slab_wgs;
                F.P = pslots_old;
                const dim3 g(slab_wgs), t(512);
                // (the unmasked pass runs on gram_st_kernel, above: only the masked one comes here)
                if (l == 1) hipLaunchKernelGGL((gram_half_st_kernel<true, true>), g, t, lds_m, ctx->stream, A);
                else hipLaunchKernelGGL((gram_half_st_kernel<true, false>), g, t, lds_m, ctx->stream, A);
            } else if (macro_st && st128 && !masked) {
                A.slab_wgs = F.P = st_wgs;
                rc = vs_gram_st_launch(ctx, A, with_first, st_wgs);
                if (rc) return rc;
            } else if (macro_st) {
                A.slab_wgs = F.P = slab_wgs;
                const dim3 grid(slab_wgs, 1);
                if (masked && mwide) hipLaunchKernelGGL((gram_macro_st_kernel<true, true, true>), grid, dim3(512), lds_m, ctx->stream, A);
                else if (masked && aligned) hipLaunchKernelGGL((gram_macro_st_kernel<true, true, false>), grid, dim3(512), lds_m, ctx->stream, A);
                else if (masked) hipLaunchKernelGGL((gram_macro_st_kernel<true, false, false>), grid, dim3(512), lds_m, ctx->stream, A);
                else if (mwide) hipLaunchKernelGGL((gram_macro_st_kernel<false, true, true>), grid, dim3(512), lds_m, ctx->stream, A);
                else if (aligned) hipLaunchKernelGGL((gram_macro_st_kernel<false, true, false>), grid, dim3(512), lds_m, ctx->stream, A);
                else hipLaunchKernelGGL((gram_macro_st_kernel<false, false, false>), grid, dim3(512), lds_m, ctx->stream, A);
            } else if (macro) {
                const dim3 grid(slab_wgs, npairs);
                bool launched = false;
                // With the first-order sums on board the kernel takes the 16-sample-row fetch: the WIDE instance of that
                // body sat at the kernel's 128 registers with three dwords in scratch (rounds 2-4), and the row-blocked
                // layout's per-fetch base and pitch pushed a whole staging quadruple out.  Beyond 128 index rows the
                // kernel is matrix-pipe-bound (busy 0.71 at k = 197, profiles/r05), not fetch-bound: same time.
                const bool mwide_now = mwide && !with_first;
#define VS2_LAUNCH_MACRO(MK, AL, FI, WI)                                                                        \
                if (!launched && masked == MK && aligned == AL && with_first == FI && mwide_now == WI) {        \
                    hipLaunchKernelGGL((gram_macro_kernel<MK, AL, FI, WI>), grid, dim3(512), lds_m, ctx->stream, A); \
                    launched = true;                                                                            \
                }
                VS2_MACRO_INSTANCES(VS2_LAUNCH_MACRO)
#undef VS2_LAUNCH_MACRO
                if (!launched) return vs_fail(VS_ERR_STATE, "no Gram kernel for this case");   // with_first is never masked
            } else if (dma32 && !masked) {
                A.slab_wgs = F.P = dma32_wgs;
                rc = vs_gram_dma_launch(ctx, A, false, dma32_wgs);
                if (rc) return rc;
            } else if (wide && !masked) {
                const size_t lds_x = sizeof(double) * (size_t)(nb_wave == 3 ? VS2_WIDE_LDS(3, false) : VS2_WIDE_LDS(4, false));
                A.slab_wgs = F.P = wide_wgs;
#define VS2_LAUNCH_WIDE(NBV, FU) \
    hipLaunchKernelGGL((gram_diag_wide_kernel<NBV, false, FU>), dim3(wide_wgs), dim3(256), lds_x, ctx->stream, A)
                if (nb_wave == 3) VS2_LAUNCH_WIDE(3, false);
                else if (K == 32) VS2_LAUNCH_WIDE(4, true);
                else VS2_LAUNCH_WIDE(4, false);
#undef VS2_LAUNCH_WIDE
            } else {
                A.slab_wgs = F.P = slab_wgs;
                const dim3 grid_diag(slab_wgs, 1);
                const int nb = (2 * K + 15) / 16;
#define VS2_LAUNCH_WAVE2(NBV, MK, AL, FU) \
    hipLaunchKernelGGL((gram_diag_wave_kernel<NBV, MK, AL, false, FU>), grid_diag, dim3(256), lds_w, ctx->stream, A)
#define VS2_LAUNCH_WAVE(NBV, FU)                                        \
    do {                                                                \
        if (masked && aligned) VS2_LAUNCH_WAVE2(NBV, true, true, FU);   \
        else if (masked) VS2_LAUNCH_WAVE2(NBV, true, false, FU);        \
        else if (aligned) VS2_LAUNCH_WAVE2(NBV, false, true, FU);       \
        else VS2_LAUNCH_WAVE2(NBV, false, false, FU);                   \
    } while (0)
                switch (nb) {
                    case 1: VS2_LAUNCH_WAVE(1, false); break;
                    case 2: if (K == 16) VS2_LAUNCH_WAVE(2, true); else VS2_LAUNCH_WAVE(2, false); break;
                    case 3: VS2_LAUNCH_WAVE(3, false); break;
                    default: if (K == 32) VS2_LAUNCH_WAVE(4, true); else VS2_LAUNCH_WAVE(4, false); break;
                }
#undef VS2_LAUNCH_WAVE
#undef VS2_LAUNCH_WAVE2
            }
        }
        VS_HIP(hipGetLastError());
        if (with_first && !masked) {
            {
                vs_launch_scope scope(ctx, VS_KERNEL_REDUCE_FIRST);
                hipLaunchKernelGGL(first_head_kernel, dim3(P_head), dim3(VS2_HEAD_THREADS), 0, ctx->stream, f_dev, rows, k, ctx->tune.f_shift,
                                   ctx->partials + (size_t)P_diag * plen);
            }
            VS_HIP(hipGetLastError());
            rc = vs_fold_partials(ctx, P_first, k, 1, rows, nullptr, 1, f_dev, rank, max_ranks, acc_dev);
            if (rc) return rc;
        }
        F.masked = masked ? 1 : 0;
        {
            vs_launch_scope scope(ctx, VS_KERNEL_SECOND_FOLD);
            vs_gram_fold_args Fp = F;
            size_t fold_width = width;
            if (dma32 && !masked) {   // gram_dma_kernel's partials at 17 .. 32 index rows: two tiles of 32 x 32
                Fp.st = 2;
                Fp.super = 32;
                Fp.nsuper = 1;
                Fp.npairs = 2;
                fold_width = 2 * 32 * 32;
            }
            hipLaunchKernelGGL(gram_fold_kernel, dim3((unsigned)(fold_width / 64)), dim3(VS2_FOLD_THREADS), 0, ctx->stream, Fp);
        }
        VS_HIP(hipGetLastError());
    }
    if (!st) {
        vs_launch_scope scope(ctx, VS_KERNEL_SECOND_FOLD);
        const size_t total = (size_t)K * K;
        hipLaunchKernelGGL(gram_convert_kernel, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)),
                           dim3(256), 0, ctx->stream, cd_dev, K, gram_dev);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
}

// Fold of P partials [S S^T, T T^T][SB][SB] (upper 16-blocks; rows and columns below K) into gram_dev: the layout of
// gram_st_kernel and of the fused evaluate + both-orders kernels (vs_scan_gram.hip), whose caller this is.
int vs_gram_fold_st(vs_context* ctx, const double* partials, uint32_t P, int K, int SB, double* gram_dev) {
    VS_REQUIRE(K >= 1 && K <= SB && (2 * SB * SB) % 64 == 0, "bad fold shape K=%d SB=%d", K, SB);
    VS_HIP(hipMemsetAsync(gram_dev, 0, sizeof(double) * VS_GRAM_LEN(K), ctx->stream));
    vs_gram_fold_args F;
    F.partials = partials;
    F.P = P;
    F.K = K;
    F.nsuper = 1;
    F.npairs = 2;
    F.super = SB;
    F.st = 2;
    F.gram = gram_dev;
    F.counters = ctx->counters;
    F.masked = 0;
    const uint32_t nseg = vs_fold_segments(P);
    F.seg_sums = nullptr;
    if (nseg > 1) {   // stream-ordered behind the first-order fold, which is done with the same scratch
        int rc = vs_workspace_fold(ctx, sizeof(double) * nseg * 2 * SB * SB);
        if (rc) return rc;
        F.seg_sums = ctx->fold_scratch;
    }
    {
        vs_launch_scope scope(ctx, VS_KERNEL_SECOND_FOLD);
        hipLaunchKernelGGL(gram_fold_kernel, dim3((unsigned)(2 * SB * SB / 64), nseg), dim3(VS2_FOLD_THREADS), 0, ctx->stream, F);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
}

extern "C" int vs_reduce_second_order(vs_context* ctx, const double* f_dev, uint64_t rows, int k, int l,
                                      int flags, double* gram_dev) {
    VS_GUARD_BEGIN
    return second_order_impl(ctx, f_dev, rows, k, l, flags, gram_dev, false, 0, 1, nullptr);
    VS_GUARD_END
}

extern "C" int vs_reduce_first_second_order(vs_context* ctx, const double* f_dev, uint64_t rows, int k, int l,
                                            int flags, int rank, int max_ranks, double* acc_dev,
                                            double* gram_dev) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && acc_dev && gram_dev, "null argument");
    VS_HIP(hipSetDevice(ctx->device));
    VS_REQUIRE(k >= 1 && l >= 1, "k and l must be positive");
    VS_REQUIRE(max_ranks >= 1 && rank >= 0 && rank < max_ranks, "rank %d not in [0,%d)", rank, max_ranks);
    VS_REQUIRE(rows <= 0xfffffff0ull, "rows per GPU must stay below 2^32");
    VS_REQUIRE(rows == 0 || f_dev, "f_dev is null");
    const bool one_pass = l == 1 && !(flags & VS_REDUCE_NAN_TRIM) && rows >= 1 && ctx->tune.reduce_both != 0;
    // the first-order dot products ride on gram_half_st_kernel (33 .. 64 columns, even row count) and, where the [C; D]
    // macro kernel runs (beyond 128 columns; 49 .. 64 with an odd row count), on its diagonal pairs; the other shapes
    // up to 128 columns take the first-order kernel and the S/T macro kernel one after the other (4.8 + 13.9 ms at
    // k = 128, n = 2^24: the one-pass [C; D] form took 22.4; riding on the nine-tile S/T kernel spills in its loop)
    const bool even = rows % 2 == 0 && ((uintptr_t)f_dev & 15) == 0;
    const bool half_route = 2 * k > VS2_SUPER && k <= VS2_HALF && even && ctx->tune.gram_wide != 0 && ctx->tune.gram_st != 0;
    // 65 .. 128 columns, even row count: gram_st_kernel carries them too (two waves per SIMD, 256 VGPRs)
    const bool st128_route = k > VS2_HALF && k <= VS2_MACRO && even && ctx->tune.gram_wide != 0 && ctx->tune.gram_st != 0;
    if (one_pass && (k > VS2_MACRO || half_route || st128_route || (k > 48 && k <= 64) ||
                     (2 * k > VS2_SUPER && ctx->tune.gram_st == 0)) &&
        (uint64_t)k <= 4096)
        return second_order_impl(ctx, f_dev, rows, k, 1, 0, gram_dev, true, rank, max_ranks, acc_dev);
    const bool combined = one_pass && 2 * k <= VS2_SUPER;
    if (!combined) {
        int rc = vs_reduce_first_order(ctx, f_dev, rows, k, l, flags, rank, max_ranks, acc_dev);
        if (rc) return rc;
        return vs_reduce_second_order(ctx, f_dev, rows, k, l, flags, gram_dev);
    }
    const int K = k;
    const uint64_t nchunks = (rows + VS2_CHUNK - 1) / VS2_CHUNK;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    const int nb = (2 * K + 15) / 16;
    const uint64_t want = (uint64_t)cus * (ctx->tune.gram_wgs_per_cu > 0 ? ctx->tune.gram_wgs_per_cu
                                                                          : VS2_WAVES_PER_SIMD(nb, true));
    const bool aligned = rows % 2 == 0 && ((uintptr_t)f_dev & 15) == 0;
    // gram_diag_wide_kernel, two workgroups per CU; 25 .. 31 columns stay on the 16-row kernel: with four blocks and a
    // tile set known only at run time the wide form spills beside the first-order sums (0.84 vs 0.40 ms at k = 28)
    const bool wide = aligned && nb >= 3 && !(nb == 4 && K != 32) && ctx->tune.gram_wide != 0;
    const uint64_t wide_units = (rows + VS2_XCHUNK - 1) / VS2_XCHUNK;
    const uint64_t wide_want = (uint64_t)cus * 2;   // all that are resident (LDS)
    const uint32_t P = wide ? (uint32_t)(wide_units < wide_want ? wide_units : wide_want)
                            : (uint32_t)(nchunks < want ? nchunks : want);
    const size_t plen = VS_PLEN(k);
    // the wide kernel with three blocks leaves the five scalar sums of the record to first_head_kernel (P_head more records)
    const bool head_in_kernel = wide && nb == 4 && K == 32;   // gram_diag_wide_kernel<4, true, FULL>
    const uint32_t P_head = wide && !head_in_kernel ? (uint32_t)((rows + 4095) / 4096 < (uint64_t)cus ? (rows + 4095) / 4096 : cus) : 0;
    // workspace: first-order records [P + P_head][plen] + folded [plen] (vs_fold_partials' layout), then Gram partials
    size_t first_doubles = (size_t)(P + P_head) * plen + plen;
    first_doubles = (first_doubles + 1) & ~(size_t)1;
    int rc = vs_workspace_partials(ctx, sizeof(double) * (first_doubles + (size_t)P * VS2_TILE_OUT));
    if (rc) return rc;
    rc = gram_raise_lds_limits(ctx);
    if (rc) return rc;
    VS_HIP(hipMemsetAsync(gram_dev, 0, sizeof(double) * VS_GRAM_LEN(K), ctx->stream));
    vs_gram_args A;
    A.f = f_dev;
    A.f_shift = ctx->tune.f_shift;
    A.rows = rows;
    A.k = k;
    A.l = 1;
    A.K = K;
    A.nsuper = 1;
    A.slab_wgs = P;
    A.partials = ctx->partials + first_doubles;
    A.first_partials = ctx->partials;
    A.keep = nullptr;
    A.counters = ctx->counters;
    const size_t lds_w = sizeof(double) * (size_t)VS2_WAVE_LDS(nb, true);
    if (wide) {
        const size_t lds_x = sizeof(double) * (size_t)(nb == 3 ? VS2_WIDE_LDS(3, true) : VS2_WIDE_LDS(4, true));
        vs_launch_scope scope(ctx, VS_KERNEL_SECOND_ORDER);
#define VS2_LAUNCH_WIDE(NBV, FU) \
    hipLaunchKernelGGL((gram_diag_wide_kernel<NBV, true, FU>), dim3(P), dim3(256), lds_x, ctx->stream, A)
        if (nb == 3) VS2_LAUNCH_WIDE(3, false);
        else VS2_LAUNCH_WIDE(4, true);
#undef VS2_LAUNCH_WIDE
    } else {
        vs_launch_scope scope(ctx, VS_KERNEL_SECOND_ORDER);
#define VS2_LAUNCH_BOTH(NBV, FU)                                                                                          \
    do {                                                                                                                  \
        if (aligned) hipLaunchKernelGGL((gram_diag_wave_kernel<NBV, false, true, true, FU>), dim3(P), dim3(256), lds_w,   \
                                        ctx->stream, A);                                                                  \
        else hipLaunchKernelGGL((gram_diag_wave_kernel<NBV, false, false, true, FU>), dim3(P), dim3(256), lds_w,          \
                                ctx->stream, A);                                                                          \
    } while (0)
        switch (nb) {
            case 1: VS2_LAUNCH_BOTH(1, false); break;
            case 2: if (K == 16) VS2_LAUNCH_BOTH(2, true); else VS2_LAUNCH_BOTH(2, false); break;
            case 3: VS2_LAUNCH_BOTH(3, false); break;
            default: if (K == 32) VS2_LAUNCH_BOTH(4, true); else VS2_LAUNCH_BOTH(4, false); break;
        }
#undef VS2_LAUNCH_BOTH
    }
    VS_HIP(hipGetLastError());
    if (P_head) {
        vs_launch_scope scope(ctx, VS_KERNEL_REDUCE_FIRST);
        hipLaunchKernelGGL(first_head_kernel, dim3(P_head), dim3(VS2_HEAD_THREADS), 0, ctx->stream, f_dev, rows, k, ctx->tune.f_shift,
                           ctx->partials + (size_t)P * plen);
        VS_HIP(hipGetLastError());
    }
    rc = vs_fold_partials(ctx, P + P_head, k, 1, rows, nullptr, 1, f_dev, rank, max_ranks, acc_dev);
    if (rc) return rc;
    vs_gram_fold_args F;
    F.partials = A.partials;
    F.P = P;
    F.K = K;
    F.nsuper = 1;
    F.npairs = 1;
    F.super = VS2_SUPER;
    F.st = 1;
    F.gram = gram_dev;
    F.counters = ctx->counters;
    F.masked = 0;
    F.seg_sums = nullptr;
    {
        vs_launch_scope scope(ctx, VS_KERNEL_SECOND_FOLD);
        hipLaunchKernelGGL(gram_fold_kernel, dim3((unsigned)(VS2_TILE_OUT / 64)), dim3(VS2_FOLD_THREADS), 0, ctx->stream, F);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_finish_second_order(vs_context* ctx, const double* gram_dev, const double* first_dev,
                                      uint64_t n, int k, int l, double* sens2_dev, double* sens2n_dev) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && gram_dev && first_dev && sens2_dev && sens2n_dev, "null argument");
    VS_HIP(hipSetDevice(ctx->device));
    VS_REQUIRE(k >= 1 && l >= 1 && n >= 2, "bad dimensions");
    const size_t total = (size_t)k * l * k * l;
    {
        vs_launch_scope scope(ctx, VS_KERNEL_FINISH);
        hipLaunchKernelGGL(finish_second_kernel, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)),
                           dim3(256), 0, ctx->stream, gram_dev, first_dev, n, k, l, sens2_dev, sens2n_dev);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
    VS_GUARD_END
}
