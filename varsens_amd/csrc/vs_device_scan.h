// Segmented-scan form of the separable objectives: O(k) folds per design row instead of the
// O(k^2) of the chain scheme (vs_device_gen.h).  Device code shared by the precompiled kernels
// (vs_scan.hip) and by the kernels hiprtc builds for user separable objectives (vs_rtc.hip embeds
// this file as text after vs_device_gen.h).
//
// Reference counterpart: the four evaluation loops of Objective.__init__
// (varsens/saltelli.py:329-355) for an objective f(x) = t_0(x_0) op t_1(x_1) op ... op t_{k-1}(x_{k-1}).
//
// With ta_c = t_c(M_1[r][c]) and tb_c = t_c(M_2[r][c]) the 2k+2 values of design row r are
//   a   = ta_0 op ... op ta_{k-1}                     f(M_1[r])
//   b   = tb_0 op ... op tb_{k-1}                     f(M_2[r])
//   c_i = tb_0 .. tb_{i-1} op ta_i op tb_{i+1} ..     f(N_j[i][r])   (M_2 with column i from M_1)
//   d_i = ta_0 .. ta_{i-1} op tb_i op ta_{i+1} ..     f(N_nj[i][r])
// i.e. c_i = PB_i op ta_i op SB_{i+1} with PB the prefix and SB the suffix folds of the tb's.  A
// workgroup of W waves owns 64 design rows (row = lane) and cuts the k columns into W contiguous
// segments of L columns; wave q generates the coordinates and terms of its segment into REGISTERS
// (no LDS tile: the only shared data are the 2W segment totals per row), then
//   1. folds its segment totals TA_q, TB_q (left to right) and publishes them in LDS;
//   2. after ONE workgroup barrier folds the totals of the segments before it (left to right,
//      from the identity) into PA_q, PB_q and of the segments after it (right to left) into
//      SA_q, SB_q -- and all W totals into a and b (every wave, same order, same bits);
//   3. walks its segment backwards for the in-segment suffixes, forwards for the prefixes and
//      emits c_i = (pb op ta_j) op sb_j, d_i = (pa op tb_j) op sa_j.
// The association order is fixed by (L, W) alone and restated on the CPU by the test-side checker
// (evaluate_scan), so parity with it is bit-exact; against the
// strict left-to-right fold the values differ by reassociation only (<= ~k/2 ulp, typically 1-2).
// Serial depth per wave: L + W dependent operations instead of k, in four independent chains.
#ifndef VS_DEVICE_SCAN_H
#define VS_DEVICE_SCAN_H

// The unit-cube pair (coordinate c of the sequence rows behind M_1[r] and M_2[r]) for the scan kernels,
// whose column loop is unrolled L times: only the compact generator paths are inlined (Halton below
// 2^32 -- the host sends designs whose indices reach 2^32 to the chain kernel --, Sobol byte tables,
// raw points), and pow() of the power scaling sits behind one out-of-line copy.
// Sobol points of the 64 CONSECUTIVE indices base .. base + 63 of a block's M_1 rows (base a multiple of 64: the lane
// index is the low six bits).  Bit j of the Gray code is i_j ^ i_{j+1}: from bit 6 on it is the block's alone, bit 5 is
// lane bit 5 ^ bit 6 of the base, bits 0 .. 4 are the lane's.  So the three upper bytes of the Gray code are wave-uniform --
// their table entries come through the scalar cache -- and only byte 0 is looked up per lane: one gather instead of
// three or four (round 4: the table gathers are what a Sobol design costs beyond 32 columns, 512 KB of tables at k = 128).
struct vs_sobol_block {
    bool fast;     // wave-uniform: the block's M_1 indices are base + lane
    vs_u32 gb;     // Gray code of the base (bits 6 .. 31 are those of every index of the block)
    vs_u32 b0;     // this lane's byte 0 of the Gray code
};

__device__ __forceinline__ vs_sobol_block vs_sobol_block_of(const vs_design_dev& D, vs_u64 r, int lane) {
    vs_sobol_block S;
    // lane 0 of a block is always a row of the design (rows are clamped from the block's end): its index is the base
    const vs_u32 base = (vs_u32)__builtin_amdgcn_readfirstlane((int)(vs_u32)(D.first_index + r));
    S.fast = D.generator == VS_GEN_SOBOL && (base & 63u) == 0;
    S.gb = base ^ (base >> 1);
    S.b0 = ((((vs_u32)lane ^ ((vs_u32)lane >> 1)) ^ ((base >> 1) & 32u)) & 63u) | (S.gb & 192u);
    return S;
}

__device__ __forceinline__ vs_u32 vs_sobol_bits_block(const vs_sobol_block& S, const vs_u32* __restrict__ tab, int nbits) {
    vs_u32 xh = tab[256 + ((S.gb >> 8) & 255u)];           // uniform addresses: scalar loads
    if (nbits > 16) xh ^= tab[512 + ((S.gb >> 16) & 255u)];
    if (nbits > 24) xh ^= tab[768 + (S.gb >> 24)];
    return xh ^ tab[S.b0];
}

// Which form of the digit step a scan kernel runs.  The fused kernels (no f-values: bound by the vector pipe) take the
// three-instruction form -- always where the host routed a straight-line kernel (ALL: every column has VS_BASE_HORNER),
// by the column's flag in the general code.  The kernels that WRITE f-values keep the four-instruction loop: they are
// bound by their stores, and inside a sustained evaluate -> reduce step the evaluation is SLOWER with the shorter
// arithmetic phase at every k up to 96 (k = 32: 0.40-0.41 ms against 0.365, same box; tools/eval_in_step.py,
// profiles/r04/ab_halton_forms.txt) although it is faster launched alone -- except the eight waves x 16 columns of
// 97 .. 128 columns and x 32 of 129 .. 256, whose arithmetic and store phases overlap worst (config 5: 9.25 -> 8.50 ms;
// k = 200, n = 2^20: 0.82 -> 0.67) -- and, asked again in round 5 in the row-blocked layout (tools/ab_short_digits.sh,
// profiles/r05/ab_short_digits.txt), the one-wave kernels of up to 16 columns, which have no barrier to meet at
// (k = 16, n = 2^22 inside the step: 0.2075 -> 0.1775 ms; k = 32 +2 %, k = 48 +14 %, k = 64 +-0, k = 96 +4 %: unchanged there):
// VS_SCAN_SHORT_DIGITS.
#ifdef VS_HALTON_DIGITS   // A/B build (tools/ab_halton_forms.sh): the four-instruction loop and Markstein's quotient everywhere
#define VS_SCAN_SHORT_DIGITS(FUSED, L, W) false
#else
#ifdef VS_SCAN_SHORT_ALL   // A/B build: the short digit form in every evaluation kernel too
#define VS_SCAN_SHORT_DIGITS(FUSED, L, W) true
#else
#define VS_SCAN_SHORT_DIGITS(FUSED, L, W) ((FUSED) || ((L) >= 16 && (W) == 8) || (W) == 1)
#endif
#endif
template <bool FUSED, bool ALL>
__device__ __forceinline__ void vs_halton_pair(vs_u32 i1, vs_u32 i2, double d1, double d2, const vs_base& B, double& x1, double& x2,
                                               bool may_be_2) {   // base 2 is column 0: only a wave's first column can be it
    if constexpr (!FUSED) vs_radical_inverse_pair_digits(i1, i2, d1, d2, B, x1, x2, may_be_2);
    else if constexpr (ALL) vs_radical_inverse_pair_horner(i1, i2, d1, d2, B, x1, x2, may_be_2);
    else vs_radical_inverse_pair(i1, i2, d1, d2, B, x1, x2, may_be_2);
}

template <bool FUSED>
__device__ __forceinline__ void vs_unit_pair(const vs_design_dev& D, const vs_base& B, vs_u64 r, vs_u64 src2, int c,
                                             double& x1, double& x2, const vs_sobol_block& SB) {
    if (D.generator == VS_GEN_HALTON) {  // B = D.bases[c], fetched one column ahead by the caller
        const vs_u32 i1 = (vs_u32)(D.first_index + r), i2 = (vs_u32)(D.first_index + src2);
        vs_halton_pair<FUSED, false>(i1, i2, (double)i1, (double)i2, B, x1, x2, c == 0);
    } else if (D.generator == VS_GEN_SOBOL) {
        // (the scan kernels take Sobol points from vs_scan_block's table look-ups, issued for all columns at once)
        if (SB.fast) x1 = (double)vs_sobol_bits_block(SB, D.sobol_tabs + 1024 * c, D.sobol_bits) * (1.0 / 4294967296.0);
        else x1 = vs_sobol_point_tab(D.first_index + r, D.sobol_tabs + 1024 * c, D.sobol_bits);
        x2 = vs_sobol_point_tab(D.first_index + src2, D.sobol_tabs + 1024 * c, D.sobol_bits);
    } else {
        x1 = D.raw[r * (vs_u64)D.k + c];
        x2 = D.raw[src2 * (vs_u64)D.k + c];
    }
}

__device__ __attribute__((noinline, const)) double vs_scale_power(double lower, double ratio, double lhi, double llo, double p) {
    return vs_power_scale(lower, ratio, lhi, llo, p);  // scale.py:62 (vs_device_gen.h)
}

// POWER = false (the fused kernels' general variant, GENV 0): no power scaling in the kernel at all -- a call
// in a kernel that carries 4L sums in registers costs spills around every call site.  Power-scaled designs get
// their own fused instances (GENV 3, built-in objectives), so that Sobol / raw designs and the run-time compiled
// kernels do not pay for it.
template <bool POWER>
__device__ __forceinline__ double vs_apply_scale_compact(const vs_design_dev& D, double p, int c) {
    if (D.scale_mode == VS_SCALE_LINEAR) return p * D.width[c] + D.lower[c];   // scale.py:33
    if (POWER && D.scale_mode == VS_SCALE_POWER)
        return vs_scale_power(D.lower[c], D.width[c], D.width[D.k + c], D.width[2 * D.k + c], p);
    return p;
}

// Per-column constants, all wave-uniform (scalar registers)
struct vs_scan_col {
    vs_base B;      // Halton constants of the column
    double w, lo;   // linear scaling: x * w + lo
    double pre;     // the term's parameter (Term::pre), when it has one per column
    double pre2;    // a second per-column value (Term::pre2; the g-function's rounded reciprocal), 0 if none
};

// Term concept: pre(c, params), pre2(c, params, lk) -> the values eval() needs from memory for column c (0 if none;
//               lk = l*k: arrays that follow the l*k parameters of all outputs are reached through it);
//               eval(x, c, params, pre, pre2) -> t_c(x)
template <class Term, int GENV, bool MULTI>
__device__ __forceinline__ void vs_scan_col_load(vs_scan_col& C, const vs_design_dev& D, const double* __restrict__ params,
                                                 int c, int k, int lk) {
    C.B = {2u, 1u, 2.0, 0.5, 0.5};
    C.w = 1.0;
    C.lo = 0.0;
    C.pre = 0.0;
    C.pre2 = 0.0;
    if (c < k) {
        if (GENV == 1 || GENV == 2 || (GENV != 4 && D.generator == VS_GEN_HALTON)) C.B = D.bases[c];
        if (GENV == 2 || (GENV == 4 && D.scale_mode == VS_SCALE_LINEAR)) {   // (GENV 4, identity scaling: x * 1 + 0)
            C.w = D.width[c];
            C.lo = D.lower[c];
        }
        if (!MULTI) {
            C.pre = Term::pre(c, params);
            C.pre2 = Term::pre2(c, params, lk);
        }
    }
}

// Per-wave state carried across the row blocks of a persistent workgroup (fused reduce only).
// The five head sums sab, sa, sb, s1, s2 (numbered 0..4) are functions of a and b, which every wave of the
// workgroup computes with the same bits: sum z is kept by wave z mod W, in its slot z / W -- ceil(5 / W) doubles per
// lane instead of five, all of them in wave 0 (rounds 1-3), where they were what the 128-register fused kernel
// spilled to scratch and carried through it once per block.  Each sum is still one per-lane chain over the
// workgroup's blocks followed by one wave sum: same bits whoever keeps it.
#define VS_HEAD_SLOTS(W) ((5 + (W) - 1) / (W))
template <int L, int W = 1>
struct vs_scan_acc {
    double v[L][4];   // ac, bd, ad, bc of the wave's L columns
    double h[VS_HEAD_SLOTS(W)];
};

template <int L, int W>
__device__ __forceinline__ void vs_head_add(vs_scan_acc<L, W>& acc, int q, double a, double b, double shift) {
    const double dx = a - shift, dy = b - shift;
#pragma unroll
    for (int z = 0; z < 5; ++z) {
        if (W == 1 || q == z % W) {   // q is wave-uniform: a scalar branch
            double& h = acc.h[z / W];
            if (z == 0) h = fma(a, b, h);
            else if (z == 1) h += a;
            else if (z == 2) h += b;
            else if (z == 3) h += dx + dy;
            else {
                h = fma(dx, dx, h);
                h = fma(dy, dy, h);
            }
        }
    }
}

// wave sums of the head sums this wave keeps, into out[z] (global or LDS) by lane 63
template <int L, int W>
__device__ __forceinline__ void vs_head_store(const vs_scan_acc<L, W>& acc, int q, int lane, double* __restrict__ out) {
#pragma unroll
    for (int z = 0; z < 5; ++z) {
        if (W == 1 || q == z % W) {
            const double t = vs_wave_sum_dpp(acc.h[z / W]);
            if (lane == 63) out[z] = t;
        }
    }
}

// Several outputs in one kernel (the fused both-orders kernels): the head sums of output O live in LDS, heads[5 O ..
// 5 O + 4], kept by wave O mod W -- per block one wave sum of each of the five and one read-add-write by lane 63 (some
// thirty instructions of the block's thousands).  In registers (5 LMAX sums spread over the waves like the single
// output's, or per output) they either cost the eight-wave kernel the registers it does not have, or the compiler
// merged the waves' branches into one update through a selected address and moved the whole array to scratch.
template <int O, int W>
__device__ __forceinline__ void vs_head_add_lds(double* __restrict__ heads, int q, int lane, bool valid, double a, double b,
                                                double shift) {
    if (q == O % W) {
        const double dx = a - shift, dy = b - shift;
        const double v0 = valid ? a * b : 0.0, v1 = valid ? a : 0.0, v2 = valid ? b : 0.0, v3 = valid ? dx + dy : 0.0,
                     v4 = valid ? fma(dx, dx, dy * dy) : 0.0;
        const double t0 = vs_wave_sum_dpp(v0), t1 = vs_wave_sum_dpp(v1), t2 = vs_wave_sum_dpp(v2), t3 = vs_wave_sum_dpp(v3),
                     t4 = vs_wave_sum_dpp(v4);
        if (lane == 63) {
            double* h = heads + 5 * O;
            h[0] += t0;
            h[1] += t1;
            h[2] += t2;
            h[3] += t3;
            h[4] += t4;
        }
    }
}

// One 64-row block.  Term::eval(x, c, params) is the per-coordinate term; FOLD is VS_OBJ_LINEAR
// (sum) or anything else (product).  `tot` is this block's LDS slab [2][W][64].
// FUSED = false: stores the 2k+2 values (planar f-values, non-temporal); FUSED = true: adds the
// products a*c_i, b*d_i, a*d_i, b*c_i and the head sums into `acc` and stores nothing.
// TLDS: the 2L terms of the wave live in a wave-private LDS slab `tl` ([2][L][64], conflict-free)
// instead of registers -- the shape for k > 64 in the fused kernel, whose 4L register sums leave no
// room for them.
// The in-segment suffixes are kept for VS_SCAN_SUB columns at a time: a backward walk over the
// segment leaves one checkpoint per sub-block (the fold of everything after it), and the output walk
// rebuilds the sub-block's suffixes from its checkpoint -- the same right-to-left chain, so the same
// bits as keeping all L suffixes, with 2(L/SUB + SUB) registers instead of 2L.
#define VS_SCAN_SUB 4
// Scan position (segment q, place j) -> column.  Segments take every W-th column, not L adjacent ones: the cost of
// a column is its Halton digit count (23 digits in base 2 at 2^23 points, 3 from base 211 on), and the waves of a
// workgroup meet at a barrier -- with adjacent columns the wave holding the first primes does twice the average
// work (k = 32: 83 of 188 digits per point in wave 0 of 4; k = 128: 124 of 490 in wave 0 of 8) while the others
// wait.  Interleaved, every wave gets one column of each run of W primes.  The folds run in scan-position order.
#define VS_SCAN_COLUMN(q, j, W) ((j) * (W) + (q))

// Round 5: the look-ups from LDS.  A 64-lane gather costs the texture path ~54 clocks whatever its addresses, on the one
// such unit a CU has (see vs_scan_block); the same gather from LDS is a ds_read_b32 -- two clocks of the LDS pipe when
// the lanes' addresses fall into one 64-byte table: 16 banks, equal addresses broadcast.  So the points come from NIBBLE
// tables (8 positions x 16 entries x 4 B = 512 B per column; the design keeps them behind its byte tables) in LDS of
// the wave's own, and a point is the XOR of 6 ds_read_b32 (8 beyond 2^24 indices) whose addresses are the lane's nibbles
// (one VGPR per position and point, the same for every column) + a compile-time offset.  Two ways (SNIB, vs_scan_block):
//   2: the tables of all L columns of the wave, staged once per kernel (vs_scan_body) -- the persistent fused kernels, W L
//      512 B beside their slabs (16 KiB at 32 columns, 48 at 96);
//   1: ONE KiB per wave, through which the tables pass two columns at a time, every 64-row block: a wave-wide 16-byte
//      load fetches the pair after next, the pair before it is written over the slot -- LDS operations of a wave complete
//      in order, so the look-ups of the pair that was there are done, no barrier.  Where whole tables have no room (64
//      KiB at 128 columns: beside the terms of the fused kernel of 97 .. 128 columns, the tiles of the both-orders
//      kernels) and in the evaluation kernels, one block per workgroup, where it is the same fetch.
// All look-ups of a block are issued AHEAD of the column loop, as the gathers were: scalar loads and LDS share a counter,
// and the column loop waits for its scalar prefetch (all of the counter: scalar loads answer out of order) once per
// column -- look-ups issued a column ahead inside the loop were drained a few instructions later, their latency in the
// open eight times per block instead of once (k = 32 fused: 0.337 ms against 0.284).
#define VS_SOBOL_NIB 128   // words per column
#ifndef VS_SOBOL_GROUP
#define VS_SOBOL_GROUP 2   // columns whose look-ups are in flight together (vs_sobol_points_lds)
#endif
#define VS_SOBOL_LDS_BYTES(SNIB, L, W) ((size_t)(W) * ((SNIB) == 2 ? (L) : (SNIB) == 1 ? 2 : 0) * VS_SOBOL_NIB * 4)

// the lane's piece of the tables of scan positions j0, j0 + 1 (lanes 0 .. 31: j0, 32 .. 63: j0 + 1)
template <int W>
__device__ __forceinline__ uint4 vs_sobol_fetch(const vs_design_dev& D, int q, int lane, int j0) {
    int c = VS_SCAN_COLUMN(q, j0 + (lane >> 5), W);
    c = c < D.k ? c : D.k - 1;   // (scan positions beyond k: some column's tables, never used; a branch here is a branch per pair)
    return *(const uint4*)(D.sobol_tabs + 1024 * (size_t)D.k + (size_t)c * VS_SOBOL_NIB + 4 * (lane & 31));
}
__device__ __forceinline__ void vs_sobol_put(vs_u32* __restrict__ slot, int lane, uint4 v) {
    *(uint4*)(slot + 4 * lane) = v;   // (lane >> 5) * 128 + 4 * (lane & 31)
    // no instruction: ends the compiler's search for stores that might clobber the design constants loaded further
    // down (see the zeroing of the sums' slab in vs_scan_body)
    __builtin_amdgcn_wave_barrier();
}
// g: Gray code of the point's index; col: the column's 128 words in LDS; NN: nibble positions read -- 6, or 8 for designs
// beyond 2^24 indices (a position beyond the design's Gray-code bits holds nibble 0, whose entry is 0: no test inside,
// a branch around a look-up would put a wait for LDS in front of what follows)
template <int NN>
__device__ __forceinline__ vs_u32 vs_sobol_bits_lds(vs_u32 g, const vs_u32* __restrict__ col) {
    vs_u32 x = 0u;
#pragma unroll
    for (int b = 0; b < NN; ++b) x ^= col[16 * b + ((g >> (4 * b)) & 15u)];
    return x;
}
// the bits of the two points (Gray codes g1, g2) of the wave's L columns
template <int SNIB, int NN, int L, int W, int GROUP>
__device__ __forceinline__ void vs_sobol_points_lds(const vs_design_dev& D, vs_u32* __restrict__ snib, int q, int lane, vs_u32 g1,
                                                    vs_u32 g2, vs_u32 (&sx1)[L], vs_u32 (&sx2)[L]) {
    static_assert(L % 2 == 0, "the Sobol tables pass through LDS two columns at a time");
    uint4 pre = {0u, 0u, 0u, 0u};
    if (SNIB == 1) {
        vs_sobol_put(snib, lane, vs_sobol_fetch<W>(D, q, lane, 0));
        if (L > 2) pre = vs_sobol_fetch<W>(D, q, lane, 2);
    }
#pragma unroll
    for (int j = 0; j < L; ++j) {
        if (SNIB == 1 && j > 0 && j % 2 == 0) {
            vs_sobol_put(snib, lane, pre);
            if (j + 2 < L) pre = vs_sobol_fetch<W>(D, q, lane, j + 2);
        }
        const vs_u32* col = snib + (SNIB == 1 ? j & 1 : j) * VS_SOBOL_NIB;
        sx1[j] = vs_sobol_bits_lds<NN>(g1, col);   // (columns beyond k: whatever the slot holds; never used)
        sx2[j] = vs_sobol_bits_lds<NN>(g2, col);
        // two columns' look-ups in flight (24 .. 32 registers), not all L: left alone the compiler issues every look-up
        // of the block before the first XOR, 12 L results in registers (four at a time measured the same: 0.397 ms at
        // k = 64 either way); one (GROUP, the caller's choice) where the kernel has no registers to spare
        if (j % GROUP == GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- fused evaluate + BOTH orders (NBH > 0; k <= 16 NBH <= 32 columns, one output, four waves) -----------------
// Instead of storing c_i, d_i the output walk writes s_i = c_i + d_i and t_i = c_i - d_i of the block's 64 rows
// into an LDS tile [index row][sample row = lane]: s_i at tile row i, t_i at 16 NBH + i (rows i >= k stay zero),
// a and b behind them; after a second barrier the four waves multiply the tile on the fp64 matrix cores (S S^T and
// T T^T, varsens/saltelli.py:612-619) and take the first-order dot products off the same fragments -- no f-values
// in HBM for the whole of compute_varsens.  Pitch 68 doubles: a wave writes 512 contiguous bytes per tile row, and
// the b128 fragment reads (index row lane & 15, sample pair lane >> 4) fall on the banks of the pitch-36 tile of
// gram_diag_wide_kernel (136 r mod 64 = 72 r mod 64 = 8 r): conflict-free.
#define VS_GT_PITCH 68
#define VS_GT_ROWS(NBH) (32 * (NBH) + 2)
#define VS_GT_DOUBLES(NBH) (VS_GT_ROWS(NBH) * VS_GT_PITCH)
// with up to LMAX outputs: two rows (a_o, b_o) per output behind the index rows
#define VS_GT_DOUBLES_M(NBH, LMAX) ((32 * (NBH) + 2 * (LMAX)) * VS_GT_PITCH)
// LDS doubles of those kernels: totals slabs (one; two with several outputs), the tile, the outputs' variance shifts and
// head sums
#define VS_GT_LDS_DOUBLES(W, NBH, LMAX) \
    (((LMAX) > 1 ? 2 : 1) * 2 * (W) * VS_EVAL_ROWS + VS_GT_DOUBLES_M(NBH, LMAX) + ((LMAX) > 1 ? 6 * (LMAX) : 0))
typedef double vs_d2 __attribute__((ext_vector_type(2)));
typedef double vs_d4 __attribute__((ext_vector_type(4)));

// Wave q of four owns block X = {S0, S1, T0, T1}[q] (NBH = 2) and accumulates
//   d: X X^T over all 64 samples (16 MFMAs), o: its half of the samples of S0 S1^T (q < 2) or T0 T1^T (8 MFMAs),
//   pa, pb: a . x and b . x of its 16 index rows;
// NBH = 1: X = S0 (q < 2) or T0, each wave over its half of the samples (d: 8 MFMAs; no o).
struct vs_gram_regs {
    vs_d4 d, o;
    double pa, pb;
};

// abrow: tile row (behind the 2 HALF index rows) of the a the lane's index row belongs to (b one row further): 0 with
// one output; with several, 2 x the output of index row m = column * l + output -- a per-lane row (see vs_gram_abrow)
template <int NBH>
__device__ __forceinline__ void vs_gram_consume(const double* __restrict__ tile, int q, int lane, vs_gram_regs& G,
                                                int abrow = 0) {
    constexpr int P = VS_GT_PITCH, HALF = 16 * NBH;
    const int part = q >> 1, odd = q & 1;
    const double* base = tile + (size_t)(part * HALF + (lane & 15)) * P + 2 * (lane >> 4);   // block 0 of S or of T
    const double* rx = base + (NBH == 2 ? 16 * odd * P : 0);                                  // the wave's own block
    const double* rab = tile + (size_t)(2 * HALF + abrow) * P + 2 * (lane >> 4);              // a; b one row further
    // eight quarters of 8 samples: lane group g = lane >> 4 holds samples 8 h + 2 g, + 1 of its index row
    auto diag = [&](int h) {
        const vs_d2 x = *(const vs_d2*)(rx + 8 * h);
        const vs_d2 av = *(const vs_d2*)(rab + 8 * h), bv = *(const vs_d2*)(rab + P + 8 * h);
        G.d = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], x[0], G.d, 0, 0, 0);
        G.d = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], x[1], G.d, 0, 0, 0);
        G.pa = fma(av[0], x[0], G.pa);
        G.pb = fma(bv[0], x[0], G.pb);
        G.pa = fma(av[1], x[1], G.pa);
        G.pb = fma(bv[1], x[1], G.pb);
    };
    if (NBH == 2) {
#pragma unroll
        for (int h = 0; h < 8; ++h) diag(h);
#pragma unroll
        for (int h2 = 0; h2 < 4; ++h2) {   // rows of block 0 x columns of block 1, this wave's half of the samples
            const int h = 4 * odd + h2;
            const vs_d2 x0 = *(const vs_d2*)(base + 8 * h), x1 = *(const vs_d2*)(base + 16 * P + 8 * h);
            G.o = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[0], x1[0], G.o, 0, 0, 0);
            G.o = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[1], x1[1], G.o, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int h2 = 0; h2 < 4; ++h2) diag(4 * odd + h2);
    }
}

// End of the kernel: the workgroup's Gram partial [S S^T, T T^T][16 NBH][16 NBH] (upper 16-blocks; the layout
// gram_fold_kernel reads with st = 2) and its first-order record (layout of reduce_first_kernel's partials).
// `lds`: at least 2 * 256 + 128 + 8 doubles, free by now; `acc`: the head sums sab, sa, sb, s1, s2 (vs_scan_acc) of the
// one output.  l outputs: index row m = column * l + output; rec holds l records of 5 + 4k.
// heads: null with one output (the sums are in `acc`), else the LDS sums of vs_head_add_lds
template <int NBH, int L>
__device__ __forceinline__ void vs_gram_epilogue(double* __restrict__ lds, int q, int lane, vs_gram_regs& G,
                                                 const vs_scan_acc<L, 4>& acc, const double* __restrict__ heads, int k, int l,
                                                 double* __restrict__ gout, double* __restrict__ rec) {
    constexpr int SB = 16 * NBH;
    const int part = q >> 1, odd = q & 1;
    double* red2 = lds;               // [part][16][16]: the odd waves' share of a tile the even waves store
    double* dots = lds + 512;         // [wave][16 rows][a, b]
    double* hdl = lds + 512 + 128;    // the five head sums (one output)
    const double* hd = heads ? heads : hdl;
    G.pa += __shfl_xor(G.pa, 16, 64);
    G.pa += __shfl_xor(G.pa, 32, 64);
    G.pb += __shfl_xor(G.pb, 16, 64);
    G.pb += __shfl_xor(G.pb, 32, 64);
    __syncthreads();   // every wave is done with the tile
    if (lane < 16) {
        dots[(q * 16 + lane) * 2 + 0] = G.pa;
        dots[(q * 16 + lane) * 2 + 1] = G.pb;
    }
    if (odd) {
        const vs_d4 share = NBH == 2 ? G.o : G.d;
#pragma unroll
        for (int v = 0; v < 4; ++v) red2[part * 256 + ((lane >> 4) + 4 * v) * 16 + (lane & 15)] = share[v];
    }
    if (!heads) vs_head_store(acc, q, lane, hdl);
    __syncthreads();
    double* o = gout + (size_t)part * SB * SB;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int row = (lane >> 4) + 4 * v, col = lane & 15;   // f64 C/D layout
        if (NBH == 2) {
            o[(16 * odd + row) * SB + 16 * odd + col] = G.d[v];
            if (!odd) o[row * SB + 16 + col] = G.o[v] + red2[part * 256 + row * 16 + col];
        } else if (!odd) {
            o[row * SB + col] = G.d[v] + red2[part * 256 + row * 16 + col];
        }
    }
    // record: head, ac[k], bd[k], ad[k], bc[k] with a.c = (a.s + a.t) / 2, a.d = (a.s - a.t) / 2
    const size_t plen = 5 + 4 * (size_t)k;
    for (size_t ee = threadIdx.x; ee < plen * l; ee += 256) {
        const int out = (int)(ee / plen);
        const size_t e = ee - out * plen;
        double v = hd[5 * out + (e < 5 ? e : 0)];
        if (e >= 5) {
            const int slot = (int)((e - 5) / k), j = (int)((e - 5) % k) * l + out;   // j: the index row
            const int which = (slot == 0 || slot == 2) ? 0 : 1;
            double xs, xt;
            if (NBH == 2) {   // s row j: wave j >> 4, t row j: wave 2 + (j >> 4)
                xs = dots[j * 2 + which];
                xt = dots[(32 + j) * 2 + which];
            } else {          // two waves per block, half of the samples each
                xs = dots[j * 2 + which] + dots[(16 + j) * 2 + which];
                xt = dots[(32 + j) * 2 + which] + dots[(48 + j) * 2 + which];
            }
            v = (slot == 0 || slot == 3) ? 0.5 * (xs + xt) : 0.5 * (xs - xt);
        }
        rec[ee] = v;
    }
}

// ---- the same beyond 32 columns: NB waves x 16 columns, tile of NB 16-row blocks per half, two waves per SIMD at 256
// VGPRs.  NB = 8 (65 .. 128 columns; tile [258][68] = 137 KiB: one workgroup of eight waves per CU) and NB = 4
// (33 .. 64 columns; 69 KiB: two workgroups of four waves per CU).  Tile ownership of gram_st_kernel
// (vs_gram_st.hip): the wave with block row R owns row R of S S^T (NB - R tiles) and row NB-1-R of T T^T (R + 1):
// NB + 1 accumulator tiles, the diagonal ones reuse the A fragments; body compiled per R (straight-line code, every
// tile a compile-time fact), and per number NBL of live blocks (rows from k on are zeros in the tile; multiplying all
// NB blocks whatever k is made k = 96 and k = 48 slower than the two-kernel route: 3.5 vs 3.35, 1.42 vs 1.28 ms).
template <int NB>
struct vs_gramn_regs {
    vs_d4 acc[NB + 1];
    double pas, pbs, pat, pbt;   // a . s, b . s of the 16 rows of S block R; a . t, b . t of T block NB-1-R
};

// Software pipeline over the eight 8-sample quarters of the block, pinned with sched_barrier (left alone every
// ds_read_b128 was followed by s_waitcnt lgkmcnt(0) and its two MFMAs): the A fragments of S row R and T row NB-1-R
// and a, b of quarter h + 1 are read a quarter ahead (two register sets); the NB - 1 other operand blocks of quarter
// h are read at its top and land behind the four MFMAs of the two diagonal tiles, which need the A fragments only.
// (All fragments a quarter ahead -- 88 VGPRs of fragments at NB = 8 -- spilled inside the MFMA section.)
struct vs_gramn_afrags {
    vs_d2 as, at, av, bv;
    vs_d2 avt, bvt;   // several outputs: a, b of the T rows' outputs (av, bv: of the S rows')
};

// NBL <= NB: live 16-row blocks (ceil(k / 16)); tiles with a block row or column from NBL on are all zeros and are
// left out at compile time (the waves' tile counts then differ by at most one or two: k = 96: 6, 5, 5, 5, 5, 5, 5, 6)
// MULTI: several outputs -- the a, b a lane's index row meets are those of the row's output: tile rows abs (S rows of
// block R) and abt (T rows of block NB-1-R) behind the 2 HALF index rows, per lane (vs_gram_abrow)
// (NB = 8 with all eight blocks live -- 113 .. 128 index rows -- takes 2 or 4 outputs only: the output of a row then
// repeats every 16 rows, abs == abt, and one pair of fragments serves both; with two pairs the nine-tile body no longer
// fits its 256 registers.  Up to seven live blocks there is room: three outputs are taken too.)
template <int R, int NB, int NBL, bool MULTI = false>
__device__ __forceinline__ void vs_gramn_consume_row(const double* __restrict__ tile, int lane, vs_gramn_regs<NB>& G,
                                                     int abs = 0, int abt = 0) {
    constexpr bool TWO = MULTI && (NB < 8 || NBL < 8);   // separate a, b fragments for the T rows
    constexpr int P = VS_GT_PITCH, TR = NB - 1 - R, HALF = 16 * NB;
    constexpr bool S_LIVE = R < NBL, T_LIVE = TR < NBL;
    const double* base = tile + (size_t)(lane & 15) * P + 2 * (lane >> 4);   // S block 0; T block 0 is HALF rows further
    // MULTI: an offset of zero the compiler cannot see through, different text in every block row's body: otherwise the
    // a, b fragments of the first quarter -- the same addresses in all NB bodies -- are hoisted above the dispatch on
    // the wave index and one of them is carried through it in scratch (the nine-tile body has no register to spare)
    int zero = 0;
    if (MULTI) asm("; block row %1" : "+s"(zero) : "n"(R));
    const double* rab = tile + (size_t)(2 * HALF + (MULTI ? abs : 0)) * P + 2 * (lane >> 4) + zero;
    const double* rabt = tile + (size_t)(2 * HALF + (MULTI ? abt : 0)) * P + 2 * (lane >> 4) + zero;
    auto load_a = [&](int h, vs_gramn_afrags& F) {
        if (!TWO || S_LIVE) {
            F.av = *(const vs_d2*)(rab + 8 * h);
            F.bv = *(const vs_d2*)(rab + P + 8 * h);
        }
        if (TWO && T_LIVE) {
            F.avt = *(const vs_d2*)(rabt + 8 * h);
            F.bvt = *(const vs_d2*)(rabt + P + 8 * h);
        }
        if (S_LIVE) F.as = *(const vs_d2*)(base + (size_t)(16 * R) * P + 8 * h);
        if (T_LIVE) F.at = *(const vs_d2*)(base + (size_t)(HALF + 16 * TR) * P + 8 * h);
    };
    vs_gramn_afrags F[2];
    load_a(0, F[0]);
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        vs_d2 b[NB - 1];
#pragma unroll
        for (int sl = 1; sl < NB - R; ++sl)
            if (S_LIVE && R + sl < NBL) b[sl - 1] = *(const vs_d2*)(base + (size_t)(16 * (R + sl)) * P + 8 * h);
#pragma unroll
        for (int j = 1; j <= R; ++j)
            if (T_LIVE && TR + j < NBL) b[NB - R - 1 + j - 1] = *(const vs_d2*)(base + (size_t)(HALF + 16 * (TR + j)) * P + 8 * h);
        if (h + 1 < 8) load_a(h + 1, F[(h + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const vs_gramn_afrags& A = F[h & 1];
        if (S_LIVE) {
            G.acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(A.as[0], A.as[0], G.acc[0], 0, 0, 0);
            G.acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(A.as[1], A.as[1], G.acc[0], 0, 0, 0);
        }
        if (T_LIVE) {
            G.acc[NB - R] = __builtin_amdgcn_mfma_f64_16x16x4f64(A.at[0], A.at[0], G.acc[NB - R], 0, 0, 0);
            G.acc[NB - R] = __builtin_amdgcn_mfma_f64_16x16x4f64(A.at[1], A.at[1], G.acc[NB - R], 0, 0, 0);
        }
        if (S_LIVE) {
            G.pas = fma(A.av[0], A.as[0], G.pas);
            G.pbs = fma(A.bv[0], A.as[0], G.pbs);
            G.pas = fma(A.av[1], A.as[1], G.pas);
            G.pbs = fma(A.bv[1], A.as[1], G.pbs);
        }
        if (T_LIVE) {
            const vs_d2 ta_ = TWO ? A.avt : A.av, tb_ = TWO ? A.bvt : A.bv;
            G.pat = fma(ta_[0], A.at[0], G.pat);
            G.pbt = fma(tb_[0], A.at[0], G.pbt);
            G.pat = fma(ta_[1], A.at[1], G.pat);
            G.pbt = fma(tb_[1], A.at[1], G.pbt);
        }
#pragma unroll
        for (int sl = 1; sl < NB - R; ++sl) {
            if (S_LIVE && R + sl < NBL) {
                G.acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(A.as[0], b[sl - 1][0], G.acc[sl], 0, 0, 0);
                G.acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(A.as[1], b[sl - 1][1], G.acc[sl], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 1; j <= R; ++j) {
            if (T_LIVE && TR + j < NBL) {
                G.acc[NB - R + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(A.at[0], b[NB - R - 1 + j - 1][0], G.acc[NB - R + j], 0, 0, 0);
                G.acc[NB - R + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(A.at[1], b[NB - R - 1 + j - 1][1], G.acc[NB - R + j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the wave's NB + 1 tiles into the workgroup's partial [S S^T, T T^T][16 NB][16 NB] (upper 16-blocks)
template <int R, int NB>
__device__ __forceinline__ void vs_gramn_store_row(double* __restrict__ gout, int lane, int nbl, const vs_gramn_regs<NB>& G) {
    constexpr int TR = NB - 1 - R, SB = 16 * NB;
#pragma unroll
    for (int sl = 0; sl < NB + 1; ++sl) {
        const bool ss = sl < NB - R;
        const int rowblk = ss ? R : TR, blk = ss ? R + sl : TR + (sl - (NB - R));
        if (rowblk < nbl && blk < nbl) {   // blocks from ceil(k / 16) on are all zeros: the fold never reads them
            double* o = gout + (ss ? 0 : (size_t)SB * SB);
#pragma unroll
            for (int v = 0; v < 4; ++v)
                o[(size_t)(16 * rowblk + (lane >> 4) + 4 * v) * SB + 16 * blk + (lane & 15)] = G.acc[sl][v];
        }
    }
}

// q < NB: one body per block row (the s_barrier count is the same in all of them: none)
#define VS_GRAMN_DISPATCH(CALL)          \
    switch (q) {                         \
        case 0: CALL(0); break;          \
        case 1: CALL(1); break;          \
        case 2: CALL(2); break;          \
        case 3: CALL(3); break;          \
        case 4: CALL((NB > 4 ? 4 : 0)); break; \
        case 5: CALL((NB > 4 ? 5 : 0)); break; \
        case 6: CALL((NB > 4 ? 6 : 0)); break; \
        default: CALL((NB > 4 ? 7 : 0)); break; \
    }

template <int NB, int NBL, bool MULTI = false>
__device__ __forceinline__ void vs_gramn_consume(const double* __restrict__ tile, int q, int lane, vs_gramn_regs<NB>& G,
                                                 int abs = 0, int abt = 0) {
#define VS_GRAMN_CONSUME(RR) vs_gramn_consume_row<RR, NB, NBL, MULTI>(tile, lane, G, abs, abt)
    VS_GRAMN_DISPATCH(VS_GRAMN_CONSUME)
#undef VS_GRAMN_CONSUME
}

// tile row offset (behind the index rows) of the a that index row m = column * l + output meets: 2 x output
__device__ __forceinline__ int vs_gram_abrow(int m, int l) { return 2 * (m % l); }

// `lds`: at least 4 * 16 NB + 8 doubles, free by now; l outputs: index row m = column * l + output, l records
template <int NB, int L>
__device__ __forceinline__ void vs_gramn_epilogue(double* __restrict__ lds, int q, int lane, int nbl, vs_gramn_regs<NB>& G,
                                                  const vs_scan_acc<L, NB>& acc, const double* __restrict__ heads, int k,
                                                  int l, double* __restrict__ gout, double* __restrict__ rec) {
    constexpr int HALF = 16 * NB;
    double* dots = lds;          // [s rows 0..HALF-1, t rows 0..HALF-1][a, b]
    double* hdl = lds + 4 * HALF;
    const double* hd = heads ? heads : hdl;
    G.pas += __shfl_xor(G.pas, 16, 64);
    G.pas += __shfl_xor(G.pas, 32, 64);
    G.pbs += __shfl_xor(G.pbs, 16, 64);
    G.pbs += __shfl_xor(G.pbs, 32, 64);
    G.pat += __shfl_xor(G.pat, 16, 64);
    G.pat += __shfl_xor(G.pat, 32, 64);
    G.pbt += __shfl_xor(G.pbt, 16, 64);
    G.pbt += __shfl_xor(G.pbt, 32, 64);
    __syncthreads();   // every wave is done with the tile
    if (lane < 16) {
        dots[(16 * q + lane) * 2 + 0] = G.pas;
        dots[(16 * q + lane) * 2 + 1] = G.pbs;
        dots[(HALF + 16 * (NB - 1 - q) + lane) * 2 + 0] = G.pat;
        dots[(HALF + 16 * (NB - 1 - q) + lane) * 2 + 1] = G.pbt;
    }
    if (!heads) vs_head_store(acc, q, lane, hdl);
    __syncthreads();
#define VS_GRAMN_STORE(RR) vs_gramn_store_row<RR, NB>(gout, lane, nbl, G)
    VS_GRAMN_DISPATCH(VS_GRAMN_STORE)
#undef VS_GRAMN_STORE
    const size_t plen = 5 + 4 * (size_t)k;
    for (size_t ee = threadIdx.x; ee < plen * l; ee += 64 * NB) {
        const int out = (int)(ee / plen);
        const size_t e = ee - out * plen;
        double v = hd[5 * out + (e < 5 ? e : 0)];
        if (e >= 5) {
            const int slot = (int)((e - 5) / k), j = (int)((e - 5) % k) * l + out;   // j: the index row
            const int which = (slot == 0 || slot == 2) ? 0 : 1;
            const double xs = dots[j * 2 + which], xt = dots[(HALF + j) * 2 + which];
            v = (slot == 0 || slot == 3) ? 0.5 * (xs + xt) : 0.5 * (xs - xt);
        }
        rec[ee] = v;
    }
}

// The folds of the W segment totals a wave with segment Q needs: the whole row (left to right), the prefix of the
// segments before its own -- the value that chain has when it reaches segment Q, the same additions in the same order --
// and the suffix of those after it (right to left).  One body per Q behind a switch on the wave-uniform q: 2 W + 2 (W - 1
// - Q) folds and nothing else.  (Until round 4 one body served every wave with `w < q` / `w > q` inside the unrolled
// loops: the compiler turned those into selects -- at W = 4 22 folds and 28 v_cndmask per block and output, at W = 8
// 46 and 60.)
template <int FOLD, int W, int Q>
__device__ __forceinline__ void vs_fold_segments_q(const double* __restrict__ totl, double& Pa, double& Pb, double& Sa,
                                                   double& Sb, double& a, double& b) {
    constexpr int T = VS_EVAL_ROWS;
    const double ident = FOLD == VS_OBJ_LINEAR ? 0.0 : 1.0;
    double xa = ident, xb = ident;
    Pa = ident;
    Pb = ident;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        if (w == Q) {
            Pa = xa;
            Pb = xb;
        }
        xa = vs_fold<FOLD>(xa, totl[(0 * W + w) * T]);
        xb = vs_fold<FOLD>(xb, totl[(1 * W + w) * T]);
    }
    a = xa;
    b = xb;
    Sa = ident;
    Sb = ident;
#pragma unroll
    for (int w = W - 1; w > Q; --w) {
        Sa = vs_fold<FOLD>(totl[(0 * W + w) * T], Sa);
        Sb = vs_fold<FOLD>(totl[(1 * W + w) * T], Sb);
    }
}

template <int FOLD, int W>
__device__ __forceinline__ void vs_fold_segments_of(const double* __restrict__ totl, int q, double& Pa, double& Pb,
                                                    double& Sa, double& Sb, double& a, double& b) {
#define VS_FOLD_Q(QQ) vs_fold_segments_q<FOLD, W, (QQ) < W ? (QQ) : 0>(totl, Pa, Pb, Sa, Sb, a, b)
    if constexpr (W == 1) {
        VS_FOLD_Q(0);
    } else {
        switch (q) {
            case 0: VS_FOLD_Q(0); break;
            case 1: VS_FOLD_Q(1); break;
            case 2: VS_FOLD_Q(2); break;
            case 3: VS_FOLD_Q(3); break;
            case 4: VS_FOLD_Q(4); break;
            case 5: VS_FOLD_Q(5); break;
            case 6: VS_FOLD_Q(6); break;
            default: VS_FOLD_Q(7); break;
        }
    }
#undef VS_FOLD_Q
}

// Everything behind the terms of one output: totals exchange, prefixes, suffix checkpoints, outputs.
// ta/tb: the wave's 2L terms in registers (TLDS: in its LDS slab `tl` instead); Ta/Tb their totals.
// NBH > 0 (FUSED): the outputs go into the Gram tile `gt` (tile + lane) instead of the 4L register sums.
// SLDS > 0 (FUSED): the 4 sums of the wave's LAST SLDS columns live in its LDS slab `sl` ([4 SLDS][64] + lane)
// instead of registers (see vs_scan_body).
// Several outputs in the Gram tile (NBH > 0, HOUT >= 0 = the output): column i of this output is index row i * rmul + radd
// (rmul = l, radd = output), its a, b go to rows abrow, abrow + 1 behind the index rows, its head sums to LDS (`heads`).
template <int FOLD, int L, int W, bool FUSED, bool TLDS, int NBH = 0, int SLDS = 0, int HOUT = -1>
__device__ __forceinline__ void vs_scan_finish(const double (&ta)[TLDS ? 1 : L], const double (&tb)[TLDS ? 1 : L],
                                               const double* __restrict__ tl, double Ta, double Tb,
                                               double* __restrict__ tot, int q, int lane, double* __restrict__ f,
                                               vs_u64 rows /* evaluation: the pitch between the planes at f */, int k, bool valid, double shift, vs_scan_acc<L, W>& acc,
                                               double* __restrict__ gt = nullptr, double* __restrict__ sl = nullptr,
                                               int rmul = 1, int radd = 0, int abrow = 0, double* __restrict__ heads = nullptr) {
    constexpr int T = VS_EVAL_ROWS;
    constexpr int SUB = L < VS_SCAN_SUB ? L : VS_SCAN_SUB;
    constexpr int NSUB = L / SUB;
    static_assert(L % SUB == 0, "columns per wave must be a multiple of the suffix sub-block");
#define VS_TA(j) (TLDS ? tl[(size_t)(j) * T] : ta[TLDS ? 0 : (j)])
#define VS_TB(j) (TLDS ? tl[(size_t)(L + (j)) * T] : tb[TLDS ? 0 : (j)])
    tot[(0 * W + q) * T + lane] = Ta;
    tot[(1 * W + q) * T + lane] = Tb;
    __syncthreads();
    // ---- folds over the other segments: prefix (left to right), suffix (right to left), whole row
    double Pa, Pb, Sa, Sb, a, b;
    vs_fold_segments_of<FOLD, W>(tot + lane, q, Pa, Pb, Sa, Sb, a, b);
    // ---- checkpoints of the right-to-left suffix chain: ca[m] = fold of the a-terms after sub-block m
    double ca[NSUB], cb[NSUB];
    {
        double ra = Sa, rb2 = Sb;
#pragma unroll
        for (int m = NSUB - 1; m >= 0; --m) {
            ca[m] = ra;
            cb[m] = rb2;
            // opaque copies: otherwise the compiler recognises the sub-block suffixes below as the values of
            // this very chain and keeps all 2L of them in registers instead of rebuilding 2*SUB at a time
            if (NSUB > 1) asm("" : "+v"(ca[m]), "+v"(cb[m]));
            if (m > 0) {
#pragma unroll
                for (int j = SUB - 1; j >= 0; --j) {
                    ra = vs_fold<FOLD>(VS_TA(m * SUB + j), ra);
                    rb2 = vs_fold<FOLD>(VS_TB(m * SUB + j), rb2);
                }
            }
        }
    }
    // evaluation: f(M_1[r]) and f(M_2[r]) leave now.  Stored at the end (rounds 1-2) they stayed live through the
    // whole output walk; at 16 columns per wave, held to 128 registers, that made the compiler spill three terms and
    // reload them between the stores -- each reload behind `s_waitcnt vmcnt(0)`, a wait for every store issued so
    // far to reach memory (stores and loads share the counter): six per wave (profiles/r03/eval_k128_n24_before.json:
    // scratch 28 B, SQ_INSTS_VMEM_RD 10 per wave).  Now 118 registers, nothing spilled.
    if (!FUSED && q == 0 && valid) {
        VS_STORE(f + lane, a);
        VS_STORE(f + rows + lane, b);
    }
    // ---- outputs, sub-block by sub-block
    double pa = Pa, pb = Pb;
#pragma unroll
    for (int m = 0; m < NSUB; ++m) {
        double ua[SUB], ub[SUB], sa[SUB], sb[SUB];  // the sub-block's terms and the suffixes AFTER each column
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
            ua[j] = VS_TA(m * SUB + j);
            ub[j] = VS_TB(m * SUB + j);
        }
        sa[SUB - 1] = ca[m];
        sb[SUB - 1] = cb[m];
#pragma unroll
        for (int j = SUB - 2; j >= 0; --j) {
            sa[j] = vs_fold<FOLD>(ua[j + 1], sa[j + 1]);
            sb[j] = vs_fold<FOLD>(ub[j + 1], sb[j + 1]);
        }
#pragma unroll
        for (int j = 0; j < SUB; ++j) {
            const int jj = m * SUB + j;
            const int i = VS_SCAN_COLUMN(q, jj, W);
            const double cj = vs_fold<FOLD>(vs_fold<FOLD>(pb, ua[j]), sb[j]);  // N_j[i]: M_2 with column i from M_1
            const double dj = vs_fold<FOLD>(vs_fold<FOLD>(pa, ub[j]), sa[j]);  // N_nj[i]: M_1 with column i from M_2
            pa = vs_fold<FOLD>(pa, ua[j]);
            pb = vs_fold<FOLD>(pb, ub[j]);
            if (NBH > 0) {
                if (i < k) {  // rows past the block's last design row are zeros in the tile
                    const int m = i * rmul + radd;
                    gt[(size_t)m * VS_GT_PITCH] = valid ? cj + dj : 0.0;
                    gt[(size_t)(16 * NBH + m) * VS_GT_PITCH] = valid ? cj - dj : 0.0;
                }
            } else if (FUSED) {
                if (valid) {  // columns beyond k hold the identity; their sums are dropped at the end
                    if (SLDS > 0 && jj >= L - SLDS) {
                        double* s4 = sl + (size_t)(4 * (jj - (L - SLDS))) * VS_EVAL_ROWS;
                        s4[0 * VS_EVAL_ROWS] = fma(a, cj, s4[0 * VS_EVAL_ROWS]);
                        s4[1 * VS_EVAL_ROWS] = fma(b, dj, s4[1 * VS_EVAL_ROWS]);
                        s4[2 * VS_EVAL_ROWS] = fma(a, dj, s4[2 * VS_EVAL_ROWS]);
                        s4[3 * VS_EVAL_ROWS] = fma(b, cj, s4[3 * VS_EVAL_ROWS]);
                        __builtin_amdgcn_wave_barrier();   // see the zeroing of the slab in vs_scan_body
                    } else {
                        acc.v[jj][0] = fma(a, cj, acc.v[jj][0]);
                        acc.v[jj][1] = fma(b, dj, acc.v[jj][1]);
                        acc.v[jj][2] = fma(a, dj, acc.v[jj][2]);
                        acc.v[jj][3] = fma(b, cj, acc.v[jj][3]);
                    }
                }
            } else if (i < k && valid) {
                // wave-uniform plane address + lane: scalar base and one vector offset per store.  (The scalar-base
                // form of the store instruction, through inline asm, was measured in round 3: k = 64 3 % faster,
                // k = 32 7 % slower, k = 128 the same.)
                VS_STORE(f + (vs_u64)(2 + i) * rows + lane, cj);
                VS_STORE(f + (vs_u64)(2 + k + i) * rows + lane, dj);
            }
        }
        // one sub-block's suffixes alive at a time.  Also in the fused first-order kernel: left alone the scheduler
        // forms every c_j, d_j of the segment first and multiplies afterwards, with all of them alive beside the 4 L sums
        // (18 of the 30 spilled dwords of the 8 x 4 kernel came from that).
        if ((!FUSED && L >= 16) || (FUSED && (NBH == 0 || HOUT >= 0))) __builtin_amdgcn_sched_barrier(0);
    }
#undef VS_TA
#undef VS_TB
    if (NBH > 0 && q == 0) {
        gt[(size_t)(32 * NBH + abrow) * VS_GT_PITCH] = valid ? a : 0.0;
        gt[(size_t)(32 * NBH + abrow + 1) * VS_GT_PITCH] = valid ? b : 0.0;
    }
    if constexpr (HOUT >= 0) vs_head_add_lds<HOUT, W>(heads, q, lane, valid, a, b, shift);
    else if (FUSED && valid) vs_head_add(acc, q, a, b, shift);
}

// The outputs of the fused both-orders kernels with several outputs, one after the other from the coordinates xa, xb of
// the wave's L columns (O is a compile-time fact: head sums, totals slab and tile rows are static), guarded by the
// wave-uniform O < l.  The two totals slabs alternate per output, one barrier each -- a wave reaches output O + 2
// only behind the barrier of O + 1, which every wave passes after it has read the totals of O.
template <int O, class Term, int FOLD, int L, int W, int NBH, int LMAX>
__device__ __forceinline__ void vs_scan_gram_outputs(const vs_eval_args& A, const double* __restrict__ params, int q, int lane,
                                                     double* __restrict__ tot, vs_u64 rows, int k, int lk, bool valid,
                                                     const double (&xa)[L], const double (&xb)[L],
                                                     double* __restrict__ shifts, vs_scan_acc<L, W>& acc,
                                                     double* __restrict__ gt) {
    if constexpr (O < LMAX) {
        constexpr int T = VS_EVAL_ROWS;
        if (O < A.l) {
            const double ident = FOLD == VS_OBJ_LINEAR ? 0.0 : 1.0;
            const double* po = params + (size_t)O * k;
            double ua[L], ub[L];
            double Ta = ident, Tb = ident;
#pragma unroll
            for (int j = 0; j < L; ++j) {
                const int c = VS_SCAN_COLUMN(q, j, W);
                ua[j] = ub[j] = ident;
                if (c < k) {
                    const double pre = Term::pre(c, po), pre2 = Term::pre2(c, po, lk);
                    ua[j] = Term::eval(xa[j], c, po, pre, pre2);
                    ub[j] = Term::eval(xb[j], c, po, pre, pre2);
                }
                Ta = vs_fold<FOLD>(Ta, ua[j]);
                Tb = vs_fold<FOLD>(Tb, ub[j]);
            }
            // shifts: LDS, [LMAX] variance shifts, then the [5 LMAX] head sums
            vs_scan_finish<FOLD, L, W, true, false, NBH, 0, O>(
                ua, ub, nullptr, Ta, Tb, tot + (O & 1) * (2 * W * T), q, lane, nullptr, rows, k, valid, shifts[O], acc, gt,
                nullptr, A.l, O, 2 * O, shifts + LMAX);
            // ends the compiler's search for stores that might clobber the design constants loaded further down (see
            // the zeroing of the sums' slab in vs_scan_body): no instruction
            __builtin_amdgcn_wave_barrier();
        }
        vs_scan_gram_outputs<O + 1, Term, FOLD, L, W, NBH, LMAX>(A, params, q, lane, tot, rows, k, lk, valid, xa, xb, shifts, acc,
                                                                gt);
    }
}

// One 64-row block.  MULTI (evaluation only, l > 1): the scaled coordinates of the segment stay in
// registers and every output o builds its own terms from them (parameters params + o*k, plane o of
// the f-values; the two totals slabs alternate per output, one barrier each) -- the generator runs
// once per row whatever l is.  Otherwise one output (`params`, `fplane` are that output's).
// MULTI with NBH > 0 (the fused both-orders kernels, up to LMAX outputs): every output's s, t rows go into the one
// Gram tile (index row = column * l + output), its a, b into their own two tile rows; `shifts`: LMAX variance shifts.
template <class Term, int FOLD, int L, int W, bool FUSED, bool TLDS, bool MULTI, int GENV, int NBH = 0, int SLDS = 0,
          int LMAX = 1, int SNIB = 0>
__device__ __forceinline__ void vs_scan_block(const vs_eval_args& A, const double* __restrict__ params,
                                              int q, int lane, vs_u64 rb, double* __restrict__ tot,
                                              double* __restrict__ tl, double* __restrict__ fplane, double shift,
                                              vs_scan_acc<L, W>& acc, double* __restrict__ gt = nullptr,
                                              double* __restrict__ sl = nullptr, double* __restrict__ shifts = nullptr,
                                              vs_u32* __restrict__ snib = nullptr) {
    // SNIB: Sobol points come from nibble tables in snib, the wave's own LDS (vs_sobol_points_lds) -- 1: a KiB through
    // which the tables pass two columns at a time, every block; 2: the tables of all L columns, staged once per kernel
    // (vs_scan_body) --; 0: from the byte tables in memory
    constexpr int T = VS_EVAL_ROWS;
    // q is loop-invariant in the persistent walk; tied to the block index here, or the optimiser hoists
    // every column's addresses and masks (all derived from q) out of that loop and spills scalars for
    // them.  (Not `volatile`: a volatile asm counts as a memory clobber and turns every scalar load of
    // design constants behind it into a vector load.)
    {
        const vs_u32 rb_lo = (vs_u32)rb;
        asm("" : "+s"(q) : "s"(rb_lo));
    }
    const int k = A.D.k;
    const vs_u64 rows = A.rows;
    const vs_u64 lr = rb * T + lane;
    const bool valid = lr < rows;
    const vs_u64 r = A.row_begin + (valid ? lr : rows - 1);  // clamp: loads stay in range, stores are guarded
    const double ident = FOLD == VS_OBJ_LINEAR ? 0.0 : 1.0;
    const vs_u64 src2 = vs_perm_at(A.D, r);  // row of the un-shuffled second half behind M_2[r]
    // wave-uniform: the block's first row in plane 0 of the row-blocked layout (vs_f_tile; a 64-row block lies inside
    // one row block), and the pitch between the planes of these rows; `fplane` is the start of the f-values
    const vs_ftile ft = vs_f_tile(rows, (2 * (vs_u64)k + 2) * (vs_u64)A.l, A.f_shift, rb * T);
    const vs_u64 fpitch = ft.pitch;
    double* f = FUSED ? nullptr : fplane + ft.base + (MULTI ? 0 : (vs_u64)blockIdx.y * (2 * (vs_u64)k + 2)) * fpitch;
#ifdef VS_SCAN_STAMPS   // diagnostic build (tools/scan_stamps.py): shader-clock stamps of every 512th block, behind the f-values
    // (all stores at the end: a global store in front of the column loop would keep the design constants out of the
    // scalar cache)
    // s_memtime through a plain (non-volatile) asm tied to a value of the place it stamps: the builtin counts as a
    // memory clobber and would do the same to the scalar loads
    long long stamp0, stamp1 = 0;
    asm("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(stamp0) : "v"(lane));
#endif

    // ---- this wave's segment: coordinates -> terms (registers, or the wave's LDS slab) and their totals
    double ta[TLDS ? 1 : L], tb[TLDS ? 1 : L];
    double Ta = ident, Tb = ident;
    // Everything a column needs from memory -- the Halton constants, the scaling bounds, the term's
    // parameter: 14 scalar registers -- is fetched while the column BEFORE it is computed: a scalar load
    // takes hundreds of cycles, and each column's digit loop starts on these values.
    // GENV 1 / 2: Halton below 2^32 with identity / linear scaling -- straight-line code per column;
    // GENV 0: any generator and scaling, decided by (wave-uniform) branches per column (fused: no power scaling);
    // GENV 3: as 0, power scaling included also when fused;
    // GENV 4 (round 5): Sobol with identity or linear scaling -- straight-line as 1 / 2, points from LDS (SNIB != 0).  The
    // general code with the same look-ups spends as long on its branches per column as Halton's digit loops take (k = 64,
    // n = 2^21, fused first order: 0.396 ms; compiled for Sobol and nothing else 0.299; Halton's own instance 0.32).
    // Sobol designs (the general column code): a point is the XOR of table entries selected by the Gray code of its
    // index.  The look-ups of ALL the wave's columns are issued here, ahead of the column loop, which then only converts.
    // SNIB != 0 (round 5, vs_sobol_points_lds): nibble tables in LDS.  SNIB == 0 (the instances with no LDS to spare, the
    // run-time compiled kernels): byte tables in memory, three or four 64-lane gathers per point -- the M_1 point of a
    // block with aligned consecutive indices takes one instead of three (vs_sobol_block) -- at ~54 clocks of the CU's one
    // texture unit each, whatever the addresses: round 4 measured 0.448 ms (k = 32, n = 2^22, fused first order; Halton
    // 0.378), 0.84 (k = 64, n = 2^21; 0.46), 0.863 (k = 128, n = 2^20; 0.443), nibble tables IN MEMORY slower still (one
    // cache line per look-up, twice as many look-ups: 0.95 ms at k = 128), and the M_2 point as a GF(2) product on the
    // vector ALU (three instructions per Gray-code bit, no look-up) worse (0.621 / 1.036 / 1.085): profiles/r04/
    // sobol_vs_halton.txt.  From LDS: 0.275 / 0.396 / 0.464 ms, less than Halton's points cost the same column code
    // (0.364 / 0.497 / 0.516): profiles/r05/sobol_lds.txt.
    vs_sobol_block sobol_block = {false, 0u, 0u};
    vs_u32 sx1[L], sx2[L];
    bool sobol = false;
    static_assert(GENV != 4 || SNIB != 0, "the Sobol instance takes its points from LDS");
    if constexpr (GENV == 0 || GENV == 3 || GENV == 4) {
        sobol = GENV == 4 || A.D.generator == VS_GEN_SOBOL;   // wave-uniform
        if (sobol) {
            if constexpr (SNIB != 0) {
                const vs_u32 i1 = (vs_u32)(A.D.first_index + r), i2 = (vs_u32)(A.D.first_index + A.D.n + src2);
                const vs_u32 g1 = i1 ^ (i1 >> 1), g2 = i2 ^ (i2 >> 1);
                // (one column's look-ups in flight instead of two: the kernel of four waves x 8 columns held to 128 registers)
                constexpr int GROUP = L == 8 && W == 4 && NBH == 0 ? 1 : VS_SOBOL_GROUP;
                if (A.D.sobol_bits <= 24) vs_sobol_points_lds<SNIB, 6, L, W, GROUP>(A.D, snib, q, lane, g1, g2, sx1, sx2);   // wave-uniform
                else vs_sobol_points_lds<SNIB, 8, L, W, GROUP>(A.D, snib, q, lane, g1, g2, sx1, sx2);
            } else {
                sobol_block = vs_sobol_block_of(A.D, r, lane);
#pragma unroll
                for (int j = 0; j < L; ++j) {
                    const int c = VS_SCAN_COLUMN(q, j, W);
                    sx1[j] = sx2[j] = 0u;
                    if (c < k) {
                        const vs_u32* tab = A.D.sobol_tabs + 1024 * c;
                        sx1[j] = sobol_block.fast ? vs_sobol_bits_block(sobol_block, tab, A.D.sobol_bits)
                                                  : vs_sobol_bits_tab(A.D.first_index + r, tab, A.D.sobol_bits);
                        sx2[j] = vs_sobol_bits_tab(A.D.first_index + A.D.n + src2, tab, A.D.sobol_bits);
                    }
                }
            }
        }
    }
    // the two Halton indices of the row as integers and as doubles, once for all columns (the straight-line code
    // converted them again in every column: two of its ~70 vector instructions)
    const vs_u32 hi1 = (vs_u32)(A.D.first_index + r), hi2 = (vs_u32)(A.D.first_index + A.D.n + src2);
    const double hd1 = (double)hi1, hd2 = (double)hi2;
    vs_scan_col Cnext;
    const int lk = A.l * k;
    vs_scan_col_load<Term, GENV, MULTI>(Cnext, A.D, params, VS_SCAN_COLUMN(q, 0, W), k, lk);
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const int c = VS_SCAN_COLUMN(q, j, W);  // wave-uniform
        const vs_scan_col C = Cnext;
        if (j + 1 < L) vs_scan_col_load<Term, GENV, MULTI>(Cnext, A.D, params, VS_SCAN_COLUMN(q, j + 1, W), k, lk);
        double t1 = ident, t2 = ident;
        if (c < k) {
            double x1, x2;
            if (GENV == 1 || GENV == 2) {
                vs_halton_pair<VS_SCAN_SHORT_DIGITS(FUSED, L, W), true>(hi1, hi2, hd1, hd2, C.B, x1, x2, j == 0);   // VS_SCAN_COLUMN(q, j, W) = 0 needs j = 0
                t1 = GENV == 2 ? x1 * C.w + C.lo : x1;   // scale.py:33
                t2 = GENV == 2 ? x2 * C.w + C.lo : x2;
            } else if (GENV == 4) {
                t1 = (double)sx1[j] * (1.0 / 4294967296.0) * C.w + C.lo;   // scale.py:33, or x * 1 + 0 = x (x >= 0)
                t2 = (double)sx2[j] * (1.0 / 4294967296.0) * C.w + C.lo;
            } else {
                if (sobol) {
                    x1 = (double)sx1[j] * (1.0 / 4294967296.0);
                    x2 = (double)sx2[j] * (1.0 / 4294967296.0);
                } else {
                    vs_unit_pair<VS_SCAN_SHORT_DIGITS(FUSED, L, W)>(A.D, C.B, r, A.D.n + src2, c, x1, x2, sobol_block);
                }
                t1 = vs_apply_scale_compact<!FUSED || GENV == 3>(A.D, x1, c);
                t2 = vs_apply_scale_compact<!FUSED || GENV == 3>(A.D, x2, c);
            }
            if (!MULTI) {
                t1 = Term::eval(t1, c, params, C.pre, C.pre2);
                t2 = Term::eval(t2, c, params, C.pre, C.pre2);
            }
        }
        if (TLDS) {
            tl[(size_t)j * T] = t1;
            tl[(size_t)(L + j) * T] = t2;
        } else {
            ta[TLDS ? 0 : j] = t1;   // MULTI: the coordinates
            tb[TLDS ? 0 : j] = t2;
        }
        if (!MULTI) {
            Ta = vs_fold<FOLD>(Ta, t1);
            Tb = vs_fold<FOLD>(Tb, t2);
        }
        // one column at a time: left alone the scheduler hoists the scalar loads of all L columns'
        // constants to the top and spills scalars into vector lanes
        __builtin_amdgcn_sched_barrier(0);
        // Scalar loads return out of order, so the only wait there is is "all of them": without this, the
        // wait for THIS column's constants would sit at the top of the next column, behind the issue of the
        // next prefetch, and cover that too.  Touching one prefetched value here puts the wait at the end of
        // this column, where the loads have had its whole arithmetic to land.
        if (j + 1 < L) asm("" : "+s"(Cnext.B.b));
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!MULTI) {
#ifdef VS_SCAN_STAMPS
        asm("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(stamp1) : "v"(Ta), "v"(Tb));
#endif
        vs_scan_finish<FOLD, L, W, FUSED, TLDS, NBH, SLDS>(ta, tb, tl, Ta, Tb, tot, q, lane, f, fpitch, k, valid, shift, acc,
                                                           gt, sl);
#ifdef VS_SCAN_STAMPS
        if constexpr (!FUSED) {
            if ((blockIdx.x & 511) == 0 && lane == 0) {
                long long* stamps = (long long*)(A.f + (2 * (vs_u64)k + 2) * rows) + ((vs_u64)(blockIdx.x >> 9) * W + (threadIdx.x >> 6)) * 4;
                stamps[0] = stamp0;
                stamps[1] = stamp1;
                long long stamp3;
                asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(stamp3));
                stamps[3] = stamp3;
            }
        }
#endif
    } else if constexpr (NBH > 0) {
        vs_scan_gram_outputs<0, Term, FOLD, L, W, NBH, LMAX>(A, params, q, lane, tot, rows, k, lk, valid, ta, tb, shifts, acc, gt);
    } else {
        const vs_u64 plane = (2 * (vs_u64)k + 2) * fpitch;   // output o's planes of these rows
        for (int o = 0; o < A.l; ++o) {
            int qo = q;
            asm("" : "+s"(qo) : "s"(o));  // as above: keep q-derived addresses inside the loop over outputs
            const double* po = params + (size_t)o * k;
            double ua[L], ub[L];
            Ta = Tb = ident;
#pragma unroll
            for (int j = 0; j < L; ++j) {
                const int c = VS_SCAN_COLUMN(qo, j, W);
                ua[j] = ub[j] = ident;
                if (c < k) {
                    const double pre = Term::pre(c, po), pre2 = Term::pre2(c, po, lk);
                    ua[j] = Term::eval(ta[TLDS ? 0 : j], c, po, pre, pre2);
                    ub[j] = Term::eval(tb[TLDS ? 0 : j], c, po, pre, pre2);
                }
                Ta = vs_fold<FOLD>(Ta, ua[j]);
                Tb = vs_fold<FOLD>(Tb, ub[j]);
            }
            vs_scan_finish<FOLD, L, W, FUSED, false>(ua, ub, tl, Ta, Tb, tot + (o & 1) * (2 * W * T), qo, lane,
                                                     f + (vs_u64)o * plane, fpitch, k, valid, shift, acc);
        }
    }
}

// LDS doubles a workgroup of the scan kernels needs: totals slabs (two, or one when the terms take
// the room) + the wave-private term slabs + the wave-private slabs of the sums kept in LDS (SLDS columns per wave)
template <int L, int W, bool TLDS, int SLDS = 0>
struct vs_scan_lds {
    static constexpr int slabs = TLDS ? 1 : 2;
    static constexpr int totals = slabs * 2 * W * VS_EVAL_ROWS;
    static constexpr int terms = TLDS ? 2 * L * W * VS_EVAL_ROWS : 0;
    static constexpr int doubles = totals + terms + 4 * SLDS * W * VS_EVAL_ROWS;
};

// The whole kernel body.  grid.x workgroups walk the 64-row blocks (FUSED: persistent; else one block
// per workgroup, grid.x = number of blocks), grid.y = output
// (every output is its own plane of f-values / its own record; params + o*k are its parameters).
// FUSED: one partial record per workgroup and output, layout and fold of reduce_first_kernel
// (partials[(blockIdx.x * l + o) * VS_PLEN(k)], VS_PLEN = 5 + 4k).
// SLDS (FUSED): the 4 sums of the last SLDS of a wave's L columns are kept in a wave-private LDS slab ([4 SLDS][64]
// doubles per wave, read - fma - write once per block) instead of registers.  The fused kernel of 25 .. 32 columns
// (four waves x 8 columns, held to 128 registers for four waves per SIMD) is ten registers short of its 4 L = 32
// sums + 2 L terms + the generator's temporaries: rounds 1-3 let the compiler spill -- five of the sums went through
// scratch memory once per block, each reload behind `s_waitcnt vmcnt(0)`.  With the sums of its last column (two or
// four columns for the general generator paths; VS_SCAN_SUMS_IN_LDS, vs_scan_impl.h) in LDS, 2 KiB per wave and
// column: no scratch, same bits (the same fma chain per lane, whichever memory holds the running value).
template <class Term, int FOLD, int L, int W, bool FUSED, bool TLDS, bool MULTI = false, int GENV = 0, int SLDS = 0,
          int SNIB = 0>
__device__ __forceinline__ void vs_scan_body(const vs_eval_args& A, double* __restrict__ lds,
                                             const double* __restrict__ shift_src, double* __restrict__ partials) {
    constexpr int T = VS_EVAL_ROWS;
    const int k = A.D.k;
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index, provably uniform
    const int o = MULTI ? 0 : blockIdx.y;   // MULTI: all outputs inside the block walk
    const double* params = A.params + (size_t)o * k;
    double* fplane = FUSED ? nullptr : A.f;   // (vs_scan_block places the 64-row block and the output's planes)
    const double shift = FUSED ? shift_src[(size_t)o * (2 * (size_t)k + 2)] : 0.0;  // f_o(M_1[row]) of a one-row evaluation
    double* tl = lds + vs_scan_lds<L, W, TLDS, SLDS>::totals + (size_t)q * (2 * L * T) + lane;
    double* sl = lds + vs_scan_lds<L, W, TLDS, SLDS>::totals + vs_scan_lds<L, W, TLDS, SLDS>::terms +
                 (size_t)q * (4 * SLDS * T) + lane;
    vs_scan_acc<L, W> acc;
    if (FUSED) {
#pragma unroll
        for (int j = 0; j < L; ++j) acc.v[j][0] = acc.v[j][1] = acc.v[j][2] = acc.v[j][3] = 0.0;
#pragma unroll
        for (int z = 0; z < VS_HEAD_SLOTS(W); ++z) acc.h[z] = 0.0;
#pragma unroll
        for (int e = 0; e < 4 * SLDS; ++e) sl[(size_t)e * T] = 0.0;   // a lane only ever touches its own entries
        // A plain store with no barrier between it and the loads of the design constants that follow makes the
        // compiler treat those loads as possibly clobbered: they turn from scalar into vector loads (and the "+s"
        // ties on them no longer compile).  A wave barrier costs no instruction and ends that analysis, as the
        // workgroup barrier does behind the totals.
        if (SLDS > 0) __builtin_amdgcn_wave_barrier();
    }
    // Sobol designs (SNIB): the wave's LDS for the nibble tables (vs_sobol_points_lds), behind the kernel's slabs
    vs_u32* const snib = (vs_u32*)(lds + vs_scan_lds<L, W, TLDS, SLDS>::doubles) + (size_t)q * ((SNIB == 2 ? L : 2) * VS_SOBOL_NIB);
    if constexpr (SNIB == 2) {
        if (A.D.generator == VS_GEN_SOBOL) {   // wave-uniform
#pragma unroll
            for (int j0 = 0; j0 < L; j0 += 2) vs_sobol_put(snib + j0 * VS_SOBOL_NIB, lane, vs_sobol_fetch<W>(A.D, q, lane, j0));
        }
    }
    const vs_u64 nblocks = (A.rows + T - 1) / T;
    int parity = 0;
    // two totals slabs: a wave may publish the totals of block n+1 while slower waves still read
    // those of block n; it cannot reach block n+2 before everybody passed the barrier of n+1.
    // (one slab: a second barrier closes the block instead.)
    if (FUSED) {
        for (vs_u64 rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
            vs_scan_block<Term, FOLD, L, W, FUSED, TLDS, false, GENV, 0, SLDS, 1, SNIB>(A, params, q, lane, rb,
                                                                                        lds + parity * (2 * W * T), tl, fplane,
                                                                                        shift, acc, nullptr, sl, nullptr, snib);
            if (TLDS) __syncthreads();
            else parity ^= 1;
        }
    } else {
        // one block per workgroup: every load of design constants then precedes every store of the
        // kernel, so the compiler may take them through the scalar cache (a persistent walk makes them
        // "possibly clobbered" and turns them into vector loads with a wait per column)
        vs_scan_block<Term, FOLD, L, W, FUSED, TLDS, MULTI, GENV, 0, 0, 1, SNIB>(A, params, q, lane, blockIdx.x, lds, tl, fplane,
                                                                                 shift, acc, nullptr, nullptr, nullptr, snib);
    }
    if (FUSED) {
        const size_t plen = 5 + 4 * (size_t)k;
        double* out = partials + ((size_t)blockIdx.x * gridDim.y + o) * plen;
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const int i = VS_SCAN_COLUMN(q, j, W);
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                const double mine = (SLDS > 0 && j >= L - SLDS) ? sl[(size_t)(4 * (j - (L - SLDS)) + z) * T] : acc.v[j][z];
                const double v = vs_wave_sum_dpp(mine);
                if (lane == 63 && i < k) out[5 + (size_t)z * k + i] = v;
            }
        }
        vs_head_store(acc, q, lane, out);
    }
}

// The whole kernel body of the fused evaluate + both-orders pass (see vs_gram_consume): four waves, L columns each,
// k <= 16 NBH; persistent workgroups walk the 64-row blocks with TWO barriers per block -- totals published (inside
// vs_scan_finish), tile written -- and multiply block n while the faster waves generate block n + 1 (a wave writes
// the tile of block n + 1 only behind that block's first barrier, which every wave reaches after its MFMAs of block n;
// one totals slab for the same reason).  LDS: [2][4][64] totals, then the tile.
// Partials: first_partials[blockIdx.x][5 + 4k], gram_partials[blockIdx.x][2][16 NBH][16 NBH].
// LMAX > 1: the instance for 2 .. LMAX outputs (A.l of them; k l <= 16 NBH index rows, m = column * l + output): the
// generator runs once per row, every output builds its terms from the same coordinates and writes its rows into the one
// tile, a_o and b_o into rows of their own behind the index rows; a lane's first-order dot products take the a, b of
// its index row's output.  LDS: totals [slabs][2][4][64] (one slab; two with several outputs), then the tile
// [32 NBH + 2 LMAX][68].  Partials: first_partials[blockIdx.x][l][5 + 4k], gram_partials as before (16 NBH >= k l).
template <class Term, int FOLD, int L, int NBH, int GENV, int LMAX = 1, int SNIB = 0>
__device__ __forceinline__ void vs_scan_gram_body(const vs_eval_args& A, double* __restrict__ lds,
                                                  const double* __restrict__ shift_src,
                                                  double* __restrict__ first_partials,
                                                  double* __restrict__ gram_partials) {
    constexpr int T = VS_EVAL_ROWS, W = 4;
    constexpr bool MULTI = LMAX > 1;
    static_assert(NBH == 1 || NBH == 2, "16 or 32 index rows per half");
    static_assert(L * W >= 16 * NBH || L * W >= 8 || MULTI, "columns");
    const int k = A.D.k, l = MULTI ? A.l : 1;
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* tile = lds + (MULTI ? 2 : 1) * 2 * W * T;
    for (int e = threadIdx.x; e < VS_GT_DOUBLES_M(NBH, LMAX); e += W * T) tile[e] = 0.0;   // rows k l .. 16 NBH - 1 stay zero
    // the outputs' variance shifts: one scalar register pair with one output; with several, LDS behind the tile (LMAX
    // pairs of scalar registers between the blocks were more than the kernel has), and behind them the head sums
    double* shifts = tile + VS_GT_DOUBLES_M(NBH, LMAX);
    if (MULTI && (int)threadIdx.x < LMAX) shifts[threadIdx.x] = (int)threadIdx.x < l ? shift_src[(size_t)threadIdx.x * (2 * (size_t)k + 2)] : 0.0;
    if (MULTI && (int)threadIdx.x < 5 * LMAX) shifts[LMAX + threadIdx.x] = 0.0;
    const double shift0 = shift_src[0];
    vs_scan_acc<L, W> acc;
#pragma unroll
    for (int z = 0; z < VS_HEAD_SLOTS(W); ++z) acc.h[z] = 0.0;
    vs_gram_regs G;
    G.d = G.o = (vs_d4){0.0, 0.0, 0.0, 0.0};
    G.pa = G.pb = 0.0;
    // the wave's index rows: block (q & 1) of its half at NBH = 2, block 0 at NBH = 1
    const int abrow = MULTI ? vs_gram_abrow((NBH == 2 ? 16 * (q & 1) : 0) + (lane & 15), l) : 0;
    // Sobol designs (SNIB): the wave's LDS for the nibble tables (vs_sobol_points_lds), behind everything else
    vs_u32* const snib = (vs_u32*)(lds + VS_GT_LDS_DOUBLES(W, NBH, LMAX)) + (size_t)q * ((SNIB == 2 ? L : 2) * VS_SOBOL_NIB);
    if constexpr (SNIB == 2) {
        if (A.D.generator == VS_GEN_SOBOL) {   // wave-uniform
#pragma unroll
            for (int j0 = 0; j0 < L; j0 += 2) vs_sobol_put(snib + j0 * VS_SOBOL_NIB, lane, vs_sobol_fetch<W>(A.D, q, lane, j0));
        }
    }
    __syncthreads();
    const vs_u64 nblocks = (A.rows + T - 1) / T;
    for (vs_u64 rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
        vs_scan_block<Term, FOLD, L, W, true, false, MULTI, GENV, NBH, 0, LMAX, SNIB>(A, A.params, q, lane, rb, lds, nullptr,
                                                                                      nullptr, shift0, acc, tile + lane,
                                                                                      nullptr, shifts, snib);
        __syncthreads();
        vs_gram_consume<NBH>(tile, q, lane, G, abrow);
    }
    vs_gram_epilogue<NBH>(tile, q, lane, G, acc, MULTI ? shifts + LMAX : nullptr, k, l, gram_partials + (size_t)blockIdx.x * 2 * (16 * NBH) * (16 * NBH),
                          first_partials + (size_t)blockIdx.x * l * (5 + 4 * (size_t)k));
}

// Beyond 32 columns: NB waves x 16 columns (see vs_gramn_consume_row).
// Partials: first_partials[blockIdx.x][5 + 4k], gram_partials[blockIdx.x][2][16 NB][16 NB].
// LMAX > 1: the instance for 2 .. LMAX outputs (see vs_scan_gram_body): NB waves x 8 columns (k <= 8 NB: at least two
// outputs share the 16 NB index rows), NBL = ceil(k l / 16) live blocks.
template <class Term, int FOLD, int GENV, int NB, int NBL, int LMAX = 1, int SNIB = 0>
__device__ __forceinline__ void vs_scan_gramn_body(const vs_eval_args& A, double* __restrict__ lds,
                                                   const double* __restrict__ shift_src,
                                                   double* __restrict__ first_partials,
                                                   double* __restrict__ gram_partials) {
    constexpr bool MULTI = LMAX > 1;
    constexpr int T = VS_EVAL_ROWS, W = NB, L = MULTI ? 8 : 16, NBH = NB;
    static_assert(NB == 4 || NB == 8, "64 or 128 index rows per half");
    static_assert(NBL > NB / 2 && NBL <= NB, "live blocks");
    const int k = A.D.k, l = MULTI ? A.l : 1;
    const int nbl = NBL;   // = ceil(k l / 16), the host's choice of instance
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* tile = lds + (MULTI ? 2 : 1) * 2 * W * T;
    for (int e = threadIdx.x; e < VS_GT_DOUBLES_M(NBH, LMAX); e += W * T) tile[e] = 0.0;   // rows k l .. 16 NB - 1 stay zero
    double* shifts = tile + VS_GT_DOUBLES_M(NBH, LMAX);   // shifts and head sums: see vs_scan_gram_body
    if (MULTI && (int)threadIdx.x < LMAX) shifts[threadIdx.x] = (int)threadIdx.x < l ? shift_src[(size_t)threadIdx.x * (2 * (size_t)k + 2)] : 0.0;
    if (MULTI && (int)threadIdx.x < 5 * LMAX) shifts[LMAX + threadIdx.x] = 0.0;
    const double shift0 = shift_src[0];
    vs_scan_acc<L, W> acc;
#pragma unroll
    for (int z = 0; z < VS_HEAD_SLOTS(W); ++z) acc.h[z] = 0.0;
    const int abs = MULTI ? vs_gram_abrow(16 * q + (lane & 15), l) : 0;              // S rows of block q
    const int abt = MULTI ? vs_gram_abrow(16 * (NB - 1 - q) + (lane & 15), l) : 0;   // T rows of block NB-1-q
    vs_gramn_regs<NB> G;
#pragma unroll
    for (int i = 0; i < NB + 1; ++i) G.acc[i] = (vs_d4){0.0, 0.0, 0.0, 0.0};
    G.pas = G.pbs = G.pat = G.pbt = 0.0;
    // Sobol designs (SNIB): the wave's LDS for the nibble tables (vs_sobol_points_lds), behind everything else
    vs_u32* const snib = (vs_u32*)(lds + VS_GT_LDS_DOUBLES(W, NBH, LMAX)) + (size_t)q * ((SNIB == 2 ? L : 2) * VS_SOBOL_NIB);
    if constexpr (SNIB == 2) {
        if (A.D.generator == VS_GEN_SOBOL) {   // wave-uniform
#pragma unroll
            for (int j0 = 0; j0 < L; j0 += 2) vs_sobol_put(snib + j0 * VS_SOBOL_NIB, lane, vs_sobol_fetch<W>(A.D, q, lane, j0));
        }
    }
    __syncthreads();
    const vs_u64 nblocks = (A.rows + T - 1) / T;
    for (vs_u64 rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
        vs_scan_block<Term, FOLD, L, W, true, false, MULTI, GENV, NBH, 0, LMAX, SNIB>(A, A.params, q, lane, rb, lds, nullptr,
                                                                                      nullptr, shift0, acc, tile + lane,
                                                                                      nullptr, shifts, snib);
        __syncthreads();
        vs_gramn_consume<NB, NBL, MULTI>(tile, q, lane, G, abs, abt);
    }
    vs_gramn_epilogue<NB>(tile, q, lane, nbl, G, acc, MULTI ? shifts + LMAX : nullptr, k, l, gram_partials + (size_t)blockIdx.x * 2 * (16 * NB) * (16 * NB),
                          first_partials + (size_t)blockIdx.x * l * (5 + 4 * (size_t)k));
}

#endif  // VS_DEVICE_SCAN_H
