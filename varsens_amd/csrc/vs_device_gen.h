// Device-side generator code shared by the precompiled kernels (vs_generate.hip) and by the
// kernels hiprtc builds at run time for user objectives (vs_rtc.hip embeds this file as text).
// Device code only; fixed-width types are spelled vs_u32 / vs_u64 so that the text also compiles
// under hiprtc without <cstdint>.
#ifndef VS_DEVICE_GEN_H
#define VS_DEVICE_GEN_H

typedef unsigned int vs_u32;
typedef unsigned long long vs_u64;

#ifndef VARSENS_HIP_H
enum { VS_GEN_HALTON = 0, VS_GEN_SOBOL = 1, VS_GEN_RAW = 2 };
enum { VS_SCALE_IDENTITY = 0, VS_SCALE_LINEAR = 1, VS_SCALE_POWER = 2 };
enum { VS_OBJ_G_FUNCTION = 1, VS_OBJ_ISHIGAMI = 2, VS_OBJ_LINEAR = 3, VS_OBJ_POLYNOMIAL = 4, VS_OBJ_INVERT = 5,
       VS_OBJ_SOURCE = 6 };
#endif

// ---------------------------------------------------------------------------
// wave-level sums
// ---------------------------------------------------------------------------
// wave-wide sum with DPP; total valid in lanes 48..63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double vs_dpp_add(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int tlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    int thi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(thi, tlo);
}

__device__ __forceinline__ double vs_wave_sum_dpp(double v) {
    v = vs_dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = vs_dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = vs_dpp_add<0x141, 0xf>(v);  // row_half_mirror
    v = vs_dpp_add<0x140, 0xf>(v);  // row_mirror: every lane holds its row's sum
    v = vs_dpp_add<0x142, 0xa>(v);  // row_bcast15 into rows 1,3
    v = vs_dpp_add<0x143, 0xc>(v);  // row_bcast31 into rows 2,3
    return v;                       // rows 3 (lanes 48..63): sum of all 64 lanes
}


// Common shift of the one-pass variance sums: the mean of the first min(rows, 1024) values of a
// block, summed in a fixed order by ONE wave (every lane returns the same bits, and every wave or
// kernel that calls it gets the same bits).  A sample mean is within a small fraction of sigma of the
// true mean, so sum (x-s)^2 - (sum (x-s))^2/N loses nothing to cancellation, and one outlier among
// the leading values moves it by outlier/1024 only.
__device__ __forceinline__ double vs_shift_estimate(const double* a, vs_u64 rows);

// Sum over the 64 lanes of a wavefront; result valid in every lane.
__device__ __forceinline__ double vs_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double vs_shift_estimate(const double* a, vs_u64 rows) {
    const int lane = threadIdx.x & 63;
    const vs_u64 m = rows < 1024 ? rows : 1024;
    double s = 0.0;
    for (vs_u64 i = lane; i < m; i += 64) s += a[i];
    s = vs_wave_sum_dpp(s);  // total in lanes 48..63
    const int lo = __builtin_amdgcn_readlane(__double2loint(s), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(s), 63);
    return m ? __hiloint2double(hi, lo) / (double)m : 0.0;
}

#define VS_EVAL_ROWS 64
#define VS_EVAL_WAVES 4

#define VS_STORE(p, v) __builtin_nontemporal_store((v), (p))

// ---------------------------------------------------------------------------
// Halton radical inverse
// ---------------------------------------------------------------------------

struct vs_base {
    vs_u32 b;        // prime base
    vs_u32 ndig;     // digits of the largest index of this design in base b, padded to three (0: indices reach 2^32);
                     // bit 31 (VS_BASE_HORNER, set by make_base): at least three digits, largest index * b^(ndig-1)
                     // < 2^53 -- the three-instruction digit step is exact -- and b^ndig <= 2^49 -- the two-word
                     // reciprocal gives the correctly rounded quotient
    double top;      // b^(ndig-1): weight of the first reversed digit; the denominator b^ndig is top * b (exact)
    double inv_den;  // RN(1 / b^ndig)
    double inv_b;    // RN(1 / b) pushed up two ulps: floor(j * inv_b) == j / b for every j < 2^32
    double inv_lo;   // RN(1 / b^ndig - inv_den): the reciprocal's second word (vs_halton_quotients_dd)
};
#define VS_BASE_HORNER 0x80000000u
#define VS_BASE_DIGITS(B) ((B).ndig & 0x7fffffffu)

// Radical inverse of i in base b: the digit-reversed integer of i divided by
// b^digits, as ONE correctly rounded fp64 quotient of two exact integers
// (i * b < 2^53 is checked on the host).
//
// Fast path (all indices of the design < 2^32): every lane runs the same ndig
// digit steps; leading zero digits scale numerator and denominator by the same
// power of b, which leaves the quotient -- and its rounding -- unchanged.  The
// denominator is then a per-column constant and the division is Markstein's
// fma sequence on its correctly rounded reciprocal: two Newton steps, the last
// of which is provably the correctly rounded quotient.  Base 2 is a bit reversal.
__device__ __forceinline__ double vs_radical_inverse(vs_u64 i, const vs_base B) {
    if (B.ndig != 0) {
        const vs_u32 j = (vs_u32)i;
        if (B.b == 2) return (double)__brev(j) * (1.0 / 4294967296.0);
        const double bd = (double)B.b;
        if (B.ndig & VS_BASE_HORNER) {   // three instructions per digit, two for the quotient: vs_radical_inverse_pair_horner
            const double d = (double)j;
            const double q1 = floor(d * B.inv_b);
            double jj = floor(q1 * B.inv_b);
            double p = fma(q1, bd, jj);
            for (vs_u32 t = VS_BASE_DIGITS(B) - 3u; t; --t) {
                jj = floor(jj * B.inv_b);
                p = fma(p, bd, jj);
            }
            const double n = fma(fma(-bd, bd, 1.0), p, d * B.top);
            return fma(n, B.inv_den, n * B.inv_lo);
        }
        const double den = B.top * bd;
        double num = 0.0, jd = (double)j;
        for (vs_u32 t = 0; t < VS_BASE_DIGITS(B); ++t) {   // see vs_radical_inverse_pair_digits for why floor() is exact
            const double q = floor(jd * B.inv_b);
            num = fma(num, bd, fma(-q, bd, jd));
            jd = q;
        }
        double q0 = num * B.inv_den;
        double r0 = fma(-q0, den, num);
        double q1 = fma(r0, B.inv_den, q0);
        double r1 = fma(-q1, den, num);
        return fma(r1, B.inv_den, q1);
    }
    double num = 0.0, den = 1.0;
    const double bd = (double)B.b;
    while (i) {  // indices beyond 2^32: plain 64-bit digit loop and a true division
        vs_u64 q = i / B.b;
        vs_u32 d = (vs_u32)(i - q * B.b);
        num = fma(num, bd, (double)d);
        den = den * bd;
        i = q;
    }
    return num / den;
}

// The correctly rounded quotients n1 / den, n2 / den of two exact integers below den = b^ndig < 2^53: Markstein's fma
// sequence on the host-rounded reciprocal (two Newton steps, the last one provably the rounded quotient).
__device__ __forceinline__ void vs_halton_quotients(double n1, double n2, double den, double inv_den, double& x1, double& x2) {
    double a0 = n1 * inv_den, b0 = n2 * inv_den;
    double ra = fma(-a0, den, n1), rb = fma(-b0, den, n2);
    a0 = fma(ra, inv_den, a0);
    b0 = fma(rb, inv_den, b0);
    ra = fma(-a0, den, n1);
    rb = fma(-b0, den, n2);
    x1 = fma(ra, inv_den, a0);
    x2 = fma(rb, inv_den, b0);
}

// The same quotients in TWO instructions per point where the denominator is small enough (round 4).  With the
// reciprocal as a double-double y = inv_den + inv_lo + d, |d| <= u^2 y (u = 2^-53), the fma below rounds
//     V = n inv_den + RN(n inv_lo),     |V - n/den| <= 2 u^2 (n/den) (1 + 2u),
// once.  n/den is a ratio of integers with an odd (or power-of-two) denominator b^ndig: it is never a rounding
// boundary, and its distance to the nearest one is at least ulp(n/den) / (4 den) >= u (n/den) / (4 den).  So for
// den < 2^53 / 8.1 no boundary lies between V and n/den, and RN(V) = RN(n/den): the host sets VS_BASE_HORNER only for
// den <= 2^49 (twice that margin).  Tests: tests/test_radical_inverse_math.py restates it over exact rationals.
__device__ __forceinline__ void vs_halton_quotients_dd(double n1, double n2, double inv_den, double inv_lo, double& x1, double& x2) {
    x1 = fma(n1, inv_den, n1 * inv_lo);
    x2 = fma(n2, inv_den, n2 * inv_lo);
}

// Two radical inverses in the same base with interleaved digit loops (two independent dependency chains per step),
// indices below 2^32 (B.ndig != 0), one digit at a time: 4 instructions per digit and point.  The form for designs whose
// indices are too large for vs_radical_inverse_pair_horner.
__device__ __forceinline__ void vs_radical_inverse_pair_digits(vs_u32 i1, vs_u32 i2, double d1, double d2, const vs_base B,
                                                                double& x1, double& x2, bool may_be_2 = true) {   // d1, d2 = (double)i1, i2
    // Digit steps entirely in fp64: with inv_b two ulps
    // above 1/b, floor(j*inv_b) is the exact quotient for j < 2^32, b < 2^16 -- an exact multiple
    // can no longer round below its integer, and the inflation (< 2^-17 absolute) is far smaller
    // than the distance 1/b of any other quotient to the next integer.  The digit j - q*b and the
    // numerator update are exact fmas on integers below 2^53.
    if (may_be_2 && B.b == 2) {
        x1 = (double)__brev(i1) * (1.0 / 4294967296.0);
        x2 = (double)__brev(i2) * (1.0 / 4294967296.0);
        return;
    }
    const double bd = (double)B.b;
    double n1 = 0.0, n2 = 0.0;
    double j1 = d1, j2 = d2;
    vs_u32 t = VS_BASE_DIGITS(B);
    for (; t >= 2; t -= 2) {  // two digits per trip: no register rotation between steps
        const double q1 = floor(j1 * B.inv_b);
        const double q2 = floor(j2 * B.inv_b);
        n1 = fma(n1, bd, fma(-q1, bd, j1));
        n2 = fma(n2, bd, fma(-q2, bd, j2));
        j1 = floor(q1 * B.inv_b);
        j2 = floor(q2 * B.inv_b);
        n1 = fma(n1, bd, fma(-j1, bd, q1));
        n2 = fma(n2, bd, fma(-j2, bd, q2));
    }
    if (t) {
        const double q1 = floor(j1 * B.inv_b);
        const double q2 = floor(j2 * B.inv_b);
        n1 = fma(n1, bd, fma(-q1, bd, j1));
        n2 = fma(n2, bd, fma(-q2, bd, j2));
    }
    vs_halton_quotients(n1, n2, B.top * bd, B.inv_den, x1, x2);
}

// The same two numerators in THREE instructions per digit and point (round 4).  With j_0 = i, j_t = floor(j_{t-1} / b)
// the digits are d_t = j_t - b j_{t+1}, and the digit-reversed integer over T + 1 = ndig digits telescopes:
//     sum_t d_t b^(T-t) = j_0 b^T - (b^2 - 1) P,     P = sum_{t=1..T} j_t b^(T-t) = (((j_1) b + j_2) b + ...) + j_T,
// a Horner sum of the quotients themselves -- no digit is ever formed: per step one product, one floor and one fma, then
// two instructions per point and one per column (b^2 - 1).  Everything is an integer; the largest is j_0 b^T
// <= (largest index of the design) * b^(ndig-1), which the host checks to be below 2^53 before it sets VS_BASE_HORNER
// (every design below 9.49e7 indices in all, i.e. up to n = 47 million rows); so the numerator is the same exact integer
// as the digit loop's and the quotient the same bits.  Base 2 is a bit reversal in either form.
__device__ __forceinline__ void vs_radical_inverse_pair_horner(vs_u32 i1, vs_u32 i2, double d1, double d2, const vs_base B,
                                                                double& x1, double& x2, bool may_be_2 = true) {   // d1, d2 = (double)i1, i2
    if (may_be_2 && B.b == 2) {   // may_be_2: false where the caller knows the column is not the design's first
        x1 = (double)__brev(i1) * (1.0 / 4294967296.0);
        x2 = (double)__brev(i2) * (1.0 / 4294967296.0);
        return;
    }
    const double bd = (double)B.b;
    // T = ndig - 1 >= 2 steps (the host gives every column with VS_BASE_HORNER at least three digits).  The first two are
    // peeled: P starts as j_1 b + j_2 out of fresh registers, i itself stays where it is for the last line, and the loop
    // over the remaining steps updates its quotient in place -- no register copy anywhere.  One step per trip: its
    // three scalar loop instructions issue beside the other waves' vector work (two steps per trip needed a tail for
    // odd T, which the compiler turned into six more instructions and four selects for every column).
    const double q1 = floor(d1 * B.inv_b), q2 = floor(d2 * B.inv_b);
    double j1 = floor(q1 * B.inv_b), j2 = floor(q2 * B.inv_b);
    double p1 = fma(q1, bd, j1), p2 = fma(q2, bd, j2);
    for (vs_u32 t = VS_BASE_DIGITS(B) - 3u; t; --t) {
        j1 = floor(j1 * B.inv_b);
        j2 = floor(j2 * B.inv_b);
        p1 = fma(p1, bd, j1);
        p2 = fma(p2, bd, j2);
    }
    const double m = fma(-bd, bd, 1.0);          // -(b^2 - 1)
    const double n1 = fma(m, p1, d1 * B.top);    // both terms exact integers below 2^53, and so is their sum
    const double n2 = fma(m, p2, d2 * B.top);
    vs_halton_quotients_dd(n1, n2, B.inv_den, B.inv_lo, x1, x2);
}

// whichever form the column's record allows (wave-uniform branch): the general column code, the chain and run-time
// compiled kernels; the straight-line scan kernels (GENV 1 / 2) are routed to by the host only when every column has
// VS_BASE_HORNER and call vs_radical_inverse_pair_horner directly
__device__ __forceinline__ void vs_radical_inverse_pair(vs_u32 i1, vs_u32 i2, double d1, double d2, const vs_base B,
                                                        double& x1, double& x2, bool may_be_2 = true) {
    if (B.ndig & VS_BASE_HORNER) vs_radical_inverse_pair_horner(i1, i2, d1, d2, B, x1, x2, may_be_2);
    else vs_radical_inverse_pair_digits(i1, i2, d1, d2, B, x1, x2, may_be_2);
}
__device__ __forceinline__ void vs_radical_inverse_pair(vs_u32 i1, vs_u32 i2, const vs_base B, double& x1, double& x2) {
    vs_radical_inverse_pair(i1, i2, (double)i1, (double)i2, B, x1, x2);
}

// Sobol point of index i for one coordinate: XOR of the direction integers selected by the set
// bits of the Gray code of i.  The 32 direction integers of the column are fetched up front (wide
// scalar loads) so that the bit loop is pure VALU; bits above `nbits` (the Gray-code bits any
// index of the design can set) are skipped eight at a time.
// (noinline: its 32 scalar direction words must not raise the SGPR pressure of the Halton path, where
// the kernels spilled scalars into vector lanes in their prologue when this was inlined)
__device__ __attribute__((noinline)) double vs_sobol_point(vs_u64 i, const vs_u32* __restrict__ dir, int nbits) {
    const vs_u32 g = (vs_u32)(i ^ (i >> 1));
    vs_u32 x = 0;
#pragma unroll
    for (int j0 = 0; j0 < 32; j0 += 8) {
        if (j0 < nbits) {
            vs_u32 d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = dir[j0 + j];
#pragma unroll
            for (int j = 0; j < 8; ++j) x ^= (vs_u32)(-(int)((g >> (j0 + j)) & 1u)) & d[j];
        }
    }
    return (double)x * (1.0 / 4294967296.0);
}

// The same point from per-column byte tables (built once per design on the host): tab[b][v] is the
// XOR of the direction integers selected by the bits of byte value v at byte position b of the Gray
// code -- three or four cached 4-byte reads and XORs instead of ~3 vector instructions per bit.
__device__ __forceinline__ vs_u32 vs_sobol_bits_tab(vs_u64 i, const vs_u32* __restrict__ tab, int nbits) {
    const vs_u32 g = (vs_u32)(i ^ (i >> 1));
    vs_u32 x = tab[g & 255u] ^ tab[256 + ((g >> 8) & 255u)];
    if (nbits > 16) x ^= tab[512 + ((g >> 16) & 255u)];
    if (nbits > 24) x ^= tab[768 + (g >> 24)];
    return x;
}
__device__ __forceinline__ double vs_sobol_point_tab(vs_u64 i, const vs_u32* __restrict__ tab, int nbits) {
    return (double)vs_sobol_bits_tab(i, tab, nbits) * (1.0 / 4294967296.0);
}
struct vs_design_dev {
    int generator, k, scale_mode;
    int sobol_bits;  // Gray-code bits in use (VS_GEN_SOBOL)
    int halton_horner;   // VS_GEN_HALTON: every column's record has VS_BASE_HORNER (the straight-line kernels need it)
    int reserved0;
    vs_u64 n, first_index;
    const double* lower;
    const double* width;    // LINEAR: upper - lower [k]; POWER: ratio [k], then ln ratio as hi [k] and lo [k]
    const vs_u32* perm;     // entries of rows [perm_begin, perm_begin + perm_count) of the seed-1 shuffle, or null
    vs_u64 perm_begin;
    const double* raw;
    const vs_u32* dirs;
    const vs_u32* sobol_tabs;   // byte tables [k][4][256], then nibble tables [k][8][16] (VS_SOBOL_NIB words per column)
    const vs_base* bases;
};

// Unit-cube coordinate c of sequence row `seq_row` (0..2n-1: first half feeds
// M_1, second half M_2 before the shuffle).
__device__ __forceinline__ double vs_unit_point(const vs_design_dev& D, vs_u64 seq_row, int c) {
    if (D.generator == VS_GEN_HALTON) return vs_radical_inverse(D.first_index + seq_row, D.bases[c]);
    if (D.generator == VS_GEN_SOBOL) return vs_sobol_point_tab(D.first_index + seq_row, D.sobol_tabs + 1024 * c, D.sobol_bits);
    return D.raw[seq_row * (vs_u64)D.k + c];
}

// row of the un-shuffled second half that the seed-1 shuffle moved to row r (r inside the design's slice)
__device__ __forceinline__ vs_u64 vs_perm_at(const vs_design_dev& D, vs_u64 r) {
    return D.perm ? (vs_u64)D.perm[r - D.perm_begin] : r;
}

// Power scaling, lower * (upper / lower)^x (scale.py:62), as lower * exp(x ln r): ln r comes from the host as a
// double-double (lhi + llo, logl of the ratio the reference forms), y = RN(x lhi), e = the rounding error of that
// product + x llo, and exp(y + e) = exp(y) (1 + e) to first order (|e| < 1e-15): one exp per point instead of a pow
// (about a third of its instructions), 1-2 ulp from libm's pow.  A ratio without a finite logarithm keeps pow().
__device__ __forceinline__ double vs_power_scale(double lower, double ratio, double lhi, double llo, double p) {
    if (lhi - lhi == 0.0) {   // finite: wave-uniform
        const double y = p * lhi;
        const double e = fma(p, lhi, -y) + p * llo;
        const double r = exp(y);
        // exp overflowed (raw points at or above 1 with a huge ratio): fma(inf, e, inf) is NaN for e <= 0 where
        // pow(ratio, p) is inf -- the correction only applies to a finite r
        return lower * (r - r == 0.0 ? fma(r, e, r) : r);
    }
    return lower * pow(ratio, p);
}

__device__ __forceinline__ double vs_apply_scale(const vs_design_dev& D, double p, int c) {
    if (D.scale_mode == VS_SCALE_LINEAR) return p * D.width[c] + D.lower[c];   // scale.py:33
    if (D.scale_mode == VS_SCALE_POWER)                                        // scale.py:62
        return vs_power_scale(D.lower[c], D.width[c], D.width[D.k + c], D.width[2 * D.k + c], p);
    return p;
}

// M_1[r][c] / M_2[r][c] (M_2 after the shuffle)
__device__ __forceinline__ double vs_m1(const vs_design_dev& D, vs_u64 r, int c) {
    return vs_apply_scale(D, vs_unit_point(D, r, c), c);
}
__device__ __forceinline__ double vs_m2(const vs_design_dev& D, vs_u64 r, int c) {
    vs_u64 src = vs_perm_at(D, r);
    return vs_apply_scale(D, vs_unit_point(D, D.n + src, c), c);
}


// ---------------------------------------------------------------------------
// The f-value layout on the device: ROW-BLOCKED (include/varsens_hip.h).  The shard's `rows` draws are cut into row
// blocks of 2^sh rows (VS_F_SHIFT = 11: 2048, the widest row tile of any kernel); inside a row block the `planes` = l (2k+2) planes -- plane p =
// output * (2k+2) + block, block order [fM_1, fM_2, fN_j[0..k), fN_nj[0..k)] -- follow each other, each holding the
// row block's rows contiguously:
//     f[(R << sh) * planes + p * pitch_R + (r - (R << sh))],   R = r >> sh,   pitch_R = min(2^sh, rows - (R << sh))
// (the last row block is packed at its own row count, so the array has exactly planes * rows doubles; a shard of up to
// 2^sh rows is one row block: plane after plane, the planar layout of ABI <= 7).  Why: a workgroup that produces or
// consumes a tile of rows touches ALL planes of those rows; plane-major (planar) storage puts them `rows` doubles
// apart -- 258 address streams 128 MiB apart at k = 128, n = 2^24, one TLB entry and one DRAM page each -- where the
// row-blocked form keeps them within planes * 16 KiB: the k = 128, n = 2^24 evaluation stores at 5.6-5.9 instead of
// 4.1 TB/s (profiles/r05/fshift_sweep.txt, eval_blocked_layout.txt).  Every kernel works on row tiles that lie inside one row block
// (tile sizes are powers of two <= 2048 <= 2^sh, tiles aligned to their size): per tile one (base, pitch) pair,
// wave-uniform, and the address of (plane p, i-th row of the tile) is f + base + p * pitch + i.
// ---------------------------------------------------------------------------
#ifndef VS_F_SHIFT
#define VS_F_SHIFT 11          // default log2 of the rows per row block
#endif
#ifndef VS_F_SHIFT_MIN
#define VS_F_SHIFT_MIN 11      // the widest row tile of any kernel is 2048 rows (reduce_first_kernel)
#endif
struct vs_ftile {
    vs_u64 base;    // doubles from f to (plane 0, row r0)
    vs_u64 pitch;   // doubles between consecutive planes of one row
};
__host__ __device__ __forceinline__ vs_ftile vs_f_tile(vs_u64 rows, vs_u64 planes, int sh, vs_u64 r0) {
    const vs_u64 first = (r0 >> sh) << sh;            // first row of r0's row block
    const vs_u64 left = rows - first, full = (vs_u64)1 << sh;
    vs_ftile t;
    t.pitch = left < full ? left : full;
    t.base = first * planes + (r0 - first);
    return t;
}
// one element (plane p, row r): for the kernels that own a row, not a tile
__host__ __device__ __forceinline__ vs_u64 vs_f_index(vs_u64 rows, vs_u64 planes, int sh, vs_u64 p, vs_u64 r) {
    const vs_ftile t = vs_f_tile(rows, planes, sh, r);
    return t.base + p * t.pitch;
}

struct vs_eval_args {
    vs_design_dev D;
    int kind, l;
    int plain_recip;   // host side only -- G_FUNCTION: every 1 + a has a Markstein-safe reciprocal (else: general column code)
    int f_shift;       // log2 of the rows per row block of the f-value layout (vs_f_tile)
    const double* params;
    vs_u64 row_begin, rows;
    double* f;
};

// ---------------------------------------------------------------------------
// the chain scheme of separable objectives (used by eval_kernel, fused_first_kernel and the
// run-time compiled separable kernel)
// ---------------------------------------------------------------------------
#define VS_CHAINS 8                                 // chains per thread and pass
#define VS_PASS (VS_EVAL_WAVES * VS_CHAINS)         // blocks i handled per pass
#ifndef VS_TILE_AHEAD
#define VS_TILE_AHEAD 4                             // columns whose tile reads are in flight ahead of their folds
#endif



template <int KIND>
__device__ __forceinline__ double vs_fold(double acc, double t) {
    return KIND == VS_OBJ_LINEAR ? acc + t : acc * t;
}

// per-coordinate terms of the built-in separable objectives (include/varsens_hip.h)
template <int KIND>
__device__ __forceinline__ double vs_term(double x, double p) {
    if (KIND == VS_OBJ_G_FUNCTION) return (fabs(4.0 * x - 2.0) + p) / (1.0 + p);
    if (KIND == VS_OBJ_POLYNOMIAL) return 1.0 + p * (x - 0.5);
    return p * x;  // VS_OBJ_LINEAR
}

// Tile reads.  The coordinate / term tile normally lives in LDS.  For k too large for LDS it
// lives in a per-workgroup slice of an HBM scratch buffer that a persistent workgroup rewrites
// for every 64-row block; those reads must not be served from this CU's L1 (it may still hold
// the previous block's lines), so they are agent-scope relaxed loads (L2), ordered behind the
// writers' stores by the workgroup barrier (which drains vmcnt).
template <bool SCRATCH>
__device__ __forceinline__ double vs_tile_ld(const double* p) {
    if (SCRATCH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// One full pass (32 consecutive columns i0..i0+31) of the chain scheme for wave Q.
// Chain m of this wave is block i0+Q+4m; at relative column j the chains with
// Q+4m < j are running and the one with Q+4m == j starts from the prefixes.
// Everything about the schedule is a compile-time constant, so the 16 chain
// values stay in fixed registers.
template <int KIND, int Q, bool SCRATCH, int W = VS_EVAL_WAVES>
__device__ __forceinline__ void vs_chain_pass(const double* F1, const double* F2, int i0, double& pa, double& pb,
                                              double (&cj)[VS_PASS / W], double (&dj)[VS_PASS / W]) {
    constexpr int T = VS_EVAL_ROWS;
    constexpr int CH = VS_PASS / W;
    // The tile reads run VS_TILE_AHEAD columns ahead of the folds that use them (left to itself the
    // compiler issues each pair of reads right in front of its folds, and the wave sits out the LDS
    // latency once per column); the scheduling barrier keeps the reads of the next batch in front
    // of the folds of this one.
    constexpr int B = VS_TILE_AHEAD;
    double na[B], nb[B];
#pragma unroll
    for (int t = 0; t < B; ++t) {
        na[t] = vs_tile_ld<SCRATCH>(F1 + (size_t)(i0 + t) * T);
        nb[t] = vs_tile_ld<SCRATCH>(F2 + (size_t)(i0 + t) * T);
    }
#pragma unroll
    for (int j0 = 0; j0 < VS_PASS; j0 += B) {
        double ca[B], cb[B];
#pragma unroll
        for (int t = 0; t < B; ++t) {
            ca[t] = na[t];
            cb[t] = nb[t];
        }
        if (j0 + B < VS_PASS) {
#pragma unroll
            for (int t = 0; t < B; ++t) {
                na[t] = vs_tile_ld<SCRATCH>(F1 + (size_t)(i0 + j0 + B + t) * T);
                nb[t] = vs_tile_ld<SCRATCH>(F2 + (size_t)(i0 + j0 + B + t) * T);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < B; ++t) {
            const int j = j0 + t;
            const double ta = ca[t], tb = cb[t];
#pragma unroll
            for (int m = 0; m < CH; ++m) {
                if (Q + W * m < j) {
                    cj[m] = vs_fold<KIND>(cj[m], tb);
                    dj[m] = vs_fold<KIND>(dj[m], ta);
                } else if (Q + W * m == j) {
                    cj[m] = vs_fold<KIND>(pb, ta);  // N_j[i0+j]: M_2 with column i0+j from M_1
                    dj[m] = vs_fold<KIND>(pa, tb);  // N_nj[i0+j]: M_1 with column i0+j from M_2
                }
            }
            pa = vs_fold<KIND>(pa, ta);
            pb = vs_fold<KIND>(pb, tb);
        }
    }
}

// run vs_chain_pass<.., Q = q, ..> for the (wave-uniform) run-time q in [0, W)
template <int KIND, bool SCRATCH, int W, int Q = 0>
struct vs_chain_dispatch {
    static __device__ __forceinline__ void run(int q, const double* F1, const double* F2, int i0, double& pa,
                                               double& pb, double (&cj)[VS_PASS / W], double (&dj)[VS_PASS / W]) {
        if (q == Q) vs_chain_pass<KIND, Q, SCRATCH, W>(F1, F2, i0, pa, pb, cj, dj);
        else vs_chain_dispatch<KIND, SCRATCH, W, Q + 1>::run(q, F1, F2, i0, pa, pb, cj, dj);
    }
};
template <int KIND, bool SCRATCH, int W>
struct vs_chain_dispatch<KIND, SCRATCH, W, W> {
    static __device__ __forceinline__ void run(int, const double*, const double*, int, double&, double&,
                                               double (&)[VS_PASS / W], double (&)[VS_PASS / W]) {}
};


#endif  // VS_DEVICE_GEN_H
