// sens_2 / sens_2n sums for 33 .. 64 index rows per half, one output, even row count, no NaN mask
// (varsens/saltelli.py:612-619): S S^T and T T^T of S = C + D, T = C - D on the fp64 matrix cores, operands brought in
// by LDS-DMA in whole 128-byte lines, every wave on its own -- no workgroup barrier in the loop (round 5).
//
// Why (tools/gram_direct.hip, profiles/r05/gram_direct.txt).  gram_st_kernel<4, ..> (vs_gram_st.hip) stood at
// 0.94-0.96 ms for k = 64, n = 2^22 -- 0.58 of HBM, matrix pipe busy 0.63 -- through three rounds of work on its
// pipeline; its eight waves share one tile per stage and meet at a barrier every 40 MFMAs, each holding five of the 20
// accumulator tiles, so every fragment is read from LDS by all of them.  Measured this round:
//  * a wave that keeps ALL 20 tiles (160 VGPRs) and is fed from registers runs the MFMA + add loop at 0.57-0.59 ms (the
//    pipe's 0.55 at 2.4 GHz): the matrix side has no problem at this shape;
//  * what a wave-wide 16-byte load delivers from the row-blocked f-values depends on whether it fetches WHOLE LINES:
//    16 planes x 64 B per instruction (the MFMA fragment layout itself, with or without neighbouring lanes contiguous)
//    streams at 4.1-4.2 TB/s, 8 planes x 128 B at 6.8, 4 planes x 256 B at 6.1-6.2 -- so the operands cannot go straight
//    into fragment registers; they have to be transposed on the way, and a register-staged transposition (load, add,
//    ds_write, ds_read) does not fit beside 160 accumulator registers;
//  * LDS-DMA (global_load_lds_dwordx4) needs no registers at all.  The LDS side of an LDS-DMA is lane-linear, so the
//    image is [c, d][64 index rows][16 sample rows] with 128-byte rows and NO padding; the fragment reads are made
//    conflict-free by storing the 16-byte chunk r of row u at chunk position r ^ ((u >> 1) & 7) -- applied on the
//    SOURCE address of the DMA.  A ds_read_b128 lane group is {i = 0-3, 12-15 at chunk q} + {i = 4-11 at chunk q + 1}
//    (MI355X_MICROARCH.md, LDS): with this swizzle its 16 lanes meet 16 different 16-byte slots of the bank row.
// The kernel: one workgroup of eight waves per CU (two per SIMD, 246 VGPRs, no spills), persistent.  A wave takes units
// of 16 sample rows (unit index = wave + 8 (workgroup + grid * visit): neighbouring waves read neighbouring lines), DMAs
// the unit into its PRIVATE 16 KiB image (sixteen pieces of 8 planes x 128 B), reads the fragments of 8 sample rows at a
// time (ds_read_b128: two MFMA steps per read), forms s = c + d, t = c - d from the fragments (8 v_add_f64 per step of
// 20 MFMAs) and starts the DMA of its next unit as soon as the last fragments of this one are in registers; its partner
// on the SIMD multiplies meanwhile.  Sums are formed in a fixed order (wave after wave through LDS at the end): the
// same bits run to run, and whatever the row-block size of the layout.
// Measured (k = 64, n = 2^22, prototype): 0.66-0.69 ms = 6.3-6.5 TB/s = 0.80 of HBM and 0.81 of the matrix pipe at
// once, shader clock 2.25-2.35 GHz; the loop without its DMA 0.59 ms, the DMA stream alone 0.63.
// Rows of the image past K are never written (their pieces are skipped; a piece that would straddle K is moved back to
// end at row K - 1): whatever they hold ends up in rows and columns >= K of the tiles, which the fold never reads.
// A shard whose row count is not a multiple of 16 ends in a partial unit: the wave whose turn it is loads it straight
// into the fragment layout with guarded loads (zeros past the last row).
#include "vs_second.h"

namespace {

// NBL live 16-row blocks per half: 3, 4 (K = 33 .. 64): images of 64 rows per half, eight waves per CU (two per SIMD);
// 2 (K = 17 .. 32): images of 32 rows per half, sixteen waves per CU (four per SIMD, 6 tiles per wave: HBM-bound)
constexpr int dma_rows(int nbl) { return nbl <= 2 ? 32 : 64; }          // image rows per half = SB of the partial
constexpr int dma_waves(int nbl) { return nbl <= 2 ? 16 : 8; }
// doubles per unit image: [c, d][rows][16], FIRST: + [a, b][16] (fM_1 and fM_2 of the unit's sample rows)
constexpr int dma_img(int nbl, bool first) { return 2 * dma_rows(nbl) * 16 + (first ? 32 : 0); }
constexpr size_t dma_lds_bytes(int nbl, bool first) { return sizeof(double) * dma_waves(nbl) * dma_img(nbl, first); }   // 128 KiB (+ 2 KiB)

typedef __attribute__((address_space(3))) void* vs_lds_ptr;
typedef const __attribute__((address_space(1))) void* vs_glb_ptr;

// NBL: live 16-row blocks per half (2: K = 17 .. 32, 3: 33 .. 48, 4: 49 .. 64); FULLK: K == 16 NBL with NBL even (no
// clamping of rows past K)
// FIRST (vs_reduce_first_second_order, 33 .. 64 index rows): the dot products a.c_u, a.d_u, b.c_u, b.d_u of the first-order
// record (a = fM_1, b = fM_2: varsens/saltelli.py:591-596) ride along.  They do not fit beside 160 accumulator registers
// in the MFMA fragment layout (16 more accumulators per lane); here LANE u OWNS INDEX ROW u: it reads its own row of c and
// of d from the image (the swizzle makes those reads conflict-free too: a ds_read_b128 lane group holds rows with eight
// different (u >> 1) & 7, each twice with opposite u & 1) and a, b of the unit from a 17th piece of the DMA (every lane
// the same address: a broadcast), 64 v_fma_f64 per unit into FOUR accumulators per lane, a chunk of two sample rows at a
// time between the MFMAs of the unit's first two steps (16 VGPRs in flight).  To make room the fragments of one half-unit
// live in registers at a time (32 instead of 64: the reads of the next half go out right after the adds that consumed the
// last one, ahead of that step's 20 MFMAs), and the DMA side's lane constants are rebuilt per unit.  The five scalar sums
// come from first_head_kernel, as for gram_st_kernel.
// ONE: one output (index row u is plane 2 + u for c_u, 2 + K + u for d_u: a wave-uniform plane offset per piece); otherwise
// index row u = j l + o lives in plane o (2k + 2) + 2 + j (+ k for d): the lane works its plane out per piece (u / l by a
// multiplication: exact for u < 64)
template <int NBL, bool FULLK, bool FIRST, bool ONE>
__global__ void __launch_bounds__(64 * dma_waves(NBL))
__attribute__((amdgpu_waves_per_eu(dma_waves(NBL) / 4, dma_waves(NBL) / 4)))
gram_dma_kernel(vs_gram_args A) {
    static_assert(NBL >= 2 && NBL <= 4, "17 .. 64 index rows per half");
    static_assert(!FULLK || NBL == 4 || NBL == 2, "FULLK means K == 64 or K == 32");
    static_assert(!FIRST || NBL >= 3, "up to 32 index rows the wave-private kernels of vs_second.hip carry the first-order sums");
    static_assert(ONE || !FIRST, "the first-order sums ride along for one output only");
    constexpr int NT = NBL * (NBL + 1) / 2;   // upper-triangle tiles per product
    constexpr int DMA_WAVES = dma_waves(NBL), DMA_IMG = dma_img(NBL, FIRST), DMA_SB = dma_rows(NBL);
    constexpr int SRC = DMA_SB * 16;          // doubles between the c rows and the d rows of an image
    constexpr int AB = 2 * SRC;               // FIRST: a at [AB, AB + 16), b at [AB + 16, AB + 32)
    if (FIRST && blockIdx.x == 0 && threadIdx.x == 0) {
        // this kernel opens the call: reset the counters the two folds use (cf. gram_st_kernel)
        uint32_t* c = const_cast<uint32_t*>(A.counters);
        c[0] = 0;
        c[1] = 0xffffffffu;
        c[2] = 0;
        c[3] = 0;
        c[4] = 0;
    }
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* const img = lds + wave * DMA_IMG;
    const int K = A.K;
    const uint64_t rows = A.rows;
    const uint64_t units = rows >> 4;
    uint32_t G = A.slab_wgs;
    asm volatile("" : "+s"(G));   // kept in a scalar register (cf. gram_st_kernel)
    const uint64_t stride = (uint64_t)G * DMA_WAVES;

    // DMA side: lane = 8 p + r' brings plane 8 g + p of piece g, source chunk r' ^ sw(row) to chunk position r'
    // (p and the chunk constants are rebuilt inside dma() from an opaque copy of the lane index: a handful of integer
    // instructions per unit instead of four registers carried through the MFMA steps)
    // fragment side: lane = i + 16 q reads chunk 4 h + q (sample rows 8 h + 2 q, + 1) of row 16 I + i
    const int li = lane & 15, q = lane >> 4, sw = (li >> 1) & 7;
    const uint32_t fr0 = (uint32_t)(li * 16 + 2 * (q ^ sw)), fr1 = (uint32_t)(li * 16 + 2 * ((4 + q) ^ sw));

    vs_d4 accS[NT], accT[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) accS[t] = accT[t] = (vs_d4){0.0, 0.0, 0.0, 0.0};

    // Source addresses: per unit one per-lane base (plane p of a piece), per piece a wave-uniform plane offset (scalar
    // arithmetic) and the lane's chunk.  K not a multiple of 8: the piece that would straddle K brings rows K - 8 .. K - 1
    // instead (into their own places in the image, over what the piece before it wrote there: the same values) -- no lane
    // ever reads a plane past d_{K-1}, which in the last row block would be past the end of the buffer.
    const uint32_t ldiv = (65536u + (uint32_t)A.l - 1u) / (uint32_t)A.l;   // (u * ldiv) >> 16 == u / l for u < 64
    auto dma = [&](uint64_t unit) {
        VS2_FTILE(unit << 4);
        int ln = lane;
        asm volatile("" : "+v"(ln));   // not loop-invariant as far as the compiler can tell
        const int p = ln >> 3;
        const uint32_t chA = (uint32_t)((ln & 7) ^ (p >> 1)), chB = chA ^ 4u;   // pieces with g even / odd
        const uint32_t chC = (uint32_t)((ln & 7) ^ (((K - 8 + p) >> 1) & 7));   // the piece that ends at row K - 1
        const double* const lb = fb + (uint64_t)p * fp;
        // several outputs: the lane's planes are worked out per piece and unit -- left loop-invariant they would live in
        // sixteen registers between the units (an opaque copy of p keeps the optimiser from hoisting them)
        const int pv = p;
#pragma unroll
        for (int src = 0; src < 2; ++src)
#pragma unroll
            for (int g = 0; g < 2 * NBL; ++g) {
                if (!FULLK && 8 * g >= K) continue;              // wave-uniform: the piece is all padding
                const bool part = !FULLK && 8 * g + 8 > K;       // wave-uniform: the piece would straddle K
                const int urow = part ? K - 8 : 8 * g;
                const uint32_t ch = part ? chC : (g & 1) ? chB : chA;
                const double* gp;
                if (ONE) {
                    gp = lb + (uint64_t)(2 + src * K + urow) * fp + 2 * ch;
                } else {
                    const uint32_t uu = (uint32_t)(urow + pv);         // < K: no piece reaches past row K - 1
                    const uint32_t jj = (uu * ldiv) >> 16, oo = uu - jj * (uint32_t)A.l;
                    gp = fb + (uint64_t)(oo * (2u * (uint32_t)A.k + 2u) + 2u + jj + (uint32_t)(src * A.k)) * fp + 2 * ch;
                }
                __builtin_amdgcn_global_load_lds((vs_glb_ptr)gp, (vs_lds_ptr)(img + src * SRC + urow * 16), 16, 0, 2 /* nt */);
            }
        if (FIRST && ln < 16) {   // planes 0 (a) and 1 (b): lanes 0-7 and 8-15, chunk lane & 7
            const double* gp = fb + (uint64_t)(ln >> 3) * fp + 2 * (ln & 7);
            __builtin_amdgcn_global_load_lds((vs_glb_ptr)gp, (vs_lds_ptr)(img + AB), 16, 0, 2 /* nt */);
        }
    };
#define DMA_PIN() __builtin_amdgcn_sched_barrier(0)
    // The compiler does not know that an LDS-DMA writes what the fragment reads read: both waits are explicit, and the
    // "memory" clobber keeps the reads and the DMAs on their side of them.
#define DMA_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define DMA_READS_DONE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

    uint64_t u = (uint64_t)blockIdx.x * DMA_WAVES + wave;
    // FIRST: the row-owner sums of lane u = lane
    double ac = 0.0, ad = 0.0, bc = 0.0, bd = 0.0;
    {
        const int rsw = (lane >> 1) & 7;
        const double* const own = img + lane * 16;
        // chunk r (sample rows 2 r, 2 r + 1 of the unit) of this lane's rows of c and d, and of a and b
        struct ro_regs {
            vs_d2 c, d, a, b;
        };
        auto ro_read = [&](const int r, ro_regs& R) {
            const double* pc = own + 2 * (r ^ rsw);
            R.c = *(const vs_d2*)pc;
            R.d = *(const vs_d2*)(pc + SRC);
            R.a = *(const vs_d2*)(img + AB + 2 * r);
            R.b = *(const vs_d2*)(img + AB + 16 + 2 * r);
        };
        auto ro_fma = [&](const ro_regs& R) {
            ac = fma(R.a[0], R.c[0], ac);
            ad = fma(R.a[0], R.d[0], ad);
            bc = fma(R.b[0], R.c[0], bc);
            bd = fma(R.b[0], R.d[0], bd);
            ac = fma(R.a[1], R.c[1], ac);
            ad = fma(R.a[1], R.d[1], ad);
            bc = fma(R.b[1], R.c[1], bc);
            bd = fma(R.b[1], R.d[1], bd);
            // sched_barrier pins the machine schedule only: left alone the optimiser sinks all 64 fmas of a unit behind its
            // last read and keeps eight chunks alive (292 spilled registers).  The sums pass through here, so the fmas of
            // this chunk are done before the next chunk is read; the clobber keeps the reads on their side.
            asm volatile("" : "+v"(ac), "+v"(ad), "+v"(bc), "+v"(bd) : : "memory");
        };
        vs_d2 H[2][NBL];   // the fragments of ONE half-unit: [c, d][block]
        auto readh = [&](const int h) {
            const double* rd = img + (h ? fr1 : fr0);
#pragma unroll
            for (int src = 0; src < 2; ++src)
#pragma unroll
                for (int I = 0; I < NBL; ++I) H[src][I] = *(const vs_d2*)(rd + src * SRC + I * 256);
        };
        // one MFMA step on element e of H.  act: what follows the adds (which are the last readers of element 1):
        // 1: the fragments of the unit's second half, 2: wait for the next unit's DMA, then its first half.
        // r0 >= 0: the row-owner chunks r0 .. r0 + 3, a chunk's reads NT / 4 tile pairs ahead of its 8 fmas.
        auto step1 = [&](const int e, const int act, const int r0) {
            double s[NBL], t[NBL];
#pragma unroll
            for (int I = 0; I < NBL; ++I) {
                const double c = H[0][I][e], d = H[1][I][e];
                s[I] = c + d;
                t[I] = c - d;
            }
            DMA_PIN();
            if (act == 1) readh(1);
            if (act == 2) {
                DMA_LANDED();
                readh(0);
            }
            ro_regs R;
            if (r0 >= 0) ro_read(r0, R);
            DMA_PIN();
            int idx = 0, chunk = 0;
#pragma unroll
            for (int I = 0; I < NBL; ++I)
#pragma unroll
                for (int J = I; J < NBL; ++J) {
                    accS[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(s[I], s[J], accS[idx], 0, 0, 0);
                    accT[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(t[I], t[J], accT[idx], 0, 0, 0);
                    ++idx;
                    if (r0 >= 0 && chunk < 3 && idx == (NT * (chunk + 1)) / 4) {
                        DMA_PIN();
                        ro_fma(R);
                        ++chunk;
                        ro_read(r0 + chunk, R);
                        DMA_PIN();
                    }
                }
            if (r0 >= 0) {
                DMA_PIN();
                ro_fma(R);
            }
        };
        if (u < units) {
            dma(u);
            DMA_LANDED();
            readh(0);
        }
        for (; u < units; u += stride) {
            const uint64_t un = u + stride < units ? u + stride : u;
            DMA_PIN();
            step1(0, 0, FIRST ? 0 : -1);
            DMA_PIN();
            step1(1, 1, FIRST ? 4 : -1);
            DMA_PIN();
            DMA_READS_DONE();   // fragments and row-owner chunks of this unit are in registers: the image is free
            dma(un);
            step1(0, 0, -1);
            DMA_PIN();
            step1(1, 2, -1);
            DMA_PIN();
        }
        if ((rows & 15) != 0 && u == units) {
            DMA_READS_DONE();
            VS2_FTILE(units << 4);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint64_t r = (units << 4) + 2 * (4 * h + q);
                const bool in = r < rows;
#pragma unroll
                for (int src = 0; src < 2; ++src)
#pragma unroll
                    for (int I = 0; I < NBL; ++I) {
                        const uint32_t uu = (uint32_t)(16 * I + li < K ? 16 * I + li : K - 1);
                        const uint32_t jj = (uu * ldiv) >> 16, oo = uu - jj * (uint32_t)A.l;   // one output: jj = uu, oo = 0
                        const double* gp = fb + (uint64_t)(oo * (2u * (uint32_t)A.k + 2u) + 2u + jj + (uint32_t)(src * A.k)) * fp +
                                           2 * (4 * h + q);
                        H[src][I] = in ? __builtin_nontemporal_load((const vs_d2*)gp) : (vs_d2){0.0, 0.0};
                    }
                step1(0, 0, -1);
                step1(1, 0, -1);
            }
            // the row-owner sums of the partial unit, straight from memory
            const int uo = lane < K ? lane : K - 1;
            for (int r = 0; FIRST && r < 8; ++r) {
                if ((units << 4) + 2 * r >= rows) break;
                ro_regs R;
                R.c = *(const vs_d2*)(fb + (uint64_t)(2 + uo) * fp + 2 * r);
                R.d = *(const vs_d2*)(fb + (uint64_t)(2 + K + uo) * fp + 2 * r);
                R.a = *(const vs_d2*)(fb + 2 * r);
                R.b = *(const vs_d2*)(fb + fp + 2 * r);
                ro_fma(R);
            }
        }
    }
#undef DMA_PIN
#undef DMA_LANDED
#undef DMA_READS_DONE

    // the workgroup's sum in the partial's own layout [S S^T, T T^T][64][64], wave after wave
    __syncthreads();
    double* red = lds;
    int lane_e = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_e));   // rebuilt here: nothing of it is carried through the loop
    for (int w = 0; w < DMA_WAVES; ++w) {
        if (wave == w) {
            int idx = 0;
#pragma unroll
            for (int I = 0; I < NBL; ++I)
#pragma unroll
                for (int J = I; J < NBL; ++J) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int row = 16 * I + (lane_e >> 4) + 4 * v, col = 16 * J + (lane_e & 15);   // f64 C/D layout
                        double* ps = red + row * DMA_SB + col;
                        double* pt = ps + DMA_SB * DMA_SB;
                        *ps = (w ? *ps : 0.0) + accS[idx][v];
                        *pt = (w ? *pt : 0.0) + accT[idx][v];
                    }
                    ++idx;
                }
            if (FIRST) {   // [a.c, b.d, a.d, b.c][64] behind the tiles: the order of the record's slots
                double* fs = red + 2 * DMA_SB * DMA_SB + lane_e;
                fs[0] = (w ? fs[0] : 0.0) + ac;
                fs[64] = (w ? fs[64] : 0.0) + bd;
                fs[128] = (w ? fs[128] : 0.0) + ad;
                fs[192] = (w ? fs[192] : 0.0) + bc;
            }
        }
        __syncthreads();
    }
    if (FIRST) {
        // one record per workgroup: head (zeros: first_head_kernel's records carry the five scalar sums), ac[k], bd[k], ad[k], bc[k]
        const int k = A.k;
        const size_t plen = VS_PLEN(k);
        double* rec = A.first_partials + (size_t)blockIdx.x * plen;
        const double* fs = red + 2 * DMA_SB * DMA_SB;
        for (size_t e = threadIdx.x; e < plen; e += 64 * DMA_WAVES) {
            double v = 0.0;
            if (e >= VS_P_HEAD) {
                const int slot = (int)((e - VS_P_HEAD) / k), j = (int)((e - VS_P_HEAD) % k);
                v = fs[slot * 64 + j];
            }
            rec[e] = v;
        }
    }
    // upper 16-blocks only: the fold never reads the mirror blocks
    double* out = A.partials + (size_t)blockIdx.x * 2 * DMA_SB * DMA_SB;
    for (int e = threadIdx.x; e < 2 * DMA_SB * DMA_SB; e += 64 * DMA_WAVES) {
        const int rr = (e / DMA_SB) % DMA_SB, cc = e % DMA_SB;
        if ((cc >> 4) >= (rr >> 4) && (rr >> 4) < NBL && (cc >> 4) < NBL) out[e] = red[e];   // (NBL = 3: SB = 64)
    }
}

}  // namespace

#define VS_DMA_INSTANCES(X) \
    X(2, false, false, true) X(2, true, false, true) X(3, false, false, true) X(4, false, false, true) X(4, true, false, true) \
    X(3, false, true, true) X(4, false, true, true) X(4, true, true, true) \
    X(3, false, false, false) X(4, false, false, false) X(4, true, false, false)

int vs_gram_dma_prepare(vs_context* ctx) {
    (void)ctx;
#define VS_DMA_SET(NL, FK, FI, ON)                                                                                 \
    VS_HIP(hipFuncSetAttribute((const void*)gram_dma_kernel<NL, FK, FI, ON>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)dma_lds_bytes(NL, FI)));
    VS_DMA_INSTANCES(VS_DMA_SET)
#undef VS_DMA_SET
    return VS_OK;
}

bool vs_gram_dma_takes_first(int K) { return K > 32 && K <= 64; }

// A.K = k l in 17 .. 64 (several outputs: 33 .. 64), A.rows even, A.f 16-byte aligned (the caller checks); `workgroups` <= CUs.
// Partials: one per workgroup, [S S^T, T T^T][SB][SB] with SB = 64 (K > 32) or 32, upper 16-blocks
// (vs_gram_fold_args::st == 2, super == SB).
// with_first (K > 32): also one first-order record per workgroup at A.first_partials (dot products; head zero).
int vs_gram_dma_launch(vs_context* ctx, const vs_gram_args& A, bool with_first, unsigned workgroups) {
    VS_REQUIRE(A.K > 16 && A.K <= 64 && A.l >= 1 && A.K == A.k * A.l, "gram_dma_kernel takes 17 .. 64 index rows");
    VS_REQUIRE(A.l == 1 || (A.K > 32 && !with_first), "several outputs: 33 .. 64 index rows, second order alone");
    VS_REQUIRE(!with_first || (vs_gram_dma_takes_first(A.K) && A.first_partials),
               "gram_dma_kernel carries the first-order sums at 33 .. 64 index rows");
    VS_REQUIRE(A.rows % 2 == 0 && ((uintptr_t)A.f & 15) == 0, "gram_dma_kernel needs an even row count and 16-byte alignment");
    VS_REQUIRE(A.slab_wgs == workgroups && workgroups >= 1, "slab_wgs must be the grid size");
    const dim3 grid(workgroups);
    const int nbl = (A.K + 15) / 16;
    const bool full = A.K == 64 || A.K == 32;
    const bool one = A.l == 1;
    bool launched = false;
#define VS_DMA_LAUNCH(NL, FK, FI, ON)                                                                          \
    if (!launched && nbl == NL && full == FK && with_first == FI && one == ON) {                               \
        hipLaunchKernelGGL((gram_dma_kernel<NL, FK, FI, ON>), grid, dim3(64 * dma_waves(NL)), dma_lds_bytes(NL, FI), ctx->stream, A); \
        launched = true;                                                                                       \
    }
    VS_DMA_INSTANCES(VS_DMA_LAUNCH)
#undef VS_DMA_LAUNCH
    if (!launched) return vs_fail(VS_ERR_STATE, "no gram_dma_kernel instance for K=%d", A.K);
    VS_HIP(hipGetLastError());
    return VS_OK;
}
