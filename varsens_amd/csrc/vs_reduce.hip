// First-order Monte Carlo reductions of Varsens.compute_varsens
// (varsens/saltelli.py:572-609) as one streaming pass over the f-values.
//
// With a = fM_1[r], b = fM_2[r], c_j = fN_j[j][r], d_j = fN_nj[j][r]:
//   sab = sum a*b                      -> E_2   (:577)
//   sa, sb, s1 = sum (x-s), s2 = sum (x-s)^2 over the 2n values [a;b] -> var_y (:583)
//   ac_j, bd_j, ad_j, bc_j             -> U_j, U_nj (:591-596)
// Every f-value is read exactly once: 8 bytes per sample.
//
// Kernel shape (l = 1, the streaming fast path):
//   * a workgroup of 256 threads owns tiles of 256*RPT rows; each thread keeps
//     the a and b values of its RPT rows in registers (double2 loads, 16 B per
//     lane, a wave covers 1 KiB contiguous per load);
//   * it then streams the 2k blocks c_j, d_j of the tile block by block with
//     the same coalesced loads, software-prefetching block j+1 while the 4*RPT
//     fmas of block j run;
//   * the four per-thread partial dot products of block j are summed across
//     the wave with DPP row operations (no LDS traffic), and lane 63 adds them
//     to the wave's private row of an LDS accumulator -- LDS-staged partial
//     sums, fixed order, no atomics, so results are bit-reproducible;
//   * at the end the four wave rows are added and one partial record per
//     workgroup goes to HBM; a small fold kernel sums the records with
//     compensated addition.
// f-values are stored output-major ("planar", see varsens_hip.h): output o is its own
// [2k+2][rows] array, so l > 1 is l independent runs of the same streaming code
// (blockIdx.x = tile workgroup * l + output).  An odd row count (16-byte loads would
// straddle blocks) runs the same code with 8-byte loads.
#include "vs_common.h"

#define VS_WG 256
#define VS_WAVES (VS_WG / 64)
#define VS_JCHUNK 512  // blocks j handled per workgroup (bounds the LDS accumulator)

enum { CNT_NANFLAG = 0, CNT_FIRST_KEPT = 1, CNT_KEPT = 2 };

// ---------------------------------------------------------------------------
// the streaming kernel
// ---------------------------------------------------------------------------

struct vs_reduce_args {
    const double* f;
    uint64_t rows;      // rows of this shard
    int k, l;
    int jchunk;         // blocks per blockIdx.y
    uint32_t tile_wgs;  // workgroups per output (grid.x = tile_wgs * l)
    double* partials;   // [tile_wgs][l][VS_PLEN(k)]
    const uint8_t* keep;      // MASKED: 1 = row kept
    const uint32_t* counters; // MASKED: early exit + shift row
    int f_shift;        // log2 of the rows per row block of the f-value layout (vs_f_tile)
};

template <int VEC>
struct vs_vec;
template <>
struct vs_vec<2> {
    double v[2];
    __device__ __forceinline__ void load(const double* p) {
        double2 t;  // read once: non-temporal (one global_load_dwordx4 ... nt)
        t.x = __builtin_nontemporal_load(p);
        t.y = __builtin_nontemporal_load(p + 1);
        v[0] = t.x;
        v[1] = t.y;
    }
};
template <>
struct vs_vec<1> {
    double v[1];
    __device__ __forceinline__ void load(const double* p) { v[0] = *p; }
};

// VEC = 2: rows even (16-byte loads); VEC = 1: 8-byte loads.
template <int VEC, int NV, bool MASKED>
__global__ void __launch_bounds__(VS_WG)
reduce_first_kernel(vs_reduce_args A) {
    extern __shared__ double wacc[];  // [VS_WAVES][plen_local]
    if (MASKED && A.counters[CNT_NANFLAG] == 0) return;
    if (!MASKED && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        // the un-masked pass opens every call: it resets the counters the LATER kernels of this
        // call use (fold ticket, NaN flag, keep statistics) -- no host copy on the stream
        uint32_t* c = const_cast<uint32_t*>(A.counters);
        c[CNT_NANFLAG] = 0;
        c[CNT_FIRST_KEPT] = 0xffffffffu;
        c[CNT_KEPT] = 0;
        c[3] = 0;  // CNT_TICKET (un-masked fold)
        c[4] = 0;  // CNT_TICKET + 1 (masked fold)
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int k = A.k;
    const int l = A.l;
    const uint32_t o = blockIdx.x % (uint32_t)l;
    const uint32_t wg = blockIdx.x / (uint32_t)l;
    const int j0 = blockIdx.y * A.jchunk;
    const int j1 = min(k, j0 + A.jchunk);
    const int nj = j1 - j0;
    const bool head = (blockIdx.y == 0);
    const size_t plen_local = VS_P_HEAD + 4 * (size_t)nj;
    double* my = wacc + (size_t)wave * plen_local;

    for (size_t i = tid; i < plen_local * VS_WAVES; i += VS_WG) wacc[i] = 0.0;
    __syncthreads();

    const uint64_t rows = A.rows;
    // row-blocked f-values (vs_f_tile): a tile of this kernel (at most 2048 rows, aligned) lies inside one row block, so
    // per tile there is ONE wave-uniform (base, pitch); plane (o, block) of the tile's rows starts at
    // f + base + (o (2k+2) + block) pitch, rows consecutive behind it
    const uint64_t planes = (uint64_t)l * (2 * (uint64_t)k + 2), oplane = (uint64_t)o * (2 * (uint64_t)k + 2);
    // common shift of the variance sums: a leading-sample mean of fM_1 (un-masked pass: the first min(rows, 1024) rows
    // of plane (o, 0) are contiguous in the first row block), or the first kept a-value (masked pass: the leading
    // values may hold the NaNs)
    const uint64_t shift_row = MASKED ? (uint64_t)A.counters[CNT_FIRST_KEPT] : 0;
    const double shift = MASKED ? ((shift_row < rows) ? A.f[vs_f_index(rows, planes, A.f_shift, oplane, shift_row)] : 0.0)
                                : vs_shift_estimate(A.f + vs_f_index(rows, planes, A.f_shift, oplane, 0), rows);

    constexpr uint64_t TILE = (uint64_t)VS_WG * NV * VEC;
    const uint64_t ntiles = (rows + TILE - 1) / TILE;

    double t_sab = 0, t_sa = 0, t_sb = 0, t_s1 = 0, t_s2 = 0;

    for (uint64_t t = wg; t < ntiles; t += A.tile_wgs) {
        const uint64_t tile = t;
        const uint64_t base = tile * TILE + (uint64_t)tid * VEC;
        const vs_ftile ft = vs_f_tile(rows, planes, A.f_shift, tile * TILE);
        const double* fo = A.f + ft.base + oplane * ft.pitch;   // plane (o, 0), first row of the tile
        const uint64_t fpitch = ft.pitch;
        bool ok[NV];
        double keepf[NV][VEC];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const uint64_t r = base + (uint64_t)v * VS_WG * VEC;
            ok[v] = r < rows;  // rows is even when VEC == 2, so a pair is in or out as a whole
#pragma unroll
            for (int e = 0; e < VEC; ++e) keepf[v][e] = 1.0;
            if (MASKED && ok[v]) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) keepf[v][e] = A.keep[r + e] ? 1.0 : 0.0;
            }
        }
        auto load_block = [&](vs_vec<VEC>* dst, uint64_t block) {
            const double* p = fo + block * fpitch + (uint64_t)tid * VEC;   // wave-uniform plane base + this thread's rows
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (ok[v]) dst[v].load(p + (uint64_t)v * VS_WG * VEC);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) dst[v].v[e] = 0.0;
                }
                if (MASKED) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (keepf[v][e] == 0.0) dst[v].v[e] = 0.0;  // also removes the NaN itself
                }
            }
        };

        vs_vec<VEC> a[NV], b[NV];
        load_block(a, 0);
        load_block(b, 1);
        vs_vec<VEC> cn[NV], dn[NV];
        if (nj > 0) {
            load_block(cn, 2 + (uint64_t)j0);
            load_block(dn, 2 + (uint64_t)k + j0);
        }
        if (head) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const double x = a[v].v[e], y = b[v].v[e];
                    const double live = ok[v] ? keepf[v][e] : 0.0;
                    const double dx = live != 0.0 ? x - shift : 0.0;
                    const double dy = live != 0.0 ? y - shift : 0.0;
                    t_sab = fma(x, y, t_sab);
                    t_sa += x;
                    t_sb += y;
                    t_s1 += dx + dy;
                    t_s2 = fma(dx, dx, t_s2);
                    t_s2 = fma(dy, dy, t_s2);
                }
            }
        }
        for (int j = 0; j < nj; ++j) {
            vs_vec<VEC> c[NV], d[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                c[v] = cn[v];
                d[v] = dn[v];
            }
            if (j + 1 < nj) {
                load_block(cn, 2 + (uint64_t)(j0 + j + 1));
                load_block(dn, 2 + (uint64_t)k + (j0 + j + 1));
            }
            double ac = 0, bd = 0, ad = 0, bc = 0;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    ac = fma(a[v].v[e], c[v].v[e], ac);
                    bd = fma(b[v].v[e], d[v].v[e], bd);
                    ad = fma(a[v].v[e], d[v].v[e], ad);
                    bc = fma(b[v].v[e], c[v].v[e], bc);
                }
            }
            ac = vs_wave_sum_dpp(ac);
            bd = vs_wave_sum_dpp(bd);
            ad = vs_wave_sum_dpp(ad);
            bc = vs_wave_sum_dpp(bc);
            if (lane == 63) {
                double* q = my + VS_P_HEAD + 4 * (size_t)j;
                q[0] += ac;
                q[1] += bd;
                q[2] += ad;
                q[3] += bc;
            }
        }
    }
    if (head) {
        t_sab = vs_wave_sum_dpp(t_sab);
        t_sa = vs_wave_sum_dpp(t_sa);
        t_sb = vs_wave_sum_dpp(t_sb);
        t_s1 = vs_wave_sum_dpp(t_s1);
        t_s2 = vs_wave_sum_dpp(t_s2);
        if (lane == 63) {
            my[VS_P_SAB] = t_sab;
            my[VS_P_SA] = t_sa;
            my[VS_P_SB] = t_sb;
            my[VS_P_S1] = t_s1;
            my[VS_P_S2] = t_s2;
        }
    }
    __syncthreads();
    // add the wave rows in fixed order and publish this workgroup's record
    double* out = A.partials + ((size_t)wg * l + o) * VS_PLEN(k);
    for (size_t i = tid; i < plen_local; i += VS_WG) {
        double s = wacc[i];
#pragma unroll
        for (int w = 1; w < VS_WAVES; ++w) s += wacc[(size_t)w * plen_local + i];
        if (i < VS_P_HEAD) {
            if (head) out[i] = s;
        } else {
            const size_t jj = (i - VS_P_HEAD) >> 2, q = (i - VS_P_HEAD) & 3;
            // partial layout: head, then ac[k], bd[k], ad[k], bc[k]
            out[VS_P_HEAD + q * (size_t)k + (size_t)j0 + jj] = s;
        }
    }
}

// ---------------------------------------------------------------------------
// NaN-trim keep mask (saltelli.py:474-486): row r is dropped when any of the
// 2k+2 blocks holds a NaN at r in output column 0
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
nan_mask_kernel(const double* __restrict__ f, uint64_t rows, int k, int l, int f_shift, uint8_t* __restrict__ keep,
                uint32_t* counters) {
    if (counters[CNT_NANFLAG] == 0) return;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t kept = 0;
    uint32_t first = 0xffffffffu;
    const uint64_t planes = (uint64_t)l * (2 * (uint64_t)k + 2);
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        bool bad = false;
        const vs_ftile ft = vs_f_tile(rows, planes, f_shift, r);   // row r's own (base, pitch): consecutive rows coalesce
        for (int b = 0; b < 2 * k + 2; ++b) {
            double x = f[ft.base + (uint64_t)b * ft.pitch];  // planes of output 0 (saltelli.py:482)
            bad |= (x != x);
        }
        keep[r] = bad ? 0 : 1;
        if (!bad) {
            ++kept;
            first = min(first, (uint32_t)r);
        }
    }
    // wave-level combine, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        kept += __shfl_xor(kept, off, 64);
        first = min(first, (uint32_t)__shfl_xor((int)first, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counters[CNT_KEPT], kept);
        atomicMin(&counters[CNT_FIRST_KEPT], first);
    }
}

// ---------------------------------------------------------------------------
// fold: sum the workgroup records (compensated), build this rank's record
// ---------------------------------------------------------------------------
struct vs_fold_args {
    const double* partials;  // [P][l][plen]
    uint32_t P;
    int k, l;
    uint64_t rows;
    const double* f;    // for the shift value (block 0, row shift_row of each output plane) ...
    const double* shift_values;  // ... or, when non-null, explicit shift values (fused path: no f-values exist):
    int shift_stride;            //     output o's is shift_values[o * shift_stride]
    double* acc;        // record (device)
    int rank, max_ranks;
    uint32_t* counters;
    int masked;
    double* seg_sums;   // grid.y > 1: [grid.y][l*plen] sums of the segments of P (two-stage fold)
    int f_shift;        // f-value layout (vs_f_tile), for the shift values read from f
};

#define VS_FOLD_THREADS 1024
#define VS_FOLD_COLS 64

// grid.x = ceil(l*plen / 64): block b folds 64 consecutive columns of the
// flattened (l*plen) record with 16 waves striding over P.  The block that
// finishes last (agent-scope ticket) lays the folded sums out as the rank's
// accumulator record, so the whole fold is one launch.
// grid.y = segments: with thousands of records (the fused kernels leave one per workgroup, twice the resident
// set) three blocks walking all of them are latency-bound (0.039 ms for 2048 records of 133 doubles); block
// (b, s) folds segment s of the records, the last block adds the segment sums in segment order (compensated,
// like everything here) -- fixed order, bit-reproducible.  grid.y = 1 is the single-stage fold of rounds 1-2.
enum { CNT_TICKET = 3 };

__global__ void __launch_bounds__(VS_FOLD_THREADS)
fold_first_kernel(vs_fold_args A, double* folded /* [l*plen] scratch */) {
    __shared__ double s_sum[VS_FOLD_THREADS / 64][VS_FOLD_COLS];
    __shared__ double s_cmp[VS_FOLD_THREADS / 64][VS_FOLD_COLS];
    __shared__ uint32_t s_last;
    if (A.masked && A.counters[CNT_NANFLAG] == 0) return;
    const size_t width = (size_t)A.l * VS_PLEN(A.k);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t col = (size_t)blockIdx.x * VS_FOLD_COLS + lane;
    double s = 0.0, comp = 0.0;
    const uint32_t nseg = gridDim.y, per_seg = (A.P + nseg - 1) / nseg;
    const uint32_t p_lo = blockIdx.y * per_seg, p_hi = p_lo + per_seg < A.P ? p_lo + per_seg : A.P;
    if (col < width) {
        constexpr uint32_t STEP = VS_FOLD_THREADS / 64, BATCH = 8;
        for (uint32_t p0 = p_lo + wave; p0 < p_hi; p0 += STEP * BATCH) {
            double x[BATCH];  // issue the loads of a batch together: the fold is latency-, not bandwidth-bound
#pragma unroll
            for (uint32_t i = 0; i < BATCH; ++i) {
                const uint32_t p = p0 + i * STEP;
                x[i] = p < p_hi ? A.partials[(size_t)p * width + col] : 0.0;
            }
#pragma unroll
            for (uint32_t i = 0; i < BATCH; ++i) {
                const double t = s + x[i];  // Neumaier compensated sum
                comp += (fabs(s) >= fabs(x[i])) ? (s - t) + x[i] : (x[i] - t) + s;
                s = t;
            }
        }
    }
    s_sum[wave][lane] = s;
    s_cmp[wave][lane] = comp;
    __syncthreads();
    if (wave == 0 && col < width) {
        double S = 0.0, Cc = 0.0;
        for (int w = 0; w < VS_FOLD_THREADS / 64; ++w) {
            const double x = s_sum[w][lane];
            const double t = S + x;
            Cc += (fabs(S) >= fabs(x)) ? (S - t) + x : (x - t) + S;
            S = t;
            Cc += s_cmp[w][lane];
        }
        const double total = S + Cc;
        if (nseg == 1) {
            __hip_atomic_store(&folded[col], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!A.masked && total != total) atomicOr(&A.counters[CNT_NANFLAG], 1u);
        } else {
            __hip_atomic_store(&A.seg_sums[(size_t)blockIdx.y * width + col], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // publish, then take a ticket; the last block builds the record
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = atomicAdd(&A.counters[CNT_TICKET + (A.masked ? 1 : 0)], 1u);
        s_last = (t == gridDim.x * gridDim.y - 1) ? 1u : 0u;
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (!s_last) return;
    if (nseg > 1) {   // second stage: the segment sums of every column, in segment order
        for (size_t c = threadIdx.x; c < width; c += VS_FOLD_THREADS) {
            double S = 0.0, Cc = 0.0;
            for (uint32_t sg = 0; sg < nseg; ++sg) {
                const double x = __hip_atomic_load(&A.seg_sums[(size_t)sg * width + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double t = S + x;
                Cc += (fabs(S) >= fabs(x)) ? (S - t) + x : (x - t) + S;
                S = t;
            }
            const double total = S + Cc;
            __hip_atomic_store(&folded[c], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!A.masked && total != total) atomicOr(&A.counters[CNT_NANFLAG], 1u);
        }
        __syncthreads();
    }

    const int k = A.k, l = A.l;
    const size_t plen = VS_PLEN(k);
    const size_t per_out = VS_ACC_PER_OUTPUT(k);
    const size_t len = VS_ACC_LEN(k, l, A.max_ranks);
    const uint64_t kept = A.masked ? (uint64_t)A.counters[CNT_KEPT] : A.rows;
    const uint64_t shift_row = A.masked ? (uint64_t)A.counters[CNT_FIRST_KEPT] : 0;
    auto ld = [&](size_t i) { return __hip_atomic_load(&folded[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    double* slot0 = A.acc + VS_ACC_HEADER + (size_t)l * per_out;
    double* slot = slot0 + (size_t)A.rank * VS_ACC_SLOT(l);
    const size_t slot_lo = (size_t)(slot - A.acc), slot_hi = slot_lo + VS_ACC_SLOT(l);
    for (size_t i = threadIdx.x; i < len; i += blockDim.x) {
        if (i >= slot_lo && i < slot_hi) continue;  // written below
        double v = 0.0;
        if (i == 0) v = (double)kept;
        else if (i == 1) v = (double)(A.rows - kept);
        else if (i == 2) v = 1.0;
        else if (i >= VS_ACC_HEADER && i < VS_ACC_HEADER + (size_t)l * per_out) {
            const size_t o = (i - VS_ACC_HEADER) / per_out, c = (i - VS_ACC_HEADER) % per_out;
            // record: sab, sa, sb, ac.. ; partial: sab, sa, sb, s1, s2, ac..
            v = ld(o * plen + (c < 3 ? c : c + 2));
        }
        A.acc[i] = v;
    }
    // this rank's variance slot: count, mean[l], m2[l]
    const double cnt = 2.0 * (double)kept;
    if (threadIdx.x == 0) slot[0] = cnt;
    for (int o = 0; o < l; ++o) {  // every thread walks the outputs: vs_shift_estimate is a wave-wide call
        double mean = 0.0, m2 = 0.0;
        if (kept > 0) {
            // plane (o, fM_1) of the row-blocked f-values: its first min(rows, 1024) rows are contiguous (vs_f_tile)
            const uint64_t planes = (uint64_t)l * (2 * (uint64_t)k + 2), oplane = (uint64_t)o * (2 * (uint64_t)k + 2);
            const double shift = A.shift_values ? A.shift_values[(size_t)o * A.shift_stride]
                                 : (A.masked ? (shift_row < A.rows ? A.f[vs_f_index(A.rows, planes, A.f_shift, oplane, shift_row)] : 0.0)
                                             : vs_shift_estimate(A.f + vs_f_index(A.rows, planes, A.f_shift, oplane, 0), A.rows));
            const double s1 = ld(o * plen + VS_P_S1), s2 = ld(o * plen + VS_P_S2);
            mean = shift + s1 / cnt;
            m2 = s2 - s1 * (s1 / cnt);
        }
        if (threadIdx.x == 0) {
            slot[1 + o] = mean;
            slot[1 + l + o] = m2;
        }
    }
}

// ---------------------------------------------------------------------------
// finish: E_2, var_y, U_j, U_nj, sens, sens_t from a summed record
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
finish_first_kernel(const double* __restrict__ acc, uint64_t n, int k, int l, int max_ranks,
                    double* __restrict__ out) {
    const size_t per_out = VS_ACC_PER_OUTPUT(k);
    const double* slots = acc + VS_ACC_HEADER + (size_t)l * per_out;
    const size_t slot_len = VS_ACC_SLOT(l);
    const double dn = (double)n, dn1 = (double)n - 1.0;
    double* E_2 = out;
    double* var_y = out + l;
    double* U_j = out + 2 * (size_t)l;
    double* U_nj = U_j + (size_t)k * l;
    double* sens = U_nj + (size_t)k * l;
    double* sens_t = sens + (size_t)k * l;
    for (int o = 0; o < l; ++o) {
        // merge the per-rank (count, mean, m2) triples: exact pooled variance
        double cnt = 0.0, wsum = 0.0;
        for (int g = 0; g < max_ranks; ++g) {
            const double c = slots[g * slot_len];
            cnt += c;
            wsum += c * slots[g * slot_len + 1 + o];
        }
        const double mu = cnt > 0.0 ? wsum / cnt : 0.0;
        double m2 = 0.0;
        for (int g = 0; g < max_ranks; ++g) {
            const double c = slots[g * slot_len];
            const double dm = slots[g * slot_len + 1 + o] - mu;
            m2 += slots[g * slot_len + 1 + l + o] + c * dm * dm;
        }
        const double* rec = acc + VS_ACC_HEADER + (size_t)o * per_out;
        const double e2 = rec[0] / dn;            // saltelli.py:577
        const double vy = m2 / (cnt - 1.0);       // saltelli.py:583 (ddof=1 over 2*rows values)
        if (threadIdx.x == 0) {
            E_2[o] = e2;
            var_y[o] = vy;
        }
        const double* ac = rec + 3;
        const double* bd = ac + k;
        const double* ad = bd + k;
        const double* bc = ad + k;
        for (int j = threadIdx.x; j < k; j += blockDim.x) {
            const double uj = (ac[j] / dn1 + bd[j] / dn1) / 2.0;    // :591-593
            const double unj = (ad[j] / dn1 + bc[j] / dn1) / 2.0;   // :594-596
            U_j[(size_t)j * l + o] = uj;
            U_nj[(size_t)j * l + o] = unj;
            sens[(size_t)j * l + o] = (uj - e2) / vy;               // :608
            sens_t[(size_t)j * l + o] = 1.0 - ((unj - e2) / vy);    // :609
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

template <int VEC, int NV, bool MASKED>
static int launch_reduce(vs_context* ctx, const vs_reduce_args& A, dim3 grid, size_t lds) {
    {
        vs_launch_scope scope(ctx, VS_KERNEL_REDUCE_FIRST);
        hipLaunchKernelGGL((reduce_first_kernel<VEC, NV, MASKED>), grid, dim3(VS_WG), lds, ctx->stream, A);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
}

extern "C" int vs_reduce_first_order(vs_context* ctx, const double* f_dev, uint64_t rows, int k, int l,
                                     int flags, int rank, int max_ranks, double* acc_dev) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && acc_dev, "null argument");
    VS_HIP(hipSetDevice(ctx->device));
    VS_REQUIRE(k >= 1 && l >= 1, "k and l must be positive");
    VS_REQUIRE(max_ranks >= 1 && rank >= 0 && rank < max_ranks, "rank %d not in [0,%d)", rank, max_ranks);
    VS_REQUIRE(rows <= 0xfffffff0ull, "rows per GPU must stay below 2^32");
    VS_REQUIRE(rows == 0 || f_dev, "f_dev is null");
    const size_t plen = VS_PLEN(k);
    const size_t width = (size_t)l * plen;
    const bool fast = (rows % 2 == 0) && (((uintptr_t)f_dev & 15) == 0);
    int nv = fast ? ctx->tune.reduce_nv : 8;
    // a tile (256 threads x nv loads x 1 or 2 rows) must lie inside one row block of the f-value layout
    while ((uint64_t)VS_WG * nv * (fast ? 2 : 1) > (1ull << ctx->tune.f_shift) && nv > 2) nv >>= 1;
    const uint64_t tile = (uint64_t)VS_WG * nv * (fast ? 2 : 1);
    uint64_t ntiles = (rows + tile - 1) / tile;
    if (ntiles < 1) ntiles = 1;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    uint64_t cap = (uint64_t)cus * (ctx->tune.reduce_wgs_per_cu > 0 ? ctx->tune.reduce_wgs_per_cu : 1);
    if (l > 1) cap = (cap + l - 1) / l;
    const uint32_t tile_wgs = (uint32_t)(ntiles < cap ? ntiles : cap);
    const int jchunk = k < VS_JCHUNK ? k : VS_JCHUNK;
    const int jblocks = (k + jchunk - 1) / jchunk;

    // workspace: [tile_wgs][l][plen] partials + [l*plen] folded
    int rc = vs_workspace_partials(ctx, sizeof(double) * ((size_t)tile_wgs * width + width));
    if (rc) return rc;
    double* folded = ctx->partials + (size_t)tile_wgs * width;

    vs_reduce_args A;
    A.f = f_dev;
    A.rows = rows;
    A.k = k;
    A.l = l;
    A.jchunk = jchunk;
    A.tile_wgs = tile_wgs;
    A.partials = ctx->partials;
    A.keep = nullptr;
    A.counters = ctx->counters;
    A.f_shift = ctx->tune.f_shift;
    const dim3 grid(tile_wgs * l, jblocks);
    const size_t lds = sizeof(double) * VS_WAVES * (VS_P_HEAD + 4 * (size_t)jchunk);

    vs_fold_args F;
    F.partials = ctx->partials;
    F.P = tile_wgs;
    F.k = k;
    F.l = l;
    F.rows = rows;
    F.f = f_dev;
    F.shift_values = nullptr;
    F.shift_stride = 1;
    F.acc = acc_dev;
    F.rank = rank;
    F.max_ranks = max_ranks;
    F.counters = ctx->counters;
    F.seg_sums = nullptr;   // at most one record per CU: single-stage fold
    F.f_shift = ctx->tune.f_shift;
    const unsigned fold_blocks = (unsigned)((width + VS_FOLD_COLS - 1) / VS_FOLD_COLS);

    for (int pass = 0; pass < 2; ++pass) {
        const bool masked = pass == 1;
        if (masked) {
            if (!(flags & VS_REDUCE_NAN_TRIM) || rows == 0) break;
            rc = vs_workspace_keep(ctx, rows);
            if (rc) return rc;
            A.keep = ctx->keep;
            {
                vs_launch_scope scope(ctx, VS_KERNEL_NAN_MASK);
                hipLaunchKernelGGL(nan_mask_kernel, dim3((unsigned)((rows + 255) / 256 < 4096 ? (rows + 255) / 256 : 4096)),
                                   dim3(256), 0, ctx->stream, f_dev, rows, k, l, ctx->tune.f_shift, ctx->keep, ctx->counters);
            }
            VS_HIP(hipGetLastError());
        }
        if (fast) {
            switch (nv) {
#define VS_CASE(NVV)                                                                             \
    case NVV:                                                                                    \
        rc = masked ? launch_reduce<2, NVV, true>(ctx, A, grid, lds)                             \
                    : launch_reduce<2, NVV, false>(ctx, A, grid, lds);                           \
        break;
                VS_CASE(2)
                VS_CASE(4)
                VS_CASE(8)
#undef VS_CASE
                default:
                    return vs_fail(VS_ERR_ARG, "VS_REDUCE_NV must be 2, 4 or 8");
            }
        } else {
            rc = masked ? launch_reduce<1, 8, true>(ctx, A, grid, lds) : launch_reduce<1, 8, false>(ctx, A, grid, lds);
        }
        if (rc) return rc;
        F.masked = masked ? 1 : 0;
        {
            vs_launch_scope scope(ctx, VS_KERNEL_FOLD);
            hipLaunchKernelGGL(fold_first_kernel, dim3(fold_blocks), dim3(VS_FOLD_THREADS), 0, ctx->stream, F,
                               folded);
        }
        VS_HIP(hipGetLastError());
    }
    return VS_OK;
    VS_GUARD_END
}

// Fold `P` workgroup records sitting in ctx->partials ([P][l][VS_PLEN(k)]) into this rank's
// accumulator record.  Used by the fused evaluate+reduce kernel (vs_generate.hip) and by the
// combined first+second-order kernel (vs_second.hip), whose records have the layout of
// reduce_first_kernel's.  The variance shift comes from shift_dev (l values) or, when that is
// null, from the leading a-values of f_dev as in the first-order pass (output o's shift is
// shift_dev[o * shift_stride]).
int vs_fold_partials(vs_context* ctx, uint32_t P, int k, int l, uint64_t rows, const double* shift_dev,
                     int shift_stride, const double* f_dev, int rank, int max_ranks, double* acc_dev) {
    const size_t width = (size_t)l * VS_PLEN(k);
    vs_fold_args F;
    F.partials = ctx->partials;
    F.P = P;
    F.k = k;
    F.l = l;
    F.rows = rows;
    F.f = f_dev;
    F.shift_values = shift_dev;
    F.shift_stride = shift_stride;
    F.acc = acc_dev;
    F.rank = rank;
    F.max_ranks = max_ranks;
    F.counters = ctx->counters;
    F.masked = 0;
    F.f_shift = ctx->tune.f_shift;
    const uint32_t nseg = vs_fold_segments(P);
    F.seg_sums = nullptr;
    if (nseg > 1) {
        int rc = vs_workspace_fold(ctx, sizeof(double) * nseg * width);
        if (rc) return rc;
        F.seg_sums = ctx->fold_scratch;
    }
    double* folded = ctx->partials + (size_t)P * width;
    const unsigned fold_blocks = (unsigned)((width + VS_FOLD_COLS - 1) / VS_FOLD_COLS);
    {
        vs_launch_scope scope(ctx, VS_KERNEL_FOLD);
        hipLaunchKernelGGL(fold_first_kernel, dim3(fold_blocks, nseg), dim3(VS_FOLD_THREADS), 0, ctx->stream, F, folded);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
}

extern "C" int vs_finish_first_order(vs_context* ctx, const double* acc_dev, uint64_t n, int k, int l,
                                     int max_ranks, double* out_dev) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && acc_dev && out_dev, "null argument");
    VS_HIP(hipSetDevice(ctx->device));
    VS_REQUIRE(k >= 1 && l >= 1 && max_ranks >= 1 && n >= 2, "bad dimensions");
    {
        vs_launch_scope scope(ctx, VS_KERNEL_FINISH);
        hipLaunchKernelGGL(finish_first_kernel, dim3(1), dim3(256), 0, ctx->stream, acc_dev, n, k, l, max_ranks,
                           out_dev);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
    VS_GUARD_END
}
