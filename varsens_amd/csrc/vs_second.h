// Shared by the translation units of the second-order pass (vs_second.hip, vs_gram_st.hip).
#pragma once

#include "vs_common.h"

#define VS2_SUPER 64                 // index rows per super-block (4 MFMA blocks of 16)
#define VS2_CHUNK 64                 // sample rows a workgroup of four waves takes per visit (4 x VS2_WCHUNK)
#define VS2_TILE_OUT (VS2_SUPER * VS2_SUPER)
#define VS2_WCHUNK 16                // sample rows of one LDS sub-tile
#define VS2_WPITCH 20                // doubles per tile row: 16 sample rows + 4 of padding (conflict-free b128 fragments)
#define VS2_MACRO 128                // index rows per macro block
#define VS2_MTILE_OUT (VS2_MACRO * VS2_MACRO)

typedef double vs_d2 __attribute__((ext_vector_type(2)));
typedef double vs_d4 __attribute__((ext_vector_type(4)));

enum { CNT_NANFLAG = 0 };

struct vs_gram_args {
    const double* f;
    uint64_t rows;
    int k, l, K;
    int nsuper;          // ceil(2K / 64)
    uint32_t slab_wgs;   // grid.x
    double* partials;    // [slab_wgs][npairs][64][64]
    double* first_partials;  // combined pass only: [slab_wgs][VS_PLEN(k)] first-order records
    const uint8_t* keep;
    const uint32_t* counters;
    int f_shift;         // log2 of the rows per row block of the f-value layout (vs_f_tile, vs_device_gen.h)
};

// Row-blocked f-values (vs_f_tile): every fetch of the second-order kernels takes a chunk of 16, 32 or 64 consecutive
// sample rows starting at a multiple of its size -- inside one row block --, so per fetch there is one wave-uniform
//   fb = f + base   (plane 0, first sample row of the chunk)    fp = pitch (doubles between the planes of these rows)
// and what the planar layout addressed as  f + plane * rows + s0 + i  is  fb + plane * fp + i.
#define VS2_FTILE(s0)                                                                                   \
    const vs_ftile ft__ = vs_f_tile(rows, (uint64_t)A.l * (2 * (uint64_t)A.k + 2), A.f_shift, (s0));    \
    const double* const fb = A.f + ft__.base;                                                           \
    const uint64_t fp = ft__.pitch

// vs_gram_st.hip: the S/T kernel for 33 .. 128 index rows per half (one output, even row count, no mask)
int vs_gram_st_prepare(vs_context* ctx);
int vs_gram_st_launch(vs_context* ctx, const vs_gram_args& A, bool with_first, unsigned workgroups);

// vs_gram_dma.hip: 33 .. 64 index rows per half, one output, even row count, no mask, second order alone: every wave
// on its own image, filled by LDS-DMA in whole lines (round 5)
int vs_gram_dma_prepare(vs_context* ctx);
int vs_gram_dma_launch(vs_context* ctx, const vs_gram_args& A, bool with_first, unsigned workgroups);
bool vs_gram_dma_takes_first(int K);   // the shapes whose first-order sums it can carry (33 .. 64 index rows)
