/*
 * varsens_hip.h -- C ABI of libvarsens_hip.so, the MI355X (gfx950) engine behind
 * the Saltelli hot path of LoLab-MSM/varsens.
 *
 * The reference has no FFI boundary of its own: the path lives behind three
 * Python classes in varsens/saltelli.py.  Each entry point below names the
 * reference code it replaces (paths relative to the reference checkout); the
 * Python shim in varsens_amd/ binds them with ctypes.
 *
 * Conventions
 *   - plain C symbols; every function returns 0 on success, a negative
 *     vs_status on failure; vs_last_error() returns the message of the calling
 *     thread's last failure.  No C++ exception crosses the boundary.
 *   - host buffers are caller-owned, contiguous, and never retained after the
 *     call returns unless the function says "uploaded" (then they are copied).
 *   - device buffers are raw HIP device pointers (void*) owned by a context:
 *     vs_alloc / vs_free.  A context binds one GPU and one HIP stream; it is not
 *     thread-safe.  One process drives one GPU (one context per rank); ranks
 *     meet only in vs_comm_allreduce (RCCL).
 *   - there is NO host fallback: without a usable gfx950 device
 *     vs_context_create fails and nothing else can be called.
 *   - all floating point data is IEEE binary64.
 */
#ifndef VARSENS_HIP_H
#define VARSENS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: perm slices per rank, vs_accumulate, vs_context_set_eval_order, vs_eval_reduce_supported.
 * 3: the second-order sums are S S^T and T T^T (VS_GRAM_LEN doubles) instead of D C^T, D D^T, C C^T.
 * 4: VS_PERM_EMPTY, vs_profile_select.
 * 5: vs_eval_reduce_first_second_order, vs_eval_reduce_both_supported.
 * 6: vs_eval_reduce_first_second_order takes objectives with several outputs; vs_design_desc_validate.
 * 7: vs_design_desc.flags (was `reserved`, 0): VS_DESIGN_SEED1_SHUFFLE.  The field is VALIDATED from here on: a caller
 *    written against ABI <= 6 must have zeroed it (the header said "reserved, 0"); unknown bits are VS_ERR_ARG.
 * 8: the f-values on the device are ROW-BLOCKED (see "F-VALUE LAYOUT"; shards of up to 2048 rows: unchanged);
 *    vs_device_info.f_shift; vs_upload_values / vs_download_values / vs_sample_rows[_begin/_wait]: objective values and
 *    sample rows cross the host boundary in the reference's layouts, converted on the device; vs_build_id;
 *    vs_text_write / vs_text_read (the text wire format, threaded);
 *    VS_ERR_NOMEM; every entry point runs under a
 *    guard that turns C++ exceptions into status codes; a design is released through the context that made it and
 *    vs_context_destroy refuses (VS_ERR_STATE) while designs of the context are alive. */
#define VS_ABI_VERSION 8
#define VS_F_SHIFT 11   /* default log2 of the rows per row block of the f-value layout */

typedef enum vs_status {
    VS_OK = 0,
    VS_ERR_ARG = -1,      /* bad argument (shape, null pointer, unsupported size) */
    VS_ERR_HIP = -2,      /* a HIP runtime call failed */
    VS_ERR_NODEVICE = -3, /* no usable GPU */
    VS_ERR_COMM = -4,     /* RCCL failure */
    VS_ERR_STATE = -5,    /* call sequence error (e.g. comm not initialised) */
    VS_ERR_COMPILE = -6,  /* hiprtc compilation of a user objective failed */
    VS_ERR_NOMEM = -7     /* host memory exhausted (std::bad_alloc inside the library) */
} vs_status;

typedef struct vs_context vs_context;     /* one GPU + one stream */
typedef struct vs_design vs_design;       /* a Saltelli design resident on the GPU */
typedef struct vs_objective vs_objective; /* an objective function resident on the GPU */

/* ------------------------------------------------------------------ basics */

int vs_abi_version(void);
/* Identity of the sources and compiler flags this library was built from (16 hex digits; "unknown" for a build that
 * did not go through varsens_amd/build.py).  Measurement tooling stamps its counter summaries with it; bench.py quotes
 * a committed `roofline.traffic` only when the stamp is this library's. */
const char* vs_build_id(void);
const char* vs_last_error(void);
int vs_device_count(int* count);

typedef struct vs_device_info {
    char name[128];
    char arch[64];
    int32_t compute_units;
    int32_t wavefront_size;
    int32_t lds_bytes_per_cu;
    int32_t clock_mhz;
    uint64_t hbm_bytes;
    int32_t f_shift;      /* ABI 8: log2 of the rows per row block of this context's f-value layout (VS_F_SHIFT) */
    int32_t reserved;
} vs_device_info;

int vs_context_create(int device, vs_context** out);
int vs_context_destroy(vs_context* ctx);
int vs_context_info(vs_context* ctx, vs_device_info* out);
int vs_sync(vs_context* ctx);   /* waits for both streams */

/* Association order of the k-term products / sums of the separable objectives (G_FUNCTION, POLYNOMIAL,
 * LINEAR, separable source objectives).  The reference evaluates whatever Python callable it is given
 * (varsens/saltelli.py:329-355); its own tests multiply the factors with numpy.prod, i.e. left to right.
 *   VS_ORDER_FAST   (default) segmented-scan kernels: O(k) operations per design row, fixed reassociated
 *                   order (csrc/vs_device_scan.h) -- values within a few ulp of the left-to-right fold
 *   VS_ORDER_STRICT strictly left to right: bit-identical to numpy.prod / a Python loop, O(k^2) per row;
 *                   the fused evaluate+reduce path is not available for separable objectives
 * Applies to objectives created and evaluated afterwards. */
enum { VS_ORDER_FAST = 0, VS_ORDER_STRICT = 1 };
int vs_context_set_eval_order(vs_context* ctx, int order);

/* A context owns two HIP streams: 0 = main (default), 1 = auxiliary.  vs_stream_select routes
 * every following launch / collective of this context to that stream; vs_event_record marks a
 * point on a stream (8 slots), vs_stream_wait_event makes a stream wait for a marked point (a
 * never-recorded slot is a no-op).  Used to run the exchange + finishing of pass i beside the
 * evaluation of pass i+1. */
int vs_stream_select(vs_context* ctx, int which);
int vs_event_record(vs_context* ctx, int slot, int which);
int vs_stream_wait_event(vs_context* ctx, int which, int slot);

int vs_alloc(vs_context* ctx, size_t bytes, void** dptr);
int vs_free(vs_context* ctx, void* dptr);
int vs_upload(vs_context* ctx, void* dst_dev, const void* src_host, size_t bytes);
int vs_download(vs_context* ctx, void* dst_host, const void* src_dev, size_t bytes);
int vs_memset(vs_context* ctx, void* dst_dev, int byte, size_t bytes);
/* dst_dev[i] += src_dev[i] for `count` doubles on the context's stream: sums the accumulator records
 * and Gram sums of successive row ranges on the device (no reference counterpart: numpy.sum over the
 * whole arrays, varsens/saltelli.py:577,591-596,612-619, here taken range by range). */
int vs_accumulate(vs_context* ctx, double* dst_dev, const double* src_dev, size_t count);

/* Per-launch timing.  Between begin and end every kernel launched through
 * this context is bracketed by HIP events on the context's stream (no host
 * synchronisation is added).  vs_profile_end synchronises, then fills up to
 * `capacity` records in launch order and returns the number written in *count.
 * kernel_id values: VS_KERNEL_* below. */
typedef struct vs_profile_record {
    int32_t kernel_id;
    float ms;
} vs_profile_record;

enum {
    VS_KERNEL_HALTON_FILL = 1,
    VS_KERNEL_SOBOL_FILL = 2,
    VS_KERNEL_SAMPLE_FLAT = 3,
    VS_KERNEL_EVAL = 4,
    VS_KERNEL_REDUCE_FIRST = 5,  /* the sens/sens_t streaming reduction */
    VS_KERNEL_FOLD = 6,
    VS_KERNEL_FINISH = 7,
    VS_KERNEL_NAN_MASK = 8,
    VS_KERNEL_SECOND_ORDER = 9,
    VS_KERNEL_SECOND_FOLD = 10,
    VS_KERNEL_FUSED = 11,
    VS_KERNEL_ALLREDUCE = 12,   /* the RCCL all-reduce of vs_comm_allreduce (events around the collective) */
    VS_KERNEL_VALUES = 13       /* layout conversion of objective values on their way in or out (vs_upload_values ...) */
};

int vs_profile_begin(vs_context* ctx, int capacity);
int vs_profile_end(vs_context* ctx, vs_profile_record* out, int capacity, int* count);
/* Which launches the next vs_profile_begin brackets: bit VS_KERNEL_x of kernel_mask (0 = all, the default).  Each
 * bracketed launch costs two event packets on its stream -- 0.7276 ms per k = 32, n = 2^22 pass with every kernel
 * bracketed, 0.7052 with the reduction only, 0.6979 with none (tools/step_overhead.py) -- so a timed loop brackets the
 * kernel it wants to price and nothing else. */
int vs_profile_select(vs_context* ctx, uint32_t kernel_mask);

/* ------------------------------------------------ a1: low-discrepancy points
 * Replaces ghalton.Halton(k).get(m) (varsens/saltelli.py:82-84): rows are the
 * Halton points of indices start .. start+count-1 (first point of the stream
 * has index 1), bases = first k primes, row-major (count,k) into device memory.
 * Requires (start+count) * prime_k < 2^53. */
int vs_halton_fill(vs_context* ctx, int k, uint64_t start, uint64_t count, double* out_dev);

/* Capability counterpart of quantlib/sobolGen.cpp:47-60.  `directions` is a
 * host (k,32) table of left-justified 32-bit direction integers (Joe-Kuo;
 * varsens_amd ships one).  Index 0 is the origin.  start+count <= 2^32. */
int vs_sobol_fill(vs_context* ctx, int k, const uint32_t* directions, uint64_t start,
                  uint64_t count, double* out_dev);

/* -------------------------------------------- a1-a5: the design (class Sample)
 * Describes M_1, M_2 and, implicitly, N_j / N_nj (varsens/saltelli.py:62-125)
 * without materialising them. */
enum { VS_GEN_HALTON = 0, VS_GEN_SOBOL = 1, VS_GEN_RAW = 2 };
enum { VS_SCALE_IDENTITY = 0, VS_SCALE_LINEAR = 1, VS_SCALE_POWER = 2 };

typedef struct vs_design_desc {
    int32_t generator;     /* VS_GEN_* */
    int32_t k;             /* parameters */
    uint64_t n;            /* draws (global n, before any sharding) */
    uint64_t first_index;  /* sequence index that becomes row 0 of M_1; row r of the
                              un-shuffled second half is first_index + n + r
                              (saltelli.py:83-84,92,95).  Ignored for VS_GEN_RAW. */
    int32_t scale_mode;    /* VS_SCALE_*: scale.py:33 (linear), :62 (power) */
    int32_t flags;         /* VS_DESIGN_*; 0 from callers of ABI <= 6 (the field was `reserved`) */
    const double* lower;   /* host, k entries (LINEAR/POWER) */
    const double* upper;   /* host, k entries (LINEAR/POWER) */
    const uint32_t* perm;  /* host: M_2[r] = second_half[perm[r - perm_begin]] for the rows
                              [perm_begin, perm_begin + perm_count) (the seed-1 shuffle, saltelli.py:100-101);
                              NULL = identity.  perm_count = 0 means all n rows from row 0 (n entries are
                              read); perm_count = VS_PERM_EMPTY means an EMPTY slice (a rank with no rows:
                              `perm` must be non-NULL and is not read).  A rank of a multi-GPU job passes its
                              own shard's slice only; vs_eval then accepts rows of that slice, and
                              vs_sample_flat needs the whole shuffle. */
    const double* raw;     /* host (2n,k) unit-cube points for VS_GEN_RAW
                              (Sample(raw=...), saltelli.py:69-73), uploaded */
    const uint32_t* sobol_directions; /* host (k,32) for VS_GEN_SOBOL, uploaded */
    uint64_t perm_begin;   /* first design row `perm` covers */
    uint64_t perm_count;   /* entries in `perm` (0: n; VS_PERM_EMPTY: none) */
} vs_design_desc;
#define VS_PERM_EMPTY UINT64_MAX
/* flags (ABI 7).  VS_DESIGN_SEED1_SHUFFLE: `perm` is NULL and the library supplies the seed-1 shuffle of
 * numpy.random.seed(1); numpy.random.shuffle(M_2) (saltelli.py:100-101) itself, for the rows [perm_begin, perm_begin +
 * perm_count) (perm_count = 0: all n rows; VS_PERM_EMPTY: none) -- vs_seed1_permutation on the host, once per n and
 * process, and ONE device copy per (n, slice) kept in the context for every later design that asks for it (a
 * Varsens(objective, scaling, k, n) call per model of a study uploads the 4 n bytes once, not once per call: k = 32,
 * n = 2^22: 2.1 -> 0.7 ms per call).  Copies no design uses any more are dropped, least recently used first, when more
 * than four (or 1 GiB of them) are held. */
#define VS_DESIGN_SEED1_SHUFFLE 1

/* Host helper (no GPU involved): out[r] = index of the row that numpy.random.seed(1);
 * numpy.random.shuffle(M_2) (saltelli.py:100-101) moves to position r -- MT19937 init_genrand(1),
 * Fisher-Yates from the last row down, masked-rejection draws.  Feed it to vs_design_desc.perm. */
int vs_seed1_permutation(uint32_t* out, uint64_t n);

/* Host helper (no GPU involved; ABI 6): every check vs_design_create makes on a descriptor before anything is
 * uploaded -- k, n, generator, scale mode and its bounds (scale.py:33,62), raw points / direction numbers where the
 * generator needs them, and the shuffle slice of a rank (perm_begin, perm_count incl. VS_PERM_EMPTY; every entry must
 * be a row of the second half, saltelli.py:100-101).  *perm_entries (may be NULL): the number of entries of
 * desc->perm that will be read.  VS_OK or VS_ERR_ARG with vs_last_error(). */
int vs_design_desc_validate(const vs_design_desc* desc, uint64_t* perm_entries);

/* Ownership: a design belongs to the context that created it -- its tables live on that GPU and, with
 * VS_DESIGN_SEED1_SHUFFLE, it holds a share of that context's shuffle cache.  vs_design_destroy releases it through
 * that context whatever `ctx` is passed (NULL included); vs_context_destroy returns VS_ERR_STATE while designs of the
 * context are alive.  Destroy designs (and objectives) before their context. */
int vs_design_create(vs_context* ctx, const vs_design_desc* desc, vs_design** out);
int vs_design_destroy(vs_context* ctx, vs_design* design);

/* ---------------------------------- sample rows for a model that runs on the host (call stack (b), saltelli.py:308-355)
 * The reference hands sample.M_1[i], sample.M_2[i], sample.N_j[i][j], sample.N_nj[i][j] to the objective one row at a
 * time.  vs_sample_rows writes rows [row_begin, row_begin+row_count) of Sample.flat() (saltelli.py:127-160) straight
 * into caller memory: the rows are produced piece by piece on the device and leave through a pinned ring, the kernel of
 * piece i+1 running under the copy of piece i.  The asynchronous pair lets a host loop evaluate chunk i while chunk i+1
 * is produced and copied: vs_sample_rows_begin enqueues kernel and copy of a chunk into one of two library-owned pinned
 * slots and returns at once; vs_sample_rows_wait blocks until that slot is filled and returns a pointer to its
 * row_count x k doubles, valid until the next vs_sample_rows_begin on the same slot. */
int vs_sample_rows(vs_context* ctx, const vs_design* design, uint64_t row_begin, uint64_t row_count, double* host_rows);
int vs_sample_rows_begin(vs_context* ctx, const vs_design* design, uint64_t row_begin, uint64_t row_count, int slot);
int vs_sample_rows_wait(vs_context* ctx, int slot, const double** host_rows);

/* Rows [row_begin, row_begin+row_count) of Sample.flat() (saltelli.py:127-160):
 * global row order [M_1; M_2; N_j[0..k); N_nj[0..k)], 2n(k+1) rows of k values,
 * written row-major to out_dev.  Serves Sample.M_1/M_2/N_j/N_nj and export. */
int vs_sample_flat(vs_context* ctx, const vs_design* design, uint64_t row_begin,
                   uint64_t row_count, double* out_dev);

/* ----------------------------------------- a6: objectives (class Objective)
 * Built-in objectives evaluated on the GPU.  l = outputs per evaluation.
 *   G_FUNCTION  params = a[l][k]      f_o = prod_i (|4x_i-2|+a_oi)/(1+a_oi)
 *   ISHIGAMI    params = {a, b}       k must be 3, l = 1
 *   LINEAR      params = c[l][k]      f_o = sum_i c_oi x_i
 *   POLYNOMIAL  params = beta[l][k]   f_o = prod_i (1 + beta_oi (x_i - 1/2))
 *   INVERT      no params             f_o = 1 - x_o, l = k (varsens/tests/test_objective.py:5)
 * Operation order is fixed and documented in DESIGN.md (the oracle follows it). */
enum {
    VS_OBJ_G_FUNCTION = 1,
    VS_OBJ_ISHIGAMI = 2,
    VS_OBJ_LINEAR = 3,
    VS_OBJ_POLYNOMIAL = 4,
    VS_OBJ_INVERT = 5,
    VS_OBJ_SOURCE = 6 /* created by vs_objective_create_source */
};

int vs_objective_create(vs_context* ctx, int kind, int k, int l, const double* params,
                        size_t nparams, vs_objective** out);
int vs_objective_destroy(vs_context* ctx, vs_objective* obj);

/* A user objective given as the BODY of
 *     void objective(const vs_point& x, int k, const double* p, double* out)
 * in HIP C++: x[c] is coordinate c of the evaluation point (already scaled), p the `nparams`
 * parameters, out[0..l) the outputs.  Example body: "out[0] = sin(x[0]) + p[0]*x[1]*x[1];".
 * Compiled with hiprtc for gfx950 (-ffp-contract=off) into the same fused generate->evaluate
 * kernel as the built-ins; on a compile error returns VS_ERR_COMPILE and vs_last_error() holds
 * the compiler log.  Device counterpart of the arbitrary Python callable the reference takes
 * (varsens/saltelli.py:308,329-355). */
int vs_objective_create_source(vs_context* ctx, int k, int l, const char* body, const double* params,
                               size_t nparams, vs_objective** out);

/* A user objective of SEPARABLE shape, f(x) = t_0(x_0) op t_1(x_1) op ... (op = * or +, folded left to
 * right), given as the BODY of `double term(double x, int c, const double* p)` (the value of term c at
 * coordinate value x).  Compiled into the chain-scheme kernel of the built-in separable objectives, so
 * it runs at their speed (k^2/2 folds per family instead of k per point); one output, k <= 160.
 * Example (Sobol g-function): "return (fabs(4.0*x - 2.0) + p[c]) / (1.0 + p[c]);" with VS_FOLD_PRODUCT. */
enum { VS_FOLD_PRODUCT = 0, VS_FOLD_SUM = 1 };
int vs_objective_create_separable(vs_context* ctx, int k, const char* term_body, int fold,
                                  const double* params, size_t nparams, vs_objective** out);

/* F-VALUE LAYOUT on the device: ROW-BLOCKED planes (ABI 8; ABI <= 7 was planar, the special case of one row block).
 * The 2(k+1) * rows * l values of a shard -- the reference's fM_1, fM_2, fN_j, fN_nj of its rows, Objective.flat()
 * (saltelli.py:357-391) transposed to one PLANE per (output o, block b), plane p = o (2k+2) + b, block order
 * [fM_1, fM_2, fN_j[0..k), fN_nj[0..k)] -- are stored row block by row block: rows [R 2^s, (R+1) 2^s) form row block R
 * (s = VS_F_SHIFT = 11: 2048 rows; vs_device_info.f_shift says what a context uses), and inside a row block the
 * planes follow each other, each holding the row block's rows contiguously:
 *     f_dev[(R << s) * P + p * pitch_R + (r - (R << s))],   R = r >> s,  P = l (2k+2),  pitch_R = min(2^s, rows - (R << s))
 * The last row block is packed at its own row count: exactly P * rows doubles in all.  A shard of at most 2^s rows is one
 * row block -- plane after plane, f_dev[p * rows + r], the layout of ABI <= 7; for l = 1 that is Objective.flat() itself.
 * Why: a workgroup that produces or consumes a tile of rows touches every plane of those rows.  Plane-major storage puts
 * them `rows` doubles apart (k = 128, n = 2^24: 258 streams 128 MiB apart, a TLB entry and a DRAM page each); row-blocked
 * they lie within P * 16 KiB -- the evaluation of that shape stores at 5.6-5.9 instead of 4.1 TB/s.  Every block of every
 * output is still a set of contiguous runs, so all kernels run at the l = 1 speed for any l.
 * Host code never needs the formula: vs_upload_values / vs_download_values (below) convert between the reference's
 * (2(k+1) rows, l) arrays and this layout on the device.
 *
 * vs_eval replaces the four evaluation loops of Objective.__init__ (saltelli.py:329-355)
 * for design rows [row_begin, row_begin+row_count) (a shard of r in [0,n)) and writes
 * 2(k+1)*row_count*l doubles in that layout. */
int vs_eval(vs_context* ctx, const vs_design* design, const vs_objective* obj,
            uint64_t row_begin, uint64_t row_count, double* f_dev);

/* ------------------------------------------ a7: objective values from / to the host (Objective.load, Objective.flat)
 * vs_upload_values replaces the slicing, `/ scaling` and NaN scan of Objective.load (saltelli.py:449-486) for values a
 * user computed elsewhere (call stack (c): Objective(k, n, objective_vals=v); (d): files).  host_vals is the reference's
 * flat array, (2 n (k+1), l) row-major in block order [fM_1; fM_2; fN_j[0..k); fN_nj[0..k)] for the GLOBAL n; rows
 * [row_begin, row_begin+rows) of every block (a rank's shard; 0, n for all) go to f_dev in the planar layout above
 * (2 (k+1) rows l doubles).  `divide`: NULL, or 1 or l divisors -- value / divide[0] or value / divide[output], a true
 * division (the reference's x / scaling).  *nan_rows (may be NULL): how many of the shard's rows hold a NaN in output 0
 * of any block -- the rows Objective.load deletes (saltelli.py:474-486) and VS_REDUCE_NAN_TRIM leaves out --, counted
 * on the device.  The host never transposes, divides or scans: pieces of the array go through a pinned ring, the copy
 * engine and a tile-transposing kernel (one output without divisor: the copy engine writes the destination itself).
 * Synchronous: host_vals is not read after the call returns.
 * vs_download_values is the inverse, Objective.flat() of the shard (saltelli.py:357-391): (2 (k+1) rows, l) row-major. */
int vs_upload_values(vs_context* ctx, const double* host_vals, uint64_t n, int k, int l, uint64_t row_begin,
                     uint64_t rows, const double* divide, int ndivide, double* f_dev, uint64_t* nan_rows);
int vs_download_values(vs_context* ctx, const double* f_dev, uint64_t rows, int k, int l, double* host_vals);

/* ------------------------------------------------ the text wire format (host only: no GPU, no context)
 * Sample.export / Objective.export write their arrays with numpy.savetxt (saltelli.py:187,192,407,412): every value as
 * '%.18e', the values of a row joined by `delimiter` (tab for samples, one space for objectives), one row per line;
 * Sample.load / Objective.load read them with numpy.loadtxt (:217,439).  These files are how the reference fans a study
 * out over a cluster (varsens/cluster/accre-job.sh:13).  vs_text_write produces the bytes numpy.savetxt produces, and
 * vs_text_read the values numpy.loadtxt reads ('#' comments and blank lines skipped; delimiter NULL or "": any run of
 * blanks), formatted / parsed on several threads (numpy: 0.5 / 0.25 us per value on one core).  vs_text_read with
 * out = NULL only sizes the file (*rows data lines, *cols numbers in the first of them); with a buffer of `capacity`
 * doubles it fills rows x cols values row-major and fails (VS_ERR_ARG, the row named) on a line that does not hold
 * *cols numbers. */
int vs_text_write(const char* path, const double* values, uint64_t rows, uint64_t cols, const char* delimiter);
int vs_text_read(const char* path, const char* delimiter, double* out, uint64_t capacity, uint64_t* rows, uint64_t* cols);

/* ------------------------------ a7-a11: the estimators (Varsens.compute_varsens)
 * Accumulator record produced per GPU and summed across GPUs.  All doubles.
 *   [0] rows_kept   [1] rows_dropped (NaN-trimmed)   [2] ranks   [3] reserved
 *   then for each output o (stride VS_ACC_PER_OUTPUT(k)):
 *     sab, sa, sb, ac[k], bd[k], ad[k], bc[k]
 *   then `max_ranks` slots of (count, mean[l], m2[l]) -- each rank fills its own
 *   slot, so the sum-all-reduce doubles as the all-gather the variance merge needs.
 */
#define VS_ACC_HEADER 4
#define VS_ACC_PER_OUTPUT(k) (3 + 4 * (size_t)(k))
#define VS_ACC_SLOT(l) (1 + 2 * (size_t)(l))
#define VS_ACC_LEN(k, l, max_ranks) \
    (VS_ACC_HEADER + (size_t)(l) * VS_ACC_PER_OUTPUT(k) + (size_t)(max_ranks) * VS_ACC_SLOT(l))

enum { VS_REDUCE_NAN_TRIM = 1 };

/* The sens/sens_t streaming reduction over a shard's f-values (planar layout
 * above).  One pass over HBM; replaces saltelli.py:577 (E_2 sum), :583
 * (variance sums) and :591-596 (the 4k dot products).  With VS_REDUCE_NAN_TRIM,
 * rows holding a NaN in output 0 of any block are left out of every sum
 * (saltelli.py:474-492).  Writes this rank's record (slot `rank` of
 * `max_ranks`) to acc_dev (VS_ACC_LEN doubles, device). */
int vs_reduce_first_order(vs_context* ctx, const double* f_dev, uint64_t rows, int k, int l,
                          int flags, int rank, int max_ranks, double* acc_dev);

/* Fused path: evaluates the shard's 2(k+1)*row_count points and reduces them in registers --
 * no f-values are written or read (the record equals vs_eval + vs_reduce_first_order's up to
 * summation order and, for separable objectives, the association order of the k-term folds:
 * rounding-level differences, <= ~2e-13 relative on the record).  Covers G_FUNCTION / POLYNOMIAL /
 * LINEAR objectives (any number of outputs, k <= 128, identity, linear or power scaling), separable
 * source objectives with k <= 128 on designs with identity or linear scaling, and ISHIGAMI; returns
 * VS_ERR_ARG otherwise (k > 128, INVERT, general source objectives, Halton indices beyond 2^32,
 * a context in strict evaluation order; callers then
 * use the two-kernel path; vs_eval_reduce_supported tells in advance: *supported = 1 or 0).  NaN
 * values propagate (as on the reference's live-evaluation route, saltelli.py:329-355).
 * Replaces saltelli.py:329-355 + :577,583,591-596 in one kernel. */
int vs_eval_reduce_supported(vs_context* ctx, const vs_design* design, const vs_objective* obj, int* supported);
int vs_eval_reduce_first_order(vs_context* ctx, const vs_design* design, const vs_objective* obj,
                               uint64_t row_begin, uint64_t row_count, int rank, int max_ranks,
                               double* acc_dev);

/* Fused path for BOTH orders: the same evaluation walk writes s = c + d, t = c - d of every 64-row block into an
 * LDS tile, and the workgroup multiplies it on the fp64 matrix cores (S S^T, T T^T) and takes the first-order dot
 * products off the same fragments: this rank's record (acc_dev, as vs_reduce_first_order) and Gram sums (gram_dev,
 * VS_GRAM_LEN(k l) doubles, as vs_reduce_second_order) without any f-value in HBM.  Covers G_FUNCTION / POLYNOMIAL /
 * LINEAR objectives with one output and k <= 128 (every generator and scaling of the design) and with SEVERAL outputs
 * (ABI 6; the reference's two-output known-answer test, varsens/tests/test_g_function.py:77-88): 2 .. 8 outputs and
 * k l <= 32 index rows, k >= 16 and k l <= 64 (2 .. 4 outputs), or 2 or 4 outputs and k l <= 128 (3 outputs: k l <= 112) -- the shapes where
 * the one kernel is measured faster than the two-kernel route --, every scaling (power: not 33 .. 64 index rows); the generator runs once per
 * design row, every output builds its terms from the same coordinates, index row = column * l + output of ONE tile, so
 * that sens_2[:, o, :, o'] of every pair of outputs comes out of the same pass; separable source objectives with one
 * output and k <= 128 (identity or linear scaling); ISHIGAMI.  Halton indices below 2^32; not a context in strict
 * evaluation order; VS_ERR_ARG otherwise -- vs_eval_reduce_both_supported tells in
 * advance, callers then use vs_eval + vs_reduce_first_second_order.  NaN values propagate.  Results equal the
 * two-kernel path's up to summation order (<= ~2e-13 relative on the sums).
 * Replaces saltelli.py:329-355 + :577,583,591-596 + :612-619 in one kernel. */
int vs_eval_reduce_both_supported(vs_context* ctx, const vs_design* design, const vs_objective* obj, int* supported);
int vs_eval_reduce_first_second_order(vs_context* ctx, const vs_design* design, const vs_objective* obj,
                                      uint64_t row_begin, uint64_t row_count, int rank, int max_ranks,
                                      double* acc_dev, double* gram_dev);

/* Finished indices from a (summed) record; divisors use the REQUESTED global n
 * even when rows were trimmed (saltelli.py:576-577,591).  out_dev layout
 * (doubles): E_2[l], var_y[l], U_j[k][l], U_nj[k][l], sens[k][l], sens_t[k][l]
 * = (2 + 4k) * l values (saltelli.py:577-609). */
#define VS_FIRST_ORDER_LEN(k, l) ((2 + 4 * (size_t)(k)) * (size_t)(l))
int vs_finish_first_order(vs_context* ctx, const double* acc_dev, uint64_t n, int k, int l,
                          int max_ranks, double* out_dev);

/* a12: second-order Gram sums over a shard (saltelli.py:612-619).  With C = fN_j,
 * D = fN_nj viewed as (k*l, rows) matrices, the reference needs only two symmetric
 * combinations of its four tensordots: D C^T + C D^T (:612-613) and D D^T + C C^T
 * (:618-619).  With S = C + D and T = C - D these are (S S^T - T T^T) / 2 and
 * (S S^T + T T^T) / 2: two symmetric (k*l, k*l) products instead of the upper
 * triangle of the (2kl, 2kl) Gram matrix -- half the matrix-core work.  gram_dev
 * receives VS_GRAM_LEN(k*l) doubles: SS = S S^T, then TT = T T^T, row-major, both
 * triangles filled; sums over shards / ranks add entrywise.  NaN-trimmed like the
 * first-order pass when VS_REDUCE_NAN_TRIM is set. */
#define VS_GRAM_LEN(K) (2 * (size_t)(K) * (size_t)(K))
int vs_reduce_second_order(vs_context* ctx, const double* f_dev, uint64_t rows, int k, int l,
                           int flags, double* gram_dev);

/* a8-a10 + a12 in ONE pass over the f-values: Varsens.compute_varsens always needs both
 * (saltelli.py:572-622), and both passes read the same 2k blocks.  Fills acc_dev exactly like
 * vs_reduce_first_order and gram_dev exactly like vs_reduce_second_order.  Single-output
 * designs without NaN trimming take one pass: the first-order dot products ride on the Gram
 * kernels' operand fragments (k <= 32: the wave-private kernel; larger k: the diagonal pairs of
 * the macro-tile kernel, plus one small pass over fM_1, fM_2 for the scalar sums); every other
 * shape runs the two separate passes internally, so the call is always valid. */
int vs_reduce_first_second_order(vs_context* ctx, const double* f_dev, uint64_t rows, int k, int l,
                                 int flags, int rank, int max_ranks, double* acc_dev, double* gram_dev);

/* sens_2, sens_2n (each (k,l,k,l) row-major) from the summed SS, TT of
 * vs_reduce_second_order and the finished first-order block (E_2, var_y)
 * (saltelli.py:612-622). */
int vs_finish_second_order(vs_context* ctx, const double* gram_dev, const double* first_dev,
                           uint64_t n, int k, int l, double* sens2_dev, double* sens2n_dev);

/* ------------------------------------------------------- multi-GPU exchange
 * One rank per process/GPU.  Rank 0 calls vs_comm_unique_id and ships the 128
 * bytes to the others by any host channel; then every rank calls vs_comm_init.
 * vs_comm_allreduce sums `count` doubles in place over xGMI (ncclAllReduce on
 * the context's stream).  The reference has no counterpart (its only scale-out
 * is PBS array jobs over text files, varsens/cluster/accre-job.sh:13). */
#define VS_COMM_ID_BYTES 128
int vs_comm_unique_id(void* id_out);
int vs_comm_init(vs_context* ctx, const void* id, int nranks, int rank);
int vs_comm_allreduce(vs_context* ctx, double* buf_dev, size_t count);
int vs_comm_destroy(vs_context* ctx);

#ifdef __cplusplus
}
#endif
#endif /* VARSENS_HIP_H */
