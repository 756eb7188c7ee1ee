// Internal declarations shared by the translation units of libvarsens_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "varsens_hip.h"
#include "vs_guard.h"

// f-value layout (vs_device_gen.h: vs_f_tile): default and smallest log2 of the rows per row block
#ifndef VS_F_SHIFT
#define VS_F_SHIFT 11
#endif
#define VS_LDS_BYTES ((size_t)160 * 1024)   // LDS of a gfx950 CU: the most a workgroup can have
#ifndef VS_F_SHIFT_MIN
#define VS_F_SHIFT_MIN 11
#endif

struct vs_prof_slot {
    int kernel_id;
    hipEvent_t start, stop;
};

// Measurement knobs of the tools/ scripts (launch shapes, kernel variants).  Read from the
// environment ONCE, when the context is created; product code never calls getenv afterwards.
struct vs_tuning {
    int reduce_nv = 4;           // VS_REDUCE_NV: 16-byte loads per thread and block (2, 4, 8)
    int reduce_wgs_per_cu = 1;   // VS_REDUCE_WGS_PER_CU
    int eval_scratch = 0;        // VS_EVAL_SCRATCH=1: chain kernel with its tile in HBM scratch (tests force it at small k)
    int eval_waves = 0;          // VS_EVAL_WAVES_PER_WG: 4 or 8 (0 = by tile size)
    int eval_mode = 0;           // vs_context_set_eval_order / VS_EVAL_MODE: 0 = scan kernels where they apply, 1 = strict order (chain scheme)
    int fused_wgs_per_cu = 0;    // VS_FUSED_WGS_PER_CU (0 = by kernel)
    int scan_general = 0;        // VS_SCAN_GENERAL=1: the scan kernels' general column code also for Halton designs (tests)
    int scan_shape = 0;          // VS_SCAN_SHAPE=L*100+W: force the evaluation scan kernel's shape (must be instantiated)
    int fused_variant = 0;       // VS_FUSED_VARIANT=1 or L*100+W: 64 < k <= 128 fused as 8 waves x 16 columns instead of 16 x 8
    int gram_wgs_per_cu = 0;     // VS_GRAM_WGS_PER_CU (0 = by shape)
    int gram_st = 1;             // VS_GRAM_ST=0: 33..128 index rows per half keep the [C; D] macro kernel
    int gram_wide = 1;           // VS_GRAM_WIDE=0: keep gram_diag_wave_kernel for 33..64 index rows
    int gram_dma = 1;            // VS_GRAM_DMA=0: 33..64 index rows per half stay on gram_st_kernel (tests compare the two); 2: gram_dma_kernel also for 17..32
    int reduce_both = 1;         // VS_REDUCE_BOTH=0: never take the combined first+second-order kernel
    int fused_both_wide = 1;     // VS_FUSED_BOTH_WIDE=0: no fused evaluate + both-orders kernel beyond 32 columns; 2 (with
                                 // VS_TUNING=1, tools/fused_multi_sweep.sh): also the several-output shapes measured slower
    int f_shift = VS_F_SHIFT;    // VS_F_SHIFT: log2 of the rows per row block of the f-value layout (vs_f_tile): 11 .. 40; >= log2(rows)
                                 // is the planar layout of ABI <= 7 (tests run the kernels at 11 to cross row blocks at small n)
    int copy_threads = 4;        // VS_COPY_THREADS: host threads that feed / drain the pinned ring of vs_values.hip
    long profile_mask = -1;      // VS_PROFILE_MASK: bracket only these kernel ids with events
};

struct vs_context {
    int device = -1;
    vs_tuning tune;
    hipStream_t stream = nullptr;      // the stream launches currently go to (streams[selected])
    hipStream_t streams[2] = {nullptr, nullptr};  // 0 = main, 1 = auxiliary (exchange / finishing)
    hipEvent_t events[8] = {};         // cross-stream ordering slots (vs_event_record / vs_stream_wait_event)
    bool event_set[8] = {};
    hipDeviceProp_t prop;
    // partial-sum workspace of the reductions (grown on demand, never shrunk)
    double* partials = nullptr;
    size_t partials_bytes = 0;
    // HBM tile scratch of the evaluation kernel for k beyond the LDS budget
    double* scratch = nullptr;
    size_t scratch_bytes = 0;
    // NaN-trim keep mask workspace (one byte per row)
    uint8_t* keep = nullptr;
    size_t keep_bytes = 0;
    // segment sums of the two-stage folds (many partial records: the fused kernels' 2 x resident workgroups)
    double* fold_scratch = nullptr;
    size_t fold_scratch_bytes = 0;
    uint32_t* counters = nullptr;  // 64 x u32 scratch counters (device)
    uint32_t* counters_host = nullptr;  // pinned mirror
    // profiling
    bool profiling = false;
    uint32_t profile_select = 0;    // vs_profile_select: kernel ids to bracket (0 = all)
    bool gram_lds_raised = false;   // second-order kernels with more than 64 KiB of dynamic LDS: limit raised once
    std::vector<vs_prof_slot> prof;
    std::vector<vs_prof_slot> prof_pool;
    size_t prof_capacity = 0;
    // device copies of the seed-1 shuffle (VS_DESIGN_SEED1_SHUFFLE): one per (n, slice), shared by the designs that
    // asked for it (refs), dropped least-recently-used once unreferenced and more than four are held
    struct perm_entry {
        uint64_t n, begin, count;
        uint32_t* dev;
        int refs;
        uint64_t stamp;
    };
    std::vector<perm_entry> perm_cache;
    uint64_t perm_stamp = 0;
    // RCCL
    void* comm = nullptr;
    int comm_ranks = 0;
    int comm_rank = -1;
    // pinned ring, device staging and copy threads of the calls that move objective values and sample rows across
    // the host boundary (vs_values.hip); made on first use
    struct vs_staging* staging = nullptr;
    // designs made by this context that are still alive (vs_design_destroy takes them off): vs_context_destroy refuses
    // while there are any -- their tables and their shares of the shuffle cache would dangle
    int live_designs = 0;

    vs_context() = default;
    vs_context(const vs_context&) = delete;
    vs_context& operator=(const vs_context&) = delete;
    // releases whatever was created, in whatever state a failed vs_context_create left the object (vs_api.hip)
    ~vs_context();
};
void vs_staging_release(vs_context* ctx);   // vs_values.hip

struct vs_design {
    vs_context* owner = nullptr;  // the context that made it: its tables live on that GPU, a shared shuffle in its cache
    vs_design_desc d;          // host copy (pointers below replace the host ones)
    double* lower = nullptr;   // device, k (POWER: lower; LINEAR: lower)
    double* width = nullptr;   // device, k (LINEAR: upper-lower; POWER: upper/lower)
    uint32_t* perm = nullptr;  // device, n or null
    bool perm_shared = false;  // perm belongs to ctx->perm_cache (VS_DESIGN_SEED1_SHUFFLE): released, not freed
    double* raw = nullptr;     // device (2n,k) or null
    uint32_t* dirs = nullptr;  // device (k,32) or null
    uint32_t* sobol_tabs = nullptr;  // device (k,4,256): XOR of the direction integers selected by each byte value; then (k,8,16): by each nibble
    uint32_t* bases = nullptr; // device, k x vs_base (vs_device_gen.h)
    int halton_horner = 0;     // every column's record carries VS_BASE_HORNER (vs_radical_inverse_pair_horner is exact)
};

struct vs_objective {
    int kind = 0;
    int k = 0;
    int l = 1;
    double* params = nullptr;  // device
    size_t nparams = 0;
    void* module = nullptr;    // hipModule_t of a run-time compiled objective
    void* function = nullptr;  // hipFunction_t (vs_user_eval)
    void* fused_function = nullptr;  // hipFunction_t (vs_user_fused): separable source objectives with k <= 128
    void* gram_function = nullptr;   // hipFunction_t (vs_user_gram): ... evaluate + both orders
    size_t gram_lds = 0;
    int gram_threads = 256;          //   its launch shape and the side of its Gram partials [2][gram_sb][gram_sb]
    int gram_sb = 32;
    bool eval_scan = false;    // vs_user_eval is the scan kernel (separable source objective): one 64-row block per workgroup
    int eval_threads = 256;    // launch shape of vs_user_eval / vs_user_fused
    size_t eval_lds = 0;       //   (eval_lds = 0: the 2k x 64 tile of the chain / general kernels)
    int fused_threads = 0;
    size_t fused_lds = 0;
    bool scratch_tile = false; // compiled for the HBM scratch tile (k beyond the LDS budget)
    // G_FUNCTION: the l*k correctly rounded reciprocals 1 / (1 + a) follow the parameters in `params` (0 where 1 + a is
    // not a plain normal number or has an all-ones significand: that column divides); true when every column is plain --
    // then the scan kernels' specialised column code, which carries no division, may serve the objective
    bool recip_ok = false;
};

int vs_rtc_build(vs_context* ctx, vs_objective* o, const char* body, int fold);
void vs_rtc_release(vs_objective* o);
// (sobol: the launch is for a Sobol design -- the scan-form kernels then get a KiB of LDS per wave for its nibble tables)
int vs_rtc_launch(vs_context* ctx, const vs_objective* o, const void* eval_args, size_t eval_args_bytes,
                  unsigned blocks, size_t lds, bool sobol);
int vs_rtc_launch_fused(vs_context* ctx, const vs_objective* o, const void* eval_args, size_t eval_args_bytes,
                        unsigned P, const double* shift_dev, bool sobol);
// launch shape of the scan kernels for k columns (vs_scan.hip); false: k is outside them
bool vs_scan_shape(int k, bool fused, int* L, int* W, int* tlds, size_t* lds_bytes);
int vs_rtc_fused_plan(vs_context* ctx, const vs_objective* o, uint64_t rows, unsigned* P_out, bool sobol);
int vs_rtc_gram_plan(vs_context* ctx, const vs_objective* o, uint64_t rows, unsigned* P_out, bool sobol);
int vs_rtc_launch_gram(vs_context* ctx, const vs_objective* o, const void* eval_args, size_t eval_args_bytes,
                       unsigned P, const double* shift_dev, double* first_partials, double* gram_partials, bool sobol);

#define VS_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess)                                                              \
            return vs_fail(VS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                           __FILE__, __LINE__);                                             \
    } while (0)

// Brackets a launch with profiling events when enabled.
struct vs_launch_scope {
    vs_context* ctx;
    int slot = -1;
    vs_launch_scope(vs_context* c, int kernel_id);
    ~vs_launch_scope();
};

int vs_workspace_partials(vs_context* ctx, size_t bytes);
int vs_workspace_keep(vs_context* ctx, size_t bytes);
int vs_workspace_scratch(vs_context* ctx, size_t bytes);
int vs_workspace_fold(vs_context* ctx, size_t bytes);
// segments of a two-stage fold over P partial records (1: single stage, the order of rounds 1-2)
static inline uint32_t vs_fold_segments(uint32_t P) { return P >= 1024 ? 8u : P >= 512 ? 4u : 1u; }

int vs_fold_partials(vs_context* ctx, uint32_t P, int k, int l, uint64_t rows, const double* shift_dev,
                     int shift_stride, const double* f_dev, int rank, int max_ranks, double* acc_dev);

// vs_second.hip: fold of P partials [S S^T, T T^T][SB][SB] into the interface's Gram sums
int vs_gram_fold_st(vs_context* ctx, const double* partials, uint32_t P, int K, int SB, double* gram_dev);

// partial record of one workgroup for one output: 5 head values + 4 per block j
#define VS_P_SAB 0
#define VS_P_SA 1
#define VS_P_SB 2
#define VS_P_S1 3
#define VS_P_S2 4
#define VS_P_HEAD 5
#define VS_PLEN(k) (VS_P_HEAD + 4 * (size_t)(k))

// ---- device helpers (wave sums, variance shift): vs_device_gen.h, shared with the hiprtc kernels
#ifdef __HIPCC__
#include "vs_device_gen.h"
#endif
