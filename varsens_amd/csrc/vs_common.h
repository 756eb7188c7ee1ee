// Internal declarations shared by the translation units of libvarsens_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "varsens_hip.h"

struct vs_prof_slot {
    int kernel_id;
    hipEvent_t start, stop;
};

struct vs_context {
    int device = -1;
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
    uint32_t* counters = nullptr;  // 64 x u32 scratch counters (device)
    uint32_t* counters_host = nullptr;  // pinned mirror
    // profiling
    bool profiling = false;
    std::vector<vs_prof_slot> prof;
    std::vector<vs_prof_slot> prof_pool;
    size_t prof_capacity = 0;
    // RCCL
    void* comm = nullptr;
    int comm_ranks = 0;
    int comm_rank = -1;
};

struct vs_design {
    vs_design_desc d;          // host copy (pointers below replace the host ones)
    double* lower = nullptr;   // device, k (POWER: lower; LINEAR: lower)
    double* width = nullptr;   // device, k (LINEAR: upper-lower; POWER: upper/lower)
    uint32_t* perm = nullptr;  // device, n or null
    double* raw = nullptr;     // device (2n,k) or null
    uint32_t* dirs = nullptr;  // device (k,32) or null
    uint32_t* sobol_tabs = nullptr;  // device (k,4,256): XOR of the direction integers selected by each byte value
    uint32_t* bases = nullptr; // device, k x {base, magic, shift, pad}
};

struct vs_objective {
    int kind = 0;
    int k = 0;
    int l = 1;
    double* params = nullptr;  // device
    size_t nparams = 0;
    void* module = nullptr;    // hipModule_t of a run-time compiled objective
    void* function = nullptr;  // hipFunction_t (vs_user_eval)
    bool scratch_tile = false; // compiled for the HBM scratch tile (k beyond the LDS budget)
};

int vs_rtc_build(vs_context* ctx, vs_objective* o, const char* body, int fold);
void vs_rtc_release(vs_objective* o);
int vs_rtc_launch(vs_context* ctx, const vs_objective* o, const void* eval_args, size_t eval_args_bytes,
                  unsigned blocks, size_t lds);

int vs_fail(int code, const char* fmt, ...);

#define VS_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess)                                                              \
            return vs_fail(VS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                           __FILE__, __LINE__);                                             \
    } while (0)

#define VS_REQUIRE(cond, ...)                        \
    do {                                             \
        if (!(cond)) return vs_fail(VS_ERR_ARG, __VA_ARGS__); \
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

int vs_fold_partials(vs_context* ctx, uint32_t P, int k, int l, uint64_t rows, const double* shift_dev,
                     const double* f_dev, int rank, int max_ranks, double* acc_dev);

// partial record of one workgroup for one output: 5 head values + 4 per block j
#define VS_P_SAB 0
#define VS_P_SA 1
#define VS_P_SB 2
#define VS_P_S1 3
#define VS_P_S2 4
#define VS_P_HEAD 5
#define VS_PLEN(k) (VS_P_HEAD + 4 * (size_t)(k))

// ---- device helpers -------------------------------------------------------
#ifdef __HIPCC__
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
__device__ __forceinline__ double vs_shift_estimate(const double* a, uint64_t rows);

// Sum over the 64 lanes of a wavefront; result valid in every lane.
__device__ __forceinline__ double vs_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double vs_shift_estimate(const double* a, uint64_t rows) {
    const int lane = threadIdx.x & 63;
    const uint64_t m = rows < 1024 ? rows : 1024;
    double s = 0.0;
    for (uint64_t i = lane; i < m; i += 64) s += a[i];
    s = vs_wave_sum_dpp(s);  // total in lanes 48..63
    const int lo = __builtin_amdgcn_readlane(__double2loint(s), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(s), 63);
    return m ? __hiloint2double(hi, lo) / (double)m : 0.0;
}
#endif
