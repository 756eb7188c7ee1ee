// Context, memory, error and profiling plumbing of libvarsens_hip.so.
#include <cstdlib>
#include <memory>

#include "vs_common.h"

// vs_fail / vs_last_error / vs_abi_version, the seed-1 shuffle and the descriptor checks: vs_host.cpp (host only)

extern "C" int vs_device_count(int* count) {
    VS_GUARD_BEGIN
    VS_REQUIRE(count, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return vs_fail(VS_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return VS_OK;
    VS_GUARD_END
}

// Everything a context owns is released here, whatever subset of it exists: vs_context_create fails by deleting the
// half-built object, vs_context_destroy by deleting the whole one.
vs_context::~vs_context() {
    if (device >= 0) hipSetDevice(device);
    if (streams[0]) hipStreamSynchronize(streams[0]);
    if (streams[1]) hipStreamSynchronize(streams[1]);
    vs_staging_release(this);
    for (int i = 0; i < 8; ++i)
        if (events[i]) hipEventDestroy(events[i]);
    for (auto& s : prof_pool) {
        if (s.start) hipEventDestroy(s.start);
        if (s.stop) hipEventDestroy(s.stop);
    }
    if (partials) hipFree(partials);
    if (keep) hipFree(keep);
    if (fold_scratch) hipFree(fold_scratch);
    if (scratch) hipFree(scratch);
    if (counters) hipFree(counters);
    if (counters_host) hipHostFree(counters_host);
    for (auto& e : perm_cache)
        if (e.dev) hipFree(e.dev);
    if (streams[0]) hipStreamDestroy(streams[0]);
    if (streams[1]) hipStreamDestroy(streams[1]);
}

extern "C" int vs_context_create(int device, vs_context** out) {
    VS_GUARD_BEGIN
    VS_REQUIRE(out, "out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return vs_fail(VS_ERR_NODEVICE, "no HIP device available (%s); libvarsens_hip has no host fallback",
                       e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    VS_REQUIRE(device >= 0 && device < n, "device %d out of range [0,%d)", device, n);
    VS_HIP(hipSetDevice(device));
    // owned by this scope until the last step succeeded: every early return below (VS_HIP, VS_REQUIRE, an exception
    // caught by the guard) runs ~vs_context on whatever exists by then -- streams, events, counter buffers
    std::unique_ptr<vs_context> ctx(new vs_context());
    ctx->device = device;
    VS_HIP(hipGetDeviceProperties(&ctx->prop, device));
    if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0)
        return vs_fail(VS_ERR_NODEVICE, "device %d is %.48s; this library holds gfx950 code only", device,
                       ctx->prop.gcnArchName);
    VS_HIP(hipStreamCreateWithFlags(&ctx->streams[0], hipStreamNonBlocking));
    VS_HIP(hipStreamCreateWithFlags(&ctx->streams[1], hipStreamNonBlocking));
    ctx->stream = ctx->streams[0];
    {   // the only place the environment is read.  The knobs are measurement tooling (tools/, a few tests): they
        // count only under VS_TUNING=1, so that a stray VS_* variable in a user's shell cannot change kernels, and
        // a value outside a knob's range keeps the default (both said once on stderr)
        const char* gate = getenv("VS_TUNING");
        const bool tuning = gate && strtol(gate, nullptr, 0) == 1;
        bool ignored = false;
        auto knob = [&](const char* name, long dflt, long lo, long hi) {
            const char* v = getenv(name);
            if (!v || !*v) return dflt;
            if (!tuning) {
                ignored = true;
                return dflt;
            }
            char* end = nullptr;
            const long x = strtol(v, &end, 0);
            if (end == v || *end || x < lo || x > hi) {
                fprintf(stderr, "libvarsens_hip: %s=%s is outside [%ld, %ld]; keeping %ld\n", name, v, lo, hi, dflt);
                return dflt;
            }
            return x;
        };
        vs_tuning& t = ctx->tune;
        t.reduce_nv = (int)knob("VS_REDUCE_NV", t.reduce_nv, 2, 8);
        if (t.reduce_nv != 2 && t.reduce_nv != 4 && t.reduce_nv != 8) t.reduce_nv = 4;
        t.reduce_wgs_per_cu = (int)knob("VS_REDUCE_WGS_PER_CU", t.reduce_wgs_per_cu, 1, 8);
        t.eval_scratch = (int)knob("VS_EVAL_SCRATCH", t.eval_scratch, 0, 1);
        t.eval_waves = (int)knob("VS_EVAL_WAVES_PER_WG", t.eval_waves, 0, 8);
        if (t.eval_waves != 0 && t.eval_waves != 4 && t.eval_waves != 8) t.eval_waves = 0;
        t.eval_mode = (int)knob("VS_EVAL_MODE", t.eval_mode, 0, 2);
        t.fused_wgs_per_cu = (int)knob("VS_FUSED_WGS_PER_CU", t.fused_wgs_per_cu, 0, 8);
        t.fused_both_wide = (int)knob("VS_FUSED_BOTH_WIDE", t.fused_both_wide, 0, 2);
        // shapes are L*100+W; one that is not instantiated makes the launch fail with VS_ERR_ARG naming the knob
        t.fused_variant = (int)knob("VS_FUSED_VARIANT", t.fused_variant, 0, 3216);
        t.scan_shape = (int)knob("VS_SCAN_SHAPE", t.scan_shape, 0, 3216);
        t.scan_general = (int)knob("VS_SCAN_GENERAL", t.scan_general, 0, 1);
        t.gram_wgs_per_cu = (int)knob("VS_GRAM_WGS_PER_CU", t.gram_wgs_per_cu, 0, 8);
        t.gram_wide = (int)knob("VS_GRAM_WIDE", t.gram_wide, 0, 1);
        t.gram_st = (int)knob("VS_GRAM_ST", t.gram_st, 0, 1);
        t.gram_dma = (int)knob("VS_GRAM_DMA", t.gram_dma, 0, 2);
        t.reduce_both = (int)knob("VS_REDUCE_BOTH", t.reduce_both, 0, 1);
        t.copy_threads = (int)knob("VS_COPY_THREADS", t.copy_threads, 1, 16);
        t.f_shift = (int)knob("VS_F_SHIFT", t.f_shift, VS_F_SHIFT_MIN, 40);
        t.profile_mask = knob("VS_PROFILE_MASK", t.profile_mask, -1, 0x7fffffffL);
        static bool told = false;
        if (ignored && !told) {
            told = true;
            fprintf(stderr, "libvarsens_hip: VS_* tuning variables are set but VS_TUNING=1 is not: ignored\n");
        }
    }
    for (int i = 0; i < 8; ++i) VS_HIP(hipEventCreateWithFlags(&ctx->events[i], hipEventDisableTiming));
    VS_HIP(hipMalloc(&ctx->counters, 64 * sizeof(uint32_t)));
    VS_HIP(hipHostMalloc(&ctx->counters_host, 64 * sizeof(uint32_t)));
    VS_HIP(hipMemset(ctx->counters, 0, 64 * sizeof(uint32_t)));   // tickets that folds hand back at zero
    *out = ctx.release();
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_context_destroy(vs_context* ctx) {
    VS_GUARD_BEGIN
    if (!ctx) return VS_OK;
    // a design keeps device tables of this context and, with VS_DESIGN_SEED1_SHUFFLE, a share of its shuffle cache
    if (ctx->live_designs > 0)
        return vs_fail(VS_ERR_STATE, "vs_context_destroy: %d design(s) of this context are still alive; destroy them first",
                       ctx->live_designs);
    hipSetDevice(ctx->device);
    if (ctx->comm) vs_comm_destroy(ctx);
    delete ctx;
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_context_info(vs_context* ctx, vs_device_info* out) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && out, "null argument");
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", ctx->prop.name);
    snprintf(out->arch, sizeof(out->arch), "%s", ctx->prop.gcnArchName);
    out->compute_units = ctx->prop.multiProcessorCount;
    out->wavefront_size = ctx->prop.warpSize;
    out->lds_bytes_per_cu = (int32_t)ctx->prop.maxSharedMemoryPerMultiProcessor;
    out->clock_mhz = ctx->prop.clockRate / 1000;
    out->hbm_bytes = ctx->prop.totalGlobalMem;
    out->f_shift = ctx->tune.f_shift;
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_context_set_eval_order(vs_context* ctx, int order) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx, "ctx is null");
    VS_REQUIRE(order == VS_ORDER_FAST || order == VS_ORDER_STRICT, "order must be VS_ORDER_FAST or VS_ORDER_STRICT");
    ctx->tune.eval_mode = order == VS_ORDER_STRICT ? 1 : 0;
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_sync(vs_context* ctx) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx, "ctx is null");
    VS_HIP(hipStreamSynchronize(ctx->streams[0]));
    VS_HIP(hipStreamSynchronize(ctx->streams[1]));
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_stream_select(vs_context* ctx, int which) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && (which == 0 || which == 1), "stream must be 0 (main) or 1 (auxiliary)");
    ctx->stream = ctx->streams[which];
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_event_record(vs_context* ctx, int slot, int which) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && slot >= 0 && slot < 8 && (which == 0 || which == 1), "bad slot or stream");
    VS_HIP(hipEventRecord(ctx->events[slot], ctx->streams[which]));
    ctx->event_set[slot] = true;
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_stream_wait_event(vs_context* ctx, int which, int slot) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && slot >= 0 && slot < 8 && (which == 0 || which == 1), "bad slot or stream");
    if (!ctx->event_set[slot]) return VS_OK;  // nothing recorded yet: nothing to wait for
    VS_HIP(hipStreamWaitEvent(ctx->streams[which], ctx->events[slot], 0));
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_alloc(vs_context* ctx, size_t bytes, void** dptr) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && dptr, "null argument");
    *dptr = nullptr;
    if (bytes == 0) bytes = 8;
    VS_HIP(hipSetDevice(ctx->device));
    VS_HIP(hipMalloc(dptr, bytes));
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_free(vs_context* ctx, void* dptr) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx, "ctx is null");
    if (!dptr) return VS_OK;
    VS_HIP(hipStreamSynchronize(ctx->streams[0]));
    VS_HIP(hipStreamSynchronize(ctx->streams[1]));
    VS_HIP(hipFree(dptr));
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_upload(vs_context* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && (bytes == 0 || (dst_dev && src_host)), "null argument");
    if (!bytes) return VS_OK;
    VS_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    VS_HIP(hipStreamSynchronize(ctx->stream));  // the host buffer is not retained
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_download(vs_context* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && (bytes == 0 || (dst_host && src_dev)), "null argument");
    if (!bytes) return VS_OK;
    VS_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VS_HIP(hipStreamSynchronize(ctx->stream));
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_memset(vs_context* ctx, void* dst_dev, int byte, size_t bytes) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && (bytes == 0 || dst_dev), "null argument");
    if (!bytes) return VS_OK;
    VS_HIP(hipMemsetAsync(dst_dev, byte, bytes, ctx->stream));
    return VS_OK;
    VS_GUARD_END
}

// dst[i] += src[i]: sums the accumulator records / Gram sums of streamed row ranges on the device
__global__ void __launch_bounds__(256) accumulate_kernel(double* __restrict__ dst, const double* __restrict__ src,
                                                         size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        dst[i] += src[i];
}

extern "C" int vs_accumulate(vs_context* ctx, double* dst_dev, const double* src_dev, size_t count) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && (count == 0 || (dst_dev && src_dev)), "null argument");
    if (!count) return VS_OK;
    VS_HIP(hipSetDevice(ctx->device));
    const size_t blocks = (count + 255) / 256;
    vs_launch_scope scope(ctx, VS_KERNEL_FOLD);
    hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, ctx->stream,
                       dst_dev, src_dev, count);
    VS_HIP(hipGetLastError());
    return VS_OK;
    VS_GUARD_END
}

int vs_workspace_partials(vs_context* ctx, size_t bytes) {
    if (bytes <= ctx->partials_bytes) return VS_OK;
    if (ctx->partials) {   // kernels on either stream may still use the old block
        VS_HIP(hipStreamSynchronize(ctx->streams[0]));
        VS_HIP(hipStreamSynchronize(ctx->streams[1]));
        VS_HIP(hipFree(ctx->partials));
        ctx->partials = nullptr;
        ctx->partials_bytes = 0;
    }
    VS_HIP(hipMalloc(&ctx->partials, bytes));
    ctx->partials_bytes = bytes;
    return VS_OK;
}

int vs_workspace_keep(vs_context* ctx, size_t bytes) {
    if (bytes <= ctx->keep_bytes) return VS_OK;
    if (ctx->keep) {   // kernels on either stream may still use the old block
        VS_HIP(hipStreamSynchronize(ctx->streams[0]));
        VS_HIP(hipStreamSynchronize(ctx->streams[1]));
        VS_HIP(hipFree(ctx->keep));
        ctx->keep = nullptr;
        ctx->keep_bytes = 0;
    }
    VS_HIP(hipMalloc(&ctx->keep, bytes));
    ctx->keep_bytes = bytes;
    return VS_OK;
}

int vs_workspace_fold(vs_context* ctx, size_t bytes) {
    if (bytes <= ctx->fold_scratch_bytes) return VS_OK;
    if (ctx->fold_scratch) {
        VS_HIP(hipStreamSynchronize(ctx->streams[0]));
        VS_HIP(hipStreamSynchronize(ctx->streams[1]));
        VS_HIP(hipFree(ctx->fold_scratch));
        ctx->fold_scratch = nullptr;
        ctx->fold_scratch_bytes = 0;
    }
    VS_HIP(hipMalloc(&ctx->fold_scratch, bytes));
    ctx->fold_scratch_bytes = bytes;
    return VS_OK;
}

int vs_workspace_scratch(vs_context* ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return VS_OK;
    if (ctx->scratch) {
        VS_HIP(hipStreamSynchronize(ctx->streams[0]));
        VS_HIP(hipStreamSynchronize(ctx->streams[1]));
        VS_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    VS_HIP(hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return VS_OK;
}

// ---- profiling --------------------------------------------------------------

extern "C" int vs_profile_begin(vs_context* ctx, int capacity) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && capacity > 0, "bad argument");
    ctx->prof_pool.reserve((size_t)capacity);
    while (ctx->prof_pool.size() < (size_t)capacity) {
        vs_prof_slot s;
        s.kernel_id = 0;
        s.start = s.stop = nullptr;
        VS_HIP(hipEventCreate(&s.start));
        if (hipEventCreate(&s.stop) != hipSuccess) {
            hipEventDestroy(s.start);
            return vs_fail(VS_ERR_HIP, "hipEventCreate failed");
        }
        ctx->prof_pool.push_back(s);   // (reserved above)
    }
    ctx->prof.clear();
    ctx->prof.reserve((size_t)capacity);   // vs_launch_scope appends inside launches: it must not allocate there
    ctx->prof_capacity = (size_t)capacity;
    ctx->profiling = true;
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_profile_select(vs_context* ctx, uint32_t kernel_mask) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx, "ctx is null");
    VS_REQUIRE(!ctx->profiling, "vs_profile_select between vs_profile_begin and vs_profile_end");
    ctx->profile_select = kernel_mask;
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_profile_end(vs_context* ctx, vs_profile_record* out, int capacity, int* count) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && count, "null argument");
    ctx->profiling = false;
    VS_HIP(hipStreamSynchronize(ctx->streams[0]));
    VS_HIP(hipStreamSynchronize(ctx->streams[1]));
    int n = 0;
    for (auto& s : ctx->prof) {
        if (n >= capacity || !out) break;
        float ms = 0.f;
        VS_HIP(hipEventElapsedTime(&ms, s.start, s.stop));
        out[n].kernel_id = s.kernel_id;
        out[n].ms = ms;
        ++n;
    }
    *count = n;
    ctx->prof.clear();
    return VS_OK;
    VS_GUARD_END
}

vs_launch_scope::vs_launch_scope(vs_context* c, int kernel_id) : ctx(c) {
    if (!ctx->profiling || ctx->prof.size() >= ctx->prof_capacity) return;
    if (!((ctx->tune.profile_mask >> kernel_id) & 1)) return;  // tuning aid: bracket only some kernel ids
    if (ctx->profile_select && !((ctx->profile_select >> kernel_id) & 1u)) return;   // vs_profile_select
    slot = (int)ctx->prof.size();
    vs_prof_slot s = ctx->prof_pool[slot];
    s.kernel_id = kernel_id;
    ctx->prof.push_back(s);
    hipEventRecord(s.start, ctx->stream);
}

vs_launch_scope::~vs_launch_scope() {
    if (slot >= 0) hipEventRecord(ctx->prof[slot].stop, ctx->stream);
}
