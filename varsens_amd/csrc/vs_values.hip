// Objective values and sample rows crossing the host boundary: the routes of a reference user whose model the engine
// cannot see into (call stacks (b) and (c)).
//
// Reference counterparts (paths relative to the reference checkout):
//   Objective.load: slicing of the flat (2n(k+1), l) values, `/ scaling`, NaN rows   varsens/saltelli.py:449-495
//   Objective.flat                                                                    varsens/saltelli.py:357-391
//   sample.M_1[i] .. sample.N_nj[i][j] handed to the objective, row by row           varsens/saltelli.py:308-355
//
// The host keeps the reference's layouts -- values (N, l) row-major in flat block order, sample rows (rows, k) -- and
// the device its own ("planar", include/varsens_hip.h): every crossing goes through a conversion kernel here, so the
// host never transposes, divides or scans anything itself.
//
// How the bytes move (tools/pcie_modes.hip, profiles/r05/pcie_modes.txt): one hipMemcpy from pageable memory runs at
// the link's 56 GB/s only when the runtime has the pages pinned from an earlier copy; a fresh array goes at 25-31 GB/s
// (pinning 2.2 GB: 98 ms), in 32 MB pieces at 16-23 GB/s, and a copy INTO untouched pages at 13 GB/s.  A ring of
// pinned slots fed (or drained) by `copy_threads` host threads while the copy engine works on the other slots holds
// 47-52 GB/s for any caller memory: that is what every path below uses.
//   host -> device: threads memcpy piece i into pinned slot s; DMA slot s -> device stage s (or straight to the
//                   destination when nothing has to be transposed or divided); kernel: stage s -> planes.
//   device -> host: kernel: planes (or the design) -> device stage s; DMA -> pinned slot s; threads memcpy out.
// Copies run on the context's OTHER stream, kernels on the selected one; events order the two, so the kernel of piece
// i + 1 runs under the DMA of piece i.
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "vs_common.h"

#include "vs_device_gen.h"

// ---------------------------------------------------------------------------------------------
// host threads of the pinned ring
// ---------------------------------------------------------------------------------------------
namespace {

// `workers` threads sleeping on a condition variable; run(fn) executes fn(t, T) for t = 0 .. T-1 (the caller is
// t = 0) and returns when all are through.  One job at a time (the context is not thread-safe anyway).
class copy_pool {
public:
    explicit copy_pool(int threads) {
        const int extra = threads > 1 ? threads - 1 : 0;
        workers_.reserve((size_t)extra);
        try {
            for (int i = 0; i < extra; ++i) workers_.emplace_back([this, i] { loop(i + 1); });
        } catch (...) {   // a thread could not be started: keep the ones that exist
        }
    }
    ~copy_pool() {
        {
            std::lock_guard<std::mutex> hold(lock_);
            quit_ = true;
        }
        wake_.notify_all();
        for (auto& t : workers_) t.join();
    }
    int size() const { return (int)workers_.size() + 1; }
    void run(const std::function<void(int, int)>& fn) {
        if (workers_.empty()) {
            fn(0, 1);
            return;
        }
        {
            std::lock_guard<std::mutex> hold(lock_);
            job_ = &fn;
            pending_ = (int)workers_.size();
            ++generation_;
        }
        wake_.notify_all();
        fn(0, size());
        std::unique_lock<std::mutex> hold(lock_);
        done_.wait(hold, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    void loop(int me) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int, int)>* job;
            {
                std::unique_lock<std::mutex> hold(lock_);
                wake_.wait(hold, [&] { return quit_ || generation_ != seen; });
                if (quit_) return;
                seen = generation_;
                job = job_;
            }
            (*job)(me, size());
            {
                std::lock_guard<std::mutex> hold(lock_);
                --pending_;
            }
            done_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex lock_;
    std::condition_variable wake_, done_;
    const std::function<void(int, int)>* job_ = nullptr;
    uint64_t generation_ = 0;
    int pending_ = 0;
    bool quit_ = false;
};

void threaded_memcpy(copy_pool* pool, void* dst, const void* src, size_t bytes) {
    if (bytes < (1u << 20) || pool->size() == 1) {
        memcpy(dst, src, bytes);
        return;
    }
    pool->run([&](int t, int T) {   // (one reference captured: the std::function needs no allocation)
        const size_t per = ((bytes + T - 1) / T + 4095) & ~(size_t)4095;
        const size_t lo = (size_t)t * per, hi = std::min(bytes, lo + per);
        if (lo < hi) memcpy((char*)dst + lo, (const char*)src + lo, hi - lo);
    });
}

constexpr int VS_RING_SLOTS = 3;
constexpr size_t VS_RING_SLOT_BYTES = 16u << 20;

}  // namespace

struct vs_staging {
    copy_pool* pool = nullptr;
    char* pinned = nullptr;      // VS_RING_SLOTS x VS_RING_SLOT_BYTES, hipHostMalloc
    char* dev = nullptr;         // the same on the device
    hipEvent_t dma_done[VS_RING_SLOTS] = {};     // the copy engine is through with slot s (pinned side free / filled)
    hipEvent_t kernel_done[VS_RING_SLOTS] = {};  // the conversion kernel is through with device stage s
    hipEvent_t entry = nullptr;                  // orders the copy stream behind earlier work of the kernel stream
    unsigned long long* count_dev = nullptr;     // NaN-row counter
    unsigned long long* count_host = nullptr;    // pinned
    // vs_sample_rows_begin / _wait: two independent slots of their own, grown on demand
    char* rows_pinned[2] = {nullptr, nullptr};
    char* rows_dev[2] = {nullptr, nullptr};
    size_t rows_bytes[2] = {0, 0};
    hipEvent_t rows_kernel[2] = {};
    hipEvent_t rows_done[2] = {};
    bool rows_pending[2] = {false, false};
};

void vs_staging_release(vs_context* ctx) {
    vs_staging* s = ctx->staging;
    if (!s) return;
    delete s->pool;
    if (s->pinned) hipHostFree(s->pinned);
    if (s->dev) hipFree(s->dev);
    for (int i = 0; i < VS_RING_SLOTS; ++i) {
        if (s->dma_done[i]) hipEventDestroy(s->dma_done[i]);
        if (s->kernel_done[i]) hipEventDestroy(s->kernel_done[i]);
    }
    if (s->entry) hipEventDestroy(s->entry);
    if (s->count_dev) hipFree(s->count_dev);
    if (s->count_host) hipHostFree(s->count_host);
    for (int i = 0; i < 2; ++i) {
        if (s->rows_pinned[i]) hipHostFree(s->rows_pinned[i]);
        if (s->rows_dev[i]) hipFree(s->rows_dev[i]);
        if (s->rows_kernel[i]) hipEventDestroy(s->rows_kernel[i]);
        if (s->rows_done[i]) hipEventDestroy(s->rows_done[i]);
    }
    delete s;
    ctx->staging = nullptr;
}

// the ring, its events and threads: made on the first call that needs them.  A failure half-way leaves the members
// that exist in ctx->staging; the context's destructor (or the next call) deals with them.
static int staging_ready(vs_context* ctx, vs_staging** out) {
    if (!ctx->staging) ctx->staging = new vs_staging();
    vs_staging* s = ctx->staging;
    if (!s->pool) s->pool = new copy_pool(ctx->tune.copy_threads);
    if (!s->pinned) VS_HIP(hipHostMalloc((void**)&s->pinned, VS_RING_SLOTS * VS_RING_SLOT_BYTES, hipHostMallocDefault));
    if (!s->dev) VS_HIP(hipMalloc((void**)&s->dev, VS_RING_SLOTS * VS_RING_SLOT_BYTES));
    for (int i = 0; i < VS_RING_SLOTS; ++i) {
        if (!s->dma_done[i]) VS_HIP(hipEventCreateWithFlags(&s->dma_done[i], hipEventDisableTiming));
        if (!s->kernel_done[i]) VS_HIP(hipEventCreateWithFlags(&s->kernel_done[i], hipEventDisableTiming));
    }
    if (!s->entry) VS_HIP(hipEventCreateWithFlags(&s->entry, hipEventDisableTiming));
    if (!s->count_dev) VS_HIP(hipMalloc((void**)&s->count_dev, sizeof(unsigned long long)));
    if (!s->count_host) VS_HIP(hipHostMalloc((void**)&s->count_host, sizeof(unsigned long long), hipHostMallocDefault));
    for (int i = 0; i < 2; ++i) {
        if (!s->rows_kernel[i]) VS_HIP(hipEventCreateWithFlags(&s->rows_kernel[i], hipEventDisableTiming));
        if (!s->rows_done[i]) VS_HIP(hipEventCreateWithFlags(&s->rows_done[i], hipEventDisableTiming));
    }
    *out = s;
    return VS_OK;
}

static hipStream_t other_stream(vs_context* ctx) {
    return ctx->stream == ctx->streams[0] ? ctx->streams[1] : ctx->streams[0];
}

// ---------------------------------------------------------------------------------------------
// conversion kernels
// ---------------------------------------------------------------------------------------------
// The reference's order: value (i, o) at src[i * l + o], i = flat row of the shard = block * rows + row.  The device's:
// row-blocked planes (vs_f_tile, vs_device_gen.h): element (plane o (2k+2) + block, row).  A workgroup turns tiles of
// VS_TR flat rows x VS_TC outputs through LDS: the global reads (runs of min(l, VS_TC) doubles per row, rows adjacent in
// memory) and the writes (runs of consecutive rows of one plane, up to the end of a block or row block) are contiguous.
// `divide` (null, 1 or l entries): Objective.load's `x / scaling` (saltelli.py:451-468), a true division -- the
// quotient's bits are numpy's.
#define VS_TR 128
#define VS_TC 32

struct vs_values_shape {
    uint64_t rows;     // rows of the shard
    uint64_t planes;   // l (2k+2)
    int blocks;        // 2k+2
    int l;
    int f_shift;
};

// doubles from f to the value of output o at flat row i of the shard
__device__ __forceinline__ uint64_t values_at(const vs_values_shape& S, uint64_t i, int o) {
    const uint64_t b = i / S.rows, r = i - b * S.rows;
    return vs_f_index(S.rows, S.planes, S.f_shift, (uint64_t)o * S.blocks + b, r);
}

__global__ void __launch_bounds__(256)
values_in_kernel(const double* __restrict__ src, double* __restrict__ f, vs_values_shape S, uint64_t i_first, uint64_t cnt,
                 const double* __restrict__ divide, int ndivide, double divisor) {
    __shared__ double tile[VS_TC][VS_TR + 1];
    const int l = S.l;
    const uint64_t row_tiles = (cnt + VS_TR - 1) / VS_TR;
    const int col_tiles = (l + VS_TC - 1) / VS_TC;
    const uint64_t tiles = row_tiles * (uint64_t)col_tiles;
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint64_t i0 = (t / col_tiles) * VS_TR;
        const int o0 = (int)(t % col_tiles) * VS_TC;
        const int tc = min(VS_TC, l - o0);
        const int tr = (int)min((uint64_t)VS_TR, cnt - i0);
        for (int e = threadIdx.x; e < tr * tc; e += 256) {
            const int r = e / tc, c = e - r * tc;
            tile[c][r] = __builtin_nontemporal_load(src + (i0 + r) * (uint64_t)l + o0 + c);
        }
        __syncthreads();
        for (int e = threadIdx.x; e < tr * tc; e += 256) {
            const int c = e / tr, r = e - c * tr;
            double v = tile[c][r];
            if (divide) v = v / divide[ndivide == 1 ? 0 : o0 + c];
            else if (ndivide) v = v / divisor;    // one output: the divisor is a kernel argument
            f[values_at(S, i_first + i0 + r, o0 + c)] = v;
        }
        __syncthreads();
    }
}

// device layout -> the reference's order (Objective.flat, saltelli.py:357-391)
__global__ void __launch_bounds__(256)
values_out_kernel(const double* __restrict__ f, double* __restrict__ dst, vs_values_shape S, uint64_t i_first, uint64_t cnt) {
    __shared__ double tile[VS_TC][VS_TR + 1];
    const int l = S.l;
    const uint64_t row_tiles = (cnt + VS_TR - 1) / VS_TR;
    const int col_tiles = (l + VS_TC - 1) / VS_TC;
    const uint64_t tiles = row_tiles * (uint64_t)col_tiles;
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint64_t i0 = (t / col_tiles) * VS_TR;
        const int o0 = (int)(t % col_tiles) * VS_TC;
        const int tc = min(VS_TC, l - o0);
        const int tr = (int)min((uint64_t)VS_TR, cnt - i0);
        for (int e = threadIdx.x; e < tr * tc; e += 256) {
            const int c = e / tr, r = e - c * tr;
            tile[c][r] = f[values_at(S, i_first + i0 + r, o0 + c)];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < tr * tc; e += 256) {
            const int r = e / tc, c = e - r * tc;
            dst[(i0 + r) * (uint64_t)l + o0 + c] = tile[c][r];
        }
        __syncthreads();
    }
}

// rows of the shard that Objective.load deletes (saltelli.py:474-486): a NaN in output 0 of any of the 2k+2 blocks
__global__ void __launch_bounds__(256)
nan_rows_kernel(const double* __restrict__ f, vs_values_shape S, unsigned long long* __restrict__ count) {
    unsigned int bad_rows = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < S.rows; r += (uint64_t)gridDim.x * 256) {
        bool bad = false;
        const vs_ftile ft = vs_f_tile(S.rows, S.planes, S.f_shift, r);
        for (int b = 0; b < S.blocks; ++b) {
            const double x = f[ft.base + (uint64_t)b * ft.pitch];
            bad |= (x != x);
        }
        bad_rows += bad ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) bad_rows += __shfl_xor(bad_rows, off, 64);
    if ((threadIdx.x & 63) == 0 && bad_rows) atomicAdd(count, (unsigned long long)bad_rows);
}

static unsigned blocks_for(uint64_t work_items, uint64_t per_block, unsigned cap) {
    uint64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

// ---------------------------------------------------------------------------------------------
// a7: values from the host
// ---------------------------------------------------------------------------------------------
namespace {
struct piece {
    uint64_t host_off;  // doubles from the start of the host array
    uint64_t i0;        // first flat row of the shard's (2k+2) * rows
    uint64_t cnt;       // flat rows
};

// The shard's flat rows in pieces that fit a ring slot.  rows == n: the host array is one contiguous run of
// (2k+2) n rows; a true shard: one run of `rows` rows per block.
int make_pieces(uint64_t n, int k, int l, uint64_t row_begin, uint64_t rows, bool whole, std::vector<piece>* out) {
    const uint64_t per = std::max<uint64_t>(1, VS_RING_SLOT_BYTES / (8 * (uint64_t)l));
    const uint64_t blocks = 2 * (uint64_t)k + 2;
    if (whole) {
        const uint64_t total = blocks * rows;
        out->reserve((size_t)((total + per - 1) / per));
        for (uint64_t i = 0; i < total; i += per) out->push_back({i * l, i, std::min(per, total - i)});
    } else {
        out->reserve((size_t)(blocks * ((rows + per - 1) / per)));
        for (uint64_t b = 0; b < blocks; ++b)
            for (uint64_t r = 0; r < rows; r += per)
                out->push_back({((b * n + row_begin + r) * (uint64_t)l), b * rows + r, std::min(per, rows - r)});
    }
    return VS_OK;
}
}  // namespace

extern "C" int vs_upload_values(vs_context* ctx, const double* host_vals, uint64_t n, int k, int l, uint64_t row_begin,
                                uint64_t rows, const double* divide, int ndivide, double* f_dev, uint64_t* nan_rows) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx, "ctx is null");
    VS_REQUIRE(k >= 1 && l >= 1 && n >= 1, "bad dimensions");
    VS_REQUIRE(row_begin <= n && rows <= n - row_begin, "rows [%llu,+%llu) exceed n=%llu",
               (unsigned long long)row_begin, (unsigned long long)rows, (unsigned long long)n);
    VS_REQUIRE(!divide || ndivide == 1 || ndivide == l, "divide must hold 1 or l=%d divisors", l);
    if (nan_rows) *nan_rows = 0;
    if (rows == 0) return VS_OK;
    VS_REQUIRE(host_vals && f_dev, "null buffer");
    VS_HIP(hipSetDevice(ctx->device));
    vs_staging* S = nullptr;
    int rc = staging_ready(ctx, &S);
    if (rc) return rc;
    std::vector<piece> pieces;
    rc = make_pieces(n, k, l, row_begin, rows, rows == n, &pieces);
    if (rc) return rc;

    const hipStream_t ks = ctx->stream, cs = other_stream(ctx);
    VS_HIP(hipEventRecord(S->entry, ks));        // whatever still uses f_dev on the kernel stream comes first
    VS_HIP(hipStreamWaitEvent(cs, S->entry, 0));
    double* divide_dev = nullptr;
    if (divide && l > 1) {   // per-output divisors for the layout kernel (one output: a kernel argument)
        VS_HIP(hipMalloc((void**)&divide_dev, sizeof(double) * ndivide));
        const hipError_t e = hipMemcpy(divide_dev, divide, sizeof(double) * ndivide, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            hipFree(divide_dev);
            return vs_fail(VS_ERR_HIP, "upload of the divisors failed: %s", hipGetErrorString(e));
        }
    }
    // one output, no divisor, ONE row block: flat order == device order, the copy engine writes the destination itself
    const bool direct = l == 1 && !divide && rows <= (1ull << ctx->tune.f_shift);
    vs_values_shape shape;
    shape.rows = rows;
    shape.planes = (uint64_t)l * (2 * (uint64_t)k + 2);
    shape.blocks = 2 * k + 2;
    shape.l = l;
    shape.f_shift = ctx->tune.f_shift;
    bool used[VS_RING_SLOTS] = {};
    auto body = [&]() -> int {
        for (size_t p = 0; p < pieces.size(); ++p) {
            const piece& P = pieces[p];
            const int s = (int)(p % VS_RING_SLOTS);
            const size_t bytes = sizeof(double) * P.cnt * l;
            char* pin = S->pinned + (size_t)s * VS_RING_SLOT_BYTES;
            char* stage = S->dev + (size_t)s * VS_RING_SLOT_BYTES;
            if (used[s]) VS_HIP(hipEventSynchronize(S->dma_done[s]));   // the slot's last copy has left the pinned side
            threaded_memcpy(S->pool, pin, host_vals + P.host_off, bytes);
            if (direct) {
                VS_HIP(hipMemcpyAsync(f_dev + P.i0, pin, bytes, hipMemcpyHostToDevice, cs));
                VS_HIP(hipEventRecord(S->dma_done[s], cs));
            } else {
                if (used[s]) VS_HIP(hipStreamWaitEvent(cs, S->kernel_done[s], 0));   // stage s has been consumed
                VS_HIP(hipMemcpyAsync(stage, pin, bytes, hipMemcpyHostToDevice, cs));
                VS_HIP(hipEventRecord(S->dma_done[s], cs));
                VS_HIP(hipStreamWaitEvent(ks, S->dma_done[s], 0));
                {
                    vs_launch_scope scope(ctx, VS_KERNEL_VALUES);
                    hipLaunchKernelGGL(values_in_kernel, dim3(blocks_for(P.cnt, VS_TR, 4096)), dim3(256), 0, ks,
                                       (const double*)stage, f_dev, shape, P.i0, P.cnt, (const double*)divide_dev,
                                       divide ? ndivide : 0, divide ? divide[0] : 1.0);
                }
                VS_HIP(hipGetLastError());
                VS_HIP(hipEventRecord(S->kernel_done[s], ks));
            }
            used[s] = true;
        }
        VS_HIP(hipStreamSynchronize(cs));
        if (direct) {   // the counting kernel below reads what the copy stream wrote
            VS_HIP(hipEventRecord(S->entry, cs));
            VS_HIP(hipStreamWaitEvent(ks, S->entry, 0));
        }
        if (nan_rows) {
            VS_HIP(hipMemsetAsync(S->count_dev, 0, sizeof(unsigned long long), ks));
            {
                vs_launch_scope scope(ctx, VS_KERNEL_NAN_MASK);
                hipLaunchKernelGGL(nan_rows_kernel, dim3(blocks_for(rows, 256, 4096)), dim3(256), 0, ks,
                                   (const double*)f_dev, shape, S->count_dev);
            }
            VS_HIP(hipGetLastError());
            VS_HIP(hipMemcpyAsync(S->count_host, S->count_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, ks));
        }
        VS_HIP(hipStreamSynchronize(ks));
        if (nan_rows) *nan_rows = (uint64_t)*S->count_host;
        return VS_OK;
    };
    rc = body();
    if (rc) {   // nothing of this call may still be in flight when the caller's buffers go away
        hipStreamSynchronize(cs);
        hipStreamSynchronize(ks);
    }
    if (divide_dev) hipFree(divide_dev);
    return rc;
    VS_GUARD_END
}

// ---------------------------------------------------------------------------------------------
// Objective.flat: values back to the host in the reference's order
// ---------------------------------------------------------------------------------------------
// A producer fills device stage s (or names a device range to copy from), the copy engine moves it into pinned slot s,
// the host threads move it on into the caller's memory VS_RING_SLOTS - 1 pieces later.
namespace {
struct out_piece {
    const char* direct_src;  // non-null: copy from here, no kernel
    size_t bytes;
    size_t host_off;         // bytes into the caller's array
};
}  // namespace

template <class Produce>
static int drain_to_host(vs_context* ctx, vs_staging* S, size_t npieces, char* host_out,
                         const std::function<out_piece(size_t)>& describe, Produce&& produce) {
    const hipStream_t ks = ctx->stream, cs = other_stream(ctx);
    out_piece held[VS_RING_SLOTS];
    auto finalize = [&](size_t p) -> int {
        const int s = (int)(p % VS_RING_SLOTS);
        VS_HIP(hipEventSynchronize(S->dma_done[s]));
        threaded_memcpy(S->pool, host_out + held[s].host_off, S->pinned + (size_t)s * VS_RING_SLOT_BYTES, held[s].bytes);
        return VS_OK;
    };
    VS_HIP(hipEventRecord(S->entry, ks));
    VS_HIP(hipStreamWaitEvent(cs, S->entry, 0));
    for (size_t p = 0; p < npieces; ++p) {
        const int s = (int)(p % VS_RING_SLOTS);
        if (p >= VS_RING_SLOTS) {
            const int rc = finalize(p - VS_RING_SLOTS);   // frees pinned slot s (and, its DMA done, device stage s)
            if (rc) return rc;
        }
        const out_piece P = describe(p);
        held[s] = P;
        char* stage = S->dev + (size_t)s * VS_RING_SLOT_BYTES;
        const char* from = P.direct_src;
        if (!from) {
            const int rc = produce(p, stage, ks);
            if (rc) return rc;
            VS_HIP(hipEventRecord(S->kernel_done[s], ks));
            VS_HIP(hipStreamWaitEvent(cs, S->kernel_done[s], 0));
            from = stage;
        }
        VS_HIP(hipMemcpyAsync(S->pinned + (size_t)s * VS_RING_SLOT_BYTES, from, P.bytes, hipMemcpyDeviceToHost, cs));
        VS_HIP(hipEventRecord(S->dma_done[s], cs));
    }
    for (size_t p = npieces > VS_RING_SLOTS ? npieces - VS_RING_SLOTS : 0; p < npieces; ++p) {
        const int rc = finalize(p);
        if (rc) return rc;
    }
    return VS_OK;
}

extern "C" int vs_download_values(vs_context* ctx, const double* f_dev, uint64_t rows, int k, int l, double* host_vals) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx, "ctx is null");
    VS_REQUIRE(k >= 1 && l >= 1, "bad dimensions");
    if (rows == 0) return VS_OK;
    VS_REQUIRE(host_vals && f_dev, "null buffer");
    VS_HIP(hipSetDevice(ctx->device));
    vs_staging* S = nullptr;
    int rc = staging_ready(ctx, &S);
    if (rc) return rc;
    const uint64_t total = (2 * (uint64_t)k + 2) * rows;
    const uint64_t per = std::max<uint64_t>(1, VS_RING_SLOT_BYTES / (8 * (uint64_t)l));
    const size_t npieces = (size_t)((total + per - 1) / per);
    vs_values_shape shape;
    shape.rows = rows;
    shape.planes = (uint64_t)l * (2 * (uint64_t)k + 2);
    shape.blocks = 2 * k + 2;
    shape.l = l;
    shape.f_shift = ctx->tune.f_shift;
    const bool direct = l == 1 && rows <= (1ull << ctx->tune.f_shift);   // one row block: device order == flat order
    const std::function<out_piece(size_t)> describe = [&](size_t p) {
        const uint64_t i0 = p * per, cnt = std::min(per, total - i0);
        return out_piece{direct ? (const char*)(f_dev + i0) : nullptr, (size_t)(8 * cnt * l), (size_t)(8 * i0 * l)};
    };
    rc = drain_to_host(ctx, S, npieces, (char*)host_vals, describe, [&](size_t p, char* stage, hipStream_t ks) -> int {
        const uint64_t i0 = p * per, cnt = std::min(per, total - i0);
        {
            vs_launch_scope scope(ctx, VS_KERNEL_VALUES);
            hipLaunchKernelGGL(values_out_kernel, dim3(blocks_for(cnt, VS_TR, 4096)), dim3(256), 0, ks, f_dev,
                               (double*)stage, shape, i0, cnt);
        }
        VS_HIP(hipGetLastError());
        return VS_OK;
    });
    hipStreamSynchronize(other_stream(ctx));
    hipStreamSynchronize(ctx->stream);
    return rc;
    VS_GUARD_END
}

// ---------------------------------------------------------------------------------------------
// sample rows for a model that runs on the host
// ---------------------------------------------------------------------------------------------
int vs_sample_flat_launch(vs_context* ctx, const vs_design* design, uint64_t row_begin, uint64_t row_count,
                          double* out_dev, hipStream_t stream);   // vs_generate.hip

static int check_flat_rows(const vs_design* design, uint64_t row_begin, uint64_t row_count) {
    const uint64_t rows_total = 2 * design->d.n * ((uint64_t)design->d.k + 1);
    VS_REQUIRE(row_begin <= rows_total && row_count <= rows_total - row_begin,
               "flat rows [%llu,+%llu) exceed 2n(k+1)=%llu", (unsigned long long)row_begin,
               (unsigned long long)row_count, (unsigned long long)rows_total);
    VS_REQUIRE(!design->perm || design->d.perm_count == design->d.n,
               "sample rows need the whole shuffle (this design holds a slice of %llu rows)",
               (unsigned long long)design->d.perm_count);
    return VS_OK;
}

extern "C" int vs_sample_rows(vs_context* ctx, const vs_design* design, uint64_t row_begin, uint64_t row_count,
                              double* host_rows) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && design, "null argument");
    int rc = check_flat_rows(design, row_begin, row_count);
    if (rc) return rc;
    if (row_count == 0) return VS_OK;
    VS_REQUIRE(host_rows, "host_rows is null");
    VS_HIP(hipSetDevice(ctx->device));
    vs_staging* S = nullptr;
    rc = staging_ready(ctx, &S);
    if (rc) return rc;
    const uint64_t k = (uint64_t)design->d.k;
    const uint64_t per = std::max<uint64_t>(1, VS_RING_SLOT_BYTES / (8 * k));
    const size_t npieces = (size_t)((row_count + per - 1) / per);
    const std::function<out_piece(size_t)> describe = [&](size_t p) {
        const uint64_t r0 = p * per, cnt = std::min(per, row_count - r0);
        return out_piece{nullptr, (size_t)(8 * cnt * k), (size_t)(8 * r0 * k)};
    };
    rc = drain_to_host(ctx, S, npieces, (char*)host_rows, describe, [&](size_t p, char* stage, hipStream_t ks) -> int {
        const uint64_t r0 = p * per, cnt = std::min(per, row_count - r0);
        return vs_sample_flat_launch(ctx, design, row_begin + r0, cnt, (double*)stage, ks);
    });
    hipStreamSynchronize(other_stream(ctx));
    hipStreamSynchronize(ctx->stream);
    return rc;
    VS_GUARD_END
}

extern "C" int vs_sample_rows_begin(vs_context* ctx, const vs_design* design, uint64_t row_begin, uint64_t row_count,
                                    int slot) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && design, "null argument");
    VS_REQUIRE(slot == 0 || slot == 1, "slot must be 0 or 1");
    int rc = check_flat_rows(design, row_begin, row_count);
    if (rc) return rc;
    VS_REQUIRE(row_count > 0, "row_count is 0");
    VS_HIP(hipSetDevice(ctx->device));
    vs_staging* S = nullptr;
    rc = staging_ready(ctx, &S);
    if (rc) return rc;
    const size_t bytes = sizeof(double) * row_count * (size_t)design->d.k;
    if (S->rows_pending[slot]) {   // the slot's last rows were never fetched: let their copy finish before the memory is reused
        VS_HIP(hipEventSynchronize(S->rows_done[slot]));
        S->rows_pending[slot] = false;
    }
    if (bytes > S->rows_bytes[slot]) {
        if (S->rows_pinned[slot]) VS_HIP(hipHostFree(S->rows_pinned[slot]));
        S->rows_pinned[slot] = nullptr;
        if (S->rows_dev[slot]) VS_HIP(hipFree(S->rows_dev[slot]));
        S->rows_dev[slot] = nullptr;
        S->rows_bytes[slot] = 0;
        VS_HIP(hipHostMalloc((void**)&S->rows_pinned[slot], bytes, hipHostMallocDefault));
        VS_HIP(hipMalloc((void**)&S->rows_dev[slot], bytes));
        S->rows_bytes[slot] = bytes;
    }
    const hipStream_t ks = ctx->stream, cs = other_stream(ctx);
    rc = vs_sample_flat_launch(ctx, design, row_begin, row_count, (double*)S->rows_dev[slot], ks);
    if (rc) return rc;
    VS_HIP(hipEventRecord(S->rows_kernel[slot], ks));
    VS_HIP(hipStreamWaitEvent(cs, S->rows_kernel[slot], 0));
    VS_HIP(hipMemcpyAsync(S->rows_pinned[slot], S->rows_dev[slot], bytes, hipMemcpyDeviceToHost, cs));
    VS_HIP(hipEventRecord(S->rows_done[slot], cs));
    // the kernel stream must not overwrite the device slot before the copy has read it: the next begin on this slot
    // waits for rows_done on the host (above), so no device-side wait is needed
    S->rows_pending[slot] = true;
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_sample_rows_wait(vs_context* ctx, int slot, const double** host_rows) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && host_rows, "null argument");
    VS_REQUIRE(slot == 0 || slot == 1, "slot must be 0 or 1");
    *host_rows = nullptr;
    vs_staging* S = ctx->staging;
    VS_REQUIRE(S && S->rows_pending[slot], "vs_sample_rows_wait: nothing was begun on slot %d", slot);
    VS_HIP(hipEventSynchronize(S->rows_done[slot]));
    S->rows_pending[slot] = false;
    *host_rows = (const double*)S->rows_pinned[slot];
    return VS_OK;
    VS_GUARD_END
}
