// Low-discrepancy generators, the implicit Saltelli design and the fused
// generate -> scale -> column-swap -> evaluate kernel.
//
// Reference counterparts (paths relative to the reference checkout):
//   ghalton.Halton(k).get(..)          varsens/saltelli.py:82-84
//   scaling(x[0:n]), scaling(x[n:2n])  varsens/saltelli.py:92,95 + varsens/scale.py:33,62
//   seed-1 row shuffle of M_2          varsens/saltelli.py:100-101 (perm computed on host)
//   generate_N_j                       varsens/saltelli.py:112-125 (never materialised here)
//   Sample.flat                        varsens/saltelli.py:127-160
//   Objective evaluation loops         varsens/saltelli.py:329-355
//
// Built with -ffp-contract=off: every multiply/add below rounds separately
// unless written as fma(), so the arithmetic matches numpy's operation for
// operation.
#include <cmath>

#include <exception>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

#include "vs_common.h"
#include "vs_host.h"

#include "vs_device_gen.h"

static vs_design_dev to_dev(const vs_design* g) {
    vs_design_dev D;
    D.generator = g->d.generator;
    D.k = g->d.k;
    D.scale_mode = g->d.scale_mode;
    D.sobol_bits = 32;
    D.halton_horner = g->halton_horner;
    D.reserved0 = 0;
    if (g->d.generator == VS_GEN_SOBOL) {
        const uint64_t top = g->d.first_index + 2 * g->d.n;  // Gray code of i < 2^b stays below 2^b
        int b = 1;
        while (b < 32 && (1ull << b) < top) ++b;
        D.sobol_bits = b;
    }
    D.n = g->d.n;
    D.first_index = g->d.first_index;
    D.lower = g->lower;
    D.width = g->width;
    D.perm = g->perm;
    D.perm_begin = g->d.perm_begin;
    D.raw = g->raw;
    D.dirs = g->dirs;
    D.sobol_tabs = g->sobol_tabs;
    D.bases = (const vs_base*)g->bases;
    return D;
}

// ---------------------------------------------------------------------------
// host-side tables
// ---------------------------------------------------------------------------

static std::vector<uint32_t> first_primes(int k) {
    std::vector<uint32_t> p;
    for (uint32_t c = 2; (int)p.size() < k; ++c) {
        bool prime = true;
        for (uint32_t q : p) {
            if ((uint64_t)q * q > c) break;
            if (c % q == 0) { prime = false; break; }
        }
        if (prime) p.push_back(c);
    }
    return p;
}

static vs_base make_base(uint32_t b, uint64_t max_index) {
    vs_base B;
    B.b = b;
    B.ndig = 0;
    B.top = 1.0;
    B.inv_den = 1.0;
    B.inv_b = 0.5;
    B.inv_lo = 0.0;
    if (max_index < (1ull << 32)) {
        uint64_t den = 1;
        for (uint64_t v = max_index; v; v /= b) {
            ++B.ndig;
            den *= b;  // <= max_index * b < 2^53
        }
        if (B.ndig == 0) { B.ndig = 1; den = b; }
        while (B.ndig < 3 && den * b < (1ull << 53)) { ++B.ndig; den *= b; }   // leading zero digits: same quotient (vs_radical_inverse_pair_horner peels two steps)
        const uint64_t top = den / b;   // b^(ndig-1) <= max_index
        B.top = (double)top;
        B.inv_den = 1.0 / (double)den;
        // second word of the reciprocal: 1 - den * inv_den exactly (long double holds the 53 x 53 bit product of two
        // doubles only to 64 bits: form it with an fma instead), over den
        B.inv_lo = fma(-(double)den, B.inv_den, 1.0) / (double)den;
        // the three-instruction digit step (vs_radical_inverse_pair_horner) forms index * top: exact below 2^53
        // ... and takes its quotient from the two-word reciprocal: den <= 2^49 (vs_halton_quotients_dd)
        if (B.ndig >= 3 && (unsigned __int128)max_index * top < ((unsigned __int128)1 << 53) && den <= (1ull << 49))
            B.ndig |= VS_BASE_HORNER;
    }
    B.inv_b = nextafter(nextafter(1.0 / (double)b, 2.0), 2.0);
    return B;
}

// max_index: largest sequence index the kernels will ask for
static int upload_bases(vs_context* ctx, int k, uint64_t max_index, uint32_t** out_dev, uint32_t* largest,
                        int* horner = nullptr) {
    std::vector<uint32_t> p = first_primes(k);
    std::vector<vs_base> tab(k);
    int all = 1;
    for (int c = 0; c < k; ++c) {
        tab[c] = make_base(p[c], max_index);
        if (!(tab[c].ndig & VS_BASE_HORNER)) all = 0;
    }
    if (horner) *horner = all;
    *largest = p[k - 1];
    VS_HIP(hipMalloc(out_dev, sizeof(vs_base) * k));
    VS_HIP(hipMemcpyAsync(*out_dev, tab.data(), sizeof(vs_base) * k, hipMemcpyHostToDevice, ctx->stream));
    VS_HIP(hipStreamSynchronize(ctx->stream));
    return VS_OK;
}

// ---------------------------------------------------------------------------
// vs_halton_fill / vs_sobol_fill
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
halton_fill_kernel(const vs_base* __restrict__ bases, int k, uint64_t start, uint64_t total,
                   double* __restrict__ out) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; e < total; e += stride) {
        uint64_t row = e / (uint32_t)k;
        int c = (int)(e - row * (uint32_t)k);
        out[e] = vs_radical_inverse(start + row, bases[c]);
    }
}

__global__ void __launch_bounds__(256)
sobol_fill_kernel(const uint32_t* __restrict__ dirs, int k, uint64_t start, uint64_t total,
                  double* __restrict__ out) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; e < total; e += stride) {
        uint64_t row = e / (uint32_t)k;
        int c = (int)(e - row * (uint32_t)k);
        out[e] = vs_sobol_point(start + row, dirs + 32 * c, 32);
    }
}

static unsigned grid_for(uint64_t total, unsigned block, unsigned cap) {
    uint64_t g = (total + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

extern "C" int vs_halton_fill(vs_context* ctx, int k, uint64_t start, uint64_t count, double* out_dev) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx, "ctx is null");
    VS_HIP(hipSetDevice(ctx->device));
    VS_REQUIRE(k >= 1 && k <= 65536, "k=%d out of range", k);
    if (count == 0) return VS_OK;
    VS_REQUIRE(out_dev, "out_dev is null");
    uint32_t* bases = nullptr;
    uint32_t largest = 0;
    int rc = upload_bases(ctx, k, start + count - 1, &bases, &largest);
    if (rc) return rc;
    if ((long double)(start + count) * largest >= 9007199254740992.0L) {
        hipFree(bases);
        return vs_fail(VS_ERR_ARG, "Halton index range too large for exact fp64 (index*base >= 2^53)");
    }
    uint64_t total = count * (uint64_t)k;
    {
        vs_launch_scope scope(ctx, VS_KERNEL_HALTON_FILL);
        hipLaunchKernelGGL(halton_fill_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, ctx->stream,
                           (const vs_base*)bases, k, start, total, out_dev);
    }
    VS_HIP(hipGetLastError());
    VS_HIP(hipStreamSynchronize(ctx->stream));
    VS_HIP(hipFree(bases));
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_sobol_fill(vs_context* ctx, int k, const uint32_t* directions, uint64_t start,
                             uint64_t count, double* out_dev) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && directions, "null argument");
    VS_HIP(hipSetDevice(ctx->device));
    VS_REQUIRE(k >= 1, "k=%d out of range", k);
    VS_REQUIRE(start + count <= (1ull << 32), "Sobol indices must stay below 2^32");
    if (count == 0) return VS_OK;
    VS_REQUIRE(out_dev, "out_dev is null");
    uint32_t* dirs = nullptr;
    VS_HIP(hipMalloc(&dirs, sizeof(uint32_t) * 32 * k));
    VS_HIP(hipMemcpyAsync(dirs, directions, sizeof(uint32_t) * 32 * k, hipMemcpyHostToDevice, ctx->stream));
    uint64_t total = count * (uint64_t)k;
    {
        vs_launch_scope scope(ctx, VS_KERNEL_SOBOL_FILL);
        hipLaunchKernelGGL(sobol_fill_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, ctx->stream,
                           (const uint32_t*)dirs, k, start, total, out_dev);
    }
    VS_HIP(hipGetLastError());
    VS_HIP(hipStreamSynchronize(ctx->stream));
    VS_HIP(hipFree(dirs));
    return VS_OK;
    VS_GUARD_END
}

// ---------------------------------------------------------------------------
// design objects
// ---------------------------------------------------------------------------

// A design belongs to the context that made it (vs_design::owner): its tables live on that GPU and, with
// VS_DESIGN_SEED1_SHUFFLE, it holds a share of that context's shuffle cache.  It is released through its owner whatever
// `ctx` the caller passes (NULL included); vs_context_destroy refuses while designs of the context are alive.
extern "C" int vs_design_destroy(vs_context* ctx, vs_design* g) {
    VS_GUARD_BEGIN
    if (!g) return VS_OK;
    vs_context* owner = g->owner ? g->owner : ctx;
    if (owner) {  // kernels on either stream may still read the tables
        hipSetDevice(owner->device);
        hipStreamSynchronize(owner->streams[0]);
        hipStreamSynchronize(owner->streams[1]);
    }
    if (g->lower) hipFree(g->lower);
    if (g->width) hipFree(g->width);
    if (g->perm && g->perm_shared) {
        if (owner)
            for (auto& e : owner->perm_cache)
                if (e.dev == g->perm && e.refs > 0) { --e.refs; break; }
    } else if (g->perm) {
        hipFree(g->perm);
    }
    if (g->raw) hipFree(g->raw);
    if (g->dirs) hipFree(g->dirs);
    if (g->sobol_tabs) hipFree(g->sobol_tabs);
    if (g->bases) hipFree(g->bases);
    if (g->owner && g->owner->live_designs > 0) --g->owner->live_designs;
    delete g;
    return VS_OK;
    VS_GUARD_END
}

// VS_DESIGN_SEED1_SHUFFLE: the device copy of rows [begin, begin + count) of the seed-1 shuffle of n rows, from the
// context's cache or made now (host: vs_seed1_host_cached, once per n and process).  A rank without rows
// (count = 0) gets a one-entry buffer and no host permutation is computed for it.
static int seed1_shuffle_on_device(vs_context* ctx, uint64_t n, uint64_t begin, uint64_t count, uint32_t** out) {
    for (auto& e : ctx->perm_cache)
        if (e.n == n && e.begin == begin && e.count == count) {
            ++e.refs;
            e.stamp = ++ctx->perm_stamp;
            *out = e.dev;
            return VS_OK;
        }
    std::shared_ptr<const std::vector<uint32_t>> perm;
    if (count) {
        const int rc = vs_seed1_host_cached(n, &perm);
        if (rc) return rc;
    }
    // room: drop unreferenced copies, least recently used first, beyond four entries or 1 GiB of them
    for (;;) {
        uint64_t idle_bytes = 0;
        for (auto& e : ctx->perm_cache)
            if (e.refs == 0) idle_bytes += 4 * e.count;
        if (ctx->perm_cache.size() < 4 && idle_bytes + 4 * count <= (1ull << 30)) break;
        int victim = -1;
        for (size_t i = 0; i < ctx->perm_cache.size(); ++i)
            if (ctx->perm_cache[i].refs == 0 && (victim < 0 || ctx->perm_cache[i].stamp < ctx->perm_cache[victim].stamp))
                victim = (int)i;
        if (victim < 0) break;   // all in use: the cache grows with the designs alive
        hipStreamSynchronize(ctx->streams[0]);
        hipStreamSynchronize(ctx->streams[1]);
        hipFree(ctx->perm_cache[victim].dev);
        ctx->perm_cache.erase(ctx->perm_cache.begin() + victim);
    }
    ctx->perm_cache.reserve(ctx->perm_cache.size() + 1);   // before the device allocation: nothing to undo if it throws
    uint32_t* dev = nullptr;
    VS_HIP(hipMalloc(&dev, sizeof(uint32_t) * (count ? count : 1)));
    if (count) {
        const hipError_t e = hipMemcpy(dev, perm->data() + begin, sizeof(uint32_t) * count, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            hipFree(dev);
            return vs_fail(VS_ERR_HIP, "upload of the seed-1 shuffle failed: %s", hipGetErrorString(e));
        }
    }
    ctx->perm_cache.push_back({n, begin, count, dev, 1, ++ctx->perm_stamp});
    *out = dev;
    return VS_OK;
}

extern "C" int vs_design_create(vs_context* ctx, const vs_design_desc* desc, vs_design** out) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && desc && out, "null argument");
    VS_HIP(hipSetDevice(ctx->device));
    *out = nullptr;
    // every check of the descriptor: vs_host.cpp (host only; the sanitizer leg of the CPU tests runs it)
    uint64_t perm_count = 0;
    {
        const int bad = vs_design_desc_validate(desc, &perm_count);
        if (bad) return bad;
    }
    const int k = desc->k;
    const uint64_t n = desc->n;

    VS_HIP(hipSetDevice(ctx->device));
    vs_design* g = new vs_design();
    g->owner = ctx;
    ++ctx->live_designs;
    // until the last step is through, the half-built design goes with this scope -- on an error return and on an
    // exception on its way to the guard alike
    struct holder {
        vs_context* ctx;
        vs_design* g;
        ~holder() { if (g) vs_design_destroy(ctx, g); }
    } hold{ctx, g};
    g->d = *desc;
    int rc = VS_OK;
    auto bail = [&](int code) { return code; };
#define VS_TRY(call)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return bail(vs_fail(VS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)));     \
    } while (0)

    if (desc->scale_mode != VS_SCALE_IDENTITY) {
        // width: LINEAR upper - lower; POWER the ratio r = RN(upper / lower) as the reference forms it, and behind the k
        // ratios ln r as a double-double (hi[k], lo[k]; logl of the rounded ratio): the kernels take lower * r^x as
        // lower * exp(x ln r) with the product's rounding error carried along (vs_power_scale, vs_device_gen.h) -- one
        // exp instead of a pow per point, 1-2 ulp from numpy's pow.  A ratio whose logarithm is not finite (zero,
        // negative, infinite bounds) keeps pow() and its conventions.
        const bool power = desc->scale_mode == VS_SCALE_POWER;
        std::vector<double> lo(desc->lower, desc->lower + k), w(power ? 3 * (size_t)k : (size_t)k);
        for (int c = 0; c < k; ++c) {
            w[c] = desc->scale_mode == VS_SCALE_LINEAR ? desc->upper[c] - desc->lower[c]   // scale.py:33
                                                       : desc->upper[c] / desc->lower[c];  // scale.py:62
            if (power) {
                const long double L = logl((long double)w[c]);
                const double hi = (double)L;
                w[k + c] = hi;
                w[2 * (size_t)k + c] = std::isfinite(hi) ? (double)(L - (long double)hi) : 0.0;
            }
        }
        VS_TRY(hipMalloc(&g->lower, sizeof(double) * k));
        VS_TRY(hipMalloc(&g->width, sizeof(double) * w.size()));
        VS_TRY(hipMemcpyAsync(g->lower, lo.data(), sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream));
        VS_TRY(hipMemcpyAsync(g->width, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice, ctx->stream));
        VS_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (desc->flags & VS_DESIGN_SEED1_SHUFFLE) {
        const uint64_t count = desc->perm_count == VS_PERM_EMPTY ? 0 : desc->perm_count ? desc->perm_count : n;
        rc = seed1_shuffle_on_device(ctx, n, desc->perm_begin, count, &g->perm);
        if (rc) return bail(rc);
        g->perm_shared = true;
        g->d.perm_count = count;
    } else if (desc->perm) {
        VS_TRY(hipMalloc(&g->perm, sizeof(uint32_t) * (perm_count ? perm_count : 1)));
        VS_TRY(hipMemcpyAsync(g->perm, desc->perm, sizeof(uint32_t) * perm_count, hipMemcpyHostToDevice, ctx->stream));
        g->d.perm_count = perm_count;
    } else {
        g->d.perm_begin = 0;
        g->d.perm_count = 0;
    }
    if (desc->generator == VS_GEN_RAW) {
        size_t bytes = sizeof(double) * 2 * n * (size_t)k;
        VS_TRY(hipMalloc(&g->raw, bytes));
        VS_TRY(hipMemcpyAsync(g->raw, desc->raw, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    if (desc->generator == VS_GEN_SOBOL) {
        VS_TRY(hipMalloc(&g->dirs, sizeof(uint32_t) * 32 * k));
        VS_TRY(hipMemcpyAsync(g->dirs, desc->sobol_directions, sizeof(uint32_t) * 32 * k,
                              hipMemcpyHostToDevice, ctx->stream));
        // byte tables of the direction integers (vs_sobol_point_tab)
        std::vector<uint32_t> tabs;
        rc = vs_sobol_byte_tables(k, desc->sobol_directions, &tabs);
        if (rc) return bail(rc);
        // ... and nibble tables behind them, [k][8][16]: entry v of nibble position b = the XOR of the direction integers
        // 4b .. 4b + 3 selected by v -- what the scan kernels keep in LDS (vs_sobol_stage): byte b's entries v and 16 v
        tabs.resize((size_t)k * 1024 + (size_t)k * 128);
        for (int c = 0; c < k; ++c)
            for (int b = 0; b < 4; ++b)
                for (int v = 0; v < 16; ++v) {
                    tabs[(size_t)k * 1024 + ((size_t)c * 8 + 2 * b) * 16 + v] = tabs[((size_t)c * 4 + b) * 256 + v];
                    tabs[(size_t)k * 1024 + ((size_t)c * 8 + 2 * b + 1) * 16 + v] = tabs[((size_t)c * 4 + b) * 256 + 16 * v];
                }
        VS_TRY(hipMalloc(&g->sobol_tabs, sizeof(uint32_t) * tabs.size()));
        VS_TRY(hipMemcpy(g->sobol_tabs, tabs.data(), sizeof(uint32_t) * tabs.size(), hipMemcpyHostToDevice));
    }
    if (desc->generator == VS_GEN_HALTON) {
        uint32_t largest = 0;
        rc = upload_bases(ctx, k, desc->first_index + 2 * n - 1, &g->bases, &largest, &g->halton_horner);
        if (rc) return bail(rc);
        if ((long double)(desc->first_index + 2 * n) * largest >= 9007199254740992.0L)
            return bail(vs_fail(VS_ERR_ARG, "Halton index range too large for exact fp64 (index*base >= 2^53)"));
    }
    VS_TRY(hipStreamSynchronize(ctx->stream));
#undef VS_TRY
    g->d.lower = g->d.upper = nullptr;
    g->d.perm = nullptr;
    g->d.raw = nullptr;
    g->d.sobol_directions = nullptr;
    hold.g = nullptr;
    *out = g;
    return VS_OK;
    VS_GUARD_END
}

// ---------------------------------------------------------------------------
// vs_sample_flat: rows of Sample.flat()
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
sample_flat_kernel(vs_design_dev D, uint64_t row_begin, uint64_t total, double* __restrict__ out) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t k = (uint32_t)D.k;
    for (; e < total; e += stride) {
        uint64_t lr = e / k;
        int c = (int)(e - lr * k);
        uint64_t fr = row_begin + lr;
        uint64_t block = fr / D.n;
        uint64_t r = fr - block * D.n;
        // block 0: M_1, 1: M_2, 2+i: N_j[i] (M_2, column i from M_1), 2+k+i: N_nj[i]
        bool from_m1;
        if (block == 0) from_m1 = true;
        else if (block == 1) from_m1 = false;
        else if (block < 2 + k) from_m1 = ((uint64_t)c == block - 2);
        else from_m1 = ((uint64_t)c != block - 2 - k);
        out[e] = from_m1 ? vs_m1(D, r, c) : vs_m2(D, r, c);
    }
}

// the launch alone, on a given stream: vs_sample_flat below and the host-bound row fetches of vs_values.hip
int vs_sample_flat_launch(vs_context* ctx, const vs_design* design, uint64_t row_begin, uint64_t row_count,
                          double* out_dev, hipStream_t stream) {
    const uint64_t total = row_count * (uint64_t)design->d.k;
    {
        vs_launch_scope scope(ctx, VS_KERNEL_SAMPLE_FLAT);
        hipLaunchKernelGGL(sample_flat_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, stream,
                           to_dev(design), row_begin, total, out_dev);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
}

extern "C" int vs_sample_flat(vs_context* ctx, const vs_design* design, uint64_t row_begin,
                              uint64_t row_count, double* out_dev) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && design, "null argument");
    VS_HIP(hipSetDevice(ctx->device));
    const uint64_t rows_total = 2 * design->d.n * ((uint64_t)design->d.k + 1);
    VS_REQUIRE(row_begin <= rows_total && row_count <= rows_total - row_begin,
               "flat rows [%llu,+%llu) exceed 2n(k+1)=%llu", (unsigned long long)row_begin,
               (unsigned long long)row_count, (unsigned long long)rows_total);
    if (row_count == 0) return VS_OK;
    VS_REQUIRE(out_dev, "out_dev is null");
    VS_REQUIRE(!design->perm || design->d.perm_count == design->d.n,
               "vs_sample_flat needs the whole shuffle (this design holds a slice of %llu rows)",
               (unsigned long long)design->d.perm_count);
    return vs_sample_flat_launch(ctx, design, row_begin, row_count, out_dev, ctx->stream);
    VS_GUARD_END
}

// ---------------------------------------------------------------------------
// objectives
// ---------------------------------------------------------------------------

extern "C" int vs_objective_destroy(vs_context* ctx, vs_objective* obj) {
    VS_GUARD_BEGIN
    if (!obj) return VS_OK;
    if (ctx) {
        hipStreamSynchronize(ctx->streams[0]);
        hipStreamSynchronize(ctx->streams[1]);
    }
    if (obj->params) hipFree(obj->params);
    vs_rtc_release(obj);
    delete obj;
    return VS_OK;
    VS_GUARD_END
}

namespace {
// a half-built objective goes with the scope that holds it (error returns and exceptions on their way to the guard)
struct objective_holder {
    vs_context* ctx;
    vs_objective* o;
    ~objective_holder() { if (o) vs_objective_destroy(ctx, o); }
    vs_objective* release() {
        vs_objective* p = o;
        o = nullptr;
        return p;
    }
};
}  // namespace

extern "C" int vs_objective_create(vs_context* ctx, int kind, int k, int l, const double* params,
                                   size_t nparams, vs_objective** out) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && out, "null argument");
    VS_HIP(hipSetDevice(ctx->device));
    *out = nullptr;
    VS_REQUIRE(k >= 1 && l >= 1, "k and l must be positive");
    switch (kind) {
        case VS_OBJ_G_FUNCTION:
        case VS_OBJ_LINEAR:
        case VS_OBJ_POLYNOMIAL:
            VS_REQUIRE(nparams == (size_t)k * l, "objective %d needs l*k=%d parameters, got %zu", kind, k * l,
                       nparams);
            break;
        case VS_OBJ_ISHIGAMI:
            VS_REQUIRE(k == 3 && l == 1 && nparams == 2, "Ishigami needs k=3, l=1 and parameters {a,b}");
            break;
        case VS_OBJ_INVERT:
            VS_REQUIRE(l == k && nparams == 0, "invert has l=k outputs and no parameters");
            break;
        default:
            return vs_fail(VS_ERR_ARG, "unknown objective kind %d", kind);
    }
    VS_REQUIRE(nparams == 0 || params, "params is null");
    vs_objective* o = new vs_objective();
    objective_holder hold{ctx, o};
    o->kind = kind;
    o->k = k;
    o->l = l;
    o->nparams = nparams;
    if (nparams) {
        // G_FUNCTION: t = (|4x - 2| + a) / (1 + a) (varsens/tests/test_g_function.py:9-14).  The scan kernels take the
        // quotient from the correctly rounded reciprocal r = RN(1 / (1 + a)), computed here in IEEE double, by
        // Markstein's fma sequence -- two steps of q += fma(-q, d, n) * r give RN(n / d) exactly when r is the correctly
        // rounded reciprocal of a normal d whose significand is not all ones and nothing leaves the normal range
        // (checked below).  The reciprocals follow the l*k parameters in the same device buffer; a column whose
        // parameter does not meet the conditions gets the reciprocal 0, on which the scan kernels' GENERAL column code
        // divides for that column alone; the specialised column code (Halton, identity / linear scaling) carries no
        // division -- the branch costs its g-function kernels 1-4 % -- and serves objectives whose columns are all plain
        // (round 4; until round 3 one such parameter -- a = 7 - 1 ulp is enough: 1 + a is the largest double below 8 --
        // sent the whole objective to the O(k^2) chain kernels and past every fused path).
        std::vector<double> host(params, params + nparams);
        if (kind == VS_OBJ_G_FUNCTION) {
            bool ok = true;
            for (size_t i = 0; i < nparams; ++i) {
                const double d = 1.0 + params[i];
                uint64_t bits;
                memcpy(&bits, &d, sizeof(bits));
                const bool plain = std::isfinite(params[i]) && std::isfinite(d) && fabs(d) > 0x1p-200 && fabs(d) < 0x1p200 &&
                                   fabs(params[i]) < 0x1p200 && (bits & 0x000fffffffffffffull) != 0x000fffffffffffffull;
                ok = ok && plain;
                host.push_back(plain ? 1.0 / d : 0.0);   // 0: this column divides (builtin_term::eval, general column code)
            }
            o->recip_ok = ok;   // all columns plain: the specialised column code (no division compiled in) may be used
        }
        if (hipMalloc(&o->params, sizeof(double) * host.size()) != hipSuccess ||
            hipMemcpyAsync(o->params, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice, ctx->stream) !=
                hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            return vs_fail(VS_ERR_HIP, "uploading objective parameters failed");
    }
    *out = hold.release();
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_objective_create_source(vs_context* ctx, int k, int l, const char* body, const double* params,
                                          size_t nparams, vs_objective** out) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && out && body, "null argument");
    VS_HIP(hipSetDevice(ctx->device));
    *out = nullptr;
    VS_REQUIRE(k >= 1 && l >= 1 && l <= 1024, "k and l must be positive (l <= 1024)");
    VS_REQUIRE(nparams == 0 || params, "params is null");
    vs_objective* o = new vs_objective();
    objective_holder hold{ctx, o};
    o->kind = VS_OBJ_SOURCE;
    o->k = k;
    o->l = l;
    o->nparams = nparams;
    if (nparams) {
        if (hipMalloc(&o->params, sizeof(double) * nparams) != hipSuccess ||
            hipMemcpyAsync(o->params, params, sizeof(double) * nparams, hipMemcpyHostToDevice, ctx->stream) !=
                hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            return vs_fail(VS_ERR_HIP, "uploading objective parameters failed");
    }
    int rc = vs_rtc_build(ctx, o, body, 0);
    if (rc) return rc;
    *out = hold.release();
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_objective_create_separable(vs_context* ctx, int k, const char* term_body, int fold,
                                             const double* params, size_t nparams, vs_objective** out) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && out && term_body, "null argument");
    VS_HIP(hipSetDevice(ctx->device));
    *out = nullptr;
    VS_REQUIRE(k >= 1 && k <= 256, "a separable source objective needs 1 <= k <= 256");
    VS_REQUIRE(fold == VS_FOLD_PRODUCT || fold == VS_FOLD_SUM, "fold must be VS_FOLD_PRODUCT or VS_FOLD_SUM");
    VS_REQUIRE(nparams == 0 || params, "params is null");
    vs_objective* o = new vs_objective();
    objective_holder hold{ctx, o};
    o->kind = VS_OBJ_SOURCE;
    o->k = k;
    o->l = 1;
    o->nparams = nparams;
    if (nparams) {
        if (hipMalloc(&o->params, sizeof(double) * nparams) != hipSuccess ||
            hipMemcpyAsync(o->params, params, sizeof(double) * nparams, hipMemcpyHostToDevice, ctx->stream) !=
                hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            return vs_fail(VS_ERR_HIP, "uploading objective parameters failed");
    }
    int rc = vs_rtc_build(ctx, o, term_body, fold == VS_FOLD_SUM ? 2 : 1);
    if (rc) return rc;
    *out = hold.release();
    return VS_OK;
    VS_GUARD_END
}

// ---------------------------------------------------------------------------
// the fused evaluation kernel
// ---------------------------------------------------------------------------
//
// A workgroup of four waves owns 64 design rows (row = lane); the four waves
// split the work of those rows by column / by output block:
//   phase 1  wave q generates (or gathers) and scales coordinates c = q, q+4, ..
//            of M_1[r] and M_2[r] into an LDS tile [coordinate][row]
//            (conflict-free: consecutive lanes, consecutive doubles);
//   phase 2  wave q produces the objective values of blocks i = q, q+4, ..:
//            f(N_j[i][r]) and f(N_nj[i][r]); wave 0 adds f(M_1[r]), f(M_2[r]).
// N_j / N_nj are never built: point i is a column select between the two LDS
// columns of the row.  Stores are 512 B contiguous per wave instruction.
//
// Separable objectives (g-function, polynomial: product of per-coordinate
// terms; linear: sum of per-coordinate terms) are folded strictly left to
// right, so the first i terms of N_j[i] are exactly the running prefix over
// M_2's terms.  A thread carries up to 8 such chains (16 with N_nj) in
// registers and feeds all of them from one pair of LDS reads per column: k^2/2
// multiplies per family, bit-identical to folding each point on its own.
//
// Output index: f[((o*(2k+2) + block)*rows + (r-row_begin)] (planar, see varsens_hip.h).


#ifndef VS_EVAL_MIN_WAVES
#define VS_EVAL_MIN_WAVES 1
#endif
// W = waves per workgroup (4 or 8) sharing the 64-row tile: the larger the LDS tile (k), the
// fewer workgroups fit a CU, so more waves per workgroup keep the SIMDs busy.
template <int KIND, bool SCRATCH, int W>
__global__ void __launch_bounds__(VS_EVAL_ROWS * W, VS_EVAL_MIN_WAVES)
eval_kernel(vs_eval_args A, double* scratch, int tiles) {
    constexpr int CH = VS_PASS / W;  // chains per thread and pass
    extern __shared__ double lds_tile[];
    constexpr int T = VS_EVAL_ROWS;
#define TL(p) vs_tile_ld<SCRATCH>(p)
    constexpr bool SEPARABLE = KIND == VS_OBJ_G_FUNCTION || KIND == VS_OBJ_POLYNOMIAL || KIND == VS_OBJ_LINEAR;
    const int k = A.D.k;
    const int l = A.l;
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index, provably uniform
    const uint64_t rows = A.rows;
    double* lds = SCRATCH ? scratch + (size_t)blockIdx.x * tiles * k * T : lds_tile;
    const uint64_t nblocks = (rows + T - 1) / T;
    // one 64-row block per workgroup (LDS tile) or a persistent walk over blocks (scratch tile)
    for (uint64_t rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
    const uint64_t lr = rb * T + lane;
    const bool valid = lr < rows;
    const uint64_t r = A.row_begin + (valid ? lr : rows - 1);  // clamp: loads stay in range, stores are guarded
    double* X1 = lds + lane;                  // X1[c*T]
    double* X2 = lds + (size_t)k * T + lane;  // X2[c*T]
    // row-blocked f-values (vs_f_tile): the 64 rows lie in one row block; plane (o, block) of these rows is
    // fpitch doubles behind the plane before it; bases are wave-uniform
    const vs_ftile ft = vs_f_tile(rows, (2 * (uint64_t)k + 2) * (uint64_t)l, A.f_shift, rb * T);
    const uint64_t fpitch = ft.pitch;
    double* f = A.f + ft.base + lane;
    const uint64_t plane = (2 * (uint64_t)k + 2) * fpitch;
#define VS_OUT(block, o, v)                                               \
    do {                                                                  \
        if (valid) VS_STORE(f + ((uint64_t)(o) * plane + (uint64_t)(block) * fpitch), (v)); \
    } while (0)

    // ---- phase 1: coordinates (for a single-output separable objective: terms)
    const bool terms_now = SEPARABLE && l == 1;
    const bool halton_fast = A.D.generator == VS_GEN_HALTON && A.D.bases[0].ndig != 0;
    const uint64_t src2 = A.D.n + vs_perm_at(A.D, r);  // sequence row behind M_2[r]
    for (int c = q; c < k; c += W) {
        double x1, x2;
        if (halton_fast) {
            vs_radical_inverse_pair((uint32_t)(A.D.first_index + r), (uint32_t)(A.D.first_index + src2),
                                    A.D.bases[c], x1, x2);
        } else {
            x1 = vs_unit_point(A.D, r, c);
            x2 = vs_unit_point(A.D, src2, c);
        }
        x1 = vs_apply_scale(A.D, x1, c);
        x2 = vs_apply_scale(A.D, x2, c);
        if (terms_now) {
            const double p = A.params[c];
            x1 = vs_term<KIND>(x1, p);
            x2 = vs_term<KIND>(x2, p);
        }
        X1[(size_t)c * T] = x1;
        X2[(size_t)c * T] = x2;
    }
    __syncthreads();

    if (SEPARABLE) {
        double* F1 = X1;
        double* F2 = X2;
        if (l > 1) {  // keep the coordinates; terms go to a second pair of tiles
            F1 = lds + 2 * (size_t)k * T + lane;
            F2 = lds + 3 * (size_t)k * T + lane;
        }
        const double ident = KIND == VS_OBJ_LINEAR ? 0.0 : 1.0;
        for (int o = 0; o < l; ++o) {
            if (l > 1) {
                __syncthreads();  // previous output's chains are done with F1/F2
                const double* p = A.params + (size_t)o * k;
                for (int c = q; c < k; c += W) {
                    F1[(size_t)c * T] = vs_term<KIND>(TL(X1 + (size_t)c * T), p[c]);
                    F2[(size_t)c * T] = vs_term<KIND>(TL(X2 + (size_t)c * T), p[c]);
                }
                __syncthreads();
            }
            double pa = ident, pb = ident;  // folds of M_1 / M_2 terms over columns < c
            for (int i0 = 0; i0 < k; i0 += VS_PASS) {
                // this thread's chains: blocks i0 + q + 4m, m = 0..7
                double cj[CH], dj[CH];
#pragma unroll
                for (int m = 0; m < CH; ++m) cj[m] = dj[m] = ident;
                const int cend = min(k, i0 + VS_PASS);
                int c = i0;
                if (cend - i0 == VS_PASS) {
                    vs_chain_dispatch<KIND, SCRATCH, W>::run(q, F1, F2, i0, pa, pb, cj, dj);  // wave-uniform q
                    c = cend;
                } else {
                    // partial pass (k not a multiple of 32): same schedule with run-time bounds
                    for (const int stop = min(cend, i0 + q); c < stop; ++c) {
                        pa = vs_fold<KIND>(pa, TL(F1 + (size_t)c * T));
                        pb = vs_fold<KIND>(pb, TL(F2 + (size_t)c * T));
                    }
#pragma unroll
                    for (int m = 0; m < CH; ++m) {
                        if (c < cend) {
                            const double ta = TL(F1 + (size_t)c * T), tb = TL(F2 + (size_t)c * T);
#pragma unroll
                            for (int mm = 0; mm < m; ++mm) {
                                cj[mm] = vs_fold<KIND>(cj[mm], tb);
                                dj[mm] = vs_fold<KIND>(dj[mm], ta);
                            }
                            cj[m] = vs_fold<KIND>(pb, ta);
                            dj[m] = vs_fold<KIND>(pa, tb);
                            pa = vs_fold<KIND>(pa, ta);
                            pb = vs_fold<KIND>(pb, tb);
                            ++c;
#pragma unroll
                            for (int s = 1; s < W; ++s) {
                                if (c < cend) {
                                    const double ua = TL(F1 + (size_t)c * T), ub = TL(F2 + (size_t)c * T);
#pragma unroll
                                    for (int mm = 0; mm <= m; ++mm) {
                                        cj[mm] = vs_fold<KIND>(cj[mm], ub);
                                        dj[mm] = vs_fold<KIND>(dj[mm], ua);
                                    }
                                    pa = vs_fold<KIND>(pa, ua);
                                    pb = vs_fold<KIND>(pb, ub);
                                    ++c;
                                }
                            }
                        }
                    }
                }
                // columns of later passes: every chain of this pass runs on to the last column
                for (; c < k; ++c) {
                    const double ta = TL(F1 + (size_t)c * T), tb = TL(F2 + (size_t)c * T);
#pragma unroll
                    for (int m = 0; m < CH; ++m) {
                        cj[m] = vs_fold<KIND>(cj[m], tb);
                        dj[m] = vs_fold<KIND>(dj[m], ta);
                    }
                }
#pragma unroll
                for (int m = 0; m < CH; ++m) {
                    const int i = i0 + q + W * m;
                    if (i < k) {
                        VS_OUT(2 + i, o, cj[m]);
                        VS_OUT(2 + k + i, o, dj[m]);
                    }
                }
            }
            if (q == 0) {
                VS_OUT(0, o, pa);
                VS_OUT(1, o, pb);
            }
        }
    } else if (KIND == VS_OBJ_ISHIGAMI) {
        if (q == 0) {
            const double a = A.params[0], b = A.params[1];
            double s[2][2], p4[2];  // [which sample][coordinate]
            s[0][0] = sin(TL(X1));
            s[1][0] = sin(TL(X2));
            s[0][1] = sin(TL(X1 + (size_t)T));
            s[1][1] = sin(TL(X2 + (size_t)T));
            const double x3a = TL(X1 + 2 * (size_t)T), x3b = TL(X2 + 2 * (size_t)T);
            p4[0] = (x3a * x3a) * (x3a * x3a);
            p4[1] = (x3b * x3b) * (x3b * x3b);
            // value with coordinate c taken from sample u_c in {0 = M_1, 1 = M_2}
            auto val = [&](int u0, int u1, int u2) {
                const double s1 = s[u0][0], s2 = s[u1][1];
                return (s1 + a * (s2 * s2)) + (b * p4[u2]) * s1;
            };
            VS_OUT(0, 0, val(0, 0, 0));
            VS_OUT(1, 0, val(1, 1, 1));
            VS_OUT(2, 0, val(0, 1, 1));  // N_j[i]: M_2 except column i
            VS_OUT(3, 0, val(1, 0, 1));
            VS_OUT(4, 0, val(1, 1, 0));
            VS_OUT(5, 0, val(1, 0, 0));  // N_nj[i]: M_1 except column i
            VS_OUT(6, 0, val(0, 1, 0));
            VS_OUT(7, 0, val(0, 0, 1));
        }
    } else if (KIND == VS_OBJ_INVERT) {
        for (int o = q; o < k; o += W) {
            const double ia = 1.0 - TL(X1 + (size_t)o * T), ib = 1.0 - TL(X2 + (size_t)o * T);
            VS_OUT(0, o, ia);
            VS_OUT(1, o, ib);
            for (int i = 0; i < k; ++i) {
                VS_OUT(2 + i, o, (i == o) ? ia : ib);
                VS_OUT(2 + k + i, o, (i == o) ? ib : ia);
            }
        }
    }
    if (!SCRATCH) break;  // LDS tile: exactly one block per workgroup (no loop in the generated code)
    __syncthreads();      // scratch tile: it is rewritten by the next block
    }
#undef VS_OUT
#undef TL
}

template <int KIND, int W>
static int launch_eval_w(vs_context* ctx, const vs_eval_args& A, int tiles, size_t lds, uint64_t blocks, bool scratch) {
    if (!scratch) {
        if (lds > 64 * 1024)
            VS_HIP(hipFuncSetAttribute((const void*)eval_kernel<KIND, false, W>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        vs_launch_scope scope(ctx, VS_KERNEL_EVAL);
        hipLaunchKernelGGL((eval_kernel<KIND, false, W>), dim3((unsigned)blocks), dim3(VS_EVAL_ROWS * W), lds,
                           ctx->stream, A, (double*)nullptr, tiles);
    } else {
        // the tile of one workgroup does not fit LDS: persistent workgroups, tile in an HBM scratch slice
        const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        uint64_t slots = (uint64_t)cus * (16 / W);
        if (slots > blocks) slots = blocks;
        int rc = vs_workspace_scratch(ctx, (size_t)slots * lds);
        if (rc) return rc;
        vs_launch_scope scope(ctx, VS_KERNEL_EVAL);
        hipLaunchKernelGGL((eval_kernel<KIND, true, W>), dim3((unsigned)slots), dim3(VS_EVAL_ROWS * W), 0, ctx->stream,
                           A, ctx->scratch, tiles);
    }
    VS_HIP(hipGetLastError());
    return VS_OK;
}

template <int KIND>
static int launch_eval(vs_context* ctx, const vs_eval_args& A, int tiles) {
    const int k = A.D.k;
    const size_t lds = sizeof(double) * (size_t)tiles * k * VS_EVAL_ROWS;
    const uint64_t blocks = (A.rows + VS_EVAL_ROWS - 1) / VS_EVAL_ROWS;
    VS_REQUIRE(blocks <= 0x7fffffffull, "too many rows for one launch");
    const bool scratch = lds > 160 * 1024 || ctx->tune.eval_scratch == 1;
    constexpr bool SEPARABLE = KIND == VS_OBJ_G_FUNCTION || KIND == VS_OBJ_POLYNOMIAL || KIND == VS_OBJ_LINEAR;
    // waves per workgroup: with a big tile only one or two workgroups fit a CU, and eight waves
    // sharing the tile keep the SIMDs busier than four (measured: k=48 4 waves 0.72 / 8 waves 0.79 ms,
    // k=64 1.12 / 0.98, k=96 2.79 / 2.07, k=128 9.0 / 6.9; sixteen waves were never better)
    int W = 4;
    if (SEPARABLE && !scratch && lds > 56 * 1024) W = 8;
    if (SEPARABLE && ctx->tune.eval_waves) W = ctx->tune.eval_waves == 8 ? 8 : 4;
    if (SEPARABLE && W == 8) return launch_eval_w<KIND, 8>(ctx, A, tiles, lds, blocks, scratch);
    return launch_eval_w<KIND, 4>(ctx, A, tiles, lds, blocks, scratch);
}

// ---------------------------------------------------------------------------
// fused evaluate + first-order reduce: no f-values are written (kernels: vs_scan.hip, vs_rtc.hip)
// ---------------------------------------------------------------------------

// vs_scan.hip
bool vs_scan_covers(int kind, int k, bool fused);
int vs_scan_eval(vs_context* ctx, const vs_eval_args& A);
int vs_scan_fused_plan(vs_context* ctx, const vs_eval_args& A, unsigned* P);
int vs_scan_fused(vs_context* ctx, const vs_eval_args& A, unsigned P, const double* shift_dev);
unsigned vs_fused_ishigami_workgroups(vs_context* ctx, uint64_t rows);
int vs_fused_ishigami(vs_context* ctx, const vs_eval_args& A, unsigned P, const double* shift_dev, double* gram_partials);
int vs_ishigami_eval(vs_context* ctx, const vs_eval_args& A);

static bool rows_in_perm_slice(const vs_design* g, uint64_t row_begin, uint64_t row_count) {
    if (!g->perm) return true;
    return row_begin >= g->d.perm_begin && row_begin + row_count <= g->d.perm_begin + g->d.perm_count;
}

// The scan kernels inline only the compact generator paths: Halton indices must stay below 2^32
// (per-column constant digit counts), which every design with n < 2^31 - first_index/2 meets.
static bool design_compact(const vs_design* g) {
    return g->d.generator != VS_GEN_HALTON || g->d.first_index + 2 * g->d.n - 1 < (1ull << 32);
}

enum fused_route { FUSED_NONE = 0, FUSED_SCAN, FUSED_ISHIGAMI, FUSED_USER };

// which fused kernel serves (design, objective): the scan kernels, Ishigami's own, or a run-time
// compiled one.  A context in strict
// evaluation order (vs_context_set_eval_order) has no fused path for separable objectives -- the fused
// kernels fold in scan order.
static fused_route fused_route_for(const vs_context* ctx, const vs_design* design, const vs_objective* obj) {
    const bool separable = obj->kind == VS_OBJ_G_FUNCTION || obj->kind == VS_OBJ_POLYNOMIAL || obj->kind == VS_OBJ_LINEAR;
    // power scaling: the built-in objectives have fused instances for it (GENV 3); a run-time compiled one does not
    const bool scan_ok = design_compact(design) && ctx->tune.eval_mode != 1;
    if (separable && scan_ok && vs_scan_covers(obj->kind, obj->k, true)) return FUSED_SCAN;
    if (obj->kind == VS_OBJ_ISHIGAMI) return FUSED_ISHIGAMI;
    if (obj->kind == VS_OBJ_SOURCE && obj->fused_function && scan_ok && design->d.scale_mode != VS_SCALE_POWER)
        return FUSED_USER;
    return FUSED_NONE;
}

extern "C" int vs_eval_reduce_supported(vs_context* ctx, const vs_design* design, const vs_objective* obj,
                                        int* supported) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && design && obj && supported, "null argument");
    VS_HIP(hipSetDevice(ctx->device));
    *supported = (obj->k == design->d.k && fused_route_for(ctx, design, obj) != FUSED_NONE) ? 1 : 0;
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_eval_reduce_first_order(vs_context* ctx, const vs_design* design, const vs_objective* obj,
                                          uint64_t row_begin, uint64_t row_count, int rank, int max_ranks,
                                          double* acc_dev) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && design && obj && acc_dev, "null argument");
    VS_REQUIRE(obj->k == design->d.k, "objective has k=%d, design has k=%d", obj->k, design->d.k);
    VS_REQUIRE(row_begin <= design->d.n && row_count <= design->d.n - row_begin, "rows exceed n");
    VS_REQUIRE(max_ranks >= 1 && rank >= 0 && rank < max_ranks, "rank %d not in [0,%d)", rank, max_ranks);
    VS_REQUIRE(row_count >= 1, "the fused path needs at least one row");
    VS_REQUIRE(rows_in_perm_slice(design, row_begin, row_count), "rows lie outside the design's shuffle slice");
    VS_HIP(hipSetDevice(ctx->device));
    const int k = obj->k, l = obj->l;
    const fused_route route = fused_route_for(ctx, design, obj);
    if (route == FUSED_NONE)
        return vs_fail(VS_ERR_ARG, "the fused path covers G_FUNCTION / POLYNOMIAL / LINEAR objectives and separable "
                                   "source objectives with k <= 128 (identity or linear scaling), and ISHIGAMI; "
                                   "use vs_eval + vs_reduce_first_order (vs_eval_reduce_supported tells)");
    vs_eval_args A;
    A.D = to_dev(design);
    A.kind = obj->kind;
    A.plain_recip = obj->recip_ok ? 1 : 0;
    A.f_shift = ctx->tune.f_shift;
    A.l = l;
    A.params = obj->params;
    A.row_begin = row_begin;
    A.rows = row_count;
    A.f = nullptr;
    unsigned P = 1;
    int rc = VS_OK;
    if (route == FUSED_ISHIGAMI) P = vs_fused_ishigami_workgroups(ctx, row_count);
    else if (route == FUSED_USER) rc = vs_rtc_fused_plan(ctx, obj, row_count, &P, A.D.generator == VS_GEN_SOBOL);
    else rc = vs_scan_fused_plan(ctx, A, &P);
    if (rc) return rc;
    const size_t plen = VS_PLEN(k), width = (size_t)l * plen;
    // workspace: [P][l][plen] partials + [l][plen] folded + the f-values of one design row (l planes of 2k+2)
    const size_t shift_stride = 2 * (size_t)k + 2;
    rc = vs_workspace_partials(ctx, sizeof(double) * ((size_t)P * width + width + (size_t)l * shift_stride));
    if (rc) return rc;
    double* shift_buf = ctx->partials + (size_t)P * width + width;
    // the variance shift of output o is f_o(M_1[row_begin]): a one-row launch of the ordinary evaluation
    rc = vs_eval(ctx, design, obj, row_begin, 1, shift_buf);
    if (rc) return rc;
    switch (route) {
        case FUSED_SCAN: rc = vs_scan_fused(ctx, A, P, shift_buf); break;
        case FUSED_ISHIGAMI: rc = vs_fused_ishigami(ctx, A, P, shift_buf, nullptr); break;
        case FUSED_USER: rc = vs_rtc_launch_fused(ctx, obj, &A, sizeof(A), P, shift_buf, A.D.generator == VS_GEN_SOBOL); break;
        default: break;
    }
    if (rc) return rc;
    return vs_fold_partials(ctx, P, k, l, row_count, shift_buf, (int)shift_stride, nullptr, rank, max_ranks, acc_dev);
    VS_GUARD_END
}

// vs_scan_gram.hip
int vs_scan_gram_launch(vs_context* ctx, const vs_eval_args& A, int nbh, unsigned* P_io, const double* shift_dev,
                        double* first_partials, double* gram_partials, bool plan_only);
// vs_scan_gram8.hip, vs_scan_gram4.hip
int vs_scan_gram8_launch(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                         double* first_partials, double* gram_partials, bool plan_only);
int vs_scan_gram4_launch(vs_context* ctx, const vs_eval_args& A, unsigned* P_io, const double* shift_dev,
                         double* first_partials, double* gram_partials, bool plan_only);
// vs_scan_gram.hip -> vs_scan_gramm_*.hip: several outputs
int vs_scan_gram_multi_lmax(int nb);
int vs_scan_gram_multi_launch(vs_context* ctx, const vs_eval_args& A, int nb, unsigned* P_io, const double* shift_dev,
                              double* first_partials, double* gram_partials, bool plan_only);

// 16-row blocks per half of the fused both-orders kernel that takes K index rows
static int fused_both_blocks(int K) { return K <= 16 ? 1 : K <= 32 ? 2 : K <= 64 ? 4 : 8; }

// the fused both-orders kernels: separable objectives in scan form with k l <= 128 index rows -- built-in
// (vs_scan_gram.hip, vs_scan_gram4.hip, vs_scan_gram8*.hip; several outputs: vs_scan_gramm_*.hip) or, with one output,
// compiled at run time --, and Ishigami
static fused_route fused_both_route(const vs_context* ctx, const vs_design* design, const vs_objective* obj) {
    if (obj->k != design->d.k) return FUSED_NONE;
    if (obj->l != 1) {
        // several outputs of a built-in separable objective: 2 .. 8 outputs (2 or 4 beyond 64 index rows -- the eight-wave
        // kernel; at least 16 columns from 33 index rows on), every scaling (power: not 33 .. 64 index rows)
        const long K = (long)obj->k * obj->l;
        if (K > 128 || fused_route_for(ctx, design, obj) != FUSED_SCAN || ctx->tune.fused_both_wide == 0)
            return FUSED_NONE;
        const int nb = fused_both_blocks((int)K);
        // eight waves: 2 or 4 outputs; three up to 112 index rows (seven live blocks: vs_gramn_consume_row)
        if (obj->l > vs_scan_gram_multi_lmax(nb) || (nb == 8 && 16 % obj->l != 0 && K > 112)) return FUSED_NONE;
        // 33 .. 64 index rows with fewer than 16 columns (five or more outputs, or 3 x 11 .. 15): the four-wave kernel
        // spends its time in the outputs' barriers and walks and loses to vs_eval + vs_reduce_first_second_order
        // (profiles/r04/fused_both_several_outputs_sweep.txt: k=8 l=8 3.25 vs 2.54 ms, k=12 l=3 1.74 vs 1.55; from 16
        // columns on it wins: k=16 l=4 2.37 vs 2.49, k=20 l=3 2.25 vs 2.49, k=32 l=2 2.19 vs 2.46)
        if (nb == 4 && obj->k < 16 && ctx->tune.fused_both_wide != 2) return FUSED_NONE;
        // power scaling: the four-wave kernel of 33 .. 64 index rows loses there too (k=16 l=4 3.09 vs 2.44 ms, k=32 l=2
        // 3.75 vs 2.78); below and above it wins (k=16 l=2 1.01 vs 1.32, k=32 l=4 5.00 vs 5.52, k=64 l=2 5.35 vs 6.17)
        if (nb == 4 && design->d.scale_mode == VS_SCALE_POWER && ctx->tune.fused_both_wide != 2) return FUSED_NONE;
        return FUSED_SCAN;
    }
    const fused_route r = fused_route_for(ctx, design, obj);
    if (r == FUSED_ISHIGAMI) return r;
    const bool wide_k = obj->k > 32 && obj->k <= 128;   // four or eight waves x 16 columns, compiled per count of live blocks
    if (r == FUSED_SCAN && (obj->k <= 32 || (wide_k && ctx->tune.fused_both_wide != 0))) return r;
    if (r == FUSED_USER && obj->gram_function) return r;
    return FUSED_NONE;
}

extern "C" int vs_eval_reduce_both_supported(vs_context* ctx, const vs_design* design, const vs_objective* obj,
                                             int* supported) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && design && obj && supported, "null argument");
    *supported = fused_both_route(ctx, design, obj) != FUSED_NONE ? 1 : 0;
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_eval_reduce_first_second_order(vs_context* ctx, const vs_design* design, const vs_objective* obj,
                                                 uint64_t row_begin, uint64_t row_count, int rank, int max_ranks,
                                                 double* acc_dev, double* gram_dev) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && design && obj && acc_dev && gram_dev, "null argument");
    VS_REQUIRE(obj->k == design->d.k, "objective has k=%d, design has k=%d", obj->k, design->d.k);
    VS_REQUIRE(row_begin <= design->d.n && row_count <= design->d.n - row_begin, "rows exceed n");
    VS_REQUIRE(max_ranks >= 1 && rank >= 0 && rank < max_ranks, "rank %d not in [0,%d)", rank, max_ranks);
    VS_REQUIRE(row_count >= 1, "the fused path needs at least one row");
    VS_REQUIRE(rows_in_perm_slice(design, row_begin, row_count), "rows lie outside the design's shuffle slice");
    VS_HIP(hipSetDevice(ctx->device));
    const fused_route route = fused_both_route(ctx, design, obj);
    if (route == FUSED_NONE)
        return vs_fail(VS_ERR_ARG, "the fused both-orders path covers ISHIGAMI, G_FUNCTION / POLYNOMIAL / LINEAR with k l <= 128 "
                                   "(several outputs: 2 .. 8 up to 32 index rows, k >= 16 beyond, 2 .. 4 outputs beyond 64 rows -- 3 only up to 112 --, "
                                   "power scaling everywhere but at 33 .. 64 index rows of several outputs) and separable "
                                   "source objectives with one output and k <= 128 (Halton indices below 2^32, scan evaluation "
                                   "order); use vs_eval + vs_reduce_first_second_order (vs_eval_reduce_both_supported tells)");
    const int k = obj->k, l = obj->l, K = k * l;
    const bool multi = l > 1;                          // (FUSED_SCAN only)
    const bool wide = route == FUSED_SCAN && K > 32;   // 33 .. 64 / 65 .. 128 index rows: four / eight waves
    const int nbh = fused_both_blocks(K), SB = 16 * nbh;   // (= obj->gram_sb of a source objective)
    vs_eval_args A;
    A.D = to_dev(design);
    A.kind = obj->kind;
    A.plain_recip = obj->recip_ok ? 1 : 0;
    A.f_shift = ctx->tune.f_shift;
    A.l = l;
    A.params = obj->params;
    A.row_begin = row_begin;
    A.rows = row_count;
    A.f = nullptr;
    unsigned P = 1;
    int rc = VS_OK;
    if (route == FUSED_ISHIGAMI) P = vs_fused_ishigami_workgroups(ctx, row_count);
    else if (route == FUSED_USER) rc = vs_rtc_gram_plan(ctx, obj, row_count, &P, A.D.generator == VS_GEN_SOBOL);
    else if (multi) rc = vs_scan_gram_multi_launch(ctx, A, nbh, &P, nullptr, nullptr, nullptr, true);
    else if (wide) rc = nbh == 8 ? vs_scan_gram8_launch(ctx, A, &P, nullptr, nullptr, nullptr, true)
                                 : vs_scan_gram4_launch(ctx, A, &P, nullptr, nullptr, nullptr, true);
    else rc = vs_scan_gram_launch(ctx, A, nbh, &P, nullptr, nullptr, nullptr, true);
    if (rc) return rc;
    // workspace: [P][l][plen] first-order records + [l][plen] folded (vs_fold_partials' layout) + the f-values of one
    // design row (the variance shifts: l planes of 2k+2), then [P][2][SB][SB] Gram partials
    const size_t plen = VS_PLEN(k), width = (size_t)l * plen, shift_stride = 2 * (size_t)k + 2;
    size_t first_doubles = (size_t)P * width + width + (size_t)l * shift_stride;
    first_doubles = (first_doubles + 1) & ~(size_t)1;
    const size_t gram_doubles = (size_t)P * 2 * SB * SB;
    rc = vs_workspace_partials(ctx, sizeof(double) * (first_doubles + gram_doubles));
    if (rc) return rc;
    double* shift_buf = ctx->partials + (size_t)P * width + width;
    double* gram_partials = ctx->partials + first_doubles;
    rc = vs_eval(ctx, design, obj, row_begin, 1, shift_buf);   // f_o(M_1[row_begin]): a one-row evaluation
    if (rc) return rc;
    if (route == FUSED_ISHIGAMI) {   // its kernel writes 3 x 3 entries of every 16 x 16 block; the fold reads only those
        rc = vs_fused_ishigami(ctx, A, P, shift_buf, gram_partials);
    } else if (route == FUSED_USER) {
        rc = vs_rtc_launch_gram(ctx, obj, &A, sizeof(A), P, shift_buf, ctx->partials, gram_partials,
                                A.D.generator == VS_GEN_SOBOL);
    } else if (multi) {
        rc = vs_scan_gram_multi_launch(ctx, A, nbh, &P, shift_buf, ctx->partials, gram_partials, false);
    } else if (wide) {
        rc = nbh == 8 ? vs_scan_gram8_launch(ctx, A, &P, shift_buf, ctx->partials, gram_partials, false)
                      : vs_scan_gram4_launch(ctx, A, &P, shift_buf, ctx->partials, gram_partials, false);
    } else {
        rc = vs_scan_gram_launch(ctx, A, nbh, &P, shift_buf, ctx->partials, gram_partials, false);
    }
    if (rc) return rc;
    rc = vs_fold_partials(ctx, P, k, l, row_count, shift_buf, (int)shift_stride, nullptr, rank, max_ranks, acc_dev);
    if (rc) return rc;
    return vs_gram_fold_st(ctx, gram_partials, P, K, SB, gram_dev);
    VS_GUARD_END
}

extern "C" int vs_eval(vs_context* ctx, const vs_design* design, const vs_objective* obj,
                       uint64_t row_begin, uint64_t row_count, double* f_dev) {
    VS_GUARD_BEGIN
    VS_REQUIRE(ctx && design && obj, "null argument");
    VS_HIP(hipSetDevice(ctx->device));
    VS_REQUIRE(obj->k == design->d.k, "objective has k=%d, design has k=%d", obj->k, design->d.k);
    VS_REQUIRE(row_begin <= design->d.n && row_count <= design->d.n - row_begin,
               "rows [%llu,+%llu) exceed n=%llu", (unsigned long long)row_begin,
               (unsigned long long)row_count, (unsigned long long)design->d.n);
    if (row_count == 0) return VS_OK;
    VS_REQUIRE(f_dev, "f_dev is null");
    VS_REQUIRE(rows_in_perm_slice(design, row_begin, row_count),
               "rows [%llu,+%llu) lie outside the design's shuffle slice [%llu,+%llu)", (unsigned long long)row_begin,
               (unsigned long long)row_count, (unsigned long long)design->d.perm_begin,
               (unsigned long long)design->d.perm_count);
    vs_eval_args A;
    A.D = to_dev(design);
    A.kind = obj->kind;
    A.plain_recip = obj->recip_ok ? 1 : 0;
    A.f_shift = ctx->tune.f_shift;
    A.l = obj->l;
    A.params = obj->params;
    A.row_begin = row_begin;
    A.rows = row_count;
    A.f = f_dev;
    VS_HIP(hipSetDevice(ctx->device));
    const int sep_tiles = obj->l > 1 ? 4 : 2;
    // separable built-ins: the segmented scan (O(k) folds per row, association order of
    // vs_device_scan.h) wherever it applies; the chain scheme (strict left-to-right folds, O(k^2)) beyond
    // 256 columns, for Halton indices beyond 2^32, and when the context asks for the strict order
    if (vs_scan_covers(obj->kind, obj->k, false) &&
        design_compact(design) && ctx->tune.eval_scratch != 1 &&
        ctx->tune.eval_mode != 1)
        return vs_scan_eval(ctx, A);
    switch (obj->kind) {
        case VS_OBJ_G_FUNCTION: return launch_eval<VS_OBJ_G_FUNCTION>(ctx, A, sep_tiles);
        case VS_OBJ_POLYNOMIAL: return launch_eval<VS_OBJ_POLYNOMIAL>(ctx, A, sep_tiles);
        case VS_OBJ_LINEAR: return launch_eval<VS_OBJ_LINEAR>(ctx, A, sep_tiles);
        case VS_OBJ_ISHIGAMI:   // thread-per-row kernel (same operations, same bits); the tile kernel only when forced
            if (ctx->tune.eval_scratch != 1) return vs_ishigami_eval(ctx, A);
            return launch_eval<VS_OBJ_ISHIGAMI>(ctx, A, 2);
        case VS_OBJ_INVERT: return launch_eval<VS_OBJ_INVERT>(ctx, A, 2);
        case VS_OBJ_SOURCE: {
            const size_t lds = obj->eval_scan ? obj->eval_lds : sizeof(double) * 2 * (size_t)design->d.k * VS_EVAL_ROWS;
            const uint64_t blocks = (row_count + VS_EVAL_ROWS - 1) / VS_EVAL_ROWS;
            VS_REQUIRE(blocks <= 0x7fffffffull, "too many rows for one launch");
            VS_REQUIRE(!obj->eval_scan || design_compact(design),
                       "separable source objectives need Halton indices below 2^32 (or a context in strict evaluation order)");
            return vs_rtc_launch(ctx, obj, &A, sizeof(A), (unsigned)blocks, lds, A.D.generator == VS_GEN_SOBOL);
        }
    }
    return vs_fail(VS_ERR_ARG, "unknown objective kind %d", obj->kind);
    VS_GUARD_END
}
