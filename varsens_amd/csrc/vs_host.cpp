// Host-only part of libvarsens_hip.so: no HIP header, no GPU call.  Everything here also builds with a plain
// C++ compiler, which is how the sanitizer leg of the CPU tests runs it (python -m varsens_amd.build --asan-host:
// g++ -fsanitize=address,undefined over this file + tests/host_asan_main.cpp; GPU sanitizers are not available on
// the pool).
//   * the error store behind vs_last_error();
//   * vs_seed1_permutation: numpy.random.seed(1); numpy.random.shuffle(M_2) (varsens/saltelli.py:100-101);
//   * vs_design_desc_validate: every check vs_design_create makes on a descriptor before it touches the GPU
//     (varsens/saltelli.py:62-108: k, n, generator, scaling bounds, the shuffle slice of a rank);
//   * the helpers of vs_host.h that allocate on the host (the process-wide cache of seed-1 shuffles, the Sobol byte
//     tables): the sanitizer driver runs them with an `operator new` that fails on demand and expects status codes.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <utility>

#include "varsens_hip.h"
#include "vs_guard.h"
#include "vs_host.h"

static thread_local char g_err[8192] = "";

int vs_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#ifndef VS_BUILD_ID
#define VS_BUILD_ID "unknown"   // set by varsens_amd/build.py: sha256 prefix of the sources and flags (source_id())
#endif
extern "C" int vs_abi_version(void) { return VS_ABI_VERSION; }
extern "C" const char* vs_build_id(void) { return VS_BUILD_ID; }
extern "C" const char* vs_last_error(void) { return g_err; }

// ---- the seed-1 row shuffle ---------------------------------------------------------------
//
// numpy.random.seed(1); numpy.random.shuffle(M_2) (varsens/saltelli.py:100-101) is numpy's legacy
// generator: MT19937 seeded by init_genrand(1), Fisher-Yates from the last row down, each index drawn
// by masked rejection on 32-bit outputs.  Restated here so that the permutation does not depend on
// numpy keeping its legacy RandomState; tests pin it to numpy's own output and to golden G7.
namespace {
struct mt19937 {
    uint32_t key[624];
    int pos;
    explicit mt19937(uint32_t seed) {
        for (int i = 0; i < 624; ++i) {
            key[i] = seed;
            seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
    }
    void refill() {
        const uint32_t UP = 0x80000000u, LOW = 0x7fffffffu, MAT = 0x9908b0dfu;
        int i;
        for (i = 0; i < 624 - 397; ++i) {
            uint32_t y = (key[i] & UP) | (key[i + 1] & LOW);
            key[i] = key[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAT);
        }
        for (; i < 623; ++i) {
            uint32_t y = (key[i] & UP) | (key[i + 1] & LOW);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAT);
        }
        uint32_t y = (key[623] & UP) | (key[0] & LOW);
        key[623] = key[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAT);
        pos = 0;
    }
    uint32_t next() {
        if (pos == 624) refill();
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};
}  // namespace

extern "C" int vs_seed1_permutation(uint32_t* out, uint64_t n) {
    VS_GUARD_BEGIN
    VS_REQUIRE(out || n == 0, "out is null");
    VS_REQUIRE(n <= 0xffffffffull, "n=%llu out of range", (unsigned long long)n);
    if (n < 2) {
        if (n) out[0] = 0;
        return VS_OK;
    }
    for (uint64_t i = 0; i < n; ++i) out[i] = (uint32_t)i;
    // Two stages per batch: (1) draw the next BATCH swap partners -- the generator does not depend on the array, so
    // the masked rejection runs branch-free (a rejected draw is overwritten by the next one) and every partner's
    // cache line is requested as soon as it is known; (2) do the swaps, whose random reads then hit the cache.
    // Same draws in the same order as the one-loop form (i = n-1 .. 1, j = first masked draw <= i): twice as fast
    // at n = 2^20 .. 2^22 (1.3 x at 2^24, where the partners also miss the TLB; a huge-page buffer was slower:
    // its page faults cost more than the misses).
    constexpr uint32_t BATCH = 2048;
    uint32_t partner[BATCH + 1];
    mt19937 rng(1u);
    uint32_t i = (uint32_t)(n - 1);
    while (i >= 1) {
        const uint32_t want = i < BATCH ? i : BATCH;   // swaps of this batch: i, i-1, .., i-want+1
        uint32_t got = 0, ii = i;
        while (got < want) {
            const uint32_t mask = 0xffffffffu >> __builtin_clz(ii);   // ii >= 1
            const uint32_t j = rng.next() & mask;
            partner[got] = j;
            __builtin_prefetch(out + j, 1);
            const uint32_t ok = j <= ii ? 1u : 0u;
            got += ok;
            ii -= ok;
        }
        for (uint32_t t = 0; t < want; ++t, --i) {
            const uint32_t j = partner[t];
            const uint32_t v = out[i];
            out[i] = out[j];
            out[j] = v;
        }
    }
    return VS_OK;
    VS_GUARD_END
}


// ---- descriptor checks ----------------------------------------------------------------------
// *perm_entries (may be null): how many entries of desc->perm the engine will read -- 0 without a shuffle or for an
// empty slice (VS_PERM_EMPTY), n for perm_count = 0, else perm_count.
extern "C" int vs_design_desc_validate(const vs_design_desc* desc, uint64_t* perm_entries) {
    VS_GUARD_BEGIN
    VS_REQUIRE(desc, "null descriptor");
    const int k = desc->k;
    const uint64_t n = desc->n;
    VS_REQUIRE(k >= 1 && k <= 4096, "k=%d out of range [1,4096]", k);
    VS_REQUIRE(n >= 1 && n <= 0xffffffffull, "n=%llu out of range [1,2^32)", (unsigned long long)n);
    VS_REQUIRE(desc->generator >= VS_GEN_HALTON && desc->generator <= VS_GEN_RAW, "unknown generator %d",
               desc->generator);
    VS_REQUIRE(desc->scale_mode >= VS_SCALE_IDENTITY && desc->scale_mode <= VS_SCALE_POWER,
               "unknown scale mode %d", desc->scale_mode);
    if (desc->scale_mode != VS_SCALE_IDENTITY)
        VS_REQUIRE(desc->lower && desc->upper, "lower/upper bounds are required for this scale mode");
    if (desc->generator == VS_GEN_RAW) VS_REQUIRE(desc->raw, "raw points are required for VS_GEN_RAW");
    if (desc->generator == VS_GEN_SOBOL) {
        VS_REQUIRE(desc->sobol_directions, "direction numbers are required for VS_GEN_SOBOL");
        VS_REQUIRE(desc->first_index <= (1ull << 32) && desc->first_index + 2 * n <= (1ull << 32),
                   "Sobol indices must stay below 2^32");
    }
    VS_REQUIRE((desc->flags & ~VS_DESIGN_SEED1_SHUFFLE) == 0, "unknown design flags 0x%x", (unsigned)desc->flags);
    if (desc->flags & VS_DESIGN_SEED1_SHUFFLE) {   // the library's own seed-1 shuffle: nothing is read from the host
        VS_REQUIRE(!desc->perm, "VS_DESIGN_SEED1_SHUFFLE: perm must be NULL (the library supplies the shuffle)");
        const uint64_t c = desc->perm_count == VS_PERM_EMPTY ? 0 : desc->perm_count ? desc->perm_count : n;
        VS_REQUIRE(desc->perm_begin <= n && c <= n - desc->perm_begin, "perm slice [%llu,+%llu) exceeds n=%llu",
                   (unsigned long long)desc->perm_begin, (unsigned long long)c, (unsigned long long)n);
        if (perm_entries) *perm_entries = 0;
        return VS_OK;
    }
    // the shuffle may be given for a slice of rows only (a rank's shard): perm_count entries from row perm_begin
    const uint64_t count = !desc->perm ? 0
                           : desc->perm_count == VS_PERM_EMPTY ? 0   // a rank without rows: nothing is read
                           : desc->perm_count ? desc->perm_count : n;
    if (desc->perm) {
        VS_REQUIRE(desc->perm_begin <= n && count <= n - desc->perm_begin,
                   "perm slice [%llu,+%llu) exceeds n=%llu", (unsigned long long)desc->perm_begin,
                   (unsigned long long)count, (unsigned long long)n);
        for (uint64_t i = 0; i < count; ++i)
            VS_REQUIRE(desc->perm[i] < n, "perm[%llu]=%u is not a row of the second half",
                       (unsigned long long)(desc->perm_begin + i), desc->perm[i]);
    }
    if (perm_entries) *perm_entries = count;
    return VS_OK;
    VS_GUARD_END
}


// ---- host helpers that allocate (vs_host.h) ---------------------------------------------------

int vs_seed1_host_cached(uint64_t n, std::shared_ptr<const std::vector<uint32_t>>* out) {
    VS_GUARD_BEGIN
    VS_REQUIRE(out, "out is null");
    VS_REQUIRE(n >= 1 && n <= 0xffffffffull, "n=%llu out of range", (unsigned long long)n);
    static std::mutex lock;
    static std::vector<std::pair<uint64_t, std::shared_ptr<const std::vector<uint32_t>>>> kept;
    std::lock_guard<std::mutex> hold(lock);
    for (auto& h : kept)
        if (h.first == n) {
            *out = h.second;
            return VS_OK;
        }
    auto perm = std::make_shared<std::vector<uint32_t>>(n);   // 4 n bytes; std::bad_alloc -> VS_ERR_NOMEM (guard)
    const int rc = vs_seed1_permutation(perm->data(), n);
    if (rc) return rc;
    if (n <= (1ull << 25)) {   // kept for the next context / slice: up to two, 128 MiB each at most
        kept.reserve(3);
        if (kept.size() >= 2) kept.erase(kept.begin());
        kept.emplace_back(n, perm);
    }
    *out = std::move(perm);
    return VS_OK;
    VS_GUARD_END
}

int vs_sobol_byte_tables(int k, const uint32_t* directions, std::vector<uint32_t>* tabs) {
    VS_GUARD_BEGIN
    VS_REQUIRE(k >= 1 && directions && tabs, "bad argument");
    tabs->assign((size_t)k * 1024, 0u);
    for (int c = 0; c < k; ++c)
        for (int b = 0; b < 4; ++b)
            for (int v = 0; v < 256; ++v) {
                uint32_t x = 0;
                for (int j = 0; j < 8; ++j)
                    if ((v >> j) & 1) x ^= directions[(size_t)c * 32 + 8 * b + j];
                (*tabs)[((size_t)c * 4 + b) * 256 + v] = x;
            }
    return VS_OK;
    VS_GUARD_END
}
