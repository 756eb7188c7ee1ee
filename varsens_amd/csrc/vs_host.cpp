// Host-only part of libvarsens_hip.so: no HIP header, no GPU call.  Everything here also builds with a plain
// C++ compiler, which is how the sanitizer leg of the CPU tests runs it (python -m varsens_amd.build --asan-host:
// g++ -fsanitize=address,undefined over this file + tests/host_asan_main.cpp; GPU sanitizers are not available on
// the pool).
//   * the error store behind vs_last_error();
//   * vs_seed1_permutation: numpy.random.seed(1); numpy.random.shuffle(M_2) (varsens/saltelli.py:100-101);
//   * vs_design_desc_validate: every check vs_design_create makes on a descriptor before it touches the GPU
//     (varsens/saltelli.py:62-108: k, n, generator, scaling bounds, the shuffle slice of a rank);
//   * vs_text_write / vs_text_read: the text wire format of Sample.export / .load and Objective.export / .load
//     (varsens/saltelli.py:173-250, 393-447: numpy.savetxt / numpy.loadtxt), formatted and parsed on several threads;
//   * the helpers of vs_host.h that allocate on the host (the process-wide cache of seed-1 shuffles, the Sobol byte
//     tables): the sanitizer driver runs them with an `operator new` that fails on demand and expects status codes.
#include <algorithm>
#include <charconv>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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


// ---- the text wire format ---------------------------------------------------------------------
// Sample.export / Objective.export write their arrays with numpy.savetxt (varsens/saltelli.py:187,192,407,412):
// every value as '%.18e', the values of a row joined by the delimiter (tab for samples, a space for objectives), one row
// per line.  Sample.load / Objective.load read them back with numpy.loadtxt (:217,439).  These files are how the
// reference's users fan a study out over a cluster (varsens/cluster/accre-job.sh:13), and numpy spends 0.5 us per value
// writing and 0.25 us reading them on one core: a k = 32, n = 2^14 design (34.6 M values, 870 MB of text) takes 18 s to
// write.  Here the rows are formatted (std::to_chars: the digits of printf's %.18e, which are Python's) and parsed
// (std::from_chars / strtod: correctly rounded, as Python's float()) on several threads; the bytes written are
// numpy.savetxt's and the values read are numpy.loadtxt's (tests/test_abi_and_host.py, golden G14).
namespace {

int text_threads(uint64_t work_items) {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 4;
    uint64_t t = std::min<uint64_t>(hw, 8);
    if (work_items / 4096 + 1 < t) t = work_items / 4096 + 1;    // small files: not worth a thread each
    return (int)t;
}

// runs fn(t) for t = 0 .. T-1 on T threads (the caller is t = 0); an exception in any of them is rethrown here
template <class F>
void on_threads(int T, F&& fn) {
    if (T <= 1) {
        fn(0);
        return;
    }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err((size_t)T);
    th.reserve((size_t)T - 1);
    auto guarded = [&](int t) {
        try {
            fn(t);
        } catch (...) {
            err[(size_t)t] = std::current_exception();
        }
    };
    try {
        for (int t = 1; t < T; ++t) th.emplace_back(guarded, t);
    } catch (...) {   // a thread could not be started: the ones that run are joined, their ranges are not done
        for (auto& x : th) x.join();
        throw;
    }
    guarded(0);
    for (auto& x : th) x.join();
    for (auto& e : err)
        if (e) std::rethrow_exception(e);
}

// one value as numpy.savetxt's '%.18e' writes it; returns the end of the text
char* format_value(char* p, char* end, double x) {
    if (x != x) {   // Python prints every NaN as 'nan' (printf would say -nan for a set sign bit)
        memcpy(p, "nan", 3);
        return p + 3;
    }
    const std::to_chars_result r = std::to_chars(p, end, x, std::chars_format::scientific, 18);
    return r.ptr;    // "inf" / "-inf" and d.dddddddddddddddddde+XX as printf and Python's % operator give them
}

bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }

// one token as numpy.loadtxt's float() reads it; false: not a number
bool parse_value(const char* b, const char* e, double* out) {
    while (b < e && is_blank(*b)) ++b;
    while (e > b && is_blank(e[-1])) --e;
    if (b == e) return false;
    const std::from_chars_result r = std::from_chars(b, e, *out);
    if (r.ec == std::errc() && r.ptr == e) return true;
    // what from_chars refuses and float() takes: a leading '+', ...: strtod on a terminated copy
    char tmp[64];
    const size_t len = (size_t)(e - b);
    if (len >= sizeof(tmp)) return false;
    memcpy(tmp, b, len);
    tmp[len] = 0;
    char* stop = nullptr;
    const double v = strtod(tmp, &stop);
    if (stop != tmp + len) return false;
    *out = v;
    return true;
}

struct text_line {
    const char* b;
    const char* e;   // without the comment and the line end
};

// the data part of the line starting at p (up to end): false at the end of the buffer; *next = start of the next line
bool next_line(const char* p, const char* end, text_line* L, const char** next) {
    if (p >= end) return false;
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* le = nl ? nl : end;
    *next = nl ? nl + 1 : end;
    const char* hash = (const char*)memchr(p, '#', (size_t)(le - p));   // numpy.loadtxt: comments='#'
    if (hash) le = hash;
    L->b = p;
    L->e = le;
    return true;
}

bool line_is_empty(const text_line& L) {
    for (const char* c = L.b; c < L.e; ++c)
        if (!is_blank(*c)) return false;
    return true;
}

// splits a data line at the delimiter (dl == 0: runs of blanks) and hands the tokens to tok(b, e); returns their number
template <class F>
uint64_t split_line(const text_line& L, const char* delim, size_t dl, F&& tok) {
    uint64_t n = 0;
    const char* p = L.b;
    if (dl == 0) {
        for (;;) {
            while (p < L.e && is_blank(*p)) ++p;
            if (p >= L.e) break;
            const char* q = p;
            while (q < L.e && !is_blank(*q)) ++q;
            tok(p, q);
            ++n;
            p = q;
        }
        return n;
    }
    for (;;) {
        const char* q = p;
        const char* hit = nullptr;
        for (; q + dl <= L.e; ++q)
            if (*q == delim[0] && memcmp(q, delim, dl) == 0) {
                hit = q;
                break;
            }
        tok(p, hit ? hit : L.e);
        ++n;
        if (!hit) break;
        p = hit + dl;
    }
    return n;
}

}  // namespace

extern "C" int vs_text_write(const char* path, const double* values, uint64_t rows, uint64_t cols, const char* delimiter) {
    VS_GUARD_BEGIN
    VS_REQUIRE(path && delimiter, "null argument");
    VS_REQUIRE(rows == 0 || cols == 0 || values, "values is null");
    VS_REQUIRE(cols <= (1ull << 24), "too many columns");
    const size_t dl = strlen(delimiter);
    FILE* fh = fopen(path, "wb");
    if (!fh) return vs_fail(VS_ERR_ARG, "cannot open %.500s for writing", path);
    struct closer {
        FILE* f;
        ~closer() { if (f) fclose(f); }
    } hold{fh};
    if (rows && cols) {
        const size_t per_value = 26 + dl;   // -d.dddddddddddddddddde+XXX and the delimiter / newline
        const uint64_t batch_rows = std::max<uint64_t>(1, (1ull << 22) / cols);   // ~4 M values (100 MB of text) per batch
        std::vector<std::string> text;
        for (uint64_t r0 = 0; r0 < rows; r0 += batch_rows) {
            const uint64_t nr = std::min(batch_rows, rows - r0);
            const int T = text_threads(nr * cols);
            text.resize((size_t)T);
            on_threads(T, [&](int t) {
                const uint64_t lo = r0 + nr * (uint64_t)t / T, hi = r0 + nr * (uint64_t)(t + 1) / T;
                std::string& out = text[(size_t)t];
                out.resize((size_t)((hi - lo) * cols) * per_value);
                char* p = &out[0];
                char* const end = p + out.size();
                for (uint64_t r = lo; r < hi; ++r) {
                    const double* row = values + r * cols;
                    for (uint64_t c = 0; c < cols; ++c) {
                        p = format_value(p, end, row[c]);
                        if (c + 1 < cols) {
                            memcpy(p, delimiter, dl);
                            p += dl;
                        }
                    }
                    *p++ = '\n';
                }
                out.resize((size_t)(p - &out[0]));
            });
            for (int t = 0; t < T; ++t) {
                const std::string& out = text[(size_t)t];
                if (!out.empty() && fwrite(out.data(), 1, out.size(), fh) != out.size())
                    return vs_fail(VS_ERR_STATE, "short write to %.500s", path);
            }
        }
    }
    hold.f = nullptr;
    if (fclose(fh) != 0) return vs_fail(VS_ERR_STATE, "closing %.500s failed", path);
    return VS_OK;
    VS_GUARD_END
}

extern "C" int vs_text_read(const char* path, const char* delimiter, double* out, uint64_t capacity, uint64_t* rows,
                            uint64_t* cols) {
    VS_GUARD_BEGIN
    VS_REQUIRE(path && rows && cols, "null argument");
    *rows = *cols = 0;
    const size_t dl = delimiter ? strlen(delimiter) : 0;
    // the file is mapped, not copied: the threads below fault its pages in side by side
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return vs_fail(VS_ERR_ARG, "cannot open %.500s", path);
    struct mapping {
        int fd;
        void* base = MAP_FAILED;
        size_t size = 0;
        ~mapping() {
            if (base != MAP_FAILED) munmap(base, size);
            close(fd);
        }
    } file{fd};
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return vs_fail(VS_ERR_ARG, "%.500s is not a regular file", path);
    file.size = (size_t)st.st_size;
    if (file.size) {
        file.base = mmap(nullptr, file.size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (file.base == MAP_FAILED) return vs_fail(VS_ERR_STATE, "cannot map %.500s", path);
    }
    const char* const begin = file.size ? (const char*)file.base : "";
    const char* const end = begin + file.size;
    const size_t nbytes = file.size;

    // byte ranges that start at line starts, one per thread
    const int T = text_threads((uint64_t)nbytes / 20);
    std::vector<const char*> cut((size_t)T + 1);
    cut[0] = begin;
    cut[(size_t)T] = end;
    for (int t = 1; t < T; ++t) {
        const char* p = begin + nbytes / (size_t)T * (size_t)t;
        if (p < cut[(size_t)t - 1]) p = cut[(size_t)t - 1];
        const char* nl = p < end ? (const char*)memchr(p, '\n', (size_t)(end - p)) : nullptr;
        cut[(size_t)t] = nl ? nl + 1 : end;
    }
    // pass 1: data lines per range, columns of each range's first data line
    std::vector<uint64_t> nlines((size_t)T, 0), ncols((size_t)T, 0);
    on_threads(T, [&](int t) {
        const char* p = cut[(size_t)t];
        const char* const stop = cut[(size_t)t + 1];
        text_line L;
        const char* next;
        while (p < stop && next_line(p, stop, &L, &next)) {
            if (!line_is_empty(L)) {
                if (nlines[(size_t)t] == 0) ncols[(size_t)t] = split_line(L, delimiter, dl, [](const char*, const char*) {});
                ++nlines[(size_t)t];
            }
            p = next;
        }
    });
    uint64_t total = 0, width = 0;
    for (int t = 0; t < T; ++t) {
        if (nlines[(size_t)t] && !width) width = ncols[(size_t)t];
        total += nlines[(size_t)t];
    }
    *rows = total;
    *cols = width;
    if (!out) return VS_OK;   // the sizing call
    VS_REQUIRE(total * width <= capacity, "%.500s holds %llu x %llu values, the buffer %llu", path,
               (unsigned long long)total, (unsigned long long)width, (unsigned long long)capacity);
    // pass 2: every range parses its lines straight into its rows of `out`
    std::vector<uint64_t> first_row((size_t)T, 0);
    for (int t = 1; t < T; ++t) first_row[(size_t)t] = first_row[(size_t)t - 1] + nlines[(size_t)t - 1];
    std::vector<uint64_t> bad_line((size_t)T, 0);   // 1-based data row of the first malformed line of a range, or 0
    on_threads(T, [&](int t) {
        const char* p = cut[(size_t)t];
        const char* const stop = cut[(size_t)t + 1];
        uint64_t r = first_row[(size_t)t];
        text_line L;
        const char* next;
        while (p < stop && next_line(p, stop, &L, &next)) {
            if (!line_is_empty(L)) {
                double* dst = out + r * width;
                uint64_t c = 0;
                bool ok = true;
                const uint64_t got = split_line(L, delimiter, dl, [&](const char* b, const char* e) {
                    double v = 0.0;
                    if (c < width && parse_value(b, e, &v)) dst[c] = v;
                    else ok = false;
                    ++c;
                });
                if ((!ok || got != width) && !bad_line[(size_t)t]) bad_line[(size_t)t] = r + 1;
                ++r;
            }
            p = next;
        }
    });
    for (int t = 0; t < T; ++t)
        if (bad_line[(size_t)t])
            return vs_fail(VS_ERR_ARG, "%.500s: data row %llu does not hold %llu numbers", path,
                           (unsigned long long)bad_line[(size_t)t], (unsigned long long)width);
    return VS_OK;
    VS_GUARD_END
}
