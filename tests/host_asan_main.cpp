// Driver of the sanitizer leg (python -m varsens_amd.build --asan-host): the host-only entry points of the C ABI
// under AddressSanitizer + UndefinedBehaviorSanitizer, no GPU.  Prints one line per check; the CPU test
// (tests/test_abi_and_host.py::test_host_entry_points_under_sanitizers) compares the permutation checksums with the
// oracle's and expects exit status 0 -- any sanitizer report ends the program with a non-zero status
// (-fno-sanitize-recover=all).
//   * vs_seed1_permutation (varsens/saltelli.py:100-101) into buffers of EXACTLY n entries (heap: a write past the end
//     is a report), n on both sides of the 2048-swap batches;
//   * vs_design_desc_validate (varsens/saltelli.py:62-108): well-formed descriptors, every refusal, a rank's slice of
//     the shuffle in a buffer of exactly perm_count entries, VS_PERM_EMPTY with a one-entry buffer that must not be
//     read beyond, perm_count = 0 (= n entries);
//   * "no C++ exception crosses the boundary" (include/varsens_hip.h): the global `operator new` is replaced below by
//     one that throws std::bad_alloc on the i-th allocation from now; the helpers of csrc/vs_host.h that allocate (the
//     process-wide cache of seed-1 shuffles, the Sobol byte tables) are run with i = 0, 1, 2, ... until they succeed:
//     every failing run must come back as VS_ERR_NOMEM with a message -- not std::terminate --, leak nothing
//     (LeakSanitizer at exit) and leave the helper usable;
//   * vs_text_write / vs_text_read (varsens/saltelli.py:187,217,407,439: numpy.savetxt / numpy.loadtxt) in a scratch
//     directory of their own: the bytes of the special values, a threaded round trip bit for bit into a buffer of
//     exactly rows x cols doubles, every refusal, and the same failing allocator (on whichever thread it hits).
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include <unistd.h>

#include <new>

#include "varsens_hip.h"
#include "vs_host.h"

// ---- an allocator that fails on demand ------------------------------------------------------------------------------
static long g_fail_in = -1;   // < 0: never; 0: the next allocation throws; i: the i-th from now
static long g_allocations = 0;
static std::mutex g_alloc_lock;   // (vs_text_write / vs_text_read allocate on several threads)
static void* counted_alloc(std::size_t n) {
    {
        std::lock_guard<std::mutex> hold(g_alloc_lock);
        ++g_allocations;
        if (g_fail_in == 0) {
            g_fail_in = -1;
            throw std::bad_alloc();
        }
        if (g_fail_in > 0) --g_fail_in;
    }
    void* p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void* operator new(std::size_t n) { return counted_alloc(n); }
void* operator new[](std::size_t n) { return counted_alloc(n); }
void operator delete(void* p) noexcept { std::free(p); }
void operator delete[](void* p) noexcept { std::free(p); }
void operator delete(void* p, std::size_t) noexcept { std::free(p); }
void operator delete[](void* p, std::size_t) noexcept { std::free(p); }

static int failures = 0;
#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                  \
        }                                                                \
    } while (0)

static uint64_t fnv(const uint32_t* p, uint64_t n) {
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = 0; i < n; ++i)
        for (int b = 0; b < 4; ++b) {
            h ^= (p[i] >> (8 * b)) & 0xffu;
            h *= 1099511628211ull;
        }
    return h;
}

int main() {
    CHECK(vs_abi_version() == VS_ABI_VERSION);
    CHECK(std::strlen(vs_last_error()) == 0);

    // ---- the seed-1 shuffle
    const uint64_t sizes[] = {0, 1, 2, 3, 5, 13, 1024, 2047, 2048, 2049, 2050, 4097, 100003, 1u << 20};
    for (uint64_t n : sizes) {
        uint32_t* out = n ? static_cast<uint32_t*>(std::malloc(sizeof(uint32_t) * n)) : nullptr;   // exactly n entries
        const int rc = vs_seed1_permutation(out, n);
        CHECK(rc == VS_OK);
        std::vector<uint8_t> seen(n, 0);
        bool perm = true;
        for (uint64_t i = 0; i < n; ++i) {
            if (out[i] >= n || seen[out[i]]) perm = false;
            else seen[out[i]] = 1;
        }
        CHECK(perm);
        std::printf("perm %" PRIu64 " %016" PRIx64 "\n", n, fnv(out, n));
        std::free(out);
    }
    CHECK(vs_seed1_permutation(nullptr, 0) == VS_OK);
    CHECK(vs_seed1_permutation(nullptr, 7) == VS_ERR_ARG);
    CHECK(std::strstr(vs_last_error(), "null") != nullptr);
    uint32_t one[1];
    CHECK(vs_seed1_permutation(one, 0x100000000ull) == VS_ERR_ARG);

    // ---- descriptors
    const int k = 5;
    const uint64_t n = 1000;
    std::vector<double> lo(k, -1.0), hi(k, 2.0);
    vs_design_desc d;
    std::memset(&d, 0, sizeof(d));
    d.generator = VS_GEN_HALTON;
    d.k = k;
    d.n = n;
    d.first_index = 1 + 20 * k;
    d.scale_mode = VS_SCALE_IDENTITY;
    uint64_t entries = 99;
    CHECK(vs_design_desc_validate(&d, &entries) == VS_OK && entries == 0);
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_OK);
    CHECK(vs_design_desc_validate(nullptr, &entries) == VS_ERR_ARG);

    // the whole shuffle: perm_count = 0 means n entries, all read
    uint32_t* whole = static_cast<uint32_t*>(std::malloc(sizeof(uint32_t) * n));
    CHECK(vs_seed1_permutation(whole, n) == VS_OK);
    d.perm = whole;
    d.perm_begin = 0;
    d.perm_count = 0;
    CHECK(vs_design_desc_validate(&d, &entries) == VS_OK && entries == n);
    // a rank's slice in a buffer of exactly that many entries
    const uint64_t begin = 334, count = 333;
    uint32_t* slice = static_cast<uint32_t*>(std::malloc(sizeof(uint32_t) * count));
    std::memcpy(slice, whole + begin, sizeof(uint32_t) * count);
    d.perm = slice;
    d.perm_begin = begin;
    d.perm_count = count;
    CHECK(vs_design_desc_validate(&d, &entries) == VS_OK && entries == count);
    d.perm_begin = n - count + 1;   // the slice runs past the last row
    CHECK(vs_design_desc_validate(&d, &entries) == VS_ERR_ARG);
    CHECK(std::strstr(vs_last_error(), "exceeds") != nullptr);
    d.perm_begin = begin;
    slice[count - 1] = (uint32_t)n;   // not a row of the second half
    CHECK(vs_design_desc_validate(&d, &entries) == VS_ERR_ARG);
    CHECK(std::strstr(vs_last_error(), "not a row") != nullptr);
    // an empty slice (a rank without rows): perm must be non-null and is NOT read -- a one-entry buffer holding garbage
    uint32_t* none = static_cast<uint32_t*>(std::malloc(sizeof(uint32_t)));
    none[0] = 0xffffffffu;
    d.perm = none;
    d.perm_begin = n;
    d.perm_count = VS_PERM_EMPTY;
    CHECK(vs_design_desc_validate(&d, &entries) == VS_OK && entries == 0);
    d.perm_begin = 17;
    CHECK(vs_design_desc_validate(&d, &entries) == VS_OK && entries == 0);
    d.perm = nullptr;
    d.perm_begin = d.perm_count = 0;

    // scaling, generators, ranges
    d.scale_mode = VS_SCALE_LINEAR;
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_ERR_ARG);
    d.lower = lo.data();
    d.upper = hi.data();
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_OK);
    d.scale_mode = VS_SCALE_POWER;
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_OK);
    d.scale_mode = 3;
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_ERR_ARG);
    d.scale_mode = -1;
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_ERR_ARG);
    d.scale_mode = VS_SCALE_IDENTITY;
    d.generator = VS_GEN_RAW;
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_ERR_ARG);
    std::vector<double> raw(2 * n * k, 0.25);
    d.raw = raw.data();
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_OK);
    d.generator = VS_GEN_SOBOL;
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_ERR_ARG);
    std::vector<uint32_t> dirs(32 * k, 1u);
    d.sobol_directions = dirs.data();
    d.first_index = 4096;
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_OK);
    d.first_index = (1ull << 32) - 2 * n + 1;   // the last index would be 2^32
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_ERR_ARG);
    d.first_index = ~0ull - 5;                   // first_index + 2n wraps around
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_ERR_ARG);
    d.first_index = 4096;
    d.generator = 7;
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_ERR_ARG);
    d.generator = VS_GEN_HALTON;
    d.k = 0;
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_ERR_ARG);
    d.k = 4097;
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_ERR_ARG);
    d.k = k;
    d.n = 0;
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_ERR_ARG);
    d.n = 1ull << 32;
    CHECK(vs_design_desc_validate(&d, nullptr) == VS_ERR_ARG);
    CHECK(std::strstr(vs_last_error(), "out of range") != nullptr);

    std::free(whole);
    std::free(slice);
    std::free(none);

    // ---- allocation failures come back as status codes
    {
        // the entry points that allocate nothing do not care
        g_fail_in = 0;
        uint32_t small[16];
        CHECK(vs_seed1_permutation(small, 16) == VS_OK);
        vs_design_desc e;
        std::memset(&e, 0, sizeof(e));
        e.generator = VS_GEN_HALTON;
        e.k = 3;
        e.n = 10;
        CHECK(vs_design_desc_validate(&e, nullptr) == VS_OK);
        CHECK(g_fail_in == 0);   // (still armed: nothing was allocated)
        g_fail_in = -1;

        // the cache of seed-1 shuffles: fail the i-th allocation until the call gets through
        const uint64_t m = 5000;
        int refused = 0;
        std::shared_ptr<const std::vector<uint32_t>> got;
        for (long i = 0; i < 64; ++i) {
            g_fail_in = i;
            const int rc = vs_seed1_host_cached(m, &got);
            g_fail_in = -1;
            if (rc == VS_OK) break;
            CHECK(rc == VS_ERR_NOMEM);
            CHECK(std::strstr(vs_last_error(), "out of host memory") != nullptr);
            CHECK(!got);
            ++refused;
        }
        CHECK(refused >= 1);   // (it does allocate: the first runs must have been refused)
        CHECK(got && got->size() == m);
        std::vector<uint32_t> direct(m);
        CHECK(vs_seed1_permutation(direct.data(), m) == VS_OK);
        CHECK(got && std::memcmp(got->data(), direct.data(), sizeof(uint32_t) * m) == 0);
        std::shared_ptr<const std::vector<uint32_t>> again;
        const long before = g_allocations;
        CHECK(vs_seed1_host_cached(m, &again) == VS_OK && again.get() == got.get());   // from the cache ...
        CHECK(g_allocations == before);                                                 // ... without allocating
        std::printf("seed-1 cache: %d allocation failures refused with VS_ERR_NOMEM\n", refused);
        // three more sizes push the first ones out of the two-entry cache (the evicted vectors are freed: LeakSanitizer)
        for (uint64_t extra : {7001ull, 7002ull, 7003ull}) {
            std::shared_ptr<const std::vector<uint32_t>> x;
            CHECK(vs_seed1_host_cached(extra, &x) == VS_OK && x && x->size() == extra);
        }
        CHECK(vs_seed1_host_cached(0, &again) == VS_ERR_ARG);
        CHECK(vs_seed1_host_cached(m, nullptr) == VS_ERR_ARG);

        // the Sobol byte tables
        const int kk = 3;
        std::vector<uint32_t> dd(32 * kk);
        for (size_t i = 0; i < dd.size(); ++i) dd[i] = 0x80000000u >> (i % 32);
        std::vector<uint32_t> tabs;
        refused = 0;
        for (long i = 0; i < 64; ++i) {
            g_fail_in = i;
            const int rc = vs_sobol_byte_tables(kk, dd.data(), &tabs);
            g_fail_in = -1;
            if (rc == VS_OK) break;
            CHECK(rc == VS_ERR_NOMEM);
            ++refused;
        }
        CHECK(refused >= 1 && tabs.size() == (size_t)kk * 1024);
        CHECK(tabs[0] == 0 && tabs[1] == 0x80000000u && tabs[3] == 0xc0000000u && tabs[256 + 1] == 0x00800000u);
        CHECK(vs_sobol_byte_tables(0, dd.data(), &tabs) == VS_ERR_ARG);
        std::printf("sobol tables: %d allocation failures refused with VS_ERR_NOMEM\n", refused);
    }
    // ---- the text wire format
    {
        char dir[] = "/tmp/varsens_text_XXXXXX";
        CHECK(mkdtemp(dir) != nullptr);
        const std::string path = std::string(dir) + "/values.txt";
        const double inf = std::numeric_limits<double>::infinity();
        const double special[8] = {1.0, -0.0, std::nan(""), -std::nan(""), inf, -inf, 5e-324, -1.7976931348623157e308};
        CHECK(vs_text_write(path.c_str(), special, 2, 4, "\t") == VS_OK);
        {
            std::FILE* fh = std::fopen(path.c_str(), "rb");
            char text[512] = {0};
            const size_t got = fh ? std::fread(text, 1, sizeof(text) - 1, fh) : 0;
            if (fh) std::fclose(fh);
            const char* expect =
                "1.000000000000000000e+00\t-0.000000000000000000e+00\tnan\tnan\n"
                "inf\t-inf\t4.940656458412465442e-324\t-1.797693134862315708e+308\n";
            CHECK(got == std::strlen(expect) && std::memcmp(text, expect, got) == 0);
        }
        uint64_t rows = 99, cols = 99;
        CHECK(vs_text_read(path.c_str(), "\t", nullptr, 0, &rows, &cols) == VS_OK && rows == 2 && cols == 4);
        double* back = static_cast<double*>(std::malloc(sizeof(double) * 8));   // exactly rows x cols
        CHECK(vs_text_read(path.c_str(), "\t", back, 8, &rows, &cols) == VS_OK);
        CHECK(back[0] == 1.0 && back[1] == 0.0 && std::signbit(back[1]) && back[2] != back[2] && back[3] != back[3]);
        CHECK(back[4] == inf && back[5] == -inf && back[6] == 5e-324 && back[7] == -1.7976931348623157e308);
        CHECK(vs_text_read(path.c_str(), nullptr, back, 8, &rows, &cols) == VS_OK && back[6] == 5e-324);   // any blanks
        CHECK(vs_text_read(path.c_str(), "\t", back, 7, &rows, &cols) == VS_ERR_ARG);                   // buffer too small
        std::free(back);

        // a file large enough for several threads, round trip bit for bit; a space as the delimiter
        const uint64_t R = 3001, Cc = 9;
        std::vector<double> big(R * Cc);
        uint64_t state = 88172645463325252ull;
        for (auto& x : big) {
            state ^= state << 13; state ^= state >> 7; state ^= state << 17;
            uint64_t bits = state;
            if (((bits >> 52) & 0x7ff) == 0x7ff) bits &= ~(1ull << 62);   // (no NaN payloads: text cannot carry them)
            std::memcpy(&x, &bits, 8);
        }
        CHECK(vs_text_write(path.c_str(), big.data(), R, Cc, " ") == VS_OK);
        double* again = static_cast<double*>(std::malloc(sizeof(double) * R * Cc));
        CHECK(vs_text_read(path.c_str(), " ", again, R * Cc, &rows, &cols) == VS_OK && rows == R && cols == Cc);
        CHECK(std::memcmp(again, big.data(), sizeof(double) * R * Cc) == 0);

        // allocation failures, on whichever thread they land: a status code, nothing leaked, the call still usable
        int refused = 0, done = 0;
        for (long i = 0; i < 400 && !done; ++i) {
            g_fail_in = i;
            const int rc = vs_text_write(path.c_str(), big.data(), R, Cc, " ");
            g_fail_in = -1;
            if (rc == VS_OK) done = 1;
            else {
                CHECK(rc == VS_ERR_NOMEM);
                ++refused;
            }
        }
        CHECK(done && refused >= 1);
        std::printf("text write: %d allocation failures refused with VS_ERR_NOMEM\n", refused);
        refused = done = 0;
        for (long i = 0; i < 400 && !done; ++i) {
            g_fail_in = i;
            const int rc = vs_text_read(path.c_str(), " ", again, R * Cc, &rows, &cols);
            g_fail_in = -1;
            if (rc == VS_OK) done = 1;
            else {
                CHECK(rc == VS_ERR_NOMEM);
                ++refused;
            }
        }
        CHECK(done && refused >= 1 && std::memcmp(again, big.data(), sizeof(double) * R * Cc) == 0);
        std::printf("text read: %d allocation failures refused with VS_ERR_NOMEM\n", refused);
        std::free(again);

        // comments, blank lines, CRLF, a last line without its newline, '+' signs; then the refusals
        {
            std::FILE* fh = std::fopen(path.c_str(), "wb");
            std::fputs("# header\n1 2 3\n\n   \n4 5 6 # tail\r\n+7 8e0 .9", fh);
            std::fclose(fh);
            double v[9];
            CHECK(vs_text_read(path.c_str(), nullptr, v, 9, &rows, &cols) == VS_OK && rows == 3 && cols == 3);
            CHECK(v[0] == 1 && v[5] == 6 && v[6] == 7 && v[7] == 8 && v[8] == 0.9);
            fh = std::fopen(path.c_str(), "wb");
            std::fputs("1 2 3\n4 5\n", fh);
            std::fclose(fh);
            CHECK(vs_text_read(path.c_str(), nullptr, v, 9, &rows, &cols) == VS_ERR_ARG);
            CHECK(std::strstr(vs_last_error(), "data row 2") != nullptr);
            fh = std::fopen(path.c_str(), "wb");
            std::fputs("1 2 3\n4 five 6\n", fh);
            std::fclose(fh);
            CHECK(vs_text_read(path.c_str(), nullptr, v, 9, &rows, &cols) == VS_ERR_ARG);
            fh = std::fopen(path.c_str(), "wb");
            std::fclose(fh);
            CHECK(vs_text_read(path.c_str(), nullptr, v, 9, &rows, &cols) == VS_OK && rows == 0 && cols == 0);   // empty
        }
        CHECK(vs_text_write(path.c_str(), nullptr, 0, 5, " ") == VS_OK);   // no rows: an empty file
        CHECK(vs_text_read(path.c_str(), " ", nullptr, 0, &rows, &cols) == VS_OK && rows == 0);
        CHECK(vs_text_write(path.c_str(), nullptr, 2, 2, " ") == VS_ERR_ARG);
        CHECK(vs_text_write(nullptr, special, 2, 4, " ") == VS_ERR_ARG);
        CHECK(vs_text_write(dir, special, 2, 4, " ") == VS_ERR_ARG);        // a directory
        CHECK(vs_text_read(dir, " ", nullptr, 0, &rows, &cols) == VS_ERR_ARG);
        CHECK(vs_text_read((path + ".missing").c_str(), " ", nullptr, 0, &rows, &cols) == VS_ERR_ARG);
        CHECK(vs_text_read(path.c_str(), " ", nullptr, 0, nullptr, &cols) == VS_ERR_ARG);
        unlink(path.c_str());
        rmdir(dir);
    }
    std::printf("%s\n", failures ? "host checks FAILED" : "host checks ok");
    return failures ? 1 : 0;
}
