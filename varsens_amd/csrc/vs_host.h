// Host-only helpers of the library that allocate: kept in vs_host.cpp (no HIP header) so that the sanitizer driver
// (tests/host_asan_main.cpp) can run them with an `operator new` that fails on demand.  Each returns a vs_status and
// lets no exception out (VS_GUARD_BEGIN / VS_GUARD_END inside).
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

// The seed-1 shuffle of n rows (vs_seed1_permutation; varsens/saltelli.py:100-101), computed once per n and process:
// the last two n (up to 2^25 rows, 128 MiB each) are kept for the next context, slice or design.  Thread-safe.
int vs_seed1_host_cached(uint64_t n, std::shared_ptr<const std::vector<uint32_t>>* out);

// Sobol byte tables of a (k, 32) direction table: tabs[((c * 4 + b) * 256 + v)] = XOR of the direction integers of
// column c that the bits of byte value v select at byte position b (a point is the XOR of four look-ups per
// coordinate; quantlib/sobolGen.cpp:47-60 is the reference's counterpart of the generator).
int vs_sobol_byte_tables(int k, const uint32_t* directions, std::vector<uint32_t>* tabs);
