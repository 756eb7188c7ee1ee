// Error store and the guard every entry point of the C ABI runs under; no HIP header, so the host-only translation
// unit (vs_host.cpp) and the sanitizer driver (tests/host_asan_main.cpp) see the same macros as the HIP sources.
//
// include/varsens_hip.h promises "no C++ exception crosses the boundary".  The library is C++ inside (std::vector,
// std::string, new, std::thread): every `extern "C"` body sits between VS_GUARD_BEGIN and VS_GUARD_END, which turn
// whatever escapes into a status code and a message -- std::bad_alloc into VS_ERR_NOMEM, anything else into
// VS_ERR_STATE.  vs_fail formats into a thread-local array and allocates nothing.
// (tests/test_abi_and_host.py::test_every_entry_point_is_guarded reads the sources and holds them to it;
// tests/host_asan_main.cpp drives the host-only entry points with an `operator new` that fails on demand.)
#pragma once

#include <exception>
#include <new>

#include "varsens_hip.h"

int vs_fail(int code, const char* fmt, ...);

#define VS_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) return vs_fail(VS_ERR_ARG, __VA_ARGS__);  \
    } while (0)

#define VS_GUARD_BEGIN try {
#define VS_GUARD_END                                                                                      \
    }                                                                                                     \
    catch (const std::bad_alloc&) {                                                                       \
        return vs_fail(VS_ERR_NOMEM, "%s: out of host memory", __func__);                                 \
    }                                                                                                     \
    catch (const std::exception& e__) {                                                                   \
        return vs_fail(VS_ERR_STATE, "%s: unexpected C++ exception: %s", __func__, e__.what());           \
    }                                                                                                     \
    catch (...) {                                                                                         \
        return vs_fail(VS_ERR_STATE, "%s: unexpected C++ exception", __func__);                           \
    }
