// Shared host-side helpers for liblwg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/lwg.h"

namespace lwg {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

inline hipStream_t as_stream(lwg_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define LWG_FAIL(code, ...)          \
    do {                             \
        ::lwg::set_error(__VA_ARGS__); \
        return (code);               \
    } while (0)

#define LWG_REQUIRE(cond, ...)                            \
    do {                                                  \
        if (!(cond)) LWG_FAIL(LWG_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)

#define LWG_HIP(call)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) LWG_FAIL(LWG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// after a kernel launch: surfaces launch-configuration errors (the reference only printf'd them)
#define LWG_LAUNCH_CHECK(name)                                                                        \
    do {                                                                                              \
        hipError_t e_ = hipGetLastError();                                                            \
        if (e_ != hipSuccess) LWG_FAIL(LWG_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace lwg
