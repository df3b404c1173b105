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

// Per-device one-time setup (hipFuncSetAttribute opt-ins are per device; a process may switch GPUs).
struct DeviceOnce {
    unsigned long long done_mask = 0;   // bit d: set up on device d (callers are single-threaded per handle)
    static int device()
    {
        int d = 0;
        return hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64 ? d : 0;
    }
    bool done() const { return (done_mask >> device()) & 1ull; }
    void mark() { done_mask |= 1ull << device(); }
};

// compute-unit count of the current device (cached per device)
inline int device_cu_count()
{
    static int cus[64] = {0};
    const int d = DeviceOnce::device();
    if (!cus[d]) {
        hipDeviceProp_t prop;
        cus[d] = hipGetDeviceProperties(&prop, d) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    return cus[d];
}

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace lwg
