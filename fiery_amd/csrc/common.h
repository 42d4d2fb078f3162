// Shared host-side plumbing for libfiery_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "fiery_hip.h"

namespace fiery {

constexpr int kWave = 64;   // CDNA wavefront width

// thread-local error text behind fiery_last_error()
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(FIERY_ELAUNCH, "%s: %s", what, hipGetErrorString(err));
    return FIERY_OK;
}

inline hipStream_t as_stream(fiery_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int ceil_div(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

template <typename T>
inline bool aligned16(const T* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace fiery

#define FIERY_REQUIRE(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) return ::fiery::fail(FIERY_EINVAL, __VA_ARGS__); \
    } while (0)
