// Shared host-side helpers for librtpose_mi355x (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "rtpose_mi355x.h"

namespace rtpose {

// Thread-local last-error text behind rtpose_last_error().
char* err_buf();
int fail(int code, const char* fmt, ...);

#define RTPOSE_HIP_CHECK(expr)                                               \
  do {                                                                       \
    hipError_t e__ = (expr);                                                 \
    if (e__ != hipSuccess)                                                   \
      return ::rtpose::fail(RTPOSE_E_HIP(e__), "%s failed: %s (%s:%d)",      \
                            #expr, hipGetErrorString(e__), __FILE__,         \
                            __LINE__);                                       \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline size_t round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// Output channels are padded to the conv kernel's N tile.
constexpr int kConvBN = 64;
inline int cout_pad(int cout) { return ceil_div(cout, kConvBN) * kConvBN; }

}  // namespace rtpose
