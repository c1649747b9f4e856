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

// ---- per-device host state ---------------------------------------------------------------
// One process may drive several GPUs (the reference wraps its model in nn.DataParallel,
// demo/picture_demo.py:47), so nothing learnt about "the" device is cached per process:
// kernel attributes and CU counts are kept per HIP device ordinal.
constexpr int kMaxDevices = 64;

inline int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
  return d;
}

// HIP device that owns device memory `p`, or -1 (host / unregistered memory).  One look-up in the runtime's
// allocation map (~1 us): done where a pointer is handed over (bind) and once per NEW pointer on the hot path.
inline int pointer_device(const void* p) {
  hipPointerAttribute_t a;
  if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) ? a.device : -1;
}
// A launch goes to the CURRENT device with pointers that are only valid on the device that owns them: one
// process may hold several GPUs (nn.DataParallel replicas, demo/picture_demo.py:47; a mis-set LOCAL_RANK on an
// 8-GPU node), so every entry point that takes device memory refuses memory of another device - or of none -
// instead of launching.  Returns 0 or fail(...).
inline int check_device_ptr(const void* p, int dev, const char* who, const char* what) {
  const int owner = pointer_device(p);
  if (owner == dev) return 0;
  if (owner < 0)
    return fail(RTPOSE_E_INVAL, "%s: %s (%p) is not device memory of a HIP device (host pointer? freed?)", who, what, p);
  return fail(RTPOSE_E_INVAL, "%s: %s lives on HIP device %d but the current device is %d (hipSetDevice / "
                              "torch.cuda.device before the call, or bind the plan on the device that runs it)",
              who, what, owner, dev);
}
// the last pointer an entry point verified for one of its arguments (per plan, or thread_local for the free
// functions): the look-up is repeated only when the caller passes another pointer
struct CheckedPtr {
  const void* p = nullptr;
  int dev = -1;
  int check(const void* q, int cur, const char* who, const char* what) {
    if (q == p && dev == cur) return 0;
    const int rc = check_device_ptr(q, cur, who, what);
    if (!rc) {
      p = q;
      dev = cur;
    }
    return rc;
  }
};

// "done once per device" flag set (e.g. hipFuncSetAttribute of one kernel instantiation);
// benign if two threads race: the attribute is simply set twice.
struct PerDeviceOnce {
  volatile bool done[kMaxDevices];
  bool is_set(int dev) const { return done[dev]; }
  void set(int dev) { done[dev] = true; }
};

inline int device_cu_count() {
  static volatile int n_cu[kMaxDevices];
  const int dev = current_device();
  int n = n_cu[dev];
  if (!n) {
    hipDeviceProp_t prop;
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0;
    if (n <= 0) n = 256;
    n_cu[dev] = n;
  }
  return n;
}

// Division by a launch-time constant without the ~40-instruction integer division sequence: for 0 <= m < 2^31 and d >= 2,
// floor(m / d) == (m * mul) >> (32 + shift) with l = ceil(log2 d), mul = ceil(2^(31 + l) / d) < 2^32, shift = l - 1
// (error term m e / (d 2^(31 + l)) < 1 / d because m < 2^31 and e < d <= 2^l).  shift < 0 encodes d == 1.
struct FastDiv {
  unsigned mul;
  int shift;
};
inline FastDiv make_fastdiv(int d) {
  FastDiv f;
  if (d <= 1) {
    f.mul = 0;
    f.shift = -1;
    return f;
  }
  int l = 0;
  while ((1ll << l) < d) ++l;
  f.mul = (unsigned)(((1ull << (31 + l)) + (unsigned long long)d - 1) / (unsigned long long)d);
  f.shift = l - 1;
  return f;
}
#ifdef __HIPCC__
__device__ __forceinline__ int fast_div(int m, const FastDiv f) {
  return f.shift < 0 ? m : (int)(__umulhi((unsigned)m, f.mul) >> f.shift);
}
#endif

// channel-plane slices (rtpose_conv_desc.in_plane_pixels / out_plane_pixels) exist for the F(4x4,3x3) kernel only
inline bool desc_has_planes(const rtpose_conv_desc* d, int ngroups) {
  for (int g = 0; d && g < ngroups && g < 2; ++g)
    if (d[g].in_plane_pixels || d[g].out_plane_pixels) return true;
  return false;
}
#define RTPOSE_REFUSE_PLANES(d, ngroups, who)                                                                        \
  if (::rtpose::desc_has_planes(d, ngroups))                                                                         \
  return ::rtpose::fail(RTPOSE_E_INVAL, who ": channel-plane slices (in_plane_pixels / out_plane_pixels) are read and " \
                                            "written by F(4x4,3x3) launches only (zero-initialise descriptors)")

// Output channels are padded to the conv kernel's N tile.
constexpr int kConvBN = 64;
inline int cout_pad(int cout) { return ceil_div(cout, kConvBN) * kConvBN; }

}  // namespace rtpose
