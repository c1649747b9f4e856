// Helpers shared by the two Winograd kernels (conv_wino.hip, conv_wino7.hip): packed-pair arithmetic for the
// input transforms and raw buffer loads for every global operand of the multiply loops.
#pragma once
#include <hip/hip_runtime.h>

namespace rtpose {
namespace winoc {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

// 16 bytes as two packed pairs: the transform below is written on float2 so that it compiles to
// v_pk_fma_f32 / v_pk_add_f32 (fp32 VALU instructions take their cycles from the SAME ALUs the fp32 MFMAs
// run on - tools/exp/mfma_issue.hip: 64 -> 101 cycles per MFMA with 4 v_fma_f32 after each, at one or two
// waves per SIMD alike - so every VALU instruction in the multiply loop is paid for in matrix throughput).
struct F4 {
  f2 lo, hi;
};
__device__ __forceinline__ F4 fma4(float s, F4 a, F4 b) {  // s * a + b, one rounding
  const f2 ss = {s, s};
  return F4{__builtin_elementwise_fma(ss, a.lo, b.lo), __builtin_elementwise_fma(ss, a.hi, b.hi)};
}
__device__ __forceinline__ F4 mul4(float s, F4 a) {
  const f2 ss = {s, s};
  return F4{ss * a.lo, ss * a.hi};
}
__device__ __forceinline__ F4 add4(F4 a, F4 b) { return F4{a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ F4 sub4(F4 a, F4 b) { return F4{a.lo - b.lo, a.hi - b.hi}; }
__device__ __forceinline__ float4 to_float4(F4 a) { return make_float4(a.lo.x, a.lo.y, a.hi.x, a.hi.y); }

// Raw buffer loads: address = base (4 SGPRs) + per-lane byte offset (1 VGPR, fixed for the whole kernel) + uniform
// byte offset (1 SGPR, advanced by the scalar unit): no vector instruction is spent on address arithmetic.
// (Bound to the LLVM intrinsic by name: this compiler lowers __builtin_amdgcn_raw_buffer_load_b128 to a
// ONE-dword load.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 llvm_raw_buffer_load_v4f32(i32x4 rsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.v4f32");
// `bytes` = what is addressable from p (the rest of the tensor / packed filter the pointer lies in): the hardware
// clamps every access against it - an over-read returns zeros and an over-write is dropped instead of touching a
// neighbouring allocation (a weight prefetch that ran 3 ky steps past the packed filters once reached a committed
// kernel with the clamp off).  Descriptors are based at a block's own first element, so the 32-bit range only
// saturates for tensors beyond 2 GiB from there, which the per-lane offsets cannot reach anyway.
constexpr unsigned kMaxRange = 0x7ffffffeu;
__device__ __forceinline__ i32x4 make_rsrc(const void* p, size_t bytes) {
  union {
    struct {
      const void* p;
      unsigned range, cfg;
    } s;
    i32x4 v;
  } u;
  u.s.p = p;
  u.s.range = bytes < (size_t)kMaxRange ? (unsigned)bytes : kMaxRange;  // bytes addressable from p
  u.s.cfg = 0x00020000;                                                 // raw buffer, 32-bit data format
  return u.v;
}
__device__ void llvm_raw_buffer_store_f32(float v, i32x4 rsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.store.f32");
// one dword per lane; a lane whose voff is >= the descriptor's range (kNoStore > kMaxRange) stores nothing: masked
// stores without a branch around them
constexpr unsigned kNoStore = 0x7fffffffu;
__device__ __forceinline__ void bstore(float v, i32x4 r, unsigned voff, unsigned soff) {
  llvm_raw_buffer_store_f32(v, r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ F4 bload(i32x4 r, unsigned voff, unsigned soff) {
  const f32x4 v = llvm_raw_buffer_load_v4f32(r, (int)voff, (int)soff, 0);
  return F4{f2{v.x, v.y}, f2{v.z, v.w}};
}
__device__ __forceinline__ float4 bload_f4(i32x4 r, unsigned voff, unsigned soff) {
  const f32x4 v = llvm_raw_buffer_load_v4f32(r, (int)voff, (int)soff, 0);
  return make_float4(v.x, v.y, v.z, v.w);
}

}  // namespace winoc
}  // namespace rtpose
