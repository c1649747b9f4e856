// Round 6 reproducer for the finding of DESIGN.md 3.3: a wave that executes PACKED fp32 VALU instructions (v_pk_mul_f32 /
// v_pk_add_f32 - what clang's SLP vectoriser made of the x / y halves of limb_assign_kernel's sample coordinates) returns wrong
// values in lanes 48..63 when it shares a CU with the bf16 plan's kernels.  The victim here is pure ALU and checks itself:
// every lane computes the ten sample coordinates  c_i = A + i * step  (x and y) twice - once with packed instructions
// (inline asm: the instruction forms are fixed), once with v_mul_f32 / v_add_f32 - and counts, per lane, the iterations in which
// the two disagree bit for bit.  Alone the count is zero by construction (same IEEE operations).
//   variant 0: v_pk_mul_f32 -> v_pk_add_f32 -> v_cvt_i32_f32 back to back (the compiler's own sequence, no wait states)
//   variant 1: the same with s_nop 3 behind every packed instruction
//   variant 2: packed results consumed only after ~20 other instructions (no forwarding involved)
//   variant 3: control - the "packed" side also uses scalar instructions
//   variant 4: the multiplier from an SGPR pair with op_sel_hi:[1,0] (odd register = junk), as the compiler encodes (float)i
//   variant 5: neg_lo / neg_hi and the op_sel-swizzled v_pk_add_f32 of the compiler's dot product
// Built without the SLP vectoriser so that the reference side stays scalar:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -shared -fPIC tools/exp/pk_victim.hip -o tools/exp/pk_victim.so
// Driven by tools/exp/pk_beside_forward.py (the aggressor is the library's own forward on another stream).
#include <hip/hip_runtime.h>

typedef float floatx2 __attribute__((ext_vector_type(2)));

template <int V>
__device__ __forceinline__ void coords_pk(floatx2 A, floatx2 step, float fi, int& ix, int& iy) {
  floatx2 t, c;
  const floatx2 ii = {fi, fi};
  if (V == 0) {
    asm volatile(
        "v_pk_mul_f32 %0, %2, %3\n"
        "v_pk_add_f32 %1, %0, %4\n"
        : "=&v"(t), "=&v"(c)
        : "v"(step), "v"(ii), "v"(A));
    asm volatile("v_cvt_i32_f32 %0, %2\nv_cvt_i32_f32 %1, %3" : "=&v"(ix), "=&v"(iy) : "v"(c.x), "v"(c.y));
  } else if (V == 1) {
    asm volatile(
        "v_pk_mul_f32 %0, %2, %3\n"
        "s_nop 3\n"
        "v_pk_add_f32 %1, %0, %4\n"
        "s_nop 3\n"
        : "=&v"(t), "=&v"(c)
        : "v"(step), "v"(ii), "v"(A));
    asm volatile("v_cvt_i32_f32 %0, %2\nv_cvt_i32_f32 %1, %3" : "=&v"(ix), "=&v"(iy) : "v"(c.x), "v"(c.y));
  } else if (V == 4) {  // the multiplier as an SGPR pair whose ODD register holds junk, low half broadcast (op_sel_hi:[1,0])
    const unsigned long long k = (49ull << 32) | __float_as_uint(fi);
    asm volatile(
        "v_pk_mul_f32 %0, %2, %3 op_sel_hi:[1,0]\n"
        "v_pk_add_f32 %1, %0, %4\n"
        : "=&v"(t), "=&v"(c)
        : "v"(step), "s"(k), "v"(A));
    asm volatile("v_cvt_i32_f32 %0, %2\nv_cvt_i32_f32 %1, %3" : "=&v"(ix), "=&v"(iy) : "v"(c.x), "v"(c.y));
  } else if (V == 5) {  // neg_lo / neg_hi modifiers and the swizzled add the compiler uses for the dot product
    floatx2 u;
    asm volatile(
        "v_pk_mul_f32 %0, %3, %4\n"
        "v_pk_add_f32 %1, %0, %5\n"
        "v_pk_add_f32 %2, %1, %0 neg_lo:[0,1] neg_hi:[0,1]\n"   // u = c - t = A (exact when c was exact ... compared only through c)
        : "=&v"(t), "=&v"(c), "=&v"(u)
        : "v"(step), "v"(ii), "v"(A));
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(u) : "v"(c), "v"(t));
    asm volatile("v_cvt_i32_f32 %0, %2\nv_cvt_i32_f32 %1, %3" : "=&v"(ix), "=&v"(iy) : "v"(c.x), "v"(c.y));
    ix += (int)(u.x != c.y + t.x);   // the swizzled sum: lo = c.hi + t.lo, hi = c.lo + t.hi
    iy += (int)(u.y != c.x + t.y);
  } else {
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(t) : "v"(step), "v"(ii));
    asm volatile("s_nop 7\ns_nop 7\ns_nop 4");
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(c) : "v"(t), "v"(A));
    asm volatile("s_nop 7\ns_nop 7\ns_nop 4");
    asm volatile("v_cvt_i32_f32 %0, %2\nv_cvt_i32_f32 %1, %3" : "=&v"(ix), "=&v"(iy) : "v"(c.x), "v"(c.y));
  }
}

__device__ __forceinline__ void coords_scalar(floatx2 A, floatx2 step, float fi, int& ix, int& iy) {
  float tx, ty, cx, cy;
  asm volatile(
      "v_mul_f32 %0, %4, %6\n"
      "v_mul_f32 %1, %5, %6\n"
      "v_add_f32 %2, %0, %7\n"
      "v_add_f32 %3, %1, %8\n"
      : "=&v"(tx), "=&v"(ty), "=&v"(cx), "=&v"(cy)
      : "v"(step.x), "v"(step.y), "v"(fi), "v"(A.x), "v"(A.y));
  asm volatile("v_cvt_i32_f32 %0, %2\nv_cvt_i32_f32 %1, %3" : "=&v"(ix), "=&v"(iy) : "v"(cx), "v"(cy));
}

// grid (blocks), 256 threads; the first `active` threads of every block work (49 / 56 / 64 like the candidate pairs of a limb).
// in: one float4 (A.x, A.y, step.x, step.y) per (block, thread).  hist[64]: mismatching (lane, iteration) pairs per lane;
// detail: up to 64 records {block, lane, iteration, packed ix, iy, scalar ix, iy}.
template <int V>
__global__ __launch_bounds__(256) void pk_victim(const float4* __restrict__ in, int active, int rounds,
                                                 unsigned* __restrict__ hist, int* __restrict__ detail) {
  const int tid = threadIdx.x;
  if (tid >= active) return;
  const float4 v = in[(size_t)blockIdx.x * 256 + tid];
  const floatx2 A = {v.x, v.y}, step = {v.z, v.w};
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      int px, py, sx, sy;
      if (V == 3)
        coords_scalar(A, step, (float)i, px, py);
      else
        coords_pk<V>(A, step, (float)i, px, py);
      coords_scalar(A, step, (float)i, sx, sy);
      if (px != sx || py != sy) {
        atomicAdd(&hist[tid & 63], 1u);
        const unsigned k = atomicAdd(&hist[64], 1u);
        if (k < 64) {
          int* d = detail + 8 * k;
          d[0] = blockIdx.x;
          d[1] = tid;
          d[2] = i;
          d[3] = px;
          d[4] = py;
          d[5] = sx;
          d[6] = sy;
          d[7] = r;
        }
      }
    }
  }
}

extern "C" int pk_victim_launch(int variant, int blocks, int active, int rounds, const void* in, void* hist, void* detail,
                                void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const float4* p = static_cast<const float4*>(in);
  unsigned* h = static_cast<unsigned*>(hist);
  int* d = static_cast<int*>(detail);
  switch (variant) {
    case 0: hipLaunchKernelGGL(pk_victim<0>, dim3(blocks), dim3(256), 0, s, p, active, rounds, h, d); break;
    case 1: hipLaunchKernelGGL(pk_victim<1>, dim3(blocks), dim3(256), 0, s, p, active, rounds, h, d); break;
    case 2: hipLaunchKernelGGL(pk_victim<2>, dim3(blocks), dim3(256), 0, s, p, active, rounds, h, d); break;
    case 4: hipLaunchKernelGGL(pk_victim<4>, dim3(blocks), dim3(256), 0, s, p, active, rounds, h, d); break;
    case 5: hipLaunchKernelGGL(pk_victim<5>, dim3(blocks), dim3(256), 0, s, p, active, rounds, h, d); break;
    default: hipLaunchKernelGGL(pk_victim<3>, dim3(blocks), dim3(256), 0, s, p, active, rounds, h, d); break;
  }
  return (int)hipGetLastError();
}
