// Micro-benchmark (round 3): cycles per v_mfma_f32_16x16x4_f32 (32-cycle pipe time) and per v_mfma_f32_32x32x2_f32
// (64) at 1, 2 and 3 waves per SIMD, with K filler v_pk_fma_f32 after every GAP-th MFMA.  Would a 3x3 Winograd kernel
// on 16 x 16 tiles (half the accumulators per wave -> two waves per SIMD) issue at pipe rate?
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_issue16.hip -o tools/exp/mfma_issue16.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int SHAPE, int K, int GAP>
__global__ __launch_bounds__(768) void kern(float* out, int iters) {
  floatx16 acc32[4];
  floatx4 acc16[8];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 4; ++r) acc16[i][r] = 0.f;
  f2 f[16];
  for (int i = 0; i < 16; ++i) f[i] = f2{threadIdx.x * 0.001f + i, 1.f};
  float a = threadIdx.x * 0.5f, b = 1.0f;
  const f2 c1 = {1.0001f, 1.0001f}, c2 = {0.5f, 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (SHAPE == 32) acc32[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32[m & 3], 0, 0, 0);
      else acc16[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc16[m], 0, 0, 0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (K && (m % GAP) == GAP - 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) f[k & 15] = __builtin_elementwise_fma(f[k & 15], c1, c2);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += f[i].x + f[i].y;
  for (int i = 0; i < 4; ++i) s += acc32[i][0];
  for (int i = 0; i < 8; ++i) s += acc16[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int K, int GAP>
void run(int threads) {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 4000;
  hipLaunchKernelGGL((kern<SHAPE, K, GAP>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kern<SHAPE, K, GAP>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const int wps = threads / 256;
  const double per_simd = (double)ms * 1e-3 / (iters * 8.0 * wps) * 2.4e9;  // cycles at 2.4 GHz per MFMA issued on a SIMD
  printf("%dx%d  %d wave(s)/SIMD  %2d v_pk_fma_f32 after every %d: %.1f cycles per MFMA (pipe time %d)\n", SHAPE, SHAPE, wps,
         K, GAP, per_simd, SHAPE == 32 ? 64 : 32);
  hipFree(out);
}

int main() {
  for (int t : {256, 512, 768}) {
    run<32, 0, 1>(t);
    run<16, 0, 1>(t);
    run<32, 16, 8>(t);
    run<16, 16, 8>(t);
    run<16, 8, 8>(t);
  }
  return 0;
}
