// Micro-benchmark: do a wave's own VALU / LDS instructions overlap with its MFMAs on gfx950, or only
// another wave's?  N iterations of 8 independent v_mfma_f32_32x32x2_f32 with K filler v_fma_f32 after each,
// at 1 and 2 waves per SIMD.  Prints cycles per MFMA (64 = pipe-bound).
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_issue.hip -o tools/exp/mfma_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int K, int LDSR>
__global__ __launch_bounds__(512) void kern(float* out, int iters, long long* cyc) {
  __shared__ float4 lds[1024];
  floatx16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float f[16];
  for (int i = 0; i < 16; ++i) f[i] = threadIdx.x * 0.001f + i;
  float a = threadIdx.x * 0.5f, b = 1.0f;
  lds[threadIdx.x] = make_float4(a, b, a, b);
  __syncthreads();
  float4 l = make_float4(0, 0, 0, 0);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K; ++k) f[(m * K + k) & 15] = __builtin_fmaf(f[(m * K + k) & 15], 1.0001f, 0.5f);
      if (LDSR && (m & 1) == 0) {
        const float4 t = lds[(threadIdx.x + m * 64 + it) & 1023];
        l.x += t.x;
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = l.x;
  for (int i = 0; i < 16; ++i) s += f[i];
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef float f2 __attribute__((ext_vector_type(2)));
// K VALU instructions (PK: v_pk_fma_f32, else v_fma_f32) after every GAP-th MFMA
template <int K, int GAP, int PK>
__global__ __launch_bounds__(512) void kern2(float* out, int iters) {
  floatx16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f2 f[16];
  for (int i = 0; i < 16; ++i) f[i] = f2{threadIdx.x * 0.001f + i, 1.f};
  float a = threadIdx.x * 0.5f, b = 1.0f;
  const f2 c1 = {1.0001f, 1.0001f}, c2 = {0.5f, 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if ((m % GAP) == GAP - 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if (PK) f[k & 15] = __builtin_elementwise_fma(f[k & 15], c1, c2);
          else f[k & 15].x = __builtin_fmaf(f[k & 15].x, 1.0001f, 0.5f);
        }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += f[i].x + f[i].y;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K, int GAP, int PK>
void run2() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 2000;
  hipLaunchKernelGGL((kern2<K, GAP, PK>), dim3(256), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kern2<K, GAP, PK>), dim3(256), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per = (double)ms * 1e-3 / (iters * 8.0) * 2.4e9;
  printf("1 wave/SIMD: %2d %s after every %d. MFMA: %.1f cycles per MFMA -> %.1f cycles per VALU instruction\n", K,
         PK ? "v_pk_fma_f32" : "v_fma_f32   ", GAP, per, (per - 70.8) * GAP / K);
  hipFree(out);
}

template <int K, int LDSR>
void run(int threads, const char* tag) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4);
  hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  hipLaunchKernelGGL((kern<K, LDSR>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kern<K, LDSR>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  // s_memtime / readcyclecounter ticks at a fixed 100 MHz on this part, so use the wall time and assume ~2.3 GHz too
  const double per_mfma_wave = (double)ms * 1e-3 / (iters * 8.0);
  printf("%-28s K=%2d lds=%d waves/SIMD=%d : %.1f ns per MFMA per wave  (%.1f cycles at 2.4 GHz; per SIMD %.1f)\n", tag, K,
         LDSR, threads / 256, per_mfma_wave * 1e9, per_mfma_wave * 2.4e9, per_mfma_wave * 2.4e9 / (threads / 256));
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run2<4, 1, 0>();
  run2<4, 1, 1>();
  run2<8, 2, 0>();
  run2<8, 2, 1>();
  run2<16, 4, 0>();
  run2<16, 4, 1>();
  run2<16, 8, 0>();
  run2<16, 8, 1>();
  run2<32, 8, 0>();
  run2<32, 8, 1>();
  run<0, 0>(256, "1 wave/SIMD");
  run<4, 0>(256, "1 wave/SIMD");
  run<8, 0>(256, "1 wave/SIMD");
  run<12, 0>(256, "1 wave/SIMD");
  run<16, 0>(256, "1 wave/SIMD");
  run<0, 1>(256, "1 wave/SIMD");
  run<8, 1>(256, "1 wave/SIMD");
  run<0, 0>(512, "2 waves/SIMD");
  run<4, 0>(512, "2 waves/SIMD");
  run<8, 0>(512, "2 waves/SIMD");
  run<12, 0>(512, "2 waves/SIMD");
  run<16, 0>(512, "2 waves/SIMD");
  run<8, 1>(512, "2 waves/SIMD");
  return 0;
}
