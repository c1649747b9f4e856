// Micro-benchmark (round 4): what does the bf16 matrix pipe of an MI355X sustain under its power cap, as a function of the
// DATA it multiplies?  Back-to-back v_mfma_f32_32x32x16_bf16 on 8 independent accumulators, nothing else in the loop, 512 blocks
// of 256 threads (two waves per SIMD on every CU, the occupancy of conv_mfma_bf16), operands = zeros / one constant / random
// N(0, 1) values rounded to bf16 / random bit patterns.  The conv kernel's ablation builds (stale registers, unwritten outputs)
// multiply constant or zero data: their "gains" are partly clock, not schedule.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_bf16_power.hip -o tools/exp/mfma_bf16_power.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <unistd.h>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void kern(const floatx4* __restrict__ ops, const floatx4* __restrict__ opsb, float* out, int iters, int nsets) {
  floatx16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // 8 operand registers per lane, rotated so that A and B differ from MFMA to MFMA
  floatx4 v[8];
  floatx4 w[8];
  for (int i = 0; i < 8; ++i) v[i] = ops[((blockIdx.x * 8 + i) % nsets) * 256 + threadIdx.x];
  for (int i = 0; i < 8; ++i) w[i] = opsb[((blockIdx.x * 8 + i) % nsets) * 256 + threadIdx.x];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m)
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v[m]), __builtin_bit_cast(bf16x8, w[(m + 3) & 7]),
                                                       acc[m], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static unsigned short bf16_of(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}

int main(int argc, char** argv) {
  const int nsets = 64, n16 = nsets * 256 * 8;  // bf16 values
  std::vector<unsigned short> h(n16);
  floatx4 *d, *db;
  float* out;
  hipMalloc(&d, n16 * 2);
  hipMalloc(&db, n16 * 2);
  std::vector<unsigned short> hb(n16);
  hipMalloc(&out, 512 * 256 * 4);
  const char* names[] = {"zeros", "constant 1.0", "random N(0,1) bf16", "random N(0,1) * relu (half zeros)", "random bit patterns (finite)",
                         "A = relu(N(0,1)), B = N(0,1) (a conv layer's operands)",
                         "A = N(0,1), B = relu(N(0,1)) (the same, roles swapped)"};
  for (int mode : {5, 6, 5, 6, 2, 0, 3, 4, 1}) {
    srand(1);
    for (int i = 0; i < n16; ++i) {
      const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
      const float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
      unsigned short b = 0;
      if (mode == 1) b = bf16_of(1.0f);
      if (mode == 2) b = bf16_of(g);
      if (mode == 3) b = bf16_of(g > 0 ? g : 0.f);
      if (mode == 4) b = (unsigned short)((rand() & 0xffff) & ~0x4000);  // exponent < 2^1: finite, no overflow in fp32 sums
      if (mode >= 5) b = bf16_of(g > 0 ? g : 0.f);
      const unsigned short nb = bf16_of(sqrtf(-2.f * logf((rand() + 1.0f) / (RAND_MAX + 2.0f))) * cosf(6.2831853f * (rand() / (float)RAND_MAX)));
      h[i] = mode == 6 ? nb : b;
      hb[i] = mode == 5 ? nb : b;
    }
    hipMemcpy(d, h.data(), n16 * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), n16 * 2, hipMemcpyHostToDevice);
    // a cold start (2 s idle before) of ~10 ms, then ~100 ms, then ~1 s back to back: the power controller needs time to react
    usleep(2000000);
    printf("%-56s", names[mode]);
    for (int iters : {40000, 400000, 4000000}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(512), dim3(256), 0, 0, d, db, out, iters, nsets);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = 512.0 * 4 * iters * 8.0 * 2 * 32 * 32 * 16;
      printf("  %7.1f ms: %6.0f TFLOP/s (%.2f)", ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0);
    }
    printf("\n");
  }
  return 0;
}
