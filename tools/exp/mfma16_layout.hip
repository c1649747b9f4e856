// Checks the operand / result layout of v_mfma_f32_16x16x4_f32 assumed by the 16 x 16-tile Winograd kernel:
//   A (16 x 4):  lane l holds A[l % 16][l / 16];   B (4 x 16): lane l holds B[l / 16][l % 16];
//   D (16 x 16): lane l, register v holds D[4 * (l / 16) + v][l % 16].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {  // A[16][4], B[4][16], D[16][16] row-major
  const int l = threadIdx.x;
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[(4 * (l / 16) + v) * 16 + l % 16] = acc[v];
}
int main() {
  float hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = (float)((i * 7) % 11) - 5.f; hB[i] = (float)((i * 5) % 13) - 6.f; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += std::fabs(hD[i] - ref[i]) > 1e-4f;
  printf("v_mfma_f32_16x16x4_f32 layout check: %d mismatches of 256\n", bad);
  return bad != 0;
}
