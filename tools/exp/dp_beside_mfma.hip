// Reproducer attempt (round 5) for the open finding of DESIGN.md 3.3: the pose decoder's limb-scoring wave, run on a second
// stream BESIDE the bf16 plan's MFMA kernels, returned in ~1 % of the batches a score whose line integral took one sample a
// position off - in lanes 41..63 only, with bit-identical inputs.  Here the scoring loop stands alone ("victim": 304 blocks of
// 256 threads, the first `npairs` threads of each score one candidate pair as limb_assign_kernel does - two peaks, unit vector by
// fp32 division, ten samples whose coordinates go through the reference's double-precision rounding chain, two map loads per
// sample) and runs many times while an "aggressor" occupies the device on another stream:
//   bf16   back-to-back v_mfma_f32_32x32x16_bf16 (the loop of tools/exp/mfma_bf16_power.hip, N(0,1) x relu(N(0,1)) operands)
//   f32    back-to-back v_mfma_f32_32x32x2_f32
//   copy   a streaming copy (no matrix instruction)
//   none   nothing (control)
// Victim variants: dp (the r4 form: coordinates through v_cvt_f64 / v_add_f64 / v_floor_f64), int (round 5's exact integer
// form), dp-noload (the dp chain, the map value computed from the coordinates instead of loaded).  Every launch is compared on
// the device with the result of the same launch run alone; mismatches are counted per lane.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/exp/dp_beside_mfma.hip -o tools/exp/dp_beside_mfma.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

// ---- aggressors -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void mfma_bf16_loop(const floatx4* __restrict__ a, const floatx4* __restrict__ b, float* out,
                                                         int iters) {
  floatx16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  floatx4 v[8], w[8];
  for (int i = 0; i < 8; ++i) v[i] = a[((blockIdx.x * 8 + i) % 64) * 256 + threadIdx.x];
  for (int i = 0; i < 8; ++i) w[i] = b[((blockIdx.x * 8 + i) % 64) * 256 + threadIdx.x];
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

__global__ __launch_bounds__(256, 2) void mfma_f32_loop(const float* __restrict__ a, const float* __restrict__ b, float* out, int iters) {
  floatx16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[8], w[8];
  for (int i = 0; i < 8; ++i) v[i] = a[((blockIdx.x * 8 + i) % 64) * 256 + threadIdx.x];
  for (int i = 0; i < 8; ++i) w[i] = b[((blockIdx.x * 8 + i) % 64) * 256 + threadIdx.x];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[m], w[(m + 3) & 7], acc[m], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy_loop(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, int passes) {
  for (int p = 0; p < passes; ++p)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// ---- victim: the candidate-scoring loop of limb_assign_kernel (csrc/decode.hip) -----------------------------------------------
struct Peak {
  int x, y;
  float score;
  int id;
};

template <int VARIANT>  // 0 dp, 1 int, 2 dp-noload
__global__ __launch_bounds__(256) void victim(const Peak* __restrict__ peaks, const float* __restrict__ map, int h, int w, int cstride,
                                              int nA, int nB, double inv_up, int h1, float* __restrict__ out) {
  const int tid = threadIdx.x;
  const Peak* pA = peaks + (size_t)blockIdx.x * 16;
  const Peak* pB = pA + 8;
  const int npairs = nA * nB;
  const int chx = (blockIdx.x % 19) * 2, chy = chx + 1;
  for (int p = tid; p < npairs; p += 256) {
    const int a = p / nB, b = p - a * nB;
    const Peak A = pA[a], B = pB[b];
    float cand = 0.f;
    float vx = (float)(B.x - A.x), vy = (float)(B.y - A.y);
    const float norm = sqrtf(vx * vx + vy * vy);
    if (!((double)norm < 1e-12)) {
      vx = vx / norm;
      vy = vy / norm;
      const float step_x = (float)(B.x - A.x) / 10.f;
      const float step_y = (float)(B.y - A.y) / 10.f;
      float scores = 0.f;
      int crit1 = 0;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        int lx, ly, sx, sy;
        if (VARIANT == 1) {
          const float fx = (float)A.x + (float)i * step_x, fy = (float)A.y + (float)i * step_y;
          lx = (int)fx;
          ly = (int)fy;
          if (fx - (float)lx >= 0.5f) ++lx;
          if (fy - (float)ly >= 0.5f) ++ly;
          sx = lx >> 3;
          sy = ly >> 3;
        } else {
          lx = (int)((double)((float)A.x + (float)i * step_x) + 0.5);
          ly = (int)((double)((float)A.y + (float)i * step_y) + 0.5);
          sx = (int)floor((double)lx * inv_up);
          sy = (int)floor((double)ly * inv_up);
        }
        sx = min(max(sx, 0), w - 1);
        sy = min(max(sy, 0), h - 1);
        float px, py;
        if (VARIANT == 2) {  // a value that changes a lot from one map pixel to the next
          px = (float)((sx * 37 + sy * 101) & 63) * (1.f / 64.f);
          py = (float)((sx * 53 + sy * 29) & 63) * (1.f / 64.f);
        } else {
          const size_t q = ((size_t)sy * w + sx) * cstride;
          px = map[q + chx];
          py = map[q + chy];
        }
        const float s = vx * px + vy * py;
        scores = scores + s;
        if (s > 0.05f) ++crit1;
      }
      const double pen = fmin(0.0, 0.5 * (double)h1 / (double)norm - 1.0);
      const float crit2 = (float)((double)(scores / 10.f) + pen);
      cand = crit2 + (float)crit1;  // (every candidate reports, whatever its criteria)
    }
    out[(size_t)blockIdx.x * 64 + p] = cand;
  }
}

__global__ void compare(const float* __restrict__ got, const float* __restrict__ want, int n, unsigned* __restrict__ hist) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && __float_as_uint(got[i]) != __float_as_uint(want[i])) atomicAdd(&hist[i & 63], 1u);
}

static unsigned short bf16_of(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
static float gauss() {
  const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 600;
  const int NB = 304, H = 46, W = 46, CS = 57, H1 = 368;
  srand(7);
  // operands of the aggressors
  const int n16 = 64 * 256 * 8;
  std::vector<unsigned short> ha(n16), hb(n16);
  for (int i = 0; i < n16; ++i) {
    const float g = gauss();
    ha[i] = bf16_of(g > 0 ? g : 0.f);
    hb[i] = bf16_of(gauss());
  }
  std::vector<float> fa(64 * 256), fb(64 * 256);
  for (auto& v : fa) v = fmaxf(gauss(), 0.f);
  for (auto& v : fb) v = gauss();
  floatx4 *da, *db;
  float *dfa, *dfb, *dout, *csrc, *cdst;
  CHECK(hipMalloc(&da, n16 * 2));
  CHECK(hipMalloc(&db, n16 * 2));
  CHECK(hipMalloc(&dfa, fa.size() * 4));
  CHECK(hipMalloc(&dfb, fb.size() * 4));
  CHECK(hipMalloc(&dout, 512 * 256 * 4));
  const size_t copy_floats = (size_t)64 << 20;  // 256 MB each way
  CHECK(hipMalloc(&csrc, copy_floats * 4));
  CHECK(hipMalloc(&cdst, copy_floats * 4));
  CHECK(hipMemset(csrc, 1, copy_floats * 4));
  CHECK(hipMemcpy(da, ha.data(), n16 * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, hb.data(), n16 * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dfa, fa.data(), fa.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dfb, fb.data(), fb.size() * 4, hipMemcpyHostToDevice));
  // the victim's inputs: per block two groups of 8 peaks on a 368 x 368 image, a map with a different value in every pixel
  std::vector<Peak> hp((size_t)NB * 16);
  for (auto& p : hp) p = Peak{rand() % 368, rand() % 368, 1.f, 0};
  std::vector<float> hm((size_t)H * W * CS);
  for (auto& v : hm) v = (float)(rand() % 4096) / 4096.f;
  Peak* dp;
  float *dm, *dgot, *dwant;
  unsigned* dhist;
  CHECK(hipMalloc(&dp, hp.size() * sizeof(Peak)));
  CHECK(hipMalloc(&dm, hm.size() * 4));
  CHECK(hipMalloc(&dgot, NB * 64 * 4));
  CHECK(hipMalloc(&dwant, NB * 64 * 4));
  CHECK(hipMalloc(&dhist, 64 * 4));
  CHECK(hipMemcpy(dp, hp.data(), hp.size() * sizeof(Peak), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dm, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev;
  CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));

  auto run_victim = [&](int variant, int nA, int nB, float* out, hipStream_t s) {
    if (variant == 0) hipLaunchKernelGGL(victim<0>, dim3(NB), dim3(256), 0, s, dp, dm, H, W, CS, nA, nB, 0.125, H1, out);
    if (variant == 1) hipLaunchKernelGGL(victim<1>, dim3(NB), dim3(256), 0, s, dp, dm, H, W, CS, nA, nB, 0.125, H1, out);
    if (variant == 2) hipLaunchKernelGGL(victim<2>, dim3(NB), dim3(256), 0, s, dp, dm, H, W, CS, nA, nB, 0.125, H1, out);
  };
  const char* agg_names[] = {"none", "bf16 MFMA", "f32 MFMA", "copy"};
  const char* var_names[] = {"dp", "int", "dp-noload"};
  printf("%d victim launches of %d blocks per cell; mismatching (launch, block, lane) results, then the lanes they fall on\n", launches, NB);
  for (int agg : {1, 0, 2, 3, 1}) {
    for (int variant = 0; variant < 3; ++variant) {
      for (int geom = 0; geom < 2; ++geom) {
        const int nA = geom ? 8 : 7, nB = geom ? 8 : 7;
        CHECK(hipMemset(dwant, 0, NB * 64 * 4));
        CHECK(hipMemset(dgot, 0, NB * 64 * 4));
        run_victim(variant, nA, nB, dwant, s2);  // alone: the reference
        CHECK(hipStreamSynchronize(s2));
        CHECK(hipMemset(dhist, 0, 64 * 4));
        int done = 0;
        while (done < launches) {
          // one aggressor launch of ~40 ms, the victim launches beside it
          if (agg == 1) hipLaunchKernelGGL(mfma_bf16_loop, dim3(512), dim3(256), 0, s1, da, db, dout, 160000);
          if (agg == 2) hipLaunchKernelGGL(mfma_f32_loop, dim3(512), dim3(256), 0, s1, dfa, dfb, dout, 80000);
          if (agg == 3)
            hipLaunchKernelGGL(copy_loop, dim3(2048), dim3(256), 0, s1, reinterpret_cast<const float4*>(csrc),
                               reinterpret_cast<float4*>(cdst), copy_floats / 4, 64);
          CHECK(hipEventRecord(ev, s1));
          int k = 0;
          do {
            run_victim(variant, nA, nB, dgot, s2);
            hipLaunchKernelGGL(compare, dim3((NB * 64 + 255) / 256), dim3(256), 0, s2, dgot, dwant, NB * 64, dhist);
            ++k;
            ++done;
            if ((k & 15) == 0) CHECK(hipStreamSynchronize(s2));  // (keep the host from running far ahead of the aggressor)
          } while (done < launches && (agg == 0 ? k < 64 : hipEventQuery(ev) == hipErrorNotReady));
          CHECK(hipStreamSynchronize(s1));
          CHECK(hipStreamSynchronize(s2));
        }
        unsigned hist[64];
        CHECK(hipMemcpy(hist, dhist, sizeof(hist), hipMemcpyDeviceToHost));
        unsigned total = 0;
        for (unsigned v : hist) total += v;
        printf("aggressor %-9s victim %-9s pairs %2d: %6u mismatches", agg_names[agg], var_names[variant], nA * nB, total);
        if (total) {
          printf("  lanes:");
          for (int l = 0; l < 64; ++l)
            if (hist[l]) printf(" %d:%u", l, hist[l]);
        }
        printf("\n");
        fflush(stdout);
      }
    }
  }
  return 0;
}
