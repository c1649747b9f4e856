// Round 6: the candidate-scoring loop of limb_assign_kernel (csrc/decode.hip, round-5 form) on its own, compiled WITH clang's
// SLP vectoriser (default flags: the x / y halves of the sample coordinates and of the dot product become v_pk_mul_f32 /
// v_pk_add_f32) or WITHOUT (-fno-slp-vectorize -fno-vectorize), driven by tools/exp/pk_beside_forward.py beside the library's
// forward.  304 blocks of 256 threads; the first nA * nB threads of a block score one candidate pair each: two peaks, unit
// vector by fp32 division, ten samples, two map loads per sample.  Every launch's output is compared with the same launch run
// alone.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC tools/exp/limb_victim.hip -o tools/exp/limb_victim_slp.so
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -shared -fPIC tools/exp/limb_victim.hip -o tools/exp/limb_victim_noslp.so
#include <hip/hip_runtime.h>

struct Peak {
  int x, y;
  float score;
  int id;
};

__global__ __launch_bounds__(256) void limb_victim(const Peak* __restrict__ peaks, const float* __restrict__ map, int h, int w,
                                                   int cstride, int nA, int nB, int up_shift, float* __restrict__ out) {
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
    if (norm > 0.f) {
      vx = vx / norm;
      vy = vy / norm;
      const float step_x = (float)(B.x - A.x) / 10.f;
      const float step_y = (float)(B.y - A.y) / 10.f;
      float scores = 0.f;
      int crit1 = 0;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const float fx = (float)A.x + (float)i * step_x, fy = (float)A.y + (float)i * step_y;
        int lx = (int)fx, ly = (int)fy;
        if (fx - (float)lx >= 0.5f) ++lx;
        if (fy - (float)ly >= 0.5f) ++ly;
        int sx = lx >> up_shift, sy = ly >> up_shift;
        sx = min(max(sx, 0), w - 1);
        sy = min(max(sy, 0), h - 1);
        const size_t q = ((size_t)sy * w + sx) * cstride;
        const float px = map[q + chx];
        const float py = map[q + chy];
        const float s = vx * px + vy * py;
        scores = scores + s;
        if (s > 0.05f) ++crit1;
      }
      cand = scores / 10.f + (float)crit1;  // (every candidate reports, whatever its criteria)
    }
    out[(size_t)blockIdx.x * 64 + p] = cand;
  }
}

extern "C" int limb_victim_launch(int blocks, const void* peaks, const void* map, int h, int w, int cstride, int nA, int nB,
                                  void* out, void* stream) {
  hipLaunchKernelGGL(limb_victim, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     static_cast<const Peak*>(peaks), static_cast<const float*>(map), h, w, cstride, nA, nB, 3,
                     static_cast<float*>(out));
  return (int)hipGetLastError();
}
