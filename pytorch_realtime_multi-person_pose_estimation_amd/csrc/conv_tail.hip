// The two trailing 1x1 convolutions of a refinement-stage branch as ONE back-to-back GEMM launch, fp32, gfx950 (MI355X):
//   Mconv6_stageN_Lb = nn.Conv2d(128, 128, 1) + nn.ReLU,  Mconv7_stageN_Lb = nn.Conv2d(128, 38 | 19, 1)
// (lib/network/rtpose_vgg.py:120-127: the last two entries of every stage-2..6 block) and the stage-1 pair
//   conv5_4_CPM_Lb = nn.Conv2d(128, 512, 1) + nn.ReLU,  conv5_5_CPM_Lb = nn.Conv2d(512, 38 | 19, 1)   (:101-105),
// both branches of the stage in one grid.  As two launches of the generic kernel (conv_mfma.hip) these K = 128 GEMMs ran at 0.35 of the fp32 MFMA
// peak - per-block prologue / epilogue as long as the multiply loop - and the 128-channel intermediate made a
// 35 MB round trip per branch.  Here a block owns 64 pixels of one branch:
//   X [64 px x 128 ch] -> LDS -> GEMM 1 (4 waves x 32 columns, v_mfma_f32_32x32x2_f32) -> + bias, ReLU -> LDS (the A
//   operand layout again) -> GEMM 2 (2 x 2 waves of 32 px x 32 columns) -> + bias -> the stage's concat buffer.
// The intermediate never leaves the CU; 10 launches of a forward become 5.  Both GEMMs walk K in the order of the
// generic kernel (16-channel chunks, 4-channel groups, k pairs), so the sums - and the network's outputs - are the
// same bits as with the two separate launches.  Weights are the generic kernel's k = 1 packing [c / 4][cout_pad][4].
#include <hip/hip_runtime.h>

#include "common.h"

namespace rtpose {

namespace tail {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64;    // pixels per block
constexpr int KC = 128;   // input channels of both GEMMs
constexpr int N1 = 128;   // Mconv6 columns
constexpr int N2 = 64;    // Mconv7 columns (38 / 19 padded)
constexpr int PS = BM + 1;  // float4 per 4-channel plane in LDS (+1: the 8 lanes of a write group hit 8 bank groups)

struct Group {
  const float* in;
  const float* w1;
  const float* b1;
  const float* w2;
  const float* b2;
  float* out;
  int in_cstride, in_choff, in_ws, in_hs, in_lead;
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  int cout2;
};

struct Args {
  Group g[2];
  int N, H, W, M, mtiles, ngroups;
};

// NP = passes of 128 intermediate columns: 1 for Mconv6 / Mconv7 (128 -> 128 -> 38 | 19), 4 for the stage-1 pair
// conv5_4_CPM / conv5_5_CPM (128 -> 512 -> 38 | 19, rtpose_vgg.py:95-105): GEMM 1 produces 128 columns at a time, GEMM 2
// consumes them as the next 128 of its K = 512 - the accumulators of GEMM 2 live across the passes, k ascending.
#if defined(RTPOSE_EXP_TAIL_STAGGER) && !defined(RTPOSE_DEV_BUILD)
#error "RTPOSE_EXP_TAIL_STAGGER is a developer-build experiment (tools/build_dev.sh)"
#endif
#ifdef RTPOSE_EXP_TAIL_TIMELINE  // developer build: wall_clock64 stamps per block (tools/exp/tail_timeline.py)
#ifndef RTPOSE_DEV_BUILD
#error "RTPOSE_EXP_TAIL_TIMELINE is a developer-build experiment (tools/build_dev.sh)"
#endif
__device__ unsigned long long g_tail_tl[4096][8];
#define RTPOSE_TAIL_TL(i) \
  if (threadIdx.x == 0 && blockIdx.x < 4096) g_tail_tl[blockIdx.x][i] = wall_clock64()
#else
#define RTPOSE_TAIL_TL(i)
#endif

template <int NP>
__global__ __launch_bounds__(256, NP == 1 ? 3 : 2) void tail_kernel(const Args A) {
#ifdef RTPOSE_EXP_TAIL_STAGGER  // developer build: the k-th resident block of a CU (ids go round the CUs) starts k x this many 10 ns ticks late
  if (blockIdx.x < 1024) {
    const unsigned long long t0 = wall_clock64(), wait = (unsigned long long)(blockIdx.x >> 8) * (RTPOSE_EXP_TAIL_STAGGER);
    while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }
#endif
  RTPOSE_TAIL_TL(0);
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  float4* const X = lds4;
  // one pass: the intermediate takes the input tile's place (a barrier after GEMM 1); several: its own buffer
  float4* const T = NP == 1 ? lds4 : lds4 + (KC / 4) * PS;
  int* const qin = reinterpret_cast<int*>(lds4 + (NP == 1 ? 1 : 2) * (KC / 4) * PS);
  int* const qout = qin + BM;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int mt = blockIdx.x / A.ngroups, grp = blockIdx.x - mt * A.ngroups;
  const Group g = grp ? A.g[1] : A.g[0];
  constexpr int N1T = N1 * NP;  // columns of GEMM 1 = K of GEMM 2

  // ---- pixel -> element offsets of the block's 64 pixels (pixels past the end repeat the last one; not stored) ----
  if (tid < BM) {
    const int m = min(mt * BM + tid, A.M - 1);
    const int HW = A.H * A.W;
    const int n = m / HW, r = m - n * HW;
    const int y = r / A.W, x = r - y * A.W;
    qin[tid] = (g.in_lead + (n * g.in_hs + y) * g.in_ws + x) * g.in_cstride + g.in_choff;
    qout[tid] = (g.out_lead + (n * g.out_hs + y) * g.out_ws + x) * g.out_cstride + g.out_choff;
  }
  // ---- B fragments of GEMM 1, pass 0 (this wave's 32 columns, all 16 k groups): issued before the wait ----
  float4 b1v[KC / 8];
#pragma unroll
  for (int gi = 0; gi < KC / 8; ++gi)
    b1v[gi] = reinterpret_cast<const float4*>(g.w1)[(size_t)(2 * gi + kh) * N1T + wave * 32 + l31];
  __syncthreads();
  RTPOSE_TAIL_TL(1);

  // ---- X tile: 64 px x 32 planes of 16 bytes; consecutive lanes = consecutive planes of a pixel (512 B runs) ----
  // (offsets, then all loads, then all LDS writes: written as one loop the table read of piece i + 1 - LDS, like X - could
  //  not be moved across the LDS write of piece i, and the eight global round trips ran one after the other: 7.3 of a block's
  //  29 us, tools/exp/tail_timeline.py)
  {
    constexpr int NX = BM * (KC / 4) / 256;
    int qx[NX];
    float4 xv[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) qx[i] = qin[(tid + 256 * i) >> 5];
#pragma unroll
    for (int i = 0; i < NX; ++i) xv[i] = *reinterpret_cast<const float4*>(g.in + (size_t)qx[i] + (tid & 31) * 4);
#pragma unroll
    for (int i = 0; i < NX; ++i) X[(tid & 31) * PS + ((tid + 256 * i) >> 5)] = xv[i];
  }
  __syncthreads();
  RTPOSE_TAIL_TL(2);

  const int mf2 = wave & 1, nf2 = wave >> 1;
  const int col2 = nf2 * 32 + l31;
  const bool live2 = nf2 * 32 < g.cout2;  // wave-uniform
  floatx16 acc2;
  {
    const float bias2 = g.b2[col2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = bias2;
  }
#pragma unroll 1
  for (int ps = 0; ps < NP; ++ps) {
    // ---- GEMM 1: 64 px x 32 columns per wave, K = 128 -------------------------------------------------------
    const int col1 = ps * N1 + wave * 32 + l31;
    const float bias1 = g.b1[col1];
    floatx16 acc[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mf][r] = bias1;
#pragma unroll
    for (int gi = 0; gi < KC / 8; ++gi) {
      const float4 a0 = X[(2 * gi + kh) * PS + l31], a1 = X[(2 * gi + kh) * PS + 32 + l31];
      const float a0v[4] = {a0.x, a0.y, a0.z, a0.w}, a1v[4] = {a1.x, a1.y, a1.z, a1.w};
      const float bv[4] = {b1v[gi].x, b1v[gi].y, b1v[gi].z, b1v[gi].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[j], bv[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[j], bv[j], acc[1], 0, 0, 0);
      }
    }
    // B fragments of GEMM 2 for this pass' 128 k rows (requested now: their latency hides under the LDS round trip
    // of the intermediate), and GEMM 1's for the next pass
    // (round 6: the waves whose 32 output columns are all padding - columns 32..63 of the 19-column heat-map branch - skip
    //  their half of GEMM 2: their SIMD's matrix pipe goes to the co-resident blocks)
    float4 b2v[N1 / 8];
    if (live2) {
#pragma unroll
      for (int gi = 0; gi < N1 / 8; ++gi)
        b2v[gi] = reinterpret_cast<const float4*>(g.w2)[(size_t)(ps * (N1 / 4) + 2 * gi + kh) * N2 + col2];
    }
    if (ps + 1 < NP) {
#pragma unroll
      for (int gi = 0; gi < KC / 8; ++gi)
        b1v[gi] = reinterpret_cast<const float4*>(g.w1)[(size_t)(2 * gi + kh) * N1T + (ps + 1) * N1 + wave * 32 + l31];
    }
    // ReLU, then the intermediate in the A layout: T[column / 4][pixel].[column % 4]
    // (register r of a lane = pixel (r / 4) * 8 + 4 kh + r % 4 of the fragment, column wave * 32 + l31 of the pass)
    if (NP == 1) { RTPOSE_TAIL_TL(3); }
    __syncthreads();  // every wave has read its last X fragment (one pass) / its last T fragment of the previous pass
    if (NP == 1) { RTPOSE_TAIL_TL(4); }
    {
      const int cl = wave * 32 + l31;
      float* Tf = reinterpret_cast<float*>(T) + ((cl >> 2) * PS) * 4 + (cl & 3);
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = mf * 32 + (r >> 2) * 8 + 4 * kh + (r & 3);
          Tf[px * 4] = fmaxf(acc[mf][r], 0.f);
        }
    }
    __syncthreads();
    if (NP == 1) { RTPOSE_TAIL_TL(5); }

    // ---- GEMM 2: 32 px x 32 columns per wave (2 x 2 waves), the next 128 of its K ----------------------------
    if (live2) {
#pragma unroll
      for (int gi = 0; gi < N1 / 8; ++gi) {
        const float4 a = T[(2 * gi + kh) * PS + mf2 * 32 + l31];
        const float av[4] = {a.x, a.y, a.z, a.w};
        const float bv[4] = {b2v[gi].x, b2v[gi].y, b2v[gi].z, b2v[gi].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc2, 0, 0, 0);
      }
    }
  }
  RTPOSE_TAIL_TL(6);
  if (col2 < g.cout2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = mf2 * 32 + (r >> 2) * 8 + 4 * kh + (r & 3);
      if (mt * BM + px < A.M) g.out[(size_t)qout[px] + col2] = acc2[r];
    }
  }
  RTPOSE_TAIL_TL(7);
}

}  // namespace tail

// Two grouped 1x1 convs back to back: d1[g] = 128 -> 128 | 512 (+ReLU), d2[g] = that -> cout2 <= 64 (no ReLU) reading d1[g]'s
// output, which is never written.  Descriptors as for rtpose_conv2d (k = 1, plain packing); d1[g].out / lout are ignored.
int conv_tail_fits(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups) {
  if (!d1 || !d2 || ngroups < 1 || ngroups > 2) return 0;
  for (int i = 0; i < ngroups; ++i) {
    if (d1[i].k != 1 || d2[i].k != 1 || d1[i].cin != tail::KC || (d1[i].cout != tail::N1 && d1[i].cout != 4 * tail::N1) ||
        d1[i].cout != d1[0].cout || d2[i].cin != d1[i].cout ||
        d2[i].cout > tail::N2 || d2[i].cout < 1 || !d1[i].relu || d2[i].relu || d1[i].pool || d2[i].pool ||
        d1[i].out_cmap || d2[i].out_cmap || (d1[i].lin.cstride % 4) || (d1[i].lin.choff % 4) ||
        d1[i].lin.choff + tail::KC > d1[i].lin.cstride || d2[i].lout.choff + d2[i].cout > d2[i].lout.cstride)
      return 0;
  }
  return 1;
}

int conv_tail_launch(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups, int N, int H, int W,
                     hipStream_t s) {
  using namespace tail;
  if (!conv_tail_fits(d1, d2, ngroups) || N <= 0 || H <= 0 || W <= 0)
    return fail(RTPOSE_E_INVAL, "conv_tail: needs 128 -> 128 | 512 (+ReLU) -> <= 64 pointwise convs");
  RTPOSE_REFUSE_PLANES(d1, ngroups, "conv_tail");
  RTPOSE_REFUSE_PLANES(d2, ngroups, "conv_tail");
  const long M = (long)N * H * W;
  if (M > 0x7fffffffL) return fail(RTPOSE_E_INVAL, "conv_tail: tensor too large");
  Args a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < ngroups; ++i) {
    Group& g = a.g[i];
    g.in = d1[i].in;
    g.w1 = d1[i].w_packed;
    g.b1 = d1[i].bias_packed;
    g.w2 = d2[i].w_packed;
    g.b2 = d2[i].bias_packed;
    g.out = d2[i].out;
    g.in_cstride = d1[i].lin.cstride;
    g.in_choff = d1[i].lin.choff;
    g.in_ws = d1[i].lin.ws;
    g.in_hs = d1[i].lin.hs;
    g.in_lead = d1[i].lin.lead;
    g.out_cstride = d2[i].lout.cstride;
    g.out_choff = d2[i].lout.choff;
    g.out_ws = d2[i].lout.ws;
    g.out_hs = d2[i].lout.hs;
    g.out_lead = d2[i].lout.lead;
    g.cout2 = d2[i].cout;
    // 32-bit element offsets inside the kernel
    if (rtpose_layout_pixels(&d1[i].lin, N, H, W) * (size_t)d1[i].lin.cstride > 0x7fffffffULL ||
        rtpose_layout_pixels(&d2[i].lout, N, H, W) * (size_t)d2[i].lout.cstride > 0x7fffffffULL)
      return fail(RTPOSE_E_INVAL, "conv_tail: buffer too large for 32-bit offsets");
  }
  a.N = N;
  a.H = H;
  a.W = W;
  a.M = (int)M;
  a.mtiles = ceil_div((int)M, BM);
  a.ngroups = ngroups;
  const int np = d1[0].cout / N1;
  const size_t lds = (size_t)(np == 1 ? 1 : 2) * (KC / 4) * PS * sizeof(float4) + 2 * BM * sizeof(int);
  if (np == 1) {
    hipLaunchKernelGGL(tail_kernel<1>, dim3((unsigned)a.mtiles * ngroups), dim3(256), lds, s, a);
  } else {
    static PerDeviceOnce attr_set;
    const int dev = current_device();
    if (!attr_set.is_set(dev)) {
      RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tail_kernel<4>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      attr_set.set(dev);
    }
    hipLaunchKernelGGL(tail_kernel<4>, dim3((unsigned)a.mtiles * ngroups), dim3(256), lds, s, a);
  }
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace rtpose

extern "C" {

#ifdef RTPOSE_EXP_TAIL_TIMELINE
int rtpose_exp_tail_timeline(unsigned long long* out) {  // [4096 blocks][8 stamps]
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rtpose::tail::g_tail_tl), sizeof(unsigned long long) * 4096 * 8);
}
#endif

int rtpose_conv1x1_pair_fits(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups) {
  return rtpose::conv_tail_fits(d1, d2, ngroups);
}

int rtpose_conv1x1_pair(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups, int N, int H, int W,
                        void* stream) {
  return rtpose::conv_tail_launch(d1, d2, ngroups, N, H, W, rtpose::as_stream(stream));
}

}  // extern "C"
