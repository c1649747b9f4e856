// The two trailing 1x1 convolutions of a stage branch as ONE back-to-back GEMM launch in the bf16 plan (BASELINE configs[2]
// arithmetic: bf16 operands, exact products, fp32 accumulation), gfx950 (MI355X), v_mfma_f32_32x32x16_bf16:
//   Mconv6_stageN_Lb = nn.Conv2d(128, 128, 1) + nn.ReLU,  Mconv7_stageN_Lb = nn.Conv2d(128, 38 | 19, 1)      (NP = 1)
//   conv5_4_CPM_Lb   = nn.Conv2d(128, 512, 1) + nn.ReLU,  conv5_5_CPM_Lb   = nn.Conv2d(512, 38 | 19, 1)      (NP = 4)
// (lib/network/rtpose_vgg.py:120-127, :101-105), both branches of the stage in one grid - the bf16 sibling of conv_tail.hip.
// As two launches of the generic bf16 kernel these K = 128 / 512 GEMMs took 27 + 28 us (87 + 59 in stage 1) at 0.05 of the
// bf16 matrix peak: per-block prologue and epilogue around eight MFMA steps, and the intermediate's round trip through HBM.
// A block owns 64 pixels of one branch:
//   X [64 px x 128 ch] -> LDS (16-byte pieces of 8 channels: one ds_read_b128 = the K = 16 operand of a lane)
//   GEMM 1, TRANSPOSED (filters as the row operand, pixels as the column operand): wave w computes channels 32 w .. + 31 of
//     the pass for all 64 pixels; a lane then holds ONE pixel and channels 8 (r / 4) + 4 kh + r % 4 - after + bias, ReLU and the
//     rounding to bf16 the two lane halves swap 4-channel halves (v_permlane32_swap) and every lane writes whole 8-channel pieces
//     of ITS pixel into LDS: the intermediate is in the operand layout of GEMM 2 without a transpose
//   GEMM 2, transposed too: wave (mf, nf) computes output channels 32 mf .. of the pixels 32 nf ..; its accumulators live
//     across the NP passes of 128 intermediate channels (k ascending)
//   + bias -> LDS [pixel][channel] fp32 -> row-contiguous stores into the stage's concat buffer (bf16) or, for stage 6, into
//     the fp32 record the decoder reads.
// The intermediate is rounded to bf16 exactly where the two-launch form rounded it (the output of Mconv6 / conv5_4_CPM), so the
// contract is the one of two rtpose_conv2d_bf16 launches (oracle/net_oracle.py:forward_bf16_emulated); the fp32 sums run in
// another order than the generic kernel's, so results agree to the rounding of the last fp32 bit, not bit for bit.
// Weights are the generic kernel's k = 1 packing [cin / 8][cout_pad][8 bf16] (rtpose_pack_conv_weights_bf16).
#include <hip/hip_runtime.h>

#include "common.h"

namespace rtpose {

namespace tailb {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64;      // pixels per block
constexpr int KC = 128;     // input channels, and intermediate channels per pass
constexpr int N2 = 64;      // output columns (38 / 19 padded)
constexpr int PS = BM + 1;  // 16-byte pieces per 8-channel plane in LDS (+1: consecutive planes of a pixel 4 banks apart)
constexpr int OS = N2 + 1;  // floats per pixel row of the output staging

struct Group {
  const unsigned short* in;
  const uintx4* w1;
  const float* b1;
  const uintx4* w2;
  const float* b2;
  void* out;
  int in_cstride, in_choff, in_ws, in_hs, in_lead;
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  int cout2, coutp1, coutp2;
};

struct Args {
  Group g[2];
  int N, H, W, M, ngroups, out_f32, relu2;
};

__device__ __forceinline__ unsigned pack2(float a, float b) {  // (bf16(a), bf16(b)) RNE, a in the low half
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float fl2 __attribute__((ext_vector_type(2)));
  const fl2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}

constexpr size_t lds_bytes() { return (size_t)2 * 16 * PS * 16 + (size_t)BM * OS * 4 + 2 * BM * 4; }

template <int NP>
__global__ __launch_bounds__(256) void tail_bf16_kernel(const Args A) {
  extern __shared__ __attribute__((aligned(16))) uintx4 lds4[];
  uintx4* const X = lds4;                 // [16 planes][PS]: the input tile
  uintx4* const T = lds4 + 16 * PS;       // [16 planes][PS]: the intermediate of one pass
  float* const O = reinterpret_cast<float*>(lds4 + 32 * PS);  // [BM][OS]
  int* const qin = reinterpret_cast<int*>(O + BM * OS);
  int* const qout = qin + BM;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int mt = blockIdx.x / A.ngroups, grp = blockIdx.x - mt * A.ngroups;
  const Group g = grp ? A.g[1] : A.g[0];

  // ---- pixel -> element offsets of the block's 64 pixels (pixels past the end replay the last one; not stored) ----
  if (tid < BM) {
    const int m = mt * BM + tid;
    const int mc = min(m, A.M - 1);
    const int HW = A.H * A.W;
    const int n = mc / HW, r = mc - n * HW;
    const int y = r / A.W, x = r - y * A.W;
    qin[tid] = (g.in_lead + (n * g.in_hs + y) * g.in_ws + x) * g.in_cstride + g.in_choff;
    qout[tid] = m < A.M ? (g.out_lead + (n * g.out_hs + y) * g.out_ws + x) * g.out_cstride + g.out_choff : -1;
  }
  // ---- A fragments of GEMM 1, pass 0 (this wave's 32 channels, all 8 k steps): requested before the first wait ----
  uintx4 w1v[KC / 16];
#pragma unroll
  for (int ks = 0; ks < KC / 16; ++ks) w1v[ks] = g.w1[(size_t)(2 * ks + kh) * g.coutp1 + wave * 32 + l31];
  __syncthreads();

  // ---- X tile: 64 px x 16 pieces; consecutive lanes = consecutive pieces of a pixel (256-byte runs) ----
  // (offsets, then all loads, then all LDS writes: as one loop the table read of piece i + 1 could not be moved across the LDS
  //  write of piece i and the global round trips ran one after the other - found in the fp32 kernel, tools/exp/tail_timeline.py)
  {
    constexpr int NX = BM * 16 / 256;
    int qx[NX];
    uintx4 xv[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) qx[i] = qin[(tid + 256 * i) >> 4];
#pragma unroll
    for (int i = 0; i < NX; ++i) xv[i] = *reinterpret_cast<const uintx4*>(g.in + (size_t)qx[i] + (tid & 15) * 8);
#pragma unroll
    for (int i = 0; i < NX; ++i) X[(tid & 15) * PS + ((tid + 256 * i) >> 4)] = xv[i];
  }

  const int mf2 = wave & 1, nf2 = wave >> 1;
  const bool live2 = mf2 * 32 < g.cout2;  // (the heat-map branch has 19 columns: the waves of rows 32..63 have nothing to do)
  floatx16 acc2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 b = *reinterpret_cast<const float4*>(g.b2 + mf2 * 32 + 8 * j + 4 * kh);
    acc2[4 * j] = b.x;
    acc2[4 * j + 1] = b.y;
    acc2[4 * j + 2] = b.z;
    acc2[4 * j + 3] = b.w;
  }
  __syncthreads();

#pragma unroll 1
  for (int ps = 0; ps < NP; ++ps) {
    // ---- GEMM 1: channels ps * 128 + 32 wave .. + 31 (rows) x 64 pixels (columns), K = 128 ----
    floatx16 acc[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 b = *reinterpret_cast<const float4*>(g.b1 + ps * KC + wave * 32 + 8 * j + 4 * kh);
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        acc[nf][4 * j] = b.x;
        acc[nf][4 * j + 1] = b.y;
        acc[nf][4 * j + 2] = b.z;
        acc[nf][4 * j + 3] = b.w;
      }
    }
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      const bf16x8 a = __builtin_bit_cast(bf16x8, w1v[ks]);
      const uintx4 x0 = X[(2 * ks + kh) * PS + l31], x1 = X[(2 * ks + kh) * PS + 32 + l31];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, x0), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, x1), acc[1], 0, 0, 0);
    }
    // A fragments of GEMM 2 for this pass' 128 k rows (their latency hides under the LDS round trip of the intermediate),
    // and GEMM 1's for the next pass
    uintx4 w2v[KC / 16];
    if (live2) {
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks)
        w2v[ks] = g.w2[(size_t)(ps * (KC / 8) + 2 * ks + kh) * g.coutp2 + mf2 * 32 + l31];
    }
    if (NP > 1 && ps + 1 < NP) {
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks)
        w1v[ks] = g.w1[(size_t)(2 * ks + kh) * g.coutp1 + (ps + 1) * KC + wave * 32 + l31];
    }
    // ---- ReLU, round to bf16, swap halves: lane (pixel, kh) gets the 8 channels of group 2 m + kh of the wave's block ----
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        unsigned a[2], b[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          a[e] = pack2(fmaxf(acc[nf][8 * m + 2 * e], 0.f), fmaxf(acc[nf][8 * m + 2 * e + 1], 0.f));
          b[e] = pack2(fmaxf(acc[nf][8 * m + 4 + 2 * e], 0.f), fmaxf(acc[nf][8 * m + 4 + 2 * e + 1], 0.f));
          const auto sw = __builtin_amdgcn_permlane32_swap(a[e], b[e], false, false);  // lanes 32..63 of a <-> lanes 0..31 of b
          a[e] = sw[0];
          b[e] = sw[1];
        }
        const uintx4 piece = {a[0], a[1], b[0], b[1]};
        T[(wave * 4 + 2 * m + kh) * PS + nf * 32 + l31] = piece;
      }
    __syncthreads();
    // ---- GEMM 2: output channels 32 mf2 .. (rows) x pixels 32 nf2 .. (columns), this pass' 128 of its K ----
    if (live2) {
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        const uintx4 t = T[(2 * ks + kh) * PS + nf2 * 32 + l31];
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w2v[ks]), __builtin_bit_cast(bf16x8, t), acc2,
                                                       0, 0, 0);
      }
    }
    if (NP > 1) __syncthreads();  // T is rewritten by the next pass
  }

  // ---- output: registers -> O[pixel][channel] -> row-contiguous stores ----
  if (live2) {
    float* const orow = O + (nf2 * 32 + l31) * OS + mf2 * 32 + 4 * kh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc2[r];
      orow[8 * (r >> 2) + (r & 3)] = A.relu2 ? fmaxf(v, 0.f) : v;
    }
  }
  __syncthreads();
  {
    const int p = tid >> 2, c0 = (tid & 3) * 16;
    const int q = qout[p];
    if (q >= 0) {
      const float* src = O + p * OS;
      if (A.out_f32) {
        float* dst = static_cast<float*>(g.out) + q;
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c0 + c < g.cout2) dst[c0 + c] = src[c0 + c];
      } else {
        unsigned short* dst = static_cast<unsigned short*>(g.out) + q;
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c0 + c < g.cout2) dst[c0 + c] = __builtin_bit_cast(unsigned short, (__bf16)src[c0 + c]);
      }
    }
  }
}

}  // namespace tailb

// d1[g] / d2[g]: the two convs of branch g (k = 1; d1: 128 -> 128 | 512 with ReLU, d2: -> cout <= 64).  d1[g].in / lin: bf16
// input (16-byte aligned slices); d2[g].out / lout: bf16 elements, or fp32 when out_f32; d1[g].out is not touched.
int conv_tail_bf16_fits(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups) {
  if (!d1 || !d2 || ngroups < 1 || ngroups > 2) return 0;
  for (int g = 0; g < ngroups; ++g) {
    if (d1[g].k != 1 || d2[g].k != 1 || d1[g].cin != tailb::KC || !d1[g].relu || d1[g].pool || d2[g].pool) return 0;
    if (d1[g].cout != 128 && d1[g].cout != 512) return 0;
    if (d1[g].cout != d1[0].cout || d2[g].cin != d1[g].cout || d2[g].cout < 1 || d2[g].cout > tailb::N2) return 0;
    if ((d1[g].lin.cstride % 8) || (d1[g].lin.choff % 8)) return 0;
    if (d2[g].relu != d2[0].relu) return 0;
  }
  return 1;
}

int conv_tail_bf16_launch(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups, int N, int H, int W,
                          int out_f32, hipStream_t s) {
  using namespace tailb;
  if (!conv_tail_bf16_fits(d1, d2, ngroups))
    return fail(RTPOSE_E_INVAL, "conv1x1_pair_bf16: not a 128 -> 128 | 512 (ReLU) -> <= 64 pair of 1x1 convs on 16-byte aligned bf16 slices");
  if (N <= 0 || H <= 0 || W <= 0) return fail(RTPOSE_E_INVAL, "conv1x1_pair_bf16: empty tensor");
  Args a;
  memset(&a, 0, sizeof(a));
  for (int g = 0; g < ngroups; ++g) {
    if (!d1[g].in || !d1[g].w_packed || !d1[g].bias_packed || !d2[g].w_packed || !d2[g].bias_packed || !d2[g].out)
      return fail(RTPOSE_E_INVAL, "conv1x1_pair_bf16: NULL argument");
    if (rtpose_layout_pixels(&d1[g].lin, N, H, W) * (size_t)d1[g].lin.cstride >= ((size_t)1 << 31) ||
        rtpose_layout_pixels(&d2[g].lout, N, H, W) * (size_t)d2[g].lout.cstride >= ((size_t)1 << 31))
      return fail(RTPOSE_E_INVAL, "conv1x1_pair_bf16: tensors must be below 2^31 elements (32-bit offsets)");
    Group& q = a.g[g];
    q.in = reinterpret_cast<const unsigned short*>(d1[g].in);
    q.w1 = reinterpret_cast<const uintx4*>(d1[g].w_packed);
    q.b1 = d1[g].bias_packed;
    q.w2 = reinterpret_cast<const uintx4*>(d2[g].w_packed);
    q.b2 = d2[g].bias_packed;
    q.out = d2[g].out;
    q.in_cstride = d1[g].lin.cstride;
    q.in_choff = d1[g].lin.choff;
    q.in_ws = d1[g].lin.ws;
    q.in_hs = d1[g].lin.hs;
    q.in_lead = d1[g].lin.lead;
    q.out_cstride = d2[g].lout.cstride;
    q.out_choff = d2[g].lout.choff;
    q.out_ws = d2[g].lout.ws;
    q.out_hs = d2[g].lout.hs;
    q.out_lead = d2[g].lout.lead;
    q.cout2 = d2[g].cout;
    q.coutp1 = cout_pad(d1[g].cout);
    q.coutp2 = cout_pad(d2[g].cout);
    if (q.out_choff + q.cout2 > q.out_cstride) return fail(RTPOSE_E_INVAL, "conv1x1_pair_bf16: output slice exceeds cstride");
  }
  a.N = N;
  a.H = H;
  a.W = W;
  a.M = N * H * W;
  a.ngroups = ngroups;
  a.out_f32 = out_f32 ? 1 : 0;
  a.relu2 = d2[0].relu ? 1 : 0;
  const int mtiles = ceil_div(a.M, BM);
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tail_bf16_kernel<1>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tail_bf16_kernel<4>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_set.set(dev);
  }
  if (d1[0].cout == 128)
    hipLaunchKernelGGL(tail_bf16_kernel<1>, dim3(mtiles * ngroups), dim3(256), lds_bytes(), s, a);
  else
    hipLaunchKernelGGL(tail_bf16_kernel<4>, dim3(mtiles * ngroups), dim3(256), lds_bytes(), s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace rtpose

extern "C" {

int rtpose_conv1x1_pair_bf16_fits(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups) {
  return rtpose::conv_tail_bf16_fits(d1, d2, ngroups);
}

int rtpose_conv1x1_pair_bf16(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups, int N, int H, int W,
                             int out_f32, void* stream) {
  return rtpose::conv_tail_bf16_launch(d1, d2, ngroups, N, H, W, out_f32, rtpose::as_stream(stream));
}

}  // extern "C"
