// Fused pointwise (1x1) GEMM for the ShuffleNetV2 pose network (BASELINE configs[3]) — fp32,
// v_mfma_f32_32x32x2_f32, gfx950.
//
// Stands in for the module chains of lib/network/rtpose_shufflenetV2.py BasicBlock (:22-63):
//     conv_bn_relu 1x1                                              (conv.0, conv0.1, conv5, heads)
//     conv_bn depthwise 3x3  ->  conv_bn_relu 1x1                   (conv.1 -> conv.2, conv0.0 -> conv0.1)
//     ... -> torch.cat((x1, x2), 1) -> channel_shuffle(2)           (:56-62)
// as ONE launch per chain:
//
//   * GEMM view  M = pixels, N = output channels, K = input channels.  Block tile 64(M) x up to
//     256(N): the four waves split N (1 x 4 waves of 64 x {32,64}; 2 x 2 of 32 x 32 for 64-column
//     layers), so the A tile of a pixel strip is staged ONCE for all output channels (the generic conv
//     kernel used 128 x 64 tiles: the A tile was re-staged by every N tile and a K-chunk barrier sat
//     between every 16 MFMAs of a wave - these K = 24..464 GEMMs ran at 24-70 TFLOP/s).
//   * K is walked in 32-channel chunks through a double-buffered LDS tile [8 planes][66 pixels][4
//     floats] (one ds_read_b128 per A fragment, the generic kernel's image); the next chunk's pieces
//     are in flight in registers while the current chunk is multiplied (64 MFMAs per wave per barrier
//     at N = 256).  B (weights, packed [K/4][coutp][4]) streams L2 -> registers one k-group ahead.
//   * DW: the A tile is PRODUCED in the kernel: the chunk's input halo (the strip + one row above /
//     below, contiguous in the shared-gap layout) is staged in LDS, the depthwise 3x3 (+ folded BN
//     bias) is evaluated on the VALU into the A tile, then multiplied.  The depthwise activation
//     tensor never exists in HBM and its launch is gone.
//   * PT: the block also copies the pass-through half x1 of its pixels to the even channel slots of the
//     output buffer while its GEMM result goes to the odd slots through out_cmap - cat + channel_shuffle
//     cost no launch and no extra pass.
//   * 2 blocks per CU (<= 72 KB LDS, <= 256 VGPRs): one block's staging / depthwise / epilogue phases
//     run under the other's MFMAs.
#include <hip/hip_runtime.h>

#include "common.h"

namespace rtpose {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef const floatx4 __attribute__((address_space(1)))* gcf4_t;
__device__ __forceinline__ float4 pw_gload4(const void* p) {  // explicit global address space (no FLAT loads)
  const floatx4 v = *(gcf4_t)(unsigned long long)(p);
  return make_float4(v[0], v[1], v[2], v[3]);
}

struct PwView {  // one activation tensor (shared-gap padded NHWC slice)
  const float* base;
  int cstride, choff, ws, hs, lead;
};

struct PwArgs {
  PwView in;          // A source: the GEMM input (DW = 0) or the depthwise conv's input (DW = 1, gap >= 1)
  const float* dw_w;  // DW: [9][K] taps (ky*3+kx major) and
  const float* dw_b;  //     [K] bias of the depthwise conv (BN folded)
  const float* w;     // packed pointwise weights [K/4][coutp][4]
  const float* bias;  // [coutp]
  float* out;
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  const int32_t* out_cmap;  // optional [coutp]: column n -> absolute channel (< 0: not stored)
  PwView pt;                // PT: source of the pass-through channels (same N, H, W)
  const int32_t* pt_cmap;   // PT: [pt_c] source channel i -> absolute output channel
  int pt_c;
  int N, H, W, M;
  int K, coutp, cout, relu;
  int nps;  // DW: LDS plane stride (pixels) of the staged halo
};

constexpr int kPwBM = 64;   // pixels per block
constexpr int kPwQS = 66;   // LDS plane stride of the A tile (pixels): planes 8 banks apart
constexpr int kPwPL = 8;    // 16-byte channel-group planes per K chunk (32 channels)
constexpr int kPwMaxStage = 8;  // DW: staged 16-byte pieces per thread per chunk (<= 256 halo pixels)

#define RTPOSE_PW_PIN()          \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)

// WM x WN waves (WM * WN = 4), wave tile (32 MF) x (32 NFW), WM * MF = 2.
template <int WM, int MF, int NFW, bool DW>
__global__ __launch_bounds__(256, 2) void pw_gemm_f32(const PwArgs A) {
  constexpr int WN = 4 / WM;
  constexpr int BN = WN * NFW * 32;
  static_assert(WM * MF * 32 == kPwBM, "block tile is 64 pixels");
  extern __shared__ __attribute__((aligned(16))) float4 smem4[];
  __shared__ int s_qin[kPwBM], s_qout[kPwBM], s_qpt[kPwBM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int l31 = lane & 31, kh = lane >> 5;
  const int m0 = blockIdx.x * kPwBM;
  const int HW = A.H * A.W;

  // ---- pixel tables: pixel index of tile row r in the input / output / pass-through layouts ----
  if (tid < kPwBM) {
    const int m = min(m0 + tid, A.M - 1);  // rows past the end replay the last pixel (never stored)
    const int n = m / HW, r = m - n * HW;
    const int y = r / A.W, x = r - y * A.W;
    s_qin[tid] = A.in.lead + (n * A.in.hs + y) * A.in.ws + x;
    s_qout[tid] = (m0 + tid < A.M) ? A.out_lead + (n * A.out_hs + y) * A.out_ws + x : -1;
    s_qpt[tid] = A.pt.base ? A.pt.lead + (n * A.pt.hs + y) * A.pt.ws + x : 0;
  }
  __syncthreads();

  // ---- LDS carve-up (float4 units) ---------------------------------------------------------
  constexpr int ASUB = kPwPL * kPwQS;  // one A chunk buffer
  float4* a_lds = smem4;               // [2][ASUB]
  float4* st_lds = smem4 + 2 * ASUB;   // DW: [2][kPwPL * nps]
  const int nps = A.nps;

  // ---- pass-through half: x1 -> even slots (cat + channel_shuffle folded into the store) ------
  if (A.pt.base) {
    const int g4 = (A.pt_c + 3) >> 2;
    for (int it = tid; it < kPwBM * g4; it += 256) {
      const int p = it / g4, g = it - p * g4;
      const int qo = s_qout[p];
      if (qo < 0) continue;
      const float4 v = pw_gload4(A.pt.base + (size_t)s_qpt[p] * A.pt.cstride + A.pt.choff + 4 * g);
      const float vv[4] = {v.x, v.y, v.z, v.w};
      float* o = A.out + (size_t)qo * A.out_cstride;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * g + e < A.pt_c) o[A.pt_cmap[4 * g + e]] = vv[e];
    }
  }

  // ---- A staging geometry ----------------------------------------------------------------------
  // DW = 0: piece u of thread t = (pixel t/8 + 32 u, plane t%8) of the chunk, u < 2
  // DW = 1: the chunk's halo: pixels [q_org, q_org + np) x 8 planes, piece i = t + 256 u -> (i / 8, i % 8)
  const int pl = tid & 7;
  const float* in_base = A.in.base + A.in.choff + 4 * pl;
  int q_org = 0, np = 0;
  size_t goff[DW ? kPwMaxStage : 2];
  int n_st = 2;
  if (DW) {
    q_org = s_qin[0] - A.in.ws - 1;
    np = s_qin[kPwBM - 1] + A.in.ws + 1 - q_org + 1;
    n_st = (np * kPwPL + 255) >> 8;
#pragma unroll
    for (int u = 0; u < kPwMaxStage; ++u) goff[u] = (size_t)(q_org + min((tid >> 3) + 32 * u, np - 1)) * A.in.cstride;
  } else {
#pragma unroll
    for (int u = 0; u < 2; ++u) goff[u] = (size_t)s_qin[(tid >> 3) + 32 * u] * A.in.cstride;
  }
  // DW compute items of this thread: (pixel tid/8 + 32 u, plane tid%8): halo-relative pixel index
  int sp[2] = {0, 0};
  if (DW) {
    sp[0] = s_qin[tid >> 3] - q_org;
    sp[1] = s_qin[(tid >> 3) + 32] - q_org;
  }

  const int nch = (A.K + 31) >> 5;
  const int npass = A.coutp / BN;
  const float4* w4 = reinterpret_cast<const float4*>(A.w);

  for (int pass = 0; pass < npass; ++pass) {
    const int ncol = pass * BN + wn * (32 * NFW) + l31;  // column of n-fragment 0 of this lane

    floatx16 acc[MF][NFW];
#pragma unroll
    for (int fn = 0; fn < NFW; ++fn) {
      const float b0 = A.bias[ncol + fn * 32];
#pragma unroll
      for (int fm = 0; fm < MF; ++fm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[fm][fn][r] = b0;
    }

    // pieces of chunk 0 into registers
    float4 sr[DW ? kPwMaxStage : 2];
#pragma unroll
    for (int u = 0; u < (DW ? kPwMaxStage : 2); ++u) {
      sr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (u < n_st && 4 * pl < A.K) sr[u] = pw_gload4(in_base + goff[u]);
    }
    // B fragments of k-group 0
    float4 bcur[NFW], bnxt[NFW];
#pragma unroll
    for (int fn = 0; fn < NFW; ++fn) bnxt[fn] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int fn = 0; fn < NFW; ++fn) bcur[fn] = pw_gload4(w4 + (size_t)kh * A.coutp + ncol + fn * 32);

    const int gtot = A.K >> 3;  // 8-channel k-groups in all
    for (int c = 0; c < nch; ++c) {
      const int buf = c & 1;
      const int c0 = c << 5;
      float4* a_buf = a_lds + buf * ASUB;
      float4 dww[9], dwb;
      if (DW) {
        // the depthwise taps of this thread's channel group (same for both of its pixels)
        const bool ok = c0 + 4 * pl < A.K;
#pragma unroll
        for (int t = 0; t < 9; ++t)
          dww[t] = ok ? pw_gload4(A.dw_w + (size_t)t * A.K + c0 + 4 * pl) : make_float4(0.f, 0.f, 0.f, 0.f);
        dwb = ok ? pw_gload4(A.dw_b + c0 + 4 * pl) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4* st = st_lds + buf * (kPwPL * nps);
        // park the staged halo pieces, fetch the next chunk's
#pragma unroll
        for (int u = 0; u < kPwMaxStage; ++u)
          if (u < n_st && (tid >> 3) + 32 * u < np) st[pl * nps + (tid >> 3) + 32 * u] = sr[u];
        const bool nxt_ok = c + 1 < nch && c0 + 32 + 4 * pl < A.K;
#pragma unroll
        for (int u = 0; u < kPwMaxStage; ++u)
          if (u < n_st && nxt_ok) sr[u] = pw_gload4(in_base + goff[u] + c0 + 32);
        __syncthreads();  // halo of chunk c visible
        // depthwise 3x3 (+bias) -> A tile.  Tap (ky, kx) of pixel q is pixel q + (ky-1) ws + (kx-1).
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float4 v = dwb;
          const float4* s0 = st + pl * nps + sp[u] - A.in.ws - 1;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const float4 x = s0[ky * A.in.ws + kx];
              const float4 ww = dww[ky * 3 + kx];
              v.x += x.x * ww.x;
              v.y += x.y * ww.y;
              v.z += x.z * ww.z;
              v.w += x.w * ww.w;
            }
          a_buf[pl * kPwQS + (tid >> 3) + 32 * u] = v;
        }
        __syncthreads();  // A tile of chunk c visible
      } else {
#pragma unroll
        for (int u = 0; u < 2; ++u) a_buf[pl * kPwQS + (tid >> 3) + 32 * u] = sr[u];
        const bool nxt_ok = c + 1 < nch && c0 + 32 + 4 * pl < A.K;
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (nxt_ok) sr[u] = pw_gload4(in_base + goff[u] + c0 + 32);
        __syncthreads();  // A tile of chunk c visible (the other buffer was last read before the previous barrier)
      }

      // ---- multiply chunk c: ng k-groups of 8 channels, 4 MFMAs per (m, n) fragment pair each ----
      // Operands of the NEXT k-group are requested before the current group's MFMAs are issued (B from
      // L2 - also across the chunk boundary - A from LDS) and consumed one group later.  A full chunk is
      // straight-line code with the two register sets alternating: no moves, no branches around the
      // loads, so the s_waitcnt the compiler places in front of a group's MFMAs leaves the younger
      // requests in flight (in a rolled loop with guards it drained vmcnt to 0 at every group).
      const int ng = min(4, gtot - 4 * c);
      const float4* a_rd = a_buf + kh * kPwQS + wm * (32 * MF) + l31;
      const float4* w_rd = w4 + (size_t)kh * A.coutp + ncol;  // + 2 * group * coutp
#define RTPOSE_PW_BLOAD(DST, GG)                                                              \
  _Pragma("unroll") for (int fn = 0; fn < NFW; ++fn) DST[fn] = pw_gload4(w_rd + (size_t)2 * (GG) * A.coutp + fn * 32)
#define RTPOSE_PW_ALOAD(DST, GI) \
  _Pragma("unroll") for (int fm = 0; fm < MF; ++fm) DST[fm] = a_rd[2 * (GI) * kPwQS + fm * 32]
#define RTPOSE_PW_MUL(AV, BV)                                                                           \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                       \
    _Pragma("unroll") for (int fn = 0; fn < NFW; ++fn) {                                                \
      const float bv_[4] = {BV[fn].x, BV[fn].y, BV[fn].z, BV[fn].w};                                    \
      _Pragma("unroll") for (int fm = 0; fm < MF; ++fm) {                                               \
        const float av_[4] = {AV[fm].x, AV[fm].y, AV[fm].z, AV[fm].w};                                  \
        acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_[j], bv_[j], acc[fm][fn], 0, 0, 0);       \
      }                                                                                                 \
    }                                                                                                   \
  }
      float4 a0[MF], a1[MF];
      RTPOSE_PW_ALOAD(a0, 0);
      if (ng == 4) {
        const int gg = 4 * c;
        RTPOSE_PW_BLOAD(bnxt, gg + 1);
        RTPOSE_PW_ALOAD(a1, 1);
        RTPOSE_PW_PIN();
        RTPOSE_PW_MUL(a0, bcur);
        RTPOSE_PW_PIN();
        RTPOSE_PW_BLOAD(bcur, gg + 2);
        RTPOSE_PW_ALOAD(a0, 2);
        RTPOSE_PW_PIN();
        RTPOSE_PW_MUL(a1, bnxt);
        RTPOSE_PW_PIN();
        RTPOSE_PW_BLOAD(bnxt, gg + 3);
        RTPOSE_PW_ALOAD(a1, 3);
        RTPOSE_PW_PIN();
        RTPOSE_PW_MUL(a0, bcur);
        RTPOSE_PW_PIN();
        RTPOSE_PW_BLOAD(bcur, min(gg + 4, gtot - 1));  // first group of the next chunk (clamped: valid memory)
        RTPOSE_PW_PIN();
        RTPOSE_PW_MUL(a1, bnxt);
        RTPOSE_PW_PIN();
      } else {  // the short last chunk of a K that is not a multiple of 32 (K = 24, 120, 232, 464)
        for (int gi = 0; gi < ng; ++gi) {
          if (gi + 1 < ng) {
            RTPOSE_PW_BLOAD(bnxt, 4 * c + gi + 1);
            RTPOSE_PW_ALOAD(a1, gi + 1);
          }
          RTPOSE_PW_PIN();
          RTPOSE_PW_MUL(a0, bcur);
          RTPOSE_PW_PIN();
#pragma unroll
          for (int fn = 0; fn < NFW; ++fn) bcur[fn] = bnxt[fn];
#pragma unroll
          for (int fm = 0; fm < MF; ++fm) a0[fm] = a1[fm];
        }
      }
#undef RTPOSE_PW_MUL
#undef RTPOSE_PW_ALOAD
#undef RTPOSE_PW_BLOAD
    }

    // ---- epilogue: (ReLU), scatter through out_cmap.  A lane holds column ncol of rows
    //      rg*8 + 4*kh + rr of its 32-row fragment (v_mfma_f32_32x32x2_f32 C layout). ----
    int qrow[MF][16];  // output pixel of every row this lane holds (batched LDS reads)
#pragma unroll
    for (int fm = 0; fm < MF; ++fm)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) qrow[fm][rg * 4 + rr] = s_qout[wm * (32 * MF) + fm * 32 + rg * 8 + 4 * kh + rr];
#pragma unroll
    for (int fn = 0; fn < NFW; ++fn) {
      const int n = ncol + fn * 32;
      int ch = -1;
      if (n < A.cout) ch = A.out_cmap ? A.out_cmap[n] : A.out_choff + n;
      float* ocol = A.out + (ch >= 0 ? ch : 0);
#pragma unroll
      for (int fm = 0; fm < MF; ++fm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qo = qrow[fm][r];
          float v = acc[fm][fn][r];
          if (A.relu) v = fmaxf(v, 0.f);
          if (qo >= 0 && ch >= 0) ocol[(size_t)qo * A.out_cstride] = v;
        }
    }
    __syncthreads();  // the LDS buffers are re-filled by the next pass
  }
}
#undef RTPOSE_PW_PIN

// packed[c/4][coutp][4] columns [col_off, col_off + cout)  <-  w[cout][cin_src] (1x1), bias likewise:
// several layers may share one packed matrix (the PAF and heat-map heads are one 128-column GEMM)
__global__ void pack_pw_kernel(const float* __restrict__ w, const float* __restrict__ bias, int cout,
                               int cin_src, const int32_t* __restrict__ cin_map, int K, int coutp, int col_off,
                               float* __restrict__ wp, float* __restrict__ bp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cout) bp[col_off + i] = bias ? bias[i] : 0.f;
  if (i >= K * cout) return;
  const int n = i % cout, c = i / cout;
  const int src = cin_map ? cin_map[c] : (c < cin_src ? c : -1);
  const float v = (src >= 0 && src < cin_src) ? w[(size_t)n * cin_src + src] : 0.f;
  wp[((size_t)(c >> 2) * coutp + col_off + n) * 4 + (c & 3)] = v;
}

int pack_pw_launch(const float* w, const float* bias, int cout, int cin_src, const int32_t* cin_map, int K,
                   int coutp, int col_off, float* wp, float* bp, hipStream_t s) {
  if (!w || !wp || !bp || cout <= 0 || K <= 0 || (K % 8) || col_off < 0 || col_off + cout > coutp)
    return fail(RTPOSE_E_INVAL, "pack_pw: bad arguments");
  hipLaunchKernelGGL(pack_pw_kernel, dim3(ceil_div(K * cout, 256)), dim3(256), 0, s, w, bias, cout, cin_src,
                     cin_map, K, coutp, col_off, wp, bp);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

// LDS plane stride of the staged halo of a 64-pixel strip: the strip, the row gaps it crosses, at most
// one image gap, one row above and below (+1 pixel each side); rounded so that the 8 planes sit 8
// banks apart like the A tile's
int pw_halo_stride(const rtpose_layout& l, int H, int W) {
  int np = (kPwBM - 1) + ((kPwBM - 1) / W + 1) * (l.ws - W) + ((kPwBM - 1) / (H * W) + 1) * (l.hs - H) * l.ws +
           2 * l.ws + 3;
  while ((np & 3) != 2) ++np;
  return np;
}

template <int WM, int MF, int NFW, bool DW>
static int pw_launch_inst(const PwArgs& a, int grid, size_t lds, hipStream_t s) {
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  auto kern = pw_gemm_f32<WM, MF, NFW, DW>;
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_set.set(dev);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int pw_fused_launch(const rtpose_pw_desc* d, int N, int H, int W, hipStream_t s) {
  if (!d || !d->in || !d->w_packed || !d->bias_packed || !d->out) return fail(RTPOSE_E_INVAL, "pw_fused: NULL argument");
  if (N <= 0 || H <= 0 || W <= 0) return fail(RTPOSE_E_INVAL, "pw_fused: empty tensor");
  if (d->cin <= 0 || (d->cin % 8)) return fail(RTPOSE_E_INVAL, "pw_fused: cin must be a multiple of 8");
  if (d->coutp != 64 && d->coutp != 128 && (d->coutp % 256)) return fail(RTPOSE_E_INVAL, "pw_fused: coutp must be 64, 128 or a multiple of 256");
  if (d->cout <= 0 || d->cout > d->coutp) return fail(RTPOSE_E_INVAL, "pw_fused: cout exceeds coutp");
  if ((d->lin.cstride % 4) || (d->lin.choff % 4) || d->lin.choff + d->cin > d->lin.cstride)
    return fail(RTPOSE_E_INVAL, "pw_fused: input slice must be 16-byte aligned and inside the pixel");
  const bool dw = d->dw_w != nullptr;
  if (dw && (!d->dw_b || d->lin.ws < W + 1 || d->lin.hs < H + 1 || d->lin.lead < d->lin.ws + 1))
    return fail(RTPOSE_E_INVAL, "pw_fused: the depthwise input needs a layout gap of 1 and a bias");
  if (d->pt_src && (!d->pt_cmap || d->pt_c <= 0 || (d->lpt.cstride % 4) || (d->lpt.choff % 4)))
    return fail(RTPOSE_E_INVAL, "pw_fused: bad pass-through description");
  PwArgs a;
  memset(&a, 0, sizeof(a));
  a.in = PwView{d->in, d->lin.cstride, d->lin.choff, d->lin.ws, d->lin.hs, d->lin.lead};
  a.dw_w = d->dw_w;
  a.dw_b = d->dw_b;
  a.w = d->w_packed;
  a.bias = d->bias_packed;
  a.out = d->out;
  a.out_cstride = d->lout.cstride;
  a.out_choff = d->lout.choff;
  a.out_ws = d->lout.ws;
  a.out_hs = d->lout.hs;
  a.out_lead = d->lout.lead;
  a.out_cmap = d->out_cmap;
  if (d->pt_src) {
    a.pt = PwView{d->pt_src, d->lpt.cstride, d->lpt.choff, d->lpt.ws, d->lpt.hs, d->lpt.lead};
    a.pt_cmap = d->pt_cmap;
    a.pt_c = d->pt_c;
  }
  a.N = N;
  a.H = H;
  a.W = W;
  a.M = N * H * W;
  a.K = d->cin;
  a.coutp = d->coutp;
  a.cout = d->cout;
  a.relu = d->relu;
  size_t lds = (size_t)2 * kPwPL * kPwQS * 16;
  if (dw) {
    a.nps = pw_halo_stride(d->lin, H, W);
    if (a.nps * kPwPL > 256 * kPwMaxStage)
      return fail(RTPOSE_E_INVAL, "pw_fused: map too wide for the fused depthwise halo (W <= ~60)");
    lds += (size_t)2 * kPwPL * a.nps * 16;
  }
  const int grid = ceil_div(a.M, kPwBM);
  if (d->coutp == 64) return dw ? pw_launch_inst<2, 1, 1, true>(a, grid, lds, s) : pw_launch_inst<2, 1, 1, false>(a, grid, lds, s);
  if (d->coutp == 128) return dw ? pw_launch_inst<1, 2, 1, true>(a, grid, lds, s) : pw_launch_inst<1, 2, 1, false>(a, grid, lds, s);
  return dw ? pw_launch_inst<1, 2, 2, true>(a, grid, lds, s) : pw_launch_inst<1, 2, 2, false>(a, grid, lds, s);
}

}  // namespace rtpose

using namespace rtpose;

extern "C" {

size_t rtpose_packed_pw_floats(int cin_packed, int coutp) { return (size_t)cin_packed * coutp; }

int rtpose_pack_pw_weights(const float* w_oi, const float* bias, int cout, int cin_src, const int32_t* cin_map,
                           int cin_packed, int coutp, int col_off, float* w_packed, float* bias_packed,
                           void* stream) {
  return pack_pw_launch(w_oi, bias, cout, cin_src, cin_map, cin_packed, coutp, col_off, w_packed, bias_packed,
                        as_stream(stream));
}

int rtpose_pw_fused(const rtpose_pw_desc* d, int N, int H, int W, void* stream) {
  return pw_fused_launch(d, N, H, W, as_stream(stream));
}

}  // extern "C"
