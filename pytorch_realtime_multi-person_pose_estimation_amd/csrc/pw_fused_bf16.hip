// bf16 form of the fused pointwise chain of pw_fused.hip (BASELINE configs[3] "fp32 and bf16"):
// bf16 activations and pointwise weights, exact products, fp32 accumulation
// (v_mfma_f32_32x32x16_bf16), depthwise taps / biases in fp32, every activation that feeds another
// conv rounded to bf16 (round-to-nearest-even), the two heads written in fp32.  The contract is
// oracle/shufflenet_oracle.py:forward_bf16_emulated (the reference has no reduced-precision path).
//
// Same structure as the fp32 kernel (persistent 64-pixel strips x the whole N, K walked through a
// double-buffered LDS tile, depthwise 3x3 evaluated in LDS in front of the GEMM, the pass-through half
// copied by the same blocks, four-run channel layout), re-balanced for a matrix pipe that is 16x
// faster per byte - the kernel is a data-movement kernel with a GEMM inside:
//  * a 16-byte piece = 8 channels = one lane's share of a K = 16 MFMA step: a chunk of 8 planes is 64
//    channels = 4 K-steps of (MF x NFW) 32-cycle MFMAs;
//  * the epilogue transposes each wave's 64 x (32 NFW) tile through LDS and stores 16 bytes (8
//    channels of a pixel) per lane.  That needs the GEMM's columns in physical channel order: the
//    host packs the weights with a column map (a run of the four-run layout per 8-aligned group).
#include <hip/hip_runtime.h>

#include "common.h"

namespace rtpose {

namespace pwb {

typedef float pwb_f2 __attribute__((ext_vector_type(2)));

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const floatx4 __attribute__((address_space(1)))* gcf4_t;
typedef floatx4 __attribute__((address_space(1)))* gf4_t;
__device__ __forceinline__ float4 gload4(const void* p) {  // explicit global address space (no FLAT loads)
  const floatx4 v = *(gcf4_t)(unsigned long long)(p);
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void gstore4(void* p, const float4& v) {
  const floatx4 t = {v.x, v.y, v.z, v.w};
  *(gf4_t)(unsigned long long)(p) = t;
}
__device__ __forceinline__ bf16x8 as_bf8(const float4& v) {
  const floatx4 t = {v.x, v.y, v.z, v.w};
  return __builtin_bit_cast(bf16x8, t);
}
__device__ __forceinline__ unsigned short to_bf16(float v) {
  return __builtin_bit_cast(unsigned short, (__bf16)v);  // v_cvt_pk_bf16_f32: RNE
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  return (unsigned)to_bf16(lo) | ((unsigned)to_bf16(hi) << 16);
}
// 8 bf16 (one 16-byte piece) -> 8 floats
__device__ __forceinline__ void unpack8(const float4& p, float* f) {
  const unsigned u[4] = {__float_as_uint(p.x), __float_as_uint(p.y), __float_as_uint(p.z), __float_as_uint(p.w)};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(u[i] << 16);
    f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
  }
}

struct View {  // one activation tensor (shared-gap padded NHWC slice), channel counts in ELEMENTS
  const unsigned short* base;
  int cstride, choff, ws, hs, lead;
};

struct Args {
  View in;            // A source: the GEMM input (DW = 0) or the depthwise conv's input (DW = 1, gap >= 1)
  const float* dw_w;  // DW: fp32 [9][K] taps and
  const float* dw_b;  //     fp32 [K] bias of the depthwise conv (BN folded)
  const float4* w;    // packed bf16 pointwise weights [K/8][coutp][8]
  const float* bias;  // fp32 [coutp]
  void* out;          // bf16 (vec epilogue: columns [0, cout) -> channels out_choff ..) or fp32 (out_f32)
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  const int32_t* out_cmap;  // out_f32 only: column n -> absolute channel (< 0: not stored)
  int out_f32;
  View pt;  // pass-through source (interleave form, see pw_fused.hip)
  int pt_pairs, pt_a, pt_b, pt_split, pt_d0, pt_d1;
  const int32_t* in_planes;  // optional [K/8]: element offset (inside the pixel) of every 8-channel plane of K
  int N, H, W, M;
  int K, coutp, cout, relu;
  int nps;  // DW: LDS plane stride (pixels) of the staged halo
  FastDiv fHW, fW, fnp, ftx, fty;  // divisions by launch constants (see pw_fused.hip)
};

constexpr int kBM = 64;
constexpr int kQS = 66;
constexpr int kPL = 8;          // 16-byte planes per K chunk (64 channels)
constexpr int kMaxK = 1024;
constexpr int kMaxStage = 4;   // DW: staged pieces per thread per chunk (10 x 10 halo of an 8 x 8 tile)
constexpr int kTile = 8;       // DW: the block's 64 pixels are an 8 x 8 tile (see pw_fused.hip)
constexpr int kHalo = kTile + 2;

#define RTPOSE_PWB_PIN()         \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)

struct Pix {  // pixel m of the batch as (image, row, column)
  int n, y, x;
};
__device__ __forceinline__ Pix pix_of(int m, int HW, int W, const FastDiv fHW, const FastDiv fW) {
  Pix p;
  p.n = fast_div(m, fHW);
  const int r = m - p.n * HW;
  p.y = fast_div(r, fW);
  p.x = r - p.y * W;
  return p;
}
__device__ __forceinline__ int pix_q(const Pix& p, int lead, int hs, int ws) { return lead + (p.n * hs + p.y) * ws + p.x; }

template <int WM, int MF, int NFW, bool DW>
__global__ __launch_bounds__(256, 2) void pw_gemm_bf16(const Args A) {
  constexpr int WN = 4 / WM;
  constexpr int BN = WN * NFW * 32;
  static_assert(WM * MF * 32 == kBM, "block tile is 64 pixels");
  extern __shared__ __attribute__((aligned(16))) float4 smem4[];
  __shared__ int s_qout[2][kBM], s_qpt[2][kBM];
  __shared__ int s_plane[kMaxK / 8];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int l31 = lane & 31, kh = lane >> 5;
  const int HW = A.H * A.W;
  const int pl = tid & 7, px = tid >> 3;

  // ---- LDS carve-up (float4 units): A chunk buffers | staged halo OR epilogue slabs | depthwise taps ----
  constexpr int ASUB = kPL * kQS;
  constexpr int SLAB_PITCH = (32 * NFW + 8) / 8;      // float4 per slab row: the wave's columns + 16 bytes of skew
  constexpr int SLAB4 = 32 * MF * SLAB_PITCH;         // per wave
  float4* a_lds = smem4;                              // [2][ASUB]
  float4* st = smem4 + 2 * ASUB;                      // DW: [kPL][nps]; epilogue: [4 waves][SLAB4]
  const int nps = A.nps;
  const int st4 = DW ? max(kPL * nps, 4 * SLAB4) : 4 * SLAB4;
  float* dwl = reinterpret_cast<float*>(st + st4);    // DW: fp32 [10][K]

  const int nch = (A.K + 63) >> 6;
  const int kst = A.K >> 4;  // 16-channel K-steps in all
  const int npass = A.coutp / BN;
  const int tiles_x = (A.W + kTile - 1) / kTile, tiles_y = (A.H + kTile - 1) / kTile;  // DW: 8 x 8 tiles of one image
  const int nwork = (DW ? A.N * tiles_y * tiles_x : (A.M + kBM - 1) / kBM) * npass;
  const float4* in4 = reinterpret_cast<const float4*>(A.in.base);
  const unsigned in_cs4 = (unsigned)A.in.cstride >> 3, in_co4 = (unsigned)A.in.choff >> 3;

  struct Item {
    int m0, pass;    // DW = 0: first pixel of the strip
    int n, y0, x0;   // DW = 1: image and first pixel of the 8 x 8 tile
    int q0, q1;      // DW = 0: the thread's two pixels;  DW = 1: q0 = pixel index of the halo's corner (y0-1, x0-1)
  };
  auto setup = [&](int wi) -> Item {
    Item it;
    const int tile = fast_div(wi, A.fnp);
    it.pass = wi - tile * npass;
    it.m0 = tile * kBM;
    it.n = it.y0 = it.x0 = 0;
    if (DW) {
      const int r = fast_div(tile, A.ftx), tx = tile - r * tiles_x;
      it.n = fast_div(r, A.fty);
      const int ty = r - it.n * tiles_y;
      it.y0 = ty * kTile;
      it.x0 = tx * kTile;
      it.q0 = A.in.lead + (it.n * A.in.hs + it.y0 - 1) * A.in.ws + it.x0 - 1;  // >= 0: lead = ws + 1
      it.q1 = 0;
    } else {
      const int ma = min(it.m0 + px, A.M - 1), mb = min(it.m0 + px + 32, A.M - 1);  // rows past the end replay the last pixel
      it.q0 = pix_q(pix_of(ma, HW, A.W, A.fHW, A.fW), A.in.lead, A.in.hs, A.in.ws);
      it.q1 = pix_q(pix_of(mb, HW, A.W, A.fHW, A.fW), A.in.lead, A.in.hs, A.in.ws);
    }
    return it;
  };
  auto write_tables = [&](const Item& it, int par) {  // threads 0..63: output / pass-through pixel of tile row tid
    if (DW) {
      const int y = it.y0 + (tid >> 3), x = it.x0 + (tid & 7);
      const bool ok = y < A.H && x < A.W;
      const int yc = min(y, A.H - 1), xc = min(x, A.W - 1);
      s_qout[par][tid] = ok ? A.out_lead + (it.n * A.out_hs + yc) * A.out_ws + xc : -1;
      s_qpt[par][tid] = A.pt.base ? A.pt.lead + (it.n * A.pt.hs + yc) * A.pt.ws + xc : 0;
    } else {
      const int m = it.m0 + tid;
      const int mc = min(m, A.M - 1);
      const Pix pp = pix_of(mc, HW, A.W, A.fHW, A.fW);
      s_qout[par][tid] = m < A.M ? pix_q(pp, A.out_lead, A.out_hs, A.out_ws) : -1;
      s_qpt[par][tid] = A.pt.base ? pix_q(pp, A.pt.lead, A.pt.hs, A.pt.ws) : 0;
    }
  };
  // DW: halo pixel (hy, hx) of the tile, hp = 10 hy + hx, sits hy * ws + hx pixels after the corner: the
  // thread's staged pieces are hp = px + 32 u (clamped to 99) - the same offsets for every tile
  int hoff[DW ? kMaxStage : 1];
  if (DW) {
#pragma unroll
    for (int u = 0; u < kMaxStage; ++u) {
      const int hp = min(px + 32 * u, kHalo * kHalo - 1);
      hoff[u] = (hp / kHalo) * A.in.ws + hp % kHalo;
    }
  }
  // the thread's two depthwise output pixels are tile rows px and px + 32: halo index of their tap (0, 0)
  const int sp0 = (px >> 3) * kHalo + (px & 7), sp1 = sp0 + 4 * kHalo;
  // staged pieces of channel chunk c0 (a multiple of 64) of item `it` -> registers; branch-free, clamped
  float4 sr[DW ? kMaxStage : 2];
  auto load_pieces = [&](const Item& it, int c0) {
    const unsigned cofs = in_co4 + ((unsigned)s_plane[min((c0 >> 3) + pl, (A.K >> 3) - 1)] >> 3);
    if (DW) {
#pragma unroll
      for (int u = 0; u < kMaxStage; ++u)
        sr[u] = gload4(in4 + ((unsigned)(it.q0 + hoff[u]) * in_cs4 + cofs));
    } else {
      sr[0] = gload4(in4 + ((unsigned)it.q0 * in_cs4 + cofs));
      sr[1] = gload4(in4 + ((unsigned)it.q1 * in_cs4 + cofs));
    }
  };

  int wi = blockIdx.x;
  if (wi >= nwork) return;
  for (int j = tid; j < (A.K >> 3); j += 256) s_plane[j] = A.in_planes ? A.in_planes[j] : 8 * j;
  __syncthreads();
  Item cur = setup(wi);
#pragma unroll
  for (int u = 0; u < (DW ? kMaxStage : 2); ++u) sr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < kBM) write_tables(cur, 0);
  load_pieces(cur, 0);
  if (DW) {  // (visible after the first chunk's barrier)
    for (int i = tid; i < 10 * A.K; i += 256) dwl[i] = i < 9 * A.K ? A.dw_w[i] : A.dw_b[i - 9 * A.K];
  }
  int par = 0;
  int lbuf = 0;

  while (true) {
    const int wnext = wi + gridDim.x;
    const bool has_next = wnext < nwork;
    Item nxt = cur;
    if (has_next) nxt = setup(wnext);
    const int ncol = cur.pass * BN + wn * (32 * NFW) + l31;
    const unsigned w_lane = (unsigned)(kh * A.coutp + ncol);  // float4 index of this lane in a K-step's two planes

    float4 bcur[NFW], bnxt[NFW];
#pragma unroll
    for (int fn = 0; fn < NFW; ++fn) {
      bcur[fn] = gload4(A.w + (w_lane + fn * 32));
      bnxt[fn] = bcur[fn];
    }
    floatx16 acc[MF][NFW];
#pragma unroll
    for (int fn = 0; fn < NFW; ++fn) {
      const float b0 = A.bias[ncol + fn * 32];
#pragma unroll
      for (int fm = 0; fm < MF; ++fm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[fm][fn][r] = b0;
    }

    for (int c = 0; c < nch; ++c) {
      const int c0 = c << 6;
      float4* a_buf = a_lds + lbuf * ASUB;
      const bool last = c + 1 == nch;
      if (DW) {
#pragma unroll
        for (int u = 0; u < kMaxStage; ++u) st[pl * nps + px + 32 * u] = sr[u];  // (nps >= 128 slots)
        __syncthreads();  // halo of chunk c visible; every wave is past the previous item's epilogue slabs
        if (c == 0 && has_next && tid < kBM) write_tables(nxt, par ^ 1);
        // depthwise 3x3 (+bias) in fp32 on the 8 channels of this thread's plane, two pixels; rounded to bf16
        const int ch = min(c0 + 8 * pl, A.K - 8);  // (planes past K: their A planes are not read)
        // (one pixel at a time, one stencil row at a time: both pixels' 3 x 3 x 8 values in flight cost
        //  more registers than the 256-column variant has)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          pwb_f2 v[4];  // 8 channels as 4 packed pairs (v_pk_fma_f32: 4 instead of 16 VALU instructions per tap)
          {
            const float4 b0 = *reinterpret_cast<const float4*>(dwl + 9 * A.K + ch);
            const float4 b1 = *reinterpret_cast<const float4*>(dwl + 9 * A.K + ch + 4);
            v[0] = pwb_f2{b0.x, b0.y}, v[1] = pwb_f2{b0.z, b0.w}, v[2] = pwb_f2{b1.x, b1.y}, v[3] = pwb_f2{b1.z, b1.w};
          }
          const float4* s0 = st + pl * nps + (u ? sp1 : sp0);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const float4 w0 = *reinterpret_cast<const float4*>(dwl + (ky * 3 + kx) * A.K + ch);
              const float4 w1 = *reinterpret_cast<const float4*>(dwl + (ky * 3 + kx) * A.K + ch + 4);
              const pwb_f2 ww[4] = {{w0.x, w0.y}, {w0.z, w0.w}, {w1.x, w1.y}, {w1.z, w1.w}};
              float x[8];
              unpack8(s0[ky * kHalo + kx], x);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = __builtin_elementwise_fma(pwb_f2{x[2 * e], x[2 * e + 1]}, ww[e], v[e]);
            }
            RTPOSE_PWB_PIN();
          }
          a_buf[pl * kQS + px + 32 * u] =
              make_float4(__uint_as_float(pack2(v[0].x, v[0].y)), __uint_as_float(pack2(v[1].x, v[1].y)),
                          __uint_as_float(pack2(v[2].x, v[2].y)), __uint_as_float(pack2(v[3].x, v[3].y)));
          RTPOSE_PWB_PIN();
        }
        // the next chunk's halo (of this item, or chunk 0 of the next) is requested only now: the depthwise
        // phase above is the register peak of the kernel, and the MFMAs below still cover the latency
        load_pieces(last ? nxt : cur, last ? 0 : c0 + 64);
        __syncthreads();  // A tile of chunk c visible; the staged halo may be overwritten
      } else {
        a_buf[pl * kQS + px] = sr[0];
        a_buf[pl * kQS + px + 32] = sr[1];
        load_pieces(last ? nxt : cur, last ? 0 : c0 + 64);
        __syncthreads();  // A tile of chunk c visible
        if (c == 0 && has_next && tid < kBM) write_tables(nxt, par ^ 1);
      }
      lbuf ^= 1;

      // ---- multiply chunk c: ng K-steps of 16 channels, one MFMA per (m, n) fragment pair each; the operands
      //      of the next K-step are requested before the current one is multiplied (straight-line code) ----
      const int ng = min(4, kst - 4 * c);
      const float4* a_rd = a_buf + kh * kQS + wm * (32 * MF) + l31;
#define RTPOSE_PWB_BLOAD(DST, GG)                                                               \
  _Pragma("unroll") for (int fn = 0; fn < NFW; ++fn)                                            \
      DST[fn] = gload4(A.w + (size_t)(2 * (GG) * A.coutp) + (w_lane + fn * 32))
#define RTPOSE_PWB_ALOAD(DST, GI) \
  _Pragma("unroll") for (int fm = 0; fm < MF; ++fm) DST[fm] = a_rd[2 * (GI) * kQS + fm * 32]
#define RTPOSE_PWB_MUL(AV, BV)                                                                       \
  _Pragma("unroll") for (int fn = 0; fn < NFW; ++fn) {                                               \
    _Pragma("unroll") for (int fm = 0; fm < MF; ++fm)                                                \
        acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(AV[fm]), as_bf8(BV[fn]), acc[fm][fn], 0, 0, 0); \
  }
      const int gg = 4 * c, gl = kst - 1;
      float4 a0[MF], a1[MF];
      RTPOSE_PWB_ALOAD(a0, 0);
      RTPOSE_PWB_BLOAD(bnxt, min(gg + 1, gl));
      RTPOSE_PWB_ALOAD(a1, 1);
      RTPOSE_PWB_PIN();
      RTPOSE_PWB_MUL(a0, bcur);
      RTPOSE_PWB_PIN();
      if (ng > 1) {
        RTPOSE_PWB_BLOAD(bcur, min(gg + 2, gl));
        RTPOSE_PWB_ALOAD(a0, 2);
        RTPOSE_PWB_PIN();
        RTPOSE_PWB_MUL(a1, bnxt);
        RTPOSE_PWB_PIN();
      }
      if (ng > 2) {
        RTPOSE_PWB_BLOAD(bnxt, min(gg + 3, gl));
        RTPOSE_PWB_ALOAD(a1, 3);
        RTPOSE_PWB_PIN();
        RTPOSE_PWB_MUL(a0, bcur);
        RTPOSE_PWB_PIN();
      }
      if (ng > 3) {
        RTPOSE_PWB_BLOAD(bcur, min(gg + 4, gl));  // first K-step of the next chunk
        RTPOSE_PWB_PIN();
        RTPOSE_PWB_MUL(a1, bnxt);
        RTPOSE_PWB_PIN();
      }
#undef RTPOSE_PWB_MUL
#undef RTPOSE_PWB_ALOAD
#undef RTPOSE_PWB_BLOAD
    }

    // ---- epilogue ------------------------------------------------------------------------------------
    if (A.out_f32) {
      // the two heads: fp32, scattered through out_cmap (57 real columns)
      float* outf = static_cast<float*>(A.out);
#pragma unroll
      for (int fn = 0; fn < NFW; ++fn) {
        const int n = ncol + fn * 32;
        int ch = -1;
        if (n < A.cout) ch = A.out_cmap ? A.out_cmap[n] : A.out_choff + n;
#pragma unroll
        for (int fm = 0; fm < MF; ++fm)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int qo = s_qout[par][wm * (32 * MF) + fm * 32 + rg * 8 + 4 * kh + rr];
              float v = acc[fm][fn][rg * 4 + rr];
              if (A.relu) v = fmaxf(v, 0.f);
              if (qo >= 0 && ch >= 0) outf[(unsigned)qo * (unsigned)A.out_cstride + (unsigned)ch] = v;
            }
      }
    } else {
      // bf16: the wave's (32 MF) x (32 NFW) tile -> its LDS slab [row][col] -> 16 bytes (8 channels of a pixel)
      // per lane.  (The chunk loop's last barrier is behind every wave: the staged-halo area is free.)
      unsigned short* sl = reinterpret_cast<unsigned short*>(st + wave * SLAB4);
      constexpr int PITCH2 = SLAB_PITCH * 8;  // slab row pitch in bf16 elements
#pragma unroll
      for (int fn = 0; fn < NFW; ++fn)
#pragma unroll
        for (int fm = 0; fm < MF; ++fm)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              float v = acc[fm][fn][rg * 4 + rr];
              if (A.relu) v = fmaxf(v, 0.f);
              sl[(fm * 32 + rg * 8 + 4 * kh + rr) * PITCH2 + fn * 32 + l31] = to_bf16(v);
            }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // same-wave LDS traffic is in order
      constexpr int LPR = NFW * 4;    // 16-byte lanes per slab row
      constexpr int RPI = 64 / LPR;   // rows per wave-instruction
      const int lrow = lane / LPR, c16 = lane % LPR;
      const int col0 = cur.pass * BN + wn * (32 * NFW) + c16 * 8;  // first of this lane's 8 columns
      unsigned short* outh = static_cast<unsigned short*>(A.out);
#pragma unroll
      for (int it = 0; it < (32 * MF) / RPI; ++it) {
        const int row = it * RPI + lrow;
        const float4 v = *reinterpret_cast<const float4*>(sl + row * PITCH2 + c16 * 8);
        const int qo = s_qout[par][wm * (32 * MF) + row];
        // (out_cmap: the 8 columns from col0 go to the 8 channels from out_cmap[col0] - absolute, 8-aligned, the host
        //  keeps a group's channels contiguous; < 0: the group is not stored)
        const int chb = A.out_cmap ? A.out_cmap[min(col0, A.cout - 8)] : A.out_choff + col0;
        if (qo >= 0 && col0 < A.cout && chb >= 0)
          gstore4(outh + ((unsigned)qo * (unsigned)A.out_cstride + (unsigned)chb), v);
      }
    }

    // ---- pass-through half, interleave form (see pw_fused.hip): 16-byte loads of 8 channels from the two
    //      source runs, 4-byte stores of (even, odd) pairs ------------------------------------------------
    if (A.pt.base && A.pt_pairs > 0 && cur.pass == 0) {
      const int g8 = (A.pt_pairs + 7) >> 3;
      const int nit = kBM * g8;
      const float4* pt4 = reinterpret_cast<const float4*>(A.pt.base);
      const unsigned pt_cs4 = (unsigned)A.pt.cstride >> 3;
      unsigned short* outh = static_cast<unsigned short*>(A.out);
      const bool pair_ok = !(A.pt_split & 1) && !(A.pt_d0 & 1) && !((A.pt_d1 - A.pt_split) & 1);
      for (int it0 = tid; it0 < nit; it0 += 512) {
        float4 va[2], vb[2];
        int qo[2], k0[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int it = min(it0 + 256 * u, nit - 1);
          const int p = it / g8;
          k0[u] = 8 * (it - p * g8);
          qo[u] = it0 + 256 * u < nit ? s_qout[par][p] : -1;
          const unsigned so = (unsigned)s_qpt[par][p] * pt_cs4 + ((unsigned)(A.pt.choff + k0[u]) >> 3);
          va[u] = gload4(pt4 + (so + ((unsigned)A.pt_a >> 3)));
          vb[u] = gload4(pt4 + (so + ((unsigned)A.pt_b >> 3)));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (qo[u] < 0) continue;
          const unsigned ua[4] = {__float_as_uint(va[u].x), __float_as_uint(va[u].y), __float_as_uint(va[u].z),
                                  __float_as_uint(va[u].w)};
          const unsigned ub[4] = {__float_as_uint(vb[u].x), __float_as_uint(vb[u].y), __float_as_uint(vb[u].z),
                                  __float_as_uint(vb[u].w)};
          const unsigned o = (unsigned)qo[u] * (unsigned)A.out_cstride;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int k = k0[u] + e;  // pair index: outputs j = 2k (from run a), 2k + 1 (from run b)
            if (k >= A.pt_pairs) continue;
            const unsigned ea = (e & 1) ? (ua[e >> 1] >> 16) : (ua[e >> 1] & 0xffffu);
            const unsigned eb = (e & 1) ? (ub[e >> 1] >> 16) : (ub[e >> 1] & 0xffffu);
            const int j = 2 * k;
            if (pair_ok) {
              const int dj = j < A.pt_split ? A.pt_d0 + j : A.pt_d1 + j - A.pt_split;
              *reinterpret_cast<unsigned*>(outh + (o + (unsigned)dj)) = ea | (eb << 16);
            } else {
              const int da = j < A.pt_split ? A.pt_d0 + j : A.pt_d1 + j - A.pt_split;
              const int db = j + 1 < A.pt_split ? A.pt_d0 + j + 1 : A.pt_d1 + j + 1 - A.pt_split;
              outh[o + (unsigned)da] = (unsigned short)ea;
              outh[o + (unsigned)db] = (unsigned short)eb;
            }
          }
        }
      }
    }

    if (!has_next) break;
    __syncthreads();  // the epilogue slabs share LDS with the staged halo of the next item's first chunk
    cur = nxt;
    wi = wnext;
    par ^= 1;
  }
}
#undef RTPOSE_PWB_PIN

// packed[c/8][coutp][8 bf16], bias[coutp]  <-  w[cout][cin_src] (1x1, fp32), bias.  Packed input channel c reads
// source channel cin_map[c] (NULL: c, < 0: zero); packed column n (col_off + i) is output channel col_map[i]
// (NULL: i; < 0: a zero column) - the bf16 epilogue stores columns as contiguous channels, so a layer that
// writes runs of the four-run layout gets its columns in run order.
__global__ void pack_pw_bf16_kernel(const float* __restrict__ w, const float* __restrict__ bias, int cout,
                                    int cin_src, const int32_t* __restrict__ cin_map, int K, int ncols,
                                    const int32_t* __restrict__ col_map, int coutp, int col_off,
                                    unsigned short* __restrict__ wp, float* __restrict__ bp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ncols) {
    const int o = col_map ? col_map[i] : i;
    bp[col_off + i] = (o >= 0 && o < cout && bias) ? bias[o] : 0.f;
  }
  if (i >= K * ncols) return;
  const int n = i % ncols, c = i / ncols;
  const int o = col_map ? col_map[n] : n;
  const int src = cin_map ? cin_map[c] : (c < cin_src ? c : -1);
  const float v = (o >= 0 && o < cout && src >= 0 && src < cin_src) ? w[(size_t)o * cin_src + src] : 0.f;
  wp[((size_t)(c >> 3) * coutp + col_off + n) * 8 + (c & 7)] = to_bf16(v);
}

int halo_stride(const rtpose_layout& l, int H, int W) {
  int np = (kBM - 1) + ((kBM - 1) / W + 1) * (l.ws - W) + ((kBM - 1) / (H * W) + 1) * (l.hs - H) * l.ws + 2 * l.ws + 3;
  while ((np & 3) != 2) ++np;
  return np;
}

template <int WM, int MF, int NFW, bool DW>
static int launch_inst(const Args& a, int grid, size_t lds, hipStream_t s) {
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  auto kern = pw_gemm_bf16<WM, MF, NFW, DW>;
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024));
    attr_set.set(dev);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace pwb

int pack_pw_bf16_launch(const float* w, const float* bias, int cout, int cin_src, const int32_t* cin_map, int K,
                        int ncols, const int32_t* col_map, int coutp, int col_off, void* wp, float* bp,
                        hipStream_t s) {
  if (!w || !wp || !bp || cout <= 0 || K <= 0 || (K % 16) || ncols <= 0 || col_off < 0 || col_off + ncols > coutp)
    return fail(RTPOSE_E_INVAL, "pack_pw_bf16: bad arguments");
  hipLaunchKernelGGL(pwb::pack_pw_bf16_kernel, dim3(ceil_div(K * ncols, 256)), dim3(256), 0, s, w, bias, cout,
                     cin_src, cin_map, K, ncols, col_map, coutp, col_off, static_cast<unsigned short*>(wp), bp);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int pw_halo_stride_bf16(const rtpose_layout& l, int H, int W) { return pwb::halo_stride(l, H, W); }

// d->in / out / pt_src point at bf16 elements (out: fp32 when out_f32), layouts count ELEMENTS; d->w_packed from
// pack_pw_bf16_launch; out_cmap is used by the fp32 (heads) epilogue only; pass-through = interleave form only.
int pw_fused_bf16_launch(const rtpose_pw_desc* d, int out_f32, int N, int H, int W, hipStream_t s) {
  using namespace pwb;
  if (!d || !d->in || !d->w_packed || !d->bias_packed || !d->out) return fail(RTPOSE_E_INVAL, "pw_fused_bf16: NULL argument");
  if (N <= 0 || H <= 0 || W <= 0) return fail(RTPOSE_E_INVAL, "pw_fused_bf16: empty tensor");
  if (d->cin <= 0 || (d->cin % 16) || d->cin > kMaxK) return fail(RTPOSE_E_INVAL, "pw_fused_bf16: cin must be a multiple of 16, <= 1024");
  if (d->coutp != 64 && d->coutp != 128 && (d->coutp % 256)) return fail(RTPOSE_E_INVAL, "pw_fused_bf16: coutp must be 64, 128 or a multiple of 256");
  if (d->cout <= 0 || d->cout > d->coutp) return fail(RTPOSE_E_INVAL, "pw_fused_bf16: cout exceeds coutp");
  if ((d->lin.cstride % 8) || (d->lin.choff % 8) || (!d->in_planes && d->lin.choff + d->cin > d->lin.cstride))
    return fail(RTPOSE_E_INVAL, "pw_fused_bf16: input slice must be 16-byte aligned and inside the pixel");
  if (!out_f32 && ((d->lout.cstride % 8) || (d->lout.choff % 8) || (d->cout % 8) || d->lout.choff + d->cout > d->lout.cstride))
    return fail(RTPOSE_E_INVAL, "pw_fused_bf16: bf16 output = whole 8-channel groups inside the pixel");
  const bool dw = d->dw_w != nullptr;
  if (dw && (!d->dw_b || d->lin.ws < W + 1 || d->lin.hs < H + 1 || d->lin.lead < d->lin.ws + 1))
    return fail(RTPOSE_E_INVAL, "pw_fused_bf16: the depthwise input needs a layout gap of 1 and a bias");
  if (d->pt_src && (d->pt_pairs <= 0 || (d->lpt.cstride % 8) || ((d->lpt.choff + d->pt_a) % 8) || ((d->lpt.choff + d->pt_b) % 8)))
    return fail(RTPOSE_E_INVAL, "pw_fused_bf16: pass-through runs must be 16-byte aligned (interleave form only)");
  if (rtpose_layout_pixels(&d->lin, N, H, W) * (size_t)d->lin.cstride >= ((size_t)1 << 31) ||
      rtpose_layout_pixels(&d->lout, N, H, W) * (size_t)d->lout.cstride >= ((size_t)1 << 31))
    return fail(RTPOSE_E_INVAL, "pw_fused_bf16: tensors must be below 2^31 elements");
  Args a;
  memset(&a, 0, sizeof(a));
  a.in = View{reinterpret_cast<const unsigned short*>(d->in), d->lin.cstride, d->lin.choff, d->lin.ws, d->lin.hs, d->lin.lead};
  a.dw_w = d->dw_w;
  a.dw_b = d->dw_b;
  a.w = reinterpret_cast<const float4*>(d->w_packed);
  a.bias = d->bias_packed;
  a.out = d->out;
  a.out_cstride = d->lout.cstride;
  a.out_choff = d->lout.choff;
  a.out_ws = d->lout.ws;
  a.out_hs = d->lout.hs;
  a.out_lead = d->lout.lead;
  a.out_cmap = d->out_cmap;
  a.out_f32 = out_f32 ? 1 : 0;
  if (d->pt_src) {
    a.pt = View{reinterpret_cast<const unsigned short*>(d->pt_src), d->lpt.cstride, d->lpt.choff, d->lpt.ws, d->lpt.hs, d->lpt.lead};
    a.pt_pairs = d->pt_pairs;
    a.pt_a = d->pt_a;
    a.pt_b = d->pt_b;
    a.pt_split = d->pt_split;
    a.pt_d0 = d->pt_d0;
    a.pt_d1 = d->pt_d1;
  }
  a.in_planes = d->in_planes;
  a.N = N;
  a.H = H;
  a.W = W;
  a.M = N * H * W;
  a.K = d->cin;
  a.coutp = d->coutp;
  a.cout = d->cout;
  a.relu = d->relu;
  const int nfw = d->coutp >= 256 ? 2 : 1, mf = d->coutp == 64 ? 1 : 2;
  const size_t slab4 = (size_t)4 * 32 * mf * ((32 * nfw + 8) / 8);
  size_t st4 = slab4;
  if (dw) {
    a.nps = 32 * kMaxStage + 2;
    if ((size_t)kPL * a.nps > st4) st4 = (size_t)kPL * a.nps;
  }
  const size_t lds = ((size_t)2 * kPL * kQS + st4) * 16 + (dw ? (size_t)10 * d->cin * 4 : 0);
  const int npass_h = d->coutp <= 128 ? 1 : d->coutp / 256;
  a.fHW = make_fastdiv(H * W);
  a.fW = make_fastdiv(W);
  a.fnp = make_fastdiv(npass_h);
  a.ftx = make_fastdiv(ceil_div(W, kTile));
  a.fty = make_fastdiv(ceil_div(H, kTile));
  const int nwork = (dw ? N * ceil_div(H, kTile) * ceil_div(W, kTile) : ceil_div(a.M, kBM)) * npass_h;
  const int grid = nwork < 2 * device_cu_count() ? nwork : 2 * device_cu_count();
  if (d->coutp == 64) return dw ? launch_inst<2, 1, 1, true>(a, grid, lds, s) : launch_inst<2, 1, 1, false>(a, grid, lds, s);
  if (d->coutp == 128) return dw ? launch_inst<1, 2, 1, true>(a, grid, lds, s) : launch_inst<1, 2, 1, false>(a, grid, lds, s);
  return dw ? launch_inst<1, 2, 2, true>(a, grid, lds, s) : launch_inst<1, 2, 2, false>(a, grid, lds, s);
}

}  // namespace rtpose

using namespace rtpose;

extern "C" {

size_t rtpose_packed_pw_bytes_bf16(int cin_packed, int coutp) { return (size_t)(cin_packed + 64) * coutp * 2; }

int rtpose_pack_pw_weights_bf16(const float* w_oi, const float* bias, int cout, int cin_src, const int32_t* cin_map,
                                int cin_packed, int ncols, const int32_t* col_map, int coutp, int col_off,
                                void* w_packed, float* bias_packed, void* stream) {
  return pack_pw_bf16_launch(w_oi, bias, cout, cin_src, cin_map, cin_packed, ncols, col_map, coutp, col_off, w_packed,
                             bias_packed, as_stream(stream));
}

int rtpose_pw_fused_bf16(const rtpose_pw_desc* d, int out_f32, int N, int H, int W, void* stream) {
  return pw_fused_bf16_launch(d, out_f32, N, H, W, as_stream(stream));
}

}  // extern "C"
