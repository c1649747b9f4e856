// conv5 + the two heads of the ShuffleNetV2 pose network as ONE back-to-back GEMM launch - bf16 operands, fp32
// accumulation (v_mfma_f32_32x32x16_bf16), gfx950 (BASELINE configs[3], the bf16 plan).
//
// Stands in for   slim.conv_bn_relu('conv5', 464, 1024, 1)  ->  self.paf = nn.Conv2d(1024, 38, 1)
//                                                               self.heatmap = nn.Conv2d(1024, 19, 1)
// (lib/network/rtpose_shufflenetV2.py:104, :107-108, forward :143-147) under the contract of
// oracle/shufflenet_oracle.py:forward_bf16_emulated: bf16 activations and weights, exact products, fp32 sums, the
// conv5 feature rounded to bf16 (RNE) after its ReLU, the heads' maps written fp32.  As two launches (round 4) the
// 1024-channel feature made a 0.55 + 0.55 GB round trip through HBM per 128-image forward and the wide conv streamed
// its 983 KB of filters from L2 once per 64 pixels, one fragment per two MFMAs: 0.55 + 0.23 ms, 0.19 of the matrix peak.
//
// The fp32 kernel (pw_head.hip) gives every wave its own 32 pixels.  The bf16 matrix pipe is 16x faster per operand
// byte, so here the operand paths decide the shape:
//   * BLOCK ITEM = 128 pixels (four MFMA fragments), 4 waves, one block per CU.  The item's whole input tile -
//     128 pixels x K1 channels, 123 KB at K1 = 480 - is RESIDENT in LDS for the four passes over conv5's 1024 channels;
//     during the LAST pass every 64-channel chunk is overwritten, right after its last use, with the same chunk of
//     the block's NEXT item (requested a chunk earlier; in the other passes the same load instructions run on a
//     zero-extent buffer descriptor and fetch nothing - no load sits under a branch, see DESIGN.md on vmcnt).
//   * The waves split the CHANNELS: in a pass wave w owns conv5 channels [256 p + 64 w, + 64) of all 128 pixels.
//     A filter fragment (1 KB, straight from L2 through a buffer load: one SGPR offset, no address VALU) feeds FOUR
//     MFMAs, an activation fragment (one ds_read_b128 per lane) two: 32 B/clk per CU from L2, 64 B/clk from LDS at
//     the full matrix rate.  Filters are requested seven k-steps (56 MFMAs) ahead in an eight-set register ring.
//   * BOTH GEMMs ARE COMPUTED TRANSPOSED (as in pw_head.hip): C1^T[channel][pixel] = W1^T X^T leaves lane (l31, kh)
//     with the channels rg 8 + 4 kh + rr of ITS pixel - packed to bf16 after bias + ReLU, eight of them are exactly
//     the B operand of one K = 16 step of OUT^T[head column][pixel] += W2^T[column][k] relu(C1^T)[k][pixel]; the heads'
//     matrix is packed in that k order (pack_head_w2_bf16).  The 1024-channel feature exists only in registers.
//   * Each wave sums the heads over ITS 256 channels; the four partial sums of a pixel fragment meet in LDS at the
//     end of the item and are added in wave order 0..3 - fixed, so an image's maps do not depend on its batch.
// Sum order of a head value: (((wave 0 + wave 1) + wave 2) + wave 3) + bias, each wave's sum over (pass, fragment,
// 16-channel group) in MFMA order.
#include <hip/hip_runtime.h>

#include "common.h"
#include "wino_common.h"

namespace rtpose {

namespace headb {

using winoc::i32x4;
using winoc::make_rsrc;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float4 bload4(i32x4 r, unsigned voff, unsigned soff) {
  const winoc::f32x4 v = winoc::llvm_raw_buffer_load_v4f32(r, (int)voff, (int)soff, 0);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ bf16x8 as_bf8(const float4& v) {
  const floatx4 t = {v.x, v.y, v.z, v.w};
  return __builtin_bit_cast(bf16x8, t);
}
__device__ __forceinline__ unsigned short to_bf16(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
// ReLU on the bit pattern: a negative float (and -0) is a negative int, a positive one orders like its bits - one
// v_max_i32 where fmaxf costs a canonicalising v_max_f32 more (no NaN reaches here that the reference would keep)
// The compiler's hazard recogniser does not know that the asm above reads an MFMA result: the wait states between the last
// MFMA that wrote an accumulator and its first v_accvgpr_read (up to 18 for a 16-pass 32x32x16) are inserted by hand,
// once, in front of every read-out section.  (Found the hard way: registers 0 and 1 of the first fragment read stale
// values in one instance of unit_bf16_kernel whose epilogue followed the last MFMA directly.)
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
__device__ __forceinline__ float relu_bits(float v) {
  return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0));
}
// One accumulator register -> a VGPR, HERE.  The allocator otherwise moves the pass' whole C1^T tile (128 registers)
// out of the accumulator file right behind the last MFMA of GEMM 1 - the VALU of the fold cannot read AGPRs - and
// spills around it; read one at a time, a fragment's values are live for the length of its own conversion only.
__device__ __forceinline__ float acc_read(float a) {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}
// relu + RNE of two fp32 -> one dword of two bf16 (v_cvt_pk_bf16_f32)
__device__ __forceinline__ float pack_relu2(float lo, float hi) {
  const bf16x2 t = {(__bf16)relu_bits(lo), (__bf16)relu_bits(hi)};
  return __builtin_bit_cast(float, t);
}

constexpr int PXB = 128;      // pixels of a block item (four MFMA fragments)
constexpr int XP = 130;       // LDS plane pitch in float4: == 2 (mod 8), so that the 8-lane groups of a ds_write_b128
                              // (4 planes x 2 pixels) fill the 8 distinct 16-byte slots of a 128-byte bank row
constexpr int N2 = 64;        // head columns (38 + 2 + 19 + 5)
constexpr int kMaxK1 = 480;   // widest K of the wide conv: 60 planes x 130 x 16 B = 124.8 KB of the 160 KB
constexpr int kRed4 = 4 * 8 * 64;  // float4 slots of the cross-wave reduction buffer (32 KB)

struct Args {
  const unsigned short* in;  // bf16 activations
  size_t in_bytes;           // addressable from `in` (buffer descriptor range)
  int in_cstride, in_choff, in_ws, in_hs, in_lead;   // in ELEMENTS / pixels
  const int32_t* in_planes;  // optional [K1 / 8]: element offset (inside the slice) of every 8-channel plane of K1
  const void* w1;            // [K1 / 8][N1][8 bf16]   (pack_pw_bf16_launch, plain column order)
  const float* b1;           // [N1]
  const void* w2;            // [N1 / 8][64][8 bf16], k permuted (pack_head_w2_bf16)
  const float* b2;           // [64]
  float* out;
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  int N, H, W, M;
  int K1, N1, nitems;
  FastDiv fHW, fW;
};

#define RTPOSE_HB_PIN()          \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)

template <int V>
struct IntTag {
  static constexpr int value = V;
};

// A pass walks K1 in k-steps of 16 channels, four to a chunk (= one turn of the filter ring's four slots), two chunks to
// a staging granule.  K1 = 480 is 30 steps: the pass walks 32, the last two request their filter fragments through a
// zero-extent descriptor (nothing fetched) and skip their MFMAs - every pass is the same code (a separate two-step tail
// made the register allocator park the heads' 128 accumulators in VGPRs during GEMM 1 and spill).
__global__ __launch_bounds__(256, 1) void pw_head_bf16_kernel(const Args A) {
  extern __shared__ __attribute__((aligned(16))) float4 smem4[];
  __shared__ __attribute__((aligned(16))) float4 s_b1[256], s_b2[16];
  __shared__ int s_plane[kMaxK1 / 8];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int npl = A.K1 >> 3;   // 16-byte planes of K1
  const int kreal = A.K1 >> 4;          // k-steps of GEMM 1 that carry channels
  const int kst = (kreal + 7) & ~7;     // k-steps a pass runs (whole PAIRS of chunks: the staging granule)
  const int NP = A.N1 >> 8;    // passes of 256 channels
  const int nch = (npl + 7) >> 3;
  float4* const xs = smem4;                    // [npl][XP]: plane pl of pixel px at pl * XP + px
  float4* const red = smem4 + npl * XP;        // [4 waves][8][64 lanes]
  for (int j = tid; j < A.N1 / 4; j += 256) s_b1[j] = *reinterpret_cast<const float4*>(A.b1 + 4 * j);
  for (int j = tid; j < npl; j += 256) s_plane[j] = A.in_planes ? A.in_planes[j] : 8 * j;
  if (tid < 16) s_b2[tid] = *reinterpret_cast<const float4*>(A.b2 + 4 * tid);
  __syncthreads();
  int item = blockIdx.x;
  if (item >= A.nitems) return;  // (the whole block)

  const int HW = A.H * A.W;
  const i32x4 rw1 = make_rsrc(A.w1, (size_t)npl * A.N1 * 16);
  const i32x4 rw2 = make_rsrc(A.w2, (size_t)(A.N1 >> 3) * N2 * 16);
  const i32x4 rin = make_rsrc(A.in, A.in_bytes);
  const i32x4 rnull = make_rsrc(A.in, 0);
  const unsigned w1_lane = (unsigned)(kh * A.N1 + 64 * wave + l31) * 16u;  // byte offset of the lane in a k-step's two planes
  const unsigned w2_lane = (unsigned)(kh * N2 + l31) * 16u;
  const unsigned x_lane = (unsigned)(kh * XP + l31);                        // float4 index in a k-step's two LDS planes

  auto pixel_q = [&](int m, int lead, int hs, int ws) {
    const int n = fast_div(m, A.fHW), r = m - n * HW;
    const int y = fast_div(r, A.fW), x = r - y * A.W;
    return lead + (n * hs + y) * ws + x;
  };
  // staging role of a thread: plane spl of a chunk, pixels spx + 32 u.  8 consecutive lanes = 4 planes x 2 pixels
  // (conflict-free ds_write_b128), a wave = 8 planes x 8 pixels = eight whole 128-byte lines per load instruction
  const int spl = (tid & 3) | (((tid >> 3) & 1) << 2);
  const int spx = ((tid >> 2) & 1) | ((tid >> 4) << 1);
  unsigned sq[4], sqn[4];  // byte offsets of the thread's four pixels: the item in LDS / the block's next item
  auto setup_stage = [&](int it, unsigned* q) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = min(it * PXB + spx + 32 * u, A.M - 1);  // pixels past the end replay the last one (never stored)
      q[u] = ((unsigned)pixel_q(m, A.in_lead, A.in_hs, A.in_ws) * (unsigned)A.in_cstride + (unsigned)A.in_choff) * 2u;
    }
  };
  float4 sr[2][4];  // two chunks in flight: requested at the first step of a chunk pair, stored behind its eighth
  auto stage_load = [&](float4(&dst)[4], const i32x4 r, const unsigned* q, int c) {
    const unsigned pofs = (unsigned)s_plane[min(8 * c + spl, npl - 1)] * 2u;
#pragma unroll
    for (int u = 0; u < 4; ++u) dst[u] = bload4(r, q[u] + pofs, 0);
  };
  auto stage_store = [&](const float4(&src)[4], int c) {
    if (8 * c + spl < npl) {
#pragma unroll
      for (int u = 0; u < 4; ++u) xs[(8 * c + spl) * XP + spx + 32 * u] = src[u];
    }
  };

  float4 wr[8][2];   // filter fragments of GEMM 1: ring of eight k-steps x the wave's two channel fragments
  float4 xr[2][4];   // activation fragments: two k-steps x four pixel fragments
  float4 w2r[2][2];  // fragments of the heads' matrix: two 16-channel groups in flight x two head-column fragments
  floatx16 acc[2][4];  // C1^T of the pass: channel fragment f, pixel fragment j
  floatx16 o[2][4];    // OUT^T partial: head-column fragment i, pixel fragment j

  auto wload = [&](float4(&dst)[2], int p, int g) {
    // (a step past K1 reads through the zero-extent descriptor: zeros, nothing fetched)
    const i32x4 r = g < kreal ? rw1 : rnull;
    const unsigned so = ((unsigned)(2 * min(g, kreal - 1)) * (unsigned)A.N1 + (unsigned)(p * 256)) * 16u;
    dst[0] = bload4(r, w1_lane, so);
    dst[1] = bload4(r, w1_lane + 32u * 16u, so);
  };
  auto xload = [&](float4(&dst)[4], int g) {
    const float4* p = xs + (unsigned)(2 * min(g, kreal - 1)) * XP + x_lane;  // (steps past K1 multiply zero filters)
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = p[32 * j];
  };
  auto w2load = [&](float4(&dst)[2], int p, int ks) {  // 16-channel group ks of the wave's 64 channels of pass p
    const unsigned q0 = (unsigned)((p * 256 + 64 * wave + 16 * ks) >> 3);  // its first plane
    dst[0] = bload4(rw2, w2_lane, q0 * N2 * 16u);
    dst[1] = bload4(rw2, w2_lane + 32u * 16u, q0 * N2 * 16u);
  };
  auto init_acc = [&](int p) {  // conv5's bias rides in the accumulators
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      floatx16 t;  // (a whole new value: element-wise writes into acc[f][j] would keep the old tile alive)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 b = s_b1[((p * 256 + 64 * wave + 32 * f + 8 * rg) >> 2) + kh];
        t[rg * 4 + 0] = b.x;
        t[rg * 4 + 1] = b.y;
        t[rg * 4 + 2] = b.z;
        t[rg * 4 + 3] = b.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[f][j] = t;
    }
  };

  // ---- prologue of the block: the first item's tile, the first filter fragments ----------------------------------
  setup_stage(item, sq);
  for (int c = 0; c < nch; c += 2) {
    stage_load(sr[0], rin, sq, c);
    stage_load(sr[1], rin, sq, min(c + 1, nch - 1));
    stage_store(sr[0], c);
    if (c + 1 < nch) stage_store(sr[1], c + 1);
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) wload(wr[i], 0, i);  // (kst >= 8)
  __syncthreads();
  xload(xr[0], 0);

  while (true) {
    const int nitem = item + (int)gridDim.x;
    const bool has_next = nitem < A.nitems;
    setup_stage(has_next ? nitem : item, sqn);  // (no next item: the last pass re-stages this one - same bytes)
    {
      floatx16 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[i][j] = z;
    }

    // one k-step: request the filters of step + 7 and the activations of step + 1, multiply step.  (Seven steps - at the
    // rate the matrix pipe really runs at, ~3.5 k cycles: loads complete in order, so a filter request issued right
    // after the last pass' HBM requests for the next tile comes back behind them; three steps ahead, as in the first
    // version, the multiply loop sat out an HBM round trip per staging granule.)
    auto step = [&](auto i_tag, int p, int g) {
      constexpr int S = decltype(i_tag)::value & 7;
      constexpr int XS = decltype(i_tag)::value & 1;
      int g3 = g + 7, p3 = p;
      while (g3 >= kst) {  // (uniform; the heads of the next pass - or of the next item's first pass)
        g3 -= kst;
        p3 = p3 + 1 == NP ? 0 : p3 + 1;
      }
      wload(wr[(S + 7) & 7], p3, g3);
      xload(xr[XS ^ 1], g + 1 == kst ? 0 : g + 1);
      RTPOSE_HB_PIN();
      if (g < kreal) {  // (uniform, no load inside; a step past K1 - the ring and the chunk pairs want whole turns - is skipped)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int f = 0; f < 2; ++f)
            acc[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(wr[S][f]), as_bf8(xr[XS][j]), acc[f][j], 0, 0, 0);
      }
      RTPOSE_HB_PIN();
    };
    auto run_pass = [&](int p) {
      const bool lastp = p + 1 == NP;
      const i32x4 rs = lastp ? rin : rnull;  // only the last pass really fetches the next item's chunks
      w2load(w2r[0], p, 0);  // the first two groups of the fold at the end of the pass; the other two under it
      w2load(w2r[1], p, 1);
      init_acc(p);
      for (int g0 = 0; g0 < kst; g0 += 8) {
        // two chunks of the next item are requested behind the first step's filter request (the next wait for filters
        // does not cover them) and stored eight steps - 64 MFMAs, ~2 k cycles - later: one chunk ahead (round-5 first
        // version) the HBM round trip under load was longer than the chunk
        const int c0 = g0 >> 2;
        step(IntTag<0>(), p, g0);
        stage_load(sr[0], rs, sqn, c0);
        stage_load(sr[1], rs, sqn, min(c0 + 1, nch - 1));
        step(IntTag<1>(), p, g0 + 1);
        step(IntTag<2>(), p, g0 + 2);
        step(IntTag<3>(), p, g0 + 3);
        step(IntTag<4>(), p, g0 + 4);
        step(IntTag<5>(), p, g0 + 5);
        step(IntTag<6>(), p, g0 + 6);
        step(IntTag<7>(), p, g0 + 7);
        if (lastp) {
          __syncthreads();  // every wave has read the chunks c0, c0 + 1 for the last time
          stage_store(sr[0], c0);
          if (c0 + 1 < nch) stage_store(sr[1], c0 + 1);
        }
      }
      // ---- GEMM 2: fold the wave's 64 channels of this pass into its heads' sums, straight from the accumulators
      mfma_drain();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int f = ks >> 1, h = ks & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const floatx16& a = acc[f][j];
          const float4 bv = make_float4(pack_relu2(acc_read(a[8 * h + 0]), acc_read(a[8 * h + 1])),
                                        pack_relu2(acc_read(a[8 * h + 2]), acc_read(a[8 * h + 3])),
                                        pack_relu2(acc_read(a[8 * h + 4]), acc_read(a[8 * h + 5])),
                                        pack_relu2(acc_read(a[8 * h + 6]), acc_read(a[8 * h + 7])));
          o[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(w2r[ks & 1][0]), as_bf8(bv), o[0][j], 0, 0, 0);
          o[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(w2r[ks & 1][1]), as_bf8(bv), o[1][j], 0, 0, 0);
          RTPOSE_HB_PIN();  // (one fragment's eight values at a time: hoisted together they would not fit the registers)
        }
        if (ks < 2) w2load(w2r[ks & 1], p, ks + 2);
      }
      RTPOSE_HB_PIN();
    };

    for (int p = 0; p < NP; ++p) run_pass(p);

    // ---- the four partial sums of every pixel fragment meet in LDS; wave w adds the head columns 16 w .. 16 w + 15
    //      (register quadruples r4 = 2 w, 2 w + 1 of the 8 a lane holds per fragment) in wave order and stores -------
    mfma_drain();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int mo = item * PXB + 32 * j + l31;
      const bool ov = mo < A.M;
      const unsigned oq = (unsigned)pixel_q(min(mo, A.M - 1), A.out_lead, A.out_hs, A.out_ws) * (unsigned)A.out_cstride +
                          (unsigned)A.out_choff;
#pragma unroll
      for (int r4 = 0; r4 < 8; ++r4) {
        const floatx16& v = o[r4 >> 2][j];
        const int b = (r4 & 3) * 4;
        red[(wave * 8 + r4) * 64 + lane] = make_float4(acc_read(v[b]), acc_read(v[b + 1]), acc_read(v[b + 2]), acc_read(v[b + 3]));
        if (r4 & 1) { RTPOSE_HB_PIN(); }
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int r4 = 2 * wave + t;
        float4 s = red[(0 * 8 + r4) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const float4 v = red[(w * 8 + r4) * 64 + lane];
          s.x = s.x + v.x;
          s.y = s.y + v.y;
          s.z = s.z + v.z;
          s.w = s.w + v.w;
        }
        // r4 = 4 i + rg: head columns 32 i + 8 rg + 4 kh .. + 3 of pixel 32 j + l31
        const int col = 32 * (r4 >> 2) + 8 * (r4 & 3) + 4 * kh;
        const float4 b = s_b2[col >> 2];
        if (ov)
          *reinterpret_cast<float4*>(A.out + (oq + (unsigned)col)) = make_float4(s.x + b.x, s.y + b.y, s.z + b.z, s.w + b.w);
      }
      __syncthreads();  // (also: the next item's last chunk is in LDS before anyone reads it)
    }
    if (!has_next) break;
    item = nitem;
  }
}
#undef RTPOSE_HB_PIN

// w[cout][K] fp32 (+ bias) -> the heads' shared matrix [K / 8][64][8 bf16] at columns col_off .. col_off + cout - 1, with
// the k order GEMM 2 reads its B operand in: inside a 16-channel group c16, lane half kh = (c16 / 4) % 2 holds the
// channels {4 kh .. 4 kh + 3} and {8 + 4 kh .. 8 + 4 kh + 3} as elements 0..3 and 4..7 of plane 2 (c / 16) + kh
// (= the registers rg, rr of C1^T's accumulator layout).
__global__ void pack_head_w2_bf16_kernel(const float* __restrict__ w, const float* __restrict__ bias, int cout, int K,
                                         int col_off, unsigned short* __restrict__ wp, float* __restrict__ bp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cout) bp[col_off + i] = bias ? bias[i] : 0.f;
  if (i >= K * cout) return;
  const int n = i % cout, c = i / cout;
  const int c16 = c & 15;
  const int q = 2 * (c >> 4) + ((c16 >> 2) & 1), e = ((c16 >> 3) << 2) | (c16 & 3);
  wp[((size_t)q * N2 + col_off + n) * 8 + e] = to_bf16(w[(size_t)n * K + c]);
}
__global__ void zero_head_columns_bf16_kernel(unsigned short* __restrict__ wp, float* __restrict__ bp, int K, int c0, int c1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nc = c1 - c0;
  if (i < nc) bp[c0 + i] = 0.f;
  if (i >= K * nc) return;
  const int n = c0 + i % nc, c = i / nc;
  wp[((size_t)(c >> 3) * N2 + n) * 8 + (c & 7)] = 0;
}

}  // namespace headb

int pack_head_w2_bf16_launch(const float* w, const float* bias, int cout, int K, int col_off, void* wp, float* bp,
                             hipStream_t s) {
  if (!w || !wp || !bp || cout <= 0 || K <= 0 || (K % 16) || col_off < 0 || col_off + cout > headb::N2)
    return fail(RTPOSE_E_INVAL, "pack_head_w2_bf16: bad arguments");
  hipLaunchKernelGGL(headb::pack_head_w2_bf16_kernel, dim3(ceil_div(K * cout, 256)), dim3(256), 0, s, w, bias, cout, K,
                     col_off, static_cast<unsigned short*>(wp), bp);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int zero_head_columns_bf16_launch(void* wp, float* bp, int K, int c0, int c1, hipStream_t s) {
  if (!wp || !bp || K <= 0 || c0 < 0 || c1 > headb::N2 || c0 >= c1) return 0;
  hipLaunchKernelGGL(headb::zero_head_columns_bf16_kernel, dim3(ceil_div(K * (c1 - c0), 256)), dim3(256), 0, s,
                     static_cast<unsigned short*>(wp), bp, K, c0, c1);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

// d1: the wide pointwise conv (+ReLU): bf16 input slice (layouts count ELEMENTS), plain packing [cin / 8][cout][8] with
// cout a multiple of 256 (pack_pw_bf16_launch), cin a multiple of 16 up to 480; d2: the heads' shared matrix from pack_head_w2_bf16_launch, fp32 output with the columns at their channels.
int pw_head_bf16_fits(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2) {
  if (!d1 || !d2) return 0;
  if (d1->cin < 16 || (d1->cin % 16) || d1->cin > headb::kMaxK1 || d1->coutp <= 0 || (d1->coutp % 256) || d1->coutp > 1024 ||
      d1->cout != d1->coutp)
    return 0;
  if (!d1->relu || d2->relu || d1->dw_w || d2->dw_w || d1->pt_src || d2->pt_src || d1->out_cmap || d2->out_cmap) return 0;
  if (d2->cin != d1->coutp || d2->coutp != headb::N2 || d2->cout < 1 || d2->cout > headb::N2) return 0;
  if ((d1->lin.cstride % 8) || (d1->lin.choff % 8) || (!d1->in_planes && d1->lin.choff + d1->cin > d1->lin.cstride)) return 0;
  if ((d2->lout.cstride % 4) || (d2->lout.choff % 4) || d2->lout.choff + headb::N2 > d2->lout.cstride) return 0;
  return 1;
}

int pw_head_bf16_launch(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2, int N, int H, int W, hipStream_t s) {
  using namespace headb;
  if (!pw_head_bf16_fits(d1, d2))
    return fail(RTPOSE_E_INVAL, "pw_head_bf16: needs cin %% 16 == 0 (<= 480) -> cout %% 256 == 0 (+ReLU) -> 64 head columns");
  if (!d1->in || !d1->w_packed || !d1->bias_packed || !d2->w_packed || !d2->bias_packed || !d2->out)
    return fail(RTPOSE_E_INVAL, "pw_head_bf16: NULL argument");
  if (N <= 0 || H <= 0 || W <= 0) return fail(RTPOSE_E_INVAL, "pw_head_bf16: empty tensor");
  const long M = (long)N * H * W;
  const size_t in_bytes = rtpose_layout_pixels(&d1->lin, N, H, W) * (size_t)d1->lin.cstride * 2;
  if (M > 0x7fffffffL || in_bytes >= ((size_t)1 << 31) ||
      rtpose_layout_pixels(&d2->lout, N, H, W) * (size_t)d2->lout.cstride >= ((size_t)1 << 31))
    return fail(RTPOSE_E_INVAL, "pw_head_bf16: tensors must be below 2^31 bytes / floats (32-bit offsets)");
  Args a;
  memset(&a, 0, sizeof(a));
  a.in = reinterpret_cast<const unsigned short*>(d1->in);
  a.in_bytes = in_bytes;
  a.in_cstride = d1->lin.cstride;
  a.in_choff = d1->lin.choff;
  a.in_ws = d1->lin.ws;
  a.in_hs = d1->lin.hs;
  a.in_lead = d1->lin.lead;
  a.in_planes = d1->in_planes;
  a.w1 = d1->w_packed;
  a.b1 = d1->bias_packed;
  a.w2 = d2->w_packed;
  a.b2 = d2->bias_packed;
  a.out = d2->out;
  a.out_cstride = d2->lout.cstride;
  a.out_choff = d2->lout.choff;
  a.out_ws = d2->lout.ws;
  a.out_hs = d2->lout.hs;
  a.out_lead = d2->lout.lead;
  a.N = N;
  a.H = H;
  a.W = W;
  a.M = (int)M;
  a.K1 = d1->cin;
  a.N1 = d1->coutp;
  a.nitems = ceil_div((int)M, PXB);
  a.fHW = make_fastdiv(H * W);
  a.fW = make_fastdiv(W);
  const int grid = a.nitems < device_cu_count() ? a.nitems : device_cu_count();  // one block per CU
  const size_t lds = ((size_t)(a.K1 >> 3) * XP + kRed4) * 16;
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pw_head_bf16_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4608));  // (4592 bytes are static)
    attr_set.set(dev);
  }
  hipLaunchKernelGGL(pw_head_bf16_kernel, dim3(grid), dim3(256), lds, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace rtpose

extern "C" {

int rtpose_pw_head_bf16_fits(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2) { return rtpose::pw_head_bf16_fits(d1, d2); }

int rtpose_pw_head_bf16(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2, int N, int H, int W, void* stream) {
  return rtpose::pw_head_bf16_launch(d1, d2, N, H, W, rtpose::as_stream(stream));
}

int rtpose_pack_pw_head2_bf16(const float* w_oi, const float* bias, int cout, int cin, int col_off, void* w_packed,
                              float* bias_packed, void* stream) {
  return rtpose::pack_head_w2_bf16_launch(w_oi, bias, cout, cin, col_off, w_packed, bias_packed, rtpose::as_stream(stream));
}

}  // extern "C"
