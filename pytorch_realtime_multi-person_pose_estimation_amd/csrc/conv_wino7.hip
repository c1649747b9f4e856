// fp32 7x7 convolution in 1-D Winograd form F(FM, 7) along x (FM = 6; 4 behind RTPOSE_WINOGRAD7_M=4), direct along y,
// for gfx950 (MI355X): stride 1, "same" padding, fused bias (+ReLU).
//
// Stands in for the 7x7 nn.Conv2d + nn.ReLU modules of the refinement stages 2..6
// (lib/network/rtpose_vgg.py:108-127: Mconv1..5_stageN_L1/L2, cin 185 or 128 -> 128), which are 65 % of
// the network's flops.  Along x every group of FM consecutive outputs is computed from NFQ = FM + 6 "frequencies"
// (Toom-Cook interpolation points 0, +-1, +-2, +-1/2, +-3/2 [, +-2/3], inf; struct WT below):
//
//   out[y][FM gx + i][o] = sum_f AT[i][f] * sum_ky sum_c  V[y + ky - 3][gx][f][c] * U[ky][f][c][o]
//   V[r][gx][f][c] = sum_n BT[f][n] * in[r][FM gx - 3 + n][c]        U[ky][f][c][o] = sum_kx G[f][kx] * w[o][c][ky][kx]
//
// i.e. 7 NFQ multiplies per FM outputs and input channel instead of 49 FM: 3.5x fewer matrix-core flops for F(6,7)
// (84 per 6 outputs), 2.8x for F(4,7).  Through the whole network the stage outputs move by 1.5e-5 (F(6,7)) resp.
// 1e-5 (F(4,7)) against the direct sum (contract: 1e-3); F(8,7) would be 1e-4 and is not offered.
//
// MI355X shape:
//  * "position" = one group of FM output pixels.  A wave owns all NFQ frequencies of 32 consecutive positions x 32
//    output columns = NFQ v_mfma_f32_32x32x2_f32 accumulators (192 AGPRs, one wave per SIMD); the output transform
//    AT is lane-local and the bias rides in the accumulator of the point p = 1 (its AT column is all ones).
//    A block = 4 waves = 32 positions x 128 columns (wino7_f32), or - small grids - 32 positions x 32 columns with
//    the four waves splitting the frequencies (wino7s_f32); both sum in the same order: bit-identical results.
//  * V lives in LDS per 8-channel chunk as [padded row][f][c/4][gx][4 floats]: one transformed input row serves
//    the 7 output rows around it, the A fragment of (ky, f) is one ds_read_b128 per lane at
//    (row(lane) + ky, f, gx(lane)).  Double-buffered: the input segments of chunk c+2 are in flight and chunk c+1 is
//    transformed while chunk c is multiplied; one barrier per chunk (336 MFMAs per wave).
//  * fp32 VALU instructions run on the ALUs the fp32 MFMAs use, so the multiply loop has no vector address
//    arithmetic (raw buffer loads: fixed per-lane offset + scalar offset; LDS offsets are immediates where the
//    row geometry is a template parameter) and the transform is packed (float2) and issued in few full groups.
//  * B = transformed filters, packed [chunk][ky][f][c/4][cout][4] by rtpose_pack_conv_weights_winograd, straight
//    from L2 to registers up to four (ky, frequency pair) steps ahead.
//  * Launches with >= 256 tiles that are not whole rounds run as persistent blocks sharing (tile, chunk) units; a
//    split tile is continued by the next block from the saved sums (same summation order, wino7_segment).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "common.h"
#include "conv_exp.h"
#include "wino_common.h"

namespace rtpose {

size_t packed_weight_floats_wino7(int cout, int cin, int fm);

namespace wino7 {

using namespace winoc;

// F(FM, 7): FM outputs from FM + 6 inputs through NFQ = FM + 6 "frequencies" = Toom-Cook interpolation points
// 0, +-p_1 .. +-p_NP, inf.  kBT = the input transform, rows scaled by N_f = prod_{l != f} (p_f - p_l) (the filter
// transform divides by it).  Rows 2p+1 / 2p+2 (points +-p) share their even- and odd-n halves:
// V[2p+1] = E_p + O_p,  V[2p+2] = E_p - O_p; row 0 has only even, row NFQ-1 (inf) only odd columns.
//   FM = 4: points 0, +-1, +-2, +-1/2, +-3/2, inf - every entry a multiple of 1/16, exact in fp32; 70 multiplies per
//           4 outputs instead of 196 (2.8x); whole-network stage outputs move by < 1e-5
//   FM = 6: + the points +-2/3 - 84 per 6 outputs instead of 294 (3.5x); entries in 18ths / 144ths (rounded to fp32),
//           whole network 1.5e-5 (points +-3 instead: 6.4e-5, +-1/4: 2.8e-5; contract 1e-3)
template <int FM>
struct WT;
template <>
struct WT<4> {
  static constexpr int NFQ = 10, NP = 4, NSETS = 5;
  __device__ static constexpr float kPts[4] = {1.f, 2.f, 0.5f, 1.5f};
  __device__ static constexpr float kBT[10][10] = {
      {2.25f, 0.f, -12.8125f, 0.f, 17.0625f, 0.f, -7.5f, 0.f, 1.f, 0.f},
      {0.f, -2.25f, -2.25f, 10.5625f, 10.5625f, -6.5f, -6.5f, 1.f, 1.f, 0.f},
      {0.f, 2.25f, -2.25f, -10.5625f, 10.5625f, 6.5f, -6.5f, -1.f, 1.f, 0.f},
      {0.f, -1.125f, -0.5625f, 6.125f, 3.0625f, -7.f, -3.5f, 2.f, 1.f, 0.f},
      {0.f, 1.125f, -0.5625f, -6.125f, 3.0625f, 7.f, -3.5f, -2.f, 1.f, 0.f},
      {0.f, -4.5f, -9.f, 7.625f, 15.25f, -3.625f, -7.25f, 0.5f, 1.f, 0.f},
      {0.f, 4.5f, -9.f, -7.625f, 15.25f, 3.625f, -7.25f, -0.5f, 1.f, 0.f},
      {0.f, -1.5f, -1.f, 7.875f, 5.25f, -7.875f, -5.25f, 1.5f, 1.f, 0.f},
      {0.f, 1.5f, -1.f, -7.875f, 5.25f, 7.875f, -5.25f, -1.5f, 1.f, 0.f},
      {0.f, 2.25f, 0.f, -12.8125f, 0.f, 17.0625f, 0.f, -7.5f, 0.f, 1.f},
  };
};
template <>
struct WT<6> {
  static constexpr int NFQ = 12, NP = 5, NSETS = 6;
  __device__ static constexpr float kPts[5] = {1.f, 2.f, 0.5f, 1.5f, 0.666666667f};
  __device__ static constexpr float kBT[12][12] = {
      {-1.f, 0.f, 7.94444444f, 0.f, -20.3958333f, 0.f, 20.3958333f, 0.f, -7.94444444f, 0.f, 1.f, 0.f},
      {0.f, 1.f, 1.f, -6.94444444f, -6.94444444f, 13.4513889f, 13.4513889f, -6.94444444f, -6.94444444f, 1.f, 1.f, 0.f},
      {0.f, -1.f, 1.f, 6.94444444f, -6.94444444f, -13.4513889f, 13.4513889f, 6.94444444f, -6.94444444f, -1.f, 1.f, 0.f},
      {0.f, 0.5f, 0.25f, -3.84722222f, -1.92361111f, 9.23611111f, 4.61805556f, -7.88888889f, -3.94444444f, 2.f, 1.f, 0.f},
      {0.f, -0.5f, 0.25f, 3.84722222f, -1.92361111f, -9.23611111f, 4.61805556f, 7.88888889f, -3.94444444f, -2.f, 1.f, 0.f},
      {0.f, 2.f, 4.f, -7.88888889f, -15.7777778f, 9.23611111f, 18.4722222f, -3.84722222f, -7.69444444f, 0.5f, 1.f, 0.f},
      {0.f, -2.f, 4.f, 7.88888889f, -15.7777778f, -9.23611111f, 18.4722222f, 3.84722222f, -7.69444444f, -0.5f, 1.f, 0.f},
      {0.f, 0.666666667f, 0.444444444f, -5.f, -3.33333333f, 11.375f, 7.58333333f, -8.54166667f, -5.69444444f, 1.5f, 1.f, 0.f},
      {0.f, -0.666666667f, 0.444444444f, 5.f, -3.33333333f, -11.375f, 7.58333333f, 8.54166667f, -5.69444444f, -1.5f, 1.f, 0.f},
      {0.f, 1.5f, 2.25f, -8.54166667f, -12.8125f, 11.375f, 17.0625f, -5.f, -7.5f, 0.666666667f, 1.f, 0.f},
      {0.f, -1.5f, 2.25f, 8.54166667f, -12.8125f, -11.375f, 17.0625f, 5.f, -7.5f, -0.666666667f, 1.f, 0.f},
      {0.f, -1.f, 0.f, 7.94444444f, 0.f, -20.3958333f, 0.f, 20.3958333f, 0.f, -7.94444444f, 0.f, 1.f},
  };
};

struct Group {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  int in_cstride, in_choff, in_ws, in_hs, in_lead;
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  int cout, cout_pad;
  size_t in_bytes, w_bytes;  // extents of the allocations from in / w (hardware bounds clamp of the buffer loads)
};

struct Args {
  Group g[2];
  int N, H, W;
  int GX, T;    // positions per row, and in the whole batch
  int TPI;      // > 0: position strips restart with every image (TPI blocks per image); 0: one flat strip space
  int cin, relu;
  int RS;       // float4 per transformed row in LDS (>= 20 GX, chosen against bank conflicts)
  int VB;       // float4 per V buffer
  int mtiles, ntiles, ncombo, xcd_remap;
  int persist;      // 1: gridDim.x blocks share the (tile, chunk) units evenly (see wino7_f32)
  float* scratch;   // persist: one accumulator tile per block (4 waves x 12 x 16 registers x 64 lanes)
  int* flags;       // persist: flags[p] = 1 while the sums block p saved wait for block p + 1 (cleared at launch)
  int* err;         // persist: device error word (bit 0: a hand-over wait ran out, the tile's results are invalid)
};

constexpr int CK = 8, CG = 2;   // channels per chunk, 16-byte channel groups per chunk

// LDS row stride (float4): >= 20 GX, and = GX modulo 16 so that the 32 positions of a wave tile, which
// wrap from one transformed row to the next, keep landing in distinct 16-byte bank slots
__host__ __device__ constexpr int row_stride(int gx, int nfq) {
  int rs = nfq * CG * gx;
  while ((rs & 15) != (gx & 15)) ++rs;
  return rs;
}
// rows (incl. the 6 halo rows) a strip of 32 positions that stays inside one image can touch
__host__ __device__ constexpr int strip_rows(int gx) { return (31 + gx - 1) / gx + 1 + 6; }

// NI  = (row, gx, channel group) transform items per thread and chunk
// GXT = compile-time position groups per row (0: run-time).  With GXT every LDS address of the multiply loop is
//       one base register + an immediate; the run-time form pays one v_add per access.
// The input-transform role of a thread: NI items (row, gx, channel group) of the block's transformed rows.
// The last group of a row reaches past the row's own 3-pixel gap (x >= W + 3 is the NEXT row's - or the next
// image's - data) when W is not a multiple of FM.  Those inputs only meet outputs that are not stored, but
// through the transform they would cancel only up to rounding, and an image's result would depend on its
// neighbour in the batch: segments n >= 7 are therefore clamped to the last gap pixel (a zero).
// (the input descriptor is based at the block's first row, so the 32-bit offsets stay small whatever the size
// of the activation buffer)
template <int NI, int FM>
struct Xform {
  using T = WT<FM>;
  static constexpr int NFQ = T::NFQ, NP = T::NP, NHI = NFQ - 7;
  i32x4 rin;
  unsigned voff[NI], vhi[NI][NHI], pxb;
  int vdst[NI], fstride;  // LDS slot of frequency 0 of an item; float4 between frequencies
  F4 d[NI][NFQ], eo[2];

  __device__ __forceinline__ void setup(const Group& g, int W, int GX, int RS, int R0, int nrows) {
    const int tid = threadIdx.x;
    const int nitems = nrows * GX * CG;
    const long qbase = (long)g.in_lead + (long)(R0 - 3) * g.in_ws - 3;
    const size_t o0 = (size_t)qbase * g.in_cstride + g.in_choff;
    rin = make_rsrc(g.in + o0, g.in_bytes - o0 * 4);
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      // (8-wave form: the threads 256.. mirror the items of 0..255 - sibling waves share the transform of an item)
      const int i = min((tid & 255) + k * 256, nitems - 1);  // surplus threads repeat the last item (same value, same slot)
      const int r = i / (GX * CG), rem = i - r * (GX * CG);
      // channel-group-major inside a row: the 8 contiguous lanes a ds_write_b128 is serviced in then write 8
      // consecutive 16-byte slots (gx, cg interleaved they hit 4 slots twice: 12.5 % of SQ_LDS_IDX_ACTIVE in round 2)
      const int cg = RTPOSE_EXP_W7_CGMAJOR ? rem / GX : (rem & 1), gx = RTPOSE_EXP_W7_CGMAJOR ? rem - cg * GX : (rem >> 1);
      voff[k] = (unsigned)((((long)r * g.in_ws + FM * gx) * g.in_cstride + cg * 4) * 4);
      vdst[k] = r * RS + cg * GX + gx;
#pragma unroll
      for (int n = 7; n < NFQ; ++n) vhi[k][n - 7] = voff[k] + (unsigned)(min(n, W + 5 - FM * gx) * g.in_cstride * 4);
    }
    pxb = (unsigned)g.in_cstride * 4;  // bytes per pixel (uniform)
    fstride = CG * GX;
  }
  __device__ __forceinline__ void load_piece(int chunk, int k, int n) {
    const unsigned cb = (unsigned)chunk * (CK * 4);
    d[k][n] = n < 7 ? bload(rin, voff[k], cb + n * pxb) : bload(rin, vhi[k][n < 7 ? 0 : n - 7], cb);
  }
  // The transform of one item in NP + 2 groups of 10..20 packed VALU instructions.  (A lone VALU instruction between
  // two MFMAs costs ~8 cycles of matrix time, a group of 12..16 costs ~60 in all: few, full groups.)
  //   group p < NP: E_p, O_p (NP terms each), then frequencies 2p+1 = E_p + O_p and 2p+2 = E_p - O_p -> LDS
  //   group NP: frequency 0;  group NP + 1: frequency NFQ - 1 (inf)                    (NP + 1 terms each)
  __device__ __forceinline__ void tgroup(float4* vw, int k, int gidx) {
    if (gidx < NP) {
      const int row = 2 * gidx + 1;
#pragma unroll
      for (int odd = 0; odd < 2; ++odd) {
        const int n0 = odd ? 1 : 2;  // column 0 of these rows is zero
        F4 s = mul4(T::kBT[row][n0], d[k][n0]);
#pragma unroll
        for (int t = 1; t < NP; ++t) s = fma4(T::kBT[row][n0 + 2 * t], d[k][n0 + 2 * t], s);
        eo[odd] = s;
      }
      vw[vdst[k] + row * fstride] = to_float4(add4(eo[0], eo[1]));
      vw[vdst[k] + (row + 1) * fstride] = to_float4(sub4(eo[0], eo[1]));
    } else {
      const int f = gidx == NP ? 0 : NFQ - 1, n0 = gidx == NP ? 0 : 1;
      F4 s = mul4(T::kBT[f][n0], d[k][n0]);
#pragma unroll
      for (int t = 1; t < NP + 1; ++t) s = fma4(T::kBT[f][n0 + 2 * t], d[k][n0 + 2 * t], s);
      vw[vdst[k] + f * fstride] = to_float4(s);
    }
  }
};

// One segment = chunks [cb, ce) of tile (mt, c).  A tile split between two blocks is summed in the SAME order as an
// unsplit one: the block with the first chunks (cb == 0, ce < all) saves its raw accumulators to scratch slot
// `slot` and raises the flag; the block with the last chunks (cb > 0) starts from them instead of from zero and
// stores the tile.  Results are bit-identical however the launch is cut.
// FS = 2: the "two waves per SIMD" form.  A block has 8 waves; wave w = (fh = w / 4, wn = w % 4) owns the frequencies
// [fh NFW, (fh + 1) NFW) of the wave tile wn (NFW = NFQ / 2 = 6 accumulators = 96 AGPRs, so two waves fit a SIMD's
// 512 registers).  Every frequency sum still runs over chunks, ky and k pairs in the order of the 4-wave form and
// of wino7s_f32: bit-identical results.  Why: a single wave issues a v_mfma_f32_32x32x2_f32 every ~71 cycles at
// best (tools/exp/mfma_issue.hip; 65 with a second wave on the SIMD), and whatever else it issues - the transform's
// VALU groups, LDS writes, waits - stops its MFMAs altogether; with a sibling the matrix pipe keeps running.  The input
// transform is done by the waves 0..3 only (one per SIMD): the same number of VALU instructions per CU as before,
// issued while the sibling multiplies.  The halves exchange their accumulators through LDS before the output
// transform (each wave finishes 16 of the 32 positions).
template <int NI, int GXT, int FM, int FS>
__device__ __forceinline__ void wino7_segment(const Args& A, float4* V4, const int mt, const int c, const int cb,
                                              const int ce, const int nch, const int slot) {
  using T = WT<FM>;
  // frequencies, +-point pairs, B register sets (F(6,7) with two transform items per thread: 3 sets, weights only 2
  // steps ahead - the registers go to the second item's 12 segments)
  constexpr int NFQ = T::NFQ, NP = T::NP;
  constexpr int NFW = NFQ / FS;                              // frequencies of a wave
  static_assert(NFW % 2 == 0, "frequency pairs");
  constexpr int NSETS = FS == 2 ? RTPOSE_EXP_W7_FS2SETS : (FM == 6 && NI == 2) ? 3 : T::NSETS;
  constexpr int NPS = 7 * NFW / 2;                           // (ky, frequency pair) steps per chunk (NPS % NSETS == 0)
  // B prefetch distance in steps.  The 8-wave form needs the deepest ring it can have: a step is 8 MFMAs of the wave's
  // own, and with 2 steps (3 sets) the layer took 0.603 ms, with 4 (7 sets) 0.527, with 6 0.522 (4-wave form: 0.544)
  constexpr int PF = FS == 2 ? NSETS - 1 : (RTPOSE_EXP_W7_PF < NSETS ? RTPOSE_EXP_W7_PF : NSETS - 1);
  static_assert(NPS % NSETS == 0, "B register sets must rotate in step with the chunk");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = FS == 2 ? (wv & 3) : wv, fh = FS == 2 ? (wv >> 2) : 0;
  // Who transforms.  RTPOSE_EXP_W7_XSPLIT = 0: the waves 0..3 (one per SIMD) do all of it while the sibling multiplies.
  // 1: both siblings load an item's segments and each forms half of its frequency groups (even / odd group index):
  // twice the segment loads (L2 hits), the VALU groups and LDS writes split evenly - neither sibling waits at the
  // chunk barrier for the other's transform.
  constexpr bool XSPLIT = FS == 2 && RTPOSE_EXP_W7_XSPLIT;
  const bool xf = XSPLIT || wv < 4;  // this wave takes part in the input transform
  auto my_group = [&](int gi) { return !XSPLIT || (gi & 1) == fh; };
#if RTPOSE_EXP_W7_PRIO == 1
  if (FS == 2 && xf) __builtin_amdgcn_s_setprio(1);
#elif RTPOSE_EXP_W7_PRIO == 2
  if (FS == 2 && !xf) __builtin_amdgcn_s_setprio(1);
#endif
  const int l31 = lane & 31, kh = lane >> 5;
  const int nt = c % A.ntiles, grp = c / A.ntiles;
  const Group g = grp ? A.g[1] : A.g[0];
  const int GX = GXT ? GXT : A.GX;
  const int RS = GXT ? row_stride(GXT, NFQ) : A.RS;
  const int VB = GXT ? strip_rows(GXT) * row_stride(GXT, NFQ) : A.VB;
  const int PI = A.H * GX;  // positions per image

  // ---- this block's 32 positions [t0, t0 + 32) of the flat (n, y, gx) order, valid below tlim ----------
  int t0, tlim;
  if (A.TPI) {
    const int n = mt / A.TPI;
    t0 = n * PI + (mt - n * A.TPI) * 32;
    tlim = (n + 1) * PI;
  } else {
    t0 = mt * 32;
    tlim = A.T;
  }
  // rows of the padded layout it needs: [R0 - 3, R1 + 3]
  int R0, nrows;
  {
    const int n0 = t0 / PI, y0 = (t0 - n0 * PI) / GX;
    const int t1 = min(t0 + 31, tlim - 1);
    const int n1 = t1 / PI, y1 = (t1 - n1 * PI) / GX;
    R0 = n0 * g.in_hs + y0;
    nrows = n1 * g.in_hs + y1 - R0 + 7;
  }
  // ---- input transform role: NI items (row, gx, channel group), see Xform ----------------------------
  Xform<NI, FM> X;
  X.setup(g, A.W, GX, RS, R0, nrows);
  // (RTPOSE_EXP_W7_NULLDESC, off: the waves 4..7 of the 8-wave form issue the segment loads too, through a descriptor of
  //  zero extent, so that the loads are not under a wave-uniform branch and the compiler's vmcnt bookkeeping stays exact -
  //  the fix that gave conv_wino4.hip 6 %; measured 0 % here, the deep filter ring already covers it)
  if (RTPOSE_EXP_W7_NULLDESC && !xf) X.rin = make_rsrc(g.in, 0);
  const i32x4 rw = make_rsrc(g.w, g.w_bytes);
  auto load_piece = [&](int chunk, int k, int n) { X.load_piece(chunk, k, n); };
  auto tgroup = [&](float4* vw, int k, int gidx) { X.tgroup(vw, k, gidx); };

  // ---- MFMA roles ---------------------------------------------------------------------------------
  int abase;
  {
    const int t = min(t0 + l31, tlim - 1);  // positions past the end repeat the last one (not stored)
    const int n = t / PI, r = t - n * PI;
    const int y = r / GX, gx = r - y * GX;
    abase = (n * g.in_hs + y - R0) * RS + kh * GX + gx + (fh * NFW) * CG * GX;  // this wave's first frequency
  }
  const int ncol = nt * 128 + wn * 32 + l31;
  floatx16 acc[NFW];
  // this wave's rows of the scratch slot (frequency-major: the slot looks the same whichever form wrote it)
  float* sp = A.scratch + ((size_t)(slot * 4 + wn) * (NFQ * 16) + (size_t)fh * NFW * 16) * 64 + lane;
  if (cb == 0) {
#pragma unroll
    for (int f = 0; f < NFW; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    if (fh == 0) {
      const float b0 = g.bias[ncol];  // padded to cout_pad
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[1][r] = b0;
    }
  } else {
    // Continue the sums the previous block started.  It saved them FIRST THING in its life, so the wait is a
    // formality provided block p - 1 is dispatched no later than block p - the order the hardware dispatches a 1-D
    // grid in (the launch has one block per CU, all resident).  The wait is bounded all the same (~1 s): if it runs
    // out, the error word is raised (rtpose_net_device_status / the caller of rtpose_conv2d_winograd_ex reads it)
    // and the block goes on with whatever the slot holds instead of hanging the GPU.
    if (tid == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(A.flags + slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 1) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22)) {
          __hip_atomic_fetch_or(A.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
    for (int f = 0; f < NFW; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][r] = sp[(f * 16 + r) * 64];
    __syncthreads();  // all reads done before the slot is handed back
    if (tid == 0) __hip_atomic_store(A.flags + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const unsigned boff = (unsigned)((kh * g.cout_pad + ncol) * 16);
  const unsigned fstep = (unsigned)(CG * g.cout_pad * 16);  // bytes per frequency block
  // uniform byte offset of the next B fragment to fetch: the packed filters are [chunk][ky][f]; a wave walks its own
  // NFW frequencies of every ky (bnext advances it, skipping the sibling's frequencies)
  unsigned wso = ((unsigned)cb * (7 * NFQ) + (unsigned)(fh * NFW)) * fstep;
  int bl = 0;  // fragments fetched so far, modulo NFW (compile-time after unrolling)
  auto bnext = [&](int& cnt) {
    wso += fstep;
    if (++cnt == NFW) {
      cnt = 0;
      if (FS > 1) wso += (unsigned)(NFQ - NFW) * fstep;
    }
  };
  float4 bs[NSETS][2];

  if (xf) {
#pragma unroll
    for (int k = 0; k < NI; ++k)
#pragma unroll
      for (int n = 0; n < NFQ; ++n) load_piece(cb, k, n);
#pragma unroll
    for (int k = 0; k < NI; ++k)
#pragma unroll
      for (int gi = 0; gi < NP + 2; ++gi)
        if (my_group(gi)) tgroup(V4, k, gi);
    const int c1 = min(cb + 1, ce - 1);
#pragma unroll
    for (int k = 0; k < NI; ++k)
#pragma unroll
      for (int n = 0; n < NFQ; ++n) load_piece(c1, k, n);
  }
  // (the first B fragments are requested AFTER the segment loads, as in the steady state of the loop below: the
  //  compiler's in-order vmcnt bookkeeping merges the two entries of the loop, and with the B loads older than the
  //  segment loads here it made every transform group wait for the B loads in flight - 5 % of the kernel)
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    bs[s][0] = bload_f4(rw, boff, wso);
    bnext(bl);
    bs[s][1] = bload_f4(rw, boff, wso);
    bnext(bl);
  }
  __syncthreads();

#define RTPOSE_PIN()             \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)
  // One step = the frequency pair (2 fp, 2 fp + 1) of one ky = 8 MFMAs on two alternating accumulators, with a
  // filler slot after every second one.  Slot 0 / 1: the A fragments of the next step (LDS); slot 2 / 3: the B
  // fragments PF steps ahead (L2).  Slot 3 of the first steps also carries transform work of the NEXT chunk: its
  // 6 NI groups (steps 0..11), then the 10 NI segment loads of the chunk after that, two per step.
  constexpr int NG = (NP + 2) * NI, GSTR = (NI == 1 && FS == 1) ? 2 : 1, LS = NG * GSTR;  // groups, their step stride, first load step
  constexpr int LPS = RTPOSE_EXP_W7_LPS, LSTEPS = (NFQ * NI + LPS - 1) / LPS;    // segment loads per step, steps with loads
  static_assert(LS + LSTEPS <= NPS, "transform work does not fit the steps of a chunk");
  float4 a[2][2];
  for (int chunk = cb; chunk < ce; ++chunk) {
    const float4* va = V4 + ((chunk - cb) & 1) * VB + abase;
    float4* vw = V4 + ((chunk - cb + 1) & 1) * VB;
    const int c2 = min(chunk + 2, ce - 1);  // the last chunks re-stage themselves (never read)
    a[0][0] = va[0];
    a[0][1] = va[CG * GX];
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int fp = ps % (NFW / 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        {
          const float4 a0 = a[ps & 1][0], a1 = a[ps & 1][1], b0 = bs[ps % NSETS][0], b1 = bs[ps % NSETS][1];
          const float a0v[4] = {a0.x, a0.y, a0.z, a0.w}, a1v[4] = {a1.x, a1.y, a1.z, a1.w};
          const float b0v[4] = {b0.x, b0.y, b0.z, b0.w}, b1v[4] = {b1.x, b1.y, b1.z, b1.w};
          acc[2 * fp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[j], b0v[j], acc[2 * fp], 0, 0, 0);
          acc[2 * fp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[j], b1v[j], acc[2 * fp + 1], 0, 0, 0);
        }
        RTPOSE_PIN();
        if (j < 2) {  // A of the next step (the first step of a chunk is read after the barrier)
          if (ps + 1 < NPS) {
            const int kyn = (ps + 1) / (NFW / 2), fn = 2 * ((ps + 1) % (NFW / 2)) + j;
            a[(ps + 1) & 1][j] = RTPOSE_EXP_A(va[kyn * RS + fn * CG * GX], a[ps & 1][j]);
          }
        } else {      // B PF steps ahead
          bs[(ps + PF) % NSETS][j - 2] = RTPOSE_EXP_B(bload_f4(rw, boff, wso), bs[ps % NSETS][j - 2]);
          bnext(bl);
        }
        if (RTPOSE_EXP_STAGE && j == 3 && (xf || RTPOSE_EXP_W7_NULLDESC)) {
          if (ps % GSTR == 0 && ps / GSTR < NG) {
            if (xf && (RTPOSE_EXP_W7_TMASK & 1) && my_group((ps / GSTR) % (NP + 2)))
              tgroup(vw, (ps / GSTR) / (NP + 2), (ps / GSTR) % (NP + 2));
          } else if (ps >= LS && ps < LS + LSTEPS && (RTPOSE_EXP_W7_TMASK & 2)) {  // LPS segment loads per step
#pragma unroll
            for (int q = 0; q < LPS; ++q) {
              const int l = LPS * (ps - LS) + q;
              if (l < NFQ * NI) load_piece(c2, l / NFQ, l % NFQ);
            }
          }
        }
        RTPOSE_PIN();
      }
    }
    __syncthreads();
  }
#undef RTPOSE_PIN

  // ---- first part of a split tile: hand the sums over ------------------------------------------------
  if (ce < nch) {
#pragma unroll
    for (int f = 0; f < NFW; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) sp[(f * 16 + r) * 64] = acc[f][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();  // every wave's sums are out (and released) before the flag goes up
    if (tid == 0) __hip_atomic_store(A.flags + slot, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }

  // ---- epilogue: output transform AT (points 0, +-1, +-2, +-1/2, +-3/2, inf), (+ReLU), masked stores ----
  // accumulator register r of a lane = position (r / 4) * 8 + 4 kh + r % 4 of the block, column l31
  const bool col_ok = ncol < g.cout;
  float* out_base = g.out + g.out_choff + ncol;
  // FS = 2: wave (wn, fh) finishes the accumulator registers [8 fh, 8 fh + 8) = positions 16 fh .. 16 fh + 15 of the
  // tile and gets the sibling's NFW frequencies of those registers through LDS (the V buffers are free now):
  // E[wn][writer's fh][f][r % 8][lane]
  constexpr int R0E = 16 / FS;  // registers a wave finishes
  float* E = reinterpret_cast<float*>(V4);
  if (FS == 2) {
    float* ew = E + (size_t)((wn * 2 + fh) * NFW * 8) * 64 + lane;
#pragma unroll
    for (int f = 0; f < NFW; ++f)
#pragma unroll
      for (int rl = 0; rl < 8; ++rl) ew[(f * 8 + rl) * 64] = fh ? acc[f][rl] : acc[f][8 + rl];  // what the sibling finishes
    __syncthreads();
  }
  const float* er = E + (size_t)((wn * 2 + (1 - fh)) * NFW * 8) * 64 + lane;
  int tcur = t0 + (FS == 2 ? 16 * fh : 0) + 4 * kh;
  int sn = tcur / PI, sy, sx;
  {
    const int r = tcur - sn * PI;
    sy = r / GX;
    sx = r - sy * GX;
  }
#pragma unroll
  for (int rr = 0; rr < (RTPOSE_EXP_W_EPI < R0E ? RTPOSE_EXP_W_EPI : R0E); ++rr) {
    const int r = rr;  // position stepping below depends on r % 4 only
    float m[NFQ];
    if (FS == 2) {
      // (the two halves of the select are resolved per wave: fh is uniform)
#pragma unroll
      for (int f = 0; f < NFW; ++f) {
        const float own = fh ? acc[f][8 + rr] : acc[f][rr];
        const float oth = er[(f * 8 + rr) * 64];
        m[f] = fh ? oth : own;
        m[NFW + f] = fh ? own : oth;
      }
    } else {
#pragma unroll
      for (int f = 0; f < NFQ; ++f) m[f] = acc[f % NFW][rr];
    }
    float S[NP], D[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      S[p] = m[2 * p + 1] + m[2 * p + 2];
      D[p] = m[2 * p + 1] - m[2 * p + 2];
    }
    // out_i = [i == 0] M_0 + sum_p p^i (S_p for even i, D_p for odd i) + [i == FM - 1] M_inf
    float y[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      float v = (i & 1) ? D[0] : S[0];  // the point 1
      float pw[NP];
#pragma unroll
      for (int p = 1; p < NP; ++p) {
        pw[p] = 1.f;
        for (int e = 0; e < i; ++e) pw[p] *= T::kPts[p];  // compile-time constant
        v = __builtin_fmaf(pw[p], (i & 1) ? D[p] : S[p], v);
      }
      if (i == 0) v += m[0];
      if (i == FM - 1) v += m[NFQ - 1];
      y[i] = v;
    }
    if (col_ok && tcur < tlim) {
      const size_t q = (size_t)g.out_lead + (size_t)(sn * g.out_hs + sy) * g.out_ws + FM * sx;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const float v = A.relu ? fmaxf(y[i], 0.f) : y[i];
        if (FM * sx + i < A.W) out_base[(q + i) * g.out_cstride] = v;
      }
    }
    const int dstep = (r & 3) == 3 ? 5 : 1;  // next register: +1, +1, +1, +5 positions
    tcur += dstep;
    sx += dstep;
    while (sx >= GX) {
      sx -= GX;
      if (++sy >= A.H) {
        sy = 0;
        ++sn;
      }
    }
  }
  if (FS == 2) __syncthreads();  // the exchange area is the next segment's V buffer
}

// Grid: either one block per tile (persist = 0; XCD-aware order as in conv_mfma.hip), or - when there are at
// least as many tiles as blocks - gridDim.x persistent blocks that share the (tile, chunk) units of the launch
// EVENLY: block p owns units [p U / P, (p + 1) U / P) of the tile-major order, i.e. the last chunks of one tile,
// some whole tiles, the first chunks of another.  32 x 46 x 46 x (2 branches): 1152 tiles on 256 CUs are 4.5 tiles
// per CU instead of 5 rounds of whole tiles.  A tile is split between at most two blocks (units per block >= chunks
// per tile).  Order inside a block: the FIRST chunks of its last tile first (saved for block p + 1), the whole
// tiles, and the LAST chunks of its first tile at the very end, continuing the sums block p - 1 saved at its start:
// nobody waits, and the sums run in the order of an unsplit tile.
template <int NI, int GXT, int FM, int FS>
__global__ __launch_bounds__(256 * FS, 1) void wino7_f32(const Args A) {
  extern __shared__ __attribute__((aligned(16))) float4 V4[];
  const int nch = A.cin / CK;
  long u0, u1;
  if (A.persist) {
    const long U = (long)A.mtiles * A.ncombo * nch, P = gridDim.x, p = blockIdx.x;
    u0 = p * U / P;
    u1 = (p + 1) * U / P;
  } else {
    const int bi = blockIdx.x;
    int mt, c;
    if (A.xcd_remap) {
      const int xcd = bi & 7, j = bi >> 3;
      c = j % A.ncombo;
      mt = (j / A.ncombo) * 8 + xcd;
    } else {
      mt = bi % A.mtiles;
      c = bi / A.mtiles;
    }
    if (mt >= A.mtiles) return;
    u0 = ((long)mt * A.ncombo + c) * nch;
    u1 = u0 + nch;
  }
  const int tf = (int)(u0 / nch), cbf = (int)(u0 - (long)tf * nch);          // first tile, its first owned chunk
  const int tl = (int)((u1 - 1) / nch), cel = (int)(u1 - (long)tl * nch);    // last tile, end of its owned chunks
  const int nseg = tl - tf + 1;
  for (int i = 0; i < nseg; ++i) {
    // segment order: [tl if its end is not owned] , tf + 1 .. , [tf last if its start is not owned]
    int tile, cb = 0, ce = nch;
    const bool head_first = cel < nch && nseg > 1, tail_last = cbf > 0;
    if (head_first && i == 0) {
      tile = tl;
      ce = cel;
    } else if (tail_last && i == nseg - 1) {
      tile = tf;
      cb = cbf;
      if (tf == tl) ce = cel;
    } else {
      tile = tf + i - (head_first ? 1 : 0) + (tail_last ? 1 : 0);
      if (tile == tl) ce = cel;
    }
    const int mt = tile / A.ncombo, c = tile - mt * A.ncombo;
    // slot: a block saves into its own, and continues from its predecessor's
    wino7_segment<NI, GXT, FM, FS>(A, V4, mt, c, cb, ce, nch, cb > 0 ? (int)blockIdx.x - 1 : (int)blockIdx.x);
  }
}

// Small grids (few images): the same arithmetic on 4x as many blocks.  A block = 32 positions x 32 columns, its four
// waves split the FREQUENCIES (3 of the 12 each) instead of the columns; every wave still runs over all chunks and ky
// in the same order, so each frequency sum - and after the exchange through LDS the output transform - is
// bit-identical to wino7_f32's.  One 368 x 368 image: 24 tiles x 4 column tiles = 96 blocks instead of 24, each with a
// quarter of the MFMAs per chunk (the input transform is repeated by the 4 column tiles, on otherwise idle CUs).
template <int NI>
__global__ __launch_bounds__(256, 1) void wino7s_f32(const Args A) {
  constexpr int FM = 6;
  using T = WT<FM>;
  constexpr int NFQ = T::NFQ, NP = T::NP, FW = NFQ / 4;  // frequencies per wave
  constexpr int NSETS = 7, PF = 3;                        // B register sets (one per ky step of a chunk), prefetch distance
  extern __shared__ __attribute__((aligned(16))) float4 V4[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int bi = blockIdx.x;
  const int c = bi % A.ncombo, mt = bi / A.ncombo;  // the column tiles / branches of a position strip are neighbours
  const int nt = c % A.ntiles, grp = c / A.ntiles;
  const Group g = grp ? A.g[1] : A.g[0];
  const int GX = A.GX, RS = A.RS, VB = A.VB;
  const int PI = A.H * GX;
  int t0, tlim;
  if (A.TPI) {
    const int n = mt / A.TPI;
    t0 = n * PI + (mt - n * A.TPI) * 32;
    tlim = (n + 1) * PI;
  } else {
    t0 = mt * 32;
    tlim = A.T;
  }
  int R0, nrows;
  {
    const int n0 = t0 / PI, y0 = (t0 - n0 * PI) / GX;
    const int t1 = min(t0 + 31, tlim - 1);
    const int n1 = t1 / PI, y1 = (t1 - n1 * PI) / GX;
    R0 = n0 * g.in_hs + y0;
    nrows = n1 * g.in_hs + y1 - R0 + 7;
  }
  Xform<NI, FM> X;
  X.setup(g, A.W, GX, RS, R0, nrows);
  const i32x4 rw = make_rsrc(g.w, g.w_bytes);

  int abase;
  {
    const int t = min(t0 + l31, tlim - 1);
    const int n = t / PI, r = t - n * PI;
    const int y = r / GX, gx = r - y * GX;
    abase = (n * g.in_hs + y - R0) * RS + kh * GX + gx + (wv * FW) * CG * GX;  // this wave's first frequency
  }
  const int ncol = nt * 32 + l31;
  floatx16 acc[FW];
#pragma unroll
  for (int f = 0; f < FW; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
  if (wv == 0) {  // the bias rides in frequency 1 (point p = 1)
    const float b0 = g.bias[ncol];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[1][r] = b0;
  }
  const unsigned boff = (unsigned)((kh * g.cout_pad + ncol) * 16);
  const unsigned fstep = (unsigned)(CG * g.cout_pad * 16);  // bytes per frequency block
  const unsigned kstep = (unsigned)NFQ * fstep;             // bytes per ky
  const int nchunks = A.cin / CK;
  unsigned wso = (unsigned)(wv * FW) * fstep;               // this wave's frequencies of the next ky to fetch
  // (the prefetch runs PF whole ky steps = 3 x 12 frequency blocks ahead - more than the slack behind the packed
  //  filters: past the last step it re-reads the last one)
  const unsigned wso_max = wso + (unsigned)(nchunks * 7 - 1) * kstep;
  float4 bs[NSETS][FW];

#pragma unroll
  for (int k = 0; k < NI; ++k)
#pragma unroll
    for (int n = 0; n < NFQ; ++n) X.load_piece(0, k, n);
#pragma unroll
  for (int k = 0; k < NI; ++k)
#pragma unroll
    for (int gi = 0; gi < NP + 2; ++gi) X.tgroup(V4, k, gi);
  {
    const int c1 = min(1, nchunks - 1);
#pragma unroll
    for (int k = 0; k < NI; ++k)
#pragma unroll
      for (int n = 0; n < NFQ; ++n) X.load_piece(c1, k, n);
  }
#pragma unroll
  for (int s = 0; s < PF; ++s) {
#pragma unroll
    for (int f = 0; f < FW; ++f) bs[s][f] = bload_f4(rw, boff, wso + f * fstep);
    wso = min(wso + kstep, wso_max);
  }
  __syncthreads();

#define RTPOSE_PIN()             \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)
  // One step = one ky: the wave's FW frequencies x 4 MFMAs on FW alternating accumulators, a filler slot after each
  // round of FW: A fragments of the next step, B fragments PF steps ahead, transform groups / segment loads.
  constexpr int NG = (NP + 2) * NI;
  static_assert(NG <= 8 * NI && NFQ <= 12 && NI <= 2, "transform work does not fit the slots of a chunk");
  float4 a[2][FW];
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const float4* va = V4 + (chunk & 1) * VB + abase;
    float4* vw = V4 + ((chunk + 1) & 1) * VB;
    const int c2 = min(chunk + 2, nchunks - 1);
#pragma unroll
    for (int f = 0; f < FW; ++f) a[0][f] = va[f * CG * GX];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int f = 0; f < FW; ++f) {
          const float4 av = a[ky & 1][f], bv = bs[ky % NSETS][f];
          const float avv[4] = {av.x, av.y, av.z, av.w}, bvv[4] = {bv.x, bv.y, bv.z, bv.w};
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(avv[j], bvv[j], acc[f], 0, 0, 0);
        }
        RTPOSE_PIN();
        if (j == 0) {
          if (ky + 1 < 7) {
#pragma unroll
            for (int f = 0; f < FW; ++f) a[(ky + 1) & 1][f] = va[(ky + 1) * RS + f * CG * GX];
          }
        } else if (j == 1) {
#pragma unroll
          for (int f = 0; f < FW; ++f) bs[(ky + PF) % NSETS][f] = bload_f4(rw, boff, wso + f * fstep);
          wso = min(wso + kstep, wso_max);
        } else {  // j = 2, 3: steps 0..3 the transform groups of the next chunk (they read the segment registers),
                  // steps 4..6 - only then - the segment loads of the chunk after that (they overwrite them)
          const int slot = 2 * ky + (j - 2);  // 0 .. 13
#pragma unroll
          for (int q = 0; q < NI; ++q) {
            if (ky < 4) {
              const int gq = slot * NI + q;
              if (gq < NG) X.tgroup(vw, gq / (NP + 2), gq % (NP + 2));
            } else {
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const int l = ((slot - 8) * NI + q) * 2 + h;
                if (l < NFQ * NI) X.load_piece(c2, l / NFQ, l % NFQ);
              }
            }
          }
        }
        RTPOSE_PIN();
      }
    }
    __syncthreads();
  }
#undef RTPOSE_PIN

  // ---- exchange: every wave needs all 12 frequencies of the accumulator registers it stores --------------
  // LDS (the V buffers are free now): E[f][r][lane]; wave w stores registers r = 4 w .. 4 w + 3, i.e. positions
  // 8 w + 4 kh + (0..3) of the block
  float* E = reinterpret_cast<float*>(V4);
#pragma unroll
  for (int f = 0; f < FW; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) E[((wv * FW + f) * 16 + r) * 64 + lane] = acc[f][r];
  __syncthreads();
  const bool col_ok = ncol < g.cout;
  float* out_base = g.out + g.out_choff + ncol;
  int tcur = t0 + 8 * wv + 4 * kh;
  int sn = tcur / PI, sy, sx;
  {
    const int r = tcur - sn * PI;
    sy = r / GX;
    sx = r - sy * GX;
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = 4 * wv + rr;
    float m[NFQ];
#pragma unroll
    for (int f = 0; f < NFQ; ++f) m[f] = E[(f * 16 + r) * 64 + lane];
    float S[NP], D[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      S[p] = m[2 * p + 1] + m[2 * p + 2];
      D[p] = m[2 * p + 1] - m[2 * p + 2];
    }
    float y[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      float v = (i & 1) ? D[0] : S[0];
      float pw[NP];
#pragma unroll
      for (int p = 1; p < NP; ++p) {
        pw[p] = 1.f;
        for (int e = 0; e < i; ++e) pw[p] *= T::kPts[p];
        v = __builtin_fmaf(pw[p], (i & 1) ? D[p] : S[p], v);
      }
      if (i == 0) v += m[0];
      if (i == FM - 1) v += m[NFQ - 1];
      y[i] = v;
    }
    if (col_ok && tcur < tlim) {
      const size_t q = (size_t)g.out_lead + (size_t)(sn * g.out_hs + sy) * g.out_ws + FM * sx;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const float v = A.relu ? fmaxf(y[i], 0.f) : y[i];
        if (FM * sx + i < A.W) out_base[(q + i) * g.out_cstride] = v;
      }
    }
    ++tcur;
    if (++sx >= GX) {
      sx = 0;
      if (++sy >= A.H) {
        sy = 0;
        ++sn;
      }
    }
  }
}

// ---- weight packing: U[ky][f] = sum_kx G[f][kx] w[ky][kx];  packed[chunk][ky][f][cg][cout_pad][4] ------------
__global__ void pack_wino7_kernel(const float* __restrict__ w, const float* __restrict__ bias, int cout,
                                  int cin_src, const int32_t* __restrict__ cin_map, int cin_packed, int coutp,
                                  int nfq, float* __restrict__ wp, float* __restrict__ bp) {
  const size_t total = (size_t)7 * nfq * cin_packed * coutp;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)coutp) bp[i] = (i < (size_t)cout && bias) ? bias[i] : 0.f;
  if (i >= total) return;
  const int e = i & 3;
  size_t r = i >> 2;
  const int n = r % coutp;
  r /= coutp;
  const int cg = r % CG;
  r /= CG;
  const int f = r % nfq;
  r /= nfq;
  const int ky = r % 7;
  const int chunk = r / 7;
  const int c = chunk * CK + cg * 4 + e;
  const int src = cin_map ? cin_map[c] : (c < cin_src ? c : -1);
  float v = 0.f;
  if (n < cout && src >= 0 && src < cin_src) {
    const float* gw = w + (((size_t)n * cin_src + src) * 7 + ky) * 7;
    if (f == nfq - 1) {
      v = gw[6];
    } else {
      // G[f][kx] = p_f^kx / N_f, N_f = prod_{l != f} (p_f - p_l) over the nfq - 1 finite points (in double)
      const double pts[11] = {0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, 1.5, -1.5, 2.0 / 3.0, -2.0 / 3.0};
      double nf = 1.0;
      for (int l = 0; l < nfq - 1; ++l)
        if (l != f) nf *= pts[f] - pts[l];
      double s = 0.0, pw = 1.0;
      for (int kx = 0; kx < 7; ++kx) {
        s += pw * (double)gw[kx];
        pw *= pts[f];
      }
      v = (float)(s / nf);
    }
  }
  wp[i] = v;
}

// Which F(FM, 7) a conv runs in is a property of the launch (rtpose_conv_desc.wino_m; the rtpose_vgg executor
// chooses per plan and per layer, csrc/net.hip).  0 = the default: 6, or 4 with RTPOSE_WINOGRAD7_M=4 in the
// environment of the process (read once).
int wino7_default_fm() {
  static int fm = 0;
  if (!fm) {
    const char* e = getenv("RTPOSE_WINOGRAD7_M");
    fm = (e && e[0] == '4') ? 4 : 6;
  }
  return fm;
}
static int resolve_fm(int fm) { return fm == 0 ? wino7_default_fm() : fm; }

struct Plan {
  int fm, nfq, gx, rs, nrows, ni, tpi, mtiles;
  long T;
  size_t lds;
};

static int make_plan(int N, int H, int W, int hs, int fm, Plan* p) {
  p->fm = fm;
  p->nfq = p->fm + 6;
  p->gx = ceil_div(W, p->fm);
  p->rs = row_stride(p->gx, p->nfq);
  p->T = (long)N * H * p->gx;
  if (p->T > 0x7fffffffL) return -1;
  const int pi = H * p->gx;
  // maps with >= 256 positions: strips of 32 positions restart with every image (<= 6 % of padded positions;
  // a strip then never carries the gap rows between two images: fewer transformed rows, less transform work);
  // tiny maps: one flat strip space over the whole batch
  p->tpi = pi >= 256 ? ceil_div(pi, 32) : 0;
  int nrows = 0;
  if (p->tpi) {
    nrows = strip_rows(p->gx) < H + 6 ? strip_rows(p->gx) : H + 6;
    p->mtiles = N * p->tpi;
  } else {
    for (long t0 = 0; t0 < p->T; t0 += 32) {
      const long t1 = t0 + 31 < p->T - 1 ? t0 + 31 : p->T - 1;
      const int n0 = (int)(t0 / pi), y0 = (int)((t0 - (long)n0 * pi) / p->gx);
      const int n1 = (int)(t1 / pi), y1 = (int)((t1 - (long)n1 * pi) / p->gx);
      const int rows = (n1 * hs + y1) - (n0 * hs + y0) + 7;
      if (rows > nrows) nrows = rows;
    }
    p->mtiles = (int)((p->T + 31) / 32);
  }
  p->nrows = nrows;
  p->ni = ceil_div(nrows * p->gx * CG, 256);
  p->lds = (size_t)2 * nrows * p->rs * 16;
  return 0;
}

template <int NI, int GXT, int FM, int FS = 1>
static int launch_inst(const Args& a, dim3 grid, size_t lds, hipStream_t s) {
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  auto kern = wino7_f32<NI, GXT, FM, FS>;
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set.set(dev);
  }
  // FS = 2: the accumulator exchange of the epilogue (4 wave tiles x 2 halves x 6 frequencies x 8 registers x 64
  // lanes) lives where the V buffers were
  const size_t ex = FS == 2 ? (size_t)4 * 2 * (FM + 6) / 2 * 8 * 64 * sizeof(float) : 0;
  hipLaunchKernelGGL(kern, grid, dim3(256 * FS), lds > ex ? lds : ex, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

template <int NI>
static int launch_small(const Args& a, dim3 grid, size_t lds, hipStream_t s) {
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  auto kern = wino7s_f32<NI>;
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set.set(dev);
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- amplification estimate of a filter bank in a Winograd form ("numerics") ----------------
// For inputs of uniform magnitude X the products of output channel o sum, in absolute value, to
//   direct:    X * sum_{c,ky,kx} |w[o][c][ky][kx]|
//   Winograd:  X * max_i sum_f |AT[i][f]| * (sum_n |BT[f][n]|) * sum_{c,ky} |U[ky][f][c][o]|
// and the rounding error of either sum is bounded by (a depth factor) x 2^-24 x that quantity.  amp = the worst
// ratio of the two over the output channels: how much larger the element-wise error BOUND of the form is than the
// direct sum's for these filters (i.i.d. Gaussian filters: F(2x2,3x3) 3.3, F(4,7) 62, F(6,7) 115; measured errors stay
// below the bound, tests/test_wino_numerics_gpu.py).  One block per output channel, result by atomicMax on the bit
// pattern of a non-negative float.
__global__ void wino_amp_kernel(const float* __restrict__ w, int cout, int cin, int k, int fm, float* __restrict__ amp) {
  __shared__ float red[17][256];
  const int o = blockIdx.x, tid = threadIdx.x;
  float sf[16], den = 0.f;
#pragma unroll
  for (int f = 0; f < 16; ++f) sf[f] = 0.f;
  const int nfq = fm + 6;
  const double pts[11] = {0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, 1.5, -1.5, 2.0 / 3.0, -2.0 / 3.0};
  // N_f = prod_{l != f} (p_f - p_l): once per block (round 6; every thread used to rebuild it for every filter row - the
  // 117 launches of a weight load took 8.3 ms), by the same multiplications in the same order
  __shared__ double s_nf[12];
  if (k == 7 && tid < nfq - 1) {
    double nf = 1.0;
    for (int l = 0; l < nfq - 1; ++l)
      if (l != tid) nf *= pts[tid] - pts[l];
    s_nf[tid] = nf;
  }
  __syncthreads();
  for (int c = tid; c < cin; c += 256) {
    const float* gw = w + ((size_t)o * cin + c) * k * k;
    if (k == 3) {
      float u[4][3];
      for (int x = 0; x < 3; ++x) {
        const float g0 = gw[x], g1 = gw[3 + x], g2 = gw[6 + x];
        u[0][x] = g0;
        u[1][x] = 0.5f * (g0 + g1 + g2);
        u[2][x] = 0.5f * (g0 - g1 + g2);
        u[3][x] = g2;
        den += fabsf(g0) + fabsf(g1) + fabsf(g2);
      }
      for (int fy = 0; fy < 4; ++fy) {
        sf[fy * 4 + 0] += fabsf(u[fy][0]);
        sf[fy * 4 + 1] += fabsf(0.5f * (u[fy][0] + u[fy][1] + u[fy][2]));
        sf[fy * 4 + 2] += fabsf(0.5f * (u[fy][0] - u[fy][1] + u[fy][2]));
        sf[fy * 4 + 3] += fabsf(u[fy][2]);
      }
    } else {
      for (int ky = 0; ky < 7; ++ky) {
        const float* row = gw + ky * 7;
        for (int kx = 0; kx < 7; ++kx) den += fabsf(row[kx]);
        for (int f = 0; f < nfq; ++f) {
          double v;
          if (f == nfq - 1) {
            v = row[6];
          } else {
            const double nf = s_nf[f];
            double sacc = 0.0, pw = 1.0;
            for (int kx = 0; kx < 7; ++kx) {
              sacc += pw * (double)row[kx];
              pw *= pts[f];
            }
            v = sacc / nf;
          }
          sf[f] += fabsf((float)v);
        }
      }
    }
  }
#pragma unroll
  for (int f = 0; f < 16; ++f) red[f][tid] = sf[f];
  red[16][tid] = den;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st)
      for (int f = 0; f < 17; ++f) red[f][tid] += red[f][tid + st];
    __syncthreads();
  }
  if (tid == 0) {
    float num = 0.f;
    if (k == 3) {
      const float a[2][4] = {{1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, 1.f, 1.f}};  // |AT|; every row of |BT| sums to 2
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
          float v = 0.f;
          for (int fy = 0; fy < 4; ++fy)
            for (int fx = 0; fx < 4; ++fx) v += a[i][fy] * a[j][fx] * 4.f * red[fy * 4 + fx][0];
          num = fmaxf(num, v);
        }
    } else {
      const float pabs[12] = {0.f, 1.f, 1.f, 2.f, 2.f, 0.5f, 0.5f, 1.5f, 1.5f, 0.666666667f, 0.666666667f, 0.f};
      for (int i = 0; i < fm; ++i) {
        float v = 0.f;
        for (int f = 0; f < nfq; ++f) {
          float b = 0.f;  // sum_n |BT[f][n]| of the kernel's (row-scaled) table
          for (int n = 0; n < nfq; ++n) b += fabsf(fm == 4 ? WT<4>::kBT[f < 10 ? f : 0][n < 10 ? n : 0] : WT<6>::kBT[f][n]);
          float ai;  // |AT[i][f]| = |p_f|^i; the point 0 only reaches output 0, infinity only output fm - 1
          if (f == 0) ai = i == 0 ? 1.f : 0.f;
          else if (f == nfq - 1) ai = i == fm - 1 ? 1.f : 0.f;
          else {
            ai = 1.f;
            for (int e = 0; e < i; ++e) ai *= pabs[f];
          }
          v += ai * b * red[f][0];
        }
        num = fmaxf(num, v);
      }
    }
    const float r = red[16][0] > 0.f ? num / red[16][0] : 0.f;
    atomicMax(reinterpret_cast<int*>(amp), __float_as_int(r));
  }
}

}  // namespace wino7

int wino7_default_fm() { return wino7::wino7_default_fm(); }

// Hand-over scratch of the persistent launches: [flags: blocks + 1 ints][error word][pad to 256 B][one accumulator
// tile per block].  It is the CALLER's memory (the rtpose_vgg executor places it in its workspace; a caller of
// rtpose_conv2d_winograd_ex passes it): the forward path allocates nothing and can be stream-captured.  One scratch
// serves launches that are serialised on one stream; concurrent launches need one each.
static size_t wino7_scratch_flag_bytes(int blocks) { return round_up((size_t)(blocks + 2) * sizeof(int), 256); }
int* conv2d_wino7_scratch_err(void* scratch, int blocks) { return static_cast<int*>(scratch) + blocks + 1; }
size_t conv2d_wino7_scratch_bytes(int blocks) {
  return wino7_scratch_flag_bytes(blocks) + (size_t)blocks * (4 * 12 * 16 * 64) * sizeof(float);
}

// 1 when the 7x7 conv can run in F(fm, 7) form at this geometry (a kernel instance exists and the
// transformed rows of a block fit the LDS), else 0: callers then use the direct kernel
int conv2d_wino7_fits(int cin, int cout, int N, int H, int W, int hs, int fm) {
  fm = wino7::resolve_fm(fm);
  if ((fm != 4 && fm != 6) || cin <= 0 || cin % wino7::CK || cout_pad(cout) % 128 || N <= 0 || H <= 0 || W <= 0) return 0;
  wino7::Plan p;
  if (wino7::make_plan(N, H, W, hs, fm, &p)) return 0;
  return p.ni <= 2 && p.lds <= 156 * 1024;
}

// MFMA flops a launch ISSUES (what SQ_INSTS_MFMA x 4096 counts): every strip is 32 positions, padded per image
double conv2d_wino7_issued_flops(int cin, int cout, int N, int H, int W, int hs, int fm) {
  fm = wino7::resolve_fm(fm);
  wino7::Plan p;
  if (wino7::make_plan(N, H, W, hs, fm, &p)) return 0.0;
  return 2.0 * p.mtiles * 32.0 * 7.0 * p.nfq * (double)cin * cout_pad(cout);
}

int conv2d_wino7_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, int fm, void* scratch,
                        size_t scratch_bytes, hipStream_t s) {
  using namespace wino7;
  fm = resolve_fm(fm);
  if (!d || ngroups < 1 || ngroups > 2) return fail(RTPOSE_E_INVAL, "conv2d_winograd: ngroups must be 1 or 2");
  RTPOSE_REFUSE_PLANES(d, ngroups, "conv2d_winograd (7x7)");
  const rtpose_conv_desc& d0 = d[0];
  if (d0.k != 7 || d0.pool || !conv2d_wino7_fits(d0.cin, d0.cout, N, H, W, d0.lin.hs, fm))
    return fail(RTPOSE_E_INVAL, "conv2d_winograd: no F(%d,7) instance for cin %d cout %d at %d x %d x %d", fm, d0.cin,
                d0.cout, N, H, W);
  Args a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < ngroups; ++i) {
    const rtpose_conv_desc& di = d[i];
    if (di.k != 7 || di.cin != d0.cin || di.relu != d0.relu || di.pool ||
        cout_pad(di.cout) != cout_pad(d0.cout) || di.lin.ws != d0.lin.ws || di.lin.hs != d0.lin.hs)
      return fail(RTPOSE_E_INVAL, "conv2d_winograd: grouped convs must share geometry");
    if (di.lin.ws < W + 3 || di.lin.hs < H + 3 || di.lin.lead < 3 * di.lin.ws + 3)
      return fail(RTPOSE_E_INVAL, "conv2d_winograd: input layout gap smaller than the conv padding");
    if ((di.lin.cstride % 4) || (di.lin.choff % 4))
      return fail(RTPOSE_E_INVAL, "conv2d_winograd: input slice must be 16-byte aligned");
    if (di.lin.choff + di.cin > di.lin.cstride)
      return fail(RTPOSE_E_INVAL, "conv2d_winograd: input slice exceeds cstride");
    if (di.out_cmap) return fail(RTPOSE_E_INVAL, "conv2d_winograd: out_cmap is not supported");
    Group& g = a.g[i];
    g.in = di.in;
    g.w = di.w_packed;
    g.bias = di.bias_packed;
    g.out = di.out;
    g.in_cstride = di.lin.cstride;
    g.in_choff = di.lin.choff;
    g.in_ws = di.lin.ws;
    g.in_hs = di.lin.hs;
    g.in_lead = di.lin.lead;
    g.out_cstride = di.lout.cstride;
    g.out_choff = di.lout.choff;
    g.out_ws = di.lout.ws;
    g.out_hs = di.lout.hs;
    g.out_lead = di.lout.lead;
    g.cout = di.cout;
    g.cout_pad = cout_pad(di.cout);
    g.in_bytes = rtpose_layout_pixels(&di.lin, N, H, W) * (size_t)di.lin.cstride * sizeof(float);
    g.w_bytes = packed_weight_floats_wino7(di.cout, di.cin, fm) * sizeof(float);
  }
  Plan p;
  make_plan(N, H, W, d0.lin.hs, fm, &p);
  a.N = N;
  a.H = H;
  a.W = W;
  a.GX = p.gx;
  a.T = (int)p.T;
  a.cin = d0.cin;
  a.relu = d0.relu;
  a.RS = p.rs;
  a.VB = p.nrows * p.rs;
  a.TPI = p.tpi;
  a.mtiles = p.mtiles;
  a.ntiles = cout_pad(d0.cout) / 128;
  a.ncombo = a.ntiles * ngroups;
  a.xcd_remap = (a.ncombo > 1 && a.mtiles >= 64) ? 1 : 0;
  long ids = a.xcd_remap ? (long)8 * a.ncombo * ceil_div(a.mtiles, 8) : (long)a.mtiles * a.ncombo;
  if (ids > 0x7fffffffL) return fail(RTPOSE_E_INVAL, "conv2d_winograd: grid too large");
  {
    // persistent form: as many blocks as CUs share the (tile, chunk) units evenly; worth it (and valid: a tile
    // may be split between at most two blocks) when there are at least as many tiles as CUs and the tiles do
    // not already come out as whole rounds.  It needs the caller's hand-over scratch; without one (or with one
    // that is too small for this device) the launch runs one block per tile - same results, bit for bit.
    const int n_cu = device_cu_count();
    const long tiles = (long)a.mtiles * a.ncombo;
    static int persist_env = -1;
    if (persist_env < 0) {
      const char* e = dev_env("RTPOSE_W7_PERSIST");
      persist_env = e ? atoi(e) : 1;
    }
    if (persist_env && tiles >= n_cu && tiles % n_cu != 0 && scratch && scratch_bytes >= conv2d_wino7_scratch_bytes(n_cu)) {
      if ((uintptr_t)scratch & 255) return fail(RTPOSE_E_INVAL, "conv2d_winograd: scratch must be 256-byte aligned");
      a.flags = static_cast<int*>(scratch);
      a.err = a.flags + n_cu + 1;
      a.scratch = reinterpret_cast<float*>(static_cast<char*>(scratch) + wino7_scratch_flag_bytes(n_cu));
      // flags are cleared at launch start (not only by the consumer): an aborted launch cannot leave one up
      RTPOSE_HIP_CHECK(hipMemsetAsync(a.flags, 0, (size_t)(n_cu + 1) * sizeof(int), s));
      a.persist = 1;
      ids = n_cu;
    }
  }
  const dim3 grid((unsigned)ids, 1, 1);
  // small grids: the frequency-split form (wino7s_f32), bit-identical, 4 x the blocks
  static int small_env = -1;
  if (small_env < 0) {
    const char* e = dev_env("RTPOSE_W7_SMALL");
    small_env = e ? atoi(e) : 1;
  }
  if (small_env && p.fm == 6 && !a.persist && (long)a.mtiles * a.ncombo * 2 <= device_cu_count()) {
    Args b = a;
    b.ntiles = cout_pad(d0.cout) / 32;
    b.ncombo = b.ntiles * ngroups;
    const dim3 gs((unsigned)((long)b.mtiles * b.ncombo), 1, 1);
    const size_t lds_s = p.lds > (size_t)12 * 16 * 64 * 4 ? p.lds : (size_t)12 * 16 * 64 * 4;  // V buffers, then the exchange
    return p.ni == 1 ? launch_small<1>(b, gs, lds_s, s) : launch_small<2>(b, gs, lds_s, s);
  }
  // 46-wide maps (368 x 368 inputs, BASELINE configs[1]): every LDS offset of the multiply loop is an immediate
  if (p.fm == 6) {
    // two waves per SIMD (8-wave blocks, the frequencies split between sibling waves: wino7_segment FS = 2) wherever
    // a thread transforms one item per chunk; bit-identical to the 4-wave form (RTPOSE_W7_FS=1 in developer builds)
    static int fs_env = -1;
    if (fs_env < 0) {
      const char* e = dev_env("RTPOSE_W7_FS");
      fs_env = e ? atoi(e) : 2;
    }
    if (p.gx == 8 && p.tpi && p.ni == 1 && p.nrows == strip_rows(8)) {
      const size_t lds8 = (size_t)2 * strip_rows(8) * row_stride(8, 12) * 16;
      return fs_env == 2 ? launch_inst<1, 8, 6, 2>(a, grid, lds8, s) : launch_inst<1, 8, 6>(a, grid, lds8, s);
    }
    if (p.ni == 1) return fs_env == 2 ? launch_inst<1, 0, 6, 2>(a, grid, p.lds, s) : launch_inst<1, 0, 6>(a, grid, p.lds, s);
    return launch_inst<2, 0, 6>(a, grid, p.lds, s);
  }
  if (p.gx == 12 && p.tpi && p.ni == 1 && p.nrows == strip_rows(12))
    return launch_inst<1, 12, 4>(a, grid, (size_t)2 * strip_rows(12) * row_stride(12, 10) * 16, s);
  if (p.ni == 1) return launch_inst<1, 0, 4>(a, grid, p.lds, s);
  return launch_inst<2, 0, 4>(a, grid, p.lds, s);
}

int pack_weights_wino7_launch(const float* w, const float* bias, int cout, int cin_src, const int32_t* cin_map,
                              int cin_packed, int fm, float* wp, float* bp, hipStream_t s) {
  fm = wino7::resolve_fm(fm);
  if (fm != 4 && fm != 6) return fail(RTPOSE_E_INVAL, "pack_winograd: F(m,7) exists for m = 4 and m = 6");
  if (cin_packed % wino7::CK || cin_packed <= 0 || (cin_packed < cin_src && !cin_map))
    return fail(RTPOSE_E_INVAL, "pack_winograd: cin_packed must be a multiple of 8 and >= cin_src");
  const int coutp = cout_pad(cout);
  const int nfq = fm + 6;
  const size_t total = (size_t)7 * nfq * cin_packed * coutp;
  const int threads = 256;
  const unsigned blocks = (unsigned)((total + threads - 1) / threads);
  hipLaunchKernelGGL(wino7::pack_wino7_kernel, dim3(blocks), dim3(threads), 0, s, w, bias, cout, cin_src, cin_map,
                     cin_packed, coutp, nfq, wp, bp);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

size_t packed_weight_floats_wino7(int cout, int cin, int fm) {
  // + 5 steps (10 frequency blocks of 8 x cout_pad floats): the B prefetch runs up to five steps ahead
  return (size_t)(7 * (wino7::resolve_fm(fm) + 6) * cin + 96) * cout_pad(cout);
}

// amp (device, one float) <- amplification estimate of w[cout][cin][k][k] in F(2x2,3x3) (k = 3) or F(fm,7) (k = 7)
int wino_amplification_launch(const float* w, int cout, int cin, int k, int fm, float* amp, hipStream_t s) {
  if (k == 7) fm = wino7::resolve_fm(fm);
  if (!w || !amp || cout <= 0 || cin <= 0 || !(k == 3 || (k == 7 && (fm == 4 || fm == 6))))
    return fail(RTPOSE_E_INVAL, "winograd_amplification: k must be 3, or 7 with m = 4 or 6");
  RTPOSE_HIP_CHECK(hipMemsetAsync(amp, 0, sizeof(float), s));
  hipLaunchKernelGGL(wino7::wino_amp_kernel, dim3((unsigned)cout), dim3(256), 0, s, w, cout, cin, k, k == 7 ? fm : 0, amp);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace rtpose
