// One ShuffleNetV2 unit as ONE launch in the bf16 plan (BASELINE configs[3]):
//
//     x2 -> conv_bn_relu 1x1 (conv.0) -> conv_bn depthwise 3x3 (conv.1) -> conv_bn_relu 1x1 (conv.2) -> y
//
// (lib/network/rtpose_shufflenetV2.py:31-39, the `conv` branch of a stride-1 BasicBlock; torch.cat with the
// pass-through half and channel_shuffle :56-62 move no data in the zero-copy slot plan of shufflenet.hip), under the
// contract of oracle/shufflenet_oracle.py:_block_bf16: bf16 activations and pointwise weights, exact products, fp32
// sums, depthwise taps and every bias fp32, each of the three conv outputs rounded to bf16 (RNE).
//
// Rounds 2-4 ran a unit as two launches - conv.0, then depthwise + conv.2 - with the conv.0 output T1 (270 848 pixels x
// 240 channels x 2 bytes = 130 MB in the last stage) written, and read back with a 1.56x halo, per unit: 15 units, 2.5 GB
// of the 11.3 GB a 128-image forward moved, and two launches whose K = 120..240 GEMMs ran at 0.05-0.12 of the matrix peak.
// Here T1 only ever exists in LDS:
//   * WORK ITEM = an 8 x 8 tile of output pixels of one image.  The block stages the tile's 10 x 10 halo of x2 - the
//     16-byte planes the slot plan gathers - in LDS, computes T1 on all 100 halo pixels (GEMM 1: 128 rows, the last 28
//     are replays that are never stored; halo pixels outside the image are written as ZERO - the depthwise conv pads T1
//     with zeros, not with conv.0 of a zero pixel), applies the depthwise 3x3 in fp32 on the VALU from LDS to LDS, and
//     multiplies the result by conv.2's matrix (GEMM 2: 64 rows).
//   * Both GEMMs are computed TRANSPOSED (rows = channels: the filter fragment is the MFMA's A operand, straight from L2
//     through a buffer load; columns = pixels: the activation fragment is one ds_read_b128 per lane): a lane then holds,
//     for ITS pixel, groups of four consecutive channels - 8 bytes of the [plane][pixel][8 channels] LDS tile after
//     GEMM 1, 8 bytes of the output pixel after GEMM 2 - so neither result needs a transposition.
//   * The four waves split the CHANNELS of both GEMMs; a filter fragment feeds four MFMAs in GEMM 1, two in GEMM 2.
//   * One block per CU, two halo-tile buffers in LDS: the tile being worked on turns into T1 in place, the other
//     receives the block's NEXT tile, requested two 64-channel chunks at a time under GEMM 1 and under every depthwise /
//     GEMM 2 chunk of the current one (through a zero-extent descriptor where there is nothing left to fetch - no load
//     sits under a branch).  (Two blocks of 256 registers each per CU were tried first: the allocator spilled around
//     the 128 accumulators of GEMM 1.)
// y is stored into slots of the stage buffer that are FREE during the launch (zc_plan(..., y_beside_x2): other blocks
// still read x2 as their halo).  Sum order: K ascending in both GEMMs, the nine taps in row-major order after the bias
// (the order of pw_fused_bf16.hip) - fixed, independent of the batch.
#include <hip/hip_runtime.h>

#include "common.h"
#include "conv_exp.h"
#include "wino_common.h"

namespace rtpose {

namespace unitb {

using winoc::i32x4;
using winoc::make_rsrc;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float4 bload4(i32x4 r, unsigned voff, unsigned soff) {
  const winoc::f32x4 v = winoc::llvm_raw_buffer_load_v4f32(r, (int)voff, (int)soff, 0);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ void llvm_raw_buffer_store_v4f32(winoc::f32x4 v, i32x4 rsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.store.v4f32");
__device__ __forceinline__ void bstore4(const float4& v, i32x4 r, unsigned voff) {
  const winoc::f32x4 t = {v.x, v.y, v.z, v.w};
  llvm_raw_buffer_store_v4f32(t, r, (int)voff, 0, 0);
}
__device__ __forceinline__ bf16x8 as_bf8(const float4& v) {
  const floatx4 t = {v.x, v.y, v.z, v.w};
  return __builtin_bit_cast(bf16x8, t);
}
__device__ __forceinline__ float acc_read(float a) {  // one accumulator register -> a VGPR, here (see pw_head_bf16.hip)
  float v;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}
// The compiler's hazard recogniser does not know that the asm above reads an MFMA result: the wait states between the last
// MFMA that wrote an accumulator and its first v_accvgpr_read (up to 18 for a 16-pass 32x32x16) are inserted by hand,
// once, in front of every read-out section.  (Found the hard way: registers 0 and 1 of the first fragment read stale
// values in one instance of unit_bf16_kernel whose epilogue followed the last MFMA directly.)
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
__device__ __forceinline__ float relu_bits(float v) {
  return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0));
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) {  // RNE (v_cvt_pk_bf16_f32)
  const bf16x2 t = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ void unpack8(const float4& p, float* f) {  // 8 bf16 -> 8 floats
  const unsigned u[4] = {__float_as_uint(p.x), __float_as_uint(p.y), __float_as_uint(p.z), __float_as_uint(p.w)};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(u[i] << 16);
    f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
  }
}

constexpr int kTile = 8;               // output tile edge
constexpr int kHalo = kTile + 2;       // halo edge
constexpr int kNHP = kHalo * kHalo;    // 100 halo pixels
constexpr int XP1 = 106;               // LDS plane pitch (float4) of the halo tiles: >= 100, == 2 (mod 8)
constexpr int AP2 = 66;                // LDS plane pitch of the 64-pixel depthwise output chunk: == 2 (mod 8)
constexpr int kMaxK1 = 256;            // widest gather of x2 (32 planes: four staging chunks)
constexpr int kMaxKt = 256;            // widest T1

struct Args {
  const unsigned short* in;  // the stage buffer (bf16), gap >= 1
  size_t in_bytes;
  int in_cstride, in_choff, in_ws, in_hs, in_lead;
  const int32_t* in_planes;  // [K1 / 8]: element offset (inside the pixel slice) of every 8-channel plane of x2
  const void* w0;            // conv.0: [K1 / 8][C1P][8 bf16], columns = T1 channels, zero columns past Kt
  const float* b0;           // [C1P]
  const float* dw_w;         // conv.1: fp32 [9][Kt]
  const float* dw_b;         // fp32 [Kt]
  const void* w2;            // conv.2: [Kt / 8][C2P][8 bf16], columns in the order of out_cmap's groups
  const float* b2;           // [C2P]
  unsigned short* out;       // the stage buffer again (other slots)
  size_t out_bytes;
  int out_cstride, out_ws, out_hs, out_lead;
  const int32_t* out_cmap;   // [C2P]: absolute channel of a column; groups of 8 columns are contiguous; < 0: not stored
  int N, H, W;
  int K1, Kt, cout;          // cout: columns of conv.2 that exist (<= C2P, a multiple of 8)
  int tiles_x, tiles_y, nitems;
  FastDiv ftx, fty;
  unsigned long long* dbg;   // developer builds (-DRTPOSE_EXP_TIMELINE_UNIT): s_memtime stamps [block][tile < 20][16]
};

#ifdef RTPOSE_EXP_TIMELINE_UNIT
#define RTPOSE_UB_STAMP(K)                                                             \
  if (A.dbg && wave == 0 && lane == 0 && tcount < 20)                                  \
  A.dbg[((size_t)blockIdx.x * 20 + tcount) * 16 + (K)] = __builtin_amdgcn_s_memtime()
#else
#define RTPOSE_UB_STAMP(K)
#endif

#define RTPOSE_UB_PIN()          \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)

template <int V>
struct IntTag {
  static constexpr int value = V;
};

// 512 threads = 8 waves, two per SIMD: the depthwise conv, the two write-outs and the stores are VALU / LDS work that ONE
// wave per SIMD issues at ~6 cycles per instruction (measured: 24 k of a 31 k-cycle tile with 4 waves, the two GEMMs at
// the power-limited MFMA rate taking the rest); a second wave on the SIMD fills the issue slots of the first.
// WM1 = 1: the waves split T1's 256 channels eight ways (32 each) and every wave multiplies all four pixel fragments;
// WM1 = 2: T1 has 128 channels - four channel slices x two halves of the pixel fragments.  WM2 likewise for conv.2's
// columns (C2P = 256 / WM2) and the two pixel fragments of the output tile.  NCH2: 64-channel chunks of T1.
template <int WM1, int WM2, int NCH2>
__global__ __launch_bounds__(512, 1) void unit_bf16_kernel(const Args A) {
  constexpr int C1P = 256 / WM1, C2P = 256 / WM2;
  constexpr int MF1 = 4 / WM1;  // pixel fragments of GEMM 1 per wave
  constexpr int MF2 = 2 / WM2;  // pixel fragments of GEMM 2 per wave
  extern __shared__ __attribute__((aligned(16))) float4 smem4[];
  __shared__ int s_plane[kMaxK1 / 8];
  __shared__ __attribute__((aligned(16))) float s_b0[256], s_b2[256];  // both biases: an accumulator init must not queue
                                                                        // behind the HBM requests of the next tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int wc1 = wave % (8 / WM1), wm1 = wave / (8 / WM1);  // GEMM 1: channel slice, pixel-fragment half
  const int wc2 = wave % (8 / WM2), wm2 = wave / (8 / WM2);  // GEMM 2 likewise
  const int npl1 = A.K1 >> 3;                 // planes of x2
  const int nchk = (npl1 + 7) >> 3;           // staging chunks (8 planes) of x2
  const int k1real = A.K1 >> 4;               // k-steps of GEMM 1 that carry channels
  const int kst1 = (k1real + 3) & ~3;         // ... that the loop walks (whole turns of the four-slot filter ring)
  const int nplt = A.Kt >> 3;                 // planes of T1
  const int k2real = A.Kt >> 4;               // k-steps of GEMM 2 (<= 4 NCH2)
  const int nr1 = max(npl1, nplt);
  // LDS: [2][nr1][XP1] x2 halo tiles, ping-pong (the tile being worked on becomes T1, then y, in place; the other one
  // receives the block's next tile) | [nplt][AP2] depthwise output | [10][Kt] depthwise taps + bias
  float4* const a2 = smem4 + 2 * nr1 * XP1;
  float* const dwl = reinterpret_cast<float*>(a2 + nplt * AP2);
  for (int j = tid; j < npl1; j += 512) s_plane[j] = A.in_planes ? A.in_planes[j] : 8 * j;
  for (int i = tid; i < 10 * A.Kt; i += 512) dwl[i] = i < 9 * A.Kt ? A.dw_w[i] : A.dw_b[i - 9 * A.Kt];
  if (tid < C1P) s_b0[tid] = A.b0[tid];
  if (tid < C2P) s_b2[tid] = A.b2[tid];
  __syncthreads();
  int item = blockIdx.x;
  if (item >= A.nitems) return;

  const i32x4 rw0 = make_rsrc(A.w0, (size_t)npl1 * C1P * 16);
  const i32x4 rw2 = make_rsrc(A.w2, (size_t)nplt * C2P * 16);
  const i32x4 rin = make_rsrc(A.in, A.in_bytes);
  const i32x4 rnull = make_rsrc(A.in, 0);
  const i32x4 rout = make_rsrc(A.out, A.out_bytes);
  const unsigned w0_lane = (unsigned)(kh * C1P + 32 * wc1 + l31) * 16u;
  const unsigned w2_lane = (unsigned)(kh * C2P + 32 * wc2 + l31) * 16u;

  // ---- roles that do not depend on the item ---------------------------------------------------------------------------
  // staging: plane spl of an 8-plane chunk, halo pixels spx + 64 u (8 consecutive lanes = 4 planes x 2 pixels, see
  // pw_head_bf16.hip); halo pixel hp = 10 hy + hx
  const int spl = (tid & 3) | (((tid >> 3) & 1) << 2);
  const int spx = ((tid >> 2) & 1) | ((tid >> 4) << 1);
  // GEMM 1: the lane's pixel in the wave's pixel fragment j is halo pixel 32 (MF1 wm1 + j) + l31 (rows past 99 replay 99)
  unsigned x_off[MF1];  // float4 index of the lane's row in plane kh
#pragma unroll
  for (int j = 0; j < MF1; ++j) x_off[j] = (unsigned)(kh * XP1 + min(32 * (MF1 * wm1 + j) + l31, kNHP - 1));

  float4 w0r[4];         // conv.0's filter fragments: ring of four k-steps (requested three steps ahead; the GEMMs run at
                         // the power-limited MFMA rate, a ring of eight changed nothing)
  float4 w2r[4 * NCH2];  // ALL of conv.2's fragments of the wave: requested once per tile, before the next tile's HBM
                         // requests, so that GEMM 2 waits for no load at all
  float4 sr[2][2];       // two chunks of the NEXT item's halo tile in flight under the VALU phases of this one

  auto w0load = [&](float4& dst, int g) {
    const i32x4 r = g < k1real ? rw0 : rnull;  // (a step past K1: nothing fetched, and its MFMAs are skipped)
    dst = bload4(r, w0_lane, (unsigned)(2 * min(g, k1real - 1)) * (unsigned)(C1P * 16));
  };
  struct Tile {
    int n, y0, x0;
  };
  auto tile_of = [&](int it) {
    const int r_ = fast_div(it, A.ftx), tx_ = it - r_ * A.tiles_x;
    const int n = fast_div(r_, A.fty), ty_ = r_ - n * A.tiles_y;
    return Tile{n, ty_ * kTile, tx_ * kTile};
  };
  // byte offsets of the thread's two halo pixels of a tile.  A halo pixel outside the image is read from the layout's
  // zero gap (row -1 / H, column -1 / W: the shared gap of the padded NHWC layout)
  auto stage_setup = [&](const Tile& t, unsigned* q) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int hp = min(spx + 64 * u, kNHP - 1);
      const int hy = (hp * 205) >> 11, hx = hp - 10 * hy;  // hp / 10 for hp < 1024
      const int y = min(max(t.y0 - 1 + hy, -1), A.H), x = min(max(t.x0 - 1 + hx, -1), A.W);
      q[u] = ((unsigned)(A.in_lead + (t.n * A.in_hs + y) * A.in_ws + x) * (unsigned)A.in_cstride + (unsigned)A.in_choff) * 2u;
    }
  };
  // Chunks c0, c0 + 1 of a tile: requested, and later stored, as one round.  The next tile's rounds fly under the
  // VALU-only phases of this one - round 0 from the end of GEMM 1 to the end of the first depthwise chunk, round r from
  // there on under GEMM 2 of chunk r - 1 (which waits for no load) and depthwise chunk r.  Loads complete in order, so
  // an HBM request in flight makes every YOUNGER request wait for it: in the first version (two chunks requested under
  // GEMM 1, biases fetched from global memory per tile) every accumulator init and filter wait sat out an HBM round
  // trip.  (Past the tile's last chunk: the zero-extent descriptor, nothing fetched.)
  auto stage_load = [&](const unsigned* q, int c0) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int c = c0 + cc;
      const i32x4 r = c < nchk ? rin : rnull;
      const unsigned pofs = (unsigned)s_plane[min(8 * c + spl, npl1 - 1)] * 2u;
#pragma unroll
      for (int u = 0; u < 2; ++u) sr[cc][u] = bload4(r, q[u] + pofs, 0);
    }
  };
  auto stage_store = [&](float4* xs, int c0) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int pl = 8 * (c0 + cc) + spl;
      if (pl < npl1) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (spx + 64 * u < kNHP) xs[pl * XP1 + spx + 64 * u] = sr[cc][u];
      }
    }
  };

  // ---- prologue: the block's first tile ---------------------------------------------------------------------------------
  Tile cur = tile_of(item);
  {
    unsigned sq[2];
    stage_setup(cur, sq);
    for (int c = 0; c < nchk; c += 2) {
      stage_load(sq, c);
      stage_store(smem4, c);
    }
  }
  int pp = 0;  // which LDS tile buffer holds the current item
  int tcount = 0;
  (void)tcount;

  while (true) {
    RTPOSE_UB_STAMP(0);
    float4* const xs = smem4 + pp * nr1 * XP1;        // x2 halo tile of this item, then its T1, then its y
    float4* const xn = smem4 + (pp ^ 1) * nr1 * XP1;  // receives the next item's tile
    const int nitem = item + (int)gridDim.x;
    const bool has_next = nitem < A.nitems;
    const Tile nxt = tile_of(has_next ? nitem : item);  // (no next item: this one is staged again - harmless)
    unsigned sqn[2];
    stage_setup(nxt, sqn);

    // ---- GEMM 1 (transposed): T1^T[channel][halo pixel], the wave's 32 channels x 32 MF1 rows -------------------------
#pragma unroll
    for (int i = 0; i < 3; ++i) w0load(w0r[i], i);
    floatx16 acc1[MF1];
    {
      floatx16 t;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 b = *reinterpret_cast<const float4*>(s_b0 + (32 * wc1 + 8 * rg + 4 * kh));
        t[rg * 4 + 0] = b.x;
        t[rg * 4 + 1] = b.y;
        t[rg * 4 + 2] = b.z;
        t[rg * 4 + 3] = b.w;
      }
#pragma unroll
      for (int j = 0; j < MF1; ++j) acc1[j] = t;
    }
    __syncthreads();  // B0: this item's x2 tile is in LDS (staged under the previous item), the previous y tile / a2 are dead
    RTPOSE_UB_STAMP(1);
    {
      float4 xr[2][MF1];  // activation fragments: two k-steps x the wave's pixel fragments
      auto x1load = [&](float4(&dst)[MF1], int g) {
        const float4* p = xs + (unsigned)(2 * min(g, k1real - 1)) * XP1;
#pragma unroll
        for (int j = 0; j < MF1; ++j) dst[j] = p[x_off[j]];
      };
      x1load(xr[0], 0);
      auto step1 = [&](auto i_tag, int g) {
        constexpr int S = decltype(i_tag)::value & 3;
        constexpr int XS = decltype(i_tag)::value & 1;
        w0load(w0r[(S + 3) & 3], g + 3);  // (past K1: nothing fetched)
        x1load(xr[XS ^ 1], g + 1);
        RTPOSE_UB_PIN();
        if (g < k1real) {  // (uniform; no load inside)
#pragma unroll
          for (int j = 0; j < MF1; ++j)
            acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(w0r[S]), as_bf8(xr[XS][j]), acc1[j], 0, 0, 0);
        }
        RTPOSE_UB_PIN();
      };
      for (int g0 = 0; g0 < kst1; g0 += 4) {
        step1(IntTag<0>(), g0);
        step1(IntTag<1>(), g0 + 1);
        step1(IntTag<2>(), g0 + 2);
        step1(IntTag<3>(), g0 + 3);
      }
    }
    RTPOSE_UB_STAMP(2);
    // conv.2's fragments - all of them, from L2 - and BEHIND them the first round of the next tile from HBM: nothing
    // younger is waited for until that round is stored
#pragma unroll
    for (int g = 0; g < 4 * NCH2; ++g) {
      const i32x4 r = g < k2real ? rw2 : rnull;
      w2r[g] = bload4(r, w2_lane, (unsigned)(2 * min(g, k2real - 1)) * (unsigned)(C2P * 16));
    }
    stage_load(sqn, 0);
    __syncthreads();  // B1: every wave has read the x2 tile for the last time
    RTPOSE_UB_STAMP(3);

    // ---- T1 = bf16(relu(.)) -> LDS [plane][halo pixel][8 channels] in the tile's place; zero outside the image --------
    mfma_drain();
#pragma unroll
    for (int j = 0; j < MF1; ++j) {
      const int hp = 32 * (MF1 * wm1 + j) + l31;
      const int hpc = min(hp, kNHP - 1);
      const int hy = (hpc * 205) >> 11, hx = hpc - 10 * hy;
      const int y = cur.y0 - 1 + hy, x = cur.x0 - 1 + hx;
      const bool inside = y >= 0 && y < A.H && x >= 0 && x < A.W;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int pl = 4 * wc1 + rg;  // (uniform)
        const floatx16& a = acc1[j];
        uint2 v;
        v.x = pack2(relu_bits(acc_read(a[rg * 4 + 0])), relu_bits(acc_read(a[rg * 4 + 1])));
        v.y = pack2(relu_bits(acc_read(a[rg * 4 + 2])), relu_bits(acc_read(a[rg * 4 + 3])));
        if (!inside) v = make_uint2(0u, 0u);
        if (pl < nplt && hp < kNHP)
          *reinterpret_cast<uint2*>(reinterpret_cast<char*>(xs + pl * XP1 + hp) + 8 * kh) = v;
      }
      RTPOSE_UB_PIN();
    }
    RTPOSE_UB_STAMP(4);
    __syncthreads();  // B2: T1 is in LDS
    RTPOSE_UB_STAMP(5);

    // ---- depthwise 3x3 (fp32, VALU), all of T1's planes at once.  A thread owns FOUR horizontally adjacent output pixels
    //      of one plane: a tap's weights are read once for the four, a halo pixel once per row for up to three taps -
    //      38 LDS reads and ~430 VALU instructions per thread where one pixel per thread and chunk (first version) took
    //      116 and ~600, and the depthwise conv was LDS-bound.  Output pixel (r, x) goes to SLOT (x % 4) 16 + 2 r + x / 4
    //      of the plane: conflict-free 16-byte writes here, and GEMM 2's column 32 j + l31 simply IS slot 32 j + l31 ----
    {
      const int pl = tid >> 4, qr = tid & 15;
      if (pl < nplt) {
        const int ch = 8 * pl;
        const float4* s0 = xs + pl * XP1 + (qr >> 1) * kHalo + (qr & 1) * 4;  // halo pixel of tap (0, 0) of the first pixel
        const float4 bz0 = *reinterpret_cast<const float4*>(dwl + 9 * A.Kt + ch);
        const float4 bz1 = *reinterpret_cast<const float4*>(dwl + 9 * A.Kt + ch + 4);
        f2 v[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q][0] = f2{bz0.x, bz0.y};
          v[q][1] = f2{bz0.z, bz0.w};
          v[q][2] = f2{bz1.x, bz1.y};
          v[q][3] = f2{bz1.z, bz1.w};
        }
        float4 row[2][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) row[0][i] = s0[i];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          if (ky < 2) {
#pragma unroll
            for (int i = 0; i < 6; ++i) row[(ky + 1) & 1][i] = s0[(ky + 1) * kHalo + i];
          }
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float4 w0 = *reinterpret_cast<const float4*>(dwl + (ky * 3 + kx) * A.Kt + ch);
            const float4 w1 = *reinterpret_cast<const float4*>(dwl + (ky * 3 + kx) * A.Kt + ch + 4);
            const f2 ww[4] = {{w0.x, w0.y}, {w0.z, w0.w}, {w1.x, w1.y}, {w1.z, w1.w}};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float x[8];
              unpack8(row[ky & 1][q + kx], x);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[q][e] = __builtin_elementwise_fma(f2{x[2 * e], x[2 * e + 1]}, ww[e], v[q][e]);
            }
          }
          RTPOSE_UB_PIN();
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          a2[pl * AP2 + 16 * q + qr] =
              make_float4(__uint_as_float(pack2(v[q][0].x, v[q][0].y)), __uint_as_float(pack2(v[q][1].x, v[q][1].y)),
                          __uint_as_float(pack2(v[q][2].x, v[q][2].y)), __uint_as_float(pack2(v[q][3].x, v[q][3].y)));
      }
    }
    stage_store(xn, 0);   // the round requested before the T1 write-out ...
    stage_load(sqn, 2);   // ... and the second (last) one: GEMM 2 below waits for no load
    RTPOSE_UB_STAMP(6);
    __syncthreads();  // B3: the depthwise output is in a2
    RTPOSE_UB_STAMP(7);

    // ---- GEMM 2 (transposed): y^T[column][slot], every filter fragment already in registers ---------------------------
    floatx16 acc2[MF2];
    {
      floatx16 t;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 b = *reinterpret_cast<const float4*>(s_b2 + (32 * wc2 + 8 * rg + 4 * kh));
        t[rg * 4 + 0] = b.x;
        t[rg * 4 + 1] = b.y;
        t[rg * 4 + 2] = b.z;
        t[rg * 4 + 3] = b.w;
      }
#pragma unroll
      for (int j = 0; j < MF2; ++j) acc2[j] = t;
    }
    {
      float4 xq[2][MF2];
      auto x2load = [&](float4(&dst)[MF2], int g) {
        const float4* p = a2 + (unsigned)((2 * min(g, k2real - 1) + kh) * AP2 + 32 * MF2 * wm2 + l31);
#pragma unroll
        for (int j = 0; j < MF2; ++j) dst[j] = p[32 * j];
      };
      x2load(xq[0], 0);
#pragma unroll
      for (int g = 0; g < 4 * NCH2; ++g) {
        x2load(xq[(g & 1) ^ 1], g + 1);
        RTPOSE_UB_PIN();
        if (g < k2real) {  // (uniform)
#pragma unroll
          for (int j = 0; j < MF2; ++j)
            acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(w2r[g]), as_bf8(xq[g & 1][j]), acc2[j], 0, 0, 0);
        }
        RTPOSE_UB_PIN();
      }
    }
    RTPOSE_UB_STAMP(8);
    RTPOSE_UB_STAMP(9);
    stage_store(xn, 2);
    RTPOSE_UB_STAMP(10);

    // ---- epilogue: y = bf16(relu(.)) -> LDS [8-column group][slot] in the place of the dead T1 (everyone is past the
    //      depthwise conv), then 16 bytes per lane with the column groups along the lanes: groups that are neighbours in
    //      the output pixel go out as one request (8-byte stores straight from the accumulators, 32 different lines per
    //      instruction, took 10.5 k cycles of a 42 k tile)
    mfma_drain();
#pragma unroll
    for (int j = 0; j < MF2; ++j) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int pl = 4 * wc2 + rg;  // (uniform)
        const floatx16& a = acc2[j];
        uint2 v;
        v.x = pack2(relu_bits(acc_read(a[rg * 4 + 0])), relu_bits(acc_read(a[rg * 4 + 1])));
        v.y = pack2(relu_bits(acc_read(a[rg * 4 + 2])), relu_bits(acc_read(a[rg * 4 + 3])));
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(xs + pl * XP1 + 32 * (MF2 * wm2 + j) + l31) + 8 * kh) = v;
      }
      RTPOSE_UB_PIN();
    }
    __syncthreads();  // B5: the y tile is in LDS
    {
      constexpr int NG = C2P / 8;            // column groups of the packed matrix
      constexpr int PPI = 512 / NG;          // slots per pass of the block
      const int grp = tid & (NG - 1);
      const int cm = 8 * grp < A.cout ? A.out_cmap[8 * grp] : -1;  // absolute channel of the group's first column
#pragma unroll
      for (int k = 0; k < 64 / PPI; ++k) {
        const int sl = k * PPI + tid / NG;                  // slot -> pixel (see the depthwise conv)
        const int y = cur.y0 + ((sl & 15) >> 1), x = cur.x0 + (sl & 1) * 4 + (sl >> 4);
        const float4 v = xs[grp * XP1 + sl];
        if (cm >= 0 && y < A.H && x < A.W) {
          const unsigned oq = (unsigned)(A.out_lead + (cur.n * A.out_hs + y) * A.out_ws + x) * (unsigned)A.out_cstride;
          bstore4(v, rout, (oq + (unsigned)cm) * 2u);  // (descriptor + 32-bit offset: no 64-bit address held in VGPRs)
        }
      }
    }
    RTPOSE_UB_STAMP(11);
    ++tcount;
    if (!has_next) break;
    item = nitem;
    cur = nxt;
    pp ^= 1;
  }
}
#undef RTPOSE_UB_PIN

#ifdef RTPOSE_EXP_TIMELINE_UNIT
static unsigned long long* g_unit_timeline = nullptr;  // device buffer [blocks][20][16], see tools/timeline_unit.py
#endif

template <int WM1, int WM2, int NCH2>
static int launch_inst(const Args& a, int grid, size_t lds, hipStream_t s) {
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  auto kern = unit_bf16_kernel<WM1, WM2, NCH2>;
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024 - 2560));  // (2.2 KB are static)
    attr_set.set(dev);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace unitb

// d0: conv.0 (+ReLU): the stage buffer as a gather of d0->cin / 8 planes (in_planes) or a contiguous slice, layout gap
//     >= 1; w_packed [cin / 8][coutp][8 bf16] with coutp = 128 or 256 and ZERO columns past the Kt real ones.
// dw_w / dw_b in d0: conv.1's fp32 taps [9][Kt] and bias [Kt] (Kt = d2->cin).
// d2: conv.2 (+ReLU): w_packed [Kt / 8][coutp][8 bf16], coutp = 128 or 256, `cout` columns exist (a multiple of 8),
//     out_cmap[column] = absolute channel, groups of 8 columns contiguous; d2->out / lout = the stage buffer.
int unit_bf16_fits(const rtpose_pw_desc* d0, const rtpose_pw_desc* d2, int H, int W) {
  if (!d0 || !d2 || H < 1 || W < 1) return 0;
  if (d0->cin < 16 || (d0->cin % 16) || d0->cin > unitb::kMaxK1) return 0;
  if (d0->coutp != (d2->cin <= 128 ? 128 : 256)) return 0;
  if (d2->cin < 16 || (d2->cin % 16) || d2->cin > d0->coutp || d2->cin > unitb::kMaxKt) return 0;
  if (d2->coutp != 128 && d2->coutp != 256) return 0;
  if (d2->cout < 8 || (d2->cout % 8) || d2->cout > d2->coutp || !d2->out_cmap) return 0;
  if (!d0->relu || !d2->relu || !d0->dw_w || !d0->dw_b || d0->pt_src || d2->pt_src || d2->dw_w) return 0;
  if ((d0->lin.cstride % 8) || (d0->lin.choff % 8) || (d2->lout.cstride % 8)) return 0;
  if (d0->lin.ws < W + 1 || d0->lin.hs < H + 1 || d0->lin.lead < d0->lin.ws + 1) return 0;
  // the next tile's x2 is staged in two rounds of two 64-channel chunks (under the write-out + depthwise conv, under GEMM 2)
  if (d0->cin > 256) return 0;
  const size_t lds = ((size_t)2 * ((d0->cin > d2->cin ? d0->cin : d2->cin) >> 3) * unitb::XP1 + (size_t)(d2->cin >> 3) * unitb::AP2) * 16 +
                     (size_t)10 * d2->cin * 4;
  return lds <= 160 * 1024 - 2560 ? 1 : 0;
}

int unit_bf16_launch(const rtpose_pw_desc* d0, const rtpose_pw_desc* d2, int N, int H, int W, hipStream_t s) {
  using namespace unitb;
  if (!unit_bf16_fits(d0, d2, H, W)) return fail(RTPOSE_E_INVAL, "unit_bf16: shape without an instance (see unit_bf16_fits)");
  if (!d0->in || !d0->w_packed || !d0->bias_packed || !d2->w_packed || !d2->bias_packed || !d2->out)
    return fail(RTPOSE_E_INVAL, "unit_bf16: NULL argument");
  if (N <= 0) return fail(RTPOSE_E_INVAL, "unit_bf16: empty tensor");
  const size_t in_bytes = rtpose_layout_pixels(&d0->lin, N, H, W) * (size_t)d0->lin.cstride * 2;
  const size_t out_elems = rtpose_layout_pixels(&d2->lout, N, H, W) * (size_t)d2->lout.cstride;
  if (in_bytes >= ((size_t)1 << 31) || out_elems >= ((size_t)1 << 31))
    return fail(RTPOSE_E_INVAL, "unit_bf16: tensors must be below 2^31 bytes / elements (32-bit offsets)");
  Args a;
  memset(&a, 0, sizeof(a));
  a.in = reinterpret_cast<const unsigned short*>(d0->in);
  a.in_bytes = in_bytes;
  a.in_cstride = d0->lin.cstride;
  a.in_choff = d0->lin.choff;
  a.in_ws = d0->lin.ws;
  a.in_hs = d0->lin.hs;
  a.in_lead = d0->lin.lead;
  a.in_planes = d0->in_planes;
  a.w0 = d0->w_packed;
  a.b0 = d0->bias_packed;
  a.dw_w = d0->dw_w;
  a.dw_b = d0->dw_b;
  a.w2 = d2->w_packed;
  a.b2 = d2->bias_packed;
  a.out = reinterpret_cast<unsigned short*>(d2->out);
  a.out_bytes = out_elems * 2;
  a.out_cstride = d2->lout.cstride;
  a.out_ws = d2->lout.ws;
  a.out_hs = d2->lout.hs;
  a.out_lead = d2->lout.lead;
  a.out_cmap = d2->out_cmap;
  a.N = N;
  a.H = H;
  a.W = W;
  a.K1 = d0->cin;
  a.Kt = d2->cin;
  a.cout = d2->cout;
  a.tiles_x = ceil_div(W, kTile);
  a.tiles_y = ceil_div(H, kTile);
  a.nitems = N * a.tiles_x * a.tiles_y;
  a.ftx = make_fastdiv(a.tiles_x);
  a.fty = make_fastdiv(a.tiles_y);
#ifdef RTPOSE_EXP_TIMELINE_UNIT
  a.dbg = g_unit_timeline;
#endif
  const size_t lds = ((size_t)2 * ((a.K1 > a.Kt ? a.K1 : a.Kt) >> 3) * XP1 + (size_t)(a.Kt >> 3) * AP2) * 16 + (size_t)10 * a.Kt * 4;
  const int slots = device_cu_count();  // one block per CU
  const int grid = a.nitems < slots ? a.nitems : slots;
  // instances: T1 of 64 / 128 / 192-256 channels (1, 2, 4 chunks; conv.0 packed 128 / 128 / 256 columns wide) x conv.2
  // packed 128 or 256 columns wide
  const bool wide2 = d2->coutp == 256;
  const int nch2 = (a.Kt + 63) / 64;
  if (nch2 == 1) return wide2 ? launch_inst<2, 1, 1>(a, grid, lds, s) : launch_inst<2, 2, 1>(a, grid, lds, s);
  if (nch2 == 2) return wide2 ? launch_inst<2, 1, 2>(a, grid, lds, s) : launch_inst<2, 2, 2>(a, grid, lds, s);
  return wide2 ? launch_inst<1, 1, 4>(a, grid, lds, s) : launch_inst<1, 2, 4>(a, grid, lds, s);
}

}  // namespace rtpose

extern "C" {

#ifdef RTPOSE_EXP_TIMELINE_UNIT
void rtpose_debug_unit_timeline(void* device_buffer) {
  rtpose::unitb::g_unit_timeline = static_cast<unsigned long long*>(device_buffer);
}
#endif

int rtpose_unit_bf16_fits(const rtpose_pw_desc* d0, const rtpose_pw_desc* d2, int H, int W) {
  return rtpose::unit_bf16_fits(d0, d2, H, W);
}

int rtpose_unit_bf16(const rtpose_pw_desc* d0, const rtpose_pw_desc* d2, int N, int H, int W, void* stream) {
  return rtpose::unit_bf16_launch(d0, d2, N, H, W, rtpose::as_stream(stream));
}

}  // extern "C"
