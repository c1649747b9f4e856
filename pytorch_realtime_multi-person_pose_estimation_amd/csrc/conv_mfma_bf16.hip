// bf16 implicit-GEMM convolution for gfx950 (MI355X): bf16 activations and weights,
// fp32 accumulate (v_mfma_f32_32x32x16_bf16), fused bias (+ReLU) (+2x2 max-pool), output
// rounded to bf16 (round-to-nearest-even) or kept fp32 (network heads).  Two compute dtypes:
//   bf16   (SP = 1): BASELINE config 3 arithmetic - operands rounded to bf16, exact products;
//   bf16x3 (SP = 2): every fp32 operand carried as hi + lo bf16 pieces, three MFMAs per product -
//                    fp32-grade results (see conv_tile) at 3/16 of the fp32 MFMA time.
//
// Stands in for the same nn.Conv2d / nn.ReLU / nn.MaxPool2d modules as the fp32 kernel
// (conv_mfma.hip; lib/network/rtpose_vgg.py:23-35, :49-55).  The reference has no reduced-
// precision path; the contracts are oracle/net_oracle.py:forward_bf16_emulated /
// forward_bf16x3_emulated.
//
// Same skeleton as the fp32 kernel - shared-gap padded NHWC (2 bytes per element), one LDS halo
// per channel chunk re-used by all k*k taps, weights straight from L2, one barrier per chunk -
// re-balanced for a matrix pipe that is 16x faster per byte:
//  * a 16-byte piece is 8 channels = exactly one lane's share of a K=16 MFMA step, so the A
//    fragment is ONE ds_read_b128 and the B fragment ONE 16-byte load feeding a single 32-cycle
//    MFMA (fp32: four 64-cycle ones);
//  * block tile 128 x 128 in two wave arrangements (template WM): 2 x 2 waves of 64 x 64, or
//    1 x 4 waves of 128 x 32 which halves the weight bytes pulled through the L1 and reads all A
//    fragments from LDS instead (4x the L1's bandwidth); chosen per kernel size by measurement;
//  * B through buffer loads: SGPR resource + SGPR tap offset + one per-lane VGPR, so the tap loop
//    spends no VALU and no address registers on weights; the freed registers pay for a prefetch
//    ring of up to 4 taps;
//  * the next chunk's halo pieces travel through a register ring (fetched at tap t, parked in LDS
//    kHD taps later); the first halo's loads are issued before the block's setup math;
//  * epilogue: the bias rides in the accumulator from the start; each wave transposes its tile
//    through LDS and stores 16 bytes per lane.
// Per-CU timelines (tools/timeline_bf16.py): two co-resident blocks keep the matrix pipe ~96 %
// busy while both are in their tap loops; what is lost is the per-block prologue (~11k cycles,
// memory latency) and epilogue (~7k) on a 50k-cycle tile, and the tail of the last round.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "conv_exp.h"

namespace rtpose {

int conv_c64_bf16_fits(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, int out_f32, int split);
int conv_c64_bf16_launch(const rtpose_conv_desc* d, int N, int H, int W, hipStream_t s);

namespace bf {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const floatx4 __attribute__((address_space(1)))* gcf4_t;
// explicit global address space (see conv_mfma.hip: FLAT loads drag LDS reads behind vmcnt)
__device__ __forceinline__ float4 gload4(const void* p) {
  const floatx4 v = *(gcf4_t)(unsigned long long)(p);
  return make_float4(v[0], v[1], v[2], v[3]);
}
typedef floatx4 __attribute__((address_space(1)))* gf4_t;
__device__ __forceinline__ void gstore4(void* p, const float4& v) {
  const floatx4 t = {v.x, v.y, v.z, v.w};
  *(gf4_t)(unsigned long long)(p) = t;
}
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  const floatx4 v = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ bf16x8 as_bf8(const float4& v) {
  const floatx4 t = {v.x, v.y, v.z, v.w};
  return __builtin_bit_cast(bf16x8, t);
}
__device__ __forceinline__ unsigned short to_bf16(float v) {
  return __builtin_bit_cast(unsigned short, (__bf16)v);  // v_cvt_pk_bf16_f32: RNE
}
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {  // (bf16(a), bf16(b)) in one dword, a in the low half
  const floatx2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float4 as_f4(unsigned a, unsigned b, unsigned c, unsigned d) {
  return make_float4(__uint_as_float(a), __uint_as_float(b), __uint_as_float(c), __uint_as_float(d));
}

struct ConvGroup {
  const unsigned short* in;  // bf16 activations
  const float4* w;           // packed bf16 weights, 16-byte pieces
  const float* bias;         // fp32, padded to cout_pad
  void* out;                 // bf16 or fp32 activations
  int in_cstride, in_choff, in_ws, in_hs, in_lead;  // channel counts in ELEMENTS
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  int cout, cout_pad;
  const int32_t* out_cmap;
};

struct ConvArgs {
  ConvGroup g[2];
  int N, H, W, M;
  int cin;  // packed input channels (multiple of the chunk size)
  int relu, pool, out_f32;
  int vec_store;  // bf16 output, full N tiles, 16-byte aligned slices: transposed epilogue
  int tr;         // vec_store without a fused pool, k x k: the transposed-product instances (template TR), no LDS in the epilogue
  int qs;       // LDS pixels per piece plane
  int hw_lds;   // MODE 1: LDS row stride of the halo (pixels)
  int tw_log2;  // MODE 1: log2(tile width); tile height = 128 >> tw_log2
  int tiles_x, tiles_y;
  int mtiles, ntiles, nbig, ncombo, xcd_remap;
  int npersist;  // strip mode: blocks [0, npersist) walk the full-tile ids L, L + npersist, ... < nbig; blocks
                 // past them are the half tiles of the tail.  == nbig: one tile per block (not persistent)
  int dephase_cycles, n_cu;  // second-slot blocks of the first dispatch wave start this much later
  unsigned long long* dbg;  // RTPOSE_EXP_TIMELINE builds only: 8 x u64 per block
};

constexpr int kBM = 128;

// block id -> (m tile, combo = N tile + ntiles * group).  xcd_remap: the ncombo blocks of an m tile sit on one XCD (ids go
// round-robin to the 8 XCDs), so the tile's halo is fetched into one L2.  (Round 4, measured and dropped: giving every XCD
// the combos of ONE branch only, so that its L2 holds 1.6 instead of 3.2 MB of packed 7x7 filters - TCC hit rate 87.1 ->
// 88.2 %, time -0.2 %; non-temporal output stores: 0 %.  profiles/r04_bf16_conv_experiments.txt)
__device__ __forceinline__ void decode_block_id(const ConvArgs& A, int bi, int& mt, int& c) {
  if (A.xcd_remap) {
    const int xcd = bi & 7, j = bi >> 3;
    c = j % A.ncombo;
    mt = (j / A.ncombo) * 8 + xcd;
  } else {
    mt = bi % A.mtiles;
    c = bi / A.mtiles;
  }
}
// depth of the halo staging ring (taps between fetch and park).  The counter behind s_waitcnt
// is in order, so every wait for a weight piece also waits for all older staging loads: a
// staging load must be able to take a full HBM/MALL miss (1-3k cycles under load) without
// being the oldest thing a wave waits for - measured per-CU timelines showed a block that has
// the CU to itself (its partner in prologue/epilogue, or the tail) at 55 % MFMA rate with 3.
constexpr int kHD = RTPOSE_EXP_HD;
// Epilogue slab: each wave transposes its (up to) 64 x 64 bf16 tile through LDS so that a lane
// stores 16 bytes (8 output channels of one pixel) instead of 64 scattered 2-byte values -
// measured, the scalar epilogue cost 25-30 % of a 7x7 layer at bf16 MFMA speed.
constexpr int slab_bytes(int mf, int nf, int sp) { return 32 * mf * (nf * 64 * sp + 16); }  // per wave

__device__ __forceinline__ void tile_local_yx(int ml, int tw_log2, int& ty, int& tx) {
  const int qi = ml >> 2;
  const int hw_log2 = tw_log2 - 1;
  ty = ((qi >> hw_log2) << 1) + ((ml >> 1) & 1);
  tx = ((qi & ((1 << hw_log2) - 1)) << 1) + (ml & 1);
}

// KS: kernel size; CK: channels per LDS chunk (16, 32 or 64); MODE 0 strip / 1 2-D tile;
// NBUF 2: next chunk staged under the MFMAs, NBUF 1: refilled between chunks (1x1 layers);
// WM: waves along M (the block's 4 waves form WM x 4/WM); MF / NF: 32-row / 32-column fragments
// per wave (block tile 32*MF*WM pixels x 32*NF*4/WM channels).  The 128 x 128 block tile exists
// in two arrangements: 2 x 2 waves of 64 x 64 (MF = NF = 2) and 1 x 4 waves of 128 x 32
// (MF = 4, NF = 1).  The second halves the B (weight) bytes a CU pulls through its L1 per MFMA
// - every wave reads all A fragments from LDS instead, which has 4x the L1's bandwidth.
// SP = 2: "split" operands (compute dtype bf16x3).  Every fp32 value v travels as two bf16s,
// hi = bf16(v) and lo = bf16(v - hi) (16 significant bits), stored as interleaved 16-byte pieces
// [hi x 8 channels | lo x 8 channels]; a K-step issues hi*lo + lo*hi + hi*hi (the lo*lo term,
// 2^-18 relative, is dropped) - fp32-grade results from the bf16 pipe at 3/16 of the fp32
// MFMA time.
// PERSISTENT form (bi_stride > 0; strip mode, NBUF 2 only): the block walks the full-tile ids bi,
// bi + bi_stride, ... < A.nbig.  During the LAST chunk of a tile the staging ring fetches chunk 0 of the
// NEXT tile's halo into the idle LDS buffer instead of re-staging itself, so the next tile starts
// its tap loop at once: the ~11k-cycle prologue (one exposed memory round trip per block) is paid
// once per block instead of once per tile, and the epilogue's stores drain under the next tile's MFMAs.
// TR (round 4): the TRANSPOSED product - D^T = W^T X^T, i.e. the weight fragment goes in as the MFMA's row operand and the
// pixel fragment as its column operand (the registers are the same, only their order in the instruction changes).  A lane
// then holds ONE pixel (column l31) and 16 output channels (rows 8 (r / 4) + 4 kh + r % 4 of the fragment); with the
// lane -> weight-column map permuted so that those rows are the channels 16 kh .. 16 kh + 15 IN ORDER, the epilogue is
// max / convert / two 16-byte stores per fragment straight from the accumulators - no LDS transpose (64 two-byte LDS writes
// + 8 reads per lane and tile in the slab form: by ablation a quarter of a 7x7 tile's time), no slab, no barrier before the
// block's next tile.  Used for bf16 / split outputs of full N tiles without a fused pool (launch code: `tr`).
template <int KS, int CK, int MODE, int NBUF, int WM, int MF, int NF, int SP, bool TR = false>
__device__ __forceinline__ void conv_tile(const ConvArgs& A, const int grp_first, const int m0_first,
                                          const int ntile_first, float* smem, int bi = 0, const int bi_stride = 0) {
  constexpr int P = KS / 2;
  constexpr int BMT = 32 * MF * WM;
  constexpr int BN = 32 * NF * (4 / WM);  // output channels per block
  constexpr int CG = CK / 8 * SP;  // 16-byte pieces (8 channels; hi and lo when split) per pixel per chunk
  constexpr int G = CK / 16;   // K=16 MFMA steps per tap
  constexpr int GB = G * NF;   // B fragments per tap: [n-fragment][k-step] (x SP registers)
  // B register ring: the tap being multiplied + RB-1 taps in flight from L2.  Two taps of
  // lead (RB = 3) left the waves waiting on vmcnt once two blocks share a CU; the narrow-N
  // arrangement (NF = 1) has the registers for four.
  constexpr int RB = (NF == 1 && CK <= 32) ? 5 : (CK <= 32 ? RTPOSE_EXP_RB2 : 3);
  constexpr int TAPS = KS, ROWS = KS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int l31 = lane & 31, kh = lane >> 5;
  const int QS = A.qs;
  const bool persist = MODE == 0 && NBUF == 2 && bi_stride > 0;
  // (the group is carried as an index and its descriptor copied by value per tile: a pointer into the
  //  kernel argument that changes inside the loop made the compiler spill the whole ConvArgs to scratch)
  int grp = grp_first;
  int m0_arg = m0_first, ntile = ntile_first;
  int lpar = 0;  // LDS buffer holding chunk 0 of the current tile
  // the next full tile of this block (persistent form), found while the current one is set up
  bool has_next = false;
  int nx_bi = bi, nx_m0 = 0, nx_nt = 0, nx_q_origin = 0, nx_np_total = 0, nx_grp = grp_first;
  // strip-mode staging geometry of a tile: first halo pixel, 16-byte pieces of one chunk's halo
  auto strip_geom = [&](const ConvGroup& gg, int m0s, int& q_org, int& np_tot) {
    const int HW = A.H * A.W;
    const int n = m0s / HW, r = m0s - n * HW;
    const int y = r / A.W, x = r - y * A.W;
    const int qf = gg.in_lead + (n * gg.in_hs + y) * gg.in_ws + x;
    const int ml = min(m0s + 32 * MF * WM, A.M) - 1;
    const int n2 = ml / HW, r2 = ml - n2 * HW;
    const int y2 = r2 / A.W, x2 = r2 - y2 * A.W;
    const int ql = gg.in_lead + (n2 * gg.in_hs + y2) * gg.in_ws + x2;
    q_org = qf - (KS / 2) * gg.in_ws - (KS / 2);
    np_tot = (ql + (KS / 2) * gg.in_ws + (KS / 2) - q_org + 1) * (CK / 8 * SP);
  };
  // One tile.  Instantiated twice: FIRST (the block's first tile fills its own halo - the prologue) and the
  // persistent continuation, whose chunk 0 was staged by the previous tile.  Keeping the prologue out of the
  // looped copy matters: as loop-invariant code its addresses stayed live through the tap loop (+80 VGPRs).
  auto tile = [&](auto first_tag) __attribute__((always_inline)) {
  constexpr bool FIRST = decltype(first_tag)::value;
  const ConvGroup g = grp ? A.g[1] : A.g[0];

  // ---- block -> tile ---------------------------------------------------------------
  int m0 = 0, n_img = 0, y0 = 0, x0 = 0;
  int q_origin, np_pix, row_lds;
  int qc0 = 0;
  if (MODE == 0) {
    m0 = m0_arg;
    const int HW = A.H * A.W;
    const int n = m0 / HW, r = m0 - n * HW;
    const int y = r / A.W, x = r - y * A.W;
    qc0 = g.in_lead + (n * g.in_hs + y) * g.in_ws + x;
    const int ml = min(m0 + BMT, A.M) - 1;
    const int n2 = ml / HW, r2 = ml - n2 * HW;
    const int y2 = r2 / A.W, x2 = r2 - y2 * A.W;
    const int qcl = g.in_lead + (n2 * g.in_hs + y2) * g.in_ws + x2;
    q_origin = qc0 - P * g.in_ws - P;
    np_pix = qcl + P * g.in_ws + P - q_origin + 1;
    row_lds = g.in_ws;
  } else {
    int b = m0_arg;
    const int txi = b % A.tiles_x;
    b /= A.tiles_x;
    const int tyi = b % A.tiles_y;
    n_img = b / A.tiles_y;
    const int TW = 1 << A.tw_log2, TH = kBM >> A.tw_log2;
    y0 = tyi * TH;
    x0 = txi * TW;
    q_origin = g.in_lead + (n_img * g.in_hs + y0 - P) * g.in_ws + x0 - P;
    np_pix = (TH + 2 * P) * A.hw_lds;
    row_lds = A.hw_lds;
  }
  const int np_total = np_pix * CG;  // 16-byte pieces per LDS buffer

  // ---- halo staging helpers: piece idx = (pixel idx / CG, piece idx % CG) ------------
  const int in_cs4 = g.in_cstride >> 3, in_ws = g.in_ws;  // 16-byte pieces per pixel
  const float4* in_base = reinterpret_cast<const float4*>(g.in + g.in_choff);
  constexpr int PIXSET = 256 / CG;
  const int pj = tid % CG, ppix0 = tid / CG;
  const unsigned hw_inv = MODE == 1 ? (65536u + A.hw_lds - 1) / A.hw_lds : 0u;
  // float4 offset of piece (set, tid) from the halo's first pixel.  It does not depend on WHICH halo is
  // staged (this tile's next chunk or the next tile's first): the halo origin goes into the uniform base
  // pointer, so the per-thread offsets are computed once (with the origin inside, the compiler kept both
  // variants of every 64-bit address live through the tap loop).
  auto piece_rel = [&](int set) -> unsigned {
    int pix = set * PIXSET + ppix0;
    int qr;
    if (MODE == 0) {
      qr = pix;
    } else {
      pix = min(pix, np_pix - 1);
      const int hy = (int)(((unsigned)pix * hw_inv) >> 16), hx = pix - hy * A.hw_lds;
      qr = hy * in_ws + hx;
    }
    return (unsigned)qr * (unsigned)in_cs4 + (unsigned)pj;
  };
  const float4* halo_base = in_base + (size_t)q_origin * in_cs4;  // chunk 0 of this tile's halo
  float4* smem4 = reinterpret_cast<float4*>(smem);
  const int buf4 = CG * QS;
  auto piece_loff = [&](int set) -> int { return pj * QS + set * PIXSET + ppix0; };
  const int dummy_loff = NBUF * buf4 + tid;

  // ---- prologue: the first halo's loads go out before anything that does not feed their
  //      addresses (measured: the fill is ~11k cycles of pure memory latency per block)
  constexpr int kFillDepth = 10;
  const int nsets = (np_total + 255) / 256;
  // persistent: the next full tile of this block (ids past the remapped order's padding are skipped)
  has_next = false;
  nx_q_origin = q_origin;
  nx_np_total = np_total;
  nx_grp = grp;
  if (persist) {
    for (nx_bi = bi + bi_stride; nx_bi < A.nbig; nx_bi += bi_stride) {
      int mt, c;
      decode_block_id(A, nx_bi, mt, c);
      if (mt < A.mtiles) {
        has_next = true;
        nx_m0 = mt * kBM;
        nx_nt = c % A.ntiles;
        nx_grp = c / A.ntiles;
        strip_geom(nx_grp ? A.g[1] : A.g[0], nx_m0, nx_q_origin, nx_np_total);
        break;
      }
    }
  }
  const float4* nx_in_base = reinterpret_cast<const float4*>(
      nx_grp ? A.g[1].in + A.g[1].in_choff : A.g[0].in + A.g[0].in_choff);
  const int nx_nsets = (nx_np_total + 255) / 256;
  float4 pf[kFillDepth];
  const bool pro_fast = FIRST && NBUF == 2 && nsets <= kFillDepth;
#ifndef RTPOSE_EXP_NO_FILL
  if (FIRST && pro_fast) {
#pragma unroll
    for (int u = 0; u < kFillDepth; ++u)
      if (u < nsets && u * 256 + tid < np_total) pf[u] = gload4(halo_base + piece_rel(u));
  }
#endif

  // ---- per-lane A fragment bases (LDS pixel index of this lane's row) ----------------
  int abase[MF];
  int an = 0, ay = 0, ax = 0;  // strip mode: coordinates of fragment 0's row, stepped by 32 pixels
  if (MODE == 0) {
    const int m = m0 + wm * (32 * MF) + l31;
    const int HW = A.H * A.W;
    an = m / HW;
    const int r = m - an * HW;
    ay = r / A.W;
    ax = r - ay * A.W;
  }
#pragma unroll
  for (int fm = 0; fm < MF; ++fm) {
    const int ml = wm * (32 * MF) + fm * 32 + l31;
    if (MODE == 0) {
      // rows past the end of the tensor (last strip) read the last real pixel's halo
      const bool past = m0 + ml > A.M - 1;
      const int n = past ? A.N - 1 : an, y = past ? A.H - 1 : ay, x = past ? A.W - 1 : ax;
      abase[fm] = g.in_lead + (n * g.in_hs + y) * g.in_ws + x - qc0;
      ax += 32;
      while (ax >= A.W) {
        ax -= A.W;
        if (++ay >= A.H) {
          ay = 0;
          ++an;
        }
      }
    } else {
      int ty, tx;
      tile_local_yx(ml, A.tw_log2, ty, tx);
      abase[fm] = ty * A.hw_lds + tx;
    }
  }

  // ---- B operand: buffer loads = SGPR resource + SGPR tap/k-step offset + per-lane VGPR
  //      offset, so the tap loop spends no VALU and no address registers on them; reads past
  //      the packed filter (the prefetch runs RB-1 taps ahead) return zero by the bounds check
  const int nchunks = A.cin / CK;
  const int ncol = ntile * BN + wn * (32 * NF) + l31;
  // TR: fragment row l31 multiplies the weights of channel prow(l31) = 16 ((l31 >> 2) & 1) + 4 (l31 >> 3) + (l31 & 3), so that
  // register r of lane half kh is channel 16 kh + r
  const int wcol = TR ? ncol - l31 + 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3) : ncol;
  // piece plane of (k-step gi, lane half kh, hi/lo s) = (2 gi + kh) * SP + s
  const unsigned lane_b = (unsigned)(kh * SP * g.cout_pad + wcol) * 16u;
  const unsigned b_it_bytes = (unsigned)(CG * g.cout_pad) * 16u;       // bytes per (chunk, tap)
  const unsigned b_k_bytes = (unsigned)(2 * SP * g.cout_pad) * 16u;    // bytes per k-step
  const unsigned b_s_bytes = (unsigned)g.cout_pad * 16u;               // hi -> lo plane
  const unsigned w_bytes = (unsigned)nchunks * (KS * KS) * b_it_bytes;
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(g.w), 0, (int)w_bytes, 0x00020000);
  unsigned wso = 0;  // uniform: byte offset of the tap being fetched
#define RTPOSE_BLOAD(fn_, gi_, s_) \
  bload4(wrs, lane_b + (fn_) * 512u, wso + (unsigned)(gi_) * b_k_bytes + (unsigned)(s_) * b_s_bytes)

  float4 bq[RB][GB][SP];
#pragma unroll
  for (int t = 0; t + 1 < RB; ++t) {
#pragma unroll
    for (int fn = 0; fn < NF; ++fn)
#pragma unroll
      for (int gi = 0; gi < G; ++gi)
#pragma unroll
        for (int sp = 0; sp < SP; ++sp) bq[t][fn * G + gi][sp] = RTPOSE_BLOAD(fn, gi, sp);
    wso += b_it_bytes;
  }
#pragma unroll
  for (int gi = 0; gi < GB; ++gi)
#pragma unroll
    for (int sp = 0; sp < SP; ++sp) bq[RB - 1][gi][sp] = make_float4(0.f, 0.f, 0.f, 0.f);

  // up to kFillDepth pieces per thread in flight: ONE memory round trip for a 7x7 halo
  auto fill_halo = [&](const float4* src) {
    for (int set0 = 0; set0 < nsets; set0 += kFillDepth) {
      float4 t[kFillDepth];
#pragma unroll
      for (int u = 0; u < kFillDepth; ++u)
        if (set0 + u < nsets && (set0 + u) * 256 + tid < np_total) t[u] = gload4(src + piece_rel(set0 + u));
#pragma unroll
      for (int u = 0; u < kFillDepth; ++u)
        if (set0 + u < nsets && (set0 + u) * 256 + tid < np_total) smem4[piece_loff(set0 + u)] = t[u];
    }
  };
  RTPOSE_TSTAMP(6);
#ifndef RTPOSE_EXP_NO_FILL
  if (FIRST && NBUF == 2) {  // (persistent: only the block's first tile fills its own halo; lpar == 0)
    if (pro_fast) {
#pragma unroll
      for (int u = 0; u < kFillDepth; ++u)
        if (u < nsets && u * 256 + tid < np_total) smem4[piece_loff(u)] = pf[u];
    } else {
      fill_halo(halo_base);
    }
    RTPOSE_TSTAMP(7);
    __syncthreads();
  }
#endif
  RTPOSE_TSTAMP(1);

  floatx16 acc[MF][NF];
  if (TR) {
    // the 32 biases of a fragment through the scalar cache (the address is wave-uniform), selected by lane half
#pragma unroll
    for (int fn = 0; fn < NF; ++fn) {
      const float* bp = g.bias + (ntile * BN + wn * (32 * NF) + fn * 32);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float b = kh ? bp[16 + r] : bp[r];
#pragma unroll
        for (int fm = 0; fm < MF; ++fm) acc[fm][fn][r] = b;
      }
    }
  } else {
    float bias_r[NF];  // fetched now: at the epilogue this load's latency would be fully exposed
#pragma unroll
    for (int fn = 0; fn < NF; ++fn) bias_r[fn] = g.bias[ncol + fn * 32];
#pragma unroll
    for (int fm = 0; fm < MF; ++fm)
#pragma unroll
      for (int fn = 0; fn < NF; ++fn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[fm][fn][r] = bias_r[fn];  // bias rides in the accumulator
  }

  int afrag[G][MF];
#pragma unroll
  for (int gi = 0; gi < G; ++gi)
#pragma unroll
    for (int fm = 0; fm < MF; ++fm) afrag[gi][fm] = (2 * gi + kh) * SP * QS + abase[fm];  // hi plane; lo = + QS
  const int rowstep = row_lds;

  // pixel fragment x weight fragment (TR: the other way round - same registers, transposed accumulator)
  auto mm = [](const float4& px, const float4& wt, const floatx16& c) __attribute__((always_inline)) -> floatx16 {
    return TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(wt), as_bf8(px), c, 0, 0, 0)
              : __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(px), as_bf8(wt), c, 0, 0, 0);
  };
// (RTPOSE_EXP_B / _A / _STAGE: identity in production builds, see conv_exp.h)
#define RTPOSE_PIN()             \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)
  // One tap = G K-steps of MF*NF MFMAs; after K-step n: the B pieces of step n for the tap
  // two ahead, then the A pieces of step n for the next tap, then (last step, staged rows
  // only) one halo piece parked and one fetched.
#define RTPOSE_CONV_STEP(ACUR, ANXT, BCUR, BLOAD, KX, STAGE)                                   \
  {                                                                                            \
    _Pragma("unroll") for (int n = 0; n < G; ++n) {                                            \
      _Pragma("unroll") for (int fn = 0; fn < NF; ++fn) {                                      \
        _Pragma("unroll") for (int fm = 0; fm < MF; ++fm) {                                    \
          if (SP == 2) { /* small terms first: hi*lo, lo*hi, then hi*hi */                     \
            acc[fm][fn] = mm(ACUR[n][fm][0], BCUR[fn * G + n][SP - 1], acc[fm][fn]);           \
            acc[fm][fn] = mm(ACUR[n][fm][SP - 1], BCUR[fn * G + n][0], acc[fm][fn]);           \
          }                                                                                    \
          acc[fm][fn] = mm(ACUR[n][fm][0], BCUR[fn * G + n][0], acc[fm][fn]);                  \
        }                                                                                      \
      }                                                                                        \
      RTPOSE_PIN();                                                                            \
      _Pragma("unroll") for (int fn = 0; fn < NF; ++fn)                                        \
        _Pragma("unroll") for (int sp = 0; sp < SP; ++sp)                                      \
          BLOAD[fn * G + n][sp] = RTPOSE_EXP_B(RTPOSE_BLOAD(fn, n, sp), BCUR[fn * G + n][sp]); \
      if (n == G - 1) wso += b_it_bytes;                                                       \
      if (KS > 1) {                                                                            \
        if (n == 0 && (KX) == KS - 1) { /* next tap starts the next stencil row */             \
          _Pragma("unroll") for (int g2 = 0; g2 < G; ++g2)                                     \
            _Pragma("unroll") for (int fm = 0; fm < MF; ++fm) arow[g2][fm] += rowstep;         \
        }                                                                                      \
        _Pragma("unroll") for (int fm = 0; fm < MF; ++fm)                                      \
          _Pragma("unroll") for (int sp = 0; sp < SP; ++sp)                                    \
            ANXT[n][fm][sp] = RTPOSE_EXP_A(smem4[arow[n][fm] + sp * QS + (((KX) + 1 < KS) ? (KX) + 1 : 0)], \
                                           ACUR[n][fm][sp]);                                   \
      }                                                                                        \
      if (((STAGE) & RTPOSE_EXP_STAGE) != 0 && n == G - 1) {                                   \
        smem4[hl[0]] = hv[0];                                                                  \
        _Pragma("unroll") for (int d = 0; d + 1 < kHD; ++d) {                                  \
          hv[d] = hv[d + 1];                                                                   \
          hl[d] = hl[d + 1];                                                                   \
        }                                                                                      \
        if (ps < st_nsets) { /* uniform */                                                     \
          hv[kHD - 1] = gload4(RTPOSE_EXP_STAGE_SRC(next_base, halo_base) + piece_rel(ps));    \
          hl[kHD - 1] = (tid < st_np_total - ps * 256) ? hn_off + piece_loff(ps) : dummy_loff; \
        } else {                                                                               \
          hv[kHD - 1] = make_float4(0.f, 0.f, 0.f, 0.f);                                       \
          hl[kHD - 1] = dummy_loff;                                                            \
        }                                                                                      \
        ++ps;                                                                                  \
      }                                                                                        \
      RTPOSE_PIN();                                                                            \
    }                                                                                          \
  }
#define RTPOSE_CONV_ROW(STAGE)                                                              \
  {                                                                                         \
    _Pragma("unroll") for (int kx = 0; kx < TAPS; ++kx) {                                   \
      /* A sets alternate; B: multiply ring slot kx % RB, refill the slot freed by the */   \
      /* previous tap with the tap RB-1 ahead (all indices fold after unrolling)       */   \
      RTPOSE_CONV_STEP(av[kx & 1], av[(kx & 1) ^ 1], bq[kx % RB], bq[(kx + RB - 1) % RB], kx, STAGE) \
    }                                                                                       \
    /* re-normalise the register roles for the next row (a few v_mov per row) */            \
    if (KS > 1 && (TAPS & 1)) {                                                             \
      _Pragma("unroll") for (int gi = 0; gi < G; ++gi)                                      \
        _Pragma("unroll") for (int fm = 0; fm < MF; ++fm)                                   \
          _Pragma("unroll") for (int sp = 0; sp < SP; ++sp) av[0][gi][fm][sp] = av[1][gi][fm][sp]; \
    }                                                                                       \
    if (TAPS % RB != 0) {                                                                   \
      float4 t_[RB][GB][SP];                                                                \
      _Pragma("unroll") for (int r = 0; r < RB; ++r)                                        \
        _Pragma("unroll") for (int gi = 0; gi < GB; ++gi)                                   \
          _Pragma("unroll") for (int sp = 0; sp < SP; ++sp) t_[r][gi][sp] = bq[(r + TAPS) % RB][gi][sp]; \
      _Pragma("unroll") for (int r = 0; r < RB; ++r)                                        \
        _Pragma("unroll") for (int gi = 0; gi < GB; ++gi)                                   \
          _Pragma("unroll") for (int sp = 0; sp < SP; ++sp) bq[r][gi][sp] = t_[r][gi][sp];  \
    }                                                                                       \
  }

  float4 hv[kHD];
  int hl[kHD];
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (NBUF == 1) {  // every wave is past the previous chunk (barrier at the loop end)
      fill_halo(halo_base + (size_t)chunk * CG);
      __syncthreads();
    }
    const int hb_off = NBUF == 2 ? ((chunk + lpar) & 1) * buf4 : 0;
    const int hn_off = NBUF == 2 ? ((chunk + lpar + 1) & 1) * buf4 : 0;
    // the last chunk stages chunk 0 of the block's NEXT tile into the idle buffer (persistent), or
    // re-stages itself there (never read): no branch either way
    const bool to_next = chunk + 1 == nchunks && has_next;
    const int chunk_next = min(chunk + 1, nchunks - 1);
    const float4* next_base = to_next ? nx_in_base + (size_t)nx_q_origin * in_cs4
                                      : halo_base + (size_t)chunk_next * CG;  // (uniform: halo origin included)
    const int st_np_total = to_next ? nx_np_total : np_total;
    const int st_nsets = to_next ? nx_nsets : nsets;
    // rows of a chunk whose taps carry the staging code: one piece set is fetched per tap;
    // whatever is still in the ring at the end of the chunk is parked before the barrier
    const int stage_rows = min(ROWS, (st_nsets + TAPS - 1) / TAPS);
    (void)next_base;
    (void)hn_off;
    (void)st_np_total;
    int ps = 0;
#pragma unroll
    for (int d = 0; d < kHD; ++d) {
      hv[d] = make_float4(0.f, 0.f, 0.f, 0.f);
      hl[d] = dummy_loff;
    }
    int arow[G][MF];
    float4 av[2][G][MF][SP];
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
      for (int fm = 0; fm < MF; ++fm) {
        arow[gi][fm] = hb_off + afrag[gi][fm];
#pragma unroll
        for (int sp = 0; sp < SP; ++sp) {
          av[0][gi][fm][sp] = smem4[arow[gi][fm] + sp * QS];  // tap (0,0)
          av[1][gi][fm][sp] = av[0][gi][fm][sp];
        }
      }
    int ky = 0;
    if (NBUF == 2)
      for (; ky < stage_rows; ++ky) RTPOSE_CONV_ROW(1)
    for (; ky < ROWS; ++ky) RTPOSE_CONV_ROW(0)
    if (NBUF == 2) {
#pragma unroll
      for (int d = 0; d < kHD; ++d) smem4[hl[d]] = hv[d];
    }
    __syncthreads();
  }
  RTPOSE_TSTAMP(2);
#undef RTPOSE_CONV_ROW
#undef RTPOSE_CONV_STEP
#undef RTPOSE_PIN
#undef RTPOSE_BLOAD

  // ---- epilogue: bias (+ReLU) (+2x2 max-pool), masked stores, bf16 (RNE) or fp32 ---------
#ifdef RTPOSE_EXP_NO_STORE
  if (A.N > 0) {  // keep the accumulators live, skip the whole epilogue
    float t_ = 0.f;  // every accumulator stays live (else the compiler drops their MFMAs)
#pragma unroll
    for (int fm = 0; fm < MF; ++fm)
#pragma unroll
      for (int fn = 0; fn < NF; ++fn)
#pragma unroll
        for (int r = 0; r < 16; ++r) t_ += acc[fm][fn][r];
    if (t_ == 12345.678f) reinterpret_cast<float*>(g.out)[0] = t_;
    if (!has_next) return;
  }
  if (false) {
#else
  {
#endif
  unsigned short* out_h = reinterpret_cast<unsigned short*>(g.out);
  float* out_f = reinterpret_cast<float*>(g.out);
  const float relu_lo = A.relu ? 0.f : -3.0e38f;  // uniform: v = max(v, relu_lo), no select
  const bool pool = MODE == 1 && A.pool;          // (the fused 2x2 max-pool exists for 2-D tiles only)
  if (TR) {
    // lane = pixel l31 of fragment fm (column of the transposed product), registers = the channels 16 kh .. 16 kh + 15 of
    // fragment fn: 32 bytes of bf16 (split: two [hi x 8 | lo x 8] groups) = two (four) 16-byte stores, nothing through LDS
    int on = 0, oy = 0, ox = 0;
    if (MODE == 0) {
      const int m = m0 + wm * (32 * MF) + l31;
      const int HW = A.H * A.W;
      on = m / HW;
      const int r = m - on * HW;
      oy = r / A.W;
      ox = r - oy * A.W;
    }
    unsigned short* const ob = out_h + g.out_choff + (ntile * BN + wn * (32 * NF) + 16 * kh) * SP;
#pragma unroll
    for (int fm = 0; fm < MF; ++fm) {
      bool ok;
      size_t q;
      if (MODE == 0) {
        ok = m0 + wm * (32 * MF) + fm * 32 + l31 < A.M;
        q = (size_t)g.out_lead + (size_t)(on * g.out_hs + oy) * g.out_ws + ox;
        ox += 32;
        while (ox >= A.W) {
          ox -= A.W;
          if (++oy >= A.H) {
            oy = 0;
            ++on;
          }
        }
      } else {
        int ty, tx;
        tile_local_yx(wm * (32 * MF) + fm * 32 + l31, A.tw_log2, ty, tx);
        ok = (y0 + ty < A.H) && (x0 + tx < A.W);
        q = (size_t)g.out_lead + (size_t)(n_img * g.out_hs + y0 + ty) * g.out_ws + x0 + tx;
      }
      unsigned short* const op = ob + q * g.out_cstride;
#pragma unroll
      for (int fn = 0; fn < NF; ++fn) {
        unsigned hi[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          hi[i] = pack_bf2(fmaxf(acc[fm][fn][2 * i], relu_lo), fmaxf(acc[fm][fn][2 * i + 1], relu_lo));
        if (SP == 1) {
          if (ok) {
            gstore4(op + fn * 32, as_f4(hi[0], hi[1], hi[2], hi[3]));
            gstore4(op + fn * 32 + 8, as_f4(hi[4], hi[5], hi[6], hi[7]));
          }
        } else {
          unsigned lo[8];
#pragma unroll
          for (int i = 0; i < 8; ++i)
            lo[i] = pack_bf2(fmaxf(acc[fm][fn][2 * i], relu_lo) - __uint_as_float(hi[i] << 16),
                             fmaxf(acc[fm][fn][2 * i + 1], relu_lo) - __uint_as_float(hi[i] & 0xffff0000u));
          if (ok) {
            gstore4(op + fn * 64, as_f4(hi[0], hi[1], hi[2], hi[3]));
            gstore4(op + fn * 64 + 8, as_f4(lo[0], lo[1], lo[2], lo[3]));
            gstore4(op + fn * 64 + 16, as_f4(hi[4], hi[5], hi[6], hi[7]));
            gstore4(op + fn * 64 + 24, as_f4(lo[4], lo[5], lo[6], lo[7]));
          }
        }
      }
    }
    RTPOSE_TSTAMP(3);
#ifdef RTPOSE_EXP_TIMELINE
    __builtin_amdgcn_s_waitcnt(0);  // stores retired (vmcnt) - how long does the ack take?
    RTPOSE_TSTAMP(4);
    if (A.dbg && threadIdx.x == 0) A.dbg[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
#endif
  } else if (A.vec_store) {  // uniform
    // The chunk loop ended with a barrier: the halo buffer multiplied last is dead.  Persistent blocks
    // keep the slabs inside that one buffer (the other one already holds the next tile's halo), which is
    // why a wave transposes its tile in two halves of MF/2 fragments (20 KB for the four waves).
    constexpr int HALVES = MF >= 2 ? 2 : 1;
    constexpr int MFH = MF / HALVES;
    constexpr int PITCH2 = (NF * 64 * SP + 16) / 2;  // slab row pitch in bf16 elements
    const int dead4 = (persist ? ((nchunks - 1 + lpar) & 1) * buf4 : 0);  // float4 offset of the dead buffer
    unsigned short* sl = reinterpret_cast<unsigned short*>(smem) + dead4 * 8 + wave * (slab_bytes(MFH, NF, SP) / 2);
    // element of output channel c inside a slab row: c, or (split) its 8-channel group's hi piece
    const int ce = SP == 1 ? l31 : ((l31 >> 3) * 16 + (l31 & 7));
    auto put = [&](int row, int fn, float v) {
      v = fmaxf(v, relu_lo);
      unsigned short* d = sl + row * PITCH2 + fn * 32 * SP + ce;
      const unsigned short hi = to_bf16(v);
      d[0] = hi;
      if (SP == 2) d[8] = to_bf16(v - __uint_as_float((unsigned)hi << 16));
    };
    constexpr int LPR = NF * 4 * SP;  // lanes (16 bytes each) per slab row
    constexpr int RPI = 64 / LPR;   // rows per wave-instruction
    const int rows_h = pool ? 8 * MFH : 32 * MFH;  // slab rows per half
    const int lrow = lane / LPR, c16 = lane % LPR;
    unsigned short* ob = out_h + g.out_choff + (ntile * BN + wn * (32 * NF)) * SP + c16 * 8;
    const int Ho = A.H >> 1, Wo = A.W >> 1;
    // strip mode: (n, y, x) of this lane's first row by division once, then stepped by RPI pixels
    // (integer division is ~40 VALU instructions; 8 row steps of it were a third of the epilogue)
    int sn = 0, sy = 0, sx = 0;
    if (MODE == 0) {
      const int m = m0 + wm * (32 * MF) + lrow;
      const int HW = A.H * A.W;
      sn = m / HW;
      const int r = m - sn * HW;
      sy = r / A.W;
      sx = r - sy * A.W;
    }
#pragma unroll
    for (int hf = 0; hf < HALVES; ++hf) {
      if (!pool) {
#pragma unroll
        for (int fn = 0; fn < NF; ++fn)
#pragma unroll
          for (int fmh = 0; fmh < MFH; ++fmh)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr)
                put(fmh * 32 + rg * 8 + 4 * kh + rr, fn, acc[hf * MFH + fmh][fn][rg * 4 + rr]);
      } else {
#pragma unroll
        for (int fn = 0; fn < NF; ++fn)
#pragma unroll
          for (int fmh = 0; fmh < MFH; ++fmh)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
              put(fmh * 8 + rg * 2 + kh, fn,
                  fmaxf(fmaxf(acc[hf * MFH + fmh][fn][rg * 4 + 0], acc[hf * MFH + fmh][fn][rg * 4 + 1]),
                        fmaxf(acc[hf * MFH + fmh][fn][rg * 4 + 2], acc[hf * MFH + fmh][fn][rg * 4 + 3])));
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // same-wave LDS traffic is in order
      for (int it = 0; it * RPI < rows_h; ++it) {
        const int row = it * RPI + lrow;       // slab row
        const int grow = hf * rows_h + row;    // row (quad, when pooling) inside the wave's tile
        const float4 v = *reinterpret_cast<const float4*>(sl + row * PITCH2 + c16 * 8);
        int n, y, x;
        bool ok;
        if (pool) {  // MODE 1: slab row = quad index inside the wave
          const int qi = wm * (8 * MF) + grow;
          const int hw_log2 = A.tw_log2 - 1;
          n = n_img;
          y = (y0 >> 1) + (qi >> hw_log2);
          x = (x0 >> 1) + (qi & ((1 << hw_log2) - 1));
          ok = y < Ho && x < Wo;
        } else if (MODE == 0) {
          ok = m0 + wm * (32 * MF) + grow < A.M;
          n = sn;
          y = sy;
          x = sx;
          sx += RPI;  // next iteration's row (the walk runs on into the second half)
          while (sx >= A.W) {
            sx -= A.W;
            if (++sy >= A.H) {
              sy = 0;
              ++sn;
            }
          }
        } else {
          int ty, tx;
          tile_local_yx(wm * (32 * MF) + grow, A.tw_log2, ty, tx);
          n = n_img;
          y = y0 + ty;
          x = x0 + tx;
          ok = (y < A.H) && (x < A.W);
        }
        if (ok && row < rows_h) {
          const size_t q = (size_t)g.out_lead + (size_t)(n * g.out_hs + y) * g.out_ws + x;
          gstore4(ob + q * g.out_cstride, v);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // slab reads precede the next half's writes
    }
    RTPOSE_TSTAMP(3);
#ifdef RTPOSE_EXP_TIMELINE
    __builtin_amdgcn_s_waitcnt(0);  // stores retired (vmcnt) - how long does the ack take?
    RTPOSE_TSTAMP(4);
    if (A.dbg && threadIdx.x == 0) A.dbg[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
#endif
  } else {
#pragma unroll
  for (int fn = 0; fn < NF; ++fn) {
    const int ncolf = ncol + fn * 32;
    const bool col_ok = ncolf < g.cout;
    // (split layouts: out_choff counts elements = 2 x channels; out_cmap is not combined with them)
    const int och = (g.out_cmap && col_ok) ? g.out_cmap[ncolf] : (SP == 1 || A.out_f32 ? g.out_choff + ncolf : 0);
    // split: out_choff counts elements (2 per channel) and may sit inside an 8-channel group (the
    // heat-map slice of the concat buffer starts at channel 166): address by absolute channel
    const int ca = (g.out_choff >> 1) + ncolf;
    const int ochs = (ca >> 3) * 16 + (ca & 7);  // hi element, lo at +8
    if (!pool) {
#pragma unroll
      for (int fm = 0; fm < MF; ++fm) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int ml0 = wm * (32 * MF) + fm * 32 + rg * 8 + 4 * kh;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int ml = ml0 + rr;
            int n, y, x;
            bool ok;
            if (MODE == 0) {
              const int m = m0 + ml;
              ok = m < A.M;
              const int HW = A.H * A.W;
              n = m / HW;
              const int r = m - n * HW;
              y = r / A.W;
              x = r - y * A.W;
            } else {
              int ty, tx;
              tile_local_yx(ml, A.tw_log2, ty, tx);
              n = n_img;
              y = y0 + ty;
              x = x0 + tx;
              ok = (y < A.H) && (x < A.W);
            }
            const float v = fmaxf(acc[fm][fn][rg * 4 + rr], relu_lo);
            if (ok && col_ok) {
              const size_t q = (size_t)g.out_lead + (size_t)(n * g.out_hs + y) * g.out_ws + x;
              if (A.out_f32) {
                out_f[q * g.out_cstride + och] = v;
              } else if (SP == 1) {
                out_h[q * g.out_cstride + och] = to_bf16(v);
              } else {
                const unsigned short hi = to_bf16(v);
                out_h[q * g.out_cstride + ochs] = hi;
                out_h[q * g.out_cstride + ochs + 8] = to_bf16(v - __uint_as_float((unsigned)hi << 16));
              }
            }
          }
        }
      }
    } else {
      const int Ho = A.H >> 1, Wo = A.W >> 1;
      const int hw_log2 = A.tw_log2 - 1;
#pragma unroll
      for (int fm = 0; fm < MF; ++fm) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int ml0 = wm * (32 * MF) + fm * 32 + rg * 8 + 4 * kh;
          const int qi = ml0 >> 2;
          const int py = (y0 >> 1) + (qi >> hw_log2);
          const int px = (x0 >> 1) + (qi & ((1 << hw_log2) - 1));
          const float v = fmaxf(fmaxf(fmaxf(acc[fm][fn][rg * 4 + 0], acc[fm][fn][rg * 4 + 1]),
                                      fmaxf(acc[fm][fn][rg * 4 + 2], acc[fm][fn][rg * 4 + 3])), relu_lo);
          if (py < Ho && px < Wo && col_ok) {
            const size_t q = (size_t)g.out_lead + (size_t)(n_img * g.out_hs + py) * g.out_ws + px;
            if (A.out_f32) {
              out_f[q * g.out_cstride + och] = v;
            } else if (SP == 1) {
              out_h[q * g.out_cstride + och] = to_bf16(v);
            } else {
              const unsigned short hi = to_bf16(v);
              out_h[q * g.out_cstride + ochs] = hi;
              out_h[q * g.out_cstride + ochs + 8] = to_bf16(v - __uint_as_float((unsigned)hi << 16));
            }
          }
        }
      }
    }
  }
  }  // scalar epilogue
  }  // epilogue
  if (has_next) {
    if (!TR) __syncthreads();  // every wave is done with its slab before the next tile's staging ring writes that buffer
    grp = nx_grp;
    m0_arg = nx_m0;
    ntile = nx_nt;
    bi = nx_bi;
    lpar = (nchunks + lpar) & 1;  // the buffer the last chunk staged the next tile's chunk 0 into
  }
  };  // tile
  tile(std::true_type{});
  while (has_next) tile(std::false_type{});
}

// 1-D grid, block id -> (group, N tile, M tile); XCD-aware order and half-tile tail exactly
// as conv_mfma_f32 (conv_mfma.hip).
template <int KS, int CK, int MODE, int NBUF, int WM, int NF, int SP, bool TR>
__global__ __launch_bounds__(256, 2) void conv_mfma_bf16(const ConvArgs A) {
  constexpr int MF = 4 / WM;  // block M tile = 128 pixels either way
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = blockIdx.x;
  RTPOSE_TSTAMP(0);
  // The two blocks sharing a CU start together and, having equal work, stay in lock step: their
  // prologues (halo fill) and epilogues coincide and the matrix pipe idles through both (measured:
  // 13k + 9k cycles on a 96k-cycle main loop).  Blocks [n_cu, 2 n_cu) are the second slot of every
  // CU in dispatch order; holding them back half a tile keeps the pairs out of phase for the
  // whole launch, so one block's main loop covers the other's ends.
  if (A.dephase_cycles > 0 && L >= A.n_cu && L < 2 * A.n_cu) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)A.dephase_cycles) __builtin_amdgcn_s_sleep(32);
  }
  const bool small = MODE == 0 && L >= A.npersist;
  const int stride = (MODE == 0 && NBUF == 2 && A.npersist < A.nbig) ? A.npersist : 0;
  int bi = small ? A.nbig + ((L - A.npersist) >> 1) : L;
  int mt = 0, c = 0;
  for (;;) {  // (persistent blocks skip the padding ids of the remapped order)
    decode_block_id(A, bi, mt, c);
    if (mt < A.mtiles) break;
    if (small || stride == 0) return;
    bi += stride;
    if (bi >= A.nbig) return;
  }
  const int nt = c % A.ntiles, grp = c / A.ntiles;
  if (MODE == 1) {
    conv_tile<KS, CK, MODE, NBUF, WM, MF, NF, SP, TR>(A, grp, mt, nt, smem);
  } else if (!small) {
    conv_tile<KS, CK, MODE, NBUF, WM, MF, NF, SP, TR>(A, grp, mt * kBM, nt, smem, bi, stride);
  } else {
    const int m0 = mt * kBM + ((L - A.npersist) & 1) * (kBM / 2);
    if (m0 < A.M) conv_tile<KS, CK, MODE, NBUF, WM, MF / 2, NF, SP, TR>(A, grp, m0, nt, smem);
  }
}

// ---- weight packing: packed[chunk][tap][piece][cout_pad][8 bf16] <- w[cout][cin_src][k][k] ----
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                         int cout, int cin_src, int k, const int32_t* __restrict__ cin_map,
                                         int cin_packed, int ck, int sp, int coutp,
                                         unsigned short* __restrict__ wp, float* __restrict__ bp) {
  const int T = k * k;
  const size_t total = (size_t)T * cin_packed * sp * coutp;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)coutp) bp[i] = (i < (size_t)cout && bias) ? bias[i] : 0.f;
  if (i >= total) return;
  const int e = i & 7;
  size_t r = i >> 3;
  const int n = r % coutp;
  r /= coutp;
  const int plane = r % (ck / 8 * sp);  // piece plane inside the chunk: 8-channel group x (hi, lo)
  r /= (ck / 8 * sp);
  const int tap = r % T;
  const int chunk = r / T;
  const int cg = plane / sp, lo = plane % sp;
  const int c = chunk * ck + cg * 8 + e;
  const int src = cin_map ? cin_map[c] : (c < cin_src ? c : -1);
  float v = 0.f;
  if (n < cout && src >= 0 && src < cin_src) {
    const int ky = tap / k, kx = tap - ky * k;
    v = w[(((size_t)n * cin_src + src) * k + ky) * k + kx];
  }
  const unsigned short hi = to_bf16(v);
  wp[i] = lo ? to_bf16(v - __uint_as_float((unsigned)hi << 16)) : hi;
}

// ---- host side ----------------------------------------------------------------------------
// channels per LDS chunk; the packed weight order depends on it, so pack and launch share it
static int conv_ck(int cin, int k, int sp) {
  if (sp == 2 || cin % 32) return 16;  // split operands: 16 channels = 64 B per pixel per chunk
  // 3x3 layers with deep inputs: 64-channel chunks halve the chunk barriers (conv4_2 1089 -> 1128
  // TFLOP/s); the wide shallow ones (conv1_2, 64 channels at 368x368) lose with them (645 -> 599)
  static int ck3 = 0;  // developer A/B: RTPOSE_BF16_CK3=32|64 forces one size for all 3x3 layers
  if (!ck3) {
    const char* e = dev_env("RTPOSE_BF16_CK3");
    ck3 = e ? atoi(e) : -1;
  }
  if (k == 3 && cin % 64 == 0 && (ck3 == 64 || (ck3 < 0 && cin >= 256))) return 64;
  return (k == 1 && cin % 64 == 0) ? 64 : 32;
}

struct ConvPlan {
  int mode, ck, qs, hw_lds, tw_log2, tiles_x, tiles_y, grid_x, nbuf, nf, wm;
  size_t lds_bytes;
};

static int round_qs(int npix) {
  int qs = npix;
  while ((qs & 3) != 2) ++qs;
  return qs;
}
static int halo_row_lds(int tw, int p) {
  int w = tw + 2 * p;
  while ((w & 15) != 8) ++w;
  return w;
}

static int plan_conv(const rtpose_conv_desc& d, int N, int H, int W, int sp, ConvPlan* pl) {
  const int P = d.k / 2;
  pl->ck = conv_ck(d.cin, d.k, sp);
  const int M = N * H * W;
  const int max_sets = d.k * d.k;  // one piece set per tap, the ring is flushed at the chunk end
  const int cg = pl->ck / 8 * sp;
  pl->nbuf = d.k == 1 ? 1 : 2;
  const size_t tail = pl->nbuf == 2 ? 256 * 16 : 0;  // dummy park slots
  // widest map that still uses strips (see conv_mfma.hip); measured on the 92 x 92 layers: bf16
  // 841 (tiles) vs 835 TFLOP/s (strips), bf16x3 350 vs 356 - the longer tap loop of the split form
  // amortises the strips' longer halo
  static int strip_maxw = -1;
  if (strip_maxw < 0) {
    const char* e = dev_env("RTPOSE_BF16_STRIP_MAXW");
    strip_maxw = e ? atoi(e) : 0;
  }
  bool strip = (W <= (strip_maxw > 0 ? strip_maxw : (sp == 2 ? 128 : 64))) && !d.pool;
  {
    static int force_tile = -1;  // developer A/B: RTPOSE_BF16_FORCE_TILE=k forces 2-D tiles for k x k convs
    if (force_tile < 0) {
      const char* e = dev_env("RTPOSE_BF16_FORCE_TILE");
      force_tile = e ? atoi(e) : 0;
    }
    if (force_tile == d.k) strip = false;
  }
  if (strip) {
    const rtpose_layout& l = d.lin;
    const int lb = (kBM - 1) + ((kBM - 1) / W + 1) * (l.ws - W) +
                   ((kBM - 1) / (H * W) + 1) * (l.hs - H) * l.ws + 2 * P * l.ws + 2 * P + 1;
    const int qs = round_qs(lb);
    const size_t lds = (size_t)pl->nbuf * cg * qs * 16 + tail;
    if ((pl->nbuf == 2 && ceil_div(qs * cg, 256) > max_sets) || lds > 80 * 1024) strip = false;
    if (strip) {
      pl->mode = 0;
      pl->qs = qs;
      pl->hw_lds = 0;
      pl->tw_log2 = 0;
      pl->tiles_x = pl->tiles_y = 0;
      pl->grid_x = ceil_div(M, kBM);
      pl->lds_bytes = lds;
      return 0;
    }
  }
  int best_tw = 0;
  long best_cost = -1;
  for (int twl = 2; twl <= 6; ++twl) {
    const int tw = 1 << twl, th = kBM >> twl;
    if (th < 2) continue;
    const long cost = (long)ceil_div(W, tw) * ceil_div(H, th);
    const long halo = (long)(th + 2 * P) * halo_row_lds(tw, P);
    // the halo must fit the staging schedule (one piece set per tap) and the LDS budget
    if (pl->nbuf == 2 && ceil_div(round_qs((int)halo) * cg, 256) > max_sets) continue;
    if ((size_t)pl->nbuf * cg * round_qs((int)halo) * 16 + tail > 80 * 1024) continue;
    const long key = cost * 100000 + halo;
    if (best_cost < 0 || key < best_cost) {
      best_cost = key;
      best_tw = twl;
    }
  }
  if (best_cost < 0) return fail(RTPOSE_E_INVAL, "conv bf16: no 2-D tile fits the staging schedule");
  const int tw = 1 << best_tw, th = kBM >> best_tw;
  pl->mode = 1;
  pl->tw_log2 = best_tw;
  pl->hw_lds = halo_row_lds(tw, P);
  const int npix = (th + 2 * P) * pl->hw_lds;
  pl->qs = round_qs(npix);
  pl->tiles_x = ceil_div(W, tw);
  pl->tiles_y = ceil_div(H, th);
  pl->grid_x = N * pl->tiles_x * pl->tiles_y;
  pl->lds_bytes = (size_t)pl->nbuf * cg * pl->qs * 16 + tail;
  if (pl->nbuf == 2 && ceil_div(pl->qs * cg, 256) > max_sets)
    return fail(RTPOSE_E_INVAL, "conv bf16: halo too large for the staging schedule");
  if (pl->lds_bytes > 80 * 1024) return fail(RTPOSE_E_INVAL, "conv bf16: halo exceeds the LDS budget");
  return 0;
}

template <int KS, int CK, int MODE, int NBUF, int WM, int NF, int SP, bool TR>
static int launch_tr(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static PerDeviceOnce attr_set;  // zero-initialised; the attribute is per device
  const int dev = current_device();
  auto kern = conv_mfma_bf16<KS, CK, MODE, NBUF, WM, NF, SP, TR>;
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set.set(dev);
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}
// the transposed form exists for the k x k kernels (a.tr is never set for k = 1)
template <int KS, int CK, int MODE, int NBUF, int WM, int NF, int SP>
static int launch_inst(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  if constexpr (KS != 1) {
    if (a.tr) return launch_tr<KS, CK, MODE, NBUF, WM, NF, SP, true>(a, grid, lds, s);
  }
  return launch_tr<KS, CK, MODE, NBUF, WM, NF, SP, false>(a, grid, lds, s);
}

}  // namespace bf

// d[i].in / w_packed / out point at bf16 data (out: fp32 when out_f32); layouts count bf16
// ELEMENTS per pixel (split = 1: two elements per channel, [hi x 8 | lo x 8] pieces; `cin` and
// `cout` still count channels).
int conv2d_bf16_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, int out_f32, int split,
                       hipStream_t s) {
  using namespace bf;
  const int sp = split ? 2 : 1;
  if (!d || ngroups < 1 || ngroups > 2) return fail(RTPOSE_E_INVAL, "conv2d_bf16: ngroups must be 1 or 2");
  RTPOSE_REFUSE_PLANES(d, ngroups, "conv2d_bf16");
  const rtpose_conv_desc& d0 = d[0];
  if (d0.k != 1 && d0.k != 3 && d0.k != 7) return fail(RTPOSE_E_INVAL, "conv2d_bf16: k must be 1, 3 or 7");
  if (d0.cin % 16 != 0 || d0.cin <= 0) return fail(RTPOSE_E_INVAL, "conv2d_bf16: cin must be a multiple of 16");
  if (N <= 0 || H <= 0 || W <= 0) return fail(RTPOSE_E_INVAL, "conv2d_bf16: empty tensor");
  const int P = d0.k / 2;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < ngroups; ++i) {
    const rtpose_conv_desc& di = d[i];
    if (di.k != d0.k || di.cin != d0.cin || di.relu != d0.relu || di.pool != d0.pool ||
        cout_pad(di.cout) != cout_pad(d0.cout) || di.lin.ws != d0.lin.ws || di.lin.hs != d0.lin.hs)
      return fail(RTPOSE_E_INVAL, "conv2d_bf16: grouped convs must share geometry");
    if (di.lin.ws < W + P || di.lin.hs < H + P || di.lin.lead < P * di.lin.ws + P)
      return fail(RTPOSE_E_INVAL, "conv2d_bf16: input layout gap smaller than the conv padding");
    if ((di.lin.cstride % (8 * sp)) || (di.lin.choff % (8 * sp)))
      return fail(RTPOSE_E_INVAL, "conv2d_bf16: input slice must be 16-byte aligned");
    if (di.lin.choff + di.cin * sp > di.lin.cstride)
      return fail(RTPOSE_E_INVAL, "conv2d_bf16: input slice exceeds cstride");
    if (split && di.out_cmap) return fail(RTPOSE_E_INVAL, "conv2d_bf16x3: out_cmap is not supported");
    ConvGroup& g = a.g[i];
    g.in = reinterpret_cast<const unsigned short*>(di.in);
    g.w = reinterpret_cast<const float4*>(di.w_packed);
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
    g.out_cmap = di.out_cmap;
  }
  if (d0.pool && ((H | W) & 1)) return fail(RTPOSE_E_INVAL, "conv2d_bf16: fused pool needs even H and W");
  {
    // 3x3 layers with 64 input channels (conv1_2, conv2_1) have their own kernel: the whole K of a tile in one LDS halo,
    // persistent blocks (conv_c64_bf16.hip)
    static int c64_env = -1;  // developer A/B: RTPOSE_BF16_C64=0 keeps the generic kernel for them
    if (c64_env < 0) {
      const char* e = dev_env("RTPOSE_BF16_C64");
      c64_env = e ? atoi(e) : 1;
    }
    // (it reads the packing for 32-channel chunks: what conv_ck gives these layers unless a developer build forces 64)
    if (c64_env && conv_ck(d0.cin, d0.k, sp) == 32 && conv_c64_bf16_fits(d, ngroups, N, H, W, out_f32, split))
      return conv_c64_bf16_launch(d, N, H, W, s);
  }
  ConvPlan pl;
  int rc = plan_conv(d0, N, H, W, sp, &pl);
  if (rc) return rc;
  a.N = N;
  a.H = H;
  a.W = W;
  a.M = N * H * W;
  a.cin = d0.cin;
  a.relu = d0.relu;
  a.pool = d0.pool;
  a.out_f32 = out_f32 ? 1 : 0;
  a.vec_store = !out_f32;
  for (int i = 0; i < ngroups; ++i)
    if (d[i].out_cmap || (d[i].cout % kConvBN) || (d[i].lout.cstride % (8 * sp)) || (d[i].lout.choff % (8 * sp)))
      a.vec_store = 0;
#ifdef RTPOSE_EXP_SCALAR_STORE
  a.vec_store = 0;
#endif
  {
    static int tr_env = -1;  // developer A/B: RTPOSE_BF16_TR=0 keeps the LDS-transposing epilogue
    if (tr_env < 0) {
      const char* e = dev_env("RTPOSE_BF16_TR");
      tr_env = e ? atoi(e) : 1;
    }
    a.tr = (a.vec_store && !d0.pool && d0.k != 1 && tr_env) ? 1 : 0;
  }
  a.qs = pl.qs;
  a.hw_lds = pl.hw_lds;
  a.tw_log2 = pl.tw_log2;
  a.tiles_x = pl.tiles_x;
  a.tiles_y = pl.tiles_y;
  const int n_cu = device_cu_count();  // of the device this launch goes to
  a.mtiles = pl.grid_x;
  const int coutp = cout_pad(d0.cout);
  // 128-channel N tiles for the k x k layers whose cout allows it: 1 x 4 waves of 128 x 32
  // (default) or 2 x 2 waves of 64 x 64 (RTPOSE_BF16_WAVES=22, kept for A/B); else 128 x 64
  static int waves_env = 0;
  if (!waves_env) {
    const char* e = dev_env("RTPOSE_BF16_WAVES");
    waves_env = e ? atoi(e) : 14;  // 14: per-kernel-size default, 41: 1 x 4 everywhere, 22: 2 x 2 everywhere
  }
  const bool wide = d0.k != 1 && coutp % 128 == 0;
  // measured (32 x 368 x 368): 7x7 layers 1237 (1 x 4) vs 1166 (2 x 2) TFLOP/s, 3x3 layers 786 vs 812
  pl.wm = (wide && (waves_env == 14 ? d0.k == 7 : waves_env != 22)) ? 1 : 2;
  pl.nf = (wide && pl.wm == 2) ? 2 : 1;
  a.ntiles = coutp / (32 * pl.nf * (4 / pl.wm));
  a.ncombo = a.ntiles * ngroups;
  a.xcd_remap = (a.ncombo > 1 && a.mtiles >= 64) ? 1 : 0;
  const long ids = a.xcd_remap ? (long)8 * a.ncombo * ceil_div(a.mtiles, 8) : (long)a.mtiles * a.ncombo;
  if (ids > 0x7fffffffL) return fail(RTPOSE_E_INVAL, "conv2d_bf16: grid too large");
  a.nbig = (int)ids;
  if (pl.mode == 0) {  // tail quantisation, see conv_mfma.hip
    const int slots = n_cu * 2;
    const int total = (int)ids;
    const int rem = total % slots;
    if (total > slots && rem > 0 && 2 * rem <= n_cu) a.nbig = total - rem;
    if (total <= n_cu) a.nbig = 0;
  }
  // persistent strips: 2 blocks per CU walk the full tiles (k x k layers; the grouped convs share the
  // input pixel pitch, which the cross-tile staging relies on); RTPOSE_BF16_PERSIST=0 (developer builds)
  // restores one tile per block
  a.npersist = a.nbig;
  {
    static int persist_env = -1;
    if (persist_env < 0) {
      const char* e = dev_env("RTPOSE_BF16_PERSIST");
      persist_env = e ? atoi(e) : 1;
    }
    const int slots = (n_cu * 2) & ~7;  // multiple of 8: a block stays on its XCD's ids
    bool same_pitch = true;
    for (int i = 1; i < ngroups; ++i) same_pitch = same_pitch && d[i].lin.cstride == d0.lin.cstride && d[i].lin.lead == d0.lin.lead;
    // the epilogue's half-size slabs must fit inside ONE halo buffer (the other holds the next tile's halo)
    const int mf = pl.wm == 1 ? 4 : 2;
    const size_t half_slabs = (size_t)4 * slab_bytes(mf / 2, pl.nf, sp);
    const size_t buf_bytes = (size_t)(pl.ck / 8 * sp) * pl.qs * 16;
    if (persist_env && pl.mode == 0 && pl.nbuf == 2 && d0.k != 1 && same_pitch && slots >= 8 && a.nbig > slots &&
        (!a.vec_store || a.tr || half_slabs <= buf_bytes))
      a.npersist = slots;
  }
  dim3 grid((unsigned)(a.npersist + 2 * (ids - a.nbig)), 1, 1);
  {
    static int dephase_env = -1;  // percent of one block's MFMA time; 0 = off
    if (dephase_env < 0) {
      const char* e = dev_env("RTPOSE_BF16_DEPHASE");
      dephase_env = e ? atoi(e) : 0;
    }
    a.n_cu = n_cu;
    // one block's matrix time with the SIMD to itself: chunks x taps x MFMAs x 32 cycles
    const long mfma_cycles = (long)(d0.cin / pl.ck) * d0.k * d0.k * (pl.ck / 16) * 4 * 32 * (split ? 3 : 1);
    a.dephase_cycles = (pl.nbuf == 2 && (long)grid.x > 2L * n_cu) ? (int)(mfma_cycles * dephase_env / 100) : 0;
  }
#ifdef RTPOSE_EXP_TIMELINE
  {  // developer build: time stamps of the LAST 7x7 launch, dumped by rtpose_debug_timeline_dump
    extern unsigned long long* g_dbg_buf;
    extern unsigned g_dbg_blocks;
    if (!g_dbg_buf) (void)hipMalloc(&g_dbg_buf, (size_t)8192 * 8 * 8);
    if (d0.k == 7 && grid.x <= 8192) {
      (void)hipMemsetAsync(g_dbg_buf, 0, (size_t)grid.x * 64, s);
      a.dbg = g_dbg_buf;
      g_dbg_blocks = grid.x;
    }
  }
#endif
  {
    const size_t slab = (size_t)4 * slab_bytes(pl.wm == 1 ? 4 : 2, pl.nf, sp);
    if (a.vec_store && !a.tr && pl.lds_bytes < slab) pl.lds_bytes = slab;
  }
  if (split) {
#define RTPOSE_CONV_CASE_X3(KS_, MODE_, NBUF_)                                                    \
  if (d0.k == KS_ && pl.mode == MODE_) {                                                          \
    if (pl.wm == 1) return launch_inst<KS_, 16, MODE_, NBUF_, (KS_ != 1) ? 1 : 2, 1, 2>(a, grid, pl.lds_bytes, s); \
    if (pl.nf == 2) return launch_inst<KS_, 16, MODE_, NBUF_, 2, (KS_ != 1) ? 2 : 1, 2>(a, grid, pl.lds_bytes, s); \
    return launch_inst<KS_, 16, MODE_, NBUF_, 2, 1, 2>(a, grid, pl.lds_bytes, s);                  \
  }
    RTPOSE_CONV_CASE_X3(3, 0, 2)
    RTPOSE_CONV_CASE_X3(3, 1, 2)
    RTPOSE_CONV_CASE_X3(7, 0, 2)
    RTPOSE_CONV_CASE_X3(7, 1, 2)
    RTPOSE_CONV_CASE_X3(1, 0, 1)
    RTPOSE_CONV_CASE_X3(1, 1, 1)
#undef RTPOSE_CONV_CASE_X3
    return fail(RTPOSE_E_INVAL, "conv2d_bf16x3: no kernel instance for k=%d mode=%d", d0.k, pl.mode);
  }
#define RTPOSE_CONV_CASE(KS_, CK_, MODE_, NBUF_)                                                  \
  if (d0.k == KS_ && pl.ck == CK_ && pl.mode == MODE_) {                                          \
    if (pl.wm == 1) return launch_inst<KS_, CK_, MODE_, NBUF_, (KS_ != 1) ? 1 : 2, 1, 1>(a, grid, pl.lds_bytes, s); \
    if (pl.nf == 2) return launch_inst<KS_, CK_, MODE_, NBUF_, 2, (KS_ != 1) ? 2 : 1, 1>(a, grid, pl.lds_bytes, s); \
    return launch_inst<KS_, CK_, MODE_, NBUF_, 2, 1, 1>(a, grid, pl.lds_bytes, s);                 \
  }
  RTPOSE_CONV_CASE(3, 16, 0, 2)
  RTPOSE_CONV_CASE(3, 16, 1, 2)
  RTPOSE_CONV_CASE(3, 32, 0, 2)
  RTPOSE_CONV_CASE(3, 32, 1, 2)
  RTPOSE_CONV_CASE(3, 64, 0, 2)
  RTPOSE_CONV_CASE(3, 64, 1, 2)
  RTPOSE_CONV_CASE(7, 32, 0, 2)
  RTPOSE_CONV_CASE(7, 32, 1, 2)
  RTPOSE_CONV_CASE(7, 16, 0, 2)
  RTPOSE_CONV_CASE(7, 16, 1, 2)
  RTPOSE_CONV_CASE(1, 16, 0, 1)
  RTPOSE_CONV_CASE(1, 16, 1, 1)
  RTPOSE_CONV_CASE(1, 32, 0, 1)
  RTPOSE_CONV_CASE(1, 32, 1, 1)
  RTPOSE_CONV_CASE(1, 64, 0, 1)
  RTPOSE_CONV_CASE(1, 64, 1, 1)
#undef RTPOSE_CONV_CASE
  return fail(RTPOSE_E_INVAL, "conv2d_bf16: no kernel instance for k=%d ck=%d mode=%d", d0.k, pl.ck, pl.mode);
}

#ifdef RTPOSE_EXP_TIMELINE
unsigned long long* g_dbg_buf = nullptr;
unsigned g_dbg_blocks = 0;
#endif

int pack_weights_bf16_launch(const float* w, const float* bias, int cout, int cin_src, int k,
                             const int32_t* cin_map, int cin_packed, void* wp, float* bp, int split,
                             hipStream_t s) {
  const int sp = split ? 2 : 1;
  if (cin_packed % 16 || (cin_packed < cin_src && !cin_map))
    return fail(RTPOSE_E_INVAL, "pack_bf16: cin_packed must be a multiple of 16 and >= cin_src");
  if (k != 1 && k != 3 && k != 7) return fail(RTPOSE_E_INVAL, "pack_bf16: k must be 1, 3 or 7");
  const int coutp = cout_pad(cout);
  const size_t total = (size_t)k * k * cin_packed * sp * coutp;
  const int threads = 256;
  const unsigned blocks = (unsigned)((total + threads - 1) / threads);
  hipLaunchKernelGGL(bf::pack_weights_bf16_kernel, dim3(blocks), dim3(threads), 0, s, w, bias, cout, cin_src,
                     k, cin_map, cin_packed, bf::conv_ck(cin_packed, k, sp), sp, coutp,
                     reinterpret_cast<unsigned short*>(wp), bp);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace rtpose

extern "C" {

size_t rtpose_packed_weight_bytes_bf16(int cout, int cin, int k) {
  const int cinp = rtpose::ceil_div(cin, 16) * 16;
  // + two (chunk, tap) blocks of slack (<= 64 channels each): the B prefetch runs two taps ahead
  return (size_t)(k * k * cinp + 128) * rtpose::cout_pad(cout) * 2;
}

int rtpose_pack_conv_weights_bf16(const float* w_oihw, const float* bias, int cout, int cin_src, int k,
                                  const int32_t* cin_map, int cin_packed, void* w_packed,
                                  float* bias_packed, void* stream) {
  return rtpose::pack_weights_bf16_launch(w_oihw, bias, cout, cin_src, k, cin_map, cin_packed, w_packed,
                                          bias_packed, 0, rtpose::as_stream(stream));
}

size_t rtpose_packed_weight_bytes_bf16x3(int cout, int cin, int k) {
  return 2 * rtpose_packed_weight_bytes_bf16(cout, cin, k);
}

int rtpose_pack_conv_weights_bf16x3(const float* w_oihw, const float* bias, int cout, int cin_src, int k,
                                    const int32_t* cin_map, int cin_packed, void* w_packed,
                                    float* bias_packed, void* stream) {
  return rtpose::pack_weights_bf16_launch(w_oihw, bias, cout, cin_src, k, cin_map, cin_packed, w_packed,
                                          bias_packed, 1, rtpose::as_stream(stream));
}

int rtpose_conv2d_bf16x3(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, int out_f32,
                         void* stream) {
  return rtpose::conv2d_bf16_launch(d, ngroups, N, H, W, out_f32, 1, rtpose::as_stream(stream));
}

int rtpose_conv2d_bf16(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, int out_f32,
                       void* stream) {
  return rtpose::conv2d_bf16_launch(d, ngroups, N, H, W, out_f32, 0, rtpose::as_stream(stream));
}

}  // extern "C"

#ifdef RTPOSE_EXP_TIMELINE
extern "C" int rtpose_debug_timeline_dump(unsigned long long* host, unsigned cap_blocks) {
  using namespace rtpose;
  if (!g_dbg_buf) return 0;
  (void)hipDeviceSynchronize();
  const unsigned n = g_dbg_blocks < cap_blocks ? g_dbg_blocks : cap_blocks;
  (void)hipMemcpy(host, g_dbg_buf, (size_t)n * 64, hipMemcpyDeviceToHost);
  return (int)n;
}
#endif
