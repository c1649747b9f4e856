// fp32 implicit-GEMM convolution for gfx950 (MI355X), stride 1, "same" padding,
// k in {1,3,7}, fused bias (+ReLU) (+2x2 max-pool).
//
// Stands in for the nn.Conv2d / nn.ReLU / nn.MaxPool2d modules instantiated by
// lib/network/rtpose_vgg.py:23-35 and :49-55 (ATen kernels on the reference).
//
// Design (MI355X-first, not a translation of any CUDA kernel):
//  * GEMM view: M = output pixels, N = output channels, K = taps x channels.
//    Block tile 128(M) x 64(N), 4 wave64s as 2(M) x 2(N), each wave a 64 x 32
//    tile = two 32x32 accumulators of v_mfma_f32_32x32x2_f32 (exact fp32, the
//    only fp32-input matrix op on CDNA4; 64 cycles/instruction/SIMD).
//  * Activations use the shared-gap padded NHWC layout (include/rtpose_mi355x.h
//    §1): a stencil tap is a constant pixel offset, so the A operand is NOT an
//    im2col gather: per 16-channel chunk the block stages ONE halo of input
//    pixels in LDS and re-uses it for all k*k taps (49x re-use for 7x7).
//    LDS image is [channel/4][pixel][4 floats]; every A fragment read is one
//    conflict-free ds_read_b128 per lane (4 consecutive k for its k-half).
//  * The B operand (weights, pre-packed [chunk][tap][c/4][cout][4]) goes
//    straight from L2 to registers with a one-tap-deep register prefetch: a
//    wave reads 1 KiB contiguous per instruction, no LDS, no per-tap barrier.
//    One barrier per channel chunk (k*k*16 MFMAs per wave apart).
//  * The next chunk's halo is fetched one 16-byte piece per thread per tap
//    underneath the MFMAs and written to the other LDS buffer.
//  * Two tilings: MODE 0 "strip" = 128 consecutive pixels of the flattened
//    (n,y,x) space (no tile waste on the 46x46 maps: 32*46*46 = 529 * 128);
//    MODE 1 = TH x TW 2-D tiles inside one image for wide maps (bounded halo).
//  * 2 blocks per CU (launch bounds) so one block's chunk barrier / epilogue
//    hides under the other's MFMAs.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "common.h"
#include "conv_exp.h"

namespace rtpose {

typedef float floatx16 __attribute__((ext_vector_type(16)));
// Explicit global address space: if clang loses track of a pointer's provenance it
// falls back to FLAT loads, which also tick lgkmcnt and drag every LDS read behind a
// full `s_waitcnt vmcnt(0) lgkmcnt(0)` (seen in the tap loop; cost several %).
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef const floatx4 __attribute__((address_space(1)))* gcf4_t;
__device__ __forceinline__ float4 gload4(const void* p) {
  const floatx4 v = *(gcf4_t)(unsigned long long)(p);
  return make_float4(v[0], v[1], v[2], v[3]);
}

struct ConvGroup {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  int in_cstride, in_choff, in_ws, in_hs, in_lead;
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  int cout, cout_pad;
  const int32_t* out_cmap;  // optional output-channel scatter (see rtpose_conv_desc)
};

struct ConvArgs {
  ConvGroup g[2];
  int N, H, W, M;     // output == input spatial size (before pooling)
  int cin;            // packed input channels
  int relu, pool;
  int qs;             // LDS pixels per channel-group plane
  int hw_lds;         // MODE 1: LDS row stride of the halo (pixels)
  int tw_log2;        // MODE 1: log2(tile width); tile height = 128 >> tw_log2
  int tiles_x, tiles_y;
  int mtiles, ntiles, nbig;  // block-id decoding (see conv_mfma_f32)
  int ncombo, xcd_remap;
  int dephase_mode;          // 0 off, 1: ids [n_cu, 2 n_cu), 2: odd ids (first wave of blocks only)
  int dephase_cycles, n_cu;  // start-up delay that puts the 2 blocks of a CU half a tile apart
  unsigned long long* dbg;   // RTPOSE_EXP_TIMELINE builds only: 8 x u64 per block
};

constexpr int kBM = 128;
// 1x1 convs: CK-channel sub-chunks per LDS buffer (see conv_tile); developer knob (conv_exp.h)
constexpr int kTB1x1 = RTPOSE_EXP_TB1X1;

template <int KS>
struct PiecesPerTap {
  static constexpr int value = (KS == 1) ? 6 : 1;
};

// m_local (0..127) -> (ty, tx) inside a 2-D tile.  2x2 quads are 4 consecutive
// m so that a fused max-pool is a max over 4 accumulator registers of one lane.
__device__ __forceinline__ void tile_local_yx(int ml, int tw_log2, int& ty, int& tx) {
  const int qi = ml >> 2;
  const int hw_log2 = tw_log2 - 1;  // quads per tile row
  ty = ((qi >> hw_log2) << 1) + ((ml >> 1) & 1);
  tx = ((qi & ((1 << hw_log2) - 1)) << 1) + (ml & 1);
}

// NBUF = 2: the next channel chunk's halo is fetched underneath the MFMAs into a second
//           LDS buffer (2 blocks per CU).
// NBUF = 1: one halo buffer, re-filled between chunks behind a barrier; half the LDS and
//           <= 128 VGPRs, so 4 blocks share a CU and hide each other's refills
//           (occupancy instead of software pipelining).
// MF = 32-row M fragments per wave: 2 = the normal 128-pixel block tile, 1 = a 64-pixel
// half tile used only for the last, partial wave of blocks of a strip-mode launch (tail
// quantisation: see plan_conv).
// NF = 32-column N fragments per wave: 1 -> block tile 128 x 64, 2 -> 128 x 128 (wave tile
// 64 x 64).  NF = 2 halves the B (weight) and A (LDS) operand bytes per MFMA; measured, the B
// loads cost ~4.6 % of the forward and half of them ~2 %.
template <int KS, int CK, int MODE, int NBUF, int MF, int NF>
__device__ __forceinline__ void conv_tile(const ConvArgs& A, const ConvGroup& g, const int m0_arg,
                                          const int ntile, float* smem) {
  constexpr int P = KS / 2;
  constexpr int BMT = 64 * MF;  // pixels per block tile
  constexpr int CG = CK / 4;  // 16-byte channel groups per chunk
  constexpr int G = CK / 8;   // 8-deep k groups per chunk (4 MFMAs each)
  constexpr int GB = G * NF;  // B registers (float4) per tap: [n-fragment][k-group]
  constexpr int PPT = PiecesPerTap<KS>::value;
  // 1x1 convs have no spatial taps to re-use a halo over.  The tap loop can instead walk
  // the channel axis: one LDS buffer holds TB consecutive CK-channel sub-chunks of the pixel
  // tile ([TB*CG planes][pixel][4]) and tap t multiplies sub-chunk t, with the k x k
  // machinery (B/A prefetch, next buffer staged under the MFMAs) unchanged.  Measured with
  // TB = 4 (double-buffered, 2 blocks/CU): SLOWER than TB = 1 single-buffered at 4 blocks/CU
  // (VGG 1x1 layers 1.27 vs 1.04 ms, ShuffleNetV2 pointwise 13.6 vs 12.7 ms): these GEMMs
  // are short (K = 24..512), so per-block prologue/epilogue latency dominates and is hidden
  // better by occupancy than by in-block pipelining.  TB stays a knob for a future
  // multi-tile (persistent) 1x1 kernel.
  constexpr int TB = (KS == 1) ? kTB1x1 : 1;
  constexpr int CGB = CG * TB;              // channel-group planes per LDS buffer
  constexpr int TAPS = (KS == 1) ? TB : KS; // taps per "row" of the tap loop
  constexpr int ROWS = (KS == 1) ? 1 : KS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int l31 = lane & 31, kh = lane >> 5;

  const int QS = A.qs;

  // ---- block -> tile -------------------------------------------------------
  int m0 = 0, n_img = 0, y0 = 0, x0 = 0;  // MODE 0 uses m0; MODE 1 uses the rest
  int q_origin, np_pix, row_lds;
  int qc0 = 0;
  if (MODE == 0) {
    m0 = m0_arg;
    const int HW = A.H * A.W;
    const int n = m0 / HW, r = m0 - n * HW;
    const int y = r / A.W, x = r - y * A.W;
    qc0 = g.in_lead + (n * g.in_hs + y) * g.in_ws + x;
    int ml = min(m0 + BMT, A.M) - 1;
    const int n2 = ml / HW, r2 = ml - n2 * HW;
    const int y2 = r2 / A.W, x2 = r2 - y2 * A.W;
    const int qcl = g.in_lead + (n2 * g.in_hs + y2) * g.in_ws + x2;
    q_origin = qc0 - P * g.in_ws - P;
    np_pix = qcl + P * g.in_ws + P - q_origin + 1;
    row_lds = g.in_ws;
  } else {
    int b = m0_arg;  // MODE 1: the tile index
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
  const int np_total = np_pix * CGB;  // 16-byte pieces per LDS buffer

  // ---- per-lane A fragment bases (LDS pixel index of this lane's row) ------
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

  // ---- halo staging helpers -------------------------------------------------
  // A halo is staged as 16-byte pieces: piece idx = (pixel idx/CG, channel group idx%CG);
  // thread `tid` owns pieces tid, tid+256, ... ("sets").  LDS image: [group][pixel][4].
  const int in_cstride = g.in_cstride, in_ws = g.in_ws;
  const float* in_base = g.in + g.in_choff;
  constexpr int PIXSET = 256 / CGB;             // halo pixels covered by one piece set
  const int pj = tid % CGB, ppix0 = tid / CGB;  // this thread's group and first pixel
  const unsigned hw_inv = MODE == 1 ? (65536u + A.hw_lds - 1) / A.hw_lds : 0u;
  // global float offset (from the chunk's first channel) and LDS float offset of piece
  // (set, tid).  MODE 0 may run past the halo (never past the buffer: layout slack);
  // the caller masks the LDS side.
  auto piece_goff = [&](int set) -> size_t {
    int pix = set * PIXSET + ppix0;
    int q;
    if (MODE == 0) {
      q = q_origin + pix;
    } else {
      pix = min(pix, np_pix - 1);
      const int hy = (int)(((unsigned)pix * hw_inv) >> 16), hx = pix - hy * A.hw_lds;
      q = q_origin + hy * in_ws + hx;
    }
    return (size_t)q * in_cstride + pj * 4;
  };
  // LDS offsets are kept in float4 units: the compiler then knows every access is
  // 16-byte aligned and emits ds_read_b128 / ds_write_b128 (with float offsets it fell
  // back to ds_read2_b32 pairs: 2x the instructions and 4-way bank conflicts)
  float4* smem4 = reinterpret_cast<float4*>(smem);
  const int buf4 = CGB * QS;  // float4 per halo buffer
  auto piece_loff = [&](int set) -> int { return pj * QS + set * PIXSET + ppix0; };
  // a thread with nothing to park writes its stale registers to a private dummy slot
  // behind the two buffers, so the park step needs no per-thread branch
  const int dummy_loff = NBUF * buf4 + tid;
  const int nsets = (np_total + 255) / 256;  // piece sets per chunk

  // ---- B operand pointers (advance one (chunk,tap) block per tap) ---------------
  const int nchunks = A.cin / CK;
  const int ncol = ntile * (kConvBN * NF) + wn * (32 * NF) + l31;  // column of n-fragment 0
  const float4* bq[GB];
#pragma unroll
  for (int fn = 0; fn < NF; ++fn)
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
      bq[fn * G + gi] = reinterpret_cast<const float4*>(g.w) + (size_t)(2 * gi + kh) * g.cout_pad + ncol + fn * 32;
  const size_t b_it_stride = (size_t)CG * g.cout_pad;  // float4 per (chunk,tap)

  // B lives in three register sets: the tap being multiplied, the next one, and the one
  // in flight from L2 (two taps of lead: one tap was not enough to cover L2 latency
  // under load - dropping the B loads was worth +7% on the 7x7 layers)
  float4 s0[GB], s1[GB], s2[GB];
#pragma unroll
  for (int gi = 0; gi < GB; ++gi) {
    s0[gi] = gload4(bq[gi]);
    bq[gi] += b_it_stride;
    s1[gi] = gload4(bq[gi]);
  }

  // ---- halo fill used by the prologue (NBUF 2) / before every chunk (NBUF 1) ------
  auto fill_halo = [&](const float* src, int ntaps) {  // ntaps: sub-chunks that exist in this buffer
    const bool ch_ok = pj < ntaps * CG;
    constexpr int FD = (KS == 1 && kTB1x1 > 1) ? 8 : 4;  // loads in flight per thread
    for (int set0 = 0; set0 < nsets; set0 += FD) {
      float4 t[FD];
#pragma unroll
      for (int u = 0; u < FD; ++u)
        if (ch_ok && set0 + u < nsets && (set0 + u) * 256 + tid < np_total) t[u] = gload4(src + piece_goff(set0 + u));
#pragma unroll
      for (int u = 0; u < FD; ++u)
        if (ch_ok && set0 + u < nsets && (set0 + u) * 256 + tid < np_total) smem4[piece_loff(set0 + u)] = t[u];
    }
  };
  const int nbig = (nchunks + TB - 1) / TB;  // LDS buffer fills per block
  if (NBUF == 2) {
    fill_halo(in_base, min(TB, nchunks));
    __syncthreads();
  }
  // (NBUF 1, measured and rejected: issuing chunk c+1's loads before chunk c is multiplied and
  //  parking them afterwards - "load early / write late" - made the ShuffleNetV2 pointwise layers
  //  5-15 % SLOWER than the plain refill below with 4 blocks per CU hiding each other's latency.)

  RTPOSE_TSTAMP(1);
  // the bias is fetched now (at the epilogue its latency would be fully exposed) and rides in
  // the accumulator: every register of a lane belongs to the lane's output channel
  floatx16 acc[MF][NF];
#pragma unroll
  for (int fn = 0; fn < NF; ++fn) {
    const float b0 = g.bias[ncol + fn * 32];  // bias is padded to cout_pad
#pragma unroll
    for (int fm = 0; fm < MF; ++fm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fm][fn][r] = b0;
  }

  // LDS float4 offsets of this lane's A fragments inside a halo buffer: [k-group][m-frag]
  int afrag[G][MF];
#pragma unroll
  for (int gi = 0; gi < G; ++gi)
#pragma unroll
    for (int fm = 0; fm < MF; ++fm) afrag[gi][fm] = (2 * gi + kh) * QS + abase[fm];
  // float4 between consecutive taps' A fragments: one stencil row (k x k) or CG planes (1x1)
  const int rowstep = KS == 1 ? CG * QS : row_lds;

  // One tap = 8*G MFMAs with the loads for the NEXT tap threaded between them in a
  // fixed order: B straight from L2, A fragments from LDS (running row address +
  // immediate column offset), and - in the first rows of a chunk only (STAGE) - PPT
  // 16-byte pieces per thread of the next chunk's halo, parked in LDS one tap later.
  // A wave issues MFMAs back to back on its own and the VALU port stays nearly idle:
  // measured, every VALU instruction in this loop costs MFMA issue slots once two
  // waves share a SIMD (tools/exp_variants.sh drops one load stream at a time).
  // which B register (k-group) is fetched after MFMA pair n (-1 = none)
  // (RTPOSE_EXP_BSLOT / _B / _A / _STAGE: identity in production builds, see conv_exp.h)
  // memory clobber: loads/stores may not cross (IR + DAG); sched_barrier: nothing may
  // cross in the machine scheduler
#define RTPOSE_PIN()                 \
  asm volatile("" ::: "memory");     \
  __builtin_amdgcn_sched_barrier(0)
#define RTPOSE_CONV_STEP(ACUR, ANXT, BCUR, BLOAD, KX, STAGE)                                   \
  {                                                                                            \
    _Pragma("unroll") for (int n = 0; n < 4 * G; ++n) {                                        \
      const int gi_ = n >> 2, j_ = n & 3;                                                      \
      _Pragma("unroll") for (int fn = 0; fn < NF; ++fn) {                                      \
        const float bv_[4] = {BCUR[fn * G + gi_].x, BCUR[fn * G + gi_].y, BCUR[fn * G + gi_].z, \
                              BCUR[fn * G + gi_].w};                                           \
        _Pragma("unroll") for (int fm = 0; fm < MF; ++fm) {                                    \
          const float av_[4] = {ACUR[gi_][fm].x, ACUR[gi_][fm].y, ACUR[gi_][fm].z, ACUR[gi_][fm].w}; \
          acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_[j_], bv_[j_], acc[fm][fn], 0, 0, 0); \
        }                                                                                      \
      }                                                                                        \
      RTPOSE_PIN();                                                                            \
      if (RTPOSE_EXP_BSLOT(n) >= 0) {                                                          \
        const int g2 = RTPOSE_EXP_BSLOT(n);                                                    \
        bq[g2] += b_it_stride;                                                                 \
        BLOAD[g2] = (RTPOSE_EXP_HALF_B && g2 >= 1) ? BCUR[g2] : RTPOSE_EXP_B(gload4(bq[g2]), BCUR[g2]); \
      }                                                                                        \
      if (n == 1) {                                                                            \
        if (KS == 1 || (KX) == KS - 1) { /* next tap: next sub-chunk (1x1) / next stencil row */ \
          _Pragma("unroll") for (int g2 = 0; g2 < G; ++g2)                                     \
            _Pragma("unroll") for (int fm = 0; fm < MF; ++fm) arow[g2][fm] += rowstep;         \
        }                                                                                      \
        if (((STAGE) & RTPOSE_EXP_STAGE) != 0) {                                                     \
          _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                                    \
            smem4[hl[p]] = hv[p];                                                              \
            const int set = ps * PPT + p;                                                      \
            if (set < nsets) { /* uniform: sets past the halo are not fetched at all */        \
              hv[p] = gload4(next_base + piece_goff(set));                                     \
              hl[p] = (next_ch_ok && tid < np_total - set * 256) ? hn_off + piece_loff(set) : dummy_loff; \
            } else {                                                                           \
              hv[p] = make_float4(0.f, 0.f, 0.f, 0.f);                                         \
              hl[p] = dummy_loff;                                                              \
            }                                                                                  \
          }                                                                                    \
          ++ps;                                                                                \
        }                                                                                      \
      }                                                                                        \
      if (n >= 2 && n - 2 < G) {                                                               \
        _Pragma("unroll") for (int fm = 0; fm < MF; ++fm)                                      \
          ANXT[n - 2][fm] = RTPOSE_EXP_A(smem4[arow[n - 2][fm] + ((KS > 1 && (KX) + 1 < KS) ? (KX) + 1 : 0)], ACUR[n - 2][fm]); \
      }                                                                                        \
      RTPOSE_PIN();                                                                            \
    }                                                                                          \
  }
#define RTPOSE_CONV_ROW(STAGE)                                                  \
  {                                                                             \
    _Pragma("unroll") for (int kx = 0; kx < TAPS; ++kx) {                       \
      /* A sets alternate (kx & 1); B sets rotate (kx % 3): multiply s[kx%3],  */ \
      /* fill s[(kx+2)%3] with the tap two ahead                               */ \
      if (KS == 1 && kx >= nt) { /* short last buffer of a 1x1 conv */          \
      } else if (kx % 6 == 0) {                                                 \
        RTPOSE_CONV_STEP(a0, a1, s0, s2, kx, STAGE)                             \
      } else if (kx % 6 == 1) {                                                 \
        RTPOSE_CONV_STEP(a1, a0, s1, s0, kx, STAGE)                             \
      } else if (kx % 6 == 2) {                                                 \
        RTPOSE_CONV_STEP(a0, a1, s2, s1, kx, STAGE)                             \
      } else if (kx % 6 == 3) {                                                 \
        RTPOSE_CONV_STEP(a1, a0, s0, s2, kx, STAGE)                             \
      } else if (kx % 6 == 4) {                                                 \
        RTPOSE_CONV_STEP(a0, a1, s1, s0, kx, STAGE)                             \
      } else {                                                                  \
        RTPOSE_CONV_STEP(a1, a0, s2, s1, kx, STAGE)                             \
      }                                                                         \
    }                                                                           \
    /* re-normalise the register roles for the next row (a few v_mov per row) */ \
    if (TAPS & 1) {                                                             \
      _Pragma("unroll") for (int gi = 0; gi < G; ++gi)                          \
        _Pragma("unroll") for (int fm = 0; fm < MF; ++fm) a0[gi][fm] = a1[gi][fm]; \
    }                                                                           \
    _Pragma("unroll") for (int gi = 0; gi < GB; ++gi) {                         \
      if (TAPS % 3 == 1) {                                                      \
        const float4 t_ = s0[gi];                                               \
        s0[gi] = s1[gi];                                                        \
        s1[gi] = s2[gi];                                                        \
        s2[gi] = t_;                                                            \
      } else if (TAPS % 3 == 2) {                                               \
        const float4 t_ = s2[gi];                                               \
        s2[gi] = s1[gi];                                                        \
        s1[gi] = s0[gi];                                                        \
        s0[gi] = t_;                                                            \
      }                                                                         \
    }                                                                           \
  }

  // rows of a chunk whose taps carry the staging code: set s is fetched at tap s and
  // parked at tap s+1, so nsets+1 taps are needed (the host guarantees they exist)
  const int stage_rows = min(ROWS, (nsets + 1 + PPT * TAPS - 1) / (PPT * TAPS));
  float4 hv[PPT];
  int hl[PPT];
  for (int chunk = 0; chunk < nbig; ++chunk) {   // one LDS buffer (TB sub-chunks of CK channels) per turn
    const int nt = min(TB, nchunks - chunk * TB);  // taps that exist in this buffer (1x1 only: < TB at the end)
    (void)nt;
    if (NBUF == 1) {  // every wave is past the previous chunk (barrier at the loop end)
      fill_halo(in_base + (size_t)chunk * TB * CK, nt);
      __syncthreads();
    }
#ifdef RTPOSE_EXP_STAGGER
    // de-phase the four waves of the block after every barrier so that their B loads
    // do not hit the vector-memory path in the same cycles
    if (wave == 1) __builtin_amdgcn_s_sleep(RTPOSE_EXP_STAGGER);
    if (wave == 2) __builtin_amdgcn_s_sleep(2 * RTPOSE_EXP_STAGGER);
    if (wave == 3) __builtin_amdgcn_s_sleep(3 * RTPOSE_EXP_STAGGER);
#endif
    const int hb_off = NBUF == 2 ? (chunk & 1) * buf4 : 0;
    const int hn_off = NBUF == 2 ? ((chunk + 1) & 1) * buf4 : 0;
    // the last chunk re-stages itself into the idle buffer (never read): no branch
    const int chunk_next = min(chunk + 1, nbig - 1);
    const float* next_base = in_base + (size_t)chunk_next * TB * CK;
    const bool next_ch_ok = pj < min(TB, nchunks - chunk_next * TB) * CG;
    (void)next_ch_ok;
    int ps = 0;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      hv[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      hl[p] = dummy_loff;
    }
    // running LDS addresses (float4 units) of this lane's fragments on the current stencil row
    int arow[G][MF];
    float4 a0[G][MF], a1[G][MF];
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
      for (int fm = 0; fm < MF; ++fm) {
        arow[gi][fm] = hb_off + afrag[gi][fm];
        a0[gi][fm] = smem4[arow[gi][fm]];  // tap (0,0)
      }
    int ky = 0;
    if (NBUF == 2)
      for (; ky < stage_rows; ++ky) RTPOSE_CONV_ROW(1)
    for (; ky < ROWS; ++ky) RTPOSE_CONV_ROW(0)
    if (NBUF == 2) {  // park whatever is still in flight, then publish the buffer
#pragma unroll
      for (int p = 0; p < PPT; ++p) smem4[hl[p]] = hv[p];
    }
    __syncthreads();
  }
#undef RTPOSE_CONV_ROW
#undef RTPOSE_CONV_STEP
#undef RTPOSE_PIN
  RTPOSE_TSTAMP(2);

  // ---- epilogue: bias (+ReLU) (+2x2 max-pool), masked stores -----------------
#pragma unroll
  for (int fn = 0; fn < NF; ++fn) {
  const int ncolf = ncol + fn * 32;
  const bool col_ok = ncolf < g.cout;
  float* out_base = g.out + ((g.out_cmap && col_ok) ? g.out_cmap[ncolf] : g.out_choff + ncolf);
  if (!A.pool) {
    // strip mode: the lane's rows are m0 + wm*32*MF + 4*kh + {0,1,2,3, 8,9,10,11, 16,...}: pixel
    // coordinates by ONE division pair, then stepped (+1,+1,+1,+5 repeating).  An integer division
    // is ~40 VALU instructions; 32 of them per lane were longer than the whole tap loop of a
    // short-K 1x1 conv (ShuffleNetV2 pointwise: 8k MFMA cycles per block).
    int sn = 0, sy = 0, sx = 0;
    if (MODE == 0) {
      const int m = m0 + wm * (32 * MF) + 4 * kh;
      const int HW = A.H * A.W;
      sn = m / HW;
      const int r = m - sn * HW;
      sy = r / A.W;
      sx = r - sy * A.W;
    }
    auto step = [&](int d) {
      sx += d;
      while (sx >= A.W) {
        sx -= A.W;
        if (++sy >= A.H) {
          sy = 0;
          ++sn;
        }
      }
    };
    (void)step;
#pragma unroll
    for (int fm = 0; fm < MF; ++fm) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        // rows rg*8 + 4*kh + {0,1,2,3}
        const int ml0 = wm * (32 * MF) + fm * 32 + rg * 8 + 4 * kh;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int ml = ml0 + rr;
          int n, y, x;
          bool ok;
          if (MODE == 0) {
            ok = m0 + ml < A.M;
            n = sn;
            y = sy;
            x = sx;
            step(rr == 3 ? 5 : 1);  // (the walk restarts with every n-fragment)
          } else {
            int ty, tx;
            tile_local_yx(ml, A.tw_log2, ty, tx);
            n = n_img;
            y = y0 + ty;
            x = x0 + tx;
            ok = (y < A.H) && (x < A.W);
          }
          float v = acc[fm][fn][rg * 4 + rr];
          if (A.relu) v = fmaxf(v, 0.f);
          if (ok && col_ok) {
            const size_t q = (size_t)g.out_lead + (size_t)(n * g.out_hs + y) * g.out_ws + x;
            out_base[q * g.out_cstride] = v;
          }
        }
      }
    }
  } else {
    // MODE 1 only: quad = 4 consecutive m = 4 consecutive accumulator registers
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
        float v = fmaxf(fmaxf(acc[fm][fn][rg * 4 + 0], acc[fm][fn][rg * 4 + 1]),
                        fmaxf(acc[fm][fn][rg * 4 + 2], acc[fm][fn][rg * 4 + 3]));
        if (A.relu) v = fmaxf(v, 0.f);
        if (py < Ho && px < Wo && col_ok) {
          const size_t q = (size_t)g.out_lead + (size_t)(n_img * g.out_hs + py) * g.out_ws + px;
          out_base[q * g.out_cstride] = v;
        }
      }
    }
  }
  }  // fn
  RTPOSE_TSTAMP(3);
#ifdef RTPOSE_EXP_TIMELINE
  __builtin_amdgcn_s_waitcnt(0);
  RTPOSE_TSTAMP(4);
#endif
}

// Kernel: 1-D grid, block id -> (group, N tile, M tile).
//  * XCD-aware order (xcd_remap): the dispatcher places block b on XCD b % 8 (observed; used
//    for speed only).  Ids are laid out so that the (group, N tile) combinations of ONE M tile
//    are consecutive slots of ONE XCD: they run together and share that tile's input halo in
//    the XCD's L2 instead of each re-reading it from MALL/HBM (8 N tiles x 1.75 halo overlap
//    = 14 reads of every input pixel of conv4_2 before this).
//  * MODE 0 (strip) tail: ids [0, nbig) are 128-pixel tiles; the rest are pairs of 64-pixel
//    halves of the remaining tiles, dispatched last (see plan in conv2d_launch).
template <int KS, int CK, int MODE, int NBUF, int NF>
__global__ __launch_bounds__(256, NBUF == 1 ? 4 : 2) void conv_mfma_f32(const ConvArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = blockIdx.x;
  RTPOSE_TSTAMP(0);
  // Equal tiles keep the two co-resident blocks of a CU in lock step, so their prologues
  // (first halo fill, ~2 us of exposed latency) and epilogues coincide instead of hiding
  // under each other's MFMAs.  Half of the first wave of blocks starts half a tile late.
  if (A.dephase_mode && L < 2 * A.n_cu &&
      (A.dephase_mode == 1 ? L >= A.n_cu : (L & 1))) {
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < A.dephase_cycles) __builtin_amdgcn_s_sleep(32);
  }
  const bool small = MODE == 0 && L >= A.nbig;
  const int bi = small ? A.nbig + ((L - A.nbig) >> 1) : L;
  int mt, c;
  if (A.xcd_remap) {
    const int xcd = bi & 7, j = bi >> 3;
    c = j % A.ncombo;
    mt = (j / A.ncombo) * 8 + xcd;
  } else {
    mt = bi % A.mtiles;
    c = bi / A.mtiles;
  }
  if (mt >= A.mtiles) return;  // padding ids of the remapped order
  const int nt = c % A.ntiles, grp = c / A.ntiles;
  if (MODE == 1) {
    conv_tile<KS, CK, MODE, NBUF, 2, NF>(A, A.g[grp], mt, nt, smem);
  } else if (!small) {
    conv_tile<KS, CK, MODE, NBUF, 2, NF>(A, A.g[grp], mt * kBM, nt, smem);
  } else {
    const int m0 = mt * kBM + ((L - A.nbig) & 1) * (kBM / 2);
    if (m0 < A.M) conv_tile<KS, CK, MODE, NBUF, 1, NF>(A, A.g[grp], m0, nt, smem);
  }
}

// ---- weight packing -------------------------------------------------------------
// packed[chunk][tap][cg][cout_pad][4]  <-  w[cout][cin_src][k][k]
__global__ void pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                    int cout, int cin_src, int k, const int32_t* __restrict__ cin_map,
                                    int cin_packed, int ck, int coutp, float* __restrict__ wp,
                                    float* __restrict__ bp) {
  const int T = k * k;
  const size_t total = (size_t)T * cin_packed * coutp;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)coutp) bp[i] = (i < (size_t)cout && bias) ? bias[i] : 0.f;
  if (i >= total) return;
  const int e = i & 3;
  size_t r = i >> 2;
  const int n = r % coutp;
  r /= coutp;
  const int cg = r % (ck / 4);
  r /= (ck / 4);
  const int tap = r % T;
  const int chunk = r / T;
  const int c = chunk * ck + cg * 4 + e;
  int src = cin_map ? cin_map[c] : (c < cin_src ? c : -1);
  float v = 0.f;
  if (n < cout && src >= 0 && src < cin_src) {
    const int ky = tap / k, kx = tap - ky * k;
    v = w[(((size_t)n * cin_src + src) * k + ky) * k + kx];
  }
  wp[i] = v;
}

// ---- host side --------------------------------------------------------------------
static int conv_ck(int cin) { return (cin % 16 == 0) ? 16 : 8; }

struct ConvPlan {
  int mode, ck, qs, hw_lds, tw_log2, tiles_x, tiles_y, grid_x, nbuf, nf;
  size_t lds_bytes;
};
static int g_force_nbuf = 0;  // developer override (RTPOSE_CONV_NBUF=1|2)

// LDS plane size: pixel count rounded so that the 4 channel-group planes of one
// pixel land in different 16-byte bank slots on the staging writes.
static int round_qs(int npix) {
  int qs = npix;
  while ((qs & 3) != 2) ++qs;
  return qs;
}

static int halo_row_lds(int tw, int p) {
  int w = tw + 2 * p;
  while ((w & 15) != 8) ++w;  // consecutive tile rows half a bank-row apart
  return w;
}

static int plan_conv(const rtpose_conv_desc& d, int N, int H, int W, ConvPlan* pl) {
  const int P = d.k / 2;
  pl->ck = conv_ck(d.cin);
  const int M = N * H * W;
  // piece sets per thread that fit the staging schedule (set s fetched at tap s, parked at s+1)
  const int max_pieces = (d.k == 1) ? PiecesPerTap<1>::value : d.k * d.k - 1;
  const int cg = (pl->ck / 4) * (d.k == 1 ? kTB1x1 : 1);  // channel-group planes per LDS buffer
  if (!g_force_nbuf) {
    const char* e = dev_env("RTPOSE_CONV_NBUF");
    g_force_nbuf = e ? atoi(e) : -1;
  }
  // 1x1 layers: single halo buffer, 4 blocks per CU (occupancy hides the refill latency of
  // these short-K GEMMs better than a second buffer: 42.7 vs 32.6 TF/s); k x k: double buffer
  pl->nbuf = (g_force_nbuf == 1 || g_force_nbuf == 2) ? g_force_nbuf : (d.k == 1 ? 1 : 2);
  // strips waste no MFMA work on tile edges but stage a longer halo than 2-D tiles on wide maps
  // (3x3 at W = 92: 2.4x the tile vs 1.4x, against 8.9 % edge waste); widest map that still strips:
  static int strip_maxw = 0;
  if (!strip_maxw) {
    const char* e = dev_env("RTPOSE_CONV_STRIP_MAXW");
    strip_maxw = e ? atoi(e) : 128;  // measured: conv3_1..3 (92 x 92) 6.70 -> 6.33 ms as strips; 184 x 184: no change
  }
  bool strip = (W <= strip_maxw) && !d.pool;
  if (strip) {
    const rtpose_layout& l = d.lin;
    const int lb = (kBM - 1) + ((kBM - 1) / W + 1) * (l.ws - W) +
                   ((kBM - 1) / (H * W) + 1) * (l.hs - H) * l.ws + 2 * P * l.ws + 2 * P + 1;
    const int qs = round_qs(lb);
    const size_t lds = (size_t)pl->nbuf * cg * qs * 16 + 256 * 16;  // + one dummy park slot per thread
    if ((pl->nbuf == 2 && ceil_div(qs * cg, 256) > max_pieces) || lds > 80 * 1024) strip = false;
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
  // 2-D tiles: pick the tile width with the least padded work
  int best_tw = 0;
  long best_cost = -1;
  for (int twl = 2; twl <= 6; ++twl) {
    const int tw = 1 << twl, th = kBM >> twl;
    if (th < 2) continue;
    const long cost = (long)ceil_div(W, tw) * ceil_div(H, th);
    const long halo = (long)(th + 2 * P) * halo_row_lds(tw, P);
    // tie-break on the smaller halo
    const long key = cost * 100000 + halo;
    if (best_cost < 0 || key < best_cost) {
      best_cost = key;
      best_tw = twl;
    }
  }
  const int tw = 1 << best_tw, th = kBM >> best_tw;
  pl->mode = 1;
  pl->tw_log2 = best_tw;
  pl->hw_lds = halo_row_lds(tw, P);
  const int npix = (th + 2 * P) * pl->hw_lds;
  pl->qs = round_qs(npix);
  pl->tiles_x = ceil_div(W, tw);
  pl->tiles_y = ceil_div(H, th);
  pl->grid_x = N * pl->tiles_x * pl->tiles_y;
  pl->lds_bytes = (size_t)pl->nbuf * cg * pl->qs * 16 + 256 * 16;
  if (pl->nbuf == 2 && ceil_div(pl->qs * cg, 256) > max_pieces)
    return fail(RTPOSE_E_INVAL, "conv halo too large for the staging schedule");
  return 0;
}

template <int KS, int CK, int MODE, int NBUF, int NF>
static int launch_inst(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static PerDeviceOnce attr_set;  // zero-initialised; the attribute is per device
  const int dev = current_device();
  auto kern = conv_mfma_f32<KS, CK, MODE, NBUF, NF>;
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set.set(dev);
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int conv2d_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, hipStream_t s) {
  if (!d || ngroups < 1 || ngroups > 2) return fail(RTPOSE_E_INVAL, "conv2d: ngroups must be 1 or 2");
  RTPOSE_REFUSE_PLANES(d, ngroups, "conv2d");
  const rtpose_conv_desc& d0 = d[0];
  if (d0.k != 1 && d0.k != 3 && d0.k != 7) return fail(RTPOSE_E_INVAL, "conv2d: k must be 1, 3 or 7");
  if (d0.cin % 8 != 0 || d0.cin <= 0) return fail(RTPOSE_E_INVAL, "conv2d: cin must be a multiple of 8");
  if (N <= 0 || H <= 0 || W <= 0) return fail(RTPOSE_E_INVAL, "conv2d: empty tensor");
  const int P = d0.k / 2;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < ngroups; ++i) {
    const rtpose_conv_desc& di = d[i];
    if (di.k != d0.k || di.cin != d0.cin || di.relu != d0.relu || di.pool != d0.pool ||
        cout_pad(di.cout) != cout_pad(d0.cout) || di.lin.ws != d0.lin.ws || di.lin.hs != d0.lin.hs)
      return fail(RTPOSE_E_INVAL, "conv2d: grouped convs must share geometry");
    if (di.lin.ws < W + P || di.lin.hs < H + P || di.lin.lead < P * di.lin.ws + P)
      return fail(RTPOSE_E_INVAL, "conv2d: input layout gap smaller than the conv padding");
    if ((di.lin.cstride % 4) || (di.lin.choff % 4))
      return fail(RTPOSE_E_INVAL, "conv2d: input slice must be 16-byte aligned");
    if (di.lin.choff + di.cin > di.lin.cstride)
      return fail(RTPOSE_E_INVAL, "conv2d: input slice exceeds cstride");
    ConvGroup& g = a.g[i];
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
    g.out_cmap = di.out_cmap;
  }
  if (d0.pool && ((H | W) & 1)) return fail(RTPOSE_E_INVAL, "conv2d: fused pool needs even H and W");
  ConvPlan pl;
  int rc = plan_conv(d0, N, H, W, &pl);
  if (rc) return rc;
  a.N = N;
  a.H = H;
  a.W = W;
  a.M = N * H * W;
  a.cin = d0.cin;
  a.relu = d0.relu;
  a.pool = d0.pool;
  a.qs = pl.qs;
  a.hw_lds = pl.hw_lds;
  a.tw_log2 = pl.tw_log2;
  a.tiles_x = pl.tiles_x;
  a.tiles_y = pl.tiles_y;
  // ---- 1-D grid: id order, XCD-aware remap, half-tile tail ------------------------------
  const int n_cu = device_cu_count();  // of the device this launch goes to
  static int xcd_remap_env = -1;
  if (xcd_remap_env < 0) {
    const char* e = dev_env("RTPOSE_CONV_XCD_REMAP");
    xcd_remap_env = e ? atoi(e) : 1;
  }
  a.mtiles = pl.grid_x;
  // 128-wide N tiles (wave tile 64 x 64, NF = 2) are available behind RTPOSE_CONV_NF=2 but
  // NOT the default: measured on the 32 x 368 x 368 workload they help the 8-N-tile 3x3
  // layers by ~3 % (conv4_1/4_2) and lose 11 % on the 7x7 stage convs (fewer, longer blocks:
  // worse tail, and 32 MFMAs between filler slots) - 427 vs 464 img/s overall.
  static int nf_env = 0;
  if (!nf_env) {
    const char* e = dev_env("RTPOSE_CONV_NF");
    nf_env = e ? atoi(e) : 1;
  }
  pl.nf = (nf_env == 2 && d0.k != 1 && pl.ck == 16 && pl.nbuf == 2 && cout_pad(d0.cout) % 128 == 0) ? 2 : 1;
  {
    // 1x1 layers with >= 128 padded output channels: 128-wide N tiles halve the number of blocks
    // that each re-stage the same input tile (developer A/B: RTPOSE_CONV_NF1X1=1|2)
    static int nf1 = 0;
    if (!nf1) {
      const char* e = dev_env("RTPOSE_CONV_NF1X1");
      nf1 = e ? atoi(e) : 1;
    }
    if (nf1 == 2 && d0.k == 1 && pl.ck == 16 && pl.nbuf == 1 && cout_pad(d0.cout) % 128 == 0) pl.nf = 2;
  }
  a.ntiles = cout_pad(d0.cout) / (kConvBN * pl.nf);
  a.ncombo = a.ntiles * ngroups;
  a.xcd_remap = (xcd_remap_env != 0 && a.ncombo > 1 && a.mtiles >= 64) ? 1 : 0;
  const long ids = a.xcd_remap ? (long)8 * a.ncombo * ceil_div(a.mtiles, 8) : (long)a.mtiles * a.ncombo;
  if (ids > 0x7fffffffL) return fail(RTPOSE_E_INVAL, "conv2d: grid too large");
  a.nbig = (int)ids;
  if (pl.mode == 0) {
    // Tail quantisation: equal tiles on `slots` co-resident block slots run in lock-step
    // rounds; a last round with few blocks leaves most CUs idle for a whole tile time.  When
    // that remainder is small, its tiles are split into two 64-pixel halves (twice the blocks,
    // half the time).  32x46x46, cout 128 x 2 branches: 2116 tiles = 4 x 512 + 68 -> the last
    // 68 become 136 halves.
    const int slots = n_cu * (pl.nbuf == 1 ? 4 : 2);
    const int total = (int)ids;
    const int rem = total % slots;
    const char* e = dev_env("RTPOSE_CONV_NO_HALF_TILES");
    if (total > slots && rem > 0 && 2 * rem <= n_cu && !(e && e[0] == '1')) a.nbig = total - rem;
    // small batches (e.g. the reference's own one-image-at-a-time flow): fewer tiles than
    // CUs -> every tile is split, doubling the number of busy CUs
    if (total <= n_cu && !(e && e[0] == '1')) a.nbig = 0;
  }
  dim3 grid((unsigned)(a.nbig + 2 * (ids - a.nbig)), 1, 1);
#ifdef RTPOSE_EXP_TIMELINE
  {  // developer build: stamps of the LAST launch with kernel size RTPOSE_TIMELINE_K (default 1)
    extern unsigned long long* g_dbg32_buf;
    extern unsigned g_dbg32_blocks;
    static int kk = 0;
    if (!kk) {
      const char* e = dev_env("RTPOSE_TIMELINE_K");
      kk = e ? atoi(e) : 1;
    }
    if (!g_dbg32_buf) (void)hipMalloc(&g_dbg32_buf, (size_t)32768 * 8 * 8);
    if (d0.k == kk && grid.x <= 32768) {
      (void)hipMemsetAsync(g_dbg32_buf, 0, (size_t)grid.x * 64, s);
      a.dbg = g_dbg32_buf;
      g_dbg32_blocks = grid.x;
    }
  }
#endif
  {
    static int dephase_env = -1;
    if (dephase_env < 0) {
      const char* e = dev_env("RTPOSE_CONV_DEPHASE");
      dephase_env = e ? atoi(e) : 0;
    }
    a.dephase_mode = (pl.nbuf == 2 && (long)grid.x > 4L * n_cu) ? dephase_env : 0;
    a.n_cu = n_cu;
    // half of a co-resident pair's tile time: taps x 16 MFMAs x 64 cycles x 2 blocks / 2
    const long taps = (long)(d0.cin / pl.ck) * d0.k * d0.k;
    a.dephase_cycles = (int)(taps * 16 * 64);
  }
#define RTPOSE_CONV_CASE(KS_, CK_, MODE_)                                  \
  if (d0.k == KS_ && pl.ck == CK_ && pl.mode == MODE_) {                   \
    if (pl.nbuf == 1 && pl.nf == 2 && KS_ == 1 && CK_ == 16)                \
      return launch_inst<KS_, CK_, MODE_, 1, (KS_ == 1 && CK_ == 16) ? 2 : 1>(a, grid, pl.lds_bytes, s); \
    if (pl.nbuf == 1) return launch_inst<KS_, CK_, MODE_, 1, 1>(a, grid, pl.lds_bytes, s); \
    if (pl.nf == 2 && KS_ != 1 && CK_ == 16)                               \
      return launch_inst<KS_, CK_, MODE_, 2, (KS_ != 1 && CK_ == 16) ? 2 : 1>(a, grid, pl.lds_bytes, s); \
    return launch_inst<KS_, CK_, MODE_, 2, 1>(a, grid, pl.lds_bytes, s);   \
  }
  RTPOSE_CONV_CASE(3, 8, 0)
  RTPOSE_CONV_CASE(3, 8, 1)
  RTPOSE_CONV_CASE(3, 16, 0)
  RTPOSE_CONV_CASE(3, 16, 1)
  RTPOSE_CONV_CASE(7, 16, 0)
  RTPOSE_CONV_CASE(7, 16, 1)
  RTPOSE_CONV_CASE(1, 16, 0)
  RTPOSE_CONV_CASE(1, 16, 1)
  RTPOSE_CONV_CASE(1, 8, 0)
  RTPOSE_CONV_CASE(1, 8, 1)
  RTPOSE_CONV_CASE(7, 8, 0)
  RTPOSE_CONV_CASE(7, 8, 1)
#undef RTPOSE_CONV_CASE
  return fail(RTPOSE_E_INVAL, "conv2d: no kernel instance for k=%d ck=%d mode=%d", d0.k, pl.ck, pl.mode);
}

#ifdef RTPOSE_EXP_TIMELINE
unsigned long long* g_dbg32_buf = nullptr;
unsigned g_dbg32_blocks = 0;
#endif

int pack_weights_launch(const float* w, const float* bias, int cout, int cin_src, int k,
                        const int32_t* cin_map, int cin_packed, float* wp, float* bp, hipStream_t s) {
  if (cin_packed % 8 || (cin_packed < cin_src && !cin_map))
    return fail(RTPOSE_E_INVAL, "pack: cin_packed must be a multiple of 8 and >= cin_src");
  if (k != 1 && k != 3 && k != 7) return fail(RTPOSE_E_INVAL, "pack: k must be 1, 3 or 7");
  const int coutp = cout_pad(cout);
  const size_t total = (size_t)k * k * cin_packed * coutp;
  const int threads = 256;
  const unsigned blocks = (unsigned)((total + threads - 1) / threads);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(threads), 0, s, w, bias, cout, cin_src, k,
                     cin_map, cin_packed, conv_ck(cin_packed), coutp, wp, bp);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace rtpose

extern "C" {

size_t rtpose_packed_weight_floats(int cout, int cin, int k) {
  const int cinp = rtpose::ceil_div(cin, 8) * 8;
  // + two (chunk, tap) blocks of slack: the kernel's B prefetch runs two taps ahead
  return (size_t)(k * k * cinp + 32) * rtpose::cout_pad(cout);
}
size_t rtpose_packed_bias_floats(int cout) { return (size_t)rtpose::cout_pad(cout); }

int rtpose_pack_conv_weights(const float* w_oihw, const float* bias, int cout, int cin_src, int k,
                             const int32_t* cin_map, int cin_packed, float* w_packed,
                             float* bias_packed, void* stream) {
  return rtpose::pack_weights_launch(w_oihw, bias, cout, cin_src, k, cin_map, cin_packed, w_packed,
                                     bias_packed, rtpose::as_stream(stream));
}

int rtpose_conv2d(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, void* stream) {
  return rtpose::conv2d_launch(d, ngroups, N, H, W, rtpose::as_stream(stream));
}

}  // extern "C"

#ifdef RTPOSE_EXP_TIMELINE
extern "C" int rtpose_debug_timeline32_dump(unsigned long long* host, unsigned cap_blocks) {
  using namespace rtpose;
  if (!g_dbg32_buf) return 0;
  (void)hipDeviceSynchronize();
  const unsigned n = g_dbg32_blocks < cap_blocks ? g_dbg32_blocks : cap_blocks;
  (void)hipMemcpy(host, g_dbg32_buf, (size_t)n * 64, hipMemcpyDeviceToHost);
  return (int)n;
}
#endif
