// fp32 Winograd F(2x2, 3x3) convolution for gfx950 (MI355X): stride 1, "same" padding, fused bias
// (+ReLU) (+2x2 max-pool).
//
// Stands in for the 3x3 nn.Conv2d / nn.ReLU / nn.MaxPool2d modules of the VGG-19 front end and the
// stage-1 branches (lib/network/rtpose_vgg.py:23-35, :49-55, :108-127) wherever the direct
// implicit-GEMM kernel (conv_mfma.hip) is MFMA-bound: 16 multiplies per 2 x 2 output tile and input
// channel instead of 36, i.e. 2.25x fewer matrix-core flops for results that differ from the direct
// sum by a few ulp (Lavin & Gray's minimal filtering F(2x2, 3x3); the north-star bound is 1e-3).
//
//   Y = A^T [ (G g G^T) o (B^T d B) ] A        per 4 x 4 input patch d, 3 x 3 filter g, summed over cin
//
// MI355X shape (not a translation of any CUDA kernel):
//  * "wtile" = one 2 x 2 output tile.  The 16 frequencies are 16 independent GEMMs
//    [wtiles x cin] . [cin x cout]; a wave owns ALL 16 frequencies of a 32-wtile x 32-column tile
//    = 16 v_mfma_f32_32x32x2_f32 accumulators = 256 registers (the AGPR half of the 512 a wave gets at
//    one wave per SIMD), so the output transform A^T M A is lane-local: no cross-wave exchange, and
//    the fused 2 x 2 max-pool is a max over the 4 outputs a lane just produced.  The bias rides in
//    the accumulator of frequency (1,1) (A^T e11 A = all-ones).
//  * A block = 4 waves = (32 WM) consecutive wtiles of the flattened (n, ty, tx) order x (32 WN) output
//    columns; no 2-D tile waste on the 46 x 46 maps (23 x 23 wtiles).
//  * Input transform: every thread loads HALF a 4 x 4 patch of 4 channels (3 rows x 4 pixels, 16-byte raw
//    buffer loads straight from the shared-gap NHWC layout: the zero gap is the conv padding), forms 8
//    frequencies in two groups of 16 packed instructions (fp32 VALU work is paid for in matrix throughput,
//    DESIGN.md §3.0) and writes them to LDS as V[f][c/4][wtile][4], wtile slots rotated per channel group against
//    bank conflicts - the layout the A operand is read from with one ds_read_b128 per 4 MFMAs.  Double-buffered
//    per channel chunk: the patch of chunk c+2 is in flight and chunk c+1 is transformed while chunk c is
//    multiplied; one barrier per chunk.
//  * B operand = transformed weights U[chunk][f][c/4][cout][4], packed once
//    (rtpose_pack_conv_weights_winograd), straight from L2 to registers two steps (4 frequencies) ahead.
//  * With more tiles than CUs the blocks are persistent: one (n tile, group) per block, every
//    (256 / ncombo)-th m tile in turn, the next tile's first patches requested before the output transform.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "common.h"
#include "conv_exp.h"
#include "wino_common.h"

namespace rtpose {

namespace wino {

using namespace winoc;

struct Group {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  int in_cstride, in_choff, in_ws, in_hs, in_lead;
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  int cout, cout_pad;
  size_t in_bytes, w_bytes, out_bytes;  // extents of the three allocations from in / w / out (hardware bounds clamp)
};

struct Args {
  Group g[2];
  int N, H, W;
  int TY, TX, T;  // wtiles per column / row of an image, and in the whole batch
  int cin;
  int relu, pool;
  int mtiles, ntiles, ncombo, xcd_remap;
  unsigned long long* dbg;  // RTPOSE_EXP_TIMELINE builds only: 6 tiles x 8 u64 stamps per block
  int persist;  // 1: gridDim.x persistent blocks; block p keeps (n tile, group) p % ncombo and walks every (gridDim.x / ncombo)-th m tile
};

template <int WM, int WN, int CK>
__global__ __launch_bounds__(256, 1) void wino_f32(const Args A) {
  constexpr int NT = 32 * WM;  // wtiles per block
  constexpr int CG = CK / 4;   // 16-byte channel groups per chunk
  constexpr int G = CK / 8;    // 8-deep k groups per chunk (4 MFMAs each)
  static_assert(WM * WN == 4, "4 waves");
  constexpr int IPT = NT * CG * 2 / 256;  // half patches per thread and chunk: 1, or 2 = one whole 4 x 4 patch
  static_assert(NT * CG * 2 == 256 * IPT && (IPT == 1 || IPT == 2), "half patches per thread");
  constexpr int VBUF = 16 * CG * NT;  // float4 per V buffer
  extern __shared__ __attribute__((aligned(16))) float4 V4[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int l31 = lane & 31, kh = lane >> 5;

  // ---- block -> (n tile, group) and its m tiles ------------------------------------------------
  // one tile per block (XCD-aware order as in conv_mfma.hip), or persistent: block p keeps ONE (n tile, group) and
  // walks the m tiles j0, j0 + jstep, ...  The first patches of the next tile are requested right after the last
  // barrier of the current one, so their HBM latency (and the block dispatch) hides under the output transform.
  const int bi = blockIdx.x;
  int j0, jstep, c;
  if (A.persist) {
    c = bi % A.ncombo;
    j0 = bi / A.ncombo;
    jstep = gridDim.x / A.ncombo;
  } else {
    if (A.xcd_remap) {
      const int xcd = bi & 7, j = bi >> 3;
      c = j % A.ncombo;
      j0 = (j / A.ncombo) * 8 + xcd;
    } else {
      j0 = bi % A.mtiles;
      c = bi / A.mtiles;
    }
    jstep = A.mtiles;  // one tile
  }
  if (j0 >= A.mtiles) return;
  const int nt = c % A.ntiles, grp = c / A.ntiles;
  const Group g = grp ? A.g[1] : A.g[0];
  const int TT = A.TY * A.TX;

  // ---- input transform role: (channel group, wtile, upper / lower half of the patch) ---------
  // B^T over the patch rows d0..d3:  fy 0: d0 - d2,  1: d1 + d2,  2: d2 - d1,  3: d1 - d3.
  // A thread loads three rows (R0, R1, R2) and forms  ta = R0 - R2,  tb = R2 + sgn R1:
  //   upper half (waves 0,1): (d0, d1, d2), sgn +1 -> fy 0, 1;   lower half (waves 2,3): (d2, d3, d1), sgn -1 -> fy 2, 3
  // - the same instruction stream for both halves, no branch inside the pinned MFMA loop.
  const int cg = tid % CG, tl = (tid / CG) % NT;
  const int half = IPT == 2 ? 0 : __builtin_amdgcn_readfirstlane(tid / (CG * NT));  // (IPT 2: both halves, natural rows)
  const float sgn = half ? -1.f : 1.f;
  // every global operand comes through a raw buffer load: per-lane byte offset in ONE register for the whole
  // kernel, everything that moves (chunk, patch row / column, frequency) in the scalar offset
  // (the input descriptor is based at the block's first patch, so the 32-bit offsets stay small whatever the
  // size of the activation buffer)
  const i32x4 rw = make_rsrc(g.w, g.w_bytes);
  // two patch loaders: the tile being multiplied and the block's NEXT tile, whose first two chunks are fetched (and
  // the first one transformed) under the last two chunks of this one - see the tile loop
  i32x4 rin, rin_nx;
  unsigned pvoff, pvoff_nx;
  auto set_loader = [&](int mt, i32x4& r_, unsigned& v_) {
    auto patch_q = [&](int t) -> size_t {
      const int n = t / TT, r = t - n * TT;
      const int ty = r / A.TX, tx = r - ty * A.TX;
      return (size_t)g.in_lead + (size_t)(n * g.in_hs + 2 * ty - 1) * g.in_ws + (2 * tx - 1);
    };
    const size_t q0 = patch_q(min(mt * NT, A.T - 1));                // uniform: lowest address of the tile
    const size_t q = patch_q(min(mt * NT + tl, A.T - 1));            // wtiles past the end re-read the last one
    const size_t o0 = q0 * g.in_cstride + g.in_choff;
    r_ = make_rsrc(g.in + o0, g.in_bytes - o0 * 4);
    v_ = (unsigned)(((q - q0) * g.in_cstride + cg * 4) * 4);
  };
  set_loader(j0, rin, pvoff);
  constexpr int NROW = IPT == 2 ? 4 : 3, NPC = 4 * NROW;  // patch rows / 16-byte pieces a thread loads per chunk
  unsigned psoff[NROW];  // uniform byte offsets of the patch rows (of this half)
#pragma unroll
  for (int r = 0; r < NROW; ++r) {
    const int row = half ? (r == 0 ? 2 : r == 1 ? 3 : 1) : r;
    psoff[r] = (unsigned)(row * g.in_ws * g.in_cstride * 4);
  }
  const unsigned pxb = (unsigned)g.in_cstride * 4;  // bytes per pixel
  F4 p[NROW][4], ta[4], tb[4];
  auto load_piece_from = [&](const i32x4& r_, unsigned v_, int chunk, int i) {
    p[i >> 2][i & 3] = bload(r_, v_, (unsigned)chunk * (CK * 4) + psoff[i >> 2] + (i & 3) * pxb);
  };
  auto load_piece = [&](int chunk, int i) { load_piece_from(rin, pvoff, chunk, i); };
  // V[f][cg][wtile], f = fy * 4 + fx; this thread writes fy in {2 half, 2 half + 1}; B^T over x as over the rows
  // (wtile slots are rotated by SW = 8 / CG per channel group.  A ds_write_b128 is serviced in groups of 8 CONTIGUOUS
  //  lanes on 32 banks = 8 16-byte slots: the 8 lanes of a group - 8 / CG wtiles x CG groups - must land in 8 distinct
  //  slots modulo 8.  Unrotated, the CG planes of a wtile are 512 B apart = the same banks (SQ_LDS_BANK_CONFLICT was
  //  7 % of the kernel's cycles); round 2's rotation by 16 / CG was derived for 16-lane groups and left every write
  //  2-way conflicted: 26 % (this instance) / 36 % (conv1_2) of SQ_LDS_IDX_ACTIVE.  The A reads - ds_read_b128, groups
  //  of 16 lanes that are complete residue systems modulo 16 - are conflict-free under any rotation.)
  constexpr int SW = RTPOSE_EXP_W3_SW(CG);
  const int vst = (half * 8 * CG + cg) * NT + ((tl + SW * cg) % NT);
  // transform in 2 IPT groups of 16 packed VALU instructions (few, full groups: see wino_common.h): rows, then
  // columns + LDS writes; with a whole patch (IPT 2) first for fy 0, 1 and then for fy 2, 3
  auto tgroup = [&](int buf, int gidx) {
    const int hh = gidx >> 1;  // which half of the frequencies (IPT 2 only)
    if ((gidx & 1) == 0) {
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if (IPT == 2 && hh) {  // fy 2: d2 - d1, fy 3: d1 - d3
          ta[x] = sub4(p[2][x], p[1][x]);
          tb[x] = sub4(p[1][x], p[NROW - 1][x]);
        } else {               // fy 0: d0 - d2, fy 1: d1 + d2   (half patches: the permuted rows and sgn do both)
          ta[x] = sub4(p[0][x], p[2][x]);
          tb[x] = fma4(sgn, p[1][x], p[2][x]);
        }
      }
    } else {
      float4* v = V4 + buf * VBUF + vst + hh * 8 * CG * NT;
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const F4* t = (o & 4) ? tb : ta;
        const int fx = o & 3;
        v[o * CG * NT] = to_float4(fx == 0 ? sub4(t[0], t[2]) : fx == 1 ? add4(t[1], t[2]) : fx == 2 ? sub4(t[2], t[1])
                                                                                          : sub4(t[1], t[3]));
      }
    }
  };

  // ---- MFMA roles -----------------------------------------------------------------------------
  const int ncol = nt * (32 * WN) + wn * 32 + l31;
  const int arow = wm * 32 + l31;
  floatx16 acc[16];
  const float bias0 = g.bias[ncol];  // padded to cout_pad
  // B: lane offset (k half, column) in one register; (frequency of the pair, k group) and the step in the scalar offset
  const unsigned boff = (unsigned)((kh * g.cout_pad + ncol) * 16);
  const unsigned cgstep = (unsigned)(g.cout_pad * 16);     // bytes per channel group plane
  const unsigned bstep = (unsigned)(2 * CG) * cgstep;      // bytes per step (2 frequencies)
  unsigned wso = 0;                                        // uniform byte offset of the next step to fetch
  float4 bs[4][2][G];  // [step % 4][frequency of the pair][k group]

  const int nchunks = A.cin / CK;  // >= 2 (host)
  // (no __builtin_assume(nchunks >= 2) / do-while here: either removes the compiler's second copy of the accumulator
  //  initialisation on the "loop not entered" path - and with it the register allocation that keeps the chunk loop
  //  free of spills: 70-105 VGPRs went to scratch INSIDE the loop)
  // ---- the chunk pipeline runs ACROSS the tiles of a persistent block -------------------------------------------
  // Steady state of chunk c: multiply chunk c from V[par], transform chunk c + 1 (in the patch registers) into
  // V[par ^ 1], fetch chunk c + 2 into the patch registers.  At the end of a tile "chunk c + 1 / c + 2" are the first
  // chunks of the block's NEXT tile: when a tile's multiply loop ends, the next tile's chunk 0 sits transformed in
  // LDS and its chunk 1 is in flight - a tile starts multiplying right after its B fragments arrive, with no
  // exposed patch latency, transform or barrier.  (Round 2 fetched the next chunk 0 under the output transform,
  // transformed it at the tile start and then waited a full memory latency for chunk 1: 8-11 us per tile that the one
  // block per CU could not hide - 45 % of conv1_2, whose tiles multiply for 13.6 us.)
  // (the 8-channel-chunk instance - cin = 8 x odd at 64 columns, not a layer of rtpose_vgg - keeps a per-tile B ring:
  //  with the wrap the compiler peels its short chunk loop and spills 0.5 KB per lane)
  constexpr bool BRING = CK == 16;
  int par = 0;  // V buffer of the chunk being multiplied
#pragma unroll
  for (int i = 0; i < NPC; ++i) load_piece(0, i);  // chunk 0 of the first tile
#pragma unroll
  for (int gq = 0; gq < 2 * IPT; ++gq) tgroup(0, gq);
#pragma unroll
  for (int i = 0; i < NPC; ++i) load_piece(1, i);
  // B fragments of steps 0, 1 (after the patch loads, as in the steady state of the loop: see conv_wino7.hip).  The
  // B ring runs across tiles as well: the last chunk of a tile prefetches steps 0, 1 of the next tile's chunk 0
  // (wso wraps to 0 there), so that the first 32 MFMAs of a tile need nothing from memory and the stores of the
  // previous tile's output transform - older vector-memory operations that every later vmcnt wait has to see
  // retired first - drain underneath them.
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
    for (int fs = 0; fs < 2; ++fs)
#pragma unroll
      for (int gi = 0; gi < G; ++gi) bs[s2][fs][gi] = bload_f4(rw, boff, wso + (fs * CG + 2 * gi) * cgstep);
    wso += bstep;
  }
  __syncthreads();

  [[maybe_unused]] int ti = 0;  // tile counter of this block (timeline builds)
  for (int mt = j0; mt < A.mtiles; mt += jstep) {
  RTPOSE_TSTAMP3(ti, 0);
  const bool has_next = mt + jstep < A.mtiles;
  // (without a next tile the "next" loader re-reads this tile's first chunks: valid addresses, results never used)
  if (has_next) {
    set_loader(mt + jstep, rin_nx, pvoff_nx);
  } else {
    rin_nx = rin;
    pvoff_nx = pvoff;
  }
#pragma unroll
  for (int f = 0; f < 16; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = f == 5 ? bias0 : 0.f;
  if (!BRING && mt != j0) {  // (8-channel chunks: the B ring restarts with every tile)
    wso = 0;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int fs = 0; fs < 2; ++fs)
#pragma unroll
        for (int gi = 0; gi < G; ++gi) bs[s2][fs][gi] = bload_f4(rw, boff, wso + (fs * CG + 2 * gi) * cgstep);
      wso += bstep;
    }
  }
  RTPOSE_TSTAMP3(ti, 1);

  // One step = the two frequencies 2s, 2s+1 = 8 G MFMAs on two alternating accumulators.  Between MFMA
  // pairs, in a fixed (pinned) order: the A fragments of the next step (LDS), the B fragments two steps
  // ahead (L2); the last slot of step 0 / 1 also carries the two halves of the input transform of the NEXT
  // chunk, the last slots of steps 2..7 two patch loads each of the chunk after that.
#define RTPOSE_PIN()             \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)
  constexpr int SLOTS = 4 * G;  // MFMA pairs (= filler slots) per step
  float4 a[2][2][G];            // [step % 2][frequency of the pair][k group]
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const float4* vab = V4 + par * VBUF;
    const float4* va[G];  // per k group: plane (2 gi + kh), this lane's rotated wtile slot
#pragma unroll
    for (int gi = 0; gi < G; ++gi) va[gi] = vab + (2 * gi + kh) * NT + ((arow + SW * (2 * gi + kh)) % NT);
    const int nbuf = par ^ 1;
    // what the load slots of this chunk fetch: chunk + 2 of this tile, or - in the last two chunks - chunk 0 / 1 of
    // the next one (uniform selects, no branch in the pinned instruction stream)
    const bool nx = chunk + 2 >= nchunks;
    const int c2 = nx ? chunk + 2 - nchunks : chunk + 2;
    const i32x4 rl = nx ? rin_nx : rin;
    const unsigned pvl = nx ? pvoff_nx : pvoff;
    auto load_next = [&](int i) { load_piece_from(rl, pvl, c2, i); };
#pragma unroll
    for (int fs = 0; fs < 2; ++fs)
#pragma unroll
      for (int gi = 0; gi < G; ++gi) a[0][fs][gi] = va[gi][fs * CG * NT];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int slot = 0; slot < SLOTS; ++slot) {
        const int gi = slot >> 2, j = slot & 3;
        {
          const float4 a0 = a[s & 1][0][gi], a1 = a[s & 1][1][gi];
          const float4 b0 = bs[s & 3][0][gi], b1 = bs[s & 3][1][gi];
          const float a0v[4] = {a0.x, a0.y, a0.z, a0.w}, a1v[4] = {a1.x, a1.y, a1.z, a1.w};
          const float b0v[4] = {b0.x, b0.y, b0.z, b0.w}, b1v[4] = {b1.x, b1.y, b1.z, b1.w};
          acc[2 * s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[j], b0v[j], acc[2 * s], 0, 0, 0);
          acc[2 * s + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[j], b1v[j], acc[2 * s + 1], 0, 0, 0);
        }
        RTPOSE_PIN();
        if (slot < 2 * G) {  // A of the next step (the first step of a chunk is read after the barrier)
          if (s < 7)
            a[(s + 1) & 1][slot / G][slot % G] =
                RTPOSE_EXP_A(va[slot % G][(2 * (s + 1) + slot / G) * CG * NT], a[s & 1][slot / G][slot % G]);
        } else {             // B two steps ahead
          const int i = slot - 2 * G;
          bs[(s + 2) & 3][i / G][i % G] =
              RTPOSE_EXP_B(bload_f4(rw, boff, wso + ((i / G) * CG + 2 * (i % G)) * cgstep), bs[s & 3][i / G][i % G]);
          // (after step 5 of a tile's last chunk the ring wraps to the first steps of the next tile's chunk 0)
          if (slot == SLOTS - 1) wso = (BRING && s == 5 && chunk == nchunks - 1) ? 0u : wso + bstep;
        }
        if (RTPOSE_EXP_STAGE) {
          if (IPT == 1) {  // groups in the last slots of steps 0, 1; two patch loads in the last slot of steps 2..7
            if (slot == SLOTS - 1) {
              if (s < 2) {
                tgroup(nbuf, s);
              } else {
                load_next(2 * (s - 2));
                load_next(2 * (s - 2) + 1);
              }
            }
          } else {         // 4 groups in the middle and last slots of steps 0, 1; 16 loads in the B slots of steps 2..5
            if (s < 2 && (slot == SLOTS / 2 - 1 || slot == SLOTS - 1)) tgroup(nbuf, 2 * s + (slot == SLOTS - 1));
            if (s >= 2 && s < 6 && slot >= SLOTS - 4) load_next(4 * (s - 2) + slot - (SLOTS - 4));
          }
        }
        RTPOSE_PIN();
      }
    }
    __syncthreads();
    par ^= 1;
  }
  RTPOSE_TSTAMP3(ti, 2);
  rin = rin_nx;  // the next tile becomes the current one (its chunk 0 is in V[par], its chunk 1 in flight)
  pvoff = pvoff_nx;

  // (Tried in round 3 and rejected, profiles/r03_wino3_timeline_*: accumulating the TRANSPOSED tile - B fragment as the
  //  first MFMA operand, so that a lane owns one wtile and its 16 registers are 4 runs of 4 channels - with one address
  //  computation per lane and 16 16-byte stores instead of 64 dword stores: 816 instead of ~1100 VALU instructions, but
  //  the epilogue went from 8.9 k to 10.5-11.7 k cycles and the store drain from 0.3 k to 1.0 k: a lane's 16 bytes land
  //  in a cache line of their own (pixels are 512 B apart), while the dword stores below put 32 lanes on one 128-byte
  //  line.  A B prefetch distance of 3 steps instead of 2: +2 % time.  The transform below on packed pairs of wtiles:
  //  66 spilled registers.  Note for 16-byte stores in general: a VALU write of the store's data registers in the NEXT
  //  basic block is not separated from it by the compiler's hazard nops - it corrupted 4 lanes in 16.)
  // ---- epilogue: output transform A^T M A, (+ReLU) (+2x2 max-pool), masked stores -----------------
  // accumulator register r of a lane = wtile row (r / 4) * 8 + 4 kh + r % 4 of the wave tile, column l31.
  // Stores are raw buffer stores relative to the tile's first output pixel: 32-bit offsets (one multiply per
  // wtile instead of 64-bit address arithmetic per store), the +1 pixel / +1 row neighbours through the scalar
  // offset, and a lane that must not store gets an out-of-range offset instead of a branch around the store
  // (the epilogue was 12 % of the kernel: 350 address instructions and 280 branch instructions per tile).
  {
    const bool col_ok = ncol < g.cout;
    const int sc = A.pool ? 1 : 2;  // output pixels per wtile and direction
    auto wt_q = [&](int n, int ty, int tx) -> int { return (n * g.out_hs + sc * ty) * g.out_ws + sc * tx; };
    int q0;                          // uniform: first wtile of the block's tile
    {
      const int t = min(mt * NT, A.T - 1);
      const int n = t / TT, r = t - n * TT;
      const int ty = r / A.TX;
      q0 = wt_q(n, ty, r - ty * A.TX);
    }
    const size_t oo0 = ((size_t)g.out_lead + (size_t)q0) * g.out_cstride + g.out_choff;
    const i32x4 rout = make_rsrc(g.out + oo0, g.out_bytes - oo0 * 4);
    const unsigned cs4 = (unsigned)g.out_cstride * 4, row4 = (unsigned)g.out_ws * cs4;
    const unsigned col4 = (unsigned)ncol * 4;
#pragma unroll
    for (int rg = 0; rg < RTPOSE_EXP_W_EPI / 4; ++rg) {
      // 4 consecutive wtiles: one division pair, then +1 steps with a branch-free wrap
      int tcur = mt * NT + wm * 32 + rg * 8 + 4 * kh;
      int sn = tcur / TT, sy, sx;
      {
        const int r = tcur - sn * TT;
        sy = r / A.TX;
        sx = r - sy * A.TX;
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = rg * 4 + rr;
        float s[4][2];
#pragma unroll
        for (int fy = 0; fy < 4; ++fy) {
          s[fy][0] = acc[fy * 4 + 0][r] + acc[fy * 4 + 1][r] + acc[fy * 4 + 2][r];
          s[fy][1] = acc[fy * 4 + 1][r] - acc[fy * 4 + 2][r] - acc[fy * 4 + 3][r];
        }
        float y[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          y[0][x] = s[0][x] + s[1][x] + s[2][x];
          y[1][x] = s[1][x] - s[2][x] - s[3][x];
        }
        if (A.relu) {
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i >> 1][i & 1] = fmaxf(y[i >> 1][i & 1], 0.f);
        }
        const bool ok = col_ok && tcur < A.T;
        const unsigned off = (unsigned)(wt_q(sn, sy, sx) - q0) * cs4 + col4;
        if (A.pool) {  // H and W even: every wtile is one pooled pixel
          const float v = fmaxf(fmaxf(y[0][0], y[0][1]), fmaxf(y[1][0], y[1][1]));
          bstore(v, rout, ok ? off : kNoStore, 0);
        } else {
          const bool x1 = 2 * sx + 1 < A.W, y1 = 2 * sy + 1 < A.H;
          bstore(y[0][0], rout, ok ? off : kNoStore, 0);
          bstore(y[0][1], rout, (ok && x1) ? off : kNoStore, cs4);
          bstore(y[1][0], rout, (ok && y1) ? off : kNoStore, row4);
          bstore(y[1][1], rout, (ok && x1 && y1) ? off : kNoStore, row4 + cs4);
        }
        ++tcur;  // next wtile of the group
        const bool wx = sx + 1 >= A.TX, wy = wx && sy + 1 >= A.TY;
        sx = wx ? 0 : sx + 1;
        sy = wy ? 0 : (wx ? sy + 1 : sy);
        sn += wy ? 1 : 0;
      }
    }
  }
  RTPOSE_TSTAMP3(ti, 3);
#ifdef RTPOSE_EXP_TIMELINE3
  if (ti < 6) {  // when the stores of this tile have been acknowledged
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RTPOSE_TSTAMP3(ti, 4);
  }
#endif
  ++ti;
  }  // m tiles of this block
#undef RTPOSE_PIN
}

// Small grids (few images): the same arithmetic on more blocks, as wino7s_f32 (conv_wino7.hip).  A block = 32 wtiles x
// 32 columns; its four waves split the FREQUENCIES (wave w: fy = w, fx = 0..3) and exchange the accumulators through
// LDS before the output transform.  Every frequency sum runs over the chunks, k groups and k pairs in the order of
// wino_f32, the output transform is the same expression: bit-identical results, 4x (128-column blocks) or 2x
// (64-column blocks) as many blocks.  16-channel chunks only.
__global__ __launch_bounds__(256, 1) void wino3s_f32(const Args A) {
  constexpr int NT = 32, CK = 16, CG = 4, G = 2, SW = RTPOSE_EXP_W3_SW(CG);
  constexpr int VBUF = 16 * CG * NT;  // float4 per V buffer
  extern __shared__ __attribute__((aligned(16))) float4 V4[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int bi = blockIdx.x;
  const int c = bi % A.ncombo, mt = bi / A.ncombo;
  const int nt = c % A.ntiles, grp = c / A.ntiles;
  const Group g = grp ? A.g[1] : A.g[0];
  const int TT = A.TY * A.TX;

  // ---- input transform role: half a 4 x 4 patch of 4 channels, exactly as wino_f32 (IPT = 1) ------------
  const int cg = tid % CG, tl = (tid / CG) % NT;
  const int half = __builtin_amdgcn_readfirstlane(tid / (CG * NT));
  const float sgn = half ? -1.f : 1.f;
  const i32x4 rw = make_rsrc(g.w, g.w_bytes);
  i32x4 rin;
  unsigned pvoff;
  {
    auto patch_q = [&](int t) -> size_t {
      const int n = t / TT, r = t - n * TT;
      const int ty = r / A.TX, tx = r - ty * A.TX;
      return (size_t)g.in_lead + (size_t)(n * g.in_hs + 2 * ty - 1) * g.in_ws + (2 * tx - 1);
    };
    const size_t q0 = patch_q(min(mt * NT, A.T - 1));
    const size_t q = patch_q(min(mt * NT + tl, A.T - 1));
    const size_t o0 = q0 * g.in_cstride + g.in_choff;
    rin = make_rsrc(g.in + o0, g.in_bytes - o0 * 4);
    pvoff = (unsigned)(((q - q0) * g.in_cstride + cg * 4) * 4);
  }
  unsigned psoff[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int row = half ? (r == 0 ? 2 : r == 1 ? 3 : 1) : r;
    psoff[r] = (unsigned)(row * g.in_ws * g.in_cstride * 4);
  }
  const unsigned pxb = (unsigned)g.in_cstride * 4;
  F4 p[3][4], ta[4], tb[4];
  auto load_piece = [&](int chunk, int i) {
    p[i >> 2][i & 3] = bload(rin, pvoff, (unsigned)chunk * (CK * 4) + psoff[i >> 2] + (i & 3) * pxb);
  };
  const int vst = (half * 8 * CG + cg) * NT + ((tl + SW * cg) % NT);
  auto tgroup = [&](int buf, int gidx) {
    if (gidx == 0) {
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        ta[x] = sub4(p[0][x], p[2][x]);
        tb[x] = fma4(sgn, p[1][x], p[2][x]);
      }
    } else {
      float4* v = V4 + buf * VBUF + vst;
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const F4* t = (o & 4) ? tb : ta;
        const int fx = o & 3;
        v[o * CG * NT] = to_float4(fx == 0 ? sub4(t[0], t[2]) : fx == 1 ? add4(t[1], t[2]) : fx == 2 ? sub4(t[2], t[1])
                                                                                          : sub4(t[1], t[3]));
      }
    }
  };

  // ---- MFMA role: frequencies 4 wv .. 4 wv + 3 of the 32 x 32 tile ---------------------------------------
  const int ncol = nt * 32 + l31;
  floatx16 acc[4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
  if (wv == 1) {  // the bias rides in frequency (1,1) = 5
    const float b0 = g.bias[ncol];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[1][r] = b0;
  }
  const unsigned boff = (unsigned)((kh * g.cout_pad + ncol) * 16);
  const unsigned cgstep = (unsigned)(g.cout_pad * 16);  // bytes per channel group plane
  const unsigned fstep = (unsigned)CG * cgstep;         // bytes per frequency
  const unsigned cstep = 16 * fstep;                    // bytes per chunk
  const int nchunks = A.cin / CK;
  float4 bcur[4][G], bnxt[4][G];
  auto load_b = [&](float4 (&dst)[4][G], int chunk) {
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int gi = 0; gi < G; ++gi)
        dst[f][gi] = bload_f4(rw, boff, (unsigned)chunk * cstep + (unsigned)(4 * wv + f) * fstep + 2 * gi * cgstep);
  };
#pragma unroll
  for (int i = 0; i < 12; ++i) load_piece(0, i);
  tgroup(0, 0);
  tgroup(0, 1);
  {
    const int c1 = min(1, nchunks - 1);
#pragma unroll
    for (int i = 0; i < 12; ++i) load_piece(c1, i);
  }
  load_b(bcur, 0);
  __syncthreads();

#define RTPOSE_PIN()             \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const float4* vab = V4 + (chunk & 1) * VBUF;
    const int nbuf = (chunk + 1) & 1;
    const int c2 = min(chunk + 2, nchunks - 1);
    float4 a[4][G];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int gi = 0; gi < G; ++gi)
        a[f][gi] = vab[((4 * wv + f) * CG + 2 * gi + kh) * NT + ((wv * 0 + l31 + SW * (2 * gi + kh)) % NT)];
    load_b(bnxt, min(chunk + 1, nchunks - 1));  // the next chunk's B fragments (the last chunk re-reads its own)
    tgroup(nbuf, 0);
    RTPOSE_PIN();
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const float4 av = a[f][gi], bv = bcur[f][gi];
          const float avv[4] = {av.x, av.y, av.z, av.w}, bvv[4] = {bv.x, bv.y, bv.z, bv.w};
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(avv[j], bvv[j], acc[f], 0, 0, 0);
        }
        RTPOSE_PIN();
        if (gi == 0 && j == 1) tgroup(nbuf, 1);
        if (gi == 1) {
#pragma unroll
          for (int q = 0; q < 3; ++q) load_piece(c2, 3 * j + q);
        }
        RTPOSE_PIN();
      }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int gi = 0; gi < G; ++gi) bcur[f][gi] = bnxt[f][gi];
    __syncthreads();
  }
#undef RTPOSE_PIN

  // ---- exchange through LDS (the V buffers are free), then output transform + stores as in wino_f32 ------
  float* E = reinterpret_cast<float*>(V4);  // E[f][r][lane]: 16 x 16 x 64 floats = the two V buffers
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) E[((4 * wv + f) * 16 + r) * 64 + lane] = acc[f][r];
  __syncthreads();
  {
    const bool col_ok = ncol < g.cout;
    const int sc = A.pool ? 1 : 2;
    auto wt_q = [&](int n, int ty, int tx) -> int { return (n * g.out_hs + sc * ty) * g.out_ws + sc * tx; };
    int q0;
    {
      const int t = min(mt * NT, A.T - 1);
      const int n = t / TT, r = t - n * TT;
      const int ty = r / A.TX;
      q0 = wt_q(n, ty, r - ty * A.TX);
    }
    const size_t oo0 = ((size_t)g.out_lead + (size_t)q0) * g.out_cstride + g.out_choff;
    const i32x4 rout = make_rsrc(g.out + oo0, g.out_bytes - oo0 * 4);
    const unsigned cs4 = (unsigned)g.out_cstride * 4, row4 = (unsigned)g.out_ws * cs4;
    const unsigned col4 = (unsigned)ncol * 4;
    // wave wv stores accumulator registers 4 wv .. 4 wv + 3 = wtiles 8 wv + 4 kh + (0..3) of the tile
    int tcur = mt * NT + 8 * wv + 4 * kh;
    int sn = tcur / TT, sy, sx;
    {
      const int r = tcur - sn * TT;
      sy = r / A.TX;
      sx = r - sy * A.TX;
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = 4 * wv + rr;
      float m[16];
#pragma unroll
      for (int f = 0; f < 16; ++f) m[f] = E[(f * 16 + r) * 64 + lane];
      float s[4][2];
#pragma unroll
      for (int fy = 0; fy < 4; ++fy) {
        s[fy][0] = m[fy * 4 + 0] + m[fy * 4 + 1] + m[fy * 4 + 2];
        s[fy][1] = m[fy * 4 + 1] - m[fy * 4 + 2] - m[fy * 4 + 3];
      }
      float y[2][2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        y[0][x] = s[0][x] + s[1][x] + s[2][x];
        y[1][x] = s[1][x] - s[2][x] - s[3][x];
      }
      if (A.relu) {
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i >> 1][i & 1] = fmaxf(y[i >> 1][i & 1], 0.f);
      }
      const bool ok = col_ok && tcur < A.T;
      const unsigned off = (unsigned)(wt_q(sn, sy, sx) - q0) * cs4 + col4;
      if (A.pool) {
        const float v = fmaxf(fmaxf(y[0][0], y[0][1]), fmaxf(y[1][0], y[1][1]));
        bstore(v, rout, ok ? off : kNoStore, 0);
      } else {
        const bool x1 = 2 * sx + 1 < A.W, y1 = 2 * sy + 1 < A.H;
        bstore(y[0][0], rout, ok ? off : kNoStore, 0);
        bstore(y[0][1], rout, (ok && x1) ? off : kNoStore, cs4);
        bstore(y[1][0], rout, (ok && y1) ? off : kNoStore, row4);
        bstore(y[1][1], rout, (ok && x1 && y1) ? off : kNoStore, row4 + cs4);
      }
      ++tcur;
      const bool wx = sx + 1 >= A.TX, wy = wx && sy + 1 >= A.TY;
      sx = wx ? 0 : sx + 1;
      sy = wy ? 0 : (wx ? sy + 1 : sy);
      sn += wy ? 1 : 0;
    }
  }
}

// ---- weight packing: U = G g G^T, packed[chunk][f][cg][cout_pad][4]  <-  w[cout][cin_src][3][3] -----------
__global__ void pack_wino_kernel(const float* __restrict__ w, const float* __restrict__ bias, int cout,
                                 int cin_src, const int32_t* __restrict__ cin_map, int cin_packed, int ck,
                                 int coutp, float* __restrict__ wp, float* __restrict__ bp) {
  const size_t total = (size_t)16 * cin_packed * coutp;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)coutp) bp[i] = (i < (size_t)cout && bias) ? bias[i] : 0.f;
  if (i >= total) return;
  const int e = i & 3;
  size_t r = i >> 2;
  const int n = r % coutp;
  r /= coutp;
  const int cg = r % (ck / 4);
  r /= (ck / 4);
  const int f = r % 16;
  const int chunk = r / 16;
  const int c = chunk * ck + cg * 4 + e;
  const int src = cin_map ? cin_map[c] : (c < cin_src ? c : -1);
  float v = 0.f;
  if (n < cout && src >= 0 && src < cin_src) {
    const float* gw = w + ((size_t)n * cin_src + src) * 9;
    const int fy = f >> 2, fx = f & 3;
    // rows of G: (1,0,0), (.5,.5,.5), (.5,-.5,.5), (0,0,1)
    float col[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      const float g0 = gw[x], g1 = gw[3 + x], g2 = gw[6 + x];
      col[x] = fy == 0 ? g0 : fy == 1 ? 0.5f * (g0 + g1 + g2) : fy == 2 ? 0.5f * (g0 - g1 + g2) : g2;
    }
    v = fx == 0 ? col[0]
        : fx == 1 ? 0.5f * (col[0] + col[1] + col[2])
        : fx == 2 ? 0.5f * (col[0] - col[1] + col[2])
                  : col[2];
  }
  wp[i] = v;
}

#ifdef RTPOSE_EXP_TIMELINE3
static unsigned long long* g_dbgw3_buf = nullptr;
static unsigned g_dbgw3_blocks = 0;
#endif

template <int WM, int WN, int CK>
static int launch_inst(const Args& a, dim3 grid, hipStream_t s) {
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  auto kern = wino_f32<WM, WN, CK>;
  constexpr size_t lds = (size_t)2 * 16 * (CK / 4) * (32 * WM) * 16;
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set.set(dev);
  }
#ifdef RTPOSE_EXP_TIMELINE3
  {  // developer build: per-tile stamps of the LAST launch of this instance family (tools/timeline_w3.py)
    static int only_wm = -1;
    if (only_wm < 0) {
      const char* e = dev_env("RTPOSE_TIMELINE_WM");  // 1: <1,4,16> launches, 2: conv1_2
      only_wm = e ? atoi(e) : 1;
    }
    Args b = a;
    if (WM == only_wm && grid.x <= 1024) {
      if (!g_dbgw3_buf) (void)hipMalloc(&g_dbgw3_buf, (size_t)1024 * 6 * 8 * 8);
      (void)hipMemsetAsync(g_dbgw3_buf, 0, (size_t)grid.x * 6 * 64, s);
      b.dbg = g_dbgw3_buf;
      g_dbgw3_blocks = grid.x;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, b);
    RTPOSE_HIP_CHECK(hipGetLastError());
    return 0;
  }
#endif
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace wino

#ifdef RTPOSE_EXP_TIMELINE3
extern "C" int rtpose_debug_timeline_w3_dump(unsigned long long* host, unsigned cap_blocks) {
  using namespace wino;
  if (!g_dbgw3_buf) return 0;
  (void)hipDeviceSynchronize();
  const unsigned n = g_dbgw3_blocks < cap_blocks ? g_dbgw3_blocks : cap_blocks;
  (void)hipMemcpy(host, g_dbgw3_buf, (size_t)n * 6 * 64, hipMemcpyDeviceToHost);
  return (int)n;
}
#endif

// channel chunk the packed Winograd weights of a conv are laid out for (the kernel instance is chosen by
// the padded output width: >= 128 columns -> 32 wtiles x 128 columns, 16-channel chunks; 64 -> 64 x 64, 8)
// channel chunk of a conv: 16, or 8 where the input has only a multiple of 8 channels and the block is the 64 x 64 one
static int wino_ck(int cout, int cin) { return (cout_pad(cout) % 128 == 0 || cin % 16 == 0) ? 16 : 8; }

// (at least two chunks: the chunk pipeline of wino_f32 fetches two chunks ahead, across tiles)
int conv2d_wino_ok(int cin, int cout, int k) {
  return k == 3 && cin > 0 && cin % wino_ck(cout, cin) == 0 && cin / wino_ck(cout, cin) >= 2;
}

int conv2d_wino_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, hipStream_t s) {
  using namespace wino;
  if (!d || ngroups < 1 || ngroups > 2) return fail(RTPOSE_E_INVAL, "conv2d_winograd: ngroups must be 1 or 2");
  RTPOSE_REFUSE_PLANES(d, ngroups, "conv2d_winograd (2x2)");
  const rtpose_conv_desc& d0 = d[0];
  if (!conv2d_wino_ok(d0.cin, d0.cout, d0.k))
    return fail(RTPOSE_E_INVAL, "conv2d_winograd: k must be 3 and cin a multiple of %d", wino_ck(d0.cout, d0.cin));
  if (N <= 0 || H <= 0 || W <= 0) return fail(RTPOSE_E_INVAL, "conv2d_winograd: empty tensor");
  if (d0.pool && ((H | W) & 1)) return fail(RTPOSE_E_INVAL, "conv2d_winograd: fused pool needs even H and W");
  Args a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < ngroups; ++i) {
    const rtpose_conv_desc& di = d[i];
    if (di.k != 3 || di.cin != d0.cin || di.relu != d0.relu || di.pool != d0.pool ||
        cout_pad(di.cout) != cout_pad(d0.cout) || di.lin.ws != d0.lin.ws || di.lin.hs != d0.lin.hs)
      return fail(RTPOSE_E_INVAL, "conv2d_winograd: grouped convs must share geometry");
    if (di.lin.ws < W + 1 || di.lin.hs < H + 1 || di.lin.lead < di.lin.ws + 1)
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
    g.w_bytes = rtpose_packed_weight_floats_winograd(di.cout, di.cin, 3) * sizeof(float);
    g.out_bytes = rtpose_layout_pixels(&di.lout, N, di.pool ? H / 2 : H, di.pool ? W / 2 : W) *
                  (size_t)di.lout.cstride * sizeof(float);
  }
  a.N = N;
  a.H = H;
  a.W = W;
  a.TY = ceil_div(H, 2);
  a.TX = ceil_div(W, 2);
  const long T = (long)N * a.TY * a.TX;
  if (T > 0x7fffffffL) return fail(RTPOSE_E_INVAL, "conv2d_winograd: tensor too large");
  a.T = (int)T;
  a.cin = d0.cin;
  a.relu = d0.relu;
  a.pool = d0.pool;
  const int ck = wino_ck(d0.cout, d0.cin);
  const int wm = cout_pad(d0.cout) % 128 == 0 ? 1 : 2, wn = 4 / wm;
  a.mtiles = ceil_div(a.T, 32 * wm);
  a.ntiles = cout_pad(d0.cout) / (32 * wn);
  a.ncombo = a.ntiles * ngroups;
  a.xcd_remap = (a.ncombo > 1 && a.mtiles >= 64) ? 1 : 0;
  long ids = a.xcd_remap ? (long)8 * a.ncombo * ceil_div(a.mtiles, 8) : (long)a.mtiles * a.ncombo;
  if (ids > 0x7fffffffL) return fail(RTPOSE_E_INVAL, "conv2d_winograd: grid too large");
  {
    // persistent blocks when a CU would get more than one tile anyway (see wino_f32)
    const int n_cu = device_cu_count();
    static int persist_env = -1;
    if (persist_env < 0) {
      const char* e = dev_env("RTPOSE_W3_PERSIST");
      persist_env = e ? atoi(e) : 1;
    }
    if (persist_env && (long)a.mtiles * a.ncombo > n_cu && n_cu % a.ncombo == 0) {
      a.persist = 1;
      ids = n_cu;
    }
  }
  const dim3 grid((unsigned)ids, 1, 1);
  if (ck == 16 && !a.persist && (long)a.mtiles * a.ncombo * 2 <= device_cu_count()) {
    // small grids: the frequency-split form (wino3s_f32), bit-identical, 32 x 32 tiles
    Args b = a;
    b.mtiles = ceil_div(a.T, 32);
    b.ntiles = cout_pad(d0.cout) / 32;
    b.ncombo = b.ntiles * ngroups;
    static PerDeviceOnce attr_set;
    const int dev = current_device();
    if (!attr_set.is_set(dev)) {
      RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino3s_f32),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      attr_set.set(dev);
    }
    hipLaunchKernelGGL(wino3s_f32, dim3((unsigned)((long)b.mtiles * b.ncombo)), dim3(256),
                       (size_t)2 * 16 * 4 * 32 * 16, s, b);
    RTPOSE_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (wm == 1) return launch_inst<1, 4, 16>(a, grid, s);
  // 64 columns (conv1_2): 64 wtiles x 64 columns; 16-channel chunks with a whole patch per thread where cin allows
  if (ck == 16) return launch_inst<2, 2, 16>(a, grid, s);
  return launch_inst<2, 2, 8>(a, grid, s);
}

int pack_weights_wino_launch(const float* w, const float* bias, int cout, int cin_src, const int32_t* cin_map,
                             int cin_packed, float* wp, float* bp, hipStream_t s) {
  if (!conv2d_wino_ok(cin_packed, cout, 3) || (cin_packed < cin_src && !cin_map))
    return fail(RTPOSE_E_INVAL, "pack_winograd: cin_packed must be a multiple of %d and >= cin_src", wino_ck(cout, cin_packed));
  const int coutp = cout_pad(cout);
  const size_t total = (size_t)16 * cin_packed * coutp;
  const int threads = 256;
  const unsigned blocks = (unsigned)((total + threads - 1) / threads);
  hipLaunchKernelGGL(wino::pack_wino_kernel, dim3(blocks), dim3(threads), 0, s, w, bias, cout, cin_src, cin_map,
                     cin_packed, wino_ck(cout, cin_packed), coutp, wp, bp);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

// MFMA flops a launch ISSUES (what SQ_INSTS_MFMA x 4096 counts): whole tiles of 32 WM wtiles x all 16 frequencies
double conv2d_wino_issued_flops(int cin, int cout, int N, int H, int W) {
  const int wm = cout_pad(cout) % 128 == 0 ? 1 : 2;
  const double T = (double)N * ceil_div(H, 2) * ceil_div(W, 2);
  const double tiles = std::ceil(T / (32.0 * wm));
  return 2.0 * tiles * 32.0 * wm * 16.0 * (double)cin * cout_pad(cout);
}

// k = 7: csrc/conv_wino7.hip
int conv2d_wino7_fits(int cin, int cout, int N, int H, int W, int hs, int fm);
int conv2d_wino7_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, int fm, void* scratch,
                        size_t scratch_bytes, hipStream_t s);
int pack_weights_wino7_launch(const float* w, const float* bias, int cout, int cin_src, const int32_t* cin_map,
                              int cin_packed, int fm, float* wp, float* bp, hipStream_t s);
size_t packed_weight_floats_wino7(int cout, int cin, int fm);
size_t conv2d_wino7_scratch_bytes(int blocks);
int* conv2d_wino7_scratch_err(void* scratch, int blocks);
int wino_amplification_launch(const float* w, int cout, int cin, int k, int fm, float* amp, hipStream_t s);
// k = 3, F(4x4,3x3): csrc/conv_wino4.hip
int conv2d_wino4_ok(int cin, int cout);
size_t packed_weight_floats_wino4(int cout, int cin);
int conv2d_wino4_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, hipStream_t s);
int pack_weights_wino4_launch(const float* w, const float* bias, int cout, int cin_src, const int32_t* cin_map,
                              int cin_packed, float* wp, float* bp, hipStream_t s);
int wino4_amplification_launch(const float* w, int cout, int cin, float* amp, hipStream_t s);

int conv2d_winograd_fits(int k, int cin, int cout, int pool, int N, int H, int W, int hs, int fm) {
  if (k == 3) return fm == 4 ? conv2d_wino4_ok(cin, cout) : conv2d_wino_ok(cin, cout, 3);
  if (k == 7) return !pool && conv2d_wino7_fits(cin, cout, N, H, W, hs, fm);
  return 0;
}

}  // namespace rtpose

extern "C" {

// rtpose_conv_desc.wino_m selects the form per launch: k = 3: 0 / 2 = F(2x2,3x3), 4 = F(4x4,3x3); k = 7: 0 = library
// default, 4 / 6 = F(m,7).  Anything else is a descriptor that was not zero-initialised: refuse it instead of running a
// kernel on the wrong packing.
static bool wino_m_valid(const rtpose_conv_desc* d) {
  const int m = d->wino_m;
  return d->k == 3 ? (m == 0 || m == 2 || m == 4) : d->k == 7 ? (m == 0 || m == 4 || m == 6) : true;
}

int rtpose_conv2d_winograd_fits(const rtpose_conv_desc* d, int N, int H, int W) {
  if (!d || !wino_m_valid(d)) return 0;
  return rtpose::conv2d_winograd_fits(d->k, d->cin, d->cout, d->pool, N, H, W, d->lin.hs, d->wino_m == 2 ? 0 : d->wino_m);
}

size_t rtpose_packed_weight_floats_winograd7(int cout, int cin, int m) {
  return rtpose::packed_weight_floats_wino7(cout, cin, m);
}

size_t rtpose_packed_weight_floats_winograd(int cout, int cin, int k) {
  if (k == 7) return rtpose::packed_weight_floats_wino7(cout, cin, 0);
  // + 4 (chunk, frequency) blocks: the B prefetch runs two steps ahead
  return (size_t)(16 * cin + 64) * rtpose::cout_pad(cout);
}

size_t rtpose_packed_weight_floats_winograd3(int cout, int cin, int m) {
  if (m == 4) return rtpose::packed_weight_floats_wino4(cout, cin);
  return rtpose_packed_weight_floats_winograd(cout, cin, 3);
}

int rtpose_pack_conv_weights_winograd3(const float* w_oihw, const float* bias, int cout, int cin_src, int m,
                                       const int32_t* cin_map, int cin_packed, float* w_packed,
                                       float* bias_packed, void* stream) {
  if (m == 4)
    return rtpose::pack_weights_wino4_launch(w_oihw, bias, cout, cin_src, cin_map, cin_packed, w_packed, bias_packed,
                                             rtpose::as_stream(stream));
  if (m != 0 && m != 2) return rtpose::fail(RTPOSE_E_INVAL, "pack_winograd3: m must be 0 / 2 (F(2x2,3x3)) or 4 (F(4x4,3x3))");
  return rtpose::pack_weights_wino_launch(w_oihw, bias, cout, cin_src, cin_map, cin_packed, w_packed, bias_packed,
                                          rtpose::as_stream(stream));
}

int rtpose_pack_conv_weights_winograd7(const float* w_oihw, const float* bias, int cout, int cin_src, int m,
                                       const int32_t* cin_map, int cin_packed, float* w_packed,
                                       float* bias_packed, void* stream) {
  return rtpose::pack_weights_wino7_launch(w_oihw, bias, cout, cin_src, cin_map, cin_packed, m, w_packed, bias_packed,
                                           rtpose::as_stream(stream));
}

int rtpose_pack_conv_weights_winograd(const float* w_oihw, const float* bias, int cout, int cin_src, int k,
                                      const int32_t* cin_map, int cin_packed, float* w_packed,
                                      float* bias_packed, void* stream) {
  if (k == 7)
    return rtpose::pack_weights_wino7_launch(w_oihw, bias, cout, cin_src, cin_map, cin_packed, 0, w_packed,
                                             bias_packed, rtpose::as_stream(stream));
  if (k != 3) return rtpose::fail(RTPOSE_E_INVAL, "pack_winograd: k must be 3 or 7");
  return rtpose::pack_weights_wino_launch(w_oihw, bias, cout, cin_src, cin_map, cin_packed, w_packed,
                                          bias_packed, rtpose::as_stream(stream));
}

size_t rtpose_conv2d_winograd_scratch_bytes(void) {
  return rtpose::conv2d_wino7_scratch_bytes(rtpose::device_cu_count());
}

int rtpose_conv2d_winograd_ex(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, void* scratch,
                              size_t scratch_bytes, void* stream) {
  for (int g = 0; d && g < ngroups && g < 2; ++g)
    if (!wino_m_valid(&d[g]) || d[g].wino_m != d[0].wino_m)
      return rtpose::fail(RTPOSE_E_INVAL, "conv2d_winograd: rtpose_conv_desc.wino_m must be 0 / 2 / 4 for k = 3 and 0 / 4 / 6 "
                                          "for k = 7, the same in every group (zero-initialise descriptors)");
  if (d && d[0].k == 7)
    return rtpose::conv2d_wino7_launch(d, ngroups, N, H, W, d[0].wino_m, scratch, scratch_bytes,
                                       rtpose::as_stream(stream));
  if (d && d[0].k == 3 && d[0].wino_m == 4) return rtpose::conv2d_wino4_launch(d, ngroups, N, H, W, rtpose::as_stream(stream));
  return rtpose::conv2d_wino_launch(d, ngroups, N, H, W, rtpose::as_stream(stream));
}

int rtpose_conv2d_winograd_scratch_error(const void* scratch, int* error_word, void* stream) {
  if (!scratch || !error_word) return rtpose::fail(RTPOSE_E_INVAL, "winograd_scratch_error: NULL argument");
  int* err = rtpose::conv2d_wino7_scratch_err(const_cast<void*>(scratch), rtpose::device_cu_count());
  hipStream_t s = rtpose::as_stream(stream);
  RTPOSE_HIP_CHECK(hipMemcpyAsync(error_word, err, sizeof(int), hipMemcpyDeviceToHost, s));
  RTPOSE_HIP_CHECK(hipStreamSynchronize(s));
  return 0;
}

int rtpose_conv2d_winograd(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, void* stream) {
  return rtpose_conv2d_winograd_ex(d, ngroups, N, H, W, nullptr, 0, stream);
}

int rtpose_winograd_amplification(const float* w_oihw, int cout, int cin, int k, int m, float* amp_device,
                                  void* stream) {
  if (k == 3 && m == 4) return rtpose::wino4_amplification_launch(w_oihw, cout, cin, amp_device, rtpose::as_stream(stream));
  return rtpose::wino_amplification_launch(w_oihw, cout, cin, k, m, amp_device, rtpose::as_stream(stream));
}

}  // extern "C"
