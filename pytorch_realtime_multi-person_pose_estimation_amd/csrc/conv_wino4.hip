// fp32 Winograd F(4x4, 3x3) convolution for gfx950 (MI355X): stride 1, "same" padding, fused bias (+ReLU) (+2x2 max-pool).
//
// Stands in for the 3x3 nn.Conv2d + nn.ReLU (+ nn.MaxPool2d) modules of the VGG-19 front end and of stage 1
// (lib/network/rtpose_vgg.py:23-35, :95-105) where the plan selects the form (rtpose_net_options.winograd3 = 4 and the
// layer's amplification estimate is below amp_limit; rtpose_conv_desc.wino_m = 4 for a single launch).  A "wtile" is a
// 4 x 4 block of output pixels, computed from a 6 x 6 input patch through 36 "frequencies":
//
//   Y = A^T [ sum_c (G g_c G^T) o (B^T d_c B) ] A        36 multiplies per 16 outputs and input channel instead of 144
//
// i.e. 4x fewer matrix-core flops than the direct sum and 1.78x fewer than F(2x2,3x3) (conv_wino.hip), paid for with
// transforms that are no longer additions only.  Interpolation points 0, +-3/4, +-3/2, inf: every entry of B^T and A^T
// is a dyadic fraction (exact in fp32), and of the symmetric point pairs tried (oracle/winograd_tables.py,
// DESIGN.md §3.0) this one has the smallest measured element-wise error - gamma ~5 against 20 for the textbook points
// 0, +-1, +-2 (F(2x2,3x3): 1.5..18; the 7x7 forms the network already runs: F(4,7) 277, F(6,7) 439).
//
// MI355X shape (wino4_f32; the small-grid form wino4s_f32 is described at its definition):
//  * a block = 8 waves = 32 wtiles x 64 output columns, two waves per SIMD, on v_mfma_f32_16x16x4_f32.  The two waves
//    of a SIMD split the FREQUENCIES: wave (fh, wn) owns the 18 frequencies 18 fh .. 18 fh + 17 of 32 wtiles (two row tiles
//    of 16) x the 16 columns 16 wn ..: 36 accumulators x 4 registers = 144 of the 256 a wave may hold at two waves per
//    SIMD.  (A wave with all 36 frequencies of ONE row tile has the same 144 registers but fetches every filter fragment
//    for 4 MFMAs instead of 8: 32 B/clk per CU of filter data through the 64 B/clk vector-memory path - 1.18 ms per
//    256 -> 256 layer against 1.02.)  Before the output transform the siblings (waves w, w ^ 4) swap the first-pass row sums
//    of the row tile they do not finish through the LDS buffers that are dead at a tile boundary.
//  * channel chunks of 8: per chunk and frequency PAIR a wave issues 8 MFMAs fed by TWO ds_read_b128 (A: the pair's
//    values of channels 2 kq, 2 kq + 1 for wtile r16 of each row tile) and ONE 16-byte buffer load (B: the same for column
//    r16), 5 pairs ahead in a 6-entry register ring (two chunks are unrolled so that the ring rotates in step).  MFMA j of a
//    frequency contracts the channels {2 kq + j : kq = 0..3}.
//  * the 2-D input transform does not fit beside 144 accumulators in one piece (a 6 x 6 patch of 4 channels is 144
//    registers), so it runs in two 1-D stages through LDS, each 12 packed fmas + 6 LDS accesses of 16 bytes per item:
//      stage 1  item (wtile, channel group, patch row y): 6 pixels -> B^T along x -> U[fx][y]        waves 0..5 (y = wave)
//      stage 2  item (wtile, channel group, fx): U[fx][0..5] -> B^T along y -> V[fx * 6 + fy]         waves 2..7 (fx = wave - 2)
//    (three items per SIMD either way).  Pipeline over the "positions" (tile, chunk) of a persistent block, one barrier
//    per chunk: while chunk s is multiplied from V[s & 1], stage 2 turns U[(s + 1) & 1] into V[(s + 1) & 1], stage 1 turns the
//    patch rows of s + 2 (in registers) into U[s & 1], and the rows of s + 3 are requested.  Siblings on a SIMD take
//    their transform turns one step apart, so the matrix pipe always has a wave that multiplies.  Every wave issues the
//    patch loads and the U reads - the waves without an item through a zero-extent descriptor / into registers nobody
//    uses: a load under a wave-uniform branch makes the compiler's vmcnt bookkeeping assume the worst at every join.
//  * persistent blocks when there are more tiles than CUs; the blocks that multiply the SAME m tiles (one per column tile
//    / branch) sit on one XCD and share the patch lines in its L2; a left-over of at most a quarter round runs as a
//    second launch in the small form.
//  * patch rows / columns past the image (the last wtile row / column when H, W are not multiples of 4) are CLAMPED
//    to the zero gap row / column of the shared-gap layout: they would only meet outputs that are not stored, but
//    through the transforms they cancel only up to rounding, and an image would depend on its neighbour in the batch.
//  * B = transformed filters packed [chunk][frequency pair][kq][cout_pad][2 frequencies][2 channels]
//    (rtpose_pack_conv_weights_winograd3, m = 4).
//  * round 4 - CHANNEL-PLANE activations (rtpose_conv_desc.in_plane_pixels / out_plane_pixels): a chunk reads 32 bytes per
//    patch pixel.  Pixel-major (NHWC) that is a quarter of a 128-byte line; the rest of the line belongs to the next three
//    chunks, thousands of cycles later, and by then the line has left the L1 and often the L2: every line is moved up to four
//    times (FETCH_SIZE 3.7x the layer inputs).  Stored as planes of 8 channels - element (pixel q, channel c) at
//    ((c / 8) * Q + q) * 8 + c % 8, Q = the buffer's pixel slots - the 32 bytes of the six pixels of a patch row are ONE
//    192-byte run, used in full by the six loads a lane issues back to back: 3x3 layers of the 32-image forward 8.13 -> 7.50
//    ms.  Nothing else changes: the patch loader has two more runtime strides (pixel pitch 32 bytes, chunk pitch = one plane),
//    the epilogue's column offset becomes (ncol / 8) * plane + ncol % 8.  Both sides are per launch, so the first and the
//    last conv of a chain convert for free (csrc/net.hip decides per buffer).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "conv_exp.h"
#include "wino_common.h"

namespace rtpose {

namespace wino4 {

using namespace winoc;

typedef float floatx4 __attribute__((ext_vector_type(4)));

struct Group {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  int in_cstride, in_choff, in_ws, in_hs, in_lead;
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  int cout, cout_pad;
  int in_pq, out_pq;  // > 0: the slice is stored as planes of 8 channels, this many pixel slots per plane (0: pixel-major)
  size_t in_bytes, w_bytes, out_bytes;
};

// strides of a slice in either storage: floats per pixel step, float offset of the slice's first channel, bytes per
// 8-channel chunk step
__device__ __forceinline__ int px_floats(int cstride, int pq) { return pq ? 8 : cstride; }
__device__ __forceinline__ size_t ch0_floats(int choff, int pq) { return pq ? (size_t)(choff >> 3) * pq * 8 + (choff & 7) : choff; }
// float offset of output column `col` (relative to the slice) from its pixel's base
__device__ __forceinline__ unsigned col_floats(int col, int pq) { return pq ? (unsigned)(col >> 3) * pq * 8 + (col & 7) : col; }

struct Args {
  Group g[2];
  int N, H, W;
  int TY, TX, T;  // wtiles per column / row of an image, and in the whole batch
  int cin;
  int relu, pool;
  int mtiles, ntiles, ncombo, xcd_remap;
  int persist;
  int mt0;  // first m tile of this launch (a layer may run as a persistent launch of whole rounds + a launch of the rest)
#ifdef RTPOSE_EXP_TIMELINE4
  unsigned long long* dbg;  // developer build: [block][wave][64 chunks][2] s_memtime stamps of the block's first tile
#endif
};

constexpr int NT = 32;    // wtiles per block
constexpr int NC = 64;    // output columns per block
constexpr int CK = 8;     // channels per chunk
constexpr int NFP = 18;   // frequency pairs
constexpr int NPW = 9;    // frequency pairs of a wave (the two waves of a SIMD split the 18)
constexpr int NFW = 18;   // frequencies of a wave
constexpr int NB = 6;     // B ring entries (one per frequency pair)
constexpr int PF = 5;     // B prefetch distance in pairs
constexpr int VBUF = NFP * 4 * NT;  // float4 per V buffer: [pair][kq][wtile]
constexpr int UBUF = 36 * 2 * NT;   // float4 per U buffer: [fx][y][channel group][wtile]
static_assert((2 * NPW) % NB == 0, "the B ring rotates in step with two chunks");

// B^T of F(4,3) for the points 0, +-3/4, +-3/2, inf along one axis (rows scaled by N_f; the filter transform divides):
//   [81/64 0 -45/16 0 1 0], [0 -+27/16 -9/4 +-3/4 1 0], [0 -+27/32 -9/16 +-3/2 1 0], [0 81/64 0 -45/16 0 1]
// the +-p rows share their even and odd halves: 12 fmas on 4 channels = 24 packed instructions
// Every row goes to `out(f, value)` as soon as it is formed (the scheduler is fenced between the groups): the six
// inputs plus a few temporaries are all the registers the transform holds at any time, not inputs + six outputs.
template <class Out>
__device__ __forceinline__ void bt6(const F4 (&d)[6], Out out) {
  out(0, fma4(1.265625f, d[0], fma4(-2.8125f, d[2], d[4])));
  __builtin_amdgcn_sched_barrier(0);
  {
    const F4 e1 = fma4(-2.25f, d[2], d[4]), o1 = fma4(-2.25f, d[1], d[3]);
    out(1, fma4(0.75f, o1, e1));
    out(2, fma4(-0.75f, o1, e1));
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    const F4 e2 = fma4(-0.5625f, d[2], d[4]), o2 = fma4(-0.5625f, d[1], d[3]);
    out(3, fma4(1.5f, o2, e2));
    out(4, fma4(-1.5f, o2, e2));
  }
  __builtin_amdgcn_sched_barrier(0);
  out(5, fma4(1.265625f, d[1], fma4(-2.8125f, d[3], d[5])));
}

// A^T of F(4,3) along one axis: [1 1 1 1 1 0], [0 3/4 -3/4 3/2 -3/2 0], [0 9/16 9/16 9/4 9/4 0], [0 27/64 -27/64 27/8 -27/8 1]
__device__ __forceinline__ f2 splat(float s) { return f2{s, s}; }
// (in halves - outputs 0, 1 / 2, 3 - so that the epilogue can work on two output rows at a time)
__device__ __forceinline__ void at4_lo(const f2 (&m)[6], f2& y0, f2& y1) {
  const f2 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  y0 = (m[0] + s12) + s34;
  y1 = __builtin_elementwise_fma(splat(1.5f), d34, splat(0.75f) * d12);
}
__device__ __forceinline__ void at4_hi(const f2 (&m)[6], f2& y2, f2& y3) {
  const f2 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  y2 = __builtin_elementwise_fma(splat(2.25f), s34, splat(0.5625f) * s12);
  y3 = __builtin_elementwise_fma(splat(3.375f), d34, splat(0.421875f) * d12) + m[5];
}

// RT = row tiles of 16 wtiles per block: 2 = the 32 x 64 tile described above; 1 = HALF tiles, 16 wtiles x 64 columns (round 4),
// for the left-over of a layer whose tiles are not whole rounds of the 256 CUs (8.27 rounds of 32 x 64 tiles on the 92 x 92 maps,
// 4.5 on 46 x 46: the ninth / fifth round ran a quarter / half empty).  Same waves (fh, wn), same frequencies per wave, same
// sums in the same order - bit-identical; a wave multiplies ONE row tile (36 MFMAs per chunk instead of 72), the transform
// items are spread as in wino4s_f32 (waves 0..2: two patch rows each, waves 3..5: two fx each), and before the output
// transform the siblings split the row tile by wtile PAIRS: wave (fh, wn) finishes the wtiles 4 kq + 2 fh, + 1 and hands the
// first-pass row sums of the other pair over.  One tile per block (never persistent).
template <int RT>
__global__ __launch_bounds__(512, 1) void wino4_f32(const Args A) {
  constexpr int NT = 16 * RT;           // (shadow the 32-wtile constants of the namespace inside the kernel)
  constexpr int VBUF = NFP * 4 * NT;
  constexpr int UBUF = 36 * 2 * NT;
  extern __shared__ __attribute__((aligned(16))) float4 L4[];
  // LDS: [V0][U1][V1][U0] - the two buffers that are dead at a tile boundary (V1, U0) are adjacent: the exchange area
  auto Vb = [&](int i) -> float4* { return L4 + i * (VBUF + UBUF); };
  auto Ub = [&](int i) -> float4* { return L4 + (1 - i) * (VBUF + UBUF) + VBUF; };
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wv & 3, fh = wv >> 2;  // 16-column slice and frequency half (pairs 9 fh .. 9 fh + 8) of the wave
  const int r16 = lane & 15, kq = lane >> 4;

  // ---- block -> (n tile, group) and its m tiles (as wino_f32) ----------------------------------------------------
  const int bi = blockIdx.x;
  int j0, jstep, c;
  if (A.persist) {
    // Workgroups go round-robin to the 8 XCDs.  The ncombo blocks that multiply the SAME m tiles (one per column tile /
    // group) sit on one XCD: they fetch the same patch rows, 32 bytes of a 128-byte line per chunk, and only if the
    // XCD's 4 MB L2 has to hold 32 / ncombo m tiles' lines instead of 32 do the lines survive until the next chunk
    // takes its 32 bytes.  (The filters of all column tiles then stream through every L2; they are read in long runs.)
#if RTPOSE_EXP_W4_XCDMAP
    const int xcd = bi & 7, idx = bi >> 3;
    c = idx % A.ncombo;
    j0 = (idx / A.ncombo) * 8 + xcd;
#else
    c = bi % A.ncombo;
    j0 = bi / A.ncombo;
#endif
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
    jstep = A.mtiles;
  }
  if (j0 >= A.mtiles) return;
  const int nt = c % A.ntiles, grp = c / A.ntiles;
  const Group g = grp ? A.g[1] : A.g[0];
  const int TT = A.TY * A.TX;

  // ---- transform roles ----------------------------------------------------------------------------------------------
  const bool s1 = RT == 2 ? wv < 6 : wv < 3, s2 = RT == 2 ? wv >= 2 : (wv >= 3 && wv < 6);
  const int turn = wv >> 2;               // siblings on a SIMD (waves w, w + 4) take their turns one step apart
  // stage 1 lanes: (wtile, channel group) = (lane / 2, lane % 2) - the two lanes that share a pixel's 32 bytes are
  // neighbours, so the patch loads touch 32 lines per instruction, not 64; stage 2 lanes: (lane % 32, lane / 32) - a
  // plane of V per channel-group half.  U[fx][y][cg][wtile ^ 4 cg]: the swizzle keeps both access patterns off each
  // other's banks (8 consecutive stage-1 lanes write 4 + 4 slots of two planes).
  // (half tiles: 16 wtiles x 2 channel groups are 32 lanes - a wave takes two patch rows / two fx)
  const int wl1 = RT == 2 ? lane >> 1 : (lane >> 1) & 15, cg1 = lane & 1;
  const int wl = RT == 2 ? lane & 31 : lane & 15, cg = RT == 2 ? lane >> 5 : (lane >> 4) & 1;
  const int py = RT == 2 ? wv : 2 * (s1 ? wv : 0) + (lane >> 5);               // stage 1: patch row
  const int fx = RT == 2 ? max(wv - 2, 0) : 2 * (s2 ? wv - 3 : 0) + (lane >> 5);  // stage 2: frequency along x
  const i32x4 rw = make_rsrc(g.w, g.w_bytes);
  const i32x4 rnull = make_rsrc(g.in, 0);
  // the "load cursor": the (tile, chunk) position whose patch rows are requested next, 3 positions ahead of the multiply
  i32x4 rin;
  unsigned pv[4];  // byte offsets of pixel 0 and of the pixels 3, 4, 5 (clamped to the gap column) of the row
  int lt = j0, lc = 3;
  const int cs_ld = px_floats(g.in_cstride, g.in_pq);
  const unsigned pxb = (unsigned)cs_ld * 4;                                 // bytes per pixel step
  const unsigned ckb = g.in_pq ? (unsigned)g.in_pq * (CK * 4) : CK * 4;     // bytes per chunk step
  const size_t in_ch0 = ch0_floats(g.in_choff, g.in_pq);
  auto set_loader = [&](int mt, i32x4& r_, unsigned (&v_)[4]) {
    size_t q0;
    {
      const int t = min((A.mt0 + mt) * NT, A.T - 1);
      const int n = t / TT, r = t - n * TT;
      const int ty = r / A.TX, tx = r - ty * A.TX;
      q0 = (size_t)g.in_lead + (size_t)(n * g.in_hs + 4 * ty - 1) * g.in_ws + (4 * tx - 1);
    }
    const int t = min((A.mt0 + mt) * NT + wl1, A.T - 1);
    const int n = t / TT, r = t - n * TT;
    const int ty = r / A.TX, tx = r - ty * A.TX;
    const int yy = min(4 * ty - 1 + py, A.H);  // rows past the image: the zero gap row
    const size_t qq = (size_t)g.in_lead + (size_t)(n * g.in_hs + yy) * g.in_ws + (4 * tx - 1);
    const size_t o0 = q0 * cs_ld + in_ch0;
    r_ = make_rsrc(g.in + o0, g.in_bytes - o0 * 4);
    v_[0] = (unsigned)(((qq - q0) * cs_ld + cg1 * 4) * 4);
#pragma unroll
    for (int n5 = 3; n5 < 6; ++n5) v_[n5 - 2] = v_[0] + (unsigned)min(n5, A.W + 1 - 4 * tx) * pxb;  // columns past W: the gap column
  };
  F4 p[6];
  auto load_piece = [&](const i32x4& r_, const unsigned (&v_)[4], int chunk, int n5) {
    const unsigned cb = (unsigned)chunk * ckb;
#ifdef RTPOSE_EXP_W4_AUX  // cache policy bits of the patch loads (1 sc0, 2 nt, 16 sc1): no effect / nt 2x slower
    const f32x4 t = n5 < 3 ? llvm_raw_buffer_load_v4f32(r_, (int)v_[0], (int)(cb + n5 * pxb), RTPOSE_EXP_W4_AUX)
                           : llvm_raw_buffer_load_v4f32(r_, (int)v_[n5 - 2], (int)cb, RTPOSE_EXP_W4_AUX);
    p[n5] = F4{f2{t.x, t.y}, f2{t.z, t.w}};
#else
    p[n5] = n5 < 3 ? bload(r_, v_[0], cb + n5 * pxb) : bload(r_, v_[n5 - 2], cb);
#endif
  };
  const int ust = (py * 2 + cg1) * NT + (wl1 ^ (4 * cg1));     // U[fx][y = py][cg][wtile ^ 4 cg], + fx * 6 * 2 * NT
  const int uld = ((fx * 6) * 2 + cg) * NT + (wl ^ (4 * cg));  // U[fx][y][cg][wtile ^ 4 cg], + y * 2 * NT
  const int vst = ((fx * 3) * 4 + 2 * cg) * NT + wl;  // V[pair = fx * 3 + fy / 2][kq = 2 cg + h][wtile]
  auto stage1 = [&](int ubuf) {
    float4* dst = Ub(ubuf) + ust;
    bt6(p, [&](int f, F4 u) { dst[f * 6 * 2 * NT] = to_float4(u); });
  };
  F4 q[6];
  auto stage2_read = [&](int ubuf) {
    const float4* src = Ub(ubuf) + uld;
#pragma unroll
    for (int y = 0; y < 6; ++y) {
      const float4 t = src[y * 2 * NT];
      q[y] = F4{f2{t.x, t.y}, f2{t.z, t.w}};
    }
  };
  auto stage2 = [&](int vbuf) {
    float4* dst = Vb(vbuf) + vst;
    // frequencies fx * 6 + 2 i, + 1 share a 16-byte slot: channels (0, 1) -> plane 2 cg, (2, 3) -> plane 2 cg + 1
    F4 ev;
    bt6(q, [&](int f, F4 v) {
      if (f & 1) {
        dst[(f >> 1) * 4 * NT] = make_float4(ev.lo.x, ev.lo.y, v.lo.x, v.lo.y);
        dst[(f >> 1) * 4 * NT + NT] = make_float4(ev.hi.x, ev.hi.y, v.hi.x, v.hi.y);
      } else {
        ev = v;
      }
    });
  };

  // ---- MFMA roles ---------------------------------------------------------------------------------------------------
  // acc[l][0]: frequency 18 fh + l of the row tile this wave FINISHES (row tile fh), acc[l][1]: of the other row tile,
  // handed to the sibling (wave wv ^ 4) before the output transform
  const int ncol = nt * NC + wn * 16 + r16;
  floatx4 acc[NFW][RT];
  const float bias0 = g.bias[ncol];
  const unsigned boff = (unsigned)((kq * g.cout_pad + ncol) * 16);
  const unsigned fstep = (unsigned)(4 * g.cout_pad * 16);  // bytes per (chunk, frequency pair)
  const unsigned wbase = (unsigned)fh * (NPW * fstep);
  unsigned wso = wbase;
  float4 bs[NB];
  const int nchunks = A.cin / CK;  // even, >= 4 (host)
  const int aoff = fh * NPW * 4 * NT + kq * NT + r16;  // A: pair 9 fh + i, plane kq, wtile (row tile) * 16 + r16

  // ---- prologue: V[0] <- position 0, U[1] <- position 1, patch rows of position 2 in flight ---------------------------
  set_loader(j0, rin, pv);
#pragma unroll
  for (int i = 0; i < 6; ++i) load_piece(rin, pv, 0, i);
  if (s1) stage1(0);
#pragma unroll
  for (int i = 0; i < 6; ++i) load_piece(rin, pv, 1, i);
#pragma unroll
  for (int f = 0; f < PF; ++f) {
    bs[f] = bload_f4(rw, boff, wso);
    wso += fstep;
  }
  __syncthreads();
  stage2_read(0);
  if (s2) stage2(0);
  if (s1) stage1(1);
#pragma unroll
  for (int i = 0; i < 6; ++i) load_piece(rin, pv, 2, i);
  __syncthreads();

#define RTPOSE_PIN()             \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)
  for (int mt = j0; mt < A.mtiles; mt += jstep) {
#pragma unroll
    for (int f = 0; f < NFW; ++f)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[f][rt][v] = 0.f;

    float4 a0, a1;
#pragma unroll 1
    for (int c2 = 0; c2 < nchunks; c2 += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // chunk c2 + h multiplies from V[h]
        const float4* va = Vb(h) + aoff;
        // position + 3: the chunk whose patch rows are requested now (past the block's last tile: that tile again)
        if (lc == nchunks) {
          lc = 0;
          lt += jstep;
          if (lt < A.mtiles) set_loader(lt, rin, pv);
        }
        const int c3 = lc++;
        const i32x4 rl = s1 ? rin : rnull;
        if (mt == j0) RTPOSE_TSTAMP4(c2 + h, 0);
#if RTPOSE_EXP_W4_A3
        // A fragments in a ring of three: the fragment of half-step k + 2 (k = 2 i + row tile) is requested right after the
        // MFMAs of half-step k are issued - 8 MFMAs ahead instead of 4.  (Per-wave timelines, tools/timeline_w4.py: a wave
        // took 455..800 cycles per 256-cycle step, two LDS round trips per step on its critical path.)
        float4 ar[3];
        ar[0] = va[fh * 16];
        ar[1] = va[(fh ^ 1) * 16];
#else
        a0 = va[RT == 2 ? fh * 16 : 0];
        if (RT == 2) a1 = va[(fh ^ 1) * 16];
#endif
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
#if RTPOSE_EXP_W4_PRIO == 1  // the siblings of a SIMD take the matrix pipe's priority in turns, step by step
          if (fh == (i & 1)) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
#elif RTPOSE_EXP_W4_PRIO == 2  // the younger sibling (wave w + 4) always has it
          if (i == 0 && fh) __builtin_amdgcn_s_setprio(1);
#elif RTPOSE_EXP_W4_PRIO == 3  // in turns, chunk by chunk
          if (i == 0) {
            if (fh == h) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
          }
#endif
          const float4 bv = bs[(h * NPW + i) % NB];
#if RTPOSE_EXP_W4_A3
          const float4 a0 = ar[(2 * i) % 3], a1 = ar[(2 * i + 1) % 3];
#endif
          acc[2 * i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bv.x, acc[2 * i][0], 0, 0, 0);
          acc[2 * i + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bv.z, acc[2 * i + 1][0], 0, 0, 0);
          acc[2 * i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bv.y, acc[2 * i][0], 0, 0, 0);
          acc[2 * i + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bv.w, acc[2 * i + 1][0], 0, 0, 0);
          RTPOSE_PIN();
#if RTPOSE_EXP_W4_PRIO >= 4  // yield the matrix pipe to the sibling after every group of 4 MFMAs
          __builtin_amdgcn_s_sleep(RTPOSE_EXP_W4_PRIO - 3);
#endif
#if RTPOSE_EXP_W4_A3
          if (i < NPW - 1) ar[(2 * i + 2) % 3] = RTPOSE_EXP_A(va[(i + 1) * 4 * NT + fh * 16], ar[(2 * i) % 3]);
#else
          if (i < NPW - 1) a0 = RTPOSE_EXP_A(va[(i + 1) * 4 * NT + (RT == 2 ? fh * 16 : 0)], a0);
#endif
          {
            // B PF pairs ahead.  After a chunk's ninth pair the sibling's nine are skipped; after the tile's last chunk
            // the ring wraps to the next tile.
            const int L = h * NPW + i + PF;
            bs[L % NB] = RTPOSE_EXP_B(bload_f4(rw, boff, wso), bs[(h * NPW + i) % NB]);
            if (L % NPW == NPW - 1) wso = (c2 + L / NPW == nchunks - 1) ? wbase : wso + (NPW + 1) * fstep;
            else wso += fstep;
          }
          RTPOSE_PIN();
          if (RT == 2) {
            acc[2 * i][RT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bv.x, acc[2 * i][RT - 1], 0, 0, 0);
            acc[2 * i + 1][RT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, bv.z, acc[2 * i + 1][RT - 1], 0, 0, 0);
            acc[2 * i][RT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bv.y, acc[2 * i][RT - 1], 0, 0, 0);
            acc[2 * i + 1][RT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, bv.w, acc[2 * i + 1][RT - 1], 0, 0, 0);
          }
          RTPOSE_PIN();
#if RTPOSE_EXP_W4_PRIO >= 4
          __builtin_amdgcn_s_sleep(RTPOSE_EXP_W4_PRIO - 3);
#endif
#if RTPOSE_EXP_W4_A3
          if (i < NPW - 1) ar[(2 * i + 3) % 3] = RTPOSE_EXP_A(va[(i + 1) * 4 * NT + (fh ^ 1) * 16], ar[(2 * i + 1) % 3]);
#else
          if (RT == 2 && i < NPW - 1) a1 = RTPOSE_EXP_A(va[(i + 1) * 4 * NT + (fh ^ 1) * 16], a1);
#endif
          if (RTPOSE_EXP_STAGE) {
            if (i < 2) {
              if (s1 && i == turn) stage1(h);              // patch rows of position + 2 -> U[h]
            } else if (i < 4) {
              if (i == 2 + turn) stage2_read(h ^ 1);       // U[h ^ 1] = position + 1 (the waves 0, 1 read and discard: see the loads)
            } else if (i < 6) {
              if (s2 && i == 4 + turn) stage2(h ^ 1);      // -> V[h ^ 1]
            }
#ifndef RTPOSE_EXP_W4_NOLOAD  // (timing only: the transform on stale registers)
            if (i >= RTPOSE_EXP_W4_L0 && i < RTPOSE_EXP_W4_L0 + 3) {
              // Issued by ALL waves: the waves 6, 7 (no stage-1 item) go through a descriptor of zero extent - every lane is
              // out of range, nothing is fetched.  Under a wave-uniform branch the compiler's vmcnt bookkeeping has to assume
              // the loads were NOT issued: every later wait for a filter fragment then also waited for these patch loads,
              // one or two steps after their issue (0.92 -> see DESIGN.md §3.0).
              load_piece(rl, pv, c3, 2 * (i - RTPOSE_EXP_W4_L0));  // patch rows of position + 3
              load_piece(rl, pv, c3, 2 * (i - RTPOSE_EXP_W4_L0) + 1);
            }
#endif
          }
          RTPOSE_PIN();
        }
        if (mt == j0) RTPOSE_TSTAMP4(c2 + h, 1);
        __syncthreads();
      }
    }
    if (mt == j0) RTPOSE_TSTAMP4(63, 0);

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    // 1. every wave runs the first pass of the output transform (along y) on its own frequencies fx = 3 fh .. 3 fh + 2,
    //    for both row tiles;  2. the siblings (waves w, w ^ 4: same columns, the two frequency halves) swap the row
    //    sums of the row tile they do not finish, through the two LDS buffers that are dead at a tile boundary (V1, U0),
    //    in two rounds (one per wtile pair);  3. second pass (along x), + bias (+ReLU) (+2x2 max-pool), masked stores.
    // Register v of acc[l][0] = wtile fh * 16 + 4 kq + v of the tile, column ncol: 16 lanes = 64 contiguous bytes.
    f2 sk[3][RT][4], sr[3][RT][4];  // [fx - 3 fh][wtile pair][output row]: kept, received from the sibling
    if (RT == 2) {
      float4* const E = L4 + VBUF + UBUF;
#pragma unroll
      for (int vp = 0; vp < RT; ++vp) {
        f2 sg[3][4];
#pragma unroll
        for (int xl = 0; xl < 3; ++xl) {
          f2 m[6];
#pragma unroll
          for (int y = 0; y < 6; ++y) m[y] = f2{acc[xl * 6 + y][RT - 1][2 * vp], acc[xl * 6 + y][RT - 1][2 * vp + 1]};
          at4_lo(m, sg[xl][0], sg[xl][1]);
          at4_hi(m, sg[xl][2], sg[xl][3]);
        }
#pragma unroll
        for (int xl = 0; xl < 3; ++xl)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
            E[(wv * 6 + xl * 2 + hh) * 64 + lane] =
                make_float4(sg[xl][2 * hh].x, sg[xl][2 * hh].y, sg[xl][2 * hh + 1].x, sg[xl][2 * hh + 1].y);
#pragma unroll
        for (int xl = 0; xl < 3; ++xl) {
          f2 m[6];
#pragma unroll
          for (int y = 0; y < 6; ++y) m[y] = f2{acc[xl * 6 + y][0][2 * vp], acc[xl * 6 + y][0][2 * vp + 1]};
          at4_lo(m, sk[xl][vp][0], sk[xl][vp][1]);
          at4_hi(m, sk[xl][vp][2], sk[xl][vp][3]);
        }
        __syncthreads();
#pragma unroll
        for (int xl = 0; xl < 3; ++xl)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const float4 t = E[((wv ^ 4) * 6 + xl * 2 + hh) * 64 + lane];
            sr[xl][vp][2 * hh] = f2{t.x, t.y};
            sr[xl][vp][2 * hh + 1] = f2{t.z, t.w};
          }
        __syncthreads();
      }
    } else {
      // half tile: ONE row tile, split between the siblings by wtile pair - the wave keeps the pair fh (registers 2 fh, + 1 of
      // every accumulator) and sends the row sums of the other one.  The block has no next tile: every buffer is dead.
      float4* const E = L4;
      auto rowsums = [&](int pair, f2 (&dst)[3][RT][4]) {
#pragma unroll
        for (int xl = 0; xl < 3; ++xl) {
          f2 m[6];
#pragma unroll
          for (int y = 0; y < 6; ++y)
            m[y] = pair ? f2{acc[xl * 6 + y][0][2], acc[xl * 6 + y][0][3]} : f2{acc[xl * 6 + y][0][0], acc[xl * 6 + y][0][1]};
          at4_lo(m, dst[xl][0][0], dst[xl][0][1]);
          at4_hi(m, dst[xl][0][2], dst[xl][0][3]);
        }
      };
      rowsums(fh ^ 1, sr);  // (sr as scratch: the sums that leave)
#pragma unroll
      for (int xl = 0; xl < 3; ++xl)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
          E[(wv * 6 + xl * 2 + hh) * 64 + lane] =
              make_float4(sr[xl][0][2 * hh].x, sr[xl][0][2 * hh].y, sr[xl][0][2 * hh + 1].x, sr[xl][0][2 * hh + 1].y);
      rowsums(fh, sk);
      __syncthreads();
#pragma unroll
      for (int xl = 0; xl < 3; ++xl)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const float4 t = E[((wv ^ 4) * 6 + xl * 2 + hh) * 64 + lane];
          sr[xl][0][2 * hh] = f2{t.x, t.y};
          sr[xl][0][2 * hh + 1] = f2{t.z, t.w};
        }
    }
    auto finish = [&](auto fhc) {
      constexpr int FH = decltype(fhc)::value;
      // row sums of fx = 0..5: the wave's own 3 FH .. 3 FH + 2, the sibling's from sr
      auto S = [&](int x, int vp, int i) -> f2 { return (x / 3 == FH) ? sk[x % 3][vp][i] : sr[x % 3][vp][i]; };
      const bool col_ok = ncol < g.cout;
      const int sc = A.pool ? 2 : 4;
      auto wt_q = [&](int n, int ty, int tx) -> int { return (n * g.out_hs + sc * ty) * g.out_ws + sc * tx; };
      int q0;
      {
        const int t = min((A.mt0 + mt) * NT, A.T - 1);
        const int n = t / TT, r = t - n * TT;
        const int ty = r / A.TX;
        q0 = wt_q(n, ty, r - ty * A.TX);
      }
      const int cs_st = px_floats(g.out_cstride, g.out_pq);
      const size_t oo0 = ((size_t)g.out_lead + (size_t)q0) * cs_st + ch0_floats(g.out_choff, g.out_pq);
      const i32x4 rout = make_rsrc(g.out + oo0, g.out_bytes - oo0 * 4);
      const unsigned cs4 = (unsigned)cs_st * 4, row4 = (unsigned)g.out_ws * cs4;
      const unsigned col4 = col_floats(ncol, g.out_pq) * 4;
      int tcur = (A.mt0 + mt) * NT + (RT == 2 ? FH * 16 + 4 * kq : 4 * kq + 2 * FH);
      int sn = tcur / TT, sy, sx;
      {
        const int r = tcur - sn * TT;
        sy = r / A.TX;
        sx = r - sy * A.TX;
      }
#pragma unroll
      for (int vp = 0; vp < RT; ++vp) {
        // geometry of the two wtiles of the pair
        unsigned off[2];
        bool okv[2];
        int ylim[2], xlim[2];  // valid output rows / columns of the wtile (<= 4)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          okv[e] = col_ok && tcur < A.T;
          off[e] = (unsigned)(wt_q(sn, sy, sx) - q0) * cs4 + col4;
          ylim[e] = A.H - 4 * sy;
          xlim[e] = A.W - 4 * sx;
          ++tcur;
          const bool wx = sx + 1 >= A.TX, wy = wx && sy + 1 >= A.TY;
          sx = wx ? 0 : sx + 1;
          sy = wy ? 0 : (wx ? sy + 1 : sy);
          sn += wy ? 1 : 0;
        }
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
          // second pass: along x (fx -> output column j) for the output rows 2 ih, 2 ih + 1
          f2 yy[2][4];
#pragma unroll
          for (int il = 0; il < 2; ++il) {
            f2 m[6];
#pragma unroll
            for (int x = 0; x < 6; ++x) m[x] = S(x, vp, 2 * ih + il);
            at4_lo(m, yy[il][0], yy[il][1]);
            at4_hi(m, yy[il][2], yy[il][3]);
          }
          if (A.pool) {
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
              f2 mx = __builtin_elementwise_max(__builtin_elementwise_max(yy[0][2 * jh], yy[0][2 * jh + 1]),
                                                __builtin_elementwise_max(yy[1][2 * jh], yy[1][2 * jh + 1]));
              mx = mx + splat(bias0);
              if (A.relu) mx = __builtin_elementwise_max(mx, splat(0.f));
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                const bool ok = okv[e] && 2 * ih < ylim[e] && 2 * jh < xlim[e];
                bstore(mx[e], rout, ok ? off[e] : kNoStore, ih * row4 + jh * cs4);
              }
            }
          } else {
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int i = 2 * ih + il;
                f2 o = yy[il][j] + splat(bias0);
                if (A.relu) o = __builtin_elementwise_max(o, splat(0.f));
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                  const bool ok = okv[e] && i < ylim[e] && j < xlim[e];
                  bstore(o[e], rout, ok ? off[e] : kNoStore, i * row4 + j * cs4);
                }
              }
          }
        }
      }
    };
    if (fh == 0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
    if (mt == j0) RTPOSE_TSTAMP4(63, 1);
  }  // m tiles of this block
#undef RTPOSE_PIN
}

// Small grids (few images): the same arithmetic on 8x as many blocks.  A block = 6 waves = 16 wtiles x 16 columns;
// wave w multiplies the six frequencies fx = w (3 pairs x 4 MFMAs per chunk), the waves 0..2 run stage 1 of the input
// transform (two patch rows each), the waves 3..5 stage 2 (two fx each).  Every frequency still sums over the chunks
// and the two channel halves of a chunk in wino4_f32's order, the transforms are the same functions of the same
// values: bit-identical results.  Before the output transform the waves exchange their row sums through LDS; the
// waves 0..3 finish one (wtile pair, output row pair) each.  One 368 x 368 image, conv3_2: 34 x 16 = 544 blocks of
// 74 KB LDS (two per CU) instead of 17 x 4 = 68.
constexpr int NTS = 16;                   // wtiles per block
constexpr int VBUFS = NFP * 4 * NTS;      // float4 per V buffer: [pair][kq][wtile]
constexpr int UBUFS = 36 * 2 * NTS;       // float4 per U buffer: [fx][y][channel group][wtile ^ 4 cg]

__global__ __launch_bounds__(384, 2) void wino4s_f32(const Args A) {
  extern __shared__ __attribute__((aligned(16))) float4 L4[];
  auto Vb = [&](int i) -> float4* { return L4 + i * VBUFS; };
  auto Ub = [&](int i) -> float4* { return L4 + 2 * VBUFS + i * UBUFS; };
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..5: the frequencies fx = wv
  const int r16 = lane & 15, kq = lane >> 4;

  // block -> (m tile, 16-column tile, group)
  const int bi = blockIdx.x;
  const int mt = A.mt0 + bi % A.mtiles, c = bi / A.mtiles;
  const int nt = c % A.ntiles, grp = c / A.ntiles;
  const Group g = grp ? A.g[1] : A.g[0];
  const int TT = A.TY * A.TX;
  const int nchunks = A.cin / CK;  // even, >= 4 (host)

  // ---- transform roles ----------------------------------------------------------------------------------------------
  const bool s1 = wv < 3;
  const int hi5 = lane >> 5;
  const int wl1 = (lane >> 1) & 15, cg1 = lane & 1, py = 2 * (s1 ? wv : 0) + hi5;  // stage 1: (wtile, channel group, patch row)
  const int wl = lane & 15, cg = (lane >> 4) & 1, fx = 2 * (s1 ? 0 : wv - 3) + hi5;  // stage 2: (wtile, channel group, fx)
  const i32x4 rw = make_rsrc(g.w, g.w_bytes);
  i32x4 rin;
  unsigned pv0, pvx;  // byte offset of pixel 0 of the row; pixels of the row before the gap column (W + 1 - 4 tx)
  const int cs_ld = px_floats(g.in_cstride, g.in_pq);
  const unsigned pxb = (unsigned)cs_ld * 4;
  const unsigned ckb = g.in_pq ? (unsigned)g.in_pq * (CK * 4) : CK * 4;
  {
    size_t q0;
    {
      const int t = min(mt * NTS, A.T - 1);
      const int n = t / TT, r = t - n * TT;
      const int ty = r / A.TX, tx = r - ty * A.TX;
      q0 = (size_t)g.in_lead + (size_t)(n * g.in_hs + 4 * ty - 1) * g.in_ws + (4 * tx - 1);
    }
    const int t = min(mt * NTS + wl1, A.T - 1);
    const int n = t / TT, r = t - n * TT;
    const int ty = r / A.TX, tx = r - ty * A.TX;
    const int yy = min(4 * ty - 1 + py, A.H);  // rows past the image: the zero gap row
    const size_t qq = (size_t)g.in_lead + (size_t)(n * g.in_hs + yy) * g.in_ws + (4 * tx - 1);
    const size_t o0 = q0 * cs_ld + ch0_floats(g.in_choff, g.in_pq);
    rin = make_rsrc(g.in + o0, s1 ? g.in_bytes - o0 * 4 : 0);
    pv0 = (unsigned)(((qq - q0) * cs_ld + cg1 * 4) * 4);
    pvx = (unsigned)(A.W + 1 - 4 * tx);
  }
  F4 p[6];
  auto load_piece = [&](int chunk, int n5) {
    const unsigned cb = (unsigned)min(chunk, nchunks - 1) * ckb;  // (past the last chunk: that chunk again)
    p[n5] = n5 < 3 ? bload(rin, pv0, cb + n5 * pxb) : bload(rin, pv0 + min((unsigned)n5, pvx) * pxb, cb);
  };
  // (SQ_LDS_BANK_CONFLICT is 14 % of SQ_LDS_IDX_ACTIVE in this kernel - 0 in wino4_f32 - and a further swizzle of the
  //  two fx a stage-2 wave reads did not change it; the kernel is ~1 % of a batch-32 forward)
  const int ust = (py * 2 + cg1) * NTS + (wl1 ^ (4 * cg1));     // U[fx][y = py][cg][wtile ^ 4 cg], + fx * 6 * 2 * NTS
  const int uld = ((fx * 6) * 2 + cg) * NTS + (wl ^ (4 * cg));  // U[fx][y][cg][wtile ^ 4 cg], + y * 2 * NTS
  const int vst = ((fx * 3) * 4 + 2 * cg) * NTS + wl;           // V[pair = fx * 3 + fy / 2][kq = 2 cg + h][wtile]
  auto stage1 = [&](int ubuf) {
    float4* dst = Ub(ubuf) + ust;
    bt6(p, [&](int f, F4 u) { dst[f * 6 * 2 * NTS] = to_float4(u); });
  };
  F4 q[6];
  auto stage2_read = [&](int ubuf) {
    const float4* src = Ub(ubuf) + uld;
#pragma unroll
    for (int y = 0; y < 6; ++y) {
      const float4 t = src[y * 2 * NTS];
      q[y] = F4{f2{t.x, t.y}, f2{t.z, t.w}};
    }
  };
  auto stage2 = [&](int vbuf) {
    float4* dst = Vb(vbuf) + vst;
    F4 ev;
    bt6(q, [&](int f, F4 v) {
      if (f & 1) {
        dst[(f >> 1) * 4 * NTS] = make_float4(ev.lo.x, ev.lo.y, v.lo.x, v.lo.y);
        dst[(f >> 1) * 4 * NTS + NTS] = make_float4(ev.hi.x, ev.hi.y, v.hi.x, v.hi.y);
      } else {
        ev = v;
      }
    });
  };

  // ---- MFMA role: frequencies wv * 6 .. wv * 6 + 5 = pairs 3 wv .. 3 wv + 2, 16 wtiles x 16 columns --------------------
  const int ncol = nt * 16 + r16;
  floatx4 acc[6];
#pragma unroll
  for (int f = 0; f < 6; ++f)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[f][v] = 0.f;
  const unsigned boff = (unsigned)((kq * g.cout_pad + ncol) * 16);
  const unsigned fstep = (unsigned)(4 * g.cout_pad * 16);  // bytes per (chunk, frequency pair)
  unsigned wso = (unsigned)wv * 3 * fstep;
  float4 bs[6];  // the pairs of two chunks
  const int aoff = wv * 3 * 4 * NTS + kq * NTS + r16;

  // ---- prologue: V[0] <- chunk 0, U[1] <- chunk 1, patch rows of chunk 2 in flight ---------------------------------------
#pragma unroll
  for (int i = 0; i < 6; ++i) load_piece(0, i);
  if (s1) stage1(0);
#pragma unroll
  for (int i = 0; i < 6; ++i) load_piece(1, i);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    bs[i] = bload_f4(rw, boff, wso);
    wso += fstep;
  }
  wso += (NFP - 3) * fstep;
  __syncthreads();
  if (!s1) {
    stage2_read(0);
    stage2(0);
  }
  if (s1) stage1(1);
#pragma unroll
  for (int i = 0; i < 6; ++i) load_piece(2, i);
  __syncthreads();

#pragma unroll 1
  for (int c2 = 0; c2 < nchunks; c2 += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // chunk c2 + h multiplies from V[h]
      const float4* va = Vb(h) + aoff;
      // the filter pairs of the next chunk (past the last one: clamped by the descriptor, unused)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        bs[((h ^ 1) * 3) + i] = bload_f4(rw, boff, wso);
        wso += fstep;
      }
      wso += (NFP - 3) * fstep;
      // transform work of this chunk period: stage 1 of chunk + 2 -> U[h], stage 2 of chunk + 1: U[h ^ 1] -> V[h ^ 1]
      // (the patch loads are issued by all six waves - the waves 3..5 through a descriptor of zero extent -, not under the
      //  wave-uniform branch: the compiler's vmcnt bookkeeping would have to assume they were not issued, see wino4_f32)
      if (s1) stage1(h);
      else stage2_read(h ^ 1);
#pragma unroll
      for (int i = 0; i < 6; ++i) load_piece(c2 + h + 3, i);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float4 av = va[i * 4 * NTS], bv = bs[h * 3 + i];
        acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[2 * i], 0, 0, 0);
        acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[2 * i + 1], 0, 0, 0);
        acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[2 * i], 0, 0, 0);
        acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[2 * i + 1], 0, 0, 0);
      }
      if (!s1) stage2(h ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue: row sums (first pass of the output transform, along y) of the wave's fx -> LDS; the waves 0..3 run the
  // second pass for one (wtile pair, output row pair) each ----------------------------------------------------------------
  {
    float4* const E = L4;  // [fx][wtile pair][row pair][lane]
#pragma unroll
    for (int vp = 0; vp < 2; ++vp) {
      f2 m[6], sg[4];
#pragma unroll
      for (int y = 0; y < 6; ++y) m[y] = f2{acc[y][2 * vp], acc[y][2 * vp + 1]};
      at4_lo(m, sg[0], sg[1]);
      at4_hi(m, sg[2], sg[3]);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
        E[((wv * 2 + vp) * 2 + hh) * 64 + lane] = make_float4(sg[2 * hh].x, sg[2 * hh].y, sg[2 * hh + 1].x, sg[2 * hh + 1].y);
    }
    __syncthreads();
    if (wv < 4) {
      const int vp = wv >> 1, ih = wv & 1;
      f2 sl[6][2];
#pragma unroll
      for (int x = 0; x < 6; ++x) {
        const float4 t = E[((x * 2 + vp) * 2 + ih) * 64 + lane];
        sl[x][0] = f2{t.x, t.y};
        sl[x][1] = f2{t.z, t.w};
      }
      const bool col_ok = ncol < g.cout;
      const float bias0 = g.bias[ncol];
      const int sc = A.pool ? 2 : 4;
      auto wt_q = [&](int n, int ty, int tx) -> int { return (n * g.out_hs + sc * ty) * g.out_ws + sc * tx; };
      int q0;
      {
        const int t = min(mt * NTS, A.T - 1);
        const int n = t / TT, r = t - n * TT;
        const int ty = r / A.TX;
        q0 = wt_q(n, ty, r - ty * A.TX);
      }
      const int cs_st = px_floats(g.out_cstride, g.out_pq);
      const size_t oo0 = ((size_t)g.out_lead + (size_t)q0) * cs_st + ch0_floats(g.out_choff, g.out_pq);
      const i32x4 rout = make_rsrc(g.out + oo0, g.out_bytes - oo0 * 4);
      const unsigned cs4 = (unsigned)cs_st * 4, row4 = (unsigned)g.out_ws * cs4;
      const unsigned col4 = col_floats(ncol, g.out_pq) * 4;
      unsigned off[2];
      bool okv[2];
      int ylim[2], xlim[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int tcur = mt * NTS + 4 * kq + 2 * vp + e;
        const int sn = tcur / TT, r = tcur - sn * TT;
        const int sy = r / A.TX, sx = r - sy * A.TX;
        okv[e] = col_ok && tcur < A.T;
        off[e] = (unsigned)(wt_q(sn, sy, sx) - q0) * cs4 + col4;
        ylim[e] = A.H - 4 * sy;
        xlim[e] = A.W - 4 * sx;
      }
      f2 yy[2][4];
#pragma unroll
      for (int il = 0; il < 2; ++il) {
        f2 m[6];
#pragma unroll
        for (int x = 0; x < 6; ++x) m[x] = sl[x][il];
        at4_lo(m, yy[il][0], yy[il][1]);
        at4_hi(m, yy[il][2], yy[il][3]);
      }
      if (A.pool) {
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
          f2 mx = __builtin_elementwise_max(__builtin_elementwise_max(yy[0][2 * jh], yy[0][2 * jh + 1]),
                                            __builtin_elementwise_max(yy[1][2 * jh], yy[1][2 * jh + 1]));
          mx = mx + splat(bias0);
          if (A.relu) mx = __builtin_elementwise_max(mx, splat(0.f));
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const bool ok = okv[e] && 2 * ih < ylim[e] && 2 * jh < xlim[e];
            bstore(mx[e], rout, ok ? off[e] : kNoStore, ih * row4 + jh * cs4);
          }
        }
      } else {
#pragma unroll
        for (int il = 0; il < 2; ++il)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = 2 * ih + il;
            f2 o = yy[il][j] + splat(bias0);
            if (A.relu) o = __builtin_elementwise_max(o, splat(0.f));
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const bool ok = okv[e] && i < ylim[e] && j < xlim[e];
              bstore(o[e], rout, ok ? off[e] : kNoStore, i * row4 + j * cs4);
            }
          }
      }
    }
  }
}

// ---- weight packing: U = G g G^T for the points 0, +-3/4, +-3/2, inf (double arithmetic, one rounding) -------------
// packed[chunk][pair][kq][cout_pad][f2][e]: frequency f = fx * 6 + fy = 2 pair + f2, channel chunk * 8 + 2 kq + e
__device__ __forceinline__ double g43(int f, int k) {
  // rows of G (the B^T rows above are scaled by N_f, G divides by it)
  const double G[6][3] = {{64.0 / 81.0, 0.0, 0.0},
                          {-128.0 / 243.0, -32.0 / 81.0, -8.0 / 27.0},
                          {-128.0 / 243.0, 32.0 / 81.0, -8.0 / 27.0},
                          {32.0 / 243.0, 16.0 / 81.0, 8.0 / 27.0},
                          {32.0 / 243.0, -16.0 / 81.0, 8.0 / 27.0},
                          {0.0, 0.0, 1.0}};
  return G[f][k];
}
__device__ __forceinline__ double u43(const float* gw, int fxi, int fyi) {
  double v = 0.0;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) v += g43(fyi, ky) * g43(fxi, kx) * (double)gw[ky * 3 + kx];
  return v;
}

__global__ void pack_wino4_kernel(const float* __restrict__ w, const float* __restrict__ bias, int cout, int cin_src,
                                  const int32_t* __restrict__ cin_map, int cin_packed, int coutp,
                                  float* __restrict__ wp, float* __restrict__ bp) {
  const size_t total = (size_t)36 * cin_packed * coutp;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)coutp) bp[i] = (i < (size_t)cout && bias) ? bias[i] : 0.f;
  if (i >= total) return;
  const int e = i & 1, fh = (i >> 1) & 1;
  size_t r = i >> 2;
  const int n = r % coutp;
  r /= coutp;
  const int kq = r & 3;
  r >>= 2;
  const int fp = r % NFP;
  const int chunk = r / NFP;
  const int cc = chunk * CK + 2 * kq + e;
  const int src = cin_map ? cin_map[cc] : (cc < cin_src ? cc : -1);
  float v = 0.f;
  if (n < cout && src >= 0 && src < cin_src) {
    const int f = 2 * fp + fh;
    v = (float)u43(w + ((size_t)n * cin_src + src) * 9, f / 6, f % 6);
  }
  wp[i] = v;
}

// amplification estimate of a 3x3 filter bank in F(4x4,3x3) (the definition: conv_wino7.hip, wino_amp_kernel)
__global__ void wino4_amp_kernel(const float* __restrict__ w, int cout, int cin, float* __restrict__ amp) {
  __shared__ float red[37][256];
  const int o = blockIdx.x, tid = threadIdx.x;
  float sf[36], den = 0.f;
#pragma unroll
  for (int f = 0; f < 36; ++f) sf[f] = 0.f;
  for (int cidx = tid; cidx < cin; cidx += 256) {
    const float* gw = w + ((size_t)o * cin + cidx) * 9;
    for (int k = 0; k < 9; ++k) den += fabsf(gw[k]);
#pragma unroll
    for (int f = 0; f < 36; ++f) sf[f] += fabsf((float)u43(gw, f / 6, f % 6));
  }
#pragma unroll
  for (int f = 0; f < 36; ++f) red[f][tid] = sf[f];
  red[36][tid] = den;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st)
      for (int f = 0; f < 37; ++f) red[f][tid] += red[f][tid + st];
    __syncthreads();
  }
  if (tid == 0) {
    const float at[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f},
                            {0.f, 0.75f, 0.75f, 1.5f, 1.5f, 0.f},
                            {0.f, 0.5625f, 0.5625f, 2.25f, 2.25f, 0.f},
                            {0.f, 0.421875f, 0.421875f, 3.375f, 3.375f, 1.f}};      // |A^T|
    const float bsum[6] = {5.078125f, 5.6875f, 5.6875f, 3.90625f, 3.90625f, 5.078125f};  // sum_n |B^T[f][n]|
    float num = 0.f;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        float v = 0.f;
        for (int fxi = 0; fxi < 6; ++fxi)
          for (int fyi = 0; fyi < 6; ++fyi) v += at[i][fyi] * at[j][fxi] * bsum[fyi] * bsum[fxi] * red[fxi * 6 + fyi][0];
        num = fmaxf(num, v);
      }
    const float r = red[36][0] > 0.f ? num / red[36][0] : 0.f;
    atomicMax(reinterpret_cast<int*>(amp), __float_as_int(r));
  }
}

}  // namespace wino4

#ifdef RTPOSE_EXP_TIMELINE4
static unsigned long long* g_dbgw4_buf = nullptr;
static unsigned g_dbgw4_blocks = 0;
#endif

// 1 when the 3x3 conv has an F(4x4,3x3) instance: 8-channel chunks, at least 3 of them (the transform pipeline is 3 deep)
int conv2d_wino4_ok(int cin, int cout) { return cout > 0 && cin % 16 == 0 && cin >= 32; }

size_t packed_weight_floats_wino4(int cout, int cin) { return (size_t)36 * cin * cout_pad(cout); }

int conv2d_wino4_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, hipStream_t s) {
  using namespace wino4;
  if (!d || ngroups < 1 || ngroups > 2) return fail(RTPOSE_E_INVAL, "conv2d_winograd (4x4): ngroups must be 1 or 2");
  const rtpose_conv_desc& d0 = d[0];
  if (d0.k != 3 || !conv2d_wino4_ok(d0.cin, d0.cout))
    return fail(RTPOSE_E_INVAL, "conv2d_winograd (4x4): k must be 3 and cin a multiple of 16, >= 32");
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
    if (di.in_plane_pixels < 0 || di.out_plane_pixels < 0 ||
        (di.in_plane_pixels && ((di.lin.choff % 8) || (size_t)di.in_plane_pixels < rtpose_layout_pixels(&di.lin, N, H, W))) ||
        (di.out_plane_pixels &&
         ((di.lout.choff % 8) || (di.cout % 8) ||
          (size_t)di.out_plane_pixels < rtpose_layout_pixels(&di.lout, N, di.pool ? H / 2 : H, di.pool ? W / 2 : W))))
      return fail(RTPOSE_E_INVAL, "conv2d_winograd (4x4): channel-plane slices start at a multiple of 8 channels (the output "
                                  "has a multiple of 8 of them) and a plane holds at least the layout's pixels");
    // per-lane offsets are 32 bits and the descriptors clamp at 2 GiB: a plane slice spans (channels / 8) planes
    if ((di.in_plane_pixels && (size_t)(di.lin.choff + di.cin) / 8 * di.in_plane_pixels * 32 > (size_t)winoc::kMaxRange) ||
        (di.out_plane_pixels && (size_t)(di.lout.choff + cout_pad(di.cout)) / 8 * di.out_plane_pixels * 32 > (size_t)winoc::kMaxRange))
      return fail(RTPOSE_E_INVAL, "conv2d_winograd (4x4): channel-plane slice beyond 2 GiB");
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
    g.in_pq = di.in_plane_pixels;
    g.out_pq = di.out_plane_pixels;
    // bytes addressable from the buffer base: the whole pixel-major tensor, or the planes up to the slice's last one
    g.in_bytes = g.in_pq ? (size_t)(di.lin.choff + di.cin) / 8 * g.in_pq * 32
                         : rtpose_layout_pixels(&di.lin, N, H, W) * (size_t)di.lin.cstride * sizeof(float);
    g.w_bytes = packed_weight_floats_wino4(di.cout, di.cin) * sizeof(float);
    g.out_bytes = g.out_pq ? (size_t)(di.lout.choff + di.cout) / 8 * g.out_pq * 32
                           : rtpose_layout_pixels(&di.lout, N, di.pool ? H / 2 : H, di.pool ? W / 2 : W) *
                                 (size_t)di.lout.cstride * sizeof(float);
  }
  a.N = N;
  a.H = H;
  a.W = W;
  a.TY = ceil_div(H, 4);
  a.TX = ceil_div(W, 4);
  const long T = (long)N * a.TY * a.TX;
  if (T > 0x7fffffffL) return fail(RTPOSE_E_INVAL, "conv2d_winograd: tensor too large");
  a.T = (int)T;
  a.cin = d0.cin;
  a.relu = d0.relu;
  a.pool = d0.pool;
  a.mtiles = ceil_div(a.T, NT);
  a.ntiles = cout_pad(d0.cout) / NC;
  a.ncombo = a.ntiles * ngroups;
  a.xcd_remap = (a.ncombo > 1 && a.mtiles >= 64) ? 1 : 0;
  long ids = a.xcd_remap ? (long)8 * a.ncombo * ceil_div(a.mtiles, 8) : (long)a.mtiles * a.ncombo;
  if (ids > 0x7fffffffL) return fail(RTPOSE_E_INVAL, "conv2d_winograd: grid too large");
  const int n_cu = device_cu_count();
  auto launch_small = [&](const Args& a0, int mt0_32, long wtiles) -> int {
    // the 16 x 16 form (wino4s_f32), bit-identical, 8x the blocks: `wtiles` wtiles from m tile mt0_32 (in 32-wtile units) on
    Args b = a0;
    b.persist = 0;
    b.mt0 = 2 * mt0_32;
    b.mtiles = (int)((wtiles + NTS - 1) / NTS);
    b.ntiles = cout_pad(d0.cout) / 16;
    b.ncombo = b.ntiles * ngroups;
    static PerDeviceOnce attr_s;
    const int dev_s = current_device();
    if (!attr_s.is_set(dev_s)) {
      RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino4s_f32),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      attr_s.set(dev_s);
    }
    hipLaunchKernelGGL(wino4s_f32, dim3((unsigned)((long)b.mtiles * b.ncombo)), dim3(384),
                       (size_t)(2 * VBUFS + 2 * UBUFS) * sizeof(float4), s, b);
    RTPOSE_HIP_CHECK(hipGetLastError());
    return 0;
  };
  auto launch_half = [&](const Args& a0, int mt0_32, long wtiles) -> int {
    // half tiles (wino4_f32<1>: 16 wtiles x 64 columns, one per block), bit-identical: `wtiles` wtiles from m tile mt0_32 on
    Args b = a0;
    b.persist = 0;
    b.mt0 = 2 * mt0_32;
    b.mtiles = (int)((wtiles + 15) / 16);
    b.xcd_remap = (b.ncombo > 1 && b.mtiles >= 64) ? 1 : 0;
#ifdef RTPOSE_EXP_TIMELINE4
    b.dbg = nullptr;  // (the stamps are the main launch's)
#endif
    const long idh = b.xcd_remap ? (long)8 * b.ncombo * ceil_div(b.mtiles, 8) : (long)b.mtiles * b.ncombo;
    static PerDeviceOnce attr_h;
    const int dev_h = current_device();
    if (!attr_h.is_set(dev_h)) {
      RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_f32<1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      attr_h.set(dev_h);
    }
    hipLaunchKernelGGL(wino4_f32<1>, dim3((unsigned)idh), dim3(512), (size_t)(VBUF + UBUF) * sizeof(float4), s, b);
    RTPOSE_HIP_CHECK(hipGetLastError());
    return 0;
  };
  // launches that fill at most half the CUs with 32 x 64 tiles run the 16 x 16 form (measured: at one round and beyond
  // the big tiles win - the small form fetches the filters four times as often; tools/r3_sessions/session27.sh)
  if ((long)a.mtiles * a.ncombo * 2 <= n_cu) return launch_small(a, 0, a.T);
  int rest = 0;  // m tiles left to a second launch
  bool rest_half = false;  // ... in half tiles (else the 16 x 16 form)
  if ((long)a.mtiles * a.ncombo > n_cu && n_cu % (8 * a.ncombo) == 0) {
    // Persistent blocks: block (column tile / group c, q) takes the m tiles q, q + Pc, ...  When the tiles do not come out
    // as whole rounds and what is left over is at most a quarter of a round, the whole rounds run here and the rest as a
    // second launch in the 16 x 16 form: 2 rounds + a short launch instead of 3 (conv4_3_CPM, the stage-1 convs: 0.60 ->
    // 0.53 ms, 0.17 -> 0.155), 1 + a short one instead of 2 (conv4_4_CPM: 0.21 -> 0.15).  Half a round left over is
    // cheaper as a half-empty round of big tiles (conv4_1 / conv4_2: 0.51 -> 0.54 ms with the cut), and at 8.27 rounds
    // (conv3_x) the cut changes nothing.  The forms are bit-identical, so the cut is invisible in the results.
    const int Pc = n_cu / a.ncombo;
    const int r = a.mtiles % Pc;
    static int cut_pct = -1;  // developer switch: largest left-over (in % of a round) that is cut off
    if (cut_pct < 0) {
      const char* e = dev_env("RTPOSE_W4_CUT_PCT");
      cut_pct = e ? atoi(e) : 25;
    }
    static int half_on = -1;  // developer switch: RTPOSE_W4_HALF=0 runs a left-over of 25..50 % of a round as a round of big tiles
    if (half_on < 0) {
      const char* e = dev_env("RTPOSE_W4_HALF");
      half_on = e ? atoi(e) : 1;
    }
    if (r && (long)r * a.ncombo * 100 <= (long)cut_pct * n_cu) {
      rest = r;
      a.mtiles -= r;
    } else if (r && half_on && 2L * r * a.ncombo <= n_cu) {
      // Round 4: up to half a round left over runs as ONE round of half tiles (16 wtiles x 64 columns: half the multiplies of a
      // tile, the same transform work per wtile) instead of a round of big tiles that leaves 50..75 % of the CUs idle:
      // conv3_x 8.27 rounds -> 8 + a half-tile round, conv4_1 / conv4_2 4.5 -> 4 + one.
      rest = r;
      rest_half = true;
      a.mtiles -= r;
    }
    a.persist = 1;
    ids = n_cu;
  }
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_f32<2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set.set(dev);
  }
#ifdef RTPOSE_EXP_TIMELINE4
  {  // developer build: stamps of the LAST launch with the channel counts RTPOSE_TIMELINE_W4="cin,cout" (default 256,256)
    static int want_cin = -1, want_cout = 256;
    if (want_cin < 0) {
      const char* e = dev_env("RTPOSE_TIMELINE_W4");
      want_cin = 256;
      if (e) sscanf(e, "%d,%d", &want_cin, &want_cout);
    }
    a.dbg = nullptr;
    if (d0.cin == want_cin && d0.cout == want_cout && ids <= 1024) {
      if (!g_dbgw4_buf) (void)hipMalloc(&g_dbgw4_buf, (size_t)1024 * 8 * 64 * 2 * 8);
      (void)hipMemsetAsync(g_dbgw4_buf, 0, (size_t)ids * 8 * 64 * 2 * 8, s);
      a.dbg = g_dbgw4_buf;
      g_dbgw4_blocks = (unsigned)ids;
    }
  }
#endif
  hipLaunchKernelGGL(wino4_f32<2>, dim3((unsigned)ids), dim3(512), (size_t)(2 * VBUF + 2 * UBUF) * sizeof(float4), s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  if (rest && rest_half) return launch_half(a, a.mtiles, (long)a.T - (long)a.mtiles * NT);
  if (rest) return launch_small(a, a.mtiles, (long)a.T - (long)a.mtiles * NT);
  return 0;
}

int pack_weights_wino4_launch(const float* w, const float* bias, int cout, int cin_src, const int32_t* cin_map,
                              int cin_packed, float* wp, float* bp, hipStream_t s) {
  if (!conv2d_wino4_ok(cin_packed, cout) || (cin_packed < cin_src && !cin_map))
    return fail(RTPOSE_E_INVAL, "pack_winograd (4x4): cin_packed must be a multiple of 16, >= 32 and >= cin_src");
  const int coutp = cout_pad(cout);
  const size_t total = (size_t)36 * cin_packed * coutp;
  const int threads = 256;
  const unsigned blocks = (unsigned)((total + threads - 1) / threads);
  hipLaunchKernelGGL(wino4::pack_wino4_kernel, dim3(blocks), dim3(threads), 0, s, w, bias, cout, cin_src, cin_map,
                     cin_packed, coutp, wp, bp);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int wino4_amplification_launch(const float* w, int cout, int cin, float* amp, hipStream_t s) {
  hipLaunchKernelGGL(wino4::wino4_amp_kernel, dim3(cout), dim3(256), 0, s, w, cout, cin, amp);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

// MFMA flops a launch ISSUES: whole tiles of 32 wtiles x all 36 frequencies
double conv2d_wino4_issued_flops(int cin, int cout, int N, int H, int W) {
  const double T = (double)N * ceil_div(H, 4) * ceil_div(W, 4);
  const double tiles = std::ceil(T / 32.0);
  return 2.0 * tiles * 32.0 * 36.0 * (double)cin * cout_pad(cout);
}

}  // namespace rtpose

#ifdef RTPOSE_EXP_TIMELINE4
extern "C" int rtpose_debug_timeline_w4_dump(unsigned long long* host, unsigned cap_blocks) {
  using namespace rtpose;
  if (!g_dbgw4_buf) return 0;
  (void)hipDeviceSynchronize();
  const unsigned n = g_dbgw4_blocks < cap_blocks ? g_dbgw4_blocks : cap_blocks;
  (void)hipMemcpy(host, g_dbgw4_buf, (size_t)n * 8 * 64 * 2 * 8, hipMemcpyDeviceToHost);
  return (int)n;
}
#endif
