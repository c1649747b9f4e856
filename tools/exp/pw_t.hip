// EXPERIMENT (round 4, not part of the library; measurements: profiles/r04_shufflenet_pw_t_experiment.txt).
// To try it: copy into csrc/, add to the Makefile's SRCS, declare rtpose_pw_fused_t[_fits] and route O_PWF launches of
// csrc/shufflenet.hip through pw_t_fits / pw_t_launch.
//
// Wave-autonomous, transposed form of the fused pointwise chain of the ShuffleNetV2 pose network - fp32,
// v_mfma_f32_32x32x2_f32, gfx950 (BASELINE configs[3]).  Same module boundary as pw_fused.hip:
//     [conv_bn depthwise 3x3 ->] conv_bn_relu 1x1 [+ the pass-through half + channel_shuffle]
// (lib/network/rtpose_shufflenetV2.py BasicBlock :22-63), same descriptor (rtpose_pw_desc), same packed weights.
//
// Why a second form.  pw_fused.hip's blocks of 4 waves share a 64-pixel A tile: two barriers per 32-channel chunk,
// per-item epilogues of 64 scalar stores per lane with their address arithmetic, 1.9 (plain) to 3.4 (depthwise)
// non-MFMA VALU instructions per MFMA - on a chip whose fp32 VALU and fp32 MFMA share ALUs - and a matrix pipe that
// is busy 52-67 % of the time (profiles/r04_shufflenet_pmc_sq.txt).  Here, as in pw_head.hip:
//   * a work item is 32 pixels (depthwise: an 8 x 4 tile) and belongs to ONE wave; waves share only read-only data
//     (weights, depthwise taps): no barrier after the kernel's first instruction block;
//   * the GEMM is computed TRANSPOSED, C^T[column][pixel] = W^T X^T: A operand = weight fragment straight from L2
//     (lane = column), B operand = activation fragment (lane = pixel).  The accumulators of a lane then hold 4
//     CONSECUTIVE columns of its own pixel per register quadruple: the epilogue is (ReLU +) one 16-byte store per
//     quadruple, no address arithmetic beyond an immediate offset;
//   * depthwise: the wave stages the 10 x 6 halo of its tile (32 channels at a time, coalesced 128-byte reads) in its
//     private LDS slot and lane (pixel, kh) evaluates the 3 x 3 stencil for exactly the four 16-byte planes 2 g + kh
//     it needs as B operands - the depthwise result goes from the VALU into the matrix pipe without touching LDS;
//   * one wave per SIMD, NF <= 8 column fragments = up to 128 accumulator registers, weight fragments one k-group
//     ahead, the next chunk's activations a chunk ahead (across work items), persistent waves.
// The output columns [0, cout) are stored as the contiguous channels lout.choff ..: layers that write runs of the
// four-run layout are packed with a column map (rtpose_pack_pw_weights_cols), exactly like the bf16 plan.
#include <hip/hip_runtime.h>

#include "common.h"
#include "conv_exp.h"

namespace rtpose {

namespace pwt {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef const floatx4 __attribute__((address_space(1)))* gcf4_t;
__device__ __forceinline__ float4 gload4(const void* p) {  // explicit global address space (no FLAT loads)
  const floatx4 v = *(gcf4_t)(unsigned long long)(p);
  return make_float4(v[0], v[1], v[2], v[3]);
}

constexpr int PX = 32;        // pixels of a work item
constexpr int PS = 33;        // plain: LDS plane pitch (float4)
constexpr int TW = 8, TH = 4; // depthwise: the item's pixels are an 8 x 4 tile
constexpr int HWD = TW + 2, HHT = TH + 2, HALO = HWD * HHT;  // 10 x 6 halo
constexpr int PSH = 61;       // depthwise: LDS plane pitch of the staged halo (float4)
constexpr int kMaxK = 512;    // largest K (plane table, depthwise taps in LDS)

struct Args {
  const float* in;
  int in_cstride, in_choff, in_ws, in_hs, in_lead;
  const int32_t* in_planes;  // optional [K / 4]: channel offset (inside the slice) of every 4-channel plane of K
  const float* dw_w;         // DW: [9][K] taps and
  const float* dw_b;         //     [K] bias of the depthwise conv
  const float* w;            // [K / 4][coutp][4]
  const float* bias;         // [coutp]
  float* out;
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  const float* pt;           // pass-through source (interleave form, see pw_fused.hip), or NULL
  int pt_cstride, pt_choff, pt_ws, pt_hs, pt_lead;
  int pt_pairs, pt_a, pt_b, pt_split, pt_d0, pt_d1;
  int N, H, W, M;
  int K, coutp, cout, relu;
  int nitems, tiles_x, tiles_y;
  int npass;  // column passes of 32 NF columns: a work item is (pixel item, pass), item index = pixel item * npass + pass
};

#define RTPOSE_PWT_PIN()         \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)

// WPS = waves per SIMD the kernel is built for: 2 (NF <= 4: <= 256 registers) lets a second wave's MFMAs cover a
// wave's own stalls - store drains, the first loads of a work item - which one wave per SIMD sits out
template <int NF, bool DW, int WPS>
__global__ __launch_bounds__(256, WPS) void pw_t_f32(const Args A) {
  constexpr int NS = DW ? 8 : 4;                        // staged 16-byte pieces per lane and chunk
  constexpr int XS4 = DW ? 8 * PSH : 8 * PS;            // float4 per chunk buffer
  // LDS (dynamic): [4 waves][2 chunk buffers][XS4] | DW: [10][K / 4] depthwise taps + bias | [coutp / 4] bias |
  // [K / 4] plane table
  extern __shared__ __attribute__((aligned(16))) float4 smem4[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int k4 = A.K >> 2;
  float4* const xs = smem4 + wave * (2 * XS4);
  float4* const s_dw = smem4 + 4 * 2 * XS4;
  float4* const s_bias = s_dw + (DW ? 10 * k4 : 0);
  int* const s_plane = reinterpret_cast<int*>(s_bias + (A.coutp >> 2));
  for (int j = tid; j < k4; j += 256) s_plane[j] = A.in_planes ? A.in_planes[j] : 4 * j;
  // the bias lives in LDS: a global load in the epilogue would queue behind the item's own stores (one in-order
  // counter for loads and stores) and the wave would sit out their whole drain before its next MFMA
  for (int j = tid; j < (A.coutp >> 2); j += 256) s_bias[j] = gload4(A.bias + 4 * j);
  if (DW)
    for (int i = tid; i < 10 * k4; i += 256)
      s_dw[i] = i < 9 * k4 ? gload4(A.dw_w + 4 * (size_t)i) : gload4(A.dw_b + 4 * (size_t)(i - 9 * k4));
  __syncthreads();  // the only barrier: from here on a wave is on its own
  const int nwaves = gridDim.x * 4;
  int item = blockIdx.x * 4 + wave;
  if (item >= A.nitems) return;

  const int HW = A.H * A.W;
  const int gtot = A.K >> 3;                 // 8-channel k-groups
  const int nch = (gtot + 3) >> 2;           // 32-channel chunks
  const int ng_last = gtot - 4 * (nch - 1);  // 2 or 4 (K is a multiple of 16)
  const float4* const w4 = reinterpret_cast<const float4*>(A.w);
  const float* const in_base = A.in + A.in_choff;
  const int spl = lane & 7, spx = lane >> 3;  // staging role: plane, first pixel
  const int ty = l31 / TW, tx = l31 % TW;     // DW: this lane's pixel inside the tile

  // ---- per work item: where the lane's staged pieces come from, where its pixel goes ----
  struct Item {
    unsigned sq[DW ? 1 : 4];  // plain: element offsets of the four staged pixels; DW: of the halo's corner pixel
    unsigned oq;              // element offset of this lane's output pixel (+ out_choff)
    unsigned pq;              // ... of its pass-through source pixel (+ pt_choff)
    int pass;                 // column pass
    bool ovalid;
  };
  // DW: halo piece u of this lane = halo pixel hp = spx + 8 u, (hp / 10) rows and hp % 10 pixels after the corner
  // (recomputed per chunk - a handful of VALU instructions per 128 MFMAs - rather than held in 8 registers)
  auto setup = [&](int item_idx) -> Item {
    Item r;
    int n, y, x;
    const int it = item_idx / A.npass;
    r.pass = item_idx - it * A.npass;
    if (DW) {
      const int t = it;
      const int txi = t % A.tiles_x, rr = t / A.tiles_x;
      const int tyi = rr % A.tiles_y;
      n = rr / A.tiles_y;
      const int y0 = tyi * TH, x0 = txi * TW;
      r.sq[0] = (unsigned)(A.in_lead + (n * A.in_hs + y0 - 1) * A.in_ws + x0 - 1);  // >= 0: lead = ws + 1
      y = y0 + ty;
      x = x0 + tx;
      r.ovalid = y < A.H && x < A.W;
      y = min(y, A.H - 1);
      x = min(x, A.W - 1);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int m = min(it * PX + spx + 8 * u, A.M - 1);  // pixels past the end replay the last one (never stored)
        const int nn = m / HW, rr = m - nn * HW;
        const int yy = rr / A.W, xx = rr - yy * A.W;
        r.sq[u] = (unsigned)(A.in_lead + (nn * A.in_hs + yy) * A.in_ws + xx) * (unsigned)A.in_cstride;
      }
      const int mo = it * PX + l31;
      r.ovalid = mo < A.M;
      const int m = min(mo, A.M - 1);
      n = m / HW;
      const int rr = m - n * HW;
      y = rr / A.W;
      x = rr - y * A.W;
    }
    r.oq = (unsigned)(A.out_lead + (n * A.out_hs + y) * A.out_ws + x) * (unsigned)A.out_cstride + (unsigned)A.out_choff;
    r.pq = A.pt ? (unsigned)(A.pt_lead + (n * A.pt_hs + y) * A.pt_ws + x) * (unsigned)A.pt_cstride + (unsigned)A.pt_choff : 0u;
    return r;
  };

  float4 sr[NS];
  // chunk c of item `it` -> registers.  Planes past K (the short last chunk) are clamped to the last valid one:
  // their LDS slots exist and are never multiplied.
  auto stage_load = [&](const Item& it, int c) {
    const unsigned cofs = (unsigned)s_plane[min(8 * c + spl, k4 - 1)];
    if (DW) {
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const int hp = min(spx + 8 * u, HALO - 1);
        const unsigned hoff = (unsigned)((hp / HWD) * A.in_ws + hp % HWD);
        sr[u] = gload4(in_base + ((it.sq[0] + hoff) * (unsigned)A.in_cstride + cofs));
      }
    } else {
#pragma unroll
      for (int u = 0; u < NS; ++u) sr[u] = gload4(in_base + (it.sq[u] + cofs));
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      if (DW) {
        if (spx + 8 * u < HALO) xs[buf * XS4 + spl * PSH + spx + 8 * u] = sr[u];
      } else {
        xs[buf * XS4 + spl * PS + spx + 8 * u] = sr[u];
      }
    }
  };

  const unsigned wl = (unsigned)(kh * A.coutp + l31);  // float4 index of this lane inside a k-group's two planes
  float4 wa[NF], wb[NF];  // weight fragments: two alternating sets
  float4 xa, xb;          // plain: activation fragments likewise
  float4 xd[4];           // DW: the chunk's four activation fragments = the depthwise outputs of this lane
#define RTPOSE_PWT_WLOAD(DST, G, PASS) \
  _Pragma("unroll") for (int f = 0; f < NF; ++f)    \
      DST[f] = gload4(w4 + ((size_t)(2 * (G)) * A.coutp + (size_t)((PASS) * 32 * NF + f * 32) + wl))
#define RTPOSE_PWT_XLOAD(DST, BUF, GI) DST = xs[(BUF) * XS4 + (2 * (GI) + kh) * PS + l31]
#define RTPOSE_PWT_MUL(WV, XV)                                                                         \
  {                                                                                                    \
    const float xv_[4] = {XV.x, XV.y, XV.z, XV.w};                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
      _Pragma("unroll") for (int f = 0; f < NF; ++f) {                                                 \
        const float wv_[4] = {WV[f].x, WV[f].y, WV[f].z, WV[f].w};                                     \
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j], xv_[j], acc[f], 0, 0, 0);                \
      }                                                                                                \
    }                                                                                                  \
  }
  // DW: the 3 x 3 depthwise conv (+ bias) of chunk c, read from the staged halo in LDS buffer `buf`, for this lane's
  // pixel and the planes 2 g + kh, g < 4 -> xd[g]: the B operands of the chunk's four k-groups
  auto dw_compute = [&](int buf, int c) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int pch = 2 * g + kh;                       // plane of the chunk
      const float4* tp = s_dw + min(8 * c + pch, k4 - 1);  // (planes past K: never multiplied)
      const float4* hp = xs + buf * XS4 + pch * PSH + ty * HWD + tx;
      const float4 vb = tp[9 * k4];
      f2 lo = {vb.x, vb.y}, hi = {vb.z, vb.w};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float4 ww = tp[(ky * 3 + kx) * k4];
          const float4 xv = hp[ky * HWD + kx];
          lo = __builtin_elementwise_fma(f2{xv.x, xv.y}, f2{ww.x, ww.y}, lo);
          hi = __builtin_elementwise_fma(f2{xv.z, xv.w}, f2{ww.z, ww.w}, hi);
        }
      }
      xd[g] = make_float4(lo.x, lo.y, hi.x, hi.y);
      RTPOSE_PWT_PIN();  // one plane at a time: 18 LDS reads in flight, not 72
    }
  };

  floatx16 acc[NF];  // C^T: column fragment f, this lane's pixel
  auto init_acc = [&]() {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
  };

  // ---- prologue of the wave ----
  Item cur = setup(item);
  stage_load(cur, 0);
  RTPOSE_PWT_WLOAD(wa, 0, cur.pass);
  init_acc();
  stage_store(0);
  RTPOSE_PWT_PIN();  // (LDS serves a wave's requests in order: the reads below see the writes above)
  if (DW) {
    dw_compute(0, 0);
  } else {
    RTPOSE_PWT_XLOAD(xa, 0, 0);
  }
  int lb = 0;

  while (true) {
    const int nitem = item + nwaves;
    const bool has_next = nitem < A.nitems;
    const Item nxt = setup(has_next ? nitem : item);

    for (int c = 0; c < nch; ++c) {
      const bool lastc = c + 1 == nch;
      const int g0 = 4 * c;
      const int cn = lastc ? 0 : c + 1;  // the chunk after this one: of this item, or the first of the next
      const int nb = lb ^ 1;
      const bool short_chunk = lastc && ng_last == 2;
      // ---- group 0 (set A) ----
      RTPOSE_PWT_WLOAD(wb, g0 + 1, cur.pass);
      stage_load(lastc ? nxt : cur, cn);  // AFTER the weight request: the next wait for weights does not wait for these
      if (!DW) { RTPOSE_PWT_XLOAD(xb, lb, 1); }
      RTPOSE_PWT_PIN();
      if (DW) { RTPOSE_PWT_MUL(wa, xd[0]); } else { RTPOSE_PWT_MUL(wa, xa); }
      RTPOSE_PWT_PIN();
      // ---- group 1 (set B); the last group of a two-group chunk ----
      if (!DW && short_chunk) stage_store(nb);
      RTPOSE_PWT_WLOAD(wa, short_chunk ? 0 : g0 + 2, short_chunk ? nxt.pass : cur.pass);
      RTPOSE_PWT_PIN();
      if (!DW) { RTPOSE_PWT_XLOAD(xa, short_chunk ? nb : lb, short_chunk ? 0 : 2); }
      RTPOSE_PWT_PIN();
      if (DW) { RTPOSE_PWT_MUL(wb, xd[1]); } else { RTPOSE_PWT_MUL(wb, xb); }
      RTPOSE_PWT_PIN();
      if (!short_chunk) {
        // ---- group 2 (set A) ----
        RTPOSE_PWT_WLOAD(wb, g0 + 3, cur.pass);
        if (!DW) { RTPOSE_PWT_XLOAD(xb, lb, 3); }
        RTPOSE_PWT_PIN();
        if (DW) { RTPOSE_PWT_MUL(wa, xd[2]); } else { RTPOSE_PWT_MUL(wa, xa); }
        RTPOSE_PWT_PIN();
        // ---- group 3 (set B): the next chunk goes to LDS, its first fragments are requested ----
        if (!DW) stage_store(nb);
        RTPOSE_PWT_WLOAD(wa, lastc ? 0 : g0 + 4, lastc ? nxt.pass : cur.pass);
        RTPOSE_PWT_PIN();
        if (!DW) { RTPOSE_PWT_XLOAD(xa, nb, 0); }
        RTPOSE_PWT_PIN();
        if (DW) { RTPOSE_PWT_MUL(wb, xd[3]); } else { RTPOSE_PWT_MUL(wb, xb); }
        RTPOSE_PWT_PIN();
      }
      if (DW) {  // the next chunk's halo -> LDS -> stencil -> the next chunk's B operands
        stage_store(nb);
        RTPOSE_PWT_PIN();
        dw_compute(nb, cn);
      }
      lb = nb;
    }

    // ---- epilogue: (ReLU,) 16 bytes per lane and register quadruple ----
    if (cur.ovalid) {
      const int c0 = cur.pass * 32 * NF;  // first column of the pass
      float* const po = A.out + (cur.oq + (unsigned)(c0 + 4 * kh));
      const float4* const pb = s_bias + (c0 >> 2) + kh;
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          if (c0 + f * 32 + rg * 8 < A.cout) {  // (cout is a multiple of 8: a wave-uniform test, an immediate store offset)
            const float4 b = pb[f * 8 + rg * 2];
            float4 v = make_float4(acc[f][rg * 4 + 0] + b.x, acc[f][rg * 4 + 1] + b.y, acc[f][rg * 4 + 2] + b.z,
                                   acc[f][rg * 4 + 3] + b.w);
            if (A.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            *reinterpret_cast<float4*>(po + (f * 32 + rg * 8)) = v;
          }
        }
    }
    init_acc();

    // ---- pass-through half, interleave form (see pw_fused.hip): output channel j < 2 pt_pairs of the pixel takes
    //      source channel pt_a + j / 2 (j even) or pt_b + j / 2 (j odd) and lands at
    //      (j < pt_split ? pt_d0 + j : pt_d1 + j - pt_split).  A lane copies groups of 4 pairs of the item's pixels. ----
    if (A.pt && cur.pass == 0) {
      const int g4 = (A.pt_pairs + 3) >> 2;
      const bool v4ok = !(A.pt_split & 3) && !(A.pt_d0 & 3) && !((A.pt_d1 - A.pt_split) & 3);
      const bool v2ok = !(A.pt_split & 1) && !(A.pt_d0 & 1) && !((A.pt_d1 - A.pt_split) & 1);
      // All loads of a batch of 8 x 64 (pixel, group) pieces first, then all its stores: a store queued between two loads
      // would make the wait for the second load sit out the store's drain (one in-order counter).
      constexpr int PB = 8;
      for (int i0 = 0; i0 < PX * g4; i0 += 64 * PB) {  // (uniform trip count: every lane takes part in the shuffles)
        float4 va[PB], vb[PB];
        int pp[PB], gg[PB];
        bool ok[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
          const int i = min(i0 + lane + 64 * u, PX * g4 - 1);
          pp[u] = i / g4;
          gg[u] = i - pp[u] * g4;
          ok[u] = i0 + lane + 64 * u < PX * g4;
          // the source / destination pixel of item pixel pp: held by lane pp of the wave
          const unsigned sq = (unsigned)__shfl((int)cur.pq, pp[u], 64);
          va[u] = gload4(A.pt + (sq + (unsigned)(A.pt_a + 4 * gg[u])));
          vb[u] = gload4(A.pt + (sq + (unsigned)(A.pt_b + 4 * gg[u])));
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
          const unsigned dq = (unsigned)__shfl((int)cur.oq, pp[u], 64) - (unsigned)A.out_choff;
          const bool pv = __shfl((int)cur.ovalid, pp[u], 64) != 0;
          if (!ok[u] || !pv) continue;
          const float a4[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, b4[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
          const int k0 = 4 * gg[u];   // first pair of the group: outputs j = 2 k0 .. 2 k0 + 7
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {  // halves of 2 pairs = 4 outputs
            const int k = k0 + 2 * hh;
            const int j = 2 * k;
            if (k >= A.pt_pairs) continue;
            if (v4ok && k + 1 < A.pt_pairs) {
              const int dj = j < A.pt_split ? A.pt_d0 + j : A.pt_d1 + j - A.pt_split;
              *reinterpret_cast<float4*>(A.out + (dq + (unsigned)dj)) = make_float4(a4[2 * hh], b4[2 * hh], a4[2 * hh + 1], b4[2 * hh + 1]);
              continue;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int kk = k + e, jj = 2 * kk;
              if (kk >= A.pt_pairs) continue;
              if (v2ok) {
                const int dj = jj < A.pt_split ? A.pt_d0 + jj : A.pt_d1 + jj - A.pt_split;
                *reinterpret_cast<float2*>(A.out + (dq + (unsigned)dj)) = make_float2(a4[2 * hh + e], b4[2 * hh + e]);
              } else {
                const int da = jj < A.pt_split ? A.pt_d0 + jj : A.pt_d1 + jj - A.pt_split;
                const int db = jj + 1 < A.pt_split ? A.pt_d0 + jj + 1 : A.pt_d1 + jj + 1 - A.pt_split;
                A.out[dq + (unsigned)da] = a4[2 * hh + e];
                A.out[dq + (unsigned)db] = b4[2 * hh + e];
              }
            }
          }
        }
      }
    }

    if (!has_next) break;
    item = nitem;
    cur = nxt;
  }
#undef RTPOSE_PWT_MUL
#undef RTPOSE_PWT_XLOAD
#undef RTPOSE_PWT_WLOAD
}
#undef RTPOSE_PWT_PIN

}  // namespace pwt

// 1 when the chain described by `d` has an instance of this form
int pw_t_fits(const rtpose_pw_desc* d) {
  if (!d) return 0;
  if (d->cin < 32 || (d->cin % 16) || d->cin > pwt::kMaxK) return 0;
  if (d->coutp != 64 && d->coutp != 128 && d->coutp != 256) return 0;
  if (d->cout < 8 || (d->cout % 8) || d->cout > d->coutp || d->out_cmap) return 0;
  if ((d->lout.cstride % 4) || (d->lout.choff % 4) || d->lout.choff + d->cout > d->lout.cstride) return 0;
  if ((d->lin.cstride % 4) || (d->lin.choff % 4)) return 0;
  if (!d->in_planes && d->lin.choff + d->cin > d->lin.cstride) return 0;
  if (d->pt_src && (d->pt_pairs <= 0 || (d->lpt.cstride % 4) || ((d->lpt.choff + d->pt_a) % 4) || ((d->lpt.choff + d->pt_b) % 4)))
    return 0;
  return 1;
}

template <int NF, bool DW, int WPS>
static int pw_t_launch_inst(const pwt::Args& a0, hipStream_t s) {
  using namespace pwt;
  Args a = a0;
  a.npass = a.coutp / (32 * NF);
  a.nitems *= a.npass;
  const int k4 = a.K >> 2;
  const size_t lds = (size_t)4 * 2 * (DW ? 8 * PSH : 8 * PS) * 16 + (DW ? (size_t)10 * k4 * 16 : 0) +
                     (size_t)(a.coutp >> 2) * 16 + (size_t)k4 * 4;
  const int slots = 4 * WPS * device_cu_count();  // WPS waves per SIMD
  const int waves = a.nitems < slots ? a.nitems : slots;
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  auto kern = pw_t_f32<NF, DW, WPS>;
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         128 * 1024));
    attr_set.set(dev);
  }
  hipLaunchKernelGGL(kern, dim3(ceil_div(waves, 4)), dim3(256), lds, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int pw_t_launch(const rtpose_pw_desc* d, int N, int H, int W, hipStream_t s) {
  using namespace pwt;
  if (!pw_t_fits(d)) return fail(RTPOSE_E_INVAL, "pw_t: no instance for this chain");
  if (!d->in || !d->w_packed || !d->bias_packed || !d->out) return fail(RTPOSE_E_INVAL, "pw_t: NULL argument");
  if (N <= 0 || H <= 0 || W <= 0) return fail(RTPOSE_E_INVAL, "pw_t: empty tensor");
  const bool dw = d->dw_w != nullptr;
  if (dw && (!d->dw_b || d->lin.ws < W + 1 || d->lin.hs < H + 1 || d->lin.lead < d->lin.ws + 1))
    return fail(RTPOSE_E_INVAL, "pw_t: the depthwise input needs a layout gap of 1 and a bias");
  const long M = (long)N * H * W;
  if (M > 0x7fffffffL || rtpose_layout_pixels(&d->lin, N, H, W) * (size_t)d->lin.cstride >= ((size_t)1 << 31) ||
      rtpose_layout_pixels(&d->lout, N, H, W) * (size_t)d->lout.cstride >= ((size_t)1 << 31) ||
      (d->pt_src && rtpose_layout_pixels(&d->lpt, N, H, W) * (size_t)d->lpt.cstride >= ((size_t)1 << 31)))
    return fail(RTPOSE_E_INVAL, "pw_t: tensors must be below 2^31 floats (32-bit element offsets)");
  Args a;
  memset(&a, 0, sizeof(a));
  a.in = d->in;
  a.in_cstride = d->lin.cstride;
  a.in_choff = d->lin.choff;
  a.in_ws = d->lin.ws;
  a.in_hs = d->lin.hs;
  a.in_lead = d->lin.lead;
  a.in_planes = d->in_planes;
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
  if (d->pt_src) {
    a.pt = d->pt_src;
    a.pt_cstride = d->lpt.cstride;
    a.pt_choff = d->lpt.choff;
    a.pt_ws = d->lpt.ws;
    a.pt_hs = d->lpt.hs;
    a.pt_lead = d->lpt.lead;
    a.pt_pairs = d->pt_pairs;
    a.pt_a = d->pt_a;
    a.pt_b = d->pt_b;
    a.pt_split = d->pt_split;
    a.pt_d0 = d->pt_d0;
    a.pt_d1 = d->pt_d1;
  }
  a.N = N;
  a.H = H;
  a.W = W;
  a.M = (int)M;
  a.K = d->cin;
  a.coutp = d->coutp;
  a.cout = d->cout;
  a.relu = d->relu;
  a.tiles_x = ceil_div(W, TW);
  a.tiles_y = ceil_div(H, TH);
  a.nitems = dw ? N * a.tiles_x * a.tiles_y : ceil_div((int)M, PX);
  // developer builds: RTPOSE_PWT_FORM=1 -> one wave per SIMD with all columns of a pixel item in one pass (NF = coutp / 32)
  static const int wide = [] {
    const char* e = dev_env("RTPOSE_PWT_FORM");
    return e && e[0] == '1' ? 1 : 0;
  }();
  if (wide) {
    switch (d->coutp) {
      case 64: return dw ? pw_t_launch_inst<2, true, 1>(a, s) : pw_t_launch_inst<2, false, 1>(a, s);
      case 128: return dw ? pw_t_launch_inst<4, true, 1>(a, s) : pw_t_launch_inst<4, false, 1>(a, s);
      default: return dw ? pw_t_launch_inst<8, true, 1>(a, s) : pw_t_launch_inst<8, false, 1>(a, s);
    }
  }
  // two waves per SIMD, work items of 32 pixels x <= 128 columns (256 columns: two passes)
  if (d->coutp == 64) return dw ? pw_t_launch_inst<2, true, 2>(a, s) : pw_t_launch_inst<2, false, 2>(a, s);
  return dw ? pw_t_launch_inst<4, true, 2>(a, s) : pw_t_launch_inst<4, false, 2>(a, s);
}

}  // namespace rtpose

extern "C" {

int rtpose_pw_fused_t_fits(const rtpose_pw_desc* d) { return rtpose::pw_t_fits(d); }

int rtpose_pw_fused_t(const rtpose_pw_desc* d, int N, int H, int W, void* stream) {
  return rtpose::pw_t_launch(d, N, H, W, rtpose::as_stream(stream));
}

}  // extern "C"
