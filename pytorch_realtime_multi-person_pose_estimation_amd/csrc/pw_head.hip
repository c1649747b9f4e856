// conv5 + the two heads of the ShuffleNetV2 pose network as ONE back-to-back GEMM launch - fp32,
// v_mfma_f32_32x32x2_f32, gfx950 (BASELINE configs[3]).
//
// Stands in for   slim.conv_bn_relu('conv5', 464, 1024, 1)  ->  self.paf = nn.Conv2d(1024, 38, 1)
//                                                               self.heatmap = nn.Conv2d(1024, 19, 1)
// (lib/network/rtpose_shufflenetV2.py:104, :107-108, forward :143-147).  As two launches the 1024-channel feature made a
// 1.1 GB round trip through HBM per 128-image forward and the heads GEMM (K = 1024, 57 columns) ran at 0.44 of the
// fp32 MFMA peak.  Here it never leaves the registers:
//
//   * WAVE-AUTONOMOUS items.  A work item is 32 pixels and belongs to ONE wave, which computes all 1024 conv5
//     channels of them in NP passes of 32 NF channels and folds every pass into the heads' sums at once.  Waves
//     share nothing but the read-only weights: there is NO __syncthreads in the kernel, a wave's stalls are its own.
//   * BOTH GEMMs ARE COMPUTED TRANSPOSED.  GEMM 1:  C1^T[channel][pixel] = W1^T X^T  - the A operand of the MFMA is the
//     weight fragment (lane = channel, straight from L2), the B operand the activation fragment (lane = pixel, from the
//     wave's private LDS tile).  The accumulator of lane (l31, kh) then holds, for ITS pixel l31, the channels
//     rg * 8 + 4 kh + rr of the fragment - which is exactly what the B operand of GEMM 2,
//       OUT^T[head column][pixel] += W2^T[head column][k] * relu(C1^T)[k][pixel],
//     wants from that lane for the k pair (rg * 8 + rr, rg * 8 + 4 + rr): bias + ReLU are applied in place and the
//     accumulator registers are fed to the matrix pipe again.  No transposition, no LDS round trip of the
//     intermediate, no barrier.  The A operand of GEMM 2 is the heads' packed matrix [k / 4][64][4] read as it is.
//   * The result OUT^T has lane = pixel and 4 consecutive head columns per register quadruple: 16-byte stores.
//   * One wave per SIMD (256 threads per CU, all 512 registers): 128 accumulators of GEMM 1 + 32 of GEMM 2,
//     weight fragments one k-group (32 MFMAs, ~2 k cycles) ahead, the activation tile chunk (32 channels of the
//     wave's 32 pixels, 4 KB) staged through the wave's own double-buffered LDS slot by coalesced 128-byte reads,
//     requested a whole chunk (128 MFMAs) ahead - across passes and across work items.
//   * Persistent waves: wave w takes the items w, w + #waves, ...; the first weight fragments and the first
//     activation chunk of the next item are requested under the last MFMAs of the current one.
//
// Sum order: conv5 walks K ascending; the heads sum their K = 1024 in the order (pass, fragment, rg, rr, kh pair) -
// fixed, independent of the batch (an image's maps do not depend on its batch neighbours).
#include <hip/hip_runtime.h>

#include "common.h"

namespace rtpose {

namespace head {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef const floatx4 __attribute__((address_space(1)))* gcf4_t;
__device__ __forceinline__ float4 gload4(const void* p) {  // explicit global address space (no FLAT loads)
  const floatx4 v = *(gcf4_t)(unsigned long long)(p);
  return make_float4(v[0], v[1], v[2], v[3]);
}

constexpr int PX = 32;   // pixels of a work item (one MFMA fragment)
constexpr int PS = 33;   // LDS plane pitch in float4 (planes 4 banks apart for the 8-lane write groups)
constexpr int N2 = 64;   // head columns (38 + 2 + 19 + 5)
constexpr int kMaxK1 = 1024;  // largest K of the wide conv (plane table in LDS)

struct Args {
  const float* in;
  int in_cstride, in_choff, in_ws, in_hs, in_lead;
  const int32_t* in_planes;  // optional [K1 / 4]: channel offset (inside the slice) of every 4-channel plane of K1
  const float* w1;  // [K1 / 4][N1][4]
  const float* b1;  // [N1]
  const float* w2;  // [N1 / 4][64][4]
  const float* b2;  // [64]
  float* out;
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  int N, H, W, M;
  int K1, N1, nitems;
  FastDiv fHW, fW;  // divisions by launch constants (common.h)
};

#define RTPOSE_HEAD_PIN()        \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)

template <int NF>
__global__ __launch_bounds__(256, 1) void pw_head_f32(const Args A) {
  __shared__ __attribute__((aligned(16))) float4 xs_all[4][2 * 8 * PS];
  // both biases live in LDS (N1 <= 1024 + 64 floats): a global load between two items would queue behind the item's
  // stores - loads and stores share one in-order counter - and the wave would sit out their drain
  __shared__ __attribute__((aligned(16))) float4 s_b1[256], s_b2[16];
  __shared__ int s_plane[kMaxK1 / 4];  // channel offset of every plane of K1 (the input may be a gather of planes)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  float4* const xs = &xs_all[wave][0];
  for (int j = tid; j < A.N1 / 4; j += 256) s_b1[j] = gload4(A.b1 + 4 * j);
  for (int j = tid; j < A.K1 / 4; j += 256) s_plane[j] = A.in_planes ? A.in_planes[j] : 4 * j;
  if (tid < 16) s_b2[tid] = gload4(A.b2 + 4 * tid);
  __syncthreads();  // the only barrier
  const int nwaves = gridDim.x * 4;
  int item = blockIdx.x * 4 + wave;
  if (item >= A.nitems) return;  // (no barriers from here on: a wave may leave alone)

  const int HW = A.H * A.W;
  const int gtot = A.K1 >> 3;              // 8-channel k-groups of GEMM 1
  const int nch = (gtot + 3) >> 2;         // 32-channel chunks
  const int ng_last = gtot - 4 * (nch - 1);  // 2 or 4 (host-checked)
  const int NP = A.N1 / (32 * NF);
  const float4* const w1 = reinterpret_cast<const float4*>(A.w1);
  const float4* const w2 = reinterpret_cast<const float4*>(A.w2);
  const float* const in_base = A.in + A.in_choff;

  // staging role of a lane: 16-byte plane spl of the pixels spx + 8 u, u < 4 (a pixel's chunk = one 128-byte line)
  const int spl = lane & 7, spx = lane >> 3;
  auto pixel_q = [&](int m, int lead, int hs, int ws) {
    const int n = fast_div(m, A.fHW), r = m - n * HW;
    const int y = fast_div(r, A.fW), x = r - y * A.W;
    return lead + (n * hs + y) * ws + x;
  };
  unsigned sq[4], sqn[4];  // element offsets of the lane's four staged pixels: this item / the next one
  auto setup = [&](int it, unsigned* q) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = min(it * PX + spx + 8 * u, A.M - 1);  // pixels past the end replay the last one (never stored)
      q[u] = (unsigned)pixel_q(m, A.in_lead, A.in_hs, A.in_ws) * (unsigned)A.in_cstride;
    }
  };
  float4 sr[4];
  // chunk c of the pixels q -> registers.  Planes past K1 (the short last chunk) are clamped to the last valid one:
  // their LDS slots exist and are never multiplied.
  auto stage_load = [&](const unsigned* q, int c) {
    const unsigned cofs = (unsigned)s_plane[min(8 * c + spl, (A.K1 >> 2) - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) sr[u] = gload4(in_base + (q[u] + cofs));
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u) xs[buf * 8 * PS + spl * PS + spx + 8 * u] = sr[u];
  };

  const unsigned wl = (unsigned)(kh * A.N1 + l31);  // float4 index of this lane inside a k-group's two planes
  float4 wa[NF], wb[NF];  // weight fragments of GEMM 1: two alternating sets
  float4 xa, xb;          // activation fragments likewise
#define RTPOSE_HEAD_WLOAD(DST, G, P)                                              \
  _Pragma("unroll") for (int f = 0; f < NF; ++f)                                  \
      DST[f] = gload4(w1 + ((size_t)(2 * (G)) * A.N1 + (size_t)((P) * 32 * NF + f * 32) + wl))
#define RTPOSE_HEAD_XLOAD(DST, BUF, GI) DST = xs[(BUF) * 8 * PS + (2 * (GI) + kh) * PS + l31]
#define RTPOSE_HEAD_MUL(WV, XV)                                                                        \
  {                                                                                                    \
    const float xv_[4] = {XV.x, XV.y, XV.z, XV.w};                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
      _Pragma("unroll") for (int f = 0; f < NF; ++f) {                                                 \
        const float wv_[4] = {WV[f].x, WV[f].y, WV[f].z, WV[f].w};                                     \
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[j], xv_[j], acc[f], 0, 0, 0);                \
      }                                                                                                \
    }                                                                                                  \
  }

  floatx16 acc[NF];  // C1^T of the pass: channel fragment f, this lane's pixel
  floatx16 o0, o1;   // OUT^T: head columns 0..31 / 32..63 of this lane's pixel
  auto init_acc = [&](int f, int p) {  // bias of conv5 rides in the accumulator
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const float4 b = s_b1[p * 8 * NF + f * 8 + rg * 2 + kh];
      acc[f][rg * 4 + 0] = b.x;
      acc[f][rg * 4 + 1] = b.y;
      acc[f][rg * 4 + 2] = b.z;
      acc[f][rg * 4 + 3] = b.w;
    }
  };
  auto init_out = [&]() {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const float4 b0 = s_b2[rg * 2 + kh], b1 = s_b2[8 + rg * 2 + kh];
      o0[rg * 4 + 0] = b0.x; o0[rg * 4 + 1] = b0.y; o0[rg * 4 + 2] = b0.z; o0[rg * 4 + 3] = b0.w;
      o1[rg * 4 + 0] = b1.x; o1[rg * 4 + 1] = b1.y; o1[rg * 4 + 2] = b1.z; o1[rg * 4 + 3] = b1.w;
    }
  };

  // ---- prologue of the wave: first chunk, first fragments --------------------------------------------------------
  setup(item, sq);
  stage_load(sq, 0);
  RTPOSE_HEAD_WLOAD(wa, 0, 0);
#pragma unroll
  for (int f = 0; f < NF; ++f) init_acc(f, 0);
  init_out();
  stage_store(0);
  RTPOSE_HEAD_PIN();  // (LDS serves a wave's requests in order: the write is seen by the read below)
  RTPOSE_HEAD_XLOAD(xa, 0, 0);
  int lb = 0;  // LDS buffer of the current chunk

  while (true) {
    const int nitem = item + nwaves;
    const bool has_next = nitem < A.nitems;
    setup(has_next ? nitem : item, sqn);
    // output pixel of this lane
    const int mo = item * PX + l31;
    const bool ovalid = mo < A.M;
    const unsigned oq = (unsigned)pixel_q(min(mo, A.M - 1), A.out_lead, A.out_hs, A.out_ws) * (unsigned)A.out_cstride +
                        (unsigned)A.out_choff;

    for (int p = 0; p < NP; ++p) {
      const bool lastp = p + 1 == NP;
      for (int c = 0; c < nch; ++c) {
        const bool lastc = c + 1 == nch;
        const int g0 = 4 * c;
        // the chunk after this one: the next of the pass, the first of the next pass, the first of the next item
        const int cn = lastc ? 0 : c + 1;
        const int pn = lastc ? (lastp ? 0 : p + 1) : p;
        const bool to_next_item = lastc && lastp;
        unsigned qn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) qn[u] = to_next_item ? sqn[u] : sq[u];
        const int nb = lb ^ 1;
        // ---- group 0 (set A) ----
        RTPOSE_HEAD_WLOAD(wb, g0 + 1, p);
        stage_load(qn, cn);  // AFTER the weight request: the next wait for weights does not wait for these
        RTPOSE_HEAD_XLOAD(xb, lb, 1);
        RTPOSE_HEAD_PIN();
        RTPOSE_HEAD_MUL(wa, xa);
        RTPOSE_HEAD_PIN();
        // ---- group 1 (set B); the last group of a two-group chunk: the next chunk goes to LDS and its first
        //      fragments are requested (selected operands instead of a second code path: one MFMA region) ----
        const bool short_chunk = lastc && ng_last == 2;
        if (short_chunk) stage_store(nb);
        RTPOSE_HEAD_WLOAD(wa, short_chunk ? 0 : g0 + 2, short_chunk ? pn : p);
        RTPOSE_HEAD_PIN();
        RTPOSE_HEAD_XLOAD(xa, short_chunk ? nb : lb, short_chunk ? 0 : 2);
        RTPOSE_HEAD_PIN();
        RTPOSE_HEAD_MUL(wb, xb);
        RTPOSE_HEAD_PIN();
        if (!short_chunk) {
          // ---- group 2 (set A) ----
          RTPOSE_HEAD_WLOAD(wb, g0 + 3, p);
          RTPOSE_HEAD_XLOAD(xb, lb, 3);
          RTPOSE_HEAD_PIN();
          RTPOSE_HEAD_MUL(wa, xa);
          RTPOSE_HEAD_PIN();
          // ---- group 3 (set B): the next chunk goes to LDS, its first fragments are requested ----
          stage_store(nb);
          RTPOSE_HEAD_WLOAD(wa, lastc ? 0 : g0 + 4, pn);
          RTPOSE_HEAD_PIN();
          RTPOSE_HEAD_XLOAD(xa, nb, 0);
          RTPOSE_HEAD_PIN();
          RTPOSE_HEAD_MUL(wb, xb);
          RTPOSE_HEAD_PIN();
        }
        lb = nb;
      }

      // ---- GEMM 2: fold the pass' 32 NF channels into the heads' sums, straight from the accumulators ------------
      {
        float4 va[4][2], vb[4][2];  // W2 fragments of channel fragment f: [rg][head column half], two alternating sets
#define RTPOSE_HEAD_W2LOAD(DST, F)                                                                      \
  _Pragma("unroll") for (int rg = 0; rg < 4; ++rg) {                                                    \
    DST[rg][0] = gload4(w2 + ((size_t)((p * NF + (F)) * 8 + 2 * rg + kh) * N2 + l31));                  \
    DST[rg][1] = gload4(w2 + ((size_t)((p * NF + (F)) * 8 + 2 * rg + kh) * N2 + 32 + l31));             \
  }
#define RTPOSE_HEAD_FOLD(WV, F)                                                                         \
  _Pragma("unroll") for (int rg = 0; rg < 4; ++rg) {                                                    \
    const float w0_[4] = {WV[rg][0].x, WV[rg][0].y, WV[rg][0].z, WV[rg][0].w};                          \
    const float w1_[4] = {WV[rg][1].x, WV[rg][1].y, WV[rg][1].z, WV[rg][1].w};                          \
    _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                                  \
      const float v = fmaxf(acc[F][rg * 4 + rr], 0.f);                                                  \
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0_[rr], v, o0, 0, 0, 0);                               \
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1_[rr], v, o1, 0, 0, 0);                               \
    }                                                                                                   \
  }
        const int pi = lastp ? 0 : p + 1;  // the pass whose bias the accumulators take next (next item: pass 0)
        float4 bz[4];                      // that bias, requested before the fragment is folded, written after
#define RTPOSE_HEAD_BLOAD(F) \
  _Pragma("unroll") for (int rg = 0; rg < 4; ++rg) bz[rg] = s_b1[pi * 8 * NF + (F) * 8 + rg * 2 + kh]
#define RTPOSE_HEAD_BSET(F)                                \
  _Pragma("unroll") for (int rg = 0; rg < 4; ++rg) {       \
    acc[F][rg * 4 + 0] = bz[rg].x;                         \
    acc[F][rg * 4 + 1] = bz[rg].y;                         \
    acc[F][rg * 4 + 2] = bz[rg].z;                         \
    acc[F][rg * 4 + 3] = bz[rg].w;                         \
  }
        RTPOSE_HEAD_W2LOAD(va, 0);
#pragma unroll
        for (int f = 0; f < NF; f += 2) {
          RTPOSE_HEAD_W2LOAD(vb, f + 1);
          RTPOSE_HEAD_BLOAD(f);
          RTPOSE_HEAD_PIN();
          RTPOSE_HEAD_FOLD(va, f);
          RTPOSE_HEAD_PIN();
          RTPOSE_HEAD_BSET(f);
          if (f + 2 < NF) { RTPOSE_HEAD_W2LOAD(va, f + 2); }
          RTPOSE_HEAD_BLOAD(f + 1);
          RTPOSE_HEAD_PIN();
          RTPOSE_HEAD_FOLD(vb, f + 1);
          RTPOSE_HEAD_PIN();
          RTPOSE_HEAD_BSET(f + 1);
        }
#undef RTPOSE_HEAD_BSET
#undef RTPOSE_HEAD_BLOAD
#undef RTPOSE_HEAD_FOLD
#undef RTPOSE_HEAD_W2LOAD
      }
    }

    // ---- epilogue: OUT^T -> 16 bytes per lane and register quadruple -----------------------------------------------
    if (ovalid) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float* po = A.out + (oq + (unsigned)(rg * 8 + 4 * kh));
        *reinterpret_cast<float4*>(po) = make_float4(o0[rg * 4 + 0], o0[rg * 4 + 1], o0[rg * 4 + 2], o0[rg * 4 + 3]);
        *reinterpret_cast<float4*>(po + 32) = make_float4(o1[rg * 4 + 0], o1[rg * 4 + 1], o1[rg * 4 + 2], o1[rg * 4 + 3]);
      }
    }
    if (!has_next) break;
    init_out();
    item = nitem;
#pragma unroll
    for (int u = 0; u < 4; ++u) sq[u] = sqn[u];
  }
#undef RTPOSE_HEAD_MUL
#undef RTPOSE_HEAD_XLOAD
#undef RTPOSE_HEAD_WLOAD
}
#undef RTPOSE_HEAD_PIN

// zero the columns [c0, c1) of a packed pointwise matrix [K / 4][coutp][4] and of its bias
__global__ void zero_columns_kernel(float* __restrict__ wp, float* __restrict__ bp, int K, int coutp, int c0, int c1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nc = c1 - c0;
  if (i < nc) bp[c0 + i] = 0.f;
  if (i >= K * nc) return;
  const int n = c0 + i % nc, c = i / nc;
  wp[((size_t)(c >> 2) * coutp + n) * 4 + (c & 3)] = 0.f;
}

}  // namespace head

int pw_zero_columns_launch(float* wp, float* bp, int K, int coutp, int c0, int c1, hipStream_t s) {
  if (!wp || !bp || K <= 0 || c0 < 0 || c1 > coutp || c0 >= c1) return 0;
  hipLaunchKernelGGL(head::zero_columns_kernel, dim3(ceil_div(K * (c1 - c0), 256)), dim3(256), 0, s, wp, bp, K, coutp,
                     c0, c1);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

// d1: the wide pointwise conv (+ReLU), plain packing [cin / 4][cout][4] with cout a multiple of 256, cin a multiple
// of 16; d2: the heads' shared matrix [cout1 / 4][64][4], columns at their output channels (no column map).
int pw_head_fits(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2) {
  if (!d1 || !d2) return 0;
  if (d1->cin <= 0 || (d1->cin % 16) || d1->cin < 32 || d1->cin > head::kMaxK1 || d1->coutp <= 0 || (d1->coutp % 256) ||
      d1->cout != d1->coutp)
    return 0;
  if (!d1->relu || d2->relu || d1->dw_w || d2->dw_w || d1->pt_src || d2->pt_src || d1->out_cmap || d2->out_cmap) return 0;
  if (d2->cin != d1->coutp || d2->coutp != head::N2 || d2->cout < 1 || d2->cout > head::N2 || d1->coutp > 1024) return 0;
  if ((d1->lin.cstride % 4) || (d1->lin.choff % 4) || (!d1->in_planes && d1->lin.choff + d1->cin > d1->lin.cstride)) return 0;
  if ((d2->lout.cstride % 4) || (d2->lout.choff % 4) || d2->lout.choff + head::N2 > d2->lout.cstride) return 0;
  return 1;
}

int pw_head_launch(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2, int N, int H, int W, hipStream_t s) {
  using namespace head;
  if (!pw_head_fits(d1, d2))
    return fail(RTPOSE_E_INVAL, "pw_head: needs cin %% 16 == 0 -> cout %% 256 == 0 (+ReLU) -> 64 head columns, contiguous slices");
  if (!d1->in || !d1->w_packed || !d1->bias_packed || !d2->w_packed || !d2->bias_packed || !d2->out)
    return fail(RTPOSE_E_INVAL, "pw_head: NULL argument");
  if (N <= 0 || H <= 0 || W <= 0) return fail(RTPOSE_E_INVAL, "pw_head: empty tensor");
  const long M = (long)N * H * W;
  if (M > 0x7fffffffL || rtpose_layout_pixels(&d1->lin, N, H, W) * (size_t)d1->lin.cstride >= ((size_t)1 << 31) ||
      rtpose_layout_pixels(&d2->lout, N, H, W) * (size_t)d2->lout.cstride >= ((size_t)1 << 31))
    return fail(RTPOSE_E_INVAL, "pw_head: tensors must be below 2^31 floats (32-bit element offsets)");
  Args a;
  memset(&a, 0, sizeof(a));
  a.in = d1->in;
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
  a.nitems = ceil_div((int)M, PX);
  a.fHW = make_fastdiv(H * W);
  a.fW = make_fastdiv(W);
  const int waves = a.nitems < 4 * device_cu_count() ? a.nitems : 4 * device_cu_count();  // one wave per SIMD
  hipLaunchKernelGGL(pw_head_f32<8>, dim3(ceil_div(waves, 4)), dim3(256), 0, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace rtpose

extern "C" {

int rtpose_pw_head_fits(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2) { return rtpose::pw_head_fits(d1, d2); }

int rtpose_pw_head(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2, int N, int H, int W, void* stream) {
  return rtpose::pw_head_launch(d1, d2, N, H, W, rtpose::as_stream(stream));
}

}  // extern "C"
