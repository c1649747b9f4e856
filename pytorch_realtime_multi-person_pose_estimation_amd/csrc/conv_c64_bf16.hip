// bf16 3x3 convolution for layers with 64 INPUT channels, gfx950 (MI355X): conv1_2 = nn.Conv2d(64, 64, 3, 1, 1) + nn.ReLU +
// nn.MaxPool2d(2, 2) and conv2_1 = nn.Conv2d(64, 128, 3, 1, 1) + nn.ReLU of the VGG-19 front end
// (lib/network/rtpose_vgg.py:23-35, table :69-72) in the bf16 plan (BASELINE config 3 arithmetic: operands rounded to bf16,
// exact products, fp32 accumulation, fp32 bias / ReLU / pool, output rounded to bf16 - oracle/net_oracle.py:forward_bf16_emulated).
//
// Why its own kernel (round 6): these are the two widest maps of the network (368 x 368 and 184 x 184) with the shortest
// contraction - K = 576.  In the generic implicit-GEMM kernel (conv_mfma_bf16.hip) a 128-pixel tile holds 1.9 us of matrix
// time behind ~7.5 us of per-tile prologue (the halo's memory latency) and epilogue: conv1_2 0.49 ms = 0.26 of the bf16 MFMA
// peak and 0.27 of HBM - bound by neither.  Here:
//   * 64 channels are ONE chunk: the whole K of a tile is resident.  A block owns a 16 x 32 pixel tile; its 18 x 34 halo is
//     8 planes of 16 bytes per pixel in LDS (77 KB; plane stride 617 pixels, = 1 mod 8, so the 8 lanes that store one pixel's
//     128 bytes hit 8 different bank groups and the 32 lanes that read 32 neighbouring pixels of a plane read 512 contiguous
//     bytes).  Two blocks fit a CU: while one is in its epilogue or parks its next halo, the other multiplies - measured
//     (tools/exp/c64_timeline.py), the K loops of the two overlap almost always: the matrix pipe is busy at the rate the power
//     controller allows with these operands (profiles/r04_bf16_mfma_power_cap.txt).
//   * a wave multiplies 4 image rows x 32 pixels x 64 output channels (8 accumulators of 32 x 32): per K step of 16 channels
//     it reads 4 pixel fragments from LDS (ds_read_b128, compile-time offsets: tap and plane are immediates) and 2 filter
//     fragments through the L1 (16-byte buffer loads at a uniform offset, five steps ahead in a 6-entry register ring that
//     runs on across tile boundaries - the filters are the same for every tile) for 8 v_mfma_f32_32x32x16_bf16: half of the
//     LDS and a quarter of the L1 bandwidth at full matrix rate.
//   * persistent blocks (2 per CU) walk the tiles; the next tile's halo is requested in two halves AROUND the halves of the
//     epilogue - into the accumulator registers it has finished with - and parked behind it.  Layers with 128 output channels
//     are two passes of 64 (a block keeps its pass: grid = a multiple of the pass count).
//   * the TRANSPOSED product (filters as the row operand): a lane holds one pixel and, after one v_permlane32_swap per register
//     pair, 8 consecutive output channels of it - 16-byte stores, no LDS in the epilogue.  The fused 2 x 2 max-pool is a max
//     over the wave's row pairs (registers) and over lane pairs (one DPP quad permute) in front of that.
// Filters in the generic kernel's packing for 32-channel chunks, [chunk][tap][piece][cout_pad][8 bf16]
// (rtpose_pack_conv_weights_bf16): step i = (chunk * 9 + tap) * 2 + half of the K loop reads the pieces 2 i, 2 i + 1.
// The summation order differs from the generic kernel's (fp32 sums of exact products: the last bits), the contract is the
// same: tests/test_bf16_gpu.py runs both against the emulation.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "common.h"
#include "conv_exp.h"

namespace rtpose {

namespace c64 {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

constexpr int TH = 16, TW = 32;         // pixel tile of a block: 4 waves x 4 rows
constexpr int HR = TH + 2, HC = TW + 2;  // halo
constexpr int HP = HR * HC;              // 612 halo pixels
constexpr int PS = 617;                  // LDS pixels per plane
constexpr int NSTEP = 36;                // K steps of 16 channels: 2 chunks x 9 taps x 2 halves
constexpr int RING = 6, AHEAD = 5;       // filter ring
static_assert(NSTEP % RING == 0, "the filter ring keeps its phase across tiles");
static_assert(PS >= HP && PS % 8 == 1, "plane stride");
constexpr size_t kLdsBytes = (size_t)8 * PS * 16;

struct Args {
  const unsigned short* in;
  const void* w;
  const float* bias;
  unsigned short* out;
  int in_cstride, in_choff, in_ws, in_hs, in_lead;
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  unsigned in_bytes, w_bytes;
  int N, H, W, cout_pad, relu;
  int tiles_x, tiles_y, tiles, ntn;  // ntn = passes of 64 output channels
  FastDiv f_tpi, f_tx;
};

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {  // (bf16(a), bf16(b)), RNE, a in the low half
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float fl2 __attribute__((ext_vector_type(2)));
  const fl2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}
// max as ONE instruction: fmaxf compiles to v_max_f32 behind a canonicalising v_max_f32 x, x per operand the compiler cannot
// prove quiet (an accumulator) - 381 v_max in the pooling epilogue instead of 192
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax_lo(float lo, float a) {  // lo: wave-uniform lower bound (0 = ReLU, -inf = none)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(lo), "v"(a));
  return r;
}
// max(v, v of lane ^ 1): the DPP quad permute [1, 0, 3, 2] on the first operand.  (s_nop 1: a DPP operand written by the VALU
// instruction in front needs two wait states, and the hazard pass does not look into inline assembly)
__device__ __forceinline__ float max_with_lane_xor1(float v) {
  float r;
  asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
  return r;
}

#ifdef RTPOSE_EXP_C64_TIMELINE  // developer build: wall_clock64 stamps per block and tile (tools/exp/c64_timeline.py)
#ifndef RTPOSE_DEV_BUILD
#error "RTPOSE_EXP_C64_TIMELINE is a developer-build experiment (tools/build_dev.sh)"
#endif
__device__ unsigned long long g_c64_tl[64][24][6];
#define RTPOSE_C64_TL(i)                                                   \
  if (tid == 0 && blockIdx.x < 64 && tl_k < 24) g_c64_tl[blockIdx.x][tl_k][i] = wall_clock64()
#else
#define RTPOSE_C64_TL(i)
#endif

template <bool POOL>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_bf16(const Args A) {
  extern __shared__ __attribute__((aligned(16))) uintx4 halo[];  // [8 planes][PS pixels] x 16 bytes
  __shared__ __attribute__((aligned(16))) float s_bias[64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int pass = blockIdx.x % A.ntn;
  const int tstep = gridDim.x / A.ntn;
  int tile = blockIdx.x / A.ntn;
  if (tile >= A.tiles) return;
  if (tid < 64) s_bias[tid] = A.bias[pass * 64 + tid];

  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A.w), 0, (int)A.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t irs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A.in), 0, (int)A.in_bytes, 0x00020000);
  const unsigned lane_w = (unsigned)(kh * A.cout_pad + pass * 64 + l31) * 16u;
  const unsigned step_w = (unsigned)(2 * A.cout_pad) * 16u;  // bytes per K step
  auto wload = [&](int step, int nf) -> uintx4 {
    return __builtin_amdgcn_raw_buffer_load_b128(wrs, lane_w + (unsigned)nf * 512u, (unsigned)(step % NSTEP) * step_w, 0);
  };
  uintx4 wq[RING][2];
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) {
    wq[i][0] = wload(i, 0);
    wq[i][1] = wload(i, 1);
  }

  const int plane = tid & 7, ps = tid >> 3;  // staging role: piece `plane` of halo column `ps` in every halo row
  const uintx4* const pl = halo + kh * PS + (4 * wave) * HC + l31;
#define RTPOSE_C64_PIN()         \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)

#ifdef RTPOSE_EXP_C64_TIMELINE
  int tl_k = 0;
#endif
  // ---- halo of a tile: rows y0 - 1 .. y0 + TH, columns x0 - 1 .. x0 + TW; past the image: the zero gap row / column.  A
  // thread fetches its piece of column `ps` in all 18 rows (per-lane column offset, uniform row offset: no address
  // arithmetic per load), and the 18 x 2 pixels of the last two columns are shared out.  The loads of tile t + 1 are issued
  // in two halves around the halves of tile t's epilogue (their registers are the accumulators the epilogue has finished
  // with) and parked in LDS behind it: their latency and their issue run under the epilogue's VALU work. ----
  uintx4 v[HR], ex[2];
  int er[2], ec[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = tid + 256 * k;  // (row, column 32 + (e / 8) % 2), same plane
    er[k] = min(e >> 4, HR - 1);
    ec[k] = TW + ((e >> 3) & 1);
  }
  struct Tile {
    int n, y0, x0;
  };
  auto coords = [&](int t) {
    Tile c;
    c.n = fast_div(t, A.f_tpi);
    const int rt = t - c.n * (A.tiles_x * A.tiles_y);
    const int ty = fast_div(rt, A.f_tx);
    c.y0 = ty * TH;
    c.x0 = (rt - ty * A.tiles_x) * TW;
    return c;
  };
  auto issue_rows = [&](const Tile& c, int r0, int r1) {
    const unsigned col_b = ((unsigned)min(c.x0 + ps, A.W + 1) * (unsigned)A.in_cstride + (unsigned)(A.in_choff + plane * 8)) * 2u;
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      if (r < r0 || r >= r1) continue;
      const int yy = min(c.y0 - 1 + r, A.H);
      const unsigned row_b = (unsigned)(A.in_lead + (c.n * A.in_hs + yy) * A.in_ws - 1) * (unsigned)A.in_cstride * 2u;
      v[r] = __builtin_amdgcn_raw_buffer_load_b128(irs, col_b, row_b, 0);
    }
  };
  auto issue_extra = [&](const Tile& c) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int yy = min(c.y0 - 1 + er[k], A.H);
      const unsigned q = (unsigned)(A.in_lead + (c.n * A.in_hs + yy) * A.in_ws - 1 + min(c.x0 + ec[k], A.W + 1));
      ex[k] = __builtin_amdgcn_raw_buffer_load_b128(irs, (q * (unsigned)A.in_cstride + (unsigned)(A.in_choff + plane * 8)) * 2u, 0, 0);
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int r = 0; r < HR; ++r) halo[plane * PS + r * HC + ps] = v[r];
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + 256 * k < HR * 16) halo[plane * PS + er[k] * HC + ec[k]] = ex[k];
  };

  Tile cur = coords(tile);
  issue_rows(cur, 0, HR);
  issue_extra(cur);
  park();
  __syncthreads();

  for (;;) {
    RTPOSE_C64_TL(0);  // halo in LDS
    // ---- 36 K steps x (4 rows x 2 channel halves); the filters of step i + 5 and the pixels of step i + 1 are requested
    // before the 8 MFMAs of step i ----
    floatx16 acc[4][2];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;
    uintx4 pf[2][4];
    auto pread = [&](int i, uintx4(&dst)[4]) {
      const int c2 = i / 18, t = (i % 18) / 2, sh = i & 1;
      const int dy = t / 3, dx = t % 3;
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) dst[mf] = pl[(4 * c2 + 2 * sh) * PS + (mf + dy) * HC + dx];
    };
    pread(0, pf[0]);
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
      wq[(i + AHEAD) % RING][0] = wload(i + AHEAD, 0);
      wq[(i + AHEAD) % RING][1] = wload(i + AHEAD, 1);
      if (i + 1 < NSTEP) pread(i + 1, pf[(i + 1) & 1]);
      RTPOSE_C64_PIN();  // (else the scheduler sinks the reads to one MFMA in front of their use)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wq[i % RING][nf]),
                                                                __builtin_bit_cast(bf16x8, pf[i & 1][mf]), acc[mf][nf], 0, 0, 0);
      RTPOSE_C64_PIN();
    }
    RTPOSE_C64_TL(1);  // this wave's K loop done
    __syncthreads();  // the halo may be overwritten
    RTPOSE_C64_TL(2);

    // the next tile (the last one fetches its own halo again: no branch around the loads)
    const bool more = tile + tstep < A.tiles;
    const Tile nxt = coords(more ? tile + tstep : tile);
    issue_rows(nxt, 0, HR / 2);
    issue_extra(nxt);
    RTPOSE_C64_PIN();

    // ---- epilogue: lane = pixel x0 + l31 of a row; registers 4 j .. 4 j + 3 of a half = channels 8 j + 4 kh .. + 3 ----
    const float4* const b4 = reinterpret_cast<const float4*>(s_bias);
    const float relu_lo = A.relu ? 0.f : -__builtin_inff();
    constexpr int ROWS = POOL ? 2 : 4;
#pragma unroll
    for (int ro = 0; ro < ROWS; ++ro) {
      if (ro == ROWS / 2) {
        RTPOSE_C64_PIN();
        issue_rows(nxt, HR / 2, HR);
        RTPOSE_C64_PIN();
      }
      int oy, ox;
      bool ok;
      if (POOL) {
        oy = ((cur.y0 + 4 * wave) >> 1) + ro;
        ox = (cur.x0 + l31) >> 1;
        ok = !(l31 & 1) && oy < (A.H >> 1) && ox < (A.W >> 1);
        oy = min(oy, (A.H >> 1) - 1);
        ox = min(ox, (A.W >> 1) - 1);
      } else {
        oy = cur.y0 + 4 * wave + ro;
        ox = cur.x0 + l31;
        ok = oy < A.H && ox < A.W;
        oy = min(oy, A.H - 1);
        ox = min(ox, A.W - 1);
      }
      const size_t q = (size_t)A.out_lead + (size_t)(cur.n * A.out_hs + oy) * A.out_ws + ox;
      unsigned short* const op = A.out + q * A.out_cstride + A.out_choff + pass * 64;
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const float4 bl = b4[(nf * 32 + 16 * m + 4 * kh) >> 2], bh = b4[(nf * 32 + 16 * m + 8 + 4 * kh) >> 2];
          const float blv[4] = {bl.x, bl.y, bl.z, bl.w}, bhv[4] = {bh.x, bh.y, bh.z, bh.w};
          float lo[4], hi[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a, b;
            if (POOL) {
              a = max_with_lane_xor1(vmax(acc[2 * ro][nf][8 * m + e], acc[2 * ro + 1][nf][8 * m + e]));
              b = max_with_lane_xor1(vmax(acc[2 * ro][nf][8 * m + 4 + e], acc[2 * ro + 1][nf][8 * m + 4 + e]));
            } else {
              a = acc[ro][nf][8 * m + e];
              b = acc[ro][nf][8 * m + 4 + e];
            }
            lo[e] = vmax_lo(relu_lo, a + blv[e]);
            hi[e] = vmax_lo(relu_lo, b + bhv[e]);
          }
          unsigned a2[2], b2[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            a2[e] = pack_bf16x2(lo[2 * e], lo[2 * e + 1]);  // group 2 m, channels 4 kh + 2 e, + 1
            b2[e] = pack_bf16x2(hi[2 * e], hi[2 * e + 1]);  // group 2 m + 1
            const auto sw = __builtin_amdgcn_permlane32_swap(a2[e], b2[e], false, false);  // lanes 32..63 of a <-> 0..31 of b
            a2[e] = sw[0];
            b2[e] = sw[1];
          }
          // kh = 0: all 8 channels of group 2 m; kh = 1: of group 2 m + 1
          if (ok) *reinterpret_cast<uint4*>(op + nf * 32 + 8 * (2 * m + kh)) = make_uint4(a2[0], a2[1], b2[0], b2[1]);
        }
    }
    RTPOSE_C64_TL(3);  // stores issued
    if (!more) break;
    park();
    RTPOSE_C64_TL(4);  // next halo parked
    __syncthreads();
    tile += tstep;
    cur = nxt;
#ifdef RTPOSE_EXP_C64_TIMELINE
    ++tl_k;
#endif
  }
}

}  // namespace c64

// 1: the launch below takes the conv (else the generic kernel does)
int conv_c64_bf16_fits(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, int out_f32, int split) {
  if (!d || ngroups != 1 || out_f32 || split) return 0;
  const rtpose_conv_desc& c = d[0];
  if (c.k != 3 || c.cin != 64 || c.cout < 64 || (c.cout % 64) || c.out_cmap || c.in_plane_pixels || c.out_plane_pixels) return 0;
  if ((c.lin.cstride % 8) || (c.lin.choff % 8) || c.lin.choff + 64 > c.lin.cstride) return 0;
  if ((c.lout.cstride % 8) || (c.lout.choff % 8) || c.lout.choff + c.cout > c.lout.cstride) return 0;
  if (c.lin.ws < W + 1 || c.lin.hs < H + 1 || c.lin.lead < c.lin.ws + 1) return 0;
  if (c.pool && ((H | W) & 1)) return 0;
  // unsigned 32-bit byte offsets inside the kernel (the 2 x 368 scale of the multi-scale flow at batch 32 is 2.2 GB)
  if (rtpose_layout_pixels(&c.lin, N, H, W) * (size_t)c.lin.cstride * 2 > 0xffffffffULL) return 0;
  if ((size_t)N * ceil_div(W, c64::TW) * ceil_div(H, c64::TH) > 0x3fffffffULL) return 0;
  return 1;
}

int conv_c64_bf16_launch(const rtpose_conv_desc* d, int N, int H, int W, hipStream_t s) {
  using namespace c64;
  if (!conv_c64_bf16_fits(d, 1, N, H, W, 0, 0) || N <= 0 || H <= 0 || W <= 0)
    return fail(RTPOSE_E_INVAL, "conv_c64_bf16: needs a 3x3 conv of 64 bf16 input channels into 16-byte aligned slices");
  const rtpose_conv_desc& c = d[0];
  Args a;
  memset(&a, 0, sizeof(a));
  a.in = reinterpret_cast<const unsigned short*>(c.in);
  a.w = c.w_packed;
  a.bias = c.bias_packed;
  a.out = reinterpret_cast<unsigned short*>(c.out);
  a.in_cstride = c.lin.cstride;
  a.in_choff = c.lin.choff;
  a.in_ws = c.lin.ws;
  a.in_hs = c.lin.hs;
  a.in_lead = c.lin.lead;
  a.out_cstride = c.lout.cstride;
  a.out_choff = c.lout.choff;
  a.out_ws = c.lout.ws;
  a.out_hs = c.lout.hs;
  a.out_lead = c.lout.lead;
  a.in_bytes = (unsigned)(rtpose_layout_pixels(&c.lin, N, H, W) * (size_t)c.lin.cstride * 2);
  a.cout_pad = cout_pad(c.cout);
  a.w_bytes = (unsigned)((size_t)2 * 9 * 4 * a.cout_pad * 16);
  a.N = N;
  a.H = H;
  a.W = W;
  a.relu = c.relu;
  a.tiles_x = ceil_div(W, TW);
  a.tiles_y = ceil_div(H, TH);
  a.tiles = N * a.tiles_x * a.tiles_y;
  a.ntn = c.cout / 64;
  a.f_tpi = make_fastdiv(a.tiles_x * a.tiles_y);
  a.f_tx = make_fastdiv(a.tiles_x);
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_bf16<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_bf16<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    attr_set.set(dev);
  }
  const long items = (long)a.tiles * a.ntn;
  long blocks = 2L * device_cu_count();
  if (blocks > items) blocks = items;
  blocks -= blocks % a.ntn;
  if (c.pool) hipLaunchKernelGGL(conv3x3_c64_bf16<true>, dim3((unsigned)blocks), dim3(256), kLdsBytes, s, a);
  else hipLaunchKernelGGL(conv3x3_c64_bf16<false>, dim3((unsigned)blocks), dim3(256), kLdsBytes, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace rtpose

extern "C" {

#ifdef RTPOSE_EXP_C64_TIMELINE
int rtpose_exp_c64_timeline(unsigned long long* out) {  // [64 blocks][24 tiles][6 stamps]
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rtpose::c64::g_c64_tl), sizeof(unsigned long long) * 64 * 24 * 6);
}
#endif

int rtpose_conv3x3_c64_bf16_fits(const rtpose_conv_desc* d, int ngroups, int N, int H, int W) {
  return rtpose::conv_c64_bf16_fits(d, ngroups, N, H, W, 0, 0);
}

int rtpose_conv3x3_c64_bf16(const rtpose_conv_desc* d, int N, int H, int W, void* stream) {
  return rtpose::conv_c64_bf16_launch(d, N, H, W, rtpose::as_stream(stream));
}

}  // extern "C"
