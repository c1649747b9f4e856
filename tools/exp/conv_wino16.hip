// EXPERIMENT RECORD (round 3, not part of the library): the F(2x2,3x3) kernel re-tiled for two waves per SIMD on
// 16 x 16 x 4 MFMAs.  Measured 3.4 % faster than wino_f32<1,4,16> over the 13 launches it covers (11.30 -> 10.92 ms of
// 3x3 time at batch 32; tools/r3_sessions/session11.sh, profiles/r03_wino16_experiment.txt), then superseded by the F(4x4,3x3)
// form (csrc/conv_wino4.hip, 8.8 ms), which kept its wave layout ideas.  To build it: add it to SRCS of csrc/Makefile and
// route conv2d_wino_launch's wm == 1 case to conv2d_wino16_launch.
// fp32 Winograd F(2x2, 3x3) convolution on 16 x 16 matrix-core tiles, TWO WAVES PER SIMD, for gfx950 (MI355X): the
// large-grid form of the 3x3 convs whose padded output width is a multiple of 128 (13 launches of an rtpose_vgg
// forward: conv2_1 .. conv4_4_CPM and the stage-1 convs; lib/network/rtpose_vgg.py:23-35, :95-105).  Same module
// boundary, transforms, filter packing and V layout in LDS as wino_f32 (conv_wino.hip) - what changes is who multiplies:
//
//  * wino_f32 gives a wave all 16 frequencies of a 32-wtile x 32-column tile: 16 v_mfma_f32_32x32x2_f32 accumulators =
//    256 registers, i.e. ONE wave per SIMD, and one wave issues a 64-cycle MFMA every ~70 cycles at best and nothing
//    while it does anything else (tools/exp/mfma_issue16.hip: 70.4 cycles per MFMA alone, 65.4 with a sibling; the
//    16 x 16 x 4 shape: 35.0 / 32.7 for 32).  Round 3's timelines put its multiply loop within 4 % of that floor.
//  * here a wave owns all 16 frequencies of 32 wtiles x 16 columns on v_mfma_f32_16x16x4_f32 (2 row tiles x 16
//    frequencies x 4 registers = 128 accumulator registers), a block has 8 waves = 32 wtiles x 128 columns, two per
//    SIMD.  The output transform stays lane-local (a lane holds 4 consecutive wtiles of one column, twice).
//  * the input transform is cut into QUARTER patches so that it fits the 128 registers left: thread (q = tid / 128)
//    loads two rows of a 4 x 4 patch of 4 channels (8 x 16 bytes) and forms the four frequencies fy = q - 16 packed
//    instructions and 4 LDS writes per thread and chunk, all 512 threads; row 2 is fetched by three of the four
//    quarters (+33 % patch bytes, L1 / L2 hits).
//  * per frequency a wave issues 8 MFMAs (2 row tiles x 4 k quads of a 16-channel chunk); between them, pinned: the A
//    fragments of the next frequency (2 ds_read_b128), the B fragment 6 frequencies ahead (one 16-byte buffer load
//    per lane: 16 columns x 16 channels), and on the first frequencies of a chunk the transform of the next chunk / the
//    patch loads of the one after.  The chunk pipeline and the B ring run across the tiles of a persistent block as in
//    wino_f32.
// A k quad of the 16 x 16 x 4 MFMA contracts the channels {4 kq + j : kq = 0..3} of a chunk (lane kq holds the 16-byte
// group kq, MFMA j uses element j): the sums are grouped differently from wino_f32's k pairs, so the results differ
// from it by rounding - the small-grid form of THIS arithmetic is wino16s_f32 below (bit-identical to wino16_f32).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "common.h"
#include "conv_exp.h"
#include "wino_common.h"

namespace rtpose {

namespace wino16 {

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
  size_t in_bytes, w_bytes, out_bytes;
};

struct Args {
  Group g[2];
  int N, H, W;
  int TY, TX, T;
  int cin;
  int relu, pool;
  int mtiles, ntiles, ncombo, xcd_remap;
  int persist;
};

constexpr int NT = 32;   // wtiles per block
constexpr int CK = 16;   // channels per chunk
constexpr int CG = 4;    // 16-byte channel groups per chunk = the k lanes of a 16 x 16 x 4 MFMA
constexpr int NB = RTPOSE_W16_NB;    // B ring entries (one per frequency)
constexpr int PF = RTPOSE_W16_PF;    // B prefetch distance in frequencies
constexpr int VBUF = 16 * CG * NT;  // float4 per V buffer

__global__ __launch_bounds__(512, 1) void wino16_f32(const Args A) {
  extern __shared__ __attribute__((aligned(16))) float4 V4[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = 16-column slice of the block's 128
  const int r16 = lane & 15, kq = lane >> 4;

  // ---- block -> (n tile, group) and its m tiles (as wino_f32) ----------------------------------------------------
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
    jstep = A.mtiles;
  }
  if (j0 >= A.mtiles) return;
  const int nt = c % A.ntiles, grp = c / A.ntiles;
  const Group g = grp ? A.g[1] : A.g[0];
  const int TT = A.TY * A.TX;

  // ---- input transform role: quarter q of the patch of (wtile tl, channel group cg) -------------------------------
  // B^T over the patch rows d0..d3:  fy 0: d0 - d2,  1: d1 + d2,  2: d2 - d1,  3: d1 - d3.  Quarter q loads the rows
  // (ra, rb) = (0,2), (1,2), (2,1), (1,3) and forms t = Ra + sgn Rb (sgn = -1, +1, -1, -1): one instruction stream.
  // (channel-group-major: the 8 contiguous lanes a ds_write_b128 is serviced in write 8 consecutive slots of a plane)
  const int q = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int idx = tid & 127, tl = idx & 31, cg = idx >> 5;
  const float sgn = q == 1 ? 1.f : -1.f;
  const i32x4 rw = make_rsrc(g.w, g.w_bytes);
  i32x4 rin, rin_nx;
  unsigned pvoff, pvoff_nx;
  auto set_loader = [&](int mt, i32x4& r_, unsigned& v_) {
    auto patch_q = [&](int t) -> size_t {
      const int n = t / TT, r = t - n * TT;
      const int ty = r / A.TX, tx = r - ty * A.TX;
      return (size_t)g.in_lead + (size_t)(n * g.in_hs + 2 * ty - 1) * g.in_ws + (2 * tx - 1);
    };
    const size_t q0 = patch_q(min(mt * NT, A.T - 1));
    const size_t qq = patch_q(min(mt * NT + tl, A.T - 1));
    const size_t o0 = q0 * g.in_cstride + g.in_choff;
    r_ = make_rsrc(g.in + o0, g.in_bytes - o0 * 4);
    v_ = (unsigned)(((qq - q0) * g.in_cstride + cg * 4) * 4);
  };
  set_loader(j0, rin, pvoff);
  unsigned psoff[2];
  {
    const int ra = q == 0 ? 0 : q == 2 ? 2 : 1, rb = q == 2 ? 1 : q == 3 ? 3 : 2;
    psoff[0] = (unsigned)(ra * g.in_ws * g.in_cstride * 4);
    psoff[1] = (unsigned)(rb * g.in_ws * g.in_cstride * 4);
  }
  const unsigned pxb = (unsigned)g.in_cstride * 4;
  F4 p[2][4], t4[4];
  auto load_piece_from = [&](const i32x4& r_, unsigned v_, int chunk, int i) {
    p[i >> 2][i & 3] = bload(r_, v_, (unsigned)chunk * (CK * 4) + psoff[i >> 2] + (i & 3) * pxb);
  };
  const int vst = (q * 4 * CG + cg) * NT + tl;  // V[f = 4 q + fx][cg][wtile]
  auto tgroup = [&](int buf, int gidx) {
    if (gidx == 0) {
#pragma unroll
      for (int x = 0; x < 4; ++x) t4[x] = fma4(sgn, p[1][x], p[0][x]);
    } else {
      float4* v = V4 + buf * VBUF + vst;
#pragma unroll
      for (int fx = 0; fx < 4; ++fx)
        v[fx * CG * NT] = to_float4(fx == 0 ? sub4(t4[0], t4[2]) : fx == 1 ? add4(t4[1], t4[2]) : fx == 2 ? sub4(t4[2], t4[1])
                                                                                            : sub4(t4[1], t4[3]));
    }
  };

  // ---- MFMA roles ---------------------------------------------------------------------------------------------------
  const int ncol = nt * 128 + wn * 16 + r16;
  floatx4 acc[16][2];  // [frequency][row tile of 16 wtiles]
  const float bias0 = g.bias[ncol];
  // B: lane offset (k lane, column) in one register; the frequency / chunk in the scalar offset
  const unsigned boff = (unsigned)((kq * g.cout_pad + ncol) * 16);
  const unsigned fstep = (unsigned)(CG * g.cout_pad * 16);  // bytes per (chunk, frequency) block
  unsigned wso = 0;
  float4 bs[NB];
  const int nchunks = A.cin / CK;  // >= 2 (host)

  int par = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) load_piece_from(rin, pvoff, 0, i);
  tgroup(0, 0);
  tgroup(0, 1);
#pragma unroll
  for (int i = 0; i < 8; ++i) load_piece_from(rin, pvoff, 1, i);
#pragma unroll
  for (int f = 0; f < PF; ++f) {
    bs[f] = bload_f4(rw, boff, wso);
    wso += fstep;
  }
  __syncthreads();

#define RTPOSE_PIN()             \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)
  for (int mt = j0; mt < A.mtiles; mt += jstep) {
    const bool has_next = mt + jstep < A.mtiles;
    if (has_next) {
      set_loader(mt + jstep, rin_nx, pvoff_nx);
    } else {
      rin_nx = rin;
      pvoff_nx = pvoff;
    }
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[f][rt][v] = f == 5 ? bias0 : 0.f;

    float4 a[2][2];  // [frequency parity][row tile]
#pragma unroll 1
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      // A: plane kq, wtile rt * 16 + r16 of frequency f
      const float4* va = V4 + par * VBUF + kq * NT + r16;
      const int nbuf = par ^ 1;
      const bool nx = chunk + 2 >= nchunks;
      const int c2 = nx ? chunk + 2 - nchunks : chunk + 2;
      const i32x4 rl = nx ? rin_nx : rin;
      const unsigned pvl = nx ? pvoff_nx : pvoff;
      a[0][0] = va[0];
      a[0][1] = va[16];
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        const float4 bv = bs[f % NB];
        const float b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const float4 av = a[f & 1][rt];
          const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[f][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b4[j], acc[f][rt], 0, 0, 0);
          RTPOSE_PIN();
          if (rt == 0) {
            if (f < 15) {  // A of the next frequency (the first one of a chunk is read after the barrier)
              a[(f + 1) & 1][0] = RTPOSE_EXP_A(va[(f + 1) * CG * NT], a[f & 1][0]);
              a[(f + 1) & 1][1] = RTPOSE_EXP_A(va[(f + 1) * CG * NT + 16], a[f & 1][1]);
            }
          } else {
            // B PF frequencies ahead; PF before the end of a tile's last chunk the ring wraps to the next tile
            bs[(f + PF) % NB] = RTPOSE_EXP_B(bload_f4(rw, boff, wso), bs[f % NB]);
            wso = (f == 15 - PF && chunk == nchunks - 1) ? 0u : wso + fstep;
            if (RTPOSE_EXP_STAGE) {
              if (f < 2) tgroup(nbuf, f);                         // transform of the next chunk
              else if (f < 10) load_piece_from(rl, pvl, c2, f - 2);  // patch rows of the chunk after it
            }
          }
          RTPOSE_PIN();
        }
      }
      __syncthreads();
      par ^= 1;
    }
    rin = rin_nx;
    pvoff = pvoff_nx;

    // ---- epilogue: output transform A^T M A, (+ReLU) (+2x2 max-pool), masked stores -------------------------------
    // register v of acc[f][rt] = wtile rt * 16 + 4 kq + v of the tile, column ncol: 16 lanes = 64 contiguous bytes
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
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        int tcur = mt * NT + rt * 16 + 4 * kq;
        int sn = tcur / TT, sy, sx;
        {
          const int r = tcur - sn * TT;
          sy = r / A.TX;
          sx = r - sy * A.TX;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float s[4][2];
#pragma unroll
          for (int fy = 0; fy < 4; ++fy) {
            s[fy][0] = acc[fy * 4 + 0][rt][v] + acc[fy * 4 + 1][rt][v] + acc[fy * 4 + 2][rt][v];
            s[fy][1] = acc[fy * 4 + 1][rt][v] - acc[fy * 4 + 2][rt][v] - acc[fy * 4 + 3][rt][v];
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
            const float vv = fmaxf(fmaxf(y[0][0], y[0][1]), fmaxf(y[1][0], y[1][1]));
            bstore(vv, rout, ok ? off : kNoStore, 0);
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
  }  // m tiles of this block
#undef RTPOSE_PIN
}

}  // namespace wino16

// 1 when the 3x3 conv has a two-waves-per-SIMD instance: 128-column tiles, 16-channel chunks, >= 2 chunks
int conv2d_wino16_ok(int cin, int cout) { return cout_pad(cout) % 128 == 0 && cin % 16 == 0 && cin >= 32; }

int conv2d_wino16_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, hipStream_t s) {
  using namespace wino16;
  const rtpose_conv_desc& d0 = d[0];
  Args a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < ngroups; ++i) {
    const rtpose_conv_desc& di = d[i];
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
  a.T = N * a.TY * a.TX;
  a.cin = d0.cin;
  a.relu = d0.relu;
  a.pool = d0.pool;
  a.mtiles = ceil_div(a.T, NT);
  a.ntiles = cout_pad(d0.cout) / 128;
  a.ncombo = a.ntiles * ngroups;
  a.xcd_remap = (a.ncombo > 1 && a.mtiles >= 64) ? 1 : 0;
  long ids = a.xcd_remap ? (long)8 * a.ncombo * ceil_div(a.mtiles, 8) : (long)a.mtiles * a.ncombo;
  const int n_cu = device_cu_count();
  if ((long)a.mtiles * a.ncombo > n_cu && n_cu % a.ncombo == 0) {
    a.persist = 1;
    ids = n_cu;
  }
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino16_f32),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set.set(dev);
  }
  hipLaunchKernelGGL(wino16_f32, dim3((unsigned)ids), dim3(512), (size_t)2 * VBUF * sizeof(float4), s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace rtpose
