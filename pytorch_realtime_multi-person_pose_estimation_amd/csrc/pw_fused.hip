// Fused pointwise (1x1) GEMM for the ShuffleNetV2 pose network (BASELINE configs[3]) — fp32,
// v_mfma_f32_32x32x2_f32, gfx950.
//
// Stands in for the module chains of lib/network/rtpose_shufflenetV2.py BasicBlock (:22-63):
//     conv_bn_relu 1x1                                              (conv.0, conv0.1, conv5, heads)
//     conv_bn depthwise 3x3  ->  conv_bn_relu 1x1                   (conv.1 -> conv.2, conv0.0 -> conv0.1)
//     ... -> torch.cat((x1, x2), 1) -> channel_shuffle(2)           (:56-62)
// as ONE launch per chain:
//
//   * GEMM view  M = pixels, N = output channels, K = input channels.  Block tile 64(M) x up to
//     256(N): the four waves split N (1 x 4 waves of 64 x {32,64}; 2 x 2 of 32 x 32 for 64-column
//     layers), so the A tile of a pixel strip is staged ONCE for all output channels (the generic conv
//     kernel used 128 x 64 tiles: the A tile was re-staged by every N tile and a K-chunk barrier sat
//     between every 16 MFMAs of a wave - these K = 24..464 GEMMs ran at 24-70 TFLOP/s).
//   * K is walked in 32-channel chunks through a double-buffered LDS tile [8 planes][66 pixels][4
//     floats] (one ds_read_b128 per A fragment, the generic kernel's image); the next chunk's pieces
//     are in flight in registers while the current chunk is multiplied (64 MFMAs per wave per barrier
//     at N = 256).  B (weights, packed [K/4][coutp][4]) streams L2 -> registers one k-group ahead.
//   * DW: the A tile is PRODUCED in the kernel: the chunk's input halo (the strip + one row above /
//     below, contiguous in the shared-gap layout) is staged in LDS, the depthwise 3x3 (+ folded BN
//     bias) is evaluated on the VALU into the A tile, then multiplied.  The depthwise activation
//     tensor never exists in HBM and its launch is gone.
//   * PT: the block also copies the pass-through half x1 of its pixels to the even channel slots of the
//     output buffer while its GEMM result goes to the odd slots through out_cmap - cat + channel_shuffle
//     cost no launch and no extra pass.
//   * 2 blocks per CU (<= 63 KB LDS, <= 256 VGPRs): one block's staging / depthwise / epilogue phases
//     run under the other's MFMAs.
#include <hip/hip_runtime.h>

#include "common.h"

namespace rtpose {

typedef float pw_f2 __attribute__((ext_vector_type(2)));

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef const floatx4 __attribute__((address_space(1)))* gcf4_t;
__device__ __forceinline__ float4 pw_gload4(const void* p) {  // explicit global address space (no FLAT loads)
  const floatx4 v = *(gcf4_t)(unsigned long long)(p);
  return make_float4(v[0], v[1], v[2], v[3]);
}

struct PwView {  // one activation tensor (shared-gap padded NHWC slice)
  const float* base;
  int cstride, choff, ws, hs, lead;
};

struct PwArgs {
  PwView in;          // A source: the GEMM input (DW = 0) or the depthwise conv's input (DW = 1, gap >= 1)
  const float* dw_w;  // DW: [9][K] taps (ky*3+kx major) and
  const float* dw_b;  //     [K] bias of the depthwise conv (BN folded)
  const float* w;     // packed pointwise weights [K/4][coutp][4]
  const float* bias;  // [coutp]
  float* out;
  int out_cstride, out_choff, out_ws, out_hs, out_lead;
  const int32_t* out_cmap;  // optional [coutp]: column n -> absolute channel (< 0: not stored)
  PwView pt;                // PT: source of the pass-through channels (same N, H, W)
  const int32_t* pt_cmap;   // PT: [pt_c] source channel i -> absolute output channel
  int pt_c;
  // PT, interleave form (pt_pairs > 0, pt_cmap unused): output channel j < 2 pt_pairs takes source channel
  // pt_a + j/2 (j even) or pt_b + j/2 (j odd) and lands at (j < pt_split ? pt_d0 + j : pt_d1 + j - pt_split)
  int pt_pairs, pt_a, pt_b, pt_split, pt_d0, pt_d1;
  const int32_t* in_planes;  // optional [K/4]: channel offset (inside the pixel) of every 4-channel plane of K
  int N, H, W, M;
  int K, coutp, cout, relu;
  int nps;  // DW: LDS plane stride (pixels) of the staged halo
  // divisions by launch constants (H W, W, passes per strip, tiles per row / column): a work item's pixel arithmetic was
  // ~300 of the ~700 VALU instructions a wave issues outside its multiply loop - on ALUs it shares with the fp32 MFMAs
  FastDiv fHW, fW, fnp, ftx, fty;
};

constexpr int kPwBM = 64;   // pixels per block
constexpr int kPwQS = 66;   // LDS plane stride of the A tile (pixels): planes 8 banks apart
constexpr int kPwPL = 8;    // 16-byte channel-group planes per K chunk (32 channels)
constexpr int kPwMaxK = 1024; // largest K (s_plane table)
constexpr int kPwMaxStage = 4;  // DW: staged 16-byte pieces per thread per chunk (the 10 x 10 halo of an 8 x 8 tile)
constexpr int kPwTile = 8;      // DW: the block's 64 pixels are an 8 x 8 tile (halo 100 pixels = 1.56x; a 64-pixel strip of
                                // a 46-wide map needed 209 = 3.3x)
constexpr int kPwHalo = kPwTile + 2;

#define RTPOSE_PW_PIN()          \
  asm volatile("" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)

struct PwPix {  // pixel m of the batch as (image, row, column)
  int n, y, x;
};
__device__ __forceinline__ PwPix pw_pix(int m, int HW, int W, const FastDiv fHW, const FastDiv fW) {
  PwPix p;
  p.n = fast_div(m, fHW);
  const int r = m - p.n * HW;
  p.y = fast_div(r, fW);
  p.x = r - p.y * W;
  return p;
}
__device__ __forceinline__ int pw_q(const PwPix& p, int lead, int hs, int ws) { return lead + (p.n * hs + p.y) * ws + p.x; }

// WM x WN waves (WM * WN = 4), wave tile (32 MF) x (32 NFW), WM * MF = 2.
// PERSISTENT: the grid is 2 blocks per CU; a block walks work items (64-pixel strip, BN-column pass)
// blockIdx.x, + gridDim.x, ...  The first K chunk of the NEXT item is requested while the last chunk of
// the current one is multiplied, so an item's prologue (one exposed memory round trip) and its
// epilogue stores run under MFMAs instead of in front of / behind them.
template <int WM, int MF, int NFW, bool DW>
__global__ __launch_bounds__(256, 2) void pw_gemm_f32(const PwArgs A) {
  constexpr int WN = 4 / WM;
  constexpr int BN = WN * NFW * 32;
  static_assert(WM * MF * 32 == kPwBM, "block tile is 64 pixels");
  extern __shared__ __attribute__((aligned(16))) float4 smem4[];
  __shared__ int s_qout[2][kPwBM], s_qpt[2][kPwBM];  // per work item (parity): output / pass-through pixel of row r
  __shared__ int s_plane[kPwMaxK / 4];               // channel offset of every 4-channel plane of K

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int l31 = lane & 31, kh = lane >> 5;
  const int HW = A.H * A.W;
  // staging role: channel-group plane, first pixel.  8 consecutive lanes = 4 planes x 2 pixels (not 8 planes of one pixel): a
  // ds_write_b128 is serviced 8 lanes at a time on 32 banks, and with the planes 32 bytes apart (mod 256: what the 16-lane
  // reads want) the planes p and p + 4 of one pixel met in the same banks - every park / A-tile write 2-way conflicted
  // (SQ_LDS_BANK_CONFLICT 23 % of SQ_LDS_IDX_ACTIVE in the plain launches, 38 % in the depthwise ones)
  const int pl = (tid & 3) | ((tid >> 1) & 4), px = ((tid >> 4) << 1) | ((tid >> 2) & 1);

  // ---- LDS carve-up (float4 units) ---------------------------------------------------------
  constexpr int ASUB = kPwPL * kPwQS;  // one A chunk buffer
  float4* a_lds = smem4;               // [2][ASUB]
  float4* st = smem4 + 2 * ASUB;       // DW: [kPwPL][nps] staged halo (single buffer: parked after the
                                       //     previous chunk's depthwise phase, which a barrier closes)
  const int nps = A.nps;
  float4* dwl = st + kPwPL * nps;      // DW: [10][K/4] depthwise taps + bias, loaded once per block

  const int nch = (A.K + 31) >> 5;
  const int gtot = A.K >> 3;  // 8-channel k-groups in all
  const int npass = A.coutp / BN;
  const int tiles_x = (A.W + kPwTile - 1) / kPwTile, tiles_y = (A.H + kPwTile - 1) / kPwTile;  // DW: 8 x 8 tiles of one image
  const int nwork = (DW ? A.N * tiles_y * tiles_x : (A.M + kPwBM - 1) / kPwBM) * npass;
  const float4* w4 = reinterpret_cast<const float4*>(A.w);
  // Addresses are a uniform (scalar) base + a 32-bit per-lane element offset: the loads take the
  // SGPR-base form and no 64-bit address pairs are kept per piece (the host checks that every tensor
  // is below 2^31 floats).
  const float* in_base = A.in.base + A.in.choff;

  // Per work item, per thread: where its staged pieces come from.
  //   DW = 0: piece u = (pixel px + 32 u, plane pl) of the strip, u < 2          -> q[u] = pixel index
  //   DW = 1: the strip's halo = pixels [q_org, q_org + np) x 8 planes, piece u = (px + 32 u, pl), u < n_st;
  //           sp[u] = halo-relative index of the thread's two depthwise output pixels
  struct Item {
    int m0, pass;    // DW = 0: first pixel of the strip
    int n, y0, x0;   // DW = 1: image and first pixel of the 8 x 8 tile
    int q0, q1;      // DW = 0: the thread's two pixels;  DW = 1: q0 = pixel index of the halo's corner (y0-1, x0-1)
  };
  auto setup = [&](int wi) -> Item {
    Item it;
    const int tile = fast_div(wi, A.fnp);
    it.pass = wi - tile * npass;
    it.m0 = tile * kPwBM;
    it.n = it.y0 = it.x0 = 0;
    if (DW) {
      const int r = fast_div(tile, A.ftx), tx = tile - r * tiles_x;
      it.n = fast_div(r, A.fty);
      const int ty = r - it.n * tiles_y;
      it.y0 = ty * kPwTile;
      it.x0 = tx * kPwTile;
      it.q0 = A.in.lead + (it.n * A.in.hs + it.y0 - 1) * A.in.ws + it.x0 - 1;  // >= 0: lead = ws + 1
      it.q1 = 0;
    } else {
      const int ma = min(it.m0 + px, A.M - 1), mb = min(it.m0 + px + 32, A.M - 1);  // rows past the end replay the last pixel
      it.q0 = pw_q(pw_pix(ma, HW, A.W, A.fHW, A.fW), A.in.lead, A.in.hs, A.in.ws);
      it.q1 = pw_q(pw_pix(mb, HW, A.W, A.fHW, A.fW), A.in.lead, A.in.hs, A.in.ws);
    }
    return it;
  };
  auto write_tables = [&](const Item& it, int par) {  // threads 0..63: output / pass-through pixel of tile row tid
    if (DW) {
      const int y = it.y0 + (tid >> 3), x = it.x0 + (tid & 7);
      const bool ok = y < A.H && x < A.W;
      const int yc = min(y, A.H - 1), xc = min(x, A.W - 1);
      s_qout[par][tid] = ok ? A.out_lead + (it.n * A.out_hs + yc) * A.out_ws + xc : -1;
      s_qpt[par][tid] = A.pt.base ? A.pt.lead + (it.n * A.pt.hs + yc) * A.pt.ws + xc : 0;
    } else {
      const int m = it.m0 + tid;
      const int mc = min(m, A.M - 1);
      const PwPix pp = pw_pix(mc, HW, A.W, A.fHW, A.fW);
      s_qout[par][tid] = m < A.M ? pw_q(pp, A.out_lead, A.out_hs, A.out_ws) : -1;
      s_qpt[par][tid] = A.pt.base ? pw_q(pp, A.pt.lead, A.pt.hs, A.pt.ws) : 0;
    }
  };
  // DW: halo pixel (hy, hx) of the tile, hp = 10 hy + hx, sits hy * ws + hx pixels after the corner: the
  // thread's staged pieces are hp = px + 32 u (clamped to 99) - the same offsets for every tile
  int hoff[DW ? kPwMaxStage : 1];
  if (DW) {
#pragma unroll
    for (int u = 0; u < kPwMaxStage; ++u) {
      const int hp = min(px + 32 * u, kPwHalo * kPwHalo - 1);
      hoff[u] = (hp / kPwHalo) * A.in.ws + hp % kPwHalo;
    }
  }
  // the thread's two depthwise output pixels are tile rows px and px + 32: halo index of their tap (0, 0)
  const int sp0 = (px >> 3) * kPwHalo + (px & 7), sp1 = sp0 + 4 * kPwHalo;
  // staged pieces of channel chunk c0 of item `it` -> registers.  Branch-free: every thread always issues
  // all its loads, pixels past the halo / channel groups past K are clamped to valid addresses (their LDS
  // slots exist and are never multiplied) - per-piece predicates put every load in its own basic block,
  // which cost ~90 VGPRs and the load/MFMA interleaving.
  float4 sr[DW ? kPwMaxStage : 2];
  auto load_pieces = [&](const Item& it, int c0) {
    const unsigned cofs = (unsigned)s_plane[min((c0 >> 2) + pl, (A.K >> 2) - 1)];
    if (DW) {
#pragma unroll
      for (int u = 0; u < kPwMaxStage; ++u)
        sr[u] = pw_gload4(in_base + ((unsigned)(it.q0 + hoff[u]) * (unsigned)A.in.cstride + cofs));
    } else {
      sr[0] = pw_gload4(in_base + ((unsigned)it.q0 * (unsigned)A.in.cstride + cofs));
      sr[1] = pw_gload4(in_base + ((unsigned)it.q1 * (unsigned)A.in.cstride + cofs));
    }
  };

  int wi = blockIdx.x;
  if (wi >= nwork) return;
  for (int j = tid; j < (A.K >> 2); j += 256) s_plane[j] = A.in_planes ? A.in_planes[j] : 4 * j;
  __syncthreads();
  Item cur = setup(wi);
#pragma unroll
  for (int u = 0; u < (DW ? kPwMaxStage : 2); ++u) sr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < kPwBM) write_tables(cur, 0);
  load_pieces(cur, 0);
  if (DW) {  // (visible after the first chunk's barrier)
    const int k4 = A.K >> 2;
    for (int i = tid; i < 10 * k4; i += 256)
      dwl[i] = i < 9 * k4 ? pw_gload4(A.dw_w + 4 * (size_t)i) : pw_gload4(A.dw_b + 4 * (size_t)(i - 9 * k4));
  }
  int par = 0;   // work-item parity (tables)
  int lbuf = 0;  // LDS chunk-buffer parity, runs on across work items
  // Plain pointwise variants with <= 128 columns (BAHEAD) fetch B a whole CHUNK ahead (4 k-groups, 16 VGPRs more):
  // the counter behind s_waitcnt is in order, so with B only one k-group ahead every wait for a weight
  // fragment also waited for the chunk's staging loads (activations from HBM, issued just before it) - the
  // waves stalled on memory in the middle of every chunk.  Now everything a chunk's MFMAs read was requested
  // before the previous chunk was multiplied (stage-3 conv.0 0.115 -> 0.100 ms, heads 0.46 -> 0.37 ms).  The
  // 256-column variant does not have the registers: with 32 more it spilled inside the tap loop (0.31 -> 0.44 ms).
  constexpr bool BAHEAD = !DW && NFW == 1;
  float4 bw_cur[BAHEAD ? 4 : 1][NFW], bw_nxt[BAHEAD ? 4 : 1][NFW];
  if (BAHEAD) {
    const unsigned wl0 = (unsigned)(kh * A.coutp + cur.pass * BN + wn * (32 * NFW) + l31);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int fn = 0; fn < NFW; ++fn) {
        bw_cur[g][fn] = pw_gload4(w4 + (size_t)(2 * min(g, gtot - 1) * A.coutp) + (wl0 + fn * 32));
        bw_nxt[g][fn] = bw_cur[g][fn];
      }
  }

  while (true) {
    const int wnext = wi + gridDim.x;
    const bool has_next = wnext < nwork;
    Item nxt = cur;
    if (has_next) nxt = setup(wnext);
    const int ncol = cur.pass * BN + wn * (32 * NFW) + l31;  // column of n-fragment 0 of this lane
    const unsigned w_lane = (unsigned)(kh * A.coutp + ncol);  // float4 index of this lane in a k-group's two planes

    // B fragments of k-group 0 (their latency hides behind the first chunk's staging + barrier)
    float4 bcur[NFW], bnxt[NFW];
#pragma unroll
    for (int fn = 0; fn < NFW; ++fn) {
      bcur[fn] = pw_gload4(w4 + (w_lane + fn * 32));
      bnxt[fn] = bcur[fn];
    }
    floatx16 acc[MF][NFW];
#pragma unroll
    for (int fn = 0; fn < NFW; ++fn) {
      const float b0 = A.bias[ncol + fn * 32];
#pragma unroll
      for (int fm = 0; fm < MF; ++fm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[fm][fn][r] = b0;
    }

    for (int c = 0; c < nch; ++c) {
      const int c0 = c << 5;
      float4* a_buf = a_lds + lbuf * ASUB;
      const bool last = c + 1 == nch;
      if (DW) {
        // park the staged halo pieces, request the next chunk's (of this item, or chunk 0 of the next)
#pragma unroll
        for (int u = 0; u < kPwMaxStage; ++u) st[pl * nps + px + 32 * u] = sr[u];  // (nps >= 128 slots)
        __syncthreads();  // halo of chunk c visible; every wave is past the previous item's epilogue
        if (c == 0 && has_next && tid < kPwBM) write_tables(nxt, par ^ 1);
        // depthwise 3x3 (+bias) -> A tile.  Tap (ky, kx) of pixel q is pixel q + (ky-1) ws + (kx-1).
        const int k4 = A.K >> 2;
        const float4* wl = dwl + min((c0 >> 2) + pl, k4 - 1);  // (groups past K: their A planes are not read)
        const float4 vb = wl[9 * k4];                          // bias
        pw_f2 v0lo = {vb.x, vb.y}, v0hi = {vb.z, vb.w}, v1lo = v0lo, v1hi = v0hi;
        const float4* s0 = st + pl * nps + sp0;
        const float4* s1 = st + pl * nps + sp1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float4 ww = wl[(ky * 3 + kx) * k4];
            const float4 x0 = s0[ky * kPwHalo + kx], x1 = s1[ky * kPwHalo + kx];
            // packed fused multiply-adds (v_pk_fma_f32): fp32 VALU instructions take their cycles from the ALUs the
            // fp32 MFMAs run on - 4 instead of 16 per tap
            const pw_f2 wlo = {ww.x, ww.y}, whi = {ww.z, ww.w};
            v0lo = __builtin_elementwise_fma(pw_f2{x0.x, x0.y}, wlo, v0lo);
            v0hi = __builtin_elementwise_fma(pw_f2{x0.z, x0.w}, whi, v0hi);
            v1lo = __builtin_elementwise_fma(pw_f2{x1.x, x1.y}, wlo, v1lo);
            v1hi = __builtin_elementwise_fma(pw_f2{x1.z, x1.w}, whi, v1hi);
          }
          RTPOSE_PW_PIN();  // one stencil row (9 LDS reads) at a time: all 27 at once cost 108 VGPRs
        }
        a_buf[pl * kPwQS + px] = make_float4(v0lo.x, v0lo.y, v0hi.x, v0hi.y);
        a_buf[pl * kPwQS + px + 32] = make_float4(v1lo.x, v1lo.y, v1hi.x, v1hi.y);
        // the next chunk's halo (of this item, or chunk 0 of the next; no next item: nxt == cur, a harmless
        // re-read) is requested only now: the depthwise phase above is the register peak of the kernel, and
        // the MFMAs below still cover the latency
        load_pieces(last ? nxt : cur, last ? 0 : c0 + 32);
        __syncthreads();  // A tile of chunk c visible; the staged halo may be overwritten
      } else {
        a_buf[pl * kPwQS + px] = sr[0];
        a_buf[pl * kPwQS + px + 32] = sr[1];
        if (BAHEAD) {  // B of the next chunk (of this item, or chunk 0 of the next item's columns), BEFORE the staging loads
          const int gnext = last ? 0 : 4 * (c + 1);
          const unsigned wl = (unsigned)(kh * A.coutp + (last ? nxt.pass : cur.pass) * BN + wn * (32 * NFW) + l31);
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int fn = 0; fn < NFW; ++fn)
              bw_nxt[g][fn] = pw_gload4(w4 + (size_t)(2 * min(gnext + g, gtot - 1) * A.coutp) + (wl + fn * 32));
        }
        load_pieces(last ? nxt : cur, last ? 0 : c0 + 32);  // (no next item: nxt == cur, a harmless re-read)
        __syncthreads();  // A tile of chunk c visible (the other buffer was last read before the previous barrier)
        if (c == 0 && has_next && tid < kPwBM) write_tables(nxt, par ^ 1);
      }
      lbuf ^= 1;

      // ---- multiply chunk c: ng k-groups of 8 channels, 4 MFMAs per (m, n) fragment pair each ----
      // Operands of the NEXT k-group are requested before the current group's MFMAs are issued (B from
      // L2 - also across the chunk boundary - A from LDS) and consumed one group later.  A full chunk is
      // straight-line code with the two register sets alternating: no moves, no branches around the
      // loads, so the s_waitcnt the compiler places in front of a group's MFMAs leaves the younger
      // requests in flight (in a rolled loop with guards it drained vmcnt to 0 at every group).
      const int ng = min(4, gtot - 4 * c);
      const float4* a_rd = a_buf + kh * kPwQS + wm * (32 * MF) + l31;
#define RTPOSE_PW_BLOAD(DST, GG)                                                              \
  _Pragma("unroll") for (int fn = 0; fn < NFW; ++fn)                                          \
      DST[fn] = pw_gload4(w4 + (size_t)(2 * (GG) * A.coutp) + (w_lane + fn * 32))
#define RTPOSE_PW_ALOAD(DST, GI) \
  _Pragma("unroll") for (int fm = 0; fm < MF; ++fm) DST[fm] = a_rd[2 * (GI) * kPwQS + fm * 32]
#define RTPOSE_PW_MUL(AV, BV)                                                                           \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                       \
    _Pragma("unroll") for (int fn = 0; fn < NFW; ++fn) {                                                \
      const float bv_[4] = {BV[fn].x, BV[fn].y, BV[fn].z, BV[fn].w};                                    \
      _Pragma("unroll") for (int fm = 0; fm < MF; ++fm) {                                               \
        const float av_[4] = {AV[fm].x, AV[fm].y, AV[fm].z, AV[fm].w};                                  \
        acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_[j], bv_[j], acc[fm][fn], 0, 0, 0);       \
      }                                                                                                 \
    }                                                                                                   \
  }
      // (no rolled loop for the short last chunk of K = 24, 120, 232, 464: a second code region made the
      //  compiler keep the 64 accumulator registers twice - guarded straight-line groups instead; loads
      //  of groups past K are clamped to valid memory and never multiplied)
      const int gg = 4 * c, gl = gtot - 1;
      float4 a0[MF], a1[MF];
      RTPOSE_PW_ALOAD(a0, 0);
      if (BAHEAD) {
        RTPOSE_PW_ALOAD(a1, 1);
        RTPOSE_PW_PIN();
        RTPOSE_PW_MUL(a0, bw_cur[0]);
        RTPOSE_PW_PIN();
        if (ng > 1) {
          RTPOSE_PW_ALOAD(a0, 2);
          RTPOSE_PW_PIN();
          RTPOSE_PW_MUL(a1, bw_cur[1]);
          RTPOSE_PW_PIN();
        }
        if (ng > 2) {
          RTPOSE_PW_ALOAD(a1, 3);
          RTPOSE_PW_PIN();
          RTPOSE_PW_MUL(a0, bw_cur[2]);
          RTPOSE_PW_PIN();
        }
        if (ng > 3) {
          RTPOSE_PW_MUL(a1, bw_cur[3]);
          RTPOSE_PW_PIN();
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int fn = 0; fn < NFW; ++fn) bw_cur[g][fn] = bw_nxt[g][fn];
      } else {
      RTPOSE_PW_BLOAD(bnxt, min(gg + 1, gl));
      RTPOSE_PW_ALOAD(a1, 1);
      RTPOSE_PW_PIN();
      RTPOSE_PW_MUL(a0, bcur);
      RTPOSE_PW_PIN();
      if (ng > 1) {
        RTPOSE_PW_BLOAD(bcur, min(gg + 2, gl));
        RTPOSE_PW_ALOAD(a0, 2);
        RTPOSE_PW_PIN();
        RTPOSE_PW_MUL(a1, bnxt);
        RTPOSE_PW_PIN();
      }
      if (ng > 2) {
        RTPOSE_PW_BLOAD(bnxt, min(gg + 3, gl));
        RTPOSE_PW_ALOAD(a1, 3);
        RTPOSE_PW_PIN();
        RTPOSE_PW_MUL(a0, bcur);
        RTPOSE_PW_PIN();
      }
      if (ng > 3) {
        RTPOSE_PW_BLOAD(bcur, min(gg + 4, gl));  // first group of the next chunk
        RTPOSE_PW_PIN();
        RTPOSE_PW_MUL(a1, bnxt);
        RTPOSE_PW_PIN();
      }
      }  // BAHEAD
#undef RTPOSE_PW_MUL
#undef RTPOSE_PW_ALOAD
#undef RTPOSE_PW_BLOAD
    }

    // ---- epilogue: (ReLU), scatter through out_cmap.  A lane holds column ncol of rows
    //      rg*8 + 4*kh + rr of its 32-row fragment (v_mfma_f32_32x32x2_f32 C layout). ----
    int chn[NFW];
#pragma unroll
    for (int fn = 0; fn < NFW; ++fn) {
      const int n = ncol + fn * 32;
      chn[fn] = -1;
      if (n < A.cout) chn[fn] = A.out_cmap ? A.out_cmap[n] : A.out_choff + n;
    }
#pragma unroll
    for (int fm = 0; fm < MF; ++fm) {
      int qrow[16];  // output pixel of every row of this fragment the lane holds (batched LDS reads)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) qrow[rg * 4 + rr] = s_qout[par][wm * (32 * MF) + fm * 32 + rg * 8 + 4 * kh + rr];
#pragma unroll
      for (int fn = 0; fn < NFW; ++fn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[fm][fn][r];
          if (A.relu) v = fmaxf(v, 0.f);
          if (qrow[r] >= 0 && chn[fn] >= 0) A.out[(unsigned)qrow[r] * (unsigned)A.out_cstride + (unsigned)chn[fn]] = v;
        }
      }
    }

    // ---- pass-through half, interleave form: the next unit's x1 = this buffer's logical channels [0, h)
    //      = (even run, odd run) interleaved, written as contiguous runs: 16-byte loads, 8-byte stores,
    //      every line written once (the scatter form below wrote every line of the buffer twice) ------
    if (A.pt.base && A.pt_pairs > 0 && cur.pass == 0) {
      const int g4 = (A.pt_pairs + 3) >> 2;
      const int nit = kPwBM * g4;
      const bool vec2 = !(A.pt_split & 1) && !((A.pt_d1 - A.pt_split) & 1) && !(A.pt_d0 & 1);
      for (int it0 = tid; it0 < nit; it0 += 512) {
        float4 va[2], vb[2];
        int qo[2], k0[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int it = min(it0 + 256 * u, nit - 1);
          const int p = it / g4;
          k0[u] = 4 * (it - p * g4);
          qo[u] = it0 + 256 * u < nit ? s_qout[par][p] : -1;
          const unsigned so = (unsigned)s_qpt[par][p] * (unsigned)A.pt.cstride + (unsigned)(A.pt.choff + k0[u]);
          va[u] = pw_gload4(A.pt.base + (so + (unsigned)A.pt_a));
          vb[u] = pw_gload4(A.pt.base + (so + (unsigned)A.pt_b));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (qo[u] < 0) continue;
          const float a4[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, b4[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
          const unsigned o = (unsigned)qo[u] * (unsigned)A.out_cstride;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = k0[u] + e;  // pair index: outputs j = 2k, 2k + 1
            if (k >= A.pt_pairs) continue;
            const int j = 2 * k;
            if (vec2) {
              const int dj = j < A.pt_split ? A.pt_d0 + j : A.pt_d1 + j - A.pt_split;
              *reinterpret_cast<float2*>(A.out + (o + (unsigned)dj)) = make_float2(a4[e], b4[e]);
            } else {
              const int da = j < A.pt_split ? A.pt_d0 + j : A.pt_d1 + j - A.pt_split;
              const int db = j + 1 < A.pt_split ? A.pt_d0 + j + 1 : A.pt_d1 + j + 1 - A.pt_split;
              A.out[o + (unsigned)da] = a4[e];
              A.out[o + (unsigned)db] = b4[e];
            }
          }
        }
      }
    }

    // ---- pass-through half, scatter form: x1 -> pt_cmap slots (cat + channel_shuffle folded into the store) ------
    if (A.pt.base && A.pt_pairs == 0 && cur.pass == 0) {
      const int g4 = (A.pt_c + 3) >> 2;
      const int nit = kPwBM * g4;
      constexpr int PB = 4;  // loads in flight per thread (a load -> 4 stores chain per item exposed a full
                             // memory round trip per item: 0.25 ms per launch at 232 channels)
      for (int it0 = tid; it0 < nit; it0 += 256 * PB) {
        float4 v[PB];
        int qo[PB], gg[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
          const int it = it0 + 256 * u;
          const int p = min(it, nit - 1) / g4;
          gg[u] = min(it, nit - 1) - p * g4;
          qo[u] = it < nit ? s_qout[par][p] : -1;
          v[u] = pw_gload4(A.pt.base + ((unsigned)s_qpt[par][p] * (unsigned)A.pt.cstride + (unsigned)(A.pt.choff + 4 * gg[u])));
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
          if (qo[u] < 0) continue;
          const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          const unsigned o = (unsigned)qo[u] * (unsigned)A.out_cstride;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (4 * gg[u] + e < A.pt_c) A.out[o + (unsigned)A.pt_cmap[4 * gg[u] + e]] = vv[e];
        }
      }
    }

    if (!has_next) break;
    cur = nxt;
    wi = wnext;
    par ^= 1;
  }
}
#undef RTPOSE_PW_PIN

// packed[c/4][coutp][4] columns [col_off, col_off + cout)  <-  w[cout][cin_src] (1x1), bias likewise:
// several layers may share one packed matrix (the PAF and heat-map heads are one 128-column GEMM)
__global__ void pack_pw_kernel(const float* __restrict__ w, const float* __restrict__ bias, int cout,
                               int cin_src, const int32_t* __restrict__ cin_map, int K, int coutp, int col_off,
                               float* __restrict__ wp, float* __restrict__ bp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cout) bp[col_off + i] = bias ? bias[i] : 0.f;
  if (i >= K * cout) return;
  const int n = i % cout, c = i / cout;
  const int src = cin_map ? cin_map[c] : (c < cin_src ? c : -1);
  const float v = (src >= 0 && src < cin_src) ? w[(size_t)n * cin_src + src] : 0.f;
  wp[((size_t)(c >> 2) * coutp + col_off + n) * 4 + (c & 3)] = v;
}

// packed[c / 4][coutp][4], columns [col_off, col_off + ncols): column col_off + i holds output channel col_map[i] of
// w[cout][cin_src] (col_map NULL: i; < 0: a zero column), its K row c reads input channel cin_map[c] (NULL: c; < 0: zero)
__global__ void pack_pw_cols_kernel(const float* __restrict__ w, const float* __restrict__ bias, int cout, int cin_src,
                                    const int32_t* __restrict__ cin_map, int K, int ncols,
                                    const int32_t* __restrict__ col_map, int coutp, int col_off, float* __restrict__ wp,
                                    float* __restrict__ bp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ncols) {
    const int n = col_map ? col_map[i] : i;
    bp[col_off + i] = (n >= 0 && n < cout && bias) ? bias[n] : 0.f;
  }
  if (i >= K * ncols) return;
  const int ci = i % ncols, c = i / ncols;
  const int n = col_map ? col_map[ci] : ci;
  const int src = cin_map ? cin_map[c] : (c < cin_src ? c : -1);
  const float v = (n >= 0 && n < cout && src >= 0 && src < cin_src) ? w[(size_t)n * cin_src + src] : 0.f;
  wp[((size_t)(c >> 2) * coutp + col_off + ci) * 4 + (c & 3)] = v;
}


int pack_pw_cols_launch(const float* w, const float* bias, int cout, int cin_src, const int32_t* cin_map, int K,
                        int ncols, const int32_t* col_map, int coutp, int col_off, float* wp, float* bp,
                        hipStream_t s) {
  if (!w || !wp || !bp || cout <= 0 || K <= 0 || (K % 8) || ncols <= 0 || col_off < 0 || col_off + ncols > coutp)
    return fail(RTPOSE_E_INVAL, "pack_pw_cols: bad arguments");
  hipLaunchKernelGGL(pack_pw_cols_kernel, dim3(ceil_div(K * ncols, 256)), dim3(256), 0, s, w, bias, cout, cin_src,
                     cin_map, K, ncols, col_map, coutp, col_off, wp, bp);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int pack_pw_launch(const float* w, const float* bias, int cout, int cin_src, const int32_t* cin_map, int K,
                   int coutp, int col_off, float* wp, float* bp, hipStream_t s) {
  if (!w || !wp || !bp || cout <= 0 || K <= 0 || (K % 8) || col_off < 0 || col_off + cout > coutp)
    return fail(RTPOSE_E_INVAL, "pack_pw: bad arguments");
  hipLaunchKernelGGL(pack_pw_kernel, dim3(ceil_div(K * cout, 256)), dim3(256), 0, s, w, bias, cout, cin_src,
                     cin_map, K, coutp, col_off, wp, bp);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

// LDS plane stride of the staged halo of a 64-pixel strip: the strip, the row gaps it crosses, at most
// one image gap, one row above and below (+1 pixel each side); rounded so that the 8 planes sit 8
// banks apart like the A tile's
int pw_halo_stride(const rtpose_layout& l, int H, int W) {
  int np = (kPwBM - 1) + ((kPwBM - 1) / W + 1) * (l.ws - W) + ((kPwBM - 1) / (H * W) + 1) * (l.hs - H) * l.ws +
           2 * l.ws + 3;
  while ((np & 3) != 2) ++np;
  return np;
}

template <int WM, int MF, int NFW, bool DW>
static int pw_launch_inst(const PwArgs& a, int grid, size_t lds, hipStream_t s) {
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  auto kern = pw_gemm_f32<WM, MF, NFW, DW>;
  if (!attr_set.is_set(dev)) {
    RTPOSE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_set.set(dev);
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int pw_fused_launch(const rtpose_pw_desc* d, int N, int H, int W, hipStream_t s) {
  if (!d || !d->in || !d->w_packed || !d->bias_packed || !d->out) return fail(RTPOSE_E_INVAL, "pw_fused: NULL argument");
  if (N <= 0 || H <= 0 || W <= 0) return fail(RTPOSE_E_INVAL, "pw_fused: empty tensor");
  if (d->cin <= 0 || (d->cin % 8)) return fail(RTPOSE_E_INVAL, "pw_fused: cin must be a multiple of 8");
  if (d->coutp != 64 && d->coutp != 128 && (d->coutp % 256)) return fail(RTPOSE_E_INVAL, "pw_fused: coutp must be 64, 128 or a multiple of 256");
  if (d->cout <= 0 || d->cout > d->coutp) return fail(RTPOSE_E_INVAL, "pw_fused: cout exceeds coutp");
  if ((d->lin.cstride % 4) || (d->lin.choff % 4) || (!d->in_planes && d->lin.choff + d->cin > d->lin.cstride))
    return fail(RTPOSE_E_INVAL, "pw_fused: input slice must be 16-byte aligned and inside the pixel");
  const bool dw = d->dw_w != nullptr;
  if (dw && (!d->dw_b || d->lin.ws < W + 1 || d->lin.hs < H + 1 || d->lin.lead < d->lin.ws + 1))
    return fail(RTPOSE_E_INVAL, "pw_fused: the depthwise input needs a layout gap of 1 and a bias");
  if (d->pt_src && d->pt_pairs == 0 && (!d->pt_cmap || d->pt_c <= 0 || (d->lpt.cstride % 4) || (d->lpt.choff % 4)))
    return fail(RTPOSE_E_INVAL, "pw_fused: bad pass-through description");
  if (d->pt_src && d->pt_pairs > 0 &&
      ((d->lpt.cstride % 4) || ((d->lpt.choff + d->pt_a) % 4) || ((d->lpt.choff + d->pt_b) % 4) || d->pt_pairs < 0))
    return fail(RTPOSE_E_INVAL, "pw_fused: interleave pass-through runs must be 16-byte aligned");
  if (d->cin > kPwMaxK) return fail(RTPOSE_E_INVAL, "pw_fused: cin > 1024");
  if (rtpose_layout_pixels(&d->lin, N, H, W) * (size_t)d->lin.cstride >= ((size_t)1 << 31) ||
      rtpose_layout_pixels(&d->lout, N, H, W) * (size_t)d->lout.cstride >= ((size_t)1 << 31))
    return fail(RTPOSE_E_INVAL, "pw_fused: tensors must be below 2^31 floats (32-bit element offsets)");
  PwArgs a;
  memset(&a, 0, sizeof(a));
  a.in = PwView{d->in, d->lin.cstride, d->lin.choff, d->lin.ws, d->lin.hs, d->lin.lead};
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
  a.out_cmap = d->out_cmap;
  if (d->pt_src) {
    a.pt = PwView{d->pt_src, d->lpt.cstride, d->lpt.choff, d->lpt.ws, d->lpt.hs, d->lpt.lead};
    a.pt_cmap = d->pt_cmap;
    a.pt_c = d->pt_c;
    a.pt_pairs = d->pt_pairs;
    a.pt_a = d->pt_a;
    a.pt_b = d->pt_b;
    a.pt_split = d->pt_split;
    a.pt_d0 = d->pt_d0;
    a.pt_d1 = d->pt_d1;
  }
  a.in_planes = d->in_planes;
  a.N = N;
  a.H = H;
  a.W = W;
  a.M = N * H * W;
  a.K = d->cin;
  a.coutp = d->coutp;
  a.cout = d->cout;
  a.relu = d->relu;
  size_t lds = (size_t)2 * kPwPL * kPwQS * 16;
  if (dw) {
    a.nps = 32 * kPwMaxStage + 2;  // every staged piece has a slot (unconditional parking); planes 8 banks apart
    lds += (size_t)kPwPL * a.nps * 16 + (size_t)10 * d->cin * 4;
  }
  const int npass_h = d->coutp == 64 ? 1 : (d->coutp == 128 ? 1 : d->coutp / 256);
  a.fHW = make_fastdiv(H * W);
  a.fW = make_fastdiv(W);
  a.fnp = make_fastdiv(npass_h);
  a.ftx = make_fastdiv(ceil_div(W, kPwTile));
  a.fty = make_fastdiv(ceil_div(H, kPwTile));
  const int nwork = (dw ? N * ceil_div(H, kPwTile) * ceil_div(W, kPwTile) : ceil_div(a.M, kPwBM)) * npass_h;
  const int grid = nwork < 2 * device_cu_count() ? nwork : 2 * device_cu_count();  // persistent: 2 blocks per CU
  if (d->coutp == 64) return dw ? pw_launch_inst<2, 1, 1, true>(a, grid, lds, s) : pw_launch_inst<2, 1, 1, false>(a, grid, lds, s);
  if (d->coutp == 128) return dw ? pw_launch_inst<1, 2, 1, true>(a, grid, lds, s) : pw_launch_inst<1, 2, 1, false>(a, grid, lds, s);
  return dw ? pw_launch_inst<1, 2, 2, true>(a, grid, lds, s) : pw_launch_inst<1, 2, 2, false>(a, grid, lds, s);
}

}  // namespace rtpose

using namespace rtpose;

extern "C" {

size_t rtpose_packed_pw_floats(int cin_packed, int coutp) { return (size_t)cin_packed * coutp; }

int rtpose_pack_pw_weights(const float* w_oi, const float* bias, int cout, int cin_src, const int32_t* cin_map,
                           int cin_packed, int coutp, int col_off, float* w_packed, float* bias_packed,
                           void* stream) {
  return pack_pw_launch(w_oi, bias, cout, cin_src, cin_map, cin_packed, coutp, col_off, w_packed, bias_packed,
                        as_stream(stream));
}

int rtpose_pack_pw_weights_cols(const float* w_oi, const float* bias, int cout, int cin_src, const int32_t* cin_map,
                                int cin_packed, int ncols, const int32_t* col_map, int coutp, int col_off,
                                float* w_packed, float* bias_packed, void* stream) {
  return pack_pw_cols_launch(w_oi, bias, cout, cin_src, cin_map, cin_packed, ncols, col_map, coutp, col_off, w_packed,
                             bias_packed, as_stream(stream));
}

int rtpose_pw_fused(const rtpose_pw_desc* d, int N, int H, int W, void* stream) {
  return pw_fused_launch(d, N, H, W, as_stream(stream));
}

}  // extern "C"
