// The first layer of the VGG-19 front end for gfx950 (MI355X): conv1_1 = nn.Conv2d(3, 64, 3, 1, 1) + nn.ReLU
// (lib/network/rtpose_vgg.py:23-35, `model0.0` of the state_dict), fp32, reading the image where the caller left
// it: dense NCHW (rtpose_net_forward; the reference hands its module an NCHW tensor, rtpose_vgg.py:158) or the
// NHWC8 input buffer the image-prep kernel filled (rtpose_net_forward_prepared).
//
// Why its own kernel: K = 27 (3 channels x 9 taps) is no contraction to speak of - 0.23 GMAC per image against the
// 1.1 GB the layer WRITES for a 32-image batch.  The generic implicit-GEMM kernel (conv_mfma.hip) ran it on 8 padded
// channels at 0.50 ms after a separate 0.10 ms NCHW -> NHWC8 pass; its bound is the HBM write, 1.11 GB / ~4.5 TB/s =
// 0.25 ms.  Here one block owns an 8 x 32 pixel tile of one image: the 3 x 10 x 34 halo is fetched once from the
// image (masked loads give the zero padding), the 27 taps are the K dimension of v_mfma_f32_32x32x2_f32 directly
// (K = 28, one zero row), the 28 x 64 filter matrix lives in registers (28 values per lane), the A operand of a step
// is ONE ds_read_b32 per lane at a compile-time offset, and a wave's 2 rows x 64 channels leave as 128-byte runs per
// pixel (lane = channel).  No im2col buffer, no layout conversion pass, no 8-channel padding of the input.
// Round 4, PLANES = true: the output as 8 channel planes ([channels / 8][pixel slots][8], rtpose_conv_desc.in_plane_pixels:
// what the F(4x4,3x3) kernel of conv1_2 reads in whole lines).  Computed as the TRANSPOSED product - filters as the row
// operand, pixels as the column operand, the same registers - so that a lane holds ONE pixel and 16 channels of a 32-channel
// half: register r of lane half kh is channel 8 (r / 4) + 4 kh + r % 4, i.e. floats 4 kh .. 4 kh + 3 of plane r / 4 - a 16-byte
// store per plane, and the two lane halves of the 32 pixels of a row segment fill ONE contiguous 1 KB run of that plane.
// (First version: channels permuted so that a lane owned whole 32-byte plane entries - every store instruction then wrote
// every other 16 bytes of two runs: conv1_1 0.30 -> 0.37 ms.)
// Round 6, MODE 2 - the bf16 plan's conv1_1 (BASELINE config 3 arithmetic): image and filters rounded to bf16 (RNE, as the plan's
// input conversion and its weight packing do), products exact, fp32 accumulation, + bias, ReLU, output rounded to bf16.  It
// replaces two launches of the bf16 plan - the NCHW -> NHWC16 bf16 conversion (0.08 ms) and the generic bf16 kernel on 16 padded
// channels (0.25 ms; K = 144 for 27 taps).  Same tile, same halo (rounded on the way in); the 27 taps are K = 32 of TWO
// v_mfma_f32_32x32x16_bf16 steps per (row, channel half) - 8 x 32 cycles of the matrix pipe per wave instead of the 56 x 64 the
// first version spent on v_mfma_f32_32x32x2_f32 (correct - a product of two bf16 values is exact in fp32 - but 0.27 ms, 0.10 of
// it matrix time): the B operand of a step is 8 taps of the lane's pixel gathered from the LDS halo (8 ds_read_b32 at
// compile-time offsets, packed with 4 v_perm_b32), the A operand 8 filter taps of the lane's channel (4 x 16 bytes per lane,
// loaded once).  Transposed product as in PLANES (a lane = one pixel, its registers = channels 8 (r / 4) + 4 kh + r % 4 of a
// 32-channel half); the two lane halves trade their 4-channel halves of a group pair through v_permlane32_swap so that every
// lane stores 8 consecutive channels of its pixel: one 16-byte store.
#include <hip/hip_runtime.h>

#include "common.h"

namespace rtpose {

namespace first {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int TH = 8, TW = 32;        // pixel tile of a block (4 waves x 2 rows)
constexpr int HH = TH + 2, WW = TW + 2;
constexpr int KS = 14;                // K = 28 in steps of 2

struct Args {
  const float* x_nchw;    // dense [N, 3, H, W], or NULL: read `x_lay` below
  const float* x_lay;     // shared-gap NHWC buffer with >= 3 channels per pixel (the plan's NHWC8 input)
  int xl_cstride, xl_choff, xl_ws, xl_hs, xl_lead;
  const float* wp;        // packed filters [2 k halves][32 lanes][2 column halves][14 steps] (pack_first_kernel)
  const float* bias;      // 64 floats
  float* out;
  int o_cstride, o_choff, o_ws, o_hs, o_lead;
  int o_pq;               // PLANES: pixel slots per plane
  int N, H, W, relu, tiles_x, tiles_y;
};

__device__ __forceinline__ float round_bf16(float v) {  // RNE to bf16, as a float (v_cvt_pk_bf16_f32)
  return __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v) << 16);
}
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {  // (bf16(a), bf16(b)), a in the low half
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float fl2 __attribute__((ext_vector_type(2)));
  const fl2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}

// k = c * 9 + dy * 3 + dx (k = 27: the zero row) -> offset of the tap in the LDS halo [3][HH][WW]
__host__ __device__ constexpr int tap_off(int k) {
  return k >= 27 ? 0 : ((k / 9) * HH + (k % 9) / 3) * WW + (k % 3);
}

// MODE 0: fp32, pixel-major output; 1: fp32, channel planes (PLANES); 2: bf16 arithmetic, bf16 pixel-major output
template <int MODE>
__global__ __launch_bounds__(256, 3) void conv_first_kernel(const Args A) {
  constexpr bool PLANES = MODE != 0;  // (the transposed product: a lane holds one pixel)
  constexpr bool BF16 = MODE == 2;
  // PERSISTENT (round 6): the grid is 3 blocks per CU and a block walks the tiles blockIdx.x, + gridDim.x, ...  As one tile per
  // block every one of the 17,664 blocks (4.3 us each) loaded its filters and bias (7 + 8 16-byte loads per lane) and waited
  // for its own halo before its first MFMA; the matrix pipe was 29 % busy and the stores ran at 3.3 TB/s where a fill of the
  // same buffer runs at 6.8 (tools/exp/write_bw.py).  Now the filters and the bias stay in registers, and the next tile's halo
  // is requested before this tile's multiply and parked in the other LDS buffer behind it: one barrier per tile, no exposed
  // load in front of the multiply after the first tile (fp32: 0.336 -> 0.275 ms).
  __shared__ float halo2[2][3 * HH * WW];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int ntiles = A.N * A.tiles_y * A.tiles_x;
  constexpr int NH = (3 * HH * WW + 255) / 256;  // halo elements per thread
  // ---- halo: rows y0 - 1 .. y0 + TH, columns x0 - 1 .. x0 + TW of the 3 channels; outside the image = 0 ----
  auto fetch_halo = [&](int b, float (&hv)[NH]) {
    const int tx = b % A.tiles_x;
    b /= A.tiles_x;
    const int ty = b % A.tiles_y, n = b / A.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    int tv = tid;  // (opaque: as loop invariants the halo coordinates of the NH elements would be held across the multiply)
    asm volatile("" : "+v"(tv));
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int i = min(tv + 256 * j, 3 * HH * WW - 1);
      const int c = i / (HH * WW), r = i - c * (HH * WW);
      const int yy = y0 - 1 + r / WW, xx = x0 - 1 + r % WW;
      float v = 0.f;
      if (yy >= 0 && yy < A.H && xx >= 0 && xx < A.W) {
        v = A.x_nchw ? A.x_nchw[((size_t)(n * 3 + c) * A.H + yy) * A.W + xx]
                     : A.x_lay[((size_t)A.xl_lead + (size_t)(n * A.xl_hs + yy) * A.xl_ws + xx) * A.xl_cstride + A.xl_choff + c];
      }
      hv[j] = BF16 ? round_bf16(v) : v;
    }
  };
  auto park_halo = [&](float* dst, const float (&hv)[NH]) {
#pragma unroll
    for (int j = 0; j < NH; ++j)
      if (tid + 256 * j < 3 * HH * WW) dst[tid + 256 * j] = hv[j];
  };
  int bcur = blockIdx.x;
  if (bcur >= ntiles) return;
  {
    float hv[NH];
    fetch_halo(bcur, hv);
    park_halo(halo2[0], hv);
  }
  // ---- filters: 28 values per lane (k = 2 step + kh, column = half * 32 + l31), bias in the accumulators ----
  // (MODE 2: 4 x 8 bf16 per lane - channel half * 32 + l31, taps 16 s + 8 kh .. + 7)
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
  float wv[2][BF16 ? 1 : KS];
  uintx4 wq[2][2];
  if (BF16) {
    const uintx4* w16 = reinterpret_cast<const uintx4*>(A.wp) + (size_t)(kh * 32 + l31) * 4;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int st = 0; st < 2; ++st) wq[nh][st] = w16[nh * 2 + st];
  } else {
    const float4* w4 = reinterpret_cast<const float4*>(A.wp + (size_t)(kh * 32 + l31) * (2 * KS));
#pragma unroll
    for (int q = 0; q < 2 * KS / 4; ++q) {
      const float4 t = w4[q];
      const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) wv[(4 * q + e) / KS][(4 * q + e) % KS] = tv[e];
    }
  }
  float bias_r[2][PLANES ? 16 : 1];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    if (PLANES) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bias_r[nh][r] = A.bias[nh * 32 + 8 * (r >> 2) + 4 * kh + (r & 3)];
    } else {
      bias_r[nh][0] = A.bias[nh * 32 + l31];
    }
  }
  __syncthreads();

  for (int buf = 0;; buf ^= 1) {
  const int bnext = bcur + (int)gridDim.x;
  const bool more = bnext < ntiles;
  float hv[NH];
  if (more) fetch_halo(bnext, hv);
  int b = bcur;
  const int tx = b % A.tiles_x;
  b /= A.tiles_x;
  const int ty = b % A.tiles_y, n = b / A.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const float* const halo = halo2[buf];
  floatx16 acc[2][2];  // [row of the wave][column half]
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nh][r] = bias_r[nh][PLANES ? r : 0];

  // ---- 14 K steps: A = halo[tap(2 step + kh)] at (row 2 wave + mt, pixel l31) ----------------------------
  const float* hb = halo + (2 * wave) * WW + l31;
  if (BF16) {
    // ---- 2 K steps of 16 taps: B = the lane's pixel at taps 16 st + 8 kh .. + 7 (taps >= 27 meet zero filters) ----
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        unsigned pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int o0 = kh ? tap_off(16 * st + 8 + 2 * e) : tap_off(16 * st + 2 * e);
          const int o1 = kh ? tap_off(16 * st + 8 + 2 * e + 1) : tap_off(16 * st + 2 * e + 1);
          const unsigned lo = __float_as_uint(hb[o0 + mt * WW]), hi = __float_as_uint(hb[o1 + mt * WW]);
          pk[e] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);  // (lo >> 16) | (hi & 0xffff0000): the values are bf16 already
        }
        const uintx4 bq = {pk[0], pk[1], pk[2], pk[3]};
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
          acc[mt][nh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wq[nh][st]),
                                                                __builtin_bit_cast(bf16x8, bq), acc[mt][nh], 0, 0, 0);
      }
  }
#pragma unroll
  for (int st = 0; st < (BF16 ? 0 : KS); ++st) {
    const int off = kh ? tap_off(2 * st + 1) : tap_off(2 * st);
    const float a0 = hb[off], a1 = hb[off + WW];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      acc[0][nh] = PLANES ? __builtin_amdgcn_mfma_f32_32x32x2f32(wv[nh][st], a0, acc[0][nh], 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wv[nh][st], acc[0][nh], 0, 0, 0);
      acc[1][nh] = PLANES ? __builtin_amdgcn_mfma_f32_32x32x2f32(wv[nh][st], a1, acc[1][nh], 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wv[nh][st], acc[1][nh], 0, 0, 0);
    }
  }

  if (BF16) {
    // ---- store: lane = pixel x0 + l31 of the row; registers 4 j .. 4 j + 3 = channels 8 j + 4 kh .. + 3 of the half.  For the
    // group pair (2 m, 2 m + 1) the lane halves swap: kh = 0 ends up with all 8 channels of group 2 m, kh = 1 of 2 m + 1 ----
    unsigned short* const out16 = reinterpret_cast<unsigned short*>(A.out);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int y = y0 + 2 * wave + mt;
      const bool ok = y < A.H && x0 + l31 < A.W;
      const size_t q = (size_t)A.o_lead + (size_t)(n * A.o_hs + min(y, A.H - 1)) * A.o_ws + min(x0 + l31, A.W - 1);
      unsigned short* op = out16 + q * A.o_cstride + A.o_choff;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          unsigned a[2], b[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float v0 = acc[mt][nh][8 * m + 2 * e], v1 = acc[mt][nh][8 * m + 2 * e + 1];
            float w0 = acc[mt][nh][8 * m + 4 + 2 * e], w1 = acc[mt][nh][8 * m + 4 + 2 * e + 1];
            if (A.relu) {
              v0 = fmaxf(v0, 0.f);
              v1 = fmaxf(v1, 0.f);
              w0 = fmaxf(w0, 0.f);
              w1 = fmaxf(w1, 0.f);
            }
            a[e] = pack_bf16x2(v0, v1);  // group 2 m, channels 4 kh + 2 e, + 1
            b[e] = pack_bf16x2(w0, w1);  // group 2 m + 1
            // lanes 32..63 of `a` <-> lanes 0..31 of `b`
            const auto sw = __builtin_amdgcn_permlane32_swap(a[e], b[e], false, false);
            a[e] = sw[0];
            b[e] = sw[1];
          }
          // kh = 0: a = own channels 0..3 of group 2 m, b = the partner's 4..7 of it; kh = 1: a = the partner's 0..3 of group
          // 2 m + 1, b = own 4..7 of it
          if (ok) *reinterpret_cast<uint4*>(op + nh * 32 + 8 * (2 * m + kh)) = make_uint4(a[0], a[1], b[0], b[1]);
        }
    }
  } else if (PLANES) {
    // ---- store: lane = pixel x0 + l31 of the row, registers 4 j .. 4 j + 3 = floats 4 kh .. 4 kh + 3 of plane 4 half + j ----
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int y = y0 + 2 * wave + mt;
      if (y >= A.H || x0 + l31 >= A.W) continue;
      const size_t q = (size_t)A.o_lead + (size_t)(n * A.o_hs + y) * A.o_ws + x0 + l31;
      float* op = A.out + ((size_t)(A.o_choff >> 3) * A.o_pq + q) * 8 + 4 * kh;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v = make_float4(acc[mt][nh][4 * j], acc[mt][nh][4 * j + 1], acc[mt][nh][4 * j + 2], acc[mt][nh][4 * j + 3]);
          if (A.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
          *reinterpret_cast<float4*>(op + (size_t)(4 * nh + j) * A.o_pq * 8) = v;
        }
    }
  } else {
  // ---- store: register r of a lane = pixel x0 + (r / 4) * 8 + 4 kh + r % 4 of the row, channel half * 32 + l31 ----
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int y = y0 + 2 * wave + mt;
    if (y >= A.H) continue;
    float* orow = A.out + ((size_t)A.o_lead + (size_t)(n * A.o_hs + y) * A.o_ws + x0) * A.o_cstride + A.o_choff + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = (r >> 2) * 8 + 4 * kh + (r & 3);
      if (x0 + px < A.W) {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          const float v = acc[mt][nh][r];
          orow[(size_t)px * A.o_cstride + nh * 32] = A.relu ? fmaxf(v, 0.f) : v;
        }
      }
    }
  }
  }
  if (!more) break;
  park_halo(halo2[buf ^ 1], hv);
  __syncthreads();
  bcur = bnext;
  }
}

// w[64][3][3][3] (OIHW) -> wp[kh][l31][half][step] = w[half * 32 + l31][k = 2 step + kh] (k = 27: 0); bias copied
__global__ void pack_first_kernel(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ wp,
                                  float* __restrict__ bp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 64) bp[i] = bias ? bias[i] : 0.f;
  if (i >= 2 * 32 * 2 * KS) return;
  const int st = i % KS, nh = (i / KS) % 2, l31 = (i / (2 * KS)) % 32, kh = i / (2 * KS * 32);
  const int k = 2 * st + kh, o = nh * 32 + l31;
  wp[i] = k < 27 ? w[o * 27 + k] : 0.f;
}

// MODE 2: wp16[kh][l31][half][step][8] = bf16(w[half * 32 + l31][k = 16 step + 8 kh + e]) (k >= 27: 0), 16-byte pieces; the
// fp32 bias behind the same 2 * 32 * 2 * KS floats as in the fp32 packing
__global__ void pack_first_bf16_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                       unsigned short* __restrict__ wp16, float* __restrict__ bp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 64) bp[i] = bias ? bias[i] : 0.f;
  if (i >= 2 * 32 * 2 * 2 * 8) return;
  const int e = i & 7, st = (i >> 3) & 1, nh = (i >> 4) & 1, l31 = (i >> 5) & 31, kh = i >> 10;
  const int k = 16 * st + 8 * kh + e, o = nh * 32 + l31;
  wp16[i] = __builtin_bit_cast(unsigned short, (__bf16)(k < 27 ? w[o * 27 + k] : 0.f));
}

}  // namespace first

size_t conv_first_packed_floats() { return (size_t)2 * 32 * 2 * first::KS + 64; }

int conv_first_pack_launch(const float* w_oihw, const float* bias, float* wp, hipStream_t s, int to_bf16) {
  if (to_bf16) {
    hipLaunchKernelGGL(first::pack_first_bf16_kernel, dim3(ceil_div(2 * 32 * 2 * 2 * 8, 256)), dim3(256), 0, s, w_oihw, bias,
                       reinterpret_cast<unsigned short*>(wp), wp + 2 * 32 * 2 * first::KS);
    RTPOSE_HIP_CHECK(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(first::pack_first_kernel, dim3(ceil_div(2 * 32 * 2 * first::KS, 256)), dim3(256), 0, s, w_oihw, bias,
                     wp, wp + 2 * 32 * 2 * first::KS);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

// x_nchw != NULL: dense NCHW source; else the layout `lx` on `x_lay` (>= 3 channels per pixel)
// out_bf16: MODE 2 - `out` holds bf16 elements, `lo` counts them, `wp` was packed with to_bf16 = 1
int conv_first_launch(const float* x_nchw, const float* x_lay, const rtpose_layout* lx, const float* wp, float* out,
                      const rtpose_layout* lo, int out_plane_pixels, int relu, int N, int H, int W, hipStream_t s,
                      int out_bf16) {
  using namespace first;
  if (out_bf16 && (out_plane_pixels || (lo && ((lo->cstride % 8) || (lo->choff % 8)))))
    return fail(RTPOSE_E_INVAL, "conv_first: bf16 output is pixel-major with 16-byte aligned slices");
  if ((!x_nchw && (!x_lay || !lx)) || !wp || !out || !lo || N <= 0 || H <= 0 || W <= 0 || out_plane_pixels < 0)
    return fail(RTPOSE_E_INVAL, "conv_first: bad arguments");
  if (out_plane_pixels) {
    if ((lo->choff % 8) || (size_t)out_plane_pixels < rtpose_layout_pixels(lo, N, H, W))
      return fail(RTPOSE_E_INVAL, "conv_first: channel planes start at a multiple of 8 channels and hold the layout's pixels");
  } else if (lo->choff + 64 > lo->cstride) {
    return fail(RTPOSE_E_INVAL, "conv_first: output slice exceeds cstride");
  }
  Args a;
  memset(&a, 0, sizeof(a));
  a.x_nchw = x_nchw;
  a.x_lay = x_lay;
  if (lx) {
    a.xl_cstride = lx->cstride;
    a.xl_choff = lx->choff;
    a.xl_ws = lx->ws;
    a.xl_hs = lx->hs;
    a.xl_lead = lx->lead;
  }
  a.wp = wp;
  a.bias = wp + 2 * 32 * 2 * KS;
  a.out = out;
  a.o_cstride = lo->cstride;
  a.o_choff = lo->choff;
  a.o_ws = lo->ws;
  a.o_hs = lo->hs;
  a.o_lead = lo->lead;
  a.o_pq = out_plane_pixels;
  a.N = N;
  a.H = H;
  a.W = W;
  a.relu = relu;
  a.tiles_x = ceil_div(W, TW);
  a.tiles_y = ceil_div(H, TH);
  const long tiles = (long)N * a.tiles_x * a.tiles_y;
  if (tiles > 0x3fffffffL) return fail(RTPOSE_E_INVAL, "conv_first: too many tiles");
  // fp32: persistent, 3 blocks per CU (<= 168 registers, 8 KB of LDS each) walk the tiles.  bf16 output (MODE 2): one tile per
  // block - the persistent form bought it nothing (0.213 -> 0.210 ms) and, filling every SIMD's registers for the whole launch,
  // kept the decoder of the batch before off the CUs while it ran: host-to-host streaming of the bf16 plan 1.01 -> 0.79 of the
  // device-resident rate (profiles/r06_conv_first_persistent.txt)
  const long blocks = (out_bf16 || tiles < 3L * device_cu_count()) ? tiles : 3L * device_cu_count();
  if (out_bf16) hipLaunchKernelGGL(conv_first_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else if (out_plane_pixels) hipLaunchKernelGGL(conv_first_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(conv_first_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace rtpose

extern "C" {

size_t rtpose_conv_first_packed_floats(void) { return rtpose::conv_first_packed_floats(); }

int rtpose_pack_conv_first(const float* w_oihw, const float* bias, float* w_packed, void* stream) {
  if (!w_oihw || !w_packed) return rtpose::fail(RTPOSE_E_INVAL, "pack_conv_first: NULL argument");
  return rtpose::conv_first_pack_launch(w_oihw, bias, w_packed, rtpose::as_stream(stream), 0);
}

int rtpose_conv_first(const float* x_nchw, const float* x_layout, const rtpose_layout* lx, const float* w_packed,
                      float* out, const rtpose_layout* lout, int relu, int N, int H, int W, void* stream) {
  return rtpose::conv_first_launch(x_nchw, x_layout, lx, w_packed, out, lout, 0, relu, N, H, W, rtpose::as_stream(stream), 0);
}

int rtpose_pack_conv_first_bf16(const float* w_oihw, const float* bias, float* w_packed, void* stream) {
  if (!w_oihw || !w_packed) return rtpose::fail(RTPOSE_E_INVAL, "pack_conv_first_bf16: NULL argument");
  return rtpose::conv_first_pack_launch(w_oihw, bias, w_packed, rtpose::as_stream(stream), 1);
}

int rtpose_conv_first_bf16(const float* x_nchw, const float* x_layout, const rtpose_layout* lx, const float* w_packed,
                           void* out_bf16, const rtpose_layout* lout, int relu, int N, int H, int W, void* stream) {
  if (lout && lout->choff + 64 > lout->cstride) return rtpose::fail(RTPOSE_E_INVAL, "conv_first_bf16: output slice exceeds cstride");
  return rtpose::conv_first_launch(x_nchw, x_layout, lx, w_packed, static_cast<float*>(out_bf16), lout, 0, relu, N, H, W,
                                   rtpose::as_stream(stream), 1);
}

int rtpose_conv_first_planes(const float* x_nchw, const float* x_layout, const rtpose_layout* lx, const float* w_packed,
                             float* out, const rtpose_layout* lout, int out_plane_pixels, int relu, int N, int H, int W,
                             void* stream) {
  if (out_plane_pixels <= 0) return rtpose::fail(RTPOSE_E_INVAL, "conv_first_planes: out_plane_pixels must be positive");
  return rtpose::conv_first_launch(x_nchw, x_layout, lx, w_packed, out, lout, out_plane_pixels, relu, N, H, W,
                                   rtpose::as_stream(stream), 0);
}

}  // extern "C"
