// HBM-bound data-movement kernels around the conv path: NCHW <-> shared-gap
// padded NHWC, slice copies, 2x2 max-pool (nn.MaxPool2d(2,2,0),
// lib/network/rtpose_vgg.py:49-50) and the flip-TTA merge
// (evaluate/coco_eval.py:197-242).  All are one coalesced pass, float4 on the
// NHWC side where the slice is 16-byte aligned.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "common.h"

namespace rtpose {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

struct Lay {
  int cstride, choff, ws, hs, lead;
};
static Lay to_lay(const rtpose_layout* l) { return Lay{l->cstride, l->choff, l->ws, l->hs, l->lead}; }

__device__ __forceinline__ size_t lay_off(const Lay& l, int n, int y, int x) {
  return ((size_t)l.lead + (size_t)(n * l.hs + y) * l.ws + x) * l.cstride + l.choff;
}

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float v) {
  return __builtin_bit_cast(unsigned short, (__bf16)v);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
  return __uint_as_float((unsigned)h << 16);
}

// One thread per (pixel, channel) with channel fastest on the NHWC side; the
// NCHW side is strided by H*W, served from L2 (tensors here are small or read
// once).  Used for the 3-channel input image and the 38/19-channel outputs.
__global__ void nchw_to_layout_kernel(const float* __restrict__ src, float* __restrict__ dst, Lay l,
                                      int C, int cpad, int N, int H, int W) {
  const size_t total = (size_t)N * H * W * cpad;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = i % cpad;
  size_t p = i / cpad;
  const int x = p % W;
  p /= W;
  const int y = p % H;
  const int n = p / H;
  const float v = (c < C) ? src[(((size_t)n * C + c) * H + y) * W + x] : 0.f;
  dst[lay_off(l, n, y, x) + c] = v;
}

// x fastest on the NCHW side (coalesced writes); the NHWC reads of one wave hit
// 64 different pixels of the same channel: 64 sectors, but C (<=57) consecutive
// launches-worth of threads re-use them from L2.
__global__ void layout_to_nchw_kernel(const float* __restrict__ src, Lay l, float* __restrict__ dst,
                                      int C, int N, int H, int W) {
  const size_t total = (size_t)N * C * H * W;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = i % W;
  size_t p = i / W;
  const int y = p % H;
  p /= H;
  const int c = p % C;
  const int n = p / C;
  dst[i] = src[lay_off(l, n, y, x) + c];
}

__global__ void layout_copy_kernel(const float* __restrict__ src, Lay ls, float* __restrict__ dst,
                                   Lay ld, int C, int N, int H, int W) {
  const int c4 = C >> 2;
  const size_t total = (size_t)N * H * W * c4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (i % c4) * 4;
  size_t p = i / c4;
  const int x = p % W;
  p /= W;
  const int y = p % H;
  const int n = p / H;
  *reinterpret_cast<float4*>(dst + lay_off(ld, n, y, x) + c) =
      *reinterpret_cast<const float4*>(src + lay_off(ls, n, y, x) + c);
}

__global__ void layout_copy_scalar_kernel(const float* __restrict__ src, Lay ls,
                                          float* __restrict__ dst, Lay ld, int C, int N, int H,
                                          int W) {
  const size_t total = (size_t)N * H * W * C;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = i % C;
  size_t p = i / C;
  const int x = p % W;
  p /= W;
  const int y = p % H;
  const int n = p / H;
  dst[lay_off(ld, n, y, x) + c] = src[lay_off(ls, n, y, x) + c];
}

__global__ void maxpool2x2_kernel(const float* __restrict__ src, Lay ls, float* __restrict__ dst,
                                  Lay ld, int C, int N, int Ho, int Wo) {
  const int c4 = C >> 2;
  const size_t total = (size_t)N * Ho * Wo * c4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (i % c4) * 4;
  size_t p = i / c4;
  const int x = p % Wo;
  p /= Wo;
  const int y = p % Ho;
  const int n = p / Ho;
  const float4 a = *reinterpret_cast<const float4*>(src + lay_off(ls, n, 2 * y, 2 * x) + c);
  const float4 b = *reinterpret_cast<const float4*>(src + lay_off(ls, n, 2 * y, 2 * x + 1) + c);
  const float4 d = *reinterpret_cast<const float4*>(src + lay_off(ls, n, 2 * y + 1, 2 * x) + c);
  const float4 e = *reinterpret_cast<const float4*>(src + lay_off(ls, n, 2 * y + 1, 2 * x + 1) + c);
  float4 r;
  r.x = fmaxf(fmaxf(a.x, b.x), fmaxf(d.x, e.x));
  r.y = fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y));
  r.z = fmaxf(fmaxf(a.z, b.z), fmaxf(d.z, e.z));
  r.w = fmaxf(fmaxf(a.w, b.w), fmaxf(d.w, e.w));
  *reinterpret_cast<float4*>(dst + lay_off(ld, n, y, x) + c) = r;
}

// handle_paf_and_heat (evaluate/coco_eval.py:197-242): average a map with the
// x-mirrored, left/right-channel-swapped map of the flipped image; the PAF
// x components (even channels AFTER the swap gather, :237) change sign.
__constant__ int kSwapHeat[19] = {0, 1, 5, 6, 7, 2, 3, 4, 11, 12, 13, 8, 9, 10, 15, 14, 17, 16, 18};
__constant__ int kSwapPaf[38] = {6,  7,  8,  9,  10, 11, 0,  1,  2,  3,  4,  5,  20,
                                 21, 22, 23, 24, 25, 26, 27, 12, 13, 14, 15, 16, 17,
                                 18, 19, 28, 29, 32, 33, 30, 31, 36, 37, 34, 35};

__global__ void flip_merge_kernel(const float* __restrict__ heat, const float* __restrict__ heat_f,
                                  const float* __restrict__ paf, const float* __restrict__ paf_f,
                                  int N, int h, int w, float* __restrict__ heat_avg,
                                  float* __restrict__ paf_avg) {
  const size_t npix = (size_t)N * h * w;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * 57) return;
  const int c = i % 57;
  const size_t p = i / 57;
  const int x = p % w;
  const size_t row = p / w;  // n*h + y
  const size_t pf = row * w + (w - 1 - x);
  if (c < 19) {
    heat_avg[p * 19 + c] = (heat[p * 19 + c] + heat_f[pf * 19 + kSwapHeat[c]]) / 2.f;
  } else {
    const int k = c - 19;
    // coco_eval.py:237 negates the channels listed in swap_paf[::2] in place
    // (:236 is a no-op), then :238 gathers with swap_paf: output channel k
    // reads flipped channel swap_paf[k], negated iff swap_paf[k] is one of the
    // swap_paf[::2] entries, i.e. iff it is an even channel index.
    const int sc = kSwapPaf[k];
    float v = paf_f[pf * 38 + sc];
    if ((sc & 1) == 0) v = -v;
    paf_avg[p * 38 + k] = (paf[p * 38 + k] + v) / 2.f;
  }
}

// Fused caller-side image prep: crop_with_factor + rtpose/vgg_preprocess
// (lib/network/im_transform.py:119-134, lib/datasets/preprocessing.py:16-43) on the device:
// uint8 BGR HWC image -> cv2.resize(fx=fy=scale, INTER_LINEAR) in OpenCV's 11-bit fixed point
// -> zero pad to the network size -> normalise -> the net's NHWC8 input buffer.  Integer
// arithmetic identical to preprocess.resize_linear_u8 (the numpy restatement): bit-exact.
// Columns (resize.cpp's dx loop): taps that fall outside clamp BOTH the offset and the weight
// (f = 0).  Rows (the dy loop) keep their weight and resizeGeneric_Invoker clips the two source
// rows instead - same value in real arithmetic, not always the same 11-bit fixed-point sum.
template <bool ROWS>
__device__ __forceinline__ void lin_coeff(int d, int sn, double scale, int& s0, int& s1, int& a0, int& a1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (ROWS) {
    s0 = min(max(s, 0), sn - 1);
    s1 = min(max(s + 1, 0), sn - 1);
  } else {
    if (s < 0) {
      f = 0.f;
      s = 0;
    }
    if (s >= sn - 1) {
      f = 0.f;
      s = sn - 1;
    }
    s0 = s;
    s1 = min(s + 1, sn - 1);
  }
  a1 = (int)rintf(f * 2048.f);
  a0 = (int)rintf((1.f - f) * 2048.f);
}

// One image of a batch: where it is, how it is resized, which image slot of the network input it fills.
struct PrepImg {
  const unsigned char* img;  // device, BGR uint8 [h0][w0][3]
  double inv_scale;          // 1 / im_scale
  int h0, w0, hr, wr;        // source size; resized (valid) size inside the Hn x Wn padded input
  int flip, n;               // mirror inside the valid width; image index in the destination
};
constexpr int kPrepBatch = 64;  // descriptors per launch, passed by value (kernel argument: no device table)
struct PrepBatch {
  PrepImg im[kPrepBatch];
};

// grid (pixels of the padded Hn x Wn input / 256, images): one launch prepares a whole bucket of images
__global__ void preprocess_u8_kernel(PrepBatch batch, int mode, float* __restrict__ dst, Lay ld, int Hn, int Wn) {
  const PrepImg& d = batch.im[blockIdx.y];
  const unsigned char* __restrict__ img = d.img;
  const int h0 = d.h0, w0 = d.w0, hr = d.hr, wr = d.wr;
  const size_t total = (size_t)Hn * Wn;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = i % Wn, y = i / Wn;
  int px[3] = {0, 0, 0};  // padded area: pixel value 0 (im_transform.py:130-131)
  if (y < hr && x < wr) {
    int x0, x1, ax0, ax1, y0, y1, ay0, ay1;
    // flip: this destination column shows the x-mirrored RESIZED image (valid region only)
    lin_coeff<false>(d.flip ? wr - 1 - x : x, w0, d.inv_scale, x0, x1, ax0, ax1);
    lin_coeff<true>(y, h0, d.inv_scale, y0, y1, ay0, ay1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int s00 = img[((size_t)y0 * w0 + x0) * 3 + c], s01 = img[((size_t)y0 * w0 + x1) * 3 + c];
      const int s10 = img[((size_t)y1 * w0 + x0) * 3 + c], s11 = img[((size_t)y1 * w0 + x1) * 3 + c];
      const int r0 = s00 * ax0 + s01 * ax1, r1 = s10 * ax0 + s11 * ax1;
      int v = ((((ay0 * (r0 >> 4)) >> 16) + ((ay1 * (r1 >> 4)) >> 16) + 2) >> 2);
      px[c] = min(max(v, 0), 255);
    }
  }
  float o[3];
  if (mode == 0) {  // rtpose_preprocess: x / 256 - 0.5, channel order kept (BGR)
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (float)px[c] / 256.f - 0.5f;
  } else {          // vgg_preprocess: RGB, /255, ImageNet mean/std
    const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = ((float)px[2 - c] / 255.f - mean[c]) / sd[c];
  }
  float* q = dst + lay_off(ld, d.n, y, x);
  *reinterpret_cast<float4*>(q) = make_float4(o[0], o[1], o[2], 0.f);
  *reinterpret_cast<float4*>(q + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Multi-scale test-time augmentation: dst = beta * dst + alpha * bilinear_resize(src), dense
// NHWC, half-pixel centres, edge clamp (== F.interpolate(bilinear, align_corners=False) and
// cv2.resize INTER_LINEAR on float data).  (sy, sx) = source pixels per destination pixel.
__global__ void resize_bilinear_accum_kernel(const float* __restrict__ src, int hs, int ws,
                                             float* __restrict__ dst, int hd, int wd, int C, int N,
                                             float sy, float sx, float alpha, float beta) {
  const size_t total = (size_t)N * hd * wd * C;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = i % C;
  size_t p = i / C;
  const int x = p % wd;
  p /= wd;
  const int y = p % hd;
  const int n = p / hd;
  float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
  fy = fmaxf(fy, 0.f);
  fx = fmaxf(fx, 0.f);
  int y0 = (int)fy, x0 = (int)fx;
  y0 = min(y0, hs - 1);
  x0 = min(x0, ws - 1);
  const int y1 = min(y0 + 1, hs - 1), x1 = min(x0 + 1, ws - 1);
  const float ly = fminf(fy - (float)y0, 1.f), lx = fminf(fx - (float)x0, 1.f);
  const float* s0 = src + ((size_t)n * hs + y0) * ws * C + c;
  const float* s1 = src + ((size_t)n * hs + y1) * ws * C + c;
  const float top = s0[(size_t)x0 * C] * (1.f - lx) + s0[(size_t)x1 * C] * lx;
  const float bot = s1[(size_t)x0 * C] * (1.f - lx) + s1[(size_t)x1 * C] * lx;
  const float v = top * (1.f - ly) + bot * ly;
  dst[i] = (beta == 0.f ? 0.f : beta * dst[i]) + alpha * v;
}

// Fused test-time-augmentation merge for one scale (BASELINE config 3): reads the stage-6 maps
// of B normal passes (images [0,B)) and, if flip, B x-mirrored passes (images [B,2B)) where the
// net wrote them, forms handle_paf_and_heat's average (evaluate/coco_eval.py:197-242; mirror
// inside the first wv columns only, left/right channel swap, PAF x sign) at the four
// bilinear taps and accumulates alpha * resize(...) into the dense scale-1 maps.  Same
// expressions as flip_merge_kernel followed by resize_bilinear_accum_kernel.
__global__ void tta_accumulate_kernel(const float* __restrict__ heat, Lay lh, const float* __restrict__ paf,
                                      Lay lp, int B, int hs, int wv, float* __restrict__ acc_heat,
                                      float* __restrict__ acc_paf, int hd, int wd, float sy, float sx,
                                      float alpha, float beta, int flip) {
  const size_t total = (size_t)B * hd * wd * 57;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c57 = i % 57;
  size_t p = i / 57;
  const int x = p % wd;
  size_t r = p / wd;
  const int y = r % hd;
  const int b = (int)(r / hd);
  float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
  fy = fmaxf(fy, 0.f);
  fx = fmaxf(fx, 0.f);
  int y0 = (int)fy, x0 = (int)fx;
  y0 = min(y0, hs - 1);
  x0 = min(x0, wv - 1);
  const int y1 = min(y0 + 1, hs - 1), x1 = min(x0 + 1, wv - 1);
  const float ly = fminf(fy - (float)y0, 1.f), lx = fminf(fx - (float)x0, 1.f);
  const bool is_heat = c57 < 19;
  const int c = is_heat ? c57 : c57 - 19;
  const float* src = is_heat ? heat : paf;
  const Lay& l = is_heat ? lh : lp;
  const int sc = is_heat ? kSwapHeat[c] : kSwapPaf[c];
  const bool neg = !is_heat && (sc & 1) == 0;
  auto tap = [&](int yy, int xx) -> float {
    const float a = src[lay_off(l, b, yy, xx) + c];
    if (!flip) return a;
    float v = src[lay_off(l, B + b, yy, wv - 1 - xx) + sc];
    if (neg) v = -v;
    return (a + v) / 2.f;
  };
  const float top = tap(y0, x0) * (1.f - lx) + tap(y0, x1) * lx;
  const float bot = tap(y1, x0) * (1.f - lx) + tap(y1, x1) * lx;
  const float v = top * (1.f - ly) + bot * ly;
  float* d = is_heat ? acc_heat + p * 19 + c : acc_paf + p * 38 + c;
  *d = (beta == 0.f ? 0.f : beta * *d) + alpha * v;
}

// dst(n,y,x,c) = alpha * dst(n,y,x,c) + beta * src_dense[n][y][x][c]
__global__ void layout_axpby_kernel(float* __restrict__ dst, Lay ld, const float* __restrict__ src, int C,
                                    int N, int H, int W, float alpha, float beta) {
  const size_t total = (size_t)N * H * W * C;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = i % C;
  size_t p = i / C;
  const int x = p % W;
  p /= W;
  const int y = p % H;
  const int n = p / H;
  float* d = dst + lay_off(ld, n, y, x) + c;
  *d = alpha * *d + beta * src[i];
}

// ---- ShuffleNetV2 building blocks (lib/network/rtpose_shufflenetV2.py) ----------------
// NCHW -> layout with y = x * scale[c] + shift[c] (BatchNorm2d(3) on the input, :96).
__global__ void nchw_to_layout_affine_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                             Lay l, int C, int cpad, int N, int H, int W,
                                             const float* __restrict__ scale,
                                             const float* __restrict__ shift) {
  const size_t total = (size_t)N * H * W * cpad;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = i % cpad;
  size_t p = i / cpad;
  const int x = p % W;
  p /= W;
  const int y = p % H;
  const int n = p / H;
  float v = 0.f;
  if (c < C) {
    v = src[(((size_t)n * C + c) * H + y) * W + x];
    if (scale) v = v * scale[c] + shift[c];
  }
  dst[lay_off(l, n, y, x) + c] = v;
}

// 3x3 stride-2 pad-1 dense conv, tiny cin (stem 3->24, :97).  HBM-bound: one thread = one
// output pixel x ALL output channels (COUT/4 float4 accumulators), the 9 x 8-channel input
// taps are two float4 loads each, and the weight index depends on loop counters only, so
// the weights arrive through the scalar cache (s_load), not the vector path.
// w[ky][kx][8][COUT].  The input layout's zero gaps are the padding: no bounds tests.
// Same conv fused with the two steps in front of it (rtpose_shufflenetV2.py:96-97): reads the dense
// NCHW fp32 image directly (3 planes, x fastest = coalesced), applies the input BatchNorm2d(3) as a
// per-channel affine (the conv's zero padding is applied AFTER it, as in the reference graph) and
// writes NHWC.  Removes the NHWC8 staging tensor (555 MB at batch 128) and 5/8 of the FMAs.
template <int COUT>
__global__ void stem_conv3x3_s2_nchw_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                            const float* __restrict__ shift, const float* __restrict__ w,
                                            const float* __restrict__ bias, float* __restrict__ out, Lay lo,
                                            int N, int H, int W, int Ho, int Wo, int relu, int out_bf16) {
  constexpr int C4 = COUT / 4;
  const size_t total = (size_t)N * Ho * Wo;
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const int ox = p % Wo;
  p /= Wo;
  const int oy = p % Ho;
  const int n = p / Ho;
  float4 acc[C4];
#pragma unroll
  for (int j = 0; j < C4; ++j) acc[j] = *reinterpret_cast<const float4*>(bias + 4 * j);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    const float* plane = x + ((size_t)n * 3 + c) * H * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy + ky - 1;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox + kx - 1;
        const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const float v = in ? plane[(size_t)iy * W + ix] * sc + sh : 0.f;
        const float* wp = w + (size_t)((ky * 3 + kx) * 8 + c) * COUT;  // packed [ky][kx][8][COUT]
#pragma unroll
        for (int j = 0; j < C4; ++j) {
          const float4 ww = *reinterpret_cast<const float4*>(wp + 4 * j);
          acc[j].x += v * ww.x;
          acc[j].y += v * ww.y;
          acc[j].z += v * ww.z;
          acc[j].w += v * ww.w;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < C4; ++j) {
    if (relu) {
      acc[j].x = fmaxf(acc[j].x, 0.f);
      acc[j].y = fmaxf(acc[j].y, 0.f);
      acc[j].z = fmaxf(acc[j].z, 0.f);
      acc[j].w = fmaxf(acc[j].w, 0.f);
    }
  }
  if (out_bf16) {  // uniform: bf16 plans keep the stem output as 2-byte elements
    unsigned short* oh = reinterpret_cast<unsigned short*>(out) + lay_off(lo, n, oy, ox);
#pragma unroll
    for (int j = 0; j < C4; j += 2) {
      uint4 u;
      u.x = f32_to_bf16_rne(acc[j].x) | ((unsigned)f32_to_bf16_rne(acc[j].y) << 16);
      u.y = f32_to_bf16_rne(acc[j].z) | ((unsigned)f32_to_bf16_rne(acc[j].w) << 16);
      u.z = f32_to_bf16_rne(acc[j + 1].x) | ((unsigned)f32_to_bf16_rne(acc[j + 1].y) << 16);
      u.w = f32_to_bf16_rne(acc[j + 1].z) | ((unsigned)f32_to_bf16_rne(acc[j + 1].w) << 16);
      *reinterpret_cast<uint4*>(oh + 4 * j) = u;
    }
    return;
  }
  float* op = out + lay_off(lo, n, oy, ox);
#pragma unroll
  for (int j = 0; j < C4; ++j) *reinterpret_cast<float4*>(op + 4 * j) = acc[j];
}

// Stem conv + max-pool in ONE launch (rtpose_shufflenetV2.py:96-99: BatchNorm2d(3) -> conv 3x3 s2 p1 + BN +
// ReLU -> MaxPool2d(3, 2, 0, ceil_mode=True)).  A block owns an 8 x 8 tile of pool outputs: it stages the
// 35 x 35 x 3 input patch (affine applied, zero outside the image: the conv's padding comes after the
// BatchNorm) in LDS, evaluates the 17 x 17 x 24 conv outputs the tile's windows
// touch (13 % recomputed at tile borders) into LDS, and writes the window maxima.  The 184 x 184 x 24
// stem tensor (416 MB fp32 / 208 MB bf16 at batch 128) is never stored: the two launches it replaces
// cost 0.60 (fp32) / 0.52 ms (bf16) of a 10.5 / 4.7 ms forward.
//
// Round 5: the kernel was bound by its VALU work around the matrix instructions, not by them (per block and wave ~4300
// issue cycles of address arithmetic, masks and selects against 2240 of MFMA).  Now
//  * VEC: image rows are read as aligned float4 (W % 4 == 0: the patch columns 32 bx - 4 .. 32 bx + 35 are ten whole
//    float4 per row, each entirely inside or outside the image): 5 loads per thread instead of 15, a quarter of the
//    index arithmetic; the scalar loader stays for other widths;
//  * the ReLU and the "conv output does not exist" mask left the MFMA write-out: the pool starts from 0 (max(0, .) IS
//    the ReLU) and only the tiles at the map's right / bottom edge test the window positions; the write-out is 16
//    ds_write_b32 at immediate offsets;
//  * tap offsets / filter registers are picked from compile-time tables and loaded before the patch arrives; fragments
//    wholly below the conv map are skipped; the waves that take the odd fragments rotate with the block index.
constexpr int kSpT = 8;                   // pool outputs per tile side
constexpr int kSpS = 2 * kSpT + 1;        // conv outputs per tile side (17)
constexpr int kSpI = 2 * kSpS + 1;        // input pixels per tile side (35)
constexpr int kSpIP = 40;                 // LDS row pitch of the input patch: column j holds image column ix0 - 3 + j
constexpr int kSpX0 = 3;                  // LDS column of the tile's first input pixel (ix0 = 32 bx - 1)
constexpr int kSpC = 24;
constexpr int kSpCP = 28;                 // LDS pitch of one conv output's channels (conflict skew)
constexpr int kSpF = (kSpS * kSpS + 31) / 32;  // MFMA fragments of 32 conv positions per tile (10)
// tap k = (ky * 3 + kx) * 3 + c of the K dimension (k = 27: tap 26 again, under a zero weight)
__host__ __device__ constexpr int sp_tap_off(int k) {
  const int kk = k < 26 ? k : 26;
  return ((kk % 3) * kSpI + (kk / 3) / 3) * kSpIP + (kk / 3) % 3 + kSpX0;
}
__host__ __device__ constexpr int sp_tap_w(int k) {  // packed filters [ky][kx][8][24]
  const int kk = k < 26 ? k : 26;
  return ((kk / 3) * 8 + kk % 3) * kSpC;
}
template <int OUT_BF16, int VEC>
__global__ __launch_bounds__(256) void stem_pool_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ w,
                                                        const float* __restrict__ bias, void* __restrict__ out_v,
                                                        Lay lo, int H, int W, int H1, int W1, int H2, int W2) {
  __shared__ __attribute__((aligned(16))) float s_in[3 * kSpI][kSpIP];
  __shared__ __attribute__((aligned(16))) float s_st[kSpF * 32][kSpCP];
  const int tid = threadIdx.x;
  const int n = blockIdx.z;
  const int py0 = blockIdx.y * kSpT, px0 = blockIdx.x * kSpT;
  const int sy0 = 2 * py0, sx0 = 2 * px0;        // first conv output of the tile
  const int iy0 = 2 * sy0 - 1, ix0 = 2 * sx0 - 1;  // first input pixel of the tile (may be -1: padding)
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
  // filter registers of the matrix phase (B operand: channel l31, tap 2 j + kh), requested before the patch
  float bw[14];
#pragma unroll
  for (int j = 0; j < 14; ++j) {
    const int wi = (kh ? sp_tap_w(2 * j + 1) : sp_tap_w(2 * j)) + min(l31, kSpC - 1);
    const float t = w[wi];
    bw[j] = (l31 < kSpC && 2 * j + kh < 27) ? t : 0.f;
  }
  const float bv = l31 < kSpC ? bias[min(l31, kSpC - 1)] : 0.f;
  const float sc[3] = {scale ? scale[0] : 1.f, scale ? scale[1] : 1.f, scale ? scale[2] : 1.f};
  const float sh[3] = {scale ? shift[0] : 0.f, scale ? shift[1] : 0.f, scale ? shift[2] : 0.f};
  const float* xn = x + (size_t)n * 3 * H * W;
  if (VEC) {  // rows as aligned float4: thread = (float4 q of the row, row r0 + 25 u of the 105 patch rows)
    constexpr int NR = 5;
    const int q = tid % 10, r0 = tid / 10;
    const int ix4 = ix0 - kSpX0 + 4 * q;
    const bool xin = ix4 >= 0 && ix4 < W && tid < 250;
    const int ixc = min(max(ix4, 0), W - 4);
    float4 v[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {  // branch-free: clamped addresses, masked afterwards
      const int r = min(r0 + 25 * u, 3 * kSpI - 1);
      const int c = r / kSpI, yy = r - c * kSpI;
      const int iy = iy0 + yy;
      const bool in = xin && iy >= 0 && iy < H;
      const float4 t = *reinterpret_cast<const float4*>(xn + (size_t)(unsigned)(c * H + min(max(iy, 0), H - 1)) * (unsigned)W + ixc);
      const float a = c == 0 ? sc[0] : (c == 1 ? sc[1] : sc[2]), b = c == 0 ? sh[0] : (c == 1 ? sh[1] : sh[2]);
      v[u] = in ? make_float4(t.x * a + b, t.y * a + b, t.z * a + b, t.w * a + b) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int r = r0 + 25 * u;
      if (tid < 250 && r < 3 * kSpI) *reinterpret_cast<float4*>(&s_in[r][4 * q]) = v[u];
    }
  } else {  // any width: one pixel per load, all of a thread's loads issued before the first is used
    constexpr int NL = (3 * kSpI * kSpI + 255) / 256;
    float v[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int i = min(tid + 256 * u, 3 * kSpI * kSpI - 1);
      const int xx = i % kSpI;
      const int r = i / kSpI;
      const int yy = r % kSpI, c = r / kSpI;
      const int iy = iy0 + yy, ix = ix0 + xx;
      const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const float t = xn[(unsigned)(c * H + min(max(iy, 0), H - 1)) * (unsigned)W + (unsigned)min(max(ix, 0), W - 1)];
      const float a = c == 0 ? sc[0] : (c == 1 ? sc[1] : sc[2]), b = c == 0 ? sh[0] : (c == 1 ? sh[1] : sh[2]);
      v[u] = in ? t * a + b : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int i = tid + 256 * u;
      if (i < 3 * kSpI * kSpI) s_in[i / kSpI][i % kSpI + kSpX0] = v[u];
    }
  }
  __syncthreads();
  // conv outputs on the matrix pipe (round 4; v_mfma_f32_32x32x2_f32): the 17 x 17 positions are the M dimension (10
  // fragments of 32), the 24 output channels N (one fragment), the 27 taps K (14 steps of 2, the last one half empty).
  // A operand: lane (position, kh) reads tap k = 2 j + kh of ITS position from the input patch in LDS - an im2col that
  // exists only as 14 ds_read_b32 per fragment; B operand: the filter value of channel l31 for that tap, 14 registers
  // loaded once per block.  The scalar-FMA form this replaces issued 240 M VALU instructions per launch (address
  // arithmetic around 27 LDS reads and 324 FMAs per position, SQ_INSTS_VALU, profiles/r04_shufflenet_pmc_sq.txt)
  // and was bound by exactly that: 0.38 ms of an 8.9 / 4.05 ms forward.
  {
    typedef float floatx16 __attribute__((ext_vector_type(16)));
    // fragments whose positions all lie below the conv map are not computed (the tiles of the last tile row)
    const int nf = min(kSpF, (min(kSpS, H1 - sy0) * kSpS + 31) / 32);
    const int w0 = (wv + blockIdx.x + blockIdx.y) & 3;  // the SIMDs take turns at the third fragment
    for (int f = w0; f < nf; f += 4) {
      const int p = min(f * 32 + l31, kSpS * kSpS - 1);  // the position this lane feeds (rows past the end replay the last)
      const int sy = p / kSpS, sx = p - sy * kSpS;
      const float* ab = &s_in[2 * sy][2 * sx];
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bv;
#pragma unroll
      for (int j = 0; j < 14; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[kh ? sp_tap_off(2 * j + 1) : sp_tap_off(2 * j)], bw[j], acc, 0, 0, 0);
      // C layout: lane = channel l31, register r = position (r / 4) * 8 + 4 kh + r % 4 of the fragment
      if (l31 < kSpC) {
        float* st = &s_st[f * 32 + 4 * kh][l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) st[((r >> 2) * 8 + (r & 3)) * kSpCP] = acc[r];
      }
    }
  }
  __syncthreads();
  // window maxima: item = (pool output, 8-channel group).  The maxima start from 0 - the ReLU; windows that hang over the
  // conv map (ceil mode) skip the positions that do not exist (only tiles on the map's right / bottom edge can have any).
  const bool edge = sy0 + kSpS > H1 || sx0 + kSpS > W1;
  for (int it = tid; it < kSpT * kSpT * 3; it += 256) {
    const int g = it % 3, q = it / 3;
    const int ty = q / kSpT, tx = q - ty * kSpT;
    const int py = py0 + ty, px = px0 + tx;
    if (py >= H2 || px >= W2) continue;
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = 0.f;
    auto window = [&](auto masked) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          if (masked && !(sy0 + 2 * ty + dy < H1 && sx0 + 2 * tx + dx < W1)) continue;
          const float* sp = &s_st[(2 * ty + dy) * kSpS + 2 * tx + dx][8 * g];
          const float4 a = *reinterpret_cast<const float4*>(sp), b = *reinterpret_cast<const float4*>(sp + 4);
          m[0] = fmaxf(m[0], a.x), m[1] = fmaxf(m[1], a.y), m[2] = fmaxf(m[2], a.z), m[3] = fmaxf(m[3], a.w);
          m[4] = fmaxf(m[4], b.x), m[5] = fmaxf(m[5], b.y), m[6] = fmaxf(m[6], b.z), m[7] = fmaxf(m[7], b.w);
        }
    };
    if (edge)
      window(std::true_type{});
    else
      window(std::false_type{});
    if (OUT_BF16) {
      unsigned short* o = reinterpret_cast<unsigned short*>(out_v) + lay_off(lo, n, py, px) + 8 * g;
      uint4 u;
      u.x = f32_to_bf16_rne(m[0]) | ((unsigned)f32_to_bf16_rne(m[1]) << 16);
      u.y = f32_to_bf16_rne(m[2]) | ((unsigned)f32_to_bf16_rne(m[3]) << 16);
      u.z = f32_to_bf16_rne(m[4]) | ((unsigned)f32_to_bf16_rne(m[5]) << 16);
      u.w = f32_to_bf16_rne(m[6]) | ((unsigned)f32_to_bf16_rne(m[7]) << 16);
      *reinterpret_cast<uint4*>(o) = u;
    } else {
      float* o = reinterpret_cast<float*>(out_v) + lay_off(lo, n, py, px) + 8 * g;
      *reinterpret_cast<float4*>(o) = make_float4(m[0], m[1], m[2], m[3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(m[4], m[5], m[6], m[7]);
    }
  }
}

template <int COUT>
__global__ void stem_conv3x3_s2_kernel(const float* __restrict__ in, Lay li, const float* __restrict__ w,
                                       const float* __restrict__ bias, float* __restrict__ out, Lay lo,
                                       int N, int Ho, int Wo, int relu) {
  constexpr int C4 = COUT / 4;
  const size_t total = (size_t)N * Ho * Wo;
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const int x = p % Wo;
  p /= Wo;
  const int y = p % Ho;
  const int n = p / Ho;
  float4 acc[C4];
#pragma unroll
  for (int j = 0; j < C4; ++j) acc[j] = *reinterpret_cast<const float4*>(bias + 4 * j);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      // input pixel (2y+ky-1, 2x+kx-1): offsets of -1 land in the layout gaps
      const float* ip = in + (long long)lay_off(li, n, 2 * y + ky, 2 * x + kx) -
                        (long long)(li.ws + 1) * li.cstride;
      const float4 v0 = *reinterpret_cast<const float4*>(ip);
      const float4 v1 = *reinterpret_cast<const float4*>(ip + 4);
      const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      const float* wp = w + (size_t)(ky * 3 + kx) * 8 * COUT;
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int j = 0; j < C4; ++j) {
          const float4 ww = *reinterpret_cast<const float4*>(wp + c * COUT + 4 * j);
          acc[j].x += v[c] * ww.x;
          acc[j].y += v[c] * ww.y;
          acc[j].z += v[c] * ww.z;
          acc[j].w += v[c] * ww.w;
        }
    }
  float* op = out + lay_off(lo, n, y, x);
#pragma unroll
  for (int j = 0; j < C4; ++j) {
    float4 a = acc[j];
    if (relu) {
      a.x = fmaxf(a.x, 0.f);
      a.y = fmaxf(a.y, 0.f);
      a.z = fmaxf(a.z, 0.f);
      a.w = fmaxf(a.w, 0.f);
    }
    *reinterpret_cast<float4*>(op + 4 * j) = a;
  }
}

// MaxPool2d(3, 2, 0, ceil_mode=True): windows are clipped to the input.
__global__ void maxpool3x3s2_ceil_kernel(const float* __restrict__ src, Lay ls, float* __restrict__ dst,
                                         Lay ld, int C, int N, int H, int W, int Ho, int Wo) {
  const int c4 = C >> 2;
  const size_t total = (size_t)N * Ho * Wo * c4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (i % c4) * 4;
  size_t p = i / c4;
  const int x = p % Wo;
  p /= Wo;
  const int y = p % Ho;
  const int n = p / Ho;
  float4 r = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = 2 * y + dy, xx = 2 * x + dx;
      if (yy < H && xx < W) {
        const float4 v = *reinterpret_cast<const float4*>(src + lay_off(ls, n, yy, xx) + c);
        r.x = fmaxf(r.x, v.x);
        r.y = fmaxf(r.y, v.y);
        r.z = fmaxf(r.z, v.z);
        r.w = fmaxf(r.w, v.w);
      }
    }
  *reinterpret_cast<float4*>(dst + lay_off(ld, n, y, x) + c) = r;
}

// depthwise 3x3, pad 1 (the input layout's zero gaps), stride 1 or 2, + bias; w[9][C].
__global__ void dwconv3x3_kernel(const float* __restrict__ in, Lay li, const float* __restrict__ w,
                                 const float* __restrict__ bias, float* __restrict__ out, Lay lo, int C,
                                 int N, int Ho, int Wo, int stride) {
  const int c4 = C >> 2;
  const size_t total = (size_t)N * Ho * Wo * c4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (i % c4) * 4;
  size_t p = i / c4;
  const int x = p % Wo;
  p /= Wo;
  const int y = p % Ho;
  const int n = p / Ho;
  float4 acc = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float* ip = in + (long long)lay_off(li, n, stride * y + ky, stride * x + kx) -
                        (long long)(li.ws + 1) * li.cstride + c;
      const float4 v = *reinterpret_cast<const float4*>(ip);
      const float4 ww = *reinterpret_cast<const float4*>(w + (size_t)(ky * 3 + kx) * C + c);
      acc.x += v.x * ww.x;
      acc.y += v.y * ww.y;
      acc.z += v.z * ww.z;
      acc.w += v.w * ww.w;
    }
  *reinterpret_cast<float4*>(out + lay_off(lo, n, y, x) + c) = acc;
}

// 4 consecutive source channels per thread (one 16-byte load), 4 mapped 4-byte stores
__global__ void layout_copy_cmap4_kernel(const float* __restrict__ src, Lay ls, float* __restrict__ dst,
                                         Lay ld, int C, const int32_t* __restrict__ cmap, int N, int H,
                                         int W) {
  const int c4 = (C + 3) >> 2;
  const size_t total = (size_t)N * H * W * c4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (i % c4) * 4;
  size_t p = i / c4;
  const int x = p % W;
  p /= W;
  const int y = p % H;
  const int n = p / H;
  const float4 v = *reinterpret_cast<const float4*>(src + lay_off(ls, n, y, x) + c);
  float* d = dst + lay_off(ld, n, y, x) - ld.choff;
  const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (c + k < C) d[cmap[c + k]] = vv[k];
}

__global__ void layout_copy_cmap_kernel(const float* __restrict__ src, Lay ls, float* __restrict__ dst,
                                        Lay ld, int C, const int32_t* __restrict__ cmap, int N, int H,
                                        int W) {
  const size_t total = (size_t)N * H * W * C;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = i % C;
  size_t p = i / C;
  const int x = p % W;
  p /= W;
  const int y = p % H;
  const int n = p / H;
  dst[lay_off(ld, n, y, x) - ld.choff + cmap[c]] = src[lay_off(ls, n, y, x) + c];
}

static inline unsigned nblocks(size_t total, int threads) {
  return (unsigned)((total + threads - 1) / threads);
}


// ---- bf16 activation buffers (conv_mfma_bf16.hip): conversions at the edges of the net ----

// One thread per (pixel, 8-channel piece): 16-byte stores.  src is dense NCHW fp32 (SRC_NCHW)
// or a layout slice; channels [C, cpad) are written as zero.
template <bool SRC_NCHW>
__global__ void to_bf16_layout_kernel(const float* __restrict__ src, Lay ls, unsigned short* __restrict__ dst,
                                      Lay ld, int C, int cpad, int N, int H, int W) {
  const int pieces = cpad >> 3;
  const size_t total = (size_t)N * H * W * pieces;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  size_t p;
  int pc;
  if (SRC_NCHW) {  // x fastest: coalesced reads of each channel plane
    p = i % ((size_t)N * H * W);
    pc = (int)(i / ((size_t)N * H * W));
  } else {
    pc = (int)(i % pieces);
    p = i / pieces;
  }
  const int x = p % W;
  size_t r = p / W;
  const int y = r % H;
  const int n = (int)(r / H);
  unsigned short v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = pc * 8 + e;
    float f = 0.f;
    if (c < C) f = SRC_NCHW ? src[(((size_t)n * C + c) * H + y) * W + x] : src[lay_off(ls, n, y, x) + c];
    v[e] = f32_to_bf16_rne(f);
  }
  uint4 o;
  o.x = v[0] | ((unsigned)v[1] << 16);
  o.y = v[2] | ((unsigned)v[3] << 16);
  o.z = v[4] | ((unsigned)v[5] << 16);
  o.w = v[6] | ((unsigned)v[7] << 16);
  *reinterpret_cast<uint4*>(dst + lay_off(ld, n, y, x) + pc * 8) = o;
}

__global__ void layout_bf16_to_f32_kernel(const unsigned short* __restrict__ src, Lay ls,
                                          float* __restrict__ dst, Lay ld, int C, int N, int H, int W) {
  const size_t total = (size_t)N * H * W * C;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = i % C;
  size_t p = i / C;
  const int x = p % W;
  p /= W;
  const int y = p % H;
  const int n = p / H;
  dst[lay_off(ld, n, y, x) + c] = bf16_to_f32(src[lay_off(ls, n, y, x) + c]);
}


// ---- "split" activations of the bf16x3 plans: value v = hi + lo, hi = bf16(v), lo = bf16(v - hi),
//      stored per 8 channels as [hi x 8 | lo x 8] (32 bytes); layouts count elements (2 per channel)
template <bool SRC_NCHW>
__global__ void to_split_layout_kernel(const float* __restrict__ src, Lay ls, unsigned short* __restrict__ dst,
                                       Lay ld, int C, int cpad, int N, int H, int W) {
  const int groups = cpad >> 3;
  const size_t total = (size_t)N * H * W * groups;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  size_t p;
  int gc;
  if (SRC_NCHW) {
    p = i % ((size_t)N * H * W);
    gc = (int)(i / ((size_t)N * H * W));
  } else {
    gc = (int)(i % groups);
    p = i / groups;
  }
  const int x = p % W;
  size_t r = p / W;
  const int y = r % H;
  const int n = (int)(r / H);
  unsigned short hi[8], lo[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = gc * 8 + e;
    float f = 0.f;
    if (c < C) f = SRC_NCHW ? src[(((size_t)n * C + c) * H + y) * W + x] : src[lay_off(ls, n, y, x) + c];
    hi[e] = f32_to_bf16_rne(f);
    lo[e] = f32_to_bf16_rne(f - bf16_to_f32(hi[e]));
  }
  uint4 a, b;
  a.x = hi[0] | ((unsigned)hi[1] << 16);
  a.y = hi[2] | ((unsigned)hi[3] << 16);
  a.z = hi[4] | ((unsigned)hi[5] << 16);
  a.w = hi[6] | ((unsigned)hi[7] << 16);
  b.x = lo[0] | ((unsigned)lo[1] << 16);
  b.y = lo[2] | ((unsigned)lo[3] << 16);
  b.z = lo[4] | ((unsigned)lo[5] << 16);
  b.w = lo[6] | ((unsigned)lo[7] << 16);
  uint4* d = reinterpret_cast<uint4*>(dst + lay_off(ld, n, y, x) + gc * 16);
  d[0] = a;
  d[1] = b;
}

__global__ void layout_split_to_f32_kernel(const unsigned short* __restrict__ src, Lay ls,
                                           float* __restrict__ dst, Lay ld, int C, int N, int H, int W) {
  const size_t total = (size_t)N * H * W * C;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = i % C;
  size_t p = i / C;
  const int x = p % W;
  p /= W;
  const int y = p % H;
  const int n = p / H;
  // ls.choff counts elements and may sit inside an 8-channel group (the 19 heat-map channels
  // start at channel 166 of the concat buffer): address by absolute channel
  const int ca = (ls.choff >> 1) + c;
  const unsigned short* s = src + ((size_t)ls.lead + (size_t)(n * ls.hs + y) * ls.ws + x) * ls.cstride +
                            (ca >> 3) * 16 + (ca & 7);
  dst[lay_off(ld, n, y, x) + c] = bf16_to_f32(s[0]) + bf16_to_f32(s[8]);
}


// ---- bf16 forms of the ShuffleNetV2 building blocks (bf16 plans, BASELINE config 4 "fp32 and
//      bf16"): bf16 activations in HBM, fp32 arithmetic, 8 channels (16 bytes) per thread ----
struct bf8 {
  float v[8];
};
__device__ __forceinline__ bf8 load_bf8(const unsigned short* p) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  bf8 r;
  r.v[0] = __uint_as_float(u.x << 16);
  r.v[1] = __uint_as_float(u.x & 0xffff0000u);
  r.v[2] = __uint_as_float(u.y << 16);
  r.v[3] = __uint_as_float(u.y & 0xffff0000u);
  r.v[4] = __uint_as_float(u.z << 16);
  r.v[5] = __uint_as_float(u.z & 0xffff0000u);
  r.v[6] = __uint_as_float(u.w << 16);
  r.v[7] = __uint_as_float(u.w & 0xffff0000u);
  return r;
}
__device__ __forceinline__ void store_bf8(unsigned short* p, const bf8& r) {
  uint4 u;
  u.x = f32_to_bf16_rne(r.v[0]) | ((unsigned)f32_to_bf16_rne(r.v[1]) << 16);
  u.y = f32_to_bf16_rne(r.v[2]) | ((unsigned)f32_to_bf16_rne(r.v[3]) << 16);
  u.z = f32_to_bf16_rne(r.v[4]) | ((unsigned)f32_to_bf16_rne(r.v[5]) << 16);
  u.w = f32_to_bf16_rne(r.v[6]) | ((unsigned)f32_to_bf16_rne(r.v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = u;
}

__global__ void maxpool3x3s2_ceil_bf16_kernel(const unsigned short* __restrict__ src, Lay ls,
                                              unsigned short* __restrict__ dst, Lay ld, int C, int N, int H,
                                              int W, int Ho, int Wo) {
  const int c8 = C >> 3;
  const size_t total = (size_t)N * Ho * Wo * c8;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (i % c8) * 8;
  size_t p = i / c8;
  const int x = p % Wo;
  p /= Wo;
  const int y = p % Ho;
  const int n = p / Ho;
  bf8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r.v[e] = -INFINITY;
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = 2 * y + dy, xx = 2 * x + dx;
      if (yy < H && xx < W) {
        const bf8 v = load_bf8(src + lay_off(ls, n, yy, xx) + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) r.v[e] = fmaxf(r.v[e], v.v[e]);
      }
    }
  store_bf8(dst + lay_off(ld, n, y, x) + c, r);
}

__global__ void dwconv3x3_bf16_kernel(const unsigned short* __restrict__ in, Lay li, const float* __restrict__ w,
                                      const float* __restrict__ bias, unsigned short* __restrict__ out, Lay lo,
                                      int C, int N, int Ho, int Wo, int stride) {
  const int c8 = C >> 3;
  const size_t total = (size_t)N * Ho * Wo * c8;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (i % c8) * 8;
  size_t p = i / c8;
  const int x = p % Wo;
  p /= Wo;
  const int y = p % Ho;
  const int n = p / Ho;
  bf8 acc;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc.v[e] = bias[c + e];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const unsigned short* ip = in + (long long)lay_off(li, n, stride * y + ky, stride * x + kx) -
                                 (long long)(li.ws + 1) * li.cstride + c;
      const bf8 v = load_bf8(ip);
      const float* wp = w + (size_t)(ky * 3 + kx) * C + c;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc.v[e] += v.v[e] * wp[e];
    }
  store_bf8(out + lay_off(lo, n, y, x) + c, acc);
}


// depthwise 3x3, stride 1: FOUR consecutive output pixels of a row per thread.  The plain kernel
// issues 9 activation + 9 weight loads per output (it is load-instruction bound, ~2 TB/s); here
// the 3x6 input window and the 9 weight vectors are loaded once for four outputs (28 vs 76
// loads).  BF = 0: fp32 activations, 4 channels per thread; BF = 1: bf16 activations, 8 channels.
template <int BF>
__global__ void dwconv3x3_s1x4_kernel(const void* __restrict__ in_v, Lay li, const float* __restrict__ w,
                                      const float* __restrict__ bias, void* __restrict__ out_v, Lay lo, int C,
                                      int N, int H, int W) {
  constexpr int CV = BF ? 8 : 4;  // channels per thread
  const int cg = C / CV, xg = (W + 3) >> 2;
  const size_t total = (size_t)N * H * xg * cg;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cg) * CV;
  size_t p = i / cg;
  const int x0 = (int)(p % xg) * 4;
  p /= xg;
  const int y = (int)(p % H);
  const int n = (int)(p / H);
  float acc[4][CV];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int e = 0; e < CV; ++e) acc[o][e] = bias[c + e];
  const float* in_f = static_cast<const float*>(in_v);
  const unsigned short* in_h = static_cast<const unsigned short*>(in_v);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    float wv[3][CV];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int e = 0; e < CV; ++e) wv[kx][e] = w[(size_t)(ky * 3 + kx) * C + c + e];
    // input row y + ky - 1, columns x0 - 1 .. x0 + 4 (offsets of -1 land in the layout gaps; columns
    // past the row end read the gap / the next row and only feed outputs that are not stored)
    const long long row = (long long)lay_off(li, n, y + ky, x0) - (long long)(li.ws + 1) * li.cstride + c;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float v[CV];
      if (BF) {
        const bf8 t = load_bf8(in_h + row + (long long)j * li.cstride);
#pragma unroll
        for (int e = 0; e < CV; ++e) v[e] = t.v[e & 7];
      } else {
        const float4 t = *reinterpret_cast<const float4*>(in_f + row + (long long)j * li.cstride);
        v[0] = t.x;
        v[1] = t.y;
        v[2] = t.z;
        v[3] = t.w;
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int kx = j - o;  // input column j feeds output o through tap kx = j - o
        if (kx >= 0 && kx < 3) {
#pragma unroll
          for (int e = 0; e < CV; ++e) acc[o][e] += v[e] * wv[kx][e];
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    if (x0 + o >= W) break;
    if (BF) {
      bf8 r;
#pragma unroll
      for (int e = 0; e < 8; ++e) r.v[e] = acc[o][e & (CV - 1)];
      store_bf8(static_cast<unsigned short*>(out_v) + lay_off(lo, n, y, x0 + o) + c, r);
    } else {
      *reinterpret_cast<float4*>(static_cast<float*>(out_v) + lay_off(lo, n, y, x0 + o) + c) =
          make_float4(acc[o][0], acc[o][1], acc[o][2], acc[o][3]);
    }
  }
}

// 8 consecutive source channels per thread (one 16-byte load), up to 8 mapped 2-byte stores
__global__ void layout_copy_cmap_bf16_kernel(const unsigned short* __restrict__ src, Lay ls,
                                             unsigned short* __restrict__ dst, Lay ld, int C,
                                             const int32_t* __restrict__ cmap, int N, int H, int W) {
  const int c8 = (C + 7) >> 3;
  const size_t total = (size_t)N * H * W * c8;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (i % c8) * 8;
  size_t p = i / c8;
  const int x = p % W;
  p /= W;
  const int y = p % H;
  const int n = p / H;
  const uint4 u = *reinterpret_cast<const uint4*>(src + lay_off(ls, n, y, x) + c);
  const unsigned short e[8] = {(unsigned short)(u.x & 0xffff), (unsigned short)(u.x >> 16),
                               (unsigned short)(u.y & 0xffff), (unsigned short)(u.y >> 16),
                               (unsigned short)(u.z & 0xffff), (unsigned short)(u.z >> 16),
                               (unsigned short)(u.w & 0xffff), (unsigned short)(u.w >> 16)};
  unsigned short* d = dst + lay_off(ld, n, y, x) - ld.choff;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (c + j < C) d[cmap[c + j]] = e[j];
}

}  // namespace rtpose

using namespace rtpose;

extern "C" {

const char* rtpose_version(void) { return "rtpose_mi355x 0.1 (gfx950)"; }
const char* rtpose_last_error(void) { return rtpose::err_buf(); }

size_t rtpose_layout_pixels(const rtpose_layout* l, int N, int H, int W) {
  (void)H;
  (void)W;
  // tail slack: 2-D tiles may read up to ~36 rows past the last image, and the strip
  // mode's staging runs up to 14 piece sets (64 pixels each) past a halo
  return (size_t)l->lead + (size_t)N * l->hs * l->ws + (size_t)40 * l->ws + 4352;
}

int rtpose_nchw_to_layout(const float* src, float* dst, const rtpose_layout* ldst, int C, int cpad,
                          int N, int H, int W, void* stream) {
  if (cpad < C || ldst->choff + cpad > ldst->cstride)
    return fail(RTPOSE_E_INVAL, "nchw_to_layout: bad channel counts");
  const size_t total = (size_t)N * H * W * cpad;
  if (!total) return 0;
  hipLaunchKernelGGL(nchw_to_layout_kernel, dim3(nblocks(total, 256)), dim3(256), 0,
                     as_stream(stream), src, dst, to_lay(ldst), C, cpad, N, H, W);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_layout_to_nchw(const float* src, const rtpose_layout* lsrc, float* dst, int C, int N, int H,
                          int W, void* stream) {
  const size_t total = (size_t)N * H * W * C;
  if (!total) return 0;
  hipLaunchKernelGGL(layout_to_nchw_kernel, dim3(nblocks(total, 256)), dim3(256), 0,
                     as_stream(stream), src, to_lay(lsrc), dst, C, N, H, W);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_layout_copy(const float* src, const rtpose_layout* lsrc, float* dst,
                       const rtpose_layout* ldst, int C, int N, int H, int W, void* stream) {
  const bool vec = !(C % 4) && !(lsrc->cstride % 4) && !(lsrc->choff % 4) && !(ldst->cstride % 4) &&
                   !(ldst->choff % 4);
  if (vec) {
    const size_t total = (size_t)N * H * W * (C / 4);
    if (!total) return 0;
    hipLaunchKernelGGL(layout_copy_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream),
                       src, to_lay(lsrc), dst, to_lay(ldst), C, N, H, W);
  } else {
    const size_t total = (size_t)N * H * W * C;
    if (!total) return 0;
    hipLaunchKernelGGL(layout_copy_scalar_kernel, dim3(nblocks(total, 256)), dim3(256), 0,
                       as_stream(stream), src, to_lay(lsrc), dst, to_lay(ldst), C, N, H, W);
  }
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_maxpool2x2(const float* in, const rtpose_layout* lin, float* out, const rtpose_layout* lout,
                      int C, int N, int H, int W, void* stream) {
  if ((C % 4) || (lin->cstride % 4) || (lin->choff % 4) || (lout->cstride % 4) || (lout->choff % 4))
    return fail(RTPOSE_E_INVAL, "maxpool: channel slices must be 16-byte aligned");
  const int Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)N * Ho * Wo * (C / 4);
  if (!total) return 0;
  hipLaunchKernelGGL(maxpool2x2_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream), in,
                     to_lay(lin), out, to_lay(lout), C, N, Ho, Wo);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_nchw_to_layout_affine(const float* src, float* dst, const rtpose_layout* ldst, int C, int cpad,
                                 int N, int H, int W, const float* scale, const float* shift,
                                 void* stream) {
  if (cpad < C || ldst->choff + cpad > ldst->cstride || (scale && !shift))
    return fail(RTPOSE_E_INVAL, "nchw_to_layout_affine: bad arguments");
  const size_t total = (size_t)N * H * W * cpad;
  if (!total) return 0;
  hipLaunchKernelGGL(nchw_to_layout_affine_kernel, dim3(nblocks(total, 256)), dim3(256), 0,
                     as_stream(stream), src, dst, to_lay(ldst), C, cpad, N, H, W, scale, shift);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_stem_conv3x3_s2(const float* in, const rtpose_layout* lin, const float* w, const float* bias,
                           float* out, const rtpose_layout* lout, int cin_pad, int cout, int N, int H,
                           int W, int relu, void* stream) {
  if ((cout % 4) || (lout->cstride % 4) || (lout->choff % 4) || lin->ws < W + 1 || lin->hs < H + 1 ||
      lin->lead < lin->ws + 1 || lin->choff + cin_pad > lin->cstride)
    return fail(RTPOSE_E_INVAL, "stem_conv: unsupported layout / channel count");
  if (cin_pad != 8 || cout != 24 || (lin->cstride % 4) || (lin->choff % 4))
    return fail(RTPOSE_E_INVAL, "stem_conv: only cin_pad = 8, cout = 24 is instantiated");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const size_t total = (size_t)N * Ho * Wo;
  if (!total) return 0;
  hipLaunchKernelGGL(stem_conv3x3_s2_kernel<24>, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream), in,
                     to_lay(lin), w, bias, out, to_lay(lout), N, Ho, Wo, relu);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_stem_conv3x3_s2_nchw(const float* x_nchw, const float* scale, const float* shift, const float* w,
                                const float* bias, float* out, const rtpose_layout* lout, int cout, int N, int H,
                                int W, int relu, void* stream) {
  return rtpose_stem_conv3x3_s2_nchw_ex(x_nchw, scale, shift, w, bias, out, lout, cout, N, H, W, relu, 0, stream);
}

int rtpose_stem_conv3x3_s2_nchw_ex(const float* x_nchw, const float* scale, const float* shift, const float* w,
                                   const float* bias, void* out_v, const rtpose_layout* lout, int cout, int N,
                                   int H, int W, int relu, int out_bf16, void* stream) {
  float* out = static_cast<float*>(out_v);
  const int al = out_bf16 ? 8 : 4;
  if (cout != 24 || (lout->cstride % al) || (lout->choff % al))
    return fail(RTPOSE_E_INVAL, "stem_conv3x3_s2_nchw: only cout=24 (ShuffleNetV2 x1.0) is instantiated");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const size_t total = (size_t)N * Ho * Wo;
  if (!total) return 0;
  hipLaunchKernelGGL(stem_conv3x3_s2_nchw_kernel<24>, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream),
                     x_nchw, scale, shift, w, bias, out, to_lay(lout), N, H, W, Ho, Wo, relu, out_bf16 ? 1 : 0);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_stem_pool_nchw(const float* x_nchw, const float* scale, const float* shift, const float* w,
                          const float* bias, void* out, const rtpose_layout* lout, int cout, int N, int H, int W,
                          int out_bf16, void* stream) {
  const int al = out_bf16 ? 8 : 4;
  if (!x_nchw || !w || !bias || !out || !lout) return fail(RTPOSE_E_INVAL, "stem_pool: NULL argument");
  if (cout != 24 || (lout->cstride % al) || (lout->choff % al) || lout->choff + 24 > lout->cstride || (scale && !shift))
    return fail(RTPOSE_E_INVAL, "stem_pool: only cout=24 (ShuffleNetV2 x1.0), 16-byte aligned output slice");
  if (N <= 0 || H < 8 || W < 8) return fail(RTPOSE_E_INVAL, "stem_pool: need N >= 1, H, W >= 8");
  const int H1 = (H - 1) / 2 + 1, W1 = (W - 1) / 2 + 1;              // conv 3x3 s2 p1
  const int H2 = (H1 - 3 + 1) / 2 + 1, W2 = (W1 - 3 + 1) / 2 + 1;    // max-pool 3/2, ceil mode
  const dim3 grid(ceil_div(W2, kSpT), ceil_div(H2, kSpT), N);
  // rows as aligned float4 when the image allows it (every row starts on a 16-byte boundary)
  const bool vec = W % 4 == 0 && reinterpret_cast<uintptr_t>(x_nchw) % 16 == 0;
#define RTPOSE_STEM_POOL(B, V)                                                                                         \
  hipLaunchKernelGGL((stem_pool_kernel<B, V>), grid, dim3(256), 0, as_stream(stream), x_nchw, scale, shift, w, bias, \
                     out, to_lay(lout), H, W, H1, W1, H2, W2)
  if (out_bf16) {
    if (vec) RTPOSE_STEM_POOL(1, 1); else RTPOSE_STEM_POOL(1, 0);
  } else {
    if (vec) RTPOSE_STEM_POOL(0, 1); else RTPOSE_STEM_POOL(0, 0);
  }
#undef RTPOSE_STEM_POOL
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_maxpool3x3s2_ceil(const float* in, const rtpose_layout* lin, float* out, const rtpose_layout* lout,
                             int C, int N, int H, int W, void* stream) {
  if ((C % 4) || (lin->cstride % 4) || (lin->choff % 4) || (lout->cstride % 4) || (lout->choff % 4) || H < 3 ||
      W < 3)
    return fail(RTPOSE_E_INVAL, "maxpool3x3s2: channel slices must be 16-byte aligned, H,W >= 3");
  const int Ho = (H - 3 + 1) / 2 + 1, Wo = (W - 3 + 1) / 2 + 1;  // ceil((H-3)/2)+1
  const size_t total = (size_t)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_ceil_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream), in,
                     to_lay(lin), out, to_lay(lout), C, N, H, W, Ho, Wo);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_dwconv3x3(const float* in, const rtpose_layout* lin, const float* w, const float* bias, float* out,
                     const rtpose_layout* lout, int C, int N, int H, int W, int stride, void* stream) {
  if ((C % 4) || (lin->cstride % 4) || (lin->choff % 4) || (lout->cstride % 4) || (lout->choff % 4) ||
      (stride != 1 && stride != 2) || lin->ws < W + 1 || lin->hs < H + 1 || lin->lead < lin->ws + 1)
    return fail(RTPOSE_E_INVAL, "dwconv3x3: unsupported layout / stride");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const size_t total = (size_t)N * Ho * Wo * (C / 4);
  if (!total) return 0;
  if (stride == 1) {  // 4 output pixels per thread
    const size_t t4 = (size_t)N * H * ((W + 3) / 4) * (C / 4);
    hipLaunchKernelGGL(dwconv3x3_s1x4_kernel<0>, dim3(nblocks(t4, 256)), dim3(256), 0, as_stream(stream), in,
                       to_lay(lin), w, bias, out, to_lay(lout), C, N, H, W);
    RTPOSE_HIP_CHECK(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream), in,
                     to_lay(lin), w, bias, out, to_lay(lout), C, N, Ho, Wo, stride);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_layout_copy_cmap(const float* src, const rtpose_layout* lsrc, float* dst, const rtpose_layout* ldst,
                            int C, const int32_t* cmap, int N, int H, int W, void* stream) {
  if (!cmap) return fail(RTPOSE_E_INVAL, "layout_copy_cmap: cmap is NULL");
  const size_t total = (size_t)N * H * W * C;
  if (!total) return 0;
  // the source slice may be read 16 bytes at a time when it is aligned and either a
  // multiple of 4 wide or followed by readable (padding) channels inside the pixel
  if (!(lsrc->cstride % 4) && !(lsrc->choff % 4) && lsrc->choff + ((C + 3) & ~3) <= lsrc->cstride) {
    const size_t t4 = (size_t)N * H * W * ((C + 3) / 4);
    hipLaunchKernelGGL(layout_copy_cmap4_kernel, dim3(nblocks(t4, 256)), dim3(256), 0, as_stream(stream), src,
                       to_lay(lsrc), dst, to_lay(ldst), C, cmap, N, H, W);
  } else {
    hipLaunchKernelGGL(layout_copy_cmap_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream), src,
                       to_lay(lsrc), dst, to_lay(ldst), C, cmap, N, H, W);
  }
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_layout_axpby(float* dst, const rtpose_layout* ldst, const float* src_nhwc, int C, int N, int H,
                        int W, float alpha, float beta, void* stream) {
  const size_t total = (size_t)N * H * W * C;
  if (!total) return 0;
  hipLaunchKernelGGL(layout_axpby_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream), dst,
                     to_lay(ldst), src_nhwc, C, N, H, W, alpha, beta);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_preprocess_u8_batch(const rtpose_prep_image* images, int count, int mode, float* dst,
                               const rtpose_layout* ldst, int Hn, int Wn, void* stream) {
  if (!images || count <= 0 || !dst || !ldst || (mode != 0 && mode != 1) || ldst->cstride < 8 ||
      (ldst->cstride % 4) || (ldst->choff % 4) || Hn <= 0 || Wn <= 0)
    return fail(RTPOSE_E_INVAL, "preprocess_u8: bad arguments");
  for (int i = 0; i < count; ++i) {
    const rtpose_prep_image& d = images[i];
    if (!d.img_bgr || d.h0 <= 0 || d.w0 <= 0 || d.im_scale <= 0 || d.hr <= 0 || d.wr <= 0 || Hn < d.hr ||
        Wn < d.wr || d.n_index < 0)
      return fail(RTPOSE_E_INVAL, "preprocess_u8: bad image descriptor %d", i);
  }
  const size_t total = (size_t)Hn * Wn;
  for (int first = 0; first < count; first += kPrepBatch) {  // the descriptors travel as kernel arguments
    PrepBatch b;
    memset(&b, 0, sizeof(b));
    const int n = count - first < kPrepBatch ? count - first : kPrepBatch;
    for (int i = 0; i < n; ++i) {
      const rtpose_prep_image& d = images[first + i];
      PrepImg& o = b.im[i];
      o.img = static_cast<const unsigned char*>(d.img_bgr);
      o.inv_scale = 1.0 / d.im_scale;
      o.h0 = d.h0;
      o.w0 = d.w0;
      o.hr = d.hr;
      o.wr = d.wr;
      o.flip = d.flip ? 1 : 0;
      o.n = d.n_index;
    }
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(nblocks(total, 256), n), dim3(256), 0, as_stream(stream), b,
                       mode, dst, to_lay(ldst), Hn, Wn);
  }
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_preprocess_u8_flip(const unsigned char* img_bgr, int h0, int w0, double im_scale, int mode, float* dst,
                              const rtpose_layout* ldst, int n_index, int Hn, int Wn, int hr, int wr, int flip,
                              void* stream) {
  rtpose_prep_image d;
  d.img_bgr = img_bgr;
  d.h0 = h0;
  d.w0 = w0;
  d.im_scale = im_scale;
  d.hr = hr;
  d.wr = wr;
  d.flip = flip;
  d.n_index = n_index;
  return rtpose_preprocess_u8_batch(&d, 1, mode, dst, ldst, Hn, Wn, stream);
}

int rtpose_preprocess_u8(const unsigned char* img_bgr, int h0, int w0, double im_scale, int mode, float* dst,
                         const rtpose_layout* ldst, int n_index, int Hn, int Wn, int hr, int wr, void* stream) {
  return rtpose_preprocess_u8_flip(img_bgr, h0, w0, im_scale, mode, dst, ldst, n_index, Hn, Wn, hr, wr, 0, stream);
}

int rtpose_resize_bilinear_accum(const float* src, int hs, int ws, float* dst, int hd, int wd, int C, int N,
                                 float src_h_valid, float src_w_valid, float alpha, float beta, void* stream) {
  if (hs <= 0 || ws <= 0 || hd <= 0 || wd <= 0 || C <= 0 || N <= 0 || src_h_valid <= 0 || src_w_valid <= 0)
    return fail(RTPOSE_E_INVAL, "resize_bilinear_accum: bad sizes");
  const size_t total = (size_t)N * hd * wd * C;
  hipLaunchKernelGGL(resize_bilinear_accum_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream),
                     src, hs, ws, dst, hd, wd, C, N, src_h_valid / (float)hd, src_w_valid / (float)wd, alpha, beta);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_tta_accumulate(const float* heat, const rtpose_layout* lheat, const float* paf,
                          const rtpose_layout* lpaf, int B, int hs, int w_valid, float* acc_heat, float* acc_paf,
                          int hd, int wd, float src_h_valid, float src_w_valid, float alpha, float beta, int flip,
                          void* stream) {
  if (!heat || !paf || !acc_heat || !acc_paf || B <= 0 || hs <= 0 || w_valid <= 0 || hd <= 0 || wd <= 0 ||
      src_h_valid <= 0 || src_w_valid <= 0)
    return fail(RTPOSE_E_INVAL, "tta_accumulate: bad arguments");
  const size_t total = (size_t)B * hd * wd * 57;
  hipLaunchKernelGGL(tta_accumulate_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream), heat,
                     to_lay(lheat), paf, to_lay(lpaf), B, hs, w_valid, acc_heat, acc_paf, hd, wd,
                     src_h_valid / (float)hd, src_w_valid / (float)wd, alpha, beta, flip ? 1 : 0);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_flip_merge(const float* heat, const float* heat_flipped, const float* paf,
                      const float* paf_flipped, int N, int h, int w, float* heat_avg, float* paf_avg,
                      void* stream) {
  const size_t total = (size_t)N * h * w * 57;
  if (!total) return 0;
  hipLaunchKernelGGL(flip_merge_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream),
                     heat, heat_flipped, paf, paf_flipped, N, h, w, heat_avg, paf_avg);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}


int rtpose_nchw_to_layout_bf16(const float* src_nchw, void* dst, const rtpose_layout* ldst, int C, int cpad,
                               int N, int H, int W, void* stream) {
  if ((cpad % 8) || (ldst->cstride % 8) || (ldst->choff % 8) || cpad < C)
    return fail(RTPOSE_E_INVAL, "nchw_to_layout_bf16: slice must be 16-byte aligned, cpad >= C");
  const size_t total = (size_t)N * H * W * (cpad / 8);
  if (!total) return 0;
  hipLaunchKernelGGL(to_bf16_layout_kernel<true>, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream),
                     src_nchw, Lay{}, static_cast<unsigned short*>(dst), to_lay(ldst), C, cpad, N, H, W);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_layout_f32_to_bf16(const float* src, const rtpose_layout* lsrc, void* dst,
                              const rtpose_layout* ldst, int C, int cpad, int N, int H, int W, void* stream) {
  if ((cpad % 8) || (ldst->cstride % 8) || (ldst->choff % 8) || cpad < C)
    return fail(RTPOSE_E_INVAL, "layout_f32_to_bf16: slice must be 16-byte aligned, cpad >= C");
  const size_t total = (size_t)N * H * W * (cpad / 8);
  if (!total) return 0;
  hipLaunchKernelGGL(to_bf16_layout_kernel<false>, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream),
                     src, to_lay(lsrc), static_cast<unsigned short*>(dst), to_lay(ldst), C, cpad, N, H, W);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_layout_bf16_to_f32(const void* src, const rtpose_layout* lsrc, float* dst, const rtpose_layout* ldst,
                              int C, int N, int H, int W, void* stream) {
  const size_t total = (size_t)N * H * W * C;
  if (!total) return 0;
  hipLaunchKernelGGL(layout_bf16_to_f32_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream),
                     static_cast<const unsigned short*>(src), to_lay(lsrc), dst, to_lay(ldst), C, N, H, W);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}


int rtpose_nchw_to_layout_split(const float* src_nchw, void* dst, const rtpose_layout* ldst, int C, int cpad,
                                int N, int H, int W, void* stream) {
  if ((cpad % 8) || (ldst->cstride % 16) || (ldst->choff % 16) || cpad < C)
    return fail(RTPOSE_E_INVAL, "nchw_to_layout_split: slice must be 32-byte aligned, cpad >= C");
  const size_t total = (size_t)N * H * W * (cpad / 8);
  if (!total) return 0;
  hipLaunchKernelGGL(to_split_layout_kernel<true>, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream),
                     src_nchw, Lay{}, static_cast<unsigned short*>(dst), to_lay(ldst), C, cpad, N, H, W);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_layout_f32_to_split(const float* src, const rtpose_layout* lsrc, void* dst,
                               const rtpose_layout* ldst, int C, int cpad, int N, int H, int W, void* stream) {
  if ((cpad % 8) || (ldst->cstride % 16) || (ldst->choff % 16) || cpad < C)
    return fail(RTPOSE_E_INVAL, "layout_f32_to_split: slice must be 32-byte aligned, cpad >= C");
  const size_t total = (size_t)N * H * W * (cpad / 8);
  if (!total) return 0;
  hipLaunchKernelGGL(to_split_layout_kernel<false>, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream),
                     src, to_lay(lsrc), static_cast<unsigned short*>(dst), to_lay(ldst), C, cpad, N, H, W);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_layout_split_to_f32(const void* src, const rtpose_layout* lsrc, float* dst, const rtpose_layout* ldst,
                               int C, int N, int H, int W, void* stream) {
  if (lsrc->choff & 1) return fail(RTPOSE_E_INVAL, "layout_split_to_f32: choff counts elements (2 per channel)");
  const size_t total = (size_t)N * H * W * C;
  if (!total) return 0;
  hipLaunchKernelGGL(layout_split_to_f32_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream),
                     static_cast<const unsigned short*>(src), to_lay(lsrc), dst, to_lay(ldst), C, N, H, W);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}


int rtpose_maxpool3x3s2_ceil_bf16(const void* in, const rtpose_layout* lin, void* out, const rtpose_layout* lout,
                                  int C, int N, int H, int W, void* stream) {
  if ((C % 8) || (lin->cstride % 8) || (lin->choff % 8) || (lout->cstride % 8) || (lout->choff % 8) || H < 3 || W < 3)
    return fail(RTPOSE_E_INVAL, "maxpool3x3s2_bf16: channel slices must be 16-byte aligned, H,W >= 3");
  const int Ho = (H - 3 + 1) / 2 + 1, Wo = (W - 3 + 1) / 2 + 1;
  const size_t total = (size_t)N * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(maxpool3x3s2_ceil_bf16_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream),
                     static_cast<const unsigned short*>(in), to_lay(lin), static_cast<unsigned short*>(out),
                     to_lay(lout), C, N, H, W, Ho, Wo);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_dwconv3x3_bf16(const void* in, const rtpose_layout* lin, const float* w, const float* bias, void* out,
                          const rtpose_layout* lout, int C, int N, int H, int W, int stride, void* stream) {
  if ((C % 8) || (lin->cstride % 8) || (lin->choff % 8) || (lout->cstride % 8) || (lout->choff % 8) ||
      (stride != 1 && stride != 2) || lin->ws < W + 1 || lin->hs < H + 1 || lin->lead < lin->ws + 1)
    return fail(RTPOSE_E_INVAL, "dwconv3x3_bf16: unsupported layout / stride");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const size_t total = (size_t)N * Ho * Wo * (C / 8);
  if (!total) return 0;
  // (the 4-pixels-per-thread form measured 2.6x SLOWER here - 32 accumulators x 8 channels per
  //  thread - so the bf16 plans keep one pixel per thread: 1.57 ms vs 4.03 ms over the 19 layers)
  hipLaunchKernelGGL(dwconv3x3_bf16_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream),
                     static_cast<const unsigned short*>(in), to_lay(lin), w, bias, static_cast<unsigned short*>(out),
                     to_lay(lout), C, N, Ho, Wo, stride);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

int rtpose_layout_copy_cmap_bf16(const void* src, const rtpose_layout* lsrc, void* dst, const rtpose_layout* ldst,
                                 int C, const int32_t* cmap, int N, int H, int W, void* stream) {
  if (!cmap || (lsrc->cstride % 8) || (lsrc->choff % 8))
    return fail(RTPOSE_E_INVAL, "layout_copy_cmap_bf16: cmap is NULL or the source slice is not 16-byte aligned");
  const size_t total = (size_t)N * H * W * ((C + 7) / 8);
  if (!total) return 0;
  hipLaunchKernelGGL(layout_copy_cmap_bf16_kernel, dim3(nblocks(total, 256)), dim3(256), 0, as_stream(stream),
                     static_cast<const unsigned short*>(src), to_lay(lsrc), static_cast<unsigned short*>(dst),
                     to_lay(ldst), C, cmap, N, H, W);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
