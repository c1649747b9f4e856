// Batched pose decoding on the GPU — replaces the numpy/scipy/cv2 control flow
// of lib/utils/paf_to_pose.py (NMS :67-145, find_peaks :25-38) and the serial
// C++ of lib/pafprocess/pafprocess.cpp (process_paf :22-194) with three
// wavefront-level kernels, one launch each for a whole batch of images:
//
//   nms_refine_kernel      grid (18 parts, N)   peak test + ordered compaction +
//                                               8x bicubic patch refine/arg-max
//   limb_assign_kernel     grid (19 limbs, N)   10-sample PAF line integral for
//                                               every (a,b) pair + greedy 1:1
//   group_kernel           grid (N)             subset merge + prune (one wave)
//
// Integer outputs (peak coordinates, ids, part->peak assignments) are bit-exact
// to the reference; float scores reproduce its operation order (this file is
// compiled with -ffp-contract=off; hipcc's default correctly rounded fp32
// divide/sqrt is relied upon).  All HBM reads are of the low-resolution maps:
// the x8 nearest-neighbour up-sampling of paf_to_pose.py:382-385 is an index
// computation (floor(x * 1/8)), never materialised.
#include <hip/hip_runtime.h>

#include "common.h"
#include "decode.h"

namespace rtpose {

// pafprocess.h:16-24
__constant__ int kPairNet[19][2] = {{12, 13}, {20, 21}, {14, 15}, {16, 17}, {22, 23}, {24, 25}, {0, 1},
                                    {2, 3},   {4, 5},   {6, 7},   {8, 9},   {10, 11}, {28, 29}, {30, 31},
                                    {34, 35}, {32, 33}, {36, 37}, {18, 19}, {26, 27}};
__constant__ int kPairs[19][2] = {{1, 2}, {1, 5},   {2, 3},   {3, 4},   {5, 6},   {6, 7},   {1, 8},
                                  {8, 9}, {9, 10},  {1, 11},  {11, 12}, {12, 13}, {1, 0},   {0, 14},
                                  {14, 16}, {0, 15}, {15, 17}, {2, 16},  {5, 17}};

struct MapView {
  const float* base;
  int cstride, choff, ws, hs, lead;
};

__device__ __forceinline__ float map_at(const MapView& m, int n, int y, int x, int c) {
  return m.base[((size_t)m.lead + (size_t)(n * m.hs + y) * m.ws + x) * m.cstride + m.choff + c];
}

// ------------------------------------------------------------------------------
// 1. NMS + refine
// ------------------------------------------------------------------------------
constexpr int kMaxUp = 16;               // largest supported up-sampling factor
constexpr int kMaxDst = 5 * kMaxUp;      // widest up-sampled patch

// OpenCV interpolateCubic (A = -0.75f), float arithmetic, no contraction.
__device__ __forceinline__ void cubic_coeffs(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

// find_peaks (paf_to_pose.py:25-38) for one (image, part): 4-neighbour maximum, > thr, peaks
// compacted in row-major order into s_px / s_py (the order defines the peak ids).  Whole block.
constexpr int kPeakBatch = 5;  // 256-pixel sweeps whose loads are in flight together (a 46 x 46 map is 8.3 sweeps: 2 batches)
__device__ __forceinline__ int find_peaks_block(const MapView& heat, int n, int part, int h, int w, float thr,
                                                int pcap, int (*s_wcount)[4], int* s_px, int* s_py, int32_t* res) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int base = 0;
  const int npix = h * w;
  // Round 6: the CENTRE values of kPeakBatch sweeps are requested together (one global round trip per batch instead of one
  // per sweep), then the four neighbours of the few pixels above the threshold - still only of those: fetching them for every
  // pixel cost five times the loads and made the kernel slower (33 -> 43 us at batch 32) - again all sweeps' requests before
  // the first value is used; one pair of barriers per batch instead of per sweep.  The tests are the sweep-by-sweep form's.
  for (int start = 0; start < npix; start += 256 * kPeakBatch) {
    float v[kPeakBatch], vu[kPeakBatch], vd[kPeakBatch], vl[kPeakBatch], vr[kPeakBatch];
    int xs[kPeakBatch], ys[kPeakBatch];
#pragma unroll
    for (int b = 0; b < kPeakBatch; ++b) {
      const int idx = min(start + 256 * b + tid, npix - 1);
      const int y = idx / w, x = idx - y * w;
      ys[b] = y;
      xs[b] = x;
      v[b] = map_at(heat, n, y, x, part);
    }
#pragma unroll
    for (int b = 0; b < kPeakBatch; ++b) {
      vu[b] = vd[b] = vl[b] = vr[b] = 0.f;
      if (start + 256 * b + tid < npix && v[b] > thr) {  // (border neighbours: the pixel itself, ignored below)
        const int y = ys[b], x = xs[b];
        vu[b] = map_at(heat, n, max(y - 1, 0), x, part);
        vd[b] = map_at(heat, n, min(y + 1, h - 1), x, part);
        vl[b] = map_at(heat, n, y, max(x - 1, 0), part);
        vr[b] = map_at(heat, n, y, min(x + 1, w - 1), part);
      }
    }
    bool pk[kPeakBatch];
    unsigned long long mask[kPeakBatch];
#pragma unroll
    for (int b = 0; b < kPeakBatch; ++b) {
      const int x = xs[b], y = ys[b];
      bool p = start + 256 * b + tid < npix && v[b] > thr;
      if (p && y > 0) p = v[b] >= vu[b];
      if (p && y + 1 < h) p = v[b] >= vd[b];
      if (p && x > 0) p = v[b] >= vl[b];
      if (p && x + 1 < w) p = v[b] >= vr[b];
      pk[b] = p;
      mask[b] = __ballot(p);
      if (lane == 0) s_wcount[b][wave] = __popcll(mask[b]);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < kPeakBatch; ++b) {
      int off = base;
      for (int k = 0; k < wave; ++k) off += s_wcount[b][k];
      if (pk[b]) {
        const int pos = off + __popcll(mask[b] & ((1ull << lane) - 1ull));
        if (pos < pcap) {
          s_px[pos] = xs[b];
          s_py[pos] = ys[b];
        }
      }
      base += s_wcount[b][0] + s_wcount[b][1] + s_wcount[b][2] + s_wcount[b][3];
    }
    __syncthreads();
  }
  const int count = min(base, pcap);
  if (tid == 0) {
    res[kResPartCount + part] = count;
    if (base > pcap) atomicOr(&res[kResHeader + 2], kOverflowPeaks);
  }
  return count;
}

__global__ __launch_bounds__(256) void nms_refine_kernel(MapView heat, int h, int w, int up,
                                                         double inv_up, float thr, int pcap,
                                                         int32_t* __restrict__ result,
                                                         int result_words) {
  const int part = blockIdx.x, n = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int32_t* res = result + (size_t)n * result_words;

  __shared__ int s_sx[kMaxDst];
  __shared__ float s_alpha[kMaxDst][4];
  __shared__ int s_wcount[kPeakBatch][4];
  __shared__ int s_px[kDecodeMaxPeaks], s_py[kDecodeMaxPeaks];
  __shared__ float s_patch[4][25];
  __shared__ float s_hbuf[4][5 * kMaxDst];

  // per-destination-index source offset and cubic weights (cv2.resize INTER_CUBIC:
  // fx = (float)((dx+0.5)*scale - 0.5); sx = floor(fx); fx -= sx)
  if (tid < 5 * up) {
    // (double)(tid + 0.5) * inv_up - 0.5 rounded to float: exact in fp32 when up is a power of two (no double-precision
    // instruction then: see limb_assign_kernel)
    float fx = (up & (up - 1)) == 0 ? ((float)tid + 0.5f) * (float)inv_up - 0.5f
                                    : (float)(((double)tid + 0.5) * inv_up - 0.5);
    const int sx = (int)floorf(fx);
    fx -= (float)sx;
    float c[4];
    cubic_coeffs(fx, c);
    s_sx[tid] = sx;
    s_alpha[tid][0] = c[0];
    s_alpha[tid][1] = c[1];
    s_alpha[tid][2] = c[2];
    s_alpha[tid][3] = c[3];
  }

  const int count = find_peaks_block(heat, n, part, h, w, thr, pcap, s_wcount, s_px, s_py, res);

  // ---- refine (paf_to_pose.py:106-142): one wave per peak
  rtpose_peak* peaks = reinterpret_cast<rtpose_peak*>(res + kResPeaks) + (size_t)part * pcap;
  for (int i = wave; i < count; i += 4) {
    const int px = s_px[i], py = s_py[i];
    const int x_min = max(0, px - 2), y_min = max(0, py - 2);
    const int x_max = min(w - 1, px + 2), y_max = min(h - 1, py + 2);
    const int pw = x_max - x_min + 1, ph = y_max - y_min + 1;
    const int dw = pw * up, dh = ph * up;
    if (lane < pw * ph) {
      const int r = lane / pw, c = lane - r * pw;
      s_patch[wave][lane] = map_at(heat, n, y_min + r, x_min + c, part);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // horizontal pass: hbuf[r][dx] = sum_j patch[r][clamp(sx-1+j)] * alpha[dx][j]
    for (int e = lane; e < ph * dw; e += 64) {
      const int r = e / dw, dx = e - r * dw;
      const int sx = s_sx[dx];
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sxj = min(max(sx - 1 + j, 0), pw - 1);
        v = v + s_patch[wave][r * pw + sxj] * s_alpha[dx][j];
      }
      s_hbuf[wave][r * dw + dx] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // vertical pass + running arg-max (first maximum in row-major order)
    float best = -INFINITY;
    int best_idx = 0x7fffffff;
    for (int e = lane; e < dh * dw; e += 64) {
      const int dy = e / dw, dx = e - dy * dw;
      const int sy = s_sx[dy];
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int syj = min(max(sy - 1 + j, 0), ph - 1);
        const float t = s_hbuf[wave][syj * dw + dx] * s_alpha[dy][j];
        v = (j == 0) ? t : v + t;
      }
      if (v > best) {
        best = v;
        best_idx = e;
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int oi = __shfl_xor(best_idx, o);
      if (ob > best || (ob == best && oi < best_idx)) {
        best = ob;
        best_idx = oi;
      }
    }
    if (lane == 0) {
      const int dy = best_idx / dw, dx = best_idx - dy * dw;
      rtpose_peak p;
      p.x = x_min * up + dx;  // paf_to_pose.py:129-141 collapses to this
      p.y = y_min * up + dy;
      p.score = best;
      p.id = i;  // rebased to the running counter by the prefix kernel
      peaks[i] = p;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// The two optional branches of NMS (paf_to_pose.py:67): bool_gaussian_filt=True smooths the
// up-sampled patch with scipy.ndimage.gaussian_filter(sigma=3) before the arg-max (:121-122);
// bool_refine_center=False skips the patch altogether (:135-139).  Neither is used by the
// reference's callers, so this kernel favours clarity: the whole block works on one peak at a time.
// gaussian_filter = correlate1d along axis 0 (rows), then axis 1, mode 'reflect', each pass
// accumulated in double in scipy's symmetric-kernel order and stored as float32
// (ndimage/src/ni_filters.c NI_Correlate1D).
constexpr int kGaussR = 12;  // int(truncate 4.0 * sigma 3 + 0.5)
struct GaussW {
  double w[2 * kGaussR + 1];
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
  while (i < 0 || i >= n) {
    if (i < 0) i = -i - 1;
    if (i >= n) i = 2 * n - 1 - i;
  }
  return i;
}

__global__ __launch_bounds__(256) void nms_refine_opt_kernel(MapView heat, int h, int w, int up, double inv_up,
                                                             float thr, int pcap, int32_t* __restrict__ result,
                                                             int result_words, int flags, GaussW gw) {
  const int part = blockIdx.x, n = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int32_t* res = result + (size_t)n * result_words;
  extern __shared__ float s_dyn[];  // [2][(5 up)^2]: the up-sampled patch and the first Gaussian pass
  __shared__ int s_sx[kMaxDst];
  __shared__ float s_alpha[kMaxDst][4];
  __shared__ int s_wcount[kPeakBatch][4];
  __shared__ int s_px[kDecodeMaxPeaks], s_py[kDecodeMaxPeaks];
  __shared__ float s_patch[25];
  __shared__ float s_hbuf[5 * kMaxDst];
  __shared__ float s_best[4];
  __shared__ int s_besti[4];
  float* s_up = s_dyn;
  float* s_tmp = s_dyn + 25 * up * up;

  if (tid < 5 * up) {
    // (double)(tid + 0.5) * inv_up - 0.5 rounded to float: exact in fp32 when up is a power of two (no double-precision
    // instruction then: see limb_assign_kernel)
    float fx = (up & (up - 1)) == 0 ? ((float)tid + 0.5f) * (float)inv_up - 0.5f
                                    : (float)(((double)tid + 0.5) * inv_up - 0.5);
    const int sx = (int)floorf(fx);
    fx -= (float)sx;
    float c[4];
    cubic_coeffs(fx, c);
    s_sx[tid] = sx;
    s_alpha[tid][0] = c[0];
    s_alpha[tid][1] = c[1];
    s_alpha[tid][2] = c[2];
    s_alpha[tid][3] = c[3];
  }
  const int count = find_peaks_block(heat, n, part, h, w, thr, pcap, s_wcount, s_px, s_py, res);
  rtpose_peak* peaks = reinterpret_cast<rtpose_peak*>(res + kResPeaks) + (size_t)part * pcap;

  if (flags & RTPOSE_NMS_NO_REFINE) {
    // :135-139 + compute_resized_coords (:41-64): the peak's cell centre (c + 0.5) * up - 0.5 in
    // float64, score = the low-res map value.  Stored truncated (what process_paf's (int) cast of
    // the joint_list column makes of it, pafprocess.cpp:28-29); NMS() on the host re-derives the float.
    for (int i = tid; i < count; i += 256) {
      rtpose_peak p;
      p.x = (int)(((double)s_px[i] + 0.5) * (double)up - 0.5);
      p.y = (int)(((double)s_py[i] + 0.5) * (double)up - 0.5);
      p.score = map_at(heat, n, s_py[i], s_px[i], part);
      p.id = i;
      peaks[i] = p;
    }
    return;
  }

  for (int i = 0; i < count; ++i) {
    const int px = s_px[i], py = s_py[i];
    const int x_min = max(0, px - 2), y_min = max(0, py - 2);
    const int x_max = min(w - 1, px + 2), y_max = min(h - 1, py + 2);
    const int pw = x_max - x_min + 1, ph = y_max - y_min + 1;
    const int dw = pw * up, dh = ph * up;
    if (tid < pw * ph) {
      const int r = tid / pw, c = tid - r * pw;
      s_patch[tid] = map_at(heat, n, y_min + r, x_min + c, part);
    }
    __syncthreads();
    for (int e = tid; e < ph * dw; e += 256) {  // cv2.resize INTER_CUBIC, horizontal pass
      const int r = e / dw, dx = e - r * dw;
      const int sx = s_sx[dx];
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sxj = min(max(sx - 1 + j, 0), pw - 1);
        v = v + s_patch[r * pw + sxj] * s_alpha[dx][j];
      }
      s_hbuf[r * dw + dx] = v;
    }
    __syncthreads();
    for (int e = tid; e < dh * dw; e += 256) {  // vertical pass (same expression order as nms_refine_kernel)
      const int dy = e / dw, dx = e - dy * dw;
      const int sy = s_sx[dy];
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int syj = min(max(sy - 1 + j, 0), ph - 1);
        const float t = s_hbuf[syj * dw + dx] * s_alpha[dy][j];
        v = (j == 0) ? t : v + t;
      }
      s_up[e] = v;
    }
    __syncthreads();
    if (flags & RTPOSE_NMS_GAUSSIAN) {
      const double* wc = gw.w + kGaussR;
      for (int e = tid; e < dh * dw; e += 256) {  // axis 0
        const int y = e / dw, x = e - y * dw;
        double acc = (double)s_up[e] * wc[0];
        for (int j = -kGaussR; j < 0; ++j)
          acc += ((double)s_up[reflect_idx(y + j, dh) * dw + x] + (double)s_up[reflect_idx(y - j, dh) * dw + x]) * wc[j];
        s_tmp[e] = (float)acc;
      }
      __syncthreads();
      for (int e = tid; e < dh * dw; e += 256) {  // axis 1
        const int y = e / dw, x = e - y * dw;
        double acc = (double)s_tmp[e] * wc[0];
        for (int j = -kGaussR; j < 0; ++j)
          acc += ((double)s_tmp[y * dw + reflect_idx(x + j, dw)] + (double)s_tmp[y * dw + reflect_idx(x - j, dw)]) * wc[j];
        s_up[e] = (float)acc;
      }
      __syncthreads();
    }
    float best = -INFINITY;  // first maximum in row-major order (:125-126)
    int best_idx = 0x7fffffff;
    for (int e = tid; e < dh * dw; e += 256) {
      const float v = s_up[e];
      if (v > best) {
        best = v;
        best_idx = e;
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int oi = __shfl_xor(best_idx, o);
      if (ob > best || (ob == best && oi < best_idx)) {
        best = ob;
        best_idx = oi;
      }
    }
    if (lane == 0) {
      s_best[wave] = best;
      s_besti[wave] = best_idx;
    }
    __syncthreads();
    if (tid == 0) {
      for (int k = 1; k < 4; ++k)
        if (s_best[k] > best || (s_best[k] == best && s_besti[k] < best_idx)) {
          best = s_best[k];
          best_idx = s_besti[k];
        }
      const int dy = best_idx / dw, dx = best_idx - dy * dw;
      rtpose_peak p;
      p.x = x_min * up + dx;
      p.y = y_min * up + dy;
      p.score = best;
      p.id = i;
      peaks[i] = p;
    }
    __syncthreads();
  }
}

// ids = running counter over parts then peaks (paf_to_pose.py:141-142); also
// totals in the header.  One wave per image.
__global__ void peak_prefix_kernel(int pcap, int32_t* __restrict__ result, int result_words,
                                   int nparts) {
  const int n = blockIdx.x, lane = threadIdx.x;
  int32_t* res = result + (size_t)n * result_words;
  __shared__ int s_start[RTPOSE_NUM_PART + 1];
  if (lane == 0) {
    int acc = 0;
    for (int p = 0; p < RTPOSE_NUM_PART; ++p) {
      s_start[p] = acc;
      acc += (p < nparts) ? res[kResPartCount + p] : 0;
    }
    s_start[RTPOSE_NUM_PART] = acc;
    res[kResHeader + 0] = acc;
  }
  __syncthreads();
  rtpose_peak* peaks = reinterpret_cast<rtpose_peak*>(res + kResPeaks);
  for (int p = 0; p < nparts; ++p) {
    const int cnt = res[kResPartCount + p];
    for (int i = lane; i < cnt; i += 64) peaks[(size_t)p * pcap + i].id = s_start[p] + i;
  }
}


// ------------------------------------------------------------------------------
// std::sort(candidates.begin(), candidates.end(), comp_candidate) (pafprocess.cpp:97, :244-246)
// as libstdc++ (GCC 11 bits/stl_algo.h - what the reference links against when built with this
// image's g++) executes it, run by ONE lane.  std::sort is not stable: when two candidates of a
// limb score exactly the same (two peaks refined to the same pixel), which one the greedy scan
// meets first is decided by the introsort's moves - median-of-3 quicksort down to 16-element
// runs under a 2 floor(log2 n) depth limit (heap sort beyond it), then one insertion pass.  The
// product is "identical to pafprocess.cpp built with g++ 11", so the rare limb with such a tie
// replays those moves on its candidate list L (entries in the reference's push order, each the
// score's bits above the pair index a * nB + b); comp(a, b) = a.score > b.score.  The list lives
// in LDS when it has at most kTieLdsCands entries (a dependent access every ~100 cycles instead
// of every ~500-2000), else in the workspace.
// ------------------------------------------------------------------------------
struct SortReplay {
  unsigned long long* L;  // entry = score bits << 32 | (a * nB + b); LDS when the list fits, else the workspace
  int* stk;               // LDS, 3 ints per pending range
  // scores are positive finite floats: their bit patterns order like their values, equal iff the floats are equal
  static __device__ __forceinline__ bool gt(unsigned long long a, unsigned long long b) {
    return (unsigned)(a >> 32) > (unsigned)(b >> 32);
  }
  __device__ __forceinline__ void swap(int i, int j) {
    const unsigned long long t = L[i];
    L[i] = L[j];
    L[j] = t;
  }
  __device__ void unguarded_linear_insert(int last) {
    const unsigned long long val = L[last];
    int next = last - 1;
    unsigned long long nv = L[next];
    while (gt(val, nv)) {
      L[last] = nv;
      last = next;
      --next;
      nv = L[next];
    }
    L[last] = val;
  }
  __device__ void insertion_sort(int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
      const unsigned long long val = L[i];
      if (gt(val, L[first])) {
        for (int k = i; k > first; --k) L[k] = L[k - 1];  // move_backward(first, i, i + 1)
        L[first] = val;
      } else {
        unguarded_linear_insert(i);
      }
    }
  }
  __device__ void push_heap(int first, int hole, int top, unsigned long long value) {
    int parent = (hole - 1) / 2;
    while (hole > top && gt(L[first + parent], value)) {
      L[first + hole] = L[first + parent];
      hole = parent;
      parent = (hole - 1) / 2;
    }
    L[first + hole] = value;
  }
  __device__ void adjust_heap(int first, int hole, int len, unsigned long long value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (gt(L[first + child], L[first + child - 1])) child--;
      L[first + hole] = L[first + child];
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      L[first + hole] = L[first + child - 1];
      hole = child - 1;
    }
    push_heap(first, hole, top, value);
  }
  __device__ void heap_sort(int first, int last) {  // __partial_sort(first, last, last)
    const int len = last - first;
    if (len >= 2)
      for (int parent = (len - 2) / 2;; --parent) {  // __make_heap
        adjust_heap(first, parent, len, L[first + parent]);
        if (parent == 0) break;
      }
    while (last - first > 1) {  // __sort_heap
      --last;
      const unsigned long long value = L[last];
      L[last] = L[first];
      adjust_heap(first, 0, last - first, value);
    }
  }
  __device__ void move_median_to_first(int result, int a, int b, int c) {
    const unsigned long long va = L[a], vb = L[b], vc = L[c];
    if (gt(va, vb)) {
      if (gt(vb, vc)) swap(result, b);
      else if (gt(va, vc)) swap(result, c);
      else swap(result, a);
    } else if (gt(va, vc)) swap(result, a);
    else if (gt(vb, vc)) swap(result, c);
    else swap(result, b);
  }
  __device__ int unguarded_partition(int first, int last, int pivot) {
    const unsigned long long pv = L[pivot];  // (the pivot sits at `first - 1`, outside the range being swapped)
    for (;;) {
      while (gt(L[first], pv)) ++first;
      --last;
      while (gt(pv, L[last])) --last;
      if (!(first < last)) return first;
      swap(first, last);
      ++first;
    }
  }
  // __introsort_loop: the recursion on [cut, last) becomes a pending range (the ranges are
  // disjoint, so the order they are finished in does not change a single move)
  __device__ void introsort(int n, int depth_limit) {
    int sp = 0;
    stk[0] = 0;
    stk[1] = n;
    stk[2] = depth_limit;
    sp = 1;
    while (sp > 0) {
      --sp;
      int first = stk[3 * sp], last = stk[3 * sp + 1], depth = stk[3 * sp + 2];
      while (last - first > 16) {
        if (depth == 0) {
          heap_sort(first, last);
          break;
        }
        --depth;
        const int mid = first + (last - first) / 2;
        move_median_to_first(first, first + 1, mid, last - 1);
        const int cut = unguarded_partition(first + 1, last, first);
        stk[3 * sp] = cut;
        stk[3 * sp + 1] = last;
        stk[3 * sp + 2] = depth;
        ++sp;
        last = cut;
      }
    }
  }
  __device__ void sort_desc(int n) {
    if (n == 0) return;
    int lg = 0;
    while ((n >> (lg + 1)) > 0) ++lg;
    introsort(n, 2 * lg);
    if (n > 16) {
      insertion_sort(0, 16);
      for (int i = 16; i != n; ++i) unguarded_linear_insert(i);
    } else {
      insertion_sort(0, n);
    }
  }
};
constexpr int kSortStack = 3 * 64;  // pending ranges <= the depth limit 2 floor(log2 n) <= 40

// ------------------------------------------------------------------------------
// 2. PAF scoring + greedy assignment (pafprocess.cpp:46-124, :220-246)
// ------------------------------------------------------------------------------

// Round 5: (a) SCORES_IN_LDS is a template parameter - the score matrix is addressed with LDS instructions (or global
// ones), not through a generic pointer (flat_load / flat_store, the aperture decided per access); (b) NO double-precision
// instruction in the per-sample loop: the reference's (int)(v + 0.5) on a double and floor(l / up) are computed exactly
// in integers / fp32 (v >= 0 and up a power of two, the only case the configs use; any other `up` keeps the doubles),
// and the length penalty needs its double division only for limbs longer than half the image.
// Round 6: (c) A32 - a sample's two map values through the SGPR-base form of global_load_dword with one 32-bit byte offset
// (two 24-bit multiplies): no 64-bit integer VALU instruction between the peak loads and the last map load; (d) this file is
// compiled WITHOUT clang's vectorisers (csrc/Makefile).  With the decoder on a second stream beside the bf16 forward's first
// launches (pipeline.SideDecoder), the vectorised build of this loop - v_pk_mul_f32 (vx, vy) x (px, py) on the loaded map
// values and op_sel-swizzled v_pk_add_f32 behind them - returned, in ~1 of 150 launches, ONE candidate score computed from a
// wrong sample, always in lanes 48..63 (most often the top active lane), with maps and peaks bit-identical before and after,
// never in the serial flow, never on CUs the forward does not use.  Eight discriminating builds and three stand-alone
// victims: DESIGN.md 3.3, profiles/r06_decoder_beside_forward.txt.  (a) - (c) were each tried as the cure and are not it
// (they stay: fewer instructions); (d) is.  The RTPOSE_EXP_LIMB_* blocks below are those builds' switches
// (tools/build_dev.sh -DRTPOSE_EXP_LIMB_...; a production build refuses them).
#if !defined(RTPOSE_DEV_BUILD) && (defined(RTPOSE_EXP_LIMB_WAIT0) || defined(RTPOSE_EXP_LIMB_KEEP) || \
                                   defined(RTPOSE_EXP_LIMB_ASM_COORD) || defined(RTPOSE_EXP_LIMB_ASM_DOT) || \
                                   defined(RTPOSE_EXP_GROUP_TIMELINE))
#error "RTPOSE_EXP_LIMB_* are developer-build experiments (tools/build_dev.sh)"
#endif
template <bool SCORES_IN_LDS, bool UP_POW2, bool A32>
__global__ __launch_bounds__(256) void limb_assign_kernel(MapView paf, int h, int w, double inv_up, int up_shift,
                                                          int h1, int pcap,
                                                          const int32_t* __restrict__ result,
                                                          int result_words, int32_t* __restrict__ conn,
                                                          int conn_words, float* __restrict__ score_ws,
                                                          unsigned long long* __restrict__ tie_ws) {
  const int pair_id = blockIdx.x, n = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t* res = result + (size_t)n * result_words;
  int32_t* cn = conn + (size_t)n * conn_words + (size_t)pair_id * (1 + 3 * pcap);

  extern __shared__ __attribute__((aligned(16))) float s_score_lds[];  // [pcap * pcap] candidate scores, 0 = none (when they fit), then the tie list
  __shared__ unsigned char s_usedA[kDecodeMaxPeaks], s_usedB[kDecodeMaxPeaks];
  __shared__ float s_wbest[4];
  __shared__ int s_widx[4], s_wcnt[4];
  __shared__ int s_nconn;
  __shared__ int s_stack[kSortStack];
  // the limb's connections (a, b, score bits) as they are accepted; written to the workspace in one parallel sweep at the end
  // (round 6: thread 0 stored each one to global memory inside the greedy loop, and the store had to land before the loop's
  // barrier let the next step start - a global round trip per connection)
  __shared__ int s_conn[3 * kDecodeMaxPeaks];

  const int partA = kPairs[pair_id][0], partB = kPairs[pair_id][1];
  const int chx = kPairNet[pair_id][0], chy = kPairNet[pair_id][1];
  const int nA = res[kResPartCount + partA], nB = res[kResPartCount + partB];
  if (tid == 0) s_nconn = 0;
  if (nA == 0 || nB == 0) {
    if (tid == 0) cn[0] = 0;
    return;
  }
  const rtpose_peak* pA = reinterpret_cast<const rtpose_peak*>(res + kResPeaks) + (size_t)partA * pcap;
  const rtpose_peak* pB = reinterpret_cast<const rtpose_peak*>(res + kResPeaks) + (size_t)partB * pcap;
  for (int i = tid; i < kDecodeMaxPeaks; i += 256) {
    s_usedA[i] = 0;
    s_usedB[i] = 0;
  }

  const int npairs = nA * nB;
  // A32: the image's two PAF channel planes as wave-uniform bases (SGPR pairs) + ONE 32-bit byte offset per sample
  // (global_load_dword v, v_off, s[base:base+1]); the host picks it when every offset of an image fits 31 bits
  const char* const img_x =
      reinterpret_cast<const char*>(paf.base + ((size_t)paf.lead + (size_t)n * paf.hs * paf.ws) * paf.cstride + paf.choff + chx);
  const char* const img_y = img_x + (ptrdiff_t)(chy - chx) * (ptrdiff_t)sizeof(float);
  const unsigned pix_bytes = (unsigned)paf.cstride * (unsigned)sizeof(float);
  // the score matrix lives in LDS unless the tables were grown past what LDS holds
  // (junk maps with hundreds of peaks per part): then in the global workspace
  float* const score_g = score_ws + ((size_t)n * RTPOSE_NUM_LIMB + pair_id) * pcap * pcap;
  auto s_score = [&](int p) -> float& {
    if constexpr (SCORES_IN_LDS)
      return s_score_lds[p];
    else
      return score_g[p];
  };
  for (int p = tid; p < npairs; p += 256) {
    const int a = p / nB, b = p - a * nB;
    const rtpose_peak A = pA[a], B = pB[b];
    float cand = 0.f;
    float vx = (float)(B.x - A.x), vy = (float)(B.y - A.y);
    const float norm = sqrtf(vx * vx + vy * vy);
    if (norm > 0.f) {  // (double)norm < 1e-12 of the reference: norm is the root of a sum of integer squares, 0 or >= 1
      vx = vx / norm;
      vy = vy / norm;
      const float step_x = (float)(B.x - A.x) / 10.f;
      const float step_y = (float)(B.y - A.y) / 10.f;
      float scores = 0.f;
      int crit1 = 0;
#ifdef RTPOSE_EXP_LIMB_WAIT0  // experiment: every map load has landed before the first value is consumed
      float pxa[10], pya[10];
#endif
#ifdef RTPOSE_EXP_LIMB_KEEP  // experiment: the loads' offset registers stay live until every load has returned
      unsigned keep[10];
#endif
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        // lx = (int)(v + 0.5) evaluated in double (pafprocess.cpp:232-233): v >= 0 is a float, so v + 0.5 is exact there
        // and the cast is floor(v + 0.5) = trunc(v) + (frac(v) >= 0.5), frac exact in fp32
#ifdef RTPOSE_EXP_LIMB_ASM_COORD  // experiment: the y coordinate through opaque scalar instructions - the
        const float fx = (float)A.x + (float)i * step_x;   // vectoriser cannot pair it with x
        float fy;
        {
          float ty;
          asm("v_mul_f32 %0, %1, %2" : "=v"(ty) : "v"((float)i), "v"(step_y));
          asm("v_add_f32 %0, %1, %2" : "=v"(fy) : "v"((float)A.y), "v"(ty));
        }
#else
        const float fx = (float)A.x + (float)i * step_x, fy = (float)A.y + (float)i * step_y;
#endif
        int lx = (int)fx, ly = (int)fy;
        if (fx - (float)lx >= 0.5f) ++lx;
        if (fy - (float)ly >= 0.5f) ++ly;
        // nearest x8 up-sampling of the PAF as an index map (paf_to_pose.py:382): floor(l / up)
        int sx, sy;
        if constexpr (UP_POW2) {
          sx = lx >> up_shift;
          sy = ly >> up_shift;
        } else {
          sx = (int)floor((double)lx * inv_up);
          sy = (int)floor((double)ly * inv_up);
        }
        sx = min(max(sx, 0), w - 1);
        sy = min(max(sy, 0), h - 1);
        float px, py;
        if constexpr (A32) {
          // two full-rate 24-bit multiplies, spelled out: the compiler turns __umul24 into v_mad_u64_u32 + v_mul_lo_u32
          unsigned pix, boff;
          asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(pix) : "v"(sy), "s"(paf.ws), "v"(sx));
          asm("v_mul_u32_u24 %0, %1, %2" : "=v"(boff) : "s"(pix_bytes), "v"(pix));
          px = *reinterpret_cast<const float*>(img_x + boff);
          py = *reinterpret_cast<const float*>(img_y + boff);
#ifdef RTPOSE_EXP_LIMB_KEEP
          keep[i] = boff;
#endif
        } else {
          px = map_at(paf, n, sy, sx, chx);
          py = map_at(paf, n, sy, sx, chy);
        }
#ifdef RTPOSE_EXP_LIMB_WAIT0
        pxa[i] = px;
        pya[i] = py;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        float px = pxa[i], py = pya[i];
        asm volatile("" : "+v"(px), "+v"(py));
#endif
#ifdef RTPOSE_EXP_LIMB_ASM_DOT  // experiment: one product of the dot product opaque - no packed multiply / add there
        float typ;
        asm("v_mul_f32 %0, %1, %2" : "=v"(typ) : "v"(vy), "v"(py));
        const float s = vx * px + typ;
#else
        const float s = vx * px + vy * py;
#endif
        scores = scores + s;
        if (s > 0.05f) ++crit1;
      }
#ifdef RTPOSE_EXP_LIMB_KEEP
      asm volatile("" ::"v"(scores), "v"(keep[0]), "v"(keep[1]), "v"(keep[2]), "v"(keep[3]), "v"(keep[4]), "v"(keep[5]),
                   "v"(keep[6]), "v"(keep[7]), "v"(keep[8]), "v"(keep[9]));
#endif
      // min(0.5 h1 / norm - 1, 0) in double (cpp:238-240): 0 for every limb no longer than half the image (the quotient is
      // >= 1 then, and adding 0.0 to a float widened to double changes nothing)
      float crit2 = scores / 10.f;
      if (2.f * norm > (float)h1) {
        const double pen = fmin(0.0, 0.5 * (double)h1 / (double)norm - 1.0);
        crit2 = (float)((double)crit2 + pen);
      }
      if (crit1 > 6 && crit2 > 0.f) cand = crit2;
    }
    s_score(p) = cand;
  }
  __threadfence_block();
  __syncthreads();

  // greedy: repeatedly take the best remaining candidate whose endpoints are
  // both free == scanning the list sorted by descending score (cpp:96-124).
  // That equivalence needs the best free candidate to be UNIQUE at every step: when two free
  // candidates share the best score, std::sort's moves decide which the reference meets first
  // (and, if they do not exclude each other, the order of their connections, which orders the
  // subset rows).  Every step therefore also counts the free candidates AT the best score; a
  // count above one sends the limb to the replay below.  (Equal scores of which at most one is
  // still free when their turn comes cannot change anything: a used peak stays used.)
  const int max_conn = min(nA, nB);
  bool tie = false;
  for (int it = 0; it < max_conn; ++it) {
    float best = 0.f;
    int bidx = 0x7fffffff, cnt = 0;
    for (int p = tid; p < npairs; p += 256) {
      const float s = s_score(p);
      if (s >= best && s > 0.f) {
        const int a = p / nB, b = p - a * nB;
        if (!s_usedA[a] && !s_usedB[b]) {
          if (s > best) {
            best = s;
            bidx = p;
            cnt = 1;
          } else {
            ++cnt;  // p ascends: bidx stays the lowest
          }
        }
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int oi = __shfl_xor(bidx, o);
      const int oc = __shfl_xor(cnt, o);
      if (ob > best) {
        best = ob;
        bidx = oi;
        cnt = oc;
      } else if (ob == best) {
        bidx = min(bidx, oi);
        cnt += oc;
      }
    }
    if (lane == 0) {
      s_wbest[wave] = best;
      s_widx[wave] = bidx;
      s_wcnt[wave] = cnt;
    }
    __syncthreads();
    best = s_wbest[0];
    bidx = s_widx[0];
    cnt = s_wcnt[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      if (s_wbest[k] > best) {
        best = s_wbest[k];
        bidx = s_widx[k];
        cnt = s_wcnt[k];
      } else if (s_wbest[k] == best) {
        bidx = min(bidx, s_widx[k]);
        cnt += s_wcnt[k];
      }
    }
    if (!(best > 0.f)) break;  // uniform
    if (cnt > 1) {             // uniform
      tie = true;
      break;
    }
    if (tid == 0) {
      const int a = bidx / nB, b = bidx - a * nB;
      s_usedA[a] = 1;
      s_usedB[b] = 1;
      const int k = s_nconn++;
      s_conn[3 * k + 0] = a;
      s_conn[3 * k + 1] = b;
      s_conn[3 * k + 2] = __float_as_int(best);
    }
    __syncthreads();
  }
  __syncthreads();
  if (tie) {
    // the reference's own sequence for this limb, from the start: candidates in push order
    // (a ascending, b ascending, cpp:56-94), libstdc++'s std::sort, scan (cpp:98-123)
    for (int i = tid; i < kDecodeMaxPeaks; i += 256) {
      s_usedA[i] = 0;
      s_usedB[i] = 0;
    }
    __syncthreads();
    // candidates in push order: ordered compaction by ballot / popcount, like the peak ids
    unsigned long long* lds_list =
        reinterpret_cast<unsigned long long*>(s_score_lds + (SCORES_IN_LDS ? ((pcap * pcap + 1) & ~1) : 0));
    unsigned long long* ws_list = tie_ws + ((size_t)n * RTPOSE_NUM_LIMB + pair_id) * pcap * pcap;
    for (int pass = 0; pass < 2; ++pass) {  // pass 0 counts (LDS or workspace?), pass 1 writes
      unsigned long long* list = (s_nconn <= kTieLdsCands) ? lds_list : ws_list;  // (s_nconn = the count after pass 0)
      int base = 0;
      for (int start = 0; start < npairs; start += 256) {
        const int p = start + tid;
        const float sc = p < npairs ? s_score(p) : 0.f;
        const bool c = sc > 0.f;
        const unsigned long long mask = __ballot(c);
        if (lane == 0) s_wcnt[wave] = __popcll(mask);
        __syncthreads();
        int off = base;
        for (int k = 0; k < wave; ++k) off += s_wcnt[k];
        if (c && pass == 1)
          list[off + __popcll(mask & ((1ull << lane) - 1ull))] =
              ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned)p;
        base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
      }
      if (tid == 0) s_nconn = base;
      __syncthreads();
    }
    __threadfence_block();
    if (tid == 0) {
      const int nc = s_nconn;
      unsigned long long* list = (nc <= kTieLdsCands) ? lds_list : ws_list;
      SortReplay sr{list, s_stack};
      sr.sort_desc(nc);
      int k = 0;
      for (int c = 0; c < nc && k < max_conn; ++c) {
        const unsigned long long e = list[c];
        const int p = (int)(unsigned)e;
        const int a = p / nB, b = p - a * nB;
        if (s_usedA[a] || s_usedB[b]) continue;
        s_usedA[a] = 1;
        s_usedB[b] = 1;
        s_conn[3 * k + 0] = a;
        s_conn[3 * k + 1] = b;
        s_conn[3 * k + 2] = (int)(unsigned)(e >> 32);
        ++k;
      }
      s_nconn = k;
    }
    __syncthreads();
  }
  __syncthreads();
  const int nconn = s_nconn;
  if (tid == 0) cn[0] = nconn;
  for (int i = tid; i < 3 * nconn; i += 256) cn[1 + i] = s_conn[i];
}


// ------------------------------------------------------------------------------
// 3. Person grouping + prune (pafprocess.cpp:126-191), one wave per image
// ------------------------------------------------------------------------------
// The walk over the connections is the reference's serial loop (every connection sees the rows the previous one left), so
// what it costs is the latency of one step.  Round 6: a limb's connections are STAGED first - the 64 lanes fetch the
// (a, b, score) triples, the two peak ids and the two peak scores of up to 64 connections at a time into LDS, in parallel - and
// the serial walk then touches LDS only (round 5: three dependent global loads per connection, ~1.6 us each step: 164 us per
// batch whatever its size; now ~20).  WRITE_IDS: the running peak ids of paf_to_pose.py:141-142 (the former peak_prefix_kernel
// launch) are written here - and taken arithmetically, id = s_start[part] + index, instead of read back.
// STAGE_ALL: the connections of ALL 19 limbs are staged in one sweep over the flattened (limb, connection) list - every
// global round trip of the kernel is then taken once, with all lanes' loads in flight together, instead of once per limb (the
// host picks it when 19 * pcap staged connections fit the LDS next to the rows: pcap <= 128).
constexpr int kStageWords = 5;  // per staged connection: cid1, cid2, connection score, score of peak 2, score of peak 1
#ifdef RTPOSE_EXP_GROUP_TIMELINE  // developer build: wall_clock64 stamps of the kernel's phases, per image (tools/exp/group_timeline.py)
__device__ unsigned long long g_group_tl[256][8];
#define RTPOSE_GTL(i)                                        \
  if (lane == 0 && n < 256) g_group_tl[n][i] = wall_clock64()
#else
#define RTPOSE_GTL(i)
#endif
template <bool WRITE_IDS, bool STAGE_ALL>
__global__ __launch_bounds__(64) void group_kernel(int pcap, int hcap, int32_t* __restrict__ result,
                                                   int result_words, const int32_t* __restrict__ conn,
                                                   int conn_words, int row_cap,
                                                   float* __restrict__ rows_ws) {
  const int n = blockIdx.x, lane = threadIdx.x;
  int32_t* res = result + (size_t)n * result_words;
  const int32_t* cnb = conn + (size_t)n * conn_words;
  extern __shared__ float rows_lds[];
  // [row_cap][20 + alive]: the reference's `subset`; LDS unless grown past kLdsRows; then the staged connections of one limb
  float* rows = row_cap <= kLdsRows ? rows_lds : rows_ws + (size_t)n * row_cap * 21;
  float* stage = rows_lds + (row_cap <= kLdsRows ? (size_t)row_cap * 21 : 0);
  RTPOSE_GTL(0);

  __shared__ int s_start[RTPOSE_NUM_PART + 1];
  if (lane == 0) {
    int acc = 0;
    for (int p = 0; p < RTPOSE_NUM_PART; ++p) {
      s_start[p] = acc;
      acc += res[kResPartCount + p];
    }
    s_start[RTPOSE_NUM_PART] = acc;
    if (WRITE_IDS) res[kResHeader + 0] = acc;
  }
  __syncthreads();
  rtpose_peak* peaks = reinterpret_cast<rtpose_peak*>(res + kResPeaks);
  if (WRITE_IDS) {
    // ids = running counter over parts then peaks (paf_to_pose.py:141-142)
    for (int p = 0; p < RTPOSE_NUM_PART; ++p) {
      const int cnt = s_start[p + 1] - s_start[p];
      for (int i = lane; i < cnt; i += 64) peaks[(size_t)p * pcap + i].id = s_start[p] + i;
    }
  }
  // peak_infos_line[pos] (cpp:38-43): part-major position -> peak
  auto line_peak_score = [&](int pos) -> float {
    int p = 0;
    while (p + 1 < RTPOSE_NUM_PART && pos >= s_start[p + 1]) ++p;
    return peaks[(size_t)p * pcap + (pos - s_start[p])].score;
  };
  const int npeaks = s_start[RTPOSE_NUM_PART];

  // everything the walk needs from global memory about connection c of limb `limb`, into st[0 .. kStageWords)
  auto stage_conn = [&](int limb, int c, float* st) {
    const int part1 = kPairs[limb][0], part2 = kPairs[limb][1];
    const int32_t* cn = cnb + (size_t)limb * (1 + 3 * pcap);
    const rtpose_peak* pA = peaks + (size_t)part1 * pcap;
    const rtpose_peak* pB = peaks + (size_t)part2 * pcap;
    const int ia = cn[1 + 3 * c], ib = cn[1 + 3 * c + 1];
    const int id1 = WRITE_IDS ? s_start[part1] + ia : pA[ia].id;
    const int id2 = WRITE_IDS ? s_start[part2] + ib : pB[ib].id;
    st[0] = (float)id1;
    st[1] = (float)id2;
    st[2] = __int_as_float(cn[1 + 3 * c + 2]);
    st[3] = (id2 >= 0 && id2 < npeaks) ? (WRITE_IDS ? pB[ib].score : line_peak_score(id2)) : 0.f;
    st[4] = (id1 >= 0 && id1 < npeaks) ? (WRITE_IDS ? pA[ia].score : line_peak_score(id1)) : 0.f;
  };
  __shared__ int s_cbase[RTPOSE_NUM_LIMB + 1];
  RTPOSE_GTL(1);
  if (STAGE_ALL) {
    if (lane < RTPOSE_NUM_LIMB) s_cbase[lane + 1] = cnb[(size_t)lane * (1 + 3 * pcap)];
    __syncthreads();
    if (lane == 0) {
      s_cbase[0] = 0;
      for (int l = 0; l < RTPOSE_NUM_LIMB; ++l) s_cbase[l + 1] += s_cbase[l];
    }
    __syncthreads();
    const int total = s_cbase[RTPOSE_NUM_LIMB];
    for (int i = lane; i < total; i += 64) {
      int limb = 0;
      while (i >= s_cbase[limb + 1]) ++limb;
      stage_conn(limb, i - s_cbase[limb], stage + (size_t)i * kStageWords);
    }
    __threadfence_block();
    __syncthreads();
  }

  RTPOSE_GTL(2);
  int nrows = 0;
  bool overflow = false;
  for (int pair_id = 0; pair_id < 19; ++pair_id) {
    const int part1 = kPairs[pair_id][0], part2 = kPairs[pair_id][1];
    int nconn;
    const float* lst = stage;
    if (STAGE_ALL) {
      nconn = s_cbase[pair_id + 1] - s_cbase[pair_id];
      lst = stage + (size_t)s_cbase[pair_id] * kStageWords;
    } else {
      // stage this limb's connections, 64 in flight at a time
      nconn = cnb[(size_t)pair_id * (1 + 3 * pcap)];
      for (int c = lane; c < nconn; c += 64) stage_conn(pair_id, c, stage + (size_t)c * kStageWords);
      __threadfence_block();
      __syncthreads();
    }
    for (int c = 0; c < nconn; ++c) {
      const float* st = lst + (size_t)c * kStageWords;
      const float cid1 = st[0], cid2 = st[1], cscore = st[2], s2 = st[3];
      // search alive rows in order
      int found = 0, idx1 = 0, idx2 = 0;
      for (int r0 = 0; r0 < nrows; r0 += 64) {
        const int r = r0 + lane;
        bool hit = false;
        if (r < nrows && rows[(size_t)r * 21 + 20] != 0.f)
          hit = rows[(size_t)r * 21 + part1] == cid1 || rows[(size_t)r * 21 + part2] == cid2;
        unsigned long long m = __ballot(hit);
        while (m) {
          const int b = __ffsll((long long)m) - 1;
          if (found == 0) idx1 = r0 + b;
          if (found == 1) idx2 = r0 + b;
          ++found;
          m &= m - 1;
        }
      }
      if (found == 1) {
        float* row = rows + (size_t)idx1 * 21;
        if (lane == 0 && row[part2] != cid2) {
          row[part2] = cid2;
          row[19] = row[19] + 1.f;
          row[18] = row[18] + (s2 + cscore);
        }
      } else if (found == 2) {
        float* r1 = rows + (size_t)idx1 * 21;
        float* r2 = rows + (size_t)idx2 * 21;
        bool both = false;
        if (lane < 18) both = r1[lane] > 0.f && r2[lane] > 0.f;  // cid 0 reads as absent (cpp:155)
        const bool membership = __any(both);
        if (!membership) {
          if (lane < 18) r1[lane] = r1[lane] + (r2[lane] + 1.f);
          if (lane == 0) {
            r1[19] = r1[19] + r2[19];
            r1[18] = r1[18] + r2[18];
            r1[18] = r1[18] + cscore;
            r2[20] = 0.f;  // erase(subset_idx2): order of the survivors is kept
          }
        } else if (lane == 0) {
          r1[part2] = cid2;
          r1[19] = r1[19] + 1.f;
          r1[18] = r1[18] + (s2 + cscore);
        }
      } else if (found == 0 && pair_id < 18) {
        if (nrows < row_cap) {
          float* row = rows + (size_t)nrows * 21;
          const float s1 = st[4];
          if (lane < 18) row[lane] = (lane == part1) ? cid1 : ((lane == part2) ? cid2 : -1.f);
          if (lane == 0) {
            row[19] = 2.f;
            row[18] = (s1 + s2) + cscore;
            row[20] = 1.f;
          }
          ++nrows;
        } else {
          overflow = true;
        }
      }
      __threadfence_block();
      __syncthreads();  // single-wave block: orders the row updates
    }
  }

  RTPOSE_GTL(3);
#ifdef RTPOSE_EXP_GROUP_TIMELINE
  if (lane == 0 && n < 256) g_group_tl[n][6] = (unsigned long long)(STAGE_ALL ? s_cbase[RTPOSE_NUM_LIMB] : 0);
#endif
  // prune (cpp:187-191) and emit
  int nh = 0;
  int32_t* hparts = res + kResPeaks + 4 * RTPOSE_NUM_PART * pcap;
  float* hscore = reinterpret_cast<float*>(hparts + (size_t)RTPOSE_NUM_PART * hcap);
  for (int r = 0; r < nrows; ++r) {
    const float* row = rows + (size_t)r * 21;
    if (row[20] == 0.f) continue;
    if (row[19] < 4.f || row[18] / row[19] < 0.3f) continue;
    if (nh < hcap) {
      if (lane < 18) hparts[(size_t)nh * RTPOSE_NUM_PART + lane] = (int)row[lane];
      if (lane == 0) hscore[nh] = row[18] / row[19];
      ++nh;
    } else {
      overflow = true;
    }
  }
  if (lane == 0) {
    res[kResHeader + 1] = nh;
    if (overflow) atomicOr(&res[kResHeader + 2], kOverflowHumans);
  }
  RTPOSE_GTL(4);
}

// header + part counts of every record <- 0, except header[3] / [4] = the capacities the record is laid out for
// (max_peaks_per_part, max_humans): a record block describes itself, a consumer that parses it later - after the
// producer has grown its tables - does not need the producer's cfg of that moment.
__global__ void clear_header_kernel(int32_t* __restrict__ result, int result_words, int N, int pcap, int hcap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N * kResPeaks) {
    const int wd = i % kResPeaks;
    result[(size_t)(i / kResPeaks) * result_words + wd] = wd == kResHeader + 3 ? pcap : wd == kResHeader + 4 ? hcap : 0;
  }
}

static MapView to_view(const float* base, const rtpose_layout* l) {
  return MapView{base, l->cstride, l->choff, l->ws, l->hs, l->lead};
}

static int check_cfg(const rtpose_decode_cfg* cfg) {
  if (!cfg) return fail(RTPOSE_E_INVAL, "decode: cfg is NULL");
  if (cfg->num_keypoints < 1 || cfg->num_keypoints > RTPOSE_NUM_PART)
    return fail(RTPOSE_E_INVAL, "decode: num_keypoints must be in [1,18]");
  if (cfg->upsample < 1 || cfg->upsample > kMaxUp)
    return fail(RTPOSE_E_INVAL, "decode: upsample must be in [1,%d]", kMaxUp);
  if (cfg->max_peaks_per_part < 1 || cfg->max_peaks_per_part > kDecodeMaxPeaks)
    return fail(RTPOSE_E_INVAL, "decode: max_peaks_per_part must be in [1,%d]", kDecodeMaxPeaks);
  if (cfg->max_humans < 1) return fail(RTPOSE_E_INVAL, "decode: max_humans must be >= 1");
  return 0;
}

// scipy.ndimage._filters._gaussian_kernel1d(sigma = 3, order 0, radius 12) in float64, the sum
// taken in numpy's pairwise order (8 running partial sums, then the tail) so that the weights are
// the ones scipy correlates with, bit for bit, wherever libm's exp agrees with numpy's.
static GaussW gauss_weights() {
  GaussW g;
  const int n = 2 * kGaussR + 1;
  for (int i = -kGaussR; i <= kGaussR; ++i) g.w[i + kGaussR] = exp(-0.5 / 9.0 * (double)(i * i));
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = g.w[j];
  int i = 8;
  for (; i < n - n % 8; i += 8)
    for (int j = 0; j < 8; ++j) r[j] += g.w[i + j];
  double sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) sum += g.w[i];
  for (int k = 0; k < n; ++k) g.w[k] /= sum;
  return g;
}

int nms_launch(const float* heat, const rtpose_layout* lheat, int N, int h, int w,
               const rtpose_decode_cfg* cfg, void* result, hipStream_t s, int flags, bool with_ids) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (N <= 0 || h <= 0 || w <= 0) return fail(RTPOSE_E_INVAL, "decode: empty batch");
  const int words = decode_result_words(cfg);
  int32_t* res = static_cast<int32_t*>(result);
  hipLaunchKernelGGL(clear_header_kernel, dim3(ceil_div(N * kResPeaks, 256)), dim3(256), 0, s, res, words, N,
                     cfg->max_peaks_per_part, cfg->max_humans);
  if (flags & ~(RTPOSE_NMS_NO_REFINE | RTPOSE_NMS_GAUSSIAN)) return fail(RTPOSE_E_INVAL, "nms: unknown flag");
  if (flags) {
    const size_t dyn = (size_t)2 * 25 * cfg->upsample * cfg->upsample * sizeof(float);
    hipLaunchKernelGGL(nms_refine_opt_kernel, dim3(cfg->num_keypoints, N), dim3(256), dyn, s,
                       to_view(heat, lheat), h, w, cfg->upsample, 1.0 / (double)cfg->upsample,
                       cfg->thresh_heatmap, cfg->max_peaks_per_part, res, words, flags, gauss_weights());
  } else {
    hipLaunchKernelGGL(nms_refine_kernel, dim3(cfg->num_keypoints, N), dim3(256), 0, s, to_view(heat, lheat),
                       h, w, cfg->upsample, 1.0 / (double)cfg->upsample, cfg->thresh_heatmap,
                       cfg->max_peaks_per_part, res, words);
  }
  if (with_ids)  // (a full decode writes the ids and the peak total in group_kernel<true> instead: one launch less)
    hipLaunchKernelGGL(peak_prefix_kernel, dim3(N), dim3(64), 0, s, cfg->max_peaks_per_part, res, words,
                       cfg->num_keypoints);
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

// assignment + grouping on peak tables already in `result`
int assign_group_launch(const float* paf, const rtpose_layout* lpaf, int N, int h, int w, double inv_up,
                        int h1, const rtpose_decode_cfg* cfg, void* workspace, size_t workspace_bytes,
                        void* result, hipStream_t s, bool write_ids) {
  if (workspace_bytes < decode_workspace_bytes(cfg, N))
    return fail(RTPOSE_E_INVAL, "decode: workspace too small");
  const int pcap = cfg->max_peaks_per_part;
  const int words = decode_result_words(cfg);
  const int conn_words = decode_conn_words(cfg);
  int32_t* res = static_cast<int32_t*>(result);
  int32_t* conn = static_cast<int32_t*>(workspace);
  char* wsb = static_cast<char*>(workspace);
  float* score_ws = reinterpret_cast<float*>(wsb + decode_ws_conn_bytes(cfg, N));
  float* rows_ws = reinterpret_cast<float*>(wsb + decode_ws_conn_bytes(cfg, N) + decode_ws_score_bytes(cfg, N));
  unsigned long long* tie_ws = reinterpret_cast<unsigned long long*>(wsb + decode_ws_conn_bytes(cfg, N) + decode_ws_score_bytes(cfg, N) +
                                               decode_ws_rows_bytes(cfg, N));
  // scores (when they fit) + the LDS-resident candidate list of a limb that replays std::sort
  const size_t lds = (pcap * pcap <= kLdsPairs ? (size_t)((pcap * pcap + 1) & ~1) * sizeof(float) : 0) +
                     (size_t)kTieLdsCands * sizeof(unsigned long long);
  static PerDeviceOnce attr_set;  // zero-initialised; the attribute is per device
  const int dev = current_device();
  if (!attr_set.is_set(dev)) {
    const void* limb_kernels[8] = {reinterpret_cast<const void*>(limb_assign_kernel<true, true, true>),
                                   reinterpret_cast<const void*>(limb_assign_kernel<true, false, true>),
                                   reinterpret_cast<const void*>(limb_assign_kernel<false, true, true>),
                                   reinterpret_cast<const void*>(limb_assign_kernel<false, false, true>),
                                   reinterpret_cast<const void*>(limb_assign_kernel<true, true, false>),
                                   reinterpret_cast<const void*>(limb_assign_kernel<true, false, false>),
                                   reinterpret_cast<const void*>(limb_assign_kernel<false, true, false>),
                                   reinterpret_cast<const void*>(limb_assign_kernel<false, false, false>)};
    for (const void* k : limb_kernels)
      RTPOSE_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    const void* group_kernels[4] = {reinterpret_cast<const void*>(group_kernel<true, true>),
                                    reinterpret_cast<const void*>(group_kernel<true, false>),
                                    reinterpret_cast<const void*>(group_kernel<false, true>),
                                    reinterpret_cast<const void*>(group_kernel<false, false>)};
    for (const void* k : group_kernels)
      RTPOSE_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    attr_set.set(dev);
  }
  const int up = cfg->upsample;
  int up_shift = -1;  // log2(up) when it is a power of two
  for (int k = 0; k < 8; ++k)
    if (up == (1 << k)) up_shift = k;
  const bool in_lds = pcap * pcap <= kLdsPairs;
  // 32-bit per-sample offsets when an image's maps span less than 2^31 bytes and the pixel index fits the 24-bit multiplier
  bool a32 = (long long)lpaf->hs * lpaf->ws < (1ll << 24) && (long long)lpaf->cstride * (long long)sizeof(float) < (1 << 24) &&
             (long long)lpaf->hs * lpaf->ws * lpaf->cstride * (long long)sizeof(float) < (1ll << 31);
#ifdef RTPOSE_DEV_BUILD
  {
    static const char* e = getenv("RTPOSE_LIMB_A32");  // developer A/B of the two addressing forms (tools/exp/overlap_flake.py)
    if (e && e[0] == '0') a32 = false;
  }
#endif
  {
#define RTPOSE_LIMB_A(L, P, A)                                                                                         \
  hipLaunchKernelGGL((limb_assign_kernel<L, P, A>), dim3(RTPOSE_NUM_LIMB, N), dim3(256), lds, s, to_view(paf, lpaf), h, \
                     w, inv_up, up_shift, h1, pcap, res, words, conn, conn_words, score_ws, tie_ws)
#define RTPOSE_LIMB(L, P)              \
  do {                                 \
    if (a32) RTPOSE_LIMB_A(L, P, true); \
    else RTPOSE_LIMB_A(L, P, false);   \
  } while (0)
    if (in_lds && up_shift >= 0) RTPOSE_LIMB(true, true);
    else if (in_lds) RTPOSE_LIMB(true, false);
    else if (up_shift >= 0) RTPOSE_LIMB(false, true);
    else RTPOSE_LIMB(false, false);
#undef RTPOSE_LIMB_A
#undef RTPOSE_LIMB
  }
  const int row_cap = decode_row_cap(cfg);
  // subset rows (when they fit) + the staged connections: of all 19 limbs when that fits beside the rows, else of one limb
  const size_t rows_bytes = row_cap <= kLdsRows ? (size_t)row_cap * 21 * sizeof(float) : 0;
  const size_t all_bytes = (size_t)RTPOSE_NUM_LIMB * pcap * kStageWords * sizeof(float);
  const bool stage_all = rows_bytes + all_bytes <= 96 * 1024;
  const size_t rows_lds = rows_bytes + (stage_all ? all_bytes : (size_t)pcap * kStageWords * sizeof(float));
#define RTPOSE_GROUP(W, S)                                                                                               \
  hipLaunchKernelGGL((group_kernel<W, S>), dim3(N), dim3(64), rows_lds, s, pcap, cfg->max_humans, res, words, conn, \
                     conn_words, row_cap, rows_ws)
  if (write_ids && stage_all) RTPOSE_GROUP(true, true);
  else if (write_ids) RTPOSE_GROUP(true, false);
  else if (stage_all) RTPOSE_GROUP(false, true);
  else RTPOSE_GROUP(false, false);
#undef RTPOSE_GROUP
  RTPOSE_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace rtpose

using namespace rtpose;

extern "C" {

#ifdef RTPOSE_EXP_GROUP_TIMELINE
int rtpose_exp_group_timeline(unsigned long long* out, int n) {  // [n][8]: stamps 0..4, [6] = connections of the image
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rtpose::g_group_tl), sizeof(unsigned long long) * 8 * (n < 256 ? n : 256));
}
#endif

size_t rtpose_decode_workspace_bytes(const rtpose_decode_cfg* cfg, int N) {
  if (check_cfg(cfg) || N <= 0) return 0;
  return decode_workspace_bytes(cfg, N);
}

size_t rtpose_decode_result_bytes(const rtpose_decode_cfg* cfg, int N) {
  if (check_cfg(cfg) || N <= 0) return 0;
  return (size_t)N * decode_result_words(cfg) * sizeof(int32_t);
}

int rtpose_nms_batch_ex(const float* heat, const rtpose_layout* lheat, int N, int h, int w,
                        const rtpose_decode_cfg* cfg, int nms_flags, void* result, void* stream) {
  if (!heat || !lheat || !result) return fail(RTPOSE_E_INVAL, "nms: NULL argument");
  static thread_local CheckedPtr c_heat, c_res;
  const int dev = current_device();
  int rcd = c_heat.check(heat, dev, "nms", "the heat-map tensor");
  if (!rcd) rcd = c_res.check(result, dev, "nms", "the result block");
  if (rcd) return rcd;
  return nms_launch(heat, lheat, N, h, w, cfg, result, as_stream(stream), nms_flags, /*with_ids=*/true);
}

int rtpose_nms_batch(const float* heat, const rtpose_layout* lheat, int N, int h, int w,
                     const rtpose_decode_cfg* cfg, void* result, void* stream) {
  return rtpose_nms_batch_ex(heat, lheat, N, h, w, cfg, 0, result, stream);
}

int rtpose_gaussian_kernel1d(double* weights, int cap) {
  if (!weights || cap < 2 * kGaussR + 1) return fail(RTPOSE_E_INVAL, "gaussian_kernel1d: need room for 25 doubles");
  const GaussW g = gauss_weights();
  for (int i = 0; i < 2 * kGaussR + 1; ++i) weights[i] = g.w[i];
  return 2 * kGaussR + 1;
}

int rtpose_decode_batch(const float* heat, const rtpose_layout* lheat, const float* paf,
                        const rtpose_layout* lpaf, int N, int h, int w, const rtpose_decode_cfg* cfg,
                        void* workspace, size_t workspace_bytes, void* result, void* stream) {
  return rtpose_decode_batch_ex(heat, lheat, paf, lpaf, N, h, w, cfg, 0, workspace, workspace_bytes, result, stream);
}

int rtpose_decode_batch_ex(const float* heat, const rtpose_layout* lheat, const float* paf,
                           const rtpose_layout* lpaf, int N, int h, int w, const rtpose_decode_cfg* cfg,
                           int nms_flags, void* workspace, size_t workspace_bytes, void* result, void* stream) {
  if (!heat || !lheat || !paf || !lpaf || !workspace || !result)
    return fail(RTPOSE_E_INVAL, "decode: NULL argument");
  // (the look-ups are repeated only when a caller passes other pointers than last time on this thread)
  static thread_local CheckedPtr c_heat, c_paf, c_ws, c_res;
  const int dev = current_device();
  int rcd = c_heat.check(heat, dev, "decode", "the heat-map tensor");
  if (!rcd) rcd = c_paf.check(paf, dev, "decode", "the PAF tensor");
  if (!rcd) rcd = c_ws.check(workspace, dev, "decode", "the workspace");
  if (!rcd) rcd = c_res.check(result, dev, "decode", "the result block");
  if (rcd) return rcd;
  int rc = nms_launch(heat, lheat, N, h, w, cfg, result, as_stream(stream), nms_flags, /*with_ids=*/false);
  if (rc) return rc;
  return assign_group_launch(paf, lpaf, N, h, w, 1.0 / (double)cfg->upsample, h * cfg->upsample, cfg,
                             workspace, workspace_bytes, result, as_stream(stream), /*write_ids=*/true);
}

}  // extern "C"
