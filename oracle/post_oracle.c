/*
 * ORACLE — test infrastructure, NOT product code.
 *
 * Plain-C CPU restatement of the reference's post-processing path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it
 * (as the checker / the timed CPU baseline), never the product path.
 *
 *   oracle_nms          restates lib/utils/paf_to_pose.py:67-145 (NMS) with
 *                       find_peaks :25-38 and compute_resized_coords :41-64.
 *                       Third-party pieces restated from their published
 *                       behaviour (versions unpinned by the reference):
 *                         scipy.ndimage.maximum_filter(footprint=cross,
 *                           mode='reflect')              -> peak test below
 *                         cv2.resize(patch, fx=fy=up, INTER_CUBIC) on float32
 *                           (OpenCV imgproc/resize.cpp: A=-0.75, half-pixel
 *                           mapping, replicate border, horizontal pass then
 *                           vertical pass, left-to-right float accumulation)
 *   oracle_process_paf  restates lib/pafprocess/pafprocess.cpp:22-194 +
 *                       get_paf_vectors :220-238, roundpaf :240-242; the x8
 *                       INTER_NEAREST up-sampling of paf_to_pose.py:382-385 is
 *                       folded into the index (src = min(floor(dst * 1/up), n-1)).
 *
 * Pinning: tests/test_oracle_cpu.py checks oracle_process_paf against the
 * reference's own C++ compiled unmodified (oracle/_ref/libpafprocess_ref.so,
 * built by oracle/Makefile from /root/reference) and against the committed
 * golden vectors that library produced; oracle_nms is checked against
 * scipy.ndimage.maximum_filter (the call the reference makes) and against torch's
 * CPU bicubic.  cv2 itself is absent from the image: the bicubic restatement is
 * "parity unpinned" w.r.t. a real OpenCV build.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (no fused multiply-add, so the
 * float operation order written here is the one executed).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NUM_PART 18
#define NUM_PAIR 19
#define STEP_PAF 10

/* pafprocess.h:16-24 */
static const int PAIRS_NET[NUM_PAIR][2] = {
    {12, 13}, {20, 21}, {14, 15}, {16, 17}, {22, 23}, {24, 25}, {0, 1}, {2, 3}, {4, 5}, {6, 7},
    {8, 9}, {10, 11}, {28, 29}, {30, 31}, {34, 35}, {32, 33}, {36, 37}, {18, 19}, {26, 27}};
static const int PAIRS[NUM_PAIR][2] = {
    {1, 2}, {1, 5}, {2, 3}, {3, 4}, {5, 6}, {6, 7}, {1, 8}, {8, 9}, {9, 10}, {1, 11},
    {11, 12}, {12, 13}, {1, 0}, {0, 14}, {14, 16}, {0, 15}, {15, 17}, {2, 16}, {5, 17}};

/* ---- OpenCV cubic weights (imgproc: interpolateCubic) ------------------- */
static void cubic_coeffs(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* cv2.resize(patch[ph][pw], fx=fy=up, INTER_CUBIC) -> dst[ph*up][pw*up] */
static void resize_cubic(const float* patch, int ph, int pw, int up, float* dst) {
  const int dw = pw * up, dh = ph * up;
  const double scale = 1.0 / (double)up;
  int* xofs = (int*)malloc(sizeof(int) * (dw > dh ? dw : dh));
  float* alpha = (float*)malloc(sizeof(float) * 4 * (dw > dh ? dw : dh));
  float* hbuf = (float*)malloc(sizeof(float) * ph * dw);
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    xofs[dx] = sx;
    cubic_coeffs(fx, alpha + 4 * dx);
  }
  for (int r = 0; r < ph; ++r)
    for (int dx = 0; dx < dw; ++dx) {
      float v = 0.f;
      for (int j = 0; j < 4; ++j) {
        int sxj = clampi(xofs[dx] - 1 + j, 0, pw - 1);
        v = v + patch[r * pw + sxj] * alpha[4 * dx + j];
      }
      hbuf[r * dw + dx] = v;
    }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    float beta[4];
    cubic_coeffs(fy, beta);
    for (int dx = 0; dx < dw; ++dx) {
      float v = 0.f;
      for (int j = 0; j < 4; ++j) {
        int syj = clampi(sy - 1 + j, 0, ph - 1);
        v = v + hbuf[syj * dw + dx] * beta[j];
      }
      dst[dy * dw + dx] = v;
    }
  }
  free(xofs);
  free(alpha);
  free(hbuf);
}

/* cv2.resize(src, None, fx=up, fy=up, interpolation=INTER_CUBIC) on one float32 plane
 * (exported for oracle/cv2_restate.py, the cv2 stand-in the reference modules are imported with) */
void oracle_resize_cubic(const float* src, int h, int w, int up, float* dst) { resize_cubic(src, h, w, up, dst); }

/*
 * scipy.ndimage.gaussian_filter(img, sigma) on a float32 plane, mode='reflect', as
 * paf_to_pose.py:121-122 calls it.  Restated from scipy's published algorithm
 * (ndimage/_filters.py gaussian_filter -> gaussian_filter1d -> correlate1d,
 * src/ni_filters.c NI_Correlate1D; scipy unpinned by requirements.txt:6, 1.15.3 here):
 * one 1-D pass per axis, axis 0 (columns, i.e. along y) first, then axis 1; each pass
 * accumulates in double, symmetric-kernel form  out = w[0]*c + sum_{j=-r..-1} (x[j] + x[-j]) * w[j],
 * and is stored to the float32 output before the next pass reads it.  `weights` holds
 * the 2r+1 normalised kernel values exp(-0.5 x^2 / sigma^2) / sum (computed by the caller
 * with numpy exactly like _gaussian_kernel1d); reflect = (d c b a | a b c d | d c b a).
 */
static int reflect_idx(int i, int n) {
  while (i < 0 || i >= n) {
    if (i < 0) i = -i - 1;
    if (i >= n) i = 2 * n - 1 - i;
  }
  return i;
}
void oracle_gaussian_filter_f32(float* img, int h, int w, const double* weights, int radius) {
  float* tmp = (float*)malloc(sizeof(float) * (size_t)h * w);
  const double* wc = weights + radius;
  for (int x = 0; x < w; ++x)
    for (int y = 0; y < h; ++y) {
      double acc = (double)img[y * w + x] * wc[0];
      for (int j = -radius; j < 0; ++j)
        acc += ((double)img[reflect_idx(y + j, h) * w + x] + (double)img[reflect_idx(y - j, h) * w + x]) * wc[j];
      tmp[y * w + x] = (float)acc;
    }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      double acc = (double)tmp[y * w + x] * wc[0];
      for (int j = -radius; j < 0; ++j)
        acc += ((double)tmp[y * w + reflect_idx(x + j, w)] + (double)tmp[y * w + reflect_idx(x - j, w)]) * wc[j];
      img[y * w + x] = (float)acc;
    }
  free(tmp);
}

/*
 * NMS.  heat: dense HWC float32 [h][w][C].  Writes rows (x, y, score, id, part)
 * — the float32 joint_list of paf_to_pose_cpp (paf_to_pose.py:376-378) — in
 * part order, peaks of one part in row-major order.  Returns the number of peaks,
 * or -1 if more than `cap`.
 * refine = bool_refine_center (:105, default 1), gaussian = bool_gaussian_filt (:121, default 0;
 * gw/gr = normalised Gaussian weights [2*gr+1] and radius, see oracle_gaussian_filter_f32).
 */
int oracle_nms_ex(const float* heat, int h, int w, int C, int num_keypoints, float thr, int up, int cap,
                  float* joint_list, int refine, int gaussian, const double* gw, int gr) {
  int total = 0;
  const int win = 2; /* paf_to_pose.py:100 */
  float* ups = (float*)malloc(sizeof(float) * (5 * up) * (5 * up));
  for (int part = 0; part < num_keypoints; ++part) {
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const float v = heat[((size_t)y * w + x) * C + part];
        /* find_peaks :34-35: (maximum_filter(img, cross) == img) * (img > thr);
         * 'reflect' border == compare with in-image neighbours only */
        if (!(v > thr)) continue;
        if (y > 0 && heat[((size_t)(y - 1) * w + x) * C + part] > v) continue;
        if (y + 1 < h && heat[((size_t)(y + 1) * w + x) * C + part] > v) continue;
        if (x > 0 && heat[((size_t)y * w + x - 1) * C + part] > v) continue;
        if (x + 1 < w && heat[((size_t)y * w + x + 1) * C + part] > v) continue;
        if (total >= cap) {
          free(ups);
          return -1;
        }
        float* o = joint_list + (size_t)total * 5;
        const double cx = ((double)x + 0.5) * up - 0.5, cy = ((double)y + 0.5) * up - 0.5;
        if (!refine) { /* :135-139: refined_center = [0, 0], score = map_orig[y, x] */
          o[0] = (float)cx;
          o[1] = (float)cy;
          o[2] = v;
          o[3] = (float)total;
          o[4] = (float)part;
          ++total;
          continue;
        }
        /* :108-113 clipped 5x5 window */
        const int x_min = x - win < 0 ? 0 : x - win, y_min = y - win < 0 ? 0 : y - win;
        const int x_max = x + win > w - 1 ? w - 1 : x + win, y_max = y + win > h - 1 ? h - 1 : y + win;
        const int pw = x_max - x_min + 1, ph = y_max - y_min + 1;
        float patch[25];
        for (int r = 0; r < ph; ++r)
          for (int c = 0; c < pw; ++c)
            patch[r * pw + c] = heat[((size_t)(y_min + r) * w + x_min + c) * C + part];
        resize_cubic(patch, ph, pw, up, ups); /* :114-115 */
        if (gaussian) oracle_gaussian_filter_f32(ups, ph * up, pw * up, gw, gr); /* :121-122 */
        /* :125-126 first arg-max in row-major order */
        int best = 0;
        for (int i = 1; i < ph * up * pw * up; ++i)
          if (ups[i] > ups[best]) best = i;
        const int row = best / (pw * up), col = best % (pw * up);
        /* :129-141 in float64: (c+0.5)*up-0.5 + (loc - ((c-cmin+0.5)*up-0.5)) */
        const double pcx = ((double)(x - x_min) + 0.5) * up - 0.5, pcy = ((double)(y - y_min) + 0.5) * up - 0.5;
        o[0] = (float)(cx + ((double)col - pcx));
        o[1] = (float)(cy + ((double)row - pcy));
        o[2] = ups[best];
        o[3] = (float)total; /* cnt_total_joints */
        o[4] = (float)part;
        ++total;
      }
  }
  free(ups);
  return total;
}

int oracle_nms(const float* heat, int h, int w, int C, int num_keypoints, float thr, int up, int cap,
               float* joint_list) {
  return oracle_nms_ex(heat, h, w, C, num_keypoints, thr, up, cap, joint_list, 1, 0, 0, 0);
}

/* ---- process_paf ------------------------------------------------------------ */
typedef struct {
  int x, y;
  float score;
  int id;
} Peak;

typedef struct {
  int idx1, idx2;
  float score;
} Cand;

typedef struct {
  int cid1, cid2;
  float score;
} Conn;

static int roundpaf(float v) { return (int)(v + 0.5); }

/*
 * std::sort(candidates.begin(), candidates.end(), comp_candidate) as libstdc++ (GCC 11,
 * bits/stl_algo.h: the third-party code the reference's pafprocess.cpp:97 links against
 * when built with this image's g++) executes it: introsort = median-of-3 quicksort down
 * to 16-element runs with a 2*floor(log2 n) depth limit (heap sort beyond it), then one
 * insertion-sort pass.  Restated from the published algorithm so that the oracle can
 * reproduce the reference's choice among EQUAL-score candidates too (sort_mode 1);
 * comp(a, b) = a.score > b.score (pafprocess.cpp:244-246).
 */
static int cand_gt(const Cand* a, const Cand* b) { return a->score > b->score; }
static void cand_swap(Cand* a, Cand* b) {
  Cand t = *a;
  *a = *b;
  *b = t;
}
static void std_unguarded_linear_insert(Cand* last) {
  Cand val = *last;
  Cand* next = last - 1;
  while (cand_gt(&val, next)) {
    *last = *next;
    last = next;
    --next;
  }
  *last = val;
}
static void std_insertion_sort(Cand* first, Cand* last) {
  if (first == last) return;
  for (Cand* i = first + 1; i != last; ++i) {
    if (cand_gt(i, first)) {
      Cand val = *i;
      memmove(first + 1, first, sizeof(Cand) * (size_t)(i - first));
      *first = val;
    } else {
      std_unguarded_linear_insert(i);
    }
  }
}
static void std_push_heap(Cand* first, long hole, long top, Cand value) {
  long parent = (hole - 1) / 2;
  while (hole > top && cand_gt(first + parent, &value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}
static void std_adjust_heap(Cand* first, long hole, long len, Cand value) {
  const long top = hole;
  long child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (cand_gt(first + child, first + (child - 1))) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  std_push_heap(first, hole, top, value);
}
static void std_heap_sort(Cand* first, Cand* last) { /* __partial_sort(first, last, last) */
  const long len = last - first;
  if (len >= 2)
    for (long parent = (len - 2) / 2;; --parent) { /* __make_heap */
      std_adjust_heap(first, parent, len, first[parent]);
      if (parent == 0) break;
    }
  while (last - first > 1) { /* __sort_heap */
    --last;
    Cand value = *last;
    *last = *first;
    std_adjust_heap(first, 0, last - first, value);
  }
}
static void std_move_median_to_first(Cand* result, Cand* a, Cand* b, Cand* c) {
  if (cand_gt(a, b)) {
    if (cand_gt(b, c)) cand_swap(result, b);
    else if (cand_gt(a, c)) cand_swap(result, c);
    else cand_swap(result, a);
  } else if (cand_gt(a, c)) cand_swap(result, a);
  else if (cand_gt(b, c)) cand_swap(result, c);
  else cand_swap(result, b);
}
static Cand* std_unguarded_partition(Cand* first, Cand* last, Cand* pivot) {
  for (;;) {
    while (cand_gt(first, pivot)) ++first;
    --last;
    while (cand_gt(pivot, last)) --last;
    if (!(first < last)) return first;
    cand_swap(first, last);
    ++first;
  }
}
static void std_introsort_loop(Cand* first, Cand* last, long depth_limit) {
  while (last - first > 16) {
    if (depth_limit == 0) {
      std_heap_sort(first, last);
      return;
    }
    --depth_limit;
    Cand* mid = first + (last - first) / 2;
    std_move_median_to_first(first, first + 1, mid, last - 1);
    Cand* cut = std_unguarded_partition(first + 1, last, first);
    std_introsort_loop(cut, last, depth_limit);
    last = cut;
  }
}
static void std_sort_desc(Cand* first, Cand* last) {
  if (first == last) return;
  long n = last - first, lg = 0;
  while ((n >> (lg + 1)) > 0) ++lg;
  std_introsort_loop(first, last, lg * 2);
  if (last - first > 16) {
    std_insertion_sort(first, first + 16);
    for (Cand* i = first + 16; i != last; ++i) std_unguarded_linear_insert(i);
  } else {
    std_insertion_sort(first, last);
  }
}

/*
 * joint_list: float32 [P][5] rows (x, y, score, id(ignored), part) as handed to
 * process_paf (PEAKS macro, pafprocess.cpp:6).  paf: dense HWC [h][w][38] at the
 * network resolution; `up` = up-sampling factor the reference applied before the
 * call (1 = the map is already at peak resolution); h1 = heatmap_upsamp rows.
 * Outputs: human_parts [max_humans][18] (cid or -1), human_score, and
 * line_xys [P][3] = (x, y, score-as-bits) of peak_infos_line for the getters.
 * Returns the number of humans, -1 if more than max_humans, -2 on bad part id.
 * Equal-score candidates: see had_ties below.
 */
int oracle_process_paf(const float* joint_list, int P, const float* paf, int h, int w, int up, int h1,
                       int max_humans, int* human_parts, float* human_score, int* line_x, int* line_y,
                       float* line_score, int* had_ties, int sort_mode) {
  /* sort_mode 1: order candidates exactly as libstdc++'s std::sort would (the product's
   * contract since round 5: the GPU kernel replays the same moves on a limb with a tie);
   * sort_mode 0: ties keep (idx1, idx2) order (the contract of rounds 1-4). */
  /* *had_ties (optional, sort_mode 0 only): set when two candidates of one limb have exactly
   * equal scores.  The reference sorts with std::sort (cpp:97, not stable), so which of them
   * wins is decided by the library's introsort; without a tie both modes must agree. */
  if (had_ties) *had_ties = 0;
  /* phase 1 (cpp:24-43) */
  Peak* by_part[NUM_PART];
  int n_part[NUM_PART];
  memset(n_part, 0, sizeof(n_part));
  for (int p = 0; p < NUM_PART; ++p) by_part[p] = (Peak*)malloc(sizeof(Peak) * (P > 0 ? P : 1));
  for (int i = 0; i < P; ++i) {
    const float* r = joint_list + (size_t)i * 5;
    Peak pk;
    pk.id = i;
    pk.x = (int)r[0];
    pk.y = (int)r[1];
    pk.score = r[2];
    const int part = (int)r[4];
    if (part < 0 || part >= NUM_PART) return -2;
    by_part[part][n_part[part]++] = pk;
  }
  Peak* line = (Peak*)malloc(sizeof(Peak) * (P > 0 ? P : 1));
  int nl = 0;
  for (int p = 0; p < NUM_PART; ++p)
    for (int i = 0; i < n_part[p]; ++i) line[nl++] = by_part[p][i];
  for (int i = 0; i < nl; ++i) {
    line_x[i] = line[i].x;
    line_y[i] = line[i].y;
    line_score[i] = line[i].score;
  }
  const double inv_up = 1.0 / (double)up;

  /* phases 2+3 (cpp:46-124) */
  Conn* conns[NUM_PAIR];
  int n_conn[NUM_PAIR];
  for (int pair = 0; pair < NUM_PAIR; ++pair) {
    const int pa = PAIRS[pair][0], pb = PAIRS[pair][1];
    const int nA = n_part[pa], nB = n_part[pb];
    conns[pair] = NULL;
    n_conn[pair] = 0;
    if (nA == 0 || nB == 0) continue;
    Cand* cands = (Cand*)malloc(sizeof(Cand) * nA * nB);
    int nc = 0;
    for (int a = 0; a < nA; ++a)
      for (int b = 0; b < nB; ++b) {
        const Peak A = by_part[pa][a], B = by_part[pb][b];
        float vx = B.x - A.x, vy = B.y - A.y;
        const float norm = (float)sqrt(vx * vx + vy * vy);
        if (norm < 1e-12) continue;
        vx = vx / norm;
        vy = vy / norm;
        const float step_x = (B.x - A.x) / (float)STEP_PAF, step_y = (B.y - A.y) / (float)STEP_PAF;
        float scores = 0.0f;
        int crit1 = 0;
        for (int i = 0; i < STEP_PAF; ++i) {
          const int lx = roundpaf(A.x + i * step_x), ly = roundpaf(A.y + i * step_y);
          int sx = (int)floor(lx * inv_up), sy = (int)floor(ly * inv_up); /* INTER_NEAREST */
          sx = clampi(sx, 0, w - 1);
          sy = clampi(sy, 0, h - 1);
          const float px = paf[((size_t)sy * w + sx) * 38 + PAIRS_NET[pair][0]];
          const float py = paf[((size_t)sy * w + sx) * 38 + PAIRS_NET[pair][1]];
          const float s = vx * px + vy * py;
          scores += s;
          if (s > 0.05f) crit1 += 1;
        }
        const double pen = 0.5 * h1 / norm - 1.0;
        const float crit2 = scores / STEP_PAF + (pen < 0.0 ? pen : 0.0);
        if (crit1 > 6 && crit2 > 0 && sort_mode == 1) {
          cands[nc].idx1 = a;
          cands[nc].idx2 = b;
          cands[nc].score = crit2;
          ++nc;
        } else if (crit1 > 6 && crit2 > 0) {
          /* keep the list sorted by descending score, ties in insertion order */
          int pos = nc;
          if (had_ties)
            for (int t = 0; t < nc; ++t)
              if (cands[t].score == crit2) *had_ties = 1;
          while (pos > 0 && cands[pos - 1].score < crit2) {
            cands[pos] = cands[pos - 1];
            --pos;
          }
          cands[pos].idx1 = a;
          cands[pos].idx2 = b;
          cands[pos].score = crit2;
          ++nc;
        }
      }
    if (sort_mode == 1) std_sort_desc(cands, cands + nc);
    conns[pair] = (Conn*)malloc(sizeof(Conn) * (nA < nB ? nA : nB));
    char* usedA = (char*)calloc(nA, 1);
    char* usedB = (char*)calloc(nB, 1);
    for (int c = 0; c < nc; ++c) {
      if (usedA[cands[c].idx1] || usedB[cands[c].idx2]) continue;
      usedA[cands[c].idx1] = usedB[cands[c].idx2] = 1;
      Conn cn;
      cn.cid1 = by_part[pa][cands[c].idx1].id;
      cn.cid2 = by_part[pb][cands[c].idx2].id;
      cn.score = cands[c].score;
      conns[pair][n_conn[pair]++] = cn;
    }
    free(usedA);
    free(usedB);
    free(cands);
  }

  /* phase 4 (cpp:126-185): rows of 20 floats */
  int cap_rows = P + 1, nrows = 0;
  float* subset = (float*)malloc(sizeof(float) * 20 * cap_rows);
  for (int pair = 0; pair < NUM_PAIR; ++pair) {
    const int p1 = PAIRS[pair][0], p2 = PAIRS[pair][1];
    for (int c = 0; c < n_conn[pair]; ++c) {
      const Conn cn = conns[pair][c];
      int found = 0, i1 = 0, i2 = 0;
      for (int r = 0; r < nrows; ++r)
        if (subset[r * 20 + p1] == cn.cid1 || subset[r * 20 + p2] == cn.cid2) {
          if (found == 0) i1 = r;
          if (found == 1) i2 = r;
          found += 1;
        }
      if (found == 1) {
        float* row = subset + i1 * 20;
        if (row[p2] != cn.cid2) {
          row[p2] = cn.cid2;
          row[19] += 1;
          row[18] += line[cn.cid2].score + cn.score;
        }
      } else if (found == 2) {
        float* r1 = subset + i1 * 20;
        float* r2 = subset + i2 * 20;
        int membership = 0;
        for (int k = 0; k < 18; ++k)
          if (r1[k] > 0 && r2[k] > 0) membership = 2;
        if (membership == 0) {
          for (int k = 0; k < 18; ++k) r1[k] += (r2[k] + 1);
          r1[19] += r2[19];
          r1[18] += r2[18];
          r1[18] += cn.score;
          memmove(r2, r2 + 20, sizeof(float) * 20 * (nrows - i2 - 1));
          --nrows;
        } else {
          r1[p2] = cn.cid2;
          r1[19] += 1;
          r1[18] += line[cn.cid2].score + cn.score;
        }
      } else if (found == 0 && pair < 18) {
        float* row = subset + nrows * 20;
        for (int k = 0; k < 20; ++k) row[k] = -1;
        row[p1] = cn.cid1;
        row[p2] = cn.cid2;
        row[19] = 2;
        row[18] = line[cn.cid1].score + line[cn.cid2].score + cn.score;
        ++nrows;
      }
    }
  }
  /* phase 5 (cpp:187-191) */
  int nh = 0, rc = 0;
  for (int r = 0; r < nrows; ++r) {
    const float* row = subset + r * 20;
    if (row[19] < 4 || row[18] / row[19] < 0.3f) continue;
    if (nh >= max_humans) {
      rc = -1;
      break;
    }
    for (int k = 0; k < 18; ++k) human_parts[nh * 18 + k] = (int)row[k];
    human_score[nh] = row[18] / row[19];
    ++nh;
  }
  free(subset);
  free(line);
  for (int p = 0; p < NUM_PART; ++p) free(by_part[p]);
  for (int pair = 0; pair < NUM_PAIR; ++pair) free(conns[pair]);
  return rc ? rc : nh;
}
