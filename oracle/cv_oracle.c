/*
 * ORACLE — test infrastructure, NOT product code.
 *
 * cv2.resize(img, None, fx=s, fy=s) for uint8 HxWxC images with the default
 * interpolation (INTER_LINEAR), the call lib/network/im_transform.py:126 makes.
 * OpenCV is a third-party dependency that is absent from this image and unpinned
 * by the reference (requirements.txt does not list it); this file restates the
 * published algorithm of imgproc/resize.cpp (3.x / 4.x generic path):
 *
 *   dsize      = (cvRound(w * fx), cvRound(h * fy));  scale = 1 / fx  (double)
 *   per dst x  : f = (float)((dx + 0.5) * scale - 0.5); sx = floor(f); f -= sx;
 *                sx < 0            -> f = 0, sx = 0
 *                sx >= src_w - 1   -> f = 0, sx = src_w - 1 (columns >= xmax copy S[sx] * 2048)
 *                ialpha = saturate_cast<short>((1 - f) * 2048), saturate_cast<short>(f * 2048)
 *                (cvRound = round half to even)
 *   per dst y  : same f / sy, no clamping of f: the two source ROWS are clipped instead
 *   HResizeLinear : D[dx] = S[sx] * a0 + S[sx + 1] * a1                       (int, x2048)
 *   VResizeLinear : dst = ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
 *   exactly-2x decimation switches to INTER_AREA (resize.cpp: "INTER_AREA (fast) also is
 *   equal to INTER_LINEAR" for scale 2) = (a + b + c + d + 2) >> 2, which the formulas
 *   above already produce (f = 0.5 on both axes), so no separate branch is needed.
 *
 * PARITY UNPINNED against a real OpenCV build (none available offline); the tests
 * cross-check it against float bilinear interpolation (<= 1 LSB) only.
 */
#include <math.h>
#include <stdlib.h>

static int cv_round_d(double v) { return (int)lrint(v); } /* default FE mode: half to even */
static short sat_short_f(float v) {
  long r = lrintf(v);
  return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}
static int clipi(int v, int lo, int hi) { return v < lo ? lo : (v >= hi ? hi - 1 : v); }

void oracle_resize_dsize(int h, int w, double fx, double fy, int* dh, int* dw) {
  *dw = cv_round_d(w * fx);
  *dh = cv_round_d(h * fy);
}

/* src [h][w][c] uint8 -> dst [dh][dw][c] uint8, dh/dw from oracle_resize_dsize */
int oracle_resize_linear_u8(const unsigned char* src, int h, int w, int c, double fx, double fy,
                            unsigned char* dst) {
  int dh, dw;
  oracle_resize_dsize(h, w, fx, fy, &dh, &dw);
  if (dh <= 0 || dw <= 0) return -1;
  const double scale_x = 1.0 / fx, scale_y = 1.0 / fy;
  int* xofs = (int*)malloc(sizeof(int) * dw);
  short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
  int xmax = dw;
  for (int dx = 0; dx < dw; ++dx) {
    float f = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(f);
    f -= sx;
    if (sx < 0) {
      f = 0;
      sx = 0;
    }
    if (sx + 1 >= w) {
      if (dx < xmax) xmax = dx;
      if (sx >= w - 1) {
        f = 0;
        sx = w - 1;
      }
    }
    xofs[dx] = sx;
    ialpha[2 * dx] = sat_short_f((1.f - f) * 2048.f);
    ialpha[2 * dx + 1] = sat_short_f(f * 2048.f);
  }
  int* rows[2];
  rows[0] = (int*)malloc(sizeof(int) * (size_t)dw * c);
  rows[1] = (int*)malloc(sizeof(int) * (size_t)dw * c);
  for (int dy = 0; dy < dh; ++dy) {
    float f = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(f);
    f -= sy;
    const short b0 = sat_short_f((1.f - f) * 2048.f), b1 = sat_short_f(f * 2048.f);
    for (int k = 0; k < 2; ++k) {
      const unsigned char* S = src + (size_t)clipi(sy + k, 0, h) * w * c;
      int* D = rows[k];
      for (int dx = 0; dx < dw; ++dx)
        for (int ch = 0; ch < c; ++ch) {
          const int sx = xofs[dx];
          if (dx < xmax)
            D[dx * c + ch] = S[sx * c + ch] * ialpha[2 * dx] + S[(sx + 1) * c + ch] * ialpha[2 * dx + 1];
          else
            D[dx * c + ch] = S[sx * c + ch] * 2048;
        }
    }
    unsigned char* o = dst + (size_t)dy * dw * c;
    for (int i = 0; i < dw * c; ++i) {
      int v = (((b0 * (rows[0][i] >> 4)) >> 16) + ((b1 * (rows[1][i] >> 4)) >> 16) + 2) >> 2;
      o[i] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
  free(rows[0]);
  free(rows[1]);
  free(xofs);
  free(ialpha);
  return 0;
}
