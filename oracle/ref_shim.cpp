// extern "C" doorway into the reference's pafprocess.cpp (compiled unmodified by
// oracle/Makefile; SWIG is absent from the image).  Test infrastructure only.
#include "pafprocess.h"  // from /root/reference/lib/pafprocess via -I

extern "C" {
int ref_process_paf(int p1, int p2, int p3, float* peaks, int h1, int h2, int h3, float* heatmap, int f1,
                    int f2, int f3, float* pafmap) {
  return process_paf(p1, p2, p3, peaks, h1, h2, h3, heatmap, f1, f2, f3, pafmap);
}
int ref_get_num_humans() { return get_num_humans(); }
int ref_get_part_cid(int h, int p) { return get_part_cid(h, p); }
float ref_get_score(int h) { return get_score(h); }
int ref_get_part_x(int c) { return get_part_x(c); }
int ref_get_part_y(int c) { return get_part_y(c); }
float ref_get_part_score(int c) { return get_part_score(c); }
}
