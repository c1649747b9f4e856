// Layout of the decode result block and scratch (shared by host and device code).
#pragma once
#include "common.h"

namespace rtpose {

constexpr int kDecodeMaxPeaks = 1024;  // per (image, part) table capacity limit
constexpr int kLdsPairs = 110 * 110;   // candidate-score matrix entries that fit in LDS
constexpr int kLdsRows = 720;          // subset rows (21 floats each) that fit in LDS
constexpr int kTieLdsCands = 4096;     // candidates (8 bytes each) of a limb that replays std::sort in LDS

// word (4-byte) offsets inside one image's result record
constexpr int kResHeader = 0;      // [0] n_peaks [1] n_humans [2] overflow flags [3] max_peaks_per_part [4] max_humans
constexpr int kResPartCount = 8;   // int32[18]
constexpr int kResPeaks = 32;      // rtpose_peak[18 * pcap], then human tables

constexpr int kOverflowPeaks = 1;   // a part had more than max_peaks_per_part peaks
constexpr int kOverflowHumans = 2;  // more subset rows / humans than capacity

__host__ __device__ inline int decode_result_words(const rtpose_decode_cfg* c) {
  const int w = kResPeaks + 4 * RTPOSE_NUM_PART * c->max_peaks_per_part +
                (RTPOSE_NUM_PART + 1) * c->max_humans;
  return (w + 3) & ~3;
}
// per image: 19 x { count, (a, b, score) x pcap }
inline int decode_conn_words(const rtpose_decode_cfg* c) {
  return RTPOSE_NUM_LIMB * (1 + 3 * c->max_peaks_per_part);
}
// subset rows alive at any time before pruning (21 floats each); LDS resident up to
// kLdsRows, in the global workspace beyond
inline int decode_row_cap(const rtpose_decode_cfg* c) {
  int r = 2 * c->max_humans;
  if (r < 64) r = 64;
  return r;
}
// workspace: [conn lists][candidate-score matrices when they exceed LDS][subset rows when
// they exceed LDS][candidate lists of the limbs that replay std::sort on an exact score tie]
inline size_t decode_ws_conn_bytes(const rtpose_decode_cfg* c, int N) {
  return round_up((size_t)N * decode_conn_words(c) * sizeof(int32_t), 256);
}
inline size_t decode_ws_score_bytes(const rtpose_decode_cfg* c, int N) {
  const size_t p = (size_t)c->max_peaks_per_part;
  return p * p > (size_t)kLdsPairs ? round_up((size_t)N * RTPOSE_NUM_LIMB * p * p * sizeof(float), 256) : 0;
}
inline size_t decode_ws_rows_bytes(const rtpose_decode_cfg* c, int N) {
  const int r = decode_row_cap(c);
  return r > kLdsRows ? round_up((size_t)N * r * 21 * sizeof(float), 256) : 0;
}
// (a limb's candidate list is at most p * p long and stays in LDS up to kTieLdsCands entries: at the default capacities -
//  p = 32 ... 64 - nothing is reserved; at p = 1024 this was 5.1 GB for 32 images whatever the maps held)
inline size_t decode_ws_tie_bytes(const rtpose_decode_cfg* c, int N) {
  const size_t p = (size_t)c->max_peaks_per_part;
  if (p * p <= (size_t)kTieLdsCands) return 0;
  return round_up((size_t)N * RTPOSE_NUM_LIMB * p * p * sizeof(unsigned long long), 256);
}
inline size_t decode_workspace_bytes(const rtpose_decode_cfg* c, int N) {
  return decode_ws_conn_bytes(c, N) + decode_ws_score_bytes(c, N) + decode_ws_rows_bytes(c, N) +
         decode_ws_tie_bytes(c, N);
}

// with_ids: also run peak_prefix_kernel (the running peak ids + the peak total); a full decode leaves that to
// assign_group_launch(write_ids = true), which writes them in its grouping kernel
int nms_launch(const float* heat, const rtpose_layout* lheat, int N, int h, int w,
               const rtpose_decode_cfg* cfg, void* result, hipStream_t s, int flags, bool with_ids);
// write_ids = false: the peak tables already carry the caller's ids (legacy process_paf: ids of the caller's joint list)
int assign_group_launch(const float* paf, const rtpose_layout* lpaf, int N, int h, int w, double inv_up,
                        int h1, const rtpose_decode_cfg* cfg, void* workspace, size_t workspace_bytes,
                        void* result, hipStream_t s, bool write_ids);

}  // namespace rtpose
