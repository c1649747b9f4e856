// The seven entry points of the reference's SWIG module, same names and
// argument meaning (lib/pafprocess/pafprocess.h:53-59, pafprocess.i:14-15),
// with the arithmetic on the GPU: process_paf uploads the peak list and the
// (already up-sampled, HWC) PAF it is handed, runs the same limb_assign /
// group kernels the batched path uses (inv_up = 1: the map is indexed at full
// resolution exactly as PAF(y,x,c) at pafprocess.cpp:8) and keeps the result
// in process-global state for the getters (pafprocess.cpp:12-13), behind a
// mutex.  There is no CPU fallback: without a device process_paf returns
// RTPOSE_E_NODEVICE / a HIP error code and the getters report an empty result.
//
// This legacy path owns its device scratch (grown on demand, freed at exit):
// the reference API has no way to pass buffers in.
#include <hip/hip_runtime.h>

#include <cmath>
#include <mutex>
#include <vector>

#include "common.h"
#include "decode.h"

using namespace rtpose;

namespace {

struct LegacyState {
  std::mutex mu;
  std::vector<rtpose_peak> line;       // peak_infos_line: part-major
  std::vector<int32_t> human_parts;    // [n][18]
  std::vector<float> human_score;
  int num_humans = 0;
  // device scratch
  void* d_paf = nullptr;
  size_t paf_bytes = 0;
  void* d_res = nullptr;
  size_t res_bytes = 0;
  void* d_ws = nullptr;
  size_t ws_bytes = 0;
};

LegacyState& st() {
  static LegacyState s;
  return s;
}

int grow(void** p, size_t* have, size_t need) {
  if (*have >= need) return 0;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *have = 0;
  RTPOSE_HIP_CHECK(hipMalloc(p, need));
  *have = need;
  return 0;
}

}  // namespace

extern "C" {

int process_paf(int p1, int p2, int p3, float* peaks, int h1, int h2, int h3, float* heatmap, int f1,
                int f2, int f3, float* pafmap) {
  (void)h2;
  (void)h3;
  (void)heatmap;  // never read by the reference either (only h1, pafprocess.cpp:83)
  LegacyState& S = st();
  std::lock_guard<std::mutex> lock(S.mu);
  S.line.clear();
  S.human_parts.clear();
  S.human_score.clear();
  S.num_humans = 0;
  if (p1 < 0 || p2 < 0 || p3 < 5 || !peaks || !pafmap || f1 <= 0 || f2 <= 0 || f3 < 38)
    return fail(RTPOSE_E_INVAL, "process_paf: bad shapes (peaks [%d,%d,%d], paf [%d,%d,%d])", p1, p2, p3,
                f1, f2, f3);

  // pafprocess.cpp:24-43: bucket by part, ids in arrival order
  std::vector<rtpose_peak> by_part[RTPOSE_NUM_PART];
  int cnt = 0;
  for (int i = 0; i < p1; ++i)
    for (int j = 0; j < p2; ++j) {
      const float* r = peaks + ((size_t)i * p2 + j) * p3;
      rtpose_peak pk;
      pk.id = cnt++;
      pk.x = (int)r[0];
      pk.y = (int)r[1];
      pk.score = r[2];
      const int part = (int)r[4];
      if (part < 0 || part >= RTPOSE_NUM_PART)
        return fail(RTPOSE_E_INVAL, "process_paf: part id %d outside [0,17]", part);
      by_part[part].push_back(pk);
    }
  int pcap = 1;
  for (auto& v : by_part) {
    if ((int)v.size() > pcap) pcap = (int)v.size();
    for (auto& pk : v) S.line.push_back(pk);
  }
  if (pcap > kDecodeMaxPeaks)
    return fail(RTPOSE_E_CAPACITY, "process_paf: %d peaks of one part exceed the device table (%d)", pcap,
                kDecodeMaxPeaks);
  if (cnt == 0) return 0;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(RTPOSE_E_NODEVICE, "process_paf: no HIP device (this build has no CPU path)");

  rtpose_decode_cfg cfg;
  cfg.num_keypoints = RTPOSE_NUM_PART;
  cfg.upsample = 1;
  cfg.thresh_heatmap = 0.f;
  cfg.max_peaks_per_part = pcap;
  cfg.max_humans = 64;
  for (;;) {
    const int words = decode_result_words(&cfg);
    std::vector<int32_t> host((size_t)words, 0);
    int total = 0;
    for (int p = 0; p < RTPOSE_NUM_PART; ++p) {
      host[kResPartCount + p] = (int)by_part[p].size();
      total += (int)by_part[p].size();
      if (!by_part[p].empty())
        memcpy(&host[kResPeaks + (size_t)4 * p * pcap], by_part[p].data(),
               by_part[p].size() * sizeof(rtpose_peak));
    }
    host[kResHeader] = total;
    const size_t paf_bytes = (size_t)f1 * f2 * f3 * sizeof(float);
    int rc;
    if ((rc = grow(&S.d_paf, &S.paf_bytes, paf_bytes))) return rc;
    if ((rc = grow(&S.d_res, &S.res_bytes, (size_t)words * 4))) return rc;
    if ((rc = grow(&S.d_ws, &S.ws_bytes, decode_workspace_bytes(&cfg, 1)))) return rc;
    RTPOSE_HIP_CHECK(hipMemcpy(S.d_paf, pafmap, paf_bytes, hipMemcpyHostToDevice));
    RTPOSE_HIP_CHECK(hipMemcpy(S.d_res, host.data(), (size_t)words * 4, hipMemcpyHostToDevice));
    rtpose_layout lp;
    lp.cstride = f3;
    lp.choff = 0;
    lp.ws = f2;
    lp.hs = f1;
    lp.lead = 0;
    rc = assign_group_launch(static_cast<const float*>(S.d_paf), &lp, 1, f1, f2, 1.0, h1, &cfg, S.d_ws,
                             S.ws_bytes, S.d_res, nullptr, /*write_ids=*/false);
    if (rc) return rc;
    RTPOSE_HIP_CHECK(hipMemcpy(host.data(), S.d_res, (size_t)words * 4, hipMemcpyDeviceToHost));
    if (host[kResHeader + 2] & kOverflowHumans) {
      if (cfg.max_humans >= 16384) return fail(RTPOSE_E_CAPACITY, "process_paf: too many candidate persons");
      cfg.max_humans *= 2;
      continue;
    }
    S.num_humans = host[kResHeader + 1];
    const int32_t* hp = &host[kResPeaks + (size_t)4 * RTPOSE_NUM_PART * pcap];
    const float* hs = reinterpret_cast<const float*>(hp + (size_t)RTPOSE_NUM_PART * cfg.max_humans);
    S.human_parts.assign(hp, hp + (size_t)S.num_humans * RTPOSE_NUM_PART);
    S.human_score.assign(hs, hs + S.num_humans);
    return 0;
  }
}

int get_num_humans(void) {
  LegacyState& S = st();
  std::lock_guard<std::mutex> lock(S.mu);
  return S.num_humans;
}

int get_part_cid(int human_id, int part_id) {
  LegacyState& S = st();
  std::lock_guard<std::mutex> lock(S.mu);
  if (human_id < 0 || human_id >= S.num_humans || part_id < 0 || part_id >= RTPOSE_NUM_PART) return -1;
  return S.human_parts[(size_t)human_id * RTPOSE_NUM_PART + part_id];
}

float get_score(int human_id) {
  LegacyState& S = st();
  std::lock_guard<std::mutex> lock(S.mu);
  if (human_id < 0 || human_id >= S.num_humans) return NAN;
  return S.human_score[human_id];
}

int get_part_x(int cid) {
  LegacyState& S = st();
  std::lock_guard<std::mutex> lock(S.mu);
  if (cid < 0 || cid >= (int)S.line.size()) return -1;
  return S.line[cid].x;
}

int get_part_y(int cid) {
  LegacyState& S = st();
  std::lock_guard<std::mutex> lock(S.mu);
  if (cid < 0 || cid >= (int)S.line.size()) return -1;
  return S.line[cid].y;
}

float get_part_score(int cid) {
  LegacyState& S = st();
  std::lock_guard<std::mutex> lock(S.mu);
  if (cid < 0 || cid >= (int)S.line.size()) return NAN;
  return S.line[cid].score;
}

}  // extern "C"
