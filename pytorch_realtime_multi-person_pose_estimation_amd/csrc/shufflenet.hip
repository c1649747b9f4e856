// Native executor for the ShuffleNetV2 x1.0 pose network
// (lib/network/rtpose_shufflenetV2.py: BasicBlock :22-63, Network :80-148) —
// BASELINE config 4, the small-model / memory-bound path.
//
// Same design rules as net.hip: fixed launch list for (N,H,W), caller-owned workspace
// and weight arena, shared-gap padded NHWC activations.  Specific to this network:
//  * BatchNorm (eval) is folded into the conv weights/bias by the host; only the input
//    BatchNorm2d(3) (:96) stays an explicit per-channel affine, applied while converting
//    NCHW -> NHWC (it sits in front of a zero-padded conv, so it cannot be folded).
//  * torch.cat + channel_shuffle(2) (:56-62) never run: the two producers of a block
//    write their channels directly at the interleaved (even / odd) positions of the next
//    buffer - the 1x1 convs through the conv kernel's out_cmap, the pass-through half
//    through one strided copy.
//  * Stage buffers keep each half padded to a multiple of 8 channels ([h | pad | h | pad])
//    so both halves are 16-byte aligned slices for the MFMA 1x1 kernel; logical channel j
//    lives at physical f(j) = j < h ? j : hp + j - h.  Reading a whole such buffer (first
//    block of the next stage) goes through cin_map at weight-pack time.
//  * `downsample` leaks out of stage 1 in the reference (:113-117): the first block of
//    BOTH stride-1 stages is the two-branch block.  Reproduced.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "common.h"

namespace rtpose {
int conv2d_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, hipStream_t s);
int pack_weights_launch(const float* w, const float* bias, int cout, int cin_src, int k,
                        const int32_t* cin_map, int cin_packed, float* wp, float* bp, hipStream_t s);
// bf16 plans (conv_mfma_bf16.hip)
int conv2d_bf16_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, int out_f32, int split,
                       hipStream_t s);
int pack_weights_bf16_launch(const float* w, const float* bias, int cout, int cin_src, int k,
                             const int32_t* cin_map, int cin_packed, void* wp, float* bp, int split,
                             hipStream_t s);

// fused pointwise chains of the fp32 plan (pw_fused.hip)
int pw_fused_launch(const rtpose_pw_desc* d, int N, int H, int W, hipStream_t s);
int pack_pw_launch(const float* w, const float* bias, int cout, int cin_src, const int32_t* cin_map, int K,
                   int coutp, int col_off, float* wp, float* bp, hipStream_t s);
int pw_halo_stride(const rtpose_layout& l, int H, int W);
// conv5 + the two heads as one back-to-back launch (pw_head.hip)
int pw_head_launch(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2, int N, int H, int W, hipStream_t s);
int pw_zero_columns_launch(float* wp, float* bp, int K, int coutp, int c0, int c1, hipStream_t s);
// ... and of the bf16 plan (pw_head_bf16.hip)
int pw_head_bf16_launch(const rtpose_pw_desc* d1, const rtpose_pw_desc* d2, int N, int H, int W, hipStream_t s);
int pack_head_w2_bf16_launch(const float* w, const float* bias, int cout, int K, int col_off, void* wp, float* bp,
                             hipStream_t s);
int zero_head_columns_bf16_launch(void* wp, float* bp, int K, int c0, int c1, hipStream_t s);
// conv.0 -> depthwise -> conv.2 of a stride-1 unit as one launch (unit_bf16.hip)
int unit_bf16_fits(const rtpose_pw_desc* d0, const rtpose_pw_desc* d2, int H, int W);
int unit_bf16_launch(const rtpose_pw_desc* d0, const rtpose_pw_desc* d2, int N, int H, int W, hipStream_t s);
// column-mapped fp32 packing (pw_fused.hip): a layer's columns in the memory order of the runs it writes
int pack_pw_cols_launch(const float* w, const float* bias, int cout, int cin_src, const int32_t* cin_map, int K,
                        int ncols, const int32_t* col_map, int coutp, int col_off, float* wp, float* bp,
                        hipStream_t s);
// ... and of the bf16 plan (pw_fused_bf16.hip)
int pw_fused_bf16_launch(const rtpose_pw_desc* d, int out_f32, int N, int H, int W, hipStream_t s);
int pack_pw_bf16_launch(const float* w, const float* bias, int cout, int cin_src, const int32_t* cin_map, int K,
                        int ncols, const int32_t* col_map, int coutp, int col_off, void* wp, float* bp,
                        hipStream_t s);

// w[C][1][3][3] (+bias[C]) -> wp[9][cphys], bp[cphys]; phys channel p reads logical pmap[p] (-1: zero)
__global__ void pack_dw_kernel(const float* __restrict__ w, const float* __restrict__ b, int C,
                               const int32_t* __restrict__ pmap, int cphys, float* __restrict__ wp,
                               float* __restrict__ bp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 10 * cphys) return;
  const int t = i / cphys, p = i - t * cphys;
  const int c = pmap ? pmap[p] : (p < C ? p : -1);
  if (t < 9)
    wp[i] = (c >= 0 && c < C) ? w[c * 9 + t] : 0.f;
  else
    bp[p] = (c >= 0 && c < C && b) ? b[c] : 0.f;
}

// w[cout][cin][3][3] -> wp[ky][kx][cin_pad][cout]
__global__ void pack_stem_kernel(const float* __restrict__ w, int cout, int cin, int cin_pad,
                                 float* __restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * cin_pad * cout) return;
  const int co = i % cout;
  int r = i / cout;
  const int c = r % cin_pad;
  const int t = r / cin_pad;
  wp[i] = c < cin ? w[((size_t)co * cin + c) * 9 + t] : 0.f;
}
}  // namespace rtpose

using namespace rtpose;

namespace {

enum LKind { L_AFFINE, L_STEM, L_DW, L_PW };

struct SLayer {
  LKind kind;
  std::string name;
  int cout = 0, cin = 0;      // logical
  int cin_packed = 0;         // L_PW: packed input channels; L_DW: physical channels
  int map_id = -1;            // index into maps (cin_map for PW, phys->logical for DW), -1 identity
  size_t w_off = 0, b_off = 0;
  int coutp = 0, col_off = 0; // L_PW of a fused plan: columns of the packed matrix this layer's cout columns
                              // start at (the two heads share one 64-column matrix)
  int colmap_id = -1, ncols = 0;  // fused bf16 plans: packed column i is output channel maps[colmap_id][i]
                              // (-1: identity), ncols columns are packed / stored
  int zero_c0 = 0, zero_c1 = 0;   // columns of a shared packed matrix that no layer writes and this layer's load zeroes
};

struct SBuf {
  size_t off = 0;
  rtpose_layout lay{};
  int C = 0, H = 0, W = 0;
};

enum OKind { O_INPUT, O_STEM, O_POOL3, O_DW, O_PWF, O_STEMPOOL, O_HEAD, O_UNIT };

struct SOp {
  OKind kind;
  std::string name;
  int H = 0, W = 0;            // input spatial size of the op
  int ngroups = 1;
  int layer[2] = {-1, -1};
  int in_buf[2] = {-1, -1}, in_choff[2] = {0, 0};
  int out_buf[2] = {-1, -1}, out_choff[2] = {0, 0};
  int cmap[2] = {-1, -1};      // map ids (out_cmap / copy map)
  int relu = 0, stride = 1, C = 0;
  double flops = 0;
  // O_PWF (pw_fused.hip): layer[0] = the pointwise layer, dw_layer = the depthwise 3x3 evaluated in front
  // of it inside the kernel (-1: none), pt_buf / pt_map = pass-through half copied by the same launch
  int dw_layer = -1, pt_buf = -1, pt_map = -1, pt_c = 0, cout_store = 0;
  int planes_map = -1;          // input plane gather table (map id), -1: contiguous slice
  int pt_pairs = 0, pt_a = 0, pt_b = 0, pt_split = 0, pt_d0 = 0, pt_d1 = 0;  // interleave pass-through
};

struct Map {
  std::vector<int32_t> v;
  size_t off = 0;  // float-slot offset in the weight arena
};

}  // namespace

struct rtpose_shufflenet {
  int N = 0, H = 0, W = 0, Hm = 0, Wm = 0;  // Hm x Wm: stride-8 maps
  int bf16 = 0;  // 1: 2-byte activations + bf16 pointwise weights (fp32 accumulate); outputs stay fp32
  int device = -1;      // HIP device that owns the bound arenas
  CheckedPtr in_checked;
  std::vector<SBuf> bufs;
  std::vector<SLayer> layers;
  std::vector<SOp> ops;
  std::vector<Map> maps;
  size_t ws_floats = 0, wt_floats = 0;
  float* ws = nullptr;
  float* wt = nullptr;
  bool bound = false;
  int out_buf = -1;
  int profiling = 0;
  std::vector<hipEvent_t> ev;
  bool ev_valid = false;
};

namespace {

int add_buf(rtpose_shufflenet* n, int C, int P, int H, int W, bool f32 = false) {
  SBuf b;
  b.C = C;
  b.H = H;
  b.W = W;
  b.lay.cstride = C;
  b.lay.choff = 0;
  b.lay.ws = W + P;
  b.lay.hs = H + P;
  b.lay.lead = P * (W + P) + P;
  b.off = n->ws_floats;
  const size_t per_px = (n->bf16 && !f32) ? (size_t)C / 2 : (size_t)C;  // bf16 plans: 2-byte elements
  n->ws_floats += round_up(rtpose_layout_pixels(&b.lay, n->N, H, W) * per_px, 64);
  n->bufs.push_back(b);
  return (int)n->bufs.size() - 1;
}

int add_map(rtpose_shufflenet* n, const std::vector<int32_t>& v) {
  Map m;
  m.v = v;
  m.off = n->wt_floats;
  n->wt_floats += round_up(v.size(), 64);
  n->maps.push_back(m);
  return (int)n->maps.size() - 1;
}

int add_layer(rtpose_shufflenet* n, LKind kind, const std::string& name, int cout, int cin, int cin_packed,
              int map_id) {
  SLayer l;
  l.kind = kind;
  l.name = name;
  l.cout = cout;
  l.cin = cin;
  l.cin_packed = cin_packed;
  l.map_id = map_id;
  l.w_off = n->wt_floats;
  size_t wf = 0, bf = 0;
  switch (kind) {
    case L_AFFINE: wf = 64; bf = 64; break;
    case L_STEM: wf = (size_t)9 * 8 * cout; bf = cout; break;
    case L_DW: wf = (size_t)9 * cin_packed; bf = cin_packed; break;
    case L_PW:
      // (fused bf16 plans: a layer that writes whole slot groups of the stage buffer packs up to 256 columns -
      //  116 channels in 18 groups of 8 = 144 columns run in the 256-column instance)
      wf = n->bf16 ? (size_t)(cin_packed + 64) * (cout_pad(cout > 0 ? cout : 1) < 256 ? 256 : cout_pad(cout)) / 2
                   : (size_t)(cin_packed + 32) * cout_pad(cout + 8);
      bf = (n->bf16 && cout_pad(cout) < 256) ? 256 : rtpose_packed_bias_floats(cout);
      break;
  }
  n->wt_floats += round_up(wf, 64);
  l.b_off = n->wt_floats;
  n->wt_floats += round_up(bf, 64);
  n->layers.push_back(l);
  return (int)n->layers.size() - 1;
}


void add_dw(rtpose_shufflenet* n, const std::string& name, int H, int W, int layer, int in_buf, int out_buf,
            int stride) {
  SOp o;
  o.kind = O_DW;
  o.name = name;
  o.H = H;
  o.W = W;
  o.layer[0] = layer;
  o.in_buf[0] = in_buf;
  o.out_buf[0] = out_buf;
  o.stride = stride;
  const SLayer& l = n->layers[layer];
  o.C = l.cin_packed;
  o.flops = 2.0 * n->N * ((H - 1) / stride + 1) * ((W - 1) / stride + 1) * (double)l.cin * 9;
  n->ops.push_back(o);
}

// one fused launch: [depthwise 3x3 ->] pointwise (+ReLU) [+ pass-through half]
void add_pwf(rtpose_shufflenet* n, const std::string& name, int H, int W, int layer, int dw_layer, int in_buf,
             int in_choff, int out_buf, int out_choff, int cmap, int relu, int pt_buf = -1, int pt_map = -1,
             int pt_c = 0, int cout_store = 0) {
  SOp o;
  o.kind = O_PWF;
  o.name = name;
  o.H = H;
  o.W = W;
  o.layer[0] = layer;
  o.dw_layer = dw_layer;
  o.in_buf[0] = in_buf;
  o.in_choff[0] = in_choff;
  o.out_buf[0] = out_buf;
  o.out_choff[0] = out_choff;
  o.cmap[0] = cmap;
  o.relu = relu;
  o.pt_buf = pt_buf;
  o.pt_map = pt_map;
  o.pt_c = pt_c;
  const SLayer& l = n->layers[layer];
  o.cout_store = cout_store ? cout_store : l.cout;
  o.flops = 2.0 * n->N * H * W * (double)l.cout * l.cin;
  if (dw_layer >= 0) o.flops += 2.0 * n->N * H * W * (double)n->layers[dw_layer].cin * 9;
  n->ops.push_back(o);
}

// ---- zero-copy channel shuffle: slot plan of one stage ---------------------------------------------------------------
// torch.cat((x1, conv(x2)), 1) + channel_shuffle(2) (rtpose_shufflenetV2.py:56-62) moves no data in the fused plans.
// A stage has ONE buffer of P physical channel slots that all its units work on in place:
//   * the pass-through half x1 of a unit stays where it is - only its LOGICAL index changes (L_u[2i] = L_{u-1}[i]);
//   * the processed half y = conv(x2) is written into slots x2 just vacated (x2 is dead once conv.0 has read it, and
//     conv.0 is an earlier launch), through the pointwise kernel's column -> channel map;
//   * the next unit's x2 = L_u[h:2h] is wherever those logical channels live: the GEMM gathers K as 16-byte planes
//     (in_planes) with the weights permuted to match.
// Which slots a new channel takes is decided by WHEN it will be consumed: a channel born at logical position p of L_u
// enters x2 at unit u + 1 + min{t : p 2^t >= h}.  Channels of one producer with the same consumption time form a
// CLASS that occupies whole groups of G slots (G = one 16-byte plane: 4 fp32 / 8 bf16 channels); a unit frees exactly
// the class sizes it creates (h/2, h/4, ...), so the classes recycle each other's groups, every x2 is a handful of
// contiguous runs of whole planes and K grows from h by the class padding only (116 -> 128 fp32 / 144 bf16 in stage 3).
// Round 3 copied x1 into the next buffer in every unit: 3.6 GB (fp32) of HBM traffic per 128-image forward.
struct ZcUnit {
  std::vector<int32_t> planes;  // K / G entries: channel offset of every plane conv.0 gathers (K padded to kalign)
  std::vector<int32_t> x2map;   // K entries: packed K row -> x2 channel (0 .. h-1), -1: a zero row
  std::vector<int32_t> yslot;   // h entries: slot of y[i]
};
struct ZcStage {
  int P = 0;                    // slots in use (a multiple of G)
  std::vector<int> phys0;       // unit 0: slot of logical channel j < 2h (even j: branch conv0, odd j: branch conv)
  std::vector<ZcUnit> units;    // units 1 .. U-1 (index u - 1)
  std::vector<int> phys_final;  // slot of logical channel j of the stage's output
};

ZcStage zc_plan(int h, int U, int G, int kalign, bool y_beside_x2 = false) {
  const int INF = 1 << 30;
  auto death_from = [&](int pos, int born) {  // the unit that consumes the channel at logical position pos of L_born
    for (int u = born + 1; u < U; ++u) {
      if (pos >= h) return u;
      pos *= 2;
    }
    return INF;
  };
  ZcStage st;
  int ngroups = 0;               // slot groups handed out so far
  std::vector<int> freeg;        // free groups, ascending
  // cnt groups for one class.  A class is read as ONE run by the gather of the unit that consumes it, so it starts on a
  // boundary of min(128 bytes, its own size rounded up to a power of two) - 8 groups are a 128-byte line: a 64-channel
  // bf16 class is exactly one line of every pixel, not two halves (the buffer's pixel pitch is a multiple of 128 bytes).
  // The smallest free aligned run that fits, else new groups at the (aligned) end.
  auto alloc = [&](int cnt) {
    int al = 1;
    while (al < cnt && al < 8) al *= 2;
    std::vector<int> got;
    int best0 = -1, bestlen = 1 << 30;
    for (size_t i = 0; i < freeg.size();) {
      size_t j = i;
      while (j + 1 < freeg.size() && freeg[j + 1] == freeg[j] + 1) ++j;
      // first aligned start inside the run [freeg[i], freeg[j]]
      const int start = (freeg[i] + al - 1) / al * al;
      const int len = freeg[j] - start + 1;
      if (len >= cnt && (int)(j - i + 1) < bestlen) {
        best0 = start;
        bestlen = (int)(j - i + 1);
      }
      i = j + 1;
    }
    if (best0 < 0) {
      while (ngroups % al) freeg.push_back(ngroups++);  // (the groups skipped for alignment stay free)
      std::sort(freeg.begin(), freeg.end());
      best0 = ngroups;
      ngroups += cnt;
      for (int g = best0; g < best0 + cnt; ++g) got.push_back(g);
      return got;
    }
    for (int g = best0; g < best0 + cnt; ++g) {
      got.push_back(g);
      freeg.erase(std::find(freeg.begin(), freeg.end(), g));
    }
    return got;
  };
  // place the channels `ids` (already ordered) of one class: whole groups
  auto place = [&](const std::vector<int>& ids, std::vector<int>& slot_of) {
    const std::vector<int> g = alloc(((int)ids.size() + G - 1) / G);
    for (size_t k = 0; k < ids.size(); ++k) slot_of[ids[k]] = g[k / G] * G + (int)(k % G);
  };
  // ---- unit 0: 2h channels, classes = (producer parity, consumption time) ----
  st.phys0.assign(2 * h, -1);
  for (int par = 0; par < 2; ++par) {
    std::vector<int> deaths;
    for (int j = par; j < 2 * h; j += 2) deaths.push_back(death_from(j, 0));
    std::vector<int> uniq(deaths);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    for (int d : uniq) {
      std::vector<int> ids;
      for (int j = par; j < 2 * h; j += 2)
        if (death_from(j, 0) == d) ids.push_back(j);
      place(ids, st.phys0);
    }
  }
  std::vector<int> phys = st.phys0;
  for (int u = 1; u < U; ++u) {
    ZcUnit zu;
    // x2 = L_{u-1}[h:2h): its groups (whole classes die together: every touched group is freed)
    std::vector<int> slot2x2((size_t)ngroups * G, -1);
    for (int j = h; j < 2 * h; ++j) slot2x2[phys[j]] = j - h;
    std::vector<int> touched;
    for (int g = 0; g < ngroups; ++g) {
      bool any = false;
      for (int e = 0; e < G; ++e) any = any || slot2x2[g * G + e] >= 0;
      if (any) touched.push_back(g);
    }
    for (int g : touched) {
      zu.planes.push_back(g * G);
      for (int e = 0; e < G; ++e) zu.x2map.push_back(slot2x2[g * G + e]);
    }
    while (zu.x2map.size() % kalign) {  // (zero-weight repeats of the first plane)
      zu.planes.push_back(zu.planes[0]);
      for (int e = 0; e < G; ++e) zu.x2map.push_back(-1);
    }
    // a unit that runs as ONE launch (conv.0 -> depthwise -> conv.2, unit_bf16.hip) reads x2 with a halo while other
    // blocks of the same launch store y: y must not land in the slots x2 leaves - they are freed only afterwards
    if (!y_beside_x2) {
      for (int g : touched) freeg.push_back(g);
      std::sort(freeg.begin(), freeg.end());
    }
    // y_u: classes by consumption time, soonest first
    zu.yslot.assign(h, -1);
    std::vector<int> uniq;
    for (int i = 0; i < h; ++i) uniq.push_back(death_from(2 * i + 1, u));
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    for (int d : uniq) {
      std::vector<int> ids;
      for (int i = 0; i < h; ++i)
        if (death_from(2 * i + 1, u) == d) ids.push_back(i);
      place(ids, zu.yslot);
    }
    if (y_beside_x2) {
      for (int g : touched) freeg.push_back(g);
      std::sort(freeg.begin(), freeg.end());
    }
    std::vector<int> np(2 * h);
    for (int i = 0; i < h; ++i) {
      np[2 * i] = phys[i];
      np[2 * i + 1] = zu.yslot[i];
    }
    phys.swap(np);
    st.units.push_back(zu);
  }
  st.P = ngroups * G;
  st.phys_final = phys;
  return st;
}

// the columns a producer packs / stores when its channels `slot_of` (slot of its channel i) must be written as WHOLE
// slot groups (the bf16 epilogue stores 8 contiguous channels per lane): its groups in ascending slot order; packed column
// c holds the producer's channel cols[c] (-1: a zero column) and lands at absolute channel chan[c]
void zc_columns(const std::vector<int32_t>& slot_of, int G, std::vector<int32_t>* cols, std::vector<int32_t>* chan) {
  std::vector<int> groups;
  for (int32_t sl : slot_of) groups.push_back(sl / G);
  std::sort(groups.begin(), groups.end());
  groups.erase(std::unique(groups.begin(), groups.end()), groups.end());
  cols->assign(groups.size() * G, -1);
  chan->assign(groups.size() * G, -1);
  for (size_t k = 0; k < groups.size(); ++k)
    for (int e = 0; e < G; ++e) (*chan)[k * G + e] = groups[k] * G + e;
  for (size_t i = 0; i < slot_of.size(); ++i) {
    const int g = slot_of[i] / G;
    const size_t k = std::lower_bound(groups.begin(), groups.end(), g) - groups.begin();
    (*cols)[k * G + slot_of[i] % G] = (int32_t)i;
  }
}

void build(rtpose_shufflenet* n) {
  // channel alignment of slices that feed a pointwise conv: 8 floats (fp32 kernel: cin % 8 == 0),
  // 16 elements for bf16 plans (one K = 16 MFMA step)
  // (fp32 fused plans: 16 too - the wave-autonomous head kernel, pw_head.hip, walks K in pairs of 8-channel groups)
  const int al = 16;
  auto up8 = [](int v) { return (v + al - 1) / al * al; };
  const int H0 = n->H, W0 = n->W;
  const int H1 = (H0 - 1) / 2 + 1, W1 = (W0 - 1) / 2 + 1;          // stem 3x3 s2 p1
  const int H2 = (H1 - 3 + 1) / 2 + 1, W2 = (W1 - 3 + 1) / 2 + 1;  // maxpool 3/2 ceil
  const int H3 = (H2 - 1) / 2 + 1, W3 = (W2 - 1) / 2 + 1;          // stage 2 first block s2
  n->Hm = H3;
  n->Wm = W3;

  // ---- stem -----------------------------------------------------------------------
  const int L_aff = add_layer(n, L_AFFINE, "network.0", 3, 3, 8, -1);
  const int L_stem = add_layer(n, L_STEM, "network.1", 24, 3, 8, -1);
  const int X0 = -1;  // (no NHWC staging of the image: the stem conv reads the NCHW input itself)
  // (the stem conv's own 184 x 184 x 24 output does not exist: conv + max-pool are one launch)
  const int X1 = add_buf(n, up8(24), 1, H2, W2);  // (channels past 24 stay zero: a 1x1 conv reads them)
  {
    SOp o;
    o.kind = O_INPUT;
    o.name = "(data/bn: fused into stage1/conv)";
    o.H = H0;
    o.W = W0;
    o.layer[0] = L_aff;
    o.out_buf[0] = X0;
    n->ops.push_back(o);
    SOp s;
    s.kind = O_STEMPOOL;
    s.name = "stage1/conv+pool";
    s.H = H0;
    s.W = W0;
    s.layer[0] = L_stem;
    s.in_buf[0] = X0;
    s.out_buf[0] = X1;
    s.relu = 1;
    s.flops = 2.0 * n->N * H1 * W1 * 24.0 * 27;
    n->ops.push_back(s);
  }

  // ---- stages ------------------------------------------------------------------------
  const int widths[3] = {116, 232, 464};
  const int nblocks[3] = {4, 8, 4};
  int in_buf = X1, in_c = 24;  // previous buffer and its logical channel count
  bool in_is_stage = false;
  std::vector<int32_t> in_pmap;  // previous STAGE buffer: physical channel -> logical channel (-1: nothing lives there)
  // ... as its readers walk it (zero-copy plans): only the planes that hold live channels are gathered - K position k
  // is logical channel in_kmap[k] (-1: a zero row), plane k / G sits at channel in_planes_v[k / G] of the pixel
  std::vector<int32_t> in_kmap, in_planes_v;
  int Hc = H2, Wc = W2;
  for (int si = 0; si < 3; ++si) {
    const int C = widths[si], h = C / 2;
    const int stride = si == 0 ? 2 : 1;
    const int Ho = si == 0 ? H3 : Hc, Wo = si == 0 ? W3 : Wc;
    const std::string sp = "network." + std::to_string(3 + si) + ".";
    {
      // ================= ZERO-COPY channel shuffle (round 4), see zc_plan above =========================
      const int U = nblocks[si];
      const int in_phys = in_is_stage ? (int)in_kmap.size() : in_c;   // K of the layers that read the whole input buffer
      const int G = n->bf16 ? 8 : 4;                      // channels per 16-byte plane
      // bf16 plans run the units 1 .. U-1 of the 116- and 232-channel stages as ONE launch each (unit_bf16.hip): y next
      // to x2, not in its place.  (The 58-channel stage stays at two launches: 0.067 against 0.083 ms per unit - the
      // one-launch kernel's tiles are 128 x 128-column GEMMs whatever the width.)
      const bool one_launch_units = n->bf16 != 0 && h >= 100;
      const ZcStage zs = zc_plan(h, U, G, n->bf16 ? 16 : 8, one_launch_units);
      // physical channels of the stage buffer: a pixel is a whole number of 128-byte lines (and a multiple of 16 channels:
      // conv5 / the next stage read all of it)
      const int Pp = (zs.P + 8 * G - 1) / (8 * G) * (8 * G);
      const int S = add_buf(n, Pp, 1, Ho, Wo);
      const int Kt = up8(h);     // channels of the unit temporaries (conv.0 -> depthwise -> conv.2)
      const int T0 = add_buf(n, up8(in_phys), 0, Ho, Wo);  // conv0 branch after a separate (stride 2) depthwise conv
      const int T1a = add_buf(n, Kt, 1, Hc, Wc);           // first block: 1x1 at the INPUT resolution
      const int T1 = add_buf(n, Kt, 1, Ho, Wo);
      const int T2 = add_buf(n, Kt, 0, Ho, Wo);
      // a pointwise layer that writes into the stage buffer.  fp32: natural column order, the kernel scatters column i to
      // out_cmap[i]; bf16: the columns are packed as the whole slot groups the layer owns (col_map), the kernel stores a
      // group of 8 columns at out_cmap[first column]
      auto to_stage = [&](int layer, const std::vector<int32_t>& slot_of, int min_coutp = 64) -> int {
        if (!n->bf16) return add_map(n, slot_of);
        std::vector<int32_t> cols, chan;
        zc_columns(slot_of, G, &cols, &chan);
        SLayer& L = n->layers[layer];
        L.ncols = (int)cols.size();
        L.coutp = L.ncols <= 64 ? 64 : (L.ncols <= 128 ? 128 : (L.ncols + 255) / 256 * 256);
        if (L.coutp < min_coutp) L.coutp = min_coutp;
        cols.resize(L.coutp, -1);       // (columns past ncols: zero columns of the packed matrix)
        chan.resize(L.coutp, -1);
        L.colmap_id = add_map(n, cols);
        return add_map(n, chan);
      };
      auto to_temp = [&](int layer, int cout) {  // contiguous output channels (a unit temporary): whole 8-channel groups
        if (!n->bf16) return;
        n->layers[layer].ncols = (cout + 7) / 8 * 8;
        n->layers[layer].coutp = cout_pad(cout);
      };
      {  // -- block 0: two-branch (reference :47-53, :60-61) --
        const std::string bp = sp + "0.";
        int M_in = -1, M_inpl = -1;
        if (in_is_stage) {
          M_in = add_map(n, in_kmap);
          M_inpl = add_map(n, in_planes_v);
        }
        std::vector<int32_t> even(h), odd(h);
        for (int i = 0; i < h; ++i) {
          even[i] = zs.phys0[2 * i];
          odd[i] = zs.phys0[2 * i + 1];
        }
        const int l_c00 = add_layer(n, L_DW, bp + "conv0.0", in_c, in_c, in_phys, M_in);
        const int l_c01 = add_layer(n, L_PW, bp + "conv0.1", h, in_c, up8(in_phys), M_in);
        const int l_c0 = add_layer(n, L_PW, bp + "conv.0", h, in_c, up8(in_phys), M_in);
        const int l_c1 = add_layer(n, L_DW, bp + "conv.1", h, h, Kt, -1);
        const int l_c2 = add_layer(n, L_PW, bp + "conv.2", h, h, Kt, -1);
        const int M_even = to_stage(l_c01, even), M_odd = to_stage(l_c2, odd);
        to_temp(l_c0, h);
        if (stride == 1) {
          add_pwf(n, bp + "conv0.0+conv0.1", Ho, Wo, l_c01, l_c00, in_buf, 0, S, 0, M_even, 1);
          n->ops.back().planes_map = M_inpl;
        } else {
          add_dw(n, bp + "conv0.0", Hc, Wc, l_c00, in_buf, T0, stride);
          add_pwf(n, bp + "conv0.1", Ho, Wo, l_c01, -1, T0, 0, S, 0, M_even, 1);
        }
        add_pwf(n, bp + "conv.0", Hc, Wc, l_c0, -1, in_buf, 0, T1a, 0, -1, 1);
        n->ops.back().planes_map = M_inpl;
        if (stride == 1) {
          add_pwf(n, bp + "conv.1+conv.2", Ho, Wo, l_c2, l_c1, T1a, 0, S, 0, M_odd, 1);
        } else {
          add_dw(n, bp + "conv.1", Hc, Wc, l_c1, T1a, T2, stride);
          add_pwf(n, bp + "conv.2", Ho, Wo, l_c2, -1, T2, 0, S, 0, M_odd, 1);
        }
      }
      for (int u = 1; u < U; ++u) {  // -- blocks 1..: pass-through half + processed half (reference :31-39, :56-59) --
        const std::string bp = sp + std::to_string(u) + ".";
        const ZcUnit& zu = zs.units[u - 1];
        const int M_x2 = add_map(n, zu.x2map), M_pl = add_map(n, zu.planes);
        const int l_c0 = add_layer(n, L_PW, bp + "conv.0", h, h, (int)zu.x2map.size(), M_x2);
        const int l_c1 = add_layer(n, L_DW, bp + "conv.1", h, h, Kt, -1);
        const int l_c2 = add_layer(n, L_PW, bp + "conv.2", h, h, Kt, -1);
        // (the one-launch kernel wants both matrices 128 or 256 columns wide, every column of conv.0 packed - zeros
        //  past the real ones)
        const int M_y = to_stage(l_c2, zu.yslot, one_launch_units ? 128 : 64);
        to_temp(l_c0, h);
        if (one_launch_units) {
          SLayer& L0 = n->layers[l_c0];
          L0.coutp = Kt <= 128 ? 128 : 256;
          L0.ncols = L0.coutp;
          SOp o;
          o.kind = O_UNIT;
          o.name = bp + "conv.0+conv.1+conv.2";
          o.H = Ho;
          o.W = Wo;
          o.layer[0] = l_c0;
          o.layer[1] = l_c2;
          o.dw_layer = l_c1;
          o.in_buf[0] = S;
          o.out_buf[0] = S;
          o.planes_map = M_pl;
          o.cmap[0] = M_y;
          o.relu = 1;
          o.flops = 2.0 * n->N * Ho * Wo * ((double)h * h * 2 + (double)h * 9);
          n->ops.push_back(o);
          continue;
        }
        add_pwf(n, bp + "conv.0", Ho, Wo, l_c0, -1, S, 0, T1, 0, -1, 1);
        n->ops.back().planes_map = M_pl;
        add_pwf(n, bp + "conv.1+conv.2", Ho, Wo, l_c2, l_c1, T1, 0, S, 0, M_y, 1);
      }
      in_buf = S;
      in_c = C;
      in_is_stage = true;
      in_pmap.assign(Pp, -1);
      for (int j = 0; j < C; ++j) in_pmap[zs.phys_final[j]] = j;
      in_kmap.clear();
      in_planes_v.clear();
      for (int pl = 0; pl < Pp / G; ++pl) {
        bool any = false;
        for (int e = 0; e < G; ++e) any = any || in_pmap[pl * G + e] >= 0;
        if (!any) continue;
        in_planes_v.push_back(pl * G);
        for (int e = 0; e < G; ++e) in_kmap.push_back(in_pmap[pl * G + e]);
      }
      while (in_kmap.size() % 16) {  // (K of the readers: whole 16-channel steps; zero-weight repeats of the first plane)
        in_planes_v.push_back(in_planes_v[0]);
        for (int e = 0; e < G; ++e) in_kmap.push_back(-1);
      }
      Hc = Ho;
      Wc = Wo;
    }
  }

  // ---- conv5 + heads -------------------------------------------------------------------
  {
    const int M_in = add_map(n, in_kmap);
    const int M_inpl = in_planes_v.empty() ? -1 : add_map(n, in_planes_v);
    const int l5 = add_layer(n, L_PW, "network.6", 1024, in_c, up8((int)in_kmap.size()), M_in);
    const int lp = add_layer(n, L_PW, "paf", 38, 1024, 1024, -1);
    const int lh = add_layer(n, L_PW, "heatmap", 19, 1024, 1024, -1);
    // conv5 and the heads are ONE launch (pw_head.hip / pw_head_bf16.hip) - the 1024-channel feature has no buffer
    const int OUT = add_buf(n, 64, 0, Hc, Wc, true);  // fp32 [PAF 0..37 | 2 pad | heat 40..58 | pad]
    n->out_buf = OUT;
    // the heads share a 64-column matrix whose columns sit AT their output channels (PAF 0..37, heat-map 40..58):
    // the kernel stores 16 bytes per lane, no column map; the columns nobody owns are zeroed at load time
    SLayer& P = n->layers[lp];
    SLayer& Hh = n->layers[lh];
    P.coutp = Hh.coutp = 64;
    Hh.col_off = 40;
    P.ncols = 38;
    Hh.ncols = 19;
    P.zero_c0 = 38;
    P.zero_c1 = 40;
    Hh.zero_c0 = 59;
    Hh.zero_c1 = 64;
    if (n->bf16) {
      n->layers[l5].ncols = 1024;
      n->layers[l5].coutp = 1024;
    }
    Hh.w_off = P.w_off = n->wt_floats;
    n->wt_floats += round_up((size_t)(1024 + 32) * 64, 64);
    Hh.b_off = P.b_off = n->wt_floats;
    n->wt_floats += 64;
    SOp o;
    o.kind = O_HEAD;
    o.name = "conv5+paf+heatmap";
    o.H = Hc;
    o.W = Wc;
    o.layer[0] = l5;
    o.layer[1] = lp;
    o.in_buf[0] = in_buf;
    o.out_buf[0] = OUT;
    o.planes_map = M_inpl;
    o.relu = 1;
    o.flops = 2.0 * n->N * Hc * Wc * (1024.0 * in_c + 1024.0 * 57);
    n->ops.push_back(o);
  }
}

rtpose_layout slice(const SBuf& b, int choff) {
  rtpose_layout l = b.lay;
  l.choff = choff;
  return l;
}

}  // namespace

extern "C" {

int rtpose_shufflenet_create_ex(int N, int H, int W, int dtype, rtpose_shufflenet** out) {
  if (!out || N <= 0 || H < 32 || W < 32) return fail(RTPOSE_E_INVAL, "shufflenet_create: need N>=1, H,W>=32");
  if (dtype != RTPOSE_DTYPE_F32 && dtype != RTPOSE_DTYPE_BF16)
    return fail(RTPOSE_E_INVAL, "shufflenet_create: dtype must be RTPOSE_DTYPE_F32 or RTPOSE_DTYPE_BF16");
  rtpose_shufflenet* n = new rtpose_shufflenet();
  n->N = N;
  n->H = H;
  n->W = W;
  n->bf16 = dtype == RTPOSE_DTYPE_BF16;
  build(n);
  *out = n;
  return 0;
}

int rtpose_shufflenet_create(int N, int H, int W, rtpose_shufflenet** out) {
  return rtpose_shufflenet_create_ex(N, H, W, RTPOSE_DTYPE_F32, out);
}

void rtpose_shufflenet_destroy(rtpose_shufflenet* n) {
  if (!n) return;
  for (hipEvent_t e : n->ev) (void)hipEventDestroy(e);
  delete n;
}

size_t rtpose_shufflenet_workspace_bytes(const rtpose_shufflenet* n) { return n->ws_floats * 4; }
size_t rtpose_shufflenet_weight_bytes(const rtpose_shufflenet* n) { return n->wt_floats * 4; }

int rtpose_shufflenet_bind(rtpose_shufflenet* n, void* workspace, size_t ws_bytes, void* weights,
                           size_t wt_bytes, int zero_workspace, void* stream) {
  if (!n || !workspace || !weights) return fail(RTPOSE_E_INVAL, "shufflenet_bind: NULL argument");
  if (ws_bytes < n->ws_floats * 4 || wt_bytes < n->wt_floats * 4)
    return fail(RTPOSE_E_INVAL, "shufflenet_bind: arena too small");
  const int dev = current_device();
  int rcd = check_device_ptr(workspace, dev, "shufflenet_bind", "the workspace");
  if (!rcd) rcd = check_device_ptr(weights, dev, "shufflenet_bind", "the weight arena");
  if (rcd) return rcd;
  n->device = dev;
  n->in_checked = CheckedPtr();
  n->ws = static_cast<float*>(workspace);
  n->wt = static_cast<float*>(weights);
  hipStream_t s = as_stream(stream);
  // The workspace is cleared whatever the caller says: the zero-copy stage buffers and the unit temporaries have
  // slots no launch ever writes (class padding inside a slot group, groups skipped for alignment, Pp - P, channels
  // past h of a temporary) and the GEMMs gather them as K rows under ZERO weights - 0 x NaN from a recycled arena
  // would reach every output channel.  (`zero_workspace` is kept in the signature for the callers that pass it.)
  (void)zero_workspace;
  RTPOSE_HIP_CHECK(hipMemsetAsync(workspace, 0, n->ws_floats * 4, s));
  for (const Map& m : n->maps)
    RTPOSE_HIP_CHECK(hipMemcpyAsync(n->wt + m.off, m.v.data(), m.v.size() * 4, hipMemcpyHostToDevice, s));
  RTPOSE_HIP_CHECK(hipStreamSynchronize(s));
  n->bound = true;
  return 0;
}

int rtpose_shufflenet_num_layers(const rtpose_shufflenet* n) { return (int)n->layers.size(); }

/* kind: 0 input affine (w = scale[3], b = shift[3]); 1 stem conv [24,3,3,3]; 2 depthwise [C,1,3,3];
 * 3 pointwise [cout,cin,1,1].  BatchNorm already folded by the caller. */
int rtpose_shufflenet_layer_info(const rtpose_shufflenet* n, int idx, char* name, int name_cap, int* kind,
                                 int* cout, int* cin) {
  if (!n || idx < 0 || idx >= (int)n->layers.size()) return fail(RTPOSE_E_INVAL, "layer_info: bad index");
  const SLayer& l = n->layers[idx];
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", l.name.c_str());
  if (kind) *kind = (int)l.kind;
  if (cout) *cout = l.cout;
  if (cin) *cin = l.cin;
  return 0;
}

int rtpose_shufflenet_load(rtpose_shufflenet* n, int idx, const float* w, const float* b, void* stream) {
  if (!n || !n->bound) return fail(RTPOSE_E_STATE, "shufflenet_load: not bound");
  if (idx < 0 || idx >= (int)n->layers.size() || !w) return fail(RTPOSE_E_INVAL, "shufflenet_load: bad argument");
  const SLayer& l = n->layers[idx];
  hipStream_t s = as_stream(stream);
  const int32_t* map = l.map_id >= 0 ? reinterpret_cast<const int32_t*>(n->wt + n->maps[l.map_id].off) : nullptr;
  switch (l.kind) {
    case L_AFFINE:
      RTPOSE_HIP_CHECK(hipMemcpyAsync(n->wt + l.w_off, w, 3 * 4, hipMemcpyDeviceToDevice, s));
      RTPOSE_HIP_CHECK(hipMemcpyAsync(n->wt + l.b_off, b, 3 * 4, hipMemcpyDeviceToDevice, s));
      return 0;
    case L_STEM: {
      const int total = 9 * 8 * l.cout;
      hipLaunchKernelGGL(pack_stem_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, w, l.cout, l.cin, 8,
                         n->wt + l.w_off);
      RTPOSE_HIP_CHECK(hipMemcpyAsync(n->wt + l.b_off, b, (size_t)l.cout * 4, hipMemcpyDeviceToDevice, s));
      RTPOSE_HIP_CHECK(hipGetLastError());
      return 0;
    }
    case L_DW: {
      const int total = 10 * l.cin_packed;
      hipLaunchKernelGGL(pack_dw_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, w, b, l.cin, map,
                         l.cin_packed, n->wt + l.w_off, n->wt + l.b_off);
      RTPOSE_HIP_CHECK(hipGetLastError());
      return 0;
    }
    case L_PW:
      if (n->bf16 && l.zero_c1 > l.zero_c0) {  // a head of the bf16 plan: the k order of pw_head_bf16.hip's GEMM 2
        const int rc = pack_head_w2_bf16_launch(w, b, l.cout, l.cin, l.col_off, n->wt + l.w_off, n->wt + l.b_off, s);
        if (rc) return rc;
        return zero_head_columns_bf16_launch(n->wt + l.w_off, n->wt + l.b_off, l.cin_packed, l.zero_c0, l.zero_c1, s);
      }
      if (n->bf16) {  // column-mapped bf16 packing (see pw_fused_bf16.hip)
        const int32_t* cmap = l.colmap_id >= 0 ? reinterpret_cast<const int32_t*>(n->wt + n->maps[l.colmap_id].off) : nullptr;
        // (a column-mapped layer packs ALL coutp columns: the map marks the ones past its stored width as zero columns)
        return pack_pw_bf16_launch(w, b, l.cout, l.cin, map, l.cin_packed, cmap ? l.coutp : l.ncols, cmap, l.coutp,
                                   l.col_off, n->wt + l.w_off, n->wt + l.b_off, s);
      }
      if (l.coutp) {  // shares a packed matrix with another layer (the heads of a fused plan)
        const int rc = pack_pw_launch(w, b, l.cout, l.cin, map, l.cin_packed, l.coutp, l.col_off, n->wt + l.w_off,
                                      n->wt + l.b_off, s);
        if (rc || l.zero_c1 <= l.zero_c0) return rc;
        return pw_zero_columns_launch(n->wt + l.w_off, n->wt + l.b_off, l.cin_packed, l.coutp, l.zero_c0, l.zero_c1, s);
      }
      if (n->bf16)
        return pack_weights_bf16_launch(w, b, l.cout, l.cin, 1, map, l.cin_packed, n->wt + l.w_off,
                                        n->wt + l.b_off, 0, s);
      return pack_weights_launch(w, b, l.cout, l.cin, 1, map, l.cin_packed, n->wt + l.w_off, n->wt + l.b_off, s);
  }
  return 0;
}

int rtpose_shufflenet_set_profiling(rtpose_shufflenet* n, int enable) {
  if (!n) return fail(RTPOSE_E_INVAL, "NULL net");
  n->profiling = enable ? 1 : 0;
  if (enable && n->ev.empty()) {
    n->ev.resize(n->ops.size() + 1);
    for (auto& e : n->ev) RTPOSE_HIP_CHECK(hipEventCreate(&e));
  }
  n->ev_valid = false;
  return 0;
}

int rtpose_shufflenet_num_launches(const rtpose_shufflenet* n) { return (int)n->ops.size(); }

int rtpose_shufflenet_launch_info(rtpose_shufflenet* n, int i, float* ms, double* flops, char* name,
                                  int name_cap) {
  if (!n || i < 0 || i >= (int)n->ops.size()) return fail(RTPOSE_E_INVAL, "launch_info: bad index");
  const SOp& o = n->ops[i];
  if (flops) *flops = o.flops;
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", o.name.c_str());
  if (ms) {
    *ms = -1.f;
    float t = 0.f;
    if (n->profiling && n->ev_valid && hipEventElapsedTime(&t, n->ev[i], n->ev[i + 1]) == hipSuccess) *ms = t;
  }
  return 0;
}

int rtpose_shufflenet_forward(rtpose_shufflenet* n, const float* x_nchw, void* stream) {
  if (!n || !n->bound) return fail(RTPOSE_E_STATE, "shufflenet_forward: not bound");
  if (!x_nchw) return fail(RTPOSE_E_INVAL, "shufflenet_forward: x is NULL");
  if (current_device() != n->device)
    return fail(RTPOSE_E_STATE, "shufflenet_forward: the plan's arenas live on HIP device %d but the current device is %d",
                n->device, current_device());
  const int rcd = n->in_checked.check(x_nchw, n->device, "shufflenet_forward", "the input tensor");
  if (rcd) return rcd;
  hipStream_t s = as_stream(stream);
  const bool prof = n->profiling && !n->ev.empty();
  auto imap = [&](int id) -> const int32_t* {
    return id >= 0 ? reinterpret_cast<const int32_t*>(n->wt + n->maps[id].off) : nullptr;
  };
  for (size_t i = 0; i < n->ops.size(); ++i) {
    const SOp& o = n->ops[i];
    if (prof) RTPOSE_HIP_CHECK(hipEventRecord(n->ev[i], s));
    int rc = 0;
    switch (o.kind) {
      case O_INPUT:  // fused into the stem conv, which reads the NCHW image itself
        break;
      case O_STEM: {
        const SBuf& bo = n->bufs[o.out_buf[0]];
        const SLayer& l = n->layers[o.layer[0]];
        const SLayer& la = n->layers[n->ops[0].layer[0]];  // the input BatchNorm2d(3) as scale / shift
        rc = rtpose_stem_conv3x3_s2_nchw_ex(x_nchw, n->wt + la.w_off, n->wt + la.b_off, n->wt + l.w_off,
                                            n->wt + l.b_off, n->ws + bo.off, &bo.lay, l.cout, n->N, o.H, o.W,
                                            o.relu, n->bf16, stream);
        break;
      }
      case O_STEMPOOL: {
        const SBuf& bo = n->bufs[o.out_buf[0]];
        const SLayer& l = n->layers[o.layer[0]];
        const SLayer& la = n->layers[n->ops[0].layer[0]];  // the input BatchNorm2d(3) as scale / shift
        rc = rtpose_stem_pool_nchw(x_nchw, n->wt + la.w_off, n->wt + la.b_off, n->wt + l.w_off, n->wt + l.b_off,
                                   n->ws + bo.off, &bo.lay, l.cout, n->N, o.H, o.W, n->bf16, stream);
        break;
      }
      case O_POOL3: {
        const SBuf& bi = n->bufs[o.in_buf[0]];
        const SBuf& bo = n->bufs[o.out_buf[0]];
        rc = n->bf16 ? rtpose_maxpool3x3s2_ceil_bf16(n->ws + bi.off, &bi.lay, n->ws + bo.off, &bo.lay, o.C, n->N,
                                                     o.H, o.W, stream)
                     : rtpose_maxpool3x3s2_ceil(n->ws + bi.off, &bi.lay, n->ws + bo.off, &bo.lay, o.C, n->N, o.H,
                                                o.W, stream);
        break;
      }
      case O_DW: {
        const SBuf& bi = n->bufs[o.in_buf[0]];
        const SBuf& bo = n->bufs[o.out_buf[0]];
        const SLayer& l = n->layers[o.layer[0]];
        rc = n->bf16 ? rtpose_dwconv3x3_bf16(n->ws + bi.off, &bi.lay, n->wt + l.w_off, n->wt + l.b_off,
                                             n->ws + bo.off, &bo.lay, o.C, n->N, o.H, o.W, o.stride, stream)
                     : rtpose_dwconv3x3(n->ws + bi.off, &bi.lay, n->wt + l.w_off, n->wt + l.b_off, n->ws + bo.off,
                                        &bo.lay, o.C, n->N, o.H, o.W, o.stride, stream);
        break;
      }
      case O_PWF: {
        const SLayer& l = n->layers[o.layer[0]];
        const SBuf& bi = n->bufs[o.in_buf[0]];
        const SBuf& bo = n->bufs[o.out_buf[0]];
        rtpose_pw_desc d;
        memset(&d, 0, sizeof(d));
        d.in = n->ws + bi.off;
        d.lin = slice(bi, o.in_choff[0]);
        if (o.dw_layer >= 0) {
          const SLayer& ld = n->layers[o.dw_layer];
          d.dw_w = n->wt + ld.w_off;
          d.dw_b = n->wt + ld.b_off;
        }
        d.w_packed = n->wt + l.w_off;
        d.bias_packed = n->wt + l.b_off;
        d.out = n->ws + bo.off;
        d.lout = slice(bo, o.out_choff[0]);
        d.cin = l.cin_packed;
        d.cout = o.cout_store;
        d.coutp = l.coutp ? l.coutp : cout_pad(l.cout);
        d.relu = o.relu;
        d.out_cmap = imap(o.cmap[0]);
        d.in_planes = imap(o.planes_map);
        if (o.pt_buf >= 0) {
          const SBuf& bp = n->bufs[o.pt_buf];
          d.pt_src = n->ws + bp.off;
          d.lpt = slice(bp, 0);
          d.pt_cmap = imap(o.pt_map);
          d.pt_c = o.pt_c;
          d.pt_pairs = o.pt_pairs;
          d.pt_a = o.pt_a;
          d.pt_b = o.pt_b;
          d.pt_split = o.pt_split;
          d.pt_d0 = o.pt_d0;
          d.pt_d1 = o.pt_d1;
        }
        if (n->bf16) {
          const bool f32out = o.out_buf[0] == n->out_buf;  // the two heads write the fp32 output record
          if (!f32out) {  // columns [0, ncols): contiguous channels from out_choff (a unit temporary), or - a layer that
            d.cout = l.ncols;  // writes into the stage buffer - whole groups of 8 at out_cmap[first column of the group]
            d.out_cmap = l.colmap_id >= 0 ? imap(o.cmap[0]) : nullptr;
          }
          rc = pw_fused_bf16_launch(&d, f32out ? 1 : 0, n->N, o.H, o.W, s);
        } else {
          rc = pw_fused_launch(&d, n->N, o.H, o.W, s);
        }
        break;
      }
      case O_UNIT: {
        const SLayer &l0 = n->layers[o.layer[0]], &l1 = n->layers[o.dw_layer], &l2 = n->layers[o.layer[1]];
        const SBuf& bs = n->bufs[o.in_buf[0]];
        rtpose_pw_desc d0, d2;
        memset(&d0, 0, sizeof(d0));
        memset(&d2, 0, sizeof(d2));
        d0.in = n->ws + bs.off;
        d0.lin = slice(bs, 0);
        d0.w_packed = n->wt + l0.w_off;
        d0.bias_packed = n->wt + l0.b_off;
        d0.cin = l0.cin_packed;
        d0.cout = d0.coutp = l0.coutp;
        d0.relu = 1;
        d0.in_planes = imap(o.planes_map);
        d0.dw_w = n->wt + l1.w_off;
        d0.dw_b = n->wt + l1.b_off;
        d2.w_packed = n->wt + l2.w_off;
        d2.bias_packed = n->wt + l2.b_off;
        d2.cin = l2.cin_packed;
        d2.cout = l2.ncols;
        d2.coutp = l2.coutp;
        d2.relu = 1;
        d2.out = n->ws + bs.off;
        d2.lout = slice(bs, 0);
        d2.out_cmap = imap(o.cmap[0]);
        rc = unit_bf16_launch(&d0, &d2, n->N, o.H, o.W, s);
        break;
      }
      case O_HEAD: {
        const SLayer &l1 = n->layers[o.layer[0]], &l2 = n->layers[o.layer[1]];
        const SBuf& bi = n->bufs[o.in_buf[0]];
        const SBuf& bo = n->bufs[o.out_buf[0]];
        rtpose_pw_desc d1, d2;
        memset(&d1, 0, sizeof(d1));
        memset(&d2, 0, sizeof(d2));
        d1.in = n->ws + bi.off;
        d1.lin = slice(bi, 0);
        d1.w_packed = n->wt + l1.w_off;
        d1.bias_packed = n->wt + l1.b_off;
        d1.cin = l1.cin_packed;
        d1.in_planes = imap(o.planes_map);
        d1.cout = d1.coutp = cout_pad(l1.cout);
        d1.relu = 1;
        d2.w_packed = n->wt + l2.w_off;
        d2.bias_packed = n->wt + l2.b_off;
        d2.cin = d1.coutp;
        d2.cout = d2.coutp = 64;
        d2.out = n->ws + bo.off;
        d2.lout = slice(bo, 0);
        rc = n->bf16 ? pw_head_bf16_launch(&d1, &d2, n->N, o.H, o.W, s) : pw_head_launch(&d1, &d2, n->N, o.H, o.W, s);
        break;
      }
    }
    if (rc) return rc;
  }
  if (prof) {
    RTPOSE_HIP_CHECK(hipEventRecord(n->ev[n->ops.size()], s));
    n->ev_valid = true;
  }
  return 0;
}

/* which: 0 = PAF (38 ch), 1 = heat-map (19 ch) -> dense NCHW */
int rtpose_shufflenet_read_output(rtpose_shufflenet* n, int which, float* dst_nchw, void* stream) {
  if (!n || !n->bound || which < 0 || which > 1 || !dst_nchw)
    return fail(RTPOSE_E_INVAL, "shufflenet_read_output: bad argument");
  const SBuf& b = n->bufs[n->out_buf];
  const rtpose_layout l = slice(b, which == 0 ? 0 : 40);
  return rtpose_layout_to_nchw(n->ws + b.off, &l, dst_nchw, which == 0 ? 38 : 19, n->N, n->Hm, n->Wm, stream);
}

int rtpose_shufflenet_output_view(const rtpose_shufflenet* n, int which, const float** base, rtpose_layout* layout,
                                  int* C, int* H, int* W) {
  if (!n || !n->bound || which < 0 || which > 1) return fail(RTPOSE_E_INVAL, "output_view: bad argument");
  const SBuf& b = n->bufs[n->out_buf];
  if (base) *base = n->ws + b.off;
  if (layout) *layout = slice(b, which == 0 ? 0 : 40);
  if (C) *C = which == 0 ? 38 : 19;
  if (H) *H = n->Hm;
  if (W) *W = n->Wm;
  return 0;
}

}  // extern "C"
