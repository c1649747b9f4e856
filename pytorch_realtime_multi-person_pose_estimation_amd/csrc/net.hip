// Native executor for the rtpose_vgg network (lib/network/rtpose_vgg.py:60-225):
// builds the launch plan for a given (N, H, W), carves the caller-provided
// workspace into shared-gap padded NHWC activation buffers, packs weights into
// the caller-provided weight arena and enqueues the forward
// (rtpose_model.forward, rtpose_vgg.py:158-198) as a fixed list of launches on
// one HIP stream.
//
// What differs from the reference module graph (same arithmetic, other layout):
//  * torch.cat([L1, L2, out1], 1) (rtpose_vgg.py:165,171,177,183,189) never
//    runs: the two branch heads of stage s write straight into channel slices
//    of one 192-channel buffer laid out [out1 0..127 | PAF 128..165 | heat
//    166..184 | 7 zero], and the 185-input-channel filters are permuted to that
//    order when packed.  Two such buffers ping-pong between stages.
//  * the two branches of a stage (independent, rtpose_vgg.py:163-164) run as
//    one grouped grid per layer.
//  * MaxPool2d is fused into the epilogue of the conv in front of it when the
//    map has even height and width.
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace rtpose {
int conv2d_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, hipStream_t s);
int conv2d_winograd_fits(int k, int cin, int cout, int pool, int N, int H, int W, int hs, int fm);
int conv2d_wino_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, hipStream_t s);
int conv2d_wino7_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, int fm, void* scratch,
                        size_t scratch_bytes, hipStream_t s);
double conv2d_wino_issued_flops(int cin, int cout, int N, int H, int W);
double conv2d_wino7_issued_flops(int cin, int cout, int N, int H, int W, int hs, int fm);
size_t conv2d_wino7_scratch_bytes(int blocks);
int* conv2d_wino7_scratch_err(void* scratch, int blocks);
size_t packed_weight_floats_wino7(int cout, int cin, int fm);
int pack_weights_wino7_launch(const float* w, const float* bias, int cout, int cin_src, const int32_t* cin_map,
                              int cin_packed, int fm, float* wp, float* bp, hipStream_t s);
int wino_amplification_launch(const float* w, int cout, int cin, int k, int fm, float* amp, hipStream_t s);
int conv2d_wino4_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, hipStream_t s);
double conv2d_wino4_issued_flops(int cin, int cout, int N, int H, int W);
int wino7_default_fm();
// Mconv6 + Mconv7 of a stage as one launch (conv_tail.hip)
int conv_tail_launch(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups, int N, int H, int W,
                     hipStream_t s);
// conv1_1 (conv_first.hip)
size_t conv_first_packed_floats();
int conv_tail_bf16_launch(const rtpose_conv_desc* d1, const rtpose_conv_desc* d2, int ngroups, int N, int H, int W,
                          int out_f32, hipStream_t s);
int conv_first_pack_launch(const float* w_oihw, const float* bias, float* wp, hipStream_t s, int to_bf16);
int conv_first_launch(const float* x_nchw, const float* x_lay, const rtpose_layout* lx, const float* wp, float* out,
                      const rtpose_layout* lo, int out_plane_pixels, int relu, int N, int H, int W, hipStream_t s,
                      int out_bf16);
int pack_weights_launch(const float* w, const float* bias, int cout, int cin_src, int k,
                        const int32_t* cin_map, int cin_packed, float* wp, float* bp, hipStream_t s);
// bf16 path (conv_mfma_bf16.hip)
int conv2d_bf16_launch(const rtpose_conv_desc* d, int ngroups, int N, int H, int W, int out_f32, int split,
                       hipStream_t s);
int pack_weights_bf16_launch(const float* w, const float* bias, int cout, int cin_src, int k,
                             const int32_t* cin_map, int cin_packed, void* wp, float* bp, int split,
                             hipStream_t s);
}  // namespace rtpose

using namespace rtpose;

namespace {

struct Buf {
  size_t off_floats = 0;  // offset in the workspace
  size_t floats = 0;
  rtpose_layout lay{};
  int C = 0, H = 0, W = 0;
  // fp32 plans: the buffer is stored as 8-channel planes of `plane_px` pixel slots (mark_plane_bufs) - what the forms of
  // its producer and consumers allow; `stale`: its bytes were written in the other storage and must be cleared first
  // (the gaps of either storage are data positions of the other)
  int plane_px = 0;
  bool stale = false;
};

struct ConvW {
  std::string name;
  int cout = 0, cin_src = 0, cin_packed = 0, k = 0;
  bool cat_perm = false;   // input channels follow the cat([L1,L2,out1]) order
  // fp32 plans.  The arena holds EVERY packing a plan may run the conv in (it is shared by all plans of a module,
  // whatever their geometry and options): the direct one at w_off, and - where the form has a kernel for these
  // channel counts - F(2x2,3x3) at w_off_w3, F(4x4,3x3) at w_off_w43, F(4,7) / F(6,7) at w_off_w4 / w_off_w6.
  bool first = false;      // conv1_1 (3 -> 64, 3x3): its own kernel, packing at w_off_first (csrc/conv_first.hip)
  size_t w_off_first = 0;
  bool has_w3 = false, has_w43 = false, has_w7 = false;
  size_t w_off_w3 = 0, w_off_w43 = 0, w_off_w4 = 0, w_off_w6 = 0;
  size_t amp_off = 0;      // 4 floats in the arena: amplification estimates in F(2x2,3x3) / F(4,7) / F(6,7) / F(4x4,3x3) (0 = n/a)
  float amp[4] = {0.f, 0.f, 0.f, 0.f};  // host copy (rtpose_net_finalize_weights)
  int form = 0;            // what THIS plan runs the conv in: 0 direct, 3 = F(2x2,3x3), 43 = F(4x4,3x3), 4 = F(4,7), 6 = F(6,7)
  int H = 0, W = 0;        // map size the conv runs at in this plan
  size_t w_off = 0, b_off = 0;  // float offsets in the weight arena
};

enum OpKind { OP_INPUT, OP_CONV, OP_POOL, OP_COPY, OP_TAIL };

struct Op {
  OpKind kind;
  std::string name;
  int H = 0, W = 0;         // spatial size the op runs at
  // conv
  int ngroups = 0;
  int conv_idx[2] = {-1, -1};
  int conv2_idx[2] = {-1, -1};  // OP_TAIL: the second conv of the back-to-back pair
  int in_buf[2] = {-1, -1}, out_buf[2] = {-1, -1};
  int in_choff[2] = {0, 0}, out_choff[2] = {0, 0};
  int relu = 0, pool = 0;
  int out_f32 = 0;          // bf16 plans: this conv writes fp32 (the final stage heads)
  // pool / copy
  int C = 0;
  double flops = 0.0;
  int ks = 0;
};

}  // namespace

struct rtpose_net {
  int N = 0, H = 0, W = 0;       // input
  int bf16 = 0;                  // 1: bf16 activations/weights, fp32 accumulate (BASELINE config 3)
  int split = 0;                 // bf16 plans only: 1 = "bf16x3" split operands (hi + lo bf16 per value)
  int w3 = RTPOSE_WINO3_AUTO;    // fp32 plans, 3x3 convs: 0 direct, 1 = F(2x2,3x3), 4 = F(4x4,3x3), RTPOSE_WINO3_AUTO = per layer by amp_limit (the default)
  int w7 = RTPOSE_WINO7_AUTO;    // fp32 plans: 0 direct, 4 / 6 = F(4,7) / F(6,7), RTPOSE_WINO7_AUTO = per layer by amp_limit (the default)
  float amp_limit = 256.f;
  bool forms_final = false;      // forms chosen (AUTO: after the amplification estimates were read back)
  bool amps_read = false;
  uint64_t seen_gen = ~0ull;     // generation of the weight arena the estimates / forms above were taken from
  int n_cu = 0;                  // CUs of the device the plan was created for (sizes the hand-over scratch)
  hipEvent_t out_guard = nullptr;  // rtpose_net_set_output_guard: waited for before the first launch that writes the buffer
  int guard_op = -1;               // the stage-6 maps are read from (index of that launch)
  int persist7 = 1;              // 1: 7x7 launches whose tiles are not whole rounds run as persistent blocks with split tiles
                                 // (rtpose_net_set_persistent7; cleared by a hand-over that timed out)
  int device = -1;               // HIP device that owns the bound arenas: the only device this plan launches on
  CheckedPtr in_checked;         // last input pointer verified to live on that device
  size_t scratch_off = 0, scratch_bytes = 0;  // persistent 7x7 launches: hand-over scratch inside the workspace
  int x0f_buf = -1;              // bf16 plans: fp32 NHWC8 staging buffer for rtpose_preprocess_u8
  int H3 = 0, W3 = 0;            // stride-8 map
  std::vector<Buf> bufs;
  std::vector<ConvW> convs;
  std::vector<Op> ops;
  size_t ws_floats = 0, wt_floats = 0;
  size_t catmap_off = 0;         // int32[192] inside the weight arena
  float* ws = nullptr;
  float* wt = nullptr;
  bool bound = false;
  int keep = 0;
  int save_buf[6] = {-1, -1, -1, -1, -1, -1};
  int cat_buf[2] = {-1, -1};
  int x0_buf = -1;
  // hipGraph replay of the launch list (ops after the input conversion have fixed arguments):
  // captured once per keep_intermediates setting on a private non-blocking stream that is
  // joined to the caller's stream by events, so it also works under the legacy NULL stream
  int graph_mode = -1;             // -1 unread, 0 off (default), 1 on (RTPOSE_GRAPH=1)
  bool zeroed_at_bind = false;     // rtpose_net_bind cleared the workspace (nothing to clear before the first forward)
  int forwards = 0;                // the first forward runs directly (lazy statics, attributes)
  hipStream_t gstream = nullptr;
  hipEvent_t gev_in = nullptr, gev_out = nullptr;
  hipGraphExec_t gexec[2] = {nullptr, nullptr};
  // profiling
  int profiling = 0;
  std::vector<hipEvent_t> ev;
  bool ev_valid = false;
};

namespace {

// Plans of one module share ONE weight arena, and any of them may be the one a reload is issued through
// (rtpose_net_load_conv).  The forms of an AUTO plan depend on the filters, so every plan must notice a reload made
// through a sibling: the arena's generation - a host-side counter keyed by the arena's base address, bumped by every
// fp32 rtpose_net_load_conv - is compared with the one the plan's estimates were read at (rtpose_net.seen_gen) before
// every forward / finalize / conv_numerics.  (An arena freed and another allocated at the same address just continues
// the count; rtpose_net_bind forgets the plan's generation.)
std::mutex g_arena_mu;
std::unordered_map<const void*, uint64_t> g_arena_gen;

uint64_t arena_generation(const void* wt) {
  std::lock_guard<std::mutex> lk(g_arena_mu);
  auto it = g_arena_gen.find(wt);
  return it == g_arena_gen.end() ? 0 : it->second;
}

void arena_bump(const void* wt) {
  std::lock_guard<std::mutex> lk(g_arena_mu);
  ++g_arena_gen[wt];
}

constexpr int kCatC = 192;      // [out1 128 | PAF 38 | heat 19 | pad 7]
constexpr int kCatPaf = 128, kCatHeat = 166;

int add_buf(rtpose_net* n, int C, int P, int H, int W, bool f32 = false) {
  Buf b;
  b.C = C;
  b.H = H;
  b.W = W;
  b.lay.cstride = C;
  b.lay.choff = 0;
  b.lay.ws = W + P;
  b.lay.hs = H + P;
  b.lay.lead = P * (W + P) + P;
  // bf16 plans keep activations as 2-byte elements (C is even for every such buffer)
  // (split plans: two 2-byte elements per channel = the fp32 footprint)
  const size_t per_px = (n->bf16 && !n->split && !f32) ? (size_t)C / 2 : (size_t)C;
  b.floats = round_up(rtpose_layout_pixels(&b.lay, n->N, H, W) * per_px, 64);
  b.off_floats = n->ws_floats;
  n->ws_floats += b.floats;
  n->bufs.push_back(b);
  return (int)n->bufs.size() - 1;
}

int add_conv_w(rtpose_net* n, const std::string& name, int cout, int cin, int k, bool cat_perm, int H = 0, int W = 0) {
  ConvW c;
  c.name = name;
  c.cout = cout;
  c.cin_src = cin;
  c.cin_packed = cat_perm ? kCatC : (n->bf16 ? ceil_div(cin, 16) * 16 : ceil_div(cin, 8) * 8);
  c.k = k;
  c.cat_perm = cat_perm;
  c.H = H;
  c.W = W;
  auto take = [&](size_t floats) {
    const size_t off = n->wt_floats;
    n->wt_floats += round_up(floats, 64);
    return off;
  };
  if (n->bf16) {
    c.w_off = take(n->split ? rtpose_packed_weight_bytes_bf16x3(cout, c.cin_packed, k) / 4
                            : rtpose_packed_weight_bytes_bf16(cout, c.cin_packed, k) / 4);
    // bf16 plans (not the split ones): conv1_1 has its own kernel too (conv_first.hip MODE 2: reads the fp32 image, rounds it
    // and the filters to bf16, writes bf16) - the generic packing above stays in the arena for rtpose_net_conv introspection
    c.first = !n->split && k == 3 && cin == 3 && cout == 64;
    if (c.first) c.w_off_first = take(conv_first_packed_floats());
  } else {
    // The weight arena is shared by every plan of a module (any N x H x W, any rtpose_net_options), so its layout
    // depends on the channel counts only: the direct packing, plus every Winograd packing that has a kernel.
    // (3x3 with < 32 input channels stays direct: nothing to amortise the transforms over - conv1_1, 3 -> 64 on 8
    //  padded channels, takes 0.71 ms in Winograd form and 0.50 ms in the direct kernel)
    c.w_off = take(rtpose_packed_weight_floats(cout, c.cin_packed, k));
    c.has_w3 = k == 3 && c.cin_packed >= 32 && conv2d_winograd_fits(3, c.cin_packed, cout, 0, 1, 8, 8, 9, 0);
    c.has_w7 = k == 7 && c.cin_packed % 8 == 0 && cout_pad(cout) % 128 == 0;
    c.has_w43 = c.has_w3 && conv2d_winograd_fits(3, c.cin_packed, cout, 0, 1, 8, 8, 9, 4);
    if (c.has_w3) c.w_off_w3 = take(rtpose_packed_weight_floats_winograd(cout, c.cin_packed, 3));
    if (c.has_w43) c.w_off_w43 = take(rtpose_packed_weight_floats_winograd3(cout, c.cin_packed, 4));
    if (c.has_w7) {
      c.w_off_w4 = take(packed_weight_floats_wino7(cout, c.cin_packed, 4));
      c.w_off_w6 = take(packed_weight_floats_wino7(cout, c.cin_packed, 6));
    }
    c.amp_off = take(4);
    c.first = k == 3 && cin == 3 && cout == 64;
    if (c.first) c.w_off_first = take(conv_first_packed_floats());
  }
  c.b_off = take(rtpose_packed_bias_floats(cout));
  n->convs.push_back(c);
  return (int)n->convs.size() - 1;
}

// The form plan `n` runs conv `c` in.  (It is NOT chosen by batch size: the direct kernel sums in another order, and
// an image's maps would depend on the batch it is evaluated in.  Small grids get the frequency-split launch of the
// same arithmetic instead, conv_wino7.hip: wino7s_f32.)  A form without a kernel instance at the plan's geometry -
// F(m,7) on maps so wide that the transformed rows of a block do not fit the LDS - falls back to the next one.
int pick_form(const rtpose_net* n, const ConvW& c) {
  if (n->bf16) return 0;
  if (c.k == 3) {
    if (!c.has_w3 || !n->w3) return 0;
    if (c.has_w43 && (n->w3 == 4 || (n->w3 == RTPOSE_WINO3_AUTO && c.amp[3] <= n->amp_limit))) return 43;
    return 3;
  }
  if (c.k != 7 || !c.has_w7 || !n->w7) return 0;
  auto fits = [&](int fm) {
    return conv2d_winograd_fits(7, c.cin_packed, c.cout, 0, n->N, c.H, c.W, c.H + 3, fm) != 0;
  };
  if (n->w7 == RTPOSE_WINO7_AUTO) {
    if (c.amp[2] <= n->amp_limit && fits(6)) return 6;
    if (c.amp[1] <= n->amp_limit && fits(4)) return 4;
    return 0;
  }
  if (n->w7 == 6 && fits(6)) return 6;
  return fits(4) ? 4 : 0;
}

bool forms_need_amps(const rtpose_net* n) {
  return !n->bf16 && (n->w7 == RTPOSE_WINO7_AUTO || n->w3 == RTPOSE_WINO3_AUTO);
}

// the filters in the arena changed since this plan last looked (through this plan or a sibling): estimates and AUTO
// forms are stale
void sync_arena_generation(rtpose_net* n) {
  if (n->bf16 || !n->bound) return;
  const uint64_t g = arena_generation(n->wt);
  if (g == n->seen_gen) return;
  n->seen_gen = g;
  n->amps_read = false;
  if (forms_need_amps(n)) n->forms_final = false;
}

void pick_forms(rtpose_net* n) {
  for (ConvW& c : n->convs) c.form = pick_form(n, c);
  // the two branches of a grouped launch run one kernel: the more conservative form of the two
  for (Op& o : n->ops) {
    if (o.kind != OP_CONV || o.ngroups < 2) continue;
    ConvW &a = n->convs[o.conv_idx[0]], &b = n->convs[o.conv_idx[1]];
    const int f = a.form < b.form ? a.form : b.form;  // 3x3: 0 < 3 < 43, 7x7: 0 < 4 < 6: direct is the lowest
    a.form = b.form = f;
  }
}

// Channel-plane storage (conv_wino4.hip) for every buffer that only F(4x4,3x3) launches - and conv1_1 - touch: written by
// ONE conv (conv1_1's own kernel or a conv in form 43; a branch of a grouped launch counts) as a whole, read only by convs in
// form 43 as a whole.  Depends on the
// forms, so it is re-derived whenever they are; a buffer that changes storage is cleared before the next forward.
void mark_plane_bufs(rtpose_net* n) {
  const int nb = (int)n->bufs.size();
  std::vector<int> writers(nb, 0), readers(nb, 0);
  std::vector<char> ok(nb, 1);
  for (const Op& o : n->ops) {
    const bool conv = o.kind == OP_CONV;
    for (int g = 0; g < (o.ngroups > 0 ? o.ngroups : 1); ++g) {
      const int bi = o.in_buf[g], bo = o.out_buf[g];
      const ConvW* c = conv ? &n->convs[o.conv_idx[g]] : nullptr;
      if (bi >= 0) {
        ++readers[bi];
        if (!(conv && c->form == 43 && o.in_choff[g] == 0 && c->cin_packed == n->bufs[bi].C)) ok[bi] = 0;
      }
      if (bo >= 0) {
        ++writers[bo];
        if (!(conv && (c->form == 43 || c->first) && o.out_choff[g] == 0 && c->cout == n->bufs[bo].C))
          ok[bo] = 0;
      }
    }
  }
  for (int b = 0; b < nb; ++b) {
    Buf& bf = n->bufs[b];
    const size_t px = bf.C > 0 ? bf.floats / (size_t)bf.C : 0;  // pixel slots of the buffer (>= the layout's pixels)
    const bool planes = !n->bf16 && ok[b] && writers[b] == 1 && readers[b] >= 1 && bf.C % 8 == 0 &&
                        px * (size_t)bf.C * sizeof(float) < 0x7ffffffeull && px <= 0x7fffffffull;
    const int want = planes ? (int)px : 0;
    if (want != bf.plane_px) {
      bf.plane_px = want;
      bf.stale = true;
    }
  }
}

void add_conv_op(rtpose_net* n, int H, int W, int ngroups, const int* conv_idx, const int* in_buf,
                 const int* in_choff, const int* out_buf, const int* out_choff, int relu, int pool) {
  Op o;
  o.kind = OP_CONV;
  o.H = H;
  o.W = W;
  o.ngroups = ngroups;
  o.relu = relu;
  o.pool = pool;
  double fl = 0;
  for (int i = 0; i < ngroups; ++i) {
    o.conv_idx[i] = conv_idx[i];
    o.in_buf[i] = in_buf[i];
    o.in_choff[i] = in_choff[i];
    o.out_buf[i] = out_buf[i];
    o.out_choff[i] = out_choff[i];
    const ConvW& c = n->convs[conv_idx[i]];
    // algorithmic flops: direct-convolution count of the reference nn.Conv2d
    fl += 2.0 * n->N * H * W * (double)c.cout * c.cin_src * c.k * c.k;
    o.name += (i ? "+" : "") + c.name;
  }
  o.flops = fl;
  o.ks = n->convs[conv_idx[0]].k;
  n->ops.push_back(o);
}

void add_simple_op(rtpose_net* n, OpKind kind, const std::string& name, int H, int W, int in_buf,
                   int in_choff, int out_buf, int out_choff, int C) {
  Op o;
  o.kind = kind;
  o.name = name;
  o.H = H;
  o.W = W;
  o.in_buf[0] = in_buf;
  o.in_choff[0] = in_choff;
  o.out_buf[0] = out_buf;
  o.out_choff[0] = out_choff;
  o.C = C;
  n->ops.push_back(o);
}

// conv (+pool) writing into a buffer at the pooled or the same resolution
void plan_vgg_conv(rtpose_net* n, int conv, int H, int W, int in_buf, int out_buf_same,
                   int out_buf_pooled) {
  const int zero = 0;
  if (out_buf_pooled < 0) {
    add_conv_op(n, H, W, 1, &conv, &in_buf, &zero, &out_buf_same, &zero, 1, 0);
  } else if (!((H | W) & 1)) {
    add_conv_op(n, H, W, 1, &conv, &in_buf, &zero, &out_buf_pooled, &zero, 1, 1);
  } else {
    add_conv_op(n, H, W, 1, &conv, &in_buf, &zero, &out_buf_same, &zero, 1, 0);
    add_simple_op(n, OP_POOL, "pool", H, W, out_buf_same, 0, out_buf_pooled, 0,
                  n->bufs[out_buf_same].C);
  }
}

void build_plan(rtpose_net* n) {
  const int N = n->N;
  (void)N;
  const int H0 = n->H, W0 = n->W;
  const int H1 = H0 / 2, W1 = W0 / 2, H2 = H1 / 2, W2 = W1 / 2, H3 = H2 / 2, W3 = W2 / 2;
  n->H3 = H3;
  n->W3 = W3;

  // ---- weights, in the reference's state_dict order (rtpose_vgg.py:141-155) ----
  // model0: VGG19 first 10 convs + 2 CPM convs (rtpose_vgg.py:69-83)
  const int vgg_idx[12] = {0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25};
  const int vgg_cin[12] = {3, 64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 256};
  const int vgg_cout[12] = {64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 256, 128};
  int cw0[12];
  for (int i = 0; i < 12; ++i)
    cw0[i] = add_conv_w(n, "model0." + std::to_string(vgg_idx[i]), vgg_cout[i], vgg_cin[i], 3, false);
  // stages: branch 1 (PAF, 38) for stages 1..6, then branch 2 (heat, 19)
  int cw1[2][5], cws[2][5][7];
  for (int b = 0; b < 2; ++b) {
    const int last = b == 0 ? 38 : 19;
    const std::string sfx = "_" + std::to_string(b + 1) + ".";
    // stage 1 (rtpose_vgg.py:95-105)
    for (int i = 0; i < 5; ++i) {
      const int cin = i < 4 ? 128 : 512;
      const int cout = i < 3 ? 128 : (i == 3 ? 512 : last);
      const int k = i < 3 ? 3 : 1;
      cw1[b][i] = add_conv_w(n, "model1" + sfx + std::to_string(2 * i), cout, cin, k, false);
    }
    // stages 2..6 (rtpose_vgg.py:108-127)
    for (int s = 2; s <= 6; ++s)
      for (int i = 0; i < 7; ++i) {
        const int cin = i == 0 ? 185 : 128;
        const int cout = i < 6 ? 128 : last;
        const int k = i < 5 ? 7 : 1;
        cws[b][s - 2][i] = add_conv_w(n, "model" + std::to_string(s) + sfx + std::to_string(2 * i),
                                      cout, cin, k, i == 0, H3, W3);
      }
  }
  n->catmap_off = n->wt_floats;
  n->wt_floats += 256;  // int32[192]

  // ---- activation buffers ------------------------------------------------------
  const int X0 = add_buf(n, n->bf16 ? 16 : 8, 1, H0, W0);
  n->x0_buf = X0;
  if (n->bf16) n->x0f_buf = add_buf(n, 8, 1, H0, W0, true);
  const int A1 = add_buf(n, 64, 1, H0, W0);
  const bool even0 = !((H0 | W0) & 1), even1 = !((H1 | W1) & 1), even2 = !((H2 | W2) & 1);
  const int A2 = even0 ? -1 : add_buf(n, 64, 0, H0, W0);
  const int B0 = add_buf(n, 64, 1, H1, W1);
  const int B1 = add_buf(n, 128, 1, H1, W1);
  const int B2 = even1 ? -1 : add_buf(n, 128, 0, H1, W1);
  const int C0 = add_buf(n, 128, 1, H2, W2);
  const int C1 = add_buf(n, 256, 1, H2, W2);
  const int C2 = add_buf(n, 256, 1, H2, W2);
  const int C3 = add_buf(n, 256, 1, H2, W2);
  const int C4 = even2 ? -1 : add_buf(n, 256, 0, H2, W2);
  const int D0 = add_buf(n, 256, 1, H3, W3);
  const int D1 = add_buf(n, 512, 1, H3, W3);
  const int D2 = add_buf(n, 512, 1, H3, W3);
  const int D3 = add_buf(n, 256, 1, H3, W3);
  const int CATa = add_buf(n, kCatC, 3, H3, W3);
  const int CATb = add_buf(n, kCatC, 3, H3, W3);
  n->cat_buf[0] = CATa;
  n->cat_buf[1] = CATb;
  int T1[2], T2[2], T3[2], T4[2], U[2][6];
  for (int b = 0; b < 2; ++b) {
    T1[b] = add_buf(n, 128, 1, H3, W3);
    T2[b] = add_buf(n, 128, 1, H3, W3);
    T3[b] = add_buf(n, 128, 0, H3, W3);
    // (fp32 and bf16 plans run the trailing 1x1 pairs back to back, conv_tail.hip / conv_tail_bf16.hip: their intermediates
    //  never reach HBM; the split plan keeps them)
    T4[b] = n->split ? add_buf(n, 512, 0, H3, W3) : -1;
    for (int i = 0; i < 4; ++i) U[b][i] = add_buf(n, 128, 3, H3, W3);
    U[b][4] = add_buf(n, 128, 0, H3, W3);
    U[b][5] = n->split ? add_buf(n, 128, 0, H3, W3) : -1;
  }
  for (int s = 0; s < 6; ++s) n->save_buf[s] = add_buf(n, 57, 0, H3, W3, true);  // always fp32
  if (!n->bf16) {
    // hand-over scratch of the persistent 7x7 launches (conv_wino7.hip): part of the workspace, so that the
    // forward allocates nothing, rtpose_net_workspace_bytes tells the whole truth and the launch list can be
    // stream-captured.  One scratch: the launches of a plan are serialised on one stream.
    n->scratch_off = round_up(n->ws_floats, 64);
    n->scratch_bytes = conv2d_wino7_scratch_bytes(n->n_cu);
    n->ws_floats = n->scratch_off + round_up(n->scratch_bytes / 4, 64);
  }

  // ---- launches ------------------------------------------------------------------
  add_simple_op(n, OP_INPUT, "nchw_to_nhwc8", H0, W0, -1, 0, X0, 0, 3);
  plan_vgg_conv(n, cw0[0], H0, W0, X0, A1, -1);
  plan_vgg_conv(n, cw0[1], H0, W0, A1, A2, B0);
  plan_vgg_conv(n, cw0[2], H1, W1, B0, B1, -1);
  plan_vgg_conv(n, cw0[3], H1, W1, B1, B2, C0);
  plan_vgg_conv(n, cw0[4], H2, W2, C0, C1, -1);
  plan_vgg_conv(n, cw0[5], H2, W2, C1, C2, -1);
  plan_vgg_conv(n, cw0[6], H2, W2, C2, C3, -1);
  plan_vgg_conv(n, cw0[7], H2, W2, C3, C4, D0);
  plan_vgg_conv(n, cw0[8], H3, W3, D0, D1, -1);
  plan_vgg_conv(n, cw0[9], H3, W3, D1, D2, -1);
  plan_vgg_conv(n, cw0[10], H3, W3, D2, D3, -1);
  {  // conv4_4_CPM -> out1, written once into each concat buffer
    const int zero = 0;
    add_conv_op(n, H3, W3, 1, &cw0[11], &D3, &zero, &CATa, &zero, 1, 0);
    add_simple_op(n, OP_COPY, "out1->cat_b", H3, W3, CATa, 0, CATb, 0, 128);
  }
  const int zz[2] = {0, 0};
  const int head_off[2] = {kCatPaf, kCatHeat};
  {  // stage 1: reads out1 = CATa[0:128]; heads write CATb
    const int in0[2] = {CATa, CATa};
    int ci[2];
    ci[0] = cw1[0][0]; ci[1] = cw1[1][0];
    add_conv_op(n, H3, W3, 2, ci, in0, zz, T1, zz, 1, 0);
    ci[0] = cw1[0][1]; ci[1] = cw1[1][1];
    add_conv_op(n, H3, W3, 2, ci, T1, zz, T2, zz, 1, 0);
    ci[0] = cw1[0][2]; ci[1] = cw1[1][2];
    add_conv_op(n, H3, W3, 2, ci, T2, zz, T3, zz, 1, 0);
    const int outb[2] = {CATb, CATb};
    if (!n->split) {
      // fp32 and bf16: conv5_4_CPM (128 -> 512, ReLU) + conv5_5_CPM (512 -> 38 | 19) as one back-to-back launch
      // (conv_tail.hip / conv_tail_bf16.hip); the split (bf16x3) plan keeps two launches of its generic kernel
      ci[0] = cw1[0][4]; ci[1] = cw1[1][4];
      add_conv_op(n, H3, W3, 2, ci, T3, zz, outb, head_off, 0, 0);
      Op& t = n->ops.back();
      t.kind = OP_TAIL;
      t.ks = 1;
      for (int b = 0; b < 2; ++b) {
        t.conv2_idx[b] = t.conv_idx[b];
        t.conv_idx[b] = cw1[b][3];
        const ConvW& c1 = n->convs[t.conv_idx[b]];
        t.flops += 2.0 * n->N * H3 * W3 * (double)c1.cout * c1.cin_src;
      }
      t.name = n->convs[t.conv_idx[0]].name + "+" + n->convs[t.conv2_idx[0]].name + "|" +
               n->convs[t.conv_idx[1]].name + "+" + n->convs[t.conv2_idx[1]].name;
    } else {
      ci[0] = cw1[0][3]; ci[1] = cw1[1][3];
      add_conv_op(n, H3, W3, 2, ci, T3, zz, T4, zz, 1, 0);
      ci[0] = cw1[0][4]; ci[1] = cw1[1][4];
      add_conv_op(n, H3, W3, 2, ci, T4, zz, outb, head_off, 0, 0);
    }
    add_simple_op(n, OP_COPY, "save1", H3, W3, CATb, kCatPaf, n->save_buf[0], 0, 57);
  }
  for (int s = 2; s <= 6; ++s) {
    const int cin_buf = (s % 2 == 0) ? CATb : CATa;
    const int cout_buf = (s % 2 == 0) ? CATa : CATb;
    const int in0[2] = {cin_buf, cin_buf};
    int ci[2];
    int u0[2] = {U[0][0], U[1][0]};
    ci[0] = cws[0][s - 2][0]; ci[1] = cws[1][s - 2][0];
    add_conv_op(n, H3, W3, 2, ci, in0, zz, u0, zz, 1, 0);
    for (int i = 1; i < (n->split ? 6 : 5); ++i) {
      const int ui[2] = {U[0][i - 1], U[1][i - 1]};
      const int uo[2] = {U[0][i], U[1][i]};
      ci[0] = cws[0][s - 2][i]; ci[1] = cws[1][s - 2][i];
      add_conv_op(n, H3, W3, 2, ci, ui, zz, uo, zz, 1, 0);
    }
    if (!n->split) {
      // fp32 and bf16: Mconv6 (128 -> 128, ReLU) + Mconv7 (128 -> 38 | 19) of both branches as ONE back-to-back launch
      // (conv_tail.hip / conv_tail_bf16.hip): the 128-channel intermediate never leaves the CU.  The bf16 plan's last
      // stage writes the fp32 record the decoder and the TTA merge read (no bf16 concat slice, no save copy).
      const bool last_bf16 = n->bf16 && s == 6;
      const int ui4[2] = {U[0][4], U[1][4]};
      const int outb[2] = {last_bf16 ? n->save_buf[5] : cout_buf, last_bf16 ? n->save_buf[5] : cout_buf};
      const int off57[2] = {0, 38};
      ci[0] = cws[0][s - 2][6]; ci[1] = cws[1][s - 2][6];
      add_conv_op(n, H3, W3, 2, ci, ui4, zz, outb, last_bf16 ? off57 : head_off, 0, 0);
      Op& t = n->ops.back();
      t.out_f32 = last_bf16 ? 1 : 0;
      t.kind = OP_TAIL;
      t.ks = 1;
      for (int b = 0; b < 2; ++b) {
        t.conv2_idx[b] = t.conv_idx[b];
        t.conv_idx[b] = cws[b][s - 2][5];
        const ConvW& c1 = n->convs[t.conv_idx[b]];
        t.flops += 2.0 * n->N * H3 * W3 * (double)c1.cout * c1.cin_src;
      }
      t.name = n->convs[t.conv_idx[0]].name + "+" + n->convs[t.conv2_idx[0]].name + "|" +
               n->convs[t.conv_idx[1]].name + "+" + n->convs[t.conv2_idx[1]].name;
      if (!last_bf16)
        add_simple_op(n, OP_COPY, "save" + std::to_string(s), H3, W3, cout_buf, kCatPaf, n->save_buf[s - 1], 0, 57);
      continue;
    }
    const int ui[2] = {U[0][5], U[1][5]};
    ci[0] = cws[0][s - 2][6]; ci[1] = cws[1][s - 2][6];
    if (n->bf16 && s == 6) {
      // the decoder and the TTA merge read fp32: the last heads skip the bf16 concat buffer
      const int outb[2] = {n->save_buf[5], n->save_buf[5]};
      const int off57[2] = {0, 38};
      add_conv_op(n, H3, W3, 2, ci, ui, zz, outb, off57, 0, 0);
      n->ops.back().out_f32 = 1;
      continue;
    }
    const int outb[2] = {cout_buf, cout_buf};
    add_conv_op(n, H3, W3, 2, ci, ui, zz, outb, head_off, 0, 0);
    add_simple_op(n, OP_COPY, "save" + std::to_string(s), H3, W3, cout_buf, kCatPaf,
                  n->save_buf[s - 1], 0, 57);
  }
}

rtpose_layout slice(const Buf& b, int choff) {
  rtpose_layout l = b.lay;
  l.choff = choff;
  return l;
}

}  // namespace

extern "C" {

int rtpose_net_create_opts(int N, int H, int W, const rtpose_net_options* opt, rtpose_net** out) {
  if (!out) return fail(RTPOSE_E_INVAL, "net_create: out is NULL");
  if (!opt || opt->struct_bytes < sizeof(rtpose_net_options))
    return fail(RTPOSE_E_INVAL, "net_create: options missing or struct_bytes smaller than this library's rtpose_net_options");
  const int dtype = opt->dtype;
  if (N <= 0 || H < 8 || W < 8) return fail(RTPOSE_E_INVAL, "net_create: need N>=1 and H,W>=8");
  if (dtype != RTPOSE_DTYPE_F32 && dtype != RTPOSE_DTYPE_BF16 && dtype != RTPOSE_DTYPE_BF16X3)
    return fail(RTPOSE_E_INVAL, "net_create: dtype must be RTPOSE_DTYPE_F32, _BF16 or _BF16X3");
  if (dtype != RTPOSE_DTYPE_F32 && ((H | W) & 7))
    return fail(RTPOSE_E_INVAL, "net_create: the bf16 plan needs H and W to be multiples of 8 "
                                "(crop_with_factor pads to that, im_transform.py:128-132)");
  if (opt->winograd3 != RTPOSE_WINO_DEFAULT && opt->winograd3 != 0 && opt->winograd3 != 1 && opt->winograd3 != 4 &&
      opt->winograd3 != RTPOSE_WINO3_AUTO)
    return fail(RTPOSE_E_INVAL, "net_create: winograd3 must be RTPOSE_WINO_DEFAULT, 0, 1, 4 or RTPOSE_WINO3_AUTO");
  if (opt->winograd7 != RTPOSE_WINO_DEFAULT && opt->winograd7 != 0 && opt->winograd7 != 4 && opt->winograd7 != 6 &&
      opt->winograd7 != RTPOSE_WINO7_AUTO)
    return fail(RTPOSE_E_INVAL, "net_create: winograd7 must be RTPOSE_WINO_DEFAULT, 0, 4, 6 or RTPOSE_WINO7_AUTO");
  rtpose_net* n = new rtpose_net();
  n->N = N;
  n->H = H;
  n->W = W;
  n->bf16 = dtype != RTPOSE_DTYPE_F32;
  n->split = dtype == RTPOSE_DTYPE_BF16X3;
  n->n_cu = device_cu_count();
  {
    // defaults of the two fields: on, unless the environment of the process says otherwise (RTPOSE_WINOGRAD =
    // 0: direct kernels everywhere, 3 / 7: only that kernel size in Winograd form; RTPOSE_WINOGRAD7_M=4: F(4,7))
    const char* e = getenv("RTPOSE_WINOGRAD");
    const int env = !e ? 1 : e[0] == '0' ? 0 : e[0] == '3' ? 3 : e[0] == '7' ? 7 : 1;
    // (RTPOSE_WINOGRAD3_M=2: F(2x2,3x3) instead of F(4x4,3x3))
    const char* e3 = getenv("RTPOSE_WINOGRAD3_M");
    // Round 4: the default is the GUARDED choice - per layer, the fastest form whose amplification estimate for the
    // filters actually loaded stays under amp_limit (256): nobody here has seen pose_model.pth (README.md:19), and a
    // forced F(6,7) / F(4x4,3x3) would run whatever it holds.  He-init / N(0, 0.01) filters estimate 115-120 and
    // 42-43, so the bench plan keeps its forms bit for bit; RTPOSE_WINOGRAD3_M / RTPOSE_WINOGRAD7_M force a form.
    const char* e7 = getenv("RTPOSE_WINOGRAD7_M");
    n->w3 = opt->winograd3 != RTPOSE_WINO_DEFAULT ? opt->winograd3
            : (env == 1 || env == 3)              ? ((e3 && e3[0] == '2')   ? 1
                                                     : (e3 && e3[0] == '4') ? 4
                                                                            : RTPOSE_WINO3_AUTO)
                                                  : 0;
    n->w7 = opt->winograd7 != RTPOSE_WINO_DEFAULT ? opt->winograd7
            : (env == 1 || env == 7)              ? ((e7 && (e7[0] == '4' || e7[0] == '6')) ? wino7_default_fm()
                                                                                           : RTPOSE_WINO7_AUTO)
                                                  : 0;
    n->amp_limit = opt->amp_limit > 0.f ? opt->amp_limit : 256.f;
    // RTPOSE_W7_PERSIST=0 in the environment of the process: plans start with the split-tile launches off
    // (rtpose_net_set_persistent7 changes it per plan)
    const char* ep = getenv("RTPOSE_W7_PERSIST");
    n->persist7 = (ep && ep[0] == '0') ? 0 : 1;
  }
  build_plan(n);
  if (!forms_need_amps(n)) {  // AUTO waits for the filters (rtpose_net_finalize_weights)
    pick_forms(n);
    mark_plane_bufs(n);
    n->forms_final = true;
  }
  *out = n;
  return 0;
}

int rtpose_net_create_ex(int N, int H, int W, int dtype, rtpose_net** out) {
  rtpose_net_options o;
  o.struct_bytes = sizeof(o);
  o.dtype = dtype;
  o.winograd3 = RTPOSE_WINO_DEFAULT;
  o.winograd7 = RTPOSE_WINO_DEFAULT;
  o.amp_limit = 0.f;
  return rtpose_net_create_opts(N, H, W, &o, out);
}

int rtpose_net_create(int N, int H, int W, rtpose_net** out) {
  return rtpose_net_create_ex(N, H, W, RTPOSE_DTYPE_F32, out);
}

int rtpose_net_dtype(const rtpose_net* net) {
  return !net || !net->bf16 ? RTPOSE_DTYPE_F32 : (net->split ? RTPOSE_DTYPE_BF16X3 : RTPOSE_DTYPE_BF16);
}

void rtpose_net_destroy(rtpose_net* net) {
  if (!net) return;
  for (hipEvent_t e : net->ev) (void)hipEventDestroy(e);
  for (hipGraphExec_t g : net->gexec)
    if (g) (void)hipGraphExecDestroy(g);
  if (net->gev_in) (void)hipEventDestroy(net->gev_in);
  if (net->gev_out) (void)hipEventDestroy(net->gev_out);
  if (net->gstream) (void)hipStreamDestroy(net->gstream);
  delete net;
}

size_t rtpose_net_workspace_bytes(const rtpose_net* net) { return net->ws_floats * sizeof(float); }
size_t rtpose_net_weight_bytes(const rtpose_net* net) { return net->wt_floats * sizeof(float); }

int rtpose_net_bind(rtpose_net* net, void* workspace, size_t workspace_bytes, void* weights,
                    size_t weight_bytes, int zero_workspace, void* stream) {
  if (!net || !workspace || !weights) return fail(RTPOSE_E_INVAL, "net_bind: NULL argument");
  if (workspace_bytes < rtpose_net_workspace_bytes(net) || weight_bytes < rtpose_net_weight_bytes(net))
    return fail(RTPOSE_E_INVAL, "net_bind: arena too small");
  if (((uintptr_t)workspace | (uintptr_t)weights) & 255)
    return fail(RTPOSE_E_INVAL, "net_bind: arenas must be 256-byte aligned");
  const int dev = current_device();
  int rcd = check_device_ptr(workspace, dev, "net_bind", "the workspace");
  if (!rcd) rcd = check_device_ptr(weights, dev, "net_bind", "the weight arena");
  if (rcd) return rcd;
  if (device_cu_count() != net->n_cu)
    return fail(RTPOSE_E_STATE, "net_bind: the plan was created for a device of %d CUs, the current device %d has %d "
                                "(create the plan with that device current)", net->n_cu, dev, device_cu_count());
  net->device = dev;
  net->in_checked = CheckedPtr();
  for (hipGraphExec_t& g : net->gexec) {  // captured pointers are about to change
    if (g) (void)hipGraphExecDestroy(g);
    g = nullptr;
  }
  net->forwards = 0;
  // a zeroed workspace suits either storage of a buffer; a caller-kept one is taken to hold pixel-major data with clean gaps
  net->zeroed_at_bind = zero_workspace != 0;
  for (Buf& b : net->bufs) b.stale = !zero_workspace && b.plane_px != 0;
  net->amps_read = false;
  net->seen_gen = ~0ull;
  if (forms_need_amps(net)) net->forms_final = false;
  net->ws = static_cast<float*>(workspace);
  net->wt = static_cast<float*>(weights);
  hipStream_t s = as_stream(stream);
  if (zero_workspace) RTPOSE_HIP_CHECK(hipMemsetAsync(workspace, 0, rtpose_net_workspace_bytes(net), s));
  // the hand-over flags and the device error word of the persistent 7x7 launches must start at zero whatever the
  // caller says about the rest of the workspace
  else if (!net->bf16 && net->scratch_bytes)
    RTPOSE_HIP_CHECK(hipMemsetAsync(net->ws + net->scratch_off, 0,
                                    (size_t)((char*)(conv2d_wino7_scratch_err(net->ws + net->scratch_off, net->n_cu) + 1) -
                                             (char*)(net->ws + net->scratch_off)), s));
  // channel map of the concat input: packed c -> source channel of cat([L1,L2,out1])
  int32_t map[kCatC];
  for (int c = 0; c < kCatC; ++c) {
    if (c < 128) map[c] = 57 + c;
    else if (c < kCatHeat) map[c] = c - kCatPaf;
    else if (c < 185) map[c] = 38 + (c - kCatHeat);
    else map[c] = -1;
  }
  RTPOSE_HIP_CHECK(hipMemcpyAsync(net->wt + net->catmap_off, map, sizeof(map), hipMemcpyHostToDevice, s));
  RTPOSE_HIP_CHECK(hipStreamSynchronize(s));  // `map` is a stack buffer
  net->bound = true;
  return 0;
}

int rtpose_net_num_convs(const rtpose_net* net) { return (int)net->convs.size(); }

int rtpose_net_conv_info(const rtpose_net* net, int idx, char* name, int name_cap, int* cout, int* cin,
                         int* k) {
  if (!net || idx < 0 || idx >= (int)net->convs.size()) return fail(RTPOSE_E_INVAL, "conv_info: bad index");
  const ConvW& c = net->convs[idx];
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", c.name.c_str());
  if (cout) *cout = c.cout;
  if (cin) *cin = c.cin_src;
  if (k) *k = c.k;
  return 0;
}

static int net_on_its_device(const rtpose_net* net, const char* who);

int rtpose_net_load_conv(rtpose_net* net, int idx, const float* w_oihw, const float* bias, void* stream) {
  if (!net || !net->bound) return fail(RTPOSE_E_STATE, "net_load_conv: net not bound");
  if (idx < 0 || idx >= (int)net->convs.size()) return fail(RTPOSE_E_INVAL, "net_load_conv: bad index");
  int rcd = net_on_its_device(net, "net_load_conv");
  if (!rcd) rcd = check_device_ptr(w_oihw, net->device, "net_load_conv", "the filter tensor");
  if (!rcd && bias) rcd = check_device_ptr(bias, net->device, "net_load_conv", "the bias tensor");
  if (rcd) return rcd;
  const ConvW& c = net->convs[idx];
  const int32_t* map = c.cat_perm ? reinterpret_cast<const int32_t*>(net->wt + net->catmap_off) : nullptr;
  hipStream_t s = as_stream(stream);
  if (net->bf16) {
    if (c.first) {
      const int rcf = conv_first_pack_launch(w_oihw, bias, net->wt + c.w_off_first, s, 1);
      if (rcf) return rcf;
    }
    return pack_weights_bf16_launch(w_oihw, bias, c.cout, c.cin_src, c.k, map, c.cin_packed,
                                    net->wt + c.w_off, net->wt + c.b_off, net->split, s);
  }
  // every packing the arena holds for this conv (the plans that share the arena choose among them), and the
  // amplification estimate of each Winograd form; every plan on this arena re-reads them (sync_arena_generation)
  arena_bump(net->wt);
  net->amps_read = false;
  if (forms_need_amps(net)) net->forms_final = false;
  int rc = pack_weights_launch(w_oihw, bias, c.cout, c.cin_src, c.k, map, c.cin_packed, net->wt + c.w_off,
                               net->wt + c.b_off, s);
  if (rc) return rc;
  if (c.first) {
    rc = conv_first_pack_launch(w_oihw, bias, net->wt + c.w_off_first, s, 0);
    if (rc) return rc;
  }
  float* amp = net->wt + c.amp_off;
  RTPOSE_HIP_CHECK(hipMemsetAsync(amp, 0, 4 * sizeof(float), s));
  if (c.has_w3) {
    rc = rtpose_pack_conv_weights_winograd(w_oihw, bias, c.cout, c.cin_src, 3, map, c.cin_packed,
                                           net->wt + c.w_off_w3, net->wt + c.b_off, stream);
    if (!rc) rc = wino_amplification_launch(w_oihw, c.cout, c.cin_src, 3, 0, amp + 0, s);
    if (rc) return rc;
  }
  if (c.has_w43) {
    rc = rtpose_pack_conv_weights_winograd3(w_oihw, bias, c.cout, c.cin_src, 4, map, c.cin_packed,
                                            net->wt + c.w_off_w43, net->wt + c.b_off, stream);
    if (!rc) rc = rtpose_winograd_amplification(w_oihw, c.cout, c.cin_src, 3, 4, amp + 3, stream);
    if (rc) return rc;
  }
  if (c.has_w7) {
    rc = pack_weights_wino7_launch(w_oihw, bias, c.cout, c.cin_src, map, c.cin_packed, 4, net->wt + c.w_off_w4,
                                   net->wt + c.b_off, s);
    if (!rc)
      rc = pack_weights_wino7_launch(w_oihw, bias, c.cout, c.cin_src, map, c.cin_packed, 6, net->wt + c.w_off_w6,
                                     net->wt + c.b_off, s);
    if (!rc) rc = wino_amplification_launch(w_oihw, c.cout, c.cin_src, 7, 4, amp + 1, s);
    if (!rc) rc = wino_amplification_launch(w_oihw, c.cout, c.cin_src, 7, 6, amp + 2, s);
    if (rc) return rc;
  }
  return 0;
}

static int read_amps(rtpose_net* net, hipStream_t s) {
  if (net->amps_read || net->bf16) return 0;
  // the estimates are written by the pack kernels of rtpose_net_load_conv on whatever stream THAT call was given -
  // possibly a sibling plan's, another stream than `s` (the arena is shared by the plans of a module).  Once per
  // weight load, so the whole device is drained rather than an event kept per arena.
  RTPOSE_HIP_CHECK(hipDeviceSynchronize());
  // one contiguous read-back of the arena span that holds the estimates would drag the packed filters along;
  // 92 small copies once per weight load are cheaper
  for (ConvW& c : net->convs)
    RTPOSE_HIP_CHECK(hipMemcpyAsync(c.amp, net->wt + c.amp_off, 4 * sizeof(float), hipMemcpyDeviceToHost, s));
  RTPOSE_HIP_CHECK(hipStreamSynchronize(s));
  net->amps_read = true;
  return 0;
}

int rtpose_net_finalize_weights(rtpose_net* net, void* stream) {
  if (!net || !net->bound) return fail(RTPOSE_E_STATE, "net_finalize_weights: net not bound");
  sync_arena_generation(net);
  if (net->forms_final) return 0;
  const int rc = read_amps(net, as_stream(stream));
  if (rc) return rc;
  pick_forms(net);
  mark_plane_bufs(net);
  net->forms_final = true;
  for (hipGraphExec_t& g : net->gexec) {  // a captured launch list may hold other forms
    if (g) (void)hipGraphExecDestroy(g);
    g = nullptr;
  }
  return 0;
}

int rtpose_net_conv_numerics(rtpose_net* net, int idx, int* form, float* amp, void* stream) {
  if (!net || !net->bound) return fail(RTPOSE_E_STATE, "net_conv_numerics: net not bound");
  if (idx < 0 || idx >= (int)net->convs.size()) return fail(RTPOSE_E_INVAL, "net_conv_numerics: bad index");
  int rc = rtpose_net_finalize_weights(net, stream);  // (notices a reload made through a sibling plan)
  if (!rc && amp) rc = read_amps(net, as_stream(stream));
  if (rc) return rc;
  const ConvW& c = net->convs[idx];
  if (form) *form = c.form;
  if (amp)
    for (int i = 0; i < 4; ++i) amp[i] = c.amp[i];
  return 0;
}

int rtpose_net_graph_active(const rtpose_net* net) {
  return net && net->graph_mode == 1 && (net->gexec[0] || net->gexec[1]) ? 1 : 0;
}

int rtpose_net_device_status(rtpose_net* net, int* error_word, void* stream) {
  if (!net || !net->bound || !error_word) return fail(RTPOSE_E_STATE, "net_device_status: net not bound / NULL argument");
  *error_word = 0;
  if (net->bf16 || !net->scratch_bytes) return 0;
  hipStream_t s = as_stream(stream);
  int* err = conv2d_wino7_scratch_err(net->ws + net->scratch_off, net->n_cu);
  RTPOSE_HIP_CHECK(hipMemcpyAsync(error_word, err, sizeof(int), hipMemcpyDeviceToHost, s));
  RTPOSE_HIP_CHECK(hipMemsetAsync(err, 0, sizeof(int), s));
  RTPOSE_HIP_CHECK(hipStreamSynchronize(s));
  // a hand-over wait ran out: this device does not dispatch the grid the way the split tiles assume (a CU mask, a
  // co-tenant that starves it).  The maps of that forward are invalid - the caller is told - and the plan stops
  // splitting tiles: every later forward runs one block per tile (same results, bit for bit, a few per cent slower).
  if (*error_word & 1) return rtpose_net_set_persistent7(net, 0);  // (also drops captured launch lists: they hold the split-tile grids)
  return 0;
}

static void decide_guard_launch(rtpose_net* net);

int rtpose_net_output_guard_launch(rtpose_net* net) {
  if (!net) return fail(RTPOSE_E_INVAL, "net_output_guard_launch: NULL net");
  decide_guard_launch(net);
  return net->guard_op;
}

int rtpose_net_set_output_guard(rtpose_net* net, void* hip_event) {
  if (!net) return fail(RTPOSE_E_INVAL, "net_set_output_guard: NULL net");
  decide_guard_launch(net);
  net->out_guard = static_cast<hipEvent_t>(hip_event);
  return 0;
}

static void decide_guard_launch(rtpose_net* net) {
  if (net->guard_op < 0) {
    // Where a forward waits for the reader of its previous maps (the decoder of the batch before on a second stream,
    // pipeline.SideDecoder).  Default, every arithmetic: in front of the FIRST launch that writes the buffer
    // rtpose_net_output_view hands out - fp32: CATa, first written by conv4_4_CPM (its out1 channels), so the reader runs beside
    // the trunk's ~8 ms; bf16 / bf16x3: the stage-6 record, written by the last launch only.  RTPOSE_GUARD_WHOLE_FORWARD=1 (or
    // RTPOSE_GUARD_FINE=0): in FRONT of the launch list - the reader never runs beside this plan's kernels (costs 0.9 % of an
    // fp32 step, 1.1 % of a bf16 one).  History: in round 5 a decoder beside the bf16 plan's kernels returned a
    // limb score one sample off in ~1 % of the batches and bf16 plans waited in front; round 6 traced it to the packed-fp32
    // VALU instructions clang's SLP vectoriser had put into limb_assign_kernel's sample loop (wrong values in lanes 48..63
    // when the wave shares a CU with those kernels), the decoder is built without them (csrc/Makefile) and gave 0 differing
    // records in 240,000 decodes beside the bf16 forward on the box where the old build gave 342 in 48,000.
    const int target = net->bf16 ? net->save_buf[5] : net->cat_buf[0];
    net->guard_op = 0;
    for (size_t i = 0, found = 0; i < net->ops.size() && !found; ++i)
      for (int g = 0; g < 2; ++g)
        if (net->ops[i].out_buf[g] == target) {
          net->guard_op = (int)i;
          found = 1;
        }
    const char* e = getenv("RTPOSE_GUARD_WHOLE_FORWARD");
    const char* f = getenv("RTPOSE_GUARD_FINE");
    if ((e && e[0] == '1') || (f && f[0] == '0')) net->guard_op = 0;
#ifdef RTPOSE_DEV_BUILD
    if (const char* o = getenv("RTPOSE_GUARD_OP")) {  // experiments: the wait in front of launch <n> (negative: from the end)
      const int n = atoi(o), nops = (int)net->ops.size();
      net->guard_op = n < 0 ? (nops + n > 0 ? nops + n : 0) : (n < nops ? n : nops - 1);
    }
#endif
  }
}

int rtpose_net_set_persistent7(rtpose_net* net, int enable) {
  if (!net) return fail(RTPOSE_E_INVAL, "net_set_persistent7: NULL net");
  net->persist7 = enable ? 1 : 0;
  for (hipGraphExec_t& g : net->gexec) {  // a captured launch list holds the other grids
    if (g) (void)hipGraphExecDestroy(g);
    g = nullptr;
  }
  return 0;
}

int rtpose_net_persistent7(const rtpose_net* net) { return net && net->persist7 ? 1 : 0; }

int rtpose_net_device_status_async(rtpose_net* net, int* host_word, void* stream) {
  if (!net || !net->bound || !host_word) return fail(RTPOSE_E_STATE, "net_device_status_async: net not bound / NULL argument");
  if (net->bf16 || !net->scratch_bytes) {
    *host_word = 0;
    return 0;
  }
  hipStream_t s = as_stream(stream);
  int* err = conv2d_wino7_scratch_err(net->ws + net->scratch_off, net->n_cu);
  RTPOSE_HIP_CHECK(hipMemcpyAsync(host_word, err, sizeof(int), hipMemcpyDeviceToHost, s));
  RTPOSE_HIP_CHECK(hipMemsetAsync(err, 0, sizeof(int), s));
  return 0;
}

int rtpose_net_set_keep_intermediates(rtpose_net* net, int keep) {
  if (!net) return fail(RTPOSE_E_INVAL, "NULL net");
  net->keep = keep ? 1 : 0;
  return 0;
}

int rtpose_net_set_profiling(rtpose_net* net, int enable) {
  if (!net) return fail(RTPOSE_E_INVAL, "NULL net");
  net->profiling = enable ? 1 : 0;
  if (enable && net->ev.empty()) {
    net->ev.resize(net->ops.size() + 1);
    for (auto& e : net->ev) RTPOSE_HIP_CHECK(hipEventCreate(&e));
  }
  net->ev_valid = false;
  return 0;
}

int rtpose_net_num_launches(const rtpose_net* net) { return (int)net->ops.size(); }

int rtpose_net_launch_executed_flops(const rtpose_net* net, int i, double* flops, int* winograd) {
  if (!net || i < 0 || i >= (int)net->ops.size()) return fail(RTPOSE_E_INVAL, "launch_executed_flops: bad index");
  const Op& o = net->ops[i];
  double fl = 0.0;
  int wino = 0;
  if (o.kind == OP_TAIL)
    for (int g = 0; g < o.ngroups; ++g)
      for (int ci : {o.conv_idx[g], o.conv2_idx[g]}) {
        const ConvW& c = net->convs[ci];
        fl += 2.0 * net->N * o.H * o.W * (double)c.cin_packed * cout_pad(c.cout);
      }
  if (o.kind == OP_CONV)
    for (int g = 0; g < o.ngroups; ++g) {
      // what the matrix pipe is issued (SQ_INSTS_MFMA x 4096 of a launch): whole tiles, padded channels and columns
      const ConvW& c = net->convs[o.conv_idx[g]];
      if (c.first) {  // conv_first_kernel (fp32 and bf16 plans: the fp32 matrix instruction either way): 8 x 32 pixel tiles, K = 28 (27 taps + a zero row), 64 columns
        fl += 2.0 * net->N * ceil_div(o.H, 8) * ceil_div(o.W, 32) * 256.0 * 28.0 * 64.0;
      } else if (c.form == 3) {         // 16 frequencies per 2 x 2 wtile
        fl += conv2d_wino_issued_flops(c.cin_packed, c.cout, net->N, o.H, o.W);
        wino = 3;
      } else if (c.form == 43) {  // 36 frequencies per 4 x 4 wtile
        fl += conv2d_wino4_issued_flops(c.cin_packed, c.cout, net->N, o.H, o.W);
        wino = 43;
      } else if (c.form) {       // FM + 6 frequencies x 7 rows per group of FM pixels, 32-position strips per image
        fl += conv2d_wino7_issued_flops(c.cin_packed, c.cout, net->N, o.H, o.W, o.H + 3, c.form);
        wino = c.form;
      } else {
        fl += 2.0 * net->N * o.H * o.W * (double)c.k * c.k * (double)c.cin_packed * cout_pad(c.cout);
      }
    }
  if (flops) *flops = fl;
  if (winograd) *winograd = wino;
  return 0;
}

int rtpose_net_launch_info(rtpose_net* net, int i, float* ms, int* k, double* flops, char* name,
                           int name_cap) {
  if (!net || i < 0 || i >= (int)net->ops.size()) return fail(RTPOSE_E_INVAL, "launch_info: bad index");
  const Op& o = net->ops[i];
  if (k) *k = (o.kind == OP_CONV || o.kind == OP_TAIL) ? o.ks : 0;
  if (flops) *flops = o.flops;
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", o.name.c_str());
  if (ms) {
    *ms = -1.f;
    if (net->profiling && net->ev_valid) {
      float t = 0.f;
      hipError_t e = hipEventElapsedTime(&t, net->ev[i], net->ev[i + 1]);
      if (e == hipSuccess) *ms = t;
    }
  }
  return 0;
}

static int net_forward_impl(rtpose_net* net, const float* x_nchw, void* stream);

int rtpose_net_forward(rtpose_net* net, const float* x_nchw, void* stream) {
  if (!x_nchw) return fail(RTPOSE_E_INVAL, "net_forward: x is NULL");
  return net_forward_impl(net, x_nchw, stream);
}

int rtpose_net_forward_prepared(rtpose_net* net, void* stream) { return net_forward_impl(net, nullptr, stream); }

int rtpose_net_input_view(const rtpose_net* net, float** base, rtpose_layout* layout) {
  if (!net || !net->bound) return fail(RTPOSE_E_STATE, "net_input_view: net not bound");
  // bf16 plans expose an fp32 staging buffer; forward_prepared converts it
  const Buf& b = net->bufs[net->bf16 ? net->x0f_buf : net->x0_buf];
  if (base) *base = net->ws + b.off_floats;
  if (layout) *layout = b.lay;
  return 0;
}

static int net_run_ops(rtpose_net* net, size_t first, size_t last, const float* x_nchw, void* stream, bool prof);

// the plan's kernels go to the CURRENT device with pointers into the arenas bound on net->device
static int net_on_its_device(const rtpose_net* net, const char* who) {
  const int cur = current_device();
  if (cur == net->device) return 0;
  return fail(RTPOSE_E_STATE, "%s: the plan's arenas live on HIP device %d but the current device is %d", who,
              net->device, cur);
}

static int net_forward_impl(rtpose_net* net, const float* x_nchw, void* stream) {
  if (!net || !net->bound) return fail(RTPOSE_E_STATE, "net_forward: net not bound");
  hipStream_t s = as_stream(stream);
  int rcd = net_on_its_device(net, "net_forward");
  if (!rcd && x_nchw) rcd = net->in_checked.check(x_nchw, net->device, "net_forward", "the input tensor");
  if (rcd) return rcd;
  sync_arena_generation(net);
  bool stale = false;
  for (const Buf& b : net->bufs) stale |= b.stale;
  if (!net->forms_final || stale) {
    // deciding AUTO forms reads the amplification estimates back (92 small D2H copies + a stream synchronise) and a
    // buffer that changed storage is cleared: neither may happen inside a stream capture, where the first is illegal
    // and the second would be baked into the graph
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
      return fail(RTPOSE_E_STATE, "net_forward: the stream is capturing but the plan's forms are not final (weights were "
                                  "loaded since the last forward): call rtpose_net_finalize_weights and run one forward "
                                  "outside the capture first");
  }
  if (!net->forms_final) {  // AUTO forms, and the host did not call rtpose_net_finalize_weights since the last load
    const int rcf = rtpose_net_finalize_weights(net, stream);
    if (rcf) return rcf;
  }
  for (Buf& b : net->bufs) {  // buffers that changed between pixel-major and channel-plane storage (forms re-derived)
    if (!b.stale) continue;
    if (net->forwards > 0 || !net->zeroed_at_bind)
      RTPOSE_HIP_CHECK(hipMemsetAsync(net->ws + b.off_floats, 0, b.floats * sizeof(float), s));
    b.stale = false;
  }
  const bool prof = net->profiling && !net->ev.empty();
  const size_t nops = net->ops.size();
  if (net->graph_mode < 0) {
    const char* e = getenv("RTPOSE_GRAPH");
    // measured on MI355X / ROCm 7.2: replay is not faster than the 49 direct launches (batch-1 bf16
    // forward 1.16 ms vs 1.08 ms direct; batch 32 identical) - the launches are asynchronous and
    // the GPU-side dispatch cost is the same - so the graph path is opt-in (RTPOSE_GRAPH=1)
    net->graph_mode = (e && e[0] == '1') ? 1 : 0;
  }
  int rc = 0;
  if (net->graph_mode == 1 && !prof && net->forwards > 0 && nops > 1) {
    // op 0 (input conversion: its source pointer changes per call) runs on the caller's stream
    rc = net_run_ops(net, 0, 1, x_nchw, stream, false);
    if (rc) return rc;
    if (!net->gstream) {
      if (hipStreamCreateWithFlags(&net->gstream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&net->gev_in, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&net->gev_out, hipEventDisableTiming) != hipSuccess) {
        net->graph_mode = 0;
        (void)hipGetLastError();
        return net_run_ops(net, 1, nops, x_nchw, stream, false);
      }
    }
    const int slot = net->keep ? 1 : 0;
    if (!net->gexec[slot]) {
      hipGraph_t g = nullptr;
      hipError_t e = hipStreamBeginCapture(net->gstream, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        hipEvent_t guard = net->out_guard;  // not part of the captured list (see the replay below)
        net->out_guard = nullptr;
        rc = net_run_ops(net, 1, nops, nullptr, net->gstream, false);
        net->out_guard = guard;
        e = hipStreamEndCapture(net->gstream, &g);
        if (e == hipSuccess && !rc && g) e = hipGraphInstantiate(&net->gexec[slot], g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
      }
      if (e != hipSuccess || rc || !net->gexec[slot]) {  // no graphs on this runtime: direct launches from now on
        net->graph_mode = 0;
        net->gexec[slot] = nullptr;
        (void)hipGetLastError();
        return net_run_ops(net, 1, nops, x_nchw, stream, false);
      }
    }
    if (net->out_guard) RTPOSE_HIP_CHECK(hipStreamWaitEvent(s, net->out_guard, 0));  // (a replay cannot wait mid-list)
    RTPOSE_HIP_CHECK(hipEventRecord(net->gev_in, s));
    RTPOSE_HIP_CHECK(hipStreamWaitEvent(net->gstream, net->gev_in, 0));
    RTPOSE_HIP_CHECK(hipGraphLaunch(net->gexec[slot], net->gstream));
    RTPOSE_HIP_CHECK(hipEventRecord(net->gev_out, net->gstream));
    RTPOSE_HIP_CHECK(hipStreamWaitEvent(s, net->gev_out, 0));
    ++net->forwards;
    return 0;
  }
  rc = net_run_ops(net, 0, nops, x_nchw, stream, prof);
  if (rc) return rc;
  ++net->forwards;
  if (prof) {
    RTPOSE_HIP_CHECK(hipEventRecord(net->ev[nops], s));
    net->ev_valid = true;
  }
  return 0;
}

static int net_run_ops(rtpose_net* net, size_t first, size_t last, const float* x_nchw, void* stream, bool prof) {
  hipStream_t s = as_stream(stream);
  const int N = net->N;
  for (size_t i = first; i < last; ++i) {
    const Op& o = net->ops[i];
    if (prof) RTPOSE_HIP_CHECK(hipEventRecord(net->ev[i], s));
    if (net->out_guard && (int)i == net->guard_op) {
      RTPOSE_HIP_CHECK(hipStreamWaitEvent(s, net->out_guard, 0));
#ifdef RTPOSE_DEV_BUILD
      // experiment: behind the reader's last read, in front of the launch that rewrites them, the bf16
      // plan's maps become NaNs - a reader that returns a NaN afterwards was served a stale copy of the same address
      static const char* poison = getenv("RTPOSE_EXP_POISON");
      if (poison && poison[0] == '1' && net->bf16) {
        const Buf& b = net->bufs[net->save_buf[5]];
        RTPOSE_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(net->ws + b.off_floats), 0x7fc00000, b.floats, s));
      }
#endif
    }
    int rc = 0;
    switch (o.kind) {
      case OP_INPUT: {
        const Buf& b = net->bufs[o.out_buf[0]];
        if (net->bf16 && !net->split) {
          // conv1_1 of the bf16 plan reads fp32 itself (conv_first.hip MODE 2): the image where the caller left it when it
          // runs in the same call, else the plan's fp32 NHWC8 staging buffer (forward_prepared: filled by the caller; graph
          // replay: filled here) - the NCHW -> bf16 NHWC16 conversion launch is gone
          if (!x_nchw) break;
          bool first_reads_image = false;
          for (size_t j = first + 1; j < last && !first_reads_image; ++j)
            first_reads_image = net->ops[j].kind == OP_CONV && net->convs[net->ops[j].conv_idx[0]].first;
          if (first_reads_image) break;
          const Buf& bs = net->bufs[net->x0f_buf];
          rc = rtpose_nchw_to_layout(x_nchw, net->ws + bs.off_floats, &bs.lay, 3, 8, N, o.H, o.W, stream);
          break;
        }
        if (net->bf16) {
          rtpose_layout lb = b.lay;
          if (net->split) lb.cstride *= 2;  // elements
          const Buf& bs = net->bufs[net->x0f_buf];
          if (x_nchw && net->split)
            rc = rtpose_nchw_to_layout_split(x_nchw, net->ws + b.off_floats, &lb, 3, 16, N, o.H, o.W, stream);
          else if (x_nchw)
            rc = rtpose_nchw_to_layout_bf16(x_nchw, net->ws + b.off_floats, &lb, 3, 16, N, o.H, o.W, stream);
          else if (net->split)
            rc = rtpose_layout_f32_to_split(net->ws + bs.off_floats, &bs.lay, net->ws + b.off_floats, &lb, 3, 16,
                                            N, o.H, o.W, stream);
          else
            rc = rtpose_layout_f32_to_bf16(net->ws + bs.off_floats, &bs.lay, net->ws + b.off_floats, &lb, 3, 16,
                                           N, o.H, o.W, stream);
          break;
        }
        if (!x_nchw) break;  // forward_prepared: the input buffer was written by the caller
        // fp32 plans: conv1_1 reads the NCHW image itself (conv_first.hip) when it runs in the same call; the
        // conversion remains for graph replay, whose captured launch list reads the plan's own input buffer
        {
          bool first_reads_image = false;  // is conv1_1 (looked up by its flag, not by position) part of this call?
          for (size_t j = first + 1; j < last && !first_reads_image; ++j)
            first_reads_image = net->ops[j].kind == OP_CONV && net->convs[net->ops[j].conv_idx[0]].first;
          if (first_reads_image) break;
        }
        rc = rtpose_nchw_to_layout(x_nchw, net->ws + b.off_floats, &b.lay, 3, 8, N, o.H, o.W, stream);
        break;
      }
      case OP_CONV: {
        rtpose_conv_desc d[2];
        memset(d, 0, sizeof(d));
        for (int g = 0; g < o.ngroups; ++g) {
          const ConvW& c = net->convs[o.conv_idx[g]];
          const Buf& bi = net->bufs[o.in_buf[g]];
          const Buf& bo = net->bufs[o.out_buf[g]];
          d[g].in_plane_pixels = bi.plane_px;
          d[g].out_plane_pixels = bo.plane_px;
          d[g].in = net->ws + bi.off_floats;
          d[g].out = net->ws + bo.off_floats;
          d[g].w_packed = net->wt + (c.form == 3    ? c.w_off_w3
                                     : c.form == 43 ? c.w_off_w43
                                     : c.form == 4  ? c.w_off_w4
                                     : c.form == 6  ? c.w_off_w6
                                                    : c.w_off);
          d[g].wino_m = c.form == 4 || c.form == 6 ? c.form : c.form == 43 ? 4 : 0;
          d[g].bias_packed = net->wt + c.b_off;
          d[g].lin = slice(bi, o.in_choff[g]);
          d[g].lout = slice(bo, o.out_choff[g]);
          if (net->split) {  // split buffers: layouts count elements, 2 per channel
            d[g].lin.cstride *= 2;
            d[g].lin.choff *= 2;
            if (!o.out_f32) {
              d[g].lout.cstride *= 2;
              d[g].lout.choff *= 2;
            }
          }
          d[g].cin = c.cin_packed;
          d[g].cout = c.cout;
          d[g].k = c.k;
          d[g].relu = o.relu;
          d[g].pool = o.pool;
          d[g].out_cmap = nullptr;
        }
        if (net->convs[o.conv_idx[0]].first) {
          const ConvW& c = net->convs[o.conv_idx[0]];
          const bool direct_src = x_nchw && first == 0;  // the image itself; else the plan's NHWC8 input buffer
          if (net->bf16) {  // (fp32 source: the staging buffer, not this conv's bf16 input buffer)
            const Buf& bs = net->bufs[net->x0f_buf];
            rc = conv_first_launch(direct_src ? x_nchw : nullptr, net->ws + bs.off_floats, &bs.lay, net->wt + c.w_off_first,
                                   d[0].out, &d[0].lout, 0, o.relu, N, o.H, o.W, s, 1);
            break;
          }
          rc = conv_first_launch(direct_src ? x_nchw : nullptr, d[0].in, &d[0].lin, net->wt + c.w_off_first, d[0].out,
                                 &d[0].lout, d[0].out_plane_pixels, o.relu, N, o.H, o.W, s, 0);
          break;
        }
        const int form = net->convs[o.conv_idx[0]].form;  // grouped convs run one form (pick_forms)
        rc = net->bf16   ? conv2d_bf16_launch(d, o.ngroups, N, o.H, o.W, o.out_f32, net->split, s)
             : form == 3  ? conv2d_wino_launch(d, o.ngroups, N, o.H, o.W, s)
             : form == 43 ? conv2d_wino4_launch(d, o.ngroups, N, o.H, o.W, s)
             : form      ? conv2d_wino7_launch(d, o.ngroups, N, o.H, o.W, form,
                                               net->persist7 ? net->ws + net->scratch_off : nullptr,  // (no scratch:
                                               net->persist7 ? net->scratch_bytes : 0, s)             //  one block per tile)
                         : conv2d_launch(d, o.ngroups, N, o.H, o.W, s);
        break;
      }
      case OP_TAIL: {
        rtpose_conv_desc d1[2], d2[2];
        for (int g = 0; g < o.ngroups; ++g) {
          const ConvW &c1 = net->convs[o.conv_idx[g]], &c2 = net->convs[o.conv2_idx[g]];
          const Buf& bi = net->bufs[o.in_buf[g]];
          const Buf& bo = net->bufs[o.out_buf[g]];
          memset(&d1[g], 0, sizeof(d1[g]));
          memset(&d2[g], 0, sizeof(d2[g]));
          d1[g].in = net->ws + bi.off_floats;
          d1[g].lin = slice(bi, o.in_choff[g]);
          d1[g].w_packed = net->wt + c1.w_off;
          d1[g].bias_packed = net->wt + c1.b_off;
          d1[g].cin = c1.cin_packed;
          d1[g].cout = c1.cout;
          d1[g].k = 1;
          d1[g].relu = 1;
          d2[g].w_packed = net->wt + c2.w_off;
          d2[g].bias_packed = net->wt + c2.b_off;
          d2[g].cin = c2.cin_packed;
          d2[g].cout = c2.cout;
          d2[g].k = 1;
          d2[g].out = net->ws + bo.off_floats;
          d2[g].lout = slice(bo, o.out_choff[g]);
        }
        rc = net->bf16 ? conv_tail_bf16_launch(d1, d2, o.ngroups, N, o.H, o.W, o.out_f32, s)
                       : conv_tail_launch(d1, d2, o.ngroups, N, o.H, o.W, s);
        break;
      }
      case OP_POOL: {
        const Buf& bi = net->bufs[o.in_buf[0]];
        const Buf& bo = net->bufs[o.out_buf[0]];
        rc = rtpose_maxpool2x2(net->ws + bi.off_floats, &bi.lay, net->ws + bo.off_floats, &bo.lay, o.C,
                               N, o.H, o.W, stream);
        break;
      }
      case OP_COPY: {
        const bool is_save = o.name.rfind("save", 0) == 0;
        if (is_save && !net->keep) break;
        const Buf& bi = net->bufs[o.in_buf[0]];
        const Buf& bo = net->bufs[o.out_buf[0]];
        rtpose_layout li = slice(bi, o.in_choff[0]), lo = slice(bo, o.out_choff[0]);
        if (net->split && is_save) {
          li.cstride *= 2;
          li.choff *= 2;
          rc = rtpose_layout_split_to_f32(net->ws + bi.off_floats, &li, net->ws + bo.off_floats, &lo, o.C, N,
                                          o.H, o.W, stream);
          break;
        }
        if (net->bf16 && is_save) {  // bf16 concat slice -> fp32 record of the stage outputs
          rc = rtpose_layout_bf16_to_f32(net->ws + bi.off_floats, &li, net->ws + bo.off_floats, &lo, o.C, N,
                                         o.H, o.W, stream);
          break;
        }
        if (net->bf16 && !net->split) {  // bf16 -> bf16: move channel pairs as 4-byte words
          li.cstride /= 2; li.choff /= 2; lo.cstride /= 2; lo.choff /= 2;
          rc = rtpose_layout_copy(net->ws + bi.off_floats, &li, net->ws + bo.off_floats, &lo, o.C / 2, N, o.H,
                                  o.W, stream);
          break;
        }
        rc = rtpose_layout_copy(net->ws + bi.off_floats, &li, net->ws + bo.off_floats, &lo, o.C, N, o.H,
                                o.W, stream);
        break;
      }
    }
    if (rc) return rc;
  }
  return 0;
}

int rtpose_net_read_output(rtpose_net* net, int which, float* dst_nchw, void* stream) {
  if (!net || !net->bound) return fail(RTPOSE_E_STATE, "net_read_output: net not bound");
  if (which < 0 || which > 11 || !dst_nchw) return fail(RTPOSE_E_INVAL, "net_read_output: bad argument");
  int rcd = net_on_its_device(net, "net_read_output");
  if (!rcd) rcd = check_device_ptr(dst_nchw, net->device, "net_read_output", "the destination tensor");
  if (rcd) return rcd;
  const int stage = which / 2 + 1, br = which % 2;
  const int C = br == 0 ? 38 : 19;
  int buf, choff;
  if (net->bf16 && (net->keep || stage == 6)) {
    buf = net->save_buf[stage - 1];
    choff = br == 0 ? 0 : 38;
  } else if (net->bf16) {
    return fail(RTPOSE_E_STATE, "net_read_output: bf16 plan keeps only stage 6 (set keep_intermediates)");
  } else if (net->keep) {
    buf = net->save_buf[stage - 1];
    choff = br == 0 ? 0 : 38;
  } else {
    if (stage < 5) return fail(RTPOSE_E_STATE, "net_read_output: stage %d not kept (set keep_intermediates)", stage);
    buf = (stage % 2 == 0) ? net->cat_buf[0] : net->cat_buf[1];
    choff = br == 0 ? kCatPaf : kCatHeat;
  }
  const Buf& b = net->bufs[buf];
  const rtpose_layout l = slice(b, choff);
  return rtpose_layout_to_nchw(net->ws + b.off_floats, &l, dst_nchw, C, net->N, net->H3, net->W3, stream);
}

int rtpose_net_output_view(const rtpose_net* net, int which, const float** base, rtpose_layout* layout,
                           int* C, int* H, int* W) {
  if (!net || !net->bound || which < 0 || which > 1) return fail(RTPOSE_E_INVAL, "output_view: bad argument");
  // stage 6 writes CATa (fp32 plans) / the fp32 stage-6 record (bf16 plans)
  const Buf& b = net->bufs[net->bf16 ? net->save_buf[5] : net->cat_buf[0]];
  if (base) *base = net->ws + b.off_floats;
  if (layout) *layout = net->bf16 ? slice(b, which == 0 ? 0 : 38) : slice(b, which == 0 ? kCatPaf : kCatHeat);
  if (C) *C = which == 0 ? 38 : 19;
  if (H) *H = net->H3;
  if (W) *W = net->W3;
  return 0;
}

}  // extern "C"
