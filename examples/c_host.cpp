// A host WITHOUT Python or torch driving the whole hot path through the C ABI
// (include/rtpose_mi355x.h): plan -> arenas (hipMalloc) -> weights -> forward -> decode -> records.
// Build:  hipcc --offload-arch=gfx950 -O2 -Iinclude examples/c_host.cpp \
//             -Lpytorch_realtime_multi-person_pose_estimation_amd/lib -lrtpose_mi355x -o examples/c_host
// Run:    LD_LIBRARY_PATH=pytorch_realtime_multi-person_pose_estimation_amd/lib examples/c_host [batch] [dtype] [auto|direct|f47]
// Weights are random (no checkpoint offline): the program demonstrates the call sequence and prints
// the throughput; a real host would hand rtpose_net_load_conv the 92 OIHW tensors of pose_model.pth.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "rtpose_mi355x.h"

#define CK(expr)                                                                   \
  do {                                                                             \
    int rc__ = (expr);                                                             \
    if (rc__) {                                                                    \
      std::fprintf(stderr, "%s -> %d: %s\n", #expr, rc__, rtpose_last_error());    \
      return 1;                                                                    \
    }                                                                              \
  } while (0)
#define HK(expr)                                                                   \
  do {                                                                             \
    hipError_t e__ = (expr);                                                       \
    if (e__ != hipSuccess) {                                                       \
      std::fprintf(stderr, "%s -> %s\n", #expr, hipGetErrorString(e__));           \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

int main(int argc, char** argv) {
  const int N = argc > 1 ? std::atoi(argv[1]) : 8, H = 368, W = 368;
  const int dtype = argc > 2 ? std::atoi(argv[2]) : RTPOSE_DTYPE_F32;  // 0 fp32, 1 bf16, 2 bf16x3
  // arithmetic of the fp32 convs, chosen per plan through the ABI: default (F(2x2,3x3) + F(6,7)), "auto" (per layer
  // by the amplification estimate of the loaded filters), "direct" (no Winograd form), "f47"
  const char* mode = argc > 3 ? argv[3] : "default";
  std::printf("%s, batch %d, dtype %d, winograd %s\n", rtpose_version(), N, dtype, mode);

  rtpose_net_options opt;
  opt.struct_bytes = sizeof(opt);
  opt.dtype = dtype;
  opt.winograd3 = !std::strcmp(mode, "direct") ? 0 : !std::strcmp(mode, "auto") ? RTPOSE_WINO3_AUTO : RTPOSE_WINO_DEFAULT;
  opt.winograd7 = !std::strcmp(mode, "auto") ? RTPOSE_WINO7_AUTO : !std::strcmp(mode, "direct") ? 0
                  : !std::strcmp(mode, "f47") ? 4 : RTPOSE_WINO_DEFAULT;
  opt.amp_limit = 0.f;  // library default
  rtpose_net* net = nullptr;
  CK(rtpose_net_create_opts(N, H, W, &opt, &net));
  void *ws = nullptr, *wt = nullptr;
  const size_t ws_bytes = rtpose_net_workspace_bytes(net), wt_bytes = rtpose_net_weight_bytes(net);
  HK(hipMalloc(&ws, ws_bytes));
  HK(hipMalloc(&wt, wt_bytes));
  hipStream_t s;
  HK(hipStreamCreate(&s));
  CK(rtpose_net_bind(net, ws, ws_bytes, wt, wt_bytes, /*zero_workspace=*/1, s));

  // weights: the reference's own initialisation (N(0, 0.01), zero bias; rtpose_vgg.py:200-222), uploaded as
  // OIHW fp32 and packed by the library - the maps stay ~1e-10, so the decoder finds no peaks and the
  // printed rate is that of the forward + an (empty) decode
  std::mt19937 rng(0);
  for (int i = 0; i < rtpose_net_num_convs(net); ++i) {
    char name[64];
    int co, ci, k;
    CK(rtpose_net_conv_info(net, i, name, sizeof(name), &co, &ci, &k));
    std::vector<float> w((size_t)co * ci * k * k), b(co);
    std::normal_distribution<float> dw(0.f, 0.01f);
    for (float& v : w) v = dw(rng);
    for (float& v : b) v = 0.f;
    float *dwp = nullptr, *dbp = nullptr;
    HK(hipMalloc(&dwp, w.size() * 4));
    HK(hipMalloc(&dbp, b.size() * 4));
    HK(hipMemcpyAsync(dwp, w.data(), w.size() * 4, hipMemcpyHostToDevice, s));
    HK(hipMemcpyAsync(dbp, b.data(), b.size() * 4, hipMemcpyHostToDevice, s));
    CK(rtpose_net_load_conv(net, i, dwp, dbp, s));
    HK(hipStreamSynchronize(s));
    HK(hipFree(dwp));
    HK(hipFree(dbp));
  }

  // fixes the per-layer forms of an "auto" plan (reads the amplification estimates back once)
  CK(rtpose_net_finalize_weights(net, s));
  {
    int nform[44] = {0};
    float amp_max[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < rtpose_net_num_convs(net); ++i) {
      int form = 0;
      float amp[4];
      CK(rtpose_net_conv_numerics(net, i, &form, amp, s));
      ++nform[form];
      for (int j = 0; j < 4; ++j) amp_max[j] = amp[j] > amp_max[j] ? amp[j] : amp_max[j];
    }
    std::printf("convs by form: direct %d, F(2x2,3x3) %d, F(4x4,3x3) %d, F(4,7) %d, F(6,7) %d; worst amplification "
                "estimates %.1f / %.1f / %.1f / %.1f\n", nform[0], nform[3], nform[43], nform[4], nform[6], amp_max[0],
                amp_max[3], amp_max[1], amp_max[2]);
  }

  // input batch: dense NCHW fp32 in [-0.5, 0.5) (rtpose_preprocess range)
  std::vector<float> x((size_t)N * 3 * H * W);
  std::uniform_real_distribution<float> dx(-0.5f, 0.5f);
  for (float& v : x) v = dx(rng);
  float* xd = nullptr;
  HK(hipMalloc(&xd, x.size() * 4));
  HK(hipMemcpy(xd, x.data(), x.size() * 4, hipMemcpyHostToDevice));

  // decoder scratch + result records
  rtpose_decode_cfg cfg = {18, 8, 0.1f, 256, 256};
  const size_t dws = rtpose_decode_workspace_bytes(&cfg, N), drs = rtpose_decode_result_bytes(&cfg, N);
  void *dw_ = nullptr, *dr = nullptr;
  HK(hipMalloc(&dw_, dws));
  HK(hipMalloc(&dr, drs));
  std::vector<int32_t> rec(drs / 4);

  const float *pbase, *hbase;
  rtpose_layout lpaf, lheat;
  int C, h, w;
  CK(rtpose_net_output_view(net, 0, &pbase, &lpaf, &C, &h, &w));
  CK(rtpose_net_output_view(net, 1, &hbase, &lheat, &C, &h, &w));

  const int iters = 5;
  auto t0 = std::chrono::steady_clock::now();
  for (int it = -1; it < iters; ++it) {  // one untimed warm-up
    if (it == 0) {
      HK(hipStreamSynchronize(s));
      t0 = std::chrono::steady_clock::now();
    }
    CK(rtpose_net_forward(net, xd, s));
    CK(rtpose_decode_batch(hbase, &lheat, pbase, &lpaf, N, h, w, &cfg, dw_, dws, dr, s));
    HK(hipMemcpyAsync(rec.data(), dr, drs, hipMemcpyDeviceToHost, s));
    HK(hipStreamSynchronize(s));
  }
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const size_t words = drs / 4 / N;
  long peaks = 0, humans = 0, flags = 0;
  for (int n = 0; n < N; ++n) {
    peaks += rec[n * words + 0];
    humans += rec[n * words + 1];
    flags |= rec[n * words + 2];
  }
  std::printf("%d x (forward + decode + D2H) of %d images: %.1f images/s; maps %dx%d; %ld peaks, %ld humans, "
              "overflow flags %ld\n", iters, N, iters * N / sec, h, w, peaks, humans, flags);
  int status = -1;
  CK(rtpose_net_device_status(net, &status, s));
  std::printf("device status %d, workspace %.2f GB (incl. the persistent kernels' hand-over scratch), weights %.2f GB\n",
              status, ws_bytes / 1e9, wt_bytes / 1e9);
  rtpose_net_destroy(net);
  (void)hipFree(ws);
  (void)hipFree(wt);
  (void)hipFree(xd);
  (void)hipFree(dw_);
  (void)hipFree(dr);
  return 0;
}
