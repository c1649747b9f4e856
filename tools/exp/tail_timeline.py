"""Developer probe: phases of the one-pass fp32 1x1 pair (csrc/conv_tail.hip: tail_kernel<1>) per block, wall_clock64 stamps:
start, tables + GEMM 1 filters requested (barrier), input tile in LDS (barrier), GEMM 1 done, barrier, intermediate in LDS (barrier),
GEMM 2 done, stores issued.  Needs  ONLY=conv_tail OUT=tools/exp/lib_tailtl.so tools/build_dev.sh -DRTPOSE_EXP_TAIL_TIMELINE
    RTPOSE_LIB_PATH=tools/exp/lib_tailtl.so python tools/exp/tail_timeline.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "pytorch_realtime_multi-person_pose_estimation_amd"
pkg = importlib.import_module(PKG)
capi = pkg._capi
lib, Layout = capi.lib, capi.Layout
dev = torch.device("cuda", 0)
n, h, w, mid = 32, 46, 46, 128
stream = capi.current_stream()
lin = Layout.padded(128, h, w, 3)
lmid = Layout.padded(mid, h, w, 0)
lcat = Layout.padded(192, h, w, 3)
npx = lib.rtpose_layout_pixels(C.byref(lin), n, h, w)
keep = []
d1, d2 = (capi.ConvDesc * 2)(), (capi.ConvDesc * 2)()
cat = torch.zeros(lib.rtpose_layout_pixels(C.byref(lcat), n, h, w) * 192, device=dev)
for gi, cout in enumerate((38, 19)):
    xin = torch.randn(npx * 128, device=dev)
    mids = torch.zeros(lib.rtpose_layout_pixels(C.byref(lmid), n, h, w) * mid, device=dev)
    packs = []
    for (co, ci) in ((mid, 128), (cout, mid)):
        wt = (torch.randn(co, ci, 1, 1) * 0.05).to(dev)
        bs = torch.zeros(co, device=dev)
        wp = torch.zeros(lib.rtpose_packed_weight_floats(co, ci, 1), device=dev)
        bp = torch.zeros(lib.rtpose_packed_bias_floats(co), device=dev)
        capi.check(lib.rtpose_pack_conv_weights(capi.ptr(wt), capi.ptr(bs), co, ci, 1, None, ci, capi.ptr(wp), capi.ptr(bp), stream))
        packs.append((wp, bp))
        keep += [wt, bs, wp, bp]
    keep += [xin, mids]
    d1[gi].inp, d1[gi].w_packed, d1[gi].bias_packed, d1[gi].out = xin.data_ptr(), packs[0][0].data_ptr(), packs[0][1].data_ptr(), mids.data_ptr()
    d1[gi].lin, d1[gi].lout = lin, lmid
    d1[gi].cin, d1[gi].cout, d1[gi].k, d1[gi].relu, d1[gi].pool = 128, mid, 1, 1, 0
    d2[gi].inp, d2[gi].w_packed, d2[gi].bias_packed, d2[gi].out = mids.data_ptr(), packs[1][0].data_ptr(), packs[1][1].data_ptr(), cat.data_ptr()
    d2[gi].lin = lmid
    d2[gi].lout = Layout.padded(192, h, w, 3, choff=128 if gi == 0 else 166)
    d2[gi].cin, d2[gi].cout, d2[gi].k, d2[gi].relu, d2[gi].pool = mid, cout, 1, 0, 0
for _ in range(3):
    capi.check(lib.rtpose_conv1x1_pair(d1, d2, 2, n, h, w, stream))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    capi.check(lib.rtpose_conv1x1_pair(d1, d2, 2, n, h, w, stream))
e1.record()
torch.cuda.synchronize()
print("launch %.1f us" % (e0.elapsed_time(e1) * 100))
nb = 2 * ((n * h * w + 63) // 64)
buf = (C.c_ulonglong * (4096 * 8))()
lib.rtpose_exp_tail_timeline.restype = C.c_int
rc = lib.rtpose_exp_tail_timeline(buf)
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)[:nb]
t0 = t[:, 0].min()
print("rc", rc, "blocks", nb, "; span %.1f us (first start to last end)" % ((t[:, 7].max() - t0) / 100.0))
names = ["tables + filters", "input tile", "GEMM 1", "barrier", "T write + barrier", "GEMM 2", "stores"]
d = (t[:, 1:] - t[:, :-1]) / 100.0
print("mean phase us: " + ", ".join("%s %.2f" % (nm, d[:, i].mean()) for i, nm in enumerate(names)) + "; block lifetime %.2f" % ((t[:, 7] - t[:, 0]) / 100.0).mean())
starts = np.sort((t[:, 0] - t0) / 100.0)
ends = np.sort((t[:, 7] - t0) / 100.0)
for us in (2, 5, 10, 20, 30, 40, 50, 60, 70):
    print("  t = %2d us: %4d blocks started, %4d finished, %4d resident" % (us, (starts <= us).sum(), (ends <= us).sum(), (starts <= us).sum() - (ends <= us).sum()))
