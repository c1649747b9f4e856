"""Developer probe: phases of csrc/conv_c64_bf16.hip per block and tile (wall_clock64 stamps: K loop start, K loop done, barrier
passed, stores issued, next halo parked).  Needs the -DRTPOSE_EXP_C64_TIMELINE build:
    ONLY=conv_c64_bf16 OUT=tools/exp/lib_c64tl.so tools/build_dev.sh -DRTPOSE_EXP_C64_TIMELINE
    RTPOSE_LIB_PATH=tools/exp/lib_c64tl.so python tools/exp/c64_timeline.py [pool 0|1] [cout]"""
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
pool = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n, h, w = 32, 368 if pool else 184, 368 if pool else 184
stream = capi.current_stream()
lin = Layout.padded(64, h, w, 1)
xin = (torch.randn(lib.rtpose_layout_pixels(C.byref(lin), n, h, w) * 64, device=dev) * 0.5).to(torch.bfloat16)
ho, wo = (h // 2, w // 2) if pool else (h, w)
lout = Layout.padded(cout, ho, wo, 1)
obuf = torch.zeros(lib.rtpose_layout_pixels(C.byref(lout), n, ho, wo) * cout, device=dev, dtype=torch.bfloat16)
wt = (torch.randn(cout, 64, 3, 3) * 0.05).to(dev)
bs = torch.zeros(cout, device=dev)
wp = torch.zeros(lib.rtpose_packed_weight_bytes_bf16(cout, 64, 3) // 2, device=dev, dtype=torch.bfloat16)
bp = torch.zeros(lib.rtpose_packed_bias_floats(cout), device=dev)
capi.check(lib.rtpose_pack_conv_weights_bf16(capi.ptr(wt), capi.ptr(bs), cout, 64, 3, None, 64, capi.ptr(wp), capi.ptr(bp), stream))
d = (capi.ConvDesc * 1)()
d[0].inp, d[0].w_packed, d[0].bias_packed, d[0].out = xin.data_ptr(), wp.data_ptr(), bp.data_ptr(), obuf.data_ptr()
d[0].lin, d[0].lout = lin, lout
d[0].cin, d[0].cout, d[0].k, d[0].relu, d[0].pool = 64, cout, 3, 1, pool
for _ in range(3):
    capi.check(lib.rtpose_conv2d_bf16(d, 1, n, h, w, 0, stream))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    capi.check(lib.rtpose_conv2d_bf16(d, 1, n, h, w, 0, stream))
e1.record()
torch.cuda.synchronize()
print("launch %.1f us" % (e0.elapsed_time(e1) * 100))
buf = (C.c_ulonglong * (64 * 24 * 6))()
lib.rtpose_exp_c64_timeline.restype = C.c_int
rc = lib.rtpose_exp_c64_timeline(buf)
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 24, 6).astype(np.int64)
print("rc", rc, "; 10 ns ticks -> us; per tile: K loop, barrier wait, next tile's loads + epilogue, wait + park in LDS, barrier | tile period")
for b in (0, 1, 2, 3, 17, 40):
    print("block", b)
    for k in range(2, 12):
        s = t[b, k]
        nxt = t[b, k + 1, 0]
        if s[4] <= s[0] or nxt <= s[4]:
            continue
        print("  tile %2d: %5.2f %5.2f %5.2f %5.2f %5.2f | %6.2f" % (k, *[(s[i + 1] - s[i]) / 100.0 for i in range(4)], (nxt - s[4]) / 100.0, (nxt - s[0]) / 100.0))
ok = (t[:, 2:14, 4] > t[:, 2:14, 0]) & (t[:, 3:15, 0] > t[:, 2:14, 4])
ph = [np.mean(((t[:, 2:14, i + 1] - t[:, 2:14, i]) / 100.0)[ok]) for i in range(4)] + [np.mean(((t[:, 3:15, 0] - t[:, 2:14, 4]) / 100.0)[ok])]
print("mean over blocks 0..63, tiles 2..13: K loop %.2f, barrier %.2f, loads + epilogue %.2f, wait + park %.2f, barrier %.2f us" % tuple(ph))
