"""Time one 3x3 conv launch through the C ABI in its forms: direct, F(2x2,3x3), F(4x4,3x3).
   python tools/bench_conv3.py [N H W cin cout pool groups] ...   (default: the rtpose_vgg 3x3 shapes at batch 32)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module

pkg = import_module("pytorch_realtime_multi-person_pose_estimation_amd")
capi = import_module("pytorch_realtime_multi-person_pose_estimation_amd._capi")
lib, Layout = capi.lib, capi.Layout
dev = torch.device("cuda:0")


def bench(n, h, w, cin, cout, pool, groups, iters=10):
    g = torch.Generator().manual_seed(1)
    stream = capi.current_stream()
    lin = Layout.padded(cin, h, w, 1)
    xin = torch.zeros(lib.rtpose_layout_pixels(C.byref(lin), n, h, w) * cin, device=dev)
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    capi.check(lib.rtpose_nchw_to_layout(capi.ptr(x), capi.ptr(xin), C.byref(lin), cin, cin, n, h, w, stream))
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    cs = cout * groups
    lout = Layout.padded(cs, ho, wo, 1)
    res = {}
    outs = {}
    for form in ("direct", "f2", "f4"):
        obuf = torch.zeros(lib.rtpose_layout_pixels(C.byref(lout), n, ho, wo) * cs, device=dev)
        descs = (capi.ConvDesc * groups)()
        keep = []
        gw = torch.Generator().manual_seed(2)
        for gi in range(groups):
            wt = (torch.randn(cout, cin, 3, 3, generator=gw) * (2.0 / (cin * 9)) ** 0.5).to(dev)
            b = (torch.randn(cout, generator=gw) * 0.1).to(dev)
            bp = torch.zeros(lib.rtpose_packed_bias_floats(cout), device=dev)
            if form == "direct":
                wp = torch.zeros(lib.rtpose_packed_weight_floats(cout, cin, 3), device=dev)
                capi.check(lib.rtpose_pack_conv_weights(capi.ptr(wt), capi.ptr(b), cout, cin, 3, None, cin, capi.ptr(wp), capi.ptr(bp), stream))
            else:
                m = 4 if form == "f4" else 2
                wp = torch.zeros(lib.rtpose_packed_weight_floats_winograd3(cout, cin, m), device=dev)
                capi.check(lib.rtpose_pack_conv_weights_winograd3(capi.ptr(wt), capi.ptr(b), cout, cin, m, None, cin, capi.ptr(wp), capi.ptr(bp), stream))
            keep += [wt, b, wp, bp]
            d = descs[gi]
            d.inp, d.w_packed, d.bias_packed, d.out = xin.data_ptr(), wp.data_ptr(), bp.data_ptr(), obuf.data_ptr()
            d.lin = lin
            d.lout = Layout.padded(cs, ho, wo, 1, choff=gi * cout)
            d.cin, d.cout, d.k, d.relu, d.pool = cin, cout, 3, 1, int(pool)
            d.wino_m = 4 if form == "f4" else 0

        def run():
            if form == "direct":
                capi.check(lib.rtpose_conv2d(descs, groups, n, h, w, stream))
            else:
                capi.check(lib.rtpose_conv2d_winograd(descs, groups, n, h, w, stream))
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[form] = e0.elapsed_time(e1) / iters
        outs[form] = obuf
    ref = outs["direct"]
    sc = ref.abs().max().item()
    fl = 2.0 * n * h * w * cin * cout * 9 * groups
    print("N=%d %dx%d %d->%d pool=%d g=%d: " % (n, h, w, cin, cout, pool, groups) +
          "  ".join("%s %.3f ms (%.0f TF/s)" % (k, v, fl / v / 1e9) for k, v in res.items()) +
          "  | err f2 %.2e f4 %.2e (max|y| %.2f)" % ((outs["f2"] - ref).abs().max().item(), (outs["f4"] - ref).abs().max().item(), sc))


if len(sys.argv) > 1:
    a = [int(v) for v in sys.argv[1:]]
    bench(*a)
else:
    for shp in ((32, 368, 368, 64, 64, 1, 1), (32, 184, 184, 64, 128, 0, 1), (32, 184, 184, 128, 128, 1, 1),
                (32, 92, 92, 128, 256, 0, 1), (32, 92, 92, 256, 256, 0, 1), (32, 46, 46, 256, 512, 0, 1),
                (32, 46, 46, 512, 512, 0, 1), (32, 46, 46, 512, 256, 0, 1), (32, 46, 46, 256, 128, 0, 1),
                (32, 46, 46, 128, 128, 0, 2)):
        bench(*shp)
