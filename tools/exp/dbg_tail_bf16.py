"""Debug: where does rtpose_conv1x1_pair_bf16 differ from the emulation / the two generic launches?"""
import ctypes as C
import importlib
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd")
capi = pkg._capi
lib, Layout = capi.lib, capi.Layout
cuda = torch.device("cuda", 0)


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def run(n, h, w, mid, cout2, cs_in, ch_in, cs_out, ch_out, pad_out):
    g = torch.Generator().manual_seed(1)
    stream = capi.current_stream()
    x = rb(torch.randn(n, 128, h, w, generator=g))
    lin = Layout.padded(cs_in, h, w, 0, choff=ch_in)
    xin = torch.zeros(lib.rtpose_layout_pixels(C.byref(lin), n, h, w) * cs_in, device=cuda, dtype=torch.bfloat16)
    capi.check(lib.rtpose_nchw_to_layout_bf16(capi.ptr(x.to(cuda)), capi.ptr(xin), C.byref(lin), 128, 128, n, h, w, stream))
    lo = Layout.padded(cs_out, h, w, pad_out, choff=ch_out)
    obuf = torch.zeros(lib.rtpose_layout_pixels(C.byref(lo), n, h, w) * cs_out, device=cuda, dtype=torch.bfloat16)
    obuf2 = torch.zeros_like(obuf)
    tl = Layout.padded(mid, h, w, 0)
    tb = torch.zeros(lib.rtpose_layout_pixels(C.byref(tl), n, h, w) * mid, device=cuda, dtype=torch.bfloat16)
    w1 = torch.randn(mid, 128, 1, 1, generator=g) * (2.0 / 128) ** 0.5
    b1 = torch.randn(mid, generator=g) * 0.1
    w2 = torch.randn(cout2, mid, 1, 1, generator=g) * (1.0 / mid) ** 0.5
    b2 = torch.randn(cout2, generator=g) * 0.1
    t = rb(F.relu(F.conv2d(x.double(), rb(w1).double(), b1.double()).float()))
    ref = rb(F.conv2d(t.double(), rb(w2).double(), b2.double()).float())
    packs = []
    for (wt, bs, co, ci) in ((w1, b1, mid, 128), (w2, b2, cout2, mid)):
        wp = torch.zeros(lib.rtpose_packed_weight_bytes_bf16(co, ci, 1) // 2, device=cuda, dtype=torch.bfloat16)
        bp = torch.zeros(lib.rtpose_packed_bias_floats(co), device=cuda)
        wd, bd = wt.to(cuda), bs.to(cuda)
        capi.check(lib.rtpose_pack_conv_weights_bf16(capi.ptr(wd), capi.ptr(bd), co, ci, 1, None, ci, capi.ptr(wp), capi.ptr(bp), stream))
        torch.cuda.synchronize()
        packs.append((wp, bp))
    d1, d2 = (capi.ConvDesc * 1)(), (capi.ConvDesc * 1)()
    a, b = d1[0], d2[0]
    a.inp, a.w_packed, a.bias_packed, a.out = xin.data_ptr(), packs[0][0].data_ptr(), packs[0][1].data_ptr(), tb.data_ptr()
    a.lin, a.lout = lin, tl
    a.cin, a.cout, a.k, a.relu, a.pool = 128, mid, 1, 1, 0
    b.inp, b.w_packed, b.bias_packed, b.out = tb.data_ptr(), packs[1][0].data_ptr(), packs[1][1].data_ptr(), obuf.data_ptr()
    b.lin, b.lout = tl, lo
    b.cin, b.cout, b.k, b.relu, b.pool = mid, cout2, 1, 0, 0
    capi.check(lib.rtpose_conv1x1_pair_bf16(d1, d2, 1, n, h, w, 0, stream), "pair")
    b.out = obuf2.data_ptr()
    capi.check(lib.rtpose_conv2d_bf16(d1, 1, n, h, w, 0, stream))
    capi.check(lib.rtpose_conv2d_bf16(d2, 1, n, h, w, 0, stream))

    def read(buf):
        dense = torch.empty(n, h, w, cout2, device=cuda)
        capi.check(lib.rtpose_layout_bf16_to_f32(capi.ptr(buf), C.byref(lo), capi.ptr(dense), C.byref(Layout.dense(cout2, h, w)), cout2, n, h, w, stream))
        torch.cuda.synchronize()
        return dense.permute(0, 3, 1, 2).contiguous().cpu()
    fused, plain = read(obuf), read(obuf2)
    tol = ref.abs() * 2.0 ** -6 + 2e-3 * max(1.0, ref.abs().max().item())
    for name, o in (("fused", fused), ("generic", plain)):
        bad = (o - ref).abs() > tol
        print("%s n=%d %dx%d mid=%d cout2=%d in(cs %d, off %d) out(cs %d, off %d, pad %d): %d of %d outputs off; by channel %s; by pixel %% 64 (first 16) %s" % (
            name, n, h, w, mid, cout2, cs_in, ch_in, cs_out, ch_out, pad_out, int(bad.sum()), bad.numel(),
            bad.sum(dim=(0, 2, 3)).tolist(), (bad.permute(0, 2, 3, 1).reshape(-1, cout2).any(dim=1).reshape(-1)[:4224].reshape(-1, 64).sum(0)[:16].tolist() if n * h * w >= 4224 else "-")), flush=True)


run(2, 46, 46, 128, 38, 128, 0, 192, 0, 0)
run(2, 46, 46, 128, 38, 136, 8, 192, 0, 0)
run(2, 46, 46, 128, 38, 128, 0, 64, 3, 3)
run(2, 46, 46, 128, 38, 136, 8, 64, 3, 3)
run(2, 46, 46, 128, 19, 128, 0, 192, 38, 0)
run(2, 46, 46, 512, 38, 128, 0, 192, 0, 0)
