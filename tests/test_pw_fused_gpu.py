"""GPU parity of the fused pointwise chain kernel (csrc/pw_fused.hip, rtpose_pw_fused) against
plain torch-CPU fp32 ops: F.conv2d 1x1 (+ReLU), the depthwise 3x3 F.conv2d(groups=C, padding=1) in
front of it, and the cat + channel_shuffle store pattern (lib/network/rtpose_shufflenetV2.py:22-63).
Every wave arrangement (64 / 128 / 256+ columns), K that is and is not a multiple of the 32-channel
chunk, ragged pixel counts (last strip partly empty), strips that span image boundaries."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-4   # fp32 MFMA vs ATen fp32: different summation order only


def _to_layout(capi, x_nchw, lay, cpad, cuda):
    n, c, h, w = x_nchw.shape
    buf = torch.zeros(capi.lib.rtpose_layout_pixels(C.byref(lay), n, h, w) * lay.cstride, device=cuda)
    capi.check(capi.lib.rtpose_nchw_to_layout(capi.ptr(x_nchw.to(cuda).contiguous()), capi.ptr(buf), C.byref(lay), c,
                                              cpad, n, h, w, capi.current_stream()))
    return buf


def _from_layout(capi, buf, lay, c, n, h, w, cuda):
    out = torch.empty(n, c, h, w, device=cuda)
    capi.check(capi.lib.rtpose_layout_to_nchw(capi.ptr(buf), C.byref(lay), capi.ptr(out), c, n, h, w,
                                              capi.current_stream()))
    return out.cpu()


def _run(capi, cuda, n, h, w, cin, cout, coutp, dw, relu, pt_c=0, pad_in=1, seed=0):
    g = torch.Generator().manual_seed(seed)
    K = (cin + 7) // 8 * 8
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    lin = capi.Layout.padded(K, h, w, pad_in) if pad_in else capi.Layout.dense(K, h, w)
    xin = _to_layout(capi, x, lin, K, cuda)
    ref_in = x
    d = capi.PwDesc()
    keep = [xin]
    if dw:
        dw_w = torch.randn(cin, 1, 3, 3, generator=g) * 0.3
        dw_b = torch.randn(cin, generator=g) * 0.1
        ref_in = F.conv2d(x, dw_w, dw_b, padding=1, groups=cin)
        wp = torch.zeros(9, K)
        wp[:, :cin] = dw_w.reshape(cin, 9).t()
        bp = torch.zeros(K)
        bp[:cin] = dw_b
        wp_d, bp_d = wp.to(cuda), bp.to(cuda)
        keep += [wp_d, bp_d]
        d.dw_w, d.dw_b = wp_d.data_ptr(), bp_d.data_ptr()
    ref = F.conv2d(ref_in, wt[:, :, None, None], b)
    if relu:
        ref = F.relu(ref)
    wpk = torch.zeros(capi.lib.rtpose_packed_pw_floats(K, coutp) + 64 * coutp, device=cuda)
    bpk = torch.zeros(coutp, device=cuda)
    wt_d, b_d = wt.contiguous().to(cuda), b.to(cuda)
    capi.check(capi.lib.rtpose_pack_pw_weights(capi.ptr(wt_d), capi.ptr(b_d), cout, cin, None, K, coutp, 0,
                                               capi.ptr(wpk), capi.ptr(bpk), capi.current_stream()))
    # output: [pt | gemm] interleaved like channel_shuffle(2) when pt_c, else plain
    ctot = (2 * cout if pt_c else cout)
    cs = (ctot + 7) // 8 * 8
    lout = capi.Layout.padded(cs, h, w, 1)
    out = torch.zeros(capi.lib.rtpose_layout_pixels(C.byref(lout), n, h, w) * cs, device=cuda)
    keep += [wpk, bpk, wt_d, b_d, out]
    d.inp, d.w_packed, d.bias_packed, d.out = xin.data_ptr(), wpk.data_ptr(), bpk.data_ptr(), out.data_ptr()
    d.lin, d.lout = lin, lout
    d.cin, d.cout, d.coutp, d.relu = K, cout, coutp, 1 if relu else 0
    if pt_c:
        assert pt_c == cout
        xpt = torch.randn(n, pt_c, h, w, generator=g)
        lpt = capi.Layout.padded((pt_c + 7) // 8 * 8, h, w, 1)
        ptb = _to_layout(capi, xpt, lpt, lpt.cstride, cuda)
        odd = torch.arange(cout, dtype=torch.int32) * 2 + 1
        even = torch.arange(pt_c, dtype=torch.int32) * 2
        odd_d, even_d = odd.to(cuda), even.to(cuda)
        keep += [ptb, odd_d, even_d]
        d.out_cmap, d.pt_src, d.lpt, d.pt_cmap, d.pt_c = odd_d.data_ptr(), ptb.data_ptr(), lpt, even_d.data_ptr(), pt_c
        ref = torch.stack([xpt, ref], 2).reshape(n, 2 * cout, h, w)      # cat + channel_shuffle(2)
    capi.check(capi.lib.rtpose_pw_fused(C.byref(d), n, h, w, capi.current_stream()), "rtpose_pw_fused")
    got = _from_layout(capi, out, lout, ctot, n, h, w, cuda)
    # the layout gaps must stay untouched (zero): everything outside the real pixels
    total = out.abs().sum().item()
    inside = got.abs().sum().item()
    assert abs(total - inside) <= 1e-3 * max(1.0, inside), "kernel wrote outside the real pixels"
    scale = max(1.0, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    assert err <= TOL * scale, (err, scale)


@pytest.mark.parametrize("cin,cout,coutp", [(24, 58, 64), (64, 58, 64), (120, 116, 128), (232, 232, 256),
                                            (464, 1024, 1024), (1024, 100, 128), (32, 19, 64)])
def test_pointwise_only(capi, cuda, cin, cout, coutp):
    _run(capi, cuda, 3, 13, 17, cin, cout, coutp, dw=False, relu=True, pad_in=1, seed=cin)
    _run(capi, cuda, 2, 46, 46, cin, cout, coutp, dw=False, relu=False, pad_in=0, seed=cin + 1)


@pytest.mark.parametrize("cin,cout,coutp", [(58, 58, 64), (116, 116, 128), (232, 232, 256), (128, 116, 128),
                                            (240, 232, 256)])
def test_depthwise_then_pointwise(capi, cuda, cin, cout, coutp):
    _run(capi, cuda, 3, 13, 17, cin, cout, coutp, dw=True, relu=True, seed=cin)
    _run(capi, cuda, 2, 46, 46, cin, cout, coutp, dw=True, relu=True, seed=cin + 1)
    _run(capi, cuda, 1, 5, 60, cin, cout, coutp, dw=True, relu=False, seed=cin + 2)       # H < one 8 x 8 tile
    _run(capi, cuda, 2, 19, 133, cin, cout, coutp, dw=True, relu=True, seed=cin + 3)      # wide map, ragged tiles


@pytest.mark.parametrize("c", [58, 116, 232])
def test_unit_with_pass_through_and_shuffle(capi, cuda, c):
    _run(capi, cuda, 2, 23, 25, c, c, (c + 63) // 64 * 64, dw=True, relu=True, pt_c=c, seed=c)
    _run(capi, cuda, 4, 46, 46, c, c, (c + 63) // 64 * 64, dw=True, relu=True, pt_c=c, seed=c + 1)


@pytest.mark.parametrize("h,w_map", [(58, 46), (116, 46), (232, 46), (116, 70)])
def test_unit_in_the_four_run_layout(capi, cuda, h, w_map):
    """One ShuffleNetV2 unit (rtpose_shufflenetV2.py:31-39, :56-62) the way the fused fp32 plan runs it: the
    stage buffer keeps its 2h logical channels as four runs [even-low | even-high | odd-low | odd-high];
    launch 1 = conv.0 on x2 gathered as two runs (in_planes), launch 2 = conv.1 (depthwise, in LDS) -> conv.2 ->
    the odd runs + the next x1 = (even-low, odd-low) interleaved -> the even runs.  Checked against
    torch: chunk -> conv/bn-folded convs -> cat -> channel_shuffle(2)."""
    g = torch.Generator().manual_seed(h)
    n, hh_, q = 2, h // 2, (h // 2 + 3) // 4 * 4
    H = 9
    C4 = 4 * q
    x = torch.randn(n, 2 * h, H, w_map, generator=g)                      # logical channel order
    w0 = torch.randn(h, h, generator=g) / h ** 0.5
    b0 = torch.randn(h, generator=g) * 0.1
    wd = torch.randn(h, 1, 3, 3, generator=g) * 0.3
    bd = torch.randn(h, generator=g) * 0.1
    w2 = torch.randn(h, h, generator=g) / h ** 0.5
    b2 = torch.randn(h, generator=g) * 0.1
    x1, x2 = x[:, :h], x[:, h:]
    y = F.relu(F.conv2d(F.conv2d(F.relu(F.conv2d(x2, w0[:, :, None, None], b0)), wd, bd, padding=1, groups=h),
                        w2[:, :, None, None], b2))
    ref = torch.stack([x1, y], 2).reshape(n, 2 * h, H, w_map)            # cat + channel_shuffle(2)

    def phys(j):
        i = j >> 1
        return (2 * (j & 1) + (1 if i >= hh_ else 0)) * q + (i - hh_ if i >= hh_ else i)
    perm = torch.tensor([phys(j) for j in range(2 * h)])
    xp = torch.zeros(n, C4, H, w_map)
    xp[:, perm] = x
    lay = capi.Layout.padded(C4, H, w_map, 1)
    cur = _to_layout(capi, xp, lay, C4, cuda)
    nxt = torch.zeros_like(cur)
    K = 2 * q
    lt1 = capi.Layout.padded(K, H, w_map, 1)
    t1 = torch.zeros(capi.lib.rtpose_layout_pixels(C.byref(lt1), n, H, w_map) * K, device=cuda)
    coutp = (h + 63) // 64 * 64
    # conv.0: packed K position k reads x2 channel x2map[k]; plane j sits at channel pln[j]
    x2map = torch.full((K,), -1, dtype=torch.int32)
    for k in range(K):
        p = k if k < q else k - q
        if p < hh_:
            x2map[k] = 2 * p + (0 if k < q else 1)
    pln = torch.tensor([q + 4 * j if 4 * j < q else 3 * q + (4 * j - q) for j in range(K // 4)], dtype=torch.int32)
    keep = []

    def pack(wt, b, cmap):
        wpk = torch.zeros(capi.lib.rtpose_packed_pw_floats(K, coutp) + 64 * coutp, device=cuda)
        bpk = torch.zeros(coutp, device=cuda)
        wt_d, b_d = wt.contiguous().to(cuda), b.to(cuda)
        cm = cmap.to(cuda) if cmap is not None else None
        capi.check(capi.lib.rtpose_pack_pw_weights(capi.ptr(wt_d), capi.ptr(b_d), h, h, capi.ptr(cm) if cm is not None else None,
                                                   K, coutp, 0, capi.ptr(wpk), capi.ptr(bpk), capi.current_stream()))
        keep.extend([wpk, bpk, wt_d, b_d, cm])
        return wpk, bpk
    wp0, bp0 = pack(w0, b0, x2map)
    wp2, bp2 = pack(w2, b2, None)
    pln_d = pln.to(cuda)
    d = capi.PwDesc()
    d.inp, d.w_packed, d.bias_packed, d.out = cur.data_ptr(), wp0.data_ptr(), bp0.data_ptr(), t1.data_ptr()
    d.lin, d.lout, d.cin, d.cout, d.coutp, d.relu = lay, lt1, K, h, coutp, 1
    d.in_planes = pln_d.data_ptr()
    capi.check(capi.lib.rtpose_pw_fused(C.byref(d), n, H, w_map, capi.current_stream()), "conv.0")
    wdp = torch.zeros(9, K)
    wdp[:, :h] = wd.reshape(h, 9).t()
    bdp = torch.zeros(K)
    bdp[:h] = bd
    wdp_d, bdp_d = wdp.to(cuda), bdp.to(cuda)
    odd = torch.tensor([phys(2 * i + 1) for i in range(h)], dtype=torch.int32).to(cuda)
    d2 = capi.PwDesc()
    d2.inp, d2.w_packed, d2.bias_packed, d2.out = t1.data_ptr(), wp2.data_ptr(), bp2.data_ptr(), nxt.data_ptr()
    d2.dw_w, d2.dw_b = wdp_d.data_ptr(), bdp_d.data_ptr()
    d2.lin, d2.lout, d2.cin, d2.cout, d2.coutp, d2.relu = lt1, lay, K, h, coutp, 1
    d2.out_cmap = odd.data_ptr()
    d2.pt_src, d2.lpt = cur.data_ptr(), lay
    d2.pt_pairs, d2.pt_a, d2.pt_b, d2.pt_split, d2.pt_d0, d2.pt_d1 = hh_, 0, 2 * q, hh_, 0, q
    capi.check(capi.lib.rtpose_pw_fused(C.byref(d2), n, H, w_map, capi.current_stream()), "conv.1+conv.2+x1")
    got = _from_layout(capi, nxt, lay, C4, n, H, w_map, cuda)[:, perm]
    scale = max(1.0, ref.abs().max().item())
    assert (got[:, 0::2] - ref[:, 0::2]).abs().max().item() == 0.0          # the pass-through half is a copy
    assert (got - ref).abs().max().item() <= TOL * scale
    total, inside = nxt.abs().sum().item(), got.abs().sum().item()
    assert abs(total - inside) <= 1e-3 * max(1.0, inside), "kernel wrote outside the real channels / pixels"


@pytest.mark.parametrize("h,w_map", [(58, 46), (116, 46), (232, 46), (116, 70)])
def test_unit_in_the_four_run_layout_with_column_mapped_packing(capi, cuda, h, w_map):
    """One ShuffleNetV2 unit (rtpose_shufflenetV2.py:31-39, :56-62) the way the fp32 plan runs it since round 4:
    four-run stage buffer, launch 1 = conv.0 on x2 gathered as two runs with K padded to a multiple of 16 (the extra
    planes repeat a real plane under zero weights; stored width hp, zero columns past h), launch 2 = conv.1
    (depthwise, in LDS) -> conv.2 packed with a COLUMN MAP (rtpose_pack_pw_weights_cols) so that its columns are the odd
    runs in memory order and are stored as contiguous channels (no out_cmap), + the next x1 = (even-low, odd-low)
    interleaved -> the even runs.  Against torch: chunk -> convs -> cat -> channel_shuffle(2)."""
    g = torch.Generator().manual_seed(h + 1)
    n, hh_, q = 2, h // 2, (h // 2 + 3) // 4 * 4
    H = 9
    C4, hp = 4 * q, 2 * q
    Kp = (hp + 15) // 16 * 16
    x = torch.randn(n, 2 * h, H, w_map, generator=g)                      # logical channel order
    w0 = torch.randn(h, h, generator=g) / h ** 0.5
    b0 = torch.randn(h, generator=g) * 0.1
    wd = torch.randn(h, 1, 3, 3, generator=g) * 0.3
    bd = torch.randn(h, generator=g) * 0.1
    w2 = torch.randn(h, h, generator=g) / h ** 0.5
    b2 = torch.randn(h, generator=g) * 0.1
    x1, x2 = x[:, :h], x[:, h:]
    y = F.relu(F.conv2d(F.conv2d(F.relu(F.conv2d(x2, w0[:, :, None, None], b0)), wd, bd, padding=1, groups=h),
                        w2[:, :, None, None], b2))
    ref = torch.stack([x1, y], 2).reshape(n, 2 * h, H, w_map)            # cat + channel_shuffle(2)

    def phys(j):
        i = j >> 1
        return (2 * (j & 1) + (1 if i >= hh_ else 0)) * q + (i - hh_ if i >= hh_ else i)
    perm = torch.tensor([phys(j) for j in range(2 * h)])
    xp = torch.zeros(n, C4, H, w_map)
    xp[:, perm] = x
    lay = capi.Layout.padded(C4, H, w_map, 1)
    cur = _to_layout(capi, xp, lay, C4, cuda)
    nxt = torch.zeros_like(cur)
    lt1 = capi.Layout.padded(Kp, H, w_map, 1)
    t1 = torch.zeros(capi.lib.rtpose_layout_pixels(C.byref(lt1), n, H, w_map) * Kp, device=cuda)
    coutp = (hp + 63) // 64 * 64
    x2map = torch.full((Kp,), -1, dtype=torch.int32)
    for k in range(hp):
        p = k if k < q else k - q
        if p < hh_:
            x2map[k] = 2 * p + (0 if k < q else 1)
    pln = torch.tensor([q if 4 * j >= hp else (q + 4 * j if 4 * j < q else 3 * q + (4 * j - q)) for j in range(Kp // 4)],
                       dtype=torch.int32)
    cols = torch.full((coutp,), -1, dtype=torch.int32)          # packed column c' of conv.2 -> y channel
    for c in range(hp):
        p = c if c < q else c - q
        if p < hh_:
            cols[c] = p if c < q else hh_ + p
    keep = []

    def pack(wt, b, cin_map, col_map):
        wpk = torch.full((capi.lib.rtpose_packed_pw_floats(Kp, coutp) + 64 * coutp,), float("nan"), device=cuda)
        bpk = torch.full((coutp,), float("nan"), device=cuda)
        wt_d, b_d = wt.contiguous().to(cuda), b.to(cuda)
        im = cin_map.to(cuda) if cin_map is not None else None
        cm = col_map.to(cuda) if col_map is not None else None
        capi.check(capi.lib.rtpose_pack_pw_weights_cols(
            capi.ptr(wt_d), capi.ptr(b_d), h, h, capi.ptr(im) if im is not None else None, Kp, coutp,
            capi.ptr(cm) if cm is not None else None, coutp, 0, capi.ptr(wpk), capi.ptr(bpk), capi.current_stream()))
        keep.extend([wpk, bpk, wt_d, b_d, im, cm])
        return wpk, bpk
    wp0, bp0 = pack(w0, b0, x2map, None)
    wp2, bp2 = pack(w2, b2, None, cols)
    pln_d = pln.to(cuda)
    d = capi.PwDesc()
    d.inp, d.w_packed, d.bias_packed, d.out = cur.data_ptr(), wp0.data_ptr(), bp0.data_ptr(), t1.data_ptr()
    d.lin, d.lout, d.cin, d.cout, d.coutp, d.relu = lay, lt1, Kp, (h + 7) // 8 * 8, coutp, 1
    d.in_planes = pln_d.data_ptr()
    capi.check(capi.lib.rtpose_pw_fused(C.byref(d), n, H, w_map, capi.current_stream()), "conv.0")
    wdp = torch.zeros(9, Kp)
    wdp[:, :h] = wd.reshape(h, 9).t()
    bdp = torch.zeros(Kp)
    bdp[:h] = bd
    wdp_d, bdp_d = wdp.to(cuda), bdp.to(cuda)
    d2 = capi.PwDesc()
    d2.inp, d2.w_packed, d2.bias_packed, d2.out = t1.data_ptr(), wp2.data_ptr(), bp2.data_ptr(), nxt.data_ptr()
    d2.dw_w, d2.dw_b = wdp_d.data_ptr(), bdp_d.data_ptr()
    lodd = capi.Layout.padded(C4, H, w_map, 1, hp)                 # the odd runs = channels [hp, 2 hp) of the pixel
    d2.lin, d2.lout, d2.cin, d2.cout, d2.coutp, d2.relu = lt1, lodd, Kp, hp, coutp, 1
    d2.pt_src, d2.lpt = cur.data_ptr(), lay
    d2.pt_pairs, d2.pt_a, d2.pt_b, d2.pt_split, d2.pt_d0, d2.pt_d1 = hh_, 0, 2 * q, hh_, 0, q
    capi.check(capi.lib.rtpose_pw_fused(C.byref(d2), n, H, w_map, capi.current_stream()), "conv.1+conv.2+x1")
    got = _from_layout(capi, nxt, lay, C4, n, H, w_map, cuda)[:, perm]
    scale = max(1.0, ref.abs().max().item())
    assert (got[:, 0::2] - ref[:, 0::2]).abs().max().item() == 0.0          # the pass-through half is a copy
    assert (got - ref).abs().max().item() <= TOL * scale
    total, inside = nxt.abs().sum().item(), got.abs().sum().item()
    assert abs(total - inside) <= 1e-3 * max(1.0, inside), "kernel wrote outside the real channels / pixels"


def _rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("h", [58, 116, 232])
def test_unit_bf16_four_run_layout(capi, cuda, h):
    """The bf16 form (rtpose_pw_fused_bf16) of the same unit: bf16 activations / pointwise weights, fp32
    accumulate, depthwise taps and biases fp32, every stored activation rounded to bf16 - against the same
    arithmetic emulated in torch (oracle/shufflenet_oracle.py:_block_bf16 semantics).  Runs are 8-channel
    aligned (q = up8(h/2)); conv.2's columns are packed in run order (col_map) because the bf16 epilogue
    stores contiguous channels."""
    g = torch.Generator().manual_seed(h + 1)
    n, H, W = 2, 9, 46
    hh_, q = h // 2, (h // 2 + 7) // 8 * 8
    C4, K = 4 * q, 2 * q
    x = _rb(torch.randn(n, 2 * h, H, W, generator=g))
    w0 = torch.randn(h, h, generator=g) / h ** 0.5
    b0 = torch.randn(h, generator=g) * 0.1
    wd = torch.randn(h, 1, 3, 3, generator=g) * 0.3
    bd = torch.randn(h, generator=g) * 0.1
    w2 = torch.randn(h, h, generator=g) / h ** 0.5
    b2 = torch.randn(h, generator=g) * 0.1
    x1, x2 = x[:, :h], x[:, h:]
    t1 = _rb(F.relu(F.conv2d(x2.double(), _rb(w0)[:, :, None, None].double(), b0.double()).float()))
    t2 = _rb(F.conv2d(t1.double(), wd.double(), bd.double(), padding=1, groups=h).float())
    y = _rb(F.relu(F.conv2d(t2.double(), _rb(w2)[:, :, None, None].double(), b2.double()).float()))
    ref = torch.stack([x1, y], 2).reshape(n, 2 * h, H, W)

    def phys(j):
        i = j >> 1
        return (2 * (j & 1) + (1 if i >= hh_ else 0)) * q + (i - hh_ if i >= hh_ else i)
    perm = torch.tensor([phys(j) for j in range(2 * h)])
    xp = torch.zeros(n, C4, H, W)
    xp[:, perm] = x
    lay = capi.Layout.padded(C4, H, W, 1)
    npix = capi.lib.rtpose_layout_pixels(C.byref(lay), n, H, W)
    cur = torch.zeros(npix * C4, dtype=torch.int16, device=cuda)
    nxt = torch.zeros_like(cur)
    xs = xp.to(cuda).contiguous()
    capi.check(capi.lib.rtpose_nchw_to_layout_bf16(capi.ptr(xs), capi.ptr(cur), C.byref(lay), C4, C4, n, H, W,
                                                   capi.current_stream()))
    lt1 = capi.Layout.padded(K, H, W, 1)
    t1b = torch.zeros(capi.lib.rtpose_layout_pixels(C.byref(lt1), n, H, W) * K, dtype=torch.int16, device=cuda)
    coutp = (K + 63) // 64 * 64
    x2map = torch.full((K,), -1, dtype=torch.int32)
    colmap = torch.full((K,), -1, dtype=torch.int32)
    for k in range(K):
        p = k if k < q else k - q
        if p < hh_:
            x2map[k] = 2 * p + (0 if k < q else 1)
            colmap[k] = p if k < q else hh_ + p
    pln = torch.tensor([q + 8 * j if 8 * j < q else 3 * q + (8 * j - q) for j in range(K // 8)], dtype=torch.int32)
    keep = []

    def pack(wt, b, cin_map, ncols, col_map):
        wpk = torch.zeros(capi.lib.rtpose_packed_pw_bytes_bf16(K, coutp) // 2, dtype=torch.int16, device=cuda)
        bpk = torch.zeros(coutp, device=cuda)
        wt_d, b_d = wt.contiguous().to(cuda), b.to(cuda)
        cm = cin_map.to(cuda) if cin_map is not None else None
        cl = col_map.to(cuda) if col_map is not None else None
        capi.check(capi.lib.rtpose_pack_pw_weights_bf16(capi.ptr(wt_d), capi.ptr(b_d), h, h, capi.ptr(cm) if cm is not None else None,
                                                        K, ncols, capi.ptr(cl) if cl is not None else None, coutp, 0,
                                                        capi.ptr(wpk), capi.ptr(bpk), capi.current_stream()))
        keep.extend([wpk, bpk, wt_d, b_d, cm, cl])
        return wpk, bpk
    n0 = (h + 7) // 8 * 8
    wp0, bp0 = pack(w0, b0, x2map, n0, None)
    wp2, bp2 = pack(w2, b2, None, K, colmap)
    pln_d = pln.to(cuda)
    d = capi.PwDesc()
    d.inp, d.w_packed, d.bias_packed, d.out = cur.data_ptr(), wp0.data_ptr(), bp0.data_ptr(), t1b.data_ptr()
    d.lin, d.lout, d.cin, d.cout, d.coutp, d.relu = lay, lt1, K, n0, coutp, 1
    d.in_planes = pln_d.data_ptr()
    capi.check(capi.lib.rtpose_pw_fused_bf16(C.byref(d), 0, n, H, W, capi.current_stream()), "conv.0 bf16")
    wdp = torch.zeros(9, K)
    wdp[:, :h] = wd.reshape(h, 9).t()
    bdp = torch.zeros(K)
    bdp[:h] = bd
    wdp_d, bdp_d = wdp.to(cuda), bdp.to(cuda)
    d2 = capi.PwDesc()
    d2.inp, d2.w_packed, d2.bias_packed, d2.out = t1b.data_ptr(), wp2.data_ptr(), bp2.data_ptr(), nxt.data_ptr()
    d2.dw_w, d2.dw_b = wdp_d.data_ptr(), bdp_d.data_ptr()
    lodd = capi.Layout.padded(C4, H, W, 1, choff=2 * q)
    d2.lin, d2.lout, d2.cin, d2.cout, d2.coutp, d2.relu = lt1, lodd, K, K, coutp, 1
    d2.pt_src, d2.lpt = cur.data_ptr(), lay
    d2.pt_pairs, d2.pt_a, d2.pt_b, d2.pt_split, d2.pt_d0, d2.pt_d1 = hh_, 0, 2 * q, hh_, 0, q
    capi.check(capi.lib.rtpose_pw_fused_bf16(C.byref(d2), 0, n, H, W, capi.current_stream()), "conv.1+conv.2+x1 bf16")
    # read back: bf16 layout -> fp32 layout -> NCHW
    f32buf = torch.zeros(npix * C4, device=cuda)
    capi.check(capi.lib.rtpose_layout_bf16_to_f32(capi.ptr(nxt), C.byref(lay), capi.ptr(f32buf), C.byref(lay), C4, n, H, W,
                                                  capi.current_stream()))
    got = _from_layout(capi, f32buf, lay, C4, n, H, W, cuda)
    pads = [c for c in range(C4) if c not in set(perm.tolist())]
    assert got[:, pads].abs().max().item() == 0.0                      # the run padding stays zero
    got = got[:, perm]
    assert (got[:, 0::2] - ref[:, 0::2]).abs().max().item() == 0.0     # the pass-through half is a copy
    scale = max(1.0, ref.abs().max().item())
    # one bf16 ulp (2^-8 relative) where the fp32 dw accumulation order flips a rounding, two stages deep
    assert (got - ref).abs().max().item() <= 2e-2 * scale
    assert ((got - ref).abs() > 1e-6).float().mean().item() < 0.05    # ... and almost everywhere identical


@pytest.mark.parametrize("n,h,w,cin,c1", [(2, 46, 46, 464, 1024), (1, 9, 7, 32, 256), (3, 13, 11, 48, 512),
                                          (1, 5, 5, 464, 1024)])
def test_wide_conv_and_heads_in_one_launch(capi, cuda, n, h, w, cin, c1):
    """rtpose_pw_head (csrc/pw_head.hip): conv5 (cin -> c1, ReLU) + the two heads (c1 -> 38 | 19) of
    lib/network/rtpose_shufflenetV2.py:104-108, :143-147 as one launch against F.conv2d on the CPU - the K = 464 / 1024
    shape of the network, a short-K shape whose last chunk is whole (32), one with a two-group last chunk (48), pixel
    counts that are not a multiple of the 32-pixel work item, an input slice at a channel offset inside a wider
    pixel, and more waves than work items."""
    g = torch.Generator().manual_seed(n * 1000 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    w1 = torch.randn(c1, cin, generator=g) / cin ** 0.5
    b1 = torch.randn(c1, generator=g) * 0.1
    wp_, bp_ = torch.randn(38, c1, generator=g) / c1 ** 0.5, torch.randn(38, generator=g) * 0.1
    wh_, bh_ = torch.randn(19, c1, generator=g) / c1 ** 0.5, torch.randn(19, generator=g) * 0.1
    f = F.relu(F.conv2d(x, w1[:, :, None, None], b1))
    ref_p, ref_h = F.conv2d(f, wp_[:, :, None, None], bp_), F.conv2d(f, wh_[:, :, None, None], bh_)
    lib, stream = capi.lib, capi.current_stream()
    choff = 8
    lin = capi.Layout.padded(cin + 16, h, w, 1, choff)      # the conv's input is a slice of a wider pixel
    xin = torch.full((lib.rtpose_layout_pixels(C.byref(lin), n, h, w) * lin.cstride,), float("nan"), device=cuda)
    xin.view(-1, lin.cstride)[:, choff:choff + cin] = 0.0  # (gap pixels of the slice: zero; the rest of the pixel: NaN)
    capi.check(lib.rtpose_nchw_to_layout(capi.ptr(x.to(cuda).contiguous()), capi.ptr(xin), C.byref(lin), cin, cin, n, h, w,
                                         stream))
    w1p = torch.zeros(lib.rtpose_packed_pw_floats(cin, c1) + 64 * c1, device=cuda)
    b1p = torch.zeros(c1, device=cuda)
    w2p = torch.full((lib.rtpose_packed_pw_floats(c1, 64) + 64 * 64,), float("nan"), device=cuda)
    b2p = torch.full((64,), float("nan"), device=cuda)
    dev = lambda t: t.contiguous().to(cuda)   # noqa: E731
    keep = [dev(w1), dev(b1), dev(wp_), dev(bp_), dev(wh_), dev(bh_)]
    capi.check(lib.rtpose_pack_pw_weights(capi.ptr(keep[0]), capi.ptr(keep[1]), c1, cin, None, cin, c1, 0,
                                          capi.ptr(w1p), capi.ptr(b1p), stream))
    w2p[:c1 * 64].zero_()      # columns nobody owns must be zero (the executor zeroes them at load time)
    b2p.zero_()
    capi.check(lib.rtpose_pack_pw_weights(capi.ptr(keep[2]), capi.ptr(keep[3]), 38, c1, None, c1, 64, 0,
                                          capi.ptr(w2p), capi.ptr(b2p), stream))
    capi.check(lib.rtpose_pack_pw_weights(capi.ptr(keep[4]), capi.ptr(keep[5]), 19, c1, None, c1, 64, 40,
                                          capi.ptr(w2p), capi.ptr(b2p), stream))
    lout = capi.Layout.dense(72, h, w, 4)
    out = torch.full((lib.rtpose_layout_pixels(C.byref(lout), n, h, w) * 72,), 7.0, device=cuda)
    d1, d2 = capi.PwDesc(), capi.PwDesc()
    d1.inp, d1.w_packed, d1.bias_packed = xin.data_ptr(), w1p.data_ptr(), b1p.data_ptr()
    d1.lin, d1.cin, d1.cout, d1.coutp, d1.relu = lin, cin, c1, c1, 1
    d2.w_packed, d2.bias_packed, d2.out = w2p.data_ptr(), b2p.data_ptr(), out.data_ptr()
    d2.lout, d2.cin, d2.cout, d2.coutp, d2.relu = lout, c1, 64, 64, 0
    assert lib.rtpose_pw_head_fits(C.byref(d1), C.byref(d2)) == 1
    capi.check(lib.rtpose_pw_head(C.byref(d1), C.byref(d2), n, h, w, stream), "rtpose_pw_head")
    torch.cuda.synchronize()
    px = out.view(-1, 72).cpu()
    assert torch.equal(px[:, :4], torch.full_like(px[:, :4], 7.0)) and torch.equal(px[:, 68:], torch.full_like(px[:, 68:], 7.0))
    assert torch.equal(px[n * h * w:], torch.full_like(px[n * h * w:], 7.0))      # nothing past the last pixel
    px = px[:n * h * w]                                                           # (dense layout: pixel q = (n h + y) w + x)
    got = px[:, 4:68].reshape(n, h, w, 64).permute(0, 3, 1, 2)
    scale = max(1.0, ref_p.abs().max().item())
    assert (got[:, 0:38] - ref_p).abs().max().item() <= TOL * scale
    assert (got[:, 40:59] - ref_h).abs().max().item() <= TOL * scale
    assert got[:, 38:40].abs().max().item() == 0.0 and got[:, 59:64].abs().max().item() == 0.0
    # the same launch again: bit-identical (fixed summation order, no atomics)
    out2 = torch.full_like(out, 7.0)
    d2.out = out2.data_ptr()
    capi.check(lib.rtpose_pw_head(C.byref(d1), C.byref(d2), n, h, w, stream), "rtpose_pw_head")
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    # shapes without an instance are refused, not mis-run
    d1.cin = cin + 8
    assert lib.rtpose_pw_head_fits(C.byref(d1), C.byref(d2)) == 0
    assert lib.rtpose_pw_head(C.byref(d1), C.byref(d2), n, h, w, stream) != 0


@pytest.mark.parametrize("n,h,w,cin,c1,planes", [(20, 46, 46, 480, 1024, True), (1, 9, 7, 32, 256, False),
                                                  (3, 13, 11, 64, 512, False), (1, 5, 5, 480, 1024, False),
                                                  (2, 23, 17, 112, 768, True)])
def test_wide_conv_and_heads_in_one_launch_bf16(capi, cuda, n, h, w, cin, c1, planes):
    """rtpose_pw_head_bf16 (csrc/pw_head_bf16.hip): conv5 (cin -> c1, ReLU) + the two heads (c1 -> 38 | 19) of
    lib/network/rtpose_shufflenetV2.py:104-108, :143-147 as one launch in the arithmetic of the bf16 plan - bf16
    activations and weights, fp32 sums, the conv5 feature rounded to bf16 after its ReLU, fp32 maps - against the same
    arithmetic in float64 on the CPU (oracle/shufflenet_oracle.py:forward_bf16_emulated semantics): the K = 480 / 1024
    shape of the network with the input gathered as 16-byte planes of a wider pixel, K of 2 / 4 / 7 k-steps (whole and
    zero-padded last chunks), 1 - 4 passes, pixel counts that are not a multiple of the 128-pixel item (several items per
    block: the LDS tile is refilled under the last pass), and more blocks than items."""
    g = torch.Generator().manual_seed(n * 1000 + cin)
    x = _rb(torch.randn(n, cin, h, w, generator=g))
    w1 = _rb(torch.randn(c1, cin, generator=g) / cin ** 0.5)
    b1 = torch.randn(c1, generator=g) * 0.1
    wp_, bp_ = _rb(torch.randn(38, c1, generator=g) / c1 ** 0.5), torch.randn(38, generator=g) * 0.1
    wh_, bh_ = _rb(torch.randn(19, c1, generator=g) / c1 ** 0.5), torch.randn(19, generator=g) * 0.1
    f = _rb(F.relu(F.conv2d(x.double(), w1[:, :, None, None].double(), b1.double()).float())).double()
    ref_p = F.conv2d(f, wp_[:, :, None, None].double(), bp_.double()).float()
    ref_h = F.conv2d(f, wh_[:, :, None, None].double(), bh_.double()).float()
    lib, stream = capi.lib, capi.current_stream()
    # the input: channel c of the conv at physical channel phys[c] of a wider pixel (planes of 8 in shuffled order)
    npl = cin // 8
    if planes:
        order = torch.randperm(npl + 3, generator=g)[:npl]          # plane j of K sits at plane order[j] of the pixel
        cphys = (npl + 3) * 8
    else:
        order = torch.arange(npl) + 1                                # a contiguous slice at channel offset 8
        cphys = (npl + 2) * 8
    phys = (order[:, None] * 8 + torch.arange(8)[None, :]).reshape(-1)
    xp = torch.full((n, cphys, h, w), 3.0)                           # (channels outside the slice: junk the kernel must not read)
    xp[:, phys] = x
    lin = capi.Layout.padded(cphys, h, w, 1, 0 if planes else 8)
    npix = lib.rtpose_layout_pixels(C.byref(lin), n, h, w)
    xin = torch.zeros(npix * cphys, dtype=torch.int16, device=cuda)
    lfull = capi.Layout.padded(cphys, h, w, 1)
    capi.check(lib.rtpose_nchw_to_layout_bf16(capi.ptr(xp.to(cuda).contiguous()), capi.ptr(xin), C.byref(lfull), cphys, cphys,
                                              n, h, w, stream))
    w1p = torch.zeros(lib.rtpose_packed_pw_bytes_bf16(cin, c1) // 2, dtype=torch.int16, device=cuda)
    b1p = torch.zeros(c1, device=cuda)
    w2p = torch.full((c1 * 64 + 4096,), 0x7fc0, dtype=torch.int16, device=cuda)      # NaN until packed / zeroed
    b2p = torch.full((64,), float("nan"), device=cuda)
    dev = lambda t: t.contiguous().to(cuda)   # noqa: E731
    keep = [dev(w1), dev(b1), dev(wp_), dev(bp_), dev(wh_), dev(bh_)]
    capi.check(lib.rtpose_pack_pw_weights_bf16(capi.ptr(keep[0]), capi.ptr(keep[1]), c1, cin, None, cin, c1, None, c1, 0,
                                               capi.ptr(w1p), capi.ptr(b1p), stream))
    w2p[:c1 * 64].zero_()      # columns nobody owns must be zero (the executor zeroes them at load time)
    b2p.zero_()
    capi.check(lib.rtpose_pack_pw_head2_bf16(capi.ptr(keep[2]), capi.ptr(keep[3]), 38, c1, 0, capi.ptr(w2p), capi.ptr(b2p), stream))
    capi.check(lib.rtpose_pack_pw_head2_bf16(capi.ptr(keep[4]), capi.ptr(keep[5]), 19, c1, 40, capi.ptr(w2p), capi.ptr(b2p), stream))
    lout = capi.Layout.dense(72, h, w, 4)
    out = torch.full((lib.rtpose_layout_pixels(C.byref(lout), n, h, w) * 72,), 7.0, device=cuda)
    d1, d2 = capi.PwDesc(), capi.PwDesc()
    d1.inp, d1.w_packed, d1.bias_packed = xin.data_ptr(), w1p.data_ptr(), b1p.data_ptr()
    d1.lin, d1.cin, d1.cout, d1.coutp, d1.relu = lin, cin, c1, c1, 1
    pl_d = (order * 8).to(torch.int32).to(cuda)
    if planes:
        d1.in_planes = pl_d.data_ptr()
    d2.w_packed, d2.bias_packed, d2.out = w2p.data_ptr(), b2p.data_ptr(), out.data_ptr()
    d2.lout, d2.cin, d2.cout, d2.coutp, d2.relu = lout, c1, 64, 64, 0
    assert lib.rtpose_pw_head_bf16_fits(C.byref(d1), C.byref(d2)) == 1
    capi.check(lib.rtpose_pw_head_bf16(C.byref(d1), C.byref(d2), n, h, w, stream), "rtpose_pw_head_bf16")
    torch.cuda.synchronize()
    px = out.view(-1, 72).cpu()
    assert torch.equal(px[:, :4], torch.full_like(px[:, :4], 7.0)) and torch.equal(px[:, 68:], torch.full_like(px[:, 68:], 7.0))
    assert torch.equal(px[n * h * w:], torch.full_like(px[n * h * w:], 7.0))      # nothing past the last pixel
    px = px[:n * h * w]
    got = px[:, 4:68].reshape(n, h, w, 64).permute(0, 3, 1, 2)
    scale = max(1.0, ref_p.abs().max().item())
    # fp32 sums of exact bf16 products against float64 ones; a feature value on a bf16 rounding boundary may round the
    # other way (one bf16 ulp of one of 1024 terms)
    assert (got[:, 0:38] - ref_p).abs().max().item() <= 2e-3 * scale
    assert (got[:, 40:59] - ref_h).abs().max().item() <= 2e-3 * scale
    assert got[:, 38:40].abs().max().item() == 0.0 and got[:, 59:64].abs().max().item() == 0.0
    # the same launch again: bit-identical (fixed summation order, no atomics)
    out2 = torch.full_like(out, 7.0)
    d2.out = out2.data_ptr()
    capi.check(lib.rtpose_pw_head_bf16(C.byref(d1), C.byref(d2), n, h, w, stream), "rtpose_pw_head_bf16")
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    # an image's maps do not depend on its batch: image 0 alone (other items, other blocks) gives the same bits
    if n > 1:
        out3 = torch.full_like(out, 7.0)
        d2.out = out3.data_ptr()
        capi.check(lib.rtpose_pw_head_bf16(C.byref(d1), C.byref(d2), 1, h, w, stream), "rtpose_pw_head_bf16")
        torch.cuda.synchronize()
        assert torch.equal(out3.view(-1, 72)[:h * w], out.view(-1, 72)[:h * w])
    # shapes without an instance are refused, not mis-run
    d1.cin = 488
    assert lib.rtpose_pw_head_bf16_fits(C.byref(d1), C.byref(d2)) == 0
    assert lib.rtpose_pw_head_bf16(C.byref(d1), C.byref(d2), n, h, w, stream) != 0


@pytest.mark.parametrize("h,n,H,W,k1pad", [(232, 3, 46, 46, 24), (116, 2, 21, 13, 8), (58, 2, 9, 46, 6), (116, 12, 46, 46, 44)])
def test_unit_in_one_launch_bf16(capi, cuda, h, n, H, W, k1pad):
    """rtpose_unit_bf16 (csrc/unit_bf16.hip): conv.0 -> depthwise 3x3 -> conv.2 of a stride-1 unit
    (lib/network/rtpose_shufflenetV2.py:31-39) as ONE launch, against the bf16 arithmetic emulated in torch
    (oracle/shufflenet_oracle.py:_block_bf16 semantics: every conv output rounded to bf16, depthwise taps fp32) - x2
    gathered as shuffled 16-byte planes of a wider pixel (K padded with zero-weight repeats of the first plane), y
    stored in shuffled 8-channel groups of OTHER slots of the same buffer, maps that are not a multiple of the 8 x 8
    tile, more tiles than CUs (the tile buffers ping-pong), and the three channel widths of the network."""
    g = torch.Generator().manual_seed(h + n)
    Kt = (h + 15) // 16 * 16
    npl_real = (h + 7) // 8
    K1 = (npl_real * 8 + k1pad + 15) // 16 * 16                  # the gather may be wider than h (zero rows)
    npl1 = K1 // 8
    ng_out = (h + 7) // 8                                        # 8-channel groups of y
    nslots = npl1 + ng_out + 5                                   # planes of the pixel: x2's, y's and some nobody touches
    perm = torch.randperm(nslots, generator=g)
    in_pl = perm[:npl1].clone()                                  # plane j of K sits at plane in_pl[j]
    in_pl[npl_real:] = in_pl[0]                                  # (padding planes: repeats of plane 0 under zero weights)
    out_gr = perm[npl1:npl1 + ng_out]
    Cp = nslots * 8
    x2 = _rb(torch.randn(n, h, H, W, generator=g))
    w0 = torch.randn(h, h, generator=g) / h ** 0.5
    b0 = torch.randn(h, generator=g) * 0.1
    wd = torch.randn(h, 1, 3, 3, generator=g) * 0.3
    bd = torch.randn(h, generator=g) * 0.1
    w2 = torch.randn(h, h, generator=g) / h ** 0.5
    b2 = torch.randn(h, generator=g) * 0.1
    t1 = _rb(F.relu(F.conv2d(x2.double(), _rb(w0)[:, :, None, None].double(), b0.double()).float()))
    t2 = _rb(F.conv2d(t1.double(), wd.double(), bd.double(), padding=1, groups=h).float())
    ref = _rb(F.relu(F.conv2d(t2.double(), _rb(w2)[:, :, None, None].double(), b2.double()).float()))
    lib, stream = capi.lib, capi.current_stream()
    # the buffer: channel c of x2 at physical in_pl[c // 8] * 8 + c % 8; everything else junk that must survive
    xp = torch.full((n, Cp, H, W), 5.0)
    hp8 = npl_real * 8
    xpad = torch.zeros(n, hp8, H, W)
    xpad[:, :h] = x2
    phys_in = (in_pl[:npl_real, None] * 8 + torch.arange(8)[None, :]).reshape(-1)
    xp[:, phys_in] = xpad
    lay = capi.Layout.padded(Cp, H, W, 1)
    npix = lib.rtpose_layout_pixels(C.byref(lay), n, H, W)
    buf = torch.zeros(npix * Cp, dtype=torch.int16, device=cuda)
    capi.check(lib.rtpose_nchw_to_layout_bf16(capi.ptr(xp.to(cuda).contiguous()), capi.ptr(buf), C.byref(lay), Cp, Cp, n, H, W,
                                              stream))
    before = buf.clone()
    c1p = 128 if Kt <= 128 else 256
    ncols2 = ng_out * 8
    c2p = 128 if ncols2 <= 128 else 256
    keep = []

    def pack(wt, b, cin_map, K, ncols, col_map, coutp):
        wpk = torch.zeros(lib.rtpose_packed_pw_bytes_bf16(K, coutp) // 2, dtype=torch.int16, device=cuda)
        bpk = torch.zeros(coutp, device=cuda)
        wt_d, b_d = wt.contiguous().to(cuda), b.to(cuda)
        cm = cin_map.to(torch.int32).to(cuda) if cin_map is not None else None
        cl = col_map.to(torch.int32).to(cuda) if col_map is not None else None
        capi.check(lib.rtpose_pack_pw_weights_bf16(capi.ptr(wt_d), capi.ptr(b_d), h, h, capi.ptr(cm) if cm is not None else None,
                                                   K, ncols, capi.ptr(cl) if cl is not None else None, coutp, 0,
                                                   capi.ptr(wpk), capi.ptr(bpk), stream))
        keep.extend([wpk, bpk, wt_d, b_d, cm, cl])
        return wpk, bpk
    x2map = torch.full((K1,), -1, dtype=torch.int32)              # packed K row -> x2 channel (padding rows: zero)
    x2map[:h] = torch.arange(h, dtype=torch.int32)
    wp0, bp0 = pack(w0, b0, x2map, K1, c1p, None, c1p)            # every column packed: zeros past h
    colmap = torch.full((c2p,), -1, dtype=torch.int32)            # packed column -> y channel
    colmap[:h] = torch.arange(h, dtype=torch.int32)
    wp2, bp2 = pack(w2, b2, None, Kt, c2p, colmap, c2p)
    wdp = torch.zeros(9, Kt)
    wdp[:, :h] = wd.reshape(h, 9).t()
    bdp = torch.zeros(Kt)
    bdp[:h] = bd
    wdp_d, bdp_d = wdp.to(cuda), bdp.to(cuda)
    chan = torch.full((c2p,), -1, dtype=torch.int32)              # column -> absolute channel of the pixel
    for gi in range(ng_out):
        chan[8 * gi:8 * gi + 8] = out_gr[gi] * 8 + torch.arange(8, dtype=torch.int32)
    chan_d = chan.to(cuda)
    pl_d = (in_pl * 8).to(torch.int32).to(cuda)
    d0, d2 = capi.PwDesc(), capi.PwDesc()
    d0.inp, d0.w_packed, d0.bias_packed = buf.data_ptr(), wp0.data_ptr(), bp0.data_ptr()
    d0.lin, d0.cin, d0.cout, d0.coutp, d0.relu = lay, K1, c1p, c1p, 1
    d0.in_planes, d0.dw_w, d0.dw_b = pl_d.data_ptr(), wdp_d.data_ptr(), bdp_d.data_ptr()
    d2.w_packed, d2.bias_packed, d2.out = wp2.data_ptr(), bp2.data_ptr(), buf.data_ptr()
    d2.lout, d2.cin, d2.cout, d2.coutp, d2.relu = lay, Kt, ncols2, c2p, 1
    d2.out_cmap = chan_d.data_ptr()
    assert lib.rtpose_unit_bf16_fits(C.byref(d0), C.byref(d2), H, W) == 1
    capi.check(lib.rtpose_unit_bf16(C.byref(d0), C.byref(d2), n, H, W, stream), "rtpose_unit_bf16")
    torch.cuda.synchronize()
    f32buf = torch.zeros(npix * Cp, device=cuda)
    capi.check(lib.rtpose_layout_bf16_to_f32(capi.ptr(buf), C.byref(lay), capi.ptr(f32buf), C.byref(lay), Cp, n, H, W, stream))
    got_all = _from_layout(capi, f32buf, lay, Cp, n, H, W, cuda)
    phys_out = (out_gr[:, None] * 8 + torch.arange(8)[None, :]).reshape(-1)
    got = got_all[:, phys_out[:h]]
    assert got_all[:, phys_out[h:]].abs().max().item() == 0.0 if len(phys_out) > h else True    # group padding: relu(0) = 0
    untouched = [c for c in range(Cp) if c not in set(phys_out.tolist())]
    capi.check(lib.rtpose_layout_bf16_to_f32(capi.ptr(before), C.byref(lay), capi.ptr(f32buf), C.byref(lay), Cp, n, H, W, stream))
    was = _from_layout(capi, f32buf, lay, Cp, n, H, W, cuda)
    assert torch.equal(got_all[:, untouched], was[:, untouched])          # x2 and the bystanders are as they were
    assert torch.equal(buf.view(-1, Cp)[:, untouched], before.view(-1, Cp)[:, untouched])   # ... gaps included
    scale = max(1.0, ref.abs().max().item())
    # one bf16 ulp (2^-8 relative) where an fp32 sum lands on the other side of a rounding boundary, three stages deep
    assert (got - ref).abs().max().item() <= 2e-2 * scale
    assert ((got - ref).abs() > 1e-6).float().mean().item() < 0.05    # ... and almost everywhere identical
    # an image's result does not depend on its batch
    if n > 1:
        buf1 = before.clone()
        d0.inp = d2.out = buf1.data_ptr()
        capi.check(lib.rtpose_unit_bf16(C.byref(d0), C.byref(d2), 1, H, W, stream), "rtpose_unit_bf16")
        torch.cuda.synchronize()
        rows = lay.lead + H * (W + 1)           # every pixel slot of image 0
        assert torch.equal(buf1.view(-1, Cp)[:rows], buf.view(-1, Cp)[:rows])
    # shapes without an instance are refused, not mis-run
    d0.cin = 280
    assert lib.rtpose_unit_bf16_fits(C.byref(d0), C.byref(d2), H, W) == 0
    assert lib.rtpose_unit_bf16(C.byref(d0), C.byref(d2), n, H, W, stream) != 0


def test_bad_arguments_fail_loudly(capi, cuda):
    d = capi.PwDesc()
    assert capi.lib.rtpose_pw_fused(C.byref(d), 1, 8, 8, None) != 0
    t = torch.zeros(4096, device=cuda)
    d.inp = d.w_packed = d.bias_packed = d.out = t.data_ptr()
    d.lin = d.lout = capi.Layout.dense(8, 8, 8)
    d.cin, d.cout, d.coutp = 8, 8, 96          # coutp must be 64, 128 or k * 256
    assert capi.lib.rtpose_pw_fused(C.byref(d), 1, 8, 8, None) != 0
    d.coutp, d.dw_w, d.dw_b = 64, t.data_ptr(), t.data_ptr()   # depthwise input without a layout gap
    assert capi.lib.rtpose_pw_fused(C.byref(d), 1, 8, 8, None) != 0
