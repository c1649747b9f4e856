"""GPU parity of the bf16x3 plan (split bf16 operands, three bf16 MFMAs per product;
csrc/conv_mfma_bf16.hip SP = 2).  It is meant to deliver fp32-grade maps, so the headline check
is the fp32 path's own bound: within 1e-3 (absolute, BASELINE.json north_star) of the fp32
oracle restatement of the reference module.  Tolerances next to each check."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _run_conv_x3(capi, dev, n, h, w, cin, cout, k, relu, pool, pad_in, pad_out, seed, groups=1, out_f32=False):
    lib, Layout = capi.lib, capi.Layout
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    cin_p = (cin + 15) // 16 * 16
    ws, bs, refs = [], [], []
    for gi in range(groups):
        wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        y = F.conv2d(x.double(), wt.double(), b.double(), padding=k // 2).float()   # exact fp32-operand conv
        if relu:
            y = F.relu(y)
        if pool:
            y = F.max_pool2d(y, 2, 2, 0)
        ws.append(wt.to(dev))
        bs.append(b.to(dev))
        refs.append(y)
    stream = capi.current_stream()
    lin = Layout.padded(2 * cin_p, h, w, pad_in)          # elements: 2 per channel
    npx = lib.rtpose_layout_pixels(C.byref(lin), n, h, w)
    xin = torch.zeros(npx * 2 * cin_p, device=dev, dtype=torch.bfloat16)
    xd = x.to(dev)
    capi.check(lib.rtpose_nchw_to_layout_split(capi.ptr(xd), capi.ptr(xin), C.byref(lin), cin, cin_p, n, h, w, stream))
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    cpo = (cout + 7) // 8 * 8
    ctot = cpo * groups + 8                               # channels per pixel of the output buffer
    descs = (capi.ConvDesc * groups)()
    outs, keep = [], []
    if out_f32:
        lfull = Layout.padded(ctot, ho, wo, pad_out)
        obuf = torch.zeros(lib.rtpose_layout_pixels(C.byref(lfull), n, ho, wo) * ctot, device=dev)
    else:
        lfull = Layout.padded(2 * ctot, ho, wo, pad_out)
        obuf = torch.zeros(lib.rtpose_layout_pixels(C.byref(lfull), n, ho, wo) * 2 * ctot, device=dev,
                           dtype=torch.bfloat16)
    for gi in range(groups):
        wp = torch.zeros(lib.rtpose_packed_weight_bytes_bf16x3(cout, cin_p, k) // 2, device=dev, dtype=torch.bfloat16)
        bp = torch.zeros(lib.rtpose_packed_bias_floats(cout), device=dev)
        capi.check(lib.rtpose_pack_conv_weights_bf16x3(capi.ptr(ws[gi]), capi.ptr(bs[gi]), cout, cin, k, None, cin_p,
                                                       capi.ptr(wp), capi.ptr(bp), stream))
        keep += [wp, bp]
        d = descs[gi]
        d.inp, d.w_packed, d.bias_packed, d.out = xin.data_ptr(), wp.data_ptr(), bp.data_ptr(), obuf.data_ptr()
        d.lin = lin
        choff = gi * cpo + 8
        d.lout = Layout.padded(ctot, ho, wo, pad_out, choff=choff) if out_f32 else \
            Layout.padded(2 * ctot, ho, wo, pad_out, choff=2 * choff)
        d.cin, d.cout, d.k, d.relu, d.pool = cin_p, cout, k, int(relu), int(pool)
    capi.check(lib.rtpose_conv2d_bf16x3(descs, groups, n, h, w, int(out_f32), stream), "rtpose_conv2d_bf16x3")
    for gi in range(groups):
        choff = gi * cpo + 8
        if out_f32:
            o = torch.empty(n, cout, ho, wo, device=dev)
            lo = Layout.padded(ctot, ho, wo, pad_out, choff=choff)
            capi.check(lib.rtpose_layout_to_nchw(capi.ptr(obuf), C.byref(lo), capi.ptr(o), cout, n, ho, wo, stream))
        else:
            dense = torch.empty(n, ho, wo, cout, device=dev)
            lo = Layout.padded(2 * ctot, ho, wo, pad_out, choff=2 * choff)
            ld = Layout.dense(cout, ho, wo)
            capi.check(lib.rtpose_layout_split_to_f32(capi.ptr(obuf), C.byref(lo), capi.ptr(dense), C.byref(ld), cout,
                                                      n, ho, wo, stream))
            o = dense.permute(0, 3, 1, 2).contiguous()
        outs.append(o.cpu())
    torch.cuda.synchronize()
    return outs, refs


CASES = [
    # n, h, w, cin, cout, k, relu, pool, pad_in, pad_out
    (2, 46, 46, 128, 128, 7, 1, 0, 3, 3),     # Mconv2_stageN: strip mode
    (1, 46, 49, 185, 128, 7, 1, 0, 3, 3),     # 185 -> 192 padded input
    (3, 23, 17, 128, 38, 1, 0, 0, 0, 3),      # 1x1 head, scalar stores, ragged M
    (2, 46, 46, 256, 512, 3, 1, 0, 1, 1),     # conv4_1
    (1, 96, 80, 64, 64, 3, 1, 1, 1, 1),       # 2-D tile mode + fused pool
    (2, 72, 88, 3, 64, 3, 1, 0, 1, 1),        # conv1_1 (3 -> 16 padded input)
    (1, 70, 66, 128, 128, 7, 1, 0, 3, 0),     # 7x7 in 2-D tile mode
    (5, 6, 6, 32, 8, 7, 1, 0, 3, 3),          # strip spanning several images, cout 8
]


@pytest.mark.parametrize("case", CASES)
def test_conv_bf16x3_is_fp32_grade(capi, cuda, case):
    """One conv against the exact fp32-operand convolution: operands carry 16 significant bits and
    the lo*lo term is dropped, so the error is ~2^-16 of sum |a||b| - asserted as 3e-5 of max|ref|
    (including one more 2^-17 relative for re-splitting the stored output)."""
    n, h, w, cin, cout, k, relu, pool, pin, pout = case
    outs, refs = _run_conv_x3(capi, cuda, n, h, w, cin, cout, k, relu, pool, pin, pout, seed=hash(case) % 1000)
    scale = max(1.0, refs[0].abs().max().item())
    err = (outs[0] - refs[0]).abs().max().item()
    assert err <= 3e-5 * scale, "max abs err %g (scale %g)" % (err, scale)


def test_conv_bf16x3_grouped_fp32_out(capi, cuda):
    outs, refs = _run_conv_x3(capi, cuda, 5, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=7, groups=2, out_f32=True)
    for o, r in zip(outs, refs):
        assert (o - r).abs().max().item() <= 3e-5 * max(1.0, r.abs().max().item())


@pytest.fixture(scope="module")
def model_and_sd(pkg, cuda):
    from oracle import net_oracle
    m = pkg.get_model('vgg19')
    sd = net_oracle.he_init_state_dict(m, seed=0)
    m.load_state_dict(sd)
    m = m.cuda().float().eval()
    return m, sd


@pytest.mark.parametrize("shape", [(2, 3, 64, 72), (1, 3, 56, 40), (1, 3, 368, 368)])
def test_net_bf16x3_within_the_fp32_contract(model_and_sd, cuda, shape):
    """Whole network: every stage output within 1e-3 (absolute) of the fp32 oracle - the bound
    BASELINE.json's north_star puts on the fp32 path - and within 2e-4 of max|ref| of the
    split-operand emulation."""
    from oracle import net_oracle
    m, sd = model_and_sd
    g = torch.Generator().manual_seed(1)
    x = torch.rand(shape, generator=g) - 0.5
    (paf_r, heat_r), saved_r = net_oracle.forward(sd, x)
    m.set_compute_dtype('bf16x3')
    try:
        with torch.no_grad():
            (paf, heat), saved = m(x.to(cuda))
    finally:
        m.set_compute_dtype('fp32')
    worst = 0.0
    for i, (a, b) in enumerate(zip(saved, saved_r)):
        err = (a.cpu() - b).abs().max().item()
        worst = max(worst, err)
        assert err <= 1e-3, "stage output %d: max abs err %g vs the fp32 oracle (ref max %g)" % (
            i, err, b.abs().max().item())
    if shape[2] <= 72:
        (paf_e, heat_e), saved_e = net_oracle.forward_bf16x3_emulated(sd, x)
        for a, b in zip(saved, saved_e):
            assert (a.cpu() - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item())
    print("bf16x3 worst abs err vs fp32 oracle at %s: %.3g" % (shape, worst))


def test_bf16x3_keypoints_identical_to_fp32(pkg, model_and_sd, cuda):
    from importlib import import_module
    synth = import_module(pkg.__name__ + ".synth")
    pipeline = import_module(pkg.__name__ + ".pipeline")
    m, _ = model_and_sd
    n, hw = 4, 128
    x = (torch.rand(n, 3, hw, hw, generator=torch.Generator().manual_seed(5)) - 0.5).to(cuda)
    heat, paf, _ = synth.make_batch(n, hw, hw, seed=2, max_people=3)
    est = pipeline.PoseEstimator(m)
    scene = (torch.from_numpy(heat).to(cuda), torch.from_numpy(paf).to(cuda))
    res = {}
    for dt in ('fp32', 'bf16x3'):
        m.set_compute_dtype(dt)
        try:
            res[dt] = est(x, scene=scene, scene_alpha=2e-2)
        finally:
            m.set_compute_dtype('fp32')
    for a, b in zip(res['fp32'], res['bf16x3']):
        assert np.array_equal(a["parts"], b["parts"]), "person/part assignment differs"
        assert np.array_equal(a["peaks"][:, :2], b["peaks"][:, :2]), "peak coordinates differ"


def test_bf16x3_full_size_batch_properties(model_and_sd, cuda):
    """32 x 3 x 368 x 368: batch-position independence and determinism, bit for bit."""
    m, _ = model_and_sd
    g = torch.Generator().manual_seed(13)
    x = torch.rand(32, 3, 368, 368, generator=g) - 0.5
    perm = torch.randperm(32, generator=g)
    keep = m.keep_intermediates
    m.keep_intermediates = False
    m.set_compute_dtype('bf16x3')
    try:
        with torch.no_grad():
            (paf, heat), _ = m(x.to(cuda))
            (paf2, heat2), _ = m(x.to(cuda))
            (paf_p, heat_p), _ = m(x[perm].to(cuda))
    finally:
        m.set_compute_dtype('fp32')
        m.keep_intermediates = keep
    assert torch.equal(paf, paf2) and torch.equal(heat, heat2)
    assert torch.equal(paf_p, paf[perm.to(cuda)]) and torch.equal(heat_p, heat[perm.to(cuda)])


def test_conv_bf16x3_error_bound_wide_dynamic_range(capi, cuda):
    """Element-wise error bound with operands spanning six decades: the split keeps 16 significant
    bits of every operand whatever its magnitude (bf16 has the fp32 exponent range), so
    |error| <= ~2^-15 * sum |a||b| per output - checked as 6e-5 * conv(|x|, |w|)."""
    lib, Layout = capi.lib, capi.Layout
    n, h, w, cin, cout, k = 2, 46, 46, 64, 64, 3
    g = torch.Generator().manual_seed(99)
    x = torch.randn(n, cin, h, w, generator=g) * 10.0 ** (torch.rand(n, cin, h, w, generator=g) * 6 - 3)
    wt = torch.randn(cout, cin, k, k, generator=g) * 10.0 ** (torch.rand(cout, cin, k, k, generator=g) * 4 - 3)
    ref = F.conv2d(x.double(), wt.double(), None, padding=1)
    bound = F.conv2d(x.abs().double(), wt.abs().double(), None, padding=1) * 6e-5 + 1e-30
    stream = capi.current_stream()
    lin = Layout.padded(2 * cin, h, w, 1)
    xin = torch.zeros(lib.rtpose_layout_pixels(C.byref(lin), n, h, w) * 2 * cin, device=cuda, dtype=torch.bfloat16)
    xd, wd, bd = x.to(cuda), wt.to(cuda), torch.zeros(cout, device=cuda)
    capi.check(lib.rtpose_nchw_to_layout_split(capi.ptr(xd), capi.ptr(xin), C.byref(lin), cin, cin, n, h, w, stream))
    wp = torch.zeros(lib.rtpose_packed_weight_bytes_bf16x3(cout, cin, k) // 2, device=cuda, dtype=torch.bfloat16)
    bp = torch.zeros(lib.rtpose_packed_bias_floats(cout), device=cuda)
    capi.check(lib.rtpose_pack_conv_weights_bf16x3(capi.ptr(wd), capi.ptr(bd), cout, cin, k, None, cin, capi.ptr(wp),
                                                   capi.ptr(bp), stream))
    lo = Layout.dense(cout, h, w)
    out = torch.zeros(n * h * w * cout, device=cuda)
    d = (capi.ConvDesc * 1)()
    d[0].inp, d[0].w_packed, d[0].bias_packed, d[0].out = xin.data_ptr(), wp.data_ptr(), bp.data_ptr(), out.data_ptr()
    d[0].lin, d[0].lout = lin, lo
    d[0].cin, d[0].cout, d[0].k, d[0].relu, d[0].pool = cin, cout, k, 0, 0
    capi.check(lib.rtpose_conv2d_bf16x3(d, 1, n, h, w, 1, stream))     # fp32 output: no re-split error on top
    got = out.view(n, h, w, cout).permute(0, 3, 1, 2).cpu().double()
    viol = ((got - ref).abs() > bound).sum().item()
    assert viol == 0, "%d outputs outside the 6e-5 * sum|a||b| bound (worst ratio %.3g)" % (
        viol, ((got - ref).abs() / bound).max().item() * 6e-5)
