"""GPU parity of the ShuffleNetV2 x1.0 pose network (csrc/shufflenet.hip + the building-block
kernels in csrc/layout_ops.hip) against the oracle restatement and the golden outputs of the
reference module (imported unmodified through the slim stub; parity unpinned w.r.t. the real slim)."""
import ctypes as C
import ctypes as C_
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import PKG_NAME

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "shufflenet_small.npz")


@pytest.fixture(scope="module")
def net_sd(pkg, cuda):
    from oracle import shufflenet_oracle as so
    sn = importlib.import_module(PKG_NAME + ".shufflenet")
    m = sn.Network(1.0)
    sd = so.seeded_state_dict(m, seed=0)
    m.load_state_dict(sd)
    return m.cuda().eval(), sd


def test_state_dict_keys_match_reference_golden_order(pkg):
    sn = importlib.import_module(PKG_NAME + ".shufflenet")
    m = sn.Network(1.0)
    keys = list(m.state_dict())
    assert keys[0] == "paf.weight" and keys[4] == "network.0.weight" and "network.3.0.conv0.1.0.weight" in keys
    assert keys[-1] == "network.6.1.num_batches_tracked" and len(keys) == 345
    assert sum(p.numel() for p in m.parameters()) == 1312035


def test_forward_matches_reference_golden(net_sd, cuda):
    m, sd = net_sd
    z = np.load(GOLD)
    with torch.no_grad():
        (paf, heat), _ = m(torch.from_numpy(z["x"]).to(cuda))
    assert paf.shape == z["paf"].shape and heat.shape == z["heat"].shape
    assert np.abs(paf.cpu().numpy() - z["paf"]).max() <= 1e-3
    assert np.abs(heat.cpu().numpy() - z["heat"]).max() <= 1e-3


@pytest.mark.parametrize("shape", [(2, 3, 128, 160), (1, 3, 368, 368), (3, 3, 72, 88)])
def test_forward_matches_oracle(net_sd, cuda, shape):
    from oracle import shufflenet_oracle as so
    m, sd = net_sd
    x = torch.rand(shape, generator=torch.Generator().manual_seed(5)) - 0.5
    paf_r, heat_r = so.forward(sd, x)
    with torch.no_grad():
        (paf, heat), _ = m(x.to(cuda))
    assert paf.shape == paf_r.shape
    scale = max(1.0, paf_r.abs().max().item(), heat_r.abs().max().item())
    assert (paf.cpu() - paf_r).abs().max().item() <= 1e-3 * scale
    assert (heat.cpu() - heat_r).abs().max().item() <= 1e-3 * scale


def test_building_blocks(capi, cuda):
    """dw 3x3 (s1/s2), 3x3-s2 ceil max-pool and the stem conv against torch CPU."""
    lib, Layout = capi.lib, capi.Layout
    s = capi.current_stream()
    g = torch.Generator().manual_seed(0)
    n, h, w, c = 2, 23, 30, 24
    x = torch.randn(n, c, h, w, generator=g)
    lin = Layout.padded(c, h, w, 1)
    buf = torch.zeros(lib.rtpose_layout_pixels(C.byref(lin), n, h, w) * c, device=cuda)
    xd = x.to(cuda)
    capi.check(lib.rtpose_nchw_to_layout(capi.ptr(xd), capi.ptr(buf), C.byref(lin), c, c, n, h, w, s))
    for stride in (1, 2):
        wt = torch.randn(c, 1, 3, 3, generator=g)
        b = torch.randn(c, generator=g)
        ref = F.conv2d(x, wt, b, stride, 1, 1, c)
        ho, wo = ref.shape[2], ref.shape[3]
        wp = wt.view(c, 9).t().contiguous().to(cuda)           # [9][C]
        bd = b.to(cuda)
        lout = Layout.dense(c, ho, wo)
        ob = torch.zeros(n * ho * wo * c + 64, device=cuda)
        capi.check(lib.rtpose_dwconv3x3(capi.ptr(buf), C.byref(lin), capi.ptr(wp), capi.ptr(bd), capi.ptr(ob),
                                        C.byref(lout), c, n, h, w, stride, s))
        got = ob[:n * ho * wo * c].view(n, ho, wo, c).permute(0, 3, 1, 2).cpu()
        assert (got - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    # max-pool 3/2 ceil
    ref = F.max_pool2d(x, 3, 2, 0, ceil_mode=True)
    ho, wo = ref.shape[2], ref.shape[3]
    lout = Layout.dense(c, ho, wo)
    ob = torch.zeros(n * ho * wo * c + 64, device=cuda)
    capi.check(lib.rtpose_maxpool3x3s2_ceil(capi.ptr(buf), C.byref(lin), capi.ptr(ob), C.byref(lout), c, n, h, w, s))
    assert torch.equal(ob[:n * ho * wo * c].view(n, ho, wo, c).permute(0, 3, 1, 2).cpu(), ref)
    # stem: affine input + 3x3 s2 conv 3->24 + relu
    x3 = torch.randn(n, 3, h, w, generator=g)
    sc, sh = torch.rand(3, generator=g) + 0.5, torch.randn(3, generator=g)
    wt = torch.randn(24, 3, 3, 3, generator=g) * 0.3
    b = torch.randn(24, generator=g)
    ref = F.relu(F.conv2d(x3 * sc.view(1, 3, 1, 1) + sh.view(1, 3, 1, 1), wt, b, 2, 1))
    l8 = Layout.padded(8, h, w, 1)
    b8 = torch.zeros(lib.rtpose_layout_pixels(C.byref(l8), n, h, w) * 8, device=cuda)
    x3d, scd, shd = x3.to(cuda), sc.to(cuda), sh.to(cuda)
    capi.check(lib.rtpose_nchw_to_layout_affine(capi.ptr(x3d), capi.ptr(b8), C.byref(l8), 3, 8, n, h, w,
                                                capi.ptr(scd), capi.ptr(shd), s))
    wp = torch.zeros(3, 3, 8, 24)
    wp[:, :, :3, :] = wt.permute(2, 3, 1, 0)
    wpd, bd = wp.contiguous().to(cuda), b.to(cuda)
    ho, wo = ref.shape[2], ref.shape[3]
    lout = Layout.dense(24, ho, wo)
    ob = torch.zeros(n * ho * wo * 24 + 64, device=cuda)
    capi.check(lib.rtpose_stem_conv3x3_s2(capi.ptr(b8), C.byref(l8), capi.ptr(wpd), capi.ptr(bd), capi.ptr(ob),
                                          C.byref(lout), 8, 24, n, h, w, 1, s))
    got = ob[:n * ho * wo * 24].view(n, ho, wo, 24).permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


def test_decoder_reads_shufflenet_output_in_place(net_sd, capi, cuda):
    """The pose decoder consumes the heads' output buffer where they wrote it."""
    dec = importlib.import_module(PKG_NAME + ".decode")
    from oracle import post_oracle as po
    m, _ = net_sd
    x = (torch.rand(2, 3, 184, 184, generator=torch.Generator().manual_seed(7)) - 0.5).to(cuda)
    with torch.no_grad():
        (paf, heat), _ = m(x)
    plan = m.plan_for(x)
    lib = capi.lib
    base, lay, c, h, w = C.c_void_p(), capi.Layout(), C.c_int(), C.c_int(), C.c_int()
    capi.check(lib.rtpose_shufflenet_output_view(plan.handle, 1, C.byref(base), C.byref(lay), C.byref(c), C.byref(h), C.byref(w)))
    pbase, play = C.c_void_p(), capi.Layout()
    capi.check(lib.rtpose_shufflenet_output_view(plan.handle, 0, C.byref(pbase), C.byref(play), None, None, None))
    cfg = dec.make_cfg(None, 64, 64)
    while True:
        bufs = dec.DecodeBuffers(cfg, 2, cuda)
        dec.decode_enqueue(base, lay, pbase, play, 2, h.value, w.value, bufs)
        recs = dec.fetch(bufs)
        if not int(np.bitwise_or.reduce(recs[:, 2])):
            break
        cfg = dec.make_cfg(None, min(cfg.max_peaks_per_part * 2, 1024), min(cfg.max_humans * 2, 16384))
    for i in range(2):
        hm = heat[i].permute(1, 2, 0).contiguous().cpu().numpy()
        pf = paf[i].permute(1, 2, 0).contiguous().cpu().numpy()
        jl, r = po.paf_to_pose(hm, pf)
        out = dec.parse_image(recs[i], cfg)
        assert np.array_equal(out["peaks"], jl) and np.array_equal(out["parts"], r["parts"])


@pytest.mark.parametrize("shape", [(2, 3, 128, 160), (1, 3, 368, 368)])
def test_bf16_plan_matches_emulation_and_fp32(net_sd, cuda, shape):
    """BASELINE config 4 "fp32 and bf16": the bf16 plan against its emulation oracle (2e-2 of the map
    scale: 1-ulp rounding flips propagate through 56 layers) and against the fp32 oracle (6e-2)."""
    from oracle import shufflenet_oracle as so
    m, sd = net_sd
    g = torch.Generator().manual_seed(3)
    x = torch.rand(shape, generator=g) - 0.5
    paf_e, heat_e = so.forward_bf16_emulated(sd, x)
    paf_r, heat_r = so.forward(sd, x)
    m.set_compute_dtype('bf16')
    try:
        with torch.no_grad():
            (paf, heat), _ = m(x.to(cuda))
            (paf2, heat2), _ = m(x.to(cuda))
    finally:
        m.set_compute_dtype('fp32')
    assert torch.equal(paf, paf2) and torch.equal(heat, heat2)
    for a, e, r in ((paf, paf_e, paf_r), (heat, heat_e, heat_r)):
        scale = max(1.0, r.abs().max().item())
        assert (a.cpu() - e).abs().max().item() <= 2e-2 * scale, "vs bf16 emulation"
        assert (a.cpu() - r).abs().max().item() <= 6e-2 * scale, "vs fp32 oracle"
    with torch.no_grad():
        (paf3, _), _ = m(x.to(cuda))         # back on the fp32 plan
    assert (paf3.cpu() - paf_r).abs().max().item() <= 1e-3 * max(1.0, paf_r.abs().max().item())


def test_full_size_batch128_config4(net_sd, cuda):
    """BASELINE.json configs[3] at its full shape (128 x 3 x 368 x 368): deterministic, every image
    independent of its position / neighbours in the batch (bit-exact under a permutation, and equal
    to the same image run in a batch of 2), and images spread over the batch equal the CPU oracle."""
    from oracle import shufflenet_oracle as so
    m, sd = net_sd
    g = torch.Generator().manual_seed(21)
    x = torch.rand(128, 3, 368, 368, generator=g) - 0.5
    perm = torch.randperm(128, generator=g)
    with torch.no_grad():
        (paf, heat), _ = m(x.to(cuda))
        (paf2, heat2), _ = m(x.to(cuda))
        (paf_p, heat_p), _ = m(x[perm].to(cuda))
        (paf_s, heat_s), _ = m(x[[7, 100]].to(cuda))
    assert paf.shape == (128, 38, 46, 46) and heat.shape == (128, 19, 46, 46)
    assert torch.equal(paf, paf2) and torch.equal(heat, heat2)
    assert torch.equal(paf_p, paf[perm.to(cuda)]) and torch.equal(heat_p, heat[perm.to(cuda)])
    assert torch.equal(paf_s, paf[[7, 100]]) and torch.equal(heat_s, heat[[7, 100]])
    idx = [0, 31, 64, 77, 127, 5]
    paf_r, heat_r = so.forward(sd, x[idx])
    scale = max(1.0, paf_r.abs().max().item(), heat_r.abs().max().item())
    assert (paf[idx].cpu() - paf_r).abs().max().item() <= 1e-3 * scale
    assert (heat[idx].cpu() - heat_r).abs().max().item() <= 1e-3 * scale


@pytest.mark.parametrize("shape", [(2, 3, 368, 368), (3, 3, 97, 131), (1, 3, 64, 40)])
def test_stem_and_pool_in_one_launch(capi, cuda, shape):
    """rtpose_stem_pool_nchw == BatchNorm2d(3) (as scale / shift) -> conv 3x3 s2 p1 (+bias, BN folded) -> ReLU ->
    MaxPool2d(3, 2, 0, ceil_mode=True) of plain torch (rtpose_shufflenetV2.py:96-99), fp32 and bf16 output;
    odd sizes exercise the ceil-mode windows that hang over the conv map and partial tiles."""
    g = torch.Generator().manual_seed(shape[2])
    n, _, H, W = shape
    x = torch.rand(shape, generator=g) - 0.5
    scale, shift = torch.rand(3, generator=g) + 0.5, torch.randn(3, generator=g) * 0.1
    w = torch.randn(24, 3, 3, 3, generator=g) * 0.3
    b = torch.randn(24, generator=g) * 0.1
    xa = x * scale.view(1, 3, 1, 1) + shift.view(1, 3, 1, 1)
    ref = F.max_pool2d(F.relu(F.conv2d(xa, w, b, stride=2, padding=1)), 3, 2, 0, ceil_mode=True)
    H2, W2 = ref.shape[2], ref.shape[3]
    wp = torch.zeros(3, 3, 8, 24)
    wp[:, :, :3, :] = w.permute(2, 3, 1, 0)                 # packed [ky][kx][cin_pad 8][cout]
    d = [t.contiguous().to(cuda) for t in (x, scale, shift, wp, b)]
    for bf16, C in ((0, 24), (1, 32)):
        lay = capi.Layout.padded(C, H2, W2, 1)
        npix = capi.lib.rtpose_layout_pixels(C_.byref(lay), n, H2, W2)
        out = torch.zeros(npix * C, dtype=torch.int16 if bf16 else torch.float32, device=cuda)
        capi.check(capi.lib.rtpose_stem_pool_nchw(capi.ptr(d[0]), capi.ptr(d[1]), capi.ptr(d[2]), capi.ptr(d[3]),
                                                  capi.ptr(d[4]), capi.ptr(out), C_.byref(lay), 24, n, H, W, bf16,
                                                  capi.current_stream()), "rtpose_stem_pool_nchw")
        if bf16:
            f = torch.zeros(npix * C, device=cuda)
            capi.check(capi.lib.rtpose_layout_bf16_to_f32(capi.ptr(out), C_.byref(lay), capi.ptr(f), C_.byref(lay), C, n,
                                                          H2, W2, capi.current_stream()))
            out = f
        got = torch.empty(n, C, H2, W2, device=cuda)
        capi.check(capi.lib.rtpose_layout_to_nchw(capi.ptr(out), C_.byref(lay), capi.ptr(got), C, n, H2, W2,
                                                  capi.current_stream()))
        got = got.cpu()
        assert got[:, 24:].abs().max().item() == 0.0 if C > 24 else True
        tol = 1e-2 if bf16 else 1e-5
        assert (got[:, :24] - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()), (bf16, shape)
        assert abs(out.abs().sum().item() - got.abs().sum().item()) <= 1e-3 * got.abs().sum().item()   # gaps untouched
