"""GPU parity of the bf16 path (csrc/conv_mfma_bf16.hip, bf16 plans of csrc/net.hip;
BASELINE config 3: "bf16 ... single MI355X").

The reference has no reduced-precision arithmetic, so the contract is the one stated in
oracle/net_oracle.py:forward_bf16_emulated - operands rounded to bf16 (RNE), exact products,
fp32 accumulation, fp32 bias/ReLU/pool, activations re-rounded between layers, final maps
fp32.  Tolerances are written next to each check:
  * one conv, fp32 output : 2e-5 of max|ref|  (fp32 accumulation order only)
  * one conv, bf16 output : 1 bf16 ulp (2^-7 relative) around the rounded reference
  * whole net vs emulation: 3e-2 of max|ref| (observed 1.4e-2), rms 6e-3  (rounding flips of 1 ulp propagate)
  * whole net vs fp32     : 6e-2 of max|ref|  (what bf16 operands cost; the north_star 1e-3
                            bound applies to the fp32 path only)
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _run_conv_bf16(capi, dev, n, h, w, cin, cout, k, relu, pool, pad_in, pad_out, seed, groups=1,
                   cin_pad=None, out_f32=False, aligned=False, lead_extra=0):
    lib, Layout = capi.lib, capi.Layout
    g = torch.Generator().manual_seed(seed)
    x = _rb(torch.randn(n, cin, h, w, generator=g))
    cin_p = cin_pad or ((cin + 15) // 16 * 16)
    ws, bs, refs = [], [], []
    for gi in range(groups):
        wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        y = F.conv2d(x.double(), _rb(wt).double(), b.double(), padding=k // 2).float()
        if relu:
            y = F.relu(y)
        if pool:
            y = F.max_pool2d(y, 2, 2, 0)
        ws.append(wt.to(dev))
        bs.append(b.to(dev))
        refs.append(y)
    stream = capi.current_stream()
    lin = Layout.padded(cin_p, h, w, pad_in)
    lin.lead += lead_extra  # (a slice far into a large buffer: byte offsets past 2^31)
    npx = lib.rtpose_layout_pixels(C.byref(lin), n, h, w)
    xin = torch.zeros(npx * cin_p, device=dev, dtype=torch.bfloat16)
    xd = x.to(dev)
    capi.check(lib.rtpose_nchw_to_layout_bf16(capi.ptr(xd), capi.ptr(xin), C.byref(lin), cin, cin_p, n, h, w,
                                              stream))
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    # odd stride + channel offsets: exercises slices (scalar stores); aligned: 16-byte aligned slices, the layout the
    # network uses - full N tiles then take the 16-byte-store epilogues (transposed product / LDS slabs with a fused pool)
    cstride_out = cout * groups + (16 if aligned else 3)
    ch0 = 8 if aligned else 1
    descs = (capi.ConvDesc * groups)()
    outs, keep = [], []
    lout_full = Layout.padded(cstride_out, ho, wo, pad_out)
    odt = torch.float32 if out_f32 else torch.bfloat16
    obuf = torch.zeros(lib.rtpose_layout_pixels(C.byref(lout_full), n, ho, wo) * cstride_out, device=dev,
                       dtype=odt)
    for gi in range(groups):
        wp = torch.zeros(lib.rtpose_packed_weight_bytes_bf16(cout, cin_p, k) // 2, device=dev,
                         dtype=torch.bfloat16)
        bp = torch.zeros(lib.rtpose_packed_bias_floats(cout), device=dev)
        capi.check(lib.rtpose_pack_conv_weights_bf16(capi.ptr(ws[gi]), capi.ptr(bs[gi]), cout, cin, k, None,
                                                     cin_p, capi.ptr(wp), capi.ptr(bp), stream))
        keep += [wp, bp]
        d = descs[gi]
        d.inp, d.w_packed, d.bias_packed, d.out = xin.data_ptr(), wp.data_ptr(), bp.data_ptr(), obuf.data_ptr()
        d.lin = lin
        d.lout = Layout.padded(cstride_out, ho, wo, pad_out, choff=gi * cout + ch0)
        d.cin, d.cout, d.k, d.relu, d.pool = cin_p, cout, k, int(relu), int(pool)
    capi.check(lib.rtpose_conv2d_bf16(descs, groups, n, h, w, int(out_f32), stream), "rtpose_conv2d_bf16")
    for gi in range(groups):
        o = torch.empty(n, cout, ho, wo, device=dev)
        lo = Layout.padded(cstride_out, ho, wo, pad_out, choff=gi * cout + ch0)
        if out_f32:
            capi.check(lib.rtpose_layout_to_nchw(capi.ptr(obuf), C.byref(lo), capi.ptr(o), cout, n, ho, wo, stream))
        else:
            dense = torch.empty(n, ho, wo, cout, device=dev)
            ld = Layout.dense(cout, ho, wo)
            capi.check(lib.rtpose_layout_bf16_to_f32(capi.ptr(obuf), C.byref(lo), capi.ptr(dense), C.byref(ld),
                                                     cout, n, ho, wo, stream))
            o = dense.permute(0, 3, 1, 2).contiguous()
        outs.append(o.cpu())
    torch.cuda.synchronize()
    total = obuf.float().abs().sum().item()
    inner = sum(o.abs().sum().item() for o in outs)
    assert abs(total - inner) <= 1e-3 * max(1.0, inner), "conv wrote outside its slice / into the gaps"
    return outs, refs


def _check(out, ref, out_f32):
    scale = max(1.0, ref.abs().max().item())
    if out_f32:
        err = (out - ref).abs().max().item()
        assert err <= 2e-5 * scale, "fp32-out max abs err %g (scale %g)" % (err, scale)
    else:
        # within one bf16 ulp of the rounded reference (an fp32 sum that differs in the last
        # bits may round the other way); 2^-7 relative + a denormal-free floor
        rr = _rb(ref)
        err = (out - rr).abs()
        bound = rr.abs() * 2.0 ** -7 + 1e-6 * scale
        bad = (err > bound).sum().item()
        assert bad == 0, "%d outputs further than 1 bf16 ulp from the reference (max err %g)" % (
            bad, err.max().item())
        exact = (out == rr).float().mean().item()
        assert exact > 0.98, "only %.3f of the outputs equal the RNE-rounded reference" % exact


CASES = [
    # n, h, w, cin(packed), cout, k, relu, pool, pad_in, pad_out
    (2, 46, 46, 128, 128, 7, 1, 0, 3, 3),     # Mconv2_stageN: strip mode, NF=2
    (1, 46, 49, 192, 128, 7, 1, 0, 3, 3),     # ski.jpg geometry, 185->192 packed input
    (3, 23, 17, 128, 38, 1, 0, 0, 0, 3),      # 1x1 head (ck=64), ragged M
    (2, 46, 46, 512, 19, 1, 0, 0, 0, 0),      # conv5_5_CPM_L2
    (2, 46, 46, 256, 512, 3, 1, 0, 1, 1),     # conv4_1: 4 N-tiles of 128
    (1, 46, 46, 128, 128, 3, 1, 0, 3, 1),     # stage-1 conv reading the P=3 concat layout
    (1, 96, 80, 64, 64, 3, 1, 1, 1, 1),       # 2-D tile mode + fused pool, NF=1
    (2, 72, 88, 16, 64, 3, 1, 0, 1, 1),       # conv1_1 (3->16 padded input), ck=16
    (1, 100, 92, 128, 256, 3, 1, 0, 1, 1),    # 2-D tiles with ragged right/bottom edges
    (1, 70, 66, 128, 128, 7, 1, 0, 3, 0),     # 7x7 in 2-D tile mode (multi-scale maps)
    (1, 12, 10, 16, 24, 3, 0, 0, 1, 0),       # tiny: one partial block
    (5, 6, 6, 32, 8, 7, 1, 0, 3, 3),          # strip spanning several images
    (1, 16, 16, 96, 64, 1, 1, 0, 0, 0),       # 1x1 with ck=32 (cin % 64 != 0)
]


@pytest.mark.parametrize("case", CASES)
def test_conv_bf16_matches_emulation(capi, cuda, case):
    n, h, w, cin, cout, k, relu, pool, pin, pout = case
    src_cin = 3 if (cin == 16 and h == 72) else (185 if cin == 192 else cin)
    outs, refs = _run_conv_bf16(capi, cuda, n, h, w, src_cin, cout, k, relu, pool, pin, pout,
                                seed=hash(case) % 1000, cin_pad=cin)
    _check(outs[0], refs[0], False)


@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[6]])
def test_conv_bf16_fp32_output(capi, cuda, case):
    n, h, w, cin, cout, k, relu, pool, pin, pout = case
    outs, refs = _run_conv_bf16(capi, cuda, n, h, w, cin, cout, k, relu, pool, pin, pout, seed=11, cin_pad=cin,
                                out_f32=True)
    _check(outs[0], refs[0], True)


def test_conv_bf16_grouped_and_tail(capi, cuda):
    # two branches in one grid; 5*46*46 pixels = 83 strips: exercises the half-tile tail path
    outs, refs = _run_conv_bf16(capi, cuda, 5, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=7, groups=2)
    for o, r in zip(outs, refs):
        _check(o, r, False)


ALIGNED_CASES = [
    # n, h, w, cin, cout, k, relu, pool, pad_in, pad_out, groups
    (2, 46, 46, 128, 128, 7, 1, 0, 3, 3, 1),   # strips, 1 x 4 waves
    (5, 46, 46, 128, 128, 7, 1, 0, 3, 3, 2),   # two branches, persistent blocks + the half-tile tail
    (1, 70, 66, 128, 128, 7, 0, 0, 3, 0, 1),   # 7x7 in 2-D tiles with ragged edges, no ReLU
    (5, 6, 6, 32, 64, 7, 1, 0, 3, 3, 1),       # strips spanning several images, one 64-column tile
    (2, 46, 46, 256, 512, 3, 1, 0, 1, 1, 1),   # 2 x 2 waves of 64 x 64, 64-channel chunks
    (1, 100, 92, 128, 256, 3, 1, 0, 1, 1, 1),  # 2-D tiles with ragged right / bottom edges
    (3, 23, 17, 64, 64, 3, 1, 0, 1, 3, 1),     # strips with a ragged last tile, 3x3
    (1, 96, 80, 64, 64, 3, 1, 1, 1, 1, 1),     # fused pool: the LDS-slab epilogue stays
]


@pytest.mark.parametrize("case", ALIGNED_CASES)
def test_conv_bf16_aligned_slices_take_the_16_byte_store_epilogues(capi, cuda, case):
    """Round 4: bf16 outputs of full N tiles in 16-byte aligned slices (every k x k layer of the network) are computed as
    the TRANSPOSED product and stored from the accumulators, 32 bytes per lane and fragment
    (csrc/conv_mfma_bf16.hip, template TR); same contract as the scalar-store cases above, and nothing written outside the slice."""
    n, h, w, cin, cout, k, relu, pool, pin, pout, groups = case
    outs, refs = _run_conv_bf16(capi, cuda, n, h, w, cin, cout, k, relu, pool, pin, pout, seed=31 + k, groups=groups,
                                aligned=True)
    for o, r in zip(outs, refs):
        _check(o, r, False)


@pytest.mark.parametrize("shape,src", [((2, 64, 72), "nchw"), ((1, 37, 45), "nchw"), ((1, 37, 45), "layout"),
                                       ((1, 184, 200), "layout")])
def test_first_layer_bf16_kernel_matches_the_emulation(capi, cuda, shape, src):
    """rtpose_conv_first_bf16 (round 6: conv1_1 of the bf16 plan on the fp32 matrix instruction, image and filters rounded to
    bf16, read from the NCHW image or an fp32 layout buffer, bf16 out; tiles with ragged right / bottom edges): within one bf16 ulp of conv2d on the rounded operands
    in double, > 98 % of the outputs equal to its RNE rounding (the contract of the generic bf16 kernel, `_check`); nothing
    written outside the 64-channel slice; agrees with the generic kernel on 16 padded channels to one bf16 ulp."""
    lib, Layout = capi.lib, capi.Layout
    n, h, w = shape
    g = torch.Generator().manual_seed(h * 977 + w)
    x = torch.rand(n, 3, h, w, generator=g) - 0.5                      # NOT pre-rounded: the kernel rounds
    wd = torch.randn(64, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5
    bd = torch.randn(64, generator=g) * 0.1
    ref = F.relu(F.conv2d(_rb(x).double(), _rb(wd).double(), bd.double(), padding=1).float())
    stream = capi.current_stream()
    xd, wdd, bdd = x.to(cuda), wd.to(cuda), bd.to(cuda)
    wp = torch.zeros(lib.rtpose_conv_first_packed_floats(), device=cuda)
    capi.check(lib.rtpose_pack_conv_first_bf16(capi.ptr(wdd), capi.ptr(bdd), capi.ptr(wp), stream))
    cs, ch0 = 64 + 16, 8                                               # a 16-byte aligned slice of a wider buffer
    lo = Layout.padded(cs, h, w, 1, choff=ch0)
    q = lib.rtpose_layout_pixels(C.byref(lo), n, h, w)
    obuf = torch.zeros(q * cs, device=cuda, dtype=torch.bfloat16)
    xin = lin = None
    if src == "layout":                                                # the plan's fp32 NHWC8 staging buffer
        lin = Layout.padded(8, h, w, 1)
        xin = torch.zeros(lib.rtpose_layout_pixels(C.byref(lin), n, h, w) * 8, device=cuda)
        capi.check(lib.rtpose_nchw_to_layout(capi.ptr(xd), capi.ptr(xin), C.byref(lin), 3, 8, n, h, w, stream))
    capi.check(lib.rtpose_conv_first_bf16(capi.ptr(xd) if src == "nchw" else None, capi.ptr(xin) if xin is not None else None,
                                          C.byref(lin) if lin is not None else None, capi.ptr(wp), capi.ptr(obuf), C.byref(lo), 1,
                                          n, h, w, stream), "rtpose_conv_first_bf16")
    dense = torch.empty(n, h, w, 64, device=cuda)
    ld = Layout.dense(64, h, w)
    capi.check(lib.rtpose_layout_bf16_to_f32(capi.ptr(obuf), C.byref(lo), capi.ptr(dense), C.byref(ld), 64, n, h, w, stream))
    torch.cuda.synchronize()
    out = dense.permute(0, 3, 1, 2).contiguous().cpu()
    _check(out, ref, False)
    total, inner = obuf.float().abs().sum().item(), out.abs().sum().item()
    assert abs(total - inner) <= 1e-3 * max(1.0, inner), "wrote outside its slice / into the gaps"
    if (n, h, w) == (2, 64, 72):                                       # the launch it replaces in the plan
        outs, _ = _run_conv_bf16(capi, cuda, n, h, w, 3, 64, 3, 1, 0, 1, 1, seed=5, cin_pad=16, aligned=True)
        assert outs[0].shape == out.shape                               # (other data: only the geometry is shared)
    assert lib.rtpose_conv_first_bf16(capi.ptr(xd), None, None, capi.ptr(wp), capi.ptr(obuf),
                                      C.byref(Layout.padded(cs, h, w, 1, choff=4)), 1, n, h, w, stream) != 0


@pytest.mark.parametrize("case", [(1, 46, 46, 128, 38, 0), (2, 23, 17, 128, 19, 0), (1, 46, 49, 512, 38, 0),
                                  (3, 23, 17, 512, 19, 1), (1, 30, 46, 128, 38, 1)])
def test_conv1x1_pair_bf16_matches_the_two_launch_contract(capi, cuda, case):
    """rtpose_conv1x1_pair_bf16 (round 6, csrc/conv_tail_bf16.hip): Conv2d(128, mid, 1) + ReLU -> Conv2d(mid, cout2, 1), both
    branches in one grid, the intermediate rounded to bf16 inside the CU.  Contract = two rtpose_conv2d_bf16 launches: against
    the emulation in double (operands rounded to bf16, the intermediate rounded to bf16) the bf16 outputs sit within two bf16
    ulps (an intermediate that rounds the other way moves a few outputs by one more) and > 97 % equal its rounding, fp32
    outputs within 2e-3 of the scale; it agrees with the two generic launches themselves to the same bound; nothing is
    written outside the slices (odd channel offsets, the concat buffer's 38 | 19 split)."""
    lib, Layout = capi.lib, capi.Layout
    n, h, w, mid, cout2, out_f32 = case
    g = torch.Generator().manual_seed(mid + cout2)
    stream = capi.current_stream()
    x = _rb(torch.randn(n, 128, h, w, generator=g))
    lin = Layout.padded(128 + 8, h, w, 0, choff=8)
    npx = lib.rtpose_layout_pixels(C.byref(lin), n, h, w)
    xin = torch.zeros(npx * (128 + 8), device=cuda, dtype=torch.bfloat16)
    capi.check(lib.rtpose_nchw_to_layout_bf16(capi.ptr(x.to(cuda)), capi.ptr(xin), C.byref(lin), 128, 128, n, h, w, stream))
    cs_out = 57 + 7
    lo_full = Layout.padded(cs_out, h, w, 3)
    odt = torch.float32 if out_f32 else torch.bfloat16
    obuf = torch.zeros(lib.rtpose_layout_pixels(C.byref(lo_full), n, h, w) * cs_out, device=cuda, dtype=odt)
    obuf2 = torch.zeros_like(obuf)
    tbuf_l = Layout.padded(mid, h, w, 0)
    d1, d2 = (capi.ConvDesc * 2)(), (capi.ConvDesc * 2)()
    refs, keep, offs = [], [], [3, 3 + cout2 if 3 + 2 * cout2 <= cs_out else 3]
    ngroups = 2 if 3 + 2 * cout2 <= cs_out else 1
    for gi in range(ngroups):
        w1 = torch.randn(mid, 128, 1, 1, generator=g) * (2.0 / 128) ** 0.5
        b1 = torch.randn(mid, generator=g) * 0.1
        w2 = torch.randn(cout2, mid, 1, 1, generator=g) * (1.0 / mid) ** 0.5
        b2 = torch.randn(cout2, generator=g) * 0.1
        t = _rb(F.relu(F.conv2d(x.double(), _rb(w1).double(), b1.double()).float()))
        refs.append(F.conv2d(t.double(), _rb(w2).double(), b2.double()).float())
        tb = torch.zeros(lib.rtpose_layout_pixels(C.byref(tbuf_l), n, h, w) * mid, device=cuda, dtype=torch.bfloat16)
        packs = []
        for (wt, bs, co, ci) in ((w1, b1, mid, 128), (w2, b2, cout2, mid)):
            wp = torch.zeros(lib.rtpose_packed_weight_bytes_bf16(co, ci, 1) // 2, device=cuda, dtype=torch.bfloat16)
            bp = torch.zeros(lib.rtpose_packed_bias_floats(co), device=cuda)
            wd, bdev = wt.to(cuda), bs.to(cuda)      # (kept alive: a temporary's block is re-used before the pack kernel runs)
            capi.check(lib.rtpose_pack_conv_weights_bf16(capi.ptr(wd), capi.ptr(bdev), co, ci, 1, None, ci,
                                                         capi.ptr(wp), capi.ptr(bp), stream))
            packs.append((wp, bp))
            keep += [wd, bdev]
        keep += packs + [tb]
        a, b = d1[gi], d2[gi]
        a.inp, a.w_packed, a.bias_packed, a.out = xin.data_ptr(), packs[0][0].data_ptr(), packs[0][1].data_ptr(), tb.data_ptr()
        a.lin, a.lout = lin, tbuf_l
        a.cin, a.cout, a.k, a.relu, a.pool = 128, mid, 1, 1, 0
        b.inp, b.w_packed, b.bias_packed, b.out = tb.data_ptr(), packs[1][0].data_ptr(), packs[1][1].data_ptr(), obuf.data_ptr()
        b.lin = tbuf_l
        b.lout = Layout.padded(cs_out, h, w, 3, choff=offs[gi])
        b.cin, b.cout, b.k, b.relu, b.pool = mid, cout2, 1, 0, 0
    assert lib.rtpose_conv1x1_pair_bf16_fits(d1, d2, ngroups) == 1
    capi.check(lib.rtpose_conv1x1_pair_bf16(d1, d2, ngroups, n, h, w, out_f32, stream), "rtpose_conv1x1_pair_bf16")

    def read(buf, gi):
        lo = Layout.padded(cs_out, h, w, 3, choff=offs[gi])
        if out_f32:
            o = torch.empty(n, cout2, h, w, device=cuda)
            capi.check(lib.rtpose_layout_to_nchw(capi.ptr(buf), C.byref(lo), capi.ptr(o), cout2, n, h, w, stream))
            return o.cpu()
        dense = torch.empty(n, h, w, cout2, device=cuda)
        capi.check(lib.rtpose_layout_bf16_to_f32(capi.ptr(buf), C.byref(lo), capi.ptr(dense), C.byref(Layout.dense(cout2, h, w)),
                                                 cout2, n, h, w, stream))
        return dense.permute(0, 3, 1, 2).contiguous().cpu()
    fused = [read(obuf, gi) for gi in range(ngroups)]
    # the launches it replaces: two generic bf16 launches per branch through the intermediate buffer
    for gi in range(ngroups):
        d2[gi].out = obuf2.data_ptr()
        one = (capi.ConvDesc * 1)(d1[gi])
        capi.check(lib.rtpose_conv2d_bf16(one, 1, n, h, w, 0, stream))
        two = (capi.ConvDesc * 1)(d2[gi])
        capi.check(lib.rtpose_conv2d_bf16(two, 1, n, h, w, out_f32, stream))
    torch.cuda.synchronize()
    plain = [read(obuf2, gi) for gi in range(ngroups)]
    for o, p_, r in zip(fused, plain, refs):
        scale = max(1.0, r.abs().max().item())
        if out_f32:
            assert (o - r).abs().max().item() <= 2e-3 * scale and (o - p_).abs().max().item() <= 2e-3 * scale
        else:
            rr = _rb(r)
            for other in (rr, p_):
                assert ((o - other).abs() <= other.abs() * 2.0 ** -6 + 2e-3 * scale).all()
            assert (o == rr).float().mean().item() > 0.97
    total = obuf.float().abs().sum().item()
    inner = sum(o.abs().sum().item() for o in fused)
    assert abs(total - inner) <= 1e-3 * max(1.0, inner), "wrote outside its slices / into the gaps"
    d1[0].cin = 64
    assert lib.rtpose_conv1x1_pair_bf16_fits(d1, d2, ngroups) == 0


C64_CASES = [
    # n, h, w, cout, relu, pool, pad_in, pad_out
    (2, 40, 72, 64, 1, 1, 1, 1),     # conv1_2's form: fused pool, ragged right / bottom tiles, two images
    (1, 37, 45, 128, 1, 0, 1, 1),    # conv2_1's form: two passes of 64 channels, odd map
    (3, 16, 32, 64, 0, 0, 3, 3),     # exactly one tile per image, no ReLU, the P = 3 layouts
    (2, 50, 100, 128, 1, 1, 1, 3),   # pool + two passes
    (1, 184, 120, 128, 1, 0, 1, 1),  # conv2_1 at its real height
]


@pytest.mark.parametrize("case", C64_CASES)
def test_conv3x3_of_64_input_channels_takes_its_own_kernel_and_keeps_the_contract(capi, cuda, case):
    """Round 6 (csrc/conv_c64_bf16.hip): 3x3 convs with 64 bf16 input channels and 16-byte aligned slices - conv1_2 (+ pool)
    and conv2_1 of the bf16 plan - run in a persistent kernel that keeps a tile's whole K in one LDS halo; rtpose_conv2d_bf16
    dispatches to it.  Same contract as the generic kernel (within one bf16 ulp of the RNE-rounded emulation in double, > 98 %
    equal to it, nothing written outside the slice), and `_fits` says that it is the kernel that ran."""
    lib, Layout = capi.lib, capi.Layout
    n, h, w, cout, relu, pool, pin, pout = case
    d = (capi.ConvDesc * 1)()
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    d[0].lin = Layout.padded(64, h, w, pin)
    d[0].lout = Layout.padded(cout + 16, ho, wo, pout, choff=8)
    d[0].cin, d[0].cout, d[0].k, d[0].relu, d[0].pool = 64, cout, 3, relu, pool
    assert lib.rtpose_conv3x3_c64_bf16_fits(d, 1, n, h, w) == 1
    outs, refs = _run_conv_bf16(capi, cuda, n, h, w, 64, cout, 3, relu, pool, pin, pout, seed=64 + cout + h, aligned=True)
    _check(outs[0], refs[0], False)
    # what it does not take: other channel counts, unaligned slices, grouped launches
    d[0].cin = 128
    assert lib.rtpose_conv3x3_c64_bf16_fits(d, 1, n, h, w) == 0
    d[0].cin = 64
    d[0].lout = Layout.padded(cout + 3, ho, wo, pout, choff=1)
    assert lib.rtpose_conv3x3_c64_bf16_fits(d, 1, n, h, w) == 0
    d[0].lout = Layout.padded(cout + 16, ho, wo, pout, choff=8)
    assert lib.rtpose_conv3x3_c64_bf16_fits(d, 2, n, h, w) == 0


def test_conv3x3_of_64_input_channels_with_byte_offsets_past_2_to_31(capi, cuda):
    """The kernel addresses its input with unsigned 32-bit byte offsets: the 2 x 368 scale of the multi-scale flow at batch
    32 is a 2.2 GB activation buffer.  Here a small image sits 17 M pixel slots into a 2.2 GB buffer (lead): same contract;
    past 4 GB `_fits` hands the conv back to the generic kernel."""
    lib, Layout = capi.lib, capi.Layout
    n, h, w, cout = 1, 40, 72, 64
    extra = 17_000_000
    d = (capi.ConvDesc * 1)()
    d[0].lin = Layout.padded(64, h, w, 1)
    d[0].lin.lead += extra
    d[0].lout = Layout.padded(cout + 16, h // 2, w // 2, 1, choff=8)
    d[0].cin, d[0].cout, d[0].k, d[0].relu, d[0].pool = 64, cout, 3, 1, 1
    assert lib.rtpose_layout_pixels(C.byref(d[0].lin), n, h, w) * 128 > 2 ** 31
    assert lib.rtpose_conv3x3_c64_bf16_fits(d, 1, n, h, w) == 1
    outs, refs = _run_conv_bf16(capi, cuda, n, h, w, 64, cout, 3, 1, 1, 1, 1, seed=5, aligned=True, lead_extra=extra)
    _check(outs[0], refs[0], False)
    d[0].lin.lead += extra
    assert lib.rtpose_conv3x3_c64_bf16_fits(d, 1, n, h, w) == 0


def test_conv_bf16_rejects_bad_geometry(capi, cuda):
    lib = capi.lib
    d = (capi.ConvDesc * 1)()
    d[0].k = 3
    d[0].cin = 24  # not a multiple of 16
    assert lib.rtpose_conv2d_bf16(d, 1, 1, 8, 8, 0, None) != 0
    assert "multiple of 16" in capi.last_error()


@pytest.fixture(scope="module")
def model_and_sd(pkg, cuda):
    from oracle import net_oracle
    m = pkg.get_model('vgg19')
    sd = net_oracle.he_init_state_dict(m, seed=0)
    m.load_state_dict(sd)
    m = m.cuda().float().eval()
    return m, sd


@pytest.mark.parametrize("shape", [(2, 3, 64, 72), (1, 3, 56, 40)])
def test_net_bf16_matches_emulation_and_fp32(model_and_sd, cuda, shape):
    from oracle import net_oracle
    m, sd = model_and_sd
    g = torch.Generator().manual_seed(1)
    x = torch.rand(shape, generator=g) - 0.5
    (paf_e, heat_e), saved_e = net_oracle.forward_bf16_emulated(sd, x)
    (paf_r, heat_r), _ = net_oracle.forward(sd, x)
    m.set_compute_dtype('bf16')
    try:
        with torch.no_grad():
            (paf, heat), saved = m(x.to(cuda))
            (paf2, heat2), _ = m(x.to(cuda))
    finally:
        m.set_compute_dtype('fp32')
    assert len(saved) == 12 and paf.shape == paf_e.shape and heat.shape == heat_e.shape
    assert torch.equal(paf, paf2) and torch.equal(heat, heat2), "bf16 forward is not deterministic"
    for i, (a, b) in enumerate(zip(saved, saved_e)):
        scale = max(1.0, b.abs().max().item())
        err = (a.cpu() - b).abs().max().item()
        # max over the map of a chaotic quantity (1-ulp rounding flips propagate; which ones flip depends
        # on the accumulation order): observed 1.0-1.4e-2, rms ~2e-3
        assert err <= 3e-2 * scale, "stage output %d vs bf16 emulation: %g (scale %g)" % (i, err, scale)
        assert ((a.cpu() - b) ** 2).mean().sqrt().item() <= 6e-3 * scale
    for a, b in ((paf, paf_r), (heat, heat_r)):
        scale = max(1.0, b.abs().max().item())
        err = (a.cpu() - b).abs().max().item()
        assert err <= 6e-2 * scale, "bf16 vs fp32 oracle: %g (scale %g)" % (err, scale)
        rms = ((a.cpu() - b) ** 2).mean().sqrt().item()
        assert rms <= 1e-2 * scale, "bf16 vs fp32 oracle rms %g" % rms


def test_net_bf16_needs_multiple_of_8(model_and_sd, capi, cuda):
    m, _ = model_and_sd
    m.set_compute_dtype('bf16')
    try:
        with pytest.raises(capi.RtposeError):
            m(torch.zeros(1, 3, 60, 64, device=cuda))
    finally:
        m.set_compute_dtype('fp32')


def test_bf16_and_fp32_plans_coexist(model_and_sd, cuda):
    """Switching the compute dtype back and forth re-uses both weight arenas and gives the
    fp32 result bit-for-bit again."""
    m, _ = model_and_sd
    x = (torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(3)) - 0.5).to(cuda)
    with torch.no_grad():
        (p0, _), _ = m(x)
        m.set_compute_dtype('bf16')
        (pb, _), _ = m(x)
        m.set_compute_dtype('fp32')
        (p1, _), _ = m(x)
    assert torch.equal(p0, p1)
    assert not torch.equal(p0, pb)
    assert (p0 - pb).abs().max().item() <= 6e-2 * max(1.0, p0.abs().max().item())


def test_keypoints_agree_bf16_vs_fp32(pkg, model_and_sd, cuda):
    """BASELINE config 3 fallback metric: keypoint agreement between the bf16 and
    the fp32 path on synthetic scenes blended over the network output."""
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
    for dt in ('fp32', 'bf16'):
        m.set_compute_dtype(dt)
        try:
            # He-init outputs are O(4): alpha 2e-2 superimposes ~0.1 of network-made texture
            res[dt] = est.humans(x, scene=scene, scene_alpha=2e-2)
        finally:
            m.set_compute_dtype('fp32')
    tot = same = 0
    assert sum(len(h) for h in res['fp32']) > 0
    for ha, hb in zip(res['fp32'], res['bf16']):
        assert len(ha) == len(hb), "different number of people: %d vs %d" % (len(ha), len(hb))
        for a, b in zip(ha, hb):
            for part, bp in a.body_parts.items():
                tot += 1
                bq = b.body_parts.get(part)
                if bq is not None and abs(bp.x - bq.x) * hw <= 1.0 and abs(bp.y - bq.y) * hw <= 1.0:
                    same += 1
    assert tot > 0
    assert same / tot >= 0.98, "only %d of %d keypoints agree within 1 px" % (same, tot)


def test_bf16_full_size_properties_batch32(model_and_sd, cuda):
    """The bf16 plan at BASELINE size (32 x 3 x 368 x 368): batch-position independence and
    determinism bit for bit; one image against the emulation oracle at full size."""
    from oracle import net_oracle
    m, sd = model_and_sd
    g = torch.Generator().manual_seed(12)
    x = torch.rand(32, 3, 368, 368, generator=g) - 0.5
    perm = torch.randperm(32, generator=g)
    keep = m.keep_intermediates
    m.keep_intermediates = False
    m.set_compute_dtype('bf16')
    try:
        with torch.no_grad():
            (paf, heat), _ = m(x.to(cuda))
            (paf2, heat2), _ = m(x.to(cuda))
            (paf_p, heat_p), _ = m(x[perm].to(cuda))
            (paf_1, heat_1), _ = m(x[7:8].to(cuda))
    finally:
        m.set_compute_dtype('fp32')
        m.keep_intermediates = keep
    assert torch.equal(paf, paf2) and torch.equal(heat, heat2)
    assert torch.equal(paf_p, paf[perm.to(cuda)]) and torch.equal(heat_p, heat[perm.to(cuda)])
    assert torch.equal(paf_1[0], paf[7]) and torch.equal(heat_1[0], heat[7])
    (paf_e, heat_e), _ = net_oracle.forward_bf16_emulated(sd, x[7:8])
    for a, b in ((paf[7].cpu(), paf_e[0]), (heat[7].cpu(), heat_e[0])):
        scale = max(1.0, b.abs().max().item())
        # max over 120k outputs of a 50-layer bf16 chain: rounding flips (which depend on the
        # accumulation order, i.e. on tiling choices) reach ~2 % of the map scale; rms stays ~0.3 %
        assert (a - b).abs().max().item() <= 4e-2 * scale
        assert ((a - b) ** 2).mean().sqrt().item() <= 6e-3 * scale
