"""GPU parity of the fp32-MFMA conv kernel (csrc/conv_mfma.hip) against torch's
CPU conv2d (the ATen arithmetic the reference's nn.Conv2d runs,
lib/network/rtpose_vgg.py:23-35), called through the C ABI."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4  # relative to max|ref|; fp32 fma-chain vs ATen summation order


def _run_conv(capi, dev, n, h, w, cin, cout, k, relu, pool, pad_in, pad_out, seed, groups=1, cin_pad=None,
              winograd=False, only_images=None, skip_ref=False, wino_m=0, scratch=True, first_image=0):
    lib, Layout = capi.lib, capi.Layout
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    if only_images:   # the same random stream, but only some images go through the kernel
        x = x[first_image:first_image + only_images]
        n = only_images
    cin_p = cin_pad or ((cin + 7) // 8 * 8)
    ws, bs, refs = [], [], []
    for gi in range(groups):
        wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        y = None
        if not skip_ref:   # (the CPU reference is the slow part of a case)
            y = F.conv2d(x[:n], wt, b, padding=k // 2)
            if relu:
                y = F.relu(y)
            if pool:
                y = F.max_pool2d(y, 2, 2, 0)
        ws.append(wt.to(dev))
        bs.append(b.to(dev))
        refs.append(y)
    stream = capi.current_stream()
    lin = Layout.padded(cin_p, h, w, pad_in)
    xin = torch.zeros(lib.rtpose_layout_pixels(C.byref(lin), n, h, w) * cin_p, device=dev)
    xd = x[:n].contiguous().to(dev)
    capi.check(lib.rtpose_nchw_to_layout(capi.ptr(xd), capi.ptr(xin), C.byref(lin), cin, cin_p, n, h, w, stream))
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    cstride_out = cout * groups + 3  # odd stride + channel offsets: exercises slices
    descs = (capi.ConvDesc * groups)()
    outs, keep = [], []
    lout_full = Layout.padded(cstride_out, ho, wo, pad_out)
    obuf = torch.zeros(lib.rtpose_layout_pixels(C.byref(lout_full), n, ho, wo) * cstride_out, device=dev)
    for gi in range(groups):
        bp = torch.zeros(lib.rtpose_packed_bias_floats(cout), device=dev)
        if winograd and k == 7 and wino_m:
            wp = torch.zeros(lib.rtpose_packed_weight_floats_winograd7(cout, cin_p, wino_m), device=dev)
            capi.check(lib.rtpose_pack_conv_weights_winograd7(capi.ptr(ws[gi]), capi.ptr(bs[gi]), cout, cin, wino_m,
                                                              None, cin_p, capi.ptr(wp), capi.ptr(bp), stream))
        elif winograd and k == 3 and wino_m:
            wp = torch.zeros(lib.rtpose_packed_weight_floats_winograd3(cout, cin_p, wino_m), device=dev)
            capi.check(lib.rtpose_pack_conv_weights_winograd3(capi.ptr(ws[gi]), capi.ptr(bs[gi]), cout, cin, wino_m,
                                                              None, cin_p, capi.ptr(wp), capi.ptr(bp), stream))
        elif winograd:
            wp = torch.zeros(lib.rtpose_packed_weight_floats_winograd(cout, cin_p, k), device=dev)
            capi.check(lib.rtpose_pack_conv_weights_winograd(capi.ptr(ws[gi]), capi.ptr(bs[gi]), cout, cin, k, None,
                                                             cin_p, capi.ptr(wp), capi.ptr(bp), stream))
        else:
            wp = torch.zeros(lib.rtpose_packed_weight_floats(cout, cin_p, k), device=dev)
            capi.check(lib.rtpose_pack_conv_weights(capi.ptr(ws[gi]), capi.ptr(bs[gi]), cout, cin, k, None, cin_p,
                                                    capi.ptr(wp), capi.ptr(bp), stream))
        keep += [wp, bp]
        d = descs[gi]
        d.inp, d.w_packed, d.bias_packed, d.out = xin.data_ptr(), wp.data_ptr(), bp.data_ptr(), obuf.data_ptr()
        d.lin = lin
        d.lout = Layout.padded(cstride_out, ho, wo, pad_out, choff=gi * cout + 1)
        d.cin, d.cout, d.k, d.relu, d.pool = cin_p, cout, k, int(relu), int(pool)
        d.wino_m = wino_m if winograd else 0
    if winograd:
        assert lib.rtpose_conv2d_winograd_fits(descs, n, h, w) == 1
        # the hand-over scratch of the persistent 7x7 launches is the caller's (the library allocates nothing)
        sc = torch.zeros(lib.rtpose_conv2d_winograd_scratch_bytes() // 4, dtype=torch.int32, device=dev) if scratch else None
        capi.check(lib.rtpose_conv2d_winograd_ex(descs, groups, n, h, w, capi.ptr(sc) if scratch else None,
                                                 sc.numel() * 4 if scratch else 0, stream), "rtpose_conv2d_winograd_ex")
        if scratch:
            word = C.c_int(-1)
            capi.check(lib.rtpose_conv2d_winograd_scratch_error(capi.ptr(sc), C.byref(word), stream))
            assert word.value == 0, "device error word %d" % word.value
    else:
        capi.check(lib.rtpose_conv2d(descs, groups, n, h, w, stream), "rtpose_conv2d")
    for gi in range(groups):
        o = torch.empty(n, cout, ho, wo, device=dev)
        lo = Layout.padded(cstride_out, ho, wo, pad_out, choff=gi * cout + 1)
        capi.check(lib.rtpose_layout_to_nchw(capi.ptr(obuf), C.byref(lo), capi.ptr(o), cout, n, ho, wo, stream))
        outs.append(o.cpu())
    torch.cuda.synchronize()
    # gaps of the output buffer must still be zero (only real pixels are written)
    total = obuf.abs().sum().item()
    inner = sum(o.abs().sum().item() for o in outs)
    assert abs(total - inner) <= 1e-3 * max(1.0, inner), "conv wrote outside its slice / into the gaps"
    return outs, refs


CASES = [
    # n, h, w, cin, cout, k, relu, pool, pad_in, pad_out
    (2, 46, 46, 128, 128, 7, 1, 0, 3, 3),     # Mconv2_stageN: strip mode, exact 128-multiples
    (1, 46, 49, 192, 128, 7, 1, 0, 3, 3),     # ski.jpg geometry (46x49), 185->192 packed input
    (3, 23, 17, 128, 38, 1, 0, 0, 0, 3),      # 1x1 head into a padded concat slice, ragged M
    (2, 46, 46, 512, 19, 1, 0, 0, 0, 0),      # conv5_5_CPM_L2
    (2, 46, 46, 256, 512, 3, 1, 0, 1, 1),     # conv4_1: 8 N-tiles
    (1, 46, 46, 128, 128, 3, 1, 0, 3, 1),     # stage-1 conv reading the P=3 concat layout
    (1, 96, 80, 64, 64, 3, 1, 1, 1, 1),       # 2-D tile mode + fused pool
    (2, 72, 88, 8, 64, 3, 1, 0, 1, 1),        # conv1_1 (3->8 padded input), ck=8, 2-D tiles
    (1, 100, 92, 128, 256, 3, 1, 0, 1, 1),    # 2-D tiles with ragged right/bottom edges
    (1, 70, 66, 128, 128, 7, 1, 0, 3, 0),     # 7x7 in 2-D tile mode (multi-scale maps)
    (1, 12, 10, 16, 24, 3, 0, 0, 1, 0),       # tiny: one partial block, several images' worth of gap
    (5, 6, 6, 16, 8, 7, 1, 0, 3, 3),          # strip spanning several images
]


@pytest.mark.parametrize("case", CASES)
def test_conv_matches_torch_cpu(capi, cuda, case):
    n, h, w, cin, cout, k, relu, pool, pin, pout = case
    src_cin = 3 if cin == 8 else (185 if cin == 192 else cin)
    outs, refs = _run_conv(capi, cuda, n, h, w, src_cin, cout, k, relu, pool, pin, pout, seed=hash(case) % 1000,
                           cin_pad=cin)
    ref = refs[0]
    err = (outs[0] - ref).abs().max().item()
    assert err <= TOL * max(1.0, ref.abs().max().item()), "max abs err %g" % err


def test_grouped_branches(capi, cuda):
    outs, refs = _run_conv(capi, cuda, 2, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=7, groups=2)
    for o, r in zip(outs, refs):
        assert (o - r).abs().max().item() <= TOL * max(1.0, r.abs().max().item())


WINO7_CASES = [
    # n, h, w, cin (packed), cout, relu, pad_in, pad_out      (k = 7; csrc/conv_wino7.hip, F(6,7): groups of 6 pixels)
    (2, 46, 46, 128, 128, 1, 3, 3),      # Mconv2_stageN: 8 position groups per row (W % 6 == 4), strips per image
    (1, 46, 49, 192, 128, 1, 3, 3),      # ski.jpg geometry (46x49), 185 -> 192 packed input, W % 6 == 1
    (3, 23, 18, 64, 256, 0, 3, 0),       # W % 6 == 0, two N tiles, no ReLU, strips crossing images
    (5, 6, 7, 16, 128, 1, 3, 3),         # tiny maps: a block spans several images (W % 6 == 1)
    (2, 20, 27, 32, 128, 1, 3, 3),       # W % 6 == 3
    (1, 17, 35, 8, 128, 0, 3, 1),        # W % 6 == 5, one chunk
    (1, 30, 44, 24, 128, 1, 3, 0),       # W % 6 == 2
    (1, 70, 66, 128, 128, 1, 3, 0),      # multi-scale map, wider rows
    (1, 9, 80, 8, 128, 1, 4, 1),         # one chunk; 20 groups per row; gap wider than the padding
]


@pytest.mark.parametrize("case", WINO7_CASES)
def test_winograd7_matches_torch_cpu(capi, cuda, case):
    n, h, w, cin, cout, relu, pin, pout = case
    src_cin = 185 if cin == 192 else cin
    outs, refs = _run_conv(capi, cuda, n, h, w, src_cin, cout, 7, relu, 0, pin, pout, seed=hash(case) % 1000,
                           cin_pad=cin, winograd=True)
    ref = refs[0]
    err = (outs[0] - ref).abs().max().item()
    assert err <= TOL * max(1.0, ref.abs().max().item()), "max abs err %g" % err


def test_winograd7_grouped_branches_and_direct_agree(capi, cuda):
    outs, refs = _run_conv(capi, cuda, 2, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=7, groups=2, winograd=True)
    direct, _ = _run_conv(capi, cuda, 2, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=7, groups=2)
    for o, r, d in zip(outs, refs, direct):
        assert (o - r).abs().max().item() <= TOL * max(1.0, r.abs().max().item())
        assert (o - d).abs().max().item() <= 1e-4 * max(1.0, r.abs().max().item())


def test_winograd7_persistent_blocks_split_tiles(capi, cuda):
    """12 x 46 x 46, two branches = 288 tiles (F(6,7): 12 strips per image) >= 256 CUs: the launch runs as persistent blocks that share the
    (tile, chunk) units evenly, most tiles are split between two blocks (conv_wino7.hip: wino7_f32).  The second
    block continues the first one's sums, so the result is BIT-identical to the one-block-per-tile launch of a
    smaller batch of the same images; run twice: the hand-over flags are back to zero after a launch."""
    outs, refs = _run_conv(capi, cuda, 12, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=11, groups=2, winograd=True)
    again, _ = _run_conv(capi, cuda, 12, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=11, groups=2, winograd=True)
    for o, r, o2 in zip(outs, refs, again):
        assert (o - r).abs().max().item() <= TOL * max(1.0, r.abs().max().item())
        assert torch.equal(o, o2)
    # same generator stream: the first 2 images / both filters of an n = 2 run are those of the n = 8 run? no -
    # the inputs are drawn as one (n, c, h, w) tensor, so re-run the FIRST image alone through a slice instead
    small, _ = _run_conv(capi, cuda, 12, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=11, groups=2, winograd=True,
                         only_images=3)
    for o, sm in zip(outs, small):
        assert torch.equal(o[:3], sm)
    # without a scratch the same launch runs one block per tile: the same bits again
    plain, _ = _run_conv(capi, cuda, 12, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=11, groups=2, winograd=True,
                         scratch=False, skip_ref=True)
    for o, pl in zip(outs, plain):
        assert torch.equal(o, pl)


def test_winograd7_f47_form_selected_per_launch(capi, cuda):
    """rtpose_conv_desc.wino_m = 4 with the F(4,7) packing: the form is a property of the launch, not of the
    process (round 2 read it from the environment); persistent split tiles (14 x 46 x 46 x 2 = 504 tiles) included."""
    f4, refs = _run_conv(capi, cuda, 2, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=13, groups=2, winograd=True, wino_m=4)
    f6, _ = _run_conv(capi, cuda, 2, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=13, groups=2, winograd=True, wino_m=6,
                      skip_ref=True)
    for a, b, r in zip(f4, f6, refs):
        assert (a - r).abs().max().item() <= TOL * max(1.0, r.abs().max().item())
        assert (b - r).abs().max().item() <= TOL * max(1.0, r.abs().max().item())
        assert not torch.equal(a, b)      # two different arithmetic forms
    big, _ = _run_conv(capi, cuda, 14, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=13, groups=2, winograd=True, wino_m=4,
                       skip_ref=True)
    plain, _ = _run_conv(capi, cuda, 14, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=13, groups=2, winograd=True, wino_m=4,
                         skip_ref=True, scratch=False)
    for a, b in zip(big, plain):
        assert torch.equal(a, b)


def test_winograd3_small_grid_form_is_bit_identical(capi, cuda):
    """3x3, 9 x 46 x 46, 256 -> 512: 152 m tiles x 4 n tiles run as whole 32 x 128 tiles (wino_f32); the first image
    alone is a small grid and runs in the frequency-split form (wino3s_f32, 32 x 32 tiles): same bits.  Also the
    64-column block (conv1_2-like, pooled) against its small-grid form."""
    big, refs = _run_conv(capi, cuda, 9, 46, 46, 256, 512, 3, 1, 0, 1, 1, seed=21, winograd=True)
    one, _ = _run_conv(capi, cuda, 9, 46, 46, 256, 512, 3, 1, 0, 1, 1, seed=21, winograd=True, only_images=1)
    assert (big[0] - refs[0]).abs().max().item() <= TOL * max(1.0, refs[0].abs().max().item())
    assert torch.equal(big[0][:1], one[0])
    big, _ = _run_conv(capi, cuda, 6, 96, 80, 64, 64, 3, 1, 1, 1, 1, seed=22, winograd=True, skip_ref=True)
    one, _ = _run_conv(capi, cuda, 6, 96, 80, 64, 64, 3, 1, 1, 1, 1, seed=22, winograd=True, only_images=1,
                       skip_ref=True)
    assert torch.equal(big[0][:1], one[0])


WINO_CASES = [
    # n, h, w, cin, cout, relu, pool, pad_in, pad_out       (k = 3; csrc/conv_wino.hip)
    (2, 46, 46, 256, 512, 1, 0, 1, 1),     # conv4_1: 32 wtiles x 128 columns, 4 N tiles, XCD-ordered grid
    (1, 46, 46, 128, 128, 1, 0, 3, 1),     # stage-1 conv reading the P = 3 concat layout
    (1, 96, 80, 64, 64, 1, 1, 1, 1),       # conv1_2: 64 wtiles x 64 columns, 8-channel chunks, fused pool
    (1, 100, 92, 128, 256, 1, 0, 1, 1),    # conv3_1
    (2, 23, 17, 64, 128, 0, 0, 1, 0),      # odd H and W: the last wtile row / column is half outside
    (1, 12, 10, 32, 24, 0, 0, 1, 0),       # tiny: one partial block, ragged cout (two chunks: the minimum)
    (5, 6, 6, 32, 40, 1, 1, 1, 3),         # wtile strips spanning several images + pool
    (3, 7, 5, 24, 64, 1, 0, 2, 1),         # three 8-channel chunks (cin not a multiple of 16)
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_matches_torch_cpu(capi, cuda, case):
    n, h, w, cin, cout, relu, pool, pin, pout = case
    outs, refs = _run_conv(capi, cuda, n, h, w, cin, cout, 3, relu, pool, pin, pout, seed=hash(case) % 1000,
                           winograd=True)
    ref = refs[0]
    err = (outs[0] - ref).abs().max().item()
    assert err <= TOL * max(1.0, ref.abs().max().item()), "max abs err %g" % err


WINO4_CASES = [
    # n, h, w, cin (packed), cout, relu, pool, pad_in, pad_out      (k = 3, F(4x4,3x3): csrc/conv_wino4.hip)
    (2, 46, 46, 256, 512, 1, 0, 1, 1),     # conv4_1: 8 column tiles, H, W % 4 == 2 (clamped patch rows / columns)
    (1, 46, 46, 128, 128, 1, 0, 3, 1),     # stage-1 conv reading the P = 3 concat layout
    (3, 96, 80, 64, 64, 1, 1, 1, 1),       # conv1_2 shape: fused pool, one column tile
    (1, 100, 92, 128, 256, 1, 0, 1, 1),    # H, W % 4 == 0
    (2, 45, 47, 32, 64, 0, 0, 1, 0),       # odd sizes (W % 4 == 3, H % 4 == 1), no ReLU, 4 chunks
    (5, 7, 9, 48, 24, 1, 0, 1, 3),         # tiny maps: a block spans several images; ragged cout; 6 chunks
    (1, 30, 44, 80, 128, 1, 1, 2, 0),      # pool with H % 4 == 2; gap wider than the padding
    (40, 46, 46, 64, 128, 1, 0, 1, 1),     # persistent blocks: 180 m tiles x 2 column tiles > 256
]


@pytest.mark.parametrize("case", WINO4_CASES)
def test_winograd4_matches_torch_cpu(capi, cuda, case):
    n, h, w, cin, cout, relu, pool, pin, pout = case
    outs, refs = _run_conv(capi, cuda, n, h, w, cin, cout, 3, relu, pool, pin, pout, seed=hash(case) % 1000,
                           cin_pad=cin, winograd=True, wino_m=4)
    ref = refs[0]
    err = (outs[0] - ref).abs().max().item()
    assert err <= TOL * max(1.0, ref.abs().max().item()), "max abs err %g" % err


def test_winograd4_grouped_branches_and_batch_invariance(capi, cuda):
    outs, refs = _run_conv(capi, cuda, 2, 46, 46, 128, 128, 3, 1, 0, 3, 3, seed=9, groups=2, winograd=True, wino_m=4)
    for o, r in zip(outs, refs):
        assert (o - r).abs().max().item() <= TOL * max(1.0, r.abs().max().item())
    # an image's result does not depend on its neighbours in the batch (clamped patch rows / columns)
    big, _ = _run_conv(capi, cuda, 3, 45, 47, 64, 64, 3, 1, 0, 1, 1, seed=31, winograd=True, wino_m=4, skip_ref=True)
    one, _ = _run_conv(capi, cuda, 3, 45, 47, 64, 64, 3, 1, 0, 1, 1, seed=31, winograd=True, wino_m=4, skip_ref=True,
                       only_images=1)
    assert torch.equal(big[0][:1], one[0])


def test_winograd4_small_grid_form_is_bit_identical(capi, cuda):
    """Few tiles (batch 1): F(4x4,3x3) launches its 16 x 16 form (wino4s_f32: six waves split the frequencies) - the same
    sums in the same order as the 32 x 64 kernel with its persistent blocks: an image's result does not depend on the
    batch it is evaluated in."""
    big, refs = _run_conv(capi, cuda, 9, 46, 46, 256, 512, 3, 1, 0, 1, 1, seed=21, winograd=True, wino_m=4)
    one, _ = _run_conv(capi, cuda, 9, 46, 46, 256, 512, 3, 1, 0, 1, 1, seed=21, winograd=True, wino_m=4, only_images=1)
    assert (big[0] - refs[0]).abs().max().item() <= TOL * max(1.0, refs[0].abs().max().item())
    assert torch.equal(big[0][:1], one[0])
    # fused pool, one column tile of 64 (four of 16 in the small form), odd tile counts, two branches
    big, _ = _run_conv(capi, cuda, 20, 96, 80, 64, 64, 3, 1, 1, 1, 1, seed=22, winograd=True, wino_m=4, skip_ref=True)
    one, _ = _run_conv(capi, cuda, 20, 96, 80, 64, 64, 3, 1, 1, 1, 1, seed=22, winograd=True, wino_m=4, only_images=1,
                       skip_ref=True)
    assert torch.equal(big[0][:1], one[0])
    big, _ = _run_conv(capi, cuda, 40, 45, 47, 128, 128, 3, 1, 0, 3, 3, seed=23, groups=2, winograd=True, wino_m=4,
                       skip_ref=True)
    two, _ = _run_conv(capi, cuda, 40, 45, 47, 128, 128, 3, 1, 0, 3, 3, seed=23, groups=2, winograd=True, wino_m=4,
                       only_images=2, skip_ref=True)
    for a, b in zip(big, two):
        assert torch.equal(a[:2], b)
    # the stage-1 geometry of the bench: 144 m tiles x 4 (column tile, branch) on 256 CUs = 2.25 rounds -> two persistent
    # rounds + the last 16 m tiles as a second launch in the small form; the cut is invisible
    big, _ = _run_conv(capi, cuda, 32, 46, 46, 128, 128, 3, 1, 0, 3, 3, seed=24, groups=2, winograd=True, wino_m=4,
                       skip_ref=True)
    for first, count in ((0, 2), (30, 2)):
        part, _ = _run_conv(capi, cuda, 32, 46, 46, 128, 128, 3, 1, 0, 3, 3, seed=24, groups=2, winograd=True, wino_m=4,
                            skip_ref=True, only_images=count, first_image=first)
        for a, b in zip(big, part):
            assert torch.equal(a[first:first + count], b)


@pytest.mark.parametrize("geom", [(32, 92, 92, 32, 256, 1), (32, 46, 46, 64, 512, 1), (32, 45, 47, 64, 256, 2)])
def test_winograd4_half_tiles_of_the_left_over_round_are_bit_identical(capi, cuda, geom):
    """Round 4: when the 32 x 64 tiles of a layer are not whole rounds of the CUs and 25..50 % of a round is left over (the
    92 x 92 layers of the 32-image batch: 8.27 rounds; 46 x 46 with 512 columns: 4.5), the left-over runs as one round of HALF
    tiles (wino4_f32<1>: 16 wtiles x 64 columns).  Same sums in the same order: the images of the batch's tail - they lie in
    the left-over - and of its head are the bits of the same images run alone (a small grid: the 16 x 16 form)."""
    n, h, w, cin, cout, groups = geom
    big, _ = _run_conv(capi, cuda, n, h, w, cin, cout, 3, 1, 0, 1, 1, seed=41, groups=groups, cin_pad=cin, winograd=True,
                       wino_m=4, skip_ref=True)
    for first in (0, n - 2, n - 1):
        one, _ = _run_conv(capi, cuda, n, h, w, cin, cout, 3, 1, 0, 1, 1, seed=41, groups=groups, cin_pad=cin, winograd=True,
                           wino_m=4, skip_ref=True, only_images=1, first_image=first)
        for a, b in zip(big, one):
            assert torch.equal(a[first:first + 1], b), first
    if cin == 32:   # one image of the tail against torch's CPU conv2d as well
        g = torch.Generator().manual_seed(41)
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        y = F.relu(F.conv2d(x[n - 1:], wt, b, padding=1))
        assert (big[0][n - 1:] - y).abs().max().item() <= TOL * max(1.0, y.abs().max().item())


def _to_planes(pm, q_slots, lead_planes=0, fill=0.0):
    """pixel-major [Q, C] -> channel planes [lead_planes + C / 8][q_slots][8] (flat); unused slots / planes = `fill`."""
    q, c = pm.shape
    out = torch.full((lead_planes + c // 8, q_slots, 8), fill, device=pm.device)
    out[lead_planes:, :q] = pm.view(q, c // 8, 8).permute(1, 0, 2)
    return out.reshape(-1)


def _from_planes(buf, q_slots):
    """channel planes (flat) -> pixel-major [q_slots, 8 * planes]"""
    return buf.view(-1, q_slots, 8).permute(1, 0, 2).reshape(q_slots, -1).contiguous()


def _run_conv4_planes(capi, dev, n, h, w, cin, cout, relu, pool, pad_in, pad_out, seed, groups, in_planes, out_planes):
    """F(4x4,3x3) through the C ABI with the input and / or the output stored as channel planes
    (rtpose_conv_desc.in_plane_pixels / out_plane_pixels); same random stream as _run_conv."""
    lib, Layout = capi.lib, capi.Layout
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    ws, bs = [], []
    for gi in range(groups):
        ws.append((torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(dev))
        bs.append((torch.randn(cout, generator=g) * 0.1).to(dev))
    stream = capi.current_stream()
    lin = Layout.padded(cin, h, w, pad_in)
    q_in = lib.rtpose_layout_pixels(C.byref(lin), n, h, w)
    xin = torch.zeros(q_in * cin, device=dev)
    xd = x.contiguous().to(dev)
    capi.check(lib.rtpose_nchw_to_layout(capi.ptr(xd), capi.ptr(xin), C.byref(lin), cin, cin, n, h, w, stream))
    torch.cuda.synchronize()
    qs_in = q_in + 5                               # more slots per plane than the layout has pixels
    if in_planes:                                  # one plane of junk in front: the slice starts at channel 8
        xin = _to_planes(xin.view(q_in, cin), qs_in, lead_planes=1, fill=7.0)
        xin[:qs_in * 8] = 7.0
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    ctot = cout * groups + 8
    lout = Layout.padded(ctot, ho, wo, pad_out)
    q_out = lib.rtpose_layout_pixels(C.byref(lout), n, ho, wo)
    qs_out = q_out + 3
    obuf = torch.zeros((qs_out if out_planes else q_out) * ctot, device=dev)
    descs = (capi.ConvDesc * groups)()
    keep = []
    for gi in range(groups):
        wp = torch.zeros(lib.rtpose_packed_weight_floats_winograd3(cout, cin, 4), device=dev)
        bp = torch.zeros(lib.rtpose_packed_bias_floats(cout), device=dev)
        capi.check(lib.rtpose_pack_conv_weights_winograd3(capi.ptr(ws[gi]), capi.ptr(bs[gi]), cout, cin, 4, None, cin,
                                                          capi.ptr(wp), capi.ptr(bp), stream))
        keep += [wp, bp]
        d = descs[gi]
        d.inp, d.w_packed, d.bias_packed, d.out = xin.data_ptr(), wp.data_ptr(), bp.data_ptr(), obuf.data_ptr()
        d.lin = Layout.padded(cin + 8, h, w, pad_in, choff=8) if in_planes else lin
        d.lout = Layout.padded(ctot, ho, wo, pad_out, choff=8 + gi * cout)
        d.cin, d.cout, d.k, d.relu, d.pool, d.wino_m = cin, cout, 3, int(relu), int(pool), 4
        d.in_plane_pixels = qs_in if in_planes else 0
        d.out_plane_pixels = qs_out if out_planes else 0
    capi.check(lib.rtpose_conv2d_winograd_ex(descs, groups, n, h, w, None, 0, stream), "rtpose_conv2d_winograd_ex (planes)")
    torch.cuda.synchronize()
    pm = _from_planes(obuf, qs_out)[:q_out].contiguous() if out_planes else obuf
    outs = []
    for gi in range(groups):
        o = torch.empty(n, cout, ho, wo, device=dev)
        lo = Layout.padded(ctot, ho, wo, pad_out, choff=8 + gi * cout)
        capi.check(lib.rtpose_layout_to_nchw(capi.ptr(pm), C.byref(lo), capi.ptr(o), cout, n, ho, wo, stream))
        outs.append(o.cpu())
    torch.cuda.synchronize()
    total, inner = obuf.abs().sum().item(), sum(o.abs().sum().item() for o in outs)
    assert abs(total - inner) <= 1e-3 * max(1.0, inner), "conv wrote outside its slice / into the gaps / into the slack slots"
    return outs


PLANE_CASES = [
    # n, h, w, cin, cout, relu, pool, pad_in, pad_out, groups
    (2, 46, 46, 64, 128, 1, 0, 1, 1, 1),      # small-grid form (wino4s_f32)
    (40, 46, 46, 64, 128, 1, 0, 1, 1, 1),     # persistent blocks
    (3, 96, 80, 64, 64, 1, 1, 1, 1, 1),       # fused pool
    (5, 7, 9, 48, 24, 0, 0, 1, 3, 1),         # tiny maps, ragged cout (24 = 3 planes), no ReLU
    (20, 45, 47, 128, 128, 1, 0, 3, 3, 2),    # two branches in one grid, odd sizes, one round + a left-over launch
    (32, 92, 92, 32, 256, 1, 0, 1, 1, 1),     # 8.27 rounds: whole rounds + a round of half tiles (wino4_f32<1>)
]


@pytest.mark.parametrize("case", PLANE_CASES)
def test_winograd4_channel_planes_are_bit_identical_to_pixel_major(capi, cuda, case):
    """Round 4: the F(4x4,3x3) kernel reads and / or writes activations stored as planes of 8 channels
    (rtpose_conv_desc.in_plane_pixels / out_plane_pixels: the storage the executor keeps between the convs of the VGG front
    end) - only addresses change: every combination gives the bits of the pixel-major launch, nothing lands in the gaps, the
    slack slots of a plane or a neighbouring plane, and a junk plane in front of the slice is not read."""
    n, h, w, cin, cout, relu, pool, pin, pout, groups = case
    ref = _run_conv4_planes(capi, cuda, n, h, w, cin, cout, relu, pool, pin, pout, 77, groups, False, False)
    for inp, outp in ((True, False), (False, True), (True, True)):
        got = _run_conv4_planes(capi, cuda, n, h, w, cin, cout, relu, pool, pin, pout, 77, groups, inp, outp)
        for a, b in zip(got, ref):
            assert torch.equal(a, b), (inp, outp)
    if n <= 5:   # and the pixel-major launch is the one the other tests pin to torch's CPU conv2d
        g = torch.Generator().manual_seed(77)
        x = torch.randn(n, cin, h, w, generator=g)
        for gi in range(groups):
            wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
            b = torch.randn(cout, generator=g) * 0.1
            y = F.conv2d(x, wt, b, padding=1)
            y = F.relu(y) if relu else y
            y = F.max_pool2d(y, 2, 2, 0) if pool else y
            assert (ref[gi] - y).abs().max().item() <= TOL * max(1.0, y.abs().max().item())


def test_channel_planes_are_refused_where_no_kernel_reads_them(capi, cuda):
    lib, Layout = capi.lib, capi.Layout
    d = (capi.ConvDesc * 1)()
    buf = torch.zeros(1 << 16, device=cuda)
    d[0].inp = d[0].w_packed = d[0].bias_packed = d[0].out = buf.data_ptr()
    d[0].lin = Layout.padded(32, 8, 8, 1)
    d[0].lout = Layout.padded(64, 8, 8, 1)
    d[0].cin, d[0].cout, d[0].k, d[0].relu = 32, 64, 3, 1
    d[0].in_plane_pixels = 4096
    assert lib.rtpose_conv2d(d, 1, 1, 8, 8, None) != 0 and "channel-plane" in capi.last_error()
    d[0].wino_m = 2
    assert lib.rtpose_conv2d_winograd_ex(d, 1, 1, 8, 8, None, 0, None) != 0 and "channel-plane" in capi.last_error()
    d[0].wino_m = 4
    d[0].in_plane_pixels = 10                     # fewer slots than the layout has pixels
    assert lib.rtpose_conv2d_winograd_ex(d, 1, 1, 8, 8, None, 0, None) != 0 and "plane" in capi.last_error()
    d[0].in_plane_pixels = 4096
    d[0].lin = Layout.padded(40, 8, 8, 1, choff=4)   # a slice that does not start on a plane
    assert lib.rtpose_conv2d_winograd_ex(d, 1, 1, 8, 8, None, 0, None) != 0 and "plane" in capi.last_error()
    d[0].lin = Layout.padded(32, 8, 8, 1)
    d[0].out_plane_pixels = 4096
    d[0].cout = 20                                   # not whole planes
    assert lib.rtpose_conv2d_winograd_ex(d, 1, 1, 8, 8, None, 0, None) != 0 and "plane" in capi.last_error()


@pytest.mark.parametrize("shape", [(2, 64, 72), (1, 37, 45), (1, 368, 368)])
def test_first_layer_kernel_writes_channel_planes_bit_identically(capi, cuda, shape):
    """rtpose_conv_first_planes (the transposed product: a lane holds one pixel and 16 channels) against rtpose_conv_first:
    the same bits, the slack slots and the neighbouring planes untouched."""
    lib, Layout = capi.lib, capi.Layout
    n, h, w = shape
    g = torch.Generator().manual_seed(h * 1000 + w + 1)
    x = (torch.rand(n, 3, h, w, generator=g) - 0.5).to(cuda)
    wd = (torch.randn(64, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5).to(cuda)
    bd = (torch.randn(64, generator=g) * 0.1).to(cuda)
    stream = capi.current_stream()
    wp = torch.zeros(lib.rtpose_conv_first_packed_floats(), device=cuda)
    capi.check(lib.rtpose_pack_conv_first(capi.ptr(wd), capi.ptr(bd), capi.ptr(wp), stream))
    lo_pm = Layout.padded(64, h, w, 1)
    q = lib.rtpose_layout_pixels(C.byref(lo_pm), n, h, w)
    o_pm = torch.zeros(q * 64, device=cuda)
    capi.check(lib.rtpose_conv_first(capi.ptr(x), None, None, capi.ptr(wp), capi.ptr(o_pm), C.byref(lo_pm), 1, n, h, w, stream))
    qs = q + 7
    lo_pl = Layout.padded(80, h, w, 1, choff=8)     # planes 1..8 of a 10-plane buffer
    o_pl = torch.zeros(qs * 80, device=cuda)
    capi.check(lib.rtpose_conv_first_planes(capi.ptr(x), None, None, capi.ptr(wp), capi.ptr(o_pl), C.byref(lo_pl), qs, 1,
                                            n, h, w, stream), "rtpose_conv_first_planes")
    torch.cuda.synchronize()
    pm = _from_planes(o_pl, qs)
    assert torch.equal(pm[:q, 8:72], o_pm.view(q, 64))
    assert pm[:, :8].abs().sum().item() == 0 and pm[:, 72:].abs().sum().item() == 0 and pm[q:].abs().sum().item() == 0
    assert lib.rtpose_conv_first_planes(capi.ptr(x), None, None, capi.ptr(wp), capi.ptr(o_pl), C.byref(lo_pl), q - 1, 1,
                                        n, h, w, stream) != 0


def test_winograd_grouped_branches_and_direct_agree(capi, cuda):
    outs, refs = _run_conv(capi, cuda, 2, 46, 46, 128, 128, 3, 1, 0, 3, 3, seed=9, groups=2, winograd=True)
    direct, _ = _run_conv(capi, cuda, 2, 46, 46, 128, 128, 3, 1, 0, 3, 3, seed=9, groups=2)
    for o, r, d in zip(outs, refs, direct):
        assert (o - r).abs().max().item() <= TOL * max(1.0, r.abs().max().item())
        assert (o - d).abs().max().item() <= 2e-5 * max(1.0, r.abs().max().item())


def test_winograd_rejects_what_it_cannot_do(capi, cuda):
    lib = capi.lib
    d = (capi.ConvDesc * 1)()

    def fits(k, cin, cout, n=1, h=46, w=46, pool=0):
        d[0].k, d[0].cin, d[0].cout, d[0].pool = k, cin, cout, pool
        d[0].lin = capi.Layout.padded(cin, h, w, k // 2)
        return lib.rtpose_conv2d_winograd_fits(d, n, h, w)

    assert fits(3, 8, 128) == 0 and fits(3, 24, 64) == 1 and fits(3, 512, 512) == 1 and fits(1, 128, 128) == 0
    assert fits(3, 8, 64) == 0 and fits(3, 16, 128) == 0 and fits(3, 32, 128) == 1    # at least two chunks
    assert fits(7, 128, 128, 32) == 1 and fits(7, 192, 128, 32) == 1
    assert fits(7, 128, 38) == 0            # 64 padded columns: the stage heads are 1x1 anyway
    assert fits(7, 128, 128, 2, 184, 184) == 0   # transformed rows of a 184-wide map do not fit the LDS
    d[0].k, d[0].cin, d[0].cout = 5, 16, 128
    assert lib.rtpose_conv2d_winograd(d, 1, 1, 8, 8, None) != 0
    assert "k must be 3" in capi.last_error()


def test_winograd_random_geometries_match_direct(capi, cuda):
    """Both Winograd kernels against the direct kernel (GPU vs GPU, same packed inputs) over random geometries:
    odd / tiny / wide maps, every W modulo 6 and 2, strips crossing images, one and two branches, persistent and
    small-grid launches of the 7x7 form."""
    import random
    rnd = random.Random(20260924)
    lib = capi.lib
    d = (capi.ConvDesc * 1)()
    tried = 0
    for _ in range(40):
        k = rnd.choice((3, 7))
        n = rnd.choice((1, 2, 3, 5, 14))
        h, w = rnd.randint(3, 40), rnd.randint(3, 60)
        cin = rnd.choice((32, 48, 64, 96)) if k == 3 else rnd.choice((8, 24, 64))
        cout = rnd.choice((64, 128, 200)) if k == 3 else rnd.choice((128, 256))
        groups = rnd.choice((1, 2))
        relu = rnd.choice((0, 1))
        pool = 1 if (k == 3 and h % 2 == 0 and w % 2 == 0 and rnd.random() < 0.3) else 0
        pin = k // 2 + rnd.choice((0, 1))
        d[0].k, d[0].cin, d[0].cout, d[0].pool = k, cin, cout, pool
        d[0].lin = capi.Layout.padded(cin, h, w, pin)
        if not lib.rtpose_conv2d_winograd_fits(d, n, h, w):
            continue
        tried += 1
        seed = rnd.randint(0, 10 ** 6)
        wino, _ = _run_conv(capi, cuda, n, h, w, cin, cout, k, relu, pool, pin, 1, seed=seed, groups=groups,
                            winograd=True, skip_ref=True)
        direct, _ = _run_conv(capi, cuda, n, h, w, cin, cout, k, relu, pool, pin, 1, seed=seed, groups=groups,
                              skip_ref=True)
        for a, b in zip(wino, direct):
            err = (a - b).abs().max().item()
            assert err <= 1e-4 * max(1.0, b.abs().max().item()), (k, n, h, w, cin, cout, groups, pool, err)
        d[0].wino_m = 4
        if k == 3 and lib.rtpose_conv2d_winograd_fits(d, n, h, w):   # the F(4x4,3x3) form of the same conv
            wino4, _ = _run_conv(capi, cuda, n, h, w, cin, cout, k, relu, pool, pin, 1, seed=seed, groups=groups,
                                 winograd=True, skip_ref=True, wino_m=4)
            for a, b in zip(wino4, direct):
                err = (a - b).abs().max().item()
                assert err <= 1e-4 * max(1.0, b.abs().max().item()), ("F(4x4,3x3)", n, h, w, cin, cout, groups, pool, err)
        d[0].wino_m = 0
    assert tried >= 25


def test_conv_rejects_bad_geometry(capi, cuda):
    lib = capi.lib
    d = (capi.ConvDesc * 1)()
    d[0].k = 5
    d[0].cin = 16
    assert lib.rtpose_conv2d(d, 1, 1, 8, 8, None) != 0
    assert "k must be" in capi.last_error()


@pytest.mark.parametrize("shape", [(2, 64, 72), (1, 37, 45), (3, 8, 8), (1, 368, 368)])
def test_first_layer_kernel_matches_torch_cpu_from_both_sources(capi, cuda, shape):
    """conv1_1 (3 -> 64, 3x3, pad 1) + ReLU (csrc/conv_first.hip) against torch's CPU conv2d, reading the image as
    dense NCHW and from an NHWC8 layout buffer (the two sources the executor feeds it from): same bits both ways,
    gaps of the output layout untouched, ragged tiles (H, W not multiples of 8 / 32)."""
    lib, Layout = capi.lib, capi.Layout
    n, h, w = shape
    g = torch.Generator().manual_seed(h * 1000 + w)
    x = torch.rand(n, 3, h, w, generator=g) - 0.5
    wt = torch.randn(64, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5
    b = torch.randn(64, generator=g) * 0.1
    ref = F.relu(F.conv2d(x, wt, b, padding=1))
    stream = capi.current_stream()
    wp = torch.zeros(lib.rtpose_conv_first_packed_floats(), device=cuda)
    wd, bd = wt.to(cuda), b.to(cuda)      # (named: a temporary would be freed - and its block reused - before the launch)
    capi.check(lib.rtpose_pack_conv_first(capi.ptr(wd), capi.ptr(bd), capi.ptr(wp), stream))
    lout = Layout.padded(64 + 5, h, w, 1, choff=3)
    outs = []
    xd = x.contiguous().to(cuda)
    lin = Layout.padded(8, h, w, 1)
    xin = torch.zeros(lib.rtpose_layout_pixels(C.byref(lin), n, h, w) * 8, device=cuda)
    capi.check(lib.rtpose_nchw_to_layout(capi.ptr(xd), capi.ptr(xin), C.byref(lin), 3, 8, n, h, w, stream))
    for src in ("nchw", "layout"):
        obuf = torch.zeros(lib.rtpose_layout_pixels(C.byref(lout), n, h, w) * (64 + 5), device=cuda)
        capi.check(lib.rtpose_conv_first(capi.ptr(xd) if src == "nchw" else None, capi.ptr(xin), C.byref(lin), capi.ptr(wp),
                                         capi.ptr(obuf), C.byref(lout), 1, n, h, w, stream), "rtpose_conv_first")
        o = torch.empty(n, 64, h, w, device=cuda)
        capi.check(lib.rtpose_layout_to_nchw(capi.ptr(obuf), C.byref(lout), capi.ptr(o), 64, n, h, w, stream))
        torch.cuda.synchronize()
        assert abs(obuf.abs().sum().item() - o.abs().sum().item()) <= 1e-3 * max(1.0, o.abs().sum().item())
        outs.append(o.cpu())
    assert (outs[0] - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("shape", [(2, 46, 46, 128), (1, 9, 13, 128), (3, 23, 17, 128), (2, 46, 46, 512), (1, 11, 7, 512)])
def test_pointwise_pair_is_bit_identical_to_two_launches(capi, cuda, shape):
    """Mconv6 + Mconv7 as one back-to-back launch (csrc/conv_tail.hip, both branches grouped) against the same two
    convs as separate rtpose_conv2d launches: identical bits (same summation order), gaps of the concat-style
    output buffer untouched, and against torch's CPU conv2d within the usual tolerance."""
    lib, Layout = capi.lib, capi.Layout
    n, h, w, mid = shape            # mid = 128: Mconv6 / Mconv7; 512: conv5_4_CPM / conv5_5_CPM of stage 1
    g = torch.Generator().manual_seed(h * 100 + w)
    stream = capi.current_stream()
    couts = (38, 19)
    xs = [torch.randn(n, 128, h, w, generator=g) for _ in range(2)]
    w1 = [torch.randn(mid, 128, 1, 1, generator=g) * (2.0 / 128) ** 0.5 for _ in range(2)]
    b1 = [torch.randn(mid, generator=g) * 0.1 for _ in range(2)]
    w2 = [torch.randn(c, mid, 1, 1, generator=g) * (2.0 / mid) ** 0.5 for c in couts]
    b2 = [torch.randn(c, generator=g) * 0.1 for c in couts]
    lin = Layout.padded(128, h, w, 0)
    lmid = Layout.padded(mid, h, w, 0)
    keep = []

    def dev(t):
        t = t.contiguous().to(cuda)
        keep.append(t)
        return t

    def pack(wt, b, cout):
        cin = wt.shape[1]
        wp = torch.zeros(lib.rtpose_packed_weight_floats(cout, cin, 1), device=cuda)
        bp = torch.zeros(lib.rtpose_packed_bias_floats(cout), device=cuda)
        capi.check(lib.rtpose_pack_conv_weights(capi.ptr(dev(wt)), capi.ptr(dev(b)), cout, cin, 1, None, cin, capi.ptr(wp),
                                                capi.ptr(bp), stream))
        keep.extend([wp, bp])
        return wp, bp

    xin, p1, p2 = [], [], []
    for gi in range(2):
        buf = torch.zeros(lib.rtpose_layout_pixels(C.byref(lin), n, h, w) * 128, device=cuda)
        capi.check(lib.rtpose_nchw_to_layout(capi.ptr(dev(xs[gi])), capi.ptr(buf), C.byref(lin), 128, 128, n, h, w, stream))
        xin.append(buf)
        p1.append(pack(w1[gi], b1[gi], mid))
        p2.append(pack(w2[gi], b2[gi], couts[gi]))
    results = []
    for fused in (True, False):
        lcat = Layout.padded(192, h, w, 3)
        cat = torch.zeros(lib.rtpose_layout_pixels(C.byref(lcat), n, h, w) * 192, device=cuda)
        mids = [torch.zeros(lib.rtpose_layout_pixels(C.byref(lmid), n, h, w) * mid, device=cuda) for _ in range(2)]
        d1, d2 = (capi.ConvDesc * 2)(), (capi.ConvDesc * 2)()
        for gi in range(2):
            d1[gi].inp, d1[gi].w_packed, d1[gi].bias_packed, d1[gi].out = (xin[gi].data_ptr(), p1[gi][0].data_ptr(),
                                                                         p1[gi][1].data_ptr(), mids[gi].data_ptr())
            d1[gi].lin, d1[gi].lout = lin, lmid
            d1[gi].cin, d1[gi].cout, d1[gi].k, d1[gi].relu, d1[gi].pool = 128, mid, 1, 1, 0
            d2[gi].inp, d2[gi].w_packed, d2[gi].bias_packed, d2[gi].out = (mids[gi].data_ptr(), p2[gi][0].data_ptr(),
                                                                         p2[gi][1].data_ptr(), cat.data_ptr())
            d2[gi].lin = lmid
            d2[gi].lout = Layout.padded(192, h, w, 3, choff=128 if gi == 0 else 166)
            d2[gi].cin, d2[gi].cout, d2[gi].k, d2[gi].relu, d2[gi].pool = mid, couts[gi], 1, 0, 0
        if fused:
            assert lib.rtpose_conv1x1_pair_fits(d1, d2, 2) == 1
            capi.check(lib.rtpose_conv1x1_pair(d1, d2, 2, n, h, w, stream), "rtpose_conv1x1_pair")
        else:
            capi.check(lib.rtpose_conv2d(d1, 2, n, h, w, stream), "rtpose_conv2d")
            capi.check(lib.rtpose_conv2d(d2, 2, n, h, w, stream), "rtpose_conv2d")
        outs = []
        for gi in range(2):
            o = torch.empty(n, couts[gi], h, w, device=cuda)
            lo = Layout.padded(192, h, w, 3, choff=128 if gi == 0 else 166)
            capi.check(lib.rtpose_layout_to_nchw(capi.ptr(cat), C.byref(lo), capi.ptr(o), couts[gi], n, h, w, stream))
            outs.append(o.cpu())
        torch.cuda.synchronize()
        inner = sum(o.abs().sum().item() for o in outs)
        assert abs(cat.abs().sum().item() - inner) <= 1e-3 * max(1.0, inner), "wrote outside the two head slices"
        results.append(outs)
    for gi in range(2):
        ref = F.conv2d(F.relu(F.conv2d(xs[gi], w1[gi], b1[gi])), w2[gi], b2[gi])
        assert (results[0][gi] - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())
        assert torch.equal(results[0][gi], results[1][gi])
    d1[0].cout = 64
    assert lib.rtpose_conv1x1_pair_fits(d1, d2, 2) == 0
