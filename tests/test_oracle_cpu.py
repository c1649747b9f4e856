"""CPU suite: the oracle is pinned to the reference before anything trusts it.

* oracle_process_paf (plain C) == golden vectors produced by the reference's own
  pafprocess.cpp compiled unmodified, and == that compiled reference directly on
  many random scenes when oracle/_ref is present (it is git-ignored but travels).
* oracle_nms peak positions == scipy.ndimage.maximum_filter, the call the reference
  makes (lib/utils/paf_to_pose.py:34); its bicubic refine == torch CPU bicubic
  (same A=-0.75 / half-pixel / clamped-tap definition) to ~1e-6 with equal arg-max.
* net oracle == golden outputs of the reference module (tests/golden/net_small.npz).
"""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import PKG_NAME
from oracle import net_oracle
from oracle import post_oracle as po

GOLD_POST = os.path.join(os.path.dirname(__file__), "golden", "post_scenes.npz")
GOLD_NET = os.path.join(os.path.dirname(__file__), "golden", "net_small.npz")


@pytest.fixture(scope="module")
def synth(pkg):
    return importlib.import_module(PKG_NAME + ".synth")


def test_process_paf_oracle_matches_reference_golden():
    z = np.load(GOLD_POST)
    assert int(z["n"]) >= 5
    for i in range(int(z["n"])):
        r = po.process_paf(z["jl%d" % i], z["paf%d" % i], 8)
        assert np.array_equal(r["parts"], z["parts%d" % i])
        assert np.array_equal(r["score"].view(np.uint32), z["score%d" % i].view(np.uint32))
        assert np.array_equal(np.stack([r["line_x"], r["line_y"]], 1), z["line%d" % i])


def test_nms_oracle_reproduces_golden_joint_lists():
    z = np.load(GOLD_POST)
    for i in range(int(z["n"])):
        jl = po.nms(z["heat%d" % i])
        assert np.array_equal(jl.view(np.uint32), z["jl%d" % i].view(np.uint32))


@pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_process_paf_oracle_vs_compiled_reference_random(synth):
    rng = np.random.default_rng(42)
    hits = ties = tie_free = 0
    for trial in range(40):
        hh, ww = int(rng.integers(12, 47)) * 8, int(rng.integers(12, 50)) * 8
        people = synth.random_people(rng, int(rng.integers(1, 12)), hh, ww, drop_prob=float(rng.uniform(0, 0.3)))
        heat, paf = synth.render(people, hh, ww, noise=float(rng.uniform(0.005, 0.08)), rng=rng)
        jl = po.nms(heat)
        if len(jl) == 0:
            continue
        ref = po.ref_process_paf(jl, po.upsample_nearest(heat, 8), po.upsample_nearest(paf, 8))
        # Two peaks that refine to the same pixel give exactly equal candidate scores; the
        # reference's std::sort (not stable) then decides.  libstdcxx_sort=True replays that
        # sort, so the whole restatement is compared even on such scenes ...
        mine = po.process_paf(jl, paf, 8, libstdcxx_sort=True)
        assert np.array_equal(ref["line_x"], mine["line_x"]) and np.array_equal(ref["line_y"], mine["line_y"])
        assert np.array_equal(ref["parts"], mine["parts"]), trial
        assert np.array_equal(ref["score"].view(np.uint32), mine["score"].view(np.uint32)), trial
        hits += len(ref["parts"])
        # ... and the stable mode (ties -> lower (idx1, idx2), what the GPU kernel's arg-max loop
        # does until it meets a tie and replays the sort) must agree with it whenever there is no tie
        contract = po.process_paf(jl, paf, 8, libstdcxx_sort=False)
        ties += contract["had_ties"]
        if not contract["had_ties"]:
            assert np.array_equal(ref["parts"], contract["parts"]), trial
            assert np.array_equal(ref["score"].view(np.uint32), contract["score"].view(np.uint32)), trial
            tie_free += 1
    assert hits > 150 and tie_free >= 3


@pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_process_paf_oracle_vs_compiled_reference_junk_maps():
    """Pure-noise maps: hundreds of peaks, long candidate lists, found==2 merges."""
    rng = np.random.default_rng(7)
    compared = 0
    for trial in range(6):
        h, w = 20 + trial, 26 - trial
        heat = rng.uniform(0, 0.35, (h, w, 19)).astype(np.float32)
        paf = rng.uniform(-0.2, 1.0, (h, w, 38)).astype(np.float32)
        jl = po.nms(heat)
        ref = po.ref_process_paf(jl, po.upsample_nearest(heat, 8), po.upsample_nearest(paf, 8))
        mine = po.process_paf(jl, paf, 8, libstdcxx_sort=True)   # long lists: introsort path
        assert np.array_equal(ref["parts"], mine["parts"])
        assert np.array_equal(ref["score"].view(np.uint32), mine["score"].view(np.uint32))
        compared += 1
    assert compared >= 4


def test_nms_peaks_match_scipy_find_peaks(synth):
    heat, _, _ = synth.make_batch(3, 184, 216, seed=21)
    for b in range(3):
        jl = po.nms(heat[b])
        for j in range(18):
            exp = po.find_peaks_scipy(np.float32(0.1), heat[b][:, :, j])      # [[x, y], ...] row-major
            mine = jl[jl[:, 4] == j]
            assert len(exp) == len(mine)
            # refined coords stay inside the 5x5 window of the low-res peak, in the same order
            assert np.all(np.abs(mine[:, 0] - (exp[:, 0] * 8 + 3.5)) <= 2 * 8 + 4)
            assert np.all(np.abs(mine[:, 1] - (exp[:, 1] * 8 + 3.5)) <= 2 * 8 + 4)
    ids = po.nms(heat[0])[:, 3]
    assert np.array_equal(ids, np.arange(len(ids), dtype=np.float32))        # running counter :141-142


def test_nms_plateau_border_threshold_semantics():
    h = np.zeros((9, 9, 19), np.float32)
    h[0, 0, 0] = 0.5                      # corner accepted (reflect border)
    h[4, 4, 1] = h[4, 5, 1] = 0.4         # plateau: both pixels are peaks
    h[2, 2, 2] = np.float32(0.1)          # == threshold in float32: rejected
    h[6, 6, 3] = 0.3
    h[7, 7, 3] = 0.9                      # diagonal neighbour does not suppress (4-connectivity)
    jl = po.nms(h)
    assert [int(v) for v in jl[:, 4]] == [0, 1, 1, 3, 3]
    for j in range(4):
        assert len(po.find_peaks_scipy(np.float32(0.1), h[:, :, j])) == int((jl[:, 4] == j).sum())


def test_refine_matches_torch_bicubic():
    rng = np.random.default_rng(5)
    for _ in range(60):
        ph, pw = int(rng.integers(3, 6)), int(rng.integers(3, 6))
        heat = np.zeros((ph, pw, 19), np.float32)
        patch = rng.uniform(0.0, 0.09, (ph, pw)).astype(np.float32)
        cy, cx = min(2, ph - 1), min(2, pw - 1)
        patch[cy, cx] = 0.8
        heat[:, :, 0] = patch
        jl = po.nms(heat)
        assert len(jl) == 1
        # the whole map is the (clipped) window only if it is <= 5 wide: compare on that window
        x_min, y_min = max(0, cx - 2), max(0, cy - 2)
        win = patch[y_min:cy + 3, x_min:cx + 3]
        up = torch.nn.functional.interpolate(torch.from_numpy(win)[None, None], scale_factor=8, mode='bicubic',
                                             align_corners=False)[0, 0].numpy()
        r, c = np.unravel_index(up.argmax(), up.shape)
        assert (jl[0, 0], jl[0, 1]) == (x_min * 8 + c, y_min * 8 + r)
        assert abs(jl[0, 2] - up.max()) <= 2e-6


def test_net_oracle_reproduces_reference_golden(pkg):
    z = np.load(GOLD_NET)
    m = pkg.get_model('vgg19')
    sd = net_oracle.he_init_state_dict(m, seed=0)
    (paf, heat), saved = net_oracle.forward(sd, torch.from_numpy(z["x"]))
    # same ATen CPU kernels, same weights -> agreement to rounding (thread-count dependent order)
    assert np.abs(paf.numpy() - z["paf"]).max() <= 1e-5
    assert np.abs(heat.numpy() - z["heat"]).max() <= 1e-5
    for i in range(12):
        assert np.abs(saved[i].numpy() - z["saved%d" % i]).max() <= 1e-5


def test_winograd_tables_of_the_kernels_are_the_toom_cook_construction():
    """The input-transform tables hard-coded in csrc/conv_wino7.hip (struct WT<4>, WT<6>) equal the exact Toom-Cook
    construction (oracle/winograd_tables.py) to fp32 rounding, and that construction reproduces a 7-tap correlation
    exactly (rational arithmetic); the 3x3 kernel's F(2,3) matrices likewise."""
    import re
    from fractions import Fraction as Fr
    from oracle import winograd_tables as wt
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "pytorch_realtime_multi-person_pose_estimation_amd", "csrc", "conv_wino7.hip")).read()
    for m, pts in ((4, wt.POINTS_F4_7), (6, wt.POINTS_F6_7)):
        n = m + 6
        AT, G, BT = wt.toom_cook(m, 7, pts)
        # exactness of the algorithm itself
        d = [Fr(3 * i * i - 7 * i + 1, 5) for i in range(n)]
        g = [Fr(2 * k - 5, 3) for k in range(7)]
        U = [sum(G[f][k] * g[k] for k in range(7)) for f in range(n)]
        V = [sum(BT[f][i] * d[i] for i in range(n)) for f in range(n)]
        for i in range(m):
            assert sum(AT[i][f] * U[f] * V[f] for f in range(n)) == sum(d[i + k] * g[k] for k in range(7))
        # the table in the kernel source
        blk = src[src.index("struct WT<%d> {" % m):]
        blk = blk[blk.index("kBT[%d][%d] = {" % (n, n)):]
        blk = blk[:blk.index("};")]
        vals = [float(x) for x in re.findall(r"(-?\d+\.?\d*)f", blk.split("= {", 1)[1])]
        assert len(vals) == n * n
        for f in range(n):
            for i in range(n):
                assert abs(vals[f * n + i] - float(BT[f][i])) <= 1e-7 * max(1.0, abs(float(BT[f][i]))), (m, f, i)
        # output transform used by the epilogue: AT[i][2p+1] = +p^i, AT[i][2p+2] = (-p)^i, bias column = point 1
        assert all(AT[i][1] == 1 for i in range(m))
    # F(2,3) as the 3x3 kernel uses it (csrc/conv_wino.hip: rows of B^T, G, A^T written out in the code): the same
    # construction with rows rescaled by +-1 / 2; exact
    BT = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
    G = [[1, 0, 0], [Fr(1, 2), Fr(1, 2), Fr(1, 2)], [Fr(1, 2), Fr(-1, 2), Fr(1, 2)], [0, 0, 1]]
    AT = [[1, 1, 1, 0], [0, 1, -1, -1]]
    d = [Fr(7, 3), Fr(-2), Fr(5, 7), Fr(11, 2)]
    g = [Fr(3, 4), Fr(-5, 3), Fr(2)]
    for i in range(2):
        assert sum(AT[i][f] * sum(G[f][k] * g[k] for k in range(3)) * sum(BT[f][j] * d[j] for j in range(4))
                   for f in range(4)) == sum(d[i + k] * g[k] for k in range(3))
    ATc, Gc, BTc = wt.toom_cook(2, 3, wt.POINTS_F2_3)
    for f in range(4):   # row f of the construction = a multiple of the kernel's row (G / A^T carry the inverse factor)
        nz = [j for j in range(4) if BT[f][j] != 0][0]
        fac = BTc[f][nz] / BT[f][nz]
        assert [x * fac for x in BT[f]] == BTc[f]


def test_f43_constants_of_the_kernel_are_the_toom_cook_construction():
    """F(4x4,3x3) (csrc/conv_wino4.hip, points 0, +-3/4, +-3/2, inf): the construction is exact, every entry of B^T and
    A^T is a dyadic fraction (exactly representable in fp32), and the constants written out in the kernel's bt6 /
    at4_lo / at4_hi / g43 are those entries."""
    import re
    from fractions import Fraction as Fr
    from oracle import winograd_tables as wt
    AT, G, BT = wt.toom_cook(4, 3, wt.POINTS_F4_3)
    d = [Fr(3 * i * i - 7 * i + 1, 5) for i in range(6)]
    g = [Fr(3, 4), Fr(-5, 3), Fr(2)]
    for i in range(4):
        assert sum(AT[i][f] * sum(G[f][k] * g[k] for k in range(3)) * sum(BT[f][j] * d[j] for j in range(6))
                   for f in range(6)) == sum(d[i + k] * g[k] for k in range(3))
    for M in (AT, BT):
        for row in M:
            for v in row:
                assert v.denominator & (v.denominator - 1) == 0 and float(v) == float(np.float32(float(v)))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "pytorch_realtime_multi-person_pose_estimation_amd", "csrc", "conv_wino4.hip")).read()
    bt6 = src[src.index("__device__ __forceinline__ void bt6("):src.index("// A^T of F(4,3) along one axis")]
    consts = sorted({abs(float(x)) for x in re.findall(r"(-?\d+\.\d+)f", bt6)})
    # rows 0 / 5: 81/64, -45/16, 1;  +-3/4: E = d4 - 9/4 d2, O = 3/4 (d3 - 9/4 d1);  +-3/2: E = d4 - 9/16 d2, O = 3/2 (d3 - 9/16 d1)
    assert consts == sorted([81 / 64, 45 / 16, 9 / 4, 3 / 4, 9 / 16, 3 / 2])
    assert BT[0] == [Fr(81, 64), 0, Fr(-45, 16), 0, 1, 0] and BT[5] == [0, Fr(81, 64), 0, Fr(-45, 16), 0, 1]
    assert BT[1] == [0, Fr(-27, 16), Fr(-9, 4), Fr(3, 4), 1, 0] and BT[2] == [0, Fr(27, 16), Fr(-9, 4), Fr(-3, 4), 1, 0]
    assert BT[3] == [0, Fr(-27, 32), Fr(-9, 16), Fr(3, 2), 1, 0] and BT[4] == [0, Fr(27, 32), Fr(-9, 16), Fr(-3, 2), 1, 0]
    assert Fr(3, 4) * Fr(9, 4) == Fr(27, 16) and Fr(3, 2) * Fr(9, 16) == Fr(27, 32)
    at = src[src.index("__device__ __forceinline__ void at4_lo("):src.index("__global__ __launch_bounds__(512, 1) void wino4_f32")]
    consts = sorted({float(x) for x in re.findall(r"splat\((\d+\.\d+)f\)", at)})
    assert consts == sorted([3 / 4, 3 / 2, 9 / 16, 9 / 4, 27 / 64, 27 / 8])
    assert AT == [[1, 1, 1, 1, 1, 0], [0, Fr(3, 4), Fr(-3, 4), Fr(3, 2), Fr(-3, 2), 0],
                  [0, Fr(9, 16), Fr(9, 16), Fr(9, 4), Fr(9, 4), 0], [0, Fr(27, 64), Fr(-27, 64), Fr(27, 8), Fr(-27, 8), 1]]
    gsrc = src[src.index("const double G[6][3] = {"):]
    gsrc = gsrc[:gsrc.index("};")]
    vals = [Fr(int(a), int(b)) if b else Fr(int(float(a))) for a, b in
            re.findall(r"(-?\d+)\.0(?: / (\d+)\.0)?", gsrc.split("= {", 1)[1])]
    assert vals == [v for row in G for v in row]


def test_f43_restatement_stays_inside_its_error_bound():
    """oracle/winograd43_oracle.py (numpy float32 restatement of the F(4x4,3x3) form, exact Toom-Cook matrices) against
    the float64 direct sum: |err| <= gamma 2^-24 sum |x||w| with the limit the GPU test applies to the kernel
    (tests/test_wino_numerics_gpu.py: 100), on zero-mean and on all-positive data, map sizes that are and are not
    multiples of 4."""
    from oracle.winograd43_oracle import conv3x3_f43
    rng = np.random.default_rng(7)
    for (cin, cout, H, W, positive) in ((48, 5, 9, 14, False), (32, 4, 8, 8, True)):
        x = rng.standard_normal((cin, H, W)).astype(np.float32)
        w = (rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        if positive:
            x, w = np.abs(x) * 50 + 100, np.abs(w)
        y = conv3x3_f43(x, w, b)
        xt, wt = torch.from_numpy(x).double()[None], torch.from_numpy(w).double()
        ref = torch.nn.functional.conv2d(xt, wt, torch.from_numpy(b).double(), padding=1)[0].numpy()
        S = torch.nn.functional.conv2d(xt.abs(), wt.abs(), torch.from_numpy(b).double().abs(), padding=1)[0].numpy()
        gamma = (np.abs(y.astype(np.float64) - ref) / (2.0 ** -24 * S)).max()
        assert gamma <= 100.0, (cin, H, W, positive, gamma)
        assert np.abs(y - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())

