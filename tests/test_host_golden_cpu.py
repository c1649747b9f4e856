"""CPU suite: the host-side helpers either side of the hot path, pinned to the reference's OWN
functions.  tests/golden/host_ref.npz and ski_demo.npz were produced by oracle/make_golden_host.py,
which executes evaluate/coco_eval.py, lib/network/im_transform.py, lib/datasets/preprocessing.py,
lib/utils/paf_to_pose.py and demo/picture_demo.py UNMODIFIED (third-party modules that are absent
from the image are stubbed; cv2 = oracle/cv2_restate.py, the only piece that stays unpinned).

Checked here, all without a GPU:
  * the oracle restatements (oracle/host_oracle.py, oracle/post_oracle.c incl. the two optional NMS
    branches, oracle/tta_oracle.py) reproduce those reference outputs;
  * the product's own host code (preprocess.py: crop_with_factor & co, which get_outputs runs
    before the network) reproduces them bit for bit;
  * BASELINE configs[0]: the reference's picture_demo.py result on readme/ski.jpg.
"""
import importlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import PKG_NAME
from oracle import host_oracle as ho
from oracle import make_golden_host as mg
from oracle import net_oracle, post_oracle as po, tta_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ref():
    return np.load(os.path.join(GOLD, "host_ref.npz"))


@pytest.fixture(scope="module")
def ski():
    return np.load(os.path.join(GOLD, "ski_demo.npz"))


@pytest.fixture(scope="module")
def pre(pkg):
    return importlib.import_module(PKG_NAME + ".preprocess")


def test_handle_paf_and_heat_restatement_matches_reference(ref):
    for k, (h, w, seed) in enumerate(mg.FM_CASES):
        heat, heat_f, paf, paf_f = mg.fm_inputs(h, w, seed)
        keep = paf_f.copy()
        avg_paf, avg_heat = ho.handle_paf_and_heat(heat, heat_f, paf, paf_f)
        assert np.array_equal(avg_paf, ref["fm%d_paf" % k]) and np.array_equal(avg_heat, ref["fm%d_heat" % k])
        assert np.array_equal(paf_f, keep)          # (the reference negates the caller's array in place)


@pytest.mark.parametrize("who", ["oracle", "product"])
def test_crop_with_factor_and_preprocess_match_reference(ref, pre, who):
    m = ho if who == "oracle" else pre
    for k, (h0, w0, dest, factor, seed, keep) in enumerate(mg.CW_CASES):
        img = mg.cw_input(h0, w0, seed)
        crop, scale, real = m.crop_with_factor(img, dest, factor=factor, is_ceil=True)
        assert scale == float(ref["cw%d_scale" % k]) and tuple(real) == tuple(ref["cw%d_real" % k])
        assert crop.dtype == np.uint8 and np.array_equal(crop, ref["cw%d_crop" % k]), (who, k)
        if keep:
            assert np.array_equal(m.rtpose_preprocess(crop), ref["cw%d_rtpose" % k])
            assert np.array_equal(m.vgg_preprocess(crop), ref["cw%d_vgg" % k])


def test_gaussian_filter_restatement_is_scipy_bit_for_bit():
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(0)
    for shp in ((40, 40), (24, 40), (40, 24), (24, 24), (5, 7)):   # the patch sizes NMS can produce, and a tiny one
        a = rng.normal(size=shp).astype(np.float32)
        assert np.array_equal(gaussian_filter(a, sigma=3), po.gaussian_filter_f32(a, 3.0))


def test_nms_oracle_all_flag_settings_match_reference_nms(ref):
    for k, (hh, ww, npeople, seed) in enumerate(mg.NMS_SCENES):
        heat, _ = mg.scene(hh, ww, npeople, seed)
        n = {}
        for tag in ("default", "gauss", "norefine"):
            want = ref["nms%d_%s" % (k, tag)]
            got = po.nms(heat, refine=tag != "norefine", gaussian=tag == "gauss")
            assert np.array_equal(got, want.astype(np.float32)), (k, tag)
            n[tag] = want
        # the optional branches really change something: blurred arg-max / grid-snapped centres
        assert not np.array_equal(n["default"][:, :3], n["gauss"][:, :3])
        assert np.all((n["norefine"][:, 0] + 0.5) % 8 == 4.0)           # (c + 0.5) * 8 - 0.5


def test_paf_to_pose_oracle_matches_reference_paf_to_pose_cpp(ref):
    for k, (hh, ww, npeople, seed) in enumerate(mg.P2P_SCENES):
        heat, paf = mg.scene(hh, ww, npeople, seed)
        jl, r = po.paf_to_pose(heat, paf)
        want, want_score = ref["p2p%d_parts" % k], ref["p2p%d_score" % k]
        assert len(r["parts"]) == len(want) >= 2
        for hid in range(len(want)):
            for p in range(18):
                cid = r["parts"][hid, p]
                if cid < 0:
                    assert np.isnan(want[hid, p, 0])
                else:   # paf_to_pose.py:396-399: x / W_up, y / H_up, score
                    assert tuple(want[hid, p]) == (float(int(jl[cid, 0])) / ww, float(int(jl[cid, 1])) / hh,
                                                   float(jl[cid, 2]))
            assert np.float32(want_score[hid]) == r["score"][hid]


def test_tta_restatement_matches_reference_composition(ref):
    """oracle/tta_oracle.py over the RESTATED helpers + net oracle == the same composition over the
    reference's functions and module (the golden)."""
    sd = None

    def forward(x):
        nonlocal sd
        if sd is None:
            pkg = importlib.import_module(PKG_NAME)
            sd = net_oracle.he_init_state_dict(pkg.get_model('vgg19'), seed=0)
        (paf, heat), _ = net_oracle.forward(sd, torch.from_numpy(x))
        return paf[0].permute(1, 2, 0).numpy(), heat[0].permute(1, 2, 0).numpy()
    case = tta_oracle.TTA_CASE
    paf, heat, s1 = tta_oracle.multiscale(tta_oracle.tta_image(), forward, case["preprocess"], case["scales"], True)
    assert s1 == float(ref["tta_s1"])
    assert np.abs(paf - ref["tta_flip_paf"]).max() <= 1e-5 and np.abs(heat - ref["tta_flip_heat"]).max() <= 1e-5


def test_seeded_weight_recipes_agree(pkg):
    synth = importlib.import_module(PKG_NAME + ".synth")
    m = pkg.get_model('vgg19')
    a, b = synth.he_init_state_dict(m, 3), net_oracle.he_init_state_dict(m, 3)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    from oracle import shufflenet_oracle as so
    sn = importlib.import_module(PKG_NAME + ".shufflenet").Network(1.0)
    a, b = synth.seeded_shufflenet_state_dict(sn, 1), so.seeded_state_dict(sn, 1)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)


# ---- BASELINE configs[0]: demo/picture_demo.py on readme/ski.jpg ----------------------------
def test_ski_fixture_geometry_and_host_prep(ski, pre):
    img = ski["ski_bgr"]
    assert img.shape == (674, 712, 3) and img.dtype == np.uint8
    crop, scale, real = pre.crop_with_factor(img, 368, factor=8, is_ceil=True)       # coco_eval.py:87-91
    assert scale == float(ski["im_scale"]) == 368.0 / 674
    assert tuple(real) == (368, 389, 3) and crop.shape == (368, 392, 3)
    assert np.array_equal(crop, ski["crop"])
    assert ski["paf"].shape == (46, 49, 38) and ski["heatmap"].shape == (46, 49, 19)
    meta = json.loads(str(ski["meta"]))
    assert meta["timing_build_container_s"]["get_outputs_s"] > 0


def test_ski_oracles_reproduce_the_reference_run(ski, pkg):
    """net oracle on the reference's cropped image == the maps picture_demo.py computed; the post
    oracle on those maps == the Humans it built."""
    sd = net_oracle.he_init_state_dict(pkg.get_model('vgg19'), seed=0)
    x = torch.from_numpy(ho.rtpose_preprocess(ski["crop"])[None])
    (paf, heat), _ = net_oracle.forward(sd, x)
    assert np.abs(paf[0].permute(1, 2, 0).numpy() - ski["paf"]).max() <= 1e-5
    assert np.abs(heat[0].permute(1, 2, 0).numpy() - ski["heatmap"]).max() <= 1e-5
    # a random network's maps are junk (1575 peaks, several refining to the same pixel): exactly tied
    # candidate scores, which the reference's unstable std::sort orders its own way - replayed here
    jl = po.nms(ski["heatmap"])
    assert len(jl) > 1000
    r = po.process_paf(jl, ski["paf"], 8, libstdcxx_sort=True)
    want = ski["parts"]
    assert len(r["parts"]) == len(want) >= 1
    for hid in range(len(want)):
        for p in range(18):
            cid = r["parts"][hid, p]
            if cid < 0:
                assert np.isnan(want[hid, p, 0])
            else:
                assert tuple(want[hid, p]) == (float(int(jl[cid, 0])) / (49 * 8), float(int(jl[cid, 1])) / (46 * 8),
                                               float(jl[cid, 2]))
        assert np.float32(ski["score"][hid]) == r["score"][hid]


def test_append_result_matches_reference(pkg, pre):
    """evaluate/coco_eval.py:117-154 executed unmodified (oracle/make_golden_append.py -> tests/golden/
    append_result.json) against the product's append_result on the same seeded humans: image_id, category_id,
    the hard-coded score 1.0 and all 51 keypoint values (ORDER_COCO order, + 0.5, visibility 1) exactly."""
    from oracle import make_golden_append as mga
    common = importlib.import_module(PKG_NAME + ".common")
    with open(os.path.join(GOLD, "append_result.json")) as f:
        gold = json.load(f)
    assert len(gold) == len(mga.CASES)
    for (seed, n, p, up, image_id), want in zip(mga.CASES, gold):
        outputs = []
        pre.append_result(image_id, mga.build(mga.people(seed, n, p), common.Human, common.BodyPart), up, outputs)
        assert len(outputs) == len(want) == n
        for a, b in zip(outputs, want):
            assert a["image_id"] == b["image_id"] == image_id and a["category_id"] == b["category_id"] == 1
            assert a["score"] == b["score"] == 1.0
            assert [float(v) for v in a["keypoints"]] == b["keypoints"]
