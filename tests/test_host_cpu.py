"""CPU suite: host-side logic (result parsing, drop-in types, state_dict surface,
sharding + gather over gloo with world_size 2, loud failure without a GPU)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import PKG_NAME, ROOT


@pytest.fixture(scope="module")
def dec(pkg):
    return importlib.import_module(PKG_NAME + ".decode")


def _pack_record(dec, cfg, jl, parts, score):
    """Build a result record the way the device kernels lay it out (test helper)."""
    from oracle import post_oracle  # noqa: F401  (tests may use the oracle)
    pcap, hcap = cfg.max_peaks_per_part, cfg.max_humans
    words = (32 + 72 * pcap + 19 * hcap + 3) & ~3
    rec = np.zeros(words, np.int32)
    rec[0] = len(jl)
    rec[1] = len(parts)
    pk = rec[32:32 + 72 * pcap].reshape(18, pcap, 4)
    for p in range(18):
        rows = jl[jl[:, 4] == p]
        rec[8 + p] = len(rows)
        pk[p, :len(rows), 0] = rows[:, 0].astype(np.int32)
        pk[p, :len(rows), 1] = rows[:, 1].astype(np.int32)
        pk[p, :len(rows), 2] = rows[:, 2].copy().view(np.int32)
        pk[p, :len(rows), 3] = rows[:, 3].astype(np.int32)
    off = 32 + 72 * pcap
    rec[off:off + 18 * hcap].reshape(hcap, 18)[:len(parts)] = parts
    rec[off + 18 * hcap:off + 18 * hcap + len(score)] = score.view(np.int32)
    return rec


def test_parse_record_roundtrip_and_humans(dec):
    z = np.load(os.path.join(ROOT, "tests", "golden", "post_scenes.npz"))
    cfg = dec.make_cfg(None, 32, 16)
    rec = _pack_record(dec, cfg, z["jl3"], z["parts3"], z["score3"])
    out = dec.parse_image(rec, cfg)
    assert np.array_equal(out["peaks"], z["jl3"]) and np.array_equal(out["parts"], z["parts3"])
    assert np.array_equal(out["score"], z["score3"]) and out["flags"] == 0
    humans = dec.humans_from_record(out, 368, 368)
    assert len(humans) == len(z["parts3"])
    h0 = humans[0]
    assert h0.part_count() == int((z["parts3"][0] >= 0).sum())
    for p, bp in h0.body_parts.items():
        assert 0.0 <= bp.x < 1.0 and 0.0 <= bp.y < 1.0 and bp.part_idx == p
        assert bp.get_part_name().value == p
    assert h0.get_max_score() == max(b.score for b in h0.body_parts.values())
    assert "BodyPart:" in str(h0)


def test_result_mask_covers_what_a_record_says_and_nothing_else(dec):
    """decode.result_mask: the words of a record block that carry results.  bench.py and the GPU soak compare blocks through
    it (scratch behind the counts differs from run to run): any change of a meaningful word must show, scratch must not."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "post_scenes.npz"))
    cfg = dec.make_cfg(None, 32, 16)
    rec = _pack_record(dec, cfg, z["jl3"], z["parts3"], z["score3"])
    rec[3], rec[4] = 32, 16                       # the capacities a device record carries in its header
    blk = np.stack([rec, rec])
    m = dec.result_mask(blk)
    nh, npk = len(z["parts3"]), len(z["jl3"])
    assert m.shape == blk.shape and m[0].sum() == 5 + 18 + 4 * npk + 18 * nh + nh
    rng = np.random.default_rng(0)
    junk = blk.copy()
    junk[~m] = rng.integers(-2 ** 31, 2 ** 31 - 1, (~m).sum())             # scratch behind the counts
    assert np.array_equal(junk[m], blk[m])
    parsed = dec.parse_image(junk[1])
    assert np.array_equal(parsed["peaks"], z["jl3"]) and np.array_equal(parsed["score"], z["score3"])
    for w in np.flatnonzero(m[0])[::7]:                                    # any meaningful word
        other = blk.copy()
        other[0, w] ^= 1
        assert not np.array_equal(other[m], blk[m])
    more = blk.copy()
    more[1, 1] += 1                                                        # a count that grew is seen through either mask
    assert not np.array_equal(more[m], blk[m]) and not np.array_equal(more[dec.result_mask(more)], blk[dec.result_mask(more)])


def test_state_dict_surface_matches_reference_names(pkg):
    m = pkg.get_model('vgg19')
    sd = m.state_dict()
    keys = list(sd)
    assert len(keys) == 184
    assert keys[0] == "model0.0.weight" and keys[1] == "model0.0.bias" and keys[24] == "model1_1.0.weight"
    assert keys[-1] == "model6_2.12.bias"
    assert tuple(sd["model2_1.0.weight"].shape) == (128, 185, 7, 7)
    assert tuple(sd["model1_2.8.weight"].shape) == (19, 512, 1, 1)
    # reference init (:200-222): N(0, 0.01) weights, zero biases
    assert float(sd["model0.2.bias"].abs().max()) == 0.0
    assert 0.005 < float(sd["model0.21.weight"].std()) < 0.02
    with pytest.raises(ValueError):
        pkg.get_model('mobilenet')
    # nn.DataParallel wrapping + load_state_dict keep working (demo/picture_demo.py:45-47)
    dp = torch.nn.DataParallel(m)
    dp.module.load_state_dict({k: v.clone() for k, v in sd.items()})


def test_no_silent_cpu_fallback(pkg, dec, capi):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = pkg.get_model('vgg19')
    with pytest.raises(capi.RtposeError):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(capi.RtposeError):
        dec.paf_to_pose_cpp(np.zeros((8, 8, 19), np.float32), np.zeros((8, 8, 38), np.float32),
                            dec.default_config())
    with pytest.raises(capi.RtposeError):
        dec.decode_maps(torch.zeros(1, 8, 8, 19), torch.zeros(1, 8, 8, 38))
    pafprocess = importlib.import_module(PKG_NAME + ".pafprocess")
    with pytest.raises(TypeError):
        pafprocess.process_paf(np.zeros((4, 5), np.float32), np.zeros((8, 8, 19)), np.zeros((8, 8, 38)))
    with pytest.raises(capi.RtposeError):   # one peak, no device: loud, no CPU path
        pafprocess.process_paf(np.array([[[1, 1, .5, 0, 0]]], np.float32), np.zeros((8, 8, 19)), np.zeros((8, 8, 38)))


def test_shard_range_partitions(pkg):
    par = importlib.import_module(PKG_NAME + ".parallel")
    for n in (0, 1, 7, 32, 5000):
        for world in (1, 2, 3, 8):
            spans = [par.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import importlib, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
PKG = "pytorch_realtime_multi-person_pose_estimation_amd"
par = importlib.import_module(PKG + ".parallel"); dec = importlib.import_module(PKG + ".decode")
synth = importlib.import_module(PKG + ".synth")
from oracle import post_oracle as po
from test_host_cpu import _pack_record
rank, _, world = par.init_from_env("gloo")
n_total = 6
heat, paf, _ = synth.make_batch(n_total, 176, 176, seed=77)
lo, hi = par.shard_range(n_total, rank, world)
cfg = dec.make_cfg(None, 32, 16)
local = []
for i in range(lo, hi):                      # stand-in for the GPU decoder on this rank's shard
    jl, r = po.paf_to_pose(heat[i], paf[i])
    local.append(_pack_record(dec, cfg, jl, r["parts"], r["score"]))
local = torch.from_numpy(np.stack(local))
allrec = par.gather_records(local, world).numpy()
t = par.max_over_ranks(float(rank + 1), torch.device("cpu"))
par.barrier()
if rank == 0:
    assert t == float(world)
    assert allrec.shape[0] == n_total
    for i in range(n_total):                 # rank order == image order
        jl, r = po.paf_to_pose(heat[i], paf[i])
        out = dec.parse_image(allrec[i], cfg)
        assert np.array_equal(out["peaks"], jl) and np.array_equal(out["parts"], r["parts"])
    print("GATHER_OK")
dist.destroy_process_group()
'''


def test_batch_schedule_covers_every_item_once_with_equal_collective_counts(pkg):
    par = importlib.import_module(PKG_NAME + ".parallel")
    for n_items, world, batch in ((5000, 8, 32), (5000, 1, 32), (7, 4, 2), (64, 2, 32), (1, 3, 4)):
        seen = []
        lens = set()
        for r in range(world):
            sched = par.batch_schedule(n_items, r, world, batch)
            lens.add(len(sched))
            lo, hi = par.shard_range(n_items, r, world)
            for i0, nv in sched:
                assert 0 <= nv <= batch
                seen += list(range(i0, i0 + nv))
                assert nv == 0 or (lo <= i0 and i0 + nv <= hi)
        assert len(lens) == 1                      # same number of all_gathers on every rank
        assert sorted(seen) == list(range(n_items))


def test_two_rank_shard_and_gather_gloo(tmp_path):
    """world_size 2 on CPU (gloo): shard, decode shard (oracle stand-in), all_gather, parse."""
    import subprocess
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = 29600 + (os.getpid() % 300)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "GATHER_OK" in p.stdout, p.stdout[-3000:]


def test_bench_started_plain_with_gpus_2_runs_two_ranks_and_says_so():
    """north_star: throughput "reported at 1, 2, 4 and 8 GPUs".  `python bench.py --gpus 2` started WITHOUT a launcher
    re-executes itself through torch.distributed.run with two ranks (the driver's own launch line) and relays rank 0's
    ONE line; `ranks_seen` is the size of the process group the timed region ran in.  The GPU step is replaced by
    bench.py's stub (--stub-step: gloo on CPU ranks), everything else - relaunch, barrier, max-over-ranks timing, one
    all_gather of the record block per step - is the production skeleton.  A WORLD_SIZE that disagrees with --gpus is
    an error, not a mislabelled measurement."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    bench = os.path.join(ROOT, "bench.py")
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-step"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 3 and out["data"] == "stub"
    assert out["records_from_ranks"] == [0, 1] and out["gathered_block_ok"] and len(out["per_rank_s"]) == 2
    assert out["value"] is None                                   # a stub line can never pass for a measurement
    # a launcher world that is not --gpus: refused
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29411")
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--stub-step"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, env=env2)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and not p.stdout.strip()


def test_resize_linear_restatement_close_to_float_bilinear(pkg):
    """cv2.resize INTER_LINEAR (uint8, fixed point) restatement: identity at scale 1, within
    1 LSB of a float half-pixel bilinear elsewhere, and the ski.jpg geometry of SURVEY §3.1."""
    pre = importlib.import_module(PKG_NAME + ".preprocess")
    rng = np.random.default_rng(0)
    im = rng.integers(0, 256, (67, 71, 3), dtype=np.uint8)
    assert np.array_equal(pre.resize_linear_u8(im, 1.0, 1.0), im)
    s = 368.0 / 674
    big = rng.integers(0, 256, (674, 712, 3), dtype=np.uint8)
    out = pre.resize_linear_u8(big, s, s)
    assert out.shape == (368, 389, 3)
    t = torch.from_numpy(big).permute(2, 0, 1)[None].float()
    ys = (np.arange(368) + 0.5) / s - 0.5
    xs = (np.arange(389) + 0.5) / s - 0.5
    y0 = np.clip(np.floor(ys).astype(int), 0, 673); x0 = np.clip(np.floor(xs).astype(int), 0, 711)
    y1 = np.minimum(y0 + 1, 673); x1 = np.minimum(x0 + 1, 711)
    fy = np.clip(ys - np.floor(ys), 0, 1)[:, None, None]; fx = np.clip(xs - np.floor(xs), 0, 1)[None, :, None]
    fy[ys < 0] = 0; fx[:, xs < 0] = 0
    b = big.astype(np.float64)
    ref = (b[y0][:, x0] * (1 - fx) + b[y0][:, x1] * fx) * (1 - fy) + (b[y1][:, x0] * (1 - fx) + b[y1][:, x1] * fx) * fy
    assert np.abs(out.astype(np.float64) - ref).max() <= 1.0
    cropped, scale, shape = pre.crop_with_factor(big, 368, factor=8, is_ceil=True)
    assert cropped.shape == (368, 392, 3) and shape == (368, 389, 3) and abs(scale - s) < 1e-15
    assert not cropped[:, 389:].any()
    x = pre.rtpose_preprocess(cropped)
    assert x.shape == (3, 368, 392) and x.dtype == np.float32 and -0.5 <= x.min() and x.max() < 0.5


def _fake_person(rng, img_id, ann_id, cx, cy, size):
    kp = np.zeros((17, 3))
    kp[:, 0] = cx + rng.uniform(-0.3, 0.3, 17) * size
    kp[:, 1] = cy + rng.uniform(-0.5, 0.5, 17) * size
    kp[:, 2] = 2
    kp[rng.uniform(size=17) < 0.2, 2] = 0
    x0, y0 = kp[:, 0].min(), kp[:, 1].min()
    bw, bh = kp[:, 0].max() - x0, kp[:, 1].max() - y0
    return {"image_id": img_id, "id": ann_id, "category_id": 1, "iscrowd": 0, "keypoints": list(kp.reshape(51)),
            "num_keypoints": int((kp[:, 2] > 0).sum()), "bbox": [x0, y0, bw, bh], "area": float(bw * bh)}


def test_oks_evaluator_properties(pkg):
    """OKS/AP stand-in for pycocotools (coco_eval.py:55-75): exact detections -> AP 1; jitter lowers
    AP monotonically; a detection on the wrong person scores ~0; crowd GT is ignored."""
    oe = importlib.import_module(PKG_NAME + ".oks_eval")
    rng = np.random.default_rng(0)
    gts, aid = [], 0
    for img in range(12):
        for _ in range(int(rng.integers(1, 4))):
            aid += 1
            gts.append(_fake_person(rng, img, aid, rng.uniform(100, 500), rng.uniform(100, 400), rng.uniform(60, 200)))

    def dets(jitter):
        out = []
        for g in gts:
            k = np.array(g["keypoints"]).reshape(17, 3).copy()
            k[:, :2] += rng.normal(0, jitter, (17, 2)) * np.sqrt(g["area"])
            k[:, 2] = 1
            out.append({"image_id": g["image_id"], "category_id": 1, "keypoints": list(k.reshape(51)), "score": 1.0})
        return out

    perfect = oe.evaluate(gts, dets(0.0))
    assert abs(perfect["AP"] - 1.0) < 1e-9 and abs(perfect["AP50"] - 1.0) < 1e-9
    a1, a2, a3 = (oe.evaluate(gts, dets(j))["AP"] for j in (0.01, 0.05, 0.2))
    assert 1.0 >= a1 > a2 > a3 >= 0.0 and a1 > 0.9 and a3 < 0.3
    # OKS of identical keypoints is 1, of far-away keypoints ~0
    g = gts[0]
    d_same = {"keypoints": g["keypoints"], "score": 1}
    far = np.array(g["keypoints"]).reshape(17, 3).copy()
    far[:, :2] += 5000
    assert abs(oe.compute_oks([g], [d_same])[0, 0] - 1.0) < 1e-12
    assert oe.compute_oks([g], [{"keypoints": list(far.reshape(51)), "score": 1}])[0, 0] < 1e-6
    # crowd ground truth neither counts as a miss nor as a false positive when matched
    crowd = dict(gts[0], iscrowd=1)
    res = oe.evaluate([crowd] + gts[1:], dets(0.0))
    assert abs(res["AP"] - 1.0) < 1e-9
    # append_result keeps the reference's COCO order and +0.5 offsets (coco_eval.py:117-154)
    pre = importlib.import_module(PKG_NAME + ".preprocess")
    common = importlib.import_module(PKG_NAME + ".common")
    h = common.Human([])
    h.body_parts[0] = common.BodyPart('0-0', 0, 0.5, 0.25, 0.9)
    h.body_parts[16] = common.BodyPart('0-16', 16, 0.1, 0.2, 0.8)
    outs = []
    pre.append_result(42, [h], (400, 600), outs)
    kp = np.array(outs[0]["keypoints"]).reshape(17, 3)
    assert outs[0]["image_id"] == 42 and outs[0]["score"] == 1.0
    assert tuple(kp[0]) == (0.5 * 600 + 0.5, 0.25 * 400 + 0.5, 1)          # nose -> COCO 0
    assert tuple(kp[4]) == (0.1 * 600 + 0.5, 0.2 * 400 + 0.5, 1)           # our 16 (REar) -> COCO index 4
    assert kp[:, 2].sum() == 2


def test_eval_schedule_buckets_and_covers_every_image_once(pkg):
    """preprocess.eval_batches: images grouped by crop_with_factor's padded size, batches <= B, every image
    exactly once; the round-robin rank assignment of run_eval_batched covers every batch exactly once."""
    import importlib
    import numpy as np
    pre = importlib.import_module(PKG_NAME + ".preprocess")
    rng = np.random.default_rng(0)
    sizes = [(int(rng.integers(200, 640)), int(rng.integers(200, 640))) for _ in range(500)]
    sched = pre.eval_batches(sizes, 368, 8, 32)
    seen = sorted(i for _, idx in sched for i in idx)
    assert seen == list(range(500))
    for key, idx in sched:
        assert 1 <= len(idx) <= 32
        for i in idx:
            assert pre.prep_geometry(sizes[i][0], sizes[i][1], 368, 8)[2] == key
    for world in (1, 2, 3, 8):
        steps = -(-len(sched) // world)
        taken = sorted(step * world + r for step in range(steps) for r in range(world) if step * world + r < len(sched))
        assert taken == list(range(len(sched)))
    # TTA buckets by the source size itself (all scales of a batch must agree)
    for key, idx in pre.eval_batches(sizes[:50], 368, 8, 4, by_source_size=True):
        assert all(sizes[i] == key for i in idx)
    # geometry == crop_with_factor's
    img = rng.integers(0, 256, (123, 211, 3), dtype=np.uint8)
    crop, scale, real = pre.crop_with_factor(img, 368, factor=8, is_ceil=True)
    s, (hr, wr), (hn, wn) = pre.prep_geometry(123, 211, 368, 8)
    assert (s, (hr, wr), (hn, wn)) == (scale, tuple(real[:2]), crop.shape[:2])


def test_native_state_is_per_device_and_invalidates(pkg, monkeypatch):
    """Plans / weight arenas are keyed by (device, dtype): a forward on a second device neither evicts nor
    re-packs the first one's (nn.DataParallel replicas share these dicts); load_state_dict, _apply and
    invalidate_weights() force a re-pack, `.data` edits are caught by invalidate_weights()."""
    import contextlib
    import importlib
    import torch
    net = importlib.import_module(PKG_NAME + ".network")
    ns = importlib.import_module(PKG_NAME + "._native_state")
    packs = []

    class FakePlan(object):
        def __init__(self, n, h, w, weights, device, dtype=0, wino=None):
            self.shape, self.dtype, self.weights = (n, h, w), dtype, weights
            self.workspace = torch.empty(1 << 20, dtype=torch.float32)
            self.handle = None

    class Dev(object):
        def __init__(self, index):
            self.type, self.index = 'cuda', index

        def __eq__(self, o):
            return isinstance(o, Dev) and o.index == self.index

        def __hash__(self):
            return self.index
    monkeypatch.setattr(net, "_Plan", FakePlan)
    monkeypatch.setattr(net, "_ShapeOnly", lambda d: type("S", (), {"device": d})())
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    monkeypatch.setattr(net.lib, "rtpose_net_create_ex", lambda *a: 0)
    monkeypatch.setattr(net.lib, "rtpose_net_weight_bytes", lambda p: 1024)
    monkeypatch.setattr(net.lib, "rtpose_net_destroy", lambda p: None)
    monkeypatch.setattr(type(net.get_model('vgg19')), "_finalize", lambda self, plan: None)
    real_zeros = torch.zeros
    monkeypatch.setattr(torch, "zeros", lambda *a, **k: real_zeros(*a, **{kk: v for kk, v in k.items() if kk != "device"}))
    m = net.get_model('vgg19')
    monkeypatch.setattr(type(m), "_sync_weights", lambda self, plan, device: (
        packs.append(device.index) if self._params_key([p for p in self.parameters()]) != self._weights_key.get(
            (device.index, plan.dtype)) or self.always_resync else None,
        self._weights_key.__setitem__((device.index, plan.dtype), self._params_key([p for p in self.parameters()]))))
    d0, d1 = Dev(0), Dev(1)
    p0 = m.plan_for_shape(2, 64, 64, d0)
    p1 = m.plan_for_shape(2, 64, 64, d1)
    assert p0 is not p1 and p0.weights is not p1.weights            # separate arenas
    assert m.plan_for_shape(2, 64, 64, d0) is p0                     # device 1 evicted nothing of device 0
    assert packs == [0, 1]                                           # one pack per device, none repeated
    m.plan_for_shape(2, 64, 64, d0)
    assert packs == [0, 1]
    next(m.parameters()).data.fill_(0.5)                             # `.data` edit: invisible to _version ...
    m.plan_for_shape(2, 64, 64, d0)
    assert packs == [0, 1]
    m.invalidate_weights()                                           # ... so the caller says so
    m.plan_for_shape(2, 64, 64, d0)
    m.plan_for_shape(2, 64, 64, d1)
    assert packs == [0, 1, 0, 1]
    m.load_state_dict(m.state_dict())
    m.plan_for_shape(2, 64, 64, d0)
    assert packs[-1] == 0 and len(packs) == 5
    m.always_resync = True
    m.plan_for_shape(2, 64, 64, d0)
    assert len(packs) == 6
    # byte-bounded plan cache, per device
    monkeypatch.setattr(ns, "MAX_WORKSPACE_BYTES_PER_DEVICE", 3 * (4 << 20) + 1)
    for k in range(5):
        m.plan_for_shape(1, 64 + 8 * k, 64, d1)
    assert sum(1 for key in m._plans if key[3] == 1) <= 3 and any(key[3] == 0 for key in m._plans)
    # plans with other rtpose_net_options are other plans, on the SAME weight arena
    m.always_resync = False
    pa = m.plan_for_shape(2, 64, 64, d0)
    pb = m.set_winograd(winograd7=4).plan_for_shape(2, 64, 64, d0)
    pc = m.set_winograd(winograd7='auto', amp_limit=100.0).plan_for_shape(2, 64, 64, d0)
    assert pa is not pb and pb is not pc and pa.weights is pb.weights is pc.weights
    assert m.set_winograd().plan_for_shape(2, 64, 64, d0) is pa
    with pytest.raises(ValueError):
        m.set_winograd(winograd7=5)
    # rtpose_net_options.winograd3: default / direct / F(2x2,3x3) / F(4x4,3x3) / per-layer AUTO
    from importlib import import_module
    capi = import_module(pkg.__name__ + "._capi")
    for arg, want in ((None, capi.WINO_DEFAULT), (False, 0), (0, 0), (True, 1), (1, 1), (2, 1), (4, 4), ('auto', capi.WINO3_AUTO)):
        assert m.set_winograd(winograd3=arg)._wino[0] == want, arg
    pd = m.set_winograd(winograd3=2).plan_for_shape(2, 64, 64, d0)
    assert pd is not pa and pd.weights is pa.weights
    for bad in (3, 1.5, 'f43'):
        with pytest.raises(ValueError):
            m.set_winograd(winograd3=bad)
    m.set_winograd()


def test_modules_can_be_deep_copied_and_pickled(pkg):
    """copy.deepcopy(model), torch.save(model) and pickle (multiprocessing spawn, EMA / SWA copies, whole-model
    checkpoints) see parameters and settings; plans, arenas and the lock are rebuilt by the copy."""
    import copy
    import io
    import pickle
    import torch
    sn = importlib.import_module(PKG_NAME + ".shufflenet")
    for m in (pkg.get_model('vgg19'), sn.Network(1.0)):
        m.always_resync = True
        if hasattr(m, "set_winograd"):
            m.set_winograd(winograd7=4)
        c = copy.deepcopy(m)
        assert c._plans == {} and c._weights == {} and c._native_lock is not m._native_lock and c.always_resync
        assert all(torch.equal(a, b) and a.data_ptr() != b.data_ptr()
                   for a, b in zip(m.state_dict().values(), c.state_dict().values()))
        buf = io.BytesIO()
        torch.save(m, buf)
        buf.seek(0)
        r = torch.load(buf, weights_only=False)
        assert list(r.state_dict()) == list(m.state_dict()) and r._plans == {}
        p = pickle.loads(pickle.dumps(m))
        assert getattr(p, "_wino", None) == getattr(m, "_wino", None)
        p.invalidate_weights()          # the rebuilt lock works


# ---- OKS evaluator: known answers hand-derived from the published COCO keypoint evaluation rules -----------------
# (cocodataset.org/#keypoints-eval; COCOeval.computeOks / evaluateImg / accumulate): sigma table, OKS formula,
# greedy matching in score order, crowd / num_keypoints == 0 ground truth ignored, maxDets = 20, area ranges,
# 101-point interpolated precision.  Expected values are computed here from those rules, not through oks_eval.

_COCO_SIGMAS = [.026, .025, .025, .035, .035, .079, .079, .072, .072, .062, .062, .107, .107, .087, .087, .089, .089]


def _person(img_id, ann_id, x0, y0, side, **kw):
    """17 visible keypoints on a grid inside a side x side box; area = side^2."""
    kp = []
    for k in range(17):
        kp += [x0 + side * (0.1 + 0.2 * (k % 5)), y0 + side * (0.1 + 0.25 * (k // 5)), 2]
    a = {"image_id": img_id, "id": ann_id, "category_id": 1, "keypoints": kp, "num_keypoints": 17,
         "bbox": [x0, y0, side, side], "area": float(side * side), "iscrowd": 0}
    a.update(kw)
    return a


def _det(gt, score, dx=0.0, dy=0.0):
    kp = list(gt["keypoints"])
    for k in range(17):
        kp[3 * k] += dx
        kp[3 * k + 1] += dy
        kp[3 * k + 2] = 1
    return {"image_id": gt["image_id"], "category_id": 1, "keypoints": kp, "score": score}


def test_oks_known_answer_single_displaced_detection(pkg):
    """OKS = mean_k exp(-d^2 / (2 * area * (2 sigma_k)^2)); a lone detection is a TP for the thresholds <= OKS:
    AP = (#thresholds passed) / 10."""
    oe = importlib.import_module(PKG_NAME + ".oks_eval")
    gt = _person(1, 1, 50.0, 60.0, 80.0)
    d = 6.5
    oks = float(np.mean([np.exp(-d * d / (2.0 * 6400.0 * (2 * s) ** 2)) for s in _COCO_SIGMAS]))
    assert 0.70 < oks < 0.75, oks                       # thresholds .50 .55 .60 .65 .70 pass
    got = oe.compute_oks([gt], [_det(gt, 1.0, dx=d)])
    assert got.shape == (1, 1) and abs(got[0, 0] - oks) < 1e-12
    r = oe.evaluate([gt], [_det(gt, 1.0, dx=d)])
    # (precision is tp / (tp + fp + eps) in COCOeval: 1 - 1e-16, not 1)
    assert abs(r["AP"] - 0.5) < 1e-9 and abs(r["AP50"] - 1.0) < 1e-9 and r["AP75"] == 0.0
    assert abs(r["APm"] - 0.5) < 1e-9 and r["APl"] == -1.0     # area 6400 is 'medium' (32^2 .. 96^2), nothing 'large'
    # vertical displacement counts the same; an invisible ground-truth keypoint drops out of the mean
    gt2 = _person(1, 1, 50.0, 60.0, 80.0)
    gt2["keypoints"][3 * 16 + 2] = 0
    gt2["num_keypoints"] = 16
    oks16 = float(np.mean([np.exp(-d * d / (2.0 * 6400.0 * (2 * s) ** 2)) for s in _COCO_SIGMAS[:16]]))
    assert abs(oe.compute_oks([gt2], [_det(gt2, 1.0, dy=d)])[0, 0] - oks16) < 1e-12


def test_oks_known_answer_101_point_interpolation(pkg):
    """Two people, detections in score order TP, FP, TP: recall .5 .5 1, precision 1 .5 2/3 -> envelope 1 2/3 2/3;
    the 51 recall points 0 .. 0.50 read precision 1, the 50 points 0.51 .. 1.00 read 2/3, at every OKS threshold."""
    oe = importlib.import_module(PKG_NAME + ".oks_eval")
    g1, g2 = _person(3, 1, 10.0, 10.0, 100.0), _person(3, 2, 300.0, 40.0, 120.0)
    junk = _det(g1, 0.8, dx=5000.0)
    r = oe.evaluate([g1, g2], [_det(g1, 0.9), junk, _det(g2, 0.7)])
    want = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101.0
    assert abs(r["AP"] - want) < 1e-9 and abs(r["AP50"] - want) < 1e-9 and abs(r["AP75"] - want) < 1e-9
    # 'large' range: both people (areas 10000, 14400) count; the junk detection's own area - its keypoint bounding
    # box, 80 x 75 = 6000 (loadRes) - is 'medium', and an UNMATCHED detection outside the range is ignored there
    assert abs(r["APl"] - 1.0) < 1e-9 and r["APm"] == -1.0
    # a second detection of an already matched person is a false positive (one match per ground truth)
    r2 = oe.evaluate([g1], [_det(g1, 0.9), _det(g1, 0.8, dx=1.0)])
    assert abs(r2["AP"] - 1.0) < 1e-9       # TP first: precision 1 at recall 1 already
    r3 = oe.evaluate([g1], [_det(g1, 0.8), _det(g1, 0.9, dx=5000.0)])
    assert abs(r3["AP"] - 0.5) < 1e-9       # FP first: precision 1/2 everywhere


def test_oks_known_answer_ignore_rules_and_max_dets(pkg):
    """Crowd and num_keypoints == 0 ground truth is ignored and so are the detections matched to it; only the 20
    best-scored detections of an image are evaluated; a ground truth outside the area range is ignored there."""
    oe = importlib.import_module(PKG_NAME + ".oks_eval")
    g1 = _person(5, 1, 10.0, 10.0, 40.0)                              # area 1600: medium
    crowd = _person(5, 2, 200.0, 10.0, 50.0, iscrowd=1)
    empty = _person(5, 3, 400.0, 10.0, 50.0, num_keypoints=0)
    for k in range(17):
        empty["keypoints"][3 * k + 2] = 0
    inside = _det(empty, 0.92)                                        # inside the doubled box of `empty`: OKS 1
    dts = [_det(g1, 0.9), _det(crowd, 0.95), _det(crowd, 0.93, dx=0.5), inside]
    r = oe.evaluate([g1, crowd, empty], dts)
    assert abs(r["AP"] - 1.0) < 1e-9 and abs(r["APm"] - 1.0) < 1e-9 and r["APl"] == -1.0
    # the same with a junk detection scored above the true positive: FP, TP -> precision 1/2 at every recall
    r = oe.evaluate([g1, crowd, empty], dts + [_det(g1, 0.99, dx=4000.0)])
    assert abs(r["AP"] - 0.5) < 1e-9
    # maxDets = 20: the perfect detection ranks 21st and is never looked at
    many = [_det(g1, 0.5 + 0.01 * i, dx=3000.0 + 50 * i) for i in range(20)] + [_det(g1, 0.4)]
    assert oe.evaluate([g1], many)["AP"] == 0.0
    assert abs(oe.evaluate([g1], many[1:])["AP"] - 1.0 / 20.0) < 1e-9   # now 20th of 20: precision 1/20 at recall 1
