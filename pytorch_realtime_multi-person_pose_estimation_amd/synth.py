"""Synthetic multi-person heat-map / PAF scenes (input data for tests and bench).

There are no trained weights and no COCO images in this environment, so the
post-processing is exercised on rasterised stick figures.  The rasterisation
follows the formulas of the reference's ground-truth generators —
Gaussian blobs as in lib/datasets/heatmap.py:20-36 (sigma in input pixels,
stride 8, clipped at exponent 4.6052, sum clipped to 1) and limb unit-vector
fields as in lib/datasets/paf.py:18-68 (limb width 1 cell, overlaps averaged) —
restated here in vectorised numpy, plus U(0, noise) noise to break float ties.
"""
import numpy as np

# limb -> (partA, partB) and PAF channel pair: lib/pafprocess/pafprocess.h:16-24
PAIRS = [(1, 2), (1, 5), (2, 3), (3, 4), (5, 6), (6, 7), (1, 8), (8, 9), (9, 10), (1, 11), (11, 12),
         (12, 13), (1, 0), (0, 14), (14, 16), (0, 15), (15, 17), (2, 16), (5, 17)]
PAIRS_NET = [(12, 13), (20, 21), (14, 15), (16, 17), (22, 23), (24, 25), (0, 1), (2, 3), (4, 5), (6, 7),
             (8, 9), (10, 11), (28, 29), (30, 31), (34, 35), (32, 33), (36, 37), (18, 19), (26, 27)]

# a standing figure in unit coordinates (x right, y down), part ids of lib/utils/common.py:5-24
_TEMPLATE = np.array([
    [0.00, -0.80],   # 0 nose
    [0.00, -0.60],   # 1 neck
    [-0.18, -0.58],  # 2 r shoulder
    [-0.26, -0.32],  # 3 r elbow
    [-0.28, -0.08],  # 4 r wrist
    [0.18, -0.58],   # 5 l shoulder
    [0.26, -0.32],   # 6 l elbow
    [0.28, -0.08],   # 7 l wrist
    [-0.11, -0.05],  # 8 r hip
    [-0.12, 0.35],   # 9 r knee
    [-0.12, 0.75],   # 10 r ankle
    [0.11, -0.05],   # 11 l hip
    [0.12, 0.35],    # 12 l knee
    [0.12, 0.75],    # 13 l ankle
    [-0.04, -0.84],  # 14 r eye
    [0.04, -0.84],   # 15 l eye
    [-0.09, -0.80],  # 16 r ear
    [0.09, -0.80],   # 17 l ear
])


def random_people(rng, n_people, height, width, drop_prob=0.1):
    """Returns a list of (18,2) float arrays (input-pixel coords, NaN = joint absent)."""
    people = []
    for _ in range(n_people):
        scale = rng.uniform(0.28, 0.5) * height
        cx = rng.uniform(0.15, 0.85) * width
        cy = rng.uniform(0.45, 0.6) * height
        ang = rng.uniform(-0.25, 0.25)
        rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        pts = (_TEMPLATE + rng.normal(0, 0.015, _TEMPLATE.shape)) @ rot.T * scale + [cx, cy]
        pts = pts.astype(np.float64)
        drop = rng.uniform(size=18) < drop_prob
        outside = (pts[:, 0] < 2) | (pts[:, 0] > width - 3) | (pts[:, 1] < 2) | (pts[:, 1] > height - 3)
        pts[drop | outside] = np.nan
        people.append(pts)
    return people


def render(people, height, width, stride=8, sigma=7.0, noise=0.02, rng=None):
    """-> heat [h,w,19] float32, paf [h,w,38] float32 (HWC, like get_outputs returns)."""
    h, w = height // stride, width // stride
    start = stride / 2.0 - 0.5
    ys, xs = np.mgrid[0:h, 0:w]
    gx, gy = xs * stride + start, ys * stride + start
    heat = np.zeros((h, w, 19), np.float64)
    for pts in people:
        for j in range(18):
            if np.isnan(pts[j, 0]):
                continue
            e = ((gx - pts[j, 0]) ** 2 + (gy - pts[j, 1]) ** 2) / 2.0 / sigma / sigma
            heat[:, :, j] += np.where(e <= 4.6052, np.exp(-e), 0.0)
    heat = np.minimum(heat, 1.0)
    heat[:, :, 18] = np.maximum(1.0 - heat[:, :, :18].max(axis=2), 0.0)
    paf = np.zeros((h, w, 38), np.float64)
    for (a, b), (cx_, cy_) in zip(PAIRS, PAIRS_NET):
        acc = np.zeros((h, w, 2))
        cnt = np.zeros((h, w))
        for pts in people:
            if np.isnan(pts[a, 0]) or np.isnan(pts[b, 0]):
                continue
            ca, cb = pts[a] / stride, pts[b] / stride
            v = cb - ca
            nrm = np.linalg.norm(v)
            if nrm == 0:
                continue
            u = v / nrm
            x0 = max(int(round(min(ca[0], cb[0]) - 1)), 0)
            x1 = min(int(round(max(ca[0], cb[0]) + 1)), w)
            y0 = max(int(round(min(ca[1], cb[1]) - 1)), 0)
            y1 = min(int(round(max(ca[1], cb[1]) + 1)), h)
            if x1 <= x0 or y1 <= y0:
                continue
            yy, xx = np.mgrid[y0:y1, x0:x1]
            m = np.abs((xx - ca[0]) * u[1] - (yy - ca[1]) * u[0]) < 1
            acc[y0:y1, x0:x1, 0] += m * u[0]
            acc[y0:y1, x0:x1, 1] += m * u[1]
            cnt[y0:y1, x0:x1] += m
        cnt = np.maximum(cnt, 1)
        paf[:, :, cx_] = acc[:, :, 0] / cnt
        paf[:, :, cy_] = acc[:, :, 1] / cnt
    if noise > 0:
        rng = rng or np.random.default_rng(0)
        heat += rng.uniform(0, noise, heat.shape)
        paf += rng.uniform(-noise, noise, paf.shape)
    return heat.astype(np.float32), paf.astype(np.float32)


def make_batch(n_images, height=368, width=368, seed=1, max_people=8, noise=0.02):
    """Seeded batch: heat [N,h,w,19], paf [N,h,w,38] float32 + the people lists."""
    rng = np.random.default_rng(seed)
    heats, pafs, gt = [], [], []
    for _ in range(n_images):
        people = random_people(rng, int(rng.integers(1, max_people + 1)), height, width)
        hm, pf = render(people, height, width, noise=noise, rng=rng)
        heats.append(hm)
        pafs.append(pf)
        gt.append(people)
    return np.stack(heats), np.stack(pafs), gt


def he_init_state_dict(model, seed=0):
    """Seeded stand-in weights for `pose_model.pth` (not available offline): Kaiming-normal conv
    weights + N(0, 0.05) biases, drawn key by key in state_dict order from one torch generator.
    The reference's own init (lib/network/rtpose_vgg.py:200-222: N(0, 0.01), zero bias) drives the
    stage outputs to ~5e-11, useless as a workload or for parity.  bench.py, the tools and the
    tests all use this; oracle/net_oracle.py carries the same recipe for the build-container
    golden generators (tests/test_host_golden_cpu.py checks that the two agree)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in model.state_dict().items():
        if k.endswith('.weight'):
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            sd[k] = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
        else:
            sd[k] = torch.randn(v.shape, generator=g) * 0.05
    return sd


def seeded_shufflenet_state_dict(model, seed=0):
    """Seeded stand-in weights for the ShuffleNetV2 pose net (BASELINE configs[3]): Kaiming convs,
    BN gamma in [0.35, 0.75) so activations stay O(1), running statistics away from (0, 1).
    Same recipe as oracle/shufflenet_oracle.py:seeded_state_dict."""
    import torch
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in model.state_dict().items():
        if k.endswith("num_batches_tracked"):
            sd[k] = v.clone()
        elif k.endswith("running_var"):
            sd[k] = torch.rand(v.shape, generator=g) + 0.5
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
        elif v.dim() == 4:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            sd[k] = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif k.endswith(".weight"):
            sd[k] = torch.rand(v.shape, generator=g) * 0.4 + 0.35
        else:
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
    return sd


# ---- counter-based hash shared with the torch-free C++ host (examples/c_host.cpp: hash_uniform) ---------------------
def hash_uniform(stream, count):
    """float64 [count]: element idx of stream `stream`, uniform in [0, 1) with 53 bits - splitmix64 of a counter,
    the same bits as examples/c_host.cpp:hash_uniform produces."""
    with np.errstate(over="ignore"):
        z = (np.uint64(stream) * np.uint64(0x9E3779B97F4A7C15)
             + np.arange(count, dtype=np.uint64) * np.uint64(0xD1B54A32D192ED03) + np.uint64(0x2545F4914F6CDD1D))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def hashed_state_dict(model):
    """The stand-in for pose_model.pth that examples/c_host.cpp generates: conv i (state_dict order) takes stream 2 i
    for its OIHW weights - uniform with the He variance 2 / fan_in - and stream 2 i + 1 for its bias (+-0.05)."""
    import torch
    sd = {}
    keys = list(model.state_dict().keys())
    convs = [k[:-len(".weight")] for k in keys if k.endswith(".weight")]
    for i, prefix in enumerate(convs):
        shape = tuple(model.state_dict()[prefix + ".weight"].shape)
        fan_in = shape[1] * shape[2] * shape[3]
        sc = np.sqrt(24.0 / fan_in)
        w = ((hash_uniform(2 * i, int(np.prod(shape))) - 0.5) * sc).astype(np.float32).reshape(shape)
        b = ((hash_uniform(2 * i + 1, shape[0]) - 0.5) * 0.1).astype(np.float32)
        sd[prefix + ".weight"] = torch.from_numpy(w)
        sd[prefix + ".bias"] = torch.from_numpy(b)
    return {k: sd[k] for k in keys}


def hashed_input(n, h=368, w=368):
    """float32 [n, 3, h, w] in [-0.5, 0.5): stream 1000 of the hash (examples/c_host.cpp's input batch)."""
    return (hash_uniform(1000, n * 3 * h * w) - 0.5).astype(np.float32).reshape(n, 3, h, w)


def record_digest(recs):
    """FNV-1a 64 over the words a result record DEFINES (header[0..4], part counts, the counted peaks of every part,
    the first n_humans rows of human_parts and human_score) of int32 records [N, words] - examples/c_host.cpp prints
    the same number."""
    h = 0xCBF29CE484222325
    recs = np.ascontiguousarray(recs, dtype=np.int32)

    def feed(h, words):
        for byte in np.ascontiguousarray(words, dtype=np.int32).tobytes():
            h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        return h
    for r in recs:
        pcap, hcap, nh = int(r[3]), int(r[4]), int(r[1])
        h = feed(h, r[0:5])
        h = feed(h, r[8:26])
        for p in range(18):
            h = feed(h, r[32 + 4 * p * pcap:32 + 4 * p * pcap + 4 * int(r[8 + p])])
        off = 32 + 4 * 18 * pcap
        h = feed(h, r[off:off + 18 * nh])
        h = feed(h, r[off + 18 * hcap:off + 18 * hcap + nh])
    return h
