"""BASELINE config 3 (developer tool; bench.py is the contract): multi-scale x4 + flip
test-time augmentation on synthetic 368x368 uint8 images, fp32 vs bf16 plans.
8 forwards-worth per image (scales 0.5/1/1.5/2 x {normal, mirrored}); per step B images are
uploaded as uint8, prepared / merged on the GPU (preprocess.get_multiscale_outputs_batch) and
decoded; only the result records come back.  Also times the per-image host-prepared path."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd")
synth = importlib.import_module(pkg.__name__ + ".synth")
pre = importlib.import_module(pkg.__name__ + ".preprocess")
dec = importlib.import_module(pkg.__name__ + ".decode")

GFLOP_PER_IMAGE = 271.868 * (0.25 + 1.0 + 2.25 + 4.0) * 2   # 4 scales x 2 passes


def main(B=8, iters=5, h0=368, w0=368, dtypes=('fp32', 'bf16x3', 'bf16')):
    m = pkg.get_model('vgg19')
    m.load_state_dict(synth.he_init_state_dict(m, 0))
    m = m.cuda().eval()
    m.keep_intermediates = False
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (h0, w0, 3), dtype=np.uint8) for _ in range(B)]
    out = {}
    for dt in dtypes:
        m.set_compute_dtype(dt)

        def step():
            paf, heat, _ = pre.get_multiscale_outputs_batch(imgs, m)
            return paf, heat, dec.decode_maps(heat, paf)

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(iters):
            paf, heat, recs = step()
        torch.cuda.synchronize()
        dt_s = (time.time() - t0) / iters
        out[dt] = (paf.cpu().numpy(), heat.cpu().numpy())
        print("%s batched: %.2f ms per %d images (4 scales x flip, merge, decode, D2H) -> %.1f img/s, %.0f TFLOP/s"
              % (dt, dt_s * 1e3, B, B / dt_s, B / dt_s * GFLOP_PER_IMAGE / 1e3))
        t0 = time.time()
        pre.get_multiscale_outputs(imgs[0], m)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            pre.get_multiscale_outputs(imgs[0], m)
        torch.cuda.synchronize()
        print("%s per-image host-prepared path: %.2f ms per image" % (dt, (time.time() - t0) / 3 * 1e3))
    if 'fp32' not in out:
        return
    pa, ha = out['fp32']
    for dt in [d for d in ('bf16x3', 'bf16') if d in out]:
        pb, hb = out[dt]
        print("merged maps %s vs fp32: paf max|d| %.4g (max|ref| %.3g), heat max|d| %.4g (max|ref| %.3g)" % (
            dt, np.abs(pa - pb).max(), np.abs(pa).max(), np.abs(ha - hb).max(), np.abs(ha).max()))


if __name__ == "__main__":
    nums = [int(v) for v in sys.argv[1:] if v.isdigit()]
    dts = tuple(v for v in sys.argv[1:] if not v.isdigit())
    main(*nums, **({'dtypes': dts} if dts else {}))
