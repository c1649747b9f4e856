"""Developer tool: PCIe-inclusive throughput (host uint8 images in, host records out) of
pipeline.StreamingPoseEstimator vs the same work without copy/compute overlap."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd")
synth = importlib.import_module(pkg.__name__ + ".synth")
pipeline = importlib.import_module(pkg.__name__ + ".pipeline")


def main(nb=12, B=32):
    m = pkg.get_model('vgg19')
    m.load_state_dict(synth.he_init_state_dict(m, 0))
    m = m.cuda().eval()
    rng = np.random.default_rng(0)
    batches = [np.clip(rng.normal(128, 8, (B, 368, 368, 3)), 0, 255).astype(np.uint8) for _ in range(3)]
    # the decoder input of bench.py: synthetic scene + 1e-3 * maps (a random network draws junk peaks by the thousand,
    # and the rate would be that of decoding them: round 3's 0.9 k img/s for every dtype)
    heat, paf, _ = synth.make_batch(B, 368, 368, seed=100)
    scene = (torch.from_numpy(heat).cuda(), torch.from_numpy(paf).cuda())
    x = (torch.rand(B, 3, 368, 368) - 0.5).cuda()
    for dt in ('fp32', 'bf16x3', 'bf16'):
        m.set_compute_dtype(dt)
        # device-resident reference of the same step (bench.py's timed region): images already in HBM
        ref = pipeline.PoseEstimator(m, max_peaks_per_part=64, max_humans=64)
        for _ in range(2):
            ref(x, scene)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nb):
            importlib.import_module(pkg.__name__ + ".decode").fetch(ref.enqueue(x, scene))
        t_ref = (time.perf_counter() - t0) / nb
        for label, sc in (("scene-blended decoder input", scene), ("junk maps of the random network", None)):
            est = pipeline.StreamingPoseEstimator(m, B, 368, 368, max_peaks_per_part=64 if sc else 256,
                                                  max_humans=64 if sc else 256, scene=sc)
            for _ in est.run(batches[:2]):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for rec in est.run(batches[i % 3] for i in range(nb)):
                n += rec.shape[0]
            dt_s = time.perf_counter() - t0
            print("%-7s host-to-host (H2D overlapped, %s): %d images in %.3f s -> %.1f img/s = %.2f of the "
                  "device-resident %.1f img/s" % (dt, label, n, dt_s, n / dt_s, n / dt_s * t_ref / B, B / t_ref))


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:]])
