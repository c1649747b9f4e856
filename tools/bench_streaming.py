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
    # smooth images: junk-map peaks stay within the default table capacities
    batches = [np.clip(rng.normal(128, 8, (B, 368, 368, 3)), 0, 255).astype(np.uint8) for _ in range(3)]
    for dt in ('fp32', 'bf16x3', 'bf16'):
        m.set_compute_dtype(dt)
        est = pipeline.StreamingPoseEstimator(m, B, 368, 368, max_peaks_per_part=256, max_humans=256)
        for _ in est.run(batches[:2]):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for rec in est.run(batches[i % 3] for i in range(nb)):
            n += rec.shape[0]
        dt_s = time.perf_counter() - t0
        print("%-7s streaming (H2D overlapped): %d images in %.3f s -> %.1f img/s host-to-host" % (dt, n, dt_s, n / dt_s))


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:]])
