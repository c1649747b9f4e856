"""Developer tool: single-image (batch 1) latency of the forward and of the whole
get_outputs_gpu + paf_to_pose flow (the reference's own usage pattern, demo/picture_demo.py),
for each compute dtype."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd")
pre = importlib.import_module(pkg.__name__ + ".preprocess")
dec = importlib.import_module(pkg.__name__ + ".decode")
synth = importlib.import_module(pkg.__name__ + ".synth")
pipeline = importlib.import_module(pkg.__name__ + ".pipeline")


def main(iters=50):
    m = pkg.get_model('vgg19')
    m.load_state_dict(synth.he_init_state_dict(m, 0))
    m = m.cuda().eval()
    x = (torch.rand(1, 3, 368, 368) - 0.5).cuda()
    est = pipeline.PoseEstimator(m)
    for dt in ('fp32', 'bf16x3', 'bf16'):
        m.set_compute_dtype(dt)
        for _ in range(5):
            m.forward_native(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            m.forward_native(x)
        torch.cuda.synchronize()
        fwd = (time.perf_counter() - t0) / iters
        # random weights make junk maps (hundreds of false peaks), so the decode leg is timed on a
        # synthetic scene blended over the net output, as bench.py does
        heat_np, paf_np, _ = synth.make_batch(1, 368, 368, seed=3, max_people=4)
        scene = (torch.from_numpy(heat_np).cuda(), torch.from_numpy(paf_np).cuda())
        for _ in range(3):
            humans = est.humans(x, scene=scene)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            humans = est.humans(x, scene=scene)
        e2e = (time.perf_counter() - t0) / iters
        print("%-7s batch-1 forward %.3f ms   forward + decode + D2H + Human objects %.3f ms (%d humans) -> %.0f img/s"
              % (dt, fwd * 1e3, e2e * 1e3, len(humans[0]), 1.0 / e2e))


if __name__ == "__main__":
    main()
