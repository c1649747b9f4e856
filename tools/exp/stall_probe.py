"""Developer probe: which step of a streaming batch makes every third batch stall (tools/exp/host_copy_probe.py)?"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd")
synth = importlib.import_module(pkg.__name__ + ".synth")
pipeline = importlib.import_module(pkg.__name__ + ".pipeline")
dec = importlib.import_module(pkg.__name__ + ".decode")


def loop(name, body, n=12):
    torch.cuda.synchronize()
    st = [time.perf_counter()]
    for i in range(n):
        body(i)
        st.append(time.perf_counter())
    print("%-46s" % name, " ".join("%5.1f" % ((b - a) * 1e3) for a, b in zip(st, st[1:])), flush=True)


def main():
    B = 32
    imgs = [np.clip(np.random.default_rng(i).normal(128, 8, (B, 368, 368, 3)), 0, 255).astype(np.uint8) for i in range(3)]
    m = pkg.get_model('vgg19')
    m.load_state_dict(synth.he_init_state_dict(m, 0))
    m = m.cuda().eval()
    m.set_compute_dtype('bf16')
    heat, paf, _ = synth.make_batch(B, 368, 368, seed=100)
    scene = (torch.from_numpy(heat).cuda(), torch.from_numpy(paf).cuda())
    est = pipeline.StreamingPoseEstimator(m, B, 368, 368, max_peaks_per_part=64, max_humans=64, scene=scene)
    for _ in est.run(imgs[:2]):
        pass
    x = (torch.rand(B, 3, 368, 368) - 0.5).cuda()
    ref = pipeline.PoseEstimator(m, max_peaks_per_part=64, max_humans=64)
    ref(x, scene)
    loop("device-resident enqueue + fetch", lambda i: dec.fetch(ref.enqueue(x, scene)))
    loop("streaming: enqueue + finish, no upload", lambda i: est._finish(est._enqueue(0)))
    loop("streaming: upload + enqueue + finish", lambda i: (est._upload(0, imgs[i % 3]), est._finish(est._enqueue(0))))

    def up_direct(i):
        est.devbuf[0].copy_(torch.from_numpy(imgs[i % 3]))
        est._finish(est._enqueue(0))
    loop("streaming: pageable -> device direct", up_direct)

    def up_np(i):
        np.copyto(est.host[0].numpy(), imgs[i % 3])
        est.devbuf[0].copy_(est.host[0], non_blocking=True)
        est._finish(est._enqueue(0))
    loop("streaming: np.copyto -> pinned -> device", up_np)
    torch.set_num_threads(1)
    loop("streaming: upload (1 torch thread)", lambda i: (est._upload(0, imgs[i % 3]), est._finish(est._enqueue(0))))
    loop("device-resident again", lambda i: dec.fetch(ref.enqueue(x, scene)))

    def only_h2d(i):
        est.devbuf[0].copy_(est.host[0], non_blocking=True)
        torch.cuda.current_stream().synchronize()
    loop("H2D of the pinned batch + sync only", only_h2d)

    def h2d_then_kernel(i):
        est.devbuf[0].copy_(est.host[0], non_blocking=True)
        dec.fetch(ref.enqueue(x, scene))
    loop("H2D, then device-resident step", h2d_then_kernel)


if __name__ == "__main__":
    main()
