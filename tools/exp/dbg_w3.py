"""Scratch: where does a 3x3 Winograd launch differ from the direct kernel / from itself? (developer tool)"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
pkg = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd")
capi = pkg._capi
import test_conv_gpu as T
dev = torch.device("cuda:0")
for (n, h, w, cin, cout, pool) in ((2, 46, 46, 256, 512, 0), (1, 100, 92, 128, 256, 0), (9, 46, 46, 128, 128, 0), (4, 96, 80, 64, 64, 1)):
    a, _ = T._run_conv(capi, dev, n, h, w, cin, cout, 3, 1, pool, 1, 1, seed=5, winograd=True, skip_ref=True)
    b, _ = T._run_conv(capi, dev, n, h, w, cin, cout, 3, 1, pool, 1, 1, seed=5, winograd=True, skip_ref=True)
    d, _ = T._run_conv(capi, dev, n, h, w, cin, cout, 3, 1, pool, 1, 1, seed=5, skip_ref=True)
    a, b, d = a[0].numpy(), b[0].numpy(), d[0].numpy()
    bad = np.abs(a - d) > 1e-3 * max(1.0, np.abs(d).max())
    print("case", (n, h, w, cin, cout, pool), "rerun equal", np.array_equal(a, b), "bad vs direct %.4f" % bad.mean(), "max|d| %.3g" % np.abs(a - d).max())
    if bad.any():
        ch = bad.any(axis=(0, 2, 3))
        print("  bad channels:", np.nonzero(ch)[0][:40], "...", ch.sum(), "of", len(ch))
        px = bad.any(axis=1)
        ys, xs = np.nonzero(px[0])
        print("  image 0 bad pixels: %d of %d; first" % (px[0].sum(), px[0].size), list(zip(ys[:12], xs[:12])))
        print("  per image bad frac", bad.mean(axis=(1, 2, 3)))
