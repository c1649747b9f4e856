import sys, importlib, random
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
pkg = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd")
capi = pkg._capi
import test_conv_gpu as T
dev = torch.device("cuda", 0)
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 20260924)
worst = 0.0
lib = capi.lib
d = (capi.ConvDesc * 1)()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    k = rnd.choice((3, 7)); n = rnd.choice((1, 2, 3, 5, 14))
    h, w = rnd.randint(1, 50), rnd.randint(1, 70)
    cin = rnd.choice((16, 32, 48, 64)) if k == 3 else rnd.choice((8, 24, 64))
    cout = rnd.choice((64, 128, 200)) if k == 3 else rnd.choice((128, 256))
    groups = rnd.choice((1, 2)); relu = rnd.choice((0, 1))
    pool = 1 if (k == 3 and h % 2 == 0 and w % 2 == 0 and rnd.random() < 0.3) else 0
    pin = k // 2 + rnd.choice((0, 1))
    d[0].k, d[0].cin, d[0].cout, d[0].pool = k, cin, cout, pool
    d[0].lin = capi.Layout.padded(cin, h, w, pin)
    if not lib.rtpose_conv2d_winograd_fits(d, n, h, w):
        continue
    seed = rnd.randint(0, 10 ** 6)
    print("case", it, (k, n, h, w, cin, cout, groups, relu, pool, pin), end=" ", flush=True)
    wino, _ = T._run_conv(capi, dev, n, h, w, cin, cout, k, relu, pool, pin, 1, seed=seed, groups=groups, winograd=True, skip_ref=True)
    direct, _ = T._run_conv(capi, dev, n, h, w, cin, cout, k, relu, pool, pin, 1, seed=seed, groups=groups, skip_ref=True)
    err = max((a - b).abs().max().item() for a, b in zip(wino, direct))
    rel = err / max(1.0, max(b.abs().max().item() for b in direct))
    worst = max(worst, rel)
    print("rel %.2e" % rel, "BAD" if rel > 1e-4 else "", flush=True)
print("worst", worst)
