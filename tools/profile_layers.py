"""Per-launch HIP-event timing of one forward (developer tool; bench.py is the contract)."""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd")
synth = importlib.import_module(pkg.__name__ + ".synth")
capi = pkg._capi
lib = capi.lib


def main(n=32, h=368, w=368, iters=5, dtype='fp32'):
    dev = torch.device("cuda:0")
    m = pkg.get_model('vgg19')
    m.load_state_dict(synth.he_init_state_dict(m, 0))
    m = m.cuda().eval()
    m.set_compute_dtype(dtype)
    x = (torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(0)) - 0.5).to(dev)
    plan = m.forward_native(x)
    torch.cuda.synchronize()
    lib.rtpose_net_set_profiling(plan.handle, 1)
    nl = lib.rtpose_net_num_launches(plan.handle)
    acc = [0.0] * nl
    info = []
    import time
    t0 = time.time()
    for it in range(iters):
        m.forward_native(x)
        torch.cuda.synchronize()
        for i in range(nl):
            ms, k, fl = C.c_float(), C.c_int(), C.c_double()
            name = C.create_string_buffer(96)
            lib.rtpose_net_launch_info(plan.handle, i, C.byref(ms), C.byref(k), C.byref(fl), name, 96)
            acc[i] += ms.value
            if it == 0:
                info.append((name.value.decode(), k.value, fl.value))
    wall = (time.time() - t0) / iters
    tot = 0.0
    bykind = {}
    for i, (name, k, fl) in enumerate(info):
        ms = acc[i] / iters
        tot += ms
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0
        print("%-28s k=%d %8.3f ms %7.1f TF/s" % (name[:28], k, ms, tf))
        a = bykind.setdefault(k, [0.0, 0.0])
        a[0] += ms
        a[1] += fl
    print("sum of launches %.2f ms, wall %.2f ms/forward -> %.1f img/s, %.1f TF/s overall" % (
        tot, wall * 1e3, n / wall, sum(f for _, _, f in info) / wall / 1e12))
    for k, (ms, fl) in sorted(bykind.items()):
        print("k=%d: %.2f ms, %.1f TF/s" % (k, ms, fl / (ms * 1e-3) / 1e12 if ms else 0))


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:5]]
    main(*a, **({'dtype': sys.argv[5]} if len(sys.argv) > 5 else {}))
