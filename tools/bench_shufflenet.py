"""BASELINE config 4 (developer bench; bench.py keeps the headline contract):
ShuffleNetV2 x1.0 pose net, 368x368, batch 128, fp32, one MI355X — net only and net+decode."""
import ctypes as C
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "pytorch_realtime_multi-person_pose_estimation_amd"
pkg = importlib.import_module(PKG)
synth = importlib.import_module(pkg.__name__ + ".synth")
sn = importlib.import_module(PKG + ".shufflenet")
capi = pkg._capi
lib = capi.lib

GFLOP_PER_IMG = 5.543
MIN_BYTES_PER_IMG = 1.63e6 + 0.48e6   # read input + write outputs (weights 5.2 MB once per batch)


def main(n=128, iters=10, dtype='fp32'):
    dev = torch.device("cuda:0")
    m = sn.Network(1.0)
    m.load_state_dict(synth.seeded_shufflenet_state_dict(m, 0))
    m = m.cuda().eval()
    m.set_compute_dtype(dtype)
    x = (torch.rand(n, 3, 368, 368, generator=torch.Generator().manual_seed(0)) - 0.5).to(dev)
    plan = m.forward_native(x)
    torch.cuda.synchronize()
    nl = lib.rtpose_shufflenet_num_launches(plan.handle)

    def timed():
        for _ in range(2):
            m.forward_native(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            m.forward_native(x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters
    # the production configuration first: no per-launch events (round 6: the number this tool used to print was taken with
    # them - an event record between two launches is a 5.7 us bubble on the stream, 40 of them per forward; the launch
    # sequence of profiles/r06_shufflenet_sequence.txt), then the same with the events for the per-launch table
    wall = timed()
    lib.rtpose_shufflenet_set_profiling(plan.handle, 1)
    wall_events = timed()
    kinds = {}
    tot = 0.0
    for i in range(nl):
        ms, fl = C.c_float(), C.c_double()
        name = C.create_string_buffer(96)
        lib.rtpose_shufflenet_launch_info(plan.handle, i, C.byref(ms), C.byref(fl), name, 96)
        nm = name.value.decode()
        kind = "fused dw+pw(+x1)" if "+conv" in nm else \
            "pw" if ("conv.0" in nm or "conv.2" in nm or "conv0.1" in nm or nm in ("conv5", "paf+heatmap")) else \
            ("dw" if ("conv.1" in nm or "conv0.0" in nm) else ("copy" if "x1->even" in nm else nm))
        if os.environ.get("VERBOSE"):
            print("  %-34s %7.3f ms %7.2f TF/s" % (nm, ms.value, fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0))
        a = kinds.setdefault(kind, [0.0, 0.0, 0])
        a[0] += ms.value
        a[1] += fl.value
        a[2] += 1
        tot += ms.value
    for k, (ms, fl, cnt) in sorted(kinds.items(), key=lambda kv: -kv[1][0]):
        print("%-28s %3d launches %8.3f ms  %7.2f TF/s" % (k, cnt, ms, fl / (ms * 1e-3) / 1e12 if ms else 0))
    print("launches %d, sum %.3f ms, wall %.3f ms/forward -> %.0f img/s, %.2f TFLOP/s, %.1f GB/s vs fused-min bytes "
          "(with per-launch events: wall %.3f ms)"
          % (nl, tot, wall * 1e3, n / wall, n / wall * GFLOP_PER_IMG / 1e3, n / wall * MIN_BYTES_PER_IMG / 1e9, wall_events * 1e3))


if __name__ == "__main__" and len(sys.argv) > 3:
    main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
elif __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:]])
