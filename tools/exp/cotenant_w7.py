"""Developer probe: does the 4x-over-algorithmic fabric traffic of the F(6,7) kernel (562 MB per launch,
~1 TB/s) cost time when the memory system is busy?  The 32 x 368 x 368 fp32 forward is timed per launch (HIP events on the
launch stream) alone and next to a co-tenant that streams a 4 GB tensor through HBM on a second stream (torch.add, read +
write, ~5 TB/s when alone).  Reported: the 7x7 launches' mean time in both settings and the co-tenant's own rate."""
import ctypes as C
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd")
synth = importlib.import_module(pkg.__name__ + ".synth")
lib = pkg._capi.lib


def launch_times(m, plan, x, iters):
    nl = lib.rtpose_net_num_launches(plan.handle)
    k7, k3 = [], []
    for _ in range(iters):
        m.forward_native(x)
        torch.cuda.synchronize()
        for i in range(nl):
            ms, k, fl = C.c_float(), C.c_int(), C.c_double()
            lib.rtpose_net_launch_info(plan.handle, i, C.byref(ms), C.byref(k), C.byref(fl), None, 0)
            if ms.value > 0 and k.value == 7:
                k7.append(ms.value)
            if ms.value > 0 and k.value == 3:
                k3.append(ms.value)
    return sum(k7) / len(k7), sum(k3) / len(k3)


def main():
    dev = torch.device("cuda:0")
    m = pkg.get_model('vgg19')
    m.load_state_dict(synth.he_init_state_dict(m, 0))
    m = m.cuda().eval()
    x = (torch.rand(32, 3, 368, 368, generator=torch.Generator().manual_seed(0)) - 0.5).to(dev)
    plan = m.forward_native(x)
    torch.cuda.synchronize()
    lib.rtpose_net_set_profiling(plan.handle, 1)
    m.forward_native(x)
    torch.cuda.synchronize()
    a7, a3 = launch_times(m, plan, x, 5)
    print("alone:                    7x7 launch %.4f ms, 3x3 launch %.4f ms" % (a7, a3))
    n = 1 << 30                      # 4 GB fp32
    src, dst = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    small = torch.zeros(1 << 20, device=dev)   # 4 MB: L2-resident
    side = torch.cuda.Stream()

    def hbm_pass():      # reads 4 GB, writes 4 GB
        torch.add(src, 1.0, out=dst)

    def alu_pass():      # the same number of launches, no HBM traffic: transcendental chains on an L2-resident tensor
        for _ in range(4):
            torch.sin(small, out=small)

    for name, body in (("HBM co-tenant (torch.add over 4 GB)", hbm_pass), ("ALU co-tenant (sin on 4 MB, L2-resident)", alu_pass)):
        with torch.cuda.stream(side):
            for _ in range(3):
                body()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            for _ in range(10):
                body()
        torch.cuda.synchronize()
        alone = (time.perf_counter() - t0) / 10 * 1e3
        reps = 6
        with torch.cuda.stream(side):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            npass = int(reps * 30 / max(alone, 0.05)) + 1
            for _ in range(npass):
                body()
            e1.record()
        c7, c3 = launch_times(m, plan, x, reps)
        torch.cuda.synchronize()
        bg = e0.elapsed_time(e1) / npass
        print("%s: alone %.3f ms per pass%s; next to the forward %.3f ms per pass (x%.2f)" % (
            name, alone, " = %.2f TB/s" % (8 * n / (alone * 1e-3) / 1e12) if body is hbm_pass else "", bg, bg / alone))
        print("    forward next to it:   7x7 launch %.4f ms (x%.3f), 3x3 launch %.4f ms (x%.3f)" % (c7, c7 / a7, c3, c3 / a3))


if __name__ == "__main__":
    main()
