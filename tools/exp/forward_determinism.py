"""Round 6: is a forward bit-repeatable?  The maps of RUNS forwards over three inputs, each compared bit for bit with the first
forward over the same input (the bf16 kernels hold packed-fp32 VALU instructions beside their own MFMA waves: DESIGN.md 3.3).
    python tools/exp/forward_determinism.py RUNS [vgg:bf16 vgg:bf16x3 vgg:fp32 shufflenet:bf16 shufflenet:fp32]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "pytorch_realtime_multi-person_pose_estimation_amd"


def main(runs, cases):
    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    dev = torch.device("cuda", 0)
    for case in cases:
        net, dt = case.split(":")
        if net == "vgg":
            m = pkg.get_model('vgg19')
            m.load_state_dict(synth.he_init_state_dict(m, seed=0))
            B = 16
        else:
            sn = importlib.import_module(PKG + ".shufflenet")
            m = sn.Network(1.0)
            m.load_state_dict(synth.seeded_shufflenet_state_dict(m, seed=0))
            B = 64
        m = m.cuda().float().eval()
        m.set_compute_dtype(dt)
        xs = [(torch.rand(B, 3, 368, 368, generator=torch.Generator().manual_seed(i)) - 0.5).to(dev) for i in range(3)]

        def maps(x):
            with torch.no_grad():
                out = m(x)
            (paf, heat) = out[0]
            return paf.clone(), heat.clone()
        want = [maps(x) for x in xs]
        bad = 0
        t0 = time.time()
        for r in range(runs):
            i = r % 3
            paf, heat = maps(xs[i])
            if not (torch.equal(paf, want[i][0]) and torch.equal(heat, want[i][1])):
                bad += 1
                dp = (paf != want[i][0]).sum().item()
                dh = (heat != want[i][1]).sum().item()
                print("%s run %d input %d: %d PAF and %d heat-map values differ (max |d| %.3g)" % (
                    case, r, i, dp, dh, max((paf - want[i][0]).abs().max().item(), (heat - want[i][1]).abs().max().item())), flush=True)
        print("%s: %d of %d forwards differ from the first over the same input; %.1f s" % (case, bad, runs, time.time() - t0), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 300, tuple(a[1:]) or ("vgg:bf16", "vgg:bf16x3", "vgg:fp32", "shufflenet:bf16", "shufflenet:fp32"))
