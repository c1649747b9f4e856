"""Round 6 reproducer driver: the self-checking packed-fp32 victim (tools/exp/pk_victim.hip) on a second stream beside the
library's own forward (bf16 / fp32 plan) on the compute stream.   python tools/exp/pk_beside_forward.py STEPS [dtype ...]
dtype `none` = the victim alone.  LAUNCHES=20 victim launches per step, ACTIVE=64 lanes x 304 blocks, ROUNDS=8."""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "pytorch_realtime_multi-person_pose_estimation_amd"


def main(steps, dtypes):
    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    vic = C.CDLL(os.path.join(ROOT, "tools", "exp", "pk_victim.so"))
    vic.pk_victim_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    dev = torch.device("cuda", 0)
    L = int(os.environ.get("LAUNCHES", "20"))
    active = int(os.environ.get("ACTIVE", "64"))
    rounds = int(os.environ.get("ROUNDS", "8"))
    blocks = 304
    m = pkg.get_model('vgg19')
    m.load_state_dict(synth.he_init_state_dict(m, seed=0))
    m = m.cuda().float().eval()
    B, S = 16, 368
    x = (torch.rand(B, 3, S, S) - 0.5).to(dev)
    g = torch.Generator().manual_seed(5)
    inp = torch.empty(blocks * 256, 4)
    inp[:, 0:2] = torch.randint(0, 368, (blocks * 256, 2), generator=g).float()          # A: integer pixel coordinates
    inp[:, 2:4] = (torch.randint(-300, 300, (blocks * 256, 2), generator=g).float() / 10.0)   # step = (B - A) / 10
    inp = inp.to(dev)
    compute, side = torch.cuda.Stream(), torch.cuda.Stream()
    out = {}
    for dt in dtypes:
        if dt != "none":
            m.set_compute_dtype(dt)
            with torch.cuda.stream(compute):
                m.forward_native(x, keep_intermediates=False)
            torch.cuda.synchronize()
        for variant in (() if os.environ.get('ONLY_WAR') else (0, 1, 2, 3, 4, 5)):
            hist = torch.zeros(65, dtype=torch.int32, device=dev)
            detail = torch.zeros(64 * 8, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            t0 = time.time()
            for k in range(steps):
                if dt != "none":
                    with torch.cuda.stream(compute):
                        m.forward_native(x, keep_intermediates=False)
                for _ in range(L):
                    rc = vic.pk_victim_launch(variant, blocks, active, rounds, inp.data_ptr(), hist.data_ptr(),
                                              detail.data_ptr(), side.cuda_stream)
                    assert rc == 0, rc
                if k % 8 == 7:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            h = hist.cpu().numpy()
            lanes = {i: int(h[i]) for i in range(64) if h[i]}
            d = detail.cpu().numpy().reshape(64, 8)[:min(int(h[64]), 6)]
            print("%s variant %d: %d mismatching (lane, sample) results in %d steps x %d launches x %d blocks x %d lanes x %d x 10 "
                  "samples; lanes %s; %.1f s" % (dt, variant, int(h[64]), steps, L, blocks, active, rounds, lanes, time.time() - t0),
                  flush=True)
            for r in d:
                print("    block %d lane %d sample %d round %d: packed (%d, %d) scalar (%d, %d)" % (r[0], r[1] & 63, r[2], r[7], r[3], r[4], r[5], r[6]))
            out["%s/%d" % (dt, variant)] = {"mismatches": int(h[64]), "lanes": lanes}
        # address-register WAR victim (tools/exp/vmem_war_victim.hip): 8 loads, then their offset registers are overwritten
        if os.path.exists(os.path.join(ROOT, "tools", "exp", "vmem_war_victim.so")) and not os.environ.get("SKIP_WAR"):
            wv = C.CDLL(os.path.join(ROOT, "tools", "exp", "vmem_war_victim.so"))
            wv.war_victim_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p,
                                             C.c_void_p, C.c_void_p]
            words = 46 * 46 * 57 * 16          # the size of a 16-image map buffer
            table = torch.cat([torch.arange(words, dtype=torch.int32),
                               (torch.arange(words, dtype=torch.int64) | 0xBAD00000).to(torch.int32)]).to(dev)
            for gap in (0, 2, 8, 32):
                for act in (49, 64, 256):
                    hist = torch.zeros(65, dtype=torch.int32, device=dev)
                    detail = torch.zeros(64 * 4, dtype=torch.int32, device=dev)
                    torch.cuda.synchronize()
                    t0 = time.time()
                    for k in range(steps):
                        if dt != "none":
                            with torch.cuda.stream(compute):
                                m.forward_native(x, keep_intermediates=False)
                        for _ in range(L):
                            rc = wv.war_victim_launch(gap, blocks, act, rounds, table.data_ptr(), words, words * 4,
                                                      hist.data_ptr(), detail.data_ptr(), side.cuda_stream)
                            assert rc == 0, rc
                        if k % 8 == 7:
                            torch.cuda.synchronize()
                    torch.cuda.synchronize()
                    h = hist.cpu().numpy()
                    lanes = {i: int(h[i]) for i in range(64) if h[i]}
                    print("%s WAR victim gap %d active %d: %d wrong loads in %d steps x %d launches x %d blocks x %d lanes x %d x 8 "
                          "loads; lanes %s; %.1f s" % (dt, gap, act, int(h[64]), steps, L, blocks, act, rounds, lanes, time.time() - t0), flush=True)
                    dd = detail.cpu().numpy().reshape(64, 4)[:min(int(h[64]), 4)]
                    for r in dd:
                        print("    thread %d load %d round %d: got 0x%08x want 0x%08x" % (r[0], r[1] & 255, r[1] >> 8, r[2] & 0xffffffff, r[3] & 0xffffffff))
                    out["%s/war/%d/%d" % (dt, gap, act)] = {"mismatches": int(h[64]), "lanes": lanes}
        if os.environ.get("ONLY_WAR"):
            continue
        # the scoring loop itself (tools/exp/limb_victim.hip), SLP-vectorised and not: every launch against the same launch alone
        for build in ("slp", "noslp"):
            lv = C.CDLL(os.path.join(ROOT, "tools", "exp", "limb_victim_%s.so" % build))
            lv.limb_victim_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p]
            gg = torch.Generator().manual_seed(9)
            peaks = torch.zeros(blocks * 16, 4, dtype=torch.int32)
            peaks[:, 0:2] = torch.randint(0, 368, (blocks * 16, 2), generator=gg).int()
            peaks = peaks.to(dev)
            maps = torch.rand(46 * 46 * 57, generator=gg).to(dev)
            for nA, nB in ((7, 7), (7, 8), (8, 8)):
                outs = [torch.zeros(blocks * 64, device=dev) for _ in range(L)]
                ref = torch.zeros(blocks * 64, device=dev)
                torch.cuda.synchronize()
                lv.limb_victim_launch(blocks, peaks.data_ptr(), maps.data_ptr(), 46, 46, 57, nA, nB, ref.data_ptr(), side.cuda_stream)
                torch.cuda.synchronize()
                lanes, bad = {}, 0
                t0 = time.time()
                for k in range(steps):
                    if dt != "none":
                        with torch.cuda.stream(compute):
                            m.forward_native(x, keep_intermediates=False)
                    for o in outs:
                        lv.limb_victim_launch(blocks, peaks.data_ptr(), maps.data_ptr(), 46, 46, 57, nA, nB, o.data_ptr(), side.cuda_stream)
                    torch.cuda.synchronize()
                    for o in outs:
                        d = (o.view(torch.int32) != ref.view(torch.int32)).nonzero().flatten()
                        if len(d):
                            bad += len(d)
                            for i in d.tolist():
                                lanes[i & 63] = lanes.get(i & 63, 0) + 1
                                if bad <= 6:
                                    print("    %s limb victim %s %dx%d: block %d lane %d: %.7g vs %.7g" % (
                                        dt, build, nA, nB, i >> 6, i & 63, float(o[i]), float(ref[i])))
                print("%s limb victim (%s) %dx%d pairs: %d differing scores in %d steps x %d launches x %d blocks; lanes %s; %.1f s" % (
                    dt, build, nA, nB, bad, steps, L, blocks, dict(sorted(lanes.items())), time.time() - t0), flush=True)
                out["%s/limb_%s/%dx%d" % (dt, build, nA, nB)] = {"mismatches": bad, "lanes": lanes}
    print("SUMMARY " + json.dumps(out), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 200, tuple(a[1:]) or ("bf16", "none", "fp32"))
