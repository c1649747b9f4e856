"""Developer probe (round 6): the decoder of batch k on a second stream BESIDE the forward of batch k + 1, R decodes per
step so that the decoder is exposed to every launch of the forward, every record compared with the serial path's.

    python tools/exp/overlap_soak.py STEPS [dtype ...]         (dtype: bf16 fp32 bf16x3; default bf16)

environment
    REPEATS=12        decodes (into separate buffers) per step on the side stream; the hit histogram by repeat index
                      says which launches of the forward were the neighbours
    PEOPLE=8          every image of the three 16-image batches carries this many people (7-8 peaks per part: the
                      candidates scored by lanes >= 41 of the scoring wave, the ones the round-5 finding hit)
    CU_MASK=1         decoder stream on 32 CUs (hipExtStreamCreateWithCUMask), the forward's stream on the other 224
    RTPOSE_LIB_PATH, RTPOSE_GUARD_OP=-1 (developer build: the forward waits for the decoder in front of its LAST launch),
    RTPOSE_LIMB_A32=0|1, RTPOSE_EXP_POISON=1: see csrc/net.hip, csrc/decode.hip
Prints one line per differing record and a summary line per dtype (also as JSON for the session scripts)."""
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


def hip_runtime():
    """The libamdhip64 this process already runs on (torch's), for hipExtStreamCreateWithCUMask."""
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return C.CDLL(line.split()[-1])
    raise RuntimeError("no libamdhip64 mapped")


def masked_streams():
    hip = hip_runtime()
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    dec_mask = (C.c_uint32 * 8)()
    fwd_mask = (C.c_uint32 * 8)()
    for cu in range(256):
        if (cu // 8) % 8 == 0:
            dec_mask[cu // 32] |= 1 << (cu % 32)
        else:
            fwd_mask[cu // 32] |= 1 << (cu % 32)
    out = []
    for mask in (fwd_mask, dec_mask):
        s = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, mask)
        assert rc == 0, rc
        out.append(torch.cuda.ExternalStream(s.value))
    return out


def main(steps, dtypes):
    pkg = importlib.import_module(PKG)
    dec = importlib.import_module(PKG + ".decode")
    synth = importlib.import_module(PKG + ".synth")
    capi = pkg._capi
    lib, check, ptr = capi.lib, capi.check, capi.ptr
    dev = torch.device("cuda", 0)
    R = int(os.environ.get("REPEATS", "12"))
    people = int(os.environ.get("PEOPLE", "8"))
    B, S = 16, 368
    m = pkg.get_model('vgg19')
    m.load_state_dict(synth.he_init_state_dict(m, seed=0))
    m = m.cuda().float().eval()

    def batch(r):
        g = torch.Generator().manual_seed(300 + r)
        rng = np.random.default_rng(400 + r)
        hs, ps = [], []
        for _ in range(B):
            hm, pf = synth.render(synth.random_people(rng, people, S, S, drop_prob=0.02), S, S, noise=0.02, rng=rng)
            hs.append(hm)
            ps.append(pf)
        return ((torch.rand(B, 3, S, S, generator=g) - 0.5).to(dev),
                torch.from_numpy(np.stack(hs)).to(dev), torch.from_numpy(np.stack(ps)).to(dev))

    data = [batch(r) for r in range(3)]
    cfg = dec.make_cfg(dec.default_config(), 32, 64)
    if os.environ.get("CU_MASK") == "1":
        compute, side = masked_streams()
    else:
        compute, side = torch.cuda.Stream(), torch.cuda.Stream()
    slots = []
    for _ in range(2):
        bufs = [dec.DecodeBuffers(cfg, B, dev) for _ in range(R)]
        conn_words = B * 19 * (1 + 3 * 32)
        slots.append({"bufs": bufs, "host": [torch.empty(b.result.shape, dtype=torch.int32).pin_memory() for b in bufs],
                      "conn": [torch.empty(conn_words, dtype=torch.int32).pin_memory() for _ in bufs],
                      "maps": torch.cuda.Event(), "dec_done": torch.cuda.Event(), "done": torch.cuda.Event()})
    summary = {}
    for dt in dtypes:
        m.set_compute_dtype(dt)
        x0 = data[0][0]
        plan = m.plan_for(x0)

        def forward_blend(r, guard):
            x, sh, sp = data[r]
            check(lib.rtpose_net_set_output_guard(plan.handle, guard))
            try:
                p = m.forward_native(x, keep_intermediates=False)
            finally:
                check(lib.rtpose_net_set_output_guard(plan.handle, None))
            assert p is plan
            pbase, lpaf, _, h, w = m.output_view(plan, 0)
            hbase, lheat, _, _, _ = m.output_view(plan, 1)
            s = capi.current_stream()
            check(lib.rtpose_layout_axpby(hbase, C.byref(lheat), ptr(sh), 19, B, h, w, 1e-3, 1.0, s))
            check(lib.rtpose_layout_axpby(pbase, C.byref(lpaf), ptr(sp), 38, B, h, w, 1e-3, 1.0, s))
            return hbase, lheat, pbase, lpaf, h, w

        def decode_into(maps, bufs):
            hbase, lheat, pbase, lpaf, h, w = maps
            bufs.result.zero_()     # whole blocks compare: no left-overs of another batch behind the counts
            dec.decode_enqueue(hbase, lheat, pbase, lpaf, B, h, w, bufs)

        # the serial path's records (one stream, nothing beside the decoder), twice
        want, want_conn = [], []
        with torch.cuda.stream(compute):
            for rep in range(2):
                got = []
                for r in range(3):
                    maps = forward_blend(r, None)
                    decode_into(maps, slots[0]["bufs"][0])
                    compute.synchronize()
                    got.append((slots[0]["bufs"][0].result.cpu().numpy().reshape(B, -1).copy(),
                                slots[0]["bufs"][0].workspace[:conn_words].cpu().numpy().copy()))
                if rep == 0:
                    want = [g[0] for g in got]
                    want_conn = [g[1] for g in got]
                else:
                    print(dt, "serial path repeatable:", all(np.array_equal(a, g[0]) for a, g in zip(want, got)), flush=True)
        pcs = [w[:, dec.RES_PART_COUNT:dec.RES_PART_COUNT + 18] for w in want]
        print(dt, "peaks per part: mean %.2f, max %d" % (np.mean([p.mean() for p in pcs]), max(p.max() for p in pcs)))
        order = [0, 1, 2, 2, 1, 0, 0, 1, 2, 1]
        bad = 0
        hist = [0] * R
        bad_steps = set()
        t0 = time.time()

        def collect(k, slot, r):
            nonlocal bad
            slot["done"].synchronize()
            for j in range(R):
                got = slot["host"][j].numpy().reshape(B, -1)
                if np.array_equal(got, want[r]):
                    continue
                for b in range(B):
                    if np.array_equal(got[b], want[r][b]):
                        continue
                    bad += 1
                    hist[j] += 1
                    bad_steps.add(k)
                    a, w = dec.parse_image(got[b]), dec.parse_image(want[r][b])
                    what = []
                    for f in ("peaks", "parts", "score"):
                        if a[f].shape != w[f].shape or not np.array_equal(a[f].view(np.uint32) if a[f].dtype == np.float32 else a[f],
                                                                         w[f].view(np.uint32) if w[f].dtype == np.float32 else w[f]):
                            what.append(f)
                    nan = bool(np.isnan(a["score"]).any())
                    cw = 19 * (1 + 3 * 32)
                    ca = slot["conn"][j].numpy()[b * cw:(b + 1) * cw]
                    cb = want_conn[r][b * cw:(b + 1) * cw]
                    limbs = []
                    for limb in range(19):
                        la, lb = ca[limb * 97:limb * 97 + 97], cb[limb * 97:limb * 97 + 97]
                        na, nb = int(la[0]), int(lb[0])
                        if na != nb or not np.array_equal(la[1:1 + 3 * na], lb[1:1 + 3 * nb]):
                            cnts = want[r][b][dec.RES_PART_COUNT:dec.RES_PART_COUNT + 18]
                            nA, nB = int(cnts[synth.PAIRS[limb][0]]), int(cnts[synth.PAIRS[limb][1]])
                            fa = {(int(x[0]), int(x[1])): float(x[2:3].view(np.float32)[0]) for x in la[1:1 + 3 * na].reshape(-1, 3)}
                            fb = {(int(x[0]), int(x[1])): float(x[2:3].view(np.float32)[0]) for x in lb[1:1 + 3 * nb].reshape(-1, 3)}
                            diff = ["(%d,%d) lane %d: %.7g vs %.7g" % (p[0], p[1], (p[0] * nB + p[1]) % 256 % 64, fa[p], fb[p])
                                    for p in fa if p in fb and fa[p] != fb[p]]
                            limbs.append("limb %d (%dx%d): %s%s" % (limb, nA, nB, "; ".join(diff) or "assignment differs",
                                                                    " NaN" if any(np.isnan(v) for v in fa.values()) else ""))
                    print("%s step %d (batch %d) repeat %d image %d: %s differ%s; %s" % (
                        dt, k, r, j, b, "/".join(what) or "other words", " [NaN score]" if nan else "", " | ".join(limbs)), flush=True)

        with torch.cuda.stream(compute):
            prev = None
            for k in range(steps):
                r = order[k % len(order)]
                slot = slots[k & 1]
                guard = prev[1]["dec_done"].cuda_event if prev is not None else None
                maps = forward_blend(r, guard)
                slot["maps"].record(compute)
                side.wait_event(slot["maps"])
                with torch.cuda.stream(side):
                    for j in range(R):
                        decode_into(maps, slot["bufs"][j])
                    slot["dec_done"].record(side)
                    for j in range(R):
                        slot["host"][j].copy_(slot["bufs"][j].result, non_blocking=True)
                        slot["conn"][j].copy_(slot["bufs"][j].workspace[:conn_words], non_blocking=True)
                    slot["done"].record(side)
                if prev is not None:
                    collect(*prev)
                prev = (k, slot, r)
            collect(*prev)
        torch.cuda.synchronize()
        el = time.time() - t0
        print("%s: %d differing records in %d steps x %d decodes x %d images (%d steps hit); by repeat index %s; %.1f s" % (
            dt, bad, steps, R, B, len(bad_steps), hist, el), flush=True)
        summary[dt] = {"differing": bad, "steps": steps, "repeats": R, "steps_hit": len(bad_steps), "by_repeat": hist,
                       "seconds": round(el, 1)}
    print("SUMMARY " + json.dumps({"env": {k: os.environ.get(k) for k in (
        "RTPOSE_LIB_PATH", "RTPOSE_GUARD_OP", "RTPOSE_GUARD_FINE", "RTPOSE_LIMB_A32", "RTPOSE_EXP_POISON", "CU_MASK",
        "REPEATS", "PEOPLE")}, "result": summary}), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 200, tuple(a[1:]) or ("bf16",))
