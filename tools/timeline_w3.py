"""Developer tool: per-tile time stamps of the LAST persistent 3x3 Winograd launch (csrc/conv_wino.hip, wino_f32) of one
rtpose_vgg forward at 32 x 368 x 368.  Needs the -DRTPOSE_EXP_TIMELINE3 build (tools/build_dev.sh, RTPOSE_LIB_PATH);
RTPOSE_TIMELINE_WM=1 (default) stamps the <1,4,16> launches, 2 conv1_2's <2,2,16>.
Stamps per tile (s_memtime of wave 0): 0 tile start, 1 accumulators initialised, 2 multiply loop done,
3 output transform + stores issued, 4 stores acknowledged (vmcnt 0)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd")
synth = importlib.import_module(pkg.__name__ + ".synth")
lib = pkg._capi.lib


def main(n=32):
    m = pkg.get_model('vgg19')
    m.load_state_dict(synth.he_init_state_dict(m, 0))
    m = m.cuda().eval()
    x = (torch.rand(n, 3, 368, 368) - 0.5).cuda()
    for _ in range(3):
        m.forward_native(x)
    torch.cuda.synchronize()
    buf = np.zeros((1024, 6, 8), dtype=np.uint64)
    fn = lib.rtpose_debug_timeline_w3_dump
    fn.restype = C.c_int
    nb = fn(C.c_void_p(buf.ctypes.data), 1024)
    t = buf[:nb].astype(np.int64)
    print("blocks", nb)
    for ti in range(6):
        tt = t[:, ti]
        ok = tt[:, 3] > 0
        if not ok.any():
            continue
        tt = tt[ok]
        row = ["tile %d (%d blocks)" % (ti, len(tt))]
        for name, v in (("init", tt[:, 1] - tt[:, 0]), ("loop", tt[:, 2] - tt[:, 1]), ("epi", tt[:, 3] - tt[:, 2]),
                        ("drain", tt[:, 4] - tt[:, 3]), ("total", tt[:, 4] - tt[:, 0])):
            row.append("%s p50 %d p90 %d" % (name, np.percentile(v, 50), np.percentile(v, 90)))
        print("  ".join(row))
    # gaps between consecutive tiles of a block (stamp 0 of tile k+1 - stamp 4 of tile k) and block start skew
    if nb:
        g = t[:, 1:, 0] - t[:, :-1, 4]
        ok = (t[:, 1:, 0] > 0) & (t[:, :-1, 4] > 0)
        if ok.any():
            print("inter-tile gap p50 %d p90 %d" % (np.percentile(g[ok], 50), np.percentile(g[ok], 90)))
        s0 = t[:, 0, 0]
        s0 = s0[s0 > 0]
        print("block start skew: p50 %d max %d (counter units: s_memtime ticks)" % (np.percentile(s0 - s0.min(), 50), (s0 - s0.min()).max()))


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:2]])
