"""Developer tool: per-chunk time stamps of every wave of the F(4x4,3x3) kernel (csrc/conv_wino4.hip, wino4_f32) over the
first tile of each persistent block, for the LAST launch of one rtpose_vgg forward at 32 x 368 x 368 whose channel counts
are RTPOSE_TIMELINE_W4="cin,cout" (default 256,256 = conv3_2..4).  Needs the -DRTPOSE_EXP_TIMELINE4 build
(ONLY=conv_wino4 tools/build_dev.sh -DRTPOSE_EXP_TIMELINE4, RTPOSE_LIB_PATH).
Stamps (s_memtime, lane 0 of each wave): per chunk 0 = after the chunk's opening barrier, 1 = after its last MFMA was
issued (the wave then waits at the barrier); chunk slot 63 = epilogue begin / end."""
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
    buf = np.zeros((1024, 8, 64, 2), dtype=np.uint64)
    fn = lib.rtpose_debug_timeline_w4_dump
    fn.restype = C.c_int
    nb = fn(C.c_void_p(buf.ctypes.data), 1024)
    t = buf[:nb].astype(np.int64)
    nch = int((t[0, 0, :63, 0] > 0).sum())
    print("blocks %d, chunks per tile %d (counter: s_memtime ticks)" % (nb, nch))
    work = t[:, :, :nch, 1] - t[:, :, :nch, 0]          # a wave's own time in a chunk
    period = t[:, :, 1:nch, 0] - t[:, :, :nch - 1, 0]   # chunk period seen by a wave
    wait = t[:, :, 1:nch, 0] - t[:, :, :nch - 1, 1]     # arrival at the barrier -> start of the next chunk
    pc = lambda v, q: float(np.percentile(v, q))        # noqa: E731
    print("chunk period          p50 %.0f  p90 %.0f" % (pc(period, 50), pc(period, 90)))
    print("wave busy in a chunk  p50 %.0f  p90 %.0f" % (pc(work, 50), pc(work, 90)))
    print("barrier wait + start  p50 %.0f  p90 %.0f" % (pc(wait, 50), pc(wait, 90)))
    for w in range(8):
        print("  wave %d: busy p50 %.0f, wait p50 %.0f" % (w, pc(work[:, w], 50), pc(wait[:, w], 50)))
    # spread of the arrival times of the 8 waves of a block at the barriers
    arr = t[:, :, :nch, 1]
    spread = arr.max(1) - arr.min(1)
    print("arrival spread of a block's 8 waves  p50 %.0f  p90 %.0f" % (pc(spread, 50), pc(spread, 90)))
    ep = t[:, :, 63, 1] - t[:, :, 63, 0]
    print("epilogue  p50 %.0f  p90 %.0f" % (pc(ep, 50), pc(ep, 90)))
    first = t[:, :, 0, 0].min(1)
    print("tile total (first chunk start -> epilogue end) p50 %.0f" % pc(t[:, :, 63, 1].max(1) - first, 50))


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:2]])
