"""Developer tool: per-block time stamps of the last fp32 conv launch of kernel size
RTPOSE_TIMELINE_K (default 1) - ShuffleNetV2 forward by default, `vgg` as argv[1] for rtpose_vgg.
Needs the RTPOSE_EXP_TIMELINE build of conv_mfma.hip (tools/exp/lib_ftime.so via RTPOSE_LIB_PATH)."""
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


def main(which="shuffle"):
    if which == "vgg":
        m = pkg.get_model('vgg19')
        m.load_state_dict(synth.he_init_state_dict(m, 0))
        m = m.cuda().eval()
        x = (torch.rand(32, 3, 368, 368) - 0.5).cuda()
    else:
        sn = importlib.import_module(pkg.__name__ + ".shufflenet")
        m = sn.Network(1.0)
        m.load_state_dict(synth.seeded_shufflenet_state_dict(m, 0))
        m = m.cuda().eval()
        x = (torch.rand(128, 3, 368, 368) - 0.5).cuda()
    for _ in range(3):
        m.forward_native(x)
    torch.cuda.synchronize()
    buf = np.zeros((32768, 8), dtype=np.uint64)
    fn = lib.rtpose_debug_timeline32_dump
    fn.restype = C.c_int
    nb = fn(C.c_void_p(buf.ctypes.data), 32768)
    t = buf[:nb].astype(np.int64)
    ok = t[:, 4] > 0
    t = t[ok]
    print("blocks", nb, "with stamps", len(t))
    for name, v in (("prologue", t[:, 1] - t[:, 0]), ("main loop", t[:, 2] - t[:, 1]), ("epilogue", t[:, 3] - t[:, 2]),
                    ("store ack", t[:, 4] - t[:, 3]), ("total", t[:, 4] - t[:, 0])):
        print("%-10s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % (
            name, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))


if __name__ == "__main__":
    main(*sys.argv[1:])
