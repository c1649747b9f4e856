"""Developer tool: per-block time stamps of the last 7x7 bf16 conv launch (needs the
RTPOSE_EXP_TIMELINE build: tools/exp_variants_bf16.sh btime; RTPOSE_LIB_PATH=tools/exp/lib_btime.so)."""
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


def main(n=32, hw=368):
    m = pkg.get_model('vgg19')
    m.load_state_dict(synth.he_init_state_dict(m, 0))
    m = m.cuda().eval()
    m.set_compute_dtype('bf16')
    x = (torch.rand(n, 3, hw, hw) - 0.5).cuda()
    for _ in range(3):
        m.forward_native(x)
    torch.cuda.synchronize()
    buf = np.zeros((8192, 8), dtype=np.uint64)
    fn = lib.rtpose_debug_timeline_dump
    fn.restype = C.c_int
    nb = fn(C.c_void_p(buf.ctypes.data), 8192)
    t = buf[:nb].astype(np.int64)
    t0 = t[:, 0].min()
    start, pro, loop, epi, ack = (t[:, 0] - t0, t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3])
    hw_id = t[:, 5]
    print("blocks", nb, "span (ticks)", (t[:, 4].max() - t0))
    setup, fill, bar = t[:, 6] - t[:, 0], t[:, 7] - t[:, 6], t[:, 1] - t[:, 7]
    for name, v in (("pro:setup", setup), ("pro:fill", fill), ("pro:barrier", bar), ("prologue", pro), ("main loop", loop),
                    ("epilogue", epi), ("store ack", ack)):
        v = v[t[:, 4] > 0]
        print("%-10s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % (
            name, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
    order = np.argsort(start)
    print("start times of the first 12 / blocks 510..522 / last 8 (ticks):")
    print(start[order][:12], start[order][510:522], start[order][-8:])
    print("end times (sorted) tail:", np.sort(t[:, 4] - t0)[-8:])
    # per-CU occupancy: cu id bits of HW_ID (gfx9: [11:8] cu, [14:13] se?...) - print raw distribution
    print("distinct hw ids:", len(np.unique(hw_id & 0xffffff0)))
    np.save("gpurun_out/timeline_bf16.npy", t)


if __name__ == "__main__":
    main()
