"""Developer tool: phase time stamps (s_memtime, wave 0) of the one-launch ShuffleNetV2 unit kernel (csrc/unit_bf16.hip)
for the LAST unit launch of a bf16 forward at 128 x 368 x 368 (network.5.3: h = 232, K1 = 256, NF1 = NF2 = 2).  Needs the
-DRTPOSE_EXP_TIMELINE_UNIT build:
    ONLY=unit_bf16 OUT=tools/exp/lib_tlu.so tools/build_dev.sh -DRTPOSE_EXP_TIMELINE_UNIT
    RTPOSE_LIB_PATH=tools/exp/lib_tlu.so python tools/timeline_unit.py
Stamps per tile (wave 0): 0 loop top, 1 after B0, 2 GEMM 1 issued, 3 after B1 (conv.2's fragments + round 0 requested), 4 T1
written, 5 after B2, 6 depthwise conv done + round 0 stored, 7 after B3, 8 = 9 GEMM 2 issued, 10 round 1 stored, 11 epilogue done."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd")
synth = importlib.import_module(pkg.__name__ + ".synth")
sn = importlib.import_module(pkg.__name__ + ".shufflenet")
lib = pkg._capi.lib


def main(n=128):
    m = sn.Network(1.0)
    m.load_state_dict(synth.seeded_shufflenet_state_dict(m, 0))
    m = m.cuda().eval()
    m.set_compute_dtype('bf16')
    x = (torch.rand(n, 3, 368, 368, generator=torch.Generator().manual_seed(0)) - 0.5).cuda()
    for _ in range(3):
        m.forward_native(x)
    torch.cuda.synchronize()
    buf = torch.zeros(256 * 20 * 16, dtype=torch.int64, device="cuda")
    lib.rtpose_debug_unit_timeline.restype = None
    lib.rtpose_debug_unit_timeline(C.c_void_p(buf.data_ptr()))
    m.forward_native(x)
    torch.cuda.synchronize()
    lib.rtpose_debug_unit_timeline(None)
    t = buf.cpu().numpy().reshape(256, 20, 16)
    ntile = int((t[0, :, 0] > 0).sum())
    names = ["loop top -> B0", "GEMM 1", "w2 preload + B1", "T1 write", "B2", "depthwise + round 0", "B3", "GEMM 2",
             "-", "round 1 store", "epilogue"]
    print("tiles per block %d" % ntile)
    d = np.diff(t[:, 1:ntile - 1, :12].astype(np.int64), axis=2)      # steady-state tiles
    for k, nm in enumerate(names):
        print("  %-24s p50 %7.0f  p90 %7.0f" % (nm, np.percentile(d[:, :, k], 50), np.percentile(d[:, :, k], 90)))
    tot = t[:, 2:ntile - 1, 0].astype(np.int64) - t[:, 1:ntile - 2, 0].astype(np.int64)
    print("  tile period              p50 %7.0f  p90 %7.0f   (s_memtime ticks: 100 MHz? compare with the sum above)"
          % (np.percentile(tot, 50), np.percentile(tot, 90)))


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:2]])
