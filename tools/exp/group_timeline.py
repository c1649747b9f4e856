"""Developer probe: where group_kernel's time goes (wall_clock64 stamps: start, part offsets + ids, staging, walk, emit).
Needs the -DRTPOSE_EXP_GROUP_TIMELINE build:  ONLY=decode OUT=tools/exp/lib_gtl.so tools/build_dev.sh -DRTPOSE_EXP_GROUP_TIMELINE
    RTPOSE_LIB_PATH=tools/exp/lib_gtl.so python tools/exp/group_timeline.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "pytorch_realtime_multi-person_pose_estimation_amd"
pkg = importlib.import_module(PKG)
dec = importlib.import_module(PKG + ".decode")
synth = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda", 0)
N = 32
heat, paf, _ = synth.make_batch(N, 368, 368, seed=100)
heat, paf = torch.from_numpy(heat).to(dev), torch.from_numpy(paf).to(dev)
for _ in range(3):
    recs = dec.decode_maps(heat, paf)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * N))()
rc = pkg._capi.lib.rtpose_exp_group_timeline(buf, N)
t = np.frombuffer(buf, dtype=np.uint64).reshape(N, 8).astype(np.int64)
t0 = t[:, 0].min()
print("rc", rc, "(stamps in 10 ns ticks of wall_clock64)")
print("image: start, +offsets/ids, +staging, +walk, +emit (us); connections; humans")
for i in range(N):
    d = [(t[i, k] - (t[i, k - 1] if k else t0)) / 100.0 for k in range(5)]
    print("%2d: %6.2f %6.2f %6.2f %6.2f %6.2f   %4d conn  %2d humans" % (i, d[0], d[1], d[2], d[3], d[4], int(t[i, 6]), len(recs[i]["parts"])))
tot = (t[:, 4].max() - t0) / 100.0
print("kernel span %.2f us; mean phases: offsets %.2f staging %.2f walk %.2f emit %.2f us; walk per connection %.3f us" % (
    tot, *(np.mean([(t[:, k] - t[:, k - 1]) / 100.0 for k in range(1, 5)], axis=1)), float(np.mean((t[:, 3] - t[:, 2]) / 100.0 / np.maximum(t[:, 6], 1)))))
