import sys, importlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
pkg = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd")
capi = pkg._capi
import test_conv_gpu as T
dev = torch.device("cuda", 0)
outs, refs = T._run_conv(capi, dev, 2, 46, 46, 128, 128, 7, 1, 0, 3, 3, seed=5, winograd=True)
o, r = outs[0], refs[0]
d = (o - r).abs()
print("max err", d.max().item(), "ref max", r.abs().max().item())
print("err by column block of 32:", [round(d[:, i:i+32].max().item(), 4) for i in range(0, 128, 32)])
print("err by x mod 6:", [round(d[:, :, :, i::6].max().item(), 4) for i in range(6)])
print("err by row (first 12):", [round(d[:, :, y].max().item(), 3) for y in range(12)])
print("err by image:", [round(d[n].max().item(), 3) for n in range(2)])
