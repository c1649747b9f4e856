import os, sys, importlib, subprocess, json
import numpy as np, torch
sys.path.insert(0, "/root/repo")
mode = sys.argv[1]
os.environ["RTPOSE_WINOGRAD"] = mode
pkg = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd")
synth = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd.synth")
m = pkg.get_model('vgg19'); m.load_state_dict(synth.he_init_state_dict(m, seed=0)); m = m.cuda().float().eval()
out = {}
for (n, h, w) in [(1, 184, 232), (2, 368, 464), (1, 552, 696), (1, 736, 920), (1, 64, 72), (3, 56, 40), (1,120,152)]:
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(n, 3, h, w, generator=g) - 0.5).cuda()
    with torch.no_grad():
        (paf, heat), saved = m(x)
    np.save("/root/repo/gpurun_out/wd_%s_%d_%d_%d.npy" % (mode, n, h, w), np.concatenate([s.cpu().numpy().reshape(-1) for s in saved]))
