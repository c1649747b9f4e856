import numpy as np, glob, os
base = {}
for f in sorted(glob.glob("gpurun_out/wd_0_*.npy")):
    key = f.split("wd_0_")[1]
    a = np.load(f)
    for mode in ("3", "7"):
        b = np.load("gpurun_out/wd_%s_%s" % (mode, key))
        print(key, "mode", mode, "max|diff| %.3e  max|ref| %.3f" % (np.abs(a - b).max(), np.abs(a).max()))
