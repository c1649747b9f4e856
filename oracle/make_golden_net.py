"""Generates tests/golden/net_small.npz from the REFERENCE module itself.

Run in the build container only (needs /root/reference):
    python oracle/make_golden_net.py
Imports lib.network.rtpose_vgg.get_model from /root/reference (torch-only
imports), loads the seeded He-init state_dict (oracle/net_oracle.he_init_state_dict,
seed 0), runs a seeded 1x3x48x56 input on the CPU and stores input + the 12 stage
outputs.  Also asserts that the oracle restatement (net_oracle.forward) reproduces
the reference module bit-for-bit on the same input — that is what pins the oracle.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import net_oracle  # noqa: E402

REF = "/root/reference"


def main():
    sys.path.insert(0, REF)
    from lib.network.rtpose_vgg import get_model as ref_get_model  # reference code, imported not copied
    with contextlib.redirect_stdout(io.StringIO()):
        ref = ref_get_model('vgg19')
    sd = net_oracle.he_init_state_dict(ref, seed=0)
    ref.load_state_dict(sd)
    ref.eval()
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(1, 3, 48, 56, generator=g) - 0.5
    with torch.no_grad():
        (paf, heat), saved = ref(x)
    (paf_o, heat_o), saved_o = net_oracle.forward(sd, x)
    worst = max((a - b).abs().max().item() for a, b in zip(saved, saved_o))
    print("oracle vs reference module: max abs diff over 12 stage outputs = %g" % worst)
    assert worst <= 1e-6, "oracle restatement does not reproduce the reference module"
    out = {"x": x.numpy(), "paf": paf.numpy(), "heat": heat.numpy()}
    for i, s in enumerate(saved):
        out["saved%d" % i] = s.numpy()
    path = os.path.join(ROOT, "tests", "golden", "net_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "paf max", float(paf.abs().max()), "heat max", float(heat.abs().max()))


if __name__ == "__main__":
    main()
