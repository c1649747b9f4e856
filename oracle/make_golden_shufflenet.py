"""tests/golden/shufflenet_small.npz from the reference Network(1.0) imported UNMODIFIED through
the `network.slim` stub (oracle/shufflenet_oracle.py).  Build container only.  Asserts that the
functional restatement reproduces the reference module bit for bit."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import shufflenet_oracle as so  # noqa: E402


def main():
    ref = so.reference_network()
    sd = so.seeded_state_dict(ref, seed=0)
    ref.load_state_dict(sd)
    ref.eval()
    x = torch.rand(1, 3, 96, 112, generator=torch.Generator().manual_seed(4321)) - 0.5
    with torch.no_grad():
        (paf, heat), _ = ref(x)
    paf_o, heat_o = so.forward(sd, x)
    d = max((paf - paf_o).abs().max().item(), (heat - heat_o).abs().max().item())
    print("restatement vs reference module: max abs diff %g" % d, tuple(paf.shape), "params",
          sum(p.numel() for p in ref.parameters()))
    assert d <= 1e-6
    path = os.path.join(ROOT, "tests", "golden", "shufflenet_small.npz")
    np.savez_compressed(path, x=x.numpy(), paf=paf.numpy(), heat=heat.numpy())
    print("wrote", path, "paf max", float(paf.abs().max()), "heat max", float(heat.abs().max()))


if __name__ == "__main__":
    main()
