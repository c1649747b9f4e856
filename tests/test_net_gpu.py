"""GPU parity of the whole rtpose_vgg forward (native executor, csrc/net.hip)
against the oracle restatement (oracle/net_oracle.py) and the committed golden
vectors produced by the reference module itself (tests/golden/net_small.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "net_small.npz")
ABS_TOL = 1e-3  # BASELINE.json north_star: heatmaps/PAFs within 1e-3 fp32


@pytest.fixture(scope="module")
def model_and_sd(pkg, cuda):
    from oracle import net_oracle
    m = pkg.get_model('vgg19')
    sd = net_oracle.he_init_state_dict(m, seed=0)
    m.load_state_dict(sd)
    m = m.cuda().float().eval()
    return m, sd


@pytest.mark.parametrize("shape", [(2, 3, 64, 72), (1, 3, 56, 40), (1, 3, 61, 75), (3, 3, 40, 88), (5, 3, 104, 24)])
def test_forward_matches_oracle(model_and_sd, cuda, shape):
    from oracle import net_oracle
    m, sd = model_and_sd
    g = torch.Generator().manual_seed(1)
    x = torch.rand(shape, generator=g) - 0.5
    (paf_r, heat_r), saved_r = net_oracle.forward(sd, x)
    with torch.no_grad():
        (paf, heat), saved = m(x.to(cuda))
    assert paf.shape == paf_r.shape and heat.shape == heat_r.shape
    assert len(saved) == 12
    for i, (a, b) in enumerate(zip(saved, saved_r)):
        err = (a.cpu() - b).abs().max().item()
        assert err <= ABS_TOL, "stage output %d: max abs err %g (ref max %g)" % (i, err, b.abs().max().item())
    # much tighter than the contract in practice
    assert (paf.cpu() - paf_r).abs().max().item() <= 2e-4 * max(1.0, paf_r.abs().max().item())


def test_forward_matches_reference_golden(model_and_sd, cuda):
    m, sd = model_and_sd
    z = np.load(GOLD)
    x = torch.from_numpy(z["x"])
    with torch.no_grad():
        (paf, heat), saved = m(x.to(cuda))
    assert np.abs(paf.cpu().numpy() - z["paf"]).max() <= ABS_TOL
    assert np.abs(heat.cpu().numpy() - z["heat"]).max() <= ABS_TOL
    for i in range(12):
        assert np.abs(saved[i].cpu().numpy() - z["saved%d" % i]).max() <= ABS_TOL


def test_cpu_tensor_fails_loudly(model_and_sd, capi):
    m, _ = model_and_sd
    with pytest.raises(capi.RtposeError):
        m(torch.zeros(1, 3, 64, 64))


def test_state_dict_roundtrip_repacks(model_and_sd, cuda):
    m, sd = model_and_sd
    x = (torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(3)) - 0.5).to(cuda)
    with torch.no_grad():
        (p0, _), _ = m(x)
        sd2 = {k: v * 0.5 for k, v in m.state_dict().items()}
        m.load_state_dict(sd2)
        (p1, _), _ = m(x)
        m.load_state_dict(sd)
        (p2, _), _ = m(x)
    assert not torch.allclose(p0, p1)
    assert torch.equal(p0, p2)


def test_full_size_properties_batch32(model_and_sd, cuda):
    """BASELINE.json configs[1] at full size (32 x 3 x 368 x 368), through size-independent
    properties: (a) image i's maps do not depend on its position / neighbours in the batch
    (bit-exact under a permutation of the batch: strips of the flattened pixel space span image
    boundaries, the shared-gap layout must not leak between images), (b) the run is
    deterministic, (c) one image of the batch equals the same image run alone - and ALL 32 images
    equal the oracle at full size within the 1e-3 contract."""
    from oracle import net_oracle
    m, sd = model_and_sd
    g = torch.Generator().manual_seed(11)
    x = torch.rand(32, 3, 368, 368, generator=g) - 0.5
    perm = torch.randperm(32, generator=g)
    keep = m.keep_intermediates
    m.keep_intermediates = False
    try:
        with torch.no_grad():
            (paf, heat), _ = m(x.to(cuda))
            (paf2, heat2), _ = m(x.to(cuda))
            (paf_p, heat_p), _ = m(x[perm].to(cuda))
            (paf_1, heat_1), _ = m(x[5:6].to(cuda))
    finally:
        m.keep_intermediates = keep
    assert paf.shape == (32, 38, 46, 46) and heat.shape == (32, 19, 46, 46)
    assert torch.equal(paf, paf2) and torch.equal(heat, heat2)
    assert torch.equal(paf_p, paf[perm.to(cuda)]) and torch.equal(heat_p, heat[perm.to(cuda)])
    assert torch.equal(paf_1[0], paf[5]) and torch.equal(heat_1[0], heat[5])
    worst = 0.0
    for i0 in range(0, 32, 8):          # the CPU oracle, 8 images at a time
        (paf_r, heat_r), _ = net_oracle.forward(sd, x[i0:i0 + 8])
        e_paf = (paf[i0:i0 + 8].cpu() - paf_r).abs().amax(dim=(1, 2, 3))
        e_heat = (heat[i0:i0 + 8].cpu() - heat_r).abs().amax(dim=(1, 2, 3))
        worst = max(worst, e_paf.max().item(), e_heat.max().item())
        assert e_paf.max().item() <= ABS_TOL and e_heat.max().item() <= ABS_TOL, (i0, e_paf, e_heat)
    print("config 2, all 32 images vs the oracle: worst |err| %.2e" % worst)


def test_an_images_maps_are_the_same_bits_in_every_batch_size(model_and_sd, cuda):
    """Which launch form a conv runs in depends on the batch: small grids (one image), whole rounds of persistent blocks plus a
    left-over in the 16 x 16 form or - 25..50 % of a round - in half tiles, split tiles of the 7x7 kernel.  All forms sum in
    the same order: images 0, 11 and 39 of a 40-image batch give the bits of their batch-1 runs in batches of 5, 12, 24 and
    40 (368 x 368, fp32 default plan)."""
    m, _ = model_and_sd
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(40, 3, 368, 368, generator=g) - 0.5).to(cuda)
    keep = m.keep_intermediates
    m.keep_intermediates = False
    try:
        with torch.no_grad():
            ref = {}
            for i in (0, 11, 39):
                (p, h), _ = m(x[i:i + 1])
                ref[i] = (p.clone(), h.clone())
            for n in (5, 12, 24, 40):
                (p, h), _ = m(x[:n])
                for i, (pr, hr) in ref.items():
                    if i < n:
                        assert torch.equal(p[i:i + 1], pr) and torch.equal(h[i:i + 1], hr), (n, i)
    finally:
        m.keep_intermediates = keep

