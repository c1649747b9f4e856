"""ORACLE (test infrastructure): ShuffleNetV2 x1.0 pose network,
reference lib/network/rtpose_shufflenetV2.py (BasicBlock :22-63, Network :80-148).

The reference file imports ``network.slim`` which is NOT in the reference tree, not in
requirements.txt and has no pinned version: **parity unpinned**.  The
helper semantics are inferred from the call sites (:34-38, :42-53, :54, :98, :103):

  conv_bn_relu(name, cin, cout, k=3, stride=1, padding=0, dilation=1, groups=1)
      = Conv2d(bias=False) + BatchNorm2d + ReLU(inplace)
  conv_bn(...)        = the same without the ReLU
  channel_shuffle(name, groups) : [N, g, C/g, H, W] -> permute(0,2,1,3,4) -> flatten
  g_name(name, module): tags and returns the module

`install_slim_stub()` registers that stub so the reference file can be imported
UNMODIFIED (build container only); `forward()` is an independent functional restatement
driven by the resulting state_dict keys, usable on the GPU box.  Quirk kept: `downsample`
leaks out of the first stage (:113-117), so the first block of the two stride-1 stages
is also the two-branch block.
"""
import importlib.util
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_FILE = "/root/reference/lib/network/rtpose_shufflenetV2.py"
WIDTH = (116, 232, 464, 1024)   # width_multiplier 1.0, reference :91
BN_EPS = 1e-5


def install_slim_stub():
    class ChannelShuffle(nn.Module):
        def __init__(self, groups):
            super(ChannelShuffle, self).__init__()
            self.groups = groups

        def forward(self, x):
            n, c, h, w = x.shape
            return x.view(n, self.groups, c // self.groups, h, w).permute(0, 2, 1, 3, 4).reshape(n, c, h, w)

    def g_name(name, m):
        m.g_name = name
        return m

    def conv_bn(name, cin, cout, k=3, stride=1, padding=0, dilation=1, groups=1):
        return g_name(name, nn.Sequential(
            nn.Conv2d(cin, cout, k, stride, padding, dilation, groups, bias=False), nn.BatchNorm2d(cout)))

    def conv_bn_relu(name, cin, cout, k=3, stride=1, padding=0, dilation=1, groups=1):
        return g_name(name, nn.Sequential(
            nn.Conv2d(cin, cout, k, stride, padding, dilation, groups, bias=False), nn.BatchNorm2d(cout),
            nn.ReLU(inplace=True)))

    slim = types.ModuleType("network.slim")
    slim.g_name, slim.conv_bn, slim.conv_bn_relu = g_name, conv_bn, conv_bn_relu
    slim.channel_shuffle = lambda name, groups: g_name(name, ChannelShuffle(groups))
    pkg = types.ModuleType("network")
    pkg.slim = slim
    sys.modules.setdefault("network", pkg)
    sys.modules["network.slim"] = slim
    return slim


def reference_network():
    """The reference Network(1.0), imported unmodified (needs /root/reference)."""
    install_slim_stub()
    spec = importlib.util.spec_from_file_location("ref_rtpose_shufflenetV2", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Network(1.0)


def seeded_state_dict(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in model.state_dict().items():
        if k.endswith("num_batches_tracked"):
            sd[k] = v.clone()
        elif k.endswith("running_var"):
            sd[k] = torch.rand(v.shape, generator=g) + 0.5
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
        elif v.dim() == 4:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            sd[k] = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif k.endswith(".weight"):          # BN gamma (kept < 1 so activations stay O(1))
            sd[k] = torch.rand(v.shape, generator=g) * 0.4 + 0.35
        else:                                # BN beta / head bias
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
    return sd


# ---- functional restatement ------------------------------------------------------
def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, BN_EPS)


def _cbr(sd, p, x, relu, stride=1, padding=0, groups=1):
    y = F.conv2d(x, sd[p + ".0.weight"], None, stride, padding, 1, groups)
    y = _bn(sd, p + ".1", y)
    return F.relu(y) if relu else y


def _shuffle(x, groups=2):
    n, c, h, w = x.shape
    return x.view(n, groups, c // groups, h, w).permute(0, 2, 1, 3, 4).reshape(n, c, h, w)


def _block(sd, p, x, stride, two_branch):
    def conv(z):                                           # reference :31-46
        c = z.shape[1] if two_branch else z.shape[1]
        z = _cbr(sd, p + ".conv.0", z, True)
        z = _cbr(sd, p + ".conv.1", z, False, stride, 1, z.shape[1])
        return _cbr(sd, p + ".conv.2", z, True)
    if not two_branch:                                     # :56-59
        half = x.shape[1] // 2
        x = torch.cat((x[:, :half], conv(x[:, half:])), 1)
    else:                                                  # :60-61, :47-53
        z = _cbr(sd, p + ".conv0.0", x, False, stride, 1, x.shape[1])
        z = _cbr(sd, p + ".conv0.1", z, True)
        x = torch.cat((z, conv(x)), 1)
    return _shuffle(x)


def forward(sd, x):
    """-> (PAF [N,38,h,w], HEAT [N,19,h,w]); reference Network.forward :144-148."""
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    x = x.detach().float().cpu()
    with torch.no_grad():
        x = _bn(sd, "network.0", x)                                   # data/bn
        x = _cbr(sd, "network.1", x, True, 2, 1)                      # stage1/conv 3x3 s2
        x = F.max_pool2d(x, 3, 2, 0, ceil_mode=True)                  # stage1/pool
        for si, (nblocks, stride) in enumerate(((4, 2), (8, 1), (4, 1))):
            for b in range(nblocks):
                x = _block(sd, "network.%d.%d" % (3 + si, b), x, stride if b == 0 else 1, b == 0)
        x = _cbr(sd, "network.6", x, True)                            # conv5
        paf = F.conv2d(x, sd["paf.weight"], sd["paf.bias"])
        heat = F.conv2d(x, sd["heatmap.weight"], sd["heatmap.bias"])
    return paf, heat


# ---- bf16 restatement (BASELINE config 4 "fp32 and bf16") ------------------------------------
# The product's bf16 plan: BatchNorm folded into the conv (fp32), pointwise weights rounded to
# bf16, depthwise / stem weights and all biases fp32, fp32 accumulation, every stored activation
# rounded to bf16 (the passthrough half is a bit copy), heads written fp32.  "parity unpinned"
# like the fp32 restatement (the slim stub defines the reference); checked against forward() by
# tolerance.
def _rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _fold(sd, p):
    s = sd[p + ".1.weight"] / torch.sqrt(sd[p + ".1.running_var"] + BN_EPS)
    return sd[p + ".0.weight"] * s.view(-1, 1, 1, 1), sd[p + ".1.bias"] - sd[p + ".1.running_mean"] * s


def _cbr_bf16(sd, p, x, relu, stride=1, padding=0, groups=1, round_w=True):
    w, b = _fold(sd, p)
    if round_w:
        w = _rb(w)
    y = F.conv2d(x.double(), w.double(), b.double(), stride, padding, 1, groups).float()
    return _rb(F.relu(y) if relu else y)


def _block_bf16(sd, p, x, stride, two_branch):
    def conv(z):
        z = _cbr_bf16(sd, p + ".conv.0", z, True)
        z = _cbr_bf16(sd, p + ".conv.1", z, False, stride, 1, z.shape[1], round_w=False)   # depthwise: fp32 weights
        return _cbr_bf16(sd, p + ".conv.2", z, True)
    if not two_branch:
        half = x.shape[1] // 2
        x = torch.cat((x[:, :half], conv(x[:, half:])), 1)
    else:
        z = _cbr_bf16(sd, p + ".conv0.0", x, False, stride, 1, x.shape[1], round_w=False)
        z = _cbr_bf16(sd, p + ".conv0.1", z, True)
        x = torch.cat((z, conv(x)), 1)
    return _shuffle(x)


def forward_bf16_emulated(sd, x):
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    x = x.detach().float().cpu()
    with torch.no_grad():
        x = _bn(sd, "network.0", x)                                   # fp32 affine on the fp32 image
        x = _cbr_bf16(sd, "network.1", x, True, 2, 1, round_w=False)  # stem: fp32 weights, bf16 output
        x = F.max_pool2d(x, 3, 2, 0, ceil_mode=True)
        for si, (nblocks, stride) in enumerate(((4, 2), (8, 1), (4, 1))):
            for b in range(nblocks):
                x = _block_bf16(sd, "network.%d.%d" % (3 + si, b), x, stride if b == 0 else 1, b == 0)
        x = _cbr_bf16(sd, "network.6", x, True)
        paf = F.conv2d(x.double(), _rb(sd["paf.weight"]).double(), sd["paf.bias"].double()).float()
        heat = F.conv2d(x.double(), _rb(sd["heatmap.weight"]).double(), sd["heatmap.bias"].double()).float()
    return paf, heat
