"""ORACLE (test infrastructure, not product code): CPU restatement of the
rtpose_vgg forward pass, reference lib/network/rtpose_vgg.py:158-198.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  It restates the reference graph as plain torch-CPU fp32 functional
calls driven by a state_dict with the reference's key names:

  model0            : 12 x (conv3x3 + ReLU), MaxPool2d(2,2,0) after convs 2, 4, 8
                      (reference :69-83, :39-56)
  model1_{1,2}      : 3 x (conv3x3+ReLU), conv1x1+ReLU, conv1x1      (:95-105, :13-36)
  model{2..6}_{1,2} : 5 x (conv7x7+ReLU), conv1x1+ReLU, conv1x1      (:108-127)
  stage input s>=2  : cat([L1, L2, out1], 1)                         (:165-189)

Pinned against the reference module itself (imported from /root/reference in the
build container) by oracle/make_golden.py -> tests/golden/net_small.npz; the
third-party arithmetic underneath both is ATen's CPU conv2d (torch 2.10).
"""
import torch
import torch.nn.functional as F

VGG_CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25]
VGG_POOL_AFTER = {2, 7, 16}          # Sequential indices of the convs followed by ReLU+pool


def _conv(sd, prefix, x, relu):
    w = sd[prefix + '.weight']
    b = sd[prefix + '.bias']
    y = F.conv2d(x, w, b, stride=1, padding=w.shape[-1] // 2)
    return F.relu(y) if relu else y


def trunk(sd, x):
    for idx in VGG_CONV_IDX:
        x = _conv(sd, 'model0.%d' % idx, x, True)
        if idx in VGG_POOL_AFTER:
            x = F.max_pool2d(x, kernel_size=2, stride=2, padding=0)
    return x


def branch(sd, name, x, nconv):
    for i in range(nconv):
        x = _conv(sd, '%s.%d' % (name, 2 * i), x, relu=(i + 1 < nconv))
    return x


def forward(sd, x):
    """Returns ((out6_1, out6_2), saved_for_loss) like the reference forward."""
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    x = x.detach().float().cpu()
    with torch.no_grad():
        out1 = trunk(sd, x)
        saved = []
        inp = out1
        for s in range(1, 7):
            n = 5 if s == 1 else 7
            l1 = branch(sd, 'model%d_1' % s, inp, n)
            l2 = branch(sd, 'model%d_2' % s, inp, n)
            saved += [l1, l2]
            inp = torch.cat([l1, l2, out1], 1)
    return (saved[-2], saved[-1]), saved


def he_init_state_dict(model, seed=0):
    """Seeded Kaiming weights + N(0, 0.05) biases (the reference init, N(0, 0.01)
    with zero bias, drives the outputs to ~5e-11 — useless for a parity test)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in model.state_dict().items():
        if k.endswith('.weight'):
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            sd[k] = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
        else:
            sd[k] = torch.randn(v.shape, generator=g) * 0.05
    return sd


# ---- bf16 restatement (BASELINE config 3: "bf16 ... single MI355X") -------------------------
# The reference has no reduced-precision path, so there is nothing to pin this against but its
# own definition: operands (activations entering a conv, weights) are rounded to bf16
# (round-to-nearest-even), products are exact, accumulation / bias / ReLU / max-pool run in
# fp32, every activation that feeds another conv is rounded to bf16 again, and the final
# stage-6 maps stay fp32.  "parity unpinned" w.r.t. the reference; pinned w.r.t. forward()
# above by tolerance (tests/test_bf16_gpu.py).
def _rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _conv_bf16(sd, prefix, x, relu, round_out=True):
    w = _rb(sd[prefix + '.weight'])
    b = sd[prefix + '.bias']
    # float64 accumulation: the order-independent value every fp32 accumulation order approximates
    y = F.conv2d(x.double(), w.double(), b.double(), stride=1, padding=w.shape[-1] // 2).float()
    if relu:
        y = F.relu(y)
    return _rb(y) if round_out else y


def forward_bf16_emulated(sd, x):
    """Same graph as forward() under the bf16-operand / fp32-accumulate contract of
    csrc/conv_mfma_bf16.hip.  Returns ((out6_1, out6_2), saved_for_loss); stages 1-5 are the
    bf16-rounded values the next stage consumed, stage 6 is unrounded fp32."""
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    x = _rb(x.detach().float().cpu())
    with torch.no_grad():
        h = x
        for idx in VGG_CONV_IDX:
            h = _conv_bf16(sd, 'model0.%d' % idx, h, True)
            if idx in VGG_POOL_AFTER:   # max commutes with the monotone rounding
                h = F.max_pool2d(h, kernel_size=2, stride=2, padding=0)
        out1 = h
        saved = []
        inp = out1
        for s in range(1, 7):
            n = 5 if s == 1 else 7
            outs = []
            for br in (1, 2):
                t = inp
                for i in range(n):
                    last = i + 1 == n
                    t = _conv_bf16(sd, 'model%d_%d.%d' % (s, br, 2 * i), t, relu=not last,
                                   round_out=not (last and s == 6))
                outs.append(t)
            saved += outs
            inp = torch.cat([outs[0], outs[1], out1], 1)
    return (saved[-2], saved[-1]), saved


# ---- bf16x3 restatement ("split" operands: fp32-grade results from the bf16 matrix pipe) -------
# Every operand v is carried as hi + lo with hi = bf16(v), lo = bf16(v - hi); a product is
# hi*hi + hi*lo + lo*hi (lo*lo dropped); accumulation, bias, ReLU, max-pool in fp32; activations
# are re-split between layers (i.e. rounded to hi + lo), the stage-6 maps stay fp32.
# "parity unpinned" w.r.t. the reference (which has no such path): the product's own claim is
# the fp32 contract, checked against forward() above.
def _split(t):
    hi = _rb(t)
    lo = _rb(t - hi)
    return hi, lo


def _conv_x3(sd, prefix, x, relu, round_out=True):
    w = sd[prefix + '.weight']
    b = sd[prefix + '.bias'].double()
    pad = w.shape[-1] // 2
    xh, xl = (t.double() for t in _split(x))
    wh, wl = (t.double() for t in _split(w))
    y = (F.conv2d(xh, wh, b, padding=pad) + F.conv2d(xh, wl, None, padding=pad)
         + F.conv2d(xl, wh, None, padding=pad)).float()
    if relu:
        y = F.relu(y)
    if round_out:
        hi, lo = _split(y)
        y = hi + lo
    return y


def forward_bf16x3_emulated(sd, x):
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    x = x.detach().float().cpu()
    with torch.no_grad():
        h = x
        for idx in VGG_CONV_IDX:
            h = _conv_x3(sd, 'model0.%d' % idx, h, True)
            if idx in VGG_POOL_AFTER:
                h = F.max_pool2d(h, kernel_size=2, stride=2, padding=0)
        out1 = h
        saved = []
        inp = out1
        for s in range(1, 7):
            n = 5 if s == 1 else 7
            outs = []
            for br in (1, 2):
                t = inp
                for i in range(n):
                    last = i + 1 == n
                    t = _conv_x3(sd, 'model%d_%d.%d' % (s, br, 2 * i), t, relu=not last,
                                 round_out=not (last and s == 6))
                outs.append(t)
            saved += outs
            inp = torch.cat([outs[0], outs[1], out1], 1)
    return (saved[-2], saved[-1]), saved
