"""Host-side mirror of ``lib/network/rtpose_shufflenetV2.py`` (reference :22-148): the
ShuffleNetV2 x1.0 pose network of BASELINE config 4.

``Network(width_multiplier=1.0)`` builds the module hierarchy the reference builds once its
missing ``network.slim`` helpers are given the semantics inferred from its call sites
(**parity unpinned**: slim is not in the reference tree) — so state_dict keys are
``network.0.*`` (data/bn), ``network.1.{0,1}.*`` (stem conv+bn), ``network.{3,4,5}.{b}.conv.{i}.{0,1}.*``,
``…conv0.{i}…``, ``network.6.{0,1}.*`` (conv5), ``paf.*``, ``heatmap.*``.  ``forward`` runs the native
executor (csrc/shufflenet.hip); eval-mode BatchNorm is folded into the convs here, on the device,
whenever a parameter/buffer version changes.  Training-mode BN is out of scope (inference path).
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _capi
from ._capi import lib, check, ptr, current_stream
from ._native_state import NativeStateMixin

_WIDTH = {1.0: (116, 232, 464, 1024)}


def _conv_bn(cin, cout, k=1, stride=1, padding=0, groups=1, relu=False):
    mods = [nn.Conv2d(cin, cout, k, stride, padding, 1, groups, bias=False), nn.BatchNorm2d(cout)]
    if relu:
        mods.append(nn.ReLU(inplace=True))
    return nn.Sequential(*mods)


class BasicBlock(nn.Module):
    """reference :22-63 (parameter containers only; the arithmetic runs natively)."""

    def __init__(self, in_channels, out_channels, stride, two_branch):
        super(BasicBlock, self).__init__()
        ch = out_channels // 2
        cin = in_channels if two_branch else ch
        self.conv = nn.Sequential(_conv_bn(cin, ch, 1, relu=True),
                                  _conv_bn(ch, ch, 3, stride, 1, ch),
                                  _conv_bn(ch, ch, 1, relu=True))
        if two_branch:
            self.conv0 = nn.Sequential(_conv_bn(in_channels, in_channels, 3, stride, 1, in_channels),
                                       _conv_bn(in_channels, ch, 1, relu=True))


class _Plan(object):
    def __init__(self, n, h, w, weights, device, dtype=_capi.DTYPE_F32):
        handle = C.c_void_p()
        check(lib.rtpose_shufflenet_create_ex(n, h, w, dtype, C.byref(handle)), "rtpose_shufflenet_create_ex")
        self.handle = handle
        self.shape = (n, h, w)
        self.dtype = dtype
        ws = lib.rtpose_shufflenet_workspace_bytes(handle)
        self.workspace = torch.empty(ws // 4 + 64, dtype=torch.float32, device=device)
        check(lib.rtpose_shufflenet_bind(handle, ptr(self.workspace), ws, ptr(weights), weights.numel() * 4, 1,
                                         current_stream()), "rtpose_shufflenet_bind")

    def __del__(self):
        try:
            lib.rtpose_shufflenet_destroy(self.handle)
        except Exception:
            pass


class Network(NativeStateMixin, nn.Module):
    def __init__(self, width_multiplier=1.0):
        super(Network, self).__init__()
        if width_multiplier not in _WIDTH:
            raise ValueError("only width_multiplier=1.0 is implemented (BASELINE config 4)")
        w = _WIDTH[width_multiplier]
        mods = [nn.BatchNorm2d(3), _conv_bn(3, 24, 3, 2, 1, relu=True), nn.MaxPool2d(3, 2, 0, ceil_mode=True)]
        cin = 24
        for cout, stride, nblocks in ((w[0], 2, 4), (w[1], 1, 8), (w[2], 1, 4)):
            blocks = [BasicBlock(cin, cout, stride, True)]        # `downsample` leak: always two-branch (:113-117)
            blocks += [BasicBlock(cout, cout, 1, False) for _ in range(nblocks - 1)]
            mods.append(nn.Sequential(*blocks))
            cin = cout
        mods.append(_conv_bn(w[2], w[3], 1, relu=True))
        self.paf = nn.Conv2d(w[3], 38, 1)          # registered before `network`, as in the reference (:107-109)
        self.heatmap = nn.Conv2d(w[3], 19, 1)
        self.network = nn.Sequential(*mods)
        for m in self.modules():                                   # reference :124-128
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode='fan_in')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        self._init_native_state()   # plans / weight arenas per (device, dtype), see _native_state.py
        self.compute_dtype = 'fp32'

    def set_compute_dtype(self, dtype):
        """'fp32' or 'bf16' (BASELINE config 4 "fp32 and bf16": 2-byte activations and pointwise
        weights, fp32 accumulation; depthwise / stem weights, biases and outputs stay fp32)."""
        if dtype not in ('fp32', 'bf16'):
            raise ValueError("compute dtype must be 'fp32' or 'bf16'")
        self.compute_dtype = dtype
        return self

    # ---- native side -------------------------------------------------------
    def _module_by_prefix(self, prefix):
        m = self
        for part in prefix.split('.'):
            m = m[int(part)] if part.isdigit() else getattr(m, part)
        return m

    @staticmethod
    def _fold(conv, bn):
        """conv (no bias) followed by eval-mode BN -> (weight, bias)."""
        s = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
        w = conv.weight.detach() * s.view(-1, 1, 1, 1)
        b = bn.bias.detach() - bn.running_mean.detach() * s
        return w.float().contiguous(), b.float().contiguous()

    def _sync_weights(self, plan, device):
        wkey = (device.index, plan.dtype)
        key = self._params_key(list(self.parameters()) + list(self.buffers()))
        if key == self._weights_key.get(wkey) and not self.always_resync:
            return
        name = C.create_string_buffer(96)
        kind, co, ci = C.c_int(), C.c_int(), C.c_int()
        stream = current_stream()
        keep = []
        for i in range(lib.rtpose_shufflenet_num_layers(plan.handle)):
            check(lib.rtpose_shufflenet_layer_info(plan.handle, i, name, 96, C.byref(kind), C.byref(co), C.byref(ci)))
            m = self._module_by_prefix(name.value.decode())
            if kind.value == 0:                       # input BatchNorm2d(3) as scale/shift
                s = m.weight.detach() / torch.sqrt(m.running_var.detach() + m.eps)
                w, b = s.float().contiguous(), (m.bias.detach() - m.running_mean.detach() * s).float().contiguous()
            elif isinstance(m, nn.Conv2d):            # heads: plain conv with bias
                w, b = m.weight.detach().float().contiguous(), m.bias.detach().float().contiguous()
            else:                                     # Sequential(conv, bn[, relu])
                w, b = self._fold(m[0], m[1])
            w, b = w.to(device), b.to(device)
            exp = {1: (co.value, ci.value, 3, 3), 2: (co.value, 1, 3, 3), 3: (co.value, ci.value, 1, 1)}.get(kind.value)
            if exp and tuple(w.shape) != exp:
                raise _capi.RtposeError("layer %s: weight %s, native expects %s" % (name.value, tuple(w.shape), exp))
            keep += [w, b]
            check(lib.rtpose_shufflenet_load(plan.handle, i, ptr(w), ptr(b), stream), "rtpose_shufflenet_load")
        torch.cuda.current_stream().synchronize()
        self._weights_key[wkey] = key

    def plan_for(self, x):
        if not x.is_cuda:
            raise _capi.RtposeError("ShuffleNetV2 forward runs only on an MI355X (HIP) device tensor (no CPU fallback)")
        if self.training:
            raise _capi.RtposeError("native path implements eval-mode BatchNorm only: call .eval()")
        n, c, h, w = x.shape
        dtype = _capi.DTYPE_BF16 if self.compute_dtype == 'bf16' else _capi.DTYPE_F32
        key = (n, h, w, x.device.index, dtype)
        with self._native_lock, torch.cuda.device(x.device):
            plan = self._plans.get(key)
            if plan is None:
                wkey = (x.device.index, dtype)
                weights = self._weights.get(wkey)
                if weights is None:
                    probe = C.c_void_p()
                    check(lib.rtpose_shufflenet_create_ex(1, 64, 64, dtype, C.byref(probe)))
                    wb = lib.rtpose_shufflenet_weight_bytes(probe)
                    lib.rtpose_shufflenet_destroy(probe)
                    weights = torch.zeros(wb // 4 + 64, dtype=torch.float32, device=x.device)
                    self._weights[wkey] = weights
                plan = self._build_plan(key, lambda: _Plan(n, h, w, weights, x.device, dtype))
                self._weights_key.pop(wkey, None)   # a new plan re-binds the maps; the reload is cheap
            self._sync_weights(plan, x.device)
        return plan

    def forward_native(self, x):
        plan = self.plan_for(x)
        with torch.cuda.device(x.device):
            xin = x.detach()
            if xin.dtype != torch.float32 or not xin.is_contiguous():
                xin = xin.float().contiguous()
            check(lib.rtpose_shufflenet_forward(plan.handle, ptr(xin), current_stream()), "rtpose_shufflenet_forward")
            self._last_input = xin
        return plan

    def forward(self, x):
        """reference :144-148 -> ([PAF, HEAT], [PAF, HEAT]), NCHW fp32."""
        plan = self.forward_native(x)
        n = x.shape[0]
        hm, wm = C.c_int(), C.c_int()
        check(lib.rtpose_shufflenet_output_view(plan.handle, 0, None, None, None, C.byref(hm), C.byref(wm)))
        outs = []
        for which, ch in ((0, 38), (1, 19)):
            t = torch.empty((n, ch, hm.value, wm.value), dtype=torch.float32, device=x.device)
            check(lib.rtpose_shufflenet_read_output(plan.handle, which, ptr(t), current_stream()))
            outs.append(t)
        return [outs[0], outs[1]], [outs[0], outs[1]]
