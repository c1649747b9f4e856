"""Host-side mirror of ``lib/network/rtpose_vgg.py`` (reference :60-225).

``get_model('vgg19')`` returns an ``nn.Module`` with the reference's 13
``nn.Sequential`` children (``model0``, ``model{1..6}_{1,2}``) so that the 184
state_dict keys, ``load_state_dict``, ``.cuda()``, ``.float()``, ``.eval()`` and
``nn.DataParallel`` wrapping (demo/picture_demo.py:45-49) keep working, but
``forward`` does not run those children: it hands the input to the native
executor in librtpose_mi355x.so (csrc/net.hip), whose convolutions are the
hand-written fp32-MFMA HIP kernels.  torch is used for device memory and
streams only.

The parameter containers are plain ``nn.Conv2d`` modules (weights OIHW as in the
reference checkpoints); they are re-packed into the kernel layout on the device
whenever a parameter's version counter changes.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _capi
from ._capi import lib, check, ptr, current_stream
from ._native_state import NativeStateMixin

# (cin, cout, k) tables; 'P' = MaxPool2d(2, 2, 0).  reference :69-83, :95-127
_VGG = [(3, 64, 3), (64, 64, 3), 'P', (64, 128, 3), (128, 128, 3), 'P', (128, 256, 3),
        (256, 256, 3), (256, 256, 3), (256, 256, 3), 'P', (256, 512, 3), (512, 512, 3),
        (512, 256, 3), (256, 128, 3)]


def _stage1(nout):
    return [(128, 128, 3)] * 3 + [(128, 512, 1), (512, nout, 1)]


def _stage_t(nout):
    return [(185, 128, 7)] + [(128, 128, 7)] * 4 + [(128, 128, 1), (128, nout, 1)]


def _sequential(table, relu_after_last):
    mods = []
    for i, e in enumerate(table):
        if e == 'P':
            mods.append(nn.MaxPool2d(kernel_size=2, stride=2, padding=0))
            continue
        cin, cout, k = e
        mods.append(nn.Conv2d(cin, cout, kernel_size=k, stride=1, padding=k // 2))
        if relu_after_last or i + 1 < len(table):
            mods.append(nn.ReLU(inplace=True))
    return nn.Sequential(*mods)


class _Plan(object):
    """One native executor instance (fixed N, H, W) + its workspace."""

    def __init__(self, n, h, w, weights, device, dtype=_capi.DTYPE_F32, wino=(-1, -1, 0.0)):
        handle = C.c_void_p()
        opts = _capi.NetOptions.make(dtype, *wino)
        check(lib.rtpose_net_create_opts(n, h, w, C.byref(opts), C.byref(handle)), "rtpose_net_create_opts")
        self.handle = handle
        self.shape = (n, h, w)
        self.dtype = dtype
        self.wino = wino
        ws_bytes = lib.rtpose_net_workspace_bytes(handle)
        self.workspace = torch.empty(ws_bytes // 4 + 64, dtype=torch.float32, device=device)
        check(lib.rtpose_net_bind(handle, ptr(self.workspace), ws_bytes, ptr(weights),
                                  weights.numel() * 4, 1, current_stream()), "rtpose_net_bind")
        self.h3 = h // 2 // 2 // 2
        self.w3 = w // 2 // 2 // 2

    def __del__(self):
        try:
            lib.rtpose_net_destroy(self.handle)
        except Exception:
            pass


_DTYPES = {'fp32': _capi.DTYPE_F32, 'bf16': _capi.DTYPE_BF16, 'bf16x3': _capi.DTYPE_BF16X3}


class _ShapeOnly(object):
    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _capi.RtposeError("rtpose_vgg plans exist only on an MI355X (HIP) device; got %s" % device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())


class RtposeVGG(NativeStateMixin, nn.Module):
    """Drop-in for the module built by reference ``get_model('vgg19')``."""

    def __init__(self):
        super(RtposeVGG, self).__init__()
        # registration order == reference :141-155 (it fixes the state_dict order)
        self.model0 = _sequential(_VGG, True)
        for s in range(1, 7):
            setattr(self, 'model%d_1' % s, _sequential(_stage1(38) if s == 1 else _stage_t(38), False))
        for s in range(1, 7):
            setattr(self, 'model%d_2' % s, _sequential(_stage1(19) if s == 1 else _stage_t(19), False))
        self._initialize_weights_norm()
        self._init_native_state()   # plans / weight arenas per (device, dtype), see _native_state.py
        self.keep_intermediates = True   # reference forward returns all 12 stage outputs
        self.compute_dtype = 'fp32'
        self._wino = (_capi.WINO_DEFAULT, _capi.WINO_DEFAULT, 0.0)

    def set_winograd(self, winograd3=None, winograd7=None, amp_limit=None):
        """Arithmetic of the fp32 convs of plans created from now on (``rtpose_net_options``):
        ``winograd3``: None = library default ('auto' since round 4), False / 0 = direct kernels, True / 1 / 2 =
        F(2x2,3x3), 4 = F(4x4,3x3) forced, 'auto' = per layer F(4x4,3x3) if its amplification estimate is <=
        ``amp_limit``, else F(2x2,3x3);
        ``winograd7``: None = library default ('auto'), 0 = direct, 4 / 6 = F(4,7) / F(6,7) forced, 'auto' = per layer
        the fastest form whose amplification estimate for the loaded filters is <= ``amp_limit`` (default 256).
        All forms read one weight arena; results of different forms differ by rounding only."""
        if winograd3 is None:
            w3 = _capi.WINO_DEFAULT
        elif winograd3 == 'auto':
            w3 = _capi.WINO3_AUTO
        elif winograd3 in (1, 2):        # (True == 1)
            w3 = 1
        elif winograd3 in (0, 4):        # (False == 0)
            w3 = int(winograd3)
        else:
            raise ValueError("winograd3 must be None, False / 0, True / 1 / 2, 4 or 'auto'")
        if winograd7 is None:
            w7 = _capi.WINO_DEFAULT
        elif winograd7 == 'auto':
            w7 = _capi.WINO7_AUTO
        elif winograd7 in (0, 4, 6):
            w7 = int(winograd7)
        else:
            raise ValueError("winograd7 must be None, 0, 4, 6 or 'auto'")
        self._wino = (w3, w7, float(amp_limit or 0.0))
        return self

    def conv_numerics(self, plan):
        """[(state_dict prefix, form, (amp F(2x2,3x3), amp F(4,7), amp F(6,7), amp F(4x4,3x3)))] of a plan: form 0 =
        direct kernel, 3 = F(2x2,3x3), 43 = F(4x4,3x3), 4 / 6 = F(m,7); amp = rtpose_winograd_amplification of the
        loaded filters (0 = n/a)."""
        out = []
        form = C.c_int()
        amp = (C.c_float * 4)()
        for i, (nm, _) in enumerate(self._convs()):
            check(lib.rtpose_net_conv_numerics(plan.handle, i, C.byref(form), amp, current_stream()))
            out.append((nm, form.value, tuple(amp)))
        return out

    def device_status(self, plan):
        """Device-side error word of a plan (0 = fine; synchronises the stream)."""
        word = C.c_int()
        check(lib.rtpose_net_device_status(plan.handle, C.byref(word), current_stream()))
        return word.value

    def device_status_async(self, plan, pinned_word):
        """Queue the copy of the plan's device error word into `pinned_word` (a pinned int32 tensor of one element)
        on the current stream, without waiting; read it after an event recorded later on that stream."""
        check(lib.rtpose_net_device_status_async(plan.handle, pinned_word.data_ptr(), current_stream()))

    def set_compute_dtype(self, dtype):
        """'fp32' (reference arithmetic, v_mfma_f32_32x32x2_f32), 'bf16' (BASELINE config 3:
        bf16 operands, fp32 accumulate, v_mfma_f32_32x32x16_bf16) or 'bf16x3' (every fp32
        operand split into two bf16s, three bf16 MFMAs per product: fp32-grade maps from the
        16x faster pipe).  Parameters, inputs and outputs stay fp32 tensors either way."""
        if dtype not in _DTYPES:
            raise ValueError("compute dtype must be one of %s" % (sorted(_DTYPES),))
        self.compute_dtype = dtype
        return self

    def _initialize_weights_norm(self):
        # reference :200-222: N(0, 0.01) weights, zero bias
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.01)
                nn.init.constant_(m.bias, 0.0)

    # ---- native side -------------------------------------------------------
    def _convs(self):
        """Conv modules in the native executor's index order (== state_dict order)."""
        out = []
        names = ['model0'] + ['model%d_1' % s for s in range(1, 7)] + ['model%d_2' % s for s in range(1, 7)]
        for nm in names:
            seq = getattr(self, nm)
            for idx, m in enumerate(seq):
                if isinstance(m, nn.Conv2d):
                    out.append(('%s.%d' % (nm, idx), m))
        return out

    def _sync_weights(self, plan, device):
        convs = self._convs()
        wkey = (device.index, plan.dtype)
        key = self._params_key([t for _, m in convs for t in (m.weight, m.bias)])
        if key == self._weights_key.get(wkey) and not self.always_resync:
            return
        n = lib.rtpose_net_num_convs(plan.handle)
        if n != len(convs):
            raise _capi.RtposeError("native plan has %d convs, module has %d" % (n, len(convs)))
        name = C.create_string_buffer(64)
        co, ci, k = C.c_int(), C.c_int(), C.c_int()
        stream = current_stream()
        for i, (nm, m) in enumerate(convs):
            check(lib.rtpose_net_conv_info(plan.handle, i, name, 64, C.byref(co), C.byref(ci), C.byref(k)))
            if name.value.decode() != nm or tuple(m.weight.shape) != (co.value, ci.value, k.value, k.value):
                raise _capi.RtposeError("conv %d mismatch: native %s vs module %s" % (i, name.value, nm))
            w = m.weight.detach()
            b = m.bias.detach()
            if w.device != device or w.dtype != torch.float32 or not w.is_contiguous():
                w = w.to(device=device, dtype=torch.float32).contiguous()
            if b.device != device or b.dtype != torch.float32 or not b.is_contiguous():
                b = b.to(device=device, dtype=torch.float32).contiguous()
            check(lib.rtpose_net_load_conv(plan.handle, i, ptr(w), ptr(b), stream), "rtpose_net_load_conv")
        torch.cuda.current_stream().synchronize()  # temporaries above may be freed
        self._weights_key[wkey] = key

    def _finalize(self, plan):
        # fixes the per-layer forms of an 'auto' plan from the filters in the arena (no-op otherwise)
        check(lib.rtpose_net_finalize_weights(plan.handle, current_stream()), "rtpose_net_finalize_weights")

    def plan_for(self, x):
        if not x.is_cuda:
            raise _capi.RtposeError(
                "rtpose_vgg forward runs only on an MI355X (HIP) device tensor; got a %s tensor — "
                "there is deliberately no CPU fallback" % x.device)
        n, c, h, w = x.shape
        if c != 3:
            raise _capi.RtposeError("expected NCHW input with 3 channels")
        return self.plan_for_shape(n, h, w, x.device)

    def plan_for_shape(self, n, h, w, device):
        """The executor instance for N x 3 x H x W inputs on `device` (created on first use);
        for callers that fill the plan's input buffer themselves (rtpose_preprocess_u8)."""
        x = _ShapeOnly(device)
        dtype = _DTYPES[self.compute_dtype]
        wino = getattr(self, '_wino', (_capi.WINO_DEFAULT, _capi.WINO_DEFAULT, 0.0))
        key = (n, h, w, x.device.index, dtype, wino)
        with self._native_lock, torch.cuda.device(x.device):
            plan = self._plans.get(key)
            if plan is None:
                wkey = (x.device.index, dtype)
                weights = self._weights.get(wkey)
                if weights is None:
                    probe = C.c_void_p()
                    check(lib.rtpose_net_create_ex(1, 8, 8, dtype, C.byref(probe)))
                    wb = lib.rtpose_net_weight_bytes(probe)
                    lib.rtpose_net_destroy(probe)
                    weights = torch.zeros(wb // 4 + 64, dtype=torch.float32, device=x.device)
                    self._weights[wkey] = weights
                    self._weights_key.pop(wkey, None)
                plan = self._build_plan(key, lambda: _Plan(n, h, w, weights, x.device, dtype, wino))
            self._sync_weights(plan, x.device)
            self._finalize(plan)
        return plan

    def forward_native(self, x, keep_intermediates=False):
        """Enqueue the forward; returns the plan (outputs stay in its workspace)."""
        if not x.is_cuda:
            self.plan_for(x)  # raises: no CPU fallback
        with torch.cuda.device(x.device):
            plan = self.plan_for(x)
            xin = x.detach()
            if xin.dtype != torch.float32 or not xin.is_contiguous():
                xin = xin.float().contiguous()
            check(lib.rtpose_net_set_keep_intermediates(plan.handle, 1 if keep_intermediates else 0))
            check(lib.rtpose_net_forward(plan.handle, ptr(xin), current_stream()), "rtpose_net_forward")
            self._last_input = xin  # keep alive until the stream has consumed it
        return plan

    def read_output(self, plan, which):
        n = plan.shape[0]
        c = 38 if which % 2 == 0 else 19
        out = torch.empty((n, c, plan.h3, plan.w3), dtype=torch.float32, device=plan.workspace.device)
        check(lib.rtpose_net_read_output(plan.handle, which, ptr(out), current_stream()), "rtpose_net_read_output")
        return out

    def output_view(self, plan, which):
        """(base pointer, Layout, C, H, W) of the final PAF (0) / heat-map (1), in place."""
        base = C.c_void_p()
        lay = _capi.Layout()
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        check(lib.rtpose_net_output_view(plan.handle, which, C.byref(base), C.byref(lay), C.byref(c),
                                         C.byref(h), C.byref(w)))
        return base, lay, c.value, h.value, w.value

    def forward(self, x):
        """reference :158-198 — returns ((out6_1, out6_2), saved_for_loss[12]), NCHW fp32."""
        if not x.is_cuda:
            self.plan_for(x)  # raises: no CPU fallback
        with torch.cuda.device(x.device):
            plan = self.forward_native(x, keep_intermediates=self.keep_intermediates)
            if self.keep_intermediates:
                saved = [self.read_output(plan, i) for i in range(12)]
            else:
                last = [self.read_output(plan, 10), self.read_output(plan, 11)]
                saved = [None] * 10 + last
        return (saved[10], saved[11]), saved


def get_model(trunk='vgg19'):
    """reference lib/network/rtpose_vgg.py:60.  Only the VGG19 trunk exists (the
    reference's 'mobilenet' branch never registers block0 and cannot run)."""
    if trunk != 'vgg19':
        raise ValueError("only trunk='vgg19' is implemented (reference 'mobilenet' branch is dead code)")
    return RtposeVGG()


def use_vgg(model):  # reference :235-251 downloads ImageNet weights; no network here
    raise RuntimeError("use_vgg() needs network access to fetch vgg19-dcbb9e9d.pth; load a state_dict instead")
