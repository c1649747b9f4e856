"""Book-keeping shared by the two nn.Module front-ends (network.RtposeVGG, shufflenet.Network):
native plans, packed-weight arenas and the keys that say what an arena was packed from.

* Everything is keyed by (device index, compute dtype): a forward on a second GPU - e.g. a replica
  made by ``nn.DataParallel`` (demo/picture_demo.py:47), which shares these dicts by reference
  because ``replicate`` shallow-copies ``__dict__`` - gets its own arena and plans and never evicts
  or re-packs another device's.  One lock serialises plan creation and weight packing.
* An arena is re-packed when a parameter / buffer changed.  ``Tensor._version`` and ``data_ptr()``
  catch ``load_state_dict``, optimiser steps, ``.to()`` / ``.cuda()``; they do NOT see writes made
  through ``.data`` (``p.data.copy_()``, ``p.data.fill_()`` never bump ``_version``).  So
  ``load_state_dict`` and ``_apply`` also bump an explicit epoch, ``invalidate_weights()`` is public
  for code that mutates ``.data``, and ``always_resync = True`` re-packs on every forward.
"""
import threading

import torch

# Plans (one per input shape) keep their activation workspace alive: ~0.22 GB per 368 x 368 image for
# rtpose_vgg in fp32.  An MI355X has 288 GB, and an evaluation run over mixed-size images wants one
# plan per (batch, padded size) bucket, so the cache is bounded by BYTES per device, not by count.
MAX_WORKSPACE_BYTES_PER_DEVICE = 128 << 30
MAX_PLANS_PER_DEVICE = 256
# ... and by what the device really has: at most this fraction of its total memory goes to cached plans (a smaller or
# shared GPU gets a proportionally smaller cache), and an allocation failure evicts the oldest plans and retries.
MAX_WORKSPACE_FRACTION = 0.45

# attributes that hold device memory, native handles or locks: never copied / pickled (each copy builds its own)
_NATIVE_ATTRS = ('_plans', '_weights', '_weights_key', '_weights_epoch', '_native_lock', '_last_input')


class NativeStateMixin(object):
    def _init_native_state(self):
        self._plans = {}         # (n, h, w, device index, dtype) -> plan
        self._weights = {}       # (device index, dtype) -> packed weight arena (torch tensor)
        self._weights_key = {}   # (device index, dtype) -> what the arena was packed from
        self._weights_epoch = [0]   # boxed: shared with DataParallel replicas like the dicts above
        self._native_lock = threading.RLock()
        self.always_resync = False

    # ---- copy / pickle: copy.deepcopy(model), torch.save(model) and multiprocessing spawn see the parameters and the
    # settings, never the native state (an RLock cannot be pickled, plans and arenas belong to one process and device)
    def __getstate__(self):
        state = self.__dict__.copy()
        for k in _NATIVE_ATTRS:
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        resync = state.get('always_resync', False)
        super(NativeStateMixin, self).__setstate__(state)
        self._init_native_state()
        self.always_resync = resync

    def invalidate_weights(self):
        """Force a re-pack of the native weight arenas on the next forward.  Needed after in-place
        edits through ``.data`` (they leave ``Tensor._version`` untouched); harmless otherwise."""
        with self._native_lock:
            self._weights_epoch[0] += 1
        return self

    def load_state_dict(self, *args, **kwargs):
        r = super(NativeStateMixin, self).load_state_dict(*args, **kwargs)
        self.invalidate_weights()
        return r

    def _apply(self, fn, *args, **kwargs):
        r = super(NativeStateMixin, self)._apply(fn, *args, **kwargs)
        if hasattr(self, "_native_lock"):
            self.invalidate_weights()
        return r

    def _params_key(self, tensors):
        return (self._weights_epoch[0],) + tuple((t._version, t.data_ptr()) for t in tensors)

    def _workspace_cap(self, dev):
        try:
            _, total = torch.cuda.mem_get_info(dev)
            return min(MAX_WORKSPACE_BYTES_PER_DEVICE, int(total * MAX_WORKSPACE_FRACTION))
        except Exception:
            return MAX_WORKSPACE_BYTES_PER_DEVICE

    def _build_plan(self, key, factory):
        """Create a plan with ``factory()`` and cache it under ``key``; when the device cannot hold its workspace
        the oldest cached plans of that device are dropped and the allocation is retried."""
        dev = key[3]
        while True:
            try:
                plan = factory()
                break
            except torch.cuda.OutOfMemoryError:
                mine = [k for k in self._plans if k[3] == dev]
                if not mine:
                    raise
                self._plans.pop(mine[0])
                torch.cuda.empty_cache()
        self._remember_plan(key, plan)
        return plan

    def _remember_plan(self, key, plan):
        dev = key[3]
        mine = [k for k in self._plans if k[3] == dev]          # insertion order: oldest first
        size = lambda p: p.workspace.numel() * p.workspace.element_size()   # noqa: E731
        total = sum(size(self._plans[k]) for k in mine) + size(plan)
        cap = self._workspace_cap(dev)
        while mine and (total > cap or len(mine) >= MAX_PLANS_PER_DEVICE):
            total -= size(self._plans.pop(mine.pop(0)))          # evict plans of THIS device only
        self._plans[key] = plan
