"""Batched end-to-end path: device image batch -> rtpose_vgg forward -> pose
decode -> compact per-image records on the host.

This is what the reference does one image at a time in
evaluate/coco_eval.py:270-272 / demo/picture_demo.py:57-61
(get_outputs -> paf_to_pose_cpp), restructured for the GPU: the heat-maps and
PAFs never leave HBM (the decoder reads them in place, in the layout the last
conv wrote), and only the fixed-capacity result records cross PCIe.
"""
import ctypes as C

import numpy as np

from . import _capi, decode as dec
from ._capi import lib, check, ptr, current_stream


def _raise_on_device_error(model, plan):
    """The persistent 7x7 launches hand split tiles from block to block through device flags with a bounded wait
    (csrc/conv_wino7.hip); a wait that ran out leaves bit 0 in the plan's device error word and the maps of that launch
    are invalid.  Checked where the host has just synchronised anyway (records fetched): never silently wrong."""
    status = getattr(model, 'device_status', None)
    if status is None or getattr(plan, 'dtype', 0) != _capi.DTYPE_F32:
        return
    word = status(plan)
    if word:
        raise _capi.RtposeError("device error word %d: a split-tile hand-over of a persistent 7x7 launch timed out "
                                "(shared / CU-masked device?); the maps of this batch are invalid" % word)


class SideDecoder(object):
    """The decoder's launches and the D2H of their records on a SECOND stream, behind an event, so that the next forward
    is queued at once on the compute stream (round 5).  The decoder's four launches are latency-bound chains on small grids:
    on the compute stream they are 0.9 % of an fp32 step and 2.4 % of a bf16 one in which nothing else runs.

    The maps' buffer belongs to the plan, so the plan carries the guard (rtpose_net_set_output_guard): from the moment a
    decode is enqueued here, EVERY later forward of that plan - this object's next batch, PoseEstimator.__call__, a bare
    model(x), another estimator over the same module - waits for this decode's last read of the maps before it rewrites
    them (library default: in front of its whole launch list; RTPOSE_GUARD_FINE=1: only where it first writes the buffer.
    DESIGN.md 3.3 has the history: the decoder's kernels are built without packed-fp32 VALU instructions because those
    returned wrong values beside the bf16 plan's kernels).  The guard stays installed until the next decode replaces it or
    close() removes it - ONE guard per plan: two pipelined consumers decoding the maps of the same plan at the same time
    are not supported (the later decode's event replaces the earlier one's).  The plan's device error word (persistent 7x7 hand-over, fp32 plans) rides in front of every decode and
    is checked in wait().  Two slots (record block, pinned copy, events, error word)."""

    def __init__(self, config):
        self.config = config
        self.stream = None
        self.slots = [None, None]
        self.last = None            # slot of the decode enqueued last
        self._plans = {}            # id -> plan whose guard points at one of this object's events

    def guarded(self, plan, forward):
        """Run `forward()` (which enqueues the plan's launches on the current stream).  The guard a previous decode() left on
        the plan makes it wait for that decode's final read of the maps; nothing to do here any more - kept as the one place
        the pipelined callers enqueue their forward through."""
        return forward()

    def close(self):
        """Remove this object's guard from the plans it was installed on (their events die with the slots)."""
        for plan in self._plans.values():
            try:
                lib.rtpose_net_set_output_guard(plan.handle, None)
            except Exception:   # noqa: BLE001  (interpreter shutdown)
                pass
        self._plans = {}

    def __del__(self):
        self.close()

    def decode(self, i, maps, n, device, max_peaks_per_part, max_humans, post=None, plan=None, model=None):
        """Enqueue decode + record D2H of slot i (0 / 1) behind everything queued on the current stream so far.
        maps = (hbase, lheat, pbase, lpaf, h, w).  ``post(result_block)``, if given, runs on the side stream after the
        decode and returns the device tensor whose copy wait() hands out (bench.py: the RCCL gather of the ranks' blocks).
        ``plan`` (+ ``model``): the plan whose output buffer `maps` points into - it gets the guard and its device error
        word is read."""
        import torch
        if plan is None:
            raise _capi.RtposeError("SideDecoder.decode: plan= is required (the plan whose output buffer the maps live in "
                                    "carries the guard that keeps its next forward off them)")
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        hbase, lheat, pbase, lpaf, h, w = maps
        slot = self.slots[i]
        key = (n, device.index, max_peaks_per_part, max_humans)
        if slot is None or slot["key"] != key:
            if slot is not None:
                # the old buffers may still be read / written by side-stream work queued earlier: the caching allocator
                # hands a dropped block back to the COMPUTE stream's pool at once (round-5 advisor finding)
                slot["done"].synchronize()
                if self.last is slot:
                    self.close()        # a guard must not outlive its event
                    self.last = None
            cfg = dec.make_cfg(self.config, max_peaks_per_part, max_humans)
            slot = {"key": key, "bufs": dec.DecodeBuffers(cfg, n, device), "maps": torch.cuda.Event(),
                    "dec_done": torch.cuda.Event(), "done": torch.cuda.Event(), "host": None,
                    "err": torch.zeros(1, dtype=torch.int32).pin_memory(), "plan": None}
            self.slots[i] = slot
        bufs = slot["bufs"]
        slot["plan"] = plan
        if (model is not None and getattr(plan, 'dtype', 0) == _capi.DTYPE_F32
                and hasattr(model, 'device_status_async')):
            model.device_status_async(plan, slot["err"])       # compute stream: behind the forward, in front of `maps`
        slot["maps"].record(torch.cuda.current_stream())
        self.stream.wait_event(slot["maps"])
        with torch.cuda.stream(self.stream):
            dec.decode_enqueue(hbase, lheat, pbase, lpaf, n, h, w, bufs)
            slot["dec_done"].record(self.stream)        # the decoder's last read of the maps
            block = bufs.result.view(n, bufs.words)
            if post is not None:
                block = post(block)
            if slot["host"] is None or slot["host"].shape != block.shape:
                slot["host"] = torch.empty(block.shape, dtype=block.dtype).pin_memory()
            slot["host"].copy_(block, non_blocking=True)
            slot["done"].record(self.stream)
        check(lib.rtpose_net_set_output_guard(plan.handle, slot["dec_done"].cuda_event))
        self._plans[id(plan)] = plan
        bufs.map_hw = (h, w)
        self.last = slot
        return bufs

    def wait(self, i):
        """-> (buffers, numpy int32 view of slot i's pinned record block; valid until the slot is used again).  Raises when
        the plan's device error word came back non-zero (the maps of that batch are invalid)."""
        slot = self.slots[i]
        slot["done"].synchronize()
        word = int(slot["err"][0])
        if word:
            slot["err"][0] = 0
            if word & 1 and slot["plan"] is not None:
                # every later forward of the plan runs one block per tile (same bits, a few per cent slower); captured
                # launch lists are dropped with it
                check(lib.rtpose_net_set_persistent7(slot["plan"].handle, 0))
            raise _capi.RtposeError("device error word %d: a split-tile hand-over of a persistent 7x7 launch timed out "
                                    "(shared / CU-masked device?); the maps of this batch are invalid - the plan has "
                                    "stopped splitting tiles, run the batch again" % word)
        return slot["bufs"], slot["host"].numpy()


class PoseEstimator(object):
    def __init__(self, model, config=None, max_peaks_per_part=32, max_humans=64):
        self.model = model
        self.config = config or dec.default_config()
        self.max_peaks_per_part = max_peaks_per_part
        self.max_humans = max_humans
        self._bufs = {}

    def _buffers(self, n, device):
        key = (n, device.index, self.max_peaks_per_part, self.max_humans)
        b = self._bufs.get(key)
        if b is None:
            cfg = dec.make_cfg(self.config, self.max_peaks_per_part, self.max_humans)
            b = dec.DecodeBuffers(cfg, n, device)
            self._bufs = {key: b}
        return b

    def enqueue(self, x, scene=None, scene_alpha=1e-3):
        """Enqueue forward + decode for a device batch x [N,3,H,W]; returns the buffers.

        scene = (heat [N,h,w,19], paf [N,h,w,38]) device tensors: if given, the maps the
        decoder sees are  scene + scene_alpha * net_output  (bench / tests only: there
        are no trained weights offline, so realistic peaks are superimposed on what
        the randomly initialised network wrote; see include/rtpose_mi355x.h)."""
        import torch
        m = self.model
        with torch.cuda.device(x.device):       # everything below goes to x's device and its current stream
            plan = m.forward_native(x, keep_intermediates=False)
            n = x.shape[0]
            pbase, lpaf, _, h, w = m.output_view(plan, 0)
            hbase, lheat, _, _, _ = m.output_view(plan, 1)
            if scene is not None:
                sh, sp = scene
                s = current_stream()
                check(lib.rtpose_layout_axpby(hbase, C.byref(lheat), ptr(sh), 19, n, h, w, scene_alpha, 1.0, s))
                check(lib.rtpose_layout_axpby(pbase, C.byref(lpaf), ptr(sp), 38, n, h, w, scene_alpha, 1.0, s))
            bufs = self._buffers(n, x.device)
            dec.decode_enqueue(hbase, lheat, pbase, lpaf, n, h, w, bufs)
        bufs.map_hw = (h, w)
        bufs.plan = plan
        return bufs

    # ---- decode(k) under forward(k + 1): see SideDecoder ------------------------------------------------------------
    def submit(self, x, scene=None, scene_alpha=1e-3, post=None):
        """Enqueue forward (+ blend) on the current stream and decode + record D2H on the side stream; returns a ticket
        for collect().  At most two tickets may be outstanding (two record blocks ping-pong).  ``post(result_block)``,
        if given, runs on the side stream after the decode and returns the device tensor whose copy collect() hands
        out (bench.py: the RCCL gather of the ranks' blocks)."""
        import torch
        m = self.model
        with torch.cuda.device(x.device):
            if not hasattr(self, '_side'):
                self._side = SideDecoder(self.config)
                self._ticket = 0
            k = self._ticket
            self._ticket += 1
            plan = self._side.guarded(m.plan_for(x), lambda: m.forward_native(x, keep_intermediates=False))
            n = x.shape[0]
            pbase, lpaf, _, h, w = m.output_view(plan, 0)
            hbase, lheat, _, _, _ = m.output_view(plan, 1)
            if scene is not None:
                sh, sp = scene
                s = current_stream()
                check(lib.rtpose_layout_axpby(hbase, C.byref(lheat), ptr(sh), 19, n, h, w, scene_alpha, 1.0, s))
                check(lib.rtpose_layout_axpby(pbase, C.byref(lpaf), ptr(sp), 38, n, h, w, scene_alpha, 1.0, s))
            bufs = self._side.decode(k & 1, (hbase, lheat, pbase, lpaf, h, w), n, x.device, self.max_peaks_per_part,
                                     self.max_humans, post, plan=plan, model=m)
            bufs.plan = plan
            return k

    def collect(self, ticket):
        """Wait for the records of submit()'s ticket -> (buffers, numpy int32 view of the pinned record block; valid
        until the ticket after next is submitted).  Raises RtposeError when the plan's device error word of that batch is
        non-zero (a persistent 7x7 hand-over timed out: its maps are invalid)."""
        return self._side.wait(ticket & 1)

    def __call__(self, x, scene=None, scene_alpha=1e-3):
        """-> list of per-image dicts (decode.parse_image); grows table capacity on overflow."""
        while True:
            bufs = self.enqueue(x, scene, scene_alpha)
            recs = dec.fetch(bufs)
            _raise_on_device_error(self.model, bufs.plan)
            flags = int(np.bitwise_or.reduce(recs[:, dec.RES_HEADER + 2]))
            if flags & dec.OVERFLOW_PEAKS and self.max_peaks_per_part < dec.MAX_PEAKS_LIMIT:
                self.max_peaks_per_part = min(2 * self.max_peaks_per_part, dec.MAX_PEAKS_LIMIT)
                continue
            if flags & dec.OVERFLOW_HUMANS and self.max_humans < dec.MAX_HUMANS_LIMIT:
                self.max_humans = min(2 * self.max_humans, dec.MAX_HUMANS_LIMIT)
                continue
            if flags:
                raise _capi.RtposeError("decode tables overflowed at maximum capacity (flags=%d)" % flags)
            return [dec.parse_image(recs[i], bufs.cfg) for i in range(x.shape[0])]

    def humans(self, x, **kw):
        """-> list (per image) of list[Human], coordinates normalised like paf_to_pose_cpp."""
        out = []
        up = int(self.config.MODEL.DOWNSAMPLE)
        recs = self(x, **kw)
        h, w = x.shape[2] // 8, x.shape[3] // 8
        for r in recs:
            out.append(dec.humans_from_record(r, w * up, h * up, int(self.config.MODEL.NUM_KEYPOINTS)))
        return out


class StreamingPoseEstimator(object):
    """Host images in, host records out, the host's share of a batch hidden under the previous batch's kernels.

    The reference moves one fp32 image per call (`get_outputs`, evaluate/coco_eval.py:96-110: H2D of
    1.6 MB, D2H of 0.48 MB of maps).  Here a batch of uint8 BGR images (3 B/pixel) is copied into pinned memory
    and its H2D is queued while the previous batch is still in the network; the GPU then does
    resize + pad + normalise (rtpose_preprocess_u8) straight into the plan's input buffer, the
    forward and the decode; only the fixed-size result records come back.  Two pinned host / device
    staging pairs ping-pong.  The H2D rides on the COMPUTE stream (round 4): the 13 MB of a 32-image batch take 0.24 ms
    there - 1 % of an fp32 batch - while the same copy on a second stream took 8.9 ms (tools/exp/host_copy_probe.py,
    profiles/r04_streaming_probe.txt).  Round 5: the decoder and the D2H of the records run on a second stream
    (SideDecoder) and batch k + 1 is enqueued BEFORE the host waits for batch k's records, so the compute stream goes
    from one forward straight into the next.
    """

    def __init__(self, model, batch, h0, w0, preprocess='rtpose', config=None, max_peaks_per_part=32,
                 max_humans=64, scene=None, scene_alpha=1e-3):
        """scene = (heat [B,h,w,19], paf [B,h,w,38]) device tensors, bench / tests only: the decoder then sees
        scene + scene_alpha * maps, bench.py's decoder-input definition for a network without trained weights."""
        import torch
        from . import preprocess as pre
        self.model = model
        self.config = config or dec.default_config()
        self.B, self.h0, self.w0 = batch, h0, w0
        self.mode = {'rtpose': 0, 'vgg': 1}[preprocess]
        size, factor = int(self.config.DATASET.IMAGE_SIZE), int(self.config.MODEL.DOWNSAMPLE)
        self.im_scale = float(size) / min(h0, w0)
        self.hr, self.wr = pre._cv_round(h0 * self.im_scale), pre._cv_round(w0 * self.im_scale)
        self.hn, self.wn = pre._factor_closest(self.hr, factor), pre._factor_closest(self.wr, factor)
        self.dev = torch.device('cuda', torch.cuda.current_device())
        self.host = [torch.empty((batch, h0, w0, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.devbuf = [torch.empty((batch, h0, w0, 3), dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self.max_peaks_per_part, self.max_humans = max_peaks_per_part, max_humans
        self.scene, self.scene_alpha = scene, scene_alpha
        self.side = SideDecoder(self.config)
        self.bufs = None                    # the decode buffers of the batch enqueued last (tests read .cfg)
        self._torch = torch
        self._pre = pre

    def _upload(self, slot, images):
        torch = self._torch
        images = np.asarray(images)
        if images.shape != (self.B, self.h0, self.w0, 3) or images.dtype != np.uint8:
            raise _capi.RtposeError("StreamingPoseEstimator: every batch must be uint8 [%d, %d, %d, 3], got %s %s "
                                    "(pad a short final batch with copies and drop their records)"
                                    % (self.B, self.h0, self.w0, images.dtype, images.shape))
        # host -> pinned (the caller's array may be pageable), then the H2D queued on the compute stream: behind the
        # kernels of the batch in flight, in front of this batch's own.  Slot `slot` was last read two batches ago.
        # (a plain memcpy, NOT torch's copy_: ATen splits a 13 MB CPU copy over every core, and the OpenMP workers that
        #  then spin-wait starve the HIP runtime's own threads - every third batch stalled for 30-80 ms, measured with
        #  tools/exp/stall_probe.py; np.copyto takes 0.2 ms on one core)
        np.copyto(self.host[slot].numpy(), images)
        self.devbuf[slot].copy_(self.host[slot], non_blocking=True)

    def _enqueue(self, slot):
        """Everything of the batch whose images sit in devbuf[slot], asynchronously: image prep and forward on the
        compute stream (the forward guarded against the decoder still reading the previous maps), (scene blend,) the
        device error word, then decode + D2H of the records on the side stream."""
        m = self.model
        plan = m.plan_for_shape(self.B, self.hn, self.wn, self.dev)
        s = current_stream()
        img_bytes = self.h0 * self.w0 * 3
        base = self.devbuf[slot].data_ptr()
        self._pre.preprocess_into_plan(plan, [base + b * img_bytes for b in range(self.B)],
                                       [(self.h0, self.w0)] * self.B, int(self.config.DATASET.IMAGE_SIZE),
                                       self.mode, s)                      # ONE launch for the whole batch
        check(lib.rtpose_net_set_keep_intermediates(plan.handle, 0))
        self.side.guarded(plan, lambda: check(lib.rtpose_net_forward_prepared(plan.handle, s),
                                              "rtpose_net_forward_prepared"))
        pbase, lpaf, _, h, w = m.output_view(plan, 0)
        hbase, lheat, _, _, _ = m.output_view(plan, 1)
        if self.scene is not None:      # bench / tests only, see PoseEstimator.enqueue
            sh, sp = self.scene
            check(lib.rtpose_layout_axpby(hbase, C.byref(lheat), ptr(sh), 19, self.B, h, w, self.scene_alpha, 1.0, s))
            check(lib.rtpose_layout_axpby(pbase, C.byref(lpaf), ptr(sp), 38, self.B, h, w, self.scene_alpha, 1.0, s))
        # (the plan's device error word rides in front of the decode and is checked by side.wait(): the host never calls a
        #  stream-synchronising API while the next batch is queued - round-4 advisor finding)
        self.bufs = self.side.decode(slot, (hbase, lheat, pbase, lpaf, h, w), self.B, self.dev,
                                     self.max_peaks_per_part, self.max_humans, plan=plan, model=m)
        return slot

    def _finish(self, slot):
        """Wait for the batch enqueued into `slot` and return its records (a copy: the pinned block is reused)."""
        while True:
            bufs, host = self.side.wait(slot)    # the records of THIS batch (the next batch may already be running)
            recs = host.reshape(self.B, bufs.words).copy()
            flags = int(np.bitwise_or.reduce(recs[:, dec.RES_HEADER + 2]))
            if not flags:
                return recs
            # a crowded image overflowed a device table: grow it and run the SAME batch again - its images are still in
            # devbuf[slot] (the next upload into this slot comes after this call), its maps are not (the next batch's
            # forward has been enqueued behind it) - records are never handed out truncated.  Capacities only grow, so
            # this happens a handful of times in the life of a stream.
            if flags & dec.OVERFLOW_PEAKS and self.max_peaks_per_part < dec.MAX_PEAKS_LIMIT:
                self.max_peaks_per_part = min(2 * self.max_peaks_per_part, dec.MAX_PEAKS_LIMIT)
            elif flags & dec.OVERFLOW_HUMANS and self.max_humans < dec.MAX_HUMANS_LIMIT:
                self.max_humans = min(2 * self.max_humans, dec.MAX_HUMANS_LIMIT)
            else:
                raise _capi.RtposeError("decode tables overflowed at maximum capacity (flags=%d)" % flags)
            self._enqueue(slot)

    def run(self, batches):
        """batches: iterable of uint8 arrays [B, h0, w0, 3] (BGR).  Yields one int32 record block
        [B, words] per batch, in order (decode.parse_image(rec) / humans_from_record turn them into Humans).  The decode
        tables grow on overflow, so blocks of one run may differ in width; every record carries the capacities
        it was written with in its header (words 3, 4) and parse_image reads them from there - a consumer that
        collects blocks first, or parses a step late, never needs this object's cfg of the moment.

        Order (round 5): batch k's kernels are enqueued; batch k + 1 is taken from the iterator, copied from the
        caller's (pageable) array into pinned memory, its H2D and its kernels queued behind k's; only then does the host
        wait for k's records (an event behind their D2H on the side stream) and yield them.  The compute stream never
        waits for the host, and the decoder of k runs beside the forward of k + 1; the price is that the iterator is
        read one batch ahead."""
        it = iter(batches)
        try:
            cur = next(it)
        except StopIteration:
            return
        slot = 0
        self._upload(slot, cur)
        self._enqueue(slot)
        while True:
            try:
                nxt = next(it)
            except StopIteration:
                nxt = None
            if nxt is not None:
                self._upload(slot ^ 1, nxt)     # host copy + H2D + kernels of the next batch behind this batch's
                self._enqueue(slot ^ 1)
            yield self._finish(slot)
            if nxt is None:
                return
            slot ^= 1
