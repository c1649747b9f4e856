"""Host side of the batched GPU pose decoder (csrc/decode.hip) and the drop-in
``paf_to_pose_cpp`` (reference lib/utils/paf_to_pose.py:372-406).

Everything numeric happens in librtpose_mi355x.so; this module only moves
buffers (torch for device memory) and turns the fixed-capacity result records
into ``Human`` / ``BodyPart`` objects.  If a device table overflows the call is
repeated with doubled capacity — nothing is silently truncated.
"""
import ctypes as C
import types

import numpy as np
import torch

from . import _capi
from ._capi import lib, check, ptr, current_stream, Layout, DecodeCfg, NUM_PART
from .common import Human, BodyPart

RES_HEADER, RES_PART_COUNT, RES_PEAKS = 0, 8, 32
OVERFLOW_PEAKS, OVERFLOW_HUMANS = 1, 2
MAX_PEAKS_LIMIT = 1024
MAX_HUMANS_LIMIT = 16384


def default_config():
    """The four cfg keys the decoder reads (lib/config/default.py:40,41,69,126)."""
    return types.SimpleNamespace(
        MODEL=types.SimpleNamespace(NUM_KEYPOINTS=18, DOWNSAMPLE=8),
        DATASET=types.SimpleNamespace(IMAGE_SIZE=368),
        TEST=types.SimpleNamespace(THRESH_HEATMAP=0.1))


def make_cfg(config=None, max_peaks_per_part=32, max_humans=64):
    config = config or default_config()
    return DecodeCfg(int(config.MODEL.NUM_KEYPOINTS), int(config.MODEL.DOWNSAMPLE),
                     float(config.TEST.THRESH_HEATMAP), int(max_peaks_per_part), int(max_humans))


class DecodeBuffers(object):
    """Device scratch + result block for N images at a given capacity."""

    def __init__(self, cfg, n, device):
        self.cfg = cfg
        self.n = n
        ws = lib.rtpose_decode_workspace_bytes(C.byref(cfg), n)
        rb = lib.rtpose_decode_result_bytes(C.byref(cfg), n)
        if ws == 0 or rb == 0:
            raise _capi.RtposeError("bad decode config: " + _capi.last_error())
        self.workspace = torch.empty(ws // 4, dtype=torch.int32, device=device)
        self.result = torch.empty(rb // 4, dtype=torch.int32, device=device)
        self.words = rb // 4 // n
        self.host = torch.empty(rb // 4, dtype=torch.int32).pin_memory() if device.type == 'cuda' else None


def decode_enqueue(heat_ptr, lheat, paf_ptr, lpaf, n, h, w, bufs, nms_only=False, nms_flags=0):
    """Enqueue the decode kernels on the current stream (device pointers + layouts).
    nms_flags: _capi.NMS_NO_REFINE | _capi.NMS_GAUSSIAN (the optional branches of NMS, paf_to_pose.py:67)."""
    cfg = bufs.cfg
    if nms_only:
        check(lib.rtpose_nms_batch_ex(heat_ptr, C.byref(lheat), n, h, w, C.byref(cfg), nms_flags, ptr(bufs.result),
                                      current_stream()), "rtpose_nms_batch_ex")
    else:
        check(lib.rtpose_decode_batch_ex(heat_ptr, C.byref(lheat), paf_ptr, C.byref(lpaf), n, h, w, C.byref(cfg),
                                         nms_flags, ptr(bufs.workspace), bufs.workspace.numel() * 4,
                                         ptr(bufs.result), current_stream()), "rtpose_decode_batch_ex")


def fetch(bufs):
    """D2H of the result block -> numpy int32 [N, words] (on the device the buffers live on)."""
    with torch.cuda.device(bufs.result.device):
        bufs.host.copy_(bufs.result, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    return bufs.host.numpy().reshape(bufs.n, bufs.words)


def parse_image(rec, cfg=None):
    """One image's record -> dict(peaks=[P,5] float32 joint_list, parts=[H,18], score=[H], flags).
    The capacities the record was written with are read from its own header (words 3, 4); ``cfg`` is only
    consulted for records without them."""
    pcap, hcap = int(rec[RES_HEADER + 3]), int(rec[RES_HEADER + 4])
    if pcap <= 0 or hcap <= 0:
        if cfg is None:
            raise _capi.RtposeError("record carries no capacities and no cfg was given")
        pcap, hcap = cfg.max_peaks_per_part, cfg.max_humans
    counts = rec[RES_PART_COUNT:RES_PART_COUNT + NUM_PART]
    pk = rec[RES_PEAKS:RES_PEAKS + 4 * NUM_PART * pcap].reshape(NUM_PART, pcap, 4)
    rows = []
    for p in range(NUM_PART):
        c = int(counts[p])
        if c:
            blk = pk[p, :c]
            rows.append(np.stack([blk[:, 0].astype(np.float32), blk[:, 1].astype(np.float32),
                                  blk[:, 2].copy().view(np.float32), blk[:, 3].astype(np.float32),
                                  np.full(c, p, np.float32)], axis=1))
    peaks = np.concatenate(rows, 0) if rows else np.zeros((0, 5), np.float32)
    nh = int(rec[RES_HEADER + 1])
    off = RES_PEAKS + 4 * NUM_PART * pcap
    parts = rec[off:off + NUM_PART * hcap].reshape(hcap, NUM_PART)[:nh].copy()
    score = rec[off + NUM_PART * hcap:off + NUM_PART * hcap + hcap].copy().view(np.float32)[:nh].copy()
    return {"peaks": peaks, "parts": parts, "score": score, "flags": int(rec[RES_HEADER + 2]),
            "n_peaks": int(rec[RES_HEADER])}


def result_mask(block):
    """Boolean mask [N, words] of the words of a record block that carry results - header, part counts, the peaks of every
    part up to its count, the part assignments and scores of the humans up to the human count.  Words behind the counts are
    scratch (left-overs of earlier batches): two blocks say the same iff they agree on the mask of either
    (``np.array_equal(a[m], b[m])`` with ``m = result_mask(b)``; a differing count differs inside the mask)."""
    block = np.asarray(block)
    pcap, hcap = int(block[0, RES_HEADER + 3]), int(block[0, RES_HEADER + 4])
    mask = np.zeros(block.shape, dtype=bool)
    mask[:, RES_HEADER:RES_HEADER + 5] = True
    mask[:, RES_PART_COUNT:RES_PART_COUNT + NUM_PART] = True
    hoff = RES_PEAKS + 4 * NUM_PART * pcap
    for b in range(block.shape[0]):
        for p in range(NUM_PART):
            c = min(int(block[b, RES_PART_COUNT + p]), pcap)
            o = RES_PEAKS + 4 * p * pcap
            mask[b, o:o + 4 * c] = True
        nh = min(int(block[b, RES_HEADER + 1]), hcap)
        mask[b, hoff:hoff + NUM_PART * nh] = True
        mask[b, hoff + NUM_PART * hcap:hoff + NUM_PART * hcap + nh] = True
    return mask


def decode_maps(heat, paf, config=None, max_peaks_per_part=32, max_humans=64, nms_only=False, nms_flags=0):
    """heat [N,h,w,C>=num_keypoints], paf [N,h,w,38]: dense NHWC float32 CUDA tensors.
    Returns a list of per-image dicts (see parse_image).  Grows capacities on overflow."""
    if not heat.is_cuda:
        raise _capi.RtposeError("decode runs on the GPU only (no CPU fallback); got a CPU tensor")
    n, h, w, ch = heat.shape
    heat = heat.contiguous().float()
    paf = paf.contiguous().float()
    lheat = Layout.dense(ch, h, w)
    lpaf = Layout.dense(paf.shape[3], h, w)
    while True:
        cfg = make_cfg(config, max_peaks_per_part, max_humans)
        with torch.cuda.device(heat.device):      # kernels + stream of the maps' device, whatever is current
            bufs = DecodeBuffers(cfg, n, heat.device)
            decode_enqueue(ptr(heat), lheat, ptr(paf), lpaf, n, h, w, bufs, nms_only=nms_only, nms_flags=nms_flags)
            recs = fetch(bufs)
        flags = int(np.bitwise_or.reduce(recs[:, RES_HEADER + 2]))
        if flags & OVERFLOW_PEAKS:
            if max_peaks_per_part >= MAX_PEAKS_LIMIT:
                raise _capi.RtposeError("more than %d peaks of one part in an image" % MAX_PEAKS_LIMIT)
            max_peaks_per_part = min(2 * max_peaks_per_part, MAX_PEAKS_LIMIT)
            continue
        if flags & OVERFLOW_HUMANS:
            if max_humans >= MAX_HUMANS_LIMIT:
                raise _capi.RtposeError("more than %d person candidates in an image" % MAX_HUMANS_LIMIT)
            max_humans = min(2 * max_humans, MAX_HUMANS_LIMIT)
            continue
        return [parse_image(recs[i], cfg) for i in range(n)]


def humans_from_record(rec, up_w, up_h, num_keypoints=18):
    """paf_to_pose.py:387-404: result dict -> list[Human] with normalised coordinates."""
    humans = []
    peaks = rec["peaks"]
    for hid in range(rec["parts"].shape[0]):
        human = Human([])
        added = False
        for part_idx in range(num_keypoints):
            cid = int(rec["parts"][hid, part_idx])
            if cid < 0:
                continue
            added = True
            human.body_parts[part_idx] = BodyPart('%d-%d' % (hid, part_idx), part_idx,
                                                  float(int(peaks[cid, 0])) / up_w,
                                                  float(int(peaks[cid, 1])) / up_h, float(peaks[cid, 2]))
        if added:
            human.score = float(rec["score"][hid])
            humans.append(human)
    return humans


def NMS(heatmaps, upsampFactor=1., bool_refine_center=True, bool_gaussian_filt=False, config=None):
    """Drop-in for lib/utils/paf_to_pose.py:67 — list (per joint type) of [K,4] float64 arrays
    (x, y, score, id), all flag settings: bool_gaussian_filt smooths the up-sampled patch with
    scipy's gaussian_filter(sigma=3) arithmetic on the GPU (:121-122), bool_refine_center=False
    returns the grid-snapped cell centres (:135-139)."""
    config = config or default_config()
    up = int(upsampFactor)
    if up != upsampFactor or up < 1:
        raise ValueError("upsampFactor must be a positive integer (cfg.MODEL.DOWNSAMPLE)")
    cfgc = types.SimpleNamespace(MODEL=types.SimpleNamespace(NUM_KEYPOINTS=config.MODEL.NUM_KEYPOINTS, DOWNSAMPLE=up),
                                 TEST=config.TEST)
    dev = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
    if dev is None:
        raise _capi.RtposeError("NMS needs an MI355X (HIP) device; there is no CPU fallback")
    flags = 0
    if not bool_refine_center:
        flags |= _capi.NMS_NO_REFINE          # (the Gaussian lives inside the refine branch, :106-122)
    elif bool_gaussian_filt:
        flags |= _capi.NMS_GAUSSIAN
    heat = torch.as_tensor(np.ascontiguousarray(heatmaps, dtype=np.float32)).to(dev)[None]
    rec = decode_maps(heat, heat.new_zeros(1, heat.shape[1], heat.shape[2], 38), cfgc, nms_only=True,
                      nms_flags=flags)[0]
    pk = rec["peaks"].astype(np.float64)
    if not bool_refine_center:
        # the device keeps the centre truncated to int (what process_paf reads); the reference's
        # list carries compute_resized_coords' float64 value (c + 0.5) * up - 0.5  (:41-64)
        pk[:, 0:2] = (np.floor(pk[:, 0:2] / up) + 0.5) * up - 0.5
    return [pk[pk[:, 4] == j][:, :4] for j in range(int(config.MODEL.NUM_KEYPOINTS))]


def paf_to_pose_cpp(heatmaps, pafs, config):
    """Drop-in for lib/utils/paf_to_pose.py:372 — HWC float32 numpy (or torch) maps in,
    list[Human] out.  NMS, PAF scoring, assignment and grouping all run on the GPU."""
    dev = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
    if dev is None:
        raise _capi.RtposeError("paf_to_pose_cpp needs an MI355X (HIP) device; there is no CPU fallback")
    heat = torch.as_tensor(np.ascontiguousarray(heatmaps, dtype=np.float32) if not torch.is_tensor(heatmaps)
                           else heatmaps).to(dev)[None]
    paf = torch.as_tensor(np.ascontiguousarray(pafs, dtype=np.float32) if not torch.is_tensor(pafs)
                          else pafs).to(dev)[None]
    rec = decode_maps(heat, paf, config)[0]
    up = int(config.MODEL.DOWNSAMPLE)
    h, w = heat.shape[1], heat.shape[2]
    return humans_from_record(rec, w * up, h * up, int(config.MODEL.NUM_KEYPOINTS))
