"""Caller side of the boundary: evaluate/coco_eval.py:get_outputs (:80-114),
handle_paf_and_heat (:197-242), append_result (:117-154) and the CPU image prep it
calls (lib/network/im_transform.py:113-134, lib/datasets/preprocessing.py:16-43).

The reference does this prep with cv2 on the host; cv2 is not in this image, so
``cv2.resize(im, None, fx=s, fy=s)`` (INTER_LINEAR on uint8) is restated from
OpenCV's published fixed-point algorithm (imgproc/resize.cpp: 11-bit coefficients,
HResizeLinear -> VResizeLinear with the (x>>4, >>16, +2, >>2) rounding).  Parity of
that restatement against a real OpenCV build is UNPINNED here; when cv2 is importable
it is used instead.  The network + decoder that follow run on the GPU.
"""
import numpy as np
import torch

from . import _capi, decode as dec
from ._capi import lib, check, ptr, current_stream


def C_byref(x):
    import ctypes
    return ctypes.byref(x)

try:  # pragma: no cover - not in this image
    import cv2 as _cv2
except Exception:  # noqa: BLE001
    _cv2 = None

ORDER_COCO = [0, 15, 14, 17, 16, 5, 2, 6, 3, 7, 4, 11, 8, 12, 9, 13, 10]   # coco_eval.py:52


def _cv_round(x):
    return int(np.rint(x))          # cvRound: round half to even


def resize_linear_u8(im, fx, fy):
    """cv2.resize(im, None, fx=fx, fy=fy) for uint8 HxWxC (INTER_LINEAR)."""
    if _cv2 is not None:
        return _cv2.resize(im, None, fx=fx, fy=fy)
    h, w = im.shape[:2]
    dw, dh = _cv_round(w * fx), _cv_round(h * fy)
    sx_scale, sy_scale = 1.0 / fx, 1.0 / fy

    def coeffs(dn, sn, scale, rows):
        d = np.arange(dn, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if rows:   # resize.cpp's dy loop keeps the weight; resizeGeneric_Invoker clips the two source rows
            s0, s1 = np.clip(s, 0, sn - 1), np.clip(s + 1, 0, sn - 1)
        else:      # the dx loop clamps offset AND weight (f = 0) for taps outside the image
            lo = s < 0
            f[lo], s[lo] = 0.0, 0
            hi = s >= sn - 1
            f[hi], s[hi] = 0.0, sn - 1
            s0, s1 = s, np.minimum(s + 1, sn - 1)
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        a0 = np.rint((np.float32(1.0) - f) * np.float32(2048)).astype(np.int64)
        return s0, s1, a0, a1

    x0, x1, ax0, ax1 = coeffs(dw, w, sx_scale, False)
    y0, y1, ay0, ay1 = coeffs(dh, h, sy_scale, True)
    src = im.astype(np.int64)
    if src.ndim == 2:
        src = src[:, :, None]
    hor = src[:, x0, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]      # int, x2048
    s0, s1 = hor[y0], hor[y1]
    out = ((((ay0[:, None, None] * (s0 >> 4)) >> 16) + ((ay1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2)
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out if im.ndim == 3 else out[:, :, 0]


def _factor_closest(num, factor, is_ceil=True):
    num = np.ceil(float(num) / factor) if is_ceil else np.floor(float(num) / factor)
    return int(num) * factor


def crop_with_factor(im, dest_size=None, factor=32, is_ceil=True):
    """im_transform.py:119-134: short side -> dest_size, zero-pad right/bottom to a multiple of factor."""
    im_scale = float(dest_size) / np.min(im.shape[0:2])
    im = resize_linear_u8(im, im_scale, im_scale)
    h, w, c = im.shape
    new_h, new_w = _factor_closest(h, factor, is_ceil), _factor_closest(w, factor, is_ceil)
    im_croped = np.zeros([new_h, new_w, c], dtype=im.dtype)
    im_croped[0:h, 0:w, :] = im
    return im_croped, im_scale, im.shape


def rtpose_preprocess(image):
    """preprocessing.py:16-21"""
    image = image.astype(np.float32)
    image = image / 256. - 0.5
    return image.transpose((2, 0, 1)).astype(np.float32)


def vgg_preprocess(image):
    """preprocessing.py:32-43 (BGR -> RGB, /255, ImageNet mean/std)"""
    image = image.astype(np.float32) / 255.
    means, stds = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    out = image.copy()[:, :, ::-1]
    for i in range(3):
        out[:, :, i] = out[:, :, i] - means[i]
        out[:, :, i] = out[:, :, i] / stds[i]
    return out.transpose((2, 0, 1)).astype(np.float32)


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def get_outputs(img, model, preprocess, config=None):
    """coco_eval.py:80-114: BGR uint8 image -> (paf [h,w,38], heatmap [h,w,19], im_scale), HWC float32."""
    config = config or dec.default_config()
    im_croped, im_scale, _ = crop_with_factor(img, int(config.DATASET.IMAGE_SIZE),
                                              factor=int(config.MODEL.DOWNSAMPLE), is_ceil=True)
    if preprocess == 'rtpose':
        im_data = rtpose_preprocess(im_croped)
    elif preprocess == 'vgg':
        im_data = vgg_preprocess(im_croped)
    else:
        raise ValueError("preprocess must be 'rtpose' or 'vgg'")
    batch = torch.from_numpy(np.expand_dims(im_data, 0)).cuda().float()
    predicted_outputs, _ = model(batch)
    output1, output2 = predicted_outputs[-2], predicted_outputs[-1]
    heatmap = output2.cpu().data.numpy().transpose(0, 2, 3, 1)[0]
    paf = output1.cpu().data.numpy().transpose(0, 2, 3, 1)[0]
    return paf, heatmap, im_scale


def prep_geometry(h0, w0, dest_size, factor):
    """crop_with_factor's sizes without touching pixels: (im_scale, (hr, wr) resized, (hn, wn) padded)."""
    im_scale = float(dest_size) / min(h0, w0)
    hr, wr = _cv_round(h0 * im_scale), _cv_round(w0 * im_scale)
    return im_scale, (hr, wr), (_factor_closest(hr, factor), _factor_closest(wr, factor))


def preprocess_into_plan(plan, images_dev, sizes, dest_size, mode, stream, flips=None, slots=None):
    """ONE rtpose_preprocess_u8_batch launch: device uint8 BGR images (tensors, or raw device
    addresses) of sizes [(h0, w0), ...] -> resized / padded / normalised into the plan's input
    buffer, image k into slot slots[k] (default k), mirrored inside its valid width if flips[k]."""
    import ctypes as C
    ibase, ilay = C.c_void_p(), _capi.Layout()
    check(lib.rtpose_net_input_view(plan.handle, C.byref(ibase), C.byref(ilay)), "rtpose_net_input_view")
    n, hn, wn = plan.shape
    descs = (_capi.PrepImage * len(sizes))()
    for k, (h0, w0) in enumerate(sizes):
        im_scale = float(dest_size) / min(h0, w0)
        d = descs[k]
        d.img_bgr = images_dev[k].data_ptr() if hasattr(images_dev[k], "data_ptr") else int(images_dev[k])
        d.im_scale, d.h0, d.w0 = im_scale, h0, w0
        d.hr, d.wr = _cv_round(h0 * im_scale), _cv_round(w0 * im_scale)
        d.flip = int(bool(flips[k])) if flips is not None else 0
        d.n_index = slots[k] if slots is not None else k
        if d.n_index >= n:
            raise _capi.RtposeError("image slot %d outside the %d-image plan" % (d.n_index, n))
    check(lib.rtpose_preprocess_u8_batch(descs, len(sizes), mode, ibase, C.byref(ilay), hn, wn, stream),
          "rtpose_preprocess_u8_batch")


def get_outputs_gpu(img, model, preprocess, config=None):
    """get_outputs with the image prep on the GPU too: the uint8 image is
    uploaded as is (3 B/pixel instead of 12) and ONE kernel does resize + pad + normalise +
    NHWC packing straight into the network's input buffer.  Same return values as get_outputs;
    the resized pixels are bit-identical to resize_linear_u8."""
    import ctypes as C
    config = config or dec.default_config()
    size, factor = int(config.DATASET.IMAGE_SIZE), int(config.MODEL.DOWNSAMPLE)
    h0, w0 = img.shape[:2]
    im_scale = float(size) / min(h0, w0)
    hr, wr = _cv_round(h0 * im_scale), _cv_round(w0 * im_scale)
    hn, wn = _factor_closest(hr, factor), _factor_closest(wr, factor)
    m = _unwrap(model)
    dev = torch.device('cuda', torch.cuda.current_device())
    img_d = torch.from_numpy(np.ascontiguousarray(img, dtype=np.uint8)).to(dev)
    plan = m.plan_for(torch.empty((1, 3, hn, wn), device=dev))
    base, lay = C.c_void_p(), _capi.Layout()
    check(lib.rtpose_net_input_view(plan.handle, C.byref(base), C.byref(lay)), "rtpose_net_input_view")
    mode = {'rtpose': 0, 'vgg': 1}[preprocess]
    s = current_stream()
    check(lib.rtpose_preprocess_u8(ptr(img_d), h0, w0, im_scale, mode, base, C.byref(lay), 0, hn, wn, hr, wr, s),
          "rtpose_preprocess_u8")
    check(lib.rtpose_net_set_keep_intermediates(plan.handle, 0))
    check(lib.rtpose_net_forward_prepared(plan.handle, s), "rtpose_net_forward_prepared")
    paf = m.read_output(plan, 10).cpu().numpy().transpose(0, 2, 3, 1)[0]
    heatmap = m.read_output(plan, 11).cpu().numpy().transpose(0, 2, 3, 1)[0]
    return paf, heatmap, im_scale


def handle_paf_and_heat(normal_heat, flipped_heat, normal_paf, flipped_paf):
    """coco_eval.py:197-242 on the GPU (csrc/layout_ops.hip:flip_merge_kernel)."""
    dev = torch.device('cuda', torch.cuda.current_device())
    t = [torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev)[None]
         for a in (normal_heat, flipped_heat, normal_paf, flipped_paf)]
    h, w = t[0].shape[1], t[0].shape[2]
    oh, op = torch.empty_like(t[0]), torch.empty_like(t[2])
    check(lib.rtpose_flip_merge(ptr(t[0]), ptr(t[1]), ptr(t[2]), ptr(t[3]), 1, h, w, ptr(oh), ptr(op),
                                current_stream()), "rtpose_flip_merge")
    return op[0].cpu().numpy(), oh[0].cpu().numpy()


def append_result(image_id, humans, upsample_keypoints, outputs, num_keypoints=18):
    """coco_eval.py:117-154: COCO keypoint-results records (score hard-coded to 1.0 there)."""
    for human in humans:
        keypoints = np.zeros((18, 3))
        for i in range(num_keypoints):
            if i in human.body_parts:
                bp = human.body_parts[i]
                keypoints[i] = (bp.x * upsample_keypoints[1] + 0.5, bp.y * upsample_keypoints[0] + 0.5, 1)
        outputs.append({"image_id": image_id, "category_id": 1, "score": 1.,
                        "keypoints": list(keypoints[ORDER_COCO, :].reshape(51))})


def get_multiscale_outputs(img, model, preprocess='rtpose', scales=(0.5, 1.0, 1.5, 2.0), flip=True, config=None):
    """Multi-scale (+ horizontal flip) test-time augmentation — BASELINE config 3 / README.md:26.

    The surveyed reference commit only keeps remnants of this path (handle_paf_and_heat is
    imported but never called; `get_outputs` is single-scale), so the scale set and the merge
    resolution are an ASSUMPTION following the upstream OpenPose convention: for each scale s
    the image is resized so that its short side is s * IMAGE_SIZE, padded to a multiple of 8 and
    run through the net; every scale's PAF/heat-map (valid, un-padded region only) is bilinearly
    resized to the scale-1.0 map size and averaged; the flipped pass is merged per scale with the
    reference's handle_paf_and_heat semantics.  All arithmetic after the image prep is on the GPU.
    Returns (paf [h,w,38], heatmap [h,w,19], im_scale of the 1.0 pass)."""
    config = config or dec.default_config()
    base = int(config.DATASET.IMAGE_SIZE)
    stride = int(config.MODEL.DOWNSAMPLE)
    dev = torch.device('cuda', torch.cuda.current_device())
    prep = rtpose_preprocess if preprocess == 'rtpose' else vgg_preprocess
    h0, w0 = img.shape[:2]
    s1 = float(base) / min(h0, w0)
    hd, wd = -(-_cv_round(h0 * s1) // stride), -(-_cv_round(w0 * s1) // stride)
    acc_heat = torch.zeros(1, hd, wd, 19, device=dev)
    acc_paf = torch.zeros(1, hd, wd, 38, device=dev)
    stream = current_stream()
    m = _unwrap(model)
    for si, s in enumerate(scales):
        im_croped, im_scale, real_shape = crop_with_factor(img, int(round(base * s)), factor=stride, is_ceil=True)
        x = torch.from_numpy(np.expand_dims(prep(im_croped), 0)).to(dev)
        if flip:
            # the padded columns sit on the right of the normal pass and on the left of the flipped one:
            # mirror only inside the valid width by flipping the un-padded image region instead;
            # both passes of a scale run as ONE batch of 2
            vw = real_shape[1]
            xb = torch.cat([x, x], 0)
            xb[1, :, :, :vw] = torch.flip(x[0, :, :, :vw], dims=[2])
            (paf2, heat2), _ = model(xb)
            heat2 = heat2.permute(0, 2, 3, 1)
            paf2 = paf2.permute(0, 2, 3, 1)
            vwm = -(-vw // stride)
            hv, pv = heat2[0:1, :, :vwm].contiguous(), paf2[0:1, :, :vwm].contiguous()
            hfv, pfv = heat2[1:2, :, :vwm].contiguous(), paf2[1:2, :, :vwm].contiguous()
            mh, mp = torch.empty_like(hv), torch.empty_like(pv)
            check(lib.rtpose_flip_merge(ptr(hv), ptr(hfv), ptr(pv), ptr(pfv), 1, hv.shape[1], vwm, ptr(mh), ptr(mp),
                                        stream), "rtpose_flip_merge")
            heat, paf = mh, mp
        else:
            (paf, heat), _ = model(x)
            heat = heat.permute(0, 2, 3, 1).contiguous()
            paf = paf.permute(0, 2, 3, 1).contiguous()
        hs, ws = heat.shape[1], heat.shape[2]
        # one destination cell (stride px of the scale-1 image) spans im_scale/s1 source cells,
        # whatever the two paddings are: the map origins coincide, only the zoom differs
        ratio = im_scale / s1
        a = 1.0 / len(scales)
        beta = 0.0 if si == 0 else 1.0
        check(lib.rtpose_resize_bilinear_accum(ptr(heat), hs, ws, ptr(acc_heat), hd, wd, 19, 1, hd * ratio, wd * ratio,
                                               a, beta, stream), "rtpose_resize_bilinear_accum")
        check(lib.rtpose_resize_bilinear_accum(ptr(paf), hs, ws, ptr(acc_paf), hd, wd, 38, 1, hd * ratio, wd * ratio,
                                               a, beta, stream), "rtpose_resize_bilinear_accum")
    return acc_paf[0].cpu().numpy(), acc_heat[0].cpu().numpy(), s1


def get_multiscale_outputs_batch(imgs, model, preprocess='rtpose', scales=(0.5, 1.0, 1.5, 2.0), flip=True,
                                 config=None):
    """Batched, GPU-resident form of get_multiscale_outputs (BASELINE config 3): B uint8 BGR images
    of one size are uploaded once (3 B/pixel); per scale ONE kernel per image resizes + pads +
    normalises it (and its mirror image) straight into the input buffer of a 2B-image plan, one
    forward runs all of them, and one fused kernel (rtpose_tta_accumulate) does the flip merge,
    the resize to the scale-1 map and the running average where the net wrote its outputs.
    Same arithmetic as get_multiscale_outputs, image by image.
    Returns DEVICE tensors (paf [B,h,w,38], heat [B,h,w,19]) and the scale-1 im_scale - feed them
    to decode.decode_maps."""
    import ctypes as C
    config = config or dec.default_config()
    base = int(config.DATASET.IMAGE_SIZE)
    stride = int(config.MODEL.DOWNSAMPLE)
    imgs = np.ascontiguousarray(np.stack([np.asarray(i, dtype=np.uint8) for i in imgs]))
    B, h0, w0 = imgs.shape[:3]
    dev = torch.device('cuda', torch.cuda.current_device())
    m = _unwrap(model)
    img_d = torch.from_numpy(imgs).to(dev)
    s1 = float(base) / min(h0, w0)
    hd, wd = -(-_cv_round(h0 * s1) // stride), -(-_cv_round(w0 * s1) // stride)
    acc_heat = torch.empty(B, hd, wd, 19, device=dev)
    acc_paf = torch.empty(B, hd, wd, 38, device=dev)
    stream = current_stream()
    mode = {'rtpose': 0, 'vgg': 1}[preprocess]
    nb = 2 * B if flip else B
    img_bytes = h0 * w0 * 3
    for si, s in enumerate(scales):
        im_scale = float(int(round(base * s))) / min(h0, w0)          # crop_with_factor, im_transform.py:124
        hr, wr = _cv_round(h0 * im_scale), _cv_round(w0 * im_scale)
        hn, wn = _factor_closest(hr, stride), _factor_closest(wr, stride)
        plan = m.plan_for_shape(nb, hn, wn, dev)
        # all B images and their B mirror images in ONE launch
        srcs = [img_d.data_ptr() + b * img_bytes for b in range(B)]
        reps = 2 if flip else 1
        preprocess_into_plan(plan, srcs * reps, [(h0, w0)] * (B * reps), int(round(base * s)), mode, stream,
                             flips=[0] * B + [1] * B if flip else None)
        check(lib.rtpose_net_set_keep_intermediates(plan.handle, 0))
        check(lib.rtpose_net_forward_prepared(plan.handle, stream), "rtpose_net_forward_prepared")
        pbase, lpaf, _, hs, ws = m.output_view(plan, 0)
        hbase, lheat, _, _, _ = m.output_view(plan, 1)
        wv = -(-wr // stride) if flip else ws
        ratio = im_scale / s1
        check(lib.rtpose_tta_accumulate(hbase, C.byref(lheat), pbase, C.byref(lpaf), B, hs, wv, ptr(acc_heat),
                                        ptr(acc_paf), hd, wd, hd * ratio, wd * ratio, 1.0 / len(scales),
                                        0.0 if si == 0 else 1.0, 1 if flip else 0, stream), "rtpose_tta_accumulate")
    return acc_paf, acc_heat, s1


def imread_bgr(path):
    """cv2.imread stand-in for run_eval: a `.npy` side-car holding the BGR uint8 array if the image
    file itself is absent, else OpenCV if present, else PIL.  Returns None if unreadable."""
    import os
    if not os.path.exists(path):
        side = os.path.splitext(path)[0] + ".npy"
        return np.load(side) if os.path.exists(side) else None
    if path.endswith(".npy"):
        return np.load(path)
    try:
        import cv2
        return cv2.imread(path)
    except ImportError:
        pass
    try:
        from PIL import Image
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    except ImportError:
        return None


def run_eval(image_dir, anno_file, vis_dir, model, preprocess, config=None, imread=None, max_images=None):
    """evaluate/coco_eval.py:245-290 (the entry evaluate/evaluation.py calls): loop over the person
    images of a COCO annotation file - get_outputs (GPU image prep + forward), paf_to_pose_cpp,
    draw_humans into vis_dir, append_result - then the OKS AP of the collected results.
    pycocotools / cv2 are not needed: the annotation json is read directly, images through
    `imread` (default imread_bgr), the overlay is written as .npy when cv2 is absent."""
    import json
    import os
    from . import common, oks_eval
    config = config or dec.default_config()
    imread = imread or imread_bgr
    with open(anno_file) as f:
        ann = json.load(f)
    person_cat = [c["id"] for c in ann.get("categories", []) if c.get("name") == "person"] or [1]
    img_ids = sorted({a["image_id"] for a in ann["annotations"] if a.get("category_id", 1) in person_cat})
    files = {im["id"]: im["file_name"] for im in ann["images"]}
    if max_images:
        img_ids = img_ids[:max_images]
    print("Total number of validation images {}".format(len(img_ids)))
    outputs = []
    for i, iid in enumerate(img_ids):
        if i % 10 == 0 and i != 0:
            print("Processed {} images".format(i))
        ori = imread(os.path.join(image_dir, files[iid]))
        if ori is None:
            raise IOError("cannot read %s (no cv2 / PIL here: provide imread= or .npy images)" % files[iid])
        paf, heatmap, scale_img = get_outputs_gpu(ori, model, preprocess, config)
        humans = dec.paf_to_pose_cpp(heatmap, paf, config)
        if vis_dir:
            out = common.draw_humans(ori, humans)
            os.makedirs(vis_dir, exist_ok=True)
            try:
                import cv2
                cv2.imwrite(os.path.join(vis_dir, files[iid]), out)
            except ImportError:
                np.save(os.path.join(vis_dir, os.path.splitext(files[iid])[0] + ".npy"), out)
        up = int(config.MODEL.DOWNSAMPLE)
        upsample_keypoints = (heatmap.shape[0] * up / scale_img, heatmap.shape[1] * up / scale_img)
        append_result(iid, humans, upsample_keypoints, outputs, int(config.MODEL.NUM_KEYPOINTS))
    return oks_eval.eval_coco(outputs, anno_file, img_ids)


def _coco_person_images(anno_file, max_images=None):
    """run_eval's image list (coco_eval.py:248-252: the images that contain the person category)."""
    import json
    with open(anno_file) as f:
        ann = json.load(f)
    person_cat = [c["id"] for c in ann.get("categories", []) if c.get("name") == "person"] or [1]
    img_ids = sorted({a["image_id"] for a in ann["annotations"] if a.get("category_id", 1) in person_cat})
    info = {im["id"]: im for im in ann["images"]}
    if max_images:
        img_ids = img_ids[:max_images]
    return img_ids, info


def eval_batches(sizes, dest_size, factor, batch, by_source_size=False):
    """The batched evaluation schedule: image indices grouped by the padded network input size
    (hn, wn) crop_with_factor gives them (by the source size itself for TTA, whose scales must agree),
    each bucket cut into batches of <= `batch`; deterministic order (buckets by first occurrence).
    -> list of (key, [indices])."""
    buckets = {}
    for i, (h0, w0) in enumerate(sizes):
        key = (h0, w0) if by_source_size else prep_geometry(h0, w0, dest_size, factor)[2]
        buckets.setdefault(key, []).append(i)
    out = []
    for key, idx in buckets.items():
        for j in range(0, len(idx), batch):
            out.append((key, idx[j:j + batch]))
    return out


def run_eval_batched(image_dir, anno_file, vis_dir, model, preprocess, config=None, imread=None, max_images=None,
                     batch=32, tta_scales=None, tta_flip=False, rank=0, world=1, return_outputs=False):
    """evaluate/coco_eval.py:245-283 restructured for the GPU (BASELINE configs[2] and [4]).

    The reference loops over the person images one at a time: imread -> get_outputs (batch 1, maps
    to the host) -> paf_to_pose_cpp (maps back through numpy / SWIG) -> append_result.  Here the
    images are bucketed by their padded input size, a bucket is cut into batches of `batch`, and per
    batch: the uint8 images are uploaded (3 B/pixel), ONE kernel resizes / pads / normalises them
    into the plan's input buffer, one forward, the decoder reads the stage-6 maps where the net
    wrote them, and only the fixed-size result records come back.  Maps never cross PCIe.
    With tta_scales (+ tta_flip) every batch runs get_multiscale_outputs_batch instead (configs[2]).
    `world` ranks take the batches round-robin and exchange ONE all_gather of the record block per
    step (parallel.gather_records); every rank ends up with all results.
    The results are identical to run_eval's (same pixels, batch-invariant kernels, same decoder) and
    come out in run_eval's order.  Returns the OKS AP (or (AP, outputs) with return_outputs)."""
    import os
    from . import common, oks_eval, parallel as par
    config = config or dec.default_config()
    imread = imread or imread_bgr
    size, factor = int(config.DATASET.IMAGE_SIZE), int(config.MODEL.DOWNSAMPLE)
    nkp = int(config.MODEL.NUM_KEYPOINTS)
    img_ids, info = _coco_person_images(anno_file, max_images)
    print("Total number of validation images {}".format(len(img_ids)))
    m = _unwrap(model)
    dev = torch.device('cuda', torch.cuda.current_device())
    mode = {'rtpose': 0, 'vgg': 1}[preprocess]

    def load(i):
        ori = imread(os.path.join(image_dir, info[img_ids[i]]["file_name"]))
        if ori is None:
            raise IOError("cannot read %s (no cv2 / PIL here: provide imread= or .npy images)"
                          % info[img_ids[i]]["file_name"])
        return np.ascontiguousarray(ori, dtype=np.uint8)

    cache = {}
    sizes = []
    for i, iid in enumerate(img_ids):
        im = info[iid]
        if "height" in im and "width" in im:
            sizes.append((int(im["height"]), int(im["width"])))
        else:                                   # annotation without sizes: read the image now
            cache[i] = load(i)
            sizes.append(cache[i].shape[:2])
    sched = eval_batches(sizes, size, factor, batch, by_source_size=tta_scales is not None)
    steps = -(-len(sched) // world)
    max_peaks, max_humans = 64, 64
    bufs = None
    humans_of = {}                              # image index -> list[Human]
    stream = current_stream()
    for step in range(steps):
        mine = step * world + rank
        key, idx = sched[mine] if mine < len(sched) else (None, [])
        while True:                             # (repeats only when a device table overflowed)
            cfg = dec.make_cfg(config, max_peaks, max_humans)
            if bufs is None or bufs.cfg.max_peaks_per_part != max_peaks or bufs.cfg.max_humans != max_humans:
                bufs = dec.DecodeBuffers(cfg, batch, dev)
            if idx:
                imgs = [cache.pop(i) if i in cache else load(i) for i in idx]
                for i, im in zip(idx, imgs):
                    if tuple(im.shape[:2]) != tuple(sizes[i]):
                        raise _capi.RtposeError("image %s is %s, the annotation file says %s"
                                                % (info[img_ids[i]]["file_name"], im.shape[:2], sizes[i]))
                # One plan size per bucket: a short (tail) batch runs in the bucket's `batch`-image plan with its
                # first len(idx) slots filled - no extra plan (workspace allocation + zeroing) per tail shape, and
                # the plan cache of a COCO run stays at one plan per size bucket.  The kernels are batch-invariant,
                # so the results do not depend on what the unused slots hold.
                if tta_scales is not None:
                    padded = imgs + [imgs[-1]] * (batch - len(imgs))
                    paf_d, heat_d, _ = get_multiscale_outputs_batch(padded, model, preprocess, scales=tta_scales,
                                                                    flip=tta_flip, config=config)
                    hm, wm = heat_d.shape[1], heat_d.shape[2]
                    lheat, lpaf = _capi.Layout.dense(19, hm, wm), _capi.Layout.dense(38, hm, wm)
                    hbase, pbase = ptr(heat_d), ptr(paf_d)
                else:
                    hn, wn = key
                    plan = m.plan_for_shape(batch, hn, wn, dev)
                    up = [torch.from_numpy(im).to(dev, non_blocking=True) for im in imgs]
                    preprocess_into_plan(plan, up, [im.shape[:2] for im in imgs], size, mode, stream)
                    check(lib.rtpose_net_set_keep_intermediates(plan.handle, 0))
                    check(lib.rtpose_net_forward_prepared(plan.handle, stream), "rtpose_net_forward_prepared")
                    pbase, lpaf, _, hm, wm = m.output_view(plan, 0)
                    hbase, lheat, _, _, _ = m.output_view(plan, 1)
                # the record block always has `batch` slots (equal shapes for the gather); the decoder
                # fills the first len(idx)
                check(lib.rtpose_decode_batch(hbase, C_byref(lheat), pbase, C_byref(lpaf), len(idx), hm, wm,
                                              C_byref(cfg), ptr(bufs.workspace), bufs.workspace.numel() * 4,
                                              ptr(bufs.result), stream), "rtpose_decode_batch")
            rec_dev = bufs.result.view(batch, bufs.words)
            if world > 1:
                allrec = par.gather_records(rec_dev, world).cpu().numpy().reshape(world, batch, bufs.words)
            else:
                allrec = dec.fetch(bufs).reshape(1, batch, bufs.words)
            flags = 0
            for r in range(world):
                b = step * world + r
                nvalid = len(sched[b][1]) if b < len(sched) else 0
                if nvalid:
                    flags |= int(np.bitwise_or.reduce(allrec[r, :nvalid, dec.RES_HEADER + 2]))
            if flags & dec.OVERFLOW_PEAKS and max_peaks < dec.MAX_PEAKS_LIMIT:
                max_peaks = min(2 * max_peaks, dec.MAX_PEAKS_LIMIT)
                continue                        # every rank sees the same flags: all repeat the step together
            if flags & dec.OVERFLOW_HUMANS and max_humans < dec.MAX_HUMANS_LIMIT:
                max_humans = min(2 * max_humans, dec.MAX_HUMANS_LIMIT)
                continue
            if flags:
                raise _capi.RtposeError("decode tables overflowed at maximum capacity (flags=%d)" % flags)
            break
        for r in range(world):
            b = step * world + r
            if b >= len(sched):
                continue
            bkey, bidx = sched[b]
            for k, i in enumerate(bidx):
                rec = dec.parse_image(allrec[r, k], cfg)
                if tta_scales is not None:
                    s1, (hr1, wr1), _ = prep_geometry(bkey[0], bkey[1], size, factor)
                    hmap, wmap = -(-hr1 // factor), -(-wr1 // factor)
                else:
                    hmap, wmap = bkey[0] // factor, bkey[1] // factor
                humans_of[i] = (dec.humans_from_record(rec, wmap * factor, hmap * factor, nkp), hmap, wmap)
        if step % 10 == 0 and step:
            print("Processed {} images".format(min(step * world * batch, len(img_ids))))
    outputs = []
    for i, iid in enumerate(img_ids):           # run_eval's order
        humans, hmap, wmap = humans_of[i]
        scale_img = float(size) / min(sizes[i])
        if vis_dir and rank == 0:
            ori = load(i)
            out = common.draw_humans(ori, humans)
            os.makedirs(vis_dir, exist_ok=True)
            name = info[iid]["file_name"]
            try:
                import cv2
                cv2.imwrite(os.path.join(vis_dir, name), out)
            except ImportError:
                np.save(os.path.join(vis_dir, os.path.splitext(name)[0] + ".npy"), out)
        upsample_keypoints = (hmap * factor / scale_img, wmap * factor / scale_img)     # coco_eval.py:279
        append_result(iid, humans, upsample_keypoints, outputs, nkp)
    ap = oks_eval.eval_coco(outputs, anno_file, img_ids)
    return (ap, outputs) if return_outputs else ap
