"""ORACLE (test infrastructure, not product code): multi-scale + horizontal-flip test-time
augmentation (BASELINE configs[2]; README.md:26 "we do left and right flip").

The surveyed reference commit keeps only remnants of this path (handle_paf_and_heat is imported
but never called, get_outputs is single-scale: SURVEY.md §3.2), so the COMPOSITION below is this
project's stated assumption; every step that the reference does define is taken from the
reference (make_golden passes the reference's own functions, executed unmodified; the default
`fns` are the pinned restatements of oracle/host_oracle.py):

  per scale s: crop_with_factor(img, round(368 s), factor=8)     im_transform.py:119-134
               rtpose_preprocess / vgg_preprocess                  preprocessing.py:16-43
               forward                                             rtpose_vgg.py:158-198
               flip pass: the resized image mirrored inside its valid (un-padded) width, the
               padding stays bottom / right; merged with handle_paf_and_heat on the valid
               map columns                                         coco_eval.py:197-242
  then each scale's maps are bilinearly resized (half-pixel centres, edge clamp) to the scale-1.0
  map size and averaged; the map origins coincide, only the zoom  im_scale_s / im_scale_1  differs.
"""
import numpy as np


def resize_bilinear(src, hd, wd, sy, sx):
    """src [hs, ws, C] float32 -> [hd, wd, C]; (sy, sx) = source pixels per destination pixel.
    float32 arithmetic in the order of csrc/layout_ops.hip:resize_bilinear_accum_kernel
    (== F.interpolate(bilinear, align_corners=False, scale given))."""
    hs, ws = src.shape[:2]
    f32 = np.float32

    def taps(nd, ns, s):
        f = np.maximum((np.arange(nd, dtype=f32) + f32(0.5)) * f32(s) - f32(0.5), f32(0))
        i0 = np.minimum(f.astype(np.int64), ns - 1)
        i1 = np.minimum(i0 + 1, ns - 1)
        l = np.minimum(f - i0.astype(f32), f32(1))
        return i0, i1, l.astype(f32)
    y0, y1, ly = taps(hd, hs, sy)
    x0, x1, lx = taps(wd, ws, sx)
    lx_, ly_ = lx[None, :, None], ly[:, None, None]
    top = src[y0][:, x0] * (f32(1) - lx_) + src[y0][:, x1] * lx_
    bot = src[y1][:, x0] * (f32(1) - lx_) + src[y1][:, x1] * lx_
    return (top * (f32(1) - ly_) + bot * ly_).astype(f32)


def default_fns():
    from . import host_oracle as ho
    return {"crop_with_factor": ho.crop_with_factor, "rtpose": ho.rtpose_preprocess, "vgg": ho.vgg_preprocess,
            "handle_paf_and_heat": ho.handle_paf_and_heat}


def multiscale(img, forward, preprocess='rtpose', scales=(0.5, 1.0, 1.5, 2.0), flip=True, base=368, stride=8,
               fns=None):
    """forward(x float32 [1,3,H,W]) -> (paf [h,w,38], heat [h,w,19]) numpy HWC.
    Returns (paf [hd,wd,38], heat [hd,wd,19], im_scale of the 1.0 pass)."""
    fns = fns or default_fns()
    h0, w0 = img.shape[:2]
    s1 = float(base) / min(h0, w0)
    hd = -(-int(np.rint(h0 * s1)) // stride)
    wd = -(-int(np.rint(w0 * s1)) // stride)
    acc_heat = np.zeros((hd, wd, 19), np.float32)
    acc_paf = np.zeros((hd, wd, 38), np.float32)
    a = np.float32(1.0 / len(scales))
    for si, s in enumerate(scales):
        crop, im_scale, real = fns["crop_with_factor"](img, int(round(base * s)), factor=stride, is_ceil=True)
        paf, heat = forward(fns[preprocess](crop)[None])
        if flip:
            vw = real[1]
            cropf = crop.copy()
            cropf[:, :vw] = crop[:, :vw][:, ::-1]
            paf_f, heat_f = forward(fns[preprocess](cropf)[None])
            vwm = -(-vw // stride)
            paf, heat = fns["handle_paf_and_heat"](heat[:, :vwm].copy(), heat_f[:, :vwm].copy(),
                                                   paf[:, :vwm].copy(), paf_f[:, :vwm].copy())
        ratio = np.float32(np.float32(hd * (im_scale / s1)) / np.float32(hd)), \
            np.float32(np.float32(wd * (im_scale / s1)) / np.float32(wd))
        rh_ = resize_bilinear(np.ascontiguousarray(heat, np.float32), hd, wd, ratio[0], ratio[1])
        rp_ = resize_bilinear(np.ascontiguousarray(paf, np.float32), hd, wd, ratio[0], ratio[1])
        acc_heat = (a * rh_) if si == 0 else (acc_heat + a * rh_)
        acc_paf = (a * rp_) if si == 0 else (acc_paf + a * rp_)
    return acc_paf.astype(np.float32), acc_heat.astype(np.float32), s1


TTA_CASE = {"h0": 120, "w0": 150, "seed": 61, "scales": (0.5, 1.0, 1.5), "preprocess": "rtpose"}


def tta_image():
    rng = np.random.default_rng(TTA_CASE["seed"])
    low = rng.integers(0, 256, (16, 20, 3))
    big = np.kron(low, np.ones((8, 8, 1)))[:TTA_CASE["h0"], :TTA_CASE["w0"]]
    return np.clip(big + rng.integers(-10, 11, big.shape), 0, 255).astype(np.uint8)


def make_golden():
    """BUILD CONTAINER ONLY: the composition above over the reference's own functions and module."""
    import torch
    from . import ref_harness as rh
    rh.install()
    from evaluate.coco_eval import handle_paf_and_heat
    from lib.network.im_transform import crop_with_factor
    from lib.datasets.preprocessing import rtpose_preprocess, vgg_preprocess
    from lib.network.rtpose_vgg import get_model
    model = get_model('vgg19')
    model.load_state_dict(rh.he_init_reference_state_dict(0))
    model.eval()

    def forward(x):
        with torch.no_grad():
            (paf, heat), _ = model(torch.from_numpy(x))
        return paf[0].permute(1, 2, 0).numpy(), heat[0].permute(1, 2, 0).numpy()
    fns = {"crop_with_factor": crop_with_factor, "rtpose": rtpose_preprocess, "vgg": vgg_preprocess,
           "handle_paf_and_heat": handle_paf_and_heat}
    img = tta_image()
    out = {}
    for tag, flip in (("flip", True), ("noflip", False)):
        paf, heat, s1 = multiscale(img, forward, TTA_CASE["preprocess"], TTA_CASE["scales"], flip, fns=fns)
        out["tta_%s_paf" % tag], out["tta_%s_heat" % tag] = paf, heat
    out["tta_s1"] = np.float64(s1)
    print("tta golden: maps %s, max|paf| %.3f" % (paf.shape, np.abs(paf).max()))
    return out
