"""ORACLE (test infrastructure, not product code): numpy restatements of the caller-side helpers
either side of the hot path.  Pinned to the reference's own functions executed unmodified
(oracle/make_golden_host.py -> tests/golden/host_ref.npz; tests/test_oracle_cpu.py replays them).

  crop_with_factor     lib/network/im_transform.py:119-134 (+ _factor_closest :113-116)
  rtpose_preprocess    lib/datasets/preprocessing.py:16-21
  vgg_preprocess       lib/datasets/preprocessing.py:32-43
  handle_paf_and_heat  evaluate/coco_eval.py:197-242

cv2.resize is oracle/cv2_restate.py (restated from OpenCV's published algorithm: parity
unpinned against a real OpenCV build, cv2 is absent here).
"""
import numpy as np

from . import cv2_restate as cv2

SWAP_HEAT = np.array((0, 1, 5, 6, 7, 2, 3, 4, 11, 12, 13, 8, 9, 10, 15, 14, 17, 16, 18))        # coco_eval.py:207-208
SWAP_PAF = np.array((6, 7, 8, 9, 10, 11, 0, 1, 2, 3, 4, 5, 20, 21, 22, 23, 24, 25, 26, 27, 12, 13, 14, 15,
                     16, 17, 18, 19, 28, 29, 32, 33, 30, 31, 36, 37, 34, 35))                     # :228-230


def _factor_closest(num, factor, is_ceil=True):
    num = np.ceil(float(num) / factor) if is_ceil else np.floor(float(num) / factor)
    return int(num) * factor


def crop_with_factor(im, dest_size=None, factor=32, is_ceil=True):
    im_scale = float(dest_size) / np.min(im.shape[0:2])                    # :124
    im = cv2.resize(im, None, fx=im_scale, fy=im_scale)                   # :126 (INTER_LINEAR)
    h, w, c = im.shape
    new_h, new_w = _factor_closest(h, factor, is_ceil), _factor_closest(w, factor, is_ceil)
    im_croped = np.zeros([new_h, new_w, c], dtype=im.dtype)               # :130-131 zero pad, bottom / right
    im_croped[0:h, 0:w, :] = im
    return im_croped, im_scale, im.shape


def rtpose_preprocess(image):
    image = image.astype(np.float32)
    image = image / 256. - 0.5
    return image.transpose((2, 0, 1)).astype(np.float32)


def vgg_preprocess(image):
    image = image.astype(np.float32) / 255.
    means, stds = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    out = image.copy()[:, :, ::-1]                                        # BGR -> RGB
    for i in range(3):
        out[:, :, i] = out[:, :, i] - means[i]
        out[:, :, i] = out[:, :, i] / stds[i]
    return out.transpose((2, 0, 1)).astype(np.float32)


def handle_paf_and_heat(normal_heat, flipped_heat, normal_paf, flipped_paf):
    """-> (averaged_paf, averaged_heatmap).  Unlike the reference (:236-237 write through a view),
    the caller's flipped_paf is left untouched."""
    fp = flipped_paf[:, ::-1, :].copy()
    fp[:, :, SWAP_PAF[::2]] = -fp[:, :, SWAP_PAF[::2]]                     # :237 (:236 is a no-op)
    averaged_paf = (normal_paf + fp[:, :, SWAP_PAF]) / 2.
    averaged_heatmap = (normal_heat + flipped_heat[:, ::-1, :][:, :, SWAP_HEAT]) / 2.
    return averaged_paf, averaged_heatmap
