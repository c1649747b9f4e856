"""ORACLE (test infrastructure, not product code): a stand-in for the `cv2` module.

The reference's hot-path Python (lib/utils/paf_to_pose.py, evaluate/coco_eval.py,
lib/network/im_transform.py, lib/datasets/preprocessing.py, lib/utils/common.py,
demo/picture_demo.py) does `import cv2` at module top; OpenCV is absent from this
image and unpinned by the reference.  oracle/ref_harness.py registers THIS module as
`cv2` in sys.modules so that the reference's own code can be executed unmodified; the
handful of cv2 functions it calls are restated here from OpenCV's published algorithms:

  resize INTER_LINEAR  uint8   -> oracle/cv_oracle.c      (im_transform.py:126)
  resize INTER_CUBIC   float32 -> oracle/post_oracle.c    (paf_to_pose.py:115)
  resize INTER_NEAREST any     -> index arithmetic        (paf_to_pose.py:382-385)
  imread                       -> PIL decode, RGB -> BGR  (picture_demo.py:51)
  imwrite                      -> kept in WRITTEN[path]   (picture_demo.py:64)
  circle / line                -> simple rasterisers      (common.py:240-249; overlay only)

PARITY UNPINNED against a real OpenCV build for every one of these (and PIL's libjpeg
decode of a .jpg may differ from OpenCV's bundled libjpeg by +-1 per pixel).
"""
import ctypes as C
import os

import numpy as np

INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA = 0, 1, 2, 3
IMREAD_COLOR = 1
LINE_8 = 8
__version__ = "oracle-restatement"

WRITTEN = {}     # imwrite(path, img) lands here

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _so():
    global _lib
    if _lib is None:
        lib = C.CDLL(os.path.join(_HERE, "liboracle_post.so"))
        lib.oracle_resize_cubic.restype = None
        lib.oracle_resize_cubic.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.oracle_resize_linear_u8.restype = C.c_int
        lib.oracle_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                                C.c_void_p]
        lib.oracle_resize_dsize.restype = None
        lib.oracle_resize_dsize.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int),
                                            C.POINTER(C.c_int)]
        _lib = lib
    return _lib


def _dsize(h, w, fx, fy):
    dh, dw = C.c_int(), C.c_int()
    _so().oracle_resize_dsize(h, w, float(fx), float(fy), C.byref(dh), C.byref(dw))
    return dh.value, dw.value


def resize(src, dsize=None, fx=0.0, fy=0.0, interpolation=INTER_LINEAR):
    if dsize is not None and tuple(dsize) != (0, 0):
        raise NotImplementedError("the reference only calls cv2.resize(src, None, fx=, fy=)")
    src = np.ascontiguousarray(src)
    h, w = src.shape[:2]
    if interpolation == INTER_NEAREST:
        dh, dw = _dsize(h, w, fx, fy)
        # resizeNN: sx = min(floor(dx * (1 / fx)), w - 1)
        ys = np.minimum(np.floor(np.arange(dh) * (1.0 / fy)).astype(np.int64), h - 1)
        xs = np.minimum(np.floor(np.arange(dw) * (1.0 / fx)).astype(np.int64), w - 1)
        return np.ascontiguousarray(src[ys][:, xs])
    if interpolation == INTER_CUBIC:
        if src.dtype != np.float32 or src.ndim != 2 or fx != fy or int(fx) != fx:
            raise NotImplementedError("INTER_CUBIC is restated for float32 planes and integer factors only")
        up = int(fx)
        out = np.empty((h * up, w * up), np.float32)
        _so().oracle_resize_cubic(src.ctypes.data, h, w, up, out.ctypes.data)
        return out
    if interpolation == INTER_LINEAR:
        if src.dtype != np.uint8:
            raise NotImplementedError("INTER_LINEAR is restated for uint8 images only")
        c = 1 if src.ndim == 2 else src.shape[2]
        dh, dw = _dsize(h, w, fx, fy)
        out = np.empty((dh, dw, c), np.uint8)
        rc = _so().oracle_resize_linear_u8(src.ctypes.data, h, w, c, float(fx), float(fy), out.ctypes.data)
        if rc != 0:
            raise ValueError("empty destination")
        return out if src.ndim == 3 else out[:, :, 0]
    raise NotImplementedError("interpolation %r" % (interpolation,))


def imread(path, flags=IMREAD_COLOR):
    from PIL import Image
    if not os.path.exists(path):
        return None
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


def imwrite(path, img):
    WRITTEN[path] = np.array(img, copy=True)
    return True


def circle(img, center, radius, color, thickness=1, lineType=8, shift=0):
    """Filled disc of radius `radius + thickness/2` (close to cv2's thick circle; overlay only)."""
    h, w = img.shape[:2]
    r = radius + max(thickness, 1) / 2.0
    cx, cy = center
    y0, y1 = max(0, int(cy - r - 1)), min(h, int(cy + r + 2))
    x0, x1 = max(0, int(cx - r - 1)), min(w, int(cx + r + 2))
    if y1 <= y0 or x1 <= x0:
        return img
    yy, xx = np.mgrid[y0:y1, x0:x1]
    m = (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
    img[y0:y1, x0:x1][m] = color
    return img


def line(img, pt1, pt2, color, thickness=1, lineType=8, shift=0):
    n = int(max(abs(pt2[0] - pt1[0]), abs(pt2[1] - pt1[1]))) + 1
    for t in np.linspace(0.0, 1.0, n):
        circle(img, (int(round(pt1[0] + t * (pt2[0] - pt1[0]))), int(round(pt1[1] + t * (pt2[1] - pt1[1])))),
               0, color, thickness)
    return img
