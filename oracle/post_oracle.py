"""ORACLE wrappers (test infrastructure, not product code).

ctypes doorways to
  * oracle/liboracle_post.so        — the plain-C restatement (post_oracle.c)
  * oracle/_ref/libpafprocess_ref.so — the reference's own pafprocess.cpp compiled
                                       unmodified (oracle/Makefile), when present
plus a numpy/scipy restatement of find_peaks that calls the very scipy function
the reference calls (lib/utils/paf_to_pose.py:34).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_POST = os.path.join(HERE, "liboracle_post.so")
_REF = os.path.join(HERE, "_ref", "libpafprocess_ref.so")

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def _post():
    if not os.path.exists(_POST):
        raise RuntimeError("oracle/liboracle_post.so missing: run `make -C oracle`")
    lib = C.CDLL(_POST)
    lib.oracle_nms.restype = C.c_int
    lib.oracle_nms.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _fp]
    lib.oracle_nms_ex.restype = C.c_int
    lib.oracle_nms_ex.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _fp,
                                  C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
    lib.oracle_gaussian_filter_f32.restype = None
    lib.oracle_gaussian_filter_f32.argtypes = [_fp, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
    lib.oracle_process_paf.restype = C.c_int
    lib.oracle_process_paf.argtypes = [_fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip,
                                       _fp, _ip, _ip, _fp, _ip, C.c_int]
    return lib


def have_ref():
    return os.path.exists(_REF)


_ref_lib = None


def _ref():
    global _ref_lib
    if _ref_lib is None:
        lib = C.CDLL(_REF)
        lib.ref_process_paf.restype = C.c_int
        lib.ref_process_paf.argtypes = [C.c_int] * 3 + [_fp] + [C.c_int] * 3 + [_fp] + [C.c_int] * 3 + [_fp]
        lib.ref_get_num_humans.restype = C.c_int
        lib.ref_get_part_cid.restype = C.c_int
        lib.ref_get_part_cid.argtypes = [C.c_int, C.c_int]
        lib.ref_get_score.restype = C.c_float
        lib.ref_get_score.argtypes = [C.c_int]
        for n in ("ref_get_part_x", "ref_get_part_y"):
            getattr(lib, n).restype = C.c_int
            getattr(lib, n).argtypes = [C.c_int]
        lib.ref_get_part_score.restype = C.c_float
        lib.ref_get_part_score.argtypes = [C.c_int]
        _ref_lib = lib
    return _ref_lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def gaussian_kernel1d(sigma=3.0, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, radius) with gaussian_filter1d's radius
    (int(truncate * sigma + 0.5)): the normalised float64 weights scipy correlates with."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    return np.ascontiguousarray(phi / phi.sum(), dtype=np.float64), radius


def gaussian_filter_f32(img, sigma=3.0):
    """C restatement of scipy.ndimage.gaussian_filter(img float32, sigma) (mode='reflect')."""
    a = _f32(img).copy()
    wts, r = gaussian_kernel1d(sigma)
    _post().oracle_gaussian_filter_f32(a.ctypes.data_as(_fp), a.shape[0], a.shape[1],
                                       wts.ctypes.data_as(C.POINTER(C.c_double)), r)
    return a


def nms(heat, num_keypoints=18, thr=0.1, up=8, cap=4096, refine=True, gaussian=False):
    """heat HWC float32 -> joint_list float32 [P,5] = (x, y, score, id, part).
    refine / gaussian = NMS's bool_refine_center / bool_gaussian_filt (paf_to_pose.py:67)."""
    heat = _f32(heat)
    h, w, c = heat.shape
    out = np.zeros((cap, 5), np.float32)
    wts, r = gaussian_kernel1d(3.0)
    n = _post().oracle_nms_ex(heat.ctypes.data_as(_fp), h, w, c, num_keypoints, np.float32(thr), up, cap,
                              out.ctypes.data_as(_fp), 1 if refine else 0, 1 if gaussian else 0,
                              wts.ctypes.data_as(C.POINTER(C.c_double)), r)
    if n < 0:
        raise RuntimeError("oracle_nms: more than %d peaks" % cap)
    return out[:n].copy()


def find_peaks_scipy(thr, img):
    """lib/utils/paf_to_pose.py:25-38 verbatim semantics, via the same scipy calls."""
    from scipy.ndimage import generate_binary_structure, maximum_filter
    peaks_binary = (maximum_filter(img, footprint=generate_binary_structure(2, 1)) == img) * (img > thr)
    return np.array(np.nonzero(peaks_binary)[::-1]).T


def process_paf(joint_list, paf, up=8, max_humans=512, libstdcxx_sort=True):
    """C restatement.  paf HWC [h,w,38] at network resolution.  Returns dict.
    libstdcxx_sort=True (the contract since round 5: "identical to pafprocess.cpp built with this
    image's g++ 11") orders equal-score candidates the way the reference binary's std::sort
    does; False = ties towards the lower (idx1, idx2), the product's contract of rounds 1-4,
    kept because the two must agree on every scene whose result has had_ties False."""
    jl = _f32(joint_list).reshape(-1, 5)
    paf = _f32(paf)
    h, w, _ = paf.shape
    p = jl.shape[0]
    parts = np.zeros((max_humans, 18), np.int32)
    score = np.zeros(max_humans, np.float32)
    lx = np.zeros(max(p, 1), np.int32)
    ly = np.zeros(max(p, 1), np.int32)
    ls = np.zeros(max(p, 1), np.float32)
    ties = C.c_int(0)
    n = _post().oracle_process_paf(jl.ctypes.data_as(_fp), p, paf.ctypes.data_as(_fp), h, w, up, h * up,
                                   max_humans, parts.ctypes.data_as(_ip), score.ctypes.data_as(_fp),
                                   lx.ctypes.data_as(_ip), ly.ctypes.data_as(_ip), ls.ctypes.data_as(_fp),
                                   C.byref(ties), 1 if libstdcxx_sort else 0)
    if n < 0:
        raise RuntimeError("oracle_process_paf failed (%d)" % n)
    return {"parts": parts[:n].copy(), "score": score[:n].copy(), "line_x": lx[:p], "line_y": ly[:p],
            "line_score": ls[:p], "had_ties": bool(ties.value)}


def upsample_nearest(a, up):
    """cv2.resize(a, None, fx=up, fy=up, INTER_NEAREST) for integer up (paf_to_pose.py:382-385)."""
    return np.repeat(np.repeat(a, up, axis=0), up, axis=1)


def ref_process_paf(joint_list, heat_up, paf_up):
    """The compiled reference C++ on already up-sampled HWC maps.  Returns dict like process_paf."""
    lib = _ref()
    jl = _f32(joint_list).reshape(1, -1, 5)
    heat_up = _f32(heat_up)
    paf_up = _f32(paf_up)
    lib.ref_process_paf(1, jl.shape[1], 5, jl.ctypes.data_as(_fp), heat_up.shape[0], heat_up.shape[1],
                        heat_up.shape[2], heat_up.ctypes.data_as(_fp), paf_up.shape[0], paf_up.shape[1],
                        paf_up.shape[2], paf_up.ctypes.data_as(_fp))
    n = lib.ref_get_num_humans()
    parts = np.array([[lib.ref_get_part_cid(h, p) for p in range(18)] for h in range(n)], np.int32).reshape(n, 18)
    score = np.array([lib.ref_get_score(h) for h in range(n)], np.float32)
    p = jl.shape[1]
    return {"parts": parts, "score": score,
            "line_x": np.array([lib.ref_get_part_x(c) for c in range(p)], np.int32),
            "line_y": np.array([lib.ref_get_part_y(c) for c in range(p)], np.int32),
            "line_score": np.array([lib.ref_get_part_score(c) for c in range(p)], np.float32)}


def paf_to_pose(heat, paf, num_keypoints=18, thr=0.1, up=8):
    """lib/utils/paf_to_pose.py:372-406 on the oracle pieces -> (joint_list, result dict)."""
    jl = nms(heat, num_keypoints, thr, up)
    if jl.shape[0] == 0:
        return jl, {"parts": np.zeros((0, 18), np.int32), "score": np.zeros(0, np.float32),
                    "line_x": np.zeros(0, np.int32), "line_y": np.zeros(0, np.int32),
                    "line_score": np.zeros(0, np.float32), "had_ties": False}
    return jl, process_paf(jl, paf, up)
