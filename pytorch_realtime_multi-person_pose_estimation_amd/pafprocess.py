"""Drop-in for the SWIG module ``lib.pafprocess.pafprocess`` (reference
lib/pafprocess/pafprocess.i:14-15, pafprocess.h:53-59): same seven functions,
same argument meaning.  ``process_paf`` accepts what numpy.i's IN_ARRAY3
typemaps accept (numpy.i:1096-1126, :316-337): anything convertible to a
C-contiguous float32 3-D array, borrowed for the duration of the call; a wrong
rank raises TypeError like SWIG_fail does.  The work happens on the GPU inside
librtpose_mi355x.so (csrc/legacy_pafprocess.hip); results live in
process-global state until the next call, as in the reference.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import lib


def _in_array3(a, name):
    arr = np.ascontiguousarray(a, dtype=np.float32)
    if arr.ndim != 3:
        raise TypeError("%s: array must have 3 dimensions, given array has %d" % (name, arr.ndim))
    return arr


def process_paf(peaks, heatmap, pafmap):
    p = _in_array3(peaks, "peaks")
    h = _in_array3(heatmap, "heatmap")
    f = _in_array3(pafmap, "pafmap")
    rc = lib.process_paf(p.shape[0], p.shape[1], p.shape[2], C.c_void_p(p.ctypes.data), h.shape[0], h.shape[1],
                         h.shape[2], C.c_void_p(h.ctypes.data), f.shape[0], f.shape[1], f.shape[2],
                         C.c_void_p(f.ctypes.data))
    if rc != 0:  # the reference has no failure path; ours fails loudly (e.g. no GPU)
        raise _capi.RtposeError("process_paf failed (rc=%d): %s" % (rc, _capi.last_error()))
    return 0


def get_num_humans():
    return lib.get_num_humans()


def get_part_cid(human_id, part_id):
    return lib.get_part_cid(int(human_id), int(part_id))


def get_score(human_id):
    return lib.get_score(int(human_id))


def get_part_x(cid):
    return lib.get_part_x(int(cid))


def get_part_y(cid):
    return lib.get_part_y(int(cid))


def get_part_score(cid):
    return lib.get_part_score(int(cid))
