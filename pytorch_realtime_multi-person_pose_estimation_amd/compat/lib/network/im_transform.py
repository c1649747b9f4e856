"""lib/network/im_transform.py surface (crop_with_factor only; the rest is training-side)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _rtpose_pkg import module  # noqa: E402

_pre = module("preprocess")
crop_with_factor = _pre.crop_with_factor
_factor_closest = _pre._factor_closest
