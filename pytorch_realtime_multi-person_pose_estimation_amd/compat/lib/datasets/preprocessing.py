"""lib/datasets/preprocessing.py surface (the two normalisers the hot path uses)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _rtpose_pkg import module  # noqa: E402

_pre = module("preprocess")
rtpose_preprocess = _pre.rtpose_preprocess
vgg_preprocess = _pre.vgg_preprocess
