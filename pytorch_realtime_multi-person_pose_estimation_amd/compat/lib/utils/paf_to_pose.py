"""lib/utils/paf_to_pose.py surface: paf_to_pose_cpp (:372) and NMS (:67) on the GPU."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _rtpose_pkg import module  # noqa: E402

_d = module("decode")
paf_to_pose_cpp = _d.paf_to_pose_cpp
NMS = _d.NMS
