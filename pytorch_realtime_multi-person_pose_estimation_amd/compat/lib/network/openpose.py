"""lib/network/openpose.py surface.  evaluate/evaluation.py imports OpenPose_Model and use_vgg
from here (it only instantiates them in a commented-out line); the experimental OpenPose_Model
backbone is outside the hot-path scope: importing works, constructing it says so."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _rtpose_pkg import module  # noqa: E402

use_vgg = module("network").use_vgg


class OpenPose_Model(object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("OpenPose_Model is not part of the MI355X hot path; use "
                                  "lib.network.rtpose_vgg.get_model('vgg19')")
