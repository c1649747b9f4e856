"""lib/network/rtpose_vgg.py surface -> MI355X implementation."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _rtpose_pkg import module  # noqa: E402

_net = module("network")
get_model = _net.get_model
use_vgg = _net.use_vgg
