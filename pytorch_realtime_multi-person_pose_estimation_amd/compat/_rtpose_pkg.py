"""Locates and imports the product package (its directory name contains a hyphen)."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
PKG = "pytorch_realtime_multi-person_pose_estimation_amd"


def module(name=""):
    return importlib.import_module(PKG + ("." + name if name else ""))
