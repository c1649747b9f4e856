"""lib/utils/common.py surface."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _rtpose_pkg import module  # noqa: E402

_c = module("common")
Human, BodyPart, CocoPart = _c.Human, _c.BodyPart, _c.CocoPart
CocoColors, CocoPairs, CocoPairsRender, draw_humans = _c.CocoColors, _c.CocoPairs, _c.CocoPairsRender, _c.draw_humans
