"""lib/pafprocess/pafprocess.py (SWIG-generated in the reference) surface."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _rtpose_pkg import module  # noqa: E402

_p = module("pafprocess")
process_paf = _p.process_paf
get_num_humans, get_part_cid, get_score = _p.get_num_humans, _p.get_part_cid, _p.get_score
get_part_x, get_part_y, get_part_score = _p.get_part_x, _p.get_part_y, _p.get_part_score
