"""evaluate/coco_eval.py surface: get_outputs (:80), handle_paf_and_heat (:197), append_result (:117).
Unlike the reference this module does not parse sys.argv at import time (coco_eval.py:22-34)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _rtpose_pkg import module  # noqa: E402
from lib.config import cfg  # noqa: E402

_pre = module("preprocess")
ORDER_COCO = _pre.ORDER_COCO
handle_paf_and_heat = _pre.handle_paf_and_heat


_oks = module("oks_eval")


def eval_coco(outputs, annFile, imgIds):
    """coco_eval.py:55-75 without pycocotools: OKS AP @[.5:.95] (stats[0])."""
    return _oks.eval_coco(outputs, annFile, imgIds)


def get_outputs(img, model, preprocess):
    return _pre.get_outputs(img, model, preprocess, cfg)


def append_result(image_id, humans, upsample_keypoints, outputs):
    return _pre.append_result(image_id, humans, upsample_keypoints, outputs, cfg.MODEL.NUM_KEYPOINTS)


def run_eval(image_dir, anno_file, vis_dir, model, preprocess):
    """coco_eval.py:245-290 - what evaluate/evaluation.py calls."""
    return _pre.run_eval(image_dir, anno_file, vis_dir, model, preprocess, cfg)


def run_eval_batched(image_dir, anno_file, vis_dir, model, preprocess, **kw):
    """run_eval restructured for the GPU: images bucketed by padded size, batched, maps never leave HBM
    (same results, same order; see preprocess.run_eval_batched for batch= / tta_scales= / rank= / world=)."""
    return _pre.run_eval_batched(image_dir, anno_file, vis_dir, model, preprocess, cfg, **kw)
