"""Self-contained COCO keypoint evaluation (OKS AP) — stands in for the pycocotools calls of
evaluate/coco_eval.py:55-75 (`eval_coco`: COCO.loadRes + COCOeval(iouType='keypoints') + stats[0]).
pycocotools is not in this image; the algorithm is restated from the published COCO keypoint
evaluation protocol (cocodataset.org/#keypoints-eval; COCOeval.computeOks / evaluateImg /
accumulate): OKS with the 17 per-keypoint sigmas, detections sorted by score (max 20 per image),
greedy matching per OKS threshold 0.50:0.05:0.95 preferring non-crowd ground truth, crowd / zero-
keypoint ground truth ignored, 101-point interpolated precision, area ranges all/medium/large.
**Parity unpinned** against pycocotools (absent) and unverifiable without COCO val2017.
Host-side Python: this is evaluation bookkeeping, not the hot path.
"""
import json

import numpy as np

SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0
OKS_THRS = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
AREA_RNG = {"all": (0, 1e5 ** 2), "medium": (32 ** 2, 96 ** 2), "large": (96 ** 2, 1e5 ** 2)}
MAX_DETS = 20


def compute_oks(gts, dts):
    """[len(dts), len(gts)] OKS matrix (COCOeval.computeOks)."""
    if not gts or not dts:
        return np.zeros((len(dts), len(gts)))
    ious = np.zeros((len(dts), len(gts)))
    var = (SIGMAS * 2) ** 2
    for j, gt in enumerate(gts):
        g = np.array(gt["keypoints"], dtype=np.float64)
        xg, yg, vg = g[0::3], g[1::3], g[2::3]
        k1 = np.count_nonzero(vg > 0)
        bb = gt["bbox"]
        x0, x1 = bb[0] - bb[2], bb[0] + bb[2] * 2
        y0, y1 = bb[1] - bb[3], bb[1] + bb[3] * 2
        for i, dt in enumerate(dts):
            d = np.array(dt["keypoints"], dtype=np.float64)
            xd, yd = d[0::3], d[1::3]
            if k1 > 0:
                dx, dy = xd - xg, yd - yg
            else:   # no annotated keypoints: distance to the (doubled) box
                z = np.zeros(len(SIGMAS))
                dx = np.max((z, x0 - xd), axis=0) + np.max((z, xd - x1), axis=0)
                dy = np.max((z, y0 - yd), axis=0) + np.max((z, yd - y1), axis=0)
            e = (dx ** 2 + dy ** 2) / var / (gt["area"] + np.spacing(1)) / 2
            if k1 > 0:
                e = e[vg > 0]
            ious[i, j] = np.sum(np.exp(-e)) / e.shape[0]
    return ious


def _evaluate_img(gts, dts, area_rng):
    for g in gts:
        g["_ignore"] = bool(g.get("iscrowd", 0)) or g.get("num_keypoints", 0) == 0 or \
            g["area"] < area_rng[0] or g["area"] > area_rng[1]
    gts = sorted(gts, key=lambda g: g["_ignore"])
    dts = sorted(dts, key=lambda d: -d["score"])[:MAX_DETS]
    ious = compute_oks(gts, dts)
    T, G, D = len(OKS_THRS), len(gts), len(dts)
    gtm = -np.ones((T, G), dtype=np.int64)
    dtm = -np.ones((T, D), dtype=np.int64)
    g_ig = np.array([g["_ignore"] for g in gts], dtype=bool)
    dt_ig = np.zeros((T, D), dtype=bool)
    for ti, t in enumerate(OKS_THRS):
        for di in range(D):
            iou = min(t, 1 - 1e-10)
            m = -1
            for gi in range(G):
                if gtm[ti, gi] >= 0 and not gts[gi].get("iscrowd", 0):
                    continue
                if m > -1 and not g_ig[m] and g_ig[gi]:
                    break
                if ious[di, gi] < iou:
                    continue
                iou = ious[di, gi]
                m = gi
            if m == -1:
                continue
            dt_ig[ti, di] = g_ig[m]
            dtm[ti, di] = m
            gtm[ti, m] = di
    # unmatched detections outside the area range are ignored
    d_area = np.array([d.get("area", 0.0) for d in dts])
    out = np.logical_or(d_area < area_rng[0], d_area > area_rng[1])
    dt_ig = np.logical_or(dt_ig, np.logical_and(dtm < 0, np.repeat(out[None, :], T, 0)))
    return {"dt_scores": [d["score"] for d in dts], "dtm": dtm, "dt_ig": dt_ig, "g_ig": g_ig}


def _kp_area(d):
    k = np.array(d["keypoints"], dtype=np.float64)
    x, y = k[0::3], k[1::3]
    return float((x.max() - x.min()) * (y.max() - y.min()))   # loadRes: area of the keypoint bounding box


def evaluate(gt_annotations, results, img_ids=None):
    """gt_annotations: list of COCO 'annotations' dicts (category person); results: list of
    {image_id, category_id, keypoints[51], score}.  Returns dict(AP, AP50, AP75, APm, APl)."""
    by_img_gt, by_img_dt = {}, {}
    for a in gt_annotations:
        by_img_gt.setdefault(a["image_id"], []).append(dict(a))
    for r in results:
        r = dict(r)
        r.setdefault("area", _kp_area(r))
        by_img_dt.setdefault(r["image_id"], []).append(r)
    img_ids = sorted(set(img_ids if img_ids is not None else list(by_img_gt) + list(by_img_dt)))
    stats = {}
    for name, rng in AREA_RNG.items():
        evals = [_evaluate_img(by_img_gt.get(i, []), by_img_dt.get(i, []), rng) for i in img_ids]
        scores = np.concatenate([e["dt_scores"] for e in evals]) if evals else np.zeros(0)
        order = np.argsort(-scores, kind="mergesort")
        dtm = np.concatenate([e["dtm"] for e in evals], axis=1)[:, order] if evals else np.zeros((len(OKS_THRS), 0))
        dt_ig = np.concatenate([e["dt_ig"] for e in evals], axis=1)[:, order] if evals else np.zeros((len(OKS_THRS), 0), bool)
        npig = int(sum(np.count_nonzero(~e["g_ig"]) for e in evals))
        prec = -np.ones((len(OKS_THRS), len(REC_THRS)))
        if npig > 0:
            tps = np.logical_and(dtm >= 0, ~dt_ig)
            fps = np.logical_and(dtm < 0, ~dt_ig)
            tp_sum, fp_sum = np.cumsum(tps, axis=1, dtype=np.float64), np.cumsum(fps, axis=1, dtype=np.float64)
            for ti in range(len(OKS_THRS)):
                tp, fp = tp_sum[ti], fp_sum[ti]
                rc = tp / npig
                pr = tp / (fp + tp + np.spacing(1))
                pr = pr.tolist()
                for k in range(len(pr) - 1, 0, -1):
                    if pr[k] > pr[k - 1]:
                        pr[k - 1] = pr[k]
                inds = np.searchsorted(rc, REC_THRS, side="left")
                q = np.zeros(len(REC_THRS))
                for ri, pi in enumerate(inds):
                    if pi < len(pr):
                        q[ri] = pr[pi]
                prec[ti] = q
        stats[name] = prec

    def _ap(p, thr=None):
        if thr is not None:
            p = p[np.where(np.isclose(OKS_THRS, thr))[0]]
        p = p[p > -1]
        return float(np.mean(p)) if p.size else -1.0

    return {"AP": _ap(stats["all"]), "AP50": _ap(stats["all"], .5), "AP75": _ap(stats["all"], .75),
            "APm": _ap(stats["medium"]), "APl": _ap(stats["large"])}


def eval_coco(outputs, ann_file, img_ids):
    """evaluate/coco_eval.py:55-75 surface: returns stats[0] (AP @[.5:.95])."""
    with open(ann_file) as f:
        anns = [a for a in json.load(f)["annotations"] if a.get("category_id", 1) == 1]
    return evaluate(anns, outputs, img_ids)["AP"]
