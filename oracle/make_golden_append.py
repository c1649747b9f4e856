"""Generates tests/golden/append_result.json by executing the REFERENCE'S OWN `append_result`
(evaluate/coco_eval.py:117-154), unmodified, where it lies under /root/reference
(oracle/ref_harness.py registers stand-ins for the absent third-party modules) on seeded Human
lists built with the reference's own Human / BodyPart classes (lib/utils/common.py).

Run in the build container only:   python oracle/make_golden_append.py
The test (tests/test_host_golden_cpu.py) rebuilds the same humans from the stored seeds with the product's
Human / BodyPart and compares the records field by field (floats exactly: the arithmetic is two
multiply-adds in float64)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402

# (seed, people, probability that a part is present, (upsample_y, upsample_x), image id)
CASES = [(1, 3, 0.8, (368.0 / 0.7479, 392.0 / 0.7479), 139), (2, 1, 0.3, (46 * 8 / 1.15, 46 * 8 / 1.15), 785),
         (3, 0, 0.5, (100.0, 100.0), 7), (4, 6, 1.0, (640.0, 427.0), 42)]


def people(seed, n, p_present):
    """-> [[(part, x, y, score), ...] per human] from a seeded generator (shared with the test)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        parts = []
        for part in range(18):
            x, y, s, u = rng.random(), rng.random(), rng.random(), rng.random()
            if u < p_present:
                parts.append((part, float(x), float(y), float(s)))
        out.append(parts)
    return out


def build(humans_spec, Human, BodyPart):
    hs = []
    for hi, parts in enumerate(humans_spec):
        h = Human([])
        for part, x, y, s in parts:
            h.body_parts[part] = BodyPart('%d-%d' % (hi, part), part, x, y, s)
        h.score = 0.5
        hs.append(h)
    return hs


def main():
    rh.install()
    from evaluate.coco_eval import append_result
    from lib.utils.common import Human, BodyPart
    gold = []
    for seed, n, p, up, image_id in CASES:
        outputs = []
        append_result(image_id, build(people(seed, n, p), Human, BodyPart), up, outputs)
        gold.append([{"image_id": o["image_id"], "category_id": o["category_id"], "score": o["score"],
                      "keypoints": [float(v) for v in o["keypoints"]]} for o in outputs])
    path = os.path.join(ROOT, "tests", "golden", "append_result.json")
    with open(path, "w") as f:
        json.dump(gold, f)
    print("wrote", path, [len(g) for g in gold])


if __name__ == "__main__":
    main()
