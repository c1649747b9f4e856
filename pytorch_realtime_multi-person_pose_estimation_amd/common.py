"""Result containers of the pose decoder — same names and attributes as
lib/utils/common.py (CocoPart :5-24, Human :27-39/:63-67, BodyPart :253-274,
tables :276-284) so downstream code written against the reference keeps working.
The face / upper-body box helpers of the reference (:69-219) reference undefined
names (math, _include_part, _round) and cannot run there; they are not mirrored.
"""
from enum import Enum


class CocoPart(Enum):
    Nose = 0
    Neck = 1
    RShoulder = 2
    RElbow = 3
    RWrist = 4
    LShoulder = 5
    LElbow = 6
    LWrist = 7
    RHip = 8
    RKnee = 9
    RAnkle = 10
    LHip = 11
    LKnee = 12
    LAnkle = 13
    REye = 14
    LEye = 15
    REar = 16
    LEar = 17
    Background = 18


class BodyPart(object):
    """part_idx: part index (0 = nose); x, y: normalised coordinates; score: peak score."""
    __slots__ = ('uidx', 'part_idx', 'x', 'y', 'score')

    def __init__(self, uidx, part_idx, x, y, score):
        self.uidx = uidx
        self.part_idx = part_idx
        self.x, self.y = x, y
        self.score = score

    def get_part_name(self):
        return CocoPart(self.part_idx)

    def __str__(self):
        return 'BodyPart:%d-(%.2f, %.2f) score=%.2f' % (self.part_idx, self.x, self.y, self.score)

    __repr__ = __str__


class Human(object):
    """body_parts: dict part_idx -> BodyPart; score: mean per-part score of the person."""
    __slots__ = ('body_parts', 'pairs', 'uidx_list', 'score')

    def __init__(self, pairs):
        self.pairs = []
        self.uidx_list = set()
        self.body_parts = {}
        for pair in pairs:
            self.add_pair(pair)
        self.score = 0.0

    @staticmethod
    def _get_uidx(part_idx, idx):
        return '%d-%d' % (part_idx, idx)

    def add_pair(self, pair):
        self.pairs.append(pair)
        for pidx, idx, coord in ((pair.part_idx1, pair.idx1, pair.coord1), (pair.part_idx2, pair.idx2, pair.coord2)):
            uid = Human._get_uidx(pidx, idx)
            self.body_parts[pidx] = BodyPart(uid, pidx, coord[0], coord[1], pair.score)
            self.uidx_list.add(uid)

    def is_connected(self, other):
        return len(self.uidx_list & other.uidx_list) > 0

    def merge(self, other):
        for pair in other.pairs:
            self.add_pair(pair)

    def part_count(self):
        return len(self.body_parts)

    def get_max_score(self):
        return max(p.score for p in self.body_parts.values())

    def __str__(self):
        return ' '.join(str(x) for x in self.body_parts.values())

    __repr__ = __str__


CocoColors = [[255, 0, 0], [255, 85, 0], [255, 170, 0], [255, 255, 0], [170, 255, 0], [85, 255, 0], [0, 255, 0],
              [0, 255, 85], [0, 255, 170], [0, 255, 255], [0, 170, 255], [0, 85, 255], [0, 0, 255], [85, 0, 255],
              [170, 0, 255], [255, 0, 255], [255, 0, 170], [255, 0, 85]]

CocoPairs = [(1, 2), (1, 5), (2, 3), (3, 4), (5, 6), (6, 7), (1, 8), (8, 9), (9, 10), (1, 11),
             (11, 12), (12, 13), (1, 0), (0, 14), (14, 16), (0, 15), (15, 17), (2, 16), (5, 17)]
CocoPairsRender = CocoPairs[:-2]


def draw_humans(npimg, humans, imgcopy=False):
    """lib/utils/common.py:227-251 without cv2: filled 3-px discs at the joints and
    3-px lines along CocoPairsRender, drawn with numpy (visualisation only)."""
    import numpy as np
    if imgcopy:
        npimg = np.copy(npimg)
    h, w = npimg.shape[:2]

    def disc(cx, cy, r, col):
        y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, h), max(cx - r, 0), min(cx + r + 1, w)
        if y1 <= y0 or x1 <= x0:
            return
        yy, xx = np.mgrid[y0:y1, x0:x1]
        m = (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
        npimg[y0:y1, x0:x1][m] = col

    for human in humans:
        centers = {}
        for i in range(CocoPart.Background.value):
            if i not in human.body_parts:
                continue
            bp = human.body_parts[i]
            centers[i] = (int(bp.x * w + 0.5), int(bp.y * h + 0.5))
            disc(centers[i][0], centers[i][1], 4, CocoColors[i])
        for order, (a, b) in enumerate(CocoPairsRender):
            if a not in centers or b not in centers:
                continue
            (xa, ya), (xb, yb) = centers[a], centers[b]
            n = max(abs(xb - xa), abs(yb - ya), 1)
            for t in range(n + 1):
                disc(xa + (xb - xa) * t // n, ya + (yb - ya) * t // n, 1, CocoColors[order])
    return npimg
