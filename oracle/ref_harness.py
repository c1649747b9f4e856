"""ORACLE (test infrastructure; BUILD CONTAINER ONLY — needs /root/reference).

Executes the reference's OWN Python for the hot path, unmodified, where it lies under
/root/reference, by registering stand-ins for the third-party modules that are absent
from this image before importing it (the same trick oracle/shufflenet_oracle.py plays
with `network.slim`):

  cv2                       -> oracle/cv2_restate.py (restated resize / imread / drawing)
  yacs.config.CfgNode       -> a small attribute-dict with merge_from_file / merge_from_list
  pycocotools(.coco/.cocoeval), matplotlib, pylab -> empty shells (never called on this path)
  lib.datasets (package __init__ only) -> empty shell: it imports the training loader
                               (torchvision ...); lib/datasets/preprocessing.py itself is the real file
  lib.pafprocess.pafprocess -> the reference's pafprocess.cpp compiled unmodified
                               (oracle/_ref/libpafprocess_ref.so) behind the SWIG module's
                               surface (pafprocess.i:14-15 + numpy.i IN_ARRAY3 conversion)
  Tensor.cuda / Module.cuda -> identity (picture_demo.py:47, coco_eval.py:108 hard-code .cuda())
  torch.load                -> the seeded He-init state_dict (pose_model.pth is not available offline)

Used by oracle/make_golden_host.py to produce tests/golden/*.npz; nothing here travels to
the GPU box and nothing in tests/ imports it.
"""
import contextlib
import os
import runpy
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("RTPOSE_REFERENCE", "/root/reference")


class CfgNode(dict):
    """The slice of yacs.config.CfgNode that lib/config/default.py uses."""

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super(CfgNode, self).__init__(init_dict or {})
        self.__dict__["_frozen"] = False

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self.__dict__.get("_frozen"):
            raise AttributeError("frozen CfgNode")
        self[k] = v

    def defrost(self):
        self.__dict__["_frozen"] = False
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def freeze(self):
        self.__dict__["_frozen"] = True
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                old = self.get(k)
                if isinstance(old, float) and isinstance(v, int):
                    v = float(v)
                if isinstance(old, tuple) and isinstance(v, list):
                    v = tuple(v)
                self[k] = v

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        import ast
        opts = list(opts or [])
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            try:
                val = ast.literal_eval(val)
            except (ValueError, SyntaxError):
                pass
            node[parts[-1]] = val


def _pafprocess_module():
    """lib/pafprocess/pafprocess.py as SWIG would generate it, over the compiled reference."""
    from oracle import post_oracle as po
    lib = po._ref()
    m = types.ModuleType("lib.pafprocess.pafprocess")

    def _in3(a, name):
        a = np.ascontiguousarray(np.asarray(a), dtype=np.float32)     # numpy.i:316-337
        if a.ndim != 3:
            raise TypeError("Array must have 3 dimensions.  Given array has %d dimensions (%s)" % (a.ndim, name))
        return a

    def process_paf(peaks, heatmap, pafmap):
        import ctypes as C
        fp = C.POINTER(C.c_float)
        p, h, f = _in3(peaks, "peaks"), _in3(heatmap, "heatmap"), _in3(pafmap, "pafmap")
        return lib.ref_process_paf(p.shape[0], p.shape[1], p.shape[2], p.ctypes.data_as(fp),
                                   h.shape[0], h.shape[1], h.shape[2], h.ctypes.data_as(fp),
                                   f.shape[0], f.shape[1], f.shape[2], f.ctypes.data_as(fp))

    m.process_paf = process_paf
    m.get_num_humans = lambda: lib.ref_get_num_humans()
    m.get_part_cid = lambda h, p: lib.ref_get_part_cid(int(h), int(p))
    m.get_score = lambda h: lib.ref_get_score(int(h))
    m.get_part_x = lambda c: lib.ref_get_part_x(int(c))
    m.get_part_y = lambda c: lib.ref_get_part_y(int(c))
    m.get_part_score = lambda c: lib.ref_get_part_score(int(c))
    return m


_installed = False


def install(argv=None):
    """Register the stand-ins and put /root/reference first on sys.path."""
    global _installed
    if not os.path.isdir(REF):
        raise RuntimeError("%s not present: the reference harness only runs in the build container" % REF)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import cv2_restate
    if not _installed:
        sys.modules["cv2"] = cv2_restate
        yacs = types.ModuleType("yacs")
        yacs_config = types.ModuleType("yacs.config")
        yacs_config.CfgNode = CfgNode
        yacs.config = yacs_config
        sys.modules["yacs"], sys.modules["yacs.config"] = yacs, yacs_config
        pct = types.ModuleType("pycocotools")
        coco = types.ModuleType("pycocotools.coco")
        cocoeval = types.ModuleType("pycocotools.cocoeval")

        class _Absent(object):
            def __init__(self, *a, **k):
                raise RuntimeError("pycocotools is absent from this image (stub)")
        coco.COCO, cocoeval.COCOeval = _Absent, _Absent
        sys.modules.update({"pycocotools": pct, "pycocotools.coco": coco, "pycocotools.cocoeval": cocoeval})
        for name in ("matplotlib", "pylab"):
            sys.modules.setdefault(name, types.ModuleType(name))
        sys.path.insert(0, REF)
        import lib                       # the reference package
        # lib/datasets/__init__.py drags in the TRAINING data loader (torchvision, matplotlib.pyplot,
        # scipy.misc: out of scope and absent); register the package shell without running it so that
        # `lib.datasets.preprocessing` (the file on the hot path) is still imported from the real file
        ds = types.ModuleType("lib.datasets")
        ds.__path__ = [os.path.join(REF, "lib", "datasets")]
        sys.modules["lib.datasets"] = ds
        lib.datasets = ds
        import lib.pafprocess            # (its SWIG module was never built)
        pm = _pafprocess_module()
        sys.modules["lib.pafprocess.pafprocess"] = pm
        lib.pafprocess.pafprocess = pm
        _installed = True
    # coco_eval.py:22-35 and picture_demo.py:28-40 parse sys.argv at import
    sys.argv = list(argv) if argv is not None else ["ref_harness", "--cfg",
                                                    os.path.join(REF, "experiments", "vgg19_368x368_sgd.yaml")]
    return cv2_restate


@contextlib.contextmanager
def cpu_as_cuda(state_dict=None):
    """.cuda() -> identity, torch.load -> state_dict, for the duration of the block."""
    import torch
    saved = (torch.Tensor.cuda, torch.nn.Module.cuda, torch.load)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    if state_dict is not None:
        torch.load = lambda *a, **k: state_dict
    try:
        yield
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda, torch.load = saved


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return "%s, os.cpu_count()=%d" % (line.split(":", 1)[1].strip(), os.cpu_count())
    except OSError:
        pass
    return "unknown CPU, os.cpu_count()=%s" % os.cpu_count()


def he_init_reference_state_dict(seed=0):
    """The seeded He-init weights every test uses, keyed like the reference module's state_dict."""
    install()
    from lib.network.rtpose_vgg import get_model
    from oracle import net_oracle
    return net_oracle.he_init_state_dict(get_model('vgg19'), seed=seed)


def run_picture_demo(seed=0):
    """demo/picture_demo.py executed UNMODIFIED (runpy) on readme/ski.jpg, CPU only.  Returns its
    globals of interest + wall-clock of the two hot calls (config 1: the reference plumbing)."""
    argv = ["picture_demo.py", "--cfg", os.path.join(REF, "experiments", "vgg19_368x368_sgd.yaml"),
            "--weight", "he_init"]
    cv2r = install(argv)
    sd = he_init_reference_state_dict(seed)
    sys.argv = list(argv)
    import evaluate.coco_eval as ce
    import lib.utils.paf_to_pose as p2p
    timing = {}

    def timed(name, fn):
        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            timing[name] = time.perf_counter() - t0
            return r
        return w
    orig = (ce.get_outputs, p2p.paf_to_pose_cpp)
    ce.get_outputs = timed("get_outputs_s", orig[0])
    p2p.paf_to_pose_cpp = timed("paf_to_pose_cpp_s", orig[1])
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        with cpu_as_cuda(sd):
            g = runpy.run_path(os.path.join(REF, "demo", "picture_demo.py"), run_name="__main__")
    finally:
        os.chdir(cwd)
        ce.get_outputs, p2p.paf_to_pose_cpp = orig
    return {"oriImg_before_draw": cv2r.imread(os.path.join(REF, "readme", "ski.jpg")), "paf": g["paf"],
            "heatmap": g["heatmap"], "im_scale": g["im_scale"], "humans": g["humans"], "out": g["out"],
            "written": dict(cv2r.WRITTEN), "timing": timing, "state_dict": sd, "cfg": g["cfg"]}
