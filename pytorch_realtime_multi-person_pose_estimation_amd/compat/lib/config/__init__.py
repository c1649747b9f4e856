"""lib/config surface: `cfg` + `update_config` without yacs.  Only the keys the hot path reads
(lib/config/default.py:40,41,69,126,127,128) plus a yaml overlay; the training keys are out of scope."""
import types


def _ns(**kw):
    return types.SimpleNamespace(**kw)


cfg = _ns(
    MODEL=_ns(NAME='vgg19', NUM_KEYPOINTS=18, DOWNSAMPLE=8, TARGET_TYPE='gaussian', IMAGE_SIZE=[368, 368]),
    DATASET=_ns(IMAGE_SIZE=368),
    TEST=_ns(THRESH_HEATMAP=0.1, THRESH_PAF=0.05, NUM_INTERMED_PTS_BETWEEN_KEYPOINTS=10, FLIP_TEST=False),
)


def update_config(cfg, args):
    """default.py:139-168: overlay a yaml file (args.cfg) and KEY VALUE pairs (args.opts)."""
    path = getattr(args, "cfg", None)
    if path:
        try:
            import yaml
            with open(path) as f:
                data = yaml.safe_load(f) or {}
        except (OSError, ImportError):
            data = {}
        for sect, vals in data.items():
            if isinstance(vals, dict) and hasattr(cfg, sect):
                for k, v in vals.items():
                    setattr(getattr(cfg, sect), k, v)
    opts = getattr(args, "opts", None) or []
    for key, val in zip(opts[0::2], opts[1::2]):
        sect, _, name = key.partition(".")
        if hasattr(cfg, sect):
            old = getattr(getattr(cfg, sect), name, None)
            setattr(getattr(cfg, sect), name, type(old)(val) if old is not None else val)
    return cfg
