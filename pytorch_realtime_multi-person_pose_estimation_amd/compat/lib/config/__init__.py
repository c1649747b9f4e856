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


def _decode_opt(val):
    """yacs' _decode_cfg_value: a string is parsed as a Python literal when it is one ('False' ->
    False, '[368, 368]' -> list, '0.05' -> float), otherwise it stays a string."""
    if not isinstance(val, str):
        return val
    import ast
    try:
        return ast.literal_eval(val)
    except (ValueError, SyntaxError):
        return val


def _coerce(new, old, key):
    """yacs' _check_and_coerce_cfg_value_type: same type passes, int <-> float and tuple <-> list are
    converted, anything else is an error (instead of bool('False') == True)."""
    if old is None or type(new) is type(old):
        return new
    if isinstance(old, bool) or isinstance(new, bool):
        raise ValueError("Type mismatch for %s: %r (%s) vs %r (%s)" % (key, old, type(old).__name__, new,
                                                                       type(new).__name__))
    if isinstance(old, float) and isinstance(new, int):
        return float(new)
    if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
        return type(old)(new)
    raise ValueError("Type mismatch for %s: %r (%s) vs %r (%s)" % (key, old, type(old).__name__, new,
                                                                   type(new).__name__))


def update_config(cfg, args):
    """default.py:139-168: overlay a yaml file (args.cfg) and KEY VALUE pairs (args.opts).
    An unreadable cfg file is an error, as it is with yacs' merge_from_file."""
    path = getattr(args, "cfg", None)
    if path:
        import yaml                       # (PyYAML; absent -> ImportError, loudly)
        with open(path) as f:             # missing / unreadable -> OSError, loudly
            data = yaml.safe_load(f) or {}
        for sect, vals in data.items():
            if isinstance(vals, dict) and hasattr(cfg, sect):
                for k, v in vals.items():
                    setattr(getattr(cfg, sect), k, v)
    opts = getattr(args, "opts", None) or []
    if len(opts) % 2:
        raise ValueError("opts must be KEY VALUE pairs, got %r" % (opts,))
    for key, val in zip(opts[0::2], opts[1::2]):
        sect, _, name = key.partition(".")
        if hasattr(cfg, sect):
            old = getattr(getattr(cfg, sect), name, None)
            setattr(getattr(cfg, sect), name, _coerce(_decode_opt(val), old, key))
    return cfg
