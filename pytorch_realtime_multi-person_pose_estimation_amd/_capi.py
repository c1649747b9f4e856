"""ctypes binding of librtpose_mi355x.so (the C ABI in include/rtpose_mi355x.h).

This is the only place the package touches native code.  The library is built
in-tree (``csrc/Makefile`` -> ``lib/librtpose_mi355x.so``) and loaded from
there; if it is missing the import fails loudly — there is no Python/CPU
fallback for the hot path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTPOSE_LIB_PATH: developer override to A/B experimental builds of the same library
LIB_PATH = os.environ.get("RTPOSE_LIB_PATH") or os.path.join(_HERE, "lib", "librtpose_mi355x.so")

NUM_PART = 18
NUM_LIMB = 19
DTYPE_F32, DTYPE_BF16, DTYPE_BF16X3 = 0, 1, 2
WINO_DEFAULT, WINO7_AUTO, WINO3_AUTO = -1, 1, 3
NMS_NO_REFINE, NMS_GAUSSIAN = 1, 2


class RtposeError(RuntimeError):
    pass


class Layout(C.Structure):
    """rtpose_layout: shared-gap padded NHWC addressing (header §1)."""
    _fields_ = [("cstride", C.c_int32), ("choff", C.c_int32), ("ws", C.c_int32),
                ("hs", C.c_int32), ("lead", C.c_int32)]

    @classmethod
    def dense(cls, c, h, w, choff=0):
        return cls(c, choff, w, h, 0)

    @classmethod
    def padded(cls, c, h, w, pad, choff=0):
        ws = w + pad
        return cls(c, choff, ws, h + pad, pad * ws + pad)


class ConvDesc(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("w_packed", C.c_void_p), ("bias_packed", C.c_void_p),
                ("out", C.c_void_p), ("lin", Layout), ("lout", Layout), ("cin", C.c_int32),
                ("cout", C.c_int32), ("k", C.c_int32), ("relu", C.c_int32), ("pool", C.c_int32),
                ("out_cmap", C.c_void_p), ("wino_m", C.c_int32), ("in_plane_pixels", C.c_int32),
                ("out_plane_pixels", C.c_int32)]


class NetOptions(C.Structure):
    """rtpose_net_options: per-plan arithmetic of the fp32 convs (header §3)."""
    _fields_ = [("struct_bytes", C.c_uint32), ("dtype", C.c_int32), ("winograd3", C.c_int32),
                ("winograd7", C.c_int32), ("amp_limit", C.c_float)]

    @classmethod
    def make(cls, dtype=0, winograd3=-1, winograd7=-1, amp_limit=0.0):
        return cls(C.sizeof(cls), dtype, winograd3, winograd7, amp_limit)


class PwDesc(C.Structure):
    """rtpose_pw_desc: one fused pointwise chain (csrc/pw_fused.hip)."""
    _fields_ = [("inp", C.c_void_p), ("dw_w", C.c_void_p), ("dw_b", C.c_void_p), ("w_packed", C.c_void_p),
                ("bias_packed", C.c_void_p), ("out", C.c_void_p), ("lin", Layout), ("lout", Layout),
                ("cin", C.c_int32), ("cout", C.c_int32), ("coutp", C.c_int32), ("relu", C.c_int32),
                ("out_cmap", C.c_void_p), ("pt_src", C.c_void_p), ("lpt", Layout), ("pt_cmap", C.c_void_p),
                ("pt_c", C.c_int32), ("pt_pairs", C.c_int32), ("pt_a", C.c_int32), ("pt_b", C.c_int32),
                ("pt_split", C.c_int32), ("pt_d0", C.c_int32), ("pt_d1", C.c_int32), ("in_planes", C.c_void_p)]


class PrepImage(C.Structure):
    """rtpose_prep_image: one image of a rtpose_preprocess_u8_batch launch."""
    _fields_ = [("img_bgr", C.c_void_p), ("im_scale", C.c_double), ("h0", C.c_int32), ("w0", C.c_int32),
                ("hr", C.c_int32), ("wr", C.c_int32), ("flip", C.c_int32), ("n_index", C.c_int32)]


class DecodeCfg(C.Structure):
    _fields_ = [("num_keypoints", C.c_int32), ("upsample", C.c_int32),
                ("thresh_heatmap", C.c_float), ("max_peaks_per_part", C.c_int32),
                ("max_humans", C.c_int32)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "librtpose_mi355x.so not found at %s — build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C %s`; "
            "this package has no CPU fallback" % (LIB_PATH, os.path.join(_HERE, "csrc")))
    return C.CDLL(LIB_PATH)


lib = _load()

_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
_LP = C.POINTER(Layout)

_SIGS = {
    "rtpose_version": (C.c_char_p, []),
    "rtpose_last_error": (C.c_char_p, []),
    "rtpose_layout_pixels": (_sz, [_LP, _i, _i, _i]),
    "rtpose_packed_weight_floats": (_sz, [_i, _i, _i]),
    "rtpose_packed_bias_floats": (_sz, [_i]),
    "rtpose_pack_conv_weights": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "rtpose_conv2d": (_i, [C.POINTER(ConvDesc), _i, _i, _i, _i, _vp]),
    "rtpose_conv1x1_pair_fits": (_i, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _i]),
    "rtpose_conv1x1_pair": (_i, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _i, _i, _i, _i, _vp]),
    "rtpose_conv1x1_pair_bf16_fits": (_i, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _i]),
    "rtpose_conv1x1_pair_bf16": (_i, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _i, _i, _i, _i, _i, _vp]),
    "rtpose_conv3x3_c64_bf16_fits": (_i, [C.POINTER(ConvDesc), _i, _i, _i, _i]),
    "rtpose_conv3x3_c64_bf16": (_i, [C.POINTER(ConvDesc), _i, _i, _i, _vp]),
    "rtpose_conv_first_packed_floats": (_sz, []),
    "rtpose_pack_conv_first": (_i, [_vp, _vp, _vp, _vp]),
    "rtpose_conv_first": (_i, [_vp, _vp, _LP, _vp, _vp, _LP, _i, _i, _i, _i, _vp]),
    "rtpose_conv_first_planes": (_i, [_vp, _vp, _LP, _vp, _vp, _LP, _i, _i, _i, _i, _i, _vp]),
    "rtpose_pack_conv_first_bf16": (_i, [_vp, _vp, _vp, _vp]),
    "rtpose_conv_first_bf16": (_i, [_vp, _vp, _LP, _vp, _vp, _LP, _i, _i, _i, _i, _vp]),
    "rtpose_conv2d_winograd_fits": (_i, [C.POINTER(ConvDesc), _i, _i, _i]),
    "rtpose_packed_weight_floats_winograd": (_sz, [_i, _i, _i]),
    "rtpose_pack_conv_weights_winograd": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "rtpose_conv2d_winograd": (_i, [C.POINTER(ConvDesc), _i, _i, _i, _i, _vp]),
    "rtpose_packed_weight_floats_winograd3": (_sz, [_i, _i, _i]),
    "rtpose_pack_conv_weights_winograd3": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "rtpose_packed_weight_floats_winograd7": (_sz, [_i, _i, _i]),
    "rtpose_pack_conv_weights_winograd7": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "rtpose_conv2d_winograd_scratch_bytes": (_sz, []),
    "rtpose_conv2d_winograd_ex": (_i, [C.POINTER(ConvDesc), _i, _i, _i, _i, _vp, _sz, _vp]),
    "rtpose_conv2d_winograd_scratch_error": (_i, [_vp, C.POINTER(_i), _vp]),
    "rtpose_winograd_amplification": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "rtpose_packed_pw_floats": (_sz, [_i, _i]),
    "rtpose_pack_pw_weights": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "rtpose_pw_fused": (_i, [C.POINTER(PwDesc), _i, _i, _i, _vp]),
    "rtpose_pack_pw_weights_cols": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "rtpose_pw_head_fits": (_i, [C.POINTER(PwDesc), C.POINTER(PwDesc)]),
    "rtpose_pw_head": (_i, [C.POINTER(PwDesc), C.POINTER(PwDesc), _i, _i, _i, _vp]),
    "rtpose_packed_pw_bytes_bf16": (_sz, [_i, _i]),
    "rtpose_pack_pw_weights_bf16": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "rtpose_pw_fused_bf16": (_i, [C.POINTER(PwDesc), _i, _i, _i, _i, _vp]),
    "rtpose_unit_bf16_fits": (_i, [C.POINTER(PwDesc), C.POINTER(PwDesc), _i, _i]),
    "rtpose_unit_bf16": (_i, [C.POINTER(PwDesc), C.POINTER(PwDesc), _i, _i, _i, _vp]),
    "rtpose_pw_head_bf16_fits": (_i, [C.POINTER(PwDesc), C.POINTER(PwDesc)]),
    "rtpose_pw_head_bf16": (_i, [C.POINTER(PwDesc), C.POINTER(PwDesc), _i, _i, _i, _vp]),
    "rtpose_pack_pw_head2_bf16": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "rtpose_maxpool2x2": (_i, [_vp, _LP, _vp, _LP, _i, _i, _i, _i, _vp]),
    "rtpose_nchw_to_layout": (_i, [_vp, _vp, _LP, _i, _i, _i, _i, _i, _vp]),
    "rtpose_layout_to_nchw": (_i, [_vp, _LP, _vp, _i, _i, _i, _i, _vp]),
    "rtpose_layout_copy": (_i, [_vp, _LP, _vp, _LP, _i, _i, _i, _i, _vp]),
    "rtpose_nchw_to_layout_affine": (_i, [_vp, _vp, _LP, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "rtpose_stem_conv3x3_s2": (_i, [_vp, _LP, _vp, _vp, _vp, _LP, _i, _i, _i, _i, _i, _i, _vp]),
    "rtpose_stem_conv3x3_s2_nchw": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _LP, _i, _i, _i, _i, _i, _vp]),
    "rtpose_stem_conv3x3_s2_nchw_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _LP, _i, _i, _i, _i, _i, _i, _vp]),
    "rtpose_stem_pool_nchw": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _LP, _i, _i, _i, _i, _i, _vp]),
    "rtpose_maxpool3x3s2_ceil_bf16": (_i, [_vp, _LP, _vp, _LP, _i, _i, _i, _i, _vp]),
    "rtpose_dwconv3x3_bf16": (_i, [_vp, _LP, _vp, _vp, _vp, _LP, _i, _i, _i, _i, _i, _vp]),
    "rtpose_layout_copy_cmap_bf16": (_i, [_vp, _LP, _vp, _LP, _i, _vp, _i, _i, _i, _vp]),
    "rtpose_shufflenet_create_ex": (_i, [_i, _i, _i, _i, C.POINTER(_vp)]),
    "rtpose_maxpool3x3s2_ceil": (_i, [_vp, _LP, _vp, _LP, _i, _i, _i, _i, _vp]),
    "rtpose_dwconv3x3": (_i, [_vp, _LP, _vp, _vp, _vp, _LP, _i, _i, _i, _i, _i, _vp]),
    "rtpose_layout_copy_cmap": (_i, [_vp, _LP, _vp, _LP, _i, _vp, _i, _i, _i, _vp]),
    "rtpose_layout_axpby": (_i, [_vp, _LP, _vp, _i, _i, _i, _i, C.c_float, C.c_float, _vp]),
    "rtpose_net_create": (_i, [_i, _i, _i, C.POINTER(_vp)]),
    "rtpose_net_create_ex": (_i, [_i, _i, _i, _i, C.POINTER(_vp)]),
    "rtpose_net_create_opts": (_i, [_i, _i, _i, C.POINTER(NetOptions), C.POINTER(_vp)]),
    "rtpose_net_finalize_weights": (_i, [_vp, _vp]),
    "rtpose_net_conv_numerics": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(C.c_float), _vp]),
    "rtpose_net_device_status": (_i, [_vp, C.POINTER(_i), _vp]),
    "rtpose_net_device_status_async": (_i, [_vp, _vp, _vp]),
    "rtpose_net_set_output_guard": (_i, [_vp, _vp]),
    "rtpose_net_output_guard_launch": (_i, [_vp]),
    "rtpose_net_set_persistent7": (_i, [_vp, _i]),
    "rtpose_net_persistent7": (_i, [_vp]),
    "rtpose_net_graph_active": (_i, [_vp]),
    "rtpose_net_dtype": (_i, [_vp]),
    "rtpose_packed_weight_bytes_bf16": (_sz, [_i, _i, _i]),
    "rtpose_pack_conv_weights_bf16": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "rtpose_conv2d_bf16": (_i, [C.POINTER(ConvDesc), _i, _i, _i, _i, _i, _vp]),
    "rtpose_nchw_to_layout_bf16": (_i, [_vp, _vp, _LP, _i, _i, _i, _i, _i, _vp]),
    "rtpose_layout_f32_to_bf16": (_i, [_vp, _LP, _vp, _LP, _i, _i, _i, _i, _i, _vp]),
    "rtpose_layout_bf16_to_f32": (_i, [_vp, _LP, _vp, _LP, _i, _i, _i, _i, _vp]),
    "rtpose_packed_weight_bytes_bf16x3": (_sz, [_i, _i, _i]),
    "rtpose_pack_conv_weights_bf16x3": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "rtpose_conv2d_bf16x3": (_i, [C.POINTER(ConvDesc), _i, _i, _i, _i, _i, _vp]),
    "rtpose_nchw_to_layout_split": (_i, [_vp, _vp, _LP, _i, _i, _i, _i, _i, _vp]),
    "rtpose_layout_f32_to_split": (_i, [_vp, _LP, _vp, _LP, _i, _i, _i, _i, _i, _vp]),
    "rtpose_layout_split_to_f32": (_i, [_vp, _LP, _vp, _LP, _i, _i, _i, _i, _vp]),
    "rtpose_net_destroy": (None, [_vp]),
    "rtpose_net_workspace_bytes": (_sz, [_vp]),
    "rtpose_net_weight_bytes": (_sz, [_vp]),
    "rtpose_net_bind": (_i, [_vp, _vp, _sz, _vp, _sz, _i, _vp]),
    "rtpose_net_num_convs": (_i, [_vp]),
    "rtpose_net_conv_info": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "rtpose_net_load_conv": (_i, [_vp, _i, _vp, _vp, _vp]),
    "rtpose_net_forward": (_i, [_vp, _vp, _vp]),
    "rtpose_net_set_keep_intermediates": (_i, [_vp, _i]),
    "rtpose_net_read_output": (_i, [_vp, _i, _vp, _vp]),
    "rtpose_net_output_view": (_i, [_vp, _i, C.POINTER(_vp), _LP, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "rtpose_net_set_profiling": (_i, [_vp, _i]),
    "rtpose_net_num_launches": (_i, [_vp]),
    "rtpose_net_launch_info": (_i, [_vp, _i, C.POINTER(C.c_float), C.POINTER(_i), C.POINTER(C.c_double), C.c_char_p, _i]),
    "rtpose_net_launch_executed_flops": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(_i)]),
    "rtpose_shufflenet_create": (_i, [_i, _i, _i, C.POINTER(_vp)]),
    "rtpose_shufflenet_destroy": (None, [_vp]),
    "rtpose_shufflenet_workspace_bytes": (_sz, [_vp]),
    "rtpose_shufflenet_weight_bytes": (_sz, [_vp]),
    "rtpose_shufflenet_bind": (_i, [_vp, _vp, _sz, _vp, _sz, _i, _vp]),
    "rtpose_shufflenet_num_layers": (_i, [_vp]),
    "rtpose_shufflenet_layer_info": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "rtpose_shufflenet_load": (_i, [_vp, _i, _vp, _vp, _vp]),
    "rtpose_shufflenet_forward": (_i, [_vp, _vp, _vp]),
    "rtpose_shufflenet_read_output": (_i, [_vp, _i, _vp, _vp]),
    "rtpose_shufflenet_output_view": (_i, [_vp, _i, C.POINTER(_vp), _LP, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "rtpose_shufflenet_set_profiling": (_i, [_vp, _i]),
    "rtpose_shufflenet_num_launches": (_i, [_vp]),
    "rtpose_shufflenet_launch_info": (_i, [_vp, _i, C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_char_p, _i]),
    "rtpose_decode_workspace_bytes": (_sz, [C.POINTER(DecodeCfg), _i]),
    "rtpose_decode_result_bytes": (_sz, [C.POINTER(DecodeCfg), _i]),
    "rtpose_decode_batch": (_i, [_vp, _LP, _vp, _LP, _i, _i, _i, C.POINTER(DecodeCfg), _vp, _sz, _vp, _vp]),
    "rtpose_nms_batch": (_i, [_vp, _LP, _i, _i, _i, C.POINTER(DecodeCfg), _vp, _vp]),
    "rtpose_nms_batch_ex": (_i, [_vp, _LP, _i, _i, _i, C.POINTER(DecodeCfg), _i, _vp, _vp]),
    "rtpose_decode_batch_ex": (_i, [_vp, _LP, _vp, _LP, _i, _i, _i, C.POINTER(DecodeCfg), _i, _vp, _sz, _vp, _vp]),
    "rtpose_gaussian_kernel1d": (_i, [C.POINTER(C.c_double), _i]),
    "rtpose_preprocess_u8_batch": (_i, [C.POINTER(PrepImage), _i, _i, _vp, _LP, _i, _i, _vp]),
    "rtpose_preprocess_u8": (_i, [_vp, _i, _i, C.c_double, _i, _vp, _LP, _i, _i, _i, _i, _i, _vp]),
    "rtpose_preprocess_u8_flip": (_i, [_vp, _i, _i, C.c_double, _i, _vp, _LP, _i, _i, _i, _i, _i, _i, _vp]),
    "rtpose_tta_accumulate": (_i, [_vp, _LP, _vp, _LP, _i, _i, _i, _vp, _vp, _i, _i, C.c_float, C.c_float,
                                   C.c_float, C.c_float, _i, _vp]),
    "rtpose_net_input_view": (_i, [_vp, C.POINTER(_vp), _LP]),
    "rtpose_net_forward_prepared": (_i, [_vp, _vp]),
    "rtpose_resize_bilinear_accum": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, C.c_float, C.c_float, C.c_float, C.c_float, _vp]),
    "rtpose_flip_merge": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    # legacy SWIG-module names (lib/pafprocess/pafprocess.h:53-59)
    "process_paf": (_i, [_i, _i, _i, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "get_num_humans": (_i, []),
    "get_part_cid": (_i, [_i, _i]),
    "get_score": (C.c_float, [_i]),
    "get_part_x": (_i, [_i]),
    "get_part_y": (_i, [_i]),
    "get_part_score": (C.c_float, [_i]),
}

EXPORTED = tuple(_SIGS)

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here == header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    return lib.rtpose_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    if rc != 0:
        raise RtposeError("%s failed (rc=%d): %s" % (what or "librtpose_mi355x call", rc, last_error()))


def ptr(t):
    """Device/host address of a torch tensor or numpy array as c_void_p."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
