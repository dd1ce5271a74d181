"""ctypes binding of libmpgan_hip.so (C ABI declared in include/mpgan.h).

There is no CPU fallback: if the library is missing or no gfx950 device is
present, every compute entry point raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmpgan_hip.so")
# development only (tools/probe_variants.py): time another build of the same library; never a fallback
if os.environ.get("MPGAN_LIB_OVERRIDE"):
    LIB_PATH = os.path.abspath(os.environ["MPGAN_LIB_OVERRIDE"])

MPG_OK = 0
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
PREC_F16X1, PREC_F16F6, PREC_F16X3 = 1, 2, 3
G8_F16 = 0
MAX_SEG = 4

_ACT_IDS = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "lrelu": ACT_LRELU, "tanh": ACT_TANH}


def act_id(name):
    try:
        return _ACT_IDS[name]
    except KeyError:
        raise ValueError("unknown activation %r" % (name,))


class MpgError(RuntimeError):
    pass


class ConvSeg(ctypes.Structure):
    _fields_ = [
        ("x", ctypes.c_void_p),
        ("wpack", ctypes.c_void_p),
        ("cin", ctypes.c_int32),
        ("cgroups", ctypes.c_int32),
        ("g_off", ctypes.c_int32),
        ("kh", ctypes.c_int32),
        ("kw", ctypes.c_int32),
        ("up_log2", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
        ("pad_hi", ctypes.c_int32),
    ]


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int32),
        ("h", ctypes.c_int32),
        ("w", ctypes.c_int32),
        ("cout", ctypes.c_int32),
        ("nseg", ctypes.c_int32),
        ("seg", ConvSeg * MAX_SEG),
        ("bias", ctypes.c_void_p),
        ("act", ctypes.c_int32),
        ("leak", ctypes.c_float),
        ("pixel_norm", ctypes.c_int32),
        ("pn_eps", ctypes.c_float),
        ("post_add", ctypes.c_void_p),
        ("post_add_stride", ctypes.c_int32),
        ("post_add_coff", ctypes.c_int32),
        ("y", ctypes.c_void_p),
        ("y_g8", ctypes.c_void_p),
        ("prec", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("in_amax", ctypes.c_void_p),
    ]


class SmallPairDesc(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int32), ("h", ctypes.c_int32), ("w", ctypes.c_int32),
        ("x", ctypes.c_void_p),
        ("cin", ctypes.c_int32), ("cgroups", ctypes.c_int32), ("g_off", ctypes.c_int32), ("up_log2", ctypes.c_int32),
        ("wpack_a", ctypes.c_void_p),
        ("kh_a", ctypes.c_int32), ("kw_a", ctypes.c_int32), ("cmid", ctypes.c_int32),
        ("bias_a", ctypes.c_void_p),
        ("act_a", ctypes.c_int32), ("leak_a", ctypes.c_float),
        ("wpack_b", ctypes.c_void_p),
        ("kh_b", ctypes.c_int32), ("kw_b", ctypes.c_int32),
        ("wpack_s", ctypes.c_void_p),
        ("kh_s", ctypes.c_int32), ("kw_s", ctypes.c_int32),
        ("bias_b", ctypes.c_void_p),
        ("act_b", ctypes.c_int32), ("leak_b", ctypes.c_float), ("cout", ctypes.c_int32),
        ("y", ctypes.c_void_p),
        ("y_g8", ctypes.c_void_p),
        ("prec", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


_P = ctypes.c_void_p
_I = ctypes.c_int
_Z = ctypes.c_size_t
_F = ctypes.c_float

# name -> (restype, argtypes); mirrors include/mpgan.h one to one
PROTOTYPES = {
    "mpg_last_error": (ctypes.c_char_p, []),
    "mpg_version": (ctypes.c_char_p, []),
    "mpg_device_info": (_I, [ctypes.POINTER(_I), ctypes.c_char_p, _I]),
    "mpg_g8_bytes": (_Z, [_I, _I, _I, _I]),
    "mpg_f32_to_g8": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mpg_f32_to_g8_scaled": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "mpg_absmax": (_I, [_P, _P, _Z, _P]),
    "mpg_g8_to_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "mpg_conv_pack_size": (_Z, [_I, _I, _I, _I, _I]),
    "mpg_conv_pack_weights": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _I, _P, _Z]),
    "mpg_conv2d_fused": (_I, [_P, ctypes.POINTER(ConvDesc)]),
    "mpg_conv2d_small_pair": (_I, [_P, ctypes.POINTER(SmallPairDesc)]),
    "mpg_conv2d_direct": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _F, _P, _P, _I, _F, _P]),
    "mpg_resize_nearest": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I]),
    "mpg_resize_bilinear": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I]),
    "mpg_resize_bicubic": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I]),
    "mpg_avg_pool2": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "mpg_max_pool": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "mpg_max_pool_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mpg_pixel_norm": (_I, [_P, _P, _Z, _I, _F, _P]),
    "mpg_minibatch_stddev": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "mpg_minibatch_stddev_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "mpg_add_act": (_I, [_P, _P, _P, _Z, _I, _F, _P]),
    "mpg_axis_zoom_linear": (_I, [_P, _P, _Z, _I, _Z, _P, _I]),
    "mpg_volume_transpose": (_I, [_P, _P, _I, _I, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I), _F, _P]),
    "mpg_add_adjacent": (_I, [_P, _P, _I, _Z, _I, _I, _I, _P]),
    "mpg_cutoff": (_I, [_P, _P, _Z, _F, _P]),
    "mpg_channel_gather": (_I, [_P, _P, _I, _P, _I, _Z, _P, _P, _P, _I, _P]),
    "mpg_conv2d_transpose": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _F, _P, _I, _F, _P]),
    "mpg_depth_to_space": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    # training step
    "mpg_conv2d_wgrad": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _F, _P]),
    "mpg_conv2d_wgrad_mfma_ws_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "mpg_conv2d_wgrad_mfma": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _F, _I, _P, _Z, _P, _P, _P]),
    "mpg_conv2d_wgrad_g8": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _F, _I, _P, _P, _P]),
    "mpg_conv2d_dgrad": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _F, _P]),
    "mpg_fc_forward": (_I, [_P, _P, _I, _I, _P, _I, _F, _P, _I, _F, _P]),
    "mpg_channel_sum": (_I, [_P, _P, _Z, _I, _P]),
    "mpg_channel_sum_ordered": (_I, [_P, _P, _Z, _I, _P, _P, _Z]),
    "mpg_bn_train_fwd": (_I, [_P, _P, _Z, _I, _P, _P, _F, _I, _F, _P, _P, _P, _P, _P, _F]),
    "mpg_bn_partials_floats": (_Z, [_I]),
    "mpg_bn_train_fwd_ordered": (_I, [_P, _P, _Z, _I, _P, _P, _F, _I, _F, _P, _P, _P, _P, _P, _F, _P, _Z]),
    "mpg_bn_train_bwd": (_I, [_P, _P, _P, _Z, _I, _P, _P, _P, _F, _P, _P, _P, _P]),
    "mpg_bn_train_bwd_ordered": (_I, [_P, _P, _P, _Z, _I, _P, _P, _P, _F, _P, _P, _P, _P, _P, _Z]),
    "mpg_act_bwd": (_I, [_P, _P, _P, _Z, _I, _F, _P, _P]),
    "mpg_pixel_norm_bwd": (_I, [_P, _P, _P, _Z, _I, _F, _P]),
    "mpg_resize_nearest_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I]),
    "mpg_avg_pool2_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "mpg_lerp": (_I, [_P, _P, _P, _Z, _F, _P]),
    "mpg_tensor_resample": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mpg_tensor_resample_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mpg_advect_velocity": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "mpg_semi_lagrange": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "mpg_semi_lagrange_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "mpg_maccormack": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P]),
    "mpg_tile_gather": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P]),
    "mpg_resample_affine": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, ctypes.POINTER(ctypes.c_double),
                                ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float)]),
    "mpg_tile_orient": (_I, [_P, _P, _I, _I, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_I),
                            ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(ctypes.c_float), _P]),
    "mpg_semilagr_positions": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "mpg_pair_reduce": (_I, [_P, _P, _P, _Z, _I, _P]),
    "mpg_adam_step": (_I, [_P, _P, _P, _P, _P, _Z, _P, _F, _F, _F]),
    "mpg_adam_step_staged": (_I, [_P, _P, _P, _P, _P, _P, _Z, _P, _P, _I, _I, _F, _F, _F, _F, _F, _P, _F]),
}

_lib = None


def load():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MpgError(
            "HIP extension %s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != MPG_OK:
        raise MpgError("%s failed (%d): %s" % (what, rc, load().mpg_last_error().decode()))


def device_info():
    lib = load()
    cu = _I(0)
    name = ctypes.create_string_buffer(64)
    check(lib.mpg_device_info(ctypes.byref(cu), name, 64), "mpg_device_info")
    return cu.value, name.value.decode()
