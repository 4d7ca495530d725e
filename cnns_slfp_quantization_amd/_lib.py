"""ctypes binding of libslfp_hip.so (include/slfp.h).  No torch types cross this boundary:
device pointers are passed as integers, the stream as the raw hipStream_t handle.

There is NO fallback: if the library is missing or a call fails this raises.  The CPU
checker lives in oracle/ and is never imported from here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libslfp_hip.so")

OK, ERR_BAD_ARG, ERR_SHAPE, ERR_UNSUPPORTED, ERR_ALIGNMENT, ERR_HIP = 0, -1, -2, -3, -4, -5
FMT_ACT8, FMT_W8, FMT_SFP7, FMT_EXT = 0, 1, 2, 4
LAYOUT_NCHW, LAYOUT_NHWC = 0, 1
MFMA_DEFAULT, MFMA_F16X1, MFMA_F16X3 = 0, 1, 3

# every symbol include/slfp.h declares (tests check the .so exports exactly these)
SYMBOLS = (
    "slfp_version", "slfp_last_error", "slfp_device_count",
    "slfp_encode_f32", "slfp_decode_f32", "slfp_quantize_f32", "slfp_quantize_layerout_f32", "slfp_absmax_f32",
    "slfp_conv2d_out_shape", "slfp_conv2d_kernel_name", "slfp_conv2d_wprep_bytes",
    "slfp_conv2d_prepare_weights", "slfp_conv2d_prepare_weights_codes", "slfp_conv2d_workspace_bytes", "slfp_conv2d_fwd", "slfp_conv2d_fwd_post",
    "slfp_linear_workspace_bytes", "slfp_linear_fwd", "slfp_linear_prepare_weights", "slfp_linear_fwd_prepared",
    "slfp_nchw_to_nhwc_f32", "slfp_nhwc_to_nchw_f32", "slfp_debug_div_mismatches",
    "slfp_debug_enc_mismatches", "slfp_enc_table_ok", "slfp_dwpw_supported", "slfp_dwpw_fwd",
    "slfp_conv2d_codes_supported", "slfp_conv2d_fwd_codes", "slfp_conv2d_fwd_codes_ws", "slfp_maxpool2d_codes", "slfp_debug_code_mismatches", "slfp_debug_reload_switches",
    "slfp_debug_enc_hl_mismatches",
)


class ConvDesc(ctypes.Structure):
    """struct slfp_conv2d_desc"""
    _fields_ = [
        ("n", ctypes.c_int64), ("c_in", ctypes.c_int64), ("h", ctypes.c_int64), ("w", ctypes.c_int64),
        ("c_out", ctypes.c_int64), ("kh", ctypes.c_int64), ("kw", ctypes.c_int64),
        ("stride_h", ctypes.c_int32), ("stride_w", ctypes.c_int32), ("pad_h", ctypes.c_int32),
        ("pad_w", ctypes.c_int32), ("dil_h", ctypes.c_int32), ("dil_w", ctypes.c_int32),
        ("groups", ctypes.c_int32), ("x_layout", ctypes.c_int32), ("y_layout", ctypes.c_int32),
        ("qbits", ctypes.c_int32), ("ka", ctypes.c_float), ("kw_scale", ctypes.c_float),
        ("mfma_passes", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


class ConvIo(ctypes.Structure):
    """struct slfp_conv2d_io: float32 or 1-byte codes on either side of a layer (slfp_conv2d_fwd_codes)"""
    _fields_ = [("x_codes", ctypes.c_int32), ("y_codes", ctypes.c_int32), ("y_ka", ctypes.c_float), ("y_qbits", ctypes.c_int32)]


class SlfpError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libslfp_hip: {text} (status {code})")
        self.code = code


_lib = None


def load():
    """Load libslfp_hip.so; raises RuntimeError (never falls back) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the SLFP HIP extension is not built. "
            "Run `python -m cnns_slfp_quantization_amd.build` (needs hipcc). There is no CPU fallback.")
    # The library is handed device pointers and streams that PyTorch created, so both must sit on ONE HIP runtime
    # instance: import torch first, so that its bundled libamdhip64 is the one already in the process when the dynamic
    # loader resolves this library's dependency (loading the library first pulls in /opt/rocm's copy, and the first launch
    # then fails with "no ROCm-capable device is detected").
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, ci, cf, i64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_float, ctypes.c_int64
    dp = ctypes.POINTER(ConvDesc)
    sigs = {
        "slfp_version": (ci, []),
        "slfp_last_error": (ctypes.c_char_p, []),
        "slfp_device_count": (ci, []),
        "slfp_encode_f32": (ci, [vp, vp, sz, cf, ci, vp]),
        "slfp_decode_f32": (ci, [vp, vp, sz, ci, vp]),
        "slfp_quantize_f32": (ci, [vp, vp, sz, cf, ci, vp]),
        "slfp_quantize_layerout_f32": (ci, [vp, vp, sz, vp]),
        "slfp_absmax_f32": (ci, [vp, sz, vp, vp]),
        "slfp_conv2d_out_shape": (ci, [dp, ctypes.POINTER(i64), ctypes.POINTER(i64)]),
        "slfp_conv2d_kernel_name": (ctypes.c_char_p, [dp]),
        "slfp_conv2d_wprep_bytes": (sz, [dp]),
        "slfp_conv2d_prepare_weights": (ci, [dp, vp, vp, vp, vp]),
        "slfp_conv2d_prepare_weights_codes": (ci, [dp, vp, vp, vp, vp]),
        "slfp_conv2d_workspace_bytes": (sz, [dp]),
        "slfp_conv2d_fwd": (ci, [dp, vp, vp, vp, vp, vp, vp, vp]),
        "slfp_conv2d_fwd_post": (ci, [dp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp]),
        "slfp_linear_workspace_bytes": (sz, [i64, i64, i64]),
        "slfp_linear_fwd": (ci, [vp, vp, vp, vp, i64, i64, i64, cf, cf, ci, ci, vp, vp]),
        "slfp_linear_prepare_weights": (ci, [vp, vp, i64, i64, cf, ci, ci, vp]),
        "slfp_linear_fwd_prepared": (ci, [vp, vp, vp, vp, i64, i64, i64, cf, cf, ci, ci, vp]),
        "slfp_nchw_to_nhwc_f32": (ci, [vp, vp, i64, i64, i64, i64, vp]),
        "slfp_nhwc_to_nchw_f32": (ci, [vp, vp, i64, i64, i64, i64, vp]),
        "slfp_debug_div_mismatches": (ci, [cf, vp, vp]),
        "slfp_debug_enc_mismatches": (ci, [cf, ci, vp, vp]),
        "slfp_enc_table_ok": (ci, [cf, ci]),
        "slfp_dwpw_supported": (ci, [dp, dp]),
        "slfp_dwpw_fwd": (ci, [dp, dp, vp, vp, vp, vp, ci, vp, vp, vp, vp, ci, vp, vp]),
        "slfp_conv2d_codes_supported": (ci, [dp, ctypes.POINTER(ConvIo), ci, ci]),
        "slfp_conv2d_fwd_codes": (ci, [dp, ctypes.POINTER(ConvIo), vp, vp, vp, vp, vp, ci, vp, vp]),
        "slfp_conv2d_fwd_codes_ws": (ci, [dp, ctypes.POINTER(ConvIo), vp, vp, vp, vp, vp, ci, vp, vp, vp]),
        "slfp_maxpool2d_codes": (ci, [vp, vp, i64, i64, i64, i64, ci, ci, ci, ci, ci, ci, ci, vp]),
        "slfp_debug_code_mismatches": (ci, [cf, ci, vp, vp]),
        "slfp_debug_reload_switches": (None, []),
        "slfp_debug_enc_hl_mismatches": (ci, [cf, ci, vp, vp]),
    }
    assert set(sigs) == set(SYMBOLS)
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def last_error():
    return load().slfp_last_error().decode("utf-8", "replace")


def check(rc):
    """Map a C status to the exception the reference's Python would raise."""
    if rc == OK:
        return
    text = last_error()
    if rc in (ERR_BAD_ARG, ERR_SHAPE):
        # the reference surfaces these from F.conv2d / asserts as RuntimeError/AssertionError
        raise SlfpError(rc, text)
    raise SlfpError(rc, text)
