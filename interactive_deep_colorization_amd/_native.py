"""ctypes binding of ``libideepcolor_hip.so`` (the C ABI in ``include/ideepcolor.h``).

There is NO fallback: if the shared library has not been built, importing the
binding raises, and if no gfx950 device is visible ``idc_create`` fails with
``IDC_ERR_NO_DEVICE``.  Build with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``make -C interactive_deep_colorization_amd/csrc``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libideepcolor_hip.so")

IDC_FP32, IDC_BF16, IDC_BF16X3, IDC_BF16X6, IDC_FP16X3, IDC_FP16 = 0, 1, 2, 3, 4, 5
IDC_FLAG_DIST_HEAD, IDC_FLAG_GLOBAL_HINTS, IDC_FLAG_DIST313, IDC_FLAG_THROUGHPUT_BLOB = 0x1, 0x4, 0x8, 0x10
IDC_OK = 0
STATUS_NAMES = {0: "IDC_OK", -1: "IDC_ERR_INVALID_ARG", -2: "IDC_ERR_NO_DEVICE", -3: "IDC_ERR_HIP",
                -4: "IDC_ERR_NO_WEIGHTS", -5: "IDC_ERR_MISSING_KEY", -6: "IDC_ERR_BATCH",
                -7: "IDC_ERR_UNSUPPORTED", -8: "IDC_ERR_INTERNAL"}

# every symbol include/ideepcolor.h declares (tests check the .so exports all of them)
EXPORTED_SYMBOLS = [
    "idc_version", "idc_set_tile_policy", "idc_set_option", "idc_set_splitk_policy", "idc_device_count", "idc_last_error", "idc_create", "idc_destroy", "idc_set_io_scales",
    "idc_weights_blob_bytes", "idc_pack_weights", "idc_set_weights_host", "idc_set_weights_device",
    "idc_load_weights", "idc_weights_device_ptr", "idc_forward", "idc_forward_device", "idc_forward_dist",
    "idc_lab2rgb", "idc_forward_rgb", "idc_forward_rgb_lazy", "idc_fetch_outputs", "idc_global_histogram", "idc_forward_dist313", "idc_set_dist_temperature", "idc_set_global_hints", "idc_clear_global_hints", "idc_sync", "idc_stream", "idc_num_layers", "idc_layer_info_get", "idc_set_profiling",
    "idc_layer_times_ms", "idc_layer_times_stats", "idc_get_activation", "idc_op_conv2d", "idc_op_deconv4x4s2",
    "idc_set_image_l", "idc_set_hints", "idc_get_hint_planes", "idc_forward_resident",
    "idc_dist_bins", "idc_keep_dist", "idc_dist_at", "idc_get_dist", "idc_suggest_colors",
    "idc_stream_wait", "idc_stream_signal", "idc_alloc_host", "idc_free_host", "idc_forward_async", "idc_wait", "idc_pipeline_times",
    "idc_comm_unique_id", "idc_broadcast_weights", "idc_upsample_lab2rgb",
]
IDC_INTERP_CUBIC, IDC_INTERP_LINEAR, IDC_INTERP_NEAREST = 0, 1, 2
IDC_SRC_OUTPUT_AB, IDC_SRC_OUTPUT_AB_RAW, IDC_SRC_INPUT_AB = 0, 1, 2
IDC_UNIQUE_ID_BYTES = 128
IDC_HINT_AB, IDC_HINT_RGB = 0, 1


class IdcError(RuntimeError):
    def __init__(self, status, message):
        RuntimeError.__init__(self, "%s (%d): %s" % (STATUS_NAMES.get(status, "IDC_ERR"), status, message))
        self.status = status


class TensorDesc(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.POINTER(ctypes.c_float)),
                ("ndim", ctypes.c_int), ("dims", ctypes.c_int64 * 4)]


class Hint(ctypes.Structure):
    """idc_hint: inclusive rectangle + (a, b) or (r, g, b)."""
    _fields_ = [("y0", ctypes.c_int32), ("x0", ctypes.c_int32), ("y1", ctypes.c_int32), ("x1", ctypes.c_int32),
                ("c0", ctypes.c_float), ("c1", ctypes.c_float), ("c2", ctypes.c_float)]


class LayerInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 32), ("kernel", ctypes.c_char * 48), ("flops", ctypes.c_double),
                ("min_bytes", ctypes.c_double), ("launches", ctypes.c_int)]


_lib = None


def load():
    """Load the shared library once; raise loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libideepcolor_hip.so not found at %s -- the HIP extension is not built and there is "
            "no CPU fallback.  Run `python -c \"import __graft_entry__ as g; g.build()\"` "
            "(needs hipcc, cross-compiles for gfx950 without a GPU)." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    c_float_p = ctypes.POINTER(ctypes.c_float)
    vp, ci, cf, csz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

    def proto(name, restype, argtypes):
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes

    proto("idc_version", ci, [])
    proto("idc_set_tile_policy", ci, [ci])
    proto("idc_set_option", ci, [ctypes.c_char_p, ci])
    proto("idc_set_splitk_policy", ci, [ci])
    proto("idc_device_count", ci, [])
    proto("idc_last_error", ctypes.c_char_p, [vp])
    proto("idc_create", ci, [ci, ci, ci, ci, ci, ctypes.c_uint, ctypes.POINTER(vp)])
    proto("idc_destroy", ci, [vp])
    proto("idc_set_io_scales", ci, [vp, cf, cf, cf, cf])
    proto("idc_weights_blob_bytes", csz, [ci, ctypes.c_uint])
    proto("idc_pack_weights", ci, [ci, ctypes.c_uint, ctypes.POINTER(TensorDesc), ci, vp, csz])
    proto("idc_set_weights_host", ci, [vp, vp, csz])
    proto("idc_set_weights_device", ci, [vp, vp, csz, ci])
    proto("idc_load_weights", ci, [vp, ctypes.POINTER(TensorDesc), ci])
    proto("idc_weights_device_ptr", vp, [vp])
    proto("idc_forward", ci, [vp, ci, c_float_p, c_float_p, c_float_p, cf, c_float_p])
    proto("idc_forward_device", ci, [vp, ci, vp, vp, vp, cf, vp, ci])
    proto("idc_forward_dist", ci, [vp, ci, c_float_p, c_float_p, c_float_p, cf, c_float_p, c_float_p])
    proto("idc_global_histogram", ci, [vp, ci, vp, c_float_p, c_float_p, c_float_p])
    proto("idc_lab2rgb", ci, [vp, ci, c_float_p, c_float_p, vp, vp])
    proto("idc_forward_rgb", ci, [vp, ci, c_float_p, c_float_p, c_float_p, cf, cf, c_float_p, vp, vp])
    proto("idc_forward_rgb_lazy", ci, [vp, ci, c_float_p, c_float_p, c_float_p, cf, cf, vp])
    proto("idc_fetch_outputs", ci, [vp, ci, c_float_p, vp])
    proto("idc_forward_dist313", ci, [vp, ci, c_float_p, c_float_p, c_float_p, cf, c_float_p, c_float_p, c_float_p])
    proto("idc_set_dist_temperature", ci, [vp, cf])
    proto("idc_set_global_hints", ci, [vp, ci, c_float_p, c_float_p])
    proto("idc_clear_global_hints", ci, [vp])
    proto("idc_sync", ci, [vp])
    proto("idc_stream", vp, [vp])
    proto("idc_num_layers", ci, [vp])
    proto("idc_layer_info_get", ci, [vp, ci, ctypes.POINTER(LayerInfo)])
    proto("idc_set_profiling", ci, [vp, ci])
    proto("idc_layer_times_ms", ci, [vp, c_float_p, ci])
    proto("idc_layer_times_stats", ci, [vp, c_float_p, c_float_p, c_float_p, ci])
    proto("idc_get_activation", ci, [vp, ctypes.c_char_p, ci, c_float_p, csz,
                                     ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)])
    proto("idc_op_conv2d", ci, [ci, ci, ci, ci, ci, ci, c_float_p, ci, ci, ci, ci, c_float_p, c_float_p, ci,
                                c_float_p, c_float_p, c_float_p, c_float_p])
    proto("idc_op_deconv4x4s2", ci, [ci, ci, ci, ci, ci, ci, c_float_p, ci, c_float_p, c_float_p, ci,
                                     c_float_p, c_float_p])
    proto("idc_set_image_l", ci, [vp, ci, c_float_p])
    proto("idc_set_hints", ci, [vp, ci, ci, ctypes.POINTER(Hint), ci, cf])
    proto("idc_get_hint_planes", ci, [vp, ci, c_float_p, c_float_p])
    proto("idc_forward_resident", ci, [vp, ci, cf, cf, c_float_p, vp, vp])
    proto("idc_dist_bins", ci, [vp])
    proto("idc_keep_dist", ci, [vp, ci])
    proto("idc_dist_at", ci, [vp, ci, ci, ci, c_float_p])
    proto("idc_get_dist", ci, [vp, ci, c_float_p])
    proto("idc_suggest_colors", ci, [vp, ci, ci, ci, ci, ci, ctypes.c_uint, c_float_p, vp, vp, vp])
    proto("idc_stream_wait", ci, [vp, vp])
    proto("idc_stream_signal", ci, [vp, vp])
    proto("idc_alloc_host", vp, [csz])
    proto("idc_free_host", ci, [vp])
    proto("idc_forward_async", ci, [vp, ci, ci, c_float_p, c_float_p, c_float_p, cf, c_float_p])
    proto("idc_wait", ci, [vp, ci])
    proto("idc_pipeline_times", ci, [vp, ci, c_float_p])
    proto("idc_comm_unique_id", ci, [vp])
    proto("idc_broadcast_weights", ci, [vp, vp, ci, ci, ci])
    proto("idc_upsample_lab2rgb", ci, [vp, ci, ci, ci, ci, ci, vp, vp])
    _lib = lib
    return lib


def check(status, handle=None):
    if status != IDC_OK:
        msg = load().idc_last_error(handle)
        raise IdcError(status, msg.decode("utf-8", "replace") if msg else "")
    return status
