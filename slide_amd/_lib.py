"""ctypes binding of libslide_hip.so (include/slide_hip.h).  There is NO CPU fallback: if the HIP
library is missing or fails to load, every op raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLIDE_HIP_LIB: developer knob for A/B timing of two builds of the same library (tools/ab_build.sh)
LIB_PATH = os.environ.get("SLIDE_HIP_LIB") or os.path.join(_HERE, "libslide_hip.so")
_lib = None

EXPORTS = [
    "gather_points_kernel_wrapper", "gather_points_grad_kernel_wrapper",
    "furthest_point_sampling_kernel_wrapper", "query_ball_point_kernel_wrapper",
    "group_points_kernel_wrapper", "group_points_grad_kernel_wrapper", "three_nn_kernel_wrapper",
    "three_interpolate_kernel_wrapper", "three_interpolate_grad_kernel_wrapper", "slide_knn_points",
    "slide_knn_gather", "slide_sample_farthest_points", "slide_hip_version", "slide_hip_device_ok",
]


class SlideHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SlideHipError(
                "libslide_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python slide_amd/build.py`; there is no CPU fallback." % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.slide_hip_version.restype = ctypes.c_char_p
    return _lib


def check(status, what):
    if status != 0:
        raise SlideHipError("%s failed with HIP status %d" % (what, status))


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def stream_of(t=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
