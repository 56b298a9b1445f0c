"""ctypes binding of libslide_hip.so (include/slide_hip.h).  There is NO CPU fallback: if the HIP
library is missing or fails to load, every op raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLIDE_HIP_LIB: developer knob for A/B timing of two builds of the same library (tools/ab/ab_build.sh)
LIB_PATH = os.environ.get("SLIDE_HIP_LIB") or os.path.join(_HERE, "libslide_hip.so")
# the EXPERIMENTS build (slide_amd/build.py): the product kernels + every opt-in variant.  SLIDE_EXPERIMENTS=1 makes it the
# library of the process; `with experiments():` switches to it for a block (the plan-variant tests)
LIB_EXP_PATH = os.path.join(_HERE, "libslide_hip_exp.so")
_libs = {}
_use_exp = [os.environ.get("SLIDE_EXPERIMENTS", "0") not in ("", "0")]
ST_EXPERIMENT = -20  # status of an op whose kernel is not in the product library (csrc/engine.hip)

EXPORTS = [
    "gather_points_kernel_wrapper", "gather_points_grad_kernel_wrapper",
    "furthest_point_sampling_kernel_wrapper", "query_ball_point_kernel_wrapper",
    "group_points_kernel_wrapper", "group_points_grad_kernel_wrapper", "three_nn_kernel_wrapper",
    "three_interpolate_kernel_wrapper", "three_interpolate_grad_kernel_wrapper", "slide_knn_points",
    "slide_knn_gather", "slide_gather_rows", "slide_sample_farthest_points", "slide_hip_version", "slide_hip_device_ok",
]


class SlideHipError(RuntimeError):
    pass


def _load(path):
    if path not in _libs:
        if not os.path.exists(path):
            raise SlideHipError(
                "%s is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python slide_amd/build.py [--experiments]`; there is no CPU fallback." % (os.path.basename(path), path))
        h = ctypes.CDLL(path)
        h.slide_hip_version.restype = ctypes.c_char_p
        _libs[path] = h
    return _libs[path]


def lib():
    return _load(LIB_EXP_PATH if _use_exp[0] else LIB_PATH)


def have_experiments():
    return os.path.exists(LIB_EXP_PATH)


class experiments:
    """context manager: every library call inside the block goes to the experiments build (both libraries can be loaded in one
    process; device memory is shared, kernels and their static launch state are per library)"""

    def __enter__(self):
        self._prev = _use_exp[0]
        _load(LIB_EXP_PATH)
        _use_exp[0] = True
        return self

    def __exit__(self, *exc):
        _use_exp[0] = self._prev
        return False


def check(status, what):
    if status == ST_EXPERIMENT:
        raise SlideHipError("%s: the plan asks for a kernel that only exists in the experiments build of the library "
                            "(an opt-in SLIDE_* knob is set): build it with `python slide_amd/build.py --experiments` and run with "
                            "SLIDE_EXPERIMENTS=1" % what)
    if status != 0:
        raise SlideHipError("%s failed with HIP status %d" % (what, status))


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def stream_of(t=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
