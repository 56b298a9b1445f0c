"""CPU: the C-ABI library builds, loads, and exports every symbol include/*.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared_symbols(sub=""):
    names = []
    inc = os.path.join(REPO, "include", sub)
    for f in sorted(os.listdir(inc)):
        if not f.endswith(".h"):
            continue
        src = open(os.path.join(inc, f)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        src = re.sub(r"(?m)^\s*#.*$", "", src)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", src):
            if m.group(1) not in ("defined", "__attribute__"):
                names.append(m.group(1))
    return sorted(set(names))


def test_library_builds_and_exports_header_symbols():
    from slide_amd import build
    path = build.build()
    lib = ctypes.CDLL(path)
    syms = _declared_symbols()
    assert len(syms) >= 13
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    lib.slide_hip_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.slide_hip_version()


def test_experiments_library_exports_product_and_experiment_symbols():
    """libslide_hip_exp.so (-DSLIDE_EXPERIMENTS: product kernels + the opt-in variants) exports everything the product headers
    declare plus include/experiments/*.h; the PRODUCT library does not carry the experiments' entry points"""
    from slide_amd import build
    build.build(experiments="only")
    lib = ctypes.CDLL(build.LIB_EXP)
    syms = _declared_symbols() + _declared_symbols("experiments")
    assert "slide_resident_run" in syms
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    prod = ctypes.CDLL(build.build())
    assert not hasattr(prod, "slide_resident_run")


def test_ops_fail_loudly_without_gpu():
    import torch
    from slide_amd import _ext
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="CPU not supported"):
        _ext.furthest_point_sampling(torch.zeros(1, 8, 3), 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        _ext.gather_points(torch.zeros(1, 3, 8).transpose(1, 2), torch.zeros(1, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="int tensor"):
        _ext.group_points(torch.zeros(1, 3, 8), torch.zeros(1, 2, 2, dtype=torch.int64))
