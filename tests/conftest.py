import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")
    config.addinivalue_line("markers", "exp: exercises the EXPERIMENTS build (libslide_hip_exp.so: opt-in kernel variants that lost their A/B) -- "
                                       "not part of the default selections; run with -m exp (or -m 'gpu and exp')")


def pytest_collection_modifyitems(config, items):
    """tests of the experiments library carry the `exp` marker (explicitly, or by using the `exp_lib` fixture) and are DESELECTED unless
    the -m expression names it: the default `-m gpu` / `-m "not gpu"` runs spend their time on the product paths"""
    for it in items:
        if "exp_lib" in getattr(it, "fixturenames", ()):
            it.add_marker(pytest.mark.exp)
    if "exp" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("exp") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_spec(g, prefix="spec"):
    return [(str(n), tuple(int(x) for x in str(s).split(",")) if str(s) else ())
            for n, s in zip(g[prefix + "_names"], g[prefix + "_shapes"])]


class NoiseStream:
    """the seeded standard-normal stream tools/gen_golden.py injected into the reference samplers"""

    def __init__(self, seed):
        self.rs = np.random.RandomState(int(seed))
        self.count = 0

    def __call__(self, size):
        self.count += 1
        return self.rs.standard_normal(tuple(size)).astype(np.float32)


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def exp_lib():
    """the EXPERIMENTS build of the HIP library (slide_amd/build.py --experiments) for the duration of a test: the opt-in plan
    variants (round-2 plan, X-stationary tiles, per-point layer chains, wide tails, head + update launch, resident kernel ...) are
    not in the product library"""
    from slide_amd import _lib
    if not _lib.have_experiments():
        pytest.skip("libslide_hip_exp.so is not built (python slide_amd/build.py --experiments)")
    with _lib.experiments():
        yield
