"""bench.py's output contract on a GPU box: exactly ONE line on stdout, a JSON object with the driver's keys plus
`roofline`; the RCCL path (process group, barrier, all-gather, max-over-ranks) is taken in a single-rank run with
SLIDE_FORCE_DIST=1 -- RCCL writes its banner to the C stdout, which must not reach the JSON stream."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config"}


def _run(extra_env, *flags):
    env = dict(os.environ, **extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "3", "--no-cpu-baseline"] +
                       list(flags), cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().split("\n") if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_single_json_line(gpu_device):
    d = _run({})
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 30 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["finite"] and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0 < rf["frac"] < 1 and rf["kernel"].split(" ")[0] in \
        "".join(rf["mfma_kernels"])
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3


def test_bench_rccl_path_single_rank(gpu_device):
    d = _run({"SLIDE_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29541", "RANK": "0", "WORLD_SIZE": "1",
              "LOCAL_RANK": "0"}, "--no-roofline")
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["finite"]


def test_bench_five_category_workload(gpu_device):
    """BASELINE configs[3]: the rank's 256 shapes are five per-category segments, each its own weight set and chain pair;
    the fp32-mode leg of the `parity` object runs beside it (it once borrowed a segment-sized label vector)"""
    d = _run({}, "--workload", "five-cat", "--no-roofline", "--fp32-steps", "4")
    assert KEYS <= set(d) and d["value"] > 0 and d["config"]["finite"]
    seg = d["config"]["segments_rank0"]
    assert [e["label"] for e in seg] == [0, 2, 3, 4, 6] and sum(e["shapes"] for e in seg) == 256
    assert d["parity"]["fp32_mode_shapes_per_s"] > 0


def test_bench_self_launch_two_ranks_on_one_gpu(gpu_device):
    """`python bench.py --gpus 2` with no WORLD_SIZE re-executes itself under torch.distributed.run (what the driver's N > 1
    command line relies on); both ranks share the box's single GPU over gloo here"""
    d = _run({"SLIDE_BENCH_SHARE_GPU": "1"}, "--gpus", "2", "--no-roofline", "--no-parity")
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["finite"] and d["scaling"] == "weak"
