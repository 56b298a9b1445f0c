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
    # round 6: every BASELINE config on the line, the parity metrics named, the decode leg pinned
    cf = d["configs"]
    assert "error" not in cf, cf
    assert cf["config3_feature_ddpm"]["shapes_per_s"] > d["value"] and cf["joint_pos_batch_multiple_1"]["shapes_per_s"] > 0
    assert cf["config2_position_ddpm"]["pos_batch_multiple_1"]["batch"] == 256 and cf["config2_position_ddpm"]["pos_batch_multiple_2"]["batch"] == 512
    c4 = cf["config4_five_category_shard"]
    assert c4["finite"] and [e["label"] for e in c4["segments"]] == [0, 2, 3, 4, 6] and sum(e["shapes"] for e in c4["segments"]) == 256
    assert all(v["pos"] <= 1e-3 and v["feat"] <= 1e-3 for v in c4["forward_rel_l2_vs_fp32_mode"].values()), c4
    par = d["parity"]
    assert par["chain_1000_steps_vs_fp32_mode"]["pos"]["shapes_above_1e-3"] == 0 and par["chain_1000_steps_vs_fp32_mode"]["feat"]["shapes_above_1e-3"] == 0
    assert max(v for k, v in par["forward_max_norm_vs_fp32_mode"].items() if k not in ("pos_prec", "feat_prec")) <= 1e-3
    assert d["decode"]["shapes_per_s"] > d["decode"]["shapes_per_s_fp32_mode"] > 0 and "parity_test" in d["decode"]


def test_bench_rccl_path_single_rank(gpu_device):
    d = _run({"SLIDE_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29541", "RANK": "0", "WORLD_SIZE": "1",
              "LOCAL_RANK": "0"}, "--no-roofline", "--no-configs")
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


def test_bench_eight_launcher_processes_on_one_gpu(gpu_device):
    """multi-GPU readiness without the hardware (round 6): `python bench.py --gpus 8` as EIGHT launcher processes
    sharing this box's GPU over gloo (SLIDE_BENCH_SHARE_GPU=1) -- the driver's N = 8 command line, the rank -> shard -> gather-slot
    mapping, the max-over-ranks clock, and the per-rank host enqueue times that show whether eight eager launchers pace on one host
    (reference: pointnet2/distributed.py:47-57,171-182; mesh_evaluation.py:156-186).  A small per-rank batch keeps it short."""
    d = _run({"SLIDE_BENCH_SHARE_GPU": "1"}, "--gpus", "8", "--batch", "32", "--no-roofline", "--no-parity", "--no-decode")
    assert d["n_gpus"] == 8 and d["value"] > 0 and d["config"]["finite"] and d["scaling"] == "weak"
    ds = d["dist"]
    assert ds["rccl_ranks"] == 8 and ds["gather_order_ok"] and ds["gathered_shapes"] == 8 * 32
    assert [r for r, _, _ in ds["rank_devices"]] == list(range(8))
    assert ds["host_enqueue_ms_per_step_max"] > 0
    assert d["config"]["replay"] in ("eager", "graph") and "replay_auto" in d["config"]
