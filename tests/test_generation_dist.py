"""CPU: the N>1 path (contiguous per-rank shards, no data-path collective, one all-gather) on gloo with world_size 2,
plus the shard arithmetic against the reference's ceil-division rule."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from slide_amd.generation import all_gather_rows, batches, generate_latents, save_generated, shard_range


def test_shard_range_matches_reference_rule():
    # npz_dataset.py:90-96: per = ceil(n / world); rank r takes [r*per, (r+1)*per) clipped by slicing
    for n in (1, 7, 8, 9, 2048, 2049):
        for w in (1, 2, 3, 8):
            per = int(np.ceil(n / w))
            got = [shard_range(n, r, w) for r in range(w)]
            want = [(min(r * per, n), min((r + 1) * per, n)) for r in range(w)] if w > 1 else [(0, n)]
            assert got == want
            assert sum(e - s for s, e in got) == n
    assert list(batches(3, 10, 4)) == [(3, 7), (7, 10)]
    assert list(batches(5, 5, 4)) == []


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    labels = np.arange(n) % 13

    def run_batch(lab, lo, hi):  # stands in for the HIP samplers: value encodes the global sample index and label
        idx = torch.arange(lo, hi, dtype=torch.float32)
        return (idx[:, None, None] * 100 + torch.as_tensor(lab, dtype=torch.float32)[:, None, None]).expand(hi - lo, 16, 51).clone()

    calls = []
    inner = run_batch
    run_batch = lambda lab, lo, hi: (calls.append((lo, hi)), inner(lab, lo, hi))[1]
    full, timing = generate_latents(n, 3, labels, run_batch, rank, world)
    s0, e0 = shard_range(n, rank, world)
    assert calls == list(batches(s0, e0, 3))  # a rank with an empty slice runs NO batch (no throw-away probe chain)
    s, e = shard_range(n, rank, world)
    assert timing.shape == (e - s,)
    want = (torch.arange(n, dtype=torch.float32)[:, None, None] * 100 + torch.as_tensor(labels, dtype=torch.float32)[:, None, None]).expand(n, 16, 51)
    assert full.shape == (n, 16, 51) and torch.equal(full, want)
    g = all_gather_rows(torch.full((e - s, 2), float(rank)), n, world)
    owner = torch.cat([torch.full((shard_range(n, r, world)[1] - shard_range(n, r, world)[0],), float(r)) for r in range(world)])
    assert g.shape == (n, 2) and torch.equal(g[:, 0], owner)
    if rank == 0:
        f = save_generated(tmp, full[:, :, :3].numpy(), labels, np.zeros(n), 16, keypoint_feature=full[:, :, 3:].numpy())
        d = np.load(f)
        assert set(d.files) == {"points", "label", "category", "category_name", "timing", "keypoint_feature"}
        assert d["points"].shape == (n, 16, 3) and d["category"][min(4, n - 1)] == ("03001627" if n > 4 else "02828884") and d["category_name"][0] == "airplane"
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(11, 2), (2, 2), (2, 3)])
def test_multi_rank_generation_gloo(tmp_path, n, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
