"""CPU: the N>1 path (contiguous per-rank shards, no data-path collective, one all-gather) on gloo with world_size 2,
plus the shard arithmetic against the reference's ceil-division rule."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from slide_amd.generation import (FIVE_CATEGORIES, all_gather_rows, batches, category_layout, category_segments, decode_shard,
                                   fps_start_indices, generate_categories, generate_latents, save_generated, shard_range)


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


def test_category_segments_cover_every_shard():
    """BASELINE configs[3]: 2048 shapes, five categories (labels 0, 2, 3, 4, 6), 8 ranks: 256 per rank, <= 2 categories per rank"""
    lay = category_layout(2048)
    assert [c for c, _, _ in lay] == list(FIVE_CATEGORIES) and [hi - lo for _, lo, hi in lay] == [410, 410, 410, 409, 409]
    seen = []
    for r in range(8):
        segs = category_segments(2048, r, 8)
        assert 1 <= len(segs) <= 2 and sum(hi - lo for _, lo, hi in segs) == 256
        assert segs[0][1] == 256 * r and segs[-1][2] == 256 * (r + 1)
        seen += segs
    for c, lo, hi in lay:  # every category's range is tiled exactly by the segments carrying its label
        mine = sorted((a, b) for cc, a, b in seen if cc == c)
        assert mine[0][0] == lo and mine[-1][1] == hi and all(mine[i][1] == mine[i + 1][0] for i in range(len(mine) - 1))
    assert category_segments(3, 2, 4) == [] or sum(h - l for _, l, h in category_segments(3, 2, 4)) <= 1


def _worker_cat(rank, world, port, total):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def run_segment(cat, lo, hi):  # stands in for (position chain -> feature chain) of one category's weight set
        calls.append((cat, lo, hi))
        idx = torch.arange(lo, hi, dtype=torch.float32)
        return (idx[:, None, None] * 10 + cat).expand(hi - lo, 16, 51).clone()

    full, labels = generate_categories(total, run_segment, rank, world)
    assert calls == category_segments(total, rank, world)
    want = (torch.arange(total, dtype=torch.float32) * 10 + torch.as_tensor(labels, dtype=torch.float32))[:, None, None].expand(total, 16, 51)
    assert full.shape == (total, 16, 51) and torch.equal(full, want)
    assert set(labels.tolist()) <= set(FIVE_CATEGORIES)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total,world", [(23, 2), (7, 3)])
def test_five_category_generation_gloo(total, world):
    mp.spawn(_worker_cat, args=(world, _free_port(), total), nprocs=world, join=True)


class _StubSampler:
    """stands in for a HIP sampler on CPU: state = a tensor that encodes the chain's seed and batch row"""

    def __init__(self, n, width, seed):
        self.n, self.width, self.seed, self.T = n, width, seed, 1
        self.engine = type("E", (), {"cx": width})()

    def begin(self, lab, *xs, **kw):
        self.lab = lab

    def advance(self, steps):
        pass

    def state(self):
        base = torch.arange(self.n, dtype=torch.float32)[:, None, None] + float(self.seed % 1000) * 1000
        return (base + self.lab.float()[:, None, None]).expand(self.n, 16, self.width).clone()


def _worker_chains(rank, world, port, total):
    """CategoryChains.generate with MORE RANKS THAN SHARDS: a rank without chains must enter the same collectives
    as the ranks with chains (it used to take an object all-gather alone: gloo aborted, RCCL would hang)"""
    from slide_amd.generation import CategoryChains
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Chains(CategoryChains):
        def _make_chain(self, pos_cfg, feat_cfg, sd_p, sd_f, n, prec, seed_p, seed_f):
            return _StubSampler(n, 3, seed_p), _StubSampler(n, self.cx, seed_f)

    feat_cfg = {"pointnet_config": {"in_fea_dim": 48}}
    ch = Chains(total, rank, world, None, feat_cfg, lambda c: (None, None), torch.device("cpu"))
    assert (len(ch.chains) == 0) == (shard_range(total, rank, world)[0] >= shard_range(total, rank, world)[1])
    full, labels = ch.generate()
    assert full.shape == (total, 16, 51) and labels.shape == (total,)
    # every row carries its category label in the fractional encoding and is identical on every rank
    gathered = [torch.empty_like(full) for _ in range(world)]
    dist.all_gather(gathered, full)
    assert all(torch.equal(g, full) for g in gathered)
    assert torch.equal(full[:, 0, 0] % 1000 - torch.as_tensor(labels, dtype=torch.float32),
                       torch.cat([torch.arange(hi - lo, dtype=torch.float32) for r in range(world)
                                  for _, lo, hi in category_segments(total, r, world)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total,world", [(2, 3), (1, 2), (9, 2)])
def test_category_chains_more_ranks_than_shards_gloo(total, world):
    mp.spawn(_worker_chains, args=(world, _free_port(), total), nprocs=world, join=True)


def _worker_real_chains(rank, world, port, total, steps):
    """CategoryChains with the REAL HIP samplers at world = 2 (both ranks on the one GPU of this box, gloo)
    must equal the world = 1 result bit for bit: a shape's start noise and in-kernel noise depend on (seed, global index) only"""
    from slide_amd import configs, model_spec
    from slide_amd.generation import CategoryChains
    from slide_amd.synth import synth_state_dict
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    pc, fc = configs.position_ddpm_config(), configs.feature_ddpm_config()
    spec_p, spec_f = model_spec.denoiser_param_spec(pc["pointnet_config"]), model_spec.denoiser_param_spec(fc["pointnet_config"])
    weights = lambda c: (synth_state_dict(spec_p, seed=100 + c), synth_state_dict(spec_f, seed=200 + c))

    class Chains(CategoryChains):
        def generate(self, steps=None):  # gather through the CPU (gloo)
            from slide_amd.generation import generate_categories
            local = self.run(steps).cpu()
            base = self.segments[0][1] if self.segments else 0
            return generate_categories(self.total, lambda c, lo, hi: local[lo - base:hi - base], self.rank, self.world, self.categories,
                                       gather_device=torch.device("cpu"), row_shape=(16, self.cx))

    full, labels = Chains(total, rank, world, pc, fc, weights, dev, prec="fp16", seed=3).generate(steps=steps)
    if rank == 0:
        ref, labels1 = CategoryChains(total, 0, 1, pc, fc, weights, dev, prec="fp16", seed=3).generate(steps=steps)
        assert np.array_equal(labels, labels1)
        assert torch.isfinite(full).all() and torch.equal(full, ref.cpu()), float((full - ref.cpu()).abs().max())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_category_chains_world2_equals_world1_on_the_gpu():
    mp.spawn(_worker_real_chains, args=(2, _free_port(), 13, 8), nprocs=2, join=True)


class _FakeAutoencoder:
    """stands in for PointAutoencoder.decode on CPU: the cloud of a shape encodes its key points, features, label and the FPS start
    index it was handed -- whatever batch it sits in"""

    def __init__(self):
        self.batches = []

    def decode(self, keypoint, feature, ts=None, label=None, fps_start_idx=None):
        b = keypoint.shape[0]
        self.batches.append(b)
        assert fps_start_idx.shape == (b,) and fps_start_idx.dtype == torch.int64 and 0 <= int(fps_start_idx.min()) and int(fps_start_idx.max()) < (1 << 30)
        base = keypoint.sum(dim=(1, 2)) + 10 * feature.sum(dim=(1, 2)) + 1000 * label.float() + 1e4 * (fps_start_idx % 512).float()
        return base[:, None, None] + torch.arange(2048 * 6, dtype=torch.float32).reshape(1, 2048, 6)


def _decode_worker(rank, world, port, n, bs):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    rs = np.random.RandomState(0)
    lat = rs.standard_normal((n, 16, 51)).astype(np.float32)
    labels = np.arange(n) % 13
    s0, e0 = shard_range(n, rank, world)
    ae = _FakeAutoencoder()
    c_local = decode_shard(ae, lat[s0:e0, :, :3], lat[s0:e0, :, 3:], labels[s0:e0], bs, dev, seed=5, global_offset=s0)
    assert c_local.shape == (e0 - s0, 2048, 6) and ae.batches == [hi - lo for lo, hi in batches(0, e0 - s0, bs)]
    full = all_gather_rows(c_local, n, world)
    # one rank, one batch: the same clouds in the same order (per-shape FPS start indices are keyed on the global index)
    ref = decode_shard(_FakeAutoencoder(), lat[:, :, :3], lat[:, :, 3:], labels, n, dev, seed=5, global_offset=0)
    assert full.shape == (n, 2048, 6) and torch.equal(full, ref)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world,bs", [(11, 2, 4), (5, 3, 2), (2, 3, 4)])
def test_sharded_decode_gathers_in_order_gloo(n, world, bs):
    """BASELINE configs[4] on N ranks: every rank decodes its own latent shard, one all-gather of the (n_local, 2048, 6) clouds
    (ranks with short or EMPTY shards included) reproduces the single-rank result"""
    mp.spawn(_decode_worker, args=(world, _free_port(), n, bs), nprocs=world, join=True)


def test_fps_start_indices_depend_on_the_global_index_only():
    dev = torch.device("cpu")
    a = fps_start_indices(7, 0, 3000, 512, dev)
    assert a.shape == (3000,) and int(a.min()) >= 0 and int(a.max()) < 512 and len(set(a.tolist())) > 300
    assert torch.equal(fps_start_indices(7, 1000, 1030, 512, dev), a[1000:1030])  # across a block boundary
    assert torch.equal(fps_start_indices(7, 2047, 2050, 512, dev), a[2047:2050])
    assert not torch.equal(fps_start_indices(8, 0, 64, 512, dev), a[:64])
    assert fps_start_indices(7, 5, 5, 512, dev).shape == (0,)
