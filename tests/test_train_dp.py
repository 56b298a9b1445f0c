"""CPU (gloo, world_size 2 / 3): the data-parallel gradient averaging of the training step (slide_amd/train/dp.py) -- the counterpart
of the reference's apply_gradient_allreduce (pointnet2/distributed.py:99-151): parameters broadcast from rank 0, ONE flat bucket
all-reduced and divided by the world size, parameters without a gradient on some rank included."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _Tiny(nn.Module):
    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.a = nn.Parameter(torch.randn(5, 3, generator=g))
        self.b = nn.Parameter(torch.randn(5, generator=g))
        self.unused = nn.Parameter(torch.randn(4, generator=g))  # gets a gradient on rank 1 only
        self.frozen = nn.Parameter(torch.randn(2, generator=g), requires_grad=False)

    def forward(self, x, use_extra):
        y = (x @ self.a.t() + self.b).pow(2).mean()
        return y + (self.unused.sum() if use_extra else 0.0)


def _worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from slide_amd.train.dp import allreduce_gradients, broadcast_parameters, training_step
    net = _Tiny(seed=rank)                      # ranks start from DIFFERENT parameters ...
    broadcast_parameters(net)                   # ... and leave with rank 0's
    ref = _Tiny(seed=0)
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.equal(p, q), n
    xs = [torch.randn(6, 3, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    loss = net(xs[rank], use_extra=(rank == 1))
    loss.backward()
    bucket = allreduce_gradients(net)
    assert bucket.numel() == 5 * 3 + 5 + 4  # one flat bucket over the trainable parameters
    # expectation: the mean over ranks of the per-rank gradients, computed here on every rank from all shards
    want = {n: torch.zeros_like(p) for n, p in ref.named_parameters() if p.requires_grad}
    for r in range(world):
        m = _Tiny(seed=0)
        m(xs[r], use_extra=(r == 1)).backward()
        for n, p in m.named_parameters():
            if p.grad is not None:
                want[n] += p.grad / world
    for n, p in net.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.allclose(p.grad, want[n], rtol=1e-6, atol=1e-7), n
    assert net.frozen.grad is None
    # a full step through the helper: every rank ends with the same parameters
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=0.1)
    l_, bucket2 = training_step(net, opt, lambda: net(xs[rank], use_extra=True), bucket)
    assert bucket2 is bucket
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    for t in gathered[1:]:
        assert torch.equal(t, gathered[0])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gradient_allreduce_gloo(world):
    mp.spawn(_worker, args=(world, _free_port()), nprocs=world, join=True)


def test_single_process_is_a_no_op():
    from slide_amd.train.dp import allreduce_gradients, broadcast_parameters
    net = _Tiny(seed=3)
    net(torch.ones(2, 3), True).backward()
    g0 = net.a.grad.clone()
    broadcast_parameters(net)
    assert allreduce_gradients(net) is None and torch.equal(net.a.grad, g0)


def test_trainer_host_pieces(tmp_path):
    """EMA weight sets (pointnet2/data_utils/ema.py:20-26), newest-checkpoint search, the config's string-encoded EMA rates, the npz
    epoch iterator's rank shares"""
    import numpy as np
    import torch
    from slide_amd.train.trainer import EmaSet, find_max_ckpt, npz_batches, parse_ema_rate
    assert parse_ema_rate("[0.999, 0.9999]") == [0.999, 0.9999] and parse_ema_rate(None) is None and parse_ema_rate([0.5]) == [0.5]
    net = torch.nn.Linear(3, 2)
    ema = EmaSet(net, [0.9, 0.5])
    w0 = net.weight.detach().clone()
    with torch.no_grad():
        net.weight.add_(1.0)
    ema.update()
    sl = ema.state_list()
    assert torch.allclose(sl[0]["weight"], 0.1 * (w0 + 1) + 0.9 * w0) and torch.allclose(sl[1]["weight"], 0.5 * (w0 + 1) + 0.5 * w0)
    assert set(sl[0]) == {"weight", "bias"}
    ema2 = EmaSet(net, [0.9, 0.5])
    ema2.load_state_list(sl)
    assert torch.equal(ema2.shadows[1][0], sl[1]["weight"])
    assert find_max_ckpt(str(tmp_path)) == -1
    for i in (3, 11, 7):
        (tmp_path / ("pointnet_ckpt_%d.pkl" % i)).write_bytes(b"")
    (tmp_path / "pointnet_ckpt_99.tmp").write_bytes(b"")
    assert find_max_ckpt(str(tmp_path)) == 11
    np.savez(tmp_path / "c.npz", points=np.arange(20 * 4 * 3, dtype=np.float32).reshape(20, 4, 3), label=np.arange(20))
    seen = []
    for rank in range(2):
        ep = list(npz_batches(str(tmp_path / "c.npz"), 4, rank, 2, seed=5))
        assert len(ep) == 2 and ep[0]["points"].shape == (4, 4, 3)
        seen += [int(v) for b in ep for v in b["label"]]
    assert len(set(seen)) == 16   # disjoint shares, full batches only
