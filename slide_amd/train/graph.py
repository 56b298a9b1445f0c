"""The training step as ONE HIP graph: zero_grad -> loss -> backward -> optimizer.step is ~600 small launches (118 HIP kernels of this
library, the rest torch's packing / reduction / Adam kernels) that take 13-20 ms to ISSUE from Python against ~6 ms of device time, so
the step is launch-bound exactly like the sampling loop.  Captured once (torch.cuda.CUDAGraph drives hipStreamBeginCapture on the
current stream -- every kernel of libslide_hip.so launches on torch's current stream) and replayed, it runs at device speed.

What makes the step capturable: epilogue tables are built on the device (functions._gemm), packed weight buffers persist
(functions._packs), schedule tables are cached on the device (losses._table), the random timesteps / noise come from torch's
graph-safe Philox generator, and the optimizer holds its step counter on the device (Adam(capturable=True))."""
import torch
import torch.distributed as dist

from .dp import allreduce_gradients


class GraphedTrainingStep:
    """step = GraphedTrainingStep(net, optimizer, loss_fn); loss = step()  (a device scalar, overwritten by the next replay).
    loss_fn() must read its batch from tensors that stay at the same address (copy_ each new batch into them) and must not
    synchronise.  Under data parallelism (world size > 1) the graph holds forward + backward; the bucketed gradient all-reduce and
    the optimizer step follow eagerly (pointnet2/distributed.py:99-151 reduces after backward as well)."""

    def __init__(self, net, optimizer, loss_fn, warmup=3):
        self.net, self.optimizer = net, optimizer
        self.distributed = dist.is_initialized() and dist.get_world_size() > 1
        if not self.distributed and not all(g.get("capturable", True) for g in optimizer.param_groups):
            raise ValueError("the optimizer step is captured: construct the optimizer with capturable=True")
        self.bucket = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # eager steps: fill the caches, create the optimizer state (torch's capture protocol)
            for _ in range(warmup):
                optimizer.zero_grad(set_to_none=True)
                loss_fn().backward()
                self.bucket = allreduce_gradients(net, self.bucket)
                optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)  # backward then WRITES the gradients (static tensors of the graph's pool)
        with torch.cuda.graph(self.graph):
            loss = loss_fn()
            loss.backward()
            if not self.distributed:
                optimizer.step()
        self.loss = loss.detach()

    def __call__(self):
        self.graph.replay()
        if self.distributed:
            self.bucket = allreduce_gradients(self.net, self.bucket)
            self.optimizer.step()
        return self.loss
