"""The training step as ONE HIP graph: zero_grad -> loss -> backward -> optimizer.step is ~600 small launches (118 HIP kernels of this
library, the rest torch's packing / reduction / Adam kernels) that take 13-20 ms to ISSUE from Python against ~6 ms of device time, so
the step is launch-bound exactly like the sampling loop.  Captured once (torch.cuda.CUDAGraph drives hipStreamBeginCapture on the
current stream -- every kernel of libslide_hip.so launches on torch's current stream) and replayed, it runs at device speed.

What makes the step capturable: epilogue tables are built on the device (functions._gemm), packed weight buffers persist
(functions._packs), schedule tables are cached on the device (losses._table), the random timesteps / noise come from torch's
graph-safe Philox generator, the optimizer holds its step counter on the device (Adam(capturable=True)), and every long reduction
runs on the library's own column-sum kernel (torch's multi-block reductions return garbage on replay: functions.col_sums).

Data parallel (world size > 1): the gradients live in ONE flat fp32 bucket for good (p.grad = views of it, as DDP's
gradient_as_bucket_view), so a step is  graph A [zero the bucket, loss, backward]  ->  ONE all-reduce of the bucket (RCCL over xGMI;
2.9 / 16 MB: per-link bound ~0.2 ms)  ->  graph B [bucket / world, optimizer.step]  -- three launches from the host, no flatten /
unflatten copies (pointnet2/distributed.py:99-151 coalesces, reduces and copies back after every backward)."""
import torch
import torch.distributed as dist


class GraphedTrainingStep:
    """step = GraphedTrainingStep(net, optimizer, loss_fn); loss = step()  (a device scalar, overwritten by the next replay).
    loss_fn() must read its batch from tensors that stay at the same address (copy_ each new batch into them) and must not
    synchronise.  post_step(): captured after the optimizer step (the EMA weight sets of trainer.py).  Drop every reference to the loss of an earlier EAGER step first (use .detach() / float()): a live autograd graph
    keeps its gradient-accumulation nodes, which stay bound to the stream they were created on and would pull it into the capture."""

    def __init__(self, net, optimizer, loss_fn, warmup=3, post_step=None):
        self.net, self.optimizer = net, optimizer
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if not all(g.get("capturable", True) for g in optimizer.param_groups):
            raise ValueError("the optimizer step is captured: construct the optimizer with capturable=True")
        params = [p for p in net.parameters() if p.requires_grad]
        self.bucket = self.host_bucket = None
        # one side stream for the warm-up steps, the bucket and both captures: autograd runs a leaf's gradient accumulation on the
        # stream its .grad lives on, and a gradient that lives on the default stream would pull that stream into the capture
        side = self.stream = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        if self.world > 1:
            dev = params[0].device
            with torch.cuda.stream(side):
                self.bucket = torch.zeros(sum(p.numel() for p in params), device=dev, dtype=torch.float32)
            o = 0
            for p in params:  # the gradients ARE the bucket: backward accumulates into these views
                p.grad = self.bucket[o:o + p.numel()].view(p.shape)
                o += p.numel()
            if dist.get_backend() == "gloo":  # (CPU tests / ranks sharing one GPU: gloo reduces host memory)
                self.host_bucket = torch.empty(self.bucket.shape, dtype=torch.float32, pin_memory=True)

        def zero():
            if self.bucket is not None:
                self.bucket.zero_()
            else:
                optimizer.zero_grad(set_to_none=True)

        with torch.cuda.stream(side):  # eager steps: fill the caches, create the optimizer state (torch's capture protocol)
            for _ in range(warmup):
                zero()
                loss_fn().backward()
                self._reduce(divide=True)
                optimizer.step()
                if post_step is not None:
                    post_step()
        torch.cuda.current_stream().wait_stream(side)
        self.graph, self.graph_b = torch.cuda.CUDAGraph(), None
        if self.world == 1:
            optimizer.zero_grad(set_to_none=True)  # backward then WRITES the gradients (static tensors of the graph's pool)
        with torch.cuda.graph(self.graph, stream=side):
            if self.world > 1:
                self.bucket.zero_()
            loss = loss_fn()
            loss.backward()
            if self.world == 1:
                optimizer.step()
                if post_step is not None:
                    post_step()
        if self.world > 1:
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph.pool(), stream=side):
                self.bucket.div_(self.world)
                optimizer.step()
                if post_step is not None:
                    post_step()
        self.loss = loss.detach()

    def _reduce(self, divide=False):
        """sum of the ranks' buckets (in place).  divide: also divide by the world size (the eager warm-up steps; in the replayed
        step the division is the first node of graph B)"""
        if self.world == 1:
            return
        if self.host_bucket is not None:
            self.host_bucket.copy_(self.bucket)  # (synchronises the stream)
            dist.all_reduce(self.host_bucket)
            self.bucket.copy_(self.host_bucket)
        else:
            dist.all_reduce(self.bucket)
        if divide:
            self.bucket.div_(self.world)

    def __call__(self):
        self.graph.replay()
        if self.world > 1:
            self._reduce()
            self.graph_b.replay()
        return self.loss
