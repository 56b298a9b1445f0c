"""Data-parallel training over the GPUs of a node: parameters replicated, batches sharded, gradients averaged -- the counterpart of
the reference's apply_gradient_allreduce (pointnet2/distributed.py:99-151: broadcast from rank 0, one coalesced all-reduce of all
gradients after backward, division by the world size).

MI355X-first: the whole denoiser is 2.9 / 16 MB of fp32 gradients, so ONE flat bucket = one ring all-reduce over xGMI per step
(per-link bound: 16 MB x 2 (7/8) / 153 GB/s ~ 0.2 ms); backend "nccl" IS RCCL on ROCm, "gloo" in the CPU tests.  No hooks, no
wrapper module: call `allreduce_gradients(module)` between backward() and optimizer.step()."""
import torch
import torch.distributed as dist


def broadcast_parameters(module, src=0):
    """every rank starts from rank `src`'s parameters and buffers (distributed.py:95-97, :111-114)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in module.state_dict().values():
        if torch.is_tensor(t):
            dist.broadcast(t, src)


def allreduce_gradients(module, bucket=None):
    """averages the gradients of `module` over the ranks through ONE flat fp32 bucket (returned for reuse on the next step);
    parameters without a gradient contribute zeros, so every rank reduces the same layout"""
    params = [p for p in module.parameters() if p.requires_grad]
    if not dist.is_initialized() or dist.get_world_size() == 1 or not params:
        return bucket
    n = sum(p.numel() for p in params)
    # (gloo reduces host memory: the CPU tests, and the 2-ranks-on-one-GPU test; RCCL reduces in HBM over xGMI)
    dev = torch.device("cpu") if dist.get_backend() == "gloo" else params[0].device
    if bucket is None or bucket.numel() != n or bucket.device != dev:
        bucket = torch.empty(n, device=dev, dtype=torch.float32)
    o = 0
    for p in params:
        k = p.numel()
        if p.grad is None:
            bucket[o:o + k].zero_()
        else:
            bucket[o:o + k].copy_(p.grad.reshape(-1))
        o += k
    dist.all_reduce(bucket)
    bucket /= dist.get_world_size()
    o = 0
    for p in params:
        k = p.numel()
        if p.grad is None:
            p.grad = bucket[o:o + k].reshape(p.shape).clone()
        else:
            p.grad.copy_(bucket[o:o + k].reshape(p.shape))
        o += k
    return bucket


def training_step(net, optimizer, loss_fn, bucket=None):
    """zero_grad -> loss_fn() (scalar) -> backward -> gradient all-reduce -> optimizer.step(); returns (loss value, bucket)"""
    optimizer.zero_grad(set_to_none=False)
    loss = loss_fn()
    loss.backward()
    bucket = allreduce_gradients(net, bucket)
    optimizer.step()
    return loss.detach(), bucket
