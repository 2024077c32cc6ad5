"""Data-parallel ED-LoRA training: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" in CPU tests).

The reference wraps the whole trainer in DistributedDataParallel through accelerate (train_edlora.py:34,70), so
every optimisation step all-reduces the full 152 MB token-embedding table plus all LoRA factors. Only the concept
rows and the LoRA factors ever change (the other rows are restored after each step), so here the gradients of
exactly those tensors live in ONE flat fp32 bucket (about 1.12 M floats = 4.5 MB for two concepts at rank 4) that is
all-reduced once per optimisation step. At that size the collective is latency-bound, not link-bandwidth-bound
(7 xGMI links x ~153 GB/s per GPU), so a single un-bucketed call after backward is the right shape; there is
nothing to overlap it with that would matter (<< 1 % of a step).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from the torchrun / torch.distributed.run environment. Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def barrier():
    if world_size() > 1:
        dist.barrier()


class FlatGradBucket:
    """Gradients of the trainable tensors as views into one contiguous fp32 buffer.

    `p.grad` of every parameter aliases a slice of `self.flat`, autograd accumulates into it in place, `zero()` is
    one memset and `allreduce_mean()` is one collective on the whole buffer — no pack / unpack copies."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, 'no trainable parameters'
        dev = self.params[0].device
        assert all(p.dtype == torch.float32 for p in self.params), 'master LoRA / embedding parameters are fp32'
        self.sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)
        off = 0
        for p, n in zip(self.params, self.sizes):
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def zero(self):
        self.flat.zero_()
        # optimizers / GradScaler may have replaced .grad objects; re-alias if that happened
        off = 0
        for p, n in zip(self.params, self.sizes):
            g = p.grad
            if g is None or g.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def allreduce_mean(self):
        w = world_size()
        if w > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(w)


def reduce_loss_dict(loss_dict):
    """reference mixofshow/utils/util.py:203-229 — average the logged scalars over ranks."""
    keys = sorted(loss_dict.keys())
    vals = torch.stack([loss_dict[k].detach().float().reshape(()) for k in keys])
    w = world_size()
    if w > 1:
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
        vals = vals / w
    return {k: v for k, v in zip(keys, vals)}
