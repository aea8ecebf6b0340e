"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm,
"gloo" for the CPU tests).  The reference has no distributed code at all (single GPU, no DDP);
this is the build's own component, defined by "N-GPU result == 1-GPU result on the concatenated batch"
(SURVEY.md §8e): gradients are summed over ranks and divided by world size right before each optimiser step.

Samples are independent (InstanceNorm / no norm => no cross-sample statistics), so there is no other exchange.
Gradients are reduced tensor by tensor, largest first, as asynchronous all-reduces that RCCL pipelines over
the 7 xGMI links; flattening into buckets would only add a 1 GB copy for the U-Net (70 % of whose parameters
sit in one 725 MB tensor).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("RAYGUN_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available() and local >= torch.cuda.device_count():
            local %= torch.cuda.device_count()  # more ranks than devices: rehearsal on a smaller box (use gloo; RCCL refuses shared devices)
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _flat(g):
    """The gradient's storage as a 1-D tensor (an element-wise sum does not care about the order): contiguous tensors as they
    are, channels-last conv-weight gradients through their channels-last permutation (zero-copy)."""
    if g.is_contiguous():
        return g
    if g.dim() == 5 and g.is_contiguous(memory_format=torch.channels_last_3d):
        return g.permute(0, 2, 3, 4, 1).reshape(-1)
    if g.dim() == 4 and g.is_contiguous(memory_format=torch.channels_last):
        return g.permute(0, 2, 3, 1).reshape(-1)
    raise RuntimeError("raygun_amd.GradSync: gradients must be dense (contiguous or channels-last)")


class GradSync:
    """Callable(list_of_params): average .grad over all ranks (no-op for world size 1).

    `attach(params)` additionally overlaps the exchange with the backward pass: a post-accumulate-grad hook starts the
    asynchronous all-reduce of a parameter's gradient the moment autograd has produced it (RCCL runs it on its own stream
    while the remaining backward kernels execute), and the call before the optimiser step only waits and scales.  For
    the U-Net 70 % of the 1 GB of gradients is ready half-way through the backward pass.  Only valid where every
    attached parameter receives its gradient in ONE backward pass per step (not for the split CycleGAN losses)."""

    def __init__(self, group=None):
        self.group = group
        self._pending = {}   # id(param) -> (grad tensor, work handle)
        self._hooks = []

    def attach(self, params):
        if world_size() == 1:
            return self
        for p in params:
            if p.requires_grad:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        return self

    def detach(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _on_grad(self, p):
        g = p.grad
        if g is None:
            return
        self._pending[id(p)] = (g, dist.all_reduce(_flat(g), op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def __call__(self, params):
        ws = world_size()
        if ws == 1:
            return
        params = [p for p in params if p.grad is not None]
        started = [self._pending.pop(id(p)) for p in params if id(p) in self._pending]
        done = {id(g) for g, _ in started}
        rest = [p.grad for p in params if id(p.grad) not in done]
        rest.sort(key=lambda g: -g.numel())
        handles = [h for _, h in started]
        for g in rest:
            handles.append(dist.all_reduce(_flat(g), op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        inv = 1.0 / ws
        torch._foreach_mul_([g for g, _ in started] + rest, inv)


def broadcast_parameters(module, src=0):
    """Make every rank start from rank `src`'s weights."""
    if world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def shard_blocks(n_blocks, rank=None, world=None):
    """Static contiguous partition of a block list over ranks (block-wise predict: no collective)."""
    rank = dist.get_rank() if rank is None and world_size() > 1 else (rank or 0)
    world = world_size() if world is None else world
    per = (n_blocks + world - 1) // world
    return range(min(n_blocks, rank * per), min(n_blocks, (rank + 1) * per))
