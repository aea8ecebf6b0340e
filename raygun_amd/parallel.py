"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm,
"gloo" for the CPU tests).  The reference has no distributed code at all (single GPU, no DDP);
this is the build's own component, defined by "N-GPU result == 1-GPU result on the concatenated batch"
(SURVEY.md §8e): gradients are summed over ranks and divided by world size right before each optimiser step.

Samples are independent (InstanceNorm / no norm => no cross-sample statistics), so there is no other exchange.
Large gradients are reduced tensor by tensor, largest first, as asynchronous all-reduces that RCCL pipelines over
the 7 xGMI links (flattening them into buckets would only add a 1 GB copy for the U-Net, 70 % of whose parameters
sit in one 725 MB tensor); small ones travel in flat buckets (GradSync).
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


def launch_ranks(argv, n, timeout=None, rank0_stdout=None, env=None, poll_s=0.2, grace_s=10.0):
    """Start `n` rank processes (torchrun-style environment, rendezvous on 127.0.0.1; backend nccl = RCCL unless the environment
    says otherwise) from a parent that never touches the GPU, and SUPERVISE them: the children are polled; on the first non-zero
    exit, or when `timeout` seconds have passed, the remaining ranks are terminated (SIGTERM, SIGKILL after `grace_s`) — a dead
    rank otherwise leaves its siblings waiting in a collective until the watchdog fires, and the parent with them.  Rank 0 inherits
    `rank0_stdout` (default: the parent's stdout), the other ranks' stdout goes to the parent's stderr.  Returns the first
    non-zero exit code (124 on timeout, like timeout(1)), else 0."""
    import socket
    import subprocess
    import sys
    import time
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    try:
        err = sys.stderr.fileno()
    except Exception:   # a captured / replaced sys.stderr (pytest): the process's own fd 2
        err = 2
    procs = []
    for r in range(n):
        e = dict(os.environ if env is None else env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(list(argv), env=e, stdout=rank0_stdout if r == 0 else err))
    t0, rc = time.monotonic(), 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            if all(c == 0 for c in codes):
                return 0
            if timeout is not None and time.monotonic() - t0 > timeout:
                sys.stderr.write(f"[raygun_amd.launch_ranks] timeout after {timeout:.0f} s: terminating {sum(c is None for c in codes)} rank(s)\n")
                rc = 124
                break
            time.sleep(poll_s)
    finally:
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        t1 = time.monotonic()
        for p in live:
            try:
                p.wait(max(0.1, grace_s - (time.monotonic() - t1)))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc if rc > 0 else 128 - rc if rc < 0 else 1   # a rank killed by signal k exits "negative k": report 128 + k


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


BUCKET_BYTES = 32 << 20     # small gradients travel together in flat fp32 buckets of at most this size
SMALL_BYTES = 1 << 20       # a gradient below this size is "small"
RS_MIN_BYTES = 64 << 20     # algo "rs_ag": gradients from this size on travel as reduce-scatter + all-gather


class GradSync:
    """Callable(list_of_params): average .grad over all ranks (no-op for world size 1).

    Large gradients (>= 1 MB: 99.9 % of the U-Net's 1.04 GB, 70 % of it ONE 725 MB tensor) are all-reduced in place, one
    asynchronous RCCL call each, largest first — flattening them into buckets would only add a gigabyte of copies.  Small ones
    (biases, the 12-channel levels: ~30 tensors of a few KB for the U-Net, ~70 for the CycleGAN's four nets) are packed into
    flat fp32 buckets of <= 32 MB and reduced with one call per bucket instead of one ~10-us launch-bound collective each.

    `attach(params)` additionally overlaps the exchange with the backward pass: a post-accumulate-grad hook starts the
    asynchronous all-reduce of a large gradient the moment autograd has produced it (RCCL runs it on its own stream while
    the remaining backward kernels execute); small gradients are collected by the hooks and their buckets are started by the
    call before the optimiser step, behind the large tensors already on the wire.  For the U-Net 70 % of the gradient bytes are
    ready half-way through the backward pass.  Only valid where every attached parameter receives its gradient in ONE backward
    pass per step (not for the split CycleGAN losses, which call the object once per optimiser instead: LinkCycleLoss.py:121-135).

    Expected cost on one 8-GPU node (xGMI full mesh, 7 links x ~153 GB/s per GPU): a bandwidth-optimal all-reduce
    moves 2 (N-1)/N S per GPU — U-Net S = 1.04 GB: >= 1.7 ms over all seven links (reduce-scatter + all-gather across the mesh),
    >= 11.9 ms if a single ring binds it to one link; CycleGAN generators 0.2 GB + discriminators 0.09 GB: 0.3 / 2.3 ms and
    0.15 / 1.0 ms.  Against a 29-ms U-Net step the exchange hides behind the backward pass in either case."""

    def __init__(self, group=None, bucket_bytes=BUCKET_BYTES, small_bytes=SMALL_BYTES, prescaled=False, force=False, algo=None, rs_min_bytes=RS_MIN_BYTES,
                 timing=False):
        self.group = group
        self.force = bool(force)   # run the collectives in a one-rank group too (tests: the RCCL code path on a one-GPU box)
        # prescaled: the caller already scaled the loss (its backward seed) by 1 / world size, so the SUM is the mean and the pass
        # that divides a gigabyte of gradients afterwards (0.45 ms per U-Net step) is not needed
        self.prescaled = bool(prescaled)
        self.bucket_bytes, self.small_bytes = int(bucket_bytes), int(small_bytes)
        # algo "allreduce" (default): one dist.all_reduce per large gradient, RCCL chooses ring / tree / mesh itself.  "rs_ag": gradients of at least
        # `rs_min_bytes` go as an explicit reduce-scatter + all-gather pair over the same buffer (every GPU owns 1 / N of the sum, each
        # half moves (N-1)/N S over all seven xGMI links at once) — the same sum, rank-identical results; RAYGUN_AMD_GRADSYNC selects it from outside.
        self.algo = algo or os.environ.get("RAYGUN_AMD_GRADSYNC", "allreduce")
        if self.algo not in ("allreduce", "rs_ag"):
            raise ValueError(f"raygun_amd.GradSync: algo {self.algo!r} (allreduce | rs_ag)")
        self.rs_min_bytes = int(rs_min_bytes)
        self._pending = {}   # id(param) -> (grad tensor, work handles)  large gradients started by a hook
        self._small = {}     # id(param) -> grad tensor                   small gradients seen by a hook, bucketed at call time
        self._hooks = []
        self.stats = dict(large=0, buckets=0, small=0, rs_ag=0)   # collectives issued by the last call
        # timing: how long the compute stream (RCCL: device time between two events around the waits) or the host (gloo) waited for the
        # exchange in each call — what the backward pass did NOT hide.  Read with wait_ms().
        self.timing = bool(timing)
        self._waits = []

    def attach(self, params):
        if world_size() == 1 and not self.force:
            return self
        for p in params:
            if p.requires_grad:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        return self

    def detach(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _is_small(self, g):
        return g.numel() * g.element_size() < self.small_bytes

    def _reduce_large(self, g):
        """Start the exchange of one large gradient; returns its work handles (waited for in __call__)."""
        flat = _flat(g).reshape(-1)
        ws = dist.get_world_size(self.group)
        n = flat.numel()
        if self.algo != "rs_ag" or n * flat.element_size() < self.rs_min_bytes or n < ws or ws == 1:
            return [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        rank = dist.get_rank(self.group)
        m = n // ws
        main, tail = flat[:m * ws], flat[m * ws:]
        shard = main[rank * m:(rank + 1) * m]
        handles = []
        if dist.get_backend(self.group) == "nccl":
            # RCCL's in-place forms (receive buffer = send buffer + rank * count); both calls are queued on the communicator's stream in order
            handles.append(dist.reduce_scatter_tensor(shard, main, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            handles.append(dist.all_gather_into_tensor(main, shard, group=self.group, async_op=True))
        else:   # gloo runs asynchronous work on a thread pool in no particular order, and its buffers must not alias
            tmp = torch.empty_like(shard)
            dist.reduce_scatter_tensor(tmp, main.clone(), op=dist.ReduceOp.SUM, group=self.group)
            handles.append(dist.all_gather_into_tensor(main, tmp, group=self.group, async_op=True))
            handles.append(_Keep(tmp))
        if tail.numel():
            handles.append(dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._n_rs += 1
        return handles

    _n_rs = 0

    def _on_grad(self, p):
        g = p.grad
        if g is None:
            return
        if self._is_small(g):
            self._small[id(p)] = g
        else:
            self._pending[id(p)] = (g, self._reduce_large(g))

    def _buckets(self, grads):
        """Greedy fill in the given order: lists of gradients whose fp32 sizes sum to <= bucket_bytes."""
        out, cur, size = [], [], 0
        for g in grads:
            n = g.numel() * 4
            if cur and size + n > self.bucket_bytes:
                out.append(cur); cur, size = [], 0
            cur.append(g); size += n
        if cur:
            out.append(cur)
        return out

    def __call__(self, params):
        ws = world_size()
        if ws == 1 and not self.force:
            return
        params = [p for p in params if p.grad is not None]
        started = [self._pending.pop(id(p)) for p in params if id(p) in self._pending]
        for p in params:
            self._small.pop(id(p), None)
        done = {id(g) for g, _ in started}
        rest = [p.grad for p in params if id(p.grad) not in done]
        large = sorted((g for g in rest if not self._is_small(g)), key=lambda g: -g.numel())
        small = [g for g in rest if self._is_small(g)]
        handles = [h for _, hs in started for h in hs]
        for g in large:
            handles += self._reduce_large(g)
        inv = 1.0 / ws
        flats = []
        for bucket in self._buckets(small):
            flat = torch.cat([_flat(g).reshape(-1).float() for g in bucket])
            flats.append((flat, bucket, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
        self.stats = dict(large=len(started) + len(large), buckets=len(flats), small=len(small), rs_ag=self._n_rs)
        self._n_rs = 0
        mark = self._mark() if self.timing else None
        for h in handles:
            h.wait()
        for _, _, h in flats:
            h.wait()
        if mark is not None:
            self._waits.append((mark, self._mark()))
        big = [g for g, _ in started] + large
        if big and not self.prescaled:
            torch._foreach_mul_(big, inv)
        for flat, bucket, h in flats:
            if not self.prescaled:
                flat.mul_(inv)
            off = 0
            for g in bucket:
                n = g.numel()
                _flat(g).reshape(-1).copy_(flat[off:off + n])
                off += n

    def _mark(self):
        if dist.get_backend(self.group) == "nccl":   # wait() only makes the compute stream wait: the stall is device time between two events
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time
        return time.perf_counter()

    def wait_ms(self, reset=True):
        """Per call since the last reset: milliseconds the compute stream (RCCL) / the host (gloo) spent waiting for the exchange."""
        out = []
        for a, b in self._waits:
            if isinstance(a, float):
                out.append((b - a) * 1e3)
            else:
                b.synchronize()
                out.append(a.elapsed_time(b))
        if reset:
            self._waits = []
        return out


class _Keep:
    """A "work handle" that only keeps a staging buffer alive until the waits are over."""

    def __init__(self, t):
        self.t = t

    def wait(self):
        return True


# ---- small exchanges of the sharded non-training paths (slab mutex watershed) ---------------------------------------------------
# RCCL ("nccl") moves device tensors; gloo — the CPU tests, and the one-GPU rehearsals where several ranks share a device, which
# RCCL refuses — implements all_gather / send / recv for host tensors only, so there the payload is staged through the host.
def _host_staged(group=None):
    return dist.get_backend(group) != "nccl"


def all_gather_ints(vals, group=None):
    """[[rank 0's ints], [rank 1's ints], ...] — per-rank lists may differ in length."""
    ws = dist.get_world_size(group)
    dev = "cpu" if _host_staged(group) else torch.device("cuda", torch.cuda.current_device())
    n = torch.tensor([len(vals)], dtype=torch.int64, device=dev)
    ns = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(ns, n, group=group)
    mx = max(1, max(int(v.item()) for v in ns))
    pad = torch.zeros(mx, dtype=torch.int64, device=dev)
    if len(vals):
        pad[:len(vals)] = torch.tensor([int(v) for v in vals], dtype=torch.int64)
    bufs = [torch.zeros_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad, group=group)
    return [[int(v) for v in b[:int(k.item())].tolist()] for b, k in zip(bufs, ns)]


def all_gather_cat(t, group=None):
    """torch.cat of every rank's `t` along dim 0 (lengths may differ; trailing dims must agree), on t's device."""
    ws = dist.get_world_size(group)
    lens = [v[0] for v in all_gather_ints([t.shape[0]], group)]
    mx = max(1, max(lens))
    staged = _host_staged(group) and t.is_cuda
    src = t.cpu() if staged else t
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=src.device)
    pad[:t.shape[0]] = src
    bufs = [torch.zeros_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad, group=group)
    out = torch.cat([b[:k] for b, k in zip(bufs, lens)])
    return out.to(t.device) if staged else out


def neighbor_exchange(to_prev, to_next, group=None):
    """Send `to_prev` to rank - 1 and `to_next` to rank + 1 (None at the ends of the chain); returns (from_prev, from_next),
    tensors shaped like what was sent that way (None at the ends).  All ranks call it."""
    rank, ws = dist.get_rank(group), dist.get_world_size(group)
    staged = _host_staged(group)
    dev = (to_prev if to_prev is not None else to_next).device if (to_prev is not None or to_next is not None) else None
    h = lambda t: t.cpu().contiguous() if staged else t.contiguous()  # noqa: E731
    ops_, from_prev, from_next = [], None, None
    if rank > 0 and to_prev is not None:
        sp = h(to_prev); from_prev = torch.empty_like(sp)
        ops_ += [dist.P2POp(dist.isend, sp, dist.get_global_rank(group, rank - 1) if group is not None else rank - 1, group),
                 dist.P2POp(dist.irecv, from_prev, dist.get_global_rank(group, rank - 1) if group is not None else rank - 1, group)]
    if rank < ws - 1 and to_next is not None:
        sn = h(to_next); from_next = torch.empty_like(sn)
        ops_ += [dist.P2POp(dist.isend, sn, dist.get_global_rank(group, rank + 1) if group is not None else rank + 1, group),
                 dist.P2POp(dist.irecv, from_next, dist.get_global_rank(group, rank + 1) if group is not None else rank + 1, group)]
    for r in (dist.batch_isend_irecv(ops_) if ops_ else []):
        r.wait()
    if staged:
        from_prev = from_prev.to(dev) if from_prev is not None else None
        from_next = from_next.to(dev) if from_next is not None else None
    return from_prev, from_next


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
