"""CPU, world_size 2, gloo: the data-parallel plumbing (raygun_amd.parallel) — gradient averaging equals the
big-batch gradient, parameter broadcast, static block sharding."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out, overlap=False):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from raygun_amd import parallel
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Conv3d(1, 3, 3), torch.nn.InstanceNorm3d(3), torch.nn.ReLU(), torch.nn.Conv3d(3, 1, 1))
    parallel.broadcast_parameters(net)
    g = torch.Generator().manual_seed(7)
    data = torch.randn(world, 1, 8, 8, 8, generator=g)  # same on all ranks; rank r trains on sample r
    sync = parallel.GradSync()
    if overlap:  # all-reduces start from post-accumulate-grad hooks during backward; the call only waits and scales
        sync.attach(net.parameters())
    loss = net(data[rank:rank + 1]).pow(2).mean()
    loss.backward()
    if overlap:
        assert len(sync._pending) == len(list(net.parameters()))
    sync(list(net.parameters()))
    assert not sync._pending
    # reference: one process, whole batch (mean over equal-sized samples == mean of per-sample means)
    ref = torch.nn.Sequential(torch.nn.Conv3d(1, 3, 3), torch.nn.InstanceNorm3d(3), torch.nn.ReLU(), torch.nn.Conv3d(3, 1, 1))
    ref.load_state_dict(net.state_dict())
    ref(data).pow(2).mean().backward()
    err = max(float((p.grad - q.grad).abs().max()) for p, q in zip(net.parameters(), ref.parameters()))
    blocks = list(parallel.shard_blocks(7))
    out.put((rank, err, blocks, [float(p.detach().sum()) for p in net.parameters()]))
    torch.distributed.destroy_process_group()


import pytest


@pytest.mark.parametrize("overlap", [False, True])
def test_gradsync_equals_big_batch_and_sharding(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] < 1e-6 for r in res), res
    assert res[0][2] == [0, 1, 2, 3] and res[1][2] == [4, 5, 6]
    assert res[0][3] == res[1][3]  # broadcast made the weights identical


def test_single_process_is_noop():
    from raygun_amd import parallel
    p = torch.nn.Parameter(torch.ones(3)); p.grad = torch.full((3,), 2.0)
    parallel.GradSync()([p])
    assert p.grad.tolist() == [2.0, 2.0, 2.0]
    assert list(parallel.shard_blocks(5, rank=0, world=1)) == [0, 1, 2, 3, 4]
