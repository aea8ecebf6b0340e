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
    if overlap:  # hooks: large gradients are on the wire already, small ones wait for their bucket
        assert len(sync._pending) + len(sync._small) == len(list(net.parameters()))
    sync(list(net.parameters()))
    assert not sync._pending and not sync._small
    # all four tensors are far below 1 MB: they travel in ONE flat bucket, no per-tensor collective
    assert sync.stats == dict(large=0, buckets=1, small=4, rs_ag=0), sync.stats
    big = parallel.GradSync(small_bytes=64, bucket_bytes=256)   # thresholds scaled down: conv weights "large", biases bucketed
    for q in net.parameters():
        q.grad = torch.full_like(q, float(rank + 1))
    big(list(net.parameters()))
    assert big.stats["large"] == 1 and big.stats["small"] == 3 and big.stats["buckets"] >= 1, big.stats
    assert all(torch.allclose(q.grad, torch.full_like(q, 1.5)) for q in net.parameters())
    for q, gsave in zip(net.parameters(), [None] * 4):
        q.grad = None
    loss = net(data[rank:rank + 1]).pow(2).mean()
    loss.backward()
    sync(list(net.parameters()))
    # reference: one process, whole batch (mean over equal-sized samples == mean of per-sample means)
    ref = torch.nn.Sequential(torch.nn.Conv3d(1, 3, 3), torch.nn.InstanceNorm3d(3), torch.nn.ReLU(), torch.nn.Conv3d(3, 1, 1))
    ref.load_state_dict(net.state_dict())
    ref(data).pow(2).mean().backward()
    err = max(float((p.grad - q.grad).abs().max()) for p, q in zip(net.parameters(), ref.parameters()))
    # prescaled mode (bench.py's U-Net step): the backward is seeded with 1 / world, the SUM is then the mean and nothing is divided
    pre = parallel.GradSync(prescaled=True, small_bytes=64, bucket_bytes=256)
    for q in net.parameters():
        q.grad = None
    loss = net(data[rank:rank + 1]).pow(2).mean()
    loss.backward(torch.full_like(loss, 1.0 / world))
    pre(list(net.parameters()))
    err = max(err, max(float((p.grad - q.grad).abs().max()) for p, q in zip(net.parameters(), ref.parameters())))
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


def _tiny_cyclegan(split, grad_sync):
    """The build's CycleModel + Link/SplitCycleLoss host logic around plain torch nets (CPU: 'vanilla' GAN mode and a
    user-supplied L1 keep every op on torch — the HIP kernels have no CPU path)."""
    import itertools
    from raygun_amd.torch.losses import LinkCycleLoss, SplitCycleLoss
    from raygun_amd.torch.models import CycleModel
    torch.manual_seed(3)
    G = lambda: torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv3d(4, 1, 3, padding=1))  # noqa: E731
    D = lambda: torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, stride=2), torch.nn.LeakyReLU(0.2), torch.nn.Conv3d(4, 1, 3))  # noqa: E731
    g1, g2, d1, d2 = G(), G(), D(), D()
    model = CycleModel(g1, g2, None, None, split=split)
    oD = torch.optim.Adam(itertools.chain(d1.parameters(), d2.parameters()), lr=1e-2)
    if split:
        oG1, oG2 = torch.optim.Adam(g1.parameters(), lr=1e-2), torch.optim.Adam(g2.parameters(), lr=1e-2)
        loss = SplitCycleLoss(d1, g1, d2, g2, oG1, oG2, oD, 3, l1_loss=torch.nn.L1Loss(), gan_mode="vanilla", grad_sync=grad_sync)
    else:
        oG = torch.optim.Adam(itertools.chain(g1.parameters(), g2.parameters()), lr=1e-2)
        loss = LinkCycleLoss(d1, g1, d2, g2, oG, oD, 3, l1_loss=torch.nn.L1Loss(), gan_mode="vanilla", grad_sync=grad_sync)
    return model, loss, [g1, g2, d1, d2]


def _cyc_worker(rank, world, port, out, split):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from raygun_amd import parallel
    parallel.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(11)
    A, B = torch.randn(world, 1, 8, 8, 8, generator=g), torch.randn(world, 1, 8, 8, 8, generator=g)
    calls = []
    sync = parallel.GradSync()

    def counted(params):
        calls.append(len(params)); sync(params)
    model, loss, nets = _tiny_cyclegan(split, counted)
    for it in range(2):
        fB, cB, fA, cA = model(real_A=A[rank:rank + 1], real_B=B[rank:rank + 1])
        loss(A[rank:rank + 1], fA, cA, B[rank:rank + 1], fB, cB)
    # the same two steps in one process on the concatenated batch
    rmodel, rloss, rnets = _tiny_cyclegan(split, None)
    for it in range(2):
        fB, cB, fA, cA = rmodel(real_A=A, real_B=B)
        rloss(A, fA, cA, B, fB, cB)
    err = max(float((p - q).abs().max()) for n, m in zip(nets, rnets) for p, q in zip(n.parameters(), m.parameters()))
    out.put((rank, err, calls, [float(p.detach().double().sum()) for n in nets for p in n.parameters()]))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("split", [False, True])
def test_cyclegan_grad_sync_hook_equals_big_batch(split):
    """The CycleGAN losses call `grad_sync` once per optimiser step (Link: G1 u G2, then D — LinkCycleLoss.py:121-135,61-85;
    Split: G1, G2, D).  World size 2 over gloo, 2 iterations: the weights equal a single-process run on the concatenated
    batch (SURVEY.md 8d config 4's correctness criterion) and are identical across ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cyc_worker, args=(r, 2, port, q, split)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] < 5e-6 for r in res), res
    assert len(res[0][2]) == (3 if split else 2) * 2      # hook invocations: per optimiser, per iteration
    assert res[0][3] == res[1][3]


def _comm_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from raygun_amd import parallel
    parallel.init_from_env(backend="gloo")
    ints = parallel.all_gather_ints(list(range(rank)))                     # ragged: rank r contributes r values (rank 0 none)
    t = torch.arange(rank * 2, dtype=torch.int64).reshape(rank, 2) + 100 * rank
    cat = parallel.all_gather_cat(t)
    up, down = parallel.neighbor_exchange(torch.full((2, 3), float(rank)) if rank > 0 else None,
                                          torch.full((2, 3), float(rank) + 0.5) if rank < world - 1 else None)
    out.put((rank, ints, cat.tolist(), None if up is None else float(up[0, 0]), None if down is None else float(down[0, 0])))
    torch.distributed.destroy_process_group()


def test_small_exchanges_of_the_sharded_paths():
    """parallel.all_gather_ints / all_gather_cat / neighbor_exchange (the slab mutex watershed's exchanges), 3 ranks over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ints, cat, up, down in res:
        assert ints == [[], [0], [0, 1]]
        assert cat == [[100, 101], [200, 201], [202, 203]]
        assert up == (None if rank == 0 else rank - 1 + 0.5)            # what rank - 1 sent "to next"
        assert down == (None if rank == 2 else float(rank + 1))          # what rank + 1 sent "to prev"


def test_launch_ranks_supervises_its_children():
    """parallel.launch_ranks: a failing rank ends its siblings at once (no hang in a collective), a timeout ends everything with
    124, a rank killed by a signal is reported as 128 + signal."""
    import sys
    import time
    from raygun_amd.parallel import launch_ranks
    t = time.time()
    rc = launch_ranks([sys.executable, "-c", "import os,sys,time; (sys.exit(3) if os.environ['RANK']=='1' else time.sleep(60))"], 3)
    assert rc == 3 and time.time() - t < 20
    t = time.time()
    assert launch_ranks([sys.executable, "-c", "import time; time.sleep(60)"], 2, timeout=1.0) == 124 and time.time() - t < 20
    assert launch_ranks([sys.executable, "-c", "import os; assert os.environ['WORLD_SIZE']=='2' and os.environ['MASTER_ADDR']=='127.0.0.1'"], 2) == 0
    rc = launch_ranks([sys.executable, "-c", "import os,signal,time; (os.kill(os.getpid(), signal.SIGKILL) if os.environ['RANK']=='0' else time.sleep(60))"], 2)
    assert rc == 128 + 9


def _rsag_worker(rank, world, port, out, hooks):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from raygun_amd import parallel
    parallel.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(11 + rank)
    # a conv weight in channels-last memory whose element count (7 * 5 * 27 = 945) is no multiple of the world size (tail path), a
    # contiguous one that is, and a small one that must stay in the flat bucket
    shapes = [((7, 5, 3, 3, 3), True), ((6, 4, 3, 3, 3), False), ((5,), False)]
    def params():
        ps = []
        for shp, cl in shapes:
            p = torch.nn.Parameter(torch.zeros(shp))
            if cl:
                p.data = p.data.contiguous(memory_format=torch.channels_last_3d)
            ps.append(p)
        return ps
    grads = [torch.randn(shp, generator=g) for shp, _ in shapes]
    res = {}
    for algo in ("allreduce", "rs_ag"):
        ps = params()
        sync = parallel.GradSync(algo=algo, rs_min_bytes=1024, small_bytes=256, timing=True)
        if hooks:
            sync.attach(ps)
            (sum((p * gr).sum() for p, gr in zip(ps, grads))).backward()      # d/dp = gr: the hooks start the exchanges
        else:
            for p, gr in zip(ps, grads):
                p.grad = gr.clone().contiguous(memory_format=torch.channels_last_3d) if p.dim() == 5 and not p.is_contiguous() else gr.clone()
        sync(ps)
        assert sync.stats["rs_ag"] == (2 if algo == "rs_ag" else 0) and sync.stats["large"] == 2 and sync.stats["small"] == 1, (algo, sync.stats)
        w = sync.wait_ms()
        assert len(w) == 1 and w[0] >= 0.0 and sync.wait_ms() == []
        res[algo] = [p.grad.clone() for p in ps]
    out.put((rank, [float((a - b).abs().max()) for a, b in zip(res["allreduce"], res["rs_ag"])], [t.tolist() for t in res["rs_ag"][2:]],
             [float(t.sum()) for t in res["rs_ag"]]))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("hooks", [False, True])
def test_reduce_scatter_all_gather_path_equals_all_reduce(hooks):
    """GradSync(algo="rs_ag"): gradients >= rs_min_bytes travel as reduce-scatter + all-gather (+ an all-reduce of the < world-size tail) —
    same averages as the all-reduce path on every rank, from the call and from the backward hooks (world size 3: 945 elements leave a tail)."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_rsag_worker, args=(r, world, port, q, hooks)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, errs, small, sums in res:
        assert max(errs) <= 1e-6, (rank, errs)
    assert all(r[2] == res[0][2] and r[3] == res[0][3] for r in res)      # rank-identical results
