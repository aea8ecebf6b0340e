"""The RCCL variant of tests/test_multirank_gpu.py's data-parallel and sharded-segmentation cases: one rank per GPU, backend
"nccl" (= RCCL over xGMI).  Needs >= 2 GPUs; on a one-GPU box the same code paths run over gloo in test_multirank_gpu.py."""
import numpy as np
import pytest
import torch

from tests.test_multirank_gpu import NB8, _dp_cyclegan, _dp_unet, _seg_rank, _slab_case, run_ranks

pytestmark = pytest.mark.gpu


def _need(n):
    if torch.cuda.device_count() < n:
        pytest.skip(f"needs {n} GPUs (the gloo twin in test_multirank_gpu.py covers one-GPU boxes)")


def test_dp2_rccl_unet_equals_big_batch():
    _need(2)
    res = run_ranks(_dp_unet, 2, backend="nccl")
    assert res[0][1] == res[1][1] and max(r[0] for r in res) <= 2e-2, res


def test_dp2_rccl_link_cyclegan_equals_big_batch():
    _need(2)
    res = run_ranks(_dp_cyclegan, 2, backend="nccl")
    assert res[0][1] == res[1][1] and max(r[0] for r in res) <= 0.15 and max(r[3] for r in res) <= 0.02, [(r[0], r[3]) for r in res]


def test_segment_slabs_rccl_two_ranks_bit_exact():
    _need(2)
    from oracle import mws as O
    shape, parts = (48, 40, 44), [0, 24, 48]
    labs = run_ranks(_seg_rank, 2, shape, parts, None, (2, 2, 2), True, backend="nccl")
    w, mask = _slab_case(shape, 11, True)
    assert (np.concatenate(labs, 0) == O.slab_mws(w, NB8, 3, (2, 2, 2), mask, parts)).all()


def _rccl_world1(rank, world):
    """A ONE-rank RCCL group on the one GPU of the box: communicator set-up, asynchronous all-reduce handles started from autograd
    hooks, flat buckets — the device-tensor code path the gloo twins cannot reach — with the sums of one rank as the known answer."""
    import torch.distributed as dist
    from raygun_amd import parallel
    from tests.test_multirank_gpu import _mtlsd, _mtlsd_steps
    assert world == 1 and not dist.is_initialized()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and not parallel._host_staged()
    dev = torch.device("cuda", 0)
    from raygun_amd import synth
    raw = torch.from_numpy(synth.uniform((1, 1, 36, 36, 36), 9)).to(dev)
    sync = parallel.GradSync(small_bytes=2048, force=True)
    got = _mtlsd_steps(_mtlsd(dev), raw, 3, sync)
    ref = _mtlsd_steps(_mtlsd(dev), raw, 3, None)
    same = all(torch.equal(a, b) for a, b in zip(got, ref))
    # the small exchanges of the sharded segmentation, on device tensors
    t = torch.arange(5, device=dev)
    gathered = parallel.all_gather_cat(t)
    ints = parallel.all_gather_ints([3, 4])
    return same, sync.stats, gathered.tolist(), [list(i) for i in ints]


def test_rccl_one_rank_group_on_one_gpu():
    same, stats, gathered, ints = run_ranks(_rccl_world1, 1, backend="nccl")[0]
    assert same and stats["large"] > 0 and stats["buckets"] >= 1, stats     # sum over one rank / 1 == the plain run, bit for bit
    assert gathered == [0, 1, 2, 3, 4] and ints == [[3, 4]]
