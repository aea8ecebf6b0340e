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
