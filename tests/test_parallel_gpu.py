"""Data parallelism on real GPUs over RCCL (skipped on a 1-GPU box): DP(2) x k steps == 1-GPU big-batch run, for the U-Net
affinity system (GradSync.attach: all-reduces overlapped with backward) and the Link CycleGAN (grad_sync hook)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _mtlsd(dev):
    from raygun_amd import synth
    from raygun_amd.torch.models import MTLSDModel
    kw = {"input_nc": 1, "ngf": 4, "fmap_inc_factor": 2, "downsample_factors": [(2, 2, 2)], "constant_upsample": True}
    m = MTLSDModel(unet_kwargs=kw, num_affs=3)
    m.load_state_dict(synth.fill_state_dict(m.state_dict(), 5))
    return m.to(dev)


def _mtlsd_steps(m, raw, k, sync):
    from raygun_amd import synth
    from raygun_amd.torch.losses import WeightedMSELoss_MTLSD
    from raygun_amd.torch.optimizers import FusedAdam
    opt = FusedAdam(m.parameters(), lr=1e-3)
    lf = WeightedMSELoss_MTLSD()
    if sync is not None:
        sync.attach(m.parameters())
    for it in range(k):
        opt.zero_grad(set_to_none=True)
        affs, lsds = m(raw)
        ga = torch.from_numpy(synth.uniform(tuple(affs.shape[1:]), 60, 0, 1)).to(raw.device).expand_as(affs)
        gl = torch.from_numpy(synth.uniform(tuple(lsds.shape[1:]), 61, 0, 1)).to(raw.device).expand_as(lsds)
        lf(lsds, gl, torch.ones_like(lsds), affs, ga, torch.ones_like(affs)).backward()
        if sync is not None:
            sync(list(m.parameters()))
        opt.step()
    return [p.detach().float().cpu() for p in m.parameters()]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from raygun_amd import parallel, synth
    rank, local, world = parallel.init_from_env(backend="nccl")
    dev = torch.device("cuda", local)
    raw = torch.from_numpy(synth.uniform((world, 1, 36, 36, 36), 9)).to(dev)
    got = _mtlsd_steps(_mtlsd(dev), raw[rank:rank + 1], 3, parallel.GradSync())
    ref = _mtlsd_steps(_mtlsd(dev), raw, 3, None) if rank == 0 else None
    err = max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(got, ref)) if rank == 0 else 0.0
    out.put((rank, err, [float(p.double().sum()) for p in got]))
    torch.distributed.destroy_process_group()


def test_dp2_equals_big_batch_unet():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res[0][2] == res[1][2]                 # identical weights on both ranks (bitwise: same reduced gradients, same Adam)
    assert res[0][1] <= 2e-2, res                 # vs big batch: bf16 activations, batch-2 kernels vs two batch-1 runs
