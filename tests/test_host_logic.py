"""Host-side decisions that need no GPU (run in the CPU suite)."""
import os

import pytest


def test_winograd_gating_is_per_pass():
    """ops._wino_ok (host logic, no GPU): a pass takes the Winograd path when ITS reduction length reaches WINO_MIN_CHANNELS — config U's
    2592 -> 2592 level both ways, 432 -> 2592 only in the backward-data pass — and never for strides, paddings, other kernel sizes, the
    D<->W exchanged form or when switched off."""
    import torch
    from raygun_amd import ops
    w_sq = torch.empty(2592, 2592, 3, 3, 3, device="meta")
    w_up = torch.empty(2592, 432, 3, 3, 3, device="meta")
    ok = lambda w, **kw: ops._wino_ok(w, kw.get("stride", (1, 1, 1)), kw.get("padding", (0, 0, 0)), kw.get("dw", False), backward=kw.get("backward", False))  # noqa: E731
    assert ok(w_sq) and ok(w_sq, backward=True)
    assert not ok(w_up) and ok(w_up, backward=True)
    assert not ok(torch.empty(432, 2592, 3, 3, 3, device="meta"), backward=True) and ok(torch.empty(432, 2592, 3, 3, 3, device="meta"))
    assert not ok(w_sq, stride=(2, 2, 2)) and not ok(w_sq, padding=(1, 1, 1)) and not ok(w_sq, dw=True)
    assert not ok(torch.empty(2592, 2592, 2, 2, 2, device="meta")) and not ok(torch.empty(2592, 64, 3, 3, 3, device="meta"), backward=True)
    # column counts the batched GEMM cannot tile by 128 / 144 (rg_gemm_batched refuses them) stay on the direct kernels
    assert not ok(torch.empty(160, 1024, 3, 3, 3, device="meta")) and not ok(torch.empty(800, 4000, 3, 3, 3, device="meta"))
    assert not ok(torch.empty(4000, 800, 3, 3, 3, device="meta"), backward=True)      # backward-data of 800 -> 4000: 800 columns
    assert ok(torch.empty(144, 1024, 3, 3, 3, device="meta")) and ok(torch.empty(1024, 1152, 3, 3, 3, device="meta"), backward=True)
    from raygun_amd import _lib
    L = _lib.lib()
    assert [L.rg_gemm_batched_tile_n(n) for n in (128, 144, 150, 160, 256, 288, 310, 800, 2592, 3456)] == [8, 9, 0, 0, 8, 9, 0, 0, 9, 9]
    os.environ["RAYGUN_AMD_WINOGRAD"] = "0"
    try:
        assert not ok(w_sq)
    finally:
        del os.environ["RAYGUN_AMD_WINOGRAD"]


def test_loss_scaler_policy_and_trainer_precision_switch():
    """amp.LossScaler's scale policy (torch.cuda.amp.GradScaler's: halve on overflow, double after `growth_interval` clean steps, the
    optimizer untouched on a skipped step) with the device check stubbed out, and the Trainer's "precision" config switch."""
    import types
    from raygun_amd import amp
    from raygun_amd.torch.train.driver import Trainer

    class Opt:
        param_groups = [{"params": []}]
        grad_scale = 1.0
        calls = []

        def step(self):
            self.calls.append(self.grad_scale)

    sc = amp.LossScaler(init_scale=8.0, growth_interval=2, min_scale=1.0, max_scale=32.0)
    seq = iter([False, False, True, False, False, False, False, True, True, True, True])
    sc.found_inf = lambda params: next(seq)
    o = Opt()
    applied = [sc.step(o) for _ in range(11)]
    assert applied == [True, True, False, True, True, True, True, False, False, False, False]
    assert o.calls == [1 / 8, 1 / 8, 1 / 8, 1 / 8, 1 / 16, 1 / 16] and o.grad_scale == 1.0    # 8 -> 16 -> 8 -> 16 -> 32 ...
    assert sc.scale_value == 2.0 and sc.skipped == 5                                          # ... -> 16 -> 8 -> 4 -> 2
    with pytest.raises(RuntimeError):
        sc.step(types.SimpleNamespace(param_groups=[], step=lambda: None))                    # no grad_scale: not FusedAdam
    assert Trainer(types.SimpleNamespace(precision="f16", loss_scale_kwargs={"init_scale": 4.0})).scaler.scale_value == 4.0
    assert Trainer(types.SimpleNamespace()).scaler is None and Trainer(types.SimpleNamespace(precision="bf16")).scaler is None
    with pytest.raises(ValueError):
        Trainer(types.SimpleNamespace(precision="fp8"))
