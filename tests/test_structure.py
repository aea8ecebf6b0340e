"""CPU tests of the host logic of the drop-in classes: constructor surface, module trees, state_dict keys,
parameter counts and error behaviour pinned by the reference's own unit tests (tests/torch/networks/*.py)
and by tests/golden/structure.json (generated from the reference).  No kernels run here."""
import functools
import json
import os

import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
STRUCT = json.load(open(os.path.join(G, "structure.json")))


def test_library_exports_every_declared_symbol():
    import re
    from raygun_amd import _lib
    hdr = open(os.path.join(os.path.dirname(G), "..", "include", "raygun_hip.h")).read()
    declared = set(re.findall(r"\b(rg_[a-z0-9_]+)\s*\(", hdr))
    lib = _lib.lib()  # raises if the .so is missing or a bound symbol is absent
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/raygun_hip.h but not exported"
    assert declared == set(_lib.DECLARED), declared ^ set(_lib.DECLARED)
    assert lib.rg_version() >= 100 and not lib.rg_version() >> 16
    # the IEEE-half build of the same sources: every entry point except the mutex watershed (no 16-bit data on that path)
    half = _lib.lib("f16")
    for name in declared - _lib._F16_ABSENT:
        assert hasattr(half, name), f"{name} missing from libraygun_hip_f16.so"
    assert half.rg_version() >> 16 == 1


def test_mtlsd_structure_matches_reference():
    from raygun_amd.torch.models import MTLSDModel
    m = MTLSDModel()
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == STRUCT["mtlsd_keys"]
    assert sum(p.numel() for p in m.parameters()) == STRUCT["mtlsd_params"] == 259146829
    assert sum(p.numel() for p in MTLSDModel(num_affs=20).parameters()) == STRUCT["mtlsd20_params"]


def test_unet_attributes_and_heads():
    # reference tests/torch/networks/unet_test.py:173-184
    from raygun_amd.torch.networks import UNet
    u = UNet(input_nc=3, ngf=16, fmap_inc_factor=2, downsample_factors=[(2, 2, 2), (2, 2, 2)], num_heads=2, padding_type="same")
    assert (u.ndims, u.num_levels, u.num_heads, u.input_nc, u.output_nc) == (3, 3, 2, 3, 16)
    assert len(u.r_conv) == 2 and len(u.r_up[0]) == 2
    with pytest.raises(RuntimeError):
        UNet(1, 4, 2, [(2, 2, 2)], padding_type="weird")
    with pytest.raises(RuntimeError):
        UNet(1, 4, 2, [(2, 2, 2)], downsample_method="avg")


def test_convpass_and_downsample_layers_expose_conv_attributes():
    # reference unet_test.py:64-110
    from raygun_amd.torch.networks.UNet import ConvDownsample, ConvPass, MaxDownsample
    cd = ConvDownsample(3, 8, (3, 3, 3), (2, 2, 2), "ReLU")
    c = cd.conv_pass[0]
    assert c.kernel_size == (3, 3, 3) and c.stride == (2, 2, 2) and c.padding in ("valid", (0, 0, 0)) and c.padding_mode == "reflect"
    cp = ConvPass(2, 4, [(3, 3, 3), (3, 3, 3)], "ReLU", residual=True)
    assert len(cp.conv_pass) == 3 and hasattr(cp, "x_init_map")
    md = MaxDownsample((2, 2, 2), flexible=False)
    with pytest.raises(RuntimeError):
        md.check_mismatch(torch.Size((1, 1, 9, 8, 8)))
    with pytest.raises(RuntimeError):
        ConvPass(1, 1, [(3, 3, 3, 3, 3)], "ReLU")


def test_generator_and_discriminator_structure():
    from raygun_amd.torch.networks import NLayerDiscriminator, ResNet
    from raygun_amd.torch.systems.BaseSystem import BaseSystem
    IN3 = functools.partial(torch.nn.InstanceNorm3d)
    sys_ = BaseSystem.__new__(BaseSystem)
    sys_.ndims = 3
    g = sys_.get_network("resnet", dict(input_nc=1, ngf=64, n_blocks=6, n_downsampling=2, norm_layer=IN3, padding_type="same"))
    d = sys_.get_network("patchdiscriminator", dict(input_nc=1, ngf=64, n_layers=3, kw=4))
    assert sum(p.numel() for p in g.parameters()) == STRUCT["G_params"] == 24876225
    assert sum(p.numel() for p in d.parameters()) == STRUCT["D_params"] == 11047873
    assert {k: list(v.shape) for k, v in d.state_dict().items()} == STRUCT["D_keys"]
    assert len(g[0].model) == STRUCT["len_G_model"] and len(d.model) == STRUCT["len_D_model"]
    # reference resnet_test.py:17-22,154-164 and nlayerdiscriminator_test.py:62-80
    assert len(ResNet(3, n_downsampling=3, n_blocks=4).model) == STRUCT["len_resnet_down3_blocks4"] == 28
    assert sum(p.numel() for p in NLayerDiscriminator(3).parameters()) == STRUCT["D_bn_params"] == 11048769
    assert len(NLayerDiscriminator(3).model) == 12 and len(NLayerDiscriminator(3, n_layers=5).model) == 18
    with pytest.raises(ValueError):
        ResNet(4)
    with pytest.raises(ValueError):
        NLayerDiscriminator(4)


def test_cyclemodel_state_dict_matches_reference():
    from raygun_amd.torch.models import CycleModel
    from raygun_amd.torch.systems.BaseSystem import BaseSystem
    IN3 = functools.partial(torch.nn.InstanceNorm3d)
    s = BaseSystem.__new__(BaseSystem)
    s.ndims = 3
    kw = dict(input_nc=1, ngf=64, n_blocks=6, n_downsampling=2, norm_layer=IN3, padding_type="same")
    m = CycleModel(s.get_network("resnet", dict(kw)), s.get_network("resnet", dict(kw)))
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == STRUCT["cyclemodel_keys"]
    assert m.output_arrays == ["fake_B", "cycled_B", "fake_A", "cycled_A"]


def test_reference_checkpoint_interchange_with_oracle():
    """state_dicts load both ways between the drop-in and the plain-torch restatement (same keys and shapes)."""
    from oracle import torch_ref as O
    from raygun_amd.torch.networks import UNet
    kw = dict(input_nc=1, ngf=4, fmap_inc_factor=2, downsample_factors=[(2, 2, 2)] * 2)
    a, b = UNet(**kw), O.UNet(**kw)
    b.load_state_dict(a.state_dict())
    a.load_state_dict(b.state_dict())


def test_cpu_forward_fails_loudly():
    from raygun_amd.torch.networks import UNet
    u = UNet(1, 4, 2, [(2, 2, 2)])
    with pytest.raises(RuntimeError, match="HIP"):
        u(torch.zeros(1, 1, 20, 20, 20))


def test_norm_mode_and_init_helpers_work_on_dropin_modules():
    from raygun_amd.torch.networks import ResNet, init_weights, set_norm_mode
    g = ResNet(3, ngf=4, n_blocks=1)
    init_weights(g, "kaiming")
    assert all(float(m.bias.abs().sum()) == 0 for m in g.modules() if getattr(m, "bias", None) is not None)
    set_norm_mode(g, "fix_norms")
    assert g.training and all(not m.training for m in g.modules() if "norm" in type(m).__name__.lower())


def test_channels_last_master_weights_interchange_with_reference_checkpoints():
    """hipnn.Conv3d keeps its fp32 master weight in torch.channels_last_3d memory.  That is a memory detail:
    logical shape, state_dict keys and values interchange with plain torch modules (= reference checkpoints) both ways, and
    load_state_dict / .to() / init in place keep the layout."""
    import torch
    from raygun_amd import hipnn
    torch.manual_seed(3)
    ref = torch.nn.Conv3d(5, 7, 3)
    mine = hipnn.Conv3d(5, 7, 3)
    assert tuple(mine.weight.shape) == (7, 5, 3, 3, 3)
    if hipnn.CL_WEIGHTS:
        assert mine.weight.is_contiguous(memory_format=torch.channels_last_3d) and not mine.weight.is_contiguous()
    mine.load_state_dict(ref.state_dict())                       # reference checkpoint -> ours
    assert torch.equal(mine.weight.detach(), ref.weight.detach()) and torch.equal(mine.bias.detach(), ref.bias.detach())
    if hipnn.CL_WEIGHTS:
        assert mine.weight.is_contiguous(memory_format=torch.channels_last_3d)   # layout survives load_state_dict
        torch.nn.init.kaiming_normal_(mine.weight)
        assert mine.weight.is_contiguous(memory_format=torch.channels_last_3d)   # ... and in-place init (init_weights)
    back = torch.nn.Conv3d(5, 7, 3)
    back.load_state_dict(mine.state_dict())                      # ours -> a plain torch module
    assert torch.equal(back.weight.detach(), mine.weight.detach())
    assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
    one = hipnn.Conv3d(4, 2, 1)                                  # 1x1x1 kernels stay contiguous
    assert one.weight.is_contiguous()
