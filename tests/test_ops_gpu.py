"""GPU parity of every operator (through the C-ABI) against plain torch fp32 on the CPU.

Inputs/weights are rounded to bf16 first so that the only differences are accumulation order and
the bf16 rounding of the stored result: tolerance 1.5e-2 of the tensor's max-abs (bf16 has 8
significant bits: one output rounding is <= 2^-9 relative; gradients pass through 2-3 roundings).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from raygun_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1.5e-2


def bf(t):
    return t.to(torch.bfloat16).float()


def rnd(shape, seed, lo=-1.0, hi=1.0):
    return bf(torch.from_numpy(synth.uniform(shape, seed, lo, hi)))


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def run_pair(hip_fn, ref_fn, inputs, params, gseed=7):
    """inputs/params: lists of CPU fp32 tensors. Returns errors for output, input grads, param grads."""
    dev = torch.device("cuda")
    xi = [t.clone().to(dev).requires_grad_(True) for t in inputs]
    pi = [t.clone().to(dev).requires_grad_(True) for t in params]
    y = hip_fn(*xi, *pi)
    xr = [t.clone().requires_grad_(True) for t in inputs]
    pr = [t.clone().requires_grad_(True) for t in params]
    yr = ref_fn(*xr, *pr)
    g = rnd(tuple(yr.shape), gseed)
    y.backward(g.to(dev).to(y.dtype))
    yr.backward(g)
    errs = {"y": relerr(y, yr)}
    for i, (a, b) in enumerate(zip(xi, xr)):
        errs[f"gx{i}"] = relerr(a.grad, b.grad)
    for i, (a, b) in enumerate(zip(pi, pr)):
        errs[f"gp{i}"] = relerr(a.grad, b.grad)
    return errs


def check(errs, tol=TOL):
    bad = {k: v for k, v in errs.items() if not v <= tol}
    assert not bad, f"errors above {tol}: {bad} (all: {errs})"


CONV_CASES = [
    # cin, cout, k, stride, pad, spatial, act
    (3, 5, 3, 1, 0, (10, 9, 11), "relu"),
    (1, 12, 3, 1, 0, (12, 12, 12), "relu"),
    (12, 12, 3, 1, 0, (9, 10, 11), "none"),
    (72, 40, 3, 1, 0, (7, 8, 9), "relu"),
    (40, 144, 3, 1, 1, (6, 6, 6), "leaky"),
    (16, 8, 1, 1, 0, (5, 6, 7), "sigmoid"),
    (8, 16, 3, 2, 1, (9, 10, 12), "none"),
    (1, 8, 4, 2, 1, (12, 12, 12), "leaky"),
    (8, 8, 4, 1, 1, (7, 7, 7), "none"),
    (4, 6, 7, 1, 0, (9, 9, 10), "none"),
    (200, 24, 3, 1, 0, (5, 5, 6), "relu"),
    (6, 4, (3, 1, 2), (2, 1, 1), (1, 0, 0), (8, 6, 7), "none"),
    # wide-channel kernel (igemm2: LDS-DMA ring, BK=64): Cin >= 64 and Cout >= 48
    (96, 160, 3, 1, 1, (6, 7, 8), "relu"),
    (64, 64, 3, 2, 1, (9, 10, 12), "none"),
    (72, 80, 3, 1, 0, (7, 8, 9), "leaky"),
    (136, 48, 1, 1, 0, (5, 6, 7), "none"),
    (144, 432, 3, 1, 0, (6, 6, 7), "relu"),
    (64, 72, 7, 1, 0, (9, 9, 9), "none"),
    (128, 64, 4, 2, 1, (10, 10, 10), "leaky"),
    # line-aligned kw-reuse weight-gradient kernel (stride-1 W, output lines >= 16 voxels): 144x144/3, 64x64/7, 128x128/4 taps
    (144, 144, 3, 1, 0, (4, 5, 21), "relu"),
    (144, 288, 3, 1, 1, (3, 4, 40), "none"),
    (64, 64, 7, 1, 0, (8, 9, 26), "none"),
    (128, 128, 4, 1, 1, (4, 5, 33), "leaky"),
    (128, 128, 3, 1, 1, (3, 3, 18), "none"),
    # same kernel, flat row order (valid stride-1 conv over dense planes: the (h,w) plane walked as one line in the input pitch)
    (144, 144, 3, 1, 0, (4, 19, 19), "relu"),
    (64, 64, 7, 1, 0, (8, 30, 30), "none"),
    (80, 72, 3, 1, 0, (5, 23, 37), "none"),
    # CycleGAN generator stem/tail and PatchGAN layers (networks/ResNet.py:283-345, NLayerDiscriminator.py): folded
    # single-channel stems, one-channel tails of wide inputs, strided 4^3 convs
    (1, 64, 4, 2, 1, (12, 12, 12), "leaky"),
    (1, 32, 7, 1, 0, (10, 11, 12), "relu"),
    (64, 1, 7, 1, 0, (9, 9, 10), "tanh"),
    (256, 1, 4, 1, 1, (5, 6, 6), "none"),
    (72, 1, 3, 1, 1, (5, 6, 7), "none"),
    (24, 40, 3, 1, 1, (6, 6, 7), "relu"),
]


@pytest.mark.parametrize("cin,cout,k,s,p,sp,act", CONV_CASES)
def test_conv3d(cin, cout, k, s, p, sp, act):
    from raygun_amd import ops
    kk = (k,) * 3 if isinstance(k, int) else k
    x = rnd((2, cin) + sp, 1)
    w = rnd((cout, cin) + kk, 2) * (1.0 / np.sqrt(cin * np.prod(kk)))
    w = bf(w)
    b = rnd((cout,), 3)
    afn = {"none": lambda t: t, "relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.2), "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act]
    errs = run_pair(lambda x_, w_, b_: ops.conv3d(x_, w_, b_, s, p, act, 0.2),
                    lambda x_, w_, b_: afn(F.conv3d(x_, w_, b_, s, p)), [x], [w, b])
    check(errs)


CONVT_CASES = [
    (8, 6, 3, 2, 1, 1, (5, 6, 7), "none"),
    (16, 8, 3, 2, 1, 1, (4, 4, 5), "relu"),
    (4, 4, 2, 2, 0, 0, (5, 5, 5), "none"),
    (6, 10, 3, 1, 1, 0, (5, 6, 4), "none"),
    (4, 3, 3, 2, 0, 0, (4, 5, 6), "none"),
    (128, 64, 3, 2, 1, 1, (5, 6, 7), "relu"),
    (64, 80, 3, 2, 1, 1, (6, 5, 4), "none"),
]


@pytest.mark.parametrize("cin,cout,k,s,p,op,sp,act", CONVT_CASES)
def test_conv_transpose3d(cin, cout, k, s, p, op, sp, act):
    from raygun_amd import ops
    x = rnd((2, cin) + sp, 11)
    w = bf(rnd((cin, cout, k, k, k), 12) * (1.0 / np.sqrt(cin * k ** 3)))
    b = rnd((cout,), 13)
    afn = {"none": lambda t: t, "relu": F.relu}[act]
    errs = run_pair(lambda x_, w_, b_: ops.conv_transpose3d(x_, w_, b_, s, p, op, act),
                    lambda x_, w_, b_: afn(F.conv_transpose3d(x_, w_, b_, s, p, op)), [x], [w, b])
    check(errs)


def test_conv_no_bias_and_frozen_weight():
    from raygun_amd import ops
    x = rnd((1, 8, 6, 6, 6), 21)
    w = bf(rnd((8, 8, 3, 3, 3), 22) * 0.1)
    dev = torch.device("cuda")
    xi = x.to(dev).requires_grad_(True)
    wi = w.to(dev)  # no grad: discriminator frozen during the generator phase (LinkCycleLoss.py:117-119)
    y = ops.conv3d(xi, wi, None, 1, 1)
    y.float().sum().backward()
    xr = x.clone().requires_grad_(True)
    yr = F.conv3d(xr, w, None, 1, 1)
    yr.sum().backward()
    assert relerr(y, yr) < TOL and relerr(xi.grad, xr.grad) < TOL


@pytest.mark.parametrize("shape,factor", [((1, 3, 9, 7, 8), (2, 2, 2)), ((2, 12, 8, 8, 8), (2, 2, 2)),
                                          ((1, 20, 6, 9, 6), (1, 3, 2))])
def test_maxpool(shape, factor):
    from raygun_amd import ops
    x = rnd(shape, 31)
    errs = run_pair(lambda t: ops.maxpool3d(t, factor), lambda t: F.max_pool3d(t, factor, factor), [x], [])
    check(errs, 1e-6)


@pytest.mark.parametrize("cf,cg", [(3, 4), (12, 72), (8, 16), (72, 432)])
def test_upcat(cf, cg):
    from raygun_amd import ops
    f = rnd((1, cf, 20, 20, 20), 41)
    g = rnd((1, cg, 7, 7, 7), 42)

    def ref(f_, g_):
        up = F.interpolate(g_, scale_factor=(2, 2, 2), mode="nearest")[..., 1:13, 1:13, 1:13]
        return torch.cat([f_[..., 4:16, 4:16, 4:16], up], 1)

    errs = run_pair(lambda f_, g_: ops.upcat(f_, g_, (2, 2, 2), (12, 12, 12)), ref, [f, g], [])
    check(errs, 4e-3)  # gg sums 8 bf16 values then rounds once


@pytest.mark.parametrize("act", ["none", "relu", "leaky"])
@pytest.mark.parametrize("c", [5, 64])
def test_instance_norm(act, c):
    from raygun_amd import ops
    x = bf(rnd((2, c, 6, 7, 8), 51) * 3 + 1)
    afn = {"none": lambda t: t, "relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.2)}[act]
    errs = run_pair(lambda t: ops.instance_norm(t, 1e-5, act, 0.2), lambda t: afn(F.instance_norm(t, eps=1e-5)), [x], [])
    check(errs)


def test_instance_norm_residual():
    from raygun_amd import ops
    x = bf(rnd((1, 16, 6, 6, 6), 52) * 2)
    r = rnd((1, 16, 6, 6, 6), 53)
    errs = run_pair(lambda t, r_: ops.instance_norm(t, 1e-5, "none", 0.0, r_), lambda t, r_: F.instance_norm(t) + r_, [x, r], [])
    check(errs)


@pytest.mark.parametrize("mode", ["reflect", "zeros", "replicate"])
@pytest.mark.parametrize("pad", [1, 3])
def test_pad3d(mode, pad):
    from raygun_amd import ops
    x = rnd((2, 5, 6, 7, 8), 61)
    ref = (lambda t: F.pad(t, (pad,) * 6, mode=mode)) if mode != "zeros" else (lambda t: F.pad(t, (pad,) * 6))
    errs = run_pair(lambda t: ops.pad3d(t, pad, mode), ref, [x], [])
    check(errs, 4e-3)


def test_add_and_activation():
    from raygun_amd import ops
    a, b = rnd((1, 12, 5, 5, 5), 71), rnd((1, 12, 5, 5, 5), 72)
    check(run_pair(ops.add, lambda p, q: p + q, [a, b], []), 4e-3)
    check(run_pair(lambda t: ops.activation(t, "leaky", 0.2), lambda t: F.leaky_relu(t, 0.2), [a], []), 4e-3)
    check(run_pair(lambda t: ops.activation(t, "sigmoid"), torch.sigmoid, [a], []), 8e-3)


def test_to_cl_of_channel_slices_and_tight_channels_last_sources():
    """`to_cl` of 16-bit views whose voxel pitch does NOT cover zero-pad lanes behind the logical channels — a channel slice of a wider tensor (what
    torch.cat's backward hands on), a caller's tightly packed channels-last tensor: the values are those of the view, pad lanes come out zero, and the
    source is never read beyond its channels (the kernel once read whole 16-byte groups there: behind the last voxel that is past the allocation —
    DESIGN.md "the intermittent abort")."""
    from raygun_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    wide = ops.to_cl(torch.randn(2, 24, 6, 5, 7, generator=g).to(dev))
    cases = [wide[:, 12:24], wide[:, 4:23], wide[:, :12],
             torch.randn(1, 4, 20, 20, 20, generator=g).to(dev).to(ops.storage_dtype()).contiguous(memory_format=torch.channels_last_3d),
             torch.randn(2, 20, 3, 4, 5, generator=g).to(dev).to(ops.storage_dtype()).contiguous(memory_format=torch.channels_last_3d)[:, 3:]]
    # the geometry of the library layout over a storage that ends at the last LOGICAL element (what torch's as_strided backward returns): not is_cl
    short = torch.empty_strided((1, 1, 6, 6, 6), (1728, 1, 288, 48, 8), dtype=ops.storage_dtype(), device=dev).copy_(torch.randn(1, 1, 6, 6, 6, generator=g))
    assert short.untyped_storage().nbytes() < 6 * 6 * 6 * 8 * 2 and not ops.is_cl(short) and ops.is_cl(wide) and ops.is_cl(wide[:, :12])
    cases.append(short)
    for v in cases:
        out = ops.to_cl(v)
        assert ops.is_cl(out) and out.shape == v.shape
        assert torch.equal(out.float(), v.float())
        c, lanes = v.shape[1], ops.lanes_of(v.shape[1])
        if lanes > c and out.data_ptr() != v.data_ptr():   # a fresh tensor in the library layout has zero pad lanes (an aligned leading slice is used as is)
            pad = torch.as_strided(out, (out.shape[0], lanes - c) + tuple(out.shape[2:]), out.stride(), out.storage_offset() + c)
            assert not pad.any()
        assert torch.equal(ops.from_cl(out), v.float())


def test_losses():
    from raygun_amd import ops
    p = torch.from_numpy(synth.uniform((1, 3, 9, 9, 9), 81))
    t = torch.from_numpy(synth.uniform((1, 3, 9, 9, 9), 82))
    w = (torch.from_numpy(synth.uniform((1, 3, 9, 9, 9), 83)) > 0.2).float() * 1.5

    def wmse(p_, t_, w_):
        sc = w_ * (p_ - t_) ** 2
        return sc[w_ > 0].mean() if len(torch.nonzero(sc)) else sc.mean()

    dev = torch.device("cuda")
    for hip, ref in [(lambda q: ops.mse_const(q, 1.0), lambda q: F.mse_loss(q, torch.ones_like(q))),
                     (lambda q: ops.smooth_l1(q * 3, t.to(q.device) * 3), lambda q: F.smooth_l1_loss(q * 3, t * 3)),
                     (lambda q: ops.weighted_mse(q, t.to(q.device), w.to(q.device)), lambda q: wmse(q, t, w)),
                     (lambda q: ops.weighted_mse(q, t.to(q.device), 0 * w.to(q.device)), lambda q: wmse(q, t, 0 * w))]:
        a = p.clone().to(dev).requires_grad_(True)
        b = p.clone().requires_grad_(True)
        la, lb = hip(a), ref(b)
        (la * 2).backward(); (lb * 2).backward()
        assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(lb)))
        assert float((a.grad.cpu() - b.grad).abs().max()) <= 1e-6 + 1e-5 * float(b.grad.abs().max())


@pytest.mark.parametrize("kind,nd,act", [("instance", 3, "none"), ("instance", 3, "relu"), ("instance_rs", 2, "leaky"), ("batch", 3, "leaky"),
                                         ("batch", 2, "none"), ("batch_nomom", 3, "relu"), ("batch_noaffine", 3, "none")])
def test_affine_instance_norm_and_batch_norm_match_torch(kind, nd, act):
    """InstanceNormNd(affine=True[, track_running_stats]) and BatchNormNd (training: batch statistics + running-statistics update incl.
    num_batches_tracked and momentum=None; eval: running statistics) on the HIP norm kernels (ops.norm_affine) against torch.nn on the CPU
    with bf16-exact inputs: outputs, input gradient, d gamma / d beta, the running buffers after two training steps, and an eval pass."""
    from raygun_amd import hipnn
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(11)
    c, n = 12, 3
    shape = (n, c, 5, 6, 7) if nd == 3 else (n, c, 9, 10)
    tname = ("InstanceNorm" if kind.startswith("instance") else "BatchNorm") + f"{nd}d"
    kw = {"instance": dict(affine=True), "instance_rs": dict(affine=True, track_running_stats=True, momentum=0.05), "batch": dict(momentum=0.2),
          "batch_nomom": dict(momentum=None), "batch_noaffine": dict(affine=False)}[kind]
    ref = getattr(torch.nn, tname)(c, **kw)
    mine = getattr(hipnn, tname)(c, **kw)
    if kw.get("affine", True):
        with torch.no_grad():
            ref.weight.copy_(torch.randn(c, generator=g) * 0.5 + 1.0); ref.bias.copy_(torch.randn(c, generator=g) * 0.3)
    mine.load_state_dict(ref.state_dict()); mine = mine.to(dev)
    aref = {"none": lambda t: t, "relu": torch.relu, "leaky": lambda t: F.leaky_relu(t, 0.2)}[act]
    seq = hipnn.FusedSequential(*([mine] + ({"none": [], "relu": [hipnn.ReLU()], "leaky": [hipnn.LeakyReLU(0.2)]}[act])))   # norm + activation fuse
    for it in range(2):
        x0 = (torch.randn(shape, generator=g) * 1.5 + 0.3).bfloat16().float()
        gy = torch.randn(shape, generator=g).bfloat16().float()
        xr = x0.clone().requires_grad_(True); xm = x0.clone().to(dev).requires_grad_(True)
        yr = aref(ref(xr)); yr.backward(gy)
        ym = seq(xm); ym.backward(gy.to(dev))
        assert ym.shape == yr.shape and ym.dtype == torch.float32
        assert float((ym.detach().cpu() - yr.detach()).abs().max()) <= 2e-2 * max(1.0, float(yr.abs().max())), (kind, it)
        assert float((xm.grad.cpu() - xr.grad).abs().max()) <= 2e-2 * max(1.0, float(xr.grad.abs().max())), (kind, it)
        if kw.get("affine", True):
            for a, b in ((mine.weight.grad, ref.weight.grad), (mine.bias.grad, ref.bias.grad)):
                assert float((a.cpu() - b).abs().max()) <= 1.5e-2 * max(1.0, float(b.abs().max())), (kind, it)
            mine.zero_grad(); ref.zero_grad()
        if getattr(ref, "track_running_stats", False):
            assert torch.allclose(mine.running_mean.cpu(), ref.running_mean, rtol=1e-4, atol=1e-5)
            assert torch.allclose(mine.running_var.cpu(), ref.running_var, rtol=2e-4, atol=1e-5)
            if hasattr(ref, "num_batches_tracked") and ref.num_batches_tracked is not None and tname.startswith("Batch"):
                assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == it + 1
    if getattr(ref, "track_running_stats", False):
        ref.eval(); mine.eval()
        x0 = torch.randn(shape, generator=g).bfloat16().float()
        xr = x0.clone().requires_grad_(True); xm = x0.clone().to(dev).requires_grad_(True)
        gy = torch.randn(shape, generator=g).bfloat16().float()
        yr = aref(ref(xr)); yr.backward(gy)
        ym = seq(xm); ym.backward(gy.to(dev))
        assert float((ym.detach().cpu() - yr.detach()).abs().max()) <= 2e-2 * max(1.0, float(yr.abs().max()))
        assert float((xm.grad.cpu() - xr.grad).abs().max()) <= 2e-2 * max(1.0, float(xr.grad.abs().max()))
        if kw.get("affine", True):
            assert float((mine.weight.grad.cpu() - ref.weight.grad).abs().max()) <= 1.5e-2 * max(1.0, float(ref.weight.grad.abs().max()))
            assert float((mine.bias.grad.cpu() - ref.bias.grad).abs().max()) <= 1.5e-2 * max(1.0, float(ref.bias.grad.abs().max()))


def test_gan_losses_vanilla_and_wgangp_on_the_fused_kernel():
    """GANLoss "vanilla" (BCEWithLogitsLoss against the constant label) and "wgangp" (-/+ mean) as rg_loss_fwd_bwd kinds 3 / 4: value and
    gradient against torch on the same fp32 logits, incl. large |x| (the stable log(1 + exp(-|x|)) form) and smoothed labels."""
    from raygun_amd.torch.losses.GANLoss import GANLoss
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(9)
    x0 = (torch.randn(2, 1, 7, 6, 5, generator=g) * 4).to(dev)
    x0.view(-1)[:4] = torch.tensor([60.0, -60.0, 0.0, 1e-4], device=dev)
    for mode, labels in (("vanilla", (1.0, 0.0)), ("vanilla", (0.9, 0.1)), ("wgangp", (1.0, 0.0))):
        for real in (True, False):
            x = x0.clone().requires_grad_(True); xr = x0.clone().requires_grad_(True)
            crit = GANLoss(mode, *labels).to(dev)
            loss = crit(x, real)
            if mode == "vanilla":
                ref = F.binary_cross_entropy_with_logits(xr, torch.full_like(xr, labels[0] if real else labels[1]))
            else:
                ref = -xr.mean() if real else xr.mean()
            (loss * 3).backward(); (ref * 3).backward()
            assert abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), (mode, real)
            assert torch.allclose(x.grad, xr.grad, rtol=1e-5, atol=1e-8), (mode, real)


def test_adam_matches_torch():
    from raygun_amd import ops
    dev = torch.device("cuda")
    p0 = torch.from_numpy(synth.normal((1000,), 91))
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2, betas=(0.5, 0.999), weight_decay=0.01)
    p = p0.clone().to(dev); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for step in range(1, 4):
        g = torch.from_numpy(synth.normal((1000,), 92 + step))
        ref.grad = g.clone(); opt.step()
        ops.adam_step_(p, g.to(dev), m, v, 1e-2, 0.5, 0.999, 1e-8, 0.01, step)
    assert float((p.cpu() - ref.detach()).abs().max()) < 1e-6


def test_cpu_tensor_raises():
    from raygun_amd import ops
    with pytest.raises(RuntimeError):
        ops.conv3d(torch.zeros(1, 1, 4, 4, 4), torch.zeros(1, 1, 3, 3, 3))


def test_shape_errors_raise():
    from raygun_amd import ops
    dev = torch.device("cuda")
    with pytest.raises(RuntimeError):
        ops.conv3d(torch.zeros(1, 2, 4, 4, 4, device=dev), torch.zeros(1, 3, 3, 3, 3, device=dev))
    with pytest.raises(RuntimeError):
        ops.conv3d(torch.zeros(1, 1, 2, 2, 2, device=dev), torch.zeros(1, 1, 3, 3, 3, device=dev))


@pytest.mark.parametrize("cin,cout", [(72, 40), (432, 144), (12, 12)])
def test_fused_adam_refreshes_packed_forward_operand(cin, cout):
    """FusedAdam rewrites the bf16 forward operand of a channels-last conv weight in its own pass (rg_adam_step_pack): the
    refreshed buffer must equal a fresh rg_pack_weights of the updated master bit for bit, and the updated master must equal
    the plain fused step's."""
    from raygun_amd import hipnn, ops
    from raygun_amd.torch.optimizers import FusedAdam
    if not hipnn.CL_WEIGHTS:
        pytest.skip("channels-last master weights are disabled")
    dev = torch.device("cuda")
    conv = hipnn.Conv3d(cin, cout, 3).to(dev)
    x = rnd((1, cin, 5, 6, 7), 11).to(dev)
    opt = FusedAdam(conv.parameters(), lr=1e-2)
    for it in range(2):
        opt.zero_grad()
        conv(x).float().pow(2).mean().backward()      # forward packs the operand (iteration 0) / reuses the refreshed one
        before = conv.weight.detach().clone()
        opt.step()
        cache = conv.weight._rg_pack_cache
        assert cache["ver"] == (conv.weight._version, conv.weight.data_ptr()) and "fwd_pack" in cache
        fresh = ops.pack_weights(conv.weight.detach().clone(memory_format=torch.preserve_format), swap=0)
        assert torch.equal(cache[(0, (0, 0, 0), (1, 1, 1), None)].view(torch.int16), fresh.view(torch.int16))
        assert not torch.equal(before, conv.weight.detach())


@pytest.mark.gpu
def test_fused_adam_under_the_backward_pass_is_bit_identical():
    """FusedAdam.overlap_backward(): every update is enqueued from the parameter's post-accumulate-grad hook on a side stream (big tensors at once,
    small ones collected) and step() is the join — parameters, moments, step counts and the refreshed packed operands after 3 steps must equal the
    plain order bit for bit (same kernel, same arithmetic), including a parameter that gets no gradient and a second parameter group."""
    from raygun_amd import hipnn
    from raygun_amd.torch.optimizers import FusedAdam
    dev = torch.device("cuda")

    def make():
        torch.manual_seed(3)
        net = torch.nn.Sequential(hipnn.Conv3d(4, 96, 3), torch.nn.ReLU(), hipnn.Conv3d(96, 160, 3), torch.nn.ReLU(), hipnn.Conv3d(160, 8, 1)).to(dev)
        unused = torch.nn.Parameter(torch.ones(5, device=dev))
        opt = FusedAdam([{"params": list(net[:4].parameters())}, {"params": list(net[4:].parameters()) + [unused], "lr": 3e-3}], lr=1e-2, weight_decay=1e-3)
        return net, opt, unused
    x = rnd((1, 4, 12, 12, 12), 21).to(dev)
    nets = []
    for overlap in (False, True):
        net, opt, unused = make()
        if overlap:
            opt.overlap_backward(flush_numel=100000)     # the 96 -> 160 weight (414 720 elements) goes at once, the rest is collected
        for it in range(3):
            opt.zero_grad()
            net(x).float().pow(2).mean().backward()
            opt.step()
        torch.cuda.synchronize()
        nets.append((net, opt, unused))
    (na, oa, ua), (nb, ob, ub) = nets
    assert torch.equal(ua, ub) and len(oa.state[ua]) == 0 and len(ob.state[ub]) == 0
    for pa, pb in zip(na.parameters(), nb.parameters()):
        assert torch.equal(pa, pb)
        for k in ("exp_avg", "exp_avg_sq", "step"):
            assert torch.equal(oa.state[pa][k], ob.state[pb][k]), k
        ca, cb = getattr(pa, "_rg_pack_cache", None), getattr(pb, "_rg_pack_cache", None)
        if ca is not None and "fwd_pack" in ca:
            assert cb is not None and cb["ver"] == (pb._version, pb.data_ptr())
            assert torch.equal(ca["fwd_pack"][0].view(torch.int16), cb["fwd_pack"][0].view(torch.int16))
    ob.overlap_backward(False)
    ob.zero_grad(); nb(x).float().pow(2).mean().backward(); before = [p.detach().clone() for p in nb.parameters()]
    assert all(torch.equal(a, b) for a, b in zip(before, nb.parameters())), "hooks removed: nothing moves before step()"


@pytest.mark.gpu
@pytest.mark.parametrize("kw,stride,w", [(4, 2, 34), (7, 1, 20), (3, 2, 17)])
def test_fold_w_taps_to_channels_and_its_deterministic_backward(kw, stride, w):
    """ops._fold_w (the W taps of a one-channel input as channels: PatchGAN's first layer, the generator stems) against torch's as_strided on the CPU,
    forward and input gradient (the overlap-add of the taps, in tap order, fp32); two backward passes give the same bits."""
    from raygun_amd import ops
    dev = torch.device("cuda")
    x0 = bf(rnd((2, 1, 5, 6, w), 401))
    wo = (w - kw) // stride + 1
    xr = x0.clone().requires_grad_()
    ref = torch.as_strided(xr, (2, kw, 5, 6, wo), (xr.stride(0), 1, xr.stride(2), xr.stride(3), stride), 0)
    go = bf(rnd((2, kw, 5, 6, wo), 402))
    (gr,) = torch.autograd.grad(ref, xr, go)
    outs = []
    for _ in range(2):
        xd = x0.clone().to(dev).requires_grad_()
        y = ops._fold_w(xd, kw, stride, (0, 0, 0))
        assert tuple(y.shape) == (2, kw, 5, 6, wo) and torch.equal(y.float().cpu(), ref.detach())
        (gx,) = torch.autograd.grad(y, xd, go.to(dev).to(y.dtype))
        outs.append(gx.float().cpu())
        assert float((outs[-1] - gr).abs().max()) <= 2 ** -7 * float(gr.abs().max())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.gpu
def test_graphed_step_replays_equal_eager_steps():
    """raygun_amd.graphs.GraphedStep on a small deterministic network (convolutions incl. a forced stream-K launch, ReLU, an MSE-type loss, backward,
    FusedAdam(capturable=True)): two eager warm-up steps, one captured, three replays, on inputs rewritten in place — parameters, moments and step
    counts against the same six steps run eagerly with the ordinary optimiser.  The only arithmetic difference is the bias correction 1 - beta^t
    (float64 on the device instead of powf on the host, a last-bit difference that six Adam steps carry into the 5th digit of the losses)."""
    from raygun_amd import hipnn, ops
    from raygun_amd.graphs import GraphedStep
    from raygun_amd.torch.optimizers import FusedAdam
    dev = torch.device("cuda")
    xs = [rnd((1, 4, 14, 14, 14), 300 + i).to(dev) for i in range(6)]

    def run(graphed):
        torch.manual_seed(7)
        net = torch.nn.Sequential(hipnn.Conv3d(4, 128, 3), torch.nn.ReLU(), hipnn.Conv3d(128, 256, 3), torch.nn.ReLU(), hipnn.Conv3d(256, 8, 1)).to(dev)
        opt = FusedAdam(net.parameters(), lr=1e-3, betas=(0.8, 0.99), capturable=graphed)
        x = torch.zeros_like(xs[0])

        def body():
            opt.zero_grad()
            loss = net(x).float().pow(2).mean()
            loss.backward()
            opt.step()
            return loss
        gs = GraphedStep(body, optimizers=[opt], warmup=2) if graphed else None
        losses = []
        with ops.option("RG_GEMM5", 1), ops.option("RG_SK", 1):       # the 128 -> 256 layer runs igemm5 stream-K: flags lowered by their readers, replay after replay
            for xi in xs:
                x.copy_(xi)
                losses.append(float(gs() if graphed else body()))
        if graphed:
            assert gs.graph is not None or gs.eager_only
        return losses, [p.detach().clone() for p in net.parameters()], [opt.state[p] for p in net.parameters()]
    la, pa, sa = run(False)
    lb, pb, sb = run(True)
    assert all(abs(a - b) <= 2e-4 * abs(a) for a, b in zip(la, lb)), (la, lb)
    for a, b, ta, tb in zip(pa, pb, sa, sb):
        assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(a.abs().max()))
        assert float(ta["step"]) == float(tb["step"]) == 6.0
        assert float((ta["exp_avg"] - tb["exp_avg"]).abs().max()) <= 1e-3 * max(1e-6, float(ta["exp_avg"].abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("gs,fs,target", [((9, 9, 9), (20, 20, 20), (18, 18, 18)), ((9, 10, 11), (22, 21, 23), (16, 18, 20)),
                                           ((8, 9, 8), (16, 17, 15), (13, 15, 11)), ((10, 9, 9), (24, 18, 20), (15, 17, 13))])
def test_upcat_conv3d_equals_upsample_crop_concat_conv(gs, fs, target):
    """ops.upcat_conv3d (27 -> 8 taps on the upsampled channels, depth-to-space) against the plain composition it replaces,
    evaluated by torch in fp32 ON THE CPU (the oracle side: no GPU library in the reference path) on bf16-exact operands:
    output and all four gradients, for even and odd crop offsets."""
    from raygun_amd import ops
    dev = torch.device("cuda")
    cf, cg, cout = 16, 64, 24
    torch.manual_seed(0)
    f0 = bf(torch.randn(1, cf, *fs)); g0 = bf(torch.randn(1, cg, *gs))
    w0 = bf(torch.randn(cout, cf + cg, 3, 3, 3) * 0.05); b0 = torch.randn(cout)
    f, g, w, b = (t.clone().requires_grad_() for t in (f0, g0, w0, b0))
    up = F.interpolate(g, scale_factor=2, mode="nearest")
    cc = lambda t: t[..., (t.shape[2] - target[0]) // 2:(t.shape[2] - target[0]) // 2 + target[0],  # noqa: E731
                     (t.shape[3] - target[1]) // 2:(t.shape[3] - target[1]) // 2 + target[1],
                     (t.shape[4] - target[2]) // 2:(t.shape[4] - target[2]) // 2 + target[2]]
    ref = torch.tanh(F.conv3d(torch.cat([cc(f), cc(up)], 1), w, b))
    go = bf(torch.randn(ref.shape))
    rg = torch.autograd.grad(ref, (f, g, w, b), go)
    fd, gd, bd = (t.clone().to(dev).requires_grad_() for t in (f0, g0, b0))
    wd = w0.clone().to(dev)
    if gs[0] % 2 == 0:  # half of the cases: channels-last master weight (the float4 forms of the sub-pixel weight kernels)
        wd = wd.contiguous(memory_format=torch.channels_last_3d)
    wd.requires_grad_()
    got = ops.upcat_conv3d(fd, gd, target, wd, bd, "tanh")
    assert tuple(got.shape) == tuple(ref.shape)
    gg = torch.autograd.grad(got, (fd, gd, wd, bd), go.to(dev).to(got.dtype))

    def close(a, r, tol):
        a, r = a.float().cpu(), r.float()
        assert (a - r).abs().max().item() <= tol * r.abs().max().item() + 1e-6, ((a - r).abs().max().item(), r.abs().max().item())
    close(got, ref, 3e-2)  # pre-activations of magnitude ~10 rounded to bf16 twice
    for a, r in zip(gg, rg):
        close(a, r, 2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,mid,P,k,pad,sp,act", [(16, 16, 1, 7, 3, (10, 11, 12), "tanh"), (64, 64, 1, 7, 0, (12, 12, 14), "none"),
                                                    (24, 32, 2, 3, 1, (6, 7, 8), "none"), (8, 8, 1, 7, 3, (9, 9, 9), "sigmoid")])
def test_conv3d_linear_pair_equals_two_layers(cin, mid, P, k, pad, sp, act):
    """ops.conv3d_linear_pair — the generator tail Conv3d(ngf, ngf, 7) + Conv3d(ngf, output_nc, 1) evaluated as ONE composed
    convolution (rg_diag_sum / rg_diag_spread for P == 1, the one-hot (1,1,KW) gather otherwise) — against torch's
    two-layer evaluation in fp32 on the CPU: output and the gradients of x, w_a, b_a, w_b, b_b."""
    from raygun_amd import ops
    dev = torch.device("cuda")
    torch.manual_seed(0)
    x = bf(torch.randn(2, cin, *sp)); wa = bf(torch.randn(mid, cin, k, k, k) * (1.0 / (cin * k ** 3) ** 0.5)); ba = torch.randn(mid) * 0.1
    wb = bf(torch.randn(P, mid, 1, 1, 1) * (1.0 / mid ** 0.5)); bb = torch.randn(P) * 0.1
    afn = {"none": lambda t: t, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act]
    ref_in = [t.clone().requires_grad_(True) for t in (x, wa, ba, wb, bb)]
    yr = afn(F.conv3d(F.conv3d(ref_in[0], ref_in[1], ref_in[2], 1, pad), ref_in[3], ref_in[4]))
    g = bf(torch.randn(yr.shape))
    yr.backward(g)
    my_in = [t.clone().to(dev).requires_grad_(True) for t in (x, wa, ba, wb, bb)]
    y = ops.leave(ops.conv3d_linear_pair(my_in[0], my_in[1], my_in[2], my_in[3], my_in[4], pad, act, 0.0))
    assert tuple(y.shape) == tuple(yr.shape)
    y.backward(g.to(dev).to(y.dtype))
    errs = {"y": relerr(y, yr)}
    for name, a, b in zip("x wa ba wb bb".split(), my_in, ref_in):
        errs["g_" + name] = relerr(a.grad, b.grad)
    check(errs, 2e-2)  # the composed weight W' = w_b . w_a is rounded to bf16 once more than the two-layer evaluation's operands


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,kw,sp,pad,act", [(49, 64, 7, (34, 36, 70), 0, "relu"), (64, 49, 7, (33, 34, 134), 0, "none"),
                                                    (49, 64, 7, (32, 33, 128), 3, "none"), (64, 64, 5, (40, 30, 50), 2, "leaky")])
def test_line_convolution_shapes(cin, cout, kw, sp, pad, act):
    """The folded 7^3 stem / composed tail shapes: a (kw,1,1) convolution run D<->W exchanged (taps along the innermost axis), 49 or 64 channels in,
    <= 64 out, >= 1024 lines — forward (bias, activation, zero padding along the tap axis) and the input gradient (the tap-reversed direction)
    against torch fp32 on the CPU.  (Round 3 carried an opt-in line kernel, conv6.hip, for these shapes: a measured step-level tie, removed in round 4.)"""
    from raygun_amd import ops
    dev = torch.device("cuda")
    torch.manual_seed(11)
    d, h, w = sp                                         # the op below convolves along D: conv_kd_as_kw swaps D and W
    x0 = bf(torch.randn(1, cin, w, h, d)); w0 = bf(torch.randn(cout, cin, kw, 1, 1) * (1.0 / (cin * kw) ** 0.5)); b0 = torch.randn(cout) * 0.1
    afn = {"none": lambda t: t, "relu": torch.relu, "leaky": lambda t: F.leaky_relu(t, 0.2)}[act]
    x = x0.clone().requires_grad_()
    ref = afn(F.conv3d(x, w0, b0, 1, (pad, 0, 0)))
    go = bf(torch.randn(ref.shape))
    (gref,) = torch.autograd.grad(ref, x, go)
    xd = x0.clone().to(dev).requires_grad_()
    # the folded stem / tail tensors live D<->W exchanged in memory (ops.cl_empty_dw): the tap axis is the innermost spatial axis
    xcl = ops.enter(xd.transpose(2, 4).contiguous())
    xin = (ops.pad3d(xcl, (0, 0, pad), "zeros") if pad else xcl).transpose(2, 4)
    ycl = ops.conv_kd_as_kw(xin, w0.to(dev), b0.to(dev), act, 0.2)
    y = ops.leave(ycl)
    gdw = ops.enter(go.to(dev).transpose(2, 4).contiguous()).transpose(2, 4)     # the incoming gradient in the same exchanged layout
    (gx,) = torch.autograd.grad(ycl, xd, gdw)
    assert tuple(y.shape) == tuple(ref.shape)
    assert (y.detach().cpu() - ref.detach()).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    assert (gx.cpu() - gref).abs().max().item() <= 2e-2 * gref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("sgn", [1, -1])
def test_diag_spread_sum_on_exchanged_views_are_exact(sgn):
    """The (kd, kh) fold of the 7^3 stem / tail (ops.conv3d, ops.conv3d_linear_pair) calls the two maps on D<->W transposed views and asks for
    D<->W exchanged outputs (strided descriptors on both sides of the kernels).  Exact against numpy indexing, both signs.  (A kernel that walks the
    passive axis and writes whole records back to back was tried for this layout: 0.6 ms per CycleGAN step SLOWER than the tiled one — removed.)"""
    from raygun_amd import ops
    dev = torch.device("cuda")
    kh, kw = 3, 5
    x = rnd((2, 1, 11, 9, 150), 33)                       # logical (n, 1, D, H, W): W will be the passive axis
    xT = ops.to_cl(x.to(dev)).transpose(2, 4)             # view (n, 1, W, H, D): its D (= W above) is the memory-fastest axis
    if sgn > 0:
        oh, ow = 9 - kh + 1, 11 - kw + 1
    else:
        oh, ow = 9 + kh - 1, 11 + kw - 1
    yT = ops._DiagSpread.apply(xT, kh, kw, sgn, oh, ow, True)
    assert yT.transpose(2, 4).is_contiguous(memory_format=torch.channels_last_3d) or ops.is_cl(yT.transpose(2, 4))
    xv = x[:, 0].permute(0, 3, 2, 1)                      # (n, W, H, D) as the kernel sees it
    want = torch.zeros(2, kh * kw, 150, oh, ow)
    for a in range(kh):
        for b in range(kw):
            for hh in range(oh):
                for ww in range(ow):
                    sh, sw = hh + sgn * a, ww + sgn * b
                    if 0 <= sh < 9 and 0 <= sw < 11:
                        want[:, a * kw + b, :, hh, ww] = xv[:, :, sh, sw]
    assert torch.equal(yT.float().cpu(), want)
    z = rnd((2, kh * kw, 11, 9, 150), 34)
    zT = ops.to_cl(z.to(dev)).transpose(2, 4)
    sT = ops._DiagSum.apply(zT, kh, kw, sgn, oh, ow, None, True)
    zv = z.permute(0, 1, 4, 3, 2)
    acc = torch.zeros(2, 150, oh, ow)
    for a in range(kh):
        for b in range(kw):
            for hh in range(oh):
                for ww in range(ow):
                    sh, sw = hh + sgn * a, ww + sgn * b
                    if 0 <= sh < 9 and 0 <= sw < 11:
                        acc[:, :, hh, ww] += zv[:, a * kw + b, :, sh, sw]
    got = sT.float().cpu()[:, 0]
    assert float((got - acc).abs().max()) <= 2e-2 * max(1.0, float(acc.abs().max()))   # (the sum is rounded to bf16 once)


@pytest.mark.gpu
def test_diag_spread_sum_are_exact_adjoint_gathers():
    """rg_diag_spread / rg_diag_sum (in-plane taps <-> channels) are pure data movement + an fp32 sum: exact against numpy
    indexing on bf16-exact data, both signs."""
    from raygun_amd import ops
    dev = torch.device("cuda")
    kh, kw = 3, 5
    x = rnd((2, 1, 4, 9, 11), 31)
    oh, ow = 9 - kh + 1, 11 - kw + 1
    y = ops._DiagSpread.apply(ops.to_cl(x.to(dev)), kh, kw, 1, oh, ow)
    want = torch.stack([x[:, 0, :, a:a + oh, b:b + ow] for a in range(kh) for b in range(kw)], 1)
    assert torch.equal(y.float().cpu(), want)
    z = rnd((2, kh * kw, 4, 9, 11), 32)
    s = ops._DiagSum.apply(ops.to_cl(z.to(dev)), kh, kw, 1, oh, ow, None)
    want = sum(z[:, a * kw + b, :, a:a + oh, b:b + ow] for a in range(kh) for b in range(kw))[:, None]
    assert relerr(s, want) <= 2 ** -8


@pytest.mark.gpu
def test_fork_and_crop_gradients():
    """ops.fork / ops.crop (skip connections): forward are views, backward = one rg_add / rg_add_pad pass; against autograd on
    plain slicing."""
    from raygun_amd import ops
    dev = torch.device("cuda")
    torch.manual_seed(1)
    x0 = torch.randn(2, 12, 9, 10, 11, device=dev).bfloat16().float()
    xa = ops.to_cl(x0).requires_grad_()
    a, b = ops.fork(xa)
    c = ops.crop(b, (1, 2, 0), (6, 5, 9))
    assert c.data_ptr() != 0 and tuple(c.shape) == (2, 12, 6, 5, 9) and torch.equal(c.float(), x0[:, :, 1:7, 2:7, 0:9])
    ga = torch.randn(2, 12, 9, 10, 11, device=dev).bfloat16()
    gc = torch.randn(2, 12, 6, 5, 9, device=dev).bfloat16()
    (gx,) = torch.autograd.grad((a, c), xa, (ops.to_cl(ga), ops.to_cl(gc)))
    ref = ga.float().clone()
    ref[:, :, 1:7, 2:7, 0:9] += gc.float()
    assert (gx.float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()   # one bf16 rounding of the sum
    # crop alone (its gradient is the zero-padded tensor, channels-last), centre_crop through the same op
    xb = ops.to_cl(x0).requires_grad_()
    (gz,) = torch.autograd.grad(ops.center_crop(xb, (5, 6, 7)), xb, ops.to_cl(torch.ones(2, 12, 5, 6, 7, device=dev)))
    want = torch.zeros_like(x0); want[:, :, 2:7, 2:8, 2:9] = 1
    assert ops.is_cl(gz) and torch.equal(gz.float(), want)


@pytest.mark.gpu
def test_wide_transposed_conv_subpixel_path():
    """Stride-2 / k3 / pad-1 transposed convolutions with >= 192 input channels run as one 8*Cout-channel convolution +
    depth-to-space (ops._scatter_conv fast path): ConvTranspose3d forward and the input gradient of a strided Conv3d, against
    torch fp32 on the CPU."""
    from raygun_amd import ops
    dev = torch.device("cuda")
    torch.manual_seed(3)
    x0 = bf(torch.randn(1, 192, 5, 6, 7)); w0 = bf(torch.randn(192, 24, 3, 3, 3) * 0.05); b0 = torch.randn(24)
    x, w = x0.clone().requires_grad_(), w0.clone().requires_grad_()
    ref = torch.tanh(F.conv_transpose3d(x, w, b0, stride=2, padding=1, output_padding=1))
    xd, wd = x0.clone().to(dev).requires_grad_(), w0.clone().to(dev).requires_grad_()
    got = ops.conv_transpose3d(xd, wd, b0.to(dev), stride=2, padding=1, output_padding=1, act="tanh")
    assert tuple(got.shape) == (1, 24, 10, 12, 14)
    assert (got.float().cpu() - ref).abs().max().item() <= 2e-2
    go = bf(torch.randn(ref.shape))
    for a, r in zip(torch.autograd.grad(got, (xd, wd), go.to(dev).to(got.dtype)), torch.autograd.grad(ref, (x, w), go)):
        assert (a.float().cpu() - r).abs().max().item() <= 2e-2 * r.abs().max().item() + 1e-6
    # strided conv: its backward-data is the same scatter with 192 source channels
    xs0 = bf(torch.randn(1, 16, 10, 12, 8)); ws = bf(torch.randn(192, 16, 3, 3, 3) * 0.05); gs = bf(torch.randn(1, 192, 5, 6, 4))
    xs = xs0.clone().requires_grad_()
    (rg,) = torch.autograd.grad(F.conv3d(xs, ws, None, stride=2, padding=1), xs, gs)
    xsd = xs0.clone().to(dev).requires_grad_()
    yd = ops.conv3d(xsd, ws.to(dev), None, stride=2, padding=1)
    (gg,) = torch.autograd.grad(yd, xsd, gs.to(dev).to(yd.dtype))
    assert (gg.float().cpu() - rg).abs().max().item() <= 2e-2 * rg.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,sp,act", [(128, 64, (30, 32, 24), "relu"), (256, 128, (24, 24, 20), "none"), (128, 32, (36, 40, 32), "leaky"),
                                             (128, 64, (6, 5, 4), "none")])   # (>= 180 tiles of 256 x 256 for the first three: igemm5 without a K split)
def test_transposed_conv_subpixel_with_depth_to_space_epilogue(cin, cout, sp, act, monkeypatch):
    """rg_conv3d_igemm_d2s: the stride-2 / k3 / pad-1 transposed convolution as ONE 8*Cout-column convolution whose epilogue writes every column
    block to its output parity (no intermediate, no shuffle pass) — ConvTranspose3d forward (+ bias, + activation) and the input gradient of the
    stride-2 Conv3d, against torch fp32 on the CPU and bit for bit against the two-pass form (RAYGUN_AMD_D2S_EPILOGUE=0) where that one exists;
    problems whose grid igemm5 would have to K-split (the last case) fall back by themselves."""
    from raygun_amd import ops
    dev = torch.device("cuda")
    torch.manual_seed(5)
    x0 = bf(torch.randn(1, cin, *sp)); w0 = bf(torch.randn(cin, cout, 3, 3, 3) * (1.0 / (cin * 27) ** 0.5)); b0 = torch.randn(cout) * 0.1
    afn = {"none": lambda t: t, "relu": torch.relu, "leaky": lambda t: F.leaky_relu(t, 0.2)}[act]
    x, w = x0.clone().requires_grad_(), w0.clone().requires_grad_()
    ref = afn(F.conv_transpose3d(x, w, b0, stride=2, padding=1, output_padding=1))
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("RAYGUN_AMD_D2S_EPILOGUE", flag)
        xd, wd = x0.clone().to(dev).requires_grad_(), w0.clone().to(dev).requires_grad_()
        ops.profile_start()
        with ops.option("RG_SK", -1):    # (the two-pass form has a workspace and may run stream-K — another summation order; the fused one never does)
            got = ops.conv_transpose3d(xd, wd, b0.to(dev), stride=2, padding=1, output_padding=1, act=act, slope=0.2)
        names = list(ops.profile_stop())
        assert tuple(got.shape) == tuple(ref.shape)
        assert (got.float().cpu() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item()), flag
        go = bf(torch.randn(ref.shape))
        for a, r in zip(torch.autograd.grad(got, (xd, wd), go.to(dev).to(got.dtype)), torch.autograd.grad(ref, (x, w), go, retain_graph=True)):
            assert (a.float().cpu() - r).abs().max().item() <= 2e-2 * r.abs().max().item() + 1e-6
        outs[flag] = (got.detach().float().cpu(), names)
    fused = any("+d2s" in n for n in outs["1"][1])
    assert fused == (sp != (6, 5, 4)), outs["1"][1]              # big enough grids take the fused epilogue
    assert not any("+d2s" in n for n in outs["0"][1])
    if fused and cin >= 192:                                     # same GEMM, same rounding: the shuffle pass only moves data
        assert torch.equal(outs["1"][0], outs["0"][0])
    # the backward-data pass of the stride-2 convolution is the same scatter
    xs0 = bf(torch.randn(1, cout, *[2 * v for v in sp])); ws = bf(torch.randn(cin, cout, 3, 3, 3) * 0.05); gs = bf(torch.randn(1, cin, *sp))
    xs = xs0.clone().requires_grad_()
    (rg,) = torch.autograd.grad(F.conv3d(xs, ws, None, stride=2, padding=1), xs, gs)
    monkeypatch.setenv("RAYGUN_AMD_D2S_EPILOGUE", "1")
    xsd = xs0.clone().to(dev).requires_grad_()
    yd = ops.conv3d(xsd, ws.to(dev), None, stride=2, padding=1)
    (gg,) = torch.autograd.grad(yd, xsd, gs.to(dev).to(yd.dtype))
    assert (gg.float().cpu() - rg).abs().max().item() <= 2e-2 * rg.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("cout", [144, 128])
def test_igemm4_192_row_tiles(cout):
    """The 192-row tile (FM = 3) of igemm4 is only picked for large grids; force it (RG_FM=3) on a small problem: forward and
    backward-data against torch fp32 on the CPU."""
    from raygun_amd import ops
    with ops.option("RG_FM", 3):
        _igemm4_192_row_tiles(cout, ops)


def _igemm4_192_row_tiles(cout, ops):
    dev = torch.device("cuda")
    torch.manual_seed(5)
    x0 = bf(torch.randn(1, 64, 9, 10, 11)); w = bf(torch.randn(cout, 64, 3, 3, 3) * 0.05); b = torch.randn(cout)
    x = x0.clone().requires_grad_()
    ref = F.conv3d(x, w, b)
    xd = x0.clone().to(dev).requires_grad_()
    got = ops.conv3d(xd, w.to(dev), b.to(dev))
    assert (got.float().cpu() - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()
    go = bf(torch.randn(ref.shape))
    (rg,) = torch.autograd.grad(ref, x, go)
    (gg,) = torch.autograd.grad(got, xd, go.to(dev).to(got.dtype))
    assert (gg.float().cpu() - rg).abs().max().item() <= 1.5e-2 * rg.abs().max().item()


@pytest.mark.gpu
def test_packed_weight_cache_invalidation():
    """The packed bf16 operands are cached per (version, data_ptr).  `.data` writes do not bump the version: init_weights
    (which writes through .data, networks/utils.py:55-73) and ops.invalidate_packed must make the next forward see the new
    weights; ordinary in-place ops under no_grad bump the version by themselves."""
    from raygun_amd import hipnn, ops
    from raygun_amd.torch.networks.utils import init_weights
    dev = torch.device("cuda")
    conv = hipnn.Conv3d(8, 16, 3).to(dev)
    x = rnd((1, 8, 6, 6, 6), 3).to(dev)
    y0 = conv(x).clone()
    init_weights(conv, "normal", 0.5)
    y1 = conv(x).clone()
    ref = F.conv3d(x.cpu(), bf(conv.weight.detach().cpu()), conv.bias.detach().cpu())
    assert relerr(y1, ref) <= TOL and not torch.allclose(y0, y1)
    conv.weight.data.mul_(2.0)            # invisible to the version counter ...
    ops.invalidate_packed(conv)           # ... so the writer says so
    assert relerr(conv(x), F.conv3d(x.cpu(), bf(conv.weight.detach().cpu()), conv.bias.detach().cpu())) <= TOL
    with torch.no_grad():
        conv.weight.mul_(0.5)             # bumps the version: nothing to do
    assert relerr(conv(x), F.conv3d(x.cpu(), bf(conv.weight.detach().cpu()), conv.bias.detach().cpu())) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,sp", [(64, 128, (8, 8, 12)), (16, 24, (6, 10, 8)), (128, 64, (4, 6, 8))])
def test_strided_conv_weight_gradient_through_w_pairs(cin, cout, sp, monkeypatch):
    """Stride-2 / k3 / pad-1 convolutions and transposed convolutions take their weight gradient through ops._wgrad_wpairs (the
    strided operand viewed as W-pairs of 2C channels, stride-1 kernels): against torch fp32 on the CPU, and against the generic
    strided kernel (RAYGUN_AMD_WPAIR=0)."""
    from raygun_amd import ops
    dev = torch.device("cuda")
    x0 = rnd((2, cin) + sp, 41); w0 = rnd((cout, cin, 3, 3, 3), 42, -0.1, 0.1)
    xr, wr = x0.clone().requires_grad_(), w0.clone().requires_grad_()
    yr = F.conv3d(xr, wr, None, stride=2, padding=1)
    go = rnd(tuple(yr.shape), 43)
    (gw_ref,) = torch.autograd.grad(yr, wr, go)
    got = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("RAYGUN_AMD_WPAIR", flag)
        xd, wd = x0.clone().to(dev).requires_grad_(), w0.clone().to(dev).requires_grad_()
        y = ops.conv3d(xd, wd, None, stride=2, padding=1)
        (got[flag],) = torch.autograd.grad(y, wd, go.to(dev).to(y.dtype))
        assert relerr(got[flag], gw_ref) <= TOL
    assert relerr(got["1"], got["0"].cpu()) <= 2e-3          # same bf16 operands, fp32 accumulation in a different order
    # transposed convolution (generator up path, networks/ResNet.py:325-333): weight (cin, cout, 3,3,3), output_padding 1
    wt0 = rnd((cin, cout, 3, 3, 3), 44, -0.1, 0.1)
    xr, wr = x0.clone().requires_grad_(), wt0.clone().requires_grad_()
    yr = F.conv_transpose3d(xr, wr, None, stride=2, padding=1, output_padding=1)
    go = rnd(tuple(yr.shape), 45)
    (gw_ref,) = torch.autograd.grad(yr, wr, go)
    for flag in ("1", "0"):
        monkeypatch.setenv("RAYGUN_AMD_WPAIR", flag)
        xd, wd = x0.clone().to(dev).requires_grad_(), wt0.clone().to(dev).requires_grad_()
        y = ops.conv_transpose3d(xd, wd, None, stride=2, padding=1, output_padding=1)
        (gw,) = torch.autograd.grad(y, wd, go.to(dev).to(y.dtype))
        assert relerr(gw, gw_ref) <= TOL, flag


@pytest.mark.gpu
@pytest.mark.parametrize("nd", [3, 2])
def test_instance_norm_running_stats_matches_torch(nd):
    """hipnn.InstanceNorm{2,3}d(track_running_stats=True, momentum) against torch.nn on the CPU: training mode (output, gradient,
    running mean / unbiased running var after two batches) and eval mode (normalisation by the running statistics, gradient)."""
    from raygun_amd import hipnn
    dev = torch.device("cuda")
    shape = (2, 16, 5, 6, 7) if nd == 3 else (2, 16, 9, 11)
    mine = getattr(hipnn, f"InstanceNorm{nd}d")(16, track_running_stats=True, momentum=0.1).to(dev)
    ref = getattr(torch.nn, f"InstanceNorm{nd}d")(16, track_running_stats=True, momentum=0.1)
    for it in range(2):
        x0 = rnd(shape, 60 + it, -2, 3)
        xr = x0.clone().requires_grad_(); xd = x0.clone().to(dev).requires_grad_()
        yr = ref(xr); y = mine(xd)
        go = rnd(shape, 70 + it)
        yr.backward(go); y.backward(go.to(dev))
        assert relerr(y, yr) <= TOL and relerr(xd.grad, xr.grad) <= 2 * TOL
    assert relerr(mine.running_mean, ref.running_mean) <= 1e-2 and relerr(mine.running_var, ref.running_var) <= 1e-2
    assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked)
    mine.eval(); ref.eval()
    ref.running_mean.copy_(mine.running_mean.cpu()); ref.running_var.copy_(mine.running_var.cpu())
    x0 = rnd(shape, 80, -2, 3)
    xr = x0.clone().requires_grad_(); xd = x0.clone().to(dev).requires_grad_()
    yr = ref(xr); y = mine(xd)
    go = rnd(shape, 81)
    yr.backward(go); y.backward(go.to(dev))
    assert relerr(y, yr) <= TOL and relerr(xd.grad, xr.grad) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,sp,pad,act", [(12, 12, (36, 40, 44), 0, "relu"), (16, 9, (33, 35, 70), 1, "sigmoid"), (10, 16, (40, 37, 41), 0, "none")])
def test_halo16_thin_conv_forward_and_backward_data(cin, cout, sp, pad, act):
    """conv5.hip halo16_kernel (input halo tile in LDS, weights in registers: the 12 -> 12 U-Net levels) — forward, fused bias /
    activation, zero padding, ragged tile edges and the backward-data pass (flipped taps) against torch fp32 on the CPU, and
    identical to the persistent thin kernel it replaces (RG_NO_HALO=1)."""
    from raygun_amd import _lib, ops
    dev = torch.device("cuda")
    x0 = rnd((2, cin) + sp, 51); w0 = rnd((cout, cin, 3, 3, 3), 52, -0.2, 0.2); b0 = rnd((cout,), 53)
    afn = {"none": lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid}[act]
    xr = x0.clone().requires_grad_()
    yr = afn(F.conv3d(xr, w0, b0, padding=pad))
    go = rnd(tuple(yr.shape), 54)
    (gxr,) = torch.autograd.grad(yr, xr, go)
    outs = {}
    for flag in ("", "1"):
        with ops.option("RG_NO_HALO", 1 if flag else 0):
            xd = x0.clone().to(dev).requires_grad_()
            y = ops.conv3d(xd, w0.to(dev), b0.to(dev), 1, pad, act)
            name = _lib.lib().rg_last_kernel_name().decode()
            assert ("halo16" in name) == (flag == ""), name
            (gx,) = torch.autograd.grad(y, xd, go.to(dev).to(y.dtype))
        assert relerr(y, yr) <= TOL and relerr(gx, gxr) <= TOL
        assert float(y[:, cout:].abs().max()) == 0 if y.shape[1] > cout else True
        outs[flag] = (y.float().cpu(), gx.float().cpu())
    assert relerr(outs[""][0], outs["1"][0]) <= 2 ** -7 and relerr(outs[""][1], outs["1"][1]) <= 2 ** -7


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,sp,pad,act", [(512, 144, (6, 6, 6), 0, "relu"), (256, 130, (5, 6, 7), 1, "none"), (72, 80, (8, 8, 9), 0, "leaky"),
                                                 (448, 136, (7, 7, 7), 1, "sigmoid")])
def test_split_k_small_grid_convolutions(cin, cout, sp, pad, act):
    """Launches whose tile grid fills at most half the chip split K over several workgroups per tile (igemm4 "+splitK": fp32 partial
    sums through the caller's workspace, bias / activation in the combine pass) — forward and backward-data against torch fp32 on
    the CPU, for channel-chunked and FLAT packings, ragged channel counts and zero padding."""
    from raygun_amd import _lib, ops
    dev = torch.device("cuda")
    x0 = rnd((1, cin) + sp, 61); w0 = rnd((cout, cin, 3, 3, 3), 62, -0.05, 0.05); b0 = rnd((cout,), 63)
    afn = {"none": lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "leaky": lambda t: F.leaky_relu(t, 0.2)}[act]
    xr = x0.clone().requires_grad_()
    yr = afn(F.conv3d(xr, w0, b0, padding=pad))
    go = rnd(tuple(yr.shape), 64)
    (gxr,) = torch.autograd.grad(yr, xr, go)
    xd = x0.clone().to(dev).requires_grad_()
    y = ops.conv3d(xd, w0.to(dev), b0.to(dev), 1, pad, act, 0.2)
    assert "+splitK" in _lib.lib().rg_last_kernel_name().decode(), _lib.lib().rg_last_kernel_name().decode()
    (gx,) = torch.autograd.grad(y, xd, go.to(dev).to(y.dtype))
    assert relerr(y, yr) <= TOL and relerr(gx, gxr) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,sp,act,cl", [(136, 144, (9, 10, 11), "relu", True), (192, 256, (8, 8, 8), "none", False),
                                                (130, 200, (7, 12, 9), "leaky", True)])
def test_winograd_wide_conv_forward_and_backward(cin, cout, sp, act, cl, monkeypatch):
    """Winograd F(2x2x2, 3x3x3) path of the widest stride-1 convolutions (csrc/winograd.hip: input / weight / output transforms around 64
    batched GEMMs; production threshold 1024 channels, lowered here): forward, backward-data (the mirrored operand) and the unchanged
    backward-weights against torch fp32 on the CPU — odd and even extents (cropped last tile), ragged channel counts, both weight memory
    formats, N = 2.  The transform-domain roundings roughly double the layer's bf16 error, still inside the policy's bound."""
    from raygun_amd import _lib, ops
    monkeypatch.setattr(ops, "WINO_MIN_CHANNELS", 64)
    dev = torch.device("cuda")
    x0 = rnd((2, cin) + sp, 71); w0 = rnd((cout, cin, 3, 3, 3), 72, -0.05, 0.05); b0 = rnd((cout,), 73)
    afn = {"none": lambda t: t, "relu": torch.relu, "leaky": lambda t: F.leaky_relu(t, 0.2)}[act]
    xr = x0.clone().requires_grad_(); wr = w0.clone().requires_grad_()
    ylin = F.conv3d(xr, wr, b0)
    yr = afn(ylin)
    go = rnd(tuple(yr.shape), 74)
    xd = x0.clone().to(dev).requires_grad_()
    wd = w0.clone().to(dev)
    if cl:
        wd = wd.contiguous(memory_format=torch.channels_last_3d)
    wd.requires_grad_()
    ops.profile_start()
    y = ops.conv3d(xd, wd, b0.to(dev), 1, 0, act, 0.2)
    gx, gw = torch.autograd.grad(y, (xd, wd), go.to(dev).to(y.dtype))
    prof = ops.profile_stop()
    assert sum(d["launches"] for n, d in prof.items() if n.startswith("winograd[")) == 2, prof
    # Winograd's error is absolute (cancellation in the output transform), so activations near zero can change sign against the
    # fp32 reference; the op differentiates through ITS output, so the reference gradient uses the same mask
    slope_ = {"none": 1.0, "relu": 0.0, "leaky": 0.2}[act]
    dact = torch.where(y.detach().float().cpu() > 0, torch.ones(()), torch.full((), slope_))
    gxr, gwr = torch.autograd.grad(ylin, (xr, wr), go * dact)
    e = {"y": relerr(y, yr), "gx": relerr(gx, gxr), "gw": relerr(gw, gwr)}
    l2 = float((y.float().cpu() - yr).norm() / yr.norm())
    assert l2 <= 1e-2, l2
    check(e)
    # the direct kernels on the same operands: the two paths agree to the transform-domain rounding
    monkeypatch.setenv("RAYGUN_AMD_WINOGRAD", "0")
    y2 = ops.conv3d(xd, wd, b0.to(dev), 1, 0, act, 0.2)
    assert relerr(y, y2) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,sp", [(12, 12, (67, 68, 69)), (1, 12, (66, 66, 66)), (10, 7, (66, 70, 66))])
def test_thin_weight_gradient_blocks_of_lines(cin, cout, sp):
    """wgrad_lines_kernel (<= 16 lanes on both sides, 3x3x3 valid, >= 2^18 output voxels: the 12 -> 12 levels and the folded 1 -> 12
    stem): units of 2 x 4 output lines whose 4 x 6 input lines are staged once — extents that leave partial line blocks and a partial
    32-voxel chunk, against torch fp32 on the CPU."""
    from raygun_amd import ops
    dev = torch.device("cuda")
    x0 = rnd((1, cin) + sp, 81); w0 = rnd((cout, cin, 3, 3, 3), 82, -0.2, 0.2); b0 = rnd((cout,), 83)
    xr = x0.clone().requires_grad_(cin > 1); wr = w0.clone().requires_grad_()
    yr = F.conv3d(xr, wr, b0)
    go = rnd(tuple(yr.shape), 84)
    (gwr,) = torch.autograd.grad(yr, (wr,), go)
    xd = x0.clone().to(dev).requires_grad_(cin > 1)
    wd = w0.clone().to(dev).requires_grad_()
    ops.profile_start()
    y = ops.conv3d(xd, wd, b0.to(dev), 1, 0, "none", 0.0)
    (gw,) = torch.autograd.grad(y, (wd,), go.to(dev).to(y.dtype))
    prof = ops.profile_stop()
    assert any(n.startswith("wgrad_lines_kernel") for n in prof), list(prof)
    assert relerr(gw, gwr) <= TOL, relerr(gw, gwr)


@pytest.mark.gpu
def test_winograd_operands_follow_the_optimizer(monkeypatch):
    """The transformed weights (U, Ut) are cached on the parameter like the packed operands: after a FusedAdam step (and after a plain
    in-place update through torch.optim) the next forward must see the NEW weights — compared with the direct kernels on the same
    parameter."""
    from raygun_amd import ops
    from raygun_amd.torch.optimizers import FusedAdam
    monkeypatch.setattr(ops, "WINO_MIN_CHANNELS", 64)
    dev = torch.device("cuda")
    x = rnd((1, 96, 8, 9, 10), 95).to(dev)
    for make_opt in (lambda p: FusedAdam([p], lr=5e-2), lambda p: torch.optim.SGD([p], lr=5.0)):
        w = torch.nn.Parameter(rnd((144, 96, 3, 3, 3), 96, -0.05, 0.05).to(dev).contiguous(memory_format=torch.channels_last_3d))
        opt = make_opt(w)
        y0 = ops.conv3d(x, w, None, 1, 0, "none", 0.0)
        y0.float().pow(2).mean().backward()
        opt.step()
        y1 = ops.conv3d(x, w, None, 1, 0, "none", 0.0)
        monkeypatch.setenv("RAYGUN_AMD_WINOGRAD", "0")
        y1d = ops.conv3d(x, w, None, 1, 0, "none", 0.0)
        monkeypatch.delenv("RAYGUN_AMD_WINOGRAD")
        assert relerr(y1, y0) > 5e-2, "the step did not change the output: stale transformed weights"
        assert relerr(y1, y1d) <= TOL, relerr(y1, y1d)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,sp,k,stride,pad,act,n", [
    (64, 144, (9, 10, 11), 3, 1, 0, "relu", 1),        # one half-empty 288-column tile, ragged row tile
    (184, 288, (12, 9, 10), 3, 1, 1, "leaky", 2),      # ragged channel chunk (184 = 2 chunks + 56 lanes), zero padding (CHECK), N = 2
    (432, 432, (8, 8, 9), 3, 1, 0, "none", 1),         # the U-Net's 432-channel layers: 7 chunks, 1.5 column tiles
    (128, 576, (10, 10, 10), 2, 1, 0, "relu", 1),      # 2^3 taps (the sub-pixel layers)
    (64, 288, (13, 12, 14), 3, 2, 1, "sigmoid", 1),    # strided rows
    (128, 256, (12, 10, 11), 3, 1, 0, "relu", 1),      # FN = 8: 256-column tiles (the CycleGAN's ResNet blocks)
    (256, 512, (9, 9, 10), 3, 1, 1, "leaky", 1),       # FN = 8, two column tiles, zero padding
])
def test_igemm5_lds_dma_conv_forward_and_backward_data(cin, cout, sp, k, stride, pad, act, n):
    """gemm5.hip igemm5_kernel (256 x 288 tiles, both operands through LDS by LDS-DMA, staggered wave groups) forced on for
    shapes that exercise its row / channel / column tails, zero padding, strides, tap odometers over several chunks and the
    backward-data pass (flipped taps, full correlation): against torch fp32 on the CPU, and identical to what igemm4 computes."""
    from raygun_amd import _lib, ops
    dev = torch.device("cuda")
    x0 = rnd((n, cin) + sp, 91); w0 = rnd((cout, cin, k, k, k), 92, -0.05, 0.05); b0 = rnd((cout,), 93)
    afn = {"none": lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "leaky": lambda t: F.leaky_relu(t, 0.2)}[act]
    xr = x0.clone().requires_grad_()
    yr = afn(F.conv3d(xr, w0, b0, stride=stride, padding=pad))
    go = rnd(tuple(yr.shape), 94)
    (gxr,) = torch.autograd.grad(yr, xr, go)
    outs = {}
    for opt in (1, -1):
        with ops.option("RG_GEMM5", opt):
            xd = x0.clone().to(dev).requires_grad_()
            y = ops.conv3d(xd, w0.to(dev), b0.to(dev), stride, pad, act, 0.2)
            name = _lib.lib().rg_last_kernel_name().decode()
            assert ("igemm5" in name) == (opt == 1), name
            (gx,) = torch.autograd.grad(y, xd, go.to(dev).to(y.dtype))
            assert relerr(y, yr) <= TOL and relerr(gx, gxr) <= TOL, (opt, relerr(y, yr), relerr(gx, gxr))
            outs[opt] = (y.float().cpu(), gx.float().cpu())
    assert relerr(outs[1][0], outs[-1][0]) <= 2 ** -7 and relerr(outs[1][1], outs[-1][1]) <= 2 ** -7


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,sp,k,pad,act,n", [
    (256, 256, (18, 18, 18), 3, 0, "relu", 1),       # 16 tiles x 108 K-tiles over 256 workgroups: every tile is finished from ~16 partials
    (128, 288, (40, 38, 40), 3, 1, "leaky", 1),      # 238 tiles x 54: runs of ~50 K-tiles = the tail of one tile + the head of the next; zero padding (CHECK); backward: ragged chunk (288 = 4.5 x 64)
    (128, 256, (20, 50, 50), 3, 0, "none", 2),       # 324 tiles x 54: runs of 1.3 tiles — head piece first, then the tail, sometimes a WHOLE tile; N = 2
    (256, 512, (11, 12, 13), 2, 0, "sigmoid", 1),    # two column tiles (tile order: rows fastest), 2^3 taps, 4 chunks: runs of 1-2 K-tiles (nk = 1 and 2)
])
def test_igemm5_stream_k_schedule(cin, cout, sp, k, pad, act, n):
    """gemm5.hip stream-K (RG_SK=1 forces it wherever igemm5 runs): the (tile, K-tile) space cut into 256 equal runs, partial tiles handed over through
    the workspace and per-workgroup flags.  Forward and backward-data against torch fp32 on the CPU, and against the whole-tile schedule (RG_SK=-1) —
    shapes chosen so that runs are much shorter than a tile, about a tile, and longer than a tile."""
    from raygun_amd import _lib, ops
    dev = torch.device("cuda")
    x0 = rnd((n, cin) + sp, 191); w0 = rnd((cout, cin, k, k, k), 192, -0.05, 0.05); b0 = rnd((cout,), 193)
    afn = {"none": lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "leaky": lambda t: F.leaky_relu(t, 0.2)}[act]
    xr = x0.clone().requires_grad_()
    yr = afn(F.conv3d(xr, w0, b0, stride=1, padding=pad))
    go = rnd(tuple(yr.shape), 194)
    (gxr,) = torch.autograd.grad(yr, xr, go)
    outs = {}
    for sk in (1, -1, 1):      # twice with stream-K: the second launch finds the flags of the first (epochs, not resets)
        with ops.option("RG_GEMM5", 1), ops.option("RG_SK", sk):
            xd = x0.clone().to(dev).requires_grad_()
            y = ops.conv3d(xd, w0.to(dev), b0.to(dev), 1, pad, act, 0.2)
            name = _lib.lib().rg_last_kernel_name().decode()
            assert ("streamK" in name) == (sk == 1) and "igemm5" in name, name
            ops.profile_start()           # (the backward pass runs on autograd's thread: rg_last_kernel_name is per thread, the profile list is not)
            (gx,) = torch.autograd.grad(y, xd, go.to(dev).to(y.dtype))
            names = [n for n in ops.profile_stop() if "igemm" in n]
            assert names and all(("streamK" in n) == (sk == 1) and "igemm5" in n for n in names), names
            assert relerr(y, yr) <= TOL and relerr(gx, gxr) <= TOL, (sk, relerr(y, yr), relerr(gx, gxr))
            if sk in outs:   # the schedule is fixed by the shape: bit-identical from launch to launch
                assert torch.equal(outs[sk][0], y.float().cpu()) and torch.equal(outs[sk][1], gx.float().cpu())
            outs[sk] = (y.float().cpu(), gx.float().cpu())
    assert relerr(outs[1][0], outs[-1][0]) <= 2 ** -7 and relerr(outs[1][1], outs[-1][1]) <= 2 ** -7


@pytest.mark.gpu
def test_stream_k_is_the_default_for_the_resnet_block_convolutions():
    """The CycleGAN generator's 256 -> 256 blocks at 32^3 (reflect-padded to 34^3): 128 row tiles forward, 154 backward — neither is a whole round of 256
    workgroups, both run stream-K by default (the two K-slices + combine pass of round 3 remain behind RG_SK=-1); results within rounding of those kernels."""
    from raygun_amd import _lib, ops
    dev = torch.device("cuda")
    x0 = rnd((1, 256, 34, 34, 34), 195).to(dev); w0 = rnd((256, 256, 3, 3, 3), 196, -0.02, 0.02).to(dev); b0 = rnd((256,), 197).to(dev)
    outs = {}
    for sk in (0, -1):
        with ops.option("RG_SK", sk):
            xd = x0.clone().requires_grad_()
            y = ops.conv3d(xd, w0, b0, 1, 0, "none", 0.0)
            n1 = _lib.lib().rg_last_kernel_name().decode()
            ops.profile_start()
            (gx,) = torch.autograd.grad(y, xd, torch.ones_like(y))
            n2 = [n for n in ops.profile_stop() if "igemm" in n]
            assert ("streamK" in n1) == (sk == 0) and ("splitK" in n1) == (sk == -1) and len(n2) == 1 and ("streamK" in n2[0]) == (sk == 0), (n1, n2)
            outs[sk] = (y.float().cpu(), gx.float().cpu())
    assert relerr(outs[0][0], outs[-1][0]) <= 2 ** -7 and relerr(outs[0][1], outs[-1][1]) <= 2 ** -7


@pytest.mark.gpu
@pytest.mark.parametrize("B,m,k,n", [(2, 300, 200, 288), (3, 700, 2592, 576), (1, 257, 64, 144), (2, 1000, 448, 432)])
def test_gemm_batched_on_igemm5(B, m, k, n):
    """rg_gemm_batched (the Winograd path's 64 transform-domain GEMMs) on igemm5 and on igemm4 against torch.matmul in fp32."""
    import ctypes as C
    from raygun_amd import _lib, ops
    dev = torch.device("cuda")
    L = _lib.lib()
    kp, nr, ms_ = (k + 63) // 64 * 64, (n + 31) // 32 * 32, (n + 63) // 64 * 64
    A = torch.zeros(B, m, kp, device=dev, dtype=torch.bfloat16); A[:, :, :k] = rnd((B, m, k), 95).to(dev).to(torch.bfloat16)
    W = torch.zeros(B, nr, kp, device=dev, dtype=torch.bfloat16); W[:, :n, :k] = rnd((B, n, k), 96, -0.05, 0.05).to(dev).to(torch.bfloat16)
    ref = torch.matmul(A[:, :, :k].float(), W[:, :n, :k].float().transpose(1, 2)).cpu()
    for opt in (1, -1):
        Cm = torch.full((B, m, ms_), float("nan"), dtype=torch.bfloat16, device=dev)
        with ops.option("RG_GEMM5", opt):
            _lib.check(L.rg_gemm_batched(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(Cm.data_ptr()), B, m, k, n, kp, m * kp,
                                         nr * kp, ms_, m * ms_, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rg_gemm_batched")
        assert ("igemm5" in L.rg_last_kernel_name().decode()) == (opt == 1)
        got = Cm[:, :, :n].float().cpu()
        assert float((got - ref).abs().max() / ref.abs().max()) <= 8e-3


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,sp,k,pad,n", [
    (144, 144, (6, 9, 40), 3, 0, 1),      # a = 144 (FA 9), b = 128 + 16 (two b-tiles), line order, ragged W chunk
    (432, 288, (5, 19, 19), 3, 0, 1),     # flat-plane order (17 x 17 outputs of a 19 x 19 plane), 4 b-tiles of which the last holds 48 channels
    (200, 256, (7, 10, 34), 3, 1, 2),     # FA 8 with a ragged a-tile (200 -> 2 x 128), zero padding, N = 2
    (256, 160, (6, 12, 33), 2, 0, 1),     # 2 taps along W (the sub-pixel layers)
])
def test_wgrad_wide_tiles_ragged_shapes(cin, cout, sp, k, pad, n):
    """The wide weight-gradient kernels (producer-wave wgrad_kwp: 144- and 128-channel tiles) against torch fp32 on the CPU for both row orders, both
    a-tile widths, 2 and 3 W taps, ragged tiles and ragged unit counts."""
    from raygun_amd import ops
    dev = torch.device("cuda")
    x0 = rnd((n, cin) + sp, 97); w0 = rnd((cout, cin, k, k, k), 98, -0.05, 0.05)
    xr = x0.clone(); wr = w0.clone().requires_grad_()
    yr = F.conv3d(xr, wr, None, padding=pad)
    go = rnd(tuple(yr.shape), 99)
    (gwr,) = torch.autograd.grad(yr, wr, go)
    wd = w0.clone().to(dev).requires_grad_()
    y = ops.conv3d(x0.to(dev), wd, None, 1, pad, "none")
    (gw,) = torch.autograd.grad(y, wd, go.to(dev).to(y.dtype))
    assert relerr(gw, gwr) <= TOL, relerr(gw, gwr)


@pytest.mark.gpu
def test_igemm5_split_k_small_grid():
    """igemm5 with K-slices (the 256 -> 256 ResNet-block convolution of the CycleGAN at 32^3: 128 row tiles x 1 column tile, K = 108
    tiles -> 2 slices, fp32 partial tiles through the workspace, igemm4's combine kernel adds bias + activation): forward and
    backward-data against torch fp32 on the CPU.  Since round 4 the stream-K schedule takes this shape by default; the K-slices remain
    what the heuristic picks with stream-K off (RG_SK = -1) and for callers whose workspace only holds the slices."""
    from raygun_amd import _lib, ops
    dev = torch.device("cuda")
    x0 = rnd((1, 256, 34, 34, 34), 101); w0 = rnd((256, 256, 3, 3, 3), 102, -0.02, 0.02); b0 = rnd((256,), 103)
    xr = x0.clone().requires_grad_()
    yr = torch.relu(F.conv3d(xr, w0, b0))
    go = rnd(tuple(yr.shape), 104)
    (gxr,) = torch.autograd.grad(yr, xr, go)
    with ops.option("RG_SK", -1):
        xd = x0.clone().to(dev).requires_grad_()
        y = ops.conv3d(xd, w0.to(dev), b0.to(dev), 1, 0, "relu")
        name = _lib.lib().rg_last_kernel_name().decode()
        assert "igemm5" in name and "splitK" in name, name
        (gx,) = torch.autograd.grad(y, xd, go.to(dev).to(y.dtype))
    assert relerr(y, yr) <= TOL and relerr(gx, gxr) <= TOL, (relerr(y, yr), relerr(gx, gxr))


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,sp,k", [
    (1152, 1280, (10, 18, 18), 3),      # plane 16 x 16 = 256 (no padding), 8 output planes, 27 batches, full tiles
    (1152, 1100, (34, 17, 17), 3),      # ragged: 1100 rows (last A tile 76 rows), plane 15 x 15 = 225 -> 256 zero-padded, 32 output planes
    (1024, 2048, (17, 17, 17), 2),      # 2^3 taps (the sub-pixel layers), FN = 8 column tiles, plane 256: 256 workgroups = one round
])
def test_wgrad_gemm_form_wide_layers(cin, cout, sp, k):
    """The GEMM form of the weight gradient (gemm5.hip rg_launch_wgrad_gemm: both operands transposed once in HBM — zero-padded planes, one X copy per
    in-plane tap — then one raw-fp32-output igemm5 launch with a batch per tap) against torch fp32 on the CPU and against the producer-wave kernel
    (RG_NO_WGEMM = 1) on the same operands."""
    from raygun_amd import ops
    dev = torch.device("cuda")
    x0 = rnd((1, cin) + sp, 197); w0 = rnd((cout, cin, k, k, k), 198, -0.05, 0.05)
    wr = w0.clone().requires_grad_()
    yr = F.conv3d(x0, wr, None)
    go = rnd(tuple(yr.shape), 199)
    (gwr,) = torch.autograd.grad(yr, wr, go)
    outs = {}
    for off in (0, 1):
        with ops.option("RG_NO_WGEMM", off):
            wd = w0.clone().to(dev).contiguous(memory_format=torch.channels_last_3d).requires_grad_()     # hipnn.Conv3d keeps its masters this way
            y = ops.conv3d(x0.to(dev), wd, None, 1, 0, "none")
            ops.profile_start()
            (gw,) = torch.autograd.grad(y, wd, go.to(dev).to(y.dtype))
            names = list(ops.profile_stop())
            assert any("wgrad_gemm" in nm for nm in names) == (off == 0), names
            assert relerr(gw, gwr) <= TOL, (off, relerr(gw, gwr))
            outs[off] = gw.float().cpu()
    assert relerr(outs[0], outs[1]) <= 2 ** -9
