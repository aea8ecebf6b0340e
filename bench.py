#!/usr/bin/env python3
"""Headline benchmark (driver contract):  python bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json's primary metric: 3-D U-Net (MTLSDModel, config "U": ngf 12, x6, 3 levels,
valid, 259 146 829 parameters, num_affs 3) affinity train step on a synthetic 196^3 volume per GPU:
forward + WeightedMSELoss_MTLSD + backward + Adam (+ RCCL gradient all-reduce for N > 1), bf16 storage /
fp32 accumulate.  value = input Mvoxels/s over the whole job (196^3 * N / step time), "scaling": "weak".
The same run then times the 3-D CycleGAN link train step at 128^3 (BASELINE's second figure, ms/step) and reports it under
`secondary` in the same line at N = 1 (`--no-secondary` skips it, `--secondary-dp` adds it to N > 1 runs; `--workload cyclegan` times
only that).  With --gpus N > 1 and no
launcher (no RANK in the environment) bench.py starts its own N rank processes (RCCL).
`--workload predict` times block-wise volume prediction (first half: 196^3 -> 104^3 blocks of the
U-Net with 20 affinity channels over a synthetic uint8 volume resident in HBM); `--workload segment` times the mutex
watershed (second half: segment.py's prepare + 20-offset / sep 3 / stride 10 mutex watershed of a synthetic structured
affinity volume resident in HBM).

One JSON line is printed by rank 0, carrying `roofline` (dominant conv kernel, HIP events on the launch stream,
algorithmic FLOPs = 2*rows*taps*Cin*Cout) and `cpu_baseline` (the oracle's plain-PyTorch CPU restatement on a
bounded sample, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md: 256 CU x 4096 FLOP/clk x 2.4 GHz)
UNET_FLOP_PER_STEP = 19.47e12  # 3 x 6.4925 TFLOP - 4.7 GFLOP (no dgrad into the raw input)
CYC_FLOP_PER_STEP = 94.3e12    # SURVEY.md §8a train-step FLOP budget (link, identity lambda 0, 128^3)
CYC_FLOP_IDENTITY = 140.6e12   # the reference's own default, identity lambda 0.5 (two more generator passes per step)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="unet", choices=["unet", "cyclegan", "predict", "segment", "config5", "watershed"])
    ap.add_argument("--size", type=int, default=0, help="cube edge (default 196 unet, 128 cyclegan, 404 predict = 3^3 blocks, 256 segment)")
    ap.add_argument("--affs", default="blobs", choices=["blobs", "noise"], help="segment workload: structured affinities or pure noise (adversarial)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f16"], help="unet workload: 16-bit storage type (f16 = IEEE half + dynamic loss scaling)")
    ap.add_argument("--identity", type=float, default=0.0, help="cyclegan workload: identity-loss lambda (the reference's default config uses 0.5: two more generator passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="unet workload: do not also time the CycleGAN 128^3 step")
    ap.add_argument("--secondary-dp", action="store_true", help="(default since round 4: every --gpus N run times the CycleGAN step data-parallel as well — "
                    "BASELINE config 4 is 'DDP CycleGAN + UNet'; the flag is accepted and ignored)")
    ap.add_argument("--replicas", action="store_true", help="config5 / predict / segment with --gpus N: one private volume per GPU (no exchange) instead of ONE "
                    "global volume sharded over the ranks")
    ap.add_argument("--source", default="hbm", choices=["hbm", "zarr"], help="predict workload: the volume resident in HBM (default), or the reference's real flow — "
                    "zarr uint8 source on disk -> block runner -> zarr uint8 destinations on disk (predict.py:111-164)")
    ap.add_argument("--cpu-size", type=int, default=0)
    ap.add_argument("--rank-timeout", type=float, default=900.0, help="--gpus N without a launcher: seconds before the supervised ranks are terminated")
    return ap.parse_args()


def build_unet(dev, size, seed, precision="bf16"):
    from raygun_amd import synth
    from raygun_amd.torch.models import MTLSDModel
    from raygun_amd.torch.losses import WeightedMSELoss_MTLSD
    from raygun_amd.torch.optimizers import FusedAdam
    from raygun_amd import parallel
    torch.manual_seed(1234)  # identical random-init weights on every rank
    with torch.device(dev):
        model = MTLSDModel(num_affs=3)
    loss_fn = WeightedMSELoss_MTLSD()
    opt = FusedAdam(model.parameters(), lr=0.5e-4)
    if precision != "f16" and parallel.world_size() == 1 and os.environ.get("RAYGUN_AMD_ADAM_OVERLAP", "0") == "1":
        opt.overlap_backward()   # opt-in, as in train_step (raygun_amd/torch/train/driver.py): measured 22.38 ms against 22.19 without
    raw = torch.from_numpy(synth.uniform((1, 1, size, size, size), 42 + seed)).to(dev)
    out = 8 * ((size - 92) // 8)
    gt_a = torch.from_numpy(synth.uniform((1, 3, out, out, out), 142 + seed, 0, 1)).to(dev)
    gt_l = torch.from_numpy(synth.uniform((1, 10, out, out, out), 242 + seed, 0, 1)).to(dev)
    w_a, w_l = torch.ones_like(gt_a), torch.ones_like(gt_l)

    attached = []
    import contextlib
    from raygun_amd import amp, ops
    # f16 storage: IEEE-half activations / packed weights / activation gradients + dynamic loss scaling (raygun_amd/amp.py); the
    # inf check and its 4-byte host read are part of the optimiser call, inside the timed step
    scaler = amp.LossScaler() if precision == "f16" else None

    def step(sync, marks=None):
        if not attached:  # overlap the gradient all-reduce with the backward pass (no-op at world size 1)
            sync.attach(model.parameters()); attached.append(True)
        opt.zero_grad(set_to_none=True)
        with (ops.precision("f16") if scaler is not None else contextlib.nullcontext()):
            affs, lsds = model(raw)
            loss = loss_fn(lsds, gt_l, w_l, affs, gt_a, w_a)
        ws = parallel.world_size()
        out = scaler.scale(loss) if scaler is not None else loss
        if ws > 1 and sync.prescaled:   # mean over ranks = sum of gradients of loss / world: no division pass over 1 GB of gradients
            out.backward(torch.full_like(loss, 1.0 / ws))
        else:
            out.backward()
        sync([p for p in model.parameters()])
        do_step = (lambda: scaler.step(opt)) if scaler is not None else opt.step
        if marks is not None:   # "fwd+bwd (+Adam reported separately)": one event pair around the optimiser call
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); do_step(); e1.record()
            marks.append((e0, e1))
        else:
            do_step()
        return loss

    return step, dict(model="MTLSDModel(UNet ngf12 x6 3-level valid)+WeightedMSELoss_MTLSD+Adam", params=259146829, scaler=scaler)


def parallel_world():
    from raygun_amd import parallel
    return parallel.world_size()


def build_cyclegan(dev, size, seed, precision="bf16", identity=0.0):
    import functools
    from raygun_amd import synth
    from raygun_amd.torch.systems import CycleGAN
    torch.manual_seed(1234)
    cfg = dict(common_voxel_size=[1, 1, 1], ndims=3, gnet_type="resnet",
               gnet_kwargs=dict(input_nc=1, ngf=64, n_blocks=6, n_downsampling=2, norm_layer=functools.partial(torch.nn.InstanceNorm3d),
                                padding_type="same"),
               dnet_kwargs=dict(input_nc=1, ngf=64, n_layers=3, kw=4), loss_type="link",
               loss_kwargs={"g_lambda_dict": {s: {"l1_loss": {"cycled": 10, "identity": identity}, "gan_loss": {"fake": 1, "cycled": 0}} for s in "AB"}},
               random_seed=None)
    # RAYGUN_AMD_GRAPH=1: single-process bf16 runs replay the step from a HIP graph (raygun_amd/graphs.py).  Opt-in: measured 73.0 ms against 74.1 eager —
    # the 8 ms the GPU idles per step are dispatch latency between dependent few-microsecond kernels, which a graph replay pays as well, not host time
    #  — and the timed region then carries no per-kernel events.  Data-parallel and loss-scaled steps always run eagerly (host-side bookkeeping
    # in the gradient hooks / the scaler's inf check are not part of a replay)
    use_graph = precision != "f16" and parallel_world() == 1 and os.environ.get("RAYGUN_AMD_GRAPH", "0") == "1"
    if use_graph:
        cfg["g_optim_kwargs"] = {"betas": [0.9, 0.999], "lr": 1e-5, "weight_decay": 0, "capturable": True}
        cfg["d_optim_kwargs"] = {"betas": [0.9, 0.999], "lr": 1e-5, "weight_decay": 0, "capturable": True}
    s = CycleGAN(cfg)
    s.setup_networks()
    for n in (s.netG1, s.netG2, s.netD1, s.netD2):
        n.to(dev)
    s.setup_model()
    holder = {}
    s.setup_optimization(grad_sync=lambda params: holder["sync"](params))
    A = torch.from_numpy(synth.uniform((1, 1, size, size, size), 42 + seed)).to(dev)
    B = torch.from_numpy(synth.uniform((1, 1, size, size, size), 1042 + seed)).to(dev)

    import contextlib
    from raygun_amd import amp, ops
    scaler = amp.LossScaler() if precision == "f16" else None     # the losses own backward + step: they scale every backward and check every step through it
    if scaler is not None:
        s.loss.scaler = scaler

    def body():
        s.optimizer.zero_grad()
        with (ops.precision("f16") if scaler is not None else contextlib.nullcontext()):
            fB, cB, fA, cA = s.model(real_A=A, real_B=B)
            loss = s.loss(A, fA, cA, B, fB, cB)
        loss.backward()
        return loss

    graphed = None
    if use_graph:   # ~2 500 launches per step, the host as slow as the GPU: the step is recorded once and replayed (raygun_amd/graphs.py)
        from raygun_amd.graphs import GraphedStep
        graphed = GraphedStep(body, optimizers=[o for o in vars(s.optimizer).values() if isinstance(o, torch.optim.Optimizer)])

    def step(sync):
        holder["sync"] = sync
        # (the per-kernel profile passes time the library's launches from Python: they run the same step eagerly)
        loss = graphed() if graphed is not None and ops._prof is None else body()
        s.optimizer.step()      # (host side only: the LR schedulers; the losses own the real optimiser steps)
        return loss

    step.graphed = graphed is not None
    fused = os.environ.get("RAYGUN_AMD_FUSE_TAIL", "1") != "0"
    return step, dict(model="CycleGAN link: ResNet(ngf64,6 blocks,2 down,IN)+PatchD(ngf64,3 layers) lsgan"
                            + (f", identity lambda {identity}" if identity else "")
                            + ("; generator tail Conv3d(64,64,7)+1x1 evaluated as one composed conv (DESIGN.md 3a)" if fused else ""),
                      params=71848196, scaler=scaler)


def pmc_traffic(kernel_name, tag="unet"):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC summaries (profiles/*_pmc_{fetch,write}_size.csv,
    separate --pmc passes).  FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md §HBM (gfx950
    tallies 128-byte read requests at 64 B).  None when no profile names this kernel."""
    import csv
    import glob
    kernel_name = kernel_name.replace("wgrad_gemm_", "")    # the GEMM form of the weight gradient is igemm5_kernel<0, FN, 1> in rocprofv3's table
    key = kernel_name.split("<")[0]
    args = kernel_name[len(key):].strip("<>").split(",") if "<" in kernel_name else []
    out = {}
    for kind, scale in (("fetch", 2.0 * 1024), ("write", 1024.0)):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{tag}*_pmc_{kind}_size.csv")))
        if not files:
            return None
        best = None
        for r in csv.DictReader(open(files[-1])):
            nm = r["kernel"].replace("void ", "").replace(" ", "").replace("true", "1").replace("false", "0")
            if nm.startswith(key + "<" + ",".join(args)) or (not args and nm.startswith(key)):
                best = float(r["mean_per_dispatch"]) * scale
                break
        if best is None:
            return None
        out[kind] = best
    return out["fetch"] + out["write"]


def host_cpu():
    """(threads to use, description): the physical cores this process may run on (affinity mask / SMT siblings) and the CPU model."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    phys = avail
    try:
        import psutil
        lp, pp = psutil.cpu_count(logical=True) or avail, psutil.cpu_count(logical=False) or avail
        phys = max(1, min(avail, avail * pp // lp if avail == lp else avail))   # a restricted mask (container share) is taken as given
    except Exception:
        pass
    model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return phys, f"{model}, {phys} of {os.cpu_count()} hardware threads used"


def timed_median(fn, budget_s=45.0, reps=3):
    """1 warm-up call + up to `reps` timed calls of fn, bounded by `budget_s` of wall time: returns (median s,
    calls timed, warm-up s).  A warm-up that alone exceeds the budget is the measurement."""
    import statistics
    t0 = time.time(); fn(); warm = time.time() - t0
    ts = []
    while len(ts) < reps and (time.time() - t0) + (ts[-1] if ts else warm) <= budget_s:
        t1 = time.time(); fn(); ts.append(time.time() - t1)
    return (statistics.median(ts), len(ts), warm) if ts else (warm, 0, warm)


def cpu_baseline(workload, size):
    """The oracle (plain PyTorch fp32 restatement) timed on this host's cores: one warm-up step + the median of up to 3 steps on
    a bounded sample."""
    from oracle import torch_ref as O
    from raygun_amd import synth
    cores, cpu = host_cpu()
    torch.set_num_threads(cores)
    if workload == "unet":
        m = O.MTLSDModel()
        lf = O.WeightedMSELoss_MTLSD()
        raw = torch.from_numpy(synth.uniform((1, 1, size, size, size), 42))
        out = 8 * ((size - 92) // 8)
        ga = torch.from_numpy(synth.uniform((1, 3, out, out, out), 142, 0, 1))
        gl = torch.from_numpy(synth.uniform((1, 10, out, out, out), 242, 0, 1))
        opt = torch.optim.Adam(m.parameters(), lr=0.5e-4)
        # 1 warm-up + >= 3 timed steps (~26 s each on 128 cores: ~110 s of the driver's 600)
        dt, n, warm = timed_median(lambda: O.mtlsd_step(m, lf, opt, raw, gl, torch.ones_like(gl), ga, torch.ones_like(ga)), budget_s=170.0)
        return dict(value=size ** 3 / dt / 1e6, unit="Mvoxels/s", cores=cores, kind="port", cpu=cpu,
                    sample=f"train step (fwd+loss+bwd+Adam) of the CPU restatement at {size}^3 (torch {torch.__version__} fp32, oneDNN): "
                           f"1 warm-up step ({warm:.1f} s) + median of {n} steps = {dt:.1f} s")
    import functools
    IN3 = functools.partial(torch.nn.InstanceNorm3d)
    model, loss, _ = O.build_cyclegan(3, dict(input_nc=1, ngf=64, n_blocks=6, n_downsampling=2, norm_layer=IN3, padding_type="same"),
                                      dict(input_nc=1, ngf=64, n_layers=3, kw=4), g_lr=1e-5, d_lr=1e-5, betas=(0.9, 0.999))
    A = torch.from_numpy(synth.uniform((1, 1, size, size, size), 42)); B = torch.from_numpy(synth.uniform((1, 1, size, size, size), 1042))
    dt, n, warm = timed_median(lambda: O.cyclegan_step(model, loss, A, B), budget_s=45.0)
    return dict(value=dt * 1e3, unit="ms/step", cores=cores, kind="port", cpu=cpu,
                sample=f"link train step of the CPU restatement at {size}^3 (work ratio to 128^3: {(128 / size) ** 3:.0f}x): "
                       f"1 warm-up ({warm:.1f} s) + median of {n} steps = {dt:.1f} s")


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
MWS_BYTES_PER_VOXEL = 88  # 20 float32 affinity channels read once + one uint64 label written


def structured_affs(size, seed, device):
    """Synthetic 20-channel affinities with object structure: high inside smooth blobs, low across their boundaries, plus
    20 % uniform noise (the random-initialised U-Net of the predict workload would give pure noise)."""
    g = torch.Generator(device=device).manual_seed(1000 + seed)
    r = torch.arange(size, device=device, dtype=torch.float32)
    f = torch.sin(r / 9.1 + 1)[:, None, None] * torch.sin(r / 11.3)[None, :, None] * torch.sin(r / 8.7 + 1.0)[None, None, :]
    base = (f.abs() > 0.2).float() * 0.8 + 0.1
    return 0.8 * base[None] + 0.2 * torch.rand((20, size, size, size), generator=g, device=device)


def main_aux(a, rank, local, world, dev):
    """--workload predict / segment.  No collective on either path (8e): ranks work on their own volume."""
    from raygun_amd import _lib, ops, predict as P, segment as S, synth
    import ctypes as C
    import numpy as np

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if a.workload == "predict":
        from raygun_amd.torch.models import MTLSDModel
        size = a.size or 404
        torch.manual_seed(1234)
        with torch.device(dev):
            model = MTLSDModel(num_affs=20).eval()
        g = torch.Generator(device=dev).manual_seed(77 + rank)
        vol = torch.randint(0, 256, (size,) * 3, generator=g, device=dev, dtype=torch.uint8)
        nblk = len(P.enumerate_blocks(vol.shape, *P.roi_geometry(3, [196] * 3, [104] * 3)))
        units = nblk * 104 ** 3

        def step():
            # rank=0, world=1: every rank predicts ITS OWN whole volume (weak scaling), nothing is sharded a second time
            return P.predict_volume(model, vol, input_shape=[196] * 3, output_shape=[104] * 3, out_dtypes=[np.uint8, np.uint8], rank=0, world=1)
        zarr_info = None
        if a.source == "zarr":
            # the reference's real flow (predict.py:70,111-125,154-164 + worker.py:94-160): a chunked uint8 source container on disk, one destination dataset per
            # model output chunked by the write ROI (uint8: value * 255 clamped), Blosc / lz4 like daisy.prepare_ds's default; everything crosses PCIe and the disk
            import shutil
            import tempfile
            from raygun_amd import zarrio
            root = tempfile.mkdtemp(prefix=f"raygun_amd_bench_r{rank}_", dir=os.environ.get("RAYGUN_AMD_BENCH_TMP") or None)
            path = os.path.join(root, "vol.zarr")
            src = zarrio.prepare_ds(path, "volumes/raw", vol.shape, (4, 4, 4), np.uint8, write_size=(64, 64, 64))
            src[...] = vol.cpu().numpy()
            names = ["volumes/pred_affs", "volumes/pred_lsds"]
            zarr_info = dict(container=path, source_chunks=[64] * 3, destination_chunks=[20, 104, 104, 104], compressor=src.compressor)

            def step():
                dests = P.predict_zarr(model, path, "volumes/raw", output_ds=names, out_specs={n_: {"dtype": np.uint8} for n_ in names}, input_shape=[196] * 3,
                                       output_shape=[104] * 3, out_channels=[20, 10], rank=0, world=1)
                return dests, nblk
    else:
        size = a.size or 256
        affs = structured_affs(size, rank, dev)
        if a.affs == "noise":
            affs = torch.rand(affs.shape, generator=torch.Generator(device=dev).manual_seed(5 + rank), device=dev)
        nb = S.build_neighborhood()
        units = size ** 3

        def step(stats=None):
            w, m = S.prepare_weights(affs, 3, 0.5, 1.0)
            return S.compute_mws_segmentation(w, nb, 3, strides=[10, 10, 10], mask=m, as_tensor=True, stats=stats)

    for _ in range(a.warmup):
        out = step()
    barrier()
    ops.profile_start()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    noise_case = None
    if a.workload == "segment":  # the library bracketed the persistent Kruskal kernel of every call
        buf = (C.c_float * (a.steps + 1))()
        n = _lib.lib().rg_profile_drain(buf, a.steps + 1)
        prof = {"mws_rounds_kernel": dict(launches=n, ms=float(sum(buf[:n])), flops=0.0)}
        ops.profile_stop()
        if int(out.view(-1)[0].item()) == -1:
            raise RuntimeError("mutex watershed made no progress")
        round_stats = {}
        step(round_stats)  # one more, untimed call for the round kernel's own counters
        if a.affs == "blobs":  # the adversarial input beside the structured one: pure noise = one giant cluster, ~1100 rounds
            structured = affs
            affs = torch.rand(structured.shape, generator=torch.Generator(device=dev).manual_seed(5 + rank), device=dev)
            step(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                nout = step()
            torch.cuda.synchronize()
            nms = (time.perf_counter() - t1) / 3 * 1e3
            nstats = {}
            step(nstats)
            noise_case = dict(workload="pure-noise 20-channel affinities (adversarial: one giant cluster)", ms_per_step=round(nms, 3),
                              value=round(units / (nms * 1e-3) / 1e6, 1), unit="Mvoxels/s", segments=int(nout.max().item()), round_kernel=nstats)
            affs = structured
    else:
        prof = ops.profile_stop()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        if rank != 0:
            torch.distributed.destroy_process_group()
            return
    ms = dt / a.steps * 1e3
    if a.workload == "segment":
        d = prof["mws_rounds_kernel"]
        avg = d["ms"] / max(1, d["launches"])
        ach = units * MWS_BYTES_PER_VOXEL / (avg * 1e-3) / 1e9
        roofline = dict(bound="hbm", kernel="mws_rounds_kernel", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(ach / HBM_PEAK_GBS, 5), traffic=pmc_traffic("mws_rounds_kernel", "segment") if a.affs == "blobs" and size == 256 else None,
                        traffic_note="HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE) from the committed profiles/*_segment256_pmc_*.csv passes",
                        launches_per_step=1, avg_launch_ms=round(avg, 4),
                        note="algorithmic bytes = 88 B/voxel (20 fp32 channels + u64 label); the pass is a dependency-ordered "
                             "union-find over sorted edges, bound by round latency and device-scope atomics, not by HBM streaming")
        res = dict(metric="Mvoxels/sec mutex watershed (prepare + 20 offsets, sep 3, strides 10)", unit="Mvoxels/s", dtype="f32 keys / i32 union-find",
                   config=dict(workload=f"segment {size}^3 per GPU: synthetic " + ("structured 20-channel affinities (blobs + 20% noise), "
                                        if a.affs == "blobs" else "pure-noise 20-channel affinities (adversarial: one giant cluster), ") +
                                        "segment.py:188-199 pipeline, labels left in HBM", parallelism=f"replicas x{world}"),
                   segments=int(out.max().item()))
        res["round_kernel"] = round_stats  # edges, rounds, window visits, ms in phase A / B / C (thread 0's clock, barriers included)
        if noise_case is not None:
            res["noise_case"] = noise_case
    else:
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
        ach = dom[1]["flops"] / (dom[1]["ms"] * 1e-3) / 1e12
        roofline = dict(bound="mfma", kernel=dom[0], achieved=round(ach, 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                        frac=round(ach / PEAK_BF16_TFLOPS, 4), traffic=None, launches_per_step=dom[1]["launches"] // a.steps,
                        avg_launch_ms=round(dom[1]["ms"] / dom[1]["launches"], 4))
        if a.source == "zarr":
            import shutil
            from raygun_amd import predict as P2
            zarr_info["combiner"] = P2.COMBINER_STATS[-2:]
            zarr_info["bytes_on_disk"] = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(zarr_info["container"]) for f in fs)
            shutil.rmtree(os.path.dirname(zarr_info["container"]), ignore_errors=True)
        res = dict(metric="Mvoxels/sec block-wise predict (output voxels), 196^3 -> 104^3 blocks" + (", zarr source -> zarr destinations on disk" if a.source == "zarr" else ""),
                   unit="Mvoxels/s",
                   dtype=os.environ.get("RAYGUN_AMD_PREDICT_PRECISION", "f16"),   # the block runner's 16-bit storage type (predict_volume's default: IEEE half)
                   config=dict(workload=f"predict {size}^3 uint8 volume per GPU in {nblk} blocks, MTLSDModel(UNet ngf12 x6 3-level valid, "
                                        "20 affinities + 10 LSDs), " + ("zarr uint8 source (64^3 chunks) and zarr uint8 destinations (chunks = write ROI) ON DISK, PCIe + codec + "
                                        "file system inside the timed region" if a.source == "zarr" else "uint8 destinations in HBM"),
                           parallelism=f"one volume of {nblk} blocks per GPU x{world}"),
                   blocks_per_s=round(nblk * world / (dt / a.steps), 2), **({"zarr": zarr_info} if zarr_info else {}))
    res.update(value=round(units * world / (dt / a.steps) / 1e6, 3), n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms, 3),
               higher_is_better=True, scaling="weak", vs_baseline=None, data="synthetic", roofline=roofline)
    if world == 1 and not a.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline_aux(a.workload, a.cpu_size)
        except Exception as e:
            res["cpu_baseline"] = dict(error=repr(e))
    print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def config5_volume(gshape, z0, z1, dev, seed=77):
    """Planes [z0, z1) of the synthetic uint8 volume of --workload config5: smooth blobs + 60 grey levels of noise, both pure functions of the GLOBAL
    voxel coordinate (an integer hash for the noise), so that every rank generates exactly its part of ONE volume."""
    Z, Y, X = gshape
    y = torch.arange(Y, device=dev, dtype=torch.float32)
    x = torch.arange(X, device=dev, dtype=torch.float32)
    yi, xi = torch.arange(Y, device=dev, dtype=torch.int64)[None, :, None], torch.arange(X, device=dev, dtype=torch.int64)[None, None, :]
    out = torch.empty((z1 - z0, Y, X), dtype=torch.uint8, device=dev)
    for a0 in range(z0, z1, 32):
        a1 = min(z1, a0 + 32)
        z = torch.arange(a0, a1, device=dev, dtype=torch.float32)
        blobs = (torch.sin(z / 9.1 + 1)[:, None, None] * torch.sin(y / 11.3)[None, :, None] * torch.sin(x / 8.7 + 1.0)[None, None, :]).abs() > 0.2
        zi = torch.arange(a0, a1, device=dev, dtype=torch.int64)[:, None, None]
        h = (zi * 73856093 + seed) ^ (yi * 19349663) ^ (xi * 83492791)
        h = ((h * 2654435761) >> 11) & 0xFFFF
        out[a0 - z0:a1 - z0] = (blobs.float() * 140 + 40 + 60 * (h.float() / 65536.0)).clamp_(0, 255).to(torch.uint8)
    return out


def main_config5(a, rank, local, world, dev):
    """--workload config5 (HBM-resident): a synthetic uint8 volume (hash noise + smooth blobs)
    -> block-wise prediction with MTLSDModel(num_affs = 20) in 196^3 -> 104^3 blocks (f16 storage, uint8 affinities left on the device)
    -> rg_mws_prepare -> (slab) mutex watershed, sep 3, strides 10, mask threshold 0.5.  One JSON line: end-to-end seconds, blocks/s,
    Mvoxels/s and both rooflines (the dominant convolution kernel against the MFMA peak, the Kruskal kernel against HBM).

    --gpus N > 1: ONE global volume, as BASELINE config 5 says (reference flow: predict.py:154-170 over the whole dataset, then segment.py:139-230 over
    the whole affinity array).  pipeline.shard_plan cuts its Z axis on the block grid; every rank generates its planes (+ the read halo of its first /
    last block row) from the global coordinates, predicts its blocks (no collective), and segments its owned planes through
    segment_slabs(z0, global_shape) — one neighbour exchange of the boundary zones + an all-gather of the second-level edge records (RCCL).  The
    labels stay distributed (rank r holds its planes).  Default global volume: 404 x 404 in-plane and 3 block rows of Z per rank (weak scaling:
    27 blocks per GPU, as the one-GPU default 404^3); `--size S`: the S^3 cube over all ranks (strong scaling — 2048 is BASELINE's).  `--replicas`
    keeps the round-3 behaviour (one private volume per GPU, no exchange)."""
    import ctypes as C
    from raygun_amd import _lib, ops, pipeline, predict as P
    from raygun_amd.torch.models import MTLSDModel
    sharded = world > 1 and not a.replicas
    size = a.size or 404
    gshape = (size,) * 3 if (a.size or not sharded) else (92 + 104 * 3 * world, 404, 404)
    torch.manual_seed(1234)
    with torch.device(dev):
        model = MTLSDModel(num_affs=20).eval()
    if sharded:
        part = pipeline.shard_plan(gshape, world)[rank]
        vol = config5_volume(gshape, part["src"][0], part["src"][1], dev)
        chain_kw = dict(z0=part["src"][0], own=part["own"], global_shape=gshape)
    else:
        part = dict(src=(0, gshape[0]), own=(0, gshape[0]), rows=None)
        vol = config5_volume(gshape, 0, gshape[0], dev, seed=77 + rank)
        chain_kw = {}
    nblk = len(P.enumerate_blocks(vol.shape, *P.roi_geometry(3, [196] * 3, [104] * 3)))
    from raygun_amd import segment as S
    nbh = torch.as_tensor(S.build_neighborhood().astype("int64"))

    def shaped(x):
        """The random-initialised network's affinities are noise around 0.5 — the adversarial input of the Kruskal pass (one giant cluster
        that everything merges into one edge at a time).  A trained network predicts object structure; to time the chain on such input
        the network's output is blended 20 / 80 with a structure term computed from the block's own intensities: offset k is "attractive"
        where voxel p and p + offset_k both lie inside a blob (the same construction as --workload segment's structured_affs: the
        background ends up below the mask threshold).  The full U-Net forward still runs for every block."""
        affs, lsds = model(x)
        raw = x.float()[0, 0]                                     # (196, 196, 196) in [0, 1]
        fg = raw > 0.55
        c = fg[46:150, 46:150, 46:150]
        terms = []
        for dz, dy, dx in nbh.tolist():
            q = fg[46 + dz:150 + dz, 46 + dy:150 + dy, 46 + dx:150 + dx]
            terms.append((c & q).float() * 0.8 + 0.1)
        st = torch.stack(terms)[None]
        return 0.2 * affs.float() + 0.8 * st, lsds

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    steps, warmup = max(1, min(a.steps, 3)), max(1, min(a.warmup, 1))
    if max(gshape) <= 512 and not sharded:
        _, raw_info = pipeline.predict_and_segment(model, vol)       # the raw network output, once: the adversarial case beside the headline
        raw_case = dict(total_s=round(raw_info["total_s"], 3), segment_s=round(raw_info["segment_s"], 3), predict_s=round(raw_info["predict_s"], 3),
                        note="UNSHAPED output of the random-initialised network (noise around 0.5): adversarial for the Kruskal pass")
    else:
        raw_case = dict(skipped="the unshaped (pure-noise) case is reported by the one-GPU run at <= 512^3 only")
    model_ = shaped
    for _ in range(warmup):
        labels, info = pipeline.predict_and_segment(model_, vol, **chain_kw)
    barrier()
    ops.profile_start()
    t0 = time.perf_counter()
    tp = ts = 0.0
    for _ in range(steps):
        labels, info = pipeline.predict_and_segment(model_, vol, **chain_kw)
        tp += info["predict_s"]; ts += info["segment_s"]
    barrier()
    dt = time.perf_counter() - t0
    buf = (C.c_float * (8 * steps + 1))()
    n = _lib.lib().rg_profile_drain(buf, 8 * steps + 1)        # the library bracketed the Kruskal kernel of every mutex-watershed pass
    mws_ms = float(sum(buf[:n])) / steps
    prof = ops.profile_stop()
    nseg = int(labels.max().item())
    blocks_all, own_vox = nblk, int(labels.numel())
    if world > 1:
        t = torch.tensor([dt, tp, ts], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, tp, ts = (float(v) for v in t.tolist())
        c = torch.tensor([nblk, own_vox], device=dev, dtype=torch.int64)
        torch.distributed.all_reduce(c, op=torch.distributed.ReduceOp.SUM)
        blocks_all, own_vox = (int(v) for v in c.tolist())
        m = torch.tensor([nseg], device=dev, dtype=torch.int64)
        torch.distributed.all_reduce(m, op=torch.distributed.ReduceOp.MAX)   # labels are global ranks of first voxels: the largest one is the count
        nseg = int(m.item())
        backend = _backend_string()
        ranks_seen = torch.distributed.get_world_size()
        if rank != 0:
            torch.distributed.destroy_process_group()
            return
    else:
        backend, ranks_seen = "none (single process)", 1
    sec = dt / steps
    voxels = (gshape[0] * gshape[1] * gshape[2]) if sharded else size ** 3 * world
    assert not sharded or own_vox == voxels, (own_vox, voxels)      # the owned parts tile the global volume
    dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
    ach = dom[1]["flops"] / (dom[1]["ms"] * 1e-3) / 1e12
    hbm = labels.numel() * MWS_BYTES_PER_VOXEL / (mws_ms * 1e-3) / 1e9 if mws_ms else None
    shape_s = "x".join(str(v) for v in gshape)
    res = dict(metric="config 5 end to end: block-wise predict + mutex watershed, seconds per volume", value=round(sec, 4), unit="s", n_gpus=world, steps=steps,
               warmup=warmup, ms_per_step=round(sec * 1e3, 2), higher_is_better=False, scaling=("strong" if a.size else "weak"), vs_baseline=None,
               dtype="f16 storage / f32 keys", data="synthetic",
               config=dict(workload=(f"config5: ONE {shape_s} uint8 volume sharded over {world} ranks on the block grid (pipeline.shard_plan), " if sharded else
                                     f"config5 {shape_s} uint8 volume per GPU resident in HBM, ") +
                                    f"{blocks_all} blocks 196^3 -> 104^3 through MTLSDModel(20 affinities + 10 LSDs), uint8 affinities left on the device, "
                                    "rg_mws_prepare + mutex watershed (20 offsets, sep 3, strides 10, mask 0.5)" +
                                    ("; first level per rank, one boundary exchange, replicated second level (segment_slabs)" if sharded else ""),
                           parallelism=(f"one volume over {world} GPUs: Z parts of {[p['own'][1] - p['own'][0] for p in pipeline.shard_plan(gshape, world)]} planes"
                                        if sharded else f"one volume per GPU x{world}"), global_shape=list(gshape)),
               ranks_seen=ranks_seen, backend=backend,
               blocks_per_s=round(blocks_all * (1 if sharded else world) / (tp / steps), 2), predict_s=round(tp / steps, 4), segment_s=round(ts / steps, 4),
               mvoxels_per_s=round(voxels / sec / 1e6, 2), segments=nseg, raw_network_case=raw_case,
               affinity_shaping="SHAPED input: network output blended 20/80 with a both-voxels-inside-a-blob structure term per offset (bench.py main_config5.shaped); "
                                "the unshaped figure is `raw_network_case`",
               roofline=dict(bound="mfma", kernel=dom[0], achieved=round(ach, 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_BF16_TFLOPS, 4), traffic=None,
                             launches_per_step=dom[1]["launches"] // steps, avg_launch_ms=round(dom[1]["ms"] / dom[1]["launches"], 4)),
               roofline_segment=dict(bound="hbm", kernel="mws_rounds_kernel", achieved=None if hbm is None else round(hbm, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                                     frac=None if hbm is None else round(hbm / HBM_PEAK_GBS, 5), traffic=None, ms_per_volume=round(mws_ms, 3),
                                     note="rank 0's part; 88 B/voxel algorithmic; round-latency bound (dependency-ordered union-find), see --workload segment"))
    print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def cell_affs(size, seed, device):
    """Synthetic (3, Z, Y, X) nearest-neighbour affinities: the foreground is a union of overlapping balls (radius 10 .. 16, about 1.5 per 32^3) — so the boundary
    distance has one maximum per ball, the watershed cuts the necks between overlapping balls and the agglomeration has touching fragments to merge — with affinity
    0.85 .. 1.0 between two foreground voxels and 0.1 .. 0.25 elsewhere (15 % uniform noise)."""
    g = torch.Generator(device="cpu").manual_seed(2000 + seed)
    nb = max(4, int(1.5 * (size / 32) ** 3))
    c = (torch.rand((nb, 3), generator=g) * size).to(device)
    rad = (10 + 6 * torch.rand(nb, generator=g)).to(device)
    r = torch.arange(size, device=device, dtype=torch.float32)
    fg = torch.zeros((size,) * 3, dtype=torch.bool, device=device)
    for i in range(nb):
        d2 = ((r - c[i, 0]) ** 2)[:, None, None] + ((r - c[i, 1]) ** 2)[None, :, None] + ((r - c[i, 2]) ** 2)[None, None, :]
        fg |= d2 < rad[i] ** 2
    noise = torch.rand((3,) + (size,) * 3, generator=torch.Generator(device=device).manual_seed(3000 + seed), device=device) if device != "cpu" else \
        torch.rand((3,) + (size,) * 3, generator=torch.Generator().manual_seed(3000 + seed))
    return torch.stack([(fg & torch.roll(fg, 1, ax)).float() for ax in range(3)]) * 0.75 + 0.1 + 0.15 * noise


def main_watershed(a, rank, local, world, dev):
    """--workload watershed: the reference's DEFAULT raygun-segment route (segment.py:116-136 get_segmentation with its default thresholds 0.1 .. 0.8) on a
    synthetic (3, S, S, S) affinity volume resident in HBM: mask -> exact EDT -> seeds -> seeded watershed -> region graph -> agglomeration (host) -> one
    relabelled uint64 volume per threshold, all left on the device.  value = input Mvoxels/s; stage times from HIP events (the agglomeration is host time)."""
    import numpy as np
    from raygun_amd import watershed as W
    size = a.size or 256
    affs = cell_affs(size, rank, dev)
    thresholds = [float(t) for t in np.arange(0.1, 0.9, 0.1)]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def step(marks=None):
        ev = lambda: (lambda e: (e.record(), e)[1])(torch.cuda.Event(enable_timing=True))   # noqa: E731
        t = [ev()]
        bm = W.boundary_mask(affs, 1.0, dev); d2 = W.squared_distance_transform(bm, dev); t.append(ev())
        seeds, n = W.seeds_from_distance(d2, 10); t.append(ev())
        st = {}
        frag = W.flood(d2, seeds, bm, st); t.append(ev())
        h0 = time.perf_counter()
        ast = {}
        segs = list(W.agglomerate(affs, frag, thresholds, as_tensor=True, device=dev, stats=ast)); t.append(ev())
        if marks is not None:
            torch.cuda.synchronize()
            marks.append(dict(mask_edt_ms=t[0].elapsed_time(t[1]), seeds_ms=t[1].elapsed_time(t[2]), flood_ms=t[2].elapsed_time(t[3]),
                              region_graph_agglomerate_relabel_ms=(time.perf_counter() - h0) * 1e3, flood_rounds=st["rounds"], seeds=n, **ast))
        return segs
    for _ in range(max(1, a.warmup)):
        segs = step()
    barrier()
    marks = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        segs = step(marks)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        if rank != 0:
            torch.distributed.destroy_process_group()
            return
    ms = dt / a.steps * 1e3
    stages = {k: round(float(np.median([m[k] for m in marks])), 3) for k in marks[0] if k.endswith("_ms")}
    frag_ms = stages["mask_edt_ms"] + stages["seeds_ms"] + stages["flood_ms"]
    alg = size ** 3 * 16          # 3 float32 affinity channels read + one int32 fragment id written per voxel
    res = dict(metric="Mvoxels/sec default raygun-segment route (seeded watershed + agglomeration, 8 thresholds)", value=round(size ** 3 * world / (dt / a.steps) / 1e6, 2),
               unit="Mvoxels/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms, 3), higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype="f32 affinities / i32 squared distances / i32 union-find", data="synthetic",
               config=dict(workload=f"watershed {size}^3 per GPU: synthetic 3-channel cell affinities, segment.py:116-136 get_segmentation, thresholds 0.1 .. 0.8, "
                                    "segmentations left in HBM", parallelism=f"replicas x{world}"),
               stages_ms=stages, fragments=marks[-1]["fragments"], seeds=marks[-1]["seeds"], flood_rounds=marks[-1]["flood_rounds"], region_graph_records=marks[-1]["records"],
               segments_per_threshold=[int(torch.unique(s_).numel()) - 1 for s_ in segs],
               roofline=dict(bound="hbm", kernel="fragments (ws_mask + ws_edt + ws_maxfilter + ws_ccl + ws_flood passes)", achieved=round(alg / (frag_ms * 1e-3) / 1e9, 2),
                             peak=HBM_PEAK_GBS, unit="GB/s", frac=round(alg / (frag_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), traffic=None, avg_launch_ms=round(frag_ms, 3),
                             note="algorithmic bytes = 16 B/voxel (3 fp32 affinities in, 1 int32 fragment id out); the stage is ~25 streaming passes over 4-byte arrays "
                                  "(3 EDT, 4 max filter, 7 labelling, 4 per flood round)"))
    if world == 1 and not a.no_cpu_baseline:
        try:
            from oracle import watershed as OW
            cs = a.cpu_size or 128
            ca = cell_affs(cs, 0, "cpu").numpy()
            t1 = time.time()
            OW.get_segmentation(ca, thresholds, max_affinity_value=1.0)
            ct = time.time() - t1
            res["cpu_baseline"] = dict(value=round(cs ** 3 / ct / 1e6, 3), unit="Mvoxels/s", cores=1, kind="port",
                                       sample=f"oracle/watershed.py (scipy EDT / maximum_filter / label + the C restatement of the build-defined flood + the Python "
                                              f"restatement of waterz's merging) on the same synthetic affinities at {cs}^3: {ct:.1f} s")
        except Exception as e:
            res["cpu_baseline"] = dict(error=repr(e))
    print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def cpu_baseline_aux(workload, size):
    import numpy as np
    cores, cpu = host_cpu()
    torch.set_num_threads(cores)
    if workload == "segment":
        from oracle import mws as OM
        size = size or 192
        affs = structured_affs(size, 0, "cpu").numpy()
        t0 = time.time()
        w, m = OM.prepare(affs)
        OM.compute_mws_segmentation(w, OM.neighborhood(8).astype(np.int32), 3, (10, 10, 10), m)
        dt = time.time() - t0
        return dict(value=size ** 3 / dt / 1e6, unit="Mvoxels/s", cores=1, kind="port", cpu=cpu,
                    sample=f"numpy prepare + oracle/mws.c (sequential Kruskal, as affogato is single-threaded) on the same synthetic affinities at {size}^3, {dt:.1f} s")
    from oracle import torch_ref as O
    from raygun_amd import synth
    m = O.MTLSDModel(num_affs=20).eval()
    x = torch.from_numpy(synth.uniform((1, 1, 196, 196, 196), 42, 0, 1))
    with torch.no_grad():
        dt, n, warm = timed_median(lambda: m(x), budget_s=30.0)
    return dict(value=104 ** 3 / dt / 1e6, unit="Mvoxels/s", cores=cores, kind="port", cpu=cpu,
                sample=f"forward of the CPU restatement on one 196^3 block (torch {torch.__version__} fp32, oneDNN): 1 warm-up + median of {n} = {dt:.1f} s per block")


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N supervised rank processes (raygun_amd.parallel.launch_ranks:
    torchrun-style environment, rendezvous on 127.0.0.1, backend nccl = RCCL) from this parent, which has not touched the GPU.
    Rank 0's stdout (the JSON line) is ours; a failing rank or the overall timeout (--rank-timeout, default 900 s) ends all ranks
    and the exit code is non-zero."""
    from raygun_amd.parallel import launch_ranks as launch
    return launch([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], a.gpus, timeout=a.rank_timeout)


def _backend_string():
    b = torch.distributed.get_backend()
    if b == "nccl":     # (= RCCL on ROCm)
        try:
            b += " (RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version()) + ")"
        except Exception:
            b += " (RCCL)"
    return b


def run_train(a, workload, size, steps, warmup, rank, world, dev, precision="bf16", identity=0.0):
    """Time `steps` train steps of `workload` ('unet' | 'cyclegan') after `warmup` untimed ones: barrier + synchronize on both
    sides, MAX over ranks.  Returns the result dict (rank 0) or None."""
    import contextlib
    from raygun_amd import ops, parallel
    with contextlib.redirect_stdout(sys.stderr):  # init_weights prints like the reference's; stdout carries the JSON line only
        step, info = build_unet(dev, size, seed=rank, precision=precision) if workload == "unet" else build_cyclegan(dev, size, seed=rank, precision=precision, identity=identity)
    sync = parallel.GradSync(prescaled=(workload == "unet"), timing=world > 1)   # algo: RAYGUN_AMD_GRADSYNC = allreduce (default) | rs_ag
    # The interpreter's full (generation-2) garbage collection walks every object the imports created — ~70 ms on this image — and,
    # landing in a timed step while the host is less than that ahead of the GPU, shows up as one 80-ms step (tools/step_trend.py).
    # Long-lived objects go to the permanent generation, as raygun_amd's Trainer does before its loop.
    import gc
    gc.collect(); gc.freeze()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Every conv launch of the LAST warm-up step is bracketed by HIP events (the per-kernel table below); inside the timed region only
    # the dominant kernel of that table is (its roofline numbers are measured live there): an event pair costs ~50 us of host time and
    # stops neighbouring kernels from overlapping — fully instrumented the step measures ~4 % slower.
    table, select = None, None
    for i in range(warmup):
        if i == warmup - 1:
            ops.profile_start()
        step(sync)
    if warmup:
        table = ops.profile_stop()
        if table:
            select = ops.profile_signatures().get(max(table.items(), key=lambda kv: kv[1]["ms"])[0])
    barrier()
    graph_mode = bool(getattr(step, "graphed", False))   # the timed steps are graph replays: no per-kernel events in there (the roofline falls back to the warm-up table)
    if not graph_mode:
        ops.profile_start(select)
    t0 = time.perf_counter()
    opt_marks = [] if workload == "unet" else None
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]   # per-step spread (reported beside the contract's mean)
    step_ev[0].record()
    for i in range(steps):
        if opt_marks is not None:
            step(sync, opt_marks)
        else:
            step(sync)
        step_ev[i + 1].record()
    t_enq = time.perf_counter() - t0  # host time to enqueue the steps (the GPU runs behind); close to dt = launch-bound
    barrier()
    dt = time.perf_counter() - t0
    prof = ops.profile_stop() if not graph_mode else {}
    waits = sync.wait_ms() if world > 1 else []   # per GradSync call: what the compute stream waited for the gradient exchange (not hidden by backward)
    comm = dict(ranks_seen=torch.distributed.get_world_size() if world > 1 else 1,
                backend=_backend_string() if world > 1 else "none (single process)",
                gradsync_algo=sync.algo, gradsync_calls_per_step=round(len(waits) / max(1, steps), 2),
                allreduce_wait_ms_per_step=round(sum(waits) / max(1, steps), 3) if waits else 0.0,
                allreduce_wait_ms_max_step=round(max(waits), 3) if waits else 0.0,
                collectives_last_step=dict(sync.stats))
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return None
    ms = dt / steps * 1e3
    tab, tsteps = (table, 1) if table else (prof, steps)  # per-kernel table: the instrumented warm-up step (all conv launches)
    kernels = {k: dict(launches=v["launches"] // tsteps, ms_per_step=v["ms"] / tsteps, tflops=v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0.0)
               for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["ms"])}
    dom, dsteps = (max(tab.items(), key=lambda kv: kv[1]["ms"]) if tab else None), tsteps
    if dom and dom[0] in prof:
        dom, dsteps = (dom[0], prof[dom[0]]), steps   # the dominant kernel's launches of the TIMED steps
    roofline = None
    if dom:
        ach = dom[1]["flops"] / (dom[1]["ms"] * 1e-3) / 1e12
        roofline = dict(bound="mfma", kernel=dom[0], achieved=round(ach, 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                        frac=round(ach / PEAK_BF16_TFLOPS, 4), traffic=pmc_traffic(dom[0]) if workload == "unet" else None,
                        traffic_note="HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE) from the committed profiles/*_pmc_*.csv passes of this workload",
                        launches_per_step=dom[1]["launches"] // dsteps, avg_launch_ms=round(dom[1]["ms"] / dom[1]["launches"], 4),
                        measured="HIP events around this kernel's launches inside the timed region; the `kernels` table is the fully instrumented last warm-up step"
                        if not graph_mode else "HIP events in the instrumented (eager) last warm-up step; the timed steps are HIP-graph replays")
    flop = (UNET_FLOP_PER_STEP if workload == "unet" else (CYC_FLOP_IDENTITY if identity else CYC_FLOP_PER_STEP)) * (size / (196 if workload == "unet" else 128)) ** 3
    executed = sum(v["flops"] for v in tab.values()) / tsteps  # what the conv / weight-gradient launches actually multiplied
    res = dict(metric=f"Mvoxels/sec 3D UNet fwd+bwd 196^3 {precision}" if workload == "unet" else "CycleGAN 128^3 train-step ms",
               value=round(size ** 3 * world / (dt / steps) / 1e6, 3) if workload == "unet" else round(ms, 3),
               unit="Mvoxels/s" if workload == "unet" else "ms/step", n_gpus=world, steps=steps, warmup=warmup,
               ms_per_step=round(ms, 3), higher_is_better=workload == "unet", scaling="weak", vs_baseline=None, dtype=precision,
               data="synthetic",
               config=dict(workload=f"{workload} {size}^3 batch 1 per GPU, {info['model']}", params=info["params"],
                           parallelism=f"dp{world}", precision=f"{precision} storage, fp32 accumulate, fp32 master weights + Adam state"
                           + (", dynamic loss scaling" if precision == "f16" else "")),
               **(dict(loss_scale=info["scaler"].scale_value, skipped_steps=info["scaler"].skipped) if info.get("scaler") is not None else {}),
               # algorithmic = the reference network's FLOPs / step time; executed = the FLOPs of the launches the
               # graph rewrites leave, summed over the step's conv + weight-gradient kernels / step time
               algorithmic_tflop_per_step=round(flop / 1e12, 3), executed_tflop_per_step=round(executed / 1e12, 3),
               step_tflops=round(flop / (dt / steps) / 1e12, 2), executed_tflops=round(executed / (dt / steps) / 1e12, 2),
               step_frac_of_mfma_peak=round(flop / (dt / steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
               host_enqueue_ms_per_step=round(t_enq / steps * 1e3, 3),
               step_ms_each=[round(step_ev[i].elapsed_time(step_ev[i + 1]), 2) for i in range(steps)],
               ms_per_step_median=round(sorted(step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(steps))[steps // 2], 3),
               **(dict(optimizer_ms=round(sum(a.elapsed_time(b) for a, b in opt_marks) / len(opt_marks), 3),
                       fwd_bwd_ms=round(ms - sum(a.elapsed_time(b) for a, b in opt_marks) / len(opt_marks), 3)) if opt_marks else {}),
               executed_frac_of_mfma_peak=round(executed / (dt / steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
               comm=comm, roofline=roofline, kernels=kernels, **(dict(hip_graph="timed steps are replays of one captured step (raygun_amd/graphs.py)") if graph_mode else {}))
    return res


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:  # no launcher: be one (before anything touches the GPU)
        sys.exit(launch_ranks(a))
    if os.environ.get("RAYGUN_AMD_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RAYGUN_AMD_BENCH_WATCHDOG"]), exit=True)
    from raygun_amd import parallel
    rank, local, world = parallel.init_from_env()
    if world != max(1, a.gpus) and rank == 0:
        print(f"[bench] warning: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if a.workload == "config5":
        return main_config5(a, rank, local, world, dev)
    if a.workload in ("predict", "segment"):
        return main_aux(a, rank, local, world, dev)
    if a.workload == "watershed":
        return main_watershed(a, rank, local, world, dev)
    size = a.size or (196 if a.workload == "unet" else 128)
    res = run_train(a, a.workload, size, a.steps, a.warmup, rank, world, dev, precision=a.precision, identity=a.identity if a.workload == "cyclegan" else 0.0)
    if a.workload == "unet" and a.precision == "bf16" and world == 1 and not a.no_secondary:
        # the same step with IEEE-half storage and dynamic loss scaling: the mode whose trained precision meets north_star's 1e-3
        # (DESIGN.md 4); reported beside the bf16 headline, never instead of it
        import gc
        gc.collect(); torch.cuda.empty_cache()
        h = run_train(a, "unet", size, max(3, a.steps // 2), max(3, a.warmup), rank, world, dev, precision="f16")
        if res is not None and h is not None:
            res["f16_storage"] = {k: h[k] for k in ("value", "unit", "ms_per_step", "fwd_bwd_ms", "optimizer_ms", "steps", "warmup", "dtype", "loss_scale",
                                                     "skipped_steps") if k in h}
            res["f16_storage"]["note"] = ("same network, step and sizes with f16 storage + dynamic loss scaling (raygun_amd/amp.py); optimizer_ms "
                                          "includes the gradient inf check and its 4-byte host read")
        del h
    if a.workload == "unet" and not a.no_secondary:
        # BASELINE's metric string has a second half — "CycleGAN 128^3 train-step ms" (config 2; with N > 1: config 4): timed in
        # the same run, after the U-Net's memory is released, and reported inside the same JSON line
        import gc
        gc.collect(); torch.cuda.empty_cache()
        sec = run_train(a, "cyclegan", 128, max(3, a.steps), max(3, a.warmup), rank, world, dev)   # >= 20 steps after 5 warm-up
        if res is not None and sec is not None:
            res["secondary"] = dict(metric=sec["metric"], value=sec["value"], unit=sec["unit"], ms_per_step=sec["ms_per_step"], ms_per_step_median=sec["ms_per_step_median"],
                                    steps=sec["steps"], warmup=sec["warmup"], higher_is_better=False, config=sec["config"],
                                    algorithmic_tflop_per_step=sec["algorithmic_tflop_per_step"], executed_tflop_per_step=sec["executed_tflop_per_step"],
                                    algorithmic_tflops=sec["step_tflops"], executed_tflops=sec["executed_tflops"],
                                    executed_frac_of_mfma_peak=sec["executed_frac_of_mfma_peak"],
                                    roofline={k: sec["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches_per_step", "avg_launch_ms")}
                                    if sec["roofline"] else None, comm=sec["comm"], n_gpus=sec["n_gpus"],
                                    kernel_launches_per_step=sum(k["launches"] for k in sec["kernels"].values()))
        if world == 1:
            # SURVEY 8d config 2 asks for the identity-0.5 variant (the reference's own default_cycleGAN_conf.json) beside the headline configuration, and the
            # f16-storage step with dynamic loss scaling (the mode whose precision meets north_star's 1e-3): shorter runs, same line
            for key, kw in (("identity_0.5", dict(identity=0.5)), ("f16_storage", dict(precision="f16"))):
                gc.collect(); torch.cuda.empty_cache()
                v = run_train(a, "cyclegan", 128, max(3, a.steps // 2), 3, rank, world, dev, **kw)
                if res is not None and v is not None and "secondary" in res:
                    res["secondary"][key] = dict(ms_per_step=v["ms_per_step"], ms_per_step_median=v["ms_per_step_median"], steps=v["steps"], warmup=v["warmup"],
                                                 algorithmic_tflop_per_step=v["algorithmic_tflop_per_step"], algorithmic_tflops=v["step_tflops"],
                                                 executed_tflop_per_step=v["executed_tflop_per_step"], workload=v["config"]["workload"],
                                                 **({k: v[k] for k in ("loss_scale", "skipped_steps") if k in v}))
                del v
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if world == 1 and not a.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline(a.workload, a.cpu_size or (196 if a.workload == "unet" else 48))
        except Exception as e:  # the baseline is reported, never the product path
            res["cpu_baseline"] = dict(error=repr(e))
    print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
