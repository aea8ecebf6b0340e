"""Entry points named by the north_star — raygun-train / raygun-predict / load_system (reference: src/raygun/train.py:52-74,
load_system.py:9-21, predict.py:16-192, torch/predict/worker.py:13-160).  CPU tests pin the config handling against the
reference's own config files' shape (default_MTLSD_conf.json, experiments/.../predict_eval0_config.json); the GPU test
drives train(config) -> checkpoint -> load_saved_model -> predict(render_config) end to end on a small MTLSD system."""
import json
import os

import numpy as np
import pytest
import torch

SMALL_UNET = {"input_nc": 1, "ngf": 4, "fmap_inc_factor": 2, "downsample_factors": [[2, 2, 2]], "constant_upsample": True}


def _train_conf(tmp_path, **over):
    cfg = {"framework": "torch", "system": "MTLSD", "ndims": 3, "batch_size": 1, "synthetic_data": True,
           "model_kwargs": {"unet_kwargs": SMALL_UNET, "num_affs": 3},
           "input_shape": [36, 36, 36], "output_shape": [20, 20, 20], "voxel_size": [4, 4, 4],
           "optim_type": "Adam", "optim_kwargs": {"lr": 1e-3},
           "train_kwargs": {"checkpoint_basename": "./models/model", "num_epochs": 3, "save_every": 3, "log_every": 1},
           "random_seed": 7}
    cfg.update(over)
    p = tmp_path / "train_conf.json"
    p.write_text("// comments are allowed (jsmin in the reference)\n" + json.dumps(cfg, indent=1))
    return str(p)


def test_load_system_and_render_config_handling(tmp_path, monkeypatch):
    from raygun_amd.load_system import load_system
    import raygun_amd.torch.predict.worker as W  # noqa: F401
    import importlib
    W = importlib.import_module("raygun_amd.torch.predict.worker")
    from raygun_amd.torch.systems.MTLSD import MTLSD
    conf = _train_conf(tmp_path)
    monkeypatch.chdir(tmp_path)
    s = load_system(conf)
    assert isinstance(s, MTLSD) and s.checkpoint is None and s.model_kwargs["num_affs"] == 3
    assert s.neighborhood.shape == (20, 3)
    # render config: the reference's keys and defaults (worker.py:19-45), shapes from the train config when absent
    render = {"config_path": conf, "source_path": str(tmp_path / "vol.zarr"), "source_dataset": "volumes/raw", "checkpoint": "3",
              "output_ds": ["pred_affs", "pred_lsds"],
              "out_specs": {"pred_affs": {"dtype": "#numpy.float32#", "num_channels": 3}, "pred_lsds": {"dtype": "#numpy.float32#", "num_channels": 10}},
              "scaleShift_input": [2, -1]}
    rp = tmp_path / "render.json"; rp.write_text(json.dumps(render))
    rc, tc = W.read_render_config(str(rp))
    assert rc["crop"] == 0 and rc["num_workers"] == 16 and rc["ndims"] == 3 and rc["net_name"] is None
    assert rc["out_specs"]["pred_affs"]["dtype"] is np.float32          # "#numpy.float32#" is eval'd (read_config.py)
    assert W.shape_args(rc, tc) == dict(input_shape=[36, 36, 36], output_shape=[20, 20, 20])
    assert W.shape_args(dict(rc, input_shape=196, output_shape=104), tc) == dict(input_shape=196, output_shape=104)
    assert W.shape_args(dict(rc, read_size=None), {"side_length": 64, "ndims": 3}) == dict(read_size=64, crop=0)
    # destination naming (worker.py:78-90)
    assert W.destination_names(rc) == (str(tmp_path / "vol.zarr"), ["pred_affs", "pred_lsds"])
    d, names = W.destination_names(dict(rc, output_ds=None, net_name="netG2", dest_path="/x/y.zarr"))
    assert (d, names) == ("/x/y.zarr", ["volumes/raw_netG2_3"])
    assert W.destination_names(dict(rc, output_ds=None))[1] == ["volumes/raw_3"]
    # checkpoint discovery (worker.py:50-60): <dirname(config)>/<basename without './'>_checkpoint_<n>, None if absent,
    # and None for a `checkpoint` that is itself an existing path (the system then loads its own self.checkpoint)
    s.checkpoint_basename = "./models/model"
    assert W.resolve_checkpoint(s, conf, "3") is None
    os.makedirs(tmp_path / "models"); (tmp_path / "models" / "model_checkpoint_3").write_bytes(b"x")
    assert W.resolve_checkpoint(s, conf, "3") == str(tmp_path / "models" / "model_checkpoint_3")
    assert W.resolve_checkpoint(s, conf, str(tmp_path / "models" / "model_checkpoint_3")) is None
    # latest-checkpoint discovery at construction (BaseSystem.py:125-143), natural sort
    (tmp_path / "models" / "model_checkpoint_20").write_bytes(b"x")
    s2 = load_system(conf)
    assert s2.checkpoint.endswith("model_checkpoint_20") and s2.iteration == 20


def test_train_without_gpu_raises(tmp_path, monkeypatch):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from raygun_amd.train import train
    monkeypatch.chdir(tmp_path)
    with pytest.raises(RuntimeError, match="MI355X"):
        train(_train_conf(tmp_path))


def test_trainer_checkpoint_schema_cpu(tmp_path):
    """Trainer.save_checkpoint / load_checkpoint on plain CPU modules: gunpowder's keys (model_state_dict,
    optimizer_state_dict) + iteration; a BaseDummyOptimizer's real optimisers and schedulers are saved too."""
    from raygun_amd.torch.optimizers import BaseDummyOptimizer
    from raygun_amd.torch.train import Trainer

    class Sys:
        pass
    s = Sys()
    s.model = torch.nn.Linear(3, 2)
    og = torch.optim.Adam(s.model.parameters(), lr=1e-2)
    s.optimizer = BaseDummyOptimizer(optimizer_G=og, scheduler="LambdaLR",
                                     scheduler_kwargs={"epoch_count": 0, "n_epochs": 2, "n_epochs_decay": 2})
    s.loss = None; s.checkpoint_basename = str(tmp_path / "models" / "m"); s.iteration = 0
    s.model(torch.ones(1, 3)).sum().backward(); og.step(); s.optimizer.step()
    t = Trainer(s); t.iteration = 5
    path = t.save_checkpoint()
    assert path.endswith("m_checkpoint_5")
    ck = torch.load(path, weights_only=False)
    assert set(ck) >= {"model_state_dict", "optimizer_state_dict", "iteration", "optimizers", "schedulers"}
    assert ck["optimizer_state_dict"] == {} and "optimizer_G" in ck["optimizers"]   # the dummy optimiser holds no parameters
    w0 = s.model.weight.detach().clone()
    m0 = og.state[s.model.weight]["exp_avg"].clone()
    with torch.no_grad():
        s.model.weight.zero_()
    og.state[s.model.weight]["exp_avg"].zero_()
    t.iteration = 0
    t.load_checkpoint(path)
    assert t.iteration == 5 and torch.equal(s.model.weight, w0) and torch.equal(og.state[s.model.weight]["exp_avg"], m0)
    assert s.optimizer.schedulers["optimizer_G"].last_epoch == 1


@pytest.mark.gpu
def test_train_checkpoint_predict_round_trip(tmp_path, monkeypatch):
    """raygun-train conf.json (3 iterations on synthetic batches, checkpoint at 3) -> a fresh system discovers and loads the
    checkpoint (weights + Adam moments) -> raygun-predict render.json over a zarr volume == predict_volume with the trained
    model, bit for bit; a second train() resumes at iteration 3."""
    from raygun_amd import predict as P, synth, zarrio
    from raygun_amd.load_system import load_system
    from raygun_amd.train import train
    monkeypatch.chdir(tmp_path)
    conf = _train_conf(tmp_path)
    loss, outs = train(conf)
    ck_path = tmp_path / "models" / "model_checkpoint_3"
    assert ck_path.exists() and torch.isfinite(loss) and set(outs) == {"pred_affs", "pred_lsds"}
    ck = torch.load(str(ck_path), weights_only=False)
    assert ck["iteration"] == 3 and len(ck["optimizer_state_dict"]["state"]) == len(ck["model_state_dict"])
    s = load_system(conf)
    assert s.checkpoint.endswith("model_checkpoint_3") and s.iteration == 3
    s.build_system()                                   # restores weights + optimiser moments
    sd = s.model.state_dict()
    for k, v in ck["model_state_dict"].items():
        assert torch.equal(sd[k].cpu(), v.cpu()), k
    st = s.optimizer.state_dict()["state"]
    assert float(st[0]["step"]) == 3.0 and float(st[0]["exp_avg"].abs().sum()) > 0
    assert s.trainer.iteration == 3 and s.trainer.train(3, batches=s.batches) is None   # nothing left to do
    # predict through the render config
    vol = (synth.uniform((60, 60, 60), 4, 0, 1) * 255).astype(np.uint8)
    zpath = str(tmp_path / "vol.zarr")
    raw = zarrio.prepare_ds(zpath, "volumes/raw", vol.shape, (4, 4, 4), np.uint8, write_size=(20, 20, 20), offset=(8, 0, 4), compressor=None)
    raw[...] = vol
    render = {"config_path": conf, "source_path": zpath, "source_dataset": "volumes/raw", "checkpoint": "3",
              "output_ds": ["pred_affs", "pred_lsds"], "dest_path": zpath,
              "out_specs": {"pred_affs": {"dtype": "#numpy.float32#", "num_channels": 3}, "pred_lsds": {"dtype": "#numpy.uint8#", "num_channels": 10}},
              "scaleShift_input": [2, -1]}
    rp = tmp_path / "render.json"; rp.write_text(json.dumps(render))
    dests = P.predict(str(rp))
    a = zarrio.open_ds(zpath, "pred_affs"); l = zarrio.open_ds(zpath, "pred_lsds")
    assert a.shape == (3, 60, 60, 60) and a.dtype == np.float32 and a.chunks == (3, 20, 20, 20) and a.offset == (8, 0, 4)
    assert l.shape == (10, 60, 60, 60) and l.dtype == np.uint8
    mem, n = P.predict_volume(s.model.eval(), vol, input_shape=[36] * 3, output_shape=[20] * 3, scaleShift_input=[2, -1],
                              out_dtypes=[np.float32, np.uint8])
    assert n == 8 and np.array_equal(a[...], mem[0]) and np.array_equal(l[...], mem[1]) and mem[0].any()
    assert os.path.exists(os.path.join(zpath, "view_vol.ng"))


@pytest.mark.gpu
def test_train_f16_precision_with_loss_scaling_through_the_entry_point(tmp_path, monkeypatch):
    """raygun-train on a config with "precision": "f16" (+ "loss_scale_kwargs"): the Trainer wraps forward + loss in IEEE-half storage, scales the
    loss, FusedAdam steps with 1/scale folded in; the checkpoint carries the scaler; a resumed system restores it.  An initial scale far beyond
    half's range must be backed off (skipped steps) until the step applies."""
    from raygun_amd.load_system import load_system
    from raygun_amd.train import train
    monkeypatch.chdir(tmp_path)
    conf = _train_conf(tmp_path, precision="f16", loss_scale_kwargs={"init_scale": 2.0 ** 14, "growth_interval": 2},
                       train_kwargs={"checkpoint_basename": "./models/model", "num_epochs": 4, "save_every": 4, "log_every": 1})
    loss, outs = train(conf)
    assert torch.isfinite(loss) and outs["pred_affs"].dtype == torch.float32
    ck = torch.load(str(tmp_path / "models" / "model_checkpoint_4"), weights_only=False)
    assert ck["iteration"] == 4 and ck["loss_scaler"]["skipped"] == 0
    assert ck["loss_scaler"]["scale"] == 2.0 ** 16                      # doubled after every 2 clean steps
    assert float(ck["optimizer_state_dict"]["state"][0]["step"]) == 4.0
    s = load_system(conf)
    s.build_system()
    assert s.trainer.scaler is not None and s.trainer.scaler.scale_value == 2.0 ** 16 and s.trainer.iteration == 4
    # a scale that overflows half: backed off step by step, Adam's step count only advances when a step is applied
    d2 = tmp_path / "big"; d2.mkdir(); monkeypatch.chdir(d2)
    conf2 = _train_conf(d2, precision="f16", loss_scale_kwargs={"init_scale": 2.0 ** 40, "max_scale": 2.0 ** 41, "growth_interval": 1000},
                        train_kwargs={"checkpoint_basename": "./models/model", "num_epochs": 30, "save_every": 30, "log_every": 1})
    train(conf2)
    ck2 = torch.load(str(d2 / "models" / "model_checkpoint_30"), weights_only=False)
    sk = ck2["loss_scaler"]["skipped"]
    assert 1 <= sk < 30 and ck2["loss_scaler"]["scale"] == 2.0 ** (40 - sk)
    assert float(ck2["optimizer_state_dict"]["state"][0]["step"]) == 30 - sk


@pytest.mark.gpu
def test_train_mtlsd_from_label_sources(tmp_path, monkeypatch):
    """raygun-train on an MTLSD config WITH `sources` (label volumes) and no "synthetic_data": the batches come from
    MTLSDCropBatches — random crops + GPU-generated LSDs / affinities / balance weights (reference flow
    io/MTLSDDataPipe.py:99-131 -> BaseTrain) — for 8 iterations."""
    from raygun_amd import synth, zarrio
    from raygun_amd.load_system import load_system
    monkeypatch.chdir(tmp_path)
    shape = (56, 56, 56)
    rng = np.random.default_rng(3)
    pts = rng.uniform(0, 56, (6, 3))
    g = np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"), -1).astype(np.float32)
    lab = (np.linalg.norm(g[..., None, :] - pts, axis=-1).argmin(-1) + 1).astype(np.uint64)
    raw = ((lab % 3) * 90 + 30 + synth.uniform(shape, 5, 0, 20)).astype(np.uint8)          # raw correlates with the segments
    zpath = str(tmp_path / "gt.zarr")
    for name, arr in (("volumes/raw", raw), ("volumes/labels", lab), ("volumes/mask", np.ones(shape, np.uint8))):
        zarrio.prepare_ds(zpath, name, shape, (4, 4, 4), arr.dtype, write_size=(28, 28, 28), compressor=None)[...] = arr
    conf = _train_conf(tmp_path, synthetic_data=False, n_diagonals=0, neighborhood=[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                       sources=[{"path": zpath, "raw": "volumes/raw", "labels": "volumes/labels", "labels_mask": "volumes/mask"}],
                       lsd_kwargs={"sigma": 12, "downsample": 2}, train_kwargs={"checkpoint_basename": "./models/model", "num_epochs": 8,
                                                                                "save_every": 100, "log_every": 1})
    s = load_system(conf)
    s.build_system()
    from raygun_amd.torch.train.data import MTLSDCropBatches
    assert isinstance(s.batches, MTLSDCropBatches)
    loss, outs = s.train()
    assert torch.isfinite(loss) and set(outs) == {"pred_affs", "pred_lsds"}
    assert s.trainer.iteration == 8
