"""Block-prediction worker — mirror of raygun.torch.predict.worker.worker (reference: src/raygun/torch/predict/worker.py:13-160).

What the reference does per worker process, and what happens here:
  * render config defaults + read (:19-45)                                   -> the same keys, same defaults;
  * `system = load_system(config_path)`, checkpoint discovery (:48-62: a `checkpoint` that is not an existing path is
    looked up as <dirname(config_path)>/<checkpoint_basename without leading "./">_checkpoint_<checkpoint>; None if that
    does not exist either), `system.load_saved_model(checkpoint_path)` (:62), net selection `getattr(system.model,
    net_name)` (:63-66), `model.eval()`, `.to("cuda")` (:68-71)             -> the same, in resolve_model();
  * source / destination datasets through daisy.open_ds (:75-92)             -> raygun_amd.zarrio.open_ds;
  * `while client.acquire_block()` (daisy's scheduler hands out blocks, :94-100) -> a static partition of the block list over
    ranks (parallel.shard_blocks; SURVEY.md 8e: no collective), processed by predict.predict_volume, which holds the
    per-block arithmetic of :101-157 (normalise, scaleShift_input, forward, crop, *iinfo.max + clamp for integer
    destinations, write to the block's write ROI).
"""
import logging
import os
import sys

import torch

from ...load_system import load_system
from ...read_config import read_config

RENDER_DEFAULTS = {"crop": 0, "read_size": None, "max_retries": 2, "num_workers": 16, "ndims": None, "net_name": None,
                   "scaleShift_input": None, "output_ds": None, "out_specs": None,
                   "precision": None}  # build-owned key: "f16" (default) | "bf16" storage of the forward pass (predict.predict_volume)


def read_render_config(render_config_path):
    """worker.py:19-45 / predict.py:27-55: defaults, then the file; returns (render_config, train_config)."""
    render_config = dict(RENDER_DEFAULTS)
    render_config.update(read_config(render_config_path))
    train_config = read_config(render_config["config_path"])
    if render_config["ndims"] is None:
        render_config["ndims"] = train_config["ndims"]
    return render_config, train_config


def destination_names(render_config):
    """(dest_path, output_ds) — worker.py:78-90 == predict.py:57-68."""
    config_path, source_path = render_config["config_path"], render_config["source_path"]
    dest_path = render_config["dest_path"] if "dest_path" in render_config else \
        os.path.join(os.path.dirname(config_path), os.path.basename(source_path))
    output_ds = render_config["output_ds"]
    if output_ds is None:
        net_name, checkpoint, ds = render_config["net_name"], render_config["checkpoint"], render_config["source_dataset"]
        output_ds = [f"{ds}_{net_name}_{checkpoint}"] if net_name is not None else [f"{ds}_{checkpoint}"]
    return dest_path, output_ds


def resolve_checkpoint(system, config_path, checkpoint):
    """worker.py:50-60, including its quirk: a `checkpoint` that IS an existing path yields None — the system then loads
    its own `self.checkpoint` (BaseSystem.load_saved_model :110-113), which the system found at construction."""
    if not os.path.exists(str(checkpoint)):
        checkpoint_path = os.path.join(os.path.dirname(config_path),
                                       system.checkpoint_basename.lstrip("./") + f"_checkpoint_{checkpoint}")
        if not os.path.exists(checkpoint_path):
            checkpoint_path = None
    else:
        checkpoint_path = None
    return checkpoint_path


def resolve_model(render_config, device=None):
    """worker.py:48-73: system -> checkpoint -> (sub)network -> eval -> GPU."""
    system = load_system(render_config["config_path"])
    system.load_saved_model(resolve_checkpoint(system, render_config["config_path"], render_config["checkpoint"]))
    model = getattr(system.model, render_config["net_name"]) if render_config["net_name"] is not None else system.model
    model.eval()
    if not torch.cuda.is_available():
        raise RuntimeError("raygun_amd.worker: no GPU — the block runner has no CPU path (the reference falls back to "
                           "torch.FloatTensor on the CPU, worker.py:103-104; here that is the oracle's job)")
    model.to(device if device is not None else "cuda")
    del system
    return model


def worker(render_config_path, rank=None, world=None, device=None):
    """Predict this rank's share of the blocks.  Returns (destination arrays, number of blocks processed)."""
    from ... import parallel, zarrio
    from ...predict import predict_volume
    logger = logging.getLogger(f"crop_worker_{0 if rank is None else rank}")
    render_config, train_config = read_render_config(render_config_path)
    model = resolve_model(render_config, device)
    source = zarrio.open_ds(render_config["source_path"], render_config["source_dataset"])
    dest_path, output_ds = destination_names(render_config)
    destinations = [zarrio.open_ds(dest_path, name, "a") for name in output_ds]
    shapes = shape_args(render_config, train_config)
    dests, n = predict_volume(model, source, ndims=render_config["ndims"], scaleShift_input=render_config["scaleShift_input"],
                              out_dtypes=[d.dtype for d in destinations], rank=rank, world=world, device=device,
                              destinations=destinations, precision=render_config["precision"], **shapes)
    logger.info(f"Wrote {n} blocks to {', '.join(output_ds)}.")
    return dests, n


def shape_args(render_config, train_config):
    """predict.py:73-108: input/output_shape from the render config, else from the train config; else read_size (or the
    train config's side_length) and crop."""
    if "input_shape" in render_config or "input_shape" in train_config:
        try:
            return dict(input_shape=render_config["input_shape"], output_shape=render_config["output_shape"])
        except KeyError:
            return dict(input_shape=train_config["input_shape"], output_shape=train_config["output_shape"])
    read_size = render_config["read_size"]
    if read_size is None:
        read_size = train_config["side_length"]
    return dict(read_size=read_size, crop=render_config["crop"])


if __name__ == "__main__":
    worker(sys.argv[1])
