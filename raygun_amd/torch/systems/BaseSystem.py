"""Config -> networks / model / optimisers / loss wiring — mirror of raygun.torch.systems.BaseSystem
(reference: systems/BaseSystem.py:20-336).  get_network / setup_* / load_saved_model / checkpoint discovery /
seeding keep the reference's behaviour; the gunpowder data pipes and Train node are replaced by
raygun_amd.torch.train (synthetic or in-memory arrays, same call order per step)."""
import functools
import logging
import os
import random
import re
from glob import glob

import numpy as np
import torch

from ...read_config import read_config
from .. import networks
from ..networks.utils import init_weights
from ... import hipnn, ops


class RaygunSequential(hipnn.FusedSequential):
    """nn.Sequential(net, 1x1 conv[, activation]) of get_network (BaseSystem.py:209-224) as one raygun_amd scope.

    When `net` ends in a plain stride-1 convolution (the generators do: Conv3d(ngf, ngf, 7), ResNet.py:338) that conv and
    the 1x1 conv are one linear map; they are evaluated as a single composed convolution (ops.conv3d_linear_pair: same
    parameters, same state_dict, gradients for all four tensors), 64x fewer FLOPs for ngf = 64, output_nc = 1.
    RAYGUN_AMD_FUSE_TAIL=0 evaluates the two layers one after the other."""
    _raygun_amd = True

    def _linear_tail(self):
        mods = list(self)
        if len(mods) < 2 or os.environ.get("RAYGUN_AMD_FUSE_TAIL", "1") == "0":
            return None
        inner = getattr(mods[0], "model", None)
        if not isinstance(inner, hipnn.FusedSequential) or len(inner) == 0:
            return None
        a, b = inner[-1], mods[1]
        both3 = isinstance(a, hipnn.Conv3d) and isinstance(b, hipnn.Conv3d)
        both2 = isinstance(a, hipnn.Conv2d) and isinstance(b, hipnn.Conv2d)     # the 2-D generators (ResNet.py:6-223): a (1, KH, KW) kernel
        if not (both3 or both2):
            return None
        if any(k != 1 for k in b.kernel_size) or any(v != 1 for v in b.stride) or any(v != 1 for v in a.stride):
            return None
        if a.padding_mode != "zeros" or a.groups != 1 or b.groups != 1 or any(d != 1 for d in a.dilation):
            return None
        if b.out_channels * a.kernel_size[-1] * 2 > a.out_channels:  # only when the composed map is much thinner
            return None
        return inner, a, b

    def run(self, x, residual=None, upto=None):
        tail = self._linear_tail() if upto is None and residual is None else None
        if tail is None or not getattr(type(list(self)[0]), "_raygun_tail_ok", False):
            return super().run(x, residual, upto)
        inner, a, b = tail
        mods = list(self)
        x = inner.run(ops.enter(x), upto=len(inner) - 1)
        act = hipnn.act_of(mods[2]) if len(mods) > 2 else None
        if isinstance(a, hipnn.Conv2d):
            x = ops.conv3d_linear_pair(x, a.weight.unsqueeze(2), a.bias, b.weight.unsqueeze(2), b.bias, (0,) + tuple(a._pads()),
                                       *(act if act is not None else ("none", 0.0)))
        else:
            x = ops.conv3d_linear_pair(x, a.weight, a.bias, b.weight, b.bias, a._pads(), *(act if act is not None else ("none", 0.0)))
        rest = mods[3:] if act is not None else mods[2:]
        for m in rest:
            x = m(x)
        return x


class BaseSystem:
    DEFAULTS = {}

    def __init__(self, default_config=None, config=None):
        for key, value in (default_config or self.DEFAULTS).items():
            setattr(self, key, value if not isinstance(value, (dict, list)) else _deepcopy(value))
        if config is not None:
            for key, value in read_config(config).items():
                setattr(self, key, value)
        if not hasattr(self, "checkpoint_basename"):
            try:
                self.checkpoint_basename = os.path.join(self.model_path, self.model_name)
            except Exception:
                self.checkpoint_basename = "./models/model"
        if not hasattr(self, "checkpoint") or self.checkpoint is None:
            try:
                self.checkpoint, self.iteration = self._get_latest_checkpoint()
            except Exception:
                self.checkpoint = None
        if getattr(self, "random_seed", None) is not None:
            self.set_random_seed()
        self.logger = logging.getLogger(type(self).__name__)

    def set_random_seed(self):
        """BaseSystem.py:78-83."""
        if self.random_seed is None:
            self.random_seed = 42
        torch.manual_seed(self.random_seed)
        random.seed(self.random_seed)
        np.random.seed(self.random_seed)

    def set_device(self, id=0):
        self.device_id = id
        os.environ["CUDA_VISIBLE_DEVICES"] = str(id)

    def load_saved_model(self, checkpoint=None, cuda_available=None):
        """BaseSystem.py:100-123: loads checkpoint['model_state_dict'] (or the bare state_dict) into self.model."""
        if not hasattr(self, "model"):
            self.setup_model()
        if cuda_available is None:
            cuda_available = torch.cuda.is_available()
        if checkpoint is None:
            checkpoint = self.checkpoint
        else:
            self.checkpoint = checkpoint
        if checkpoint is not None:
            ckpt = torch.load(checkpoint, map_location=None if cuda_available else torch.device("cpu"), weights_only=False)
            self.model.load_state_dict(ckpt["model_state_dict"] if "model_state_dict" in ckpt else ckpt)
        else:
            self.logger.warning("No saved checkpoint found.")

    def _get_latest_checkpoint(self):
        """BaseSystem.py:125-143: natural-sort glob of <basename>_checkpoint_<iter>."""
        basename = self.model_path + self.model_name

        def natural_keys(text):
            return [int(c) if c.isdigit() else c for c in re.split(r"(\d+)", text)]

        checkpoints = glob(basename + "_checkpoint_*")
        checkpoints.sort(key=natural_keys)
        if len(checkpoints) > 0:
            return checkpoints[-1], int(checkpoints[-1].split("_")[-1])
        return None, 0

    def get_downsample_factors(self, net_kwargs):
        """BaseSystem.py:145-161."""
        if "downsample_factors" not in net_kwargs:
            down_factor = 2 if "down_factor" not in net_kwargs else net_kwargs.pop("down_factor")
            num_downs = 3 if "num_downs" not in net_kwargs else net_kwargs.pop("num_downs")
            net_kwargs.update({"downsample_factors": [(down_factor,) * self.ndims] * (num_downs - 1)})
        return net_kwargs

    def get_network(self, net_type="unet", net_kwargs=None):
        """BaseSystem.py:163-242 (pops final_activation/output_nc from the dict it is given, like the reference)."""
        final_activation = net_kwargs.pop("final_activation") if "final_activation" in net_kwargs else None
        output_nc = net_kwargs.pop("output_nc") if "output_nc" in net_kwargs else net_kwargs["input_nc"]
        if isinstance(final_activation, str):
            final_activation = getattr(torch.nn, final_activation)
        add_final = True
        if net_type == "unet":
            net = networks.UNet(**self.get_downsample_factors(net_kwargs))
        elif net_type == "residualunet":   # BaseSystem.py:184-187
            net = networks.ResidualUNet(**self.get_downsample_factors(net_kwargs))
        elif net_type == "resnet":
            net = networks.ResNet(self.ndims, **net_kwargs)
        elif net_type == "patchdiscriminator":
            net_kwargs["norm_layer"] = functools.partial(hipnn.layers(self.ndims)[2], affine=False, track_running_stats=False)
            net = networks.NLayerDiscriminator(self.ndims, **net_kwargs)
            add_final = False
        elif hasattr(networks, net_type):
            net = getattr(networks, net_type)(**net_kwargs)
        else:
            raise RuntimeError(f"Unknown network type requested: {net_type}")
        if add_final:
            if self.ndims not in (2, 3):
                raise ValueError(self.ndims, "Only 2D or 3D currently implemented.")
            layers = [net, hipnn.layers(self.ndims)[0](net_kwargs["ngf"], output_nc, (1,) * self.ndims,
                                                       padding="valid" if "padding_type" not in net_kwargs else net_kwargs["padding_type"])]
            if final_activation is not None:
                layers.append(hipnn.hip_layer(final_activation)())  # the reference appends the class itself (:221-222)
            net = RaygunSequential(*layers)
        activation = net_kwargs["activation"] if "activation" in net_kwargs else torch.nn.ReLU
        if activation is not None:
            init_weights(net, init_type="kaiming", nonlinearity=activation.__class__.__name__.lower())
        elif net_type == "classic":
            init_weights(net, init_type="kaiming")
        else:
            init_weights(net, init_type="normal", init_gain=0.05)
        return net

    # -- to be provided by subclasses
    def setup_networks(self):
        raise NotImplementedError()

    def setup_model(self):
        raise NotImplementedError()

    def setup_optimization(self):
        raise NotImplementedError()

    def batch_specs(self):
        """{array name: (shape, lo, hi)} of one synthetic batch — provided by subclasses."""
        raise NotImplementedError()

    def setup_datapipes(self):
        """The reference builds gunpowder graphs here (io/*DataPipe.py, out of scope).  Here: `self.batches`, an iterable of
        batch dicts (train/data.py) — random crops of the config's `sources` where that is defined for the system, counter-hash
        synthetic volumes when the config says `"synthetic_data": true` or names no sources."""
        from ..train.data import SyntheticBatches
        self.datapipes = {}
        rank = getattr(self, "distributed", (0, 0, 1))[0]
        if getattr(self, "synthetic_data", False) or not getattr(self, "sources", None):
            self.batches = SyntheticBatches(self.batch_specs(), self.device, getattr(self, "random_seed", None) or 42, rank)
        else:
            self.batches = self.source_batches(rank)

    def source_batches(self, rank):
        raise NotImplementedError(f"raygun_amd.{type(self).__name__}: training from `sources` needs the reference's target "
                                  "generation (SURVEY.md 8(f4), not built); set \"synthetic_data\": true or pass batches=")

    def setup_trainer(self):
        from ..train import Trainer
        kw = getattr(self, "train_kwargs", None) or {}
        for k in ("checkpoint_basename", "save_every", "log_every"):  # newer configs keep these under train_kwargs (BaseSystem.py:286-293)
            if k in kw:
                setattr(self, k, kw[k])
        self.trainer = Trainer(self)

    @property
    def device(self):
        import torch as _t
        d = getattr(self, "distributed", None)
        return _t.device("cuda", d[1] if d is not None else _t.cuda.current_device())

    def build_system(self):
        """BaseSystem.py:312-318 (networks, model, optimisation, data pipes, trainer) + what gunpowder's Train node does when
        it starts: modules to the GPU, latest checkpoint restored.  Under data parallelism (`self.distributed`, set by
        raygun_amd.train) rank 0's initial weights are broadcast and the optimisation gets a GradSync."""
        from ... import parallel
        if not torch.cuda.is_available():
            raise RuntimeError("raygun_amd: training runs on the MI355X only (no CPU fallback)")
        self.setup_networks()
        self.setup_model()
        dev = self.device
        self.model.to(dev)
        for name in ("netD1", "netD2"):
            if hasattr(self, name):
                getattr(self, name).to(dev)
        world = getattr(self, "distributed", (0, 0, 1))[2]
        grad_sync = None
        if world > 1:
            for m in [self.model] + [getattr(self, n) for n in ("netD1", "netD2") if hasattr(self, n)]:
                parallel.broadcast_parameters(m)
            grad_sync = parallel.GradSync()
        self.setup_optimization(grad_sync) if grad_sync is not None else self.setup_optimization()
        if hasattr(self.loss, "to"):
            self.loss.to(dev)
        self.setup_datapipes()
        self.setup_trainer()
        if getattr(self, "checkpoint", None):
            self.trainer.load_checkpoint(self.checkpoint)

    def train(self, batches=None):
        """BaseSystem.py:320-328: num_epochs from train_kwargs when present."""
        if not hasattr(self, "trainer"):
            self.build_system()
        kw = getattr(self, "train_kwargs", None) or {}
        self.batch = self.trainer.train(kw.get("num_epochs", getattr(self, "num_epochs", None)), batches=batches)
        return self.batch


def _deepcopy(v):
    import copy
    return copy.deepcopy(v)
