"""ORACLE TOOLING (test infrastructure, runs in the build container only).

Makes the reference's own hot-path modules importable from /root/reference so that
golden vectors can be generated from them.  The reference
never travels to the GPU box: only the small .npz fixtures under tests/golden/ do.
Nothing here is imported by the product package.

Third-party packages that are absent in this image (gunpowder, daisy, funlib, lsd,
waterz, zarr, affogato, skimage, jsmin, jax) are replaced by permissive placeholder
modules; raygun/__init__.py is never executed (it imports files missing from the
snapshot: models/NotACycleModel.py, losses/NotACycleLoss.py, systems/NotACycleGAN.py).
"""
import os
import re
import sys
import types

REF_SRC = "/root/reference/src"


def available():
    return os.path.isdir(os.path.join(REF_SRC, "raygun"))


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, k):
        return _Anything()

    def __add__(self, o):
        return self

    __radd__ = __add__


class _Permissive(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (_Anything,), {})


def _perm(name):
    m = _Permissive(name)
    m.__path__ = ["/nonexistent"]
    sys.modules[name] = m
    return m


def _ns(name, path="/nonexistent", **attrs):
    m = types.ModuleType(name)
    m.__path__ = [path]
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_done = False


def install():
    """Register placeholders; afterwards `import raygun.torch.networks` etc. work."""
    global _done
    if _done:
        return
    if not available():
        raise RuntimeError("reference not present at /root/reference (GPU box?)")
    _ns("raygun", os.path.join(REF_SRC, "raygun"))
    for n in [
        "funlib", "funlib.learn", "funlib.learn.torch", "funlib.learn.torch.models",
        "funlib.learn.torch.models.conv4d", "funlib.evaluate", "gunpowder", "gunpowder.nodes",
        "daisy", "lsd", "lsd.train", "lsd.train.gp", "waterz", "zarr", "affogato",
        "affogato.segmentation", "skimage", "skimage.segmentation", "jax",
    ]:
        _perm(n)
    sys.modules["gunpowder"].nodes = sys.modules["gunpowder.nodes"]
    _ns("jsmin", jsmin=lambda s: re.sub(r'(?m)(?<![:"])//.*$', "", s))
    _ns("raygun.torch.models.NotACycleModel", NotACycleModel=None)
    _ns("raygun.torch.losses.NotACycleLoss", NotACycleLoss=None)
    _ns("raygun.torch.systems.NotACycleGAN", NotACycleGAN=None)
    _ns("raygun.evaluation")
    _ns("raygun.evaluation.validate_affinities", run_validation=lambda *a, **k: None)
    _done = True
