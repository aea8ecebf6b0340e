"""Deterministic counter-based synthetic data (inputs *and* weights).

A splitmix64 finaliser over the flat element index, so that the GPU box, this
container and the golden-fixture generator reproduce identical volumes and
weights without sharing any RNG state ("counter-based hash ->
uniform/normal; NOT torch.manual_seed").  Pure numpy; no reference code.
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)
_G = np.uint64(0x9E3779B97F4A7C15)
_A = np.uint64(0xBF58476D1CE4E5B9)
_B = np.uint64(0x94D049BB133111EB)


def _mix(x):
    with np.errstate(over="ignore"):
        x = x + _G
        x = (x ^ (x >> np.uint64(30))) * _A
        x = (x ^ (x >> np.uint64(27))) * _B
        return x ^ (x >> np.uint64(31))


def _u01(n, seed, stream=0, chunk=1 << 24):
    """float64 uniforms in [0,1) for flat indices 0..n-1."""
    out = np.empty(n, dtype=np.float64)
    with np.errstate(over="ignore"):
        base = _mix(np.uint64(seed) * np.uint64(0x632BE59BD9B4E019) + np.uint64(stream) * _G)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        with np.errstate(over="ignore"):
            h = _mix(np.arange(s, e, dtype=np.uint64) ^ base)
        out[s:e] = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return out


def uniform(shape, seed, lo=-1.0, hi=1.0, dtype=np.float32):
    n = int(np.prod(shape))
    u = _u01(n, seed)
    return (lo + (hi - lo) * u).astype(dtype).reshape(shape)


def normal(shape, seed, std=1.0, dtype=np.float32):
    n = int(np.prod(shape))
    u1 = _u01(n, seed, stream=1)
    u2 = _u01(n, seed, stream=2)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return (std * z).astype(dtype).reshape(shape)


def fill_state_dict(state_dict, seed, bias_scale=0.05):
    """Deterministic weights for every tensor of a torch state_dict (in key order).

    Conv/linear weights ~ N(0, 2/fan_in) (the distribution kaiming_normal_ draws from,
    reference networks/utils.py:62-63); 1-D tensors (biases) ~ U(-bias_scale, bias_scale)
    so that the bias path is exercised (the reference zero-initialises them, :72-73).
    """
    import torch

    out = {}
    for i, (k, v) in enumerate(state_dict.items()):
        shp = tuple(v.shape)
        if not v.dtype.is_floating_point:
            out[k] = v.clone()
            continue
        if v.dim() >= 2:
            fan_in = int(np.prod(shp[1:]))
            arr = normal(shp, seed * 1000 + i, std=float(np.sqrt(2.0 / fan_in)))
        else:
            arr = uniform(shp, seed * 1000 + i, -bias_scale, bias_scale)
        out[k] = torch.from_numpy(arr).to(v.dtype)
    return out
