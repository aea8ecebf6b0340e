"""Fault localisation for the HIP path (off unless RAYGUN_AMD_DEBUG_BOUNDS=1; tests/tools only — the product path is unchanged).

With the switch on
  * every output / workspace / packed operand the Python side allocates for a library call (`empty`) sits between two 4-KiB guard
    bands filled with 0xA5;
  * after EVERY library call (`_lib.check`) the stream is synchronised and the guard bands of the most recent allocations are
    verified — a kernel that writes before or past one of its buffers is named at the call that did it, instead of corrupting a
    neighbour that fails much later (or never);
  * every tensor descriptor handed to the library (`ops.desc`) is checked against its storage: the last byte a kernel may touch
    through the view — INCLUDING the zero-padded channel lanes kernels read and write beyond the logical channel count — must lie
    inside the allocation;
  * the library re-derives the byte extents of its buffer descriptors and refuses to launch on a mismatch (rg_set_option
    "RG_DEBUG_BOUNDS", read from the same environment variable)."""
import collections
import os

import torch

ON = bool(os.environ.get("RAYGUN_AMD_DEBUG_BOUNDS"))
PAD = 4096
_recent = collections.deque(maxlen=48)   # (raw uint8 buffer, payload bytes) of the latest guarded allocations (strong refs: bounded)
stats = dict(allocs=0, checks=0, descs=0)


def empty(shape, dtype, device, channels_last_3d=False):
    """torch.empty(shape, dtype, device) — between guard bands when the debug switch is on."""
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
    if not ON:
        return torch.empty(shape, dtype=dtype, device=device,
                           memory_format=torch.channels_last_3d if channels_last_3d else torch.contiguous_format)
    item = torch.empty(0, dtype=dtype).element_size()
    n = 1
    for s in shape:
        n *= s
    nbytes = (n * item + 15) // 16 * 16
    raw = torch.full((PAD + nbytes + PAD,), 0xA5, dtype=torch.uint8, device=device)
    _recent.append((raw, nbytes))
    stats["allocs"] += 1
    flat = raw[PAD:PAD + n * item].view(dtype)
    if channels_last_3d and len(shape) == 5:
        a, b, d, h, w = shape
        return flat.view(a, d, h, w, b).permute(0, 4, 1, 2, 3)
    return flat.view(shape)


def verify(what=""):
    """Synchronise and check the guard bands of the recent allocations; raises RuntimeError naming `what`."""
    if not ON:
        return
    torch.cuda.synchronize()
    stats["checks"] += 1
    for raw, nbytes in _recent:
        head, tail = raw[:PAD], raw[PAD + nbytes:]
        if bool((head != 0xA5).any()) or bool((tail != 0xA5).any()):
            where = "before" if bool((head != 0xA5).any()) else "past"
            band = head if where == "before" else tail
            first = int((band != 0xA5).nonzero()[0])
            raise RuntimeError(f"raygun_amd DEBUG_BOUNDS: write {where} a {nbytes}-byte buffer (guard byte {first}) detected after {what}")


def check_view(t, lanes=None):
    """The last element a kernel may touch through view `t` (channel dim 1 counted with `lanes` padded lanes when its stride is 1)
    must lie inside t's storage."""
    if not ON or not torch.is_tensor(t) or t.numel() == 0:
        return
    stats["descs"] += 1
    last = t.storage_offset()
    for i, (n, st) in enumerate(zip(t.shape, t.stride())):
        ext = n
        if i == 1 and lanes is not None and st == 1 and t.dim() == 5:
            ext = max(n, lanes)
        last += (ext - 1) * st
    if last < 0 or (last + 1) * t.element_size() > t.untyped_storage().nbytes():
        raise RuntimeError(f"raygun_amd DEBUG_BOUNDS: view {tuple(t.shape)} strides {tuple(t.stride())} offset {t.storage_offset()} "
                           f"(padded lanes {lanes}) reaches element {last} of a {t.untyped_storage().nbytes()}-byte storage")
