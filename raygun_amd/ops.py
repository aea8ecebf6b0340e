"""Autograd-connected operators that call the gfx950 kernels through the C-ABI.

Tensor convention between operators: a plain ``torch.Tensor`` of dtype bf16 and logical shape
(N,C,D,H,W) whose memory is NDHWC with the channel lanes padded to a multiple of 8 and the pad
lanes zero (include/raygun_hip.h).  Crops are zero-copy views.  Anything else handed to an
operator (fp32 NCDHW from the user, a dense tensor autograd produced) is converted on the GPU by
rg_copy_convert.  There is no CPU path: CPU tensors raise.
"""
import contextlib
import ctypes as C
import functools
import os
import threading

import torch

from . import _debug, _lib
from ._lib import ACT, i3, rg_conv_params, rg_pack_params, rg_tensor

_DT = {torch.bfloat16: _lib.RG_BF16, torch.float32: _lib.RG_F32, torch.uint8: _lib.RG_U8,
       torch.int32: _lib.RG_I32, torch.float64: _lib.RG_F64, torch.float16: _lib.RG_F16}

# ---- 16-bit storage type ------------------------------------------------------------------------------------------------------
# bf16 (default; libraygun_hip.so) or IEEE half (libraygun_hip_f16.so: 11 significant bits — the precision mode of the block
# runner).  The CURRENT type is thread-local: `with ops.precision("f16"):` sets it for the calling thread, and every
# autograd Function below runs its forward under the type of its 16-bit inputs (else the current one), remembers it on its ctx
# and runs its backward — on autograd's own thread — under the same type.
_KIND = {torch.bfloat16: "bf16", torch.float16: "f16"}
_HALF = {"bf16": torch.bfloat16, "f16": torch.float16, torch.bfloat16: torch.bfloat16, torch.float16: torch.float16}
_prec = threading.local()


def storage_dtype():
    return getattr(_prec, "dtype", torch.bfloat16)


@contextlib.contextmanager
def precision(kind):
    """Run the enclosed raygun_amd calls with `kind` ("bf16" | "f16" | torch dtype) as the 16-bit storage type."""
    prev = storage_dtype()
    _prec.dtype = _HALF[kind]
    try:
        yield
    finally:
        _prec.dtype = prev


def _L():
    """The library of the current storage type."""
    return _lib.lib(_KIND[storage_dtype()])


@contextlib.contextmanager
def option(name, value):
    """Set a diagnostic switch of BOTH libraries (include/raygun_hip.h rg_set_option: "RG_FM", "RG_NO_HALO", ...) for the enclosed
    calls.  Process-wide, for tests and tools; production code never touches these."""
    libs = [_lib.lib("bf16"), _lib.lib("f16")]
    prev = [l.rg_get_option(name.encode()) for l in libs]
    if min(prev) < 0:
        raise KeyError(name)
    for l in libs:
        _lib.check(l.rg_set_option(name.encode(), int(value)), "rg_set_option")
    try:
        yield
    finally:
        for l, v in zip(libs, prev):
            l.rg_set_option(name.encode(), v)


def _first_half_dtype(args):
    for a in args:
        if torch.is_tensor(a) and a.dtype in _KIND and a.is_cuda:
            return a.dtype
    return None


class Function(torch.autograd.Function):
    """torch.autograd.Function whose forward/backward run under one 16-bit storage type (see above)."""

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        fwd, bwd = cls.__dict__.get("forward"), cls.__dict__.get("backward")
        if fwd is not None:
            f = fwd.__func__

            @functools.wraps(f)
            def forward(ctx, *args):
                dt = _first_half_dtype(args) or storage_dtype()
                ctx.rg_dtype = dt
                with precision(dt):
                    return f(ctx, *args)
            cls.forward = staticmethod(forward)
        if bwd is not None:
            b = bwd.__func__

            @functools.wraps(b)
            def backward(ctx, *grads):
                with precision(ctx.rg_dtype):
                    return b(ctx, *grads)
            cls.backward = staticmethod(backward)



_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The calling thread's current HIP stream as a C pointer.  torch.cuda.current_stream() builds a Stream object through three device-index lookups
    (~10 us; 2 000 calls per CycleGAN step on each of the two enqueueing threads = 8 of the 60 ms the host needs per step); the raw accessor the
    compiled backends use returns the same handle in under a microsecond."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"raygun_amd.{what}: expected a CUDA/HIP tensor. The product path runs only on the "
                           "MI355X HIP kernels; CPU execution is the oracle's job (oracle/), not a fallback.")


def desc(t):
    """rg_tensor descriptor of a torch tensor (rank <= 6)."""
    if _debug.ON:
        _debug.check_view(t, lanes_of(t.shape[1]) if is_cl(t) else None)   # (padded lanes exist only in the library's own layout)
    d = rg_tensor()
    d.ptr = t.data_ptr()
    d.dtype = _DT[t.dtype]
    shape, stride = t.shape, t.stride()
    n = d.rank = len(shape)
    d.sizes[:n] = shape
    d.strides[:n] = stride
    if n and shape[0] == 1:
        d.strides[0] = 0  # a length-1 leading dimension is never stepped; views renormalise its stride to odd values
    return d


def lanes_of(c):
    return (c + 7) // 8 * 8


def voxel_stride_of(c):
    """Elements between consecutive voxels: wide tensors are padded to whole 128-byte lines so that every 64-channel
    slice the conv kernels DMA (8 lanes x 16 B) is exactly one cache line instead of straddling two."""
    return lanes_of(c) if c < 64 else (c + 63) // 64 * 64


def cl_empty(n, c, d, h, w, device):
    """Uninitialised channels-last bf16 activation (every kernel writes all lanes incl. zero pad lanes)."""
    buf = _debug.empty((n, d, h, w, voxel_stride_of(c)), storage_dtype(), device)
    return buf.permute(0, 4, 1, 2, 3)[:, :c]


def cl_empty_dw(n, c, d, h, w, device):
    """(N,C,D,H,W) channels-last view whose memory is laid out with D and W exchanged: what `t.transpose(2, 4)` of an ordinary
    channels-last tensor of shape (N,C,W,H,D) looks like.  Outputs of the D<->W-swapped convolutions below are allocated
    this way so that, swapped back, they are ordinary channels-last tensors again."""
    return cl_empty(n, c, w, h, d, device).transpose(2, 4)


def is_dense(t):
    """Non-overlapping and dense: the storage is a permutation of the elements (contiguous, channels_last_3d, ...)."""
    return t.is_contiguous() or (t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d)) or \
        (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def is_cl_weight(w):
    """5-D fp32 weight in torch.channels_last_3d memory ([dim0][KD][KH][KW][dim1]) — hipnn.Conv3d keeps its masters that way."""
    return w.dim() == 5 and w.dtype == torch.float32 and not w.is_contiguous() and w.is_contiguous(memory_format=torch.channels_last_3d)


def is_cl(t):
    if t.dtype not in _KIND or t.dim() != 5 or not t.is_cuda:
        return False
    n, c, d, h, w = t.shape
    st = t.stride()
    if c > 1 and st[1] != 1:
        return False
    if st[4] < lanes_of(c) or any(s % 8 for s in (st[2], st[3], st[4])) or (n > 1 and st[0] % 8):
        return False  # (the stride of a length-1 batch dimension is arbitrary: views renormalise it)
    if t.data_ptr() % 16:
        return False
    # kernels touch all lanes_of(c) lanes of every voxel: the last voxel's pad lanes must exist (torch's own as_strided backward
    # hands back a view with this geometry over a storage that ends at the last LOGICAL element)
    last = t.storage_offset() + (n - 1) * st[0] + (d - 1) * st[2] + (h - 1) * st[3] + (w - 1) * st[4] + lanes_of(c)
    return t.numel() == 0 or last * t.element_size() <= t.untyped_storage().nbytes()


def to_cl(t, scale=1.0, shift=0.0):
    """Any (N,C,D,H,W) tensor -> channels-last bf16 (no autograd; use as_cl inside Functions)."""
    _require_cuda(t, "to_cl")
    if t.dim() != 5:
        raise RuntimeError(f"raygun_amd: expected a 5-D (N,C,D,H,W) tensor, got shape {tuple(t.shape)}")
    if scale == 1.0 and shift == 0.0 and is_cl(t) and t.dtype == storage_dtype():
        return t
    if t.dtype not in (storage_dtype(), torch.float32, torch.uint8):
        t = t.float()  # (includes a 16-bit tensor of the OTHER storage type: through fp32)
    n, c, d, h, w = t.shape
    out = cl_empty(n, c, d, h, w, t.device)
    if t.numel():
        _lib.check(_L().rg_copy_convert(C.byref(desc(t)), C.byref(desc(out)), lanes_of(c), scale, shift, _stream()),
                   "rg_copy_convert")
    return out


as_cl = to_cl


def from_cl(t, dtype=torch.float32):
    """channels-last bf16 -> dense NCDHW `dtype` (no autograd)."""
    out = _debug.empty(t.shape, dtype, t.device)
    if t.numel():
        _lib.check(_lib.lib(_KIND[t.dtype]).rg_copy_convert(C.byref(desc(t)), C.byref(desc(out)), t.shape[1], 1.0, 0.0, _stream()),
                   "rg_copy_convert")
    return out


class _ToCL(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.src_dtype = x.dtype
        return to_cl(x)

    @staticmethod
    def backward(ctx, g):
        return from_cl(to_cl(g), ctx.src_dtype if ctx.src_dtype.is_floating_point else torch.float32)


class _FromCL(Function):
    @staticmethod
    def forward(ctx, x):
        return from_cl(to_cl(x))

    @staticmethod
    def backward(ctx, g):
        return to_cl(g)


def store_block(t, dst, scale=1.0, lo=0.0, hi=255.0, flag=None):
    """The block runner's write-back in one kernel (rg_copy_convert_store): channels-last 16-bit `t` (N,C,D,H,W — any crop view) into `dst`, a
    same-shaped view of the destination array (float32, or uint8: clamp(t * scale, lo, hi) truncated); `flag` (float32[1]) is set when `t`
    holds an inf / NaN."""
    if t.numel() == 0:
        return
    if tuple(t.shape) != tuple(dst.shape) or dst.dtype not in (torch.float32, torch.uint8) or t.dtype not in _KIND:
        raise RuntimeError(f"raygun_amd.store_block: {tuple(t.shape)} {t.dtype} -> {tuple(dst.shape)} {dst.dtype}")
    _lib.check(_lib.lib(_KIND[t.dtype]).rg_copy_convert_store(C.byref(desc(t)), C.byref(desc(dst)), t.shape[1], float(scale), 0.0, float(lo), float(hi),
                                                              C.c_void_p(flag.data_ptr()) if flag is not None else None, _stream()), "rg_copy_convert_store")


_lift = threading.local()  # .on: a 4-D (N,C,H,W) tensor entered the current outermost scope and was lifted to D = 1


def lift_reset():
    _lift.on = False


def lifted():
    return getattr(_lift, "on", False)


def enter(x):
    """Network entry: user tensor -> channels-last bf16, differentiable.  2-D (N,C,H,W) tensors travel as (N,C,1,H,W):
    every 2-D layer is its 3-D kernel with one plane (reference 2-D nets: networks/ResNet.py:6-223,
    NLayerDiscriminator.py:5-101)."""
    _require_cuda(x, "enter")
    if x.dim() == 4:
        _lift.on = True
        x = x.unsqueeze(2)
    if is_cl(x):
        return x
    return _ToCL.apply(x)


def leave(x):
    """Network exit: channels-last bf16 -> fp32 NCDHW contiguous (NCHW if the input was 2-D), differentiable."""
    if isinstance(x, (list, tuple)):
        return type(x)(leave(t) for t in x)
    if x is None or not is_cl(x):
        return x
    y = _FromCL.apply(x)
    if lifted() and y.dim() == 5 and y.shape[2] == 1:
        y = y.squeeze(2)
    return y


# ---------------------------------------------------------------------------------------------- weights
def _triple(v):
    return (int(v),) * 3 if isinstance(v, int) else tuple(int(a) for a in v)


def invalidate_packed(obj):
    """Drop the packed bf16 operands cached on a parameter (or on every parameter of a module).  The caches key on
    (tensor._version, data_ptr); in-place writes THROUGH `.data` (`w.data.normal_()`, `w.data.copy_()`, EMA / clipping code)
    do not bump `_version`, so whoever writes that way after a first forward must call this — raygun_amd's own writers
    (networks.utils.init_weights, BaseCompetentLoss.clamp_weights) do.  Optimisers, load_state_dict and every ordinary
    in-place op under no_grad bump the version and need nothing."""
    if isinstance(obj, torch.nn.Module):
        for p in obj.parameters():
            invalidate_packed(p)
        return
    if hasattr(obj, "_rg_pack_cache"):
        try:
            del obj._rg_pack_cache
        except AttributeError:
            pass


def pack_weights(w, swap, k0=(0, 0, 0), kstep=(1, 1, 1), ntap=None):
    """fp32 torch-layout weight -> packed bf16 [rows16][taps][red32]; cached on the tensor per version."""
    key = (int(swap), tuple(k0), tuple(kstep), None if ntap is None else tuple(ntap))
    if storage_dtype() != torch.bfloat16:
        key = key + (_KIND[storage_dtype()],)
    cache = getattr(w, "_rg_pack_cache", None)
    ver = (w._version, w.data_ptr())
    if cache is None or cache.get("ver") != ver:
        cache = {"ver": ver}
        try:
            w._rg_pack_cache = cache
        except Exception:
            pass
    if key in cache:
        return cache[key]
    wd = w.detach()
    src_cl = is_cl_weight(wd)
    if not src_cl and (wd.dtype != torch.float32 or not wd.is_contiguous()):
        wd = wd.float().contiguous()
    p = rg_pack_params()
    p.src_channels_last = int(src_cl)
    p.dim0, p.dim1 = wd.shape[0], wd.shape[1]
    K = wd.shape[2:]
    nt = ntap if ntap is not None else K
    for i in range(3):
        p.K[i], p.k0[i], p.kstep[i], p.ntap[i] = K[i], k0[i], kstep[i], nt[i]
    p.swap = int(swap)
    rows, red = (wd.shape[1], wd.shape[0]) if swap else (wd.shape[0], wd.shape[1])
    p.red_align = _L().rg_conv3d_kalign(int(red), int(rows))
    nbytes = _L().rg_pack_weights_bytes(C.byref(p))
    if nbytes < 0:
        raise RuntimeError("raygun_amd.pack_weights: bad packing parameters")
    out = _debug.empty(nbytes // 2, storage_dtype(), w.device)
    _lib.check(_L().rg_pack_weights(C.c_void_p(wd.data_ptr()), C.c_void_p(out.data_ptr()), C.byref(p), _stream()),
               "rg_pack_weights")
    cache[key] = out
    if src_cl and not swap and ntap is None and tuple(k0) == (0, 0, 0) and tuple(kstep) == (1, 1, 1) and red % 4 == 0:
        cache["fwd_pack" if len(key) == 4 else "fwd_pack_f16"] = (out, p, key)  # FusedAdam refreshes this operand in its own pass (adam_step_)
    return out


def _conv_params(R, T, rstride, tbase, tstep, ostride, ooffset, cout, act="none", slope=0.0):
    p = rg_conv_params()
    for i in range(3):
        p.R[i], p.T[i], p.rstride[i], p.tbase[i], p.tstep[i] = R[i], T[i], rstride[i], tbase[i], tstep[i]
        p.ostride[i], p.ooffset[i] = ostride[i], ooffset[i]
    p.cout, p.act, p.slope = cout, ACT[act], float(slope)
    return p


_prof = None


_prof_select = None   # None: every conv launch is timed; else only the problems whose signature is in this set
_prof_sigs = {}       # kernel name -> problem signatures, from the last session that timed everything


def profile_signatures():
    """kernel name -> set of problem signatures seen by the last profile session that timed every launch."""
    return _prof_sigs


def profile_start(select=None):
    """Start recording a HIP-event pair (on torch's current stream = the launch stream) around every conv kernel, or — `select`
    = an iterable of problem signatures from profile_signatures() — only around those problems: an event pair costs ~50 us of
    host time and keeps neighbouring kernels from overlapping (the U-Net step measures 4 % slower fully instrumented), so
    bench.py instruments everything in the warm-up steps and only the dominant kernel inside the timed region.  Weight-gradient
    calls are bracketed inside the library (main kernel only) in both modes."""
    global _prof, _prof_select
    _prof = []
    _prof_select = None if select is None else set(select)
    L = _lib.lib("bf16")
    L.rg_profile_enable(1)  # weight-gradient calls: event pair around the main kernel only, inside the library
    L.rg_profile_drain(None, 0)


def profile_stop():
    """-> {kernel name: dict(launches, ms, flops)} ; algorithmic flops = 2*rows*taps*Cin*Cout per launch."""
    global _prof, _event_next, _prof_sigs
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    L = _lib.lib("bf16")
    L.rg_profile_enable(0)
    marked = [r for r in rec or [] if r[4]]
    buf = (C.c_float * max(1, len(marked)))()
    got = L.rg_profile_drain(buf, len(marked))
    inner = dict(zip(map(id, marked), buf[:got])) if got == len(marked) else {}
    out, sigs = {}, {}
    for r in rec or []:
        name, flops, e0, e1, _, sig = r
        ms = inner.get(id(r), None) or (e0.elapsed_time(e1) if e0 is not None else None)
        if ms is None:
            continue
        d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0))
        d["launches"] += 1
        d["ms"] += ms
        d["flops"] += flops
        if sig is not None:
            sigs.setdefault(name, set()).add(sig)
    if _prof_select is None:
        _prof_sigs = sigs
    _event_next = 0
    return out


profile_shapes = False  # True: key the records by kernel name AND problem shape (diagnostics)


_event_pool, _event_next = [], 0  # timing events are reused across profile sessions: creating two per launch cost more host time than the launch


def _drop_events():   # at interpreter exit, while torch's HIP context still exists (module globals are torn down in arbitrary order)
    global _event_next
    _event_pool.clear()
    _event_next = 0


import atexit  # noqa: E402
atexit.register(_drop_events)


def _event():
    global _event_next
    if _event_next == len(_event_pool):
        _event_pool.append(torch.cuda.Event(enable_timing=True))
    _event_next += 1
    return _event_pool[_event_next - 1]


def _timed(flops, fn, tag=None, marked=False, name=None):
    if _prof is None:
        return fn()
    sig = tag() if tag is not None else None
    timed = _prof_select is None or sig in _prof_select
    if not timed and not marked:
        return fn()
    e0 = e1 = None
    if timed and not (marked and _prof_select is not None):   # selected weight gradients: the library's own event pair is enough
        e0, e1 = _event(), _event()
        e0.record()
    r = fn()
    if e1 is not None:
        e1.record()
    name = name() if name is not None else _L().rg_last_kernel_name().decode()
    if profile_shapes and sig is not None:
        name += " " + sig
    _prof.append((name, flops, e0, e1, marked, sig))
    return r


def _conv_flops(p, n, cin, cout):
    return 2.0 * n * p.R[0] * p.R[1] * p.R[2] * p.T[0] * p.T[1] * p.T[2] * cin * cout


def _igemm(x, wp, bias, y, p):
    bptr = C.c_void_p(bias.data_ptr()) if bias is not None else None
    dx, dy = desc(x), desc(y)
    nbytes = _L().rg_conv3d_igemm_workspace_bytes(C.byref(dx), C.byref(dy), C.byref(p))   # split-K partial sums of small-grid launches
    ws = _debug.empty(nbytes, torch.uint8, x.device) if nbytes > 0 else None
    _timed(_conv_flops(p, x.shape[0], x.shape[1], y.shape[1]),
           lambda: _lib.check(_L().rg_conv3d_igemm2(C.byref(dx), C.c_void_p(wp.data_ptr()), bptr, C.byref(dy), C.byref(p),
                                                    C.c_void_p(ws.data_ptr()) if ws is not None else None, max(nbytes, 0), _stream()),
                              "rg_conv3d_igemm"),
           lambda: f"x{tuple(x.shape)} -> y{tuple(y.shape)} R{tuple(p.R)} T{tuple(p.T)} rs{tuple(p.rstride)} ts{tuple(p.tstep)}")


WINO_MIN_CHANNELS = 1024  # Winograd F(2,3)^3 only where K = Cin is long (config U's 2592 -> 2592 level); see csrc/winograd.hip


def _wino_ok(weight, stride, padding, dw_swapped, backward=False):
    """Winograd path for this convolution's forward (backward=False) or backward-data pass?  Each pass is judged by ITS reduction
    length — input channels forward, output channels backward — which must be >= WINO_MIN_CHANNELS (the transform-domain tensors are
    8x the activations: short reductions do not pay), the other side >= 128 (the GEMM's column tiles).  Config U: both passes of
    2592 -> 2592, and the backward-data pass of 432 -> 2592.  Both storage types: the transform-domain rounding doubles THAT layer's
    error, but at network level config U's f16 outputs stay where they were (7.6e-4 / 8.5e-4 / 8.4e-4 relative L2 against 8.6e-4 /
    7.5e-4 / 8.8e-4 direct, tools/f16_wino_probe.py; asserted <= 1e-3 in tests/test_networks_gpu.py)."""
    red, cols = (weight.shape[0], weight.shape[1]) if backward else (weight.shape[1], weight.shape[0])
    return (os.environ.get("RAYGUN_AMD_WINOGRAD", "1") != "0" and not dw_swapped
            and tuple(weight.shape[2:]) == (3, 3, 3) and tuple(stride) == (1, 1, 1) and tuple(padding) == (0, 0, 0)
            and red >= WINO_MIN_CHANNELS and cols >= min(128, WINO_MIN_CHANNELS) and weight.dtype == torch.float32
            and (is_cl_weight(weight) or weight.is_contiguous())
            # the batched GEMM tiles its columns by 128 or 144: channel counts without such a tiling (145..160, 305..320, 785..800, ...)
            # stay on the direct kernels, which fall back through rg_igemm4_ok to the 64/80-column tiles
            and _L().rg_gemm_batched_tile_n(int(cols)) >= 8)


def _wino_fits(n, out_spatial, k, nout):
    """The batched GEMM addresses one transform position's operands with 32-bit byte offsets (< 1 GiB each)."""
    nt = n
    for o in out_spatial:
        nt *= (int(o) + 1) // 2
    return nt * max(int(_L().rg_wino_weights_red(k)), int(_L().rg_wino_weights_red(nout))) * 2 < (1 << 30) and nt < (1 << 31)


def _wino_weights(weight):
    """(U, Ut): G g G^T of the fp32 master weight as the forward and the backward-data GEMM operand; cached like the packed
    operands (per tensor version, dropped by FusedAdam / invalidate_packed)."""
    cache = getattr(weight, "_rg_pack_cache", None)
    ver = (weight._version, weight.data_ptr())
    if cache is None or cache.get("ver") != ver:
        cache = {"ver": ver}
        try:
            weight._rg_pack_cache = cache
        except Exception:
            pass
    wkey = "wino" if storage_dtype() == torch.bfloat16 else "wino_f16"
    if wkey in cache:
        return cache[wkey]
    L = _L()
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    wd = weight.detach()
    U = _debug.empty(64 * L.rg_wino_weights_rows(cout) * L.rg_wino_weights_red(cin), storage_dtype(), weight.device)
    Ut = _debug.empty(64 * L.rg_wino_weights_rows(cin) * L.rg_wino_weights_red(cout), storage_dtype(), weight.device)
    _lib.check(L.rg_wino_weights(C.c_void_p(wd.data_ptr()), int(is_cl_weight(wd)), cout, cin, C.c_void_p(U.data_ptr()),
                                 C.c_void_p(Ut.data_ptr()), _stream()), "rg_wino_weights")
    cache[wkey] = (U, Ut)
    return U, Ut


def _wino_conv(x, Wt, bias, act, slope, y, origin):
    """y = conv(x, w) through the transform domain.  x (N, k, ...) and y (N, n, ...) channels-last; Wt the matching operand of
    _wino_weights; origin 0: valid convolution, -2: full correlation with the flipped kernel (backward-data)."""
    L = _L()
    n_, k = int(x.shape[0]), int(x.shape[1])
    nout = int(y.shape[1])
    tiles = [(int(o) + 1) // 2 for o in y.shape[2:]]
    nt = n_ * tiles[0] * tiles[1] * tiles[2]
    cs, ms = int(L.rg_wino_weights_red(k)), int(L.rg_wino_weights_red(nout))
    V = _debug.empty(64 * nt * cs, storage_dtype(), x.device)
    M = _debug.empty(64 * nt * ms, storage_dtype(), x.device)
    dx, dy = desc(x), desc(y)
    bptr = C.c_void_p(bias.data_ptr()) if bias is not None else None

    def run():
        _lib.check(L.rg_wino_input(C.byref(dx), C.c_void_p(V.data_ptr()), cs, _lib.i3(tiles), _lib.i3([origin] * 3), _stream()), "rg_wino_input")
        _lib.check(L.rg_gemm_batched(C.c_void_p(V.data_ptr()), C.c_void_p(Wt.data_ptr()), C.c_void_p(M.data_ptr()), 64, nt, k, nout, cs,
                                     nt * cs, int(L.rg_wino_weights_rows(nout)) * cs, ms, nt * ms, _stream()), "rg_gemm_batched")
        gname = L.rg_last_kernel_name().decode()
        _lib.check(L.rg_wino_output(C.c_void_p(M.data_ptr()), ms, _lib.i3(tiles), C.byref(dy), bptr, nout, ACT[act], float(slope), _stream()),
                   "rg_wino_output")
        return gname

    names = []
    # recorded with the multiplications it executes (64 per 2x2x2 tile and channel pair; the direct form has 216)
    _timed(2.0 * 64 * nt * k * nout, lambda: names.append(run()),
           lambda: f"x{tuple(x.shape)} -> y{tuple(y.shape)} origin {origin}", name=lambda: "winograd[" + names[-1] + "]")


def _wgrad(g, x, p, shape, channels_last=False):
    """fp32 weight gradient of logical `shape` (a, b, KD, KH, KW); channels_last: in torch.channels_last_3d memory, which is
    the [a][tap][b] order the kernels accumulate in (no transpose pass; with one row split no reduction pass either)."""
    dw = _debug.empty(shape, torch.float32, g.device, channels_last_3d=channels_last)
    ns = C.c_int32(0)
    dg, dx = desc(g), desc(x)
    nbytes = _L().rg_conv3d_wgrad_workspace_bytes(C.byref(dg), C.byref(dx), C.byref(p), C.byref(ns))
    if nbytes < 0:
        _lib.check(-1, "rg_conv3d_wgrad_workspace_bytes")
    ws = _debug.empty(max(nbytes, 16), torch.uint8, g.device)
    _timed(_conv_flops(p, g.shape[0], x.shape[1], g.shape[1]),
           lambda: _lib.check(_L().rg_conv3d_wgrad2(C.byref(dg), C.byref(dx), C.c_void_p(dw.data_ptr()), C.byref(p),
                                                          2 if channels_last else 0, C.c_void_p(ws.data_ptr()), nbytes, _stream()),
                              "rg_conv3d_wgrad"),
           lambda: f"g{tuple(g.shape)} x{tuple(x.shape)} T{tuple(p.T)} rs{tuple(p.rstride)} ts{tuple(p.tstep)} ns{ns.value}", marked=True)
    return dw


def _wgrad_wpairs(g, x, p_args, shape, transposed):
    """Weight gradient of a convolution with stride 2 / 3 taps / padding 1 along W on the stride-1 (kw-reuse, producer-wave) kernels:
    the strided operand, seen through a zero-copy view as W/2 voxel PAIRS of 2C channels [even | odd], makes input index
    2w + kw - 1 the pair w - 1 + kw' with  kw' = 0: [-, tap 0],  kw' = 1: [tap 1, tap 2]  — a stride-1 problem along W with two
    taps and 2C channels (one of the four slots is zero: 4/3 of the FLOPs, on kernels that run 2-2.5x faster than the generic
    strided one: 333 -> ~150 us for the generators' 64 -> 128 down convolution).  Returns None when the layout does not allow the
    view (odd W, lane padding between voxels).  g: the stride-1 operand, x: the strided one (Conv3d: input; ConvTranspose3d: the
    output gradient); shape: (A, B, KD, KH, 3) of the gradient, A = g's channels, B = x's."""
    if os.environ.get("RAYGUN_AMD_WPAIR", "1") == "0":
        return None
    R, K, stride, padding = p_args
    n, c, D, H, W = x.shape
    if stride[2] != 2 or K[2] != 3 or padding[2] != 1 or W % 2 or c % 8 or x.stride(1) != 1 or x.stride(4) != c or not is_cl(x) or R[2] != W // 2:
        return None
    st = x.stride()
    xv = torch.as_strided(x, (n, 2 * c, D, H, W // 2), (st[0], 1, st[2], st[3], 2 * c), x.storage_offset())
    p = _conv_params(R=R, T=(K[0], K[1], 2), rstride=(stride[0], stride[1], 1), tbase=(-padding[0], -padding[1], -1), tstep=(1, 1, 1),
                     ostride=(1, 1, 1), ooffset=(0, 0, 0), cout=shape[0])
    d2 = _wgrad(g, xv, p, (shape[0], 2 * c, K[0], K[1], 2))
    dw = torch.empty(shape, dtype=torch.float32, device=g.device)
    dw[..., 0] = d2[:, c:, :, :, 0]
    dw[..., 1] = d2[:, :c, :, :, 1]
    dw[..., 2] = d2[:, c:, :, :, 1]
    return dw


def _act_bwd_bias(gy, y, act, slope, want_bias):
    """g = gy*act'(y) (new tensor unless act == none), db = per-channel sum (fp32) or None."""
    if act == "none" and not want_bias:
        return gy, None
    n, c, d, h, w = gy.shape
    g = cl_empty(n, c, d, h, w, gy.device) if act != "none" else None
    db = _debug.empty(c, torch.float32, gy.device) if want_bias else None
    dgy = desc(gy)
    nbytes = _L().rg_act_bwd_bias_workspace_bytes(C.byref(dgy)) if want_bias else 0
    ws = _debug.empty(max(nbytes, 16), torch.uint8, gy.device)
    _lib.check(_L().rg_act_bwd_bias(C.byref(dgy), C.byref(desc(y)) if act != "none" else None,
                                          C.byref(desc(g)) if g is not None else None,
                                          C.c_void_p(db.data_ptr()) if want_bias else None, ACT[act], float(slope),
                                          C.c_void_p(ws.data_ptr()), nbytes, _stream()), "rg_act_bwd_bias")
    return (g if g is not None else gy), db


def _parity_classes(size, stride, pad, K):
    """Per-dim decomposition of a stride-s scatter (transposed conv / strided dgrad) into stride-1 gathers.
    For output coordinate o = s*j + q: contributing taps k = k0 + s*t, source index j + tbase - t."""
    out = []
    for q in range(stride):
        rows = (size - q + stride - 1) // stride
        k0 = (q + pad) % stride
        nt = (K - k0 + stride - 1) // stride if K > k0 else 0
        out.append(dict(q=q, rows=rows, k0=k0, ntap=nt, tbase=(q + pad - k0) // stride))
    return out


_SUBPIX_T_IDX = None


def _subpixel_weights_t(w):
    """(Csrc, Cdst, 3,3,3) weight of a stride-2 / pad-1 transposed convolution -> the (8*Cdst, Csrc, 2,2,2) weight of the
    ordinary convolution that computes all 8 output parities at once on the input grid: per dimension o = 2j + q reads
    q = 0: tap 1 at x[j];  q = 1: tap 2 at x[j], tap 0 at x[j+1]  (_parity_classes); output channel = ((qd*2+qh)*2+qw)*Cdst + c.
    Cached on the weight per version, like the packed operands."""
    global _SUBPIX_T_IDX
    cache = getattr(w, "_rg_pack_cache", None)
    ver = (w._version, w.data_ptr())
    if cache is not None and cache.get("ver") == ver and "subpix_t" in cache:
        return cache["subpix_t"]
    if _SUBPIX_T_IDX is None or _SUBPIX_T_IDX.device != w.device:
        tap = {(0, 0): 1, (0, 1): None, (1, 0): 2, (1, 1): 0}
        idx = []
        for q in range(8):
            for s_ in range(8):
                ks = [tap[((q >> (2 - d)) & 1, (s_ >> (2 - d)) & 1)] for d in range(3)]
                idx.append(27 if None in ks else (ks[0] * 3 + ks[1]) * 3 + ks[2])
        _SUBPIX_T_IDX = torch.tensor(idx, dtype=torch.long, device=w.device)
    csrc, cdst = w.shape[:2]
    wt = w.detach().float().permute(1, 0, 2, 3, 4).reshape(cdst, csrc, 27)
    wt = torch.cat([wt, wt.new_zeros(cdst, csrc, 1)], 2)
    weff = wt[:, :, _SUBPIX_T_IDX].view(cdst, csrc, 8, 8).permute(2, 0, 1, 3).reshape(8 * cdst, csrc, 2, 2, 2).contiguous()
    if cache is None or cache.get("ver") != ver:
        cache = {"ver": ver}
        try:
            w._rg_pack_cache = cache
        except Exception:
            pass
    cache["subpix_t"] = weff
    return weff


def _scatter_conv(src, w, bias, out, stride, pad, K, swap, act, slope):
    """out[o] = act(bias + sum_{i,k: o = i*s - p + k} src[i] . W[..k])  as one igemm launch per parity class."""
    size = out.shape[2:]
    subpix = (swap and tuple(stride) == (2, 2, 2) and tuple(K) == (3, 3, 3) and tuple(pad) == (1, 1, 1) and out.shape[1] % 8 == 0
              and all(o == 2 * i for o, i in zip(size, src.shape[2:])) and os.environ.get("RAYGUN_AMD_SUBPIXEL_T", "1") != "0")
    if subpix and src.shape[1] >= 128 and out.shape[1] % 16 == 0 and out.stride(4) == out.shape[1] and os.environ.get("RAYGUN_AMD_D2S_EPILOGUE", "1") != "0":
        # the 8 parity classes as ONE convolution with 8*Cout columns and 2^3 taps on the input grid, its epilogue writing every column block
        # to its own output parity (rg_conv3d_igemm_d2s, igemm5): no 8*Cout-channel intermediate, no shuffle pass — 128 -> 64 at 64^3 used to be
        # eight parity launches re-reading the input (0.45 ms per call), 256 -> 128 at 32^3 a sub-pixel convolution + depth-to-space
        n, _, D, H, W = src.shape
        p = _conv_params(R=(D, H, W), T=(2, 2, 2), rstride=(1, 1, 1), tbase=(0, 0, 0), tstep=(1, 1, 1), ostride=(2, 2, 2),
                         ooffset=(0, 0, 0), cout=8 * out.shape[1], act=act, slope=slope)
        dsrc, dout = desc(src), desc(out)
        if _L().rg_conv3d_igemm_d2s(C.byref(dsrc), None, None, C.byref(dout), C.byref(p), 1, _stream()) == 0:
            wp = pack_weights(_subpixel_weights_t(w), swap=0)
            b = bias.detach().float().contiguous() if bias is not None else None
            _timed(_conv_flops(p, src.shape[0], src.shape[1], 8 * out.shape[1]),
                   lambda: _lib.check(_L().rg_conv3d_igemm_d2s(C.byref(dsrc), C.c_void_p(wp.data_ptr()), C.c_void_p(b.data_ptr()) if b is not None else None,
                                                               C.byref(dout), C.byref(p), 0, _stream()), "rg_conv3d_igemm_d2s"),
                   lambda: f"x{tuple(src.shape)} -> y{tuple(out.shape)} R{tuple(p.R)} T{tuple(p.T)} sub-pixel + d2s epilogue")
            return
    if subpix and src.shape[1] >= 192:
        # wide inputs: the 8 parity classes as ONE convolution with 8*Cout channels and 2^3 taps on the input grid (64/27 of
        # the FLOPs, in one launch that fills the chip instead of eight 16-tile launches with 1..8-tap K loops: 256 -> 128 at
        # 32^3 takes 0.18 ms instead of 0.64), then a depth-to-space shuffle
        weff = _subpixel_weights_t(w)
        n, _, D, H, W = src.shape
        z = cl_empty(n, 8 * out.shape[1], D, H, W, src.device)
        p = _conv_params(R=(D, H, W), T=(2, 2, 2), rstride=(1, 1, 1), tbase=(0, 0, 0), tstep=(1, 1, 1), ostride=(1, 1, 1),
                         ooffset=(0, 0, 0), cout=8 * out.shape[1], act=act, slope=slope)
        _igemm(src, pack_weights(weff, swap=0), bias.repeat(8) if bias is not None else None, z, p)
        _lib.check(_L().rg_d2s_add_act(None, C.byref(desc(z)), C.byref(desc(out)), i3((0, 0, 0)), ACT["none"], 0.0, _stream()),
                   "rg_d2s_add_act")
        return
    classes = [_parity_classes(size[i], stride[i], pad[i], K[i]) for i in range(3)]
    if any(c["ntap"] == 0 for cl in classes for c in cl):
        out.zero_()  # some parity class has no tap (kernel smaller than stride): bias/act of zero handled below
        if bias is not None or act != "none":
            raise RuntimeError("raygun_amd: transposed conv with kernel < stride and bias/activation is unsupported")
    cout = out.shape[1]
    for cd in classes[0]:
        for ch in classes[1]:
            for cw in classes[2]:
                cs = (cd, ch, cw)
                if any(c["rows"] <= 0 or c["ntap"] == 0 for c in cs):
                    continue
                wp = pack_weights(w, swap, k0=[c["k0"] for c in cs], kstep=stride, ntap=[c["ntap"] for c in cs])
                p = _conv_params(R=[c["rows"] for c in cs], T=[c["ntap"] for c in cs], rstride=(1, 1, 1),
                                 tbase=[c["tbase"] for c in cs], tstep=(-1, -1, -1), ostride=stride,
                                 ooffset=[c["q"] for c in cs], cout=cout, act=act, slope=slope)
                _igemm(src, wp, bias, out, p)


# ---------------------------------------------------------------------------------------------- Conv3d
class _Conv3d(Function):
    """torch.nn.Conv3d (+ fused activation), zero padding.  Reference call sites: networks/UNet.py:58-69,
    networks/ResNet.py:283,291-298,337-338,407,416, networks/NLayerDiscriminator.py:136-176."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, act, slope, dw_swapped=False):
        """dw_swapped: x (and gy) are D<->W transposed views of ordinary channels-last tensors (conv_kd_as_kw); outputs are
        allocated so that they are ordinary again once swapped back."""
        _require_cuda(x, "conv3d")
        x = as_cl(x)
        alloc = cl_empty_dw if dw_swapped else cl_empty
        ctx.alloc = alloc
        n, cin, D, H, W = x.shape
        cout, cin_w = weight.shape[0], weight.shape[1]
        K = tuple(weight.shape[2:])
        if cin_w != cin:
            raise RuntimeError(f"Given groups=1, weight of size {list(weight.shape)}, expected input{list(x.shape)} "
                               f"to have {cin_w} channels, but got {cin} channels instead")
        O = [(i + 2 * p - k) // s + 1 for i, p, k, s in zip((D, H, W), padding, K, stride)]
        if min(O) <= 0:
            raise RuntimeError(f"Calculated padded input size per channel: {tuple(i + 2 * p for i, p in zip((D, H, W), padding))}. "
                               f"Kernel size: {K}. Kernel size can't be greater than actual input size")
        y = alloc(n, cout, *O, x.device)
        b = bias.detach().float().contiguous() if bias is not None else None
        ctx.wino = _wino_ok(weight, stride, padding, dw_swapped, backward=True) and _wino_fits(n, (D, H, W), cout, cin)
        if _wino_ok(weight, stride, padding, dw_swapped) and _wino_fits(n, O, cin, cout):
            _wino_conv(x, _wino_weights(weight)[0], b, act, slope, y, 0)
        else:
            wp = pack_weights(weight, swap=0)
            p = _conv_params(R=O, T=K, rstride=stride, tbase=[-q for q in padding], tstep=(1, 1, 1), ostride=(1, 1, 1),
                             ooffset=(0, 0, 0), cout=cout, act=act, slope=slope)
            _igemm(x, wp, b, y, p)
        ctx.save_for_backward(x, weight, y if act != "none" else None)
        ctx.cfg = (stride, padding, act, slope, bias is not None, K, O)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        stride, padding, act, slope, has_bias, K, O = ctx.cfg
        gy = as_cl(gy)
        g, db = _act_bwd_bias(gy, y, act, slope, has_bias and ctx.needs_input_grad[2])
        gx = dw = None
        n, cin, D, H, W = x.shape
        if ctx.needs_input_grad[0]:
            gx = ctx.alloc(n, cin, D, H, W, x.device)
            if ctx.wino:
                _wino_conv(g, _wino_weights(weight)[1], None, "none", 0.0, gx, -2)
            elif stride == (1, 1, 1):
                wp = pack_weights(weight, swap=1)
                p = _conv_params(R=(D, H, W), T=K, rstride=(1, 1, 1), tbase=padding, tstep=(-1, -1, -1),
                                 ostride=(1, 1, 1), ooffset=(0, 0, 0), cout=cin)
                _igemm(g, wp, None, gx, p)
            else:
                _scatter_conv(g, weight, None, gx, stride, padding, K, swap=1, act="none", slope=0.0)
        if ctx.needs_input_grad[1]:
            dw = _wgrad_wpairs(g, x, (O, K, stride, padding), tuple(weight.shape), False) if stride[2] == 2 else None
            if dw is None:
                p = _conv_params(R=O, T=K, rstride=stride, tbase=[-q for q in padding], tstep=(1, 1, 1), ostride=(1, 1, 1),
                                 ooffset=(0, 0, 0), cout=weight.shape[0])
                dw = _wgrad(g, x, p, weight.shape, channels_last=is_cl_weight(weight))
            dw = dw.to(weight.dtype)
        if db is not None:
            db = db.to(weight.dtype)
        return gx, dw, db, None, None, None, None, None


def conv_kd_as_kw(x, weight, bias, act="none", slope=0.0):
    """A stride-1, unpadded convolution whose kernel only extends along D — (KD,1,1), what is left of a 7^3 stem / tail once
    its in-plane taps are folded into channels — run with D and W exchanged: the kernels reuse staged tiles across the taps
    of the INNERMOST axis only (igemm3, wgrad_kw: 700+ TFLOP/s instead of 130-270 on a (7,1,1) kernel), and they take every
    spatial stride as a parameter, so the exchange is two zero-copy views."""
    y = _Conv3d.apply(enter(x).transpose(2, 4), weight.transpose(2, 4), bias, (1, 1, 1), (0, 0, 0), act, float(slope), True)
    return y.transpose(2, 4)


def _fold_w(x, kw, stride_w, pad):
    """Single-channel input: fold the KW taps along W into channels.  out[n, c, d, h, w'] = xpad[n, 0, d, h, w'*s + c],
    an overlapping strided view of the (zero-padded) channels-last buffer, materialised once as a 8-lane tensor.  A
    1 -> Cout conv with KD*KH*KW taps of ONE channel (31/32 of every 32-deep MFMA step zero) becomes a KW -> Cout conv with
    KD*KH taps: 343 K-steps -> 49 for the generator's 7^3 stem (networks/ResNet.py:283)."""
    x = enter(x)
    if any(pad):
        x = pad3d(x, pad, "zeros")
    return _FoldW.apply(x, int(kw), int(stride_w))


class _FoldW(Function):
    """The W taps of a one-channel tensor as channels (an overlapping strided view, materialised).  Its own autograd node because torch's
    as_strided backward scatters with index_add_ — 16-bit atomics in arrival order: 0.19 ms per call at 130^2 x 64 and the one kernel of the CycleGAN
    step whose result changed from run to run.  Here the gradient of input column w is the sum, in tap order and in fp32, of the taps that read it."""

    @staticmethod
    def forward(ctx, x, kw, stride_w):
        n, _, d, h, w = x.shape
        wo = (w - kw) // stride_w + 1
        sn, _, sd, sh, sw = x.stride()
        ctx.cfg = (tuple(x.shape), kw, stride_w, wo)
        return to_cl(torch.as_strided(x, (n, kw, d, h, wo), (sn, sw, sd, sh, sw * stride_w), x.storage_offset()))

    @staticmethod
    def backward(ctx, g):
        shape, kw, s, wo = ctx.cfg
        gx = torch.zeros(shape, dtype=torch.float32, device=g.device)
        for c in range(kw):
            gx[:, 0, :, :, c:c + s * (wo - 1) + 1:s] += g[:, c]
        return to_cl(gx), None, None


def conv3d(x, weight, bias=None, stride=1, padding=0, act="none", slope=0.0):
    stride, padding = _triple(stride), _triple(padding)
    khw = weight.shape[3] * weight.shape[4]
    if (weight.shape[1] == 1 and x.shape[1] == 1 and x.is_cuda and stride == (1, 1, 1) and 16 < khw <= 64 and weight.shape[0] > 16
            and os.environ.get("RAYGUN_AMD_FOLD_HW", "1") != "0"):
        # one-channel input, big in-plane kernel (the generator's 7^3 stem): (kh,kw) taps -> channels, KD taps remain
        x = enter(x)
        if any(padding):
            x = pad3d(x, padding, "zeros")
        kd, kh, kw = weight.shape[2:]
        if os.environ.get("RAYGUN_AMD_FOLD_DH", "1") != "0":
            # fold (kd, kh) instead: the remaining KW taps run along the memory-fastest axis of an ORDINARY channels-last tensor (no D<->W exchanged
            # views: those put the tap axis a 2-MB plane apart per row); the spread map works on the transposed view and writes the exchanged layout
            xT = x.transpose(2, 4)
            sT = _DiagSpread.apply(xT, kh, kd, 1, xT.shape[3] - kh + 1, xT.shape[4] - kd + 1, True)     # channel kh * KD + kd
            wf = weight.permute(0, 3, 2, 4, 1).reshape(weight.shape[0], kh * kd, 1, 1, kw)
            return _Conv3d.apply(sT.transpose(2, 4), wf, bias, (1, 1, 1), (0, 0, 0), act, float(slope))
        xf = _DiagSpread.apply(x, kh, kw, 1, x.shape[3] - kh + 1, x.shape[4] - kw + 1)
        wf = weight.permute(0, 3, 4, 2, 1).reshape(weight.shape[0], kh * kw, kd, 1, 1)  # [cout][(kh,kw)][kd]
        return conv_kd_as_kw(xf, wf, bias, act, slope)
    if weight.shape[1] == 1 and 1 < weight.shape[4] <= 8 and x.shape[1] == 1 and x.is_cuda:
        xf = _fold_w(x, weight.shape[4], stride[2], padding)
        wf = weight.permute(0, 4, 2, 3, 1)  # (Cout, KW, KD, KH, 1): autograd maps the gradient back
        return _Conv3d.apply(xf, wf, bias, (stride[0], stride[1], 1), (0, 0, 0), act, float(slope))
    return _Conv3d.apply(x, weight, bias, stride, padding, act, float(slope))


def conv3d_linear_pair(x, w_a, b_a, w_b, b_b, padding=0, act="none", slope=0.0):
    """act(conv1x1(conv(x, w_a, b_a, padding), w_b, b_b)) for a stride-1 conv followed DIRECTLY by a 1x1x1 conv (no
    norm/activation in between), evaluated as ONE convolution with the composed weights — the tail of every raygun
    generator, Sequential(ResnetGenerator3D[..., Conv3d(ngf, ngf, 7)], Conv3d(ngf, output_nc, 1)) (networks/ResNet.py:338 +
    systems/BaseSystem.py:209-224): 64->64 7^3 followed by 64->1 is 64x the FLOPs of the single 64->1 7^3 map it equals.

      W'[p,c,t] = sum_m w_b[p,m] w_a[m,c,t]        b' = w_b b_a + b_b

    The P-output conv with KW taps along W is run as a (P*KW)-output conv with (KD,KH,1) taps — one output channel per
    (p, kw) — followed by the fixed diagonal gather y[p,w] = sum_kw z[p*KW+kw, w+kw] (a (1,1,KW) conv with one-hot weights):
    the first stage has KW times fewer taps to re-read the 64-channel input for.  Everything is ordinary autograd over
    ops.conv3d; the einsum/permute on the (tiny) weights carries the gradients back to w_a, b_a, w_b, b_b."""
    padding = _triple(padding)
    P, M = w_b.shape[0], w_b.shape[1]
    KD, KH, KW = w_a.shape[2:]
    wc = torch.einsum("pm,mcdhw->pcdhw", w_b.reshape(P, M).float(), w_a.float())        # (P, Cin, KD, KH, KW)
    bc = None
    if b_a is not None or b_b is not None:
        bc = torch.zeros(P, dtype=torch.float32, device=w_a.device)
        if b_a is not None:
            bc = bc + w_b.reshape(P, M).float() @ b_a.float()
        if b_b is not None:
            bc = bc + b_b.float()
    if P == 1 and KH * KW <= 64 and os.environ.get("RAYGUN_AMD_FOLD_HW", "1") != "0":
        # (kh,kw) -> output channels: a KD-tap conv to KH*KW channels over the padded plane, then the diagonal gather
        x = enter(x)
        if any(padding):
            x = pad3d(x, padding, "zeros")                                                # H, W extents grow by 2*pad
        if os.environ.get("RAYGUN_AMD_FOLD_DH", "1") != "0" and KD * KH <= 64:
            # (kd, kh) -> output channels instead: the KW taps run along the memory-fastest axis of the ordinary channels-last input (conv3d's fold
            # branch has the reason); the diagonal gather then works on the D<->W transposed view and hands back the exchanged layout
            w1 = wc.permute(0, 3, 2, 1, 4).reshape(KH * KD, wc.shape[1], 1, 1, KW)       # channel kh*KD+kd, taps along W
            z = _Conv3d.apply(x, w1, None, (1, 1, 1), (0, 0, 0), "none", 0.0)
            zT = z.transpose(2, 4)
            yT = _DiagSum.apply(zT, KH, KD, 1, zT.shape[3] - KH + 1, zT.shape[4] - KD + 1, bc, True)
            return activation(yT.transpose(2, 4), act, slope)
        w1 = wc.permute(0, 3, 4, 1, 2).reshape(KH * KW, wc.shape[1], KD, 1, 1)           # channel kh*KW+kw
        z = conv_kd_as_kw(x, w1, None)
        y = _DiagSum.apply(z, KH, KW, 1, z.shape[3] - KH + 1, z.shape[4] - KW + 1, bc)
        return activation(y, act, slope)
    w1 = wc.permute(0, 4, 1, 2, 3).reshape(P * KW, wc.shape[1], KD, KH, 1)               # channel p*KW+kw <- tap kw
    z = conv3d(x, w1, None, 1, padding, "none", 0.0)                                      # W extent grows by 2*pad_w
    eye = torch.zeros(P, P * KW, 1, 1, KW, dtype=torch.float32, device=w_a.device)
    for pp in range(P):
        for k in range(KW):
            eye[pp, pp * KW + k, 0, 0, k] = 1.0
    return conv3d(z, eye, bc, 1, 0, act, slope)


class _ConvTranspose3d(Function):
    """torch.nn.ConvTranspose3d (+ fused activation).  Reference: networks/ResNet.py:325-333, networks/UNet.py:263-269."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, output_padding, act, slope):
        _require_cuda(x, "conv_transpose3d")
        x = as_cl(x)
        n, cin, D, H, W = x.shape
        if weight.shape[0] != cin:
            raise RuntimeError(f"ConvTranspose3d: expected input with {weight.shape[0]} channels, got {cin}")
        cout = weight.shape[1]
        K = tuple(weight.shape[2:])
        O = [(i - 1) * s - 2 * p + k + op for i, s, p, k, op in zip((D, H, W), stride, padding, K, output_padding)]
        y = cl_empty(n, cout, *O, x.device)
        b = bias.detach().float().contiguous() if bias is not None else None
        _scatter_conv(x, weight, b, y, stride, padding, K, swap=1, act=act, slope=slope)
        ctx.save_for_backward(x, weight, y if act != "none" else None)
        ctx.cfg = (stride, padding, act, slope, bias is not None, K, O)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        stride, padding, act, slope, has_bias, K, O = ctx.cfg
        gy = as_cl(gy)
        g, db = _act_bwd_bias(gy, y, act, slope, has_bias and ctx.needs_input_grad[2])
        gx = dw = None
        n, cin, D, H, W = x.shape
        p = _conv_params(R=(D, H, W), T=K, rstride=stride, tbase=[-q for q in padding], tstep=(1, 1, 1),
                         ostride=(1, 1, 1), ooffset=(0, 0, 0), cout=cin)
        if ctx.needs_input_grad[0]:
            gx = cl_empty(n, cin, D, H, W, x.device)
            _igemm(g, pack_weights(weight, swap=0), None, gx, p)
        if ctx.needs_input_grad[1]:
            dw = _wgrad_wpairs(x, g, ((D, H, W), K, stride, padding), tuple(weight.shape), True) if stride[2] == 2 else None
            if dw is None:
                dw = _wgrad(x, g, p, weight.shape)
            dw = dw.to(weight.dtype)
        if db is not None:
            db = db.to(weight.dtype)
        return gx, dw, db, None, None, None, None, None


def conv_transpose3d(x, weight, bias=None, stride=1, padding=0, output_padding=0, act="none", slope=0.0):
    return _ConvTranspose3d.apply(x, weight, bias, _triple(stride), _triple(padding), _triple(output_padding), act,
                                  float(slope))


# ---------------------------------------------------------------------------------------------- activation
class _Act(Function):
    """Stand-alone activation (when it cannot be fused into a conv / norm epilogue)."""

    @staticmethod
    def forward(ctx, x, act, slope):
        x = as_cl(x)
        n, c, d, h, w = x.shape
        y = cl_empty(n, c, d, h, w, x.device)
        # identity "norm" (mean 0, rstd 1) + activation == plain activation: reuse the fused apply kernel
        mr = torch.tensor([0.0, 1.0], dtype=torch.float32, device=x.device).repeat(n * c)
        _lib.check(_L().rg_instnorm_apply(C.byref(desc(x)), C.c_void_p(mr.data_ptr()), None, C.byref(desc(y)),
                                                ACT[act], float(slope), _stream()), "rg_instnorm_apply")
        ctx.save_for_backward(y)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        g, _ = _act_bwd_bias(as_cl(gy), y, ctx.cfg[0], ctx.cfg[1], False)
        return g, None, None


def activation(x, act, slope=0.0):
    _require_cuda(x, "activation")
    if act == "none":
        return x
    return _Act.apply(x, act, float(slope))


# ---------------------------------------------------------------------------------------------- max pool
class _MaxPool3d(Function):
    """torch.nn.MaxPool3d(k, stride=k) floor mode (MaxDownsample, networks/UNet.py:209-228)."""

    @staticmethod
    def forward(ctx, x, factor):
        _require_cuda(x, "maxpool3d")
        x = as_cl(x)
        n, c, D, H, W = x.shape
        O = [D // factor[0], H // factor[1], W // factor[2]]
        if min(O) <= 0:
            raise RuntimeError(f"max_pool3d(): output size too small for input {tuple(x.shape)} and kernel {factor}")
        y = cl_empty(n, c, *O, x.device)
        _lib.check(_L().rg_maxpool3d_fwd(C.byref(desc(x)), C.byref(desc(y)), i3(factor), _stream()), "rg_maxpool3d_fwd")
        ctx.save_for_backward(x)
        ctx.factor = factor
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = as_cl(gy)
        n, c, D, H, W = x.shape
        gx = cl_empty(n, c, D, H, W, x.device)
        _lib.check(_L().rg_maxpool3d_bwd(C.byref(desc(x)), C.byref(desc(gy)), C.byref(desc(gx)), i3(ctx.factor), _stream()),
                   "rg_maxpool3d_bwd")
        return gx, None


def maxpool3d(x, factor):
    return _MaxPool3d.apply(x, _triple(factor))


# ---------------------------------------------------------------------------------------------- upsample+crop+concat
class _UpCat(Function):
    """cat([centre_crop(f), centre_crop(nearest_upsample(g))], 1) — Upsample.forward, networks/UNet.py:338-351."""

    @staticmethod
    def forward(ctx, f, g, factor, out_spatial):
        _require_cuda(f, "upcat")
        f, g = as_cl(f), as_cl(g)
        n, cf = f.shape[:2]
        cg = g.shape[1]
        up = [a * b for a, b in zip(g.shape[2:], factor)]
        off_g = [(a - b) // 2 for a, b in zip(up, out_spatial)]
        off_f = [(a - b) // 2 for a, b in zip(f.shape[2:], out_spatial)]
        out = cl_empty(n, cf + cg, *out_spatial, f.device)
        _lib.check(_L().rg_upcat_fwd(C.byref(desc(f)), C.byref(desc(g)), C.byref(desc(out)), i3(factor), i3(off_f),
                                           i3(off_g), _stream()), "rg_upcat_fwd")
        ctx.cfg = (factor, off_f, off_g, tuple(f.shape), tuple(g.shape))
        return out

    @staticmethod
    def backward(ctx, go):
        factor, off_f, off_g, fs, gs = ctx.cfg
        go = as_cl(go)
        gf = cl_empty(*fs, go.device)
        gg = cl_empty(*gs, go.device)
        _lib.check(_L().rg_upcat_bwd(C.byref(desc(go)), C.byref(desc(gf)), C.byref(desc(gg)), i3(factor), i3(off_f),
                                           i3(off_g), _stream()), "rg_upcat_bwd")
        return gf, gg, None, None


def upcat(f, g, factor, out_spatial):
    return _UpCat.apply(f, g, _triple(factor), tuple(int(v) for v in out_spatial))


class _SubpixelWeights(Function):
    """weight (O, Cf+Cg, 3,3,3) -> (weight[:, :Cf], the (8*O, Cg, 2,2,2) sub-pixel kernels of weight[:, Cf:]) — see
    rg_subpixel_weights: output parity p of a dimension sees taps {0,1 | 2} (p = 0) or {0 | 1,2} (p = 1) fall on low-resolution
    cells s = 0 | 1, so the taps that share a cell are summed; output channel = ((pd*2+ph)*2+pw)*O + o.  Both results are
    channels-last fp32; the backward writes ONE gradient in the weight's own layout (no slice / stack temporaries)."""

    @staticmethod
    def forward(ctx, w, cf):
        _require_cuda(w, "subpixel_weights")
        if w.dtype != torch.float32 or w.dim() != 5 or tuple(w.shape[2:]) != (3, 3, 3) or not is_dense(w):
            raise RuntimeError("raygun_amd: subpixel_weights expects a dense fp32 (O, C, 3, 3, 3) weight")
        o, c = w.shape[:2]
        wf = _debug.empty((o, cf, 3, 3, 3), torch.float32, w.device, channels_last_3d=True)
        weff = _debug.empty((8 * o, c - cf, 2, 2, 2), torch.float32, w.device, channels_last_3d=True)
        _lib.check(_L().rg_subpixel_weights(C.byref(desc(w)), cf, C.c_void_p(wf.data_ptr()), C.c_void_p(weff.data_ptr()), _stream()),
                   "rg_subpixel_weights")
        ctx.cfg = (cf, tuple(w.shape), tuple(w.stride()))
        return wf, weff

    @staticmethod
    def backward(ctx, gwf, gweff):
        cf, shape, strides = ctx.cfg
        ref = gwf if gwf is not None else gweff
        dw = torch.empty_strided(shape, strides, dtype=torch.float32, device=ref.device)
        cl = lambda t: None if t is None else t.float().contiguous(memory_format=torch.channels_last_3d)  # noqa: E731
        gwf, gweff = cl(gwf), cl(gweff)
        _lib.check(_L().rg_subpixel_weights_bwd(C.c_void_p(gwf.data_ptr()) if gwf is not None else None,
                                                      C.c_void_p(gweff.data_ptr()) if gweff is not None else None,
                                                      C.byref(desc(dw)), cf, _stream()), "rg_subpixel_weights_bwd")
        return dw, None


def upcat_conv3d(f, g, target, weight, bias, act="none", slope=0.0):
    """act(conv3d(cat([centre_crop(f), centre_crop(nearest_upsample_x2(g))], 1), weight, bias)) for a 3x3x3 valid convolution
    — Upsample.forward + the first layer of the following ConvPass (networks/UNet.py:338-351, :70-95) — WITHOUT building the
    upsampled tensor: the upsampled operand is constant over 2x2x2 cells, so its 27 taps collapse to 8 taps on the
    low-resolution tensor (one 2^3 kernel per output parity: _SubpixelWeights), evaluated as ONE convolution with 8*Cout output
    channels on the low-resolution grid followed by a depth-to-space shuffle.  27/8 = 3.4x fewer FLOPs on the upsampled
    channels in forward, backward-data and backward-weights (for config U: 2592 of the 3024 input channels of the widest
    layer).  Same parameters, same state_dict, same gradients (autograd through ordinary ops); the two partial sums are
    rounded to bf16 before they are added."""
    f, g = enter(f), enter(g)
    cf, cg, cout = f.shape[1], g.shape[1], weight.shape[0]
    T = tuple(int(v) for v in target)
    c = [(2 * a - t) // 2 for a, t in zip(g.shape[2:], T)]            # crop offset of the upsampled tensor
    jlo = [ci >> 1 for ci in c]
    jhi = [(t - 3 + ci) >> 1 for t, ci in zip(T, c)]
    if torch.is_grad_enabled() and weight.requires_grad:
        wf, weff = _SubpixelWeights.apply(weight, cf)
    else:  # inference (the block runner): split once per weight version; the packed operands are cached on the two tensors
        cache = getattr(weight, "_rg_pack_cache", None)
        ver = (weight._version, weight.data_ptr())
        if cache is None or cache.get("ver") != ver:
            cache = {"ver": ver}
            try:
                weight._rg_pack_cache = cache
            except Exception:
                pass
        if ("subpix_up", cf) not in cache:
            with torch.no_grad():
                cache[("subpix_up", cf)] = _SubpixelWeights.apply(weight.detach(), cf)
        wf, weff = cache[("subpix_up", cf)]
    y = conv3d(center_crop(f, T), wf, bias)
    gw = crop(g, jlo, [b - a + 2 for a, b in zip(jlo, jhi)])
    z = conv3d(gw, weff, None)                                         # (N, 8*Cout, J, J, J), channels-last
    return _D2SAddAct.apply(y, z, tuple(ci & 1 for ci in c), act, float(slope))


class _D2SAddAct(Function):
    """act(y + depth_to_space(z)[crop]) in one pass, and its backward in one pass (rg_d2s_add_act / _bwd)."""

    @staticmethod
    def forward(ctx, y, z, off, act, slope):
        y, z = as_cl(y), as_cl(z)
        out = cl_empty(*y.shape, y.device)
        _lib.check(_L().rg_d2s_add_act(C.byref(desc(y)), C.byref(desc(z)), C.byref(desc(out)), i3(off), ACT[act], slope, _stream()),
                   "rg_d2s_add_act")
        ctx.save_for_backward(out if act != "none" else None)
        ctx.cfg = (off, act, slope, tuple(z.shape))
        return out

    @staticmethod
    def backward(ctx, gout):
        (out,) = ctx.saved_tensors
        off, act, slope, zs = ctx.cfg
        gout = as_cl(gout)
        g = cl_empty(*gout.shape, gout.device)
        gz = cl_empty(*zs, gout.device)
        _lib.check(_L().rg_d2s_add_act_bwd(C.byref(desc(gout)), C.byref(desc(out)) if out is not None else None, C.byref(desc(g)),
                                                 C.byref(desc(gz)), i3(off), ACT[act], slope, _stream()), "rg_d2s_add_act_bwd")
        return g, gz, None, None, None


# ---------------------------------------------------------------------------------------------- instance norm
class _InstanceNorm(Function):
    """InstanceNorm3d(affine=False, track_running_stats=False) [+act] [+residual]  (networks/ResNet.py:284-285,
    408-409,417,436-443; NLayerDiscriminator.py:154-156)."""

    @staticmethod
    def forward(ctx, x, res, eps, act, slope):
        _require_cuda(x, "instance_norm")
        x = as_cl(x)
        if res is not None:
            if act != "none":
                raise RuntimeError("raygun_amd.instance_norm: residual add is fused only without activation")
            res = as_cl(res)
        n, c, d, h, w = x.shape
        L = _L()
        dx = desc(x)
        nbytes = L.rg_instnorm_workspace_bytes(C.byref(dx)) + n * c * 8
        ws = _debug.empty(nbytes, torch.uint8, x.device)
        mr = _debug.empty(n * c * 2, torch.float32, x.device)
        _lib.check(L.rg_instnorm_stats(C.byref(dx), C.c_void_p(mr.data_ptr()), float(eps), C.c_void_p(ws.data_ptr()), nbytes,
                                       _stream()), "rg_instnorm_stats")
        y = cl_empty(n, c, d, h, w, x.device)
        _lib.check(L.rg_instnorm_apply(C.byref(dx), C.c_void_p(mr.data_ptr()), C.byref(desc(res)) if res is not None else None,
                                       C.byref(desc(y)), ACT[act], float(slope), _stream()), "rg_instnorm_apply")
        ctx.save_for_backward(x, mr, y if act != "none" else None)
        ctx.cfg = (act, slope, res is not None)
        ctx.mark_non_differentiable(mr)
        ctx.set_materialize_grads(False)   # (or autograd allocates and zero-fills a gradient for `mr` in front of every backward call: 100 launches per CycleGAN step)
        return y, mr

    @staticmethod
    def backward(ctx, gy, _gmr=None):
        if gy is None:
            return None, None, None, None, None
        x, mr, y = ctx.saved_tensors
        act, slope, has_res = ctx.cfg
        gy = as_cl(gy)
        n, c, d, h, w = x.shape
        gx = None
        if ctx.needs_input_grad[0]:
            L = _L()
            dx = desc(x)
            nbytes = L.rg_instnorm_workspace_bytes(C.byref(dx)) + n * c * 8
            ws = _debug.empty(nbytes, torch.uint8, x.device)
            gx = cl_empty(n, c, d, h, w, x.device)
            _lib.check(L.rg_instnorm_bwd(C.byref(dx), C.c_void_p(mr.data_ptr()), C.byref(desc(y)) if y is not None else None,
                                         C.byref(desc(gy)), C.byref(desc(gx)), ACT[act], float(slope),
                                         C.c_void_p(ws.data_ptr()), nbytes, _stream()), "rg_instnorm_bwd")
        return gx, (gy if has_res and ctx.needs_input_grad[1] else None), None, None, None


def instance_norm(x, eps=1e-5, act="none", slope=0.0, residual=None, return_stats=False):
    """return_stats: also the float32 (N*C, 2) tensor of (mean, 1/sqrt(biased var + eps)) per instance and channel."""
    y, mr = _InstanceNorm.apply(x, residual, float(eps), act, float(slope))
    return (y, mr.view(-1, 2)) if return_stats else y


class _NormAffine(Function):
    """y = act(xhat * gamma + beta) [+ residual] with xhat from instance statistics ("instance": InstanceNormNd(affine=True)), batch statistics
    ("batch": BatchNormNd in training mode — mean / biased variance over batch x voxels) or a given per-channel table ("fixed": eval mode with
    running statistics).  gamma / beta may be None (1 / 0).  Returns (y, per-channel or per-(n,c) (mean, biased var) the module needs for its
    running statistics).  Reference users: NLayerDiscriminator's default norm_layer (NLayerDiscriminator.py:107-115), any norm_layer argument
    of the generators (networks/ResNet.py:231-262)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, mode, eps, act, slope, fixed_mean, fixed_var):
        _require_cuda(x, "norm_affine")
        x = as_cl(x)
        if res is not None:
            if act != "none":
                raise RuntimeError("raygun_amd.norm_affine: residual add is fused only without activation")
            res = as_cl(res)
        n, c, d, h, w = x.shape
        L = _L()
        dx = desc(x)
        stats = None
        if mode == "fixed":
            rstd = torch.rsqrt(fixed_var.float() + eps)
            table = torch.stack([fixed_mean.float(), rstd], 1).repeat(n, 1).contiguous().view(-1)
        else:
            nbytes = L.rg_instnorm_workspace_bytes(C.byref(dx)) + n * c * 8
            ws = _debug.empty(nbytes, torch.uint8, x.device)
            table = _debug.empty(n * c * 2, torch.float32, x.device)
            _lib.check(L.rg_instnorm_stats(C.byref(dx), C.c_void_p(table.data_ptr()), float(eps), C.c_void_p(ws.data_ptr()), nbytes, _stream()),
                       "rg_instnorm_stats")
            t = table.view(n, c, 2)
            mean_nc, var_nc = t[:, :, 0], (1.0 / (t[:, :, 1] * t[:, :, 1]) - eps).clamp_(min=0)
            if mode == "batch":
                mean_c = mean_nc.mean(0)
                var_c = ((var_nc + mean_nc * mean_nc).mean(0) - mean_c * mean_c).clamp_(min=0)
                table = torch.stack([mean_c, torch.rsqrt(var_c + eps)], 1).repeat(n, 1).contiguous().view(-1)
                stats = torch.stack([mean_c, var_c], 1)                     # (C, 2)
            else:
                stats = torch.stack([mean_nc, var_nc], 2)                   # (N, C, 2)
        gb = None
        if gamma is not None or beta is not None:
            g1 = gamma.detach().float() if gamma is not None else torch.ones(c, device=x.device)
            b1 = beta.detach().float() if beta is not None else torch.zeros(c, device=x.device)
            gb = torch.stack([g1, b1], 1).contiguous().view(-1)
        y = cl_empty(n, c, d, h, w, x.device)
        _lib.check(L.rg_norm_apply_affine(C.byref(dx), C.c_void_p(table.data_ptr()), C.c_void_p(gb.data_ptr()) if gb is not None else None,
                                          C.byref(desc(res)) if res is not None else None, C.byref(desc(y)), ACT[act], float(slope), _stream()),
                   "rg_norm_apply_affine")
        ctx.save_for_backward(x, table, gb, y if act != "none" else None)
        ctx.cfg = (mode, act, slope, res is not None, gamma is not None, beta is not None)
        if stats is None:
            stats = table.new_zeros(0)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return y, stats

    @staticmethod
    def backward(ctx, gy, _gstats=None):
        if gy is None:
            return (None,) * 10
        x, table, gb, y = ctx.saved_tensors
        mode, act, slope, has_res, has_g, has_b = ctx.cfg
        gy = as_cl(gy)
        n, c, d, h, w = x.shape
        nvox = d * h * w
        L = _L()
        dx = desc(x)
        part = L.rg_instnorm_workspace_bytes(C.byref(dx))
        nbytes = part + n * c * 8
        ws = _debug.empty(nbytes, torch.uint8, x.device)
        gx = cl_empty(n, c, d, h, w, x.device)
        # (fixed statistics: the two means are still reduced — they are d beta / d gamma — but gx must not subtract them: see below)
        _lib.check(L.rg_norm_bwd_affine(C.byref(dx), C.c_void_p(table.data_ptr()), C.c_void_p(gb.data_ptr()) if gb is not None else None,
                                        1 if mode == "batch" else 0, C.byref(desc(y)) if y is not None else None, C.byref(desc(gy)), C.byref(desc(gx)),
                                        ACT[act], float(slope), C.c_void_p(ws.data_ptr()), nbytes, _stream()), "rg_norm_bwd_affine")
        gm = ws[part:part + n * c * 8].view(torch.float32).view(n, c, 2)
        dbeta = (gm[:, :, 0].sum(0) * nvox) if has_b else None
        dgamma = (gm[:, :, 1].sum(0) * nvox) if has_g else None
        if mode == "fixed":   # constants, not statistics of x: gx = gamma * rstd * g, through the apply kernel with a (0, rstd) table
            g, _ = _act_bwd_bias(gy, y, act, slope, False)
            t0 = table.view(-1, 2).clone(); t0[:, 0] = 0
            gonly = None
            if gb is not None:
                gonly = gb.view(-1, 2).clone(); gonly[:, 1] = 0; gonly = gonly.view(-1)
            _lib.check(L.rg_norm_apply_affine(C.byref(desc(g)), C.c_void_p(t0.data_ptr()), C.c_void_p(gonly.data_ptr()) if gonly is not None else None, None,
                                              C.byref(desc(gx)), ACT["none"], 0.0, _stream()), "rg_norm_apply_affine")
        if not ctx.needs_input_grad[0]:
            gx = None
        return gx, (gy if has_res and ctx.needs_input_grad[1] else None), dgamma, dbeta, None, None, None, None, None, None


def norm_affine(x, gamma, beta, mode="instance", eps=1e-5, act="none", slope=0.0, residual=None, mean=None, var=None):
    """(y, stats): see _NormAffine.  mode "fixed" takes the per-channel `mean` / `var` (running statistics)."""
    return _NormAffine.apply(x, residual, gamma, beta, mode, float(eps), act, float(slope), mean, var)


class _NormFixed(Function):
    """y = act((x - mean[c]) * rstd[c]) [+ residual] with GIVEN per-channel statistics — InstanceNorm(track_running_stats=True) in
    eval mode (the CycleGAN's `freeze_norms_at` / set_norm_mode("fix_norms"), networks/utils.py:31-42, models/FreezableModel.py):
    the apply kernel of the instance norm with a constant table; backward is the same kernel with mean 0 on g = gy * act'(y)."""

    @staticmethod
    def forward(ctx, x, res, mean, rstd, act, slope):
        _require_cuda(x, "norm_fixed")
        x = as_cl(x)
        if res is not None:
            if act != "none":
                raise RuntimeError("raygun_amd.norm_fixed: residual add is fused only without activation")
            res = as_cl(res)
        n, c, d, h, w = x.shape
        mr = torch.stack([mean.float(), rstd.float()], 1).repeat(n, 1).contiguous().view(-1)       # (n*c*2,)
        y = cl_empty(n, c, d, h, w, x.device)
        _lib.check(_L().rg_instnorm_apply(C.byref(desc(x)), C.c_void_p(mr.data_ptr()), C.byref(desc(res)) if res is not None else None,
                                          C.byref(desc(y)), ACT[act], float(slope), _stream()), "rg_instnorm_apply")
        ctx.save_for_backward(torch.stack([torch.zeros_like(rstd, dtype=torch.float32), rstd.float()], 1).repeat(n, 1).contiguous().view(-1),
                              y if act != "none" else None)
        ctx.cfg = (act, slope, res is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        mr0, y = ctx.saved_tensors
        act, slope, has_res = ctx.cfg
        gy = as_cl(gy)
        gx = None
        if ctx.needs_input_grad[0]:
            g, _ = _act_bwd_bias(gy, y, act, slope, False)
            gx = cl_empty(*gy.shape, gy.device)
            _lib.check(_L().rg_instnorm_apply(C.byref(desc(g)), C.c_void_p(mr0.data_ptr()), None, C.byref(desc(gx)), ACT["none"], 0.0, _stream()),
                       "rg_instnorm_apply")
        return gx, (gy if has_res and ctx.needs_input_grad[1] else None), None, None, None, None


def norm_fixed(x, mean, rstd, act="none", slope=0.0, residual=None):
    return _NormFixed.apply(x, residual, mean, rstd, act, float(slope))


# ---------------------------------------------------------------------------------------------- padding / add
class _Pad3d(Function):
    """ReflectionPad3d / zero pad (networks/ResNet.py:270,337,392)."""

    @staticmethod
    def forward(ctx, x, pad, mode):
        _require_cuda(x, "pad3d")
        x = as_cl(x)
        n, c, d, h, w = x.shape
        if mode == 1 and any(p >= s for p, s in zip(pad, (d, h, w))):
            raise RuntimeError(f"Argument #4: Padding size should be less than the corresponding input dimension, "
                               f"but got: padding {pad} at input {tuple(x.shape)}")
        y = cl_empty(n, c, d + 2 * pad[0], h + 2 * pad[1], w + 2 * pad[2], x.device)
        _lib.check(_L().rg_pad3d_fwd(C.byref(desc(x)), C.byref(desc(y)), i3(pad), mode, _stream()), "rg_pad3d_fwd")
        ctx.cfg = (pad, mode, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, gy):
        pad, mode, xs = ctx.cfg
        gy = as_cl(gy)
        gx = cl_empty(*xs, gy.device)
        _lib.check(_L().rg_pad3d_bwd(C.byref(desc(gy)), C.byref(desc(gx)), i3(pad), mode, _stream()), "rg_pad3d_bwd")
        return gx, None, None


def pad3d(x, pad, mode="reflect"):
    return _Pad3d.apply(x, _triple(pad), {"zeros": 0, "reflect": 1, "replicate": 2}[mode])


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        _require_cuda(a, "add")
        a, b = as_cl(a), as_cl(b)
        y = cl_empty(*a.shape, a.device)
        _lib.check(_L().rg_add(C.byref(desc(a)), C.byref(desc(b)), C.byref(desc(y)), _stream()), "rg_add")
        return y

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _Add.apply(a, b)


class _DiagSpread(Function):
    """y[.,h,w][kh*KW+kw] = x[.,h+s*kh,w+s*kw][0] on an (oh, ow) output grid; adjoint: _DiagSum with -s."""

    @staticmethod
    def forward(ctx, x, kh, kw, sign, oh, ow, dw=False):
        _require_cuda(x, "diag_spread")
        x = as_cl(x)
        n, _, d, h, w = x.shape
        y = (cl_empty_dw if dw else cl_empty)(n, kh * kw, d, oh, ow, x.device)   # dw: memory with D and W exchanged (the caller transposes back)
        _lib.check(_L().rg_diag_spread(C.byref(desc(x)), C.byref(desc(y)), kh, kw, sign, _stream()), "rg_diag_spread")
        ctx.cfg = (kh, kw, sign, h, w, d > 1 and w > 1 and x.stride(2) < x.stride(4))   # (input was itself a D<->W exchanged view: so is its gradient)
        return y

    @staticmethod
    def backward(ctx, g):
        kh, kw, sign, h, w, in_dw = ctx.cfg
        return _DiagSum.apply(g, kh, kw, -sign, h, w, None, in_dw), None, None, None, None, None, None


class _DiagSum(Function):
    """y[.,h,w][0] = bias + sum_{kh,kw} x[.,h+s*kh,w+s*kw][kh*KW+kw] on an (oh, ow) grid; adjoint: _DiagSpread with -s."""

    @staticmethod
    def forward(ctx, x, kh, kw, sign, oh, ow, bias, dw=False):
        _require_cuda(x, "diag_sum")
        x = as_cl(x)
        n, _, d, h, w = x.shape
        y = (cl_empty_dw if dw else cl_empty)(n, 1, d, oh, ow, x.device)
        b = bias.detach().float().contiguous() if bias is not None else None
        _lib.check(_L().rg_diag_sum(C.byref(desc(x)), C.byref(desc(y)), kh, kw, sign,
                                          C.c_void_p(b.data_ptr()) if b is not None else None, _stream()), "rg_diag_sum")
        ctx.cfg = (kh, kw, sign, h, w, bias is not None, d > 1 and w > 1 and x.stride(2) < x.stride(4))
        return y

    @staticmethod
    def backward(ctx, g):
        kh, kw, sign, h, w, has_bias, in_dw = ctx.cfg
        gb = None
        if has_bias and ctx.needs_input_grad[6]:
            gb = _act_bwd_bias(as_cl(g), None, "none", 0.0, True)[1].reshape(1)
        return _DiagSpread.apply(g, kh, kw, -sign, h, w, in_dw), None, None, None, None, None, gb, None


class _Fork(Function):
    """x -> (x, x) for an activation with two consumers (the U-Net's skip connections): the two gradients are summed by the
    rg_add kernel instead of torch's strided add on padded channels-last views (1.0 ms for 72 x 92^3 on MI355X, 20x slower)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return ga if gb is None else gb
        ga, gb = as_cl(ga), as_cl(gb)
        y = cl_empty(*ga.shape, ga.device)
        _lib.check(_L().rg_add(C.byref(desc(ga)), C.byref(desc(gb)), C.byref(desc(y)), _stream()), "rg_add")
        return y


def fork(x):
    return _Fork.apply(x)


class _Crop(Function):
    """Zero-copy spatial crop whose backward zero-pads with one rg_add_pad pass, channels-last (torch: NCDHW zeros + strided
    copy, which the next kernel would have to convert back)."""

    @staticmethod
    def forward(ctx, x, off, shape):
        ctx.cfg = (tuple(x.shape), off)
        return x[..., off[0]:off[0] + shape[0], off[1]:off[1] + shape[1], off[2]:off[2] + shape[2]]

    @staticmethod
    def backward(ctx, g):
        shape, off = ctx.cfg
        g = as_cl(g)
        y = cl_empty(*shape, g.device)
        _lib.check(_L().rg_add_pad(None, C.byref(desc(g)), C.byref(desc(y)), i3(off), _stream()), "rg_add_pad")
        return y, None, None


def crop(x, off, shape):
    """x[..., off : off + shape] per spatial dimension (zero-copy)."""
    off, shape = tuple(int(v) for v in off), tuple(int(v) for v in shape)
    if x.dim() == 5 and x.requires_grad and is_cl(x) and tuple(x.shape[-3:]) != shape:
        return _Crop.apply(x, off, shape)
    return x[..., off[0]:off[0] + shape[0], off[1]:off[1] + shape[1], off[2]:off[2] + shape[2]]


def center_crop(x, shape):
    """Zero-copy centre crop (networks/UNet.py:97-106,327-336)."""
    return crop(x, [(a - b) // 2 for a, b in zip(x.shape[-3:], shape)], shape)


# ---------------------------------------------------------------------------------------------- losses
_LOSS_OUT = 3 + 3 * 512


class _Loss(Function):
    """Fused loss value + gradient (losses/GANLoss.py:56-58, LinkCycleLoss.py:99-106, WeightedMSELoss_MTLSD.py:11-25)."""

    @staticmethod
    def forward(ctx, kind, pred, target, weight, const_target):
        _require_cuda(pred, "loss")
        p = pred.detach().float().contiguous().view(-1)
        t = target.detach().float().contiguous().view(-1) if target is not None else None
        w = weight.detach().float().contiguous().view(-1) if weight is not None else None
        if t is not None and t.numel() != p.numel():
            raise RuntimeError(f"The size of tensor a ({tuple(pred.shape)}) must match the size of tensor b ({tuple(target.shape)})")
        out = _debug.empty(_LOSS_OUT, torch.float32, p.device)
        grad = _debug.empty(p.shape, p.dtype, p.device)
        _lib.check(_L().rg_loss_fwd_bwd(kind, C.byref(desc(p)), C.byref(desc(t)) if t is not None else None,
                                              C.byref(desc(w)) if w is not None else None, float(const_target), 1.0,
                                              C.byref(desc(grad)), C.c_void_p(out.data_ptr()), _stream()), "rg_loss_fwd_bwd")
        n = p.numel()
        if kind == 2:
            # WeightedMSELoss_MTLSD.py:15-23 without the host sync: mean over w>0 if any scaled != 0 else mean(scaled) (== 0)
            cnt = torch.clamp(out[1], min=1.0)
            loss = torch.where(out[2] > 0, out[0] / cnt, out[0] * 0.0)
            gnorm = torch.where(out[2] > 0, 1.0 / cnt, out[0] * 0.0)
        else:
            loss = out[0] / n
            gnorm = torch.full((), 1.0 / n, dtype=torch.float32, device=p.device)
        ctx.save_for_backward(grad, gnorm)
        ctx.shape, ctx.dtype = pred.shape, pred.dtype
        return loss

    @staticmethod
    def backward(ctx, gl):
        grad, gnorm = ctx.saved_tensors
        g = (grad * (gnorm * gl)).view(ctx.shape).to(ctx.dtype)
        return None, g, None, None, None


def mse_const(pred, target_value):
    return _Loss.apply(0, pred, None, None, float(target_value))


def bce_logits_const(pred, target_value):
    """BCEWithLogitsLoss(pred, full_like(pred, target_value)) (GANLoss "vanilla")."""
    return _Loss.apply(3, pred, None, None, float(target_value))


def mean_signed(pred, sign):
    """sign * mean(pred) (GANLoss "wgangp": -mean for real, +mean for fake)."""
    return _Loss.apply(4, pred, None, None, float(sign))


def smooth_l1(pred, target):
    return _Loss.apply(1, pred, target, None, 0.0)


def weighted_mse(pred, target, weight):
    return _Loss.apply(2, pred, target, weight, 0.0)


# ---------------------------------------------------------------------------------------------- Adam
def nonfinite_check_(t, flag):
    """flag (float32[1], device) <- 1 if the dense fp32 tensor `t` holds an inf / NaN (never cleared here)."""
    if t.dtype != torch.float32 or not is_dense(t):
        raise RuntimeError("raygun_amd.nonfinite_check_: dense fp32 tensor expected")
    _lib.check(_lib.lib("bf16").rg_nonfinite_check(C.c_void_p(t.data_ptr()), t.numel(), C.c_void_p(flag.data_ptr()), _stream()),
               "rg_nonfinite_check")


def adam_step_(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """In-place fused Adam on dense fp32 tensors of one layout (torch.optim.Adam semantics; optimizers/utils.py:4-9).
    If `p` is a channels-last conv weight whose forward operand is packed and current, the kernel rewrites that bf16
    operand too (same values, same rounding as rg_pack_weights) and the cache entry follows the new version."""
    cache = getattr(p, "_rg_pack_cache", None)
    fwd = None
    if cache is not None and cache.get("ver") == (p._version, p.data_ptr()):
        fwd = cache.get("fwd_pack_f16") or cache.get("fwd_pack")   # (f16-storage training packs with the f16 library)
    if fwd is not None and os.environ.get("RAYGUN_AMD_ADAM_PACK", "1") != "0":
        _lib.check(_lib.lib(_KIND[fwd[0].dtype]).rg_adam_step_pack(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                                C.c_void_p(v.data_ptr()), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                                                grad_scale, C.c_void_p(fwd[0].data_ptr()), C.byref(fwd[1]), _stream()),
                   "rg_adam_step_pack")
        return fwd
    _lib.check(_lib.lib("bf16").rg_adam_step(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                       C.c_void_p(v.data_ptr()), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                                       grad_scale, _stream()), "rg_adam_step")
    return None


def adam_step_multi_(items, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, hyp=None):
    """adam_step_ for a list of (p, g, m, v) in ONE launch per 24 tensors (rg_adam_step_multi).  Returns, per item, the refreshed forward-pack record or None
    (as adam_step_ does).  All tensors must be dense fp32 of one layout per item; 16-bit forward packs of ONE storage type (mixed types: two calls).
    hyp: a device float32[3] = (lr, 1 - beta1^step, sqrt(1 - beta2^step)) that the kernel reads instead of `lr` / `step` (graph capture: rg_adam_step_multi_hyp)."""
    n = len(items)
    if n == 0:
        return []
    fwds = []
    for p, g, m, v in items:
        cache = getattr(p, "_rg_pack_cache", None)
        fwd = None
        if cache is not None and cache.get("ver") == (p._version, p.data_ptr()) and os.environ.get("RAYGUN_AMD_ADAM_PACK", "1") != "0":
            fwd = cache.get("fwd_pack_f16") or cache.get("fwd_pack")
        fwds.append(fwd)
    kinds = {_KIND[f[0].dtype] for f in fwds if f is not None}
    if len(kinds) > 1:      # (never in practice: a network trains in one storage type)
        out = [None] * n
        for kind in kinds:
            idx = [i for i, f in enumerate(fwds) if f is not None and _KIND[f[0].dtype] == kind]
            for i, r in zip(idx, adam_step_multi_([items[i] for i in idx], lr, beta1, beta2, eps, weight_decay, step, grad_scale, hyp)):
                out[i] = r
        rest = [i for i, f in enumerate(fwds) if f is None]
        adam_step_multi_([items[i] for i in rest], lr, beta1, beta2, eps, weight_decay, step, grad_scale, hyp)
        return out
    L = _lib.lib(kinds.pop() if kinds else "bf16")
    VP, I64 = C.c_void_p * n, C.c_int64 * n
    P = VP(*[p.data_ptr() for p, _, _, _ in items]); G = VP(*[g.data_ptr() for _, g, _, _ in items])
    M = VP(*[m.data_ptr() for _, _, m, _ in items]); V = VP(*[v.data_ptr() for _, _, _, v in items])
    N = I64(*[p.numel() for p, _, _, _ in items])
    PK = VP(*[(f[0].data_ptr() if f is not None else None) for f in fwds])
    PP = VP(*[(C.addressof(f[1]) if f is not None else None) for f in fwds])
    if hyp is not None:
        if hyp.dtype != torch.float32 or not hyp.is_cuda or hyp.numel() < 3 or not hyp.is_contiguous():
            raise RuntimeError("raygun_amd.adam_step_multi_: hyp must be a contiguous CUDA float32 tensor of 3 values")
        _lib.check(L.rg_adam_step_multi_hyp(P, G, M, V, N, PK, PP, n, lr, beta1, beta2, eps, weight_decay, max(int(step), 1), grad_scale,
                                            C.c_void_p(hyp.data_ptr()), _stream()), "rg_adam_step_multi_hyp")
        return fwds
    _lib.check(L.rg_adam_step_multi(P, G, M, V, N, PK, PP, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, _stream()), "rg_adam_step_multi")
    return fwds
