"""Block-wise volume prediction — the hot part of raygun.predict / raygun.torch.predict.worker without daisy/zarr
(reference: src/raygun/predict.py:73-108 ROI arithmetic, :154-164 block task; src/raygun/torch/predict/worker.py:94-160
per-block normalise -> forward -> crop -> convert -> write).

  * read ROI = input_shape, write ROI = output_shape centred in it (context = (read - write)//2, predict.py:92-96), or
    read_size / crop (predict.py:98-108); blocks tile the volume on the write-ROI grid;
  * blocks whose read ROI does not fit inside the source are skipped (daisy `fit="valid"` default; enumeration order is
    PARITY UNPINNED — results are order independent because write ROIs are disjoint, predict.py:159);
  * the daisy scheduler and its worker processes are replaced by a static partition of the block list over ranks
    (parallel.shard_blocks): no collective, every rank writes only its own blocks;
  * per block: uint8/uint16 source -> (x - iinfo.min)/iinfo.max [* scale + shift] is fused into the layout-conversion
    kernel (rg_copy_convert), the forward runs under no_grad in bf16, outputs are cropped, then either kept as float32 or
    scaled by iinfo(dtype).max and clamped for integer destinations (worker.py:133-149).
"""
import os
import time

import numpy as np
import torch

from . import hipnn, ops, parallel


def roi_geometry(ndims=3, input_shape=None, output_shape=None, read_size=None, crop=0, voxel_size=(1, 1, 1)):
    """-> (read_shape, write_shape, context) in voxels (predict.py:73-108; voxel_size cancels for isotropic voxels)."""
    if input_shape is not None:
        if not isinstance(input_shape, (list, tuple)):
            input_shape = (1,) * (3 - ndims) + (input_shape,) * ndims
            output_shape = (1,) * (3 - ndims) + (output_shape,) * ndims
        read = tuple(int(v) for v in input_shape)
        write = tuple(int(v) for v in output_shape)
        context = tuple((r - w) // 2 for r, w in zip(read, write))
    else:
        read = (1,) * (3 - ndims) + (int(read_size),) * ndims
        c = (0,) * (3 - ndims) + (int(crop),) * ndims
        write = tuple(r - 2 * k for r, k in zip(read, c))
        context = c
    if any(w <= 0 for w in write):
        raise ValueError(f"write ROI {write} is empty for read ROI {read}")
    return read, write, context


def enumerate_blocks(volume_shape, read, write, context):
    """Write-ROI origins (z, y, x) of every block whose read ROI lies inside the volume, raster order."""
    axes = []
    for n, r, w in zip(volume_shape, read, write):
        axes.append([o for o in range(0, n, w) if o + r <= n])
    return [(z + context[0], y + context[1], x + context[2]) for z in axes[0] for y in axes[1] for x in axes[2]]


_TORCH_DT = {np.dtype(np.float32): torch.float32, np.dtype(np.uint8): torch.uint8, np.dtype(np.int16): torch.int16,
             np.dtype(np.int32): torch.int32, np.dtype(np.float16): torch.float16}


def _to_destination(out, dtype, on_device=False):
    """worker.py:133-149: integer destinations get *iinfo.max + clamp, float ones pass through."""
    if np.issubdtype(np.dtype(dtype), np.integer):
        info = np.iinfo(dtype)
        out = torch.clamp(out * float(info.max), float(info.min), float(info.max))
    if on_device:
        return out.to(_TORCH_DT[np.dtype(dtype)])
    return out.cpu().numpy().astype(dtype)


FALLBACK_BLOCKS = []   # block indices the f16 runner re-ran with bf16 storage (diagnostics; appended per call)
COMBINER_STATS = []    # per destination of the last on-disk runs: chunks written once / chunks that went through the locked read-modify-write
NONFINITE_BLOCKS = []  # block indices whose output is inf / NaN even with bf16 storage: stored as the reference stores them, and reported


@torch.no_grad()
def predict_volume(model, source, input_shape=None, output_shape=None, read_size=None, crop=0, ndims=3,
                   scaleShift_input=None, out_dtypes=None, out_channels=None, rank=None, world=None, device=None,
                   destinations=None, precision=None):
    """Predict a (Z,Y,X) integer volume block by block.

    model: callable taking a (1,1,z,y,x) tensor and returning a tensor or tuple of tensors (worker.py:116-119).
    Returns (list of destination ndarrays, number of blocks this rank processed).  Destinations are zero where no block
    wrote (borders; other ranks' blocks when world > 1).  A `source` that is a tensor already on the GPU keeps the whole
    pipeline there: the destinations are device tensors and nothing crosses PCIe.  Any other array-like with `.shape`,
    `.dtype` and slice reads (a numpy array, a `zarrio.Array`) is read block by block, one block ahead of the GPU; given
    `destinations` (numpy arrays or `zarrio.Array`s) are written in place.  On-disk chunks are only written whole when the
    dataset's chunk shape equals the write ROI (what `zarrio.prepare_ds(..., write_size=...)` lays down).

    `precision`: 16-bit storage type of the forward pass, "f16" (default for prediction: IEEE half, 11 significant bits —
    config U's affinities land within 1e-3 relative L2 of the fp32 reference; same kernel speed) or "bf16" (the
    training type).  RAYGUN_AMD_PREDICT_PRECISION overrides the default.  Half's largest value is 65504: a block whose f16 run
    produces a non-finite output (fp32 and bf16 would not overflow there) is predicted again with bf16 storage and logged
    (`predict.FALLBACK_BLOCKS`); a block that is non-finite with bf16 storage as well is the network's own inf / NaN: it is stored as the
    reference stores it (worker.py:133-149), recorded in `predict.NONFINITE_BLOCKS` and logged (RAYGUN_AMD_PREDICT_STRICT=1: FloatingPointError)."""
    precision = precision or os.environ.get("RAYGUN_AMD_PREDICT_PRECISION", "f16")
    with ops.precision(precision):
        return _predict_volume(model, source, input_shape, output_shape, read_size, crop, ndims, scaleShift_input, out_dtypes,
                               out_channels, rank, world, device, destinations)


def _predict_volume(model, source, input_shape, output_shape, read_size, crop, ndims, scaleShift_input, out_dtypes, out_channels,
                    rank, world, device, destinations):
    resident = torch.is_tensor(source) and source.is_cuda
    if resident:
        src_np_dtype = np.dtype(str(source.dtype).replace("torch.", ""))
        if device is None:
            device = source.device
    else:
        if not (hasattr(source, "shape") and hasattr(source, "dtype") and hasattr(source, "__getitem__")):
            source = np.asarray(source)
        src_np_dtype = np.dtype(source.dtype)
    if len(source.shape) != 3:
        raise ValueError("source must be a (Z, Y, X) volume")
    if not np.issubdtype(src_np_dtype, np.integer):
        raise TypeError("source must be an integer volume (worker.py:109-110 normalises with np.iinfo)")
    dev = torch.device(device if device is not None else "cuda")
    read, write, context = roi_geometry(ndims, input_shape, output_shape, read_size, crop)
    blocks = enumerate_blocks(source.shape, read, write, context)
    mine = parallel.shard_blocks(len(blocks), rank, world)
    info = np.iinfo(src_np_dtype)
    scale, shift = 1.0 / float(info.max), -float(info.min) / float(info.max)
    if scaleShift_input is not None:  # worker.py:112-114
        scale, shift = scale * scaleShift_input[0], shift * scaleShift_input[0] + scaleShift_input[1]
    if hasattr(model, "eval"):
        model.eval()
    # on-disk flow (blockio.py): decoded source chunks are cached across blocks, destination chunks are assembled in memory and written once
    from . import blockio
    use_blockio = os.environ.get("RAYGUN_AMD_BLOCKIO", "1") != "0"
    reader_src = source
    if use_blockio and not resident and hasattr(source, "_read_chunk") and hasattr(source, "_chunk_ranges"):
        reader_src = blockio.ChunkCache(source)
    combiners = []
    if use_blockio and destinations is not None:
        my_set = set(mine)
        for d in destinations:
            if hasattr(d, "_write_chunk") and hasattr(d, "_chunk_ranges"):
                lead = tuple(d.shape[:-3])
                def roi(b):
                    wz, wy, wx = blocks[b]
                    return ((0,) * len(lead) + (wz, wy, wx), lead + (wz + write[0], wy + write[1], wx + write[2]))
                kind = blockio.DeviceChunkCombiner if np.dtype(d.dtype) in (np.dtype(np.uint8), np.dtype(np.float32)) else blockio.ChunkCombiner
                kw = dict(device=dev) if kind is blockio.DeviceChunkCombiner else {}
                d._combiner = kind(d, [roi(b) for b in mine], [roi(b) for b in range(len(blocks)) if b not in my_set], **kw)
                combiners.append(d)

    AHEAD = 3                  # blocks the reader thread runs ahead of the GPU
    pinned = []                # AHEAD + 1 reusable pinned staging buffers (a fresh pin_memory() per block costs a page-locking system call of milliseconds)
    uploaded = {}              # staging buffer -> event recorded behind its asynchronous upload: the reader waits for it before it overwrites the buffer

    def fetch(bi, slot):  # host side of one block: slice (or chunk reads + decompression), contiguous, into a pinned staging buffer
        wz, wy, wx = blocks[bi]
        rz, ry, rx = wz - context[0], wy - context[1], wx - context[2]
        a = np.ascontiguousarray(reader_src[rz:rz + read[0], ry:ry + read[1], rx:rx + read[2]])
        if not a.dtype.isnative:
            a = a.astype(a.dtype.newbyteorder("="))
        t = torch.from_numpy(a)
        while len(pinned) <= slot:
            pinned.append(None)
        if pinned[slot] is None or pinned[slot].shape != t.shape or pinned[slot].dtype != t.dtype:
            pinned[slot] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        ev = uploaded.pop(pinned[slot].data_ptr(), None)
        if ev is not None:
            ev.synchronize()
        pinned[slot].copy_(t)
        return pinned[slot]

    n_done = 0
    state = dict(drain=None, destinations=destinations)
    ahead = None
    if not resident and len(mine):
        from concurrent.futures import ThreadPoolExecutor
        reader = ThreadPoolExecutor(max_workers=1)
        import collections
        ahead = collections.deque(reader.submit(fetch, mine[j], j % (AHEAD + 1)) for j in range(min(AHEAD, len(mine))))

    def run_block(bi, host):
        """Forward + crop + convert + hand-over of one block; returns a 0-d device bool: every output value finite."""
        wz, wy, wx = blocks[bi]
        rz, ry, rx = wz - context[0], wy - context[1], wx - context[2]
        if resident:
            blk = source[rz:rz + read[0], ry:ry + read[1], rx:rx + read[2]][None, None]
        else:
            blk = host.to(dev, non_blocking=True)[None, None]
            ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(dev)); uploaded[host.data_ptr()] = ev
        if blk.dtype not in (torch.uint8,):
            blk = blk.float()
        x = ops.to_cl(blk, scale, shift)  # normalisation fused into the NCDHW -> NDHWC bf16 conversion
        with hipnn.scope():               # an enclosing scope: a raygun_amd network hands its outputs over channels-last, 16-bit (no fp32 copy)
            outs = model(x)
        if not isinstance(outs, (tuple, list)):
            outs = (outs,)
        if state["destinations"] is None:
            dd = []
            for i, o in enumerate(outs):
                c = o.shape[1]
                dt = np.float32 if out_dtypes is None else out_dtypes[i]
                shp = ((c,) if c > 1 else ()) + tuple(source.shape)
                dd.append((shp, np.dtype(dt)))
            # OPT-IN (RAYGUN_AMD_PREDICT_HBM_BUDGET = bytes): destinations this call allocates itself live in HBM while the blocks run also for a host-resident
            # source, one download at the end.  Measured (404^3, 20 + 10 channels): 14.6 / 20.1 blocks/s (uint8 / float32) against 19.5 / 24.2 for the drain thread's
            # per-block pinned downloads, which overlap the forward passes — a 2-8 GB pageable download at the end does not — so the drain thread stays the default
            total = sum(int(np.prod(shp)) * dt.itemsize for shp, dt in dd)
            on_dev = resident or (total <= int(os.environ.get("RAYGUN_AMD_PREDICT_HBM_BUDGET", "0")) and all(dt in _TORCH_DT for _, dt in dd))
            state["destinations"] = [torch.zeros(shp, dtype=_TORCH_DT[dt], device=dev) if on_dev else np.zeros(shp, dtype=dt) for shp, dt in dd]
            state["download"] = on_dev and not resident
        dests = state["destinations"]
        dres = all(torch.is_tensor(d) and d.is_cuda for d in dests)
        out_dt = {id(d): (np.dtype(str(d.dtype).replace("torch.", "")) if torch.is_tensor(d) else d.dtype) for d in dests}
        finite = None
        flag = None
        for o, dest in zip(outs, dests):
            odt = np.dtype(out_dt[id(dest)])
            if ops.is_cl(o) and o.dim() == 5 and o.shape[0] == 1 and not ops.lifted() and odt in (np.dtype(np.uint8), np.dtype(np.float32)):
                # one kernel: crop view -> [* 255, clamp, truncate] -> the ROI of the destination, with the inf / NaN check (worker.py:133-149)
                o = o.detach()
                if crop and crop != 0 and input_shape is None:
                    o = o[..., crop:-crop, crop:-crop, crop:-crop]
                if tuple(o.shape[-3:]) != tuple(write):
                    off = [(a - b) // 2 for a, b in zip(o.shape[-3:], write)]
                    o = o[..., off[0]:off[0] + write[0], off[1]:off[1] + write[1], off[2]:off[2] + write[2]]
                if flag is None:
                    flag = torch.zeros(1, dtype=torch.float32, device=dev)
                roi = (Ellipsis, slice(wz, wz + write[0]), slice(wy, wy + write[1]), slice(wx, wx + write[2]))
                if dres:
                    tgt = dest[roi]
                else:
                    tgt = torch.empty(((o.shape[1],) if dest.ndim == 4 else ()) + tuple(write), dtype=_TORCH_DT[odt], device=dev)
                view = tgt if tgt.dim() == 4 else tgt[None]
                if view.shape[0] != o.shape[1]:
                    raise ValueError(f"destination {tuple(dest.shape)} does not match the network's {o.shape[1]} output channels")
                isint = odt == np.dtype(np.uint8)
                ops.store_block(o, view[None], 255.0 if isint else 1.0, 0.0, 255.0, flag)
                if not dres:
                    comb = getattr(dest, "_combiner", None)
                    if comb is not None and hasattr(comb, "dev"):       # chunk buffers in HBM: GPU copies now, one contiguous download per finished chunk (blockio.py)
                        lead = () if tgt.dim() == 3 else (0,)
                        comb.put(list(lead) + [wz, wy, wx], ([tgt.shape[0]] if tgt.dim() == 4 else []) + [wz + write[0], wy + write[1], wx + write[2]], tgt)
                    else:
                        if state["drain"] is None:
                            state["drain"] = _Drain(dev)
                        state["drain"].put(tgt, dest, roi)
                continue
            o = ops.leave(o) if ops.is_cl(o) else o
            o = o.detach().float()[0]
            if crop and crop != 0 and input_shape is None:
                o = o[..., crop:-crop, crop:-crop, crop:-crop]
            if tuple(o.shape[-3:]) != tuple(write):
                off = [(a - b) // 2 for a, b in zip(o.shape[-3:], write)]
                o = o[..., off[0]:off[0] + write[0], off[1]:off[1] + write[1], off[2]:off[2] + write[2]]
            f = torch.isfinite(o).all()
            finite = f if finite is None else finite & f
            if dres:
                dest[..., wz:wz + write[0], wy:wy + write[1], wx:wx + write[2]] = _to_destination(o if dest.ndim == 4 else o[0], out_dt[id(dest)], True)
            elif np.dtype(out_dt[id(dest)]) not in _TORCH_DT:  # e.g. uint16: no torch twin, convert on the host
                dest[..., wz:wz + write[0], wy:wy + write[1], wx:wx + write[2]] = _to_destination(o if dest.ndim == 4 else o[0], out_dt[id(dest)])
            else:  # convert on the GPU, then let the drain thread download and scatter while the next block computes
                if state["drain"] is None:
                    state["drain"] = _Drain(dev)
                state["drain"].put(_to_destination(o if dest.ndim == 4 else o[0], out_dt[id(dest)], True).contiguous(), dest,
                                   (Ellipsis, slice(wz, wz + write[0]), slice(wy, wy + write[1]), slice(wx, wx + write[2])))
        if flag is not None:
            f = (flag == 0)[0]
            finite = f if finite is None else finite & f
        return finite

    try:
        flags = []
        for k, bi in enumerate(mine):
            host = None
            if not resident:
                host = ahead.popleft().result()
                if k + AHEAD < len(mine):      # (its staging slot was last used by block k - 1, whose upload was enqueued before block k - 1's forward)
                    ahead.append(reader.submit(fetch, mine[k + AHEAD], (k + AHEAD) % (AHEAD + 1)))
            flags.append(run_block(bi, host))
            n_done += 1
        # IEEE-half storage overflows at 65504 where fp32 (the reference) and bf16 stay finite, and integer destinations would
        # clamp the inf / NaN silently: the per-block flags are read ONCE, after the last block (no per-block sync), and a block
        # with a non-finite output is predicted again with bf16 storage — its destination region is simply written a second time
        bad = [bi for bi, ok in zip(mine, torch.stack(flags).tolist()) if not ok] if flags else []
        if bad:
            import logging
            log = logging.getLogger("raygun_amd.predict")
            if ops.storage_dtype() != torch.bfloat16:
                log.warning("non-finite output in %d of %d block(s) with f16 storage (half's range is 65504): re-running them with bf16 storage", len(bad), len(mine))
                with ops.precision("bf16"):
                    again = [run_block(bi, None if resident else fetch(bi, AHEAD)) for bi in bad]
                FALLBACK_BLOCKS.extend(bad)
                bad = [bi for bi, ok in zip(bad, torch.stack(again).tolist()) if not ok]
            if bad:
                # bf16 has fp32's range: the network itself produced inf / NaN.  The reference stores what it got (worker.py:133-149: float
                # destinations keep the value, integer ones the clamped product) — so do we: the blocks are recorded and reported, the other
                # blocks' work is not thrown away.  RAYGUN_AMD_PREDICT_STRICT=1 turns the report into an error.
                NONFINITE_BLOCKS.extend(bad)
                msg = f"raygun_amd.predict: non-finite network output in {len(bad)} block(s) with bf16 storage (first: block {bad[0]} at {blocks[bad[0]]})"
                if os.environ.get("RAYGUN_AMD_PREDICT_STRICT", "0") == "1":
                    raise FloatingPointError(msg)
                log.warning(msg + "; stored as the reference stores it (see predict.NONFINITE_BLOCKS)")
    except BaseException:  # a failing block must not leave the download / prefetch threads behind
        if state["drain"] is not None:
            try:
                state["drain"].close()
            except BaseException:
                pass
        if not resident and len(mine):
            reader.shutdown(wait=False)
        for d in combiners:
            try:
                d._combiner.close()
            except BaseException:
                pass
            del d._combiner
        raise
    if state["drain"] is not None:
        state["drain"].close()
    if not resident and len(mine):
        reader.shutdown()
    for d in combiners:       # every chunk this rank owns alone has been written exactly once by now (what is left are chunks of unfinished regions)
        d._combiner.close()
        COMBINER_STATS.append(dict(d._combiner.stats))
        del d._combiner
    if reader_src is not source:
        reader_src.close()
    destinations = state["destinations"]
    if state.get("download") and destinations is not None:      # allocated in HBM for a host-side caller: hand back host arrays, as before
        destinations = [d.cpu().numpy() for d in destinations]
    return destinations, n_done


def predict_zarr(model, source_path, source_dataset, dest_path=None, output_ds=None, out_specs=None, input_shape=None,
                 output_shape=None, read_size=None, crop=0, ndims=3, scaleShift_input=None, out_channels=None, rank=None,
                 world=None, device=None, precision=None):
    """The on-disk flow of raygun.predict (predict.py:70,111-125,154-164) without daisy / zarr: open the source dataset, lay
    down one destination dataset per model output — same extent, voxel size and offset as the source, chunked by the write
    ROI, dtype from `out_specs[name]["dtype"]` (default: the source's, as the reference does), a channel axis when the output
    has more than one channel — and predict into them block by block.  Returns the destination arrays."""
    from . import zarrio
    src = zarrio.open_ds(source_path, source_dataset)
    dest_path = dest_path or source_path
    read, write, _ = roi_geometry(ndims, input_shape, output_shape, read_size, crop)
    names = list(output_ds) if output_ds is not None else [f"{source_dataset}_pred"]
    n_ch = list(out_channels) if out_channels is not None else [None] * len(names)
    dests = []
    if (rank or 0) == 0:
        for name, c in zip(names, n_ch):
            spec = dict(dtype=src.dtype, num_channels=c if c and c > 1 else None)
            spec.update((out_specs or {}).get(name, {}))
            dests.append(zarrio.prepare_ds(dest_path, name, src.shape, src.voxel_size, spec["dtype"], write_size=write,
                                           num_channels=spec["num_channels"], offset=src.offset))
    if world and world > 1:
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        elif (rank or 0) != 0:
            raise RuntimeError("raygun_amd.predict_zarr: world > 1 without an initialised process group — ranks other than 0 "
                               "cannot know when the destination datasets exist (initialise torch.distributed, or let rank "
                               "0 prepare them first and pass them to predict_volume)")
        dests = [zarrio.open_ds(dest_path, name, "a") for name in names]
    predict_volume(model, src, input_shape, output_shape, read_size, crop, ndims, scaleShift_input,
                   [d.dtype for d in dests], out_channels, rank, world, device, destinations=dests, precision=precision)
    return dests


def predict(render_config_path=None, rank=None, world=None):
    """`raygun-predict render.json` — mirror of raygun.predict.predict (reference: src/raygun/predict.py:16-192).

    Same render-config keys and defaults (:27-55), destination path / dataset naming (:57-68), ROI arithmetic (:73-108) and
    destination datasets (:111-125: one per output_ds entry over the source's ROI, voxel size and offset, dtype = the
    source's unless out_specs overrides it, chunked by the write ROI, deleted first) — through raygun_amd.zarrio instead of
    daisy.  The daisy task + worker pool (:130-167) is replaced by the static block partition: this process (or, under
    torch.distributed, every rank) runs raygun_amd.torch.predict.worker.worker over its share.  `launch_command`
    (LSF submission) is ignored; the neuroglancer view script (:171-189) is written as the reference does.
    Returns the list of destination arrays."""
    import logging
    import sys
    from . import zarrio
    from .torch.predict.worker import destination_names, read_render_config, shape_args, worker
    if render_config_path is None:
        render_config_path = sys.argv[1]
    logger = logging.getLogger(__name__)
    render_config, train_config = read_render_config(render_config_path)
    if rank is None and world is None:
        rank, _, world = parallel.init_from_env()
    source = zarrio.open_ds(render_config["source_path"], render_config["source_dataset"])
    dest_path, output_ds = destination_names(render_config)
    shapes = shape_args(render_config, train_config)
    _, write, _ = roi_geometry(render_config["ndims"], **shapes)
    out_specs = render_config["out_specs"]
    if (rank or 0) == 0:
        for dest_dataset in output_ds:
            these_specs = {"dtype": source.dtype, "num_channels": None}
            if out_specs is not None and dest_dataset in out_specs:
                these_specs.update(out_specs[dest_dataset])
            zarrio.prepare_ds(dest_path, dest_dataset, source.shape[-3:], source.voxel_size, these_specs["dtype"], write_size=write,
                              num_channels=these_specs["num_channels"], offset=source.offset, delete=True)
    if world and world > 1:
        if not torch.distributed.is_initialized():
            raise RuntimeError("raygun_amd.predict: world > 1 needs an initialised torch.distributed process group")
        torch.distributed.barrier()
    logger.info("Running blockwise prediction...")
    dests, n = worker(render_config_path, rank=rank, world=world)
    if world and world > 1:
        torch.distributed.barrier()
    if (rank or 0) == 0:
        source_path = render_config["source_path"]
        view_script = os.path.join(dest_path, f"view_{os.path.basename(source_path).rstrip('.n5').rstrip('.zarr')}.ng")
        for dest_dataset in output_ds:
            if not os.path.exists(view_script):
                with open(view_script, "w") as f:
                    f.write(f"neuroglancer -f {source_path} -d {render_config['source_dataset']} -f {dest_path} -d {dest_dataset} ")
            else:
                with open(view_script, "a") as f:
                    f.write(f"{dest_dataset} ")
    logger.info("Done.")
    return dests


class _Drain:
    """Host side of the block runner: one thread that downloads finished blocks through two pinned staging buffers on its own
    HIP stream and scatters them into the destination arrays (a multi-threaded torch CPU copy), so PCIe and the host memory
    traffic overlap the next block's forward pass.  At most two blocks are in flight."""

    def __init__(self, dev):
        import queue
        import threading
        self.dev, self.q, self.err = dev, queue.Queue(maxsize=2), None
        self.stream = torch.cuda.Stream(dev)
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1))
        self.stage, self.turn, self.stats = {}, 0, [0.0, 0.0, 0.0]
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def put(self, gpu, dest, where):
        if self.err is not None:
            raise self.err
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        gpu.record_stream(self.stream)
        t0 = time.perf_counter()
        self.q.put((gpu, ev, dest, where))
        self.stats[2] += time.perf_counter() - t0

    def _run(self):
        try:
            with torch.cuda.device(self.dev):
                while True:
                    item = self.q.get()
                    if item is None:
                        return
                    gpu, ev, dest, where = item
                    key = (gpu.dtype, gpu.numel(), self.turn & 1)
                    self.turn += 1
                    if key not in self.stage:
                        self.stage[key] = torch.empty(gpu.numel(), dtype=gpu.dtype).pin_memory()
                    host = self.stage[key].view(gpu.shape)
                    t0 = time.perf_counter()
                    with torch.cuda.stream(self.stream):
                        self.stream.wait_event(ev)
                        host.copy_(gpu, non_blocking=True)
                    self.stream.synchronize()
                    t1 = time.perf_counter()
                    self._scatter(dest, where, host.numpy())
                    self.stats[0] += t1 - t0; self.stats[1] += time.perf_counter() - t1
        except BaseException as e:  # surfaced by the next put() / close()
            self.err = e
            while True:
                if self.q.get() is None:
                    return

    def _scatter(self, dest, where, block):
        """dest[where] = block, split over a few threads (numpy releases the GIL inside the copies; a fresh destination is
        also page-faulted in here, which is what dominates a first pass).  Chunked on-disk destinations are written by this
        one thread (a chunk must not be read-modified-written from two threads)."""
        if not isinstance(dest, np.ndarray):
            key = tuple(slice(None) if w is Ellipsis else w for w in where) if dest.ndim == 4 else tuple(where[1:])
            comb = getattr(dest, "_combiner", None)
            if comb is not None:      # write-combining: the chunk goes to disk once, when its last block has arrived (blockio.py)
                lo = [0 if k == slice(None) else k.start for k in key]
                hi = [n if k == slice(None) else k.stop for k, n in zip(key, dest.shape)]
                comb.put(lo, hi, block)
            else:
                dest[key] = block
            return
        if block.ndim == 3:
            z = block.shape[0]
            parts = [(dest, (slice(where[1].start + i0, where[1].start + min(i0 + 16, z)), where[2], where[3]), block[i0:i0 + 16]) for i0 in range(0, z, 16)]
        else:
            parts = [(dest, (c,) + tuple(where[1:]), block[c]) for c in range(block.shape[0])]

        def one(p):
            p[0][p[1]] = p[2]
        list(self.pool.map(one, parts))

    def close(self):
        self.q.put(None)
        self.t.join()
        self.pool.shutdown()
        if os.environ.get("RAYGUN_AMD_DRAIN_STATS"):
            print(f"[drain] download {self.stats[0]:.3f} s, scatter {self.stats[1]:.3f} s, producer blocked {self.stats[2]:.3f} s", flush=True)
        if self.err is not None:
            raise self.err


if __name__ == "__main__":
    predict()
