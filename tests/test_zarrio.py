"""raygun_amd.zarrio against the zarr v2 storage spec: chunk files laid down BY HAND here (independent of the writer) are read
back, and what the writer produces is checked at the byte level.  (The zarr package is not installed: parity unpinned.)"""
import json
import os
import zlib

import numpy as np
import pytest

from raygun_amd import zarrio


def _hand_made(root, sep=".", order="C", compressor=None, dtype="<u2"):
    """A (5, 7) array in (2, 3) chunks: edge chunks full-size, chunk (1, 2) missing (-> fill_value 9)."""
    os.makedirs(root)
    full = np.arange(35, dtype=dtype).reshape(5, 7)
    meta = {"zarr_format": 2, "shape": [5, 7], "chunks": [2, 3], "dtype": dtype, "compressor": compressor, "fill_value": 9,
            "order": order, "filters": None}
    if sep != ".":
        meta["dimension_separator"] = sep
    json.dump(meta, open(os.path.join(root, ".zarray"), "w"))
    json.dump({"resolution": [4, 4], "offset": [8, 0]}, open(os.path.join(root, ".zattrs"), "w"))
    for i in range(3):
        for j in range(3):
            if (i, j) == (1, 2):
                continue
            ch = np.full((2, 3), 77, dtype=dtype)  # padding of edge chunks is arbitrary
            blk = full[2 * i:2 * i + 2, 3 * j:3 * j + 3]
            ch[:blk.shape[0], :blk.shape[1]] = blk
            raw = ch.tobytes(order=order)
            if compressor:
                raw = zlib.compress(raw)
            p = os.path.join(root, f"{i}.{j}") if sep == "." else os.path.join(root, str(i), str(j))
            os.makedirs(os.path.dirname(p), exist_ok=True)
            open(p, "wb").write(raw)
    full = full.copy()
    full[2:4, 6:7] = 9
    return full


@pytest.mark.parametrize("sep,order,comp,dtype", [(".", "C", None, "<u2"), ("/", "C", {"id": "zlib", "level": 1}, "<u2"),
                                                  (".", "F", None, "<f4"), (".", "C", {"id": "zlib", "level": 5}, "|u1")])
def test_reads_hand_made_container(tmp_path, sep, order, comp, dtype):
    want = _hand_made(str(tmp_path / "a.zarr"), sep, order, comp, dtype)
    a = zarrio.open(str(tmp_path / "a.zarr"))
    assert a.shape == (5, 7) and a.chunks == (2, 3) and a.dtype == np.dtype(dtype) and a.attrs["resolution"] == [4, 4]
    assert np.array_equal(a[...], want) and np.array_equal(a[1:5, 2:7], want[1:5, 2:7])
    assert np.array_equal(a[3], want[3]) and np.array_equal(a[:, -1], want[:, -1]) and a[4, 6] == want[4, 6]
    with pytest.raises(PermissionError):
        a[0, 0] = 1


def test_writer_layout_and_partial_writes(tmp_path):
    g = zarrio.open(str(tmp_path / "v.zarr"), "a")
    a = g.create_dataset("volumes/raw", (5, 6, 7), chunks=(2, 3, 4), dtype=np.uint8, compressor=None, fill_value=3)
    assert json.load(open(tmp_path / "v.zarr" / "volumes" / ".zgroup")) == {"zarr_format": 2}
    ref = np.full((5, 6, 7), 3, np.uint8)
    assert np.array_equal(a[...], ref) and not os.path.exists(tmp_path / "v.zarr/volumes/raw/0.0.0")
    rng = np.random.default_rng(0)
    for lo, hi in (((0, 0, 0), (2, 3, 4)), ((1, 2, 3), (5, 6, 7)), ((4, 0, 5), (5, 1, 6))):
        blk = rng.integers(0, 255, [b - a_ for a_, b in zip(lo, hi)], dtype=np.uint8)
        sl = tuple(slice(a_, b) for a_, b in zip(lo, hi))
        a[sl] = blk; ref[sl] = blk
        assert np.array_equal(zarrio.open(str(tmp_path / "v.zarr"))["volumes/raw"][...], ref)
    a[2] = 7; ref[2] = 7
    a[:, 1, 2:4] = np.arange(2, dtype=np.uint8); ref[:, 1, 2:4] = np.arange(2)
    assert np.array_equal(a[...], ref)
    # byte-level: chunk (0,0,0) is the raw C-order block, 24 bytes
    raw = open(tmp_path / "v.zarr/volumes/raw/0.0.0", "rb").read()
    assert raw == ref[0:2, 0:3, 0:4].tobytes()
    # compressed sibling, `f[name] = array`, attrs
    g["seg"] = ref.astype(np.uint64)
    g["seg"].attrs["offset"] = [1, 2, 3]
    s = zarrio.open(str(tmp_path / "v.zarr"))["seg"]
    assert s.dtype == np.uint64 and np.array_equal(s[...], ref) and s.attrs["offset"] == [1, 2, 3]
    meta = json.load(open(tmp_path / "v.zarr/seg/.zarray"))
    # the default compressor is zarr's own default (Blosc / lz4) when the library's host-side codec is there, else zlib
    blosc = zarrio._native() is not None
    assert meta["compressor"] == ({"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0} if blosc else {"id": "zlib", "level": 1})
    assert meta["zarr_format"] == 2
    first = open(tmp_path / "v.zarr/seg" / ".".join(["0"] * 3), "rb").read()
    payload = zarrio.blosc_decode(first) if blosc else zlib.decompress(first)
    assert not blosc or (first[0] == 2 and first[3] == 8)           # a Blosc frame of 8-byte elements
    assert np.array_equal(np.frombuffer(payload, "<u8").reshape(meta["chunks"])[:5, :6, :7], ref[:meta["chunks"][0]])


def test_unsupported_codec_is_loud(tmp_path):
    root = str(tmp_path / "b.zarr")
    _hand_made(root)
    meta = json.load(open(os.path.join(root, ".zarray")))
    meta["compressor"] = {"id": "zstd", "level": 3}
    json.dump(meta, open(os.path.join(root, ".zarray"), "w"))
    with pytest.raises(NotImplementedError, match="zstd"):
        zarrio.open(root)


def _lz4_encode(data):
    """A small greedy LZ4 block ENCODER written from the block format description (test-side only): 4-byte hash matches,
    overlapping matches allowed, the last 5 bytes always literals (the format's end-of-block rule)."""
    data = bytes(data); n = len(data); out = bytearray(); table = {}; i = anchor = 0

    def emit(lit, mlen, off):
        tok_l, tok_m = min(len(lit), 15), (min(mlen - 4, 15) if mlen else 0)
        out.append((tok_l << 4) | tok_m)
        if len(lit) >= 15:
            r = len(lit) - 15
            while r >= 255:
                out.append(255); r -= 255
            out.append(r)
        out.extend(lit)
        if mlen:
            out.extend(off.to_bytes(2, "little"))
            if mlen - 4 >= 15:
                r = mlen - 4 - 15
                while r >= 255:
                    out.append(255); r -= 255
                out.append(r)

    while i + 4 <= n - 5:
        key = data[i:i + 4]; cand = table.get(key); table[key] = i
        if cand is not None and i - cand <= 65535:
            m = 4
            while i + m < n - 5 and data[cand + m] == data[i + m]:
                m += 1
            emit(data[anchor:i], m, i - cand)
            i += m; anchor = i
        else:
            i += 1
    emit(data[anchor:], 0, 0)
    return bytes(out)


def _blosc_frame(payload, typesize, blocksize, shuffle, codec="lz4", split=True):
    """A c-blosc 1.x frame laid down by hand from the published layout (header, block starts, per-block split streams)."""
    import zlib as _z
    nbytes = len(payload)
    flags = {"none": 0, "byte": 1, "bit": 4}[shuffle] | (0 if split else 0x10) | ({"lz4": 1, "zlib": 3}[codec] << 5)
    nblocks = (nbytes + blocksize - 1) // blocksize
    blocks = []
    for b in range(nblocks):
        blk = np.frombuffer(payload[b * blocksize:(b + 1) * blocksize], np.uint8)
        nel = len(blk) // typesize
        if shuffle == "byte" and typesize > 1:
            blk = np.concatenate([blk[:nel * typesize].reshape(nel, typesize).T.reshape(-1), blk[nel * typesize:]])
        elif shuffle == "bit":
            n8 = nel & ~7
            bits = np.unpackbits(blk[:n8 * typesize].reshape(n8, typesize), axis=1, bitorder="little")   # [element][bit]
            blk = np.concatenate([np.packbits(bits.T, axis=1, bitorder="little").reshape(-1), blk[n8 * typesize:]])
        blk = blk.tobytes()
        leftover = len(blk) != blocksize
        nsplit = typesize if (split and typesize <= 16 and blocksize // typesize >= 128 and not leftover) else 1
        ne = len(blk) // nsplit
        body = bytearray()
        for k in range(nsplit):
            part = blk[k * ne:(k + 1) * ne]
            comp = _lz4_encode(part) if codec == "lz4" else _z.compress(part, 1)
            if len(comp) >= len(part):
                comp = part                                   # stored stream: compressed size == uncompressed size
            body += len(comp).to_bytes(4, "little") + comp
        blocks.append(bytes(body))
    starts, pos = [], 16 + 4 * nblocks
    for blk in blocks:
        starts.append(pos); pos += len(blk)
    head = bytes([2, 1, flags, typesize]) + nbytes.to_bytes(4, "little") + blocksize.to_bytes(4, "little") + pos.to_bytes(4, "little")
    return head + b"".join(x.to_bytes(4, "little") for x in starts) + b"".join(blocks)


@pytest.mark.parametrize("dtype,shuffle,codec,split", [("<u2", "byte", "lz4", True), ("<f4", "byte", "lz4", False), ("u1", "none", "lz4", True),
                                                       ("<u8", "bit", "lz4", True), ("<f4", "byte", "zlib", True)])
def test_blosc_frames_laid_down_by_hand(tmp_path, dtype, shuffle, codec, split):
    """zarr's default compressor: Blosc frames (lz4 / zlib inside, byte / bit shuffle, split and unsplit blocks, a short last block,
    a stored stream) built by hand from the frame layout with the test's own LZ4 encoder, read back through blosc_decode, through
    both LZ4 decoders, and through an array whose .zarray names the blosc compressor.  (Unpinned against numcodecs: not installed.)"""
    rng = np.random.default_rng(5)
    ref = (np.cumsum(rng.integers(0, 3, size=(6, 40, 50)), axis=2) * (rng.random((6, 40, 50)) > 0.2)).astype(dtype)   # compressible, with runs
    ref[1] = rng.integers(0, 2 ** 8, size=(40, 50)).astype(dtype)                                                       # one noisy plane
    payload = ref.tobytes()
    ts = np.dtype(dtype).itemsize
    frame = _blosc_frame(payload, ts, 4096 * ts, shuffle, codec, split)
    assert len(frame) < len(payload)
    assert zarrio.blosc_decode(frame) == payload
    sample = payload[:5000]
    enc = _lz4_encode(sample)
    assert zarrio._lz4_block_py(enc, len(sample)) == sample and zarrio._lz4_block(enc, len(sample)) == sample
    assert zarrio._lz4_block_py(_lz4_encode(b"ab" * 300 + b"tail!"), 605) == b"ab" * 300 + b"tail!"      # overlapping match (offset 2)
    root = str(tmp_path / "bl.zarr")
    os.makedirs(root)
    json.dump({"zarr_format": 2, "shape": [6, 40, 50], "chunks": [6, 40, 50], "dtype": np.dtype(dtype).str, "order": "C", "fill_value": 0,
               "filters": None, "compressor": {"id": "blosc", "cname": codec, "clevel": 5, "shuffle": 1, "blocksize": 0}},
              open(os.path.join(root, ".zarray"), "w"))
    open(os.path.join(root, "0.0.0"), "wb").write(frame)
    a = zarrio.open(root)
    assert np.array_equal(a[...], ref)
    w = zarrio.open(root, mode="r+")
    w[0, 0, 0] = 7                                              # read-modify-write of a Blosc chunk: re-encoded by blosc_encode (zlib inside)
    ref2 = ref.copy(); ref2[0, 0, 0] = 7
    assert np.array_equal(zarrio.open(root)[...], ref2)
    stored = bytes([2, 1, 0x2, ts]) + len(payload).to_bytes(4, "little") * 2 + (16 + len(payload)).to_bytes(4, "little") + payload
    assert zarrio.blosc_decode(stored) == payload               # memcpy frame
    bad = bytearray(frame); bad[2] = (bad[2] & 0x1f) | (4 << 5)
    with pytest.raises(NotImplementedError, match="zstd"):
        zarrio.blosc_decode(bytes(bad))


def test_prepare_and_open_ds(tmp_path):
    a = zarrio.prepare_ds(str(tmp_path / "p.zarr"), "volumes/pred", (10, 12, 14), (4, 4, 4), np.float32, write_size=(5, 6, 7), num_channels=3,
                          offset=(40, 0, 8))
    assert a.shape == (3, 10, 12, 14) and a.chunks == (3, 5, 6, 7)
    a[:, 5:10, 0:6, 7:14] = 1.5
    b = zarrio.open_ds(str(tmp_path / "p.zarr"), "volumes/pred")
    assert b.voxel_size == (4, 4, 4) and b.offset == (40, 0, 8) and b[1, 7, 3, 9] == 1.5 and b[1, 0, 0, 0] == 0


def test_n5_dataset_laid_down_by_hand(tmp_path):
    """An N5 dataset written by hand from the N5 spec (attributes.json in fastest-axis-first order, big-endian blocks with the
    [mode, ndim, dims] header, gzip and raw compression, cropped edge blocks, a missing block) read through open_ds."""
    import gzip as _gz
    ref = (np.arange(5 * 6 * 7).reshape(5, 6, 7) * 3).astype(np.uint16)       # numpy order (z, y, x)
    for comp in ("gzip", "raw"):
        root = tmp_path / f"{comp}.n5"
        ds = root / "volumes" / "raw"
        os.makedirs(ds)
        json.dump({"dimensions": [7, 6, 5], "blockSize": [4, 3, 2], "dataType": "uint16", "compression": {"type": comp},
                   "resolution": [4, 8, 40], "offset": [0, 16, 80]}, open(ds / "attributes.json", "w"))
        for bz in range(3):
            for by in range(2):
                for bx in range(2):
                    if (bz, by, bx) == (2, 1, 1):
                        continue                                   # missing block = zeros
                    blk = ref[bz * 2:bz * 2 + 2, by * 3:by * 3 + 3, bx * 4:bx * 4 + 4]
                    dims = list(reversed(blk.shape))               # N5 order: x, y, z
                    payload = blk.astype(">u2").tobytes()
                    head = (0).to_bytes(2, "big") + (3).to_bytes(2, "big") + b"".join(d.to_bytes(4, "big") for d in dims)
                    os.makedirs(ds / str(bx) / str(by), exist_ok=True)
                    open(ds / str(bx) / str(by) / str(bz), "wb").write(head + (_gz.compress(payload) if comp == "gzip" else payload))
        a = zarrio.open_ds(str(root), "volumes/raw")
        want = ref.copy(); want[4:, 3:, 4:] = 0
        assert a.shape == (5, 6, 7) and a.chunks == (2, 3, 4) and a.voxel_size == (40, 8, 4) and a.offset == (80, 16, 0)
        got = a[...]
        assert got.dtype == np.uint16 and np.array_equal(got, want)
        assert np.array_equal(a[1:4, 2:5, 3:7], want[1:4, 2:5, 3:7])
        with pytest.raises(PermissionError):
            a[0, 0, 0] = 1


@pytest.mark.parametrize("dtype,cfg", [("<f4", {"clevel": 5, "shuffle": 1}), ("<u2", {"clevel": 1, "shuffle": 0, "blocksize": 1000}), ("u1", {}),
                                       ("<u8", {"clevel": 9, "shuffle": 1, "blocksize": 4096}), ("<f4", {"cname": "zlib", "clevel": 5, "shuffle": 1})])
def test_blosc_encode_frames(tmp_path, dtype, cfg):
    """Writing Blosc frames (what zarr / daisy.prepare_ds arrays expect): header fields as published for c-blosc 1.x, blocks addressed by
    the start table, round trip through blosc_decode for compressible, noisy (stored form) and empty chunks, a short last block, and
    through an array created with the blosc compressor.  (Unpinned against numcodecs: not installed here.)"""
    rng = np.random.default_rng(3)
    ts = np.dtype(dtype).itemsize
    smooth = (np.cumsum(rng.integers(0, 2, size=70001), dtype=np.int64) % 200).astype(dtype).tobytes()
    noisy = rng.integers(0, 256, size=5003, dtype=np.uint8).tobytes()
    for payload in (smooth, noisy, b"", smooth[:ts * 3]):
        f = zarrio.blosc_encode(payload, {**cfg, "_typesize": ts})
        assert f[0] == 2 and f[1] == 1 and f[3] == ts
        nbytes, bsize, cbytes = (int.from_bytes(f[4 + 4 * i:8 + 4 * i], "little") for i in range(3))
        want_codec = 1 if (str(cfg.get("cname", "lz4")).startswith("lz4") and len(payload) and zarrio._lz4_compress(b"probe") is not None) else 3
        assert nbytes == len(payload) and cbytes == len(f) and bsize % ts == 0 and (f[2] >> 5) == want_codec and f[2] & 0x10      # lz4 by the library, else zlib
        assert nbytes == 0 or bsize <= max(nbytes, ts), "header block size beyond the buffer (c-blosc clamps it; its decompressor checks)"
        assert zarrio.blosc_decode(f) == payload
        if payload is smooth:
            assert len(f) < len(payload) * (0.85 if want_codec == 1 else 0.5) and not f[2] & 0x2      # (1000-byte blocks without shuffle: LZ4 gains least)
            nblocks = (nbytes + bsize - 1) // bsize
            starts = [int.from_bytes(f[16 + 4 * b:20 + 4 * b], "little") for b in range(nblocks)]
            assert starts[0] == 16 + 4 * nblocks and starts == sorted(starts) and bool(f[2] & 0x1) == (ts > 1 and cfg.get("shuffle", 1) != 0)
        if payload is noisy:
            assert f[2] & 0x2 and len(f) == len(payload) + 16          # nothing gained: the stored form
    root = str(tmp_path / "b.zarr"); os.makedirs(root)
    json.dump({"zarr_format": 2, "shape": [9, 20, 30], "chunks": [4, 20, 16], "dtype": np.dtype(dtype).str, "order": "C", "fill_value": 0,
               "filters": None, "compressor": {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}},
              open(os.path.join(root, ".zarray"), "w"))
    a = zarrio.open(root, mode="r+")
    vol = (np.arange(9 * 20 * 30).reshape(9, 20, 30) % 251).astype(dtype)
    a[...] = vol
    assert np.array_equal(zarrio.open(root)[...], vol)
    raw = open(os.path.join(root, "0.0.0"), "rb").read()
    assert raw[0] == 2 and int.from_bytes(raw[4:8], "little") == 4 * 20 * 16 * ts       # a Blosc frame of one whole chunk
    try:   # pinned against the real codec wherever numcodecs is installed (it is not in this image)
        from numcodecs import Blosc
    except ImportError:
        return
    for payload in (smooth, noisy, smooth[:ts * 3]):
        assert bytes(Blosc().decode(zarrio.blosc_encode(payload, {**cfg, "_typesize": ts}))) == payload


@pytest.mark.parametrize("comp", ["default", None, {"type": "bzip2", "blockSize": 9}, {"type": "xz", "preset": 6}, {"type": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}])
def test_n5_write_round_trip_and_block_layout(tmp_path, comp):
    """Writing N5 (the reference's containers end in .n5 or .zarr): prepare_ds on a .n5 path lays down attributes.json in N5's (x, y, z) order with the
    resolution / offset attributes reversed like daisy's, region writes produce mode-0 blocks — header [0, ndim, dims fastest-first], big-endian payload,
    edge blocks cropped — that the hand-written reader of test_n5_dataset_laid_down_by_hand's format decodes, and open_ds reads the volume back."""
    import bz2, gzip, lzma
    root = str(tmp_path / "out.n5")
    a = zarrio.prepare_ds(root, "volumes/pred", (10, 12, 14), (4, 4, 8), np.float32, write_size=(4, 5, 6), num_channels=3, offset=(40, 0, 16), compressor=comp)
    meta = json.load(open(os.path.join(root, "volumes/pred", "attributes.json")))
    assert json.load(open(os.path.join(root, "attributes.json")))["n5"] == "2.0.0"
    assert meta["dimensions"] == [14, 12, 10, 3] and meta["blockSize"] == [6, 5, 4, 3] and meta["dataType"] == "float32"
    assert meta["resolution"] == [8, 4, 4] and meta["offset"] == [16, 0, 40]
    assert a.shape == (3, 10, 12, 14) and a.chunks == (3, 4, 5, 6) and a.voxel_size == (4, 4, 8) and a.offset == (40, 0, 16)
    rng = np.random.default_rng(2)
    vol = rng.standard_normal((3, 10, 12, 14)).astype(np.float32)
    a[:, 0:4] = vol[:, 0:4]
    a[:, 4:10, 0:7] = vol[:, 4:10, 0:7]                    # partial blocks: read-modify-write
    a[:, 4:10, 7:12] = vol[:, 4:10, 7:12]
    b = zarrio.open_ds(root, "volumes/pred")
    assert b.voxel_size == (4, 4, 8) and b.offset == (40, 0, 16) and b.dtype.kind == "f"
    assert np.array_equal(b[...], vol)
    with pytest.raises(PermissionError):
        b[0, 0, 0, 0] = 1.0
    # the last block along every spatial axis, decoded by hand: idx (c=0, z=2, y=2, x=2) -> path x/y/z/c, dims cropped to (3, 2, 2, 2)
    raw = open(os.path.join(root, "volumes/pred", "2", "2", "2", "0"), "rb").read()
    assert int.from_bytes(raw[0:2], "big") == 0 and int.from_bytes(raw[2:4], "big") == 4
    dims = [int.from_bytes(raw[4 + 4 * i:8 + 4 * i], "big") for i in range(4)]
    assert dims == [2, 2, 2, 3]
    kind = (meta["compression"] or {}).get("type")
    dec = {"raw": lambda x: x, "gzip": gzip.decompress, "bzip2": bz2.decompress, "xz": lzma.decompress, "blosc": zarrio.blosc_decode}[kind]
    blk = np.frombuffer(dec(raw[20:]), dtype=">f4").reshape(3, 2, 2, 2)
    assert np.array_equal(blk, vol[:, 8:10, 10:12, 12:14])
    assert kind == ("gzip" if comp == "default" else "raw" if comp is None else comp["type"])


def test_blockio_chunk_cache_and_write_combining(tmp_path):
    """raygun_amd.blockio — the on-disk block runner's host side: cached source chunks give the array's own slices; block writes that straddle chunks
    (write ROIs start at `context`) are assembled in memory and every chunk is written ONCE; chunks another rank's blocks touch go through the locked
    read-modify-write.  Either way the container ends up as if every block had been written directly."""
    from raygun_amd import blockio, predict as P
    rng = np.random.default_rng(2)
    shape = (70, 61, 64)
    src = zarrio.prepare_ds(str(tmp_path / "s.zarr"), "raw", shape, (4, 4, 4), np.uint8, write_size=(16, 16, 16))
    vol = rng.integers(0, 255, shape, dtype=np.uint8)
    src[...] = vol
    cache = blockio.ChunkCache(src, max_bytes=40 * 16 ** 3, threads=3)        # small: the LRU must evict and re-read
    for _ in range(25):
        lo = [int(rng.integers(0, n - 1)) for n in shape]
        hi = [int(rng.integers(a + 1, n + 1)) for a, n in zip(lo, shape)]
        sl = tuple(slice(a, b) for a, b in zip(lo, hi))
        assert np.array_equal(cache[sl], vol[sl])
    assert cache.decoded > 0
    cache.close()
    read, write, ctx = P.roi_geometry(3, [36] * 3, [20] * 3)
    blocks = P.enumerate_blocks(shape, read, write, ctx)
    data = {b: rng.integers(0, 255, (3,) + tuple(write), dtype=np.uint8) for b in blocks}
    want = np.zeros((3,) + shape, np.uint8)
    for (z, y, x), blk in data.items():
        want[:, z:z + 20, y:y + 20, x:x + 20] = blk
    roi = lambda b: ((0, b[0], b[1], b[2]), (3, b[0] + 20, b[1] + 20, b[2] + 20))      # noqa: E731
    for name, mine in (("alone", blocks), ("shared", blocks[:len(blocks) // 2])):
        dst = zarrio.prepare_ds(str(tmp_path / "d.zarr"), name, shape, (4, 4, 4), np.uint8, write_size=write, num_channels=3)
        others = [b for b in blocks if b not in mine]
        comb = blockio.ChunkCombiner(dst, [roi(b) for b in mine], [roi(b) for b in others], threads=3)
        for b in mine:
            comb.put(*roi(b), data[b])
        comb.close()
        for b in others:                                     # "the other rank": plain array writes (locked read-modify-write)
            dst[(slice(None),) + tuple(slice(o, o + 20) for o in b)] = data[b]
        assert np.array_equal(zarrio.open_ds(str(tmp_path / "d.zarr"), name)[...], want)
        if name == "alone":
            assert comb.stats["chunks_rmw"] == 0 and comb.stats["chunks_written_once"] == len(set(i for b in blocks for i in dst._chunk_ranges(*[list(v) for v in roi(b)])))
        else:
            assert comb.stats["chunks_rmw"] > 0 and comb.stats["chunks_written_once"] > 0
