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
    assert meta["compressor"] == {"id": "zlib", "level": 1} and meta["zarr_format"] == 2
    first = open(tmp_path / "v.zarr/seg" / ".".join(["0"] * 3), "rb").read()
    assert np.array_equal(np.frombuffer(zlib.decompress(first), "<u8").reshape(meta["chunks"])[:5, :6, :7], ref[:meta["chunks"][0]])


def test_unsupported_codec_is_loud(tmp_path):
    root = str(tmp_path / "b.zarr")
    _hand_made(root)
    meta = json.load(open(os.path.join(root, ".zarray")))
    meta["compressor"] = {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}
    json.dump(meta, open(os.path.join(root, ".zarray"), "w"))
    with pytest.raises(NotImplementedError, match="blosc"):
        zarrio.open(root)


def test_prepare_and_open_ds(tmp_path):
    a = zarrio.prepare_ds(str(tmp_path / "p.zarr"), "volumes/pred", (10, 12, 14), (4, 4, 4), np.float32, write_size=(5, 6, 7), num_channels=3,
                          offset=(40, 0, 8))
    assert a.shape == (3, 10, 12, 14) and a.chunks == (3, 5, 6, 7)
    a[:, 5:10, 0:6, 7:14] = 1.5
    b = zarrio.open_ds(str(tmp_path / "p.zarr"), "volumes/pred")
    assert b.voxel_size == (4, 4, 4) and b.offset == (40, 0, 8) and b[1, 7, 3, 9] == 1.5 and b[1, 0, 0, 0] == 0
