"""Contents of the HDF5 fixtures (tests/golden/hdf5/*.h5): shared by the generator (make_hdf5_fixtures.py, which writes them
with the real libhdf5) and the reader test (tests/test_hdf5_host.py, which regenerates the expected arrays from this table).

Entry: (key, shape, dtype string, storage) - storage is a dict of libhdf5 dataset-creation options."""
import zlib

import numpy as np


def content(key: str, shape, dtype: str) -> np.ndarray:
    """Deterministic values for dataset ``key``: seeded by the name, so the files need no sidecar."""
    rs = np.random.RandomState(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    dt = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64))
    if dt.kind == "f":
        a = rs.standard_normal(n).astype(dt.newbyteorder("="))
    else:
        info = np.iinfo(dt)
        a = rs.randint(max(info.min, -1000), min(info.max, 1000) + 1, size=n).astype(dt.newbyteorder("="))
    return a.reshape(shape)


def _videos(n, prefix="%02d_%04d.npy"):
    out = []
    for i in range(n):
        out.append((prefix % (i // 30 + 1, i), (1 + i % 4, 2, 3), "<f4", {}))
    return out


FILES = {
    # what h5py writes by default: superblock 0, symbol-table groups (300 names: a two-level group B-tree), contiguous data
    "default_many.h5": dict(libver="earliest", datasets=_videos(300) + [
        ("grp/sub/x", (5, 4), "<f4", {}),
        ("scalar", (), "<f8", {}),
        ("gt_frames.npy", (37,), "<i8", {}),
        ("bytes", (11, 3), "|u1", {}),
        ("doubles", (6, 2), "<f8", {}),
        ("big_endian", (4, 5), ">f4", {}),
        ("be_int", (9,), ">i2", {}),
        ("compact", (3, 4), "<f4", {"layout": "compact"}),
        ("never_written", (4, 3), "<f4", {"fill": 7.5, "write": False}),
        ("zero_rows", (0, 3), "<f4", {}),
        ("with_attribute", (2, 2), "<f4", {"attr": True}),
    ]),
    # chunked storage with the v1 B-tree chunk index and the standard filters
    "chunked.h5": dict(libver="earliest", datasets=[
        ("plain", (10, 7, 5), "<f4", {"chunks": (4, 3, 5)}),
        ("gzip", (33, 16, 8), "<f4", {"chunks": (8, 16, 8), "deflate": 4}),
        ("gzip_shuffle", (21, 9), "<f8", {"chunks": (5, 4), "deflate": 6, "shuffle": True}),
        ("shuffle_fletcher", (12, 6), "<i4", {"chunks": (5, 6), "shuffle": True, "fletcher32": True}),
        ("many_chunks", (150, 4), "<f4", {"chunks": (1, 4)}),
        ("partly_allocated", (8, 6), "<f4", {"chunks": (4, 3), "fill": -1.0, "write": False}),
        ("feat_like.npy", (24, 16, 64), "<f4", {"chunks": (1, 16, 64), "deflate": 1}),
    ]),
    # libver='latest': superblock 3, version-2 object headers, link messages (compact and dense), layout version 4
    "latest.h5": dict(libver="latest", datasets=[
        ("a.npy", (3, 2, 3), "<f4", {}),
        ("b.npy", (2, 2, 3), "<f4", {}),
        ("single_chunk", (6, 5), "<f4", {"chunks": (6, 5)}),
        ("single_chunk_gzip", (6, 5), "<f4", {"chunks": (6, 5), "deflate": 3}),
        ("implicit", (9, 4), "<f4", {"chunks": (2, 4), "alloc_early": True}),
        ("fixed_array", (10, 7), "<f4", {"chunks": (3, 4)}),
        ("fixed_array_gzip", (10, 7), "<f8", {"chunks": (3, 4), "deflate": 2, "shuffle": True}),
        ("fixed_array_paged", (1100, 2), "<i4", {"chunks": (1, 2)}),
    ] + [("many/%03d.npy" % i, (2, 3), "<f4", {}) for i in range(40)]),
    # a 512-byte user block in front of the superblock
    "userblock.h5": dict(libver="earliest", userblock=512, datasets=[("x.npy", (4, 4), "<f4", {})]),
}
