"""lstc_vad_amd.hdf5 (dependency-free HDF5 reader) against files written by the real libhdf5 1.10.6
(tests/golden/hdf5/*.h5, recipe tests/golden/make_hdf5_fixtures.py).  Replaces the reference's h5py reads
(utils/load_dataset.py:33-46, :113-119, :409-411, :466-499)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from hdf5_cases import FILES, content  # noqa: E402

from lstc_vad_amd import hdf5  # noqa: E402

H5DIR = os.path.join(HERE, "golden", "hdf5")


@pytest.mark.parametrize("fname", list(FILES))
def test_every_dataset_reads_back_bit_for_bit(fname):
    spec = FILES[fname]
    with hdf5.File(os.path.join(H5DIR, fname)) as f:
        for key, shape, dtype, st in spec["datasets"]:
            d = f[key]
            assert isinstance(d, hdf5.Dataset), key
            assert d.shape == tuple(shape) and d.dtype == np.dtype(dtype), (key, d.shape, d.dtype)
            got = d[...]
            if st.get("write", True):
                want = content(key, shape, dtype)
            else:
                want = np.full(shape, st["fill"], np.dtype(dtype))
            assert got.shape == tuple(shape)
            assert np.array_equal(got, want), key
            assert got.flags.owndata or got.base is None or not isinstance(got.base, memoryview)
        top = sorted({k.split("/")[0] for k, *_ in spec["datasets"]})
        assert f.keys() == top and len(f) == len(top)
        assert all(k in f for k in top) and "no_such_key.npy" not in f
        with pytest.raises(KeyError):
            f["no_such_key.npy"]


def test_h5py_style_access_patterns_of_the_reference_loaders():
    """``h5[key + '.npy'][:]`` (whole read), then ``[:, :n_patch, :]`` slices, ``np.array(dataset)``, nested paths, iteration."""
    with hdf5.File(os.path.join(H5DIR, "default_many.h5"), "r") as f:
        key = "03_0075"
        a = f[key + ".npy"][:]
        want = content(key + ".npy", (1 + 75 % 4, 2, 3), "<f4")
        assert a.dtype == np.float32 and np.array_equal(a, want)
        assert np.array_equal(f[key + ".npy"][:, :1, :], want[:, :1, :])
        assert np.array_equal(np.array(f[key + ".npy"]), want)
        assert np.array_equal(f[key + ".npy"][1:, 0, ::2], want[1:, 0, ::2])
        v = f[key + ".npy"].view()                              # zero-copy window on the file map
        assert v is not None and not v.flags.writeable and np.array_equal(v, want)
        assert isinstance(f["grp"], hdf5.Group) and f["grp"].keys() == ["sub"]
        assert np.array_equal(f["grp"]["sub"]["x"][:], f["/grp/sub/x"][:])
        assert f["scalar"].shape == () and f["scalar"][()] == content("scalar", (), "<f8")
        assert f["gt_frames.npy"][:].dtype == np.dtype("<i8")
        assert f["big_endian"][:].astype(np.float32).dtype == np.float32
        names = [k for k in f if k.endswith(".npy") and k[:2].isdigit()]
        assert len(names) == 300 and names == sorted(names)
        del v
    with pytest.raises(ValueError):
        hdf5.File(os.path.join(H5DIR, "default_many.h5"), "w")


def test_chunked_layouts_and_filters_are_the_ones_claimed():
    """Guards the fixture itself: the files exercise the structures the reader claims (else the test above proves less)."""
    with hdf5.File(os.path.join(H5DIR, "chunked.h5")) as f:
        assert f["plain"]._layout[0] == "chunked_v1" and f["plain"]._filters == []
        assert [fid for fid, _ in f["gzip_shuffle"]._filters] == [2, 1]
        assert [fid for fid, _ in f["shuffle_fletcher"]._filters] == [2, 3]
        assert len(list(f["many_chunks"]._chunks())) == 150
        assert list(f["partly_allocated"]._chunks()) == []
    with hdf5.File(os.path.join(H5DIR, "latest.h5")) as f:
        kinds = {k: f[k]._layout[0] for k in ("a.npy", "single_chunk", "single_chunk_gzip", "implicit", "fixed_array",
                                               "fixed_array_gzip", "fixed_array_paged")}
        assert kinds == {"a.npy": "contiguous", "single_chunk": "single", "single_chunk_gzip": "single", "implicit": "implicit",
                         "fixed_array": "fixed_array", "fixed_array_gzip": "fixed_array", "fixed_array_paged": "fixed_array"}
        assert len(f["many"]) == 40
        raw = open(os.path.join(H5DIR, "latest.h5"), "rb").read()
        assert raw[8] == 3 and b"OHDR" in raw and b"FRHP" in raw and b"BTHD" in raw and b"FAHD" in raw
    raw = open(os.path.join(H5DIR, "default_many.h5"), "rb").read()
    assert raw[8] == 0 and raw.count(b"SNOD") > 32 and raw.count(b"TREE") >= 3      # two-level group B-tree
    raw = open(os.path.join(H5DIR, "userblock.h5"), "rb").read()
    assert raw[:8] != b"\x89HDF\r\n\x1a\n" and raw[512:520] == b"\x89HDF\r\n\x1a\n"


def test_not_hdf5_and_unsupported_features_fail_loudly(tmp_path):
    p = tmp_path / "junk.h5"
    p.write_bytes(b"not an hdf5 file" * 100)
    with pytest.raises(hdf5.HDF5Error):
        hdf5.File(str(p))
    p.write_bytes(b"")
    with pytest.raises(OSError):
        hdf5.File(str(p))


def test_feature_archive_reads_hdf5_without_h5py():
    """FeatureArchive is what the dataset classes open (load_dataset.py): *.h5 now goes through lstc_vad_amd.hdf5."""
    from lstc_vad_amd.archive import FeatureArchive
    with FeatureArchive(os.path.join(H5DIR, "chunked.h5")) as arc:
        assert arc.kind == "h5" and "feat_like.npy" in arc and "missing.npy" not in arc
        a = arc["feat_like.npy"]
        assert isinstance(a, np.ndarray) and np.array_equal(a, content("feat_like.npy", (24, 16, 64), "<f4"))
        assert "gzip" in arc.keys()


@pytest.mark.skipif(not os.path.exists("/opt/conda/lib/libhdf5.so.103"), reason="no libhdf5 here: fixtures cannot be regenerated")
def test_fixture_recipe_reproduces_the_committed_files(tmp_path):
    """The committed .h5 files are what tests/golden/make_hdf5_fixtures.py writes with libhdf5 (same datasets, same values;
    bytes differ only in the modification time stamps of version-1 object headers, so contents are compared)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_hdf5_fixtures.py"), "--out", str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for fname, spec in FILES.items():
        with hdf5.File(str(tmp_path / fname)) as new, hdf5.File(os.path.join(H5DIR, fname)) as old:
            for key, *_ in spec["datasets"]:
                assert np.array_equal(new[key][...], old[key][...]), (fname, key)
