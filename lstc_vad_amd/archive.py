"""Feature containers: ``key -> ndarray`` with the access pattern the reference uses on its HDF5 files.

The reference opens ``h5py.File(path, 'r')`` and reads whole datasets with ``h5[key + '.npy'][:]``
(utils/load_dataset.py:33-44, :285-286, :409-411).  ``h5py`` is not part of this image, so three interchangeable
backends sit behind one class, picked from the path:

* ``*.h5`` / ``*.hdf5``  - HDF5 through ``lstc_vad_amd.hdf5`` (own reader of the file format: no h5py / libhdf5 needed);
* ``*.npz``              - a numpy archive whose member names are the HDF5 keys (``"01_001.npy"`` ...);
* a directory            - one ``.npy`` file per key (memory-mapped on read).

``write_archive`` produces the two numpy layouts (used by tests, the golden generator and ``tools/h5_to_npz.py``).
"""
from __future__ import annotations

import os

import numpy as np


class FeatureArchive:
    def __init__(self, path: str, mode: str = "r"):
        if mode != "r":
            raise ValueError("FeatureArchive is read-only")
        self.path = path
        self._h5 = self._npz = None
        if os.path.isdir(path):
            self.kind = "dir"
        elif path.endswith(".npz"):
            self.kind = "npz"
            self._npz = np.load(path, allow_pickle=False)
        elif os.path.exists(path):
            from . import hdf5
            self.kind = "h5"
            self._h5 = hdf5.File(path, "r")
        else:
            raise FileNotFoundError(path)

    def __getitem__(self, key: str) -> np.ndarray:
        if self.kind == "npz":
            return self._npz[key]
        if self.kind == "dir":
            f = os.path.join(self.path, key if key.endswith(".npy") else key + ".npy")
            return np.load(f, mmap_mode="r")
        return self._h5[key][:]

    def shape(self, key: str) -> tuple:
        """Shape of a member without reading its data where the container allows (HDF5 object header, .npy header of a directory
        archive; an .npz member is decompressed): what a rank needs of a video it does not own (pipeline, sharded passes)."""
        if self.kind == "h5":
            return tuple(self._h5[key].shape)
        return tuple(self[key].shape)

    def __contains__(self, key: str) -> bool:
        if self.kind == "npz":
            return key in self._npz.files
        if self.kind == "dir":
            return os.path.exists(os.path.join(self.path, key if key.endswith(".npy") else key + ".npy"))
        return key in self._h5

    def keys(self):
        if self.kind == "npz":
            return list(self._npz.files)
        if self.kind == "dir":
            return sorted(f for f in os.listdir(self.path) if f.endswith(".npy"))
        return list(self._h5.keys())

    def close(self):
        if self._npz is not None:
            self._npz.close()
        if self._h5 is not None:
            self._h5.close()
        self._npz = self._h5 = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def write_archive(path: str, arrays: dict) -> str:
    """Write ``{key: ndarray}`` as ``.npz`` (path ends with .npz) or as a directory of ``.npy`` files."""
    if path.endswith(".npz"):
        np.savez(path, **{k: np.asarray(v) for k, v in arrays.items()})
    else:
        os.makedirs(path, exist_ok=True)
        for k, v in arrays.items():
            np.save(os.path.join(path, k if k.endswith(".npy") else k + ".npy"), np.asarray(v))
    return path
