#!/usr/bin/env python3
"""Convert a feature HDF5 file of the reference (keys "<video>.npy" -> [n_clips, n_patch, d] arrays,
utils/load_dataset.py:33-44) into the numpy layouts lstc_vad_amd.archive.FeatureArchive also reads.  Optional: the archive
class opens HDF5 files directly (lstc_vad_amd/hdf5.py); a directory of .npy files is the memory-mappable alternative.

    python tools/h5_to_npz.py SHT_I3D_16PATCH.h5 sht_feats_dir          # directory of <key> files (memory-mappable)
    python tools/h5_to_npz.py SHT_I3D_16PATCH.h5 sht_feats.npz          # single archive

Needs neither h5py nor libhdf5: the file is read with lstc_vad_amd.hdf5."""
import os
import sys

import numpy as np


def main():
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from lstc_vad_amd import hdf5
    src, dst = sys.argv[1:]
    with hdf5.File(src, "r") as h5:
        keys = list(h5.keys())
        if dst.endswith(".npz"):
            np.savez(dst, **{k: h5[k][:] for k in keys})
        else:
            os.makedirs(dst, exist_ok=True)
            for k in keys:
                np.save(os.path.join(dst, k if k.endswith(".npy") else k + ".npy"), h5[k][:])
    print(f"{len(keys)} datasets -> {dst}")


if __name__ == "__main__":
    main()
