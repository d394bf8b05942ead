#!/usr/bin/env python3
"""Convert a feature HDF5 file of the reference (keys "<video>.npy" -> [n_clips, n_patch, d] arrays,
utils/load_dataset.py:33-44) into the numpy layouts lstc_vad_amd.archive.FeatureArchive reads without h5py.

    python tools/h5_to_npz.py SHT_I3D_16PATCH.h5 sht_feats_dir          # directory of <key> files (memory-mappable)
    python tools/h5_to_npz.py SHT_I3D_16PATCH.h5 sht_feats.npz          # single archive

Run it where h5py is installed (it is not part of the MI355X image); the output travels to the GPU box."""
import os
import sys

import numpy as np


def main():
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    import h5py
    src, dst = sys.argv[1:]
    with h5py.File(src, "r") as h5:
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
