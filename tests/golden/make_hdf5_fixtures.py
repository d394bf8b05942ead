"""Write tests/golden/hdf5/*.h5 with the REAL HDF5 library (libhdf5 1.10.x through ctypes; no h5py in this image).

    python tests/golden/make_hdf5_fixtures.py [--out DIR] [--lib /opt/conda/lib/libhdf5.so.103]

The files are data: tests/test_hdf5_host.py reads them with lstc_vad_amd.hdf5 (which shares no code with libhdf5) and
compares against hdf5_cases.content().  This script runs only where a libhdf5 exists; the committed files travel."""
import argparse
import ctypes as C
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from hdf5_cases import FILES, content  # noqa: E402

hid = C.c_int64
hsize = C.c_uint64


def load(path):
    if path is None:
        cands = sorted(glob.glob("/opt/conda/lib/libhdf5.so.*")) + sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*"))
        if not cands:
            raise SystemExit("no libhdf5 found; pass --lib")
        path = cands[0]
    lib = C.CDLL(path)
    lib.H5open()
    sig = {
        "H5Fcreate": (hid, [C.c_char_p, C.c_uint, hid, hid]), "H5Fclose": (C.c_int, [hid]),
        "H5Pcreate": (hid, [hid]), "H5Pclose": (C.c_int, [hid]),
        "H5Pset_libver_bounds": (C.c_int, [hid, C.c_int, C.c_int]), "H5Pset_userblock": (C.c_int, [hid, hsize]),
        "H5Pset_chunk": (C.c_int, [hid, C.c_int, C.POINTER(hsize)]), "H5Pset_deflate": (C.c_int, [hid, C.c_uint]),
        "H5Pset_shuffle": (C.c_int, [hid]), "H5Pset_fletcher32": (C.c_int, [hid]), "H5Pset_layout": (C.c_int, [hid, C.c_int]),
        "H5Pset_fill_value": (C.c_int, [hid, hid, C.c_void_p]), "H5Pset_alloc_time": (C.c_int, [hid, C.c_int]),
        "H5Pset_create_intermediate_group": (C.c_int, [hid, C.c_uint]),
        "H5Screate_simple": (hid, [C.c_int, C.POINTER(hsize), C.POINTER(hsize)]), "H5Screate": (hid, [C.c_int]),
        "H5Sclose": (C.c_int, [hid]),
        "H5Dcreate2": (hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), "H5Dclose": (C.c_int, [hid]),
        "H5Dwrite": (C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
        "H5Acreate2": (hid, [hid, C.c_char_p, hid, hid, hid, hid]), "H5Awrite": (C.c_int, [hid, hid, C.c_void_p]),
        "H5Aclose": (C.c_int, [hid]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib, path


def g(lib, name):
    return C.c_int64.in_dll(lib, name).value


def file_type(lib, dt: np.dtype):
    table = {"<f4": "H5T_IEEE_F32LE_g", ">f4": "H5T_IEEE_F32BE_g", "<f8": "H5T_IEEE_F64LE_g", ">f8": "H5T_IEEE_F64BE_g",
             "<i8": "H5T_STD_I64LE_g", "<i4": "H5T_STD_I32LE_g", ">i2": "H5T_STD_I16BE_g", "|u1": "H5T_STD_U8LE_g"}
    return g(lib, table[dt.str])


def mem_type(lib, dt: np.dtype):
    table = {"f4": "H5T_NATIVE_FLOAT_g", "f8": "H5T_NATIVE_DOUBLE_g", "i8": "H5T_NATIVE_INT64_g", "i4": "H5T_NATIVE_INT32_g",
             "i2": "H5T_NATIVE_INT16_g", "u1": "H5T_NATIVE_UINT8_g"}
    return g(lib, table[dt.str.lstrip("<>|=")])


def ok(rc, what):
    if rc < 0:
        raise RuntimeError(f"libhdf5: {what} failed ({rc})")
    return rc


def write_file(lib, path, spec):
    fapl = ok(lib.H5Pcreate(g(lib, "H5P_CLS_FILE_ACCESS_ID_g")), "H5Pcreate(fapl)")
    fcpl = ok(lib.H5Pcreate(g(lib, "H5P_CLS_FILE_CREATE_ID_g")), "H5Pcreate(fcpl)")
    if spec["libver"] == "latest":
        ok(lib.H5Pset_libver_bounds(fapl, 2, 2), "libver")          # H5F_LIBVER_V110 == LATEST in 1.10
    if spec.get("userblock"):
        ok(lib.H5Pset_userblock(fcpl, spec["userblock"]), "userblock")
    f = ok(lib.H5Fcreate(path.encode(), 2, fcpl, fapl), "H5Fcreate")   # H5F_ACC_TRUNC
    lcpl = ok(lib.H5Pcreate(g(lib, "H5P_CLS_LINK_CREATE_ID_g")), "H5Pcreate(lcpl)")
    ok(lib.H5Pset_create_intermediate_group(lcpl, 1), "intermediate groups")
    for key, shape, dtype, st in spec["datasets"]:
        dt = np.dtype(dtype)
        native = dt.newbyteorder("=")
        dcpl = ok(lib.H5Pcreate(g(lib, "H5P_CLS_DATASET_CREATE_ID_g")), "H5Pcreate(dcpl)")
        if "chunks" in st:
            ok(lib.H5Pset_chunk(dcpl, len(shape), (hsize * len(shape))(*st["chunks"])), "chunk")
        if st.get("shuffle"):
            ok(lib.H5Pset_shuffle(dcpl), "shuffle")
        if "deflate" in st:
            ok(lib.H5Pset_deflate(dcpl, st["deflate"]), "deflate")
        if st.get("fletcher32"):
            ok(lib.H5Pset_fletcher32(dcpl), "fletcher32")
        if st.get("layout") == "compact":
            ok(lib.H5Pset_layout(dcpl, 0), "layout")
        if st.get("alloc_early"):
            ok(lib.H5Pset_alloc_time(dcpl, 1), "alloc time")
        if "fill" in st:
            fv = np.array([st["fill"]], native)
            ok(lib.H5Pset_fill_value(dcpl, mem_type(lib, native), fv.ctypes.data), "fill value")
        if len(shape):
            space = ok(lib.H5Screate_simple(len(shape), (hsize * len(shape))(*shape), None), "dataspace")
        else:
            space = ok(lib.H5Screate(0), "scalar dataspace")
        d = ok(lib.H5Dcreate2(f, key.encode(), file_type(lib, dt), space, lcpl, dcpl, 0), f"H5Dcreate2({key})")
        if st.get("write", True) and int(np.prod(shape, dtype=np.int64)) > 0:
            a = np.ascontiguousarray(content(key, shape, dtype).astype(native))
            ok(lib.H5Dwrite(d, mem_type(lib, native), 0, 0, 0, a.ctypes.data), f"H5Dwrite({key})")
        if st.get("attr"):
            asp = ok(lib.H5Screate_simple(1, (hsize * 1)(3), None), "attr space")
            at = ok(lib.H5Acreate2(d, b"note", g(lib, "H5T_IEEE_F32LE_g"), asp, 0, 0), "H5Acreate2")
            v = np.array([1, 2, 3], np.float32)
            ok(lib.H5Awrite(at, g(lib, "H5T_NATIVE_FLOAT_g"), v.ctypes.data), "H5Awrite")
            lib.H5Aclose(at)
            lib.H5Sclose(asp)
        lib.H5Dclose(d)
        lib.H5Sclose(space)
        lib.H5Pclose(dcpl)
    lib.H5Pclose(lcpl)
    ok(lib.H5Fclose(f), "H5Fclose")
    lib.H5Pclose(fapl)
    lib.H5Pclose(fcpl)


def write_world(lib, out):
    """The SHT / UCF feature archives and the UCF ground truth of the synthetic test world (pipeline_world.py) as HDF5 files,
    laid out as the reference's files are: one dataset per video, key "<video>.npy", default (contiguous) storage."""
    import tempfile
    import pipeline_world as pw
    from lstc_vad_amd.models import Encoder, Regressor, Classifier
    with tempfile.TemporaryDirectory() as tmp:
        W = pw.build(tmp, Encoder, Regressor, Classifier)
        for tag in ("sht_feats", "ucf_feats", "ucf_gt"):
            z = np.load(W[tag])
            spec = dict(libver="earliest", datasets=[])
            arrays = {k: z[k] for k in z.files}
            path = os.path.join(out, f"world_{tag}.h5")
            _write_arrays(lib, path, arrays)
            print(os.path.basename(path), os.path.getsize(path), "bytes,", len(arrays), "datasets")


def _write_arrays(lib, path, arrays):
    f = ok(lib.H5Fcreate(path.encode(), 2, 0, 0), "H5Fcreate")
    for key, a in arrays.items():
        a = np.ascontiguousarray(a)
        space = ok(lib.H5Screate_simple(a.ndim, (hsize * a.ndim)(*a.shape), None), "dataspace")
        d = ok(lib.H5Dcreate2(f, key.encode(), file_type(lib, a.dtype), space, 0, 0, 0), f"H5Dcreate2({key})")
        ok(lib.H5Dwrite(d, mem_type(lib, a.dtype), 0, 0, 0, a.ctypes.data), f"H5Dwrite({key})")
        lib.H5Dclose(d)
        lib.H5Sclose(space)
    ok(lib.H5Fclose(f), "H5Fclose")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "hdf5"))
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    lib, path = load(a.lib)
    os.makedirs(a.out, exist_ok=True)
    for name, spec in FILES.items():
        write_file(lib, os.path.join(a.out, name), spec)
        print(name, os.path.getsize(os.path.join(a.out, name)), "bytes")
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    write_world(lib, a.out)
    print("written with", path)


if __name__ == "__main__":
    main()
