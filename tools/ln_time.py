"""Time the LayerNorm kernels (plain and pack-emitting) on one MI355X: python tools/ln_time.py [rows d] [--lib path.so ...].

Extra --lib arguments name alternative builds of csrc/rowops.hip (hipcc -shared ... -DLSTC_LN_BWD_VARIANT=n) whose
lstc_layernorm_bwd_drop_pack is timed and compared bit for bit with the in-tree library's."""
import ctypes as C
import sys

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from lstc_vad_amd import _lib
from lstc_vad_amd.functional import dev_ptr, stream_ptr

args = [a for a in sys.argv[1:]]
libs = []
while "--lib" in args:
    i = args.index("--lib")
    libs.append(args[i + 1])
    del args[i:i + 2]
rows, d = (int(args[0]), int(args[1])) if len(args) >= 2 else (100352, 2048)
lib = _lib.load()
dev = "cuda"
x = torch.randn(rows, d, device=dev)
dz = torch.randn(rows, d, device=dev)
gamma, beta = torch.randn(d, device=dev), torch.randn(d, device=dev)
y, dx, df = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
pk = torch.empty(int(lib.lstc_pack1_bytes(rows, d)), device=dev, dtype=torch.uint8)
seed = 0x1234567890ABCDEF


def timed(name, fn, gb, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        rc = fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:42s} {us:8.1f} us  {gb / us * 1e3:7.2f} TB/s (rc {rc})", flush=True)


B = rows * d * 4 / 1e9
s = stream_ptr()
timed("layernorm_fwd", lambda: lib.lstc_layernorm_fwd(dev_ptr(x), dev_ptr(gamma), dev_ptr(beta), dev_ptr(y), dev_ptr(mean), dev_ptr(rstd), rows, d, 1e-6, s), 2 * B)
timed("layernorm_fwd_pack", lambda: lib.lstc_layernorm_fwd_pack(dev_ptr(x), dev_ptr(gamma), dev_ptr(beta), dev_ptr(y), dev_ptr(mean), dev_ptr(rstd), rows, d, 1e-6, dev_ptr(pk), s), 2.5 * B)
timed("pack1", lambda: lib.lstc_pack1(dev_ptr(y), rows, d, d, 0, dev_ptr(pk), s), 1.5 * B)
timed("dropout_apply", lambda: lib.lstc_dropout_apply(dev_ptr(dz), dev_ptr(df), rows * d, 0.2, seed, s), 2 * B)
for npart in (512, 768, 1024):
    part = torch.empty(3, npart, d, device=dev)
    timed(f"layernorm_bwd n_partial={npart}", lambda: lib.lstc_layernorm_bwd(dev_ptr(dz), dev_ptr(x), dev_ptr(gamma), dev_ptr(mean), dev_ptr(rstd), dev_ptr(dx), dev_ptr(part), npart, rows, d, s), 3 * B)
    timed(f"layernorm_bwd_drop_pack n_partial={npart}", lambda: lib.lstc_layernorm_bwd_drop_pack(dev_ptr(dz), dev_ptr(x), dev_ptr(gamma), dev_ptr(mean), dev_ptr(rstd), dev_ptr(dx), dev_ptr(part), npart, rows, d, 0.2, seed, dev_ptr(pk), s), 3.5 * B)
ref_dx, ref_pk, ref_part = dx.clone(), pk.clone(), part.double().sum(1)
vp, i32, i64, f32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64
for path in libs:
    alt = C.CDLL(path)
    fn = alt.lstc_layernorm_bwd_drop_pack
    fn.argtypes, fn.restype = [vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, f32, u64, vp, vp], C.c_int
    for npart in (512, 768, 1024):
        part = torch.empty(3, npart, d, device=dev)
        dx.zero_(); pk.zero_()
        timed(f"{path} n_partial={npart}", lambda: fn(dev_ptr(dz), dev_ptr(x), dev_ptr(gamma), dev_ptr(mean), dev_ptr(rstd), dev_ptr(dx), dev_ptr(part), npart, rows, d, 0.2, seed, dev_ptr(pk), s), 3.5 * B)
    tiles = rows * d * 2
    print("   dx identical:", torch.equal(dx, ref_dx), " pack identical:", torch.equal(pk[:tiles], ref_pk[:tiles]),
          " partial sums max rel diff:", float(((part.double().sum(1) - ref_part).abs().max() / ref_part.abs().max())))
