"""Per-launch table of the GEMMs of one LTN-SHT training step (HIP events around every lstc_gemm): shape class, ms, TFLOP/s.
python tools/gemm_launch_table.py [fp32|bf16|f32x3] [config] [pairs per step, default 32]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
import bench
from lstc_vad_amd import functional as Fn
from lstc_vad_amd.engine import TrainStep
from lstc_vad_amd.models import Classifier, Encoder, Regressor
dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
cfg = sys.argv[2] if len(sys.argv) > 2 else "ltn_sht"
mode, ekw, skw, drops, _ = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
bs, pn, L, P, d = (int(sys.argv[3]) if len(sys.argv) > 3 else 32), 32, skw["part_len"], skw["n_patch"], ekw["d_model"]
args = Namespace(batch_size=bs, part_num=pn, part_len=L, n_patch=P, lambda_1=0.01, lambda_MIL=1.0, lambda_CE=0.8, lambda_BCE=1.0,
                 lambda_normal=0.2, lambda_abnormal=2.0, temporal_only=False, clip_grad=False)
torch.manual_seed(0)
enc = Encoder(n_layers=3, n_head=8, d_k=256, d_v=256, MHA_attn_dropout=drops[0], MHA_fc_dropout=drops[1], FFN_dropout=drops[2],
              weight_init=(mode != "LTN"), **ekw).to(dev).train()
head = (Classifier(d, drops[3]) if mode == "LTN" else Regressor(d, drops[3])).to(dev).train()
Fn.set_compute_dtype(dtype)
ts = TrainStep(args, mode, enc, head, 1e-7, 1e-5, 1e-3)
g = torch.Generator(device=dev).manual_seed(1)
T = pn * L
nf = 0.5 * torch.relu(torch.randn(bs, T, P, d, device=dev, generator=g)); af = 0.5 * torch.relu(torch.randn(bs, T, P, d, device=dev, generator=g))
al = torch.rand(bs, T, 1, device=dev, generator=g)
for _ in range(2):
    ts.step(nf, af, al)
# wrap _launch_gemm to record the descriptor next to the events
rows = []
orig = Fn._launch_gemm
def rec(dsc, flops):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(dsc, flops); e1.record()
    rows.append((dsc.M, dsc.N, dsc.K, dsc.transA, dsc.transB, dsc.flags, dsc.split_k, max(dsc.batch, 1), dsc.dtype, flops, e0, e1))
Fn._launch_gemm = rec
ob = Fn.gemm_batched
def recb(a, b, c, M, N, K, lda, ldb, ldc, ta, tb, batch, *rest, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = ob(a, b, c, M, N, K, lda, ldb, ldc, ta, tb, batch, *rest, **kw); e1.record()
    rows.append((M, N, K, int(ta), int(tb), 0, 1, batch, -1, 2.0 * M * N * K * batch, e0, e1))
    return r
Fn.gemm_batched = recb
ts.step(nf, af, al)
torch.cuda.synchronize()
agg = collections.OrderedDict()
tot_ms = tot_fl = 0.0
for (M, N, K, ta, tb, fl, sp, ba, dt, flops, e0, e1) in rows:
    ms = e0.elapsed_time(e1)
    key = (M, N, K, ta, tb, fl, sp, ba)
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += flops
    tot_ms += ms; tot_fl += flops
print(f"{dtype} {cfg}: {len(rows)} launches, {tot_ms:.2f} ms, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  M={key[0]:7d} N={key[1]:5d} K={key[2]:7d} tA={key[3]} tB={key[4]} flags={key[5]:3d} split={key[6]} batch={key[7]:2d}  x{n:2d}  {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
