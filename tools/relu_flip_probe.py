"""Debug probe (GPU box): ReLU decisions of the HIP FFN hidden vs f64 pre-activations on the same input, per layer."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import test_hip_parity as T
from lstc_vad_amd import functional as Fn
name = sys.argv[1]
z, mode, skw, d, enc, head, nf, af, al = T._full_width_models(name)
enc, head = enc.to('cuda').train(), head.to('cuda').train()
args = T._args(mode, skw)
nf, af, al = (torch.from_numpy(x).to('cuda') for x in (nf, af, al))
cap = {}
hs = [l.pos_ffn.register_forward_hook(lambda m, i, o, k=k: cap.__setitem__(k, i[0].detach().clone())) for k, l in enumerate(enc.layer_stack)]
T._step(enc, head, mode, args, nf, af, al, d, False)
for k, l in enumerate(enc.layer_stack):
    x2 = cap[k].view(-1, d)
    w1, b1 = l.pos_ffn.w_1.weight.detach(), l.pos_ffn.w_1.bias.detach()
    h = Fn.gemm(x2, w1, trans_b=True, bias=b1, relu=True)
    pre64 = x2.double() @ w1.double().t() + b1.double()            # f64 on the GPU through ATen: probe only
    pre32 = x2 @ w1.t() + b1
    flip = (h > 0) != (pre64 > 0)
    flip32 = (pre32 > 0) != (pre64 > 0)
    idx = flip.nonzero()
    print(f"layer {k}: elements {h.numel()}  |pre64| < 1e-6: {int((pre64.abs() < 1e-6).sum())}  < 1e-7: {int((pre64.abs() < 1e-7).sum())}  HIP flips {int(flip.sum())}  rocBLAS-f32 flips {int(flip32.sum())}"
          f"  max|h - relu(pre64)| {float((h.double() - pre64.clamp_min(0)).abs().max()):.2e}  rocBLAS {float((pre32.double() - pre64).abs().max()):.2e}")
    for t, j in idx.tolist()[:8]:
        print("    flip at token", t, "unit", j, "pre64", float(pre64[t, j]), "hip h", float(h[t, j]))
