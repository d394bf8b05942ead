"""Debug probe (GPU box): layer-0 MultiHeadAttention backward of a full-width case, HIP vs torch-CPU f64 / f32 autograd of the
oracle's mha_forward on the SAME inputs (x = embedded batch, dz = the gradient the real step delivers, captured by a hook)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import test_hip_parity as T
from oracle import lstc_oracle as orc
from util import oracle_cfgs
from cases import FULL_CASES
name = sys.argv[1]
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 0
z, mode, skw, d, enc, head, nf, af, al = T._full_width_models(name)
_, ekw, _, _ = FULL_CASES[name]
ecfg, st = oracle_cfgs(mode, dict(ekw), dict(skw))
enc, head = enc.to('cuda').train(), head.to('cuda').train()
args = T._args(mode, skw)
nf, af, al = (torch.from_numpy(x).to('cuda') for x in (nf, af, al))
cap = {}
which = sys.argv[3] if len(sys.argv) > 3 else 'mha'
mha = enc.layer_stack[layer].slf_attn if which == 'mha' else enc.layer_stack[layer].pos_ffn
def fh(mod, inp, out):
    cap['x'] = inp[0].detach().clone()
    inp[0].register_hook(lambda g: cap.__setitem__('dx', g.detach().clone()))
    (out[0] if isinstance(out, tuple) else out).register_hook(lambda g: cap.__setitem__('dz', g.detach().clone()))
h = mha.register_forward_hook(fh)
enc_out, outputs, loss, sc = T._step(enc, head, mode, args, nf, af, al, d, True)
loss.backward()
h.remove()
hip = {k: p.grad.detach().cpu().double() for k, p in mha.named_parameters() if p.grad is not None}
if 'dx' in cap: hip['INPUT'] = cap['dx'].cpu().double()
x, dz = cap['x'].cpu(), cap['dz'].cpu()
print('x', tuple(x.shape), 'mean', float(x.mean()), 'std', float(x.std()), ' dz absmax', float(dz.abs().max()), 'rms', float(dz.pow(2).mean().sqrt()))
pre = f'layer_stack.{layer}.slf_attn.' if which == 'mha' else f'layer_stack.{layer}.pos_ffn.'
torch.set_num_threads(32)
res = {}
for dt in (torch.float64, torch.float32):
    P = {pre + k: v.detach().cpu().to(dt).requires_grad_(True) for k, v in mha.named_parameters()}
    xin = x.to(dt).requires_grad_(True)
    if which == 'mha':
        P[pre + 'relative_position_index'] = mha.relative_position_index.cpu()
        y, _ = orc.mha_forward(P, pre, xin, ecfg, True)
    else:
        y = orc.ffn_forward(P, pre, xin, ecfg, True)
    y.backward(dz.to(dt))
    P[pre + 'INPUT'] = xin
    res[dt] = {k[len(pre):]: v.grad.double() for k, v in P.items() if v.is_floating_point() and v.grad is not None}
for k in hip:
    r64 = res[torch.float64][k].reshape(hip[k].shape); m = float(r64.abs().max())
    print(f"{k:34s} max {m:.3e}  HIP-f64 {float((hip[k]-r64).abs().max())/m:.3e}  torchf32-f64 {float((res[torch.float32][k].reshape(r64.shape)-r64).abs().max())/m:.3e}")
