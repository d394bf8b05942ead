"""Debug probe (GPU box): gradient arriving at every block output of a full-width case - HIP step vs the oracle in f64 (CPU)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import test_hip_parity as T
from oracle import lstc_oracle as orc
from util import oracle_cfgs
from cases import FULL_CASES
name = sys.argv[1]
cls_only = len(sys.argv) > 2 and sys.argv[2] == 'cls'
z, mode, skw, d, enc, head, nf, af, al = T._full_width_models(name)
_, ekw, _, seed = FULL_CASES[name]
ecfg, st = oracle_cfgs(mode, dict(ekw), dict(skw))
enc_P = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in enc.state_dict().items()}
head_P = {k: v.detach().clone().double() for k, v in head.state_dict().items()}
enc, head = enc.to('cuda').train(), head.to('cuda').train()
args = T._args(mode, skw)
nfd, afd, ald = (torch.from_numpy(x).to('cuda') for x in (nf, af, al))
hip = {}
def mk(tag):
    def fh(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        hip[tag + '.out'] = o.detach().cpu().double()
        o.register_hook(lambda g: hip.__setitem__(tag + '.dz', g.detach().cpu().double()))
    return fh
hs = []
for i, l in enumerate(enc.layer_stack[:-1] if cls_only else enc.layer_stack):
    hs.append(l.slf_attn.register_forward_hook(mk(f'mha{i}')))
    hs.append(l.pos_ffn.register_forward_hook(mk(f'ffn{i}')))
enc_out, outputs, loss, sc = T._step(enc, head, mode, args, nfd, afd, ald, d, cls_only)
loss.backward()
torch.cuda.synchronize()
# ---- f64 oracle with hooks on the same tensors
torch.set_default_dtype(torch.float64)
torch.Tensor.float = lambda self: self.double()
torch.set_num_threads(32)
ref = {}
_mha, _ffn = orc.mha_forward, orc.ffn_forward
def mha2(P, pre, x, *a, **k):
    y, att = _mha(P, pre, x, *a, **k)
    tag = 'mha' + pre.split('.')[1]
    ref[tag + '.out'] = y.detach()
    if y.requires_grad:
        y.register_hook(lambda g: ref.__setitem__(tag + '.dz', g.detach()))
    return y, att
def ffn2(P, pre, x, *a, **k):
    y = _ffn(P, pre, x, *a, **k)
    tag = 'ffn' + pre.split('.')[1]
    ref[tag + '.out'] = y.detach()
    if y.requires_grad:
        y.register_hook(lambda g: ref.__setitem__(tag + '.dz', g.detach()))
    return y
orc.mha_forward, orc.ffn_forward = mha2, ffn2
enc_S = {k: torch.zeros_like(v) for k, v in enc_P.items() if v.is_floating_point()}
head_S = {k: torch.zeros_like(v) for k, v in head_P.items()}
out, *_ = orc.train_step(enc_P, head_P, enc_S, head_S, ecfg, st, torch.from_numpy(nf).double(), torch.from_numpy(af).double(), torch.from_numpy(al).double())
for k in sorted(hip):
    if k not in ref:
        continue
    a, b = hip[k], ref[k]
    if a.shape != b.shape:
        print(k, 'shape', tuple(a.shape), tuple(b.shape)); continue
    dlt = a - b
    print(f"{k:10s} absmax {float(b.abs().max()):.3e} rms {float(b.pow(2).mean().sqrt()):.3e}  err max {float(dlt.abs().max()):.3e} rms {float(dlt.pow(2).mean().sqrt()):.3e} mean {float(dlt.mean()):.3e}"
          f"  | per-row-sum err rms {float(dlt.sum(-1).pow(2).mean().sqrt()):.3e}  ref row-sum rms {float(b.sum(-1).pow(2).mean().sqrt()):.3e}")
