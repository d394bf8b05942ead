import sys, os, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import test_hip_parity as T
from cases import sample_index
name=sys.argv[1]
z, mode, skw, d, enc, head, nf, af, al = T._full_width_models(name)
enc, head = enc.to('cuda').train(), head.to('cuda').train()
args=T._args(mode, skw)
nf, af, al = (torch.from_numpy(x).to('cuda') for x in (nf, af, al))
enc_out, outputs, loss, sc = T._step(enc, head, mode, args, nf, af, al, d, True)
loss.backward()
for k,p in enc.named_parameters():
    if p.grad is None: continue
    g=p.grad.detach().reshape(-1)
    idx=sample_index(g.numel())
    gs=z[f'enc_gs.{k}']; gmax=float(z[f'enc_gmax.{k}'])
    e=(g[torch.from_numpy(idx).cuda()].cpu().numpy().astype(np.float64)-gs.astype(np.float64))
    print(f"{k:60s} max|err|/gmax {np.abs(e).max()/(gmax+1e-30):.3e}  gmax {gmax:.3e}")
    if 'table' in k and gmax>0:
        H=p.shape[1]
        order=np.argsort(-np.abs(e))[:12]
        for o in order:
            print("    row", idx[o]//H, "head", idx[o]%H, "err", e[o], "ref", gs[o])
