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

if len(sys.argv) > 2 and sys.argv[2] == "bf16":
    from lstc_vad_amd import functional as Fn
    z, mode, skw, d, enc, head, nf, af, al = T._full_width_models(name)
    enc, head = enc.to('cuda').train(), head.to('cuda').train()
    nf, af, al = (torch.from_numpy(x).to('cuda') for x in (nf, af, al))
    Fn.set_compute_dtype("bf16")
    enc_out, outputs, loss, sc = T._step(enc, head, mode, args, nf, af, al, d, True)
    loss.backward()
    Fn.set_compute_dtype("fp32")
    print("bf16: max|score diff|", float((outputs.reshape(z["outputs"].shape).cpu() - torch.from_numpy(z["outputs"])).abs().max()), "loss", float(sc[0]), float(z["scalars"][0]))
    for k, p in enc.named_parameters():
        if p.grad is None or p.numel() < 4096 or float(z[f"enc_gnorm.{k}"]) == 0.0:
            continue
        gs = torch.from_numpy(z[f"enc_gs.{k}"]).double()
        got = p.grad.detach().reshape(-1)[torch.from_numpy(sample_index(p.numel())).cuda()].cpu().double()
        cos = float((got * gs).sum() / (got.norm() * gs.norm() + 1e-30))
        print(f"   {k:50s} cos(sampled) {cos:.4f}  norm ratio {float(p.grad.double().norm()) / float(z[f'enc_gnorm.{k}']):.4f}")
