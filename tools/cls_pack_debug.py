import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import test_act16_gpu as T
from lstc_vad_amd import functional as Fn
name = sys.argv[1] if len(sys.argv) > 1 else "ltn_ubnormal_full_256"
res = {}
for cp in (1, 0):
    Fn._CLS_PACK = bool(cp)
    z, a16, c16, _, _ = T._run_step(name, "bf16")
    res[cp] = a16[0]
    print("cls_pack", cp, c16)
Fn._CLS_PACK = True
_, a32, c32, _, _ = T._run_step(name, "fp32")
o32, s32, g32 = a32[0]
for cp in (1, 0):
    o, s_, g = res[cp]
    print(f"== cls_pack={cp}: scores vs f32act {float((o-o32).abs().max()):.3e} vs ref {float((o.reshape(z['outputs'].shape).cpu()-torch.as_tensor(z['outputs'])).abs().max()):.3e}")
    for k in g32:
        if g32[k].numel() < 4096 or float(g32[k].norm()) == 0: continue
        a, b = g[k].double().reshape(-1), g32[k].double().reshape(-1)
        cos = float((a*b).sum()/(a.norm()*b.norm()+1e-30))
        if cos < 0.995: print(f"   {k:50s} cos {cos:.4f} norm ratio {float(a.norm()/b.norm()):.4f} |g| {float(b.norm()):.3e}")
hp = T._hp()
zz, mode, skw, d, enc, head, nf, af, al = hp._full_width_models(name)
args = hp._args(mode, skw)
pn = args.part_num
for cp in (1, 0):
    for col in (0, 1):
        sc = torch.softmax(res[cp][0], -1)[:, col].reshape(-1, pn)
        s3 = torch.softmax(o32, -1)[:, col].reshape(-1, pn)
        top2 = s3.topk(2, 1).values
        print("cls_pack", cp, "col", col, "bags", sc.shape[0], "argmax agree", int((sc.argmax(1) == s3.argmax(1)).sum()), "min top1-top2 gap of the f32act run", float((top2[:,0]-top2[:,1]).min()))
