"""Where does the f32x3 training step differ from the exact-f32 one?  Records every GEMM output of one step in both modes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
from lstc_vad_amd import functional as Fn
from lstc_vad_amd.losses import training_loss
from lstc_vad_amd.models import Classifier, Encoder
dev = torch.device("cuda", 0)
torch.manual_seed(3)
ekw = dict(n_layers=3, n_head=8, d_k=256, d_v=256, d_model=2048, d_inner=4096, MHA_layerNorm=True, FFN_layerNorm=True,
           relative_pe=True, window_size=4, window_depth=3)
enc = Encoder(MHA_attn_dropout=0.0, MHA_fc_dropout=0.0, FFN_dropout=0.0, weight_init=True, **ekw).to(dev).train()
head = Classifier(2048, 0.0).to(dev).train()
bs, pn, L, P, d = 2, 6, 3, 16, 2048
nf = (0.5 * torch.relu(torch.randn(bs, pn * L, P, d))).to(dev); af = (0.5 * torch.relu(torch.randn(bs, pn * L, P, d))).to(dev)
u = torch.rand(bs, pn * L, 1); al = torch.where(u > 0.65, u, torch.zeros_like(u)).to(dev)
args = Namespace(batch_size=bs, part_num=pn, part_len=L, n_patch=P, lambda_1=0.01, lambda_MIL=1.0, lambda_CE=0.8, lambda_BCE=1.0,
                 lambda_normal=0.2, lambda_abnormal=2.0, temporal_only=False, clip_grad=False)
orig = Fn.gemm
rec = {}
def run(mode):
    log = []
    def g(a, b, **kw):
        out = orig(a, b, **kw)
        log.append((kw.get("relu", False), kw.get("relu_mask") is not None, kw.get("trans_a", False), out.detach().clone()))
        return out
    Fn.gemm = g
    Fn.set_compute_dtype(mode)
    enc.zero_grad(set_to_none=True); head.zero_grad(set_to_none=True)
    cls = enc.forward_cls(nf.reshape(bs * pn, L * P, d), af.reshape(bs * pn, L * P, d))
    loss, sc = training_loss(args, "LTN", head(cls), al)
    loss.backward()
    Fn.gemm = orig; Fn.set_compute_dtype("fp32")
    return log
for seed in range(3, 9):
    torch.manual_seed(seed)
    with torch.no_grad():
        for prm in list(enc.parameters()) + list(head.parameters()):
            if prm.dim() > 1:
                torch.nn.init.xavier_uniform_(prm)
    nf = (0.5 * torch.relu(torch.randn(bs, pn * L, P, d))).to(dev); af = (0.5 * torch.relu(torch.randn(bs, pn * L, P, d))).to(dev)
    l1, l3 = run("fp32"), run("f32x3")
    flips = sum(int(((a[3] > 0) != (b[3] > 0)).sum()) for a, b in zip(l1, l3) if a[0])
    fro = [float((a[3] - b[3]).double().norm() / (a[3].double().norm() + 1e-30)) for a, b in zip(l1, l3)]
    mx = [float((a[3] - b[3]).abs().max() / (a[3].abs().max() + 1e-30)) for a, b in zip(l1, l3)]
    fwd = max(fro[:18]); bwd = max(fro[18:])
    print(f"seed {seed}: relu flips {flips}; worst rel Frobenius diff fwd {fwd:.2e} bwd {bwd:.2e}; worst rel max diff fwd {max(mx[:18]):.2e} bwd {max(mx[18:]):.2e}")
