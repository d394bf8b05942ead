"""One forward + one backward launch of the fused attention kernels at a layer shape, for rocprofv3 --pmc passes (tools/pmc_attn.sh):
python3 tools/attn_one.py [fp32|bf16] [S] [L] [N] [dk] [packed|packin]   (packin: Q|K|V / dO as packed bf16 operands)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstc_vad_amd import functional as Fn
from lstc_vad_amd.models.MultiHeadAttention import relative_position_index_3d
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
S, L, N, dk = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 49), (3, 3), (4, 2048), (5, 256)))
packed = len(sys.argv) > 6 and sys.argv[6] == "packed"
packin = len(sys.argv) > 6 and sys.argv[6] == "packin"
H, dev = 8, "cuda"
M = N * S
q, k, v, do = (torch.randn(M, H * dk, device=dev) for _ in range(4))
idx = relative_position_index_3d(L, 4).to(dev)
tab = torch.randn((2 * L - 1) * 49, H, device=dev) * 0.1
Fn.set_compute_dtype(mode)
if packin:
    Fn.set_x3_threshold(0, 0, 0)
    qkv_p, do_p = Fn.pack3(torch.cat([q, k, v], dim=1), False), Fn.pack3(do, False)
    for _ in range(2):
        o, p = Fn.attn_fwd(qkv_p, None, None, N, S, H, dk, dk, tab, idx, 0.2, 7)
        Fn.attn_bwd(do_p, qkv_p, None, None, p, N, S, H, dk, dk, tab, idx, 0.2, 7)
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(2):
    o, p = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, tab, idx, 0.2, 7, packed=packed)
    Fn.attn_bwd(do, q, k, v, p, N, S, H, dk, dk, tab, idx, 0.2, 7, packed=packed)
torch.cuda.synchronize()
