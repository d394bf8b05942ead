"""Race hunt: the packed-input attention kernels launched repeatedly on the same operands must reproduce their first outputs
bit for bit (probabilities, O, dQ | dK | dV, bias-table gradient), also while another stream keeps the memory system busy."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstc_vad_amd import functional as Fn
from lstc_vad_amd.models.MultiHeadAttention import relative_position_index_3d
dev = "cuda"
Fn.set_compute_dtype("bf16"); Fn.set_x3_threshold(0, 0, 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
side = torch.cuda.Stream()
junk = torch.randn(64 << 20, device=dev)
bad = 0
for (S, L, N, H, dk) in ((81, 5, 1024, 8, 256), (49, 3, 1024, 8, 256), (17, 1, 2048, 8, 256), (33, 2, 512, 4, 64)):
    M = N * S
    g = torch.Generator(device=dev).manual_seed(S)
    qkv = torch.randn(M, 3 * H * dk, device=dev, generator=g)
    do = torch.randn(M, H * dk, device=dev, generator=g)
    idx = relative_position_index_3d(L, 4).to(dev)
    tab = torch.randn((2 * L - 1) * 49, H, device=dev, generator=g) * 0.3
    qkv_p, do_p = Fn.pack3(qkv, False), Fn.pack3(do, False)
    first = None
    for r in range(reps):
        if r % 2:
            with torch.cuda.stream(side):          # a bandwidth hog next to the kernels under test
                junk.mul_(1.0001)
        o, pr = Fn.attn_fwd(qkv_p, None, None, N, S, H, dk, dk, tab, idx, 0.2, 7)
        gq, _, _, dt = Fn.attn_bwd(do_p, qkv_p, None, None, pr, N, S, H, dk, dk, tab, idx, 0.2, 7)
        torch.cuda.synchronize()
        n_o, n_g = M * H * dk * 2, M * 3 * H * dk * 2
        cur = (pr.clone(), o.buf[:n_o].clone(), gq.buf[:n_g].clone(), dt.clone())
        if first is None:
            first = cur
        else:
            for name, a, b in zip(("probs", "O", "dQKV", "dtable"), first, cur):
                if not torch.equal(a, b):
                    bad += 1
                    print(f"MISMATCH S={S} rep {r} {name}: {(a != b).sum().item()} elements", flush=True)
    print(f"S={S} N={N}: {reps} repetitions done", flush=True)
print("STRESS", "FAILED" if bad else "ok", bad)
