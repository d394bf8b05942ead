"""BASELINE config 3 on the synthetic (learnable) world in two GEMM modes: runs tools/coteach_loop_synthetic.sh once per mode and
compares the pseudo-label files and the logged losses.  usage: coteach_modes_compare.py OUT_DIR [STEPS]"""
import glob, os, re, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
steps = sys.argv[2] if len(sys.argv) > 2 else "6"
res = {}
for mode in ("fp32", "bf16"):
    d = os.path.join(out, mode)
    r = subprocess.run(["bash", os.path.join(root, "tools", "coteach_loop_synthetic.sh"), d, mode, steps], capture_output=True, text=True,
                       env=dict(os.environ, PYTHONPATH=root))
    print(mode, "rc", r.returncode, r.stdout[-600:], r.stderr[-300:] if r.returncode else "")
    labels = {f: np.load(os.path.join(d, f), allow_pickle=True).tolist() for f in ("STN_pseudo_labels.npy", "LTN_pseudo_labels.npy")}
    logs = {}
    for st in ("stn", "ltn", "mce"):
        txt = "".join(open(f, errors="ignore").read() for f in sorted(glob.glob(os.path.join(d, st, "**", "*"), recursive=True)) if os.path.isfile(f) and not f.endswith((".ckpt", ".npy", ".pth")))
        logs[st] = [[float(x) for x in re.findall(r"(?:loss|err|l1) (-?\d+\.\d+)", ln)] for ln in txt.splitlines() if "]: " in ln and "loss" in ln]
        logs[st + "_auc"] = re.findall(r"now test_AUC is (\S+)", txt)
    res[mode] = (labels, logs)
A, B = res["fp32"], res["bf16"]
for f in A[0]:
    same = tot = 0; mv = 0.0
    for k in A[0][f]:
        a, b = np.asarray(A[0][f][k]), np.asarray(B[0][f][k])
        same += int(((a > 0) == (b > 0)).sum()); tot += a.size
        both = (a > 0) & (b > 0)
        if both.any(): mv = max(mv, float(np.abs(a - b)[both].max()))
    print(f, "zero pattern equal on", same, "of", tot, "max |label diff| where both > 0:", mv, " nonzero frac fp32", float(np.mean(np.concatenate([np.asarray(v).ravel() for v in A[0][f].values()]) > 0)))
for st in ("stn", "ltn", "mce"):
    a, b = np.array(A[1][st]), np.array(B[1][st])
    print(st, "steps", a.shape, "max |loss term diff| per step:", np.abs(a - b).max(1).round(4).tolist() if a.shape == b.shape and a.size else (a.shape, b.shape))
    print("   fp32 first/last:", a[0].tolist() if a.size else None, a[-1].tolist() if a.size else None, " auc", A[1][st + "_auc"], B[1][st + "_auc"])
