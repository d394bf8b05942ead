#!/usr/bin/env python3
"""Headline benchmark: snippets/sec of one full LSTC_VAD training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config ltn_sht|stn_sht|ltn_ucf|ltn_ubnormal]

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): LTN temporal transformer,
part_len=3, n_head=8, d_k=d_v=256, relative position bias (window 4), d_model=2048, n_hidden=4096, 3 layers,
MHA + FFN LayerNorm, fp32; B=64 videos (--batch_size 32 pairs), T=32 parts, P=16 patches per GPU;
reference dropout rates (0.2/0.2/0.1/0.6); synthetic I3D-like features resident in HBM.
A step = sequence reshape + Encoder + Classifier + MIL/CE loss + backward + [gradient all-reduce] + Adagrad.
For N>1 launch with ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...``: one process
per GPU over RCCL, every rank gets its own B=64 videos (weak scaling), value = all ranks' snippets / max time.

Prints ONE JSON line (rank 0) with `roofline` (the dominant kernel = the exact-f32 MFMA GEMM; achieved = the
launches' algorithmic 2MNK FLOPs / their HIP-event durations, measured live on the launch stream) and
`cpu_baseline` (the oracle = CPU restatement of the reference, timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (mode, encoder kwargs, step kwargs, dropout (attn, fc, ffn, head))
    "ltn_sht": ("LTN", dict(d_model=2048, d_inner=4096, MHA_layerNorm=True, FFN_layerNorm=True, relative_pe=True,
                            window_size=4, window_depth=3), dict(part_len=3, n_patch=16), (0.2, 0.2, 0.1, 0.6)),
    "stn_sht": ("STN", dict(d_model=2048, d_inner=3027, FFN_layerNorm=True), dict(part_len=1, n_patch=16),
                (0.1, 0.1, 0.1, 0.6)),
    "ltn_ucf": ("LTN", dict(d_model=2048, d_inner=4096, MHA_layerNorm=True, FFN_layerNorm=True, relative_pe=True,
                            window_size=4, window_depth=2), dict(part_len=2, n_patch=9), (0.2, 0.2, 0.1, 0.6)),
    "ltn_ubnormal": ("LTN", dict(d_model=1024, d_inner=4096, MHA_layerNorm=True, FFN_layerNorm=True, relative_pe=True,
                                 window_size=4, window_depth=5), dict(part_len=5, n_patch=16), (0.2, 0.2, 0.1, 0.6)),
}
FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def train_flops_per_sequence(S, d, Hd, F, n_layers, c):
    """F_fwd = L*[8 S d Hd + 4 S^2 Hd + 4 S d F] + head; F_alg = 3 F_fwd - 6 S d Hd."""
    f_fwd = n_layers * (8 * S * d * Hd + 4 * S * S * Hd + 4 * S * d * F) + 2 * (512 * d + 512 * 32 + 32 * c)
    return 3 * f_fwd - 6 * S * d * Hd


def cpu_baseline(cfg_name, threads):
    """Time the oracle's full training step (forward, loss, backward, Adagrad) on the host cores on a bounded
    sample of the same workload: same model width, 2+2 videos x 8 parts (32 sequences)."""
    from oracle import lstc_oracle as orc
    mode, ekw, skw, drops = CONFIGS[cfg_name]
    torch.set_num_threads(threads)
    bs, pn, L, P, d = 2, 8, skw["part_len"], skw["n_patch"], ekw["d_model"]
    ecfg = orc.EncoderCfg(n_layers=3, n_head=8, d_k=256, d_v=256, MHA_attn_dropout=drops[0], MHA_fc_dropout=drops[1],
                          FFN_dropout=drops[2], **ekw)
    st = orc.StepCfg(mode=mode, batch_size=bs, part_num=pn, part_len=L, n_patch=P, head_dropout=drops[3])
    g = torch.Generator().manual_seed(0)
    enc_P = {k: torch.randn(s, generator=g) * 0.02 for k, s in orc.encoder_param_shapes(ecfg).items()}
    for k in enc_P:
        if k.endswith("layer_norm.weight"):
            enc_P[k] = torch.ones_like(enc_P[k])
    if ekw.get("relative_pe"):
        for i in range(3):
            enc_P[f"layer_stack.{i}.slf_attn.relative_position_index"] = orc.relative_position_index_3d(L, 4)
    kind = "classifier" if mode == "LTN" else "regressor"
    head_P = {k: torch.randn(s, generator=g) * 0.02 for k, s in orc.head_param_shapes(d, kind).items()}
    enc_S = {k: torch.zeros_like(v) for k, v in enc_P.items() if v.is_floating_point()}
    head_S = {k: torch.zeros_like(v) for k, v in head_P.items()}
    nf = 0.5 * torch.relu(torch.randn(bs, pn * L, P, d, generator=g))
    af = 0.5 * torch.relu(torch.randn(bs, pn * L, P, d, generator=g))
    al = torch.rand(bs, pn * L, 1, generator=g)
    snippets = 2 * bs * pn * L
    times = []
    for it in range(3):
        t0 = time.perf_counter()
        _, enc_P, head_P, enc_S, head_S, _, _ = orc.train_step(enc_P, head_P, enc_S, head_S, ecfg, st, nf, af, al)
        times.append(time.perf_counter() - t0)
        if it >= 1 and sum(times) > 25:
            break
    t = min(times[1:]) if len(times) > 1 else times[0]
    return {"value": round(snippets / t, 2), "unit": "snippets/s", "cores": threads, "kind": "port",
            "sample": f"oracle train_step (fwd+loss+bwd+Adagrad), same model, {2 * bs} videos x {pn} parts x "
                      f"{L} snippets = {snippets} snippets/step, best of {max(1, len(times) - 1)} after 1 warm-up, "
                      f"{t:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="ltn_sht", choices=list(CONFIGS) + ["mixed_ubn_sht"],
                    help="mixed_ubn_sht = BASELINE config 5: half the videos UBnormal (d=1024, L=5), half SHT (d=2048, L=3), two "
                         "model pairs stepped in one iteration (lstc_vad_amd.engine.MixedStep)")
    ap.add_argument("--batch_size", type=int, default=32, help="normal/abnormal pairs per GPU (B = 2*batch_size videos)")
    ap.add_argument("--part_num", type=int, default=32)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "f32x3"],
                    help="GEMM compute type: fp32 = exact-f32 MFMA (headline, parity mode); f32x3 = f32-accurate products on the 16-bit "
                         "matrix cores (2 scaled f16 planes per operand, 3 plane products, csrc/gemm_pk.hip); bf16 = bf16 MFMA on f32 storage")
    ap.add_argument("--h2d", action="store_true", help="also time the step with the batch arriving from pinned host memory "
                    "every step through lstc_vad_amd.feed.PinnedFeeder (reported as pcie_inclusive, never as value)")
    ap.add_argument("--naive-last-layer", action="store_true", help="evaluate the last encoder layer for every token like the "
                    "reference (A/B only: the default skips rows/projections nobody reads, with identical results)")
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-events", action="store_true")
    ap.add_argument("--no-f32x3", action="store_true", help="skip the second timed pass in f32x3 mode (reported as the \"f32x3\" "
                    "object of the JSON line; `value` is always the --dtype mode, exact-f32 MFMA by default)")
    a = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the LSTC_VAD hot path here is HIP-only (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("LSTC_FORCE_DIST", "0") == "1"      # exercise the RCCL code path on a single GPU
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    if a.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.engine import TrainStep
    from lstc_vad_amd.models import Classifier, Encoder, Regressor

    Fn.set_compute_dtype(a.dtype)
    from lstc_vad_amd.engine import MixedStep

    def make(cfg_name, bs, seed_off=0):
        mode, ekw, skw, drops = CONFIGS[cfg_name]
        if a.no_dropout:
            drops = (0.0, 0.0, 0.0, 0.0)
        pn, L, P, d = a.part_num, skw["part_len"], skw["n_patch"], ekw["d_model"]
        args = Namespace(batch_size=bs, part_num=pn, part_len=L, n_patch=P, lambda_1=0.01, lambda_MIL=1.0, lambda_CE=0.8,
                         lambda_BCE=1.0, lambda_normal=0.2, lambda_abnormal=2.0, temporal_only=False, clip_grad=False)
        torch.manual_seed(seed_off)       # same replica on every rank
        enc = Encoder(n_layers=3, n_head=8, d_k=256, d_v=256, MHA_attn_dropout=drops[0], MHA_fc_dropout=drops[1],
                      FFN_dropout=drops[2], weight_init=(mode != "LTN"), **ekw).to(dev).train()
        head = (Classifier(d, drops[3]) if mode == "LTN" else Regressor(d, drops[3])).to(dev).train()
        ts = TrainStep(args, mode, enc, head, lr_encoder=1e-4, lr_head=1e-2, weight_decay=1e-3, cls_only=not a.naive_last_layer)
        gen = torch.Generator(device=dev).manual_seed(1000 + rank + 97 * seed_off)      # every rank its own videos
        T = pn * L
        nf = 0.5 * torch.relu(torch.randn(bs, T, P, d, device=dev, generator=gen))
        af = 0.5 * torch.relu(torch.randn(bs, T, P, d, device=dev, generator=gen))
        u = torch.rand(bs, T, 1, device=dev, generator=gen)
        al = torch.where(u > 0.9, u, torch.zeros_like(u))            # pseudo labels, rule of README.md:27
        return ts, (nf, af, al)

    mixed = a.config == "mixed_ubn_sht"
    if mixed:
        parts = [make("ltn_ubnormal", a.batch_size // 2, 0), make("ltn_sht", a.batch_size // 2, 1)]
        ms_ = MixedStep([p[0] for p in parts])
        run_step = lambda: ms_.step([p[1] for p in parts])[-1]
        names = ["ltn_ubnormal", "ltn_sht"]
    else:
        ts, (nf, af, al) = make(a.config, a.batch_size)
        run_step = lambda: ts.step(nf, af, al)
        names = [a.config]
    mode, ekw, skw, drops = CONFIGS[names[-1]]
    bs, pn, L, P, d = (a.batch_size // 2 if mixed else a.batch_size), a.part_num, skw["part_len"], skw["n_patch"], ekw["d_model"]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        run_step()
    sync()
    prof = None if a.no_gemm_events else []
    Fn.set_gemm_profiling(prof)
    pack_prof = [] if (prof is not None and a.dtype == "f32x3") else None
    Fn._pack_prof = pack_prof
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sc = run_step()
    sync()
    dt = time.perf_counter() - t0
    Fn.set_gemm_profiling(None)
    Fn._pack_prof = None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    scal = [float(x) for x in sc.cpu()]
    hbm_peak = torch.cuda.max_memory_allocated(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    # ---- second pass: the same K steps with the big products on the 16-bit matrix cores (f32-accurate split operands,
    # csrc/gemm_pk.hip).  Reported next to the headline, never as `value`.
    x3 = None
    if a.dtype == "fp32" and not a.no_f32x3:
        Fn.set_compute_dtype("f32x3")
        for _ in range(max(1, a.warmup)):
            run_step()
        sync()
        gp, pp = ([], []) if prof is not None else (None, None)
        Fn.set_gemm_profiling(gp)
        Fn._pack_prof = pp
        t3 = time.perf_counter()
        for _ in range(a.steps):
            sc3 = run_step()
        sync()
        dt3 = time.perf_counter() - t3
        Fn.set_gemm_profiling(None)
        Fn._pack_prof = None
        Fn.set_compute_dtype("fp32")
        if world > 1:
            tt = torch.tensor([dt3], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt3 = float(tt.item())
        x3 = {"dt": dt3, "gp": gp, "pp": pp, "loss": float(sc3.cpu()[0]), "hbm": torch.cuda.max_memory_allocated(dev)}
    pcie = None
    if a.h2d and world == 1 and not mixed:
        from lstc_vad_amd.feed import PinnedFeeder
        host = tuple(t.cpu() for t in (nf, torch.zeros_like(al), af, al))
        feeder = PinnedFeeder((host for _ in range(a.steps + 2)), dev)
        t1 = None
        for i, (hnf, _, haf, hal) in enumerate(feeder):
            if i == 2:
                torch.cuda.synchronize(); t1 = time.perf_counter()
            ts.step(hnf, haf, hal)
        torch.cuda.synchronize()
        dth = (time.perf_counter() - t1) / a.steps
        pcie = {"value": round(2 * bs * pn * L / dth, 1), "unit": "snippets/s", "ms_per_step": round(1e3 * dth, 3),
                "batch_MB": round(sum(t.numel() * 4 for t in host) / 1e6, 1),
                "note": "batch copied pageable->pinned->HBM every step on a side stream, overlapped with the previous step"}

    snippets_per_step = sum(2 * bs * pn * CONFIGS[n][2]["part_len"] for n in names) * world
    value = snippets_per_step * a.steps / dt
    if rank == 0:
        S = 1 + (L * P if mode == "LTN" else P)
        nseq = 2 * bs * pn * (1 if mode == "LTN" else L)
        f_seq = train_flops_per_sequence(S, d, 2048, ekw["d_inner"], 3, 2 if mode == "LTN" else 1)
        alg_flops = 0.0
        for n in names:
            m_, e_, s_, _ = CONFIGS[n]
            S_ = 1 + (s_["part_len"] * s_["n_patch"] if m_ == "LTN" else s_["n_patch"])
            alg_flops += train_flops_per_sequence(S_, e_["d_model"], 2048, e_["d_inner"], 3, 2 if m_ == "LTN" else 1) * \
                2 * bs * pn * (1 if m_ == "LTN" else s_["part_len"])
        roof = None
        if prof:
            fl = sum(p[0] for p in prof)
            ms = sum(p[1].elapsed_time(p[2]) for p in prof)
            ach = fl / (ms * 1e-3) / 1e12
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "gemm_pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    traffic = None if mixed else json.load(open(pmc)).get(a.config)
                except Exception:
                    traffic = None
            # f32x3: three f16 MFMA products per f32 product -> the f32-equivalent ceiling is the 16-bit peak / 3
            peak = FP32_MFMA_PEAK_TFLOPS if a.dtype == "fp32" else (round(2500.0 / 3, 1) if a.dtype == "f32x3" else 2500.0)
            kname = {"fp32": "gemm_f32_kernel (v_mfma_f32_32x32x2_f32)",
                     "f32x3": "gemm_pk_kernel (3 x v_mfma_f32_32x32x16_f16 per f32 product, packed 2-plane operands; small "
                              "products on gemm_f32_kernel)",
                     "bf16": "gemm_bf16c_kernel (v_mfma_f32_32x32x16_bf16, f32 operands in HBM)"}[a.dtype]
            roof = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "launches_per_step": len(prof) // a.steps,
                    "gemm_ms_per_step": round(ms / a.steps, 3),
                    "gemm_flops_per_step": fl / a.steps,
                    # SURVEY 8(d) counts the full last layer; the step skips its dead rows (only the CLS token of the
                    # last layer is read), so executed GEMM FLOPs < algorithmic FLOPs.  frac is on EXECUTED work.
                    "step_algorithmic_tflop": round(alg_flops / 1e12, 3),
                    "step_executed_gemm_tflop": round(fl / a.steps / 1e12, 3),
                    "step_frac_of_peak_executed": round(fl / a.steps / (dt / a.steps) / 1e12 / peak, 4)}
            if pack_prof:
                pms = sum(q[1].elapsed_time(q[2]) for q in pack_prof)
                roof["pack_ms_per_step"] = round(pms / a.steps, 3)
                roof["pack_launches_per_step"] = len(pack_prof) // a.steps
                roof["pack_input_GBps"] = round(sum(q[0] for q in pack_prof) / (pms * 1e-3) / 1e9, 1)
        out = {"metric": "snippets/sec training step (B=64,T=32,P=16,d=2048)", "value": round(value, 1),
               "unit": "snippets/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": {"fp32": "f32", "f32x3": "f32 (operands split into 2 scaled f16 planes, 3 f16-MFMA products, f32 accumulate)",
                         "bf16": "bf16 (f32 storage/accumulate)"}[a.dtype], "data": "synthetic",
               "config": {"workload": ("mixed batch (BASELINE config 5): per GPU 32 UBnormal videos (d=1024, L=5, S=81) + 32 SHT "
                                       "videos (d=2048, L=3, S=49), two model pairs, one iteration; second model: " if mixed else "") +
                                      f"{names[-1]}: {mode} full training step (fwd+loss+bwd+"
                                      f"{'allreduce+' if world > 1 else ''}Adagrad), per GPU B={2 * bs} videos x T={pn} parts x "
                                      f"L={L} snippets x P={P} patches, d_model={d}, n_hidden={ekw['d_inner']}, S={S}, "
                                      f"{nseq} sequences/GPU/step, dropout={'off' if a.no_dropout else 'reference rates'}, last layer: "
                                      f"{'all tokens (naive)' if a.naive_last_layer else 'CLS token only, K/V projections re-associated (exact)'}",
                          "global_videos": 2 * bs * world * len(names), "parallelism": f"dp{world}"},
               "loss": scal[0], "hbm_peak_GB": round(hbm_peak / 1e9, 2), "roofline": roof}
        if x3 is not None:
            o3 = {"value": round(snippets_per_step * a.steps / x3["dt"], 1), "unit": "snippets/s",
                  "ms_per_step": round(1e3 * x3["dt"] / a.steps, 3), "loss": x3["loss"],
                  "hbm_peak_GB": round(x3["hbm"] / 1e9, 2),
                  "dtype": "f32 storage and accumulation; products of the large GEMMs on the f16 matrix cores: operands scaled by a "
                           "power of two and split into two f16 planes (x*s = h + l to 2^-24), three plane products hh + hl + lh",
                  "accuracy": "error against f64 <= the exact-f32 MFMA kernel's (tests/test_hip_parity.py::test_f32x3_*, "
                              "tools/x3_probe.hip: rms 0.82e-7 vs 0.92e-7 for an f32 fma chain); golden training parity at the f32 tolerances"}
            if x3["gp"]:
                fl3 = sum(q[0] for q in x3["gp"])
                ms3 = sum(q[1].elapsed_time(q[2]) for q in x3["gp"])
                pk3 = round(2500.0 / 3, 1)
                o3["roofline"] = {"bound": "mfma", "kernel": "gemm_pk2s_kernel (3 x v_mfma_f32_32x32x16_f16 per f32 product)",
                                  "achieved": round(fl3 / (ms3 * 1e-3) / 1e12, 2), "peak": pk3, "unit": "TFLOP/s (f32-equivalent)",
                                  "frac": round(fl3 / (ms3 * 1e-3) / 1e12 / pk3, 4), "gemm_ms_per_step": round(ms3 / a.steps, 3)}
                if x3["pp"]:
                    pms = sum(q[1].elapsed_time(q[2]) for q in x3["pp"])
                    o3["roofline"]["pack_ms_per_step"] = round(pms / a.steps, 3)
            out["f32x3"] = o3
        if pcie:
            out["pcie_inclusive"] = pcie
        if world == 1 and not a.no_cpu_baseline and not mixed:
            # torch-CPU sgemm on the GPU box's host peaks at 16-32 threads (tools/cpu_threads_scan.py: 1.3 TFLOP/s
            # at 16-32, 0.5 at 128 of 256 hardware threads), so the baseline uses min(32, available) threads
            out["cpu_baseline"] = cpu_baseline(names[-1], min(32, len(os.sched_getaffinity(0))))
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
