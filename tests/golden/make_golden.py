#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the REAL reference.

Run in the build container only (it needs /root/reference, which never travels to
the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the unmodified reference classes (``models.Encoder``, ``models.Regressor``,
``models.Classifier``) and loss functions (``Train.*.get_MIL_loss`` etc.) on CPU with
``cv2`` / ``h5py`` stubbed (both are dead imports for this path), feeds
them inputs and weights from ``lstc_vad_amd.synthetic`` (portable counter-based
generator), runs forward + loss + backward + two ``torch.optim.Adagrad`` steps exactly
like the reference train loops do, and stores inputs and expected outputs.  All dropout
rates are 0 (the reference's masks come from torch's generator and cannot be replayed
elsewhere).  The fixtures are DATA: inputs + expected outputs; no reference source text.
"""
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
for name in ("cv2", "h5py"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, HERE)
from _refimport import assert_reference, bind_reference     # noqa: E402

# The repo's own utils/ models/ Train/ Test/ shims share the reference's top-level names: bind the reference's packages
# by file location first (a regular package would otherwise shadow the reference's namespace package utils/).
bind_reference(REF)
sys.path.insert(1, ROOT)

from models.Encoder import Encoder as RefEncoder            # noqa: E402  (reference)
from models.Regressor import Regressor as RefRegressor      # noqa: E402
from models.Classifier import Classifier as RefClassifier   # noqa: E402
from models.MultiHeadAttention import MultiHeadAttention as RefMHA   # noqa: E402
import Train.temporal_transformer_shanghaitech as ref_ltn   # noqa: E402
import Train.spatio_transformer_shanghaitech as ref_stn     # noqa: E402
import Train.spatio_transformer_MIL_CE as ref_coteach       # noqa: E402
import utils.utils as ref_utils                             # noqa: E402
import utils.eval_utils as ref_eval_utils                   # noqa: E402
from utils.eval_utils import eval as ref_eval               # noqa: E402

assert_reference(RefEncoder, RefRegressor, RefClassifier, RefMHA, ref_ltn, ref_stn, ref_coteach, ref_utils,
                 ref_eval_utils, ref_eval)

from lstc_vad_amd import synthetic as syn                   # noqa: E402
assert os.path.realpath(syn.__file__).startswith(os.path.realpath(ROOT) + os.sep)

torch.set_num_threads(4)
OUT_DIR = HERE      # ``--out DIR`` writes elsewhere (tests/test_golden_recipes.py regenerates into a temp dir and diffs)


from cases import CASES, FULL_CASES, PACKED_CASES, HEADLINE_CASES, fill_params, sample_index   # noqa: E402


def make_args(enc_kw, st_kw):
    return Namespace(batch_size=st_kw["batch_size"], part_num=st_kw["part_num"], part_len=st_kw["part_len"],
                     n_patch=st_kw["n_patch"], d_model=enc_kw["d_model"], lambda_1=0.01,
                     lambda_MIL=1.0, lambda_CE=0.8, lambda_BCE=1.0, lambda_normal=0.2, lambda_abnormal=2.0,
                     temporal_only=st_kw.get("temporal_only", False), clip_grad=st_kw.get("clip_grad", False))


def ref_forward_loss(mode, args, enc, head, tnf, taf, tal):
    """Forward + loss exactly as the reference train loops spell them, on the reference's own modules and loss functions."""
    bs, pn, L, P, d = args.batch_size, args.part_num, args.part_len, args.n_patch, args.d_model
    if mode == "LTN":      # Train/temporal_transformer_shanghaitech.py:103-134
        norm_labs = torch.zeros([bs, pn, 2]); norm_labs[:, :, 0] += 1
        ab = tal.view([bs, pn, L]).mean(dim=-1).view([bs, pn, 1])
        tmp = torch.zeros([bs, pn, 2]); tmp[:, :, 1] = ab[:, :, 0]; tmp[:, :, 0] = 1 - tmp[:, :, 1]
        clip_labs = torch.cat([norm_labs, tmp], dim=0)
        feats = torch.cat([tnf.float().view([bs * pn, L * P, d]), taf.float().view([bs * pn, L * P, d])], dim=0)
        enc_out = enc(feats)
        cls = enc_out[:, 0, :].float().view([bs * 2, pn, d])
        outputs = head(cls).view([bs * 2 * pn, -1])
        score = outputs[:, 1]
        if not args.temporal_only:
            aux = ref_ltn.get_CE_loss(args, outputs, clip_labs.view([bs * 2 * pn, -1]))
        else:
            aux = torch.zeros(())
        mil, err, l1 = ref_ltn.get_MIL_loss(args, score)
        loss = args.lambda_MIL * mil + args.lambda_CE * aux
    elif mode == "STN":    # Train/spatio_transformer_shanghaitech.py:90-101
        feats = torch.cat([tnf.float().view([bs * pn * L, P, d]), taf.float().view([bs * pn * L, P, d])], dim=0)
        enc_out = enc(feats)
        cls = enc_out[:, 0, :].float().view([bs * 2, pn * L, d])
        outputs = head(cls).view([bs * 2, pn * L, -1])
        loss, err, l1 = ref_stn.get_MIL_loss(args, outputs)
        mil, aux, score = loss, torch.zeros(()), outputs.reshape(-1)
    else:                  # Train/spatio_transformer_MIL_CE.py:156-181
        norm_labs = torch.zeros([bs, pn, 2]); norm_labs[:, :, 0] += 1
        ab = tal.view([bs, pn, L]).mean(dim=-1).view([bs, pn, 1])
        tmp = torch.zeros([bs, pn, 2]); tmp[:, :, 1] = ab[:, :, 0]; tmp[:, :, 0] = 1 - tmp[:, :, 1]
        clip_labs = torch.cat([norm_labs, tmp], dim=0)
        feats = torch.cat([tnf.float().view([bs * pn * L, P, d]), taf.float().view([bs * pn * L, P, d])], dim=0)
        enc_out = enc(feats)
        outputs = head(enc_out[:, 0, :])
        mil, err, l1 = ref_coteach.get_MIL_loss(args, outputs, L)
        aux = ref_coteach.get_BCE_loss(args, torch.mean(outputs.view([bs * 2, pn, L]), dim=-1), clip_labs)
        loss = args.lambda_BCE * aux + mil
        score = outputs.reshape(-1)
    return enc_out, outputs, score, loss, mil, err, l1, aux


def scalars_of(loss, mil, err, l1, aux):
    return np.array([loss.item(), mil.item(), err.item(), l1.item(), float(aux)], np.float64)


def run_case(name, mode, enc_kw, st_kw, seed):
    bs, pn, L, P = st_kw["batch_size"], st_kw["part_num"], st_kw["part_len"], st_kw["n_patch"]
    d = enc_kw["d_model"]
    args = make_args(enc_kw, st_kw)
    enc = RefEncoder(n_layers=3, MHA_attn_dropout=0.0, MHA_fc_dropout=0.0, FFN_dropout=0.0, position_dropout=0.0,
                     weight_init=False, **enc_kw)
    head = RefRegressor(d, 0.0, weight_init=False) if mode != "LTN" else RefClassifier(d, 0.0, weight_init=False)
    fill_params(enc, seed)
    fill_params(head, seed + 1)
    enc.train(); head.train()
    nf, nl, af, al = syn.training_batch(bs, pn, L, P, d, seed=seed, with_pseudo=True, threshold=0.6)
    out = {"seed": np.int64(seed)}
    for k, v in enc.state_dict().items():
        out["enc_init." + k] = v.numpy().copy()
    for k, v in head.state_dict().items():
        out["head_init." + k] = v.numpy().copy()
    out.update(norm_feats=nf, abnorm_feats=af, abnorm_labs=al)

    # Adagrad exactly as Train/temporal_transformer_shanghaitech.py:83-85
    opt = torch.optim.Adagrad([{"params": enc.parameters(), "lr": 1e-4},
                               {"params": head.parameters(), "lr": 1e-2}], weight_decay=1e-3)
    tnf, taf, tal = torch.from_numpy(nf), torch.from_numpy(af), torch.from_numpy(al)
    for step in range(2):
        enc_out, outputs, score, loss, mil, err, l1, aux = ref_forward_loss(mode, args, enc, head, tnf, taf, tal)
        opt.zero_grad()
        loss.backward()
        if args.clip_grad:
            torch.nn.utils.clip_grad_norm_(enc.parameters(), 10)
            torch.nn.utils.clip_grad_norm_(head.parameters(), 10)
        if step == 0:
            out["enc_out"] = enc_out.detach().numpy().copy()
            out["outputs"] = outputs.detach().numpy().copy()
            out["score"] = score.detach().numpy().copy()
            out["scalars"] = scalars_of(loss, mil, err, l1, aux)
            for k, p in enc.named_parameters():
                if p.grad is not None:
                    out["enc_grad." + k] = p.grad.numpy().copy()
            for k, p in head.named_parameters():
                if p.grad is not None:
                    out["head_grad." + k] = p.grad.numpy().copy()
        else:
            out["scalars_step2"] = scalars_of(loss, mil, err, l1, aux)
        opt.step()
    for k, v in enc.state_dict().items():
        out["enc_after2." + k] = v.numpy().copy()
    for k, v in head.state_dict().items():
        out["head_after2." + k] = v.numpy().copy()

    # eval-mode forward and a short-tail sequence (S = 1 + 1*P) on the initial weights
    enc2 = RefEncoder(n_layers=3, MHA_attn_dropout=0.3, MHA_fc_dropout=0.3, FFN_dropout=0.3, position_dropout=0.3,
                      weight_init=False, **enc_kw)
    fill_params(enc2, seed)
    enc2.eval()
    with torch.no_grad():
        if mode == "LTN":
            x = torch.from_numpy(nf).view([bs * pn, L * P, d])[:3]
            out["eval_enc_out"] = enc2(x).numpy().copy()
            if L > 1:   # tail part with fewer clips (Train/pseudo_labels_generator_temporal.py:110-137)
                out["eval_tail_enc_out"] = enc2(x[:, : (L - 1) * P]).numpy().copy()
                out["eval_tail1_enc_out"] = enc2(x[:, :P]).numpy().copy()
        else:
            x = torch.from_numpy(nf).view([bs * pn * L, P, d])[:3]
            out["eval_enc_out"] = enc2(x).numpy().copy()
    np.savez_compressed(os.path.join(OUT_DIR, name + ".npz"), **out)
    print(f"{name}: loss {out['scalars'][0]:.6f} -> wrote {name}.npz "
          f"({os.path.getsize(os.path.join(OUT_DIR, name + '.npz')) / 1024:.0f} KiB)")


RELU_EDGE = 4e-6     # |W1 x + b1| below which a ReLU decision is treated as summation-order dependent (f32, K = 2048, |x| ~ 1)
HEAD_EDGE_FACTOR = 4 # the head's first Linear (models/Classifier.py:8-10, models/Regressor.py:7-9) sits behind three encoder layers:
                     # its input already carries their accumulated f32 noise, so its band is 4x the FFN band


def run_full_case(name, mode, enc_kw, st_kw, seed, relu_edge=None):
    """Full-width step on the reference: two training steps; keeps samples, norms and scalars only."""
    bs, pn, L, P = st_kw["batch_size"], st_kw["part_num"], st_kw["part_len"], st_kw["n_patch"]
    d = enc_kw["d_model"]
    args = make_args(enc_kw, st_kw)
    torch.set_num_threads(8)
    enc = RefEncoder(n_layers=3, MHA_attn_dropout=0.0, MHA_fc_dropout=0.0, FFN_dropout=0.0, position_dropout=0.0,
                     weight_init=False, **enc_kw)
    head = RefRegressor(d, 0.0, weight_init=False) if mode != "LTN" else RefClassifier(d, 0.0, weight_init=False)
    fill_params(enc, seed)
    fill_params(head, seed + 1)
    enc.train(); head.train()
    nf, nl, af, al = syn.training_batch(bs, pn, L, P, d, seed=seed, with_pseudo=True, threshold=0.6)
    out = {"seed": np.int64(seed)}
    opt = torch.optim.Adagrad([{"params": enc.parameters(), "lr": 1e-4},
                               {"params": head.parameters(), "lr": 1e-2}], weight_decay=1e-3)
    tnf, taf, tal = torch.from_numpy(nf), torch.from_numpy(af), torch.from_numpy(al)
    # ReLU decisions that depend on the summation order.  At this size each FFN layer holds ~2e7 hidden units per step and a few
    # dozen of their pre-activations W1 x + b1 lie within float32 rounding of zero (|pre| < RELU_EDGE): whether such a unit
    # counts as active is decided by the order in which an implementation adds the 2048 products (any two f32 BLAS disagree
    # on a handful of them - measured: 4 / 1 / 2 per layer between rocBLAS sgemm and an f64 product, the same between the
    # HIP kernel and f64), and one flipped unit with a large upstream gradient moves the small layer-0 attention gradients by
    # up to 1e-3 of their maximum.  The fixture therefore records WHICH units the reference run found on the edge and what it
    # decided there (its pre-activation value); the GPU test lets the HIP step take the same decisions at exactly these
    # units (tests/test_hip_parity.py, ``_align_relu_edges``) and compares everything else at the strict tolerances.
    edges = {}
    hooks = []
    for li, layer in enumerate(enc.layer_stack):
        def grab(mod, inp, outp, li=li):
            if li in edges:
                return                              # first step only
            pre = outp.detach().reshape(-1, outp.shape[-1])
            t, j = torch.nonzero(pre.abs() < (relu_edge or RELU_EDGE), as_tuple=True)
            edges[li] = (torch.stack([t, j], 1).numpy().astype(np.int64), pre[t, j].numpy().copy())
        hooks.append(layer.pos_ffn.w_1.register_forward_hook(grab))

    # the head's hidden Linear(d, 512) + ReLU (models/Classifier.py:8-10, models/Regressor.py:7-9): same recording, key "head";
    # row = row of the head's [-1, d] input view
    def grab_head(mod, inp, outp):
        if "head" in edges:
            return
        pre = outp.detach().reshape(-1, outp.shape[-1])
        t, j = torch.nonzero(pre.abs() < HEAD_EDGE_FACTOR * (relu_edge or RELU_EDGE), as_tuple=True)
        edges["head"] = (torch.stack([t, j], 1).numpy().astype(np.int64), pre[t, j].numpy().copy())
    hooks.append((head.classifier if mode == "LTN" else head.regressor)[0].register_forward_hook(grab_head))
    if mode == "LTN":
        # inference on the initial weights, as the pseudo-label generator feeds it (Train/pseudo_labels_generator_temporal.py:
        # 110-146: full parts of part_len clips, then the video's short tail as a SHORTER sequence - S = 1 + (L-1) P and 1 + P;
        # for L = 3 that is S = 33 and S = 17): P(abnormal) of the first 8 sequences of the normal batch
        enc.eval(); head.eval()
        with torch.no_grad():
            x = tnf.float().view([bs * pn, L * P, d])[:8]
            for tag, xs in (("full", x), ("tail", x[:, :(L - 1) * P]), ("tail1", x[:, :P])):
                out["eval_scores_" + tag] = head(enc(xs)[:, 0, :]).numpy().copy()
        enc.train(); head.train()
        edges.clear()                                   # the hooks fired on these passes too: record the TRAINING step's units
    for step in range(2):
        enc_out, outputs, score, loss, mil, err, l1, aux = ref_forward_loss(mode, args, enc, head, tnf, taf, tal)
        opt.zero_grad()
        loss.backward()
        if args.clip_grad:                   # Train/temporal_transformer_shanghaitech.py:139-141
            # the float64 norm of the same gradients next to the one torch's clip_grad_norm_ computes: on the CPU its float32
            # norm of norms over 100.7 M elements comes out 3.1e-4 LOW here (10.40038 against 10.40359), and the clip
            # coefficient - hence every clipped gradient the fixture holds - inherits that
            out[f"clip_total_norm_f64_step{step}"] = np.array(
                [float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None))) for m in (enc, head)])
            out[f"clip_total_norm_step{step}"] = np.array([float(torch.nn.utils.clip_grad_norm_(enc.parameters(), 10)),
                                                           float(torch.nn.utils.clip_grad_norm_(head.parameters(), 10))], np.float64)
        if step == 0:
            for h in hooks:
                h.remove()
            for li, (tj, val) in edges.items():
                out[f"relu_edge.{li}"] = tj             # [n, 2]: (token row of the [N*S, n_hidden] hidden, hidden unit)
                out[f"relu_edge_pre.{li}"] = val        # the reference's pre-activation there (its decision: > 0)
            eo = enc_out.detach()
            out["cls_rows"] = eo[:, 0, :].numpy().copy()[::max(1, eo.shape[0] // 16)][:16]       # 16 sequences' CLS rows
            out["tok_rows"] = eo[::max(1, eo.shape[0] // 8), eo.shape[1] // 2, :].numpy().copy()[:8]   # 8 mid-sequence rows
            out["outputs"] = outputs.detach().numpy().copy()
            out["score"] = score.detach().numpy().copy()
            out["scalars"] = scalars_of(loss, mil, err, l1, aux)
            for pre, mod in (("enc", enc), ("head", head)):
                for k, p in mod.named_parameters():
                    if p.grad is None:
                        continue
                    g = p.grad.detach().double().reshape(-1)
                    out[f"{pre}_gnorm.{k}"] = np.float64(g.norm().item())
                    out[f"{pre}_gmax.{k}"] = np.float64(g.abs().max().item())
                    out[f"{pre}_gs.{k}"] = p.grad.detach().reshape(-1).numpy()[sample_index(g.numel())].copy()
        else:
            out["scalars_step2"] = scalars_of(loss, mil, err, l1, aux)
        opt.step()
    for pre, mod in (("enc", enc), ("head", head)):
        for k, p in mod.named_parameters():
            out[f"{pre}_w2s.{k}"] = p.detach().reshape(-1).numpy()[sample_index(p.numel())].copy()
    torch.set_num_threads(4)
    np.savez_compressed(os.path.join(OUT_DIR, name + ".npz"), **out)
    print(f"{name}: loss {out['scalars'][0]:.6f} -> wrote {name}.npz "
          f"({os.path.getsize(os.path.join(OUT_DIR, name + '.npz')) / 1024:.0f} KiB)")


def misc():
    out = {}
    for (L, ws) in [(3, 4), (2, 4), (5, 4), (2, 3), (1, 4)]:
        m = RefMHA(2, 8, 4, 4, relative_pe=True, window_size=ws, window_depth=L)
        out[f"relidx3d_L{L}_ws{ws}"] = m.relative_position_index.numpy().copy()
    for ws in (3, 4):
        m = RefMHA(2, 8, 4, 4, relative_pe_2D=True, window_size=ws)
        out[f"relidx2d_ws{ws}"] = m.relative_position_index.numpy().copy()
    # AUC of fixed score / label vectors through utils/eval_utils.py:139-143 (incl. ties and [N,1] shape)
    sc = syn.uniform((400,), 7, 1).astype(np.float64)
    sc = np.round(sc * 20) / 20.0                      # many ties
    lb = (syn.uniform((400,), 7, 2) + 0.3 * sc > 0.7).astype(np.float64)
    out["auc_scores"], out["auc_labels"] = sc, lb
    out["auc_value"] = np.float64(ref_eval(list(sc.reshape(-1, 1)), list(lb), None))
    sc2 = syn.uniform((1000,), 8, 1).astype(np.float64)
    lb2 = (syn.uniform((1000,), 8, 2) < 0.2).astype(np.float64)
    out["auc2_scores"], out["auc2_labels"] = sc2, lb2
    out["auc2_value"] = np.float64(ref_eval(list(sc2), list(lb2), None))
    # parameter counts at full width
    e = RefEncoder(3, 8, 256, 256, 2048, 4096, MHA_layerNorm=True, FFN_layerNorm=True, weight_init=False,
                   relative_pe=True, window_size=4, window_depth=3)
    out["ltn_param_count"] = np.int64(sum(p.numel() for p in e.parameters()))
    out["ltn_state_keys"] = np.array(list(e.state_dict().keys()))
    e = RefEncoder(3, 8, 256, 256, 2048, 3027, FFN_layerNorm=True, weight_init=False)
    out["stn_param_count"] = np.int64(sum(p.numel() for p in e.parameters()))
    out["stn_state_keys"] = np.array(list(e.state_dict().keys()))
    out["classifier_param_count"] = np.int64(sum(p.numel() for p in RefClassifier(2048).parameters()))
    out["regressor_param_count"] = np.int64(sum(p.numel() for p in RefRegressor(2048).parameters()))
    out["classifier_state_keys"] = np.array(list(RefClassifier(8).state_dict().keys()))
    out["regressor_state_keys"] = np.array(list(RefRegressor(8).state_dict().keys()))
    np.savez_compressed(os.path.join(OUT_DIR, "misc.npz"), **out)
    print("misc: auc", out["auc_value"], out["auc2_value"], "ltn params", out["ltn_param_count"],
          "stn params", out["stn_param_count"])


if __name__ == "__main__":
    if "--out" in sys.argv:
        OUT_DIR = sys.argv[sys.argv.index("--out") + 1]
    only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
    for i, (name, (mode, ekw, skw)) in enumerate(CASES.items()):
        if only is None or name in only:
            run_case(name, mode, ekw, skw, seed=11 + i)
    if only is None or "misc" in only:
        misc()
    if "--skip-full-width" not in sys.argv:
        for name, (mode, ekw, skw, seed) in list(FULL_CASES.items()) + list(PACKED_CASES.items()):
            if only is None or name in only:
                # the 256-sequence cases sum two to four times as many tokens into every gradient: a wider band of recorded
                # ReLU-edge units (the summation-order noise of a pre-activation grows with the operand magnitudes it adds up)
                run_full_case(name, mode, ekw, skw, seed, relu_edge=3 * RELU_EDGE if name in PACKED_CASES else None)
    # the headline-size steps (minutes of CPU, tens of GB): only when named
    for name, (mode, ekw, skw, seed) in HEADLINE_CASES.items():
        if only is not None and name in only:
            run_full_case(name, mode, ekw, skw, seed, relu_edge=3 * RELU_EDGE)
