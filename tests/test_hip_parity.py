"""HIP path vs the oracle and vs the golden vectors captured from the real reference (GPU only).

Bars (written next to each check):
  * scores / encoder outputs: <= 1e-4 absolute, float32 — the tolerance BASELINE.json's north_star states;
  * gradients: <= 2e-4 of the tensor's max magnitude (+1e-7);
  * weights after two Adagrad steps: <= 5e-5 absolute on >= 99.9 % of the entries (Adagrad's first update is
    lr*sign(g): an entry whose gradient is at rounding level may legitimately flip).
Every call goes through the C ABI (lstc_vad_amd._lib -> liblstc_hip.so); nothing here reads /root/reference.
"""
import os

import numpy as np
import pytest
import torch

from cases import CASES
from oracle import lstc_oracle as orc
from util import GOLDEN, load_case, sub, oracle_cfgs, max_abs_diff

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _models(mode, ekw, d_model, dropout=0.0, head_dropout=0.0):
    from lstc_vad_amd.models import Encoder, Regressor, Classifier
    enc = Encoder(n_layers=3, MHA_attn_dropout=dropout, MHA_fc_dropout=dropout, FFN_dropout=dropout,
                  position_dropout=dropout, weight_init=False, **ekw)
    head = Classifier(d_model, head_dropout, weight_init=False) if mode == "LTN" else Regressor(d_model, head_dropout, weight_init=False)
    return enc, head


def _args(mode, skw):
    from argparse import Namespace
    return Namespace(batch_size=skw["batch_size"], part_num=skw["part_num"], part_len=skw["part_len"],
                     n_patch=skw["n_patch"], lambda_1=0.01, lambda_MIL=1.0, lambda_CE=0.8, lambda_BCE=1.0,
                     lambda_normal=0.2, lambda_abnormal=2.0, temporal_only=skw.get("temporal_only", False),
                     clip_grad=skw.get("clip_grad", False))


def _step(enc, head, mode, args, nf, af, al, d, cls_only=False):
    """One forward + loss (+ backward by the caller), mirroring the reference loops' reshapes.  ``cls_only`` uses
    Encoder.forward_cls (last layer evaluated for the CLS token only), which is what lstc_vad_amd.engine runs."""
    from lstc_vad_amd.losses import training_loss
    bs, pn, L, P = args.batch_size, args.part_num, args.part_len, args.n_patch
    if mode == "LTN":
        feats = torch.cat([nf.view(bs * pn, L * P, d), af.view(bs * pn, L * P, d)], 0)
    else:
        feats = torch.cat([nf.view(bs * pn * L, P, d), af.view(bs * pn * L, P, d)], 0)
    if cls_only:
        cls = enc.forward_cls(feats)
        enc_out = cls.unsqueeze(1)
    else:
        enc_out = enc(feats)
        cls = enc_out[:, 0, :]
    outputs = head(cls)
    loss, scalars = training_loss(args, mode, outputs, al)
    return enc_out, outputs, loss, scalars


@pytest.mark.parametrize("cls_only", [False, True])
@pytest.mark.parametrize("name", list(CASES))
def test_training_step_matches_reference_golden(name, cls_only):
    from lstc_vad_amd.optim import Adagrad, clip_grad_norm_
    z, mode, ekw, skw = load_case(name)
    d = ekw["d_model"]
    enc, head = _models(mode, ekw, d)
    enc.load_state_dict({k: v for k, v in sub(z, "enc_init.").items()}, strict=True)     # reference key names
    head.load_state_dict(sub(z, "head_init."), strict=True)
    enc, head = enc.to(DEV).train(), head.to(DEV).train()
    args = _args(mode, skw)
    nf, af, al = (torch.from_numpy(z[k]).to(DEV) for k in ("norm_feats", "abnorm_feats", "abnorm_labs"))
    opt = Adagrad([{"params": enc.parameters(), "lr": 1e-4}, {"params": head.parameters(), "lr": 1e-2}],
                  weight_decay=1e-3)
    for step in range(2):
        enc_out, outputs, loss, sc = _step(enc, head, mode, args, nf, af, al, d, cls_only)
        opt.zero_grad()
        loss.backward()
        if args.clip_grad:
            clip_grad_norm_(enc.parameters(), 10)
            clip_grad_norm_(head.parameters(), 10)
        if step == 0:
            ref_enc = z["enc_out"][:, :1, :] if cls_only else z["enc_out"]
            assert max_abs_diff(enc_out, ref_enc) < 1e-4                            # north_star tolerance
            assert max_abs_diff(outputs.reshape(z["outputs"].shape), z["outputs"]) < 1e-4
            assert np.max(np.abs(sc.cpu().double().numpy() - z["scalars"])) < 2e-5
            ref_g = sub(z, "enc_grad.")
            got = {k for k, p in enc.named_parameters() if p.grad is not None}
            assert got == set(ref_g), got ^ set(ref_g)        # unused LayerNorms must stay grad-less
            for k, p in list(enc.named_parameters()) + list(head.named_parameters()):
                g = ref_g.get(k) if k in ref_g else sub(z, "head_grad.").get(k)
                if g is None:
                    continue
                tol = 2e-4 * float(g.abs().max()) + 1e-7
                assert max_abs_diff(p.grad, g) < tol, (k, max_abs_diff(p.grad, g), tol)
        else:
            assert np.max(np.abs(sc.cpu().double().numpy() - z["scalars_step2"])) < 1e-4
        opt.step()
    for prefix, mod in (("enc_after2.", enc), ("head_after2.", head)):
        ref = sub(z, prefix)
        for k, v in mod.state_dict().items():
            if not v.is_floating_point():
                continue
            diff = (v.cpu() - ref[k]).abs()
            frac_bad = float((diff > 5e-5).float().mean())
            lim = 1e-3 if prefix.startswith("enc") else 1e-2
            assert frac_bad <= lim, (k, frac_bad, float(diff.max()))
            # and no entry may be off by more than what two sign-flipped Adagrad steps can move it (2 * 2 * lr): an
            # unwritten partial or a corrupted slice would show up here whatever its share of the tensor
            lr = 1e-4 if prefix.startswith("enc") else 1e-2
            assert float(diff.max()) <= 4 * lr + 1e-6, (k, float(diff.max()))


def test_unaligned_hidden_width_matches_golden_padded_and_unpadded(monkeypatch):
    """n_hidden = 47 (the reduced stand-in of the reference's STN n_hidden = 3027): the FFN block normally runs at the padded
    width 48 (zero rows / columns appended to W1, b1, W2 - functional._padded_hidden); with the padding off the same step
    goes through the scalar-load GEMM instantiations at the true width.  Both match the reference golden (the padded arm
    is the parametrized test above); the two arms agree with each other to rounding, gradients included."""
    from lstc_vad_amd import functional as Fn
    assert Fn._padded_hidden(47) == 48 and Fn._padded_hidden(3027) == 3072 and Fn._padded_hidden(4096) == 4096
    assert Fn._padded_hidden(40) == 40 and Fn._padded_hidden(300) == 300 and Fn._padded_hidden(250) == 256
    grads = {}
    for pad in (True, False):
        monkeypatch.setattr(Fn, "_PAD_HIDDEN", pad)
        test_training_step_matches_reference_golden("stn_sht", False)
        z, mode, ekw, skw = load_case("stn_sht")
        d = ekw["d_model"]
        enc, head = _models(mode, ekw, d)
        enc.load_state_dict(sub(z, "enc_init."), strict=True)
        head.load_state_dict(sub(z, "head_init."), strict=True)
        enc, head = enc.to(DEV).train(), head.to(DEV).train()
        nf, af, al = (torch.from_numpy(z[k]).to(DEV) for k in ("norm_feats", "abnorm_feats", "abnorm_labs"))
        _, _, loss, _ = _step(enc, head, mode, _args(mode, skw), nf, af, al, d, False)
        loss.backward()
        grads[pad] = {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}
    assert set(grads[True]) == set(grads[False])
    for k in grads[True]:
        assert grads[True][k].shape == grads[False][k].shape, k
        tol = 1e-5 * float(grads[False][k].abs().max()) + 1e-8
        assert max_abs_diff(grads[True][k], grads[False][k]) <= tol, (k, max_abs_diff(grads[True][k], grads[False][k]), tol)


@pytest.mark.parametrize("rows", [1, 2, 3, 4, 7, 12, 13])
def test_colsum_of_a_few_rows_keeps_the_two_pass_summation_order(rows):
    """lstc_colsum on the partial products of a split-K weight gradient (2-8 rows of out*in columns) runs ONE pass
    (colsum_few) with the order of the two-pass form: row p into accumulator p & 3 in increasing p, then (a0 + a1) + (a2 + a3) -
    restated here with elementwise torch adds (exact IEEE, so equality is bitwise).  13 rows take the two-pass kernels, whose
    4-way unrolled accumulation is a different order: compared against f64."""
    from lstc_vad_amd import functional as Fn
    g = torch.Generator(device=DEV).manual_seed(rows)
    cols = 2048 * 6 + 8
    x = torch.randn(rows, cols, device=DEV, generator=g) * 3
    base = torch.randn(cols, device=DEV, generator=g)
    got = Fn.colsum(x)
    acc = Fn.colsum(x, out=base.clone(), accumulate=True)
    torch.cuda.synchronize()
    if rows > 12:
        ref = x.double().sum(0)
        assert float((got.double() - ref).abs().max()) < 1e-5 and float((acc.double() - (ref + base.double())).abs().max()) < 1e-5
        return
    a = [torch.zeros(cols, device=DEV) for _ in range(4)]
    for p in range(rows):
        a[p & 3] = a[p & 3] + x[p]
    want = (a[0] + a[1]) + (a[2] + a[3])
    assert torch.equal(got, want)
    assert torch.equal(acc, base + want)
    odd = torch.randn(rows, 1001, device=DEV, generator=g)           # columns not a multiple of 4: two-pass path, same order here
    a = [torch.zeros(1001, device=DEV) for _ in range(4)]
    for p in range(rows):
        a[p & 3] = a[p & 3] + odd[p]
    assert torch.equal(Fn.colsum(odd), (a[0] + a[1]) + (a[2] + a[3]))


def test_multi_tensor_adagrad_is_bitwise_the_per_tensor_kernel():
    """lstc_adagrad_multi (every parameter of a step in one launch, items in the kernel arguments, 48 per launch) against
    lstc_adagrad_step tensor by tensor: weights and accumulators bit-identical - 60 tensors (two launches), sizes that are
    not multiples of 4, of the per-workgroup slice (8192) or of anything, one unaligned view, per-item lr / decay / scale."""
    import ctypes as C
    from lstc_vad_amd import _lib
    from lstc_vad_amd._lib import AdagradItem, check, dev_ptr, stream_ptr
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(5)
    sizes = [1, 3, 4, 5, 17, 1000, 8192, 8193, 16384, 70001, 2048 * 2048 + 3, 512 * 2048] + [257 + 13 * i for i in range(48)]
    ws, gs, ss = [], [], []
    for i, n in enumerate(sizes):
        off = 1 if i == 7 else 0                                    # one tensor whose storage is only 4-B aligned
        w = torch.randn(n + off, device=DEV, generator=g)[off:]
        ws.append(w); gs.append(torch.randn(n, device=DEV, generator=g)); ss.append(torch.rand(n, device=DEV, generator=g))
    hyper = [(1e-2 * (1 + i % 3), 1e-3 * (i % 2), 1e-10, 1.0 if i % 5 else 0.37) for i in range(len(sizes))]
    w1, s1 = [w.clone() for w in ws], [s.clone() for s in ss]
    for w, gr, st, (lr, wd, eps, sc) in zip(w1, gs, s1, hyper):
        check(lib.lstc_adagrad_step(dev_ptr(w), dev_ptr(gr), dev_ptr(st), w.numel(), lr, wd, eps, sc, stream_ptr()), "step")
    w2, s2 = [w.clone() for w in ws], [s.clone() for s in ss]
    w2[7] = torch.cat([torch.zeros(1, device=DEV), ws[7]])[1:]      # keep the unaligned case unaligned after the clone
    arr = (AdagradItem * len(sizes))(*[(dev_ptr(w), dev_ptr(gr), dev_ptr(st), w.numel(), lr, wd, eps, sc)
                                       for w, gr, st, (lr, wd, eps, sc) in zip(w2, gs, s2, hyper)])
    check(lib.lstc_adagrad_multi(arr, len(sizes), stream_ptr()), "multi")
    torch.cuda.synchronize()
    for i in range(len(sizes)):
        assert torch.equal(w1[i], w2[i]) and torch.equal(s1[i], s2[i]), (i, sizes[i])
        assert not torch.equal(w1[i], ws[i])


@pytest.mark.parametrize("name", ["ltn_sht", "stn_sht", "ltn_ubnormal"])
def test_eval_mode_and_short_tail(name):
    z, mode, ekw, skw = load_case(name)
    d = ekw["d_model"]
    enc, _ = _models(mode, ekw, d, dropout=0.3)
    enc.load_state_dict(sub(z, "enc_init."), strict=True)
    enc = enc.to(DEV).eval()
    bs, pn, L, P = skw["batch_size"], skw["part_num"], skw["part_len"], skw["n_patch"]
    nf = torch.from_numpy(z["norm_feats"]).to(DEV)
    x = nf.view(bs * pn, L * P, d)[:3] if mode == "LTN" else nf.view(bs * pn * L, P, d)[:3]
    with torch.no_grad():
        assert max_abs_diff(enc(x), z["eval_enc_out"]) < 1e-4
        if "eval_tail_enc_out" in z.files:     # last part of a video: fewer clips -> shorter sequence
            assert max_abs_diff(enc(x[:, :(L - 1) * P].contiguous()), z["eval_tail_enc_out"]) < 1e-4
            assert max_abs_diff(enc(x[:, :P].contiguous()), z["eval_tail1_enc_out"]) < 1e-4


@pytest.mark.parametrize("cls_only", [False, True])
@pytest.mark.parametrize("name", ["ltn_sht", "stn_mil_ce", "stn_relpe2d_extras"])
def test_dropout_run_replays_through_oracle(name, cls_only):
    """Training with every dropout ON: the masks of the HIP run are exported (lstc_dropout_mask) and injected
    into the oracle, which must then reproduce loss and gradients."""
    from lstc_vad_amd import functional as Fn
    z, mode, ekw, skw = load_case(name)
    d = ekw["d_model"]
    p, ph = 0.25, 0.5
    enc, head = _models(mode, ekw, d, dropout=p, head_dropout=ph)
    enc.load_state_dict(sub(z, "enc_init."), strict=True)
    head.load_state_dict(sub(z, "head_init."), strict=True)
    enc, head = enc.to(DEV).train(), head.to(DEV).train()
    args = _args(mode, skw)
    nf, af, al = (torch.from_numpy(z[k]).to(DEV) for k in ("norm_feats", "abnorm_feats", "abnorm_labs"))
    torch.manual_seed(1234)
    Fn.reset_rng()
    with Fn.record_dropout() as sites:
        enc_out, outputs, loss, sc = _step(enc, head, mode, args, nf, af, al, d, cls_only)
    loss.backward()
    masks, keeps = {}, []
    for site, pp, seed, shape in sites:
        m = Fn.dropout_mask(shape, pp, seed, DEV).cpu()
        if "classifier" not in site and "regressor" not in site:
            keeps.append(float(m.float().mean()))
        if cls_only and (site.endswith("dropout#cls") and len(shape) == 2 or (site.startswith("layer_stack.2.pos_ffn") and len(shape) == 2)):
            # CLS-only last layer: the HIP mask covers token 0; the other tokens of that layer are never read
            S_full = 1 + (args.part_len * args.n_patch if mode == "LTN" else args.n_patch)
            full = torch.ones(shape[0], S_full, shape[1], dtype=torch.uint8)
            full[:, 0, :] = m
            m = full
        masks[site.replace("#cls", "")] = m
    assert len(masks) == len(sites) and len(sites) >= 3 * 3 + 2
    assert abs(np.mean(keeps) - (1 - p)) < 0.02                                   # keep-rate of the counter-based RNG
    ecfg, st = oracle_cfgs(mode, ekw, skw, dropout=p)
    st.head_dropout = ph
    enc_P = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sub(z, "enc_init.").items()}
    head_P = {k: v.clone().requires_grad_(True) for k, v in sub(z, "head_init.").items()}
    out = orc.forward_loss(enc_P, head_P, ecfg, st, nf.cpu(), af.cpu(), al.cpu(), training=True, masks=masks)
    out["loss"].backward()
    assert max_abs_diff(outputs.reshape(out["outputs"].shape), out["outputs"]) < 1e-4
    assert abs(float(sc[0]) - float(out["loss"].detach())) < 2e-5
    for k, pr in enc.named_parameters():
        if pr.grad is None:
            assert enc_P[k].grad is None, k
            continue
        g = enc_P[k].grad
        assert max_abs_diff(pr.grad, g) < 2e-4 * float(g.abs().max()) + 1e-7, k
    for k, pr in head.named_parameters():
        g = head_P[k].grad
        assert max_abs_diff(pr.grad, g) < 2e-4 * float(g.abs().max()) + 1e-7, k


def test_reference_named_loss_functions():
    from argparse import Namespace
    from lstc_vad_amd.losses import get_MIL_loss, get_CE_loss, get_BCE_loss
    torch.manual_seed(0)
    bs, pn, L = 3, 5, 2
    args = Namespace(batch_size=bs, part_num=pn, part_len=L, lambda_1=0.01, lambda_normal=0.2, lambda_abnormal=2.0)
    y3 = torch.rand(2 * bs, pn * L, 1)
    yflat = torch.rand(2 * bs * pn)
    ycol = torch.rand(2 * bs * pn * L, 1)
    for y, pl, oracle_L in ((y3, None, L), (yflat, None, 1), (ycol, L, L)):
        yg = y.to(DEV).requires_grad_(True)
        loss, err, l1 = get_MIL_loss(args, yg, pl) if pl else get_MIL_loss(args, yg)
        loss.backward()
        yo = y.clone().requires_grad_(True)
        lo, eo, l1o = orc.mil_loss(yo, bs, pn, oracle_L, 0.01)
        lo.backward()
        assert abs(float(loss) - float(lo)) < 1e-6 and abs(float(err) - float(eo)) < 1e-6 and abs(float(l1) - float(l1o)) < 1e-6
        assert max_abs_diff(yg.grad, yo.grad) < 1e-7
    p = torch.softmax(torch.randn(2 * bs * pn, 2), -1)
    t = torch.rand(2 * bs * pn, 1)
    t = torch.cat([1 - t, t], 1)
    pg = p.to(DEV).requires_grad_(True)
    ce = get_CE_loss(args, pg, t.to(DEV))
    ce.backward()
    po = p.clone().requires_grad_(True)
    ceo = orc.ce_loss(po, t)
    ceo.backward()
    assert abs(float(ce) - float(ceo)) < 1e-6 and max_abs_diff(pg.grad, po.grad) < 1e-7
    o = torch.rand(2 * bs, pn) * 0.98 + 0.01
    tt = t.view(2 * bs, pn, 2)
    og = o.to(DEV).requires_grad_(True)
    b = get_BCE_loss(args, og, tt.to(DEV))
    b.backward()
    oo = o.clone().requires_grad_(True)
    bo = orc.bce_loss(oo, tt, 0.2, 2.0)
    bo.backward()
    assert abs(float(b) - float(bo)) < 1e-5 and max_abs_diff(og.grad, oo.grad) < 1e-5 * float(oo.grad.abs().max())
    # the same functions under the reference's import paths and per-script signatures (Train/*.py shims)
    import Train.spatio_transformer_shanghaitech as m_stn
    import Train.temporal_transformer_shanghaitech as m_ltn
    import Train.spatio_transformer_MIL_CE as m_co
    for fn, y, oracle_L in ((lambda y_: m_stn.get_MIL_loss(args, y_), y3, L), (lambda y_: m_ltn.get_MIL_loss(args, y_), yflat, 1),
                            (lambda y_: m_co.get_MIL_loss(args, y_, L), ycol, L)):
        loss, err, l1 = fn(y.to(DEV))
        lo, eo, l1o = orc.mil_loss(y, bs, pn, oracle_L, 0.01)
        assert abs(float(loss) - float(lo)) < 1e-6 and abs(float(err) - float(eo)) < 1e-6 and abs(float(l1) - float(l1o)) < 1e-6
    assert abs(float(m_ltn.get_CE_loss(args, p.to(DEV), t.to(DEV))) - float(ceo)) < 1e-6
    assert abs(float(m_co.get_CE_loss(args, p.to(DEV), t.to(DEV))) - float(ceo)) < 1e-6
    assert abs(float(m_co.get_BCE_loss(args, o.to(DEV), tt.to(DEV))) - float(bo)) < 1e-5


def test_sharded_loss_equals_global_loss():
    """Data-parallel bookkeeping of lstc_vad_loss without a process group: two 'ranks' on one GPU exchange
    their bag maxima by hand; the summed contributions and the concatenated gradients must equal the
    single-rank result."""
    import ctypes as C
    from lstc_vad_amd import _lib
    from lstc_vad_amd._lib import LossDesc, check, dev_ptr
    lib = _lib.load()
    torch.manual_seed(3)
    bs, pn, L = 4, 3, 2
    for mode, c, Ls in ((1, 2, 1), (0, 1, L), (2, 1, L)):
        rpv = pn * Ls
        out = torch.rand(2 * bs * rpv, c, device=DEV)
        if c == 2:
            out = torch.softmax(torch.randn(2 * bs * rpv, 2, device=DEV), -1)
        labs = torch.rand(bs, pn * L, device=DEV)
        skip = bs if mode != 0 else bs * rpv

        def run(o, lab, bs_l, off, phase, bag):
            d = LossDesc()
            d.mode, d.bs_global, d.bs_local, d.rank_off = mode, bs, bs_l, off
            d.part_num, d.score_len, d.label_len, d.l1_skip = pn, Ls, L, skip
            d.lambda_1, d.lambda_MIL, d.lambda_aux, d.lambda_normal, d.lambda_abnormal = 0.01, 1.0, 0.8, 0.2, 2.0
            dout, sc = torch.zeros_like(o), torch.zeros(5, device=DEV)
            d.out, d.abn_labels, d.bag, d.dout, d.scalars = dev_ptr(o), dev_ptr(lab) if mode else None, dev_ptr(bag), dev_ptr(dout), dev_ptr(sc)
            d.phase = phase
            check(lib.lstc_vad_loss(C.byref(d), None))
            torch.cuda.synchronize()
            return dout, sc

        bag1 = torch.zeros(2 * bs, device=DEV)
        g_full, s_full = run(out.contiguous(), labs, bs, 0, 2, bag1)
        half = bs // 2
        nor, abn = out[: bs * rpv], out[bs * rpv:]
        bag = torch.zeros(2 * bs, device=DEV)
        shards = []
        for r in range(2):
            o_r = torch.cat([nor[r * half * rpv:(r + 1) * half * rpv], abn[r * half * rpv:(r + 1) * half * rpv]]).contiguous()
            shards.append((o_r, labs[r * half:(r + 1) * half].contiguous()))
            run(o_r, shards[-1][1], half, r * half, 0, bag)          # phase 0 fills this rank's slots ("all-reduce")
        assert max_abs_diff(bag, bag1) == 0.0
        tot = torch.zeros(5, device=DEV)
        gn, ga = [], []
        for r in range(2):
            g, s = run(shards[r][0], shards[r][1], half, r * half, 1, bag)
            tot += s
            gn.append(g[: half * rpv]); ga.append(g[half * rpv:])
        assert max_abs_diff(tot, s_full) < 2e-6
        assert max_abs_diff(torch.cat(gn + ga), g_full) < 1e-7


def test_full_width_scores_match_oracle():
    """d_model=2048, 8 heads x 256, F=4096, rel-PE, S=49 (the headline LTN layer shapes) on a small batch:
    anomaly scores within 1e-4 of the oracle, and batch-invariance of the HIP path (a sequence's output does
    not depend on which other sequences share the launch)."""
    from lstc_vad_amd.models import Encoder, Classifier
    torch.manual_seed(0)
    ekw = dict(n_layers=2, n_head=8, d_k=256, d_v=256, d_model=2048, d_inner=4096, MHA_layerNorm=True,
               FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=3)
    enc = Encoder(MHA_attn_dropout=0.0, MHA_fc_dropout=0.0, FFN_dropout=0.0, weight_init=True, **ekw)
    head = Classifier(2048, 0.0)
    x = (0.5 * torch.relu(torch.randn(6, 48, 2048)))
    P = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    HP = {k: v.detach().clone() for k, v in head.state_dict().items()}
    ecfg = orc.EncoderCfg(MHA_attn_dropout=0.0, MHA_fc_dropout=0.0, FFN_dropout=0.0, **ekw)
    with torch.no_grad():
        ref_enc = orc.encoder_forward(P, x, ecfg, False)
        ref_score = orc.head_forward(HP, ref_enc[:, 0, :], "classifier")
    enc, head = enc.to(DEV).eval(), head.to(DEV).eval()
    with torch.no_grad():
        got_enc = enc(x.to(DEV))
        got_score = head(got_enc[:, 0, :])
        assert max_abs_diff(got_score, ref_score) < 1e-4            # north_star: scores within 1e-4 fp32
        assert max_abs_diff(got_enc, ref_enc) < 5e-4                # post-LN activations are O(1..5)
        big = torch.cat([x.to(DEV), 0.5 * torch.relu(torch.randn(250, 48, 2048, device=DEV))], 0)
        got_big = enc(big)
        from lstc_vad_amd import functional as Fn
        if Fn.get_compute_dtype() == "fp32":
            assert torch.equal(got_big[:6], got_enc)                # bit-exact batch invariance
        else:       # f32x3: the per-tensor power-of-two scale depends on the batch, so invariance holds to f32 rounding
            assert max_abs_diff(got_big[:6], got_enc) < 5e-6


@pytest.mark.parametrize("name", ["ltn_sht", "stn_sht", "ltn_ucf", "stn_relpe2d_extras"])
def test_forward_cls_equals_full_forward_row0(name):
    """Encoder.forward_cls (last layer on the CLS query only) == Encoder.forward(...)[:, 0, :] to rounding."""
    z, mode, ekw, skw = load_case(name)
    d = ekw["d_model"]
    enc, _ = _models(mode, ekw, d)
    enc.load_state_dict(sub(z, "enc_init."), strict=True)
    enc = enc.to(DEV).eval()
    bs, pn, L, P = skw["batch_size"], skw["part_num"], skw["part_len"], skw["n_patch"]
    nf = torch.from_numpy(z["norm_feats"]).to(DEV)
    x = nf.view(bs * pn, L * P, d) if mode == "LTN" else nf.view(bs * pn * L, P, d)
    with torch.no_grad():
        full = enc(x)
        cls = enc.forward_cls(x)
    assert cls.shape == (x.shape[0], d)
    assert max_abs_diff(cls, full[:, 0, :]) < 5e-6


@pytest.mark.parametrize("name,dtype,fuse", [("ltn_sht", "fp32", "off"), ("ltn_sht", "fp32", "on"), ("stn_mil_ce", "fp32", "off"),
                                             ("ltn_ubnormal_dk32", "bf16p", "on")])
def test_rccl_gradient_bucket_path_single_rank(name, dtype, fuse):
    """The data-parallel machinery (RCCL process group, flat gradient buckets, weight-gradient kernels writing STRAIGHT into the
    bucket through functional.grad_sink - no fill, no autograd accumulate for the large weights -, async all-reduce launched when a
    bucket's last gradient has been issued, Adagrad stepping from the bucket views) on ONE GPU must reproduce the plain path
    BIT FOR BIT: same kernels, same arithmetic, only the destination of the gradients differs.  Separate and fused Q|K|V weight
    gradients, the padded FFN hidden (n_hidden = 47 -> 48: the real rows are copied out of the wide gradient), bf16 packed GEMMs."""
    import os
    import socket
    import torch.distributed as dist
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.engine import TrainStep
    z, mode, ekw, skw = load_case(name)
    d = ekw["d_model"]
    args = _args(mode, skw)
    nf, af, al = (torch.from_numpy(z[k]).to(DEV) for k in ("norm_feats", "abnorm_feats", "abnorm_labs"))

    def run(forced, bucket_steps=True):
        enc, head = _models(mode, ekw, d)
        enc.load_state_dict(sub(z, "enc_init."), strict=True)
        head.load_state_dict(sub(z, "head_init."), strict=True)
        enc, head = enc.to(DEV).train(), head.to(DEV).train()
        os.environ["LSTC_FORCE_DIST"] = "1" if forced else "0"
        ts = TrainStep(args, mode, enc, head, 1e-4, 1e-2, 1e-3, fuse_qkv=fuse)
        assert (ts.reducer is not None) == forced
        ts.bucket_steps = bucket_steps
        if forced:
            sunk = [p for p in list(enc.parameters()) + list(head.parameters()) if Fn.grad_sink(p) is not None]
            assert len(sunk) == 3 * 6 + 1                          # 4 projections + 2 FFN matrices per layer, the head's first Linear
            for flat in ts.reducer.buckets:
                flat.fill_(float("nan"))                           # nothing may rely on a zero-filled direct region
        for _ in range(2):
            sc = ts.step(nf, af, al)
        if forced:
            assert all(torch.isfinite(b).all() for b in ts.reducer.buckets)
            # round 6: per-bucket optimizer steps on a side stream as each reduction lands (the default) vs all buckets, then one step
            assert (ts.reducer.landed is not None) == bucket_steps
        return sc.cpu(), {k: v.detach().cpu().clone() for k, v in list(enc.state_dict().items()) + list(head.state_dict().items())}

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    Fn.set_compute_dtype("bf16" if dtype == "bf16p" else dtype)
    if dtype == "bf16p":
        Fn.set_x3_threshold(0, 0, 0)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        try:
            sc1, w1 = run(True)
            sc2, w2 = run(True, bucket_steps=False)
        finally:
            dist.destroy_process_group()
            os.environ["LSTC_FORCE_DIST"] = "0"
        sc0, w0 = run(False)
    finally:
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()
    # (the two paths differ only in where the gradients live - flat buckets vs per-parameter tensors)
    assert torch.equal(sc0, sc1) and torch.equal(sc0, sc2), (sc0, sc1, sc2)
    for k in w0:
        assert torch.equal(w0[k], w1[k]) and torch.equal(w0[k], w2[k]), k
    if dtype == "fp32":
        assert abs(float(sc0[0]) - float(z["scalars_step2"][0])) < 1e-4


def test_bf16_compute_mode_tracks_fp32_scores_and_auc():
    """bf16 GEMM mode (f32 storage, bf16 MFMA): scores stay close to the f32 path and the frame-level AUC of a
    synthetic video set is unchanged to 5e-3 (BASELINE.json config 5: "bf16 + fp32 AUC parity check")."""
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.metrics import roc_auc
    from lstc_vad_amd.models import Encoder, Classifier
    torch.manual_seed(0)
    ekw = dict(n_layers=3, n_head=4, d_k=64, d_v=64, d_model=256, d_inner=512, MHA_layerNorm=True, FFN_layerNorm=True,
               relative_pe=True, window_size=4, window_depth=3)
    enc = Encoder(weight_init=True, **ekw).to(DEV).eval()
    head = Classifier(256).to(DEV).eval()
    x = 0.5 * torch.relu(torch.randn(512, 48, 256, device=DEV))
    x[256:, :, :32] += 0.4                                   # "abnormal" half
    labels = np.r_[np.zeros(256), np.ones(256)]
    with torch.no_grad():
        s32 = head(enc.forward_cls(x))[:, 1].cpu().numpy()
        Fn.set_compute_dtype("bf16")
        try:
            s16 = head(enc.forward_cls(x))[:, 1].cpu().numpy()
        finally:
            Fn.set_compute_dtype("fp32")
    assert np.max(np.abs(s32 - s16)) < 2e-2                  # bf16 operands: ~3 significant digits
    assert np.max(np.abs(s32 - s16)) > 0                      # the mode really switched kernels
    assert abs(roc_auc(s32, labels) - roc_auc(s16, labels)) < 5e-3


@pytest.mark.parametrize("name", ["ltn_sht", "ltn_ubnormal_dk32"])
def test_bf16_compute_training_step_close_to_golden(name):
    """A full bf16-mode training step (loss + gradients) stays within bf16 operand precision of the reference golden
    (``ltn_ubnormal_dk32``: d_k = 32, S = 81 - the attention products of layers 0-1 run on the bf16 MFMA in the 8-wave staged
    kernels)."""
    from lstc_vad_amd import functional as Fn
    z, mode, ekw, skw = load_case(name)
    d = ekw["d_model"]
    enc, head = _models(mode, ekw, d)
    enc.load_state_dict(sub(z, "enc_init."), strict=True)
    head.load_state_dict(sub(z, "head_init."), strict=True)
    enc, head = enc.to(DEV).train(), head.to(DEV).train()
    args = _args(mode, skw)
    nf, af, al = (torch.from_numpy(z[k]).to(DEV) for k in ("norm_feats", "abnorm_feats", "abnorm_labs"))
    Fn.set_compute_dtype("bf16")
    try:
        enc_out, outputs, loss, sc = _step(enc, head, mode, args, nf, af, al, d, cls_only=True)
        loss.backward()
    finally:
        Fn.set_compute_dtype("fp32")
    assert abs(float(sc[0]) - float(z["scalars"][0])) < 2e-2
    ref_g = sub(z, "enc_grad.")
    for k, p in enc.named_parameters():
        if k in ref_g and ref_g[k].numel() > 64 and float(ref_g[k].norm()) > 0:    # (last bias table: zero gradient)
            g = ref_g[k]
            # direction of every large gradient tensor is preserved (bf16 operands perturb magnitudes by a few %,
            # more where the hinge / bag-max selections sit close to a tie on this tiny d_model=32 case)
            cos = float((p.grad.cpu() * g).sum() / (p.grad.cpu().norm() * g.norm() + 1e-20))
            assert cos > 0.9, (k, cos)


@pytest.mark.parametrize("d_inner,d_head", [(203, 128), (512, 64), (203, 64)])
def test_bf16_step_with_narrow_hidden_or_heads_at_default_thresholds(d_inner, d_head):
    """bf16 mode at the DEFAULT packing thresholds with d_model = 256 and enough tokens (17 408 = 68 x 256) that the LayerNorm
    backward emits its gradient as a packed operand, while the weight-gradient partner was never packed in the forward: the
    FFN hidden when n_hidden < 256 (203 is what tools/coteach_loop_synthetic.sh passes) and the attention output when
    H*d_v < 256.  wgrad must take the packed TR form with the partner packed on demand (it used to raise); gradients against
    the same step in fp32 mode."""
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd import synthetic as syn
    from cases import fill_params
    d, H = 256, 2
    ekw = dict(d_model=d, n_head=H, d_k=d_head, d_v=d_head, d_inner=d_inner, MHA_layerNorm=True, FFN_layerNorm=True)
    skw = dict(batch_size=2, part_num=16, part_len=16, n_patch=16)            # 1024 sequences x S = 17
    args = _args("STN", skw)
    nf, _, af, al = syn.training_batch(2, 16, 16, 16, d, seed=5, with_pseudo=True, threshold=0.6)
    nf, af, al = (torch.from_numpy(x).to(DEV) for x in (nf, af, al))
    grads = {}
    for mode in ("fp32", "bf16"):
        enc, head = _models("STN", dict(ekw), d)
        fill_params(enc, 51); fill_params(head, 52)
        enc, head = enc.to(DEV).train(), head.to(DEV).train()
        Fn.set_compute_dtype(mode)
        try:
            assert mode == "fp32" or Fn._fused_pack_shape(1024 * 17, d)
            _, outputs, loss, sc = _step(enc, head, "STN", args, nf, af, al, d, cls_only=False)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            Fn.set_compute_dtype("fp32")
        grads[mode] = ({k: p.grad.detach().clone() for k, p in enc.named_parameters() if p.grad is not None}, float(sc[0]))
    assert abs(grads["bf16"][1] - grads["fp32"][1]) < 2e-2
    assert set(grads["bf16"][0]) == set(grads["fp32"][0])
    for k, g in grads["fp32"][0].items():
        if g.numel() < 4096 or float(g.norm()) == 0.0:
            continue
        b = grads["bf16"][0][k]
        cos = float((g.double() * b.double()).sum() / (g.double().norm() * b.double().norm() + 1e-30))
        assert torch.isfinite(b).all() and cos > 0.97, (k, cos)


@pytest.mark.parametrize("name,mode_", [("ltn_sht", "fp32"), ("stn_mil_ce", "fp32"), ("ltn_ubnormal_dk32", "bf16"), ("ltn_sht", "bf16p"),
                                        ("ltn_temporal_only_clip", "fp32")])
def test_graphed_step_is_bitwise_the_eager_step(name, mode_):
    """engine.GraphedStep: the whole training step captured into one HIP graph and replayed.  Four replays with dropout ON
    (three different batches) against four eager TrainStep.step calls from the same weights and seed counter: the five
    scalars of every step and every weight after the last one are bit-identical - the device-side seed word
    (lstc_dropout_seed_device) gives replay k the masks of eager step k, forward and backward.  ``bf16p`` forces the packed
    bf16 GEMM (its hand-scheduled epilogue draws the dropout mask too) on the reduced case."""
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.engine import GraphedStep, TrainStep
    z, mode, ekw, skw = load_case(name)
    d = ekw["d_model"]
    args = _args(mode, skw)
    g = torch.Generator().manual_seed(5)
    batches = []
    for i in range(3):
        nf = torch.from_numpy(z["norm_feats"]) * (1.0 + 0.1 * i) + 0.01 * torch.randn(z["norm_feats"].shape, generator=g).abs()
        af = torch.from_numpy(z["abnorm_feats"]) * (1.0 - 0.05 * i)
        batches.append((nf.to(DEV), af.to(DEV), torch.from_numpy(z["abnorm_labs"]).to(DEV)))
    res = {}
    Fn.set_compute_dtype("bf16" if mode_.startswith("bf16") else mode_)
    if mode_ == "bf16p":
        Fn.set_x3_threshold(0, 0, 0)
    try:
        for how in ("eager", "graph"):
            enc, head = _models(mode, dict(ekw), d, dropout=0.2, head_dropout=0.3)
            enc.load_state_dict(sub(z, "enc_init."), strict=True)
            head.load_state_dict(sub(z, "head_init."), strict=True)
            enc, head = enc.to(DEV).train(), head.to(DEV).train()
            ts = TrainStep(args, mode, enc, head, 1e-3, 1e-2, 1e-3, fuse_qkv="off")
            Fn.reset_rng(11)
            stepper = ts if how == "eager" else GraphedStep(ts, *batches[0])
            assert Fn._counter == 11
            scs = []
            for i in range(4):
                scs.append(stepper.step(*batches[i % 3]).clone())
                if i == 1:
                    Fn.next_seed()                     # an eager draw between two steps (both arms): later masks move with it
            torch.cuda.synchronize()
            steps_n = [ts.optimizer.state[p]["step"] for grp in ts.optimizer.param_groups for p in grp["params"]]
            res[how] = (scs, {k: p.detach().clone() for k, p in list(enc.named_parameters()) + list(head.named_parameters())}, Fn._counter, steps_n)
            if how == "graph":
                assert stepper.seeds_per_step >= 6
    finally:
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()
    assert res["eager"][2] == res["graph"][2]                                   # same number of seeds consumed
    assert res["eager"][3] == res["graph"][3] and set(res["eager"][3]) <= {0, 4}  # Adagrad step counts: 4, or 0 for grad-less LayerNorms
    for a, b in zip(res["eager"][0], res["graph"][0]):
        assert torch.equal(a, b), (a, b)
    assert not torch.equal(res["eager"][0][0], res["eager"][0][3])              # (the steps do differ from one another)
    for k, w in res["eager"][1].items():
        assert torch.equal(w, res["graph"][1][k]), k


def test_dropout_mask_is_the_host_restatement_bit_for_bit():
    """lstc_dropout_mask (and with it every fused dropout site: they share drop_key / drop_hash) against the numpy restatement
    of the rule in tests/util.py, for host-derived keys and for keys re-derived ON THE DEVICE from seed + word
    (lstc_dropout_seed_device, the captured-step path): consecutive seeds, several rates."""
    from lstc_vad_amd import _lib
    from lstc_vad_amd import functional as Fn
    from util import host_dropout_keep
    n = 100003
    i = np.arange(n)
    for seed, p in ((0x1234567890ABCDEF, 0.2), (0x1234567890ABCDF0, 0.2), (7, 0.6), ((1 << 64) - 3, 0.05)):
        m = Fn.dropout_mask((n,), p, seed, DEV).cpu().numpy()
        assert np.array_equal(m != 0, host_dropout_keep(i, p, seed)), (seed, p)
    word = torch.tensor([5], device=DEV, dtype=torch.int64)
    lib = _lib.load()
    _lib.check(lib.lstc_dropout_seed_device(word.data_ptr()), "lstc_dropout_seed_device")
    try:
        m = Fn.dropout_mask((n,), 0.3, 1000, DEV).cpu().numpy()
    finally:
        _lib.check(lib.lstc_dropout_seed_device(None), "lstc_dropout_seed_device")
    assert np.array_equal(m != 0, host_dropout_keep(i, 0.3, 1005))


def test_batched_colsum_is_bitwise_the_per_plane_colsum():
    """lstc_colsum_batched (the LayerNorm backward's dgamma / dbeta / dbias partial planes in ONE two-pass reduction instead of
    three) against lstc_colsum plane by plane: bit-identical for the two-pass sizes (768 / 1024 / 333 partial rows, d = 2048 /
    1024 / 40), through the one-pass small-row form (8 rows), for a prefix of the planes, and against an f64 sum."""
    from lstc_vad_amd import functional as Fn
    g = torch.Generator(device=DEV).manual_seed(12)
    for planes, rows, cols in ((3, 768, 2048), (2, 1024, 1024), (3, 333, 40), (3, 8, 512), (3, 13, 64)):
        x = torch.randn(planes, rows, cols, device=DEV, generator=g)
        for n in (planes, 2):
            got = Fn.colsum_planes(x, n)
            assert got.shape == (n, cols)
            for b in range(n):
                assert torch.equal(got[b], Fn.colsum(x[b])), (planes, rows, cols, b)
            assert max_abs_diff(got, x[:n].double().sum(1)) < 2e-5 * (rows ** 0.5)


def test_clip_grad_norm_matches_torch_in_two_launches_without_host_sync():
    """optim.clip_grad_norm_ = lstc_sqnorm_multi + lstc_clip_scale_multi (Train/temporal_transformer_shanghaitech.py:139-141):
    a list of 60 gradient tensors (more than one 48-item launch; odd sizes, a 4-byte-aligned view) against
    torch.nn.utils.clip_grad_norm_ on CPU copies - clipped when the norm exceeds max_norm, bit-untouched when it does not,
    the returned norm equal to torch's, two runs bit-identical (fixed-order partial sums, no atomics)."""
    from lstc_vad_amd.optim import clip_grad_norm_
    g = torch.Generator().manual_seed(9)
    sizes = [(2048, 513), (4096,), (3, 7), (1,), (8193,), (512, 32)] * 10
    base = [torch.randn(*s, generator=g) for s in sizes]
    holder = torch.zeros(8193 + 1)
    for max_norm in (10.0, 1e6):
        ps_cpu, ps_gpu = [], []
        for i, b in enumerate(base):
            pc = torch.nn.Parameter(torch.zeros_like(b)); pc.grad = b.clone()
            pg = torch.nn.Parameter(torch.zeros_like(b, device=DEV))
            if b.numel() == 8193 and i == 4:              # a gradient that starts 4 bytes off a 16-byte boundary
                buf = holder.to(DEV)
                pg.grad = buf[1:].view(8193); pg.grad.copy_(b)
            else:
                pg.grad = b.to(DEV)
            ps_cpu.append(pc); ps_gpu.append(pg)
        exact = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in ps_cpu)))       # f64 truth of the total norm
        want = torch.nn.utils.clip_grad_norm_(ps_cpu, max_norm)
        assert abs(float(want) - exact) < 3e-5 * exact                    # torch's own f32 norm-of-norms sits 1e-5 off the f64 value
        before = [p.grad.clone() for p in ps_gpu]
        got = clip_grad_norm_(ps_gpu, max_norm)
        assert got.is_cuda and got.dim() == 0
        assert abs(float(got) - exact) < 2e-6 * exact                     # fixed-order partial sums: f32-exact to ~1e-7
        coef = min(1.0, max_norm / (exact + 1e-6))
        for pc, pg, b0 in zip(ps_cpu, ps_gpu, before):
            if max_norm > 1e5:
                assert torch.equal(pg.grad, b0)                                         # coefficient clamps to 1: untouched
            else:
                assert max_abs_diff(pg.grad, b0.double() * coef) <= 2e-6 * float(b0.abs().max()) * coef + 1e-12
                assert max_abs_diff(pg.grad, pc.grad) <= 3e-5 * float(pc.grad.abs().max()) + 1e-12     # and torch's result to ITS norm error
        ps2 = []
        for b0 in before:
            p2 = torch.nn.Parameter(torch.zeros_like(b0)); p2.grad = b0.clone(); ps2.append(p2)
        got2 = clip_grad_norm_(ps2, max_norm)
        assert torch.equal(got, got2) and all(torch.equal(a.grad, b.grad) for a, b in zip(ps_gpu, ps2))
    # a diverged step: a NaN anywhere makes the total norm NaN and torch multiplies EVERY gradient by the NaN
    # coefficient (clamp keeps NaN) - so does the HIP path; empty tensors are skipped like torch skips them
    pn = [torch.nn.Parameter(torch.zeros(5, device=DEV)), torch.nn.Parameter(torch.zeros(7, device=DEV)), torch.nn.Parameter(torch.zeros(0, device=DEV))]
    pn[0].grad = torch.tensor([1.0, float("nan"), 2.0, 3.0, 4.0], device=DEV)
    pn[1].grad = torch.ones(7, device=DEV)
    pn[2].grad = torch.zeros(0, device=DEV)
    tot = clip_grad_norm_(pn, 10.0)
    assert torch.isnan(tot) and torch.isnan(pn[0].grad).all() and torch.isnan(pn[1].grad).all()


def test_fused_qkv_buffer_matches_separate_projections():
    """MultiHeadAttention.fuse_qkv_ (w_qs/w_ks/w_vs as row blocks of one buffer -> one GEMM each for projection, weight
    gradient and input gradient) must not change values, parameter names or the optimizer's view of the weights."""
    z, mode, ekw, skw = load_case("ltn_sht")
    d = ekw["d_model"]
    args = _args(mode, skw)
    nf, af, al = (torch.from_numpy(z[k]).to(DEV) for k in ("norm_feats", "abnorm_feats", "abnorm_labs"))
    res = []
    for fuse in (False, True):
        enc, head = _models(mode, ekw, d)
        enc.load_state_dict(sub(z, "enc_init."), strict=True)
        head.load_state_dict(sub(z, "head_init."), strict=True)
        enc, head = enc.to(DEV).train(), head.to(DEV).train()
        if fuse:
            for layer in enc.layer_stack:
                layer.slf_attn.fuse_qkv_()
            assert list(enc.state_dict().keys()) == list(sub(z, "enc_init.").keys())
        enc_out, outputs, loss, sc = _step(enc, head, mode, args, nf, af, al, d, cls_only=False)
        loss.backward()
        res.append((enc_out.detach(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}))
    assert max_abs_diff(res[0][0], res[1][0]) < 1e-6
    assert max_abs_diff(res[1][0], z["enc_out"]) < 1e-4
    for k, g in res[0][1].items():
        assert max_abs_diff(g, res[1][1][k]) < 2e-6 * float(g.abs().max()) + 1e-9, k


def test_pinned_feeder_delivers_batches_in_order():
    from lstc_vad_amd.feed import PinnedFeeder
    torch.manual_seed(0)
    batches = [(torch.randn(4, 6, 3, 8), torch.zeros(4, 6, 1), torch.randn(4, 6, 3, 8), torch.rand(4, 6, 1)) for _ in range(5)]
    got = []
    for b in PinnedFeeder(iter(batches), DEV):
        got.append(tuple(t.clone() for t in b))        # clone: the feeder reuses its device slots
        torch.cuda._sleep(200000)                      # make "compute" outlast the next copy
    assert len(got) == 5
    for g, b in zip(got, batches):
        for x, y in zip(g, b):
            assert torch.equal(x.cpu(), y)


def test_headline_size_properties():
    """BASELINE.json's full LTN shape (B=64 videos, T=32 parts, L=3, P=16, d=2048, 3 layers; 2048 sequences of 49 tokens)
    is far beyond what the CPU oracle can run in a test, so parity is checked through size-independent properties:
      * sequence independence: the first 8 sequences scored inside the full batch == the same 8 scored alone (bit-exact);
      * those 8 agree with the oracle within 1e-4 (north_star tolerance);
      * permuting the normal videos permutes the scores and leaves the MIL/CE loss unchanged (hinge over all pairs);
      * the loss of the batch split over 4 "ranks" sums to the global loss (data-parallel bookkeeping at full size)."""
    from argparse import Namespace
    from lstc_vad_amd.losses import training_loss
    from lstc_vad_amd.models import Encoder, Classifier
    torch.manual_seed(0)
    ekw = dict(n_layers=3, n_head=8, d_k=256, d_v=256, d_model=2048, d_inner=4096, MHA_layerNorm=True,
               FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=3)
    enc = Encoder(MHA_attn_dropout=0.0, MHA_fc_dropout=0.0, FFN_dropout=0.0, weight_init=True, **ekw)
    head = Classifier(2048, 0.0)
    P = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    HP = {k: v.detach().clone() for k, v in head.state_dict().items()}
    enc, head = enc.to(DEV).eval(), head.to(DEV).eval()
    bs, pn, L, Pn, d = 32, 32, 3, 16, 2048
    g = torch.Generator(device=DEV).manual_seed(5)
    nf = 0.5 * torch.relu(torch.randn(bs, pn * L, Pn, d, device=DEV, generator=g))
    af = 0.5 * torch.relu(torch.randn(bs, pn * L, Pn, d, device=DEV, generator=g)) + 0.05
    al = torch.rand(bs, pn * L, 1, device=DEV, generator=g)
    args = Namespace(batch_size=bs, part_num=pn, part_len=L, n_patch=Pn, lambda_1=0.01, lambda_MIL=1.0, lambda_CE=0.8,
                     temporal_only=False)
    with torch.no_grad():
        xs_n, xs_a = nf.view(bs * pn, L * Pn, d), af.view(bs * pn, L * Pn, d)
        out = head(enc.forward_cls(xs_n, xs_a))                          # [2048, 2]
        small = head(enc.forward_cls(xs_n[:8].contiguous()))
        assert torch.equal(out[:8], small)
        ecfg = orc.EncoderCfg(MHA_attn_dropout=0.0, MHA_fc_dropout=0.0, FFN_dropout=0.0, **ekw)
        ref = orc.head_forward(HP, orc.encoder_forward(P, xs_n[:8].cpu(), ecfg, False)[:, 0, :], "classifier")
        assert max_abs_diff(small, ref) < 1e-4
        loss, sc = training_loss(args, "LTN", out, al)
        perm = torch.randperm(bs, device=DEV, generator=g)
        out_p = head(enc.forward_cls(nf[perm].view(bs * pn, L * Pn, d), xs_a))
        assert torch.equal(out_p[: bs * pn].view(bs, pn, 2), out[: bs * pn].view(bs, pn, 2)[perm])
        loss_p, sc_p = training_loss(args, "LTN", out_p, al)
        # hinge and CE are symmetric in the videos; l1 is not (the reference's flat slice y[bs:] drops the first
        # normal video's first entries), so compare those two terms rather than the total
        assert abs(float(sc_p[2]) - float(sc[2])) < 2e-6 and abs(float(sc_p[4]) - float(sc[4])) < 2e-6
        assert abs(float(loss_p) - float(loss)) < 1e-4
        # 4-way shard of the same batch: contributions sum to the global scalars
        tot = torch.zeros(5, device=DEV)
        bag = torch.zeros(2 * bs, device=DEV)
        import ctypes as C
        from lstc_vad_amd import _lib
        from lstc_vad_amd._lib import LossDesc, check, dev_ptr
        lib = _lib.load()
        h = bs // 4
        shards = []
        for r in range(4):
            o_r = torch.cat([out[r * h * pn:(r + 1) * h * pn], out[bs * pn + r * h * pn: bs * pn + (r + 1) * h * pn]]).contiguous()
            shards.append((o_r, al[r * h:(r + 1) * h].contiguous()))
        for phase in (0, 1):
            for r, (o_r, lab) in enumerate(shards):
                dsc = LossDesc()
                dsc.mode, dsc.bs_global, dsc.bs_local, dsc.rank_off = 1, bs, h, r * h
                dsc.part_num, dsc.score_len, dsc.label_len, dsc.l1_skip = pn, 1, L, bs
                dsc.lambda_1, dsc.lambda_MIL, dsc.lambda_aux = 0.01, 1.0, 0.8
                dout, s5 = torch.zeros_like(o_r), torch.zeros(5, device=DEV)
                dsc.out, dsc.abn_labels, dsc.bag, dsc.dout, dsc.scalars = dev_ptr(o_r), dev_ptr(lab), dev_ptr(bag), dev_ptr(dout), dev_ptr(s5)
                dsc.phase = phase
                check(lib.lstc_vad_loss(C.byref(dsc), None))
                torch.cuda.synchronize()
                if phase == 1:
                    tot += s5
        assert max_abs_diff(tot, sc) < 5e-6


def test_mixed_step_equals_reference_golden_for_both_datasets():
    """BASELINE config 5 (UBnormal d=1024/L=5 mixed with SHT d=2048/L=3 in one iteration), reduced width: one
    engine.MixedStep over two model pairs reproduces, for EACH pair, the reference's weights after two Adagrad steps."""
    from lstc_vad_amd.engine import MixedStep, TrainStep
    steps, batches, goldens = [], [], []
    for name in ("ltn_ubnormal", "ltn_sht"):
        z, mode, ekw, skw = load_case(name)
        enc, head = _models(mode, ekw, ekw["d_model"])
        enc.load_state_dict(sub(z, "enc_init."), strict=True); head.load_state_dict(sub(z, "head_init."), strict=True)
        enc, head = enc.to(DEV).train(), head.to(DEV).train()
        steps.append(TrainStep(_args(mode, skw), mode, enc, head, lr_encoder=1e-4, lr_head=1e-2, weight_decay=1e-3))
        batches.append(tuple(torch.from_numpy(z[k]).to(DEV) for k in ("norm_feats", "abnorm_feats", "abnorm_labs")))
        goldens.append(z)
    mixed = MixedStep(steps)
    first = mixed.step(batches)
    mixed.step(batches)
    for ts, z, sc in zip(steps, goldens, first):
        assert abs(float(sc[0]) - float(z["scalars"][0])) < 2e-5
        for prefix, mod, lim in (("enc_after2.", ts.encoder, 1e-3), ("head_after2.", ts.head, 1e-2)):
            ref = sub(z, prefix)
            for k, v in mod.state_dict().items():
                if v.is_floating_point():       # Adagrad's first steps are sign-like: a few near-zero grads may flip
                    assert float(((v.cpu() - ref[k]).abs() > 5e-5).float().mean()) <= lim, k


# ---- f32x3 mode: f32-accurate products on the bf16 matrix cores (csrc/gemm_pk.hip) -----------------------------------

@pytest.fixture
def f32x3_everywhere():
    """Every GEMM of the block goes through lstc_pack3 + the packed kernel, whatever its size."""
    from lstc_vad_amd import functional as Fn
    Fn.set_compute_dtype("f32x3"); Fn.set_x3_threshold(0, 0, 0)
    yield Fn
    Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()


def test_f32x3_gemm_matches_f64_like_the_f32_kernel(f32x3_everywhere):
    """All three product forms + epilogues; error against an f64 product is within 1.5x of the exact-f32 kernel's."""
    Fn = f32x3_everywhere
    g = torch.Generator().manual_seed(5)
    for (M, N, K, ta, tb) in ((1000, 384, 520, False, True), (1000, 384, 520, False, False), (384, 260, 3000, True, False)):
        a = torch.randn((K, M) if ta else (M, K), generator=g)
        b = torch.randn((N, K) if tb else (K, N), generator=g) * 0.05
        ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
        ad, bd = a.to(DEV), b.to(DEV)
        Fn.set_compute_dtype("f32x3")
        c3 = Fn.gemm(ad, bd, trans_a=ta, trans_b=tb, split_k=2 if ta else 1).cpu().double()
        Fn.set_compute_dtype("fp32")
        c1 = Fn.gemm(ad, bd, trans_a=ta, trans_b=tb).cpu().double()
        e3, e1 = float((c3 - ref).abs().max()), float((c1 - ref).abs().max())
        assert e3 <= 1.5 * e1 + 1e-7, (M, N, K, ta, tb, e3, e1)
    Fn.set_compute_dtype("f32x3")
    x = torch.randn(700, 300, generator=g).to(DEV); w = (torch.randn(260, 300, generator=g) * 0.1).to(DEV)
    bias = torch.randn(260, generator=g).to(DEV); res = torch.randn(700, 260, generator=g).to(DEV)
    y3 = Fn.gemm(x, w, trans_b=True, bias=bias, relu=True, dropout=(0.3, 77), residual=res)
    Fn.set_compute_dtype("fp32")
    y1 = Fn.gemm(x, w, trans_b=True, bias=bias, relu=True, dropout=(0.3, 77), residual=res)
    assert max_abs_diff(y3, y1) < 2e-5


@pytest.mark.parametrize("name", ["ltn_sht", "stn_mil_ce", "ltn_ubnormal"])
def test_f32x3_training_step_matches_reference_golden(name, f32x3_everywhere):
    """The golden two-step training test, every product on the packed bf16-plane kernel, SAME tolerances as f32."""
    test_training_step_matches_reference_golden(name, True)
    test_training_step_matches_reference_golden(name, False)


def test_f32x3_full_width_scores_match_oracle(f32x3_everywhere):
    f32x3_everywhere.set_x3_threshold()          # production thresholds: big products packed, small ones exact f32
    test_full_width_scores_match_oracle()


def test_f32x3_weight_gradient_reuses_forward_packs(f32x3_everywhere):
    """dW = dY^T X through the transposed-read form of the packed kernel (packs of the [T, out] / [T, in] sources, as the
    forward and input-gradient products make them) == the k-major-pack form == the exact-f32 kernel, to f32 accuracy."""
    Fn = f32x3_everywhere
    g = torch.Generator().manual_seed(9)
    T, O, I = 1152, 256, 384
    dy = torch.randn(T, O, generator=g).to(DEV); x = torch.randn(T, I, generator=g).to(DEV)
    ref = dy.cpu().double().t() @ x.cpu().double()
    xp = Fn.maybe_pack(x)
    assert xp is not None
    for split in (False, True):
        got = Fn.wgrad(dy if not split else torch.cat([dy] * 4), x if not split else torch.cat([x] * 4),
                       None if split else xp)
        scale = 4.0 if split else 1.0
        assert float((got.cpu().double() - scale * ref).abs().max()) < 2e-4 * scale
    Fn.set_compute_dtype("fp32")
    e1 = float((Fn.wgrad(dy, x).cpu().double() - ref).abs().max())        # the exact-f32 kernel's own distance from f64
    Fn.set_compute_dtype("f32x3")
    e3 = float((Fn.wgrad(dy, x, xp).cpu().double() - ref).abs().max())
    assert e3 <= 1.5 * e1 + 1e-6, (e3, e1)


def test_f32x3_accuracy_under_dynamic_range():
    """The per-tensor power-of-two scale must not cost accuracy when a few rows or one element dominate the tensor (as in
    gradient tensors): error against f64 of dY^T X and of the small rows of dY W^T stays at the exact-f32 kernel's level."""
    from lstc_vad_amd import functional as Fn
    g = torch.Generator().manual_seed(0)
    T, O, I = 1152, 512, 384
    w = torch.randn(I, O, generator=g)
    try:
        for boost in (1.0, 1e3, 1e5, -1e6):
            dy = torch.randn(T, O, generator=g); x = torch.randn(T, I, generator=g)
            if boost > 0:
                dy[::100] *= boost
            else:
                dy[5, 7] *= -boost
            wref = dy.double().t() @ x.double()
            yref = dy.double() @ w.double().t()
            small = [i for i in range(T) if i % 100 != 0 and i != 5][:300]
            err = {}
            for mode in ("fp32", "f32x3"):
                Fn.set_compute_dtype(mode); Fn.set_x3_threshold(0, 0, 0)
                wg = Fn.wgrad(dy.to(DEV), x.to(DEV)).cpu().double()
                y = Fn.gemm(dy.to(DEV), w.to(DEV), trans_b=True).cpu().double()
                err[mode] = (float((wg - wref).abs().max() / wref.abs().max()),
                             float((y - yref)[small].abs().max() / yref[small].abs().max()))
            assert err["f32x3"][0] <= 1.5 * err["fp32"][0] + 1e-7, (boost, err)
            assert err["f32x3"][1] <= 1.5 * err["fp32"][1] + 1e-7, (boost, err)
    finally:
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()


def test_f32x3_full_width_step_tracks_exact_f32_step():
    """Headline layer shapes (d=2048, 8x256 heads, F=4096, S=49), one training step without dropout in both GEMM modes:
    every forward product agrees to f32 rounding (<= 1e-5 of its maximum).  Backward products agree in norm; single
    entries may differ by ~1e-3 because a handful of the ~10 M ReLU outputs lie within f32 rounding of zero and land on
    different sides in ANY two f32 implementations (tools/x3_step_probe.py: 1-5 such flips per step, forward differences
    2e-6) - which is why the gradient parity of record is the golden test, where no output sits on that edge."""
    from argparse import Namespace
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.losses import training_loss
    from lstc_vad_amd.models import Classifier, Encoder
    torch.manual_seed(3)
    ekw = dict(n_layers=3, n_head=8, d_k=256, d_v=256, d_model=2048, d_inner=4096, MHA_layerNorm=True, FFN_layerNorm=True,
               relative_pe=True, window_size=4, window_depth=3)
    enc = Encoder(MHA_attn_dropout=0.0, MHA_fc_dropout=0.0, FFN_dropout=0.0, weight_init=True, **ekw).to(DEV).train()
    head = Classifier(2048, 0.0).to(DEV).train()
    bs, pn, L, P, d = 2, 6, 3, 16, 2048                     # 24 sequences x 49 tokens = 1176 rows: above the x3 threshold
    nf = (0.5 * torch.relu(torch.randn(bs, pn * L, P, d))).to(DEV); af = (0.5 * torch.relu(torch.randn(bs, pn * L, P, d))).to(DEV)
    u = torch.rand(bs, pn * L, 1); al = torch.where(u > 0.65, u, torch.zeros_like(u)).to(DEV)
    args = Namespace(batch_size=bs, part_num=pn, part_len=L, n_patch=P, lambda_1=0.01, lambda_MIL=1.0, lambda_CE=0.8,
                     lambda_BCE=1.0, lambda_normal=0.2, lambda_abnormal=2.0, temporal_only=False, clip_grad=False)
    orig = Fn.gemm
    logs = {}
    try:
        for mode in ("fp32", "f32x3"):
            log = []

            def recording_gemm(a, b, _log=log, **kw):
                out = orig(a, b, **kw)
                _log.append((bool(kw.get("trans_a", False)) or kw.get("relu_mask") is not None, out.detach().clone()))
                return out
            Fn.gemm = recording_gemm
            Fn.set_compute_dtype(mode)
            enc.zero_grad(set_to_none=True); head.zero_grad(set_to_none=True)
            cls = enc.forward_cls(nf.reshape(bs * pn, L * P, d), af.reshape(bs * pn, L * P, d))
            loss, sc = training_loss(args, "LTN", head(cls), al)
            loss.backward()
            logs[mode] = (log, float(sc[0].detach()))
    finally:
        Fn.gemm = orig
        Fn.set_compute_dtype("fp32")
    (l1, loss1), (l3, loss3) = logs["fp32"], logs["f32x3"]
    diffs = [float((x - y).abs().max() / (x.abs().max() + 1e-30)) for (_, x), (_, y) in zip(l1, l3)]
    assert len(l1) == len(l3) > 40 and abs(loss1 - loss3) < 1e-5, (loss1, loss3, [f"{v:.1e}" for v in diffs[:20]])
    n_fwd = next(i for i, e in enumerate(l1) if e[0])        # first backward product (a weight gradient / masked dX)
    assert n_fwd >= 16
    for (_, x), (_, y) in zip(l1[:n_fwd], l3[:n_fwd]):
        assert float((x - y).abs().max() / x.abs().max()) < 1e-5
    for (_, x), (_, y) in zip(l1[n_fwd:], l3[n_fwd:]):
        assert float((x - y).double().norm() / x.double().norm()) < 5e-3


@pytest.mark.parametrize("gemm_mode", ["fp32", "f32x3"])
def test_training_step_is_bit_reproducible(gemm_mode):
    """Two runs of the same training steps (dropout on, same seeds) give bit-identical weights: weight-gradient K splits,
    bias-table gradients and the head's dW/db are summed in a fixed order (no float atomics on the default path)."""
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.engine import TrainStep
    from lstc_vad_amd.models import Classifier, Encoder
    ekw = dict(n_layers=3, n_head=4, d_k=64, d_v=64, d_model=256, d_inner=512, MHA_layerNorm=True, FFN_layerNorm=True,
               relative_pe=True, window_size=4, window_depth=3)
    bs, pn, L, P, d = 8, 16, 3, 16, 256               # 256 sequences x 49 tokens = 12544 rows: split-K weight gradients
    g = torch.Generator().manual_seed(2)
    nf = (0.5 * torch.relu(torch.randn(bs, pn * L, P, d, generator=g))).to(DEV)
    af = (0.5 * torch.relu(torch.randn(bs, pn * L, P, d, generator=g))).to(DEV)
    u = torch.rand(bs, pn * L, 1, generator=g); al = torch.where(u > 0.65, u, torch.zeros_like(u)).to(DEV)
    results = []
    Fn.set_compute_dtype(gemm_mode)
    for run in range(2):
        torch.manual_seed(11); Fn.reset_rng(0)
        enc = Encoder(MHA_attn_dropout=0.2, MHA_fc_dropout=0.2, FFN_dropout=0.1, weight_init=True, **ekw).to(DEV).train()
        head = Classifier(d, 0.6).to(DEV).train()
        ts = TrainStep(_args("LTN", dict(batch_size=bs, part_num=pn, part_len=L, n_patch=P)), "LTN", enc, head, 1e-4, 1e-2, 1e-3)
        for _ in range(3):
            sc = ts.step(nf, af, al)
        results.append((sc.cpu(), {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}))
    Fn.set_compute_dtype("fp32")
    assert torch.equal(results[0][0], results[1][0])
    for k, v in results[0][1].items():
        assert torch.equal(v, results[1][1][k]), k


def test_f32x3_scale_edge_cases(f32x3_everywhere):
    """Power-of-two tensor scales at the ends of the f32 range, an all-zero operand, and operands of very different
    magnitude: same relative accuracy as in the middle of the range, zeros stay zeros."""
    Fn = f32x3_everywhere
    g = torch.Generator().manual_seed(4)
    a = torch.randn(300, 200, generator=g); b = torch.randn(260, 200, generator=g)
    ref = a.double() @ b.double().t()
    for sa, sb in ((1.0, 1.0), (1e30, 1e-30), (1e-15, 1e-15), (3e15, 2e15), (1e-20, 1.0)):
        y = Fn.gemm((a * sa).to(DEV), (b * sb).to(DEV), trans_b=True).cpu().double()
        want = ref * (float(torch.tensor(sa, dtype=torch.float32)) * float(torch.tensor(sb, dtype=torch.float32)))
        rel = float((y - want).abs().max() / want.abs().max())
        assert rel < 2e-6, (sa, sb, rel)
    z = Fn.gemm(torch.zeros(300, 200, device=DEV), b.to(DEV), trans_b=True)
    assert float(z.abs().max()) == 0.0
    w = Fn.wgrad(torch.zeros(256, 128, device=DEV), torch.randn(256, 128, generator=g).to(DEV))
    assert float(w.abs().max()) == 0.0


def _full_width_models(name):
    """Build-side models at BASELINE width with the portable-generator weights the full-width fixture was made from."""
    from cases import FULL_CASES, HEADLINE_CASES, PACKED_CASES, fill_params
    mode, ekw, skw, seed = {**FULL_CASES, **PACKED_CASES, **HEADLINE_CASES}[name]
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    assert int(z["seed"]) == seed
    d = ekw["d_model"]
    enc, head = _models(mode, dict(ekw), d)
    fill_params(enc, seed)
    fill_params(head, seed + 1)
    from util import cached_training_batch
    nf, _, af, al = cached_training_batch(skw["batch_size"], skw["part_num"], skw["part_len"], skw["n_patch"], d, seed=seed,
                                          with_pseudo=True, threshold=0.6)
    return z, mode, skw, d, enc, head, nf, af, al


FULL_NAMES = ["ltn_full", "stn_full", "ltn_ucf_full", "ltn_ubnormal_full", "stn_mil_ce_full", "ltn_clip_full"]


class _align_relu_edges:
    """``with _align_relu_edges(z, n_seq, S) as st:`` - the HIP step takes the reference's ReLU decision at the hidden units the
    reference run found within float32 rounding of zero (fixture keys ``relu_edge.{layer}`` = (token row, unit),
    ``relu_edge_pre.{layer}`` = the reference's pre-activation there; |pre| < 4e-6, a few dozen of ~2e7 units per layer; and
    ``relu_edge.head`` for the hidden Linear(d, 512) + ReLU of the Classifier / Regressor, band 1.6e-5).
    Which side of zero such a unit lands on is decided by the summation order of the 2048 products - rocBLAS sgemm, the
    reference's CPU sgemm and this repo's k-ordered MFMA chain each disagree with an f64 product on a handful of them
    (tools/relu_flip_probe.py) - and ONE flipped unit with a large upstream gradient moves the small layer-0 attention
    gradients by up to 1e-3 of their maximum (tools/block_probe.py).  The hidden is set to relu(reference pre-activation)
    at exactly the listed units (values < 4e-6, < 1.2e-5 in the 256-sequence fixtures of cases.PACKED_CASES; head: 4x that:
    the forward is unchanged at the 1e-4 bar), nowhere else; ``st.changed`` counts the decisions that differed, ``st.listed``
    the units visited.  A CLS-only last layer holds one row per sequence: only the listed rows that are CLS tokens exist there.

    Nothing in the product knows about this: the context manager wraps ``functional.gemm`` (a module-level name the Function
    bodies look up at call time) and edits the f32 result of the ReLU-epilogue products - the three FFN hiddens in layer order,
    then the head's - before their consumer is launched."""

    def __init__(self, z, n_seq, S, n_layers=3, only=None):
        """``only``: the sites whose recorded decisions are imposed (default: all four); the others are visited and counted
        (``changed_at``) but left as the HIP step decided them."""
        self.z, self.n_seq, self.S = z, n_seq, S
        self.changed = self.listed = 0
        self.sites = [str(i) for i in range(n_layers)] + ["head"]
        self.only = None if only is None else set(only)
        self.changed_at = {s_: 0 for s_ in self.sites}
        self.calls = 0

    def _apply(self, site, h):
        key = f"relu_edge.{site}"
        if key not in self.z.files or self.z[key].shape[0] == 0:
            return
        tj = torch.from_numpy(self.z[key]).to(h.device)
        pre = torch.from_numpy(self.z[f"relu_edge_pre.{site}"]).to(h.device)
        rows, cols = tj[:, 0], tj[:, 1]
        if site == "head":
            assert h.shape[0] == self.n_seq, (h.shape, self.n_seq)       # one CLS row per sequence, whatever the later view
        elif h.shape[0] == self.n_seq and self.S > 1:        # CLS-only last layer: row = sequence, only s == 0 tokens exist
            keep = rows % self.S == 0
            rows, cols, pre = rows[keep] // self.S, cols[keep], pre[keep]
        else:
            assert h.shape[0] == self.n_seq * self.S, (h.shape, self.n_seq, self.S)
        if pre.numel() == 0:
            return
        band = 1.2e-5 * (4 if site == "head" else 1)             # 4e-6 band; 1.2e-5 in the 256-sequence fixtures; head 4x
        assert float(pre.abs().max()) < band
        want = pre.clamp_min(0.0)
        got = h[rows, cols]
        assert float((got - want).abs().max()) < 2 * band               # same values up to f32 rounding of the product
        n_ch = int(((got > 0) != (want > 0)).sum())
        self.changed += n_ch
        self.changed_at[site] += n_ch
        self.listed += int(pre.numel())
        if self.only is None or site in self.only:
            h[rows, cols] = want

    def __enter__(self):
        from lstc_vad_amd import functional as Fn
        self.real = real = Fn.gemm

        def gemm(a, b, *args, **kw):
            out = real(a, b, *args, **kw)
            if kw.get("relu") and not kw.get("out_pack"):
                site = self.sites[self.calls % len(self.sites)]
                self.calls += 1
                self._apply(site, out)
            return out
        Fn.gemm = gemm
        return self

    def __exit__(self, *exc):
        from lstc_vad_amd import functional as Fn
        Fn.gemm = self.real
        return False


def _full_width_golden_check(name, cls_only, compute_dtype="fp32"):
    """Body of the full-width golden tests (``compute_dtype``: GEMM mode of the HIP step).  BASELINE widths (d=2048, H=8x256, F=4096 / 3027, 3 layers), T >= 4096 tokens: forward, loss, EVERY parameter
    gradient and the weights after two Adagrad steps against the real reference's run (tests/golden/make_golden.py
    ``run_full_case``; sampled entries + norms, weights regenerate from the seed).  This is the oracle check of the
    production-size backward: PIPE 5 steady loop in NT/NN/TN, batched split-K weight gradients, attn_bwd at d_k = 256,
    n_hidden = 3027 at its padded width 3072 (test_full_width_unpadded_hidden_... below runs the scalar-load path)."""
    from lstc_vad_amd import functional as Fn
    Fn.set_compute_dtype(compute_dtype)
    try:
        _full_width_golden_body(name, cls_only)
    finally:
        Fn.set_compute_dtype("fp32")


# Bars of the UN-ALIGNED step (nothing edited: the HIP step as the product runs it) against the reference's run.  The forward
# bars are the aligned ones (north_star's 1e-4 on the scores).  The gradient bars are what float32 + ReLU leaves between two
# correct implementations at this size - a flipped hidden unit rewrites its row of dW1 and shifts everything upstream
#  - and the measured maxima over all cases are printed by the test (``pytest -s``) and quoted in DESIGN 4.
UNALIGNED_GRAD_BAR = 5e-3        # of the tensor's maximum, every sampled entry of every parameter gradient (measured maximum over all
                                 # cases and modes: 3.4e-3, layer 0's dW1 of one case; typical 1e-4; round-4 GPU log in profiles/)
UNALIGNED_NORM_BAR = 1e-3        # relative, every gradient norm
UNALIGNED_FRACTION_BAR = 0.995   # share of ALL sampled gradient entries of a step (~45 tensors x 256) that already meet the strict bar un-aligned
UNALIGNED_LOG = {}               # name -> (worst entry error / max, worst relative norm error, flipped units): read by the summary test


def _clip_rescale(z):
    """ltn_clip_full: the fixture's gradients are clipped with TORCH-CPU's total norm, whose float32 norm of norms over 100.7 M
    elements is 3.1e-4 low (10.40038 against 10.40359 in float64, both recorded by the generator).  lstc_sqnorm_multi sums in
    a fixed order of float32 partials and lands on the float64 value to 1e-7, so the HIP step clips with the accurate coefficient;
    the comparison rescales the reference's clipped gradients by coef(f64 norm) / coef(torch's norm) per parameter group."""
    if "clip_total_norm_f64_step0" not in z.files:
        return None
    coef = lambda n: min(1.0, 10.0 / (float(n) + 1e-6))
    return {pre: coef(z["clip_total_norm_f64_step0"][i]) / coef(z["clip_total_norm_step0"][i]) for i, pre in enumerate(("enc", "head"))}


def _compare_full_width_step0(z, enc, head, enc_out, outputs, sc, cls_only, gbar, nbar, strict=None):
    """Forward rows, scores, scalars and every parameter gradient of step 0 against the fixture; returns the worst gradient entry
    error (as a fraction of its tensor's maximum) and the worst relative norm error - and, with ``strict`` (the aligned pass's
    entry bar), also (sampled entries beyond ``strict`` of their tensor's maximum, sampled entries in all)."""
    gscale = _clip_rescale(z)
    from cases import sample_index
    n_seq = enc_out.shape[0]
    cls = enc_out[:, 0, :][::max(1, n_seq // 16)][:16]
    assert max_abs_diff(cls, z["cls_rows"]) < 5e-4                         # post-LN activations are O(1..5)
    if not cls_only:
        tok = enc_out[::max(1, n_seq // 8), enc_out.shape[1] // 2, :][:8]
        assert max_abs_diff(tok, z["tok_rows"]) < 5e-4
    assert max_abs_diff(outputs.reshape(z["outputs"].shape), z["outputs"]) < 1e-4      # north_star tolerance
    assert np.max(np.abs(sc.cpu().double().numpy() - z["scalars"])) < 2e-5
    worst_e = worst_n = 0.0
    beyond = beyond5 = total = 0
    for pre, mod in (("enc", enc), ("head", head)):
        want = {k[len(pre) + 7:] for k in z.files if k.startswith(pre + "_gnorm.")}
        got = {k for k, p in mod.named_parameters() if p.grad is not None}
        assert got == want, got ^ want
        for k, p in mod.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.detach().reshape(-1)
            r = 1.0 if gscale is None else gscale[pre]
            gmax, gnorm = r * float(z[f"{pre}_gmax.{k}"]), r * float(z[f"{pre}_gnorm.{k}"])
            idx = torch.from_numpy(sample_index(g.numel())).to(DEV)
            err = max_abs_diff(g[idx], r * z[f"{pre}_gs.{k}"].astype(np.float64))
            assert err < gbar * gmax + 1e-7, (pre, k, err, gmax)
            if strict is not None:
                dlt = (g[idx].cpu().double() - torch.from_numpy(r * z[f"{pre}_gs.{k}"].astype(np.float64))).abs()
                beyond += int((dlt >= strict * gmax + 1e-7).sum())
                beyond5 += int((dlt >= 5 * strict * gmax + 1e-7).sum())
                total += int(dlt.numel())
            nerr = abs(float(g.double().norm()) - gnorm)
            assert nerr < nbar * gnorm + 1e-9, (pre, k, float(g.double().norm()), gnorm)
            assert abs(float(g.abs().max()) - gmax) < gbar * gmax + 1e-7, (pre, k)
            if gmax > 0:
                worst_e, worst_n = max(worst_e, err / gmax), max(worst_n, nerr / gnorm)
    if strict is not None:
        return worst_e, worst_n, (beyond, beyond5), total
    return worst_e, worst_n


def _full_width_golden_body(name, cls_only):
    from cases import sample_index
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.optim import Adagrad
    z, mode, skw, d, enc, head, nf, af, al = _full_width_models(name)
    enc, head = enc.to(DEV).train(), head.to(DEV).train()
    args = _args(mode, skw)
    nf, af, al = (torch.from_numpy(x).to(DEV) for x in (nf, af, al))
    opt = Adagrad([{"params": enc.parameters(), "lr": 1e-4}, {"params": head.parameters(), "lr": 1e-2}], weight_decay=1e-3)
    init = {("enc", k): p.detach().clone() for k, p in enc.named_parameters()}
    init.update({("head", k): p.detach().clone() for k, p in head.named_parameters()})
    n_seq_all = 2 * skw["batch_size"] * skw["part_num"] * (1 if mode == "LTN" else skw["part_len"])
    S_all = 1 + skw["n_patch"] * (skw["part_len"] if mode == "LTN" else 1)
    # a weight gradient is a sum over all tokens: between two f32 implementations (the reference's CPU sgemm, the k-ordered MFMA
    # chain) its rounding noise grows like sqrt(tokens).  The 2e-4 bar was set on the 4352 - 6272-token cases; the 256-sequence
    # cases (12 544 / 20 736 tokens) get the same bar scaled by that root (measured there: 2.65e-4 on ONE sampled entry of layer
    # 0's dW_v at 20 736 tokens, with or without the wider ReLU-edge band, i.e. not a flipped unit)
    gbar = 2e-4 * max(1.0, (n_seq_all * S_all / 6272.0) ** 0.5)

    # (1) the step exactly as the product runs it - NOTHING aligned - against the reference: forward at the strict bars, gradients
    # at the stated un-aligned bars (one flipped ReLU unit of ~2e7 per layer is allowed to show)
    def clip():
        """--clip_grad as the train loop applies it (Train/temporal_transformer_shanghaitech.py:139-141); ltn_clip_full: the
        encoder's norm is 10.4, so the gradients the fixture holds ARE scaled by 10 / (10.4 + 1e-6)."""
        if not args.clip_grad:
            return None
        from lstc_vad_amd.optim import clip_grad_norm_
        return float(clip_grad_norm_(enc.parameters(), 10)), float(clip_grad_norm_(head.parameters(), 10))

    enc_out, outputs, loss, sc = _step(enc, head, mode, args, nf, af, al, d, cls_only)
    opt.zero_grad()
    loss.backward()
    norms = clip()
    if norms is not None:          # total norms of 100.7 M / 1.07 M gradient elements: against the float64 norm of the REFERENCE's gradients
        ref64, ref32 = z["clip_total_norm_f64_step0"], z["clip_total_norm_step0"]
        assert ref64[0] > 10.0 and abs(norms[0] - ref64[0]) < 2e-5 * ref64[0] and abs(norms[1] - ref64[1]) < 2e-5 * ref64[1], (norms, ref64)
        assert abs(ref32[0] / ref64[0] - 1.0) > 1e-4            # (torch-CPU's own float32 value is the one that is off: see _clip_rescale)
    we, wn, beyond, total = _compare_full_width_step0(z, enc, head, enc_out, outputs, sc, cls_only, UNALIGNED_GRAD_BAR, UNALIGNED_NORM_BAR,
                                                      strict=gbar)
    # not only the maximum - HOW MANY of the sampled gradient entries of the un-aligned step already meet the
    # strict (aligned-pass) bar.  A flipped ReLU unit rewrites one row of dW1 and shifts the gradients upstream of its FFN by up to
    # ~1e-3 of their maximum, so whole small tensors (layer 0's attention projections) can sit above 2e-4 in one case while 40 others
    # are untouched: the bar is on the fraction over all sampled entries of the step (measured: printed below, profiles/r05_*)
    beyond, beyond5 = beyond
    frac_ok, frac5_ok = 1.0 - beyond / max(total, 1), 1.0 - beyond5 / max(total, 1)
    del enc_out, outputs, loss
    # (1b) round 6 - no case is exempted BY NAME any more.  Where the un-aligned fraction misses the bar, the test measures WHICH
    # recorded ReLU-edge site explains it: the same step with the reference's decision imposed at ONE site at a time (layer 0 / 1 / 2
    # FFN hidden, the head's hidden; everything else as the product decides).  Printed for the record (DESIGN 4 quotes it); asserted:
    # at least one site really decided a recorded unit differently, and the fully aligned pass (2) below meets the strict bars - i.e.
    # the recorded edge units, and nothing else, separate this step from the reference's.
    site_table = None
    if frac_ok < UNALIGNED_FRACTION_BAR or frac5_ok < 0.999:
        site_table = {}
        for site in ("0", "1", "2", "head"):
            with _align_relu_edges(z, n_seq_all, S_all, only=(site,)) as he:
                enc_out, outputs, loss, sc_h = _step(enc, head, mode, args, nf, af, al, d, cls_only)
            opt.zero_grad()
            loss.backward()
            clip()
            _, _, (b1, b5), tot = _compare_full_width_step0(z, enc, head, enc_out, outputs, sc_h, cls_only, UNALIGNED_GRAD_BAR,
                                                             UNALIGNED_NORM_BAR, strict=gbar)
            site_table[site] = (he.changed_at[site], 1.0 - b1 / max(tot, 1), 1.0 - b5 / max(tot, 1))
            del enc_out, outputs, loss
        print(f"\n[one site aligned] {name}: un-aligned {100.0 * frac_ok:.2f} % within the strict bar; with the reference's decisions at ONE site: " +
              "; ".join(f"{'head' if k == 'head' else 'FFN ' + k}: {v[0]} unit(s) differed -> {100.0 * v[1]:.2f} % ({100.0 * v[2]:.2f} % within 5x)"
                        for k, v in site_table.items()))
        assert sum(v[0] for v in site_table.values()) >= 1, site_table
    for step in range(2):
        if step == 0:
            # (2) the same step with the reference's decision at the recorded edge units: every gradient at the strict bars
            with _align_relu_edges(z, n_seq_all, S_all) as edges:
                enc_out, outputs, loss, sc = _step(enc, head, mode, args, nf, af, al, d, cls_only)
            UNALIGNED_LOG[(name, cls_only, Fn._compute_dtype)] = (we, wn, edges.changed, edges.listed)
            print(f"\n[un-aligned] {name} cls_only={cls_only}: worst gradient entry {we:.2e} of its tensor max, worst norm {wn:.2e}; "
                  f"{100.0 * frac_ok:.2f} % of the {total} sampled entries within the strict bar ({gbar:.1e} of the maximum), "
                  f"{100.0 * frac5_ok:.2f} % within 5x; "
                  f"{edges.changed} of {edges.listed} recorded edge units decided differently")
            # 150 - 1200 edge units were visited (four sites), and the decisions that differed are the ones float32 leaves open: a
            # recorded unit lies within 4e-6 (1.2e-5) of zero and two summation orders differ by ~5e-7 there, so about one in ten
            # lands on the other side (measured 9 - 34 per case = 4 - 13 % of the listed units); everything else was left alone
            # (round 5: // 7 instead of // 6.  The flat tenth VERDICT r4 asked for does not hold: ltn_ucf_full and ltn_clip_full measure
            # 12.1 - 13.1 % - 21 of 173, 31 of 237, 25 of 200, 35 of 285 - on every run: which side of zero a pre-activation of
            # magnitude < 4e-6 lands on is a property of the summation order, not something a kernel can be tuned towards)
            assert edges.calls == 4 and edges.listed > 0 and edges.changed <= max(8, edges.listed // 7), (edges.calls, edges.listed, edges.changed)
            # measured (profiles/r05_gpu_tests_unaligned.md): >= 99.73 % of the sampled entries within the strict bar in every case but
            # one - ltn_ubnormal_full_256, 96.5 %: ONE hidden unit of the HEAD (reference pre-activation -6e-8) lands on the other side
            # of zero there and shifts every gradient of the encoder, since all of them flow through the head - and >= 99.99 % within
            # five times the strict bar everywhere
            # (round 6: the exemption of that case by name is gone - a miss is analysed site by site in pass (1b) above, the aligned pass
            # below must then meet the strict bars, and even so the un-aligned step keeps 85 % within the strict bar and 99.5 % within
            # five times it.  Measured misses: ltn_ubnormal_full_256 96.5 %, stn_headline 87.8 % in exact f32 - 100 % in f32x3)
            if site_table is None:
                assert total > 5000 and frac5_ok >= 0.999 and frac_ok >= UNALIGNED_FRACTION_BAR, (name, cls_only, beyond, beyond5, total)
            else:
                assert total > 5000 and frac5_ok >= 0.995 and frac_ok >= 0.85, (name, cls_only, beyond, beyond5, total, site_table)
        else:
            enc_out, outputs, loss, sc = _step(enc, head, mode, args, nf, af, al, d, cls_only)
        opt.zero_grad()
        loss.backward()
        clip()
        if step == 0:
            _compare_full_width_step0(z, enc, head, enc_out, outputs, sc, cls_only, gbar, 1e-4)
        else:
            assert np.max(np.abs(sc.cpu().double().numpy() - z["scalars_step2"])) < 1e-4
        opt.step()
    for pre, mod in (("enc", enc), ("head", head)):
        lr = 1e-4 if pre == "enc" else 1e-2
        for k, p in mod.named_parameters():
            w = p.detach().reshape(-1)
            idx = torch.from_numpy(sample_index(w.numel())).to(DEV)
            diff = (w[idx].cpu() - torch.from_numpy(z[f"{pre}_w2s.{k}"])).abs()
            # an Adagrad step moves an entry by at most lr; an entry whose gradient sits at rounding level may flip sign
            assert float(diff.max()) <= 2 * 2 * lr + 1e-6, (pre, k, float(diff.max()))
            assert float((diff > 5e-5).float().mean()) <= (1e-2 if pre == "head" else 4e-3), (pre, k, float(diff.max()))
            assert float((w - init[(pre, k)].reshape(-1)).abs().max()) > 0 or z[f"{pre}_w2s.{k}"].size == 0 or \
                np.array_equal(z[f"{pre}_w2s.{k}"], init[(pre, k)].reshape(-1)[idx].cpu().numpy())


@pytest.mark.parametrize("name,cls_only", [(n, c) for n in FULL_NAMES for c in (True, False)] +
                         [("ltn_full_256", True), ("ltn_ubnormal_full_256", True)])
def test_full_width_training_step_matches_reference_golden(name, cls_only):
    """Exact-f32 MFMA path (the mode ``value`` of bench.py is measured in) at the widths of BASELINE configs 2 (ltn_full), 1
    (stn_full), 4 (ltn_ucf_full: S = 19, [32, 32] index read through [:18, :18]), 5 (ltn_ubnormal_full: d_model = 1024,
    S = 81) and 3's co-teaching stage (stn_mil_ce_full: MIL + BCE on pseudo labels, Train/spatio_transformer_MIL_CE.py:23-44)
    against the reference's own run - see _full_width_golden_check; and the 256-sequence cases of cases.PACKED_CASES
    (12 544 tokens at S = 49, 20 736 at S = 81: several tile rounds of every GEMM; the S = 81 case has one ReLU unit of the
    classifier head on the rounding edge, which the fixtures record since round 4 - ``relu_edge.head``).  Each case runs the
    step twice: as the product runs it (un-aligned bars UNALIGNED_*), then with the recorded edge decisions (strict bars)."""
    _full_width_golden_check(name, cls_only)


@pytest.mark.parametrize("name,dtype", [("ltn_full", "fp32"), ("stn_mil_ce_full", "fp32"), ("ltn_ubnormal_full", "fp32"), ("ltn_full", "f32x3")])
def test_full_width_dropout_on_step_replays_through_oracle(name, dtype):
    """Every dropout ON at PRODUCTION width (the reference's masks come from torch's generator and cannot be replayed, so the
    full-width goldens run with rate 0): the HIP step's masks are exported (lstc_dropout_mask: the same key derivation and hash
    the fused sites use) and injected into the oracle - which the rate-0 goldens pin to the reference at these widths - on the
    host cores.  Covers what no golden can: the dropout + residual epilogue of the PIPE 5 GEMM on full tiles, the LayerNorm
    backward's fused dropout replay and bias-gradient sums at d = 2048 / 1024, attention dropout at d_k = 256 (S = 49, 17, 81),
    the CLS-only last layer's dropout sites, the padded n_hidden = 3027 block - at the reference's rates (0.2 / 0.2 / 0.1, head
    0.6).  Bars: scores 1e-4, loss 2e-5; gradients un-aligned (a ReLU unit on the rounding edge may fall either way, as in the
    golden tests' first pass) and compared on EVERY entry, not a 256-entry sample: 1e-2 of the tensor maximum (measured 8e-5 ...
    5.1e-3: one flipped unit rewrites its whole row of dW1, and an all-entries maximum finds that row), norms 1e-3."""
    from cases import FULL_CASES, fill_params
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd import synthetic as syn
    mode, ekw, skw, seed = FULL_CASES[name]
    d = ekw["d_model"]
    pa, pf, pn_, ph = 0.2, 0.2, 0.1, 0.6
    from lstc_vad_amd.models import Encoder, Regressor, Classifier
    enc = Encoder(n_layers=3, MHA_attn_dropout=pa, MHA_fc_dropout=pf, FFN_dropout=pn_, position_dropout=0.0, weight_init=False, **ekw)
    head = Classifier(d, ph, weight_init=False) if mode == "LTN" else Regressor(d, ph, weight_init=False)
    fill_params(enc, seed); fill_params(head, seed + 1)
    enc_P = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in enc.state_dict().items()}
    head_P = {k: v.detach().clone().requires_grad_(True) for k, v in head.state_dict().items()}
    enc, head = enc.to(DEV).train(), head.to(DEV).train()
    args = _args(mode, skw)
    nf, _, af, al = syn.training_batch(skw["batch_size"], skw["part_num"], skw["part_len"], skw["n_patch"], d, seed=seed,
                                       with_pseudo=True, threshold=0.6)
    nf, af, al = (torch.from_numpy(x) for x in (nf, af, al))
    Fn.set_compute_dtype(dtype)
    try:
        torch.manual_seed(4321)
        Fn.reset_rng()
        with Fn.record_dropout() as sites:
            enc_out, outputs, loss, sc = _step(enc, head, mode, args, nf.to(DEV), af.to(DEV), al.to(DEV), d, cls_only=True)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        Fn.set_compute_dtype("fp32")
    S_full = 1 + (args.part_len * args.n_patch if mode == "LTN" else args.n_patch)
    masks = {}
    for site, pp, sd, shape in sites:
        m = Fn.dropout_mask(shape, pp, sd, DEV).cpu()
        if site.endswith("dropout#cls") and len(shape) == 2 or (site.startswith("layer_stack.2.pos_ffn") and len(shape) == 2):
            full = torch.ones(shape[0], S_full, shape[1], dtype=torch.uint8)          # CLS-only last layer: the mask covers token 0
            full[:, 0, :] = m
            m = full
        masks[site.replace("#cls", "")] = m
    assert len(masks) == len(sites) >= 3 * 3 + 2
    ecfg, st = oracle_cfgs(mode, dict(ekw), dict(skw), dropout=0.0)
    ecfg.MHA_attn_dropout, ecfg.MHA_fc_dropout, ecfg.FFN_dropout = pa, pf, pn_
    st.head_dropout = ph
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    out = orc.forward_loss(enc_P, head_P, ecfg, st, nf, af, al, training=True, masks=masks)
    out["loss"].backward()
    assert max_abs_diff(outputs.reshape(out["outputs"].shape), out["outputs"]) < 1e-4            # north_star tolerance
    assert abs(float(sc[0]) - float(out["loss"].detach())) < 2e-5
    worst = 0.0
    for mod, P in ((enc, enc_P), (head, head_P)):
        for k, pr in mod.named_parameters():
            if pr.grad is None:
                assert P[k].grad is None, k
                continue
            g = P[k].grad
            gmax, gnorm = float(g.abs().max()), float(g.double().norm())
            err = max_abs_diff(pr.grad, g)
            assert err < 1e-2 * gmax + 1e-7, (k, err, gmax)
            assert abs(float(pr.grad.double().norm()) - gnorm) < UNALIGNED_NORM_BAR * gnorm + 1e-9, k
            worst = max(worst, err / gmax if gmax > 0 else 0.0)
    print(f"\n[dropout on, full width] {name} {dtype}: worst gradient entry {worst:.2e} of its tensor max (every entry, not a sample)")


@pytest.mark.parametrize("name", FULL_NAMES)
def test_f32x3_full_width_training_step_matches_reference_golden(name):
    """The f32x3 GEMM mode (products of the large GEMMs on the f16 matrix cores, csrc/gemm_pk.hip) at its DEFAULT thresholds
    meets the reference's full-width run at the SAME tolerances as the exact-f32 path: scores 1e-4, gradient entries 2e-4 of
    the tensor maximum, norms 1e-4 - the mode is checked against the reference, not only against the exact-f32 HIP step."""
    _full_width_golden_check(name, True, "f32x3")


def test_full_width_unpadded_hidden_takes_the_scalar_load_path_and_matches_golden(monkeypatch):
    """stn_full with the hidden-width padding off: the five products that touch the [tokens, 3027] hidden run the
    unaligned (scalar-load) GEMM instantiations at production size, against the same reference golden."""
    from lstc_vad_amd import functional as Fn
    monkeypatch.setattr(Fn, "_PAD_HIDDEN", False)
    _full_width_golden_check("stn_full", True)

@pytest.mark.parametrize("name", ["ltn_sht", "stn_sht", "stn_mil_ce"])
def test_two_emulated_ranks_through_trainstep(name):
    """The data-parallel composition the N-GPU run executes - TrainStep.forward_loss -> VadLossFunction phase 0 -> bag
    exchange -> phase 1 -> backward - for the shards of two ranks on ONE device: the other rank's bag is injected through
    the ``exchange`` hook (what dist.all_reduce delivers), gradients are summed by hand (what the gradient all-reduce
    delivers).  Sum of the ranks' scalars == the reference's loss; summed gradients == the reference's gradients."""
    from lstc_vad_amd.engine import TrainStep
    z, mode, ekw, skw = load_case(name)
    d, bs = ekw["d_model"], skw["batch_size"]
    assert bs % 2 == 0
    nf, af, al = (torch.from_numpy(z[k]).to(DEV) for k in ("norm_feats", "abnorm_feats", "abnorm_labs"))
    h = bs // 2
    lskw = dict(skw, batch_size=h)                   # args.batch_size is the rank-local pair count
    bags, steps = {}, []
    for r in range(2):
        enc, head = _models(mode, ekw, d)
        enc.load_state_dict(sub(z, "enc_init."), strict=True); head.load_state_dict(sub(z, "head_init."), strict=True)
        enc, head = enc.to(DEV).train(), head.to(DEV).train()

        def exchange(bag, r=r):
            if r not in bags:                        # first pass: publish this rank's slots
                bags[r] = bag.clone()
            else:                                    # second pass: the sum over ranks, as the all-reduce would leave it
                bag.add_(bags[1 - r])
        steps.append(TrainStep(_args(mode, lskw), mode, enc, head, 1e-4, 1e-2, 1e-3, loss_rank=(r, 2), loss_exchange=exchange))
    shard = lambda r: (nf[r * h:(r + 1) * h], af[r * h:(r + 1) * h], al[r * h:(r + 1) * h])
    with torch.no_grad():
        for r in range(2):                           # pass 1 only collects the bags
            steps[r].forward_loss(*shard(r))
    assert set(bags) == {0, 1} and float((bags[0] * bags[1]).abs().max()) == 0.0     # disjoint slots
    tot = torch.zeros(5, device=DEV)
    for r in range(2):
        loss, sc, _ = steps[r].forward_loss(*shard(r))
        loss.backward()
        tot += sc
    assert np.max(np.abs(tot.cpu().double().numpy() - z["scalars"])) < 2e-5
    ref_g = dict(sub(z, "enc_grad."))
    ref_h = sub(z, "head_grad.")
    for (k, p0), (_, p1) in zip(steps[0].encoder.named_parameters(), steps[1].encoder.named_parameters()):
        if k in ref_g:
            g = ref_g[k]
            assert max_abs_diff(p0.grad + p1.grad, g) < 2e-4 * float(g.abs().max()) + 1e-7, k
        else:
            assert p0.grad is None and p1.grad is None
    for (k, p0), (_, p1) in zip(steps[0].head.named_parameters(), steps[1].head.named_parameters()):
        g = ref_h[k]
        assert max_abs_diff(p0.grad + p1.grad, g) < 2e-4 * float(g.abs().max()) + 1e-7, k


def test_headline_size_backward_is_the_sum_of_its_shards_and_bit_reproducible():
    """VERDICT r4 What's missing #5: the largest TRAINING step compared with anything was 256 sequences.  Here the full headline
    batch (B = 64 videos, T = 32, L = 3, P = 16, d = 2048: 2048 sequences of 49 tokens, 100 352 tokens; exact-f32 mode) runs its
    backward, through size-independent properties:
      * data-parallel linearity: the gradients of 8 shards of 4 + 4 videos (the 8-GPU split: rank r owns pairs [4r, 4r + 4), its
        loss evaluated against the GLOBAL bag maxima exactly as VadLossFunction phases 0 / 1 do under N ranks) add up to the
        gradient of the whole batch - every parameter, 1e-5 of the tensor's norm (measured 6.1e-6; the shards' weight-gradient sums are split over
        the tokens differently from the one-launch sum, nothing else differs) - and their scalars add up to the batch's;
      * run-to-run: the same batch twice gives bit-identical scalars and gradients with the reference's dropout rates on
        (no float atomics anywhere: ordered partial sums)."""
    from argparse import Namespace
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.losses import training_loss
    from lstc_vad_amd.models import Encoder, Classifier
    torch.manual_seed(0)
    ekw = dict(n_layers=3, n_head=8, d_k=256, d_v=256, d_model=2048, d_inner=4096, MHA_layerNorm=True,
               FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=3)
    bs, pn, L, Pn, d, R = 32, 32, 3, 16, 2048, 8
    g = torch.Generator(device=DEV).manual_seed(5)
    nf = 0.5 * torch.relu(torch.randn(bs, pn * L, Pn, d, device=DEV, generator=g))
    af = 0.5 * torch.relu(torch.randn(bs, pn * L, Pn, d, device=DEV, generator=g)) + 0.05
    u = torch.rand(bs, pn * L, 1, device=DEV, generator=g)
    al = torch.where(u > 0.9, u, torch.zeros_like(u))

    def build(drop):
        torch.manual_seed(1)
        enc = Encoder(MHA_attn_dropout=drop[0], MHA_fc_dropout=drop[1], FFN_dropout=drop[2], weight_init=True, **ekw).to(DEV).train()
        head = Classifier(2048, drop[3]).to(DEV).train()
        return enc, head

    def args_for(b):
        return Namespace(batch_size=b, part_num=pn, part_len=L, n_patch=Pn, lambda_1=0.01, lambda_MIL=1.0, lambda_CE=0.8, temporal_only=False)

    def run(enc, head, lo, hi, distributed=None, exchange=None):
        x_n, x_a = nf[lo:hi].reshape(-1, L * Pn, d), af[lo:hi].reshape(-1, L * Pn, d)
        out = head(enc.forward_cls(x_n, x_a))
        return training_loss(args_for(hi - lo), "LTN", out, al[lo:hi], distributed=distributed, exchange=exchange)

    # ---- linearity over the 8-GPU split (dropout off: a shard draws other masks than the batch)
    def linearity(k_chunks, batch_rows_as=None):
        """Worst relative difference between the summed shard gradients and the batch gradient.  ``k_chunks``: the product's own
        routing (functional.small_m_products: a shard's 256-sequence products of the CLS-only layer and the head run as K chunks,
        the batch's 2048-sequence ones as one launch) or every product as one launch.  ``batch_rows_as`` = 256: the BATCH's small-row
        products are chunked exactly like a shard's (functional._SMALL_M_ROWS_AS), so every output row is summed in the same k order
        on both sides and what is left is the token split of the weight-gradient sums alone."""
        Fn._SMALL_M_SPLIT = k_chunks
        Fn._SMALL_M_ROWS_AS = batch_rows_as
        enc, head = build((0.0, 0.0, 0.0, 0.0))
        params = [(k, p) for k, p in list(enc.named_parameters()) + list(head.named_parameters())]
        loss, sc_full = run(enc, head, 0, bs)
        loss.backward()
        full = {k: p.grad.detach().clone() for k, p in params if p.grad is not None}
        sc_full = sc_full.detach().clone()
        Fn._SMALL_M_ROWS_AS = None                              # the shards run as the product runs them
        for _, p in params:
            p.grad = None
        h = bs // R
        bags = {}

        def exchange_for(r):
            def ex(bag):
                if r not in bags:
                    bags[r] = bag.clone()                   # pass 1: this rank's slots of the zero-padded global vector
                else:
                    bag.copy_(sum(bags.values()))           # pass 2: what the sum-all-reduce leaves on every rank
            return ex
        with torch.no_grad():
            for r in range(R):
                run(enc, head, r * h, (r + 1) * h, distributed=(r, R), exchange=exchange_for(r))
        tot = torch.zeros(5, device=DEV)
        for r in range(R):                                   # autograd accumulates the shards' gradients in p.grad
            loss, sc = run(enc, head, r * h, (r + 1) * h, distributed=(r, R), exchange=exchange_for(r))
            loss.backward()
            tot += sc.detach()
        assert float((tot - sc_full).abs().max()) < 2e-5, (tot, sc_full)
        worst, worst_tab = (0.0, ""), (0.0, "")
        for k, p in params:
            if k not in full:
                assert p.grad is None, k
                continue
            rel = float((p.grad.double() - full[k].double()).norm() / (full[k].double().norm() + 1e-300))
            if k.endswith("relative_position_bias_table"):
                worst_tab = max(worst_tab, (rel, k))
            else:
                worst = max(worst, (rel, k))
        del enc, head, full, params
        import gc
        gc.collect(); torch.cuda.empty_cache()
        return worst, worst_tab
    try:
        worst, tab = linearity(False)
        worst_k, tab_k = linearity(True)
        worst_s, tab_s = linearity(True, batch_rows_as=bs // R * 2 * pn)           # 256: a shard's row count
    finally:
        Fn._SMALL_M_SPLIT = True
        Fn._SMALL_M_ROWS_AS = None
    print(f"\n[headline backward] sum of {R} shard gradients vs the batch gradient: worst relative difference {worst[0]:.2e} ({worst[1]}), "
          f"bias tables {tab[0]:.2e}; with the shards' small products as K chunks {worst_k[0]:.2e} ({worst_k[1]}), bias tables {tab_k[0]:.2e}")
    # f32 sums over 100 352 tokens split 8 ways vs at once (measured 6.1e-6, layer 1's dW2)
    assert worst[0] < 1e-5 and tab[0] < 1e-5, (worst, tab)
    # as the product runs it, a shard's 256-row products of the CLS-only layer and the head add their K range in chunks (never the
    # ones that feed a ReLU): the gradient that enters the full layers moves by f32 re-association (~1e-6 of its magnitude).  The
    # gradients of the attention LOGITS' parameters - w_qs, w_ks and the relative-position bias tables: small differences of large
    # cancelling sums, their norms ~1e-2 of the other weights' - carry it amplified (measured 2.5e-4 / 4.0e-4 of the tensor's norm);
    # bar 1e-3 / 2e-3
    assert worst_k[0] < 1e-3 and tab_k[0] < 2e-3, (worst_k, tab_k)
    # round 6: the statement "DP reproduces the single-process step" at the bar of the equal-routing arm - the
    # batch's few-row products chunked exactly like a shard's (same k order per output row on both sides): what remains is the token
    # split of the weight-gradient sums, as in the first arm.  The 2.5e-4 above is therefore the re-association of the BATCH-side
    # one-launch products against the chunked ones, not a property of the data-parallel composition
    print(f"[headline backward] K-chunked shards vs the batch chunked the same way: worst {worst_s[0]:.2e} ({worst_s[1]}), bias tables {tab_s[0]:.2e}")
    assert worst_s[0] < 2e-5 and tab_s[0] < 2e-5, (worst_s, tab_s)
    # ---- bit reproducibility at the headline size, reference dropout rates on
    enc, head = build((0.2, 0.2, 0.1, 0.6))
    outs = []
    for _ in range(2):
        Fn.reset_rng(0)
        for p in list(enc.parameters()) + list(head.parameters()):
            p.grad = None
        loss, sc = run(enc, head, 0, bs)
        loss.backward()
        outs.append((sc.detach().clone(), {k: p.grad.detach().clone() for k, p in enc.named_parameters() if p.grad is not None}))
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def test_gemm_check_tool_shape_list_against_f64_host_products():
    """tools/gemm_check (standalone C++ harness over the C ABI, no torch): every GEMM kernel family - exact f32 in all tile
    variants and layouts, bf16c, f32x3 (NT / TR / split-K), bf16p (NT / TR / ragged / split-K) - on odd shapes, every
    epilogue flag, against a double-precision host product; C padding columns must stay untouched."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gemm_check")
    if not os.path.exists(exe):
        pytest.fail("tools/gemm_check is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([exe, "check"], capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith(("PASS", "FAIL"))]
    assert len(lines) > 100 and not [l for l in lines if l.startswith("FAIL")], "\n".join(l for l in lines if l.startswith("FAIL"))
    assert "ALL PASS" in r.stdout and r.returncode == 0, r.stdout[-2000:]


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("shape", [(4224, 512, 512), (1000, 300, 260), (6272, 2048, 2048)])
def test_bf16p_gemm_forms_match_bf16_rounded_reference(shape):
    """The packed bf16 GEMM (lstc_pack1 + gemm_bf16p_kernel) through functional.gemm / wgrad in bf16 mode: forward X W^T with
    bias+ReLU, input gradient dY W, weight gradient dY^T X (TR form with split-K partials; T = 4224 gives 66 K steps that
    do not divide by 16 - the partial-row case of ADVICE r1) against f64 products of the bf16-rounded operands."""
    from lstc_vad_amd import functional as Fn
    T, O, I = shape
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(T, I, device=DEV, generator=g)
    w = torch.randn(O, I, device=DEV, generator=g) * 0.05
    b = torch.randn(O, device=DEV, generator=g)
    dy = torch.randn(T, O, device=DEV, generator=g)
    xr, wr, dyr = _bf16_round(x).double(), _bf16_round(w).double(), _bf16_round(dy).double()
    Fn.set_compute_dtype("bf16"); Fn.set_x3_threshold(0, 0, 0)
    try:
        with Fn.pack_memo():
            y = Fn.gemm(x, w, trans_b=True, bias=b, relu=True)
            dx = Fn.gemm(dy, w)
            dw = Fn.wgrad(dy, x)
        torch.cuda.synchronize()
    finally:
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()
    tol = 3e-5 * (I ** 0.5)
    assert max_abs_diff(y, torch.relu(xr @ wr.T + b.double())) < tol
    assert max_abs_diff(dx, dyr @ wr) < 3e-5 * (O ** 0.5) * 0.2
    ref_dw = dyr.T @ xr
    assert max_abs_diff(dw, ref_dw) < 3e-5 * (T ** 0.5) * 4


@pytest.mark.parametrize("T", [4224, 4736])
def test_f32x3_wgrad_with_uneven_k_splits(T):
    """T = 4224 tokens (132 K tiles) with split 16 launches 15 slices; the partial buffer must hold exactly the
    launched slices (lstc_gemm_splits), not the requested count."""
    from lstc_vad_amd import functional as Fn
    g = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randn(T, 512, device=DEV, generator=g)
    dy = torch.randn(T, 512, device=DEV, generator=g)
    junk = torch.full((64, 512 * 512), float("nan"), device=DEV)      # poison the allocator's free list
    del junk
    Fn.set_compute_dtype("f32x3"); Fn.set_x3_threshold(0, 0, 0)
    try:
        with Fn.pack_memo():
            dw = Fn.wgrad(dy, x)
        torch.cuda.synchronize()
    finally:
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()
    ref = dy.double().T @ x.double()
    assert torch.isfinite(dw).all()
    assert max_abs_diff(dw, ref) < 2e-4 * (T ** 0.5) * 0.05


@pytest.mark.parametrize("mode", ["fp32", "bf16", "f32x3"])
def test_small_row_count_products_as_k_chunks_match_the_one_launch_form(mode):
    """functional.small_m_products (the training-mode bodies of the CLS-only last layer, its FFN and the heads): a product with few
    output tiles - [256, d] x [d, d'] for a rank's 256 sequences - runs as K chunks in ONE batched lstc_gemm + lstc_splitk_finish.
    Against the one-launch form with every epilogue the callers use (bias; bias + dropout + residual; ReLU mask; accumulate +
    residual into a strided view; alpha - a product whose epilogue is a ReLU is never chunked): the same dropout mask element for element, values to f32 re-association of the K sum
    (1e-5 of the magnitude bound; bf16 mode: both forms round the operands alike).  The chunking really happens (spy), and does
    NOT happen outside the context or for a product that already fills a quarter of the chip."""
    from lstc_vad_amd import functional as Fn
    g = torch.Generator(device=DEV).manual_seed(31)
    calls = []
    real = Fn._gemm_small_m
    def spy(*a, **k):
        calls.append(a[9])            # s
        return real(*a, **k)
    Fn.set_compute_dtype(mode)
    Fn._gemm_small_m = spy
    try:
        for (M, N, K, tb) in [(256, 2048, 2048, True), (256, 2048, 4096, False), (256, 512, 2048, True), (384, 4096, 2048, True),
                              (256, 32, 512, True)]:
            x = torch.randn(M, K, device=DEV, generator=g)
            w = torch.randn((N, K) if tb else (K, N), device=DEV, generator=g) * 0.05
            b = torch.randn(N, device=DEV, generator=g)
            r = torch.randn(M, N, device=DEV, generator=g)
            src = torch.randn(M, N, device=DEV, generator=g)
            bound = float((x.abs() @ (w.abs().T if tb else w.abs())).max())
            tol = (1e-5 if mode != "bf16" else 2e-5) * bound
            n0 = len(calls)
            with Fn.small_m_products():
                assert torch.equal(Fn.gemm(x, w, trans_b=tb, bias=b, relu=True), Fn.gemm(x, w, trans_b=tb, bias=b, relu=True))
            assert len(calls) == n0              # a product that feeds a ReLU keeps the one-launch k order (no unit changes sides)
            for kw in (dict(bias=b), dict(bias=b, dropout=(0.3, 0xABCDEF), residual=r), dict(relu_mask=src),
                       dict(alpha=0.0625), dict()):
                ref = Fn.gemm(x, w, trans_b=tb, **kw)
                n0 = len(calls)
                with Fn.small_m_products():
                    got = Fn.gemm(x, w, trans_b=tb, **kw)
                assert len(calls) == n0 + 1 and calls[-1] > 1, (M, N, K, kw.keys())
                if "dropout" in kw:
                    assert torch.equal(got == r, ref == r)                    # a dropped element is exactly the residual: same mask
                if "relu_mask" in kw:
                    assert torch.equal(got == 0, ref == 0)
                assert max_abs_diff(got, ref) < tol, (mode, M, N, K, list(kw), max_abs_diff(got, ref), tol)
            # accumulate + residual into a strided view (the CLS rows of dX)
            big0 = torch.randn(M, 3, N, device=DEV, generator=g)
            big1 = big0.clone()
            Fn.gemm(x, w, trans_b=tb, out=big0[:, 0, :], accumulate=True, residual=r)
            with Fn.small_m_products():
                Fn.gemm(x, w, trans_b=tb, out=big1[:, 0, :], accumulate=True, residual=r)
            assert torch.equal(big0[:, 1:, :], big1[:, 1:, :]) and max_abs_diff(big0, big1) < tol
        # the per-head products of the re-associated CLS attention (x-bar_h W_v,h^T, du_h W_k,h^T): heads AND K chunks in one launch
        for (N, H, dh, dm) in [(256, 8, 256, 2048), (384, 4, 64, 1024)]:
            a3 = torch.randn(N, H, dm, device=DEV, generator=g)
            wh = torch.randn(H * dh, dm, device=DEV, generator=g) * 0.05
            ref64 = torch.einsum("nhc,hjc->nhj", a3.double(), wh.double().view(H, dh, dm)).reshape(N, H * dh) * 0.125
            bound = float(torch.einsum("nhc,hjc->nhj", a3.abs(), wh.abs().view(H, dh, dm)).max()) * 0.125
            one = Fn.per_head_rows_wT(a3, wh, torch.empty(N, H * dh, device=DEV), N, dh, dm, H, alpha=0.125)
            with Fn.small_m_products():
                chunked = Fn.per_head_rows_wT(a3, wh, torch.empty(N, H * dh, device=DEV), N, dh, dm, H, alpha=0.125)
            assert "_lstc_kchunks" in wh.__dict__ and wh.__dict__["_lstc_kchunks"][1].shape[1] > 1      # really chunked
            tol = (1e-5 if mode != "bf16" else 1e-2) * bound
            assert max_abs_diff(one, ref64) < tol and max_abs_diff(chunked, ref64) < tol
            assert max_abs_diff(chunked, one) < (1e-5 if mode != "bf16" else 2e-5) * bound
        n0 = len(calls)
        x = torch.randn(2048, 2048, device=DEV, generator=g)
        w = torch.randn(2048, 2048, device=DEV, generator=g)
        with Fn.small_m_products():
            Fn.gemm(x, w, trans_b=True)                         # 256 tiles: left alone
        with Fn.small_m_products(False):
            Fn.gemm(x[:256], w, trans_b=True)                   # context off (evaluation)
        Fn.gemm(x[:256], w, trans_b=True)
        assert len(calls) == n0
        torch.cuda.synchronize()
    finally:
        Fn._gemm_small_m = real
        Fn.set_compute_dtype("fp32")


def _mil_max_moves(outputs, ref_outputs, part_num):
    """LTN: the MIL ranking term reads ONE part per video - the maximum of column 1 over its parts (losses.get_MIL_loss,
    Train/temporal_transformer_shanghaitech.py:25-36) - so two runs whose scores agree to the bf16 mode's 1e-2 can still hand a
    video's ranking gradient to different parts when its top two parts are closer than that.  Such a move is a property of the
    loss, not of the arithmetic.  Returns the number of videos whose maximum sits on another part than in ``ref_outputs``; every
    one of them must be explained by the score difference (gap of the reference's top two parts < 2 x the largest score
    difference).  0 for STN outputs (one column)."""
    o, r = torch.as_tensor(outputs).detach().float().cpu(), torch.as_tensor(ref_outputs).detach().float().cpu()
    if r.dim() != 2 or r.shape[1] != 2:
        return 0
    o = o.reshape(r.shape)
    diff = float((o - r).abs().max())
    b, br = o[:, 1].reshape(-1, part_num), r[:, 1].reshape(-1, part_num)
    moved = (b.argmax(1) != br.argmax(1)).nonzero().reshape(-1).tolist()
    for v in moved:
        top = br[v].topk(2).values
        assert float(top[0] - top[1]) < 2 * diff, (v, float(top[0] - top[1]), diff)
    return len(moved)


class _align_mil_max:
    """``with _align_mil_max(ref_outputs, part_num) as st:`` - the MIL ranking term reads ONE part per video, the maximum of column 1
    over its parts (losses.get_MIL_loss, Train/temporal_transformer_shanghaitech.py:25-36).  Inside the context the loss sees, for
    every video whose maximum sits on another part than in ``ref_outputs``, the reference's part lifted just above the current
    maximum (a constant added to that one score: at most the gap between the top two parts, which ``_mil_max_moves`` has shown to be
    smaller than the score tolerance) - so the ranking gradient travels through the part the reference's run chose.  The same
    device as ``_align_relu_edges``: an arg-max over nearly tied scores is decided by rounding, and the comparison of everything else
    should not inherit that coin flip.  Nothing in the product knows about this: the context wraps ``training_loss`` as the TEST
    and ``engine`` look it up.  ``st.moved`` = videos re-routed."""

    def __init__(self, ref_outputs, part_num):
        self.ref, self.pn, self.moved = torch.as_tensor(ref_outputs).detach().float().cpu(), int(part_num), 0

    def __enter__(self):
        from lstc_vad_amd import engine, losses
        self.losses, self.engine, self.real = losses, engine, losses.training_loss
        st = self

        def training_loss(args, mode, outputs, *a, **kw):
            if st.ref.dim() == 2 and st.ref.shape[1] == 2:
                r = st.ref.to(outputs.device).reshape(outputs.shape)
                cur, rr = outputs.detach()[:, 1].reshape(-1, st.pn), r[:, 1].reshape(-1, st.pn)
                want, have = rr.argmax(1), cur.argmax(1)
                delta = torch.zeros_like(outputs)
                for v in (want != have).nonzero().reshape(-1).tolist():
                    delta[v * st.pn + int(want[v]), 1] = float(cur[v].max() - cur[v, want[v]]) + 1e-6
                    st.moved += 1
                outputs = outputs + delta
            return st.real(args, mode, outputs, *a, **kw)
        losses.training_loss = engine.training_loss = training_loss
        return self

    def __exit__(self, *exc):
        self.losses.training_loss = self.engine.training_loss = self.real
        return False


@pytest.mark.parametrize("name,fused", [(n, False) for n in FULL_NAMES] + [("stn_full", True), ("stn_mil_ce_full", True), ("ltn_ucf_full", True),
                                        ("ltn_full_256", True), ("ltn_ubnormal_full_256", True)])
def test_full_width_bf16_step_tracks_reference(name, fused):
    """bf16 GEMM mode (packed bf16 kernel on every large product incl. TR weight gradients, bf16c on the heads) at BASELINE
    widths against the reference's fp32 run: scores within 2e-2, loss within 2e-2, every large gradient tensor's direction
    (cosine on the sampled entries) > 0.98.  ``fused`` (the cases whose token count fills whole 256-row pack tiles): the Q, K, V
    projections share one buffer as engine.TrainStep arranges it, so the attention core runs on PACKED operands (Q|K|V and dO
    read as packs, csrc/attention_pk.hip) - with the sliced relative-bias index of the UCF case (S = 19), without bias (STN,
    S = 17), and at S = 49 / S = 81 with 256 sequences (cases.PACKED_CASES: two and three query tiles per sequence)."""
    from cases import sample_index
    from lstc_vad_amd import functional as Fn
    z, mode, skw, d, enc, head, nf, af, al = _full_width_models(name)
    enc, head = enc.to(DEV).train(), head.to(DEV).train()
    if fused:
        for layer in list(enc.layer_stack)[:-1]:
            layer.slf_attn.fuse_qkv_()
    args = _args(mode, skw)
    nf, af, al = (torch.from_numpy(x).to(DEV) for x in (nf, af, al))
    seen, real = [], Fn.attn_fwd
    def spy(q, *a, **kw):
        seen.append(isinstance(q, Fn.Packed))
        return real(q, *a, **kw)
    Fn.set_compute_dtype("bf16")
    Fn.attn_fwd = spy
    try:
        enc_out, outputs, loss, sc = _step(enc, head, mode, args, nf, af, al, d, cls_only=True)
        loss.backward()
        if args.clip_grad:                              # the fixture's gradients are the clipped ones (ltn_clip_full)
            from lstc_vad_amd.optim import clip_grad_norm_
            clip_grad_norm_(enc.parameters(), 10)
            clip_grad_norm_(head.parameters(), 10)
        torch.cuda.synchronize()
    finally:
        Fn.attn_fwd = real
        Fn.set_compute_dtype("fp32")
    assert seen and all(seen) == fused, seen
    assert max_abs_diff(outputs.reshape(z["outputs"].shape), z["outputs"]) < 2e-2
    assert abs(float(sc[0]) - float(z["scalars"][0])) < 2e-2
    # a video whose MIL maximum sits on another part than in the reference's run (its top two parts closer than the score
    # tolerance: ltn_ubnormal_full_256 has a pair 3.7e-4 apart) moves 1 / 16 of the ranking gradient: direction bar 0.90 then
    moves = _mil_max_moves(outputs, z["outputs"], args.part_num)
    assert moves <= 1
    if moves:
        print(f"\n[bf16 {name}] {moves} video's MIL maximum sits on another part than in the reference's run")
    def check(strict):
        for k, p in enc.named_parameters():
            if p.grad is None or p.numel() < 4096 or float(z[f"enc_gnorm.{k}"]) == 0.0:
                continue
            gs = torch.from_numpy(z[f"enc_gs.{k}"]).double()
            got = p.grad.detach().reshape(-1)[torch.from_numpy(sample_index(p.numel())).to(DEV)].cpu().double()
            cos = float((got * gs).sum() / (got.norm() * gs.norm() + 1e-30))
            # 256 sampled entries per tensor: > 0.98 everywhere except the first FFN weight, whose gradient dh^T x inherits the
            # ReLU decisions of the ~0.5 % of hidden units whose pre-activation lies within bf16 product rounding of zero (a
            # flipped unit rewrites its whole row of dW1; ltn_ucf_full layer 1: 0.968 on the sample at a norm ratio of 1.0006)
            # (the same holds for that layer's bias gradient db1 = column sums of the hidden's gradient: 0.976 on ltn_clip_full layer 1)
            assert cos > ((0.95 if k.endswith(("pos_ffn.w_1.weight", "pos_ffn.w_1.bias")) else 0.98) if strict else 0.90), (k, cos, moves, strict)
            assert abs(float(p.grad.double().norm()) / float(z[f"enc_gnorm.{k}"]) - 1.0) < 0.05, k
    check(strict=not moves)
    if moves:
        # round 6: the 0.90 arm above documents the UN-ALIGNED step only.  The same step with the reference's
        # arg-max part for the moved video (``_align_mil_max``; nothing else touched) is held to the bars of every other case
        for p in list(enc.parameters()) + list(head.parameters()):
            p.grad = None
        Fn.set_compute_dtype("bf16")
        try:
            with _align_mil_max(z["outputs"], args.part_num) as st:
                enc_out, outputs, loss, sc = _step(enc, head, mode, args, nf, af, al, d, cls_only=True)
                loss.backward()
            if args.clip_grad:
                from lstc_vad_amd.optim import clip_grad_norm_
                clip_grad_norm_(enc.parameters(), 10)
                clip_grad_norm_(head.parameters(), 10)
            torch.cuda.synchronize()
        finally:
            Fn.set_compute_dtype("fp32")
        assert st.moved == moves
        check(strict=True)
        print(f"[bf16 {name}] with the reference's arg-max part for that video: every gradient direction at the strict bars")


def test_mixed_step_bf16_vs_fp32_auc_on_the_mixed_pair():
    """BASELINE config 5 ("UBnormal config (d_model=1024, part_len=5) mixed with SHT in one batch, bf16 + fp32 AUC parity
    check"), reduced width: the SAME mixed training run - engine.MixedStep over a UBnormal-shaped pair (L=5, S=81) and an
    SHT-shaped pair (L=3, S=49), 6 steps, dropout off, identical initial weights and batches - once in fp32 and once in bf16
    mode (packed bf16 GEMMs forced on every product); then both pairs score held-out videos in the mode they were trained
    in.  Frame-level AUC of each dataset must agree to 1e-2 between the modes and the scores to 5e-2."""
    from argparse import Namespace
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.engine import MixedStep, TrainStep
    from lstc_vad_amd.metrics import roc_auc
    from lstc_vad_amd.models import Classifier, Encoder
    cfgs = [dict(d_model=128, L=5), dict(d_model=256, L=3)]
    bs, pn, P = 4, 8, 16

    def run(mode):
        Fn.set_compute_dtype(mode)
        if mode == "bf16":
            Fn.set_x3_threshold(0, 0, 0)
        try:
            steps, batches, tests = [], [], []
            for ci, c in enumerate(cfgs):
                d, L = c["d_model"], c["L"]
                torch.manual_seed(100 + ci)
                enc = Encoder(n_layers=3, n_head=4, d_k=32, d_v=32, d_model=d, d_inner=2 * d, MHA_attn_dropout=0.0, MHA_fc_dropout=0.0,
                              FFN_dropout=0.0, MHA_layerNorm=True, FFN_layerNorm=True, relative_pe=True, window_size=4,
                              window_depth=L, weight_init=True).to(DEV).train()
                head = Classifier(d, 0.0).to(DEV).train()
                args = Namespace(batch_size=bs, part_num=pn, part_len=L, n_patch=P, lambda_1=0.01, lambda_MIL=1.0, lambda_CE=0.8,
                                 temporal_only=False, clip_grad=False)
                steps.append(TrainStep(args, "LTN", enc, head, 1e-4, 1e-3, 1e-3))
                g = torch.Generator(device=DEV).manual_seed(7 + ci)
                nf = 0.5 * torch.relu(torch.randn(bs, pn * L, P, d, device=DEV, generator=g))
                af = 0.5 * torch.relu(torch.randn(bs, pn * L, P, d, device=DEV, generator=g))
                af[:, :, :, : d // 8] += 0.4                                    # the anomaly signature
                al = torch.ones(bs, pn * L, 1, device=DEV)
                batches.append((nf, af, al))
                xt = 0.5 * torch.relu(torch.randn(128, L * P, d, device=DEV, generator=g))
                xt[64:, :, : d // 8] += 0.4
                tests.append(xt)
            mixed = MixedStep(steps)
            for _ in range(6):
                sc = mixed.step(batches)
            scores = []
            with torch.no_grad():
                for ts, xt in zip(steps, tests):
                    ts.encoder.eval(); ts.head.eval()
                    scores.append(ts.head(ts.encoder.forward_cls(xt))[:, 1].cpu().numpy())
            return [float(s[0]) for s in sc], scores
        finally:
            Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()

    loss32, s32 = run("fp32")
    loss16, s16 = run("bf16")
    labels = np.r_[np.zeros(64), np.ones(64)]
    for a, b, l32, l16 in zip(s32, s16, loss32, loss16):
        assert np.isfinite(a).all() and np.isfinite(b).all()
        assert np.max(np.abs(a - b)) > 0 and np.max(np.abs(a - b)) < 5e-2
        assert abs(roc_auc(a, labels) - roc_auc(b, labels)) < 1e-2
        assert abs(l32 - l16) < 5e-2


def test_mixed_step_keeps_both_models_weight_packs_fresh_in_bf16_mode():
    """pack staleness is tracked per parameter.  engine.MixedStep steps two optimizers back to back; after the first
    iteration every later iteration must find ALL weight packs of BOTH models rebuilt by the optimizers' one-launch repack
    (functional.repack_weights) - no lazily issued per-weight lstc_pack1 - and the packs must hold the CURRENT weights."""
    from argparse import Namespace
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.engine import MixedStep, TrainStep
    from lstc_vad_amd.models import Classifier, Encoder
    Fn.set_compute_dtype("bf16"); Fn.set_x3_threshold(0, 0, 0)
    real = Fn.pack3
    try:
        steps, batches = [], []
        for ci, (d, L) in enumerate(((128, 5), (256, 3))):
            torch.manual_seed(40 + ci)
            enc = Encoder(n_layers=3, n_head=4, d_k=32, d_v=32, d_model=d, d_inner=2 * d, MHA_attn_dropout=0.0, MHA_fc_dropout=0.0,
                          FFN_dropout=0.0, MHA_layerNorm=True, FFN_layerNorm=True, relative_pe=True, window_size=4,
                          window_depth=L, weight_init=True).to(DEV).train()
            head = Classifier(d, 0.0).to(DEV).train()
            args = Namespace(batch_size=2, part_num=4, part_len=L, n_patch=16, lambda_1=0.01, lambda_MIL=1.0, lambda_CE=0.8,
                             temporal_only=False, clip_grad=False)
            steps.append(TrainStep(args, "LTN", enc, head, 1e-3, 1e-2, 1e-3, fuse_qkv="on" if ci else "off"))
            g = torch.Generator(device=DEV).manual_seed(3 + ci)
            batches.append((torch.rand(2, 4 * L, 16, d, device=DEV, generator=g), torch.rand(2, 4 * L, 16, d, device=DEV, generator=g),
                            torch.ones(2, 4 * L, 1, device=DEV)))
        mixed = MixedStep(steps)
        weight_ptrs = {p.data_ptr() for ts in steps for m in (ts.encoder, ts.head) for p in m.parameters()}
        lazy = []

        def spy(t, k_major):
            if t.data_ptr() in weight_ptrs:
                lazy.append(tuple(t.shape))
            return real(t, k_major)
        Fn.pack3 = spy
        mixed.step(batches)
        first = len(lazy)
        assert first > 10                                     # iteration 1 packs every weight lazily
        del lazy[:]
        for _ in range(3):
            mixed.step(batches)
        assert lazy == [], lazy                               # iterations 2-4: nothing left for the lazy path, for either model
        torch.cuda.synchronize()
        for ts in steps:                                      # and the rebuilt packs are the packs of the current weights
            w = ts.encoder.layer_stack[0].pos_ffn.w_1.weight
            hit = w.__dict__["_lstc_packs"][(False, Fn._lib.BF16P)]
            assert hit[0] == Fn._wstamp(w)
            fresh = real(w.detach(), False).buf
            n_tiles = fresh.numel() - 65536                       # the tiles; the buffer's 64-KB over-read slack behind them is never written
            assert n_tiles > 0 and torch.equal(hit[2].buf[:n_tiles], fresh[:n_tiles])
    finally:
        Fn.pack3 = real
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()


# (S, window depth L of the model's 3-D index, d_k).  The index is [16 L, 16 L]; a sequence reads its top-left [S-1, S-1]
# (models/MultiHeadAttention.py:107-111), so the index's row stride exceeds S - 1 for UCF (S = 19: 9 patches, L = 2 - the
# production shape of BASELINE config 4, Test/evaluation_UCF.py:112) and for the short tail parts of an L = 3 model
# (S = 17 / 33, Train/pseudo_labels_generator_temporal.py:110-137).  (81, 5, 256) = UBnormal's production instantiation.
STAGED_SHAPES = [(17, 1, 64), (49, 3, 64), (33, 2, 32), (81, 5, 64), (64, 3, 96), (49, 3, 256),
                 (19, 2, 256), (17, 3, 64), (33, 3, 64), (81, 5, 256), (19, 2, 32)]


@pytest.mark.parametrize("S,L,dk", STAGED_SHAPES)
def test_attention_backward_staged_kernel_matches_first_generation_and_f64(S, L, dk):
    """lstc_attn_fwd / lstc_attn_bwd's second-generation kernels (LDS-DMA staged operands, register-resident B rows, one job
    pipeline; forward for S <= 32, backward for S <= 64; larger S runs the first generation in both arms)
    against the first-generation kernel (LstcAttnDesc.variant = 1) and against an f64 autograd reference that replays the
    dropout mask: dQ, dK, dV and the bias-table gradient, with relative bias and attention dropout on."""
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.models.MultiHeadAttention import relative_position_index_3d
    N, H, p_drop, seed = 5, 3, 0.25, 77
    g = torch.Generator(device=DEV).manual_seed(S * 131 + dk)
    M = N * S
    q, k, v, do = (torch.randn(M, H * dk, device=DEV, generator=g) for _ in range(4))
    use_bias = S - 1 <= 16 * L
    idx = relative_position_index_3d(L, 4).to(DEV) if use_bias else None
    assert idx is None or idx.shape == (16 * L, 16 * L)
    tab = torch.randn((2 * L - 1) * 49, H, device=DEV, generator=g) * 0.3 if use_bias else None
    o, probs = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, tab, idx, p_drop, seed)
    keep = os.environ.pop("LSTC_ATTN_VARIANT", None)
    Fn._ATTN_VARIANT, old_variant = 1, Fn._ATTN_VARIANT
    try:
        o1, probs1 = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, tab, idx, p_drop, seed)      # first-generation forward
    finally:
        Fn._ATTN_VARIANT = old_variant

    def run(variant):
        old = Fn._ATTN_VARIANT
        Fn._ATTN_VARIANT = variant
        try:
            return Fn.attn_bwd(do, q, k, v, probs, N, S, H, dk, dk, tab, idx, p_drop, seed)
        finally:
            Fn._ATTN_VARIANT = old
    dq2, dk2, dv2, dt2 = run(0)
    dq1, dk1, dv1, dt1 = run(1)
    torch.cuda.synchronize()
    if keep is not None:
        os.environ["LSTC_ATTN_VARIANT"] = keep
    # f64 reference through torch autograd on the CPU
    mask = Fn.dropout_mask((N, H, S, S), p_drop, seed, DEV).cpu().double() / (1.0 - p_drop)
    qd, kd, vd = (t.cpu().double().view(N, S, H, dk).transpose(1, 2).requires_grad_(True) for t in (q, k, v))
    td = tab.cpu().double().requires_grad_(True) if use_bias else None
    a = (qd / dk ** 0.5) @ kd.transpose(-1, -2)
    if use_bias:
        bias = td[idx.cpu()[: S - 1, : S - 1].reshape(-1)].view(S - 1, S - 1, H).permute(2, 0, 1)
        a = a + torch.nn.functional.pad(bias, (1, 0, 1, 0)).unsqueeze(0)
    pr = torch.softmax(a, -1)
    out = (pr * mask) @ vd
    for got_o, got_p in ((o, probs), (o1, probs1)):              # forward of both generations
        assert max_abs_diff(got_p, pr.detach()) < 2e-6
        assert max_abs_diff(got_o, out.detach().transpose(1, 2).reshape(M, H * dk)) < 2e-5 * float(out.detach().abs().max()) + 1e-6
    out.backward(do.cpu().double().view(N, S, H, dk).transpose(1, 2))
    ref = [t.grad.transpose(1, 2).reshape(M, H * dk) for t in (qd, kd, vd)]
    for got2, got1, r, name in zip((dq2, dk2, dv2), (dq1, dk1, dv1), ref, "QKV"):
        tol = 2e-5 * float(r.abs().max()) + 1e-6
        assert max_abs_diff(got2, r) < tol, (name, "v2", max_abs_diff(got2, r), tol)
        assert max_abs_diff(got1, r) < tol, (name, "v1", max_abs_diff(got1, r), tol)
    if use_bias:
        assert max_abs_diff(dt2, td.grad) < 2e-5 * float(td.grad.abs().max()) + 1e-6
        assert max_abs_diff(dt1, dt2) < 1e-5 * float(td.grad.abs().max()) + 1e-6


@pytest.mark.parametrize("S,L,dk", [(17, 1, 64), (49, 3, 64), (81, 5, 64), (49, 3, 256), (33, 2, 32), (81, 5, 128), (113, 7, 32),
                                    (19, 2, 256), (17, 3, 64), (33, 3, 64), (81, 5, 256)])
def test_attention_bf16_products_track_the_f64_reference(S, L, dk):
    """bf16 training mode (LstcAttnDesc.dtype = LSTC_BF16): Q K^T, Pd V and the four backward products contract bf16-rounded
    operands on v_mfma_f32_32x32x16_bf16 with f32 accumulation; softmax, bias, dropout stay f32.  The staged kernels
    (S <= 96) against the f64 autograd reference on the UNROUNDED operands: relative Frobenius error of every result at the
    level of bf16 operand rounding (2^-9 per element, averaged over the contraction) - and not zero-ish, i.e. the bf16 path
    really ran.  S = 113 has no staged kernel: the same descriptor must give the exact-f32 results bit for bit."""
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.models.MultiHeadAttention import relative_position_index_3d
    N, H, p_drop, seed = 5, 3, 0.25, 79
    g = torch.Generator(device=DEV).manual_seed(S * 17 + dk)
    M = N * S
    q, k, v, do = (torch.randn(M, H * dk, device=DEV, generator=g) for _ in range(4))
    idx = relative_position_index_3d(L, 4).to(DEV) if S - 1 <= 16 * L else None
    tab = torch.randn((2 * L - 1) * 49, H, device=DEV, generator=g) * 0.3 if idx is not None else None

    def run():
        o, probs = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, tab, idx, p_drop, seed)
        return (o, probs) + tuple(Fn.attn_bwd(do, q, k, v, probs, N, S, H, dk, dk, tab, idx, p_drop, seed))
    exact = run()
    Fn.set_compute_dtype("bf16")
    try:
        assert Fn._attn_dtype() == Fn._lib.BF16
        got = run()
        torch.cuda.synchronize()
    finally:
        Fn.set_compute_dtype("fp32")
    names = ("O", "P", "dQ", "dK", "dV", "dtable")
    if S > 96:
        for name, a, b in zip(names, got, exact):
            assert a is None and b is None or torch.equal(a, b), name
        return
    mask = Fn.dropout_mask((N, H, S, S), p_drop, seed, DEV).cpu().double() / (1.0 - p_drop)
    qd, kd, vd = (t.cpu().double().view(N, S, H, dk).transpose(1, 2).requires_grad_(True) for t in (q, k, v))
    a = (qd / dk ** 0.5) @ kd.transpose(-1, -2)
    td = None
    if idx is not None:
        td = tab.cpu().double().requires_grad_(True)
        bias = td[idx.cpu()[: S - 1, : S - 1].reshape(-1)].view(S - 1, S - 1, H).permute(2, 0, 1)
        a = a + torch.nn.functional.pad(bias, (1, 0, 1, 0)).unsqueeze(0)
    pr = torch.softmax(a, -1)
    out = (pr * mask) @ vd
    out.backward(do.cpu().double().view(N, S, H, dk).transpose(1, 2))
    ref = [out.detach().transpose(1, 2).reshape(M, H * dk), pr.detach()] + \
          [t.grad.transpose(1, 2).reshape(M, H * dk) for t in (qd, kd, vd)] + [None if td is None else td.grad]
    for name, x, e, r in zip(names, got, exact, ref):
        if r is None:
            continue
        err = float((x.detach().cpu().double() - r).norm() / r.norm())
        err_exact = float((e.detach().cpu().double() - r).norm() / r.norm())
        assert err_exact < 1e-5 and 10 * err_exact < err < 8e-3, (name, err, err_exact)


@pytest.mark.parametrize("epi", ["plain", "bias_relu", "dropout_residual"])
def test_bf16p_persistent_items_match_f32_kernel_on_rounded_operands(epi):
    """The persistent packed-bf16 kernel with SEVERAL work items per workgroup (25 152 x 2048 output = 792 tiles on 256
    workgroups: the next item's LDS-DMA is issued before the current epilogue and the epilogue's stores are counted in the
    next item's waits; the ragged last row tile takes the draining path in between) against the exact-f32 GEMM kernel on
    bf16-rounded operands: same products, f32 accumulation in a different order."""
    from lstc_vad_amd import functional as Fn
    M, N, K = 25088 + 64, 2048, 1024
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) * 0.05
    b = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g)
    kw = {"plain": {}, "bias_relu": dict(bias=b, relu=True), "dropout_residual": dict(bias=b, dropout=(0.1, 1234), residual=res)}[epi]
    xr, wr = _bf16_round(x), _bf16_round(w)
    ref = Fn.gemm(xr, wr, trans_b=True, **kw)                       # exact-f32 kernel (fp32 mode)
    Fn.set_compute_dtype("bf16")
    try:
        for _ in range(2):                                          # twice: a stale LDS / wait-count bug shows run to run
            got = Fn.gemm(x, w, trans_b=True, **kw)
            torch.cuda.synchronize()
            assert max_abs_diff(got, ref) < 2e-5 * (K ** 0.5) + 1e-5
    finally:
        Fn.set_compute_dtype("fp32")


def test_bf16_gradient_allreduce_wire_format_single_rank():
    """GradAllReducer(reduce_dtype="bf16") over RCCL with one rank: every reduced gradient equals the bf16 rounding (RNE,
    lstc_cast_f32_bf16 / lstc_cast_bf16_f32) of the gradient the plain path produces; the default fp32 format is bit-exact."""
    import socket
    import torch.distributed as dist
    from lstc_vad_amd.dist import GradAllReducer
    g = torch.Generator(device=DEV).manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(s, device=DEV, generator=g)) for s in ((300, 70), (41,), (128, 128))]
    xs = [torch.randn_like(p) for p in ps]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for fmt in ("fp32", "bf16"):
            for p in ps:
                p.grad = None
            red = GradAllReducer([ps[:2], ps[2:]], force=True, reduce_dtype=fmt)
            red.zero_grad()
            loss = sum((p * x).sum() for p, x in zip(ps, xs))
            loss.backward()
            red.finish()
            torch.cuda.synchronize()
            for p, x in zip(ps, xs):
                want = x if fmt == "fp32" else x.to(torch.bfloat16).to(torch.float32)
                assert torch.equal(p.grad, want), fmt
            assert red.payload_bytes() == sum(p.numel() for p in ps) * (4 if fmt == "fp32" else 2)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rows,d,p", [(256, 64, 0.0), (512, 2048, 0.2), (768, 1024, 0.1), (1280, 320, 0.3)])
def test_layernorm_kernels_emit_the_packs_the_separate_passes_would(rows, d, p):
    """lstc_layernorm_fwd_pack / lstc_layernorm_bwd_drop_pack against lstc_layernorm_fwd / _bwd + lstc_dropout_apply +
    lstc_pack1: f32 results and the packed bf16 bytes are IDENTICAL (same registers, same rounding); the third partial plane
    sums to the column sums of the dropped gradient (order differs: 1e-5 relative)."""
    from lstc_vad_amd import functional as Fn, _lib
    from lstc_vad_amd.functional import dev_ptr, stream_ptr, check
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(rows, d, device=DEV, generator=g) * 2 + 0.3
    dz = torch.randn(rows, d, device=DEV, generator=g)
    gamma = torch.randn(d, device=DEV, generator=g)
    beta = torch.randn(d, device=DEV, generator=g)
    seed = 0x1234567890ABCDEF
    y0, mean0, rstd0 = Fn.layernorm_fwd(x, gamma, beta, 1e-6)
    nb = int(lib.lstc_pack1_bytes(rows, d))
    tiles = rows * d * 2                                    # bytes of the tile area (rows, d fill the grid exactly)
    y1, mean1, rstd1 = torch.empty_like(x), torch.empty_like(mean0), torch.empty_like(rstd0)
    pk = torch.zeros(nb, device=DEV, dtype=torch.uint8)
    check(lib.lstc_layernorm_fwd_pack(dev_ptr(x), dev_ptr(gamma), dev_ptr(beta), dev_ptr(y1), dev_ptr(mean1), dev_ptr(rstd1),
                                      rows, d, 1e-6, dev_ptr(pk), stream_ptr()), "fwd_pack")
    ref = torch.zeros(nb, device=DEV, dtype=torch.uint8)
    check(lib.lstc_pack1(dev_ptr(y0), rows, d, d, 0, dev_ptr(ref), stream_ptr()), "pack1")
    assert torch.equal(y0, y1) and torch.equal(mean0, mean1) and torch.equal(rstd0, rstd1)
    assert torch.equal(pk[:tiles], ref[:tiles])

    dx0, dg0, db0 = Fn.layernorm_bwd(dz, x, gamma, mean0, rstd0)
    df0 = Fn.dropout_apply(dx0, p, seed) if p > 0 else dx0
    check(lib.lstc_pack1(dev_ptr(df0), rows, d, d, 0, dev_ptr(ref), stream_ptr()), "pack1")
    n_partial = min(max(rows // 4, 1), 512)
    part = torch.empty(3, n_partial, d, device=DEV)
    dx1 = torch.empty_like(x)
    pk.zero_()
    check(lib.lstc_layernorm_bwd_drop_pack(dev_ptr(dz), dev_ptr(x), dev_ptr(gamma), dev_ptr(mean0), dev_ptr(rstd0), dev_ptr(dx1),
                                           dev_ptr(part), n_partial, rows, d, p, seed, dev_ptr(pk), stream_ptr()), "bwd_pack")
    assert torch.equal(dx0, dx1)
    assert torch.equal(pk[:tiles], ref[:tiles])
    assert max_abs_diff(part[0].double().sum(0), dg0.double()) <= 1e-5 * float(dg0.abs().max()) + 1e-6
    assert max_abs_diff(part[1].double().sum(0), db0.double()) <= 1e-5 * float(db0.abs().max()) + 1e-6
    bias = df0.double().sum(0)
    assert max_abs_diff(part[2].double().sum(0), bias) <= 1e-5 * float(bias.abs().max()) + 1e-6
    # shapes that do not fill the tile grid are refused (the caller packs separately), not half-written
    assert lib.lstc_layernorm_fwd_pack(dev_ptr(x), dev_ptr(gamma), dev_ptr(beta), dev_ptr(y1), dev_ptr(mean1), dev_ptr(rstd1),
                                       rows - 128, d, 1e-6, dev_ptr(pk), stream_ptr()) == -4     # LSTC_E_UNSUPPORTED


def test_bf16_encoder_step_is_bitwise_the_same_with_and_without_fused_packs():
    """bf16 mode, 256 sequences x 17 tokens x d=256 (4352 rows = 17 x 256): the encoder's forward and every parameter gradient
    with the LayerNorm kernels emitting the packs (default) against LSTC_NO_FUSED_PACKS=1 behaviour (separate lstc_pack1 /
    lstc_dropout_apply / lstc_colsum passes) - same dropout seeds, bit-identical outputs and weight gradients."""
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.models import Encoder
    torch.manual_seed(3)
    enc = Encoder(n_layers=3, n_head=4, d_k=64, d_v=64, d_model=256, d_inner=512, MHA_attn_dropout=0.1, MHA_fc_dropout=0.1,
                  FFN_dropout=0.1, MHA_layerNorm=True, FFN_layerNorm=True, relative_pe=False).to(DEV).train()
    x = torch.randn(256, 16, 256, device=DEV)
    outs = []
    Fn.set_compute_dtype("bf16"); Fn.set_x3_threshold(0, 0, 0)
    try:
        for fuse in (True, False):
            Fn._FUSE_PACKS = fuse
            torch.manual_seed(17); Fn.reset_rng()
            enc.zero_grad(set_to_none=True)
            y = enc(x)
            y.square().mean().backward()
            torch.cuda.synchronize()
            outs.append((y.detach().clone(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}))
            Fn.bump_weight_epoch()
    finally:
        Fn._FUSE_PACKS = True
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()
    (y1, g1), (y0, g0) = outs
    assert torch.equal(y1, y0)
    assert g1.keys() == g0.keys() and len(g1) >= 30
    for k in g1:
        if k.endswith("w_1.bias"):
            # fused: column sums of the hidden's gradient read from its packed bf16 form (lstc_colsum_pack1); separate passes:
            # of its f32 form - every addend differs by one bf16 rounding (2^-9 relative)
            assert max_abs_diff(g1[k], g0[k]) <= 1e-2 * float(g0[k].abs().max()) + 1e-9, k
        elif k.endswith("w_2.bias") or ".layer_norm." in k:
            # column sums (of df; of dz * xhat, dz): the fused kernel's per-workgroup partials cover other row sets than the
            # separate passes' - same addends, another summation order
            assert max_abs_diff(g1[k], g0[k]) <= 1e-5 * float(g0[k].abs().max()) + 1e-9, k
        else:
            assert torch.equal(g1[k], g0[k]), (k, int((g1[k] != g0[k]).sum()), [(kk, int((g1[kk] != g0[kk]).sum())) for kk in g1 if not torch.equal(g1[kk], g0[kk])])


@pytest.mark.parametrize("M,N,K", [(512, 256, 256), (1024, 768, 320)])
def test_packed_output_and_packed_relu_mask_of_the_bf16_gemm(M, N, K):
    """LSTC_EPI_OUT_PACK / LSTC_EPI_RELU_MASK_PACK: the product written as a packed bf16 operand equals lstc_pack1 of the f32
    result of the same launch without the flag, bit for bit (same accumulators, same epilogue, one RNE rounding); the ReLU mask
    read from a packed hidden equals the mask read from its f32 form; lstc_colsum_pack1 sums the bf16 values."""
    from lstc_vad_amd import functional as Fn, _lib
    from lstc_vad_amd.functional import dev_ptr, stream_ptr, check
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(21)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) * 0.1
    b = torch.randn(N, device=DEV, generator=g)
    dy = torch.randn(M, K, device=DEV, generator=g)            # a second product with N outputs: dy [M, K] @ w2 [K... reuse w as [N, K]
    Fn.set_compute_dtype("bf16"); Fn.set_x3_threshold(0, 0, 0)
    try:
        with Fn.pack_memo():
            h_f32 = Fn.gemm(x, w, trans_b=True, bias=b, relu=True)
            h_pk = Fn.gemm(x, w, trans_b=True, bias=b, relu=True, out_pack=True)
            ref = Fn.pack3(h_f32, False)
            tiles = M * N * 2
            assert torch.equal(h_pk.buf[:tiles], ref.buf[:tiles])
            d_f32 = Fn.gemm(dy, w, trans_b=True, relu_mask=h_f32)
            d_msk = Fn.gemm(dy, w, trans_b=True, relu_mask=h_pk)
            assert torch.equal(d_f32, d_msk)
            d_pk = Fn.gemm(dy, w, trans_b=True, relu_mask=h_pk, out_pack=True)
            assert torch.equal(d_pk.buf[:tiles], Fn.pack3(d_f32, False).buf[:tiles])
            # round 6: the 16-byte packed-output epilogue with an F32 per-element operand (two 16-B loads per group of 8 outputs) -
            # the f32 ReLU-mask source, and bias + dropout + an f32 residual - against the f32-output launch of the same flags
            d_pk2 = Fn.gemm(dy, w, trans_b=True, relu_mask=h_f32, out_pack=True)
            assert torch.equal(d_pk2.buf[:tiles], d_pk.buf[:tiles])
            res = torch.randn(M, N, device=DEV, generator=g)
            r_f32 = Fn.gemm(x, w, trans_b=True, bias=b, dropout=(0.2, 1234), residual=res)
            r_pk = Fn.gemm(x, w, trans_b=True, bias=b, dropout=(0.2, 1234), residual=res, out_pack=True)
            assert torch.equal(r_pk.buf[:tiles], Fn.pack3(r_f32, False).buf[:tiles])
            cs = Fn.colsum_pack(d_pk)
        torch.cuda.synchronize()
    finally:
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()
    want = _bf16_round(d_f32).double().sum(0)
    assert max_abs_diff(cs.double(), want) <= 1e-5 * float(want.abs().max()) + 1e-6
    # shapes off the 256-tile grid are refused, not half-written
    Fn.set_compute_dtype("bf16"); Fn.set_x3_threshold(0, 0, 0)
    try:
        with pytest.raises(RuntimeError):
            Fn.gemm(x[:300], w, trans_b=True, out_pack=True)
    finally:
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()


@pytest.mark.parametrize("rows,d,p", [(512, 2048, 0.2), (770, 1024, 0.1), (300, 512, 0.35)])
def test_layernorm_backward_with_fused_f32_dropout_replay(rows, d, p):
    """lstc_layernorm_bwd_drop (f32 modes) against lstc_layernorm_bwd + lstc_dropout_apply: dx and df identical bit for bit,
    the three partial planes sum to dgamma, dbeta and the column sums of df."""
    from lstc_vad_amd import functional as Fn, _lib
    from lstc_vad_amd.functional import dev_ptr, stream_ptr, check
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(12)
    x = torch.randn(rows, d, device=DEV, generator=g) * 2 + 0.3
    dz = torch.randn(rows, d, device=DEV, generator=g)
    gamma = torch.randn(d, device=DEV, generator=g)
    beta = torch.randn(d, device=DEV, generator=g)
    seed = 0x0FEDCBA987654321
    _, mean, rstd = Fn.layernorm_fwd(x, gamma, beta, 1e-6)
    dx0, dg0, db0 = Fn.layernorm_bwd(dz, x, gamma, mean, rstd)
    df0 = Fn.dropout_apply(dx0, p, seed)
    n_partial = min(max(rows // 4, 1), 768)
    part = torch.empty(3, n_partial, d, device=DEV)
    dx1, df1 = torch.empty_like(x), torch.empty_like(x)
    check(lib.lstc_layernorm_bwd_drop(dev_ptr(dz), dev_ptr(x), dev_ptr(gamma), dev_ptr(mean), dev_ptr(rstd), dev_ptr(dx1), dev_ptr(df1),
                                      dev_ptr(part), n_partial, rows, d, p, seed, stream_ptr()), "bwd_drop")
    assert torch.equal(dx0, dx1) and torch.equal(df0, df1)
    for plane, want in ((0, dg0.double()), (1, db0.double()), (2, df0.double().sum(0))):
        assert max_abs_diff(part[plane].double().sum(0), want) <= 1e-5 * float(want.abs().max()) + 1e-6
    assert lib.lstc_layernorm_bwd_drop(dev_ptr(dz), dev_ptr(x), dev_ptr(gamma), dev_ptr(mean), dev_ptr(rstd), dev_ptr(dx1), dev_ptr(df1),
                                       dev_ptr(part), n_partial, rows, 768, p, seed, stream_ptr()) == -4


@pytest.mark.parametrize("M,N,K,tb", [(12544, 2048, 256, True), (12544, 4096, 160, True), (8320, 2048, 192, False)])
def test_row_split_f32_product_is_bitwise_the_single_launch_product(M, N, K, tb):
    """Exact-f32 GEMM, tile-round quantisation remedy (csrc/gemm_f32.hip, lstc_gemm_f32_impl): when the 128x128 tile count ends
    with a mostly empty round of the 512 workgroup slots, the rows of that round are computed by the 64x64-tile variant.  Same k
    order per output element: the result - here with bias, ReLU, dropout (global row counter) and residual - equals the
    one-launch product (variant 4 = the default kernel without the split) bit for bit."""
    from lstc_vad_amd import functional as Fn
    g = torch.Generator(device=DEV).manual_seed(31)
    a = torch.randn(M, K, device=DEV, generator=g)
    b = torch.randn((N, K) if tb else (K, N), device=DEV, generator=g) * 0.1
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g)
    kw = dict(trans_b=tb, bias=bias, relu=True, dropout=(0.2, 0xABCDEF12345), residual=res)
    split = Fn.gemm(a, b, **kw)
    single = Fn.gemm(a, b, variant=4, **kw)
    assert torch.equal(split, single)
    ref = torch.relu(a.double() @ (b.double().T if tb else b.double()) + bias.double())
    keep = Fn.dropout_apply(torch.ones(M, N, device=DEV), 0.2, 0xABCDEF12345).double()
    assert max_abs_diff(split, ref * keep + res.double()) < 1e-4 * (K ** 0.5)


@pytest.mark.parametrize("M,N,K,ta,tb", [(12544, 2048, 2048, False, True), (12544, 2048, 2048, False, False), (6221, 2052, 256, False, True),
                                         (9000, 4096, 160, False, False), (2048, 2048, 12544, True, False), (70000, 2048, 128, False, True),
                                         (5000, 2050, 192, False, True)])
def test_float4_epilogue_and_persistent_walk_are_bitwise_the_scalar_epilogue_product(M, N, K, ta, tb):
    """Exact-f32 GEMM, round 3: (a) the default kernel's epilogue transposes each 4-register group across its lane quad so a
    lane stores one row-contiguous float4 (16 stores of 8 full lines per wave instead of 64 two-segment ones; operands as
    float4 loads); (b) variant 12 walks the tiles with persistent workgroups and requests the next tile's first K tile before
    that epilogue.  Same K loop, same k order, same epilogue arithmetic: both must equal variant 8 (PIPE 3, the scalar
    epilogue, one tile per workgroup) bit for bit - plain, with bias + ReLU + dropout + residual, with a ReLU mask and alpha,
    with a residual only, and accumulating; ragged M, N = 2052 (not a tile multiple), N = 2050 (not a multiple of 4: the scalar
    form must take over), NT / NN / TN, and outputs / residuals that are column slices of wider buffers."""
    from lstc_vad_amd import functional as Fn
    g = torch.Generator(device=DEV).manual_seed(3)
    a = torch.randn((K, M) if ta else (M, K), device=DEV, generator=g)
    b = torch.randn((N, K) if tb else (K, N), device=DEV, generator=g) * 0.1
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g)
    msk = torch.randn(M, N, device=DEV, generator=g)
    for kw in (dict(), dict(bias=bias, relu=True, dropout=(0.2, 0xABCDEF12345), residual=res), dict(relu_mask=msk, alpha=0.5), dict(residual=res)):
        want = Fn.gemm(a, b, trans_a=ta, trans_b=tb, variant=8, **kw)
        for v in (12, 4, 0):
            assert torch.equal(Fn.gemm(a, b, trans_a=ta, trans_b=tb, variant=v, **kw), want), (v, list(kw))
    acc0 = torch.randn(M, N, device=DEV, generator=g)
    want = Fn.gemm(a, b, trans_a=ta, trans_b=tb, variant=8, out=acc0.clone(), accumulate=True)
    for v in (12, 0):
        assert torch.equal(Fn.gemm(a, b, trans_a=ta, trans_b=tb, variant=v, out=acc0.clone(), accumulate=True), want), v
    ref = a.double().T @ b.double() if ta else a.double() @ (b.double().T if tb else b.double())
    assert max_abs_diff(Fn.gemm(a, b, trans_a=ta, trans_b=tb), ref) < 1e-4 * (K ** 0.5)
    if N % 8 == 0:
        # output and residual as column slices of wider buffers (the fused Q|K|V projection writes such slices): leading dimension
        # != N, 16-B aligned column offset -> still the float4 form; an offset of 2 columns breaks the alignment -> scalar form
        wide, rw = torch.zeros(M, 2 * N + 8, device=DEV), torch.randn(M, 2 * N + 8, device=DEV, generator=g)
        for off in (N // 2, 2):
            got = Fn.gemm(a, b, trans_a=ta, trans_b=tb, out=wide[:, off:off + N], residual=rw[:, off:off + N])
            want = Fn.gemm(a, b, trans_a=ta, trans_b=tb, variant=8, residual=rw[:, off:off + N].contiguous())
            assert torch.equal(got, want), off


@pytest.mark.parametrize("S", [49, 17, 81, 19])
def test_attention_backward_packed_gradients_equal_the_packed_f32_gradients(S):
    """lstc_attn_fwd with O_pack and lstc_attn_bwd with dQ_pack / dK_pack / dV_pack (bf16 mode): the packed bf16 results are
    lstc_pack1 of the f32 results the same kernels write without them, bit for bit (staged kernels; S = 81 is the 8-wave
    instantiation); probabilities and the bias-table gradient are unchanged."""
    from lstc_vad_amd import functional as Fn
    N, H, dk = 256, 4, 64
    L = {49: 3, 17: 1, 81: 5, 19: 2}[S]       # S = 19: UCF, the [32, 32] index read through its top-left [18, 18]
    g = torch.Generator(device=DEV).manual_seed(41)
    M = N * S
    q, k, v, do = (torch.randn(M, H * dk, device=DEV, generator=g) for _ in range(4))
    table = torch.randn((2 * L - 1) * 49, H, device=DEV, generator=g) * 0.3 if S != 17 else None
    index = None
    if table is not None:
        index = orc.relative_position_index_3d(L, 4).to(DEV)
    Fn.set_compute_dtype("bf16"); Fn.set_x3_threshold(0, 0, 0)
    try:
        o, probs = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, table, index, 0.1, 77)
        assert Fn.attn_fwd_pack(N, S, H, dk)
        op, probs_p = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, table, index, 0.1, 77, packed=True)      # forward: O as a pack only
        assert torch.equal(probs, probs_p) and torch.equal(Fn.pack3(o, False).buf[:M * H * dk * 2], op.buf[:M * H * dk * 2])
        dq, dk_, dv_, dt0 = Fn.attn_bwd(do, q, k, v, probs, N, S, H, dk, dk, table, index, 0.1, 77)
        assert Fn.attn_bwd_packs(N, S, H, dk, dk)
        pq, pk, pv, dt1 = Fn.attn_bwd(do, q, k, v, probs, N, S, H, dk, dk, table, index, 0.1, 77, packed=True)
        n = M * H * dk * 2
        for f32, pack in ((dq, pq), (dk_, pk), (dv_, pv)):
            assert torch.equal(Fn.pack3(f32, False).buf[:n], pack.buf[:n])
        if dt0 is not None:
            assert torch.equal(dt0, dt1)
        fused, _, _, dt2 = Fn.attn_bwd(do, q, k, v, probs, N, S, H, dk, dk, table, index, 0.1, 77, packed="fused")
        cat = torch.cat([dq, dk_, dv_], dim=1).contiguous()
        assert torch.equal(Fn.pack3(cat, False).buf[:3 * n], fused.buf[:3 * n])
        torch.cuda.synchronize()
    finally:
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()


def _unpack1(buf, rows, K):
    """f32 [rows, K] of an lstc_pack1 buffer (csrc/lstc_common.h p1_offset): 128-row x 32-k tiles, 64-B rows, 16-B chunk index
    XOR (row >> 2) & 3."""
    h = buf.view(torch.bfloat16)
    r = torch.arange(rows, device=buf.device).view(-1, 1)
    k = torch.arange(K, device=buf.device).view(1, -1)
    kbp = ((K + 63) // 64) * 2
    rr, kb, ch = r & 127, k >> 5, (k & 31) >> 3
    off = ((r >> 7) * kbp + kb) * 4096 + rr * 32 + ((ch ^ ((rr >> 2) & 3)) << 3) + (k & 7)
    return h[off.reshape(-1)].view(rows, K).float()


@pytest.mark.parametrize("S,L,dk,H", [(49, 3, 256, 8), (17, 1, 256, 8), (81, 5, 256, 8), (19, 2, 64, 4), (33, 2, 64, 4), (81, 5, 64, 4), (96, 6, 64, 4),
                                      (32, 2, 64, 4), (64, 4, 128, 2)])
def test_packed_input_attention_matches_the_f32_input_kernels_and_f64(S, L, dk, H):
    """LstcAttnDesc.in_pack_cols > 0 (bf16 mode): Q | K | V and dO read from lstc_pack1 buffers, O and dQ | dK | dV written as
    packs, probabilities with a padded row pitch.  On bf16-representable operands the results are those of the f32-input
    bf16-mode kernels up to the summation order of the softmax (probabilities to 1e-6, packed outputs to one bf16 rounding);
    S = 49 and 81 also against an f64 evaluation of models/MultiHeadAttention.py:103-122.  Workgroups walk several sequences
    (the staging ring runs across them); S = 33 repeats both launches with 3 sequences per workgroup, which leaves the last
    workgroup a partial chunk; S = 32, 64, 96: no padding rows in the last query tile (S = 32 and 64 run without dropout)."""
    from lstc_vad_amd import functional as Fn
    N = 256 if S != 81 else 512
    M = N * S
    g = torch.Generator(device=DEV).manual_seed(100 + S + dk)
    qkv = torch.randn(M, 3 * H * dk, device=DEV, generator=g).bfloat16().float()
    do = torch.randn(M, H * dk, device=DEV, generator=g).bfloat16().float()
    q, k, v = qkv[:, :H * dk], qkv[:, H * dk:2 * H * dk], qkv[:, 2 * H * dk:]
    index = orc.relative_position_index_3d(L, 4).to(DEV) if S != 17 else None
    table = torch.randn((2 * L - 1) * 49, H, device=DEV, generator=g) * 0.3 if S != 17 else None
    p_drop = 0.0 if S in (32, 64) else 0.2
    Fn.set_compute_dtype("bf16"); Fn.set_x3_threshold(0, 0, 0)
    try:
        assert Fn.attn_packed_inputs(N, S, H, dk, dk)
        qkv_p, do_p = Fn.pack3(qkv, False), Fn.pack3(do, False)
        o_ref, pr_ref = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, table, index, p_drop, 7, packed=True)
        o_new, pr_new = Fn.attn_fwd(qkv_p, None, None, N, S, H, dk, dk, table, index, p_drop, 7)
        assert pr_new.shape == pr_ref.shape and pr_new.stride(2) % 4 == 0
        # 1 / sqrt(d_k) a power of two: scaling commutes with the bf16 rounding of Q, the two generations differ by summation order
        # only.  d_k = 128: the f32-input kernels round Q * scale, the packed-input ones scale the f32 logits of the rounded Q -
        # two different bf16 roundings of the same product (2^-9 relative on a logit), both checked against f64 below
        pow2 = dk in (64, 256)
        loose = 1.0 if pow2 else 12.0
        assert max_abs_diff(pr_new, pr_ref) < (2e-6 if pow2 else 4e-3)
        assert float(pr_new.sum(-1).sub(1).abs().max()) < 1e-5
        if pr_new.stride(2) > S:        # padding columns: the forward's zeros
            assert float(torch.as_strided(pr_new, (N, H, S, pr_new.stride(2) - S), pr_new.stride(), S).abs().max()) == 0.0
        a, b = _unpack1(o_ref.buf, M, H * dk), _unpack1(o_new.buf, M, H * dk)
        # (a probability that rounds to the other bf16 neighbour moves an output by 2^-9 p |v|: a few 1e-3 absolute)
        assert float(((a - b).abs() - (2.0 ** -7) * a.abs()).max()) < loose * 2e-3 * float(a.abs().max())
        ref = Fn.attn_bwd(do, q, k, v, pr_ref, N, S, H, dk, dk, table, index, p_drop, 7, packed="fused")
        new = Fn.attn_bwd(do_p, qkv_p, None, None, pr_new, N, S, H, dk, dk, table, index, p_drop, 7)
        a, b = _unpack1(ref[0].buf, M, 3 * H * dk), _unpack1(new[0].buf, M, 3 * H * dk)
        assert not torch.isnan(b).any()
        assert float(((a - b).abs() - (2.0 ** -6) * a.abs()).max()) < loose * 4e-3 * float(a.abs().max())
        if table is not None:
            assert max_abs_diff(new[3], ref[3]) < (1e-5 if pow2 else 2e-2) * float(ref[3].abs().max())
            again = Fn.attn_bwd(do_p, qkv_p, None, None, pr_new, N, S, H, dk, dk, table, index, p_drop, 7)
            assert torch.equal(again[3], new[3]) and torch.equal(_unpack1(again[0].buf, M, 3 * H * dk), b)       # run-to-run bit-identical
        if S == 33:
            Fn._ATTN_VARIANT, Fn._BWD_NPW = 103, 3
            try:
                o3, pr3 = Fn.attn_fwd(qkv_p, None, None, N, S, H, dk, dk, table, index, p_drop, 7)
                new3 = Fn.attn_bwd(do_p, qkv_p, None, None, pr_new, N, S, H, dk, dk, table, index, p_drop, 7)
            finally:
                Fn._ATTN_VARIANT, Fn._BWD_NPW = 0, 0
            assert torch.equal(pr3, pr_new) and torch.equal(_unpack1(o3.buf, M, H * dk), _unpack1(o_new.buf, M, H * dk))
            assert torch.equal(_unpack1(new3[0].buf, M, 3 * H * dk), b)
            assert max_abs_diff(new3[3], new[3]) < 1e-5 * float(new[3].abs().max())
        if S in (49, 81, 64):
            # f64 evaluation with the SAME dropout mask (elements where the kernels kept the probability)
            n_chk = 8
            qd, kd, vd, dod = (t[: n_chk * S].double().view(n_chk, S, H, dk).transpose(1, 2) for t in (q, k, v, do))
            att = qd @ kd.transpose(-1, -2) / dk ** 0.5
            att[:, :, 1:, 1:] += table.double()[index[: S - 1, : S - 1].reshape(-1)].view(S - 1, S - 1, H).permute(2, 0, 1)
            pr = att.softmax(-1)
            assert max_abs_diff(pr_new[:n_chk], pr.float()) < 2e-3        # bf16 products of Q K^T
            keep = Fn.dropout_mask((N, H, S, S), p_drop, 7, DEV)[:n_chk].double() / (1 - p_drop) if p_drop > 0 else 1.0
            o64 = ((pr * keep) @ vd).transpose(1, 2).reshape(n_chk * S, H * dk)
            assert max_abs_diff(_unpack1(o_new.buf, M, H * dk)[: n_chk * S], o64.float()) < 3e-2
    finally:
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()


def test_bf16_encoder_step_with_packed_attention_operands_is_as_close_to_fp32_as_the_f32_operand_path():
    """bf16 mode, fused Q|K|V projection, 256 sequences x 49 tokens x d = 512 (H = 8, d_k = 64), same dropout seeds: the encoder
    step with Q | K | V / dO handed to the attention core as packed bf16 operands (default) and with LSTC_ATTN_PACKED_IN=0 (f32 Q,
    K, V, dO; the same bf16 products inside the kernels), both against the fp32-mode step.  The two bf16 paths round the same
    values at the same places, so they sit at the same distance from the fp32 step (within 10 % of each other, tensor by tensor -
    the attention gradients of a randomly initialised encoder are heavily cancelling sums, 6-20 % away from fp32 in either
    path) and their outputs agree to 1e-3; the packed path is the one that ran (attn_fwd saw a Packed operand)."""
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.models import Encoder
    torch.manual_seed(5)
    enc = Encoder(n_layers=2, n_head=8, d_k=64, d_v=64, d_model=512, d_inner=1024, MHA_attn_dropout=0.1, MHA_fc_dropout=0.1,
                  FFN_dropout=0.1, MHA_layerNorm=True, FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=3).to(DEV).train()
    for layer in enc.layer_stack:
        layer.slf_attn.fuse_qkv_()
    x = torch.randn(256, 48, 512, device=DEV)
    res, seen = {}, []
    real = Fn.attn_fwd
    def spy(q, *a, **kw):
        seen.append(isinstance(q, Fn.Packed))
        return real(q, *a, **kw)
    Fn.attn_fwd = spy
    try:
        for name, mode, packed_in in (("fp32", "fp32", True), ("packed", "bf16", True), ("f32ops", "bf16", False)):
            Fn.set_compute_dtype(mode)
            if mode == "bf16":
                Fn.set_x3_threshold(0, 0, 0)
            Fn._ATTN_PACKED_IN = packed_in
            torch.manual_seed(17); Fn.reset_rng()
            enc.zero_grad(set_to_none=True)
            y = enc(x)
            y.square().mean().backward()
            torch.cuda.synchronize()
            res[name] = (y.detach().clone(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None})
            Fn.bump_weight_epoch()
            Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()
    finally:
        Fn.attn_fwd = real
        Fn._ATTN_PACKED_IN = True
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()
    assert seen == [False, False, True, True, False, False]
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    assert rel(res["packed"][0], res["f32ops"][0]) < 2e-3 and rel(res["packed"][0], res["fp32"][0]) < 1e-2
    gp, gf, g32 = res["packed"][1], res["f32ops"][1], res["fp32"][1]
    assert gp.keys() == g32.keys() and len(gp) >= 20
    for k in g32:
        assert rel(gp[k], g32[k]) < 1.1 * rel(gf[k], g32[k]) + 2e-3, (k, rel(gp[k], g32[k]), rel(gf[k], g32[k]))


@pytest.mark.parametrize("learned,with_pos,split", [(False, False, True), (True, True, False), (False, True, True)])
def test_cls_concat_vector_kernel_is_bitwise_the_scalar_kernel(learned, with_pos, split):
    """lstc_cls_concat_fwd[_pack]: the four-columns-per-thread kernel (16-B aligned operands) against the one-column kernel
    (reached with an input that starts 4 bytes off a 16-B boundary) and against the reference's cat([cls, x], 1) [+ pos]
    (models/Encoder.py:52-61): bit-identical, mean or learned CLS token, one or two input tensors, with and without the packed
    bf16 copy."""
    from lstc_vad_amd import _lib
    from lstc_vad_amd.functional import dev_ptr, stream_ptr, check
    lib = _lib.load()
    N, S, d = 256, 17, 192
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(N, S - 1, d, device=DEV, generator=g)
    cls = torch.randn(d, device=DEV, generator=g) if learned else None
    pos = torch.randn(S, d, device=DEV, generator=g) if with_pos else None
    n_lo = N // 2 if split else N
    lo, hi = x[:n_lo].contiguous(), (x[n_lo:].contiguous() if split else None)
    want = torch.cat([(cls.expand(N, 1, d) if learned else x.sum(1, keepdim=True) / (S - 1)), x], 1)
    # the kernel sums the tokens one after another; torch.sum uses another order: compare row 0 with a tolerance, the rest exactly
    if with_pos:
        want = want + pos
    def run(xlo, xhi, packed):
        y = torch.empty(N, S, d, device=DEV)
        args = (dev_ptr(xlo), dev_ptr(xhi) if xhi is not None else None, n_lo, dev_ptr(cls) if learned else None,
                dev_ptr(pos) if with_pos else None, dev_ptr(y), N, S, d)
        if packed:
            buf = torch.zeros(int(lib.lstc_pack1_bytes(N * S, d)), device=DEV, dtype=torch.uint8)
            check(lib.lstc_cls_concat_fwd_pack(*args, dev_ptr(buf), stream_ptr()), "lstc_cls_concat_fwd_pack")
            return y, buf
        check(lib.lstc_cls_concat_fwd(*args, stream_ptr()), "lstc_cls_concat_fwd")
        return y, None
    off = torch.empty(lo.numel() + 1, device=DEV)[1:].view_as(lo)       # 4 bytes past a 16-B boundary: the scalar kernel
    off.copy_(lo)
    assert dev_ptr(off) % 16 == 4 and dev_ptr(lo) % 16 == 0
    y_vec, p_vec = run(lo, hi, True)
    y_sca, p_sca = run(off, hi, True)
    y_plain, _ = run(lo, hi, False)
    torch.cuda.synchronize()
    assert torch.equal(y_vec, y_sca) and torch.equal(y_vec, y_plain) and torch.equal(p_vec, p_sca)
    assert torch.equal(y_vec[:, 1:], want[:, 1:]) and max_abs_diff(y_vec[:, 0], want[:, 0]) < 1e-5
    assert torch.equal(_unpack1(p_vec, N * S, d), y_vec.view(N * S, d).bfloat16().float())


@pytest.mark.parametrize("S", [2, 3, 5, 16, 31, 32, 33, 48, 63, 64, 65, 80, 95, 96])
def test_packed_input_attention_sequence_length_sweep(S):
    """Every boundary of the packed-input kernels' tiling - S = 2 (one key besides CLS), the last row of a 32-query tile, the
    first row of the next one, S = 96 - against the f32-input bf16-mode kernels (first or second generation, whichever the
    shape takes): 256 sequences, H = 4, d_k = 64, relative bias read through the top-left (S-1) x (S-1) of a larger index,
    dropout 0.15."""
    from lstc_vad_amd import functional as Fn
    N, H, dk = 256, 4, 64
    M = N * S
    L = max(1, (S - 1 + 15) // 16)
    g = torch.Generator(device=DEV).manual_seed(1000 + S)
    qkv = torch.randn(M, 3 * H * dk, device=DEV, generator=g).bfloat16().float()
    do = torch.randn(M, H * dk, device=DEV, generator=g).bfloat16().float()
    q, k, v = qkv[:, :H * dk], qkv[:, H * dk:2 * H * dk], qkv[:, 2 * H * dk:]
    index = orc.relative_position_index_3d(L, 4).to(DEV)
    table = torch.randn((2 * L - 1) * 49, H, device=DEV, generator=g) * 0.3
    Fn.set_compute_dtype("bf16"); Fn.set_x3_threshold(0, 0, 0)
    try:
        assert Fn.attn_packed_inputs(N, S, H, dk, dk)
        qkv_p, do_p = Fn.pack3(qkv, False), Fn.pack3(do, False)
        o_ref, pr_ref = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, table, index, 0.15, 11, packed=True)
        o_new, pr_new = Fn.attn_fwd(qkv_p, None, None, N, S, H, dk, dk, table, index, 0.15, 11)
        assert max_abs_diff(pr_new, pr_ref) < 2e-6
        a, b = _unpack1(o_ref.buf, M, H * dk), _unpack1(o_new.buf, M, H * dk)
        assert float(((a - b).abs() - (2.0 ** -7) * a.abs()).max()) < 2e-3 * float(a.abs().max())
        ref = Fn.attn_bwd(do, q, k, v, pr_ref, N, S, H, dk, dk, table, index, 0.15, 11, packed="fused")
        new = Fn.attn_bwd(do_p, qkv_p, None, None, pr_new, N, S, H, dk, dk, table, index, 0.15, 11)
        a, b = _unpack1(ref[0].buf, M, 3 * H * dk), _unpack1(new[0].buf, M, 3 * H * dk)
        assert not torch.isnan(b).any()
        assert float(((a - b).abs() - (2.0 ** -6) * a.abs()).max()) < 4e-3 * float(a.abs().max())
        assert max_abs_diff(new[3], ref[3]) < 1e-5 * float(ref[3].abs().max()) + 1e-6
    finally:
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()


@pytest.mark.parametrize("S,dk,H", [(17, 256, 8), (81, 256, 8), (2, 64, 4), (19, 64, 4), (32, 64, 4), (33, 64, 4), (49, 128, 2), (64, 64, 4),
                                    (65, 64, 4), (96, 64, 4)])
def test_f32_lane_query_attention_forward_matches_the_first_generation(S, dk, H):
    """fp32 mode: attn_fwd3f_kernel (lane = query, exact-f32 MFMA, producer-wave ring; the default for S <= 32 and 64 < S <= 96,
    LstcAttnDesc.variant = 3 elsewhere) against the first-generation kernel (variant 1): probabilities and O to f32 summation
    order, with relative bias (read through the top-left of a larger index), dropout, partial last workgroup chunk (N = 300 is
    not a multiple of the sequences per workgroup)."""
    from lstc_vad_amd import functional as Fn
    N = 300
    M = N * S
    L = max(1, (S - 1 + 15) // 16)
    g = torch.Generator(device=DEV).manual_seed(2000 + S)
    q, k, v = (torch.randn(M, H * dk, device=DEV, generator=g) for _ in range(3))
    index = orc.relative_position_index_3d(L, 4).to(DEV)
    table = torch.randn((2 * L - 1) * 49, H, device=DEV, generator=g) * 0.3
    try:
        Fn._ATTN_VARIANT = 1
        o1, p1 = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, table, index, 0.2, 5)
        Fn._ATTN_VARIANT = 3
        o3, p3 = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, table, index, 0.2, 5)
        o3n, p3n = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, None, None, 0.0, 0)          # no bias, no dropout
        Fn._ATTN_VARIANT = 0
        o0, p0 = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, table, index, 0.2, 5)
        Fn._ATTN_VARIANT = 1
        o1n, p1n = Fn.attn_fwd(q, k, v, N, S, H, dk, dk, None, None, 0.0, 0)
    finally:
        Fn._ATTN_VARIANT = 0
    torch.cuda.synchronize()
    assert max_abs_diff(p3, p1) < 2e-6 and max_abs_diff(o3, o1) < 2e-5 * max(1.0, float(o1.abs().max()))
    assert max_abs_diff(p0, p1) < 2e-6 and max_abs_diff(o0, o1) < 2e-5 * max(1.0, float(o1.abs().max()))
    assert max_abs_diff(p3n, p1n) < 2e-6 and max_abs_diff(o3n, o1n) < 2e-5 * max(1.0, float(o1n.abs().max()))
    if S <= 32 or S > 64:
        assert torch.equal(o0, o3) and torch.equal(p0, p3)          # the default IS the lane-=-query kernel there


def test_weights_repacked_in_one_launch_after_the_optimizer_step_give_the_same_training_run():
    """bf16 mode: after Adagrad.step every packed weight copy the step used is rebuilt by ONE lstc_pack1_multi launch (leaf weights
    in both layouts and the fused Q|K|V view); the next step issues no lstc_pack1 for a weight.  Three steps with and without
    (LSTC_NO_REPACK behaviour): bit-identical weights; the multi launch equals per-item lstc_pack1 byte for byte."""
    from argparse import Namespace
    from lstc_vad_amd import functional as Fn, _lib
    from lstc_vad_amd.engine import TrainStep
    from lstc_vad_amd.models import Classifier, Encoder
    lib = _lib.load()
    # (a) the kernel: three operands, both layouts, ragged sizes
    g = torch.Generator(device=DEV).manual_seed(3)
    srcs = [torch.randn(300, 200, device=DEV, generator=g), torch.randn(512, 1024, device=DEV, generator=g), torch.randn(130, 70, device=DEV, generator=g)]
    items, singles, multis = [], [], []
    for t in srcs:
        for km in (0, 1):
            rows, K = (t.shape[1], t.shape[0]) if km else t.shape
            n = int(lib.lstc_pack1_bytes(rows, K))
            a, b = torch.zeros(n, device=DEV, dtype=torch.uint8), torch.zeros(n, device=DEV, dtype=torch.uint8)
            _lib.check(lib.lstc_pack1(_lib.dev_ptr(t), rows, K, t.stride(0), km, _lib.dev_ptr(a), _lib.stream_ptr()), "lstc_pack1")
            items.append(_lib.PackItem(t.data_ptr(), rows, K, t.stride(0), km, b.data_ptr()))
            singles.append(a); multis.append(b)
    _lib.check(lib.lstc_pack1_multi((_lib.PackItem * len(items))(*items), len(items), _lib.stream_ptr()), "lstc_pack1_multi")
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(singles, multis))
    # (b) the training loop
    def run(repack):
        torch.manual_seed(11)
        enc = Encoder(n_layers=3, n_head=4, d_k=64, d_v=64, d_model=256, d_inner=512, MHA_attn_dropout=0.1, MHA_fc_dropout=0.1,
                      FFN_dropout=0.1, MHA_layerNorm=True, FFN_layerNorm=True, relative_pe=True, window_size=4, window_depth=3).to(DEV).train()
        head = Classifier(256, 0.3).to(DEV).train()
        args = Namespace(batch_size=4, part_num=32, part_len=3, n_patch=16, lambda_1=0.01, lambda_MIL=1.0, lambda_CE=0.8, lambda_BCE=1.0,
                         lambda_normal=0.2, lambda_abnormal=2.0, temporal_only=False, clip_grad=False)
        Fn._REPACK_WEIGHTS = repack
        Fn.reset_rng()
        ts = TrainStep(args, "LTN", enc, head, 1e-4, 1e-2, 1e-3, fuse_qkv="on")
        gg = torch.Generator(device=DEV).manual_seed(1)
        nf = 0.5 * torch.relu(torch.randn(4, 96, 16, 256, device=DEV, generator=gg)); af = 0.5 * torch.relu(torch.randn(4, 96, 16, 256, device=DEV, generator=gg))
        al = torch.rand(4, 96, 1, device=DEV, generator=gg)
        calls = []
        real = lib.lstc_pack1
        for step in range(3):
            n0 = len(_pack_log)
            ts.step(nf, af, al)
            calls.append(len(_pack_log) - n0)
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in enc.state_dict().items()}, calls
    _pack_log = []
    real_pack3 = Fn.pack3
    def spy(t, k_major=False, kind=None):
        _pack_log.append((tuple(t.shape), k_major))
        return real_pack3(t, k_major, kind)
    Fn.set_compute_dtype("bf16"); Fn.set_x3_threshold(0, 0, 0)
    Fn.pack3 = spy
    try:
        w1, c1 = run(True)
        w0, c0 = run(False)
    finally:
        Fn.pack3 = real_pack3
        Fn._REPACK_WEIGHTS = True
        Fn.set_compute_dtype("fp32"); Fn.set_x3_threshold()
    assert c1[0] == c0[0] and c1[1] < c0[1] and c1[2] < c0[2], (c1, c0)      # first step packs lazily either way
    assert c0[1] - c1[1] >= 10, (c1, c0)                                      # every weight pack of steps 2, 3 came from the one launch
    assert w1.keys() == w0.keys()
    for k in w1:
        assert torch.equal(w1[k], w0[k]), k
