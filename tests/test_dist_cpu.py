"""world_size-2 gloo tests (CPU) of the data-parallel machinery: bucketed gradient all-reduce with autograd hooks
(lstc_vad_amd.dist) and the sharded-loss bookkeeping, using the oracle as the model (the HIP kernels need a GPU;
the distributed logic does not)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from lstc_vad_amd.dist import GradAllReducer
    from oracle import lstc_oracle as orc
    from util import load_case, sub, oracle_cfgs
    z, mode, ekw, skw = load_case("ltn_sht")
    ecfg, st = oracle_cfgs(mode, ekw, skw)
    bs = skw["batch_size"]
    nf, af, al = (torch.from_numpy(z[k]) for k in ("norm_feats", "abnorm_feats", "abnorm_labs"))
    enc_P = {k: (torch.nn.Parameter(v.clone()) if v.is_floating_point() else v) for k, v in sub(z, "enc_init.").items()}
    head_P = {k: torch.nn.Parameter(v.clone()) for k, v in sub(z, "head_init.").items()}
    used = [k for k in sub(z, "enc_grad.")]                      # parameters that receive a gradient
    buckets = [list(head_P.values())] + [[enc_P[k] for k in used if k.startswith(f"layer_stack.{i}.")] for i in (2, 1, 0)]
    red = GradAllReducer(buckets)
    red.zero_grad()
    # this rank's shard: pairs [rank*bs/world, (rank+1)*bs/world)
    h = bs // world
    sl = slice(rank * h, (rank + 1) * h)
    stl = orc.StepCfg(**{**st.__dict__, "batch_size": h})
    # global-count loss on the local shard: forward locally, exchange bag maxima, form the rank's contribution
    pn, L, P, d = st.part_num, st.part_len, st.n_patch, ecfg.d_model
    x = torch.cat([nf[sl].reshape(h * pn, L * P, d), af[sl].reshape(h * pn, L * P, d)], 0)
    enc = orc.encoder_forward(enc_P, x, ecfg, True)
    out = orc.head_forward(head_P, enc[:, 0, :], "classifier", 0.0, True)
    score = out[:, 1]
    bag_l = score.reshape(2 * h, pn).max(dim=-1)[0]
    bag = torch.zeros(2 * bs)
    bag[rank * h:(rank + 1) * h] = bag_l[:h].detach()
    bag[bs + rank * h: bs + (rank + 1) * h] = bag_l[h:].detach()
    dist.all_reduce(bag)                                          # the 2*bs-float exchange of SURVEY 8e
    nor_g, abn_g = bag[:bs], bag[bs:]
    # hinge with gradients flowing only through this rank's entries; pairs booked on the rank owning the normal video
    nor_l, abn_l = bag_l[:h], bag_l[h:]
    err = torch.relu(1 - abn_g[None, :] + nor_l[:, None]).sum() / bs ** 2
    err_abn_grad_only = torch.relu(1 - abn_l[None, :] + nor_g[:, None]).sum() / bs ** 2
    err_for_grad = err + (err_abn_grad_only - err_abn_grad_only.detach())
    gidx_n = torch.arange(h * pn) + rank * h * pn
    gidx_a = bs * pn + rank * h * pn + torch.arange(h * pn)
    gidx = torch.cat([gidx_n, gidx_a])
    l1 = (score * (gidx >= bs).float()).sum() / (2 * bs * pn - bs)
    labs = orc.soft_targets(al[sl], h, pn, L).reshape(2 * h * pn, 2)
    ce = -(labs * torch.log_softmax(out, -1)).sum() / (2 * bs * pn)
    loss = st.lambda_MIL * (err_for_grad + st.lambda_1 * l1) + st.lambda_CE * ce
    loss.backward()
    red.finish()
    tot = torch.tensor([float(loss.detach())])
    dist.all_reduce(tot)
    if rank == 0:
        ref_g = sub(z, "enc_grad.")
        worst = 0.0
        for k, g in ref_g.items():
            worst = max(worst, float((enc_P[k].grad - g).abs().max()) / max(1e-6, float(g.abs().max())))
        for k, g in sub(z, "head_grad.").items():
            worst = max(worst, float((head_P[k].grad - g).abs().max()) / max(1e-6, float(g.abs().max())))
        q.put((float(tot), float(z["scalars"][0]), worst, red.payload_bytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_step_equals_single_process_reference():
    """Two gloo ranks, each with half of the golden batch: summed loss == reference loss, all-reduced gradients
    == the reference's single-process gradients (golden vectors)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    tot, ref_loss, worst, payload = res
    assert abs(tot - ref_loss) < 2e-6
    assert worst < 2e-4
    assert payload > 0


def test_bucket_order_and_unused_parameters():
    from lstc_vad_amd.dist import encoder_head_buckets
    from lstc_vad_amd.models import Encoder, Classifier
    enc = Encoder(3, 2, 4, 4, 8, 16, MHA_layerNorm=False, FFN_layerNorm=True, relative_pe=True, window_depth=2)
    head = Classifier(8)
    b = encoder_head_buckets(enc, head)
    assert len(b) == 4 and len(b[0]) == 6
    ids = {id(p) for g in b for p in g}
    assert ids == {id(p) for p in enc.used_parameters()} | {id(p) for p in head.parameters()}
    assert id(enc.layer_norm.weight) not in ids and id(enc.layer_stack[0].slf_attn.layer_norm.weight) not in ids
    assert b[1][0] is next(iter(enc.layer_stack[2].parameters()))      # last layer first (backward order)
    # bench.py --buckets N: the finest backward-ordered pieces (head, per layer FFN then attention, rest) merged into N
    # consecutive groups - every used parameter exactly once, backward order kept, 1 <= groups <= pieces
    flat_default = [id(p) for g in b for p in g]
    for n in (1, 2, 3, 5, 7, 50):
        g_n = encoder_head_buckets(enc, head, n)
        ids_n = [id(p) for g in g_n for p in g]
        assert len(g_n) == min(n, 7) and sorted(ids_n) == sorted(flat_default) and len(set(ids_n)) == len(ids_n)
        assert ids_n[:6] == flat_default[:6]                           # head first
    from lstc_vad_amd.dist import direct_grad_parameters
    dp = direct_grad_parameters(enc, head)
    assert len(dp) == 3 * 6 + 1 and all(p.dim() == 2 for p in dp) and {id(p) for p in dp} <= ids


def _mixed_worker(rank, world, port, q):
    """engine.MixedStep on gloo: two model pairs of different widths, both backward passes issued before any wait; the
    reduced gradients must equal the single-process gradients over the union of the shards."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from types import SimpleNamespace
    from lstc_vad_amd.dist import GradAllReducer
    from lstc_vad_amd.engine import MixedStep

    class SGD:                                   # plain in-place update: keeps the arithmetic easy to restate
        def __init__(self, params): self.params = list(params)
        def zero_grad(self, set_to_none=True): pass
        def step(self):
            with torch.no_grad():
                for p in self.params:
                    p -= 0.1 * p.grad

    class TinyStep:                              # the attributes MixedStep touches on a TrainStep
        def __init__(self, d, seed):
            g = torch.Generator().manual_seed(seed)
            self.encoder = torch.nn.Linear(d, 5)
            self.head = torch.nn.Linear(5, 1)
            with torch.no_grad():
                for p in list(self.encoder.parameters()) + list(self.head.parameters()):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.3)
            self.args = SimpleNamespace(clip_grad=False)
            self.reducer = GradAllReducer([list(self.head.parameters()), list(self.encoder.parameters())])
            self.optimizer = SGD(list(self.encoder.parameters()) + list(self.head.parameters()))

        def forward_loss(self, nf, af, al):
            y = self.head(torch.tanh(self.encoder(torch.cat([nf, af], 0))))
            loss = (y ** 2).sum() + (y[nf.shape[0]:] * al).sum()
            return loss, loss.detach(), y

    def data(d, seed):
        g = torch.Generator().manual_seed(seed)
        return torch.randn(4, d, generator=g), torch.randn(4, d, generator=g), torch.rand(4, 1, generator=g)

    steps = [TinyStep(3, 1), TinyStep(7, 2)]
    full = [data(3, 11), data(7, 12)]
    h = 4 // world
    shard = [tuple(t[rank * h:(rank + 1) * h] for t in b) for b in full]
    MixedStep(steps).step(shard)
    # single-process restatement on the full batches
    worst = 0.0
    for (d, seed), b, ts in zip(((3, 1), (7, 2)), full, steps):
        ref = TinyStep.__new__(TinyStep)
        g = torch.Generator().manual_seed(seed)
        ref.encoder, ref.head = torch.nn.Linear(d, 5), torch.nn.Linear(5, 1)
        with torch.no_grad():
            for p in list(ref.encoder.parameters()) + list(ref.head.parameters()):
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
        loss, _, _ = TinyStep.forward_loss(ref, *b)
        loss.backward()
        for p, r in zip(list(ts.encoder.parameters()) + list(ts.head.parameters()),
                        list(ref.encoder.parameters()) + list(ref.head.parameters())):
            worst = max(worst, float((p.detach() - (r.detach() - 0.1 * r.grad)).abs().max()))
    if rank == 0:
        q.put(worst)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_mixed_step_two_models_one_reduction_stream():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mixed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    worst = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert worst < 1e-6


def _direct_worker(rank, world, port, q):
    """Gradient sinks on gloo: a Function that honours functional.grad_sink / deliver (what the HIP Functions do) writes its weight
    gradient INTO the all-reduce bucket and returns None to autograd; biases travel through autograd's accumulate.  The bucket
    starts every step poisoned with NaN in its direct region (nothing fills it: a gradient that is not written shows up)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.dist import GradAllReducer

    class Lin(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w)
            return x @ w.t() + b

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            out = Fn.grad_sink(w)
            dw = torch.mm(dy.t(), x, out=out) if out is not None else dy.t() @ x
            return dy @ w, Fn.deliver(w, dw), dy.sum(0)

    def make(seed):
        g = torch.Generator().manual_seed(seed)
        ws = [torch.nn.Parameter(torch.randn(6, 5, generator=g)), torch.nn.Parameter(torch.randn(4, 6, generator=g))]
        bs = [torch.nn.Parameter(torch.randn(6, generator=g)), torch.nn.Parameter(torch.randn(4, generator=g))]
        return ws, bs

    def loss_of(ws, bs, x):
        h = torch.tanh(Lin.apply(x, ws[0], bs[0]))
        return (Lin.apply(h, ws[1], bs[1]) ** 2).sum()

    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(8, 5, generator=g) for _ in range(2)]
    ws, bs = make(1)
    red = GradAllReducer([[bs[1], ws[1]], [bs[0], ws[0]]], direct=ws)           # weights are laid out FIRST in their bucket
    assert red.buckets[0].numel() == 28 and red._accum_from == [24, 30]
    assert ws[1].grad.data_ptr() == red.buckets[0].data_ptr() and Fn.grad_sink(ws[1]) is not None
    rws, rbs = make(1)                                                           # single-process reference on the full batch
    worst, launched_early = 0.0, True
    for step in range(2):
        red.zero_grad()
        for bi, flat in enumerate(red.buckets):
            flat[:red._accum_from[bi]] = float("nan")                            # the direct region is never filled by the reducer
        h = 8 // world
        loss_of(ws, bs, xs[step][rank * h:(rank + 1) * h]).backward()
        launched_early &= len(red._handles) == 2                                 # both buckets went out from inside backward
        red.finish()
        for p in rws + rbs:
            p.grad = None
        loss_of(rws, rbs, xs[step]).backward()
        for p, r in zip(ws + bs, rws + rbs):
            assert p.grad is not None and torch.isfinite(p.grad).all()
            worst = max(worst, float((p.grad - r.grad).abs().max()))
        with torch.no_grad():
            for p, r in zip(ws + bs, rws + rbs):
                p -= 0.05 * p.grad
                r -= 0.05 * r.grad
    # a model reused WITHOUT its reducer: once somebody else owns .grad the sink is ignored and autograd delivers the gradient
    for p in ws + bs:
        p.grad = None
    assert Fn.grad_sink(ws[0]) is None
    loss_of(ws, bs, xs[0]).backward()
    assert all(p.grad is not None and p.grad.data_ptr() != red.buckets[1].data_ptr() for p in ws)
    if rank == 0:
        q.put((worst, launched_early))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_direct_gradient_sinks_write_into_the_bucket_without_fill_or_accumulate():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_direct_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    worst, launched_early = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert worst < 1e-5 and launched_early


def test_direct_gradient_sinks_refuse_double_and_missing_deliveries_in_every_mode():
    """a sink is written, not accumulated into.  On ONE rank without a process group (the inactive bucket path) and with
    overlap off, the reducer used to skip its per-parameter bookkeeping: a weight used twice in one backward, or two backward
    passes before finish(), silently kept only the last weight gradient while the biases accumulated - and a weight whose Function
    did not run kept last step's gradient (zero_grad() does not clear the direct region).  Both now raise."""
    sys.path.insert(0, ROOT)
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.dist import GradAllReducer

    class Lin(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w)
            return x @ w.t() + b

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            out = Fn.grad_sink(w)
            dw = torch.mm(dy.t(), x, out=out) if out is not None else dy.t() @ x
            return dy @ w, Fn.deliver(w, dw), dy.sum(0)

    g = torch.Generator().manual_seed(3)
    w = torch.nn.Parameter(torch.randn(4, 4, generator=g))
    w2 = torch.nn.Parameter(torch.randn(4, 4, generator=g))
    b = torch.nn.Parameter(torch.randn(4, generator=g))
    x = torch.randn(3, 4, generator=g)
    for overlap in (True, False):
        red = GradAllReducer([[b, w, w2]], overlap=overlap, direct=[w, w2])
        assert not red.active
        # every direct weight delivers once: fine
        red.zero_grad()
        Lin.apply(Lin.apply(x, w, b), w2, b).sum().backward()
        red.finish()
        want = w.grad.clone()
        # a second backward before zero_grad(): refused
        with pytest.raises(RuntimeError, match="twice"):
            Lin.apply(Lin.apply(x, w, b), w2, b).sum().backward()
        # one weight shared by two Functions in one backward: refused
        red.zero_grad()
        with pytest.raises(RuntimeError, match="twice"):
            Lin.apply(Lin.apply(x, w, b), w, b).sum().backward()
        # a direct weight whose Function did not run this step: its slot would hold last step's values
        red.zero_grad()
        Lin.apply(x, w, b).sum().backward()
        with pytest.raises(RuntimeError, match="not produced"):
            red.finish()
        assert want is not None
        for p in (w, w2, b):
            p.grad = None
            p.__dict__.pop("_lstc_grad_sink", None)
    # the layout keeps direct slots 16-byte aligned
    odd = torch.nn.Parameter(torch.randn(3, generator=g))
    with pytest.raises(RuntimeError, match="16-byte"):
        GradAllReducer([[odd, w]], direct=[odd, w])
