"""One training step of the LSTC_VAD loops as a reusable object (used by Train/*.py, bench.py, smoke()).

Mirrors the body of the reference's inner loops — Train/temporal_transformer_shanghaitech.py:103-143 (LTN),
Train/spatio_transformer_shanghaitech.py:90-113 (STN), Train/spatio_transformer_MIL_CE.py:156-213 (STN + BCE) —
with the DataLoader batch contract of SURVEY.md 8a row A0:
``norm_feats, abnorm_feats`` ``[bs, part_num*part_len, n_patch, d_model]``, ``abnorm_labs`` ``[bs, part_num*part_len(,1)]``.
Under data parallelism each rank calls it with its own ``bs`` pairs.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .dist import GradAllReducer, direct_grad_parameters, encoder_head_buckets
from .losses import training_loss
from .optim import Adagrad, clip_grad_norm_


class TrainStep:
    def __init__(self, args, mode: str, encoder, head, lr_encoder: float, lr_head: float, weight_decay: float,
                 group=None, cls_only: bool = True, loss_rank=None, loss_exchange=None, fuse_qkv="auto",
                 grad_reduce_dtype: str = "fp32", n_buckets=None):
        # cls_only=False evaluates the last encoder layer for every token like the reference does (its extra rows are
        # never read); kept for A/B measurements — results are identical (tests/test_hip_parity.py)
        self.cls_only = cls_only
        # (rank, world) + bag exchange override for the loss: lets a test run the shards of several ranks on one device
        # through this very object (tests/test_hip_parity.py::test_two_emulated_ranks_through_trainstep)
        self.loss_rank, self.loss_exchange = loss_rank, loss_exchange
        self.args, self.mode, self.encoder, self.head, self.group = args, mode, encoder, head, group
        # MultiHeadAttention.fuse_qkv_() (one projection / dW / dX GEMM per layer instead of three).  At the full headline batch
        # the 128x128-tile GEMMs gain nothing from being wider (304.0 vs 301.3 ms per LTN step), but a rank of a strong-scaled
        # job holds few tokens: at 8 GPUs 12 544 tokens = 1568 tiles per projection = 3.06 rounds of the chip's 512 workgroup
        # slots, i.e. 4 rounds at 77 % occupancy, where the fused 4704-tile product runs 9.2 -> 10 rounds at 92 %.  "auto"
        # fuses when that tile-round arithmetic gains more than 3 %.
        if fuse_qkv == "auto":
            fuse_qkv = self._qkv_fusion_pays(args, mode, encoder)
        if fuse_qkv is True or fuse_qkv == "on":
            for layer in list(encoder.layer_stack)[:-1] if cls_only else list(encoder.layer_stack):
                layer.slf_attn.fuse_qkv_()
        self.fused_qkv = bool(fuse_qkv is True or fuse_qkv == "on")
        self.optimizer = Adagrad([{"params": encoder.parameters(), "lr": lr_encoder},
                                  {"params": head.parameters(), "lr": lr_head}], weight_decay=weight_decay)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        force = os.environ.get("LSTC_FORCE_DIST", "0") == "1" and dist.is_available() and dist.is_initialized()
        # measurement hook (bench.py at N > 1): a list to receive (before backward, after backward, after reducer.finish()) HIP events
        # of every step on the launch stream - backward time and the part of the gradient all-reduce that the backward did not hide
        self.comm_events = None
        # LSTC_BUCKET_STEPS=1: per-bucket optimizer steps on a side stream as each bucket's reduction lands (weights
        # bit-identical to the one-step order).  OFF by default: on the one box this build can measure - the RCCL bucket path forced onto
        # one rank, where a reduction lands at once and hides nothing - the 7 + 7 launches beside the backward cost more than the one
        # launch behind it saves (tools/r06_bucket_steps_ab.sh, profiles/r06_bucket_steps_ab.txt: 4 pairs per rank bf16 7.25 -> 7.31 ms,
        # fp32 36.95 -> 37.15; full batch bf16 39.1 -> 39.4).  What it is FOR - stepping six of seven buckets under the exposed wait for the
        # last reduction over xGMI - needs N > 1 devices to show.
        self.bucket_steps = os.environ.get("LSTC_BUCKET_STEPS", "0") == "1"
        # default bucketing: head, then per layer (last to first) its FFN half and its attention half - 2 x layers + 1 buckets of
        # <= 67 MB at the LTN widths.  Only the LAST bucket's reduction (layer 0's attention weights, ready when the backward ends)
        # cannot hide under the backward; halving it halves what is exposed, and 67 MB is still far above the size at which
        # RCCL's rings reach their bandwidth (bench.py --buckets N merges the pieces into N groups for comparison runs)
        self.reducer = (GradAllReducer(encoder_head_buckets(encoder, head, "finest" if n_buckets is None else n_buckets), group,
                                       force=force, reduce_dtype=grad_reduce_dtype,
                                       direct=direct_grad_parameters(encoder, head))
                        if (self.world > 1 or force) else None)

    @staticmethod
    def _qkv_fusion_pays(args, mode, encoder) -> bool:
        from . import functional as Fn
        try:
            attn = encoder.layer_stack[0].slf_attn
            tokens = args.part_len * args.n_patch if mode == "LTN" else args.n_patch
            n_seq = 2 * args.batch_size * args.part_num * (1 if mode == "LTN" else args.part_len)
            M, Hd = n_seq * (1 + tokens), attn.n_head * attn.d_k
        except Exception:
            return False
        if (Fn.get_compute_dtype() == "bf16" and Fn.attn_packed_inputs(n_seq, 1 + tokens, attn.n_head, attn.d_k, attn.d_v) and
                Fn.packed_out_shape(M, attn.n_head * (2 * attn.d_k + attn.d_v))):
            return True        # bf16 mode: the fused projection writes Q | K | V as ONE packed operand that the attention core reads
        tile, slots = (256, 256) if Fn.get_compute_dtype() == "bf16" else (128, 512)
        tiles = -(-M // tile) * -(-Hd // tile)
        eff = lambda t: (t / slots) / -(-t // slots)
        return eff(3 * tiles) > eff(tiles) + 0.03

    def sequences(self, norm_feats, abnorm_feats):
        """A1: [bs, pn*L, P, d] x2 -> [N, S-1, d], normal sequences first (the loss relies on this order)."""
        a = self.args
        d = norm_feats.shape[-1]
        tokens = a.part_len * a.n_patch if self.mode == "LTN" else a.n_patch
        return torch.cat([norm_feats.float().reshape(-1, tokens, d), abnorm_feats.float().reshape(-1, tokens, d)], 0)

    def forward_loss(self, norm_feats, abnorm_feats, abnorm_labs):
        a, d = self.args, norm_feats.shape[-1]
        tokens = a.part_len * a.n_patch if self.mode == "LTN" else a.n_patch
        from .feed import LazyRows
        if isinstance(norm_feats, LazyRows) or isinstance(abnorm_feats, LazyRows):
            # an HBM-resident feed handed over clip INDICES (feed.LazyRows): batch formation, cat and CLS concat are one pass over the
            # bank (lstc_cls_concat_gather_fwd) - where the encoder takes its input that way; otherwise gather now
            if (self.cls_only and isinstance(norm_feats, LazyRows) and norm_feats.pairs_with(abnorm_feats) and
                    not self.encoder.input_layerNorm and norm_feats.bank.dim() == 3 and norm_feats.bank.shape[1] == a.n_patch):
                Lc = a.part_len if self.mode == "LTN" else 1
                n_seq = 2 * norm_feats.bs * norm_feats.rows // Lc
                cls = self.encoder.forward_cls((norm_feats.bank, norm_feats.idx_flat, n_seq, Lc))
                outputs = self.head(cls)
                loss, scalars = training_loss(self.args, self.mode, outputs, abnorm_labs, group=self.group,
                                              distributed=self.loss_rank, exchange=self.loss_exchange)
                return loss, scalars, outputs
            norm_feats = norm_feats.materialize() if isinstance(norm_feats, LazyRows) else norm_feats
            abnorm_feats = abnorm_feats.materialize() if isinstance(abnorm_feats, LazyRows) else abnorm_feats
        # normal sequences first, abnormal second (A1); the cat itself is fused into the CLS-concat kernel
        if self.cls_only:
            cls = self.encoder.forward_cls(norm_feats.float().reshape(-1, tokens, d), abnorm_feats.float().reshape(-1, tokens, d))
        else:
            cls = self.encoder(self.sequences(norm_feats, abnorm_feats))[:, 0, :]
        outputs = self.head(cls)
        loss, scalars = training_loss(self.args, self.mode, outputs, abnorm_labs, group=self.group,
                                      distributed=self.loss_rank, exchange=self.loss_exchange)
        return loss, scalars, outputs

    def step(self, norm_feats, abnorm_feats, abnorm_labs):
        """forward, loss, backward, [gradient all-reduce], [clip], Adagrad.  Returns the 5 scalars
        (loss, MIL, err, l1, aux) as a device tensor — no host sync happens here."""
        loss, scalars, _ = self.forward_loss(norm_feats, abnorm_feats, abnorm_labs)
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)
        ev = self.comm_events
        if ev is not None:
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            marks[0].record()
        loss.backward()
        if ev is not None:
            marks[1].record()
        if self.reducer is not None and self.bucket_steps and not getattr(self.args, "clip_grad", False):
            # per-bucket optimizer: Adagrad + the bf16 weight repack of bucket k run on a side stream as soon as
            # bucket k's reduction has landed, beside the backward of the layers below - not behind the LAST bucket's reduction.
            # (--clip_grad needs the norm of ALL gradients first: it keeps the one-step order below.)
            self.reducer.finish(on_bucket=lambda bi, params: self.optimizer.step(only=params))
            if ev is not None:
                # exposed communication keeps its meaning: end of the backward -> the LAST bucket's reduction has landed
                marks[2] = self.reducer.landed if self.reducer.landed is not None else marks[1]
                ev.append(marks)
            return scalars
        if self.reducer is not None:
            self.reducer.finish()
        if ev is not None:
            marks[2].record()                  # the launch stream reaches this only when every bucket's reduction has landed
            ev.append(marks)
        if getattr(self.args, "clip_grad", False):
            clip_grad_norm_(self.encoder.parameters(), 10)
            clip_grad_norm_(self.head.parameters(), 10)
        self.optimizer.step()
        return scalars


class GraphedStep:
    """``TrainStep.step`` captured ONCE into a HIP graph and replayed per step: forward, loss, backward and the Adagrad
    launch - about 340 kernel launches in the LTN step - become one graph launch.  At the headline batch the GPU is never
    idle between launches, so this buys nothing there; it matters for the small per-rank batches of a strong-scaled job,
    where the launch train itself (2.7 of 8.5 ms at 4 pairs per rank in bf16 mode) is what is left of the step.

    What makes the capture replayable:
      * dropout: every kernel takes its mask seed by value, so a replay would repeat one mask.  The capture is bracketed by
        ``lstc_dropout_seed_device`` (include/lstc_hip.h): each launch then uses  seed + *word  read on the device, and the
        graph's last node adds the number of seeds a step draws to the word - replay k sees exactly the masks the k-th eager
        step would have drawn (``functional.next_seed`` is advanced alongside), so the weights are bit-identical to eager
        steps (tests/test_hip_parity.py::test_graphed_step_is_bitwise_the_eager_step);
      * inputs live in static buffers (``step`` copies the batch in); gradients, activations and split-K partials come out
        of the graph's private memory pool; the optimizer's one launch carries the parameter pointers, which never move;
      * no host sync and no host-dependent control flow inside a step (the five scalars stay on the device).
      * ``--clip_grad`` (Train/temporal_transformer_shanghaitech.py:139-141) is two multi-tensor launches per parameter list
        with the clip coefficient formed on the device (optim.clip_grad_norm_): captured like everything else.
    Not captured: the gradient all-reduce / bag exchange of a multi-rank job - ``GraphedStep`` refuses that and the caller
    keeps ``TrainStep.step``.

    While the capture is open ``lstc_dropout_seed_device`` points EVERY dropout-drawing launch of the process at the seed word:
    no other thread or stream may launch dropout kernels during construction (evaluation / feeder threads must be idle)."""

    def __init__(self, ts: TrainStep, norm_feats, abnorm_feats, abnorm_labs, warmup: int = 1):
        from . import _lib
        from . import functional as Fn
        if ts.reducer is not None or ts.world > 1 or ts.loss_exchange is not None:
            raise RuntimeError("GraphedStep: the multi-rank step (gradient all-reduce, bag exchange) is not captured")
        self.ts = ts
        dev = norm_feats.device
        self.nf, self.af, self.al = (torch.empty_like(t) for t in (norm_feats, abnorm_feats, abnorm_labs))
        for dst, src in ((self.nf, norm_feats), (self.af, abnorm_feats), (self.al, abnorm_labs)):
            dst.copy_(src)
        self.seed_word = torch.zeros(1, device=dev, dtype=torch.int64)
        params = [p for g in ts.optimizer.param_groups for p in g["params"]]
        # 1. warm-up on a side stream (lazy kernel attributes, autograd threads, allocator pools), then put weights, Adagrad
        #    state and the seed counter back: constructing a GraphedStep leaves the training state untouched
        keep_w = [p.detach().clone() for p in params]
        keep_s = [ts.optimizer.state[p]["sum"].clone() for p in params]
        keep_n = [ts.optimizer.state[p]["step"] for p in params]
        c0 = Fn._counter
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                Fn.reset_rng(c0)
                ts.step(self.nf, self.af, self.al)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.seeds_per_step = Fn._counter - c0
        with torch.no_grad():
            for p, w, s_, n in zip(params, keep_w, keep_s, keep_n):
                p.copy_(w)
                ts.optimizer.state[p]["sum"].copy_(s_)
                ts.optimizer.state[p]["step"] = n
        del keep_w, keep_s
        Fn.bump_weight_epoch()
        Fn.reset_rng(c0)
        # 2. capture one step; the seeds drawn here are baked in by value and offset by the device word at run time
        lib = _lib.load()
        self.graph = torch.cuda.CUDAGraph()
        ts.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize(dev)
        _lib.check(lib.lstc_dropout_seed_device(self.seed_word.data_ptr()), "lstc_dropout_seed_device")
        try:
            with torch.cuda.graph(self.graph):
                loss, scalars, _ = ts.forward_loss(self.nf, self.af, self.al)
                loss.backward()
                if getattr(ts.args, "clip_grad", False):
                    clip_grad_norm_(ts.encoder.parameters(), 10)
                    clip_grad_norm_(ts.head.parameters(), 10)
                ts.optimizer.step()
                self.seed_word.add_(self.seeds_per_step)
        finally:
            _lib.check(lib.lstc_dropout_seed_device(None), "lstc_dropout_seed_device")
        self.scalars = scalars
        if Fn._counter - c0 != self.seeds_per_step:
            raise RuntimeError("GraphedStep: a step must draw the same number of dropout seeds every time")
        Fn.reset_rng(c0)                      # nothing ran: the first replay IS the step with these seeds
        for p, n in zip(params, keep_n):
            ts.optimizer.state[p]["step"] = n
        # eager Adagrad.step skips parameters without a gradient (LayerNorms the reference builds but never calls): only the
        # ones the captured step really updates get their ``step`` count bumped per replay (state_dict parity with eager runs)
        self._params = [p for p in params if p.grad is not None]
        self._word_host = 0                   # host mirror of the device seed word (the graph's last node adds seeds_per_step)
        self._c0 = c0                         # host seed counter the baked seeds were drawn at: word = Fn._counter - c0 at replay

    def step(self, norm_feats, abnorm_feats, abnorm_labs):
        """One optimisation step on this batch (shapes as at construction).  Returns the five scalars (a fresh device
        tensor; no host sync)."""
        from . import functional as Fn
        if norm_feats is not self.nf:
            self.nf.copy_(norm_feats.reshape(self.nf.shape))
            self.af.copy_(abnorm_feats.reshape(self.af.shape))
            self.al.copy_(abnorm_labs.reshape(self.al.shape))
        # the graph's seeds are (baked seed + *word): keep the device word in step with the host counter, so eager seed draws
        # between replays (an evaluation pass with dropout, another TrainStep) never make two steps share a mask
        want = Fn._counter - self._c0
        if want != self._word_host:
            self.seed_word.fill_(want)
        self.graph.replay()
        self._word_host = want + self.seeds_per_step
        Fn.reset_rng(Fn._counter + self.seeds_per_step)
        for p in self._params:
            self.ts.optimizer.state[p]["step"] += 1
        Fn.bump_weight_epoch()
        return self.scalars.clone()


class MixedStep:
    """Several (encoder, head) replicas with different feature widths stepped in ONE iteration - BASELINE config 5
    (UBnormal d_model=1024 / part_len=5 mixed with ShanghaiTech d_model=2048 / part_len=3 in one batch).

    Videos of different datasets cannot share a tensor (different d_model and part_len), so the batch is a list of
    per-dataset sub-batches, each with its own model pair.  All forwards and backwards are issued first; every model's
    gradient buckets are reduced asynchronously as its backward produces them, so the all-reduces of model k overlap the
    compute of model k+1 ("gradients of both in one all-reduce bucket stream"); only then come the waits, the optional
    clipping and the Adagrad steps."""

    def __init__(self, steps):
        self.steps = list(steps)

    def step(self, batches):
        """``batches``: one ``(norm_feats, abnorm_feats, abnorm_labs)`` per TrainStep.  Returns the per-model scalar tensors."""
        scalars, ends = [], []
        for ts, (nf, af, al) in zip(self.steps, batches):
            loss, sc, _ = ts.forward_loss(nf, af, al)
            if ts.reducer is not None:
                ts.reducer.zero_grad()
            else:
                ts.optimizer.zero_grad(set_to_none=True)
            ev = getattr(ts, "comm_events", None)
            if ev is not None:
                marks = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                marks[0].record()
            loss.backward()
            if ev is not None:
                marks[1].record()
                ev.append(marks)                     # marks[2] is set below, when this model's reductions have been waited for
            scalars.append(sc)
            ends.append(None)
            if ts.reducer is not None and ts.reducer.active and loss.is_cuda:
                ends[-1] = torch.cuda.Event()
                ends[-1].record()                    # this model's backward is issued: its last bucket's weights are free from here on
        if any(getattr(ts, "comm_events", None) for ts in self.steps):
            all_done = torch.cuda.Event(enable_timing=True)
            all_done.record()                        # every model's backward is issued: exposed communication counts from here
            for ts in self.steps:
                if getattr(ts, "comm_events", None):
                    ts.comm_events[-1].append(all_done)
        for ts, end in zip(self.steps, ends):
            if ts.reducer is not None and getattr(ts, "bucket_steps", False) and not getattr(ts.args, "clip_grad", False):
                # per-bucket Adagrad + repack on a side stream as each reduction lands (TrainStep.step has the same arm)
                ts.reducer.finish(on_bucket=lambda bi, params, ts=ts: ts.optimizer.step(only=params), backward_end=end)
                if ts.comm_events:
                    m = ts.comm_events[-1]
                    m[2] = ts.reducer.landed if ts.reducer.landed is not None else m[1]
                continue
            if ts.reducer is not None:
                ts.reducer.finish()
            if getattr(ts, "comm_events", None):
                ts.comm_events[-1][2].record()
            if getattr(ts.args, "clip_grad", False):
                clip_grad_norm_(ts.encoder.parameters(), 10)
                clip_grad_norm_(ts.head.parameters(), 10)
            ts.optimizer.step()
        return scalars
