"""Data-parallel gradient exchange: ONE sum-all-reduce of the gradients per step over RCCL/xGMI.

The reference's only multi-GPU mechanism is single-process ``nn.DataParallel`` (e.g.
Train/temporal_transformer_shanghaitech.py:76-78): per step it broadcasts every parameter (~403 MB), gathers
the full encoder output (~822 MB) to GPU 0 and reduces the gradients there.  Here each GPU
is its own process holding a replica; videos are sharded across ranks (sequences are independent through
encoder and head), the loss kernel divides by GLOBAL counts, so the only exchange is

  * 2*bs floats of bag maxima inside the loss (lstc_vad_amd.functional.VadLossFunction), and
  * a SUM all-reduce of the parameter gradients — bucketed per encoder layer in backward order and
    launched asynchronously from autograd hooks, so the reduction of layer i overlaps the backward
    of layers < i.  xGMI is point-to-point (7 links/GPU): a few large buckets (~130 MB each for the
    LTN) keep every link busy with large messages instead of many small ones.

Parameters that never receive a gradient (LayerNorms the reference constructs but does not call) are left
out of the buckets, mirroring Adagrad skipping ``grad is None`` entries.
The class is device-agnostic (works with ``gloo`` on CPU tensors), which is how tests/test_dist_cpu.py covers it.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


class _Done:
    """Handle of a collective that has already completed (host-staged gloo path)."""

    def wait(self):
        return True


def all_reduce_sum(t: torch.Tensor, group=None, async_op: bool = False):
    """SUM all-reduce of a device (or host) tensor.  Over RCCL (backend "nccl"): ``dist.all_reduce`` on the device, asynchronous if
    asked.  Over gloo with a DEVICE tensor - the functional checks that put several ranks on one GPU (LSTC_SHARE_DEVICE), never a
    measured configuration - the tensor is staged through the host HERE, synchronously: torch's own gloo path (pinned staging
    copies on side streams, reduction on worker threads) returned wrong sums in roughly one of six 8-rank runs with eight
    processes time-slicing one MI355X (round 6: tools/r06_flake_probe8.sh), with every stream dependency of this
    repository's side in place.  ``LSTC_GLOO_DEVICE_TENSORS=1`` restores torch's path.  Returns a handle with ``wait()``."""
    import os
    if t.is_cuda and dist.get_backend(group) == "gloo" and os.environ.get("LSTC_GLOO_DEVICE_TENSORS", "0") != "1":
        torch.cuda.current_stream(t.device).synchronize()
        h = t.detach().to("cpu")
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return _Done()
    h = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return h if async_op else _Done()


class GradAllReducer:
    def __init__(self, param_groups: Sequence[Iterable[torch.nn.Parameter]], group=None, overlap: bool = True,
                 force: bool = False, reduce_dtype: str = "fp32", direct: Iterable[torch.nn.Parameter] = ()):
        """``param_groups``: lists of parameters, one per bucket, ordered the way backward produces them
        (head first, last encoder layer next, ...).  Every listed parameter MUST receive a gradient each step.
        ``direct``: parameters whose backward WRITES the gradient into the bucket itself (``functional.grad_sink`` /
        ``deliver``: the weight-gradient GEMM's output, or the ordered sum of its split-K partials, lands in the bucket - no
        fill, no accumulate pass; ``direct_grad_parameters`` lists them for an Encoder + head pair).  They are laid out first
        in their bucket; only the remainder (biases, LayerNorms, bias tables: a few hundred KB) is zero-filled per step and
        accumulated into by autograd."""
        self.group = group
        self.overlap = overlap
        # reduce_dtype "bf16": a bucket is rounded to bf16 (lstc_cast_f32_bf16), summed by RCCL at half the bytes and widened
        # back into the f32 bucket (for the bf16 compute mode at 8 GPUs, where a rank's step is ~10 ms against 407 MB of fp32
        # gradients).  Off by default: the fp32 reduction is what reproduces the reference's single-process gradients.
        if reduce_dtype not in ("fp32", "bf16"):
            raise ValueError(reduce_dtype)
        self.reduce_dtype = reduce_dtype
        self._half: List[torch.Tensor] = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())     # force: exercise the RCCL path on one GPU
        self.buckets: List[torch.Tensor] = []
        self._views = []
        self._pending: List[int] = []
        self._sizes: List[int] = []
        self._handles = []
        self._bucket_of = {}
        self._view_ptr = {}
        self._accum_from: List[int] = []           # per bucket: first element of the autograd-accumulated (zero-filled) tail
        self._direct: List[torch.nn.Parameter] = []       # parameters with a gradient sink, and which of them delivered this step
        self._delivered = set()
        self._ready_ev = {}                        # bucket -> event on the launch stream where its all-reduce was issued (per-bucket steps)
        self._side = self._land = None             # streams of finish(on_bucket=...)
        self.landed = None                         # event: the LAST bucket's reduction has landed (exposed-communication measurement)
        direct_ids = {id(p) for p in direct}
        for bi, params in enumerate(param_groups):
            params = [p for p in params if p.requires_grad]
            params = [p for p in params if id(p) in direct_ids] + [p for p in params if id(p) not in direct_ids]   # stable
            n = sum(p.numel() for p in params)
            if n == 0:
                continue
            flat = torch.zeros(n, device=params[0].device, dtype=params[0].dtype)
            off = 0
            views = []
            self._accum_from.append(sum(p.numel() for p in params if id(p) in direct_ids))
            for p in params:
                if id(p) in direct_ids and off % 4:
                    # a sink is the C operand of a GEMM / the destination of a 16-B-vectorised ordered sum: the one-pass and the
                    # two-pass column sums add in different orders, so "same values as the plain path" needs the aligned form
                    raise RuntimeError("GradAllReducer: a direct gradient slot must start on a 16-byte boundary (every direct "
                                       "weight in front of it needs numel % 4 == 0)")
                v = flat[off:off + p.numel()].view_as(p)
                p.grad = v                       # the optimizer reads the gradient here either way
                views.append((p, v))
                off += p.numel()
                self._bucket_of[p] = len(self.buckets)
                self._view_ptr[p] = v.data_ptr()
                if id(p) in direct_ids:
                    # the producing kernel writes into ``v`` and the Function calls back (functional.deliver): always installed -
                    # also on one rank without a process group - so the single-rank bucket path runs what N ranks run
                    p.__dict__["_lstc_grad_sink"] = (v, self._sink_done)
                    self._direct.append(p)
                elif self.active and overlap:
                    p.register_post_accumulate_grad_hook(self._hook)    # autograd accumulates in place into the bucket
            self.buckets.append(flat)
            if reduce_dtype == "bf16":
                self._half.append(torch.empty(n, device=flat.device, dtype=torch.bfloat16))
            self._views.append(views)
            self._sizes.append(len(params))
            self._pending.append(len(params))

    # ------------------------------------------------------------------------------------------
    def zero_grad(self):
        """Replaces ``optimizer.zero_grad()``: clears the flat buckets and re-arms the hooks."""
        for bi, flat in enumerate(self.buckets):
            if self._accum_from[bi] < flat.numel():
                flat[self._accum_from[bi]:].zero_()        # direct gradients are overwritten by their kernels: no fill
            self._pending[bi] = self._sizes[bi]
            for p, v in self._views[bi]:
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    p.grad = v
        self._handles = []
        self._delivered.clear()

    def _sink_done(self, p):
        """A direct gradient has been issued into its slot (called from the Function's backward, on the launch stream).  The
        bookkeeping runs in EVERY mode (one rank without a process group, overlap off): a sink is WRITTEN, not accumulated into,
        so a weight that delivers twice between two zero_grad() calls (shared by two Functions, or two backward passes) would
        silently keep only its last gradient while its biases accumulate through autograd - refuse that loudly."""
        if id(p) in self._delivered:
            raise RuntimeError("GradAllReducer: a weight delivered its gradient twice since zero_grad() (its bucket slot is written, "
                               "not accumulated: shared weights / several backward passes per step are not supported by the "
                               "bucket path)")
        self._delivered.add(id(p))
        if self.active and self.overlap:
            self._hook(p)

    def _check_direct(self):
        """Every direct weight whose sink is in force must have delivered exactly once this step: zero_grad() does not clear the
        direct region, so a weight whose Function did not run would hand LAST step's gradient to the optimizer."""
        missing = [p for p in self._direct if id(p) not in self._delivered and p.grad is not None
                   and p.grad.data_ptr() == self._view_ptr[p]]
        if missing:
            raise RuntimeError(f"GradAllReducer: {len(missing)} direct weight gradient(s) were not produced this step "
                               f"(shapes {[tuple(p.shape) for p in missing[:4]]}): their bucket slots hold stale values")

    def _hook(self, p):
        bi = self._bucket_of[p]
        if p.grad is None or p.grad.data_ptr() != self._view_ptr[p]:
            return            # somebody else owns .grad now (the model is being used without this reducer): not our step
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            if self.buckets[bi].is_cuda:
                ev = self._ready_ev.get(bi)
                if ev is None:
                    ev = self._ready_ev[bi] = torch.cuda.Event()
                ev.record()                        # everything the backward has issued so far, on the launch stream
            self._handles.append(self._launch(bi, True))
        elif self._pending[bi] < 0:
            # a parameter reported twice in one backward (a weight used by two Functions): a direct gradient would have been
            # OVERWRITTEN in its bucket slot, and the bucket's all-reduce has already been launched
            raise RuntimeError("GradAllReducer: a parameter delivered its gradient twice in one backward pass (shared weights are "
                               "not supported by the bucket path; call zero_grad() before every backward)")

    def finish(self, on_bucket=None, backward_end=None):
        """Call after ``backward()`` and before the optimizer step.

        ``on_bucket(bi, params)`` (engine.TrainStep passes its per-bucket Adagrad + weight repack when
        ``--clip_grad`` is off): instead of waiting for ALL buckets and then stepping everything, bucket k is stepped as soon as
        ITS reduction has landed - on a side stream, beside the backward of the layers below it.  Order on the side stream =
        the order the buckets became ready (head, then layers last to first).  Bucket k's weights may still be read by the
        backward kernels issued before bucket k + 1 became ready (a Function delivers a weight gradient before it has issued
        every product that reads the weight), so the side stream waits for the launch-stream event recorded when the NEXT bucket's
        all-reduce was issued (the end of the backward for the last one) before it touches bucket k's weights.  The launch stream
        waits for the side stream at the end: the next forward sees every weight updated.  Weights are bit-identical to the
        all-buckets-then-one-step order (the update is elementwise)."""
        self._check_direct()
        self.landed = None
        if not self.active:
            if on_bucket is not None:
                for bi in range(len(self.buckets)):
                    on_bucket(bi, [p for p, _ in self._views[bi]])
            return
        if on_bucket is not None and self.overlap:
            if any(n != 0 for n in self._pending):
                missing = [bi for bi, n in enumerate(self._pending) if n]
                raise RuntimeError(f"GradAllReducer: buckets {missing} did not receive all gradients this step")
            if not self.buckets[0].is_cuda:                      # host tensors (gloo tests): no streams to play with
                for bi, h in self._handles:
                    h.wait()
                    self._widen(bi)
                    on_bucket(bi, [p for p, _ in self._views[bi]])
                self._handles = []
                return
            dev = self.buckets[0].device
            main = torch.cuda.current_stream(dev)
            if self._side is None:
                self._side, self._land = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            end = backward_end                                   # MixedStep: this model's backward ended before the next model's began
            if end is None:
                end = torch.cuda.Event()
                end.record(main)
            order = [bi for bi, _ in self._handles]
            for j, (bi, h) in enumerate(self._handles):
                gate = self._ready_ev[order[j + 1]] if j + 1 < len(order) else end
                self._side.wait_event(gate)
                with torch.cuda.stream(self._side):
                    h.wait()                                     # the side stream waits for this bucket's reduction only
                    self._widen(bi)
                    on_bucket(bi, [p for p, _ in self._views[bi]])
            if self._handles:
                # when did the LAST reduction land?  (its own stream: the side stream may still be stepping earlier buckets)
                with torch.cuda.stream(self._land):
                    self._handles[-1][1].wait()
                    self.landed = torch.cuda.Event(enable_timing=True)
                    self.landed.record()
            main.wait_stream(self._side)
            main.wait_stream(self._land)
            self._handles = []
            return
        if not self.overlap:
            for bi in range(len(self.buckets)):
                self._launch(bi, False)
                self._widen(bi)
            return
        if any(n != 0 for n in self._pending):
            missing = [bi for bi, n in enumerate(self._pending) if n]
            raise RuntimeError(f"GradAllReducer: buckets {missing} did not receive all gradients this step")
        for bi_h in self._handles:
            bi, h = bi_h
            h.wait()
            self._widen(bi)
        self._handles = []

    def _launch(self, bi, async_op):
        """All-reduce bucket ``bi`` (after rounding it to bf16 when reduce_dtype == "bf16"); returns (bucket, handle)."""
        flat = self.buckets[bi]
        if self.reduce_dtype == "bf16":
            from . import _lib
            half = self._half[bi]
            _lib.check(_lib.load().lstc_cast_f32_bf16(_lib.dev_ptr(flat), _lib.dev_ptr(half), flat.numel(), _lib.stream_ptr()),
                       "lstc_cast_f32_bf16")
            flat = half
        h = all_reduce_sum(flat, self.group, async_op=async_op)
        return (bi, h)

    def _widen(self, bi):
        if self.reduce_dtype == "bf16":
            from . import _lib
            flat, half = self.buckets[bi], self._half[bi]
            _lib.check(_lib.load().lstc_cast_bf16_f32(_lib.dev_ptr(half), _lib.dev_ptr(flat), flat.numel(), _lib.stream_ptr()),
                       "lstc_cast_bf16_f32")

    def payload_bytes(self) -> int:
        src = self._half if self.reduce_dtype == "bf16" else self.buckets
        return sum(b.numel() * b.element_size() for b in src)


def direct_grad_parameters(encoder, head) -> List[torch.nn.Parameter]:
    """The weights of an ``Encoder`` + head pair whose gradient kernels write into a gradient sink (lstc_vad_amd.functional:
    MHAFunction / MHAClsFunction / MHAClsAssocFunction / FFNFunction / HeadFunction): the four attention projections and the two
    FFN matrices of every layer and the head's first Linear - 99.9 % of the gradient bytes."""
    out = []
    for layer in encoder.layer_stack:
        a = layer.slf_attn
        out += [a.w_qs.weight, a.w_ks.weight, a.w_vs.weight, a.fc.weight]
        if getattr(layer, "FFN_need", True):
            out += [layer.pos_ffn.w_1.weight, layer.pos_ffn.w_2.weight]
    seq = getattr(head, "classifier", None) or getattr(head, "regressor", None)
    if seq is not None:
        out.append(seq[0].weight)
    return out


def encoder_head_buckets(encoder, head, n_buckets=None) -> List[List[torch.nn.Parameter]]:
    """Backward-ordered buckets for an ``Encoder`` + head pair: head, then encoder layers last to first
    (layer-level parameters that feed the first layer — cls_token / position_enc / input LayerNorm — go last).
    ``n_buckets`` (None = one bucket per layer + head [+ rest]; "finest" = every piece its own bucket - what engine.TrainStep uses:
    the bucket that becomes ready LAST, layer 0's attention weights, is the one whose all-reduce no backward work can hide, so it
    should be as small as a still-large message allows: 67 MB instead of 134 MB at the LTN widths): the finest backward-ordered pieces - head, then per
    layer its FFN parameters and its attention parameters, layers last to first, then the rest - merged into that many
    consecutive groups of roughly equal bytes (1 = a single all-reduce after the backward; up to 2 x layers + 2).  xGMI is
    point-to-point, so few large messages are the starting point; the knob exists so one run can be compared with another
    (bench.py --buckets)."""
    used = {id(p) for p in encoder.used_parameters()}
    pieces = [list(head.parameters())]
    per_layer = []
    in_layers = set()
    for layer in reversed(list(encoder.layer_stack)):
        ps = [p for p in layer.parameters() if id(p) in used]
        in_layers.update(id(p) for p in ps)
        ffn_ids = {id(p) for p in layer.pos_ffn.parameters()} if hasattr(layer, "pos_ffn") else set()
        ffn = [p for p in ps if id(p) in ffn_ids]
        attn = [p for p in ps if id(p) not in ffn_ids]
        per_layer.append(ps)
        pieces += [g for g in (ffn, attn) if g]              # backward reaches a layer's FFN before its attention
    rest = [p for p in encoder.parameters() if id(p) in used and id(p) not in in_layers]
    if rest:
        pieces.append(rest)
    if n_buckets is None:
        return [list(head.parameters())] + per_layer + ([rest] if rest else [])
    if n_buckets == "finest":
        return [g for g in pieces if g]
    n_buckets = max(1, min(int(n_buckets), len(pieces)))
    size = lambda g: sum(p.numel() for p in g)
    total, out, cur, acc = sum(size(g) for g in pieces), [], [], 0
    for i, g in enumerate(pieces):
        cur += g
        acc += size(g)
        left_pieces, left_groups = len(pieces) - i - 1, n_buckets - len(out) - 1
        if left_groups > 0 and (acc >= total * (len(out) + 1) / n_buckets or left_pieces == left_groups):
            out.append(cur); cur = []
    out.append(cur)
    return [g for g in out if g]
