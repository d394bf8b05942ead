"""Host -> HBM batch staging: pinned double buffers + a copy stream, so the H2D transfer of batch i+1 overlaps the
training step of batch i.

The reference moves every batch with a synchronous ``.cuda()`` on the compute stream
(Train/temporal_transformer_shanghaitech.py:115-118): 805 MB per step at the headline LTN shape = 12.8 ms at PCIe Gen5
x16 (63 GB/s) in front of every step.  Here the loader thread's tensors are copied into two pinned host slabs and sent
with ``non_blocking=True`` on a side HIP stream; the compute stream only waits on the copy's event.
torch is used purely for memory/stream plumbing."""
from __future__ import annotations

import torch


class PinnedFeeder:
    def __init__(self, batches, device, depth: int = 2):
        """``batches``: iterable of tuples of CPU tensors (the DataLoader contract: norm_feats, norm_labs, abnorm_feats,
        abnorm_labs).  Yields the same tuples as device tensors."""
        self.batches, self.device, self.depth = batches, torch.device(device), depth
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._slabs = [None] * depth          # pinned host buffers, reused
        self._dev = [None] * depth            # device buffers, reused (the step must finish before slot reuse)
        self._ready = [None] * depth          # copy-done events
        self._consumed = [None] * depth       # compute-done events (protect device buffer reuse)

    def _stage(self, slot, batch):
        if self._slabs[slot] is None or any(s.shape != b.shape or s.dtype != b.dtype for s, b in zip(self._slabs[slot], batch)):
            self._slabs[slot] = [torch.empty(b.shape, dtype=b.dtype).pin_memory() for b in batch]
            self._dev[slot] = [torch.empty(b.shape, dtype=b.dtype, device=self.device) for b in batch]
        if self._ready[slot] is not None:
            self._ready[slot].synchronize()            # previous H2D from this pinned slab has completed
        for s, b in zip(self._slabs[slot], batch):
            s.copy_(b)                                  # pageable -> pinned (host memcpy)
        with torch.cuda.stream(self.copy_stream):
            if self._consumed[slot] is not None:
                self.copy_stream.wait_event(self._consumed[slot])   # the step that read this device slot is done
            for d, s in zip(self._dev[slot], self._slabs[slot]):
                d.copy_(s, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._ready[slot] = ev

    def __iter__(self):
        it = iter(self.batches)
        pending = []
        slot = 0
        for _ in range(self.depth - 1):
            b = next(it, None)
            if b is None:
                break
            self._stage(slot, b)
            pending.append(slot)
            slot = (slot + 1) % self.depth
        while pending:
            b = next(it, None)
            if b is not None:
                self._stage(slot, b)
                pending.append(slot)
                slot = (slot + 1) % self.depth
            cur = pending.pop(0)
            torch.cuda.current_stream(self.device).wait_event(self._ready[cur])
            yield tuple(self._dev[cur])
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._consumed[cur] = ev


class LazyRows:
    """``bank[idx]`` that has NOT been formed: one half (``kind`` 0 = normal, 1 = abnormal videos) of a training batch
    ``[2, bs, rows]`` of clip indices into an HBM-resident bank ``[clips, P, d]``.  ``engine.TrainStep`` hands the index vector to
    the fused gather + CLS-concat kernel (lstc_cls_concat_gather_fwd: the gathered batch is never written); everything else calls
    ``materialize()`` (lstc_gather_rows) and sees the ordinary ``[bs, rows, P, d]`` tensor."""

    def __init__(self, bank, idx_flat, kind, bs, rows):
        self.bank, self.idx_flat, self.kind, self.bs, self.rows = bank, idx_flat, kind, bs, rows
        self.shape = (bs, rows) + tuple(bank.shape[1:])
        self.device = bank.device

    def idx(self):
        n = self.bs * self.rows
        return self.idx_flat[self.kind * n:(self.kind + 1) * n]

    def materialize(self):
        from . import functional as F
        return F.gather_rows(self.bank, self.idx()).reshape(self.shape)

    def pairs_with(self, other) -> bool:
        """Is ``other`` the abnormal half of the batch this is the normal half of (one flat index vector, normal first)?"""
        return (isinstance(other, LazyRows) and other.idx_flat is self.idx_flat and self.kind == 0 and other.kind == 1 and
                (other.bs, other.rows) == (self.bs, self.rows))


class ResidentBank:
    """Features resident in HBM as ONE ``[total_clips, ...row]`` tensor; a training batch is formed on the device by
    ``lstc_gather_rows`` from clip indices the host sampler produced (MI355X-first data feed: a whole training set fits in
    288 GB, so no feature bytes cross PCIe per step).

    The index / label arrays of a step (a few tens of KB) travel through a ring of pinned host slabs with truly
    asynchronous copies on the compute stream, so the host keeps running ahead of the GPU (a pageable ``.to(device)`` would
    block the host until the previous step has drained)."""

    def __init__(self, bank: torch.Tensor, depth: int = 4):
        from . import functional as F
        self.F, self.bank, self.depth = F, bank, depth
        self._slot = 0
        self._pinned = [None] * depth
        self._done = [None] * depth

    def _stage(self, arrays):
        import numpy as np
        s = self._slot
        self._slot = (s + 1) % self.depth
        if self._done[s] is not None:
            self._done[s].synchronize()                 # the copies issued from this slab `depth` steps ago have completed
        slabs = self._pinned[s]
        if slabs is None or any(p.shape != tuple(a.shape) or p.numpy().dtype != a.dtype for p, a in zip(slabs, arrays)):
            slabs = self._pinned[s] = [torch.from_numpy(np.empty(a.shape, a.dtype)).pin_memory() for a in arrays]
        outs = []
        for p, a in zip(slabs, arrays):
            p.numpy()[...] = a
            outs.append(p.to(self.bank.device, non_blocking=True))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.bank.device))
        self._done[s] = ev
        return outs

    def gather(self, idx, *extras, lazy=False):
        """``idx``: int64 numpy array of clip rows (any shape); returns ``bank[idx]`` with shape ``idx.shape + row`` plus the
        ``extras`` (small numpy arrays, e.g. labels) as device tensors.  ``lazy`` (``idx`` of shape [2, bs, rows]): the two
        halves come back as ``LazyRows`` - the indices are on the device, the rows are gathered by whoever consumes them."""
        # the fused gather reads bank[(idx * P + patch) ...] with no bound of its own (lstc_cls_concat_gather_fwd, lstc_gather_rows):
        # a stale or negative clip index would be a silent out-of-bounds read of HBM - reject it here, on a few hundred integers
        if idx.size and (int(idx.min()) < 0 or int(idx.max()) >= int(self.bank.shape[0])):
            raise IndexError(f"clip index out of range for a bank of {int(self.bank.shape[0])} clips: [{int(idx.min())}, {int(idx.max())}]")
        dev = self._stage([idx.reshape(-1)] + list(extras))
        if lazy:
            if idx.ndim != 3 or idx.shape[0] != 2:
                raise ValueError("gather(lazy=True) takes the [2, bs, rows] index array of a normal / abnormal pair batch")
            return ((LazyRows(self.bank, dev[0], 0, idx.shape[1], idx.shape[2]), LazyRows(self.bank, dev[0], 1, idx.shape[1], idx.shape[2])),
                    *dev[1:])
        out = self.F.gather_rows(self.bank, dev[0]).reshape(tuple(idx.shape) + tuple(self.bank.shape[1:]))
        return (out, *dev[1:])
