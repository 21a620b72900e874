"""Row-slot packing of ragged batches for the recurrence kernels (the training distribution's example
lengths vary, ``pit/data.py:49-77``).

A persistent recurrence launch costs its NUMBER OF TIME STEPS - ~2.5 us forward, ~3.5 us backward per step whether 1 or 32 rows
take part (``DESIGN.md`` section 3.3) -, and a PackedSequence batch runs the longest sequence's steps while its rows thin out
(lengths ~ U[3 s, 6 s]: 23 of 32 rows busy on average).  A :class:`SlotLayout` places the sequences END TO END into ``slots`` row
slots (longest-processing-time first: every sequence goes to the slot that is shortest so far), so that ``B`` sequences take about
``sum(lengths) / slots`` steps: 44 examples of that distribution fit the ~390 steps that 32 of them cost as a PackedSequence.
The kernels reset (h, c) at every sequence boundary through per-step row masks (``ptmi_lstm_forward_persistent_slots``): per
sequence the results are those of one sequence per row, i.e. of ``torch.nn.LSTM`` on the PackedSequence
(``pit/model.py:60-66,97``).  torch's PackedSequence cannot express the layout; the models keep their list-of-tensors contract and
scatter / gather at the edges (:meth:`SlotLayout.scatter_rows` / :meth:`gather_rows`: one index pass each).
"""
import functools

import numpy as np
import torch

from ... import _lib

__all__ = ['SlotLayout', 'StaticSlots']


class _SlotMeta:
    """What ``ops.lstm._LstmLayerFn`` reads of a batch layout (the fields of ``ops.lstm._PackMeta``) for rows = [T, slots]."""

    def __init__(self, layout, device):
        T, S = layout.T, layout.slots
        self.key = ('slots', S, tuple(layout.lengths), tuple(layout.slot), tuple(layout.t0))
        self.T, self.max_batch, self.rows = T, S, T * S
        self.bs_host = np.full(T, S, dtype=np.int32)
        self.offs_host = (np.arange(T, dtype=np.int64) * S)
        self.bs_dev = _lib.host_to_device(self.bs_host, torch.int32, device)
        self.offs_dev = _lib.host_to_device(self.offs_host, torch.int64, device)
        self.bs0 = S
        self.equal_lengths = False                    # no shifted-view / hand-off-plane shortcuts: rows of different sequences neighbour
        rows = self.rows
        # predecessor row per direction (forward sense), `rows` = none (the zero row ops.lstm._recurrent_operands appends)
        alive, first, last = layout.alive, layout.first, layout.last          # [T, S] bool
        idx = np.arange(rows, dtype=np.int64).reshape(T, S)
        prev = np.full((2, T, S), rows, dtype=np.int64)
        has_f = alive & ~first
        has_f[0] = False
        prev[0][has_f] = (idx - S)[has_f]
        has_r = alive & ~last
        has_r[-1] = False
        prev[1][has_r] = (idx + S)[has_r]
        self.prev_dev = _lib.host_to_device(prev.reshape(2, rows), torch.int64, device)
        bits = (1 << np.arange(S, dtype=np.uint64))
        masks = np.stack([(m.astype(np.uint64) * bits).sum(1, dtype=np.uint64) for m in (alive, first, last)], 1)     # [T, 3]
        self.masks_dev = _lib.host_to_device(masks.view(np.int64).reshape(-1), torch.int64, device)
        self.first_rows = self.last_rows = self.prev_h0_dev = self.padded_rows = None        # (initial / final states: not for slots)


class SlotLayout:
    """``lengths[b]`` frames of example ``b`` occupy slot ``slot[b]``, time indices ``t0[b] .. t0[b] + lengths[b] - 1`` of a
    ``[T, slots]`` grid (row index ``t * slots + slot``)."""

    def __init__(self, lengths, slots=32, device='cpu'):
        self._place(lengths, slots)
        self.device = torch.device(device)
        self.rows_dev = _lib.host_to_device(self.rows_host, torch.int64, self.device)
        self._meta = None
        self._ex_rows = {}           # padded_time -> device index (scatter and gather of a step ask for the same one)

    def _place(self, lengths, slots):
        """The host side of the layout: which slot and start step every sequence gets, the per-step row masks."""
        lengths = [int(n) for n in lengths]
        assert 1 <= slots <= 64 and len(lengths) >= 1 and min(lengths) >= 1, (slots, lengths)
        self.lengths, self.slots = lengths, int(slots)
        fill = np.zeros(slots, dtype=np.int64)
        self.slot, self.t0 = [0] * len(lengths), [0] * len(lengths)
        for b in sorted(range(len(lengths)), key=lambda i: (-lengths[i], i)):         # longest first, each into the emptiest slot
            s = int(np.argmin(fill))
            self.slot[b], self.t0[b] = s, int(fill[s])
            fill[s] += lengths[b]
        self.T = int(fill.max())
        T, S = self.T, self.slots
        self.alive = np.zeros((T, S), dtype=bool)
        self.first = np.zeros((T, S), dtype=bool)
        self.last = np.zeros((T, S), dtype=bool)
        rows = []
        for b, (n, s, t0) in enumerate(zip(lengths, self.slot, self.t0)):
            self.alive[t0:t0 + n, s] = True
            self.first[t0, s] = True
            self.last[t0 + n - 1, s] = True
            rows.append((t0 + np.arange(n, dtype=np.int64)) * S + s)
        #: grid row of frame t of example b, examples one after the other (the order of ``torch.cat(list_of_examples)``)
        self.rows_host = np.concatenate(rows)
        self.occupancy = float(sum(lengths)) / float(T * S)

    @staticmethod
    def cached(lengths, slots, device):
        """The layout of a length pattern (batches repeat shapes; real data brings a new pattern every step: ~1 ms of numpy)."""
        device = torch.device(device)
        return _cached_layout(tuple(int(n) for n in lengths), int(slots), (device.type, device.index))

    @property
    def meta(self):
        if self._meta is None:
            self._meta = _SlotMeta(self, self.device)
        return self._meta

    def _example_rows(self, padded_time):
        """Index of frame t of example b in a flattened batch-major ``[B, padded_time]`` tensor, examples one after the other."""
        hit = self._ex_rows.get(int(padded_time))
        if hit is None:
            idx = np.concatenate([b * padded_time + np.arange(n, dtype=np.int64) for b, n in enumerate(self.lengths)])
            hit = self._ex_rows[int(padded_time)] = _lib.host_to_device(idx, torch.int64, self.device)
        return hit

    def scatter_rows(self, padded):
        """Batch-major zero-padded ``[B, T_max, ...]`` -> grid rows ``[T * slots, ...]`` (idle rows zero); differentiable."""
        B, Tm = padded.shape[:2]
        assert B == len(self.lengths) and Tm >= max(self.lengths), (padded.shape, self.lengths)
        flat = padded.reshape(B * Tm, *padded.shape[2:])
        src = flat.index_select(0, self._example_rows(Tm))
        out = flat.new_zeros((self.T * self.slots,) + tuple(flat.shape[1:]))
        return out.index_copy(0, self.rows_dev, src)

    def gather_rows(self, grid, padded_time=None):
        """Grid rows ``[T * slots, ...]`` -> batch-major zero-padded ``[B, T_max, ...]``; differentiable."""
        Tm = max(self.lengths) if padded_time is None else int(padded_time)
        B = len(self.lengths)
        src = grid.index_select(0, self.rows_dev)
        out = grid.new_zeros((B * Tm,) + tuple(grid.shape[1:]))
        return out.index_copy(0, self._example_rows(Tm), src).view(B, Tm, *grid.shape[1:])


@functools.lru_cache(maxsize=32)
def _cached_layout(lengths, slots, device_key):
    return SlotLayout(lengths, slots, torch.device(*device_key))


class _StaticSlotMeta:
    """The fields ``ops.lstm._LstmLayerFn`` reads of a batch layout for a grid of FIXED size whose pattern is device data."""

    def __init__(self, T, S, prev_dev, masks_dev, device):
        self.key = ('static slots', S, T)
        self.T, self.max_batch, self.rows = T, S, T * S
        self.bs_host = np.full(T, S, dtype=np.int32)
        self.offs_host = (np.arange(T, dtype=np.int64) * S)
        self.bs_dev = _lib.host_to_device(self.bs_host, torch.int32, device)
        self.offs_dev = _lib.host_to_device(self.offs_host, torch.int64, device)
        self.bs0 = S
        self.equal_lengths = False
        self.prev_dev, self.masks_dev = prev_dev, masks_dev
        self.first_rows = self.last_rows = self.prev_h0_dev = self.padded_rows = None


class _GatherRows(torch.autograd.Function):
    """``out[i] = src[index[i]]`` (``index[i] == src rows``: a zero row) whose backward is the INVERSE gather: every source row goes to
    at most one output row, so ``grad_src[j] = grad_out[inverse[j]]`` - no scatter-add, no atomics, a fixed number of launches whatever
    the pattern in the two index tensors (they are device data of a captured step)."""

    @staticmethod
    def forward(ctx, src, index, inverse):
        ctx.save_for_backward(inverse)
        ctx.rows = src.shape[0]
        pad = torch.cat([src, src.new_zeros((1,) + tuple(src.shape[1:]))], 0)
        return pad.index_select(0, index)

    @staticmethod
    def backward(ctx, grad):
        inverse, = ctx.saved_tensors
        pad = torch.cat([grad, grad.new_zeros((1,) + tuple(grad.shape[1:]))], 0)
        return pad.index_select(0, inverse), None, None


class StaticSlots:
    """A row-slot layout of FIXED capacity whose length pattern is DEVICE data: ``examples`` sequences of at most ``padded_time``
    frames, end to end in ``slots`` row slots of ``steps`` time steps.  Every shape a kernel launch depends on - the grid ``[steps,
    slots]``, the batch-major padded tensors ``[examples, padded_time, ...]`` - is a constant of the object; which grid row holds which
    frame, where sequences start and end (the recurrences' per-step row masks, ``ptmi_lstm_*_persistent_slots``), the predecessor rows
    of the weight-gradient operand and the examples' frame counts are tensors that :meth:`set` rewrites for every batch.  One captured
    optimizer step (``train.graphed.GraphedStep``) therefore serves every batch that FITS - the variable-length utterances the reference
    trains on (``pit/data.py:20-33,49-77``: lengths ~ U[3 s, 6 s]) - instead of one graph per length pattern.

    An example carries the object next to its waveforms (``dict(y=..., s=..., num_samples=<int32 device tensor>, slots=StaticSlots)``);
    ``PermutationInvariantTrainingModel.forward`` takes the layout from there.  Idle grid rows (a slot shorter than ``steps``) are zero
    in every activation and contribute nothing to any gradient.
    """

    def __init__(self, examples, slots, steps, padded_time, device):
        self.examples, self.slots, self.steps, self.padded_time = int(examples), int(slots), int(steps), int(padded_time)
        assert 1 <= self.slots <= 64
        self.device = torch.device(device)
        B, S, T, Tm = self.examples, self.slots, self.steps, self.padded_time
        i64 = dict(dtype=torch.int64, device=self.device)
        #: true frame count per example (what ``PaddedList.lengths_dev`` carries: the losses read it)
        self.frames = torch.zeros(B, dtype=torch.int32, device=self.device)
        #: flat batch-major row ``b * padded_time + t`` held by grid row ``t' * slots + s`` (``examples * padded_time``: idle -> zero row)
        self.src_of_grid = torch.full((T * S,), B * Tm, **i64)
        #: the inverse: grid row of flat row ``b * padded_time + t`` (``steps * slots``: a padding frame -> zero row)
        self.grid_of_flat = torch.full((B * Tm,), T * S, **i64)
        self.prev = torch.full((2, T * S), T * S, **i64)
        self.masks = torch.zeros(3 * T, **i64)
        self.meta = _StaticSlotMeta(T, S, self.prev, self.masks, self.device)
        self.lengths = None                     # python view of the pattern last set (host bookkeeping only; no kernel reads it)
        self._host = None

    # -- what train.graphed needs of an object inside an example: its tensors in a fixed order, and what a graph bakes in of it
    def static_tensors(self):
        return [self.frames, self.src_of_grid, self.grid_of_flat, self.prev, self.masks]

    def signature(self):
        return ('StaticSlots', self.examples, self.slots, self.steps, self.padded_time, str(self.device))

    def clone(self):
        other = StaticSlots(self.examples, self.slots, self.steps, self.padded_time, self.device)
        for dst, src in zip(other.static_tensors(), self.static_tensors()):
            dst.copy_(src)
        other.lengths = self.lengths
        return other

    def fits(self, lengths):
        lengths = [int(n) for n in lengths]
        if len(lengths) != self.examples or max(lengths) > self.padded_time or min(lengths) < 1:
            return False
        return SlotLayout(lengths, self.slots).T <= self.steps if sum(lengths) <= self.steps * self.slots else False

    def set(self, lengths):
        """The pattern of a batch (``lengths[b]`` frames of example ``b``) into the device tensors: numpy on the host (~1 ms), one pinned
        staging buffer, five non-blocking copies on the current stream.  Raises ``ValueError`` when the batch does not fit."""
        lengths = [int(n) for n in lengths]
        B, S, T, Tm = self.examples, self.slots, self.steps, self.padded_time
        if len(lengths) != B or max(lengths) > Tm or min(lengths) < 1:
            raise ValueError(f'StaticSlots({B} examples of at most {Tm} frames): got {len(lengths)} examples, longest {max(lengths)}')
        lay = SlotLayout.__new__(SlotLayout)
        SlotLayout._place(lay, lengths, S)
        if lay.T > T:
            raise ValueError(f'StaticSlots: the batch needs {lay.T} steps in {S} slots, the layout has {T}')
        rows, flat_rows = lay.rows_host, np.concatenate([b * Tm + np.arange(n, dtype=np.int64) for b, n in enumerate(lengths)])
        src = np.full(T * S, B * Tm, dtype=np.int64)
        src[rows] = flat_rows
        inv = np.full(B * Tm, T * S, dtype=np.int64)
        inv[flat_rows] = rows
        alive = np.zeros((T, S), dtype=bool)
        first, last = alive.copy(), alive.copy()
        alive[:lay.T], first[:lay.T], last[:lay.T] = lay.alive, lay.first, lay.last
        idx = np.arange(T * S, dtype=np.int64).reshape(T, S)
        prev = np.full((2, T, S), T * S, dtype=np.int64)
        has_f = alive & ~first
        has_f[0] = False
        prev[0][has_f] = (idx - S)[has_f]
        has_r = alive & ~last
        has_r[-1] = False
        prev[1][has_r] = (idx + S)[has_r]
        bits = (1 << np.arange(S, dtype=np.uint64))
        masks = np.stack([(m.astype(np.uint64) * bits).sum(1, dtype=np.uint64) for m in (alive, first, last)], 1).view(np.int64).reshape(-1)
        if self.device.type != 'cuda':
            for dst, val in zip(self.static_tensors(), (np.asarray(lengths, np.int32), src, inv, prev.reshape(2, -1), masks)):
                dst.copy_(torch.from_numpy(np.ascontiguousarray(val)))
        else:
            if self._host is None:
                self._host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in self.static_tensors()]
                self._event = None
            if self._event is not None:
                self._event.synchronize()       # (the previous pattern's copies have read the pinned words)
            for host, dst, val in zip(self._host, self.static_tensors(), (np.asarray(lengths, np.int32), src, inv, prev.reshape(2, -1), masks)):
                host.copy_(torch.from_numpy(np.ascontiguousarray(val)).view(host.shape))
                dst.copy_(host, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        self.lengths = lengths
        self.occupancy = float(sum(lengths)) / float(T * S)
        return self

    def scatter_rows(self, padded):
        """Batch-major zero-padded ``[examples, padded_time, ...]`` -> grid rows ``[steps * slots, ...]`` (idle rows zero)."""
        B, Tm = padded.shape[:2]
        assert (B, Tm) == (self.examples, self.padded_time), (padded.shape, self.examples, self.padded_time)
        return _GatherRows.apply(padded.reshape(B * Tm, *padded.shape[2:]), self.src_of_grid, self.grid_of_flat)

    def gather_rows(self, grid):
        """Grid rows -> batch-major ``[examples, padded_time, ...]`` (frames past an example's own count zero)."""
        assert grid.shape[0] == self.steps * self.slots, (grid.shape, self.steps, self.slots)
        out = _GatherRows.apply(grid, self.grid_of_flat, self.src_of_grid)
        return out.view(self.examples, self.padded_time, *grid.shape[1:])
