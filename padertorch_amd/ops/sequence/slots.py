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

__all__ = ['SlotLayout']


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
        self.device = torch.device(device)
        self.rows_dev = _lib.host_to_device(self.rows_host, torch.int64, self.device)
        self.occupancy = float(sum(lengths)) / float(T * S)
        self._meta = None
        self._ex_rows = {}           # padded_time -> device index (scatter and gather of a step ask for the same one)

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
