"""Switch between packed / padded / list-of-tensor sequences.

Same public names as ``padertorch/ops/sequence/pack_module.py:17-34`` (pure data movement: the
torch ``rnn`` utilities are the implementation in the reference too), plus :class:`PaddedList`,
the container this package uses to hand a ragged batch to the HIP kernels without re-padding.
"""
import numpy as np
import torch
from torch.nn.utils.rnn import PackedSequence  # noqa: F401
from torch.nn.utils.rnn import pad_packed_sequence
from torch.nn.utils.rnn import pack_padded_sequence
from torch.nn.utils.rnn import pack_sequence as _torch_pack_sequence
from torch.nn.utils.rnn import pad_sequence

from ... import _lib

__all__ = [
    'pack_sequence',
    'unpack_sequence',
    'pad_sequence',
    'unpad_sequence',
    'pad_packed_sequence',
    'pack_padded_sequence',
    'PaddedList',
]


class PaddedList(list):
    """A python list of per-example tensors that are views into ONE padded buffer.

    Behaves exactly like the ``list`` the reference's data contract asks for
    (``pit/model.py:81-82,110``: ``list[(T_b, ...)]``, sorted by descending length), but keeps
    the padded storage and the lengths so that kernels can consume the batch in place.

    Attributes:
        padded: ``[B, T, ...]`` (``batch_first``) or ``[T, B, ...]`` tensor
        lengths: python list of ints (descending)
        lengths_dev: int32 device tensor ``[B]``
    """

    def __init__(self, padded, lengths, batch_first=True, lengths_dev=None):
        lengths = [int(l) for l in lengths]
        T = padded.shape[1 if batch_first else 0]
        rows = padded.unbind(0 if batch_first else 1)        # one call, B views
        super().__init__(r if l == T else r[:l] for r, l in zip(rows, lengths))
        self.padded = padded
        self.lengths = lengths
        self.batch_first = batch_first
        self.ragged = any(l != T for l in lengths)
        if lengths_dev is None and self.ragged:
            lengths_dev = _lib.host_to_device(lengths, torch.int32, padded.device)
        #: int32 device tensor [B], or None when every example fills the padded length
        self.lengths_dev = lengths_dev

    def intact(self):
        """Are the list's entries still the views into ``padded`` it was built with?  A caller may treat the object as the
        plain ``list`` of the reference's batch contract and replace, drop or re-order entries; consumers that read ``padded``
        instead of the entries ask first and fall back to the entries otherwise.  (In-place edits of an entry ARE edits of
        ``padded``: they need no check.)"""
        pad, bf = self.padded, self.batch_first
        if len(self) != len(self.lengths):
            return False
        step = pad.stride(0 if bf else 1) * pad.element_size()
        inner = pad.stride()[1:] if bf else pad.stride()[:1] + pad.stride()[2:]
        base = pad.data_ptr()
        return all(torch.is_tensor(v) and v.data_ptr() == base + b * step and v.shape[0] == n and v.stride() == inner
                   for b, (v, n) in enumerate(zip(self, self.lengths)))

    def to(self, device):
        if device is None:
            return self
        dev = torch.device(device)
        mine = self.padded.device
        if dev.type == mine.type and (dev.index is None or dev.index == mine.index):
            return self                 # already there: the same object (keeps what the producer attached, e.g. packed_log1p)
        ld = None if self.lengths_dev is None else self.lengths_dev.to(device)
        return PaddedList(self.padded.to(device), self.lengths, self.batch_first, ld)


def as_padded(seq, batch_first=True):
    """list of tensors (descending length) or :class:`PaddedList` -> (padded, lengths, lengths_dev)."""
    if isinstance(seq, PaddedList) and seq.intact():
        if seq.batch_first == batch_first:
            return seq.padded, seq.lengths, seq.lengths_dev
        return seq.padded.transpose(0, 1).contiguous(), seq.lengths, seq.lengths_dev
    lengths = [int(t.shape[0]) for t in seq]
    padded = pad_sequence(list(seq), batch_first=batch_first)
    ragged = any(l != lengths[0] for l in lengths)
    return padded, lengths, (_lib.host_to_device(lengths, torch.int32, padded.device) if ragged else None)


def pack_sequence(sequences, enforce_sorted=True):
    """``torch.nn.utils.rnn.pack_sequence`` (``pack_module.py:14``); PaddedList skips the re-pad."""
    if isinstance(sequences, PaddedList) and not sequences.intact():
        sequences = list(sequences)                    # entries were replaced: what the caller sees in the list is the data
    if isinstance(sequences, PaddedList) and not sequences.ragged and not sequences.batch_first \
            and sequences.padded.is_contiguous():
        padded = sequences.padded                      # time-major, equal lengths: a view
        T, B = padded.shape[:2]
        return PackedSequence(padded.view(T * B, *padded.shape[2:]), torch.full((T,), B, dtype=torch.int64))
    if isinstance(sequences, PaddedList):
        return pack_padded_sequence(sequences.padded, torch.tensor(sequences.lengths),
                                    batch_first=sequences.batch_first, enforce_sorted=enforce_sorted)
    return _torch_pack_sequence(sequences, enforce_sorted=enforce_sorted)


def unpack_sequence(packed_sequence: PackedSequence) -> list:
    """``pack_module.py:29-30``; returns a :class:`PaddedList` (time-major storage)."""
    bs = packed_sequence.batch_sizes
    if packed_sequence.sorted_indices is None and len(bs) and int(bs[0]) == int(bs[-1]):
        # equal lengths: the packed rows ARE the time-major padded tensor (no fill, no copy, and none in
        # the backward pass either)
        T, B = len(bs), int(bs[0])
        data = packed_sequence.data
        return PaddedList(data.view(T, B, *data.shape[1:]), [T] * B, batch_first=False)
    data = packed_sequence.data
    if data.is_cuda and packed_sequence.sorted_indices is None and len(bs):
        # ragged: one scatter into the zeroed time-major tensor and one gather in the backward pass (torch's pad_packed_sequence copies
        # a segment per batch-size change, and its backward one slice per TIME STEP: 1.2 ms of ~3 us launches for 32 examples of 3-6 s)
        from .. import lstm as _lstm
        meta = _lstm.pack_meta(bs, data.device)
        padded = _PackedToPadded.apply(data, meta.padded_rows, meta.T, meta.max_batch)
        lengths = (meta.bs_host[None, :] > np.arange(meta.max_batch)[:, None]).sum(1).tolist()       # (numpy: see ops.features)
        return PaddedList(padded, lengths, batch_first=False)
    padded, lengths = pad_packed_sequence(packed_sequence)
    return PaddedList(padded, lengths.tolist(), batch_first=False)


class _PackedToPadded(torch.autograd.Function):
    """``data [rows, ...]`` (PackedSequence order) -> zero-padded ``[T, B, ...]``; ``index[r]`` = row of the flattened result."""

    @staticmethod
    def forward(ctx, data, index, T, B):
        out = data.new_zeros((T * B,) + tuple(data.shape[1:]))
        out.index_copy_(0, index, data)
        ctx.save_for_backward(index)
        return out.view(T, B, *data.shape[1:])

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        return g.reshape((-1,) + tuple(g.shape[2:])).index_select(0, index), None, None, None


def unpad_sequence(padded_sequence: torch.Tensor, lengths: list):
    """``pack_module.py:33-34``."""
    return [padded_sequence[:l, b, ...] for b, l in enumerate(lengths)]
