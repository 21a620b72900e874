"""Deep-clustering embedding model, drop-in for ``padertorch/contrib/tcl/dc.py:8-84``
(identical constructor kwargs and ``state_dict`` layout ``blstm.*`` + ``linear.*``)."""
import torch
from torch.nn.utils.rnn import PackedSequence

from padertorch_amd import _lib
from padertorch_amd import base
from padertorch_amd import ops
from padertorch_amd.ops.sequence.pack_module import PaddedList, as_padded


class DeepClusteringModel(base.Model):
    #: run the BLSTM time recurrence in the hand-written HIP kernels (False: torch.nn.LSTM / MIOpen)
    hip_blstm = True
    #: ragged GPU batches on row slots (see PermutationInvariantTrainingModel.row_slots); None: off
    row_slots = None

    def __init__(
            self,
            F=257,
            recurrent_layers=2,
            units=600,
            E=20,
            input_feature_transform='identity'
    ):
        """Constructor schema of the reference (``contrib/tcl/dc.py:8-37``): ``recurrent_layers`` BLSTM layers of ``units`` cells
        per direction over ``F``-bin magnitude frames (optionally through ``input_feature_transform``: ``'identity'``,
        ``'log1p'`` or ``'log'``), then one dense layer to an ``E``-dimensional embedding per time-frequency bin."""
        super().__init__()
        self.E = E
        self.F = F
        self.input_feature_transform = input_feature_transform
        self.blstm = torch.nn.LSTM(F, units, recurrent_layers, bidirectional=True)
        self.linear = torch.nn.Linear(2 * units, F * E)

    #: feature transforms applied to the packed magnitudes ahead of the BLSTM (reference ``dc.py:48-57``)
    _TRANSFORMS = {
        'identity': lambda h: h,
        'log1p': lambda h: ops.sequence.log1p(h),
        'log': lambda h: ops.sequence.log(PackedSequence(h.data + 1e-10, h.batch_sizes)),
    }

    def _embed_rows(self, rows):
        """Packed BLSTM outputs ``[tb, 2 units]`` -> unit-norm embeddings ``[tb, E, F]`` (Hershey 2016, p. 2)."""
        e = ops.linear.linear(self.linear, rows, ops.gemm.UNIT_RANGE).view(-1, self.E, self.F)     # 'tb (e f) -> tb e f'
        if e.is_cuda and e.dtype == torch.float32:
            return ops.unit_norm(e)                    # one HIP pass forward, one backward (csrc/norm.hip)
        return torch.nn.functional.normalize(e, dim=-2)

    def forward(self, batch):
        """batch: dictionary with lists of tensors -> list of embeddings ``(T_b, E, F)``."""
        try:
            transform = self._TRANSFORMS[self.input_feature_transform]
        except KeyError:
            raise NotImplementedError(self.input_feature_transform) from None
        if isinstance(batch.get('slots'), ops.sequence.StaticSlots):
            return self._forward_static_slots(batch['Y_abs'], batch['slots'], transform)
        if self.row_slots and self.hip_blstm:
            out = self._forward_row_slots(batch['Y_abs'], transform)
            if out is not None:
                return out
        packed = getattr(batch['Y_abs'], 'packed_log1p', None)
        if packed is not None and not packed.matches(batch['Y_abs']):
            packed = None             # the list was edited since the feature kernel wrote its log-magnitudes: recompute from it
        input_planes = None
        if packed is not None and self.input_feature_transform == 'log1p':
            # written by the feature kernel itself (ops.pit_features): PackedSequence rows of log1p(Y_abs) + their fp16 planes
            h = PackedSequence(packed.data, packed.batch_sizes)
            if packed.planes() is not None:
                input_planes = (packed.planes(), ops.features.LOG1P_SCALE_WORD_VALUE)
        else:
            h = transform(ops.pack_sequence(batch['Y_abs']))
        F = h.data.shape[1]
        assert F == self.F, f'self.F = {self.F} != F = {F}'
        why = 'hip_blstm = False' if not self.hip_blstm else ops.lstm.unsupported_reason(self.blstm, h.data)
        if why is None:
            h = ops.packed_lstm(self.blstm, h, input_planes=input_planes)        # HIP time recurrence (csrc/lstm_split.hip)
        else:
            if h.data.is_cuda:
                _lib.leaving_native_path('the BLSTM of DeepClusteringModel', why)
            h = self.blstm(h)[0]
        return ops.unpack_sequence(PackedSequence(self._embed_rows(h.data), h.batch_sizes))

    def _forward_row_slots(self, Y_abs, transform):
        """``forward`` on the row-slot layout (``ops.sequence.SlotLayout``: the examples end to end in ``row_slots`` rows); ``None``
        when it does not apply (CPU tensors, equal lengths, an LSTM the kernels do not cover)."""
        padded, lengths, lengths_dev = as_padded(Y_abs)                          # [B, T_max, F], zero padded
        if not padded.is_cuda or len(set(lengths)) == 1 or ops.lstm.unsupported_reason(self.blstm, padded) is not None:
            return None
        if self.input_feature_transform == 'log':
            return None                                 # log(0 + 1e-10) of the idle rows is not zero: the PackedSequence path
        layout = ops.sequence.SlotLayout.cached(tuple(lengths), int(self.row_slots), padded.device)
        assert padded.shape[-1] == self.F, f'self.F = {self.F} != F = {padded.shape[-1]}'
        T, S = layout.T, layout.slots
        x = transform(PackedSequence(layout.scatter_rows(padded), torch.full((T,), S, dtype=torch.int64)))     # identity / log1p: 0 -> 0
        h = ops.packed_lstm(self.blstm, x, meta=layout.meta).data
        e = layout.gather_rows(self._embed_rows(h), padded.shape[1])                                             # [B, T_max, E, F]
        return PaddedList(e, lengths, True, lengths_dev)

    def _forward_static_slots(self, Y_abs, slots, transform):
        """``forward`` on a row-slot layout of fixed capacity whose length pattern is device data (``ops.sequence.StaticSlots`` carried by
        the batch as ``batch['slots']``): no launch depends on the examples' lengths - one captured optimizer step serves ragged batches
        (``train.graphed``).  Same results per example as :meth:`_forward_row_slots`."""
        padded = Y_abs.padded if isinstance(Y_abs, PaddedList) and Y_abs.intact() else as_padded(Y_abs)[0]
        why = ops.lstm.unsupported_reason(self.blstm, padded)
        assert why is None and self.hip_blstm, f'StaticSlots batches run on the HIP recurrence only ({why})'
        assert self.input_feature_transform != 'log', "log(0 + 1e-10) of the idle rows is not zero: 'identity' / 'log1p' only on row slots"
        assert padded.shape[-1] == self.F, f'self.F = {self.F} != F = {padded.shape[-1]}'
        T, S = slots.steps, slots.slots
        x = transform(PackedSequence(slots.scatter_rows(padded), torch.full((T,), S, dtype=torch.int64)))
        h = ops.packed_lstm(self.blstm, x, meta=slots.meta).data
        e = slots.gather_rows(self._embed_rows(h))                               # [B, padded_time, E, F], padding frames zero
        return PaddedList(e, [slots.padded_time] * slots.examples, True, slots.frames)

    def review(self, batch, model_out):
        """Mean deep-clustering loss of the batch (reference ``dc.py:73-84``: per-example loop over
        re-laid-out copies).  A :class:`PaddedList` output is consumed in place by ONE fused HIP
        pass; anything else follows the reference loop through ``deep_clustering_loss``."""
        wide = len(model_out) > 0 and model_out[0].shape[1] + batch['target_mask'][0].shape[1] > 32       # E + K > 32: the per-example loop below
        if isinstance(model_out, PaddedList) and model_out.intact() and not wide:
            tm, _, _ = as_padded(batch['target_mask'])
            loss, _ = ops.losses.dc_loss_batched(
                model_out.padded, tm, model_out.lengths_dev,
                embedding_batch_first=model_out.batch_first)
            return {'losses': {'dc_loss': loss}}
        dc_loss = list()
        for embedding, target_mask in zip(model_out, batch['target_mask']):
            E, K = embedding.shape[1], target_mask.shape[1]
            dc_loss.append(ops.losses.deep_clustering_loss(
                embedding.permute(0, 2, 1).reshape(-1, E),        # 't e f -> (t f) e'
                target_mask.permute(0, 2, 1).reshape(-1, K),      # 't k f -> (t f) k'
            ))
        return {'losses': {'dc_loss': torch.mean(torch.stack(dc_loss))}}
