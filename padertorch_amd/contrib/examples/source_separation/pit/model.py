"""BLSTM mask estimator trained with permutation-invariant training, MI355X-native.

Drop-in for ``padertorch/contrib/examples/source_separation/pit/model.py:11-151``
(``PermutationInvariantTrainingModel``): identical constructor kwargs, identical ``state_dict``
keys/shapes (``blstm.*``, ``linear1.*``, ``linear2.*``; gate order i,f,g,o, two bias vectors), the
same ``forward(batch) -> list[(T_b, K, F)]`` / ``review(batch, out) -> dict`` contract.

What changed underneath:
  * ``review`` (reference ``:112-140``: python loop over examples x 2 losses x K! permutations)
    is ONE fused HIP pass over the whole ragged batch (``pit_mse_ips_losses``);
  * the model output stays in the time-major padded buffer the packed BLSTM produced; the list
    handed back is a :class:`PaddedList` of views, so nothing is re-padded or copied;
  * the summary images (``:142-147``) are only rendered when ``self.create_snapshot`` is set
    (allowed by ``base.py:300-306``), which removes a device->host sync from every step;
  * ``batch`` may carry raw waveforms (``y``, ``s``, the keys of ``pit/data.py:66-68``) instead of
    precomputed features; they are turned into ``Y_abs`` / ``X_abs`` / ``cos_phase_difference`` on
    the device by the fused STFT front-end (``ops.pit_features``).
"""
import torch
from torch.nn.utils.rnn import PackedSequence

from padertorch_amd import _lib
from padertorch_amd import base
from padertorch_amd import ops
from padertorch_amd.ops.mappings import ACTIVATION_FN_MAP
from padertorch_amd.ops.sequence.pack_module import PaddedList, as_padded
from padertorch_amd.summary import mask_to_image, stft_to_image


class PermutationInvariantTrainingModel(base.Model):
    """Implements a variant of Permutation Invariant Training [1].

    [1] Kolbaek 2017, https://arxiv.org/pdf/1703.06284.pdf
    """
    #: run the BLSTM time recurrence in the hand-written HIP kernels (False: torch.nn.LSTM / MIOpen)
    hip_blstm = True
    #: ragged batches on the GPU: place the examples END TO END into this many row slots (16 / 32 / 64; ops.sequence.SlotLayout) so that
    #: the recurrences run ~ sum(frames) / row_slots steps instead of the longest example's - worth it when the batch has more
    #: examples than slots (the data pipeline forms batches of ~1.4 x row_slots examples of the U[3 s, 6 s] distribution); None: off
    row_slots = None

    def __init__(
            self,
            F=257,
            recurrent_layers=3,
            units=600,
            K=2,
            dropout_input=0.,
            dropout_hidden=0.,
            dropout_linear=0.,
            output_activation='relu'
    ):
        """Constructor schema of the reference (``pit/model.py:27-66``; the names are the config keys of its checkpoints).

        ``F`` frequency bins per frame (``size // 2 + 1`` of the STFT) go through ``recurrent_layers`` BLSTM layers of ``units``
        cells per direction and two dense layers to ``K`` masks per bin, squashed by ``output_activation`` (a key of
        ``ACTIVATION_FN_MAP``).  The three dropout probabilities (each at most 0.5) act on the network input, between the
        BLSTM layers (``torch.nn.LSTM``'s own ``dropout``) and in front of the first dense layer.
        """
        super().__init__()
        self.K = K
        self.F = F

        assert dropout_input <= 0.5, dropout_input
        self.dropout_input = torch.nn.Dropout(dropout_input)
        assert dropout_hidden <= 0.5, dropout_hidden
        self.blstm = torch.nn.LSTM(F, units, recurrent_layers, bidirectional=True,
                                   dropout=dropout_hidden)
        assert dropout_linear <= 0.5, dropout_linear
        self.dropout_linear = torch.nn.Dropout(dropout_linear)
        self.relu = torch.nn.ReLU()
        self.linear1 = torch.nn.Linear(2 * units, 2 * units)
        self.linear2 = torch.nn.Linear(2 * units, F * K)
        self.output_activation = ACTIVATION_FN_MAP[output_activation]()

    def prepare_batch(self, batch):
        """Waveforms -> features on the device when the batch does not carry them yet."""
        if 'Y_abs' in batch or 'y' not in batch:
            return batch
        slots = batch.get('slots')
        feats = ops.pit_features(batch['y'], batch.get('s'), batch.get('num_samples'),
                                 num_frames_dev=slots.frames if isinstance(slots, ops.sequence.StaticSlots) else None)
        out = dict(batch)
        out.update(feats)
        return out

    def example_to_device(self, example, device=None, memo=None):
        return self.prepare_batch(super().example_to_device(example, device, memo))

    def forward(self, batch):
        """
        Args:
            batch: Dictionary with lists of tensors (``Y_abs[b]: (T_b, F)``, descending ``T_b``)

        Returns: List of mask tensors, each list element has shape (T, K, F)
        """
        batch = self.prepare_batch(batch)
        if isinstance(batch.get('slots'), ops.sequence.StaticSlots):
            return self._forward_static_slots(batch['Y_abs'], batch['slots'])
        if self.row_slots and self.hip_blstm:
            out = self._forward_row_slots(batch['Y_abs'])
            if out is not None:
                return out
        packed = getattr(batch['Y_abs'], 'packed_log1p', None)
        if packed is not None and not packed.matches(batch['Y_abs']):
            packed = None             # the list was edited since the feature kernel wrote its log-magnitudes: recompute from it
        input_planes = None
        if packed is not None and not (self.training and self.dropout_input.p > 0):
            # the feature kernel has written log1p(Y_abs) in PackedSequence order itself (and as fp16 planes for the first
            # projection): no pack_sequence / log1p / scale / split pass (ops.pit_features, csrc/stft.hip)
            h = PackedSequence(packed.data, packed.batch_sizes)
            if packed.planes() is not None:
                input_planes = (packed.planes(), ops.features.LOG1P_SCALE_WORD_VALUE)
            F = h.data.shape[1]
            assert F == self.F, f'self.F = {self.F} != F = {F}'
        else:
            h = ops.pack_sequence(batch['Y_abs'])

            _, F = h.data.size()
            assert F == self.F, f'self.F = {self.F} != F = {F}'

            h_data = self.dropout_input(h.data)
            h_data = ops.sequence.log1p(h_data)
            h = PackedSequence(h_data, h.batch_sizes)

        # Returns tensor with shape (t, b, num_directions * hidden_size)
        why = 'hip_blstm = False' if not self.hip_blstm else ops.lstm.unsupported_reason(self.blstm, h.data)
        if why is None:
            h = ops.packed_lstm(self.blstm, h, input_planes=input_planes)        # HIP time recurrence (csrc/lstm_split.hip)
        else:
            if h.data.is_cuda:                        # (CPU tensors - the reference Trainer's test_run on the host - are torch's business)
                _lib.leaving_native_path('the BLSTM of PermutationInvariantTrainingModel', why)
            h, _ = self.blstm(h)                      # library LSTM (MIOpen)

        h_data = self.dropout_linear(h.data)
        # BLSTM outputs lie in (-1, 1) (x 2 at most under dropout): no operand scaling pass for the split GEMM
        h_data = self._dense(h_data)

        mask = PackedSequence(h_data.view(-1, self.K, self.F), h.batch_sizes)  # 'tb (k f) -> tb k f'
        return ops.unpack_sequence(mask)

    def _dense(self, h):
        """linear1 -> relu -> linear2 -> output activation on BLSTM output rows (``pit/model.py:98-104``).  A plain ``torch.nn.ReLU``
        behind a layer is computed in that layer's GEMM epilogue (``ops.linear.linear(..., activation='relu')``), which also leaves the
        next operand's scale behind; any other activation module is applied as it is."""
        a1 = 'relu' if type(self.relu) is torch.nn.ReLU else None
        a2 = 'relu' if type(self.output_activation) is torch.nn.ReLU else None
        # BLSTM outputs lie in (-1, 1) (x 2 at most under dropout): no operand scaling pass for the split GEMM
        h = ops.linear.linear(self.linear1, h, ops.gemm.UNIT_RANGE, activation=a1)
        if a1 is None:
            h = self.relu(h)
        h = ops.linear.linear(self.linear2, h, activation=a2)
        if a2 is None:
            h = self.output_activation(h)
        return h

    def _forward_row_slots(self, Y_abs):
        """``forward`` for a ragged batch on the row-slot layout (``row_slots``): the same network, rows = [T, slots] with the
        examples end to end in the slots; returns the masks as a batch-major :class:`PaddedList` (what ``review`` consumes), or
        ``None`` when the layout does not apply (CPU tensors, equal lengths, an LSTM the kernels do not cover)."""
        padded, lengths, lengths_dev = as_padded(Y_abs)                          # [B, T_max, F], zero padded
        if not padded.is_cuda or len(set(lengths)) == 1 or ops.lstm.unsupported_reason(self.blstm, padded) is not None:
            return None
        layout = ops.sequence.SlotLayout.cached(tuple(lengths), int(self.row_slots), padded.device)
        F = padded.shape[-1]
        assert F == self.F, f'self.F = {self.F} != F = {F}'
        x = ops.sequence.log1p(self.dropout_input(layout.scatter_rows(padded)))   # log1p(0) = 0: idle rows stay zero
        T, S = layout.T, layout.slots
        h = ops.packed_lstm(self.blstm, PackedSequence(x, torch.full((T,), S, dtype=torch.int64)), meta=layout.meta).data
        h = self._dense(self.dropout_linear(h))
        masks = layout.gather_rows(h.view(-1, self.K, self.F), padded.shape[1])   # 'tb (k f) -> tb k f', back to one example per row
        return PaddedList(masks, lengths, True, lengths_dev)

    def _forward_static_slots(self, Y_abs, slots):
        """``forward`` on a row-slot layout of fixed capacity whose length pattern is device data (``ops.sequence.StaticSlots``, carried
        by the batch as ``batch['slots']``): the same network and results as :meth:`_forward_row_slots`, but no launch depends on the
        examples' lengths - what a captured optimizer step needs to serve ragged batches (``train.graphed``)."""
        padded = Y_abs.padded if isinstance(Y_abs, PaddedList) and Y_abs.intact() else as_padded(Y_abs)[0]
        why = ops.lstm.unsupported_reason(self.blstm, padded)
        assert why is None and self.hip_blstm, f'StaticSlots batches run on the HIP recurrence only ({why})'
        assert padded.shape[-1] == self.F, f'self.F = {self.F} != F = {padded.shape[-1]}'
        x = ops.sequence.log1p(self.dropout_input(slots.scatter_rows(padded)))    # log1p(0) = 0: idle rows stay zero
        T, S = slots.steps, slots.slots
        h = ops.packed_lstm(self.blstm, PackedSequence(x, torch.full((T,), S, dtype=torch.int64)), meta=slots.meta).data
        h = self._dense(self.dropout_linear(h))
        masks = slots.gather_rows(h.view(-1, self.K, self.F))                     # [B, padded_time, K, F], padding frames zero
        return PaddedList(masks, [slots.padded_time] * slots.examples, True, slots.frames)

    @torch.no_grad()
    def separate(self, y, num_samples=None, stft=None):
        """Mixture waveforms -> separated waveforms, entirely on the device.

        The evaluation path of ``pit/evaluate.py:149-163`` (``model(batch)``, ``Z = mask * Y[:, None, :]``,
        ``paderbox.istft(Z, 512, 128)``, cut to the signal length) without leaving the GPU: HIP STFT,
        mask estimator, complex masking, HIP iSTFT.

        Args:
            y: list of ``(N_b,)`` tensors (descending length) or a padded ``[B, N]`` tensor
            num_samples: lengths when ``y`` is padded and ragged
        Returns: list of ``(K, N_b)`` tensors.
        """
        from padertorch_amd.ops.features import _pad_rows
        if stft is None:
            stft = ops.STFT(512, 128)
        if isinstance(y, (list, tuple)):
            num_samples = [int(t.shape[-1]) for t in y]
            y = _pad_rows(list(y), max(num_samples))
        if num_samples is None:
            num_samples = [y.shape[-1]] * y.shape[0]
        # ONE transform of the mixture: the complex spectrum (frames past a row's own count are zero), its magnitude as the
        # model input (round 2 ran the fused feature kernel for |Y| and the STFT again for Y)
        ns = torch.tensor(num_samples, dtype=torch.int32, device=y.device)
        Y = stft(y, num_samples=ns)                                       # [B, T, F] complex
        frames = [int(stft.samples_to_frames(n)) for n in num_samples]
        from padertorch_amd.ops.sequence.pack_module import PaddedList
        masks = self.forward(dict(Y_abs=PaddedList(Y.abs(), frames, True)))   # PaddedList, [T, B, K, F]
        m = masks.padded if not masks.batch_first else masks.padded.transpose(0, 1)
        Z = m.permute(1, 2, 0, 3) * Y[:, None, :, :]                      # 't b k f -> b k t f' times Y
        z = stft.inverse(Z.contiguous())                                  # [B, K, samples]
        return [z[b, :, :n] for b, n in enumerate(num_samples)]

    def review(self, batch, model_out):
        batch = self.prepare_batch(batch)
        if isinstance(model_out, PaddedList) and model_out.intact():
            mask, mask_bf = model_out.padded, model_out.batch_first
            lengths_dev = model_out.lengths_dev
        else:
            mask, _, lengths_dev = as_padded(model_out, batch_first=True)
            mask_bf = True
        Y, _, _ = as_padded(batch['Y_abs'])
        X, _, _ = as_padded(batch['X_abs'])
        C, _, _ = as_padded(batch['cos_phase_difference'])
        # MSE loss and ideal-phase-sensitive loss of every example, batch means (reference :117-140)
        loss, _, _ = ops.losses.pit_mse_ips_losses(
            mask, Y, X, C, lengths_dev, mask_batch_first=mask_bf)
        # (ops.scalars.pick: loss[i] whose backward costs no launch)
        review = dict(losses={'pit_mse_loss': ops.scalars.pick(loss, 0), 'pit_ips_loss': ops.scalars.pick(loss, 1)})

        if self.create_snapshot:
            # tensorboard images of the batch's first example (reference :141-150; note its quirk: every 'estimation_<k>' shows
            # source 0 of the targets)
            first = 0
            masks = model_out[first]
            review['images'] = {
                'observation': stft_to_image(batch['Y_abs'][first]),
                **{f'mask_{k}': mask_to_image(masks[:, k, :]) for k in range(masks.shape[1])},
                **{f'estimation_{k}': stft_to_image(batch['X_abs'][first][:, 0, :]) for k in range(masks.shape[1])},
            }
        return review
