"""``MaskEstimator`` of ``padertorch/contrib/jensheit/mask_estimator_example/modul.py:45-158``.

Multi-channel observations ``[C, T_b, F]`` per example -> masked per-utterance normalisation -> (B)LSTM ->
``fully_connected_stack`` -> speech / noise masks (+ optional VAD head), every channel treated as an own sequence.
Same constructor arguments, output keys (``MaskKeys``) and ``state_dict`` layout (``fully_connected.linear_<i>.*``,
``recurrent.lstm.*``, ``normalization.*``, ``linear_vad.*``) as the reference, so its checkpoints load.  The compute runs
on this package's HIP path: ``modules.Normalization`` (``csrc/norm.hip``), ``modules.StatefulLSTM`` on a PackedSequence
(``csrc/lstm_split.hip``), the dense layers on the planes GEMM (``modules.fully_connected.PlanesLinear``).

Two details of the reference that are kept because results depend on them:
* the channels of all examples are flattened example-major (``[b0c0, b0c1, ..., b1c0, ...]``) but the result is folded
  back with ``'(c b) t f -> b c t f'`` (``modul.py:131``), i.e. channel-major; for ``C > 1`` and ``B > 1`` the two orders
  differ, and the reference's is reproduced;
* ``input_dropout`` is constructed but never applied in ``forward`` (``modul.py:107,116-133``).

``finalize_dogmatic_config`` builds the reference's default sub-configs (``:47-83``) with this package's factories; the
``Configurable`` machinery itself is out of scope: ``MaskEstimator.from_defaults(num_features=...)`` is
the stand-alone equivalent of ``MaskEstimator.from_config(MaskEstimator.get_config({...}))``.
"""
import torch
from torch.nn.utils.rnn import PackedSequence

from .... import base
from ....modules.fully_connected import fully_connected_stack
from ....modules.normalization import Normalization
from ....modules.recurrent import StatefulLSTM
from ....ops.mappings import ACTIVATION_FN_MAP
from ....ops.sequence import pack_sequence, pad_sequence, unpack_sequence, unpad_sequence

__all__ = ['MaskKeys', 'MaskEstimator']


class MaskKeys:
    OBSERVATION = 'observation'
    SPEECH_IMAGE = 'speech_image'
    SPEECH_MASK_PRED = 'speech_mask_prediction'
    SPEECH_TARGET = 'speech_target'
    NOISE_MASK_PRED = 'noise_mask_prediction'
    SPEECH_MASK_LOGITS = 'speech_mask_logits'
    NOISE_MASK_LOGITS = 'noise_mask_logits'
    SPEECH_MASK_TARGET = 'speech_mask_target'
    NOISE_MASK_TARGET = 'noise_mask_target'
    OBSERVATION_STFT = 'observation_stft'
    OBSERVATION_ABS = 'observation_abs'
    MASK_ESTIMATOR_STATE = 'mask_estimator_state'
    SPEECH_PRED = 'speech_prediction'
    NUM_FRAMES = 'num_frames'
    NUM_SAMPLES = 'num_samples'
    SPEECH_SOURCE = 'speech_source'
    VAD = 'vad'
    VAD_LOGITS = 'vad_logits'


_K = MaskKeys


class MaskEstimator(base.Module):
    @classmethod
    def finalize_dogmatic_config(cls, config):
        """Fill ``config`` (a dict with at least ``num_features``) with the reference's default sub-configurations."""
        F = config['num_features']
        rec = config.setdefault('recurrent', {})
        for k, v in dict(factory=StatefulLSTM, input_size=F, hidden_size=256, bidirectional=True, batch_first=False).items():
            rec.setdefault(k, v)
        width = rec['hidden_size'] * (2 if rec['bidirectional'] else 1)
        fc = config.setdefault('fully_connected', {})
        for k, v in dict(factory=fully_connected_stack, input_size=width, hidden_size=[1024] * 3, output_size=2 * F).items():
            fc.setdefault(k, v)
        assert rec['input_size'] == F, (rec['input_size'], F)
        assert fc['output_size'] == 2 * F, (fc['output_size'], F)
        if 'normalization' not in config or config['normalization'] is not None:
            norm = config.setdefault('normalization', {})
            for k, v in dict(factory=Normalization, data_format='tbf', shape=(1, 1, 1, F), statistics_axis='t',
                             independent_axis='f', batch_axis='b', sequence_axis='t').items():
                norm.setdefault(k, v)
            assert norm['shape'][-1] % F == 0, (norm['shape'], F)
        return config

    @classmethod
    def from_defaults(cls, num_features=513, **updates):
        """The module the reference's ``get_config`` / ``from_config`` pair builds for ``{'num_features': ..., **updates}``
        (sub-dicts of ``updates`` override single entries of the default sub-configurations)."""
        config = cls.finalize_dogmatic_config(dict(num_features=num_features, **updates))
        return base._instantiate(dict(factory=cls, **config))

    def __init__(self, fully_connected, recurrent: StatefulLSTM, normalization: Normalization, num_features: int = 513,
                 input_dropout: float = 0.5, use_log: bool = False, use_powerspectrum: bool = False,
                 separate_masks: bool = True, output_activation: str = 'sigmoid', reuse_states: bool = False,
                 vad: bool = False):
        super().__init__()
        if use_log or use_powerspectrum:
            raise NotImplementedError       # (the reference raises here as well)
        self.fully_connected = fully_connected
        self.normalization = normalization
        self.recurrent = recurrent
        self.num_features = num_features
        self.input_dropout = torch.nn.Dropout(input_dropout)
        self.use_log = use_log
        self.use_powerspectrum = use_powerspectrum
        self.separate_masks = separate_masks
        self.output_activation = output_activation
        self.vad = vad
        if vad:
            self.linear_vad = torch.nn.Linear(2 * num_features, 1)
        self.reuse_states = reuse_states

    def forward(self, x, num_frames):
        """``x``: list (length B, frames descending) of ``[C, T_b, F]`` magnitudes; ``num_frames``: list of the ``T_b``.
        Returns ``{key: [B, C, T, F]}`` (``T`` = longest example; the model wrapper cuts every example to its frames)."""
        C = x[0].shape[0]
        frames = [int(n) for n in num_frames for _ in range(C)]
        seqs = [channel for example in x for channel in example]            # B * C sequences [T_b, F]
        if self.normalization:
            h = pad_sequence(seqs, batch_first=False)
            h = self.normalization(h, torch.tensor(frames, dtype=torch.float32))
            seqs = unpad_sequence(h, frames)
        packed = pack_sequence(seqs)
        if not self.reuse_states:
            del self.recurrent.states
        packed = self.recurrent(packed)
        packed = PackedSequence(self.fully_connected(packed.data), packed.batch_sizes)
        out = unpack_sequence(packed).padded.transpose(0, 1)                # zero-padded [B * C, T, 2 F] (one scatter on the device)
        n = out.shape[0] // C
        out = out.reshape(C, n, *out.shape[1:]).transpose(0, 1)               # '(c b) t f -> b c t f': the reference's fold
        act = ACTIVATION_FN_MAP[self.output_activation]
        F = self.num_features
        logits = out[..., :F]
        result = {_K.SPEECH_MASK_PRED: act()(logits), _K.SPEECH_MASK_LOGITS: logits}
        if self.separate_masks:
            noise = out[..., F:]
            result.update({_K.NOISE_MASK_PRED: act()(noise), _K.NOISE_MASK_LOGITS: noise})
        if self.vad:
            vad_logits = self.linear_vad(out)
            result.update({_K.VAD_LOGITS: vad_logits, _K.VAD: act()(vad_logits)})
        return result
