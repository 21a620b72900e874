"""``SimpleMaskEstimator``: masked per-utterance normalisation -> BLSTM -> three dense layers -> sigmoid, trained
with binary cross-entropy on a speech and a noise mask.

Drop-in for ``padertorch/contrib/examples/speech_enhancement/mask_estimator/model.py:6-90``: the same
constructor arguments, the same ``forward(batch) -> dict`` / ``review(batch, output) -> dict`` contract and the
same ``state_dict`` (the layers sit in ``self.net`` at the reference's positions: ``net.0`` normalisation,
``net.1`` LSTM, ``net.3`` / ``net.6`` / ``net.8`` linears).  Differences: the normalisation and the LSTM run on the
HIP kernels (``modules.Normalization``, ``modules.StatefulLSTM``), and the summary images - a device -> host copy
each - are rendered only when ``create_snapshot`` is set (``base.py:300-306`` allows that) instead of on every
review.
"""
import torch
from torch.nn import functional as F

from ..... import base, modules
from .....ops import mappings
from .....summary import mask_to_image, stft_to_image

#: review keys of the images and where their data comes from: (dict, key, renderer)
_IMAGES = (
    ('speech_mask', 'output', 'speech_mask_prediction', mask_to_image),
    ('observed_stft', 'batch', 'observation_abs', stft_to_image),
    ('noise_mask', 'output', 'noise_mask_prediction', mask_to_image),
)


def _layers(num_features, num_units, dropout, activation):
    act = mappings.ACTIVATION_FN_MAP[activation]
    hidden = num_units // 4                 # per direction
    yield modules.Normalization('btf', (1, 1, num_features), statistics_axis='t', independent_axis='f',
                                batch_axis='b', sequence_axis='t')
    yield modules.StatefulLSTM(num_features, hidden, bidirectional=True, batch_first=True, save_states=False)
    for width_in in (2 * hidden, num_units):
        yield torch.nn.Dropout(dropout)
        yield torch.nn.Linear(width_in, num_units)
        yield act()
    yield torch.nn.Linear(num_units, 2 * num_features)      # speech mask | noise mask
    yield torch.nn.Sigmoid()


class SimpleMaskEstimator(base.Model):
    def __init__(self, num_features, num_units=1024, dropout=0.5, activation='elu'):
        super().__init__()
        self.num_features = num_features
        self.net = torch.nn.Sequential(*_layers(num_features, num_units, dropout, activation))

    def forward(self, batch):
        speech, noise = self.net(batch['observation_abs']).split(self.num_features, dim=-1)
        return dict(speech_mask_prediction=speech, noise_mask_prediction=noise)

    def review(self, batch, output):
        loss = sum(F.binary_cross_entropy(output[f'{k}_mask_prediction'], batch[f'{k}_mask_target'])
                   for k in ('noise', 'speech'))
        review = dict(loss=loss)
        if self.create_snapshot:
            review['images'] = self.add_images(batch, output)
        return review

    @staticmethod
    def add_images(batch, output):
        """First example of the (batch-first) tensors as tensorboard images (reference ``:74-90``); the target masks
        are added under the reference's condition (``'speech_mask_prediction' in batch``)."""
        src = dict(batch=batch or {}, output=output)
        images = {name: render(src[where][key], True) for name, where, key, render in _IMAGES if key in src[where]}
        if batch is not None and 'speech_mask_prediction' in batch:
            images['speech_mask_target'] = mask_to_image(batch['speech_mask_target'], True)
            if 'speech_mask_target' in batch:
                images['noise_mask_target'] = mask_to_image(batch['noise_mask_target'], True)
        return images
