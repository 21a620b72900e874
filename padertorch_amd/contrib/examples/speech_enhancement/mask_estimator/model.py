"""``SimpleMaskEstimator`` (reference:
``padertorch/contrib/examples/speech_enhancement/mask_estimator/model.py:6-90``) on the HIP path:
masked per-utterance ``Normalization`` -> BLSTM (``StatefulLSTM``) -> 3 linear layers -> sigmoid,
binary cross-entropy on the speech and noise masks.  Same module tree and ``state_dict`` keys
(``net.0.gamma``, ``net.1.lstm.weight_ih_l0``, ``net.3.weight`` ...).
"""
import torch

from ..... import base, modules
from .....ops import mappings
from .....summary import mask_to_image, stft_to_image


class SimpleMaskEstimator(base.Model):
    def __init__(self, num_features, num_units=1024, dropout=0.5, activation='elu'):
        super().__init__()
        self.num_features = num_features
        self.net = torch.nn.Sequential(
            modules.Normalization('btf', (1, 1, num_features), statistics_axis='t', independent_axis='f',
                                  batch_axis='b', sequence_axis='t'),
            modules.StatefulLSTM(num_features, num_units // 4, bidirectional=True, batch_first=True,
                                 save_states=False),
            torch.nn.Dropout(dropout),
            torch.nn.Linear((num_units // 4) * 2, num_units),
            mappings.ACTIVATION_FN_MAP[activation](),
            torch.nn.Dropout(dropout),
            torch.nn.Linear(num_units, num_units),
            mappings.ACTIVATION_FN_MAP[activation](),
            # twice num_features for speech and noise_mask
            torch.nn.Linear(num_units, 2 * num_features),
            # Output activation to force outputs between 0 and 1
            torch.nn.Sigmoid()
        )

    def forward(self, batch):
        x = batch['observation_abs']
        out = self.net(x)
        return dict(
            speech_mask_prediction=out[..., :self.num_features],
            noise_mask_prediction=out[..., self.num_features:],
        )

    def review(self, batch, output):
        noise_mask_loss = torch.nn.functional.binary_cross_entropy(
            output['noise_mask_prediction'], batch['noise_mask_target'])
        speech_mask_loss = torch.nn.functional.binary_cross_entropy(
            output['speech_mask_prediction'], batch['speech_mask_target'])
        return dict(loss=noise_mask_loss + speech_mask_loss, images=self.add_images(batch, output))

    @staticmethod
    def add_images(batch, output):
        speech_mask = output['speech_mask_prediction']
        observation = batch['observation_abs']
        images = dict()
        images['speech_mask'] = mask_to_image(speech_mask, True)
        images['observed_stft'] = stft_to_image(observation, True)
        if 'noise_mask_prediction' in output:
            images['noise_mask'] = mask_to_image(output['noise_mask_prediction'], True)
        if batch is not None and 'speech_mask_prediction' in batch:
            images['speech_mask_target'] = mask_to_image(batch['speech_mask_target'], True)
            if 'speech_mask_target' in batch:
                images['noise_mask_target'] = mask_to_image(batch['noise_mask_target'], True)
        return images
