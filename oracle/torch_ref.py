"""torch-CPU (fp32) port of the reference hot path.  ORACLE - test infrastructure only.

This is the reference *algorithm* (dense conv1d STFT, ``torch.nn.LSTM`` on a PackedSequence,
python-loop O(K!) ``pit_loss``, Adam + global-norm clipping) restated so it can travel to the
GPU box where ``/root/reference`` does not exist.  It serves (i) as the fp32 checker of the HIP
model path and (ii) as ``bench.py``'s ``cpu_baseline`` (kind "port").

Follows
  * ``padertorch/ops/_stft.py:11-23,103-174``                          -> :class:`ConvSTFT`
  * ``padertorch/ops/losses/source_separation.py:34-124, 13-31``       -> :func:`pit_loss`, :func:`deep_clustering_loss`
  * ``padertorch/contrib/examples/source_separation/pit/model.py:27-151`` -> :class:`PITModelRef`
  * ``padertorch/contrib/tcl/dc.py:8-84``                              -> :class:`DCModelRef`
  * ``padertorch/train/trainer.py:541-551,608-620,512-532`` and
    ``padertorch/train/optimizer.py:31-42,71-90``                       -> :func:`train_step`
Checked against the real reference import by ``tests/golden/make_golden.py`` (fixture G6).
"""
import itertools

import numpy as np
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import PackedSequence, pack_sequence, pad_packed_sequence

from . import stft_np


class ConvSTFT:
    """The reference's forward STFT: dense ``[2F,1,L]`` DFT*window kernel + strided conv1d."""

    def __init__(self, size=512, shift=128, window='blackman', window_length=None,
                 fading='full', pad=True):
        self.size, self.shift = size, shift
        self.L = size if window_length is None else window_length
        self.fading, self.pad = fading, pad
        w = stft_np.get_window(window, False, self.L)
        k = np.arange(self.L)
        n = np.arange(size // 2 + 1)[:, None]
        kern = np.concatenate([np.cos(-2 * np.pi * n * k / size) * w,
                               np.sin(-2 * np.pi * n * k / size) * w], axis=0)
        self.kernel = torch.from_numpy(kern).unsqueeze(1)      # float64, like _stft.py:23

    def __call__(self, x):
        org = x.shape
        x = x.reshape(-1, org[-1])
        left, right = stft_np.fading_pad_width(self.L, self.shift, self.fading)
        if left or right:
            x = F.pad(x, (left, right))
        if self.pad:
            T = x.shape[-1]
            if T < self.L:
                x = F.pad(x, (0, self.L - T))
            elif self.shift != 1 and (T + self.shift - self.L) % self.shift != 0:
                x = F.pad(x, (0, self.shift - ((T + self.shift - self.L) % self.shift)))
        enc = F.conv1d(x.unsqueeze(1), self.kernel.to(x), stride=self.shift)   # [B, 2F, frames]
        enc = enc.reshape(*org[:-1], *enc.shape[-2:]).transpose(-1, -2)
        re, im = torch.chunk(enc, 2, dim=-1)
        return torch.complex(re.contiguous(), im.contiguous())


def features_from_waveforms(stft, s_list, y_list):
    """pit/data.py:52-75 on torch tensors: lists of (K,N_b) / (N_b,) -> dict of lists."""
    out = dict(Y_abs=[], X_abs=[], cos_phase_difference=[], num_frames=[])
    for s, y in zip(s_list, y_list):
        S = stft(s)                       # (K, T, F)
        Y = stft(y)                       # (T, F)
        X = S.permute(1, 0, 2)
        out['Y_abs'].append(Y.abs())
        out['X_abs'].append(X.abs().contiguous())
        out['cos_phase_difference'].append(torch.cos(torch.angle(Y)[:, None, :] - torch.angle(X)))
        out['num_frames'].append(Y.shape[0])
    return out


def pit_loss(estimate, target, axis, loss_fn=F.mse_loss, return_permutation=False):
    """source_separation.py:94-124 (brute force, first minimum wins)."""
    sources = estimate.size()[axis]
    assert sources < 30
    assert estimate.size() == target.size()
    perms = list(itertools.permutations(range(sources)))
    cands = []
    indexer = [slice(None)] * estimate.ndim
    for p in perms:
        indexer[axis] = p
        cands.append(loss_fn(estimate[tuple(indexer)], target))
    min_loss, idx = torch.min(torch.stack(cands), dim=0)
    if return_permutation:
        return min_loss, perms[int(idx)]
    return min_loss


def deep_clustering_loss(x, t):
    """source_separation.py:26-31."""
    N = x.size()[0]
    return (torch.sum((x.t() @ x) ** 2) - 2 * torch.sum((x.t() @ t) ** 2)
            + torch.sum((t.t() @ t) ** 2)) / N ** 2


def unpack_sequence(packed):
    padded, lengths = pad_packed_sequence(packed)
    return [padded[:l, b] for b, l in enumerate(lengths)]


class PITModelRef(torch.nn.Module):
    """pit/model.py:27-151 with identical parameter names/shapes."""

    def __init__(self, F=257, recurrent_layers=3, units=600, K=2, dropout_input=0.,
                 dropout_hidden=0., dropout_linear=0., output_activation='relu'):
        super().__init__()
        self.K, self.F = K, F
        self.dropout_input = torch.nn.Dropout(dropout_input)
        self.blstm = torch.nn.LSTM(F, units, recurrent_layers, bidirectional=True,
                                   dropout=dropout_hidden)
        self.dropout_linear = torch.nn.Dropout(dropout_linear)
        self.relu = torch.nn.ReLU()
        self.linear1 = torch.nn.Linear(2 * units, 2 * units)
        self.linear2 = torch.nn.Linear(2 * units, F * K)
        acts = dict(relu=torch.nn.ReLU, sigmoid=torch.nn.Sigmoid, identity=torch.nn.Identity,
                    tanh=torch.nn.Tanh, elu=torch.nn.ELU, leaky_relu=torch.nn.LeakyReLU)
        self.output_activation = acts[output_activation]()

    def forward(self, batch):
        h = pack_sequence(batch['Y_abs'])
        data = torch.log1p(self.dropout_input(h.data))
        h, _ = self.blstm(PackedSequence(data, h.batch_sizes))
        d = self.linear2(self.relu(self.linear1(self.dropout_linear(h.data))))
        d = self.output_activation(d)
        d = d.reshape(d.shape[0], self.K, self.F)            # 'tb (k f) -> tb k f'
        return unpack_sequence(PackedSequence(d, h.batch_sizes))

    def review(self, batch, model_out):
        mse, ips = [], []
        for mask, obs, tgt, cpd in zip(model_out, batch['Y_abs'], batch['X_abs'],
                                       batch['cos_phase_difference']):
            est = mask * obs[:, None, :]
            mse.append(pit_loss(est, tgt, axis=-2))
            ips.append(pit_loss(est, tgt * cpd, axis=-2))
        return dict(losses=dict(pit_mse_loss=torch.mean(torch.stack(mse)),
                                pit_ips_loss=torch.mean(torch.stack(ips))))


class DCModelRef(torch.nn.Module):
    """contrib/tcl/dc.py:8-84."""

    def __init__(self, F=257, recurrent_layers=2, units=600, E=20,
                 input_feature_transform='identity'):
        super().__init__()
        self.E, self.F = E, F
        self.input_feature_transform = input_feature_transform
        self.blstm = torch.nn.LSTM(F, units, recurrent_layers, bidirectional=True)
        self.linear = torch.nn.Linear(2 * units, F * E)

    def forward(self, batch):
        h = pack_sequence(batch['Y_abs'])
        data = h.data
        if self.input_feature_transform == 'log1p':
            data = torch.log1p(data)
        elif self.input_feature_transform == 'log':
            data = torch.log(data + 1e-10)
        elif self.input_feature_transform != 'identity':
            raise NotImplementedError(self.input_feature_transform)
        h, _ = self.blstm(PackedSequence(data, h.batch_sizes))
        d = self.linear(h.data).reshape(-1, self.E, self.F)  # 'tb (e f) -> tb e f'
        d = F.normalize(d, dim=-2)
        return unpack_sequence(PackedSequence(d, h.batch_sizes))

    def review(self, batch, model_out):
        losses = []
        for emb, tm in zip(model_out, batch['target_mask']):
            x = emb.permute(0, 2, 1).reshape(-1, emb.shape[1])       # 't e f -> (t f) e'
            t = tm.permute(0, 2, 1).reshape(-1, tm.shape[1])         # 't k f -> (t f) k'
            losses.append(deep_clustering_loss(x, t))
        return dict(losses=dict(dc_loss=torch.mean(torch.stack(losses))))


def review_to_loss(review, loss_weights):
    """trainer.py:608-620: sum of weight*loss over non-zero weights."""
    loss = 0.
    for k, v in review['losses'].items():
        w = 1. if loss_weights is None else loss_weights[k]
        if w != 0:
            loss = loss + w * v
    return loss


def train_step(model, optimizer, batches, loss_weights=None, gradient_clipping=1.):
    """One optimizer step over ``len(batches)`` virtual-minibatch examples.

    trainer.py:357-393 (accumulate, NOT average), :512-532 (clip -> step -> zero_grad),
    optimizer.py:31-42 (``clip_grad_norm_``).
    Returns (list of losses, grad_norm).
    """
    losses = []
    for batch in batches:
        out = model(batch)
        loss = review_to_loss(model.review(batch, out), loss_weights)
        loss.backward()
        losses.append(float(loss))
    gn = torch.nn.utils.clip_grad_norm_(list(model.parameters()), gradient_clipping)
    optimizer.step()
    optimizer.zero_grad()
    return losses, float(gn)
