"""The model classes drop into the REFERENCE ``padertorch.Trainer`` unchanged.

Part 1 (always): the surface the reference Trainer and its hooks touch on a model
(``padertorch/train/trainer.py:100-104,541-566``, ``train/hooks.py:380-392,538-550``): a ``torch.nn.Module`` with
``example_to_device(example, device)``, ``__call__(example)``, ``review(example, out)``, ``modify_summary`` and a
settable ``create_snapshot`` flag; the reference's ``state_dict`` keys.

Part 2 (build container only, skipped where /root/reference does not exist): the real reference Trainer - imported
with the stand-ins of tests/golden/ref_shim for its absent third-party dependencies - trains OUR model class for
three iterations on the CPU.  The product has no CPU path, so the one HIP-only op of the review
(``ops.losses.pit_mse_ips_losses``) is replaced by the oracle's loop for the duration of the test; everything else
(model class, forward, review, PaddedList outputs, example_to_device, summaries, checkpoints) is the shipped code
driven by the reference's loop."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import padertorch_amd as pta
from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
from padertorch_amd.contrib.examples.speech_enhancement.mask_estimator.model import SimpleMaskEstimator
from padertorch_amd.contrib.tcl.dc import DeepClusteringModel

REPO = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize('cls,kw', [(PermutationInvariantTrainingModel, dict(F=9, recurrent_layers=2, units=4, K=2)),
                                    (DeepClusteringModel, dict(F=9, recurrent_layers=2, units=4)),
                                    (SimpleMaskEstimator, dict(num_features=9, num_units=16))])
def test_model_surface_the_reference_trainer_touches(cls, kw):
    m = cls(**kw)
    assert isinstance(m, torch.nn.Module)                                   # trainer.py:100-104
    for name in ('example_to_device', 'review', 'modify_summary', 'forward'):
        assert callable(getattr(m, name)), name
    assert m.create_snapshot is False                                       # hooks.py:387 sets it, :392 resets it
    m.create_snapshot = True
    assert m.create_snapshot is True
    ex = dict(a=np.arange(3, dtype=np.float32), b=[np.ones(2, dtype=np.float32)], c='text')
    moved = m.example_to_device(ex, 'cpu')                                  # trainer.py:545
    assert torch.is_tensor(moved['a']) and torch.is_tensor(moved['b'][0]) and moved['c'] == 'text'
    summary = dict(scalars=dict(loss=[1., 3.]), histograms={}, images={})
    assert m.modify_summary(summary)['scalars']['loss'] == 2.              # hooks.py: mean of the collected scalars


def test_state_dict_keys_are_the_reference_keys(g6):
    pit = PermutationInvariantTrainingModel(F=9, recurrent_layers=2, units=4, K=2)
    ref_keys = {k[len('pit_sd_'):] for k in g6 if k.startswith('pit_sd_')}
    assert set(pit.state_dict()) == ref_keys
    dc = DeepClusteringModel(F=9, recurrent_layers=1, units=4)               # the golden's DC model has one layer
    ref_keys = {k[len('dc_sd_'):] for k in g6 if k.startswith('dc_sd_')}
    assert set(dc.state_dict()) == ref_keys


def _oracle_pit_losses(mask, observation, target, cos_phase_difference=None, lengths=None, *, mask_batch_first=True,
                       data_batch_first=True):
    """CPU stand-in with the interface of ops.losses.pit_mse_ips_losses: the reference's loop (pit/model.py:117-140)
    through the oracle's pit_loss."""
    from oracle import torch_ref
    m = mask if mask_batch_first else mask.transpose(0, 1)
    B = m.shape[0]
    lens = [int(n) for n in lengths] if lengths is not None else [m.shape[1]] * B
    mse, ips = [], []
    for b in range(B):
        T = lens[b]
        est = m[b, :T] * observation[b, :T, None, :]
        mse.append(torch_ref.pit_loss(est, target[b, :T], axis=-2))
        ips.append(torch_ref.pit_loss(est, target[b, :T] * cos_phase_difference[b, :T], axis=-2))
    return torch.stack([torch.stack(mse).mean(), torch.stack(ips).mean()]), None, None


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='needs the reference tree (build container)')
def test_reference_trainer_trains_our_model(g6, tmp_path, monkeypatch):
    monkeypatch.syspath_prepend('/root/reference')                          # stays for the lazy imports of the loop
    monkeypatch.syspath_prepend(str(REPO / 'tests' / 'golden' / 'ref_shim'))
    try:
        import padertorch as pt                      # the REFERENCE
    except Exception as e:                           # pragma: no cover
        pytest.skip(f'reference not importable: {e!r}')
    from padertorch_amd.ops import losses
    monkeypatch.setattr(losses, 'pit_mse_ips_losses', _oracle_pit_losses)
    monkeypatch.setattr(pta.ops.losses, 'pit_mse_ips_losses', _oracle_pit_losses, raising=False)

    model = PermutationInvariantTrainingModel(F=9, recurrent_layers=2, units=4, K=2)
    model.load_state_dict({k[len('pit_sd_'):]: torch.from_numpy(v) for k, v in g6.items()
                           if isinstance(v, np.ndarray) and k.startswith('pit_sd_')})
    Ts = [int(t) for t in g6['Ts']]
    batch = {k: [g6[f'in_{k}_{i}'] for i in range(len(Ts))] for k in ['Y_abs', 'X_abs', 'cos_phase_difference']}
    batch['num_frames'] = Ts
    examples = [{k: [v[b] for b in idx] for k, v in batch.items()} for idx in g6['train_example_indices']]

    trainer = pt.Trainer(model, str(tmp_path), pt.optimizer.Adam(gradient_clipping=1.),
                         loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.), summary_trigger=(1, 'iteration'),
                         checkpoint_trigger=(1000, 'iteration'), stop_trigger=(3, 'iteration'), virtual_minibatch_size=2)
    trainer.train(examples, device='cpu')
    assert trainer.iteration == 3
    # the reference loop over OUR model lands on the parameters the reference loop over ITS OWN model produced
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g6['pit_sd3_' + k], atol=1e-6, err_msg=k)
    assert (tmp_path / 'checkpoints' / 'ckpt_latest.pth').exists()
