"""Pin oracle/torch_ref.py (the torch-CPU port used as fp32 checker + cpu_baseline) against G6."""
import numpy as np
import torch

from oracle import stft_np, torch_ref


def _batch(g6):
    Ts = [int(t) for t in g6['Ts']]
    b = {k: [torch.from_numpy(g6[f'in_{k}_{i}']) for i in range(len(Ts))]
         for k in ['Y_abs', 'X_abs', 'cos_phase_difference', 'target_mask']}
    b['num_frames'] = Ts
    return b


def _load(model, g6, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(v) for k, v in g6.items()
          if isinstance(v, np.ndarray) and k.startswith(prefix)}
    model.load_state_dict(sd, strict=True)          # identical keys/shapes (appendix B.5)
    return model


def test_pit_model_vs_reference(g6):
    model = _load(torch_ref.PITModelRef(F=9, recurrent_layers=2, units=4, K=2), g6, 'pit_sd_')
    batch = _batch(g6)
    masks = model(batch)
    for b, m in enumerate(masks):
        np.testing.assert_allclose(m.detach().numpy(), g6[f'pit_mask_{b}'], atol=1e-6)
    rv = model.review(batch, masks)
    np.testing.assert_allclose(rv['losses']['pit_mse_loss'].item(), g6['pit_mse_loss'], atol=1e-6)
    np.testing.assert_allclose(rv['losses']['pit_ips_loss'].item(), g6['pit_ips_loss'], atol=1e-6)
    # minibatch loss == mean of single-example losses (tests/test_models/test_bss.py:153-192)
    np.testing.assert_allclose(g6['pit_single_losses'].mean(0),
                               [g6['pit_mse_loss'], g6['pit_ips_loss']], atol=1e-6)
    rv['losses']['pit_ips_loss'].backward()
    for n, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), g6[f'pit_grad_{n}'], atol=1e-6, err_msg=n)


def test_pit_three_trainer_steps_vs_reference(g6):
    """Adam(clip=1), virtual_minibatch_size=2, accumulate-not-average: trainer.py:357-393,512-532."""
    model = _load(torch_ref.PITModelRef(F=9, recurrent_layers=2, units=4, K=2), g6, 'pit_sd_')
    batch = _batch(g6)
    exs = [{k: [v[b] for b in idx] for k, v in batch.items()} for idx in g6['train_example_indices']]
    opt = torch.optim.Adam(model.parameters())
    for i in range(3):
        torch_ref.train_step(model, opt, exs[2 * i:2 * i + 2],
                             loss_weights=dict(pit_ips_loss=1., pit_mse_loss=0.), gradient_clipping=1.)
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g6['pit_sd3_' + k], atol=1e-6, err_msg=k)


def test_dc_model_vs_reference(g6):
    model = _load(torch_ref.DCModelRef(F=9, recurrent_layers=1, units=4, E=3), g6, 'dc_sd_')
    batch = _batch(g6)
    emb = model(batch)
    for b, m in enumerate(emb):
        np.testing.assert_allclose(m.detach().numpy(), g6[f'dc_emb_{b}'], atol=1e-6)
    rv = model.review(batch, emb)
    np.testing.assert_allclose(rv['losses']['dc_loss'].item(), g6['dc_loss'], atol=1e-6)
    rv['losses']['dc_loss'].backward()
    for n, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), g6[f'dc_grad_{n}'], atol=1e-6, err_msg=n)


def test_default_model_param_count():
    assert sum(p.numel() for p in torch_ref.PITModelRef().parameters()) == 23_480_914
    assert sum(p.numel() for p in torch_ref.DCModelRef().parameters()) == 18_945_940


def test_conv_stft_matches_rfft_oracle(g3):
    st = torch_ref.ConvSTFT(512, 128)
    Y = st(torch.from_numpy(g3['y']).double()).numpy()
    np.testing.assert_allclose(Y, stft_np.stft(g3['y'], 512, 128), atol=1e-9)
    f = torch_ref.features_from_waveforms(st, [torch.from_numpy(g3['s'])], [torch.from_numpy(g3['y'])])
    np.testing.assert_allclose(f['Y_abs'][0].numpy(), g3['Y_abs'], atol=2e-5)
    np.testing.assert_allclose(f['X_abs'][0].numpy(), g3['X_abs'], atol=2e-5)
    assert f['num_frames'] == [15]
