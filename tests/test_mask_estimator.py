"""contrib/jensheit MaskEstimator (reference ``contrib/jensheit/mask_estimator_example/modul.py:45-158``)
and ``fully_connected_stack`` (``modules/fully_connected.py:9-66``) against golden g11 (the reference's own module built from its
``finalize_dogmatic_config`` defaults, tests/golden/make_golden.py::g11)."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

GOLDEN = Path(__file__).parent / 'golden'


@pytest.fixture(scope='module')
def g11():
    return np.load(GOLDEN / 'g11_mask_estimator.npz')


def _build(case, F):
    from padertorch_amd.contrib.jensheit.mask_estimator_example import MaskEstimator
    upd = dict(recurrent=dict(hidden_size=8), fully_connected=dict(hidden_size=[16, 12, 16], dropout=0.))
    for k, v in case['updates'].items():
        upd[k] = {**upd.get(k, {}), **v} if isinstance(v, dict) else v
    upd['fully_connected']['input_size'] = 16 if upd['recurrent'].get('bidirectional', True) else 8
    return MaskEstimator.from_defaults(num_features=F, **upd)


def test_fully_connected_stack_layout():
    """Layer names = the reference's state_dict keys (its doctest, fully_connected.py:37-47); hidden_size None / int / list."""
    from padertorch_amd.modules import fully_connected_stack
    net = fully_connected_stack(513, [1024, 1024], 1024)
    assert [n for n, _ in net.named_children()] == ['dropout_0', 'linear_0', 'relu_0', 'dropout_1', 'linear_1', 'relu_1',
                                                    'dropout_2', 'linear_2']
    assert net.linear_0.weight.shape == (1024, 513) and net.dropout_1.p == 0.5
    assert [n for n, _ in fully_connected_stack(4, None, 3).named_children()] == ['dropout_0', 'linear_0']
    net = fully_connected_stack(4, 5, 3, activation='elu', dropout=0.1, output_activation='sigmoid')
    assert [n for n, _ in net.named_children()] == ['dropout_0', 'linear_0', 'elu_0', 'dropout_1', 'linear_1', 'sigmoid_1']
    assert [n for n, _ in fully_connected_stack(4, (5,), 3, output_activation='identity').named_children()][-1] == 'linear_1'
    with pytest.raises(TypeError):
        fully_connected_stack(4, 'x', 3)
    x = torch.randn(7, 4)
    net.eval()
    want = torch.sigmoid(torch.nn.functional.linear(torch.nn.functional.elu(torch.nn.functional.linear(
        x, net.linear_0.weight, net.linear_0.bias)), net.linear_1.weight, net.linear_1.bias))
    np.testing.assert_allclose(net(x).detach().numpy(), want.detach().numpy(), rtol=1e-6, atol=1e-6)       # CPU tensors: torch's path


def test_defaults_and_state_dict_keys(g11):
    """finalize_dogmatic_config produces the reference's default sub-configurations; the built module has the reference's
    state_dict keys and shapes, and loads its weights."""
    from padertorch_amd.contrib.jensheit.mask_estimator_example import MaskEstimator, MaskKeys
    from padertorch_amd.modules import StatefulLSTM, Normalization, fully_connected_stack
    cfg = MaskEstimator.finalize_dogmatic_config(dict(num_features=513))
    assert cfg['recurrent'] == dict(factory=StatefulLSTM, input_size=513, hidden_size=256, bidirectional=True, batch_first=False)
    assert cfg['fully_connected'] == dict(factory=fully_connected_stack, input_size=512, hidden_size=[1024] * 3, output_size=1026)
    assert cfg['normalization'] == dict(factory=Normalization, data_format='tbf', shape=(1, 1, 1, 513), statistics_axis='t',
                                        independent_axis='f', batch_axis='b', sequence_axis='t')
    assert MaskEstimator.finalize_dogmatic_config(dict(num_features=5, normalization=None))['normalization'] is None
    assert MaskKeys.SPEECH_MASK_PRED == 'speech_mask_prediction' and MaskKeys.VAD_LOGITS == 'vad_logits'
    for case in json.loads(str(g11['cases'])):
        me = _build(case, case['F'])
        sd = {k[len(case['key']) + 4:]: torch.from_numpy(g11[k]) for k in g11.files if k.startswith(case['key'] + '/sd/')}
        assert set(me.state_dict()) == set(sd), case['key']
        me.load_state_dict(sd, strict=True)
    with pytest.raises(NotImplementedError):
        MaskEstimator.from_defaults(num_features=5, use_log=True)


@pytest.mark.gpu
def test_mask_estimator_vs_reference(g11):
    """Forward (every output key), loss and the gradient of every parameter against the reference's, for ragged
    multi-channel batches (incl. the reference's channel-major fold of an example-major flattening), the VAD head, and
    the variant without normalisation / unidirectional / tanh."""
    dev = torch.device('cuda:0')
    for case in json.loads(str(g11['cases'])):
        key = case['key']
        me = _build(case, case['F'])
        me.load_state_dict({k[len(key) + 4:]: torch.from_numpy(g11[k]) for k in g11.files if k.startswith(key + '/sd/')})
        me.to(dev).eval()
        x = [torch.from_numpy(g11[f'{key}/x{b}']).to(dev) for b in range(len(case['frames']))]
        out = me(x, case['frames'])
        assert set(out) == {k[len(key) + 5:] for k in g11.files if k.startswith(key + '/out/')}
        loss = 0.
        for k, v in sorted(out.items()):
            np.testing.assert_allclose(v.detach().cpu().numpy(), g11[f'{key}/out/{k}'], atol=2e-5, err_msg=f'{key} {k}')
            loss = loss + (v * torch.from_numpy(g11[f'{key}/w/{k}']).to(dev)).sum()
        np.testing.assert_allclose(float(loss), float(g11[f'{key}/loss']), rtol=1e-4, atol=1e-4)
        loss.backward()
        for k, p in me.named_parameters():
            want = g11[f'{key}/grad/{k}']
            if want.size == 0:
                assert p.grad is None or float(p.grad.abs().max()) == 0., (key, k)
                continue
            got = p.grad.cpu().numpy()
            assert np.abs(got - want).max() <= 2e-4 * max(np.abs(want).max(), 1e-3), (key, k, np.abs(got - want).max())
        # states: dropped between calls unless reuse_states (modul.py:126-127)
        assert me.recurrent.states is not None
        me(x, case['frames'])
