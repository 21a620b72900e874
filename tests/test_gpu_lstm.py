"""HIP packed (B)LSTM recurrence vs torch.nn.LSTM on the CPU (fp32), forward and all gradients."""
import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_sequence, PackedSequence

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('I,H,layers,bidir,lens', [
    (9, 4, 2, True, [6, 5, 3]),              # the G6 toy shape
    (7, 24, 1, False, [11, 11, 4, 1]),
    (33, 40, 3, True, [17] * 5),
    (257, 600, 1, True, [40, 37, 37, 20, 9, 3]),          # model size, ragged, B < 16
    (1200, 600, 1, True, [23] * 40 + [11] * 5),           # B > 32: two M chunks
])
def test_packed_lstm_vs_torch_cpu(I, H, layers, bidir, lens):
    from padertorch_amd.ops import packed_lstm
    torch.manual_seed(I + H)
    ref = torch.nn.LSTM(I, H, layers, bidirectional=bidir)
    dut = torch.nn.LSTM(I, H, layers, bidirectional=bidir)
    dut.load_state_dict(ref.state_dict())
    dut = dut.to(DEV)
    xs = [torch.randn(l, I) for l in lens]
    xr = [x.clone().requires_grad_(True) for x in xs]
    xd = [x.clone().to(DEV).requires_grad_(True) for x in xs]
    yr, _ = ref(pack_sequence(xr))
    yd = packed_lstm(dut, pack_sequence(xd))
    assert isinstance(yd, PackedSequence) and torch.equal(yd.batch_sizes, yr.batch_sizes)
    np.testing.assert_allclose(yd.data.detach().cpu().numpy(), yr.data.detach().numpy(), atol=2e-6)
    g = torch.randn(yr.data.shape)
    (yr.data * g).sum().backward()
    (yd.data * g.to(DEV)).sum().backward()
    for a, b in zip(xd, xr):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), atol=2e-5, rtol=1e-4)
    for (n, pd), pr in zip(dut.named_parameters(), ref.parameters()):
        scale = max(1., pr.grad.abs().max().item())
        np.testing.assert_allclose(pd.grad.cpu().numpy(), pr.grad.numpy(), atol=3e-5 * scale, err_msg=n)


def test_model_uses_hip_lstm_and_matches_library_lstm():
    """Same weights, HIP recurrence vs torch.nn.LSTM (MIOpen) inside the PIT model."""
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    torch.manual_seed(0)
    model = PermutationInvariantTrainingModel(F=257, recurrent_layers=2, units=64, K=2).to(DEV)
    batch = dict(Y_abs=[torch.rand(t, 257, device=DEV) for t in [30, 28, 9]])
    model.hip_blstm = True
    a = model(batch)
    model.hip_blstm = False
    b = model(batch)
    for x, y in zip(a, b):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), atol=1e-5)
