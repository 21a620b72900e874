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
    (64, 600, 1, True, [9] * 60 + [7] * 6 + [2] * 4),     # B = 70: row tiles run as several persistent launches
])
def test_packed_lstm_vs_torch_cpu(I, H, layers, bidir, lens, monkeypatch):
    from padertorch_amd.ops import packed_lstm, lstm as L
    monkeypatch.setattr(L, 'CHECK_PERSISTENT_ERRORS', True)
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


@pytest.mark.library_path
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


@pytest.mark.parametrize('lens', [[9, 9, 7, 4, 4, 1], [6] * 5])     # ragged / equal lengths (shifted-view weight gradients)
@pytest.mark.parametrize('persistent', [True, False])
def test_initial_and_final_states_vs_torch_cpu(persistent, lens, monkeypatch):
    """``lstm(packed, (h0, c0))`` semantics of torch.nn.LSTM: the initial state applies to every
    sequence's first processed step (per direction, also for ragged batches), (h_n, c_n) are the states
    after each sequence's last step; gradients w.r.t. inputs and weights include the state terms."""
    from padertorch_amd.ops import packed_lstm, lstm as L
    monkeypatch.setattr(L, 'CHECK_PERSISTENT_ERRORS', True)
    monkeypatch.setattr(L, 'PERSISTENT', persistent)
    torch.manual_seed(11)
    I, H, layers = 13, 24, 2
    ref = torch.nn.LSTM(I, H, layers, bidirectional=True)
    dut = torch.nn.LSTM(I, H, layers, bidirectional=True)
    dut.load_state_dict(ref.state_dict())
    dut = dut.to(DEV)
    h0, c0 = torch.randn(layers * 2, len(lens), H), torch.randn(layers * 2, len(lens), H)
    xs = [torch.randn(l, I) for l in lens]
    xr = [x.clone().requires_grad_(True) for x in xs]
    xd = [x.clone().to(DEV).requires_grad_(True) for x in xs]
    yr, (hr, cr) = ref(pack_sequence(xr), (h0, c0))
    yd, (hd, cd) = packed_lstm(dut, pack_sequence(xd), hx=(h0.to(DEV), c0.to(DEV)))
    np.testing.assert_allclose(yd.data.detach().cpu().numpy(), yr.data.detach().numpy(), atol=3e-6)
    np.testing.assert_allclose(hd.detach().cpu().numpy(), hr.detach().numpy(), atol=3e-6)
    np.testing.assert_allclose(cd.detach().cpu().numpy(), cr.detach().numpy(), atol=3e-6)
    g = torch.randn(yr.data.shape)
    (yr.data * g).sum().backward()
    (yd.data * g.to(DEV)).sum().backward()
    for a, b in zip(xd, xr):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), atol=2e-5, rtol=1e-4)
    for (n, pd), pr in zip(dut.named_parameters(), ref.parameters()):
        np.testing.assert_allclose(pd.grad.cpu().numpy(), pr.grad.numpy(),
                                   atol=3e-5 * max(1., pr.grad.abs().max().item()), err_msg=n)
    # zero-state call with return_state == plain call + states
    y0, (hn0, cn0) = packed_lstm(dut, pack_sequence([x.detach() for x in xd]), return_state=True)
    yr0, (hr0, cr0) = ref(pack_sequence(xs))
    np.testing.assert_allclose(hn0.detach().cpu().numpy(), hr0.detach().numpy(), atol=3e-6)
    np.testing.assert_allclose(cn0.detach().cpu().numpy(), cr0.detach().numpy(), atol=3e-6)
    # a state that requires a gradient: served by the persistent split kernels (test_gradients_wrt_the_initial_state), refused -
    # in the backward pass - by the one-launch-per-timestep kernels
    out = packed_lstm(dut, pack_sequence([x.detach() for x in xd]), hx=(h0.to(DEV).requires_grad_(True), c0.to(DEV)))
    if not persistent:
        with pytest.raises(NotImplementedError):
            out[0].data.sum().backward()


def test_stateful_lstm_streams_like_the_reference_module():
    """StatefulLSTM (modules/recurrent.py:5-47): chunked calls with carried states == one long call."""
    from padertorch_amd.modules import StatefulLSTM
    torch.manual_seed(5)
    m = StatefulLSTM(17, 20, num_layers=2, bidirectional=False, batch_first=True, save_states=True).to(DEV)
    ref = torch.nn.LSTM(17, 20, 2, batch_first=True)
    ref.load_state_dict({k[len('lstm.'):]: v.cpu() for k, v in m.state_dict().items()})
    x = torch.randn(3, 30, 17)
    want, _ = ref(x)
    got = torch.cat([m(x[:, :12].to(DEV)), m(x[:, 12:19].to(DEV)), m(x[:, 19:].to(DEV))], 1)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), atol=5e-6)
    assert m.states is not None and m.states[0].shape == (2, 3, 20)
    del m.states
    np.testing.assert_allclose(m(x.to(DEV)).detach().cpu().numpy(), want.detach().numpy(), atol=5e-6)
    m2 = StatefulLSTM(17, 20, bidirectional=True, save_states=False).to(DEV)
    assert m2(x.to(DEV)).shape == (3, 30, 40) and m2.states is None


def test_stateful_lstm_carries_the_graph_across_calls():
    """``modules/recurrent.py:42``: the reference stores ``(h_n, c_n)`` as ``torch.nn.LSTM`` returns them - with their graph.  A loss on
    the second chunk's output then reaches the FIRST chunk's input and the parameters through the carried states: equal to autograd
    through torch's LSTM on the CPU; backward a second time through the freed first chunk raises, as with the reference."""
    from padertorch_amd.modules import StatefulLSTM
    torch.manual_seed(6)
    for bidir in (False, True):
        m = StatefulLSTM(9, 24, num_layers=2, bidirectional=bidir, batch_first=True, save_states=True).to(DEV)
        ref = torch.nn.LSTM(9, 24, 2, batch_first=True, bidirectional=bidir)
        ref.load_state_dict({k[len('lstm.'):]: v.cpu() for k, v in m.state_dict().items()})
        x1, x2 = torch.randn(4, 11, 9), torch.randn(4, 7, 9)
        a1, a2 = x1.clone().requires_grad_(), x2.clone().requires_grad_()
        y1, st = ref(a1)
        y2, _ = ref(a2, st)
        g = torch.randn_like(y2)
        (y2 * g).sum().backward()
        b1, b2 = x1.to(DEV).requires_grad_(), x2.to(DEV).requires_grad_()
        z1 = m(b1)
        assert m.states[0].requires_grad and m.states[1].requires_grad
        z2 = m(b2)
        np.testing.assert_allclose(z2.detach().cpu().numpy(), y2.detach().numpy(), atol=5e-6)
        (z2 * g.to(DEV)).sum().backward()
        assert float(a1.grad.abs().max()) > 0
        np.testing.assert_allclose(b1.grad.cpu().numpy(), a1.grad.numpy(), atol=2e-5 * float(a1.grad.abs().max()) + 1e-7)
        np.testing.assert_allclose(b2.grad.cpu().numpy(), a2.grad.numpy(), atol=2e-5 * float(a2.grad.abs().max()) + 1e-7)
        for (k, p), (_, q) in zip(m.lstm.named_parameters(), ref.named_parameters()):
            np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.numpy(), atol=2e-5 * float(q.grad.abs().max()) + 1e-7, err_msg=k)
        z3 = m(torch.randn(4, 3, 9, device=DEV))
        with pytest.raises(RuntimeError, match='second time|already been freed'):
            z3.sum().backward()
        del m.states


def test_random_configurations_vs_torch_cpu():
    """scripts/fuzz_lstm.py: 16 random (B, T, H, I, layers, directions, ragged / equal lengths, initial
    state) configurations - outputs, final states, input and parameter gradients against torch.nn.LSTM on
    the CPU (covers partial row tiles, H not a multiple of 16, T = 1, tile-group launches)."""
    import importlib.util
    from pathlib import Path
    path = Path(__file__).resolve().parent.parent / 'scripts' / 'fuzz_lstm.py'
    spec = importlib.util.spec_from_file_location('fuzz_lstm', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(16, 7, verbose=False) < 2e-5


def test_full_size_properties(monkeypatch):
    """BASELINE size (32 x 253 frames, 3 x BLSTM-600): properties that need no CPU reference.
    (a) the persistent kernels and the one-launch-per-timestep kernels (independent implementations) agree;
    (b) sequences are independent: the first 16 sequences alone (other tile shape: 16 x 8 units, one row tile)
        give the same outputs and input gradients as inside the batch of 32 (16 x 12 units, two row tiles);
    (c) time reversal: reversing every sequence and swapping the two directions' weights reverses the output."""
    from padertorch_amd.ops import packed_lstm, lstm as L
    monkeypatch.setattr(L, 'CHECK_PERSISTENT_ERRORS', True)
    torch.manual_seed(3)
    B, T, I, H = 32, 253, 257, 600
    lstm = torch.nn.LSTM(I, H, 3, bidirectional=True).to(DEV)
    xs = [torch.randn(T, I, device=DEV, requires_grad=True) for _ in range(B)]
    g = torch.randn(T * B, 2 * H, device=DEV)

    def run(seqs, gout):
        for x in seqs:
            x.grad = None
        y = packed_lstm(lstm, pack_sequence(seqs)).data
        (y * gout).sum().backward()
        return y.detach(), [x.grad.clone() for x in seqs]

    y, gx = run(xs, g)
    # (a)
    monkeypatch.setattr(L, 'PERSISTENT', False)
    y2, gx2 = run(xs, g)
    monkeypatch.setattr(L, 'PERSISTENT', True)
    assert float((y - y2).abs().max()) < 2e-5
    assert max(float((a - b).abs().max()) for a, b in zip(gx, gx2)) < 2e-4
    # (b)
    half = 16
    yv, gv = y.view(T, B, 2 * H), g.view(T, B, 2 * H)
    yh, gxh = run(xs[:half], gv[:, :half].reshape(T * half, 2 * H).contiguous())
    assert float((yh.view(T, half, 2 * H) - yv[:, :half]).abs().max()) < 2e-5
    assert max(float((a - b).abs().max()) for a, b in zip(gxh, gx[:half])) < 2e-4
    # (c)
    sd = lstm.state_dict()
    swapped = torch.nn.LSTM(I, H, 3, bidirectional=True).to(DEV)
    # layer l > 0 sees [h_fwd, h_bwd] of the layer below, which swap places too: permute the input columns
    new = {}
    for k, v in sd.items():
        k2 = k[:-len('_reverse')] if k.endswith('_reverse') else k + '_reverse'
        if k.startswith('weight_ih_l') and not k.startswith('weight_ih_l0'):
            v = torch.cat([v[:, H:], v[:, :H]], 1)
        new[k2] = v
    swapped.load_state_dict(new)
    with torch.no_grad():
        yr = packed_lstm(swapped, pack_sequence([x.detach().flip(0) for x in xs[:8]])).data.view(T, 8, 2, H)
        y8 = packed_lstm(lstm, pack_sequence([x.detach() for x in xs[:8]])).data.view(T, 8, 2, H)
    assert float((yr.flip(0).flip(2) - y8).abs().max()) < 2e-5


def test_timed_out_persistent_launch_is_reported(monkeypatch):
    """A bounded spin that runs out (here: a poll budget of ONE) ends the launch, bumps the device's error word and
    `check_errors` raises once; the next, normal launch is clean again."""
    from padertorch_amd.ops import lstm as L
    from padertorch_amd.ops import packed_lstm
    from torch.nn.utils.rnn import pack_sequence
    if not L.PERSISTENT:
        pytest.skip('persistent kernels disabled')
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(40, 200, 1, bidirectional=True).cuda()
    xs = [torch.randn(100, 40, device='cuda') for _ in range(4)]
    L.check_errors()                                      # nothing pending from earlier tests
    monkeypatch.setenv('PTMI_LSTM_MAX_POLLS', '1')
    with torch.no_grad(), pytest.raises(RuntimeError, match='timed out'):
        packed_lstm(lstm, pack_sequence(xs))              # inference calls check the word themselves
    monkeypatch.delenv('PTMI_LSTM_MAX_POLLS')
    with torch.no_grad():
        y = packed_lstm(lstm, pack_sequence(xs))
    L.check_errors()
    import copy
    ref = copy.deepcopy(lstm).cpu()(pack_sequence([x.cpu() for x in xs]))[0].data
    assert float((y.data.cpu() - ref).detach().abs().max()) < 2e-5


@pytest.mark.parametrize('I,H,ndir', [(257, 600, 2), (1200, 600, 2), (40, 36, 1), (7, 20, 2)])
def test_weight_prep_forms(I, H, ndir):
    """ptmi_lstm_weight_prep: stacked / padded / transposed parameter forms and the two absmax words, bit for bit."""
    import padertorch_amd.ops.library  # noqa: F401
    torch.manual_seed(I + H)
    G, KP = 4 * H, (H + 15) // 16 * 16
    w_ih = [torch.randn(G, I, device='cuda') * 0.3 for _ in range(ndir)]
    w_hh = [torch.randn(G, H, device='cuda') * 0.2 for _ in range(ndir)]
    b_ih = [torch.randn(G, device='cuda') for _ in range(ndir)]
    b_hh = [torch.randn(G, device='cuda') for _ in range(ndir)]
    w_hh[-1][3, 5] = -7.5                                   # the maximum: negative, in the last direction
    cat, bias, w_pad, w_t, amax = torch.ops.ptmi.lstm_weight_prep(w_ih, w_hh, b_ih, b_hh, KP)
    Ipad = (I + 3) // 4 * 4
    assert cat.shape == (ndir * G, Ipad) and w_pad.shape == (ndir, G, KP) and w_t.shape == (ndir, H, G)
    assert torch.equal(cat[:, :I], torch.cat(w_ih, 0)) and float(cat[:, I:].abs().sum()) == 0
    assert torch.equal(bias, torch.cat([a + b for a, b in zip(b_ih, b_hh)]))
    assert torch.equal(w_pad[:, :, :H], torch.stack(w_hh)) and float(w_pad[:, :, H:].abs().sum()) == 0
    assert torch.equal(w_t, torch.stack(w_hh).transpose(1, 2))
    got = amax.view(torch.float32).cpu()
    assert float(got[0]) == float(torch.cat(w_ih, 0).abs().max()) and float(got[1]) == 7.5


def test_weight_prep_of_parameters_at_odd_addresses():
    """Parameters that are views at 4-byte (not 16-byte) aligned addresses of a flat buffer, input width a multiple of 4: the float4
    copy of the stacked input weights must not be taken (round 6's ``vec_ih``)."""
    import padertorch_amd.ops.library  # noqa: F401
    torch.manual_seed(5)
    I, H, ndir = 8, 20, 2
    G, KP = 4 * H, 32
    flat = torch.randn(1 + ndir * (G * I + 1), device='cuda')
    w_ih = [flat[1 + d * (G * I + 1):1 + d * (G * I + 1) + G * I].view(G, I) for d in range(ndir)]
    assert all(w.data_ptr() % 16 for w in w_ih)
    w_hh = [torch.randn(G, H, device='cuda') for _ in range(ndir)]
    b = [torch.randn(G, device='cuda') for _ in range(ndir)]
    cat, bias, w_pad, w_t, amax = torch.ops.ptmi.lstm_weight_prep(w_ih, w_hh, b, b, KP)
    assert torch.equal(cat, torch.cat(w_ih, 0)) and torch.equal(w_t, torch.stack(w_hh).transpose(1, 2))
    got = amax.view(torch.float32).cpu()
    assert float(got[0]) == float(torch.cat(w_ih, 0).abs().max()) and float(got[1]) == float(torch.stack(w_hh).abs().max())


@pytest.mark.parametrize('lens', [[130] * 32, [200, 180, 180, 131, 77, 64, 3]])
def test_in_place_weight_gradients_and_time_ranges(lens, monkeypatch):
    """The Trainer's path - weight gradients accumulated in place on the side stream - against autograd through torch's CPU
    LSTM; then the backward recurrence of the same layer in TWO launches over step ranges (the C ABI's
    ptmi_lstm_backward_persistent_range, whose whole-range form the initial-state gradients use): bit-identical gate gradients."""
    import copy
    from padertorch_amd.ops import lstm as L
    from padertorch_amd.ops import packed_lstm
    torch.manual_seed(len(lens))
    I, H = 36, 600
    ref = torch.nn.LSTM(I, H, 2, bidirectional=True)
    xs = [torch.randn(n, I) for n in lens]
    w = [torch.randn(n, 2 * H) for n in lens]
    xr = [x.clone().requires_grad_() for x in xs]
    out = ref(pack_sequence(xr))[0]
    (out.data * pack_sequence(w).data).sum().backward()
    want = {k: p.grad.clone() for k, p in ref.named_parameters()}
    monkeypatch.setattr(L, 'DEFER_WGRAD', True)
    for chunks in (1,):
        net = copy.deepcopy(ref).cuda()
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
        xg = [x.cuda().requires_grad_() for x in xs]
        y = packed_lstm(net, pack_sequence(xg))
        (y.data * pack_sequence([v.cuda() for v in w]).data).sum().backward()
        L.sync_deferred()
        torch.cuda.synchronize()
        L.check_errors()
        assert float((y.data.cpu() - out.data).abs().max()) < 2e-5
        for a, b in zip(xg, xr):
            assert float((a.grad.cpu() - b.grad).abs().max()) < 2e-4 * max(1.0, float(b.grad.abs().max()))
        for k, p in net.named_parameters():
            scale = max(1.0, float(want[k].abs().max()))
            assert float((p.grad.cpu() - want[k]).abs().max()) < 3e-4 * scale, (chunks, k)
    # one layer's backward recurrence through the range entry point: [0, T) in one launch == [0, T/2) + [T/2, T) in two
    from padertorch_amd import _lib
    lib = _lib.load()
    if not lib.ptmi_lstm_split_enabled():
        return
    meta = L.pack_meta(pack_sequence(xs).batch_sizes, torch.device(DEV))
    T, B, rows = meta.T, meta.max_batch, meta.rows
    torch.manual_seed(1)
    gates = torch.rand(rows, 2 * 4 * H, device=DEV)          # saved activations: any values in (0, 1)
    c = torch.randn(rows, 2 * H, device=DEV)
    dhy = torch.randn(rows, 2 * H, device=DEV)
    w_t = (torch.randn(2, H, 4 * H, device=DEV) * 0.05).contiguous()
    res = []
    for cuts in ([0, T], [0, T // 2, T]):
        dg = torch.empty_like(gates)
        flags = torch.empty(int(lib.ptmi_lstm_scratch_elems(T, 2, B, H, 1)), dtype=torch.int32, device=DEV)
        carry = torch.empty(2, B, H, device=DEV)
        for a, b in zip(cuts, cuts[1:]):
            assert torch.ops.ptmi.lstm_recurrence_backward_range(gates, c, None, dhy, w_t, dg, flags, carry, meta.bs_dev, meta.offs_dev,
                                                                 T, B, rows, H, 2, a, b, False)
        torch.cuda.synchronize()
        L.check_errors()
        res.append(dg)
    assert torch.equal(res[0], res[1])


@pytest.mark.gpu
@pytest.mark.parametrize('B,T,H', [(32, 37, 600), (16, 20, 40), (64, 25, 600), (48, 9, 100)])
def test_input_gradient_from_handoff_planes(B, T, H, monkeypatch):
    """Equal-length batches whose size is a multiple of 16: dx = dgates W_ih runs on the bf16 planes GEMM straight from the
    backward recurrence's hand-off copy (ops.lstm.DX_FROM_HANDOFF).  Same gradients as the in-register split GEMM on the
    row-major dgates (both fp32-equivalent), and as torch's CPU LSTM."""
    from padertorch_amd.ops import lstm as L
    from padertorch_amd import _lib
    if not _lib.load().ptmi_lstm_handoff_cols(H, 1):
        pytest.skip('split recurrence kernels not active')
    torch.manual_seed(B + T + H)
    I = 2 * H
    lstm = torch.nn.LSTM(I, H, 1, bidirectional=True).cuda()
    xs = [torch.randn(T, I, device='cuda') * 0.5 for _ in range(B)]
    w = torch.randn(T * B, 2 * H, device='cuda')

    def run(flag):
        monkeypatch.setattr(L, 'DX_FROM_HANDOFF', flag)
        x = torch.nn.utils.rnn.pack_sequence(xs).data.clone().requires_grad_(True)
        packed = torch.nn.utils.rnn.PackedSequence(x, torch.full((T,), B, dtype=torch.int64))
        lstm.zero_grad()
        y = L.packed_lstm(lstm, packed)
        (y.data * w).sum().backward()
        return x.grad.clone()

    dx_planes = run(True)
    dx_split = run(False)
    scale = float(dx_split.abs().max())
    assert float((dx_planes - dx_split).abs().max()) < 2e-5 * scale
    # reference: torch CPU
    ref = torch.nn.LSTM(I, H, 1, bidirectional=True)
    ref.load_state_dict(lstm.state_dict())
    xc = torch.nn.utils.rnn.pack_sequence([t.cpu() for t in xs]).data.clone().requires_grad_(True)
    yc, _ = ref(torch.nn.utils.rnn.PackedSequence(xc, torch.full((T,), B, dtype=torch.int64)))
    (yc.data * w.cpu()).sum().backward()
    assert float((dx_planes.cpu() - xc.grad).abs().max()) < 1e-4 * scale


@pytest.mark.gpu
@pytest.mark.parametrize('lens', [[9, 9, 9, 9], [12, 10, 7, 7, 3]])
@pytest.mark.parametrize('with_hx', [True, False])
def test_gradients_through_the_final_state(lens, with_hx):
    """``packed_lstm(..., return_state=True)`` returns ``(h_n, c_n)`` WITH their graph, like ``torch.nn.LSTM`` (the reference's
    StatefulLSTM hands them on, modules/recurrent.py:42): a loss on the output, on h_n and on c_n - parameter, input and initial-state
    gradients against torch's CPU LSTM (the gradient of c_n enters the backward kernel at every sequence's last step:
    ptmi_lstm_backward_persistent_states), two layers, both directions, equal and ragged lengths."""
    from padertorch_amd.ops import lstm as L
    torch.manual_seed(21)
    I, H, layers = 10, 12, 2
    B = len(lens)
    lstm = torch.nn.LSTM(I, H, layers, bidirectional=True).cuda()
    ref = torch.nn.LSTM(I, H, layers, bidirectional=True)
    ref.load_state_dict(lstm.state_dict())
    xs = [torch.randn(n, I) for n in lens]
    h0 = torch.randn(2 * layers, B, H) * 0.5
    c0 = torch.randn(2 * layers, B, H) * 0.5
    w = torch.randn(sum(lens), 2 * H)
    wh, wc = torch.randn(2 * layers, B, H), torch.randn(2 * layers, B, H)

    def run(mod, dev, fn):
        mod.zero_grad()
        xd = [x.to(dev).requires_grad_(True) for x in xs]
        hh = h0.to(dev).requires_grad_(True)
        cc = c0.to(dev).requires_grad_(True)
        out, (hn, cn) = fn(mod, pack_sequence(xd), (hh, cc) if with_hx else None)
        ((out.data * w.to(dev)).sum() + (hn * wh.to(dev)).sum() + (cn * wc.to(dev)).sum()).backward()
        states = (hh.grad.cpu(), cc.grad.cpu()) if with_hx else ()
        return (out.data.detach().cpu(), hn.detach().cpu(), cn.detach().cpu(), [x.grad.cpu() for x in xd],
                [p.grad.cpu().clone() for p in mod.parameters()], states)

    got = run(lstm, 'cuda', lambda m, p, hx: L.packed_lstm(m, p, hx=hx, return_state=True))
    want = run(ref, 'cpu', lambda m, p, hx: m(p, hx))
    for a, b in zip(got[:3], want[:3]):
        assert float((a - b).abs().max()) < 2e-5
    for a, b in zip(got[3] + got[4] + list(got[5]), want[3] + want[4] + list(want[5])):
        assert float((a - b).abs().max()) < 1e-4 * max(1., float(b.abs().max())), float((a - b).abs().max())


@pytest.mark.gpu
def test_nan_travels_through_the_data_as_flag_hand_off():
    """The persistent kernels synchronise on the DATA (planes pre-filled with 0xFFFF in every 16-bit value; a consumer waits
    while a value it needs still is that pattern).  A NaN hidden state must pass as data - the conversions produce the
    canonical quiet NaN, not the fill pattern - and not stall the recurrence into its time-out."""
    from padertorch_amd.ops import lstm as L
    torch.manual_seed(3)
    B, T, I, H = 32, 40, 64, 600
    lstm = torch.nn.LSTM(I, H, 1, bidirectional=True).cuda()
    xs = [torch.randn(T, I, device='cuda') for _ in range(B)]
    xs[5][7, 3] = float('nan')                    # poisons sequence 5 from step 7 on (forward) / up to step 7 (reverse)
    packed = torch.nn.utils.rnn.pack_sequence(xs)
    before = int(L.error_count(torch.device('cuda', 0)).item())
    with torch.no_grad():
        y = L.packed_lstm(lstm, packed)
    torch.cuda.synchronize()
    out = y.data.view(T, B, 2 * H)
    assert int(L.error_count(torch.device('cuda', 0)).item()) == before          # no wait ran out
    assert torch.isnan(out[7:, 5, :H]).all() and torch.isnan(out[:8, 5, H:]).all()
    ok = torch.ones(T, B, dtype=torch.bool, device='cuda')
    ok[:, 5] = False
    assert torch.isfinite(out[ok]).all()                                       # the other sequences are untouched
    assert torch.isfinite(out[:7, 5, :H]).all() and torch.isfinite(out[8:, 5, H:]).all()


@pytest.mark.gpu
@pytest.mark.parametrize('lens', [[9, 9, 9, 9], [12, 10, 7, 7, 3]])
def test_gradients_wrt_the_initial_state(lens):
    """``packed_lstm(lstm, packed, hx=(h0, c0))`` with states that require a gradient: ``dL/dh0`` (through ``W_hh`` into each
    sequence's first processed step) and ``dL/dc0`` (the recurrence kernel's cell-state gradient behind its last step) against
    torch's CPU LSTM, two layers, both directions, ragged batch."""
    from padertorch_amd.ops import lstm as L
    torch.manual_seed(11)
    I, H, layers = 10, 12, 2
    B = len(lens)
    lstm = torch.nn.LSTM(I, H, layers, bidirectional=True).cuda()
    ref = torch.nn.LSTM(I, H, layers, bidirectional=True)
    ref.load_state_dict(lstm.state_dict())
    xs = [torch.randn(n, I) for n in lens]
    h0 = torch.randn(2 * layers, B, H) * 0.5
    c0 = torch.randn(2 * layers, B, H) * 0.5
    w = torch.randn(sum(lens), 2 * H)

    def run(mod, dev, fn):
        hh = h0.to(dev).requires_grad_(True)
        cc = c0.to(dev).requires_grad_(True)
        packed = pack_sequence([x.to(dev) for x in xs])
        out = fn(mod, packed, (hh, cc))
        (out.data * w.to(dev)).sum().backward()
        return out.data.detach().cpu(), hh.grad.cpu(), cc.grad.cpu()

    y, gh, gc = run(lstm, 'cuda', lambda m, p, hx: L.packed_lstm(m, p, hx=hx)[0])
    yr, ghr, gcr = run(ref, 'cpu', lambda m, p, hx: m(p, hx)[0])
    assert float((y - yr).abs().max()) < 2e-5
    assert float((gh - ghr).abs().max()) < 1e-4 * max(1., float(ghr.abs().max())), float((gh - ghr).abs().max())
    assert float((gc - gcr).abs().max()) < 1e-4 * max(1., float(gcr.abs().max())), float((gc - gcr).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('wgs,threads,lds', [(256, 256, 0), (64, 512, 65536), (512, 256, 0)])
def test_recurrences_next_to_a_cu_occupying_kernel(wgs, threads, lds, monkeypatch):
    """the persistent recurrence kernels need all their workgroups co-resident; a communication kernel on
    another queue (RCCL's channels during the bucketed all-reduce of the data-parallel Trainer, trainer.py:396-442) holds CUs
    meanwhile.  Stand-in: ptmi_test_occupy (libptmi_testhooks.so) keeps `wgs` workgroups resident for ~1.5 ms, launched back to back on a third
    stream while a BLSTM layer of the BASELINE size runs forward and backward on the main stream: no timed-out wait, and
    bit-identical results to the undisturbed run (the hand-off protocol does not depend on timing)."""
    from padertorch_amd import _lib
    from padertorch_amd.ops import packed_lstm, lstm as L
    lib = _lib.load()
    torch.manual_seed(11)
    B, T, I, H = 32, 120, 64, 600
    lstm = torch.nn.LSTM(I, H, 1, bidirectional=True).to(DEV)
    xs = [torch.randn(T, I, device=DEV) for _ in range(B)]
    g = torch.randn(T * B, 2 * H, device=DEV)

    def run(disturb):
        lstm.zero_grad()
        xd = [x.clone().requires_grad_(True) for x in xs]
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        if disturb:
            with torch.cuda.stream(side):
                for _ in range(12):          # ~18 ms of occupancy: covers the forward and the backward recurrence
                    _lib.check(_lib.test_hooks().ptmi_test_occupy(wgs, threads, lds, 150000, _lib.stream(torch.device(DEV))), 'occupy')
        y = packed_lstm(lstm, pack_sequence(xd))
        (y.data * g).sum().backward()
        torch.cuda.synchronize()
        L.check_errors()                     # raises on a timed-out bounded spin
        return y.data.detach().clone(), [x.grad.clone() for x in xd], [p.grad.clone() for p in lstm.parameters()]

    base = run(False)
    got = run(True)
    assert torch.equal(base[0], got[0])
    for a, b in zip(base[1] + base[2], got[1] + got[2]):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('B,T,H,ndir', [(32, 21, 600, 2), (16, 9, 40, 2), (64, 11, 600, 2), (48, 7, 100, 1), (16, 5, 20, 1)])
def test_backward_recurrence_hands_the_weight_gradient_operand_on(B, T, H, ndir):
    """``ptmi_lstm_backward_persistent_planes``: the gate gradients leave the backward recurrence as bf16 (hi, lo) planes of
    ``dgates^T`` - bit for bit what ``ptmi_pack_planes_t_bf16`` makes of the row-major fp32 gate gradients of the plain launch
    (both are the same two roundings of the same fp32 values), with and without the row-major tensor next to them, for the whole
    recurrence and for two launches over step ranges (each range's planes hold that range's rows per direction); odd row counts
    (rows % 32 == 16) get their half k block zeroed."""
    from padertorch_amd import _lib
    from padertorch_amd.ops import lstm as L
    lib = _lib.load()
    rows, G = T * B, 4 * H
    if not lib.ptmi_lstm_backward_planes_ok(T, ndir, B, rows, H):
        pytest.skip('split recurrence kernels not active')
    assert not lib.ptmi_lstm_backward_planes_ok(T, ndir, B, rows - 1, H)          # ragged batches keep the row-major route
    assert not lib.ptmi_lstm_backward_planes_ok(T, ndir, B + 8, T * (B + 8), H)   # so do batches that are no multiple of 16
    meta = L.pack_meta(torch.full((T,), B, dtype=torch.int64), torch.device(DEV))
    torch.manual_seed(B + T + H)
    gates = torch.rand(rows, ndir * G, device=DEV)
    c = torch.randn(rows, ndir * H, device=DEV)
    dhy = torch.randn(rows, ndir * H, device=DEV)
    w_t = (torch.randn(ndir, H, G, device=DEV) * 0.05).contiguous()
    n_scratch = int(lib.ptmi_lstm_scratch_elems(T, ndir, B, H, 1))

    dg_ref, _ = torch.ops.ptmi.lstm_recurrence_backward(gates, c, None, dhy, w_t, meta.bs_dev, meta.offs_dev, meta.bs_host.ctypes.data,
                                                        meta.offs_host.ctypes.data, T, B, rows, H, ndir, True)
    torch.cuda.synchronize()

    def planes_of(rows_d):           # per direction: the transposing bf16 pack of its [n, 4H] block
        return torch.cat([torch.ops.ptmi.pack_planes_bf16(r, True).view(torch.int16) for r in rows_d])

    dgv = dg_ref.view(rows, ndir, G)
    for with_rows in (True, False):
        for cuts in ([0, T], [0, T // 2, T]):
            dg = torch.full_like(gates, float('nan')) if with_rows else None
            flags = torch.empty(n_scratch, dtype=torch.int32, device=DEV)
            carry = torch.empty(ndir, B, H, device=DEV)
            for a, b in zip(cuts, cuts[1:]):
                n = (b - a) * B
                planes = torch.full((ndir * int(lib.ptmi_planes_elems(G, n)),), -1, dtype=torch.int16, device=DEV).view(torch.bfloat16)
                assert torch.ops.ptmi.lstm_recurrence_backward_planes(gates, c, None, dhy, w_t, dg, planes, flags, carry, meta.bs_dev,
                                                                      meta.offs_dev, T, B, rows, H, ndir, a, b, False)
                torch.cuda.synchronize()
                L.check_errors()
                part = [((T - b) * B, (T - a) * B), (a * B, b * B)][:ndir]
                want = planes_of([dgv[r0:r1, d] for d, (r0, r1) in enumerate(part)])
                assert torch.equal(planes.view(torch.int16), want), (with_rows, cuts, a, b)
            if with_rows:
                assert torch.equal(dg, dg_ref)
    # the bias gradient and max |dgates| still come out of the launch (scratch tail)
    nflags = int(lib.ptmi_lstm_flags_elems(T, ndir, B)) + 8
    db = flags[flags.numel() - nflags - ndir * G:flags.numel() - nflags].view(torch.float32)
    torch.testing.assert_close(db, dg_ref.sum(0), rtol=2e-5, atol=2e-5 * float(dg_ref.abs().sum(0).max()))


@pytest.mark.parametrize('I,H,layers,bidir,bias,lens', [
    (9, 6, 2, True, True, [6, 5, 3]),
    (33, 10, 3, True, True, [17] * 5),
    (20, 601, 1, True, True, [12, 12, 7]),
    (7, 5, 2, False, True, [11, 11, 4, 1]),
    (12, 24, 2, True, False, [9, 8, 8, 2]),                # bias=False
    (12, 7, 1, True, False, [9, 8]),
])
def test_any_hidden_size_and_no_bias_vs_torch_cpu(I, H, layers, bidir, bias, lens, monkeypatch):
    """``hidden_size % 4 != 0`` and ``bias=False`` (the reference builds ``torch.nn.LSTM(F, units)`` for any ``units``,
    ``pit/model.py:60-66``): the kernels run the zero-padded LSTM (``ops.lstm._PaddedLstm``), outputs, states and every gradient equal
    torch's LSTM on the CPU - strict mode: nothing goes to MIOpen."""
    from padertorch_amd.ops import packed_lstm, lstm as L
    monkeypatch.setattr(L, 'CHECK_PERSISTENT_ERRORS', True)
    torch.manual_seed(I + H)
    ref = torch.nn.LSTM(I, H, layers, bidirectional=bidir, bias=bias)
    dut = torch.nn.LSTM(I, H, layers, bidirectional=bidir, bias=bias)
    dut.load_state_dict(ref.state_dict())
    dut = dut.to(DEV)
    assert L.unsupported_reason(dut, torch.zeros(1, I, device=DEV)) is None
    xs = [torch.randn(l, I) for l in lens]
    xr = [x.clone().requires_grad_(True) for x in xs]
    xd = [x.clone().to(DEV).requires_grad_(True) for x in xs]
    yr, (hr, cr) = ref(pack_sequence(xr))
    yd, (hd, cd) = packed_lstm(dut, pack_sequence(xd), return_state=True)
    np.testing.assert_allclose(yd.data.detach().cpu().numpy(), yr.data.detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(hd.detach().cpu().numpy(), hr.detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(cd.detach().cpu().numpy(), cr.detach().numpy(), atol=4e-6)
    g = torch.randn(yr.data.shape)
    (yr.data * g).sum().backward()
    (yd.data * g.to(DEV)).sum().backward()
    for a, b in zip(xd, xr):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), atol=2e-5, rtol=1e-4)
    for (n, pd), pr in zip(dut.named_parameters(), ref.parameters()):
        scale = max(1., pr.grad.abs().max().item())
        np.testing.assert_allclose(pd.grad.cpu().numpy(), pr.grad.numpy(), atol=3e-5 * scale, err_msg=n)
    # the plain call (no states), without a graph
    with torch.no_grad():
        y2 = packed_lstm(dut, pack_sequence([x.to(DEV) for x in xs]))
    np.testing.assert_allclose(y2.data.cpu().numpy(), yr.data.detach().numpy(), atol=2e-6)


def test_pit_model_with_an_odd_number_of_units_trains_on_the_hip_path(tmp_path):
    """``PermutationInvariantTrainingModel(units=50)``: three Trainer steps (in-place side-stream machinery around a padded BLSTM)
    against the oracle's."""
    import padertorch_amd as pt
    from oracle import features_np, torch_ref
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    rng = np.random.RandomState(0)
    exs = [features_np.synthetic_mixture(rng, n) for n in (4000, 3300, 2100)]
    torch.manual_seed(0)
    kw = dict(F=257, recurrent_layers=2, units=50, K=2)
    model = PermutationInvariantTrainingModel(**kw)
    ref = torch_ref.PITModelRef(**kw)
    ref.load_state_dict(model.state_dict())
    lw = dict(pit_ips_loss=1., pit_mse_loss=0.)
    t = pt.Trainer(model, tmp_path, pt.optimizer.Adam(gradient_clipping=1.), loss_weights=lw)
    t.to(torch.device(DEV))
    t._flat = t.optimizer.use_flat_grads()
    t.op_context.defer_wgrad = True
    model.train()
    f = [features_np.pre_batch_transform(s, y) for s, y in exs]
    batch = {k: [torch.from_numpy(e[k]) for e in f] for k in ['Y_abs', 'X_abs', 'cos_phase_difference']}
    opt = torch.optim.Adam(ref.parameters())
    for _ in range(3):
        loss, _, _, _ = t.train_step(model, {k: [v.to(DEV) for v in vs] for k, vs in batch.items()}, DEV)
        loss.backward()
        t.optimizer_step()
        ref_losses, _ = torch_ref.train_step(ref, opt, [batch], lw, 1.)
        assert abs(float(loss) - ref_losses[0]) < 1e-4, (float(loss), ref_losses[0])
    torch.cuda.synchronize()
    for (k, v), vr in zip(model.state_dict().items(), ref.state_dict().values()):
        np.testing.assert_allclose(v.cpu().numpy(), vr.numpy(), atol=3e-4, err_msg=k)
