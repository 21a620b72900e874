"""HIP STFT / iSTFT / feature front-end vs the oracle and the reference goldens (GPU, via the C ABI)."""
import numpy as np
import pytest
import torch

from oracle import features_np, stft_np

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _stft(c, **kw):
    from padertorch_amd.ops import STFT
    return STFT(c['size'], c['shift'], window=c['window'], window_length=c['window_length'],
                fading=c['fading'], pad=c['pad'], **kw)


def test_option_grid_vs_reference(g2):
    """Every golden case: forward within 1e-4*max|X| (fp32 FFT vs fp64 reference), inverse 1e-5."""
    for c in g2['cases']:
        x = torch.from_numpy(g2[c['x']].astype(np.float32)).to(DEV)
        st = _stft(c)
        X = st(x)
        ref = g2[c['name'] + '_X']
        assert tuple(X.shape) == ref.shape, c           # framing: exact
        tol = 1e-4 * np.abs(ref).max()
        np.testing.assert_allclose(X.cpu().numpy(), ref, atol=tol, err_msg=str(c))
        xi = st.inverse(torch.from_numpy(ref.astype(np.complex64)).to(DEV))
        refi = g2[c['name'] + '_xi']
        assert tuple(xi.shape) == refi.shape, c
        np.testing.assert_allclose(xi.cpu().numpy(), refi, atol=1e-4 * max(1., np.abs(refi).max()),
                                   err_msg=str(c))


def test_known_answer_generic_path(g1):
    """size=4 (not a fast-path size) through the generic kernel: cb/transform.py:219-232."""
    from padertorch_amd.ops import STFT
    c = g1['cb_stft']
    st = STFT(c['kwargs']['size'], c['kwargs']['shift'], window=c['kwargs']['window'],
              fading=c['kwargs']['fading'])
    X = st(torch.tensor(c['input'], dtype=torch.float32, device=DEV)).cpu().numpy()
    np.testing.assert_allclose(X.real, c['real'], atol=1e-5)
    np.testing.assert_allclose(X.imag, c['imag'], atol=1e-5)


@pytest.mark.parametrize('size,shift,wl', [(100, 40, None), (150, 64, 90), (4096, 1024, None)])
def test_generic_sizes_vs_oracle(size, shift, wl):
    from padertorch_amd.ops import STFT
    rng = np.random.RandomState(size)
    x = rng.standard_normal((3, 5 * size + 17)).astype(np.float32)
    st = STFT(size, shift, window='hann', window_length=wl)
    X = st(torch.from_numpy(x).to(DEV))
    ref = stft_np.stft(x, size, shift, window='hann', window_length=wl)
    assert tuple(X.shape) == ref.shape
    np.testing.assert_allclose(X.cpu().numpy(), ref, atol=2e-4 * np.abs(ref).max())
    xi = st.inverse(X)
    np.testing.assert_allclose(xi.cpu().numpy()[..., :x.shape[-1]], x, atol=2e-4)


@pytest.mark.parametrize('seed', range(24))
def test_random_options_vs_oracle(seed):
    """Random (size, shift, window_length, window, fading, pad, batch, signal length) against the numpy oracle: shapes, values, the
    frame bookkeeping, and the inverse where the options reconstruct (fading 'full' / True with pad)."""
    from padertorch_amd.ops import STFT
    rng = np.random.RandomState(500 + seed)
    size = int(rng.choice([32, 64, 100, 128, 256, 400, 512, 1024]))
    wl = int(rng.choice([size, size, max(8, size // 2), max(8, int(size * 0.8))]))
    shift = int(rng.randint(max(1, wl // 8), wl // 2 + 1))
    fading = [None, 'full', 'half', True, False][int(rng.randint(0, 5))]
    pad = bool(rng.randint(0, 2))
    window = str(rng.choice(['hann', 'blackman', 'hamming']))
    n = int(rng.randint(wl if not pad else 1, 6 * size + 50))
    B = int(rng.randint(1, 5))
    x = rng.standard_normal((B, n)).astype(np.float32)
    st = STFT(size, shift, window=window, window_length=wl, fading=fading, pad=pad)
    ref = stft_np.stft(x, size, shift, window=window, window_length=wl, fading=fading, pad=pad)
    X = st(torch.from_numpy(x).to(DEV))
    assert tuple(X.shape) == ref.shape, (X.shape, ref.shape)
    # (the frame bookkeeping is paderbox's formula, which - like the reference - counts 0 frames for a signal shorter than the window
    # although the padded transform yields one: equal to the oracle always, to the transform from one window on)
    assert st.samples_to_frames(n) == stft_np.samples_to_frames(n, wl, shift, pad=pad, fading=fading)
    padded_n = n + (0 if fading in (None, False) else (1 + (fading != 'half')) * (wl - shift))
    assert padded_n < wl or st.samples_to_frames(n) == ref.shape[-2]
    np.testing.assert_allclose(X.cpu().numpy(), ref, atol=2e-4 * max(np.abs(ref).max(), 1e-3))
    xi = st.inverse(X).cpu().numpy()
    ri = stft_np.istft(ref, size, shift, window=window, window_length=wl, fading=fading)
    assert xi.shape == ri.shape and st.frames_to_samples(ref.shape[-2]) == ri.shape[-1]
    np.testing.assert_allclose(xi, ri, atol=3e-4 * max(np.abs(ri).max(), 1e-3))


def test_representations(g2):
    from padertorch_amd.ops import STFT
    x = torch.from_numpy(g2['x_s512_h128'].astype(np.float32)).to(DEV)
    for rep in ['concat', 'stacked']:
        st = STFT(512, 128, complex_representation=rep)
        X = st(x)
        ref = g2[f'rep_{rep}_X']
        assert tuple(X.shape) == ref.shape
        np.testing.assert_allclose(X.cpu().numpy(), ref, atol=2e-4)
        xi = st.inverse(X)
        np.testing.assert_allclose(xi.cpu().numpy(), g2[f'rep_{rep}_xi'], atol=1e-4)


def test_doctest_shapes_and_frame_counts(g1):
    from padertorch_amd.ops import STFT
    for d in g1['doctest_shapes']:
        st = STFT(d['size'], d['shift'], window_length=d['window_length'], complex_representation=d['rep'])
        if 'inp' in d:
            assert list(st(torch.rand(d['inp'], device=DEV)).shape) == d['out']
        else:
            assert list(st.inverse(torch.rand(d['inverse_inp'], device=DEV)).shape) == d['inverse_out']
    for fc in g1['frame_counts']:
        st = STFT(fc['size'], fc['shift'], window_length=fc['window_length'], fading=fc['fading'],
                  complex_representation='concat')
        for n, fr in zip(fc['samples'], fc['frames']):
            assert st(torch.rand(n, device=DEV)).shape == (fr, fc['size'] + 2)


@pytest.mark.parametrize('B,N', [(4, 32000), (64, 64000)])
def test_full_size_roundtrip_and_linearity(B, N):
    """BASELINE sizes through size-independent properties: stft->istft identity, linearity, and a
    strided sample of bins against the fp64 oracle."""
    from padertorch_amd.ops import STFT
    g = torch.Generator(device='cpu').manual_seed(0)
    x = (0.1 * torch.randn(B, N, generator=g)).to(DEV)
    z = (0.1 * torch.randn(B, N, generator=g)).to(DEV)
    st = STFT(512, 128)
    X = st(x)
    T = (N + 384 + 127) // 128
    assert X.shape == (B, T, 257)
    np.testing.assert_allclose(st.inverse(X)[..., :N].cpu().numpy(), x.cpu().numpy(), atol=2e-6)
    lin = st(2 * x - 3 * z) - (2 * X - 3 * st(z))
    assert lin.abs().max().item() < 2e-4
    rows = [0, B // 2, B - 1]
    ref = stft_np.stft(x[rows].cpu().numpy(), 512, 128)
    np.testing.assert_allclose(X[rows].cpu().numpy(), ref, atol=1e-4 * np.abs(ref).max())


def test_ragged_rows_and_edge_lengths():
    """Empty-ish, shorter-than-window and ragged rows (zero padded rows + num_samples)."""
    from padertorch_amd.ops import STFT
    st = STFT(512, 128)
    rng = np.random.RandomState(1)
    lens = [1500, 1000, 513, 512, 511, 130, 1]
    x = np.zeros((len(lens), max(lens)), np.float32)
    for b, n in enumerate(lens):
        x[b, :n] = rng.standard_normal(n)
    ns = torch.tensor(lens, dtype=torch.int32, device=DEV)
    X = st(torch.from_numpy(x).to(DEV), num_samples=ns).cpu().numpy()
    for b, n in enumerate(lens):
        ref = stft_np.stft(x[b, :n], 512, 128)
        np.testing.assert_allclose(X[b, :ref.shape[0]], ref, atol=1e-4 * np.abs(ref).max())
        assert np.all(X[b, ref.shape[0]:] == 0)
    for n in [1, 127, 128, 129, 511, 512, 513]:
        xs = rng.standard_normal(n).astype(np.float32)
        ref = stft_np.stft(xs, 512, 128)
        got = st(torch.from_numpy(xs).to(DEV)).cpu().numpy()
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, atol=1e-4 * max(np.abs(ref).max(), 1e-3))


def test_autograd_adjoints():
    """backward(stft) and backward(istft) against autograd through the dense fp64 oracle maths."""
    from padertorch_amd.ops import STFT
    from oracle.torch_ref import ConvSTFT
    for kw in [dict(size=512, shift=128), dict(size=256, shift=10, window_length=20, window='hann'),
               dict(size=64, shift=24, window_length=50, window='hann', fading='half')]:
        st = STFT(**kw)
        ref = ConvSTFT(kw['size'], kw['shift'], window=kw.get('window', 'blackman'),
                       window_length=kw.get('window_length'), fading=kw.get('fading', 'full'))
        g = torch.Generator().manual_seed(3)
        x = torch.randn(2, 1300, generator=g)
        xd = x.to(DEV).requires_grad_(True)
        X = st(xd)
        gw = torch.randn(X.shape, dtype=torch.complex64, generator=g)
        (X * gw.to(DEV).conj()).real.sum().backward()
        xr = x.double().requires_grad_(True)
        Xr = ref(xr)
        (Xr * gw.to(torch.complex128).conj()).real.sum().backward()
        np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(),
                                   atol=2e-4 * xr.grad.abs().max().item(), err_msg=str(kw))
        # inverse: compare with autograd through irfft + overlap-add
        S = torch.randn(2, 9, kw['size'] // 2 + 1, dtype=torch.complex64, generator=g)
        Sd = S.to(DEV).requires_grad_(True)
        y = st.inverse(Sd)
        gy = torch.randn(y.shape, generator=g)
        (y * gy.to(DEV)).sum().backward()
        L = kw.get('window_length') or kw['size']
        ws = torch.from_numpy(stft_np.biorthogonal_window(
            stft_np.get_window(kw.get('window', 'blackman'), False, L), kw['shift']))
        Sr = S.to(torch.complex128).requires_grad_(True)
        fr = torch.fft.irfft(Sr, n=kw['size'])[..., :L] * ws
        n = (9 - 1) * kw['shift'] + L
        out = torch.zeros(2, n, dtype=torch.float64)
        for t in range(9):
            out[:, t * kw['shift']:t * kw['shift'] + L] = out[:, t * kw['shift']:t * kw['shift'] + L] + fr[:, t]
        left, right = stft_np.fading_pad_width(L, kw['shift'], kw.get('fading', 'full'))
        out = out[:, left:n - right]
        np.testing.assert_allclose(y.detach().cpu().numpy(), out.detach().numpy(), atol=1e-5)
        (out * gy.double()).sum().backward()
        # torch's irfft backward treats DC/Nyquist imaginary parts like ours: zero gradient
        np.testing.assert_allclose(Sd.grad.cpu().numpy(), Sr.grad.numpy(),
                                   atol=2e-4 * Sr.grad.abs().max().item(), err_msg=str(kw))


def test_pit_features_vs_reference(g3):
    from padertorch_amd.ops import pit_features
    s = torch.from_numpy(g3['s']).to(DEV)
    y = torch.from_numpy(g3['y']).to(DEV)
    f = pit_features([y], [s])
    assert f['num_frames'] == [int(g3['num_frames'])]
    np.testing.assert_allclose(f['Y_abs'][0].cpu().numpy(), g3['Y_abs'], atol=2e-5)
    np.testing.assert_allclose(f['X_abs'][0].cpu().numpy(), g3['X_abs'], atol=2e-5)
    # cos(phase difference) is ill-conditioned where |Y| or |X| is ~0: weight the error by magnitude
    w = np.minimum(g3['Y_abs'][:, None, :], g3['X_abs'])
    err = np.abs(f['cos_phase_difference'][0].cpu().numpy() - g3['cos_phase_difference']) * w
    assert err.max() < 2e-5, err.max()


def test_pit_features_ragged_batch_vs_oracle():
    from padertorch_amd.ops import pit_features
    rng = np.random.RandomState(7)
    lens = [4000, 3500, 3499, 900]
    exs = [features_np.synthetic_mixture(rng, n) for n in lens]
    ref = [features_np.pre_batch_transform(s, y) for s, y in exs]
    f = pit_features([torch.from_numpy(y).to(DEV) for _, y in exs],
                     [torch.from_numpy(s).to(DEV) for s, _ in exs])
    assert f['num_frames'] == [r['num_frames'] for r in ref]
    for b, r in enumerate(ref):
        assert f['Y_abs'][b].shape == r['Y_abs'].shape and f['X_abs'][b].shape == r['X_abs'].shape
        np.testing.assert_allclose(f['Y_abs'][b].cpu().numpy(), r['Y_abs'], atol=2e-5)
        np.testing.assert_allclose(f['X_abs'][b].cpu().numpy(), r['X_abs'], atol=2e-5)
        w = np.minimum(r['Y_abs'][:, None, :], r['X_abs'])
        err = np.abs(f['cos_phase_difference'][b].cpu().numpy() - r['cos_phase_difference']) * w
        assert err.max() < 2e-5
    # padded frames are exactly zero
    assert f['Y_abs'].padded[3, f['num_frames'][3]:].abs().max().item() == 0


@pytest.mark.parametrize('size,shift,K', [(400, 160, 2), (100, 25, 3), (6, 2, 1)])
def test_pit_features_of_any_even_stft_size_vs_oracle(size, shift, K):
    """``paderbox.stft`` - and with it the reference's ``pre_batch_transform`` (``pit/data.py:52-75``) - takes any STFT size; sizes that
    are not powers of two in 64..2048 run the direct-DFT feature kernel (``pit_features_generic_kernel``): same outputs, the packed
    log-magnitude rows and planes included, ragged batch."""
    from torch.nn.utils.rnn import pack_sequence
    from padertorch_amd.ops import pit_features, STFT
    rng = np.random.RandomState(size)
    lens = [2000, 1711, 1710, 400]
    exs = [features_np.synthetic_mixture(rng, n, K=K) for n in lens]
    ref = [features_np.pre_batch_transform(s, y, size, shift) for s, y in exs]
    f = pit_features([torch.from_numpy(y).to(DEV) for _, y in exs], [torch.from_numpy(s).to(DEV) for s, _ in exs], stft=STFT(size, shift))
    assert f['num_frames'] == [r['num_frames'] for r in ref]
    for b, r in enumerate(ref):
        assert f['Y_abs'][b].shape == r['Y_abs'].shape and f['X_abs'][b].shape == r['X_abs'].shape
        np.testing.assert_allclose(f['Y_abs'][b].cpu().numpy(), r['Y_abs'], atol=2e-5)
        np.testing.assert_allclose(f['X_abs'][b].cpu().numpy(), r['X_abs'], atol=2e-5)
        w = np.minimum(r['Y_abs'][:, None, :], r['X_abs'])
        err = np.abs(f['cos_phase_difference'][b].cpu().numpy() - r['cos_phase_difference']) * w
        assert err.max() < 2e-5
    assert f['Y_abs'].padded[3, f['num_frames'][3]:].abs().max().item() == 0
    pk = f['Y_abs'].packed_log1p
    assert pk is not None
    want = pack_sequence([torch.log1p(a) for a in f['Y_abs']])
    assert torch.equal(pk.batch_sizes, want.batch_sizes)
    torch.testing.assert_close(pk.data, want.data, atol=1e-6, rtol=2e-7)
    planes = pk.planes()
    if planes is not None:
        rows, F = want.data.shape
        KB = (F + 31) // 32
        p = planes.view((rows + 15) // 16, KB, 2, 4, 16, 8).float()
        val = (p[:, :, 0] + p[:, :, 1]).permute(0, 3, 1, 2, 4).reshape(-1, KB * 32) / 512.
        torch.testing.assert_close(val[:rows, :F], pk.data, atol=2e-7, rtol=3e-7)
        assert float(val[:, F:].abs().max()) == 0.


@pytest.mark.parametrize('lens', [[4000, 3500, 3499, 900], [2048] * 32, [3000] * 5])
def test_pit_features_write_the_packed_log_magnitude_input(lens):
    """SURVEY row a9 (``pit/model.py:91-94``, ``ops/sequence/pointwise.py:37``): the feature kernel itself writes the first BLSTM
    layer's input - log1p(Y_abs) in PackedSequence order, fp32 and as fp16 (hi, lo) planes with the fixed scale 2^9 - equal to what
    pack_sequence + log1p (+ the pack pass) make of its Y_abs: rows bit-exact positions, values to an ulp of log1p, planes =
    hi + lo of 2^9 x within fp16's 22 bits, padding zero; two buffers per shape in turn - the NEXT call of the same shape (the following
    batch's features, made ahead of time by a prefetcher) leaves the planes alone, the one after it retires them."""
    from torch.nn.utils.rnn import pack_sequence
    from padertorch_amd.ops import pit_features
    from padertorch_amd.ops import gemm as G
    rng = np.random.RandomState(len(lens))
    exs = [features_np.synthetic_mixture(rng, n) for n in lens]
    ys = [torch.from_numpy(y).to(DEV) for _, y in exs]
    ss = [torch.from_numpy(s).to(DEV) for s, _ in exs]
    f = pit_features(ys, ss)
    pk = f['Y_abs'].packed_log1p
    assert pk is not None
    want = pack_sequence([torch.log1p(a) for a in f['Y_abs']])
    assert torch.equal(pk.batch_sizes, want.batch_sizes) and pk.data.shape == want.data.shape
    torch.testing.assert_close(pk.data, want.data, atol=1e-6, rtol=2e-7)
    planes = pk.planes()
    assert planes is not None
    rows, F = want.data.shape
    KB = (F + 31) // 32
    p = planes.view((rows + 15) // 16, KB, 2, 4, 16, 8).float()          # [row tile][k block][plane][k group][row][8]
    val = (p[:, :, 0] + p[:, :, 1]).permute(0, 3, 1, 2, 4).reshape(-1, KB * 32) / 512.        # [rows padded][k padded]
    torch.testing.assert_close(val[:rows, :F], pk.data, atol=2e-7, rtol=3e-7)
    assert float(val[rows:].abs().max() if val.shape[0] > rows else 0.) == 0. and float(val[:, F:].abs().max()) == 0.
    hi = p[:, :, 0].permute(0, 3, 1, 2, 4).reshape(-1, KB * 32)[:rows, :F]
    assert torch.equal(hi, (pk.data * 512.).half().float())              # the hi plane is THE fp16 rounding of 2^9 x
    # the first projection on these planes == on planes packed from the fp32 rows with the same scale word
    w = torch.randn(48, F, device=DEV) * 0.1
    word = G.scale_word(torch.device(DEV), 16.0)
    ya, yb = torch.empty(rows, 48, device=DEV), torch.empty(rows, 48, device=DEV)
    wp = G.pack_n(w)
    torch.ops.ptmi.gemm_planes_(ya, planes, word, wp[0], wp[1], None, rows, 48, F, False, 1)
    G.mm_planes_(yb, G.pack_n(pk.data, word), wp, rows, 48, F, split_k=1)
    assert torch.equal(ya, yb)
    f2 = pit_features(ys, ss)
    assert pk.planes() is planes and f2['Y_abs'].packed_log1p.planes() is not None
    assert f2['Y_abs'].packed_log1p.planes().data_ptr() != planes.data_ptr()
    torch.ops.ptmi.gemm_planes_(ya, planes, word, wp[0], wp[1], None, rows, 48, F, False, 1)      # still the first call's values
    assert torch.equal(ya, yb)
    f3 = pit_features(ys, ss)
    assert pk.planes() is None and f2['Y_abs'].packed_log1p.planes() is not None and f3['Y_abs'].packed_log1p.planes() is not None


def test_models_take_the_packed_log_magnitude_when_the_list_is_untouched():
    """The model consumes ``Y_abs.packed_log1p`` (no pack / log1p kernels) and gives the masks of the generic path bit for bit
    apart from the first projection's operand scale (2^9 instead of the measured one): compared at 1e-6."""
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    rng = np.random.RandomState(3)
    lens = [4000, 3300, 2100]
    exs = [features_np.synthetic_mixture(rng, n) for n in lens]
    torch.manual_seed(0)
    model = PermutationInvariantTrainingModel(F=257, recurrent_layers=2, units=24, K=2).to(DEV).eval()
    f = pt.ops.pit_features([torch.from_numpy(y).to(DEV) for _, y in exs], [torch.from_numpy(s).to(DEV) for s, _ in exs])
    with torch.no_grad():
        fused = model(f)
        plain = model(dict(Y_abs=list(f['Y_abs'])))          # a plain list: pack_sequence + log1p + measured scale
    for a, b in zip(fused, plain):
        torch.testing.assert_close(a, b, atol=1e-6, rtol=0)


@pytest.mark.gpu
def test_edited_feature_list_is_not_served_from_the_packed_log_magnitude():
    """anything that edits ``Y_abs`` between ``ops.pit_features`` and ``forward`` - an in-place gain on the padded
    buffer or on one example's view, a replaced list entry - must reach the model (the reference always packs the list it is
    given, pit/model.py:91-94): ``PackedLog1p.matches`` notices, and both models recompute from the list."""
    import padertorch_amd as pt
    from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
    from padertorch_amd.contrib.tcl.dc import DeepClusteringModel
    rng = np.random.RandomState(4)
    exs = [features_np.synthetic_mixture(rng, n) for n in (3000, 3000, 2400)]
    ys = [torch.from_numpy(y).to(DEV) for _, y in exs]
    torch.manual_seed(1)
    for model in (PermutationInvariantTrainingModel(F=257, recurrent_layers=1, units=16, K=2).to(DEV).eval(),
                  DeepClusteringModel(F=257, recurrent_layers=1, units=16, E=4, input_feature_transform='log1p').to(DEV).eval()):
        def run(edit):
            f = pt.ops.pit_features(ys)
            assert f['Y_abs'].packed_log1p.matches(f['Y_abs'])
            edit(f['Y_abs'])
            stale = not f['Y_abs'].packed_log1p.matches(f['Y_abs'])
            with torch.no_grad():
                got = model(dict(Y_abs=f['Y_abs']))
                want = model(dict(Y_abs=[t.clone() for t in f['Y_abs']]))      # plain list of the edited values: the generic path
            return stale, got, want

        def gain(lst): lst.padded.mul_(3.)
        def view_gain(lst): lst[1].mul_(0.25)
        def replace(lst): lst[2] = lst[2] * 2.
        for edit in (gain, view_gain, replace):
            stale, got, want = run(edit)
            assert stale, edit.__name__
            for a, b in zip(got, want):
                assert torch.equal(a, b), edit.__name__          # the same generic path on the same values: bit-identical
        stale, got, want = run(lambda lst: None)
        assert not stale


def test_pit_features_keep_a_nan_sample_like_the_reference():
    """A NaN sample in the mixture (or a source) makes exactly the frames that cover it NaN in |Y| (|X|), in the cosine of the phase
    difference and in the packed log-magnitude input - the oracle's pattern (``np.abs`` / ``np.angle``, reference ``pit/data.py:67-75``),
    so that the loss and with it ``Trainer``'s non-finite check (``trainer.py:622-636``) see it.  (Until round 5 the magnitude /
    phasor select tested ``|X|^2 > 0`` and turned a NaN bin into magnitude 0, phase 0: a finite loss on corrupt data.)"""
    from padertorch_amd.ops import pit_features
    rng = np.random.RandomState(3)
    exs = [features_np.synthetic_mixture(rng, n) for n in (4000, 3500)]
    exs[0][1][1000] = np.nan            # mixture of example 0
    exs[1][0][1, 2000] = np.nan         # source 1 of example 1
    ref = [features_np.pre_batch_transform(s, y) for s, y in exs]
    f = pit_features([torch.from_numpy(y).to(DEV) for _, y in exs], [torch.from_numpy(s).to(DEV) for s, _ in exs])
    for b, r in enumerate(ref):
        for key in ('Y_abs', 'X_abs', 'cos_phase_difference'):
            got = f[key][b].cpu().numpy()
            assert np.array_equal(np.isnan(got), np.isnan(r[key])), (b, key, int(np.isnan(got).sum()), int(np.isnan(r[key]).sum()))
            ok = ~np.isnan(r[key])
            if key != 'cos_phase_difference':
                np.testing.assert_allclose(got[ok], r[key][ok], atol=2e-5)
    assert np.isnan(ref[0]['Y_abs']).any() and np.isnan(ref[1]['X_abs']).any() and not np.isnan(ref[1]['Y_abs']).any()
    packed = f['Y_abs'].packed_log1p
    assert packed is not None and bool(torch.isnan(packed.data).any())
    # frame t of example b is packed row offs[t] + b: the NaN rows of the packed input are the NaN frames of example 0
    bs = packed.batch_sizes.numpy()
    offs = np.concatenate([[0], np.cumsum(bs)])[:-1]
    nan_rows = set(np.nonzero(torch.isnan(packed.data).any(1).cpu().numpy())[0].tolist())
    want_rows = {int(offs[t]) for t in np.nonzero(np.isnan(ref[0]['Y_abs']).any(1))[0]}
    assert nan_rows == want_rows


def test_pit_features_of_a_frame_below_the_normal_range_are_finite():
    """A last frame that holds ONE sample, under the window's first tap (a Blackman window's is ~1e-17, not 0): |X|^2 ~ 1e-38 lies below
    fp32's normal range, where the hardware's reciprocal square root flushes its operand - until round 6 the magnitude came out as inf
    and the cosine as NaN (3841 = 30 x 128 + 1 samples; found by a ragged batch of the training distribution, ``pit/data.py:20-33``).
    Magnitudes within the usual tolerance of the oracle, every value finite, the cosine of the tiny bins in [-1, 1]."""
    from padertorch_amd.ops import pit_features
    rng = np.random.RandomState(11)
    hit = 0
    for n in (3841, 3969, 4097, 2561):
        exs = [features_np.synthetic_mixture(rng, n), features_np.synthetic_mixture(rng, n - 700)]
        ref = [features_np.pre_batch_transform(s, y) for s, y in exs]
        f = pit_features([torch.from_numpy(y).to(DEV) for _, y in exs], [torch.from_numpy(s).to(DEV) for s, _ in exs])
        for b, r in enumerate(ref):
            for key in ('Y_abs', 'X_abs', 'cos_phase_difference'):
                got = f[key][b].cpu().numpy()
                assert np.isfinite(got).all(), (n, b, key)
                if key != 'cos_phase_difference':
                    np.testing.assert_allclose(got, r[key], atol=2e-5)
                else:
                    assert np.abs(got).max() <= 1. + 1e-5
            hit += int((r['Y_abs'][-1] < 1e-15).all())
        packed = f['Y_abs'].packed_log1p
        assert packed is not None and bool(torch.isfinite(packed.data).all())
    assert hit >= 4, hit         # (the cases do contain such frames)
