"""The C-ABI library loads and exports every symbol include/ptmi.h declares (no GPU needed)."""
import ctypes
import re
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (REPO / 'include' / 'ptmi.h').read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ptmi_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_header():
    from padertorch_amd import _lib
    from padertorch_amd.build import build
    build()
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in ptmi.h but not exported'
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)
    _lib.load()


def test_frame_bookkeeping_is_bit_exact(g1, g2):
    """ptmi_stft_num_frames / ptmi_istft_num_samples: integer parity with the reference."""
    from padertorch_amd import _lib
    from padertorch_amd.ops import STFT
    lib = _lib.load()
    for fc in g1['frame_counts']:
        st = STFT(fc['size'], fc['shift'], window_length=fc['window_length'], fading=fc['fading'])
        for n, fr in zip(fc['samples'], fc['frames']):
            assert lib.ptmi_stft_num_frames(st._geom, n) == fr == st.samples_to_frames(n)
    for c in g2['cases']:
        st = STFT(c['size'], c['shift'], window=c['window'], window_length=c['window_length'],
                  fading=c['fading'], pad=c['pad'])
        n = g2[c['x']].shape[-1]
        assert lib.ptmi_stft_num_frames(st._geom, n) == c['frames'] == st.samples_to_frames(n)
        assert lib.ptmi_istft_num_samples(st._geom, c['frames']) == c['samples_back'] \
            == st.frames_to_samples(c['frames'])


def test_no_cpu_fallback():
    import torch
    from padertorch_amd.ops import STFT, pit_loss
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        STFT(512, 128)(torch.zeros(2, 1000))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pit_loss(torch.zeros(4, 2, 5), torch.zeros(4, 2, 5), axis=1)


def test_stft_argument_checks():
    from padertorch_amd.ops import STFT
    with pytest.raises(AssertionError, match='even FFT sizes'):
        STFT(511, 128)
    with pytest.raises(AssertionError):
        STFT(512, 128, complex_representation='polar')
    with pytest.raises(AssertionError):
        STFT(512, 128, fading='quarter')


def test_kernels_are_registered_torch_ops_without_a_cpu_kernel():
    """The hot-path kernels are torch custom ops (torch.ops.ptmi.*) over the C ABI; they have a CUDA kernel only."""
    import torch
    import padertorch_amd  # noqa: F401  (registers the library)
    names = {'stft_forward', 'istft_forward', 'pit_features', 'pit_loss_forward', 'pit_loss_backward', 'dc_loss_forward',
             'dc_loss_backward', 'unit_norm_forward', 'unit_norm_backward', 'lstm_recurrence_forward',
             'lstm_recurrence_backward', 'absmax', 'gemm_planes_', 'pack_planes_n', 'pack_planes_t', 'pit_features_packed'}
    for n in names:
        op = getattr(torch.ops.ptmi, n)
        assert op.default._schema.name == f'ptmi::{n}'
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f'ptmi::{n}', 'CUDA')
        assert not torch._C._dispatch_has_kernel_for_dispatch_key(f'ptmi::{n}', 'CPU')
    assert torch.ops.ptmi.gemm_planes_.default._schema.arguments[0].alias_info.is_write     # out is written in place
    with pytest.raises(NotImplementedError):
        torch.ops.ptmi.absmax(torch.ones(2, 2), 2, 2, 2)                                    # CPU tensor: no kernel, no fallback
