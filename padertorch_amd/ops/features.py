"""On-device PIT feature front-end.

Replaces the numpy data-pipeline transform ``pre_batch_transform``
(``padertorch/contrib/examples/source_separation/pit/data.py:49-77``: ``stft(s,512,128)``,
``stft(y,512,128)``, ``|.|`` and ``cos(angle(Y) - angle(X))``) by one fused HIP kernel
(``ptmi_pit_features``) that turns the raw waveforms of a whole batch into the model's inputs
directly in HBM - the H2D copy shrinks from (1+2K) feature maps to (1+K) waveforms
 and the STFTs never leave the device.
"""
import numpy as np
import torch

from .. import _lib
from . import library  # noqa: F401  (registers torch.ops.ptmi.*)
from ._stft import STFT
from .sequence.pack_module import PaddedList

__all__ = ['pit_features', 'PackedLog1p']

_default_stft = None

#: fp16 operand scale of the log-magnitude planes: 2^9 (log1p(FLT_MAX) 2^9 < 65504: no input can overflow); as the float whose
#: exponent makes ptmi_gemm_planes take that scale (2^13 / 16)
LOG1P_SCALE_WORD_VALUE = 16.0

_PLANES = {}        # (device, rows, F, stream) -> [zero-initialised planes buffer, generation]


class PackedLog1p:
    """``log1p(Y_abs)`` as the ``PackedSequence`` the model builds in ``pit/model.py:91-94`` (``data [rows, F]`` fp32,
    ``batch_sizes``), written by the feature kernel itself, plus the same matrix as fp16 planes for the first input projection.
    The planes live in a buffer that the next ``pit_features`` call of the same shape overwrites: :meth:`planes` returns them
    only while they are still this call's (otherwise the consumer packs ``data`` itself)."""

    def __init__(self, data, batch_sizes, planes_entry, generation):
        self.data = data
        self.batch_sizes = batch_sizes
        self._entry = planes_entry
        self._generation = generation
        self._source = None

    def bind(self, source: PaddedList):
        """Remember what ``source`` (the ``Y_abs`` list this was computed with) looks like right now: its buffer, the version
        counter the buffer shares with all its views, its lengths."""
        self._source = (source.padded.data_ptr(), source.padded._version, tuple(source.padded.shape), tuple(source.lengths),
                        self.data._version)
        return self

    def matches(self, source) -> bool:
        """Is ``source`` still the list this was computed from, element for element and value for value?  False after an
        in-place edit of the padded buffer or of any of its views (they bump the shared version counter), after a list entry
        was replaced or the list re-ordered, after a device move, or after an in-place edit of ``data``: the consumer then
        packs ``source`` and takes ``log1p`` itself, as the reference always does (``pit/model.py:91-94``)."""
        if self._source is None or not isinstance(source, PaddedList) or not source.batch_first:
            return False
        ptr, version, shape, lengths, data_version = self._source
        pad = source.padded
        if (pad.data_ptr() != ptr or pad._version != version or tuple(pad.shape) != shape or tuple(source.lengths) != lengths
                or self.data._version != data_version or len(source) != len(lengths)):
            return False
        step = pad.stride(0) * pad.element_size()
        return all(torch.is_tensor(v) and v.data_ptr() == ptr + b * step and v.shape[0] == n and v._version == version
                   and v.stride() == pad.stride()[1:] for b, (v, n) in enumerate(zip(source, lengths)))

    def planes(self):
        if self._entry is not None and self._entry[1] == self._generation:
            return self._entry[0]
        return None


def _pit_features_device_lengths(y, s, ns_dev, frames_dev, stft):
    """:func:`pit_features` for a padded batch whose example lengths are DEVICE data (``ns_dev`` samples, ``frames_dev`` frames per
    example, both int32 ``[B]``): the launch depends on the padded shape only."""
    assert ns_dev.dtype == torch.int32 and ns_dev.is_cuda and frames_dev is not None and frames_dev.dtype == torch.int32, \
        'device-side lengths: num_samples and num_frames_dev as int32 device tensors'
    B, N = y.shape
    if s is not None:
        assert s.dim() == 3 and s.shape[0] == B and s.shape[2] == N and s.dtype == torch.float32
        s = s.contiguous()
    T = int(_lib.load().ptmi_stft_num_frames(stft._geom, N))
    tb = stft._tables.get(y.device)
    g = stft._geom
    geom = [g.size, g.shift, g.window_length, g.pad_left, g.pad_right, g.pad]
    Y_abs, X_abs, cos_pd = torch.ops.ptmi.pit_features(y, s, ns_dev, tb['window'], tb['twiddle'], geom, T)
    frames = [T] * B
    out = dict(Y_abs=PaddedList(Y_abs, frames, True, frames_dev), num_frames=frames_dev)
    out['Y_abs'].packed_log1p = None
    if s is not None:
        out['X_abs'] = PaddedList(X_abs, frames, True, frames_dev)
        out['cos_phase_difference'] = PaddedList(cos_pd, frames, True, frames_dev)
    return out


def _pad_rows(rows, dim_last_pad_to):
    out = rows[0].new_zeros((len(rows),) + tuple(rows[0].shape[:-1]) + (dim_last_pad_to,))
    for b, r in enumerate(rows):
        out[b, ..., :r.shape[-1]] = r
    return out


def pit_features(y, s=None, num_samples=None, stft: STFT = None, packed_log1p=True, num_frames_dev=None):
    """Waveforms -> ``dict(Y_abs, X_abs, cos_phase_difference, num_frames)`` (model batch contract).

    Args:
        y: mixtures: list of ``(N_b,)`` tensors (sorted by descending length) or a padded ``[B, N]``
        s: sources: list of ``(K, N_b)`` tensors or a padded ``[B, K, N]`` (None: only ``Y_abs``)
        num_samples: list of ints when ``y`` / ``s`` are padded and ragged
        stft: an :class:`STFT` (default ``STFT(512, 128)`` = paderbox defaults used by the example)
        packed_log1p: also write ``log1p(Y_abs)`` in PackedSequence order (``Y_abs.packed_log1p``: :class:`PackedLog1p`), the
            first BLSTM layer's input of both example models, so that no pack / log1p / scale / split pass follows the kernel
        num_frames_dev: with ``num_samples`` as an int32 DEVICE tensor ``[B]`` (a batch whose length pattern is device data,
            ``ops.sequence.StaticSlots``): the examples' frame counts as an int32 device tensor; the host then knows only the padded
            shapes - every list entry spans all ``T`` frames of the padded tensors (frames past an example's own count are zero),
            ``lengths_dev`` carries the true counts to the kernels that need them (the losses)
    Every entry of the result is a :class:`PaddedList` (list of per-example views, e.g.
    ``Y_abs[b]: (T_b, F)``, ``X_abs[b]: (T_b, K, F)``).
    """
    global _default_stft
    if stft is None:
        if _default_stft is None:
            _default_stft = STFT(512, 128)
        stft = _default_stft
    if isinstance(y, (list, tuple)):
        num_samples = [int(t.shape[-1]) for t in y]
        y = _pad_rows(list(y), max(num_samples))
        if s is not None:
            s = _pad_rows(list(s), max(num_samples))
    _lib.require_gpu(y, s)
    assert y.dim() == 2 and y.dtype == torch.float32, (y.shape, y.dtype)
    y = y.contiguous()
    B, N = y.shape
    if torch.is_tensor(num_samples):
        return _pit_features_device_lengths(y, s, num_samples, num_frames_dev, stft)
    K = 0
    if s is not None:
        assert s.dim() == 3 and s.shape[0] == B and s.shape[2] == N and s.dtype == torch.float32
        s = s.contiguous()
        K = s.shape[1]
    if num_samples is None:
        num_samples = [N] * B
    lib = _lib.load()
    per_len = {n: int(lib.ptmi_stft_num_frames(stft._geom, n)) for n in set(num_samples)}
    frames = [per_len[n] for n in num_samples]
    T = max(frames)
    F = stft.size // 2 + 1
    dev = y.device
    ragged = any(n != N for n in num_samples)
    ns_dev = _lib.host_to_device(num_samples, torch.int32, dev) if ragged else None
    tb = stft._tables.get(dev)
    g = stft._geom
    geom = [g.size, g.shift, g.window_length, g.pad_left, g.pad_right, g.pad]
    packed = None
    if packed_log1p and T > 0 and all(a >= b for a, b in zip(frames, frames[1:])):
        # the model's first-layer input on the way out: log1p(Y_abs) in PackedSequence order, fp32 and as fp16 planes
        from . import gemm as _gemm, lstm as _lstm
        # (numpy, not torch: a CPU tensor op over T x B elements wakes torch's whole intra-op thread pool - tens of ms on a 256-core host)
        bs = torch.from_numpy((np.asarray(frames)[None, :] > np.arange(T)[:, None]).sum(1).astype(np.int64)) if ragged else \
            torch.full((T,), B, dtype=torch.int64)
        meta = _lstm.pack_meta(bs, dev)
        lp = torch.empty((meta.rows, F), dtype=torch.float32, device=dev)
        entry = None
        if _gemm.planes_enabled():
            # (one buffer per shape AND stream: a call on another stream - a prefetching copy stream, a second host thread - must not
            #  overwrite planes that a GEMM queued on this stream has not read yet)
            from . import capture as _capture
            key = (dev.type, dev.index, meta.rows, F, torch.cuda.current_stream(dev).cuda_stream)
            if _capture.ACTIVE:
                # a captured step runs on the capture's own stream: a ring made here would be allocated - and zero-filled - by the graph,
                # i.e. twice 8 MB of fills at every replay.  The ring of this shape that the warm-up steps made (any stream) serves: the
                # replays are ordered on one stream, nobody else touches it while the graph lives.
                for k in _PLANES:
                    if k[:4] == key[:4]:
                        key = k
                        break
            # TWO buffers in turn: a call retires the planes of the call before the previous one, so the features of the NEXT batch can be
            # made (data.DevicePrefetcher with a model's example_to_device) while this batch's first projection has yet to read its planes
            ring = _PLANES.get(key)
            if ring is None:
                if len(_PLANES) > 8:
                    _PLANES.clear()
                ring = _PLANES[key] = [[[torch.zeros(int(lib.ptmi_planes_elems(meta.rows, F)), dtype=torch.float16, device=dev), 0]
                                        for _ in range(2)], 0]
            ring[1] ^= 1
            entry = ring[0][ring[1]]
            entry[1] += 1
        Y_abs, X_abs, cos_pd = torch.ops.ptmi.pit_features_packed(
            y, s, ns_dev, tb['window'], tb['twiddle'], geom, T, lp, None if entry is None else entry[0],
            meta.offs_dev if ragged else None)
        packed = PackedLog1p(lp, bs, entry, None if entry is None else entry[1])
    else:
        Y_abs, X_abs, cos_pd = torch.ops.ptmi.pit_features(y, s, ns_dev, tb['window'], tb['twiddle'], geom, T)
    fl = _lib.host_to_device(frames, torch.int32, dev) if ragged else None
    out = dict(Y_abs=PaddedList(Y_abs, frames, True, fl), num_frames=frames)
    #: consumed by PermutationInvariantTrainingModel.forward / DeepClusteringModel.forward when the list is handed on untouched
    out['Y_abs'].packed_log1p = None if packed is None else packed.bind(out['Y_abs'])
    if K:
        out['X_abs'] = PaddedList(X_abs, frames, True, fl)
        out['cos_phase_difference'] = PaddedList(cos_pd, frames, True, fl)
    return out
