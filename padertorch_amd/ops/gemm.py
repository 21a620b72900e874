"""fp32 matrix products on the fp16 matrix cores of the MI355X (``csrc/gemm_planes.hip``).

The dense layers of the hot path - the LSTM input projections and ``torch.nn.Linear`` layers of
``padertorch/contrib/examples/source_separation/pit/model.py:60-66,97-104`` / ``contrib/tcl/dc.py:32-40`` and
their input / weight gradients - are ``mm(x, y)`` calls here: fp32 tensors in, fp32 tensor out, every
product evaluated as three fp16 products of the operands' (hi, lo) halves with fp32 accumulation, which
is as close to the fp64 result as an exact fp32 GEMM (``tests/test_gpu_gemm.py``) at several times its
speed (fp32 MFMA runs at 1/16 of the fp16 rate on this part).

``PRODUCTS = 1`` multiplies the hi planes only - plain 16-bit operands (fp16 for activations and weights, bf16 for the gate
gradients the backward recurrence hands on), fp32 accumulation: BASELINE's "bf16" run of the model, reduced precision, reported
as a delta, never the default.
"""
import os
import threading
import weakref

import torch

from .. import _lib
from . import library  # noqa: F401  (registers torch.ops.ptmi.*)

__all__ = ['mm', 'absmax', 'weight_absmax', 'UNIT_RANGE', 'PRODUCTS', 'ENABLED', 'usable']

#: False: the dense layers go back to the BLAS library (torch.addmm / F.linear): the round-1 path, kept for A/B runs
ENABLED = True

#: 3 = split 16-bit halves (fp32-equivalent), 1 = the hi halves only (plain 16-bit operands)
PRODUCTS = 3

#: pass as ``amax_x`` / ``amax_y`` for operands known to lie in a range fp16 covers as it is (activations in
#: [-1, 1], log-magnitude features): no scale, no reduction pass
UNIT_RANGE = 'unit'

#: Caches of values derived from parameters (operand scales, packed planes).  Keys are ``id(parameter)``; every entry also holds a
#: WEAK reference to the parameter it was made from and is valid only while that very object is alive, its ``_version`` and its
#: storage pointer are unchanged: a freed model's ids and device pointers can be reused by the next one (checkpoints evaluated in
#: sequence), and then version and pointer alone would match.  Writing through ``p.data`` does not bump ``_version``: call
#: :func:`invalidate` after such an edit.
_WEIGHT_AMAX = {}


def _refs(params):
    return tuple(weakref.ref(p) for p in params)


def _same(refs, params):
    return refs is not None and len(refs) == len(params) and all(r() is p for r, p in zip(refs, params))


#: device -> (event, {id(parameter): (weak reference, version)}): the kernel that last rewrote these parameters in place
#: (:func:`note_update`)
_UPDATED = {}


def note_update(params):
    """An optimizer has just rewritten ``params`` in place on the current stream (and bumped their versions): remember an event
    behind that kernel.  Work that only reads these parameters - their operand forms of the next step, made on a side stream -
    then waits for THIS event instead of for everything the main stream has queued since (:func:`update_event`)."""
    params = [p for p in params if p.is_cuda]
    if not params:
        return
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(params[0].device))
    _UPDATED[params[0].device] = (ev, {id(p): (weakref.ref(p), p._version) for p in params})


def update_event(params):
    """The event of :func:`note_update` if every one of ``params`` was written by that kernel and by nothing since (same
    objects, same versions); else ``None`` (wait for the stream)."""
    hit = _UPDATED.get(params[0].device) if params else None
    if hit is None:
        return None
    for p in params:
        e = hit[1].get(id(p))
        if e is None or e[0]() is not p or e[1] != p._version:
            return None
    return hit[0]


def invalidate():
    """Drop every cached parameter-derived value (after in-place edits that bypass autograd's version counter)."""
    _WEIGHT_AMAX.clear()
    _WEIGHT_PLANES.clear()
    _UPDATED.clear()
    _KNOWN.clear()
    from . import lstm as _lstm
    _lstm._STACKED.clear()


class _Words(threading.local):
    """Zeroed device words handed out one by one (views into blocks of 64: ONE fill launch per 64 words; a block lives as long as a view
    of it does) - the targets of the epilogues that leave an operand maximum behind (``gemm_planes_relu_``, ``relu_backward_absmax``)."""
    def __init__(self):
        self.blocks = {}


_WORDS = _Words()


def zero_word(device):
    from . import capture as _capture
    if _capture.ACTIVE:                 # a captured step: words the graph itself zeroes at every replay
        return _capture.zero_word(device)
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ent = _WORDS.blocks.get(key)
    if ent is None or ent[1] >= 64:
        ent = _WORDS.blocks[key] = [torch.zeros(64, dtype=torch.int32, device=device), 0]
    ent[1] += 1
    return ent[0][ent[1] - 1:ent[1]]


def absmax(x):
    """Device word (int32 tensor [1]) holding the float bits of ``max |x|`` of a 2-D fp32 tensor with one unit stride."""
    assert x.dim() == 2 and x.dtype == torch.float32
    _lib.require_gpu(x)
    if x.stride(1) == 1 or x.shape[1] == 1:
        rows, cols, ld = x.shape[0], x.shape[1], x.stride(0)
    else:
        assert x.stride(0) == 1, x.stride()
        rows, cols, ld = x.shape[1], x.shape[0], x.stride(1)
    return torch.ops.ptmi.absmax(x, rows, cols, max(ld, cols))


#: id(parameter) -> (weak reference, forms this parameter has been asked for: 'n' / 't' / ('h', ndir, H, cols)): what
#: :func:`prefetch_known` re-makes after an optimizer step
_KNOWN = {}
_WAITED = {}                # stream -> the prefetch event it has waited for last


class _Prefetching(threading.local):
    """The event new cache entries carry while :func:`prefetch_known` fills the caches on a side stream - per host thread (a model
    per thread: another thread's ordinary cache fills must not be tagged with this thread's prefetch event)."""
    event = None

    def __getitem__(self, _):
        return self.event

    def __setitem__(self, _, value):
        self.event = value


_PREFETCHING = _Prefetching()


def _cached(cache, key, p, make, limit, form=None):
    """``make()`` cached per ``key`` until the parameter ``p`` is modified (version, storage, identity).  Entries made by
    :func:`prefetch_known` on a side stream carry an event: a hit makes the current stream wait for it (nothing, once it has
    fired) and keeps the tensors alive for that stream."""
    hit = cache.get(key)
    if hit is not None and hit[0] == p._version and hit[1] == p.data_ptr() and _same(hit[3], (p,)):
        if hit[4] is not None and hit[4] is not _PREFETCHING[0]:
            cur = torch.cuda.current_stream(p.device)
            if _WAITED.get(cur.cuda_stream) is not hit[4]:       # one wait per stream and prefetch (a barrier packet each otherwise)
                cur.wait_event(hit[4])
                _WAITED[cur.cuda_stream] = hit[4]
            # (no record_stream: the tensors' memory belongs to the preparation stream's pool, whose every piece of work is
            # enqueued behind the optimizer kernel and so behind every reader - see ops.lstm._stacked_weights)
        return hit[2]
    if len(cache) > limit:
        cache.clear()
    with torch.no_grad():
        v = make()
    cache[key] = (p._version, p.data_ptr(), v, _refs((p,)), _PREFETCHING[0])
    if form is not None:
        known = _KNOWN.get(id(p))
        if known is None or known[0]() is not p:
            if len(_KNOWN) > 256:
                _KNOWN.clear()
            known = _KNOWN[id(p)] = (weakref.ref(p), set())
        known[1].add(form)
    return v


def prefetch_known(device, everything=False):
    """Re-make, on the CURRENT stream (a side stream ordered behind the optimizer kernel: ``ops.lstm.packed_lstm``), the operand
    forms - maximum, fp16 planes in either orientation - that the dense layers asked for in earlier steps and that an optimizer step
    has made stale: a handful of small launches (~5 us each, 45 us per step of the PIT model) that otherwise sit in front of the
    first use on the main stream, between the last recurrence and the loss."""
    todo = []
    # only parameters the optimizer kernel behind this stream's wait has written, at the version it left them with: anything else
    # (a parameter another writer has touched since - weight clipping, an EMA copy-in - or one this event does not cover) is packed
    # lazily on its consumer's stream, behind its writer
    updated = _UPDATED.get(device)
    covered = updated[1] if updated is not None else {}
    for pid, (ref, forms) in list(_KNOWN.items()):
        q = ref()
        if q is None:
            del _KNOWN[pid]
        elif q.device == device and q.dim() == 2:
            e = covered.get(id(q))
            # (everything: a captured step - the stream forks at the head of the graph, behind the previous replay's optimizer kernel)
            if everything or (e is not None and e[0]() is q and e[1] == q._version):
                todo.append((q, sorted(forms, key=str)))
    if not todo or _PREFETCHING[0] is not None:
        return
    ev = _PREFETCHING[0] = torch.cuda.Event()
    try:
        for q, forms in todo:
            for form in forms:
                if form == 'n':
                    weight_planes(q)
                elif form == 't':
                    weight_planes_t(q)
                else:
                    weight_planes_h(q, *form[1:])
    finally:
        _PREFETCHING[0] = None
        ev.record(torch.cuda.current_stream(device))


def weight_absmax(p):
    """``absmax`` of a parameter, cached until the parameter is modified in place (optimizer step)."""
    return _cached(_WEIGHT_AMAX, id(p), p, lambda: absmax(p.detach() if p.dim() == 2 else p.detach().reshape(-1, p.shape[-1])), 256)


def weights_absmax(params):
    """``absmax`` over several parameters that are used as ONE operand (the two directions' ``weight_ih`` stacked)."""
    params = tuple(params)
    if len(params) == 1:
        return weight_absmax(params[0])
    key = tuple(id(p) for p in params)
    sig = tuple((p._version, p.data_ptr()) for p in params)
    hit = _WEIGHT_AMAX.get(key)
    if hit is not None and hit[0] == sig and _same(hit[3], params):
        return hit[2]
    v = weight_absmax(params[0])
    for p in params[1:]:
        v = torch.maximum(v, weight_absmax(p))       # non-negative floats order like their bit patterns
    _WEIGHT_AMAX[key] = (sig, None, v, _refs(params))
    return v


def seed_weights_absmax(params, word):
    """Enter an ``absmax`` word computed elsewhere (``ptmi_lstm_weight_prep``) for ``params`` used as one operand."""
    params = tuple(params)
    if len(_WEIGHT_AMAX) > 256:
        _WEIGHT_AMAX.clear()
    if len(params) == 1:
        p = params[0]
        _WEIGHT_AMAX[id(p)] = (p._version, p.data_ptr(), word, _refs((p,)), None)
    else:
        _WEIGHT_AMAX[tuple(id(p) for p in params)] = (tuple((p._version, p.data_ptr()) for p in params), None, word, _refs(params))


#: weight-gradient GEMMs (both operands reduce over their OUTER axis) on pre-split fp16 planes (csrc/gemm_planes.hip)
PLANES = True


def planes_enabled():
    return ENABLED and PLANES


def pack_t(x, amax=None):
    """``x [k, c]`` (fp32 CUDA, unit inner stride; any row stride) -> ``(planes, amax word or None)`` of the operand whose
    rows are ``x``'s columns and whose reduction axis is ``k`` (``torch.ops.ptmi.pack_planes_t``).  ``amax``: the device
    word from :func:`absmax` (measured here when ``None``) or :data:`UNIT_RANGE`."""
    # (a single column has no inner stride to speak of: torch leaves whatever stride it had, .contiguous() included)
    assert x.dim() == 2 and (x.stride(1) == 1 or x.shape[1] == 1) and x.dtype == torch.float32, (x.shape, x.stride(), x.dtype)
    ax = None if amax is UNIT_RANGE else (absmax(x) if amax is None else amax)
    return torch.ops.ptmi.pack_planes_t(x, ax), ax


def pack_n(x, amax=None):
    """``x [r, k]`` (reduction axis contiguous) -> ``(planes, amax word or None)`` of that ``r x k`` operand."""
    assert x.dim() == 2 and (x.stride(1) == 1 or x.shape[1] == 1) and x.dtype == torch.float32, (x.shape, x.stride(), x.dtype)
    ax = None if amax is UNIT_RANGE else (absmax(x) if amax is None else amax)
    return torch.ops.ptmi.pack_planes_n(x, ax), ax


def mm_planes_(out, a, b, M, N, K, accumulate=False, split_k=None, bias=None):
    """``out [M, N] (+)= A B^T`` for operands ``a = (planes, amax)``, ``b = (planes, amax)`` from :func:`pack_t` (``A`` is
    ``M x K``, ``B`` is ``N x K``): ``torch.ops.ptmi.gemm_planes_``."""
    assert out.dim() == 2 and out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32, (out.shape, out.stride())
    sk = auto_split_k(M, N, K) if split_k is None else int(split_k)
    torch.ops.ptmi.gemm_planes_(out, a[0], a[1], b[0], b[1], bias, M, N, K, bool(accumulate), sk)
    return out


_WEIGHT_PLANES = {}


def weight_planes(p):
    """``pack_n`` of a 2-D parameter used as the ``W`` of ``x W^T``, cached until the parameter is modified."""
    return _cached(_WEIGHT_PLANES, id(p), p, lambda: pack_n(p.detach(), weight_absmax(p)), 64, 'n')


def weight_planes_t(p):
    """``pack_t`` of a 2-D parameter ``W [out, in]`` used as the right operand of ``g W`` (rows = input features, reduction
    over the outputs), cached until the parameter is modified."""
    return _cached(_WEIGHT_PLANES, ('t', id(p)), p, lambda: pack_t(p.detach(), weight_absmax(p)), 64, 't')


_SCALE_WORDS = {}


def scale_word(device, value=8.0):
    """Device word whose float value makes ``ptmi_gemm_planes`` take an operand scale of ``2^13 / value``: 8.0 for the forward
    recurrence's hand-off planes, which hold fp16 halves of ``2^10 h``."""
    key = (device.type, device.index, float(value))
    if key not in _SCALE_WORDS:
        _SCALE_WORDS[key] = torch.tensor([value], dtype=torch.float32, device=device).view(torch.int32)
    return _SCALE_WORDS[key]


def pad_direction_blocks(w, ndir, H, cols):
    """``w [n, ndir * H]`` -> ``[n, ndir * cols]``: every direction's ``H`` input columns followed by ``cols - H`` zero columns
    (the k layout of the forward recurrence's hand-off planes, ``ptmi_lstm_handoff_cols``)."""
    out = w.new_zeros((w.shape[0], ndir, cols))
    out[:, :, :H] = w.reshape(w.shape[0], ndir, H)
    return out.view(w.shape[0], ndir * cols)


def pack_n_direction_blocks(w, ndir, H, cols, amax):
    """``pack_n(pad_direction_blocks(w, ndir, H, cols), amax)[0]`` without the padded fp32 copy: every direction's column block is
    packed into its k blocks of the wider operand (``torch.ops.ptmi.pack_planes_into_``)."""
    assert w.dim() == 2 and w.stride(1) == 1 and w.shape[1] == ndir * H and w.dtype == torch.float32, (w.shape, w.stride())
    if cols % 32:
        return pack_n(pad_direction_blocks(w, ndir, H, cols), amax)[0]
    n = w.shape[0]
    out = torch.empty((n + 15) // 16 * (ndir * cols // 32) * 1024, dtype=torch.float16, device=w.device)
    for d in range(ndir):
        torch.ops.ptmi.pack_planes_into_(out, w[:, d * H:(d + 1) * H], amax, False, ndir * cols // 32, d * cols // 32, cols // 32)
    return out


def weight_planes_h(p, ndir, H, cols):
    """``pack_n`` of a 2-D parameter ``W [out, ndir * H]`` with its input columns laid out like the hand-off planes
    (:func:`pad_direction_blocks`), cached until the parameter is modified."""
    def make():
        amax = weight_absmax(p)
        return pack_n_direction_blocks(p.detach(), ndir, H, cols, amax), amax
    return _cached(_WEIGHT_PLANES, ('h', id(p), cols), p, make, 64, ('h', ndir, H, cols))


def stacked_planes_t_bf16(w, ndir, cols, key_params=None):
    """bf16 planes of the right operand of ``dgates @ w`` for ``w [ndir * G, I]`` (the two directions' ``weight_ih`` stacked)
    when ``dgates`` is taken from the backward recurrence's hand-off planes, whose k axis has ``cols >= G`` columns per
    direction (``ptmi_lstm_handoff_cols``; the surplus is zero there): rows = input features, k = direction-major columns.
    Cached per parameter version when ``key_params`` (the Parameters ``w`` was built from) are given."""
    key = sig = None
    if key_params is not None:
        key = ('tb', cols) + tuple(id(q) for q in key_params)
        sig = tuple((q._version, q.data_ptr()) for q in key_params)
        hit = _WEIGHT_PLANES.get(key)
        if hit is not None and hit[0] == sig and _same(hit[3], tuple(key_params)):
            return hit[2]
        if len(_WEIGHT_PLANES) > 64:
            _WEIGHT_PLANES.clear()
    G = w.shape[0] // ndir
    with torch.no_grad():
        if cols != G and cols % 32 == 0 and w.stride(1) == 1:
            # every direction's rows into its k blocks of the wider operand (no padded fp32 copy)
            planes = torch.empty((w.shape[1] + 15) // 16 * (ndir * cols // 32) * 1024, dtype=torch.bfloat16, device=w.device)
            for d in range(ndir):
                torch.ops.ptmi.pack_planes_into_(planes, w.detach()[d * G:(d + 1) * G], None, True, ndir * cols // 32, d * cols // 32,
                                                 cols // 32)
        else:
            if cols != G:
                wp = w.new_zeros((ndir, cols, w.shape[1]))
                wp[:, :G] = w.view(ndir, G, -1)
                w = wp.view(ndir * cols, -1)
            planes = torch.ops.ptmi.pack_planes_bf16(w.detach().contiguous(), True)
    if key is not None:
        _WEIGHT_PLANES[key] = (sig, None, planes, _refs(key_params))
    return planes


def usable(*tensors):
    """The split GEMM applies: enabled, fp32 CUDA operands."""
    return ENABLED and all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


def auto_split_k(M, N, K):
    """The MOST K ranges a call may use (what its slab workspace is sized for): weight-gradient shapes - few output tiles, K = all
    rows of the batch - may be cut into up to 12; how many ranges and which tile run is the cost model's choice in
    ``csrc/gemm_planes.hip`` (``pick_big``: e.g. 2400 x 1200 x 8096 -> 76 tiles of 128 x 320 x 3 ranges = one round of the CUs, 127 us
    against 148 on round 3's 128 x 128 slabs; x 32192 -> 40 tiles of 256 x 320 x 6 ranges, 448 against 531 us), a function of the
    shape alone.  Shapes with many output tiles run without a split."""
    tiles = -(-M // 128) * -(-N // 128)
    if tiles >= 384:
        return 1
    return max(1, min(12, K // 256))


#: weight-gradient GEMMs that run BESIDE a persistent recurrence: up to this many reduction rows on the 128 x 128 kernel, whose
#: workgroups share a CU with the recurrence's (``co_resident_split_k``); longer ones on big tiles on the CUs the recurrence leaves free.
#: 0 since the end of round 4 (every such GEMM on the big tiles): with the recurrences' step time down to 2.3 / 3.6 us the
#: weight-gradient queue, not the recurrence, ends the backward phase, and workgroups that share CUs with a recurrence slow it by more than
#: they gain (B = 32, 8 kHz: 6.645 -> 6.60 ms per step; until then 16384: 7.15 against 7.18; ``profiles/r4_ab_step.txt``)
CO_RESIDENT_MAX_K = 0


def co_resident_split_k(M, N, K):
    """``split_k`` of a weight-gradient GEMM that runs next to a recurrence (``ptmi_gemm_planes``: negative = exactly that many k ranges
    on the 128 x 128 kernel, measured: until about two workgroups per CU exist, every range at least 16 k steps)."""
    if K > CO_RESIDENT_MAX_K:
        return auto_split_k(M, N, K)
    tiles = -(-M // 128) * -(-N // 128)
    if tiles >= 1024:
        return 1
    return -max(1, min(512 // tiles, K // 512, 8))


def mm(x, y, bias=None, out=None, accumulate=False, amax_x=None, amax_y=None, split_k=None):
    """``out (+)= x @ y + bias`` for fp32 CUDA tensors ``x [M, K]``, ``y [K, N]`` with one unit stride each (transposed views
    and column blocks of wider matrices are consumed in place): either operand is split into planes by the pack pass of its
    storage order (``pack_planes_n``: reduction axis contiguous, ``pack_planes_t``: reduction axis outermost), then
    ``gemm_planes_``.

    ``amax_x`` / ``amax_y``: device words from :func:`absmax` / :func:`weight_absmax` over the operand (computed here
    when ``None``), or :data:`UNIT_RANGE`.  ``split_k``: K ranges per output tile (default: :func:`auto_split_k`).
    """
    assert x.dim() == 2 and y.dim() == 2 and x.shape[1] == y.shape[0], (x.shape, y.shape)
    assert x.dtype == y.dtype == torch.float32, (x.dtype, y.dtype)
    _lib.require_gpu(x, y, bias, out)
    M, K = x.shape
    N = y.shape[1]
    if out is None:
        assert not accumulate
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    assert out.shape == (M, N) and (out.stride(1) == 1 or N == 1) and out.dtype == torch.float32, (out.shape, out.stride())
    if M == 0 or N == 0:
        return out
    if K == 0:
        if not accumulate:
            out.zero_()
            if bias is not None:
                out += bias
        return out
    if bias is not None:
        assert bias.shape == (N,) and bias.is_contiguous() and bias.dtype == torch.float32

    def planes(t, reduce_first, amax):
        # t's reduction axis is axis 0 (y) or axis 1 (x); the operand's rows are the other axis
        red, other = (0, 1) if reduce_first else (1, 0)
        if not ((t.stride(red) == 1 or t.shape[red] == 1) or (t.stride(other) == 1 or t.shape[other] == 1)):
            t = t.contiguous()
        if t.stride(red) == 1 or t.shape[red] == 1:         # reduction axis contiguous: rows of the operand = rows in memory
            src = t.t() if reduce_first else t
            if src.stride(1) != 1:
                src = src.contiguous()
            return pack_n(src, amax)
        src = t if reduce_first else t.t()                  # [k, c] with unit inner stride
        if src.stride(1) != 1:
            src = src.contiguous()
        return pack_t(src, amax)

    a = planes(x, False, amax_x)
    b = planes(y, True, amax_y)
    return mm_planes_(out, a, b, M, N, K, accumulate=accumulate, split_k=split_k, bias=bias)
