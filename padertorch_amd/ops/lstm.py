"""Packed-sequence (B)LSTM on MI355X: drop-in compute path for ``torch.nn.LSTM(PackedSequence)``.

``packed_lstm(lstm, packed)`` evaluates a ``torch.nn.LSTM`` module (its own parameters - the
``state_dict`` layout of the reference models is untouched) on a
``PackedSequence`` exactly like ``lstm(packed)[0]`` as used at
``padertorch/contrib/examples/source_separation/pit/model.py:97`` and ``contrib/tcl/dc.py:61``:

* per layer ONE dense GEMM ``X [W_ih_fwd; W_ih_rev]^T + (b_ih + b_hh)`` for both directions (BLAS);
* the time recurrence runs in the HIP kernels of ``csrc/lstm.hip`` (``ptmi_lstm_forward`` /
  ``ptmi_lstm_backward``): one launch per timestep for both directions, exact fp32 on the matrix
  cores, fused gate non-linearities, activations saved in place for the backward pass;
* weight / input gradients are dense GEMMs on the saved gate gradients.
"""
import ctypes
import functools
import os

import numpy as np
import torch
from torch.nn.utils.rnn import PackedSequence

from .. import _lib
from . import context as _context
from . import gemm as _gemm
from . import library  # noqa: F401  (registers torch.ops.ptmi.*)

__all__ = ['packed_lstm']


class _PackMeta:
    """Device-side bookkeeping of one ``batch_sizes`` vector (cached: batches repeat shapes)."""

    def __init__(self, batch_sizes, device):
        self.key = tuple(batch_sizes)
        bs = np.asarray(batch_sizes, dtype=np.int64)
        assert np.all(bs[:-1] >= bs[1:]), 'batch_sizes must be non-increasing (sorted sequences)'
        self.T = int(len(bs))
        self.max_batch = int(bs[0]) if self.T else 0
        offs = np.concatenate([[0], np.cumsum(bs)])
        self.rows = int(offs[-1])
        # host-side copies: the C ABI turns them into per-launch kernel arguments
        self.bs_host = np.ascontiguousarray(bs, dtype=np.int32)
        self.offs_host = np.ascontiguousarray(offs[:-1], dtype=np.int64)
        # device copies for the persistent kernels (read in-kernel, step by step)
        self.bs_dev = _lib.host_to_device(self.bs_host, torch.int32, device)          # (no host synchronisation: _lib.host_to_device)
        self.offs_dev = _lib.host_to_device(self.offs_host, torch.int64, device)
        # index of the predecessor row (forward sense) per direction; `rows` = "no predecessor"
        # (vectorised: a new length pattern every step - real training data - must not cost the host milliseconds)
        t_row = np.repeat(np.arange(self.T), bs)                      # time step / batch index of every packed row
        b_row = np.arange(self.rows) - offs[t_row] if self.T else np.zeros(0, np.int64)
        bs_next = np.append(bs[1:], 0) if self.T else bs
        prev = np.full((2, self.rows), self.rows, dtype=np.int64)
        if self.T:
            prev[0] = np.where(t_row > 0, offs[np.maximum(t_row - 1, 0)] + b_row, self.rows)
            prev[1] = np.where(b_row < bs_next[t_row], offs[t_row + 1] + b_row, self.rows)
        self.prev_dev = _lib.host_to_device(prev, torch.int64, device)
        # equal-length batch: the predecessor of packed row r is row r - bs[0] (forward direction) or
        # r + bs[0] (reverse direction), which `_LstmLayerFn` turns into shifted views of a padded buffer
        self.bs0 = int(bs[0]) if self.T else 0
        self.equal_lengths = bool(self.T and (bs == bs[0]).all())
        # per sequence b: rows of its first / last processed step per direction (initial / final states),
        # and the predecessor table with "no predecessor" pointing at row rows + 1 + b (= h0[b])
        lens = (bs[None, :] > np.arange(self.max_batch)[:, None]).sum(1) if self.T else np.zeros(0, np.int64)
        b_idx = np.arange(self.max_batch)
        end_rows = offs[np.maximum(lens - 1, 0)] + b_idx
        self.first_rows = _lib.host_to_device(np.stack([b_idx, end_rows]), torch.int64, device)
        self.last_rows = _lib.host_to_device(np.stack([end_rows, b_idx]), torch.int64, device)
        prev_h0 = prev.copy()
        row_b = b_row
        for d in range(2):
            fresh = prev[d] == self.rows
            prev_h0[d, fresh] = self.rows + 1 + row_b[fresh]
        self.prev_h0_dev = _lib.host_to_device(prev_h0, torch.int64, device)
        # packed row (t, b) -> row t * max_batch + b of the time-major padded tensor (ops.sequence.unpack_sequence)
        self.padded_rows = _lib.host_to_device(t_row * self.max_batch + b_row, torch.int64, device)


@functools.lru_cache(maxsize=64)
def _meta(batch_sizes_key, device_key):
    return _PackMeta(batch_sizes_key, torch.device(*device_key))


def pack_meta(batch_sizes, device):
    return _meta(tuple(int(b) for b in batch_sizes.tolist()), (device.type, device.index))


#: run the forward recurrence as ONE persistent launch per layer (W_hh resident in registers)
PERSISTENT = True
#: read back the error words of the persistent kernels after every call (host sync; tests only)
CHECK_PERSISTENT_ERRORS = False
#: Accumulate the weight gradients of the LSTM layers straight into the parameters' ``.grad`` buffers
#: (set by the Trainer when it owns flat gradient buffers) - and do so on a SIDE stream where that is
#: safe: the backward recurrence of the next (lower) layer occupies ~150-200 of the 256 CUs exclusively
#: (its workgroups hold a CU's whole register file), so the dW GEMMs of the layer above run on the idle
#: CUs meanwhile (bench config: 18.2 -> 17.4 ms per step, C3: 64 -> 60).
#: HAZARD and its guard: the persistent recurrence kernels need all their workgroups co-resident.  A
#: kernel running next to them must never wait for its own not-yet-dispatched workgroups, or the two
#: starve each other: hipBLASLt's Stream-K GEMMs (``SK3`` in their names) do, and with them the first
#: unsynchronised step at B = 64, T = 503 hung the GPU every time.  rocBLAS' tiled kernels do not (split-K
#: goes through a second kernel), so the side stream is used ONLY for shapes whose TunableOp entry pins
#: a rocBLAS solution (``_side_stream_safe``; ``scripts/tune_gemms.py --overlap`` produces them; 15 cold
#: starts over five configurations ran clean); every other shape accumulates on the main stream.
#: ``sync_deferred()`` must run before anything reads the gradients (the Trainer does).
DEFER_WGRAD = False
#: called with the list of parameters whose gradients have just been accumulated in place (the data-parallel
#: Trainer issues the layer's all-reduce from it; autograd's post-accumulate hooks do not fire for in-place writes)
GRAD_READY_HOOK = None
#: called in the FORWARD pass with the parameters of a module whose weight gradients the backward pass will accumulate in place:
#: one call per use, so that a module applied twice is reported ready only after its last backward use (GradBuckets.expect)
GRAD_USE_HOOK = None
#: False: the deferred accumulation runs on the current stream (same GEMM shapes, no overlap; used by the
#: one-off GEMM tuning, which must not time kernels next to a running recurrence)
WGRAD_SIDE_STREAM = True
_WGRAD_STREAMS = {}


# (Always on since they were measured - rounds 2-4, DESIGN.md sections 3.3 / 3.9 / 4 - and no switches any more: per-version cached stacked
#  weights; the first layer's weight gradients on both queues (the step's tail); the backward scratch's data-as-flag pattern written by the
#  forward recurrence kernel; the forward recurrence's hand-off planes as operand A of the next projection / dense layer; the gate gradients
#  handed to the weight-gradient GEMMs as bf16 planes of dgates^T by the backward kernel; both directions' dW_ih as one launch.)
#: attribute of a packed_lstm output tensor whose recurrence has left it as hand-off planes: (version, (scratch, cols), ndir, H)
HANDOFF_ATTR = '_ptmi_handoff_planes'


def handoff_planes_of(x):
    """``((scratch, cols), ndir, H)`` when ``x`` is an output tensor of :func:`packed_lstm` that still holds what its recurrence
    wrote (same object, no in-place edit since), else ``None``."""
    rec = getattr(x, HANDOFF_ATTR, None)
    if rec is None or rec[0] != x._version:
        return None
    return rec[1:]
#: LSTM input gradients on the planes GEMM straight from the backward recurrence's hand-off planes (no pack pass)
DX_FROM_HANDOFF = True
#: the top layer's backward recurrence runs in two launches for batches of at least this many packed rows (see _LstmLayerFn.backward)
SPLIT_TOP_BACKWARD_ROWS = 16384
_WGRAD_DONE = {}
#: captured steps (ops.capture): a layer's weight-gradient launches are ENQUEUED behind the next lower layer's recurrence launch (they
#: still wait for the event recorded where they used to be enqueued).  The hipGraph executor lays a captured step out by following a
#: node's FIRST-captured successor on the same queue: with the side-stream chain captured first, the next recurrence ended up behind
#: that chain on one queue (rocprofv3 timeline of the replay: 0.43 ms of weight-gradient GEMMs in front of the first layer's backward
#: recurrence instead of beside it); with the recurrence captured first the chain gets a queue of its own.
_PENDING_WGRAD = []


def flush_pending_wgrad(start=None):
    """Enqueue the deferred weight-gradient launches.  ``start``: an event on the main queue that launches without an event of their own
    (``ops.linear``: the dense layers' weight gradients) wait for - recorded in FRONT of the recurrence launch they are enqueued behind,
    i.e. they start beside that recurrence instead of beside the dense layers' input-gradient chain that leads up to it."""
    while _PENDING_WGRAD:
        _PENDING_WGRAD.pop(0)(start)


# (Measured in round 2 and not kept - DESIGN.md sections 3.9 / 4 have the numbers -: the pattern fill ahead of time on a side stream,
# the weight gradients' forward-data operand planes packed during the forward pass, a layer's weight gradients started behind its
# recurrence instead of behind its input-gradient GEMM, the backward recurrence cut into several launches.)


def _wgrad_stream(device):
    """The weight-gradient side stream that belongs to the CURRENT stream of ``device``: one per (device, main stream), so that
    two host threads that drive their own models on their own streams (reference ``trainer.py:412-420``) do not serialise on, or
    order themselves through, one shared side queue.  (The backward pass runs on autograd's thread with the forward pass' stream
    current, i.e. it finds the forward pass' side stream.)"""
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    # (keyed by the raw handle: torch hands out stream wrappers afresh on every call, so there is no object to hold weakly.  torch's
    #  streams come from a fixed pool per device and are never destroyed - a handle seen again IS the same queue -, so the table is
    #  bounded by the pool; an external stream that was destroyed and whose handle came back would find its predecessor's side stream,
    #  which is a valid side stream for it too.)
    key = (device.type, index, torch.cuda.current_stream(device).cuda_stream)
    if key not in _WGRAD_STREAMS:
        if len(_WGRAD_STREAMS) >= 64:
            sync_deferred()                     # nothing may be pending on a side stream that is let go
            _WGRAD_STREAMS.clear()
        _WGRAD_STREAMS[key] = torch.cuda.Stream(device=device)
    return _WGRAD_STREAMS[key]


_SIDE_SAFE = {}


def gemm_keys_safe(keys):
    """True when every TunableOp GEMM key has a pinned rocBLAS solution in the loaded results: rocBLAS
    kernels are plain tiled GEMMs (split-K through a second kernel), hipBLASLt's carry a Stream-K mode
    that spins on sibling workgroups - the latter must not run next to a persistent recurrence kernel
    (see DEFER_WGRAD).  Unknown shapes run on the main stream."""
    keys = tuple(keys)
    if keys not in _SIDE_SAFE:
        ok = False
        try:
            import torch.cuda.tunable as tunable
            if tunable.is_enabled():
                res = {params: sol for _op, params, sol, _t in tunable.get_results()}
                ok = all('Rocblas' in res.get(k, '') for k in keys)
        except Exception:
            ok = False
        _SIDE_SAFE[keys] = ok
    return _SIDE_SAFE[keys]


def wgrad_key(n_in, n_out, rows, ld_g=None, ld_x=None):
    """TunableOp key of ``W.grad[n_out, n_in].addmm_(g[rows, n_out].t(), x[rows, n_in])`` (``ld_g`` / ``ld_x``:
    row strides of ``g`` / ``x`` when they are column blocks of wider matrices)."""
    return f'nt_{n_in}_{n_out}_{rows}_ld_{ld_x or n_in}_{ld_g or n_out}_{n_in}'


def _side_stream_safe(rows, I, H, ndir, shifted_views):
    """`shifted_views`: h_{t-1} is a column block of the padded output buffer (row stride ndir * H), not a
    gathered copy."""
    G = 4 * H
    return gemm_keys_safe((wgrad_key(I, G, rows, ndir * G),
                           wgrad_key(H, G, rows, ndir * G, ndir * H if shifted_views else None)))


def warm_side_stream(device, nbytes=1 << 30):
    """Create the weight-gradient side stream of ``device`` and exercise everything it will need (its
    hardware queue, the allocator pool of that stream, the BLAS handle, the kernels) while the GPU is
    otherwise IDLE.  First use of a stream next to a running persistent recurrence kernel has been
    observed to stall the GPU (queue creation / first large allocations while a kernel that needs all of
    its workgroups co-resident is only partly dispatched); after this warm-up it does not."""
    device = torch.device(device)
    torch.cuda.synchronize(device)
    side = _wgrad_stream(device)
    with torch.cuda.stream(side):
        big = torch.empty(nbytes // 4, dtype=torch.float32, device=device)      # grows the side pool once
        a = torch.randn(512, 256, device=device)
        idx = torch.arange(512, device=device)
        acc = torch.zeros(256, 256, device=device)
        acc.addmm_(a.t(), torch.cat([a, a[:1]], 0).index_select(0, idx))
        acc.add_(a.sum(0))
        del big, a, idx, acc
    torch.cuda.synchronize(device)


def sync_deferred(device=None):
    """Make the current stream wait for every deferred weight-gradient accumulation."""
    flush_pending_wgrad()
    want = None
    if device is not None:
        device = torch.device(device)
        want = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    from . import capture as _capture
    for (typ, idx, main_handle), side in list(_WGRAD_STREAMS.items()):
        if want is None or want == (typ, idx):
            cur = torch.cuda.current_stream(torch.device(typ, idx))
            if _capture.ACTIVE and main_handle != cur.cuda_stream:
                continue            # a captured step joins ITS side stream; a wait for a stream outside the capture is no edge of the graph
            cur.wait_stream(side)


#: ONE word per device that every persistent launch whose bounded spin runs out increments (``ptmi_lstm_set_error_sink``):
#: the Trainer stages its value with the gradient norm once per optimizer step, ``check_errors`` reads it with a host
#: sync.  (Round 1 kept a view of every call's own error word and folded them with four small torch kernels per check.)
_ERR_SINK = {}      # (device type, index) -> [int32 device tensor [1], count the host has already reported]


def _error_sink(device):
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    ent = _ERR_SINK.get(key)
    if ent is None:
        with torch.cuda.device(key[1]):
            word = torch.zeros(1, dtype=torch.int32, device=device)
            _lib.check(_lib.load().ptmi_lstm_set_error_sink(word.data_ptr()), 'ptmi_lstm_set_error_sink')
        ent = _ERR_SINK[key] = [word, 0]
    return ent


def error_count(device):
    """0-dim int32 DEVICE tensor: persistent LSTM launches on ``device`` that timed out so far (no kernel, no host sync:
    the caller decides when the value crosses over and hands it to :func:`errors_since_last_report`)."""
    return _error_sink(device)[0][0]


def errors_since_last_report(device, count):
    """``count``: a host copy of :func:`error_count`.  True when it is beyond what has been reported before."""
    ent = _error_sink(device)
    new = int(count) > ent[1]
    ent[1] = max(ent[1], int(count))
    return new


def error_word(device):
    """0-dim int32 DEVICE tensor, non-zero iff a bounded spin of a persistent LSTM kernel on ``device`` ran out and has
    not been reported yet (no host sync)."""
    ent = _error_sink(device)
    return ent[0][0] - ent[1]


def raise_timeout(device):
    raise RuntimeError(
        f'padertorch_amd: a persistent LSTM kernel on {device} timed out waiting for a step counter '
        '(workgroups not co-resident, e.g. the GPU is shared with another long-running kernel). '
        'Set padertorch_amd.ops.lstm.PERSISTENT = False.')


def check_errors():
    """Raise if a bounded spin of a persistent LSTM kernel ran out since the last report (the recurrence results
    are then invalid).  One 4-byte device-to-host copy per device that has run such a kernel."""
    from . import capture as _capture
    if _capture.ACTIVE:         # (a host read: not inside a captured step - GraphedStep stages the word and checks it after the replay)
        return
    for key, ent in list(_ERR_SINK.items()):
        device = torch.device(key[0], key[1])
        if errors_since_last_report(device, int(ent[0])):
            raise_timeout(device)


_STACKED = {}


_PREP_STREAMS = {}


def _prep_stream(device, first_layer=False):
    """The parameter-form queue (``first_layer``: a second one, for the first BLSTM layer's forms of a captured step - see
    :func:`begin_captured_step`)."""
    key = (device.type, device.index, bool(first_layer))
    if key not in _PREP_STREAMS:
        _PREP_STREAMS[key] = torch.cuda.Stream(device=device)
    return _PREP_STREAMS[key]


_EARLY_FORK = {}            # (device type, index) -> the preparation streams a captured step has forked at its very start

#: where a captured step makes the parameter forms (A/B switch): 'off' = as the eager step orders them (first layer's on the main
#: queue behind the feature kernel, the others' on a queue that forks at the first BLSTM call); 'one' = the others' queue forks at the
#: head of the graph; 'two' = that, and the first layer's forms on a second queue forked there too
CAPTURE_FORK = 'two'


def begin_captured_step(device):
    """First thing inside the capture of an optimizer step (``train.graphed``, behind ``ops.capture.zero_block``): the preparation
    queue forks HERE - in front of the front-end kernels - instead of at the first BLSTM call, so that a replay makes every parameter
    form (all BLSTM layers', the dense layers' of the earlier steps) NEXT TO the feature kernel instead of behind it.  They read the
    parameters only, which the previous replay's optimizer kernel wrote.  (Round 5's replay: first recurrence at 490 us of the step;
    with this and the shorter ``lstm_weight_prep`` at 280 - of which the step keeps 40-60 us, the forms now run beside the first
    recurrence and slow it: DESIGN 4.4, ``profiles/r6_head_of_step.txt``.)"""
    from . import capture as _capture
    assert _capture.ACTIVE
    device = torch.device(device)
    if CAPTURE_FORK == 'off':
        return
    pre, pre0 = _prep_stream(device), (_prep_stream(device, True) if CAPTURE_FORK == 'two' else None)
    pre.wait_stream(torch.cuda.current_stream(device))
    if pre0 is not None:
        pre0.wait_stream(torch.cuda.current_stream(device))
    _EARLY_FORK[(device.type, device.index)] = (pre, pre0)
    # (Only the forks: the work itself is enqueued where it always was, by the first BLSTM call - a replay submits its nodes in
    #  capture order, and with the forms captured FIRST the feature kernel, head of the critical path, started 170 us later.  Two
    #  queues: a replay runs the nodes of one captured stream in order, and the first projection - whose operands are the feature
    #  kernel's output and the FIRST layer's forms - landed on the form queue behind all the other layers' forms, at 350 us.)


def end_captured_step(device):
    """Last thing inside the capture: every side queue the step has forked joins the capturing stream."""
    device = torch.device(device)
    for side in _EARLY_FORK.pop((device.type, device.index), None) or ():
        if side is not None:
            torch.cuda.current_stream(device).wait_stream(side)


def _stacked_stale(params):
    flat = tuple(p for ps in params for p in ps)
    hit = _STACKED.get(tuple(id(p) for p in flat))
    return hit is None or hit[0] != tuple((p._version, p.data_ptr()) for p in flat) or not _gemm._same(hit[2], flat)


class _LazyPlanes:
    """``ops.gemm.pack_n(w, amax)`` on first use (on the stream current then; ``w`` and ``amax`` were written on the stream the
    forms were made on, whose ``ready`` event every consumer has waited for)."""

    def __init__(self, w, amax):
        self.w, self.amax, self.value = w, amax, None

    def get(self):
        if self.value is None:
            with torch.no_grad():
                self.value = _gemm.pack_n(self.w, self.amax)
        return self.value


def _w_ih_planes(forms):
    v = forms['w_ih_planes']
    return v.get() if isinstance(v, _LazyPlanes) else v


def _stacked_weights(params, KP, stream=None):
    """The per-layer operand forms of a BLSTM layer's parameters - both directions' ``weight_ih`` stacked (and, for an
    input width that is not a multiple of 4, zero-padded along the reduction axis), the summed biases, ``weight_hh``
    stacked, padded to ``KP`` columns and transposed - cached until a parameter is modified (``_version`` / storage):
    they change once per optimizer step, not per micro-step or layer call (six concatenation / padding / transposition
    kernels per layer and pass, ~70 us of launch-bound work per layer of the B = 32 step).  Detached: only for calls
    whose weight gradients do not travel through autograd (``DEFER_WGRAD`` path, or no graph at all)."""
    key = tuple(id(p) for ps in params for p in ps)
    sig = tuple((p._version, p.data_ptr()) for ps in params for p in ps)
    flat_ps = tuple(p for ps in params for p in ps)
    hit = _STACKED.get(key)
    if hit is not None and hit[0] == sig and _gemm._same(hit[2], flat_ps):      # (weak references: see ops.gemm._WEIGHT_AMAX)
        return hit[1]
    if len(_STACKED) > 64:
        _STACKED.clear()
    with torch.no_grad():
        p0 = params[0][0]
        if p0.is_cuda and all(p.dtype == torch.float32 and p.is_contiguous() for ps in params for p in ps):
            # one launch: every parameter is read once (csrc/lstm_prep.hip); on `stream` when the caller prefetches the
            # forms of later layers next to the first layer's work (consumers wait for forms['ready'])
            main = torch.cuda.current_stream(p0.device)
            with torch.cuda.stream(stream if stream is not None else main):
                w_ih_k, bias, w_pad, w_t, amax = torch.ops.ptmi.lstm_weight_prep(
                    [ps[0].detach() for ps in params], [ps[1].detach() for ps in params], [ps[2].detach() for ps in params],
                    [ps[3].detach() for ps in params], KP)
                I, H = p0.shape[1], params[0][1].shape[1]
                # the stacked input weights with their input columns laid out like the previous layer's hand-off planes (H columns
                # per direction padded to the planes' width): that layer's scratch then is operand A of this layer's projection
                planes = planes_h = None
                ndir_ = len(params)
                cols_ = int(_lib.load().ptmi_lstm_handoff_cols(H, 0)) if _gemm.planes_enabled() else 0
                if cols_ and I == ndir_ * H:
                    planes_h = ((_gemm.pack_n_direction_blocks(w_ih_k[:, :I], ndir_, H, cols_, amax[0:1]), amax[0:1]), cols_)
                    if cols_ == H:
                        planes = planes_h[0]
                # fp16 planes of the stacked input weights as they are (the W of x W^T on csrc/gemm_planes.hip): at once for a
                # layer that has no other form (the first: its input is no hidden state), on first use (a dropout between the
                # layers, an initial state) for the others
                if _gemm.planes_enabled() and planes is None:
                    planes = _gemm.pack_n(w_ih_k[:, :I], amax[0:1]) if planes_h is None else _LazyPlanes(w_ih_k[:, :I], amax[0:1])
                # bf16 planes of W_ih as the right operand of dx = dgates W_ih on the backward recurrence's planes (layers whose
                # input needs a gradient: not the first); built here, off the backward pass' critical path
                planes_dx = None
                cols_b = int(_lib.load().ptmi_lstm_handoff_cols(H, 1)) if (planes is not None and DX_FROM_HANDOFF) else 0
                if cols_b and I == ndir_ * H:
                    planes_dx = (_gemm.stacked_planes_t_bf16(w_ih_k[:, :I], ndir_, cols_b), cols_b)
            forms = {'w_ih': w_ih_k[:, :I], 'bias': bias, 'w_hh': w_pad[:, :, :H], 'w_pad': w_pad, 'w_t': w_t, 'w_ih_planes_dx': planes_dx,
                     'w_ih_kpad': w_ih_k if w_ih_k.shape[1] != I else None, 'ready': None, 'w_ih_planes': planes,
                     'w_ih_planes_h': planes_h}
            if stream is not None:
                forms['ready'] = torch.cuda.Event()
                forms['ready'].record(stream)
                # These tensors come from the preparation stream's pool and are read on the main stream.  They are NOT marked with
                # record_stream(main): the allocator would then record one event per tensor on the MAIN queue when they are freed
                # (20 marker packets = a 75 us bubble behind the top layer's recurrence, where the previous step's graph let go
                # of them: scripts/phase_events.py, rocprofv3 --hip-runtime-trace).  Their memory can only be handed out again
                # by an allocation on the preparation stream, and every piece of work on that stream is enqueued behind a wait
                # for the optimizer kernel / the main stream (packed_lstm), i.e. behind every reader of the old forms.
            _gemm.seed_weights_absmax([ps[0] for ps in params], amax[0:1])
            _gemm.seed_weights_absmax([ps[1] for ps in params], amax[1:2])
        else:
            w_ih = torch.cat([ps[0] for ps in params], 0)
            bias = torch.cat([ps[2] + ps[3] for ps in params], 0)
            w_hh = torch.stack([ps[1] for ps in params], 0)
            H = w_hh.shape[2]
            kpad = -w_ih.shape[1] % 4
            forms = {
                'w_ih': w_ih, 'bias': bias, 'w_hh': w_hh,
                'w_ih_kpad': torch.nn.functional.pad(w_ih, (0, kpad)) if kpad else None,
                'w_pad': torch.nn.functional.pad(w_hh, (0, KP - H)).contiguous() if KP != H else w_hh.contiguous(),
                'w_t': w_hh.transpose(1, 2).contiguous(),
            }
    _STACKED[key] = (sig, forms, _gemm._refs(flat_ps))
    return forms


class _LstmLayerFn(torch.autograd.Function):
    """x [rows, I] -> hy [rows, ndir*H] for one layer (both directions)."""

    @staticmethod
    def forward(ctx, x, w_ih, bias, w_hh, meta, h0=None, c0=None, params=None, x_unit=False, anchor=None, forms=None, prev=None,
                handoff=None, top=False, oc=None):
        # oc: ops.context.Effective of the LSTM module this layer belongs to (None: the process defaults)
        # prev: {'planes': (scratch, cols)} of the layer whose output `x` is (its hand-off planes as this projection's operand);
        # handoff: dict this call leaves its own planes in
        # anchor: a Parameter of the layer when (w_ih, bias, w_hh) are the cached detached forms (`forms`), so that the
        # node stays in the graph although none of its tensor inputs may require a gradient (first layer)
        lib = _lib.load()
        ndir, G, H = w_hh.shape
        assert G == 4 * H
        KP = (H + 15) // 16 * 16
        ctx.set_materialize_grads(False)         # an unused output (the cell states of a call whose c_n nobody differentiates) comes back as None
        stateful = h0 is not None or c0 is not None
        if stateful:            # [ndir, B, H]; their gradients: see backward (persistent split kernels)
            h0 = torch.zeros_like(c0) if h0 is None else h0.detach().to(torch.float32).contiguous()
            c0 = torch.zeros_like(h0) if c0 is None else c0.detach().to(torch.float32).contiguous()
            assert h0.shape == c0.shape == (ndir, meta.max_batch, H), (h0.shape, c0.shape, meta.max_batch)
        st = _lib.stream(x.device)
        # hand-off scratch of this layer's backward pass when one will come: its data-as-flag pattern is written by the forward
        # recurrence kernel itself (below); the forward scratch is allocated and filled by the op
        scratch_f = scratch_b = None
        pre_f, pre_b = False, 0
        fills = int(lib.ptmi_lstm_forward_fills(meta.T, ndir, meta.max_batch, H)) if (
            PERSISTENT and scratch_b is None and x.is_cuda and any(ctx.needs_input_grad)) else 0
        if fills:
            # the forward recurrence itself writes the pattern into the planes of this layer's backward scratch (an idle
            # wavefront per workgroup, a slice per time step)
            scratch_b = torch.empty(int(lib.ptmi_lstm_scratch_elems(meta.T, ndir, meta.max_batch, H, 1)), dtype=torch.int32,
                                    device=x.device)
            pre_b = fills        # (2: the planes' pattern and the zeroed words behind them - the value the backward call takes as `prefilled`)
            fill_b = scratch_b
        else:
            fill_b = None
        use_gemm = _gemm.usable(x, w_ih)
        # operand ranges of the split GEMM: the layer input is taken as it is when it is a hidden state (|h| < 1,
        # a dropout scale aside), measured otherwise; the stacked weights' maximum is cached per optimizer step
        hplanes = prev.get('planes') if prev else None
        xplanes = prev.get('xplanes') if prev else None
        if not (xplanes is not None and use_gemm and _gemm.planes_enabled() and forms is not None
                and forms.get('w_ih_planes') is not None):
            xplanes = None
        amax_x = ((_gemm.UNIT_RANGE if x_unit else _gemm.scale_word(x.device, xplanes[1]) if xplanes is not None
                   else _gemm.absmax(x)) if use_gemm else None)
        amax_w = ((_gemm.weights_absmax([ps[0] for ps in params]) if params is not None else _gemm.absmax(w_ih))
                  if use_gemm else None)
        if xplanes is not None:
            # the producer of x has left it as fp16 planes with a fixed operand scale (the feature kernel: 2^9 log1p|Y|);
            # the same scale word serves the weight gradient's pack of the fp32 x in the backward pass
            word = amax_x
            gates = torch.empty((meta.rows, ndir * G), dtype=torch.float32, device=x.device)
            wpl = _w_ih_planes(forms)
            torch.ops.ptmi.gemm_planes_(gates, xplanes[0], word, wpl[0], wpl[1], bias, meta.rows, ndir * G, x.shape[1], False, 1)
        elif (hplanes is not None and use_gemm and _gemm.planes_enabled() and forms is not None
                and forms.get('w_ih_planes_h') is not None and forms['w_ih_planes_h'][1] == hplanes[1]):
            # the previous layer's recurrence has left its output as fp16 (hi, lo) planes of 2^10 h in fragment order (its
            # hand-off copy): operand A of this projection as it lies, no pack pass
            gates = torch.empty((meta.rows, ndir * G), dtype=torch.float32, device=x.device)
            kh = (x.shape[1] // H) * hplanes[1]
            wpl = forms['w_ih_planes_h'][0]
            torch.ops.ptmi.gemm_planes_(gates, hplanes[0], _gemm.scale_word(x.device), wpl[0], wpl[1], bias, meta.rows, ndir * G, kh,
                                        False, _gemm.auto_split_k(meta.rows, ndir * G, kh))
        elif use_gemm and _gemm.planes_enabled() and forms is not None and forms.get('w_ih_planes') is not None:
            # both operands as fp16 planes: the input split once here, the stacked weights' planes come with the forms
            gates = torch.empty((meta.rows, ndir * G), dtype=torch.float32, device=x.device)
            _gemm.mm_planes_(gates, _gemm.pack_n(x, amax_x), _w_ih_planes(forms), meta.rows, ndir * G, x.shape[1], bias=bias)
        elif use_gemm:
            # an input width that is not a multiple of 4 (F = 257) would send the projection and its weight gradient
            # down the kernel's unaligned (scalar-load) path: zero-pad the reduction axis of both operands instead
            kpad = -x.shape[1] % 4
            if kpad:
                x_in = x
                x = torch.nn.functional.pad(x_in, (0, kpad))
                w_ih_k = forms['w_ih_kpad'] if forms is not None else torch.nn.functional.pad(w_ih, (0, kpad))
                gates = _gemm.mm(x, w_ih_k.t(), bias=bias, amax_x=amax_x, amax_y=amax_w)
                x = x[:, :x_in.shape[1]]                  # view with the padded row stride: what the backward pass multiplies
            else:
                gates = _gemm.mm(x, w_ih.t(), bias=bias, amax_x=amax_x, amax_y=amax_w)
        else:
            gates = torch.addmm(bias, x, w_ih.t())
        if stateful:        # h0 W_hh^T enters the pre-activations of each sequence's first processed step
            gv = gates.view(meta.rows, ndir, G)
            for d in range(ndir):
                gv[:, d].index_add_(0, meta.first_rows[d], h0[d] @ w_hh[d].t())
        if forms is not None:
            w_pad = forms['w_pad']
        else:
            w_pad = torch.nn.functional.pad(w_hh, (0, KP - H)).contiguous() if KP != H else w_hh.contiguous()
        # equal-length batch: bs[0] rows of "state before the first step" (zero or h0) in front of and
        # behind the output rows, so that the backward pass reads h_{t-1} as a shifted view (no gather)
        pad = meta.bs0 if meta.equal_lengths else 0
        masks = getattr(meta, 'masks_dev', None)          # row-slot batch (ops.sequence.SlotLayout): idle rows stay zero
        assert masks is None or not stateful, 'row-slot batches take no initial states'
        ext = (torch.zeros if masks is not None else torch.empty)((meta.rows + 2 * pad, ndir * H), dtype=torch.float32, device=x.device)
        hy = ext[pad:pad + meta.rows]
        if pad:
            # (both ends in ONE fill launch: a [2, pad, C] view over the first and the last `pad` rows)
            C_ = ndir * H
            torch.as_strided(ext, (2, pad, C_), ((pad + meta.rows) * C_, C_, 1)).zero_()
            if stateful:
                ext[:pad].view(pad, ndir, H)[:, 0] = h0[0]
                if ndir > 1:
                    ext[pad + meta.rows:].view(pad, ndir, H)[:, 1] = h0[1]
        ctx.ext = ext if pad else None
        # split-precision recurrence: the scale of W_hh's fp16 halves comes from its maximum (cached per optimizer step)
        amax_whh = None
        if PERSISTENT and lib.ptmi_lstm_split_enabled():
            amax_whh = (_gemm.weights_absmax([ps[1] for ps in params]) if params is not None
                        else _gemm.absmax(w_pad.view(-1, KP)))
        if PERSISTENT:
            _error_sink(x.device)           # the word a timed-out launch reports to (set before the first launch)
        c, flags = torch.ops.ptmi.lstm_recurrence_forward(
            gates, hy, c0, w_pad, amax_whh, meta.bs_dev, meta.offs_dev, meta.bs_host.ctypes.data, meta.offs_host.ctypes.data,
            meta.T, meta.max_batch, meta.rows, H, KP, ndir, PERSISTENT, scratch_f, pre_f, fill_b, masks)
        if fill_b is not None and flags is None:        # the persistent launch was refused: nothing was filled
            pre_b = 0
        # (a row-slot batch's planes are operands as well: its kernels write zeros for the idle slot steps)
        if handoff is not None and flags is not None and not stateful and (meta.equal_lengths or masks is not None) and meta.bs0 % 16 == 0:
            cols_out = int(lib.ptmi_lstm_handoff_cols(H, 0))
            if cols_out:
                handoff['planes'] = (flags, cols_out)
        ctx.scratch_b = (scratch_b, pre_b)
        if flags is not None:
            if CHECK_PERSISTENT_ERRORS:
                check_errors()
        ctx.save_for_backward(x, w_ih, w_hh, gates, c, hy, h0, c0)
        ctx.gemm = (amax_x, amax_w) if use_gemm else None
        ctx.meta = meta
        ctx.params = params
        ctx.forms = forms
        ctx.top = bool(top)          # the layer whose backward pass runs first (nothing else is on the weight-gradient queue then)
        ctx.oc = oc if oc is not None else _context.effective(None)
        if stateful:
            return hy, c            # (c: differentiable too - the gradient of the FINAL cell state comes back through it, see backward)
        return hy

    @staticmethod
    def backward(ctx, dhy, _dc=None):
        meta = ctx.meta
        if dhy is None:             # only the cell states were used
            dhy = torch.zeros((meta.rows, ctx.saved_tensors[2].shape[0] * ctx.saved_tensors[2].shape[2]), dtype=torch.float32,
                              device=ctx.saved_tensors[0].device)
        h0 = c0 = db_kernel = amax_kernel = None
        lib = _lib.load()
        st = _lib.stream(dhy.device)
        x, w_ih, w_hh, gates, c, hy, h0, c0 = ctx.saved_tensors
        ndir, G, H = w_hh.shape
        state_grad = False
        carry = None
        gm, params = ctx.gemm, ctx.params
        has_grads = params is not None and all(p.is_leaf and p.grad is not None for ps in params for p in ps)
        if ctx.forms is not None and not has_grads:
            raise RuntimeError('packed_lstm: the forward pass ran on the cached stacked weights (in-place weight gradients), '
                               'but a parameter of the layer has no .grad buffer any more')
        # weight gradients accumulated in place (see DEFER_WGRAD), on the side stream where that is safe: the split GEMM
        # kernels never wait for sibling workgroups - always safe next to a persistent recurrence -, library kernels only
        # when their shape is pinned to a rocBLAS solution
        oc = ctx.oc
        in_place = (oc.defer_wgrad or ctx.forms is not None) and has_grads
        use_side = in_place and oc.wgrad_side_stream and (
            gm is not None or _side_stream_safe(meta.rows, x.shape[1], H, ndir, ctx.ext is not None))
        main = torch.cuda.current_stream(x.device) if in_place else None
        side = _wgrad_stream(x.device) if use_side else main
        before_recurrence = None
        if _PENDING_WGRAD and dhy.is_cuda:       # (deferred launches of the layers above: they start where this layer's recurrence starts)
            before_recurrence = torch.cuda.Event()
            before_recurrence.record(torch.cuda.current_stream(dhy.device))
        operands, xplanes = [None], {}

        dgplanes = {}

        def wgrad_rows(dg, ranges, amax_dg, both_queues=False, dg_t=None):
            """dW_ih, dW_hh of every direction d over the rows ranges[d] = (r0, r1) of the packed batch, on `side`
            (both_queues: all but the forward direction's dW_hh on the main stream - the step's tail, see `both` below).
            dg_t: the kernel's bf16 planes of dgates^T for exactly these row ranges (then `dg` is None)."""
            # the first layer's weight gradients have the chip to themselves (the step's tail): big tiles; every other layer's run
            # beside the next recurrence: short ones on the kernel whose workgroups share CUs with the recurrence's
            split_of = _gemm.auto_split_k if not ctx.needs_input_grad[0] else _gemm.co_resident_split_k
            # both directions' dW_ih = [dgates_f | dgates_r]^T x share the operand x and the kernel's planes of dgates^T lie behind each
            # other: ONE launch with a two-part output (ptmi_gemm_planes_bf16_two) instead of two GEMMs + two slab reductions
            fused_ih = False
            if (dg_t is not None and ndir == 2 and ranges[0] == ranges[1] and ranges[0][1] > ranges[0][0]):
                ga, gb = params[0][0].grad, params[1][0].grad
                if ga.stride() == gb.stride() and (ga.data_ptr() ^ gb.data_ptr()) & 15 == 0 and ga.is_contiguous():
                    r0, r1 = ranges[0]
                    with torch.cuda.stream(main if both_queues else side):
                        key = (r0, r1)
                        if key not in xplanes:
                            xplanes[key] = torch.ops.ptmi.pack_planes_bf16(x[r0:r1], True)
                        torch.ops.ptmi.gemm_planes_bf16_two_(ga, gb, dg_t, 0, xplanes[key], 2 * G, x.shape[1], r1 - r0, True,
                                                             split_of(2 * G, x.shape[1], r1 - r0))
                    fused_ih = True
            for d, ((p_wih, p_whh, _, _), (r0, r1)) in enumerate(zip(params, ranges)):
                q_ih = main if both_queues else side
                q_hh = main if both_queues and d == 1 else side
                with torch.cuda.stream(q_ih):
                    if operands[0] is None:
                        operands[0] = _recurrent_operands(meta, dg, hy, ctx.ext, h0, ndir, H)
                    dgd, h_prev = operands[0][d]
                    if r1 <= r0:
                        continue
                    if dg_t is not None:
                        # both operands reduce over the packed rows: dgates^T comes from the recurrence as bf16 planes; the layer
                        # input (once for both directions) and the previous hidden state are packed to match (bf16: no scale)
                        k = r1 - r0
                        key = (r0, r1)
                        a_off = d * int(lib.ptmi_planes_elems(G, k)) * 2
                        if key not in xplanes and not fused_ih:
                            xplanes[key] = torch.ops.ptmi.pack_planes_bf16(x[r0:r1], True)
                        if not fused_ih:
                            torch.ops.ptmi.gemm_planes_bf16_(p_wih.grad, dg_t, a_off, xplanes[key], None, G, x.shape[1], k, True,
                                                             split_of(G, x.shape[1], k))
                        with torch.cuda.stream(q_hh):
                            hpl = torch.ops.ptmi.pack_planes_bf16(h_prev[r0:r1], True)
                            torch.ops.ptmi.gemm_planes_bf16_(p_whh.grad, dg_t, a_off, hpl, None, G, H, k, True, split_of(G, H, k))
                        continue
                    dgt = dgd[r0:r1].t()
                    if gm is not None and _gemm.planes_enabled():
                        # both operands reduce over the batch's rows (their outer axis): split them into fp16 planes once
                        # (dg for two GEMMs, the layer input for both directions) and run the plain 16-bit GEMM
                        k = r1 - r0
                        key = (r0, r1)
                        dgp = dgplanes.get((d, key))
                        if dgp is None:
                            dgp = _gemm.pack_t(dgd[r0:r1], amax_dg)
                        if key not in xplanes:
                            xplanes[key] = _gemm.pack_t(x[r0:r1], gm[0])
                        _gemm.mm_planes_(p_wih.grad, dgp, xplanes[key], G, x.shape[1], k, accumulate=True, split_k=split_of(G, x.shape[1], k))
                        with torch.cuda.stream(q_hh):
                            hpl = _gemm.pack_t(h_prev[r0:r1], _gemm.UNIT_RANGE if h0 is None else None)
                            _gemm.mm_planes_(p_whh.grad, dgp, hpl, G, H, k, accumulate=True, split_k=split_of(G, H, k))
                    elif gm is not None:
                        _gemm.mm(dgt, x[r0:r1], out=p_wih.grad, accumulate=True, amax_x=amax_dg, amax_y=gm[0])
                        _gemm.mm(dgt, h_prev[r0:r1], out=p_whh.grad, accumulate=True, amax_x=amax_dg,
                                 amax_y=_gemm.UNIT_RANGE if h0 is None else None)
                    else:
                        p_wih.grad.addmm_(dgt, x[r0:r1])
                        p_whh.grad.addmm_(dgt, h_prev[r0:r1])

        todo = [(0, meta.rows)] * ndir                  # row ranges whose weight gradients are still to be accumulated
        use_tp, dg_t = False, None
        dhy = dhy.contiguous()
        w_t = ctx.forms['w_t'] if ctx.forms is not None else w_hh.transpose(1, 2).contiguous()      # [ndir, H, 4H]
        if PERSISTENT:
            _error_sink(dhy.device)
        dg = flags = None
        T = meta.T
        # gradients w.r.t. the initial state: the range entry point leaves the cell-state gradient behind the last step
        state_grad = h0 is not None and any(ctx.needs_input_grad[5:7])
        # gradient w.r.t. the FINAL cell state: `_dc` is the gradient of the cell-state tensor this layer returned; packed_lstm
        # exposes only each sequence's last row of it (c_n), so only those rows can carry a gradient: gathered into [ndir, B, H]
        # and handed to the kernel, which adds it to the cell-state gradient at each sequence's last step
        dcn = None
        if _dc is not None and h0 is not None:
            dcv = _dc.reshape(meta.rows, ndir, H)
            dcn = torch.stack([dcv[meta.last_rows[d], d] for d in range(ndir)]).contiguous()
            state_grad = True                    # (the states entry point of the range launcher)
        if state_grad and not (PERSISTENT and lib.ptmi_lstm_split_enabled()):
            raise NotImplementedError('gradients w.r.t. the LSTM states need the persistent split kernels')
        carry = None
        # The TOP layer's backward recurrence runs while the weight-gradient queue is still empty; for long batches, where
        # that queue is the critical one of the backward phase (16 kHz configurations: 11.9 ms of GEMMs and pack passes
        # beside 9.7 ms of recurrences), it runs as two launches over step ranges and the finished half's weight gradients
        # start under the second launch (ptmi_lstm_backward_persistent_range).  For every layer, or at B = 32 / T = 253,
        # the same cut measured neutral to slower (a recurrence next to GEMMs loses what the GEMMs gain): c3 23.97 -> 23.55 ms
        # with the top layer in two launches, 23.35 / 23.33 in three / four, 23.70 with every layer in two.
        masks = getattr(meta, 'masks_dev', None)          # row-slot batch
        # (The BOTTOM layer in two launches - its first half's weight gradients under its second launch instead of in the step's tail -
        #  measured in the captured step, round 5: c2 6.60 -> 6.90 ms, c3 21.4 -> 22.0, c5 18.1 -> 18.5.  The replay's timeline: the side
        #  queue has no room in that window - the layer above's weight gradients (0.45 ms) run there, and as captured they ended up BEHIND
        #  the first range's -, two launches take 36 us longer than one, and the last range's GEMMs lose the two-queue tail.)
        chunks = 2 if (getattr(ctx, 'top', False) and PERSISTENT and use_side and gm is not None
                       and _gemm.planes_enabled() and lib.ptmi_lstm_split_enabled() and T >= 128
                       and meta.rows >= SPLIT_TOP_BACKWARD_ROWS and masks is None) else 1
        # the gate gradients as bf16 planes of dgates^T straight from the kernel (no row-major fp32 tensor at all when the
        # input gradient takes the hand-off planes, or is not needed)
        cols_dx = int(lib.ptmi_lstm_handoff_cols(H, 1)) if DX_FROM_HANDOFF else 0
        uniform_rows = (meta.equal_lengths or masks is not None) and meta.bs0 % 16 == 0        # rows = [T, batch]: the planes are operands
        dx_needs_rows = ctx.needs_input_grad[0] and not (cols_dx and uniform_rows)
        use_tp = bool(PERSISTENT and in_place and gm is not None and _gemm.planes_enabled()
                      and not state_grad and not dx_needs_rows
                      and lib.ptmi_lstm_backward_planes_ok(T, ndir, meta.max_batch, meta.rows, H))
        if use_tp:
            flags = ctx.scratch_b[0] if ctx.scratch_b[0] is not None else torch.empty(
                int(lib.ptmi_lstm_scratch_elems(T, ndir, meta.max_batch, H, 1)), dtype=torch.int32, device=dhy.device)
            pre = int(ctx.scratch_b[1]) if flags is ctx.scratch_b[0] else 0
            cuts = [T * i // chunks for i in range(chunks + 1)]
            carry = torch.empty((ndir, meta.max_batch, H), dtype=torch.float32, device=dhy.device) if chunks > 1 else None
            B_ = meta.max_batch

            def launch_tp(i):
                n_rows = (cuts[i + 1] - cuts[i]) * B_
                planes = torch.empty(ndir * int(lib.ptmi_planes_elems(G, n_rows)), dtype=torch.bfloat16, device=dhy.device)
                ok = torch.ops.ptmi.lstm_recurrence_backward_planes(
                    gates, c, c0, dhy, w_t, None, planes, flags, carry, meta.bs_dev, meta.offs_dev, T, B_, meta.rows, H, ndir,
                    cuts[i], cuts[i + 1], pre, masks)
                # rows of this range per direction (forward direction: processed from the last time index down)
                part = [((T - cuts[i + 1]) * B_, (T - cuts[i]) * B_), (cuts[i] * B_, cuts[i + 1] * B_)][:ndir]
                return ok, planes, part
            ok, dg_t, part_t = launch_tp(0)
            if ok:
                for i in range(1, chunks):
                    done = torch.cuda.Event()
                    done.record(main)
                    finished, finished_part = dg_t, part_t
                    # (the next range's launch is enqueued FIRST: a captured step is laid out in capture order, and the recurrence must
                    #  not end up behind the side queue's GEMMs - _PENDING_WGRAD)
                    ok, dg_t, part_t = launch_tp(i)
                    if not ok:
                        raise RuntimeError('ptmi_lstm_backward_persistent_planes: a later range was refused')
                    side.wait_event(done)
                    wgrad_rows(None, finished_part, None, dg_t=finished)        # the finished range, under the next launch
                    finished.record_stream(side)
                todo = part_t
            else:
                use_tp, dg_t, flags = False, None, None
        else:
            dg_t = None
        if not use_tp and (chunks > 1 or state_grad):
            dg = torch.empty_like(gates)
            flags = ctx.scratch_b[0] if ctx.scratch_b[0] is not None else torch.empty(
                int(lib.ptmi_lstm_scratch_elems(T, ndir, meta.max_batch, H, 1)), dtype=torch.int32, device=dhy.device)
            carry = torch.empty((ndir, meta.max_batch, H), dtype=torch.float32, device=dhy.device)
            cuts = [T * i // chunks for i in range(chunks + 1)]
            nflags = int(lib.ptmi_lstm_flags_elems(T, ndir, meta.max_batch)) + 8
            amax_word = flags[flags.numel() - nflags:flags.numel() - nflags + 1]
            offs = [int(v) for v in meta.offs_host[:T]] + [meta.rows]

            def launch(i):
                return torch.ops.ptmi.lstm_recurrence_backward_range(
                    gates, c, c0, dhy, w_t, dg, flags, carry, meta.bs_dev, meta.offs_dev, T, meta.max_batch, meta.rows, H,
                    ndir, cuts[i], cuts[i + 1], int(ctx.scratch_b[1]) if flags is ctx.scratch_b[0] else 0, dcn)
            if launch(0):
                for i in range(1, chunks):
                    snap = amax_word.clone()                     # max |dgates| so far: the operand scale of this part
                    done = torch.cuda.Event()
                    done.record(main)
                    side.wait_event(done)
                    s0, s1 = cuts[i - 1], cuts[i]                # steps finished by the previous launch
                    part = [(offs[T - s1], offs[T - s0]), (offs[s0], offs[s1])][:ndir]
                    wgrad_rows(dg, part, snap)
                    snap.record_stream(side)
                    if not launch(i):
                        raise RuntimeError('ptmi_lstm_backward_persistent_range: a later range was refused')
                if chunks > 1:
                    s0 = cuts[chunks - 1]
                    todo = [(offs[0], offs[T - s0]), (offs[s0], offs[T])][:ndir]
            else:
                dg = flags = None                                # not resident: the one-call path decides
                if state_grad:
                    raise NotImplementedError('gradients w.r.t. the initial LSTM state: this configuration cannot run on the '
                                              'persistent kernels')
        if dg is None and not use_tp:
            dg, flags = torch.ops.ptmi.lstm_recurrence_backward(
                gates, c, c0, dhy, w_t, meta.bs_dev, meta.offs_dev, meta.bs_host.ctypes.data, meta.offs_host.ctypes.data,
                T, meta.max_batch, meta.rows, H, ndir, PERSISTENT, ctx.scratch_b[0], int(ctx.scratch_b[1]), masks)
        if flags is not None:
            if CHECK_PERSISTENT_ERRORS:
                check_errors()
            # scratch tail: [bias gradient [ndir * 4H] | 8 words, word 0 = max |dgates| | slots | error words]
            nflags = int(lib.ptmi_lstm_flags_elems(T, ndir, meta.max_batch)) + 8
            db_kernel = flags[flags.numel() - nflags - ndir * G:flags.numel() - nflags].view(torch.float32)
            if lib.ptmi_lstm_split_enabled():
                amax_kernel = flags[flags.numel() - nflags:flags.numel() - nflags + 1]
        flush_pending_wgrad(before_recurrence)          # (captured steps: the layer above's weight gradients, behind this layer's recurrence launch)
        amax_dg = None
        if gm is not None:
            amax_x, amax_w = gm
            # one scale for the whole gate-gradient tensor (both directions): the backward kernel tracked its maximum
            amax_dg = amax_kernel if (amax_kernel is not None or dg is None) else _gemm.absmax(dg)
            cols = int(lib.ptmi_lstm_handoff_cols(H, 1)) if (DX_FROM_HANDOFF and flags is not None and amax_kernel is not None) else 0
            if not ctx.needs_input_grad[0]:
                dx = None
            elif cols and _gemm.planes_enabled() and (meta.equal_lengths or getattr(meta, 'masks_dev', None) is not None) and meta.bs0 % 16 == 0:
                # the recurrence has left its gate gradients as bf16 (hi, lo) planes in fragment order at the start of its
                # scratch (the hand-off copy): for a batch of equal lengths they ARE operand A of dx = dgates W_ih
                pdx = ctx.forms.get('w_ih_planes_dx') if ctx.forms is not None else None
                wplanes = pdx[0] if (pdx is not None and pdx[1] == cols) else _gemm.stacked_planes_t_bf16(
                    w_ih, ndir, cols, None if params is None else [ps[0] for ps in params])
                dx = torch.empty((meta.rows, w_ih.shape[1]), dtype=torch.float32, device=dhy.device)
                torch.ops.ptmi.gemm_planes_bf16_(dx, flags, 0, wplanes, None, meta.rows, w_ih.shape[1], ndir * cols, False,
                                                 _gemm.auto_split_k(meta.rows, w_ih.shape[1], ndir * cols))
            else:
                dx = _gemm.mm(dg, w_ih, amax_x=amax_dg, amax_y=amax_w)
        else:
            dx = dg @ w_ih if ctx.needs_input_grad[0] else None           # [rows, I]
        if in_place:
            # the first layer's weight gradients are the step's tail (nothing but the optimizer follows), and the side queue reaches
            # them ~0.25 ms after the main queue has gone idle (it still has the layer above's GEMMs: scripts/phase_events.py): all
            # but the forward direction's dW_hh go to the main queue, so that the side queue is done first and the optimizer does
            # not start behind a cross-queue hand-over (small launches with ~12 us of dispatch gap between dependent kernels of
            # one queue; a wait for an event that has not fired yet costs 30-60 us)
            both = (use_side and ndir > 1 and gm is not None and _gemm.planes_enabled()
                    and not ctx.needs_input_grad[0] and todo[0] == (0, meta.rows))
            if both:
                for p in (params[0][0], params[1][0], params[1][1]):     # earlier side-stream accumulations into the same .grad views
                    ev = _WGRAD_DONE.get(id(p))
                    if ev is not None:
                        main.wait_event(ev)
                operands[0] = _recurrent_operands(meta, dg, hy, ctx.ext, h0, ndir, H)
                # shared between the queues: packed before they part
                if dg_t is not None:
                    xplanes[todo[0]] = torch.ops.ptmi.pack_planes_bf16(x, True)
                else:
                    xplanes[todo[0]] = _gemm.pack_t(x, gm[0])
                    dgplanes[(0, todo[0])] = _gemm.pack_t(operands[0][0][0], amax_dg)
            from . import capture as _capture
            # captured steps: enqueue behind the next lower layer's recurrence launch (_PENDING_WGRAD); the side stream waits for the
            # event of THIS point, as it would have
            later = bool(_capture.ACTIVE and use_side and not both and ctx.needs_input_grad[0])
            here = None
            if later:
                here = torch.cuda.Event()
                here.record(main)
            ext_ = ctx.ext

            def accumulate(_start=None):
                if use_side:
                    if later:
                        side.wait_event(here)
                    else:
                        side.wait_stream(main)
                else:
                    main.wait_stream(_wgrad_stream(x.device))      # earlier accumulations into the same .grad views
                wgrad_rows(dg, todo, amax_dg, both_queues=both, dg_t=dg_t)
                with torch.cuda.stream(side):
                    if db_kernel is not None and all(ps[2].grad.is_contiguous() and ps[3].grad.is_contiguous() for ps in params):
                        # the kernel's bias sums into all 2 ndir bias gradients: one launch (was one small `add_` per bias vector)
                        torch.ops.ptmi.lstm_bias_grad_add_(db_kernel, [ps[2].grad for ps in params], [ps[3].grad for ps in params])
                    else:
                        for d, (_, _, p_bih, p_bhh) in enumerate(params):
                            db_d = operands[0][d][0].sum(0) if db_kernel is None else db_kernel[d * G:(d + 1) * G]
                            p_bih.grad.add_(db_d)
                            p_bhh.grad.add_(db_d)
                done = None
                if use_side:
                    done = torch.cuda.Event()
                    done.record(side)
                    for ps in params:
                        for p in ps[:2]:
                            _WGRAD_DONE[id(p)] = done
                if both:
                    main.wait_event(done)                          # the main queue is now behind both
                    for t in (dgplanes[(0, todo[0])][:1] if dg_t is None else (xplanes[todo[0]],)):
                        t.record_stream(side)
                    if oc.grad_ready_hook is not None:
                        side.wait_stream(main)                     # whoever orders itself after `side` sees every gradient
                if use_side:
                    for t in (x, hy) + tuple(v for v in (dg, dg_t, h0, ext_, db_kernel, amax_dg) if v is not None and torch.is_tensor(v)) \
                            + tuple(h_prev for _, h_prev in operands[0]):
                        t.record_stream(side)           # keep the operands alive until the side stream is done
                if oc.grad_ready_hook is not None:
                    oc.grad_ready_hook([p for ps in params for p in ps])

            if later:
                _PENDING_WGRAD.append(accumulate)
            else:
                accumulate()
            gh0 = gc0 = None
            if state_grad:
                gh0, gc0 = _state_grads(meta, dg, w_hh, carry, ndir, G, ctx.needs_input_grad)
            return (dx, None, None, None, None, gh0, gc0) + (None,) * 8
        db = dg.sum(0) if db_kernel is None else db_kernel
        if gm is not None:
            dw_ih = _gemm.mm(dg.t(), x, amax_x=amax_dg, amax_y=amax_x)
            dw_hh = torch.stack([_gemm.mm(a.t(), b, amax_x=amax_dg, amax_y=_gemm.UNIT_RANGE if h0 is None else None)
                                 for a, b in _recurrent_operands(meta, dg, hy, ctx.ext, h0, ndir, H)])
        else:
            dw_ih = dg.t() @ x                                            # [ndir*4H, I]
            dw_hh = torch.stack([a.t() @ b for a, b in _recurrent_operands(meta, dg, hy, ctx.ext, h0, ndir, H)])
        gh0 = gc0 = None
        if state_grad:
            gh0, gc0 = _state_grads(meta, dg, w_hh, carry, ndir, G, ctx.needs_input_grad)
        return (dx, dw_ih, db, dw_hh, None, gh0, gc0) + (None,) * 8


def _state_grads(meta, dg, w_hh, carry, ndir, G, needs):
    """Gradients w.r.t. (h0, c0) ``[ndir, B, H]``: ``h0`` entered the pre-activations of each sequence's first processed step
    through ``W_hh`` (``dh0 = dgates_first W_hh``), ``c0`` through that step's forget gate (the kernel's cell-state gradient
    behind its last step)."""
    dgv = dg.view(meta.rows, ndir, G)
    gh0 = torch.stack([dgv[meta.first_rows[d], d] @ w_hh[d] for d in range(ndir)]) if needs[5] else None
    gc0 = carry.clone() if needs[6] else None
    return gh0, gc0


def _recurrent_operands(meta, dg, hy, ext, h0, ndir, H):
    """Per direction d the operands (a, b) of dW_hh[d] = a^T @ b: the gate gradients of every row and the
    hidden state that row's step consumed (zero / h0 for a sequence's first processed step)."""
    rows, G = meta.rows, 4 * H
    dgv = dg.view(rows, ndir, G) if dg is not None else None
    if ext is not None:             # equal lengths: the padded buffer of the forward pass, shifted by one step
        n0 = meta.bs0
        extv = ext.view(rows + 2 * n0, ndir, H)
        return [(dgv[:, d] if dgv is not None else None, extv[:rows, 0] if d == 0 else extv[2 * n0:, 1]) for d in range(ndir)]
    parts = [hy.view(rows, ndir, H), hy.new_zeros(1, ndir, H)]      # row `rows` = zero state
    prev = meta.prev_dev
    if h0 is not None:                                              # rows rows+1+b = h0[:, b]
        parts.append(h0.transpose(0, 1))
        prev = meta.prev_h0_dev
    hy_pad = torch.cat(parts, 0)
    return [(dgv[:, d] if dgv is not None else None, hy_pad[:, d].index_select(0, prev[d])) for d in range(ndir)]


def unsupported_reason(lstm, data):
    """Why :func:`packed_lstm` cannot evaluate ``lstm`` on ``data`` (``None``: it can).  Any ``hidden_size`` and ``bias=False`` are
    covered (round 5: :class:`_PaddedLstm`); ``proj_size`` and non-fp32 modules are not."""
    if not isinstance(lstm, (torch.nn.LSTM, _PaddedLstm)):
        return f'{type(lstm).__name__} is not a torch.nn.LSTM'
    if lstm.proj_size != 0:
        return 'proj_size != 0'
    if not data.is_cuda:
        return 'the input is not on the GPU'
    if data.dtype != torch.float32:
        return f'input dtype {data.dtype} (fp32 only)'
    return None


def supported(lstm, data):
    return unsupported_reason(lstm, data) is None


class _PaddedLstm:
    """A ``torch.nn.LSTM`` whose ``hidden_size`` is not a multiple of 4 (the kernels' unit granularity: 16-byte weight rows), or that has
    no biases, as the LSTM the kernels DO run: every gate block of every parameter zero-padded from H to H4 = 4 ceil(H / 4) units (the
    reference constructs ``torch.nn.LSTM(F, units)`` for any ``units``, ``pit/model.py:60-66``).  A padded unit's pre-activations are 0 at
    every step - i = f = o = 1/2, g = 0, so c = h = 0 for ever - and its columns of ``W_hh`` / of the next layer's ``W_ih`` are zero:
    the real units compute what they compute in the unpadded LSTM, bit for bit in exact arithmetic.  The padded tensors are functions of
    the module's parameters (``F.pad``): gradients reach the parameters through autograd, the padding's gradients are dropped there."""
    proj_size = 0
    bias = True

    def __init__(self, lstm):
        H, ndir = lstm.hidden_size, 2 if lstm.bidirectional else 1
        H4 = (H + 3) // 4 * 4
        self.real_hidden, self.hidden_size = H, H4
        self.num_layers, self.bidirectional, self.dropout, self.training = lstm.num_layers, lstm.bidirectional, lstm.dropout, lstm.training
        ctx = lstm.__dict__.get(_context._ATTR)
        if ctx is not None:
            self.__dict__[_context._ATTR] = ctx
        pad = torch.nn.functional.pad

        def gate_rows(w):                     # [4 H, ...] -> [4 H4, ...]: every gate block padded
            w4 = w.reshape(4, H, *w.shape[1:])
            return pad(w4, (0, 0) * (w4.dim() - 2) + (0, H4 - H)).reshape(4 * H4, *w.shape[1:])
        for layer in range(lstm.num_layers):
            for sfx in (['', '_reverse'] if lstm.bidirectional else ['']):
                w_ih = getattr(lstm, f'weight_ih_l{layer}{sfx}')
                w_hh = getattr(lstm, f'weight_hh_l{layer}{sfx}')
                if layer > 0:                 # the input is the padded output of the layer below: [.., ndir, H4]
                    w_ih = pad(w_ih.reshape(4 * H, ndir, H), (0, H4 - H)).reshape(4 * H, ndir * H4)
                setattr(self, f'weight_ih_l{layer}{sfx}', gate_rows(w_ih))
                setattr(self, f'weight_hh_l{layer}{sfx}', gate_rows(pad(w_hh, (0, H4 - H))))
                for name in ('bias_ih', 'bias_hh'):
                    b = getattr(lstm, f'{name}_l{layer}{sfx}') if lstm.bias else w_hh.new_zeros(4 * H)
                    setattr(self, f'{name}_l{layer}{sfx}', gate_rows(b))


def _packed_lstm_padded(lstm, packed, training, hx, return_state, meta):
    """:func:`packed_lstm` for a module :class:`_PaddedLstm` covers: run the padded LSTM, slice the real units out."""
    shadow = _PaddedLstm(lstm)
    H, H4 = shadow.real_hidden, shadow.hidden_size
    ndir = 2 if lstm.bidirectional else 1
    if hx is not None:
        hx = tuple(torch.nn.functional.pad(t, (0, H4 - H)) for t in hx)
    out = packed_lstm(shadow, packed, training=lstm.training if training is None else training, hx=hx, return_state=return_state, meta=meta)
    states = None
    if return_state or hx is not None:          # (a PackedSequence is a tuple itself: ask what was asked for)
        out, states = out
        states = tuple(t[..., :H] for t in states)
    rows = out.data.shape[0]
    data = out.data.view(rows, ndir, H4)[:, :, :H].reshape(rows, ndir * H)
    out = PackedSequence(data, out.batch_sizes)
    return out if states is None else (out, states)


def packed_lstm(lstm: torch.nn.LSTM, packed: PackedSequence, training=None, hx=None, return_state=False, input_planes=None, meta=None):
    """``lstm(packed, hx)`` through the HIP recurrence.

    ``input_planes = (planes, scale value)``: ``packed.data`` once more as fp16 (hi, lo) planes in the layout of
    ``ptmi_pack_planes_n`` (written by the producer of the data, e.g. ``ops.pit_features``), with the float whose exponent gives
    their operand scale (``ops.gemm.scale_word``): the first layer's projection takes them as they lie.

    ``hx = (h_0, c_0)``, each ``[num_layers * num_directions, B, H]`` like ``torch.nn.LSTM``; gradients flow into the initial
    state and back from the final one ``(h_n, c_n)`` (persistent split kernels).  Returns the output PackedSequence, or
    ``(output, (h_n, c_n))`` when ``hx`` is given or ``return_state`` is set.
    """
    data = packed.data
    _lib.require_gpu(data)
    if not supported(lstm, data):
        raise NotImplementedError('packed_lstm needs an fp32 torch.nn.LSTM with proj_size=0: ' + str(unsupported_reason(lstm, data)))
    if isinstance(lstm, torch.nn.LSTM) and (lstm.hidden_size % 4 != 0 or not lstm.bias):
        # (a producer's fp16 planes of the input - ops.pit_features - are not used on this path: the padded first layer packs its input)
        return _packed_lstm_padded(lstm, packed, training, hx, return_state, meta)
    assert packed.sorted_indices is None, 'sequences must be sorted by length (enforce_sorted=True)'
    training = lstm.training if training is None else training
    oc = _context.effective(lstm)
    if meta is None:
        meta = pack_meta(packed.batch_sizes, data.device)
    else:       # a row-slot layout (ops.sequence.SlotLayout.meta): rows = [T, slots], several sequences end to end per slot
        assert meta.rows == data.shape[0] and hx is None and not return_state and input_planes is None, 'row-slot batches: plain calls only'
        if not (PERSISTENT and _lib.load().ptmi_lstm_split_enabled()):
            raise NotImplementedError('row-slot batches need the persistent split recurrence kernels')
    sfx = ['', '_reverse'] if lstm.bidirectional else ['']
    ndir, H = len(sfx), lstm.hidden_size
    want_state = return_state or hx is not None
    if hx is not None:
        h_all, c_all = hx
        assert h_all.shape == c_all.shape == (lstm.num_layers * ndir, meta.max_batch, H), (h_all.shape, meta.max_batch)
    h_n, c_n = [], []
    h = data.contiguous()
    all_params = [tuple((getattr(lstm, f'weight_ih_l{layer}{s}'), getattr(lstm, f'weight_hh_l{layer}{s}'),
                         getattr(lstm, f'bias_ih_l{layer}{s}'), getattr(lstm, f'bias_hh_l{layer}{s}')) for s in sfx)
                  for layer in range(lstm.num_layers)]
    flat_params = [p for ps_ in all_params for ps in ps_ for p in ps]
    if (data.is_cuda and any(_stacked_stale(ps_) for ps_ in all_params)
            and all(p.is_cuda and p.dtype == torch.float32 for p in flat_params)
            and (not (torch.is_grad_enabled() and any(p.requires_grad for p in flat_params))
                 or (oc.defer_wgrad and all(p.requires_grad and p.is_leaf and p.grad is not None for p in flat_params)))):
        # after an optimizer step: the operand forms of ALL layers on a side stream, next to whatever the main stream is
        # doing (the front-end kernels, the first projection), instead of one launch in front of every layer's projection
        pre = forked = _prep_stream(data.device)
        updated = _gemm.update_event(flat_params)
        from . import capture as _capture_
        early = _EARLY_FORK.get((data.device.type, data.device.index)) if _capture_.ACTIVE else None
        pre0 = early[1] if early else None
        early = bool(early)
        if early:
            pass                          # a captured step: forked at the head of the graph (begin_captured_step), nothing to wait for
        elif updated is not None:
            pre.wait_event(updated)       # behind the optimizer kernel, i.e. next to the step's front-end, not behind it
        else:
            pre.wait_stream(torch.cuda.current_stream(data.device))
        for layer, ps_ in enumerate(all_params):
            # (the first layer's too in a captured step: a cross-queue edge of a graph costs no host time)
            if _stacked_stale(ps_) and (layer > 0 or pre0 is not None):
                _stacked_weights(ps_, (H + 15) // 16 * 16, stream=pre if layer > 0 else pre0)
        if updated is not None or early:
            with torch.cuda.stream(pre):
                _gemm.prefetch_known(data.device, everything=early)       # the dense layers' operand forms of the last steps, behind them
        # the first layer's forms are needed at once: on the main queue itself (a cross-queue wait in front of the first
        # projection was measured to cost the main queue 110-260 us; the later layers' forms are long done when their
        # projection is reached, and a wait for a finished event costs nothing)
    else:
        forked = pre0 = None
    prev_handoff = None
    for layer in range(lstm.num_layers):
        params = all_params[layer]
        # no graph, or weight gradients accumulated in place by the backward pass (the Trainer's flat bucket): the layer
        # runs on the cached stacked / padded / transposed forms of its parameters
        graph = torch.is_grad_enabled() and any(p.requires_grad for ps in params for p in ps)
        in_place = oc.defer_wgrad and all(p.requires_grad and p.is_leaf and p.grad is not None for ps in params for p in ps)
        forms = anchor = None
        if data.is_cuda and (not graph or in_place):
            forms = _stacked_weights(params, (H + 15) // 16 * 16)
            if forms.get('ready') is not None:
                torch.cuda.current_stream(data.device).wait_event(forms['ready'])
            w_ih, bias, w_hh = forms['w_ih'], forms['bias'], forms['w_hh']
            anchor = params[0][0] if graph else None
        else:
            w_ih = torch.cat([ps[0] for ps in params], 0)
            bias = torch.cat([ps[2] + ps[3] for ps in params], 0)
            w_hh = torch.stack([ps[1] for ps in params], 0)
        if want_state:
            sl = slice(layer * ndir, (layer + 1) * ndir)
            h0 = hx[0][sl] if hx is not None else data.new_zeros(ndir, meta.max_batch, H)
            c0 = hx[1][sl] if hx is not None else data.new_zeros(ndir, meta.max_batch, H)
            if oc.grad_use_hook is not None and graph and in_place:
                oc.grad_use_hook([p for ps in params for p in ps])
            h, c = _LstmLayerFn.apply(h, w_ih, bias, w_hh, meta, h0, c0, params, layer > 0, anchor, forms, None, None, False, oc)
            prev_handoff = None
            # (h_n / c_n keep their graph, like torch.nn.LSTM's: the gradient of h_n joins this layer's output gradient, that of c_n
            #  reaches the backward kernel through `c`; modules.StatefulLSTM detaches what it carries between calls)
            hv, cv = h.view(meta.rows, ndir, H), c.view(meta.rows, ndir, H)
            h_n += [hv[meta.last_rows[d], d] for d in range(ndir)]
            c_n += [cv[meta.last_rows[d], d] for d in range(ndir)]
        else:
            out_handoff = {}
            if oc.grad_use_hook is not None and graph and in_place:
                oc.grad_use_hook([p for ps in params for p in ps])
            if layer == 0 and input_planes is not None and prev_handoff is None:
                prev_handoff = {'xplanes': input_planes}
            h = _LstmLayerFn.apply(h, w_ih, bias, w_hh, meta, None, None, params, layer > 0, anchor, forms, prev_handoff, out_handoff,
                                   layer + 1 == lstm.num_layers, oc)
            prev_handoff = out_handoff
        if lstm.dropout > 0 and training and layer + 1 < lstm.num_layers:
            h = torch.nn.functional.dropout(h, lstm.dropout, True)
            prev_handoff = None               # the next layer's input is no longer this layer's output
    from . import capture as _capture
    if _capture.ACTIVE and forked is not None:
        # a captured step: the preparation stream has forked from the capturing stream (above) and must join it again, whether or not
        # a layer has waited for its forms (by now they are long done: the wait is free)
        torch.cuda.current_stream(data.device).wait_stream(forked)
        if pre0 is not None:
            torch.cuda.current_stream(data.device).wait_stream(pre0)
    if not (torch.is_grad_enabled() and h.requires_grad):
        # inference: nobody will run Trainer.clip_grad (which reads the watchdog words of the persistent kernels during
        # training) - check them here, so that results of a timed-out launch are never returned silently
        check_errors()
    if prev_handoff and prev_handoff.get('planes') is not None:
        # the recurrence has left this very tensor as fp16 planes too: ops.linear takes them as operand A when it is handed the SAME
        # tensor object, unmodified (the record travels WITH the tensor - until round 3 it was a process global naming "the last
        # call's output", a side channel between two ops that two models or two host threads would have shared)
        setattr(h, HANDOFF_ATTR, (h._version, prev_handoff['planes'], ndir, H))
    out = PackedSequence(h, packed.batch_sizes)
    if want_state:
        return out, (torch.stack(h_n), torch.stack(c_n))
    return out
