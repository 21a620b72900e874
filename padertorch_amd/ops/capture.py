"""What the ops need to know while an optimizer step is CAPTURED into a hipGraph (``train.graphed.GraphedStep``).

The eager step's host side orders itself with state that lives ACROSS steps: parameter-derived operand forms cached per
``Parameter._version``, the preparation stream waiting for the PREVIOUS step's optimizer event, weight-gradient events of the last
step, words of zeroed buffers handed out one by one.  Inside a capture every dependency has to be an edge of the graph: a stream
may only wait for events recorded in the same capture, every value a replay reads must be (re)made by a node of the graph, and a
buffer a replay accumulates into has to be zeroed by the graph itself.  :func:`capture_mode` empties those caches on the way in (so
that the forms are rebuilt INSIDE the capture, on streams that fork from and join the capturing stream) and on the way out (their
entries hold captured events and tensors of the graph's memory pool: nothing an eager step may wait for or overwrite).
"""
import contextlib

import torch

__all__ = ['ACTIVE', 'capture_mode', 'zero_word', 'zero_block', 'reset_step_caches']

#: True while a step is being captured (host thread of the capture AND autograd's worker threads read it)
ACTIVE = False

_zero_blocks = []       # [[block of zeroed int32 words (allocated and zero-filled INSIDE the capture), words handed out, stream]]
_shared = []            # [block zero-filled by the FIRST node of the capture, words handed out]: every stream of the capture forks behind it


def zero_block(device, words=4096):
    """Call as the first thing inside ``torch.cuda.graph(...)``: ONE fill node at the head of the graph zeroes the words every
    accumulating epilogue of the step starts from, whatever stream it runs on - a stream joins a capture only by waiting for an
    event recorded behind this node.  (Round 5 filled a block per stream and pool, 64 / 256 words each: five or six 5 us fill
    launches per replay, most of them on the critical path between the dense layers.)"""
    assert ACTIVE
    _shared[:] = [torch.zeros(words, dtype=torch.int32, device=device), 0]


def zero_word(device, block_words=64):
    """A zeroed int32 word for an accumulating epilogue (``ops.gemm.zero_word`` / ``ops.library._zero_word``) during a capture: from
    blocks that are allocated - and therefore zero-filled - by nodes of the graph, so that every replay starts from zero.  (The
    eager pools hand out words of blocks that were filled once, when they were allocated; a replay would accumulate into the last
    replay's maximum.)"""
    # (a block belongs to the stream it was zero-filled on: a word handed to another stream's kernel would not be ordered behind the fill)
    if _shared and _shared[0].device == device and _shared[1] < _shared[0].numel():
        _shared[1] += 1
        return _shared[0][_shared[1] - 1:_shared[1]]
    stream = torch.cuda.current_stream(device).cuda_stream
    ent = next((e for e in reversed(_zero_blocks) if e[2] == stream and e[0].device == device and e[1] < e[0].numel()), None)
    if ent is None:
        ent = [torch.zeros(block_words, dtype=torch.int32, device=device), 0, stream]
        _zero_blocks.append(ent)
    ent[1] += 1
    return ent[0][ent[1] - 1:ent[1]]


def reset_step_caches():
    """Forget everything the ops have cached from earlier steps that orders work (events) or stands for parameter values (forms)."""
    from . import gemm as _gemm, lstm as _lstm
    known = dict(_gemm._KNOWN)      # which forms the dense layers' parameters are used in: knowledge, not state - kept
    _gemm.invalidate()              # operand scales / planes / stacked LSTM forms / the optimizer's update event
    _gemm._KNOWN.update(known)
    _lstm._EARLY_FORK.clear()
    _gemm._WAITED.clear()
    _lstm._WGRAD_DONE.clear()


@contextlib.contextmanager
def capture_mode():
    """Everything between ``__enter__`` and ``__exit__`` runs with :data:`ACTIVE` set and starts (and leaves) with empty step caches."""
    global ACTIVE
    assert not ACTIVE, 'nested capture'
    reset_step_caches()
    del _zero_blocks[:]
    del _shared[:]
    ACTIVE = True
    try:
        yield
    finally:
        ACTIVE = False
        if _shared:
            _zero_blocks.append([_shared[0], _shared[0].numel(), None])       # (stays alive with the graph like the others)
            del _shared[:]
        reset_step_caches()
        # (the zero blocks stay alive with the graph: its nodes write them)
