"""Host -> device transfer of the NEXT batch under the current step.

The reference moves a batch inside the step (``Trainer.step``: ``model.example_to_device(example, device)``,
``padertorch/train/trainer.py:549``), i.e. the waveforms cross PCIe in front of the feature kernels (12.3 MB per
optimizer step at BASELINE configs[1]: ~0.2 ms of the step).  :class:`DevicePrefetcher` wraps any iterable of nested
examples (numpy arrays / pinned or pageable tensors) and hands out device copies that were issued one iteration early on a
copy stream of their own; the consumer's stream only waits for the copy's event.  The examples it yields are what
``example_to_device`` would have produced, so ``Trainer.step`` finds nothing left to move.
"""
import torch

from .batch import example_to_device

__all__ = ['DevicePrefetcher']


def _tensors(example):
    """Every tensor an example holds - also the ones that hang on its containers instead of sitting in them: a
    ``PaddedList``'s padded buffer and device lengths, the ``PackedLog1p`` record a feature front-end attaches to it
    (``ops.pit_features``: the packed first-layer input; its fp16 planes belong to a ring of their own, see ``release``)."""
    if torch.is_tensor(example):
        yield example
    elif isinstance(example, dict):
        for v in example.values():
            yield from _tensors(v)
    elif isinstance(example, (list, tuple)):
        for name in ('padded', 'lengths_dev'):
            t = getattr(example, name, None)
            if torch.is_tensor(t):
                yield t
        packed = getattr(example, 'packed_log1p', None)
        if packed is not None and torch.is_tensor(getattr(packed, 'data', None)):
            yield packed.data
        for v in example:
            yield from _tensors(v)


class DevicePrefetcher:
    """``for example in DevicePrefetcher(iterable, device): ...`` - ``example`` is already on ``device``.

    ``to_device``: the function that moves one example (default: :func:`example_to_device`; a model's own
    ``example_to_device`` can be passed when it does more than moving, as long as that work may run on the copy stream).
    Host tensors should be pinned (``tensor.pin_memory()``) for the copy to be asynchronous.
    """

    def __init__(self, iterable, device, to_device=None, release=None):
        """``release``: how the copy stream's memory pool learns that the consumer is done with an example.  Default:
        ``'record_stream'`` for the plain :func:`example_to_device`, ``'mark'`` for any other ``to_device`` - a function that does
        feature work allocates intermediates and writes buffers of its own on the copy stream (``ops.pit_features``: the planes ring
        of the packed first-layer input, two slots per shape and stream) which no ``record_stream`` mark reaches; only ``'mark'``
        orders their reuse behind the consumer.  An explicit ``'record_stream'`` with such a function is refused.
        ``'record_stream'``: every tensor of an example is marked as used on the consumer's stream (the caching allocator
        then records one event per tensor on THAT stream when the tensor is freed - a few microseconds of queue time each).
        ``'mark'``: no marks; instead the copy stream waits, in front of every example it prepares, for everything the consumer's stream
        has been handed so far.  Valid when every use of an example is enqueued on the consumer's stream - or synchronised into it - before
        the next example is asked for (``Trainer.train`` with one micro-step per optimizer step: ``optimizer_step`` waits for the
        weight-gradient stream).  It also orders ``to_device`` work that writes buffers of its own (``ops.pit_features``' planes) behind
        their last reader."""
        plain = to_device is None or to_device is example_to_device
        if release is None:
            release = 'record_stream' if plain else 'mark'
        assert release in ('record_stream', 'mark'), release
        if release == 'record_stream' and not plain:
            raise ValueError("DevicePrefetcher(release='record_stream') is only safe for the plain example_to_device: a to_device that "
                             "computes on the copy stream (features) needs release='mark' (the default for it)")
        self.iterable = iterable
        self.device = torch.device(device)
        self.to_device = to_device or example_to_device
        self.release = release
        self._stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None

    def _issue(self, example):
        if self._stream is None:
            return self.to_device(example, self.device), None
        if self.release == 'mark':
            mark = torch.cuda.Event()
            mark.record(torch.cuda.current_stream(self.device))
            self._stream.wait_event(mark)
        with torch.cuda.stream(self._stream):
            moved = self.to_device(example, self.device)
            done = torch.cuda.Event()
            done.record(self._stream)
        return moved, done

    def __iter__(self):
        it = iter(self.iterable)
        try:
            nxt = self._issue(next(it))
        except StopIteration:
            return
        while nxt is not None:
            moved, done = nxt
            try:
                nxt = self._issue(next(it))       # the copy of the following batch runs under this batch's step
            except StopIteration:
                nxt = None
            if done is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(done)
                if self.release == 'record_stream':
                    for t in _tensors(moved):
                        t.record_stream(cur)      # allocated on the copy stream's pool, used (and freed) on the consumer's
            yield moved

    def __len__(self):
        return len(self.iterable)
