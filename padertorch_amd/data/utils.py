"""``collate_fn`` (``padertorch/data/utils.py:21-69``): list of examples -> example of lists."""

__all__ = ['StaticSlotBatcher', 'collate_fn', 'row_slot_batches']


def collate_fn(batch):
    """Moves the list inside of dict / dataclass recursively.

    >>> collate_fn([{'a': 1}, {'a': 2}])
    {'a': [1, 2]}
    >>> collate_fn([{'a': {'b': [1, 2]}}, {'a': {'b': [3, 4]}}])
    {'a': {'b': [[1, 2], [3, 4]]}}
    """
    assert isinstance(batch, (tuple, list)), (type(batch), batch)
    first = batch[0]
    if isinstance(first, dict):
        for b in batch[1:]:
            assert first.keys() == b.keys(), batch
        return first.__class__({k: collate_fn(batch.__class__([b[k] for b in batch])) for k in first})
    if hasattr(first, '__dataclass_fields__'):
        return first.__class__(**{k: collate_fn(batch.__class__([getattr(b, k) for b in batch]))
                                  for k in first.__dataclass_fields__})
    return batch


def row_slot_batches(examples, row_slots=32, fill=2.0, key='num_samples', collate=True):
    """Batches for a model that runs ragged batches on row slots (``model.row_slots``, ``ops.sequence.SlotLayout``): the recurrences
    cost ~ sum(lengths) / row_slots time steps once about two sequences lie end to end in every slot, so a batch takes
    ``round(fill * row_slots)`` examples of the stream instead of ``row_slots``; inside a batch the examples are sorted by
    descending length like ``Sorter`` does (``padertorch/data/batch.py:133-158``; the models' batch contract).  A generator over
    the collated batches (``collate_fn``), in stream order; the last batch takes what is left.

    >>> exs = [{'num_samples': n} for n in (5, 9, 2, 7, 3)]
    >>> [b['num_samples'] for b in row_slot_batches(exs, row_slots=2, fill=1.5)]
    [[9, 5, 2], [7, 3]]
    """
    import operator
    size = max(1, int(round(fill * row_slots)))
    get = key if callable(key) else operator.itemgetter(key)
    group = []
    for ex in examples:
        group.append(ex)
        if len(group) == size:
            group.sort(key=get, reverse=True)
            yield collate_fn(group) if collate else group
            group = []
    if group:
        group.sort(key=get, reverse=True)
        yield collate_fn(group) if collate else group


class StaticSlotBatcher:
    """Collated waveform batches (``y``: list of ``(N_b,)``, ``s``: list of ``(K, N_b)``, ``num_samples``; the keys of the reference's
    ``pre_batch_transform`` input, ``pit/data.py:49-77``) -> examples whose length pattern is DEVICE data, so that one captured optimizer
    step serves all of them (``ops.sequence.StaticSlots``, ``train.graphed``; the models read ``batch['slots']``)::

        batcher = StaticSlotBatcher(examples=64, slots=32, max_samples=6 * 8000, device='cuda:0')
        for batch in row_slot_batches(stream, row_slots=32, fill=2.0):
            example = batcher(batch)          # dict(y [B, max_samples], s [B, K, max_samples], num_samples int32 [B], slots)

    ``steps``: the grid's capacity in time steps, or a LIST of capacities (buckets: a batch takes the smallest grid it fits; default: enough for every batch whose frames sum to ``headroom`` x the mean of a
    U[max / 2, max] length distribution, rounded up to 8).  A batch that does not fit - more examples, a longer example, more frames than
    the grid has room for - comes back as it is (the model then takes its host-side route: ``model.row_slots`` / PackedSequence, eagerly);
    ``refused`` counts them.  Two layouts are used in turn: the tables of batch i + 1 may be written while batch i's step still reads its
    own.  The padded tensors are made on ``device`` from the host tensors of the batch (pinned memory makes the copies asynchronous).
    """

    def __init__(self, examples, slots, max_samples, device, stft=None, steps=None, headroom=1.08):
        import torch
        from ..ops import STFT
        from ..ops.sequence import StaticSlots
        self.stft = stft if stft is not None else STFT(512, 128)
        self.examples, self.slots, self.max_samples = int(examples), int(slots), int(max_samples)
        self.device = torch.device(device)
        self.padded_time = int(self.stft.samples_to_frames(self.max_samples))
        if steps is None:
            mean = 0.75 * self.padded_time * self.examples / self.slots
            steps = max(self.padded_time, int(-(-headroom * mean // 8) * 8))
        # several capacities = BUCKETS: a batch takes the smallest grid it fits (a step costs its grid's time steps, used or idle), and
        # every bucket is one example signature, i.e. one captured graph (set ``trainer.graph_capacity`` to the number of buckets: two are kept by default)
        self.buckets = sorted({int(v) for v in (steps if isinstance(steps, (list, tuple)) else [steps])})
        self.steps = self.buckets[-1]
        self._rings = {cap: [StaticSlots(self.examples, self.slots, cap, self.padded_time, self.device) for _ in range(2)] for cap in self.buckets}
        self._turn = {cap: 0 for cap in self.buckets}
        self.refused = 0
        self.taken = {cap: 0 for cap in self.buckets}

    def frames_of(self, num_samples):
        return [int(self.stft.samples_to_frames(int(n))) for n in num_samples]

    def __call__(self, batch):
        import numpy as np
        import torch
        from ..ops.sequence import SlotLayout
        num_samples = [int(n) for n in batch['num_samples']]
        frames = self.frames_of(num_samples)
        ok = len(frames) == self.examples and max(num_samples) <= self.max_samples and min(frames) >= 1 and sum(frames) <= self.steps * self.slots
        cap = None
        if ok:
            probe = SlotLayout.__new__(SlotLayout)
            SlotLayout._place(probe, frames, self.slots)
            cap = next((c for c in self.buckets if probe.T <= c), None)
        if cap is None:
            self.refused += 1
            return batch
        self.taken[cap] += 1
        B, N = self.examples, self.max_samples

        def padded(rows, lead):
            out = torch.zeros((B,) + lead + (N,), dtype=torch.float32)
            for b, r in enumerate(rows):
                r = torch.as_tensor(np.asarray(r) if not torch.is_tensor(r) else r, dtype=torch.float32)
                out[b, ..., :r.shape[-1]] = r
            return out.pin_memory().to(self.device, non_blocking=True) if self.device.type == 'cuda' else out
        out = {k: v for k, v in batch.items() if k not in ('y', 's', 'num_samples')}
        out['y'] = padded(batch['y'], ())
        if batch.get('s') is not None:
            K = int(np.asarray(batch['s'][0]).shape[0]) if not torch.is_tensor(batch['s'][0]) else int(batch['s'][0].shape[0])
            out['s'] = padded(batch['s'], (K,))
        out['num_samples'] = torch.tensor(num_samples, dtype=torch.int32).to(self.device)
        self._turn[cap] ^= 1
        out['slots'] = self._rings[cap][self._turn[cap]].set(frames)
        return out
