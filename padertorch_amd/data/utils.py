"""``collate_fn`` (``padertorch/data/utils.py:21-69``): list of examples -> example of lists."""

__all__ = ['collate_fn', 'row_slot_batches']


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
