"""``collate_fn`` (``padertorch/data/utils.py:21-69``): list of examples -> example of lists."""

__all__ = ['collate_fn']


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
