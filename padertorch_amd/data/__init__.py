from . import batch, utils  # noqa: F401
from .batch import example_to_device, Sorter  # noqa: F401
from .utils import collate_fn, row_slot_batches, StaticSlotBatcher  # noqa: F401
from .prefetch import DevicePrefetcher  # noqa: F401
