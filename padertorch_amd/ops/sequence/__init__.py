from . import mask
from . import pack_module
from . import pointwise

from .pack_module import *  # noqa: F401,F403
from .pack_module import as_padded  # noqa: F401
from .pointwise import *  # noqa: F401,F403
from .mask import compute_mask  # noqa: F401
from .slots import SlotLayout, StaticSlots  # noqa: F401
