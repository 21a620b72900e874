from .losses import *  # noqa: F401,F403
from . import losses
from . import sequence
from . import mappings
from . import scalars  # noqa: F401

from ._stft import STFT  # noqa: F401
from .einsum import *  # noqa: F401,F403
from .sequence import *  # noqa: F401,F403
from .features import pit_features  # noqa: F401
from . import gemm  # noqa: F401
from . import lstm  # noqa: F401
from . import linear  # noqa: F401
from .lstm import packed_lstm  # noqa: F401
from .unit_norm import unit_norm  # noqa: F401
