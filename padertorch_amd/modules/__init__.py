from . import normalization, recurrent  # noqa: F401
from .normalization import Normalization, InputNormalization, normalize  # noqa: F401
from .recurrent import StatefulLSTM  # noqa: F401
