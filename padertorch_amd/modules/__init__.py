from . import fully_connected, normalization, recurrent  # noqa: F401
from .fully_connected import fully_connected_stack  # noqa: F401
from .normalization import Normalization, InputNormalization, normalize  # noqa: F401
from .recurrent import StatefulLSTM  # noqa: F401
