from . import module_stft  # noqa: F401
from .module_stft import stft, istft  # noqa: F401
