from . import optimizer, trainer, trigger  # noqa: F401
from .trainer import Trainer, StopTraining  # noqa: F401
