from . import regression
from . import source_separation
from .regression import *  # noqa: F401,F403
from .source_separation import *  # noqa: F401,F403
from .source_separation import pit_mse_ips_losses, dc_loss_batched  # noqa: F401
