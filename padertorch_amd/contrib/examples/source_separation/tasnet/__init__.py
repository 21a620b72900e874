from .tas_coders import StftEncoder, IstftDecoder  # noqa: F401
from .loss import tasnet_loss  # noqa: F401
