from . import utils, io, transform, array  # noqa: F401
