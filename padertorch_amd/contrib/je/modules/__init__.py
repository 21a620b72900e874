from . import features  # noqa: F401
