from . import new_subdir  # noqa: F401
