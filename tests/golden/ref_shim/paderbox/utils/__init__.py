from . import nested, mapping  # noqa: F401
