from functools import cached_property  # noqa: F401
