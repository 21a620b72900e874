from .modul import MaskEstimator, MaskKeys  # noqa: F401
