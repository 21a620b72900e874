from .model import PermutationInvariantTrainingModel  # noqa: F401
