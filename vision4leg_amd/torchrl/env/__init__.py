from .base_wrapper import Normalizer, NormObs, NormObsWithImg  # noqa: F401
