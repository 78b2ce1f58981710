from .fisher import SNWeightedPowerSpectrumLikelihood   # noqa: F401
