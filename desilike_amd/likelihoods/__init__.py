from .base import BaseLikelihood, BaseGaussianLikelihood, ObservablesGaussianLikelihood, SumLikelihood
from . import galaxy_clustering   # noqa: F401
