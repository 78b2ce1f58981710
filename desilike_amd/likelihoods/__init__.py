from .base import BaseLikelihood, BaseGaussianLikelihood, ObservablesGaussianLikelihood, SumLikelihood
