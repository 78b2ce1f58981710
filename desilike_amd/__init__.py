"""MI355X-native implementation of desilike's theory -> observable -> Gaussian-likelihood hot path.

Public surface mirrors desilike's for this path (same class / argument / parameter names):

    from desilike_amd import vmap
    from desilike_amd.theories.galaxy_clustering import ShapeFitPowerSpectrumTemplate, KaiserTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
"""
from ._lib import Context, LibraryError  # noqa: F401
from .base import BaseCalculator, PipelineError, vmap  # noqa: F401
from .parameter import Parameter, ParameterPrior, ParameterCollection, Samples  # noqa: F401

__version__ = '0.1.0'
