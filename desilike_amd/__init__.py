"""MI355X-native implementation of desilike's theory -> observable -> Gaussian-likelihood hot path."""
from ._lib import Context, LibraryError  # noqa: F401

__version__ = '0.1.0'
