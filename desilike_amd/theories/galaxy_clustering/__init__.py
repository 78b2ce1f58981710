from .power_template import (BasePowerSpectrumTemplate, FixedPowerSpectrumTemplate, StandardPowerSpectrumTemplate,
                             ShapeFitPowerSpectrumTemplate, BAOPowerSpectrumTemplate)
from .full_shape import KaiserTracerPowerSpectrumMultipoles, EFTLikeKaiserTracerPowerSpectrumMultipoles
from .bao import DampedBAOWigglesTracerPowerSpectrumMultipoles, DampedBAOWigglesTracerCorrelationFunctionMultipoles
