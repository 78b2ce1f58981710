from .power_template import (BasePowerSpectrumTemplate, FixedPowerSpectrumTemplate, StandardPowerSpectrumTemplate,
                             ShapeFitPowerSpectrumTemplate, BAOPowerSpectrumTemplate, TurnOverPowerSpectrumTemplate, BandVelocityPowerSpectrumTemplate, find_turn_over)
from .full_shape import (KaiserTracerPowerSpectrumMultipoles, SimpleTracerPowerSpectrumMultipoles, EFTLikeKaiserTracerPowerSpectrumMultipoles, KaiserTracerCorrelationFunctionMultipoles,
                         EFTLikeKaiserTracerCorrelationFunctionMultipoles, LPTVelocileptorsTracerPowerSpectrumMultipoles,
                         REPTVelocileptorsTracerPowerSpectrumMultipoles, EmulatedTracerPowerSpectrumMultipoles,
                         LPTVelocileptorsTracerCorrelationFunctionMultipoles, REPTVelocileptorsTracerCorrelationFunctionMultipoles,
                         TNSTracerPowerSpectrumMultipoles, EFTLikeTNSTracerPowerSpectrumMultipoles, TNSTracerCorrelationFunctionMultipoles,
                         EFTLikeTNSTracerCorrelationFunctionMultipoles, PNGTracerPowerSpectrumMultipoles, PNGTracerVelocityPowerSpectrumMultipoles)
from .bao import (DampedBAOWigglesTracerPowerSpectrumMultipoles, DampedBAOWigglesTracerCorrelationFunctionMultipoles,
                  ResummedBAOWigglesTracerPowerSpectrumMultipoles, ResummedBAOWigglesTracerCorrelationFunctionMultipoles,
                  SimpleBAOWigglesTracerPowerSpectrumMultipoles, SimpleBAOWigglesTracerCorrelationFunctionMultipoles,
                  FlexibleBAOWigglesTracerPowerSpectrumMultipoles, FlexibleBAOWigglesTracerCorrelationFunctionMultipoles)
