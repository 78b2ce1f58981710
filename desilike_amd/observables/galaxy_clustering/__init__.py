from .window import (WindowedPowerSpectrumMultipoles, window_matrix_bininteg, SystematicTemplatePowerSpectrumMultipoles,
                     TopHatFiberCollisionsPowerSpectrumMultipoles)
from .power_spectrum import TracerPowerSpectrumMultipolesObservable
from .correlation_function import (WindowedCorrelationFunctionMultipoles, TracerCorrelationFunctionMultipolesObservable,
                                   SystematicTemplateCorrelationFunctionMultipoles, TopHatFiberCollisionsCorrelationFunctionMultipoles,
                                   FiberCollisionsCorrelationFunctionMultipoles, window_matrix_RR)
from .covariance import ObservablesCovarianceMatrix, BoxFootprint, CutskyFootprint, BaseFootprint
