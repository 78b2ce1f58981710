from .window import (WindowedPowerSpectrumMultipoles, window_matrix_bininteg, SystematicTemplatePowerSpectrumMultipoles,
                     TopHatFiberCollisionsPowerSpectrumMultipoles)
from .power_spectrum import TracerPowerSpectrumMultipolesObservable
from .correlation_function import (WindowedCorrelationFunctionMultipoles, TracerCorrelationFunctionMultipolesObservable,
                                   SystematicTemplateCorrelationFunctionMultipoles, TopHatFiberCollisionsCorrelationFunctionMultipoles,
                                   FiberCollisionsCorrelationFunctionMultipoles)
from .covariance import ObservablesCovarianceMatrix, BoxFootprint, CutskyFootprint, BaseFootprint
