class PowerSpectrumInterpolator1D(object):
    pass


class PowerSpectrumInterpolator2D(object):
    pass
