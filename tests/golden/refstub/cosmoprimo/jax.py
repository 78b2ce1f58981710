class Interpolator1D(object):
    pass


class Interpolator2D(object):
    pass
