"""Stand-in for the un-vendored third-party package ``lsstypes`` (test infrastructure only): plain containers that keep what they are given -- the reference
wraps its results in them (e.g. ``CovarianceMatrix(value=..., observable=...)``, observables/galaxy_clustering/covariance.py:340-342); nothing is computed here."""


class _Container(object):

    def __init__(self, *args, **kwargs):
        self.args = args
        self.__dict__.update(kwargs)


class CovarianceMatrix(_Container):
    pass


class WindowMatrix(_Container):
    pass


class ObservableTree(_Container):
    pass


class ObservableLeaf(_Container):
    pass


class Mesh2SpectrumPoles(_Container):
    pass


class Mesh2SpectrumPole(_Container):
    pass


class Count2CorrelationPoles(_Container):
    pass


class Count2CorrelationPole(_Container):
    pass


def read(*args, **kwargs):
    raise NotImplementedError
