"""Empty stand-in for the un-vendored third-party package ``lsstypes`` (test infrastructure only)."""


class CovarianceMatrix(object):
    pass


class WindowMatrix(object):
    pass


class ObservableTree(object):
    pass


class ObservableLeaf(object):
    pass


class Mesh2SpectrumPoles(object):
    pass


class Mesh2SpectrumPole(object):
    pass


def read(*args, **kwargs):
    raise NotImplementedError
