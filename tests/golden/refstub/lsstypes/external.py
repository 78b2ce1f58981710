def from_pypower(*args, **kwargs):
    raise NotImplementedError


def from_pycorr(*args, **kwargs):
    raise NotImplementedError
