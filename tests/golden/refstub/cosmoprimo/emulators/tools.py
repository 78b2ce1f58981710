class Emulator(object):
    pass


class PointEmulatorEngine(object):
    pass


class TaylorEmulatorEngine(object):
    pass


class MLPEmulatorEngine(object):
    pass


class Operation(object):
    pass


class PCAOperation(object):
    pass
