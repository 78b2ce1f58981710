class Samples(object):
    pass
