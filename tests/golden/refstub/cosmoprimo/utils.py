def flatarray(*args, **kwargs):
    def wrapper(func):
        return func
    return wrapper


def addproperty(*args, **kwargs):
    def wrapper(cls):
        return cls
    return wrapper
