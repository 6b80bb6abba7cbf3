registry = {}


def register(id, **kwargs):
    registry[id] = kwargs


def make(id, **kwargs):
    raise NotImplementedError


def spec(id):
    return registry[id]
