import numpy as np

from ..spaces import Box


def batch_space(space, n=1):
    return Box(np.repeat(space.low[None], n, 0), np.repeat(space.high[None], n, 0), dtype=space.dtype)
