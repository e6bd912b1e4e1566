"""
Shape metadata for observation / action spaces.

The reference consumes gymnasium spaces only for their dims on this path
(utils/misc.py:17-46,200-346).  gymnasium is not a dependency here: these two
classes carry the same attributes (`shape`, `dtype`, `n`, `low`, `high`,
`seed`), and real gymnasium Box/Discrete objects are accepted wherever these are
(duck-typed through the helpers below).
"""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def seed(self, seed=None):
        return [seed]


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def seed(self, seed=None):
        return [seed]


def is_discrete(space):
    return hasattr(space, "n") and not hasattr(space, "nvec") and np.issubdtype(space.dtype, np.integer)


def get_space_dtype_str(space):
    """utils/misc.py:17-46 for the two space kinds on this path."""
    if is_discrete(space):
        return "discrete"
    if np.issubdtype(space.dtype, np.floating):
        return "continuous"
    return "unknown"


def get_space_shape(space):
    """utils/misc.py:200-247: Discrete -> (1,), Box -> its shape."""
    return (1,) if is_discrete(space) else tuple(space.shape)


def get_flattened_space_length(space):
    return int(np.prod(get_space_shape(space)))


def get_action_prediction_shape(space):
    """utils/misc.py:295-346: Discrete -> (n,), Box -> its shape."""
    return (space.n,) if is_discrete(space) else tuple(space.shape)
