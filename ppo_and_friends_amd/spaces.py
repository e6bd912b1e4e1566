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


class MultiDiscrete:
    """gymnasium.spaces.MultiDiscrete attributes used on this path: nvec, start, shape, dtype."""

    def __init__(self, nvec, dtype=np.int64, start=None):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.start = np.zeros_like(self.nvec) if start is None else np.asarray(start, dtype=np.int64)
        self.shape = tuple(self.nvec.shape)
        self.dtype = np.dtype(dtype)

    def seed(self, seed=None):
        return [seed]


class MultiBinary:
    """gymnasium.spaces.MultiBinary attributes used on this path: n, shape, dtype (int8, as gymnasium's)."""

    def __init__(self, n):
        self.n = int(n)
        self.shape = (self.n,)
        self.dtype = np.dtype(np.int8)

    def seed(self, seed=None):
        return [seed]


def is_multi_discrete(space):
    return hasattr(space, "nvec")


def is_multi_binary(space):
    return type(space).__name__ == "MultiBinary"


def get_agent_shared_space(space, num_agents):
    """
    utils/misc.py:349-396: the space spanning all agents of a group (agent-shared ICM): Box -> the
    flattened concatenation, Discrete -> MultiDiscrete([n] * A), MultiDiscrete -> tiled nvec.
    """
    if is_multi_discrete(space):
        return MultiDiscrete(np.tile(space.nvec, num_agents), dtype=space.dtype,
                             start=np.tile(getattr(space, "start", np.zeros_like(space.nvec)), num_agents))
    if is_discrete(space):
        return MultiDiscrete([int(space.n)] * num_agents, dtype=space.dtype)
    if np.issubdtype(space.dtype, np.floating):
        low = np.tile(np.asarray(space.low, dtype=space.dtype).reshape(-1), num_agents)
        high = np.tile(np.asarray(space.high, dtype=space.dtype).reshape(-1), num_agents)
        return Box(low, high, low.shape, space.dtype)
    raise NotImplementedError(f"get_agent_shared_space: unsupported space {type(space)}")


def is_discrete(space):
    return hasattr(space, "n") and not hasattr(space, "nvec") and np.issubdtype(space.dtype, np.integer) \
        and not is_multi_binary(space)


def get_space_dtype_str(space):
    """utils/misc.py:17-46 for the space kinds on this path."""
    if is_multi_discrete(space):
        return "multi-discrete"
    if is_multi_binary(space):
        return "multi-binary"
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
    """utils/misc.py:295-346: Discrete -> (n,), MultiDiscrete -> (sum(nvec),), Box -> its shape."""
    if is_multi_discrete(space):
        return (int(np.sum(space.nvec)),)
    return (space.n,) if (is_discrete(space) or is_multi_binary(space)) else tuple(space.shape)
