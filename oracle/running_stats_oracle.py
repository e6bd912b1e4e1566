"""
TEST INFRASTRUCTURE (see oracle/__init__.py) -- numpy restatement of the running
mean/variance tracker and the value normaliser.  PINNED by tests/golden/g4.

Follows /root/reference/utils/stats.py:9-94 (RunningMeanStd) and
/root/reference/utils/misc.py:61-128 (RunningStatNormalizer.normalize/denormalize).
"""
import numpy as np


class RunningMeanStd:
    """stats.py:9-94.  mean/variance float32 arrays, count a Python float (1e-4)."""

    def __init__(self, shape=(), epsilon=1e-4):
        self.mean = np.zeros(shape, dtype=np.float32)
        self.variance = np.ones(shape, dtype=np.float32)
        self.count = epsilon

    def update(self, data, gathered=None):
        # stats.py:47-50: with >1 ranks the raw data of every rank is gathered
        # first; the caller passes the list of per-rank arrays as `gathered`.
        if gathered is not None:
            data = np.concatenate(gathered)
        data = np.asarray(data)
        batch_mean = np.mean(data, axis=0)          # float32 in -> float32 out
        batch_var = np.var(data, axis=0)            # population variance
        self.integrate(batch_mean, batch_var, data.shape[0])

    def integrate(self, batch_mean, batch_variance, batch_size):
        # stats.py:73-94 (Chan et al. parallel merge)
        delta = batch_mean - self.mean
        new_count = self.count + batch_size
        self.mean = self.mean + (delta * (batch_size / new_count))
        m_a = self.variance * self.count
        m_b = batch_variance * batch_size
        m_2 = m_a + m_b + np.square(delta) * self.count * batch_size / \
            (self.count + batch_size)
        self.variance = m_2 / (self.count + batch_size)
        self.count += batch_size


def normalize(data, mean, variance, epsilon=1e-8):
    """misc.py:106-111: (x - mean) / sqrt(var + eps), all float32."""
    d = np.asarray(data, dtype=np.float32)
    return (d - np.float32(mean)) / np.sqrt(np.float32(variance) + np.float32(epsilon))


def denormalize(data, mean, variance, epsilon=1e-8):
    """misc.py:124-128: mean + x * sqrt(var + eps), all float32."""
    d = np.asarray(data, dtype=np.float32)
    return np.float32(mean) + d * np.sqrt(np.float32(variance) + np.float32(epsilon))
