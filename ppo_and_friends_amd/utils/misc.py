"""
RunningStatNormalizer -- stand-in for utils/misc.py:61-172, on the device.
normalize() = stats update (optional) + (x - mean) / sqrt(var + eps)   misc.py:84-111
denormalize() = mean + x * sqrt(var + eps)                             misc.py:113-128
"""
import os
import pickle

import torch

from .. import kernels as K
from . import mpi_utils
from .reference_io import dump_running_stats, load_running_stats
from .stats import RunningMeanStd


class RunningStatNormalizer:

    def __init__(self, name, device, test_mode=False, epsilon=1e-8):
        self.device = torch.device(device)
        self.name = name
        self.test_mode = test_mode
        self.running_stats = RunningMeanStd(device=self.device)
        self.epsilon = float(epsilon)

    def normalize(self, data, update_stats=True, gather_stats=True, out=None):
        if update_stats:
            self.running_stats.update(data, gather_stats)
        return K.normalize(data.contiguous(), self.running_stats.mean_t, self.running_stats.var_t,
                           self.epsilon, out=out)

    def denormalize(self, data, out=None):
        return K.denormalize(data.contiguous(), self.running_stats.mean_t,
                             self.running_stats.var_t, self.epsilon, out=out)

    def save_info(self, path):
        """misc.py:130-145: `<name>_stats_<rank>.pickle` (a plain dict of numpy state here)."""
        if self.test_mode:
            return
        f = os.path.join(path, "{}_stats_{}.pickle".format(self.name, mpi_utils.get_rank()))
        with open(f, "wb") as fh:
            dump_running_stats(self.running_stats.state_dict(), fh)     # a reference RunningMeanStd object

    def load_info(self, path):
        """misc.py:147-172 incl. the rank-0 fallback when restarting with more ranks."""
        r = 0 if self.test_mode else mpi_utils.get_rank()
        f = os.path.join(path, "{}_stats_{}.pickle".format(self.name, r))
        if not os.path.exists(f):
            f = os.path.join(path, "{}_stats_0.pickle".format(self.name))
        with open(f, "rb") as fh:
            self.running_stats.load_state_dict(load_running_stats(fh))
