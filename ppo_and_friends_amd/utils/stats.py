"""
Device-resident running mean / variance tracker -- the stand-in for
utils/stats.py:9-94 (RunningMeanStd).  State (mean, variance float32[W], count
float64) lives in HBM; update() is two launches (batch moments, Chan merge +
integrate) and, across ranks, one all-gather of (n, mean, M2) records instead of
the reference's allgather of the raw data (stats.py:47-50).
"""
import torch

from .. import kernels as K
from . import mpi_utils


class RunningMeanStd:

    def __init__(self, shape=(), epsilon=1e-4, device="cuda"):
        self.shape = tuple(shape)
        self.width = 1
        for s in self.shape:
            self.width *= int(s)
        self.device = torch.device(device)
        self.mean_t = torch.zeros(self.width, dtype=torch.float32, device=self.device)
        self.var_t = torch.ones(self.width, dtype=torch.float32, device=self.device)
        self.count_t = torch.full((1,), float(epsilon), dtype=torch.float64, device=self.device)
        self._moments = torch.empty(1 + 2 * self.width, dtype=torch.float64, device=self.device)

    # host views with the reference's attribute names (they synchronise: not for the hot loop)
    @property
    def mean(self):
        return self.mean_t.reshape(self.shape).cpu().numpy()

    @property
    def variance(self):
        return self.var_t.reshape(self.shape).cpu().numpy()

    @property
    def count(self):
        return float(self.count_t.item())

    def batch_record(self, data):
        """(n, mean, M2) float64 record of this rank's batch ([n, W] float32 on the device)."""
        return K.batch_moments(data.reshape(-1, self.width) if self.width > 1 else data.reshape(-1),
                               self.width, self._moments)

    def integrate_records(self, records):
        """records float64 [R, 1+2W] (already gathered across ranks)."""
        K.running_moments_integrate(records.contiguous(), self.mean_t, self.var_t, self.count_t)

    def update(self, data, gather_stats=True):
        """stats.py:29-60."""
        rec = self.batch_record(data)
        if gather_stats and mpi_utils.distributed_path():
            rec = mpi_utils.allgather_records(rec)
        self.integrate_records(rec)

    def state_dict(self):
        return {"mean": self.mean, "variance": self.variance, "count": self.count}

    def load_state_dict(self, sd):
        self.mean_t.copy_(torch.as_tensor(sd["mean"], dtype=torch.float32).reshape(-1))
        self.var_t.copy_(torch.as_tensor(sd["variance"], dtype=torch.float32).reshape(-1))
        self.count_t.fill_(float(sd["count"]))
