"""
Status-driven schedules for the host scalars of the update -- learning rates, entropy / intrinsic-reward
weights, bootstrap clip bounds -- and the policy freeze cycle.  Stand-in for the reference's
utils/schedulers.py (same class names, constructor arguments and call protocol, so a reference
`policy_settings` dict that passes e.g. `lr=LinearScheduler("timesteps", ...)` works unchanged):

    CallableValue            schedulers.py:11-29     a constant behind the scheduler protocol
    StatusScheduler          :257-314                base: reads status_dict[status_preface][status_key]
    LogScheduler             :317-341                max - log(step) * (max - min) / log(status_max), clamped
    LinearScheduler          :344-368                max - step * (max - min) / status_max, clamped
    LinearStepScheduler      :371-445                piecewise constant, advanced by compare_fn(step, trigger)
    ChangeInStateScheduler   :448-495                compare_fn(current, cached) with per-call or persistent caching
    FreezeCyclingScheduler   :31-254                 one policy group trains while the others are frozen

A schedule is a pure function of the status dict the trainer maintains (`iteration`, `timesteps`, scores ...);
PPO.learn evaluates them once per iteration (ppo.py:2254, 1406-1428) and the resulting floats reach the
kernels as launch arguments (entropy weight, clip bounds) or through a device scalar (learning rate).
Pinned against the reference by tests/golden/g7_schedulers.npz.
"""
import os

import numpy as np

from .mpi_utils import get_rank, rank_print


class CallableValue:
    """A constant with the scheduler protocol (finalize / save_info / load_info / call)."""

    def __init__(self, val):
        self.val = val

    @property
    def value(self):
        return self.val

    def finalize(self, *args, **kw_args):
        pass

    def save_info(self, *args, **kw_args):
        pass

    def load_info(self, *args, **kw_args):
        pass

    def __call__(self, *args, **kw_args):
        return self.val


class StatusScheduler:
    """Base of the schedules keyed on one numeric entry of the status dict."""

    def __init__(self, status_key, status_preface="global status"):
        self.status_key = status_key
        self.status_preface = status_preface
        self.finalized = False
        self.status_dict = None

    def finalize(self, status_dict):
        self.status_dict = status_dict
        if self.status_key != "":
            section = status_dict[self.status_preface]
            if self.status_key not in section:
                raise KeyError(f"status_key {self.status_key!r} does not exist in status_dict[{self.status_preface!r}]; "
                               f"available keys: {list(section.keys())}")
            float(section[self.status_key])               # the value must be a number (raises otherwise)
        self.finalized = True

    def _get_step(self):
        assert self.finalized, "scheduler used before finalize(status_dict)"
        if self.status_key == "":
            return 0
        return self.status_dict[self.status_preface][self.status_key]

    def __call__(self, *args, **kw_args):
        raise NotImplementedError


class LogScheduler(StatusScheduler):
    """Logarithmic decay from max_value (step <= 1) to min_value (step >= status_max)."""

    def __init__(self, status_key, status_max, max_value, min_value, **kw_args):
        super().__init__(status_key=status_key, **kw_args)
        self.status_max, self.max_value, self.min_value = status_max, max_value, min_value
        self.numerator = np.log(self.status_max) / (max_value - min_value)

    def __call__(self):
        value = self.max_value - (np.log(self._get_step()) / self.numerator)
        return max(min(value, self.max_value), self.min_value)


class LinearScheduler(StatusScheduler):
    """Linear decay from max_value at step 0 to min_value at step status_max, clamped to that range."""

    def __init__(self, status_key, status_max, max_value, min_value, **kw_args):
        super().__init__(status_key=status_key, **kw_args)
        self.status_max, self.max_value, self.min_value = status_max, max_value, min_value

    def __call__(self):
        value = self.max_value - (self._get_step() * ((self.max_value - self.min_value) / self.status_max))
        return min(max(value, self.min_value), self.max_value)


class LinearStepScheduler(StatusScheduler):
    """
    initial_value until compare_fn(status, status_triggers[0]) first holds; from then on step_values[k], where k
    advances past every trigger the status has crossed (it never moves back).  The very first iteration always
    returns initial_value: the status dict has not been written yet.
    """

    def __init__(self, initial_value, status_key, status_triggers, step_values, compare_fn=np.greater, **kw_args):
        super().__init__(status_key=status_key, **kw_args)
        if len(status_triggers) == 0:
            raise ValueError("LinearStepScheduler requires at least one status trigger")
        if len(status_triggers) != len(step_values):
            raise ValueError("status_triggers and step_values must contain the same number of entries")
        self.initial_value = initial_value
        self.status_triggers, self.step_values = status_triggers, step_values
        self.max_idx = len(step_values) - 1
        self.range_idx = -1
        self.compare_fn = compare_fn

    def __call__(self):
        if self.status_dict["global status"]["iteration"] == 0:
            return self.initial_value
        step = self._get_step()
        while self.range_idx < self.max_idx and self.compare_fn(step, self.status_triggers[self.range_idx + 1]):
            self.range_idx += 1
        return self.initial_value if self.range_idx < 0 else self.step_values[self.range_idx]


class ChangeInStateScheduler(StatusScheduler):
    """
    compare_fn(current_status, cached_status); the cache follows the status every call, or -- persistent -- only
    when the comparison holds.  The first call caches and returns False.
    """

    def __init__(self, status_key, compare_fn=np.not_equal, persistent=False, **kw_args):
        super().__init__(status_key=status_key, **kw_args)
        self.compare_fn, self.persistent = compare_fn, persistent
        self.prev_status = None

    def __call__(self):
        step = self._get_step()
        if self.prev_status is None:
            self.prev_status = step
            return False
        changed = self.compare_fn(step, self.prev_status)
        if not self.persistent or changed:
            self.prev_status = step
        return changed


class FreezeCyclingScheduler:
    """
    Freeze cycling: all policy groups but one are frozen (PPOPolicy.freeze: no weight updates), and every
    `iterations` iterations the active group moves on.  Policies named in no group form groups of their own.
    A policy is saved under the iteration number as tag whenever it is frozen.  Cycling starts at iteration
    delay + 1.  `active_idx` survives restarts through FreezeCyclingScheduler.yaml in the state path.
    """

    def __init__(self, policy_groups, iterations, delay=-1, verbose=False):
        self.policy_groups = [list(g) for g in policy_groups]
        self.iterations, self.delay, self.verbose = iterations, delay, verbose
        self.num_groups = len(self.policy_groups)
        self.active_idx = 0
        self.status_dict = self.policies = self.policy_ids = self.state_path = None
        self.finalized = False

    def finalize(self, state_path, status_dict, policies):
        self.state_path, self.status_dict, self.policies = state_path, status_dict, policies
        self.policy_ids = tuple(policies.keys())
        grouped = set()
        for group in self.policy_groups:
            for policy_id in group:
                if policy_id not in policies:
                    raise KeyError(f"policy {policy_id!r} from policy group {group} is not a valid policy")
                grouped.add(policy_id)
        self.policy_groups += [[policy_id] for policy_id in policies if policy_id not in grouped]
        self.num_groups = len(self.policy_groups)
        self.finalized = True

    def _info_file(self):
        return os.path.join(self.state_path, "FreezeCyclingScheduler.yaml")

    def save_info(self):
        import yaml
        with open(self._info_file(), "w") as out_f:
            yaml.dump({"active_idx": self.active_idx}, out_f, default_flow_style=False)

    def load_info(self):
        import yaml
        if os.path.exists(self._info_file()):
            with open(self._info_file(), "r") as in_f:
                self.active_idx = yaml.safe_load(in_f)["active_idx"]

    def _freeze_group(self, group_idx):
        if self.verbose:
            rank_print(f"****Freezing policies: {self.policy_groups[group_idx]}****")
        for policy_id in self.policy_groups[group_idx]:
            self.policies[policy_id].freeze()
            if get_rank() == 0:
                self.policies[policy_id].save(self.state_path, self.status_dict["global status"]["iteration"])

    def _unfreeze_group(self, group_idx):
        if self.verbose:
            rank_print(f"****Un-freezing policies: {self.policy_groups[group_idx]}****")
        for policy_id in self.policy_groups[group_idx]:
            self.policies[policy_id].unfreeze()

    def __call__(self):
        iteration = self.status_dict["global status"]["iteration"]
        if iteration == self.delay + 1:
            if self.verbose:
                rank_print("****Beginning freeze cycling!****")
            for group_idx in range(self.num_groups):
                self._freeze_group(group_idx)
            self._unfreeze_group(self.active_idx)
        elif iteration > self.delay + 1 and iteration % self.iterations == 0:
            previous = self.active_idx
            self.active_idx = (self.active_idx + 1) % self.num_groups
            self._freeze_group(previous)
            self._unfreeze_group(self.active_idx)
