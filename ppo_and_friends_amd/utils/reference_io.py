"""
Checkpoint interchange with the reference.

The reference pickles live `RunningMeanStd` objects (utils/stats.py:9-94):
    <policy>-value_normalizer_stats_<rank>.pickle     one instance            utils/misc.py:141-145
    ActorRunningObsStats_<rank>.pickle ...            {agent_id: instance}    environments/filter_wrappers.py:296-311,489
A pickle names the class by module path, so files written here carry
`ppo_and_friends.utils.stats.RunningMeanStd` and hold the attributes that class has (mean, variance
float32 arrays, count float) -- `ppoaf test` / a resumed reference run unpickles them as its own objects.
Reading accepts both those and the plain-dict payloads earlier versions of this package wrote.

Optimiser files (`actor_optim_<rank>` ..., policies/ppo_policy.py:1239-1300) are `torch.save`d
`torch.optim.Adam.state_dict()`s: `adam_state_dict` / `load_adam_state_dict` convert between that layout
(per-parameter exp_avg / exp_avg_sq / step) and the flat device buckets of FlatAdam.
"""
import contextlib
import io
import pickle
import sys
import types

import numpy as np
import torch

_REF_MODULE = "ppo_and_friends.utils.stats"


class _RunningMeanStdState:
    """Attribute bag with the reference class's fields; pickled under the reference's class path."""

    def __init__(self, mean, variance, count):
        self.mean = np.asarray(mean, dtype=np.float32)
        self.variance = np.asarray(variance, dtype=np.float32)
        self.count = float(count)


_RunningMeanStdState.__module__ = _REF_MODULE
_RunningMeanStdState.__qualname__ = _RunningMeanStdState.__name__ = "RunningMeanStd"


@contextlib.contextmanager
def _reference_namespace():
    """
    Make `ppo_and_friends.utils.stats.RunningMeanStd` resolvable while pickling.  If the reference package
    is installed its own class is used; otherwise placeholder modules are registered for the duration.
    """
    try:
        import importlib
        mod = importlib.import_module(_REF_MODULE)
        yield getattr(mod, "RunningMeanStd")
        return
    except Exception:
        pass
    names = ["ppo_and_friends", "ppo_and_friends.utils", _REF_MODULE]
    saved = {n: sys.modules.get(n) for n in names}
    try:
        for n in names:
            sys.modules[n] = types.ModuleType(n)
        sys.modules[_REF_MODULE].RunningMeanStd = _RunningMeanStdState
        yield _RunningMeanStdState
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m


def _as_reference_object(cls, state):
    obj = cls.__new__(cls)
    f = lambda x: np.asarray(x) if np.asarray(x).dtype == np.float64 else np.asarray(x, dtype=np.float32)
    obj.mean = f(state["mean"])                       # float32 arrays; float64 once the reference's own arithmetic
    obj.variance = f(state["variance"])               # promoted them (reward statistics)
    obj.count = float(state["count"])
    return obj


def dump_running_stats(state, fh):
    """state = {"mean", "variance", "count"} or {agent_id: such a dict}; written as reference objects."""
    single = "mean" in state and "variance" in state and "count" in state
    with _reference_namespace() as cls:
        payload = _as_reference_object(cls, state) if single else \
            {k: _as_reference_object(cls, v) for k, v in state.items()}
        pickle.dump(payload, fh)


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == _REF_MODULE and name == "RunningMeanStd":
            return _RunningMeanStdState
        return super().find_class(module, name)


def _to_state(o):
    if isinstance(o, dict) and "mean" in o:
        return {"mean": np.asarray(o["mean"]), "variance": np.asarray(o["variance"]), "count": float(o["count"])}
    return {"mean": np.asarray(o.mean), "variance": np.asarray(o.variance), "count": float(o.count)}


def load_running_stats(fh):
    """-> {"mean", "variance", "count"} or {agent_id: such a dict}, from either payload kind."""
    o = _Unpickler(io.BytesIO(fh.read())).load()
    if isinstance(o, dict) and "mean" not in o:
        return {k: _to_state(v) for k, v in o.items()}
    return _to_state(o)


# ---------------------------------------------------------------------------------- optimisers
def adam_state_dict(network, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-5):
    """torch.optim.Adam.state_dict() layout for the parameters of `network` (in .parameters() order)."""
    state, off = {}, 0
    params = list(network.parameters())
    for i, p in enumerate(params):
        n = p.numel()
        state[i] = {"step": torch.tensor(float(step)),
                    "exp_avg": exp_avg[off:off + n].detach().cpu().reshape(p.shape).clone(),
                    "exp_avg_sq": exp_avg_sq[off:off + n].detach().cpu().reshape(p.shape).clone()}
        off += (n + 3) // 4 * 4
    group = {"lr": float(lr), "betas": tuple(betas), "eps": float(eps), "weight_decay": 0, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(params)))}
    return {"state": state if step > 0 else {}, "param_groups": [group]}


def load_adam_state_dict(sd, network, exp_avg, exp_avg_sq):
    """-> (step, lr); fills the flat moment buckets from a torch.optim.Adam state_dict."""
    off, step = 0, 0
    exp_avg.zero_(); exp_avg_sq.zero_()
    for i, p in enumerate(network.parameters()):
        n = p.numel()
        st = sd["state"].get(i)
        if st is not None:
            exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            step = int(float(st["step"]))
        off += (n + 3) // 4 * 4
    return step, float(sd["param_groups"][0]["lr"])
