"""
Import recipe for the *unmodified* reference (build container only; /root/reference never travels).

A scratch directory under /tmp goes on sys.path holding
  (i)   a symlink  ppo_and_friends -> /root/reference        (mirrors setup.py:6-13)
  (ii)  a stand-in for `mpi4py`: rank 0 of 1 (collectives = identity) by default; with PPOAF_FAKE_MPI_SIZE=R in the
        environment a REAL R-process communicator over a unix socket (rank 0 is the hub): allgather / allreduce /
        Bcast / barrier move the ranks' objects and fold them in rank order, so the reference's own
        mpi_avg_gradients / RunningMeanStd.update(allgather) / per-epoch allreduces run as they do under mpirun
  (iii) metadata-only stand-ins for `gymnasium(.spaces)` and `gym(.spaces)`:
        Box / Discrete / MultiDiscrete / MultiBinary / Tuple / Dict carrying shape, dtype, n, nvec,
        low, high -- NO arithmetic any fixture depends on, so every recorded number comes from the
        reference's own code and torch.  `flatten_space` only concatenates bounds (shape metadata).

Nothing here is shipped: tests/ may call it to regenerate fixtures; the GPU box never does.
"""
import os
import shutil
import sys
import tempfile

REFERENCE = "/root/reference"

_MPI4PY = '''
import numpy as _np
class _Comm:
    def Get_rank(self): return 0
    def Get_size(self): return 1
    def allreduce(self, x, op=None): return x
    def allgather(self, x): return [x]
    def Allreduce(self, send, recv, op=None): _np.copyto(recv, send)
    def Bcast(self, buf, root=0): return None
    def bcast(self, x, root=0): return x
    def barrier(self): return None
    def Barrier(self): return None
    def Abort(self, code=1): raise RuntimeError('MPI Abort')
class MPI:
    COMM_WORLD = _Comm()
    SUM = 'sum'; MAX = 'max'; MIN = 'min'
'''

# R processes (make_golden_update.py starts them): PPOAF_FAKE_MPI_RANK / _SIZE / _ADDR (a unix socket path).
# Every collective is one gather-to-rank-0 + scatter of the rank-ordered list; reductions fold that list in rank order
# with the Python operator mpi4py's lowercase (pickle) collectives apply: SUM -> a + b, MAX -> max, MIN -> min.
# (For R = 2 a floating-point sum does not depend on the order at all: a + b == b + a bit for bit.)
_MPI4PY_MULTI = '''
import os as _os, time as _time
import numpy as _np
from multiprocessing.connection import Listener as _Listener, Client as _Client
_RANK = int(_os.environ["PPOAF_FAKE_MPI_RANK"]); _SIZE = int(_os.environ["PPOAF_FAKE_MPI_SIZE"])
_ADDR = _os.environ["PPOAF_FAKE_MPI_ADDR"]
class _Comm:
    def __init__(self):
        if _RANK == 0:
            lst = _Listener(_ADDR, family="AF_UNIX")
            self._peers = {}
            for _ in range(_SIZE - 1):
                c = lst.accept(); self._peers[c.recv()] = c
        else:
            for _ in range(600):
                try:
                    self._hub = _Client(_ADDR, family="AF_UNIX"); break
                except (FileNotFoundError, ConnectionRefusedError):
                    _time.sleep(0.1)
            self._hub.send(_RANK)
    def _everyone(self, x):
        """-> [rank 0's x, rank 1's x, ...] on every rank."""
        if _RANK == 0:
            parts = [x] + [self._peers[r].recv() for r in range(1, _SIZE)]
            for r in range(1, _SIZE):
                self._peers[r].send(parts)
            return parts
        self._hub.send(x)
        return self._hub.recv()
    def Get_rank(self): return _RANK
    def Get_size(self): return _SIZE
    def allgather(self, x): return self._everyone(x)
    def allreduce(self, x, op=None):
        parts = self._everyone(x)
        out = parts[0]
        for p in parts[1:]:
            out = (out + p) if op in (None, "sum") else (max(out, p) if op == "max" else min(out, p))
        return out
    def Allreduce(self, send, recv, op=None): _np.copyto(recv, self.allreduce(_np.asarray(send), op))
    def Bcast(self, buf, root=0):
        parts = self._everyone(_np.array(buf, copy=True) if _RANK == root else None)
        if _RANK != root:
            _np.copyto(buf, parts[root])
    def bcast(self, x, root=0): return self._everyone(x if _RANK == root else None)[root]
    def barrier(self): self._everyone(None)
    def Barrier(self): self._everyone(None)
    def Abort(self, code=1): raise RuntimeError('MPI Abort')
class MPI:
    COMM_WORLD = _Comm()
    SUM = 'sum'; MAX = 'max'; MIN = 'min'
'''

_GYMNASIUM_SPACES = '''
"""Metadata-only stand-in (shape / dtype / bounds).  See tests/golden/ref_import.py."""
import numpy as np
class Space:
    def __init__(self, shape=None, dtype=None):
        self._shape = None if shape is None else tuple(int(s) for s in shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
    @property
    def shape(self): return self._shape
    def seed(self, s=None): return [s]
    def __eq__(self, other):
        return (type(self) is type(other) and self._shape == other._shape and self.dtype == other.dtype
                and all(np.array_equal(getattr(self, k, None), getattr(other, k, None)) for k in ("low", "high", "n", "nvec")))
    __hash__ = object.__hash__
class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        super().__init__(shape, dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self._shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self._shape).copy()
class Discrete(Space):
    def __init__(self, n, start=0):
        super().__init__((), np.int64); self.n = n; self.start = start
class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64):
        self.nvec = np.asarray(nvec, dtype=dtype); super().__init__(self.nvec.shape, dtype)
class MultiBinary(Space):
    def __init__(self, n):
        self.n = n; super().__init__((n,) if np.isscalar(n) else tuple(n), np.int8)
class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces); super().__init__(None, None)
    def __iter__(self): return iter(self.spaces)
    def __len__(self): return len(self.spaces)
    def __getitem__(self, i): return self.spaces[i]
class Dict(Space):
    def __init__(self, spaces=None):
        self.spaces = dict(spaces or {}); super().__init__(None, None)
    def __getitem__(self, k): return self.spaces[k]
    def __setitem__(self, k, v): self.spaces[k] = v
    def __iter__(self): return iter(self.spaces)
    def keys(self): return self.spaces.keys()
def flatten_space(space):
    if isinstance(space, Box):
        return Box(space.low.flatten(), space.high.flatten(), dtype=space.dtype)
    if isinstance(space, Tuple):
        subs = [flatten_space(s) for s in space.spaces]
        return Box(np.concatenate([s.low for s in subs]), np.concatenate([s.high for s in subs]),
                   dtype=np.result_type(*[s.dtype for s in subs]))
    raise NotImplementedError(type(space))
class utils:
    flatten_space = staticmethod(flatten_space)
'''

_GYMNASIUM = '''
from . import spaces
class Env: pass
class Wrapper: pass
'''

_OLD_GYM_SPACES = '''
"""Old-gym stand-in: distinct classes nobody instantiates (the reference only type-checks against them)."""
class Box: pass
class Discrete: pass
class MultiDiscrete: pass
class MultiBinary: pass
class Tuple: pass
class Dict: pass
'''


def _write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as fh:
        fh.write(text)


def make_scratch():
    """Create the scratch import directory, put it first on sys.path, return its path."""
    sys.dont_write_bytecode = True
    scratch = tempfile.mkdtemp(prefix="ppoaf_golden_")
    os.symlink(REFERENCE, os.path.join(scratch, "ppo_and_friends"))
    multi = int(os.environ.get("PPOAF_FAKE_MPI_SIZE", "1")) > 1
    _write(os.path.join(scratch, "mpi4py", "__init__.py"), _MPI4PY_MULTI if multi else _MPI4PY)
    _write(os.path.join(scratch, "gymnasium", "__init__.py"), _GYMNASIUM)
    _write(os.path.join(scratch, "gymnasium", "spaces", "__init__.py"), _GYMNASIUM_SPACES)
    _write(os.path.join(scratch, "gym", "__init__.py"), "from . import spaces\n")
    _write(os.path.join(scratch, "gym", "spaces", "__init__.py"), _OLD_GYM_SPACES)
    sys.path.insert(0, scratch)
    return scratch


def drop_scratch(scratch):
    if scratch in sys.path:
        sys.path.remove(scratch)
    shutil.rmtree(scratch, ignore_errors=True)
