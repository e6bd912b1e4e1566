"""
TEST INFRASTRUCTURE -- ctypes access to oracle/_build/libppoaf_oracle.so (oracle/gae_oracle.c,
built by `make -C oracle` / __graft_entry__.build()).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libppoaf_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = C.CDLL(_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def gae_rtg_episode(rewards, values, ending_value, ending_reward, gamma, lambd, clip, use_gae=True):
    r = np.ascontiguousarray(rewards, dtype=np.float64)
    v = np.ascontiguousarray(values, dtype=np.float32)
    L = len(r)
    adv = np.zeros(L); rtg = np.zeros(L)
    hc, lo, hi = (0, 0.0, 0.0) if clip is None else (1, float(clip[0]), float(clip[1]))
    load().ppoaf_oracle_gae_rtg_episode(_p(r), _p(v), C.c_int64(L), C.c_double(ending_value),
                                        C.c_double(ending_reward), C.c_double(gamma), C.c_double(lambd),
                                        C.c_int(hc), C.c_double(lo), C.c_double(hi), C.c_int(int(use_gae)),
                                        _p(adv), _p(rtg))
    return adv, rtg


def gae_rtg_tmajor(rewards, values, boot_value, boot_reward, end_kind, gamma=0.99, lambd=0.95,
                   clip=(-100.0, 100.0), use_gae=True):
    r = np.ascontiguousarray(rewards, dtype=np.float32)
    v = np.ascontiguousarray(values, dtype=np.float32)
    bv = np.ascontiguousarray(boot_value, dtype=np.float32)
    br = np.ascontiguousarray(boot_reward, dtype=np.float32)
    ek = np.ascontiguousarray(end_kind, dtype=np.int8)
    T, E = r.shape
    adv = np.zeros((T, E), dtype=np.float32); rtg = np.zeros((T, E), dtype=np.float32)
    hc, lo, hi = (0, 0.0, 0.0) if clip is None else (1, float(clip[0]), float(clip[1]))
    load().ppoaf_oracle_gae_rtg_tmajor(_p(r), _p(v), _p(bv), _p(br), _p(ek), C.c_int32(T), C.c_int64(E),
                                       C.c_double(gamma), C.c_double(lambd), C.c_int(hc), C.c_double(lo),
                                       C.c_double(hi), C.c_int(int(use_gae)), _p(adv), _p(rtg))
    return adv, rtg
