"""
Builds ppo_and_friends_amd/csrc/libppoaf_hip.so for gfx950 with hipcc, in-tree
(the .so is git-ignored but travels to the GPU box with the repo snapshot).

    python -m ppo_and_friends_amd.csrc.build [--force]

hipcc cross-compiles without a GPU.  Each .hip is compiled to an object only
when it (or a header) is newer than the object, then everything is linked.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libppoaf_hip.so")
OBJ_DIR = os.path.join(HERE, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hpp")]
    hs.append(os.path.join(ROOT, "include", "ppoaf_hip.h"))
    return hs


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_mtime = max(os.path.getmtime(h) for h in headers())
    objs, procs = [], []
    for src in sources():
        s = os.path.join(HERE, src)
        o = os.path.join(OBJ_DIR, src[:-4] + ".o")
        objs.append(o)
        stale = force or _newer(s, o) or hdr_mtime > os.path.getmtime(o)
        if stale:
            cmd = [HIPCC, *FLAGS, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for: {failed}")
    if force or procs or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
