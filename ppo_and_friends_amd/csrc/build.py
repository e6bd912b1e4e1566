"""
Builds ppo_and_friends_amd/csrc/libppoaf_hip.so for gfx950 with hipcc, in-tree
(the .so is git-ignored but travels to the GPU box with the repo snapshot).

    python -m ppo_and_friends_amd.csrc.build [--force]

hipcc cross-compiles without a GPU.  Staleness is decided by CONTENT, not by mtime: every object
carries a stamp = sha256(source text + every header's text + compiler flags + hipcc version); a .hip
is recompiled when its stamp differs, and the library is relinked when the set of stamps differs
from the one recorded next to it (a checkout that keeps an old .so beside newer sources with equal
mtimes therefore rebuilds).
"""
import hashlib
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


def _read(path):
    with open(path, "rb") as fh:
        return fh.read()


def _toolchain_id():
    try:
        return subprocess.run([HIPCC, "--version"], capture_output=True, check=True).stdout
    except Exception:                                     # no compiler: stamps still compare the sources
        return b"no-hipcc"


def _stamp_of(path):
    return _read(path).decode().strip() if os.path.exists(path) else ""


def source_stamps():
    """{source file: content stamp} -- what the in-tree objects / library must have been built from."""
    common = hashlib.sha256()
    for h in sorted(headers()):
        common.update(os.path.basename(h).encode() + b"\0" + _read(h))
    common.update(" ".join(FLAGS).encode() + _toolchain_id())
    return {src: hashlib.sha256(common.digest() + _read(os.path.join(HERE, src))).hexdigest() for src in sources()}


def library_is_current():
    """True when libppoaf_hip.so was linked from exactly the sources in the tree (by content)."""
    stamps = source_stamps()
    return os.path.exists(LIB) and _stamp_of(LIB + ".stamp") == hashlib.sha256(
        "".join(f"{k}:{v};" for k, v in sorted(stamps.items())).encode()).hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamps = source_stamps()
    objs, procs = [], []
    for src in sources():
        s = os.path.join(HERE, src)
        o = os.path.join(OBJ_DIR, src[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or _stamp_of(o + ".stamp") != stamps[src]:
            cmd = [HIPCC, *FLAGS, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            if os.path.exists(o + ".stamp"):
                os.remove(o + ".stamp")
            procs.append((src, o, subprocess.Popen(cmd)))
    failed = []
    for src, o, p in procs:
        if p.wait() != 0:
            failed.append(src)
        else:
            with open(o + ".stamp", "w") as fh:
                fh.write(stamps[src])
    if failed:
        raise RuntimeError(f"hipcc failed for: {failed}")
    lib_stamp = hashlib.sha256("".join(f"{k}:{v};" for k, v in sorted(stamps.items())).encode()).hexdigest()
    if force or procs or not os.path.exists(LIB) or _stamp_of(LIB + ".stamp") != lib_stamp:
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(LIB + ".stamp", "w") as fh:
            fh.write(lib_stamp)
    elif verbose:
        print(f"{LIB} is current (content stamp {lib_stamp[:16]})", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
