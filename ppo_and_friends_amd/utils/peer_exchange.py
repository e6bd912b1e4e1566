"""
Host side of K17 (csrc/peer_exchange.hip): the per-mini-batch gradient exchange of DD-PPO between the
ranks of ONE node, over peer mappings of each rank's exchange slots (xGMI on an MI355X node).

Replaces the Allreduce inside the update loop (reference utils/mpi_utils.py:65-86, called from
ppo.py:2443-2448 per mini-batch).  torch.distributed stays the control plane: it carries the IPC blobs
once at start-up, the agreement votes below, and every per-epoch / per-iteration collective; the
per-mini-batch chain itself contains only this library's kernels, so it is captured and replayed as a
hipGraph exactly like the single-rank chain.

Selection (`PPOAF_GRAD_EXCHANGE`): "auto" (default) tries the peer exchange and keeps it only if every
rank (a) lives on the same host, (b) created / exported / connected without error and (c) passed the
start-up self-test -- several exchanges of changing random buckets compared bit for bit with the
rank-ordered sum of the all-gathered inputs (the gathering goes through torch.distributed, i.e. RCCL).
Every decision is a MIN vote over all ranks, so the ranks always agree on the path.  "rccl" skips the
attempt; "peer" raises instead of falling back.  The fallback is the eager loop with dist.all_reduce.
"""
import ctypes as C
import os
import socket

import torch
import torch.distributed as dist

from .. import _lib
from . import mpi_utils

BLOB_BYTES = 128
MAX_RANKS = 16


def _vote(ok):
    """True only if every rank says True (control-plane collective)."""
    if not mpi_utils.is_initialized() or dist.get_world_size() == 1:
        return bool(ok)
    return mpi_utils.allreduce_scalars([1.0 if ok else 0.0], op="min")[0] > 0.5


def _all_gather_bytes(payload):
    """bytes of equal length from every rank, in rank order."""
    if not mpi_utils.is_initialized() or dist.get_world_size() == 1:
        return [payload]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, payload)
    return out


class PeerExchange:
    """One exchange object per flat gradient bucket."""

    wait_seconds = 20.0          # in-kernel budget for a peer to show up (ranks are aligned by the epoch's collectives)

    def __init__(self, n_floats, device, memory_kind=0):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.rank, self.world = mpi_utils.get_rank(), mpi_utils.get_num_procs()
        self.n_floats = int(n_floats)
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ppoaf_peer_exchange_create(self.rank, self.world, self.n_floats, int(memory_kind),
                                                           C.byref(self.handle)), "peer_exchange_create")

    def export(self):
        blob = C.create_string_buffer(BLOB_BYTES)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ppoaf_peer_exchange_export(self.handle, blob), "peer_exchange_export")
        return blob.raw

    def connect(self, blobs):
        assert len(blobs) == self.world and all(len(b) == BLOB_BYTES for b in blobs)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ppoaf_peer_exchange_connect(self.handle, b"".join(blobs)), "peer_exchange_connect")

    def allreduce(self, src, dst, split_floats=None, norm_scale=1.0, norm_out=None, stream=None, wait_seconds=None):
        """One launch: dst = rank-ordered sum of every rank's src; squared norms of the two segments."""
        assert src.is_cuda and dst.is_cuda and src.dtype == dst.dtype == torch.float32
        assert src.numel() == dst.numel() == self.n_floats and src.is_contiguous() and dst.is_contiguous()
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.ppoaf_peer_exchange_allreduce(
            self.handle, src.data_ptr(), dst.data_ptr(),
            self.n_floats if split_floats is None else int(split_floats), float(norm_scale),
            None if norm_out is None else norm_out.data_ptr(),
            float(self.wait_seconds if wait_seconds is None else wait_seconds), stream)
        if rc != 0:
            _lib.check(rc, "peer_exchange_allreduce")

    def status(self):
        """(exchanges completed, sequence number of a timed-out wait or 0, memory kind, ranks); synchronises."""
        out = (C.c_int64 * 4)()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ppoaf_peer_exchange_status(self.handle, out), "peer_exchange_status")
        return tuple(int(v) for v in out)

    def check(self):
        done, timed_out, _, _ = self.status()
        if timed_out:
            raise _lib.PpoafError(f"peer exchange: rank {self.rank} waited more than {self.wait_seconds:.0f} s for its "
                                  f"peers at exchange {timed_out} ({done} completed); the gradients of that step are invalid")

    def close(self):
        if self.handle:
            self.lib.ppoaf_peer_exchange_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ self-test
    def self_test(self, rounds=4):
        """Exchanges of changing random data against the rank-ordered sum of the all-gathered inputs."""
        dev, n = self.device, self.n_floats
        split = (n // 8) * 4
        ok = True
        norms = torch.zeros(2, dtype=torch.float64, device=dev)
        for r in range(rounds):
            g = torch.Generator(device="cpu").manual_seed(7919 * (r + 1) + self.rank)
            src = torch.randn(n, generator=g).to(dev)
            parts = [torch.empty_like(src) for _ in range(self.world)]
            if self.world > 1:
                if mpi_utils._needs_staging(src):
                    host = [torch.empty(n) for _ in range(self.world)]
                    dist.all_gather(host, src.cpu())
                    parts = [h.to(dev) for h in host]
                else:
                    dist.all_gather(parts, src)
            else:
                parts = [src]
            want = torch.zeros_like(src)
            for p in parts:
                want = want + p
            got = torch.empty_like(src)
            self.allreduce(src, got, split_floats=split, norm_scale=0.5, norm_out=norms, wait_seconds=5.0)
            torch.cuda.synchronize(dev)
            wn = torch.stack([((want[:split] * 0.5).double() ** 2).sum(), ((want[split:] * 0.5).double() ** 2).sum()])
            ok = ok and torch.equal(got, want) and bool(torch.allclose(norms, wn, rtol=1e-12, atol=0.0))
            # in place, as the update loop uses it
            buf = src.clone()
            self.allreduce(buf, buf, wait_seconds=5.0)
            torch.cuda.synchronize(dev)
            ok = ok and torch.equal(buf, want)
        done, timed_out, _, _ = self.status()
        return ok and timed_out == 0 and done == 2 * rounds


def requested():
    return os.environ.get("PPOAF_GRAD_EXCHANGE", "auto").strip().lower()


def open_exchange(n_floats, device):
    """
    -> (PeerExchange or None, reason).  Collective: every rank must call it, and every rank gets the same
    answer.  None means "use the RCCL all-reduce path".
    """
    mode = requested()
    if mode not in ("auto", "peer", "rccl"):
        raise ValueError(f"PPOAF_GRAD_EXCHANGE={mode!r}: expected auto, peer or rccl")

    def refuse(reason):
        if mode == "peer":
            raise _lib.PpoafError("PPOAF_GRAD_EXCHANGE=peer but the peer exchange is unavailable: " + reason)
        return None, reason

    if mode == "rccl":
        return None, "PPOAF_GRAD_EXCHANGE=rccl"
    if not mpi_utils.distributed_path():
        return None, "single rank"
    device = torch.device(device)
    world = mpi_utils.get_num_procs()
    hosts = _all_gather_bytes(socket.gethostname().encode())
    if not _vote(device.type == "cuda" and world <= MAX_RANKS and len(set(hosts)) == 1 and n_floats % 4 == 0):
        return refuse(f"needs CUDA devices of one host, at most {MAX_RANKS} ranks and a bucket of a multiple of 4 floats")
    # exchange memory kinds in order of preference (PPOAF_PEER_MEMORY pins one): the first kind with which every
    # rank creates, exports, connects and passes the self-test is used
    names = {1: "uncached", 2: "fine-grained", 3: "coarse-grained"}
    forced = os.environ.get("PPOAF_PEER_MEMORY", "").strip().lower()
    kinds = [k for k, v in names.items() if v == forced] or [1, 2, 3]
    reasons = []
    for kind in kinds:
        x, err = None, ""
        try:
            x = PeerExchange(n_floats, device, memory_kind=kind)
            blob = x.export()
        except Exception as exc:                               # noqa: BLE001 -- any failure means "next option", by vote
            blob, err = b"\0" * BLOB_BYTES, f"{type(exc).__name__}: {exc}"
        blobs = _all_gather_bytes(blob)
        if not _vote(x is not None and not err):
            if x is not None:
                x.close()
            reasons.append(f"{names[kind]}: create/export failed on a rank" + (f" ({err})" if err else ""))
            continue
        try:
            x.connect(blobs)
        except Exception as exc:                               # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"
        if not _vote(not err):
            x.close()
            reasons.append(f"{names[kind]}: connect failed on a rank" + (f" ({err})" if err else ""))
            continue
        if mpi_utils.is_initialized() and world > 1:
            dist.barrier()                                     # every rank's flag words are zeroed and mapped
        try:
            good = x.self_test()
        except Exception as exc:                               # noqa: BLE001
            good, err = False, f"{type(exc).__name__}: {exc}"
        if _vote(good):
            return x, f"peer mappings ({names[kind]} slots), self-test passed"
        x.close()
        if mpi_utils.is_initialized() and world > 1:
            dist.barrier()                                     # nobody still reads a slot that is about to be freed
        reasons.append(f"{names[kind]}: self-test against the gathered sum failed on a rank" + (f" ({err})" if err else ""))
    return refuse("; ".join(reasons))
