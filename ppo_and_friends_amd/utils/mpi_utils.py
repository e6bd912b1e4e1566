"""
Rank utilities and DD-PPO collectives -- the stand-in for the reference's
utils/mpi_utils.py (mpi4py on COMM_WORLD) with one process per GPU and
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the
CPU tests of the host logic).

Reference call sites replaced:
  broadcast_model_parameters   mpi_utils.py:50-63   per-tensor Bcast  -> ONE broadcast of the flat bucket
  mpi_avg / mpi_avg_gradients  mpi_utils.py:65-111  per-tensor pickled allreduce
                                                    -> ONE all-reduce(SUM) of the flat gradient bucket;
                                                       the 1/num_procs is folded into the Adam kernel
  RunningMeanStd allgather     stats.py:47-50       raw data allgather -> all-gather of (n, mean, M2)
  comm.barrier()               ppo.py:2221,2468...  not needed: collectives on a stream order the ranks

No process group (single rank) is the common case: every function then
degenerates to the identity, as the reference's do for num_procs == 1.
"""
import os
import sys

import torch
import torch.distributed as dist


def is_initialized():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if is_initialized() else 0


def get_num_procs():
    return dist.get_world_size() if is_initialized() else 1


def rehearsing():
    """
    PPOAF_REHEARSE_MULTI_RANK=1: take every N > 1 code path (process group, per-mini-batch gradient
    all-reduce, record all-gathers, eager launches) even with ONE rank.  A one-GPU box cannot host two
    RCCL ranks, so this is how the RCCL call sequence of the multi-GPU path is exercised there.
    """
    return os.environ.get("PPOAF_REHEARSE_MULTI_RANK", "0") == "1"


def distributed_path():
    """True when collectives have to run: more than one rank, or a rehearsal of that path."""
    return is_initialized() and (dist.get_world_size() > 1 or rehearsing())


def init_process_group_from_env(backend=None):
    """
    torchrun-style bootstrap (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), the
    replacement for `mpirun -n N` (README.md:88-103).  Returns (rank, world, local_rank).
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or rehearsing()) and not is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL / K17 (no effect once HIP is up)
        if backend is None:
            backend = os.environ.get("PPOAF_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
            try:
                # the per-mini-batch gradient all-reduce is a 270 KB, latency-bound message on the critical
                # path: its RCCL kernels go to a high-priority stream
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                kw["pg_options"] = opts
            except Exception:
                pass
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def rank_print(msg, root=0, debug=False):
    """mpi_utils.py:11-35."""
    if root == get_rank():
        print("{}: {}".format(root, msg))
    sys.stdout.flush()


def set_torch_threads():
    """mpi_utils.py:37-48: cap intra-op threads at threads / num_procs."""
    if torch.get_num_threads() == 1:
        return
    torch.set_num_threads(max(int(torch.get_num_threads() / get_num_procs()), 1))


def _needs_staging(t):
    """gloo has no device collectives in this build: device tensors go through a host copy
    (used by the 2-ranks-on-one-GPU test of the N>1 control flow; RCCL runs take the direct path)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def broadcast_flat(flat, root=0):
    """One broadcast of a flat parameter bucket from `root`."""
    if distributed_path():
        if _needs_staging(flat):
            h = flat.cpu()
            dist.broadcast(h, src=root)
            flat.copy_(h)
        else:
            dist.broadcast(flat, src=root)
    return flat


def broadcast_model_parameters(model):
    """
    mpi_utils.py:50-63.  Models built by this package keep their parameters in
    one flat bucket (`model.flat_params`): that is broadcast in one message;
    foreign nn.Modules fall back to one broadcast per tensor.
    """
    if not distributed_path():
        return
    flat = getattr(model, "flat_params", None)
    if flat is not None:
        broadcast_flat(flat)
        return
    for p in model.parameters():
        dist.broadcast(p.data, src=0)


def allreduce_sum_(t):
    """In-place SUM all-reduce (identity on one rank)."""
    if distributed_path():
        if _needs_staging(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def mpi_avg(data):
    """mpi_utils.py:65-86: average a float / int / tensor across ranks."""
    n = get_num_procs()
    if not distributed_path():
        return data
    if torch.is_tensor(data):
        t = data.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t / n
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" \
        else torch.device("cpu")
    t = torch.tensor([float(data)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item() / n


def mpi_avg_gradients(model):
    """
    mpi_utils.py:89-111.  For this package's flat-bucket models the SUM is one
    all-reduce of `model.flat_grads` and the division by num_procs happens in
    the fused Adam kernel (grad_scale); for foreign modules gradients are
    averaged per tensor, in place.
    """
    n = get_num_procs()
    if not distributed_path():
        return
    flat = getattr(model, "flat_grads", None)
    if flat is not None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return
    for p in model.parameters():
        if p.grad is None:
            continue
        dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
        p.grad.div_(n)


def allgather_records(rec):
    """
    All-gather a small fixed-size record (e.g. the (n, mean, M2) moments of a
    batch): returns [R, len(rec)] (R = 1 without a process group).
    """
    n = get_num_procs()
    if not distributed_path():
        return rec.reshape(1, -1)
    if _needs_staging(rec):
        h = rec.reshape(1, -1).cpu().contiguous()
        out = torch.empty(n, rec.numel(), dtype=rec.dtype)
        dist.all_gather_into_tensor(out, h)
        return out.to(rec.device)
    out = torch.empty(n, rec.numel(), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec.reshape(1, -1).contiguous())
    return out


def bucket_checksum(tensors):
    """
    Order-sensitive int64 checksum of the BIT PATTERNS of float32 tensors (wrapping arithmetic, exact and
    deterministic): replicas that are bitwise identical -- what synchronous DD-PPO maintains -- agree on it.
    """
    acc = None
    for t in tensors:
        bits = t.detach().reshape(-1).view(torch.int32).to(torch.int64)
        w = torch.arange(1, bits.numel() + 1, dtype=torch.int64, device=bits.device) % 65521 + 1
        c = (bits * w).sum()
        acc = c if acc is None else acc * 1000003 + c
    return acc.reshape(1)


def replicas_agree(tensors):
    """True when every rank holds the same checksum of `tensors` (one tiny all-gather)."""
    if not distributed_path() or get_num_procs() == 1:
        return True
    allc = allgather_records(bucket_checksum(tensors))
    return bool((allc == allc[0]).all().item())


def allreduce_scalars(values, op="sum"):
    """One packed all-reduce for a list of Python scalars (ppo.py:2471-2475, 1991-2094)."""
    if not distributed_path():
        return list(values)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" \
        else torch.device("cpu")
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    rop = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op]
    dist.all_reduce(t, op=rop)
    return t.tolist()
