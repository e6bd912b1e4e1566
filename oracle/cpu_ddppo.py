"""
TEST INFRASTRUCTURE (see oracle/__init__.py) -- the reference's DD-PPO launch model on CPU cores, for bench.py's
`cpu_baseline` / `cpu_baseline_mpi` legs: R processes ("mpirun -n R ppoaf train ...", README.md:88-103), each with
its own envs_per_proc environments and shuffle stream (seed + rank, ppoaf_cli.py:419), meeting per mini-batch in

  * an all-gather of the raw rewards-to-go inside the value normaliser          utils/stats.py:47-50
  * one averaged all-reduce PER PARAMETER TENSOR after each backward            utils/mpi_utils.py:89-111
  * a barrier                                                                   ppo.py:2468
and per epoch in the scalar all-reduces of ppo.py:2471-2476.  torch.distributed/gloo stands in for mpi4py (absent
from this image; gloo's shared-memory/TCP loopback collectives are, if anything, cheaper than pickled MPI objects).
Every rank is a fresh python process that never touches a GPU; intra-op threads = cores / R as
utils/mpi_utils.py:37-48 sets them.

    python -m oracle.cpu_ddppo --world 8 --rank 3 --port 29533 --envs 64 --ts 128 --epochs 10 --batch 256 --threads 2
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class GlooComm:
    """The collective surface of oracle/cpu_ppo_loop.ThreadComm for R processes over torch.distributed/gloo (mpi4py is
    absent from this image): all-gather of Python / numpy objects, sum all-reduce of an ndarray or a scalar, barrier.
    For R = 2 gloo's sum is bitwise the rank-ordered fold (a + b == b + a)."""

    def __init__(self):
        import torch.distributed as dist
        self.dist, self.rank, self.size = dist, dist.get_rank(), dist.get_world_size()

    def allgather(self, x):
        parts = [None] * self.size
        self.dist.all_gather_object(parts, x)
        return parts

    def allreduce_sum(self, x):
        import numpy as np
        import torch
        if isinstance(x, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(x).copy())
            self.dist.all_reduce(t)
            return t.numpy()
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t)
        return type(x)(t.item())

    def barrier(self):
        self.dist.barrier()


def _worker(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from oracle import cpu_ppo_loop

    torch.set_num_threads(max(int(args.threads), 1))
    R, rank = args.world, args.rank
    if R > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{args.port}", rank=rank, world_size=R)
    comm = GlooComm() if R > 1 else None
    E, T, O, NA = args.envs, args.ts, 4, 2
    rng = np.random.default_rng(1234 + rank)
    obs_table = rng.standard_normal((T + 1, E, O), dtype=np.float32)
    rew_table = np.ones((T, E), dtype=np.float32)
    ppo = cpu_ppo_loop.CpuPPO(O, NA, batch_size=args.batch, seed=1)        # same initial weights on every rank
    ppo.loader_generator = torch.Generator().manual_seed(1 + rank)

    def barrier():
        if R > 1:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    ppo.rollout(obs_table, rew_table)
    barrier()
    t1 = time.perf_counter()
    # the epoch loop itself is the pinned restatement (R = 2 reference fixtures, tests/test_oracle_update_golden.py):
    # per mini-batch the all-gather inside the value normaliser, one averaged all-reduce PER PARAMETER TENSOR after each
    # backward, a barrier; per epoch the scalar all-reduces of ppo.py:2471-2476
    cpu_ppo_loop.train_on_rollout(ppo, args.epochs, comm=comm)
    barrier()
    t2 = time.perf_counter()
    if rank == 0:
        print("CPU_DDPPO " + json.dumps(dict(env_steps=R * E * T, rollout_s=t1 - t0, update_s=t2 - t1, wall_s=t2 - t0,
                                             ranks=R, threads=torch.get_num_threads())), flush=True)
    if R > 1:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(R, envs_per_rank, T=128, epochs=10, batch=256, threads=1, timeout=600):
    """Launch R fresh CPU-only processes, wait, return rank 0's timing dict (+ env_steps_per_s)."""
    port = _free_port()
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="",
               OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_ddppo", "--world", str(R), "--rank", str(r), "--port", str(port),
                               "--envs", str(envs_per_rank), "--ts", str(T), "--epochs", str(epochs), "--batch", str(batch),
                               "--threads", str(threads)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(R)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        raise
    if any(p.returncode != 0 for p in procs):
        raise RuntimeError("cpu_ddppo rank failed:\n" + "\n".join(o[1][-2000:] for o in outs))
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("CPU_DDPPO ")][-1]
    res = json.loads(line[len("CPU_DDPPO "):])
    res["env_steps_per_s"] = res["env_steps"] / res["wall_s"]
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--port", type=int, default=29500)
    ap.add_argument("--envs", type=int, default=8)
    ap.add_argument("--ts", type=int, default=128)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--threads", type=int, default=1)
    _worker(ap.parse_args())
