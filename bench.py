#!/usr/bin/env python3
"""
bench.py -- env-steps/s of the PPO hot path (rollout + GAE + update) on N MI355X.

A "step" is ONE PPO iteration of BASELINE.json config C2 on every rank:
  rollout of E=4096 envs x T=128 steps (batched actor/critic inference + sampling),
  dataset build (all GAE / rewards-to-go scans in one launch),
  10 epochs x (E*T/256) shuffled mini-batches of clipped-surrogate + value +
  entropy loss, backward, [RCCL all-reduce], clip + Adam.
Inputs are synthetic fixed-length trajectories: obs ~ N(0,1)
float32 resident in HBM before the timed region, reward 1.0 (CartPole),
default_rng(1234 + rank).  value = N * E * T * K / max-over-ranks wall time.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line; `roofline` is the GAE kernel (the kernel the metric
names) timed with HIP events on the launch stream inside the timed region;
`cpu_baseline` is oracle/cpu_ppo_loop.py (a port with the reference's loop
structure) on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
GAE_BYTES_PER_TRANSITION = 16  # read r, V; write adv, rtg


def pmc_traffic():
    """
    HBM bytes per launch of the GAE kernels from the committed PMC passes (profiles/*_gae_pmc.csv,
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs; KB units).  Corrections per
    MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide (16 B/lane) coalesced read on gfx950
    -> doubled for the streaming kernel; the chunked kernel reads 4 B/lane in 64-B segments
    -> taken as is; WRITE_SIZE is exact.  Counters cannot be collected inside a timed run, so
    these are the profiled values of the same kernels at the same sizes, not of this very run.
    """
    import glob, re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_gae_pmc.csv")))
    if not files:
        return {}
    vals = {}
    for line in open(files[-1]):
        m = re.match(r"^(.*),(FETCH_SIZE|WRITE_SIZE),dispatches=\d+,avg=([0-9.]+)", line)
        if m:
            vals[(m.group(1), m.group(2))] = float(m.group(3))
    out = {}
    for name, mult in (("ppoaf::gae_rtg_chunked_kernel", 1.0), ("void ppoaf::gae_rtg_stream_kernel<4, 8>", 2.0)):
        f, w = vals.get((name, "FETCH_SIZE")), vals.get((name, "WRITE_SIZE"))
        if f is not None and w is not None:
            out[name] = int((mult * f + w) * 1024)
    out["_source"] = os.path.relpath(files[-1], ROOT)
    return out


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--envs", type=int, default=4096)
    p.add_argument("--ts", type=int, default=128)
    p.add_argument("--batch-size", type=int, default=256)
    p.add_argument("--epochs", type=int, default=10)
    p.add_argument("--no-graphs", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-envs", type=int, default=256)
    p.add_argument("--no-saturating", action="store_true")
    p.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5"],
                   help="C2 (default, the metric's config) | C3 dims: HalfCheetah O=17, Box(6), actor 128^3 / "
                        "critic 256^3, E=2048, ICM + obs/reward normalisers and clippers | C4 dims: SimpleSpread MAPPO, 3 agents, "
                        "O=18, O_c=54, Discrete(5), E=1024 per rank | C5 dims: same env, MATPolicy (embedding 64, "
                        "1 block, 1 head, critic view local), E=1024 per rank")
    return p.parse_args()


def main():
    args = parse()
    # dmabuf IPC (the K17 peer mappings, RCCL's own P2P): must be in the environment before HIP initialises
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from ppo_and_friends_amd.utils import mpi_utils
    from ppo_and_friends_amd.ppo import PPO
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.spaces import Box, Discrete
    from ppo_and_friends_amd import kernels as K

    rank, world, local_rank = mpi_utils.init_process_group_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("PPOAF_SHARE_DEVICE", "0") == "1":
        local_rank = 0          # rehearsal on a one-GPU box (with PPOAF_BACKEND=gloo): all ranks on device 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    E, T, O, NA = args.envs, args.ts, 4, 2
    A, critic_view, act_space, pargs, workload = 1, "local", Discrete(NA), {}, None
    filters = dict(normalize_obs=False, normalize_rewards=False)     # C2/C4/C5 as SURVEY.md §8(d) defines them
    if args.config == "C3":
        E, O = (2048 if args.envs == 4096 else args.envs), 17
        act_space = Box(-1.0, 1.0, (6,), np.float32)
        pargs = dict(actor_kw_args=dict(hidden_size=128), critic_kw_args=dict(hidden_size=256), enable_icm=True)
        filters = dict(normalize_obs=True, normalize_rewards=True, obs_clip=(-10.0, 10.0), reward_clip=(-10.0, 10.0))
        workload = (f"C3 dims (HalfCheetah-v4: O=17, Box(6) tanh-Gaussian, actor 128^3, critic 256^3, ICM, "
                    f"obs/reward normalisers + clippers), envs_per_proc={E}, ts_per_rollout={T}")
    elif args.config == "C4":
        E, O, NA, A, critic_view = (1024 if args.envs == 4096 else args.envs), 18, 5, 3, "policy"
        act_space = Discrete(NA)
        pargs = dict(actor_kw_args=dict(hidden_size=128), critic_kw_args=dict(hidden_size=256))
        workload = (f"C4 dims (MPE simple_spread MAPPO: 3 agents share one policy, O=18, O_c=54, Discrete(5), "
                    f"actor 128^3, critic 256^3), envs_per_proc={E}, ts_per_rollout={T}")

    policy_class = None
    if args.config == "C5":
        from ppo_and_friends_amd.policies.mat_policy import MATPolicy
        E, O, NA, A, critic_view = (1024 if args.envs == 4096 else args.envs), 18, 5, 3, "local"
        act_space, policy_class, pargs = Discrete(NA), MATPolicy, {}
        workload = (f"C5 dims (MPE simple_spread MATPolicy: 3 agents, O=18, Discrete(5), embedding 64, 1 block, "
                    f"1 head, attention core on f32 MFMA), envs_per_proc={E}, ts_per_rollout={T}")

    env_gen = lambda: SyntheticFixedLengthEnv(E, O, act_space, T, device, reward="ones" if args.config == "C2" else "uniform",
                                              seed=1234, rank=rank, num_agents=A, critic_view=critic_view)
    obs_space = Box(-np.inf, np.inf, (O,), np.float32)
    cobs_space = Box(-np.inf, np.inf, (O * A if critic_view == "policy" else O,), np.float32)
    settings = {"cartpole": (policy_class, obs_space, cobs_space, act_space, pargs)}
    ppo = PPO(env_gen, settings, device=device, random_seed=1, envs_per_proc=E, ts_per_rollout=T,
              batch_size=args.batch_size, epochs_per_iter=args.epochs, use_graphs=not args.no_graphs, **filters)
    pol = ppo.policies["cartpole"]

    def barrier():
        if mpi_utils.distributed_path():
            dist.barrier()
        torch.cuda.synchronize()

    gae_events = []

    def iteration(timed):
        ppo.rollout_buffer_hook = gae_events if timed else None
        ppo.rollout()
        ppo.train_on_rollout()

    # GAE launch timing: HIP events on the launch stream, around the K1 launch of every timed rollout
    import ppo_and_friends_amd.utils.episode_info as ei
    orig = ei.RolloutBuffer.compute_advantages

    def timed_compute(self, *a, **kw):
        hook = getattr(ppo, "rollout_buffer_hook", None)
        if hook is None:
            return orig(self, *a, **kw)
        ev = (K.event_create(), K.event_create())      # stamped with the kernel's own begin / end
        r = orig(self, *a, timing_events=ev, **kw)
        hook.append(ev)
        return r

    ei.RolloutBuffer.compute_advantages = timed_compute

    for _ in range(args.warmup):
        iteration(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        iteration(True)
    barrier()
    dt = time.perf_counter() - t0
    dt = float(mpi_utils.allreduce_scalars([dt], op="max")[0])        # MAX over ranks

    env_steps = world * E * T * args.steps
    value = env_steps / dt
    gae_ms = [K.event_elapsed_ms(a, b) for a, b in gae_events]
    gae_avg_s = (sum(gae_ms) / max(len(gae_ms), 1)) * 1e-3
    gae_bytes = GAE_BYTES_PER_TRANSITION * E * T * A
    achieved = gae_bytes / gae_avg_s / 1e9 if gae_avg_s > 0 else 0.0
    pmc = pmc_traffic() if (E, T) == (4096, 128) else {}
    roofline = {"kernel": "gae_rtg_chunked_kernel" if E < (1 << 17) else "gae_rtg_stream_kernel", "bound": "hbm", "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": pmc.get("ppoaf::gae_rtg_chunked_kernel"), "traffic_source": pmc.get("_source"),
                "bytes_per_launch": gae_bytes,
                "avg_launch_us": round(gae_avg_s * 1e6, 3), "launches": len(gae_ms),
                "timing": "kernel begin/end events (hipExtLaunchKernelGGL) on the launch stream, timed region",
                "note": "config size (8.4 MB, fits L2/MALL) is latency-bound; see roofline_saturating"}

    # which per-mini-batch gradient exchange the update loops actually used (N > 1 or its rehearsal)
    fused = [f for f in getattr(ppo, "_fused", {}).values() if f is not None]
    peer = bool(fused) and all(getattr(f, "xchg", None) is not None for f in fused)
    exchange = None if not mpi_utils.distributed_path() else \
        ("K17 peer mappings (xGMI), in-graph" if peer else "RCCL all-reduce, eager loop")
    if exchange is not None and fused:                       # why that path (self-test verdict / fallback reason)
        exchange += f" [{getattr(fused[0], 'xchg_reason', '')}]"

    # N > 1 only, after the timed region: what one gradient exchange of this bucket costs on this node, with the
    # K17 kernel and with the process group's all-reduce (back-to-back launches, stream events) -- recorded so that
    # a multi-GPU run documents the latency its scaling number rests on
    probe = None
    if mpi_utils.distributed_path() and fused:
        n_f = pol.policy_grads.numel()
        buf = torch.zeros(n_f, dtype=torch.float32, device=device)
        reps = 200

        def timed(fn):
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps
        probe = {"bucket_floats": n_f, "launches": reps}
        x = getattr(fused[0], "xchg", None)
        if x is not None:
            for _ in range(5):
                x.allreduce(buf, buf)
            probe["k17_exchange_us"] = round(timed(lambda: x.allreduce(buf, buf)), 2)
            x.check()
        if not mpi_utils._needs_staging(buf):                 # device collectives (RCCL); gloo rehearsals skip it
            for _ in range(5):
                dist.all_reduce(buf)
            probe["process_group_allreduce_us"] = round(timed(lambda: dist.all_reduce(buf)), 2)

    out = {"metric": "env_steps_per_sec", "value": round(value, 1), "unit": "env-steps/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": (workload + f", batch_size={args.batch_size}, epochs_per_iter={args.epochs}"
                                   if workload else
                                   "C2 CartPole-v1 MLP (4->128x3->2 actor, ->1 critic), "
                                   f"envs_per_proc={E}, ts_per_rollout={T}, batch_size={args.batch_size}, "
                                   f"epochs_per_iter={args.epochs}, fixed-length synthetic trajectories"),
                      "agent_steps_per_iteration": world * E * T * A,
                      "global_env_steps_per_iteration": world * E * T,
                      "parallelism": f"dp{world}", "hip_graphs": (not args.no_graphs) and (not mpi_utils.distributed_path() or peer),
                      "multi_rank_path": mpi_utils.distributed_path(),
                      "gradient_exchange": exchange, "exchange_probe": probe,
                      "rollout_s": round(ppo.status_dict["global status"]["rollout time"], 4),
                      "train_s": round(ppo.status_dict["global status"]["train time"], 4)},
           "roofline": roofline}

    if rank == 0 and not args.no_saturating:
        # companion: same kernel at a bandwidth-saturating size (N = 2^28 transitions)
        Es = (1 << 28) // T
        r = torch.rand(T, Es, device=device); v = torch.randn(T, Es, device=device)
        b = torch.randn(Es, device=device)
        adv = torch.empty_like(r); rtg = torch.empty_like(r)
        for _ in range(2):
            K.gae_rtg_tmajor(r, v, b, b, None, adv_out=adv, rtg_out=rtg)
        reps = 5
        evs = [(K.event_create(), K.event_create()) for _ in range(reps)]
        for ev in evs:
            K.gae_rtg_tmajor(r, v, b, b, None, adv_out=adv, rtg_out=rtg, timing_events=ev)
        torch.cuda.synchronize()
        s = sum(K.event_elapsed_ms(a, c) for a, c in evs) * 1e-3 / reps
        bts = GAE_BYTES_PER_TRANSITION * T * Es
        out["roofline_saturating"] = {"kernel": "gae_rtg_stream_kernel<4, 8>", "bound": "hbm",
                                      "transitions": T * Es, "achieved": round(bts / s / 1e9, 1),
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(bts / s / 1e9 / HBM_PEAK_GBS, 4),
                                      "avg_launch_us": round(s * 1e6, 1),
                                      "traffic": pmc_traffic().get("void ppoaf::gae_rtg_stream_kernel<4, 8>")}
        del r, v, b, adv, rtg

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "C2":
        from oracle import cpu_ppo_loop
        Ec = args.cpu_sample_envs
        import os as _os
        cands = [c for c in (1, 4, 8, 16) if c <= (_os.cpu_count() or 1)]
        th = cpu_ppo_loop.pick_threads(cands, T)
        cb = cpu_ppo_loop.time_iteration(Ec, T, O, NA, epochs=args.epochs, batch_size=args.batch_size,
                                         threads=th)
        out["cpu_baseline"] = {"value": round(cb["env_steps_per_s"], 1), "unit": "env-steps/s",
                               "cores": cb["threads"], "kind": "port",
                               "sample": f"one PPO iteration of the same workload at envs_per_proc={Ec} "
                                         f"(T={T}, batch {args.batch_size}, {args.epochs} epochs; per-transition "
                                         f"cost is flat in E): rollout {cb['rollout_s']:.2f}s + update "
                                         f"{cb['update_s']:.2f}s; torch threads={cb['threads']} (fastest of "
                                         f"{cands} on a probe; the reference's default would be all "
                                         f"{os.cpu_count()} host cpus, which is slower)"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if mpi_utils.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
