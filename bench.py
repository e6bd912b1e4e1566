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

Rank 0 prints ONE JSON line:
  roofline            the GAE kernel (the kernel the metric names), HIP kernel events inside the timed region
  roofline_update     the kernel that dominates the step: FLOP per launch / its launch time vs the f32-MFMA peak.
                      Single rank, MLP policies: the two-XCD persistent kernel (ONE launch per epoch = all full
                      mini-batches; begin / end events of epoch launches made right after the timed region).
                      Otherwise K12 / K15 fwd_bwd (kernel begin/end events on eager launches of the same chain right
                      after the timed region -- the in-region launches are hipGraph nodes, which cannot carry stamps)
  config.other_configs  short runs (2 steps) of the other BASELINE configs' shapes (C3, C4, C5) in the same process
                      (N = 1; at N > 1 only with PPOAF_BENCH_OTHER_CONFIGS_MULTI=1)
  cpu_baseline        oracle/cpu_ppo_loop.py (a port with the reference's loop structure, pinned against fixtures
                      recorded from the reference) on a bounded sample: 1 process; cpu_baseline_mpi: R = 8 processes
                      in the reference's launch model (oracle/cpu_ddppo.py, gloo); cpu_baseline_c1: C1 exactly.
                      Rank 0, N = 1 only; fresh CPU-only child processes, run before the GPU is touched.
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
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA (v_mfma_f32_16x16x4_f32), 155 TF measured
GAE_BYTES_PER_TRANSITION = 16  # read r, V; write adv, rtg
STREAM_KERNEL = "void ppoaf::gae_rtg_stream_kernel<4, 1, 1024, false, 1>"   # the large-E form of K1 (csrc/gae.hip)


def device_clocks():
    """What the driver reports for the GPU's clocks right after the measurement (rocm-smi, read-only; None if unavailable)."""
    import subprocess
    try:
        p = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(p.stdout).values()))
        return {k: v for k, v in card.items() if "clock" in k.lower()}            # the driver's own wording, untouched
    except Exception:                                       # noqa: BLE001 -- a report field, never a reason to fail the bench
        return None


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
    for name, mult in (("ppoaf::gae_rtg_chunked_kernel", 1.0), (STREAM_KERNEL, 2.0)):
        pick = lambda c: next((v for (k, cc), v in vals.items() if cc == c and k.startswith(name)), None)   # template arguments vary
        f, w = pick("FETCH_SIZE"), pick("WRITE_SIZE")
        if f is not None and w is not None:
            out[name] = int((mult * f + w) * 1024)
    out["_source"] = os.path.relpath(files[-1], ROOT)
    return out


def update_pmc_traffic(config, kernel):
    """
    HBM-side bytes per launch of an update kernel from the committed PMC passes (profiles/*_update_pmc.csv:
    `bash tools/update_pmc.sh`, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of tools/update_pmc_driver.py,
    KB units).  FETCH_SIZE is doubled: these kernels read 16 B per lane, which gfx950 tallies at half
    (MI355X_MICROARCH.md, HBM), calibrated in this very access pattern on the slab reduce (16 slabs x 271 KB read:
    2 x FETCH_SIZE = 4.41 MB) and the Adam launch (4 buckets: 2 x FETCH_SIZE = 1.12 MB).  -> dict or None.
    """
    import glob, re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_update_pmc.csv")))
    for path in reversed(files):                    # the newest collection that holds this (config, kernel)
        vals = {}
        for line in open(path):
            m = re.match(r"^(C\d),(.*),(FETCH_SIZE|WRITE_SIZE),dispatches=(\d+),avg=([0-9.]+)", line)
            if m and m.group(1) == config and m.group(2).replace("ppoaf::", "").startswith(kernel.split("<")[0]):
                vals[m.group(3)] = (float(m.group(5)), int(m.group(4)))
        if len(vals) == 2:
            return {"bytes": int((2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024),
                    "fetch_kb_raw": vals["FETCH_SIZE"][0], "write_kb": vals["WRITE_SIZE"][0], "dispatches": vals["FETCH_SIZE"][1],
                    "source": os.path.relpath(path, ROOT)}
    return None


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
    p.add_argument("--no-other-configs", action="store_true")
    p.add_argument("--cpu-ranks", type=int, default=4,
                   help="ranks of the multi-process CPU baseline (the GPU pool allows at most 6 processes that have "
                        "loaded the ROCm runtime at once -- importing torch counts -- so 8 ranks cannot be started there)")
    p.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5"],
                   help="C2 (default, the metric's config) | C3 dims: HalfCheetah O=17, Box(6), actor 128^3 / "
                        "critic 256^3, E=2048, ICM + obs/reward normalisers and clippers | C4 dims: SimpleSpread MAPPO, 3 agents, "
                        "O=18, O_c=54, Discrete(5), E=1024 per rank | C5 dims: same env, MATPolicy (embedding 64, "
                        "1 block, 1 head, critic view local), E=1024 per rank")
    return p.parse_args()


def cpu_baselines(args):
    """
    The reference's CPU path, timed on this box's host cores BEFORE anything touches the GPU, in fresh CPU-only
    child processes (oracle/cpu_ddppo.py): one process (best of a few intra-op thread counts, found on a tiny
    probe: the reference's own default -- all host cpus per process -- is far slower for 128-wide layers), R
    processes in its mpirun launch model, and C1 exactly.
    """
    from oracle import cpu_ddppo
    # the host cores this job may use: its affinity mask, capped at the pool's per-GPU CPU share (16)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    try:
        model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        model = "unknown"
    T, B, K = args.ts, args.batch_size, args.epochs
    cands = [c for c in (1, 4, 8) if c <= cores]
    probe = {th: cpu_ddppo.run_ranks(1, 16, T, 1, B, th)["env_steps_per_s"] for th in cands}
    th = max(probe, key=probe.get)
    Ec = args.cpu_sample_envs
    one = cpu_ddppo.run_ranks(1, Ec, T, K, B, th)
    out = {"cpu_baseline": {
        "value": round(one["env_steps_per_s"], 1), "unit": "env-steps/s", "cores": th, "kind": "port",
        "sample": f"one PPO iteration of the C2 workload at envs_per_proc={Ec} (T={T}, batch {B}, {K} epochs; per-transition "
                  f"cost is flat in E): rollout {one['rollout_s']:.2f}s + update {one['update_s']:.2f}s; 1 process, torch "
                  f"threads={th} (fastest of {cands} on a probe: {({k: round(v) for k, v in probe.items()})} env-steps/s)",
        "cpu_model": model, "host_cpus": os.cpu_count(), "cores_used_at_most": cores}}
    R = max(1, min(args.cpu_ranks, cores))
    if R > 1:
        tpr = max(cores // R, 1)                                   # utils/mpi_utils.py:37-48
        Er = max(Ec // R, 8)
        many = cpu_ddppo.run_ranks(R, Er, T, K, B, tpr)
        out["cpu_baseline_mpi"] = {
            "value": round(many["env_steps_per_s"], 1), "unit": "env-steps/s", "ranks": R, "cores": R * tpr, "kind": "port",
            "cpu_model": model,
            "sample": f"the reference's launch model (mpirun -n {R}; {R} ranks is what this pool's process guard admits): "
                      f"{R} gloo processes x envs_per_proc={Er}, {tpr} torch "
                      f"thread(s) each, per mini-batch an all-gather of the raw rewards-to-go, one all-reduce per parameter "
                      f"tensor and a barrier; rollout {many['rollout_s']:.2f}s + update {many['update_s']:.2f}s"}
    c1 = cpu_ddppo.run_ranks(1, 8, 128, 10, 256, 1)
    out["cpu_baseline_c1"] = {"value": round(c1["env_steps_per_s"], 1), "unit": "env-steps/s", "cores": 1, "kind": "port",
                              "sample": "BASELINE configs[0] exactly: envs_per_proc=8, ts_per_rollout=128, 1 CPU rank, 1 thread, "
                                        f"batch 256, 10 epochs: {c1['wall_s']:.2f}s per iteration"}
    return out


CONFIGS = ("C2", "C3", "C4", "C5")


def build_config(name, args, device, rank):
    """-> (ppo, pol, dims) for one BASELINE config's shapes (SURVEY.md §8 sizes)."""
    import numpy as np
    from ppo_and_friends_amd.ppo import PPO
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.spaces import Box, Discrete
    T = args.ts
    E, O, NA = args.envs, 4, 2
    A, critic_view, act_space, pargs, policy_class = 1, "local", Discrete(NA), {}, None
    filters = dict(normalize_obs=False, normalize_rewards=False)     # C2/C4/C5 as SURVEY.md §8(d) defines them
    workload = (f"C2 CartPole-v1 MLP (4->128x3->2 actor, ->1 critic), envs_per_proc={E}, ts_per_rollout={T}, "
                "fixed-length synthetic trajectories")
    if name == "C3":
        E, O = (2048 if args.envs == 4096 else args.envs), 17
        act_space = Box(-1.0, 1.0, (6,), np.float32)
        pargs = dict(actor_kw_args=dict(hidden_size=128), critic_kw_args=dict(hidden_size=256), enable_icm=True)
        filters = dict(normalize_obs=True, normalize_rewards=True, obs_clip=(-10.0, 10.0), reward_clip=(-10.0, 10.0))
        workload = (f"C3 dims (HalfCheetah-v4: O=17, Box(6) tanh-Gaussian, actor 128^3, critic 256^3, ICM, "
                    f"obs/reward normalisers + clippers), envs_per_proc={E}, ts_per_rollout={T}")
    elif name == "C4":
        E, O, NA, A, critic_view = (1024 if args.envs == 4096 else args.envs), 18, 5, 3, "policy"
        act_space = Discrete(NA)
        pargs = dict(actor_kw_args=dict(hidden_size=128), critic_kw_args=dict(hidden_size=256))
        workload = (f"C4 dims (MPE simple_spread MAPPO: 3 agents share one policy, O=18, O_c=54, Discrete(5), "
                    f"actor 128^3, critic 256^3), envs_per_proc={E}, ts_per_rollout={T}")
    elif name == "C5":
        from ppo_and_friends_amd.policies.mat_policy import MATPolicy
        E, O, NA, A, critic_view = (1024 if args.envs == 4096 else args.envs), 18, 5, 3, "local"
        act_space, policy_class, pargs = Discrete(NA), MATPolicy, {}
        workload = (f"C5 dims (MPE simple_spread MATPolicy: 3 agents, O=18, Discrete(5), embedding 64, 1 block, "
                    f"1 head, attention core on f32 MFMA), envs_per_proc={E}, ts_per_rollout={T}")
    env_gen = lambda: SyntheticFixedLengthEnv(E, O, act_space, T, device, reward="ones" if name == "C2" else "uniform",
                                              seed=1234, rank=rank, num_agents=A, critic_view=critic_view)
    obs_space = Box(-np.inf, np.inf, (O,), np.float32)
    cobs_space = Box(-np.inf, np.inf, (O * A if critic_view == "policy" else O,), np.float32)
    settings = {"cartpole": (policy_class, obs_space, cobs_space, act_space, pargs)}
    ppo = PPO(env_gen, settings, device=device, random_seed=1, envs_per_proc=E, ts_per_rollout=T,
              batch_size=args.batch_size, epochs_per_iter=args.epochs, use_graphs=not args.no_graphs, save_state=False, **filters)
    workload += f", batch_size={args.batch_size}, epochs_per_iter={args.epochs}"
    return ppo, ppo.policies["cartpole"], dict(E=E, T=T, A=A, O=O, workload=workload)




def update_kernel_roofline(ppo, pol, B, launches=64, config="C2"):
    """
    roofline_update: the kernel that dominates the step.  After the timed region the same per-mini-batch chain
    (fwd_bwd -> reduce -> Adam, so every launch sees weights the Adam launch has just rewritten, as in the timed
    chain) is launched eagerly `launches` times with the fwd_bwd kernel's own begin / end stamped into HIP events.
    FLOP per launch = 3 x (forward FLOP per row) x rows: forward, dgrad and wgrad each cost one forward's worth.
    """
    import ctypes as C
    import torch
    from ppo_and_friends_amd import _lib
    from ppo_and_friends_amd import kernels as K
    fused = ppo._fused_updater("cartpole", B)
    if fused is None:
        return None
    lib, st = _lib.load(), K.stream()
    ppo.rollout()                                                       # a fresh dataset for the probe's epoch tables
    pol.train()
    N = pol.buffer.num_transitions
    fused.begin_epoch(torch.randperm(N, device=pol.device))
    args = fused._args_for(B)
    ref = C.byref(args)
    lin = lambda net: sum(2 * m.weight.numel() for m in net.modules() if isinstance(m, torch.nn.Linear))
    evs = []
    if pol.agent_grouping:                                              # K15: one row = one env = A tokens
        A = pol.num_agents
        lin_f, att_f = A * lin(pol.actor_critic), 3 * 2 * 2 * A * A * 64   # linears per token; 3 attention cores (QK^T, PV)
        fwd = lin_f + att_f
        kernel, desc = "mat_update_fwd_bwd_kernel", f"3 x ({A} tokens x 2 x sum(Linear weights) + 3 attention cores) x B"
        mat_flop = 3 * fwd * B
        if getattr(fused, "split", False):
            # split-wgrad chain: the linears' weight-gradient third runs in mat_update_wgrad_kernel (the narrow first layers'
            # and the heads' few wgrad FLOPs stay here and are not counted)
            kernel = "mat_update_fwd_bwd_kernel<true>"
            desc = (f"(2 x {A} tokens x 2 x sum(Linear weights) + 3 x 3 attention cores) x B (forward + dgrad; the linears' "
                    "wgrad third runs in mat_update_wgrad_kernel)")
            mat_flop = (2 * lin_f + 3 * att_f) * B
        opt, ac = pol.actor_critic_optim, pol.actor_critic
        clip = pol.gradient_clip
        for _ in range(launches):
            ev = (K.event_create(), K.event_create())
            _lib.check(lib.ppoaf_mat_update_fwd_bwd_timed(ref, ev[0], ev[1], st), "mat fwd_bwd")
            _lib.check(lib.ppoaf_mat_update_reduce(ref, st), "mat reduce")
            _lib.check(lib.ppoaf_adam_step_prenormed(
                ac.flat_params.data_ptr(), ac.flat_grads.data_ptr(), opt.exp_avg.data_ptr(), opt.exp_avg_sq.data_ptr(),
                ac.flat_params.numel(), opt.step_count.data_ptr(), opt.lr.data_ptr(), opt.betas[0], opt.betas[1], opt.eps,
                1.0, float(clip) if clip is not None else 0.0, opt.norm_scratch.data_ptr(), fused.norm_partials,
                opt.grad_norm.data_ptr(), st), "adam")
            evs.append(ev)
    else:
        fwd = lin(pol.actor) + lin(pol.critic)
        ha, hc = args.actor.hidden // 16, args.critic.hidden // 16
        kernel, desc = f"ppo_update_fwd_bwd_kernel<{ha}, {hc}>", "3 x 2 x sum(Linear weights of actor + critic) x B"
        passes = 3
        if fused.split:
            # split-wgrad chain: this kernel runs the forward and the input-gradient pass; the weight-gradient third is
            # ppo_update_wgrad_kernel's (the output layer's few wgrad FLOPs stay here and are not counted)
            passes = 2
            desc = ("2 x 2 x sum(Linear weights of actor + critic) x B (forward + dgrad; the wgrad third runs in the tail "
                    "launch, ppo_update_wgrad_adam_kernel)")
        if fused.split:
            kernel = kernel.replace(">", ", true>")                  # the split-wgrad instantiation of the same kernel
        if getattr(args, "row_pairs", 0):
            kernel = f"ppo_update_fwd_bwd_pair_kernel<{ha}, {hc}>"   # a 256-wide network's row tiles on workgroup pairs
        for _ in range(launches):
            ev = (K.event_create(), K.event_create())
            fused.gradient_only(args, ev)                          # fwd_bwd (timed) + wgrad launch / slab reduce
            _lib.check(lib.ppoaf_ppo_update_adam(ref, 3 if fused.split else 0, st), "adam")
            evs.append(ev)
    torch.cuda.synchronize()
    us = sorted(K.event_elapsed_ms(a, b) * 1e3 for a, b in evs)
    avg = sum(us) / len(us)
    flop = mat_flop if pol.agent_grouping else passes * fwd * B
    tf = flop / (avg * 1e-6) / 1e12
    pmc = update_pmc_traffic(config, kernel)
    return {"kernel": kernel, "bound": "mfma", "achieved": round(tf, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 5), "flop_per_launch": int(flop), "flop_formula": desc,
            "avg_launch_us": round(avg, 2), "median_launch_us": round(us[len(us) // 2], 2), "launches": len(us),
            "launches_per_step": (N // B) * ppo.epochs_per_iter,
            "timing": "kernel begin/end events (hipExtLaunchKernelGGL) on eager launches of the update chain right after the timed "
                      "region (in-region launches are hipGraph nodes: 18.3 us for this kernel at C2 under replay, "
                      "profiles/r04_C2_kernel_stats.csv); profiles/ holds the rocprofv3 summary of the same command",
            "traffic": None if pmc is None else pmc["bytes"],
            "traffic_source": None if pmc is None else f"{pmc['source']}: 2 x FETCH_SIZE + WRITE_SIZE per launch ({pmc['dispatches']} launches)"}


def run_config(name, args, device, rank, world, steps, warmup, with_gae_roofline):
    """Time `steps` PPO iterations of one config -> result dict (value, ms_per_step, rooflines, exchange info)."""
    import torch
    import torch.distributed as dist
    from ppo_and_friends_amd.utils import mpi_utils
    from ppo_and_friends_amd import kernels as K
    import ppo_and_friends_amd.utils.episode_info as ei

    ppo, pol, d = build_config(name, args, device, rank)
    E, T, A = d["E"], d["T"], d["A"]
    from ppo_and_friends_amd.fused_update import FusedPolicyUpdate

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
    try:
        for _ in range(warmup):
            iteration(False)
        barrier()
        t0 = time.perf_counter()
        step_ts = []
        for _ in range(steps):
            iteration(True)
            if os.environ.get("PPOAF_BENCH_STEP_TIMES") == "1":           # diagnostic only: one host sync per step
                torch.cuda.synchronize()
                step_ts.append(round(time.perf_counter() - t0, 4))
        if step_ts:
            print(f"[bench] {name}: cumulative step end times (s): {step_ts}", file=sys.stderr, flush=True)
        barrier()
        dt = time.perf_counter() - t0
    finally:
        ei.RolloutBuffer.compute_advantages = orig
    dt = float(mpi_utils.allreduce_scalars([dt], op="max")[0])        # MAX over ranks
    res = {"value": round(world * E * T * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "warmup": warmup, "workload": d["workload"], "E": E, "T": T, "A": A,
           "rollout_s": round(ppo.status_dict["global status"]["rollout time"], 4),
           "train_s": round(ppo.status_dict["global status"]["train time"], 4)}
    if with_gae_roofline:
        gae_ms = [K.event_elapsed_ms(a, b) for a, b in gae_events]
        gae_avg_s = (sum(gae_ms) / max(len(gae_ms), 1)) * 1e-3
        gae_bytes = GAE_BYTES_PER_TRANSITION * E * T * A
        achieved = gae_bytes / gae_avg_s / 1e9 if gae_avg_s > 0 else 0.0
        pmc = pmc_traffic() if (E, T) == (4096, 128) else {}
        res["roofline"] = {
            "kernel": ("gae_rtg_chunked_kernel<32, 8, 8>" if E >= 8192 else "gae_rtg_chunked_kernel<16, 8, 4>") if E < (1 << 17) else "gae_rtg_stream_kernel", "bound": "hbm",
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": pmc.get("ppoaf::gae_rtg_chunked_kernel"), "traffic_source": pmc.get("_source"),
            "bytes_per_launch": gae_bytes, "avg_launch_us": round(gae_avg_s * 1e6, 3), "launches": len(gae_ms),
            "median_launch_us": round(sorted(gae_ms)[len(gae_ms) // 2] * 1e3, 3) if gae_ms else None,
            "min_launch_us": round(min(gae_ms) * 1e3, 3) if gae_ms else None,
            "max_launch_us": round(max(gae_ms) * 1e3, 3) if gae_ms else None,
            "timing": "kernel begin/end events (hipExtLaunchKernelGGL) on the launch stream, timed region",
            "note": "config size (8.4 MB, fits L2/MALL) is latency-bound; see roofline_saturating"}

    # which per-mini-batch gradient exchange the update loops actually used (N > 1 or its rehearsal)
    fused = [f for f in getattr(ppo, "_fused", {}).values() if f is not None]
    peer = bool(fused) and all(getattr(f, "xchg", None) is not None for f in fused)
    c_loop = FusedPolicyUpdate._rccl_comm_cache not in ("unset", None)
    exchange = None if not mpi_utils.distributed_path() else \
        ("K17 peer mappings (xGMI), in-graph" if peer else
         ("RCCL all-reduce, chain issued from C (ppoaf_ppo_update_chain_allreduce)" if c_loop else "RCCL all-reduce, eager loop"))
    if exchange is not None and fused:                       # why that path (self-test verdict / fallback reason)
        exchange += f" [{getattr(fused[0], 'xchg_reason', '')}]"
    res["gradient_exchange"], res["peer"] = exchange, peer

    # N > 1 only, after the timed region: what one gradient exchange of this bucket costs on this node, with the
    # K17 kernel and with the process group's all-reduce (back-to-back launches, stream events) -- recorded so that
    # a multi-GPU run documents the latency its scaling number rests on
    probe = None
    if mpi_utils.distributed_path() and fused and with_gae_roofline:
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
    res["exchange_probe"] = probe
    if not mpi_utils.distributed_path():                      # single rank: the eager chain is this rank's own
        res["roofline_update"] = update_kernel_roofline(ppo, pol, args.batch_size, config=name)
    del ppo, pol, fused
    import gc
    gc.collect()                 # the trainer holds reference cycles (graphs, hooks): release its device memory and
    torch.cuda.empty_cache()     # graph executables NOW, not in the middle of the next config's timed region
    return res


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON: libraries chat on file descriptor 1 (RCCL prints its version banner
    # there when the process group initialises), so everything but the final line is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # dmabuf IPC (the K17 peer mappings, RCCL's own P2P): must be in the environment before HIP initialises
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    env_rank = int(os.environ.get("RANK", "0"))
    cpu = {}
    if env_world == 1 and env_rank == 0 and not args.no_cpu_baseline and args.config == "C2":
        cpu = cpu_baselines(args)                             # CPU-only child processes, before the GPU is touched

    import torch
    import torch.distributed as dist
    from ppo_and_friends_amd.utils import mpi_utils
    from ppo_and_friends_amd import kernels as K
    from ppo_and_friends_amd import _lib

    rank, world, local_rank = mpi_utils.init_process_group_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("PPOAF_SHARE_DEVICE", "0") == "1":
        local_rank = 0          # rehearsal on a one-GPU box (with PPOAF_BACKEND=gloo): all ranks on device 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    notes = {}

    def run_measured(name, steps, warmup, with_gae):
        return run_config(name, args, device, rank, world, steps, warmup, with_gae)

    main_res = run_measured(args.config, args.steps, args.warmup, True)
    update_note = notes.get(args.config)
    others = {}
    # N > 1: the line the driver's scaling run reads is C2's; the other shapes' N > 1 paths (256-wide critics on row pairs,
    # overlapped PPO / ICM epochs) are rehearsed by `PPOAF_REHEARSE_MULTI_RANK=1 bench.py --config <C>` and the two-process
    # tests, and run here only on request -- a failure in one of them must not cost the headline measurement
    others_wanted = world == 1 or os.environ.get("PPOAF_BENCH_OTHER_CONFIGS_MULTI", "0") == "1"
    if args.config == "C2" and not args.no_other_configs and others_wanted:
        for name in CONFIGS:
            if name == "C2":
                continue
            r = run_measured(name, 2, 1, False)
            others[name] = {"value": r["value"], "unit": "env-steps/s", "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                            "warmup": r["warmup"], "workload": r["workload"],
                            "agent_steps_per_iteration": world * r["E"] * r["T"] * r["A"],
                            "gradient_exchange": r["gradient_exchange"], "roofline_update": r.get("roofline_update")}

    E, T, A = main_res["E"], main_res["T"], main_res["A"]
    out = {"metric": "env_steps_per_sec", "value": main_res["value"], "unit": "env-steps/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": main_res["ms_per_step"], "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": main_res["workload"],
                      "agent_steps_per_iteration": world * E * T * A,
                      "global_env_steps_per_iteration": world * E * T,
                      "parallelism": f"dp{world}",
                      "hip_graphs": (not args.no_graphs) and (not mpi_utils.distributed_path() or main_res["peer"]),
                      "multi_rank_path": mpi_utils.distributed_path(),
                      "gradient_exchange": main_res["gradient_exchange"], "exchange_probe": main_res["exchange_probe"],
                      "rollout_s": main_res["rollout_s"], "train_s": main_res["train_s"],
                      "update_kernel": update_note or (main_res.get("roofline_update") or {}).get("kernel"),
                      "other_configs": others or None},
           "roofline": main_res["roofline"]}
    if main_res.get("roofline_update") is not None:
        out["roofline_update"] = main_res["roofline_update"]

    if rank == 0 and not args.no_saturating:
        # companion: same kernel at a bandwidth-saturating size (N = 2^28 transitions)
        Es = (1 << 28) // T
        r = torch.rand(T, Es, device=device); v = torch.randn(T, Es, device=device)
        b = torch.randn(Es, device=device)
        adv = torch.empty_like(r); rtg = torch.empty_like(r)
        for _ in range(3):
            K.gae_rtg_tmajor(r, v, b, b, None, adv_out=adv, rtg_out=rtg)
        # protocol (round 4): 20 timed launches back to back after 3 warm-ups; `achieved` comes from the MEDIAN launch,
        # min / max and the clocks the driver reports ride along -- five launches and a mean (rounds 1-3) moved +-9 %
        # between boxes of the pool (5.17 and 5.91 TB/s on two driver runs of the same kernel)
        reps = 20
        evs = [(K.event_create(), K.event_create()) for _ in range(reps)]
        for ev in evs:
            K.gae_rtg_tmajor(r, v, b, b, None, adv_out=adv, rtg_out=rtg, timing_events=ev)
        torch.cuda.synchronize()
        us = sorted(K.event_elapsed_ms(a, c) * 1e3 for a, c in evs)
        med = 0.5 * (us[reps // 2 - 1] + us[reps // 2])
        bts = GAE_BYTES_PER_TRANSITION * T * Es
        gbs = lambda t_us: round(bts / (t_us * 1e-6) / 1e9, 1)
        out["roofline_saturating"] = {"kernel": "gae_rtg_stream_kernel<4, 1, 1024>", "bound": "hbm",
                                      "transitions": T * Es, "achieved": gbs(med),
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(gbs(med) / HBM_PEAK_GBS, 4),
                                      "avg_launch_us": round(sum(us) / reps, 1), "median_launch_us": round(med, 1),
                                      "min_launch_us": round(us[0], 1), "max_launch_us": round(us[-1], 1), "launches": reps,
                                      "achieved_best": gbs(us[0]), "achieved_worst": gbs(us[-1]),
                                      "clocks": device_clocks(),
                                      "traffic": pmc_traffic().get(STREAM_KERNEL), "traffic_source": pmc_traffic().get("_source")}
        del r, v, b, adv, rtg

    out.update(cpu)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if mpi_utils.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
