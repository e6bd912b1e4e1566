"""
-m gpu: the N>1 DD-PPO control flow on real kernels.  Two processes share the one GPU of the box
(collectives over gloo, staged through the host -- RCCL refuses two ranks on one device); each rank has
its own envs / seeds, exactly as under torchrun.  Checked against the reference's DD-PPO semantics
restated on the CPU (oracle/cpu_ppo_loop.ddppo_train_epoch): moment records of every rank feed every
rank's value normaliser, gradients are averaged per mini-batch, all ranks stay weight-identical.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

E, T, B, O, NA, SEED = 12, 16, 32, 4, 2, 11


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank(rank, world, port, out, mode):
    if mode == "peer_slabs":
        # the slab chain with K17 fused into the slab reduce launch: what N > 1 ranks ran before the fused tail launch
        # (and still run where the tail's exchange cannot be opened)
        mode = "peer"
        os.environ["PPOAF_FUSED_TAIL"] = "0"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
                      PPOAF_GRAD_EXCHANGE=mode)
    import torch.distributed as dist
    from ppo_and_friends_amd.utils import mpi_utils
    mpi_utils.init_process_group_from_env(backend="gloo")
    from ppo_and_friends_amd.ppo import PPO, PermutationLoader
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.spaces import Box, Discrete
    dev = torch.device("cuda", 0)
    env_gen = lambda: SyntheticFixedLengthEnv(E, O, Discrete(NA), T, dev, reward="uniform", seed=500, rank=rank)
    sp = Box(-np.inf, np.inf, (O,), np.float32)
    ppo = PPO(env_gen, {"p": (None, sp, sp, Discrete(NA), {})}, device=dev, random_seed=SEED, normalize_obs=False, normalize_rewards=False,
              envs_per_proc=E, ts_per_rollout=T, batch_size=B, epochs_per_iter=2, update_mode="fused")
    pol = ppo.policies["p"]
    w0 = pol.policy_params.detach().cpu().clone()           # after the rank-0 broadcast
    ppo.rollout()
    loader = PermutationLoader(pol.dataset, B, ppo.loader_generator)
    pol.train()
    stats = []
    for _ in range(2):
        ppo._ppo_batch_train(loader, "p")
        stats.append(dict(ppo.status_dict["p"]))
    vs = ppo.value_normalizers["p"].running_stats
    fused = [f for f in getattr(ppo, "_fused", {}).values() if f is not None]
    from ppo_and_friends_amd.fused_update import FusedPolicyUpdate
    out[rank] = dict(w0=w0, w=pol.policy_params.detach().cpu().clone(),
                     peer_exchange=[getattr(f, "xchg", None) is not None for f in fused],
                     split=[bool(f.split) for f in fused], wgrad_exchange=[getattr(f, "xchg_sp", None) is not None for f in fused],
                     exp_avg=pol.policy_exp_avg.detach().cpu().clone(),
                     actor_sd={k: v.detach().cpu().clone() for k, v in pol.actor.state_dict().items()},
                     critic_sd={k: v.detach().cpu().clone() for k, v in pol.critic.state_dict().items()},
                     obs=ppo.env.obs_table.cpu().numpy(), rew=ppo.env.reward_table.cpu().numpy(),
                     actions=pol.buffer.actions[..., 0].cpu().numpy(),
                     stats=[{k: float(v) for k, v in s.items() if isinstance(v, (int, float)) and not isinstance(v, bool)} for s in stats],
                     vn=np.array([vs.mean, vs.variance, vs.count], dtype=np.float64))
    dist.barrier()
    dist.destroy_process_group()


# "peer": K17 exchange over IPC mappings inside hipGraph-replayed chains: the split-wgrad chain whose fused tail launch
# carries the exchange inside every weight-gradient job (ppoaf_ppo_update_wgrad_adam_exchange: two launches per
# mini-batch); "peer_slabs": the slab chain, K17 fused into the slab reduce launch (PPOAF_FUSED_TAIL=0); "rccl":
# the eager loop with the process group's all-reduce (gloo here, staged through the host)
@pytest.fixture(scope="module", params=["peer", "peer_slabs", "rccl"])
def run2(request):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank, args=(world, _free_port(), out, request.param), nprocs=world, join=True)
    res = [out[r] for r in range(world)]
    for r in res:
        assert r["peer_exchange"] == [request.param.startswith("peer")], r["peer_exchange"]
        if request.param in ("peer", "peer_slabs"):
            assert r["split"] == r["wgrad_exchange"] == [request.param == "peer"], (r["split"], r["wgrad_exchange"])
    res[0]["mode"] = request.param
    return res


def test_ranks_start_and_stay_identical(run2):
    r0, r1 = run2
    assert torch.equal(r0["w0"], r1["w0"]), "rank-0 broadcast of the policy bucket"
    assert torch.equal(r0["w"], r1["w"]), "synchronous DD-PPO keeps replicas identical"
    assert r0["stats"] == r1["stats"]
    np.testing.assert_array_equal(r0["vn"], r1["vn"])
    assert not np.array_equal(r0["obs"], r1["obs"]), "each rank rolls out its own envs"


def test_two_rank_update_matches_ddppo_oracle(run2):
    from oracle import cpu_ppo_loop
    ranks = []
    for r, res in enumerate(run2):
        cpu = cpu_ppo_loop.CpuPPO(O, NA, batch_size=B, seed=SEED + r)
        # weights BEFORE training = rank 0's broadcast weights: rebuild them from w0 via rank 0's layout
        ranks.append((cpu, res))
    # initial weights: take them from a fresh policy bucket layout (actor then critic, padded to 4 floats)
    w0 = run2[0]["w0"].numpy()

    def load(net, keys_sd, flat, off):
        sd = {}
        for k, v in keys_sd.items():
            if not k.startswith("sequential_net."):
                continue
            n = v.numel()
            sd[k.replace("sequential_net.", "")] = torch.tensor(flat[off:off + n]).reshape(v.shape).clone()
            off += (n + 3) // 4 * 4
        net.load_state_dict(sd)
        return off

    for cpu, res in ranks:
        off = load(cpu.actor, res["actor_sd"], w0, 0)
        load(cpu.critic, res["critic_sd"], w0, off)
        cpu.loader_generator = torch.Generator().manual_seed(SEED + ranks.index((cpu, res)))
        cpu.rollout(res["obs"], res["rew"], actions=res["actions"])
    cpus = [c for c, _ in ranks]
    for epoch in range(2):
        ref = cpu_ppo_loop.ddppo_train_epoch(cpus)
        got = run2[0]["stats"][epoch]
        for k in ("actor loss", "critic loss", "kl avg", "weighted entropy"):
            np.testing.assert_allclose(got[k], ref[k], rtol=2e-5, atol=2e-6, err_msg=f"epoch {epoch} {k}")
    flat_ref = np.concatenate([p.detach().reshape(-1).numpy() for p in cpus[0].actor.parameters()])
    got_actor = np.concatenate([v.reshape(-1).numpy() for k, v in run2[0]["actor_sd"].items()])
    np.testing.assert_allclose(got_actor, flat_ref, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(run2[0]["vn"][:2], [cpus[0].value_stats.mean, cpus[0].value_stats.variance],
                               rtol=1e-5, atol=1e-6)
    assert run2[0]["vn"][2] == cpus[0].value_stats.count


# ----------------------------------------------------------------------------------------------------
# Two ranks against TWO RANKS OF THE REFERENCE (fixtures g12_*_r2*, recorded under the two-process mpi4py stand-in of
# tests/golden/ref_import.py): no oracle in between.
def _fixture_rank(rank, world, port, out, name, mode):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", PPOAF_GRAD_EXCHANGE=mode, PPOAF_SHARE_DEVICE="1")
    import torch.distributed as dist
    from ppo_and_friends_amd.utils import mpi_utils
    mpi_utils.init_process_group_from_env(backend="gloo")
    import test_gpu_reference_golden as G
    from ppo_and_friends_amd.fused_update import FusedPolicyUpdate
    g = G.RankView(np.load(os.path.join(here, "golden", name + ".npz"), allow_pickle=False), rank)
    dev = torch.device("cuda", 0)
    probe = {}

    def first_minibatch(ppo, pol, pi):
        """ONE mini-batch through the selected N > 1 path with everything it advances put back afterwards: what the
        exchange left in the gradient bucket is the SUM over the ranks (1/R is folded into the Adam kernel) =
        R x what mpi_avg_gradients left in .grad on the reference's ranks (utils/mpi_utils.py:89-111)."""
        c = G._cfg(g)
        fused = ppo._fused_updater("agent", c["batch_size"])
        perm = torch.as_tensor(np.asarray(pi[g["epoch_perms"][0]], dtype=np.int64), device=dev)
        fused.begin_epoch(perm)
        state = [pol.policy_params, pol.policy_exp_avg, pol.policy_exp_avg_sq, pol.policy_step_counts, pol.policy_norm_scratch,
                 fused.vn_mean, fused.vn_var, fused.vn_count, fused.cursor, fused.totals, pol.buffer.values]
        keep = [t.clone() for t in state]
        n_full, tail, n_done = fused.n_full, fused.tail, fused.n_done
        fused.n_full, fused.tail = 1, 0
        try:
            fused.run_epoch()
            torch.cuda.synchronize()
        finally:
            fused.n_full, fused.tail, fused.n_done = n_full, tail, n_done
        grads = pol.policy_grads.clone()
        for t, k in zip(state, keep):
            t.copy_(k)
        for tag, net in (("actor", pol.actor), ("critic", pol.critic)):
            got = G.params_in_bucket_order(pol, grads, net) / world
            want = g[f"mb0_{tag}_avg_grad"]
            scale = np.abs(want).max()
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5 * scale, err_msg=f"{tag}: rank-averaged gradient")
        probe["done"] = True

    # (the ICM scenario's PPO / ICM epoch pairs have no single-launch probe of this kind; their first mini-batch is covered by
    # the epoch statistics)
    want_probe = name == "g12_c2_r2" or (name == "g12_c4_r2" and mode == "rccl")
    ppo, ran = G.run_kl_stop_scenario(g, name, "fused", dev, first_minibatch=first_minibatch if want_probe else None)
    pol = ppo.policies["agent"]
    fused = [f for f in getattr(ppo, "_fused", {}).values() if f is not None]
    out[rank] = dict(ran=ran, probe=bool(probe), w=pol.policy_params.detach().cpu().clone(),
                     w_icm=pol.icm_model.flat_params.detach().cpu().clone() if pol.enable_icm else None,
                     peer_exchange=[getattr(f, "xchg", None) is not None for f in fused],
                     kl=float(ppo.status_dict["agent"]["kl avg"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["peer", "rccl"])
@pytest.mark.parametrize("name", ["g12_c2_r2", "g12_c4_r2", "g12_c2_icm_r2_klstop"])
def test_two_ranks_reproduce_two_ranks_of_the_reference(name, mode):
    """
    Each process is one rank of the product holding that rank's tables, actions and shuffles of the reference's own
    two-rank run (`mpirun -n 2` semantics through the mpi4py stand-in): rollout, dataset and the all-reduced status
    block, the rank-averaged first-mini-batch gradient (mpi_avg_gradients, utils/mpi_utils.py:89-111), every epoch's
    all-reduced statistics (ppo.py:2468-2475) and the value normaliser fed with both ranks' data (utils/stats.py:47-50),
    the KL early stop taken by both ranks after the same epoch (ppo.py:2221-2232; g12_c2_icm_r2_klstop: after 3 of 4
    epochs), final weights.  `peer`: the K17 exchange inside graph-replayed chains (the 256-wide critic of g12_c4_r2 on
    workgroup pairs; PPO and ICM epochs overlapped on two streams); `rccl`: the all-reduce loops.
    All fixture checks run inside the rank processes (test_gpu_reference_golden.run_kl_stop_scenario).
    """
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"), allow_pickle=False)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_fixture_rank, args=(2, _free_port(), out, name, mode), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    want = [int(x) for x in g["r0.epochs_run"]] if "r0.epochs_run" in g.files else None
    for r in (r0, r1):
        assert r["peer_exchange"] and all(x == (mode == "peer") for x in r["peer_exchange"]), r["peer_exchange"]
        if want is not None:
            assert r["ran"] == want and want[0] < 4, (r["ran"], want)      # both ranks left the epoch loop together, early
    assert r0["ran"] == r1["ran"] and r0["kl"] == r1["kl"]
    assert torch.equal(r0["w"], r1["w"]), "synchronous DD-PPO keeps replicas identical"
    if r0["w_icm"] is not None:
        assert torch.equal(r0["w_icm"], r1["w_icm"])


# ----------------------------------------------------------------------------------------------------
# ICM (K14) and MAT (K15) updates on two ranks: the K17 exchange inside hipGraph-replayed chains against
# the eager loop with the process group's all-reduce -- same sums, so the same training.
def _rank_kind(rank, world, port, out, mode, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PPOAF_GRAD_EXCHANGE=mode,
                      PPOAF_SHARE_DEVICE="1")          # one GPU for both ranks
    import torch.distributed as dist
    from ppo_and_friends_amd.utils import mpi_utils
    mpi_utils.init_process_group_from_env(backend="gloo")
    from ppo_and_friends_amd.ppo import PPO
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.spaces import Box, Discrete
    dev = torch.device("cuda", 0)
    if kind in ("wide", "wide_tail"):
        # C4 shape: 3 agents share the policy, 128-wide actor, 256-wide critic on the concatenated observations.  "wide": 369
        # weight-gradient workgroups per rank -- two ranks on ONE GPU cannot keep 2 x 369 resident, so the chain ends in the wgrad
        # launch + a K17 launch + Adam (on a node: the fused tail).  "wide_tail": shallower networks (170 workgroups per rank), so
        # that the two ranks here DO run the path a node runs -- row pairs + the fused tail launch with the exchange inside
        E_, T_, O_, NA_, B_, A_ = 8, 32, 18, 5, 32, 3
        env_gen = lambda: SyntheticFixedLengthEnv(E_, O_, Discrete(NA_), T_, dev, reward="uniform", seed=79, rank=rank,
                                                  num_agents=A_, critic_view="policy")
        sp, csp = Box(-np.inf, np.inf, (O_,), np.float32), Box(-np.inf, np.inf, (A_ * O_,), np.float32)
        depths = (1, 2) if kind == "wide_tail" else (3, 3)
        settings = {"p": (None, sp, csp, Discrete(NA_), dict(actor_kw_args=dict(hidden_size=128, hidden_depth=depths[0]),
                                                             critic_kw_args=dict(hidden_size=256, hidden_depth=depths[1])))}
    elif kind in ("icm", "guard"):
        E_, T_, O_, NA_, B_ = 16, 64, 6, 3, 16                 # 64 mini-batches per epoch: two graph chunks
        env_gen = lambda: SyntheticFixedLengthEnv(E_, O_, Discrete(NA_), T_, dev, reward="uniform", seed=77, rank=rank)
        sp = Box(-np.inf, np.inf, (O_,), np.float32)
        settings = {"p": (None, sp, sp, Discrete(NA_), dict(enable_icm=kind == "icm"))}
    else:
        from ppo_and_friends_amd.policies.mat_policy import MATPolicy
        E_, T_, O_, NA_, B_, A_ = 16, 40, 18, 5, 16, 3           # 40 mini-batches: one graph chunk + eager rest
        env_gen = lambda: SyntheticFixedLengthEnv(E_, O_, Discrete(NA_), T_, dev, reward="uniform", seed=78, rank=rank,
                                                  num_agents=A_)
        sp = Box(-np.inf, np.inf, (O_,), np.float32)
        settings = {"p": (MATPolicy, sp, sp, Discrete(NA_), {})}
    ppo = PPO(env_gen, settings, device=dev, random_seed=SEED, normalize_obs=False, normalize_rewards=False,
              envs_per_proc=E_, ts_per_rollout=T_, batch_size=B_, epochs_per_iter=2, update_mode="fused")
    pol = ppo.policies["p"]
    ppo.rollout()
    ppo.train_on_rollout()
    fused = [f for f in getattr(ppo, "_fused", {}).values() if f is not None]
    from ppo_and_friends_amd.fused_update import FusedPolicyUpdate
    res = dict(peer_exchange=[getattr(f, "xchg", None) is not None for f in fused],
               split=[bool(getattr(f, "split", False)) for f in fused],
               tail=[f.tail_reason() == "" for f in fused if hasattr(f, "tail_reason")],
               pairs=[f.pairs_reason() == "" for f in fused if hasattr(f, "pairs_reason")],
               stats={k: float(v) for k, v in ppo.status_dict["p"].items()
                      if isinstance(v, (int, float)) and not isinstance(v, bool)})
    if kind == "guard":
        # replicas agree after a normal iteration; then rank 1 is corrupted behind the framework's back
        res["agree_before"] = ppo._guard_replicas()
        if rank == 1:
            pol.policy_params[7] += 1e-3
        res["agree_after_corruption"] = ppo._guard_replicas()        # restores rank 0's state, leaves the peer path
        res["w_healed"] = pol.policy_params.detach().cpu().clone()
        res["peer_exchange_after"] = [getattr(f, "xchg", None) is not None for f in fused]
        ppo.rollout()
        ppo.train_on_rollout()                                       # continues on the all-reduce path
        res["w"] = pol.policy_params.detach().cpu().clone()
    elif kind in ("wide", "wide_tail"):
        res["w"] = pol.policy_params.detach().cpu().clone()
    elif kind == "icm":
        res["w"] = pol.policy_params.detach().cpu().clone()
        res["w_icm"] = pol.icm_model.flat_params.detach().cpu().clone()
    else:
        res["w"] = pol.actor_critic.flat_params.detach().cpu().clone()
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["icm", "mat", "wide", "wide_tail"])
def test_peer_exchange_equals_allreduce_path(kind):
    runs = {}
    for mode in ("peer", "rccl"):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_rank_kind, args=(2, _free_port(), out, mode, kind), nprocs=2, join=True)
        runs[mode] = [out[r] for r in range(2)]
        r0, r1 = runs[mode]
        assert r0["peer_exchange"] and all(x == (mode == "peer") for x in r0["peer_exchange"]), r0["peer_exchange"]
        assert torch.equal(r0["w"], r1["w"]), f"{mode}: replicas identical"
        if kind == "icm":
            assert torch.equal(r0["w_icm"], r1["w_icm"])
        if kind in ("wide", "wide_tail"):
            # with K17 the 256-wide critic runs the split-wgrad chain on row pairs (+ an exchange launch, or the fused tail with the
            # exchange inside where both ranks' launches fit the one GPU); the all-reduce loop keeps slabs
            assert r0["split"] == [mode == "peer"] and r0["pairs"] == [mode == "peer"], (r0["split"], r0["pairs"])
            assert r0["tail"] == [mode == "peer" and kind == "wide_tail"], r0["tail"]
    a, b = runs["peer"][0], runs["rccl"][0]
    # (wide compares the split-wgrad chain with the slab chain: the same sums in another association)
    wtol = dict(rtol=1e-4, atol=2e-5) if kind in ("wide", "wide_tail") else dict(rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a["w"], b["w"], **wtol)
    if kind == "icm":
        torch.testing.assert_close(a["w_icm"], b["w_icm"], rtol=1e-5, atol=1e-6)
    for k in a["stats"]:
        np.testing.assert_allclose(a["stats"][k], b["stats"][k], err_msg=k, **wtol)


def test_diverged_replicas_are_detected_and_healed():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_kind, args=(2, _free_port(), out, "peer", "guard"), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    for r in (r0, r1):
        assert r["peer_exchange"] == [True] and r["agree_before"] is True
        assert r["agree_after_corruption"] is False and r["peer_exchange_after"] == [False]
    assert torch.equal(r0["w_healed"], r1["w_healed"])
    assert torch.equal(r0["w"], r1["w"]) and not torch.equal(r0["w"], r0["w_healed"])


def test_bench_contract_under_the_drivers_two_rank_launch():
    """
    The driver's N > 1 command -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- rehearsed with N = 2 on the one GPU of
    the box (PPOAF_BACKEND=gloo: RCCL refuses two ranks on a device; PPOAF_SHARE_DEVICE=1: both ranks on device 0).
    One JSON line from rank 0, whole-job value, weak scaling, the K17 exchange inside graph-replayed chains.
    """
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PPOAF_BACKEND="gloo", PPOAF_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PPOAF_BENCH_OTHER_CONFIGS_MULTI="1")     # the other shapes' N > 1 paths too (off by default at N > 1)
    env.pop("PPOAF_GRAD_EXCHANGE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--epochs", "1", "--no-cpu-baseline", "--no-saturating"]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["metric"] == "env_steps_per_sec" and out["value"] > 0 and out["higher_is_better"] is True
    cfg = out["config"]
    assert cfg["parallelism"] == "dp2" and cfg["global_env_steps_per_iteration"] == 2 * 4096 * 128
    assert cfg["multi_rank_path"] is True and cfg["hip_graphs"] is True and cfg["gradient_exchange"].startswith("K17")
    assert abs(out["value"] - 2 * 4096 * 128 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-3
    others = cfg["other_configs"]                    # the other BASELINE configs' shapes ride in the same line
    assert sorted(others) == ["C3", "C4", "C5"] and all(o["value"] > 0 for o in others.values())
    assert "roofline" in out and out["roofline"]["bound"] == "hbm"


@pytest.mark.parametrize("kind", ["ppo", "icm", "mat"])
def test_rccl_fallback_loops_agree(kind):
    """
    The N > 1 fallback when the K17 exchange is not available: fwd_bwd -> reduce -> RCCL all-reduce -> [norm +] Adam per
    mini-batch, for the PPO (K12), ICM (K14) and MAT (K15) updates.  Issued from C in one call per 256 mini-batches
    (`ppoaf_{ppo,icm,mat}_update_chain_allreduce`, the library's own RCCL communicator, id over torch.distributed) or
    from the Python loop (FusedPolicyUpdate.rccl_loop = "python"): the same kernels in the same order -- bitwise equal parameters and
    moments (the clip norms are fixed-order sums: no atomics).  One rank rehearsing the N > 1 path (a one-GPU box cannot
    host two RCCL ranks).
    """
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for loop in ("c", "python"):
        env = dict(os.environ, PPOAF_REHEARSE_MULTI_RANK="1", PPOAF_GRAD_EXCHANGE="rccl", PPOAF_TEST_RCCL_LOOP=loop,
                   PPOAF_FALLBACK_KIND=kind,
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PPOAF_BACKEND"):
            env.pop(k, None)
        p = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "rccl_fallback_run.py")], cwd=root, env=env,
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
        outs[loop] = json.loads(line[len("RESULT "):])
    assert outs["c"]["c_loop"] is True and outs["python"]["c_loop"] is False
    want_fused = {"ppo": ["FusedPolicyUpdate"], "icm": ["FusedIcmUpdate", "FusedPolicyUpdate"], "mat": ["FusedMatUpdate"]}[kind]
    assert outs["c"]["fused"] == outs["python"]["fused"] == want_fused
    per_update = 2 * 2 * (16 * 32 // 64)                       # iterations x epochs x mini-batches
    assert outs["c"]["steps"] == outs["python"]["steps"] == per_update * (2 if kind == "icm" else 1)
    assert outs["c"]["digest"] == outs["python"]["digest"], outs
