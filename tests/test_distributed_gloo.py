"""
CPU tests of the N>1 path: world_size-2 `gloo` process groups exercising the host-side
collective layer that replaces the reference's mpi4py calls (utils/mpi_utils.py here;
SURVEY.md §2.2(ii)): rank-0 parameter broadcast, flat-bucket gradient SUM + 1/R, the
(n, mean, M2) moment-record exchange that replaces the raw-data allgather of
utils/stats.py:47-50, and the packed scalar reductions.  The HIP kernels are not
involved (no GPU here); the arithmetic the records feed is checked with the oracle.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.running_stats_oracle import RunningMeanStd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _chan_merge(records):
    """Numpy restatement of the record merge done by running_moments_integrate_kernel / K12."""
    n, m, M2 = 0.0, 0.0, 0.0
    for nb, mb, qb in records:
        if nb <= 0:
            continue
        d = mb - m
        nn = n + nb
        m += d * (nb / nn)
        M2 += qb + d * d * n * nb / nn
        n = nn
    return n, m, M2


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ppo_and_friends_amd.utils import mpi_utils
    r, w, _ = mpi_utils.init_process_group_from_env(backend="gloo")
    assert (r, w) == (rank, world) and mpi_utils.get_num_procs() == world
    res = {}
    # 1. parameter broadcast: every rank ends with rank 0's bucket (mpi_utils.py:50-63)
    flat = torch.full((1000,), float(rank + 1))
    mpi_utils.broadcast_flat(flat)
    res["bcast"] = flat.clone()
    # 2. gradient exchange: SUM all-reduce of the flat bucket, 1/R applied by the consumer
    g = torch.arange(8, dtype=torch.float32) * (rank + 1)
    mpi_utils.allreduce_sum_(g)
    res["grad_avg"] = g / world
    # mpi_avg on scalars and tensors (mpi_utils.py:65-86)
    res["avg_scalar"] = mpi_utils.mpi_avg(float(rank))
    res["avg_tensor"] = mpi_utils.mpi_avg(torch.tensor([1.0, 2.0]) * (rank + 1))
    # 3. moment records: rank-local (n, mean, M2) per mini-batch, all-gathered once per epoch
    rng = np.random.default_rng(100 + rank)
    nb, B = 3, 64
    data = (rng.standard_normal((nb, B)) * (rank + 1) + rank).astype(np.float32)
    rec = np.stack([np.full(nb, float(B)), data.mean(1, dtype=np.float64),
                    ((data - data.mean(1, keepdims=True, dtype=np.float64)) ** 2).sum(1)], axis=1)
    allr = mpi_utils.allgather_records(torch.tensor(rec.reshape(-1))).view(world, nb, 3)
    res["records"] = allr.permute(1, 0, 2).contiguous()       # [nb, R, 3] as K12 reads them
    res["data"] = torch.tensor(data)
    # 4. packed scalar reductions (ppo.py:2471-2475, 1991-2094)
    res["sum"] = mpi_utils.allreduce_scalars([1.0, rank, 2.5], "sum")
    res["max"] = mpi_utils.allreduce_scalars([rank, -rank], "max")
    res["min"] = mpi_utils.allreduce_scalars([rank, -rank], "min")
    # 5. the rollout statistics block across ranks (ppo.py:1978-2099: SUM / MAX / MIN allreduces)
    from ppo_and_friends_amd.utils.rollout_stats import rollout_statistics
    rs = np.random.default_rng(7 + rank)
    T, E = 12, 5
    term = rs.uniform(0, 1, (T, E)) < 0.15
    boot = np.zeros((T, E), bool); boot[-1] = ~term[-1]
    arr = dict(r=rs.uniform(-1, 1, (T, E)), nat=rs.uniform(-2, 2, (T, E)), nr=rs.uniform(-3, 3, (T, E)),
               term=term, boot=boot, omin=rs.uniform(-5, -1, T), omax=rs.uniform(1, 5, T))
    tt = torch.as_tensor
    res["stats"] = rollout_statistics(tt(arr["r"]), tt(arr["nat"]), tt(term), tt(boot), tt(arr["nr"]), 1,
                                      (tt(arr["r"].min()), tt(arr["r"].max())), (tt(arr["nat"].min()), tt(arr["nat"].max())),
                                      (tt(arr["omin"].min()), tt(arr["omax"].max())), T)
    res["stats_in"] = arr
    # 6. the environment-filter record exchange (one all-gather per env step instead of raw data)
    rec = torch.arange(7, dtype=torch.float64) + 10 * rank
    res["filter_records"] = mpi_utils.allgather_records(rec)
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def two_ranks():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    return [out[r] for r in range(world)]


def test_broadcast_and_gradient_average(two_ranks):
    r0, r1 = two_ranks
    assert torch.equal(r0["bcast"], torch.full((1000,), 1.0)) and torch.equal(r1["bcast"], r0["bcast"])
    exp = torch.arange(8, dtype=torch.float32) * 1.5          # (1x + 2x) / 2
    assert torch.equal(r0["grad_avg"], exp) and torch.equal(r1["grad_avg"], exp)
    assert r0["avg_scalar"] == 0.5 and r1["avg_scalar"] == 0.5
    assert torch.equal(r0["avg_tensor"], torch.tensor([1.5, 3.0]))


def test_moment_records_equal_raw_data_allgather(two_ranks):
    """Merging the gathered records == the reference's allgather of raw data + np.mean / np.var."""
    r0, r1 = two_ranks
    assert torch.equal(r0["records"], r1["records"])           # every rank integrates the same records
    recs = r0["records"].numpy()
    ref = RunningMeanStd()
    mine_mean, mine_var, mine_count = np.float32(0.0), np.float32(1.0), 1e-4
    for k in range(recs.shape[0]):
        parts = [r0["data"][k].numpy(), r1["data"][k].numpy()]
        ref.update(None, gathered=parts)                       # stats.py:47-54 on the concatenation
        n, m, M2 = _chan_merge(recs[k])
        tmp = RunningMeanStd()
        tmp.mean, tmp.variance, tmp.count = mine_mean, mine_var, mine_count
        tmp.integrate(np.float32(m), np.float32(M2 / n), n)    # stats.py:73-94
        mine_mean, mine_var, mine_count = tmp.mean, tmp.variance, tmp.count
    np.testing.assert_allclose(mine_mean, ref.mean, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(mine_var, ref.variance, rtol=2e-6, atol=1e-7)
    assert mine_count == ref.count


def test_packed_scalar_reductions(two_ranks):
    r0, r1 = two_ranks
    assert r0["sum"] == [2.0, 1.0, 5.0] == r1["sum"]
    assert r0["max"] == [1.0, 0.0] and r0["min"] == [0.0, -1.0]


def test_rollout_statistics_across_ranks(two_ranks):
    """Per-rank literal loops (oracle) combined the way the reference's allreduces do == the 2-rank result."""
    from oracle.rollout_stats_oracle import rollout_statistics_loop
    r0, r1 = two_ranks
    per = []
    for r in (r0, r1):
        a = r["stats_in"]
        per.append(rollout_statistics_loop(a["r"], a["nat"], np.zeros_like(a["r"]), a["omin"], a["omax"], a["term"],
                                           a["boot"], a["nr"]))
    eps = per[0]["total episodes"] + per[1]["total episodes"]
    got = r0["stats"]
    assert got == r1["stats"]
    np.testing.assert_allclose(got["total episodes"], eps, rtol=1e-12)
    np.testing.assert_allclose(got["score avg"], (per[0]["score avg"] * per[0]["total episodes"] +
                                                  per[1]["score avg"] * per[1]["total episodes"]) / eps, rtol=1e-12)
    np.testing.assert_allclose(got["top score"], max(p["top score"] for p in per), rtol=1e-12)
    np.testing.assert_allclose(got["reward range"], (min(p["reward range"][0] for p in per),
                                                     max(p["reward range"][1] for p in per)), rtol=1e-12)
    np.testing.assert_allclose(got["longest episode"], max(p["longest episode"] for p in per))
    np.testing.assert_allclose(got["shortest episode"], min(p["shortest episode"] for p in per))
    np.testing.assert_allclose(got["average episode"], (per[0]["average episode"] + per[1]["average episode"]) / 2)
    assert torch.equal(r0["filter_records"], torch.stack([torch.arange(7, dtype=torch.float64),
                                                          torch.arange(7, dtype=torch.float64) + 10]))


def _checksum_rank(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from ppo_and_friends_amd.utils import mpi_utils
    mpi_utils.init_process_group_from_env(backend="gloo")
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(1000, generator=g), torch.randn(77, generator=g)
    same = mpi_utils.replicas_agree([a, b])
    if rank == 1:
        a[400] = torch.nextafter(a[400], torch.tensor(10.0))        # one ulp on one rank
    differ = mpi_utils.replicas_agree([a, b])
    swapped = mpi_utils.bucket_checksum([b, a]).item() != mpi_utils.bucket_checksum([a, b]).item()
    out[rank] = (same, differ, swapped)
    dist.barrier()
    dist.destroy_process_group()


def test_replica_checksum_sees_one_ulp():
    """utils.mpi_utils.replicas_agree: the guard behind the K17 peer exchange (ppo.PPO._guard_replicas)."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_checksum_rank, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in range(2):
        assert out[r] == (True, False, True)


def _ddppo_rank(rank, world, port, name):
    """One gloo process = one rank of the reference's R = 2 run, replayed through the CPU port with cpu_ddppo.GlooComm."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.dirname(here)]
    import torch.distributed as dist
    from oracle import cpu_ddppo
    import test_oracle_update_golden as replay
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = np.load(os.path.join(here, "golden", name + ".npz"), allow_pickle=False)
    replay.replay_fixture(replay.RankView(g, rank), replay.RANK_SCENARIOS[name], cpu_ddppo.GlooComm())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["g12_c2_r2", "g12_c2_icm_r2_klstop"])
def test_cpu_ddppo_over_gloo_reproduces_two_ranks_of_the_reference(name):
    """
    oracle/cpu_ddppo.py -- the R-process CPU baseline bench.py times (`cpu_baseline_mpi`) -- on the collectives it really
    uses (torch.distributed / gloo between processes), against fixtures recorded from TWO ranks of the unmodified
    reference: per-tensor gradient averaging (utils/mpi_utils.py:89-111), the value normaliser's all-gather of raw data
    (utils/stats.py:47-50), all-reduced epoch totals and the KL early stop both ranks take together (ppo.py:2468-2475,
    2221-2232).  Every check is made inside the rank processes (test_oracle_update_golden.replay_fixture).
    """
    import torch.multiprocessing as mp
    mp.spawn(_ddppo_rank, args=(2, _free_port(), name), nprocs=2, join=True)
