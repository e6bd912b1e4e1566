"""
-m gpu: K17, the peer-mapped gradient exchange (csrc/peer_exchange.hip), through the C ABI.

Three processes share the one GPU of the box (IPC mappings between processes work on one device exactly
as across xGMI peers; the control plane is gloo).  Every rank can regenerate every other rank's input
from the seed, so the expected result -- the sum in rank order, which is what the reference's
mpi_avg_gradients (utils/mpi_utils.py:65-86) computes up to summation order -- is formed locally and
compared bit for bit.  Also: in-place use, the two clip norms, hipGraph capture + replay with changing
data, a long back-to-back run (slot reuse), and a missing peer (bounded wait, error word, no hang).
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_FLOATS = 34_820            # about one network of the C2 bucket (67 720 floats in all); any multiple of 4
WORLD = 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _src(rank, step, dev):
    g = torch.Generator().manual_seed(100_003 * step + 17 * rank + 1)
    return torch.randn(N_FLOATS, generator=g).to(dev)


def _want(step, world, dev):
    acc = torch.zeros(N_FLOATS, device=dev)
    for r in range(world):
        acc = acc + _src(r, step, dev)
    return acc


def _rank(rank, world, port, out, memory):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PPOAF_GRAD_EXCHANGE="peer",
                      PPOAF_PEER_MEMORY=memory)
    import torch.distributed as dist
    from ppo_and_friends_amd.utils import mpi_utils, peer_exchange
    mpi_utils.init_process_group_from_env(backend="gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    res = {}
    x, why = peer_exchange.open_exchange(N_FLOATS, dev)        # includes the start-up self-test
    res["opened"] = x is not None
    res["why"] = why
    done0 = x.status()[0]
    split = 8192
    norms = torch.zeros(2, dtype=torch.float64, device=dev)

    # 1. out of place + norms
    step = 1
    got = torch.empty(N_FLOATS, device=dev)
    x.allreduce(_src(rank, step, dev), got, split_floats=split, norm_scale=1.0 / world, norm_out=norms)
    torch.cuda.synchronize()
    want = _want(step, world, dev)
    res["sum_exact"] = torch.equal(got, want)
    w = (want / world).double()
    res["norms_close"] = bool(torch.allclose(norms, torch.stack([(w[:split] ** 2).sum(), (w[split:] ** 2).sum()]), rtol=1e-6))
    res["norms"] = norms.cpu().tolist()

    # 2. long back-to-back run, in place, no host synchronisation in between (slot reuse, flag ordering)
    ok = True
    bufs = [_src(rank, 10 + k, dev) for k in range(64)]
    for b in bufs:
        x.allreduce(b, b)
    torch.cuda.synchronize()
    for k, b in enumerate(bufs):
        ok = ok and torch.equal(b, _want(10 + k, world, dev))
    res["back_to_back_exact"] = ok

    # 3. hipGraph: capture 8 exchanges of a staging buffer, replay with new data each time
    stage = [torch.zeros(N_FLOATS, device=dev) for _ in range(8)]
    outs = [torch.zeros(N_FLOATS, device=dev) for _ in range(8)]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for a, b in zip(stage, outs):
            x.allreduce(a, b)                                  # warm-up (real exchanges of zeros)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for a, b in zip(stage, outs):
            x.allreduce(a, b)
    ok = True
    for rep in range(3):
        for k, a in enumerate(stage):
            a.copy_(_src(rank, 1000 + 10 * rep + k, dev))
        g.replay()
        torch.cuda.synchronize()
        for k, b in enumerate(outs):
            ok = ok and torch.equal(b, _want(1000 + 10 * rep + k, world, dev))
    res["graph_replay_exact"] = ok
    done, timed_out, kind, n = x.status()
    res["count"] = done - done0
    res["timed_out"] = timed_out
    res["memory_kind"] = kind
    # every rank's results identical by construction; cross-check one checksum through the control plane
    res["checksum"] = float(outs[-1].double().sum().item())

    # 3b. other bucket sizes: one float4, a partial last group, and more than 256 groups' worth (strided groups)
    ok = True
    for n in (4, 1028, 300_000):
        y, why_n = peer_exchange.open_exchange(n, dev)
        g2 = torch.Generator().manual_seed(n + rank)
        mine = torch.randn(n, generator=g2).to(dev)
        want_n = torch.zeros(n, device=dev)
        for r in range(world):
            want_n = want_n + torch.randn(n, generator=torch.Generator().manual_seed(n + r)).to(dev)
        for _ in range(3):                                    # both slots and their reuse
            got_n = torch.empty(n, device=dev)
            y.allreduce(mine, got_n)
            torch.cuda.synchronize()
            ok = ok and torch.equal(got_n, want_n)
        y.close()
        dist.barrier()
    res["other_sizes_exact"] = ok

    # 4. a peer that never shows up: the wait is bounded and reported
    dist.barrier()
    if rank != world - 1:
        junk = torch.zeros(N_FLOATS, device=dev)
        x.allreduce(junk, junk, wait_seconds=0.25)
        torch.cuda.synchronize()
        res["missing_peer_reported"] = x.status()[1] != 0
        import time
        t0 = time.time()                                      # a broken exchange does not spend the budget again
        x.allreduce(junk, junk, wait_seconds=30.0)
        torch.cuda.synchronize()
        res["later_waits_fail_fast"] = (time.time() - t0) < 5.0
        try:
            x.check()
            res["check_raises"] = False
        except Exception:
            res["check_raises"] = True
    dist.barrier()
    out[rank] = res
    x.close()
    dist.destroy_process_group()


# the three kinds of exchange memory open_exchange tries in turn on a node (uncached first)
MEMORY = {"uncached": 1, "fine-grained": 2, "coarse-grained": 3}


@pytest.fixture(scope="module", params=list(MEMORY))
def run3(request):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank, args=(WORLD, _free_port(), out, request.param), nprocs=WORLD, join=True)
    res = [out[r] for r in range(WORLD)]
    for r in res:
        assert r["memory_kind"] == MEMORY[request.param] and request.param in r["why"], (r["memory_kind"], r["why"])
    return res


def test_exchange_opens_and_self_test_passes(run3):
    for r in run3:
        assert r["opened"], r["why"]
        assert r["memory_kind"] in (1, 2, 3)


def test_sum_is_rank_ordered_and_bit_exact(run3):
    for r in run3:
        assert r["sum_exact"] and r["back_to_back_exact"]
        assert r["norms_close"], r["norms"]
    assert run3[0]["norms"] == run3[1]["norms"] == run3[2]["norms"], "fixed-order norm: identical on every rank"


def test_other_bucket_sizes(run3):
    for r in run3:
        assert r["other_sizes_exact"]


def test_graph_replay(run3):
    for r in run3:
        assert r["graph_replay_exact"]
        assert r["count"] == 1 + 64 + 8 + 3 * 8 and r["timed_out"] == 0
    assert run3[0]["checksum"] == run3[1]["checksum"] == run3[2]["checksum"]


def test_missing_peer_is_bounded_and_reported(run3):
    for r in run3[:-1]:
        assert r["missing_peer_reported"] and r["check_raises"] and r["later_waits_fail_fast"]


def test_four_ranks_group_and_flag_indexing():
    """
    R = 4 on the one GPU of the box: the per-(group, rank) flag words and the alternating slots at a rank count the
    3-process fixture does not reach.  (The pool's process guard admits at most 6 processes with the GPU open, so 4
    ranks + the test runner is the largest world that can be rehearsed here; R = 8 is the driver's 8-GPU run.)
    """
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank, args=(4, _free_port(), out, "uncached"), nprocs=4, join=True)
    res = [out[r] for r in range(4)]
    for r in res:
        assert r["opened"], r["why"]
        assert r["sum_exact"] and r["back_to_back_exact"] and r["other_sizes_exact"] and r["graph_replay_exact"]
        assert r["norms_close"], r["norms"]
    assert len({str(r["norms"]) for r in res}) == 1 and len({r["checksum"] for r in res}) == 1
    for r in res[:-1]:
        assert r["missing_peer_reported"] and r["later_waits_fail_fast"]


def test_single_rank_exchange_is_the_identity():
    from ppo_and_friends_amd.utils.peer_exchange import PeerExchange
    dev = torch.device("cuda", 0)
    x = PeerExchange(N_FLOATS, dev)
    src = torch.randn(N_FLOATS, device=dev)
    dst = torch.empty_like(src)
    norms = torch.zeros(2, dtype=torch.float64, device=dev)
    x.allreduce(src, dst, split_floats=4096, norm_scale=1.0, norm_out=norms)
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    torch.testing.assert_close(norms, torch.stack([(src[:4096].double() ** 2).sum(), (src[4096:].double() ** 2).sum()]), rtol=1e-12, atol=0)
    assert x.status()[:2] == (1, 0)
    x.close()
