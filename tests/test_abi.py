"""
CPU tests of the drop-in boundary: the C-ABI library builds, loads, and exports
every symbol include/ppoaf_hip.h declares, and the ctypes table matches the
header.  No compute call is made (there is no GPU here).
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ppoaf_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|const char\*|void\*)\s+(ppoaf_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


@pytest.fixture(scope="module")
def built_lib():
    from ppo_and_friends_amd.csrc import build
    build.build(verbose=False)
    from ppo_and_friends_amd import _lib
    return _lib


def test_header_declares_the_hot_path():
    d = _declared()
    for name in ("ppoaf_gae_rtg_tmajor", "ppoaf_gae_rtg_traj", "ppoaf_ppo_loss_fwd_bwd",
                 "ppoaf_minibatch_gather", "ppoaf_running_moments_integrate",
                 "ppoaf_categorical_sample", "ppoaf_clip_adam_step"):
        assert name in d


def test_library_exports_every_declared_symbol(built_lib):
    lib = built_lib.load()
    for name in _declared():
        assert hasattr(lib, name), f"libppoaf_hip.so lacks {name}"
    assert lib.ppoaf_abi_version() == 7


def test_ctypes_table_matches_header(built_lib):
    d = _declared()
    assert set(d) == set(built_lib.SIGNATURES), set(d) ^ set(built_lib.SIGNATURES)
    for name, n_args in d.items():
        assert len(built_lib.SIGNATURES[name][1]) == n_args, name


def test_cpu_tensor_is_refused_not_computed(built_lib):
    """The product path has no CPU fallback: host tensors raise before any launch."""
    import torch
    from ppo_and_friends_amd import kernels
    x = torch.zeros(4, 4)
    with pytest.raises(built_lib.PpoafError):
        kernels.gae_rtg_tmajor(x, x, torch.zeros(4), torch.zeros(4))


def test_missing_library_fails_loudly(built_lib, monkeypatch):
    monkeypatch.setattr(built_lib, "_lib", None)
    monkeypatch.setattr(built_lib, "LIB_PATH", "/nonexistent/libppoaf_hip.so")
    with pytest.raises(built_lib.PpoafError):
        built_lib.load()


def test_every_entry_point_sits_under_a_reference_citation():
    """
    include/ppoaf_hip.h is the drop-in boundary: every compute entry point is declared under a block comment that
    names the reference interface it replaces, as file:line (a `.py:` path with line numbers).  Helpers that have
    no reference counterpart (version / error string / events / diagnostics) are listed explicitly.
    """
    src = open(HEADER).read()
    no_counterpart = {"ppoaf_abi_version", "ppoaf_last_error", "ppoaf_device_cu_count", "ppoaf_event_create",
                      "ppoaf_event_destroy", "ppoaf_event_elapsed_ms"}
    cite = re.compile(r"[\w/]+\.py:\d+")
    last_cited_comment = -1
    pos = 0
    missing = []
    for m in re.finditer(r"/\*.*?\*/|\b(?:int|const char\*|void\*)\s+(ppoaf_\w+)\s*\(", src, flags=re.S):
        if m.group(1) is None:                                   # a comment
            if cite.search(m.group(0)):
                last_cited_comment = m.start()
            continue
        name = m.group(1)
        if name in no_counterpart:
            continue
        if last_cited_comment < 0:
            missing.append(name)
    assert not missing, missing
    # and the sections are specific: a few spot checks of the citation each section carries
    for needle in ("episode_info.py:223-293", "ppo.py:2325-2333", "mpi_utils.py:65-86", "filter_wrappers.py:155-268",
                   "mat_policy.py:441-519"):
        assert needle in src, needle


def test_bounded_wait_guard_reports_on_the_host():
    """
    The launches with in-kernel hand-overs (row pairs, fused tail) report a wait that ran out of its budget through an error
    word; the host side must notice -- never train on -- it: `_persistent_failure` names the reason and switches the form off
    (the epoch is then redone without it, tests/test_gpu_recovery.py), `_check_persistent` raises it.  Pure host logic:
    checked on fabricated control blocks.
    """
    import types
    import pytest
    import torch
    from ppo_and_friends_amd import _lib
    from ppo_and_friends_amd.fused_update import FusedPolicyUpdate

    def block(tail_error, pair_error):
        ns = types.SimpleNamespace(_tail_ctl=torch.tensor([5, 0, tail_error, 0] + [0] * 12, dtype=torch.int32), _tail_used=True,
                                   _split_space=torch.zeros(64, dtype=torch.uint8), _pair_region=16, _pairs_used=True,
                                   _graphs={"x": 1}, _args={"sig": 3, 256: object()})
        ns._split_space[16:20].view(torch.int32).fill_(pair_error)
        ns._persistent_failure = lambda: FusedPolicyUpdate._persistent_failure(ns)
        return ns

    ok = block(0, 0)
    assert FusedPolicyUpdate._persistent_failure(ok) == "" and ok._tail_used is False and ok._pairs_used is False
    bad = block(1, 0)
    assert "ran out of time" in FusedPolicyUpdate._persistent_failure(bad) and bad._tail_disabled and not bad._graphs
    bad = block(0, 1)
    assert "partner" in FusedPolicyUpdate._persistent_failure(bad) and bad._pairs_disabled and bad._args == {"sig": 3}
    with pytest.raises(_lib.PpoafError, match="ran out of time"):
        FusedPolicyUpdate._check_persistent(block(1, 0))


def test_no_memset_in_capturable_paths():
    """Source-level guard: hipMemsetAsync stays out of every entry point that may be captured into a hipGraph.  The
    allowed uses are never captured: the one-time clears of ppoaf_peer_exchange_create."""
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ppo_and_friends_amd", "csrc")
    allowed = {("peer_exchange.hip", "local"), ("peer_exchange.hip", "x->base")}
    found = set()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".hpp")):
            for m in re.finditer(r"hipMemset(?:Async)?\(\s*([^,]+),", open(os.path.join(root, f)).read()):
                found.add((f, m.group(1).strip()))
    assert found <= allowed, found - allowed


def test_row_pair_and_split_workspace_layout_on_the_host(built_lib):
    """
    The host-side arithmetic of round 4's row pairs (no launch, no GPU): the split workspace grows by exactly the record
    region (header + (2 depth - 3) exchanges x 32 tiles x 2 halves x 16 KB per 256-wide network), the error word sits where
    the panels of the full batch size end, and shapes the pairs do not cover report -1.
    """
    import ctypes as C
    lib = built_lib.load()

    def desc(in_dim, hidden, depth, out_dim, offset):
        size, pad4 = 0, lambda x: (x + 3) // 4 * 4
        for l in range(depth + 1):
            i = in_dim if l == 0 else hidden
            o = out_dim if l == depth else hidden
            size += pad4(i * o) + pad4(o)
        return built_lib.MlpDesc(in_dim=in_dim, hidden=hidden, depth=depth, out_dim=out_dim, activation=0, offset=offset,
                                 size=size, log_std_offset=-1)

    def args(critic_hidden, critic_depth, pairs, B=256, stride=256):
        a = built_lib.PpoUpdateArgs()
        a.actor = desc(18, 128, 3, 5, 0)
        a.critic = desc(54, critic_hidden, critic_depth, 1, a.actor.size)
        a.bucket_total = a.actor.size + a.critic.size
        for f in ("params", "grads", "exp_avg", "exp_avg_sq", "slabs", "step_counts", "lr", "norm_scratch", "obs", "critic_obs",
                  "raw_actions", "advantages", "old_log_probs", "rewards_to_go", "values", "perm", "cursor", "vn_mean", "vn_var",
                  "vn_count", "loss_partials", "totals"):
            setattr(a, f, 0x10000)                       # never dereferenced by the layout entry points
        a.head_kind, a.n_rows, a.B, a.batch_stride, a.n_ranks = 0, 4096, B, stride, 1
        a.beta1, a.beta2, a.adam_eps, a.grad_scale = 0.9, 0.999, 1e-5, 1.0
        a.row_pairs = pairs
        return a

    def ws_bytes(a):
        n = C.c_int64(0)
        assert lib.ppoaf_ppo_update_split_workspace_bytes(C.byref(a), C.byref(n)) == 0, lib.ppoaf_last_error()
        return n.value

    def err_off(a):
        n = C.c_int64(7)
        assert lib.ppoaf_ppo_update_row_pairs_error_offset(C.byref(a), C.byref(n)) == 0, lib.ppoaf_last_error()
        return n.value

    plain, paired = ws_bytes(args(256, 3, 0)), ws_bytes(args(256, 3, 1))
    assert paired - plain == 256 + 3 * 32 * 2 * 16384
    assert ws_bytes(args(256, 2, 1)) - ws_bytes(args(256, 2, 0)) == 256 + 1 * 32 * 2 * 16384
    assert err_off(args(256, 3, 1)) == plain and err_off(args(256, 3, 1, B=64, stride=256)) == plain      # a tail mini-batch: same place
    assert err_off(args(256, 3, 0)) == -1 and err_off(args(128, 3, 1)) == -1 and err_off(args(256, 5, 1)) == -1
