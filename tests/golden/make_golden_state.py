#!/usr/bin/env python3
"""
Checkpoint cross-read with the UNMODIFIED reference (build container only, recipe of ref_import.py).

  1. reference -> product.  The reference's own PPO object (ppo.py:126-2567) trains one small iteration over the
     table env (obs / reward normalisers + clippers and the value normaliser on) and `PPO.save()`s
     (ppo.py:2569-2618, policies/ppo_policy.py:1215-1247, utils/misc.py:130-146, environments/filter_wrappers.py:296-311,
     489-500).  The state directory it wrote is committed as data under tests/golden/ref_state/, next to
     g14_ref_state.npz = what it held in memory at that moment (weights, Adam moments / step / lr, normaliser
     statistics, status counters).  Tests: the product's PPO(load_state=True) reads that directory.
  2. product -> reference.  The product's PPO (host logic only: constructed on the CPU device, fixed seed, Adam moments
     and statistics filled with a recognisable pattern) `save()`s a state directory; the reference's PPO is then
     constructed with load_state=True on it (ppo.py:521-544, 625-634 -> PPOPolicy.load :1249-1300,
     RunningStatNormalizer.load_info misc.py:147-172, the wrappers' load_info) and g14_product_state_as_read_by_reference.npz records every tensor it ended up with.  Tests: the
     product re-creates that state directory from the same seed and the reference-side tensors must equal its own.

Usage:  python tests/golden/make_golden_state.py
"""
import os
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402
import make_golden_update as mgu  # noqa: E402  (table env written against the reference's env-wrapper API)

from state_common import E, T, O, NA, B, SEED, product_ppo, fill_product_state  # noqa: E402
STATE_DIR = os.path.join(HERE, "ref_state")


def reference_ppo(state_path, load_state=False):
    from ppo_and_friends.ppo import PPO
    from ppo_and_friends.networks.ppo_networks.feed_forward import FeedForwardNetwork
    from gymnasium.spaces import Discrete
    TableEnv = mgu.table_env_class()
    tables = mgu.make_tables(SEED, T, E, 1, O, "uniform", 0.05, obs_scale=2.0, obs_shift=0.5)
    counter = {"n": 0}
    pmap = lambda agent_id: "agent"

    def env_generator():
        e = max(counter["n"] - 1, 0)
        counter["n"] += 1
        return TableEnv(tables, e, Discrete(NA), 1, critic_view="local", policy_mapping_fn=pmap)

    template = TableEnv(tables, 0, Discrete(NA), 1, critic_view="local", policy_mapping_fn=pmap)
    settings = {"agent": (None, template.observation_space["agent0"], template.critic_observation_space["agent0"], Discrete(NA),
                          dict(ac_network=FeedForwardNetwork, actor_kw_args={"hidden_size": 32}, critic_kw_args={"hidden_size": 32}))}
    torch.manual_seed(SEED)
    np.random.seed(SEED)
    ppo = PPO(env_generator=env_generator, policy_settings=settings, policy_mapping_fn=pmap, device="cpu", random_seed=SEED,
              envs_per_proc=E, max_ts_per_ep=200, batch_size=B, ts_per_rollout=T, epochs_per_iter=2, normalize_obs=True,
              normalize_rewards=True, obs_clip=(-5.0, 5.0), reward_clip=(-5.0, 5.0), state_path=state_path, load_state=load_state,
              save_train_scores=False, save_avg_ep_len=False, save_running_time=False, save_bs_info=False, checkpoint_every=10 ** 9)
    return ppo, tables


def walk_env(env):
    while env is not None and hasattr(env, "env"):
        yield env
        env = env.env


def in_memory(ppo):
    """Everything the reference holds that a checkpoint is meant to carry, as plain arrays."""
    pol = ppo.policies["agent"]
    out = {}
    for tag, net, opt in (("actor", pol.actor, pol.actor_optim), ("critic", pol.critic, pol.critic_optim)):
        for k, v in net.state_dict().items():
            out[f"{tag}.{k}"] = v.detach().cpu().numpy().copy()
        sd = opt.state_dict()
        out[f"{tag}_optim.lr"] = np.array([sd["param_groups"][0]["lr"]], dtype=np.float64)
        for i, st in sd["state"].items():
            out[f"{tag}_optim.{i}.step"] = np.array([float(st["step"])])
            out[f"{tag}_optim.{i}.exp_avg"] = st["exp_avg"].numpy().copy()
            out[f"{tag}_optim.{i}.exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
    rs = ppo.value_normalizers["agent"].running_stats
    out["value_stats"] = np.array([rs.mean, rs.variance, rs.count], dtype=np.float64)
    for w in walk_env(ppo.env):
        cls = type(w).__name__
        for attr in ("actor_running_stats", "critic_running_stats", "running_stats"):
            val = getattr(w, attr, None)
            if isinstance(val, dict):
                for a, st in val.items():
                    out[f"{cls}.{attr}.{a}.mean"] = np.asarray(st.mean, dtype=np.float64)
                    out[f"{cls}.{attr}.{a}.variance"] = np.asarray(st.variance, dtype=np.float64)
                    out[f"{cls}.{attr}.{a}.count"] = np.array([st.count], dtype=np.float64)
    gs = ppo.status_dict["global status"]
    out["status.iteration_timesteps"] = np.array([gs["iteration"], gs["timesteps"]], dtype=np.float64)
    return out


def main():
    scratch = ref_import.make_scratch()
    try:
        # ---------------------------------------------------------------- 1. reference writes
        tmp = tempfile.mkdtemp(prefix="ppoaf_ref_state_")
        ppo, _ = reference_ppo(tmp)
        ppo.learn(E * T)                                   # one iteration (rollout + 2 epochs), then its own save()
        ppo.save()
        mem = in_memory(ppo)
        if os.path.isdir(STATE_DIR):
            shutil.rmtree(STATE_DIR)
        keep = []
        for r, _, fs in os.walk(tmp):
            for f in fs:
                rel = os.path.relpath(os.path.join(r, f), tmp)
                if rel.startswith("curves") or rel.endswith(".npy") or not (rel == "state_0.pickle" or "/latest/" in rel):
                    continue                               # the resume tag only (numbered / "best" checkpoints are copies)
                keep.append(rel)
                os.makedirs(os.path.dirname(os.path.join(STATE_DIR, rel)), exist_ok=True)
                shutil.copyfile(os.path.join(tmp, rel), os.path.join(STATE_DIR, rel))
        mem["files"] = np.array(sorted(keep))
        np.savez_compressed(os.path.join(HERE, "g14_ref_state.npz"), **mem)
        print("reference wrote:", sorted(keep))
        shutil.rmtree(tmp, ignore_errors=True)

        # ---------------------------------------------------------------- 2. product writes, reference reads
        ptmp = tempfile.mkdtemp(prefix="ppoaf_product_state_")
        prod = fill_product_state(product_ppo(ptmp))
        prod.save()
        # the reference's resume path is its constructor with load_state=True (ppo.py:521-544, 625-634: load_status,
        # load_env_info(<state>/env_info, tag), load_policy per policy) -- what `ppoaf train --load-state` runs.  (Its
        # PPO.load() method, ppo.py:2717-2730, hands load_env_info the state path instead of <state>/env_info and cannot
        # find the files PPO.save() wrote: quirk Q15, not reproduced.)
        rppo, _ = reference_ppo(ptmp, load_state=True)
        got = in_memory(rppo)
        np.savez_compressed(os.path.join(HERE, "g14_product_state_as_read_by_reference.npz"), **got)
        print("reference read the product's state dir:", len(got), "arrays")
        shutil.rmtree(ptmp, ignore_errors=True)
    finally:
        ref_import.drop_scratch(scratch)


if __name__ == "__main__":
    main()
