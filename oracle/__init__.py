"""
oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements (numpy / torch-CPU / plain C) of the LLNL/ppo_and_friends hot
path, each function citing the reference file:line it follows.  Only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import anything
from here, and only as the checker / reported baseline -- never as the product
path.  The product (ppo_and_friends_amd/) has no import of this package and
fails loudly if its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"): every module is PINNED against outputs of the unmodified reference
run in the build container (tests/golden/*.npz):
  * episode_info_oracle, running_stats_oracle, attention blocks, network init:   g1-g7 (make_golden.py)
  * ppo_loss_oracle, cpu_ppo_loop, icm_oracle, filter_oracle, lstm_oracle, mat_oracle, rollout_stats_oracle:
    g8-g13 (make_golden_update.py): unit vectors plus whole iterations of the reference's own PPO object over a
    table-driven env (the reference imports here with single-rank `mpi4py` and metadata-only `gymnasium` stand-ins,
    tests/golden/ref_import.py).  Still restated from text only: the ICM-for-MAT branches of mat_oracle.
"""
