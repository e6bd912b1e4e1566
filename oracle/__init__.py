"""
oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements (numpy / torch-CPU / plain C) of the LLNL/ppo_and_friends hot
path, each function citing the reference file:line it follows.  Only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import anything
from here, and only as the checker / reported baseline -- never as the product
path.  The product (ppo_and_friends_amd/) has no import of this package and
fails loudly if its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * episode_info_oracle, running_stats_oracle, attention_oracle: PINNED against
    outputs of the unmodified reference modules run in the build container
    (tests/golden/*.npz, generator tests/golden/make_golden.py).
  * ppo_loss_oracle, distributions_oracle, icm_oracle: the reference modules
    import `gymnasium`, which this image lacks, and the reference's own tests
    hold no numeric vectors for them -> restated from text; "parity unpinned"
    beyond torch's own primitives (MSELoss/HuberLoss/Categorical/Normal).
"""
