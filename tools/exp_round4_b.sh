#!/bin/bash
# round 4 evidence collection on one MI355X box: kernel statistics of every config, the fused tail's stamps, GAE sweep +
# counters, update counters for C2 / C3
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
bash tools/profile_bench.sh r04 C2 C3 C4 C5 2>&1 | grep -v "^\[profile_bench\] working" | cut -c1-220
PPOAF_LIB=tools/libppoaf_hip_tailstamps.so python tools/tail_stamps.py 2>/dev/null | tee gpurun_out/prof/r04_tail_stamps_C2.txt
CRITIC_H=256 PPOAF_LIB=tools/libppoaf_hip_tailstamps.so python tools/tail_stamps.py 2>/dev/null | tee gpurun_out/prof/r04_tail_stamps_critic256.txt
VARIANTS=0,14,17,18,13 SKEWS=0,1088 python tools/gae_sweep.py 2>/dev/null | tee gpurun_out/prof/r04_gae_sweep.txt
bash tools/gae_pmc.sh r04 2>&1 | tail -8
bash tools/update_pmc.sh r04 C2 C3 2>&1 | tail -30
