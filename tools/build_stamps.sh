#!/bin/bash
# diagnostic build with in-kernel stamps -> tools/libppoaf_hip_stamps.so
set -e
cd /root/repo/ppo_and_friends_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPPOAF_STAMPS -shared *.hip -o /root/repo/tools/libppoaf_hip_stamps.so
