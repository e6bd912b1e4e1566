#!/bin/bash
# diagnostic build of the library with extra -D flags -> tools/libppoaf_hip_<name>.so (select it with PPOAF_LIB=<path>)
#   bash tools/build_variant.sh nt -DPPOAF_XCU_LOADS_NT
set -e
name=$1; shift
cd "$(dirname "$0")/../ppo_and_friends_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -shared *.hip -o ../../tools/libppoaf_hip_$name.so
echo tools/libppoaf_hip_$name.so
