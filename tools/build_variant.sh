#!/bin/bash
# diagnostic build of the library with extra -D flags -> tools/libppoaf_hip_<name>.so (select it with PPOAF_LIB=<path>)
#   bash tools/build_variant.sh tailstamps -DPPOAF_TAIL_STAMPS
set -e
name=$1; shift
cd "$(dirname "$0")/../ppo_and_friends_amd/csrc"
obj=$(mktemp -d /tmp/ppoaf_variant_XXXX)
pids=()
for f in *.hip; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c "$f" -o "$obj/${f%.hip}.o" &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 "$obj"/*.o -o ../../tools/libppoaf_hip_$name.so
rm -rf "$obj"
echo tools/libppoaf_hip_$name.so
