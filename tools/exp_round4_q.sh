#!/bin/bash
# same-box A/B: arrival-order requests + input rows requested up front in the row-tiled body (tools/libppoaf_hip_oldrt.so = before)
# (the "before" library: check out the parent of the change, `bash tools/build_variant.sh <name>`, come back -- variant libraries are not kept in the tree)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_row_pairs.py -q -x > gpurun_out/t_fs.log 2>&1 || { tail -30 gpurun_out/t_fs.log; exit 1; }
tail -2 gpurun_out/t_fs.log
( PPOAF_LIB=tools/libppoaf_hip_stamps.so timeout -k 10 150 python tools/phase_stamps.py 2>&1 | grep -v "amdgpu.ids\|^net 1\|        0$" ) | head -12
run() {  # label, env..., -- bench args
    label=$1; shift
    envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout -k 10 200 python bench.py $* > gpurun_out/ab_$label.json 2> gpurun_out/ab_$label.err || { echo "$label FAILED"; tail -5 gpurun_out/ab_$label.err; return 1; }
    python - <<PY
import json
d = json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1])
print('$label', d['value'], d['ms_per_step'])
PY
}
B="--no-cpu-baseline --no-saturating --no-other-configs --steps 3 --warmup 1"
OLD=PPOAF_LIB=$PWD/tools/libppoaf_hip_oldrt.so
run C2_new PPOAF_X=0 -- --config C2 $B &&
run C2_old $OLD -- --config C2 $B &&
run C2_new2 PPOAF_X=0 -- --config C2 $B &&
run C2_old2 $OLD -- --config C2 $B &&
run C4_new PPOAF_X=0 -- --config C4 $B &&
run C4_old $OLD -- --config C4 $B &&
run C3_new PPOAF_X=0 -- --config C3 $B &&
run C3_old $OLD -- --config C3 $B
