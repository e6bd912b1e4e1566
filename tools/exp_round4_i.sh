#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_full_size.py -q -x -k "row_pairs" > gpurun_out/t_pairs.log 2>&1 || { tail -30 gpurun_out/t_pairs.log; exit 1; }
tail -2 gpurun_out/t_pairs.log
( PPOAF_LIB=tools/libppoaf_hip_stamps4.so timeout -k 10 150 python tools/pair_stamps.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/pair_stamps_v3.txt
cat gpurun_out/pair_stamps_v3.txt
run() {  # label, env..., -- bench args
    label=$1; shift
    envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout -k 10 200 python bench.py $* > gpurun_out/ab_$label.json 2> gpurun_out/ab_$label.err || { echo "$label FAILED"; tail -5 gpurun_out/ab_$label.err; return 1; }
    python - <<PY
import json
d = json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1])
print('$label', d['value'], d['ms_per_step'], d['config'].get('update_kernel'), (d.get('roofline_update') or {}).get('avg_launch_us'))
PY
}
B="--no-cpu-baseline --no-saturating --no-other-configs --steps 3 --warmup 1"
run C4_chain PPOAF_WS=0 -- --config C4 $B &&
run C3 PPOAF_X=0 -- --config C3 $B &&
run C2 PPOAF_X=0 -- --config C2 $B
