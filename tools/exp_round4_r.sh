#!/bin/bash
# same-box A/B at C5: K15 / K16 fragment loads in even / odd chunk order, the diagnostic warm-up switch gone (tools/libppoaf_hip_oldmat.so = before)
# (the "before" library: check out the parent of the change, `bash tools/build_variant.sh <name>`, come back -- variant libraries are not kept in the tree)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_end_to_end.py -q -x -k "mat or MAT or c5" > gpurun_out/t_mat.log 2>&1 || { tail -30 gpurun_out/t_mat.log; exit 1; }
tail -2 gpurun_out/t_mat.log
run() {  # label, env..., -- bench args
    label=$1; shift
    envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout -k 10 200 python bench.py $* > gpurun_out/ab_$label.json 2> gpurun_out/ab_$label.err || { echo "$label FAILED"; tail -5 gpurun_out/ab_$label.err; return 1; }
    python - <<PY
import json
d = json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1])
print('$label', d['value'], d['ms_per_step'], (d.get('roofline_update') or {}).get('avg_launch_us'))
PY
}
B="--no-cpu-baseline --no-saturating --no-other-configs --steps 3 --warmup 1"
OLD=PPOAF_LIB=$PWD/tools/libppoaf_hip_oldmat.so
run C5_new PPOAF_X=0 -- --config C5 $B &&
run C5_old $OLD -- --config C5 $B &&
run C5_new2 PPOAF_X=0 -- --config C5 $B &&
run C5_old2 $OLD -- --config C5 $B
