#!/bin/bash
# same-box A/B: the fused tail's input panels as whole lines against operand-order loads (tools/libppoaf_hip_oldtail.so)
# (the "before" library: check out the parent of the change, `bash tools/build_variant.sh <name>`, come back -- variant libraries are not kept in the tree)
set -o pipefail
mkdir -p gpurun_out
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
OLD=PPOAF_LIB=$PWD/tools/libppoaf_hip_oldtail.so
run C2_lines PPOAF_X=0 -- --config C2 $B &&
run C2_old $OLD -- --config C2 $B &&
run C2_lines2 PPOAF_X=0 -- --config C2 $B &&
run C2_old2 $OLD -- --config C2 $B &&
run C4_lines PPOAF_X=0 -- --config C4 $B &&
run C4_old $OLD -- --config C4 $B
