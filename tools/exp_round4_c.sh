#!/bin/bash
# round 4: row pairs for 256-wide critics -- parity first, then C3 / C4 A/B on one box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_full_size.py -q -x -k "row_pairs" > gpurun_out/t_pairs.log 2>&1 || { tail -30 gpurun_out/t_pairs.log; exit 1; }
tail -2 gpurun_out/t_pairs.log
run() {  # label, pairs (1/0), env..., -- bench args
    label=$1; pairs=$2; shift 2
    envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout -k 10 200 python -c "
import sys, runpy
from ppo_and_friends_amd import fused_update
fused_update.FusedPolicyUpdate.row_pairs = bool($pairs)
sys.argv = ['bench.py'] + '$*'.split()
runpy.run_path('bench.py', run_name='__main__')
" > gpurun_out/ab_$label.json 2> gpurun_out/ab_$label.err || { echo "$label FAILED"; tail -5 gpurun_out/ab_$label.err; return 1; }
    python - <<PY
import json
d = json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1])
print('$label', d['value'], d['ms_per_step'], d['config'].get('update_kernel'), (d.get('roofline_update') or {}).get('avg_launch_us'))
PY
}
B="--no-cpu-baseline --no-saturating --no-other-configs --steps 2 --warmup 1"
run C3_pairs 1 PPOAF_X=0 -- --config C3 $B &&
run C3_tiles 0 PPOAF_X=0 -- --config C3 $B &&
run C4_chain_pairs 1 PPOAF_WS=0 -- --config C4 $B &&
run C4_chain_tiles 0 PPOAF_WS=0 -- --config C4 $B &&
run C4_ws 1 PPOAF_X=0 -- --config C4 $B &&
run C3_pairs_again 1 PPOAF_X=0 -- --config C3 $B
