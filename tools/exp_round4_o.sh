#!/bin/bash
# round 4: K14 as one launch per mini-batch -- parity (bitwise the three launches), the ICM tests, then C3
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_full_size.py -q -x -k "icm_single_launch" > gpurun_out/t_icmf.log 2>&1 || { tail -40 gpurun_out/t_icmf.log; exit 1; }
tail -2 gpurun_out/t_icmf.log
timeout -k 10 700 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_full_size.py tests/test_gpu_reference_golden.py -q -x -k "icm or ICM" > gpurun_out/t_icm.log 2>&1 || { tail -40 gpurun_out/t_icm.log; exit 1; }
tail -2 gpurun_out/t_icm.log
run() {  # label, fuse (1/0), -- bench args
    label=$1; fuse=$2; shift 3
    timeout -k 10 200 python -c "
import sys, runpy
from ppo_and_friends_amd import fused_update
fused_update.FusedIcmUpdate.fuse_kernels = bool($fuse)
sys.argv = ['bench.py'] + '$*'.split()
runpy.run_path('bench.py', run_name='__main__')
" > gpurun_out/ab_$label.json 2> gpurun_out/ab_$label.err || { echo "$label FAILED"; tail -5 gpurun_out/ab_$label.err; return 1; }
    python - <<PY
import json
d = json.loads(open('gpurun_out/ab_$label.json').read().strip().splitlines()[-1])
print('$label', d['value'], d['ms_per_step'])
PY
}
B="--no-cpu-baseline --no-saturating --no-other-configs --steps 3 --warmup 1"
run C3_one_launch 1 -- --config C3 $B &&
run C3_three_launches 0 -- --config C3 $B &&
run C3_one_launch2 1 -- --config C3 $B &&
run C3_three_launches2 0 -- --config C3 $B
