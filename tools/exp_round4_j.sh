#!/bin/bash
# round 4: whole-line forward weight sets in the row-tiled body -- parity of the K12 forms, C2 stamps, C2 / C3 / C4 / C5 steps
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_row_pairs.py -q -x > gpurun_out/t_fs.log 2>&1 || { tail -30 gpurun_out/t_fs.log; exit 1; }
tail -2 gpurun_out/t_fs.log
( PPOAF_LIB=tools/libppoaf_hip_stamps.so timeout -k 10 150 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/phase_stamps_lines.txt
cat gpurun_out/phase_stamps_lines.txt
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
run C2 PPOAF_X=0 -- --config C2 $B &&
run C4 PPOAF_X=0 -- --config C4 $B &&
run C3 PPOAF_X=0 -- --config C3 $B &&
run C5 PPOAF_X=0 -- --config C5 $B &&
run C2_again PPOAF_X=0 -- --config C2 $B
