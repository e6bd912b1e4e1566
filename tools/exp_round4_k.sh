#!/bin/bash
# round 4: K14 with every weight set requested a phase ahead (forward sets as whole lines) -- parity, then C3
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_full_size.py tests/test_gpu_reference_golden.py -q -x -k "icm or ICM" > gpurun_out/t_icm.log 2>&1 || { tail -40 gpurun_out/t_icm.log; exit 1; }
tail -2 gpurun_out/t_icm.log
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
run C3 PPOAF_X=0 -- --config C3 $B &&
run C3_again PPOAF_X=0 -- --config C3 $B
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python3 $R/bench.py --no-cpu-baseline --no-saturating --no-other-configs --steps 1 --warmup 1 --config C3 > /tmp/prof_c3.log 2>&1 || { tail -5 /tmp/prof_c3.log; exit 1; }
f=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out/prof; cp $f $R/gpurun_out/prof/r04k_C3_kernel_stats.csv
head -9 $f | cut -d, -f1-4 | cut -c1-160
