#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
bash tools/profile_bench.sh r04 C3 2>&1 | grep -v "^\[profile_bench\] working" | cut -c1-160
PPOAF_REHEARSE_MULTI_RANK=1 python bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline --no-saturating --no-other-configs > gpurun_out/prof/r04_C3_rehearse_multi_rank.json 2>/dev/null
python - <<PY
import json
d = json.loads(open('gpurun_out/prof/r04_C3_rehearse_multi_rank.json').read().strip().splitlines()[-1])
print('rehearsal C3', d['value'], d['ms_per_step'], d['config'].get('gradient_exchange'))
PY
