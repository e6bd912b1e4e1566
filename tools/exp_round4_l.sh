#!/bin/bash
# round 4, final evidence on one MI355X box: kernel statistics of every config, stamps of the row-tiled / row-pair kernels,
# GAE and update counters, one-rank rehearsals of the N > 1 paths, the default bench line
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
bash tools/profile_bench.sh r04 C2 C3 C4 C5 2>&1 | grep -v "^\[profile_bench\] working" | cut -c1-200
PPOAF_LIB=tools/libppoaf_hip_stamps.so python tools/phase_stamps.py 2>/dev/null | grep -v "^net 1\|        0$" | tee gpurun_out/prof/r04_phase_stamps_C2.txt
PPOAF_LIB=tools/libppoaf_hip_stamps4.so python tools/pair_stamps.py 2>/dev/null | tee gpurun_out/prof/r04_pair_stamps.txt
PAIRS=0 PPOAF_LIB=tools/libppoaf_hip_stamps4.so python tools/pair_stamps.py 2>/dev/null | tee -a gpurun_out/prof/r04_pair_stamps.txt
bash tools/gae_pmc.sh r04 2>&1 | tail -8
bash tools/update_pmc.sh r04 C2 C3 C4 2>&1 | tail -40
for C in C2 C3 C4 C5; do
  PPOAF_REHEARSE_MULTI_RANK=1 python bench.py --config $C --steps 2 --warmup 1 --no-cpu-baseline --no-saturating --no-other-configs > gpurun_out/prof/r04_${C}_rehearse_multi_rank.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open('gpurun_out/prof/r04_${C}_rehearse_multi_rank.json').read().strip().splitlines()[-1])
print('rehearsal $C', d['value'], d['ms_per_step'], d['config'].get('gradient_exchange'))
PY
done
python bench.py > gpurun_out/prof/r04_bench_default.json 2>/dev/null; cut -c1-600 gpurun_out/prof/r04_bench_default.json
