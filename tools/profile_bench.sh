#!/bin/bash
# rocprofv3 kernel statistics of bench.py for the given configs -> gpurun_out/prof/<tag>_<config>_kernel_stats.csv
# (only the small summaries leave the box; a heartbeat keeps the run from looking hung while the trace is post-processed)
#   tools/profile_bench.sh r02 C2 C4
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
( while true; do echo "[profile_bench] working $(date +%T)"; sleep 45; done ) &
HB=$!
for CFG in "$@"; do
  rm -rf /tmp/prof_$CFG
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$CFG -o $CFG -- \
      python3 $R/bench.py --config $CFG --steps 1 --warmup 1 --epochs 3 --no-cpu-baseline --no-saturating --no-other-configs \
      > $R/gpurun_out/prof/${TAG}_${CFG}_under_rocprof.json 2> /tmp/prof_$CFG.err
  echo "[profile_bench] $CFG rc=$?"
  f=$(find /tmp/prof_$CFG -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/prof/${TAG}_${CFG}_kernel_stats.csv && head -8 "$f"
  rm -rf /tmp/prof_$CFG
done
kill $HB
