set -u
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-saturating --no-other-configs"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], (d.get("roofline_update") or {}).get("kernel"), (d.get("roofline_update") or {}).get("avg_launch_us"))'
timeout -k 10 200 $B --config C4 2>/dev/null | python -c "$P" C4_default
PPOAF_WS=0 timeout -k 10 200 $B --config C4 2>/dev/null | python -c "$P" C4_chain_fused_tail
PPOAF_WS=0 PPOAF_FUSED_TAIL=0 timeout -k 10 200 $B --config C4 2>/dev/null | python -c "$P" C4_chain_3launch
timeout -k 10 200 $B --config C2 2>/dev/null | python -c "$P" C2_default
PPOAF_XCD_PER_NETWORK=1 timeout -k 10 200 $B --config C2 2>/dev/null | python -c "$P" C2_one_xcd_per_network
timeout -k 10 200 $B --config C3 2>/dev/null | python -c "$P" C3_default
PPOAF_FUSED_TAIL=0 timeout -k 10 200 $B --config C3 2>/dev/null | python -c "$P" C3_3launch
