R=$GRAFT_REPO_ROOT
for i in 1 2 3; do timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 20 --warmup 3 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("run", d["value"], d["ms_per_step"], d["stages_ms_per_launch"])'; done
