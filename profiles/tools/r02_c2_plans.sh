# headline configuration, plan slabs against plan fused_up on one box (bench defaults otherwise)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rv16; mkdir -p $O
for rep in 1 2 3 4 5; do for plan in slabs fused_up; do
  timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 50 --plan $plan > $O/x.json 2> $O/x.err || { echo failed; continue; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["aggregate"], d["stages_ms_per_launch"]["wta"])' $O/x.json $plan
done; done
