#!/bin/bash
# Round 3, Stage A: the row sweeps (plan fused_up, plan pairs) against plan slabs at 16 / 32 / 48 frames per
# launch on ONE box -- does a sweep with 4.8-7.2 waves per SIMD issue better than at 2.4?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_stageA; mkdir -p $O
for B in 16 32 48; do for plan in slabs fused_up pairs; do
  timeout -k 10 240 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 20 --batch $B --chunk $B --plan $plan > $O/${plan}_b$B.json 2> $O/${plan}_b$B.err || { echo "$plan b$B failed"; continue; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); s=d["stages_ms_per_launch"]; print(sys.argv[2], "B", sys.argv[3], "pairs/s", d["value"], "ms/step", d["ms_per_step"], "ms/16", round(d["ms_per_step"]*16/int(sys.argv[3]),3), {k: round(v,3) for k,v in s.items()})' $O/${plan}_b$B.json $plan $B | tee -a $O/summary.txt
done; done
