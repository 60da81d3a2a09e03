#!/bin/bash
# GPU box: randomised differential campaign (engine vs oracle) and soak runs on the final code of the round; stops at the first failure.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_campaign; mkdir -p $O; cd $R
BUDGET_S=${BUDGET_S:-600} SEED=${SEED:-41} timeout -k 10 900 python3 profiles/tools/parity_fuzz.py > $O/fuzz.txt 2> $O/fuzz.err || { tail -3 $O/fuzz.txt; tail -3 $O/fuzz.err; exit 1; }
tail -1 $O/fuzz.txt
STEPS=2500 timeout -k 10 300 python3 profiles/tools/soak.py > $O/soak_slabs.txt 2> $O/soak.err || { tail -3 $O/soak_slabs.txt; exit 1; }; tail -1 $O/soak_slabs.txt
PLAN=fused_up STEPS=1500 timeout -k 10 300 python3 profiles/tools/soak.py > $O/soak_fused.txt 2>> $O/soak.err || { tail -3 $O/soak_fused.txt; exit 1; }; tail -1 $O/soak_fused.txt
